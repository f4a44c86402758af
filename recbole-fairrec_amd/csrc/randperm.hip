// torch.randperm(n) on the device, bit-exact with the CPU generator's result ("shuffles on device").
//
// Replaces: the epoch shuffle of the training loader, recbole/data/interaction.py:293-297 (`torch.randperm(self.length)`
// behind Interaction.shuffle, called by AbstractDataLoader.__iter__, abstract_dataloader.py:81-84).  The third-party
// arithmetic underneath is ATen's randperm_cpu (aten/src/ATen/native/TensorFactories.cpp; n < 2^32 / 20):
//
//     r[i] = i;   for i in 0 .. n-2:  z = mt19937() % (n - i);  swap(r[i], r[i + z])
//
// a sequential Fisher-Yates over single 32-bit outputs of torch's CPU generator (at::mt19937, the standard MT19937).  On a
// host core that loop is a chain of cache misses: 30 ns per element, 270 ms for the 8.4 M interactions of a 1024-step epoch
// of BASELINE.json configs[1] -- seven times the epoch's training time on the GPU.  The same permutation without the chain:
//
//   * the swap TARGETS t_i = i + z_i depend on the generator only (kernel 1: one workgroup walks the MT recurrence, three
//     barrier-separated phases per block of 624 words, and streams the raw words out; tempering and the modulo are done by
//     the wide kernel behind it);
//   * position i is final after step i, and holds what position t_i held just before: the value the LAST EARLIER step that
//     targeted t_i put there -- that step's own "value at my position before my step" -- or t_i itself if nobody did.
//     With G(q) = the value at position q just before step q:   G(q) = G(last step that targeted q), or q if none,
//     i.e. the root of q in the forest parent(q) = max { i' < q : t_i' = q };   r[i] = G(pred_i) or t_i, pred_i = the last
//     step before i with the same target;   r[i] = G(i) for a self swap;   r[n-1] = G(n-1).
//     Steps that share a target are few (about ln(n / (n - q)) for position q), so they are kept as a linked list per
//     target built with one atomic exchange per step (kernel 2) and walked by every step of the list (kernel 3); the parent
//     chains are as short and are walked directly.  No sort, no scan, nothing sequential but the generator.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.hpp"
#include "kernels.hpp"

namespace fr {

static constexpr int RP_MT_N = 624, RP_MT_M = 397, RP_MT_THREADS = 256;

__device__ __forceinline__ uint32_t rp_mix(uint32_t a, uint32_t b) {
    const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

__device__ __forceinline__ uint32_t rp_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// kernel 1: the generator's next n_draws words, UNTEMPERED, in draw order; the state (key[624], pos) is left where the draws
// put it.  One workgroup of four waves (one per SIMD of a CU): a block of 624 words is three barrier-separated phases --
// [0, 227) reads old words only, [227, 454) reads what the first phase wrote, [454, 624) what the second wrote (word 623
// also needs new[0]) -- and is streamed out while the next one is formed.  Everything per draw that is NOT sequential
// (tempering, the modulo, the atomics) is left to the wide kernels behind it.
__global__ __launch_bounds__(RP_MT_THREADS) void randperm_words_kernel(uint32_t* __restrict__ state, uint32_t* __restrict__ raw,
                                                                       long long n_draws) {
    __shared__ uint32_t buf[2][RP_MT_N];
    const int tid = threadIdx.x;
    int cur = 0;
    for (int k = tid; k < RP_MT_N; k += RP_MT_THREADS) buf[0][k] = state[k];
    const int pos0 = (int)state[RP_MT_N];
    __syncthreads();
    constexpr int D = RP_MT_N - RP_MT_M;     // 227
    // what is left of the block the state holds
    const int first = n_draws < (long long)(RP_MT_N - pos0) ? (int)n_draws : RP_MT_N - pos0;
    for (int k = tid; k < first; k += RP_MT_THREADS) raw[k] = buf[0][pos0 + k];
    long long done = first;
    int pos = pos0 + first;
    // whole blocks: every word goes out from the register it was formed in (no second pass over the block)
    while (done < n_draws) {
        const uint32_t* o = buf[cur];
        uint32_t* nw = buf[cur ^ 1];
        const long long room = n_draws - done;        // words of this block that are draws: min(room, 624)
        uint32_t* dst = raw + done;
        if (tid < D) {
            const uint32_t v = o[tid + RP_MT_M] ^ rp_mix(o[tid], o[tid + 1]);
            nw[tid] = v;
            if (tid < room) dst[tid] = v;
        }
        __syncthreads();
        if (tid < D) {
            const uint32_t v = nw[tid] ^ rp_mix(o[tid + D], o[tid + D + 1]);
            nw[tid + D] = v;
            if (tid + D < room) dst[tid + D] = v;
        }
        __syncthreads();
        if (tid < RP_MT_N - 1 - 2 * D) {
            const uint32_t v = nw[tid + D] ^ rp_mix(o[tid + 2 * D], o[tid + 2 * D + 1]);
            nw[tid + 2 * D] = v;
            if (tid + 2 * D < room) dst[tid + 2 * D] = v;
        } else if (tid == RP_MT_N - 1 - 2 * D) {
            const uint32_t v = nw[RP_MT_M - 1] ^ rp_mix(o[RP_MT_N - 1], nw[0]);
            nw[RP_MT_N - 1] = v;
            if (RP_MT_N - 1 < room) dst[RP_MT_N - 1] = v;
        }
        __syncthreads();
        cur ^= 1;
        const int take = room < (long long)RP_MT_N ? (int)room : RP_MT_N;
        done += take;
        pos = take;
    }
    for (int k = tid; k < RP_MT_N; k += RP_MT_THREADS) state[k] = buf[cur][k];
    if (tid == 0) state[RP_MT_N] = (uint32_t)pos;
}

__global__ __launch_bounds__(256) void randperm_init_kernel(int32_t* __restrict__ parent, int32_t* __restrict__ head, long long n) {
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q < n) {
        parent[q] = -1;
        head[q] = -1;
    }
}

// kernel 2: t[i] = i + temper(word i) % (n - i), in place of the word; step i joins the list of its target (arrival order; the
// walk below orders by step) and bids for parent(target)
__global__ __launch_bounds__(256) void randperm_link_kernel(int32_t* __restrict__ t, int32_t* __restrict__ parent,
                                                            int32_t* __restrict__ head, int32_t* __restrict__ next, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n - 1) return;
    const uint32_t x = rp_temper((uint32_t)t[i]);
    const int32_t p = (int32_t)(i + (long long)(x % (uint32_t)(n - i)));      // n - i <= n < 2^31
    t[i] = p;
    if (p == (int32_t)i) return;                       // a self swap moves nothing
    atomicMax(&parent[p], (int32_t)i);
    next[i] = atomicExch(&head[p], (int32_t)i);
}

__device__ __forceinline__ int32_t rp_root(const int32_t* __restrict__ parent, int32_t q) {
    int32_t par;
    while ((par = parent[q]) >= 0) q = par;            // parent(q) < q: the walk ends
    return q;
}

// kernel 3: the permutation
__global__ __launch_bounds__(256) void randperm_resolve_kernel(const int32_t* __restrict__ t, const int32_t* __restrict__ parent,
                                                               const int32_t* __restrict__ head, const int32_t* __restrict__ next,
                                                               int64_t* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (i == n - 1) {
        out[i] = rp_root(parent, (int32_t)i);
        return;
    }
    const int32_t p = t[i];
    if (p == (int32_t)i) {
        out[i] = rp_root(parent, (int32_t)i);
        return;
    }
    int32_t pred = -1;                                  // the last step before i that targeted p
    for (int32_t j = head[p]; j >= 0; j = next[j])
        if (j < (int32_t)i && j > pred) pred = j;
    out[i] = pred >= 0 ? rp_root(parent, pred) : p;
}

}  // namespace fr

using namespace fr;

extern "C" size_t fr_randperm_workspace_bytes(int64_t n) { return n < 1 ? 0 : 4 * align_up((size_t)n * sizeof(int32_t), 256); }

extern "C" int fr_randperm(uint32_t* state, int64_t n, int64_t* out, void* ws, size_t ws_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FR_CHECK_ARG(state && n >= 0, "fr_randperm: null state / negative n");
    // (ATen takes another algorithm from 2^32 / 20 elements on; the targets are held as int32)
    FR_CHECK_ARG(n < (int64_t)(UINT32_MAX / 20), "fr_randperm: n = %lld is beyond the generator-per-element form", (long long)n);
    if (n == 0) return FR_OK;
    FR_CHECK_ARG(out && ws && ws_bytes >= fr_randperm_workspace_bytes(n), "fr_randperm: output / workspace");
    const size_t stride = align_up((size_t)n * sizeof(int32_t), 256);
    int32_t* t = reinterpret_cast<int32_t*>(ws);
    int32_t* parent = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(ws) + stride);
    int32_t* head = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(ws) + 2 * stride);
    int32_t* next = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(ws) + 3 * stride);
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(randperm_init_kernel, dim3(blocks), dim3(256), 0, stream, parent, head, (long long)n);
    if (n > 1) {
        hipLaunchKernelGGL(randperm_words_kernel, dim3(1), dim3(RP_MT_THREADS), 0, stream, state, reinterpret_cast<uint32_t*>(t),
                           (long long)(n - 1));
        hipLaunchKernelGGL(randperm_link_kernel, dim3(blocks), dim3(256), 0, stream, t, parent, head, next, (long long)n);
    }
    hipLaunchKernelGGL(randperm_resolve_kernel, dim3(blocks), dim3(256), 0, stream, t, parent, head, next, out, (long long)n);
    FR_CHECK_LAUNCH();
    return FR_OK;
}
