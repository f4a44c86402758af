// Negative sampler on the device, bit-exact with the reference's host sampler.
//
// Replaces: recbole/sampler/sampler.py:240-241 Sampler._uni_sampling = np.random.randint(1, item_num, n), and the
// rejection loop of AbstractSampler.sample_by_key_ids (:145-197).  The third-party arithmetic underneath is numpy's
// legacy RandomState: MT19937 (mt19937_seed / mt19937_gen) and, for the int64 default dtype on a range below 2^32,
// masked rejection on single 32-bit outputs (buffered_bounded_masked_uint32).  The generator state lives in device
// memory in numpy's own layout (key[624], pos), so the stream can be handed to and taken back from
// np.random.set_state / get_state at any point.
//
// One workgroup does a whole call: the MT recurrence is sequential in blocks of 624 words (each block = four
// barrier-separated parallel phases), the accept/reject filter and the "which positions collide with the user's
// used-set" compaction are block-wide ordered scans, and the re-draw rounds loop inside the kernel until no position
// is left -- the number of rounds is data dependent, and the stream position after the call must be exact before the
// next batch draws from it, so the loop can not be cut short on the host side without a sync.
#include "common.hpp"
#include "kernels.hpp"

namespace fr {

static constexpr int MT_N = 624, MT_M = 397;
static constexpr int SAMPLER_THREADS = 1024;

__device__ __forceinline__ uint32_t mt_mix(uint32_t a, uint32_t b) {
    const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// new state block from the old one (mt19937_gen's three loops as parallel phases: [0,227) reads old words only,
// [227,454) and [454,623) read words the previous phase produced, word 623 reads new[0] and new[396])
__device__ __forceinline__ void mt_twist(const uint32_t* __restrict__ o, uint32_t* __restrict__ n, int t) {
    if (t < MT_N - MT_M) n[t] = o[t + MT_M] ^ mt_mix(o[t], o[t + 1]);
    __syncthreads();
    if (t >= MT_N - MT_M && t < 2 * (MT_N - MT_M)) n[t] = n[t - (MT_N - MT_M)] ^ mt_mix(o[t], o[t + 1]);
    __syncthreads();
    if (t >= 2 * (MT_N - MT_M) && t < MT_N - 1) n[t] = n[t - (MT_N - MT_M)] ^ mt_mix(o[t], o[t + 1]);
    __syncthreads();
    if (t == MT_N - 1) n[t] = n[MT_M - 1] ^ mt_mix(o[t], n[0]);
    __syncthreads();
}

// ordered block-wide exclusive scan of one flag per thread; returns the rank, `total` = number of set flags
__device__ __forceinline__ int flag_scan(bool flag, int* wave_cnt, int& total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const unsigned long long b = __ballot(flag);
    const int in_wave = __popcll(b & ((1ull << lane) - 1ull));
    __syncthreads();   // wave_cnt may still be read by the previous call
    if (lane == 0) wave_cnt[wid] = __popcll(b);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SAMPLER_THREADS / 64; ++w) {
        const int c = wave_cnt[w];
        off += w < wid ? c : 0;
        tot += c;
    }
    total = tot;
    return off + in_wave;
}

__device__ __forceinline__ bool used_contains(const int32_t* __restrict__ items, long long lo, long long hi, int v) {
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        const int x = items[mid];
        if (x == v) return true;
        if (x < v) lo = mid + 1;
        else hi = mid;
    }
    return false;
}

// Two calling modes: one call over `total` positions keyed by key_ids[i % n_keys]  (call_offsets == nullptr), or a
// SEQUENCE of calls c = 0..n_calls-1, call c drawing positions [call_offsets[c], call_offsets[c+1]) for the single key
// call_keys[c] -- each call completes its re-draw rounds before the next one draws, exactly like consecutive
// sample_by_user_ids calls on one numpy stream (the evaluation loader samples user by user, general_dataloader.py:141-146).
__global__ __launch_bounds__(SAMPLER_THREADS) void sample_negatives_kernel(
    uint32_t* __restrict__ state, long long low, uint32_t span, uint32_t mask, const int64_t* __restrict__ key_ids,
    long long n_keys, long long total_all, const int64_t* __restrict__ call_keys,
    const int64_t* __restrict__ call_offsets, long long n_calls, const int64_t* __restrict__ used_indptr,
    const int32_t* __restrict__ used_items, long long n_users, int64_t* __restrict__ out_all,
    int32_t* __restrict__ list_a, int32_t* __restrict__ list_b, int32_t* __restrict__ rounds_out, uint32_t* err) {
    __shared__ uint32_t mt[2][MT_N];
    __shared__ int wave_cnt[SAMPLER_THREADS / 64];
    __shared__ int s_last;
    const int t = threadIdx.x;
    int cur = 0;
    if (t < MT_N) mt[0][t] = state[t];
    int pos = (int)state[MT_N];
    __syncthreads();

    if (span == 0) {   // numpy: a one-value range consumes nothing
        const long long n_out = call_offsets ? call_offsets[n_calls] : total_all;
        for (long long e = t; e < n_out; e += SAMPLER_THREADS) out_all[e] = low;
        if (t == 0 && rounds_out) rounds_out[0] = 1;
        return;
    }

    int rounds = 0;
    const long long calls = call_offsets ? n_calls : 1;
    for (long long call = 0; call < calls; ++call) {
    const long long o0 = call_offsets ? call_offsets[call] : 0;
    const long long total = call_offsets ? call_offsets[call + 1] - o0 : total_all;
    int64_t* __restrict__ out = out_all + o0;
    const long long call_key = call_keys ? call_keys[call] : -1;
    const int32_t* list = nullptr;      // positions to (re)draw, ascending; nullptr = all of [0, total)
    int32_t* next = list_a;
    long long need = total;
    while (need > 0) {
        // ---- draw `need` values, in position order, from the continuing stream ----
        long long produced = 0;
        while (produced < need) {
            if (pos == MT_N) {
                mt_twist(mt[cur], mt[cur ^ 1], t);
                cur ^= 1;
                pos = 0;
            }
            uint32_t v = 0;
            bool acc = false;
            if (t >= pos && t < MT_N) {
                v = mt_temper(mt[cur][t]) & mask;
                acc = v <= span;
            }
            int cnt;
            const int k = flag_scan(acc, wave_cnt, cnt);
            const long long remaining = need - produced;
            if (acc && k < remaining) {
                const long long e = produced + k;
                out[list ? (long long)list[e] : e] = low + (long long)v;
                if (k == remaining - 1) s_last = t;     // the draw that yields the last value ends the consumption
            }
            __syncthreads();
            if (cnt >= remaining) {
                pos = s_last + 1;
                produced = need;
            } else {
                pos = MT_N;
                produced += cnt;
            }
            __syncthreads();   // s_last is rewritten in the next iteration
        }
        ++rounds;
        if (!used_indptr) break;
        __threadfence_block();
        // ---- which of the positions just drawn hit their key's used-set?  (ordered compaction -> next round) ----
        long long kept = 0;
        for (long long base = 0; base < need; base += SAMPLER_THREADS) {
            const long long e = base + t;
            bool hit = false;
            long long i = 0;
            if (e < need) {
                i = list ? (long long)list[e] : e;
                const long long u = call_keys ? call_key : key_ids[i % n_keys];
                if (u < 0 || u >= n_users) {
                    if (err) atomicOr(err, FR_DEV_ERR_INDEX_RANGE);
                } else {
                    hit = used_contains(used_items, used_indptr[u], used_indptr[u + 1], (int)out[i]);
                }
            }
            int cnt;
            const int k = flag_scan(hit, wave_cnt, cnt);
            if (hit) next[kept + k] = (int32_t)i;
            kept += cnt;
        }
        __syncthreads();
        __threadfence_block();
        list = next;
        next = (next == list_a) ? list_b : list_a;
        need = kept;
    }
    __syncthreads();   // the next call reuses the collision lists
    }
    __syncthreads();
    if (t < MT_N) state[t] = mt[cur][t];
    if (t == 0) {
        state[MT_N] = (uint32_t)pos;
        if (rounds_out) rounds_out[0] = rounds;
    }
}

// mt19937_seed(state, seed): Knuth's LCG over the 624 words, pos = 624 (np.random.seed(int))
__global__ void mt19937_seed_kernel(uint32_t* __restrict__ state, uint32_t seed) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t s = seed;
    for (int i = 0; i < MT_N; ++i) {
        state[i] = s;
        s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)i + 1u;
    }
    state[MT_N] = MT_N;
}

}  // namespace fr

using namespace fr;

extern "C" int fr_mt19937_seed(uint32_t* state, uint32_t seed, void* stream_) {
    FR_CHECK_ARG(state, "fr_mt19937_seed: null state");
    hipLaunchKernelGGL(mt19937_seed_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_, state, seed);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" size_t fr_sample_negatives_workspace_bytes(int64_t total) {
    return total < 1 ? 0 : 2 * align_up((size_t)total * sizeof(int32_t), 256);
}

static int sample_launch(uint32_t* state, int64_t low, int64_t high, const int64_t* key_ids, int64_t n_keys, int64_t total,
                         const int64_t* call_keys, const int64_t* call_offsets, int64_t n_calls, int64_t max_call,
                         const int64_t* used_indptr, const int32_t* used_items, int64_t n_users, int64_t* out,
                         int32_t* rounds_out, void* ws, size_t ws_bytes, uint32_t* err_flag, hipStream_t stream) {
    int32_t *la = nullptr, *lb = nullptr;
    if (used_indptr) {
        FR_CHECK_ARG(ws && ws_bytes >= fr_sample_negatives_workspace_bytes(max_call), "fr_sample_negatives: workspace too small");
        la = (int32_t*)ws;
        lb = (int32_t*)((char*)ws + align_up((size_t)max_call * sizeof(int32_t), 256));
    }
    const uint32_t span = (uint32_t)(high - 1 - low);
    uint32_t mask = span;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    ProfScope prof(K_SAMPLE_NEG, stream);
    FR_LAUNCH(prof, sample_negatives_kernel, dim3(1), dim3(SAMPLER_THREADS), 0, stream, state, (long long)low, span, mask,
              key_ids, (long long)n_keys, (long long)total, call_keys, call_offsets, (long long)n_calls, used_indptr,
              used_items, (long long)n_users, out, la, lb, rounds_out, err_flag);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_sample_negatives(uint32_t* state, int64_t low, int64_t high, const int64_t* key_ids, int64_t n_keys,
                                   int32_t num, const int64_t* used_indptr, const int32_t* used_items, int64_t n_users,
                                   int64_t* out, int32_t* rounds_out, void* ws, size_t ws_bytes, uint32_t* err_flag,
                                   void* stream_) {
    FR_CHECK_ARG(state && out && n_keys >= 1 && num >= 1 && n_keys * (int64_t)num <= (1ll << 30),
                 "fr_sample_negatives: bad size");
    FR_CHECK_ARG(high > low && high - 1 - low < 0xffffffffll, "fr_sample_negatives: range [%lld, %lld) not below 2^32",
                 (long long)low, (long long)high);
    FR_CHECK_ARG(!used_indptr || (used_items && key_ids && n_users >= 1 && ws), "fr_sample_negatives: used-set arguments");
    const int64_t total = n_keys * (int64_t)num;
    return sample_launch(state, low, high, key_ids, n_keys, total, nullptr, nullptr, 1, total, used_indptr, used_items,
                         n_users, out, rounds_out, ws, ws_bytes, err_flag, (hipStream_t)stream_);
}

// A sequence of single-key calls on one stream: call c fills out[call_offsets[c] .. call_offsets[c+1]) for key
// call_keys[c]; `max_call` >= the longest call sizes the workspace (fr_sample_negatives_workspace_bytes(max_call)).
extern "C" int fr_sample_negatives_calls(uint32_t* state, int64_t low, int64_t high, const int64_t* call_keys,
                                         const int64_t* call_offsets, int64_t n_calls, int64_t max_call,
                                         const int64_t* used_indptr, const int32_t* used_items, int64_t n_users,
                                         int64_t* out, void* ws, size_t ws_bytes, uint32_t* err_flag, void* stream_) {
    FR_CHECK_ARG(state && out && call_keys && call_offsets && n_calls >= 1 && max_call >= 1 && max_call <= (1ll << 30),
                 "fr_sample_negatives_calls: bad argument");
    FR_CHECK_ARG(high > low && high - 1 - low < 0xffffffffll, "fr_sample_negatives_calls: range not below 2^32");
    FR_CHECK_ARG(used_indptr && used_items && n_users >= 1 && ws, "fr_sample_negatives_calls: used-set arguments");
    return sample_launch(state, low, high, nullptr, 1, 0, call_keys, call_offsets, n_calls, max_call, used_indptr, used_items,
                         n_users, out, nullptr, ws, ws_bytes, err_flag, (hipStream_t)stream_);
}
