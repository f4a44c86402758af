// Device-side pieces of the mask-free dropout (csrc/dropout.hip), shared with the kernels that fold a dropout into their own
// pass (BatchNorm's forward apply, csrc/mlp.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace fr {

// Philox4x32-10 (Salmon et al., SC'11): 4 x 32 random bits per (counter, key).
__device__ __forceinline__ uint4 philox4x32(uint4 c, uint2 k) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += 0x9E3779B9u;
        k.y += 0xBB67AE85u;
    }
    return c;
}

// The keep factors (0 or `scale`) of the four elements of group g = (offset + element index) / 4 under call counter `ctr`.
__device__ __forceinline__ float4 drop_keep4(unsigned long long seed, unsigned long long ctr, unsigned long long g, unsigned thr,
                                             float scale) {
    const uint4 r = philox4x32(make_uint4((unsigned)g, (unsigned)(g >> 32), (unsigned)ctr, (unsigned)(ctr >> 32)),
                               make_uint2((unsigned)seed, (unsigned)(seed >> 32)));
    return make_float4(r.x >= thr ? scale : 0.f, r.y >= thr ? scale : 0.f, r.z >= thr ? scale : 0.f, r.w >= thr ? scale : 0.f);
}

// state = {call counter, ticket}.  Every workgroup takes the counter through ONE load of its first thread; when `tick` is set
// that thread then draws a ticket whose increment depends on the loaded value (so the load has completed), and the holder of
// the last ticket -- every workgroup has read the counter by then -- advances it and resets the tickets.  Called by all
// threads of a 1-D launch; `sh` is one shared word.  Returns the counter value of this launch.
// (`first`, `nblocks`: which workgroup records `used_out` and how many take a ticket -- a 1-D launch by default)
__device__ __forceinline__ unsigned long long drop_counter_enter(const unsigned long long* __restrict__ ctr_src,
                                                                 unsigned long long* __restrict__ used_out,
                                                                 unsigned long long* __restrict__ tick,
                                                                 unsigned long long* sh, const bool first = blockIdx.x == 0,
                                                                 const unsigned nblocks = gridDim.x) {
    if (threadIdx.x == 0) {
        const unsigned long long c = __hip_atomic_load(ctr_src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *sh = c;
        if (used_out && first) *used_out = c;
        if (tick) {
            const unsigned long long t = atomicAdd(&tick[1], 1ull + (c >> 63));
            if (t == nblocks - 1) {
                tick[1] = 0ull;
                tick[0] = c + 1ull;
            }
        }
    }
    __syncthreads();
    return *sh;
}

// keep <=> 32 random bits >= threshold
inline unsigned drop_threshold(float p) {
    const double t = (double)p * 4294967296.0;
    return t >= 4294967295.0 ? 0xFFFFFFFFu : (unsigned)t;
}

}  // namespace fr
