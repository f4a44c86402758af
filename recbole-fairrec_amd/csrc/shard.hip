// Row-sharded multi-GPU support: tables are split by `row mod G` (owner) / `row div G` (local row).
// The data path between ranks is RCCL all-to-all on FIXED-capacity buffers ([G, cap] slots, -1 = padding), so
// a step needs no host synchronisation and can be captured in a hipGraph; overflowing a bucket sets an error bit.
//
// The reference is single-device (no collectives at all); these kernels implement the
// exchange plan of SURVEY.md §8-e around torch.distributed.all_to_all_single.
#include "common.hpp"
#include "kernels.hpp"

namespace fr {

static constexpr int BUCKET_THREADS = 1024;
static constexpr int MAX_OWNERS = 16;

__device__ __forceinline__ int bucket_scan_1024(int x, int* scratch, int& total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int inc = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int y = __shfl_up(inc, o, 64);
        if (lane >= o) inc += y;
    }
    __syncthreads();   // scratch may still be read by the previous call
    if (lane == 63) scratch[wid] = inc;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const int v = scratch[w];
        woff += w < wid ? v : 0;
        tot += v;
    }
    total = tot;
    return woff + inc - x;
}

// Stable partition of the batch positions by owner rank.
//   send_ids[o*cap + k] = local row (idx div G) of the k-th position whose owner is o, -1 beyond the bucket's fill
//   slot_of[pos]        = o*cap + k  (where the answer for `pos` will sit in every [G, cap, ...] reply), -1 on overflow
//   counts[o]           = bucket fill
__global__ __launch_bounds__(BUCKET_THREADS) void bucket_by_owner_kernel(const int64_t* __restrict__ idx, int M, int G,
                                                                         int cap, int64_t* __restrict__ send_ids,
                                                                         int32_t* __restrict__ slot_of,
                                                                         int32_t* __restrict__ counts, uint32_t* err) {
    extern __shared__ __align__(16) unsigned char smem[];
    int* cnt = reinterpret_cast<int*>(smem);            // [G][1024]
    int* scratch = cnt + G * BUCKET_THREADS;            // [32]
    const int tid = threadIdx.x;
    const int C = (M + BUCKET_THREADS - 1) / BUCKET_THREADS;
    const int lo = tid * C, hi = min(lo + C, M);
    for (int o = 0; o < G; ++o) cnt[o * BUCKET_THREADS + tid] = 0;
    for (int j = tid; j < G * cap; j += BUCKET_THREADS) send_ids[j] = -1;
    bool bad = false;
    for (int j = lo; j < hi; ++j) {
        long long r = idx[j];
        if (r < 0) {
            bad = true;
            r = 0;
        }
        cnt[(int)(r % G) * BUCKET_THREADS + tid] += 1;
    }
    for (int o = 0; o < G; ++o) {
        int total;
        const int mine = cnt[o * BUCKET_THREADS + tid];
        const int ex = bucket_scan_1024(mine, scratch, total);
        cnt[o * BUCKET_THREADS + tid] = ex;
        if (tid == 0) {
            counts[o] = total < cap ? total : cap;
            if (total > cap && err) atomicOr(err, FR_DEV_ERR_BUCKET_OVERFLOW);
        }
    }
    __syncthreads();   // the -1 fill (other threads' stores) is complete before the real ids overwrite it
    for (int j = lo; j < hi; ++j) {
        long long r = idx[j];
        if (r < 0) r = 0;
        const int o = (int)(r % G);
        const int k = cnt[o * BUCKET_THREADS + tid]++;
        if (k < cap) {
            send_ids[o * cap + k] = r / G;
            slot_of[j] = o * cap + k;
        } else {
            slot_of[j] = -1;
        }
    }
    if (bad && err) atomicOr(err, FR_DEV_ERR_INDEX_RANGE);
}

// out[j, :] = slot >= 0 ? src[slot_of[j], :] : 0        (replies in [G*cap, D] slot order -> batch order)
__global__ __launch_bounds__(256) void unbucket_rows_kernel(const float* __restrict__ src,
                                                            const int32_t* __restrict__ slot_of, int M, int D,
                                                            float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= M) return;
    const int s = uniform(slot_of[j]);
    for (int d = lane; d < D; d += 64) out[(size_t)j * D + d] = s >= 0 ? src[(size_t)s * D + d] : 0.f;
}

// dst[slot_of[j], :] = scale[j] * src[j, :]   (batch order -> [G*cap, D] slot order; dst pre-zeroed by the caller)
__global__ __launch_bounds__(256) void bucket_rows_kernel(const float* __restrict__ src,
                                                          const float* __restrict__ scale,
                                                          const int32_t* __restrict__ slot_of, int M, int D,
                                                          float* __restrict__ dst) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= M) return;
    const int s = uniform(slot_of[j]);
    if (s < 0) return;
    const float c = scale ? scale[j] : 1.f;
    for (int d = lane; d < D; d += 64) dst[(size_t)s * D + d] = c * src[(size_t)j * D + d];
}

}  // namespace fr

using namespace fr;

extern "C" int fr_bucket_by_owner(const int64_t* idx, int64_t M, int32_t G, int32_t cap, int64_t* send_ids,
                                  int32_t* slot_of, int32_t* counts, uint32_t* err_flag, void* stream_) {
    FR_CHECK_ARG(idx && send_ids && slot_of && counts, "fr_bucket_by_owner: null pointer");
    FR_CHECK_ARG(M >= 1 && M <= 65536 && G >= 1 && G <= MAX_OWNERS && cap >= 1, "fr_bucket_by_owner: bad size");
    const size_t lds = ((size_t)G * BUCKET_THREADS + 64) * sizeof(int);
    static bool attr_set = false;
    if (!attr_set) {
        FR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(bucket_by_owner_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)(((size_t)MAX_OWNERS * BUCKET_THREADS + 64) * sizeof(int))));
        attr_set = true;
    }
    hipStream_t stream = (hipStream_t)stream_;
    ProfScope prof(K_BUCKET, stream);
    FR_LAUNCH(prof, bucket_by_owner_kernel, dim3(1), dim3(BUCKET_THREADS), lds, stream, idx, (int)M, (int)G,
                       (int)cap, send_ids, slot_of, counts, err_flag);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_unbucket_rows(const float* src, const int32_t* slot_of, int64_t M, int32_t dim, float* out,
                                void* stream_) {
    FR_CHECK_ARG(src && slot_of && out && M >= 0 && dim >= 1, "fr_unbucket_rows: bad argument");
    if (M == 0) return FR_OK;
    hipStream_t stream = (hipStream_t)stream_;
    ProfScope prof(K_UNBUCKET, stream);
    FR_LAUNCH(prof, unbucket_rows_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream, src, slot_of, (int)M,
                       (int)dim, out);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_bucket_rows(const float* src, const float* scale, const int32_t* slot_of, int64_t M, int32_t dim,
                              float* dst, void* stream_) {
    FR_CHECK_ARG(src && slot_of && dst && M >= 0 && dim >= 1, "fr_bucket_rows: bad argument");
    if (M == 0) return FR_OK;
    hipStream_t stream = (hipStream_t)stream_;
    ProfScope prof(K_BUCKET_ROWS, stream);
    FR_LAUNCH(prof, bucket_rows_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream, src, scale, slot_of,
                       (int)M, (int)dim, dst);
    FR_CHECK_LAUNCH();
    return FR_OK;
}
