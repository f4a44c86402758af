// Batch-index sort + segmentation: one 1024-thread workgroup per index list, keys held in LDS.
//
// Replaces torch.unique(return_inverse=True) (focf.py:77-78, nfcf.py:79-80) and the implicit
// duplicate-row accumulation of embedding_dense_backward: the backward/Adam kernels consume the
// segments so that duplicate rows are reduced in ascending batch order (the order the reference's
// CPU index_put_/embedding backward accumulates in) -- deterministic, no atomics.
//
// Keys are (row id << 32 | batch position), so the bitonic network needs no stability.
#include "common.hpp"
#include "kernels.hpp"

namespace fr {

static constexpr int SORT_THREADS = 1024;

__device__ __forceinline__ int block_exclusive_scan_1024(int x, int* scratch /*>=17 ints*/, int& total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int inc = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int y = __shfl_up(inc, o, 64);
        if (lane >= o) inc += y;
    }
    if (lane == 63) scratch[wid] = inc;
    __syncthreads();
    if (wid == 0) {
        int w = lane < 16 ? scratch[lane] : 0;
        int winc = w;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            int y = __shfl_up(winc, o, 64);
            if (lane >= o) winc += y;
        }
        if (lane < 16) scratch[lane] = winc - w;  // exclusive wave offsets
        if (lane == 15) scratch[16] = winc;
    }
    __syncthreads();
    total = scratch[16];
    return scratch[wid] + inc - x;
}

__global__ __launch_bounds__(SORT_THREADS) void sort_segments_kernel(SortJob job0, SortJob job1, int M, int P,
                                                                     uint32_t* err) {
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);
    int* scratch = reinterpret_cast<int*>(smem + (size_t)P * 8);
    float* fscratch = reinterpret_cast<float*>(scratch + 32);

    const SortJob job = blockIdx.x == 0 ? job0 : job1;
    const int tid = threadIdx.x;

    bool bad = false;
    for (int j = tid; j < P; j += SORT_THREADS) {
        unsigned long long k = ~0ull;
        if (j < M) {
            long long r = job.idx[j];
            if (r < 0 || r >= job.n_rows) {
                bad = true;
                r = 0;
            }
            k = ((unsigned long long)r << 32) | (unsigned)j;
        }
        keys[j] = k;
    }
    if (bad && err) atomicOr(err, FR_DEV_ERR_INDEX_RANGE);

    // optional min/max of a float column (the sensitive attribute): the group of a row is its rank
    // among the values present in the batch (focf.py:77)
    if (job.aux) {
        float lo = INFINITY, hi = -INFINITY;
        for (int j = tid; j < M; j += SORT_THREADS) {
            float a = job.aux[j];
            lo = fminf(lo, a);
            hi = fmaxf(hi, a);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo = fminf(lo, __shfl_xor(lo, o, 64));
            hi = fmaxf(hi, __shfl_xor(hi, o, 64));
        }
        if ((tid & 63) == 0) {
            fscratch[tid >> 6] = lo;
            fscratch[16 + (tid >> 6)] = hi;
        }
    }
    __syncthreads();
    if (job.aux && tid == 0) {
        float lo = fscratch[0], hi = fscratch[16];
        for (int w = 1; w < 16; ++w) {
            lo = fminf(lo, fscratch[w]);
            hi = fmaxf(hi, fscratch[16 + w]);
        }
        job.aux_minmax[0] = lo;
        job.aux_minmax[1] = hi;
    }

    // bitonic sort, ascending
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < (P >> 1); i += SORT_THREADS) {
                int a = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                int b = a + j;
                unsigned long long ka = keys[a], kb = keys[b];
                bool up = (a & k) == 0;
                if ((ka > kb) == up) {
                    keys[a] = kb;
                    keys[b] = ka;
                }
            }
            __syncthreads();
        }
    }

    // segment heads + exclusive scan
    const int C = P / SORT_THREADS;
    const int base = tid * C;
    int heads = 0;
    for (int q = 0; q < C; ++q) {
        int j = base + q;
        if (j < M) {
            unsigned r = (unsigned)(keys[j] >> 32);
            heads += (j == 0 || (unsigned)(keys[j - 1] >> 32) != r) ? 1 : 0;
        }
    }
    int total;
    int seg = block_exclusive_scan_1024(heads, scratch, total) - 1;  // index of the segment open at `base`
    for (int q = 0; q < C; ++q) {
        int j = base + q;
        if (j < M) {
            unsigned long long k = keys[j];
            unsigned r = (unsigned)(k >> 32);
            bool head = (j == 0 || (unsigned)(keys[j - 1] >> 32) != r);
            if (head) {
                ++seg;
                job.seg_start[seg] = j;
                job.seg_row[seg] = (int)r;
            }
            int b = (int)(unsigned)k;
            job.perm[j] = b;
            if (job.seg_of) job.seg_of[b] = seg;
        }
    }
    if (tid == 0) {
        job.seg_start[total] = M;
        job.n_seg[0] = total;
    }
}

int sort_pow2(int64_t M) {
    int P = 2 * SORT_THREADS;
    while (P < M) P <<= 1;
    return P;
}

int launch_sort(const SortJob& a, const SortJob* b, int64_t M, uint32_t* err, hipStream_t stream) {
    FR_CHECK_ARG(M >= 0 && M <= FR_SORT_MAX, "sort: M=%lld exceeds FR_SORT_MAX=%d", (long long)M, FR_SORT_MAX);
    const int P = sort_pow2(M);
    const size_t lds = (size_t)P * 8 + 64 * sizeof(int);
    static bool attr_set = false;
    if (!attr_set) {
        FR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sort_segments_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(sort_segments_kernel, dim3(b ? 2 : 1), dim3(SORT_THREADS), lds, stream, a, b ? *b : a, (int)M,
                       P, err);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

}  // namespace fr

extern "C" int fr_sort_segments(const int64_t* idx, int64_t M, int64_t n_rows, int32_t* perm, int32_t* seg_start,
                                int32_t* seg_row, int32_t* seg_of, int32_t* n_seg, uint32_t* err_flag, void* stream) {
    FR_CHECK_ARG(idx && perm && seg_start && seg_row && n_seg, "fr_sort_segments: null pointer");
    fr::SortJob j{idx, n_rows, perm, seg_start, seg_row, seg_of, n_seg, nullptr, nullptr};
    return fr::launch_sort(j, nullptr, M, err_flag, (hipStream_t)stream);
}
