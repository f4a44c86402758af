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
static constexpr int BUCKET_STAGE_MAX = 16384;   // id lists / slot counts up to this length are staged in LDS (64 KB)
static constexpr size_t BUCKET_LDS_MAX = 160 * 1024;

template <typename T>
__device__ __forceinline__ T bucket_scan_1024(T x, T* scratch, T& total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    T inc = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        T y = __shfl_up(inc, o, 64);
        if (lane >= o) inc += y;
    }
    __syncthreads();   // scratch may still be read by the previous call
    if (lane == 63) scratch[wid] = inc;
    __syncthreads();
    T woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const T v = scratch[w];
        woff += w < wid ? v : 0;
        tot += v;
    }
    total = tot;
    return woff + inc - x;
}

// Stable partition of the batch positions by owner rank.  Slot (o, k) of this id list sits at o*stride + offset + k of
// the exchange buffers (stride = cap, offset = 0: a [G, cap] buffer of its own; stride = T*cap, offset = t*cap: list t
// of a shared [G, T, cap] buffer).
//   send_ids[slot(o,k)] = local row (idx div G) of the k-th position whose owner is o, -1 beyond the bucket's fill
//   slot_of[pos]        = slot(o,k)  (where the answer for `pos` will sit in every reply buffer), -1 on overflow
//   counts[o]           = bucket fill
// aux != nullptr: (min, max) of the float column aux[0..M) is written, as two floats, into the int64 slot
// o*stride + aux_slot of every owner's chunk, so the batch-wide extrema ride along with the id exchange.
struct BucketJob {
    const int64_t* idx;
    int offset;          // first slot of this list inside every owner's chunk
    int32_t* slot_of;
    int32_t* counts;
    const float* aux;    // optional float column whose (min, max) goes to slot aux_slot of every chunk
    int aux_slot;
};

// LDS plan of the bucket kernel (ints): cnt [G][1024] | scratch [64] | tot [16] | ids [M] (staged) | sendst [G*cap]
// (staged send).  Staging keeps every global access coalesced: the id column is read once with unit stride, and
// slot_of / send_ids are written with unit stride from LDS at the end (a thread's own range of positions is
// contiguous, so direct stores would touch 64 cache lines per wave instruction).
// positions per thread: the next power of two of ceil(M / 1024), so that position -> (thread, offset) is a shift/mask
static __host__ __device__ inline int bucket_range(long long M) {
    int c = 1;
    while ((long long)c * BUCKET_THREADS < M) c <<= 1;
    return c;
}

struct BucketPlan {
    bool stage_ids, stage_send;
    size_t lds_bytes;
};

static BucketPlan bucket_plan(int64_t M, int G, int cap) {
    BucketPlan p;
    size_t ints = (size_t)G * BUCKET_THREADS + 64 + 16;
    p.stage_ids = M <= BUCKET_STAGE_MAX;
    if (p.stage_ids) ints += (size_t)BUCKET_THREADS * (bucket_range(M) + 1);
    p.stage_send = p.stage_ids && (size_t)G * cap <= BUCKET_STAGE_MAX && (ints + (size_t)G * cap) * 4 <= BUCKET_LDS_MAX;
    if (p.stage_send) ints += (size_t)G * cap;
    p.lds_bytes = ints * 4;
    return p;
}

// One workgroup per id list (blockIdx.x).  Row ids are < 2^31 (checked by the table kernels), so owner and local row
// come from 32-bit division.
// holes: an id of -1 is an empty position (it goes to no owner, slot -1, no error) instead of an out-of-range id.
__global__ __launch_bounds__(BUCKET_THREADS) void bucket_by_owner_kernel(BucketJob j0, BucketJob j1, int M, int G,
                                                                         int cap, int stride, int stage_ids,
                                                                         int stage_send,
                                                                         int64_t* __restrict__ send_ids,
                                                                         uint32_t* err, int holes) {
    constexpr unsigned HOLE = 0xFFFFFFFFu;
    extern __shared__ __align__(16) unsigned char smem[];
    const BucketJob job = blockIdx.x == 0 ? j0 : j1;
    const int64_t* __restrict__ idx = job.idx;
    const int offset = job.offset;
    int* cnt = reinterpret_cast<int*>(smem);            // [G][1024]
    int* scratch = cnt + G * BUCKET_THREADS;            // [64]
    int* tot = scratch + 64;                            // [16] bucket fills
    unsigned* ids = reinterpret_cast<unsigned*>(tot + 16);        // [M]: row ids, later the slots
    const int tid = threadIdx.x;
    const int C = bucket_range(M), logC = __ffs(C) - 1;
    const int Cp = C + 1;   // odd stride between the threads' ranges in `ids`: no LDS bank conflicts in the range loops
    unsigned* sendst = ids + (stage_ids ? BUCKET_THREADS * Cp : 0);   // [G*cap]: local rows in slot order
    const int lo = min(tid * C, M), hi = min(lo + C, M);
    auto pad = [&](int j) { return (j >> logC) * Cp + (j & (C - 1)); };
    // owner / local row of a row id: shift and mask when G is a power of two (1, 2, 4, 8 GPUs)
    const bool pow2 = (G & (G - 1)) == 0;
    const int logG = __ffs(G) - 1;
    auto owner = [&](unsigned r) { return pow2 ? (int)(r & (unsigned)(G - 1)) : (int)(r % (unsigned)G); };
    auto local = [&](unsigned r) { return pow2 ? r >> logG : r / (unsigned)G; };
    for (int o = 0; o < G; ++o) cnt[o * BUCKET_THREADS + tid] = 0;
    if (!stage_send)
        for (int o = 0; o < G; ++o)
            for (int k = tid; k < cap; k += BUCKET_THREADS) send_ids[o * stride + offset + k] = -1;
    bool bad = false;
    if (stage_ids) {
        for (int j = tid; j < M; j += BUCKET_THREADS) {
            const long long r64 = idx[j];
            const bool hole = holes && r64 == -1;
            const bool oob = !hole && (r64 < 0 || r64 > 0x7fffffffll);
            bad |= oob;
            ids[pad(j)] = hole ? HOLE : (oob ? 0u : (unsigned)r64);
        }
        __syncthreads();
    }
    auto id_at = [&](int j) -> unsigned {
        if (stage_ids) return ids[tid * Cp + (j - lo)];
        const long long r64 = idx[j];
        if (holes && r64 == -1) return HOLE;
        const bool oob = r64 < 0 || r64 > 0x7fffffffll;
        bad |= oob;
        return oob ? 0u : (unsigned)r64;
    };
    for (int j = lo; j < hi; ++j) {
        const unsigned r = id_at(j);
        if (r != HOLE) cnt[owner(r) * BUCKET_THREADS + tid] += 1;
    }
    auto publish = [&](int o, int total) {
        tot[o] = total < cap ? total : cap;
        job.counts[o] = total < cap ? total : cap;
        if (total > cap && err) atomicOr(err, FR_DEV_ERR_BUCKET_OVERFLOW);
    };
    if (stage_ids) {   // M <= 16384: four owners' counts share one 64-bit scan (16 bits each can not overflow)
        for (int o4 = 0; o4 < G; o4 += 4) {
            unsigned long long mine = 0;
            for (int q = 0; q < 4 && o4 + q < G; ++q)
                mine |= (unsigned long long)cnt[(o4 + q) * BUCKET_THREADS + tid] << (16 * q);
            unsigned long long total;
            const unsigned long long ex = bucket_scan_1024(mine, reinterpret_cast<unsigned long long*>(scratch), total);
            for (int q = 0; q < 4 && o4 + q < G; ++q) {
                cnt[(o4 + q) * BUCKET_THREADS + tid] = (int)((ex >> (16 * q)) & 0xffffu);
                if (tid == 0) publish(o4 + q, (int)((total >> (16 * q)) & 0xffffu));
            }
        }
    } else {
        for (int o = 0; o < G; ++o) {
            int total;
            const int mine = cnt[o * BUCKET_THREADS + tid];
            const int ex = bucket_scan_1024(mine, scratch, total);
            cnt[o * BUCKET_THREADS + tid] = ex;
            if (tid == 0) publish(o, total);
        }
    }
    __syncthreads();   // the -1 fill (other threads' stores) is complete before the real ids overwrite it
    for (int j = lo; j < hi; ++j) {
        const unsigned r = id_at(j);
        const bool hole = r == HOLE;
        const int o = hole ? 0 : owner(r);
        const int k = hole ? cap : cnt[o * BUCKET_THREADS + tid]++;
        const int slot = k < cap ? o * stride + offset + k : -1;
        if (k < cap) {
            if (stage_send) sendst[o * cap + k] = local(r);
            else send_ids[o * stride + offset + k] = (long long)local(r);
        }
        if (stage_ids) ids[tid * Cp + (j - lo)] = (unsigned)slot;
        else job.slot_of[j] = slot;
    }
    if (stage_ids) {
        __syncthreads();
        for (int j = tid; j < M; j += BUCKET_THREADS) job.slot_of[j] = (int)ids[pad(j)];
        if (stage_send)
            for (int o = 0; o < G; ++o) {
                const int fill = tot[o];
                for (int k = tid; k < cap; k += BUCKET_THREADS)
                    send_ids[o * stride + offset + k] = k < fill ? (long long)sendst[o * cap + k] : -1ll;
            }
    }
    if (bad && err) atomicOr(err, FR_DEV_ERR_INDEX_RANGE);
    if (job.aux) {
        float lo_v = INFINITY, hi_v = -INFINITY;
        for (int j = tid; j < M; j += BUCKET_THREADS) {
            const float a = job.aux[j];
            lo_v = fminf(lo_v, a);
            hi_v = fmaxf(hi_v, a);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo_v = fminf(lo_v, __shfl_xor(lo_v, o, 64));
            hi_v = fmaxf(hi_v, __shfl_xor(hi_v, o, 64));
        }
        float* fs = reinterpret_cast<float*>(scratch);
        __syncthreads();
        if ((tid & 63) == 0) {
            fs[2 * (tid >> 6)] = lo_v;
            fs[2 * (tid >> 6) + 1] = hi_v;
        }
        __syncthreads();
        if (tid < G) {
            for (int w = 0; w < BUCKET_THREADS / 64; ++w) {
                lo_v = fminf(lo_v, fs[2 * w]);
                hi_v = fmaxf(hi_v, fs[2 * w + 1]);
            }
            float* dst = reinterpret_cast<float*>(send_ids + (size_t)tid * stride + job.aux_slot);
            dst[0] = lo_v;
            dst[1] = hi_v;
        }
    }
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

// ---- item-owner-computes schedule (fairrec/sharded.py, ShardedFocfEngineV2): records to the item owners --------------
// A record chunk is [4 * cap + 1] int64: item local rows (written by the bucket kernel, -1 = empty) | user ids | rating
// bits | sst bits | the sender's (min, max) of sst as a float pair.
__global__ __launch_bounds__(256) void shard_pack_records_kernel(const int32_t* __restrict__ slot, const int64_t* __restrict__ user,
                                                                 const float* __restrict__ rating, const float* __restrict__ sst,
                                                                 int B, int cap, int64_t* __restrict__ send) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const int s = slot[b];
    if (s < 0) return;        // overflowed its bucket (error bit already set)
    send[s + cap] = user[b];
    send[s + 2 * cap] = (long long)__float_as_int(rating[b]);
    send[s + 3 * cap] = sst ? (long long)__float_as_int(sst[b]) : 0ll;
}

// what an item owner keeps of the G record chunks it received: per record slot s = g * cap + k the item row (-1 = empty),
// the user id (-1 = empty), its own position as "item slot" (-1 = empty), rating (0 when empty) and sst; per sender the
// (min, max) pair
__global__ __launch_bounds__(256) void shard_unpack_records_kernel(const int64_t* __restrict__ recv, int G, int cap,
                                                                   int64_t* __restrict__ iid, int64_t* __restrict__ uid,
                                                                   int32_t* __restrict__ islot, float* __restrict__ rating,
                                                                   float* __restrict__ sst, int64_t* __restrict__ mm) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    const int S = 4 * cap + 1;
    if (s < G) mm[s] = recv[(size_t)s * S + 4 * cap];
    if (s >= G * cap) return;
    const int g = s / cap, k = s % cap;
    const int64_t* c = recv + (size_t)g * S;
    const long long it = c[k];
    const bool held = it >= 0;
    iid[s] = it;
    uid[s] = held ? c[cap + k] : -1ll;
    islot[s] = held ? s : -1;
    rating[s] = held ? __int_as_float((int)c[2 * cap + k]) : 0.f;
    sst[s] = held ? __int_as_float((int)c[3 * cap + k]) : 0.f;
}

// Number of distinct real ids of a list, by marking a bitmap over the owner's rows (kept all-zero between calls): the count an
// item owner reports to the other ranks must not wait for the sort of the list (it is exchanged by a collective, and every
// collective of the step queues behind it).
__global__ __launch_bounds__(256) void shard_mark_distinct_kernel(const int64_t* __restrict__ ids, int n, unsigned* __restrict__ bitmap,
                                                                  int* __restrict__ count) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    const long long id = ids[s];
    if (id < 0) return;
    const unsigned bit = 1u << (id & 31);
    if (!(atomicOr(&bitmap[id >> 5], bit) & bit)) atomicAdd(count, 1);
}
__global__ __launch_bounds__(256) void shard_clear_distinct_kernel(const int64_t* __restrict__ ids, int n, unsigned* __restrict__ bitmap,
                                                                   int* __restrict__ count, float* __restrict__ out) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s == 0) out[0] = (float)*count;
    if (s >= n) return;
    const long long id = ids[s];
    if (id >= 0) bitmap[id >> 5] = 0u;
}
__global__ void shard_zero_int_kernel(int* p) { *p = 0; }

// After the owner-side fairness kernel: (fairness sum, squared-error sum) of THIS owner out of the reply tails, and the
// tails' first entry replaced by every owner's distinct-item count (gathered one step ahead): the gradient kernel sums
// them to K of the global batch.
__global__ void shard_post_fair_kernel(float* __restrict__ reply, const float* __restrict__ k_all, int G, int cap, int tail,
                                       float* __restrict__ sums) {
    const int g = threadIdx.x;
    if (g == 0) {
        sums[0] = reply[cap + 1];
        sums[1] = reply[cap + 2];
    }
    if (g < G) reply[(size_t)g * (cap + tail) + cap] = k_all[g];
}

// loss of the global batch from the all-reduced (fairness sum, squared-error sum) and the owners' distinct-item counts
__global__ void shard_loss_finish_kernel(const float* __restrict__ sums, const float* __restrict__ k_all, int G, float inv_n,
                                         float fair_weight, int fair, float* __restrict__ loss) {
    if (threadIdx.x != 0) return;
    float K = 0.f;
    for (int g = 0; g < G; ++g) K += k_all[g];
    const float mse = sums[1] * inv_n, fv = fair ? sums[0] / K : 0.f;
    loss[0] = mse + fair_weight * fv;
    loss[1] = mse;
    loss[2] = fv;
}

}  // namespace fr

using namespace fr;

extern "C" int fr_shard_pack_records(const int32_t* slot, const int64_t* user, const float* rating, const float* sst,
                                     int64_t B, int32_t cap, int64_t* send, void* stream_) {
    FR_CHECK_ARG(slot && user && rating && send && B >= 1 && cap >= 1, "fr_shard_pack_records: bad argument");
    hipLaunchKernelGGL(shard_pack_records_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, slot,
                       user, rating, sst, (int)B, (int)cap, send);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_shard_unpack_records(const int64_t* recv, int32_t G, int32_t cap, int64_t* iid, int64_t* uid,
                                       int32_t* islot, float* rating, float* sst, int64_t* mm, void* stream_) {
    FR_CHECK_ARG(recv && iid && uid && islot && rating && sst && mm && G >= 1 && cap >= 1,
                 "fr_shard_unpack_records: bad argument");
    const long long n = (long long)G * cap;
    hipLaunchKernelGGL(shard_unpack_records_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, recv,
                       (int)G, (int)cap, iid, uid, islot, rating, sst, mm);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_shard_count_distinct(const int64_t* ids, int64_t n, int64_t n_rows, uint32_t* bitmap, int32_t* count,
                                       float* out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FR_CHECK_ARG(ids && bitmap && count && out && n >= 1 && n_rows >= 1, "fr_shard_count_distinct: bad argument");
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(shard_mark_distinct_kernel, dim3(blocks), dim3(256), 0, stream, ids, (int)n, bitmap, count);
    hipLaunchKernelGGL(shard_clear_distinct_kernel, dim3(blocks), dim3(256), 0, stream, ids, (int)n, bitmap, count, out);
    hipLaunchKernelGGL(shard_zero_int_kernel, dim3(1), dim3(1), 0, stream, count);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_shard_post_fair(float* reply, const float* k_all, int32_t G, int32_t cap, float* sums, void* stream_) {
    FR_CHECK_ARG(reply && k_all && sums && G >= 1 && G <= 64 && cap >= 1, "fr_shard_post_fair: bad argument");
    hipLaunchKernelGGL(shard_post_fair_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_, reply, k_all, (int)G, (int)cap,
                       (int)FR_SHARD_TAIL, sums);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_shard_loss_finish(const float* sums, const float* k_all, int32_t G, int64_t n_global, float fair_weight,
                                    int32_t fair, float* loss, void* stream_) {
    FR_CHECK_ARG(sums && k_all && loss && G >= 1 && n_global >= 1, "fr_shard_loss_finish: bad argument");
    hipLaunchKernelGGL(shard_loss_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_, sums, k_all, (int)G,
                       1.f / (float)n_global, fair_weight, (int)fair, loss);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

static int launch_bucket(const BucketJob& a, const BucketJob* b, int64_t M, int32_t G, int32_t cap, int32_t stride,
                         int64_t* send_ids, uint32_t* err_flag, hipStream_t stream, int holes = 0) {
    const BucketPlan plan = bucket_plan(M, G, cap);
    static bool attr_set = false;
    if (!attr_set) {
        FR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(bucket_by_owner_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)BUCKET_LDS_MAX));
        attr_set = true;
    }
    ProfScope prof(K_BUCKET, stream);
    FR_LAUNCH(prof, bucket_by_owner_kernel, dim3(b ? 2 : 1), dim3(BUCKET_THREADS), plan.lds_bytes, stream, a, b ? *b : a,
              (int)M, (int)G, (int)cap, (int)stride, (int)plan.stage_ids, (int)plan.stage_send, send_ids, err_flag, holes);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

static bool bucket_args_ok(int64_t M, int32_t G, int32_t cap, int32_t stride, int32_t offset, const float* aux,
                           int32_t aux_slot) {
    return M >= 1 && M <= 65536 && G >= 1 && G <= MAX_OWNERS && cap >= 1 && offset >= 0 && stride >= offset + cap &&
           (!aux || (aux_slot >= 0 && aux_slot < stride && (aux_slot < offset || aux_slot >= offset + cap)));
}

extern "C" int fr_bucket_by_owner(const int64_t* idx, int64_t M, int32_t G, int32_t cap, int32_t stride, int32_t offset,
                                  int64_t* send_ids, int32_t* slot_of, int32_t* counts, const float* aux,
                                  int32_t aux_slot, uint32_t* err_flag, void* stream_) {
    FR_CHECK_ARG(idx && send_ids && slot_of && counts, "fr_bucket_by_owner: null pointer");
    FR_CHECK_ARG(bucket_args_ok(M, G, cap, stride, offset, aux, aux_slot), "fr_bucket_by_owner: bad size / slot");
    BucketJob a{idx, offset, slot_of, counts, aux, aux_slot};
    return launch_bucket(a, nullptr, M, G, cap, stride, send_ids, err_flag, (hipStream_t)stream_);
}

// The same for a list with empty positions (id -1: the padding of a received exchange buffer): they go to no owner.
extern "C" int fr_bucket_by_owner_sparse(const int64_t* idx, int64_t M, int32_t G, int32_t cap, int32_t stride,
                                         int32_t offset, int64_t* send_ids, int32_t* slot_of, int32_t* counts,
                                         uint32_t* err_flag, void* stream_) {
    FR_CHECK_ARG(idx && send_ids && slot_of && counts, "fr_bucket_by_owner_sparse: null pointer");
    FR_CHECK_ARG(bucket_args_ok(M, G, cap, stride, offset, nullptr, 0), "fr_bucket_by_owner_sparse: bad size / slot");
    BucketJob a{idx, offset, slot_of, counts, nullptr, 0};
    return launch_bucket(a, nullptr, M, G, cap, stride, send_ids, err_flag, (hipStream_t)stream_, 1);
}

extern "C" int fr_bucket_pair_by_owner(const int64_t* idx_a, const int64_t* idx_b, int64_t M, int32_t G, int32_t cap,
                                       int32_t stride, int32_t offset_a, int32_t offset_b, int64_t* send_ids,
                                       int32_t* slot_a, int32_t* slot_b, int32_t* counts, const float* aux,
                                       int32_t aux_slot, uint32_t* err_flag, void* stream_) {
    FR_CHECK_ARG(idx_a && idx_b && send_ids && slot_a && slot_b && counts, "fr_bucket_pair_by_owner: null pointer");
    FR_CHECK_ARG(bucket_args_ok(M, G, cap, stride, offset_a, aux, aux_slot) &&
                     bucket_args_ok(M, G, cap, stride, offset_b, aux, aux_slot) &&
                     (offset_a + cap <= offset_b || offset_b + cap <= offset_a),
                 "fr_bucket_pair_by_owner: bad size / overlapping lists");
    BucketJob a{idx_a, offset_a, slot_a, counts, nullptr, 0};
    BucketJob b{idx_b, offset_b, slot_b, counts + G, aux, aux_slot};
    return launch_bucket(a, &b, M, G, cap, stride, send_ids, err_flag, (hipStream_t)stream_);
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
