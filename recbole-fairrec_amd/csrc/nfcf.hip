// NFCF loss head: sigmoid + BCE on the scorer output and the differential-fairness regulariser.
//
// Replaces nfcf.py:73 (sigmoid), :105 (nn.BCELoss), :76-97 (get_differential_fairness: 2x torch.unique,
// 2x index_put_(accumulate), smoothed per-(item, group) mean scores, max pairwise |log ratio|, mean over items)
// and their autograd.  The scorer MLP itself runs on mlp.hip, the embeddings on table.hip.
#include "common.hpp"
#include "kernels.hpp"
#include "table.hpp"

namespace fr {

// y = last MLP layer's output AFTER its ReLU (layers.py:63-70), out = sigmoid(y).
// dy[b] = d mean(BCE) / d y[b]  with torch's BCE backward: (out - label) / max(out*(1-out), 1e-12) / B * out*(1-out)
// With `sst`: mm_part[2 * block] = (min, max) of the sensitive attribute over the block's POSITIVE rows
// (torch.unique(sst[label == 1]), nfcf.py:79; reduced by the fairness kernels).
__global__ __launch_bounds__(256) void nfcf_bce_kernel(const float* __restrict__ y, const float* __restrict__ label, int B,
                                                       float* __restrict__ out, float* __restrict__ dy,
                                                       float* __restrict__ part, const float* __restrict__ sst,
                                                       float* __restrict__ mm_part) {
    __shared__ float red[4], lo_s[4], hi_s[4];
    const int b = blockIdx.x * 256 + threadIdx.x;
    float l = 0.f;
    if (sst) {
        float lo = INFINITY, hi = -INFINITY;
        if (b < B && label[b] == 1.f) lo = hi = sst[b];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo = fminf(lo, __shfl_xor(lo, o, 64));
            hi = fmaxf(hi, __shfl_xor(hi, o, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            lo_s[threadIdx.x >> 6] = lo;
            hi_s[threadIdx.x >> 6] = hi;
        }
    }
    if (b < B) {
        const float o = 1.f / (1.f + __expf(-y[b]));
        const float t = label[b];
        // torch.nn.functional.binary_cross_entropy clamps both logs at -100
        const float lo = fmaxf(__logf(o), -100.f), l1 = fmaxf(__logf(1.f - o), -100.f);
        l = -(t * lo + (1.f - t) * l1);
        out[b] = o;
        const float s = o * (1.f - o);
        dy[b] = (o - t) / fmaxf(s, 1e-12f) / (float)B * s;
    }
    l = wave_sum(l);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = l;
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
        if (sst) {
            mm_part[2 * blockIdx.x] = fminf(fminf(lo_s[0], lo_s[1]), fminf(lo_s[2], lo_s[3]));
            mm_part[2 * blockIdx.x + 1] = fmaxf(fmaxf(hi_s[0], hi_s[1]), fmaxf(hi_s[2], hi_s[3]));
        }
    }
}

// (min, max) over the per-block pairs of nfcf_bce_kernel, by the first wave of the calling workgroup; every thread gets
// the pair after the barrier inside.
__device__ __forceinline__ float2 pos_minmax(const float* __restrict__ mm_part, int n, float2* sh) {
    if (threadIdx.x < 64) {
        float lo = INFINITY, hi = -INFINITY;
        for (int q = threadIdx.x; q < n; q += 64) {
            lo = fminf(lo, mm_part[2 * q]);
            hi = fmaxf(hi, mm_part[2 * q + 1]);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo = fminf(lo, __shfl_xor(lo, o, 64));
            hi = fmaxf(hi, __shfl_xor(hi, o, 64));
        }
        if (threadIdx.x == 0) *sh = make_float2(lo, hi);
    }
    __syncthreads();
    return *sh;
}

static constexpr int DF_THREADS = 256, DF_GROUP = 16;

// pass 1: per distinct item (16 lanes): score sums / counts of its positive rows per group; stats[k] = (S0,S1,n0,n1)
__global__ __launch_bounds__(DF_THREADS) void nfcf_df_stats_kernel(TableWs w, const float* __restrict__ out,
                                                                   const float* __restrict__ label,
                                                                   const float* __restrict__ sst,
                                                                   const float* __restrict__ mm_part, int n_mm,
                                                                   float* __restrict__ minmax,
                                                                   float4* __restrict__ stats, int* __restrict__ kpart,
                                                                   uint32_t* err) {
    const int sub = threadIdx.x & (DF_GROUP - 1), gib = threadIdx.x / DF_GROUP;
    const int k = blockIdx.x * (DF_THREADS / DF_GROUP) + gib;
    const int nseg = w.nseg[0];
    __shared__ float2 mm_sh;
    const float2 mm = pos_minmax(mm_part, n_mm, &mm_sh);
    const float smin = mm.x, smax = mm.y;
    if (blockIdx.x == 0 && threadIdx.x == 0) {      // for the second pass
        minmax[0] = smin;
        minmax[1] = smax;
    }
    __shared__ int cnt[DF_THREADS / DF_GROUP];
    int has = 0;
    if (k < nseg) {
        float s0 = 0.f, s1 = 0.f, n0 = 0.f, n1 = 0.f;
        bool bad = false;
        for (int j = w.seg_start[k] + sub; j < w.seg_start[k + 1]; j += DF_GROUP) {
            const int b = w.perm[j];
            if (label[b] == 1.f) {
                const float s = sst[b];
                bad |= (s != smin && s != smax);
                if (s == smin) { s0 += out[b]; n0 += 1.f; }
                else { s1 += out[b]; n1 += 1.f; }
            }
        }
        if (bad && err) atomicOr(err, FR_DEV_ERR_SST_GROUPS);
        s0 = group_sum<DF_GROUP>(s0); s1 = group_sum<DF_GROUP>(s1);
        n0 = group_sum<DF_GROUP>(n0); n1 = group_sum<DF_GROUP>(n1);
        has = (n0 + n1) > 0.f ? 1 : 0;
        if (sub == 0) stats[k] = make_float4(s0, s1, n0, n1);
    }
    if (sub == 0) cnt[gib] = has;
    __syncthreads();
    if (threadIdx.x == 0) {
        int c = 0;
#pragma unroll
        for (int q = 0; q < DF_THREADS / DF_GROUP; ++q) c += cnt[q];
        kpart[blockIdx.x] = c;
    }
}

// pass 2: K = number of items with a positive row; eps_k and the gradient wrt the positive rows' scores
__global__ __launch_bounds__(DF_THREADS) void nfcf_df_coef_kernel(TableWs w, const float* __restrict__ out,
                                                                  const float* __restrict__ label,
                                                                  const float* __restrict__ sst,
                                                                  const float* __restrict__ minmax,
                                                                  const float4* __restrict__ stats,
                                                                  const int* __restrict__ kpart, int n_kpart,
                                                                  float fair_weight, float* __restrict__ dy,
                                                                  float* __restrict__ part, float* __restrict__ kout) {
    const int sub = threadIdx.x & (DF_GROUP - 1), gib = threadIdx.x / DF_GROUP;
    const int k = blockIdx.x * (DF_THREADS / DF_GROUP) + gib;
    const int nseg = w.nseg[0];
    __shared__ float red[DF_THREADS / DF_GROUP];
    __shared__ int Ksh;
    if (threadIdx.x < 64) {   // every block re-derives K from the pass-1 partials in the same fixed order
        int c = 0;
        for (int q = threadIdx.x; q < n_kpart; q += 64) c += kpart[q];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        if (threadIdx.x == 0) Ksh = c;
    }
    __syncthreads();
    const int K = Ksh;
    if (blockIdx.x == 0 && threadIdx.x == 0) kout[0] = (float)K;
    const float smin = minmax[0], smax = minmax[1];
    float eps = 0.f;
    if (k < nseg && K > 0 && smin != smax) {
        const float4 st = stats[k];
        if (st.z + st.w > 0.f) {
            const float alpha = 1.f / (float)K;                  // dirichlet_alpha, nfcf.py:85-86
            const float M0 = (st.x + alpha) / (st.z + 1.f), M1 = (st.y + alpha) / (st.w + 1.f);
            const float d = __logf(M0) - __logf(M1);
            eps = fabsf(d);
            const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            const float g0 = fair_weight * sgn / (float)K / M0 / (st.z + 1.f);
            const float g1 = -fair_weight * sgn / (float)K / M1 / (st.w + 1.f);
            for (int j = w.seg_start[k] + sub; j < w.seg_start[k + 1]; j += DF_GROUP) {
                const int b = w.perm[j];
                if (label[b] == 1.f) {
                    const float o = out[b];
                    dy[b] += ((sst[b] == smin) ? g0 : g1) * o * (1.f - o);   // through the sigmoid
                }
            }
        }
    }
    if (sub == 0) red[gib] = eps;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < DF_THREADS / DF_GROUP; ++q) s += red[q];
        part[blockIdx.x] = s;
    }
}

// loss[0] = bce_sum / B + fair_weight * eps_sum / K ; loss[1] = bce ; loss[2] = DF term
__global__ __launch_bounds__(256) void nfcf_finalize_kernel(const float* __restrict__ bce_part, int n_bce,
                                                            const float* __restrict__ df_part, int n_df,
                                                            const float* __restrict__ kout, int B, float fair_weight,
                                                            float* __restrict__ loss) {
    __shared__ float red[2][4];
    float a = 0.f, f = 0.f;
    for (int q = threadIdx.x; q < n_bce; q += 256) a += bce_part[q];
    for (int q = threadIdx.x; q < n_df; q += 256) f += df_part[q];
    a = wave_sum(a);
    f = wave_sum(f);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = f; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
        f = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
        const float bce = a / (float)B;
        const float K = (n_df > 0) ? kout[0] : 0.f;
        const float df = (n_df > 0 && K > 0.f) ? f / K : 0.f;
        loss[0] = bce + (n_df > 0 ? fair_weight * df : 0.f);
        loss[1] = bce;
        loss[2] = df;
    }
}


// ---- differential fairness on the GLOBAL batch of a row-sharded step (one process per GPU) ------------------------------
// nfcf.py:76-97 computes M[k, g], K and the mean of eps over the whole batch.  With the item table row-sharded, every
// interaction's (score, label, group) travels to the owner of its item in the slot its item id took in the lookup's id
// exchange; the owner, whose sorted segments of the received ids ARE the per-item groups of the global batch (members in
// rank order, then batch position = the order of the concatenated batch), reduces the per-(item, group) sums and sends
// them back to every member; the requester then forms M, eps and dLoss/dscore for its own rows.  K = sum of the owners'
// counts of items with a positive row, and the groups present (min, max of the positive rows' attribute) ride in the
// tails of the two exchanges.  Buffers (one chunk of cap slots + 1 tail per peer rank):
//   rec   float2 [G, cap + 1]: (score, or -1 for a row with label != 1; attribute); tail = the sender's (min, max)
//   reply float4 [G, cap + 1]: (S0 | sign bit = "this member reports the item's eps", S1, n0, n1);
//                              tail = (K_owner, smin, smax, 0)
__device__ __forceinline__ long long df_phys(int j, int cap) { return (long long)(j / cap) * (cap + 1) + j % cap; }

// requester: records into the slots of the item lookup (slot[b] = o * S + off + k of the packed id exchange)
__global__ __launch_bounds__(256) void nfcf_df_pack_kernel(const float* __restrict__ out, const float* __restrict__ label,
                                                           const float* __restrict__ sst, const int32_t* __restrict__ slot,
                                                           int S, int off, int cap, int B, int G, float2* __restrict__ rec,
                                                           float* __restrict__ part, unsigned int* __restrict__ ticket) {
    __shared__ float lo_s[4], hi_s[4];
    __shared__ bool last;
    const int b = blockIdx.x * 256 + threadIdx.x;
    float lo = INFINITY, hi = -INFINITY;
    if (b < B) {
        const int sl = slot[b];
        const bool pos = label[b] == 1.f;
        const float s = sst[b];
        // sl < 0: fr_bucket_by_owner found the owner's bucket full (device error bit FR_DEV_ERR_BUCKET_OVERFLOW is set and the
        // step is void): no slot to write to -- the row is skipped like in fr_bucket_rows / fr_shard_pack_records
        if (sl >= 0) rec[(long long)(sl / S) * (cap + 1) + (sl % S - off)] = make_float2(pos ? out[b] : -1.f, s);
        if (pos) lo = hi = s;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o, 64));
        hi = fmaxf(hi, __shfl_xor(hi, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        lo_s[threadIdx.x >> 6] = lo;
        hi_s[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(part + 2 * blockIdx.x, fminf(fminf(lo_s[0], lo_s[1]), fminf(lo_s[2], lo_s[3])), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(part + 2 * blockIdx.x + 1, fmaxf(fmaxf(hi_s[0], hi_s[1]), fmaxf(hi_s[2], hi_s[3])),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = t == gridDim.x - 1;
        if (last) *ticket = 0u;
    }
    __syncthreads();
    if (!last) return;
    if (threadIdx.x < 64) {      // the last block to arrive: (min, max) over the blocks -> the tail of every chunk
        lo = INFINITY; hi = -INFINITY;
        for (int q = threadIdx.x; q < (int)gridDim.x; q += 64) {
            lo = fminf(lo, __hip_atomic_load(part + 2 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            hi = fmaxf(hi, __hip_atomic_load(part + 2 * q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo = fminf(lo, __shfl_xor(lo, o, 64));
            hi = fmaxf(hi, __shfl_xor(hi, o, 64));
        }
        for (int g = threadIdx.x; g < G; g += 64) rec[(long long)g * (cap + 1) + cap] = make_float2(lo, hi);
    }
}

// owner: per distinct received item (16 lanes) the sums over its positive members, written back to every positive member
__global__ __launch_bounds__(DF_THREADS) void nfcf_df_owner_kernel(TableWs w, const float2* __restrict__ rec, int G, int cap,
                                                                   float4* __restrict__ reply, int* __restrict__ kpart,
                                                                   unsigned int* __restrict__ ticket, uint32_t* err) {
    const int sub = threadIdx.x & (DF_GROUP - 1), gib = threadIdx.x / DF_GROUP;
    const int k = blockIdx.x * (DF_THREADS / DF_GROUP) + gib;
    const int nseg = w.nseg[0];
    __shared__ float2 mm_sh;
    __shared__ int cnt[DF_THREADS / DF_GROUP];
    __shared__ bool last;
    if (threadIdx.x < 64) {      // the groups present in the global batch: over the G senders' tails
        float lo = INFINITY, hi = -INFINITY;
        for (int g = threadIdx.x; g < G; g += 64) {
            const float2 t = rec[(long long)g * (cap + 1) + cap];
            lo = fminf(lo, t.x);
            hi = fmaxf(hi, t.y);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo = fminf(lo, __shfl_xor(lo, o, 64));
            hi = fmaxf(hi, __shfl_xor(hi, o, 64));
        }
        if (threadIdx.x == 0) mm_sh = make_float2(lo, hi);
    }
    __syncthreads();
    const float smin = mm_sh.x, smax = mm_sh.y;
    int has = 0;
    if (k < nseg) {
        const int j0 = w.seg_start[k], j1 = w.seg_start[k + 1];
        float s0 = 0.f, s1 = 0.f, n0 = 0.f, n1 = 0.f;
        int first = 0x7fffffff;       // sorted position of this lane's first positive member
        bool bad = false;
        for (int j = j0 + sub; j < j1; j += DF_GROUP) {
            const float2 r = rec[df_phys(w.perm[j], cap)];
            if (r.x >= 0.f) {
                bad |= (r.y != smin && r.y != smax);
                if (r.y == smin) { s0 += r.x; n0 += 1.f; }
                else { s1 += r.x; n1 += 1.f; }
                first = min(first, j);
            }
        }
        if (bad && err) atomicOr(err, FR_DEV_ERR_SST_GROUPS);
        s0 = group_sum<DF_GROUP>(s0); s1 = group_sum<DF_GROUP>(s1);
        n0 = group_sum<DF_GROUP>(n0); n1 = group_sum<DF_GROUP>(n1);
#pragma unroll
        for (int o = DF_GROUP / 2; o > 0; o >>= 1) first = min(first, __shfl_xor(first, o, 64));
        has = (n0 + n1) > 0.f ? 1 : 0;
        for (int j = j0 + sub; j < j1; j += DF_GROUP) {
            const long long ph = df_phys(w.perm[j], cap);
            if (rec[ph].x >= 0.f)
                reply[ph] = make_float4(j == first ? __uint_as_float(__float_as_uint(s0) | 0x80000000u) : s0, s1, n0, n1);
        }
    }
    if (sub == 0) cnt[gib] = has;
    __syncthreads();
    if (threadIdx.x == 0) {
        int c = 0;
#pragma unroll
        for (int q = 0; q < DF_THREADS / DF_GROUP; ++q) c += cnt[q];
        __hip_atomic_store(kpart + blockIdx.x, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = t == gridDim.x - 1;
        if (last) *ticket = 0u;
    }
    __syncthreads();
    if (!last || threadIdx.x >= 64) return;
    int c = 0;     // K of this owner: items with a positive row, summed over the blocks in a fixed order
    for (int q = threadIdx.x; q < (int)gridDim.x; q += 64) c += __hip_atomic_load(kpart + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    for (int g = threadIdx.x; g < G; g += 64) reply[(long long)g * (cap + 1) + cap] = make_float4((float)c, smin, smax, 0.f);
}

// requester: M, eps and the gradient of fair_weight * DF wrt this rank's positive rows; `scale` = G (the engine averages the
// ranks' gradients, the DF term is not a per-rank mean).  loss[0] += fair_weight * scale * (sum of the eps this rank
// reports) / K, so that the mean of the ranks' losses is the loss of the global batch; loss[2] = its DF share.
__global__ __launch_bounds__(256) void nfcf_df_apply_kernel(const float4* __restrict__ reply, const int32_t* __restrict__ slot,
                                                            int S, int off, int cap, int G, const float* __restrict__ out,
                                                            const float* __restrict__ label, const float* __restrict__ sst,
                                                            int B, float fair_weight, float scale, float* __restrict__ dy,
                                                            float* __restrict__ loss, float* __restrict__ part,
                                                            unsigned int* __restrict__ ticket) {
    __shared__ float red[4];
    __shared__ bool last;
    float Kf = 0.f;
    for (int g = 0; g < G; ++g) Kf += reply[(long long)g * (cap + 1) + cap].x;      // fixed order: the same bits on every rank
    const float4 tail = reply[cap];
    const float smin = tail.y, smax = tail.z;
    const int b = blockIdx.x * 256 + threadIdx.x;
    float eps_mine = 0.f;
    if (b < B && label[b] == 1.f && Kf > 0.f && smin != smax) {
        const int sl = slot[b];
        // (sl < 0: the row found no slot in its owner's bucket -- no reply to read; the overflow error bit voids the step)
        const float4 st = sl >= 0 ? reply[(long long)(sl / S) * (cap + 1) + (sl % S - off)] : make_float4(0.f, 0.f, 0.f, 0.f);
        const bool rep = (__float_as_uint(st.x) & 0x80000000u) != 0u;
        const float S0 = fabsf(st.x);
        const float alpha = 1.f / Kf;                  // dirichlet_alpha, nfcf.py:85-86
        const float M0 = (S0 + alpha) / (st.z + 1.f), M1 = (st.y + alpha) / (st.w + 1.f);
        const float d = __logf(M0) - __logf(M1);
        const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        const float g0 = fair_weight * sgn / Kf / M0 / (st.z + 1.f);
        const float g1 = -fair_weight * sgn / Kf / M1 / (st.w + 1.f);
        const float o = out[b];
        dy[b] += scale * ((sst[b] == smin) ? g0 : g1) * o * (1.f - o);   // through the sigmoid
        if (rep) eps_mine = fabsf(d);
    }
    eps_mine = wave_sum(eps_mine);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = eps_mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(part + blockIdx.x, ((red[0] + red[1]) + red[2]) + red[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = t == gridDim.x - 1;
        if (last) *ticket = 0u;
    }
    __syncthreads();
    if (!last || threadIdx.x >= 64) return;
    float e = 0.f;
    for (int q = threadIdx.x; q < (int)gridDim.x; q += 64) e += __hip_atomic_load(part + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    e = wave_sum(e);
    if (threadIdx.x == 0) {
        const float df = (Kf > 0.f && smin != smax) ? scale * e / Kf : 0.f;
        loss[0] += fair_weight * df;
        loss[2] = df;
    }
}

}  // namespace fr

using namespace fr;

extern "C" size_t fr_nfcf_loss_workspace_bytes(int64_t B) {
    if (B < 1) return 0;
    const size_t nb = (size_t)(B + 255) / 256, ndf = (size_t)(B * DF_GROUP + DF_THREADS - 1) / DF_THREADS;
    return align_up(nb * 4, 256) + align_up(ndf * 4, 256) * 2 + align_up((size_t)B * 16, 256) + 256 * 2 + align_up(nb * 8, 256);
}

// y [B] = scorer output after its ReLU; writes out = sigmoid(y) [B], dy [B] = dLoss/dy, loss[3].
// item_ws = the item table's training workspace (segments of the batch's item ids) when the differential-fairness
// term is on (stage finetune), else NULL.
extern "C" int fr_nfcf_loss(const float* y, const float* label, const float* sst, int64_t B, float fair_weight,
                            void* item_ws, size_t item_ws_bytes, int32_t dim, float* out, float* dy, float* loss, void* ws,
                            size_t ws_bytes, uint32_t* err_flag, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FR_CHECK_ARG(y && label && out && dy && loss && ws && B >= 1, "fr_nfcf_loss: bad argument");
    FR_CHECK_ARG(ws_bytes >= fr_nfcf_loss_workspace_bytes(B), "fr_nfcf_loss: workspace too small");
    const int nb = (int)((B + 255) / 256), ndf = (int)((B * DF_GROUP + DF_THREADS - 1) / DF_THREADS);
    char* p = (char*)ws;
    float* bce_part = (float*)p; p += align_up((size_t)nb * 4, 256);
    float* df_part = (float*)p; p += align_up((size_t)ndf * 4, 256);
    int* kpart = (int*)p; p += align_up((size_t)ndf * 4, 256);
    float4* stats = (float4*)p; p += align_up((size_t)B * 16, 256);
    float* minmax = (float*)p; p += 256;
    float* kout = (float*)p; p += 256;
    float* mm_part = (float*)p;
    const bool df = item_ws != nullptr;
    {
        ProfScope prof(K_NFCF_LOSS, stream);
        FR_LAUNCH(prof, nfcf_bce_kernel, dim3(nb), dim3(256), 0, stream, y, label, (int)B, out, dy, bce_part,
                  df ? sst : (const float*)nullptr, mm_part);
    }
    FR_CHECK_LAUNCH();
    if (df) {
        FR_CHECK_ARG(sst, "fr_nfcf_loss: the fairness term needs the sst column");
        TableWs tw = table_layout(item_ws, B, dim);
        FR_CHECK_ARG(item_ws_bytes >= tw.bytes, "fr_nfcf_loss: item workspace too small");
        // the sort that fills the segments runs on the side stream behind fr_table_gather_train: wait for it here
        if (int rc = side_join(item_ws, stream)) return rc;
        hipLaunchKernelGGL(nfcf_df_stats_kernel, dim3(ndf), dim3(DF_THREADS), 0, stream, tw, (const float*)out, label, sst,
                           (const float*)mm_part, nb, minmax, stats, kpart, err_flag);
        FR_CHECK_LAUNCH();
        hipLaunchKernelGGL(nfcf_df_coef_kernel, dim3(ndf), dim3(DF_THREADS), 0, stream, tw, (const float*)out, label, sst,
                           (const float*)minmax, (const float4*)stats, (const int*)kpart, ndf, fair_weight, dy, df_part,
                           kout);
        FR_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(nfcf_finalize_kernel, dim3(1), dim3(256), 0, stream, (const float*)bce_part, nb,
                       (const float*)df_part, df ? ndf : 0, (const float*)kout, (int)B, fair_weight, loss);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_nfcf_loss_tail(const float* label, const float* sst, int64_t B, float fair_weight, void* item_ws,
                                 size_t item_ws_bytes, int32_t dim, const float* out, float* dy, float* loss,
                                 const float* bce_part, const float* mm_part, int32_t n_part, void* ws, size_t ws_bytes,
                                 uint32_t* err_flag, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FR_CHECK_ARG(label && out && dy && loss && bce_part && ws && B >= 1 && n_part >= 1, "fr_nfcf_loss_tail: bad argument");
    FR_CHECK_ARG(ws_bytes >= fr_nfcf_loss_workspace_bytes(B), "fr_nfcf_loss_tail: workspace too small");
    const int nb = (int)((B + 255) / 256), ndf = (int)((B * DF_GROUP + DF_THREADS - 1) / DF_THREADS);
    char* p = (char*)ws;
    p += align_up((size_t)nb * 4, 256);
    float* df_part = (float*)p; p += align_up((size_t)ndf * 4, 256);
    int* kpart = (int*)p; p += align_up((size_t)ndf * 4, 256);
    float4* stats = (float4*)p; p += align_up((size_t)B * 16, 256);
    float* minmax = (float*)p; p += 256;
    float* kout = (float*)p;
    const bool df = item_ws != nullptr;
    if (df) {
        FR_CHECK_ARG(sst && mm_part, "fr_nfcf_loss_tail: the fairness term needs the sst column and the (min, max) partials");
        TableWs tw = table_layout(item_ws, B, dim);
        FR_CHECK_ARG(item_ws_bytes >= tw.bytes, "fr_nfcf_loss_tail: item workspace too small");
        if (int rc = side_join(item_ws, stream)) return rc;
        ProfScope prof(K_NFCF_LOSS, stream);
        FR_LAUNCH(prof, nfcf_df_stats_kernel, dim3(ndf), dim3(DF_THREADS), 0, stream, tw, out, label, sst, mm_part, (int)n_part,
                  minmax, stats, kpart, err_flag);
        FR_CHECK_LAUNCH();
        hipLaunchKernelGGL(nfcf_df_coef_kernel, dim3(ndf), dim3(DF_THREADS), 0, stream, tw, out, label, sst, (const float*)minmax,
                           (const float4*)stats, (const int*)kpart, ndf, fair_weight, dy, df_part, kout);
        FR_CHECK_LAUNCH();
    }
    // (closing the loss inside pass 2 through an arrival ticket was measured: 11.7 us against 5.4 + 4.8 for the two launches)
    hipLaunchKernelGGL(nfcf_finalize_kernel, dim3(1), dim3(256), 0, stream, bce_part, (int)n_part, (const float*)df_part,
                       df ? ndf : 0, (const float*)kout, (int)B, fair_weight, loss);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

/* ---- differential fairness of a row-sharded NFCF step on the GLOBAL batch (three launches around two all-to-alls) ---- */
extern "C" size_t fr_nfcf_df_workspace_bytes(int64_t B, int64_t n_slots) {
    if (B < 1 || n_slots < 1) return 0;
    const size_t nb = (size_t)(B + 255) / 256;
    const size_t ndf = (size_t)(n_slots * DF_GROUP + DF_THREADS - 1) / DF_THREADS;
    return align_up(nb * 8, 256) + align_up(ndf * 4, 256) + align_up(nb * 4, 256) + 256;
}

struct DfWs {
    float* mm_part;
    int* kpart;
    float* eps_part;
    unsigned int* ticket;     // [3], kept zero between launches
};
static DfWs df_layout(void* ws, int64_t B, int64_t n_slots) {
    const size_t nb = (size_t)(B + 255) / 256;
    const size_t ndf = (size_t)(n_slots * DF_GROUP + DF_THREADS - 1) / DF_THREADS;
    char* p = (char*)ws;
    DfWs w;
    w.mm_part = (float*)p; p += align_up(nb * 8, 256);
    w.kpart = (int*)p; p += align_up(ndf * 4, 256);
    w.eps_part = (float*)p; p += align_up(nb * 4, 256);
    w.ticket = (unsigned int*)p;
    return w;
}

extern "C" int fr_nfcf_df_pack(const float* out, const float* label, const float* sst, const int32_t* slot, int32_t slot_stride,
                               int32_t slot_off, int32_t cap, int64_t B, int32_t G, float* rec, void* ws, size_t ws_bytes,
                               void* stream_) {
    FR_CHECK_ARG(out && label && sst && slot && rec && ws && B >= 1 && G >= 1 && cap >= 1 && slot_stride >= cap,
                 "fr_nfcf_df_pack: bad argument");
    FR_CHECK_ARG(ws_bytes >= fr_nfcf_df_workspace_bytes(B, (int64_t)G * cap), "fr_nfcf_df_pack: workspace too small");
    const DfWs w = df_layout(ws, B, (int64_t)G * cap);
    hipLaunchKernelGGL(nfcf_df_pack_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, out, label,
                       sst, slot, slot_stride, slot_off, cap, (int)B, G, (float2*)rec, w.mm_part, w.ticket);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_nfcf_df_owner(void* item_ws, size_t item_ws_bytes, int32_t dim, const float* rec, int32_t G, int32_t cap,
                                float* reply, void* ws, size_t ws_bytes, int64_t B, uint32_t* err_flag, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t M = (int64_t)G * cap;
    FR_CHECK_ARG(item_ws && rec && reply && ws && G >= 1 && cap >= 1 && B >= 1, "fr_nfcf_df_owner: bad argument");
    FR_CHECK_ARG(ws_bytes >= fr_nfcf_df_workspace_bytes(B, M), "fr_nfcf_df_owner: workspace too small");
    TableWs tw = table_layout(item_ws, M, dim);
    FR_CHECK_ARG(item_ws_bytes >= tw.bytes, "fr_nfcf_df_owner: item workspace too small");
    if (int rc = side_join(item_ws, stream)) return rc;     // the owner's sort of the received ids runs on the side stream
    const DfWs w = df_layout(ws, B, M);
    const unsigned ndf = (unsigned)((M * DF_GROUP + DF_THREADS - 1) / DF_THREADS);
    hipLaunchKernelGGL(nfcf_df_owner_kernel, dim3(ndf), dim3(DF_THREADS), 0, stream, tw, (const float2*)rec, G, cap,
                       (float4*)reply, w.kpart, w.ticket + 1, err_flag);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_nfcf_df_apply(const float* reply, const int32_t* slot, int32_t slot_stride, int32_t slot_off, int32_t cap,
                                int32_t G, const float* out, const float* label, const float* sst, int64_t B, float fair_weight,
                                float scale, float* dy, float* loss, void* ws, size_t ws_bytes, void* stream_) {
    FR_CHECK_ARG(reply && slot && out && label && sst && dy && loss && ws && B >= 1 && G >= 1 && cap >= 1,
                 "fr_nfcf_df_apply: bad argument");
    FR_CHECK_ARG(ws_bytes >= fr_nfcf_df_workspace_bytes(B, (int64_t)G * cap), "fr_nfcf_df_apply: workspace too small");
    const DfWs w = df_layout(ws, B, (int64_t)G * cap);
    hipLaunchKernelGGL(nfcf_df_apply_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream_,
                       (const float4*)reply, slot, slot_stride, slot_off, cap, G, out, label, sst, (int)B, fair_weight, scale,
                       dy, loss, w.eps_part, w.ticket + 2);
    FR_CHECK_LAUNCH();
    return FR_OK;
}
