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
