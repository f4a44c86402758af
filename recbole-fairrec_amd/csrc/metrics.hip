// Evaluation metrics on the device: the reductions of recbole/evaluator/metrics.py without the host
// round trip and without the reference's Python loops over interactions (:950-960) or over items x groups (:1331-1335).
//
//   topk_metrics_kernel      Hit / MRR / NDCG / Recall / Precision / MAP @ 1..K from the `rec.topk` matrix (:40-232)
//   group_sums_kernel        per (item segment, group): sum of scores, count, count of positives -- the tables every
//                            fairness metric starts from (:948-970, :1322-1335); same segment machinery as the FOCF
//                            fairness term (16 lanes per segment, fixed order)
//   fair_from_stats_kernel   Value / Absolute / Under / Over unfairness (:972-979 and siblings) and DifferentialFairness
//                            (:1337-1342, float32 tables like the reference) as means over the items
// All sums are double precision in a fixed order (bit-reproducible).
#include "common.hpp"
#include "kernels.hpp"

namespace fr {

__device__ __forceinline__ double wave_sum_d(double x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

// one wave per 64 users (lane = user); part[block][m][k], m = hit, mrr, ndcg, recall, precision, map
__global__ __launch_bounds__(64) void topk_metrics_kernel(const int32_t* __restrict__ rec_topk, long long U, int K,
                                                          double* __restrict__ part) {
    const int lane = threadIdx.x;
    const long long u = (long long)blockIdx.x * 64 + lane;
    const bool ok = u < U;
    const int32_t* row = rec_topk + (ok ? u : 0) * (long long)(K + 1);
    const int pos_len = ok ? row[K] : 1;
    const int ilen = pos_len < K ? pos_len : K;
    int csum = 0;
    double first_rr = 0.0, dcg = 0.0, idcg = 0.0, sum_pre = 0.0;
    double* out = part + (size_t)blockIdx.x * 6 * K;
    for (int k = 0; k < K; ++k) {
        const bool hit = ok && row[k] != 0;
        const double disc = 1.0 / log2((double)(k + 2));
        if (hit && csum == 0) first_rr = 1.0 / (double)(k + 1);
        csum += hit ? 1 : 0;
        if (hit) dcg += disc;
        if (k < ilen) idcg += disc;
        const double v0 = ok && csum > 0 ? 1.0 : 0.0;
        const double v1 = ok ? first_rr : 0.0;
        const double v2 = ok ? dcg / idcg : 0.0;
        const double v3 = ok ? (double)csum / (double)pos_len : 0.0;
        const double v4 = ok ? (double)csum / (double)(k + 1) : 0.0;
        // MAP (metrics.py:100-139): sum over the hits so far of Precision@j, over min(k + 1, min(positives, K))
        if (hit) sum_pre += (double)csum / (double)(k + 1);
        const double v5 = ok && ilen > 0 ? sum_pre / (double)(k + 1 < ilen ? k + 1 : ilen) : 0.0;
        const double s0 = wave_sum_d(v0), s1 = wave_sum_d(v1), s2 = wave_sum_d(v2), s3 = wave_sum_d(v3),
                     s4 = wave_sum_d(v4), s5 = wave_sum_d(v5);
        if (lane == 0) {
            out[0 * K + k] = s0;
            out[1 * K + k] = s1;
            out[2 * K + k] = s2;
            out[3 * K + k] = s3;
            out[4 * K + k] = s4;
            out[5 * K + k] = s5;
        }
    }
}

// rec_topk[u][k] = 1 if the k-th ranked item of row u is one of its positives, rec_topk[u][K] = number of positives of
// row u (collector.py:146-154 without the dense [users, items] 0/1 matrix): the positives arrive as the SORTED keys
// row * n_items + item; membership and the per-row count are binary searches.
__device__ __forceinline__ long long lower_bound_ll(const int64_t* __restrict__ a, long long n, long long key) {
    long long lo = 0, hi = n;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void eval_hits_kernel(const int64_t* __restrict__ topk_idx, long long U, int K,
                                                        long long n_items, const int64_t* __restrict__ pos_keys,
                                                        long long n_pos, int32_t* __restrict__ rec_topk) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= U * (K + 1)) return;
    const long long u = t / (K + 1);
    const int k = (int)(t % (K + 1));
    if (k == K) {
        rec_topk[t] = (int32_t)(lower_bound_ll(pos_keys, n_pos, (u + 1) * n_items) - lower_bound_ll(pos_keys, n_pos, u * n_items));
    } else {
        const long long key = u * n_items + topk_idx[u * K + k];
        const long long p = lower_bound_ll(pos_keys, n_pos, key);
        rec_topk[t] = (p < n_pos && pos_keys[p] == key) ? 1 : 0;
    }
}

// out[j] = (sum over blocks, in block order, of part[b][j]) * scale          j < n
__global__ __launch_bounds__(256) void column_sum_kernel(const double* __restrict__ part, long long blocks, int n,
                                                         double scale, double* __restrict__ out) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    double a = 0.0;
    for (long long b = 0; b < blocks; ++b) a += part[b * n + j];
    out[j] = a * scale;
}

static constexpr int GS_GROUP = 16, GS_THREADS = 256, GS_MAXG = 8;

// stats[k][g][0..2] = sum value, count, sum wtrue over the members perm[seg_start[k] .. seg_start[k+1]) with group g
__global__ __launch_bounds__(GS_THREADS) void group_sums_kernel(const int64_t* __restrict__ perm,
                                                                const int64_t* __restrict__ seg_start, long long K,
                                                                const int32_t* __restrict__ group,
                                                                const float* __restrict__ value,
                                                                const float* __restrict__ wtrue, int G,
                                                                double* __restrict__ stats) {
    const int sub = threadIdx.x & (GS_GROUP - 1);
    const long long k = (long long)blockIdx.x * (GS_THREADS / GS_GROUP) + threadIdx.x / GS_GROUP;
    if (k >= K) return;   // whole 16-lane groups leave together
    double sv[GS_MAXG], sc[GS_MAXG], st[GS_MAXG];
#pragma unroll
    for (int g = 0; g < GS_MAXG; ++g) sv[g] = sc[g] = st[g] = 0.0;
    const long long j0 = seg_start[k], j1 = seg_start[k + 1];
    for (long long j = j0 + sub; j < j1; j += GS_GROUP) {
        const long long b = perm ? perm[j] : j;
        const int g = group[b];
        const double v = (double)value[b], t = wtrue ? (double)wtrue[b] : 0.0;
#pragma unroll
        for (int q = 0; q < GS_MAXG; ++q)
            if (q == g) {
                sv[q] += v;
                sc[q] += 1.0;
                st[q] += t;
            }
    }
#pragma unroll
    for (int q = 0; q < GS_MAXG; ++q) {
        if (q >= G) break;
#pragma unroll
        for (int o = GS_GROUP / 2; o > 0; o >>= 1) {
            sv[q] += __shfl_xor(sv[q], o, 64);
            sc[q] += __shfl_xor(sc[q], o, 64);
            st[q] += __shfl_xor(st[q], o, 64);
        }
        if (sub == 0) {
            double* o3 = stats + ((size_t)k * G + q) * 3;
            o3[0] = sv[q];
            o3[1] = sc[q];
            o3[2] = st[q];
        }
    }
}

// part[block][0..4] = sums over the block's items of: value, absolute, under, over terms (G == 2) and the DF epsilon
__global__ __launch_bounds__(256) void fair_from_stats_kernel(const double* __restrict__ stats, long long K, int G,
                                                              double* __restrict__ part) {
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    double t[5] = {0, 0, 0, 0, 0};
    if (k < K) {
        const double* s = stats + (size_t)k * G * 3;
        if (G == 2) {
            const double n0 = s[1] + 1e-5, n1 = s[4] + 1e-5;                  // sst_num += 1e-5, metrics.py:962
            const double p0 = s[0] / n0, p1 = s[3] / n1, r0 = s[2] / n0, r1 = s[5] / n1;
            t[0] = fabs((p0 - r0) - (p1 - r1));
            t[1] = fabs(fabs(p0 - r0) - fabs(p1 - r1));
            t[2] = fabs((r0 - p0 > 0 ? r0 - p0 : 0.0) - (r1 - p1 > 0 ? r1 - p1 : 0.0));
            t[3] = fabs((p0 - r0 > 0 ? p0 - r0 : 0.0) - (p1 - r1 > 0 ? p1 - r1 : 0.0));
        }
        // DifferentialFairness: float32 table (score_sum + 1/K) / (count + 1), epsilon = max pairwise |log a - log b|
        float eps = 0.f;
        const double alpha = 1.0 / (double)K;
        for (int i = 0; i < G; ++i) {
            const float a = (float)((s[i * 3] + alpha) / (s[i * 3 + 1] + 1.0));
            for (int j = i + 1; j < G; ++j) {
                const float b = (float)((s[j * 3] + alpha) / (s[j * 3 + 1] + 1.0));
                const float e = fabsf(logf(a) - logf(b));
                eps = e > eps ? e : eps;
            }
        }
        t[4] = (double)eps;
    }
    __shared__ double red[4][5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const double w = wave_sum_d(t[q]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][q] = w;
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        const int q = threadIdx.x;
        part[(size_t)blockIdx.x * 5 + q] = ((red[0][q] + red[1][q]) + red[2][q]) + red[3][q];
    }
}

// ---- ranking of the candidates of an evaluation batch (uniN protocol): one wave per user -------------------------------------
// Reference: trainer.py:441-456 scatters every user's candidates (its positives and the negatives sampled for them) into a dense
// [users, n_items] matrix of -inf, collector.py:149 takes torch.topk(matrix, max(topk)) of it.  The evaluation loader emits a
// batch user by user, so a user's candidates are a contiguous SEGMENT of the batch (seg_start), and its list is the K best
// DISTINCT items of that segment (an item drawn twice scores the same twice and occupies one cell of the dense row).  A wave keeps
// the K + 1 best so far in lanes 0..K (sorted, descending) and walks the segment 64 candidates at a time, each time pulling the
// chunk's best through a wave maximum until it no longer beats the (K + 1)-th.  flags[u]: bit 0 = two of the first K + 1 entries
// score EQUAL or a candidate's score is NaN (which of them the reference ranks first is torch.topk's CPU order: the caller
// ranks that row on the host, fr_topk_like_torch_cpu), bit 1 = fewer than K + 1 distinct candidates (the reference's list then continues into -inf cells).
// Replaces three device sorts over the batch's candidates per evaluation batch (1.3 ms -> 0.1 ms at 280 k candidates).
constexpr int TOPK_MAXK = 62;
__global__ __launch_bounds__(256) void eval_topk_segments_kernel(const int64_t* __restrict__ seg_start, long long n_users,
                                                                 const int64_t* __restrict__ items, const float* __restrict__ scores,
                                                                 int k, int64_t* __restrict__ topk_idx, int32_t* __restrict__ flags) {
    const int lane = threadIdx.x & 63;
    const long long u = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (u >= n_users) return;
    const long long s = seg_start[u], e = seg_start[u + 1];
    float t_score = -INFINITY;      // lane j < n_in: the j-th best so far
    long long t_item = 0;
    int n_in = 0;
    bool any_nan = false;
    for (long long base = s; base < e; base += 64) {
        const bool have = base + lane < e;
        float c_score = have ? scores[base + lane] : -INFINITY;
        const long long c_item = have ? items[base + lane] : 0;
        // (torch.topk ranks a NaN above every number; a user with one is handed to the host's ranking like a tied one)
        any_nan |= __ballot(have && c_score != c_score) != 0ull;
        unsigned long long valid = __ballot(have && c_score == c_score);
        while (valid) {
            float m = (valid >> lane) & 1ull ? c_score : -INFINITY;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
            const float kth = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t_score), k));
            if (n_in == k + 1 && !(m > kth)) break;       // nothing left in this chunk enters the first K + 1
            const unsigned long long at = __ballot(((valid >> lane) & 1ull) && c_score == m);
            const int src = __builtin_ctzll(at);
            valid &= ~(1ull << src);
            const long long it = (long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)c_item, src) |
                                 ((long long)__builtin_amdgcn_readlane((int)(c_item >> 32), src) << 32);
            if (__ballot(lane < n_in && t_item == it)) continue;       // the same item again: one cell of the dense row
            // behind every entry that scores >= m (so equal scores keep their order of arrival; which order the reference has
            // is settled on the host when the flag below says it matters)
            const int p = __popcll(__ballot(lane < n_in && t_score >= m));
            const float up_s = __shfl_up(t_score, 1, 64);
            const long long up_i = (long long)(unsigned)__shfl_up((int)(unsigned)t_item, 1, 64) |
                                   ((long long)__shfl_up((int)(t_item >> 32), 1, 64) << 32);
            if (lane > p) { t_score = up_s; t_item = up_i; }
            if (lane == p) { t_score = m; t_item = it; }
            if (n_in < k + 1) ++n_in;
            if (lane >= n_in) t_score = -INFINITY;
        }
    }
    const float prev = __shfl_up(t_score, 1, 64);
    const bool tie = __ballot(lane >= 1 && lane < n_in && lane <= k && t_score == prev) != 0ull;
    if (lane < k) topk_idx[u * k + lane] = lane < n_in ? t_item : 0;       // (short lists are padded with [PAD] item 0)
    if (lane == 0) flags[u] = (tie || any_nan ? 1 : 0) | (n_in < k + 1 ? 2 : 0);
}

// out[q] = the score of candidate item q_items[q] in the segment of user row q_rows[q], -inf if it is not one (a lookup in the
// reference's dense -inf matrix: collector.py:160-175 reads positives' and negatives' scores back out of it)
// (16 lanes per query walk the segment side by side -- every load of the walk in flight at once; one lane per query with a
// break on the match was ~100 dependent loads: 38 us per launch against 13 for the ranking itself.  A candidate drawn twice
// carries the same score twice, so which copy answers does not matter.)
constexpr int LOOKUP_LANES = 16;
__global__ __launch_bounds__(256) void eval_lookup_segments_kernel(const int64_t* __restrict__ seg_start, long long n_users,
                                                                   const int64_t* __restrict__ items, const float* __restrict__ scores,
                                                                   const int64_t* __restrict__ q_rows, const int64_t* __restrict__ q_items,
                                                                   long long n_q, float* __restrict__ out) {
    const long long q = ((long long)blockIdx.x * 256 + threadIdx.x) / LOOKUP_LANES;
    const int sub = threadIdx.x & (LOOKUP_LANES - 1);
    const bool live = q < n_q;
    const long long r = live ? q_rows[q] : -1;
    float v = -INFINITY;
    bool found = false;
    if (r >= 0 && r < n_users) {
        const long long it = q_items[q];
        const long long j1 = seg_start[r + 1];
#pragma unroll 4
        for (long long j = seg_start[r] + sub; j < j1; j += LOOKUP_LANES)
            if (items[j] == it) {
                v = scores[j];
                found = true;
            }
    }
    // the value of the first lane of the group that found it (not a maximum: a score may be NaN and has to come back as NaN)
    const int lane = threadIdx.x & 63, base = lane & ~(LOOKUP_LANES - 1);
    const unsigned bits = (unsigned)(__ballot(found) >> base) & ((1u << LOOKUP_LANES) - 1u);
    const float got = __shfl(v, base + (bits ? __ffs((int)bits) - 1 : 0), 64);
    if (live && sub == 0) out[q] = bits ? got : -INFINITY;
}

}  // namespace fr

using namespace fr;

extern "C" size_t fr_topk_metrics_workspace_bytes(int64_t n_users, int32_t k) {
    return n_users < 1 || k < 1 ? 0 : (size_t)((n_users + 63) / 64) * 6 * k * sizeof(double);
}

extern "C" int fr_topk_metrics(const int32_t* rec_topk, int64_t n_users, int32_t k, double* out, void* ws, size_t ws_bytes,
                               void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FR_CHECK_ARG(rec_topk && out && ws && n_users >= 1 && k >= 1 && ws_bytes >= fr_topk_metrics_workspace_bytes(n_users, k),
                 "fr_topk_metrics: bad argument");
    const long long blocks = (n_users + 63) / 64;
    hipLaunchKernelGGL(topk_metrics_kernel, dim3((unsigned)blocks), dim3(64), 0, stream, rec_topk, (long long)n_users, (int)k,
                       (double*)ws);
    FR_CHECK_LAUNCH();
    hipLaunchKernelGGL(column_sum_kernel, dim3((unsigned)((6 * k + 255) / 256)), dim3(256), 0, stream, (const double*)ws,
                       blocks, 6 * k, 1.0 / (double)n_users, out);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_eval_topk_segments(const int64_t* seg_start, int64_t n_users, const int64_t* items, const float* scores,
                                     int32_t k, int64_t* topk_idx, int32_t* flags, void* stream_) {
    FR_CHECK_ARG(seg_start && items && scores && topk_idx && flags && n_users >= 1 && k >= 1 && k <= TOPK_MAXK,
                 "fr_eval_topk_segments: bad argument (1 <= k <= %d)", TOPK_MAXK);
    hipLaunchKernelGGL(eval_topk_segments_kernel, dim3((unsigned)((n_users + 3) / 4)), dim3(256), 0, (hipStream_t)stream_, seg_start,
                       (long long)n_users, items, scores, (int)k, topk_idx, flags);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_eval_lookup_segments(const int64_t* seg_start, int64_t n_users, const int64_t* items, const float* scores,
                                       const int64_t* q_rows, const int64_t* q_items, int64_t n_q, float* out, void* stream_) {
    FR_CHECK_ARG(seg_start && items && scores && (n_q == 0 || (q_rows && q_items && out)) && n_users >= 1 && n_q >= 0,
                 "fr_eval_lookup_segments: bad argument");
    if (n_q == 0) return FR_OK;
    hipLaunchKernelGGL(eval_lookup_segments_kernel, dim3((unsigned)((n_q * LOOKUP_LANES + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, seg_start,
                       (long long)n_users, items, scores, q_rows, q_items, (long long)n_q, out);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_eval_hits(const int64_t* topk_idx, int64_t n_rows, int32_t k, int64_t n_items, const int64_t* pos_keys,
                            int64_t n_pos, int32_t* rec_topk, void* stream_) {
    FR_CHECK_ARG(topk_idx && rec_topk && (pos_keys || n_pos == 0) && n_rows >= 1 && k >= 1 && n_items >= 1 && n_pos >= 0,
                 "fr_eval_hits: bad argument");
    const long long n = (long long)n_rows * (k + 1);
    hipLaunchKernelGGL(eval_hits_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, topk_idx,
                       (long long)n_rows, (int)k, (long long)n_items, pos_keys, (long long)n_pos, rec_topk);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_group_sums(const int64_t* perm, const int64_t* seg_start, int64_t n_segments, const int32_t* group,
                             const float* value, const float* wtrue, int32_t n_groups, double* stats, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FR_CHECK_ARG(seg_start && group && value && stats && n_segments >= 1 && n_groups >= 1 && n_groups <= GS_MAXG,
                 "fr_group_sums: bad argument (1 <= n_groups <= %d)", GS_MAXG);
    const long long per_block = GS_THREADS / GS_GROUP;
    hipLaunchKernelGGL(group_sums_kernel, dim3((unsigned)((n_segments + per_block - 1) / per_block)), dim3(GS_THREADS), 0,
                       stream, perm, seg_start, (long long)n_segments, group, value, wtrue, (int)n_groups, stats);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" size_t fr_fair_metrics_workspace_bytes(int64_t n_segments) {
    return n_segments < 1 ? 0 : (size_t)((n_segments + 255) / 256) * 5 * sizeof(double);
}

extern "C" int fr_fair_metrics_from_stats(const double* stats, int64_t n_segments, int32_t n_groups, double* out, void* ws,
                                          size_t ws_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FR_CHECK_ARG(stats && out && ws && n_segments >= 1 && n_groups >= 1 && n_groups <= GS_MAXG &&
                     ws_bytes >= fr_fair_metrics_workspace_bytes(n_segments), "fr_fair_metrics_from_stats: bad argument");
    const long long blocks = (n_segments + 255) / 256;
    hipLaunchKernelGGL(fair_from_stats_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, stats, (long long)n_segments,
                       (int)n_groups, (double*)ws);
    FR_CHECK_LAUNCH();
    hipLaunchKernelGGL(column_sum_kernel, dim3(1), dim3(256), 0, stream, (const double*)ws, blocks, 5,
                       1.0 / (double)n_segments, out);
    FR_CHECK_LAUNCH();
    return FR_OK;
}
