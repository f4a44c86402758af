// PFCN scoring / loss kernels: row-wise dot products of gathered rows, BPR loss (plain and the reference's
// [B] + [B,1] -> [B,B] broadcast form), softmax cross-entropy for multi-class discriminators.
//
// Replaces pfcn_pmf.py:182-186 / pfcn_biasedmf.py:192-195 (torch.mul(...).sum(-1), BPRLoss loss.py:45-47) and
// pfcn_biasedmf.py:211-216 (nn.CrossEntropyLoss) with their autograd.
#include "common.hpp"
#include "kernels.hpp"

namespace fr {

// out[b] = sum_d a[b,d] * b[b,d]                         (one wave per row)
__global__ __launch_bounds__(256) void rowdot_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, int B,
                                                         int D, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= B) return;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s = fmaf(a[(size_t)r * D + d], b[(size_t)r * D + d], s);
    s = wave_sum(s);
    if (lane == 0) out[r] = s;
}

// da[b,:] = g[b] * b[b,:],  db[b,:] = g[b] * a[b,:]
__global__ __launch_bounds__(256) void rowdot_bwd_kernel(const float* __restrict__ g, const float* __restrict__ a,
                                                         const float* __restrict__ b, int B, int D,
                                                         float* __restrict__ da, float* __restrict__ db) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= B) return;
    const float gr = g[r];
    for (int d = lane; d < D; d += 64) {
        const size_t i = (size_t)r * D + d;
        if (da) da[i] = gr * b[i];
        if (db) db[i] = gr * a[i];
    }
}

// The same with the rows of `a` [A, D] reused by R row blocks of `b` [R*A, D] (one user row against its positive AND its
// negative item row, both gathered in one [2B, D] lookup): out[r*A + i] = a[i] . b[r*A + i]; one wave per row of a.
__global__ __launch_bounds__(256) void rowdot_rep_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, int A,
                                                             int R, int D, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= A) return;
    for (int r = 0; r < R; ++r) {
        const size_t row = (size_t)r * A + i;
        float s = 0.f;
        for (int d = lane; d < D; d += 64) s = fmaf(a[(size_t)i * D + d], b[row * D + d], s);
        s = wave_sum(s);
        if (lane == 0) out[row] = s;
    }
}

// da[i,:] = sum_r g[r*A + i] * b[r*A + i,:] (in r order),  db[r*A + i,:] = g[r*A + i] * a[i,:]
__global__ __launch_bounds__(256) void rowdot_rep_bwd_kernel(const float* __restrict__ g, const float* __restrict__ a,
                                                             const float* __restrict__ b, int A, int R, int D,
                                                             float* __restrict__ da, float* __restrict__ db) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= A) return;
    for (int d = lane; d < D; d += 64) {
        const float av = a[(size_t)i * D + d];
        float s = 0.f;
        for (int r = 0; r < R; ++r) {
            const size_t row = (size_t)r * A + i;
            const float gr = g[row];
            if (da) s = fmaf(gr, b[row * D + d], s);
            if (db) db[row * D + d] = gr * av;
        }
        if (da) da[(size_t)i * D + d] = s;
    }
}

// ... with the gradient of `a` left UNSUMMED: da_sep[r*A + i,:] = g[r*A + i] * b[r*A + i,:] -- what R separate rowdot_bwd
// launches write (the same single product per element), for a caller whose autograd graph adds them up itself
__global__ __launch_bounds__(256) void rowdot_rep_bwd_sep_kernel(const float* __restrict__ g, const float* __restrict__ a,
                                                                 const float* __restrict__ b, int A, int R, int D,
                                                                 float* __restrict__ da_sep, float* __restrict__ db) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= A) return;
    for (int d = lane; d < D; d += 64) {
        const float av = a[(size_t)i * D + d];
        for (int r = 0; r < R; ++r) {
            const size_t row = (size_t)r * A + i;
            const float gr = g[row];
            if (da_sep) da_sep[row * D + d] = gr * b[row * D + d];
            if (db) db[row * D + d] = gr * av;
        }
    }
}

__device__ __forceinline__ float bpr_term(float x, float& dterm) {
    // -log(1e-10 + sigmoid(x)) and its derivative  -sigmoid'(x) / (1e-10 + sigmoid(x))
    const float s = 1.f / (1.f + __expf(-x));
    dterm = -s * (1.f - s) / (1e-10f + s);
    return -__logf(1e-10f + s);
}

// plain BPR (PMF): loss = mean_b term(pos_b - neg_b); dpos = dterm / B, dneg = -dterm / B
__global__ __launch_bounds__(256) void bpr_kernel(const float* __restrict__ pos, const float* __restrict__ neg, int B,
                                                  float* __restrict__ dpos, float* __restrict__ dneg,
                                                  float* __restrict__ part) {
    __shared__ float red[4];
    const int b = blockIdx.x * 256 + threadIdx.x;
    float l = 0.f;
    if (b < B) {
        float dt;
        l = bpr_term(pos[b] - neg[b], dt);
        dpos[b] = dt / (float)B;
        dneg[b] = -dt / (float)B;
    }
    l = wave_sum(l);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = l;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

// BiasedMF's broadcast form: x_ij = a_j + c_i, loss = mean_ij term(x_ij).
// One block per OUTER_ROWS values of i; thread t walks the columns j = t, t + 256, ... and evaluates every (i, j) pair
// ONCE: the pair's derivative goes into the thread's column sum and into its private row-sum register, the row sums are
// reduced over the block at the end (butterfly + wave order: fixed order).  Per-block column partials ga_part[block][j]
// are added in block order by a second kernel.  B^2 transcendental groups, nothing of size B^2 is ever stored (the
// reference materialises the [B,B] matrix).
static constexpr int OUTER_ROWS = 16;      // (32 until round 4: 256 workgroups at B = 8192 left every SIMD with ONE wave of this
                                           // transcendental-bound loop; 512 give the quarter-rate unit a second wave to issue from)

// -log(1e-10 + sigmoid(x)) and its derivative with hardware exp / log / rcp (1-2 ulp): with e = exp(-x), r = 1/(1+e):
// sigmoid = r, 1 - sigmoid = e*r, so  term = -log(1e-10 + r),  dterm = -(r*r*e) / (1e-10 + r)
__device__ __forceinline__ float bpr_term_e(float e, float& dterm);
__device__ __forceinline__ float bpr_term_fast(float x, float& dterm) { return bpr_term_e(__expf(-x), dterm); }
// ... from e = exp(-x) itself: the outer form has x_ij = a_j + c_i, so exp(-x_ij) = exp(-a_j) exp(-c_i) is ONE multiplication per
// pair instead of an exponential (B exponentials per row block and per thread column instead of B^2; |a|, |c| < 40 or the
// plain form is used: no overflow of a factor)
__device__ __forceinline__ float bpr_term_e(float e, float& dterm) {
    const float r = __builtin_amdgcn_rcpf(1.f + e);
    const float den = 1e-10f + r;
    dterm = -(r * r * e) * __builtin_amdgcn_rcpf(den);
    return -__logf(den);
}

// a2 / c2 (optional): the inputs are the differences a - a2 and c - c2, formed here; ndc (optional) receives -dc.
// Rectangular form: Nc rows (c) x Na columns (a), every term scaled by `inv` (the square loss: Na = Nc = B, inv = 1 / B^2);
// a row-sharded step evaluates its rows / columns of the GLOBAL batch's matrix with it (fr_bpr_outer_rect).
__global__ __launch_bounds__(256) void bpr_outer_kernel(const float* __restrict__ a, const float* __restrict__ a2,
                                                        const float* __restrict__ c, const float* __restrict__ c2, int Na,
                                                        int Nc, float inv, float* __restrict__ dc, float* __restrict__ ndc,
                                                        float* __restrict__ ga_part, float* __restrict__ loss_part) {
    const int B = Nc;      // rows
    __shared__ float cs[OUTER_ROWS], ecs[OUTER_ROWS];
    __shared__ float red[4][OUTER_ROWS + 1];
    __shared__ int rows_small;
    const int i0 = blockIdx.x * OUTER_ROWS;
    if (threadIdx.x == 0) rows_small = 1;
    __syncthreads();
    if (threadIdx.x < OUTER_ROWS) {
        const float cv = (i0 + threadIdx.x < B) ? c[i0 + threadIdx.x] - (c2 ? c2[i0 + threadIdx.x] : 0.f) : 0.f;
        cs[threadIdx.x] = cv;
        ecs[threadIdx.x] = __expf(-cv);
        if (!(fabsf(cv) < 40.f)) rows_small = 0;
    }
    __syncthreads();
    const bool rows_ok = rows_small != 0;
    const int ni = min(OUTER_ROWS, B - i0);
    float lsum = 0.f;
    float rs[OUTER_ROWS];
#pragma unroll
    for (int i = 0; i < OUTER_ROWS; ++i) rs[i] = 0.f;
    for (int j = threadIdx.x; j < Na; j += 256) {
        const float aj = a2 ? a[j] - a2[j] : a[j];
        float gcol = 0.f;
        if (rows_ok && fabsf(aj) < 40.f) {
            const float ea = __expf(-aj);
#pragma unroll
            for (int i = 0; i < OUTER_ROWS; ++i) {
                if (i < ni) {
                    float dt;
                    lsum += bpr_term_e(ea * ecs[i], dt);
                    gcol += dt;
                    rs[i] += dt;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < OUTER_ROWS; ++i) {
                if (i < ni) {
                    float dt;
                    lsum += bpr_term_fast(aj + cs[i], dt);
                    gcol += dt;
                    rs[i] += dt;
                }
            }
        }
        ga_part[(size_t)blockIdx.x * Na + j] = gcol * inv;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    lsum = wave_sum(lsum);
    if (lane == 0) red[wave][OUTER_ROWS] = lsum;
#pragma unroll
    for (int i = 0; i < OUTER_ROWS; ++i) {
        const float v = wave_sum(rs[i]);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < ni) {
        const int i = threadIdx.x;
        const float v = (((red[0][i] + red[1][i]) + red[2][i]) + red[3][i]) * inv;
        if (dc) dc[i0 + i] = v;
        if (ndc) ndc[i0 + i] = -v;
    }
    if (threadIdx.x == 0)
        loss_part[blockIdx.x] = (((red[0][OUTER_ROWS] + red[1][OUTER_ROWS]) + red[2][OUTER_ROWS]) + red[3][OUTER_ROWS]) * inv;
}

// da[j] = sum over the row blocks, in block order within quarters, quarters in order (fixed order => reproducible)
// (one extra workgroup behind the column blocks sums the loss partials -- sum_partials_kernel's body, its launch saved)
__global__ __launch_bounds__(256) void bpr_outer_reduce_kernel(const float* __restrict__ ga_part, int nblk, int B,
                                                               float* __restrict__ da, float* __restrict__ nda,
                                                               const float* __restrict__ loss_part, float* __restrict__ loss) {
    __shared__ float red[4][64];
    if (loss && blockIdx.x == gridDim.x - 1) {
        float a = 0.f;
        for (int q = threadIdx.x; q < nblk; q += 256) a += loss_part[q];
        a = wave_sum(a);
        if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = a;
        __syncthreads();
        if (threadIdx.x == 0) loss[0] = (((red[0][0] + red[0][1]) + red[0][2]) + red[0][3]) * 1.f;
        return;
    }
    const int e = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + e;
    const int per = (nblk + 3) / 4, k0 = part * per, k1 = min(nblk, k0 + per);
    float s = 0.f;
    if (j < B) {
#pragma unroll 16
        for (int k = k0; k < k1; ++k) s += ga_part[(size_t)k * B + j];   // (sixteen loads in flight; the sum stays in k order)
    }
    red[part][e] = s;
    __syncthreads();
    if (part == 0 && j < B) {
        const float v = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
        da[j] = v;
        if (nda) nda[j] = -v;
    }
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ part, int n, float scale,
                                                           float* __restrict__ out) {
    __shared__ float red[4];
    float a = 0.f;
    for (int q = threadIdx.x; q < n; q += 256) a += part[q];
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (((red[0] + red[1]) + red[2]) + red[3]) * scale;
}

// softmax cross-entropy, mean over rows: one thread per row, C <= 64 classes
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* __restrict__ logits, const long long* __restrict__ label,
                                                         int M, int C, float* __restrict__ dlogits,
                                                         float* __restrict__ part, uint32_t* err) {
    __shared__ float red[4];
    const int m = blockIdx.x * 256 + threadIdx.x;
    float l = 0.f;
    if (m < M) {
        const float* z = logits + (size_t)m * C;
        float mx = z[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, z[c]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += __expf(z[c] - mx);
        long long y = label[m];
        if (y < 0 || y >= C) {
            if (err) atomicOr(err, FR_DEV_ERR_INDEX_RANGE);
            y = 0;
        }
        const float lse = mx + __logf(se);
        l = lse - z[y];
        for (int c = 0; c < C; ++c)
            dlogits[(size_t)m * C + c] = (__expf(z[c] - lse) - (c == (int)y ? 1.f : 0.f)) / (float)M;
    }
    l = wave_sum(l);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = l;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

}  // namespace fr

using namespace fr;

extern "C" int fr_rowdot_fwd(const float* a, const float* b, int64_t B, int32_t dim, float* out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FR_CHECK_ARG(a && b && out && B >= 1 && dim >= 1, "fr_rowdot_fwd: bad argument");
    ProfScope prof(K_ROWDOT, stream);
    FR_LAUNCH(prof, rowdot_fwd_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, stream, a, b, (int)B, (int)dim, out);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_rowdot_rep_fwd(const float* a, const float* b, int64_t A, int32_t reps, int32_t dim, float* out,
                                 void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FR_CHECK_ARG(a && b && out && A >= 1 && reps >= 1 && dim >= 1, "fr_rowdot_rep_fwd: bad argument");
    ProfScope prof(K_ROWDOT, stream);
    FR_LAUNCH(prof, rowdot_rep_fwd_kernel, dim3((unsigned)((A + 3) / 4)), dim3(256), 0, stream, a, b, (int)A, (int)reps,
              (int)dim, out);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_rowdot_rep_bwd(const float* g, const float* a, const float* b, int64_t A, int32_t reps, int32_t dim,
                                 float* da, float* db, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FR_CHECK_ARG(g && a && b && (da || db) && A >= 1 && reps >= 1 && dim >= 1, "fr_rowdot_rep_bwd: bad argument");
    ProfScope prof(K_ROWDOT, stream);
    FR_LAUNCH(prof, rowdot_rep_bwd_kernel, dim3((unsigned)((A + 3) / 4)), dim3(256), 0, stream, g, a, b, (int)A, (int)reps,
              (int)dim, da, db);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_rowdot_rep_bwd_sep(const float* g, const float* a, const float* b, int64_t A, int32_t reps, int32_t dim,
                                     float* da_sep, float* db, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FR_CHECK_ARG(g && a && b && (da_sep || db) && A >= 1 && reps >= 1 && dim >= 1, "fr_rowdot_rep_bwd_sep: bad argument");
    ProfScope prof(K_ROWDOT, stream);
    FR_LAUNCH(prof, rowdot_rep_bwd_sep_kernel, dim3((unsigned)((A + 3) / 4)), dim3(256), 0, stream, g, a, b, (int)A, (int)reps,
              (int)dim, da_sep, db);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_rowdot_bwd(const float* g, const float* a, const float* b, int64_t B, int32_t dim, float* da, float* db,
                             void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FR_CHECK_ARG(g && a && b && (da || db) && B >= 1 && dim >= 1, "fr_rowdot_bwd: bad argument");
    ProfScope prof(K_ROWDOT, stream);
    FR_LAUNCH(prof, rowdot_bwd_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, stream, g, a, b, (int)B, (int)dim, da,
              db);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" size_t fr_bpr_workspace_bytes(int64_t B, int32_t outer) {
    if (B < 1) return 0;
    const size_t nb = outer ? (size_t)(B + fr::OUTER_ROWS - 1) / fr::OUTER_ROWS : (size_t)(B + 255) / 256;
    return align_up(nb * 4, 256) + (outer ? nb * (size_t)B * 4 : 0);
}

// BPRLoss (loss.py:45-47): loss[0] = mean -log(1e-10 + sigmoid(pos - neg)); dpos, dneg = dLoss/d(pos, neg)
extern "C" int fr_bpr(const float* pos, const float* neg, int64_t B, float* loss, float* dpos, float* dneg, void* ws,
                      size_t ws_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FR_CHECK_ARG(pos && neg && loss && dpos && dneg && ws && B >= 1 && ws_bytes >= fr_bpr_workspace_bytes(B, 0),
                 "fr_bpr: bad argument");
    const int nb = (int)((B + 255) / 256);
    {
        ProfScope prof(K_BPR, stream);
        FR_LAUNCH(prof, bpr_kernel, dim3(nb), dim3(256), 0, stream, pos, neg, (int)B, dpos, dneg, (float*)ws);
    }
    FR_CHECK_LAUNCH();
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, stream, (const float*)ws, nb, 1.f / (float)B, loss);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

// The [B] + [B,1] broadcast of PFCN_BiasedMF: loss[0] = mean_{i,j} -log(1e-10 + sigmoid(a_j + c_i)),
// a = (u.p - u.n) per row, c = (pos item bias - neg item bias) per row; da, dc = dLoss/d(a, c).
// fr_bpr_outer2: the same on the four score / bias columns themselves -- a = pos - neg and c = pos_bias - neg_bias are
// formed in the kernel, and the gradients of all four come out (d_neg = -d_pos, d_neg_bias = -d_pos_bias).
static int bpr_outer_impl(const float* a, const float* a2, const float* c, const float* c2, int64_t B, float* loss, float* da,
                          float* nda, float* dc, float* ndc, void* ws, size_t ws_bytes, hipStream_t stream);

extern "C" int fr_bpr_outer2(const float* pos, const float* neg, const float* pos_bias, const float* neg_bias, int64_t B,
                             float* loss, float* d_pos, float* d_neg, float* d_pos_bias, float* d_neg_bias, void* ws,
                             size_t ws_bytes, void* stream_) {
    FR_CHECK_ARG(pos && neg && pos_bias && neg_bias && d_neg && d_neg_bias, "fr_bpr_outer2: bad argument");
    return bpr_outer_impl(pos, neg, pos_bias, neg_bias, B, loss, d_pos, d_neg, d_pos_bias, d_neg_bias, ws, ws_bytes,
                          (hipStream_t)stream_);
}

extern "C" int fr_bpr_outer(const float* a, const float* c, int64_t B, float* loss, float* da, float* dc, void* ws,
                            size_t ws_bytes, void* stream_) {
    return bpr_outer_impl(a, nullptr, c, nullptr, B, loss, da, nullptr, dc, nullptr, ws, ws_bytes, (hipStream_t)stream_);
}

static int bpr_outer_impl(const float* a, const float* a2, const float* c, const float* c2, int64_t B, float* loss, float* da,
                          float* nda, float* dc, float* ndc, void* ws, size_t ws_bytes, hipStream_t stream) {
    FR_CHECK_ARG(a && c && loss && da && dc && ws && B >= 1 && ws_bytes >= fr_bpr_workspace_bytes(B, 1),
                 "fr_bpr_outer: bad argument");
    const int nb = (int)((B + OUTER_ROWS - 1) / OUTER_ROWS);
    float* loss_part = (float*)ws;
    float* ga_part = (float*)((char*)ws + align_up((size_t)nb * 4, 256));
    {
        ProfScope prof(K_BPR, stream);
        FR_LAUNCH(prof, bpr_outer_kernel, dim3(nb), dim3(256), 0, stream, a, a2, c, c2, (int)B, (int)B,
                  1.f / ((float)B * (float)B), dc, ndc, ga_part, loss_part);
    }
    FR_CHECK_LAUNCH();
    hipLaunchKernelGGL(bpr_outer_reduce_kernel, dim3((unsigned)((B + 63) / 64) + 1), dim3(256), 0, stream,
                       (const float*)ga_part, nb, (int)B, da, nda, (const float*)loss_part, loss);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

// The same term matrix for Nc rows and Na columns that need not be the same batch: loss[0] = inv * sum_{i < Nc, j < Na}
// -log(1e-10 + sigmoid(a_j + c_i)), da[j] = inv * sum_i f'(a_j + c_i), dc[i] = inv * sum_j f'(a_j + c_i) (either may be
// NULL).  A row-sharded PFCN_BiasedMF step evaluates its share of the GLOBAL batch's [G B, G B] matrix with two calls
// (its columns against every row, its rows against every column; inv = 1 / (G B)^2).
extern "C" size_t fr_bpr_outer_rect_workspace_bytes(int64_t Na, int64_t Nc) {
    if (Na < 1 || Nc < 1) return 0;
    const size_t nb = (size_t)(Nc + fr::OUTER_ROWS - 1) / fr::OUTER_ROWS;
    return align_up(nb * 4, 256) + nb * (size_t)Na * 4;
}

extern "C" int fr_bpr_outer_rect(const float* a, int64_t Na, const float* c, int64_t Nc, float inv, float* loss, float* da,
                                 float* dc, void* ws, size_t ws_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FR_CHECK_ARG(a && c && loss && ws && Na >= 1 && Nc >= 1 && Na < (1ll << 31) && Nc < (1ll << 31) &&
                 ws_bytes >= fr_bpr_outer_rect_workspace_bytes(Na, Nc), "fr_bpr_outer_rect: bad argument");
    const int nb = (int)((Nc + OUTER_ROWS - 1) / OUTER_ROWS);
    float* loss_part = (float*)ws;
    float* ga_part = (float*)((char*)ws + align_up((size_t)nb * 4, 256));
    {
        ProfScope prof(K_BPR, stream);
        FR_LAUNCH(prof, bpr_outer_kernel, dim3(nb), dim3(256), 0, stream, a, (const float*)nullptr, c, (const float*)nullptr,
                  (int)Na, (int)Nc, inv, dc, (float*)nullptr, ga_part, loss_part);
    }
    FR_CHECK_LAUNCH();
    if (da) {
        hipLaunchKernelGGL(bpr_outer_reduce_kernel, dim3((unsigned)((Na + 63) / 64) + 1), dim3(256), 0, stream,
                           (const float*)ga_part, nb, (int)Na, da, (float*)nullptr, (const float*)loss_part, loss);
        FR_CHECK_LAUNCH();
    } else {
        hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, stream, (const float*)loss_part, nb, 1.f, loss);
        FR_CHECK_LAUNCH();
    }
    return FR_OK;
}

// nn.CrossEntropyLoss (mean): loss[0], dlogits [M, C]
extern "C" int fr_softmax_ce(const float* logits, const int64_t* label, int64_t M, int32_t C, float* loss, float* dlogits,
                             void* ws, size_t ws_bytes, uint32_t* err_flag, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int nb = (int)((M + 255) / 256);
    FR_CHECK_ARG(logits && label && loss && dlogits && ws && M >= 1 && C >= 1 && C <= 64 && ws_bytes >= (size_t)nb * 4,
                 "fr_softmax_ce: bad argument");
    hipLaunchKernelGGL(softmax_ce_kernel, dim3(nb), dim3(256), 0, stream, logits, (const long long*)label, (int)M, (int)C,
                       dlogits, (float*)ws, err_flag);
    FR_CHECK_LAUNCH();
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, stream, (const float*)ws, nb, 1.f / (float)M, loss);
    FR_CHECK_LAUNCH();
    return FR_OK;
}
