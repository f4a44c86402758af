// Dense layers of the fair models on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact f32, 1e-4 parity
// rules out bf16 inputs -- SURVEY.md §7 hard part 7).
//
// Replaces recbole/model/layers.py:56-85 (MLPLayers = per layer Dropout -> Linear -> [BatchNorm1d] -> activation,
// INCLUDING the last layer) and its autograd: nfcf.py:40 (scorer [2D,128,64,1]), pfcn_biasedmf.py:113-142
// (filters / discriminators).  One kernel per layer direction:
//   linear_fwd        Y  = act((X o mask*scale) W^T + b)
//   linear_bwd_input  dX = ((dY o act'(Y)) W) o mask*scale
//   linear_bwd_weight dW = (dY o act'(Y))^T (X o mask*scale)   (split over the batch, fixed-order reduction), db
// Tiles: 256 threads = 4 waves, 64x64 output tile, each wave one 32x32 MFMA tile, 32-deep K chunks through LDS.
#include <stdlib.h>

#include <algorithm>

#include "common.hpp"
#include "kernels.hpp"
#include "mlp_glds.hpp"
#include "dropout.hpp"
#include "mlp_act.hpp"
#include "mlp_bn_math.hpp"

namespace fr {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// An [M, K] fp32 matrix that may be the column-wise concatenation of two row-major blocks (cat(U[u], I[i])).
struct CatMat {
    const float* a;
    const float* b;   // may be null
    int ka, kb;       // widths; K = ka + kb
    __device__ __forceinline__ float at(long long m, int k) const {
        return k < ka ? a[m * ka + k] : b[m * kb + (k - ka)];
    }
};

struct CatOut {
    float* a;
    float* b;
    int ka, kb;
    __device__ __forceinline__ void put(long long m, int k, float v) const {
        if (k < ka) a[m * ka + k] = v;
        else b[m * kb + (k - ka)] = v;
    }
};

static constexpr int TM = 64, TN = 64, TK = 32, LDT = TK + 1;

__device__ __forceinline__ void store_tile(const f32x16& acc, int wm, int wn, int lane, int m0, int n0, int M, int N,
                                           const float* __restrict__ bias, int act, float* __restrict__ Y) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int col = n0 + wn * 32 + (lane & 31);
        if (row < M && col < N) Y[(size_t)row * N + col] = act_fwd(acc[r] + (bias ? bias[col] : 0.f), act);
    }
}

// Y[M,N] = act((X o mask*scale) W^T + b);  W is [N, K] row-major (nn.Linear.weight)
__global__ __launch_bounds__(256) void linear_fwd_kernel(CatMat X, const unsigned char* __restrict__ mask, float scale,
                                                         const float* __restrict__ W, const float* __restrict__ bias,
                                                         int M, int N, int K, int act, float* __restrict__ Y) {
    __shared__ float Xs[TM * LDT];
    __shared__ float Ws[TN * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * TM, n0 = blockIdx.y * TN;
    f32x16 acc = {0};
    constexpr int Q = (TM * TK) / 256;
    float xq[Q], wq[Q];                      // the next K chunk, fetched while the current one is multiplied
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int e = q * 256 + tid, r = e / TK, c = e % TK;
            const int m = m0 + r, k = k0 + c, n = n0 + r;
            float x = 0.f;
            if (m < M && k < K) {
                x = X.at(m, k);
                if (mask) x = mask[(size_t)m * K + k] ? x * scale : 0.f;
            }
            xq[q] = x;
            wq[q] = (n < N && k < K) ? W[(size_t)n * K + k] : 0.f;
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += TK) {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int e = q * 256 + tid, r = e / TK, c = e % TK;
            Xs[r * LDT + c] = xq[q];
            Ws[r * LDT + c] = wq[q];
        }
        __syncthreads();
        if (k0 + TK < K) fetch(k0 + TK);
#pragma unroll
        for (int kk = 0; kk < TK; kk += 2) {
            const float a = Xs[(wm * 32 + (lane & 31)) * LDT + kk + (lane >> 5)];
            const float b = Ws[(wn * 32 + (lane & 31)) * LDT + kk + (lane >> 5)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    store_tile(acc, wm, wn, lane, m0, n0, M, N, bias, act, Y);
}


// ---- forward, fast form ---------------------------------------------------------------------------------------------
// The same product for the shapes the models use (K a multiple of 4 and both blocks of a concatenated input a multiple of
// 4 wide), with the operand traffic cut to what the f32 matrix pipe (64 cycles per 32x32x2 MFMA) can be fed with:
//   * a workgroup owns 64 rows x BN columns (BN = 128: every wave a 32 x 64 strip = two MFMA tiles that share their A
//     fragment; BN = 64 when that would leave CUs without a workgroup), so X is read from HBM once per 128 columns;
//   * 32-deep K chunks in LDS with the K index permuted so that the 16 values a lane feeds to the chunk's 16 MFMA steps
//     (k = 2 i + lane / 32) are contiguous: four ds_read_b128 per fragment instead of sixteen ds_read_b32 (row stride
//     36 floats: conflict-free for the 16 rows of a b128 lane group);
//   * two LDS buffers: the global loads of chunk c + 1 are in flight while chunk c is multiplied, one barrier per chunk.
template <int BN>
__global__ __launch_bounds__(256) void linear_fwd_fast_kernel(CatMat X, const unsigned char* __restrict__ mask, float scale,
                                                              const float* __restrict__ W, const float* __restrict__ bias,
                                                              int M, int N, int K, int act, float* __restrict__ Y) {
    constexpr int BM = 64, CK = 32, LD = 36, JT = BN / 64;      // JT MFMA tiles per wave along N
    constexpr int AQ = BM * CK / 4 / 256, BQ = BN * CK / 4 / 256;   // float4 loads per thread and chunk
    __shared__ __align__(16) float As[2][BM * LD];
    __shared__ __align__(16) float Bs[2][BN * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    f32x16 acc[JT];
#pragma unroll
    for (int j = 0; j < JT; ++j) acc[j] = f32x16{0};
    float4 aq[AQ], bq[BQ];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < AQ; ++q) {
            const int e = q * 256 + tid, r = e >> 3, c = (e & 7) * 4;       // 8 float4 per 32-wide row
            const int m = m0 + r, k = k0 + c;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < M && k < K) {
                x = k < X.ka ? *reinterpret_cast<const float4*>(X.a + (size_t)m * X.ka + k)
                             : *reinterpret_cast<const float4*>(X.b + (size_t)m * X.kb + (k - X.ka));
                if (mask) {
                    const uchar4 mk = *reinterpret_cast<const uchar4*>(mask + (size_t)m * K + k);
                    x.x = mk.x ? x.x * scale : 0.f; x.y = mk.y ? x.y * scale : 0.f;
                    x.z = mk.z ? x.z * scale : 0.f; x.w = mk.w ? x.w * scale : 0.f;
                }
            }
            aq[q] = x;
        }
#pragma unroll
        for (int q = 0; q < BQ; ++q) {
            const int e = q * 256 + tid, r = e >> 3, c = (e & 7) * 4;
            const int n = n0 + r, k = k0 + c;
            bq[q] = (n < N && k < K) ? *reinterpret_cast<const float4*>(W + (size_t)n * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    // k = c .. c + 3 of a row go to positions (c / 2, c / 2 + 1) of the even half and of the odd half
    auto stage = [&](int buf) {
#pragma unroll
        for (int q = 0; q < AQ; ++q) {
            const int e = q * 256 + tid, r = e >> 3, c = (e & 7) * 4;
            float* d = &As[buf][r * LD + (c >> 1)];
            *reinterpret_cast<float2*>(d) = make_float2(aq[q].x, aq[q].z);
            *reinterpret_cast<float2*>(d + 16) = make_float2(aq[q].y, aq[q].w);
        }
#pragma unroll
        for (int q = 0; q < BQ; ++q) {
            const int e = q * 256 + tid, r = e >> 3, c = (e & 7) * 4;
            float* d = &Bs[buf][r * LD + (c >> 1)];
            *reinterpret_cast<float2*>(d) = make_float2(bq[q].x, bq[q].z);
            *reinterpret_cast<float2*>(d + 16) = make_float2(bq[q].y, bq[q].w);
        }
    };
    fetch(0);
    stage(0);
    __syncthreads();
    const int half = (lane >> 5) * 16, lr = lane & 31;
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += CK, buf ^= 1) {
        const bool more = k0 + CK < K;
        if (more) fetch(k0 + CK);
        float4 af[4], bf[JT][4];
        const float* ap = &As[buf][(wm * 32 + lr) * LD + half];
#pragma unroll
        for (int v = 0; v < 4; ++v) af[v] = *reinterpret_cast<const float4*>(ap + 4 * v);
#pragma unroll
        for (int j = 0; j < JT; ++j) {
            const float* bp = &Bs[buf][((wn * JT + j) * 32 + lr) * LD + half];
#pragma unroll
            for (int v = 0; v < 4; ++v) bf[j][v] = *reinterpret_cast<const float4*>(bp + 4 * v);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float a4[4] = {af[v].x, af[v].y, af[v].z, af[v].w};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int j = 0; j < JT; ++j) {
                    const float b4[4] = {bf[j][v].x, bf[j][v].y, bf[j][v].z, bf[j][v].w};
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s], b4[s], acc[j], 0, 0, 0);
                }
            }
        }
        if (more) stage(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < JT; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int col = n0 + (wn * JT + j) * 32 + lr;
            if (row < M && col < N) Y[(size_t)row * N + col] = act_fwd(acc[j][r] + (bias ? bias[col] : 0.f), act);
        }
    }
}

// dX[M,K] = ((dY o act'(Y)) W) o mask*scale
__global__ __launch_bounds__(256) void linear_bwd_input_kernel(const float* __restrict__ dY, const float* __restrict__ Y,
                                                               int act, const float* __restrict__ W,
                                                               const unsigned char* __restrict__ mask, float scale,
                                                               int M, int N, int K, CatOut dX) {
    __shared__ float Gs[TM * LDT];          // [m][n chunk]
    __shared__ float Ws[TK * (TN + 1)];     // [n chunk][k tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * TM, c0 = blockIdx.y * TN;   // output tile: rows m, columns k
    f32x16 acc = {0};
    constexpr int Q = (TM * TK) / 256;
    float gq[Q], wq[Q];                      // the next N chunk, fetched while the current one is multiplied
    auto fetch = [&](int n0) {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int e = q * 256 + tid;
            {
                const int r = e / TK, c = e % TK;
                const int m = m0 + r, n = n0 + c;
                float g = 0.f;
                if (m < M && n < N) g = dY[(size_t)m * N + n] * act_bwd(Y[(size_t)m * N + n], act);
                gq[q] = g;
            }
            {
                const int r = e / TN, c = e % TN;          // r: n within chunk, c: k within tile
                const int n = n0 + r, k = c0 + c;
                wq[q] = (n < N && k < K) ? W[(size_t)n * K + k] : 0.f;
            }
        }
    };
    fetch(0);
    for (int n0 = 0; n0 < N; n0 += TK) {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int e = q * 256 + tid;
            Gs[(e / TK) * LDT + e % TK] = gq[q];
            Ws[(e / TN) * (TN + 1) + e % TN] = wq[q];
        }
        __syncthreads();
        if (n0 + TK < N) fetch(n0 + TK);
#pragma unroll
        for (int kk = 0; kk < TK; kk += 2) {
            const float a = Gs[(wm * 32 + (lane & 31)) * LDT + kk + (lane >> 5)];
            const float b = Ws[(kk + (lane >> 5)) * (TN + 1) + wn * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int col = c0 + wn * 32 + (lane & 31);
        if (row < M && col < K) {
            float v = acc[r];
            if (mask) v = mask[(size_t)row * K + col] ? v * scale : 0.f;
            dX.put(row, col, v);
        }
    }
}

// slab[s][N][K] = sum over the rows of split s of (dY o act'(Y))[m,n] * (X o mask*scale)[m,k]
__global__ __launch_bounds__(256) void linear_bwd_weight_kernel(const float* __restrict__ dY, const float* __restrict__ Y,
                                                                int act, CatMat X, const unsigned char* __restrict__ mask,
                                                                float scale, int M, int N, int K, int rows_per_split,
                                                                float* __restrict__ slab, float* __restrict__ bslab) {
    __shared__ float Gs[TK * (TM + 1)];   // [m chunk][n tile]
    __shared__ float Xs[TK * (TN + 1)];   // [m chunk][k tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int n0 = blockIdx.x * TM, c0 = blockIdx.y * TN, s = blockIdx.z;
    const int mlo = s * rows_per_split, mhi = min(M, mlo + rows_per_split);
    f32x16 acc = {0};
    float bsum = 0.f;                        // bias gradient of column n0 + tid (blocks of the first k tile, tid < TM)
    const bool do_bias = bslab && blockIdx.y == 0 && tid < TM;
    constexpr int Q = (TM * TK) / 256;
    float gq[Q], xq[Q];                      // the next chunk's operands, fetched while the current chunk is multiplied
    auto fetch = [&](int mb) {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int e = q * 256 + tid;
            const int r = e / TM, c = e % TM;     // r: m within chunk, c: column within tile
            const int m = mb + r;
            const int n = n0 + c, k = c0 + c;
            float g = 0.f, x = 0.f;
            if (m < mhi && n < N) g = dY[(size_t)m * N + n] * act_bwd(Y[(size_t)m * N + n], act);
            if (m < mhi && k < K) {
                x = X.at(m, k);
                if (mask) x = mask[(size_t)m * K + k] ? x * scale : 0.f;
            }
            gq[q] = g;
            xq[q] = x;
        }
    };
    if (mlo < mhi) fetch(mlo);
    for (int mb = mlo; mb < mhi; mb += TK) {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int e = q * 256 + tid;
            const int r = e / TM, c = e % TM;
            Gs[r * (TM + 1) + c] = gq[q];
            Xs[r * (TN + 1) + c] = xq[q];
        }
        __syncthreads();
        if (mb + TK < mhi) fetch(mb + TK);
#pragma unroll
        for (int kk = 0; kk < TK; kk += 2) {
            const float a = Gs[(kk + (lane >> 5)) * (TM + 1) + wm * 32 + (lane & 31)];
            const float b = Xs[(kk + (lane >> 5)) * (TN + 1) + wn * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        if (do_bias) {
#pragma unroll
            for (int r = 0; r < TK; ++r) bsum += Gs[r * (TM + 1) + tid];
        }
        __syncthreads();
    }
    if (do_bias && n0 + tid < N) bslab[(size_t)s * N + n0 + tid] = bsum;
    float* out = slab + (size_t)s * N * K;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = n0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int col = c0 + wn * 32 + (lane & 31);
        if (row < N && col < K) out[(size_t)row * K + col] = acc[r];
    }
}

// dW[i] = sum_s slab[s][i] and db[n] = sum_s bslab[s][n]: 64 outputs per workgroup, the splits of an output summed by 4
// threads over contiguous quarters and combined in quarter order (fixed order => reproducible)
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ bslab,
                                                          int splits, long long n, int nb, float* __restrict__ out,
                                                          float* __restrict__ db) {
    __shared__ float red[4][64];
    const int e = threadIdx.x & 63, part = threadIdx.x >> 6;
    const long long i = (long long)blockIdx.x * 64 + e;
    const int per = (splits + 3) / 4, s0 = part * per, s1 = min(splits, s0 + per);
    float a = 0.f;
    if (i < n) {
        for (int s = s0; s < s1; ++s) a += slab[(size_t)s * n + i];
    } else if (i < n + nb) {
        const long long j = i - n;
        for (int s = s0; s < s1; ++s) a += bslab[(size_t)s * nb + j];
    }
    red[part][e] = a;
    __syncthreads();
    if (part == 0) {
        const float t = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
        if (i < n) out[i] = t;
        else if (i < n + nb) db[i - n] = t;
    }
}

}  // namespace fr

using namespace fr;

static inline int act_ok(int act) { return act >= ACT_NONE && act <= ACT_TANH; }

// ---- layers with ONE output (the last layer of a scorer / discriminator: recbole/model/layers.py MLPLayers([..., 1])) ------
// A [M, K] x [K] product is a row-wise dot: the matrix tiles above spend a 64 x 64 tile (and three launches in the backward
// pass) on a single column.  16 lanes per row, one float4 each per 64 columns; K % 64 == 0, K <= 512.
static constexpr int N1_MAXQ = 8;

__global__ __launch_bounds__(256) void linear_n1_fwd_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                            const float* __restrict__ bias, int M, int K, int act,
                                                            float* __restrict__ Y) {
    const int sub = threadIdx.x & 15;
    const long long m = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    float acc = 0.f;
    if (m < M) {
        const float4* x = reinterpret_cast<const float4*>(X + m * K);
        const float4* w = reinterpret_cast<const float4*>(W);
        for (int q = sub; q < K / 4; q += 16) {
            const float4 a = x[q], b = w[q];
            acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
        }
    }
    acc = group_sum<16>(acc);
    if (m < M && sub == 0) Y[m] = act_fwd(acc + (bias ? bias[0] : 0.f), act);
}

// Backward of the same layer in one pass over X: dz = dY o act'(Y);  dX[m, :] = dz[m] * W;  per-workgroup partial sums of
// dW = sum_m dz[m] X[m, :] and db = sum_m dz[m] into slab[blockIdx][K] / bslab[blockIdx] (summed by slab_reduce_kernel in
// workgroup order: reproducible).
__global__ __launch_bounds__(256) void linear_n1_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ Y, int act,
                                                            const float* __restrict__ X, const float* __restrict__ W, int M,
                                                            int K, int rows_per_block, float relu_scale,
                                                            float* __restrict__ dX, float* __restrict__ slab,
                                                            float* __restrict__ bslab) {
    const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int nq = K / 64;                                       // float4 columns per lane
    float4 acc[N1_MAXQ];
#pragma unroll
    for (int q = 0; q < N1_MAXQ; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    float accb = 0.f;
    float4 w[N1_MAXQ];
#pragma unroll
    for (int q = 0; q < N1_MAXQ; ++q)
        w[q] = q < nq ? reinterpret_cast<const float4*>(W)[q * 16 + sub] : make_float4(0.f, 0.f, 0.f, 0.f);
    const long long m0 = (long long)blockIdx.x * rows_per_block;
    const long long m1 = m0 + rows_per_block < M ? m0 + rows_per_block : M;
    for (long long m = m0 + grp; m < m1; m += 16) {
        const float dz = dY[m] * act_bwd(Y[m], act);
        const float4* x = reinterpret_cast<const float4*>(X + m * K);
        float4* dx = dX ? reinterpret_cast<float4*>(dX + m * K) : nullptr;
#pragma unroll
        for (int q = 0; q < N1_MAXQ; ++q)
            if (q < nq) {
                const float4 a = x[q * 16 + sub];
                acc[q].x = fmaf(dz, a.x, acc[q].x); acc[q].y = fmaf(dz, a.y, acc[q].y);
                acc[q].z = fmaf(dz, a.z, acc[q].z); acc[q].w = fmaf(dz, a.w, acc[q].w);
                if (dx) {
                    float4 o = make_float4(dz * w[q].x, dz * w[q].y, dz * w[q].z, dz * w[q].w);
                    if (relu_scale > 0.f) {   // on through the dropped ReLU that produced X (X = relu(z) o keep)
                        o.x = a.x > 0.f ? o.x * relu_scale : 0.f; o.y = a.y > 0.f ? o.y * relu_scale : 0.f;
                        o.z = a.z > 0.f ? o.z * relu_scale : 0.f; o.w = a.w > 0.f ? o.w * relu_scale : 0.f;
                    }
                    dx[q * 16 + sub] = o;
                }
            }
        if (sub == 0) accb += dz;
    }
    // the 16 row groups of the workgroup, combined in group order
    __shared__ float4 red[16][16];
    __shared__ float redb[16];
    float* out = slab + (size_t)blockIdx.x * K;
#pragma unroll
    for (int q = 0; q < N1_MAXQ; ++q) {
        if (q >= nq) break;
        red[grp][sub] = acc[q];
        __syncthreads();
        if (grp == 0) {
            float4 s = red[0][sub];
            for (int g = 1; g < 16; ++g) {
                const float4 t = red[g][sub];
                s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
            }
            reinterpret_cast<float4*>(out)[q * 16 + sub] = s;
        }
        __syncthreads();
    }
    if (sub == 0) redb[grp] = accb;
    __syncthreads();
    if (threadIdx.x == 0 && bslab) {
        float s = redb[0];
        for (int g = 1; g < 16; ++g) s += redb[g];
        bslab[blockIdx.x] = s;
    }
}

static inline bool n1_ok(int32_t N, int K, int32_t k1, const void* mask, const void* x0, const void* W) {
    static const bool off = getenv("FAIRREC_LINEAR_SLOW") != nullptr || getenv("FAIRREC_LINEAR_NO_N1") != nullptr;
    return !off && N == 1 && k1 == 0 && !mask && K % 64 == 0 && K <= 64 * N1_MAXQ && (((uintptr_t)x0 | (uintptr_t)W) & 15) == 0;
}


extern "C" int fr_linear_fwd(const float* x0, int32_t k0, const float* x1, int32_t k1, const uint8_t* mask, float scale,
                             const float* W, const float* bias, int64_t M, int32_t N, int32_t act, float* Y,
                             void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FR_CHECK_ARG(x0 && W && Y && M >= 1 && N >= 1 && k0 >= 1 && k1 >= 0 && (k1 == 0 || x1) && act_ok(act),
                 "fr_linear_fwd: bad argument");
    const int K = k0 + k1;
    CatMat X{x0, x1, k0, k1};
    prof_work(K_LINEAR_FWD, 2.0 * (double)M * N * K);
    ProfScope prof(K_LINEAR_FWD, stream);
    if (n1_ok(N, K, k1, mask, x0, W)) {
        FR_LAUNCH(prof, linear_n1_fwd_kernel, dim3((unsigned)((M + 15) / 16)), dim3(256), 0, stream, x0, W, bias, (int)M, K,
                  (int)act, Y);
        FR_CHECK_LAUNCH();
        return FR_OK;
    }
    const bool aligned = K % 4 == 0 && k0 % 4 == 0 && k1 % 4 == 0 && ((uintptr_t)x0 & 15) == 0 && ((uintptr_t)W & 15) == 0 &&
                         (!x1 || ((uintptr_t)x1 & 15) == 0) && (!mask || ((uintptr_t)mask & 3) == 0);
    static const bool slow_only = getenv("FAIRREC_LINEAR_SLOW") != nullptr;
    static const bool no_glds = getenv("FAIRREC_LINEAR_NO_GLDS") != nullptr;
    if (aligned && !slow_only && !no_glds && !mask && K % 32 == 0 && k0 % 32 == 0 && N >= 8) {
        // LDS-DMA kernels (mlp_glds.hip): no dropout mask, 32-element reduction chunks
        return glds_linear_fwd(GlMat{x0, x1, k0, k1, k0}, W, bias, M, (int)N, K, (int)act, Y, stream);
    }
    if (aligned && !slow_only) {
        const long long mb = (M + 63) / 64;
        if (N > 64 && mb * ((N + 127) / 128) >= 256) {      // wide tiles once they still fill the chip
            FR_LAUNCH(prof, linear_fwd_fast_kernel<128>, dim3((unsigned)mb, (unsigned)((N + 127) / 128)), dim3(256), 0, stream,
                      X, mask, scale, W, bias, (int)M, (int)N, K, (int)act, Y);
        } else {
            FR_LAUNCH(prof, linear_fwd_fast_kernel<64>, dim3((unsigned)mb, (unsigned)((N + 63) / 64)), dim3(256), 0, stream, X,
                      mask, scale, W, bias, (int)M, (int)N, K, (int)act, Y);
        }
    } else {
        FR_LAUNCH(prof, linear_fwd_kernel, dim3((unsigned)((M + TM - 1) / TM), (unsigned)((N + TN - 1) / TN)), dim3(256), 0,
                  stream, X, mask, scale, W, bias, (int)M, (int)N, K, (int)act, Y);
    }
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_linear_bwd_input(const float* dY, const float* Y, int32_t act, const float* W, const uint8_t* mask,
                                   float scale, int64_t M, int32_t N, float* dx0, int32_t k0, float* dx1, int32_t k1,
                                   void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FR_CHECK_ARG(dY && Y && W && dx0 && M >= 1 && N >= 1 && k0 >= 1 && k1 >= 0 && (k1 == 0 || dx1) && act_ok(act),
                 "fr_linear_bwd_input: bad argument");
    const int K = k0 + k1;
    prof_work(K_LINEAR_BWD_INPUT, 2.0 * (double)M * N * K);
    CatOut dX{dx0, dx1, k0, k1};
    static const bool no_glds = getenv("FAIRREC_LINEAR_NO_GLDS") != nullptr || getenv("FAIRREC_LINEAR_SLOW") != nullptr;
    if (!no_glds && !mask && act == ACT_NONE && N % 32 == 0 && K % 32 == 0 && k0 % 32 == 0 && ((uintptr_t)dY & 15) == 0 &&
        ((uintptr_t)W & 15) == 0)
        return glds_linear_bwd_input(dY, W, M, (int)N, K, dx0, k0, dx1, k1, stream);
    ProfScope prof(K_LINEAR_BWD_INPUT, stream);
    FR_LAUNCH(prof, linear_bwd_input_kernel, dim3((unsigned)((M + TM - 1) / TM), (unsigned)((K + TN - 1) / TN)), dim3(256),
              0, stream, dY, Y, (int)act, W, mask, scale, (int)M, (int)N, K, dX);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

// dA = (dY W) o scale o [Xd > 0]: the input gradient of a layer whose input Xd [M, K] is the previous layer's ReLU output
// dropped in place, taken on through that ReLU in the epilogue (fr_linear_bwd_input + fr_act_bwd_dropped in one launch).
// Fast form only: FR_EUNSUPPORTED unless N % 32 == 0, K % 32 == 0 and the operands are 16-byte aligned.
extern "C" int fr_linear_bwd_input_relu(const float* dY, const float* W, int64_t M, int32_t N, int32_t K, const float* Xd,
                                        float scale, float* dA, void* stream_) {
    FR_CHECK_ARG(dY && W && Xd && dA && M >= 1 && N >= 1 && K >= 1 && scale > 0.f, "fr_linear_bwd_input_relu: bad argument");
    prof_work(K_LINEAR_BWD_INPUT, 2.0 * (double)M * N * K);
    static const bool no_glds = getenv("FAIRREC_LINEAR_NO_GLDS") != nullptr || getenv("FAIRREC_LINEAR_SLOW") != nullptr;
    if (no_glds || N % 32 != 0 || K % 32 != 0 || (((uintptr_t)dY | (uintptr_t)W) & 15) != 0) {
        set_error("fr_linear_bwd_input_relu: shape not supported (N %% 32 == 0, K %% 32 == 0, 16-byte aligned operands)");
        return FR_EUNSUPPORTED;
    }
    return glds_linear_bwd_input(dY, W, M, (int)N, (int)K, dA, (int)K, nullptr, 0, (hipStream_t)stream_, Xd, scale);
}

// dA = (dY W) o act'(Yin): the input gradient of a layer whose input Yin [M, K] is the previous layer's activation output,
// taken on through that activation in the epilogue (fr_linear_bwd_input + the next layer's fr_act_bwd in one launch: one
// pass over an [M, K] tensor and one launch less per hidden layer of an MLP's backward pass; the product is rounded to
// fp32 and then multiplied, exactly as the two launches do).  Fast form only, like fr_linear_bwd_input_relu.
extern "C" int fr_linear_bwd_input_act(const float* dY, const float* W, int64_t M, int32_t N, int32_t K, const float* Yin,
                                       int32_t act, float* dA, void* stream_) {
    FR_CHECK_ARG(dY && W && Yin && dA && M >= 1 && N >= 1 && K >= 1 && act >= 1 && act <= 4, "fr_linear_bwd_input_act: bad argument");
    prof_work(K_LINEAR_BWD_INPUT, 2.0 * (double)M * N * K);
    static const bool no_glds = getenv("FAIRREC_LINEAR_NO_GLDS") != nullptr || getenv("FAIRREC_LINEAR_SLOW") != nullptr;
    if (no_glds || N % 32 != 0 || K % 32 != 0 || (((uintptr_t)dY | (uintptr_t)W) & 15) != 0) {
        set_error("fr_linear_bwd_input_act: shape not supported (N %% 32 == 0, K %% 32 == 0, 16-byte aligned operands)");
        return FR_EUNSUPPORTED;
    }
    return glds_linear_bwd_input(dY, W, M, (int)N, (int)K, dA, (int)K, nullptr, 0, (hipStream_t)stream_, Yin, 1.f, (int)act);
}

// dX = (dY W) o keep, stored, AND the backward statistics of the BatchNorm layer whose (dropped) output X is: per 32-row tile
// and column sum dA and sum dA xhat (dA = dX o act'(Yb)) into bn_ws, exactly where fr_bn_bwd's statistics launch leaves its
// per-chunk sums -- fr_bn_bwd_ex(have_stats = 1) then runs the apply launch alone.  One launch instead of three (product,
// dropout, statistics) per BatchNorm layer of a backward pass.  p = 0: no dropout between the layers.  Fast form only
// (N % 32 == 0, K % 32 == 0, aligned operands, the macro-tile kernels, M <= 32768): FR_EUNSUPPORTED otherwise.
extern "C" int fr_linear_bwd_input_bnstats(const float* dY, const float* W, int64_t M, int32_t N, int32_t K, float* dX,
                                           const float* Yb, const float* xhat_b, int32_t act_b, void* bn_ws, size_t bn_ws_bytes,
                                           float p, uint64_t seed, uint64_t offset, const int64_t* used, void* stream_) {
    FR_CHECK_ARG(dY && W && dX && Yb && xhat_b && bn_ws && M >= 1 && N >= 1 && K >= 1 && act_ok(act_b) &&
                     bn_ws_bytes >= fr_bn_workspace_bytes(M, K), "fr_linear_bwd_input_bnstats: bad argument");
    FR_CHECK_ARG(p >= 0.f && p < 1.f && (p == 0.f || (used && offset % 4 == 0)), "fr_linear_bwd_input_bnstats: bad dropout arguments");
    prof_work(K_LINEAR_BWD_INPUT, 2.0 * (double)M * N * K);
    static const bool no_glds = getenv("FAIRREC_LINEAR_NO_GLDS") != nullptr || getenv("FAIRREC_LINEAR_SLOW") != nullptr;
    if (no_glds || N % 32 != 0 || K % 32 != 0 || (((uintptr_t)dY | (uintptr_t)W | (uintptr_t)dX) & 15) != 0 || M > 32768) {
        set_error("fr_linear_bwd_input_bnstats: shape not supported (N %% 32 == 0, K %% 32 == 0, 16-byte aligned operands, M <= 32768)");
        return FR_EUNSUPPORTED;
    }
    const GlBnb bnb{(float*)bn_ws, Yb, xhat_b, (int)act_b, p, (unsigned long long)seed, (unsigned long long)offset,
                    (const unsigned long long*)used};
    return glds_linear_bwd_input(dY, W, M, (int)N, (int)K, dX, (int)K, nullptr, 0, (hipStream_t)stream_, nullptr, 1.f, 0, &bnb);
}

// row splits of the weight gradient (slab bounded by 64 MiB)
static long long bwd_weight_splits(int64_t M, int32_t N, int32_t K) {
    // >= 128 rows per split: at B = 8192 that is 64 splits, i.e. 64 x (N/64) x (K/64) workgroups -- enough to fill 256 CUs
    // for the 128..512-wide layers of the reference's MLPs
    // ... 256 rows (twice the chunks per workgroup: a longer pipeline behind the same prologue) where that still leaves 512
    // workgroups of one 64 x 64 macro tile each: [8192, 512] -> 128 21.7 -> 18.0 us; narrower layers keep 128 rows
    long long rows = 128;
    if (M >= 4096 && (long long)((N + 63) / 64) * ((K + 63) / 64) * ((M + 255) / 256) >= 512) rows = 256;
    long long splits = (M + rows - 1) / rows;
    const long long cap = std::max<long long>(1, (64ll << 20) / ((long long)N * (K + 1) * (long long)sizeof(float)));
    return std::max<long long>(1, std::min<long long>(splits, std::min<long long>(cap, 1024)));
}

extern "C" size_t fr_linear_bwd_weight_workspace_bytes(int64_t M, int32_t N, int32_t K) {
    if (M < 1 || N < 1 || K < 1) return 0;
    return (size_t)bwd_weight_splits(M, N, K) * N * (K + 1) * sizeof(float);
}

extern "C" int fr_linear_bwd_weight(const float* dY, const float* Y, int32_t act, const float* x0, int32_t k0,
                                    const float* x1, int32_t k1, const uint8_t* mask, float scale, int64_t M, int32_t N,
                                    float* dW, float* db, void* ws, size_t ws_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FR_CHECK_ARG(dY && Y && x0 && dW && ws && M >= 1 && N >= 1 && k0 >= 1 && k1 >= 0 && (k1 == 0 || x1) && act_ok(act),
                 "fr_linear_bwd_weight: bad argument");
    const int K = k0 + k1;
    const long long splits = bwd_weight_splits(M, N, K);
    prof_work(K_LINEAR_BWD_WEIGHT, 2.0 * (double)M * N * K);
    const int rows_per_split = (int)(((M + splits - 1) / splits + TK - 1) / TK * TK);
    FR_CHECK_ARG(ws_bytes >= (size_t)splits * N * (K + 1) * sizeof(float), "fr_linear_bwd_weight: workspace too small");
    CatMat X{x0, x1, k0, k1};
    float* slab = (float*)ws;
    float* bslab = db ? slab + (size_t)splits * N * K : nullptr;
    static const bool no_glds = getenv("FAIRREC_LINEAR_NO_GLDS") != nullptr || getenv("FAIRREC_LINEAR_SLOW") != nullptr;
    if (!no_glds && !mask && act == ACT_NONE && N % 32 == 0 && K % 32 == 0 && k0 % 32 == 0 && ((uintptr_t)dY & 15) == 0 &&
        ((uintptr_t)x0 & 15) == 0 && (!x1 || ((uintptr_t)x1 & 15) == 0)) {
        int rc = glds_linear_bwd_weight(dY, GlMat{x0, x1, k0, k1, k0}, M, (int)N, K, (int)splits, rows_per_split, slab, bslab,
                                        stream);
        if (rc) return rc;
    } else {
        ProfScope prof(K_LINEAR_BWD_WEIGHT, stream);
        FR_LAUNCH(prof, linear_bwd_weight_kernel,
                  dim3((unsigned)((N + TM - 1) / TM), (unsigned)((K + TN - 1) / TN), (unsigned)splits), dim3(256), 0, stream,
                  dY, Y, (int)act, X, mask, scale, (int)M, (int)N, K, rows_per_split, slab, bslab);
    }
    FR_CHECK_LAUNCH();
    const long long n = (long long)N * K, tot = n + (db ? N : 0);
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((tot + 63) / 64)), dim3(256), 0, stream, (const float*)slab,
                       (const float*)bslab, (int)splits, n, db ? (int)N : 0, dW, db);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

// The weight gradients of several layers of one backward pass in two launches (all the products, then all the slab sums)
// instead of two per layer.  Jobs in the fast form only (N, K, k0 multiples of 32, no mask, dY already at the pre-activation);
// a job with dY == NULL only sums `n_parts` partial results [n_parts][N * K] that somebody else wrote (`parts`).
extern "C" size_t fr_linear_bwd_weight_multi_workspace_bytes(const fr_wgrad_job* jobs, int32_t n, int64_t M) {
    size_t tot = 0;
    for (int j = 0; jobs && j < n; ++j)
        if (jobs[j].dY) tot += align_up(fr_linear_bwd_weight_workspace_bytes(M, jobs[j].N, jobs[j].k0 + jobs[j].k1), 256);
    return tot;
}

extern "C" int fr_linear_bwd_weight_multi(const fr_wgrad_job* jobs, int32_t n, int64_t M, void* ws, size_t ws_bytes,
                                          void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FR_CHECK_ARG(jobs && n >= 1 && n <= FR_WGRAD_MAX && M >= 1, "fr_linear_bwd_weight_multi: bad argument (1..FR_WGRAD_MAX jobs)");
    FR_CHECK_ARG(ws_bytes >= fr_linear_bwd_weight_multi_workspace_bytes(jobs, n, M) && (ws || ws_bytes == 0),
                 "fr_linear_bwd_weight_multi: workspace too small");
    GlWJob q[FR_WGRAD_MAX];
    char* p = (char*)ws;
    for (int j = 0; j < n; ++j) {
        const fr_wgrad_job& f = jobs[j];
        const int K = f.k0 + f.k1;
        FR_CHECK_ARG(f.dW && f.N >= 1 && f.k0 >= 1 && f.k1 >= 0, "fr_linear_bwd_weight_multi: bad job");
        if (!f.dY) {
            FR_CHECK_ARG(f.parts && f.n_parts >= 1 && !f.db, "fr_linear_bwd_weight_multi: a sum-only job needs parts and no db");
            q[j] = GlWJob{nullptr, GlMat{}, f.N, K, f.n_parts, 0, const_cast<float*>(f.parts), nullptr, f.dW, nullptr};
            continue;
        }
        if (f.N % 32 || K % 32 || f.k0 % 32 || ((uintptr_t)f.dY & 15) || ((uintptr_t)f.x0 & 15) || (f.x1 && ((uintptr_t)f.x1 & 15)) ||
            !f.x0 || (f.k1 > 0 && !f.x1)) {
            set_error("fr_linear_bwd_weight_multi: job %d is not in the fast form (N, K, k0 multiples of 32, 16-byte aligned)", j);
            return FR_EUNSUPPORTED;
        }
        const long long splits = bwd_weight_splits(M, f.N, K);
        const int rows_per_split = (int)(((M + splits - 1) / splits + TK - 1) / TK * TK);
        prof_work(K_LINEAR_BWD_WEIGHT, 2.0 * (double)M * f.N * K);
        float* slab = (float*)p;
        p += align_up((size_t)splits * f.N * (K + 1) * sizeof(float), 256);
        q[j] = GlWJob{f.dY, GlMat{f.x0, f.x1, f.k0, f.k1, f.k0}, f.N, K, (int)splits, rows_per_split, slab,
                      f.db ? slab + (size_t)splits * f.N * K : nullptr, f.dW, f.db};
    }
    return glds_linear_bwd_weight_multi(q, n, M, stream);
}

// Both backward products of a layer with ONE output in one pass over X (plus the slab reduction): what
// fr_linear_bwd_weight + fr_linear_bwd_input compute for N == 1, k1 == 0, no mask.  dX may be NULL (first layer of a
// model whose input needs no gradient).  Returns FR_EUNSUPPORTED when the shape does not suit (K % 64, K <= 512, 16-byte
// aligned X and W): the caller then takes the two general calls.  ws as for fr_linear_bwd_weight.  relu_scale > 0: X is the
// previous layer's output dropped in place (relu(z) o keep, keep = 0 or relu_scale) and dX comes out as the gradient at z.
extern "C" int fr_linear_n1_bwd(const float* dY, const float* Y, int32_t act, const float* X, int32_t K, const float* W,
                                int64_t M, float relu_scale, float* dX, float* dW, float* db, void* ws, size_t ws_bytes,
                                void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FR_CHECK_ARG(dY && Y && X && W && dW && ws && M >= 1 && K >= 1 && act_ok(act), "fr_linear_n1_bwd: bad argument");
    if (!n1_ok(1, K, 0, nullptr, X, W) || (dX && ((uintptr_t)dX & 15) != 0)) {
        set_error("fr_linear_n1_bwd: shape not supported (K %% 64 == 0, K <= %d, 16-byte aligned operands)", 64 * N1_MAXQ);
        return FR_EUNSUPPORTED;
    }
    const long long splits = bwd_weight_splits(M, 1, K);
    prof_work(K_LINEAR_BWD_WEIGHT, (dX ? 4.0 : 2.0) * (double)M * K);
    const int rows_per_block = (int)((M + splits - 1) / splits);
    FR_CHECK_ARG(ws_bytes >= (size_t)splits * (K + 1) * sizeof(float), "fr_linear_n1_bwd: workspace too small");
    float* slab = (float*)ws;
    float* bslab = db ? slab + (size_t)splits * K : nullptr;
    const int blocks = (int)((M + rows_per_block - 1) / rows_per_block);
    {
        ProfScope prof(K_LINEAR_BWD_WEIGHT, stream);
        FR_LAUNCH(prof, linear_n1_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, dY, Y, (int)act, X, W, (int)M, (int)K,
                  rows_per_block, relu_scale, dX, slab, bslab);
    }
    FR_CHECK_LAUNCH();
    const long long tot = K + (db ? 1 : 0);
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((tot + 63) / 64)), dim3(256), 0, stream, (const float*)slab,
                       (const float*)bslab, blocks, (long long)K, db ? 1 : 0, dW, db);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

// out = dY o act'(Y): lets the two backward products of a layer without BatchNorm take their fast form (which does not
// differentiate the activation itself) on one shared pre-pass
extern "C" int fr_act_bwd(const float* dY, const float* Y, int32_t act, int64_t n, float* out, void* stream_) {
    FR_CHECK_ARG(dY && Y && out && n >= 1 && n % 4 == 0 && act_ok(act), "fr_act_bwd: bad argument (n must be a multiple of 4)");
    FR_CHECK_ARG((((uintptr_t)dY | (uintptr_t)Y | (uintptr_t)out) & 15) == 0, "fr_act_bwd: 16-byte alignment required");
    return launch_act_bwd(dY, Y, (int)act, 1.f, (long long)n, out, (hipStream_t)stream_);
}

// ... with a dropout of the activation's OUTPUT folded in: Y is the dropped output Yd = act(z) o keep, dY the gradient with
// respect to Yd, `scale` = 1/(1-p).  ReLU only: there Yd > 0 exactly where z > 0 and the element was kept, so
// dY o keep o act'(z) = dY o scale o [Yd > 0] and neither z nor the keep pattern is needed.
extern "C" int fr_act_bwd_dropped(const float* dY, const float* Yd, float scale, int64_t n, float* out, void* stream_) {
    FR_CHECK_ARG(dY && Yd && out && n >= 1 && n % 4 == 0 && scale > 0.f, "fr_act_bwd_dropped: bad argument (n must be a multiple of 4)");
    FR_CHECK_ARG((((uintptr_t)dY | (uintptr_t)Yd | (uintptr_t)out) & 15) == 0, "fr_act_bwd_dropped: 16-byte alignment required");
    return launch_act_bwd(dY, Yd, ACT_RELU, scale, (long long)n, out, (hipStream_t)stream_);
}

// ---- BatchNorm1d on batch statistics (training mode) --------------------------------------------------------------
// Replaces nn.BatchNorm1d inside MLPLayers(bn=True) (layers.py:66-67; PFCN filters / discriminators, which the
// reference never switches to eval mode -- SURVEY.md App. B-3).  The batch is cut into row chunks so that a
// [8192, 256] activation fills the chip: grid = (column blocks of 64) x (row chunks).  A "stats" launch leaves one
// partial per (chunk, column); a one-workgroup-per-64-columns "fold" launch combines the partials in chunk order (fixed
// order => reproducible) into the column statistics; the "apply" launch writes its chunk.  (Folding inside every apply
// workgroup, and 128-row chunks -- 0.5 wave per SIMD, eight dependent batches of loads per wave -- made the two apply
// kernels 24 and 22 us at [8192, 128], eight times their memory time.)  Forward partials are (mean, M2) pairs folded with Chan's formula, so the variance is the
// two-pass one (no E[x^2]-E[x]^2 cancellation).
namespace fr {

constexpr int BN_THREADS = 256;     // 4 waves: lane = column, waves stride over the rows of the chunk
constexpr int BN_WAVES = BN_THREADS / 64;

static inline int bn_chunk_rows(int64_t M) {       // >= 32 rows per chunk (8 per wave: two batches of loads), <= 1024 chunks
    long long rc = 32;
    while ((M + rc - 1) / rc > 1024) rc *= 2;
    return (int)rc;
}

__device__ __forceinline__ float col_reduce(float a, float (*red)[64], int wave, int lane) {
    __syncthreads();
    red[wave][lane] = a;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < BN_WAVES; ++w) s += red[w][lane];
    return s;
}

// part[(chunk * N + n) * 2 + {0,1}] = mean and sum of squared deviations of column n over the chunk's rows
__global__ __launch_bounds__(BN_THREADS) void bn_fwd_stats_kernel(const float* __restrict__ Z, int M, int N, int rc,
                                                                  float* __restrict__ part) {
    __shared__ float red[BN_WAVES][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane;
    const int m0 = blockIdx.y * rc, m1 = min(M, m0 + rc);
    const bool ok = n < N;
    float s = 0.f;
    if (ok) {
#pragma unroll 8
        for (int m = m0 + wave; m < m1; m += BN_WAVES) s += Z[(size_t)m * N + n];
    }
    const float mean = col_reduce(s, red, wave, lane) / (float)(m1 - m0);
    float q = 0.f;
    if (ok) {
#pragma unroll 8
        for (int m = m0 + wave; m < m1; m += BN_WAVES) {
            const float d = Z[(size_t)m * N + n] - mean;
            q = fmaf(d, d, q);
        }
    }
    const float m2 = col_reduce(q, red, wave, lane);
    if (ok && wave == 0) {
        part[((size_t)blockIdx.y * N + n) * 2] = mean;
        part[((size_t)blockIdx.y * N + n) * 2 + 1] = m2;
    }
}

// Y = act(gamma * (Z - mean) / sqrt(var + eps) + beta); xhat and invstd are kept for the backward;
// running_mean / running_var follow torch (momentum, unbiased variance).
// fin[2n], fin[2n+1] = mean and 1/sqrt(var + eps) of column n; running statistics and invstd_out updated here.
// Two passes over the chunk partials (count_c, mean_c, M2_c): mean = sum count_c mean_c / M, then
// M2 = sum [M2_c + count_c (mean_c - mean)^2] -- the exact decomposition of the two-pass variance, as plain sums (four
// waves take a quarter of the chunks each in ascending order, the quarters are added in order: a fixed tree).  The
// sequential Chan update it replaces was a chain of 2 divisions per chunk: 20 us for 256 chunks.
__global__ __launch_bounds__(BN_THREADS) void bn_fwd_fold_kernel(const float* __restrict__ part, int chunks, int M, int N, int rc,
                                                                 float eps, float momentum, float* __restrict__ rmean,
                                                                 float* __restrict__ rvar, float* __restrict__ fin,
                                                                 float* __restrict__ invstd_out, long long* __restrict__ nbt,
                                                                 int nbt_inc) {
    __shared__ float sh[BN_WAVES][64];
    // num_batches_tracked += passes of this layer (layers.py's BatchNorm1d bookkeeping) rides here: no launch of its own
    if (nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += nbt_inc;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane;
    const int q = (chunks + BN_WAVES - 1) / BN_WAVES, c0 = wave * q, c1 = min(chunks, c0 + q);
    const bool ok = n < N;
    // A quarter of up to 64 chunks is requested at once and kept in registers for both passes: ONE memory latency in this
    // launch, which is nothing but a latency chain (7.6 us with the two passes loading eight at a time; same sums, same order).
    constexpr int QMAX = 64;
    const int nn = ok ? n : N - 1;
    float s = 0.f, m2 = 0.f;
    if (q <= QMAX) {
        float2 pv[QMAX];
#pragma unroll
        for (int k = 0; k < QMAX; ++k)
            if (c0 + k < c1) pv[k] = reinterpret_cast<const float2*>(part)[(size_t)(c0 + k) * N + nn];
#pragma unroll
        for (int k = 0; k < QMAX; ++k)
            if (c0 + k < c1) s = fmaf((float)(min(M, (c0 + k + 1) * rc) - (c0 + k) * rc), pv[k].x, s);
        sh[wave][lane] = s;
        __syncthreads();
        const float mean_ = (((sh[0][lane] + sh[1][lane]) + sh[2][lane]) + sh[3][lane]) / (float)M;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < QMAX; ++k)
            if (c0 + k < c1) {
                const float nb = (float)(min(M, (c0 + k + 1) * rc) - (c0 + k) * rc);
                const float d = pv[k].x - mean_;
                m2 += fmaf(nb * d, d, pv[k].y);
            }
        s = mean_;
    } else {
#pragma unroll 8
        for (int c = c0; c < c1; ++c) {
            const float nb = (float)(min(M, (c + 1) * rc) - c * rc);
            s = fmaf(nb, part[((size_t)c * N + nn) * 2], s);
        }
        sh[wave][lane] = s;
        __syncthreads();
        const float mean_ = (((sh[0][lane] + sh[1][lane]) + sh[2][lane]) + sh[3][lane]) / (float)M;
        __syncthreads();
#pragma unroll 8
        for (int c = c0; c < c1; ++c) {
            const float nb = (float)(min(M, (c + 1) * rc) - c * rc);
            const float2 p = reinterpret_cast<const float2*>(part)[(size_t)c * N + nn];
            const float d = p.x - mean_;
            m2 += fmaf(nb * d, d, p.y);
        }
        s = mean_;
    }
    const float mean = s;
    sh[wave][lane] = m2;
    __syncthreads();
    if (wave != 0 || !ok) return;
    m2 = ((sh[0][lane] + sh[1][lane]) + sh[2][lane]) + sh[3][lane];
    const float var = m2 / (float)M;
    const float invstd = 1.f / sqrtf(var + eps);
    fin[2 * n] = mean;
    fin[2 * n + 1] = invstd;
    invstd_out[n] = invstd;
    if (rmean) {
        rmean[n] = bn_running(rmean[n], momentum, mean);
        rvar[n] = bn_running(rvar[n], momentum, M > 1 ? m2 / (float)(M - 1) : var);
    }
}

__global__ __launch_bounds__(BN_THREADS) void bn_fwd_apply_kernel(const float* __restrict__ Z, const float* __restrict__ fin,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, int M, int N, int rc,
                                                                  int act, float* __restrict__ Y, float* __restrict__ xhat) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane;
    if (n >= N) return;
    const float mean = fin[2 * n], invstd = fin[2 * n + 1];
    const float g = gamma[n], b = beta[n];
    const int m0 = blockIdx.y * rc, m1 = min(M, m0 + rc);
#pragma unroll 8
    for (int m = m0 + wave; m < m1; m += BN_WAVES) {
        const float xh = (Z[(size_t)m * N + n] - mean) * invstd;
        xhat[(size_t)m * N + n] = xh;
        Y[(size_t)m * N + n] = act_fwd(fmaf(g, xh, b), act);
    }
}

// The same pass with the dropout of the NEXT layer's input folded in: also writes Yd = Y o keep (csrc/dropout.hpp: the
// pattern of element i is that of group (off4 + i / 4), exactly what fr_dropout_apply would draw for Y at that offset).
// Four consecutive columns per thread (one Philox call per thread); N % 4 == 0.
__global__ __launch_bounds__(256) void bn_fwd_apply_drop_kernel(const float* __restrict__ Z, const float* __restrict__ fin,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, long long quads, int N4,
                                                                int act, float* __restrict__ Y, float* __restrict__ xhat,
                                                                float* __restrict__ Yd, unsigned thr, float scale,
                                                                unsigned long long seed, unsigned long long off4,
                                                                const unsigned long long* __restrict__ ctr_src,
                                                                unsigned long long* __restrict__ used_out,
                                                                unsigned long long* __restrict__ tick) {
    __shared__ unsigned long long ctr_s;
    const unsigned long long ctr = drop_counter_enter(ctr_src, used_out, tick, &ctr_s);
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= quads) return;
    const int c4 = (int)(q % N4);
    const float4 z = reinterpret_cast<const float4*>(Z)[q];
    const float4 f0 = reinterpret_cast<const float4*>(fin)[2 * c4], f1 = reinterpret_cast<const float4*>(fin)[2 * c4 + 1];
    const float4 g = reinterpret_cast<const float4*>(gamma)[c4], b = reinterpret_cast<const float4*>(beta)[c4];
    const float4 xh = make_float4((z.x - f0.x) * f0.y, (z.y - f0.z) * f0.w, (z.z - f1.x) * f1.y, (z.w - f1.z) * f1.w);
    const float4 y = make_float4(act_fwd(fmaf(g.x, xh.x, b.x), act), act_fwd(fmaf(g.y, xh.y, b.y), act),
                                 act_fwd(fmaf(g.z, xh.z, b.z), act), act_fwd(fmaf(g.w, xh.w, b.w), act));
    const float4 k = drop_keep4(seed, ctr, off4 + (unsigned long long)q, thr, scale);
    reinterpret_cast<float4*>(xhat)[q] = xh;
    reinterpret_cast<float4*>(Y)[q] = y;
    reinterpret_cast<float4*>(Yd)[q] = make_float4(y.x * k.x, y.y * k.y, y.z * k.z, y.w * k.w);
}

// part[(chunk * N + n) * 2 + {0,1}] = sum dA, sum dA * xhat over the chunk's rows,  dA = dY o act'(Y)
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_stats_kernel(const float* __restrict__ dY, const float* __restrict__ Y,
                                                                  int act, const float* __restrict__ xhat, int M, int N,
                                                                  int rc, float* __restrict__ part) {
    __shared__ float red[BN_WAVES][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane;
    const int m0 = blockIdx.y * rc, m1 = min(M, m0 + rc);
    const bool ok = n < N;
    float s1 = 0.f, s2 = 0.f;
    if (ok) {
#pragma unroll 8
        for (int m = m0 + wave; m < m1; m += BN_WAVES) {
            const size_t i = (size_t)m * N + n;
            bn_bwd_acc(dY[i], act_bwd(Y[i], act), xhat[i], s1, s2);
        }
    }
    const float t1 = col_reduce(s1, red, wave, lane);
    const float t2 = col_reduce(s2, red, wave, lane);
    if (ok && wave == 0) {
        part[((size_t)blockIdx.y * N + n) * 2] = t1;
        part[((size_t)blockIdx.y * N + n) * 2 + 1] = t2;
    }
}

// fin[2n], fin[2n+1] = sum dA, sum dA * xhat of column n = dbeta, dgamma (four waves sum a quarter of the chunks each in
// ascending order, wave 0 adds the quarters in order)
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_fold_kernel(const float* __restrict__ part, int chunks, int N,
                                                                 float* __restrict__ fin, float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta) {
    __shared__ float sh[BN_WAVES][2][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane;
    const int q = (chunks + BN_WAVES - 1) / BN_WAVES, c0 = wave * q, c1 = min(chunks, c0 + q);
    float sum_da = 0.f, sum_dax = 0.f;
    constexpr int QMAX = 64;         // as bn_fwd_fold_kernel: a quarter of up to 64 chunks in one round of requests
    const int nn = n < N ? n : N - 1;
    if (q <= QMAX) {
        float2 pv[QMAX];
#pragma unroll
        for (int k = 0; k < QMAX; ++k)
            if (c0 + k < c1) pv[k] = reinterpret_cast<const float2*>(part)[(size_t)(c0 + k) * N + nn];
#pragma unroll
        for (int k = 0; k < QMAX; ++k)
            if (c0 + k < c1) {
                sum_da += pv[k].x;
                sum_dax += pv[k].y;
            }
    } else {
#pragma unroll 8
        for (int c = c0; c < c1; ++c) {
            const float2 p = reinterpret_cast<const float2*>(part)[(size_t)c * N + nn];
            sum_da += p.x;
            sum_dax += p.y;
        }
    }
    sh[wave][0][lane] = sum_da;
    sh[wave][1][lane] = sum_dax;
    __syncthreads();
    if (wave != 0 || n >= N) return;
    for (int w = 1; w < BN_WAVES; ++w) {
        sum_da += sh[w][0][lane];
        sum_dax += sh[w][1][lane];
    }
    fin[2 * n] = sum_da;
    fin[2 * n + 1] = sum_dax;
    dgamma[n] = sum_dax;
    dbeta[n] = sum_da;
}

// dZ = invstd * gamma * (dA - mean(dA) - xhat * mean(dA * xhat))
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_apply_kernel(const float* __restrict__ dY, const float* __restrict__ Y,
                                                                  int act, const float* __restrict__ xhat,
                                                                  const float* __restrict__ invstd,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ fin, int M, int N, int rc,
                                                                  float* __restrict__ dZ) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane;
    if (n >= N) return;
    const float sum_da = fin[2 * n], sum_dax = fin[2 * n + 1];
    const float isg = __fmul_rn(invstd[n], gamma[n]);
    const float a1 = sum_da / (float)M, a2 = sum_dax / (float)M;
    const int m0 = blockIdx.y * rc, m1 = min(M, m0 + rc);
#pragma unroll 8
    for (int m = m0 + wave; m < m1; m += BN_WAVES) {
        const size_t i = (size_t)m * N + n;
        dZ[i] = bn_bwd_dz(dY[i], act_bwd(Y[i], act), xhat[i], a1, a2, isg);
    }
}

// ---- the fold INSIDE the apply launch ------------------------------------------------------------------------------------------
// A fold launch is 5-8 us of latency for 64 KB of partials per block of 64 columns, and the apply launch behind it cannot start
// before it ends.  Here every apply workgroup -- 64 columns x a band of rows -- folds its own columns first (the chunk partials of a
// quarter requested at once: one memory latency; the same chains as the fold kernels, so the same bits) and goes straight on to
// its rows; the band is tall enough (>= 128 rows at [8192, 256]) that the partials are re-read by ~64 workgroups per column
// block out of L2, 32 MB per launch.  The workgroups of the first band write what the fold kernel wrote (invstd, running
// statistics, batch counter; dgamma / dbeta).  One launch per BatchNorm operator less, forward and backward.
__device__ __forceinline__ float2 bn_fold_stats_inline(const float* __restrict__ part, int chunks, int M, int N, int rc, int n,
                                                       float (*sh)[64]) {      // (mean, M2) of column n, valid in every wave
    constexpr int QMAX = 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = (chunks + BN_WAVES - 1) / BN_WAVES, c0 = wave * q, c1 = min(chunks, c0 + q);
    const int nn = n < N ? n : N - 1;
    float s = 0.f, m2 = 0.f, mean;
    if (q <= QMAX) {
        float2 pv[QMAX];
#pragma unroll
        for (int k = 0; k < QMAX; ++k)
            if (c0 + k < c1) pv[k] = reinterpret_cast<const float2*>(part)[(size_t)(c0 + k) * N + nn];
#pragma unroll
        for (int k = 0; k < QMAX; ++k)
            if (c0 + k < c1) s = fmaf((float)(min(M, (c0 + k + 1) * rc) - (c0 + k) * rc), pv[k].x, s);
        sh[wave][lane] = s;
        __syncthreads();
        mean = (((sh[0][lane] + sh[1][lane]) + sh[2][lane]) + sh[3][lane]) / (float)M;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < QMAX; ++k)
            if (c0 + k < c1) {
                const float nb = (float)(min(M, (c0 + k + 1) * rc) - (c0 + k) * rc);
                const float d = pv[k].x - mean;
                m2 += fmaf(nb * d, d, pv[k].y);
            }
    } else {
#pragma unroll 8
        for (int c = c0; c < c1; ++c) s = fmaf((float)(min(M, (c + 1) * rc) - c * rc), part[((size_t)c * N + nn) * 2], s);
        sh[wave][lane] = s;
        __syncthreads();
        mean = (((sh[0][lane] + sh[1][lane]) + sh[2][lane]) + sh[3][lane]) / (float)M;
        __syncthreads();
#pragma unroll 8
        for (int c = c0; c < c1; ++c) {
            const float nb = (float)(min(M, (c + 1) * rc) - c * rc);
            const float2 p = reinterpret_cast<const float2*>(part)[(size_t)c * N + nn];
            const float d = p.x - mean;
            m2 += fmaf(nb * d, d, p.y);
        }
    }
    sh[wave][lane] = m2;
    __syncthreads();
    m2 = ((sh[0][lane] + sh[1][lane]) + sh[2][lane]) + sh[3][lane];
    __syncthreads();
    return make_float2(mean, m2);
}

template <bool DROP>
__global__ __launch_bounds__(BN_THREADS) void bn_fwd_apply_fold_kernel(
    const float* __restrict__ Z, const float* __restrict__ part, int chunks, int M, int N, int rc, float eps, float momentum,
    float* __restrict__ rmean, float* __restrict__ rvar, float* __restrict__ invstd_out, long long* __restrict__ nbt, int nbt_inc,
    const float* __restrict__ gamma, const float* __restrict__ beta, int act, float* __restrict__ Y, float* __restrict__ xhat,
    int band, float* __restrict__ Yd, unsigned thr, float scale, unsigned long long seed, unsigned long long off4,
    const unsigned long long* __restrict__ ctr_src, unsigned long long* __restrict__ used_out, unsigned long long* __restrict__ tick) {
    __shared__ float sh[BN_WAVES][64];
    __shared__ float4 cst[64];          // (mean, invstd, gamma, beta) of the block's columns
    __shared__ unsigned long long ctr_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane;
    unsigned long long ctr = 0;
    if (DROP) ctr = drop_counter_enter(ctr_src, used_out, tick, &ctr_s, blockIdx.x == 0 && blockIdx.y == 0, gridDim.x * gridDim.y);
    const float2 st = bn_fold_stats_inline(part, chunks, M, N, rc, n, sh);
    const float var = st.y / (float)M;
    const float invstd = 1.f / sqrtf(var + eps);
    if (wave == 0) {
        cst[lane] = n < N ? make_float4(st.x, invstd, gamma[n], beta[n]) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (blockIdx.y == 0 && n < N) {     // what bn_fwd_fold_kernel leaves behind, once per column
            invstd_out[n] = invstd;
            if (rmean) {
                rmean[n] = bn_running(rmean[n], momentum, st.x);
                rvar[n] = bn_running(rvar[n], momentum, M > 1 ? st.y / (float)(M - 1) : var);
            }
        }
        if (nbt && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) *nbt += nbt_inc;
    }
    __syncthreads();
    const int m0 = blockIdx.y * band, m1 = min(M, m0 + band);
    if (!DROP) {
        if (n >= N) return;
        const float4 c = cst[lane];
#pragma unroll 16
        for (int m = m0 + wave; m < m1; m += BN_WAVES) {
            const float xh = (Z[(size_t)m * N + n] - c.x) * c.y;
            xhat[(size_t)m * N + n] = xh;
            Y[(size_t)m * N + n] = act_fwd(fmaf(c.z, xh, c.w), act);
        }
    } else {
        // four neighbouring columns per thread (one Philox call, bn_fwd_apply_drop_kernel's groups: element offset / 4)
        const int N4 = N >> 2, c4l = threadIdx.x & 15, c4 = blockIdx.x * 16 + c4l;
        if (c4 >= N4) return;
        const float4 k0 = cst[4 * c4l], k1 = cst[4 * c4l + 1], k2 = cst[4 * c4l + 2], k3 = cst[4 * c4l + 3];
#pragma unroll 8
        for (int m = m0 + (threadIdx.x >> 4); m < m1; m += BN_THREADS / 16) {
            const long long qd = (long long)m * N4 + c4;
            const float4 z = reinterpret_cast<const float4*>(Z)[qd];
            const float4 xh = make_float4((z.x - k0.x) * k0.y, (z.y - k1.x) * k1.y, (z.z - k2.x) * k2.y, (z.w - k3.x) * k3.y);
            const float4 y = make_float4(act_fwd(fmaf(k0.z, xh.x, k0.w), act), act_fwd(fmaf(k1.z, xh.y, k1.w), act),
                                         act_fwd(fmaf(k2.z, xh.z, k2.w), act), act_fwd(fmaf(k3.z, xh.w, k3.w), act));
            const float4 k = drop_keep4(seed, ctr, off4 + (unsigned long long)qd, thr, scale);
            reinterpret_cast<float4*>(xhat)[qd] = xh;
            reinterpret_cast<float4*>(Y)[qd] = y;
            reinterpret_cast<float4*>(Yd)[qd] = make_float4(y.x * k.x, y.y * k.y, y.z * k.z, y.w * k.w);
        }
    }
}

__global__ __launch_bounds__(BN_THREADS) void bn_bwd_apply_fold_kernel(const float* __restrict__ dY, const float* __restrict__ Y,
                                                                       int act, const float* __restrict__ xhat,
                                                                       const float* __restrict__ invstd,
                                                                       const float* __restrict__ gamma,
                                                                       const float* __restrict__ part, int chunks, int M, int N,
                                                                       int band, float* __restrict__ dgamma,
                                                                       float* __restrict__ dbeta, float* __restrict__ dZ) {
    __shared__ float sh[BN_WAVES][2][64];
    __shared__ float2 tot[64];
    constexpr int QMAX = 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane, nn = n < N ? n : N - 1;
    const int q = (chunks + BN_WAVES - 1) / BN_WAVES, c0 = wave * q, c1 = min(chunks, c0 + q);
    float sum_da = 0.f, sum_dax = 0.f;
    if (q <= QMAX) {       // bn_bwd_fold_kernel's chains
        float2 pv[QMAX];
#pragma unroll
        for (int k = 0; k < QMAX; ++k)
            if (c0 + k < c1) pv[k] = reinterpret_cast<const float2*>(part)[(size_t)(c0 + k) * N + nn];
#pragma unroll
        for (int k = 0; k < QMAX; ++k)
            if (c0 + k < c1) {
                sum_da += pv[k].x;
                sum_dax += pv[k].y;
            }
    } else {
#pragma unroll 8
        for (int c = c0; c < c1; ++c) {
            const float2 p = reinterpret_cast<const float2*>(part)[(size_t)c * N + nn];
            sum_da += p.x;
            sum_dax += p.y;
        }
    }
    sh[wave][0][lane] = sum_da;
    sh[wave][1][lane] = sum_dax;
    __syncthreads();
    if (wave == 0) {
        for (int w = 1; w < BN_WAVES; ++w) {
            sum_da += sh[w][0][lane];
            sum_dax += sh[w][1][lane];
        }
        tot[lane] = make_float2(sum_da, sum_dax);
        if (blockIdx.y == 0 && n < N) {
            dgamma[n] = sum_dax;
            dbeta[n] = sum_da;
        }
    }
    __syncthreads();
    if (n >= N) return;
    const float2 t = tot[lane];
    const float isg = __fmul_rn(invstd[n], gamma[n]);
    const float a1 = t.x / (float)M, a2 = t.y / (float)M;
    const int m0 = blockIdx.y * band, m1 = min(M, m0 + band);
#pragma unroll 16
    for (int m = m0 + wave; m < m1; m += BN_WAVES) {
        const size_t i = (size_t)m * N + n;
        dZ[i] = bn_bwd_dz(dY[i], act_bwd(Y[i], act), xhat[i], a1, a2, isg);
    }
}

// rows of an apply workgroup's band: ~256 workgroups per launch, never less than a statistics chunk
static inline int bn_apply_band(int64_t M, int32_t N, int rc) {
    const long long cols = (N + 63) / 64;
    static const long long target = getenv("FAIRREC_BN_APPLY_WGS") ? atoll(getenv("FAIRREC_BN_APPLY_WGS")) : 256;
    long long bands = std::max<long long>(1, target / cols);
    long long band = (M + bands - 1) / bands;
    band = std::max<long long>(band, rc);
    return (int)((band + 3) / 4 * 4);
}

}  // namespace fr

extern "C" size_t fr_bn_workspace_bytes(int64_t M, int32_t N) {
    if (M < 1 || N < 1) return 0;
    const int rc = bn_chunk_rows(M);
    return ((size_t)((M + rc - 1) / rc) + 1) * N * 2 * sizeof(float);   // per-chunk partials + the folded column statistics
}

static int bn_fwd_impl(const float* Z, const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                       float* running_var, int64_t M, int32_t N, int32_t act, float* Y, float* xhat, float* invstd, void* ws,
                       size_t ws_bytes, hipStream_t stream, float* Yd, float p, uint64_t seed, uint64_t offset,
                       const int64_t* counter, int64_t* used_out, int64_t* tick_state, bool have_stats = false,
                       int64_t* nbt = nullptr, int32_t nbt_inc = 0);

extern "C" int fr_bn_fwd(const float* Z, const float* gamma, const float* beta, float eps, float momentum,
                         float* running_mean, float* running_var, int64_t M, int32_t N, int32_t act, float* Y,
                         float* xhat, float* invstd, void* ws, size_t ws_bytes, void* stream_) {
    return bn_fwd_impl(Z, gamma, beta, eps, momentum, running_mean, running_var, M, N, act, Y, xhat, invstd, ws, ws_bytes,
                       (hipStream_t)stream_, nullptr, 0.f, 0, 0, nullptr, nullptr, nullptr);
}

// fr_bn_fwd that also writes Yd = dropout(Y), the next layer's input, in its last launch: what fr_dropout_apply(Y, ...,
// out = Yd) would give with the same (p, seed, offset, counter, used_out, tick_state); N % 4 == 0.
extern "C" int fr_bn_fwd_drop(const float* Z, const float* gamma, const float* beta, float eps, float momentum,
                              float* running_mean, float* running_var, int64_t M, int32_t N, int32_t act, float* Y,
                              float* xhat, float* invstd, void* ws, size_t ws_bytes, float* Yd, float p, uint64_t seed,
                              uint64_t offset, const int64_t* counter, int64_t* used_out, int64_t* tick_state,
                              void* stream_) {
    FR_CHECK_ARG(Yd && counter && N % 4 == 0 && p >= 0.f && p < 1.f && offset % 4 == 0 &&
                     (((uintptr_t)Z | (uintptr_t)Y | (uintptr_t)Yd | (uintptr_t)xhat | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0,
                 "fr_bn_fwd_drop: bad argument (N % 4 == 0, 16-byte aligned tensors)");
    return bn_fwd_impl(Z, gamma, beta, eps, momentum, running_mean, running_var, M, N, act, Y, xhat, invstd, ws, ws_bytes,
                       (hipStream_t)stream_, Yd, p, seed, offset, counter, used_out, tick_state);
}

static int bn_fwd_impl(const float* Z, const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                       float* running_var, int64_t M, int32_t N, int32_t act, float* Y, float* xhat, float* invstd, void* ws,
                       size_t ws_bytes, hipStream_t stream, float* Yd, float p, uint64_t seed, uint64_t offset,
                       const int64_t* counter, int64_t* used_out, int64_t* tick_state, bool have_stats, int64_t* nbt,
                       int32_t nbt_inc) {
    FR_CHECK_ARG(Z && gamma && beta && Y && xhat && invstd && ws && M >= 1 && N >= 1 && act_ok(act) &&
                     ws_bytes >= fr_bn_workspace_bytes(M, N), "fr_bn_fwd: bad argument");
    const int rc = bn_chunk_rows(M);
    const dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + rc - 1) / rc));
    ProfScope prof(K_BN_FWD, stream);
    if (!have_stats) {      // (else: the producing product's epilogue wrote the partials, fr_linear_fwd_bnstats)
        FR_LAUNCH(prof, bn_fwd_stats_kernel, grid, dim3(BN_THREADS), 0, stream, Z, (int)M, (int)N, rc, (float*)ws);
        FR_CHECK_LAUNCH();
    }
    float* fin = (float*)ws + (size_t)grid.y * N * 2;
    if (getenv("FAIRREC_BN_FOLD_SEPARATE") == nullptr) {      // the fold inside the apply launch (one launch less)
        const int band = bn_apply_band(M, N, rc);
        const dim3 g2(grid.x, (unsigned)((M + band - 1) / band));
        if (Yd)
            FR_LAUNCH(prof, bn_fwd_apply_fold_kernel<true>, g2, dim3(BN_THREADS), 0, stream, Z, (const float*)ws, (int)grid.y, (int)M,
                      (int)N, rc, eps, momentum, running_mean, running_var, invstd, (long long*)nbt, (int)nbt_inc, gamma, beta, (int)act,
                      Y, xhat, band, Yd, drop_threshold(p), 1.f / (1.f - p), (unsigned long long)seed, (unsigned long long)(offset / 4),
                      (const unsigned long long*)counter, (unsigned long long*)used_out, (unsigned long long*)tick_state);
        else
            FR_LAUNCH(prof, bn_fwd_apply_fold_kernel<false>, g2, dim3(BN_THREADS), 0, stream, Z, (const float*)ws, (int)grid.y, (int)M,
                      (int)N, rc, eps, momentum, running_mean, running_var, invstd, (long long*)nbt, (int)nbt_inc, gamma, beta, (int)act,
                      Y, xhat, band, (float*)nullptr, 0u, 1.f, 0ull, 0ull, (const unsigned long long*)nullptr,
                      (unsigned long long*)nullptr, (unsigned long long*)nullptr);
        FR_CHECK_LAUNCH();
        return FR_OK;
    }
    FR_LAUNCH(prof, bn_fwd_fold_kernel, dim3(grid.x), dim3(BN_THREADS), 0, stream, (const float*)ws, (int)grid.y, (int)M, (int)N, rc, eps,
              momentum, running_mean, running_var, fin, invstd, (long long*)nbt, (int)nbt_inc);
    FR_CHECK_LAUNCH();
    if (Yd) {
        const long long quads = (long long)M * N / 4;
        FR_LAUNCH(prof, bn_fwd_apply_drop_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, stream, Z, (const float*)fin,
                  gamma, beta, quads, (int)(N / 4), (int)act, Y, xhat, Yd, drop_threshold(p), 1.f / (1.f - p),
                  (unsigned long long)seed, (unsigned long long)(offset / 4), (const unsigned long long*)counter,
                  (unsigned long long*)used_out, (unsigned long long*)tick_state);
    } else {
        FR_LAUNCH(prof, bn_fwd_apply_kernel, grid, dim3(BN_THREADS), 0, stream, Z, (const float*)fin, gamma, beta, (int)M, (int)N, rc,
                  (int)act, Y, xhat);
    }
    FR_CHECK_LAUNCH();
    return FR_OK;
}

// Z = X W^T + b for a layer with BatchNorm behind it, the BatchNorm's per-chunk statistics written by the product's own
// epilogue into `bn_ws` (fr_bn_workspace_bytes(M, N): what fr_bn_fwd's first launch would leave there): follow with
// fr_bn_fwd_ex(..., have_stats = 1).  Fast form only (K, k0 multiples of 32, no mask, 32-row chunks, i.e. M <= 32768, the
// macro-tile kernels): FR_EUNSUPPORTED otherwise -- the caller then takes fr_linear_fwd + fr_bn_fwd.
extern "C" int fr_linear_fwd_bnstats(const float* x0, int32_t k0, const float* x1, int32_t k1, const float* W, const float* bias,
                                     int64_t M, int32_t N, float* Z, void* bn_ws, size_t bn_ws_bytes, void* stream_) {
    FR_CHECK_ARG(x0 && W && Z && bn_ws && M >= 1 && N >= 1 && k0 >= 1 && k1 >= 0 && (k1 == 0 || x1), "fr_linear_fwd_bnstats: bad argument");
    FR_CHECK_ARG(bn_ws_bytes >= fr_bn_workspace_bytes(M, N), "fr_linear_fwd_bnstats: BatchNorm workspace too small");
    const int K = k0 + k1;
    static const bool off = getenv("FAIRREC_LINEAR_SLOW") != nullptr || getenv("FAIRREC_LINEAR_NO_GLDS") != nullptr ||
                            getenv("FAIRREC_BN_STATS_LAUNCH") != nullptr;
    const bool aligned = ((uintptr_t)x0 & 15) == 0 && ((uintptr_t)W & 15) == 0 && (!x1 || ((uintptr_t)x1 & 15) == 0);
    if (off || !aligned || K % 32 != 0 || k0 % 32 != 0 || N < 8 || bn_chunk_rows(M) != 32 || !glds_shared_form()) {
        set_error("fr_linear_fwd_bnstats: not in the fast form");
        return FR_EUNSUPPORTED;
    }
    prof_work(K_LINEAR_FWD, 2.0 * (double)M * N * K);
    return glds_linear_fwd(GlMat{x0, x1, k0, k1, k0}, W, bias, M, (int)N, K, ACT_NONE, Z, (hipStream_t)stream_, (float*)bn_ws);
}

// fr_bn_fwd / fr_bn_fwd_drop in one entry (Yd == NULL: no dropout of the output), `have_stats` != 0: the statistics launch is
// skipped -- fr_linear_fwd_bnstats left the per-chunk partials in `ws`.
extern "C" int fr_bn_fwd_ex(const float* Z, const float* gamma, const float* beta, float eps, float momentum,
                            float* running_mean, float* running_var, int64_t M, int32_t N, int32_t act, float* Y, float* xhat,
                            float* invstd, void* ws, size_t ws_bytes, int32_t have_stats, float* Yd, float p, uint64_t seed,
                            uint64_t offset, const int64_t* counter, int64_t* used_out, int64_t* tick_state,
                            int64_t* num_batches_tracked, int32_t passes, void* stream_) {
    FR_CHECK_ARG(!Yd || (counter && N % 4 == 0 && p >= 0.f && p < 1.f && offset % 4 == 0 &&
                         (((uintptr_t)Z | (uintptr_t)Y | (uintptr_t)Yd | (uintptr_t)xhat | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0),
                 "fr_bn_fwd_ex: bad argument (N % 4 == 0, 16-byte aligned tensors)");
    return bn_fwd_impl(Z, gamma, beta, eps, momentum, running_mean, running_var, M, N, act, Y, xhat, invstd, ws, ws_bytes,
                       (hipStream_t)stream_, Yd, p, seed, offset, counter, used_out, tick_state, have_stats != 0,
                       num_batches_tracked, passes);
}

extern "C" int fr_bn_bwd(const float* dY, const float* Y, int32_t act, const float* xhat, const float* invstd,
                         const float* gamma, int64_t M, int32_t N, float* dZ, float* dgamma, float* dbeta, void* ws,
                         size_t ws_bytes, void* stream_) {
    return fr_bn_bwd_ex(dY, Y, act, xhat, invstd, gamma, M, N, dZ, dgamma, dbeta, ws, ws_bytes, 0, stream_);
}

// have_stats != 0: ws already holds the per-chunk sums (fr_linear_bwd_input_bnstats left them): the apply launch alone
extern "C" int fr_bn_bwd_ex(const float* dY, const float* Y, int32_t act, const float* xhat, const float* invstd,
                            const float* gamma, int64_t M, int32_t N, float* dZ, float* dgamma, float* dbeta, void* ws,
                            size_t ws_bytes, int32_t have_stats, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FR_CHECK_ARG(dY && Y && xhat && invstd && gamma && dZ && dgamma && dbeta && ws && M >= 1 && N >= 1 && act_ok(act) &&
                     ws_bytes >= fr_bn_workspace_bytes(M, N), "fr_bn_bwd: bad argument");
    const int rc = bn_chunk_rows(M);
    FR_CHECK_ARG(!have_stats || (rc == 32 && getenv("FAIRREC_BN_FOLD_SEPARATE") == nullptr),
                 "fr_bn_bwd_ex: statistics from a product's epilogue are per 32-row tile (M <= 32768, fold inside the apply launch)");
    const dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + rc - 1) / rc));
    ProfScope prof(K_BN_BWD, stream);
    if (!have_stats) {
        FR_LAUNCH(prof, bn_bwd_stats_kernel, grid, dim3(BN_THREADS), 0, stream, dY, Y, (int)act, xhat, (int)M, (int)N, rc,
                  (float*)ws);
        FR_CHECK_LAUNCH();
    }
    if (getenv("FAIRREC_BN_FOLD_SEPARATE") == nullptr) {
        const int band = bn_apply_band(M, N, rc);
        FR_LAUNCH(prof, bn_bwd_apply_fold_kernel, dim3(grid.x, (unsigned)((M + band - 1) / band)), dim3(BN_THREADS), 0, stream, dY, Y,
                  (int)act, xhat, invstd, gamma, (const float*)ws, (int)grid.y, (int)M, (int)N, band, dgamma, dbeta, dZ);
        FR_CHECK_LAUNCH();
        return FR_OK;
    }
    float* fin = (float*)ws + (size_t)grid.y * N * 2;
    FR_LAUNCH(prof, bn_bwd_fold_kernel, dim3(grid.x), dim3(BN_THREADS), 0, stream, (const float*)ws, (int)grid.y, (int)N, fin, dgamma,
              dbeta);
    FR_CHECK_LAUNCH();
    FR_LAUNCH(prof, bn_bwd_apply_kernel, grid, dim3(BN_THREADS), 0, stream, dY, Y, (int)act, xhat, invstd, gamma,
              (const float*)fin, (int)M, (int)N, rc, dZ);
    FR_CHECK_LAUNCH();
    return FR_OK;
}
