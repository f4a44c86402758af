// An MLP with BatchNorm behind every layer (layers.py:56-85 with bn=True: Dropout -> Linear -> BatchNorm1d -> activation, the
// last layer included; pfcn_biasedmf.py:113-142: PFCN's filters and discriminators) as ONE launch per layer and direction.
//
// The layered form (mlp.hip) spends three launches per layer forward -- product (+ chunk statistics), fold, normalise (+ the
// next layer's dropout) -- and four backward -- statistics, fold, apply, input-gradient product -- each a few us of work
// behind a launch boundary: 86 BatchNorm launches of 5-7 us were 40 % of a PFCN step (profiles/README.md, round 4).  A batch
// statistic needs every row, so a layer cannot be normalised by the launch that produces it; but the launch that CONSUMES it
// can normalise on the way in.  Here:
//
//   bnl_fwd_kernel   Z_l = A_l W_l^T + b_l for a tile of 32 rows and ALL columns, where the tile of
//                    A_l = dropout(act(gamma (Z_{l-1} - mean) invstd + beta)) is formed from Z_{l-1} while it is loaded into LDS
//                    (the MLP's input for l = 0); the (mean, M2) partials of the tile's 32 rows come out of the accumulators
//                    (as the layered product's epilogue forms them), and the workgroup that arrives last folds all tiles'
//                    partials into the layer's (mean, invstd), running statistics and batch counter -- bn_fwd_fold_kernel's
//                    arithmetic, so the statistics have the layered form's bits.
//   bnl_out_kernel   the MLP's output act(BN(Z_last)) (elementwise).
//   bnl_top_kernel   backward, top layer: column sums of dY s and dY s xhat per tile (s = act'(y); y, xhat re-formed from
//                    Z_last), folded by the last workgroup (= dbeta, dgamma).
//   bnl_bwd_kernel   dZ_l = invstd gamma (dy s - mean(dy s) - xhat mean(dy s xhat)) formed on the way into LDS (dy: the gradient
//                    at the layer's output), dA_l = dZ_l W_l for the tile and all input columns, back through the dropout of
//                    A_l -- that is the dy of the layer below -- and that layer's column sums from it, fold.
// Every expression and every summation order is the layered form's (mlp_bn_math.hpp; the products add their reduction in the
// layered kernels' parts; partial sums per 32-row chunk in bn_bwd_stats_kernel's four chains): where both forms apply they
// agree bit for bit, forward and backward.
//
// What a forward pass leaves in memory per layer: Z_l, the folded statistics and (when the weights train) A_l, for
// dW_l = dZ_l^T A_l, which goes through fr_linear_bwd_weight_multi with all the other layers' in one launch.  No xhat, no
// normalised output, no dropped copy: they are re-formed where they are used.  Dropout patterns are csrc/dropout.hpp's
// (seed, call counter, element offset), the offsets the layered form hands out, so the two forms drop the same elements.
//
// Shapes: input widths multiples of 32, <= 256; output widths <= 256, any (a discriminator's last layer has 1 or a handful
// of outputs: the tile is zero-padded in LDS).  Anything else stays on the layered form (FR_EUNSUPPORTED).
#include "common.hpp"
#include "kernels.hpp"
#include "dropout.hpp"
#include "mlp_act.hpp"
#include "mlp_glds.hpp"
#include "mlp_bn_math.hpp"

namespace fr {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;

constexpr int BNL_THREADS = 256, BNL_WAVES = 4, BNL_MAXW = 256;
constexpr int BNL_NBUF = 3;      // a wave's ring of W blocks: two chunks requested ahead (one ahead left 0.9 us of every 1.75 us chunk waiting)

// -DBNL_TRACE=1 (a diagnostic build: make VARIANT=bnltrace EXTRA=-DBNL_TRACE=1): thread 0 of every workgroup stamps the phases of
// its launch with the 100 MHz wall clock; scratch/bnl_trace.py reads them back through fr_bnl_trace_read.
#ifndef BNL_TRACE
#define BNL_TRACE 0
#endif
#if BNL_TRACE
__device__ unsigned long long bnl_trace[4096 * 8];
#define BNL_STAMP(k)                                                                        \
    do {                                                                                    \
        if (threadIdx.x == 0 && blockIdx.x < 4096) bnl_trace[blockIdx.x * 8 + (k)] = wall_clock64(); \
    } while (0)
#else
#define BNL_STAMP(k) \
    do {             \
    } while (0)
#endif

struct BnlSrc {                // device form of fr_bn_src
    const float* Z;
    const float* fin;          // [width][2] mean, invstd; nullptr: Z holds the values themselves
    const float* gamma;
    const float* beta;
    int act;
    unsigned thr;              // dropout threshold (0: none)
    float scale;
    unsigned long long seed, off4;
};

__device__ __forceinline__ float ld_dev(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_dev(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// xhat and y = act(gamma xhat + beta) of four neighbouring columns 4 kq .. 4 kq + 3 (bn_fwd_apply_drop_kernel's expressions)
__device__ __forceinline__ void bnl_norm4(const BnlSrc& s, const float4 z, const int kq, float4& xh, float4& y) {
    if (!s.fin) {
        xh = make_float4(0.f, 0.f, 0.f, 0.f);
        y = z;
        return;
    }
    const float4 f0 = reinterpret_cast<const float4*>(s.fin)[2 * kq], f1 = reinterpret_cast<const float4*>(s.fin)[2 * kq + 1];
    const float4 g = reinterpret_cast<const float4*>(s.gamma)[kq], b = reinterpret_cast<const float4*>(s.beta)[kq];
    xh = make_float4((z.x - f0.x) * f0.y, (z.y - f0.z) * f0.w, (z.z - f1.x) * f1.y, (z.w - f1.z) * f1.w);
    y = make_float4(act_fwd(fmaf(g.x, xh.x, b.x), s.act), act_fwd(fmaf(g.y, xh.y, b.y), s.act),
                    act_fwd(fmaf(g.z, xh.z, b.z), s.act), act_fwd(fmaf(g.w, xh.w, b.w), s.act));
}
__device__ __forceinline__ void bnl_norm1(const BnlSrc& s, const float z, const int n, float& xh, float& y) {
    if (!s.fin) {
        xh = 0.f;
        y = z;
        return;
    }
    xh = (z - s.fin[2 * n]) * s.fin[2 * n + 1];
    y = act_fwd(fmaf(s.gamma[n], xh, s.beta[n]), s.act);
}

__device__ __forceinline__ float2 ld_dev2(const float* p) {     // an 8-byte aligned pair, device scope
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32)));
}

// Per-column constants of a source, parked in LDS once per workgroup: colc[n] = (mean, invstd, gamma, beta).  The loops that
// form a tile then take them from LDS instead of four more global round trips per 16-byte unit (one wave per SIMD hides none).
__device__ __forceinline__ void bnl_park_columns(const BnlSrc& s, int width, float4* colc) {
    if (!s.fin) return;
    for (int n = threadIdx.x; n < width; n += BNL_THREADS)
        colc[n] = make_float4(s.fin[2 * n], s.fin[2 * n + 1], s.gamma[n], s.beta[n]);
}
__device__ __forceinline__ void bnl_norm4c(const BnlSrc& s, const float4* colc, const float4 z, const int kq, float4& xh, float4& y) {
    if (!s.fin) {
        xh = make_float4(0.f, 0.f, 0.f, 0.f);
        y = z;
        return;
    }
    const float4 c0 = colc[4 * kq], c1 = colc[4 * kq + 1], c2 = colc[4 * kq + 2], c3 = colc[4 * kq + 3];
    xh = make_float4((z.x - c0.x) * c0.y, (z.y - c1.x) * c1.y, (z.z - c2.x) * c2.y, (z.w - c3.x) * c3.y);
    y = make_float4(act_fwd(fmaf(c0.z, xh.x, c0.w), s.act), act_fwd(fmaf(c1.z, xh.y, c1.w), s.act),
                    act_fwd(fmaf(c2.z, xh.z, c2.w), s.act), act_fwd(fmaf(c3.z, xh.w, c3.w), s.act));
}
constexpr int BNL_UMAX = BNL_MAXW / 32;      // 16-byte units of a 32-row tile per thread: width / 32

// wait until this wave's stage of the current chunk has landed: `left` newer chunks exist, at most NBUF - 1 of them were requested
// (4 TPW LDS-DMA instructions each)
template <int TPW>
__device__ __forceinline__ void bnl_wait_newer(int left) {
    static_assert(BNL_NBUF == 3, "the wait counts below are written for a ring of three");
    if (left >= 2) {
        if (TPW == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else if (left == 1) {
        if (TPW == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

// position of the 16-byte unit (row, slot) inside a 32 x 32 block of the LDS image (in units of float4)
__device__ __forceinline__ int img_unit(int row, int slot) { return row * 8 + (slot ^ ((row >> 1) & 7)); }

// Arrival of a workgroup whose partials (device-scope stores) are on their way.  The fold of the partials is one latency chain
// per (column, quarter of the tiles) -- done by ONE workgroup for all columns it was 20-28 us behind an 8 us product -- so the
// LAST F workgroups to arrive share it, a block of 64 columns each (F = min(4, blocks of 64 columns, workgroups)): each of them
// waits until everyone has arrived (the others it waits for are running or about to be dispatched: F <= 4 waiters never keep
// them from a slot; the wait is bounded anyway), folds its blocks, and the last one out puts the two counters back to zero.
// Returns the folder index, or -1 for a workgroup that is done.  tk[0]: arrivals, tk[1]: folders that are through.
__device__ __forceinline__ int bnl_folders(int N) {
    const int nb = (N + 63) >> 6;
    const int f = nb < 4 ? nb : 4;
    return f < (int)gridDim.x ? f : (int)gridDim.x;
}
__device__ __forceinline__ int bnl_arrive(unsigned* tk, int F, int* sh_f) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned T = gridDim.x;
        const unsigned t = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int rank = (int)(T - 1u - t);
        const int f = rank < F ? rank : -1;
        BNL_STAMP(6);
        if (f >= 0) {
            for (unsigned spin = 0; spin < (1u << 24) && __hip_atomic_load(tk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < T; ++spin)
                __builtin_amdgcn_s_sleep(4);
        }
        *sh_f = f;
    }
    __syncthreads();
    return *sh_f;
}
__device__ __forceinline__ void bnl_leave(unsigned* tk, int F) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned d = __hip_atomic_fetch_add(tk + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)d == F - 1) {
            __hip_atomic_store(tk + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// bn_fwd_fold_kernel (mlp.hip) for the blocks of 64 columns f, f + F, ...: part[(c * N + n) * 2] = (mean, M2) of the 32 rows of
// tile c.  The same chains (four waves take a quarter of the tiles each, in ascending order; the quarters are added in order);
// a quarter of up to 64 tiles is REQUESTED at once (64 pairs in registers): one memory latency instead of one per batch of the
// two passes -- this runs at the tail of a launch with nothing to overlap it.
constexpr int BNL_QMAX = 64;
__device__ __forceinline__ void bnl_fold_stats(const float* part, int chunks, int M, int N, float eps, float momentum, float* rmean, float* rvar,
                               float* fin, float (*sh)[64], int f, int F) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = (chunks + BNL_WAVES - 1) / BNL_WAVES, c0 = wave * q, c1 = min(chunks, c0 + q);
    for (int nb = f * 64; nb < N; nb += F * 64) {
        const int n = nb + lane;
        const bool ok = n < N;
        const int nn = ok ? n : N - 1;
        float s = 0.f, m2 = 0.f;
        if (q <= BNL_QMAX) {
            float2 pv[BNL_QMAX];
#pragma unroll
            for (int k = 0; k < BNL_QMAX; ++k)
                if (c0 + k < c1) pv[k] = ld_dev2(part + ((size_t)(c0 + k) * N + nn) * 2);
#pragma unroll
            for (int k = 0; k < BNL_QMAX; ++k)
                if (c0 + k < c1) s = fmaf((float)(min(M, (c0 + k + 1) * 32) - (c0 + k) * 32), pv[k].x, s);
            sh[wave][lane] = s;
            __syncthreads();
            const float mean = (((sh[0][lane] + sh[1][lane]) + sh[2][lane]) + sh[3][lane]) / (float)M;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < BNL_QMAX; ++k)
                if (c0 + k < c1) {
                    const float cnt = (float)(min(M, (c0 + k + 1) * 32) - (c0 + k) * 32);
                    const float d = pv[k].x - mean;
                    m2 += fmaf(cnt * d, d, pv[k].y);
                }
            sh[wave][lane] = m2;
            __syncthreads();
            if (wave == 0 && ok) {
                m2 = ((sh[0][lane] + sh[1][lane]) + sh[2][lane]) + sh[3][lane];
                const float var = m2 / (float)M;
                const float invstd = 1.f / sqrtf(var + eps);
                fin[2 * n] = mean;
                fin[2 * n + 1] = invstd;
                if (rmean) {
                    rmean[n] = bn_running(rmean[n], momentum, mean);
                    rvar[n] = bn_running(rvar[n], momentum, M > 1 ? m2 / (float)(M - 1) : var);
                }
            }
            __syncthreads();
            continue;
        }
#pragma unroll 16
        for (int c = c0; c < c1; ++c) s = fmaf((float)(min(M, (c + 1) * 32) - c * 32), ld_dev(part + ((size_t)c * N + nn) * 2), s);
        sh[wave][lane] = s;
        __syncthreads();
        const float mean = (((sh[0][lane] + sh[1][lane]) + sh[2][lane]) + sh[3][lane]) / (float)M;
        __syncthreads();
#pragma unroll 16
        for (int c = c0; c < c1; ++c) {
            const float cnt = (float)(min(M, (c + 1) * 32) - c * 32);
            const float2 p = ld_dev2(part + ((size_t)c * N + nn) * 2);
            const float d = p.x - mean;
            m2 += fmaf(cnt * d, d, p.y);
        }
        sh[wave][lane] = m2;
        __syncthreads();
        if (wave == 0 && ok) {
            m2 = ((sh[0][lane] + sh[1][lane]) + sh[2][lane]) + sh[3][lane];
            const float var = m2 / (float)M;
            const float invstd = 1.f / sqrtf(var + eps);
            fin[2 * n] = mean;
            fin[2 * n + 1] = invstd;
            if (rmean) {
                rmean[n] = bn_running(rmean[n], momentum, mean);
                rvar[n] = bn_running(rvar[n], momentum, M > 1 ? m2 / (float)(M - 1) : var);
            }
        }
        __syncthreads();
    }
}

// bn_bwd_fold_kernel for the blocks of 64 columns f, f + F, ...: part[(c * N + n) * 2] = (sum dy s, sum dy s xhat) over the rows of tile c
__device__ __forceinline__ void bnl_fold_sums(const float* part, int chunks, int N, float* sums, float* dgamma, float* dbeta, float (*sh)[64], int f,
                              int F) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = (chunks + BNL_WAVES - 1) / BNL_WAVES, c0 = wave * q, c1 = min(chunks, c0 + q);
    for (int nb = f * 64; nb < N; nb += F * 64) {
        const int n = nb + lane;
        const int nn = n < N ? n : N - 1;
        float s1 = 0.f, s2 = 0.f;
        if (q <= BNL_QMAX) {
            float2 pv[BNL_QMAX];
#pragma unroll
            for (int k = 0; k < BNL_QMAX; ++k)
                if (c0 + k < c1) pv[k] = ld_dev2(part + ((size_t)(c0 + k) * N + nn) * 2);
#pragma unroll
            for (int k = 0; k < BNL_QMAX; ++k)
                if (c0 + k < c1) {
                    s1 += pv[k].x;
                    s2 += pv[k].y;
                }
        } else {
#pragma unroll 16
            for (int c = c0; c < c1; ++c) {
                const float2 p = ld_dev2(part + ((size_t)c * N + nn) * 2);
                s1 += p.x;
                s2 += p.y;
            }
        }
        sh[wave][lane] = s1;
        sh[BNL_WAVES + wave][lane] = s2;
        __syncthreads();
        if (wave == 0 && n < N) {
            for (int w = 1; w < BNL_WAVES; ++w) {
                s1 += sh[w][lane];
                s2 += sh[BNL_WAVES + w][lane];
            }
            sums[2 * n] = s1;
            sums[2 * n + 1] = s2;
            if (dgamma) dgamma[n] = s2;
            if (dbeta) dbeta[n] = s1;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------
struct BnlFwdArgs {
    BnlSrc in;
    int M, K, N;
    int cpp;                   // chunks per part of the reduction: the layered product's summation order (glds_pick_ks)
    const float* W;
    const float* bias;
    float* Z;
    float* A_out;
    float* part;
    float eps, momentum;
    float* rmean;
    float* rvar;
    float* fin_out;
    long long* nbt;
    int nbt_inc;
    unsigned* ticket;
    const unsigned long long* ctr_src;
    unsigned long long* used_out;
    unsigned long long* tick;
};

template <int TPW>   // output tiles of 32 columns per wave
__global__ __launch_bounds__(BNL_THREADS) void bnl_fwd_kernel(BnlFwdArgs a) {
    extern __shared__ __align__(16) float lds[];   // image of A: [K / 32][32 rows][32 floats]; then per wave [2][TPW][1024] of W
    __shared__ unsigned long long ctr_s;
    __shared__ int last_s;
    __shared__ float sh[BNL_WAVES][64];
    __shared__ float4 colc[BNL_MAXW];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i0 = blockIdx.x * 32;
    const int nchunk = a.K >> 5, ntiles = (a.N + 31) >> 5;
    float* img = lds;
    float* my = lds + (size_t)nchunk * 1024 + (size_t)wave * (BNL_NBUF * TPW * 1024);

    // ---- this wave's tiles of W; a tile past the matrix repeats the wave's first (computed, never stored) -------------
    const bool has = wave < ntiles;
    int tj[TPW];
    bool valid[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int t = wave + BNL_WAVES * i;
        valid[i] = t < ntiles;
        tj[i] = valid[i] ? t : (has ? wave : 0);
    }
    const int srow = lane >> 3, sslot = lane & 7;
    const float* pb[TPW][4];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = 8 * q + srow;
            int rr = tj[i] * 32 + row;
            rr = rr < a.N ? rr : a.N - 1;
            pb[i][q] = a.W + (size_t)rr * a.K + ((sslot ^ ((row >> 1) & 7)) << 2);
        }
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < TPW; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                __builtin_amdgcn_global_load_lds((glb_vp)pb[i][q], (lds_vp)(my + (buf * TPW + i) * 1024 + q * 256), 16, 0, 0);
                pb[i][q] += 32;
            }
    };
    BNL_STAMP(0);
    if (has) {
#pragma unroll
        for (int b = 0; b < BNL_NBUF - 1; ++b)
            if (b < nchunk) stage(b);
    }

    unsigned long long ctr = 0;
    if (a.in.thr) ctr = drop_counter_enter(a.ctr_src, a.used_out, a.tick, &ctr_s);

    // ---- the tile of A, formed on the way into LDS: every thread's K / 32 units are requested first, then worked on ----------
    bnl_park_columns(a.in, a.K, colc);
    {
        const int K4 = a.K >> 2, U = a.K >> 5;
        float4 zv[BNL_UMAX];
#pragma unroll
        for (int i = 0; i < BNL_UMAX; ++i)
            if (i < U) {
                const int u = tid + i * BNL_THREADS, row = u / K4, kq = u - row * K4;
                const int gr = i0 + row < a.M ? i0 + row : a.M - 1;
                zv[i] = reinterpret_cast<const float4*>(a.in.Z + (size_t)gr * a.K)[kq];
            }
        __syncthreads();                     // the parked columns
#pragma unroll
        for (int i = 0; i < BNL_UMAX; ++i)
            if (i < U) {
                const int u = tid + i * BNL_THREADS, row = u / K4, kq = u - row * K4;
                const int gr = i0 + row < a.M ? i0 + row : a.M - 1;
                float4 xh, y;
                bnl_norm4c(a.in, colc, zv[i], kq, xh, y);
                if (a.in.thr) {
                    const float4 k = drop_keep4(a.in.seed, ctr, a.in.off4 + (unsigned long long)gr * K4 + kq, a.in.thr, a.in.scale);
                    y = make_float4(y.x * k.x, y.y * k.y, y.z * k.z, y.w * k.w);
                }
                if (a.A_out && i0 + row < a.M) reinterpret_cast<float4*>(a.A_out + (size_t)gr * a.K)[kq] = y;
                reinterpret_cast<float4*>(img)[(kq >> 3) * 256 + img_unit(row, kq & 7)] = y;
            }
    }
    __syncthreads();
    BNL_STAMP(1);

    // ---- Z tile = A W^T: A fragments from the shared image, W fragments from this wave's own ring -----------------------
    f32x16 acc[TPW][2];
#pragma unroll
    for (int i = 0; i < TPW; ++i) acc[i][0] = acc[i][1] = f32x16{0};
    f32x16 tot[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) tot[i] = f32x16{0};
    int in_part = 0;
    const int r = lane & 31, h = lane >> 5;
    if (has) {
        const unsigned ibase = (unsigned)(size_t)(__attribute__((address_space(3))) float*)img;
        const unsigned mbase = (unsigned)(size_t)(__attribute__((address_space(3))) float*)my;
        unsigned rn[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) rn[j] = r * 128 + (((2 * j + h) ^ ((r >> 1) & 7)) << 4);
        int buf = 0, nxt = BNL_NBUF - 1;      // ring slots of chunk t and of chunk t + NBUF - 1
        for (int t = 0; t < nchunk; ++t) {
            if (t + BNL_NBUF - 1 < nchunk) stage(nxt);
            bnl_wait_newer<TPW>(nchunk - 1 - t);        // chunk t has landed; up to NBUF - 1 newer ones stay in flight
            f4 av[4], bv[TPW][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(av[j]) : "v"(ibase + t * 4096 + rn[j]));
#pragma unroll
            for (int i = 0; i < TPW; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    asm volatile("ds_read_b128 %0, %1" : "=v"(bv[i][j]) : "v"(mbase + (buf * TPW + i) * 4096 + rn[j]));
            if constexpr (TPW == 2)
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]), "+v"(bv[0][0]), "+v"(bv[0][1]), "+v"(bv[0][2]),
                               "+v"(bv[0][3]), "+v"(bv[TPW - 1][0]), "+v"(bv[TPW - 1][1]), "+v"(bv[TPW - 1][2]), "+v"(bv[TPW - 1][3]));
            else
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]), "+v"(bv[0][0]), "+v"(bv[0][1]), "+v"(bv[0][2]),
                               "+v"(bv[0][3]));
#pragma unroll
            for (int q = 0; q < 16; q += 2)
#pragma unroll
                for (int i = 0; i < TPW; ++i) {
                    acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q >> 2][q & 3], bv[i][q >> 2][q & 3], acc[i][0], 0, 0, 0);
                    acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(q + 1) >> 2][(q + 1) & 3], bv[i][(q + 1) >> 2][(q + 1) & 3],
                                                                     acc[i][1], 0, 0, 0);
                }
            buf = buf + 1 == BNL_NBUF ? 0 : buf + 1;
            nxt = nxt + 1 == BNL_NBUF ? 0 : nxt + 1;
            if (++in_part == a.cpp) {     // a part of the reduction ends: the parts are added in order, as the layered product adds them
                in_part = 0;
#pragma unroll
                for (int i = 0; i < TPW; ++i) {
                    tot[i] += acc[i][0] + acc[i][1];
                    acc[i][0] = acc[i][1] = f32x16{0};
                }
            }
        }
    }

    BNL_STAMP(2);
    // ---- epilogue: + bias, the tile's (mean, M2) per column, Z ----------------------------------------------------------
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int col = tj[i] * 32 + r;
        if (has && valid[i] && col < a.N) {
            const f32x16 c = tot[i];
            const float bias = a.bias ? a.bias[col] : 0.f;
            float sum = 0.f, cnt = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ro = (e & 3) + 8 * (e >> 2);
                if (i0 + 4 * h + ro < a.M) {
                    sum += c[e] + bias;
                    cnt += 1.f;
                }
            }
            sum += __shfl_xor(sum, 32, 64);
            cnt += __shfl_xor(cnt, 32, 64);
            const float mean = sum / cnt;
            float m2 = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ro = (e & 3) + 8 * (e >> 2);
                if (i0 + 4 * h + ro < a.M) {
                    const float d = (c[e] + bias) - mean;
                    m2 = fmaf(d, d, m2);
                }
            }
            m2 += __shfl_xor(m2, 32, 64);
            if (h == 0) {
                st_dev(a.part + ((size_t)blockIdx.x * a.N + col) * 2, mean);
                st_dev(a.part + ((size_t)blockIdx.x * a.N + col) * 2 + 1, m2);
            }
            float* zp = a.Z + (size_t)(i0 + 4 * h) * a.N + col;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ro = (e & 3) + 8 * (e >> 2);
                if (i0 + 4 * h + ro < a.M) zp[(size_t)ro * a.N] = c[e] + bias;
            }
        }
    }

    // ---- the last workgroup to arrive folds the partials of all tiles ---------------------------------------------------
    BNL_STAMP(3);
    const int F = bnl_folders(a.N), f = bnl_arrive(a.ticket, F, &last_s);
    BNL_STAMP(4);
    if (f >= 0) {
        bnl_fold_stats(a.part, (int)gridDim.x, a.M, a.N, a.eps, a.momentum, a.rmean, a.rvar, a.fin_out, sh, f, F);
        if (a.nbt && f == 0 && tid == 0) *a.nbt += a.nbt_inc;
        BNL_STAMP(5);
        bnl_leave(a.ticket, F);
    }
}

// Y = act(BN(Z)): the output of the MLP's last layer
__global__ __launch_bounds__(256) void bnl_out_kernel(BnlSrc s, long long total, int N, float* __restrict__ Y) {
    if ((N & 3) == 0) {
        const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
        if (q * 4 >= total) return;
        const int kq = (int)(q % (N >> 2));
        float4 xh, y;
        bnl_norm4(s, reinterpret_cast<const float4*>(s.Z)[q], kq, xh, y);
        reinterpret_cast<float4*>(Y)[q] = y;
    } else {
        for (int e = 0; e < 4; ++e) {
            const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4 + e;
            if (i >= total) return;
            float xh, y;
            bnl_norm1(s, s.Z[i], (int)(i % N), xh, y);
            Y[i] = y;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------------------
// The (sum dy s, sum dy s xhat) partials of one 32-row tile for the columns of layer `L` (s = act'(y); y, xhat re-formed from the
// layer's Z), in bn_bwd_stats_kernel's order: four chains over the rows w, w + 4, ... (its four waves), added in order.  dy of
// (row, n) comes from `dy_at(row, n)`; rows past M contribute nothing.
template <typename F>
__device__ __forceinline__ void bnl_tile_sums(const BnlSrc& L, int M, int N, int i0, float* part, F dy_at) {
    for (int n = threadIdx.x; n < N; n += BNL_THREADS) {
        const float mean = L.fin[2 * n], is = L.fin[2 * n + 1], g = L.gamma[n], b = L.beta[n];
        float zr[32];
#pragma unroll
        for (int row = 0; row < 32; ++row) zr[row] = L.Z[(size_t)(i0 + row < M ? i0 + row : M - 1) * N + n];   // all 32 in flight
        float dyr[32];
#pragma unroll
        for (int row = 0; row < 32; ++row) dyr[row] = i0 + row < M ? dy_at(row, n) : 0.f;      // a row past M adds exactly nothing
        float p1[4], p2[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int row = w; row < 32; row += 4) {
                const float xh = (zr[row] - mean) * is;
                const float y = act_fwd(fmaf(g, xh, b), L.act);
                bn_bwd_acc(dyr[row], act_bwd(y, L.act), xh, s1, s2);
            }
            p1[w] = s1;
            p2[w] = s2;
        }
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            t1 += p1[w];
            t2 += p2[w];
        }
        st_dev(part + ((size_t)blockIdx.x * N + n) * 2, t1);
        st_dev(part + ((size_t)blockIdx.x * N + n) * 2 + 1, t2);
    }
}

__global__ __launch_bounds__(BNL_THREADS) void bnl_top_kernel(const float* __restrict__ dY, BnlSrc top, int M, int N, float* part,
                                                              float* sums, float* dgamma, float* dbeta, unsigned* ticket) {
    __shared__ int last_s;
    __shared__ float sh[2 * BNL_WAVES][64];
    const int i0 = blockIdx.x * 32;
    bnl_tile_sums(top, M, N, i0, part, [&](int row, int n) { return dY[(size_t)(i0 + row) * N + n]; });
    const int F = bnl_folders(N), f = bnl_arrive(ticket, F, &last_s);
    if (f >= 0) {
        bnl_fold_sums(part, (int)gridDim.x, N, sums, dgamma, dbeta, sh, f, F);
        bnl_leave(ticket, F);
    }
}

struct BnlBwdArgs {
    int M, N, K;
    int cpp;
    const float* G;          // [M, N] gradient at this layer's output y
    BnlSrc self;             // Z, fin, gamma of this layer
    const float* sums;       // [N][2] column sums of G and G xhat
    float* dZ_out;           // [M, N] or nullptr
    const float* W;          // [N, K]
    BnlSrc below;            // the layer below (Z == nullptr: none) and the dropout of THIS layer's input
    const unsigned long long* used;
    float* dA;               // [M, K]: G of the layer below, or the gradient of the MLP's input
    float* part;
    float* sums_out;
    float* dgamma;
    float* dbeta;
    unsigned* ticket;
};

template <int TPW>
__global__ __launch_bounds__(BNL_THREADS) void bnl_bwd_kernel(BnlBwdArgs a) {
    extern __shared__ __align__(16) float lds[];   // image of dZ [ceil(N / 32)][32][32], per wave [2][TPW][1024] of W; then t1 [32][K]
    __shared__ unsigned long long ctr_s;
    __shared__ int last_s;
    __shared__ float sh[2 * BNL_WAVES][64];
    __shared__ float4 colc[BNL_MAXW], cold[BNL_MAXW];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i0 = blockIdx.x * 32;
    const int nred = (a.N + 31) >> 5, ntiles = a.K >> 5;
    float* img = lds;
    float* my = lds + (size_t)nred * 1024 + (size_t)wave * (BNL_NBUF * TPW * 1024);

    const bool has = wave < ntiles;
    int tj[TPW];
    bool valid[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int t = wave + BNL_WAVES * i;
        valid[i] = t < ntiles;
        tj[i] = valid[i] ? t : (has ? wave : 0);
    }
    const int srow = lane >> 3, sslot = lane & 7;
    // block (chunk c, tile j) of W: rows n = 32 c + row (clamped below N: the image holds zeros there), columns 32 j ...
    auto stage = [&](int chunk, int buf) {
#pragma unroll
        for (int i = 0; i < TPW; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = 8 * q + srow;
                int rr = chunk * 32 + row;
                rr = rr < a.N ? rr : a.N - 1;
                const float* p = a.W + (size_t)rr * a.K + tj[i] * 32 + ((sslot ^ ((row >> 1) & 7)) << 2);
                __builtin_amdgcn_global_load_lds((glb_vp)p, (lds_vp)(my + (buf * TPW + i) * 1024 + q * 256), 16, 0, 0);
            }
    };
    if (has) {
#pragma unroll
        for (int b = 0; b < BNL_NBUF - 1; ++b)
            if (b < nred) stage(b, b);
    }

    if (tid == 0) ctr_s = a.below.thr ? *a.used : 0ull;

    // ---- dZ = invstd gamma (dy s - mean(dy s) - xhat mean(dy s xhat)), formed on the way into LDS (bn_bwd_apply_kernel's
    // expression: y, s = act'(y) and xhat re-formed from Z) ------------------------------------------------------------------
    {
        const float fm = (float)a.M;
        const int act = a.self.act;
        if ((a.N & 3) == 0) {
            // (mean, invstd, gamma, beta) and (sum / M, sum / M, invstd gamma, -) per column in LDS; a thread's units requested first
            bnl_park_columns(a.self, a.N, colc);
            for (int n = tid; n < a.N; n += BNL_THREADS)
                cold[n] = make_float4(a.sums[2 * n] / fm, a.sums[2 * n + 1] / fm, __fmul_rn(a.self.fin[2 * n + 1], a.self.gamma[n]), 0.f);
            const int N4 = a.N >> 2, P4 = nred * 8, U = nred;
            float4 gv[BNL_UMAX], zv[BNL_UMAX];
#pragma unroll
            for (int i = 0; i < BNL_UMAX; ++i)
                if (i < U) {
                    const int u = tid + i * BNL_THREADS, row = u / P4, kq = u - row * P4;
                    const int gr = i0 + row < a.M ? i0 + row : a.M - 1;
                    const int kc = kq < N4 ? kq : N4 - 1;
                    gv[i] = reinterpret_cast<const float4*>(a.G + (size_t)gr * a.N)[kc];
                    zv[i] = reinterpret_cast<const float4*>(a.self.Z + (size_t)gr * a.N)[kc];
                }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < BNL_UMAX; ++i)
                if (i < U) {
                    const int u = tid + i * BNL_THREADS, row = u / P4, kq = u - row * P4;
                    float4 dz = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (kq < N4) {
                        const int gr = i0 + row < a.M ? i0 + row : a.M - 1;
                        float4 xh, y;
                        bnl_norm4c(a.self, colc, zv[i], kq, xh, y);
                        const float4 d0 = cold[4 * kq], d1 = cold[4 * kq + 1], d2 = cold[4 * kq + 2], d3 = cold[4 * kq + 3];
                        dz.x = bn_bwd_dz(gv[i].x, act_bwd(y.x, act), xh.x, d0.x, d0.y, d0.z);
                        dz.y = bn_bwd_dz(gv[i].y, act_bwd(y.y, act), xh.y, d1.x, d1.y, d1.z);
                        dz.z = bn_bwd_dz(gv[i].z, act_bwd(y.z, act), xh.z, d2.x, d2.y, d2.z);
                        dz.w = bn_bwd_dz(gv[i].w, act_bwd(y.w, act), xh.w, d3.x, d3.y, d3.z);
                        if (a.dZ_out && i0 + row < a.M) reinterpret_cast<float4*>(a.dZ_out + (size_t)gr * a.N)[kq] = dz;
                    }
                    reinterpret_cast<float4*>(img)[(kq >> 3) * 256 + img_unit(row, kq & 7)] = dz;
                }
        } else {
            const int P = nred * 32;
            for (int e = tid; e < 32 * P; e += BNL_THREADS) {
                const int row = e / P, n = e - row * P;
                float dz = 0.f;
                if (n < a.N) {
                    const int gr = i0 + row < a.M ? i0 + row : a.M - 1;
                    const size_t i = (size_t)gr * a.N + n;
                    float xh, y;
                    bnl_norm1(a.self, a.self.Z[i], n, xh, y);
                    dz = bn_bwd_dz(a.G[i], act_bwd(y, act), xh, a.sums[2 * n] / fm, a.sums[2 * n + 1] / fm,
                                   __fmul_rn(a.self.fin[2 * n + 1], a.self.gamma[n]));
                    if (a.dZ_out && i0 + row < a.M) a.dZ_out[i] = dz;
                }
                const int kq = n >> 2;
                img[((kq >> 3) * 256 + img_unit(row, kq & 7)) * 4 + (n & 3)] = dz;
            }
        }
    }
    __syncthreads();

    if (!a.dA && !a.below.Z) return;         // only dZ was asked for (the MLP's input takes no gradient)

    // ---- dA tile = dZ W: dZ fragments from the image, W fragments (reduction index along W's rows) from the wave's ring ---
    f32x16 acc[TPW][2];
#pragma unroll
    for (int i = 0; i < TPW; ++i) acc[i][0] = acc[i][1] = f32x16{0};
    f32x16 tot[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) tot[i] = f32x16{0};
    int in_part = 0;
    const int r = lane & 31, h = lane >> 5;
    if (has) {
        const unsigned ibase = (unsigned)(size_t)(__attribute__((address_space(3))) float*)img;
        const unsigned mbase = (unsigned)(size_t)(__attribute__((address_space(3))) float*)my;
        unsigned rn[4], rt[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) rn[j] = r * 128 + (((2 * j + h) ^ ((r >> 1) & 7)) << 4);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = 8 * (q >> 2) + 4 * h + (q & 3);
            rt[q] = row * 128 + ((((r >> 2) ^ ((row >> 1) & 7)) << 4) | ((r & 3) << 2));
        }
        int buf = 0, nxt = BNL_NBUF - 1;
        for (int t = 0; t < nred; ++t) {
            if (t + BNL_NBUF - 1 < nred) stage(t + BNL_NBUF - 1, nxt);
            bnl_wait_newer<TPW>(nred - 1 - t);
            f4 av[4];
            float bt[TPW][16];
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(av[j]) : "v"(ibase + t * 4096 + rn[j]));
#pragma unroll
            for (int i = 0; i < TPW; ++i)
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    asm volatile("ds_read_b32 %0, %1" : "=v"(bt[i][q]) : "v"(mbase + (buf * TPW + i) * 4096 + rt[q]));
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]), "+v"(bt[0][0]), "+v"(bt[0][1]), "+v"(bt[0][2]),
                           "+v"(bt[0][3]), "+v"(bt[0][4]), "+v"(bt[0][5]), "+v"(bt[0][6]), "+v"(bt[0][7]), "+v"(bt[0][8]),
                           "+v"(bt[0][9]), "+v"(bt[0][10]), "+v"(bt[0][11]), "+v"(bt[0][12]), "+v"(bt[0][13]), "+v"(bt[0][14]),
                           "+v"(bt[0][15]));
            if constexpr (TPW == 2)
                asm volatile(""
                             : "+v"(bt[TPW - 1][0]), "+v"(bt[TPW - 1][1]), "+v"(bt[TPW - 1][2]), "+v"(bt[TPW - 1][3]), "+v"(bt[TPW - 1][4]),
                               "+v"(bt[TPW - 1][5]), "+v"(bt[TPW - 1][6]), "+v"(bt[TPW - 1][7]), "+v"(bt[TPW - 1][8]), "+v"(bt[TPW - 1][9]),
                               "+v"(bt[TPW - 1][10]), "+v"(bt[TPW - 1][11]), "+v"(bt[TPW - 1][12]), "+v"(bt[TPW - 1][13]),
                               "+v"(bt[TPW - 1][14]), "+v"(bt[TPW - 1][15]));
#pragma unroll
            for (int q = 0; q < 16; q += 2)
#pragma unroll
                for (int i = 0; i < TPW; ++i) {
                    acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q >> 2][q & 3], bt[i][q], acc[i][0], 0, 0, 0);
                    acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(q + 1) >> 2][(q + 1) & 3], bt[i][q + 1], acc[i][1], 0, 0, 0);
                }
            buf = buf + 1 == BNL_NBUF ? 0 : buf + 1;
            nxt = nxt + 1 == BNL_NBUF ? 0 : nxt + 1;
            if (++in_part == a.cpp || t + 1 == nred) {
                in_part = 0;
#pragma unroll
                for (int i = 0; i < TPW; ++i) {
                    tot[i] += acc[i][0] + acc[i][1];
                    acc[i][0] = acc[i][1] = f32x16{0};
                }
            }
        }
    }

    // ---- the dA tile through LDS (row-major): back through the dropout of the layer's input in whole 16-byte units, then -- with
    // a layer below -- that layer's column sums --------------------------------------------------------------------------------
    __syncthreads();                         // every wave is done with the image and its ring: t1 reuses the space
    float* t1 = lds;
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        if (has && valid[i]) {
            const f32x16 c = tot[i];
            const int col = tj[i] * 32 + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) t1[((e & 3) + 8 * (e >> 2) + 4 * h) * a.K + col] = c[e];
        }
    }
    __syncthreads();
    const unsigned long long ctr = ctr_s;
    const bool chain = a.below.Z != nullptr;
    {
        const int K4 = a.K >> 2, units = 32 * K4;
        for (int u = tid; u < units; u += BNL_THREADS) {
            const int row = u / K4, kq = u - row * K4;
            const bool live = i0 + row < a.M;
            const int gr = live ? i0 + row : a.M - 1;
            float4 v = reinterpret_cast<const float4*>(t1)[u];
            if (a.below.thr) {
                const float4 k = drop_keep4(a.below.seed, ctr, a.below.off4 + (unsigned long long)gr * K4 + kq, a.below.thr, a.below.scale);
                v = make_float4(v.x * k.x, v.y * k.y, v.z * k.z, v.w * k.w);
                if (chain) reinterpret_cast<float4*>(t1)[u] = v;
            }
            if (live && a.dA) reinterpret_cast<float4*>(a.dA + (size_t)gr * a.K)[kq] = v;
        }
    }
    if (!chain) return;
    __syncthreads();
    bnl_tile_sums(a.below, a.M, a.K, i0, a.part, [&](int row, int n) { return t1[row * a.K + n]; });
    const int F = bnl_folders(a.K), f = bnl_arrive(a.ticket, F, &last_s);
    if (f >= 0) {
        bnl_fold_sums(a.part, (int)gridDim.x, a.K, a.sums_out, a.dgamma, a.dbeta, sh, f, F);
        bnl_leave(a.ticket, F);
    }
}

}  // namespace fr

using namespace fr;

static int bnl_src(const fr_bn_src* s, int width, const char* who, BnlSrc* out, bool need_z) {
    FR_CHECK_ARG(s && (s->Z || !need_z) && s->act >= ACT_NONE && s->act <= ACT_TANH && s->drop_p >= 0.f && s->drop_p < 1.f &&
                     (s->drop_off & 3ull) == 0,
                 "%s: bad source descriptor", who);
    FR_CHECK_ARG(!s->fin || (s->gamma && s->beta), "%s: statistics without gamma / beta", who);
    FR_CHECK_ARG(((uintptr_t)s->Z & 15) == 0 && ((uintptr_t)s->fin & 15) == 0 && ((uintptr_t)s->gamma & 15) == 0 &&
                     ((uintptr_t)s->beta & 15) == 0,
                 "%s: 16-byte aligned tensors needed", who);
    (void)width;
    out->Z = s->Z;
    out->fin = s->fin;
    out->gamma = s->gamma;
    out->beta = s->beta;
    out->act = s->act;
    out->thr = s->drop_p > 0.f ? drop_threshold(s->drop_p) : 0u;
    out->scale = s->drop_p > 0.f ? 1.f / (1.f - s->drop_p) : 1.f;
    out->seed = s->drop_seed;
    out->off4 = s->drop_off >> 2;
    return FR_OK;
}

static bool bnl_widths_ok(int32_t K, int32_t N) { return K >= 32 && K % 32 == 0 && K <= BNL_MAXW && N >= 1 && N <= BNL_MAXW; }

#if BNL_TRACE
extern "C" __attribute__((visibility("default"))) int fr_bnl_trace_read(unsigned long long* out, int n_words) {
    FR_CHECK_HIP(hipDeviceSynchronize());
    FR_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(bnl_trace), (size_t)n_words * 8));
    return FR_OK;
}
#endif

extern "C" size_t fr_bnl_workspace_bytes(int64_t M, int32_t width) {
    if (M < 1 || width < 1) return 0;
    return (size_t)((M + 31) / 32) * (size_t)width * 2 * sizeof(float);
}

template <typename Kern>
static int bnl_lds_attr(Kern kern, size_t ldsb, size_t* have) {
    if (ldsb > *have) {
        FR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
        *have = ldsb;
    }
    return FR_OK;
}

extern "C" int fr_bnl_fwd(const fr_bn_src* in, int64_t M, int32_t K, const float* W, const float* bias, int32_t N, float* Z,
                          float* A_out, float eps, float momentum, float* running_mean, float* running_var, int64_t* nbt,
                          int32_t nbt_inc, float* fin_out, void* ws, size_t ws_bytes, uint32_t* ticket, const uint64_t* drop_state,
                          uint64_t* drop_used, uint64_t* drop_tick, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FR_CHECK_ARG(in && W && Z && fin_out && ws && ticket && M >= 1 && M <= 0x7fffffffLL / BNL_MAXW, "fr_bnl_fwd: bad argument");
    if (!bnl_widths_ok(K, N)) {
        set_error("fr_bnl_fwd: widths K = %d, N = %d outside the fused form (K %% 32 == 0, both <= %d)", K, N, BNL_MAXW);
        return FR_EUNSUPPORTED;
    }
    FR_CHECK_ARG(ws_bytes >= fr_bnl_workspace_bytes(M, N), "fr_bnl_fwd: workspace too small");
    FR_CHECK_ARG(((uintptr_t)W & 15) == 0 && ((uintptr_t)A_out & 15) == 0 && ((uintptr_t)fin_out & 15) == 0, "fr_bnl_fwd: 16-byte aligned tensors needed");
    FR_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "fr_bnl_fwd: running_mean and running_var go together");
    BnlFwdArgs a;
    int rc;
    if ((rc = bnl_src(in, K, "fr_bnl_fwd", &a.in, true))) return rc;
    FR_CHECK_ARG(!a.in.thr || drop_state, "fr_bnl_fwd: dropout without its counter");
    a.M = (int)M;
    a.K = K;
    a.N = N;
    a.cpp = K / 32 / glds_pick_ks((long long)((M + 31) / 32) * ((N + 31) / 32), K / 32);
    a.W = W;
    a.bias = bias;
    a.Z = Z;
    a.A_out = A_out;
    a.part = (float*)ws;
    a.eps = eps;
    a.momentum = momentum;
    a.rmean = running_mean;
    a.rvar = running_var;
    a.fin_out = fin_out;
    a.nbt = (long long*)nbt;
    a.nbt_inc = nbt_inc;
    a.ticket = ticket;
    a.ctr_src = (const unsigned long long*)drop_state;
    a.used_out = (unsigned long long*)drop_used;
    a.tick = (unsigned long long*)drop_tick;
    const int tpw = N > 128 ? 2 : 1;
    const size_t ldsb = ((size_t)(K / 32) * 1024 + (size_t)BNL_WAVES * BNL_NBUF * tpw * 1024) * sizeof(float);
    const dim3 grid((unsigned)((M + 31) / 32));
    prof_work(K_LINEAR_FWD, 2.0 * (double)M * N * K);
    ProfScope prof(K_LINEAR_FWD, stream);
    static size_t attr1 = 0, attr2 = 0;
    if (tpw == 2) {
        if ((rc = bnl_lds_attr(bnl_fwd_kernel<2>, ldsb, &attr2))) return rc;
        FR_LAUNCH(prof, bnl_fwd_kernel<2>, grid, dim3(BNL_THREADS), ldsb, stream, a);
    } else {
        if ((rc = bnl_lds_attr(bnl_fwd_kernel<1>, ldsb, &attr1))) return rc;
        FR_LAUNCH(prof, bnl_fwd_kernel<1>, grid, dim3(BNL_THREADS), ldsb, stream, a);
    }
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_bnl_out(const fr_bn_src* src, int64_t M, int32_t N, float* Y, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FR_CHECK_ARG(src && Y && M >= 1 && N >= 1 && ((uintptr_t)Y & 15) == 0, "fr_bnl_out: bad argument");
    BnlSrc s;
    int rc;
    if ((rc = bnl_src(src, N, "fr_bnl_out", &s, true))) return rc;
    FR_CHECK_ARG(s.thr == 0, "fr_bnl_out: the MLP's output is not dropped");
    const long long total = (long long)M * N;
    ProfScope prof(K_BN_FWD, stream);
    FR_LAUNCH(prof, bnl_out_kernel, dim3((unsigned)((total + 1023) / 1024)), dim3(256), 0, stream, s, total, (int)N, Y);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_bnl_bwd_top(const float* dY, const fr_bn_src* top, int64_t M, int32_t N, float* sums, float* dgamma, float* dbeta,
                              void* ws, size_t ws_bytes, uint32_t* ticket, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FR_CHECK_ARG(dY && top && sums && ws && ticket && M >= 1 && M <= 0x7fffffffLL / BNL_MAXW, "fr_bnl_bwd_top: bad argument");
    if (N < 1 || N > BNL_MAXW) {
        set_error("fr_bnl_bwd_top: width %d outside the fused form (<= %d)", N, BNL_MAXW);
        return FR_EUNSUPPORTED;
    }
    FR_CHECK_ARG(ws_bytes >= fr_bnl_workspace_bytes(M, N) && ((uintptr_t)sums & 15) == 0, "fr_bnl_bwd_top: workspace too small / sums unaligned");
    BnlSrc s;
    int rc;
    if ((rc = bnl_src(top, N, "fr_bnl_bwd_top", &s, true))) return rc;
    FR_CHECK_ARG(s.fin && s.thr == 0, "fr_bnl_bwd_top: the top layer comes with its statistics and without dropout");
    ProfScope prof(K_BN_BWD, stream);
    FR_LAUNCH(prof, bnl_top_kernel, dim3((unsigned)((M + 31) / 32)), dim3(BNL_THREADS), 0, stream, dY, s, (int)M, (int)N, (float*)ws,
              sums, dgamma, dbeta, ticket);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_bnl_bwd(const float* G, const fr_bn_src* self, const float* sums, int64_t M, int32_t N, const float* W, int32_t K,
                          float* dZ_out, const fr_bn_src* below, const uint64_t* drop_used, float* dA, float* sums_below,
                          float* dgamma_below, float* dbeta_below, void* ws, size_t ws_bytes, uint32_t* ticket, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FR_CHECK_ARG(G && self && sums && W && below && M >= 1 && M <= 0x7fffffffLL / BNL_MAXW, "fr_bnl_bwd: bad argument");
    if (!bnl_widths_ok(K, N)) {
        set_error("fr_bnl_bwd: widths K = %d, N = %d outside the fused form (K %% 32 == 0, both <= %d)", K, N, BNL_MAXW);
        return FR_EUNSUPPORTED;
    }
    BnlBwdArgs a;
    int rc;
    if ((rc = bnl_src(self, N, "fr_bnl_bwd", &a.self, true)) || (rc = bnl_src(below, K, "fr_bnl_bwd", &a.below, false))) return rc;
    FR_CHECK_ARG(a.self.fin && a.self.gamma && a.self.beta, "fr_bnl_bwd: the layer comes with its statistics");
    FR_CHECK_ARG(!a.below.Z || (a.below.fin && dA && sums_below && ws && ticket && ws_bytes >= fr_bnl_workspace_bytes(M, K)),
                 "fr_bnl_bwd: a layer below needs its statistics, dA, sums_below, workspace, ticket");
    FR_CHECK_ARG(a.below.Z || !a.below.fin, "fr_bnl_bwd: statistics of a layer below without its Z");
    FR_CHECK_ARG(dA || dZ_out, "fr_bnl_bwd: nothing to compute");
    FR_CHECK_ARG(!a.below.thr || drop_used, "fr_bnl_bwd: dropout without the counter value of the forward pass");
    FR_CHECK_ARG(((uintptr_t)G & 15) == 0 && ((uintptr_t)sums & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)dZ_out & 15) == 0 &&
                     ((uintptr_t)dA & 15) == 0 && ((uintptr_t)sums_below & 15) == 0,
                 "fr_bnl_bwd: 16-byte aligned tensors needed");
    a.M = (int)M;
    a.N = N;
    a.K = K;
    a.cpp = N % 32 == 0 ? N / 32 / glds_pick_ks((long long)((M + 31) / 32) * (K / 32), N / 32) : (N + 31) / 32;
    a.G = G;
    a.sums = sums;
    a.dZ_out = dZ_out;
    a.W = W;
    a.used = (const unsigned long long*)drop_used;
    a.dA = dA;
    a.part = (float*)ws;
    a.sums_out = sums_below;
    a.dgamma = dgamma_below;
    a.dbeta = dbeta_below;
    a.ticket = ticket;
    const int tpw = K > 128 ? 2 : 1;
    const size_t gemm = ((size_t)((N + 31) / 32) * 1024 + (size_t)BNL_WAVES * BNL_NBUF * tpw * 1024) * sizeof(float);
    const size_t tiles = (size_t)32 * K * sizeof(float);
    const size_t ldsb = gemm > tiles ? gemm : tiles;
    const dim3 grid((unsigned)((M + 31) / 32));
    prof_work(K_LINEAR_BWD_INPUT, 2.0 * (double)M * N * K);
    ProfScope prof(K_LINEAR_BWD_INPUT, stream);
    static size_t attr1 = 0, attr2 = 0;
    if (tpw == 2) {
        if ((rc = bnl_lds_attr(bnl_bwd_kernel<2>, ldsb, &attr2))) return rc;
        FR_LAUNCH(prof, bnl_bwd_kernel<2>, grid, dim3(BNL_THREADS), ldsb, stream, a);
    } else {
        if ((rc = bnl_lds_attr(bnl_bwd_kernel<1>, ldsb, &attr1))) return rc;
        FR_LAUNCH(prof, bnl_bwd_kernel<1>, grid, dim3(BNL_THREADS), ldsb, stream, a);
    }
    FR_CHECK_LAUNCH();
    return FR_OK;
}
