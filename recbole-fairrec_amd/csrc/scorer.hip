// The NFCF scorer -- MLPLayers([2 D, n1, n2, 1], dropout) on cat(U[user], I[item]), nfcf.py:40, :68-73 -- as ONE forward and
// ONE backward launch (the weight gradients are a third: fr_linear_bwd_weight_multi, mlp_glds.hip).
//
// Replaces, per training step, the layer-by-layer form of fairrec/model/layers.py: 3 dropout launches, 2 fp32-MFMA layers,
// the one-output layer, sigmoid + BCE (forward: 7 launches) and the one-output backward, 2 input-gradient products and the
// input dropout again (backward: 4 launches).  Same arithmetic per layer (layers.py:56-85: Dropout -> Linear -> ReLU, the
// last layer included), same dropout pattern (csrc/dropout.hpp: a function of seed, call counter and the element's offset
// in the call -- the offsets are the layered form's), fp32 MFMA products; a row's activations never leave the workgroup.
//
// Shape of a workgroup: 32 batch rows, 8 waves.
//   forward   X tile [32, k0 + k1] -> LDS (dropped on the way; the dropped rows also go to memory for the weight gradient)
//             Z1 = X W1^T: wave (g, t) owns output columns 32 t .. 32 t + 31 over half g of the reduction (g = 0: the user
//             block, 1: the item block); W1 arrives in 32-wide reduction chunks, all rows, double-buffered through LDS;
//             the two halves meet in LDS; + b1, ReLU, dropout -> H1 (LDS and memory)
//             Z2 = H1 W2^T: wave = (column tile, 32-wide reduction chunk), partials meet in LDS; + b2, ReLU, dropout -> H2
//             y = relu(H2 . w3 + b3): 16 lanes per row; sigmoid, BCE and dLoss/dy of the BCE term (nfcf.py:73, :105)
//   backward  dz3 = dy [y > 0]; dz2 = dz3 w3 o [H2 > 0] scale; dz1 = (dz2 W2) o [H1 > 0] scale; dX = (dz1 W1) o keep
//             (only the blocks whose table trains: the finetune stage freezes the user table, nfcf.py:66);
//             dz1 / dz2 / dz3 go to memory for the weight-gradient launch, per-workgroup partials of dW3 / db3 with them.
// Fragment layout of v_mfma_f32_32x32x2_f32 as in mlp_glds.hip: step q = 4 j + e of a 32-chunk multiplies reduction
// elements 8 j + 4 h + e (h = lane / 32); accumulator e holds row (e & 3) + 8 (e >> 2) + 4 h, column lane % 32.
#include "common.hpp"
#include "kernels.hpp"
#include "dropout.hpp"

namespace fr {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f4 = __attribute__((ext_vector_type(4))) float;

constexpr int SC_THREADS = 512, SC_ROWS = 32;

struct ScDrop {
    int on;
    float scale;
    unsigned thr;
    unsigned long long seed;
    const unsigned long long* ctr;   // forward: the module's call counter; backward: the value the forward used
    unsigned long long* used;
    unsigned long long* tick;
    unsigned long long g_x0, g_x1, g_h1, g_h2;   // offset / 4 of the four dropped tensors
};

struct ScShape {
    int k0, k1, n1, n2, B;
    const float *W1, *b1, *W2, *b2, *W3, *b3;
};

struct ScFwd {
    ScShape s;
    ScDrop d;
    const float *x0, *x1;
    float *x0d, *x1d, *h1, *h2, *y;
    const float *label, *sst;
    float *out, *dy, *bce_part, *mm_part;
    float* loss;                 // no fairness term behind this launch: the last workgroup to finish writes loss[3]
};

__device__ unsigned int sc_ticket;   // arrivals of a forward launch that closes its own loss (back to 0 when it ends)

struct ScBwd {
    ScShape s;
    ScDrop d;
    const float *dy, *gscale, *y, *h1, *h2;
    float *dz1, *dz2, *dz3, *dx0, *dx1, *w3part;
};

__device__ __forceinline__ f4 keep_mul(f4 v, float4 k) {
    v[0] *= k.x; v[1] *= k.y; v[2] *= k.z; v[3] *= k.w;
    return v;
}

#define SC_MFMA16(av, bv, acc0, acc1)                                                                   \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                  \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j_][0], bv[j_][0], acc0, 0, 0, 0);                \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j_][1], bv[j_][1], acc1, 0, 0, 0);                \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j_][2], bv[j_][2], acc0, 0, 0, 0);                \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j_][3], bv[j_][3], acc1, 0, 0, 0);                \
    }

#ifdef FR_SC_TRACE
__device__ unsigned long long sc_trace[64];
#define SC_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) sc_trace[i] = __builtin_readcyclecounter(); } while (0)
#else
#define SC_STAMP(i) do {} while (0)
#endif

__global__ __launch_bounds__(SC_THREADS) void scorer_fwd_kernel(ScFwd a) {
    extern __shared__ __align__(16) float lds[];
    __shared__ unsigned long long ctr_s;
    __shared__ float lrow[SC_ROWS], lo_s[SC_ROWS], hi_s[SC_ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const ScShape& s = a.s;
    const int K0 = s.k0 + s.k1, HS1 = s.n1 + 4, HS2 = s.n2 + 4;
    const int i0 = blockIdx.x * SC_ROWS;
    const int NCH = K0 >> 6;                        // 32-chunks per half of the reduction (half 0 = block x0, 1 = x1)
    // a stage of the ring = the 32-chunk of step t of both halves: X [2][32][36] | W1 [2][n1][36] (rows padded to 144 bytes)
    constexpr int XST = 2 * SC_ROWS * 36;
    const int STAGE = XST + 2 * s.n1 * 36;
    SC_STAMP(0);

    // ---- what this thread brings per step: one float4 of X, four of W1 ----------------------------------------------------
    const int xh = tid >> 8, xr = (tid >> 3) & 31, xs = tid & 7;
    const bool xok = i0 + xr < s.B;
    const long long xg = xok ? i0 + xr : s.B - 1;
    const int kx = xh ? s.k1 : s.k0;
    const float* xsrc = (xh ? a.x1 : a.x0) + xg * kx + xs * 4;
    float* xdst = a.d.on ? (xh ? a.x1d : a.x0d) + xg * kx + xs * 4 : nullptr;
    const unsigned long long xgrp = (xh ? a.d.g_x1 : a.d.g_x0) + (((unsigned long long)xg * kx) >> 2) + xs;
    const int xoff = (xh * SC_ROWS + xr) * 36 + xs * 4;
    const int wq = s.n1 * 8;                        // float4 per half and chunk
    const float* wsrc[4];
    int woff[4];
    bool wok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + SC_THREADS * i;
        wok[i] = idx < 2 * wq;
        const int id = wok[i] ? idx : 0;
        const int hf = id >= wq, rem = id - hf * wq;
        wsrc[i] = s.W1 + (size_t)(rem >> 3) * K0 + hf * NCH * 32 + (rem & 7) * 4;
        woff[i] = XST + (hf * s.n1 + (rem >> 3)) * 36 + (rem & 7) * 4;
    }
    // chunks 0, 1 and 2 are fetched together; from then on chunk t + 3 is fetched during step t, written to its stage during
    // step t + 1 (in front of that step's MFMAs, under which the writes then run) and multiplied in step t + 3
    struct Fetch {
        f4 x, w[4];
    } fa_, fb_, fc_;
    auto fetch = [&](Fetch& f, int t) {
        f.x = *reinterpret_cast<const f4*>(xsrc + t * 32);
#pragma unroll
        for (int i = 0; i < 4; ++i) f.w[i] = *reinterpret_cast<const f4*>(wsrc[i] + t * 32);
    };
    fetch(fa_, 0);
    fetch(fb_, NCH > 1 ? 1 : 0);
    fetch(fc_, NCH > 2 ? 2 : 0);

    // ---- operands of the later phases, fetched now: their latency passes under the first product ----------------------------------
    const int half = wave >> 2, tw = wave & 3, n0 = tw * 32;
    const bool live = n0 < s.n1;
    const int T2 = s.n2 >> 5, C2 = s.n1 >> 5;
    const int t2 = wave % T2, part = wave / T2;
    const bool live2 = part < C2;
    f4 b2v[4];
    {
        const float* wp = s.W2 + (size_t)(t2 * 32 + r) * s.n1 + (live2 ? part : 0) * 32 + h * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) b2v[j] = *reinterpret_cast<const f4*>(wp + j * 8);
    }
    const float bias1 = s.b1[live ? n0 + r : 0], bias2 = s.b2[t2 * 32 + r];
    const int row3 = tid >> 4, sub = tid & 15;
    const bool has3 = sub < (s.n2 >> 2);
    const f4 w3v = *reinterpret_cast<const f4*>(s.W3 + (has3 ? sub : 0) * 4);
    const float bias3 = s.b3[0];
    const int b3row = i0 + row3 < s.B ? i0 + row3 : s.B - 1;
    const float lab = a.label ? a.label[b3row] : 0.f, sstv = a.sst ? a.sst[b3row] : 0.f;

    unsigned long long ctr = 0;
    if (a.d.on) ctr = drop_counter_enter(a.d.ctr, a.d.used, a.d.tick, &ctr_s);
    SC_STAMP(1);
    auto put = [&](Fetch& f, int t, float* stage) {      // the fetched chunk t: X dropped (and to memory), everything into the stage
        if (a.d.on) {
            f.x = keep_mul(f.x, drop_keep4(a.d.seed, ctr, xgrp + (unsigned long long)t * 8, a.d.thr, a.d.scale));
            if (xok) *reinterpret_cast<f4*>(xdst + t * 32) = f.x;
        }
        *reinterpret_cast<f4*>(stage + xoff) = f.x;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (wok[i]) *reinterpret_cast<f4*>(stage + woff[i]) = f.w[i];
    };
    put(fa_, 0, lds);
    if (NCH > 1) put(fb_, 1, lds + STAGE);
    __syncthreads();
    SC_STAMP(3);

    // ---- Z1 = X W1^T: a ring of three stages; the fragments of chunk t + 1 are read, chunk t + 2 written and chunk t + 3
    // fetched under the MFMAs of chunk t, so that neither LDS traffic nor fetch latency sits between two steps' products ------
    const int fa = (half * SC_ROWS + r) * 36 + h * 4, fb = XST + (half * s.n1 + (live ? n0 : 0) + r) * 36 + h * 4;
    f4 av[4], bv[4], an[4], bn[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        av[j] = *reinterpret_cast<const f4*>(lds + fa + j * 8);
        bv[j] = *reinterpret_cast<const f4*>(lds + fb + j * 8);
    }
    // (the first fragments have landed before the loop is entered: with reads still in flight at its head the compiler's wait
    // insertion, which must cover both ways into the loop, puts an LDS wait in front of every step's MFMAs)
    __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0)
    f32x16 acc0 = {0}, acc1 = {0};
    // one step; `cur` holds chunk t + 2 (fetched a step ago, or before the loop), `nxt` receives chunk t + 3
    auto step = [&](int t, Fetch& cur, Fetch& nxt) {
        if (t + 1 < NCH) {
            const float* nx = lds + ((t + 1) % 3) * STAGE;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                an[j] = *reinterpret_cast<const f4*>(nx + fa + j * 8);
                bn[j] = *reinterpret_cast<const f4*>(nx + fb + j * 8);
            }
        }
        if (t + 2 < NCH) put(cur, t + 2, lds + ((t + 2) % 3) * STAGE);
        // (fetched after the put, whose wait for `cur` would otherwise cover these loads as well: the memory counter is in order)
        if (t + 3 < NCH) fetch(nxt, t + 3);
        if (live) { SC_MFMA16(av, bv, acc0, acc1) }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            av[j] = an[j];
            bv[j] = bn[j];
        }
        SC_STAMP(4 + (t < 8 ? t : 7));
    };
    for (int t = 0; t < NCH; t += 2) {      // (two steps per trip: the two fetch buffers swap roles without a copy)
        step(t, fc_, fa_);
        if (t + 1 < NCH) step(t + 1, fa_, fc_);
    }
    f32x16 acc = acc0 + acc1;
    float* red = lds;                      // [4][16][64]
    float* H1s = lds + 4096;               // [32][HS1]
    float* H2s = H1s + SC_ROWS * HS1;      // [32][HS2]
    float* red2 = H2s + SC_ROWS * HS2;     // [<= 6][16][64]
    if (half == 1 && live) {
#pragma unroll
        for (int e = 0; e < 16; ++e) red[(tw * 16 + e) * 64 + lane] = acc[e];
    }
    __syncthreads();
    if (half == 0 && live) {
        const int col = n0 + r;
        float other[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) other[e] = red[(tw * 16 + e) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
            const float v = (acc[e] + other[e]) + bias1;
            H1s[row * HS1 + col] = v > 0.f ? v : 0.f;
        }
    }
    __syncthreads();
    {
        const int q1 = s.n1 >> 2;
        const float inv = 1.f / (float)q1;
        for (int idx = tid; idx < SC_ROWS * q1; idx += SC_THREADS) {
            const int row = (int)(((float)idx + 0.5f) * inv), c4 = idx - row * q1;
            const long long gr = i0 + row;
            f4 v = *reinterpret_cast<f4*>(H1s + row * HS1 + c4 * 4);
            if (a.d.on) {
                v = keep_mul(v, drop_keep4(a.d.seed, ctr, a.d.g_h1 + (((unsigned long long)gr * s.n1) >> 2) + c4, a.d.thr, a.d.scale));
                *reinterpret_cast<f4*>(H1s + row * HS1 + c4 * 4) = v;
            }
            if (gr < s.B) *reinterpret_cast<f4*>(a.h1 + gr * s.n1 + c4 * 4) = v;
        }
    }
    __syncthreads();
    SC_STAMP(12);

    // ---- Z2 = H1 W2^T ------------------------------------------------------------------------------------------------------
    f32x16 c0 = {0}, c1 = {0};
    if (live2) {
        const float* xa = H1s + r * HS1 + part * 32 + h * 4;
        f4 av[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) av[j] = *reinterpret_cast<const f4*>(xa + j * 8);
        SC_MFMA16(av, b2v, c0, c1)
    }
    acc = c0 + c1;
    if (live2 && part > 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) red2[((t2 * (C2 - 1) + part - 1) * 16 + e) * 64 + lane] = acc[e];
    }
    __syncthreads();
    if (live2 && part == 0) {
        const int col = t2 * 32 + r;
#pragma unroll
        for (int pp = 1; pp < 4; ++pp) {       // the other chunks' partials, in chunk order (C2 <= 4)
            if (pp < C2) {
                float other[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) other[e] = red2[((t2 * (C2 - 1) + pp - 1) * 16 + e) * 64 + lane];
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] += other[e];
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float v = acc[e] + bias2;
            const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
            H2s[row * HS2 + col] = v > 0.f ? v : 0.f;
        }
    }
    __syncthreads();
    {
        const int q2 = s.n2 >> 2;
        const float inv = 1.f / (float)q2;
        for (int idx = tid; idx < SC_ROWS * q2; idx += SC_THREADS) {
            const int row = (int)(((float)idx + 0.5f) * inv), c4 = idx - row * q2;
            const long long gr = i0 + row;
            f4 v = *reinterpret_cast<f4*>(H2s + row * HS2 + c4 * 4);
            if (a.d.on) {
                v = keep_mul(v, drop_keep4(a.d.seed, ctr, a.d.g_h2 + (((unsigned long long)gr * s.n2) >> 2) + c4, a.d.thr, a.d.scale));
                *reinterpret_cast<f4*>(H2s + row * HS2 + c4 * 4) = v;
            }
            if (gr < s.B) *reinterpret_cast<f4*>(a.h2 + gr * s.n2 + c4 * 4) = v;
        }
    }
    __syncthreads();
    SC_STAMP(13);

    // ---- y = relu(H2 . w3 + b3), sigmoid, BCE -----------------------------------------------------------------------------------
    {
        float sum = 0.f;
        if (has3) {
            const f4 hv = *reinterpret_cast<const f4*>(H2s + row3 * HS2 + sub * 4);
            sum = (hv[0] * w3v[0] + hv[1] * w3v[1]) + (hv[2] * w3v[2] + hv[3] * w3v[3]);
        }
        sum = group_sum<16>(sum);
        if (sub == 0) {
            const float z = sum + bias3;
            const float yv = z > 0.f ? z : 0.f;
            const int b = i0 + row3;
            float l = 0.f, lo = INFINITY, hi = -INFINITY;
            if (b < s.B) {
                a.y[b] = yv;
                if (a.label) {
                    // as nfcf_bce_kernel (csrc/nfcf.hip): torch's binary_cross_entropy clamps both logs at -100
                    const float o = 1.f / (1.f + __expf(-yv));
                    const float l0 = fmaxf(__logf(o), -100.f), l1 = fmaxf(__logf(1.f - o), -100.f);
                    l = -(lab * l0 + (1.f - lab) * l1);
                    a.out[b] = o;
                    const float sg = o * (1.f - o);
                    a.dy[b] = (o - lab) / fmaxf(sg, 1e-12f) / (float)s.B * sg;
                    if (a.sst && lab == 1.f) lo = hi = sstv;
                }
            }
            lrow[row3] = l;
            lo_s[row3] = lo;
            hi_s[row3] = hi;
        }
    }
    __syncthreads();
    if (tid == 0 && a.label) {
        float l = 0.f, lo = INFINITY, hi = -INFINITY;
        for (int i = 0; i < SC_ROWS; ++i) {
            l += lrow[i];
            lo = fminf(lo, lo_s[i]);
            hi = fmaxf(hi, hi_s[i]);
        }
        if (a.loss) __hip_atomic_store(a.bce_part + blockIdx.x, l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else a.bce_part[blockIdx.x] = l;
        if (a.sst) {
            a.mm_part[2 * blockIdx.x] = lo;
            a.mm_part[2 * blockIdx.x + 1] = hi;
        }
    }
    if (a.loss) {      // loss = mean BCE (nfcf.py:105-107): summed by whoever arrives last, in workgroup order
        __shared__ bool last;
        if (tid == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this workgroup's partial (a device-scope store) has landed
            const unsigned t = __hip_atomic_fetch_add(&sc_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = t == gridDim.x - 1;
            if (last) sc_ticket = 0u;
        }
        __syncthreads();
        if (last && tid < 64) {
            float v = 0.f;
            for (int q = tid; q < (int)gridDim.x; q += 64) v += __hip_atomic_load(a.bce_part + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v = wave_sum(v);
            if (tid == 0) {
                const float bce = v / (float)s.B;
                a.loss[0] = bce;
                a.loss[1] = bce;
                a.loss[2] = 0.f;
            }
        }
    }
    SC_STAMP(14);
}

__global__ __launch_bounds__(SC_THREADS) void scorer_bwd_kernel(ScBwd a) {
    extern __shared__ __align__(16) float lds[];
    __shared__ unsigned long long ctr_s;
    __shared__ float d3s[SC_ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const ScShape& s = a.s;
    const int K0 = s.k0 + s.k1, HS1 = s.n1 + 4, HS2 = s.n2 + 4;
    const int i0 = blockIdx.x * SC_ROWS;
    const int cbeg = a.dx0 ? 0 : s.k0, cend = a.dx1 ? K0 : s.k0;
    const int DS = (cend > cbeg ? cend - cbeg : 0) + 4;
    float* DZ2s = lds;                        // [32][HS2]
    float* H2s = DZ2s + SC_ROWS * HS2;        // [32][HS2]
    float* DZ1s = H2s + SC_ROWS * HS2;        // [32][HS1]
    float* red = DZ1s + SC_ROWS * HS1;        // [4][16][64]
    float* DXs = red + 4096;                  // [32][DS]
    unsigned long long ctr = 0;
    SC_STAMP(16);
    // ---- the weights' fragments of both products and the ReLU masks, fetched before anything else: their latency passes under
    // the elementwise phases (the products below would otherwise wait for each chunk's 16 strided loads in turn) --------------
    const int T1 = s.n1 >> 5, C = s.n2 >> 5, C1 = s.n1 >> 5;
    const int tile1 = wave % T1, part1 = wave / T1;
    const bool live1 = part1 < C;
    const int ntile = cend > cbeg ? (cend - cbeg) >> 5 : 0;
    f4 w2f[4], h1m[4], w1f[4][4];
    {
        const float* wp = s.W2 + (size_t)((live1 ? part1 : 0) * 32 + 4 * h) * s.n1 + tile1 * 32 + r;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) w2f[j][e] = wp[(size_t)(8 * j + e) * s.n1];
    }
    auto fetch_w1 = [&](int tile) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < C1) {
                const float* wp = s.W1 + (size_t)(c * 32 + 4 * h) * K0 + cbeg + tile * 32 + r;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) w1f[c][j][e] = wp[(size_t)(8 * j + e) * K0];
            }
        }
    };
    // (w1f holds one [n1, 32] column block as four chunks of 16 floats per lane: f4 w1f[c][j], element e)
    if (wave < ntile) fetch_w1(wave);
    if (a.d.on) ctr = drop_counter_enter(a.d.ctr, nullptr, nullptr, &ctr_s);
    SC_STAMP(17);

    if (tid < SC_ROWS) {
        const int b = i0 + tid;
        float d = 0.f;
        if (b < s.B) {
            d = a.dy[b];
            if (a.gscale) d *= a.gscale[0];
            d = a.y[b] > 0.f ? d : 0.f;
            a.dz3[b] = d;
        }
        d3s[tid] = d;
    }
    __syncthreads();
    {
        const int q2 = s.n2 >> 2;
        for (int idx = tid; idx < SC_ROWS * q2; idx += SC_THREADS) {
            const int row = idx / q2, c4 = idx - row * q2;
            const bool ok = i0 + row < s.B;
            const long long gr = ok ? i0 + row : s.B - 1;
            const f4 hv = *reinterpret_cast<const f4*>(a.h2 + gr * s.n2 + c4 * 4);
            const f4 wv = *reinterpret_cast<const f4*>(s.W3 + c4 * 4);
            const float d = d3s[row];
            f4 dz;
#pragma unroll
            for (int e = 0; e < 4; ++e) dz[e] = hv[e] > 0.f ? d * wv[e] * a.d.scale : 0.f;
            *reinterpret_cast<f4*>(DZ2s + row * HS2 + c4 * 4) = dz;
            *reinterpret_cast<f4*>(H2s + row * HS2 + c4 * 4) = hv;
            if (ok) *reinterpret_cast<f4*>(a.dz2 + gr * s.n2 + c4 * 4) = dz;
        }
    }
    __syncthreads();
    SC_STAMP(18);
#pragma unroll
    for (int e = 0; e < 16; ++e) {      // the ReLU masks of the next product's epilogue: on their way during the dW3 shares
        const long long gr = i0 + (e & 3) + 8 * (e >> 2) + 4 * h;
        h1m[e >> 2][e & 3] = a.h1[(gr < s.B ? gr : s.B - 1) * s.n1 + tile1 * 32 + r];
    }
    {   // this tile's share of dW3 (columns 0 .. n2 - 1) and db3 (column n2): four runs of eight rows, added in run order
        const int k = tid & 127, run = tid >> 7;
        float sum = 0.f;
        if (k <= s.n2) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int row = run * 8 + q;
                sum += d3s[row] * (k < s.n2 ? H2s[row * HS2 + k] : 1.f);
            }
            red[run * 128 + k] = sum;
        }
        __syncthreads();
        if (run == 0 && k <= s.n2)
            a.w3part[(size_t)blockIdx.x * (s.n2 + 1) + k] = ((red[k] + red[128 + k]) + red[256 + k]) + red[384 + k];
        __syncthreads();
    }

    SC_STAMP(19);
    // ---- dz1 = (dz2 W2) o [H1 > 0] scale -----------------------------------------------------------------------------------
    {
        const int tile = tile1, part = part1;
        const bool live = live1;
        f32x16 c0 = {0}, c1 = {0};
        if (live) {
            const float* xa = DZ2s + r * HS2 + part * 32 + h * 4;
            f4 av[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) av[j] = *reinterpret_cast<const f4*>(xa + j * 8);
            SC_MFMA16(av, w2f, c0, c1)
        }
        f32x16 acc = c0 + c1;
        if (live && part > 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) red[(tile * 16 + e) * 64 + lane] = acc[e];
        }
        __syncthreads();
        if (live && part == 0) {
            const int col = tile * 32 + r;
            float other[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) other[e] = C > 1 ? red[(tile * 16 + e) * 64 + lane] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
                float v = acc[e];
                if (C > 1) v += other[e];
                const long long gr = i0 + row;
                v = (gr < s.B && h1m[e >> 2][e & 3] > 0.f) ? v * a.d.scale : 0.f;
                DZ1s[row * HS1 + col] = v;
                if (gr < s.B) a.dz1[gr * s.n1 + col] = v;
            }
        }
        __syncthreads();
    }
    SC_STAMP(20);
    if (cend <= cbeg) return;

    // ---- dX = (dz1 W1) o keep, the blocks that train -------------------------------------------------------------------------
    {
        for (int tile = wave; tile < ntile; tile += SC_THREADS / 64) {
            if (tile != wave) fetch_w1(tile);        // (the first block of every wave came in at the top of the kernel)
            f32x16 c0 = {0}, c1 = {0};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c < C1) {
                    const float* xa = DZ1s + r * HS1 + c * 32 + h * 4;
                    f4 av[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) av[j] = *reinterpret_cast<const f4*>(xa + j * 8);
                    SC_MFMA16(av, w1f[c], c0, c1)
                }
            }
            const f32x16 acc = c0 + c1;
#pragma unroll
            for (int e = 0; e < 16; ++e) DXs[((e & 3) + 8 * (e >> 2) + 4 * h) * DS + tile * 32 + r] = acc[e];
        }
        __syncthreads();
        SC_STAMP(21);
        const int qx = (cend - cbeg) >> 2;
        for (int idx = tid; idx < SC_ROWS * qx; idx += SC_THREADS) {
            const int row = idx / qx, c4 = idx - row * qx;
            const long long gr = i0 + row;
            if (gr >= s.B) continue;
            const int col = cbeg + c4 * 4;
            const bool second = col >= s.k0;
            const int cc = (second ? col - s.k0 : col) >> 2;
            f4 v = *reinterpret_cast<const f4*>(DXs + row * DS + c4 * 4);
            if (a.d.on) {
                const unsigned long long g = second ? a.d.g_x1 + (((unsigned long long)gr * s.k1) >> 2) + cc
                                                    : a.d.g_x0 + (((unsigned long long)gr * s.k0) >> 2) + cc;
                v = keep_mul(v, drop_keep4(a.d.seed, ctr, g, a.d.thr, a.d.scale));
            }
            *reinterpret_cast<f4*>((second ? a.dx1 + gr * s.k1 : a.dx0 + gr * s.k0) + cc * 4) = v;
        }
    }
    SC_STAMP(22);
}

// out[i] = sum over the parts p of part[p * n + i]: one wave per output, lane l adds parts l, l + 64, ... in ascending order and
// the 64 lane sums meet in wave_sum's fixed butterfly (the same bits on every run)
__global__ __launch_bounds__(256) void parts_sum_kernel(const float* __restrict__ part, int parts, int n, float* __restrict__ out) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    float s = 0.f;
    for (int p = lane; p < parts; p += 64) s += part[(size_t)p * n + i];
    s = wave_sum(s);
    if (lane == 0) out[i] = s;
}

static bool shape_ok(const fr_scorer* s) {
    return s && s->k0 >= 32 && s->k0 % 32 == 0 && s->k1 == s->k0 && s->k0 + s->k1 <= 512 && s->n1 >= 32 && s->n1 % 32 == 0 &&
           s->n1 <= 128 && s->n2 >= 32 && s->n2 % 32 == 0 && s->n2 <= 64 && s->p >= 0.f && s->p < 1.f;
}

static int fill(const fr_scorer* s, int64_t B, ScShape* sh, ScDrop* d) {
    *sh = ScShape{s->k0, s->k1, s->n1, s->n2, (int)B, s->W1, s->b1, s->W2, s->b2, s->W3, s->b3};
    d->on = s->p > 0.f;
    d->scale = d->on ? 1.f / (1.f - s->p) : 1.f;
    d->thr = drop_threshold(s->p);
    d->seed = s->seed;
    d->g_x0 = s->off_x0 / 4;
    d->g_x1 = s->off_x1 / 4;
    d->g_h1 = s->off_h1 / 4;
    d->g_h2 = s->off_h2 / 4;
    return 0;
}

}  // namespace fr

using namespace fr;

#ifdef FR_SC_TRACE
extern "C" __attribute__((visibility("default"))) int fr_debug_scorer_trace(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(sc_trace), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int fr_scorer_supported(const fr_scorer* s) { return shape_ok(s) ? 1 : 0; }

extern "C" int64_t fr_scorer_blocks(int64_t B) { return B < 1 ? 0 : (B + SC_ROWS - 1) / SC_ROWS; }

extern "C" int fr_scorer_fwd(const fr_scorer* s, const float* x0, const float* x1, int64_t B, const int64_t* counter,
                             int64_t* used_out, int64_t* tick_state, float* x0d, float* x1d, float* h1, float* h2, float* y,
                             const float* label, const float* sst, float* out, float* dy, float* bce_part, float* mm_part,
                             float* loss, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FR_CHECK_ARG(shape_ok(s), "fr_scorer_fwd: shape not supported (k0 == k1, multiples of 32, k0 + k1 <= 512; n1 <= 128, n2 <= 64, multiples of 32)");
    FR_CHECK_ARG(x0 && x1 && h1 && h2 && y && B >= 1 && B < (1ll << 31) && s->W1 && s->b1 && s->W2 && s->b2 && s->W3 && s->b3,
                 "fr_scorer_fwd: bad argument");
    FR_CHECK_ARG(s->p == 0.f || (counter && used_out && x0d && x1d && s->off_x0 % 4 == 0 && s->off_x1 % 4 == 0 &&
                                 s->off_h1 % 4 == 0 && s->off_h2 % 4 == 0),
                 "fr_scorer_fwd: dropout needs the counter, the `used` word, the dropped-input buffers and offsets that are multiples of 4");
    FR_CHECK_ARG(!label || (out && dy && bce_part && (!sst || mm_part)), "fr_scorer_fwd: the loss head needs out, dy and the partial buffers");
    FR_CHECK_ARG((((uintptr_t)x0 | (uintptr_t)x1 | (uintptr_t)h1 | (uintptr_t)h2 | (uintptr_t)x0d | (uintptr_t)x1d |
                   (uintptr_t)s->W1 | (uintptr_t)s->W2 | (uintptr_t)s->W3) & 15) == 0, "fr_scorer_fwd: 16-byte alignment required");
    ScFwd a{};
    fill(s, B, &a.s, &a.d);
    a.d.ctr = (const unsigned long long*)counter;
    a.d.used = (unsigned long long*)used_out;
    a.d.tick = (unsigned long long*)tick_state;
    a.x0 = x0; a.x1 = x1; a.x0d = x0d; a.x1d = x1d; a.h1 = h1; a.h2 = h2; a.y = y;
    a.label = label; a.sst = label ? sst : nullptr; a.out = out; a.dy = dy; a.bce_part = bce_part; a.mm_part = mm_part;
    a.loss = label ? loss : nullptr;
    const int K0 = s->k0 + s->k1;
    const size_t ring = 3 * ((size_t)2 * SC_ROWS * 36 + (size_t)2 * s->n1 * 36);
    const size_t scratch = 4096 + (size_t)SC_ROWS * (s->n1 + 4) + (size_t)SC_ROWS * (s->n2 + 4) + 6 * 1024;
    const size_t ldsb = (ring > scratch ? ring : scratch) * sizeof(float);
    static size_t attr = 0;
    if (ldsb > attr) {
        FR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(scorer_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
        attr = ldsb;
    }
    prof_work(K_LINEAR_FWD, 2.0 * (double)B * ((double)K0 * s->n1 + (double)s->n1 * s->n2 + s->n2));
    ProfScope prof(K_LINEAR_FWD, stream);
    FR_LAUNCH(prof, scorer_fwd_kernel, dim3((unsigned)fr_scorer_blocks(B)), dim3(SC_THREADS), ldsb, stream, a);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_scorer_bwd(const fr_scorer* s, const float* dy, const float* gscale, const float* y, const float* h1,
                             const float* h2, int64_t B, const int64_t* used, float* dz1, float* dz2, float* dz3, float* dx0,
                             float* dx1, float* w3part, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FR_CHECK_ARG(shape_ok(s), "fr_scorer_bwd: shape not supported");
    FR_CHECK_ARG(dy && y && h1 && h2 && dz1 && dz2 && dz3 && w3part && B >= 1 && B < (1ll << 31) && s->W1 && s->W2 && s->W3,
                 "fr_scorer_bwd: bad argument");
    FR_CHECK_ARG(s->p == 0.f || used, "fr_scorer_bwd: dropout needs the counter value the forward used");
    FR_CHECK_ARG((((uintptr_t)h1 | (uintptr_t)h2 | (uintptr_t)dz1 | (uintptr_t)dz2 | (uintptr_t)dx0 | (uintptr_t)dx1 |
                   (uintptr_t)s->W3) & 15) == 0, "fr_scorer_bwd: 16-byte alignment required");
    ScBwd a{};
    fill(s, B, &a.s, &a.d);
    a.d.ctr = (const unsigned long long*)used;
    a.dy = dy; a.gscale = gscale; a.y = y; a.h1 = h1; a.h2 = h2;
    a.dz1 = dz1; a.dz2 = dz2; a.dz3 = dz3; a.dx0 = dx0; a.dx1 = dx1; a.w3part = w3part;
    const int kx = (dx0 ? s->k0 : 0) + (dx1 ? s->k1 : 0);
    const size_t ldsb = ((size_t)2 * SC_ROWS * (s->n2 + 4) + (size_t)SC_ROWS * (s->n1 + 4) + 4096 + (size_t)SC_ROWS * (kx + 4)) * sizeof(float);
    static size_t attr = 0;
    if (ldsb > attr) {
        FR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(scorer_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
        attr = ldsb;
    }
    prof_work(K_LINEAR_BWD_INPUT, 2.0 * (double)B * ((double)s->n2 * s->n1 + (double)s->n1 * kx + s->n2));
    ProfScope prof(K_LINEAR_BWD_INPUT, stream);
    FR_LAUNCH(prof, scorer_bwd_kernel, dim3((unsigned)fr_scorer_blocks(B)), dim3(SC_THREADS), ldsb, stream, a);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_parts_sum(const float* part, int32_t parts, int64_t n, float* out, void* stream_) {
    FR_CHECK_ARG(part && out && parts >= 1 && n >= 1 && n < (1ll << 31), "fr_parts_sum: bad argument");
    hipLaunchKernelGGL(parts_sum_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream_, part, (int)parts,
                       (int)n, out);
    FR_CHECK_LAUNCH();
    return FR_OK;
}
