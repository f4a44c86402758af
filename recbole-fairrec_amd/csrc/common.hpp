// Shared host/device helpers for libfairrec_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "fairrec_hip.h"

#ifndef FR_ADAM_PRECISE
#define FR_ADAM_PRECISE 0
#endif

namespace fr {

constexpr int WAVE = 64;

// ---- host-side error plumbing -------------------------------------------------------------------
void set_error(const char* fmt, ...);

#define FR_CHECK_ARG(cond, ...)          \
    do {                                 \
        if (!(cond)) {                   \
            ::fr::set_error(__VA_ARGS__); \
            return FR_EINVAL;            \
        }                                \
    } while (0)

#define FR_CHECK_HIP(expr)                                                                          \
    do {                                                                                            \
        hipError_t e__ = (expr);                                                                    \
        if (e__ != hipSuccess) {                                                                    \
            ::fr::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return FR_EHIP;                                                                         \
        }                                                                                           \
    } while (0)

#define FR_CHECK_LAUNCH()  FR_CHECK_HIP(hipGetLastError())

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- device helpers -----------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;  // every lane holds the total; fixed butterfly order => deterministic
}

template <int W>
__device__ __forceinline__ float group_sum(float x) {  // sum over aligned groups of W lanes
#pragma unroll
    for (int o = W / 2; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

__device__ __forceinline__ int uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }

// Slot layout of an exchange buffer (fairrec_hip.h): logical row j sits at (j / chunk) * stride + j % chunk rows
// from the pointer; chunk == 0 is the dense layout.  Lets several tables share one [G, T, cap, ...] all-to-all
// buffer without copies.
struct Lay {
    int chunk, stride;
    __host__ __device__ __forceinline__ long long at(long long j) const {
        return chunk ? (j / chunk) * (long long)stride + (j % chunk) : j;
    }
};

// Adam constants in the form the kernels consume.
struct AdamC {
    const float2* sc;  // per-step (step_size, 1/sqrt(bias_correction2)); entry 0 unused
    int cap;
    float wd, b1, omb1, b2, omb2, eps;
    float k1, k2;  // (1-beta1)*wd and (1-beta2)*wd^2: a replayed step's gradient is wd*p
    float inv_k1, inv_k2;
};

inline AdamC make_adamc(const fr_adam* a) {
    AdamC c;
    c.sc = reinterpret_cast<const float2*>(a->scalars);
    c.cap = a->cap;
    c.wd = (float)a->weight_decay;
    c.b1 = (float)a->beta1;
    c.omb1 = (float)(1.0 - a->beta1);  // torch: lerp weight / addcmul value are Python doubles cast to fp32 once
    c.b2 = (float)a->beta2;
    c.omb2 = (float)(1.0 - a->beta2);
    c.eps = (float)a->eps;
    c.k1 = (float)((1.0 - a->beta1) * a->weight_decay);
    c.k2 = (float)((1.0 - a->beta2) * a->weight_decay * a->weight_decay);
    c.inv_k1 = a->weight_decay != 0.0 ? (float)(1.0 / ((1.0 - a->beta1) * a->weight_decay)) : 0.f;
    c.inv_k2 = a->weight_decay != 0.0 ? (float)(1.0 / ((1.0 - a->beta2) * a->weight_decay * a->weight_decay)) : 0.f;
    return c;
}

// One Adam step on one element, in the op order of torch/optim/adam.py::_single_tensor_adam
// (grad.add(param, alpha=wd); exp_avg.lerp_; exp_avg_sq.mul_().addcmul_(); sqrt/bc2_sqrt + eps; addcdiv_).
// `gd` is the data gradient (0 for a replayed step).  ss = lr/bc1, ib = 1/sqrt(bc2).
// v_sqrt_f32 / v_rcp_f32 are 1-ulp; the division by bc2_sqrt is a multiplication by its fp32 reciprocal.
__device__ __forceinline__ void adam_elem(float& p, float& m, float& v, float gd, float ss, float ib, const AdamC& c) {
    float g = fmaf(c.wd, p, gd);
    m = fmaf(c.omb1, g - m, m);
    v = fmaf(c.omb2 * g, g, v * c.b2);
#if FR_ADAM_PRECISE
    float den = fmaf(__fsqrt_rn(v), ib, c.eps);
    p = p + __fdiv_rn(-ss * m, den);
#else
    float den = fmaf(__builtin_amdgcn_sqrtf(v), ib, c.eps);
    p = fmaf(-ss * m, __builtin_amdgcn_rcpf(den), p);
#endif
}

// A replayed step: the data gradient is zero, so g = wd*p and
//   m <- beta1*m + (1-beta1)*wd*p ;  v <- beta2*v + (1-beta2)*wd^2*p^2   (8 VALU + sqrt + rcp per element).
__device__ __forceinline__ void adam_zero(float& p, float& m, float& v, float ss, float ib, const AdamC& c) {
    m = fmaf(c.k1, p, c.b1 * m);
    v = fmaf(c.k2 * p, p, c.b2 * v);
#if FR_ADAM_PRECISE
    float den = fmaf(__fsqrt_rn(v), ib, c.eps);
    p = p + __fdiv_rn(-ss * m, den);
#else
    float den = fmaf(__builtin_amdgcn_sqrtf(v), ib, c.eps);
    p = fmaf(-ss * m, __builtin_amdgcn_rcpf(den), p);
#endif
}

// The per-step scalar table is written by the host before any launch and never changes while kernels
// run: read it through the constant address space so that a wave-uniform index becomes an s_load
// (scalar cache, no VALU/VMEM work in the replay loop).
typedef const float __attribute__((address_space(4))) * ConstFPtr;

// Entry j of the table: (step_size, 1/sqrt(bc2), A, B); A and B drive the scaled replay below.
__device__ __forceinline__ float4 step_scalars4(const AdamC& c, int j) {
    ConstFPtr sc = (ConstFPtr)c.sc;
    const int k = 4 * (j < c.cap ? j : c.cap);
    return make_float4(sc[k], sc[k + 1], sc[k + 2], sc[k + 3]);
}

__device__ __forceinline__ float2 step_scalars(const AdamC& c, int j) {
    const float4 s = step_scalars4(c, j);
    return make_float2(s.x, s.y);
}

// The same replayed step on scaled moments m' = m/k1, v' = v/k2 (k1 = (1-beta1)*wd, k2 = (1-beta2)*wd^2):
//   m' <- beta1*m' + p ;  v' <- beta2*v' + p^2 ;  p <- p - m' / (sqrt(v')*A_j + B_j)
// with A_j = sqrt(k2)/(sqrt(bc2_j)*step_size_j*k1), B_j = eps/(step_size_j*k1) from the host table:
// 5 VALU + sqrt + rcp per element instead of 8 + 2.  Same mathematics, rounding differs by a few ulp.
__device__ __forceinline__ void adam_zero_scaled(float& p, float& ms, float& vs, float A, float Bc, const AdamC& c) {
    ms = fmaf(c.b1, ms, p);
    vs = fmaf(c.b2, vs, p * p);
    const float den = fmaf(__builtin_amdgcn_sqrtf(vs), A, Bc);
    p = fmaf(-ms, __builtin_amdgcn_rcpf(den), p);
}

// A row (or the part of it one lane owns): element e of lane l is column l + 64*e.
template <int E>
struct RowFrag {
    float x[E];
};

template <int E>
__device__ __forceinline__ void load_row(RowFrag<E>& f, const float* base, int D, int lane) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
        int d = lane + 64 * e;
        f.x[e] = d < D ? base[d] : 0.f;
    }
}

template <int E>
__device__ __forceinline__ void store_row(const RowFrag<E>& f, float* base, int D, int lane) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
        int d = lane + 64 * e;
        if (d < D) base[d] = f.x[e];
    }
}

// Replay the zero-data-gradient steps (from, to] on NR row fragments at once (independent chains
// interleave in the VALU).  `from`/`to` are wave-uniform; scalars of the next 4 steps are fetched
// (s_load) while the current 4 are applied.
typedef float float16_t_ __attribute__((ext_vector_type(16)));
typedef const float16_t_ __attribute__((address_space(4))) * ConstF16Ptr;

template <int E, int NR, bool SCALED>
__device__ __forceinline__ void replay_loop(RowFrag<E>* (&p)[NR], RowFrag<E>* (&m)[NR], RowFrag<E>* (&v)[NR], int j,
                                            int to, const AdamC& c) {
    auto one = [&](float s0, float s1, float s2, float s3) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if (SCALED) adam_zero_scaled(p[r]->x[e], m[r]->x[e], v[r]->x[e], s2, s3, c);
                else adam_zero(p[r]->x[e], m[r]->x[e], v[r]->x[e], s0, s1, c);
            }
        }
    };
    // main part: 4 steps per iteration, their 16 scalars fetched by ONE s_load_dwordx16 issued one iteration ahead
    const int lim = to < c.cap ? to : c.cap;
    if (j + 3 <= lim) {
        ConstF16Ptr tab = (ConstF16Ptr)(c.sc + 2 * j);   // c.sc is float2*: entry j starts at float 4*j
        float16_t_ cur = tab[0];
        for (; j + 7 <= lim; j += 4) {
            ++tab;
            const float16_t_ nxt = tab[0];
            one(cur[0], cur[1], cur[2], cur[3]);
            one(cur[4], cur[5], cur[6], cur[7]);
            one(cur[8], cur[9], cur[10], cur[11]);
            one(cur[12], cur[13], cur[14], cur[15]);
            cur = nxt;
        }
        one(cur[0], cur[1], cur[2], cur[3]);
        one(cur[4], cur[5], cur[6], cur[7]);
        one(cur[8], cur[9], cur[10], cur[11]);
        one(cur[12], cur[13], cur[14], cur[15]);
        j += 4;
    }
    // remainder (< 4 steps inside the table, and every step beyond `cap`, where the scalars are constant)
    for (; j <= to; ++j) {
        const float4 s = step_scalars4(c, j);
        one(s.x, s.y, s.z, s.w);
    }
}

// The scaled replay on N values per lane held as explicit 2-vectors: v_pk_fma_f32 / v_pk_mul_f32 do two elements per
// issue slot (the two transcendentals stay one element each), 4.5 instead of 7 VALU instructions per 64-element row and
// step when two rows (or two 64-column halves of one row) go through together.  Left to the SLP vectoriser the loop came
// out packing m with v of ONE row behind extra moves (7.2 instructions per row and step, measured with SQ_INSTS_VALU).
typedef float v2f_ __attribute__((ext_vector_type(2)));
typedef const v2f_ __attribute__((address_space(4))) * ConstF2Ptr;

template <int NP, bool ODD>
__device__ __forceinline__ void replay_scaled_step(v2f_ (&P)[NP ? NP : 1], v2f_ (&M)[NP ? NP : 1], v2f_ (&V)[NP ? NP : 1],
                                                   float& ps, float& ms, float& vs, float A, float Bc, float b1,
                                                   float b2) {
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        M[k] = __builtin_elementwise_fma(v2f_{b1, b1}, M[k], P[k]);
        V[k] = __builtin_elementwise_fma(v2f_{b2, b2}, V[k], P[k] * P[k]);
        const v2f_ sq = {__builtin_amdgcn_sqrtf(V[k].x), __builtin_amdgcn_sqrtf(V[k].y)};
        const v2f_ den = __builtin_elementwise_fma(sq, v2f_{A, A}, v2f_{Bc, Bc});
        const v2f_ r = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
        P[k] = __builtin_elementwise_fma(-M[k], r, P[k]);
    }
    if (ODD) {
        ms = fmaf(b1, ms, ps);
        vs = fmaf(b2, vs, ps * ps);
        const float den = fmaf(__builtin_amdgcn_sqrtf(vs), A, Bc);
        ps = fmaf(-ms, __builtin_amdgcn_rcpf(den), ps);
    }
}

template <int E, int NR>
__device__ __forceinline__ void replay_scaled(RowFrag<E>* (&p)[NR], RowFrag<E>* (&m)[NR], RowFrag<E>* (&v)[NR], int j,
                                              int to, const AdamC& c) {
    constexpr int N = E * NR, NP = N / 2;
    constexpr bool ODD = (N & 1) != 0;
    // element k of the flat list = (row k % NR, fragment k / NR): the two rows of a pair share a 2-vector
    v2f_ P[NP ? NP : 1], M[NP ? NP : 1], V[NP ? NP : 1];
    float ps = 0.f, ms = 0.f, vs = 0.f;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int a = 2 * k, b = 2 * k + 1;
        P[k] = v2f_{p[a % NR]->x[a / NR] , p[b % NR]->x[b / NR]};
        M[k] = v2f_{m[a % NR]->x[a / NR] * c.inv_k1, m[b % NR]->x[b / NR] * c.inv_k1};
        V[k] = v2f_{v[a % NR]->x[a / NR] * c.inv_k2, v[b % NR]->x[b / NR] * c.inv_k2};
    }
    if (ODD) {
        ps = p[(N - 1) % NR]->x[(N - 1) / NR];
        ms = m[(N - 1) % NR]->x[(N - 1) / NR] * c.inv_k1;
        vs = v[(N - 1) % NR]->x[(N - 1) / NR] * c.inv_k2;
    }
    auto one = [&](float A, float Bc) { replay_scaled_step<NP, ODD>(P, M, V, ps, ms, vs, A, Bc, c.b1, c.b2); };
    // main part: 4 steps per iteration; their (A, B) pairs are fetched (four s_load_dwordx2: 8 SGPRs, not the 16 of whole
    // table entries) one iteration ahead
    const int lim = to < c.cap ? to : c.cap;
    if (j + 3 <= lim) {
        ConstF2Ptr tab = (ConstF2Ptr)(c.sc + 2 * j) + 1;   // c.sc is float2*: entry j = floats 4j..4j+3, (A, B) = its 2nd half
        v2f_ c0 = tab[0], c1 = tab[2], c2 = tab[4], c3 = tab[6];
        for (; j + 7 <= lim; j += 4) {
            tab += 8;
            const v2f_ n0 = tab[0], n1 = tab[2], n2 = tab[4], n3 = tab[6];
            one(c0.x, c0.y);
            one(c1.x, c1.y);
            one(c2.x, c2.y);
            one(c3.x, c3.y);
            c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        }
        one(c0.x, c0.y);
        one(c1.x, c1.y);
        one(c2.x, c2.y);
        one(c3.x, c3.y);
        j += 4;
    }
    // remainder (< 4 steps inside the table, and every step beyond `cap`, where the scalars are constant)
    for (; j <= to; ++j) {
        const float4 s = step_scalars4(c, j);
        one(s.z, s.w);
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int a = 2 * k, b = 2 * k + 1;
        p[a % NR]->x[a / NR] = P[k].x;
        p[b % NR]->x[b / NR] = P[k].y;
        m[a % NR]->x[a / NR] = M[k].x * c.k1;
        m[b % NR]->x[b / NR] = M[k].y * c.k1;
        v[a % NR]->x[a / NR] = V[k].x * c.k2;
        v[b % NR]->x[b / NR] = V[k].y * c.k2;
    }
    if (ODD) {
        p[(N - 1) % NR]->x[(N - 1) / NR] = ps;
        m[(N - 1) % NR]->x[(N - 1) / NR] = ms * c.k1;
        v[(N - 1) % NR]->x[(N - 1) / NR] = vs * c.k2;
    }
}

template <int E, int NR>
__device__ __forceinline__ void replay_n(RowFrag<E>* (&p)[NR], RowFrag<E>* (&m)[NR], RowFrag<E>* (&v)[NR], int from,
                                         int to, const AdamC& c) {
    if (from >= to) return;
    if (!FR_ADAM_PRECISE && c.k1 != 0.f) {   // wave-uniform: weight decay on -> scaled moments
        replay_scaled<E, NR>(p, m, v, from + 1, to, c);
    } else {
        replay_loop<E, NR, false>(p, m, v, from + 1, to, c);
    }
}

template <int E>
__device__ __forceinline__ void replay(RowFrag<E>& p, RowFrag<E>& m, RowFrag<E>& v, int from, int to, const AdamC& c,
                                       int /*lane*/) {
    RowFrag<E>* pp[1] = {&p};
    RowFrag<E>* mm[1] = {&m};
    RowFrag<E>* vv[1] = {&v};
    replay_n<E, 1>(pp, mm, vv, from, to, c);
}

template <int E>
__device__ __forceinline__ void replay2(RowFrag<E>& p0, RowFrag<E>& m0, RowFrag<E>& v0, RowFrag<E>& p1,
                                        RowFrag<E>& m1, RowFrag<E>& v1, int from, int to, const AdamC& c,
                                        int /*lane*/) {
    RowFrag<E>* pp[2] = {&p0, &p1};
    RowFrag<E>* mm[2] = {&m0, &m1};
    RowFrag<E>* vv[2] = {&v0, &v1};
    replay_n<E, 2>(pp, mm, vv, from, to, c);
}

}  // namespace fr
