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

// Adam constants in the form the kernels consume.
struct AdamC {
    const float2* sc;  // per-step (step_size, 1/sqrt(bias_correction2)); entry 0 unused
    int cap;
    float wd, b1, omb1, b2, omb2, eps;
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

__device__ __forceinline__ float2 step_scalars(const AdamC& c, int j) {
    return c.sc[j < c.cap ? j : c.cap];
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

__device__ __forceinline__ float lane_bcast(float x, int src_lane /*wave-uniform*/) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), src_lane));
}

// Replay the zero-data-gradient steps (from, to] on a row fragment; `from`/`to` are wave-uniform.
// The per-step scalars are fetched 64 steps at a time (one coalesced load, lane q holds step base+q) and
// broadcast with v_readlane, so the dependent chain of a replayed step holds no memory access.
template <int E>
__device__ __forceinline__ void replay(RowFrag<E>& p, RowFrag<E>& m, RowFrag<E>& v, int from, int to, const AdamC& c,
                                       int lane) {
    for (int base = from + 1; base <= to; base += 64) {
        const float2 s = step_scalars(c, base + lane);
        const int n = (to - base + 1) < 64 ? (to - base + 1) : 64;
        for (int q = 0; q < n; ++q) {
            const float ss = lane_bcast(s.x, q), ib = lane_bcast(s.y, q);
#pragma unroll
            for (int e = 0; e < E; ++e) adam_elem(p.x[e], m.x[e], v.x[e], 0.f, ss, ib, c);
        }
    }
}

// Same for two rows at once (independent chains interleave in the VALU).
template <int E>
__device__ __forceinline__ void replay2(RowFrag<E>& p0, RowFrag<E>& m0, RowFrag<E>& v0, RowFrag<E>& p1,
                                        RowFrag<E>& m1, RowFrag<E>& v1, int from, int to, const AdamC& c, int lane) {
    for (int base = from + 1; base <= to; base += 64) {
        const float2 s = step_scalars(c, base + lane);
        const int n = (to - base + 1) < 64 ? (to - base + 1) : 64;
        for (int q = 0; q < n; ++q) {
            const float ss = lane_bcast(s.x, q), ib = lane_bcast(s.y, q);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                adam_elem(p0.x[e], m0.x[e], v0.x[e], 0.f, ss, ib, c);
                adam_elem(p1.x[e], m1.x[e], v1.x[e], 0.f, ss, ib, c);
            }
        }
    }
}

}  // namespace fr
