// The BatchNorm expressions the layered kernels (mlp.hip) and the one-launch-per-layer kernels (mlp_bn.hip) share, written
// with explicit fused / unfused operations: the two forms then round alike by construction (left to the compiler, `a * b + c`
// is fused or not depending on the code around it), which is what lets the tests hold one against the other bit for bit.
#pragma once
#include <hip/hip_runtime.h>

namespace fr {

// one row's contribution to a column's sums: s1 += dy s, s2 += (dy s) xhat        (s = act'(y))
__device__ __forceinline__ void bn_bwd_acc(float dy, float s, float xh, float& s1, float& s2) {
    s1 = fmaf(dy, s, s1);
    s2 = fmaf(__fmul_rn(dy, s), xh, s2);
}

// dz = invstd gamma (dy s - mean(dy s) - xhat mean(dy s xhat));  a1 = sum(dy s) / M, a2 = sum(dy s xhat) / M, isg = invstd * gamma
__device__ __forceinline__ float bn_bwd_dz(float dy, float s, float xh, float a1, float a2, float isg) {
    return __fmul_rn(isg, fmaf(-a2, xh, fmaf(dy, s, -a1)));
}

// running statistic <- (1 - momentum) old + momentum value
__device__ __forceinline__ float bn_running(float old, float momentum, float value) {
    return fmaf(momentum, value, __fmul_rn(1.f - momentum, old));
}

}  // namespace fr
