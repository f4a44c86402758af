// Activation codes of MLPLayers (layers.py:88-118) and their forward / derivative-through-the-output forms, shared by the
// dense-layer translation units (mlp.hip, mlp_bn.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace fr {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_LEAKY = 2, ACT_SIGMOID = 3, ACT_TANH = 4 };

__device__ __forceinline__ float act_fwd(float x, int act) {
    switch (act) {
        case ACT_RELU: return x > 0.f ? x : 0.f;
        case ACT_LEAKY: return x > 0.f ? x : 0.01f * x;
        case ACT_SIGMOID: return 1.f / (1.f + __expf(-x));
        case ACT_TANH: return tanhf(x);
        default: return x;
    }
}

// derivative expressed through the OUTPUT y = act(x)
__device__ __forceinline__ float act_bwd(float y, int act) {
    switch (act) {
        case ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case ACT_LEAKY: return y > 0.f ? 1.f : 0.01f;
        case ACT_SIGMOID: return y * (1.f - y);
        case ACT_TANH: return 1.f - y * y;
        default: return 1.f;
    }
}

}  // namespace fr
