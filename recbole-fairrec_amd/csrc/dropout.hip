// Dropout as a pure function of (seed, call counter, element index): out = x * keep, keep = 0 with probability p and
// 1/(1-p) otherwise.  Nothing is stored: the backward pass regenerates the pattern of its forward call from the counter
// value that call used.  The counter lives in device memory and is advanced by the forward's last launch itself, so a
// captured step (hipGraph) draws a fresh pattern on every replay with no host involvement.
//
// Replaces, per MLP forward of the reference's `nn.Dropout` layers (recbole/model/layers.py:62-63), torch's
// rand / ge / cast / mul launches and a [B, sum of layer widths] fp32 mask read twice per step.  The pattern is NOT
// torch's (the reference's dropout stream is not reproducible across devices either); tests/ pin the arithmetic with
// recorded masks through MLPLayers.forced_masks and this generator through its own properties.
#include "kernels.hpp"
#include "dropout.hpp"

namespace fr {

// Up to two tensors per launch (the two blocks of a first layer's input): job 1 takes the workgroups from `blocks0` on.
struct DropJob {
    const float* x;
    float* out;
    long long n;
    unsigned long long off4;
};

__global__ __launch_bounds__(256) void dropout_apply_kernel(DropJob j0, DropJob j1, unsigned blocks0, unsigned thr,
                                                            float scale, unsigned long long seed,
                                                            const unsigned long long* __restrict__ ctr_src,
                                                            unsigned long long* __restrict__ used_out,
                                                            unsigned long long* __restrict__ tick) {
    __shared__ unsigned long long ctr_s;
    const unsigned long long ctr = drop_counter_enter(ctr_src, used_out, tick, &ctr_s);
    const bool second = blockIdx.x >= blocks0;
    const float* __restrict__ x = second ? j1.x : j0.x;
    float* __restrict__ out = second ? j1.out : j0.out;
    const long long n = second ? j1.n : j0.n;
    const unsigned long long off4 = second ? j1.off4 : j0.off4;
    const long long q = (long long)(blockIdx.x - (second ? blocks0 : 0u)) * 256 + threadIdx.x;   // this thread's 4 elements
    const long long i = q * 4;
    if (i >= n) return;
    const unsigned long long g = off4 + (unsigned long long)q;
    const float4 kp = drop_keep4(seed, ctr, g, thr, scale);
    const float k0 = kp.x, k1 = kp.y, k2 = kp.z, k3 = kp.w;
    if (i + 4 <= n) {
        const float4 v = *reinterpret_cast<const float4*>(x + i);
        *reinterpret_cast<float4*>(out + i) = make_float4(v.x * k0, v.y * k1, v.z * k2, v.w * k3);
    } else {
        const float k[4] = {k0, k1, k2, k3};
        for (int e = 0; i + e < n; ++e) out[i + e] = x[i + e] * k[e];
    }
}

}  // namespace fr

extern "C" int fr_dropout_apply2(const float* x0, int64_t n0, uint64_t offset0, float* out0, const float* x1, int64_t n1,
                                 uint64_t offset1, float* out1, float p, uint64_t seed, const int64_t* counter,
                                 int64_t* used_out, int64_t* tick_state, void* stream_) {
    using namespace fr;
    hipStream_t stream = (hipStream_t)stream_;
    FR_CHECK_ARG(x0 && out0 && counter && n0 >= 1 && p >= 0.f && p < 1.f && offset0 % 4 == 0, "fr_dropout_apply: bad argument");
    FR_CHECK_ARG((((uintptr_t)x0 | (uintptr_t)out0) & 15) == 0, "fr_dropout_apply: 16-byte alignment required");
    FR_CHECK_ARG(!x1 || (out1 && n1 >= 1 && offset1 % 4 == 0 && (((uintptr_t)x1 | (uintptr_t)out1) & 15) == 0),
                 "fr_dropout_apply2: bad second tensor");
    const unsigned thr = drop_threshold(p);
    const unsigned b0 = (unsigned)(((n0 + 3) / 4 + 255) / 256), b1 = x1 ? (unsigned)(((n1 + 3) / 4 + 255) / 256) : 0u;
    const DropJob j0{x0, out0, (long long)n0, (unsigned long long)(offset0 / 4)};
    const DropJob j1{x1, out1, (long long)(x1 ? n1 : 0), (unsigned long long)(offset1 / 4)};
    hipLaunchKernelGGL(dropout_apply_kernel, dim3(b0 + b1), dim3(256), 0, stream, j0, j1, b0, thr, 1.f / (1.f - p),
                       (unsigned long long)seed, (const unsigned long long*)counter, (unsigned long long*)used_out,
                       (unsigned long long*)tick_state);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_dropout_apply(const float* x, int64_t n, float p, uint64_t seed, uint64_t offset, const int64_t* counter,
                                int64_t* used_out, int64_t* tick_state, float* out, void* stream_) {
    return fr_dropout_apply2(x, n, offset, out, nullptr, 0, 0, nullptr, p, seed, counter, used_out, tick_state, stream_);
}

// ---- several small device-to-device copies in one launch ------------------------------------------------------------------
// A captured training step reads its batch from static tensors; refreshing them column by column costs one ~5 us copy
// launch per column (fairrec/graph.py).  One launch: workgroup b copies 4 KiB chunk (b - first[j]) of job j.
namespace fr {
struct CopyJobs {
    const unsigned char* src[FR_COPY_MAX];
    unsigned char* dst[FR_COPY_MAX];
    unsigned long long bytes[FR_COPY_MAX];
    unsigned first[FR_COPY_MAX + 1];     // first workgroup of job j
    int n;
};

__global__ __launch_bounds__(256) void copy_many_kernel(CopyJobs J) {
    int j = 0;
    while (j + 1 < J.n && blockIdx.x >= J.first[j + 1]) ++j;
    const unsigned long long base = (unsigned long long)(blockIdx.x - J.first[j]) * 4096ull;
    const unsigned char* s = J.src[j] + base;
    unsigned char* d = J.dst[j] + base;
    const unsigned long long left = J.bytes[j] - base;
    const unsigned len = left < 4096ull ? (unsigned)left : 4096u;
    if (((((uintptr_t)s) | ((uintptr_t)d)) & 15) == 0) {
        const unsigned o = threadIdx.x * 16;
        if (o + 16 <= len) *reinterpret_cast<uint4*>(d + o) = *reinterpret_cast<const uint4*>(s + o);
        else
            for (unsigned e = o; e < len; ++e) d[e] = s[e];
    } else {
        for (unsigned e = threadIdx.x; e < len; e += 256) d[e] = s[e];
    }
}
}  // namespace fr

extern "C" int fr_copy_many(const void* const* src, void* const* dst, const int64_t* bytes, int32_t n, void* stream_) {
    using namespace fr;
    hipStream_t stream = (hipStream_t)stream_;
    FR_CHECK_ARG(src && dst && bytes && n >= 1 && n <= FR_COPY_MAX, "fr_copy_many: bad argument (1..FR_COPY_MAX jobs)");
    CopyJobs J{};
    unsigned blocks = 0;
    int m = 0;
    for (int j = 0; j < n; ++j) {
        FR_CHECK_ARG(bytes[j] >= 0 && (bytes[j] == 0 || (src[j] && dst[j])), "fr_copy_many: bad job");
        if (bytes[j] == 0) continue;
        J.src[m] = (const unsigned char*)src[j];
        J.dst[m] = (unsigned char*)dst[j];
        J.bytes[m] = (unsigned long long)bytes[j];
        J.first[m] = blocks;
        blocks += (unsigned)((bytes[j] + 4095) / 4096);
        ++m;
    }
    if (m == 0) return FR_OK;
    J.first[m] = blocks;
    J.n = m;
    hipLaunchKernelGGL(copy_many_kernel, dim3(blocks), dim3(256), 0, stream, J);
    FR_CHECK_LAUNCH();
    return FR_OK;
}
