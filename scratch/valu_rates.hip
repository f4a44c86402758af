// VALU issue / throughput rates on gfx950 as the replay loop sees them: W waves per SIMD, each running a chain of one
// instruction kind with ILP independent chains.  Prints SIMD cycles per wave-instruction.
//   hipcc -O3 --offload-arch=gfx950 scratch/valu_rates.hip -o scratch/bin/valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v2f __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int KIND, int ILP>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float x[ILP];
    v2f y[ILP];
    for (int i = 0; i < ILP; ++i) { x[i] = threadIdx.x * 1e-3f + i + 1.f; y[i] = v2f{x[i], x[i] + 0.5f}; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                if (KIND == 0) x[i] = fmaf(x[i], a, b);
                if (KIND == 1) x[i] = __builtin_amdgcn_sqrtf(x[i]);
                if (KIND == 2) x[i] = __builtin_amdgcn_rcpf(x[i]);
                if (KIND == 3) y[i] = __builtin_elementwise_fma(y[i], v2f{a, a}, v2f{b, b});
                if (KIND == 4) { x[i] = fmaf(x[i], a, b); x[i] = __builtin_amdgcn_sqrtf(x[i]); }      // fma -> sqrt dependent
                if (KIND == 5) x[i] = x[i] * a;
                if (KIND == 6) y[i] = y[i] * y[i];
                if (KIND == 7) x[i] = __builtin_amdgcn_rsqf(x[i]);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < ILP; ++i) s += x[i] + y[i].x + y[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND, int ILP>
void run(const char* name, int waves_per_simd, float* out) {
    const int iters = 2000;
    const int blocks = 256 * waves_per_simd;      // 256 CUs x (4 waves per block = 1 per SIMD) x waves_per_simd
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<KIND, ILP>), dim3(blocks), dim3(256), 0, 0, out, iters, 0.999f, 1e-3f);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double n_inst = (double)iters * 8 * ILP * (KIND == 4 ? 2 : 1) * waves_per_simd;   // wave-instructions per SIMD
    printf("%-22s ILP %d  waves/SIMD %d : %6.2f cycles per wave-instruction (2.4 GHz assumed)\n", name, ILP, waves_per_simd,
           best * 1e-3 * 2.4e9 / n_inst);
}

int main() {
    float* out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    for (int w : {1, 2, 4, 8}) {
        if (w == 1) { run<0, 1>("v_fma_f32", 1, out); run<0, 4>("v_fma_f32", 1, out); run<3, 1>("v_pk_fma_f32", 1, out); run<3, 4>("v_pk_fma_f32", 1, out);
                      run<1, 1>("v_sqrt_f32", 1, out); run<1, 4>("v_sqrt_f32", 1, out); run<2, 4>("v_rcp_f32", 1, out); run<4, 4>("fma->sqrt", 1, out); }
        if (w == 2) { run<0, 1>("v_fma_f32", 2, out); run<0, 4>("v_fma_f32", 2, out); run<3, 1>("v_pk_fma_f32", 2, out); run<3, 4>("v_pk_fma_f32", 2, out);
                      run<1, 1>("v_sqrt_f32", 2, out); run<1, 4>("v_sqrt_f32", 2, out); run<4, 4>("fma->sqrt", 2, out); }
        if (w == 4) { run<0, 1>("v_fma_f32", 4, out); run<0, 4>("v_fma_f32", 4, out); run<3, 1>("v_pk_fma_f32", 4, out); run<3, 4>("v_pk_fma_f32", 4, out);
                      run<1, 1>("v_sqrt_f32", 4, out); run<1, 4>("v_sqrt_f32", 4, out); run<4, 4>("fma->sqrt", 4, out); }
        if (w == 8) { run<0, 1>("v_fma_f32", 8, out); run<0, 4>("v_fma_f32", 8, out); run<3, 1>("v_pk_fma_f32", 8, out); run<3, 4>("v_pk_fma_f32", 8, out);
                      run<1, 1>("v_sqrt_f32", 8, out); run<1, 4>("v_sqrt_f32", 8, out); run<2, 1>("v_rcp_f32", 8, out); run<4, 4>("fma->sqrt", 8, out);
                      run<5, 4>("v_mul_f32", 8, out); run<6, 4>("v_pk_mul_f32", 8, out); run<7, 4>("v_rsq_f32", 8, out); }
    }
    return 0;
}
