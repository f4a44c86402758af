// Dense layers, fast form: fp32 MFMA GEMMs whose operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4) in
// full 128-byte lines and LDS -> registers as MFMA fragments, one wave per 32 x 32 output tile and no workgroup
// barrier in the main loop.  Same products as mlp.hip (which keeps the general kernels: dropout masks, widths that are
// not multiples of 32, activations differentiated inside the backward kernels):
//   GL_FWD     Y[M,N]  = act(X[M,K] W[N,K]^T + b)           reduction over K
//   GL_BWD_IN  dX[M,K] = dY[M,N] W[N,K]                     reduction over N
//   GL_BWD_W   slab[s][N,K] = dY[rows of s, N]^T X[rows of s, K], bslab[s][N] = column sums of dY    reduction over M
// (layers.py:56-85 and its autograd; nfcf.py:40, pfcn_biasedmf.py:113-142.)
//
// Why this shape.  v_mfma_f32_32x32x2_f32 takes 64 cycles; measured (scratch/mfma_peak.hip) one wave per SIMD reaches
// 69 % of the 157 TFLOP/s peak with two accumulators, two waves per SIMD 83 %, four 94 %: a [8192, 256] -> 128 layer
// has only 1024 tiles of 32 x 32, so the tiles' reductions are split between waves until every SIMD has two.  What
// bounded the earlier kernels was operand delivery: fragment-shaped global loads (32 rows x 32 bytes per instruction)
// cost four times the L1 tag look-ups of full lines, and LDS staging through registers costs a ds_write pass.  Here
// a stage instruction moves 8 rows x 128 bytes straight into a wave-private, XOR-swizzled LDS image; a fragment is
// one ds_read_b128 (4 consecutive reduction elements of a row: the reduction index is permuted consistently on both
// operands) or, for an operand whose reduction index runs along its rows, 4 ds_read_b32 (conflict-free).
// The LDS reads are inline asm: next to an LDS-DMA in flight the compiler waits vmcnt(0) before any LDS read it
// can see, which would drain the prefetch every chunk.
#include "common.hpp"
#include "kernels.hpp"
#include "mlp_glds.hpp"
#include "dropout.hpp"
#include "mlp_bn_math.hpp"

namespace fr {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;

constexpr int GL_WAVES = 8, GL_NBUF = 2, GL_STAGE_FLOATS = 2048;   // per wave and stage: A block + B block, 32 x 32 floats each

// the 128-byte piece [col0, col0 + 32) of row `row` of a matrix that may be two row-major blocks side by side
// derivative of an activation expressed through its OUTPUT y (csrc/mlp.hip: act_bwd; codes 1 relu, 2 leakyrelu, 3 sigmoid, 4 tanh)
__device__ __forceinline__ float gl_act_bwd(float y, int act) {
    switch (act) {
        case 1: return y > 0.f ? 1.f : 0.f;
        case 2: return y > 0.f ? 1.f : 0.01f;
        case 3: return y * (1.f - y);
        case 4: return 1.f - y * y;
        default: return 1.f;
    }
}

// the value lane J of every aligned group of four lanes holds, in all four (DPP quad_perm: no LDS traffic)
template <int J>
__device__ __forceinline__ float quad_bcast(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), J | (J << 2) | (J << 4) | (J << 6),
                                                                 0xf, 0xf, true));
}
__device__ __forceinline__ float quad_bcast_j(float v, int j) {      // j is a constant after unrolling
    switch (j) {
        case 0: return quad_bcast<0>(v);
        case 1: return quad_bcast<1>(v);
        case 2: return quad_bcast<2>(v);
        default: return quad_bcast<3>(v);
    }
}

__device__ __forceinline__ const float* gl_piece(const GlMat& m, long long row, int col0) {
    return col0 < m.split ? m.a + row * m.lda + col0 : m.b + row * m.ldb + (col0 - m.split);
}

template <int MODE, bool KSPLIT>   // KSPLIT: several waves of the workgroup share a tile's reduction (g.ks > 1)
__global__ __launch_bounds__(GL_WAVES * 64) void linear_glds_kernel(GlArgs g) {
    extern __shared__ __align__(16) float lds[];   // [GL_WAVES][GL_NBUF][2][1024]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // known uniform: tile, part and trip counts in SGPRs
    constexpr bool A_T = MODE == GL_BWD_W, B_T = MODE != GL_FWD;   // operand whose reduction index runs along its rows

    // ---- which tile, which part of the reduction ------------------------------------------------------------------
    const int ntiles = g.tiles_i * g.tiles_j;
    int tile, part;
    if (MODE == GL_BWD_W) {
        const long long id = (long long)blockIdx.x * GL_WAVES + wave;
        tile = (int)(id % ntiles);
        part = (int)(id / ntiles);
    } else if (KSPLIT) {
        const int tpb = GL_WAVES / g.ks;
        tile = blockIdx.x * tpb + wave % tpb;
        part = wave / tpb;
    } else {
        tile = blockIdx.x * GL_WAVES + wave;
        part = 0;
    }
    const bool live = tile < ntiles && (MODE != GL_BWD_W || part < g.parts);
    const int ti = live ? tile / g.tiles_j : 0, tj = live ? tile % g.tiles_j : 0;
    const int i0 = ti * 32, j0 = tj * 32;
    const int c0 = part * g.chunks_per_part;
    int nchunk = 0;
    if (live) {
        const int total = (g.R + 31) / 32;
        nchunk = total - c0 < g.chunks_per_part ? total - c0 : g.chunks_per_part;
        if (nchunk < 0) nchunk = 0;
    }

    // ---- staging: instruction i of a block brings rows 8i .. 8i+7; lane l -> row 8i + l/8, LDS slot l % 8, and the
    // 16-byte chunk it fetches is slot ^ ((row >> 1) & 7) (the image is lane-linear; the swizzle sits in the source) ----
    float* my = lds + (size_t)wave * GL_NBUF * GL_STAGE_FLOATS;
    const int srow = lane >> 3, sslot = lane & 7;
    // Source pointers of the 4 + 4 stage instructions, kept running from chunk to chunk (a row-contiguous operand moves 32
    // floats along its rows, a transposed one 32 rows down); computed in full -- row clamps, which of two side-by-side
    // blocks -- only for a part's first chunk, where a row-contiguous operand crosses into its second block, and for
    // the last (partial) chunk of a transposed operand.
    const float *pa[4], *pb[4];
    auto set_ptrs = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 8 * i + srow, c = (sslot ^ ((row >> 1) & 7)) * 4;
            if (A_T) {   // rows = reduction index, columns = the tile's output rows
                long long rr = (long long)chunk * 32 + row;
                rr = rr < g.R ? rr : g.R - 1;
                pa[i] = gl_piece(g.A, rr, i0) + c;
            } else {
                long long rr = i0 + row;
                rr = rr < g.rowsA ? rr : g.rowsA - 1;
                pa[i] = gl_piece(g.A, rr, chunk * 32) + c;
            }
            if (B_T) {
                long long rr = (long long)chunk * 32 + row;
                rr = rr < g.R ? rr : g.R - 1;
                pb[i] = gl_piece(g.B, rr, j0) + c;
            } else {
                long long rr = j0 + row;
                rr = rr < g.rowsB ? rr : g.rowsB - 1;
                pb[i] = gl_piece(g.B, rr, chunk * 32) + c;
            }
        }
    };
    const long long stepA = A_T ? 32ll * (i0 < g.A.split ? g.A.lda : g.A.ldb) : 32;
    const long long stepB = B_T ? 32ll * (j0 < g.B.split ? g.B.lda : g.B.ldb) : 32;
    auto stage = [&](int chunk, int buf) {
        float* ab = my + buf * GL_STAGE_FLOATS;
        float* bb = ab + 1024;
        const bool fresh = chunk == c0 || (!A_T && chunk * 32 == g.A.split) || (!B_T && chunk * 32 == g.B.split) ||
                           ((A_T || B_T) && (long long)chunk * 32 + 32 > g.R);
        if (fresh) set_ptrs(chunk);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((glb_vp)pa[i], (lds_vp)(ab + i * 256), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_vp)pb[i], (lds_vp)(bb + i * 256), 16, 0, 0);
            pa[i] += stepA;
            pb[i] += stepB;
        }
    };

    // ---- fragment addresses ----------------------------------------------------------------------------------------
    // MFMA step q = 4 j + e of a chunk multiplies reduction elements 8 j + 4 h + e (h = lane / 32)
    const int r = lane & 31, h = lane >> 5;
    const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) float*)my;
    unsigned rn[4];    // row-contiguous operand: row r, chunk slot 2 j + h
#pragma unroll
    for (int j = 0; j < 4; ++j) rn[j] = lbase + r * 128 + (((2 * j + h) ^ ((r >> 1) & 7)) << 4);
    unsigned rt[16];   // transposed operand: row 8 j + 4 h + e, float r
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int row = 8 * (q >> 2) + 4 * h + (q & 3);
        rt[q] = lbase + row * 128 + ((((r >> 2) ^ ((row >> 1) & 7)) << 4) | ((r & 3) << 2));
    }

    if (nchunk > 0) stage(c0, 0);
    f32x16 acc0 = {0}, acc1 = {0};
    float bsum = 0.f;
    for (int t0 = 0; t0 < nchunk; t0 += GL_NBUF) {
#pragma unroll
        for (int buf = 0; buf < GL_NBUF; ++buf) {
            const int t = t0 + buf;
            if (t < nchunk) {
                if (t + 1 < nchunk) {   // the next chunk into the buffer consumed one step ago, then wait for this one
                    stage(c0 + t + 1, buf ^ 1);
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                // (a fragment register must not be touched before the wait below: the compiler takes an asm output for
                // available at once, so the vectors go through the wait whole and are taken apart after it)
                f4 av[4], bv[4];
                float at[16], bt[16];
                if (!A_T) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(av[j]) : "v"(rn[j]), "n"(buf * 8192));
                } else {
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(at[q]) : "v"(rt[q]), "n"(buf * 8192));
                }
                if (!B_T) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bv[j]) : "v"(rn[j]), "n"(buf * 8192 + 4096));
                } else {
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(bt[q]) : "v"(rt[q]), "n"(buf * 8192 + 4096));
                }
                // the values exist after this wait (the operand lists tie the fragments to it)
                if (!A_T) {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)"
                                 : "+v"(at[0]), "+v"(at[1]), "+v"(at[2]), "+v"(at[3]), "+v"(at[4]), "+v"(at[5]), "+v"(at[6]),
                                   "+v"(at[7]), "+v"(at[8]), "+v"(at[9]), "+v"(at[10]), "+v"(at[11]), "+v"(at[12]), "+v"(at[13]),
                                   "+v"(at[14]), "+v"(at[15]));
                }
                if (!B_T) {
                    asm volatile("" : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]));
                } else {
                    asm volatile(""
                                 : "+v"(bt[0]), "+v"(bt[1]), "+v"(bt[2]), "+v"(bt[3]), "+v"(bt[4]), "+v"(bt[5]), "+v"(bt[6]),
                                   "+v"(bt[7]), "+v"(bt[8]), "+v"(bt[9]), "+v"(bt[10]), "+v"(bt[11]), "+v"(bt[12]), "+v"(bt[13]),
                                   "+v"(bt[14]), "+v"(bt[15]));
                }
                float a[16], b[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    a[q] = A_T ? at[q] : av[q >> 2][q & 3];
                    b[q] = B_T ? bt[q] : bv[q >> 2][q & 3];
                }
                if (A_T) {   // the reduction runs over batch rows: the last chunk may reach past them
                    const long long red0 = (long long)(c0 + t) * 32;
                    if (red0 + 32 > g.R) {
#pragma unroll
                        for (int q = 0; q < 16; ++q)
                            if (red0 + 8 * (q >> 2) + 4 * h + (q & 3) >= g.R) a[q] = 0.f;
                    }
                    if (g.bslab) {
#pragma unroll
                        for (int q = 0; q < 16; ++q) bsum += a[q];
                    }
                }
#pragma unroll
                for (int q = 0; q < 16; q += 2) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[q], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q + 1], b[q + 1], acc1, 0, 0, 0);
                }
            }
        }
    }
    f32x16 acc = acc0 + acc1;

    if (MODE == GL_BWD_W) {
        if (!live) return;
        float* out = g.slab + (size_t)part * g.out_rows * g.out_cols;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = i0 + (e & 3) + 8 * (e >> 2) + 4 * h;
            out[(size_t)row * g.out_cols + j0 + r] = acc[e];
        }
        if (g.bslab && tj == 0) {
            bsum += __shfl_xor(bsum, 32, 64);
            if (h == 0) g.bslab[(size_t)part * g.out_rows + i0 + r] = bsum;
        }
        return;
    }

    // ---- the parts of a tile's reduction meet in LDS (the staging buffers are free now) ----------------------------
    if (KSPLIT) {
        const int tpb = GL_WAVES / g.ks;
        __syncthreads();
        float* red = lds + (size_t)(wave % tpb) * 1024 * (g.ks - 1);
        if (part > 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) red[(part - 1) * 1024 + e * 64 + lane] = acc[e];
        }
        __syncthreads();
        if (part > 0) return;
        for (int p = 0; p < g.ks - 1; ++p) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] += red[p * 1024 + e * 64 + lane];
        }
    }
    if (!live) return;
    const int col = j0 + r;
    if (MODE == GL_FWD) {
        // (a short reduction makes the epilogue a visible share of a wave's life: one uniform branch per wave, then
        // straight-line code -- a switch per element cost 20 % of the kernel at K = 128)
        if (col < g.out_cols) {
            const float bias = g.bias ? g.bias[col] : 0.f;
            float* yp = g.Y + (size_t)(i0 + 4 * h) * g.out_cols + col;
            if (g.act <= 2) {   // none, relu, leakyrelu: the slope of the negative side
                const float neg = g.act == 0 ? 1.f : (g.act == 1 ? 0.f : 0.01f);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int ro = (e & 3) + 8 * (e >> 2);
                    const float v = acc[e] + bias;
                    if (i0 + 4 * h + ro < g.out_rows) yp[(size_t)ro * g.out_cols] = v > 0.f ? v : v * neg;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int ro = (e & 3) + 8 * (e >> 2);
                    const float v = acc[e] + bias;
                    if (i0 + 4 * h + ro < g.out_rows)
                        yp[(size_t)ro * g.out_cols] = g.act == 3 ? 1.f / (1.f + __expf(-v)) : tanhf(v);
                }
            }
        }
    } else {   // GL_BWD_IN: the input gradient may be two blocks side by side
        float* base = col < g.o_split ? g.o_a + col : g.o_b + (col - g.o_split);
        const int ld = col < g.o_split ? g.o_lda : g.o_ldb;
        if (g.relu_src) {   // on through the dropped ReLU that produced this layer's input (one uniform branch per wave)
            const float* src = g.relu_src + col;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = i0 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (row < g.out_rows) {
                    const float y = src[(size_t)row * g.out_cols];
                    base[(size_t)row * ld] = g.src_act ? acc[e] * gl_act_bwd(y, g.src_act) : (y > 0.f ? acc[e] * g.relu_scale : 0.f);
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = i0 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (row < g.out_rows) base[(size_t)row * ld] = acc[e];
            }
        }
    }
}

__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ Y, int act,
                                                      float scale, long long n4, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const f4 d = reinterpret_cast<const f4*>(dY)[i], y = reinterpret_cast<const f4*>(Y)[i];
    f4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float s;
        switch (act) {
            case 1: s = y[e] > 0.f ? 1.f : 0.f; break;
            case 2: s = y[e] > 0.f ? 1.f : 0.01f; break;
            case 3: s = y[e] * (1.f - y[e]); break;
            case 4: s = 1.f - y[e] * y[e]; break;
            default: s = 1.f;
        }
        o[e] = d[e] * (s * scale);
    }
    reinterpret_cast<f4*>(out)[i] = o;
}

// =====================================================================================================================
// The same three products with the operand chunks SHARED by the four waves of a 64 x 64 macro tile (2 x 2 tiles of
// 32 x 32): per 32-element reduction chunk the group brings two 32-row blocks of A and two of B -- each wave stages ONE of
// the four -- for four tiles' worth of MFMAs, half the L2 -> LDS bytes per flop of the wave-private form above
// (scratch/gemm_glds64.hip: 15-20 % faster from [8192,128] -> 128 to [1.1 M,128] -> 128).  One s_barrier per chunk: a wave
// waits for its own block of chunk t, the barrier makes every block of chunk t visible AND proves that all four waves are
// done reading the buffer of chunk t - 1, into which the next stage then goes (ring of NBUF buffers).  Every output
// element is accumulated in exactly the order of the wave-private kernel (same parts, same chunk order, same two
// accumulators), so the two forms are bit-identical (tests/test_mlp_hip.py).  KS reduction parts of a macro tile = KS
// groups of four waves in one workgroup.
template <int MODE, int KS, int NBUF>
__device__ __forceinline__ void glds64_body(const GlArgs& g, const unsigned bid, float* lds) {   // lds: [KS][NBUF][A0 A1 B0 B1][1024]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr bool A_T = MODE == GL_BWD_W, B_T = MODE != GL_FWD;
    const int w4 = wave & 3, wi = w4 >> 1, wj = w4 & 1;
    const int mt_j = (g.tiles_j + 1) / 2, nmt = ((g.tiles_i + 1) / 2) * mt_j;
    int mt, part;
    if (MODE == GL_BWD_W) {
        mt = (int)(bid % (unsigned)nmt);
        part = (int)(bid / (unsigned)nmt);
    } else {
        mt = bid;
        part = wave >> 2;
    }
    const int mi = mt / mt_j, mj = mt % mt_j;
    const int ti = 2 * mi + wi, tj = 2 * mj + wj;
    const bool live = ti < g.tiles_i && tj < g.tiles_j;
    const int i0 = ti * 32, j0 = tj * 32;
    const int c0 = part * g.chunks_per_part;
    int nchunk;
    {
        const int total = (g.R + 31) / 32;
        nchunk = total - c0 < g.chunks_per_part ? total - c0 : g.chunks_per_part;
        if (nchunk < 0) nchunk = 0;
    }
    const int nloop = KS > 1 ? g.chunks_per_part : nchunk;      // every group of the workgroup meets at every barrier

    // ---- staging: this wave brings block w4 of the group's four (0, 1: the two 32-row blocks of A; 2, 3: of B); a block
    // past the matrix (odd tile count) repeats the last one ---------------------------------------------------------------
    float* grp = lds + (size_t)(KS > 1 ? part : 0) * NBUF * 4096;
    const bool stA = w4 < 2;
    int s0;   // first output row (A) / column (B) of the staged block
    if (stA) {
        const int t = 2 * mi + w4;
        s0 = 32 * (t < g.tiles_i ? t : g.tiles_i - 1);
    } else {
        const int t = 2 * mj + (w4 - 2);
        s0 = 32 * (t < g.tiles_j ? t : g.tiles_j - 1);
    }
    const int srow = lane >> 3, sslot = lane & 7;
    const float* ps[4];
    auto set_ptrs = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 8 * i + srow, c = (sslot ^ ((row >> 1) & 7)) * 4;
            if (stA) {
                if (A_T) {
                    long long rr = (long long)chunk * 32 + row;
                    rr = rr < g.R ? rr : g.R - 1;
                    ps[i] = gl_piece(g.A, rr, s0) + c;
                } else {
                    long long rr = s0 + row;
                    rr = rr < g.rowsA ? rr : g.rowsA - 1;
                    ps[i] = gl_piece(g.A, rr, chunk * 32) + c;
                }
            } else {
                if (B_T) {
                    long long rr = (long long)chunk * 32 + row;
                    rr = rr < g.R ? rr : g.R - 1;
                    ps[i] = gl_piece(g.B, rr, s0) + c;
                } else {
                    long long rr = s0 + row;
                    rr = rr < g.rowsB ? rr : g.rowsB - 1;
                    ps[i] = gl_piece(g.B, rr, chunk * 32) + c;
                }
            }
        }
    };
    const bool sT = stA ? A_T : B_T;
    const GlMat& sm = stA ? g.A : g.B;
    const long long step = sT ? 32ll * (s0 < sm.split ? sm.lda : sm.ldb) : 32;
    auto stage = [&](int chunk, int buf) {
        float* dst = grp + buf * 4096 + w4 * 1024;
        const bool fresh = chunk == c0 || (!sT && chunk * 32 == sm.split) || (sT && (long long)chunk * 32 + 32 > g.R);
        if (fresh) set_ptrs(chunk);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((glb_vp)ps[i], (lds_vp)(dst + i * 256), 16, 0, 0);
            ps[i] += step;
        }
    };

    // ---- fragment addresses: A from block wi, B from block 2 + wj of the current buffer ---------------------------------
    const int r = lane & 31, h = lane >> 5;
    const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) float*)grp;
    const unsigned abase = lbase + wi * 4096, bbase = lbase + (2 + wj) * 4096;
    unsigned rn[4];    // row-contiguous operand: row r, chunk slot 2 j + h (offset inside a block)
#pragma unroll
    for (int j = 0; j < 4; ++j) rn[j] = r * 128 + (((2 * j + h) ^ ((r >> 1) & 7)) << 4);
    unsigned rt[16];   // transposed operand: row 8 j + 4 h + e, float r
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int row = 8 * (q >> 2) + 4 * h + (q & 3);
        rt[q] = row * 128 + ((((r >> 2) ^ ((row >> 1) & 7)) << 4) | ((r & 3) << 2));
    }
    unsigned aaddr[A_T ? 16 : 4], baddr[B_T ? 16 : 4];
#pragma unroll
    for (int q = 0; q < (A_T ? 16 : 4); ++q) aaddr[q] = abase + (A_T ? rt[q] : rn[q]);
#pragma unroll
    for (int q = 0; q < (B_T ? 16 : 4); ++q) baddr[q] = bbase + (B_T ? rt[q] : rn[q]);

#pragma unroll
    for (int b = 0; b < NBUF - 1; ++b)
        if (b < nchunk) stage(c0 + b, b);
    f32x16 acc0 = {0}, acc1 = {0};
    float bsum = 0.f;
    for (int t0 = 0; t0 < nloop; t0 += NBUF) {
#pragma unroll
        for (int buf = 0; buf < NBUF; ++buf) {
            const int t = t0 + buf;
            if (t < nloop) {
                // own block of chunk t has landed (newer chunks stay in flight) ...
                const int newer = nchunk - 1 - t < NBUF - 2 ? nchunk - 1 - t : NBUF - 2;
                if (NBUF >= 4 && newer >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (NBUF >= 3 && newer == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                // ... and everyone's; all waves are also done with the buffer of chunk t - 1, which the next stage reuses
                __builtin_amdgcn_s_barrier();
                if (t + NBUF - 1 < nchunk) stage(c0 + t + NBUF - 1, (buf + NBUF - 1) % NBUF);
                if (t < nchunk) {
                    f4 av[4], bv[4];
                    float at[16], bt[16];
                    if (!A_T) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(av[j]) : "v"(aaddr[j]), "n"(buf * 16384));
                    } else {
#pragma unroll
                        for (int q = 0; q < 16; ++q)
                            asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(at[q]) : "v"(aaddr[q]), "n"(buf * 16384));
                    }
                    if (!B_T) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bv[j]) : "v"(baddr[j]), "n"(buf * 16384));
                    } else {
#pragma unroll
                        for (int q = 0; q < 16; ++q)
                            asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(bt[q]) : "v"(baddr[q]), "n"(buf * 16384));
                    }
                    if (!A_T) {
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]));
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)"
                                     : "+v"(at[0]), "+v"(at[1]), "+v"(at[2]), "+v"(at[3]), "+v"(at[4]), "+v"(at[5]), "+v"(at[6]),
                                       "+v"(at[7]), "+v"(at[8]), "+v"(at[9]), "+v"(at[10]), "+v"(at[11]), "+v"(at[12]),
                                       "+v"(at[13]), "+v"(at[14]), "+v"(at[15]));
                    }
                    if (!B_T) {
                        asm volatile("" : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]));
                    } else {
                        asm volatile(""
                                     : "+v"(bt[0]), "+v"(bt[1]), "+v"(bt[2]), "+v"(bt[3]), "+v"(bt[4]), "+v"(bt[5]), "+v"(bt[6]),
                                       "+v"(bt[7]), "+v"(bt[8]), "+v"(bt[9]), "+v"(bt[10]), "+v"(bt[11]), "+v"(bt[12]),
                                       "+v"(bt[13]), "+v"(bt[14]), "+v"(bt[15]));
                    }
                    float a[16], b[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        a[q] = A_T ? at[q] : av[q >> 2][q & 3];
                        b[q] = B_T ? bt[q] : bv[q >> 2][q & 3];
                    }
                    if (A_T) {   // the reduction runs over batch rows: the last chunk may reach past them
                        const long long red0 = (long long)(c0 + t) * 32;
                        if (red0 + 32 > g.R) {
#pragma unroll
                            for (int q = 0; q < 16; ++q)
                                if (red0 + 8 * (q >> 2) + 4 * h + (q & 3) >= g.R) a[q] = 0.f;
                        }
                        if (g.bslab) {
#pragma unroll
                            for (int q = 0; q < 16; ++q) bsum += a[q];
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 16; q += 2) {
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[q], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q + 1], b[q + 1], acc1, 0, 0, 0);
                    }
                }
            }
        }
    }
    f32x16 acc = acc0 + acc1;

    if (MODE == GL_BWD_W) {
        if (!live) return;
        float* out = g.slab + (size_t)part * g.out_rows * g.out_cols;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = i0 + (e & 3) + 8 * (e >> 2) + 4 * h;
            out[(size_t)row * g.out_cols + j0 + r] = acc[e];
        }
        if (g.bslab && tj == 0) {
            bsum += __shfl_xor(bsum, 32, 64);
            if (h == 0) g.bslab[(size_t)part * g.out_rows + i0 + r] = bsum;
        }
        return;
    }

    if (KS > 1) {   // the parts of a tile's reduction meet in LDS (the staging buffers are free after the barrier)
        __syncthreads();
        float* red = lds + (size_t)w4 * 1024 * (KS - 1);
        if (part > 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) red[(part - 1) * 1024 + e * 64 + lane] = acc[e];
        }
        __syncthreads();
        if (part > 0) return;
        for (int p = 0; p < KS - 1; ++p) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] += red[p * 1024 + e * 64 + lane];
        }
    }
    if (!live) return;
    const int col = j0 + r;
    if (MODE == GL_FWD) {
        if (col < g.out_cols) {
            const float bias = g.bias ? g.bias[col] : 0.f;
            float* yp = g.Y + (size_t)(i0 + 4 * h) * g.out_cols + col;
            if (g.bn_part) {
                // BatchNorm behind this layer: the column statistics of this tile's 32 rows -- mean, then the sum of squared
                // deviations from it, as bn_fwd_stats_kernel forms them for a 32-row chunk -- from the accumulators, so that
                // the statistics launch (and its pass over Z) is not needed.  Lane (r, h) holds 16 rows of column r.
                float sum = 0.f, cnt = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int ro = (e & 3) + 8 * (e >> 2);
                    if (i0 + 4 * h + ro < g.out_rows) {
                        sum += acc[e] + bias;
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
                    if (i0 + 4 * h + ro < g.out_rows) {
                        const float d = (acc[e] + bias) - mean;
                        m2 = fmaf(d, d, m2);
                    }
                }
                m2 += __shfl_xor(m2, 32, 64);
                if (h == 0) {
                    g.bn_part[((size_t)ti * g.out_cols + col) * 2] = mean;
                    g.bn_part[((size_t)ti * g.out_cols + col) * 2 + 1] = m2;
                }
            }
            if (g.act <= 2) {
                const float neg = g.act == 0 ? 1.f : (g.act == 1 ? 0.f : 0.01f);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int ro = (e & 3) + 8 * (e >> 2);
                    const float v = acc[e] + bias;
                    if (i0 + 4 * h + ro < g.out_rows) yp[(size_t)ro * g.out_cols] = v > 0.f ? v : v * neg;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int ro = (e & 3) + 8 * (e >> 2);
                    const float v = acc[e] + bias;
                    if (i0 + 4 * h + ro < g.out_rows)
                        yp[(size_t)ro * g.out_cols] = g.act == 3 ? 1.f / (1.f + __expf(-v)) : tanhf(v);
                }
            }
        }
    } else {
        float* base = col < g.o_split ? g.o_a + col : g.o_b + (col - g.o_split);
        const int ld = col < g.o_split ? g.o_lda : g.o_ldb;
        if constexpr (MODE == GL_BWD_IN_BN) {
            // (see GlArgs) the tile through the dropout's keep pattern, stored; then the BatchNorm layer's backward sums over
            // the tile's 32 rows, ADDED IN bn_bwd_stats_kernel's ORDER so that the sums are its bits: that kernel's wave w
            // walks rows w, w + 4, ..., w + 28 of a 32-row chunk and the four waves' sums are added 0, 1, 2, 3.  Lane (r, h)
            // holds rows (e & 3) + 8 (e >> 2) + 4 h of column r, so chain w alternates between the two lanes of a column:
            // each lane gets its partner's terms by a shuffle and both walk all four chains.  ld == out_cols (one block).
            const unsigned long long ctr = g.dr_on ? *g.dr_used : 0ull;
            const float* yb = g.bnb_y + col;
            const float* xb = g.bnb_xhat + col;
            // The keep factors: one Philox call covers four neighbouring columns of a row, i.e. the four lanes of a quad.  Lane q
            // of a quad makes the calls of the quad's rows w + 8 q + 4 h (w = 0..3) -- four calls per lane instead of sixteen,
            // v_mul_hi_u32 being quarter rate -- and the others take their column's component from it below.
            // (requested before the Philox arithmetic, which then runs under their latency: nothing else does in this epilogue)
            float ya[16], xa[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = i0 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const int rc = row < g.out_rows ? row : g.out_rows - 1;
                ya[e] = yb[(size_t)rc * g.out_cols];
                xa[e] = xb[(size_t)rc * g.out_cols];
            }
            float4 kq[4];
            if (g.dr_on) {
                const int q = r & 3;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const size_t i4 = ((size_t)(i0 + w + 8 * q + 4 * h) * g.out_cols + (col & ~3)) >> 2;
                    kq[w] = drop_keep4(g.dr_seed, ctr, g.dr_off4 + (unsigned long long)i4, g.dr_thr, g.dr_scale);
                }
            }
            float c1[4] = {0.f, 0.f, 0.f, 0.f}, c2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const int e = w + 4 * j;
                    const int row = i0 + w + 8 * j + 4 * h;
                    float keep = 1.f;
                    if (g.dr_on) {      // (uniform) quad lane j holds this row's call
                        const float k0 = quad_bcast_j(kq[w].x, j), k1 = quad_bcast_j(kq[w].y, j), k2 = quad_bcast_j(kq[w].z, j),
                                    k3 = quad_bcast_j(kq[w].w, j);
                        const int q = r & 3;
                        keep = q == 0 ? k0 : (q == 1 ? k1 : (q == 2 ? k2 : k3));
                    }
                    float d = 0.f, sa = 0.f, xh = 0.f;      // (a row past the batch adds exact zeros: the chain skips it)
                    if (row < g.out_rows) {
                        d = acc[e];
                        if (g.dr_on) d = d * keep;
                        base[(size_t)row * ld] = d;
                        sa = gl_act_bwd(ya[e], g.bnb_act);
                        xh = xa[e];
                    }
                    const float dp = __shfl_xor(d, 32, 64), sp = __shfl_xor(sa, 32, 64), xp = __shfl_xor(xh, 32, 64);
                    // row w + 8 j (the h = 0 lane's), then row w + 8 j + 4 (the h = 1 lane's)
                    bn_bwd_acc(h ? dp : d, h ? sp : sa, h ? xp : xh, c1[w], c2[w]);
                    bn_bwd_acc(h ? d : dp, h ? sa : sp, h ? xh : xp, c1[w], c2[w]);
                }
            }
            if (h == 0) {
                g.bnb_part[((size_t)(i0 >> 5) * g.out_cols + col) * 2] = (((0.f + c1[0]) + c1[1]) + c1[2]) + c1[3];
                g.bnb_part[((size_t)(i0 >> 5) * g.out_cols + col) * 2 + 1] = (((0.f + c2[0]) + c2[1]) + c2[2]) + c2[3];
            }
            return;
        }
        if (g.relu_src) {
            const float* src = g.relu_src + col;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = i0 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (row < g.out_rows) {
                    const float y = src[(size_t)row * g.out_cols];
                    base[(size_t)row * ld] = g.src_act ? acc[e] * gl_act_bwd(y, g.src_act) : (y > 0.f ? acc[e] * g.relu_scale : 0.f);
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = i0 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (row < g.out_rows) base[(size_t)row * ld] = acc[e];
            }
        }
    }
}

template <int MODE, int KS, int NBUF>
__global__ __launch_bounds__(256 * KS) void linear_glds64_kernel(GlArgs g) {
    extern __shared__ __align__(16) float lds[];
    glds64_body<MODE, KS, NBUF>(g, blockIdx.x, lds);
}

// Several weight gradients of one backward pass in ONE launch (the three layers of the NFCF scorer, csrc/scorer.hip): job j
// owns the workgroups [first[j], first[j + 1]); each is exactly linear_glds64_kernel<GL_BWD_W> on that job's arguments.
constexpr int GL_MULTI_MAX = 8;
struct GlMulti {
    GlArgs g[GL_MULTI_MAX];
    unsigned first[GL_MULTI_MAX + 1];
    int n;
};
__global__ __launch_bounds__(256) void linear_glds64_wmulti_kernel(GlMulti m) {
    extern __shared__ __align__(16) float lds[];
    int j = 0;
    while (j + 1 < m.n && blockIdx.x >= m.first[j + 1]) ++j;
    glds64_body<GL_BWD_W, 1, 3>(m.g[j], blockIdx.x - m.first[j], lds);
}

// ... and their slab sums in one launch: job j = out[i] = sum over splits (four interleaved-by-quarter ascending chains, as
// slab_reduce_kernel) of slab[s][i] for the weights, of bslab[s][i - n] for the bias behind them
struct SlabJobs {
    const float* slab[GL_MULTI_MAX];
    const float* bslab[GL_MULTI_MAX];
    float* out[GL_MULTI_MAX];
    float* db[GL_MULTI_MAX];
    long long n[GL_MULTI_MAX];
    int nb[GL_MULTI_MAX], splits[GL_MULTI_MAX];
    int wide[GL_MULTI_MAX];      // many splits of a short vector: one WAVE per output (lane l adds splits l, l + 64, ..., then a butterfly)
    unsigned first[GL_MULTI_MAX + 1];
    int njobs;
};
__global__ __launch_bounds__(256) void slab_reduce_multi_kernel(SlabJobs J) {
    __shared__ float red[4][64];
    int j = 0;
    while (j + 1 < J.njobs && blockIdx.x >= J.first[j + 1]) ++j;
    const float* __restrict__ slab = J.slab[j];
    const float* __restrict__ bslab = J.bslab[j];
    const long long n = J.n[j];
    const int nb = J.nb[j], splits = J.splits[j];
    const int e = threadIdx.x & 63, part = threadIdx.x >> 6;
    if (J.wide[j]) {
        const long long o = (long long)(blockIdx.x - J.first[j]) * 4 + part;
        if (o >= n + nb) return;
        const float* __restrict__ src = o < n ? slab + o : bslab + (o - n);
        const long long stride = o < n ? n : nb;
        float a = 0.f;
        for (int sp = e; sp < splits; sp += 64) a += src[(size_t)sp * stride];
        a = wave_sum(a);
        if (e == 0) {
            if (o < n) J.out[j][o] = a;
            else J.db[j][o - n] = a;
        }
        return;
    }
    const long long i = (long long)(blockIdx.x - J.first[j]) * 64 + e;
    const int per = (splits + 3) / 4, s0 = part * per, s1 = min(splits, s0 + per);
    float a = 0.f;
    if (i < n) {
        for (int s = s0; s < s1; ++s) a += slab[(size_t)s * n + i];
    } else if (i < n + nb) {
        const long long k = i - n;
        for (int s = s0; s < s1; ++s) a += bslab[(size_t)s * nb + k];
    }
    red[part][e] = a;
    __syncthreads();
    if (part == 0) {
        const float t = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
        if (i < n) J.out[j][i] = t;
        else if (i < n + nb) J.db[j][i - n] = t;
    }
}

template <int MODE, int KS, int NBUF>
static int launch64(const GlArgs& g, long long blocks, hipStream_t stream, int kind) {
    static bool attr_set = false;
    const size_t ldsb = (size_t)KS * NBUF * 4096 * sizeof(float);
    if (!attr_set) {
        FR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(linear_glds64_kernel<MODE, KS, NBUF>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
        attr_set = true;
    }
    ProfScope prof((KernelKind)kind, stream);
    FR_LAUNCH(prof, (linear_glds64_kernel<MODE, KS, NBUF>), dim3((unsigned)blocks), dim3(256 * KS), ldsb, stream, g);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

// the macro-tile form of a launch described for the wave-private kernel (same g.ks / g.parts / chunks_per_part)
template <int MODE>
static int launch_shared(const GlArgs& g, hipStream_t stream, int kind) {
    const long long nmt = (long long)((g.tiles_i + 1) / 2) * ((g.tiles_j + 1) / 2);
    if (MODE == GL_BWD_W) return launch64<MODE, 1, 3>(g, nmt * g.parts, stream, kind);
    if (g.ks == 1) return launch64<MODE, 1, 3>(g, nmt, stream, kind);
    if (g.ks == 2) return launch64<MODE, 2, 3>(g, nmt, stream, kind);
    return launch64<MODE, 4, 2>(g, nmt, stream, kind);
}

static bool use_shared() {   // (read per call: a test flips it inside one process to compare the two forms bit for bit)
    return getenv("FAIRREC_LINEAR_NO_SHARED") == nullptr;
}

template <int MODE, bool KSPLIT>
static int launch_mode2(const GlArgs& g, long long blocks, hipStream_t stream, int kind) {
    static bool attr_set = false;
    const size_t ldsb = (size_t)GL_WAVES * GL_NBUF * GL_STAGE_FLOATS * sizeof(float);
    if (!attr_set) {
        FR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(linear_glds_kernel<MODE, KSPLIT>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
        attr_set = true;
    }
    ProfScope prof((KernelKind)kind, stream);
    FR_LAUNCH(prof, (linear_glds_kernel<MODE, KSPLIT>), dim3((unsigned)blocks), dim3(GL_WAVES * 64), ldsb, stream, g);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

template <int MODE>
static int launch_mode(const GlArgs& g, long long blocks, hipStream_t stream, int kind) {
    return g.ks > 1 ? launch_mode2<MODE, true>(g, blocks, stream, kind) : launch_mode2<MODE, false>(g, blocks, stream, kind);
}

// reduction parts per tile so that the chip has about two waves per SIMD (2048), dividing the chunk count evenly
static int pick_ks(long long ntiles, int chunks) {
    int ks = 1;
    while (ks < 4 && ntiles * ks < 2048 && chunks % (2 * ks) == 0 && chunks / (2 * ks) >= 2) ks *= 2;
    return ks;
}

bool glds_shared_form() { return use_shared(); }
int glds_pick_ks(long long ntiles, int chunks) { return pick_ks(ntiles, chunks); }

int glds_linear_fwd(const GlMat& X, const float* W, const float* bias, int64_t M, int N, int K, int act, float* Y,
                    hipStream_t stream, float* bn_part) {
    if (!bn_part && (!X.b || X.split >= K) && X.lda == K) {      // many rows, few columns: the streaming form (mlp_stream.hip)
        int rc;
        if (stream_linear_fwd(X.a, W, bias, M, N, K, act, Y, stream, &rc)) return rc;
    }
    GlArgs g{};
    g.bn_part = bn_part;
    g.A = X;
    g.B = GlMat{W, nullptr, K, 0, K};
    g.rowsA = (int)M;
    g.rowsB = N;
    g.R = K;
    g.tiles_i = (int)((M + 31) / 32);
    g.tiles_j = (N + 31) / 32;
    const long long ntiles = (long long)g.tiles_i * g.tiles_j;
    g.ks = pick_ks(ntiles, K / 32);
    g.chunks_per_part = K / 32 / g.ks;
    g.bias = bias;
    g.act = act;
    g.Y = Y;
    g.out_rows = (int)M;
    g.out_cols = N;
    if (use_shared()) return launch_shared<GL_FWD>(g, stream, K_LINEAR_FWD);
    const int tpb = GL_WAVES / g.ks;
    return launch_mode<GL_FWD>(g, (ntiles + tpb - 1) / tpb, stream, K_LINEAR_FWD);
}

int glds_linear_bwd_input(const float* dY, const float* W, int64_t M, int N, int K, float* dx0, int k0, float* dx1, int k1,
                          hipStream_t stream, const float* relu_src, float relu_scale, int src_act, const GlBnb* bnb) {
    if (bnb && (!use_shared() || k1 != 0 || relu_src || M > 32768)) {
        set_error("fr_linear_bwd_input_bnstats: needs the macro-tile kernels, one input block and M <= 32768");
        return FR_EUNSUPPORTED;
    }
    if (!bnb && k1 == 0 && (!relu_src || src_act != 0)) {
        int rc;
        if (stream_linear_bwd_input(dY, W, M, N, K, dx0, relu_src, src_act, stream, &rc)) return rc;
    }
    GlArgs g{};
    g.relu_src = relu_src;
    g.relu_scale = relu_scale;
    g.src_act = src_act;
    g.A = GlMat{dY, nullptr, N, 0, N};
    g.B = GlMat{W, nullptr, K, 0, K};
    g.rowsA = (int)M;
    g.rowsB = N;
    g.R = N;
    g.tiles_i = (int)((M + 31) / 32);
    g.tiles_j = K / 32;
    const long long ntiles = (long long)g.tiles_i * g.tiles_j;
    g.ks = pick_ks(ntiles, N / 32);
    g.chunks_per_part = N / 32 / g.ks;
    g.o_a = dx0;
    g.o_b = dx1;
    g.o_lda = k0;
    g.o_ldb = k1;
    g.o_split = k0;
    g.out_rows = (int)M;
    g.out_cols = K;
    if (bnb) {
        g.bnb_part = bnb->part;
        g.bnb_y = bnb->y;
        g.bnb_xhat = bnb->xhat;
        g.bnb_act = bnb->act;
        g.dr_on = bnb->p > 0.f;
        g.dr_thr = drop_threshold(bnb->p);
        g.dr_scale = 1.f / (1.f - bnb->p);
        g.dr_seed = bnb->seed;
        g.dr_off4 = bnb->offset / 4;
        g.dr_used = bnb->used;
    }
    if (bnb) return launch_shared<GL_BWD_IN_BN>(g, stream, K_LINEAR_BWD_INPUT);
    if (use_shared()) return launch_shared<GL_BWD_IN>(g, stream, K_LINEAR_BWD_INPUT);
    const int tpb = GL_WAVES / g.ks;
    return launch_mode<GL_BWD_IN>(g, (ntiles + tpb - 1) / tpb, stream, K_LINEAR_BWD_INPUT);
}

int glds_linear_bwd_weight(const float* dY, const GlMat& X, int64_t M, int N, int K, int splits, int rows_per_split,
                           float* slab, float* bslab, hipStream_t stream) {
    if (!X.b || X.split >= K) {
        int rc;
        if (stream_linear_bwd_weight(dY, X.a, M, N, K, splits, rows_per_split, slab, bslab, stream, &rc)) return rc;
    }
    GlArgs g{};
    g.A = GlMat{dY, nullptr, N, 0, N};
    g.B = X;
    g.R = (int)M;
    g.rowsA = g.rowsB = (int)M;
    g.tiles_i = N / 32;
    g.tiles_j = K / 32;
    g.ks = 1;
    g.parts = splits;
    g.chunks_per_part = rows_per_split / 32;
    g.slab = slab;
    g.bslab = bslab;
    g.out_rows = N;
    g.out_cols = K;
    if (use_shared()) return launch_shared<GL_BWD_W>(g, stream, K_LINEAR_BWD_WEIGHT);
    const long long nw = (long long)g.tiles_i * g.tiles_j * splits;
    return launch_mode<GL_BWD_W>(g, (nw + GL_WAVES - 1) / GL_WAVES, stream, K_LINEAR_BWD_WEIGHT);
}

int glds_linear_bwd_weight_multi(const GlWJob* jobs, int n, int64_t M, hipStream_t stream) {
    GlMulti m{};
    SlabJobs J{};
    unsigned blocks = 0, rblocks = 0;
    int ng = 0;
    for (int j = 0; j < n; ++j) {
        const GlWJob& q = jobs[j];
        J.slab[j] = q.slab;
        J.bslab[j] = q.bslab;
        J.out[j] = q.dW;
        J.db[j] = q.db;
        J.n[j] = (long long)q.N * q.K;
        J.nb[j] = q.db ? q.N : 0;
        J.splits[j] = q.splits;
        J.first[j] = rblocks;
        J.wide[j] = q.splits > 64 && J.n[j] + J.nb[j] <= 4096;
        rblocks += (unsigned)((J.n[j] + J.nb[j] + (J.wide[j] ? 3 : 63)) / (J.wide[j] ? 4 : 64));
        if (!q.dY) continue;            // sums only: the slabs were written by somebody else (the scorer's last layer)
        if (!q.X.b || q.X.split >= q.K) {       // many rows: the slabs of this job by the streaming kernel (mlp_stream.hip)
            int rc;
            if (stream_linear_bwd_weight(q.dY, q.X.a, M, q.N, q.K, q.splits, q.rows_per_split, q.slab, q.bslab, stream, &rc)) {
                if (rc) return rc;
                continue;
            }
        }
        GlArgs& g = m.g[ng];
        g = GlArgs{};
        g.A = GlMat{q.dY, nullptr, q.N, 0, q.N};
        g.B = q.X;
        g.R = (int)M;
        g.rowsA = g.rowsB = (int)M;
        g.tiles_i = q.N / 32;
        g.tiles_j = q.K / 32;
        g.ks = 1;
        g.parts = q.splits;
        g.chunks_per_part = q.rows_per_split / 32;
        g.slab = q.slab;
        g.bslab = q.bslab;
        g.out_rows = q.N;
        g.out_cols = q.K;
        m.first[ng] = blocks;
        blocks += (unsigned)(((g.tiles_i + 1) / 2) * ((g.tiles_j + 1) / 2)) * (unsigned)q.splits;
        ++ng;
    }
    m.first[ng] = blocks;
    m.n = ng;
    J.first[n] = rblocks;
    J.njobs = n;
    if (ng > 0) {
        static bool attr_set = false;
        const size_t ldsb = (size_t)3 * 4096 * sizeof(float);
        if (!attr_set) {
            FR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(linear_glds64_wmulti_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
            attr_set = true;
        }
        ProfScope prof(K_LINEAR_BWD_WEIGHT, stream);
        FR_LAUNCH(prof, linear_glds64_wmulti_kernel, dim3(blocks), dim3(256), ldsb, stream, m);
        FR_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(slab_reduce_multi_kernel, dim3(rblocks), dim3(256), 0, stream, J);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

int launch_act_bwd(const float* dY, const float* Y, int act, float scale, long long n, float* out, hipStream_t stream) {
    const long long n4 = n / 4;
    hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, dY, Y, act, scale, n4, out);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

}  // namespace fr
