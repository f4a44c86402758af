// Dense layers over MANY rows and few columns (a filter MLP over a whole embedding table: fairgo_gcn.py:173-185 -- [11 M, 128]
// through 128 -> 128 -> 64 -> 128): the forward and input-gradient products as ONE continuous stream per workgroup.
//
// The macro-tile kernels (mlp_glds.hip) give every 64 x 64 output tile its own workgroup: at K = 128 that is four 32-deep
// chunks of work behind a cold start (the first operand blocks: a memory latency with nothing to overlap), the weights fetched
// again by every workgroup, and the input rows fetched once per 64 output columns.  Priced per launch ([1.8 M, 128] x 128): 0.38 ms
// of MFMA time at peak, 0.38 ms of HBM time for input + output, 1.0 ms measured -- neither.  Here a workgroup is PERSISTENT: it
// keeps the whole weight matrix in LDS (<= 64 KB), takes row tiles of 128 rows x ALL output columns in turn, and treats its tiles'
// reduction chunks as one sequence: four LOADER waves issue the LDS-DMA of chunk g + 3 ... g + 6 (as many ring slots as LDS
// leaves) while eight COMPUTE waves run the MFMAs of chunk g -- two per SIMD, one wave's fragment reads under the other's MFMAs;
// at 128 columns a wave owns two row blocks of one column tile (four accumulator chains, the W fragments read once for both).  Input rows are read once, weights once per workgroup, and
// the start-up latency is paid once per workgroup instead of once per tile.  Splitting the roles also keeps loads and stores in
// different waves: s_waitcnt vmcnt counts both, they may retire out of order with respect to each other, and a compute wave that
// stores a finished tile would otherwise have to drain the prefetch to know its next chunk has landed.
//
// Same sums in the same order as the macro-tile kernels (chunks ascending, even / odd reduction steps in two accumulators, one
// part): bit-identical outputs (tests/test_mlp_hip.py).  Shapes: output width 64 or 128, reduction length a multiple of 32 up to
// 128, M >= 32768 (the weight gradient: 65536); everything else stays with mlp_glds.hip.
#include <stdlib.h>

#include "common.hpp"
#include "kernels.hpp"
#include "mlp_glds.hpp"

namespace fr {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;

constexpr int ST_LOADERS = 4;        // loader waves

struct StArgs {
    const float* A;      // [M, R] row-major
    const float* W;      // [N, K] row-major (forward: R = K, C = N; input gradient: R = N, C = K)
    int ldw;
    int M, R, C;
    int cpp;             // chunks per part of a tile's reduction: the macro-tile kernels' summation order for this shape (glds_pick_ks)
    int slots;           // ring slots (a chunk of a row tile each: ST_RB blocks); slots - 2 chunks are requested ahead
    const float* bias;   // forward
    int act;
    float* out;          // [M, C]
    const float* src;    // input gradient, optional: Y = act(z) of the layer below, [M, C]
    int src_act;
};

__device__ __forceinline__ float st_act_bwd(float y, int act) {
    switch (act) {
        case 1: return y > 0.f ? 1.f : 0.f;
        case 2: return y > 0.f ? 1.f : 0.01f;
        case 3: return y * (1.f - y);
        case 4: return 1.f - y * y;
        default: return 1.f;
    }
}

template <int N>
__device__ __forceinline__ void st_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int PER>     // PER LDS-DMA instructions per chunk and loader wave; at most 6 chunks are ever in flight behind the one awaited
__device__ __forceinline__ void st_wait_newer(int newer) {
    switch (newer) {
        case 0: st_wait_vm<0>(); break;
        case 1: st_wait_vm<PER>(); break;
        case 2: st_wait_vm<2 * PER>(); break;
        case 3: st_wait_vm<3 * PER>(); break;
        case 4: st_wait_vm<4 * PER>(); break;
        case 5: st_wait_vm<5 * PER>(); break;
        default: st_wait_vm<6 * PER>(); break;
    }
}

// B_T: the reduction index runs along W's rows (input gradient); CT = C / 32 column tiles; RPW: row blocks per compute wave (two
// at 128 columns: four accumulator chains per wave and the W fragments read once for both -- half the barriers per MFMA)
// ST_RB: 32-row blocks of a row tile (128 rows, or 64 where the registers of two row blocks per wave do not fit: the input gradient
// through an activation at 128 columns keeps 32 `src` values per lane in flight)
template <bool B_T, int CT, int ST_RB, int RPW, bool SRC, bool ONEPART>
__global__ __launch_bounds__((ST_RB / RPW * CT + ST_LOADERS) * 64) void linear_stream_kernel(StArgs g) {
    extern __shared__ __align__(16) float lds[];    // W image [R / 32][CT][1024], then the ring [slots][ST_RB][1024]
    constexpr int NCW = ST_RB / RPW * CT;
    static_assert(NCW == 8, "eight compute waves: two per SIMD");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int RC = g.R >> 5, SLOTS = g.slots, AHEAD = g.slots - 2;
    float* wimg = lds;
    float* ring = lds + (size_t)RC * CT * 1024;
    const int ntiles = (g.M + 32 * ST_RB - 1) / (32 * ST_RB);
    const int n_my = (int)blockIdx.x < ntiles ? (ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int G = n_my * RC;                         // this workgroup's chunks, tile after tile
    const int srow = lane >> 3, sslot = lane & 7;

    if (wave >= NCW) {
        // ================= loader waves: instruction lw of every block ===================================================================
        const int lw = wave - NCW;
        const int row = 8 * lw + srow, sw = (sslot ^ ((row >> 1) & 7)) << 2;
        // the weight image, once: block (c, j) = 32 reduction elements x 32 output columns
        for (int b = 0; b < RC * CT; ++b) {
            const int c = b / CT, j = b - c * CT;
            const float* p;
            if (B_T) {
                int n = c * 32 + row;
                n = n < g.R ? n : g.R - 1;
                p = g.W + (size_t)n * g.ldw + j * 32 + sw;
            } else {
                int n = j * 32 + row;
                n = n < g.C ? n : g.C - 1;
                p = g.W + (size_t)n * g.ldw + c * 32 + sw;
            }
            __builtin_amdgcn_global_load_lds((glb_vp)p, (lds_vp)(wimg + (size_t)b * 1024 + lw * 256), 16, 0, 0);
        }
        int st_t = 0, st_c = 0, st_slot = 0;         // the next chunk to request: (tile, chunk) -> ring slot
        auto stage = [&]() {
            const long long r0 = ((long long)blockIdx.x + (long long)st_t * gridDim.x) * (32 * ST_RB) + row;
            float* dst = ring + (size_t)st_slot * (ST_RB * 1024) + lw * 256;
#pragma unroll
            for (int b = 0; b < ST_RB; ++b) {
                long long rr = r0 + b * 32;
                rr = rr < g.M ? rr : g.M - 1;
                __builtin_amdgcn_global_load_lds((glb_vp)(g.A + rr * g.R + st_c * 32 + sw), (lds_vp)(dst + b * 1024), 16, 0, 0);
            }
            if (++st_c == RC) {
                st_c = 0;
                ++st_t;
            }
            st_slot = st_slot + 1 == SLOTS ? 0 : st_slot + 1;
        };
        int issued = 0;
        for (; issued < AHEAD && issued < G; ++issued) stage();
        st_wait_newer<ST_RB>(issued);                 // the weight image has landed (everything older than the chunks)
        __builtin_amdgcn_s_barrier();                 // B_w
        for (int k = 0; k < G; ++k) {
            if (issued < G) {
                // chunk k + AHEAD goes into the slot chunk k - 2 held: every compute wave is past B_(k-1), i.e. through the
                // fragment reads AND the MFMAs of chunk k - 2
                stage();
                ++issued;
            }
            st_wait_newer<ST_RB>(issued - 1 - k);     // chunk k has landed
            __builtin_amdgcn_s_barrier();             // B_k
        }
        return;
    }

    // ================= compute waves: RPW row blocks x one column tile each, two waves per SIMD ========================================
    const int wr = wave / CT, wj = wave - wr * CT;
    const int r = lane & 31, h = lane >> 5;
    const unsigned wbase = (unsigned)(size_t)(__attribute__((address_space(3))) float*)wimg + wj * 4096;
    const unsigned rbase = (unsigned)(size_t)(__attribute__((address_space(3))) float*)ring + wr * RPW * 4096;
    unsigned rn[4], rtt[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) rn[j] = r * 128 + (((2 * j + h) ^ ((r >> 1) & 7)) << 4);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int row = 8 * (q >> 2) + 4 * h + (q & 3);
        rtt[q] = row * 128 + ((((r >> 2) ^ ((row >> 1) & 7)) << 4) | ((r & 3) << 2));
    }
    // (ONEPART: the reduction of a tile is one part -- every shape with >= 2048 output tiles -- and needs no running total)
    f32x16 acc[RPW][2], tot[ONEPART ? 1 : RPW];
#pragma unroll
    for (int p = 0; p < RPW; ++p) acc[p][0] = acc[p][1] = f32x16{0};
#pragma unroll
    for (int p = 0; p < (ONEPART ? 1 : RPW); ++p) tot[p] = f32x16{0};
    const int col = wj * 32 + r;
    const float bias = (!B_T && g.bias) ? g.bias[col] : 0.f;      // once per wave: a load per tile would be a memory latency per tile
    float sv[SRC ? RPW : 1][16];                                   // the tile's `src` values, requested when the tile begins
    int in_part = 0;

    __builtin_amdgcn_s_barrier();                     // B_w: the weight image
    int c = 0, t = 0, slot = 0;
    for (int k = 0; k < G; ++k) {
        if (SRC && c == 0) {
#pragma unroll
            for (int p = 0; p < RPW; ++p) {
                const long long i0 = ((long long)blockIdx.x + (long long)t * gridDim.x) * (32 * ST_RB) + (wr * RPW + p) * 32 + 4 * h;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    long long row = i0 + (e & 3) + 8 * (e >> 2);
                    row = row < g.M ? row : g.M - 1;
                    sv[SRC ? p : 0][e] = g.src[(size_t)row * g.C + col];
                }
            }
        }
        __builtin_amdgcn_s_barrier();                 // B_k: chunk k is in LDS
        f4 av[RPW][4], bv[4];
        float bt[16];
        const unsigned ab = rbase + (unsigned)slot * (ST_RB * 4096), bb = wbase + (unsigned)(c * CT) * 4096;
#pragma unroll
        for (int p = 0; p < RPW; ++p)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(av[p][j]) : "v"(ab + p * 4096 + rn[j]));
        if (!B_T) {
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(bv[j]) : "v"(bb + rn[j]));
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(av[0][0]), "+v"(av[0][1]), "+v"(av[0][2]), "+v"(av[0][3]), "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]));
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) asm volatile("ds_read_b32 %0, %1" : "=v"(bt[q]) : "v"(bb + rtt[q]));
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(av[0][0]), "+v"(av[0][1]), "+v"(av[0][2]), "+v"(av[0][3]), "+v"(bt[0]), "+v"(bt[1]), "+v"(bt[2]), "+v"(bt[3]),
                           "+v"(bt[4]), "+v"(bt[5]), "+v"(bt[6]), "+v"(bt[7]), "+v"(bt[8]), "+v"(bt[9]), "+v"(bt[10]), "+v"(bt[11]),
                           "+v"(bt[12]), "+v"(bt[13]), "+v"(bt[14]), "+v"(bt[15]));
        }
        if constexpr (RPW == 2) asm volatile("" : "+v"(av[RPW - 1][0]), "+v"(av[RPW - 1][1]), "+v"(av[RPW - 1][2]), "+v"(av[RPW - 1][3]));
#pragma unroll
        for (int q = 0; q < 16; q += 2)
#pragma unroll
            for (int p = 0; p < RPW; ++p) {
                acc[p][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[p][q >> 2][q & 3], B_T ? bt[q] : bv[q >> 2][q & 3], acc[p][0], 0, 0, 0);
                acc[p][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[p][(q + 1) >> 2][(q + 1) & 3], B_T ? bt[q + 1] : bv[(q + 1) >> 2][(q + 1) & 3],
                                                                 acc[p][1], 0, 0, 0);
            }
        slot = slot + 1 == SLOTS ? 0 : slot + 1;
        if (!ONEPART && ++in_part == g.cpp) {        // a part of the reduction ends: parts are added in order (the macro-tile kernels' KS split)
            in_part = 0;
#pragma unroll
            for (int p = 0; p < RPW; ++p) {
                tot[ONEPART ? 0 : p] += acc[p][0] + acc[p][1];
                acc[p][0] = acc[p][1] = f32x16{0};
            }
        }
        if (++c < RC) continue;
        // ---- a tile is complete: the macro-tile kernels' epilogues --------------------------------------------------------------------
        c = 0;
        const long long tile0 = ((long long)blockIdx.x + (long long)t * gridDim.x) * (32 * ST_RB);
        ++t;
#pragma unroll
        for (int p = 0; p < RPW; ++p) {
            const long long i0 = tile0 + (wr * RPW + p) * 32;
            f32x16 a;
            if (ONEPART) {
                a = acc[p][0] + acc[p][1];
                acc[p][0] = acc[p][1] = f32x16{0};
            } else {
                a = tot[ONEPART ? 0 : p];
                tot[ONEPART ? 0 : p] = f32x16{0};
            }
            float* op = g.out + (size_t)(i0 + 4 * h) * g.C + col;
            if (!B_T) {
                if (g.act <= 2) {
                    const float neg = g.act == 0 ? 1.f : (g.act == 1 ? 0.f : 0.01f);
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int ro = (e & 3) + 8 * (e >> 2);
                        const float v = a[e] + bias;
                        if (i0 + 4 * h + ro < g.M) op[(size_t)ro * g.C] = v > 0.f ? v : v * neg;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int ro = (e & 3) + 8 * (e >> 2);
                        const float v = a[e] + bias;
                        if (i0 + 4 * h + ro < g.M) op[(size_t)ro * g.C] = g.act == 3 ? 1.f / (1.f + __expf(-v)) : tanhf(v);
                    }
                }
            } else if (SRC) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int ro = (e & 3) + 8 * (e >> 2);
                    if (i0 + 4 * h + ro < g.M) op[(size_t)ro * g.C] = a[e] * st_act_bwd(sv[SRC ? p : 0][e], g.src_act);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int ro = (e & 3) + 8 * (e >> 2);
                    if (i0 + 4 * h + ro < g.M) op[(size_t)ro * g.C] = a[e];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The weight gradient dW = dY^T X over many rows: slab[s] = dY[rows of split s]^T X[rows of split s] for ALL of [N, K] by one
// workgroup (the macro-tile kernel gives every 64 x 64 piece of it a workgroup of its own: dY and X are then read N / 64 and
// K / 64 times over).  Nothing is resident here -- both operands stream, 32 rows of each per chunk, (N + K) / 32 blocks -- and the
// accumulators stay in registers for the whole split.  Compute waves: one output row-tile i (32 columns of dY) and one or two
// column tiles j (of X) each, eight waves at 128 x 128 / 128 x 64 / 64 x 128.  Splits, chunk order, the two accumulators and the
// column sums of dY (bslab) are the macro-tile kernel's: the same slabs bit for bit, and the same slab reduction behind them.
struct WgArgs {
    const float* dY;     // [M, N]
    const float* X;      // [M, K]
    int M, N, K;
    int splits, chunks_per_split;
    float* slab;         // [splits][N][K]
    float* bslab;        // [splits][N] or nullptr
};

constexpr int WG_SLOTS = 4, WG_AHEAD = 2;    // a chunk is (N + K) / 32 blocks = up to 32 KB: two ahead is > 3 us of MFMAs

template <int NB>
__device__ __forceinline__ void wg_wait_newer(int newer) {     // NB LDS-DMA instructions per chunk and loader wave
    if (newer <= 0) st_wait_vm<0>();
    else if (newer == 1) st_wait_vm<NB>();
    else st_wait_vm<2 * NB>();
}

template <int NT, int KT>        // N / 32, K / 32
__global__ __launch_bounds__(((NT * KT >= 8 ? 8 : NT * KT) + ST_LOADERS) * 64) void wgrad_stream_kernel(WgArgs g) {
    constexpr int T = NT * KT, NCW = T >= 8 ? 8 : T, TPW = T / NCW, NB = NT + KT;
    static_assert(TPW == 1 || (TPW == 2 && KT == 4), "two tiles per wave share the dY tile: K = 128");
    extern __shared__ __align__(16) float lds[];    // ring [WG_SLOTS][NT + KT][1024]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int srow = lane >> 3, sslot = lane & 7;
    const int CPS = g.chunks_per_split;
    const int total_chunks = (g.M + 31) >> 5;
    // this workgroup's splits s = blockIdx.x, + gridDim.x, ...; split s has chunks [s CPS, min((s + 1) CPS, total)) (possibly none)
    const int n_my = (int)blockIdx.x < g.splits ? (g.splits - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    auto chunks_of = [&](int s) {
        const int n = total_chunks - s * CPS;
        return n < 0 ? 0 : (n < CPS ? n : CPS);
    };

    if (wave >= NCW) {
        // ================= loader waves: instruction lw of every block of every chunk ====================================================
        const int lw = wave - NCW;
        int G = 0;
        for (int k = 0; k < n_my; ++k) G += chunks_of((int)blockIdx.x + k * (int)gridDim.x);
        int st_k = 0, st_c = 0, st_slot = 0;
        while (st_k < n_my && chunks_of((int)blockIdx.x + st_k * (int)gridDim.x) == 0) ++st_k;
        auto stage = [&]() {
            const int s = (int)blockIdx.x + st_k * (int)gridDim.x;
            const int row = 8 * lw + srow;
            long long rr = ((long long)s * CPS + st_c) * 32 + row;
            rr = rr < g.M ? rr : g.M - 1;
            const int sw = (sslot ^ ((row >> 1) & 7)) << 2;
            float* dst = lds + (size_t)st_slot * NB * 1024 + lw * 256;
#pragma unroll
            for (int b = 0; b < NT; ++b)
                __builtin_amdgcn_global_load_lds((glb_vp)(g.dY + rr * g.N + b * 32 + sw), (lds_vp)(dst + b * 1024), 16, 0, 0);
#pragma unroll
            for (int b = 0; b < KT; ++b)
                __builtin_amdgcn_global_load_lds((glb_vp)(g.X + rr * g.K + b * 32 + sw), (lds_vp)(dst + (NT + b) * 1024), 16, 0, 0);
            if (++st_c == chunks_of(s)) {
                st_c = 0;
                ++st_k;
                while (st_k < n_my && chunks_of((int)blockIdx.x + st_k * (int)gridDim.x) == 0) ++st_k;
            }
            st_slot = st_slot + 1 == WG_SLOTS ? 0 : st_slot + 1;
        };
        int issued = 0;
        for (; issued < WG_AHEAD && issued < G; ++issued) stage();
        for (int k = 0; k < G; ++k) {
            if (issued < G) {       // chunk k + 2 into the slot chunk k - 2 held: the compute waves are past B_(k-1), through chunk k - 2
                stage();
                ++issued;
            }
            wg_wait_newer<NB>(issued - 1 - k);
            __builtin_amdgcn_s_barrier();             // B_k
        }
        return;
    }

    // ================= compute waves ===================================================================================================
    const int ti = TPW == 2 ? wave >> 1 : wave / KT;                 // row tile of the output = column tile of dY
    const int tj0 = TPW == 2 ? (wave & 1) : wave - ti * KT;          // column tile(s) of the output = of X: tj0 (and tj0 + 2)
    const int r = lane & 31, h = lane >> 5;
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds;
    unsigned rtt[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int row = 8 * (q >> 2) + 4 * h + (q & 3);
        rtt[q] = row * 128 + ((((r >> 2) ^ ((row >> 1) & 7)) << 4) | ((r & 3) << 2));
    }
    int slot = 0;
    for (int k = 0; k < n_my; ++k) {
        const int s = (int)blockIdx.x + k * (int)gridDim.x;
        const int nchunk = chunks_of(s);
        f32x16 acc[TPW][2];
#pragma unroll
        for (int i = 0; i < TPW; ++i) acc[i][0] = acc[i][1] = f32x16{0};
        float bsum = 0.f;
        for (int c = 0; c < nchunk; ++c) {
            __builtin_amdgcn_s_barrier();             // B: this chunk is in LDS
            float at[16], bt[TPW][16];
            const unsigned sb = base + (unsigned)slot * (NB * 4096);
#pragma unroll
            for (int q = 0; q < 16; ++q) asm volatile("ds_read_b32 %0, %1" : "=v"(at[q]) : "v"(sb + ti * 4096 + rtt[q]));
#pragma unroll
            for (int i = 0; i < TPW; ++i)
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    asm volatile("ds_read_b32 %0, %1" : "=v"(bt[i][q]) : "v"(sb + (NT + tj0 + 2 * i) * 4096 + rtt[q]));
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(at[0]), "+v"(at[1]), "+v"(at[2]), "+v"(at[3]), "+v"(at[4]), "+v"(at[5]), "+v"(at[6]), "+v"(at[7]),
                           "+v"(at[8]), "+v"(at[9]), "+v"(at[10]), "+v"(at[11]), "+v"(at[12]), "+v"(at[13]), "+v"(at[14]), "+v"(at[15]));
#pragma unroll
            for (int i = 0; i < TPW; ++i)
                asm volatile(""
                             : "+v"(bt[i][0]), "+v"(bt[i][1]), "+v"(bt[i][2]), "+v"(bt[i][3]), "+v"(bt[i][4]), "+v"(bt[i][5]), "+v"(bt[i][6]),
                               "+v"(bt[i][7]), "+v"(bt[i][8]), "+v"(bt[i][9]), "+v"(bt[i][10]), "+v"(bt[i][11]), "+v"(bt[i][12]),
                               "+v"(bt[i][13]), "+v"(bt[i][14]), "+v"(bt[i][15]));
            // the reduction runs over batch rows: the last chunk may reach past them
            const long long red0 = ((long long)s * CPS + c) * 32;
            if (red0 + 32 > g.M) {
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    if (red0 + 8 * (q >> 2) + 4 * h + (q & 3) >= g.M) at[q] = 0.f;
            }
            if (g.bslab) {
#pragma unroll
                for (int q = 0; q < 16; ++q) bsum += at[q];
            }
#pragma unroll
            for (int q = 0; q < 16; q += 2)
#pragma unroll
                for (int i = 0; i < TPW; ++i) {
                    acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(at[q], bt[i][q], acc[i][0], 0, 0, 0);
                    acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(at[q + 1], bt[i][q + 1], acc[i][1], 0, 0, 0);
                }
            slot = slot + 1 == WG_SLOTS ? 0 : slot + 1;
        }
        float* out = g.slab + (size_t)s * g.N * g.K;
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const f32x16 a = acc[i][0] + acc[i][1];
            const int j0 = (tj0 + 2 * i) * 32;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = ti * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                out[(size_t)row * g.K + j0 + r] = a[e];
            }
        }
        if (g.bslab && tj0 == 0) {
            bsum += __shfl_xor(bsum, 32, 64);
            if (h == 0) g.bslab[(size_t)s * g.N + ti * 32 + r] = bsum;
        }
    }
}

template <int NT, int KT>
static int launch_wgrad(const WgArgs& a, hipStream_t stream) {
    constexpr int NCW = NT * KT >= 8 ? 8 : NT * KT;
    const size_t ldsb = (size_t)WG_SLOTS * (NT + KT) * 4096;
    const int blocks = a.splits < 256 ? a.splits : 256;
    static size_t have = 0;
    if (ldsb > have) {
        FR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_stream_kernel<NT, KT>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
        have = ldsb;
    }
    ProfScope prof(K_LINEAR_BWD_WEIGHT, stream);
    FR_LAUNCH(prof, (wgrad_stream_kernel<NT, KT>), dim3((unsigned)blocks), dim3((NCW + ST_LOADERS) * 64), ldsb, stream, a);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

static bool stream_enabled() { return getenv("FAIRREC_LINEAR_NO_STREAM") == nullptr; }

template <bool B_T, int CT, int ST_RB, int RPW, bool SRC, bool ONEPART>
static int launch_stream_ct(StArgs a, hipStream_t stream, int kind) {
    // as many ring slots as the 160 KB of LDS leave beside the weight image (with a margin), at most 8
    const size_t wbytes = (size_t)(a.R / 32) * CT * 4096, slot = (size_t)ST_RB * 4096;
    int slots = (int)((152 * 1024 - wbytes) / slot);
    a.slots = slots > 8 ? 8 : slots;
    const size_t ldsb = wbytes + (size_t)a.slots * slot;
    const int ntiles = (a.M + 32 * ST_RB - 1) / (32 * ST_RB);
    const int blocks = ntiles < 256 ? ntiles : 256;      // one persistent workgroup per CU
    static size_t have = 0;
    if (ldsb > have) {
        FR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(linear_stream_kernel<B_T, CT, ST_RB, RPW, SRC, ONEPART>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
        have = ldsb;
    }
    ProfScope prof((KernelKind)kind, stream);
    FR_LAUNCH(prof, (linear_stream_kernel<B_T, CT, ST_RB, RPW, SRC, ONEPART>), dim3((unsigned)blocks), dim3((ST_RB / RPW * CT + ST_LOADERS) * 64), ldsb, stream, a);
    FR_CHECK_LAUNCH();
    return FR_OK;
}
template <bool B_T, bool ONEPART>
static int launch_stream_p(const StArgs& a, hipStream_t stream, int kind) {
    if (B_T && a.src)      // (two row blocks per wave + 32 `src` values in flight fit the registers only without the running total)
        return a.C == 128 ? (ONEPART ? launch_stream_ct<B_T, 4, 4, 2, true, ONEPART>(a, stream, kind)
                                     : launch_stream_ct<B_T, 4, 2, 1, true, ONEPART>(a, stream, kind))
                          : launch_stream_ct<B_T, 2, 4, 1, true, ONEPART>(a, stream, kind);
    return a.C == 128 ? launch_stream_ct<B_T, 4, 4, 2, false, ONEPART>(a, stream, kind)
                      : launch_stream_ct<B_T, 2, 4, 1, false, ONEPART>(a, stream, kind);
}
template <bool B_T>
static int launch_stream(const StArgs& a, hipStream_t stream, int kind) {
    return a.cpp == a.R / 32 ? launch_stream_p<B_T, true>(a, stream, kind) : launch_stream_p<B_T, false>(a, stream, kind);
}

static bool stream_shape(int64_t M, int R, int C) {
    // (measured against the macro-tile kernels, profiles/r05_stream_gemm.txt: 0.8 x at 16 K rows -- half the CUs idle --, 1.2-1.4 x from 32 K)
    return stream_enabled() && M >= 32768 && M <= 0x7fffffffLL / 128 && R % 32 == 0 && R >= 32 && R <= 128 && (C == 64 || C == 128);
}

// Y = act(X W^T + b) for one row-major X [M, K]; FR_EUNSUPPORTED-style `false` when the shape is not this kernel's
bool stream_linear_fwd(const float* X, const float* W, const float* bias, int64_t M, int N, int K, int act, float* Y,
                       hipStream_t stream, int* rc) {
    if (!stream_shape(M, K, N)) return false;
    StArgs a{X, W, K, (int)M, K, N, K / 32 / glds_pick_ks((long long)((M + 31) / 32) * (N / 32), K / 32), 0, bias, act, Y, nullptr, 0};
    *rc = launch_stream<false>(a, stream, K_LINEAR_FWD);
    return true;
}

// dX = dY W (optionally o act'(Yin) of the layer below) for one row-major dX [M, K]
bool stream_linear_bwd_input(const float* dY, const float* W, int64_t M, int N, int K, float* dX, const float* src, int src_act,
                             hipStream_t stream, int* rc) {
    if (!stream_shape(M, N, K)) return false;
    StArgs a{dY, W, K, (int)M, N, K, N / 32 / glds_pick_ks((long long)((M + 31) / 32) * (K / 32), N / 32), 0, nullptr, 0, dX, src, src_act};
    *rc = launch_stream<true>(a, stream, K_LINEAR_BWD_INPUT);
    return true;
}

}  // namespace fr

namespace fr {
// slab[s] = dY[rows of s]^T X[rows of s] (+ bslab) for one row-major X [M, K]: the splits of glds_linear_bwd_weight, its slabs
bool stream_linear_bwd_weight(const float* dY, const float* X, int64_t M, int N, int K, int splits, int rows_per_split, float* slab,
                              float* bslab, hipStream_t stream, int* rc) {
    // (0.8 x the macro-tile kernel at 16-32 K rows, 1.0-1.1 x from 64 K, 1.3 x at 11 M)
    if (!stream_enabled() || M < 65536 || M > 0x7fffffffLL / 128 || rows_per_split % 32 != 0 || (N != 64 && N != 128) || (K != 64 && K != 128))
        return false;
    WgArgs a{dY, X, (int)M, N, K, splits, rows_per_split / 32, slab, bslab};
    if (N == 128 && K == 128) *rc = launch_wgrad<4, 4>(a, stream);
    else if (N == 128) *rc = launch_wgrad<4, 2>(a, stream);
    else if (K == 128) *rc = launch_wgrad<2, 4>(a, stream);
    else *rc = launch_wgrad<2, 2>(a, stream);
    return true;
}
}  // namespace fr
