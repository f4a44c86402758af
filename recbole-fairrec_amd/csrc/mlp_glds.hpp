// Fast dense-layer kernels (mlp_glds.hip): what mlp.hip's entry points call when a layer has the fast form.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fr {

enum { GL_FWD = 0, GL_BWD_IN = 1, GL_BWD_W = 2,
       GL_BWD_IN_BN = 3 };      // GL_BWD_IN with the BatchNorm-backward epilogue (GlArgs::bnb_*): an instance of its own, so that
                                // the plain input-gradient kernels keep their registers

// a row-major matrix whose columns [0, split) live in `a` (leading dimension lda) and the rest in `b` (cat(U[u], I[i]))
struct GlMat {
    const float* a;
    const float* b;
    int lda, ldb, split;
};

struct GlArgs {
    GlMat A, B;
    int rowsA, rowsB;        // rows of a row-contiguous operand are clamped below these
    int R;                   // reduction length
    int tiles_i, tiles_j;    // 32 x 32 output tiles
    int ks;                  // reduction parts per tile inside a workgroup (forward, input gradient)
    int parts;               // ... or row splits, one slab each (weight gradient)
    int chunks_per_part;     // 32-element reduction chunks per part
    const float* bias;
    int act;
    float* Y;
    int out_rows, out_cols;
    float *o_a, *o_b;        // input gradient: columns [0, o_split) -> o_a, the rest -> o_b
    int o_lda, o_ldb, o_split;
    float *slab, *bslab;     // weight gradient
    float* bn_part;          // forward, optional: per (32-row tile, column) (mean, M2) partials of Z = X W^T + b for the BatchNorm behind
    const float* relu_src;   // input gradient, optional: the layer's input Xd = relu(z) o keep, leading dimension out_cols;
    float relu_scale;        //   the result is then the gradient at z: (dY W) o relu_scale o [Xd > 0]
    int src_act;             // ... or (src_act != 0) the layer's input is Y = act(z), activation code src_act, and the result the
                             //   gradient at z: (dY W) o act'(Y) -- the activation's backward pass in this epilogue
    // input gradient, optional (macro-tile form only): the layer's input is the output of a BatchNorm layer, dropped on its way
    // up.  The epilogue then (1) multiplies the tile by the dropout's keep pattern (regenerated: dropout.hpp) before it stores
    // it -- the gradient at the BatchNorm layer's activation output -- and (2) leaves that layer's backward statistics of the
    // tile's 32 rows, sum dA and sum dA xhat per column (dA = the stored gradient o act'(Y)), in bnb_part[(tile, column)]:
    // what bn_bwd_stats_kernel computes for a 32-row chunk, so that neither a dropout launch nor a statistics launch runs
    // between this product and the BatchNorm layer's apply launch.
    float* bnb_part;
    const float* bnb_y;      // [M, out_cols] the BatchNorm layer's activation output
    const float* bnb_xhat;   // [M, out_cols] its normalised input
    int bnb_act;
    int dr_on;               // a dropout between the two layers
    unsigned dr_thr;
    float dr_scale;
    unsigned long long dr_seed, dr_off4;
    const unsigned long long* dr_used;
};
struct GlBnb {               // the same, as the entry point hands it over
    float* part;
    const float *y, *xhat;
    int act;
    float p;
    unsigned long long seed, offset;
    const unsigned long long* used;
};

// Y = act(X W^T + b); needs K % 32 == 0, X.split % 32 == 0, 16-byte aligned rows
int glds_linear_fwd(const GlMat& X, const float* W, const float* bias, int64_t M, int N, int K, int act, float* Y,
                    hipStream_t stream, float* bn_part = nullptr);
int glds_pick_ks(long long ntiles, int chunks);   // reduction parts per 32 x 32 tile of the forward / input-gradient products (their summation order)
bool glds_shared_form();     // the macro-tile kernels are selected (FAIRREC_LINEAR_NO_SHARED unset): the only ones that write bn_part
// dX = dY W; needs N % 32 == 0, K % 32 == 0, k0 % 32 == 0
int glds_linear_bwd_input(const float* dY, const float* W, int64_t M, int N, int K, float* dx0, int k0, float* dx1, int k1,
                          hipStream_t stream, const float* relu_src = nullptr, float relu_scale = 1.f, int src_act = 0,
                          const GlBnb* bnb = nullptr);
// slab[s] = dY[rows of s]^T X[rows of s], bslab[s] = column sums of dY[rows of s] (bslab may be null); needs N % 32 == 0,
// K % 32 == 0, X.split % 32 == 0, rows_per_split % 32 == 0
int glds_linear_bwd_weight(const float* dY, const GlMat& X, int64_t M, int N, int K, int splits, int rows_per_split,
                           float* slab, float* bslab, hipStream_t stream);
// Several weight gradients in one launch + their slab sums in a second (fr_linear_bwd_weight_multi).  A job without dY only
// sums slabs that are already there (splits x [N, K] then, if db, splits x [N]).
struct GlWJob {
    const float* dY;
    GlMat X;
    int N, K, splits, rows_per_split;
    float *slab, *bslab, *dW, *db;
};
int glds_linear_bwd_weight_multi(const GlWJob* jobs, int n, int64_t M, hipStream_t stream);
// The same two products for many rows and 64 / 128 output columns as one stream per persistent workgroup (mlp_stream.hip);
// false: not that kernel's shape (the caller goes on to the macro-tile kernels), true: launched, *rc = its status.
bool stream_linear_fwd(const float* X, const float* W, const float* bias, int64_t M, int N, int K, int act, float* Y,
                       hipStream_t stream, int* rc);
bool stream_linear_bwd_input(const float* dY, const float* W, int64_t M, int N, int K, float* dX, const float* src, int src_act,
                             hipStream_t stream, int* rc);
bool stream_linear_bwd_weight(const float* dY, const float* X, int64_t M, int N, int K, int splits, int rows_per_split, float* slab,
                              float* bslab, hipStream_t stream, int* rc);
// out = dY o act'(Y) elementwise (n % 4 == 0)
int launch_act_bwd(const float* dY, const float* Y, int act, float scale, long long n, float* out, hipStream_t stream);

}  // namespace fr
