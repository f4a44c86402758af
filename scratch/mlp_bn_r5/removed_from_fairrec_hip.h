/* ---- an MLP with BatchNorm behind every layer, one launch per layer and direction (csrc/mlp_bn.hip) -----------------------------
 * Replaces, for MLPLayers(..., bn=True) (layers.py:56-85; pfcn_biasedmf.py:113-142: PFCN's filters and discriminators), the
 * three launches per layer forward and four backward of fr_linear_fwd_bnstats / fr_bn_fwd_ex / fr_bn_bwd / fr_linear_bwd_input:
 * a layer is NORMALISED BY THE LAUNCH THAT CONSUMES IT.  fr_bn_src describes an [M, width] activation that such a launch forms
 * while loading it: act(gamma (Z - mean) invstd + beta) from the pre-BatchNorm output Z of the layer below and that layer's
 * folded statistics fin = [width][2] (mean, 1 / sqrt(var + eps)) -- or Z as it is when fin == NULL (the MLP's input) -- with
 * dropout (csrc/dropout.hpp's stream: seed, the pass's call counter, element offset drop_off, a multiple of 4) on top when
 * drop_p > 0.
 *   fr_bnl_fwd      Z[M, N] = in W^T + bias; A_out (may be NULL) = the formed input, kept for the weight gradient; the layer's
 *                   statistics -> fin_out [N][2], running_mean / running_var (momentum; may be NULL), *nbt += nbt_inc.
 *                   drop_state / drop_used / drop_tick as fr_dropout_apply's `state`, `used_out`, `tick`.
 *   fr_bnl_out      Y[M, N] = the formed activation (the MLP's output).
 *   fr_bnl_bwd_top  sums [N][2] = column sums of dY s and dY s xhat for the top layer (s = act'(y); = dbeta, dgamma, also written
 *                   to dbeta / dgamma when not NULL).
 *   fr_bnl_bwd      G[M, N] = the gradient at the output y of layer `self` (Z, fin, gamma, beta, act): dY for the top layer, the
 *                   dA of the call above otherwise.  dZ = invstd gamma (G s - sum(G s) / M - xhat sum(G s xhat) / M) -> dZ_out
 *                   (may be NULL); dA[M, K] = dZ W back through the dropout of the layer's input (below->drop_*; drop_used =
 *                   the counter value fr_bnl_fwd recorded); with below->Z != NULL also the sums of the layer below (from dA
 *                   and that layer's Z) -> sums_below / dgamma_below / dbeta_below.
 * ws: fr_bnl_workspace_bytes(M, width of the statistics written); ticket: TWO zero-initialised device words the launches of a
 * stream share (zero again when a launch ends).  Widths: K % 32 == 0, K, N <= 256, otherwise FR_EUNSUPPORTED (the
 * layered entries take every shape).  The same operations in the same order as the layered form (csrc/mlp_bn_math.hpp, the
 * products' summation order, the statistics' chunks): bit-identical where both forms apply (every width a multiple of 32), and
 * the same dropout patterns (tests/test_mlp_hip.py). */
typedef struct fr_bn_src {
    const float* Z;
    const float* fin;
    const float* gamma;
    const float* beta;
    int32_t act;
    float drop_p;
    uint64_t drop_seed;
    uint64_t drop_off;
} fr_bn_src;
FR_API size_t fr_bnl_workspace_bytes(int64_t M, int32_t width);
FR_API int fr_bnl_fwd(const fr_bn_src* in, int64_t M, int32_t K, const float* W, const float* bias, int32_t N, float* Z, float* A_out,
                      float eps, float momentum, float* running_mean, float* running_var, int64_t* nbt, int32_t nbt_inc,
                      float* fin_out, void* ws, size_t ws_bytes, uint32_t* ticket, const uint64_t* drop_state, uint64_t* drop_used,
                      uint64_t* drop_tick, void* stream);
FR_API int fr_bnl_out(const fr_bn_src* src, int64_t M, int32_t N, float* Y, void* stream);
FR_API int fr_bnl_bwd_top(const float* dY, const fr_bn_src* top, int64_t M, int32_t N, float* sums, float* dgamma, float* dbeta,
                          void* ws, size_t ws_bytes, uint32_t* ticket, void* stream);
FR_API int fr_bnl_bwd(const float* G, const fr_bn_src* self, const float* sums, int64_t M, int32_t N, const float* W, int32_t K,
                      float* dZ_out, const fr_bn_src* below, const uint64_t* drop_used, float* dA, float* sums_below,
                      float* dgamma_below, float* dbeta_below, void* ws, size_t ws_bytes, uint32_t* ticket, void* stream);

