/* mobocmf_hip.h -- C-ABI of the MI355X (gfx950) MFDGP hot path.
 *
 * Drop-in boundary for the variational multi-fidelity deep-GP layer + ELBO of fernandezdaniel/MOBOCMF.
 * The reference has no FFI of its own (it is pure Python on GPyTorch); these entry points replace the
 * arithmetic that the reference reaches through the following call sites (paths under /root/reference):
 *
 *   mobocmf_layer_forward / _backward   mobocmf/layers/mfdgp_hidden_layer.py:232-243 (prior over cat[Z,X]),
 *                                       :286 -> gpytorch UnwhitenedVariationalStrategy.forward (Gram, Cholesky,
 *                                       triangular solves, predictive moments) and kl_divergence(); backward =
 *                                       what torch autograd does for blackbox_mfdgp_fitter.py:168
 *   mobocmf_propagate_forward/_backward mobocmf/layers/mfdgp_hidden_layer.py:263-274 (hidden-sample propagation)
 *   mobocmf_elbo_data_forward/_backward mobocmf/mlls/variational_elbo_mf.py:31-35 + GaussianLikelihood.expected_log_prob
 *   mobocmf_acq_moments_forward/_backward  mobocmf/models/mfdgp.py:258-260 (moments over the S samples)
 *   mobocmf_jes_forward                 mobocmf/acquisition_functions/JESMOC_MFDGP.py:52
 *   mobocmf_predictive_covariance       eval branch of the variational strategy: K_nn - K_nm K_mm^-1 K_mn + C^T C
 *                                       (the full covariance GPyTorch materialises in .eval(), JESMOC_MFDGP.py:42)
 *   mobocmf_adam_step                   torch.optim.Adam.step at blackbox_mfdgp_fitter.py:169
 *
 * Conventions: all pointers are DEVICE pointers to row-major float64 unless stated; sizes are explicit;
 * every function enqueues work on `stream` and returns immediately (no allocation, no free, no host sync;
 * re-entrant, one process per GPU).  The library holds NO mutable state between calls: every kernel-selection knob
 * travels with the call in a `mobocmf_tuning` (a pointer in the layer descriptor, or an argument of the standalone
 * product entry points; NULL = the compiled-in defaults), block-activity arrays and probe events are explicit arguments,
 * and the workspace-size queries take the same descriptor / tuning as the launch they size for.  Two host threads with
 * different tunings on different streams do not interact.  (One write-once bit per kernel instantiation and device
 * remembers that the kernel's dynamic-LDS attribute was set on that device.)  Return value: MOBOCMF_OK or an error
 * code; a non-positive-definite K_mm is reported through the device word `info` (0 = OK, k>0 = pivot k
 * failed), mirroring LAPACK potrf / torch.linalg.cholesky_ex, so the caller may retry with more jitter
 * (gpytorch psd_safe_cholesky semantics) without a sync on the fast path.
 */
#ifndef MOBOCMF_HIP_H
#define MOBOCMF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mobocmf_stream_t; /* hipStream_t */

#define MOBOCMF_MAX_D 32    /* input dimensions a layer accepts (x columns; the Gram kernels keep a row in registers) */
#define MOBOCMF_MAX_XDIV 48 /* sample replicas per base row (num_samples_for_acquisition / _for_training) */

enum {
    MOBOCMF_OK = 0,
    MOBOCMF_BAD_ARG = 1,
    MOBOCMF_WORKSPACE_TOO_SMALL = 2,
    MOBOCMF_HIP_ERROR = 3,
    MOBOCMF_NOT_PD = 4, /* only returned by mobocmf_check_info (host-side, synchronising) */
    MOBOCMF_BAD_ARCH = 5
};

/* Kernel-selection knobs of one call.  Fill with mobocmf_tuning_init() (compiled-in defaults, struct_size), change what is
 * needed, and pass the pointer with the call; NULL everywhere means the defaults.  Knobs never change results beyond
 * summation order; they exist for size sweeps, A/B timing and the parity tests.  A workspace-size query and the launch it
 * sizes for must be given the same values (syrk_workgroups changes the slab count, tile_rows the partial-row count): the
 * launches re-derive their layout from the tuning they receive and return MOBOCMF_WORKSPACE_TOO_SMALL on a mismatch. */
typedef struct mobocmf_tuning {
    uint32_t struct_size;    /* sizeof(mobocmf_tuning) of the caller (versioning; set by mobocmf_tuning_init) */
    int32_t small_gemm_max;  /* largest dimension of an M x M product on the small-operand kernel (default 384, <= 512) */
    int32_t small_panel_max; /* largest K of an M x N' panel product on the whole-block panel kernel (default 512, <= 512) */
    int32_t tile_rows;       /* tile height of the M x N' panel products: 0 = by shape (default) | 64 | 128 */
    int32_t pair_mode;       /* one workgroup does the row blocks (p, n-1-p) of a triangular product: 0 auto | 1 never | 2 always */
    int32_t mid_gemm_max;    /* largest dimension of a plain M x M product on the mid-size kernel (default 1024; 0 = off) */
    int32_t mid_gemm_waves;  /* its form: 32 (default) = 32 x 64 tiles, 64-k stages | 8 | 4 = 64 x 64 tiles on 8 / 4 wavefronts */
    int32_t syrk_workgroups; /* workgroups a k-sliced weighted syrk may occupy: 0 = by shape (default) | 16..4096 */
    int32_t sparse_backward; /* 1 (default): a layer backward skips 128-column blocks whose upstream gradients are all
                              * exactly zero (the rows of other fidelities, variational_elbo_mf.py:33-38) -- found on the
                              * device, same numbers; 0: the dense backward (A/B timing, parity tests) */
    int32_t potrf_cols;      /* the blocked Cholesky: 0 (default) = all 64-column steps in ONE launch where it applies (128 < M <= 1024;
                              * panel and trailing-update workgroups resident together), the four-column panel kernel elsewhere |
                              * 4 | 1 = one launch pair per 64 columns, 4 / 1 columns per hand-over of its panel kernel */
} mobocmf_tuning;
int mobocmf_tuning_init(mobocmf_tuning* t);

/* One variational GP layer.
 * kind 0 (first layer):  k = alpha * RBF_ard(x, x')                              hyp = [alpha, ls[0..d)]
 * kind 1 (layer >= 1):   k = a1*RBF(x,x';ls1) * (nu*f*f' + af*RBF(f,f';lsf)) + a2*RBF(x,x';ls2)
 *                        hyp = [a1, af, nu, a2, lsf, ls1[0..d), ls2[0..d)]
 * The data matrix X~ = [x[n / xdiv], f[n]] is never materialised: row n of the layer input uses row n/xdiv of
 * `x` (S-fold sample replication, mfdgp.py:248 / SURVEY F7) and f[n] (kind 1 only).  Inducing inputs Z~ = [Zx, zf].
 */
typedef struct {
    int32_t kind;    /* 0 | 1 */
    int32_t d;       /* columns of x and Zx, 1..MOBOCMF_MAX_D */
    int32_t M;       /* inducing points */
    int32_t xdiv;    /* 1..MOBOCMF_MAX_XDIV */
    int64_t Np;      /* rows through the layer (N'), Np % xdiv == 0 */
    int32_t branch;  /* 0: training branch (clamp(k_nn - q, 0)); 1: eval branch (no clamp) */
    int32_t want_dx; /* backward also returns d/dx (acquisition optimisation) */
    double jitter;   /* added to diag(K_mm); gpytorch variational_cholesky_jitter = 1e-6 */
    double min_var;  /* MultivariateNormal.variance clamp; gpytorch min_variance = 1e-10 */
    int32_t phase;   /* MOBOCMF_PHASE_*: which half of the layer call to run (0 = both) */
    int32_t reserved;
    const mobocmf_tuning* tuning; /* kernel-selection knobs of THIS call and of the size queries made with this descriptor;
                                   * NULL = defaults */
    void* const* probe_events;    /* diagnostic, NULL = off: MOBOCMF_PROBE_EVENTS hipEvent_t (created by the caller with timing
                                   * enabled; NULL entries are skipped) recorded on the call's stream around the
                                   * grid-filling launches of this layer's PANEL half:
                                   *   [9] Gram forward [0] A = L^-1 K [1] C = U^T A [2]  ...  [3] dA [4] ... [5] weighted
                                   *   syrk + slab reduction [6] ... [7] dK = L^-T dA [8] Gram backward [10]
                                   * -- per-kernel durations inside a training step without a profiler (bench.py:
                                   * per_kernel_instep_ms).  Not under stream capture. */
} mobocmf_layer_desc;
#define MOBOCMF_PROBE_EVENTS 11

/* A layer call has two halves that only meet in `saved` (forward) / `scratch` (backward):
 *   CHAIN  the M x M work that depends on the parameters alone: K_mm, its Cholesky and inverse, U = L^-1 L_S,
 *          a = L^-1 m, the KL (forward); the M x M backward chain, Cholesky backward, Gram backward of K_mm (backward);
 *   PANEL  the M x N' work: K_mn, A, C, moments (forward); dA, the weighted syrk, dK, Gram backward of K_mn (backward).
 * Forward order: CHAIN then PANEL.  Backward order: PANEL then CHAIN (same private `scratch` for both calls).
 * A caller may run the CHAIN halves of several layers on another stream, overlapping the latency-bound chain of one
 * layer with the grid-filling PANEL work of another.  Pointers a half does not touch may be NULL.
 * In a split backward the CHAIN half OVERWRITES g_zf / g_hyp with its own contribution (the caller adds the PANEL
 * half's); CHAIN_ONLY additionally treats the PANEL half's contribution as zero (no upstream mean/var gradient). */
#define MOBOCMF_PHASE_ALL 0
#define MOBOCMF_PHASE_CHAIN 1
#define MOBOCMF_PHASE_PANEL 2
#define MOBOCMF_PHASE_CHAIN_ONLY 3   /* backward only */
#define MOBOCMF_PHASE_PANEL_INPUTS 4 /* PANEL half with the parameters held constant: backward yields g_f / g_x (and the
                                      * K_mn share of g_zf / g_hyp) but skips the M x M contractions the CHAIN half would
                                      * consume -- acquisition optimisation against a fixed model */

int mobocmf_version(void);
/* 1 if the current HIP device is gfx950. */
int mobocmf_device_arch_ok(void);

/* Bytes of the `saved` buffer (forward -> backward state: L, L^-1, U, K_mn, A, C, ...) and of the scratch
 * buffer (dead after each call). */
int mobocmf_layer_workspace_bytes(const mobocmf_layer_desc* desc, size_t* saved_bytes, size_t* scratch_bytes);

/* Leading bytes of `saved` that hold the CHAIN half's state (L, L^-1, U, a, ...).  They do not depend on Np: a caller
 * whose parameters are fixed (acquisition optimisation) runs the CHAIN half once and copies these bytes to the front of
 * the `saved` buffer of every later PANEL call. */
int mobocmf_layer_chain_state_bytes(const mobocmf_layer_desc* desc, size_t* bytes);

/* mean[Np], var[Np] (clamped at min_var), kl[1] = KL(q(u) || p(u)), info[1]. */
int mobocmf_layer_forward(const mobocmf_layer_desc* desc, const double* x, const double* f, const double* Zx,
                          const double* zf, const double* hyp, const double* m, const double* L_S /* M x M, ld M */,
                          double* mean, double* var, double* kl, int32_t* info, void* saved, size_t saved_bytes,
                          void* scratch, size_t scratch_bytes, mobocmf_stream_t stream);

/* Gradients of  sum(g_mean*mean) + sum(g_var*var) + g_kl[0]*kl.
 * Outputs (overwritten): g_f[Np] (kind 1), g_zf[M] (kind 1), g_hyp[hyp_len], g_m[M], g_LS[M x M] (lower),
 * g_x[(Np/xdiv) x d] (only if want_dx).  g_kl is a device scalar. */
int mobocmf_layer_backward(const mobocmf_layer_desc* desc, const double* x, const double* f, const double* Zx,
                           const double* zf, const double* hyp, const double* m, const double* L_S,
                           const double* g_mean, const double* g_var, const double* g_kl, double* g_f, double* g_zf,
                           double* g_hyp, double* g_m, double* g_LS, double* g_x, void* saved, size_t saved_bytes,
                           void* scratch, size_t scratch_bytes, mobocmf_stream_t stream);

/* ---- The same layer call for SEVERAL layers of one model (n <= 4, equal M): the CHAIN halves of all layers run as ONE
 * z-batched sequence of launches.  A chain is a serial string of latency-bound M x M kernels (Cholesky panels, triangular
 * inverse, a dozen M x M products); the layers' chains are independent of each other (Z~_l = [Z_x, m_{l-1}] depends on
 * parameters only), so batching them divides the length of that string by the number of layers.
 * Memory: one CHAIN BLOCK per layer (mobocmf_chain_block_bytes; all M x M-sized: the chain state L, L^-1, U, ... that the
 * PANEL halves read, H / Hc / da that the PANEL backward leaves for the chain backward, and the chain's scratch), the n
 * blocks `block_stride` bytes apart (a multiple of 256, >= the block size) starting at `blocks`, which holds `blocks_bytes`
 * >= n * block_stride bytes (checked: MOBOCMF_WORKSPACE_TOO_SMALL); per layer a PANEL `saved`
 * and `scratch` (mobocmf_panel_workspace_bytes).  Call order of a step: layers_chain_forward; layer_panel_forward per layer
 * (bottom up); layer_panel_backward per layer (top down); layers_chain_backward.  Arrays of pointers are HOST arrays of n
 * entries (device pointers inside).  The PANEL backward OVERWRITES g_hyp / g_zf with the K_mn share, the chain backward
 * OVERWRITES its own g_hyp / g_zf arguments with the K_mm share (the caller adds the two).  had_panel[l] = 0: layer l got no
 * upstream mean / var gradient (only its KL was differentiated); 1: its PANEL backward ran; 2: it ran AND the g_hyp[l] / g_zf[l]
 * passed to the chain backward already hold its share -- the chain backward adds to them instead of overwriting.
 * had_panel[l] |= 4 (l >= 1, kind 1): zf[l] IS the variational mean m[l-1] of the previous layer of this call (the reference's
 * Z~_l = [Z_x, m_{l-1}], mfdgp_hidden_layer.py:555-556): the complete gradient of zf[l] -- this call's K_mm share plus, under
 * 2, the PANEL share found in g_zf[l] -- is ADDED to g_m[l-1] and g_zf[l] is not written; the caller hands g_m[l-1] to m[l-1]
 * and nothing to zf[l]. */
int mobocmf_chain_block_bytes(const mobocmf_layer_desc* desc, size_t* block_bytes, size_t* state_bytes);
int mobocmf_panel_workspace_bytes(const mobocmf_layer_desc* desc, size_t* saved_bytes, size_t* scratch_bytes);
int mobocmf_layers_chain_forward(int32_t n, const mobocmf_layer_desc* const* desc, const double* const* Zx,
                                 const double* const* zf, const double* const* hyp, const double* const* m,
                                 const double* const* L_S, double* const* kl, int32_t* const* info, void* blocks,
                                 size_t block_stride, size_t blocks_bytes, mobocmf_stream_t stream);
int mobocmf_layers_chain_backward(int32_t n, const mobocmf_layer_desc* const* desc, const double* const* Zx,
                                  const double* const* zf, const double* const* hyp, const double* const* g_kl,
                                  const int32_t* had_panel, double* const* g_zf, double* const* g_hyp, double* const* g_m,
                                  double* const* g_LS, void* blocks, size_t block_stride, size_t blocks_bytes,
                                  mobocmf_stream_t stream);
int mobocmf_layer_panel_forward(const mobocmf_layer_desc* desc, const double* x, const double* f, const double* Zx,
                                const double* zf, const double* hyp, double* mean, double* var, void* chain_block,
                                size_t block_bytes, void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes,
                                mobocmf_stream_t stream);
/* desc->phase == MOBOCMF_PHASE_PANEL_INPUTS: parameters held constant (no H / Hc / da for a chain backward). */
int mobocmf_layer_panel_backward(const mobocmf_layer_desc* desc, const double* x, const double* f, const double* Zx,
                                 const double* zf, const double* hyp, const double* g_mean, const double* g_var,
                                 double* g_f, double* g_zf, double* g_hyp, double* g_x, void* chain_block,
                                 size_t block_bytes, void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes,
                                 mobocmf_stream_t stream);

/* cov[Np x Np] = K_nn - A^T A + C^T C (A = L^-1 K_mn, C = U^T A): the full predictive covariance of the eval branch of the
 * variational strategy (the dense matrix GPyTorch materialises in .eval(), JESMOC_MFDGP.py:42 -> mfdgp.py:248), written as
 * the full symmetric matrix with leading dimension ldcov >= Np.  Only the layer's CHAIN state is read (`chain_state` = the
 * first mobocmf_layer_chain_state_bytes of the `saved` buffer a forward / CHAIN-half call filled: L^-1 and U), so a frozen
 * chain serves any number of input batches.  The contractions over the M inducing points are symmetric rank-M updates on
 * the MFMA (lower 128-tiles only), one column panel at a time: the scratch (mobocmf_predictive_covariance_workspace_bytes)
 * grows with Np x panel width, not Np^2, and Np is not capped.  Ordinary layer scratch sizes do not include it. */
int mobocmf_predictive_covariance_workspace_bytes(const mobocmf_layer_desc* desc, size_t* scratch_bytes);
int mobocmf_predictive_covariance(const mobocmf_layer_desc* desc, const double* x, const double* f, const double* Zx,
                                  const double* zf, const double* hyp, double* cov, int64_t ldcov, const void* chain_state,
                                  size_t chain_state_bytes, void* scratch, size_t scratch_bytes, mobocmf_stream_t stream);

/* Dense prior covariance K[i][j] = k([x1[i], f1[i]], [x2[j], f2[j]]) of one layer's kernel (the lazy prior of
 * MFDGPHiddenLayer.forward, mfdgp_hidden_layer.py:232-243, evaluated): x1 [n1 x d], x2 [n2 x d], f1 [n1] / f2 [n2] (kind 1,
 * else NULL).  K is row-major with ldk >= n2 and must hold round_up(n1, 32) rows (rows >= n1 are written as zeros). */
int mobocmf_gram_forward(int32_t kind, int32_t d, const double* x1, const double* f1, int64_t n1, const double* x2,
                         const double* f2, int64_t n2, const double* hyp, double* K, int64_t ldk, mobocmf_stream_t stream);

/* One random-Fourier-feature function sample of a layer evaluated at n points (mfdgp_hidden_layer.py:326-337, :402-444; the
 * Pareto-grid evaluation of moop.py:232-272), without materialising the F x n feature matrix:
 *   kind 0:  out[i] = sum_j theta[j] s0 cos(W1[j].x_i + b1[j])
 *   kind 1:  out[i] = sum_j theta[j] s0 fprev[i] cos(W1[j].x_i + b1[j]) + theta[F+j] s1 cos(W1[j].x_i + Wf[j] fprev[i] + b1[j])
 *                          + theta[2F+j] s2 cos(W2[j].x_i + b2[j])
 * x [n x d], W1 / W2 [F x d], b1 / b2 / Wf [F], theta [F] or [3F]; fprev = the previous layer's sample at the same points
 * (the layer recursion is one call per layer).  s0 = sqrt(2 alpha / F) (kind 0) resp. sqrt(2 a1 nu / F), s1 = sqrt(2 a1 af / F),
 * s2 = sqrt(2 a2 / F). */
int mobocmf_rff_eval(int32_t kind, int32_t d, int32_t F, int64_t n, const double* x, const double* fprev, const double* W1,
                     const double* b1, const double* Wf, const double* W2, const double* b2, const double* theta, double s0,
                     double s1, double s2, double* out, mobocmf_stream_t stream);

/* f~[n] = mean[n/div] + sqrt(var[n/div]) * eps[n],  n < n_out  (mfdgp_hidden_layer.py:263-274). */
int mobocmf_propagate_forward(const double* mean, const double* var, const double* eps, double* f_out, int64_t n_out,
                              int32_t div, mobocmf_stream_t stream);
/* The same with eps ~ N(0, 1) drawn inside the launch (the reference draws it with torch.normal, mfdgp_hidden_layer.py:274):
 * counter-based Philox4x32-10 keyed by (seed, call counter, row index), Box-Muller in float64.  rng_state = 3 device int64
 * {seed, calls, ticket} owned by the caller (ticket zero before the first call; every call leaves it zero and advances
 * `calls` by one, so a captured step replays with fresh eps).  eps_out[n_out] receives the draw (the backward pass needs it). */
int mobocmf_propagate_rng_forward(const double* mean, const double* var, int64_t* rng_state, double* f_out, double* eps_out,
                                  int64_t n_out, int32_t div, mobocmf_stream_t stream);
/* g_mean[n_out/div], g_var[n_out/div] from g_f[n_out]. */
int mobocmf_propagate_backward(const double* var, const double* eps, const double* g_f, double* g_mean, double* g_var,
                               int64_t n_out, int32_t div, mobocmf_stream_t stream);
/* Prefix propagation: the previous layer holds n_prev >= n_out/div rows of which only the FIRST n_out/div were propagated
 * (the next layer is evaluated on a prefix of the batch -- the rows that can reach the loss, see mobocmf_elbo_forward);
 * g_mean / g_var [n_prev] get zeros beyond the prefix.  add_mean / add_var [n_prev] (each may be NULL): gradients the same
 * moments receive from their other consumer -- the ELBO data term of the previous layer's own fidelity
 * (variational_elbo_mf.py:33-35) -- added in this launch (autograd would spend one element-wise launch per sum). */
int mobocmf_propagate_backward_prefix(const double* var, const double* eps, const double* g_f, double* g_mean, double* g_var,
                                      int64_t n_out, int32_t div, int64_t n_prev, const double* add_mean,
                                      const double* add_var, mobocmf_stream_t stream);

/* out[0] = (1/div) * sum_{n : fid[n/div] == level} -0.5 * (((y[n/div]-mean[n])^2 + var[n]) / tau + log tau + log 2pi)
 * tau is a device scalar.  n < n_rows. */
int mobocmf_elbo_data_forward(const double* mean, const double* var, const double* y, const double* fid,
                              const double* tau, double level, int64_t n_rows, int32_t div, double* out,
                              void* scratch, size_t scratch_bytes, mobocmf_stream_t stream);
/* g_mean[n_rows], g_var[n_rows], g_tau[1] scaled by the device scalar g_out. */
int mobocmf_elbo_data_backward(const double* mean, const double* var, const double* y, const double* fid,
                               const double* tau, double level, int64_t n_rows, int32_t div, const double* g_out,
                               double* g_mean, double* g_var, double* g_tau, void* scratch, size_t scratch_bytes,
                               mobocmf_stream_t stream);

/* GPyTorch's shortcut branch (inputs identical to the inducing inputs: q(u) itself): the marginal
 * variances var[i] = max(sum_{j<=i} L_S[i][j]^2, min_var) of S = L_S L_S^T (L_S M x M row-major, lower triangle read), and
 * the gradient g_LS[i][j] = 2 L_S[i][j] g_var[i] (j <= i, var[i] above the floor; 0 elsewhere). */
int mobocmf_shortcut_var_forward(const double* L_S, int32_t M, double min_var, double* var, mobocmf_stream_t stream);
int mobocmf_shortcut_var_backward(const double* L_S, const double* var, const double* g_var, int32_t M, double min_var,
                                  double* g_LS, mobocmf_stream_t stream);

/* ELBO tail (variational_elbo_mf.py:37-51): out2[0] = sum data_terms - scale * sum kls, out2[1] = scale * sum kls
 * (scale = batch / num_data).  Arrays of device-scalar pointers are HOST arrays, at most 8 entries each.
 * Backward: g2[0] = gradient w.r.t. every data term = g_elbo; g2[1] = w.r.t. every KL = scale * (g_skl - g_elbo);
 * g_elbo / g_skl are device scalars, either may be NULL (= 0). */
int mobocmf_elbo_combine_forward(int32_t n_data, const double* const* data_terms, int32_t n_kl, const double* const* kls,
                                 double scale, double* out2, mobocmf_stream_t stream);
int mobocmf_elbo_combine_backward(const double* g_elbo, const double* g_skl, double scale, double* g2,
                                  mobocmf_stream_t stream);

/* The whole ELBO of variational_elbo_mf.py:24-51 in one call (a reduction launch + a one-block tail; the same in backward;
 * the per-fidelity functions above take two launches per fidelity + mobocmf_elbo_combine): for every layer l < L with mean[l]
 * non-NULL the masked expected log-likelihood of the rows with fid == l (mean[l] / var[l] hold B * div[l] rows, row i belongs
 * to y[i / div[l]]; averaged over the div[l] samples of a row), noise tau_l = lo[l] + (hi[l] - lo[l]) sigmoid(raw_noise[l][0])
 * (hi[l] <= lo[l]: raw_noise[l][0] is tau itself); n_kl KL scalars; scale = batch / num_data:
 *   out3[0] = sum data terms - scale * sum kls,  out3[1] = scale * sum kls,  out3[2] = -out3[0]  (the loss).
 * rows (HOST array of L entries, or NULL = B everywhere): layer l holds only the FIRST rows[l] <= B rows of the batch
 * (rows[l] * div[l] entries).  The reference evaluates every layer at every row and then masks (:33-38); a row of fidelity
 * f reaches the loss only through layers 0..f, so with the batch ordered by descending fidelity layer l needs the prefix
 * of rows with fidelity >= l and nothing else (mobocmf_amd.util.graphed_step orders the batch this way once).
 * Arrays of pointers / per-layer values are HOST arrays of L (n_kl) entries, L, n_kl <= 8.  scratch >= 8 * 512 * 8 bytes.  Backward: g_mean[l] / g_var[l] (B * div[l]), g_raw_noise[l] (w.r.t. the RAW parameter; may be
 * NULL per layer), g_kl[0] = the gradient w.r.t. every KL = scale * (g_skl - g_elbo); g_elbo / g_skl device scalars or NULL. */
int mobocmf_elbo_forward(int32_t L, const double* const* mean, const double* const* var, const int32_t* div,
                         const double* const* raw_noise, const double* lo, const double* hi, const double* y,
                         const double* fid, int64_t B, const int64_t* rows, int32_t n_kl, const double* const* kls, double scale,
                         double* out3, void* scratch, size_t scratch_bytes, mobocmf_stream_t stream);
int mobocmf_elbo_backward(int32_t L, const double* const* mean, const double* const* var, const int32_t* div,
                          const double* const* raw_noise, const double* lo, const double* hi, const double* y,
                          const double* fid, int64_t B, const int64_t* rows, double scale, const double* g_elbo,
                          const double* g_skl, double* const* g_mean, double* const* g_var, double* const* g_raw_noise,
                          double* g_kl, void* scratch, size_t scratch_bytes, mobocmf_stream_t stream);

/* The same pair with the noise given as the RAW parameter of an Interval constraint (mfdgp.py:116):
 * tau = lo + (hi - lo) * sigmoid(raw_noise[0]) is evaluated inside the kernels and g_tau is the gradient w.r.t. the raw
 * parameter (hi <= lo: raw_noise is taken as tau itself, i.e. the plain functions above). */
int mobocmf_elbo_data_interval_forward(const double* mean, const double* var, const double* y, const double* fid,
                                       const double* raw_noise, double lo, double hi, double level, int64_t n_rows,
                                       int32_t div, double* out, void* scratch, size_t scratch_bytes,
                                       mobocmf_stream_t stream);
int mobocmf_elbo_data_interval_backward(const double* mean, const double* var, const double* y, const double* fid,
                                        const double* raw_noise, double lo, double hi, double level, int64_t n_rows,
                                        int32_t div, const double* g_out, double* g_mean, double* g_var, double* g_tau,
                                        void* scratch, size_t scratch_bytes, mobocmf_stream_t stream);

/* mus[t] = mean_s mu~[t*S+s];  vars[t] = mean_s(var~ + mu~^2) - mus^2   (mfdgp.py:258-260). */
int mobocmf_acq_moments_forward(const double* mu_t, const double* var_t, double* mus, double* vars, int64_t T,
                                int32_t S, mobocmf_stream_t stream);
int mobocmf_acq_moments_backward(const double* mu_t, const double* g_mus, const double* g_vars, double* g_mu_t,
                                 double* g_var_t, int64_t T, int32_t S, mobocmf_stream_t stream);
/* acq[t] = 0.5 * max(log v_uncond[t] - log v_cond[t], 0)   (JESMOC_MFDGP.py:52). */
int mobocmf_jes_forward(const double* v_uncond, const double* v_cond, double* acq, int64_t T, mobocmf_stream_t stream);

/* Fused Adam update (torch.optim.Adam defaults: no weight decay, no amsgrad) of one flat parameter
 * segment; `step` is the 1-based step count.  mask (may be NULL) = 0 freezes an element. */
int mobocmf_adam_step(double* param, const double* grad, double* exp_avg, double* exp_avg_sq, const double* mask,
                      int64_t n, double lr, double beta1, double beta2, double eps, int64_t step,
                      mobocmf_stream_t stream);

/* The same update for ALL parameter tensors of a model in one launch (<= 40 tensors per launch; more are chunked).
 * The arrays of pointers / sizes are HOST arrays (copied into the kernel arguments, so a captured graph replays them);
 * step_state is ONE device int64 holding the number of completed steps: the launch uses step_state[0] + 1 for the bias
 * corrections and a trailing one-thread launch increments it (no host-side step count: capturable).  Tensors without a
 * gradient are simply left out by the caller. */
int mobocmf_adam_multi(int32_t n_tensors, double* const* params, const double* const* grads, double* const* exp_avg,
                       double* const* exp_avg_sq, const int64_t* sizes, double lr, double beta1, double beta2, double eps,
                       int64_t* step_state, mobocmf_stream_t stream);

/* Constrained hyper-parameters of a layer in one launch: out = softplus of n_tensors raw parameter tensors (HOST arrays of
 * device pointers / sizes, copied into the kernel arguments: capturable), concatenated in the given order -- the `hyp`
 * vector of mobocmf_layer_forward built from GPyTorch's raw_outputscale / raw_lengthscale / raw_variance parameters with
 * their Positive() constraints (gpytorch.constraints.Positive = softplus; mfdgp_hidden_layer.py:43-47,68-88).  The
 * backward writes one gradient tensor per raw tensor.  n_tensors <= 16. */
int mobocmf_softplus_pack(int32_t n_tensors, const double* const* raw, const int32_t* sizes, double* out,
                          mobocmf_stream_t stream);
int mobocmf_softplus_pack_backward(int32_t n_tensors, const double* const* raw, const int32_t* sizes, const double* g_out,
                                   double* const* g_raw, mobocmf_stream_t stream);
/* The same with one upstream-gradient pointer PER TENSOR (g_out[i]: sizes[i] doubles, NULL = zero): for a caller that packs
 * the hyper-parameters of several layers in one launch and hands the segments out as separate tensors. */
int mobocmf_softplus_pack_backward_v(int32_t n_tensors, const double* const* raw, const int32_t* sizes,
                                     const double* const* g_out, double* const* g_raw, mobocmf_stream_t stream);

/* ---- Conditioned training: the theta / omega factor losses of blackbox_mfdgp_fitter.py:227-243 and the glue
 * around them -- a few hundred flops that cost ~100 framework launches per iteration of a launch-bound loop.
 *
 * mobocmf_cond_factors_forward:  loss[0] = sum_{p < P, t < T} [ coef_c c(p,t) + coef_1mc (1 - c(p,t)) ] with
 *   c(p,t) = prod_k Phi((cs_mean[k][t] - thresholds[k]) / sqrt(cs_var[k][t]))
 *          * prod_j Phi((pareto_front[p][j] - fs_mean[j][t]) / sqrt(fs_var[j][t]))
 *   (n_obj, n_con <= 8; one row pointer per objective / constraint, T entries each; pareto_front [P x n_obj] row-major).
 *   omega factors (:235-243): coef_c = log eps, coef_1mc = log(1 - eps), P = Pareto points, T = the x~ points.
 *   theta factors (:227-233): n_obj = 0, n_con = 1, P = 1, T = Pareto points, coef_c = log(1 - eps), coef_1mc = log eps.
 *   g_*: d loss / d (every input entry) for an upstream gradient of 1, formed in the same launch.
 * mobocmf_scale_segments:   out_i[e] = g[0] * coef_i * in_i[e]   (g: device scalar, NULL = 1; coef NULL = 1): the backward of
 *   anything whose forward formed its own gradient.
 * mobocmf_gather_segments:  out = segments back to back, zeros where in[i] is NULL: the backward of a split into row ranges.
 * mobocmf_scalar_combine:   out[0] = sum_i coef[i] * x[i][0]: a loss assembled from scalar terms (coef: HOST array).
 * n <= 32 segments; sizes / coef are HOST arrays. */
int mobocmf_cond_factors_forward(int32_t n_obj, int32_t n_con, int32_t P, int32_t T, const double* const* fs_mean,
                                 const double* const* fs_var, const double* const* cs_mean, const double* const* cs_var,
                                 const double* pareto_front, const double* thresholds, double coef_c, double coef_1mc,
                                 double* loss, double* const* g_fs_mean, double* const* g_fs_var, double* const* g_cs_mean,
                                 double* const* g_cs_var, mobocmf_stream_t stream);
int mobocmf_scale_segments(int32_t n, const double* const* in, double* const* out, const int64_t* sizes, const double* coef,
                           const double* g, mobocmf_stream_t stream);
int mobocmf_gather_segments(int32_t n, const double* const* in, const int64_t* sizes, double* out, mobocmf_stream_t stream);
int mobocmf_scalar_combine(int32_t n, const double* const* x, const double* coef, double* out, mobocmf_stream_t stream);

/* ---- The whole ELBO step of SMALL surrogates in ONE launch (blackbox_mfdgp_fitter.py:161-171 at the reference's own sizes:
 * M = N = tens of points, examples/example_acquisition_mfdgp_forrester/...py:51-62).  At those sizes a step through the layer
 * entry points above is ~50 dependent launches of ~4.7 us each whatever they compute.  Here ONE workgroup per surrogate runs
 * zero_grad + MFDGP.forward + VariationalELBOMF + backward + Adam (torch.optim.Adam semantics) as a sequence of
 * workgroup-barrier-separated phases: the M x M chain lives in LDS, the M x N' panels in `work`; several surrogates = several
 * workgroups of the same launch.  Same algebra and the same results as the layer path, rows ordered by DESCENDING
 * fidelity with layer l evaluated on the first rows[l] of them.
 *
 * Limits: L <= 3 layers, M <= 32 inducing points shared by all layers (Z~_l = [Zx, m_{l-1}], mfdgp_hidden_layer.py:555-556),
 * d <= 8, train branch only.  mobocmf_tiny_model is read by the kernel from DEVICE memory (`dev_models`: the caller uploads the
 * array once); `host_models` is the same array in host memory, used for validation and launch geometry only.
 *   raw[l][s]: the raw (unconstrained, softplus) kernel parameters in the packed order of the hyper-parameter vector above --
 *     layer 0: s = 0 outputscale (1), s = 1 lengthscales (d); layers >= 1: s = 0..4 a1, af, nu, a2, lsf (1 each), s = 5 ls1 (d),
 *     s = 6 ls2 (d).  m[l] (M), L_S[l] (M x M row-major, lower triangle used), raw_noise[l] (Interval(noise_lo, noise_hi)).
 *   trainable[l]: bit s (< 7) raw[l][s], bit 7 m, bit 8 L_S, bit 9 raw_noise -- cleared bits are left untouched by the update.
 *   rng[l] (l >= 1): {seed, calls, ticket} of the layer's eps stream as in mobocmf_propagate_rng_forward (same draws for the
 *     same state; `calls` advances by one per step); eps[l] != NULL replaces the draw (rows[l] * S values).
 *   Flat layout of grad / adam_m / adam_v (mobocmf_tiny_flat_len doubles): per layer [packed hyper-parameters | m | L_S], then
 *     raw_noise of every layer.  grad (optional) receives d(-ELBO)/d(raw parameters) of the step.
 *   out[0] = ELBO, out[1] = scaled KL, out[2] = -ELBO (before the update); info[l] = 0 or the failed Cholesky pivot (1-based).
 * do_update = 0: gradients only (no parameter, optimiser or rng-counter write); 1: the step; 2: forward only (out, top_mean /
 * top_var; draws the random rows of x) -- a conditioned iteration is mode 2, the factor launches on top_mean / top_var, mode 1;
 * 3: input gradients (the parameters are constants -- acquisition search, JESMOC_MFDGP.py:38-52): `grad` receives d / d x
 * (N x d) of <seed_gmean, top mean> + <seed_gvar, top var> (times seed_scale) and nothing else is written -- predictive
 * moments of fitted models are mode 2 with branch = 1, eps[l] = the layer's fixed samples tiled over the test points;
 * 4: mode 1 with the factor terms of the conditioned loss formed INSIDE the launch (mobocmf_tiny_coupling: ONE launch per
 * conditioned iteration instead of mode 2 + factor launches + mode 1). */
#define MOBOCMF_TINY_MAX_LAYERS 3
#define MOBOCMF_TINY_MAX_M 32
#define MOBOCMF_TINY_MAX_D 8
typedef struct mobocmf_tiny_model {
    int32_t L, M, d, S;
    int32_t N;                               /* batch rows, ordered by descending fidelity */
    int32_t rows[MOBOCMF_TINY_MAX_LAYERS];   /* rows[0] == N >= rows[1] >= ... >= 1 */
    uint32_t trainable[MOBOCMF_TINY_MAX_LAYERS];
    int32_t branch;                          /* 0: train branch (clamp(k_nn - q, 0)), 1: eval branch (no clamp), as the layers */
    const double* x;                         /* N x d */
    const double* y;                         /* N */
    const double* fid;                       /* N (levels as doubles, as VariationalELBOMF compares them) */
    const double* Zx;                        /* M x d */
    double* raw[MOBOCMF_TINY_MAX_LAYERS][7];
    double* m[MOBOCMF_TINY_MAX_LAYERS];
    double* L_S[MOBOCMF_TINY_MAX_LAYERS];
    double* raw_noise[MOBOCMF_TINY_MAX_LAYERS];
    double noise_lo[MOBOCMF_TINY_MAX_LAYERS], noise_hi[MOBOCMF_TINY_MAX_LAYERS];
    int64_t* rng[MOBOCMF_TINY_MAX_LAYERS];
    const double* eps[MOBOCMF_TINY_MAX_LAYERS];
    double* adam_m;
    double* adam_v;
    int64_t* steps_done;                     /* completed Adam steps (device word, advanced by the launch) */
    double* work;                            /* mobocmf_tiny_work_bytes */
    double* grad;                            /* optional */
    double* out;                             /* 3 doubles */
    int32_t* info;                           /* L words */
    double kl_scale;                         /* d loss / d KL_l: batch / num_data (variational_elbo_mf.py:44-47) */
    double jitter;
    /* Conditioned training (blackbox_mfdgp_fitter.py:270-343), all optional (NULL / 0: the plain ELBO step).  The loss of a
     * model is  -sum_rows row_weight[b] E_q[log p(y_b | f)] + kl_scale sum_l KL_l  + <seeds, top-layer moments>: rows scored
     * with another weight (the batch term's num_data / B, a Pareto point's 1) or not at all (weight 0: x~ rows, the Pareto
     * rows of a constraint), and gradients of terms formed OUTSIDE the launch (the theta / omega factors,
     * mobocmf_cond_factors_forward) entering at the top layer's columns. */
    const double* row_weight;                /* N (NULL: 1) */
    const double* seed_gmean;                /* rows[L-1] * S: d(outside term) / d mean of the top layer's columns (NULL: none;
                                              * mode 4 WRITES both arrays itself before its backward reads them) */
    const double* seed_gvar;
    double seed_scale;                       /* the outside term's coefficient in the loss (the fitter's -1) */
    double* top_mean;                        /* rows[L-1] * S: the top layer's moments, written by every mode (NULL: not) */
    double* top_var;
    int64_t* xrng;                           /* {seed, calls}: modes 2 / 4 draw rows [rand_row0, rand_row0 + rand_rows) of x from */
    int32_t rand_row0, rand_rows;            /* U(0,1) (the x~ of :276; x must be writable); modes 1 / 4 of model 0 advance calls */
    /* mode 4 (the whole conditioned iteration in ONE launch with an in-launch barrier): what couples the models, and this model's part in it */
    const struct mobocmf_tiny_coupling* coupling;
    int32_t role, role_index;                /* 0: objective role_index of the coupling, 1: constraint role_index */
} mobocmf_tiny_model;
/* The theta / omega factors of blackbox_mfdgp_fitter.py:227-243 over the models of one launch (device-resident, shared).
 * Every model holds, at its TOP layer with S = 1, the P Pareto points in columns [0, P) and the T points x~ in [P, P + T).
 * After its forward every workgroup publishes its top-layer moments, the workgroups of the launch meet at a barrier (an arrival
 * counter in device memory -- an ordinary launch, not a cooperative one: mobocmf_tiny_elbo_step checks n_models against the
 * device's occupancy for the kernel, n_models <= 64, 1 <= T <= 256), and every workgroup forms the factor gradients of ITS model -- the omega factors from all models' moments at x~, a constraint's theta
 * factors from its own at the Pareto points -- into its seed_gmean / seed_gvar before running its backward and update. */
typedef struct mobocmf_tiny_coupling {
    int32_t n_obj, n_con, P, T;
    int32_t obj_model[8], con_model[8];      /* index in the launch's model array of every objective / constraint */
    const double* front;                     /* P x n_obj (Pareto front, columns in objective order) */
    const double* thresholds;                /* n_con */
    double log_eps, log_1m_eps;
    double* losses;                          /* n_con + 1: the theta factor term of every constraint, then the omega term */
    int64_t* barrier;                        /* device word, zero-initialised ONCE by the caller: arrivals at the in-launch barrier.
                                              * Monotonic: launch k of the record waits for k * n_models arrivals, so ONE record
                                              * serves ONE launch at a time (launches of a record are ordered on one stream) and
                                              * always with the same n_models */
    int32_t* status;                         /* device word, zero-initialised by the caller, only ever OR'd by the launches:
                                              * bit 0 a workgroup gave up waiting at the barrier (it left its model untouched),
                                              * bit 1 the launch did not match the record (n_models, T, P): nothing was updated.
                                              * Sticky across launches -- the caller reads it whenever it checks (every 1000
                                              * iterations in the fitter) and clears it after rolling back */
    int32_t n_models;                        /* workgroups (= models) of the launches this record is for; the kernel checks */
    int32_t reserved;
} mobocmf_tiny_coupling;
int mobocmf_tiny_flat_len(const mobocmf_tiny_model* model, int64_t* len);
int mobocmf_tiny_work_bytes(const mobocmf_tiny_model* model, size_t* bytes);
int mobocmf_tiny_elbo_step(const mobocmf_tiny_model* host_models, const mobocmf_tiny_model* dev_models, int32_t n_models,
                           double lr, double beta1, double beta2, double eps, int32_t do_update, mobocmf_stream_t stream);

/* ---- The same step for MID-SIZE surrogates (M <= 128 inducing points: the reference's own BO loop runs M = N from 15 to 75,
 * examples/toy_synthetic_2D_JESMOCMF/toy_synthetic_2D_JESMOCMF.py:25,102-103,305-331 with mfdgp.py:295-298; BASELINE config 2 has
 * M = 128), still ONE launch per step for a whole group of surrogates, by SEVERAL workgroups per surrogate: the step is ~11
 * phases separated by an in-launch barrier of the surrogate's workgroups (arrival counter + agent-scope fences); every product
 * runs on v_mfma_f64_16x16x4_f64.  Same descriptor (mobocmf_tiny_model; M <= MOBOCMF_COOP_MAX_M, d <= 8, L <= 3), same flat
 * layout of grad / adam_m / adam_v (mobocmf_tiny_flat_len), same draws, same results as the layer path up to summation order;
 * `work` is sized by mobocmf_coop_work_bytes.  do_update: 0 gradients only | 1 the step | 2 forward only | 3 input gradients
 * (as mobocmf_tiny_elbo_step's mode 3: `grad` <- N x d, d/dx of <seed_gmean, top mean> + <seed_gvar, top var>; needs `grad`
 * and the seeds; nothing else is written) | 4 the conditioned iteration in one launch (mobocmf_tiny_coupling with n_models = the models of the launch; the barrier of its record is not
 * used: the whole grid meets on the launch's own sync words).
 * wgs_per_model: workgroups sharing one surrogate, 1..64, or 0 = chosen from the widest phase (at most 32); *wgs_used (may be
 * NULL) receives the choice.  Every workgroup of the launch must be resident at once (n_models * wgs_per_model <= what the
 * device holds of this kernel: checked, MOBOCMF_BAD_ARG otherwise -- an ordinary launch, not a cooperative one, so that it can
 * be captured into a graph).  sync_words: 16 * (n_models + 1) device int64, zero-initialised ONCE by the caller and then left to
 * the launches (monotonic arrival counters: a group of words serves one launch at a time, always with the same wgs_per_model
 * and n_models).  A wait that does not end within ~0.3 s is abandoned: info[0] = -1, out[2] = NaN, nothing is updated by that
 * workgroup. */
#define MOBOCMF_COOP_MAX_M 128
/* OR'd into do_update 2 or 3 of mobocmf_coop_elbo_step: the parameters are the ones of an earlier launch on the same `work`
 * (an acquisition search against fitted models, JESMOC_MFDGP.py:137-184): K_mm, its Cholesky and inverse, U, a and the KL stay as
 * that launch left them and are not formed again (35-65 us of a ~140 us launch).  The caller vouches for it. */
#define MOBOCMF_STEP_CHAIN_VALID 16
int mobocmf_coop_work_bytes(const mobocmf_tiny_model* model, size_t* bytes);
int mobocmf_coop_elbo_step(const mobocmf_tiny_model* host_models, const mobocmf_tiny_model* dev_models, int32_t n_models,
                           int32_t wgs_per_model, int64_t* sync_words, double lr, double beta1, double beta2, double eps,
                           int32_t do_update, int32_t* wgs_used, mobocmf_stream_t stream);

/* ---- Exact-GP comparison baselines (mobocmf/models/mfgp.py:24-141,145-184; mfgp_lin.py:101-189) on the layer's kernels.
 * The reference inherits exact inference from GPyTorch's ExactGP; here the Gram matrices come from mobocmf_gram_forward,
 * the multi-fidelity combination is one element-wise launch, the factorisation / triangular inverse are the variational
 * layer's (blocked Cholesky + MFMA), the predictive moments its triangular product with the column-statistics epilogue.
 *
 * mobocmf_mf_kernel_combine: out[i][j] = s1[i] s2[j] Ks[i][j] + ntab[min(l1[i], l2[j])] Kn[i][j] (+ diag if i == j), i < n1,
 *   j < n2; zero (identity on the diagonal when diag != 0) in the padding up to rows_p x cols_p.  Ks / Kn: the two ARD-RBF Gram
 *   matrices (signal / noise kernel, ld), l1 / l2 int32 fidelity levels, s1 / s2 per-row / per-column signal factors (NULL = 1),
 *   ntab[level] the noise factor (MFKernel: the level itself; MFKernel_lin: its rho polynomial).
 * mobocmf_exact_gp_factor: K = the n x n training covariance with the noise on its diagonal; leaves L, L^-1, a = L^-1 y in
 *   `state`, mll[0] = log N(y | 0, K), info = 0 or the failed pivot.
 * mobocmf_exact_gp_predict: mean[j] = k_j^T K^-1 y, var[j] = kss[j] - k_j^T K^-1 k_j for the nt columns k_j of Kts [n x nt].
 * Sizes from mobocmf_exact_gp_workspace_bytes (state for a given n; scratch for n and up to nt test points per call). */
int mobocmf_mf_kernel_combine(int64_t n1, int64_t n2, const double* Ks, const double* Kn, int64_t ld, const double* s1,
                              const double* s2, const int32_t* l1, const int32_t* l2, const double* ntab, double diag,
                              double* out, int64_t ldo, int64_t rows_p, int64_t cols_p, mobocmf_stream_t stream);
int mobocmf_exact_gp_workspace_bytes(int32_t n, int64_t nt, size_t* state_bytes, size_t* scratch_bytes);
int mobocmf_exact_gp_factor(int32_t n, const double* K, int64_t ldk, const double* y, double* mll, int32_t* info, void* state,
                            size_t state_bytes, void* scratch, size_t scratch_bytes, const mobocmf_tuning* tuning,
                            mobocmf_stream_t stream);
int mobocmf_exact_gp_predict(int32_t n, int64_t nt, const double* Kts, int64_t ld, const double* kss, double* mean,
                             double* var, const void* state, size_t state_bytes, void* scratch, size_t scratch_bytes,
                             const mobocmf_tuning* tuning, mobocmf_stream_t stream);

/* The f64 MFMA GEMM used by the layer (exposed for tests and for the roofline measurement of bench.py):
 * C[Mr x Nc] (+)= alpha * A[Mr x Kd] * B, B is [Kd x Nc] (trans_b = 0) or [Nc x Kd] (trans_b = 1).
 * Mr, Nc multiples of 128, Kd multiple of 16, leading dimensions even, pointers 16-byte aligned.
 * tri: bit 0 A lower-triangular, bit 1 A upper-triangular, bit 2 B lower, bit 3 B upper (square operands).
 * tuning: NULL = defaults (the small / mid-size operand thresholds decide which kernel a product runs on). */
int mobocmf_gemm_f64(int32_t tri, int32_t trans_b, int32_t Mr, int64_t Nc, int64_t Kd, const double* A, int64_t lda,
                     const double* B, int64_t ldb, double* C, int64_t ldc, double alpha, int32_t accumulate,
                     const mobocmf_tuning* tuning, mobocmf_stream_t stream);

/* The same kernel with the epilogues the layer launches it with (tests, and bench.py's per-variant roofline):
 *   epi 0  plain store (dK = L^-T dA);
 *   epi 1  store + partial column statistics, mobocmf_gemm_colstat_rows() partial rows of Nc entries each (two per row
 *          block of the tile height the launch uses): sum_p colsq_part[p][n] =
 *          sum_i C[i][n]^2 and (coldot_part non-NULL) sum_p coldot_part[p][n] = sum_i avec[i] C[i][n]   (A = L^-1 K -> q,
 *          mean; C = U^T A -> r);
 *   epi 2  C[i][n] = alpha bscale[n] (A B)[i][n] + avec[i] gmu[n] - 2 Aaux[i][n] cgv[n]   (dA), and (rowdot_part
 *          non-NULL) rowdot_part[slice][i] = partial sums over 64-column slices of Aaux[i][n] gmu[n]   (da).
 * B is [Kd x Nc] (no transposed form).  stream_out: non-temporal stores of C.  Pointers an epilogue does not use: NULL.
 * col_activity (NULL = dense): DEVICE array of one int32 per 128 columns of C; the column blocks marked 0 are left unwritten
 * (their row-dot partials zeroed) -- how the tests drive the block skipping of the layer backward directly.  The tile
 * height / pairing come from `tuning`. */
int mobocmf_gemm_f64_epilogue(int32_t tri, int32_t epi, int32_t Mr, int64_t Nc, int64_t Kd, const double* A, int64_t lda,
                              const double* B, int64_t ldb, double* C, int64_t ldc, double alpha, int32_t stream_out,
                              double* colsq_part, double* coldot_part, const double* avec, const double* bscale,
                              const double* gmu, const double* cgv, const double* Aaux, double* rowdot_part,
                              const int32_t* col_activity, const mobocmf_tuning* tuning, mobocmf_stream_t stream);

/* Partial rows an epi-1 launch of that shape writes to colsq_part / coldot_part under the same `tuning` (tile height). */
int mobocmf_gemm_colstat_rows(int32_t tri, int32_t Mr, int64_t Nc, int64_t Kd, const mobocmf_tuning* tuning, int32_t* rows);

/* The weighted symmetric rank-k update of the layer backward, H[Mr x Mr] = A diag(w) A^T with A [Mr x Kd] (k contiguous,
 * lda even) and w [Kd]: k-sliced over one round of resident workgroups into slabs (workspace), the slabs added and the
 * result written as the FULL symmetric matrix.  Both M x M contractions over N' of the reference's autograd backward
 * (A diag(gv) C^T for dU and dA K^T for dL^-1) reduce to it.  Mr, Kd multiples of 128.
 * k_activity (NULL = dense): DEVICE array of one int32 per 128 entries of the contraction; the blocks marked 0 (w is zero
 * throughout them) are left out.  The size query and the launch take the same `tuning` (syrk_workgroups sets the slab count). */
int mobocmf_syrk_workspace_bytes(int32_t Mr, int64_t Kd, const mobocmf_tuning* tuning, size_t* bytes);
int mobocmf_syrk_weighted_f64(int32_t Mr, int64_t Kd, const double* A, int64_t lda, const double* w, double* H,
                              void* workspace, int64_t workspace_bytes, const int32_t* k_activity,
                              const mobocmf_tuning* tuning, mobocmf_stream_t stream);

/* Host-side, synchronising: copies the device word and returns MOBOCMF_OK or MOBOCMF_NOT_PD (pivot in *pivot: the 1-based
 * column of the first non-positive pivot; -1 = a one-launch form -- the cooperative step, the one-launch Cholesky
 * (mobocmf_tuning.potrf_cols = 0) -- abandoned a bounded in-launch wait because its workgroups were not resident together:
 * the results of that call are invalid, repeat it with less concurrent work or with the launch-per-step form). */
int mobocmf_check_info(const int32_t* info, int32_t* pivot, mobocmf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MOBOCMF_HIP_H */
