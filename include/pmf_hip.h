/*
 * pmf_hip.h -- C ABI of libpmf_hip.so, the MI355X (gfx950) implementation of PathMatFac's
 * `fit!` gradient-descent loop.
 *
 * Drop-in boundary.  In the reference the ONLY caller of the inner loop is `mf_fit!`
 * (src/fit.jl:9-38), which forwards to `MF.fit!(model.matfac, model.data; ...)` (src/fit.jl:24);
 * `gpu(model)` / `cpu(model)` move the parameters (analyses/scripts/julia/fit_matfac.jl:325-340).
 * A host (the Julia shim in pathmatfac.jl_amd/julia/PathMatFacHIP.jl via `ccall`, or the Python
 * ctypes binding in pathmatfac.jl_amd/_lib.py) re-implements `mf_fit!` as
 *     marshal (pmf_set_*)  ->  pmf_fit  ->  unmarshal (pmf_get_*)
 * and everything above `mf_fit!` in src/fit.jl runs unchanged.  Each entry point below cites the
 * reference interface it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *  - every function returns 0 on success, <0 on error; pmf_last_error() gives the (thread-local) message.
 *    (The reference throws Julia exceptions / @assert: src/model.jl:125-184, src/util.jl:189.)
 *  - matrices are COLUMN-MAJOR as in Julia.  D is M x N float32 with NaN = missing; X is K x M; Y is K x N.
 *  - column / row ranges are 1-BASED INCLUSIVE (Julia UnitRange, src/util.jl:187-197) so the shim passes
 *    ranges verbatim; `batch_of_row` is 0-based int32 (rowval-1 of the one-hot CSC row_batches matrix,
 *    src/util.jl:200-210, 588-593).
 *  - host pointers are borrowed only for the duration of a call (copy-in / copy-out); the library owns all
 *    device memory.  One pmf_ctx = one device = one host thread at a time.  Calls are synchronous.
 *  - no torch / HIP types in the signatures: `void*` stream and device pointers are raw hipStream_t / addresses.
 */
#ifndef PMF_HIP_H
#define PMF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pmf_ctx pmf_ctx;

/* noise-model kinds (src/util.jl:128 VALID_LOSSES; MatFac NormalNoise / BernoulliNoise / PoissonNoise) */
#define PMF_NOISE_NORMAL 0
#define PMF_NOISE_BERNOULLI 1
#define PMF_NOISE_POISSON 2

/* optimizers: AdaGrad = Flux.Optimise.AdaGrad as constructed at src/fit.jl:41-43 and applied through
 * src/optimizers.jl:6-13 ; Adam = Flux.Optimise.Adam (named by the north-star; no reference call site) */
#define PMF_OPT_ADAGRAD 0
#define PMF_OPT_ADAM 1

/* storage type of the data matrix on the device */
#define PMF_STORE_F32 0
#define PMF_STORE_BF16 1

/* termination codes of pmf_fit == MatFac.fit!'s history["term_code"] (src/fit.jl:63) */
#define PMF_TERM_MAX_EPOCHS 0
#define PMF_TERM_LOSS_INCREASE 1
#define PMF_TERM_ABS_TOL 2
#define PMF_TERM_REL_TOL 3
#define PMF_TERM_NONFINITE 4

/* parameter groups (for pmf_get_grad / pmf_grad_device_ptr) */
#define PMF_PARAM_X 0
#define PMF_PARAM_Y 1
#define PMF_PARAM_LOGSIGMA 2
#define PMF_PARAM_MU 3
#define PMF_PARAM_LOGDELTA 4
#define PMF_PARAM_THETA 5

/* kwargs of mf_fit! / MF.fit! (src/fit.jl:9-36) that reach the kernel library */
typedef struct pmf_fit_opts {
  int32_t update_X;            /* src/fit.jl:10 */
  int32_t update_Y;            /* src/fit.jl:11 */
  int32_t update_col_layers;   /* src/fit.jl:13 */
  int32_t frozen_layers;       /* bit (l-1): layer l is a FrozenLayer (src/layers.jl:299-363, freeze_layer! :337) */
  int32_t frozen_regs;         /* bit (l-1): regs[l] is a FrozenRegularizer (src/regularizers.jl:950-999) */
  int32_t max_epochs;          /* src/fit.jl:58 */
  int32_t epoch;               /* 1-based starting epoch (src/fit.jl:56-58, :69) */
  int32_t tol_max_iters;       /* MatFac default 3 */
  int32_t keep_trace;          /* keep_history (src/fit.jl:20) */
  int32_t verbosity;           /* src/fit.jl:59 */
  int32_t print_iter;          /* test/runtests.jl:1335 */
  int32_t reserved;
  double abs_tol;              /* src/fit.jl:935 */
  double rel_tol;              /* src/fit.jl:934 */
  int64_t capacity;            /* src/fit.jl:938 -- accepted for API compatibility; D is streamed in one fused pass */
} pmf_fit_opts;

/* history Dict returned by MF.fit! (src/fit.jl:61-69): "term_code", "epochs", loss trace */
typedef struct pmf_fit_result {
  int32_t term_code;
  int32_t epochs;      /* last epoch index executed (src/fit.jl:69 resumes from h["epochs"]) */
  int32_t n_trace;     /* number of losses written to loss_trace */
  int32_t trace_cap;   /* in: capacity of loss_trace */
  double final_loss;
  double *loss_trace;  /* in: host buffer (may be NULL) */
  double seconds;      /* wall time of the epoch loop */
} pmf_fit_result;

const char *pmf_last_error(void);
int pmf_version(void);

/* gpu(model) / cpu(model): analyses/scripts/julia/fit_matfac.jl:325-340, src/transform.jl:78-94.
 * pmf_device_count: HIP devices visible to this process (CUDA.device! / the per-process device choice of
 * analyses/scripts/julia/script_util.jl:278-306): a launcher that pins one device per rank leaves ONE, device 0. */
int pmf_device_count(int *n);
int pmf_create(int device, pmf_ctx **out);
int pmf_destroy(pmf_ctx *ctx);
/* adopt an existing hipStream_t (e.g. the host framework's current stream); NULL = library-owned (non-blocking) stream.
 * The legacy DEFAULT stream is not NULL here: pass hipStreamLegacy ((hipStream_t)1).  A host whose collectives are
 * ordered behind the default stream (torch reports it as handle 0) must do so, or the library's kernels on its own
 * stream are not ordered with them. */
int pmf_set_stream(pmf_ctx *ctx, void *hip_stream);
int pmf_synchronize(pmf_ctx *ctx);

/* model.data (src/model.jl:12).  Host copy-in, or adopt a device-resident matrix without copying.
 * `M` is the number of LOCAL rows when the samples are sharded across GPUs. */
int pmf_set_data(pmf_ctx *ctx, const float *D_colmajor, int64_t M, int64_t N, int store);
int pmf_set_data_device(pmf_ctx *ctx, const void *D_device, int64_t M, int64_t N, int store);

/* model.matfac.X / .Y (K x M, K x N; src/model.jl:69-72, fields used at src/fit.jl:133-136) */
int pmf_set_factors(pmf_ctx *ctx, const float *X, const float *Y, int K);
int pmf_set_X(pmf_ctx *ctx, const float *X, int K);
int pmf_set_Y(pmf_ctx *ctx, const float *Y, int K);
int pmf_get_factors(pmf_ctx *ctx, float *X, float *Y);

/* col_transform.layers[1].logsigma (ColScale, src/layers.jl:9-22), layers[3].mu (ColShift, :53-66) */
int pmf_set_col_params(pmf_ctx *ctx, const float *logsigma, const float *mu);
int pmf_get_col_params(pmf_ctx *ctx, float *logsigma, float *mu);

/* col_transform.layers[2].logdelta / layers[4].theta : BatchArray (src/batch_array.jl:5-15).
 * n views = length(ba.col_ranges); view v: col_range, nb = size(values[v],1), one-hot row_batches[v]
 * given as batch_of_row, values nb x Nv column-major.  n = 0 makes layers 2 and 4 the identity
 * (src/layers.jl:243, src/transform.jl:64-65). */
int pmf_set_n_batch_views(pmf_ctx *ctx, int n);
int pmf_set_batch_view(pmf_ctx *ctx, int v, int64_t col_start1, int64_t col_stop1, int nb,
                       const int32_t *batch_of_row, const float *logdelta, const float *theta);
int pmf_get_batch_view(pmf_ctx *ctx, int v, float *logdelta, float *theta);

/* matfac.noise_model: CompositeNoise{noises, col_ranges} (src/fit.jl:227) + per-column weights set through
 * MF.set_weight! (src/fit.jl:157, 180) */
int pmf_set_noise(pmf_ctx *ctx, int n_ranges, const int64_t *starts1, const int64_t *stops1, const int32_t *kinds,
                  const float *weights);

/* matfac.X_reg.  A regularizer is a sum of terms p_t * reg_t (CompositeRegularizer, src/regularizers.jl:616-643).
 *   pmf_clear_xreg            : `X -> 0`                    (src/fit.jl:769, src/transform.jl:70)
 *   pmf_add_xreg_l2           : L2Regularizer(weights[K])   (src/regularizers.jl:11-33)
 *   pmf_add_xreg_group        : GroupRegularizer            (src/regularizers.jl:345-359, 423-446);
 *                               w is n_groups x K with K contiguous (group_weights[g][k]) */
int pmf_clear_xreg(pmf_ctx *ctx);
int pmf_add_xreg_l2(pmf_ctx *ctx, const float *w, float p);
int pmf_add_xreg_group(pmf_ctx *ctx, int n_groups, const int64_t *starts1, const int64_t *stops1, const float *w,
                       float p);

/* matfac.Y_reg.
 *   pmf_add_yreg_group : GroupRegularizer on feature ranges (construct_minimal_regularizer,
 *                        src/regularizers.jl:750-774; construct_Y_reg :715-717)
 *   pmf_add_yreg_l2    : L2Regularizer
 *   pmf_add_yreg_ard   : ARDRegularizer(alpha, beta per view range)  (src/regularizers.jl:526-585)
 *   pmf_add_yreg_fsard : FeatureSetARDReg call + rrule (src/featureset_ard.jl:135-150); alpha[N], beta[K x N] */
int pmf_clear_yreg(pmf_ctx *ctx);
int pmf_add_yreg_l2(pmf_ctx *ctx, const float *w, float p);
int pmf_add_yreg_group(pmf_ctx *ctx, int n_groups, const int64_t *starts1, const int64_t *stops1, const float *w,
                       float p);
int pmf_add_yreg_ard(pmf_ctx *ctx, int n_ranges, const int64_t *starts1, const int64_t *stops1, const float *alpha,
                     const float *beta, float p);
int pmf_add_yreg_fsard(pmf_ctx *ctx, const float *alpha, const float *beta, float p);

/* matfac.col_transform_reg = SequenceReg (src/regularizers.jl:896-926):
 *   regs[1], regs[3] : ColParamReg(col_ranges, weights, centers) on logsigma / mu   (:462-487)
 *   regs[2], regs[4] : BatchArrayReg(centers, weights) on logdelta / theta          (:781-815)
 * Pass n_ranges = 0 / NULL arrays for `x -> 0`.  Batch arrays are flat over (view, batch). */
int pmf_set_layer_regs(pmf_ctx *ctx, int n_ranges, const int64_t *starts1, const int64_t *stops1,
                       const float *w_logsigma, const float *c_logsigma, const float *w_mu, const float *c_mu,
                       const float *w_logdelta, const float *c_logdelta, const float *w_theta,
                       const float *c_theta);

/* construct_optimizer (src/fit.jl:41-43) ; `opt.eta *= 0.5` (src/fit.jl:64) ; fresh state per stage (src/fit.jl:55) */
int pmf_set_optimizer(pmf_ctx *ctx, int kind, float lr, float eps, float beta1, float beta2);
int pmf_set_lr(pmf_ctx *ctx, float lr);
int pmf_get_lr(pmf_ctx *ctx, float *lr);
int pmf_reset_optimizer_state(pmf_ctx *ctx);
/* the optimizer's per-parameter state in the parameter's own shape (view: batch view for logdelta / theta): AdaGrad's
 * `acc` (Flux.Optimise.AdaGrad, applied through src/optimizers.jl:6-13; starts at eps) or Adam's second moment, and
 * Adam's first moment.  The state lives as long as the optimizer object does in the reference: across the mf_fit! calls
 * of one mf_fit_adapt_lr! (src/fit.jl:55-69), whatever is re-marshalled in between. */
int pmf_get_opt_state(pmf_ctx *ctx, int which, int view, float *acc, float *mom);

/* MF.fit!(model.matfac, model.data; ...) as called from mf_fit! (src/fit.jl:24-36) */
int pmf_fit(pmf_ctx *ctx, const pmf_fit_opts *opts, pmf_fit_result *result);

/* Step-level API (what pmf_fit is made of), exposed so that a multi-GPU host can place the cross-GPU
 * reduction of grad(Y) and of the loss between the two halves of an epoch (DESIGN.md "multi-GPU"):
 *   pmf_epoch_begin     : fused data pass -> local data loss partials + data gradients (async on the stream)
 *   pmf_epoch_step_local: regularizer gradient + optimizer step of the row-local parameters (X)
 *   pmf_epoch_step_shared: same for the replicated parameters (Y and column layers), after their gradients
 *                          have been summed across ranks
 *   pmf_epoch_loss      : finishes the deterministic loss reduction and returns the local loss
 *                         (data partial + regularizer terms); `shared_terms` receives the part that is
 *                         replicated on every rank (so the host can avoid counting it once per rank) */
int pmf_epoch_begin(pmf_ctx *ctx, const pmf_fit_opts *opts);
int pmf_epoch_step_local(pmf_ctx *ctx, const pmf_fit_opts *opts);
int pmf_epoch_step_shared(pmf_ctx *ctx, const pmf_fit_opts *opts);
int pmf_epoch_loss(pmf_ctx *ctx, double *local_loss, double *shared_terms);

/* Multi-GPU (the reference is single-GPU -- one process per GPU is its habit for independent fits,
 * analyses/scripts/julia/script_util.jl:278-306).  The samples (rows of D, columns of X, batch_of_row) are sharded over
 * the ranks, one process and one pmf_ctx per GPU; Y, the column layers, the noise model and their regularizers are
 * replicated.  With a communicator attached, pmf_fit sums over the ranks, once per epoch: the partial grad(Y) (in a few
 * column chunks, each all-reduced beside the data pass of the other chunks), the rank-local part of the loss (data term +
 * X regularizer, two doubles) and -- when the column layers train -- their gradients; every rank then takes the same
 * steps of the replicated parameters and the same termination decision.  The step-level API below does not use the
 * communicator (there the host places its own collectives).
 *   pmf_comm_get_unique_id : ncclGetUniqueId; call on ONE rank, hand the PMF_COMM_ID_BYTES bytes to all (any side channel)
 *   pmf_comm_init          : ncclCommInitRank[Config] over RCCL / xGMI; collective over the ranks.  librccl.so.1 is loaded
 *                            here (dlopen), never before.  With nranks > 1 the data pass leaves PMF_COMM_CTAS CUs (env,
 *                            DEFAULT 4; 0 = none) to the collective's kernels, and this communicator alone is configured
 *                            for as many workgroups (ncclConfig_t.maxCTAs): the library sets NO environment variable and
 *                            touches nothing process-wide; other communicators of the host keep RCCL's defaults.
 *   pmf_comm_init_host     : the same protocol over a host callback `fn(user, host_buf, count, dtype)` that must sum
 *                            host_buf (dtype 0 = float32, 1 = float64) in place over the ranks and return 0; the library
 *                            stages device <-> pinned host memory around it.  For hosts without a usable RCCL ring
 *                            (tests: two ranks sharing one GPU).
 *   pmf_comm_set_chunks    : column chunks per data pass; 0 = automatic (1 on one rank, up to 4 with more; chosen from N, K
 *                            and the MEAN rows per rank, all-reduced once per pmf_fit, so that ranks with shards of
 *                            different heights issue the same collectives).  A non-zero value must be the same on all ranks.
 *   A pmf_fit that FAILS on a rank of a multi-rank communicator drains its streams and marks the communicator unusable
 *   (the peers may be waiting in a collective it never issued): a rank failure is fatal for the group.
 *   pmf_comm_allreduce     : sum (op 0) / maximum (op 1) over the ranks of a HOST buffer (dtype 0 = float32, 1 = float64),
 *                            for what the host keeps between the GD stages: the statistics of pmf_stats that feed
 *                            init_logsigma! / reweight_col_losses! / theta_delta_em (src/fit.jl:125-187, 326-375) must be
 *                            summed over the row shards.  No-op without a communicator.
 *   pmf_comm_info          : rank, size, transport, chunks of the last pmf_fit, reserved CUs, collectives issued */
#define PMF_COMM_ID_BYTES 128
#define PMF_COMM_NONE 0
#define PMF_COMM_RCCL 1
#define PMF_COMM_HOST 2
typedef int (*pmf_host_allreduce_fn)(void *user, void *host_buf, int64_t count, int dtype);
int pmf_comm_get_unique_id(void *id_out);
int pmf_comm_init(pmf_ctx *ctx, int rank, int nranks, const void *unique_id);
int pmf_comm_init_host(pmf_ctx *ctx, int rank, int nranks, pmf_host_allreduce_fn fn, void *user);
int pmf_comm_destroy(pmf_ctx *ctx);
int pmf_comm_set_chunks(pmf_ctx *ctx, int n_chunks);
int pmf_comm_allreduce(pmf_ctx *ctx, void *host_buf, int64_t count, int dtype, int op);
int pmf_comm_info(pmf_ctx *ctx, int *rank, int *nranks, int *transport, int *n_chunks, int *reserved_cus,
                  int64_t *n_collectives);

/* Diagnostics: the batch-layer variant the last data pass took (0 none, 1 LDS table with panel-local slots, 2 per-entry
 * gathers), the last layer pass (1 MFMA layer pass, 2 VALU kernel) and the columns of the dense batch table.  Views with
 * more than 15 batches stay on variant 1 as long as no 256-row (128-row for K > 64) panel holds more than 15 distinct
 * batches of one view (src/batch_array.jl:78-147 places no bound on the batch count). */
int pmf_debug_last_path(pmf_ctx *ctx, int *bmode, int *layer_path, int *slots);
/* the kernel family the last fused data pass ran on: 0 exact f32 (pmf_fused_kernel), 1 pmf_fused_sb_kernel, 2 pmf_fused_sb2_kernel,
 * 4 pmf_fused_sb4_kernel, 8 pmf_fused_sb8_kernel (split modes; tests assert that the intended variant really ran) */
int pmf_debug_last_kernel(pmf_ctx *ctx, int *kernel);

/* raw device addresses of the gradient buffers (float32) and their element counts, for in-place collectives */
int pmf_grad_device_ptr(pmf_ctx *ctx, int which, void **ptr, int64_t *n_elements);
/* copy a gradient of the last pmf_epoch_begin to the host in the reference's shape (tests / diagnostics) */
int pmf_get_grad(pmf_ctx *ctx, int which, int view, float *out);

/* MF.forward(matfac) (src/simulate_params.jl:247): Z = layers(X'Y) for all local rows, M x N column-major */
int pmf_forward(pmf_ctx *ctx, float *Z_host);

/* Masked column statistics for the closed-form initialisers that sit between the GD stages (local rows only; a
 * multi-GPU host sums them across ranks).  All outputs are optional (NULL = not wanted).
 *   col_n[N]        MF.column_nonnan                         (src/fit.jl:140, 447; src/regularizers.jl:765)
 *   col_sum, col_sumsq[N]  -> MF.batched_column_nanvar       (src/regularizers.jl:766)
 *   col_sqerr[N]    MF.link_col_sqerr                        (src/fit.jl:138, 444)
 *   col_ssq_grad[N] MF.batched_column_ssq_grads              (src/fit.jl:166)
 *   batch_count / batch_sqerr: ba_map(isfinite), ba_map(MF.sqerr_func) (src/fit.jl:332, 355, 454-456;
 *                   src/batch_array.jl:305-334), flat over (view, column, batch) like logdelta / theta
 * use_factors = 0 evaluates the model with X'Y = 0, which is how the reference calls these (src/fit.jl:133-136, 160-163). */
int pmf_stats(pmf_ctx *ctx, int use_factors, float *col_n, float *col_sum, float *col_sumsq, float *col_sqerr,
              float *col_ssq_grad, float *batch_count, float *batch_sqerr);

/* FeatureSetARD outer loop, one view: update_A! / update_A_inner! (src/featureset_ard.jl:214-294) with ISTAOptimiser.update!
 * (src/optimizers.jl:26-62) and gamma_normal_loss + its pull-back (:154-186), run on the device against the context's
 * resident Y (columns col_start1:col_stop1 = the view's col_range).  A starts from 0 (:286); S is the view's L x N_v
 * feature-set matrix ROW-major (S[l*N_v + j]; the reference holds it as a sparse CSC, src/util.jl:453-477); alpha = the
 * regularizer's alpha[cr]; lambda = the optimiser's per-factor L1 weights (update_lambda!, :189-209); ssq_grad is the
 * optimiser's accumulator (in / out, persists across calls like A_opts); A and ssq_grad are L x K with the factor index
 * contiguous (A[l*K + k]).  On return A = A_best, *best_loss its loss, *epochs_run the updates performed, and
 * beta[:, cr] = (alpha0 - 1) (v0 + A'S) (:292) is written to beta_out (K x N_v column-major, may be NULL) and into the
 * device copy of the Y regularizer's beta when one is attached (pmf_add_yreg_fsard). */
int pmf_fsard_update_A(pmf_ctx *ctx, int64_t col_start1, int64_t col_stop1, int L, const float *S, const float *alpha,
                       const float *lambda, float alpha0, float v0, float lr, float *ssq_grad, float *A, int max_epochs,
                       int term_iter, double atol, double *best_loss, int *epochs_run, float *beta_out);

/* Arithmetic of the three matrix products of the fused data pass (no reference counterpart: the reference computes
 * in Float32 on the GPU, src/fit.jl:24 via MatFac):
 *   PMF_PREC_F32    (default) exact f32 MFMA, v_mfma_f32_32x32x2_f32
 *   PMF_PREC_BF16X3 split-bf16: every f32 operand as a bf16 hi/lo pair, three bf16 MFMAs per product, f32 accumulation;
 *                   a six-term forward product as accurate as the f32 MFMA (2e-7 of max|Z|), three-term gradient products
 *                   (4e-6).  Used where a kernel variant exists (every K <= 128; batch layers unless a 256-row panel meets more
 *                   than 15 batches of one view, pmf_debug_last_path); every other launch silently stays exact.
 * pmf_get_precision also reports how many fused launches of this context took the split-bf16 kernel. */
#define PMF_PREC_F32 0
#define PMF_PREC_BF16X3 1
int pmf_set_precision(pmf_ctx *ctx, int mode);
int pmf_get_precision(pmf_ctx *ctx, int *mode, int64_t *split_launches);

/* per-launch timing of the fused data-pass kernel (HIP events on the library's stream): mean milliseconds and
 * number of launches since the last reset */
int pmf_kernel_time(pmf_ctx *ctx, double *mean_ms, int64_t *launches, int reset);
/* fill the data matrix on the device with synthetic values (benchmarks; no host transfer):
 * D = X'Y*exp(logsigma)+mu + noise*N(0,1) by a counter-based RNG; frac_nan entries set to NaN */
int pmf_synth_data(pmf_ctx *ctx, uint64_t seed, float noise, float frac_nan);

#ifdef __cplusplus
}
#endif
#endif /* PMF_HIP_H */
