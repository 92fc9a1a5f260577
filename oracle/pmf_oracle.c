/*
 * pmf_oracle.c -- CPU ORACLE for the PathMatFac `fit!` gradient-descent path.
 *
 * THIS FILE IS TEST INFRASTRUCTURE.  It is a plain-C restatement of the reference
 * algorithm, used only by tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg as the checker / reported CPU baseline.  Nothing in the
 * product package (pathmatfac.jl_amd/) links, imports or calls it.
 *
 * Pinning status
 * --------------
 *  * PINNED by the reference's own known-answer tests (tests/golden/ JSON files,
 *    transcribed from /root/reference/test/runtests.jl): BatchArray +,*,exp and
 *    their pullbacks (runtests.jl:203-240), ba_map (:244-257), layer identities
 *    (:373-413), GroupRegularizer (:739-764), ARDRegularizer (:766-777),
 *    BatchArrayReg (:780-792), FeatureSetARDReg loss/grad (:864-875).
 *  * PARITY UNPINNED: the inner epoch loop, noise-model formulas and termination
 *    logic live in the un-vendored dependency MatFac.jl (Manifest.toml:558-564,
 *    git-tree-sha1 37d124152593f8e04a24d210d3054a98a2a7bbc9) which is absent from
 *    /root/reference and cannot be fetched.  Those parts (o_noise_loss_grad,
 *    o_fit's loop/termination) are SELF-SPECIFIED here and documented in DESIGN.md.
 *
 * Every function cites the reference file:line (relative to /root/reference) it follows.
 *
 * Precision: `real` is double by default (the parity anchor); -DORACLE_F32 builds
 * the same code in float (used as the timed CPU "port" baseline).
 *
 * Layout conventions (Julia's): all matrices column-major.  D is M x N (float, NaN =
 * missing); X is K x M; Y is K x N.  Ranges are 1-based inclusive, as Julia UnitRanges.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef ORACLE_F32
typedef float real;
#define R_EXP expf
#define R_LOG logf
#define R_SQRT sqrtf
#define R_LOG1P log1pf
#define R_FABS fabsf
#else
typedef double real;
#define R_EXP exp
#define R_LOG log
#define R_SQRT sqrt
#define R_LOG1P log1p
#define R_FABS fabs
#endif

#define O_KIND_NORMAL 0
#define O_KIND_BERNOULLI 1
#define O_KIND_POISSON 2

#define O_REG_NONE 0
#define O_REG_L2 1
#define O_REG_GROUP 2
#define O_REG_ARD 3
#define O_REG_FSARD 4

#define O_OPT_ADAGRAD 0
#define O_OPT_ADAM 1

#define O_TERM_MAX_EPOCHS 0
#define O_TERM_LOSS_INCREASE 1
#define O_TERM_ABS_TOL 2
#define O_TERM_REL_TOL 3
#define O_TERM_NONFINITE 4

typedef struct {
  int32_t kind;     /* O_REG_* */
  int32_t n_ranges; /* GROUP / ARD: number of column ranges */
  real p;           /* CompositeRegularizer mixture weight (regularizers.jl:641-643); 1 for a bare regularizer */
  const int64_t *start1, *stop1; /* 1-based inclusive ranges (GROUP, ARD) */
  const real *w;    /* L2: K weights ; GROUP: n_ranges x K (range-major, K contiguous) */
  const real *a;    /* ARD: alpha per range ; FSARD: alpha[N] */
  const real *b;    /* ARD: beta per range  ; FSARD: beta[K x N] column-major */
} o_regterm;

typedef struct {
  int64_t M, N;
  int32_t K;
  int32_t n_bv;    /* number of batch views (BatchArray col_ranges) */
  const float *D;  /* M x N */
  real *X, *Y;     /* K x M, K x N */
  real *logsigma, *mu; /* N */
  const int64_t *bv_start1, *bv_stop1; /* per view column range */
  const int32_t *bv_nb;                /* per view number of row batches */
  const int32_t *bv_bor;               /* n_bv x M (view-major): 0-based row-batch index of each row */
  const int64_t *bv_off;               /* n_bv+1 offsets into logdelta/theta (each view nb x Nv col-major) */
  real *logdelta, *theta;              /* flat BatchArray values */
  int32_t has_batch;                   /* layers 2,4 are BatchScale/BatchShift (else identity) */
  int32_t n_noise;
  const int64_t *nz_start1, *nz_stop1;
  const int32_t *nz_kind;
  const real *col_weight; /* N */
  int32_t n_xreg, n_yreg;
  const o_regterm *xreg, *yreg;
  /* SequenceReg (regularizers.jl:896-926): regs[1],regs[3] ColParamReg ; regs[2],regs[4] BatchArrayReg */
  int32_t has_colreg, n_cr;
  const int64_t *cr_start1, *cr_stop1;
  const real *cr_w_logsigma, *cr_c_logsigma, *cr_w_mu, *cr_c_mu; /* per range */
  int32_t has_batchreg;
  const int64_t *bvb_off; /* n_bv+1 cumulative nb */
  const real *br_w_logdelta, *br_c_logdelta, *br_w_theta, *br_c_theta; /* per (view,batch) */
} o_model;

typedef struct {
  int32_t update_X, update_Y, update_col_layers;
  int32_t frozen_layers; /* bit (l-1) set: layer l wrapped in FrozenLayer (layers.jl:299-363) */
  int32_t frozen_regs;   /* bit (l-1) set: regs[l] wrapped in FrozenRegularizer (regularizers.jl:950-999) */
  int32_t opt_kind;
  int32_t max_epochs, epoch; /* epoch: 1-based starting epoch (fit.jl:56-58) */
  int32_t tol_max_iters;
  int32_t chunk_rows;    /* row-batch size (MatFac `capacity / N`, cf. batch_array.jl:322-327) ; <=0: all rows */
  real lr, eps, beta1, beta2;
  real abs_tol, rel_tol;
} o_opts;

/* optimizer state: index 0 X, 1 Y, 2 logsigma, 3 mu, 4 logdelta, 5 theta */
typedef struct {
  real *acc[6]; /* AdaGrad accumulator / Adam second moment */
  real *mom[6]; /* Adam first moment */
  real bp1[6], bp2[6]; /* Adam running beta powers (Flux Adam keeps βp per parameter) */
  int32_t initialized;
} o_optstate;

/* ------------------------------------------------------------------------- */
/* BatchArray primitives (src/batch_array.jl).  A "view" of rows [i0,i0+m) is  */
/* expressed by passing bor + i0 (the one-hot row_batches matrix restricted to */
/* those rows: batch_array.jl:83-100, util.jl:513-529,576-578).               */
/* ------------------------------------------------------------------------- */

/* batch_array.jl:121-129  A + B : view(result,:,cr) .+= row_batches[j]*values[j] */
void o_ba_add(real *A, int64_t m, int64_t ldA, int32_t n_bv, const int64_t *start1, const int64_t *stop1,
              const int32_t *nb, const int32_t *bor, int64_t bor_ld, const int64_t *off, const real *values) {
  for (int32_t v = 0; v < n_bv; ++v) {
    const int64_t c0 = start1[v] - 1, c1 = stop1[v];
    const real *val = values + off[v];
    const int32_t *b = bor + (int64_t)v * bor_ld;
#pragma omp parallel for schedule(static)
    for (int64_t j = c0; j < c1; ++j)
      for (int64_t i = 0; i < m; ++i)
        if (b[i] >= 0) A[i + ldA * j] += val[b[i] + (int64_t)nb[v] * (j - c0)];
  }
}

/* batch_array.jl:136-147 pullback of +: values_bar[j] = transpose(row_batches[j]) * view(result_bar,:,cbr)
 * (accumulates into values_bar so that row chunks sum up). A_bar = copy(result_bar) is the identity. */
void o_ba_add_pullback(const real *Zbar, int64_t m, int64_t ldZ, int32_t n_bv, const int64_t *start1,
                       const int64_t *stop1, const int32_t *nb, const int32_t *bor, int64_t bor_ld,
                       const int64_t *off, real *values_bar) {
  for (int32_t v = 0; v < n_bv; ++v) {
    const int64_t c0 = start1[v] - 1, c1 = stop1[v];
    real *vb = values_bar + off[v];
    const int32_t *b = bor + (int64_t)v * bor_ld;
#pragma omp parallel for schedule(static)
    for (int64_t j = c0; j < c1; ++j)
      for (int64_t i = 0; i < m; ++i)
        if (b[i] >= 0) vb[b[i] + (int64_t)nb[v] * (j - c0)] += Zbar[i + ldZ * j];
  }
}

/* batch_array.jl:174-181  A * B : view(result,:,cbr) .*= row_batches[j]*values[j] */
void o_ba_mul(real *A, int64_t m, int64_t ldA, int32_t n_bv, const int64_t *start1, const int64_t *stop1,
              const int32_t *nb, const int32_t *bor, int64_t bor_ld, const int64_t *off, const real *values) {
  for (int32_t v = 0; v < n_bv; ++v) {
    const int64_t c0 = start1[v] - 1, c1 = stop1[v];
    const real *val = values + off[v];
    const int32_t *b = bor + (int64_t)v * bor_ld;
#pragma omp parallel for schedule(static)
    for (int64_t j = c0; j < c1; ++j)
      for (int64_t i = 0; i < m; ++i)
        if (b[i] >= 0) A[i + ldA * j] *= val[b[i] + (int64_t)nb[v] * (j - c0)];
  }
}

/* batch_array.jl:192-209 pullback of A*B.  A is the *input* of the product.
 *   A_bar[:,cbr] = result_bar[:,cbr] .* buffers[j]                (:194-199)
 *   values_bar[j] = transpose(row_batches[j]) * (A[:,cbr] .* result_bar[:,cbr])   (:203-204)
 * Columns outside every col_range pass through unchanged (:194-195).
 * Abar may alias Zbar (in-place).  (Q2: the reference overwrites its captured A; we do not.) */
void o_ba_mul_pullback(const real *A, const real *Zbar, real *Abar, int64_t m, int64_t ld, int32_t n_bv,
                       const int64_t *start1, const int64_t *stop1, const int32_t *nb, const int32_t *bor,
                       int64_t bor_ld, const int64_t *off, const real *values, real *values_bar) {
  for (int32_t v = 0; v < n_bv; ++v) {
    const int64_t c0 = start1[v] - 1, c1 = stop1[v];
    const real *val = values + off[v];
    real *vb = values_bar ? values_bar + off[v] : NULL;
    const int32_t *b = bor + (int64_t)v * bor_ld;
#pragma omp parallel for schedule(static)
    for (int64_t j = c0; j < c1; ++j)
      for (int64_t i = 0; i < m; ++i) {
        const real zb = Zbar[i + ld * j];
        if (b[i] >= 0) {
          const int64_t e = b[i] + (int64_t)nb[v] * (j - c0);
          if (vb) vb[e] += A[i + ld * j] * zb;
          Abar[i + ld * j] = zb * val[e];
        } else {
          /* Extension: a row with batch index -1 is left untouched by every BatchArray op (identity).
           * The reference constructor never produces such rows (ids_to_ind_mat is one-hot per row,
           * util.jl:200-210), so this branch is unreachable from reference-shaped inputs. */
          Abar[i + ld * j] = zb;
        }
      }
  }
}

/* batch_array.jl:230-237 exp(ba) ; :246-256 pullback values_bar = Z_bar.values .* Z.values */
void o_ba_exp(const real *values, int64_t n, real *out) {
  for (int64_t e = 0; e < n; ++e) out[e] = R_EXP(values[e]);
}
void o_ba_exp_pullback(const real *zbar_values, const real *z_values, int64_t n, real *values_bar) {
  for (int64_t e = 0; e < n; ++e) values_bar[e] = zbar_values[e] * z_values[e];
}

/* batch_array.jl:305-317 ba_batch_colsums! / :320-334 ba_map : per (batch, column) sums of Q over rows */
void o_ba_colsums(const real *Q, int64_t m, int64_t ldQ, int32_t n_bv, const int64_t *start1,
                  const int64_t *stop1, const int32_t *nb, const int32_t *bor, int64_t bor_ld,
                  const int64_t *off, real *values_acc) {
  o_ba_add_pullback(Q, m, ldQ, n_bv, start1, stop1, nb, bor, bor_ld, off, values_acc);
}

/* ------------------------------------------------------------------------- */
/* Column layers (src/layers.jl)                                              */
/* ------------------------------------------------------------------------- */

/* layers.jl:20-22 ColScale: Z .* transpose(exp.(logsigma)) */
void o_colscale(real *Z, int64_t m, int64_t ld, int64_t N, const real *logsigma) {
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < N; ++j) {
    const real s = R_EXP(logsigma[j]);
    for (int64_t i = 0; i < m; ++i) Z[i + ld * j] *= s;
  }
}
/* layers.jl:34-48 ColScale rrule: Z_bar = sigma' .* result_bar ; logsigma_bar = vec(sum(Z_bar; dims=1))
 * (Q1: as coded, logsigma_bar omits the input factor). In-place on Zbar. */
void o_colscale_pullback(real *Zbar, int64_t m, int64_t ld, int64_t N, const real *logsigma,
                         real *logsigma_bar) {
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < N; ++j) {
    const real s = R_EXP(logsigma[j]);
    real acc = 0;
    for (int64_t i = 0; i < m; ++i) {
      Zbar[i + ld * j] *= s;
      acc += Zbar[i + ld * j];
    }
    if (logsigma_bar) logsigma_bar[j] += acc;
  }
}
/* layers.jl:64-66 ColShift: Z .+ transpose(mu) */
void o_colshift(real *Z, int64_t m, int64_t ld, int64_t N, const real *mu) {
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < N; ++j)
    for (int64_t i = 0; i < m; ++i) Z[i + ld * j] += mu[j];
}
/* layers.jl:78-90 ColShift rrule: mu_bar = vec(sum(result_bar; dims=1)) ; Z_bar = copy(result_bar) */
void o_colshift_pullback(const real *Zbar, int64_t m, int64_t ld, int64_t N, real *mu_bar) {
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < N; ++j) {
    real acc = 0;
    for (int64_t i = 0; i < m; ++i) acc += Zbar[i + ld * j];
    mu_bar[j] += acc;
  }
}

/* ------------------------------------------------------------------------- */
/* Forward product, noise model, gradient products                             */
/* ------------------------------------------------------------------------- */

/* transpose(X)*Y (MatFac `forward`; same expression at layers.jl:283, impute.jl:42, runtests.jl:363) */
static void o_xty(const real *X, const real *Y, int32_t K, int64_t i0, int64_t m, int64_t N, real *A) {
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < N; ++j) {
    const real *y = Y + (int64_t)K * j;
    for (int64_t i = 0; i < m; ++i) {
      const real *x = X + (int64_t)K * (i0 + i);
      real s = 0;
#pragma omp simd reduction(+ : s)
      for (int32_t k = 0; k < K; ++k) s += x[k] * y[k];
      A[i + m * j] = s;
    }
  }
}

static inline real o_softplus(real z) { return (z > 0 ? z : 0) + R_LOG1P(R_EXP(-R_FABS(z))); }
static inline real o_sigmoid(real z) {
  if (z >= 0) return (real)1 / ((real)1 + R_EXP(-z));
  const real e = R_EXP(z);
  return e / ((real)1 + e);
}

/* SELF-SPECIFIED (MatFac noise models): for each finite D_ij
 *   normal    : l = 0.5 w_j (z-y)^2          g = w_j (z - y)
 *   bernoulli : l = w_j (softplus(z) - y z)  g = w_j (sigmoid(z) - y)      (logit link)
 *   poisson   : l = w_j (exp(z) - y z)       g = w_j (exp(z) - y)          (log link)
 * non-finite D_ij contribute 0 loss and 0 gradient (transform.jl:55-57 NaN padding).
 * Z is overwritten with G = dloss/dZ. Returns the chunk's loss. */
static double o_noise_loss_grad(real *Z, const float *D, int64_t i0, int64_t m, int64_t M, int64_t N,
                                int32_t n_noise, const int64_t *s1, const int64_t *e1, const int32_t *kind,
                                const real *w) {
  double total = 0;
  for (int32_t r = 0; r < n_noise; ++r) {
    const int64_t c0 = s1[r] - 1, c1 = e1[r];
    const int32_t kd = kind[r];
#pragma omp parallel for schedule(static) reduction(+ : total)
    for (int64_t j = c0; j < c1; ++j) {
      double colsum = 0;
      const real wj = w[j];
      for (int64_t i = 0; i < m; ++i) {
        const float yf = D[(i0 + i) + M * j];
        real *zp = &Z[i + m * j];
        if (!isfinite(yf)) { *zp = 0; continue; }
        const real y = (real)yf, z = *zp;
        real l, g;
        if (kd == O_KIND_NORMAL) {
          const real d = z - y;
          l = (real)0.5 * wj * d * d;
          g = wj * d;
        } else if (kd == O_KIND_BERNOULLI) {
          l = wj * (o_softplus(z) - y * z);
          g = wj * (o_sigmoid(z) - y);
        } else {
          const real e = R_EXP(z);
          l = wj * (e - y * z);
          g = wj * (e - y);
        }
        colsum += (double)l;
        *zp = g;
      }
      total += colsum;
    }
  }
  (void)N;
  return total;
}

/* Zygote adjoint of transpose(X)*Y:  gX[:,i] = Y * Abar[i,:]' ; gY += X_chunk * Abar */
static void o_grad_products(const real *X, const real *Y, int32_t K, int64_t i0, int64_t m, int64_t N,
                            const real *Abar, real *gX, real *gY) {
  if (gX) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < m; ++i) {
      real *g = gX + (int64_t)K * (i0 + i);
      for (int64_t j = 0; j < N; ++j) {
        const real a = Abar[i + m * j];
        if (a == 0) continue;
        const real *y = Y + (int64_t)K * j;
#pragma omp simd
        for (int32_t k = 0; k < K; ++k) g[k] += a * y[k];
      }
    }
  }
  if (gY) {
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < N; ++j) {
      real *g = gY + (int64_t)K * j;
      for (int64_t i = 0; i < m; ++i) {
        const real a = Abar[i + m * j];
        if (a == 0) continue;
        const real *x = X + (int64_t)K * (i0 + i);
#pragma omp simd
        for (int32_t k = 0; k < K; ++k) g[k] += a * x[k];
      }
    }
  }
}

/* Forward through the 4 column layers in order 1..4 (layers.jl:227-229, 240-253) for rows [i0,i0+m).
 * Z (m x N, ld=m) holds A on entry, layer output on exit.  Z1 (optional) receives the output of
 * layer 1 (the input of BatchScale, needed by its pullback). */
void o_layers_forward(const o_model *mdl, real *Z, real *Z1, int64_t i0, int64_t m) {
  const int64_t N = mdl->N;
  o_colscale(Z, m, m, N, mdl->logsigma);                       /* layer 1: ColScale */
  if (Z1) memcpy(Z1, Z, sizeof(real) * (size_t)(m * N));
  if (mdl->has_batch && mdl->n_bv > 0) {                       /* layer 2: BatchScale  layers.jl:120-122 */
    const int64_t tot = mdl->bv_off[mdl->n_bv];
    real *delta = (real *)malloc(sizeof(real) * (size_t)(tot > 0 ? tot : 1));
    o_ba_exp(mdl->logdelta, tot, delta);
    o_ba_mul(Z, m, m, mdl->n_bv, mdl->bv_start1, mdl->bv_stop1, mdl->bv_nb, mdl->bv_bor + i0, mdl->M,
             mdl->bv_off, delta);
    free(delta);
  }
  o_colshift(Z, m, m, N, mdl->mu);                             /* layer 3: ColShift */
  if (mdl->has_batch && mdl->n_bv > 0)                         /* layer 4: BatchShift  layers.jl:199-201 */
    o_ba_add(Z, m, m, mdl->n_bv, mdl->bv_start1, mdl->bv_stop1, mdl->bv_nb, mdl->bv_bor + i0, mdl->M,
             mdl->bv_off, mdl->theta);
}

/* Full forward Z = layers(X'Y) for all rows, written to Zout (M x N col-major). Test helper. */
void o_forward(const o_model *mdl, real *Zout) {
  o_xty(mdl->X, mdl->Y, mdl->K, 0, mdl->M, mdl->N, Zout);
  o_layers_forward(mdl, Zout, NULL, 0, mdl->M);
}

/* One pass over the data: loss and gradients of the data term (likelihood) only.
 * Gradient outputs may be NULL (not wanted); non-NULL ones are ACCUMULATED INTO (caller zeroes).
 * Layer-parameter gradients honour frozen_layers: a FrozenLayer yields no tangent (layers.jl:314-325). */
double o_data_pass(const o_model *mdl, const o_opts *opt, real *gX, real *gY, real *g_logsigma, real *g_mu,
                   real *g_logdelta, real *g_theta) {
  const int64_t M = mdl->M, N = mdl->N;
  int64_t chunk = (opt && opt->chunk_rows > 0) ? opt->chunk_rows : M;
  if (chunk > M) chunk = M;
  const int fl = opt ? opt->frozen_layers : 0;
  if (fl & 1) g_logsigma = NULL;
  if (fl & 2) g_logdelta = NULL;
  if (fl & 4) g_mu = NULL;
  if (fl & 8) g_theta = NULL;
  const int batch = mdl->has_batch && mdl->n_bv > 0;
  const int64_t tot = batch ? mdl->bv_off[mdl->n_bv] : 0;
  real *Z = (real *)malloc(sizeof(real) * (size_t)(chunk * N));
  real *Z1 = batch ? (real *)malloc(sizeof(real) * (size_t)(chunk * N)) : NULL;
  real *delta = batch ? (real *)malloc(sizeof(real) * (size_t)(tot > 0 ? tot : 1)) : NULL;
  real *delta_bar = (batch && g_logdelta) ? (real *)calloc((size_t)(tot > 0 ? tot : 1), sizeof(real)) : NULL;
  if (batch) o_ba_exp(mdl->logdelta, tot, delta);
  double loss = 0;
  for (int64_t i0 = 0; i0 < M; i0 += chunk) {
    const int64_t m = (M - i0 < chunk) ? (M - i0) : chunk;
    o_xty(mdl->X, mdl->Y, mdl->K, i0, m, N, Z);
    o_layers_forward(mdl, Z, Z1, i0, m);
    loss += o_noise_loss_grad(Z, mdl->D, i0, m, M, N, mdl->n_noise, mdl->nz_start1, mdl->nz_stop1,
                              mdl->nz_kind, mdl->col_weight);
    /* backward, layers 4 -> 1.  Z now holds result_bar. */
    if (batch && g_theta)
      o_ba_add_pullback(Z, m, m, mdl->n_bv, mdl->bv_start1, mdl->bv_stop1, mdl->bv_nb, mdl->bv_bor + i0, M,
                        mdl->bv_off, g_theta);
    if (g_mu) o_colshift_pullback(Z, m, m, N, g_mu);
    if (batch)
      o_ba_mul_pullback(Z1, Z, Z, m, m, mdl->n_bv, mdl->bv_start1, mdl->bv_stop1, mdl->bv_nb,
                        mdl->bv_bor + i0, M, mdl->bv_off, delta, delta_bar);
    o_colscale_pullback(Z, m, m, N, mdl->logsigma, g_logsigma);
    o_grad_products(mdl->X, mdl->Y, mdl->K, i0, m, N, Z, gX, gY);
  }
  if (delta_bar) { /* exp pullback, batch_array.jl:249-253 */
    for (int64_t e = 0; e < tot; ++e) g_logdelta[e] += delta_bar[e] * delta[e];
    free(delta_bar);
  }
  free(Z);
  free(Z1);
  free(delta);
  return loss;
}

/* ------------------------------------------------------------------------- */
/* Regularizers (src/regularizers.jl, src/featureset_ard.jl)                   */
/* ------------------------------------------------------------------------- */

/* One regularizer term applied to P (K x n col-major). Returns p*loss, accumulates p*grad into g (may be NULL). */
double o_regterm_apply(const o_regterm *t, const real *P, int32_t K, int64_t n, real *g) {
  double loss = 0;
  const real p = t->p;
  switch (t->kind) {
    case O_REG_NONE: /* x -> 0  (regularizers.jl:670, fit.jl:769) */
      return 0;
    case O_REG_L2: { /* regularizers.jl:21-33: 0.5*sum(weights .* sum(X.*X, dims=2)); grad weights .* X */
      for (int64_t i = 0; i < n; ++i)
        for (int32_t k = 0; k < K; ++k) {
          const real x = P[k + (int64_t)K * i], gx = t->w[k] * x;
          loss += 0.5 * (double)(gx * x);
          if (g) g[k + (int64_t)K * i] += p * gx;
        }
      break;
    }
    case O_REG_GROUP: { /* regularizers.jl:423-446: 0.5*sum_g sum(w_g .* X[:,idx_g].^2); grad w_g .* X[:,idx_g] */
      for (int32_t r = 0; r < t->n_ranges; ++r) {
        const real *w = t->w + (int64_t)K * r;
        for (int64_t i = t->start1[r] - 1; i < t->stop1[r]; ++i)
          for (int32_t k = 0; k < K; ++k) {
            const real x = P[k + (int64_t)K * i], gx = w[k] * x;
            loss += 0.5 * (double)(gx * x);
            if (g) g[k + (int64_t)K * i] += p * gx;
          }
      }
      break;
    }
    case O_REG_ARD: { /* regularizers.jl:546-585: (0.5+a)*sum(log(1+(0.5/b) X^2)); grad (1/b)(0.5+a) X / buffer */
      for (int32_t r = 0; r < t->n_ranges; ++r) {
        const real a = t->a[r], b = t->b[r];
        for (int64_t i = t->start1[r] - 1; i < t->stop1[r]; ++i)
          for (int32_t k = 0; k < K; ++k) {
            const real x = P[k + (int64_t)K * i];
            const real buf = (real)1 + ((real)0.5 / b) * (x * x);
            loss += (double)(((real)0.5 + a) * R_LOG(buf));
            if (g) g[k + (int64_t)K * i] += p * (((real)1 / b) * ((real)0.5 + a) * x / buf);
          }
      }
      break;
    }
    case O_REG_FSARD: { /* featureset_ard.jl:135-150: b = 1 + (0.5/beta) Y^2; sum((0.5+alpha)' .* sum(log b));
                           grad (alpha+0.5)' .* Y ./ (b .* beta)  (loss_bar = 1; Q5) */
      for (int64_t j = 0; j < n; ++j)
        for (int32_t k = 0; k < K; ++k) {
          const real y = P[k + (int64_t)K * j], be = t->b[k + (int64_t)K * j];
          const real b = (real)1 + ((real)0.5 / be) * (y * y);
          loss += (double)(((real)0.5 + t->a[j]) * R_LOG(b));
          if (g) g[k + (int64_t)K * j] += p * ((t->a[j] + (real)0.5) * y / (b * be));
        }
      break;
    }
    default:
      return NAN;
  }
  return (double)p * loss;
}

/* regularizers.jl:482-487 ColParamReg: 0.5*sum_r w_r*sum((v[r] .- c_r).^2) ; AD gradient w_r (v - c_r) */
double o_colparamreg(int32_t n_ranges, const int64_t *start1, const int64_t *stop1, const real *w,
                     const real *c, const real *v, real *g) {
  double loss = 0;
  for (int32_t r = 0; r < n_ranges; ++r)
    for (int64_t j = start1[r] - 1; j < stop1[r]; ++j) {
      const real d = v[j] - c[r];
      loss += 0.5 * (double)(w[r] * d * d);
      if (g) g[j] += w[r] * d;
    }
  return loss;
}

/* regularizers.jl:795-815 BatchArrayReg: diffs = v .- centers (per batch row) ; 0.5*sum(w .* d .* d); grad w .* d */
double o_batcharrayreg(int32_t n_bv, const int64_t *start1, const int64_t *stop1, const int32_t *nb,
                       const int64_t *off, const int64_t *bvb_off, const real *w, const real *c,
                       const real *values, real *g) {
  double loss = 0;
  for (int32_t v = 0; v < n_bv; ++v) {
    const int64_t Nv = stop1[v] - start1[v] + 1;
    for (int64_t j = 0; j < Nv; ++j)
      for (int32_t b = 0; b < nb[v]; ++b) {
        const int64_t e = off[v] + b + (int64_t)nb[v] * j;
        const real d = values[e] - c[bvb_off[v] + b], ww = w[bvb_off[v] + b];
        loss += 0.5 * (double)(ww * d * d);
        if (g) g[e] += ww * d;
      }
  }
  return loss;
}

/* ------------------------------------------------------------------------- */
/* Optimizers (src/optimizers.jl:6-13 + Flux 0.13.13 AdaGrad/Adam)             */
/* ------------------------------------------------------------------------- */

/* AdaGrad: acc starts at eps (fill!(similar, o.epsilon)); acc += g*g ; p -= eta*g/(sqrt(acc)+eps) */
void o_adagrad_step(real *p, const real *g, real *acc, int64_t n, real eta, real eps) {
  for (int64_t e = 0; e < n; ++e) {
    acc[e] += g[e] * g[e];
    p[e] -= g[e] * (eta / (R_SQRT(acc[e]) + eps));
  }
}
/* Adam (Flux.Optimise.Adam apply!): mt = b1 mt + (1-b1) g ; vt = b2 vt + (1-b2) g^2 ;
 * delta = mt/(1-b1^t) / (sqrt(vt/(1-b2^t)) + eps) * eta.  No reference counterpart (north-star). */
void o_adam_step(real *p, const real *g, real *m, real *v, int64_t n, real eta, real eps, real b1, real b2,
                 real *bp1, real *bp2) {
  const real c1 = (real)1 - *bp1, c2 = (real)1 - *bp2;
  for (int64_t e = 0; e < n; ++e) {
    m[e] = b1 * m[e] + ((real)1 - b1) * g[e];
    v[e] = b2 * v[e] + ((real)1 - b2) * g[e] * g[e];
    p[e] -= m[e] / c1 / (R_SQRT(v[e] / c2) + eps) * eta;
  }
  *bp1 *= b1;
  *bp2 *= b2;
}

static int64_t o_param_len(const o_model *m, int which) {
  switch (which) {
    case 0: return (int64_t)m->K * m->M;
    case 1: return (int64_t)m->K * m->N;
    case 2: case 3: return m->N;
    default: return (m->has_batch && m->n_bv > 0) ? m->bv_off[m->n_bv] : 0;
  }
}

void o_optstate_init(const o_model *m, const o_opts *opt, o_optstate *st) {
  for (int w = 0; w < 6; ++w) {
    const int64_t n = o_param_len(m, w);
    for (int64_t e = 0; e < n; ++e) {
      if (st->acc[w]) st->acc[w][e] = (opt->opt_kind == O_OPT_ADAGRAD) ? opt->eps : 0;
      if (st->mom[w]) st->mom[w][e] = 0;
    }
    st->bp1[w] = opt->beta1;
    st->bp2[w] = opt->beta2;
  }
  st->initialized = 1;
}

static void o_step(const o_model *m, const o_opts *opt, o_optstate *st, int which, real *p, const real *g) {
  const int64_t n = o_param_len(m, which);
  if (n == 0) return;
  if (opt->opt_kind == O_OPT_ADAGRAD)
    o_adagrad_step(p, g, st->acc[which], n, opt->lr, opt->eps);
  else
    o_adam_step(p, g, st->mom[which], st->acc[which], n, opt->lr, opt->eps, opt->beta1, opt->beta2,
                &st->bp1[which], &st->bp2[which]);
}

/* ------------------------------------------------------------------------- */
/* Full loss + gradient at the current parameters (data + regularizers)        */
/* ------------------------------------------------------------------------- */

/* Gradient buffers (each may be NULL) are overwritten. Regularizer terms are evaluated only for the
 * parameter groups being updated (see DESIGN.md "epoch semantics"; cf. fit_lbfgs.jl:48-52). */
double o_loss_and_grads(const o_model *mdl, const o_opts *opt, real *gX, real *gY, real *g_logsigma,
                        real *g_mu, real *g_logdelta, real *g_theta, double *data_loss_out) {
  const int64_t nX = o_param_len(mdl, 0), nY = o_param_len(mdl, 1), nB = o_param_len(mdl, 4);
  if (gX) memset(gX, 0, sizeof(real) * (size_t)nX);
  if (gY) memset(gY, 0, sizeof(real) * (size_t)nY);
  if (g_logsigma) memset(g_logsigma, 0, sizeof(real) * (size_t)mdl->N);
  if (g_mu) memset(g_mu, 0, sizeof(real) * (size_t)mdl->N);
  if (g_logdelta && nB) memset(g_logdelta, 0, sizeof(real) * (size_t)nB);
  if (g_theta && nB) memset(g_theta, 0, sizeof(real) * (size_t)nB);
  const int ul = opt->update_col_layers;
  double loss = o_data_pass(mdl, opt, opt->update_X ? gX : NULL, opt->update_Y ? gY : NULL,
                            ul ? g_logsigma : NULL, ul ? g_mu : NULL, ul ? g_logdelta : NULL,
                            ul ? g_theta : NULL);
  if (data_loss_out) *data_loss_out = loss;
  if (opt->update_X)
    for (int32_t t = 0; t < mdl->n_xreg; ++t) loss += o_regterm_apply(&mdl->xreg[t], mdl->X, mdl->K, mdl->M, gX);
  if (opt->update_Y)
    for (int32_t t = 0; t < mdl->n_yreg; ++t) loss += o_regterm_apply(&mdl->yreg[t], mdl->Y, mdl->K, mdl->N, gY);
  if (ul) { /* SequenceReg: sum(map((f,x)->f(x), regs, layers))  regularizers.jl:903-905 */
    const int fl = opt->frozen_layers, fr = opt->frozen_regs;
    const int batch = mdl->has_batch && mdl->n_bv > 0;
    if (mdl->has_colreg && !((fl | fr) & 1))
      loss += o_colparamreg(mdl->n_cr, mdl->cr_start1, mdl->cr_stop1, mdl->cr_w_logsigma, mdl->cr_c_logsigma,
                            mdl->logsigma, g_logsigma);
    if (mdl->has_batchreg && batch && !((fl | fr) & 2))
      loss += o_batcharrayreg(mdl->n_bv, mdl->bv_start1, mdl->bv_stop1, mdl->bv_nb, mdl->bv_off, mdl->bvb_off,
                              mdl->br_w_logdelta, mdl->br_c_logdelta, mdl->logdelta, g_logdelta);
    if (mdl->has_colreg && !((fl | fr) & 4))
      loss += o_colparamreg(mdl->n_cr, mdl->cr_start1, mdl->cr_stop1, mdl->cr_w_mu, mdl->cr_c_mu, mdl->mu, g_mu);
    if (mdl->has_batchreg && batch && !((fl | fr) & 8))
      loss += o_batcharrayreg(mdl->n_bv, mdl->bv_start1, mdl->bv_stop1, mdl->bv_nb, mdl->bv_off, mdl->bvb_off,
                              mdl->br_w_theta, mdl->br_c_theta, mdl->theta, g_theta);
  }
  return loss;
}

/* ------------------------------------------------------------------------- */
/* The epoch loop (SELF-SPECIFIED replacement for MatFac.fit!, called at fit.jl:24)
 *
 * for epoch = opts.epoch .. max_epochs:
 *    loss_e, grads = full loss/gradient at the current parameters
 *    every trainable parameter group takes one optimizer step (simultaneously)
 *    trace[n++] = loss_e
 *    if !isfinite(loss_e)                     -> "nonfinite"
 *    if n>1 and loss_e > loss_{e-1}           -> "loss_increase"   (fit.jl:63)
 *    if |loss_{e-1}-loss_e| < abs_tol         -> tol counter++ (abs) ; elseif |..|/|loss_e| < rel_tol -> counter++ (rel)
 *    else counter = 0 ; counter >= tol_max_iters -> "abs_tol" / "rel_tol"
 * falling out of the loop                      -> "max_epochs"
 * `epochs` returned = last epoch index executed (fit.jl:69 resumes from h["epochs"]).
 * ------------------------------------------------------------------------- */
int o_fit(o_model *mdl, const o_opts *opt, o_optstate *st, double *trace, int32_t trace_cap, int32_t *n_trace,
          int32_t *epochs_out) {
  const int64_t nX = o_param_len(mdl, 0), nY = o_param_len(mdl, 1), nB = o_param_len(mdl, 4);
  real *gX = opt->update_X ? (real *)malloc(sizeof(real) * (size_t)nX) : NULL;
  real *gY = opt->update_Y ? (real *)malloc(sizeof(real) * (size_t)nY) : NULL;
  const int ul = opt->update_col_layers;
  real *gls = ul ? (real *)malloc(sizeof(real) * (size_t)mdl->N) : NULL;
  real *gmu = ul ? (real *)malloc(sizeof(real) * (size_t)mdl->N) : NULL;
  real *gld = (ul && nB) ? (real *)malloc(sizeof(real) * (size_t)nB) : NULL;
  real *gth = (ul && nB) ? (real *)malloc(sizeof(real) * (size_t)nB) : NULL;
  if (!st->initialized) o_optstate_init(mdl, opt, st);
  int term = O_TERM_MAX_EPOCHS, tol_iters = 0, n = 0, last_epoch = opt->epoch - 1;
  double prev = 0;
  for (int epoch = opt->epoch; epoch <= opt->max_epochs; ++epoch) {
    const double loss = o_loss_and_grads(mdl, opt, gX, gY, gls, gmu, gld, gth, NULL);
    if (opt->update_X) o_step(mdl, opt, st, 0, mdl->X, gX);
    if (opt->update_Y) o_step(mdl, opt, st, 1, mdl->Y, gY);
    if (ul) {
      const int fl = opt->frozen_layers;
      if (!(fl & 1)) o_step(mdl, opt, st, 2, mdl->logsigma, gls);
      if (!(fl & 4)) o_step(mdl, opt, st, 3, mdl->mu, gmu);
      if (nB && !(fl & 2)) o_step(mdl, opt, st, 4, mdl->logdelta, gld);
      if (nB && !(fl & 8)) o_step(mdl, opt, st, 5, mdl->theta, gth);
    }
    if (n < trace_cap) trace[n] = loss;
    ++n;
    last_epoch = epoch;
    if (!isfinite(loss)) { term = O_TERM_NONFINITE; break; }
    if (n > 1) {
      const double diff = prev - loss;
      if (diff < 0) { term = O_TERM_LOSS_INCREASE; break; }
      int which = -1;
      if (fabs(diff) < (double)opt->abs_tol) which = O_TERM_ABS_TOL;
      else if (fabs(diff / loss) < (double)opt->rel_tol) which = O_TERM_REL_TOL;
      if (which >= 0) {
        if (++tol_iters >= opt->tol_max_iters) { term = which; break; }
      } else {
        tol_iters = 0;
      }
    }
    prev = loss;
  }
  *n_trace = n;
  *epochs_out = last_epoch;
  free(gX); free(gY); free(gls); free(gmu); free(gld); free(gth);
  return term;
}

/* ------------------------------------------------------------------------- */
/* Masked column / (batch, column) statistics used by the closed-form initialisers between the GD stages.
 * The MatFac helpers they restate are un-vendored; the formulas are SELF-SPECIFIED from their call sites:
 *   col_n      MF.column_nonnan            (fit.jl:140, 447; regularizers.jl:765)   #finite entries per column
 *   col_sum/sq -> MF.batched_column_nanvar (regularizers.jl:766)
 *   col_sqerr  MF.link_col_sqerr           (fit.jl:138, 444)    sum_i (invlink(z_ij) - D_ij)^2 over finite entries
 *   col_ssqg   MF.batched_column_ssq_grads (fit.jl:166)         sum_i (dloss/dz)^2
 *   b_n, b_sqerr  ba_map(isfinite) / ba_map(MF.sqerr_func)      (fit.jl:332, 355, 454-456; batch_array.jl:305-334)
 * use_factors = 0: X'Y treated as 0 (the reference zeroes X and Y around these calls, fit.jl:133-136, 160-163). */
void o_stats(const o_model *mdl, int use_factors, real *col_n, real *col_sum, real *col_sumsq, real *col_sqerr,
             real *col_ssqg, real *b_n, real *b_sqerr) {
  const int64_t M = mdl->M, N = mdl->N;
  real *Z = (real *)calloc((size_t)(M * N), sizeof(real));
  if (use_factors) o_xty(mdl->X, mdl->Y, mdl->K, 0, M, N, Z);
  o_layers_forward(mdl, Z, NULL, 0, M);
  int32_t *kind = (int32_t *)malloc(sizeof(int32_t) * (size_t)N);
  for (int32_t r = 0; r < mdl->n_noise; ++r)
    for (int64_t j = mdl->nz_start1[r] - 1; j < mdl->nz_stop1[r]; ++j) kind[j] = mdl->nz_kind[r];
  real *R2 = (real *)calloc((size_t)(M * N), sizeof(real));   /* squared residuals (0 where missing) */
  real *F = (real *)calloc((size_t)(M * N), sizeof(real));    /* finite indicator */
  for (int64_t j = 0; j < N; ++j) {
    double n = 0, s1 = 0, s2 = 0, se = 0, sg = 0;
    for (int64_t i = 0; i < M; ++i) {
      const float yf = mdl->D[i + M * j];
      if (!isfinite(yf)) continue;
      const real y = (real)yf, z = Z[i + M * j];
      real pred;
      if (kind[j] == O_KIND_NORMAL) pred = z;
      else if (kind[j] == O_KIND_BERNOULLI) pred = o_sigmoid(z);
      else pred = R_EXP(z);
      const real g = mdl->col_weight[j] * (pred - y);
      const real r = pred - y;
      n += 1; s1 += y; s2 += (double)y * y; se += (double)r * r; sg += (double)g * g;
      R2[i + M * j] = r * r;
      F[i + M * j] = 1;
    }
    if (col_n) col_n[j] = (real)n;
    if (col_sum) col_sum[j] = (real)s1;
    if (col_sumsq) col_sumsq[j] = (real)s2;
    if (col_sqerr) col_sqerr[j] = (real)se;
    if (col_ssqg) col_ssqg[j] = (real)sg;
  }
  if (mdl->has_batch && mdl->n_bv > 0 && b_n && b_sqerr) {
    const int64_t tot = mdl->bv_off[mdl->n_bv];
    memset(b_n, 0, sizeof(real) * (size_t)tot);
    memset(b_sqerr, 0, sizeof(real) * (size_t)tot);
    o_ba_colsums(F, M, M, mdl->n_bv, mdl->bv_start1, mdl->bv_stop1, mdl->bv_nb, mdl->bv_bor, M, mdl->bv_off, b_n);
    o_ba_colsums(R2, M, M, mdl->n_bv, mdl->bv_start1, mdl->bv_stop1, mdl->bv_nb, mdl->bv_bor, M, mdl->bv_off, b_sqerr);
  }
  free(Z); free(kind); free(R2); free(F);
}

int o_sizeof_real(void) { return (int)sizeof(real); }

#ifdef _OPENMP
#include <omp.h>
int o_num_threads(void) { return omp_get_max_threads(); }
/* bench.py's cpu_baseline leg sizes the team by the work (a 500 x 200 problem on 256 threads spends its time in the
 * OpenMP barriers) and by the CPUs the process may actually use (cgroup quota), not by the host's thread count */
void o_set_num_threads(int n) { omp_set_num_threads(n < 1 ? 1 : n); }
#else
int o_num_threads(void) { return 1; }
void o_set_num_threads(int n) { (void)n; }
#endif
