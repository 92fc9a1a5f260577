/* examples/fit_c.c -- the drop-in boundary from plain C: marshal -> pmf_fit -> unmarshal, exactly what the Julia shim
 * (pathmatfac.jl_amd/julia/PathMatFacHIP.jl, mf_fit!) does through ccall and the Python binding through ctypes.
 *
 *   gcc -O2 -Iinclude examples/fit_c.c -Lpathmatfac.jl_amd -lpmf_hip -Wl,-rpath,$PWD/pathmatfac.jl_amd -lm -o /tmp/fit_c
 *   /tmp/fit_c          (needs an MI355X; prints the loss trace of a 600 x 300, K = 8 Gaussian fit)
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "pmf_hip.h"

#define CHK(call)                                                         \
  do {                                                                    \
    if ((call) != 0) {                                                    \
      fprintf(stderr, "%s failed: %s\n", #call, pmf_last_error());        \
      return 1;                                                           \
    }                                                                     \
  } while (0)

static float frand(void) { return (float)rand() / (float)RAND_MAX - 0.5f; }

int main(void) {
  const int64_t M = 600, N = 300;
  const int K = 8;
  float *Xt = malloc(sizeof(float) * K * M), *Yt = malloc(sizeof(float) * K * N);
  float *X = malloc(sizeof(float) * K * M), *Y = malloc(sizeof(float) * K * N);
  float *D = malloc(sizeof(float) * M * N), *zeros = calloc(N, sizeof(float)), *ones = malloc(sizeof(float) * N);
  double trace[64];
  srand(1);
  for (int64_t e = 0; e < K * M; ++e) { Xt[e] = frand(); X[e] = 0.2f * frand(); }
  for (int64_t e = 0; e < K * N; ++e) { Yt[e] = frand(); Y[e] = 0.2f * frand(); }
  for (int64_t j = 0; j < N; ++j) {                       /* D is column-major, NaN = missing */
    ones[j] = 1.f;
    for (int64_t i = 0; i < M; ++i) {
      float z = 0.f;
      for (int k = 0; k < K; ++k) z += Xt[i * K + k] * Yt[j * K + k];
      D[j * M + i] = (i * 7 + j) % 50 == 0 ? NAN : z + 0.05f * frand();
    }
  }
  pmf_ctx *ctx = NULL;
  CHK(pmf_create(0, &ctx));
  CHK(pmf_set_data(ctx, D, M, N, PMF_STORE_F32));
  CHK(pmf_set_factors(ctx, X, Y, K));
  CHK(pmf_set_col_params(ctx, zeros, zeros));            /* logsigma = 0, mu = 0 */
  CHK(pmf_set_n_batch_views(ctx, 0));
  const int64_t s1 = 1, e1 = N;                           /* 1-based inclusive, as a Julia UnitRange */
  const int32_t kind = PMF_NOISE_NORMAL;
  CHK(pmf_set_noise(ctx, 1, &s1, &e1, &kind, ones));
  CHK(pmf_clear_xreg(ctx));
  CHK(pmf_clear_yreg(ctx));
  CHK(pmf_set_optimizer(ctx, PMF_OPT_ADAGRAD, 0.05f, 1e-8f, 0.9f, 0.999f));
  pmf_fit_opts o = {0};
  o.update_X = 1; o.update_Y = 1; o.max_epochs = 40; o.epoch = 1; o.tol_max_iters = 3; o.keep_trace = 1;
  o.abs_tol = 1e-9; o.rel_tol = 1e-9; o.capacity = 100000000;
  pmf_fit_result r = {0};
  r.loss_trace = trace; r.trace_cap = 64;
  CHK(pmf_fit(ctx, &o, &r));
  CHK(pmf_get_factors(ctx, X, Y));
  printf("term_code %d after epoch %d: loss %.6g -> %.6g\n", r.term_code, r.epochs, trace[0], r.final_loss);
  CHK(pmf_destroy(ctx));
  return r.final_loss < trace[0] ? 0 : 2;
}
