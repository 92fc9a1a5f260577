/* examples/fit_c.c -- the drop-in boundary from plain C: marshal -> pmf_fit -> unmarshal, exactly what the Julia shim
 * (pathmatfac.jl_amd/julia/PathMatFacHIP.jl, mf_fit!) does through ccall and the Python binding through ctypes.
 *
 *   gcc -O2 -Iinclude examples/fit_c.c -Lpathmatfac.jl_amd -lpmf_hip -Wl,-rpath,$PWD/pathmatfac.jl_amd -lm -o /tmp/fit_c
 *   /tmp/fit_c          (needs an MI355X; prints the loss trace of a 600 x 300, K = 8 Gaussian fit)
 *
 * The fit is then repeated with a one-rank RCCL communicator attached (pmf_comm_get_unique_id / pmf_comm_init: the
 * multi-GPU entry points, exactly as a process-per-GPU host calls them with its own rank): the collectives run and the
 * results must be bit-identical.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
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

  /* the same fit as rank 0 of a one-rank job */
  float *X2 = malloc(sizeof(float) * K * M), *Y2 = malloc(sizeof(float) * K * N);
  double trace2[64];
  char uid[PMF_COMM_ID_BYTES];
  srand(1);
  for (int64_t e = 0; e < K * M; ++e) { (void)frand(); X2[e] = 0.2f * frand(); }
  for (int64_t e = 0; e < K * N; ++e) { (void)frand(); Y2[e] = 0.2f * frand(); }
  CHK(pmf_comm_get_unique_id(uid));                       /* rank 0 creates it and hands it to every rank */
  CHK(pmf_comm_init(ctx, 0, 1, uid));
  CHK(pmf_set_factors(ctx, X2, Y2, K));
  CHK(pmf_set_optimizer(ctx, PMF_OPT_ADAGRAD, 0.05f, 1e-8f, 0.9f, 0.999f));
  pmf_fit_result r2 = {0};
  r2.loss_trace = trace2; r2.trace_cap = 64;
  CHK(pmf_fit(ctx, &o, &r2));
  CHK(pmf_get_factors(ctx, X2, Y2));
  int rank = -1, nranks = -1, transport = -1, chunks = -1, cus = -1;
  int64_t ncoll = 0;
  CHK(pmf_comm_info(ctx, &rank, &nranks, &transport, &chunks, &cus, &ncoll));
  CHK(pmf_comm_destroy(ctx));
  const int same = r2.epochs == r.epochs && r2.term_code == r.term_code &&
                   memcmp(trace, trace2, sizeof(double) * (size_t)r.n_trace) == 0 &&
                   memcmp(X, X2, sizeof(float) * K * M) == 0 && memcmp(Y, Y2, sizeof(float) * K * N) == 0;
  printf("one-rank RCCL communicator: rank %d of %d, transport %d, %lld collectives, results %s\n", rank, nranks, transport,
         (long long)ncoll, same ? "bit-identical" : "DIFFER");
  CHK(pmf_destroy(ctx));
  if (!same || transport != PMF_COMM_RCCL || ncoll < 2 * r.epochs) return 3;
  return r.final_loss < trace[0] ? 0 : 2;
}
