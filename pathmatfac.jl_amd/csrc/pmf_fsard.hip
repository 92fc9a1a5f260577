// pmf_fsard.hip -- FeatureSetARD outer loop on the device (pmf_fsard_update_A).
#include "pmf_ctx.h"

// ------------------------------------------------------------------------------------------------
// FeatureSetARD outer loop: update_A! (src/featureset_ard.jl:214-294) -- nonnegative projected ISTA with AdaGrad step
// sizes (ISTAOptimiser, src/optimizers.jl:26-62) on the view's assignment matrix A (L feature sets x K factors),
// minimising gamma_normal_loss(A) + sum_k lambda_k |A_lk| (src/featureset_ard.jl:154-162 and its rrule :164-186; the
// primal of the rrule is never used for a value: update_A_inner! calls the plain function).  Up to max_epochs serial
// iterations of two small kernels per view, no host round trip inside a batch of iterations:
//   k_fsard_grad  : (column slices of the view) A'S, the loss at A, grad_{A'S}, partial S * grad' per workgroup
//   k_fsard_step  : (one workgroup) fixed-order sum of the partials, loss bookkeeping exactly as update_A_inner!
//                   (best loss, A_best, termination counter), then the ISTA update of A unless the loop has ended.
// Y stays where the fit left it (the context's device copy); beta = beta0 (v0 + A'S) is written straight into the
// Y regularizer's device array (featureset_ard.jl:292) as well as returned.
// ------------------------------------------------------------------------------------------------
struct FsardArgs {
  const float *Y;          // context Y, Kp x N, column j at Y + j*Kp
  const float *S;          // L x Nv row-major (feature set l's weights over the view's columns)
  const float *alpha;      // Nv
  const float *lambda;     // K
  float *A, *A_best, *ssq; // L x K (k contiguous)
  float *gpart;            // [n_wg][L*K] partial gradients
  double *lpart;           // [n_wg] partial data losses
  double *state;           // [0] best loss, [1] loss of the last evaluated A, [4] calibration constant (unused)
  int32_t *istate;         // [0] done, [1] term_count, [2] evaluations so far, [3] max_epochs, [4] term_iter
  float *beta_dev;         // ard_beta + c0*Kp (may be null)
  int64_t c0, Nv;
  int32_t L, K, Kp, CW, n_wg;
  float alpha0, v0, lr;
  double atol;
};

// thread = column (CW = blockDim.x columns per sub-slice); A in LDS; the sub-slice's grad_{A'S} tile [K][CW] in LDS;
// then thread -> (l, k) outputs, each a dot product over the sub-slice.  final_beta != 0: only write beta (no gradient).
__global__ __launch_bounds__(256) void k_fsard_grad(const FsardArgs a, int final_beta) {
  extern __shared__ __attribute__((aligned(16))) char smem_fs[];
  float *As = reinterpret_cast<float *>(smem_fs);      // [L][K]
  float *Gt = As + a.L * a.K;                           // [K][CW + 1] (odd row stride: the dot products below read a column of rows)
  const int GS = a.CW + 1;
  __shared__ double sh[4];
  if (!final_beta && a.istate[0]) return;               // the loop has ended: nothing left to evaluate
  const int tid = threadIdx.x, NT = blockDim.x;
  const float *Asrc = final_beta ? a.A_best : a.A;
  for (int e = tid; e < a.L * a.K; e += NT) As[e] = Asrc[e];
  const int64_t per = (a.Nv + a.n_wg - 1) / a.n_wg;
  const int64_t j_lo = blockIdx.x * per, j_hi = j_lo + per < a.Nv ? j_lo + per : a.Nv;
  const float beta0 = a.alpha0 - 1.f;
  constexpr int MAXO = 64;                              // (l, k) outputs per thread: L*K <= 64 * 256
  float gacc[MAXO];
#pragma unroll
  for (int o = 0; o < MAXO; ++o) gacc[o] = 0.f;
  double lacc = 0.0;
  __syncthreads();
  for (int64_t js = j_lo; js < j_hi; js += a.CW) {
    const int64_t j = js + tid;
    const bool live = tid < a.CW && j < j_hi;
    // ---- A'S for this column, k in chunks of 32 (accumulators in registers)
    for (int k0 = 0; k0 < a.K; k0 += 32) {
      float acc[32];
#pragma unroll
      for (int q = 0; q < 32; ++q) acc[q] = 0.f;
      if (live) {
        for (int l = 0; l < a.L; ++l) {
          const float sv = a.S[(int64_t)l * a.Nv + j];
          if (sv != 0.f) {
            const float *ar = As + l * a.K + k0;
#pragma unroll
            for (int q = 0; q < 32; ++q)
              if (k0 + q < a.K) acc[q] = fmaf(ar[q], sv, acc[q]);
          }
        }
      }
      const float al = live ? a.alpha[j] : 0.f;
#pragma unroll
      for (int q = 0; q < 32; ++q) {
        const int k = k0 + q;
        if (k < a.K) {
          float g = 0.f;
          if (live) {
            const float beta = beta0 * (a.v0 + acc[q]);
            if (final_beta) {
              if (a.beta_dev) a.beta_dev[(j) * a.Kp + k] = beta;
            } else {
              const float y = a.Y[(a.c0 + j) * a.Kp + k];
              const float b2 = beta + 0.5f * y * y;
              lacc += (double)(-al * logf(beta) + (al + 0.5f) * logf(b2) - logf(fabsf(y) + 1e-9f));
              g = beta0 * (-al / beta + (al + 0.5f) / b2);
            }
          }
          if (!final_beta && tid < a.CW) Gt[k * GS + tid] = g;
        }
      }
      if (!final_beta && live && k0 == 0) lacc -= (double)((al + 0.5f) * logf(al + 0.5f) - al * logf(al));   // calibration term, once per column
    }
    if (final_beta) continue;
    __syncthreads();
    // ---- partial grad_A[l][k] += sum_j S[l][js + j] * Gt[k][j]
    const int ncol = (int)(j_hi - js < a.CW ? j_hi - js : a.CW);
#pragma unroll
    for (int o = 0; o < MAXO; ++o) {
      const int e = tid + o * NT;
      if (e < a.L * a.K) {
        const int l = e / a.K, k = e - l * a.K;
        const float *sr = a.S + (int64_t)l * a.Nv + js;
        const float *gr = Gt + k * GS;
        float sacc = 0.f;
        for (int jj = 0; jj < ncol; ++jj) {
          const float sv = sr[jj];                       // (S is sparse: most feature sets skip most columns)
          if (sv != 0.f) sacc = fmaf(sv, gr[jj], sacc);
        }
        gacc[o] += sacc;
      }
    }
    __syncthreads();
  }
  if (final_beta) return;
#pragma unroll
  for (int o = 0; o < MAXO; ++o) {
    const int e = tid + o * NT;
    if (e < a.L * a.K) a.gpart[(int64_t)blockIdx.x * a.L * a.K + e] = gacc[o];
  }
  const double ls = block_reduce_sum(lacc, sh);
  if (tid == 0) a.lpart[blockIdx.x] = ls;
}

// one workgroup: loss(A_t) = sum of partials + sum lambda_k |A_lk|; bookkeeping of update_A_inner! (:239-268); then
// ISTAOptimiser.update! (optimizers.jl:46-62) unless the loop has ended
__global__ __launch_bounds__(1024) void k_fsard_step(const FsardArgs a) {
  __shared__ double sh[16];
  __shared__ int s_improved, s_done;
  if (a.istate[0]) return;
  const int tid = threadIdx.x, NT = blockDim.x, n = a.L * a.K;
  double reg = 0.0;
  for (int e = tid; e < n; e += NT) reg += (double)(a.lambda[e % a.K] * fabsf(a.A[e]));
  // block reduce (16 waves)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) reg += __shfl_xor(reg, off, 64);
  if ((tid & 63) == 0) sh[tid >> 6] = reg;
  __syncthreads();
  if (tid == 0) {
    double loss = 0.0;
    for (int q = 0; q < (NT >> 6); ++q) loss += sh[q];
    for (int w = 0; w < a.n_wg; ++w) loss += a.lpart[w];
    const int evals = a.istate[2];
    int improved = 0, term = a.istate[1];
    if (evals == 0) { a.state[0] = loss; improved = 1; }                 // "Iteration 0": best_loss = loss(A), A_best = A
    else if (loss < a.state[0]) {
      const double diff = a.state[0] - loss;
      a.state[0] = loss;
      improved = 1;
      term = diff > a.atol ? 0 : term + 1;
    } else {
      term += 1;
    }
    a.state[1] = loss;
    a.istate[1] = term;
    a.istate[2] = evals + 1;
    const int done = (term >= a.istate[4]) || (evals >= a.istate[3]);    // evals == max_epochs: the last update has been evaluated
    s_improved = improved;
    s_done = done;
  }
  __syncthreads();
  const bool improved = s_improved != 0, done = s_done != 0;
  for (int e = tid; e < n; e += NT) {
    float av = a.A[e];
    if (improved) a.A_best[e] = av;
    if (!done) {
      float g = 0.f;
      for (int w = 0; w < a.n_wg; ++w) g += a.gpart[(int64_t)w * n + e];   // fixed order
      const float ssq = a.ssq[e] + g * g;                                 // optimizers.jl:50
      a.ssq[e] = ssq;
      const float eta = a.lr / sqrtf(ssq);                                // :51
      av = fmaxf(av - eta * g, 0.f);                                      // :55-56
      av = fmaxf(fabsf(av) - a.lambda[e % a.K] * eta, 0.f);               // ist_proj! :40-42, :61
      a.A[e] = av;
    }
  }
  __syncthreads();
  if (tid == 0 && done) a.istate[0] = 1;
}

extern "C" int pmf_fsard_update_A(pmf_ctx *c, int64_t col_start1, int64_t col_stop1, int L, const float *S, const float *alpha,
                                  const float *lambda, float alpha0, float v0, float lr, float *ssq_grad, float *A,
                                  int max_epochs, int term_iter, double atol, double *best_loss, int *epochs_run,
                                  float *beta_out) {
  PMFCHK(ctx_bind(c));
  if (c->K == 0) return pmf_fail("factors not set");
  if (col_start1 < 1 || col_stop1 > c->N || col_start1 > col_stop1) return pmf_fail("bad column range %lld:%lld", (long long)col_start1, (long long)col_stop1);
  if (L <= 0 || !S || !alpha || !lambda || !ssq_grad || !A) return pmf_fail("null / empty argument");
  const int K = c->K;
  const int64_t Nv = col_stop1 - col_start1 + 1, n = (int64_t)L * K;
  if (n > 64 * 256) return pmf_fail("L x K = %lld exceeds the ISTA kernel's capacity (16384)", (long long)n);
  int CW = 256;
  while (CW > 32 && (size_t)(n + (int64_t)K * (CW + 1)) * 4 > 150 * 1024) CW >>= 1;
  if ((size_t)(n + (int64_t)K * (CW + 1)) * 4 > 150 * 1024) return pmf_fail("L x K too large for LDS");
  const int n_wg = (int)std::max<int64_t>(1, std::min<int64_t>(32, (Nv + CW - 1) / CW));
  // device buffers (freed on return)
  float *dS = nullptr, *dal = nullptr, *dlam = nullptr, *dA = nullptr, *dAb = nullptr, *dssq = nullptr, *dgp = nullptr;
  double *dlp = nullptr, *dst = nullptr;
  int32_t *dis = nullptr;
  auto cleanup = [&]() { dev_free(&dS); dev_free(&dal); dev_free(&dlam); dev_free(&dA); dev_free(&dAb); dev_free(&dssq); dev_free(&dgp); dev_free(&dlp); dev_free(&dst); dev_free(&dis); };
  int rc = 0;
  do {
    if ((rc = dev_alloc(&dS, (size_t)(L * Nv), false)) < 0) break;
    if ((rc = dev_alloc(&dal, (size_t)Nv, false)) < 0) break;
    if ((rc = dev_alloc(&dlam, (size_t)K, false)) < 0) break;
    if ((rc = dev_alloc(&dA, (size_t)n)) < 0) break;          // A .= 0 (featureset_ard.jl:286)
    if ((rc = dev_alloc(&dAb, (size_t)n)) < 0) break;
    if ((rc = dev_alloc(&dssq, (size_t)n, false)) < 0) break;
    if ((rc = dev_alloc(&dgp, (size_t)(n_wg * n))) < 0) break;
    if ((rc = dev_alloc(&dlp, (size_t)n_wg)) < 0) break;
    if ((rc = dev_alloc(&dst, (size_t)8)) < 0) break;
    if ((rc = dev_alloc(&dis, (size_t)8)) < 0) break;
    hipError_t e = hipMemcpy(dS, S, sizeof(float) * (size_t)(L * Nv), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dal, alpha, sizeof(float) * (size_t)Nv, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dlam, lambda, sizeof(float) * (size_t)K, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dssq, ssq_grad, sizeof(float) * (size_t)n, hipMemcpyHostToDevice);
    const int32_t is0[8] = {0, 0, 0, max_epochs, term_iter, 0, 0, 0};
    if (e == hipSuccess) e = hipMemcpy(dis, is0, sizeof(is0), hipMemcpyHostToDevice);
    if (e != hipSuccess) { rc = pmf_fail("pmf_fsard_update_A: %s", hipGetErrorString(e)); break; }
    FsardArgs a;
    memset(&a, 0, sizeof(a));
    a.Y = c->P[1].p; a.S = dS; a.alpha = dal; a.lambda = dlam; a.A = dA; a.A_best = dAb; a.ssq = dssq; a.gpart = dgp; a.lpart = dlp;
    a.state = dst; a.istate = dis; a.beta_dev = (c->has_ard && c->ard_beta) ? c->ard_beta + (col_start1 - 1) * c->Kp : nullptr;
    a.c0 = col_start1 - 1; a.Nv = Nv; a.L = L; a.K = K; a.Kp = c->Kp; a.CW = CW; a.n_wg = n_wg;
    a.alpha0 = alpha0; a.v0 = v0; a.lr = lr; a.atol = atol;
    const size_t lds = (size_t)(n + (int64_t)K * (CW + 1)) * 4;
    if ((rc = ensure_dyn_lds(c, (const void *)k_fsard_grad, lds)) < 0) break;
    int32_t h_is[8];
    // evaluations 0 .. max_epochs (the update after evaluation t gives A_{t+1}); the host looks at the `done` flag once
    // per batch of iterations
    const int batch = std::max(8, term_iter);
    bool done = false;
    for (int t = 0; t <= max_epochs && !done; t += batch) {
      for (int q = 0; q < batch && t + q <= max_epochs; ++q) {
        hipLaunchKernelGGL(k_fsard_grad, dim3(n_wg), dim3(256), lds, c->stream, a, 0);
        hipLaunchKernelGGL(k_fsard_step, dim3(1), dim3(1024), 0, c->stream, a);
      }
      e = hipGetLastError();
      if (e == hipSuccess) e = hipMemcpyAsync(h_is, dis, sizeof(h_is), hipMemcpyDeviceToHost, c->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
      if (e != hipSuccess) { rc = pmf_fail("pmf_fsard_update_A: %s", hipGetErrorString(e)); break; }
      done = h_is[0] != 0;
    }
    if (rc < 0) break;
    // A .= A_best ; beta[:, cr] = beta0 (v0 + A'S)   (featureset_ard.jl:272, 292)
    float *dbeta_tmp = nullptr;
    if (beta_out && !a.beta_dev) {   // no device regularizer array to write into: a temporary with the view's columns
      if ((rc = dev_alloc(&dbeta_tmp, (size_t)(Nv * c->Kp))) < 0) break;
      a.beta_dev = dbeta_tmp;
    }
    if (a.beta_dev) hipLaunchKernelGGL(k_fsard_grad, dim3(n_wg), dim3(256), lds, c->stream, a, 1);
    double h_st[8];
    e = hipMemcpyAsync(h_st, dst, sizeof(h_st), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(h_is, dis, sizeof(h_is), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(A, dAb, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(ssq_grad, dssq, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost);
    if (e == hipSuccess && beta_out)
      e = hipMemcpy2D(beta_out, sizeof(float) * K, a.beta_dev, sizeof(float) * c->Kp, sizeof(float) * K, (size_t)Nv, hipMemcpyDeviceToHost);
    dev_free(&dbeta_tmp);
    if (e != hipSuccess) { rc = pmf_fail("pmf_fsard_update_A: %s", hipGetErrorString(e)); break; }
    if (best_loss) *best_loss = h_st[0];
    if (epochs_run) *epochs_run = h_is[2] - 1;
  } while (0);
  cleanup();
  return rc;
}
