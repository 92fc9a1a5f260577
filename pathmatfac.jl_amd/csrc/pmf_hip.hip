// pmf_hip.hip -- libpmf_hip.so : MI355X (gfx950) implementation of PathMatFac's fit! loop behind the C ABI of
// include/pmf_hip.h.  Written for CDNA4 only (wave64, v_mfma_f32_32x32x2_f32, 160 KiB LDS).
#include "pmf_ctx.h"

// ------------------------------------------------------------------------------------------------
// error handling
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
int pmf_fail(const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return -1;
}
extern "C" const char *pmf_last_error(void) { return g_err.c_str(); }
extern "C" int pmf_version(void) { return 1; }


int ctx_bind(pmf_ctx *c) {
  if (!c) return pmf_fail("null context");
  HIPCHK(hipSetDevice(c->device));
  return 0;
}

// hipMemset on device memory is queued on the NULL stream and may return before it has run; the library's kernels and
// copies run on a NON-BLOCKING stream, which the NULL stream does not order.  Without the wait a later upload on the
// library's stream can be overtaken by the fill (seen with two processes sharing one GPU: a regularizer's beta zeroed
// after its upload).
int memset_now(void *p, int v, size_t bytes) {
  HIPCHK(hipMemset(p, v, bytes));
  HIPCHK(hipStreamSynchronize(nullptr));
  return 0;
}

static int param_alloc(ParamBuf &b, int64_t n) {
  b.n = n;
  PMFCHK(dev_alloc(&b.p, n));
  PMFCHK(dev_alloc(&b.g, n));
  PMFCHK(dev_alloc(&b.acc, n));
  PMFCHK(dev_alloc(&b.mom, n));
  dev_free(&b.wq);
  dev_free(&b.cq);
  return 0;
}
static void param_free(ParamBuf &b) {
  dev_free(&b.p); dev_free(&b.g); dev_free(&b.acc); dev_free(&b.mom); dev_free(&b.wq); dev_free(&b.cq);
  b.n = 0;
}

int pmf_ensure_dyn_lds(PmfDynLds *cache, const void *kern, size_t lds) {
  auto it = cache->find(kern);
  if (it != cache->end() && it->second >= lds) return 0;
  HIPCHK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  (*cache)[kern] = lds;
  return 0;
}
int ensure_dyn_lds(pmf_ctx *c, const void *kern, size_t lds) { return pmf_ensure_dyn_lds(&c->dyn_lds, kern, lds); }

static int ensure_scratch(pmf_ctx *c, size_t bytes) {
  if (c->scratch_bytes >= bytes) return 0;
  if (c->scratch) HIPCHK(hipFree(c->scratch));
  c->scratch = nullptr;
  HIPCHK(hipMalloc(&c->scratch, bytes));
  c->scratch_bytes = bytes;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// small kernels
// ------------------------------------------------------------------------------------------------
__global__ void k_fill(float *p, int64_t n, float v) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) p[e] = v;
}

// colp[j] = {exp(logsigma_j), mu_j, w_j, meta_j}; btab[e] = {exp(logdelta_e), theta_e}
__global__ void k_prepare(const float *logsigma, const float *mu, const float *colw, const int32_t *colmeta,
                          float4 *colp, int64_t N, const float *logdelta, const float *theta, float2 *btab,
                          int64_t nbt) {
  const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e < N) colp[e] = make_float4(expf(logsigma[e]), mu[e], colw[e], __int_as_float(colmeta[e]));
  if (e < nbt) btab[e] = make_float2(expf(logdelta[e]), theta[e]);
}

// Dense form of the batch tables for the fused kernel: btd[j*16 + b] = {exp(logdelta), theta} of (column j's view,
// batch b), identity {1, 0} for b >= the view's batch count (slot 15 is always identity: rows outside every batch),
// for columns outside every view and for the pad columns up to the next multiple of 32.  A tile's 32 columns are then
// one contiguous 4-KiB read, staged in LDS, and the epilogue needs no view arithmetic.
struct DenseBtabArgs {
  const int32_t *colmeta;
  const float2 *btab;
  float2 *btd;
  uint8_t *colview;
  int64_t N, Npad;
  int32_t nbs_shift;
  ViewDesc views[PMF_MAXV];
};
// (nbs = 1 << nbs_shift slots per column; the LAST slot is always the identity: rows outside every batch)
__global__ void k_dense_btab(const DenseBtabArgs a) {
  const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= (a.Npad << a.nbs_shift)) return;
  const int64_t j = e >> a.nbs_shift;
  const int b = (int)(e & ((1 << a.nbs_shift) - 1));
  float2 out = make_float2(1.f, 0.f);
  int v = -1;
  if (j < a.N) {
    v = (a.colmeta[j] >> 2) - 1;
    if (v >= 0 && b < a.views[v].nb && b < (1 << a.nbs_shift) - 1) out = a.btab[a.views[v].tab_off + (j - a.views[v].c0) * a.views[v].nb + b];
  }
  a.btd[e] = out;
  if (b == 0) a.colview[j] = (uint8_t)(v < 0 ? 255 : v);
}

// dense quadratic weights from ranges: wq[k, i] += p * w[g, k] for i in range g   (GroupRegularizer / L2Regularizer)
__global__ void k_expand_group(float *wq, int Kp, int K, int64_t n, const int64_t *s1, const int64_t *e1,
                               const float *w, int n_groups, float p) {
  const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= n * Kp) return;
  const int k = (int)(e % Kp);
  const int64_t i = e / Kp;
  if (k >= K) return;
  float add = 0.f;
  for (int g = 0; g < n_groups; ++g)
    if (i + 1 >= s1[g] && i + 1 <= e1[g]) add += w[(int64_t)g * K + k];
  wq[e] += p * add;
}
// same for a 1-D parameter (ColParamReg): per range weight and centre
__global__ void k_expand_colparam(float *wq, float *cq, int64_t n, const int64_t *s1, const int64_t *e1,
                                  const float *w, const float *c, int n_ranges) {
  const int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (j >= n) return;
  float ww = 0.f, cc = 0.f;
  for (int r = 0; r < n_ranges; ++r)
    if (j + 1 >= s1[r] && j + 1 <= e1[r]) { ww = w[r]; cc = c[r]; }
  wq[j] = ww;
  cq[j] = cc;
}
// BatchArrayReg: per (view, batch) weight and centre broadcast over the view's columns
__global__ void k_expand_batchreg(float *wq, float *cq, int64_t n, const int32_t *val_view, const int64_t *val_off,
                                  const int32_t *nbs, const int64_t *bvb_off, const float *w, const float *c) {
  const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= n) return;
  const int v = val_view[e];
  const int b = (int)((e - val_off[v]) % nbs[v]);
  wq[e] = w[bvb_off[v] + b];
  cq[e] = c[bvb_off[v] + b];
}
// ARDRegularizer ranges -> dense alpha[N], beta[Kp x N]
__global__ void k_expand_ard(float *alpha, float *beta, int Kp, int64_t N, const int64_t *s1, const int64_t *e1,
                             const float *a, const float *b, int n_ranges) {
  const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= N * Kp) return;
  const int64_t j = e / Kp;
  float aa = 0.f, bb = 1.f;
  bool hit = false;
  for (int r = 0; r < n_ranges; ++r)
    if (j + 1 >= s1[r] && j + 1 <= e1[r]) { aa = a[r]; bb = b[r]; hit = true; }
  // columns outside every range are not regularized: encode as alpha = -0.5 (factor (0.5+alpha) = 0), beta = 1
  if (!hit) { aa = -0.5f; bb = 1.f; }
  beta[e] = bb;
  if (e % Kp == 0) alpha[j] = aa;
}
__global__ void k_pad_copy(float *dst, const float *src, int Kp, int K, int64_t n, float padval) {
  const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= n * Kp) return;
  const int k = (int)(e % Kp);
  dst[e] = k < K ? src[(e / Kp) * K + k] : padval;
}

// column-major source block (rows 0..M-1, columns col0..col0+ncols-1, leading dimension M) -> tile-major D
// (bf16 storage: round to nearest even, NaN stays NaN)
__global__ void k_tile_D(const float *src, int64_t M, int64_t col0, int64_t ncols, void *dst, int64_t nRB, int bf16) {
  const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= M * ncols) return;
  const int64_t i = e % M, jl = e / M;
  if (bf16) reinterpret_cast<__bf16 *>(dst)[pmf_d_off16(i, col0 + jl, nRB)] = (__bf16)src[e];
  else reinterpret_cast<float *>(dst)[pmf_d_off(i, col0 + jl, nRB)] = src[e];
}
__global__ void k_fill16(uint16_t *p, int64_t n, uint16_t v) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) p[e] = v;
}
// one entry of the tile-major data matrix, whatever its storage type (the rarely-run scalar kernels)
__device__ __forceinline__ float pmf_d_get(const void *D, int64_t i, int64_t j, int64_t nRB, int bf16) {
  if (bf16) return __uint_as_float((uint32_t)reinterpret_cast<const uint16_t *>(D)[pmf_d_off16(i, j, nRB)] << 16);
  return reinterpret_cast<const float *>(D)[pmf_d_off(i, j, nRB)];
}


struct StepArgs {
  float *p, *g, *acc, *mom;
  const float *wq, *cq;          // quadratic regularizer (may be null)
  const float *ard_alpha, *ard_beta;  // ARD-type regularizer (Y only; may be null)
  float ard_scale;
  int64_t n;   // elements
  int Kp, K;   // leading dimension and number of live rows (Kp == K == 1 for vectors)
  int opt_kind;
  float lr, eps, b1, b2, c1, c2;  // c1 = 1 - beta1^t, c2 = 1 - beta2^t
  int do_step;   // 0: only evaluate the regularizer loss (parameter frozen for stepping)
  int use_reg;
  double *reg_partial;  // [REG_SLOTS]
};

// Regularizer gradient + optimizer step, one pass over a parameter tensor.
//   quadratic : 0.5*wq*(p-cq)^2  (L2Regularizer regularizers.jl:21-33, GroupRegularizer :423-446,
//               ColParamReg :482-487, BatchArrayReg :795-815)
//   ARD       : (0.5+alpha_j) log(1 + (0.5/beta) p^2), grad (alpha_j+0.5) p / (b beta)
//               (ARDRegularizer :546-585, FeatureSetARDReg featureset_ard.jl:135-150)
//   AdaGrad   : acc += g^2 ; p -= eta g/(sqrt(acc)+eps)   (optimizers.jl:6-13 ; acc starts at eps)
//   Adam      : Flux.Optimise.Adam
__global__ __launch_bounds__(256) void k_reg_step(const StepArgs a) {
  __shared__ double sh[4];
  double lacc = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < a.n; e += stride) {
    const int k = (int)(e % a.Kp);
    if (k >= a.K) continue;
    float p = a.p[e];
    float g = a.g[e];
    if (a.use_reg) {
      if (a.wq) {
        const float d = p - (a.cq ? a.cq[e] : 0.f);
        const float gr = a.wq[e] * d;
        lacc += 0.5 * (double)(gr * d);
        g += gr;
      }
      if (a.ard_beta) {
        const int64_t j = e / a.Kp;
        const float be = a.ard_beta[e], al = a.ard_alpha[j];
        const float b = 1.f + (0.5f / be) * (p * p);
        lacc += (double)(a.ard_scale * (0.5f + al) * logf(b));
        g += a.ard_scale * ((al + 0.5f) * p / (b * be));
      }
    }
    if (a.do_step) {
      if (a.opt_kind == PMF_OPT_ADAGRAD) {
        const float acc = a.acc[e] + g * g;
        a.acc[e] = acc;
        p -= g * (a.lr / (sqrtf(acc) + a.eps));
      } else {
        const float m = a.b1 * a.mom[e] + (1.f - a.b1) * g;
        const float v = a.b2 * a.acc[e] + (1.f - a.b2) * g * g;
        a.mom[e] = m;
        a.acc[e] = v;
        p -= m / a.c1 / (sqrtf(v / a.c2) + a.eps) * a.lr;
      }
      a.p[e] = p;
    }
  }
  const double s = block_reduce_sum(lacc, sh);
  if (threadIdx.x == 0 && a.reg_partial) a.reg_partial[blockIdx.x] = s;
}

// fixed-order reduction of the loss partial slabs -> out[0..4]
__global__ __launch_bounds__(256) void k_loss_reduce(const double *data_partial, int64_t n_data, const double *reg_partial,
                                                     const RegCounts reg_counts, double *out, int mask) {
  __shared__ double sh[4];
  for (int which = 0; which < 5; ++which) {
    if (!((mask >> which) & 1)) continue;   // (uniform) the pipelined loop of pmf_fit reduces the rank-local and the replicated terms separately
    const double *src = which == 0 ? data_partial : reg_partial + (int64_t)(which - 1) * REG_SLOTS;
    const int64_t n = which == 0 ? n_data : reg_counts.c[which - 1];
    double v = 0.0;
    for (int64_t e = threadIdx.x; e < n; e += blockDim.x) v += src[e];
    const double s = block_reduce_sum(v, sh);
    if (threadIdx.x == 0) out[which] = s;
    __syncthreads();
  }
}
int launch_loss_reduce(pmf_ctx *c, const RegCounts &rc, int mask) {
  k_loss_reduce<<<1, 256, 0, c->stream>>>(c->loss_partial, c->n_macro, c->reg_partial, rc, c->d_loss, mask);
  HIPCHK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Layer-parameter gradient pass (update_col_layers stages S2/S7: init_theta! fit.jl:106-122, fit_joint :987-1000).
// thread = column, sequential over a chunk of rows; per-(batch, column) sums live in LDS (no contention:
// a thread only touches its own column).  Pull-backs as coded in the reference:
//   theta_bar[b,j]    = sum_{i in b} g                      (batch_array.jl:141-143)
//   mu_bar[j]         = sum_i g                             (layers.jl:83)
//   logdelta_bar[b,j] = delta[b,j] * sum_{i in b} g * (a*sigma_j)   (batch_array.jl:203-204, 250)
//   logsigma_bar[j]   = sigma_j * sum_i g * delta           (layers.jl:40-41; Q1: omits the input factor)
// ------------------------------------------------------------------------------------------------
struct LayerGradArgs {
  const void *D;
  int d_bf16;
  const float *X, *Y;
  const float4 *colp;
  const int32_t *bor;
  const float2 *btab;
  float *g_logsigma, *g_mu, *g_logdelta, *g_theta;  // any may be null (only nullness is used by the kernel: see part)
  // per block row (blockIdx.y) one private vector [mu N][logsigma N][theta nbt][logdelta nbt] of part_stride floats, every entry
  // written by exactly one thread; k_sum_parts adds the block rows in fixed order (no float atomics: bitwise reproducible)
  float *part;
  int64_t part_stride, nbt;
  double *loss_partial;                             // may be null; one slot per block (flattened grid)
  int64_t M, N, nRB;
  int Kp, K, rows_per_block, max_nb;
  ViewDesc views[PMF_MAXV];
  int64_t val_off[PMF_MAXV];
};

// out[e] = sum over p = 0 .. n_parts-1, in that order, of part[p * stride + e]
__global__ __launch_bounds__(256) void k_sum_parts(const float *__restrict__ part, int64_t stride, int n_parts, float *__restrict__ out, int64_t n) {
  const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= n) return;
  float t = 0.f;
  for (int p = 0; p < n_parts; ++p) t += part[(int64_t)p * stride + e];
  out[e] = t;
}
static int sum_parts(pmf_ctx *c, const float *part, int64_t stride, int n_parts, float *out, int64_t n) {
  if (!out || n <= 0) return 0;
  k_sum_parts<<<(unsigned)((n + 255) / 256), 256, 0, c->stream>>>(part, stride, n_parts, out, n);
  HIPCHK(hipGetLastError());
  return 0;
}

// One wave per workgroup: thread = column (64 consecutive columns), sequential over a chunk of rows.  The column of Y
// lives in registers (32*KB floats), four rows of X at a time are staged in LDS and read back as broadcast 16-B reads
// (a single wave needs no barrier: LDS operations of one wave complete in order), per-(batch, column) sums in LDS
// (no contention), one plain store per (batch, column) per workgroup at the end (private partials, see LayerGradArgs).
template <int KB>
__global__ __launch_bounds__(64) void k_layer_grad(const LayerGradArgs a) {
  constexpr int Kp = 32 * KB;
  constexpr int RG = 4;   // rows per group
  extern __shared__ __attribute__((aligned(16))) char smem_lg[];
  float *xs = reinterpret_cast<float *>(smem_lg);  // [RG][Kp] current rows of X (broadcast)
  float *bacc = xs + RG * Kp;                      // [2][max_nb][64] per-(batch,column) sums
  __shared__ double sh[4];
  const int tid = threadIdx.x;
  const int64_t j = blockIdx.x * 64 + tid;
  const bool col_ok = j < a.N;
  const int64_t jc = col_ok ? j : a.N - 1;
  const int64_t r0 = (int64_t)blockIdx.y * a.rows_per_block;
  int64_t r1 = r0 + a.rows_per_block;
  if (r1 > a.M) r1 = a.M;
  const float4 cp = a.colp[jc];
  const int meta = __float_as_int(cp.w);
  const int kind = col_ok ? (meta & 3) : 3;
  const int v = (meta >> 2) - 1;
  ViewDesc vd = a.views[v >= 0 ? v : 0];
  for (int e = tid; e < 2 * a.max_nb * 64; e += 64) bacc[e] = 0.f;
  float smu = 0.f, sls = 0.f;
  double lacc = 0.0;
  float4 yr[Kp / 4];
  {
    const float4 *y4 = reinterpret_cast<const float4 *>(a.Y + jc * Kp);
#pragma unroll
    for (int q = 0; q < Kp / 4; ++q) yr[q] = y4[q];
  }
  for (int64_t ib = r0; ib < r1; ib += RG) {
    // stage RG rows of X (contiguous in memory: X is Kp x M column-major); rows past the chunk are clamped
    __builtin_amdgcn_wave_barrier();
    for (int e4 = tid; e4 < RG * Kp / 4; e4 += 64) {
      const int rr = (e4 * 4) / Kp;
      const int64_t i = ib + rr < r1 ? ib + rr : r1 - 1;
      reinterpret_cast<float4 *>(xs)[e4] = reinterpret_cast<const float4 *>(a.X + i * Kp)[(e4 * 4 % Kp) / 4];
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int rr = 0; rr < RG; ++rr) {
      const int64_t i = ib + rr;
      if (i >= r1) break;
      const float4 *x4 = reinterpret_cast<const float4 *>(xs + rr * Kp);
      float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
      for (int q = 0; q < Kp / 4; ++q) {
        const float4 xv = x4[q];
        acc0 = fmaf(xv.x, yr[q].x, acc0);
        acc1 = fmaf(xv.y, yr[q].y, acc1);
        acc0 = fmaf(xv.z, yr[q].z, acc0);
        acc1 = fmaf(xv.w, yr[q].w, acc1);
      }
      const float acc = acc0 + acc1;
      float dl = 1.f, th = 0.f;
      int b = -1;
      if (v >= 0) {
        b = a.bor[(int64_t)v * a.M + i];
        if (b >= 0) {
          const float2 dt = a.btab[vd.tab_off + (jc - vd.c0) * vd.nb + b];
          dl = dt.x;
          th = dt.y;
        }
      }
      const float yv = pmf_d_get(a.D, i, jc, a.nRB, a.d_bf16);
      const float z1 = acc * cp.x;
      const float z = fmaf(z1, dl, cp.y + th);
      float l, g;
      if (kind == PMF_NOISE_NORMAL) {
        const float d = z - yv;
        g = cp.z * d;
        l = 0.5f * g * d;
      } else if (kind == PMF_NOISE_BERNOULLI) {
        const float e = __expf(-fabsf(z));
        const float sp = fmaxf(z, 0.f) + __logf(1.f + e);
        const float r = __frcp_rn(1.f + e);
        const float sg = z >= 0.f ? r : e * r;
        l = cp.z * (sp - yv * z);
        g = cp.z * (sg - yv);
      } else {
        const float e = __expf(z);
        l = cp.z * (e - yv * z);
        g = cp.z * (e - yv);
      }
      const bool ok = (kind != 3) && (fabsf(yv) <= 3.402823466e38f);
      if (!ok) { l = 0.f; g = 0.f; }
      lacc += (double)l;
      smu += g;
      sls += g * dl;
      if (b >= 0) {
        bacc[b * 64 + tid] += g;
        bacc[(a.max_nb + b) * 64 + tid] += g * z1;
      }
    }
  }
  if (col_ok) {
    float *pp = a.part + (int64_t)blockIdx.y * a.part_stride;
    if (a.g_mu) pp[j] = smu;
    if (a.g_logsigma) pp[a.N + j] = sls * cp.x;
    if (v >= 0) {
      for (int b = 0; b < vd.nb; ++b) {
        const int64_t e = a.val_off[v] + (j - vd.c0) * vd.nb + b;
        if (a.g_theta) pp[2 * a.N + e] = bacc[b * 64 + tid];
        if (a.g_logdelta) {
          const float dlt = a.btab[vd.tab_off + (j - vd.c0) * vd.nb + b].x;
          pp[2 * a.N + a.nbt + e] = bacc[(a.max_nb + b) * 64 + tid] * dlt;
        }
      }
    }
  }
  const double s = block_reduce_sum(lacc, sh);
  if (tid == 0 && a.loss_partial) a.loss_partial[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = s;
}

// Called a handful of times per fit, not per epoch.
// ------------------------------------------------------------------------------------------------
struct StatsArgs {
  const void *D;
  int d_bf16;
  const float *X, *Y;
  const float4 *colp;
  const int32_t *bor;
  const float2 *btab;
  float *col_n, *col_sum, *col_sumsq, *col_sqerr, *col_ssqg;  // N each: block row 0's vector of the PRIVATE partials -- block row y
  float *b_n, *b_sqerr;                                       // writes at + y * part_stride; flat like theta (may be null)
  int64_t part_stride;                                        // k_sum_parts adds the block rows in fixed order (no float atomics)
  int64_t M, N, nRB;
  int Kp, K, rows_per_block, max_nb, use_factors;
  ViewDesc views[PMF_MAXV];
  int64_t val_off[PMF_MAXV];
};

// Same structure as k_layer_grad<KB>: one wave per workgroup, thread = column, Y column in registers, four rows of X
// at a time through LDS (broadcast 16-B reads, no barrier), per-(batch, column) sums in LDS.
template <int KB>
__global__ __launch_bounds__(64) void k_stats(const StatsArgs a) {
  constexpr int Kp = 32 * KB;
  constexpr int RG = 4;
  extern __shared__ __attribute__((aligned(16))) char smem_st[];
  float *xs = reinterpret_cast<float *>(smem_st);   // [RG][Kp]
  float *bacc = xs + RG * Kp;                        // [2][max_nb][64]
  const int tid = threadIdx.x;
  const int64_t j = blockIdx.x * 64 + tid;
  const bool col_ok = j < a.N;
  const int64_t jc = col_ok ? j : a.N - 1;
  const int64_t r0 = (int64_t)blockIdx.y * a.rows_per_block;
  int64_t r1 = r0 + a.rows_per_block;
  if (r1 > a.M) r1 = a.M;
  const float4 cp = a.colp[jc];
  const int meta = __float_as_int(cp.w);
  const int kind = meta & 3;
  const int v = (meta >> 2) - 1;
  const ViewDesc vd = a.views[v >= 0 ? v : 0];
  for (int e = tid; e < 2 * a.max_nb * 64; e += 64) bacc[e] = 0.f;
  float sn = 0.f, s1 = 0.f, s2 = 0.f, se = 0.f, sg = 0.f;
  float4 yr[Kp / 4];
  {
    const float4 *y4 = reinterpret_cast<const float4 *>(a.Y + jc * Kp);
#pragma unroll
    for (int q = 0; q < Kp / 4; ++q) yr[q] = y4[q];
  }
  for (int64_t ib = r0; ib < r1; ib += RG) {
    if (a.use_factors) {
      __builtin_amdgcn_wave_barrier();
      for (int e4 = tid; e4 < RG * Kp / 4; e4 += 64) {
        const int rr = (e4 * 4) / Kp;
        const int64_t i = ib + rr < r1 ? ib + rr : r1 - 1;
        reinterpret_cast<float4 *>(xs)[e4] = reinterpret_cast<const float4 *>(a.X + i * Kp)[(e4 * 4 % Kp) / 4];
      }
      __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int rr = 0; rr < RG; ++rr) {
      const int64_t i = ib + rr;
      if (i >= r1) break;
      float acc = 0.f;
      if (a.use_factors) {
        const float4 *x4 = reinterpret_cast<const float4 *>(xs + rr * Kp);
        float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
        for (int q = 0; q < Kp / 4; ++q) {
          const float4 xv = x4[q];
          acc0 = fmaf(xv.x, yr[q].x, acc0);
          acc1 = fmaf(xv.y, yr[q].y, acc1);
          acc0 = fmaf(xv.z, yr[q].z, acc0);
          acc1 = fmaf(xv.w, yr[q].w, acc1);
        }
        acc = acc0 + acc1;
      }
      float dl = 1.f, th = 0.f;
      int b = -1;
      if (v >= 0) {
        b = a.bor[(int64_t)v * a.M + i];
        if (b >= 0) {
          const float2 dt = a.btab[vd.tab_off + (jc - vd.c0) * vd.nb + b];
          dl = dt.x;
          th = dt.y;
        }
      }
      const float yv = pmf_d_get(a.D, i, jc, a.nRB, a.d_bf16);
      if (!(fabsf(yv) <= 3.402823466e38f)) continue;
      const float z = fmaf(acc * cp.x, dl, cp.y + th);
      float pred, g;
      if (kind == PMF_NOISE_NORMAL) { pred = z; g = cp.z * (z - yv); }
      else if (kind == PMF_NOISE_BERNOULLI) { pred = 1.f / (1.f + __expf(-z)); g = cp.z * (pred - yv); }
      else { pred = __expf(z); g = cp.z * (pred - yv); }
      const float r = pred - yv;
      sn += 1.f; s1 += yv; s2 += yv * yv; se += r * r; sg += g * g;
      if (b >= 0) {
        bacc[b * 64 + tid] += 1.f;
        bacc[(a.max_nb + b) * 64 + tid] += r * r;
      }
    }
  }
  if (col_ok) {
    const int64_t po = (int64_t)blockIdx.y * a.part_stride;
    if (a.col_n) a.col_n[po + j] = sn;
    if (a.col_sum) a.col_sum[po + j] = s1;
    if (a.col_sumsq) a.col_sumsq[po + j] = s2;
    if (a.col_sqerr) a.col_sqerr[po + j] = se;
    if (a.col_ssqg) a.col_ssqg[po + j] = sg;
    if (v >= 0 && a.b_n) {
      for (int b = 0; b < vd.nb; ++b) {
        const int64_t e = a.val_off[v] + (j - vd.c0) * vd.nb + b;
        a.b_n[po + e] = bacc[b * 64 + tid];
        a.b_sqerr[po + e] = bacc[(a.max_nb + b) * 64 + tid];
      }
    }
  }
}

// Z = layers(X'Y) materialised (MF.forward, simulate_params.jl:247); also used by pmf_synth_data.
struct ForwardArgs {
  const float *X, *Y;
  const float4 *colp;
  const int32_t *bor;
  const float2 *btab;
  float *Z;
  int64_t M, N, nRB;   // nRB > 0: write Z in the tile-major data layout (synthetic data; z_bf16: as bf16), else column-major
  int z_bf16;
  int Kp, K;
  int synth;
  uint64_t seed;
  float noise, frac_nan;
  ViewDesc views[PMF_MAXV];
};
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__global__ __launch_bounds__(256) void k_forward(const ForwardArgs a) {
  // block: 256 consecutive rows of one column -> coalesced stores along i
  const int64_t j = blockIdx.x;
  const int64_t i = blockIdx.y * 256ll + threadIdx.x;
  __shared__ float ys[128];
  for (int k = threadIdx.x; k < a.Kp; k += 256) ys[k] = a.Y[j * a.Kp + k];
  __syncthreads();
  if (i >= a.M) return;
  const float *x = a.X + i * a.Kp;
  float acc = 0.f;
  for (int k = 0; k < a.K; ++k) acc = fmaf(x[k], ys[k], acc);
  const float4 cp = a.colp[j];
  const int meta = __float_as_int(cp.w);
  const int v = (meta >> 2) - 1;
  float dl = 1.f, th = 0.f;
  if (v >= 0) {
    const ViewDesc vd = a.views[v];
    const int b = a.bor[(int64_t)v * a.M + i];
    if (b >= 0) {
      const float2 dt = a.btab[vd.tab_off + (j - vd.c0) * vd.nb + b];
      dl = dt.x;
      th = dt.y;
    }
  }
  float z = fmaf(acc * cp.x, dl, cp.y + th);
  if (a.synth) {
    const uint64_t ctr = (uint64_t)(j * a.M + i);
    const uint64_t r1 = splitmix64(a.seed ^ (ctr * 2ull));
    const uint64_t r2 = splitmix64(a.seed ^ (ctr * 2ull + 1ull));
    const float u1 = ((r1 >> 40) + 1.0f) * (1.0f / 16777217.0f);
    const float u2 = (r1 & 0xFFFFFFull) * (1.0f / 16777216.0f);
    const float nrm = sqrtf(-2.f * logf(u1)) * cosf(6.28318530718f * u2);
    const int kind = meta & 3;
    if (kind == PMF_NOISE_NORMAL) z += a.noise * nrm;
    else if (kind == PMF_NOISE_BERNOULLI) z = (z + a.noise * nrm) > 0.f ? 1.f : 0.f;
    else z = floorf(__expf(fminf(z, 10.f)));
    const float u3 = (r2 >> 40) * (1.0f / 16777216.0f);
    if (u3 < a.frac_nan) z = __int_as_float(0x7fc00000);
  }
  if (a.nRB > 0 && a.z_bf16) reinterpret_cast<__bf16 *>(a.Z)[pmf_d_off16(i, j, a.nRB)] = (__bf16)z;
  else a.Z[a.nRB > 0 ? pmf_d_off(i, j, a.nRB) : j * a.M + i] = z;
}

// ------------------------------------------------------------------------------------------------
// host side helpers
// ------------------------------------------------------------------------------------------------
static inline int nblocks(int64_t n, int bs) { return (int)((n + bs - 1) / bs); }

static int upload_padded(pmf_ctx *c, float *dst, const float *src, int64_t n, float padval) {
  // src: host K x n ; dst: device Kp x n
  if (c->K == c->Kp) {
    HIPCHK(hipMemcpyAsync(dst, src, sizeof(float) * (size_t)(n * c->K), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
  }
  PMFCHK(ensure_scratch(c, sizeof(float) * (size_t)(n * c->K)));
  HIPCHK(hipMemcpyAsync(c->scratch, src, sizeof(float) * (size_t)(n * c->K), hipMemcpyHostToDevice, c->stream));
  k_pad_copy<<<nblocks(n * c->Kp, 256), 256, 0, c->stream>>>(dst, (const float *)c->scratch, c->Kp, c->K, n, padval);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}
static int download_padded(pmf_ctx *c, float *dst_host, const float *src_dev, int64_t n) {
  HIPCHK(hipMemcpy2DAsync(dst_host, sizeof(float) * c->K, src_dev, sizeof(float) * c->Kp, sizeof(float) * c->K,
                          (size_t)n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}
template <typename T>
static int upload_vec(pmf_ctx *c, T *dst, const T *src, size_t n) {
  if (n == 0) return 0;
  HIPCHK(hipMemcpyAsync(dst, src, sizeof(T) * n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

static int set_K(pmf_ctx *c, int K) {
  if (K <= 0 || K > 128) return pmf_fail("K=%d unsupported (1..128)", K);
  if (c->M <= 0 || c->N <= 0) return pmf_fail("pmf_set_data must be called before pmf_set_factors");
  if (K == c->K && c->P[0].n == (int64_t)c->Kp * c->M && c->P[1].n == (int64_t)c->Kp * c->N) return 0;
  c->K = K;
  c->KB = (K + 31) / 32;
  c->Kp = 32 * c->KB;
  PMFCHK(param_alloc(c->P[0], (int64_t)c->Kp * c->M));
  PMFCHK(param_alloc(c->P[1], (int64_t)c->Kp * c->N));
  dev_free(&c->ard_alpha);
  dev_free(&c->ard_beta);
  c->has_ard = false;
  c->state_init = false;
  return 0;
}

extern "C" int pmf_device_count(int *n) {
  if (!n) return pmf_fail("null output");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  *n = ndev;
  return 0;
}

extern "C" int pmf_create(int device, pmf_ctx **out) {
  if (!out) return pmf_fail("null out");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return pmf_fail("device %d out of range (%d visible)", device, ndev);
  HIPCHK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
    return pmf_fail("libpmf_hip is built for gfx950 (MI355X) only; device %d is %s", device, prop.gcnArchName);
  pmf_ctx *c = new pmf_ctx();
  c->device = device;
  c->n_cu = prop.multiProcessorCount;
  HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  c->own_stream = true;
  HIPCHK(hipMalloc((void **)&c->d_loss, sizeof(double) * 8));
  HIPCHK(hipHostMalloc((void **)&c->h_loss, sizeof(double) * 8));
  HIPCHK(hipMalloc((void **)&c->reg_partial, sizeof(double) * 4 * REG_SLOTS));
  PMFCHK(memset_now(c->reg_partial, 0, sizeof(double) * 4 * REG_SLOTS));
  {
    const char *pe = getenv("PMF_PRECISION");   // development / benchmark override of the default (exact f32)
    if (pe && std::string(pe) == "bf16x3") c->precision = PMF_PREC_BF16X3;
  }
  *out = c;
  return 0;
}

extern "C" int pmf_destroy(pmf_ctx *c) {
  if (!c) return 0;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  for (auto &b : c->P) param_free(b);
  if (c->D) (void)hipFree(c->D);
  c->D = nullptr;
  dev_free(&c->tflags);
  dev_free(&c->bor); dev_free(&c->btab); dev_free(&c->btd); c->btd_cap = 0; c->btd_ok = false; dev_free(&c->colview); c->colview_cap = 0; dev_free(&c->LG); c->LG_cap = 0; dev_free(&c->lgrad_part); c->lgrad_part_cap = 0; dev_free(&c->d_views); c->views_dirty = true; dev_free(&c->d_val_view);
  dev_free(&c->colmeta); dev_free(&c->colw); dev_free(&c->colp);
  dev_free(&c->ard_alpha); dev_free(&c->ard_beta);
  comm_release(c);
  for (auto &ps : c->pslots) { dev_free(&ps.pm); dev_free(&ps.row_slot); }
  for (auto &ws : c->splits) { dev_free(&ws.wg_begin); dev_free(&ws.c_off); dev_free(&ws.c_idx); dev_free(&ws.piece_base); dev_free(&ws.d_piece_base_abs); }
  dev_free(&c->gx_part); dev_free(&c->gx_off); dev_free(&c->gx_idx);
  if (c->ev_host) (void)hipEventDestroy(c->ev_host);
  dev_free(&c->gy_slabs); dev_free(&c->xsb); dev_free(&c->ysb); dev_free(&c->sb8_scale); dev_free(&c->sb8_max); dev_free(&c->loss_partial); dev_free(&c->reg_partial); dev_free(&c->d_loss);
  if (c->h_loss) (void)hipHostFree(c->h_loss);
  if (c->scratch) (void)hipFree(c->scratch);
  for (auto &e : c->ev_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return 0;
}

extern "C" int pmf_set_precision(pmf_ctx *c, int mode) {
  if (!c) return pmf_fail("null context");
  if (mode != PMF_PREC_F32 && mode != PMF_PREC_BF16X3) return pmf_fail("unknown precision mode %d", mode);
  c->precision = mode;
  return 0;
}

extern "C" int pmf_get_precision(pmf_ctx *c, int *mode, int64_t *split_launches) {
  if (!c) return pmf_fail("null context");
  if (mode) *mode = c->precision;
  if (split_launches) *split_launches = c->sb_launches;
  return 0;
}

extern "C" int pmf_set_stream(pmf_ctx *c, void *s) {
  PMFCHK(ctx_bind(c));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (s == nullptr) {
    if (!c->own_stream) {
      HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
      c->own_stream = true;
    }
    return 0;
  }
  if (c->own_stream && c->stream) HIPCHK(hipStreamDestroy(c->stream));
  c->stream = (hipStream_t)s;
  c->own_stream = false;
  return 0;
}
extern "C" int pmf_synchronize(pmf_ctx *c) {
  PMFCHK(ctx_bind(c));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

static int data_shape_changed(pmf_ctx *c, int64_t M, int64_t N) {
  if (M <= 0 || N <= 0) return pmf_fail("empty data matrix (%lld x %lld)", (long long)M, (long long)N);
  if (M == c->M && N == c->N) return 0;
  c->M = M;
  c->N = N;
  for (auto &b : c->P) param_free(b);
  c->K = c->Kp = c->KB = 0;
  PMFCHK(param_alloc(c->P[2], N));
  PMFCHK(param_alloc(c->P[3], N));
  c->n_bv = 0;
  c->h_bor.clear();
  c->views_serial++;
  c->views.clear();
  c->views_dirty = true;
  c->val_off.clear();
  c->bvb_off.clear();
  dev_free(&c->bor); dev_free(&c->btab); dev_free(&c->btd); c->btd_cap = 0; c->btd_ok = false; dev_free(&c->colview); c->colview_cap = 0; dev_free(&c->LG); c->LG_cap = 0; dev_free(&c->lgrad_part); c->lgrad_part_cap = 0; dev_free(&c->d_views); c->views_dirty = true; dev_free(&c->d_val_view);
  PMFCHK(dev_alloc(&c->colmeta, (size_t)N));
  PMFCHK(dev_alloc(&c->colw, (size_t)N));
  PMFCHK(dev_alloc(&c->colp, (size_t)N));
  k_fill<<<nblocks(N, 256), 256, 0, c->stream>>>(c->colw, N, 1.f);
  HIPCHK(hipGetLastError());
  c->mixed = false;
  c->prepared = false;
  c->state_init = false;
  return 0;
}

// The device copy of D is tile-major (pmf_d_off): nRB = ceil(M/32) row blocks x ceil(N/64)*2 column blocks of
// 32 x 32 floats, NaN-filled outside the matrix.
static int alloc_tiled_D(pmf_ctx *c, int64_t M, int64_t N, int store) {
  const int64_t Npad = (N + PMF_DPAD - 1) / PMF_DPAD * PMF_DPAD;
  const int64_t nRB = (M + 31) / 32;
  const int64_t nfl = nRB * 32 * Npad;
  const size_t esz = store == PMF_STORE_BF16 ? 2 : 4;
  if (!(c->D && c->D_M == M && c->D_Npad == Npad && c->store == store)) {
    if (c->D) (void)hipFree(c->D);
    c->D = nullptr;
    HIPCHK(hipMalloc(&c->D, esz * (size_t)nfl));
    c->own_D = true;
    c->D_M = M;
    c->D_Npad = Npad;
    c->nRB = nRB;
  }
  c->tflags_valid = false;
  c->store = store;
  if (store == PMF_STORE_BF16) k_fill16<<<(int)std::min<int64_t>(nblocks(nfl, 256), 65536), 256, 0, c->stream>>>((uint16_t *)c->D, nfl, (uint16_t)0x7fc0);
  else k_fill<<<(int)std::min<int64_t>(nblocks(nfl, 256), 65536), 256, 0, c->stream>>>((float *)c->D, nfl, __builtin_nanf(""));
  HIPCHK(hipGetLastError());
  return 0;
}

static int tile_from_device(pmf_ctx *c, const float *src, int64_t col0, int64_t ncols) {
  const int64_t n = c->M * ncols;
  c->tflags_valid = false;
  k_tile_D<<<nblocks(n, 256), 256, 0, c->stream>>>(src, c->M, col0, ncols, c->D, c->nRB, c->store == PMF_STORE_BF16);
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int pmf_set_data(pmf_ctx *c, const float *D, int64_t M, int64_t N, int store) {
  PMFCHK(ctx_bind(c));
  if (!D) return pmf_fail("null data pointer");
  if (store != PMF_STORE_F32 && store != PMF_STORE_BF16) return pmf_fail("unknown storage type %d", store);
  PMFCHK(data_shape_changed(c, M, N));
  PMFCHK(alloc_tiled_D(c, M, N, store));
  // upload in column chunks of <= 256 MiB through a staging buffer, tiling each chunk on the device
  const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(N, (64ll << 20) / std::max<int64_t>(M, 1)));
  PMFCHK(ensure_scratch(c, sizeof(float) * (size_t)(M * chunk)));
  for (int64_t c0 = 0; c0 < N; c0 += chunk) {
    const int64_t nc = std::min(chunk, N - c0);
    HIPCHK(hipMemcpyAsync(c->scratch, D + c0 * M, sizeof(float) * (size_t)(M * nc), hipMemcpyHostToDevice, c->stream));
    PMFCHK(tile_from_device(c, (const float *)c->scratch, c0, nc));
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  return 0;
}
extern "C" int pmf_set_data_device(pmf_ctx *c, const void *D, int64_t M, int64_t N, int store) {
  PMFCHK(ctx_bind(c));
  if (store != PMF_STORE_F32 && store != PMF_STORE_BF16) return pmf_fail("unknown storage type %d", store);
  PMFCHK(data_shape_changed(c, M, N));
  PMFCHK(alloc_tiled_D(c, M, N, store));      // D == NULL: left all-NaN for pmf_synth_data
  if (D) PMFCHK(tile_from_device(c, (const float *)D, 0, N));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int pmf_set_factors(pmf_ctx *c, const float *X, const float *Y, int K) {
  PMFCHK(ctx_bind(c));
  if (!X || !Y) return pmf_fail("null factor pointer");
  PMFCHK(set_K(c, K));
  PMFCHK(upload_padded(c, c->P[0].p, X, c->M, 0.f));
  PMFCHK(upload_padded(c, c->P[1].p, Y, c->N, 0.f));
  return 0;
}
extern "C" int pmf_set_X(pmf_ctx *c, const float *X, int K) {
  PMFCHK(ctx_bind(c));
  if (K != c->K) return pmf_fail("pmf_set_X: K=%d differs from the current K=%d (use pmf_set_factors)", K, c->K);
  return upload_padded(c, c->P[0].p, X, c->M, 0.f);
}
extern "C" int pmf_set_Y(pmf_ctx *c, const float *Y, int K) {
  PMFCHK(ctx_bind(c));
  if (K != c->K) return pmf_fail("pmf_set_Y: K=%d differs from the current K=%d (use pmf_set_factors)", K, c->K);
  return upload_padded(c, c->P[1].p, Y, c->N, 0.f);
}
extern "C" int pmf_get_factors(pmf_ctx *c, float *X, float *Y) {
  PMFCHK(ctx_bind(c));
  if (c->K == 0) return pmf_fail("factors not set");
  if (X) PMFCHK(download_padded(c, X, c->P[0].p, c->M));
  if (Y) PMFCHK(download_padded(c, Y, c->P[1].p, c->N));
  return 0;
}

extern "C" int pmf_set_col_params(pmf_ctx *c, const float *logsigma, const float *mu) {
  PMFCHK(ctx_bind(c));
  if (c->N == 0) return pmf_fail("data not set");
  if (logsigma) PMFCHK(upload_vec(c, c->P[2].p, logsigma, (size_t)c->N));
  if (mu) PMFCHK(upload_vec(c, c->P[3].p, mu, (size_t)c->N));
  c->prepared = false;
  return 0;
}
extern "C" int pmf_get_col_params(pmf_ctx *c, float *logsigma, float *mu) {
  PMFCHK(ctx_bind(c));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (logsigma) HIPCHK(hipMemcpy(logsigma, c->P[2].p, sizeof(float) * (size_t)c->N, hipMemcpyDeviceToHost));
  if (mu) HIPCHK(hipMemcpy(mu, c->P[3].p, sizeof(float) * (size_t)c->N, hipMemcpyDeviceToHost));
  return 0;
}

// ---- batch views -------------------------------------------------------------------------------
static int rebuild_colmeta_views(pmf_ctx *c) {
  // refresh the view id bits of colmeta from the current views (kind bits are kept)
  std::vector<int32_t> meta((size_t)c->N);
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpy(meta.data(), c->colmeta, sizeof(int32_t) * (size_t)c->N, hipMemcpyDeviceToHost));
  for (auto &m : meta) m &= 3;
  for (int v = 0; v < c->n_bv; ++v) {
    const ViewDesc &vd = c->views[v];
    if (vd.nb <= 0) continue;
    for (int64_t j = vd.c0; j < vd.c1; ++j) meta[(size_t)j] |= (v + 1) << 2;
  }
  HIPCHK(hipMemcpy(c->colmeta, meta.data(), sizeof(int32_t) * (size_t)c->N, hipMemcpyHostToDevice));
  c->prepared = false;
  return 0;
}

extern "C" int pmf_set_n_batch_views(pmf_ctx *c, int n) {
  PMFCHK(ctx_bind(c));
  if (n < 0 || n > PMF_MAXV) return pmf_fail("n_batch_views=%d out of range (0..%d)", n, PMF_MAXV);
  if (c->N == 0) return pmf_fail("data not set");
  // Same number of views as before: keep the views, their values and -- above all -- the optimizer state.  Both hosts
  // re-marshal the model before every MF.fit! call, and mf_fit_adapt_lr! (src/fit.jl:55-69) resumes with the SAME
  // AdaGrad object after halving eta: its accumulators must survive.  pmf_set_batch_view replaces what changed.
  if (n == c->n_bv && (int)c->views.size() == n && (n == 0 || c->bor)) return 0;
  c->n_bv = n;
  c->h_bor.assign((size_t)n, std::vector<int32_t>());
  c->views_serial++;
  c->views.assign((size_t)n, ViewDesc{0, 0, 0, 0, 0});
  c->views_dirty = true;
  c->val_off.assign((size_t)n + 1, 0);
  c->bvb_off.assign((size_t)n + 1, 0);
  param_free(c->P[4]);
  param_free(c->P[5]);
  dev_free(&c->btab);
  dev_free(&c->d_val_view);
  PMFCHK(dev_alloc(&c->bor, (size_t)std::max<int64_t>(1, (int64_t)n * c->M)));
  if (n > 0) PMFCHK(memset_now(c->bor, 0xFF, sizeof(int32_t) * (size_t)((int64_t)n * c->M)));
  PMFCHK(rebuild_colmeta_views(c));
  c->state_init = false;
  return 0;
}

extern "C" int pmf_set_batch_view(pmf_ctx *c, int v, int64_t s1, int64_t e1, int nb, const int32_t *batch_of_row,
                                  const float *logdelta, const float *theta) {
  PMFCHK(ctx_bind(c));
  if (v < 0 || v >= c->n_bv) return pmf_fail("batch view %d out of range (n=%d)", v, c->n_bv);
  if (s1 < 1 || e1 > c->N || s1 > e1) return pmf_fail("batch view %d: bad column range %lld:%lld", v, (long long)s1, (long long)e1);
  if (nb <= 0) return pmf_fail("batch view %d: nb=%d", v, nb);
  if (v > 0 && c->views[v - 1].nb == 0) return pmf_fail("batch views must be set in order (view %d is unset)", v - 1);
  if (v > 0 && s1 - 1 < c->views[v - 1].c1) return pmf_fail("batch view %d overlaps / precedes view %d", v, v - 1);
  const int64_t Nv = e1 - s1 + 1;
  for (int64_t i = 0; i < c->M; ++i)
    if (batch_of_row[i] < -1 || batch_of_row[i] >= nb) return pmf_fail("batch view %d: batch_of_row[%lld]=%d out of range", v, (long long)i, batch_of_row[i]);
  const bool same_shape = c->views[v].nb == nb && c->views[v].c0 == s1 - 1 && c->views[v].c1 == e1 && c->P[4].n > 0 &&
                          c->val_off[v + 1] - c->val_off[v] == Nv * nb;
  c->views[v].c0 = s1 - 1;
  c->views[v].c1 = e1;
  c->views[v].nb = nb;
  c->views_dirty = true;
  HIPCHK(hipMemcpy(c->bor + (int64_t)v * c->M, batch_of_row, sizeof(int32_t) * (size_t)c->M, hipMemcpyHostToDevice));
  if ((int)c->h_bor.size() < c->n_bv) c->h_bor.resize((size_t)c->n_bv);
  if (c->h_bor[(size_t)v].size() != (size_t)c->M || memcmp(c->h_bor[(size_t)v].data(), batch_of_row, sizeof(int32_t) * (size_t)c->M) != 0) {
    c->h_bor[(size_t)v].assign(batch_of_row, batch_of_row + c->M);
    c->views_serial++;
  }
  if (!same_shape) {
    // (re)compute offsets for views >= v and reallocate the flat arrays once the last view is known
    for (int u = v; u < c->n_bv; ++u) {
      const int64_t sz = c->views[u].nb > 0 ? (c->views[u].c1 - c->views[u].c0) * c->views[u].nb : 0;
      c->val_off[u + 1] = c->val_off[u] + sz;
      c->bvb_off[u + 1] = c->bvb_off[u] + c->views[u].nb;
      c->views[u].tab_off = c->val_off[u];
    }
    const int64_t tot = c->val_off[c->n_bv];
    // keep already uploaded values of earlier views
    std::vector<float> old_ld, old_th;
    const int64_t keep = c->val_off[v];
    if (c->P[4].n >= keep && keep > 0) {
      old_ld.resize((size_t)keep);
      old_th.resize((size_t)keep);
      HIPCHK(hipMemcpy(old_ld.data(), c->P[4].p, sizeof(float) * (size_t)keep, hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(old_th.data(), c->P[5].p, sizeof(float) * (size_t)keep, hipMemcpyDeviceToHost));
    }
    PMFCHK(param_alloc(c->P[4], tot));
    PMFCHK(param_alloc(c->P[5], tot));
    if (!old_ld.empty()) {
      HIPCHK(hipMemcpy(c->P[4].p, old_ld.data(), sizeof(float) * old_ld.size(), hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(c->P[5].p, old_th.data(), sizeof(float) * old_th.size(), hipMemcpyHostToDevice));
    }
    PMFCHK(dev_alloc(&c->btab, (size_t)std::max<int64_t>(1, tot)));
    std::vector<int32_t> vv((size_t)std::max<int64_t>(1, tot), 0);
    for (int u = 0; u < c->n_bv; ++u)
      for (int64_t e = c->val_off[u]; e < c->val_off[u + 1]; ++e) vv[(size_t)e] = u;
    PMFCHK(dev_alloc(&c->d_val_view, vv.size()));
    HIPCHK(hipMemcpy(c->d_val_view, vv.data(), sizeof(int32_t) * vv.size(), hipMemcpyHostToDevice));
    c->state_init = false;
  }
  const int64_t off = c->val_off[v];
  if (logdelta) HIPCHK(hipMemcpy(c->P[4].p + off, logdelta, sizeof(float) * (size_t)(Nv * nb), hipMemcpyHostToDevice));
  if (theta) HIPCHK(hipMemcpy(c->P[5].p + off, theta, sizeof(float) * (size_t)(Nv * nb), hipMemcpyHostToDevice));
  PMFCHK(rebuild_colmeta_views(c));
  return 0;
}

extern "C" int pmf_get_batch_view(pmf_ctx *c, int v, float *logdelta, float *theta) {
  PMFCHK(ctx_bind(c));
  if (v < 0 || v >= c->n_bv) return pmf_fail("batch view %d out of range (n=%d)", v, c->n_bv);
  HIPCHK(hipStreamSynchronize(c->stream));
  const int64_t off = c->val_off[v], n = c->val_off[v + 1] - off;
  if (logdelta) HIPCHK(hipMemcpy(logdelta, c->P[4].p + off, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost));
  if (theta) HIPCHK(hipMemcpy(theta, c->P[5].p + off, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost));
  return 0;
}

// ---- noise model -------------------------------------------------------------------------------
extern "C" int pmf_set_noise(pmf_ctx *c, int n_ranges, const int64_t *s1, const int64_t *e1, const int32_t *kinds,
                             const float *weights) {
  PMFCHK(ctx_bind(c));
  if (c->N == 0) return pmf_fail("data not set");
  std::vector<int32_t> meta((size_t)c->N, 3);
  bool mixed = false;
  for (int r = 0; r < n_ranges; ++r) {
    if (s1[r] < 1 || e1[r] > c->N || s1[r] > e1[r]) return pmf_fail("noise range %d: bad columns %lld:%lld", r, (long long)s1[r], (long long)e1[r]);
    if (kinds[r] < 0 || kinds[r] > 2) return pmf_fail("noise range %d: unsupported kind %d (normal=0, bernoulli=1, poisson=2)", r, kinds[r]);
    if (kinds[r] != PMF_NOISE_NORMAL) mixed = true;
    for (int64_t j = s1[r] - 1; j < e1[r]; ++j) meta[(size_t)j] = kinds[r];
  }
  for (int64_t j = 0; j < c->N; ++j)
    if (meta[(size_t)j] == 3) return pmf_fail("noise model does not cover column %lld", (long long)(j + 1));
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpy(c->colmeta, meta.data(), sizeof(int32_t) * (size_t)c->N, hipMemcpyHostToDevice));
  if (weights) HIPCHK(hipMemcpy(c->colw, weights, sizeof(float) * (size_t)c->N, hipMemcpyHostToDevice));
  c->mixed = mixed;
  c->h_kind.assign(meta.begin(), meta.end());
  c->kind_version++;
  PMFCHK(rebuild_colmeta_views(c));
  return 0;
}

// ---- regularizers ------------------------------------------------------------------------------
static int add_quad_ranges(pmf_ctx *c, int which, int n_groups, const int64_t *s1, const int64_t *e1, const float *w, float p) {
  if (c->K == 0) return pmf_fail("factors must be set before their regularizers");
  ParamBuf &b = c->P[which];
  const int64_t n = which == 0 ? c->M : c->N;
  for (int g = 0; g < n_groups; ++g)
    if (s1[g] < 1 || e1[g] > n || s1[g] > e1[g]) return pmf_fail("regularizer range %d: bad %lld:%lld (n=%lld)", g, (long long)s1[g], (long long)e1[g], (long long)n);
  if (!b.wq) PMFCHK(dev_alloc(&b.wq, (size_t)b.n));
  int64_t *ds = nullptr, *de = nullptr;
  float *dw = nullptr;
  PMFCHK(dev_alloc(&ds, (size_t)n_groups, false));
  PMFCHK(dev_alloc(&de, (size_t)n_groups, false));
  PMFCHK(dev_alloc(&dw, (size_t)n_groups * c->K, false));
  HIPCHK(hipMemcpy(ds, s1, sizeof(int64_t) * n_groups, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(de, e1, sizeof(int64_t) * n_groups, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dw, w, sizeof(float) * (size_t)n_groups * c->K, hipMemcpyHostToDevice));
  k_expand_group<<<nblocks(b.n, 256), 256, 0, c->stream>>>(b.wq, c->Kp, c->K, n, ds, de, dw, n_groups, p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  dev_free(&ds); dev_free(&de); dev_free(&dw);
  return 0;
}
static int add_quad_l2(pmf_ctx *c, int which, const float *w, float p) {
  const int64_t s1 = 1, e1 = which == 0 ? c->M : c->N;
  return add_quad_ranges(c, which, 1, &s1, &e1, w, p);
}
extern "C" int pmf_clear_xreg(pmf_ctx *c) {
  PMFCHK(ctx_bind(c));
  dev_free(&c->P[0].wq);
  return 0;
}
extern "C" int pmf_add_xreg_l2(pmf_ctx *c, const float *w, float p) {
  PMFCHK(ctx_bind(c));
  return add_quad_l2(c, 0, w, p);
}
extern "C" int pmf_add_xreg_group(pmf_ctx *c, int n, const int64_t *s1, const int64_t *e1, const float *w, float p) {
  PMFCHK(ctx_bind(c));
  return add_quad_ranges(c, 0, n, s1, e1, w, p);
}
extern "C" int pmf_clear_yreg(pmf_ctx *c) {
  PMFCHK(ctx_bind(c));
  dev_free(&c->P[1].wq);
  c->has_ard = false;
  return 0;
}
extern "C" int pmf_add_yreg_l2(pmf_ctx *c, const float *w, float p) {
  PMFCHK(ctx_bind(c));
  return add_quad_l2(c, 1, w, p);
}
extern "C" int pmf_add_yreg_group(pmf_ctx *c, int n, const int64_t *s1, const int64_t *e1, const float *w, float p) {
  PMFCHK(ctx_bind(c));
  return add_quad_ranges(c, 1, n, s1, e1, w, p);
}
extern "C" int pmf_add_yreg_ard(pmf_ctx *c, int n_ranges, const int64_t *s1, const int64_t *e1, const float *alpha,
                                const float *beta, float p) {
  PMFCHK(ctx_bind(c));
  if (c->K == 0) return pmf_fail("factors must be set before their regularizers");
  if (c->has_ard) return pmf_fail("only one ARD-type term (ARD or FeatureSetARD) is supported on Y");
  for (int r = 0; r < n_ranges; ++r)
    if (s1[r] < 1 || e1[r] > c->N || s1[r] > e1[r]) return pmf_fail("ARD range %d: bad %lld:%lld", r, (long long)s1[r], (long long)e1[r]);
  if (!c->ard_alpha) PMFCHK(dev_alloc(&c->ard_alpha, (size_t)c->N));
  if (!c->ard_beta) PMFCHK(dev_alloc(&c->ard_beta, (size_t)c->P[1].n));
  int64_t *ds = nullptr, *de = nullptr;
  float *da = nullptr, *db = nullptr;
  PMFCHK(dev_alloc(&ds, (size_t)n_ranges, false));
  PMFCHK(dev_alloc(&de, (size_t)n_ranges, false));
  PMFCHK(dev_alloc(&da, (size_t)n_ranges, false));
  PMFCHK(dev_alloc(&db, (size_t)n_ranges, false));
  HIPCHK(hipMemcpy(ds, s1, sizeof(int64_t) * n_ranges, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(de, e1, sizeof(int64_t) * n_ranges, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(da, alpha, sizeof(float) * n_ranges, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(db, beta, sizeof(float) * n_ranges, hipMemcpyHostToDevice));
  k_expand_ard<<<nblocks(c->P[1].n, 256), 256, 0, c->stream>>>(c->ard_alpha, c->ard_beta, c->Kp, c->N, ds, de, da, db, n_ranges);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  dev_free(&ds); dev_free(&de); dev_free(&da); dev_free(&db);
  c->ard_scale = p;
  c->has_ard = true;
  return 0;
}
extern "C" int pmf_add_yreg_fsard(pmf_ctx *c, const float *alpha, const float *beta, float p) {
  PMFCHK(ctx_bind(c));
  if (c->K == 0) return pmf_fail("factors must be set before their regularizers");
  if (c->has_ard) return pmf_fail("only one ARD-type term (ARD or FeatureSetARD) is supported on Y");
  if (!c->ard_alpha) PMFCHK(dev_alloc(&c->ard_alpha, (size_t)c->N));
  if (!c->ard_beta) PMFCHK(dev_alloc(&c->ard_beta, (size_t)c->P[1].n));
  PMFCHK(upload_vec(c, c->ard_alpha, alpha, (size_t)c->N));
  PMFCHK(upload_padded(c, c->ard_beta, beta, c->N, 1.f));
  c->ard_scale = p;
  c->has_ard = true;
  return 0;
}

extern "C" int pmf_set_layer_regs(pmf_ctx *c, int n_ranges, const int64_t *s1, const int64_t *e1,
                                  const float *w_ls, const float *c_ls, const float *w_mu, const float *c_mu,
                                  const float *w_ld, const float *c_ld, const float *w_th, const float *c_th) {
  PMFCHK(ctx_bind(c));
  if (c->N == 0) return pmf_fail("data not set");
  for (int which = 2; which <= 5; ++which) { dev_free(&c->P[which].wq); dev_free(&c->P[which].cq); }
  if (n_ranges > 0) {
    int64_t *ds = nullptr, *de = nullptr;
    float *dw = nullptr, *dc = nullptr;
    PMFCHK(dev_alloc(&ds, (size_t)n_ranges, false));
    PMFCHK(dev_alloc(&de, (size_t)n_ranges, false));
    PMFCHK(dev_alloc(&dw, (size_t)n_ranges, false));
    PMFCHK(dev_alloc(&dc, (size_t)n_ranges, false));
    HIPCHK(hipMemcpy(ds, s1, sizeof(int64_t) * n_ranges, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(de, e1, sizeof(int64_t) * n_ranges, hipMemcpyHostToDevice));
    for (int t = 0; t < 2; ++t) {
      const float *w = t == 0 ? w_ls : w_mu, *cc = t == 0 ? c_ls : c_mu;
      if (!w || !cc) continue;
      ParamBuf &b = c->P[2 + t];
      PMFCHK(dev_alloc(&b.wq, (size_t)c->N));
      PMFCHK(dev_alloc(&b.cq, (size_t)c->N));
      HIPCHK(hipMemcpy(dw, w, sizeof(float) * n_ranges, hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(dc, cc, sizeof(float) * n_ranges, hipMemcpyHostToDevice));
      k_expand_colparam<<<nblocks(c->N, 256), 256, 0, c->stream>>>(b.wq, b.cq, c->N, ds, de, dw, dc, n_ranges);
      HIPCHK(hipGetLastError());
      HIPCHK(hipStreamSynchronize(c->stream));
    }
    dev_free(&ds); dev_free(&de); dev_free(&dw); dev_free(&dc);
  }
  if (c->n_bv > 0 && c->P[4].n > 0) {
    const int64_t nbt = c->bvb_off[c->n_bv];
    std::vector<int32_t> nbs((size_t)c->n_bv);
    for (int v = 0; v < c->n_bv; ++v) nbs[v] = c->views[v].nb;
    int32_t *dnb = nullptr;
    int64_t *dvo = nullptr, *dbo = nullptr;
    float *dw = nullptr, *dc = nullptr;
    PMFCHK(dev_alloc(&dnb, (size_t)c->n_bv, false));
    PMFCHK(dev_alloc(&dvo, (size_t)c->n_bv + 1, false));
    PMFCHK(dev_alloc(&dbo, (size_t)c->n_bv + 1, false));
    PMFCHK(dev_alloc(&dw, (size_t)nbt, false));
    PMFCHK(dev_alloc(&dc, (size_t)nbt, false));
    HIPCHK(hipMemcpy(dnb, nbs.data(), sizeof(int32_t) * c->n_bv, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dvo, c->val_off.data(), sizeof(int64_t) * (c->n_bv + 1), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dbo, c->bvb_off.data(), sizeof(int64_t) * (c->n_bv + 1), hipMemcpyHostToDevice));
    for (int t = 0; t < 2; ++t) {
      const float *w = t == 0 ? w_ld : w_th, *cc = t == 0 ? c_ld : c_th;
      if (!w || !cc) continue;
      ParamBuf &b = c->P[4 + t];
      PMFCHK(dev_alloc(&b.wq, (size_t)b.n));
      PMFCHK(dev_alloc(&b.cq, (size_t)b.n));
      HIPCHK(hipMemcpy(dw, w, sizeof(float) * (size_t)nbt, hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(dc, cc, sizeof(float) * (size_t)nbt, hipMemcpyHostToDevice));
      k_expand_batchreg<<<nblocks(b.n, 256), 256, 0, c->stream>>>(b.wq, b.cq, b.n, c->d_val_view, dvo, dnb, dbo, dw, dc);
      HIPCHK(hipGetLastError());
      HIPCHK(hipStreamSynchronize(c->stream));
    }
    dev_free(&dnb); dev_free(&dvo); dev_free(&dbo); dev_free(&dw); dev_free(&dc);
  }
  return 0;
}

// ---- optimizer ---------------------------------------------------------------------------------
extern "C" int pmf_set_optimizer(pmf_ctx *c, int kind, float lr, float eps, float b1, float b2) {
  PMFCHK(ctx_bind(c));
  if (kind != PMF_OPT_ADAGRAD && kind != PMF_OPT_ADAM) return pmf_fail("unknown optimizer kind %d", kind);
  c->opt_kind = kind;
  c->lr = lr;
  c->eps = eps;
  c->b1 = b1;
  c->b2 = b2;
  c->state_init = false;
  return 0;
}
extern "C" int pmf_set_lr(pmf_ctx *c, float lr) {
  PMFCHK(ctx_bind(c));
  c->lr = lr;
  return 0;
}
extern "C" int pmf_get_lr(pmf_ctx *c, float *lr) {
  PMFCHK(ctx_bind(c));
  *lr = c->lr;
  return 0;
}
extern "C" int pmf_reset_optimizer_state(pmf_ctx *c) {
  PMFCHK(ctx_bind(c));
  c->state_init = false;
  return 0;
}
static int init_opt_state(pmf_ctx *c) {
  for (int w = 0; w < 6; ++w) {
    ParamBuf &b = c->P[w];
    if (b.n == 0) continue;
    k_fill<<<nblocks(b.n, 256), 256, 0, c->stream>>>(b.acc, b.n, c->opt_kind == PMF_OPT_ADAGRAD ? c->eps : 0.f);
    k_fill<<<nblocks(b.n, 256), 256, 0, c->stream>>>(b.mom, b.n, 0.f);
    HIPCHK(hipGetLastError());
    b.bp1 = c->b1;
    b.bp2 = c->b2;
  }
  c->state_init = true;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// epoch machinery
// ------------------------------------------------------------------------------------------------
static int prepare(pmf_ctx *c) {
  const int64_t nbt = c->n_bv > 0 ? c->val_off[c->n_bv] : 0;
  const int64_t n = std::max(c->N, nbt);
  k_prepare<<<nblocks(n, 256), 256, 0, c->stream>>>(c->P[2].p, c->P[3].p, c->colw, c->colmeta, c->colp, c->N,
                                                    c->P[4].p, c->P[5].p, c->btab, nbt);
  HIPCHK(hipGetLastError());
  c->btd_ok = false;
  if (c->n_bv > 0) {
    if (!c->d_views) PMFCHK(dev_alloc(&c->d_views, (size_t)PMF_MAXV, false));   // (prepare() runs every layer epoch: no re-allocation)
    if (c->views_dirty) {
      HIPCHK(hipMemcpyAsync(c->d_views, c->views.data(), sizeof(ViewDesc) * (size_t)c->n_bv, hipMemcpyHostToDevice, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      c->views_dirty = false;
    }
    int nb_max = 0;
    for (int v = 0; v < c->n_bv; ++v) nb_max = std::max(nb_max, (int)c->views[v].nb);
    int shift = 4;
    while ((1 << shift) < nb_max + 1) ++shift;
    const int64_t Npad = (c->N + 31) / 32 * 32;
    if (nb_max <= 255 && (Npad << shift) < (1ll << 31)) {   // (the kernels index the table with 32 bits)
      c->nbs = 1 << shift;
      if (Npad * c->nbs > c->btd_cap) {
        PMFCHK(dev_alloc(&c->btd, (size_t)(Npad * c->nbs), false));
        c->btd_cap = Npad * c->nbs;
      }
      if (Npad > c->colview_cap) {
        PMFCHK(dev_alloc(&c->colview, (size_t)Npad, false));
        c->colview_cap = Npad;
      }
      DenseBtabArgs da;
      memset(&da, 0, sizeof(da));
      da.colmeta = c->colmeta; da.btab = c->btab; da.btd = c->btd; da.colview = c->colview; da.N = c->N; da.Npad = Npad; da.nbs_shift = shift;
      for (int v = 0; v < c->n_bv; ++v) da.views[v] = c->views[v];
      k_dense_btab<<<nblocks(Npad * c->nbs, 256), 256, 0, c->stream>>>(da);
      HIPCHK(hipGetLastError());
      c->btd_ok = true;
    }
  }
  c->prepared = true;
  return 0;
}

// One workgroup per 32x32 tile of the tile-major D: flag = 1 iff all 1024 entries are finite.  The fused kernel takes
// its packed-math epilogue (no per-entry missing-value mask) on flagged tiles.  Recomputed lazily whenever D changes.
__global__ void k_tile_flags(const void *__restrict__ D, int64_t ntiles, uint32_t *__restrict__ flags, int bf16) {
  const int64_t t = blockIdx.x;
  if (t >= ntiles) return;
  bool ok;
  if (bf16) {   // 256 threads x 8 B; finite <=> the exponent field is not all ones
    const uint2 v = reinterpret_cast<const uint2 *>(reinterpret_cast<const uint16_t *>(D) + t * 1024)[threadIdx.x];
    ok = (v.x & 0x7f80u) != 0x7f80u && (v.x & 0x7f800000u) != 0x7f800000u && (v.y & 0x7f80u) != 0x7f80u && (v.y & 0x7f800000u) != 0x7f800000u;
  } else {
    const float4 v = reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(D) + t * 1024)[threadIdx.x];   // 256 threads x 16 B
    ok = fabsf(v.x) <= 3.402823466e38f && fabsf(v.y) <= 3.402823466e38f && fabsf(v.z) <= 3.402823466e38f && fabsf(v.w) <= 3.402823466e38f;
  }
  const int all_ok = __syncthreads_and(ok ? 1 : 0);
  if (threadIdx.x == 0) flags[t] = all_ok ? 1u : 0u;
}

static int ensure_tile_flags(pmf_ctx *c) {
  if (c->tflags_valid) return 0;
  const int64_t ntiles = c->nRB * (c->D_Npad / 32);
  if (ntiles > c->tflags_cap) {
    PMFCHK(dev_alloc(&c->tflags, (size_t)ntiles, false));
    c->tflags_cap = ntiles;
  }
  if (ntiles > 0x7fffffff) return pmf_fail("too many tiles");
  k_tile_flags<<<(unsigned)ntiles, 256, 0, c->stream>>>(c->D, ntiles, c->tflags, c->store == PMF_STORE_BF16);
  HIPCHK(hipGetLastError());
  c->tflags_valid = true;
  return 0;
}

// gY = sum of the private slabs of the workgroups that visited a column tile, in workgroup order (fixed summation
// order: grad(Y) is bitwise reproducible).  The work sequence (pmf_fused_kernel) is segment-major, then row panel,
// then tile; workgroup g owns [g*T/G, (g+1)*T/G).  Inside segment cs (tiles [cs*S, cs*S + n_rp*tps_cs) of the
// sequence) it touched tile ti iff its range, taken relative to the segment start, contains an index == ti mod tps_cs.
__global__ __launch_bounds__(64) void k_gy_reduce(const float *__restrict__ slabs, int64_t stride,
                                                  const int32_t *__restrict__ c_off, const int32_t *__restrict__ c_idx,
                                                  int Kp, int64_t N, float *__restrict__ gY, int ct0,
                                                  const float4 *__restrict__ colscale) {   // non-null: the slabs hold sums NOT yet scaled by sigma_j (pmf_fused_sb8_kernel)
  // blockIdx.x = column tile of the chunk that starts at tile ct0, blockIdx.y = 256-float slice of its 32 x Kp elements (one float4 per thread).  The
  // workgroups that visited the tile are listed in c_idx[c_off[ct] .. c_off[ct+1]) (built on the host with the work
  // split, compute_work_split); their slabs are summed four at a time so that four independent loads are in flight.
  const int ct = blockIdx.x;   // (chunk-relative: indexes c_off)
  const int64_t e0 = (int64_t)(ct0 + ct) * 32 * Kp;
  const int64_t rem = (int64_t)Kp * N - e0;
  const int nel = rem > 32 * Kp ? 32 * Kp : (int)rem;   // a multiple of Kp, Kp a multiple of 32
  const int q = (blockIdx.y * 64 + threadIdx.x) * 4;
  if (q >= nel) return;
  const float *base = slabs + e0 + q;
  const int c0 = c_off[ct], n = c_off[ct + 1] - c0;
  const int32_t *ci = c_idx + c0;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
  int c = 0;
  for (; c + 4 <= n; c += 4) {   // fixed order: ((s0 + s4 + ...) + (s1 + s5 + ...)) + ... -> bitwise reproducible
    const float4 v0 = *reinterpret_cast<const float4 *>(base + (int64_t)ci[c] * stride);
    const float4 v1 = *reinterpret_cast<const float4 *>(base + (int64_t)ci[c + 1] * stride);
    const float4 v2 = *reinterpret_cast<const float4 *>(base + (int64_t)ci[c + 2] * stride);
    const float4 v3 = *reinterpret_cast<const float4 *>(base + (int64_t)ci[c + 3] * stride);
    a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
    a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
    a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
    a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
  }
  for (; c < n; ++c) {
    const float4 v = *reinterpret_cast<const float4 *>(base + (int64_t)ci[c] * stride);
    a0.x += v.x; a0.y += v.y; a0.z += v.z; a0.w += v.w;
  }
  const float sg = colscale ? colscale[(e0 + q) / Kp].x : 1.f;   // (the four elements belong to one column: Kp is a multiple of 4)
  *reinterpret_cast<float4 *>(gY + e0 + q) =
      make_float4(((a0.x + a1.x) + (a2.x + a3.x)) * sg, ((a0.y + a1.y) + (a2.y + a3.y)) * sg, ((a0.z + a1.z) + (a2.z + a3.z)) * sg,
                  ((a0.w + a1.w) + (a2.w + a3.w)) * sg);
}

// gX = sum of the per-piece partial slabs of a row panel, in work-sequence order (fixed summation order: grad(X) is
// bitwise reproducible).  blockIdx.x = row panel, blockIdx.y = 256-float4 slice of its BM x Kp elements; the slots of
// the panel's pieces are listed in gx_idx[gx_off[rp] .. gx_off[rp+1]) (built on the host with the work split).
__global__ __launch_bounds__(256) void k_gx_reduce(const float *__restrict__ part, int64_t slot_stride,
                                                   const int32_t *__restrict__ gx_off, const int32_t *__restrict__ gx_idx,
                                                   int64_t panel_floats, int64_t total_floats, float *__restrict__ gX) {
  const int rp = blockIdx.x;
  const int64_t q = ((int64_t)blockIdx.y * 256 + threadIdx.x) * 4;
  if (q >= panel_floats) return;
  const int64_t e = (int64_t)rp * panel_floats + q;
  if (e >= total_floats) return;   // rows past M (Kp*M is a multiple of 4: a float4 is all in or all out)
  const int c0 = gx_off[rp], n = gx_off[rp + 1] - c0;
  const int32_t *ci = gx_idx + c0;
  const float *base = part + q;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
  int c = 0;
  for (; c + 2 <= n; c += 2) {   // fixed order: (s0 + s2 + ...) + (s1 + s3 + ...)
    const float4 v0 = *reinterpret_cast<const float4 *>(base + (int64_t)ci[c] * slot_stride);
    const float4 v1 = *reinterpret_cast<const float4 *>(base + (int64_t)ci[c + 1] * slot_stride);
    a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
    a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
  }
  if (c < n) {
    const float4 v = *reinterpret_cast<const float4 *>(base + (int64_t)ci[c] * slot_stride);
    a0.x += v.x; a0.y += v.y; a0.z += v.z; a0.w += v.w;
  }
  *reinterpret_cast<float4 *>(gX + e) = make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w);
}


// Cost-balanced split of the fused kernel's work sequence for one column chunk (tiles [ct0, ct0 + n_ct); sequence =
// segment-major, row panel, tile) into `grid` contiguous ranges.  A tile's estimated cost depends on the noise models of
// its 32 columns (measured on MI355X: a Bernoulli tile costs 1.4x a Gaussian one, Poisson 1.3x); with Gaussian-only
// data this is the even split g*T/G.  Also numbers the pieces (the part of one (segment, row panel) unit inside one
// workgroup's range) in sequence order: piece p of the chunk owns slot p of the chunk's gX partial slabs.
// Cost of a Bernoulli / Poisson tile in sixteenths of a Gaussian tile's, per kernel family and k-block count.  MEASURED: the
// weight that minimises the launch time at 100000 x 50000 with 20 % Bernoulli columns (scripts/kbench_mixed.py ... mixed under
// PMF_W_BERN=w; round 3).  It is not the ratio of the two tiles' times in isolation (1.19-1.58): the workgroups that own the
// Bernoulli columns do VALU work at the clock the OTHER workgroups' matrix work leaves them.  One weight for all kernels
// (22, right for the exact kernel at K = 64) cost the split kernels 7-22 % and the exact kernel at K <= 32 12 %.
// With batch layers EVERY tile pays the table look-ups (w_batch, from the batch-only against the plain launch at the same size),
// which lowers the Bernoulli tiles' relative weight.
static void noise_tile_weights(int KB, bool split, bool batch, int64_t &w_gauss, int64_t &w_bern, int64_t &w_pois) {
  static const int64_t wb_exact[4] = {26, 22, 20, 20}, wb_split[4] = {31, 30, 24, 24};
  static const int64_t we_exact[4] = {6, 6, 5, 4}, we_split[4] = {12, 8, 6, 6};   // (swept on the all-features launch: PMF_W_BATCH)
  w_bern = (split ? wb_split : wb_exact)[KB - 1];
  if (getenv("PMF_W_BERN")) w_bern = std::max(16, atoi(getenv("PMF_W_BERN")));
  w_pois = 16 + ((w_bern - 16) * 5 + 3) / 6;       // (exp only against exp + log + rcp: the exact kernel's 21 against 22)
  if (getenv("PMF_W_POIS")) w_pois = std::max(16, atoi(getenv("PMF_W_POIS")));
  int64_t w_batch = batch ? (split ? we_split : we_exact)[KB - 1] : 0;
  if (batch && getenv("PMF_W_BATCH")) w_batch = std::max(0, atoi(getenv("PMF_W_BATCH")));
  w_gauss = 16 + w_batch;
  w_bern += w_batch;
  w_pois += w_batch;
}

static int compute_work_split(pmf_ctx *c, WorkSplit &ws, int grid, int64_t n_rp, int64_t ct0, int64_t n_ct, int64_t tps, int64_t n_cseg, bool split) {
  int64_t w_gauss, w_bern, w_pois;
  noise_tile_weights(c->KB, split, c->n_bv > 0, w_gauss, w_bern, w_pois);
  const int64_t key[11] = {c->M, c->N, c->Kp, grid, n_rp, tps, n_cseg, c->kind_version, ct0, n_ct, c->mixed ? (w_gauss * 64 + w_bern) * 64 + w_pois : 0};
  if (ws.wg_begin && memcmp(key, ws.key, sizeof(key)) == 0) return 0;
  std::vector<int64_t> tw((size_t)n_ct, w_gauss);
  if (c->mixed && (int64_t)c->h_kind.size() == c->N) {
    for (int64_t ct = 0; ct < n_ct; ++ct) {
      int64_t wmax = w_gauss;
      for (int64_t j = (ct0 + ct) * 32; j < std::min<int64_t>(c->N, (ct0 + ct) * 32 + 32); ++j) {
        const int k = c->h_kind[(size_t)j];
        wmax = std::max<int64_t>(wmax, k == PMF_NOISE_BERNOULLI ? w_bern : (k == PMF_NOISE_POISSON ? w_pois : w_gauss));
      }
      tw[(size_t)ct] = wmax;
    }
  }
  // per segment: first tile, tile count, weight of one row panel's sweep
  std::vector<int64_t> seg_t0((size_t)n_cseg), seg_nt((size_t)n_cseg), seg_w((size_t)n_cseg), seg_pref((size_t)n_cseg + 1, 0);
  for (int64_t cs = 0; cs < n_cseg; ++cs) {
    seg_t0[(size_t)cs] = cs * tps;
    seg_nt[(size_t)cs] = cs == n_cseg - 1 ? n_ct - cs * tps : tps;
    int64_t w = 0;
    for (int64_t t = 0; t < seg_nt[(size_t)cs]; ++t) w += tw[(size_t)(seg_t0[(size_t)cs] + t)];
    seg_w[(size_t)cs] = w;
    seg_pref[(size_t)cs + 1] = seg_pref[(size_t)cs] + w * n_rp;
  }
  const int64_t W = seg_pref[(size_t)n_cseg], T = n_rp * n_ct;
  std::vector<int64_t> wb((size_t)grid + 1, 0);
  for (int g = 1; g < grid; ++g) {
    const int64_t target = (int64_t)((__int128)W * g / grid);
    int64_t cs = 0;
    while (cs + 1 < n_cseg && seg_pref[(size_t)cs + 1] <= target) ++cs;
    int64_t rem = target - seg_pref[(size_t)cs];
    const int64_t rp = std::min<int64_t>(n_rp - 1, rem / seg_w[(size_t)cs]);
    rem -= rp * seg_w[(size_t)cs];
    int64_t ti = 0;
    while (ti + 1 < seg_nt[(size_t)cs] && rem >= tw[(size_t)(seg_t0[(size_t)cs] + ti)]) { rem -= tw[(size_t)(seg_t0[(size_t)cs] + ti)]; ++ti; }
    int64_t idx = cs * n_rp * tps + rp * seg_nt[(size_t)cs] + ti;
    idx = std::max(idx, wb[(size_t)g - 1]);      // monotone
    wb[(size_t)g] = std::min(idx, T);
  }
  wb[(size_t)grid] = T;
  // which workgroups visit a column tile: inside segment cs (items [s0, s1) of the sequence) workgroup g visits tile
  // ti iff its range, clipped to the segment, contains an index == ti modulo the segment's tile count
  std::vector<int32_t> h_off((size_t)n_ct + 1, 0), h_idx;
  for (int64_t ct = 0; ct < n_ct; ++ct) {
    const int64_t cs = std::min<int64_t>(ct / tps, n_cseg - 1);
    const int64_t tps_cs = seg_nt[(size_t)cs], ti = ct - cs * tps;
    const int64_t s0 = cs * n_rp * tps, s1 = s0 + n_rp * tps_cs;
    int g_lo = (int)(std::upper_bound(wb.begin(), wb.begin() + grid, s0) - wb.begin()) - 1;
    if (g_lo < 0) g_lo = 0;
    for (int g = g_lo; g < grid && wb[(size_t)g] < s1; ++g) {
      const int64_t lo = std::max(wb[(size_t)g], s0), hi = std::min(wb[(size_t)g + 1], s1);
      if (hi <= lo) continue;
      const int64_t first = (lo - s0) % tps_cs;
      const int64_t d = (ti - first + tps_cs) % tps_cs;
      if (d < hi - lo) h_idx.push_back(g);
    }
    h_off[(size_t)ct + 1] = (int32_t)h_idx.size();
  }
  if (h_idx.empty()) h_idx.push_back(0);
  // the pieces, walked exactly as the kernel walks them
  ws.h_piece_base.assign((size_t)grid, 0);
  ws.piece_rp.clear();
  const int64_t seg_block = n_rp * tps;
  for (int g = 0; g < grid; ++g) {
    ws.h_piece_base[(size_t)g] = (int32_t)ws.piece_rp.size();
    for (int64_t widx = wb[(size_t)g]; widx < wb[(size_t)g + 1];) {
      const int64_t cs = std::min<int64_t>(widx / seg_block, n_cseg - 1);
      const int64_t tps_cs = seg_nt[(size_t)cs];
      const int64_t rem = widx - cs * seg_block;
      const int64_t rp = rem / tps_cs, ti0 = rem - rp * tps_cs;
      widx += std::min<int64_t>(tps_cs - ti0, wb[(size_t)g + 1] - widx);
      ws.piece_rp.push_back((int32_t)rp);
    }
  }
  PMFCHK(dev_alloc(&ws.wg_begin, (size_t)grid + 1, false));
  PMFCHK(dev_alloc(&ws.c_off, (size_t)n_ct + 1, false));
  PMFCHK(dev_alloc(&ws.c_idx, h_idx.size(), false));
  PMFCHK(dev_alloc(&ws.piece_base, (size_t)grid, false));
  PMFCHK(dev_alloc(&ws.d_piece_base_abs, (size_t)grid, false));
  HIPCHK(hipMemcpyAsync(ws.wg_begin, wb.data(), sizeof(int64_t) * ((size_t)grid + 1), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(ws.c_off, h_off.data(), sizeof(int32_t) * h_off.size(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(ws.c_idx, h_idx.data(), sizeof(int32_t) * h_idx.size(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));   // the sources are pageable host memory
  memcpy(ws.key, key, sizeof(key));
  ws.grid = grid;
  ws.serial = ++c->split_serial;
  return 0;
}

#ifdef PMF_STAMPS
static unsigned long long *g_stamps = nullptr;
extern "C" int pmf_debug_stamps(unsigned long long *out, int n) {
  if (!g_stamps) return -1;
  return hipMemcpy(out, g_stamps, sizeof(unsigned long long) * (size_t)n, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif

int harvest_events(pmf_ctx *c) {
  for (size_t e = 0; e < c->ev_used; ++e) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->ev_pool[e].first, c->ev_pool[e].second) == hipSuccess) {
      c->kernel_ms_sum += ms;
      c->kernel_launches += 1;
    }
  }
  c->ev_used = 0;
  return 0;
}

// Panel-local batch slots for row panels of BM rows (PanelSlots): built on the host from the views' row -> batch maps,
// cached until a view's rows change.  Returns through ps.ok whether every (panel, view) has at most 15 distinct batches.
static int ensure_panel_slots(pmf_ctx *c, int BM, PanelSlots **out) {
  PanelSlots &ps = c->pslots[BM == 128 ? 0 : (BM == 256 ? 1 : 2)];
  *out = &ps;
  if (ps.serial == c->views_serial && ps.BM == BM) return 0;
  const int64_t n_rp = (c->M + BM - 1) / BM;
  std::vector<uint8_t> pm((size_t)(n_rp * c->n_bv * 16), 255), rs((size_t)((int64_t)c->n_bv * c->M), 15);
  bool ok = true;
  std::vector<int> slot_of;
  for (int v = 0; v < c->n_bv && ok; ++v) {
    if ((int64_t)c->h_bor[(size_t)v].size() != c->M) { ok = false; break; }
    const int32_t *bor = c->h_bor[(size_t)v].data();
    slot_of.assign((size_t)std::max<int>(c->views[v].nb, 1), -1);
    for (int64_t rp = 0; rp < n_rp && ok; ++rp) {
      uint8_t *pmv = pm.data() + (rp * c->n_bv + v) * 16;
      int used = 0;
      const int64_t i1 = std::min<int64_t>(c->M, (rp + 1) * BM);
      for (int64_t i = rp * BM; i < i1; ++i) {
        const int b = bor[i];
        if (b < 0) continue;             // row in no batch: identity slot 15
        if (b > 254) { ok = false; break; }
        int sl = slot_of[(size_t)b];
        if (sl < 0) {
          if (used == 15) { ok = false; break; }
          sl = used++;
          slot_of[(size_t)b] = sl;
          pmv[sl] = (uint8_t)b;
        }
        rs[(size_t)((int64_t)v * c->M + i)] = (uint8_t)sl;
      }
      for (int q = 0; q < used; ++q) slot_of[pmv[q]] = -1;
    }
  }
  ps.ok = ok;
  ps.BM = BM;
  ps.serial = c->views_serial;
  if (ok) {
    PMFCHK(dev_alloc(&ps.pm, pm.size(), false));
    PMFCHK(dev_alloc(&ps.row_slot, std::max<size_t>(rs.size(), 1), false));
    HIPCHK(hipMemcpy(ps.pm, pm.data(), pm.size(), hipMemcpyHostToDevice));
    if (!rs.empty()) HIPCHK(hipMemcpy(ps.row_slot, rs.data(), rs.size(), hipMemcpyHostToDevice));
  }
  return 0;
}

// Variant and chunking of a fused data pass.
//   variant: waves per workgroup NW and 32-row blocks per wave RBW (the workgroup's row panel is 32*NW*RBW rows)
//     K <= 32 : 8 waves x 2 row blocks (per-tile overheads amortised over twice the MFMA work; x 1 with batch layers)
//     K <= 64 : 8 waves x 1          K <= 128 : 4 waves x 1 (one wave per SIMD, whole register file)
//   PMF_RBW=1 forces one row block for K <= 32 (development comparison; 4 waves x 2 blocks at K = 64 measured 3 % slower
//   than 8 x 1 and was removed)
//   chunks: the column tiles are walked in S contiguous chunks, one launch each.  S = 1 unless the context has a
//   communicator with more than one rank (then the all-reduce of chunk s's grad(Y) runs beside the launches of the later
//   chunks and of the next epoch's earlier ones, pmf_fit) or pmf_comm_set_chunks asked for it.
FusedGeom fused_geometry(pmf_ctx *c, bool want_gx, bool want_gy, bool allow_chunks) {
  FusedGeom g;
  const char *rbwenv = getenv("PMF_RBW");
  // (the batch-layer epilogue of two row blocks does not fit the 256-register budget: RBW = 2 spills and is 1.5x slower)
  g.NW = c->KB <= 2 ? 8 : 4;
  g.RBW = (c->KB == 1 && c->n_bv == 0) ? 2 : 1;
  if (rbwenv && c->KB == 1 && atoi(rbwenv) == 1) g.RBW = 1;
  // split-bf16 products (opt-in, pmf_set_precision): K <= 64; one row block per wave
  // (batch layers: through the dense LDS table only, i.e. <= 15 batches per view, and as many views as LDS has room for)
  const int sb_max_bv = c->KB == 1 ? SbCfg<1>::max_bv : (c->KB == 2 ? SbCfg<2>::max_bv : Sb4Cfg<4>::max_bv);
  // batch layers: the LDS-table path needs the dense table and <= 15 distinct batches per (view, row panel); every
  // variant with batch layers has one row block per wave, so the panel height is known here
  if (c->n_bv > 0) {
    g.bmode = 2;
    if (c->btd_ok && ensure_panel_slots(c, 32 * g.NW, &g.ps) == 0 && g.ps && g.ps->ok) g.bmode = 1;
  }
  const bool sb_batch_ok = c->n_bv == 0 || (g.bmode == 1 && c->n_bv <= sb_max_bv);
  g.sb = c->precision == PMF_PREC_BF16X3 && sb_batch_ok && (want_gx || want_gy) && !getenv("PMF_DEBUG_FLAGS");
  if (g.sb) g.RBW = 1;
  // 96 < K <= 128, both gradients: the 256-row-panel kernel (pmf_fused_sb8.hip.inc: four waves x two row blocks; PMF_SB8=0
  // keeps pmf_fused_sb4_kernel's 128-row panel).  Batch layers need the panel-local slots of the taller panel.
  // (also 32 < K <= 64: four waves x four row blocks, 512-row panel; PMF_SB8=4 restricts it to K > 96)
  {
    const char *e8 = getenv("PMF_SB8");
    const int m8 = e8 ? atoi(e8) : 1;
    g.sb8 = g.sb && want_gx && want_gy && m8 != 0 && (c->KB == 4 || (c->KB == 2 && m8 != 4));
  }
  const int bm8 = c->KB == 4 ? Sb8Cfg<4>::BM : Sb8Cfg<2>::BM;
  if (g.sb8 && c->n_bv > 0) {
    PanelSlots *ps8 = nullptr;
    if (c->n_bv <= (c->KB == 4 ? Sb8Cfg<4>::max_bv : Sb8Cfg<2>::max_bv) && ensure_panel_slots(c, bm8, &ps8) == 0 && ps8 && ps8->ok) g.ps = ps8;
    else g.sb8 = false;
  }
  if (g.sb8) { g.NW = 4; g.RBW = bm8 / 128; }
  g.BM = 32 * g.NW * g.RBW;
  g.n_rp = (c->M + g.BM - 1) / g.BM;
  g.n_ct_all = (c->N + PMF_BN - 1) / PMF_BN;
  int reserve = c->comm.nranks > 1 ? c->comm.reserve_cus : 0;
  if (const char *e = getenv("PMF_RESERVE_CUS")) reserve = std::max(0, std::min(atoi(e), c->n_cu / 2));   // (tests / A-B: a one-rank run with the multi-rank grid)
  g.grid_max = std::max(1, c->n_cu - reserve);
  int S = 1;
  if (allow_chunks && want_gy) {
    S = c->n_chunks_req > 0 ? c->n_chunks_req : 1;
    if (c->n_chunks_req <= 0 && c->comm.nranks > 1) {
      // Automatic.  In the pipelined loop of pmf_fit the all-reduce of chunk s has until the Y step of chunk s, i.e. it
      // runs beside the data pass of the other S - 1 chunks (this epoch's s+1.. and the next epoch's ..s-1): with one
      // chunk the collective is fully exposed, with two or more it is hidden as long as it is shorter than (S-1)/S of a
      // pass.  Every extra launch costs ~0.06 ms of prologue, tail and launch gap (measured with a one-rank communicator,
      // 25000 rows x 50000, K = 64: 1 / 2 / 4 / 8 chunks = 4.27 / 4.33 / 4.45 / 4.79 ms per epoch), so: two chunks, more
      // only while a chunk's all-reduce (~30 us + bytes / algorithm bandwidth; 60 GB/s assumed for an 8-GPU xGMI ring
      // at these sizes, PMF_COMM_ALGBW_GBPS overrides) would not fit beside the rest of the pass.
      const char *bw = getenv("PMF_COMM_ALGBW_GBPS");
      const double algbw = (bw && atof(bw) > 0 ? atof(bw) : 60.0) * 1e9;
      // Rank-invariant inputs only (every rank must choose the same S: it is the number and size of the collectives):
      // N, Kp, the arithmetic mode and the MEAN rows per rank that pmf_fit has all-reduced (comm.m_mean), never the local
      // row count or the kernel variant this rank's batch layout happens to allow.
      const int64_t Mref = c->comm.m_mean > 0 ? c->comm.m_mean : c->M;
      const double bytes = 4.0 * (double)c->Kp * (double)c->N;
      const double t_pass = 6.0 * (double)Mref * (double)c->N * (double)c->Kp / (c->precision == PMF_PREC_BF16X3 ? 200e12 : 115e12);
      S = 2;
      while (S < 4 && 30e-6 + bytes / S / algbw > t_pass * (S - 1) / S) ++S;
      // a chunk should give every workgroup a few dozen tiles at least (each launch pays its prologue and its tail).  Counted
      // in REFERENCE panels of 256 rows, the unit the threshold was calibrated in -- not g.BM: the panel height belongs to the
      // kernel family THIS rank runs (512 rows for pmf_fused_sb8_kernel at K <= 64, 128 where a rank's batch layout sends it to
      // pmf_fused_sb2_kernel), and two ranks that disagreed on it chose different S, i.e. different collectives.
      constexpr int64_t REF_PANEL = 256;
      const int64_t n_rp_ref = (Mref + REF_PANEL - 1) / REF_PANEL;
      const int64_t min_tiles = 32ll * std::max(1, c->n_cu - c->comm.reserve_cus);
      while (S > 1 && (n_rp_ref * g.n_ct_all) / S < min_tiles) --S;
    }
    S = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(S, PMF_MAX_CHUNKS), g.n_ct_all));
  }
  g.S = S;
  for (int s = 0; s < S; ++s) {
    g.ct0[s] = g.n_ct_all * s / S;
    g.nct[s] = g.n_ct_all * (s + 1) / S - g.ct0[s];
  }
  return g;
}

// Column segmentation of one chunk.  The work split is balanced to a tile whatever the segmentation, so the segment
// length only trades
//   (a) the fixed cost of a piece (X panel, first Y / D tile, gX flush: ~8 us) -- a workgroup walks
//       (T/G)/tps + 2 pieces -- against
//   (b) k_gy_reduce, which reads the private slabs of the ~G*tps/n_ct + 1 workgroups that visited a column tile
//       (Kp*N*4 bytes each at ~4 TB/s).
// tps* = sqrt(a/b) minimises a/tps + b*tps.
static void chunk_segments(pmf_ctx *c, const FusedGeom &g, int s, int &grid, int64_t &tps, int64_t &n_cseg) {
  const int64_t n_ct = g.nct[s], n_rp = g.n_rp;
  grid = (int)std::min<int64_t>(n_rp * n_ct, (int64_t)g.grid_max);
  const double tiles_per_wg = (double)(n_rp * n_ct) / grid;
  const double a_cost = tiles_per_wg * 8e-6;
  const double b_cost = (double)grid * (double)c->Kp * (double)n_ct * 32.0 * 4.0 / ((double)n_ct * 4e12);
  tps = (int64_t)std::llround(std::sqrt(a_cost / std::max(b_cost, 1e-12)));
  {
    const char *ts = getenv("PMF_TPS_SCALE");   // development: scale the model's segment length
    if (ts) tps = (int64_t)std::llround((double)tps * atof(ts));
  }
  tps = std::max<int64_t>(std::min<int64_t>(8, n_ct), std::min<int64_t>(tps, n_ct));
  // equal segments; the LAST one takes the remainder (it is longer, never tiny: every piece of work pays the fixed
  // prologue, so a 5-tile last segment once made one workgroup 20 % late)
  n_cseg = std::max<int64_t>(1, n_ct / tps);
}

// Everything a data pass needs before its first launch: work splits of all chunks (cached), the gX slot map, buffers.
int prepare_fused_pass(pmf_ctx *c, const FusedGeom &g, bool want_gx, bool want_gy) {
  if ((int)c->splits.size() < g.S) c->splits.resize((size_t)g.S);
  int64_t grid_sum = 0, serial_sum = 0;
  for (int s = 0; s < g.S; ++s) {
    int grid; int64_t tps, n_cseg;
    chunk_segments(c, g, s, grid, tps, n_cseg);
    PMFCHK(compute_work_split(c, c->splits[(size_t)s], grid, g.n_rp, g.ct0[s], g.nct[s], tps, n_cseg, g.sb));
    grid_sum += grid;
  }
  serial_sum = c->split_serial * PMF_MAX_CHUNKS + g.S;   // (split_serial is bumped by every recomputed split)
  if (grid_sum > c->loss_cap) {
    PMFCHK(dev_alloc(&c->loss_partial, (size_t)grid_sum));
    c->loss_cap = grid_sum;
  }
  c->n_macro = grid_sum;                 // loss partials: one per workgroup and chunk
  const int64_t slab_stride = (int64_t)c->Kp * c->N;
  if (want_gy && (size_t)g.grid_max * (size_t)slab_stride > c->gy_slabs_cap) {
    dev_free(&c->gy_slabs);
    // never read before written, EXCEPT by pmf_fused_sb8_kernel, which loads the old values of a ragged last tile's absent
    // columns without clamping (they are accumulated and never stored): one tile of padding keeps those loads in bounds
    PMFCHK(dev_alloc(&c->gy_slabs, (size_t)g.grid_max * (size_t)slab_stride + (size_t)PMF_BN * (size_t)c->Kp, false));
    c->gy_slabs_cap = (size_t)g.grid_max * (size_t)slab_stride;
  }
  if (want_gx && serial_sum != c->gx_serial) {
    // slot map of the gX partial slabs: chunk s owns slots [slot_base, slot_base + pieces); per row panel, its slots in
    // work-sequence order (chunk, then segment, then position inside the unit) -- the order k_gx_reduce sums them in
    std::vector<std::vector<int32_t>> lists((size_t)g.n_rp);
    int32_t base = 0;
    for (int s = 0; s < g.S; ++s) {
      WorkSplit &ws = c->splits[(size_t)s];
      ws.slot_base = base;
      for (size_t p = 0; p < ws.piece_rp.size(); ++p) lists[(size_t)ws.piece_rp[p]].push_back(base + (int32_t)p);
      std::vector<int32_t> abs(ws.h_piece_base);
      for (auto &v : abs) v += base;
      HIPCHK(hipMemcpy(ws.d_piece_base_abs, abs.data(), sizeof(int32_t) * abs.size(), hipMemcpyHostToDevice));
      base += (int32_t)ws.piece_rp.size();
    }
    std::vector<int32_t> off((size_t)g.n_rp + 1, 0), idx;
    for (int64_t rp = 0; rp < g.n_rp; ++rp) {
      idx.insert(idx.end(), lists[(size_t)rp].begin(), lists[(size_t)rp].end());
      off[(size_t)rp + 1] = (int32_t)idx.size();
    }
    if (idx.empty()) idx.push_back(0);
    PMFCHK(dev_alloc(&c->gx_off, off.size(), false));
    PMFCHK(dev_alloc(&c->gx_idx, idx.size(), false));
    HIPCHK(hipMemcpy(c->gx_off, off.data(), sizeof(int32_t) * off.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->gx_idx, idx.data(), sizeof(int32_t) * idx.size(), hipMemcpyHostToDevice));
    const size_t need = (size_t)base * (size_t)g.BM * (size_t)c->Kp;
    if (need > c->gx_part_cap) {
      dev_free(&c->gx_part);
      PMFCHK(dev_alloc(&c->gx_part, need, false));   // every slot is written whole by its piece before it is read
      c->gx_part_cap = need;
    }
    c->gx_serial = serial_sum;
  }
  PMFCHK(ensure_tile_flags(c));
  if (g.sb) {
    const size_t xblk = c->KB == 1 ? SbCfg<1>::BLK : (c->KB == 2 ? std::max<size_t>(SbCfg<2>::BLK, Sb8Cfg<2>::XBLK) : Sb4Cfg<4>::XBLK);
    const size_t yblk = c->KB == 1 ? SbCfg<1>::BLK : (c->KB == 2 ? std::max<size_t>(SbCfg<2>::BLK, Sb8Cfg<2>::YBLK) : std::max<size_t>(Sb4Cfg<4>::YBLK, Sb8Cfg<4>::YBLK));
    const size_t xb = (size_t)c->nRB * xblk, yb = (size_t)g.n_ct_all * yblk;
    // (+ sixteen zeroed row blocks: pmf_fused_sb8_kernel reads the blocks of a ragged last panel without clamping)
    if (xb > c->xsb_cap) { dev_free(&c->xsb); c->xsb_cap = 0; PMFCHK(dev_alloc(&c->xsb, xb + 16 * xblk, true)); c->xsb_cap = xb; }
    if (yb > c->ysb_cap) { dev_free(&c->ysb); c->ysb_cap = 0; PMFCHK(dev_alloc(&c->ysb, yb, false)); c->ysb_cap = yb; }
    if (g.sb8 && !c->sb8_scale) {
      PMFCHK(dev_alloc(&c->sb8_scale, (size_t)(1 + PMF_MAX_CHUNKS), false));
      PMFCHK(dev_alloc(&c->sb8_max, (size_t)(1 + PMF_MAX_CHUNKS)));
    }
  }
  return 0;
}

// split-bf16 operand images (k_sb_split): X once per pass, sigma*Y per chunk (its columns only: the Y step of a later
// chunk of the previous epoch may not have run yet when an earlier chunk is launched, pmf_fit)
// pmf_fused_sb8_kernel's images: the power-of-two pre-scale of the f16 pair first (a device scalar: no host round trip)
static int sb8_split(pmf_ctx *c, const float *src, const float4 *colp, int64_t n, int64_t nblk, int slot, int transposed, char *out) {
  Sb8ScaleArgs sa = {src, colp, n, c->Kp, c->sb8_max + slot, c->sb8_scale + slot};
  PMFCHK(pmf_launch_sb8_scale(c->stream, sa));
  Sb8SplitArgs sp = {src, colp, c->sb8_scale + slot, n, nblk, transposed, out};
  return pmf_launch_sb8_split(c->stream, sp, c->KB);
}
static int sb_split_x(pmf_ctx *c, bool sb8) {
  if (sb8) return sb8_split(c, c->P[0].p, nullptr, c->M, c->nRB, 0, 1, c->xsb);
  if (c->KB > 2) {
    Sb4SplitArgs s4 = {c->P[0].p, nullptr, c->M, c->nRB, c->Kp, 1, c->xsb};
    return pmf_launch_sb4_split(c->stream, s4);
  }
  SbSplitArgs sx = {c->P[0].p, nullptr, c->M, c->nRB, c->xsb};
  return c->KB == 1 ? pmf_launch_sb_split_1(c->stream, sx) : pmf_launch_sb_split_2(c->stream, sx);
}
static int sb_split_y(pmf_ctx *c, int64_t ct0, int64_t nct, bool sb8, int chunk) {
  if (sb8) {
    const int64_t c0 = ct0 * 32;
    return sb8_split(c, c->P[1].p + c0 * c->Kp, c->colp + c0, std::min<int64_t>(c->N - c0, nct * 32), nct, 1 + chunk, 0,
                     c->ysb + (size_t)ct0 * (c->KB == 4 ? Sb8Cfg<4>::YBLK : Sb8Cfg<2>::YBLK));
  }
  if (c->KB > 2) {
    const int64_t c0 = ct0 * 32;
    Sb4SplitArgs s4 = {c->P[1].p + c0 * c->Kp, c->colp + c0, c->N - c0, nct, c->Kp, 0, c->ysb + (size_t)ct0 * Sb4Cfg<4>::YBLK};
    return pmf_launch_sb4_split(c->stream, s4);
  }
  const size_t blk = c->KB == 1 ? SbCfg<1>::BLK : SbCfg<2>::BLK;
  const int64_t col0 = ct0 * 32;
  SbSplitArgs sy = {c->P[1].p + col0 * c->Kp, c->colp + col0, c->N - col0, nct, c->ysb + (size_t)ct0 * blk};
  return c->KB == 1 ? pmf_launch_sb_split_1(c->stream, sy) : pmf_launch_sb_split_2(c->stream, sy);
}

// One chunk of the data pass: the fused kernel over column tiles [ct0, ct0 + nct) and the fixed-order reduction of its
// private gY slabs; after the LAST chunk, the fixed-order reduction of the gX partial slabs.
int launch_fused_chunk(pmf_ctx *c, const FusedGeom &g, int s, bool want_gx, bool want_gy) {
  WorkSplit &ws = c->splits[(size_t)s];
  const int grid = ws.grid;
  int grid_dummy; int64_t tiles_per_seg, n_cseg;
  chunk_segments(c, g, s, grid_dummy, tiles_per_seg, n_cseg);
  const int64_t n_ct = g.nct[s], n_rp = g.n_rp;
  const int64_t slab_stride = (int64_t)c->Kp * c->N;
  int64_t loss_off = 0;
  for (int q = 0; q < s; ++q) loss_off += c->splits[(size_t)q].grid;
  FusedArgs a;
  memset(&a, 0, sizeof(a));
  a.tflags = c->tflags;
  a.wg_begin = ws.wg_begin;
  a.D = c->D; a.X = c->P[0].p; a.Y = c->P[1].p; a.gX = c->P[0].g; a.gY = c->P[1].g;
  a.colp = c->colp; a.bor = c->bor; a.btab = c->btab; a.loss_partial = c->loss_partial + loss_off;
  a.btd = g.bmode == 1 ? c->btd : nullptr; a.n_bv = c->n_bv;
  if (g.bmode == 1) {
    a.pm = g.ps->pm; a.row_slot = g.ps->row_slot; a.colview = c->colview;
    a.nbs_shift = 0;
    while ((1 << a.nbs_shift) < c->nbs) ++a.nbs_shift;
  }
  c->last_bmode = g.bmode;
  a.nRB = c->nRB; a.gy_slabs = c->gy_slabs; a.slab_stride = slab_stride; a.n_rp = n_rp;
  a.M = c->M; a.N = c->N; a.n_tiles = n_rp * n_ct; a.tps = (int)tiles_per_seg; a.n_ct = (int)n_ct; a.n_cseg = (int)n_cseg;
  a.want_gx = want_gx; a.want_gy = want_gy;
  a.ct0 = (int32_t)g.ct0[s];
  a.gx_part = c->gx_part; a.piece_base = ws.d_piece_base_abs; a.gx_slot_stride = (int64_t)g.BM * c->Kp;
  {
    const char *dbg = getenv("PMF_DEBUG_FLAGS");
    a.dbg = dbg ? atoi(dbg) : 0;
  }
#ifdef PMF_STAMPS
  {
    static unsigned long long *d_stamps = nullptr;
    if (!d_stamps) HIPCHK(hipMalloc((void **)&d_stamps, sizeof(unsigned long long) * 16 * 8 * 1024));
    HIPCHK(hipMemsetAsync(d_stamps, 0, sizeof(unsigned long long) * 16 * 8 * 1024, c->stream));
    a.stamps = d_stamps;
    g_stamps = d_stamps;
  }
#endif
  a.views = c->d_views;
  const bool batch = c->n_bv > 0;
  if (g.sb) {
    if (s == 0) PMFCHK(sb_split_x(c, g.sb8));
    PMFCHK(sb_split_y(c, g.ct0[s], n_ct, g.sb8, s));
    a.Xsb = c->xsb; a.Ysb = c->ysb;
    if (g.sb8) { a.sb_scale_x = c->sb8_scale; a.sb_scale_y = c->sb8_scale + 1 + s; }
  }
  // timing events
  if (c->ev_used == c->ev_pool.size()) {
    if (c->ev_pool.size() >= 4096) {
      HIPCHK(hipStreamSynchronize(c->stream));
      harvest_events(c);
    } else {
      hipEvent_t e0, e1;
      HIPCHK(hipEventCreate(&e0));
      HIPCHK(hipEventCreate(&e1));
      c->ev_pool.emplace_back(e0, e1);
    }
  }
  auto &ev = c->ev_pool[c->ev_used++];
  HIPCHK(hipEventRecord(ev.first, c->stream));
  int rc = 0;
  if (g.sb) {
    const bool d16 = c->store == PMF_STORE_BF16;
    typedef int (*sb_fn)(PmfDynLds *, hipStream_t, const FusedArgs &, int, bool, bool, bool, bool);
    // K in 33..64: the four-wave, two-row-block variant (pmf_fused_sb2.hip.inc); PMF_SB2=0 selects the eight-wave one
    static const bool use_sb2 = !(getenv("PMF_SB2") && atoi(getenv("PMF_SB2")) == 0);
    const sb_fn fn = c->KB == 1 ? (d16 ? pmf_launch_fused_sb_1_bf16 : pmf_launch_fused_sb_1)
                   : c->KB == 2 ? (use_sb2 ? (d16 ? pmf_launch_fused_sb2_bf16 : pmf_launch_fused_sb2)
                                           : (d16 ? pmf_launch_fused_sb_2_bf16 : pmf_launch_fused_sb_2))
                   : c->KB == 3 ? (d16 ? pmf_launch_fused_sb4_3_bf16 : pmf_launch_fused_sb4_3)
                                : (d16 ? pmf_launch_fused_sb4_4_bf16 : pmf_launch_fused_sb4_4);
    c->last_kernel = g.sb8 ? 8 : (c->KB >= 3 ? 4 : (c->KB == 2 && use_sb2 ? 2 : 1));
    if (g.sb8) rc = (c->KB == 4 ? (d16 ? pmf_launch_fused_sb8_4_bf16 : pmf_launch_fused_sb8_4)
                                : (d16 ? pmf_launch_fused_sb8_2_bf16 : pmf_launch_fused_sb8_2))(&c->dyn_lds, c->stream, a, grid, batch, c->mixed);
    else rc = fn(&c->dyn_lds, c->stream, a, grid, batch, c->mixed, want_gx, want_gy);
    c->sb_launches += 1;
  } else {
    typedef int (*ex_fn)(PmfDynLds *, hipStream_t, const FusedArgs &, int, bool, bool);
    const bool d16 = c->store == PMF_STORE_BF16;
    ex_fn fn = nullptr;
    switch (c->KB * 10 + g.RBW) {
      case 11: fn = d16 ? pmf_launch_fused_exact_11_bf16 : pmf_launch_fused_exact_11; break;
      case 12: fn = d16 ? pmf_launch_fused_exact_12_bf16 : pmf_launch_fused_exact_12; break;
      case 21: fn = d16 ? pmf_launch_fused_exact_21_bf16 : pmf_launch_fused_exact_21; break;
      case 31: fn = d16 ? pmf_launch_fused_exact_31_bf16 : pmf_launch_fused_exact_31; break;
      case 41: fn = d16 ? pmf_launch_fused_exact_41_bf16 : pmf_launch_fused_exact_41; break;
      default: return pmf_fail("unsupported KB=%d", c->KB);
    }
    c->last_kernel = 0;
    rc = fn(&c->dyn_lds, c->stream, a, grid, batch, c->mixed);
  }
  PMFCHK(rc);
  HIPCHK(hipEventRecord(ev.second, c->stream));   // the events bracket pmf_fused_kernel alone (= rocprofv3's kernel duration)
  if (want_gy && !(a.dbg & 8)) {
    k_gy_reduce<<<dim3((unsigned)n_ct, (unsigned)(32 * c->Kp / 256)), 64, 0, c->stream>>>(c->gy_slabs, slab_stride, ws.c_off, ws.c_idx,
                                                                                      c->Kp, c->N, c->P[1].g, (int)g.ct0[s],
                                                                                      g.sb8 ? c->colp : nullptr);
    HIPCHK(hipGetLastError());
  }
  if (want_gx && s == g.S - 1) {
    const int64_t panel = (int64_t)g.BM * c->Kp;
    k_gx_reduce<<<dim3((unsigned)g.n_rp, (unsigned)((panel / 4 + 255) / 256)), 256, 0, c->stream>>>(
        c->gx_part, panel, c->gx_off, c->gx_idx, panel, (int64_t)c->Kp * c->M, c->P[0].g);
    HIPCHK(hipGetLastError());
  }
  return 0;
}

// the whole data pass in one go (step-level API, single-chunk callers)
static int launch_fused(pmf_ctx *c, bool want_gx, bool want_gy) {
  const FusedGeom g = fused_geometry(c, want_gx, want_gy, /*allow_chunks=*/c->n_chunks_req > 0);   // (chunks only on request)
  PMFCHK(prepare_fused_pass(c, g, want_gx, want_gy));
  for (int s = 0; s < g.S; ++s) PMFCHK(launch_fused_chunk(c, g, s, want_gx, want_gy));
  return 0;
}

static int launch_layer_grad(pmf_ctx *c, const pmf_fit_opts *o, bool with_loss) {
  LayerGradArgs a;
  memset(&a, 0, sizeof(a));
  a.D = c->D; a.d_bf16 = c->store == PMF_STORE_BF16; a.X = c->P[0].p; a.Y = c->P[1].p; a.colp = c->colp; a.bor = c->bor; a.btab = c->btab;
  const int fl = o->frozen_layers;
  a.g_logsigma = (fl & 1) ? nullptr : c->P[2].g;
  a.g_logdelta = ((fl & 2) || c->n_bv == 0) ? nullptr : c->P[4].g;
  a.g_mu = (fl & 4) ? nullptr : c->P[3].g;
  a.g_theta = ((fl & 8) || c->n_bv == 0) ? nullptr : c->P[5].g;
  a.M = c->M; a.N = c->N; a.nRB = c->nRB; a.Kp = c->Kp; a.K = c->K;
  int max_nb = 1;
  for (int v = 0; v < c->n_bv; ++v) {
    a.views[v] = c->views[v];
    a.val_off[v] = c->val_off[v];
    max_nb = std::max(max_nb, c->views[v].nb);
  }
  a.max_nb = max_nb;
  const int gx = nblocks(c->N, 64);
  // ~32 single-wave workgroups per CU hide the FMA / LDS latencies; at least 256 rows per workgroup keep the final
  // atomics (64 * (2 + 2 nb) per workgroup) negligible
  int64_t gy = std::max<int64_t>(1, std::min<int64_t>((32ll * c->n_cu + gx - 1) / gx, (c->M + 255) / 256));
  a.rows_per_block = (int)((c->M + gy - 1) / gy);
  gy = (c->M + a.rows_per_block - 1) / a.rows_per_block;
  const int64_t nslots = (int64_t)gx * gy;
  if (with_loss) {
    if (nslots > c->loss_cap) {
      PMFCHK(dev_alloc(&c->loss_partial, (size_t)nslots));
      c->loss_cap = nslots;
    }
    c->n_macro = nslots;
    a.loss_partial = c->loss_partial;
  }
  // private partials per block row, summed in fixed order afterwards (every entry of a vector is written by its block)
  const int64_t nbt = c->n_bv > 0 ? c->val_off[c->n_bv] : 0;
  a.nbt = nbt;
  a.part_stride = 2 * c->N + 2 * nbt;
  if ((size_t)(a.part_stride * gy) > c->lgrad_part_cap) {
    PMFCHK(dev_alloc(&c->lgrad_part, (size_t)(a.part_stride * gy), false));
    c->lgrad_part_cap = (size_t)(a.part_stride * gy);
  }
  a.part = c->lgrad_part;
  const size_t lds = sizeof(float) * (size_t)(4 * c->Kp + 2 * max_nb * 64);
  if (lds > 160 * 1024) return pmf_fail("too many row batches per view (%d) for the layer-gradient kernel", max_nb);
  void (*kern)(const LayerGradArgs) = nullptr;
  switch (c->KB) {
    case 1: kern = k_layer_grad<1>; break;
    case 2: kern = k_layer_grad<2>; break;
    case 3: kern = k_layer_grad<3>; break;
    case 4: kern = k_layer_grad<4>; break;
    default: return pmf_fail("unsupported KB=%d", c->KB);
  }
  PMFCHK(ensure_dyn_lds(c, (const void *)kern, lds));
  hipLaunchKernelGGL(kern, dim3(gx, (unsigned)gy), dim3(64), lds, c->stream, a);
  HIPCHK(hipGetLastError());
  PMFCHK(sum_parts(c, a.part, a.part_stride, (int)gy, a.g_mu, c->N));
  PMFCHK(sum_parts(c, a.part + c->N, a.part_stride, (int)gy, a.g_logsigma, c->N));
  PMFCHK(sum_parts(c, a.part + 2 * c->N, a.part_stride, (int)gy, a.g_theta, nbt));
  PMFCHK(sum_parts(c, a.part + 2 * c->N + nbt, a.part_stride, (int)gy, a.g_logdelta, nbt));
  return 0;
}

// Layer-parameter gradients through the MFMA layer pass (its own loss only in layer-only epochs); K <= 64 and <= 15 batches per view, otherwise the
// VALU kernel above (PMF_LAYER_OLD=1 forces it, for comparison).  K <= 128; views with <= 15 batches.
static int layer_pass_waves(pmf_ctx *c) {   // waves per workgroup of the layer pass
  const char *lnwenv = getenv("PMF_LAYER_NW");
  return (c->KB <= 2 && !(lnwenv && atoi(lnwenv) == 4)) ? 8 : 4;
}
static bool layer_pass_eligible(pmf_ctx *c) {
  const char *e = getenv("PMF_LAYER_OLD");
  if (e && atoi(e) == 1) return false;
  if (c->KB > 4 || (c->n_bv > 0 && !c->btd_ok)) return false;
  // the batch table of the unit's 64 columns (nbs slots each) lives in LDS next to the Y tiles and the X panels
  return pmf_layer_pass_lds(c->KB, layer_pass_waves(c), c->n_bv > 0 ? c->nbs : 16) <= 160 * 1024;
}
static int launch_layer_pass(pmf_ctx *c, const pmf_fit_opts *o, bool with_loss) {
  const int lnw = layer_pass_waves(c);
  const int nbs = c->n_bv > 0 ? c->nbs : 16;
  int nbs_shift = 0;
  while ((1 << nbs_shift) < nbs) ++nbs_shift;
  const int64_t n_ct = (c->N + PMF_BN - 1) / PMF_BN, n_rp = (c->M + 32 * lnw - 1) / (32 * lnw);
  const int64_t n_seg = (n_ct + PMF_LS - 1) / PMF_LS;
  int64_t R = std::max<int64_t>(1, std::min<int64_t>(n_rp, (4ll * c->n_cu + n_seg - 1) / n_seg));
  const int grid = (int)std::min<int64_t>(n_seg * R, c->n_cu);
  // one private [N][nbs] table per (row range, wave, lane half): plain read-modify-writes instead of float atomics, summed
  // in fixed order by k_layer_map.  R * N is bounded by ~4 n_cu * 64 columns, so the tables take ~130 MB at nbs = 16.
  const int64_t lg_stride = c->N * nbs, n_parts = R * lnw * 2;
  if (lg_stride * n_parts > c->LG_cap) {
    PMFCHK(dev_alloc(&c->LG, (size_t)(lg_stride * n_parts), false));
    c->LG_cap = lg_stride * n_parts;
  }
  HIPCHK(hipMemsetAsync(c->LG, 0, sizeof(float2) * (size_t)(lg_stride * n_parts), c->stream));
  if (with_loss) {
    if (grid > c->loss_cap) {
      PMFCHK(dev_alloc(&c->loss_partial, (size_t)grid));
      c->loss_cap = grid;
    }
    c->n_macro = grid;
  }
  LayerPassArgs a;
  memset(&a, 0, sizeof(a));
  a.D = c->D; a.d_bf16 = c->store == PMF_STORE_BF16; a.nRB = c->nRB; a.X = c->P[0].p; a.Y = c->P[1].p; a.colp = c->colp; a.bor = c->bor;
  a.btd = c->n_bv > 0 ? c->btd : nullptr; a.LG = c->LG; a.lg_stride = lg_stride; a.loss_partial = with_loss ? c->loss_partial : nullptr;
  a.M = c->M; a.N = c->N; a.n_bv = c->n_bv; a.n_ct = (int)n_ct; a.n_rp = (int)n_rp; a.n_seg = (int)n_seg; a.R = (int)R;
  a.nbs_shift = nbs_shift;
  PMFCHK(pmf_launch_layer_pass(&c->dyn_lds, c->stream, c->KB, lnw, c->mixed, grid, a));
  LayerMapArgs m;
  memset(&m, 0, sizeof(m));
  const int fl = o->frozen_layers;
  m.LG = c->LG; m.lg_stride = lg_stride; m.n_parts = (int32_t)n_parts; m.btd = a.btd; m.colp = c->colp; m.N = c->N; m.n_bv = c->n_bv; m.nbs_shift = nbs_shift;
  m.g_logsigma = (fl & 1) ? nullptr : c->P[2].g;
  m.g_logdelta = ((fl & 2) || c->n_bv == 0) ? nullptr : c->P[4].g;
  m.g_mu = (fl & 4) ? nullptr : c->P[3].g;
  m.g_theta = ((fl & 8) || c->n_bv == 0) ? nullptr : c->P[5].g;
  for (int v = 0; v < c->n_bv; ++v) { m.views[v] = c->views[v]; m.val_off[v] = c->val_off[v]; }
  return pmf_launch_layer_map(c->stream, m);
}

int check_ready(pmf_ctx *c) {
  if (!c->D) return pmf_fail("data not set");
  if (c->K == 0) return pmf_fail("factors not set");
  for (int v = 0; v < c->n_bv; ++v)
    if (c->views[v].nb == 0) return pmf_fail("batch view %d declared but not set", v);
  return 0;
}

// Regularizer + optimizer step of elements [e0, e0 + n) of parameter `which` (whole tensor: e0 = 0, n = b.n).  For X / Y,
// e0 and n are multiples of Kp (whole columns): the pipelined loop of pmf_fit steps Y one column chunk at a time.
// `max_blocks` bounds the grid (= loss partials appended to the slab); `advance` moves Adam's running beta powers on
// (once per epoch, with the last slice).
int step_param_range(pmf_ctx *c, int which, int64_t e0, int64_t n, bool do_step, bool use_reg, int reg_slot,
                            int *reg_count, int max_blocks, bool advance) {
  ParamBuf &b = c->P[which];
  if (b.n == 0 || n == 0) return 0;
  StepArgs s;
  memset(&s, 0, sizeof(s));
  s.p = b.p + e0; s.g = b.g + e0; s.acc = b.acc + e0; s.mom = b.mom + e0;
  s.wq = b.wq ? b.wq + e0 : nullptr; s.cq = b.cq ? b.cq + e0 : nullptr;
  s.Kp = which <= 1 ? c->Kp : 1;
  s.K = which <= 1 ? c->K : 1;
  if (which == 1 && c->has_ard) { s.ard_alpha = c->ard_alpha + e0 / s.Kp; s.ard_beta = c->ard_beta + e0; s.ard_scale = c->ard_scale; }
  s.n = n;
  s.opt_kind = c->opt_kind; s.lr = c->lr; s.eps = c->eps; s.b1 = c->b1; s.b2 = c->b2;
  s.c1 = 1.f - b.bp1; s.c2 = 1.f - b.bp2;
  s.do_step = do_step; s.use_reg = use_reg;
  const int grid = (int)std::min<int64_t>(max_blocks, nblocks(n, 256));
  s.reg_partial = c->reg_partial + (int64_t)reg_slot * REG_SLOTS + *reg_count;
  if (*reg_count + grid > REG_SLOTS) return pmf_fail("internal: regularizer partial slab overflow");
  k_reg_step<<<grid, 256, 0, c->stream>>>(s);
  HIPCHK(hipGetLastError());
  *reg_count += grid;
  if (advance && do_step && c->opt_kind == PMF_OPT_ADAM) { b.bp1 *= c->b1; b.bp2 *= c->b2; }
  return 0;
}
int step_param(pmf_ctx *c, int which, bool do_step, bool use_reg, int reg_slot, int *reg_count) {
  // the four layer parameters share one slab of loss partials: each gets a quarter (k_reg_step is grid-stride)
  return step_param_range(c, which, 0, c->P[which].n, do_step, use_reg, reg_slot, reg_count, reg_slot == 2 ? REG_SLOTS / 4 : REG_SLOTS, true);
}
// layer l <-> param: 1 logsigma(2), 2 logdelta(4), 3 mu(3), 4 theta(5)
int step_layers(pmf_ctx *c, const pmf_fit_opts *o, int *reg_count) {
  const int fl = o->frozen_layers, fr = o->frozen_regs;
  const int pmap[4] = {2, 4, 3, 5};
  for (int l = 0; l < 4; ++l) {
    const bool frozen = (fl >> l) & 1;
    const bool reg_on = !frozen && !((fr >> l) & 1);
    if (frozen) continue;  // FrozenLayer: no step, regularizer evaluates to 0 (regularizers.jl:508-510, 887-889)
    PMFCHK(step_param(c, pmap[l], true, reg_on, 2, reg_count));
  }
  c->prepared = false;
  return 0;
}

// what every epoch starts with: optimizer state, prepared column / batch tables, cleared layer gradients
int epoch_open(pmf_ctx *c, const pmf_fit_opts *o) {
  if (!c->state_init) PMFCHK(init_opt_state(c));
  if (!c->prepared || o->update_col_layers) PMFCHK(prepare(c));
  // (grad(X) and grad(Y) need no clearing: k_gx_reduce / k_gy_reduce overwrite every element)
  if (o->update_col_layers)
    for (int w = 2; w < 6; ++w)
      if (c->P[w].n) HIPCHK(hipMemsetAsync(c->P[w].g, 0, sizeof(float) * (size_t)c->P[w].n, c->stream));
  return 0;
}
int epoch_layer_pass(pmf_ctx *c, const pmf_fit_opts *o, bool with_loss) {
  if (layer_pass_eligible(c)) { c->last_layer_path = 1; return launch_layer_pass(c, o, with_loss); }
  c->last_layer_path = 2;
  return launch_layer_grad(c, o, with_loss);
}

// Which batch-layer variants the last launches took (tests and benchmarks: a silent fall-back to the slow paths is a
// performance bug).  bmode: 0 none, 1 LDS table with panel-local slots, 2 per-entry gathers; layer_path: 1 MFMA layer
// pass, 2 VALU layer kernel; slots: columns of the dense batch table.
extern "C" int pmf_debug_last_kernel(pmf_ctx *c, int *kernel) {
  if (!c) return pmf_fail("null context");
  if (kernel) *kernel = c->last_kernel;
  return 0;
}

extern "C" int pmf_debug_last_path(pmf_ctx *c, int *bmode, int *layer_path, int *slots) {
  if (!c) return pmf_fail("null context");
  if (bmode) *bmode = c->last_bmode;
  if (layer_path) *layer_path = c->last_layer_path;
  if (slots) *slots = c->nbs;
  return 0;
}

extern "C" int pmf_epoch_begin(pmf_ctx *c, const pmf_fit_opts *o) {
  PMFCHK(ctx_bind(c));
  PMFCHK(check_ready(c));
  if (!o) return pmf_fail("null opts");
  PMFCHK(epoch_open(c, o));
  for (int q = 0; q < 4; ++q) c->reg_counts[q] = 0;
  const bool fused = o->update_X || o->update_Y || !o->update_col_layers;
  if (fused) PMFCHK(launch_fused(c, o->update_X != 0, o->update_Y != 0));
  if (o->update_col_layers) PMFCHK(epoch_layer_pass(c, o, !fused));
  return 0;
}

extern "C" int pmf_epoch_step_local(pmf_ctx *c, const pmf_fit_opts *o) {
  PMFCHK(ctx_bind(c));
  if (o->update_X) PMFCHK(step_param(c, 0, true, true, 0, &c->reg_counts[0]));
  return 0;
}

extern "C" int pmf_epoch_step_shared(pmf_ctx *c, const pmf_fit_opts *o) {
  PMFCHK(ctx_bind(c));
  if (o->update_Y) PMFCHK(step_param(c, 1, true, true, 1, &c->reg_counts[1]));
  if (o->update_col_layers) PMFCHK(step_layers(c, o, &c->reg_counts[2]));
  return 0;
}

extern "C" int pmf_epoch_loss(pmf_ctx *c, double *local_loss, double *shared_terms) {
  PMFCHK(ctx_bind(c));
  RegCounts rc;
  for (int q = 0; q < 4; ++q) rc.c[q] = c->reg_counts[q];
  PMFCHK(launch_loss_reduce(c, rc, 0x1f));
  HIPCHK(hipMemcpyAsync(c->h_loss, c->d_loss, sizeof(double) * 5, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  harvest_events(c);
  const double shared = c->h_loss[2] + c->h_loss[3];
  if (local_loss) *local_loss = c->h_loss[0] + c->h_loss[1] + shared;
  if (shared_terms) *shared_terms = shared;
  return 0;
}


extern "C" int pmf_grad_device_ptr(pmf_ctx *c, int which, void **ptr, int64_t *n) {
  PMFCHK(ctx_bind(c));
  if (which < 0 || which > 5) return pmf_fail("bad parameter id %d", which);
  if (ptr) *ptr = c->P[which].g;
  if (n) *n = c->P[which].n;
  return 0;
}

extern "C" int pmf_get_grad(pmf_ctx *c, int which, int view, float *out) {
  PMFCHK(ctx_bind(c));
  if (which < 0 || which > 5) return pmf_fail("bad parameter id %d", which);
  HIPCHK(hipStreamSynchronize(c->stream));
  if (which <= 1) return download_padded(c, out, c->P[which].g, which == 0 ? c->M : c->N);
  if (which <= 3) {
    HIPCHK(hipMemcpy(out, c->P[which].g, sizeof(float) * (size_t)c->N, hipMemcpyDeviceToHost));
    return 0;
  }
  if (view < 0 || view >= c->n_bv) return pmf_fail("batch view %d out of range", view);
  const int64_t off = c->val_off[view], n = c->val_off[view + 1] - off;
  HIPCHK(hipMemcpy(out, c->P[which].g + off, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost));
  return 0;
}

// optimizer state of one parameter group in the reference's shape: AdaGrad's accumulator (or Adam's second moment)
// and Adam's first moment (Flux.Optimise.AdaGrad.acc / Adam's (mt, vt), reached through src/optimizers.jl:6-13)
extern "C" int pmf_get_opt_state(pmf_ctx *c, int which, int view, float *acc, float *mom) {
  PMFCHK(ctx_bind(c));
  if (which < 0 || which > 5) return pmf_fail("bad parameter id %d", which);
  if (!c->state_init) return pmf_fail("optimizer state not initialised yet (no epoch has run since pmf_set_optimizer)");
  HIPCHK(hipStreamSynchronize(c->stream));
  const ParamBuf &b = c->P[which];
  if (which <= 1) {
    const int64_t n = which == 0 ? c->M : c->N;
    if (acc) PMFCHK(download_padded(c, acc, b.acc, n));
    if (mom) PMFCHK(download_padded(c, mom, b.mom, n));
    return 0;
  }
  int64_t off = 0, n = c->N;
  if (which >= 4) {
    if (view < 0 || view >= c->n_bv) return pmf_fail("batch view %d out of range", view);
    off = c->val_off[view];
    n = c->val_off[view + 1] - off;
  }
  if (acc) HIPCHK(hipMemcpy(acc, b.acc + off, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost));
  if (mom) HIPCHK(hipMemcpy(mom, b.mom + off, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost));
  return 0;
}



static int run_forward(pmf_ctx *c, float *Zdev, int synth, uint64_t seed, float noise, float frac_nan, int64_t nRB = 0) {
  PMFCHK(check_ready(c));
  PMFCHK(prepare(c));
  ForwardArgs a;
  memset(&a, 0, sizeof(a));
  a.X = c->P[0].p; a.Y = c->P[1].p; a.colp = c->colp; a.bor = c->bor; a.btab = c->btab; a.Z = Zdev;
  a.z_bf16 = nRB > 0 && c->store == PMF_STORE_BF16;
  a.M = c->M; a.N = c->N; a.nRB = nRB; a.Kp = c->Kp; a.K = c->K; a.synth = synth; a.seed = seed; a.noise = noise; a.frac_nan = frac_nan;
  for (int v = 0; v < c->n_bv; ++v) a.views[v] = c->views[v];
  if ((c->M + 255) / 256 > 65535) return pmf_fail("M too large for the forward kernel grid");
  hipLaunchKernelGGL(k_forward, dim3((unsigned)c->N, (unsigned)((c->M + 255) / 256)), dim3(256), 0, c->stream, a);
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int pmf_forward(pmf_ctx *c, float *Z_host) {
  PMFCHK(ctx_bind(c));
  if (!Z_host) return pmf_fail("null output");
  const size_t bytes = sizeof(float) * (size_t)(c->M * c->N);
  float *Z = nullptr;
  HIPCHK(hipMalloc((void **)&Z, bytes));
  int rc = run_forward(c, Z, 0, 0, 0.f, 0.f);
  if (rc == 0) {
    hipError_t e = hipMemcpyAsync(Z_host, Z, bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) rc = pmf_fail("copy back failed: %s", hipGetErrorString(e));
  }
  (void)hipFree(Z);
  return rc;
}

extern "C" int pmf_synth_data(pmf_ctx *c, uint64_t seed, float noise, float frac_nan) {
  PMFCHK(ctx_bind(c));
  if (!c->D) return pmf_fail("data buffer not allocated (pmf_set_data_device(ctx, NULL, M, N, store))");
  c->tflags_valid = false;
  PMFCHK(run_forward(c, (float *)c->D, 1, seed, noise, frac_nan, c->nRB));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int pmf_stats(pmf_ctx *c, int use_factors, float *col_n, float *col_sum, float *col_sumsq, float *col_sqerr,
                         float *col_ssq_grad, float *batch_count, float *batch_sqerr) {
  PMFCHK(ctx_bind(c));
  PMFCHK(check_ready(c));
  PMFCHK(prepare(c));
  const int64_t nbt = c->n_bv > 0 ? c->val_off[c->n_bv] : 0;
  const size_t nfl = (size_t)(5 * c->N + 2 * nbt);
  StatsArgs a;
  memset(&a, 0, sizeof(a));
  a.D = c->D; a.d_bf16 = c->store == PMF_STORE_BF16; a.X = c->P[0].p; a.Y = c->P[1].p; a.colp = c->colp; a.bor = c->bor; a.btab = c->btab;
  a.M = c->M; a.N = c->N; a.nRB = c->nRB; a.Kp = c->Kp; a.K = c->K; a.use_factors = use_factors;
  int max_nb = 1;
  for (int v = 0; v < c->n_bv; ++v) {
    a.views[v] = c->views[v];
    a.val_off[v] = c->val_off[v];
    max_nb = std::max(max_nb, c->views[v].nb);
  }
  a.max_nb = max_nb;
  const int gx = nblocks(c->N, 64);
  int64_t gy = std::max<int64_t>(1, std::min<int64_t>((32ll * c->n_cu + gx - 1) / gx, (c->M + 255) / 256));
  a.rows_per_block = (int)((c->M + gy - 1) / gy);
  gy = (c->M + a.rows_per_block - 1) / a.rows_per_block;
  // scratch: [results nfl][gy private partial vectors of nfl]; every entry of a partial vector is written by its block row,
  // the block rows are then added in fixed order (k_sum_parts): no float atomics, bitwise reproducible statistics
  PMFCHK(ensure_scratch(c, sizeof(float) * std::max<size_t>(nfl * (size_t)(gy + 1), 1)));
  float *buf = (float *)c->scratch, *part = buf + nfl;
  a.part_stride = (int64_t)nfl;
  a.col_n = part; a.col_sum = part + c->N; a.col_sumsq = part + 2 * c->N; a.col_sqerr = part + 3 * c->N; a.col_ssqg = part + 4 * c->N;
  a.b_n = nbt ? part + 5 * c->N : nullptr;
  a.b_sqerr = nbt ? part + 5 * c->N + nbt : nullptr;
  const size_t lds = sizeof(float) * (size_t)(4 * c->Kp + 2 * max_nb * 64);
  if (lds > 160 * 1024) return pmf_fail("too many row batches per view (%d) for the statistics kernel", max_nb);
  void (*kst)(const StatsArgs) = nullptr;
  switch (c->KB) {
    case 1: kst = k_stats<1>; break;
    case 2: kst = k_stats<2>; break;
    case 3: kst = k_stats<3>; break;
    case 4: kst = k_stats<4>; break;
    default: return pmf_fail("unsupported KB=%d", c->KB);
  }
  PMFCHK(ensure_dyn_lds(c, (const void *)kst, lds));
  hipLaunchKernelGGL(kst, dim3(gx, (unsigned)gy), dim3(64), lds, c->stream, a);
  HIPCHK(hipGetLastError());
  PMFCHK(sum_parts(c, part, (int64_t)nfl, (int)gy, buf, (int64_t)nfl));
  HIPCHK(hipStreamSynchronize(c->stream));
  float *outs[5] = {col_n, col_sum, col_sumsq, col_sqerr, col_ssq_grad};
  for (int q = 0; q < 5; ++q)
    if (outs[q]) HIPCHK(hipMemcpy(outs[q], buf + (int64_t)q * c->N, sizeof(float) * (size_t)c->N, hipMemcpyDeviceToHost));
  if (batch_count && nbt) HIPCHK(hipMemcpy(batch_count, buf + 5 * c->N, sizeof(float) * (size_t)nbt, hipMemcpyDeviceToHost));
  if (batch_sqerr && nbt) HIPCHK(hipMemcpy(batch_sqerr, buf + 5 * c->N + nbt, sizeof(float) * (size_t)nbt, hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int pmf_kernel_time(pmf_ctx *c, double *mean_ms, int64_t *launches, int reset) {
  PMFCHK(ctx_bind(c));
  HIPCHK(hipStreamSynchronize(c->stream));
  harvest_events(c);
  if (mean_ms) *mean_ms = c->kernel_launches > 0 ? c->kernel_ms_sum / (double)c->kernel_launches : 0.0;
  if (launches) *launches = c->kernel_launches;
  if (reset) { c->kernel_ms_sum = 0.0; c->kernel_launches = 0; }
  return 0;
}
