// pmf_ctx.h -- internal to libpmf_hip.so: the context behind the opaque `pmf_ctx` of include/pmf_hip.h and the host
// helpers its translation units share.
//   pmf_hip.hip       C ABI (marshalling, step-level API, statistics), small kernels, work split, data-pass launches
//   pmf_comm_fit.hip  cross-rank exchange (RCCL via dlopen / host-staged transport) and the epoch loop pmf_fit
//   pmf_fsard.hip     the FeatureSetARD update_A! solver (ISTA on the device)
#ifndef PMF_CTX_H
#define PMF_CTX_H
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "pmf_common.h"

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
struct ParamBuf {  // one trainable parameter tensor with its gradient, optimizer state and quadratic regularizer
  float *p = nullptr, *g = nullptr, *acc = nullptr, *mom = nullptr;
  float *wq = nullptr, *cq = nullptr;  // dense quadratic weights / centres: 0.5*wq*(p-cq)^2 (nullptr = none)
  int64_t n = 0;
  float bp1 = 0.f, bp2 = 0.f;  // Adam running beta powers
};

// Panel-local batch slots.  The fused kernels look the batch parameters of (row, column) up in a 16-slot per-column table
// in LDS.  With more than 15 batches in a view the slots are numbered PER ROW PANEL: slot s of (panel, view) is the s-th
// distinct batch among the panel's rows (rows are normally sorted by batch: a 256-row panel holds a few), slot 15 the
// identity.  pm[(rp * n_bv + v) * 16 + s] = that batch (255 = unused), row_slot[v * M + i] = the slot of row i.
// ok = every (panel, view) has at most 15 distinct batches; otherwise the launch takes the gather fallback.
struct PanelSlots {
  int64_t serial = -1;
  int BM = 0;
  bool ok = false;
  uint8_t *pm = nullptr, *row_slot = nullptr;
};

// Cached work split of one column chunk of the fused data pass (see compute_work_split).
struct WorkSplit {
  int64_t key[11] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
  int grid = 0;
  int64_t serial = 0;
  int64_t *wg_begin = nullptr;     // [grid + 1] device: first work item of each workgroup
  int32_t *c_off = nullptr, *c_idx = nullptr;   // per column tile of the chunk: the workgroups that visit it (CSR; k_gy_reduce)
  int32_t *piece_base = nullptr;   // [grid] device: chunk-relative slot of each workgroup's first piece (gX partial buffer)
  std::vector<int32_t> h_piece_base;
  std::vector<int32_t> piece_rp;   // host: row panel of every piece, in slot order
  int32_t slot_base = 0;           // first slot of this chunk in the flat gX partial buffer (set with the CSR)
  int32_t *d_piece_base_abs = nullptr;   // [grid] device: piece_base + slot_base
};

// Cross-rank exchange of the sharded fit: rows are sharded, Y / column layers replicated; per epoch the
// partial grad(Y) (and layer gradients, and the local loss) are summed over the ranks.  Two transports behind one
// interface: RCCL (ncclAllReduce on a communication stream of the library's own, overlapped with the data pass), or a
// host-staged callback (tests: two ranks sharing one GPU cannot form an RCCL ring).
struct Comm {
  int rank = 0, nranks = 1;
  void *nccl = nullptr;            // ncclComm_t (RCCL transport)
  pmf_host_allreduce_fn host_fn = nullptr;
  void *host_user = nullptr;
  hipStream_t stream = nullptr;    // communication stream
  void *stage = nullptr;           // pinned staging buffer of the host-staged transport
  size_t stage_bytes = 0;
  int reserve_cus = 0;             // CUs left to the collective's kernels by the fused pass (RCCL transport)
  int cta_cap = 0;                 // workgroups the communicator is configured for (ncclConfig_t.maxCTAs); 0 = RCCL's default
  void *dstage = nullptr;          // device staging buffer of pmf_comm_allreduce (host buffers), grown on demand
  size_t dstage_bytes = 0;
  bool broken = false;             // a fit failed with collectives possibly unmatched: the communicator must be destroyed
  int64_t m_mean = 0;              // rows per rank averaged over the ranks (pmf_fit): the rank-invariant size behind the chunk count
  std::vector<hipEvent_t> ev_ready, ev_done;   // per chunk: gY slice complete / its all-reduce complete
  hipEvent_t ev_loss_ready = nullptr, ev_loss_done = nullptr, ev_layer_ready = nullptr, ev_layer_done = nullptr;
  int64_t n_allreduce = 0;         // collectives issued (diagnostics / tests)
};

struct pmf_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = true;
  int n_cu = 256;
  int64_t M = 0, N = 0;
  int K = 0, Kp = 0, KB = 0;
  void *D = nullptr;   // tile-major copy of the data matrix, f32 (pmf_d_off) or bf16 (pmf_d_off16, `store`), always library-owned
  bool own_D = false;
  int64_t D_M = 0, D_Npad = 0, nRB = 0;
  uint32_t *tflags = nullptr;     // per 32x32 tile of D: 1 = all 1024 entries finite (fast epilogue path of the fused kernel)
  int64_t tflags_cap = 0;
  bool tflags_valid = false;
  int store = PMF_STORE_F32;
  ParamBuf P[6];  // X, Y, logsigma, mu, logdelta, theta
  // batch views
  int n_bv = 0;
  std::vector<ViewDesc> views;
  std::vector<int64_t> val_off;  // per view offset into the flat logdelta/theta arrays
  std::vector<int64_t> bvb_off;  // per view offset into the flat per-(view,batch) arrays
  int32_t *bor = nullptr;
  float2 *btab = nullptr;
  bool views_dirty = true;        // `views` changed since the last upload
  ViewDesc *d_views = nullptr;    // device copy of `views` for the fused kernel's global-gather fallback
  float2 *LG = nullptr;           // [R * waves * 2][N][nbs] {S_G, S_Q}: the private tables of the layer pass (pmf_layers.hip.inc)
  int64_t LG_cap = 0;
  float *lgrad_part = nullptr;    // [block rows][2N + 2 nbt] private partials of k_layer_grad (fixed-order k_sum_parts)
  size_t lgrad_part_cap = 0;
  float2 *btd = nullptr;          // dense per-column batch table [ceil(N/32)*32][nbs] (fused kernels, layer pass); see k_dense_btab
  int64_t btd_cap = 0;
  bool btd_ok = false;            // the dense table is built (every view has <= 255 batches)
  int nbs = 16;                   // slots per column of the dense table: 16, or the power of two above the largest batch count
  int64_t colview_cap = 0;
  uint8_t *colview = nullptr;     // [ceil(N/32)*32] view of every column, 255 = none (k_dense_btab)
  std::vector<std::vector<int32_t>> h_bor;   // host copy of batch_of_row per view (panel-local slot maps)
  PanelSlots pslots[3];           // panel-local batch slots for row panels of 128 / 256 / 512 rows (ensure_panel_slots)
  int64_t views_serial = 0;       // bumped whenever a view's rows / shape change
  int last_kernel = 0;            // fused kernel family of the last data pass (pmf_debug_last_kernel)
  int last_bmode = 0, last_layer_path = 0;   // diagnostics (pmf_debug_last_path)
  int32_t *d_val_view = nullptr;  // per flat value element: view id
  // noise model / prepared column parameters
  int32_t *colmeta = nullptr;  // kind | (view+1)<<2
  float *colw = nullptr;
  float4 *colp = nullptr;
  bool mixed = false;
  bool prepared = false;
  // ARD-type regularizer on Y
  float *ard_alpha = nullptr, *ard_beta = nullptr;
  float ard_scale = 0.f;
  bool has_ard = false;
  // optimizer
  int opt_kind = PMF_OPT_ADAGRAD;
  float lr = 1.f, eps = 1e-8f, b1 = 0.9f, b2 = 0.999f;
  bool state_init = false;
  // loss plumbing
  double *loss_partial = nullptr;
  std::vector<uint8_t> h_kind;    // host copy of the per-column noise kind (cost model of the work split)
  std::vector<WorkSplit> splits;  // cached work split of every column chunk of the fused pass (compute_work_split)
  int n_chunks_req = 0;           // column chunks per data pass: 0 = automatic (1 on one GPU; pmf_comm_set_chunks)
  float *gx_part = nullptr;       // [pieces][BM x Kp] per-piece partial sums of gX (fused kernel), summed by k_gx_reduce
  size_t gx_part_cap = 0;         // floats
  int32_t *gx_off = nullptr, *gx_idx = nullptr;   // per row panel: the slots of its pieces, in work-sequence order (CSR)
  int64_t gx_serial = -1;         // sum of the splits' serials the CSR was built for
  int64_t split_serial = 0;       // bumped whenever a split is recomputed
  Comm comm;                      // cross-rank exchange (pmf_comm_init*); nranks == 1: none
  int last_chunks = 1;            // column chunks of the last pmf_fit's data pass (pmf_comm_info)
  hipEvent_t ev_host = nullptr;   // "the epoch's loss has reached the host" (pmf_fit)
  int64_t kind_version = 0;
  float *gy_slabs = nullptr;      // [grid][Kp x N] private per-workgroup gY partial sums of the fused kernel
  size_t gy_slabs_cap = 0;        // floats
  int precision = PMF_PREC_F32;   // products of the fused data pass: exact f32 MFMA, or split-bf16 (pmf_set_precision)
  char *xsb = nullptr, *ysb = nullptr;   // split-bf16 operand images of X / sigma*Y, rebuilt every epoch (k_sb_split)
  size_t xsb_cap = 0, ysb_cap = 0;       // bytes
  float *sb8_scale = nullptr;     // [1 + chunks] power-of-two pre-scales of pmf_fused_sb8_kernel's f16 images: X, then Y per chunk
  uint32_t *sb8_max = nullptr;    // [1 + chunks] bit patterns of max |operand| (k_sb8_absmax)
  int64_t sb_launches = 0;        // fused launches that took the split-bf16 kernel (pmf_get_precision)
  int64_t loss_cap = 0;
  int64_t n_macro = 0;
  double *reg_partial = nullptr;  // [4][REG_SLOTS]
  double *d_loss = nullptr;       // device [8]
  double *h_loss = nullptr;       // pinned host [8]
  // fused-kernel timing
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  size_t ev_used = 0;
  double kernel_ms_sum = 0.0;
  int64_t kernel_launches = 0;
  int reg_counts[4] = {0, 0, 0, 0};  // used slots of the regularizer partial slabs (0 X, 1 Y, 2 column layers)
  // scratch
  void *scratch = nullptr;
  size_t scratch_bytes = 0;
  // largest dynamic-LDS size set so far per kernel ON THIS CONTEXT'S DEVICE (hipFuncSetAttribute is per device: a
  // process-wide cache would leave a second GPU's kernels without the attribute)
  PmfDynLds dyn_lds;
};

#define REG_SLOTS 1024
#define PMF_MAX_CHUNKS 16

// Geometry of one fused data pass: kernel variant, row panels, column chunks.
struct FusedGeom {
  int NW = 8, RBW = 1, BM = 256, grid_max = 256, S = 1;
  int bmode = 0;      // batch layers: 0 none, 1 LDS table with panel-local slots, 2 per-entry global gathers (fallback)
  PanelSlots *ps = nullptr;
  bool sb = false;    // split-bf16 products: pmf_fused_sb_kernel (K <= 64) or pmf_fused_sb4_kernel (64 < K <= 128)
  bool sb8 = false;   // ... of them, pmf_fused_sb8_kernel: 96 < K <= 128, both gradients, 256-row panel
  int64_t n_rp = 0, n_ct_all = 0;
  int64_t ct0[PMF_MAX_CHUNKS], nct[PMF_MAX_CHUNKS];
};

struct RegCounts { int c[4]; };

// hipMemset on device memory is queued on the NULL stream and may return before it has run; the library's kernels and
// copies run on a NON-BLOCKING stream, which the NULL stream does not order (memset_now, pmf_hip.hip).
int memset_now(void *p, int v, size_t bytes);
template <typename T>
static int dev_alloc(T **p, size_t n, bool zero = true) {
  if (*p) {
    HIPCHK(hipFree(*p));
    *p = nullptr;
  }
  if (n == 0) n = 1;
  HIPCHK(hipMalloc((void **)p, n * sizeof(T)));
  if (zero) PMFCHK(memset_now(*p, 0, n * sizeof(T)));
  return 0;
}
template <typename T>
static void dev_free(T **p) {
  if (*p) (void)hipFree(*p);
  *p = nullptr;
}

__device__ __forceinline__ double block_reduce_sum(double v, double *sh) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0)
    for (int q = 0; q < (int)(blockDim.x >> 6); ++q) s += sh[q];
  return s;  // valid on thread 0
}

// ---- defined in pmf_hip.hip
int ctx_bind(pmf_ctx *c);
int ensure_dyn_lds(pmf_ctx *c, const void *kern, size_t lds);
int check_ready(pmf_ctx *c);
int harvest_events(pmf_ctx *c);
int epoch_open(pmf_ctx *c, const pmf_fit_opts *o);
int epoch_layer_pass(pmf_ctx *c, const pmf_fit_opts *o, bool with_loss);
FusedGeom fused_geometry(pmf_ctx *c, bool want_gx, bool want_gy, bool allow_chunks);
int prepare_fused_pass(pmf_ctx *c, const FusedGeom &g, bool want_gx, bool want_gy);
int launch_fused_chunk(pmf_ctx *c, const FusedGeom &g, int s, bool want_gx, bool want_gy);
int step_param_range(pmf_ctx *c, int which, int64_t e0, int64_t n, bool do_step, bool use_reg, int reg_slot, int *reg_count,
                     int max_blocks, bool advance);
int step_param(pmf_ctx *c, int which, bool do_step, bool use_reg, int reg_slot, int *reg_count);
int step_layers(pmf_ctx *c, const pmf_fit_opts *o, int *reg_count);
// fixed-order reduction of the loss partial slabs into d_loss[which] for every bit `which` of mask (0 data term, 1 X reg,
// 2 Y reg, 3 layer regs, 4 spare)
int launch_loss_reduce(pmf_ctx *c, const RegCounts &rc, int mask);
// ---- defined in pmf_comm_fit.hip
int comm_release(pmf_ctx *c);
bool comm_active(const pmf_ctx *c);
#endif
