// pmf_common.h -- what the translation units of libpmf_hip.so share: the kernels' argument structs, the device layout of
// the data matrix, small device helpers, and the host-side launchers each kernel file exports.  The library is built
// from several translation units so that they compile in parallel (csrc/Makefile):
//   pmf_hip.hip        C ABI, small kernels, work split, epoch loop, communicator
//   pmf_k_fused.hip    pmf_fused_kernel (pmf_fused.hip.inc), once per (K blocks, row blocks per wave)
//   pmf_k_sb.hip       pmf_fused_sb_kernel + k_sb_split (pmf_fused_sb.hip.inc), once per K-block count
//   pmf_k_layers.hip   pmf_layer_kernel + k_layer_map (pmf_layers.hip.inc)
#ifndef PMF_COMMON_H
#define PMF_COMMON_H
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>
#include <unordered_map>

#include "../../include/pmf_hip.h"

// ---- errors (defined in pmf_hip.hip)
int pmf_fail(const char *fmt, ...);
#define HIPCHK(x)                                                                                    \
  do {                                                                                               \
    hipError_t e_ = (x);                                                                             \
    if (e_ != hipSuccess) return pmf_fail("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)
#define PMFCHK(x)          \
  do {                     \
    int r_ = (x);          \
    if (r_ < 0) return r_; \
  } while (0)

// largest dynamic-LDS size set so far per kernel on one context's device (hipFuncSetAttribute is per device)
typedef std::unordered_map<const void *, size_t> PmfDynLds;
int pmf_ensure_dyn_lds(PmfDynLds *cache, const void *kern, size_t lds);

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define PMF_MAXV 16
#define PMF_PM_BYTES 256   // LDS: a piece's slot -> batch map [16 views][16 slots] (fused kernels with batch layers)
#define PMF_BN 32      // columns per tile
#define PMF_DPAD 64    // D is padded with NaN columns to a multiple of this

// Device layout of the data matrix.  The library owns its copy of D, so it is free to re-lay it out once at upload:
// D is stored as 32 x 32 tiles, tile (rb, cb) at ((cb * nRB) + rb) * 1024 floats (the 8 row blocks of a workgroup's
// panel are contiguous for one column block), and INSIDE a tile in the register order of the MFMA accumulator:
// float offset q*256 + lane*4 + e holds element (i = lane & 31, j = rowmap(4q + e, lane >> 5)).  A wave therefore
// streams its tile with four global_load_dwordx4, each instruction 1 KiB contiguous.  Rows / columns beyond M / N
// are NaN (= missing), so the kernel needs no bounds checks on D.
__host__ __device__ __forceinline__ int64_t pmf_d_off(int64_t i, int64_t j, int64_t nRB) {
  const int il = (int)(i & 31), jl = (int)(j & 31);
  const int h = (jl >> 2) & 1;
  const int r = (jl & 3) + 4 * (jl >> 3);
  return (((j >> 5) * nRB) + (i >> 5)) * 1024 + (r >> 2) * 256 + (h * 32 + il) * 4 + (r & 3);
}

struct ViewDesc {
  int64_t c0;       // 0-based first column of the view
  int64_t c1;       // one past the last column
  int64_t tab_off;  // offset of the view's table in btab (float2 {delta, theta}, nb x Nv column-major)
  int32_t nb;
  int32_t pad;
};

struct FusedArgs {
  const void *D;        // tile-major f32 (pmf_d_off) or bf16 (pmf_d_off16); nRB row blocks x roundup(N,64)/32 column blocks
  const uint32_t *tflags;  // per 32x32 tile (same order as the tiles of D): 1 = every entry finite (k_tile_flags)
  int64_t nRB;
  const float *X;
  const float *Y;
  float *gX;
  float *gY;            // (unused by the kernel: gY is produced by k_gy_reduce from the private slabs)
  float *gy_slabs;      // [gridDim.x][Kp x N] per-workgroup private partial sums of gY
  int64_t slab_stride;  // Kp * N
  int64_t n_rp;         // row panels
  int64_t n_tiles;      // n_rp * (column tiles): the kernel's work
  const int64_t *wg_begin;  // [grid + 1] first work item of every workgroup's range (cost-balanced on the host)
  int32_t tps;          // column tiles per segment (the last segment may have fewer)
  int32_t n_ct;         // column tiles
  const float4 *colp;   // per column {sigma, mu, weight, meta}; meta = kind | (view+1)<<2
  const int32_t *bor;   // n_bv x M
  const float2 *btab;
  const float2 *btd;    // dense per-column batch table [N padded to 32][1 << nbs_shift] {delta, theta} (last slot = identity); null = not available
  // panel-local batch slots (PanelSlots, pmf_hip.hip): slot s of (row panel rp, view v) holds batch pm[(rp*n_bv + v)*16 + s]
  // (255 = unused); row_slot[v*M + i] = the slot of row i (15 = no batch); colview[j] = view of column j (255 = none)
  const uint8_t *pm, *row_slot, *colview;
  int32_t nbs_shift;
  int32_t n_bv;         // batch views
  double *loss_partial; // one slot per workgroup
  int64_t M, N;
  int32_t n_cseg;
  int32_t want_gx, want_gy;
  unsigned long long *stamps;  // diagnostic build only (PMF_STAMPS)
  int32_t dbg;          // development ablation bits (PMF_DEBUG_FLAGS): 1 no slab reduce, 2 no D loads, 4 no epilogue, 8 no gY atomics
  const ViewDesc *views;   // device array [n_bv] (a by-value array indexed per lane would be copied to scratch)
  const char *Xsb, *Ysb;   // split-bf16 operand images of X and sigma*Y (pmf_fused_sb.hip.inc); null on the exact-f32 path
  // One launch covers the column tiles [ct0, ct0 + n_ct) (a "chunk": the whole matrix on one GPU; with several ranks the
  // columns are walked in a few chunks so that the all-reduce of a finished chunk's grad(Y) overlaps the next chunk).
  int32_t ct0;
  // grad(X) without atomics: every piece STORES its 32*NW*RBW x Kp partial sum to a slot of its own; k_gx_reduce sums
  // the slots of a row panel in a fixed order (bitwise reproducible gX).  Slot of a workgroup's p-th piece =
  // piece_base[wg] + p (pieces are numbered in the order of the work sequence; built on the host with the work split).
  float *gx_part;
  const int32_t *piece_base;   // [grid]
  int64_t gx_slot_stride;      // floats per slot = BM * Kp
  // pmf_fused_sb8_kernel: power-of-two pre-scales of the f16 operand images of X and of sigma*Y (device scalars written by
  // k_sb8_scale right before the images; the forward accumulators are multiplied by 1 / (sx sy))
  const float *sb_scale_x, *sb_scale_y;
};

__device__ __forceinline__ int pmf_rowmap(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ __forceinline__ bool pmf_finite(float y) { return fabsf(y) <= 3.402823466e38f; }  // false for NaN, +-Inf

// loss and dloss/dz of one entry.  SELF-SPECIFIED noise models (MatFac is un-vendored):
//   normal 0.5 w (z-y)^2 ; bernoulli w (softplus(z) - y z) ; poisson w (exp(z) - y z)
__device__ __forceinline__ void pmf_noise(int kind, float z, float y, float w, float &l, float &g) {
  if (kind == PMF_NOISE_NORMAL) {
    const float d = z - y;
    g = w * d;
    l = 0.5f * g * d;
  } else if (kind == PMF_NOISE_BERNOULLI) {
    const float e = __expf(-fabsf(z));
    const float sp = fmaxf(z, 0.f) + __logf(1.f + e);
    const float r = __frcp_rn(1.f + e);
    const float sg = z >= 0.f ? r : e * r;
    l = w * (sp - y * z);
    g = w * (sg - y);
  } else {
    const float e = __expf(z);
    l = w * (e - y * z);
    g = w * (e - y);
  }
}

#define PMF_SCHED_FENCE() asm volatile("" ::: "memory")

// Streaming (non-temporal) 16-B accesses for data that is touched once per pass (the D tiles, the private gY slab
// entries): they do not displace the Y tiles of the column segment from the XCD's L2.
typedef float pmf_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 pmf_load_stream(const float4 *p) {
  const pmf_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const pmf_f32x4 *>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void pmf_store_stream(float4 *p, const float4 &v) {
  const pmf_f32x4 vv = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(vv, reinterpret_cast<pmf_f32x4 *>(p));
}

// Diagnostic build only (-DPMF_STAMPS): per-phase shader-cycle totals, one row of 16 counters per workgroup,
// written to a buffer nothing else reads (cdna_hip_programming.md section 7, "In-kernel stamps").
#if defined(PMF_STAMPS) && defined(PMF_STAMP_FROM)
// two-stamp variant (-DPMF_STAMPS -DPMF_STAMP_FROM=a -DPMF_STAMP_TO=b): only the interval from stamp a to stamp b is timed
// (into slot 1), so that the diagnostic costs two scalar loads and four registers instead of a stamp at every phase edge
#define PMF_STAMP(slot)                                                                      \
  do {                                                                                       \
    if ((slot) == PMF_STAMP_FROM || (slot) == PMF_STAMP_TO) {                                \
      unsigned long long t_;                                                                 \
      __builtin_amdgcn_sched_barrier(0);                                                     \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");             \
      __builtin_amdgcn_sched_barrier(0);                                                     \
      if ((slot) == PMF_STAMP_TO) st_acc[1] += t_ - st_prev;                                 \
      if ((slot) == PMF_STAMP_FROM) st_prev = t_;                                            \
    }                                                                                        \
  } while (0)
#elif defined(PMF_STAMPS)
#define PMF_STAMP(slot)                                                                      \
  do {                                                                                       \
    unsigned long long t_;                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    st_acc[slot] += t_ - st_prev;                                                            \
    st_prev = t_;                                                                            \
  } while (0)
#else
#define PMF_STAMP(slot) do { } while (0)
#endif


// ---- split-bf16 operand images (pmf_fused_sb.hip.inc)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// KB = K blocks of 32 factors: 1 (K <= 32) or 2 (K <= 64).  An image is [32 rows][32 KB] bf16 = 2 KB KiB, rows of 64 KB bytes.
// Chunk swizzle f(row) per row length (both kinds of read conflict-free, checked with the bank rules of
// MI355X_MICROARCH.md "LDS" and by SQ_LDS_BANK_CONFLICT = 0):
//   64-B rows  : (row >> 2) & 3        (four rows per 256-B bank window; a transposed read takes four whole rows)
//   128-B rows : x ^ ((x & 1) << 2), x = (row >> 1) & 7
template <int KB>
__host__ __device__ __forceinline__ int pmf_sb_f(int row) {
  if (KB == 1) return (row >> 2) & 3;
  const int x = (row >> 1) & 7;
  return x ^ ((x & 1) << 2);
}
template <int KB>
__host__ __device__ __forceinline__ int pmf_sb_off(int row, int ch) { return 64 * KB * row + 16 * (ch ^ pmf_sb_f<KB>(row)); }

template <int KB>
struct SbCfg {
  static constexpr int NW = 8, Kp = 32 * KB, SLAB = 32 * Kp;
  static constexpr int IMG = 2048 * KB;    // bytes of one [32][Kp] bf16 image
  static constexpr int BLK = 3 * IMG;      // global block of 32 rows: hi image, mid image, lo image
  static constexpr int LDSP = 2 * IMG;     // the part every product reads (hi, mid); the lo image only feeds the forward
  static constexpr size_t lds_bytes = 2 * LDSP + IMG + NW * LDSP + NW * SLAB * 4 + 8 * PMF_BN * 4 + 16 + 8 * NW;
  // batch-layer variants add the tile's dense batch table [32 columns][16] float2 and, per wave, [n_bv views][32 rows] batch ids
  static constexpr size_t lds_batch(int n_bv) { return PMF_BN * 16 * sizeof(float2) + PMF_PM_BYTES + (size_t)NW * n_bv * 32; }
  static constexpr int max_bv = KB == 1 ? PMF_MAXV : (int)((160 * 1024 - lds_bytes - PMF_BN * 16 * sizeof(float2) - PMF_PM_BYTES) / (NW * 32));
};

// src: [n][Kp] f32 (rows = samples of X or columns of Y, K padded to Kp); scale = colp (sigma in .x) or null.
// out: one block per 32 rows: hi, mid and lo image.  Rows >= n are zero.  One thread per (row, 16-B chunk).
struct SbSplitArgs {
  const float *src;
  const float4 *colp;
  int64_t n, nblk;
  char *out;
};

// pmf_fused_sb2_kernel (32 < K <= 64, four waves x two row blocks): LDS footprint
struct Sb2Cfg {
  static constexpr size_t lds_bytes = (2 * 2 * 4096 + 4096 + 8 * 2 * 4096 + 4 * 8192 + 4 * 4096 + 8 * PMF_BN * 4 + 16 + 8 * 4 + 15) / 16 * 16;
  static constexpr size_t lds_batch(int n_bv) { return PMF_BN * 16 * sizeof(float2) + PMF_PM_BYTES + (size_t)4 * 2 * n_bv * 32; }
  static constexpr int max_bv = PMF_MAXV;
};

// transposed LDS read of eight bf16 (two ds_read_b64_tr_b16)
__device__ __forceinline__ bf16x8 pmf_sb_tr8(const char *p0, const char *p1) {
  typedef s16x4 __attribute__((address_space(3))) * lds_p;
  const s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p0));
  const s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p1));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 t = __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, t);
}


// ---- split-bf16 operands for 64 < K <= 128 (pmf_fused_sb4.hip.inc): 256-byte image rows whatever K
__host__ __device__ __forceinline__ int pmf_sb4_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }
template <int KB>
struct Sb4Cfg {
  static constexpr int NW = 4, Kp = 32 * KB, SLAB = 32 * Kp;
  static constexpr int IMG = 32 * 256;     // bytes of one [32 rows][128 k] bf16 image
  static constexpr int XBLK = 5 * IMG;     // per 32 samples: hi, mid, lo row images + hi, mid transposed images
  static constexpr int YBLK = 3 * IMG;     // per 32 features: hi, mid, lo row images
  static constexpr size_t lds_bytes = (2 * 2 * IMG + IMG + NW * SLAB * 4 + 8 * PMF_BN * 4 + 16 + 8 * NW + 15) / 16 * 16;
  static constexpr size_t lds_batch(int n_bv) { return PMF_BN * 16 * sizeof(float2) + PMF_PM_BYTES + (size_t)NW * n_bv * 32; }
  static constexpr int max_bv = PMF_MAXV;
};
struct Sb4SplitArgs {
  const float *src;
  const float4 *colp;
  int64_t n, nblk;
  int32_t Ksrc, transposed;   // transposed: also write the two transposed images (X); block stride 5 images, else 3
  char *out;
};
// ---- pmf_fused_sb8_kernel (pmf_fused_sb8.hip.inc): K = 128 (96 < K <= 128), 256-row panel, four waves x two row blocks.
// The forward product runs on f16 PAIRS (hi = f16(s a), lo' = f16((s a - hi) 2^11); s a power of two that brings max|a| to
// [2^12, 2^13)): 22 significant bits in two terms and three MFMAs per k-step (hi hi into one accumulator, hi lo' + lo' hi into a
// second one worth 2^-11) -- 1e-7 of max|Z| at every operand scale (scripts/f16x2_probe.hip), where bf16 needs three terms
// and six MFMAs.  The gradient products stay on bf16 (hi, mid), whose range needs no care.
// Operand block of 32 rows: [f16 hi][f16 lo'] images in k-step-major order (pmf_sb8_f_off: the 16-B chunk of k-step S, lane
// half h', row r at 1024 S + 512 h' + 16 r -- the 64 lanes of a forward fragment read read 1 KiB contiguous, no swizzle, and
// the k-step is an immediate offset: one address register instead of eight) + two bf16 images:
//   X: hi, mid TRANSPOSED ([k][i], 64-byte rows, 16-B chunk c of row k at c ^ ((k >> 2) & 3): conflict-free ds_read_b128 with
//      lane = k) -- GEMM3's B operand;   Y: hi, mid row images (pmf_sb4_off) -- GEMM2's B operand through transposed reads.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int KB>
__host__ __device__ __forceinline__ int pmf_sb8_f_off(int row, int ch) { return 1024 * (ch % (2 * KB)) + 512 * (ch / (2 * KB)) + 16 * row; }
__host__ __device__ __forceinline__ int pmf_sb8_xt_off(int k, int c) { return 64 * k + 16 * (c ^ ((k >> 2) & 3)); }
// KB = 4 (96 < K <= 128): two row blocks per wave, 256-row panel.  KB = 2 (32 < K <= 64): four row blocks per wave, 512-row panel.
// Every image of a 32-row block is 64 Kp bytes (f16 pair, bf16 pair, 64-byte-row transposed images alike).
template <int KB>
struct Sb8Cfg {
  static_assert(KB == 4 || KB == 2, "pmf_fused_sb8_kernel: KB = 4 or 2");
  static constexpr int NW = 4, RB = 8 / KB, Kp = 32 * KB, BM = 32 * NW * RB, NBLK = NW * RB;
  static constexpr int IMG = 64 * Kp;
  static constexpr int XBLK = 4 * IMG, YBLK = 4 * IMG;
  // LDS: Y bf16 hi + mid (double buffered) | Y f16 hi + lo' (single) | one G image (hi, lo: 4 KiB) per row block of the panel |
  // X^T hi of the panel's row blocks | column parameters, tile kind / view, loss partials
  static constexpr size_t lds_bytes = (2 * 2 * IMG + 2 * IMG + NBLK * 4096 + NBLK * IMG + 8 * PMF_BN * 4 + 16 + 8 * NW + 15) / 16 * 16;
  static constexpr size_t lds_batch(int n_bv) { return PMF_BN * 16 * sizeof(float2) + PMF_PM_BYTES + (size_t)NW * RB * n_bv * 32; }
  static constexpr int max_bv = (int)((160 * 1024 - lds_bytes - PMF_BN * 16 * sizeof(float2) - PMF_PM_BYTES) / (NW * RB * 32)) < PMF_MAXV
                                    ? (int)((160 * 1024 - lds_bytes - PMF_BN * 16 * sizeof(float2) - PMF_PM_BYTES) / (NW * RB * 32)) : PMF_MAXV;
};
// bf16 row image of a Y tile (GEMM2's transposed reads): the layouts of the other split kernels (256-byte rows at Kp = 128,
// 128-byte rows at Kp = 64), conflict-free for ds_read_b64_tr_b16
template <int KB>
__host__ __device__ __forceinline__ int pmf_sb8_y_off(int row, int ch) { return KB == 4 ? pmf_sb4_off(row, ch) : pmf_sb_off<2>(row, ch); }
struct Sb8SplitArgs {
  const float *src;       // [n][Kp] f32 (Kp = 32 KB)
  const float4 *colp;     // .x = sigma (Y) or null (X)
  const float *scale;     // device scalar: power-of-two pre-scale of the f16 images
  int64_t n, nblk;
  int32_t transposed;     // X: the bf16 images are the transposed ones
  char *out;
};
struct Sb8ScaleArgs {
  const float *src;
  const float4 *colp;
  int64_t n;              // rows of src
  int32_t Kp;             // floats per row
  uint32_t *max_bits;     // scratch word (zeroed by the launcher)
  float *scale_out;       // the device scalar the images and the fused kernel read
};

// Index into the dense batch table of entry (column j, panel-local slot s) for a column of view vw (255 = none), given the
// piece's slot -> batch map Pm (LDS, [n_bv][16]).  Unused slots and columns outside every view hit the identity slot.
// (32-bit element index: the host keeps the table below 2^31 entries, so the load is scalar base + one offset register)
__device__ __forceinline__ uint32_t pmf_bt_index(const FusedArgs &a, const unsigned char *Pm, int64_t j, int s, int vw) {
  const int ident = (1 << a.nbs_shift) - 1;
  int b = ident;
  if (vw != 255) { const int q = Pm[vw * 16 + s]; if (q != 255) b = q; }
  return ((uint32_t)j << a.nbs_shift) + (uint32_t)b;
}

// Entry e (= column-in-tile * 16 + slot) of the LDS batch table of the tile at column j; vw = pmf_bt_view(a, j, e).
__device__ __forceinline__ int pmf_bt_view(const FusedArgs &a, int64_t j, int e) { return a.colview[j + (e >> 4)]; }
__device__ __forceinline__ float2 pmf_bt_entry(const FusedArgs &a, const unsigned char *Pm, int64_t j, int e, int vw) {
  return a.btd[pmf_bt_index(a, Pm, j + (e >> 4), e & 15, vw)];
}

// Workgroup b of a persistent grid of g -> its position in the work order.  Workgroups are dispatched to the 8 XCDs round
// robin (b % 8); the workgroups of one XCD get CONSECUTIVE ranges of the work, so that they share column segments whose Y
// tiles stay in that XCD's L2.  Any g: XCD x holds g/8 workgroups, the first g % 8 XCDs one more (a grid that leaves a few
// CUs to RCCL is not a multiple of 8).
__device__ __forceinline__ int pmf_xcd_wg(unsigned b, unsigned g) {
  const unsigned x = b & 7, q = g >> 3, r = g & 7;
  return (int)(x * q + (x < r ? x : r) + (b >> 3));
}

// ---- the data matrix tile in registers, f32 or bf16 storage
// bf16 device layout of D (PMF_STORE_BF16): 32 x 32 tiles of 2 KiB in the same tile order as the f32 layout; inside a
// tile 32-bit word q*256 + lane*4 + e holds accumulator registers 8q + 2e (low half) and 8q + 2e + 1 (high half) of
// lane = h*32 + (i & 31): a wave streams its tile with two 1-KiB loads.
__host__ __device__ __forceinline__ int64_t pmf_d_off16(int64_t i, int64_t j, int64_t nRB) {
  const int il = (int)(i & 31), jl = (int)(j & 31);
  const int h = (jl >> 2) & 1;
  const int r = (jl & 3) + 4 * (jl >> 3);
  return (((j >> 5) * nRB) + (i >> 5)) * 1024 + ((r >> 3) * 256 + (h * 32 + il) * 4 + ((r & 7) >> 1)) * 2 + (r & 1);
}
// DB: the data matrix is stored as bf16 (PMF_STORE_BF16): two 1-KiB wave loads per tile instead of four; the registers
// keep the raw words until the epilogue converts them (a load's result must not be touched where it is issued).
template <bool DB>
struct PmfDTile;
template <>
struct PmfDTile<false> {
  f32x16 v;
  static constexpr int NCH = 4;
  __device__ __forceinline__ float get(int r) const { return v[r]; }
  __device__ __forceinline__ void load(const void *D, int64_t tile, int lane, int q) {
    const float4 x = pmf_load_stream(reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(D) + tile * 1024) + lane + 64 * q);
    v[4 * q + 0] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
  }
  __device__ __forceinline__ void touch() const { const f32x16 k = v; asm volatile("" ::"v"(k)); }
};
template <>
struct PmfDTile<true> {
  uint32_t w[8];   // word e of chunk q = registers 8q + 2e (low half) and 8q + 2e + 1 (high half)
  static constexpr int NCH = 2;
  __device__ __forceinline__ float get(int r) const {
    const uint32_t x = w[r >> 1];
    return __uint_as_float((r & 1) ? (x & 0xffff0000u) : (x << 16));
  }
  __device__ __forceinline__ void load(const void *D, int64_t tile, int lane, int q) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 x = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(reinterpret_cast<const uint16_t *>(D) + tile * 1024) + lane + 64 * q);
    w[4 * q + 0] = x.x; w[4 * q + 1] = x.y; w[4 * q + 2] = x.z; w[4 * q + 3] = x.w;
  }
  __device__ __forceinline__ void touch() const {
    const uint32_t a0 = w[0], a1 = w[1], a2 = w[2], a3 = w[3], a4 = w[4], a5 = w[5], a6 = w[6], a7 = w[7];
    asm volatile("" ::"v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
  }
};


// ---- layer pass (pmf_layers.hip.inc)
#define PMF_LS 2   // column tiles per unit (4 was measured: more per-tile state in registers, more spills, 19.6 vs 13.5 ms)

struct LayerPassArgs {
  const void *D;        // tile-major f32 or bf16 (d_bf16)
  int32_t d_bf16;
  int64_t nRB;
  const float *X, *Y;
  const float4 *colp;
  const int32_t *bor;      // n_bv x M row -> batch (or -1)
  const float2 *btd;       // dense batch table [N padded to 32][nbs] {delta, theta}; null when there are no batch views
  float2 *LG;              // [R * NW * 2][N][nbs] {S_G, S_Q}: one PRIVATE table per (row range rr, wave, lane half), zeroed by the host
  int64_t lg_stride;       // before the launch (float2 per table = N << nbs_shift); k_layer_map sums the tables in fixed order
  int32_t nbs_shift;       // nbs = 1 << nbs_shift slots per column (last = identity: row in no batch); 16 without batch views
  double *loss_partial;    // one per workgroup; null = the loss is computed elsewhere (combined epochs)
  int64_t M, N;
  int32_t n_bv, n_ct, n_rp, n_seg, R;
};

struct LayerMapArgs {
  const float2 *LG;        // n_parts private tables of lg_stride float2 each (see LayerPassArgs)
  int64_t lg_stride;
  int32_t n_parts;
  const float2 *btd;
  const float4 *colp;
  float *g_logsigma, *g_mu, *g_logdelta, *g_theta;   // any may be null
  int64_t N;
  int32_t n_bv, nbs_shift;
  ViewDesc views[PMF_MAXV];
  int64_t val_off[PMF_MAXV];
};

// ---- launchers exported by the kernel files
int pmf_launch_fused_exact_11(PmfDynLds *cache, hipStream_t stream, const FusedArgs &a, int grid, bool batch, bool mixed);
int pmf_launch_fused_exact_12(PmfDynLds *cache, hipStream_t stream, const FusedArgs &a, int grid, bool batch, bool mixed);
int pmf_launch_fused_exact_21(PmfDynLds *cache, hipStream_t stream, const FusedArgs &a, int grid, bool batch, bool mixed);
int pmf_launch_fused_exact_31(PmfDynLds *cache, hipStream_t stream, const FusedArgs &a, int grid, bool batch, bool mixed);
int pmf_launch_fused_exact_41(PmfDynLds *cache, hipStream_t stream, const FusedArgs &a, int grid, bool batch, bool mixed);
int pmf_launch_fused_exact_11_bf16(PmfDynLds *cache, hipStream_t stream, const FusedArgs &a, int grid, bool batch, bool mixed);
int pmf_launch_fused_exact_12_bf16(PmfDynLds *cache, hipStream_t stream, const FusedArgs &a, int grid, bool batch, bool mixed);
int pmf_launch_fused_exact_21_bf16(PmfDynLds *cache, hipStream_t stream, const FusedArgs &a, int grid, bool batch, bool mixed);
int pmf_launch_fused_exact_31_bf16(PmfDynLds *cache, hipStream_t stream, const FusedArgs &a, int grid, bool batch, bool mixed);
int pmf_launch_fused_exact_41_bf16(PmfDynLds *cache, hipStream_t stream, const FusedArgs &a, int grid, bool batch, bool mixed);
int pmf_launch_fused_sb_1(PmfDynLds *cache, hipStream_t stream, const FusedArgs &a, int grid, bool batch, bool mixed, bool want_gx, bool want_gy);
int pmf_launch_fused_sb_2(PmfDynLds *cache, hipStream_t stream, const FusedArgs &a, int grid, bool batch, bool mixed, bool want_gx, bool want_gy);
int pmf_launch_fused_sb_1_bf16(PmfDynLds *cache, hipStream_t stream, const FusedArgs &a, int grid, bool batch, bool mixed, bool want_gx, bool want_gy);
int pmf_launch_fused_sb_2_bf16(PmfDynLds *cache, hipStream_t stream, const FusedArgs &a, int grid, bool batch, bool mixed, bool want_gx, bool want_gy);
int pmf_launch_fused_sb2(PmfDynLds *cache, hipStream_t stream, const FusedArgs &a, int grid, bool batch, bool mixed, bool want_gx, bool want_gy);
int pmf_launch_fused_sb2_bf16(PmfDynLds *cache, hipStream_t stream, const FusedArgs &a, int grid, bool batch, bool mixed, bool want_gx, bool want_gy);
int pmf_launch_fused_sb4_4(PmfDynLds *cache, hipStream_t stream, const FusedArgs &a, int grid, bool batch, bool mixed, bool want_gx, bool want_gy);
int pmf_launch_fused_sb4_4_bf16(PmfDynLds *cache, hipStream_t stream, const FusedArgs &a, int grid, bool batch, bool mixed, bool want_gx, bool want_gy);
int pmf_launch_fused_sb4_3(PmfDynLds *cache, hipStream_t stream, const FusedArgs &a, int grid, bool batch, bool mixed, bool want_gx, bool want_gy);
int pmf_launch_fused_sb4_3_bf16(PmfDynLds *cache, hipStream_t stream, const FusedArgs &a, int grid, bool batch, bool mixed, bool want_gx, bool want_gy);
int pmf_launch_sb4_split(hipStream_t stream, const Sb4SplitArgs &a);
int pmf_launch_fused_sb8_4(PmfDynLds *cache, hipStream_t stream, const FusedArgs &a, int grid, bool batch, bool mixed);
int pmf_launch_fused_sb8_4_bf16(PmfDynLds *cache, hipStream_t stream, const FusedArgs &a, int grid, bool batch, bool mixed);
int pmf_launch_fused_sb8_2(PmfDynLds *cache, hipStream_t stream, const FusedArgs &a, int grid, bool batch, bool mixed);
int pmf_launch_fused_sb8_2_bf16(PmfDynLds *cache, hipStream_t stream, const FusedArgs &a, int grid, bool batch, bool mixed);
int pmf_launch_sb8_split(hipStream_t stream, const Sb8SplitArgs &a, int KB);
int pmf_launch_sb8_scale(hipStream_t stream, const Sb8ScaleArgs &a);
int pmf_launch_sb_split_1(hipStream_t stream, const SbSplitArgs &a);
int pmf_launch_sb_split_2(hipStream_t stream, const SbSplitArgs &a);
size_t pmf_layer_pass_lds(int KB, int lnw, int nbs);   // dynamic LDS of the layer pass (<= 160 KiB: eligible)
int pmf_launch_layer_pass(PmfDynLds *cache, hipStream_t stream, int KB, int lnw, bool mixed, int grid, const LayerPassArgs &a);   // (a.d_bf16 picks the storage variant)
int pmf_launch_layer_map(hipStream_t stream, const LayerMapArgs &m);
#endif
