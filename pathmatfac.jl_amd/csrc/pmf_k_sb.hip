// pmf_k_sb.hip -- pmf_fused_sb_kernel and k_sb_split for ONE K-block count (-DPMF_KB=1|2), with their launchers.
#include "pmf_common.h"
#include "pmf_fused_sb.hip.inc"

#ifndef PMF_KB
#error "compile with -DPMF_KB=<1|2>"
#endif
#ifndef PMF_DB
#define PMF_DB 0
#endif
#define PMF_CAT2(a, b) a##b
#define PMF_NAME2(p, kb) PMF_CAT2(p, kb)
#if PMF_DB
#define PMF_CAT3(a, b, c) a##b##c
#define PMF_NAME3(p, kb, sfx) PMF_CAT3(p, kb, sfx)
#define PMF_SBNAME(kb) PMF_NAME3(pmf_launch_fused_sb_, kb, _bf16)
#else
#define PMF_SBNAME(kb) PMF_NAME2(pmf_launch_fused_sb_, kb)
#endif

int PMF_SBNAME(PMF_KB)(PmfDynLds *cache, hipStream_t stream, const FusedArgs &a, int grid, bool batch, bool mixed,
                                            bool want_gx, bool want_gy) {
  void (*kern)(const FusedArgs) = nullptr;
  // (the batch-layer variants exist with the per-tile noise-model dispatch only: MIXED = true also serves uniform models)
#define PMF_SB_PICK_G(MX, BT) (want_gx && want_gy ? pmf_fused_sb_kernel<PMF_KB, MX, true, true, BT, PMF_DB != 0>                                        \
                               : want_gx ? pmf_fused_sb_kernel<PMF_KB, MX, true, false, BT, PMF_DB != 0> : pmf_fused_sb_kernel<PMF_KB, MX, false, true, BT, PMF_DB != 0>)
  kern = batch ? PMF_SB_PICK_G(true, true) : (mixed ? PMF_SB_PICK_G(true, false) : PMF_SB_PICK_G(false, false));
#undef PMF_SB_PICK_G
  const size_t lds = SbCfg<PMF_KB>::lds_bytes + (batch ? SbCfg<PMF_KB>::lds_batch(a.n_bv) : 0);
  PMFCHK(pmf_ensure_dyn_lds(cache, (const void *)kern, lds));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, a);
  HIPCHK(hipGetLastError());
  return 0;
}

#if !PMF_DB
int PMF_NAME2(pmf_launch_sb_split_, PMF_KB)(hipStream_t stream, const SbSplitArgs &a) {
  const int64_t n = a.nblk * 32 * (4 * PMF_KB);   // one thread per (row, 16-B chunk)
  k_sb_split<PMF_KB><<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(a);
  HIPCHK(hipGetLastError());
  return 0;
}
#endif
