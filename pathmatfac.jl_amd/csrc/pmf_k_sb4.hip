// pmf_k_sb4.hip -- pmf_fused_sb4_kernel (64 < K <= 128) for one K-block count and one storage type of D
// (-DPMF_KB=4 -DPMF_DB=0|1), k_sb4_split, and their launchers.
#ifndef PMF_KB
#error "compile with -DPMF_KB=<3|4> -DPMF_DB=<0|1>"
#endif
#ifndef PMF_DB
#define PMF_DB 0
#endif
#include "pmf_common.h"
#include "pmf_fused_sb4.hip.inc"

#define PMF_CAT2(a, b) a##b
#define PMF_NAME2(p, kb) PMF_CAT2(p, kb)
#if PMF_DB
#define PMF_CAT3(a, b, c) a##b##c
#define PMF_NAME3(p, kb, sfx) PMF_CAT3(p, kb, sfx)
#define PMF_SB4NAME(kb) PMF_NAME3(pmf_launch_fused_sb4_, kb, _bf16)
#else
#define PMF_SB4NAME(kb) PMF_NAME2(pmf_launch_fused_sb4_, kb)
#endif

int PMF_SB4NAME(PMF_KB)(PmfDynLds *cache, hipStream_t stream, const FusedArgs &a, int grid, bool batch, bool mixed, bool want_gx,
                        bool want_gy) {
  void (*kern)(const FusedArgs) = nullptr;
#define PMF_SB_PICK_G(MX, BT) (want_gx && want_gy ? pmf_fused_sb4_kernel<PMF_KB, MX, true, true, BT, PMF_DB != 0>                                   \
                               : want_gx ? pmf_fused_sb4_kernel<PMF_KB, MX, true, false, BT, PMF_DB != 0>                                          \
                                         : pmf_fused_sb4_kernel<PMF_KB, MX, false, true, BT, PMF_DB != 0>)
  kern = batch ? PMF_SB_PICK_G(true, true) : (mixed ? PMF_SB_PICK_G(true, false) : PMF_SB_PICK_G(false, false));
#undef PMF_SB_PICK_G
  const size_t lds = Sb4Cfg<PMF_KB>::lds_bytes + (batch ? Sb4Cfg<PMF_KB>::lds_batch(a.n_bv) : 0);
  PMFCHK(pmf_ensure_dyn_lds(cache, (const void *)kern, lds));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
  HIPCHK(hipGetLastError());
  return 0;
}

#if !PMF_DB && PMF_KB == 4   // (one copy: see k_sb4_split)
int pmf_launch_sb4_split(hipStream_t stream, const Sb4SplitArgs &a) {
  if (a.nblk <= 0) return 0;
  k_sb4_split<<<(unsigned)a.nblk, 256, 0, stream>>>(a);
  HIPCHK(hipGetLastError());
  return 0;
}
#endif
