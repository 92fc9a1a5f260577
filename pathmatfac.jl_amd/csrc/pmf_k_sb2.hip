// pmf_k_sb2.hip -- pmf_fused_sb2_kernel (32 < K <= 64; four waves x two row blocks) for one storage type of D
// (-DPMF_DB=0|1) and its launcher.
#ifndef PMF_DB
#define PMF_DB 0
#endif
#include "pmf_common.h"
#include "pmf_fused_sb2.hip.inc"

#if PMF_DB
#define PMF_SB2NAME pmf_launch_fused_sb2_bf16
#else
#define PMF_SB2NAME pmf_launch_fused_sb2
#endif

int PMF_SB2NAME(PmfDynLds *cache, hipStream_t stream, const FusedArgs &a, int grid, bool batch, bool mixed, bool want_gx, bool want_gy) {
  void (*kern)(const FusedArgs) = nullptr;
#define PMF_SB_PICK_G(MX, BT) (want_gx && want_gy ? pmf_fused_sb2_kernel<MX, true, true, BT, PMF_DB != 0>                                   \
                               : want_gx ? pmf_fused_sb2_kernel<MX, true, false, BT, PMF_DB != 0>                                          \
                                         : pmf_fused_sb2_kernel<MX, false, true, BT, PMF_DB != 0>)
  kern = batch ? PMF_SB_PICK_G(true, true) : (mixed ? PMF_SB_PICK_G(true, false) : PMF_SB_PICK_G(false, false));
#undef PMF_SB_PICK_G
  const size_t lds = Sb2Cfg::lds_bytes + (batch ? Sb2Cfg::lds_batch(a.n_bv) : 0);
  PMFCHK(pmf_ensure_dyn_lds(cache, (const void *)kern, lds));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
  HIPCHK(hipGetLastError());
  return 0;
}
