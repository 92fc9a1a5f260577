// pmf_k_fused.hip -- pmf_fused_kernel for ONE (K blocks, row blocks per wave) pair, chosen at compile time
// (-DPMF_KB=.. -DPMF_RBW=..; csrc/Makefile builds the five pairs in parallel), and its launcher.
#ifndef PMF_KB
#error "compile with -DPMF_KB=<1..4> -DPMF_RBW=<1|2> -DPMF_DB=<0|1>"
#endif
#ifndef PMF_DB
#define PMF_DB 0
#endif
#include "pmf_common.h"
#include "pmf_fused.hip.inc"
template <int KB, int NW, int RBW>
static int launch_fused_t(PmfDynLds *cache, hipStream_t stream, const FusedArgs &a, int grid, bool batch, bool mixed) {
  using Cfg = FusedCfg<KB, NW, RBW>;
  const size_t lds = Cfg::lds_bytes + (batch ? Cfg::lds_batch_extra : 0);
  // gradient mode (compile-time in the kernel): 0 both, 1 grad(X) only, 2 grad(Y) only, 3 run-time flags
  int gm = a.dbg != 0 ? 3 : (a.want_gx && a.want_gy ? 0 : (a.want_gx ? 1 : (a.want_gy ? 2 : 3)));
  void (*kern)(const FusedArgs) = nullptr;
  const int bmode = !batch ? 0 : (a.btd ? 1 : 2);
  if (bmode == 2 && gm != 0) gm = 3;   // the gather fallback (> 15 batches per view) has no single-gradient variants
#define PMF_PICK_G(BM, MX) (gm == 0 ? pmf_fused_kernel<KB, NW, RBW, BM, MX, 0, PMF_DB != 0> : gm == 1 ? pmf_fused_kernel<KB, NW, RBW, BM, MX, 1, PMF_DB != 0> : \
                            gm == 2 ? pmf_fused_kernel<KB, NW, RBW, BM, MX, 2, PMF_DB != 0> : pmf_fused_kernel<KB, NW, RBW, BM, MX, 3, PMF_DB != 0>)
  if (bmode == 0) kern = mixed ? PMF_PICK_G(0, true) : PMF_PICK_G(0, false);
  else if (bmode == 1) kern = mixed ? PMF_PICK_G(1, true) : PMF_PICK_G(1, false);
  else if (gm == 0) kern = mixed ? pmf_fused_kernel<KB, NW, RBW, 2, true, 0, PMF_DB != 0> : pmf_fused_kernel<KB, NW, RBW, 2, false, 0, PMF_DB != 0>;
  else kern = mixed ? pmf_fused_kernel<KB, NW, RBW, 2, true, 3, PMF_DB != 0> : pmf_fused_kernel<KB, NW, RBW, 2, false, 3, PMF_DB != 0>;
#undef PMF_PICK_G
  PMFCHK(pmf_ensure_dyn_lds(cache, (const void *)kern, lds));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, stream, a);
  HIPCHK(hipGetLastError());
  return 0;
}


#define PMF_CAT3(a, b, c) a##b##c
#define PMF_CAT4(a, b, c, d) a##b##c##d
#if PMF_DB
#define PMF_NAME(kb, rbw) PMF_CAT4(pmf_launch_fused_exact_, kb, rbw, _bf16)
#else
#define PMF_NAME(kb, rbw) PMF_CAT3(pmf_launch_fused_exact_, kb, rbw)
#endif
int PMF_NAME(PMF_KB, PMF_RBW)(PmfDynLds *cache, hipStream_t stream, const FusedArgs &a, int grid, bool batch, bool mixed) {
  return launch_fused_t<PMF_KB, (PMF_KB <= 2 ? 8 : 4), PMF_RBW>(cache, stream, a, grid, batch, mixed);
}
