// pmf_k_sb8.hip -- pmf_fused_sb8_kernel (tall row panel, f16-pair forward, cross-wave GEMM3) for one K-block count
// (-DPMF_KB=4: 96 < K <= 128, 256-row panel; -DPMF_KB=2: 32 < K <= 64, 512-row panel) and one storage type of D
// (-DPMF_DB=0|1), its operand-image kernels and launchers.
#ifndef PMF_DB
#define PMF_DB 0
#endif
#ifndef PMF_KB
#define PMF_KB 4
#endif
#include "pmf_common.h"
#include "pmf_fused_sb8.hip.inc"

#define PMF_CAT_(a, b, c) a##b##c
#define PMF_CAT(a, b, c) PMF_CAT_(a, b, c)
#if PMF_DB
#define PMF_SB8NAME PMF_CAT(pmf_launch_fused_sb8_, PMF_KB, _bf16)
#else
#define PMF_SB8NAME PMF_CAT(pmf_launch_fused_sb8_, PMF_KB, )
#endif

int PMF_SB8NAME(PmfDynLds *cache, hipStream_t stream, const FusedArgs &a, int grid, bool batch, bool mixed) {
  void (*kern)(const FusedArgs) = batch ? pmf_fused_sb8_kernel<PMF_KB, true, true, PMF_DB != 0>
                                        : (mixed ? pmf_fused_sb8_kernel<PMF_KB, true, false, PMF_DB != 0> : pmf_fused_sb8_kernel<PMF_KB, false, false, PMF_DB != 0>);
  const size_t lds = Sb8Cfg<PMF_KB>::lds_bytes + (batch ? Sb8Cfg<PMF_KB>::lds_batch(a.n_bv) : 0);
  if (lds > 160 * 1024) return pmf_fail("pmf_fused_sb8_kernel: %zu bytes of LDS", lds);
  PMFCHK(pmf_ensure_dyn_lds(cache, (const void *)kern, lds));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
  HIPCHK(hipGetLastError());
  return 0;
}

#if !PMF_DB && PMF_KB == 4
int pmf_launch_sb8_scale(hipStream_t stream, const Sb8ScaleArgs &a) {
  HIPCHK(hipMemsetAsync(a.max_bits, 0, sizeof(uint32_t), stream));
  const int64_t n4 = a.n * (a.Kp / 4);
  const int grid = (int)(n4 / 256 / 8 > 1024 ? 1024 : (n4 / 256 / 8 < 1 ? 1 : n4 / 256 / 8));
  hipLaunchKernelGGL(k_sb8_absmax, dim3(grid), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(k_sb8_scale_fin, dim3(1), dim3(1), 0, stream, a);
  HIPCHK(hipGetLastError());
  return 0;
}
int pmf_launch_sb8_split(hipStream_t stream, const Sb8SplitArgs &a, int KB) {
  if (a.nblk <= 0) return 0;
  if (KB == 4) hipLaunchKernelGGL(k_sb8_split<4>, dim3((unsigned)a.nblk), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(k_sb8_split<2>, dim3((unsigned)a.nblk), dim3(256), 0, stream, a);
  HIPCHK(hipGetLastError());
  return 0;
}
#endif
