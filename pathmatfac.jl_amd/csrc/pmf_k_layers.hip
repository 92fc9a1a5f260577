// pmf_k_layers.hip -- pmf_layer_kernel + k_layer_map (pmf_layers.hip.inc) and their launchers.
#include "pmf_common.h"
#include "pmf_layers.hip.inc"

int pmf_launch_layer_pass(PmfDynLds *cache, hipStream_t stream, int KB, int lnw, bool mixed, int grid, const LayerPassArgs &a) {
  void (*kern)(const LayerPassArgs) = nullptr;
  size_t lds = 0;
#define PMF_LK(KBv, NWv) (a.d_bf16 ? (mixed ? pmf_layer_kernel<KBv, NWv, true, true> : pmf_layer_kernel<KBv, NWv, false, true>) \
                                   : (mixed ? pmf_layer_kernel<KBv, NWv, true, false> : pmf_layer_kernel<KBv, NWv, false, false>))
  switch (KB) {
    case 1: kern = PMF_LK(1, 8); lds = LayerCfg<1, 8>::lds_bytes; break;
    case 2:
      if (lnw == 8) { kern = PMF_LK(2, 8); lds = LayerCfg<2, 8>::lds_bytes; }
      else { kern = PMF_LK(2, 4); lds = LayerCfg<2, 4>::lds_bytes; }
      break;
    case 3: kern = PMF_LK(3, 4); lds = LayerCfg<3, 4>::lds_bytes; break;
    default: kern = PMF_LK(4, 4); lds = LayerCfg<4, 4>::lds_bytes; break;
  }
#undef PMF_LK
  PMFCHK(pmf_ensure_dyn_lds(cache, (const void *)kern, lds));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * lnw), lds, stream, a);
  HIPCHK(hipGetLastError());
  return 0;
}

int pmf_launch_layer_map(hipStream_t stream, const LayerMapArgs &m) {
  k_layer_map<<<(unsigned)((m.N + 255) / 256), 256, 0, stream>>>(m);
  HIPCHK(hipGetLastError());
  return 0;
}
