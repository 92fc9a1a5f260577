// pmf_k_layers.hip -- pmf_layer_kernel + k_layer_map (pmf_layers.hip.inc) and their launchers.
#include "pmf_common.h"
#include "pmf_layers.hip.inc"

size_t pmf_layer_pass_lds(int KB, int lnw, int nbs) {
  switch (KB) {
    case 1: return LayerCfg<1, 8>::lds(nbs);
    case 2: return lnw == 8 ? LayerCfg<2, 8>::lds(nbs) : LayerCfg<2, 4>::lds(nbs);
    case 3: return LayerCfg<3, 4>::lds(nbs);
    default: return LayerCfg<4, 4>::lds(nbs);
  }
}

int pmf_launch_layer_pass(PmfDynLds *cache, hipStream_t stream, int KB, int lnw, bool mixed, int grid, const LayerPassArgs &a) {
  void (*kern)(const LayerPassArgs) = nullptr;
  size_t lds = 0;
#define PMF_LK3(KBv, NWv, Wv) (a.d_bf16 ? (mixed ? pmf_layer_kernel<KBv, NWv, true, true, Wv> : pmf_layer_kernel<KBv, NWv, false, true, Wv>) \
                                        : (mixed ? pmf_layer_kernel<KBv, NWv, true, false, Wv> : pmf_layer_kernel<KBv, NWv, false, false, Wv>))
#define PMF_LK(KBv, NWv) (a.nbs_shift > 4 ? PMF_LK3(KBv, NWv, true) : PMF_LK3(KBv, NWv, false))
  switch (KB) {
    case 1: kern = PMF_LK(1, 8); break;
    case 2: kern = lnw == 8 ? PMF_LK(2, 8) : PMF_LK(2, 4); break;
    case 3: kern = PMF_LK(3, 4); break;
    default: kern = PMF_LK(4, 4); break;
  }
  lds = pmf_layer_pass_lds(KB, lnw, 1 << a.nbs_shift);
#undef PMF_LK
#undef PMF_LK3
  PMFCHK(pmf_ensure_dyn_lds(cache, (const void *)kern, lds));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * lnw), lds, stream, a);
  HIPCHK(hipGetLastError());
  return 0;
}

int pmf_launch_layer_map(hipStream_t stream, const LayerMapArgs &m) {
  if (m.n_parts > 1) {
    k_layer_reduce<<<(unsigned)((m.lg_stride + 255) / 256), 256, 0, stream>>>(const_cast<float2 *>(m.LG), m.lg_stride, m.n_parts);
    HIPCHK(hipGetLastError());
  }
  k_layer_map<<<(unsigned)((m.N + 255) / 256), 256, 0, stream>>>(m);
  HIPCHK(hipGetLastError());
  return 0;
}
