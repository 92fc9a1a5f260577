"""Per-phase cycle shares of pmf_fused_sb2_kernel from the diagnostic (-DPMF_STAMPS) build (development aid).
Build: PMF_LIB=$PWD/pathmatfac.jl_amd/libpmf_hip_stamps.so PMF_BUILD_DIR=.build_stamps pathmatfac.jl_amd/csrc/build.sh -DPMF_STAMPS
Read the SHARES: the stamps' fences forbid overlaps the real kernel has."""
import ctypes as C
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import pmf_import
pkg = pmf_import.load()
lib_path = Path(__file__).resolve().parent.parent / "pathmatfac.jl_amd" / "libpmf_hip_stamps.so"
M, N, K = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (100000, 50000, 64)))
ctx = pkg.Context(0, lib_path=lib_path)
rng = np.random.default_rng(1)
ctx.set_data_device(None, M, N)
ctx.set_factors((rng.standard_normal((K, M)) * 0.3).astype(np.float32), (rng.standard_normal((K, N)) * 0.3).astype(np.float32))
ctx.set_col_params(np.zeros(N, np.float32), np.zeros(N, np.float32)); ctx.set_batch_views([])
ctx.set_noise([(1, N)], ["normal"], np.ones(N, np.float32)); ctx.synth_data(seed=5, noise=0.1)
ctx.set_precision("bf16x3")
o = ctx.make_opts(update_X=True, update_Y=True)
for _ in range(5): ctx.epoch_begin(o)
ctx.epoch_loss(); ctx.kernel_time(reset=True)
ctx.epoch_begin(o); ctx.epoch_loss()
ms, n = ctx.kernel_time()
buf = np.zeros(16 * 8 * 1024, np.uint64)
assert ctx.lib.pmf_debug_stamps(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), buf.size) == 0
st = buf.reshape(-1, 16).astype(np.float64)
st = st[st.sum(1) > 0]
import os
if os.environ.get("PMF_SB2P", "0") == "1":   # (only with scripts/experiments/sb2_row_block_pipeline.patch applied)
    names_p = {0: "tile top", 1: "phase 1: forward A | slab reduce", 2: "phase 2: forward B | epilogue A", 3: "phase 3: GEMM2+3 A | epilogue B",
               4: "phase 4: GEMM2+3 B | loads", 10: "B2 wait", 9: "slab writes (LDS)", 5: "stage_store", 6: "B1 wait", 11: "piece prologue / flush"}
else:
    names_p = None
names = {0: "loop top", 1: "epilogue math", 2: "G split + image writes", 3: "GEMM2 (+ load issue, a3/b2 reads)", 4: "GEMM3",
         9: "slab writes (LDS)", 5: "stage_store", 6: "B1 wait", 8: "forward + slab reduce", 7: "private-slab stores",
         10: "B2 wait", 11: "piece prologue / flush"}
clk = np.median(st[:, 14] / st[:, 15]) * 100.0
tot = st[:, :14].sum()
print(f"{M}x{N} K={K}: kernel {ms:.3f} ms (stamped build), {st.shape[0]} waves, in-kernel clock {clk:.0f} MHz; cycles per wave {st[:, :14].sum(1).mean():.4g}")
ntile = (M + 255) // 256 * ((N + 31) // 32) / 256.0
if names_p:
    for q in (0, 1, 2, 3, 4, 10, 9, 5, 6, 11):
        print(f"  {names_p[q]:36s} {st[:, q].sum()/tot*100:6.2f} %   {st[:, q].mean()/ntile:7.0f} cycles per tile")
else:
    for q in (0, 1, 2, 3, 4, 9, 5, 6, 8, 7, 10, 11):
        print(f"  {names[q]:36s} {st[:, q].sum()/tot*100:6.2f} %   {st[:, q].mean()/ntile:7.0f} cycles per tile")
