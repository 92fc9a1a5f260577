"""Per-phase cycle shares of the fused kernel from the diagnostic (-DPMF_STAMPS) build.  Read the SHARES, not the
absolute time: the stamps' fences forbid overlaps the real kernel has."""
import ctypes as C
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import pmf_import
pkg = pmf_import.load()
lib_path = Path(__file__).resolve().parent.parent / "pathmatfac.jl_amd" / "libpmf_hip_stamps.so"
M, N, K = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (40000, 20000, 64)))
ctx = pkg.Context(0, lib_path=lib_path)
rng = np.random.default_rng(1)
ctx.set_data_device(None, M, N)
ctx.set_factors((rng.standard_normal((K, M)) * 0.3).astype(np.float32), (rng.standard_normal((K, N)) * 0.3).astype(np.float32))
ctx.set_col_params(np.zeros(N, np.float32), np.zeros(N, np.float32)); ctx.set_batch_views([])
ctx.set_noise([(1, N)], ["normal"], np.ones(N, np.float32)); ctx.synth_data(seed=5, noise=0.1)
import os
if os.environ.get("PMF_ZERO") == "1":   # all-zero operands and data: what the clock does when the datapaths do not toggle
    ctx.set_factors(np.zeros((K, M), np.float32), np.zeros((K, N), np.float32)); ctx.synth_data(seed=5, noise=0.0)
o = ctx.make_opts(update_X=True, update_Y=True)
for _ in range(int(sys.argv[4]) if len(sys.argv) > 4 else 3): ctx.epoch_begin(o)   # >= 2 s of launches before reading the clock
ctx.epoch_loss(); ctx.kernel_time(reset=True)
ctx.epoch_begin(o); ctx.epoch_loss()
ms, n = ctx.kernel_time()
buf = np.zeros(16 * 8 * 256, np.uint64)
assert ctx.lib.pmf_debug_stamps(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), buf.size) == 0
st = buf.reshape(-1, 16).astype(np.float64)
st = st[st.sum(1) > 0]
names = ["loop/B2 exit", "epilogue", "gY slab prefetch issue (split kernel: G split + image writes)", "GEMM2", "GEMM3+slab wr", "stage_store", "B1 wait", "reduce+slab store",
         "forward", "B2 wait", "(pre-flush)", "macro prologue/flush", "  Y/colp load issue", "  D load issue"]
clk = np.median(st[:, 14] / st[:, 15]) * 100.0
st = st[:, :14]
tot = st.sum()
print(f'in-kernel clock (s_memtime / s_memrealtime x 100 MHz, median over waves): {clk:.0f} MHz')
print(f"{M}x{N} K={K}: kernel {ms:.3f} ms (stamped build), {st.shape[0]} waves; cycles per wave {st.sum(1).mean():.3g}")
for q, nm in enumerate(names):
    print(f"  {nm:24s} {st[:, q].sum()/tot*100:6.2f} %")
# per-wave-slot view: do the two waves of a SIMD (w and w+4?) behave differently?
full = buf.reshape(-1, 8, 16).astype(np.float64)
full = full[full[:, :, :14].sum((1, 2)) > 0]
print("wave  total(Mcyc)  barrier%  epi0%  GEMM2%  GEMM3%  stage%  Yhalf%  loop%")
for w in range(8):
    r = full[:, w, :14]
    t = r.sum()
    print(f"  {w}   {r.sum(1).mean()/1e6:8.2f}   {r[:, 6].sum()/t*100:6.2f}  {r[:, 1].sum()/t*100:6.2f}  {r[:, 3].sum()/t*100:6.2f}  {r[:, 4].sum()/t*100:6.2f}  {r[:, 5].sum()/t*100:6.2f}  {r[:, 8].sum()/t*100:6.2f}  {r[:, 0].sum()/t*100:6.2f}")
tot14 = buf.reshape(-1, 16).astype(np.float64)
tot14 = tot14[tot14[:, :14].sum(1) > 0][:, 14]
print(f"whole-kernel shader clocks per wave: mean {tot14.mean():.4g}  min {tot14.min():.4g}  max {tot14.max():.4g}  -> max = {tot14.max()/clk/1e3:.3f} ms at the in-kernel clock")
perwg = buf.reshape(-1, 8, 16).astype(np.float64)[:, 0, 14]
perwg = perwg[perwg > 0]
srt = np.sort(perwg)
print("per-workgroup clocks (wave 0) percentiles 0/25/50/75/100:", [f"{np.percentile(srt, q):.4g}" for q in (0, 25, 50, 75, 100)])
