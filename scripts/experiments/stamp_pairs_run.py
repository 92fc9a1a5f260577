"""Per-phase cycles of pmf_fused_sb2p_kernel from the two-stamp diagnostic libraries built by scripts/stamp_pairs.sh
(development aid): python scripts/stamp_pairs_run.py M N K"""
import ctypes as C
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import pmf_import
pkg = pmf_import.load()
M, N, K = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (100000, 50000, 64)))
names = {(0, 1): "phase 1: forward A | slab reduce", (1, 2): "phase 2: forward B | epilogue A", (2, 3): "phase 3: GEMM2+3 A | epilogue B",
         (3, 4): "phase 4: GEMM2+3 B | loads", (4, 10): "B2 wait", (10, 9): "slab writes (LDS)", (9, 5): "stage_store", (5, 6): "B1 wait",
         (6, 0): "loop edge + tile top"}
rng = np.random.default_rng(1)
X = (rng.standard_normal((K, M)) * 0.3).astype(np.float32); Y = (rng.standard_normal((K, N)) * 0.3).astype(np.float32)
ntile = (M + 255) // 256 * ((N + 31) // 32) / 256.0
tot = 0.0
for pair, nm in names.items():
    lp = ROOT / "gpurun_scratch" / f"stamp_{pair[0]}_{pair[1]}.so"
    if not lp.exists():
        print(f"{nm}: {lp} missing"); continue
    ctx = pkg.Context(0, lib_path=lp)
    ctx.set_data_device(None, M, N)
    ctx.set_factors(X, Y)
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
    st = st[st[:, 14] > 0]
    clk = np.median(st[:, 14] / st[:, 15]) * 100.0
    cyc = st[:, 1].mean() / ntile
    tot += cyc
    print(f"  {nm:36s} {cyc:7.0f} cycles per tile   (kernel {ms:.3f} ms, {clk:.0f} MHz, whole kernel {st[:, 14].mean()/ntile:.0f} cycles per tile)", flush=True)
    ctx.close()
print(f"  sum {tot:.0f}")
