"""A/B of two builds of the library in one process on one GPU (development aid):
python scripts/ab_libs.py libA.so libB.so [M N K]"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import pmf_import
pkg = pmf_import.load()
libs = sys.argv[1:3]
M, N, K = (int(x) for x in sys.argv[3:6]) if len(sys.argv) > 5 else (200000, 50000, 64)
rng = np.random.default_rng(1)
X0 = (rng.standard_normal((K, M)) * 0.1).astype(np.float32); Y0 = (rng.standard_normal((K, N)) * 0.1).astype(np.float32)
ctxs = []
for lp in libs:
    ctx = pkg.Context(0, lib_path=Path(lp).resolve())
    ctx.set_data_device(None, M, N)
    ctx.set_factors((rng.standard_normal((K, M)) * 0.3).astype(np.float32), (rng.standard_normal((K, N)) * 0.3).astype(np.float32))
    ctx.set_col_params(np.zeros(N, np.float32), np.zeros(N, np.float32)); ctx.set_batch_views([])
    ctx.set_noise([(1, N)], ["normal"], np.ones(N, np.float32)); ctx.synth_data(seed=5, noise=0.1)
    ctx.set_factors(X0, Y0)
    ctx.set_optimizer("adagrad", lr=0.05)
    ctxs.append(ctx)
for rnd in range(int(__import__("os").environ.get("PMF_AB_ROUNDS", "3"))):
    for lp, ctx in zip(libs, ctxs):
        ctx.fit(update_X=True, update_Y=True, max_epochs=1, abs_tol=0, rel_tol=0)
        ctx.kernel_time(reset=True)
        ctx.fit(update_X=True, update_Y=True, max_epochs=6, epoch=2, abs_tol=0, rel_tol=0)
        ms, n = ctx.kernel_time()
        print(f"round {rnd} {Path(lp).name:24s} fused kernel {ms:.3f} ms", flush=True)
