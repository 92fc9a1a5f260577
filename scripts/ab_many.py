"""A/B of several builds of the library on one GPU in one process (development aid):
PMF_PRECISION=bf16x3 python scripts/ab_many.py M N K lib1.so lib2.so ..."""
import os
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import pmf_import
pkg = pmf_import.load()
M, N, K = (int(x) for x in sys.argv[1:4])
libs = sys.argv[4:]
store = os.environ.get("PMF_AB_STORE", "f32")
rng = np.random.default_rng(1)
X0 = (rng.standard_normal((K, M)) * 0.1).astype(np.float32); Y0 = (rng.standard_normal((K, N)) * 0.1).astype(np.float32)
Xt = (rng.standard_normal((K, M)) * 0.3).astype(np.float32); Yt = (rng.standard_normal((K, N)) * 0.3).astype(np.float32)
zero = os.environ.get("PMF_AB_ZERO") == "1"     # all-zero operands and data: the same instruction stream at the least switching power
if zero:
    X0 *= 0; Y0 *= 0; Xt *= 0; Yt *= 0
res = {}
for rnd in range(int(os.environ.get("PMF_AB_ROUNDS", "2"))):
    for lp in libs:
        ctx = pkg.Context(0, lib_path=Path(lp).resolve())
        ctx.set_data_device(None, M, N, store=store)
        ctx.set_factors(Xt, Yt)
        ctx.set_col_params(np.zeros(N, np.float32), np.zeros(N, np.float32)); ctx.set_batch_views([])
        ctx.set_noise([(1, N)], ["normal"], np.ones(N, np.float32)); ctx.synth_data(seed=5, noise=0.0 if zero else 0.1)
        ctx.set_factors(X0, Y0)
        ctx.set_optimizer("adagrad", lr=0.05)
        ctx.fit(update_X=True, update_Y=True, max_epochs=1, abs_tol=0, rel_tol=0)
        ctx.kernel_time(reset=True)
        ctx.fit(update_X=True, update_Y=True, max_epochs=5, epoch=2, abs_tol=0, rel_tol=0)
        ms, n = ctx.kernel_time()
        res.setdefault(lp, []).append(ms)
        ctx.close()
for lp in libs:
    print(f"{Path(lp).name:28s} fused kernel " + " ".join(f"{m:8.3f}" for m in res[lp]) + " ms", flush=True)
