"""Fused-kernel timing of the general path (development aid): mixed Gaussian / Bernoulli columns, two batch views
(scale + shift per (batch, column)), missing entries.  python scripts/kbench_mixed.py M N K"""
import sys
import time
from pathlib import Path
import numpy as np
import os
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import pmf_import
pkg = pmf_import.load()
M, N, K = (int(x) for x in sys.argv[1:4])
mode = sys.argv[4] if len(sys.argv) > 4 else "all"   # all | batch | mixed | nan
rng = np.random.default_rng(3)
ctx = pkg.Context(0, lib_path=(Path(os.environ["PMF_LIB"]).resolve() if os.environ.get("PMF_LIB") else None))
ctx.set_data_device(None, M, N)
ctx.set_factors((rng.standard_normal((K, M)) * 0.3).astype(np.float32), (rng.standard_normal((K, N)) * 0.3).astype(np.float32))
ctx.set_col_params((rng.standard_normal(N) * 0.1).astype(np.float32), rng.standard_normal(N).astype(np.float32))
nb = int(os.environ.get("PMF_NB", "8"))   # batches per view (rows sorted by batch)
h = N // 2
views = []
for (s, e) in ((1, h), (h + 1, N)):
    nv = e - s + 1
    views.append(dict(start1=s, stop1=e, batch_of_row=np.sort(rng.integers(0, nb, M)).astype(np.int32),
                      logdelta=(0.1 * rng.standard_normal((nb, nv))).astype(np.float32),
                      theta=(0.1 * rng.standard_normal((nb, nv))).astype(np.float32)))
ctx.set_batch_views(views if mode in ("all", "batch") else [])
nbern = int(sys.argv[5]) if len(sys.argv) > 5 else N // 5
if mode in ("all", "mixed"):
    if nbern >= N:   # every column Bernoulli: the per-tile cost of that noise model without any imbalance
        ctx.set_noise([(1, N)], ["bernoulli"], np.ones(N, np.float32))
    else:
        ctx.set_noise([(1, nbern), (nbern + 1, N)], ["bernoulli", "normal"], np.ones(N, np.float32))
else:
    ctx.set_noise([(1, N)], ["normal"], np.ones(N, np.float32))
ctx.synth_data(seed=7, noise=0.1, frac_nan=0.05 if mode in ("all", "nan") else 0.0)
ctx.set_factors((rng.standard_normal((K, M)) * 0.1).astype(np.float32), (rng.standard_normal((K, N)) * 0.1).astype(np.float32))
ctx.set_optimizer("adam", lr=0.01)
r = ctx.fit(update_X=True, update_Y=True, max_epochs=2, abs_tol=0, rel_tol=0)
ctx.kernel_time(reset=True)
t0 = time.time()
r = ctx.fit(update_X=True, update_Y=True, max_epochs=5, epoch=3, abs_tol=0, rel_tol=0)
wall = (time.time() - t0) / 3
ms, n = ctx.kernel_time()
fl = 6.0 * M * N * K
print(f"[{mode}] {M}x{N} K={K}: epoch wall {wall*1e3:.2f} ms; fused kernel {ms:.3f} ms x{n} -> {fl/ms/1e9:.1f} TF/s "
      f"({fl/ms/1e9/157.3*100:.1f}% of f32 MFMA peak); loss {r['loss'][0]:.5g} -> {r['loss'][-1]:.5g} {r['term_code']} nb={nb} path={ctx.last_path()} kernel family {ctx.last_kernel()} precision {ctx.get_precision()}")
