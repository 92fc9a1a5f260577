"""Timing of a layer-parameter epoch (update_col_layers: mu / logsigma / theta / logdelta gradients; development aid)."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import pmf_import
pkg = pmf_import.load()
M, N, K = (int(x) for x in sys.argv[1:4])
rng = np.random.default_rng(3)
import os
ctx = pkg.Context(0, lib_path=(Path(os.environ["PMF_LIB"]).resolve() if os.environ.get("PMF_LIB") else None))
ctx.set_data_device(None, M, N)
ctx.set_factors((rng.standard_normal((K, M)) * 0.3).astype(np.float32), (rng.standard_normal((K, N)) * 0.3).astype(np.float32))
ctx.set_col_params((rng.standard_normal(N) * 0.1).astype(np.float32), rng.standard_normal(N).astype(np.float32))
nb, h = int(os.environ.get("PMF_NB", "8")), N // 2
views = []
for (s, e) in ((1, h), (h + 1, N)):
    nv = e - s + 1
    views.append(dict(start1=s, stop1=e, batch_of_row=np.sort(rng.integers(0, nb, M)).astype(np.int32),
                      logdelta=(0.1 * rng.standard_normal((nb, nv))).astype(np.float32),
                      theta=(0.1 * rng.standard_normal((nb, nv))).astype(np.float32)))
ctx.set_batch_views(views)
ctx.set_noise([(1, N)], ["normal"], np.ones(N, np.float32))
ctx.synth_data(seed=7, noise=0.1, frac_nan=0.02)
ctx.set_optimizer("adagrad", lr=0.1)
for flags in (dict(update_col_layers=True), dict(update_X=True, update_Y=True, update_col_layers=True)):
    r = ctx.fit(max_epochs=2, abs_tol=0, rel_tol=0, **flags)
    t0 = time.time()
    r = ctx.fit(max_epochs=5, epoch=3, abs_tol=0, rel_tol=0, **flags)
    print(f"{M}x{N} K={K} {flags}: {(time.time()-t0)/3*1e3:.2f} ms/epoch; loss {r['loss'][0]:.5g} -> {r['loss'][-1]:.5g} nb={nb} path={ctx.last_path()}")
