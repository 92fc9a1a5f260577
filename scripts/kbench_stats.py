"""Timing of pmf_stats (development aid)."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import pmf_import
pkg = pmf_import.load()
M, N, K = (int(x) for x in sys.argv[1:4])
rng = np.random.default_rng(3)
ctx = pkg.Context(0)
ctx.set_data_device(None, M, N)
ctx.set_factors((rng.standard_normal((K, M)) * 0.3).astype(np.float32), (rng.standard_normal((K, N)) * 0.3).astype(np.float32))
ctx.set_col_params(np.zeros(N, np.float32), np.zeros(N, np.float32)); ctx.set_batch_views([])
ctx.set_noise([(1, N)], ["normal"], np.ones(N, np.float32)); ctx.synth_data(seed=7, noise=0.1, frac_nan=0.02)
for uf in (False, True):
    ctx.stats(use_factors=uf)
    t0 = time.time(); ctx.stats(use_factors=uf); print(f"{M}x{N} K={K} pmf_stats(use_factors={uf}): {(time.time()-t0)*1e3:.1f} ms")
