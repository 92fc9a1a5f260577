"""Per-epoch time line at the 20000 x 10000, K = 32 configuration (development aid)."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import pmf_import
pkg = pmf_import.load()
M, N, K = 20000, 10000, 32
rng = np.random.default_rng(1)
ctx = pkg.Context(0)
ctx.set_data_device(None, M, N)
ctx.set_factors((rng.standard_normal((K, M)) * 0.3).astype(np.float32), (rng.standard_normal((K, N)) * 0.3).astype(np.float32))
ctx.set_col_params(np.zeros(N, np.float32), np.zeros(N, np.float32)); ctx.set_batch_views([])
ctx.set_noise([(1, N)], ["normal"], np.ones(N, np.float32)); ctx.synth_data(seed=5, noise=0.1)
ctx.set_factors((rng.standard_normal((K, M)) * 0.1).astype(np.float32), (rng.standard_normal((K, N)) * 0.1).astype(np.float32))
ctx.clear_yreg(); ctx.add_yreg_fsard(np.full(N, 1.001, np.float32), np.full((K, N), 0.001, np.float32))
ctx.set_optimizer("adam", lr=0.01)
ctx.fit(update_X=True, update_Y=True, max_epochs=5, abs_tol=0, rel_tol=0)
ctx.kernel_time(reset=True)
t0 = time.time()
r = ctx.fit(update_X=True, update_Y=True, max_epochs=205, epoch=6, abs_tol=0, rel_tol=0)
wall = (time.time() - t0) / 200
ms, n = ctx.kernel_time()
print(f"{M}x{N} K={K}: {wall*1e3:.3f} ms/epoch wall, fused kernel {ms:.3f} ms, {1/wall:.0f} iters/s; loss {r['loss'][0]:.5g} -> {r['loss'][-1]:.5g}")
