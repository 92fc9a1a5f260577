"""Fused-kernel timing of X-only epochs (transform: Y and the layers fixed; development aid)."""
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
ctx.set_noise([(1, N)], ["normal"], np.ones(N, np.float32)); ctx.synth_data(seed=7, noise=0.1)
ctx.set_optimizer("adam", lr=0.01)
for flags in (dict(update_X=True), dict(update_Y=True), dict(update_X=True, update_Y=True)):
    ctx.fit(max_epochs=2, abs_tol=0, rel_tol=0, **flags)
    ctx.kernel_time(reset=True)
    r = ctx.fit(max_epochs=5, epoch=3, abs_tol=0, rel_tol=0, **flags)
    ms, n = ctx.kernel_time()
    print(f"{M}x{N} K={K} {flags}: fused kernel {ms:.3f} ms")
