"""A/B of fused-kernel variants in one process (PMF_NW env is read at every launch)."""
import os, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import pmf_import
pkg = pmf_import.load()
for (M, N, K) in ((50000, 50000, 64), (20000, 10000, 32), (50000, 50000, 32)):
    ctx = pkg.Context(0)
    rng = np.random.default_rng(1)
    ctx.set_data_device(None, M, N)
    ctx.set_factors((rng.standard_normal((K, M)) * 0.3).astype(np.float32), (rng.standard_normal((K, N)) * 0.3).astype(np.float32))
    ctx.set_col_params(np.zeros(N, np.float32), np.zeros(N, np.float32)); ctx.set_batch_views([])
    ctx.set_noise([(1, N)], ["normal"], np.ones(N, np.float32)); ctx.synth_data(seed=5, noise=0.1)
    o = ctx.make_opts(update_X=True, update_Y=True)
    res = {}
    for rnd in range(3):
        for nw in ("8", "4"):
            os.environ["PMF_NW"] = nw
            for _ in range(2): ctx.epoch_begin(o)
            ctx.epoch_loss(); ctx.kernel_time(reset=True)
            for _ in range(5): ctx.epoch_begin(o)
            loss, _ = ctx.epoch_loss()
            ms, n = ctx.kernel_time()
            res.setdefault(nw, []).append(ms)
    gX = {}
    for nw in ("8", "4"):
        os.environ["PMF_NW"] = nw
        ctx.epoch_begin(o); l, _ = ctx.epoch_loss(); gX[nw] = (l, ctx.get_grad("Y"))
    print(f"M={M} N={N} K={K}: " + "  ".join(f"NW={nw}: {min(v):.3f} ms ({6.0*M*N*K/min(v)/1e9:.1f} TF/s)" for nw, v in res.items()),
          f"| loss rel diff {abs(gX['8'][0]-gX['4'][0])/abs(gX['8'][0]):.1e} gY rel diff {np.abs(gX['8'][1]-gX['4'][1]).max()/np.abs(gX['8'][1]).max():.1e}", flush=True)
    ctx.close()
