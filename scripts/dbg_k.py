import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import pmf_import
from problems import make_problem, rel_err, to_context, to_oracle
pkg = pmf_import.load()
ctx = pkg.Context(0)
for K, M, N in ((64, 200, 150), (96, 200, 150), (100, 200, 150), (128, 140, 65), (128, 128, 32), (128, 128, 64), (128, 32, 96)):
    p = make_problem(M=M, N=N, K=K, seed=11, col_params=True)
    to_context(p, ctx)
    o = ctx.make_opts(update_X=True, update_Y=True)
    ctx.epoch_begin(o)
    loss, _ = ctx.epoch_loss()
    gX, gY = ctx.get_grad("X"), ctx.get_grad("Y")
    m = to_oracle(p)
    lo, go = m.loss_and_grads(update_X=True, update_Y=True)
    ex = np.abs(gX - go["X"]); ey = np.abs(gY - go["Y"])
    print(f"K={K} M={M} N={N}: loss rel {abs(loss-lo)/abs(lo):.2e} gX {rel_err(gX, go['X']):.2e} gY {rel_err(gY, go['Y']):.2e}; "
          f"bad gX rows(k) {np.unique(np.where(ex > 1e-3*np.abs(go['X']).max())[0])[:12]} cols(i) {np.unique(np.where(ex > 1e-3*np.abs(go['X']).max())[1])[:12]}", flush=True)
