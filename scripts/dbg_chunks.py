"""debug: chunked vs unchunked single data pass on the sharded-test problem, both precisions"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import numpy as np
import pmf_import
from problems import make_problem, rel_err, shard_problem, to_context, to_oracle
pkg = pmf_import.load()
CASE = dict(M=1500, N=420, K=48, seed=31, bernoulli_frac=0.2, nan_frac=0.05, weights=True, col_params=True, n_views=2,
            batch_views=2, n_batches=6, xreg="group", yreg="fsard", random_init=True, n_groups=5, scale=0.5)
ctx = pkg.Context(0)
p0 = make_problem(**CASE)
for shard in (None, (0, 750), (750, 1500)):
    p = p0 if shard is None else shard_problem(p0, *shard)
    m = to_oracle(p)
    lo_, go = m.loss_and_grads() if hasattr(m, "loss_and_grads") else (None, None)
    for prec in ("f32", "bf16x3"):
        ctx.set_precision(prec)
        res = {}
        for ch in (0, 3):
            ctx.comm_set_chunks(ch)
            to_context(p, ctx)
            ctx.set_optimizer("adagrad", lr=0.05)
            o = ctx.make_opts(update_X=True, update_Y=True)
            ctx.epoch_begin(o)
            loss = ctx.epoch_loss()[0]
            res[ch] = (loss, ctx.get_grad("X"), ctx.get_grad("Y"))
            r = ctx.fit(update_X=True, update_Y=True, max_epochs=3, abs_tol=0, rel_tol=0)
            print(shard, prec, "chunks", ch, "pass loss", loss, "fit loss", r["loss"], "sb launches", ctx.get_precision()[1])
        print("   gX diff", rel_err(res[3][1], res[0][1]), "gY diff", rel_err(res[3][2], res[0][2]))
ctx.comm_set_chunks(0)
print("---- host-staged 1-rank communicator")
p = shard_problem(p0, 0, 750)
for prec in ("f32", "bf16x3"):
    for comm in (False, True):
        ctx.set_precision(prec)
        if comm:
            ctx.comm_init_host(0, 1, lambda arr: None)
        ctx.comm_set_chunks(3)
        to_context(p, ctx)
        ctx.set_optimizer("adagrad", lr=0.05)
        r = ctx.fit(update_X=True, update_Y=True, max_epochs=3, abs_tol=0, rel_tol=0)
        print(prec, "comm", comm, r["loss"], ctx.comm_info())
        if comm:
            ctx.comm_destroy()
