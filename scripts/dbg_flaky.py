"""two independent processes on one GPU, each repeating the same fit: any run-to-run difference is a race"""
import sys, os, subprocess
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    import pmf_import
    from problems import make_problem, shard_problem, to_context
    pkg = pmf_import.load()
    CASE = dict(M=1500, N=420, K=48, seed=31, bernoulli_frac=0.2, nan_frac=0.05, weights=True, col_params=True, n_views=2,
                batch_views=2, n_batches=6, xreg="group", yreg="fsard", random_init=True, n_groups=5, scale=0.5)
    p = shard_problem(make_problem(**CASE), 750, 1500)
    ctx = pkg.Context(0)
    mode = sys.argv[2]
    if mode == "host":
        ctx.comm_init_host(0, 1, lambda arr: None)
    ref = None
    bad = 0
    for it in range(40):
        ctx.comm_set_chunks(3)
        to_context(p, ctx)
        ctx.set_optimizer("adagrad", lr=0.05)
        r = ctx.fit(update_X=True, update_Y=True, max_epochs=4, abs_tol=0, rel_tol=0)
        if ref is None:
            ref = r["loss"]
        elif not np.array_equal(ref, r["loss"]):
            bad += 1
            print("MISMATCH", it, r["loss"], ref, r["term_code"], flush=True)
    print("child done", mode, "bad", bad, ref, flush=True)
else:
    for mode in ("none", "host"):
        procs = [subprocess.Popen([sys.executable, __file__, "child", mode]) for _ in range(2)]
        print(mode, [p.wait() for p in procs], flush=True)
