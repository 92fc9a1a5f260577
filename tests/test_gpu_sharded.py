"""Two ranks of the row-sharded fit loop (pathmatfac.jl_amd/parallel.py) on REAL HIP contexts, world_size 2 on one GPU.

tests/test_parallel_gloo.py covers the host logic with an oracle test double on the CPU; bench.py exercises RCCL with a
one-rank group.  Here both ranks run the library itself (each its own pmf context on cuda:0, its own row shard of D and
X, replicated Y and layers), all-reduce the library's grad(Y) buffer and the loss through torch.distributed (gloo: two
processes on one device cannot form an RCCL ring) and must reproduce the single-context fit and the fp64 oracle."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from problems import make_problem, rel_err, shard_problem, to_context, to_oracle

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent

CASE = dict(M=1500, N=420, K=48, seed=31, bernoulli_frac=0.2, nan_frac=0.05, weights=True, col_params=True, n_views=2,
            batch_views=2, n_batches=6, xreg="group", yreg="fsard", random_init=True, n_groups=5, scale=0.5)
EPOCHS, LR = int(os.environ.get("PMF_SHARD_EPOCHS", "8")), 0.05


def _flags(mode):
    return dict(update_X=True, update_Y=True) if mode == "factors" else dict(update_col_layers=True)


def _layer_params(ctx, n_views):
    ls, mu = ctx.get_col_params()
    out = [ls, mu]
    for v in range(n_views):
        out += list(ctx.get_batch_view(v))
    return np.concatenate([np.asarray(a, np.float64).ravel() for a in out])


def _worker(rank, world, port, outdir, precision, mode="factors"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import pmf_import
    pkg = pmf_import.load()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    p = make_problem(**CASE)
    lo, hi = pkg.parallel.shard_rows(p["M"], world, rank)
    ctx = pkg.Context(0)
    # (no set_stream here on purpose: fit_distributed must adopt torch's current stream itself when world_size > 1)
    ctx.set_precision(precision)
    to_context(shard_problem(p, lo, hi), ctx)
    ctx.set_optimizer("adagrad", lr=LR)
    h = pkg.parallel.fit_distributed(ctx, dist=dist, max_epochs=EPOCHS, abs_tol=0, rel_tol=0, **_flags(mode))
    X, Y = ctx.get_factors()
    np.savez(Path(outdir) / f"rank{rank}.npz", X=X, Y=Y, loss=h["loss"], lo=lo, hi=hi, term=h["term_code"],
             epochs=h["epochs"], split_launches=ctx.get_precision()[1], layers=_layer_params(ctx, len(p["batch_views"])))
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_two_rank_sharded_hip_fit_matches_single_context_and_oracle(ctx, tmp_path, precision):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    code = ("import sys; sys.path[:0] = [%r, %r]; import test_gpu_sharded as t; "
            "t._worker(int(sys.argv[1]), 2, int(sys.argv[2]), sys.argv[3], sys.argv[4], *sys.argv[5:])") % (str(ROOT), str(ROOT / "tests"))
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), str(port), str(tmp_path), precision]) for r in range(2)]
    for pr in procs:
        assert pr.wait(timeout=600) == 0
    outs = [np.load(tmp_path / f"rank{k}.npz") for k in range(2)]
    p = make_problem(**CASE)
    # single context, whole matrix
    ctx.set_precision(precision)
    try:
        to_context(p, ctx)
        ctx.set_optimizer("adagrad", lr=LR)
        r1 = ctx.fit(update_X=True, update_Y=True, max_epochs=EPOCHS, abs_tol=0, rel_tol=0)
        X1, Y1 = ctx.get_factors()
    finally:
        ctx.set_precision("f32")
    m = to_oracle(p)
    ro = m.fit(update_X=True, update_Y=True, lr=LR, max_epochs=EPOCHS, abs_tol=0, rel_tol=0)
    assert int(outs[0]["lo"]) == 0 and int(outs[0]["hi"]) == int(outs[1]["lo"]) and int(outs[1]["hi"]) == p["M"]
    X = np.concatenate([o["X"] for o in outs], axis=1)
    for o in outs:
        assert str(o["term"]) == r1["term_code"] == ro["term_code"] and int(o["epochs"]) == r1["epochs"] == ro["epochs"]
        np.testing.assert_allclose(o["loss"], ro["loss"], rtol=5e-5)
        np.testing.assert_allclose(o["loss"], r1["loss"], rtol=2e-5)
        if precision == "bf16x3":
            assert int(o["split_launches"]) == EPOCHS
    np.testing.assert_array_equal(outs[0]["Y"], outs[1]["Y"])      # the replicated Y stays bit-identical across ranks
    assert rel_err(outs[0]["Y"], Y1) <= 2e-4 and rel_err(X, X1) <= 2e-4, (rel_err(outs[0]["Y"], Y1), rel_err(X, X1))
    assert rel_err(outs[0]["Y"], m.Y) <= 2e-3 and rel_err(X, m.X) <= 2e-3


def test_two_rank_sharded_layer_stage_matches_single_context(ctx, tmp_path):
    """The column / batch layer stage (update_col_layers: mu, theta, logdelta trained, X and Y fixed): the layer
    gradients of the two row shards are summed through the all-reduce (a row batch spans both ranks), and every rank
    must end with the single-context parameters."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    code = ("import sys; sys.path[:0] = [%r, %r]; import test_gpu_sharded as t; "
            "t._worker(int(sys.argv[1]), 2, int(sys.argv[2]), sys.argv[3], sys.argv[4], *sys.argv[5:])") % (str(ROOT), str(ROOT / "tests"))
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), str(port), str(tmp_path), "f32", "layers"]) for r in range(2)]
    for pr in procs:
        assert pr.wait(timeout=600) == 0
    outs = [np.load(tmp_path / f"rank{k}.npz") for k in range(2)]
    p = make_problem(**CASE)
    to_context(p, ctx)
    ctx.set_optimizer("adagrad", lr=LR)
    r1 = ctx.fit(update_col_layers=True, max_epochs=EPOCHS, abs_tol=0, rel_tol=0)
    ref = _layer_params(ctx, len(p["batch_views"]))
    for o in outs:
        assert str(o["term"]) == r1["term_code"] and int(o["epochs"]) == r1["epochs"]
        np.testing.assert_allclose(o["loss"], r1["loss"], rtol=2e-5)
        assert rel_err(o["layers"], ref) <= 2e-4, rel_err(o["layers"], ref)
    np.testing.assert_array_equal(outs[0]["layers"], outs[1]["layers"])
