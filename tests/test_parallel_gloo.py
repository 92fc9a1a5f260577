"""world_size-2 CPU test (gloo) of the row-sharded fit loop (pathmatfac.jl_amd/parallel.py).

The GPU context is replaced by a test double that implements the same step-level interface
(pmf_epoch_begin / _step_local / _step_shared / _loss of include/pmf_hip.h) on top of the fp64 CPU oracle, so the
host logic under test is the real one: sharding, asynchronous all-reduce of grad(Y), the replicated Y step, the
loss assembly (shared terms counted once) and the termination decision taken identically on every rank."""
import os
import socket
import sys
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


class OracleStepContext:
    """Step-level test double: one rank's shard evaluated by the oracle (fp64), AdaGrad as optimizers.jl:6-13."""

    def __init__(self, p, lo, hi, lr):
        import copy
        from problems import to_oracle
        q = copy.deepcopy(p)
        q["D"] = np.asfortranarray(p["D"][lo:hi])
        q["X"] = np.asfortranarray(p["X"][:, lo:hi])
        q["M"] = hi - lo
        for t in q["xreg"]:                      # group ranges over global rows -> clipped to the shard
            if t["kind"] == "group":
                s = [max(a, lo + 1) - lo for a in t["start1"]]
                e = [min(b, hi) - lo for b in t["stop1"]]
                keep = [i for i in range(len(s)) if e[i] >= s[i]]
                t["start1"], t["stop1"] = [s[i] for i in keep], [e[i] for i in keep]
                t["w"] = np.asarray(t["w"])[keep]
        self.m = to_oracle(q)
        self.lr, self.eps = lr, 1e-8
        self.accX = np.full(self.m.X.shape, self.eps)
        self.accY = np.full(self.m.Y.shape, self.eps)
        self.gY_data = np.zeros(self.m.Y.size)     # persistent buffer, like the library's device gradient

    @staticmethod
    def make_opts(**kw):
        return SimpleNamespace(**kw)

    def epoch_begin(self, o):
        m = self.m
        nx, ny = m.m.n_xreg, m.m.n_yreg
        self.loss_full, gf = m.loss_and_grads(update_X=o.update_X, update_Y=o.update_Y)
        m.m.n_xreg = 0
        self.loss_no_x, gnx = m.loss_and_grads(update_X=o.update_X, update_Y=o.update_Y)
        m.m.n_yreg = 0
        self.loss_data, gd = m.loss_and_grads(update_X=o.update_X, update_Y=o.update_Y)
        m.m.n_xreg, m.m.n_yreg = nx, ny
        self.gX_data = gd["X"].copy()
        self.gY_data[:] = gd["Y"].ravel(order="F")
        self.gX_reg, self.gY_reg = gf["X"] - gd["X"], (gf["Y"] - gd["Y"])
        self.shared = self.loss_no_x - self.loss_data          # Y regularizer: replicated on every rank

    def grad_tensor(self, which):
        import torch
        assert which == "Y"
        return torch.from_numpy(self.gY_data)                  # aliases the buffer: all_reduce works in place

    def _adagrad(self, p, g, acc):
        acc += g * g
        p -= g * (self.lr / (np.sqrt(acc) + self.eps))

    def epoch_step_local(self, o):
        if o.update_X:
            self._adagrad(self.m.X, self.gX_data + self.gX_reg, self.accX)

    def epoch_step_shared(self, o):
        if o.update_Y:
            gY = self.gY_data.reshape(self.m.Y.shape, order="F") + self.gY_reg
            self._adagrad(self.m.Y, gY, self.accY)

    def epoch_loss(self):
        return self.loss_full, self.shared


def _problem():
    from problems import make_problem
    return make_problem(M=61, N=40, K=4, seed=21, bernoulli_frac=0.25, nan_frac=0.1, weights=True, col_params=True,
                        xreg="group", yreg="fsard", random_init=True, n_groups=4)


def _worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import pmf_import
    pkg = pmf_import.load()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = _problem()
    lo, hi = pkg.parallel.shard_rows(p["M"], world, rank)
    ctx = OracleStepContext(p, lo, hi, lr=0.05)
    h = pkg.parallel.fit_distributed(ctx, dist=dist, update_X=True, update_Y=True, max_epochs=12, abs_tol=0,
                                     rel_tol=0, loss_device="cpu")
    np.savez(Path(outdir) / f"rank{rank}.npz", X=ctx.m.X, Y=ctx.m.Y, loss=h["loss"], lo=lo, hi=hi,
             term=h["term_code"], epochs=h["epochs"])
    # a learning rate that diverges must stop every rank at the same epoch with "loss_increase"
    ctx2 = OracleStepContext(p, lo, hi, lr=50.0)
    h2 = pkg.parallel.fit_distributed(ctx2, dist=dist, update_X=True, update_Y=True, max_epochs=40, abs_tol=0,
                                      rel_tol=0, loss_device="cpu")
    np.savez(Path(outdir) / f"rank{rank}_div.npz", term=h2["term_code"], epochs=h2["epochs"], loss=h2["loss"])
    dist.destroy_process_group()


def test_sharded_fit_matches_single_process_oracle(tmp_path):
    import torch.multiprocessing as mp
    from problems import to_oracle
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    p = _problem()
    ref = to_oracle(p)
    r = ref.fit(update_X=True, update_Y=True, lr=0.05, max_epochs=12, abs_tol=0, rel_tol=0)
    outs = [np.load(tmp_path / f"rank{k}.npz") for k in range(2)]
    for o in outs:
        assert str(o["term"]) == r["term_code"] and int(o["epochs"]) == r["epochs"]
        np.testing.assert_allclose(o["loss"], r["loss"], rtol=1e-11)
        np.testing.assert_allclose(o["Y"], ref.Y, rtol=1e-9, atol=1e-12)        # replicated and identical
    # (the test double derives the regularizer gradient as a difference of two oracle calls, which rounds
    #  differently per shard; the HIP path evaluates it from the replicated Y alone, bit-identically)
    np.testing.assert_allclose(outs[0]["Y"], outs[1]["Y"], rtol=1e-12, atol=1e-15)
    X = np.concatenate([o["X"] for o in outs], axis=1)
    np.testing.assert_allclose(X, ref.X, rtol=1e-9, atol=1e-12)
    assert int(outs[0]["lo"]) == 0 and int(outs[0]["hi"]) == int(outs[1]["lo"]) and int(outs[1]["hi"]) == p["M"]
    d = [np.load(tmp_path / f"rank{k}_div.npz") for k in range(2)]
    r2 = to_oracle(p).fit(update_X=True, update_Y=True, lr=50.0, max_epochs=40, abs_tol=0, rel_tol=0)
    for o in d:
        assert str(o["term"]) == "loss_increase" == r2["term_code"] and int(o["epochs"]) == r2["epochs"]
