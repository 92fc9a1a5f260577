"""The library-native exchange of the row-sharded fit (pmf_comm_init / pmf_comm_init_host, include/pmf_hip.h "Multi-GPU")
and the pipelined, chunked epoch loop of pmf_fit behind it.

* one-rank RCCL communicator: the collectives really run (librccl is loaded, ncclAllReduce on the library's buffers on
  its communication stream) and the fit must be BIT-identical to the fit without a communicator;
* forced column chunks on one GPU: same results as the unchunked pass (to rounding: the summation order of gY / gX over
  the row panels changes with the work split) and bitwise reproducible run to run;
* two ranks on REAL HIP contexts sharing the test box's one GPU, through the C epoch loop itself, with the host-staged
  transport (gloo on the CPU moves the bytes: two processes on one device cannot form an RCCL ring): identical
  termination, losses and replicated Y on both ranks, parameters equal to the single-context fit and the fp64 oracle;
* grad(X) is bitwise reproducible (fixed-order k_gx_reduce instead of float atomics)."""
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
WIDE = dict(M=2100, N=2600, K=64, seed=5, xreg="group", yreg="fsard", random_init=True, scale=0.5, weights=True)
EPOCHS, LR = 8, 0.05


def _fit(ctx, p, opt="adagrad", **kw):
    to_context(p, ctx)
    ctx.set_optimizer(opt, lr=LR if opt == "adagrad" else 0.01)
    r = ctx.fit(max_epochs=EPOCHS, abs_tol=0, rel_tol=0, **kw)
    X, Y = ctx.get_factors()
    return r, X, Y


@pytest.mark.parametrize("opt", ["adagrad", "adam"])
def test_one_rank_rccl_communicator_is_bit_identical(pkg, ctx, opt):
    p = make_problem(**CASE)
    r0, X0, Y0 = _fit(ctx, p, opt, update_X=True, update_Y=True)
    ctx.comm_init(0, 1, pkg._lib.comm_unique_id())
    try:
        info = ctx.comm_info()
        assert info["transport"] == "rccl" and info["nranks"] == 1
        r1, X1, Y1 = _fit(ctx, p, opt, update_X=True, update_Y=True)
        n1 = ctx.comm_info()["n_collectives"]
        assert n1 >= 2 * EPOCHS, n1          # grad(Y) + the loss, every epoch (plus the speculative passes)
        np.testing.assert_array_equal(r1["loss"], r0["loss"])
        np.testing.assert_array_equal(X1, X0)
        np.testing.assert_array_equal(Y1, Y0)
        # the layer stage exchanges the four layer gradients too
        to_context(p, ctx)
        ctx.set_optimizer("adagrad", lr=LR)
        rl = ctx.fit(update_col_layers=True, frozen_layers=0b0001, max_epochs=4, abs_tol=0, rel_tol=0)
        assert ctx.comm_info()["n_collectives"] > n1 + 4
    finally:
        ctx.comm_destroy()
    to_context(p, ctx)
    ctx.set_optimizer("adagrad", lr=LR)
    rl0 = ctx.fit(update_col_layers=True, frozen_layers=0b0001, max_epochs=4, abs_tol=0, rel_tol=0)
    np.testing.assert_allclose(rl["loss"], rl0["loss"], rtol=1e-8)   # (the layer pass sums with float atomics)


def _grads(ctx, p):
    to_context(p, ctx)
    ctx.set_optimizer("adagrad", lr=LR)
    o = ctx.make_opts(update_X=True, update_Y=True)
    ctx.epoch_begin(o)
    ctx.epoch_step_local(ctx.make_opts())     # (no-op steps: flags off)
    loss = ctx.epoch_loss()[0]
    return loss, ctx.get_grad("X"), ctx.get_grad("Y")


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
@pytest.mark.parametrize("chunks", [2, 5])
def test_chunked_data_pass_matches_unchunked_and_oracle(ctx, precision, chunks):
    p = make_problem(**WIDE)
    ctx.set_precision(precision)
    try:
        l0, gX0, gY0 = _grads(ctx, p)
        r0, X0, Y0 = _fit(ctx, p, update_X=True, update_Y=True)
        ctx.comm_set_chunks(chunks)
        l1, gX1, gY1 = _grads(ctx, p)
        r1, X1, Y1 = _fit(ctx, p, update_X=True, update_Y=True)
        assert ctx.comm_info()["n_chunks"] == chunks
        r2, X2, Y2 = _fit(ctx, p, update_X=True, update_Y=True)
    finally:
        ctx.comm_set_chunks(0)
        ctx.set_precision("f32")
    # one data pass: same loss and gradients up to the order the row panels / pieces are summed in
    assert abs(l1 - l0) <= 1e-7 * abs(l0)
    assert rel_err(gX1, gX0) <= 2e-6 and rel_err(gY1, gY0) <= 2e-6, (rel_err(gX1, gX0), rel_err(gY1, gY0))
    np.testing.assert_array_equal(r2["loss"], r1["loss"])       # run to run: bitwise
    np.testing.assert_array_equal(X2, X1)
    np.testing.assert_array_equal(Y2, Y1)
    # the fits: AdaGrad's g / sqrt(acc) amplifies rounding on near-zero gradients -- the parity tolerance applies
    np.testing.assert_allclose(r1["loss"], r0["loss"], rtol=5e-6)
    assert rel_err(X1, X0) <= 2e-3 and rel_err(Y1, Y0) <= 2e-3, (rel_err(X1, X0), rel_err(Y1, Y0))
    m = to_oracle(p)
    ro = m.fit(update_X=True, update_Y=True, lr=LR, max_epochs=EPOCHS, abs_tol=0, rel_tol=0)
    np.testing.assert_allclose(r1["loss"], ro["loss"], rtol=5e-5)
    if precision == "f32":
        assert rel_err(X1, m.X) <= 2e-3 and rel_err(Y1, m.Y) <= 2e-3, (rel_err(X1, m.X), rel_err(Y1, m.Y))
    # (split-bf16: the chunked and the unchunked fit agree above; against the oracle the first AdaGrad steps,
    #  +-lr*sign(g), turn its 4e-6 gradient-product error into sign flips of near-zero gradients on a problem this size --
    #  tests/test_gpu_split_bf16.py holds the kernel's own oracle comparisons)


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_grad_x_and_fit_are_bitwise_reproducible(ctx, precision):
    p = make_problem(M=3000, N=1900, K=64, seed=9, xreg="l2", yreg="fsard", random_init=True, nan_frac=0.03, scale=0.5)
    ctx.set_precision(precision)
    try:
        outs = []
        for _ in range(2):
            to_context(p, ctx)
            ctx.set_optimizer("adagrad", lr=LR)
            o = ctx.make_opts(update_X=True, update_Y=True)
            ctx.epoch_begin(o)
            gX, gY = ctx.get_grad("X"), ctx.get_grad("Y")
            r = ctx.fit(update_X=True, update_Y=True, max_epochs=20, abs_tol=0, rel_tol=0)
            X, Y = ctx.get_factors()
            outs.append((gX, gY, r["loss"], X, Y))
    finally:
        ctx.set_precision("f32")
    for a, b in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(a, b)


def _worker(rank, world, port, outdir, precision, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import pmf_import
    pkg = pmf_import.load()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = make_problem(**CASE)
    lo, hi = pkg.parallel.shard_rows(p["M"], world, rank)
    ctx = pkg.Context(0)
    ctx.set_precision(precision)

    def allreduce(arr):                       # numpy view of the library's pinned staging buffer
        t = torch.from_numpy(arr)
        dist.all_reduce(t)

    ctx.comm_init_host(rank, world, allreduce)
    ctx.comm_set_chunks(3)
    to_context(shard_problem(p, lo, hi), ctx)
    ctx.set_optimizer("adagrad", lr=LR)
    flags = dict(update_X=True, update_Y=True) if mode == "factors" else dict(update_col_layers=True)
    h = ctx.fit(max_epochs=EPOCHS, abs_tol=0, rel_tol=0, **flags)          # the C loop: pmf_fit with a communicator
    X, Y = ctx.get_factors()
    ls, mu = ctx.get_col_params()
    layers = [ls, mu]
    for v in range(len(p["batch_views"])):
        layers += list(ctx.get_batch_view(v))
    # a second segment that stops on its own (loss increase under a large step), to see every rank stop together
    ctx.set_lr(50.0)
    h2 = ctx.fit(max_epochs=EPOCHS + 30, epoch=EPOCHS + 1, abs_tol=0, rel_tol=0, **flags)
    np.savez(Path(outdir) / f"rank{rank}.npz", X=X, Y=Y, loss=h["loss"], lo=lo, hi=hi, term=h["term_code"],
             epochs=h["epochs"], loss2=h2["loss"], term2=h2["term_code"], epochs2=h2["epochs"],
             n_chunks=ctx.comm_info()["n_chunks"], n_coll=ctx.comm_info()["n_collectives"],
             layers=np.concatenate([np.asarray(a, np.float64).ravel() for a in layers]))
    ctx.comm_destroy()
    ctx.close()
    dist.destroy_process_group()


def _run_two_ranks(tmp_path, precision, mode):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    code = ("import sys; sys.path[:0] = [%r, %r]; import test_gpu_comm as t; "
            "t._worker(int(sys.argv[1]), 2, int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5])") % (str(ROOT), str(ROOT / "tests"))
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), str(port), str(tmp_path), precision, mode]) for r in range(2)]
    for pr in procs:
        assert pr.wait(timeout=600) == 0
    return [np.load(tmp_path / f"rank{k}.npz") for k in range(2)]


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_two_ranks_through_the_c_loop_match_single_context_and_oracle(ctx, tmp_path, precision):
    outs = _run_two_ranks(tmp_path, precision, "factors")
    p = make_problem(**CASE)
    ctx.set_precision(precision)
    try:
        to_context(p, ctx)
        ctx.set_optimizer("adagrad", lr=LR)
        r1 = ctx.fit(update_X=True, update_Y=True, max_epochs=EPOCHS, abs_tol=0, rel_tol=0)
        X1, Y1 = ctx.get_factors()
        ctx.set_lr(50.0)
        r2 = ctx.fit(update_X=True, update_Y=True, max_epochs=EPOCHS + 30, epoch=EPOCHS + 1, abs_tol=0, rel_tol=0)
    finally:
        ctx.set_precision("f32")
    m = to_oracle(p)
    ro = m.fit(update_X=True, update_Y=True, lr=LR, max_epochs=EPOCHS, abs_tol=0, rel_tol=0)
    assert r2["term_code"] in ("loss_increase", "nonfinite")
    X = np.concatenate([o["X"] for o in outs], axis=1)
    for o in outs:
        assert int(o["n_chunks"]) == 3 and int(o["n_coll"]) >= EPOCHS * 4
        assert str(o["term"]) == r1["term_code"] == ro["term_code"] and int(o["epochs"]) == r1["epochs"]
        np.testing.assert_allclose(o["loss"], ro["loss"], rtol=5e-5)
        np.testing.assert_allclose(o["loss"], r1["loss"], rtol=2e-5)
        assert str(o["term2"]) == r2["term_code"] and int(o["epochs2"]) == r2["epochs"]
    np.testing.assert_array_equal(outs[0]["loss"], outs[1]["loss"])      # every rank sees the same numbers
    np.testing.assert_array_equal(outs[0]["loss2"], outs[1]["loss2"])
    np.testing.assert_array_equal(outs[0]["Y"], outs[1]["Y"])            # the replicated Y stays bit-identical
    assert rel_err(outs[0]["Y"], Y1) <= 2e-3 and rel_err(X, X1) <= 2e-3, (rel_err(outs[0]["Y"], Y1), rel_err(X, X1))
    assert rel_err(outs[0]["Y"], m.Y) <= 2e-3 and rel_err(X, m.X) <= 2e-3


def test_two_ranks_layer_stage_through_the_c_loop(ctx, tmp_path):
    outs = _run_two_ranks(tmp_path, "f32", "layers")
    p = make_problem(**CASE)
    to_context(p, ctx)
    ctx.set_optimizer("adagrad", lr=LR)
    r1 = ctx.fit(update_col_layers=True, max_epochs=EPOCHS, abs_tol=0, rel_tol=0)
    ls, mu = ctx.get_col_params()
    ref = [ls, mu]
    for v in range(len(p["batch_views"])):
        ref += list(ctx.get_batch_view(v))
    # (the worker goes on with a second, diverging segment: compare the first segment's losses only)
    for o in outs:
        assert str(o["term"]) == r1["term_code"] and int(o["epochs"]) == r1["epochs"]
        np.testing.assert_allclose(o["loss"], r1["loss"], rtol=2e-5)
        assert rel_err(o["layers"], np.concatenate([np.asarray(a, np.float64).ravel() for a in ref])) <= 2e-4
    np.testing.assert_array_equal(outs[0]["layers"], outs[1]["layers"])
    np.testing.assert_array_equal(outs[0]["loss2"], outs[1]["loss2"])


# ---- automatic column chunks must be a COLLECTIVE decision (rank-invariant inputs): two ranks with shards of different
# heights near the threshold where the local rule of round 2 gave S = 1 on one rank and S = 2 on the other
AUTO_N, AUTO_K = 5120, 64                 # 160 column tiles
AUTO_ROWS = (102 * 256, 104 * 256 + 5)    # 102 and 105 row panels; mean 26370 rows = 104 panels: 104 * 160 / 2 >= 32 * 256 -> S = 2


def _worker_auto(rank, world, port, outdir, precision="f32", sb8_off_on_rank=-1):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    if rank == sb8_off_on_rank:      # this rank alone keeps 32 < K <= 64 on pmf_fused_sb2_kernel (128-row panels, not 512)
        os.environ["PMF_SB8"] = "4"
    import torch
    import torch.distributed as dist
    import pmf_import
    pkg = pmf_import.load()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = pkg.Context(0)
    ctx.set_precision(precision)

    def allreduce(arr):
        dist.all_reduce(torch.from_numpy(arr))

    ctx.comm_init_host(rank, world, allreduce)          # (no comm_set_chunks: automatic)
    Ml, N, K = AUTO_ROWS[rank], AUTO_N, AUTO_K
    rng_y = np.random.default_rng(7)
    Yt = (rng_y.standard_normal((K, N)) * 0.3).astype(np.float32)
    Y0 = (rng_y.standard_normal((K, N)) * 0.1).astype(np.float32)
    rng_x = np.random.default_rng(8 + rank)
    Xt = (rng_x.standard_normal((K, Ml)) * 0.3).astype(np.float32)
    X0 = (rng_x.standard_normal((K, Ml)) * 0.1).astype(np.float32)
    ctx.set_data_device(None, Ml, N)
    ctx.set_factors(Xt, Yt)
    ctx.set_col_params(np.zeros(N, np.float32), np.zeros(N, np.float32))
    ctx.set_batch_views([])
    ctx.set_noise([(1, N)], ["normal"], np.ones(N, np.float32))
    ctx.synth_data(seed=100 + rank, noise=0.1)
    ctx.set_factors(X0, Y0)
    ctx.clear_xreg()
    ctx.clear_yreg()
    ctx.add_yreg_fsard(np.full(N, 1.001, np.float32), np.full((K, N), 0.001, np.float32))
    ctx.set_optimizer("adam", lr=0.01)
    h = ctx.fit(update_X=True, update_Y=True, max_epochs=4, abs_tol=0, rel_tol=0)
    info = ctx.comm_info()
    _, Y = ctx.get_factors()
    np.savez(Path(outdir) / f"auto{rank}.npz", loss=h["loss"], n_chunks=info["n_chunks"], n_coll=info["n_collectives"], Y=Y,
             family=ctx.last_kernel())
    ctx.comm_destroy()
    ctx.close()
    dist.destroy_process_group()


def test_two_ranks_unequal_shards_choose_the_same_automatic_chunks(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    code = ("import sys; sys.path[:0] = [%r, %r]; import test_gpu_comm as t; "
            "t._worker_auto(int(sys.argv[1]), 2, int(sys.argv[2]), sys.argv[3])") % (str(ROOT), str(ROOT / "tests"))
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), str(port), str(tmp_path)]) for r in range(2)]
    for pr in procs:
        assert pr.wait(timeout=600) == 0
    a, b = (np.load(tmp_path / f"auto{k}.npz") for k in range(2))
    assert int(a["n_chunks"]) == int(b["n_chunks"]) == 2, (int(a["n_chunks"]), int(b["n_chunks"]))
    assert int(a["n_coll"]) == int(b["n_coll"])
    np.testing.assert_array_equal(a["loss"], b["loss"])
    np.testing.assert_array_equal(a["Y"], b["Y"])
    assert a["loss"][-1] < a["loss"][0]


def test_one_rank_rccl_with_the_per_communicator_cta_cap(pkg, ctx, monkeypatch):
    """ncclCommInitRankConfig (maxCTAs = the reserved CUs) is how a multi-rank communicator is created; a one-rank
    communicator takes the same path under PMF_COMM_CAP_ONE_RANK and must work and give the plain fit's bits."""
    p = make_problem(**CASE)
    r0, X0, Y0 = _fit(ctx, p, "adagrad", update_X=True, update_Y=True)
    monkeypatch.setenv("PMF_COMM_CAP_ONE_RANK", "1")
    assert "NCCL_MAX_NCHANNELS" not in os.environ
    ctx.comm_init(0, 1, pkg._lib.comm_unique_id())
    try:
        r1, X1, Y1 = _fit(ctx, p, "adagrad", update_X=True, update_Y=True)
        assert ctx.comm_info()["n_collectives"] >= 2 * EPOCHS
        out = ctx.comm_allreduce(np.arange(5, dtype=np.float64))
        np.testing.assert_array_equal(out, np.arange(5, dtype=np.float64))
    finally:
        ctx.comm_destroy()
    assert "NCCL_MAX_NCHANNELS" not in os.environ      # the library sets nothing process-wide
    np.testing.assert_array_equal(r1["loss"], r0["loss"])
    np.testing.assert_array_equal(X1, X0)
    np.testing.assert_array_equal(Y1, Y0)


def test_two_ranks_on_different_kernel_families_choose_the_same_automatic_chunks(tmp_path):
    """The automatic chunk count must not depend on the kernel family a rank happens to run: in split mode at K = 64 rank 0
    runs pmf_fused_sb8_kernel (512-row panels), rank 1 is kept on pmf_fused_sb2_kernel (128-row panels).  Counted in each
    rank's own panels (a first version) the threshold gave S = 1 on rank 0 (52 panels x 160 tiles / 2 < 32 x 256) and S = 2 on
    rank 1 (207 panels): different collectives.  Counted in reference panels both choose S = 2."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    code = ("import sys; sys.path[:0] = [%r, %r]; import test_gpu_comm as t; "
            "t._worker_auto(int(sys.argv[1]), 2, int(sys.argv[2]), sys.argv[3], 'bf16x3', 1)") % (str(ROOT), str(ROOT / "tests"))
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), str(port), str(tmp_path)]) for r in range(2)]
    try:
        for pr in procs:
            assert pr.wait(timeout=240) == 0
    finally:
        for pr in procs:      # (a rank left waiting in a collective must not outlive the test)
            if pr.poll() is None:
                pr.kill()
    a, b = (np.load(tmp_path / f"auto{k}.npz") for k in range(2))
    assert (int(a["family"]), int(b["family"])) == (8, 2), (int(a["family"]), int(b["family"]))
    assert int(a["n_chunks"]) == int(b["n_chunks"]) == 2, (int(a["n_chunks"]), int(b["n_chunks"]))
    assert int(a["n_coll"]) == int(b["n_coll"])
    np.testing.assert_array_equal(a["loss"], b["loss"])
    np.testing.assert_array_equal(a["Y"], b["Y"])
    assert a["loss"][-1] < a["loss"][0]

