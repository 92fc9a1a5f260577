"""BASELINE configs[4] as ONE workload (1M x 100k, K = 128, full model, D stored bf16, bf16 MFMA; one rank's shard =
125000 x 100000): K = 128, two batch views x 8 row batches, 20 % Bernoulli columns, 10 % missing entries, column scale /
shift, group regularizer on X + feature-set-ARD on Y, data rounded to bf16 -- all through pmf_fused_sb4_kernel (the launch
counter is asserted):

  * at oracle size: loss, both gradients (data term, and total with the regularizers kept), a 6-epoch trajectory, against the
    fp64 oracle fed the same bf16-rounded matrix;
  * at the per-rank shard size (125000 x 100000, --store bf16, full model): the size-independent properties of
    test_config2_full_size_properties -- loss against the independent statistics kernel, Richardson directional derivative,
    bitwise reproducible loss / gX / gY.
"""
import numpy as np
import pytest

from problems import make_problem, rel_err, to_context, to_oracle
from test_gpu_parity import FIT_TOL, GRAD_TOL, LOSS_RTOL, grads_of
from test_gpu_split_bf16 import bf16_round

pytestmark = pytest.mark.gpu

CONFIG4 = dict(K=128, bernoulli_frac=0.2, n_views=2, batch_views=2, n_batches=8, nan_frac=0.1, weights=True, col_params=True,
               xreg="group", yreg="fsard", n_groups=6, scale=0.35)


@pytest.fixture()
def sctx(ctx):
    ctx.set_precision("bf16x3")
    n0 = ctx.get_precision()[1]
    yield ctx, n0
    ctx.set_precision("f32")


@pytest.mark.parametrize("shape", [(900, 700), (2600, 420)])
def test_config4_loss_and_gradients_match_oracle(sctx, shape):
    ctx, n0 = sctx
    p = make_problem(seed=51, M=shape[0], N=shape[1], **CONFIG4)
    p["D"] = np.asfortranarray(bf16_round(p["D"]))
    to_context(p, ctx)
    ctx.set_data(p["D"], store="bf16")
    try:
        loss, g = grads_of(ctx, p, update_X=True, update_Y=True)
        assert ctx.get_precision() == ("bf16x3", n0 + 1) and ctx.last_kernel() == 8, "pmf_fused_sb8_kernel was not launched"
        assert ctx.last_path()["bmode"] == 1                          # batch layers through the LDS table (panel-local slots)
        m = to_oracle(p)
        lo, go = m.loss_and_grads(update_X=True, update_Y=True)
        m.m.n_xreg = 0
        m.m.n_yreg = 0
        _, gd = m.loss_and_grads(update_X=True, update_Y=True)
        assert abs(loss - gd["data_loss"]) <= LOSS_RTOL * abs(gd["data_loss"]), (loss, gd["data_loss"])
        assert rel_err(g["X"], gd["X"]) <= GRAD_TOL, rel_err(g["X"], gd["X"])
        assert rel_err(g["Y"], gd["Y"]) <= GRAD_TOL, rel_err(g["Y"], gd["Y"])
        # with the regularizers kept: total loss, and total gradients recovered from Adam's first moment after one step at lr -> 0
        ctx.set_optimizer("adam", lr=1e-30, beta1=0.9)
        o = ctx.make_opts(update_X=True, update_Y=True)
        ctx.epoch_begin(o)
        ctx.epoch_step_local(o)
        ctx.epoch_step_shared(o)
        tot, _ = ctx.epoch_loss()
        assert abs(tot - lo) <= LOSS_RTOL * abs(lo), (tot, lo)
        for which in ("X", "Y"):
            _, mom = ctx.get_opt_state(which)
            assert rel_err(mom.astype(np.float64) / 0.1, go[which]) <= GRAD_TOL, which
    finally:
        ctx.set_data(p["D"])


@pytest.mark.parametrize("opt", ["adam", "adagrad"])
def test_config4_fit_trajectory_matches_oracle(sctx, opt):
    ctx, n0 = sctx
    p = make_problem(seed=53, M=900, N=700, random_init=True, **CONFIG4)
    p["D"] = np.asfortranarray(bf16_round(p["D"]))
    lr = 0.01 if opt == "adam" else 0.05
    to_context(p, ctx)
    ctx.set_data(p["D"], store="bf16")
    try:
        ctx.set_optimizer(opt, lr=lr)
        r = ctx.fit(update_X=True, update_Y=True, max_epochs=6, abs_tol=0, rel_tol=0)
        assert ctx.get_precision()[1] == n0 + 6 and ctx.last_kernel() == 8
        X, Y = ctx.get_factors()
    finally:
        ctx.set_data(p["D"])
    m = to_oracle(p)
    ro = m.fit(update_X=True, update_Y=True, opt=opt, lr=lr, max_epochs=6, abs_tol=0, rel_tol=0)
    assert r["term_code"] == ro["term_code"] and r["epochs"] == ro["epochs"]
    np.testing.assert_allclose(r["loss"], ro["loss"], rtol=5e-5)
    tol = 8 * FIT_TOL if opt == "adagrad" else 2 * FIT_TOL      # (AdaGrad's first steps are +-lr sign(g): tests/test_gpu_split_bf16.py)
    assert rel_err(X, m.X) <= tol and rel_err(Y, m.Y) <= tol, (rel_err(X, m.X), rel_err(Y, m.Y))


def full_model_size_properties(sctx, M, N, K, store):
    """Size-independent properties of a full-model pass (20 % Bernoulli columns, column scale / shift, two batch views x 8 row
    batches, 10 % missing) through pmf_fused_sb8_kernel: bitwise reproducible loss / gX / gY, the Gaussian part of the loss
    against the independent statistics kernel, Richardson directional derivative."""
    ctx, n0 = sctx
    rng = np.random.default_rng(9)
    nb, nbat = N // 5, 8
    X0 = (rng.standard_normal((K, M), dtype=np.float32) * 0.2)
    Y0 = (rng.standard_normal((K, N), dtype=np.float32) * 0.2)
    w = (0.5 + rng.random(N)).astype(np.float32)
    logsigma = (0.1 * rng.standard_normal(N)).astype(np.float32)
    mu = (0.3 * rng.standard_normal(N)).astype(np.float32)
    views = []
    for (s, e) in ((1, N // 2), (N // 2 + 1, N)):
        bor = np.sort(rng.integers(0, nbat, size=M)).astype(np.int32)
        views.append(dict(start1=s, stop1=e, batch_of_row=bor,
                          logdelta=(0.25 * rng.standard_normal((nbat, e - s + 1))).astype(np.float32),
                          theta=(0.25 * rng.standard_normal((nbat, e - s + 1))).astype(np.float32)))
    ctx.set_data_device(None, M, N, store=store)
    try:
        ctx.set_factors(X0, Y0)
        ctx.set_col_params(logsigma, mu)
        ctx.set_batch_views(views)
        ctx.set_noise([(1, nb), (nb + 1, N)], ["bernoulli", "normal"], w)
        ctx.clear_xreg()
        ctx.clear_yreg()
        ctx.set_layer_regs()
        ctx.synth_data(seed=321, noise=0.3, frac_nan=0.1)
        Xs = X0 + 0.05 * rng.standard_normal((K, M), dtype=np.float32)
        Ys = Y0 + 0.05 * rng.standard_normal((K, N), dtype=np.float32)
        o = ctx.make_opts(update_X=True, update_Y=True)

        def loss_grad(X, Y, want_grad=True):
            ctx.set_factors(X, Y)
            ctx.epoch_begin(o)
            loss, _ = ctx.epoch_loss()
            return (loss, ctx.get_grad("X"), ctx.get_grad("Y")) if want_grad else loss

        L0, gX, gY = loss_grad(Xs, Ys)
        assert ctx.get_precision() == ("bf16x3", n0 + 1) and ctx.last_path()["bmode"] == 1 and ctx.last_kernel() == 8
        L0b, gXb, gYb = loss_grad(Xs, Ys)
        assert L0 == L0b and np.array_equal(gY, gYb) and np.array_equal(gX, gXb)
        del gXb, gYb
        st = ctx.stats(use_factors=True)
        n_obs = float(st["n"].astype(np.float64).sum())
        assert abs(n_obs / (float(M) * N) - 0.9) < 1e-3
        L_gauss = 0.5 * float(np.sum(w[nb:].astype(np.float64) * st["sqerr"][nb:].astype(np.float64)))
        n_bern = float(st["n"][:nb].astype(np.float64).sum())
        assert 0.05 < (L0 - L_gauss) / n_bern < 1.5, (L0, L_gauss, n_bern)
        g2 = float(np.sum(gX.astype(np.float64) ** 2) + np.sum(gY.astype(np.float64) ** 2))
        e = 0.01 * L0 / g2
        r = []
        for ee in (e, 0.5 * e):
            Le = loss_grad(Xs - np.float32(ee) * gX, Ys - np.float32(ee) * gY, want_grad=False)
            r.append((L0 - Le) / (ee * g2))
        assert 0.5 < r[0] < 1.0 and r[0] < r[1] < 1.0, r
        assert abs(2 * r[1] - r[0] - 1.0) <= 1e-2, r
        # the Gaussian columns alone against pmf_stats exactly
        w0 = w.copy()
        w0[:nb] = 0.0
        ctx.set_noise([(1, nb), (nb + 1, N)], ["bernoulli", "normal"], w0)
        Lg = loss_grad(Xs, Ys, want_grad=False)
        assert abs(Lg - L_gauss) <= 5e-5 * L_gauss, (Lg, L_gauss)
    finally:
        ctx.set_data_device(None, 64, 64)      # release the matrix
        ctx.set_batch_views([])


def test_config4_shard_size_properties(sctx):
    """One rank's shard of configs[4]: 125000 x 100000, K = 128, D stored bf16 (25 GB), full model."""
    full_model_size_properties(sctx, 125000, 100000, 128, "bf16")


def test_headline_size_full_model_properties_k64(sctx):
    """The headline matrix (200000 x 50000, K = 64, f32) with the full model: the 512-row-panel instance of the kernel with its
    batch-layer variant (table planes, five-view limit, late X^T mid loads) at a size with many pieces per workgroup."""
    full_model_size_properties(sctx, 200000, 50000, 64, "f32")


# ---- pmf_fused_sb8_kernel specifics (csrc/pmf_fused_sb8.hip.inc) --------------------------------------------------------
@pytest.mark.parametrize("sx,sy", [(1e-4, 1e4), (3e3, 1.0 / 3e3), (1e-6, 1e-3)])
def test_sb8_f16_prescale_handles_operand_ranges(sctx, sx, sy):
    """The forward runs on f16 pairs: operands beyond f16's range (|sigma Y| up to 1e5 here: > 65504) or far below its normal
    range must come out like the exact product, through the power-of-two pre-scale of the operand images (k_sb8_absmax).
    The same problem with X scaled by sx and Y by sy (first case: X'Y unchanged) against the fp64 oracle."""
    ctx, n0 = sctx
    p = make_problem(seed=57, M=700, N=420, K=128, nan_frac=0.05, weights=True, col_params=True, scale=0.35)
    p["X"] = np.asfortranarray((p["X"] * sx).astype(np.float32))
    p["Y"] = np.asfortranarray((p["Y"] * sy).astype(np.float32))
    if abs(sx * sy - 1.0) > 1e-6:       # keep the residuals O(noise): the data follows the scaled product
        p["D"] = np.asfortranarray(np.where(np.isnan(p["D"]), np.nan, p["D"] * np.float32(sx * sy)).astype(np.float32))
        p["mu"] = (p["mu"] * np.float32(sx * sy)).astype(np.float32)
    to_context(p, ctx)
    loss, g = grads_of(ctx, p, update_X=True, update_Y=True)
    assert ctx.last_kernel() == 8
    m = to_oracle(p)
    m.m.n_xreg = 0
    m.m.n_yreg = 0
    _, gd = m.loss_and_grads(update_X=True, update_Y=True)
    assert np.isfinite(loss) and abs(loss - gd["data_loss"]) <= LOSS_RTOL * abs(gd["data_loss"]), (loss, gd["data_loss"])
    assert rel_err(g["X"], gd["X"]) <= GRAD_TOL and rel_err(g["Y"], gd["Y"]) <= GRAD_TOL, (rel_err(g["X"], gd["X"]), rel_err(g["Y"], gd["Y"]))


def test_sb8_column_chunks_and_kernel_selection(sctx):
    """Column chunks (one launch and one pre-scale of sigma Y per chunk) give the unchunked gradients; single-gradient passes
    and PMF_SB8=0 stay on the 128-row kernel."""
    ctx, n0 = sctx
    p = make_problem(seed=59, M=900, N=1300, K=128, nan_frac=0.05, weights=True, col_params=True, scale=0.35, xreg="l2", yreg="fsard")
    to_context(p, ctx)
    l0, g0 = grads_of(ctx, p, update_X=True, update_Y=True)
    assert ctx.last_kernel() == 8
    ctx.comm_set_chunks(3)
    try:
        l1, g1 = grads_of(ctx, p, update_X=True, update_Y=True)
        assert ctx.last_kernel() == 8
    finally:
        ctx.comm_set_chunks(0)
    assert abs(l1 - l0) <= 1e-7 * abs(l0)
    assert rel_err(g1["X"], g0["X"]) <= 2e-6 and rel_err(g1["Y"], g0["Y"]) <= 2e-6
    _, gx_only = grads_of(ctx, p, update_X=True)
    assert ctx.last_kernel() == 4
    assert rel_err(gx_only["X"], g0["X"]) <= 2e-5      # (three bf16 terms against two f16 terms in the forward: both 1e-7 of max|Z|)
    import os
    os.environ["PMF_SB8"] = "0"
    try:
        l4, g4 = grads_of(ctx, p, update_X=True, update_Y=True)
        assert ctx.last_kernel() == 4
    finally:
        del os.environ["PMF_SB8"]
    assert abs(l4 - l0) <= 2e-6 * abs(l0) and rel_err(g4["X"], g0["X"]) <= 2e-5 and rel_err(g4["Y"], g0["Y"]) <= 2e-5


# ---- the 512-row-panel instance of the same kernel (32 < K <= 64: four waves x four row blocks) --------------------------
K64_CASES = ["ragged_k64_nan", "mixed_k48", "many_panels_k40", "many_panels_two_tiles", "mixed_batch_nan_k64", "batch_many_panels_k40"]


@pytest.mark.parametrize("name", K64_CASES)
def test_k64_both_gradients_run_sb8_and_sb2_agrees(sctx, name):
    """With both gradients a 32 < K <= 64 pass runs pmf_fused_sb8_kernel<2> (family 8); PMF_SB8=4 keeps that family for
    K > 96 only and puts the pass back on pmf_fused_sb2_kernel (family 2).  Both against the fp64 oracle on the same problem
    (the parity cases of tests/test_gpu_split_bf16.py), so the older kernel keeps its both-gradient coverage."""
    import os
    from test_gpu_split_bf16 import CASES
    ctx, n0 = sctx
    p = make_problem(seed=11, **CASES[name])
    to_context(p, ctx)
    m = to_oracle(p)
    m.m.n_xreg = 0
    m.m.n_yreg = 0
    _, gd = m.loss_and_grads(update_X=True, update_Y=True)
    for env, family in ((None, 8), ("4", 2)):
        if env is not None:
            os.environ["PMF_SB8"] = env
        try:
            loss, g = grads_of(ctx, p, update_X=True, update_Y=True)
        finally:
            os.environ.pop("PMF_SB8", None)
        assert ctx.last_kernel() == family, (ctx.last_kernel(), family)
        assert abs(loss - gd["data_loss"]) <= LOSS_RTOL * abs(gd["data_loss"]) + 1e-6, (family, loss, gd["data_loss"])
        assert rel_err(g["X"], gd["X"]) <= GRAD_TOL and rel_err(g["Y"], gd["Y"]) <= GRAD_TOL, (family, rel_err(g["X"], gd["X"]), rel_err(g["Y"], gd["Y"]))
    assert ctx.get_precision()[1] == n0 + 2


def test_k64_more_batch_views_than_the_tall_panel_holds_fall_back_to_sb2(sctx):
    """The 512-row panel's LDS has room for five views' panel-local batch slots (Sb8Cfg<2>::max_bv); a model with six
    batch-layer views runs the 128-row kernel instead, with the same results as the oracle."""
    ctx, n0 = sctx
    p = make_problem(seed=61, M=1500, N=360, K=64, n_views=6, batch_views=6, n_batches=4, nan_frac=0.05, col_params=True, scale=0.5)
    to_context(p, ctx)
    loss, g = grads_of(ctx, p, update_X=True, update_Y=True)
    assert ctx.last_kernel() == 2 and ctx.get_precision()[1] == n0 + 1
    m = to_oracle(p)
    m.m.n_xreg = 0
    m.m.n_yreg = 0
    _, gd = m.loss_and_grads(update_X=True, update_Y=True)
    assert abs(loss - gd["data_loss"]) <= LOSS_RTOL * abs(gd["data_loss"]) + 1e-6
    assert rel_err(g["X"], gd["X"]) <= GRAD_TOL and rel_err(g["Y"], gd["Y"]) <= GRAD_TOL
    p5 = make_problem(seed=61, M=1500, N=360, K=64, n_views=5, batch_views=5, n_batches=4, nan_frac=0.05, col_params=True, scale=0.5)
    to_context(p5, ctx)
    grads_of(ctx, p5, update_X=True, update_Y=True)
    assert ctx.last_kernel() == 8


@pytest.mark.parametrize("opt", ["adagrad", "adam"])
def test_k64_fit_trajectory_through_sb8(sctx, opt):
    """Ten epochs of the mixed-noise, batch-layer K = 64 problem.  AdaGrad's first step is lr g / (|g| + eps): an entry of gY
    that is 1e-5 of max|gY| moves by the full lr in the direction of its sign, so the three-term gradient products (4e-6 of
    max|g|) show up as 5e-3 of max|Y| after ten epochs -- in this kernel and in pmf_fused_sb2_kernel alike (4.8e-3; the exact
    kernel: 2.5e-5).  Adam's bias-corrected first step has the same property but its eps-floor sits lower: 8e-6."""
    ctx, n0 = sctx
    from test_gpu_split_bf16 import CASES
    p = make_problem(seed=13, random_init=True, **CASES["mixed_batch_nan_k64"])
    lr = 0.05 if opt == "adagrad" else 0.01
    to_context(p, ctx)
    ctx.set_optimizer(opt, lr=lr)
    r = ctx.fit(update_X=True, update_Y=True, max_epochs=10, abs_tol=0, rel_tol=0)
    assert ctx.get_precision()[1] == n0 + 10 and ctx.last_kernel() == 8
    m = to_oracle(p)
    ro = m.fit(update_X=True, update_Y=True, opt=opt, lr=lr, max_epochs=10, abs_tol=0, rel_tol=0)
    np.testing.assert_allclose(r["loss"], ro["loss"], rtol=5e-5)
    X, Y = ctx.get_factors()
    tol = FIT_TOL if opt == "adam" else 1e-2
    assert rel_err(X, m.X) <= tol and rel_err(Y, m.Y) <= tol, (rel_err(X, m.X), rel_err(Y, m.Y))


def test_noise_tile_weights_change_the_split_not_the_result(sctx):
    """The work split weighs Bernoulli / Poisson tiles by measured per-kernel costs (noise_tile_weights, pmf_hip.hip).  Whatever
    the weights (PMF_W_BERN / PMF_W_BATCH override them), loss and gradients are those of the oracle: a weight only moves the
    boundaries between workgroups."""
    import os
    ctx, n0 = sctx
    p = make_problem(seed=71, M=9000, N=700, K=64, bernoulli_frac=0.3, poisson_frac=0.2, n_views=3, batch_views=2, n_batches=6,
                     nan_frac=0.05, weights=True, col_params=True, scale=0.4)
    m = to_oracle(p)
    m.m.n_xreg = 0
    m.m.n_yreg = 0
    _, gd = m.loss_and_grads(update_X=True, update_Y=True)
    res = []
    for wb, we in ((None, None), ("16", "0"), ("48", "20")):
        for k, v in (("PMF_W_BERN", wb), ("PMF_W_BATCH", we)):
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        try:
            to_context(p, ctx)       # (a new model: the cached split is keyed on the weights, this also drops it)
            loss, g = grads_of(ctx, p, update_X=True, update_Y=True)
        finally:
            os.environ.pop("PMF_W_BERN", None)
            os.environ.pop("PMF_W_BATCH", None)
        assert ctx.last_kernel() == 8
        assert abs(loss - gd["data_loss"]) <= LOSS_RTOL * abs(gd["data_loss"])
        assert rel_err(g["X"], gd["X"]) <= GRAD_TOL and rel_err(g["Y"], gd["Y"]) <= GRAD_TOL
        res.append((loss, g))
    for loss, g in res[1:]:
        assert abs(loss - res[0][0]) <= 1e-6 * abs(res[0][0])
        assert rel_err(g["X"], res[0][1]["X"]) <= 1e-5 and rel_err(g["Y"], res[0][1]["Y"]) <= 1e-5
