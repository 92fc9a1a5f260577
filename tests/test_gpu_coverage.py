"""Device paths that round 1 left without an oracle comparison ("what's weak"):

* every regularizer the library implements, KEPT during the comparison: {X: none, l2, group, composite} x {Y: none, l2,
  group, ard with an uncovered column range, fsard} -- the regularizer's value (the epoch's loss and its replicated part)
  and its gradient, recovered exactly from one AdaGrad step (acc = eps + g^2, step direction = -sign g), against the
  oracle's total gradient;
* the joint epoch (fit_joint, src/fit.jl:987-1000): update_X = update_Y = update_col_layers in one epoch -- loss from the
  fused pass, layer gradients from the layer pass, all steps simultaneous -- against o_fit;
* BASELINE configs[2] at its full size (20000 x 10000, 8000 Gaussian + 2000 Bernoulli columns, 2 views x 8 row batches,
  10 % missing) through size-independent properties;
* mf_fit_adapt_lr! over several segments: the optimizer state must survive the re-marshal between segments
  (src/fit.jl:55-69)."""
import numpy as np
import pytest

from problems import make_problem, rel_err, to_context, to_oracle

pytestmark = pytest.mark.gpu
LOSS_RTOL, GRAD_TOL, FIT_TOL = 2e-5, 2e-4, 2e-3
EPS = 1e-8


def _total_gradient_from_one_adagrad_step(ctx, which, before):
    """After ONE AdaGrad step from fresh state: acc = eps + g^2 and p_new - p_old = -lr g / (sqrt(acc) + eps)."""
    acc, _ = ctx.get_opt_state(which)
    X, Y = ctx.get_factors()
    after = X if which == "X" else Y
    mag = np.sqrt(np.maximum(acc.astype(np.float64) - EPS, 0.0))
    return -np.sign(after.astype(np.float64) - before.astype(np.float64)) * mag


@pytest.mark.parametrize("yreg", [None, "l2", "group", "ard_gap", "fsard"])
@pytest.mark.parametrize("xreg", [None, "l2", "group", "composite"])
def test_regularizer_value_and_gradient_match_oracle(ctx, xreg, yreg):
    p = make_problem(seed=41, M=333, N=190, K=24, n_views=3, bernoulli_frac=0.2, nan_frac=0.05, weights=True,
                     col_params=True, xreg=xreg, yreg=yreg, random_init=True, scale=0.6)
    to_context(p, ctx)
    ctx.set_optimizer("adagrad", lr=1e-3, eps=EPS)
    o = ctx.make_opts(update_X=True, update_Y=True)
    ctx.epoch_begin(o)
    ctx.epoch_step_local(o)
    ctx.epoch_step_shared(o)
    loss, shared = ctx.epoch_loss()
    m = to_oracle(p)
    lo, go = m.loss_and_grads(update_X=True, update_Y=True)
    assert abs(loss - lo) <= LOSS_RTOL * abs(lo), (loss, lo)
    # the replicated part of the loss = the Y regularizer's value (what a sharded host must count once)
    m.m.n_xreg = 0
    l_noX, _ = m.loss_and_grads(update_X=True, update_Y=True)
    yreg_val = l_noX - go["data_loss"]
    assert abs(shared - yreg_val) <= LOSS_RTOL * max(abs(yreg_val), 1e-3 * abs(lo)), (shared, yreg_val)
    for which in ("X", "Y"):
        g = _total_gradient_from_one_adagrad_step(ctx, which, p[which])
        assert rel_err(g, go[which]) <= GRAD_TOL, (which, rel_err(g, go[which]))
    if yreg == "ard_gap":   # the uncovered view's columns see the data gradient only
        s, e = p["view_ranges"][1]
        m2 = to_oracle(p)
        m2.m.n_yreg = 0
        _, gd = m2.loss_and_grads(update_X=True, update_Y=True)
        gY = _total_gradient_from_one_adagrad_step(ctx, "Y", p["Y"])
        assert rel_err(gY[:, s - 1:e], gd["Y"][:, s - 1:e]) <= GRAD_TOL
        assert rel_err(gY, gd["Y"]) > 10 * GRAD_TOL          # ... and the covered ones do not


@pytest.mark.parametrize("yreg", ["l2", "group", "ard_gap"])
@pytest.mark.parametrize("opt", ["adagrad", "adam"])
def test_fit_trajectory_with_y_regularizers_matches_oracle(ctx, yreg, opt):
    p = make_problem(seed=43, M=420, N=260, K=32, n_views=3, nan_frac=0.05, weights=True, col_params=True,
                     xreg="group", yreg=yreg, random_init=True, scale=0.6)
    lr = 0.05 if opt == "adagrad" else 0.01
    to_context(p, ctx)
    ctx.set_optimizer(opt, lr=lr)
    r = ctx.fit(update_X=True, update_Y=True, max_epochs=6, abs_tol=0, rel_tol=0)
    m = to_oracle(p)
    ro = m.fit(update_X=True, update_Y=True, opt=opt, lr=lr, max_epochs=6, abs_tol=0, rel_tol=0)
    assert r["term_code"] == ro["term_code"] and r["epochs"] == ro["epochs"]
    np.testing.assert_allclose(r["loss"], ro["loss"], rtol=5e-5)
    X, Y = ctx.get_factors()
    assert rel_err(X, m.X) <= FIT_TOL and rel_err(Y, m.Y) <= FIT_TOL, (rel_err(X, m.X), rel_err(Y, m.Y))


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
@pytest.mark.parametrize("frozen", [0b0111, 0b0001])
def test_joint_epoch_matches_oracle(ctx, precision, frozen):
    """fit_joint (src/fit.jl:987-1000): X, Y and the unfrozen column layers in the same epochs.  frozen = 0b0111 is the
    reference's own call (layers 1:3 frozen, theta trained); 0b0001 also trains logdelta and mu."""
    p = make_problem(seed=44, M=420, N=260, K=32, bernoulli_frac=0.2, n_views=2, batch_views=2, n_batches=8, nan_frac=0.1,
                     weights=True, col_params=True, xreg="group", yreg="fsard", layer_regs=True, random_init=True, scale=0.6)
    ctx.set_precision(precision)
    try:
        to_context(p, ctx)
        ctx.set_optimizer("adagrad", lr=0.05)
        kw = dict(update_X=True, update_Y=True, update_col_layers=True, frozen_layers=frozen, max_epochs=6, abs_tol=0, rel_tol=0)
        r = ctx.fit(**kw)
        X, Y = ctx.get_factors()
        ls, mu = ctx.get_col_params()
        views = [ctx.get_batch_view(v) for v in range(2)]
    finally:
        ctx.set_precision("f32")
    m = to_oracle(p)
    ro = m.fit(lr=0.05, **kw)
    assert r["term_code"] == ro["term_code"] and r["epochs"] == ro["epochs"]
    np.testing.assert_allclose(r["loss"], ro["loss"], rtol=5e-5)
    assert rel_err(X, m.X) <= FIT_TOL and rel_err(Y, m.Y) <= FIT_TOL, (rel_err(X, m.X), rel_err(Y, m.Y))
    np.testing.assert_array_equal(ls, p["logsigma"])                       # layer 1 frozen in both cases
    if frozen == 0b0111:
        np.testing.assert_array_equal(mu, p["mu"])
    else:
        assert rel_err(mu, m.mu) <= 3 * FIT_TOL
    for v in range(2):
        ld, th = views[v]
        assert rel_err(th, m.theta[v]) <= 3 * FIT_TOL, rel_err(th, m.theta[v])
        if frozen == 0b0111:
            np.testing.assert_array_equal(ld, p["batch_views"][v]["logdelta"])
        else:
            assert rel_err(ld, m.logdelta[v]) <= 3 * FIT_TOL


def test_config2_full_size_properties(ctx):
    """BASELINE configs[2]: 20000 x 10000, 2000 Bernoulli + 8000 Gaussian columns, BatchArray shift / scale on 2 views x 8
    row batches, 10 % missing entries, K = 32.  The oracle takes minutes at this size; size-independent properties:
      * the fused pass's Gaussian-column loss equals 0.5 sum_j w_j sqerr_j from the independent statistics kernel, and
        the observed-entry count is what was masked;
      * the gradient is the gradient of the loss it reports: directional derivative along -g by Richardson extrapolation
        (the loss is smooth, not quadratic, on the Bernoulli columns: third-order term bounded by the step);
      * loss, grad(X), grad(Y) bitwise reproducible; layer gradients (theta) consistent with the loss the same way."""
    M, N, K = 20000, 10000, 32
    rng = np.random.default_rng(7)
    nb, nbat = 2000, 8
    X0 = (rng.standard_normal((K, M)) * 0.3).astype(np.float32)
    Y0 = (rng.standard_normal((K, N)) * 0.3).astype(np.float32)
    w = (0.5 + rng.random(N)).astype(np.float32)
    logsigma = (0.1 * rng.standard_normal(N)).astype(np.float32)
    mu = (0.3 * rng.standard_normal(N)).astype(np.float32)
    views = []
    for (s, e) in ((1, N // 2), (N // 2 + 1, N)):
        bor = np.sort(rng.integers(0, nbat, size=M)).astype(np.int32)
        views.append(dict(start1=s, stop1=e, batch_of_row=bor,
                          logdelta=(0.25 * rng.standard_normal((nbat, e - s + 1))).astype(np.float32),
                          theta=(0.25 * rng.standard_normal((nbat, e - s + 1))).astype(np.float32)))
    ctx.set_data_device(None, M, N)
    ctx.set_factors(X0, Y0)
    ctx.set_col_params(logsigma, mu)
    ctx.set_batch_views(views)
    ctx.set_noise([(1, nb), (nb + 1, N)], ["bernoulli", "normal"], w)
    ctx.clear_xreg()
    ctx.clear_yreg()
    ctx.set_layer_regs()
    ctx.synth_data(seed=123, noise=0.3, frac_nan=0.1)          # D ~ model(X0, Y0) + noise, Bernoulli columns 0/1, 10 % NaN
    Xs = (X0 + 0.1 * rng.standard_normal((K, M))).astype(np.float32)   # evaluate away from the generating point
    Ys = (Y0 + 0.1 * rng.standard_normal((K, N))).astype(np.float32)
    o = ctx.make_opts(update_X=True, update_Y=True)

    def loss_grad(X, Y, want_grad=True):
        ctx.set_factors(X, Y)
        ctx.epoch_begin(o)
        loss, _ = ctx.epoch_loss()
        return (loss, ctx.get_grad("X"), ctx.get_grad("Y")) if want_grad else loss

    L0, gX, gY = loss_grad(Xs, Ys)
    L0b, gXb, gYb = loss_grad(Xs, Ys)
    assert L0 == L0b and np.array_equal(gY, gYb) and np.array_equal(gX, gXb)
    st = ctx.stats(use_factors=True)
    n_obs = float(st["n"].astype(np.float64).sum())
    assert abs(n_obs / (M * N) - 0.9) < 1e-3
    # Gaussian columns: loss = 0.5 w sqerr; Bernoulli columns: bounded below by 0, checked through the derivative
    L_gauss = 0.5 * float(np.sum(w[nb:].astype(np.float64) * st["sqerr"][nb:].astype(np.float64)))
    # ... the rest is the Bernoulli columns' w (softplus(z) - y z) > 0: between 0.05 and 1.5 per observed entry here
    n_bern = float(st["n"][:nb].astype(np.float64).sum())
    assert 0.05 < (L0 - L_gauss) / n_bern < 1.5, (L0, L_gauss, n_bern)
    g2 = float(np.sum(gX.astype(np.float64) ** 2) + np.sum(gY.astype(np.float64) ** 2))
    e = 0.01 * L0 / g2
    r = []
    for ee in (e, 0.5 * e):
        Le = loss_grad((Xs - ee * gX).astype(np.float32), (Ys - ee * gY).astype(np.float32), want_grad=False)
        r.append((L0 - Le) / (ee * g2))
    assert 0.5 < r[0] < 1.0 and r[0] < r[1] < 1.0, r
    assert abs(2 * r[1] - r[0] - 1.0) <= 1e-2, r
    # the same with the Gaussian columns alone against pmf_stats exactly: zero weight on the Bernoulli columns
    w0 = w.copy()
    w0[:nb] = 0.0
    ctx.set_noise([(1, nb), (nb + 1, N)], ["bernoulli", "normal"], w0)
    Lg = loss_grad(Xs, Ys, want_grad=False)
    assert abs(Lg - L_gauss) <= 5e-5 * L_gauss, (Lg, L_gauss)
    ctx.set_noise([(1, nb), (nb + 1, N)], ["bernoulli", "normal"], w)
    # layer gradients: d loss / d theta along -g_theta
    ol = ctx.make_opts(update_col_layers=True, frozen_layers=0b0111)
    ctx.set_factors(Xs, Ys)
    ctx.epoch_begin(ol)
    Ll, _ = ctx.epoch_loss()
    assert abs(Ll - L0) <= 2e-6 * L0                      # the layer pass reports the same loss as the fused pass
    gth = [ctx.get_grad("theta", v) for v in range(2)]
    g2t = float(sum(np.sum(g.astype(np.float64) ** 2) for g in gth))
    et = 0.01 * L0 / g2t
    rt = []
    for ee in (et, 0.5 * et):
        ctx.set_batch_views([dict(v, theta=(v["theta"] - ee * g).astype(np.float32)) for v, g in zip(views, gth)])
        ctx.epoch_begin(ol)
        rt.append((L0 - ctx.epoch_loss()[0]) / (ee * g2t))
    ctx.set_batch_views(views)
    assert abs(2 * rt[1] - rt[0] - 1.0) <= 1e-2, rt


def test_adapt_lr_keeps_optimizer_state_across_segments(pkg):
    """mf_fit_adapt_lr! (src/fit.jl:46-75): on "loss_increase" eta is halved and MF.fit! resumes with the SAME optimizer
    object.  Both hosts re-marshal the model before every segment; the accumulators must survive that.  Reference
    behaviour = the oracle keeping one OptState across its fit() calls."""
    import test_gpu_host as th
    model = th.reference_fit_setup(pkg, seed=3)
    mo = th.oracle_of(model)
    # lr = 32: large enough to overshoot more than once
    pkg.mf_fit_adapt_lr_(model, lr=32.0, update_X=True, update_Y=True, max_epochs=60, min_lr=0.2, abs_tol=0, rel_tol=0,
                         verbosity=0)
    ctx = model.device_context()
    # the oracle through the same host loop semantics: segments until eta < min_lr, state kept
    lr, epoch, segs = 32.0, 1, 0
    while lr >= 0.2:
        ro = mo.fit(update_X=True, update_Y=True, lr=lr, max_epochs=60, epoch=epoch, abs_tol=0, rel_tol=0)
        segs += 1
        if ro["term_code"] != "loss_increase":
            break
        lr *= 0.5
        epoch = ro["epochs"]
    assert segs >= 2, segs                                # the case really resumes after a loss increase
    X, Y = model.matfac.X, model.matfac.Y
    assert rel_err(X, mo.X) <= 3 * FIT_TOL and rel_err(Y, mo.Y) <= 3 * FIT_TOL, (rel_err(X, mo.X), rel_err(Y, mo.Y))
    acc, _ = ctx.get_opt_state("Y")
    acc_o = mo._st_bufs[2].reshape(acc.shape, order="F")
    assert rel_err(acc, acc_o) <= 1e-3, rel_err(acc, acc_o)    # a state that had been reset would be orders of magnitude smaller
    model.release_device()
