"""Views with more than 15 row batches (src/batch_array.jl:78-147 places no bound on the batch count; the TCGA-scale
models of the reference have dozens of batches per assay).

The fused kernels keep a 32-column x 16-slot {delta, theta} table in LDS.  Slots are local to a (row panel, view): with
samples grouped by batch a 256-row panel meets a handful of the view's batches, so the table is gathered per panel from
the dense [N][slots] table through the panel's slot -> batch map (variant 1).  A panel with more than 15 distinct batches
of one view (scrambled rows) sends the launch to the per-entry gather variant (2).  The layer pass keeps the whole
[64 columns][slots] table in LDS.  Every test asserts the variant that ran (pmf_debug_last_path): a silent fall-back
would pass parity and lose the speed."""
import numpy as np
import pytest

from problems import make_problem, rel_err, to_context, to_oracle
from test_gpu_parity import FIT_TOL, GRAD_TOL, LOSS_RTOL, grads_of

pytestmark = pytest.mark.gpu

BASE = dict(n_views=3, batch_views=2, n_batches=40, nan_frac=0.05, weights=True, col_params=True, bernoulli_frac=0.2,
            scale=0.5)
CASES = {
    # name: (problem, expected fused variant, slots of the dense table)
    "sorted40_k16": (dict(BASE, M=2100, N=230, K=16, batch_order="sorted"), 1, 64),
    "sorted40_k64": (dict(BASE, M=2100, N=230, K=64, batch_order="sorted"), 1, 64),
    "sorted40_k128": (dict(BASE, M=1500, N=130, K=128, batch_order="sorted", scale=0.3), 1, 64),
    "sorted40_k80": (dict(BASE, M=1500, N=130, K=80, batch_order="sorted", scale=0.3), 1, 64),
    "random40_k16": (dict(BASE, M=2100, N=230, K=16, batch_order="random"), 2, 64),
    "random40_k64": (dict(BASE, M=1000, N=130, K=64, batch_order="random"), 2, 64),
    # one sorted and one scrambled view: a single view over the limit sends the whole launch to variant 2
    "mixed40_k32": (dict(BASE, M=2100, N=230, K=32), 2, 64),
    "sorted100_k8": (dict(BASE, M=4000, N=100, K=8, n_batches=100, batch_order="sorted"), 1, 128),
    # 16 batches: one more than the 15 usable slots of the 16-slot table -> 32 slots
    "sorted16_k32": (dict(BASE, M=1200, N=100, K=32, n_batches=16, batch_order="sorted"), 1, 32),
}


def _check_grads(ctx, p, variant, slots):
    loss, g = grads_of(ctx, p, update_X=True, update_Y=True)
    lp = ctx.last_path()
    assert lp["bmode"] == variant and lp["slots"] == slots, lp
    m = to_oracle(p)
    m.m.n_xreg = 0
    m.m.n_yreg = 0
    _, gd = m.loss_and_grads(update_X=True, update_Y=True)
    assert abs(loss - gd["data_loss"]) <= LOSS_RTOL * abs(gd["data_loss"]) + 1e-6, (loss, gd["data_loss"])
    assert rel_err(g["X"], gd["X"]) <= GRAD_TOL, rel_err(g["X"], gd["X"])
    assert rel_err(g["Y"], gd["Y"]) <= GRAD_TOL, rel_err(g["Y"], gd["Y"])


@pytest.mark.parametrize("name", list(CASES))
def test_many_batches_loss_and_gradients_match_oracle(ctx, name):
    kw, variant, slots = CASES[name]
    p = make_problem(seed=21, **kw)
    to_context(p, ctx)
    _check_grads(ctx, p, variant, slots)


@pytest.mark.parametrize("name", ["sorted40_k16", "sorted40_k64", "sorted40_k128", "random40_k64"])
def test_many_batches_split_bf16_loss_and_gradients_match_oracle(ctx, name):
    kw, variant, slots = CASES[name]
    p = make_problem(seed=22, **kw)
    to_context(p, ctx)
    ctx.set_precision("bf16x3")
    try:
        n0 = ctx.get_precision()[1]
        _check_grads(ctx, p, variant, slots)
        # the split kernels have no gather variant: scrambled rows run the exact kernel
        assert ctx.get_precision()[1] == n0 + (1 if variant == 1 else 0)
    finally:
        ctx.set_precision("f32")


@pytest.mark.parametrize("which", ["X", "Y"])
def test_many_batches_single_factor_gradient(ctx, which):
    kw, variant, slots = CASES["sorted40_k64"]
    p = make_problem(seed=23, **kw)
    to_context(p, ctx)
    flags = dict(update_X=which == "X", update_Y=which == "Y")
    loss, g = grads_of(ctx, p, **flags)
    assert ctx.last_path()["bmode"] == variant
    m = to_oracle(p)
    m.m.n_xreg = 0
    m.m.n_yreg = 0
    _, go = m.loss_and_grads(**flags)
    assert abs(loss - go["data_loss"]) <= LOSS_RTOL * abs(go["data_loss"]) + 1e-6
    assert rel_err(g[which], go[which]) <= GRAD_TOL


@pytest.mark.parametrize("name,layer_path", [("sorted40_k16", 1), ("random40_k64", 1), ("sorted40_k128", 1),
                                             ("sorted100_k8", 1), ("sorted16_k32", 1)])
def test_many_batches_layer_gradients_match_oracle(ctx, name, layer_path):
    kw, _, slots = CASES[name]
    p = make_problem(seed=24, layer_regs=True, **kw)
    to_context(p, ctx)
    loss, g = grads_of(ctx, p, update_col_layers=True)
    lp = ctx.last_path()
    assert lp["layer_path"] == layer_path and lp["slots"] == slots, lp
    m = to_oracle(p)
    m.m.has_colreg = 0
    m.m.has_batchreg = 0
    lo, go = m.loss_and_grads(update_col_layers=True)
    assert abs(loss - go["data_loss"]) <= LOSS_RTOL * abs(go["data_loss"])
    assert rel_err(g["mu"], go["mu"]) <= GRAD_TOL
    assert rel_err(g["logsigma"], go["logsigma"]) <= GRAD_TOL
    for v in range(len(p["batch_views"])):
        assert rel_err(g["theta"][v], go["theta"][v]) <= GRAD_TOL
        assert rel_err(g["logdelta"][v], go["logdelta"][v]) <= GRAD_TOL


def test_very_many_batches_take_the_valu_layer_kernel(ctx):
    """200 batches: a 256-slot table of 64 columns does not fit in LDS next to the X panels -> the VALU layer kernel; the
    data pass still runs variant 1 (sorted rows)."""
    p = make_problem(seed=25, layer_regs=True, **dict(BASE, M=6000, N=70, K=8, n_batches=200, batch_order="sorted"))
    to_context(p, ctx)
    loss, g = grads_of(ctx, p, update_col_layers=True)
    lp = ctx.last_path()
    assert lp["layer_path"] == 2 and lp["slots"] == 256, lp
    m = to_oracle(p)
    m.m.has_colreg = 0
    m.m.has_batchreg = 0
    lo, go = m.loss_and_grads(update_col_layers=True)
    assert abs(loss - go["data_loss"]) <= LOSS_RTOL * abs(go["data_loss"])
    for v in range(len(p["batch_views"])):
        assert rel_err(g["theta"][v], go["theta"][v]) <= GRAD_TOL
        assert rel_err(g["logdelta"][v], go["logdelta"][v]) <= GRAD_TOL
    _check_grads(ctx, p, 1, 256)


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_many_batches_joint_fit_trajectory_matches_oracle(ctx, precision):
    """fit! with factors and layers trained together (fit.jl:498-520) on a 40-batch model."""
    kw, variant, _ = CASES["sorted40_k64"]
    p = make_problem(seed=26, random_init=True, layer_regs=True, xreg="l2", yreg="l2", **kw)
    to_context(p, ctx)
    ctx.set_precision(precision)
    try:
        ctx.set_optimizer("adagrad", lr=0.05)
        kwf = dict(update_X=True, update_Y=True, update_col_layers=True, max_epochs=8, abs_tol=0, rel_tol=0)
        r = ctx.fit(**kwf)
        lp = ctx.last_path()
        assert lp["bmode"] == variant and lp["layer_path"] == 1, lp
        m = to_oracle(p)
        ro = m.fit(opt="adagrad", lr=0.05, **kwf)
        np.testing.assert_allclose(r["loss"], ro["loss"], rtol=1e-4)
        X, Y = ctx.get_factors()
        # (bf16x3: AdaGrad's first steps are +-lr sign(g); the three-term gradient products' 4e-6 flips a few near-zero
        # entries by O(lr) -- the max-norm tolerance is widened as in test_split_bf16_k128_fit_trajectory, the loss is not)
        tol = 2 * FIT_TOL if precision == "f32" else 8 * FIT_TOL
        assert rel_err(X, m.X) <= tol, rel_err(X, m.X)
        assert rel_err(Y, m.Y) <= tol, rel_err(Y, m.Y)
        for v in range(len(p["batch_views"])):
            ld, th = ctx.get_batch_view(v)
            assert rel_err(th, m.theta[v]) <= 3 * FIT_TOL, rel_err(th, m.theta[v])
            assert rel_err(ld, m.logdelta[v]) <= 3 * FIT_TOL, rel_err(ld, m.logdelta[v])
    finally:
        ctx.set_precision("f32")
