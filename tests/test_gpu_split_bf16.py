"""GPU parity tests of the opt-in split-bf16 ("bf16x3") data pass (pmf_set_precision, csrc/pmf_fused_sb.hip.inc)
against the fp64 CPU oracle, through the C ABI, on the same seeded inputs as the exact-f32 tests.

Tolerances: the same as tests/test_gpu_parity.py -- loss 2e-5 relative, gradients 2e-4 of max|gradient|, fitted factors
2e-3 after 10 epochs.  The forward uses the six-term product (2.3e-7 of max|Z| on MI355X, the same as the exact f32 MFMA:
scripts/bf16x3_probe.hip), the two gradient products the three-term one (4e-6).  A first version with a three-term
forward failed these tests at 9e-4: the problems are evaluated at the generating factors, where G = w (Z - D) is 0.1 and
|Z| is up to 45, so an error of 4e-6 |Z| in Z is 1e-3 of G.  Every test also checks that the split kernel really was the
one launched (pmf_get_precision counts its launches): a silent fall back to the exact kernel would pass parity."""
import numpy as np
import pytest

from problems import make_problem, rel_err, to_context, to_oracle
import test_gpu_parity as exact_tests
from test_gpu_parity import FIT_TOL, GRAD_TOL, LOSS_RTOL, grads_of

pytestmark = pytest.mark.gpu

CASES = {
    "ragged_k64_nan": dict(M=777, N=333, K=64, yreg="fsard", xreg="l2", nan_frac=0.1, weights=True, col_params=True),
    "k33_small": dict(M=40, N=50, K=33, col_params=True),
    "mixed_k48": dict(M=600, N=420, K=48, bernoulli_frac=0.25, poisson_frac=0.15, n_views=3, nan_frac=0.08, weights=True,
                      col_params=True, scale=0.4),
    # several row panels per workgroup and two column segments: the private gY slabs accumulate by read-modify-write
    "many_panels_k40": dict(M=70000, N=600, K=40, xreg="l2", weights=True, col_params=True, scale=0.5),
    "one_row_panel_many_cols": dict(M=33, N=1500, K=64, nan_frac=0.02),
    # two-tile segments and more work items than workgroups: consecutive pieces of a workgroup re-read private-slab
    # entries they stored a moment ago (the prefetch runs a tile ahead of the store)
    "many_panels_two_tiles": dict(M=70000, N=40, K=64, weights=True, scale=0.5),
    # K <= 32: the one-k-block variant (64-byte image rows, its own swizzle; the gY tile has half as many float4 as threads)
    "ragged_k32": dict(M=301, N=143, K=32, yreg="fsard", xreg="group", weights=True, col_params=True),
    "k10_pad": dict(M=260, N=70, K=10, yreg="ard", n_views=2, col_params=True),
    "tiny_k2": dict(M=5, N=7, K=2),
    "mixed_k17": dict(M=600, N=420, K=17, bernoulli_frac=0.25, poisson_frac=0.15, n_views=3, nan_frac=0.08, weights=True,
                      col_params=True, scale=0.4),
    "many_panels_k8": dict(M=70000, N=600, K=8, xreg="l2", weights=True, col_params=True),
    # batch scale / shift layers (BASELINE config 3 flavour) through the LDS-staged dense batch table
    "mixed_batch_nan_k32": dict(M=420, N=260, K=32, bernoulli_frac=0.2, n_views=2, batch_views=2, n_batches=8,
                                nan_frac=0.1, weights=True, col_params=True, xreg="composite", yreg="fsard"),
    "mixed_batch_nan_k64": dict(M=420, N=260, K=64, bernoulli_frac=0.2, n_views=2, batch_views=2, n_batches=8,
                                nan_frac=0.1, weights=True, col_params=True, scale=0.5),
    "poisson_batch_k8": dict(M=150, N=90, K=8, poisson_frac=0.3, bernoulli_frac=0.2, n_views=3, batch_views=2,
                             weights=True, col_params=True, scale=0.4),
    "batch_many_panels_k40": dict(M=30000, N=300, K=40, n_views=3, batch_views=3, n_batches=15, nan_frac=0.05, scale=0.5,
                                  col_params=True),
}


@pytest.fixture()
def sctx(ctx):
    ctx.set_precision("bf16x3")
    n0 = ctx.get_precision()[1]
    yield ctx, n0
    ctx.set_precision("f32")


@pytest.mark.parametrize("name", list(CASES))
def test_split_bf16_loss_and_gradients_match_oracle(sctx, name):
    ctx, n0 = sctx
    p = make_problem(seed=11, **CASES[name])
    to_context(p, ctx)
    loss, g = grads_of(ctx, p, update_X=True, update_Y=True)
    assert ctx.get_precision() == ("bf16x3", n0 + 1), "the split-bf16 kernel was not launched"
    m = to_oracle(p)
    m.m.n_xreg = 0
    m.m.n_yreg = 0
    _, gd = m.loss_and_grads(update_X=True, update_Y=True)
    assert abs(loss - gd["data_loss"]) <= LOSS_RTOL * abs(gd["data_loss"]) + 1e-6, (loss, gd["data_loss"])
    assert rel_err(g["X"], gd["X"]) <= GRAD_TOL, rel_err(g["X"], gd["X"])
    assert rel_err(g["Y"], gd["Y"]) <= GRAD_TOL, rel_err(g["Y"], gd["Y"])


@pytest.mark.parametrize("opt", ["adagrad", "adam"])
@pytest.mark.parametrize("name", ["ragged_k64_nan", "ragged_k32"])
def test_split_bf16_fit_trajectory_matches_oracle(sctx, opt, name):
    ctx, n0 = sctx
    p = make_problem(seed=13, random_init=True, **CASES[name])
    lr = 0.05 if opt == "adagrad" else 0.01
    to_context(p, ctx)
    ctx.set_optimizer(opt, lr=lr)
    r = ctx.fit(update_X=True, update_Y=True, max_epochs=10, abs_tol=0, rel_tol=0)
    assert ctx.get_precision()[1] == n0 + 10
    m = to_oracle(p)
    ro = m.fit(update_X=True, update_Y=True, opt=opt, lr=lr, max_epochs=10, abs_tol=0, rel_tol=0)
    assert r["term_code"] == ro["term_code"] and r["epochs"] == ro["epochs"]
    np.testing.assert_allclose(r["loss"], ro["loss"], rtol=5e-5)
    X, Y = ctx.get_factors()
    assert rel_err(X, m.X) <= FIT_TOL, rel_err(X, m.X)
    assert rel_err(Y, m.Y) <= FIT_TOL, rel_err(Y, m.Y)


def test_split_bf16_agrees_with_exact_kernel_at_config_size(sctx):
    """20000 x 10000, K = 64 (BASELINE configs[1] shape at the headline K): loss and both gradients of the split-bf16
    kernel against the exact-f32 kernel on the same device data, and gY bitwise reproducible run to run."""
    ctx, n0 = sctx
    M, N, K = 20000, 10000, 64
    rng = np.random.default_rng(23)
    ctx.set_data_device(None, M, N)
    ctx.set_factors((rng.standard_normal((K, M)) * 0.3).astype(np.float32), (rng.standard_normal((K, N)) * 0.3).astype(np.float32))
    ctx.set_col_params((rng.standard_normal(N) * 0.1).astype(np.float32), rng.standard_normal(N).astype(np.float32))
    ctx.set_batch_views([])
    ctx.set_noise([(1, N)], ["normal"], (0.5 + rng.random(N)).astype(np.float32))
    ctx.synth_data(seed=5, noise=0.5, frac_nan=0.03)
    o = ctx.make_opts(update_X=True, update_Y=True)

    def run():
        ctx.epoch_begin(o)
        loss, _ = ctx.epoch_loss()
        return loss, ctx.get_grad("X"), ctx.get_grad("Y")

    l1, gx1, gy1 = run()
    l2, _, gy2 = run()
    assert ctx.get_precision()[1] == n0 + 2
    assert l1 == l2 and np.array_equal(gy1, gy2)
    ctx.set_precision("f32")
    l0, gx0, gy0 = run()
    assert ctx.get_precision()[1] == n0 + 2
    assert abs(l1 - l0) <= LOSS_RTOL * abs(l0), (l1, l0)
    assert rel_err(gx1, gx0) <= GRAD_TOL, rel_err(gx1, gx0)
    assert rel_err(gy1, gy0) <= GRAD_TOL, rel_err(gy1, gy0)


def test_split_bf16_falls_back_to_exact_kernel_outside_its_scope(sctx):
    """A launch whose batch layers need the per-entry gather variant (a row panel that meets more than 15 batches of one
    view) has no split-bf16 variant: it must run the exact kernel (and say so through the launch counter), not fail."""
    ctx, n0 = sctx
    for case in (dict(M=300, N=100, K=16, n_views=2, batch_views=2, n_batches=20, nan_frac=0.05, col_params=True),
                 dict(M=300, N=100, K=80, n_views=2, batch_views=2, n_batches=20, nan_frac=0.05, col_params=True)):
        p = make_problem(seed=11, **case)
        to_context(p, ctx)
        loss, g = grads_of(ctx, p, update_X=True, update_Y=True)
        m = to_oracle(p)
        m.m.n_xreg = 0
        m.m.n_yreg = 0
        _, gd = m.loss_and_grads(update_X=True, update_Y=True)
        assert rel_err(g["Y"], gd["Y"]) <= GRAD_TOL
    assert ctx.get_precision()[1] == n0


@pytest.mark.parametrize("which", ["X", "Y"])
@pytest.mark.parametrize("name", ["ragged_k64_nan", "mixed_k48", "many_panels_k40", "ragged_k32", "many_panels_k8",
                                  "mixed_batch_nan_k32", "mixed_batch_nan_k64"])
def test_split_bf16_single_factor_gradient_matches_oracle(sctx, name, which):
    """grad(X)-only launches (transform: Y and the layers fixed, transform.jl) take the variant without GEMM3 / slabs,
    grad(Y)-only launches the one without GEMM2."""
    ctx, n0 = sctx
    p = make_problem(seed=13, **CASES[name])
    to_context(p, ctx)
    flags = dict(update_X=which == "X", update_Y=which == "Y")
    loss, g = grads_of(ctx, p, **flags)
    assert ctx.get_precision()[1] == n0 + 1
    m = to_oracle(p)
    m.m.n_xreg = 0
    m.m.n_yreg = 0
    _, go = m.loss_and_grads(**flags)
    assert abs(loss - go["data_loss"]) <= LOSS_RTOL * abs(go["data_loss"]) + 1e-6
    assert rel_err(g[which], go[which]) <= GRAD_TOL, rel_err(g[which], go[which])


def test_split_bf16_headline_size_properties(sctx):
    """The size-independent properties of tests/test_gpu_parity.py at BASELINE.json's headline configuration
    (200000 x 50000, K = 64: loss against the independent statistics kernel, gradient consistent with the loss by a
    Richardson-extrapolated directional derivative, bitwise reproducibility), through the split-bf16 kernel."""
    ctx, n0 = sctx
    exact_tests.test_headline_size_loss_and_gradient_consistency(ctx)
    assert ctx.get_precision()[1] == n0 + 4, "the split-bf16 kernel was not launched"


# ---- 64 < K <= 128: pmf_fused_sb4_kernel (four waves, X operands in registers; csrc/pmf_fused_sb4.hip.inc) -------------
CASES4 = {
    # 64 < K <= 96: the three-K-block instantiation of the same kernel (96-factor parameter rows, 256-byte image rows)
    "k65": dict(M=300, N=200, K=65, col_params=True, nan_frac=0.05),
    "k80_mixed_batch": dict(M=420, N=260, K=80, bernoulli_frac=0.2, n_views=2, batch_views=2, n_batches=8, nan_frac=0.1,
                            weights=True, col_params=True, scale=0.4),
    "k96_many_panels": dict(M=40000, N=600, K=96, xreg="l2", weights=True, col_params=True, scale=0.3),
    "k100": dict(M=200, N=150, K=100, yreg="group", xreg="l2", col_params=True),
    "k128_nan": dict(M=140, N=65, K=128, nan_frac=0.05, col_params=True),
    "k128_ragged": dict(M=777, N=333, K=128, yreg="fsard", xreg="l2", nan_frac=0.1, weights=True, col_params=True, scale=0.4),
    "k128_mixed": dict(M=600, N=420, K=128, bernoulli_frac=0.25, poisson_frac=0.15, n_views=3, nan_frac=0.08, weights=True,
                       col_params=True, scale=0.3),
    "k112_batch": dict(M=420, N=260, K=112, bernoulli_frac=0.2, n_views=2, batch_views=2, n_batches=8, nan_frac=0.1,
                       weights=True, col_params=True, scale=0.4),
    # several row panels per workgroup and several column segments: private-slab read-modify-write, many gX slots
    "k128_many_panels": dict(M=40000, N=600, K=128, xreg="l2", weights=True, col_params=True, scale=0.3),
    "k128_one_panel_many_cols": dict(M=33, N=1500, K=128, nan_frac=0.02, scale=0.4),
}


@pytest.mark.parametrize("name", list(CASES4))
def test_split_bf16_k128_loss_and_gradients_match_oracle(sctx, name):
    ctx, n0 = sctx
    p = make_problem(seed=11, **CASES4[name])
    to_context(p, ctx)
    loss, g = grads_of(ctx, p, update_X=True, update_Y=True)
    assert ctx.get_precision() == ("bf16x3", n0 + 1), "the split-bf16 kernel was not launched"
    m = to_oracle(p)
    m.m.n_xreg = 0
    m.m.n_yreg = 0
    _, gd = m.loss_and_grads(update_X=True, update_Y=True)
    assert abs(loss - gd["data_loss"]) <= LOSS_RTOL * abs(gd["data_loss"]) + 1e-6, (loss, gd["data_loss"])
    assert rel_err(g["X"], gd["X"]) <= GRAD_TOL, rel_err(g["X"], gd["X"])
    assert rel_err(g["Y"], gd["Y"]) <= GRAD_TOL, rel_err(g["Y"], gd["Y"])


@pytest.mark.parametrize("which", ["X", "Y"])
@pytest.mark.parametrize("name", ["k128_ragged", "k112_batch", "k80_mixed_batch"])
def test_split_bf16_k128_single_factor_gradient_matches_oracle(sctx, name, which):
    ctx, n0 = sctx
    p = make_problem(seed=13, **CASES4[name])
    to_context(p, ctx)
    flags = dict(update_X=which == "X", update_Y=which == "Y")
    loss, g = grads_of(ctx, p, **flags)
    assert ctx.get_precision()[1] == n0 + 1
    m = to_oracle(p)
    m.m.n_xreg = 0
    m.m.n_yreg = 0
    _, go = m.loss_and_grads(**flags)
    assert abs(loss - go["data_loss"]) <= LOSS_RTOL * abs(go["data_loss"]) + 1e-6
    assert rel_err(g[which], go[which]) <= GRAD_TOL, rel_err(g[which], go[which])


@pytest.mark.parametrize("opt", ["adagrad", "adam"])
def test_split_bf16_k128_fit_trajectory_matches_oracle(sctx, opt):
    ctx, n0 = sctx
    p = make_problem(seed=13, random_init=True, **CASES4["k128_ragged"])
    lr = 0.05 if opt == "adagrad" else 0.01
    to_context(p, ctx)
    ctx.set_optimizer(opt, lr=lr)
    r = ctx.fit(update_X=True, update_Y=True, max_epochs=10, abs_tol=0, rel_tol=0)
    assert ctx.get_precision()[1] == n0 + 10
    m = to_oracle(p)
    ro = m.fit(update_X=True, update_Y=True, opt=opt, lr=lr, max_epochs=10, abs_tol=0, rel_tol=0)
    assert r["term_code"] == ro["term_code"] and r["epochs"] == ro["epochs"]
    np.testing.assert_allclose(r["loss"], ro["loss"], rtol=5e-5)
    X, Y = ctx.get_factors()
    # AdaGrad's first steps are +-lr sign(g): on near-zero gradients the three-term gradient products' 4e-6 (of 128-term
    # dot products) becomes an O(lr) difference in a few entries; the max-norm tolerance is widened for it, the loss trace
    # above is not
    tol = 8 * FIT_TOL if opt == "adagrad" else 2 * FIT_TOL
    assert rel_err(X, m.X) <= tol, rel_err(X, m.X)
    assert rel_err(Y, m.Y) <= tol, rel_err(Y, m.Y)


# ---- the data matrix stored as bf16 (PMF_STORE_BF16; BASELINE configs[4] "D stored bf16") ------------------------------
def bf16_round(D):
    """float32 -> nearest bfloat16 (ties to even) -> float32, NaN kept: what k_tile_D's cast does on the device."""
    u = np.ascontiguousarray(D, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    out = r.astype(np.uint32).view(np.float32).reshape(D.shape)
    return np.where(np.isnan(D), np.float32(np.nan), out).astype(np.float32)


@pytest.mark.parametrize("name", ["ragged_k64_nan", "mixed_k48", "ragged_k32", "mixed_batch_nan_k64", "k128_ragged", "k112_batch",
                                  "k128_many_panels"])
def test_bf16_stored_data_matches_oracle_on_the_rounded_matrix(sctx, name):
    """Tolerance of bf16 storage: NONE beyond the rounding of D itself -- fed the same bf16-rounded matrix, the oracle must
    agree at the usual f32 tolerances (D enters the loss only through z - y and y z)."""
    ctx, n0 = sctx
    p = make_problem(seed=17, **(CASES[name] if name in CASES else CASES4[name]))
    p["D"] = np.asfortranarray(bf16_round(p["D"]))
    to_context(p, ctx)
    ctx.set_data(p["D"], store="bf16")
    loss, g = grads_of(ctx, p, update_X=True, update_Y=True)
    assert ctx.get_precision()[1] == n0 + 1
    m = to_oracle(p)
    m.m.n_xreg = 0
    m.m.n_yreg = 0
    _, gd = m.loss_and_grads(update_X=True, update_Y=True)
    assert abs(loss - gd["data_loss"]) <= LOSS_RTOL * abs(gd["data_loss"]) + 1e-6, (loss, gd["data_loss"])
    assert rel_err(g["X"], gd["X"]) <= GRAD_TOL, rel_err(g["X"], gd["X"])
    assert rel_err(g["Y"], gd["Y"]) <= GRAD_TOL, rel_err(g["Y"], gd["Y"])
    # the unrounded matrix through the same call gives the same device bytes (the library rounds to nearest even)
    st = ctx.stats(use_factors=True)
    so = m.stats(use_factors=True)
    np.testing.assert_array_equal(st["n"], so["n"].astype(np.float32))
    assert rel_err(st["sqerr"], so["sqerr"]) <= 1e-4


def test_bf16_stored_data_fit_and_layers(sctx):
    """A fit on bf16-stored data (X, Y through the split kernel; the theta stage through the scalar layer kernel, which
    reads either storage type) against the oracle on the rounded matrix."""
    ctx, n0 = sctx
    p = make_problem(seed=19, random_init=True, layer_regs=True, **CASES["mixed_batch_nan_k64"])
    p["D"] = np.asfortranarray(bf16_round(p["D"]))
    to_context(p, ctx)
    ctx.set_data(p["D"], store="bf16")
    ctx.set_optimizer("adagrad", lr=0.05)
    r = ctx.fit(update_X=True, update_Y=True, max_epochs=6, abs_tol=0, rel_tol=0)
    m = to_oracle(p)
    ro = m.fit(update_X=True, update_Y=True, lr=0.05, max_epochs=6, abs_tol=0, rel_tol=0)
    np.testing.assert_allclose(r["loss"], ro["loss"], rtol=5e-5)
    kw = dict(update_col_layers=True, frozen_layers=0b0111, max_epochs=4, abs_tol=0, rel_tol=0)
    ctx.set_optimizer("adagrad", lr=1.0)
    m.reset_optimizer()
    r2 = ctx.fit(**kw)
    ro2 = m.fit(lr=1.0, **kw)
    np.testing.assert_allclose(r2["loss"], ro2["loss"], rtol=5e-5)
    ctx.set_data(p["D"])            # back to f32 storage for the tests that follow


@pytest.mark.parametrize("name", ["ragged_k64_nan", "mixed_batch_nan_k32", "k128_ragged", "k100", "many_panels_k8"])
def test_bf16_stored_data_through_the_exact_kernel(ctx, name):
    """PMF_STORE_BF16 with the default precision: the exact-f32 kernels (data pass and MFMA layer pass) read the bf16
    tiles too (two 1-KiB loads per tile, converted in the epilogue).  Oracle on the rounded matrix, usual tolerances."""
    kw = dict(CASES[name] if name in CASES else CASES4[name])
    if name == "mixed_batch_nan_k32":
        kw["layer_regs"] = True
    p = make_problem(seed=23, **kw)
    p["D"] = np.asfortranarray(bf16_round(p["D"]))
    to_context(p, ctx)
    ctx.set_data(p["D"], store="bf16")
    try:
        n0 = ctx.get_precision()[1]
        loss, g = grads_of(ctx, p, update_X=True, update_Y=True)
        assert ctx.get_precision() == ("f32", n0)
        m = to_oracle(p)
        m.m.n_xreg = 0
        m.m.n_yreg = 0
        _, gd = m.loss_and_grads(update_X=True, update_Y=True)
        assert abs(loss - gd["data_loss"]) <= LOSS_RTOL * abs(gd["data_loss"]) + 1e-6, (loss, gd["data_loss"])
        assert rel_err(g["X"], gd["X"]) <= GRAD_TOL and rel_err(g["Y"], gd["Y"]) <= GRAD_TOL
        if p["batch_views"]:
            lossl, gl = grads_of(ctx, p, update_col_layers=True)
            ml = to_oracle(p)
            ml.m.has_colreg = 0
            ml.m.has_batchreg = 0
            _, go = ml.loss_and_grads(update_col_layers=True)
            assert abs(lossl - go["data_loss"]) <= LOSS_RTOL * abs(go["data_loss"])
            assert rel_err(gl["mu"], go["mu"]) <= GRAD_TOL
            for v in range(len(p["batch_views"])):
                assert rel_err(gl["theta"][v], go["theta"][v]) <= GRAD_TOL
                assert rel_err(gl["logdelta"][v], go["logdelta"][v]) <= GRAD_TOL
    finally:
        ctx.set_data(p["D"])
