"""The reference's own regularizer known answers (test/runtests.jl:739-792, 864-875) fed to the DEVICE: the host mirror of
the reference's constructors marshals each regularizer through the C ABI, one pmf_epoch_begin + pmf_epoch_step_shared at
lr -> 0 evaluates it on the GPU, and the regularizer's value (`shared_terms` of pmf_epoch_loss) and its gradient (recovered
exactly from Adam's first moment after one step from fresh state: m = (1 - beta1) g) are compared with the reference's
closed forms.  The data term is switched off (every entry of D missing, X = 0), as in the reference tests, which call the
regularizer alone.  The CPU oracle is pinned by the same vectors in tests/test_oracle_kat.py; here it is not involved."""
import json
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / "golden"
B1 = 0.9


def _bare_context(ctx, M, N, K, Y):
    ctx.set_data(np.full((M, N), np.nan, np.float32))          # no observed entry: data loss 0, data gradient 0
    ctx.set_factors(np.zeros((K, M), np.float32), np.asarray(Y, np.float32))
    ctx.set_col_params(np.zeros(N, np.float32), np.zeros(N, np.float32))
    ctx.set_batch_views([])
    ctx.set_noise([(1, N)], ["normal"], np.ones(N, np.float32))
    ctx.clear_xreg()
    ctx.clear_yreg()
    ctx.set_layer_regs()


def _value_and_grad_Y(ctx):
    ctx.set_optimizer("adam", lr=1e-30, beta1=B1)               # lr -> 0: the parameters do not move
    o = ctx.make_opts(update_Y=True)
    ctx.epoch_begin(o)
    ctx.epoch_step_shared(o)
    loss, shared = ctx.epoch_loss()
    _, mom = ctx.get_opt_state("Y")
    assert loss == shared                                        # nothing but the Y regularizer contributes
    return shared, mom.astype(np.float64) / (1.0 - B1)


def test_group_regularizer_literal(pkg, ctx):
    """runtests.jl:739-764: GroupRegularizer([1,1,1,2,2,2]; K=3): value 0.5*sum(Y.^2), gradient Y."""
    rng = np.random.default_rng(2)
    test_Y = rng.standard_normal((3, 6)).astype(np.float32)
    _bare_context(ctx, 4, 6, 3, test_Y)
    reg = pkg.regularizers.GroupRegularizer([1, 1, 1, 2, 2, 2], K=3)
    assert [(g.start, g.stop) for g in reg.group_idx] == [(1, 3), (4, 6)]
    reg.add_to(ctx, "Y")
    val, g = _value_and_grad_Y(ctx)
    Y64 = test_Y.astype(np.float64)
    assert val == pytest.approx(0.5 * np.sum(Y64 ** 2), rel=1e-6)                      # :750
    np.testing.assert_allclose(g, Y64, rtol=2e-6, atol=1e-7)                           # :761


def test_ard_regularizer_literal(pkg, ctx):
    """runtests.jl:766-777: ARDRegularizer([1,1,1,2,2]): (0.5 + alpha) sum(log(b)), b = 1 + (0.5/beta) Y.^2; gradient
    ((0.5 + alpha)/beta) Y ./ b."""
    rng = np.random.default_rng(3)
    K, N = 3, 5
    test_Y = rng.standard_normal((K, N)).astype(np.float32)
    _bare_context(ctx, 2, N, K, test_Y)
    reg = pkg.regularizers.ARDRegularizer([1, 1, 1, 2, 2])
    reg.add_to(ctx, "Y")
    val, g = _value_and_grad_Y(ctx)
    a, b0 = float(reg.alpha[0]), float(reg.beta[0])
    Y64 = test_Y.astype(np.float64)
    b = 1 + (0.5 / b0) * Y64 * Y64
    assert val == pytest.approx((0.5 + a) * np.sum(np.log(b)), rel=2e-6)              # :773
    np.testing.assert_allclose(g, ((0.5 + a) / b0) * Y64 / b, rtol=5e-6, atol=1e-6)   # :775


def test_batcharray_reg_literal(pkg, ctx):
    """runtests.jl:780-792, all numbers literal: value 0.5*sum(w .* v.^2) = 26.9744, gradient w .* v = v."""
    g = json.loads((GOLD / "batch_array_reg.json").read_text())
    vals = [np.array([d["1"], d["2"]], dtype=np.float32) for d in g["values"]]
    cr = [(1, 3), (4, 5), (6, 6)]
    rbs = [np.array(g["row_batches"][k], dtype=np.int32) - 1 for k in ("cat", "dog", "fish")]
    M, N, K = 5, 6, 2
    ctx.set_data(np.full((M, N), np.nan, np.float32))
    ctx.set_factors(np.zeros((K, M), np.float32), np.zeros((K, N), np.float32))
    ctx.set_col_params(np.zeros(N, np.float32), np.zeros(N, np.float32))
    ctx.set_batch_views([dict(start1=s, stop1=e, batch_of_row=rb, logdelta=np.zeros_like(v), theta=v)
                         for (s, e), rb, v in zip(cr, rbs, vals)])
    ctx.set_noise([(1, N)], ["normal"], np.ones(N, np.float32))
    ctx.clear_xreg()
    ctx.clear_yreg()
    ones = [np.full(2, g["weight"], np.float32) for _ in vals]
    zeros = [np.zeros(2, np.float32) for _ in vals]
    ctx.set_layer_regs(w_logdelta=ones, c_logdelta=zeros, w_theta=ones, c_theta=zeros)
    ctx.set_optimizer("adam", lr=1e-30, beta1=B1)
    o = ctx.make_opts(update_col_layers=True, frozen_layers=0b0111)     # only layer 4 (theta) and its regularizer live
    ctx.epoch_begin(o)
    ctx.epoch_step_shared(o)
    loss, shared = ctx.epoch_loss()
    assert shared == pytest.approx(g["loss"], rel=1e-6) and loss == shared            # 26.9744
    for v, want in enumerate(vals):
        _, mom = ctx.get_opt_state("theta", v)
        np.testing.assert_allclose(mom.astype(np.float64) / (1.0 - B1), want, rtol=2e-6, atol=1e-7)


def test_featureset_ard_literal(pkg, ctx):
    """runtests.jl:814-875: the regularizer built from the test's feature sets (beta = alpha0 - 1 everywhere, :853) equals the
    calibrated gamma-normal loss gnl(Y) - gnl(0), its gradient d gnl / dY = (alpha + 0.5) Y / (beta + 0.5 Y^2)."""
    f = json.loads((GOLD / "featureset_ard.json").read_text())
    K, N = f["K"], f["N"]
    reg = pkg.regularizers.construct_featureset_ard(K, f["feature_ids"], f["feature_views"], f["feature_sets"],
                                                    alpha0=np.float32(f["alpha0"]), v0=np.float32(f["v0"]))
    assert [tuple(A.shape) for A in reg.A] == [tuple(s) for s in f["A_shapes"]]
    np.testing.assert_allclose(reg.beta, np.float32(f["alpha0"]) - np.float32(1), rtol=1e-6)
    rng = np.random.default_rng(4)
    Y = (rng.standard_normal((K, N)) * 0.3).astype(np.float32)
    _bare_context(ctx, 2, N, K, Y)
    reg.add_to(ctx, "Y")
    val, g = _value_and_grad_Y(ctx)
    alpha = reg.alpha.astype(np.float64)
    beta = reg.beta.astype(np.float64)
    Y64 = Y.astype(np.float64)

    def gnl(Yv):                                                                       # :864
        return -np.sum(alpha[None, :] * np.log(beta)) + np.sum((alpha + 0.5)[None, :] * np.log(beta + 0.5 * Yv * Yv))
    assert val == pytest.approx(gnl(Y64) - gnl(np.zeros_like(Y64)), rel=5e-6)          # :869-871
    np.testing.assert_allclose(g, (alpha + 0.5)[None, :] * Y64 / (beta + 0.5 * Y64 * Y64), rtol=1e-5, atol=1e-5)   # :873-874
