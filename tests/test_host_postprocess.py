"""CPU tests of the host-side post-processing steps: whiten!, rotate_by_svd!, reorder_by_importance!,
reweight_eb! are small dense linear algebra on the parameter arrays; their defining invariants are checked."""
import numpy as np


def small_model(pkg, seed=0, K=4, M=30, N=24):
    rng = np.random.default_rng(seed)
    D = rng.standard_normal((M, N)).astype(np.float32)
    m = pkg.make_model(D, K=K, feature_views=[1] * 10 + [2] * 14, sample_conditions=["a"] * 12 + ["b"] * 18,
                       Y_ard=True, rng=rng)
    m.matfac.col_transform.layers[0].logsigma[...] = rng.standard_normal(N) * 0.2
    return m


def product(m):
    mf = m.matfac
    return (mf.X.astype(np.float64).T @ mf.Y.astype(np.float64)) * np.exp(mf.col_transform.layers[0].logsigma)[None, :]


def test_whiten_preserves_the_scaled_product_and_normalises(pkg):
    m = small_model(pkg)
    before = product(m)
    pkg.whiten_(m)
    np.testing.assert_allclose(product(m), before, rtol=2e-5, atol=1e-6)             # fit.jl:504-527: magnitude only moves
    np.testing.assert_allclose(np.sqrt(np.mean(m.matfac.X.astype(np.float64) ** 2, axis=1)), 1.0, rtol=1e-5)
    for cr in pkg.util.ids_to_ranges(m.feature_views):
        y = m.matfac.Y[:, cr.slice0()].astype(np.float64)
        assert abs(np.sqrt(np.mean(y * y, axis=1)).max() - 1.0) < 1e-5


def test_whiten_zero_view(pkg):
    m = small_model(pkg, seed=1)
    m.matfac.Y[:, :10] = 0
    pkg.whiten_(m)
    assert np.all(m.matfac.Y[:, :10] == 0) and np.all(m.matfac.col_transform.layers[0].logsigma[:10] == np.float32(-1e9))


def test_rotate_by_svd_preserves_product_and_orthogonalises_Y(pkg):
    m = small_model(pkg, seed=2)
    before = product(m)
    pkg.rotate_by_svd_(m)
    np.testing.assert_allclose(product(m), before, rtol=1e-4, atol=1e-5)             # fit.jl:530-543
    G = m.matfac.Y.astype(np.float64) @ m.matfac.Y.astype(np.float64).T
    assert np.allclose(G - np.diag(np.diag(G)), 0, atol=1e-4)                          # rows of S*Vt are orthogonal
    assert np.all(np.diff(np.diag(G)) <= 1e-6)                                         # singular values descending


def test_reorder_by_importance(pkg):
    m = small_model(pkg, seed=3)
    mf = m.matfac
    mf.Y[...] *= np.array([1.0, 3.0, 0.5, 2.0], dtype=np.float32)[:, None]
    X0, Y0 = mf.X.copy(), mf.Y.copy()
    mf.X_reg = pkg.regularizers.L2Regularizer(np.array([10, 20, 30, 40], dtype=np.float32))
    pkg.reorder_by_importance_(m)
    order = np.argsort(-np.sum(Y0.astype(np.float64) ** 2, axis=1), kind="stable")
    assert np.array_equal(mf.Y, Y0[order]) and np.array_equal(mf.X, X0[order])         # fit.jl:546-552
    assert np.array_equal(mf.X_reg.weights, np.array([10, 20, 30, 40], dtype=np.float32)[order])   # reorder_reg!


def test_reweight_eb(pkg):
    rng = np.random.default_rng(4)
    X = rng.standard_normal((3, 20))
    reg = pkg.regularizers.GroupRegularizer(["a"] * 8 + ["b"] * 12, K=3)
    pkg.reweight_eb_(reg, X)
    for g, w in zip(reg.group_idx, reg.group_weights):                                  # regularizers.jl:406-420
        s = np.linalg.svd(X[:, g.slice0()], compute_uv=False)
        np.testing.assert_allclose(w, 1.0 / s[0] ** 2, rtol=1e-6)
    l2 = pkg.regularizers.L2Regularizer(3, 1.0)
    pkg.reweight_eb_(l2, X)
    np.testing.assert_allclose(l2.weights, 1.0 / np.linalg.svd(X, compute_uv=False)[0] ** 2, rtol=1e-6)   # :39-47
    ard = pkg.regularizers.ARDRegularizer([1, 1, 2, 2])
    pkg.reweight_eb_(ard, X[:, :4])
    assert all(a == np.float32(0.001) for a in ard.alpha) and all(b == np.float32(0.001) for b in ard.beta)  # Q12


def test_update_lambda(pkg):
    rng = np.random.default_rng(5)
    reg = pkg.regularizers.construct_featureset_ard(3, list(range(1, 13)), [1] * 6 + [2] * 6,
                                                    [[[1, 2, 3], [4, 5, 6]], [[7, 8], [9, 10, 11, 12]]])
    Y = rng.standard_normal((3, 12)).astype(np.float32)
    pkg.update_lambda_(reg, Y)
    for v, cr in enumerate(reg.col_ranges):                                             # featureset_ard.jl:189-209
        yv = Y[:, cr.slice0()].astype(np.float64)
        ms = np.mean(yv * yv, axis=1)
        den = ms - min(ms.min(), 0.8) + 1e-3
        np.testing.assert_allclose(reg.lambda_[v], 2 * np.mean(reg.S[v]) / den, rtol=1e-5)
