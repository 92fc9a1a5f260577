"""SURVEY row N4 on the CPU: the pipeline's data file (fit_matfac.jl:60-103, script_util.jl) -> make_model; the parameter file
in both layouts; whiten! / rotate_by_svd! / reorder_by_importance! (src/fit.jl:504-555) of the product (fit.py) against the
independent restatement in oracle/postprocess_oracle.py, including the all-zero-view branch."""
import numpy as np
import pytest

import test_gpu_host as th
from oracle import postprocess_oracle as po


def _toy_file(tmp_path, layout):
    rng = np.random.default_rng(0)
    M = 12
    assays = ["mrnaseq"] * 5 + ["mutation"] * 3 + ["methylation"] * 4 + ["rppa"] * 2
    genes = [f"G{i}" for i in range(len(assays))]
    D = rng.standard_normal((M, len(assays))).astype(np.float32)
    D[2, 3] = np.nan
    D[:, 7] = (D[:, 7] > 0)
    ids = [f"S{i}" for i in range(M)]
    groups = ["BRCA", "LUAD", "BRCA", "LUAD", "COAD", "BRCA", "COAD", "LUAD", "BRCA", "COAD", "LUAD", "BRCA"]
    bfeat = ["mrnaseq", "methylation", "rppa"]
    bc = np.array([[f"TCGA-{'AB'[i % 2]}{i % 3}-01A-11R-{1000 + (i % 3)}-07", f"TCGA-XX-02-{2000 + (i % 2)}-05" if i % 4 else "", f"TCGA-{i}-9-9"]
                   for i in range(M)])
    path = tmp_path / f"omic_{layout}.npz"
    import pmf_import
    data_io = pmf_import.load().data_io
    data_io.save_omic_npz(path, assays, genes, ids, groups, D, barcodes=bc, barcode_features=bfeat, layout=layout)
    return path, D, assays, genes, ids, groups, bc


def test_barcode_to_batch_is_the_last_two_terms(pkg):
    barcode_to_batch = pkg.data_io.barcode_to_batch
    assert barcode_to_batch("TCGA-A1-01A-11R-1000-07") == "1000-07"          # script_util.jl:147-157
    assert barcode_to_batch("") == "" and barcode_to_batch("x") == "x" and barcode_to_batch("a-b") == "a-b"


@pytest.mark.parametrize("layout", ["h5py", "julia"])
def test_data_file_loader(pkg, tmp_path, layout):
    data_io = pkg.data_io
    path, D, assays, genes, ids, groups, bc = _toy_file(tmp_path, layout)
    z = np.load(path, allow_pickle=False)
    assert set(data_io.OMIC_KEYS) | set(data_io.BARCODE_KEYS) | {"layout"} == set(z.files)
    assert z["omic_data/data"].shape == ((len(assays), 12) if layout == "h5py" else (12, len(assays)))   # what h5py would show
    omic, sid, cond, fg, fa = data_io.load_omic_data(path, ["mrnaseq", "methylation"])                 # fit_matfac.jl:60-82
    keep = [i for i, a in enumerate(assays) if a in ("mrnaseq", "methylation")]
    np.testing.assert_array_equal(omic, D[:, keep])
    assert list(fa) == [assays[i] for i in keep] and list(fg) == [genes[i] for i in keep]
    assert list(sid) == ids and list(cond) == groups
    b = data_io.load_batches(path, ["mrnaseq", "mutation", "methylation"])                             # :85-103
    assert set(b) == {"mrnaseq", "methylation"}                                                        # BATCHED_ASSAYS only
    assert b["mrnaseq"] == [data_io.barcode_to_batch(x) for x in bc[:, 0]]
    assert b["methylation"][0] == "" and b["methylation"][1] == "2001-05"
    assert data_io.load_batches(path, ["mutation"]) is None
    nf = data_io.nan_fractions(omic, fa)
    assert nf["mrnaseq"] == pytest.approx(1 / 60) and nf["methylation"] == 0.0


def test_var_filter_and_model_from_data_file(pkg, tmp_path):
    data_io = pkg.data_io
    path, D, assays, genes, ids, groups, bc = _toy_file(tmp_path, "h5py")
    cv = data_io.column_variances(D)
    col = D[:, 0].astype(np.float64)
    assert cv[0] == pytest.approx(np.sum(col * col) - np.mean(col) ** 2)            # script_util.jl:39 as coded
    keep = data_io.var_filter(D, assays, 0.5)                                        # top half per assay by that variance
    for a in set(assays):
        idx = [i for i, x in enumerate(assays) if x == a]
        thr = np.quantile(cv[idx], 0.5)
        assert [i for i in idx if cv[i] >= thr] == [i for i in keep if assays[i] == a]
    model = data_io.model_from_data_file(path, ["mrnaseq", "methylation", "mutation"], K=3,
                                         distribution_map={"mutation": "bernoulli"}, rng=np.random.default_rng(1))
    srt = np.argsort(np.array(groups), kind="stable")                                # fit_matfac.jl:212-216
    assert list(model.sample_ids) == [ids[i] for i in srt] and list(model.sample_conditions) == [groups[i] for i in srt]
    # columns grouped by (distribution, view) (src/model.jl:50-54): bernoulli first, then normal by view
    assert list(model.feature_views) == ["mutation"] * 3 + ["methylation"] * 4 + ["mrnaseq"] * 5
    assert model.feature_ids[0].endswith("_mutation") and model.data.shape == (12, 12)
    raw = D[srt][:, [i for i, a in enumerate(assays) if a in ("mrnaseq", "methylation", "mutation")]]
    np.testing.assert_array_equal(model.data, raw[:, np.asarray(model.data_idx) - 1])
    l4 = model.matfac.col_transform.unwrapped(4)                                     # batch layers on the batched assays only
    assert [len(c) for c in l4.theta.col_ranges] == [4, 5]


@pytest.mark.parametrize("layout", ["h5py", "julia"])
def test_parameter_file_layouts(pkg, tmp_path, layout):
    model = th.reference_fit_setup(pkg, seed=5)
    rng = np.random.default_rng(2)
    for A in model.matfac.Y_reg.A:
        A[...] = np.abs(rng.standard_normal(A.shape)).astype(np.float32)
    path = tmp_path / "p.npz"
    pkg.save_params_npz(model, path, layout=layout)
    z = np.load(path, allow_pickle=False)
    K, M, N = 4, 40, 60
    assert str(z["layout"]) == layout
    if layout == "h5py":     # what `f["Y"][:,:]` gives a Python reader of the reference's HDF5: transposed
        assert z["X"].shape == (M, K) and z["Y"].shape == (N, K) and z["theta/values_1"].shape == (30, 4)
        np.testing.assert_array_equal(z["Y"][:, :].transpose(), model.matfac.Y)      # vis_factors.py:94
    else:
        assert z["X"].shape == (K, M) and z["Y"].shape == (K, N)
    other = th.reference_fit_setup(pkg, seed=6)
    pkg.load_params_npz(other, path)
    np.testing.assert_array_equal(other.matfac.Y, model.matfac.Y)
    reg = other.matfac.Y_reg                                                         # beta follows A (featureset_ard.jl:292)
    for Av, Sv, cr in zip(reg.A, reg.S, reg.col_ranges):
        np.testing.assert_allclose(reg.beta[:, cr.slice0()], (reg.alpha0 - 1) * (reg.v0 + Av.T @ Sv), rtol=1e-6)
    bad = th.reference_fit_setup(pkg, seed=6)
    bad.data_idx = np.asarray(bad.data_idx)[::-1].copy()
    with pytest.raises(ValueError, match="data_idx"):
        pkg.load_params_npz(bad, path)
    bad2 = th.reference_fit_setup(pkg, seed=6)
    bad2.feature_views = list(bad2.feature_views)[::-1]
    with pytest.raises(ValueError, match="feature views"):
        pkg.load_params_npz(bad2, path)


def _fitted_like(pkg, seed, zero_view=False):
    rng = np.random.default_rng(seed)
    M, N, K = 30, 24, 4
    D = rng.standard_normal((M, N)).astype(np.float32)
    fs = {1: [[1, 2, 3], [4, 5, 6, 7]], 2: [[11, 12], [13, 14, 15, 16, 17]]}
    m = pkg.make_model(D, K=K, feature_views=[1] * 10 + [2] * 14, sample_conditions=["a"] * 12 + ["b"] * 18, Y_fsard=True,
                       feature_sets_dict=fs, rng=rng)
    mf = m.matfac
    mf.X[...] = (rng.standard_normal((K, M)) * np.array([0.5, 2.0, 1.0, 0.1])[:, None]).astype(np.float32)
    mf.Y[...] = (rng.standard_normal((K, N)) * np.array([1.0, 0.2, 3.0, 0.7])[:, None]).astype(np.float32)
    if zero_view:
        mf.Y[:, :10] = 0
    mf.col_transform.layers[0].logsigma[...] = (rng.standard_normal(N) * 0.2).astype(np.float32)
    reg = mf.Y_reg
    reg.beta[...] = np.abs(rng.standard_normal(reg.beta.shape)).astype(np.float32)
    for A in reg.A:
        A[...] = np.abs(rng.standard_normal(A.shape)).astype(np.float32)
    reg.lambda_ = tuple(np.abs(rng.standard_normal(K)).astype(np.float32) for _ in reg.A)
    return m


@pytest.mark.parametrize("zero_view", [False, True])
def test_postprocessing_matches_the_independent_restatement(pkg, zero_view):
    m = _fitted_like(pkg, 7, zero_view)
    mf = m.matfac
    vr = [(cr.start, cr.stop) for cr in pkg.util.ids_to_ranges(m.feature_views)]
    X0, Y0, ls0 = mf.X.copy(), mf.Y.copy(), mf.col_transform.layers[0].logsigma.copy()
    # ---- whiten!
    Xo, Yo, lso = po.whiten(X0, Y0, ls0, vr)
    pkg.whiten_(m)
    np.testing.assert_allclose(mf.X, Xo, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(mf.Y, Yo, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(mf.col_transform.layers[0].logsigma, lso, rtol=2e-6, atol=1e-6)
    if zero_view:
        assert np.all(mf.Y[:, :10] == 0) and np.all(mf.col_transform.layers[0].logsigma[:10] == np.float32(-1e9))
    # ---- rotate_by_svd!  (signs of the singular vectors are LAPACK's: compare after a per-factor joint sign alignment)
    X1, Y1 = mf.X.copy(), mf.Y.copy()
    Xo, Yo = po.rotate_by_svd(X1, Y1)
    pkg.rotate_by_svd_(m)
    Xa, Ya = po.align_signs(Xo, Yo, mf.X, mf.Y)
    scale = np.abs(Ya).max()
    assert np.abs(mf.Y - Ya).max() <= 2e-5 * scale and np.abs(mf.X - Xa).max() <= 2e-5 * np.abs(Xa).max()
    np.testing.assert_allclose(mf.X.astype(np.float64).T @ mf.Y, X1.astype(np.float64).T @ Y1, atol=2e-5 * scale)
    # ---- reorder_by_importance! with reorder_reg! on FeatureSetARD (Y) and Group (X)
    mf.Y[...] *= np.array([0.01, 3.0, 0.5, 20.0], dtype=np.float32)[:, None]         # (rotation left them sorted already)
    reg, xreg = mf.Y_reg, mf.X_reg
    gw = [w.copy() for w in getattr(xreg, "group_weights", ())]
    want = po.reorder_by_importance(mf.X.copy(), mf.Y.copy(), group_weights=gw or None,
                                    fsard=dict(beta=reg.beta.copy(), A=[A.copy() for A in reg.A], **{"lambda": [l.copy() for l in reg.lambda_]}))
    pkg.reorder_by_importance_(m)
    assert want["order"] != [0, 1, 2, 3] and sorted(want["order"]) == [0, 1, 2, 3]
    ssq = np.sum(want["Y"].astype(np.float64) ** 2, axis=1)
    assert np.all(np.diff(ssq) <= 0)
    np.testing.assert_array_equal(mf.X, want["X"])
    np.testing.assert_array_equal(mf.Y, want["Y"])
    np.testing.assert_array_equal(reg.beta, want["fsard"]["beta"])
    for a, b in zip(reg.A, want["fsard"]["A"]):
        np.testing.assert_array_equal(a, b)
    for a, b in zip(reg.lambda_, want["fsard"]["lambda"]):
        np.testing.assert_array_equal(a, b)
    if gw:
        for a, b in zip(xreg.group_weights, want["group_weights"]):
            np.testing.assert_array_equal(a, b)


def test_reorder_ties_keep_original_order():
    Y = np.array([[1.0, 0.0], [0.0, 2.0], [1.0, 0.0], [2.0, 0.0]])
    assert po.importance_order(Y) == [1, 3, 0, 2]        # ssq = 1, 4, 1, 4: stable among equals
