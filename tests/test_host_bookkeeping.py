"""Index / ID bookkeeping of the host package against the reference's known answers (bit-exact)."""
import json
from pathlib import Path

import numpy as np
import pytest

GOLD = Path(__file__).resolve().parent / "golden"


def load(name):
    return json.loads((GOLD / name).read_text())


def R(pairs):
    return [tuple(p) for p in pairs]


def test_util_known_answers(pkg):
    u = pkg.util
    g = load("util_index.json")
    for c in g["is_contiguous"]:
        assert u.is_contiguous(c["in"]) is c["out"]
    for c in g["ids_to_ranges"]:
        assert [tuple(r) for r in u.ids_to_ranges(c["in"])] == R(c["out"])
    for c in g["subset_ranges"]:
        new, lo, hi = u.subset_ranges(R(c["ranges"]), tuple(c["rng"]))
        assert [tuple(r) for r in new] == R(c["out"][0]) and (lo, hi) == (c["out"][1], c["out"][2])
    ind = u.ids_to_ind_mat(g["ids_to_ind_mat"]["in"])
    assert ind.dtype == bool and np.array_equal(ind.astype(int), np.array(g["ids_to_ind_mat"]["out"]))
    assert u.value_to_idx(g["value_to_idx"]["in"]) == g["value_to_idx"]["out"]
    k = g["keymatch"]
    assert u.keymatch(k["l_keys"], k["r_keys"]) == (k["l_idx"], k["r_idx"])
    x = [np.nan if v is None else v for v in g["nanstats"]["in"]]
    assert u.nansum(x) == g["nanstats"]["nansum"] and u.nanmean(x) == g["nanstats"]["nanmean"]
    assert u.nanvar(x) == g["nanstats"]["nanvar"]
    assert u.nansum([1.0, np.nan, 2.0, np.inf, 3.0]) == np.inf          # runtests.jl:60


def test_subset_ranges_edge_cases(pkg):
    u = pkg.util
    assert u.subset_ranges([], (1, 5)) == ([], 1, 0)                    # util.jl:216-218
    assert u.subset_ranges([(3, 4)], (5, 9)) == ([], 1, 0)              # util.jl:222-224
    assert u.subset_ranges([(1, 2), (5, 6), (8, 10)], (3, 4)) == ([], 1, 0)   # falls entirely into a gap
    new, lo, hi = u.subset_ranges([(1, 2), (5, 6), (8, 10)], (2, 9))
    assert [tuple(r) for r in new] == [(2, 2), (5, 6), (8, 9)] and (lo, hi) == (1, 3)
    with pytest.raises(AssertionError):
        u.ids_to_ranges([1, 1, 5, 5, 1])                                  # util.jl:189


def test_batch_array_constructor_and_views(pkg):
    g = load("batch_array_5x7.json")
    inp = g["inputs"]
    vd = [{int(k): v for k, v in d.items()} for d in inp["values"]]
    ba = pkg.BatchArray.from_views(inp["col_batches"], inp["row_batches"], vd)
    assert [tuple(r) for r in ba.col_ranges] == R(g["ctor"]["col_ranges"])            # runtests.jl:139
    for got, want in zip(ba.row_batches_dense(), g["ctor"]["row_batches"]):
        assert np.array_equal(got.astype(int), np.array(want))                          # :140-143
    for got, want in zip(ba.values, g["ctor"]["values"]):
        assert np.array_equal(got, np.array(want))                                       # :144-146
    # view(ba, 2:4, 2:6)   (:151-163) -- also with an index vector (:165-174)
    gv = g["view_2to4_2to6"]
    for rows in ((2, 4), [2, 3, 4]):
        v = ba.view(rows, tuple(gv["cols"]))
        assert [tuple(r) for r in v.col_ranges] == R(gv["col_ranges"])
        for got, want in zip(v.row_batches_dense(), gv["row_batches"]):
            assert np.array_equal(got.astype(int)[:, :np.array(want).shape[1]], np.array(want))
        assert np.array_equal(v.row_selector_dense(5), np.array(gv["row_selector"]))
        for got, want in zip(v.values, gv["values"]):
            assert np.array_equal(got, np.array(want))
    vv = ba.view((2, 4), (2, 6)).view((1, 2), (1, 3))                                   # nested view (:160)
    assert [tuple(r) for r in vv.col_ranges] == [(1, 2)]
    # gaps (:178-189)
    gg = g["gappy"]
    gba = pkg.BatchArray.from_views(gg["col_batches"], inp["row_batches"], vd)
    assert [tuple(r) for r in gba.col_ranges] == R(gg["col_ranges"])
    ev = gba.view(None, tuple(gg["empty_view_cols"]))
    assert ev.col_ranges == () and ev.row_batches == () and ev.values == ()
    # zero (:193-199)
    z = ba.zero()
    assert z.col_ranges == ba.col_ranges
    for got, want in zip(z.values, g["zero_values"]):
        assert np.array_equal(got, np.array(want))


def test_featureset_ard_constructor(pkg):
    g = load("featureset_ard.json")
    reg = pkg.regularizers.construct_featureset_ard(g["K"], g["feature_ids"], g["feature_views"], g["feature_sets"],
                                                    featureset_ids=None, alpha0=g["alpha0"], v0=g["v0"], lr=0.1)
    assert [tuple(r) for r in reg.col_ranges] == R(g["col_ranges"])                    # runtests.jl:849
    assert list(reg.featureset_ids) == g["featureset_ids"]                              # :850
    assert reg.alpha0 == np.float32(g["alpha0"]) and reg.v0 == np.float32(g["v0"])      # :851-852
    assert np.all(reg.beta == np.float32(np.float32(g["alpha0"]) - np.float32(1)))      # :853
    assert [list(A.shape) for A in reg.A] == g["A_shapes"]                              # :855-856
    for v, sets in enumerate(g["feature_sets"]):                                        # :833-839, :860-861
        S = np.zeros((len(sets), 20))
        for l, s in enumerate(sets):
            S[l, np.array(s) - 1 - 20 * v] = 1 / np.sqrt(len(s))
        np.testing.assert_allclose(reg.S[v], S, rtol=1e-6)


def test_model_assembly_sorts_columns_by_distribution_and_view(pkg):
    """model.jl:50-54: sortperm(zip(distributions, views)) -- contiguous noise ranges, permutation kept in data_idx."""
    rng = np.random.default_rng(0)
    D = rng.standard_normal((20, 12)).astype(np.float32)
    views = ["b"] * 4 + ["a"] * 8
    dists = ["normal"] * 6 + ["bernoulli"] * 6
    m = pkg.make_model(D, K=3, feature_views=views, feature_distributions=dists,
                       sample_conditions=["x"] * 10 + ["y"] * 10, batch_dict={"a": [1] * 5 + [2] * 15}, rng=rng)
    assert list(m.data_idx) == [7, 8, 9, 10, 11, 12, 5, 6, 1, 2, 3, 4]
    assert np.array_equal(m.data, D[:, np.array(m.data_idx) - 1])
    nm = m.matfac.noise_model
    assert [tuple(r) for r in nm.col_ranges] == [(1, 6), (7, 12)] and nm.noises == ("bernoulli", "normal")
    ct = m.matfac.col_transform
    assert [type(l).__name__ for l in ct.layers] == ["ColScale", "BatchScale", "ColShift", "BatchShift"]   # runtests.jl:431-434
    assert [tuple(r) for r in ct.layers[1].logdelta.col_ranges] == [(1, 8)]
    assert ct.layers[1].logdelta.values[0].shape == (2, 8)
    # freeze / unfreeze (layers.jl:337-363; runtests.jl:443-451)
    pkg.layers.freeze_layer_(ct, [1, 2, 3])
    assert ct.frozen_mask() == 0b0111
    pkg.layers.unfreeze_layer_(ct, [1, 2, 3])
    assert ct.frozen_mask() == 0
    # no batch_dict -> layers 2 and 4 are functions (runtests.jl:417-422)
    m2 = pkg.make_model(D, K=3, feature_views=views, rng=rng)
    assert [type(l).__name__ for l in m2.matfac.col_transform.layers] == ["ColScale", "Identity", "ColShift", "Identity"]
    # validation errors (model.jl:135-184)
    with pytest.raises(AssertionError):
        pkg.make_model(D, sample_conditions=["x", "y"] * 10)
    with pytest.raises(AssertionError):
        pkg.make_model(D, feature_distributions=["gaussian"] * 12)


def test_row_sharding_is_a_partition(pkg):
    sh = pkg.parallel.shard_rows
    for M, W in ((200000, 8), (10, 3), (7, 8), (1, 1)):
        bounds = [sh(M, W, r) for r in range(W)]
        assert bounds[0][0] == 0 and bounds[-1][1] == M
        assert all(bounds[i][1] == bounds[i + 1][0] for i in range(W - 1))
        sizes = [b - a for a, b in bounds]
        assert max(sizes) - min(sizes) <= 1
