"""Parameter interchange (.npz with the dataset keys of analyses/scripts/julia/bson_to_hdf.jl:18-71), CPU only."""
import numpy as np

import test_gpu_host as th


def test_npz_has_the_reference_hdf5_keys_and_round_trips(pkg, tmp_path):
    model = th.reference_fit_setup(pkg, seed=5)
    rng = np.random.default_rng(1)
    for v in model.matfac.col_transform.unwrapped(4).theta.values:
        v[...] = rng.standard_normal(v.shape).astype(np.float32)     # (the file holds float32, like the GPU model)
    d = pkg.model_io.model_to_dict(model)
    want = {"feature_ids", "feature_views", "sample_ids", "sample_conditions", "data_idx", "X", "Y", "logsigma", "mu",
            "logdelta/values_1", "logdelta/col_range_1", "logdelta/values_2", "logdelta/col_range_2",
            "theta/values_1", "theta/col_range_1", "theta/batch_ids_1", "theta/values_2", "theta/col_range_2",
            "theta/batch_ids_2", "fsard/A/1", "fsard/S/1", "fsard/A/2", "fsard/S/2"}
    assert set(d) == want
    K, M, N = 4, 40, 60
    assert d["X"].shape == (K, M) and d["Y"].shape == (K, N)                      # Julia's shapes
    assert d["theta/values_1"].shape == (4, 30) and list(d["theta/col_range_1"]) == list(range(1, 31))
    assert list(d["theta/col_range_2"]) == list(range(31, 61))                    # collect(cr): 1-based, inclusive
    assert list(d["theta/batch_ids_1"]) == ["rowbatch1", "rowbatch2", "rowbatch3", "rowbatch4"]
    assert d["fsard/S/1"].shape[1] == 30 and d["fsard/A/1"].shape == (d["fsard/S/1"].shape[0], K)
    path = tmp_path / "params.npz"
    pkg.save_params_npz(model, path)
    other = th.reference_fit_setup(pkg, seed=6)
    assert not np.array_equal(other.matfac.X, model.matfac.X)
    pkg.load_params_npz(other, path)
    for a, b in ((other.matfac.X, model.matfac.X), (other.matfac.Y, model.matfac.Y)):
        np.testing.assert_array_equal(a, b)
    for a, b in zip(other.matfac.col_transform.unwrapped(4).theta.values, model.matfac.col_transform.unwrapped(4).theta.values):
        np.testing.assert_array_equal(a, b)
    z = np.load(path, allow_pickle=False)                                         # plain arrays: loads without pickle
    assert z["sample_conditions"].dtype.kind == "U"
