"""Parameter interchange of a fitted model.

The reference's own checkpoint is a Julia-object BSON (src/model_io.jl: out of scope, Julia-only); what its analysis
pipeline exchanges with Python / R is the flat HDF5 written by analyses/scripts/julia/bson_to_hdf.jl:18-71.  h5py is not
installed in this image, so the container here is a NumPy .npz with EXACTLY the dataset keys of that HDF5 file
("logdelta/values_1", "theta/batch_ids_2", "fsard/A/1", ...: 1-based indices as Julia writes them), column ranges as the
collected 1-based indices (`collect(cr)`, bson_to_hdf.jl:45).

LAYOUT.  HDF5.jl stores a Julia matrix with its dimensions reversed; the reference's Python consumers therefore see "X" as
M x K and "Y" as N x K and transpose them (analyses/scripts/python/vis_factors.py:94, score_matfac.py:185-194:
`f["Y"][:,:].transpose()`).  The file written here carries an explicit entry `layout`:
    "h5py"  (default)  every matrix as h5py would expose the reference's file -- X is M x K, Y is N x K, batch values
                       N_view x n_batches, fsard/A K x L, fsard/S N_view x L -- so a reader ported from those tools works as is;
    "julia"            matrices in Julia's logical shape (X is K x M, ...).
`model_to_dict` returns Julia's shapes (the in-memory convention of this package); `load_params_npz` reads either layout.
"""
import numpy as np

from .layers import BatchScale, BatchShift
from .regularizers import FeatureSetARDReg


def _strs(v):
    return np.asarray([str(x) for x in v])


def model_to_dict(model):
    """The datasets of bson_to_hdf.jl's write_model_to_hdf, keyed by their HDF5 paths."""
    mf = model.matfac
    ct = mf.col_transform
    d = {
        "feature_ids": _strs(model.feature_ids), "feature_views": _strs(model.feature_views),          # :28-29
        "sample_ids": _strs(model.sample_ids), "sample_conditions": _strs(model.sample_conditions),    # :30-31
        "data_idx": np.asarray(model.data_idx, dtype=np.int64),                                        # :32
        "X": np.asarray(mf.X, np.float32), "Y": np.asarray(mf.Y, np.float32),                          # :36-37
        "logsigma": np.asarray(ct.unwrapped(1).logsigma, np.float32),                                  # :38
        "mu": np.asarray(ct.unwrapped(3).mu, np.float32),                                              # :39
    }
    l2, l4 = ct.unwrapped(2), ct.unwrapped(4)
    if isinstance(l2, BatchScale):                                                                      # :43-49
        for i, (v, cr) in enumerate(zip(l2.logdelta.values, l2.logdelta.col_ranges), start=1):
            d[f"logdelta/values_{i}"] = np.asarray(v, np.float32)
            d[f"logdelta/col_range_{i}"] = np.arange(cr.start, cr.stop + 1, dtype=np.int64)
    if isinstance(l4, BatchShift):                                                                      # :51-59
        th = l4.theta
        for i, (v, cr, ids) in enumerate(zip(th.values, th.col_ranges, th.row_batch_ids), start=1):
            d[f"theta/values_{i}"] = np.asarray(v, np.float32)
            d[f"theta/col_range_{i}"] = np.arange(cr.start, cr.stop + 1, dtype=np.int64)
            d[f"theta/batch_ids_{i}"] = _strs(ids)
    if isinstance(mf.Y_reg, FeatureSetARDReg):                                                          # :63-68
        for i, (A, S) in enumerate(zip(mf.Y_reg.A, mf.Y_reg.S), start=1):
            d[f"fsard/A/{i}"] = np.asarray(A, np.float32)
            d[f"fsard/S/{i}"] = np.asarray(S, np.float32)
    return d


def save_params_npz(model, path, layout="h5py"):
    if layout not in ("h5py", "julia"):
        raise ValueError("layout must be 'h5py' or 'julia'")
    d = model_to_dict(model)
    if layout == "h5py":
        d = {k: (np.ascontiguousarray(v.T) if v.ndim == 2 else v) for k, v in d.items()}
    d["layout"] = np.asarray(layout)
    np.savez(path, **d)


def load_params_npz(model, path):
    """Puts the arrays of a parameter file back into a model of the same structure (same samples, features, views, column
    permutation, batches).  After A is restored, FeatureSetARD's beta is recomputed as update_A! leaves it:
    beta[:, cr] = (alpha0 - 1) (v0 + A'S)  (src/featureset_ard.jl:292)."""
    z = np.load(path, allow_pickle=False)
    lay = str(z["layout"]) if "layout" in z else "julia"
    if lay not in ("h5py", "julia"):
        raise ValueError(f"unknown layout {lay!r}")

    def get(key):
        a = z[key]
        return a.T if (lay == "h5py" and a.ndim == 2) else a
    mf = model.matfac
    ct = mf.col_transform
    if list(_strs(model.feature_ids)) != list(z["feature_ids"]) or list(_strs(model.sample_ids)) != list(z["sample_ids"]):
        raise ValueError("parameter file belongs to a model with other sample / feature ids")
    if list(_strs(model.feature_views)) != list(z["feature_views"]):
        raise ValueError("parameter file belongs to a model with other feature views")
    if not np.array_equal(np.asarray(model.data_idx, dtype=np.int64), np.asarray(z["data_idx"], dtype=np.int64)):
        raise ValueError("parameter file belongs to a model with another column permutation (data_idx)")
    if get("X").shape != mf.X.shape or get("Y").shape != mf.Y.shape:
        raise ValueError(f"factor shapes {get('X').shape}, {get('Y').shape} do not match the model's {mf.X.shape}, {mf.Y.shape} "
                         f"(layout {lay!r})")
    mf.X[...] = get("X")
    mf.Y[...] = get("Y")
    ct.unwrapped(1).logsigma[...] = z["logsigma"]
    ct.unwrapped(3).mu[...] = z["mu"]
    l2, l4 = ct.unwrapped(2), ct.unwrapped(4)
    if isinstance(l2, BatchScale):
        for i, v in enumerate(l2.logdelta.values, start=1):
            v[...] = get(f"logdelta/values_{i}")
    if isinstance(l4, BatchShift):
        for i, v in enumerate(l4.theta.values, start=1):
            v[...] = get(f"theta/values_{i}")
    reg = mf.Y_reg
    if isinstance(reg, FeatureSetARDReg):
        A = list(reg.A)
        for i in range(len(A)):
            A[i][...] = get(f"fsard/A/{i + 1}")
        beta0 = np.float32(reg.alpha0) - np.float32(1)
        for Av, Sv, cr in zip(reg.A, reg.S, reg.col_ranges):
            reg.beta[:, cr.slice0()] = beta0 * (np.float32(reg.v0) + np.asarray(Av, np.float32).T @ np.asarray(Sv, np.float32))
    return model
