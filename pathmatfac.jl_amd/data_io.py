"""The data file of the reference's analysis pipeline: the loader `fit_matfac.jl` runs before it builds a
PathMatFacModel (analyses/scripts/julia/fit_matfac.jl:60-103, 193-246) and the helpers it takes from `script_util.jl`.

Datasets (HDF5 paths in the reference; the same strings are the keys of the `.npz` container used here -- h5py is not
installable in this image, exactly as for the parameter file of model_io.py):

    omic_data/data             M x N  float   NaN = missing            (script_util.jl:165-180 save_omic_data)
    omic_data/feature_assays   N      str     assay of every column ("mrnaseq", "methylation", "mutation", "cna", "rppa")
    omic_data/feature_genes    N      str
    omic_data/instances        M      str     sample ids
    omic_data/instance_groups  M      str     sample conditions (cancer types)
    barcodes/data              M x A  str     per (sample, assay) TCGA barcode, "" = none
    barcodes/features          A      str     the assay of every barcode column
    target                     M [x T]        optional

LAYOUT.  HDF5.jl writes a Julia array with its dimensions reversed, so a reader that goes through h5py sees
`omic_data/data` as N x M and `barcodes/data` as A x M ("(assays) x (samples)",
analyses/scripts/python/corrupt_mask_data.py:24) and transposes (prep_tcga_splits.py:15,21).  The container carries an
explicit entry `layout`: "h5py" (default of the writers here: every matrix stored as h5py would expose it, so code ported
from the reference's Python tools reads it unchanged) or "julia" (matrices in Julia's logical shape).  The loaders accept
both and always RETURN Julia's logical shapes (M x N), which is what `make_model` takes.
"""
import numpy as np

from .model import make_model

# script_util.jl:7-11, 25
DISTRIBUTION_MAP = {"mrnaseq": "normal", "methylation": "normal", "mutation": "bernoulli_sq_hinge", "cna": "ordinal_sq_hinge3",
                    "rppa": "normal"}
BATCHED_ASSAYS = {"mrnaseq", "methylation"}
OMIC_KEYS = ("omic_data/data", "omic_data/feature_assays", "omic_data/feature_genes", "omic_data/instances",
             "omic_data/instance_groups")
BARCODE_KEYS = ("barcodes/data", "barcodes/features")


def _strs(v):
    return np.asarray([str(x) for x in np.asarray(v).ravel()]).reshape(np.asarray(v).shape)


def _open(path_or_dict):
    if isinstance(path_or_dict, dict):
        return path_or_dict
    return np.load(path_or_dict, allow_pickle=False)


def _layout(z):
    lay = str(z["layout"]) if "layout" in z else "julia"
    if lay not in ("h5py", "julia"):
        raise ValueError(f"unknown layout {lay!r} (expected 'h5py' or 'julia')")
    return lay


def _matrix(z, key):
    """A 2-D dataset in Julia's logical shape whatever the container's layout."""
    a = np.asarray(z[key])
    return a.T if (_layout(z) == "h5py" and a.ndim == 2) else a


def save_omic_npz(path, feature_assays, feature_genes, instance_names, instance_groups, omic_matrix, barcodes=None,
                  barcode_features=None, target=None, layout="h5py"):
    """save_omic_data (script_util.jl:165-180) plus the barcode group the TCGA preprocessing adds; same assertions."""
    omic_matrix = np.asarray(omic_matrix)
    assert omic_matrix.shape[1] == len(feature_assays)
    assert omic_matrix.shape[0] == len(instance_names)
    assert len(instance_names) == len(instance_groups)
    assert len(feature_genes) == len(feature_assays)
    t = (lambda a: np.ascontiguousarray(np.asarray(a).T)) if layout == "h5py" else (lambda a: np.asarray(a))
    d = {"layout": np.asarray(layout), "omic_data/data": t(omic_matrix), "omic_data/feature_assays": _strs(feature_assays),
         "omic_data/feature_genes": _strs(feature_genes), "omic_data/instances": _strs(instance_names),
         "omic_data/instance_groups": _strs(instance_groups)}
    if barcodes is not None:
        barcodes = _strs(barcodes)
        assert barcodes.shape == (omic_matrix.shape[0], len(barcode_features))
        d["barcodes/data"] = t(barcodes)
        d["barcodes/features"] = _strs(barcode_features)
    if target is not None:
        d["target"] = t(np.asarray(target))
    np.savez(path, **d)


def load_omic_data(omic_file, omic_types):
    """load_omic_data (fit_matfac.jl:60-82): keep the columns whose assay is in `omic_types`; returns
    (omic_data M x N', sample_ids, sample_conditions, feature_genes, feature_assays)."""
    z = _open(omic_file)
    missing = [k for k in OMIC_KEYS if k not in z]
    if missing:
        raise KeyError(f"data file lacks {missing}")
    feature_assays = _strs(z["omic_data/feature_assays"])
    omic_set = set(omic_types)
    kept = np.array([a in omic_set for a in feature_assays], dtype=bool)                  # :64-66
    feature_assays = feature_assays[kept]                                                  # :69
    feature_genes = _strs(z["omic_data/feature_genes"])[kept]                              # :70
    omic_data = np.asarray(_matrix(z, "omic_data/data"))
    if omic_data.shape[1] != kept.size:
        raise ValueError(f"omic_data/data is {omic_data.shape} (Julia shape) but there are {kept.size} feature assays: wrong `layout`?")
    omic_data = omic_data[:, kept]                                                         # :73-74
    sample_ids = _strs(z["omic_data/instances"])                                           # :77
    sample_conditions = _strs(z["omic_data/instance_groups"])                              # :78
    if omic_data.shape[0] != sample_ids.size:
        raise ValueError("omic_data/data and omic_data/instances disagree on the number of samples")
    return omic_data, sample_ids, sample_conditions, feature_genes, feature_assays


def nan_fractions(omic_data, feature_assays):
    """print_nan_fractions (fit_matfac.jl:40-57) as a value: fraction of missing entries per assay, in order of appearance."""
    out = {}
    fa = np.asarray(feature_assays)
    for a in dict.fromkeys(fa.tolist()):
        blk = omic_data[:, fa == a]
        out[a] = float(np.mean(~np.isfinite(blk))) if blk.size else 0.0
    return out


def barcode_to_batch(barcode):
    """script_util.jl:147-157: the last two '-' separated terms of a TCGA barcode; "" stays ""."""
    if barcode == "":
        return ""
    terms = barcode.split("-")
    return "-".join(terms[max(len(terms) - 2, 0):])


def load_batches(omic_file, omic_types):
    """load_batches (fit_matfac.jl:85-103): {assay: batch id of every sample} for the batched assays among `omic_types`;
    None when there is none."""
    z = _open(omic_file)
    if any(k not in z for k in BARCODE_KEYS):
        raise KeyError("data file has no barcodes/data, barcodes/features")
    barcode_data = _strs(_matrix(z, "barcodes/data"))            # M x A
    batch_columns = [str(c) for c in np.asarray(z["barcodes/features"]).ravel()]
    assay_to_col = {a: i for i, a in enumerate(batch_columns)}
    result = {}
    for a in omic_types:
        if a in BATCHED_ASSAYS:
            result[a] = [barcode_to_batch(b) for b in barcode_data[:, assay_to_col[a]]]
    return result or None


def column_variances(data):
    """script_util.jl:27-46, as coded: col_sq_sums - mean^2 (NOT the mean of squares: the reference's own quirk, kept), 0
    for all-missing columns."""
    data = np.asarray(data, dtype=np.float64)
    nan_idx = ~np.isfinite(data)
    M = data.shape[0]
    counts = M - nan_idx.sum(axis=0)
    filled = np.where(nan_idx, 0.0, data)
    with np.errstate(divide="ignore", invalid="ignore"):
        means = filled.sum(axis=0) / counts
        var = (filled * filled).sum(axis=0) - means * means
    var[counts == 0] = 0
    return var


def var_filter(data, feature_groups, frac):
    """script_util.jl:62-78: per assay keep the columns whose variance reaches the (1 - frac) quantile; sorted 0-based
    column indices."""
    q = 1.0 - frac
    col_var = column_variances(data)
    groups = {}
    for i, g in enumerate(feature_groups):
        groups.setdefault(g, []).append(i)
    keep = set()
    for g, idx in groups.items():
        idx = np.asarray(idx)
        gv = col_var[idx]
        thr = np.quantile(gv, q)                  # (Julia's `quantile` default and numpy's "linear" are the same definition)
        keep.update(idx[gv >= thr].tolist())
    return sorted(keep)


def model_from_data_file(omic_file, omic_types, K=10, use_batch=True, use_conditions=True, var_filter_frac=1.0,
                         distribution_map=None, **model_kwargs):
    """The data side of fit_matfac.jl's main (:193-246, 285): load, variance filter, sort the samples by condition, attach
    batches, then PathMatFacModel(D; ...) = make_model.  `distribution_map` overrides script_util.jl's DISTRIBUTION_MAP (the
    library implements "normal", "bernoulli", "poisson": the hinge / ordinal losses of the production map are out of
    scope, DESIGN.md section 8, so a caller that holds mutation / cna columns must say which supported loss to use)."""
    D, sample_ids, sample_conditions, genes, assays = load_omic_data(omic_file, omic_types)
    keep = var_filter(D, assays, var_filter_frac)                                           # :208-209
    D, genes, assays = D[:, keep], genes[keep], assays[keep]
    srt = np.argsort(sample_conditions, kind="stable")                                      # :212 sortperm
    sample_conditions, sample_ids, D = sample_conditions[srt], sample_ids[srt], D[srt, :]
    kw = dict(model_kwargs)
    if use_conditions:
        kw["sample_conditions"] = list(sample_conditions)                                   # :229-231
    if use_batch:
        bd = load_batches(omic_file, omic_types)                                            # :233-236
        if bd is not None:
            kw["batch_dict"] = {a: [v[i] for i in srt] for a, v in bd.items()}
            kw.setdefault("sample_conditions", list(sample_conditions))
    dmap = dict(DISTRIBUTION_MAP)
    dmap.update(distribution_map or {})
    kw["feature_views"] = list(assays)                                                      # :243
    kw["feature_distributions"] = [dmap[a] for a in assays]                                 # :244
    kw["feature_ids"] = [f"{g}_{a}" for g, a in zip(genes, assays)]                         # :246
    return make_model(D, K=K, sample_ids=list(sample_ids), **kw)
