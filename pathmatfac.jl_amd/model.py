"""PathMatFacModel (src/model.jl): validating constructor, column permutation, model assembly."""
import numpy as np

from . import matfac as MF
from ._lib import Context
from .layers import construct_model_layers
from .regularizers import construct_layer_reg, construct_X_reg, construct_Y_reg
from .util import is_contiguous


class PathMatFacModel:  # model.jl:6-28
    def __init__(self, matfac, data, sample_ids, sample_conditions, feature_ids, feature_views, data_idx):
        self.matfac = matfac
        self.data = data
        self.sample_ids = sample_ids
        self.sample_conditions = sample_conditions
        self.feature_ids = feature_ids
        self.feature_views = feature_views
        self.data_idx = data_idx          # 1-based permutation: model column j is raw column data_idx[j]
        self._ctx = None
        self._ctx_data_id = None

    # ---- gpu(model) / cpu(model)  (analyses/scripts/julia/fit_matfac.jl:325-340)
    def device_context(self, device=0):
        """The model's pmf_ctx; the data matrix is uploaded once and stays resident in HBM."""
        if self._ctx is None:
            self._ctx = Context(device)
        key = (id(self.data), None if self.data is None else self.data.shape)
        if self._ctx_data_id != key:
            if self.data is None:
                raise ValueError("model.data is nothing")
            self._ctx.set_data(self.data)
            self._ctx_data_id = key
        return self._ctx

    def invalidate_device_data(self):
        """Call after editing model.data IN PLACE: the HBM copy is keyed on the array's identity and shape, so an in-place
        edit is not seen; the next device_context() uploads the matrix again."""
        self._ctx_data_id = None

    def release_device(self):
        if self._ctx is not None:
            self._ctx.close()
        self._ctx = None
        self._ctx_data_id = None


def assemble_model(D, K, sample_ids, sample_conditions, feature_ids, feature_views, feature_distributions, batch_dict,
                   feature_sets_dict, featureset_names, feature_graphs, sample_graphs, lambda_X_l2,
                   lambda_X_condition, lambda_X_graph, lambda_Y_l2, lambda_Y_selective_l1, lambda_Y_graph, Y_ard,
                   Y_feature_set_ard, alpha0, v0, lambda_layer, rng=None):
    """model.jl:37-81.  Columns are permuted so that (distribution, view) pairs form contiguous blocks
    (stable sortperm of the zipped pairs, :50-54); `data_idx` keeps the permutation (1-based)."""
    M, N = D.shape
    keys = list(zip(feature_distributions, feature_views))
    perm = sorted(range(N), key=lambda j: keys[j])          # Python's sort is stable, like Julia's sortperm
    data_idx = np.array(perm, dtype=np.int64) + 1
    feature_ids = [feature_ids[j] for j in perm]
    feature_views = [feature_views[j] for j in perm]
    feature_distributions = [feature_distributions[j] for j in perm]
    D = np.asfortranarray(np.asarray(D, dtype=np.float32)[:, perm])
    col_layers = construct_model_layers(feature_views, batch_dict, rng=rng)
    layer_reg = construct_layer_reg(feature_views, batch_dict, col_layers, lambda_layer)
    X_reg = construct_X_reg(K, M, sample_ids, sample_conditions, sample_graphs, lambda_X_l2, lambda_X_condition,
                            lambda_X_graph, Y_ard, Y_feature_set_ard)
    Y_reg = construct_Y_reg(K, N, feature_ids, feature_views, feature_sets_dict, feature_graphs, lambda_Y_l2,
                            lambda_Y_selective_l1, lambda_Y_graph, Y_ard, Y_feature_set_ard, featureset_names,
                            alpha0, v0)
    matfac = MF.MatFacModel(M, N, K, feature_distributions, col_transform=col_layers, X_reg=X_reg, Y_reg=Y_reg,
                            col_transform_reg=layer_reg, rng=rng)
    return PathMatFacModel(matfac, D, sample_ids, sample_conditions, feature_ids, feature_views, data_idx)


def make_model(D, K=10, sample_ids=None, sample_conditions=None, feature_ids=None, feature_views=None,
               feature_distributions=None, batch_dict=None, sample_graphs=None, feature_sets_dict=None,
               featureset_names=None, feature_graphs=None, lambda_X_l2=None, lambda_X_condition=1.0,
               lambda_X_graph=1.0, lambda_Y_l2=1.0, lambda_Y_selective_l1=None, lambda_Y_graph=None,
               lambda_layer=1.0, Y_ard=False, Y_fsard=False, fsard_alpha0=np.float32(1.001),
               fsard_v0=np.float32(0.8), rng=None):
    """PathMatFacModel(D; K=10, ...) -- the validating constructor of model.jl:92-196 (same keyword names)."""
    D = np.asarray(D)
    M, N = D.shape
    if feature_graphs is not None:
        K = len(feature_graphs)
        if sample_graphs is not None:
            assert K == len(sample_graphs), "`sample_graphs` and `feature_graphs` must have equal length; or one of them must be nothing"
    elif sample_graphs is not None:
        K = len(sample_graphs)
    if sample_ids is not None:
        assert len(sample_ids) == len(set(sample_ids)), "`sample_ids` must be unique"
        assert len(sample_ids) == M, "`sample_ids` must be nothing or have length equal to size(D,1)"
    else:
        sample_ids = list(range(1, M + 1))
    if sample_conditions is not None:
        assert len(sample_conditions) == M, "`sample_conditions` must be nothing or have length equal to size(D,1)"
        assert is_contiguous(sample_conditions), "`sample_conditions` must be contiguous; I.e., samples must be grouped by condition."
    if feature_ids is not None:
        assert len(feature_ids) == len(set(feature_ids)), "`feature_ids` must be left default, or set to a vector of unique identifiers"
        assert len(feature_ids) == N, "`feature_ids` must have length equal to dim(D,2)"
    else:
        feature_ids = list(range(1, N + 1))
    if batch_dict is not None:
        assert feature_views is not None, "`feature_views` must be provided whenever `batch_dict` is provided"
        assert sample_conditions is not None, "`sample_conditions` must be provided whenever `batch_dict` is provided"
        assert set(batch_dict.keys()) <= set(feature_views), "The `batch_dict` keys must be a subset of `feature_views`"
        for v in batch_dict.values():
            assert len(v) == M, "Each value of `batch_dict` must be a vector of length size(D,1)"
    if feature_views is not None:
        assert len(feature_views) == N, "`feature_views` must be nothing or have length equal to size(D,2)"
    else:
        feature_views = [1] * N
    if feature_distributions is not None:
        assert len(feature_distributions) == N, "`feature_distributions` must (a) be nothing or have length equal to size(D,2)"
        assert all(d in MF.VALID_LOSSES for d in feature_distributions), f"Each entry of `feature_distributions` must be one of {set(MF.VALID_LOSSES)}"
    else:
        feature_distributions = ["normal"] * N
    if Y_fsard:
        assert feature_sets_dict is not None, "`feature_sets_dict` must be provided whenever `Y_fsard` is true."
    return assemble_model(D, K, list(sample_ids), None if sample_conditions is None else list(sample_conditions),
                          list(feature_ids), list(feature_views), list(feature_distributions), batch_dict,
                          feature_sets_dict, featureset_names, feature_graphs, sample_graphs, lambda_X_l2,
                          lambda_X_condition, lambda_X_graph, lambda_Y_l2, lambda_Y_selective_l1, lambda_Y_graph,
                          Y_ard, Y_fsard, fsard_alpha0, fsard_v0, lambda_layer, rng=rng)
