"""transform (src/transform.jl): embed new samples with Y and the column layers held constant."""
import copy

import numpy as np

from .fit import mf_fit_adapt_lr_
from .layers import Identity, freeze_layer_, unfreeze_layer_
from .model import PathMatFacModel
from .regularizers import SequenceReg, ZeroReg, freeze_reg_
from .util import keymatch


def transform(model, D, use_gpu=True, feature_ids=None, sample_ids=None, verbosity=1, print_prefix="",
              max_epochs=1000, lr=1.0, capacity=10 ** 8, **fit_kwargs):
    """transform(model, D; ...) (transform.jl:6-106).  Returns a new PathMatFacModel whose matfac.X (K x M_new)
    holds the embedding.  `use_gpu` is accepted for API compatibility: the fit always runs on the GPU."""
    K, N = model.matfac.Y.shape
    D = np.asarray(D, dtype=np.float32)
    M_new, N_new = D.shape
    old_idx = list(range(1, N + 1))
    new_idx = list(range(1, N_new + 1))
    if feature_ids is None:
        assert N_new == N, "Columns of D do not match columns of training data. Provide `feature_ids` to ensure they match."
    else:
        old_idx, new_idx = keymatch(model.feature_ids, list(feature_ids))     # transform.jl:35
    if sample_ids is not None:
        assert len(sample_ids) == M_new, "`sample_ids` must have length == size(D,1)"
    else:
        sample_ids = list(range(1, M_new + 1))
    # new data padded to the training columns with NaN (transform.jl:55-57)
    new_data = np.full((M_new, N), np.nan, dtype=np.float32, order="F")
    new_data[:, np.array(old_idx, dtype=np.int64) - 1] = D[:, np.array(new_idx, dtype=np.int64) - 1]
    mf = copy.deepcopy(model.matfac)
    mf.Y_reg = ZeroReg()                                    # :61
    mf.col_transform.set_layer_(2, Identity())              # :64
    mf.col_transform.set_layer_(4, Identity())              # :65
    mf.X = np.zeros((K, M_new), dtype=np.float32, order="F")  # :68-69
    mf.X_reg = ZeroReg()                                    # :70
    new_model = PathMatFacModel(mf, new_data, sample_ids, None, list(model.feature_ids), list(model.feature_views),
                                np.array(model.data_idx))
    freeze_layer_(mf.col_transform, [1, 2, 3, 4])           # :84
    if isinstance(mf.col_transform_reg, SequenceReg):
        freeze_reg_(mf.col_transform_reg, [1, 2, 3, 4])     # :85
    mf_fit_adapt_lr_(new_model, update_X=True, verbosity=verbosity, print_prefix="    " + print_prefix,
                     max_epochs=max_epochs, lr=lr, capacity=capacity, **fit_kwargs)   # :88-90
    unfreeze_layer_(mf.col_transform, [1, 2, 3, 4])         # :96
    new_model.release_device()
    return new_model
