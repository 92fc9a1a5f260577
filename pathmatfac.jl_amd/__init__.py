"""pathmatfac.jl_amd -- MI355X-native drop-in for PathMatFac.jl's `fit!` gradient-descent path.

Host-side mirror of the reference's operator interface (PathMatFacModel / fit! / transform and the structs
they are made of) on top of the C-ABI HIP library libpmf_hip.so.  Import through `pmf_import.load()`.
Naming: the reference's `name!` is `name_` here.
"""
from . import _lib  # noqa: F401
from ._lib import Context, PMFError, load_library  # noqa: F401
from . import util, batch_array, layers, regularizers, optimizers, matfac, model, fit, transform as _transform, parallel  # noqa: F401,E501
from .batch_array import BatchArray  # noqa: F401
from .model import PathMatFacModel, make_model  # noqa: F401
from .fit import (mf_fit_, mf_fit_adapt_lr_, init_theta_, init_factors_, construct_optimizer, init_mu_, init_logsigma_,  # noqa: F401
                  reweight_col_losses_, construct_minimal_regularizer, init_batch_effects_, theta_delta_em, whiten_,
                  rotate_by_svd_, reorder_by_importance_, reweight_eb_, basic_fit_, fit_ard_, fit_non_ard_,
                  fit_feature_set_ard_, fit_)
from . import featureset_ard  # noqa: F401
from .featureset_ard import update_A_, update_lambda_  # noqa: F401
from .transform import transform  # noqa: F401
from . import model_io  # noqa: F401
from .model_io import save_params_npz, load_params_npz  # noqa: F401
from . import data_io  # noqa: F401
from .data_io import load_omic_data, load_batches, save_omic_npz, model_from_data_file  # noqa: F401
