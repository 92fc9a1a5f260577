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
from .fit import mf_fit_, mf_fit_adapt_lr_, init_theta_, init_factors_, construct_optimizer  # noqa: F401
from .transform import transform  # noqa: F401
