"""pathmatfac.jl_amd -- MI355X-native drop-in for PathMatFac.jl's `fit!` gradient-descent path.

Host-side mirror of the reference's operator interface (PathMatFacModel / fit! / transform and the structs
they are made of) on top of the C-ABI HIP library libpmf_hip.so.  Import through `pmf_import.load()`.
"""
from . import _lib  # noqa: F401
from ._lib import Context, PMFError, load_library  # noqa: F401
