"""Imports the product package, whose directory name (`pathmatfac.jl_amd`) is not a valid Python
identifier, under the module name `pathmatfac_jl_amd`."""
import importlib.util
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent
PKG_DIR = ROOT / "pathmatfac.jl_amd"
MOD_NAME = "pathmatfac_jl_amd"


def load():
    if MOD_NAME in sys.modules:
        return sys.modules[MOD_NAME]
    spec = importlib.util.spec_from_file_location(MOD_NAME, PKG_DIR / "__init__.py",
                                                  submodule_search_locations=[str(PKG_DIR)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[MOD_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
