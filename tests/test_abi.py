"""The C-ABI shared library loads without a GPU and exports every symbol include/pmf_hip.h declares."""
import ctypes
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    txt = (ROOT / "include" / "pmf_hip.h").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pmf_[a-z_A-Z0-9]+)\s*\(", txt)))


def test_header_symbols_are_exported(pkg):
    syms = declared_symbols()
    assert len(syms) >= 40
    lib = ctypes.CDLL(str(ROOT / "pathmatfac.jl_amd" / "libpmf_hip.so"))
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/pmf_hip.h but not exported"
    assert sorted(pkg._lib.EXPORTS) == syms          # the ctypes binding covers exactly the header
    assert lib.pmf_version() == 1


def test_structs_match_header_layout(pkg):
    # pmf_fit_opts: 12 int32 + 2 double + 1 int64 ; pmf_fit_result: 4 int32 + double + pointer + double
    assert ctypes.sizeof(pkg._lib.FitOpts) == 12 * 4 + 2 * 8 + 8
    assert ctypes.sizeof(pkg._lib.FitResult) == 4 * 4 + 8 + 8 + 8


def test_missing_library_fails_loudly(pkg, tmp_path):
    import pytest
    with pytest.raises(pkg.PMFError):
        pkg._lib.load_library(tmp_path / "nope.so")
