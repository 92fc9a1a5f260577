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


def test_plain_c_host_compiles_and_links(tmp_path):
    """examples/fit_c.c uses the boundary from C alone (no Python, no torch): it must compile against the header and
    link against the library (running it needs the GPU: tests/test_gpu_host.py)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        import pytest
        pytest.skip("gcc not available")
    exe = tmp_path / "fit_c"
    cmd = ["gcc", "-O2", "-Wall", "-Werror", f"-I{ROOT / 'include'}", str(ROOT / "examples" / "fit_c.c"),
           f"-L{ROOT / 'pathmatfac.jl_amd'}", "-lpmf_hip", f"-Wl,-rpath,{ROOT / 'pathmatfac.jl_amd'}", "-lm", "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert exe.exists()
