"""Guards of the build that run on the CPU with every test run ("the guard exists only if someone remembers
to"): the machine code inside the built libpmf_hip.so is disassembled and scanned for the hipcc VGPR->AGPR pair-copy
miscompile that produced a wrong grad(X) in round 1 (scripts/scan_agpr_pair_copy.py)."""
import importlib.util
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _scanner():
    spec = importlib.util.spec_from_file_location("scan_agpr", ROOT / "scripts" / "scan_agpr_pair_copy.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_scanner_flags_the_round1_pattern_and_nothing_else():
    s = _scanner()
    bad = """
0000000000001000 <_Z17pmf_fused_kernelILi4ELi4ELi1ELi0ELb0ELi0ELb0EEv9FusedArgs>:
	v_pk_mul_f32 v[20:21], v[20:21], v[30:31]
	v_accvgpr_write_b32 a64, v20          // the ISA of commit 0e0be6b's launch_bounds(256, 1) build
	v_accvgpr_write_b32 a65, v20
	v_accvgpr_write_b32 a66, v56
	v_accvgpr_write_b32 a67, v56
"""
    hits, ex = s.scan_lines(bad.splitlines(), "synthetic")
    assert hits == 2 and "pmf_fused_kernel" in ex[0] and "a64, a65 <- v20" in ex[0]
    good = """
	v_accvgpr_write_b32 a64, v20
	v_accvgpr_write_b32 a65, v21
	v_accvgpr_write_b32 a67, v21          // odd -> even pair boundary, same source: a broadcast, not a pair copy
	v_accvgpr_write_b32 a68, v21
	v_add_f32 v1, v2, v3
	v_accvgpr_write_b32 a69, v21
"""
    assert s.scan_lines(good.splitlines(), "synthetic")[0] == 0


def test_built_library_has_no_agpr_pair_copy():
    s = _scanner()
    lib = ROOT / "pathmatfac.jl_amd" / "libpmf_hip.so"
    assert lib.exists(), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    total, examples, n_co, n_mfma = s.scan_library(lib)
    assert n_co >= 20 and n_mfma >= 18, (n_co, n_mfma)        # every kernel translation unit was found and disassembled
    assert total == 0, examples
