#!/bin/bash
# Scans the machine code of the built library for the VGPR->AGPR pair-copy miscompile.  CPU only, ~10 s.
# (tests/test_build_scan.py runs the same scan with every CPU test run.)  An argument names another library / object file.
set -eu
HERE="$(cd "$(dirname "$0")" && pwd)"
python3 "$HERE/scan_agpr_pair_copy.py" --lib "${1:-$HERE/../pathmatfac.jl_amd/libpmf_hip.so}"
