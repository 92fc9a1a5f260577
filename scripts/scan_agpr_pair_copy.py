"""Scans device assembly (.s from -save-temps) for the hipcc ROCm 7.2 / gfx950 VGPR->AGPR pair-copy miscompile found in
round 2: two consecutive `v_accvgpr_write_b32` into an even/odd AGPR pair from the SAME VGPR.
usage: python scripts/scan_agpr_pair_copy.py file.s [...]"""
import re
import sys
pat = re.compile(r"v_accvgpr_write_b32 a(\d+), (v\d+)")
total = 0
for fn in sys.argv[1:]:
    prev = None
    kern = "?"
    hits = 0
    for line in open(fn, errors="replace"):
        if line.startswith("_Z") and line.rstrip().endswith(":") or (line.startswith("_Z") and ":" in line):
            kern = line.split(":")[0]
        m = pat.search(line)
        if m:
            cur = (int(m.group(1)), m.group(2))
            if prev and cur[0] == prev[0] + 1 and prev[0] % 2 == 0 and cur[1] == prev[1]:
                hits += 1
                if hits <= 3:
                    print(f"{fn}: {kern[:80]}: a{prev[0]}, a{cur[0]} <- {cur[1]}")
            prev = cur
        elif "v_accvgpr" not in line:
            prev = None
    total += hits
    print(f"{fn}: {hits} suspicious pair copies")
print("total", total)
