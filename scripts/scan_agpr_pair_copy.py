"""Scans gfx950 device code for the hipcc ROCm 7.2 VGPR->AGPR pair-copy miscompile found in round 2:
two consecutive `v_accvgpr_write_b32` into an even/odd AGPR pair from the SAME VGPR (the second write re-reads the first
source: both halves of a packed result end up holding the even element -- wrong gX, no diagnostic).

usage:  python scripts/scan_agpr_pair_copy.py file.s [...]            device assembly from -save-temps
        python scripts/scan_agpr_pair_copy.py --lib libpmf_hip.so      the code objects INSIDE a built library / object file:
                                                                       .hip_fatbin -> offload bundles -> llvm-objdump -d
The --lib form looks at the machine code that actually ships (tests/test_build_scan.py runs it on every CPU test run)."""
import re
import struct
import subprocess
import sys
import tempfile
from pathlib import Path

LLVM = Path("/opt/rocm/lib/llvm/bin")
PAT = re.compile(r"v_accvgpr_write_b32 a(\d+), (v\d+)")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def scan_lines(lines, label):
    """(hits, examples) for one stream of assembly / disassembly lines."""
    prev, kern, hits, ex = None, "?", 0, []
    for line in lines:
        s = line.strip()
        if s.endswith(">:") and "<" in s:                       # objdump label: 0000000000001000 <_Z...>:
            kern = s[s.index("<") + 1:-2]
        elif line.startswith("_Z") and ":" in line:             # compiler assembly label
            kern = line.split(":")[0]
        m = PAT.search(line)
        if m:
            cur = (int(m.group(1)), m.group(2))
            if prev and cur[0] == prev[0] + 1 and prev[0] % 2 == 0 and cur[1] == prev[1]:
                hits += 1
                if len(ex) < 3:
                    ex.append(f"{label}: {kern[:90]}: a{prev[0]}, a{cur[0]} <- {cur[1]}")
            prev = cur
        elif "v_accvgpr" not in line:
            prev = None
    return hits, ex


def code_objects(path):
    """The gfx950 code objects embedded in an ELF's .hip_fatbin section (one offload bundle per translation unit)."""
    with tempfile.TemporaryDirectory() as td:
        fb = Path(td) / "fatbin"
        subprocess.run([str(LLVM / "llvm-objcopy"), "--dump-section", f".hip_fatbin={fb}", str(path), str(Path(td) / "unused.o")],
                       check=True, capture_output=True)
        blob = fb.read_bytes()
    out, pos = [], 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            break
        n, = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, idlen = struct.unpack_from("<QQQ", blob, q)
            ident = blob[q + 24:q + 24 + idlen].decode()
            q += 24 + idlen
            if "gfx950" in ident and size > 0:
                out.append(blob[pos + off:pos + off + size])
        pos = q
    return out


def scan_library(path):
    total, examples, n_mfma_kernels = 0, [], 0
    cos = code_objects(path)
    with tempfile.TemporaryDirectory() as td:
        for i, co in enumerate(cos):
            f = Path(td) / f"co{i}.o"
            f.write_bytes(co)
            dis = subprocess.run([str(LLVM / "llvm-objdump"), "-d", "--mcpu=gfx950", str(f)], check=True, capture_output=True, text=True).stdout
            n_mfma_kernels += int("v_mfma" in dis)
            h, ex = scan_lines(dis.splitlines(), f"{Path(path).name}#co{i}")
            total += h
            examples += ex
    return total, examples, len(cos), n_mfma_kernels


def main(argv):
    if len(argv) >= 2 and argv[0] == "--lib":
        total, ex, n, nm = scan_library(argv[1])
        for e in ex:
            print(e)
        print(f"{argv[1]}: {n} gfx950 code objects ({nm} with MFMAs), {total} suspicious pair copies")
        return 1 if total else 0
    total = 0
    for fn in argv:
        h, ex = scan_lines(open(fn, errors="replace"), fn)
        for e in ex:
            print(e)
        print(f"{fn}: {h} suspicious pair copies")
        total += h
    print("total", total)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
