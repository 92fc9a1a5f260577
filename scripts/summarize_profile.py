"""Condenses a scripts/profile.sh output directory into profiles/<tag>_summary.md + the kernel stats csv."""
import collections
import csv
import glob
import shutil
import sys
from pathlib import Path

tag = sys.argv[1]
src = Path("gpurun_out") / f"prof_{tag}"
dst = Path("profiles")
dst.mkdir(exist_ok=True)
import os
stats = max(glob.glob(str(src / "stats" / "*" / "*_kernel_stats.csv")), key=os.path.getmtime)   # newest run
shutil.copy(stats, dst / f"{tag}_kernel_stats.csv")
lines = [f"# rocprofv3 summary `{tag}` (python bench.py, 200000x50000 f32, K=64, 1 x MI355X)\n",
         "## kernel-trace --stats (top kernels)\n", "| kernel | calls | avg ms | % |", "|---|---|---|---|"]
for r in list(csv.DictReader(open(stats)))[:6]:
    lines.append(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['AverageNs'])/1e6:.3f} | {float(r['Percentage']):.2f} |")
notes = {"FETCH_SIZE": "KB; gfx950 reports 1/2 of wide streaming reads -> x2 x1024 bytes",
         "WRITE_SIZE": "KB; exact for 16-B stores and float atomics", "TCC_EA0_ATOMIC_sum": "x64 B = atomic bytes",
         "SQ_WAVE_CYCLES": "quad-cycles; = WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY",
         "SQ_VALU_MFMA_BUSY_CYCLES": "cycles; 64 per v_mfma_f32_32x32x2_f32, 32 per v_mfma_f32_32x32x16_bf16"}
traffic = {}
# bench.py times both arithmetic modes in one process: the exact kernel and the opt-in split-bf16 kernel
for kern, title in (("pmf_fused_kernel", "exact f32 (headline)"), ("pmf_fused_sb_kernel", "split-bf16 (pmf_set_precision)")):
    vals = {}
    for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
        fs = glob.glob(str(src / sub / "*" / "*_counter_collection.csv"))
        if not fs:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
            if kern + "<" in r["Kernel_Name"] or r["Kernel_Name"].startswith(kern):
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            vals[k] = sum(v) / len(v)
    if not vals:
        continue
    lines.append(f"\n## PMC passes (separate runs), mean per launch of {kern} -- {title}\n")
    lines += ["| counter | value | note |", "|---|---|---|"]
    for k in sorted(vals):
        lines.append(f"| {k} | {vals[k]:.4g} | {notes.get(k, '')} |")
    if "FETCH_SIZE" in vals:
        lines.append(f"\nHBM read per launch = FETCH_SIZE x 2 x 1024 = {vals['FETCH_SIZE']*2*1024/1e9:.2f} GB "
                     f"(algorithmic: 4*M*N = 40.00 GB).")
    if "WRITE_SIZE" in vals:
        lines.append(f"HBM write per launch = WRITE_SIZE x 1024 = {vals['WRITE_SIZE']*1024/1e9:.2f} GB "
                     f"(gY atomics: {vals.get('TCC_EA0_ATOMIC_sum', 0)*64/1e9:.2f} GB).")
    if "SQ_WAVE_CYCLES" in vals:
        w = vals["SQ_WAVE_CYCLES"]
        lines.append(f"Wave time split: waitcnt/barrier {vals['SQ_WAIT_ANY']/w*100:.1f} %, issue-stall "
                     f"{vals['SQ_WAIT_INST_ANY']/w*100:.1f} %, issuing {vals['SQ_ACTIVE_INST_ANY']/w*100:.1f} %.")
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        traffic[kern] = (vals["FETCH_SIZE"] * 2 * 1024, vals["WRITE_SIZE"] * 1024)
(dst / f"{tag}_summary.md").write_text("\n".join(lines) + "\n")
print("\n".join(lines))
print("traffic (read, write bytes per launch):", traffic)
