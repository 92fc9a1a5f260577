"""Condenses gpurun_out/prof_r3/<name>/ (scripts/profile_r3.sh) into profiles/r3_<name>_kernel_stats.csv and
profiles/r3_<name>_summary.md: top kernels with their average duration, and the per-launch PMC means of every kernel that
took more than 2 % of the time (FETCH_SIZE x 2 x 1024 = bytes read, the gfx950 correction of MI355X_MICROARCH.md)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys
from pathlib import Path

src_root = Path("gpurun_out/prof_r3")
dst = Path("profiles")
dst.mkdir(exist_ok=True)
WORK = {"headline": "python bench.py (200000 x 50000 f32, K = 64; exact kernel, then the split-bf16 kernel on the same data)",
        "config4_shard": "python bench.py --M 125000 --N 100000 --K 128 --precision bf16x3 --store bf16 (one rank's shard of configs[4])",
        "config1": "python bench.py --M 20000 --N 10000 --K 32 (BASELINE configs[1])",
        "general": "scripts/kbench_mixed.py 100000 50000 64 all (20 % Bernoulli columns, 2 batch views x 8 batches, 5 % missing)",
        "layers": "scripts/kbench_layers.py 100000 50000 64 (layer-only epochs, then X + Y + layers epochs)",
        "xonly_yonly": "scripts/kbench_xonly.py 100000 50000 64 (grad(X)-only, grad(Y)-only, both)",
        "config4_shard_full": "python bench.py --M 125000 --N 100000 --K 128 --precision bf16x3 --store bf16 --full-model (configs[4] shard: "
                              "20 % Bernoulli columns, column + batch layers, 10 % missing)",
        "config2_full": "python bench.py --M 20000 --N 10000 --K 32 --full-model (configs[2]: exact kernel, then the split-bf16 kernel)"}
out_json = {}
for name in sorted(p.name for p in src_root.iterdir() if p.is_dir()):
    src = src_root / name
    st = glob.glob(str(src / "stats" / "*" / "*_kernel_stats.csv"))
    if not st:
        print("no stats for", name)
        continue
    stats = max(st, key=os.path.getmtime)
    shutil.copy(stats, dst / f"r3_{name}_kernel_stats.csv")
    rows = list(csv.DictReader(open(stats)))
    lines = [f"# rocprofv3 summary r3 / {name}\n", f"workload: `{WORK.get(name, name)}`, 1 x MI355X\n",
             "## kernel-trace --stats (top kernels)\n", "| kernel | calls | avg ms | % |", "|---|---|---|---|"]
    big = []
    for r in rows[:8]:
        lines.append(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['AverageNs'])/1e6:.3f} | {float(r['Percentage']):.2f} |")
        if float(r["Percentage"]) > 2.0 and ("pmf_" in r["Name"] or "k_layer" in r["Name"]):   # (not the synthetic-data generator / setup kernels)
            big.append(r["Name"])
    out_json[name] = {"kernels": {r["Name"][:120]: {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6} for r in rows[:8]}}
    for kern in big:
        vals = {}
        for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
            fs = glob.glob(str(src / sub / "*" / "*_counter_collection.csv"))
            if not fs:
                continue
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
                if r["Kernel_Name"] == kern:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, v in agg.items():
                vals[k] = sum(v) / len(v)
        if not vals:
            continue
        lines.append(f"\n## PMC (separate passes), mean per launch of `{kern[:100]}`\n")
        lines += ["| counter | value |", "|---|---|"]
        for k in sorted(vals):
            lines.append(f"| {k} | {vals[k]:.5g} |")
        extra = {}
        if "FETCH_SIZE" in vals:
            extra["hbm_read_bytes_per_launch"] = vals["FETCH_SIZE"] * 2 * 1024
            lines.append(f"\nHBM read per launch = FETCH_SIZE x 2 x 1024 = {extra['hbm_read_bytes_per_launch']/1e9:.2f} GB")
        if "WRITE_SIZE" in vals:
            extra["hbm_write_bytes_per_launch"] = vals["WRITE_SIZE"] * 1024
            lines.append(f"HBM write per launch = WRITE_SIZE x 1024 = {extra['hbm_write_bytes_per_launch']/1e9:.2f} GB")
        if "SQ_WAVE_CYCLES" in vals:
            w = vals["SQ_WAVE_CYCLES"]
            lines.append(f"wave time: waitcnt/barrier {vals['SQ_WAIT_ANY']/w*100:.1f} %, issue-stall {vals['SQ_WAIT_INST_ANY']/w*100:.1f} %, "
                         f"issuing {vals['SQ_ACTIVE_INST_ANY']/w*100:.1f} %; SQ_LDS_BANK_CONFLICT {vals.get('SQ_LDS_BANK_CONFLICT', 0):.3g}")
        out_json[name].setdefault("pmc", {})[kern[:120]] = {**{k: vals[k] for k in vals}, **extra}
    (dst / f"r3_{name}_summary.md").write_text("\n".join(lines) + "\n")
    print("\n".join(lines))
(dst / "r3_profiles.json").write_text(json.dumps(out_json, indent=1))
