"""bench.py's output contract on a small workload (the driver runs it with the defaults): exactly one line on stdout, a JSON
object with the contract's keys, `roofline` and `cpu_baseline`; the full-model flavour; the one-rank RCCL path with a
chunked data pass (PMF_FORCE_DIST=1), whose losses must equal the plain run's (same arithmetic, same order)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
        "data", "config", "roofline"}


def run_bench(*args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--M", "3000", "--N", "1500", "--K", "64", "--steps", "4", "--warmup", "1",
                        *args], capture_output=True, text=True, env=e, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_line_contract_small():
    d = run_bench(env={"PMF_BENCH_SKIP_OTHER_CPU": "1"})
    assert KEYS <= set(d), KEYS - set(d)
    assert d["metric"] == "fit_iters_per_sec" and d["unit"] == "iters/s" and d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] in ("mfma", "hbm") and 0.0 < rf["frac"] < 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert rf["launches"] == 4 and rf["kernel_ms"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert d["other_precision"]["kernel_taken"] is True
    assert d["loss_last"] < d["loss_first"]


def test_bench_full_model_and_one_rank_rccl():
    a = run_bench("--full-model", "--no-cpu-baseline")
    assert a["config"]["full_model"] is True and "Bernoulli" in a["config"]["workload"]
    b = run_bench("--full-model", "--no-cpu-baseline", env={"PMF_FORCE_DIST": "1", "PMF_BENCH_CHUNKS": "3"})
    assert b["config"]["rccl_ranks"] == 1 and b["config"]["column_chunks"] == 3 and b["config"]["collectives_issued"] > 0
    # chunked + communicator: same arithmetic; the gY reduction order inside a chunk is the plain pass's
    assert abs(a["loss_last"] - b["loss_last"]) <= 1e-6 * abs(a["loss_last"]), (a["loss_last"], b["loss_last"])
