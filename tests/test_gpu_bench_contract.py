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


def test_bench_under_the_drivers_launcher_one_rank_real_rccl():
    """The launch path of the N > 1 contract with N = 1: torch.distributed.run starts the rank, which reads RANK / LOCAL_RANK /
    WORLD_SIZE, maps LOCAL_RANK to its device, forms a (one-rank) RCCL communicator in the library and prints ONE JSON line."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    e = dict(os.environ, PMF_FORCE_DIST="1", PMF_BENCH_SKIP_OTHER_CPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "1", "--M", "3000", "--N", "1500", "--K", "64", "--steps", "4",
                        "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, env=e, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["config"]["rccl_ranks"] == 1 and d["config"]["collectives_issued"] > 0
    assert d["loss_last"] < d["loss_first"]


def test_bench_pinned_launcher_and_rank_failure():
    """A launcher that pins one device per rank leaves one visible device: LOCAL_RANK = 3 must run on device 0.  A rank that
    cannot get a device ends the job non-zero with its own message on stderr (no JSON line, no hang)."""
    args = ["--gpus", "1", "--M", "1000", "--N", "640", "--K", "32", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    e3 = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="3")
    r3 = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True, env=e3, timeout=600)
    assert r3.returncode == 0, r3.stderr[-2000:]
    assert json.loads([ln for ln in r3.stdout.splitlines() if ln.strip()][0])["n_gpus"] == 1
    e0 = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", HIP_VISIBLE_DEVICES="")
    r0 = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True, env=e0, timeout=600)
    assert r0.returncode != 0 and "[bench rank 0]" in r0.stderr, (r0.returncode, r0.stderr[-500:])
    assert not [ln for ln in r0.stdout.splitlines() if ln.strip()]
