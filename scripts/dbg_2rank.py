import socket, subprocess, sys, os, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
out = tempfile.mkdtemp()
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
code = ("import sys; sys.path[:0] = [%r, %r]; import test_gpu_comm as t; "
        "t._worker(int(sys.argv[1]), 2, int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5])") % (str(ROOT), str(ROOT / "tests"))
env = dict(os.environ, PMF_DEBUG_LOSS="1")
procs = [subprocess.Popen([sys.executable, "-c", code, str(r), str(port), out, prec, "factors"], env=env) for r in range(2)]
print([p.wait() for p in procs])
