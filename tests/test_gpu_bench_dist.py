"""bench.py's N > 1 path end to end on the one-GPU test box: two ranks (gloo instead of RCCL, both on device 0) run the
sharded tape with the per-step all-reduce of the partial sums and the sharded device-resident sampler; the line rank 0
prints must be the last line on stdout, carry the whole-job rate and pass its own all-reduce self-check."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_bench_line():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo", BENCH_FORCE_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--loci", "1500", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = r.stdout.strip().splitlines()[-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["scaling"] == "weak" and d["value"] > 0
    assert d["allreduce_check"] == "ok"
    assert d["roofline"]["frac"] > 0 and d["cpu_baseline"] is None
    smp = d["device_resident_sampler"]
    assert smp["n_gpus"] == 2 and smp["iterations_per_s"] > 0 and 0.2 < smp["acceptance"] < 0.9
