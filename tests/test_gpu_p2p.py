"""bpa_p2p_*: the one-shot all-reduce between processes (here: two ranks sharing the test box's GPU; the mailboxes travel
as hipIpc handles exactly as between the GPUs of a node) — bit-equal to the rank-order sum, also back to back."""
import json
import os
import re
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


# (eight ranks on ONE GPU also pass — tools note in DESIGN section 7 — but depend on how the driver time-slices eight
# processes whose kernels wait for each other; between GPUs of a node every rank has its own device)
@pytest.mark.parametrize("world", [2])
def test_p2p_allreduce_between_processes(world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(HERE, "p2p_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [json.loads(m) for m in re.findall(r"\{[^{}]*\}", r.stdout)]          # (the ranks' lines may run together)
    assert sorted(d["rank"] for d in lines) == list(range(world))
    assert all(d["worst"] == 0.0 and d["chained_ok"] for d in lines)
    assert [d["timed_out"] for d in sorted(lines, key=lambda d: d["rank"])] == [True] + [None]*(world - 1)       # bounded wait
