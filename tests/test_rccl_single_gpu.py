"""The RCCL leg of bench.py on the one GPU of a test box: backend 'nccl' with world size 1, in a child process
(a process group must not leak into the test process)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_rccl_leg_initialises_and_reduces_on_one_gpu():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rccl-selftest", "--channels", "256"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)["rccl_selftest"]
    assert r["backend"] == "nccl" and r["world_size"] == 1 and r["grads_unchanged"] and r["flat_grad_elements"] == 256 * 256 + 256
    assert r["allreduce"]["backend"] == "rccl" and r["allreduce"]["allreduce_us"] > 0
