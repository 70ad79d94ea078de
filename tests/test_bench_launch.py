"""bench.py --gpus N without a launcher: the parent must start N ranks as a child torchrun job BEFORE it touches the
GPU (a process that has initialised HIP must not exec/fork GPU children on this pool) and relay rank 0's JSON line."""
import json
import os
import subprocess
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_gpus_n_spawns_torchrun_before_any_gpu_call(monkeypatch, capsys):
    import torch
    import bench
    calls = {}

    def fake_run(cmd, env=None, stdout=None, text=None):
        calls["cmd"], calls["env"] = cmd, env
        return types.SimpleNamespace(returncode=0, stdout='noise\n{"n_gpus": 2, "value": 1.0}\n')

    def boom(*a, **k):
        raise AssertionError("the parent touched the GPU before spawning its ranks")

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(torch.cuda, "is_available", boom)
    monkeypatch.setattr(torch.cuda, "device_count", boom)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = calls["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "2"
    assert "--master-addr" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert calls["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    out = capsys.readouterr().out.strip().splitlines()
    assert len(out) == 1 and json.loads(out[0])["n_gpus"] == 2          # exactly rank 0's JSON line is relayed


def test_launched_rank_does_not_respawn(monkeypatch):
    """Under the driver's own torchrun (WORLD_SIZE set) bench.py must not start another launcher."""
    import bench
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(bench, "spawn_ranks", lambda n: (_ for _ in ()).throw(AssertionError("respawned")))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    import torch
    monkeypatch.setattr(torch.cuda, "is_available", lambda: False)
    with pytest.raises(AssertionError, match="needs a HIP device"):
        bench.main()
