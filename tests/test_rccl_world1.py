"""RCCL on the one GPU of the box: a world of one rank accepts every collective the multi-GPU paths issue, on the
tensors they issue them on (one process per GPU: torch.distributed "nccl"; one process for the node: the group's own
RCCL transport).  Each runs in a process of its own with a time limit, so that a collective that never returns ends the
test instead of the suite."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(script):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script)], cwd=ROOT, capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    return r.stdout


def test_torch_distributed_nccl_world_of_one():
    assert "rehearsal: ok" in _run("rehearse_rccl_world1.py")


def test_group_transport_rccl_world_of_one():
    out = _run("rehearse_group_rccl.py")
    assert "rehearsal: ok" in out and "DIFFERENT" not in out
