"""bench.py starts its own ranks when it is run plainly with --gpus N (no WORLD_SIZE in the environment)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=timeout, env=env, cwd=ROOT)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_plain_invocation_spawns_the_ranks():
    out = _run(["--gpus", "2", "--rendezvous-only"], 300)
    assert out == {"n_gpus": 2, "rendezvous": "ok"}


@pytest.mark.gpu
def test_two_rank_rehearsal_through_the_plain_entry():
    """Two ranks on the one GPU of the test box (gloo process group, device-side halo exchange over HIP IPC where it
    verifies, else the staged one): `python bench.py --gpus 2 ...` prints ONE JSON line with n_gpus = 2."""
    out = _run(["--gpus", "2", "--backend", "gloo", "--same-device", "--steps", "40", "--warmup", "4", "--repeats", "3",
                "--no-cpu-baseline", "--workload", "rae2822_37k"], 900)
    assert out["n_gpus"] == 2 and out["steps"] == 40 and out["value"] > 0
    assert out["config"]["halo"]["timeouts"] in (None, 0)
