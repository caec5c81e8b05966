"""Direct xGMI halo exchange (IPC-mapped buffers + device flags) with 2 processes sharing the one GPU of the
test box: must reproduce the reference exchange bit for bit (nv = 1 and 3) and survive HIP-graph capture of
overlapped sweeps (scripts/rehearse_xgmi.py)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_xgmi_halo_two_ranks_one_gpu():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "scripts", "rehearse_xgmi.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "xgmi == reference exchange: True" in r.stdout
    assert "graph-captured overlapped sweeps with xGMI exchange match: True" in r.stdout
    assert "fused exchange + sweep step matches exchange-then-sweep: True" in r.stdout


def test_device_fas_across_two_ranks_one_gpu():
    """`FAS!` over `multigrid(dom)` across ranks, device resident (distributed.RankLevels, solver.FAS hooks; 2 processes on
    the one GPU, gloo group): reproduces the one-partition device V-cycle on the owned cells (scripts/rehearse_fas.py)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "scripts", "rehearse_fas.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "distributed device V-cycle == one-partition device V-cycle on the owned cells: True" in r.stdout, r.stdout[-1500:]


def test_device_point_implicit_across_two_ranks_one_gpu():
    """The point-implicit smoother (orphan src/point_implicit.jl) across ranks, device resident (point_implicit.py with
    distributed.RankOps; 2 processes on the one GPU, gloo group): right-hand side and inverse Hutchinson blocks bit for bit,
    two relaxation steps to the rounding of the all-reduced sums, against the one-partition device smoother
    (scripts/rehearse_pi.py)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "scripts", "rehearse_pi.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "distributed device point-implicit smoother == one-partition device smoother on the owned cells: True" in r.stdout, \
        r.stdout[-1500:]
