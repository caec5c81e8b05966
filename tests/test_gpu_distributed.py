"""distributed.py with the real kernels: 4 processes sharing the one GPU of the test box (gloo handshake, staged
exchange) run `impose_bc!` + the image-only Euler sweep + a fixed-point update + the all-reduced norm on 4 partitions of
the RAE2822 case and reproduce the one-partition run (scripts/rehearse_distributed.py)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_distributed_bc_sweep_norm_four_ranks_one_gpu():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "scripts", "rehearse_distributed.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "match the one-partition run: True" in r.stdout
