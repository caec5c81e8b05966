"""Host-side profile (cProfile) of configs[4] V-cycles at a size where the host's launch rate bounds them.
Run on the GPU box: python scripts/profile_host_config5.py [workload]"""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import ibamd  # noqa: E402
from ibamd.closures import config5_boundary_conditions, navier_stokes_wray_agarwal_residual  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "sphere3d_1.6M"
msh = bench.build_mesh(wl)
fam = [("farfield", [(d, sd) for d in (1, 2, 3) for sd in (False, True)])]
dom = ibamd.Domain(msh, hypercube_families=fam, max_partition_size=10 ** 9)
cds, prol, coar = ibamd.multigrid(dom, max_levels=2)
doms = [dom] + list(cds)
levels = [ibamd.to_backend(d.partitions[1], ibamd.hip) for d in doms]
for a in list(prol) + list(coar):
    ibamd.to_backend(a)
for d in doms:
    for v in d.boundaries.values():
        for b in v.values():
            ibamd.to_backend(b, ibamd.hip)
nc = levels[0].nc
rng = np.random.default_rng(0)
Q0 = np.empty((nc, 6), np.float32)
Q0[:, 0] = 1e5 * (1 + 0.02 * rng.uniform(-1, 1, nc))
Q0[:, 1] = 288.15 * (1 + 0.02 * rng.uniform(-1, 1, nc))
Q0[:, 2] = 100.0
Q0[:, 3:5] = rng.uniform(-1, 1, (nc, 2))
Q0[:, 5] = 4.5e-5
Q = ibamd.hip(Q0)
FAR = [1.0e5, 288.15, 100.0, 0.0, 0.0]


def f(level, q):
    config5_boundary_conditions(doms[level], q, FAR)
    return navier_stokes_wray_agarwal_residual(levels[level], q), 2e-7


def cycles(k):
    for _ in range(k):
        ibamd.FAS(f, Q, coarseners=coar, prolongators=prol, n_iter=2, rtol=1e-9)
    torch.cuda.synchronize()


cycles(2)
t0 = time.perf_counter()
cycles(5)
print(f"{wl}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per V-cycle")
pr = cProfile.Profile()
pr.enable()
cycles(5)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
