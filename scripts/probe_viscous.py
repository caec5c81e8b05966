"""ibh_viscous_residual on a 3-D sphere octree: LDS-shared faces (default) against one thread per cell (ibh_set_tuning viscous_per_cell).
Run on the GPU box: python scripts/probe_viscous.py [workload]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import ibamd  # noqa: E402
from ibamd import _lib, cfd  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "sphere3d_4.6M"
msh = bench.build_mesh(wl)
dom = ibamd.Domain(msh, max_partition_size=10 ** 9, boundaries=False)
part = dom.partitions[1]
dpart = ibamd.to_backend(part, ibamd.hip)
nc = part.centers.shape[0]
rng = np.random.default_rng(0)
P = np.empty((nc, 5), np.float32)
P[:, 0] = 1e5 * (1 + 0.02 * rng.uniform(-1, 1, nc))
P[:, 1] = 288.15 * (1 + 0.02 * rng.uniform(-1, 1, nc))
P[:, 2:] = 50 * rng.uniform(-1, 1, (nc, 3))
P = ibamd.hip(P)
mut = ibamd.hip((1e-4 * rng.uniform(0, 1, nc)).astype(np.float32))
gV = ibamd.cell_gradient(dpart, P[:, 2:])
fluid = cfd.Fluid()
out = {"workload": wl, "cells": int(nc)}
res = {}
for name, var in (("lds_shared_faces_us", 0), ("thread_per_cell_us", 1), ("lds_shared_faces_again_us", 0)):
    _lib.call("ibh_set_tuning", b"viscous_per_cell", var)
    R = ibamd.hip(np.zeros((nc, 5), np.float32))
    for _ in range(3):
        cfd.viscous_residual(dpart, fluid, P, gV, mut, R, velocity_gradients_only=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        cfd.viscous_residual(dpart, fluid, P, gV, mut, R, velocity_gradients_only=True)
    e1.record()
    torch.cuda.synchronize()
    out[name] = round(e0.elapsed_time(e1) * 1e3 / 20, 1)
    R = ibamd.hip(np.zeros((nc, 5), np.float32))
    cfd.viscous_residual(dpart, fluid, P, gV, mut, R, velocity_gradients_only=True)
    res[var] = R.clone()
_lib.call("ibh_set_tuning", b"viscous_per_cell", 0)
out["bit_identical"] = bool(torch.equal(res[0], res[1]))
print(json.dumps(out))
