"""A/B of the 3-D Euler sweep forms on one workload (run on the GPU box): python scripts/probe_3d_euler.py [workload]
column form (strip3e::sweep_block) as one block per wave / persistent chains, thread-per-cell single kernel, two-kernel
form."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
import ibamd  # noqa: E402
from ibamd import _lib  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "sphere3d_1.6M"
msh = bench.build_mesh(name)
dom = ibamd.Domain(msh, max_partition_size=10 ** 9, boundaries=False)
(part,) = dom.partitions.values()
dpart = ibamd.to_backend(part, ibamd.hip)
rng = np.random.default_rng(1)
n = part.centers.shape[0]
Ph = np.empty((n, 5), dtype=np.float32)
Ph[:, 0] = 1e5 * (1 + 0.05 * rng.uniform(-1, 1, n))
Ph[:, 1] = 288.15 * (1 + 0.05 * rng.uniform(-1, 1, n))
for k in range(3):
    Ph[:, 2 + k] = 100.0 * (1 + 0.1 * rng.uniform(-1, 1, n))
P = ibamd.hip(Ph)
R = torch.zeros_like(P)


def timed(fn, n=10, reps=15):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(n):
                fn()
        g.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            g.replay()
            e1.record(side)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / n)
    ts.sort()
    return ts[len(ts) // 2]


out = {"workload": name, "cells": int(dpart.nc), "blocks": int(dpart.info["full_blocks"]),
       "single_kernel_blocks": int(dpart.info["fusable_blocks"])}
ref = None
# (round 4: 0 = one block per wave at two waves per SIMD, the default; 514 = persistent waves, each working through a chain of
# blocks with the first loads of the next block requested by LDS-DMA during the z fluxes)
for key, var in (("thread_per_cell_us", 512), ("cols_block_per_wave_us", 0), ("cols_persistent_chain_us", 514)):
    _lib.call("ibh_set_tuning", b"quad_variant", var)
    R.zero_()
    out[key] = round(timed(lambda: ibamd.residual_euler_hll(dpart, P, out=R)), 3)
    r = R.clone()
    if ref is None:
        ref = r
    else:
        out[key.replace("_us", "_maxdiff")] = [float((r[:, v] - ref[:, v]).abs().max() / ref[:, v].abs().max())
                                               for v in range(5)]
_lib.call("ibh_set_tuning", b"quad_variant", 0)
out["two_kernel_us"] = round(timed(lambda: ibamd.residual_euler_hll(dpart, P, out=R, flags=ibamd.IBH_NO_FUSE)), 3)
out["two_kernel_maxdiff"] = [float((R[:, v] - ref[:, v]).abs().max() / ref[:, v].abs().max()) for v in range(5)]
out["frac_cols"] = round(40.0 * dpart.nc / (out["cols_block_per_wave_us"] * 1e-6) / 8e12, 4)
print(json.dumps(out))
