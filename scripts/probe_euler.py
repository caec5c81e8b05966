"""Parts of the 2-D Euler quad sweep (run on the GPU box): quads only, single blocks only, both, the per-block kernel.
python scripts/probe_euler.py [workload]"""
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

name = sys.argv[1] if len(sys.argv) > 1 else "rae2822_0.87M"
msh = bench.build_mesh(name)
dom = ibamd.Domain(msh, max_partition_size=10 ** 9, boundaries=False)
(part,) = dom.partitions.values()
dpart = ibamd.to_backend(part, ibamd.hip)
rng = np.random.default_rng(1)
n = part.centers.shape[0]
Ph = np.empty((n, 4), dtype=np.float32)
Ph[:, 0] = 1e5 * (1 + 0.05 * rng.uniform(-1, 1, n))
Ph[:, 1] = 288.15 * (1 + 0.05 * rng.uniform(-1, 1, n))
Ph[:, 2] = 100.0 * (1 + 0.1 * rng.uniform(-1, 1, n))
Ph[:, 3] = 100.0 * (1 + 0.1 * rng.uniform(-1, 1, n))
P = ibamd.hip(Ph)
R = torch.zeros_like(P)


def timed(fn, n=50, reps=20):
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


out = {"workload": name, "cells": int(dpart.nc), "quads": int(dpart.info["quads"]),
       "quad_singles": int(dpart.info["quad_singles"])}
for key, parts in (("both_us", 3), ("quads_only_us", 1), ("singles_only_us", 2)):
    _lib.call("ibh_set_tuning", b"quad_parts", parts)
    out[key] = round(timed(lambda: ibamd.residual_euler_hll(dpart, P, out=R)), 3)
_lib.call("ibh_set_tuning", b"quad_parts", 3)
_lib.call("ibh_set_tuning", b"quad_singles_first", 1)
out["both_singles_first_us"] = round(timed(lambda: ibamd.residual_euler_hll(dpart, P, out=R)), 3)
_lib.call("ibh_set_tuning", b"quad_singles_first", 0)
_lib.call("ibh_set_tuning", b"quad_parts", 3)
out["per_block_kernel_us"] = round(timed(lambda: ibamd.residual_euler_hll(dpart, P, out=R, flags=ibamd.IBH_NO_QUAD)), 3)
print(json.dumps(out))
