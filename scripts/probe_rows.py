"""A/B of the 2-D scalar sweep forms on one workload (GPU box): python scripts/probe_rows.py [workload]
row / column sweep (rows2::sweep_rows, eight blocks per wave) against the quad sweep (+ single blocks)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import ibamd  # noqa: E402
from ibamd import _lib  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "rae2822_0.87M"
msh = bench.build_mesh(name)
dom = ibamd.Domain(msh, max_partition_size=10 ** 9, boundaries=False)
(part,) = dom.partitions.values()
dpart = ibamd.to_backend(part, ibamd.hip)
u_h, C_h = bench.synthetic_fields(part.centers)
u, C = ibamd.hip(u_h), ibamd.hip(C_h)
ud = torch.zeros(dpart.nc, dtype=torch.float32, device=u.device)


def timed(fn, n=50, reps=15):
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
       "row_sweep_eligible": bool(dpart.info["row_sweep"]), "quads": int(dpart.info["quads"]),
       "quad_singles": int(dpart.info["quad_singles"])}
ref = None
for key, rows in (("quad_sweep_us", 0), ("row_sweep_us", 1)):
    _lib.call("ibh_set_tuning", b"rows", rows)
    ud.zero_()
    out[key] = round(timed(lambda: ibamd.residual_advection(dpart, u, C, out=ud)), 3)
    r = ud.clone()
    if ref is None:
        ref = r
    else:
        out["maxdiff"] = float((r - ref).abs().max() / ref.abs().max())
_lib.call("ibh_set_tuning", b"rows", 0)
# the blocks outside quads: per-block body in the quad launch (0), second launch of the row sweep (1), row waves inside the
# quad launch (2), the latter two with either grid order
for rs in (0, 1, 2):
    for sf in (0, 1):
        if rs == 1 and sf:
            continue
        _lib.call("ibh_set_tuning", b"rows_singles", rs)
        _lib.call("ibh_set_tuning", b"quad_singles_first", sf)
        out["singles_form_%d_singles_first_%d_us" % (rs, sf)] = round(timed(lambda: ibamd.residual_advection(dpart, u, C, out=ud)), 3)
_lib.call("ibh_set_tuning", b"rows_singles", -1)
_lib.call("ibh_set_tuning", b"quad_singles_first", 0)
for key in ("quad_sweep_us", "row_sweep_us"):
    out[key.replace("_us", "_frac")] = round(16.0 * dpart.nc / (out[key] * 1e-6) / 8e12, 4)
print(json.dumps(out))
