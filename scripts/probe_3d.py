"""A/B of the 3-D scalar sweep forms on one workload (run on the GPU box): python scripts/probe_3d.py [workload]
column form (cols3::sweep_cols), strip form (strip3::sweep_strip), thread-per-cell single kernel, two-kernel form."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import ibamd  # noqa: E402
from ibamd import _lib  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "sphere3d_4.6M"
msh = bench.build_mesh(name)
dom = ibamd.Domain(msh, max_partition_size=10 ** 9, boundaries=False)
(part,) = dom.partitions.values()
dpart = ibamd.to_backend(part, ibamd.hip)
u_h, C_h = bench.synthetic_fields(part.centers)
u, C = ibamd.hip(u_h), ibamd.hip(C_h)
ud = torch.zeros(dpart.nc, dtype=torch.float32, device=u.device)


def timed(fn, n=20, reps=15):
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
for key, var in (("strip_3_waves_per_simd_us", 518), ("cols_3_waves_per_simd_us", 0), ("cols_4_waves_per_simd_us", 519),
                 ("cols_5_waves_per_simd_us", 520), ("strip_2_waves_per_simd_us", 515), ("strip_4_waves_per_simd_us", 514),
                 ("thread_per_cell_us", 512)):
    _lib.call("ibh_set_tuning", b"quad_variant", var)
    out[key] = round(timed(lambda: ibamd.residual_advection(dpart, u, C, out=ud)), 3)
    r = ud.clone()
    if ref is None:
        ref = r
    else:
        out[key.replace("_us", "_maxdiff")] = float((r - ref).abs().max() / ref.abs().max())
_lib.call("ibh_set_tuning", b"quad_variant", 0)
out["two_kernel_us"] = round(timed(lambda: ibamd.residual_advection(dpart, u, C, out=ud, flags=ibamd.IBH_NO_FUSE)), 3)
out["two_kernel_maxdiff"] = float((ud - ref).abs().max() / ref.abs().max())
print(json.dumps(out))
