"""The closure of test/advection.jl:67-83 at OPERATOR granularity (every operator and broadcast node a launch, device
arrays in, device arrays out) next to the fused sweep, on the headline mesh (run on the GPU box)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import ibamd  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "rae2822_0.87M"
msh = bench.build_mesh(name)
dom = ibamd.Domain(msh, max_partition_size=10 ** 9, boundaries=False)
(part,) = dom.partitions.values()
dpart = ibamd.to_backend(part, ibamd.hip)
u_h, C_h = bench.synthetic_fields(part.centers)
u, C = ibamd.HipArray(u_h), ibamd.HipArray(C_h)
ud = ibamd.HipArray(torch.zeros(dpart.nc, dtype=torch.float32, device="cuda"))


def closure(part, u, ud, C):
    D = ibamd.JST_sensor(part, u)
    for dim in range(1, part.ndims + 1):
        Cf = ibamd.at_faces(part, C.col(dim), dim)
        gu = ibamd.cell_gradient(part, u, dim)
        uL, uR = ibamd.MUSCL(part, u, gu, dim, D=D, high_order=True)
        ud -= ibamd.green_gauss(part, (uL + uR) * Cf / 2 + abs(Cf) * (uL - uR) / 2, dim)


def wall(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


out = {"workload": name, "cells": int(dpart.nc)}
out["operator_closure_wall_us"] = round(wall(lambda: closure(dpart, u, ud, C)), 1)
ut, Ct, udt = u.t, C.t, ud.t
out["fused_sweep_eager_wall_us"] = round(wall(lambda: ibamd.residual_advection(dpart, ut, Ct, out=udt)), 1)
# GPU time of the closure: events around a batch
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(20):
    closure(dpart, u, ud, C)
e1.record()
torch.cuda.synchronize()
out["operator_closure_gpu_us"] = round(e0.elapsed_time(e1) * 1e3 / 20, 1)
# the same closure captured once in a HIP graph (ibamd.GraphedClosure) and replayed
g = ibamd.GraphedClosure(closure, dpart, u, ud, C)
ref = ud.t.clone()
closure(dpart, u, ud, C)
eager = ud.t.clone()
ud.t.copy_(ref)
g()
torch.cuda.synchronize()
out["graphed_closure_equals_eager"] = bool(torch.equal(ud.t, eager))
out["graphed_closure_wall_us"] = round(wall(g), 1)
torch.cuda.synchronize()
e0.record()
for _ in range(20):
    g()
e1.record()
torch.cuda.synchronize()
out["graphed_closure_gpu_us"] = round(e0.elapsed_time(e1) * 1e3 / 20, 1)
print(json.dumps(out))
