"""Where the duration of the 2-D sweep goes: times ibh_probe_sweep modes 0..2 and the sweep itself with HIP events over
a graph of back-to-back launches (run on the GPU box):  python scripts/probe_sweep.py [workload]"""
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
u_h, C_h = bench.synthetic_fields(part.centers)
dpart = ibamd.to_backend(part, ibamd.hip)
u, C = ibamd.hip(u_h), ibamd.hip(C_h)
ud = torch.zeros(dpart.nc, dtype=torch.float32, device=u.device)
ldc = C.stride(1)


def probe(mode):
    _lib.call("ibh_set_stream", _lib.c_vp(torch.cuda.current_stream().cuda_stream))
    _lib.call("ibh_probe_sweep", dpart.handle, u.data_ptr(), C.data_ptr(), ldc, ud.data_ptr(), mode)


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
for mode, key in ((0, "dispatch_only_us"), (1, "stream_16B_per_cell_us"), (2, "stream_tables_gathers_us")):
    out[key] = round(timed(lambda: probe(mode)), 3)
if os.environ.get("PROBE_DETAIL"):
    for bits, key in ((3, "own+tables"), (2, "tables_only"), (2 + 4, "tables+hu0"), (3 + 4, "own+tables+hu0"),
                      (3 + 4 + 16, "own+tables+hu0+hd0"), (3 + 4 + 16 + 64, "own+tables+hu0+hd0+hc0"),
                      (3 + 4 + 16 + 64 + 256, "all_k0"), (3 + 4 + 8 + 16 + 32, "own+tables+hu01+hd01"),
                      (1 + 2 + 256, "own+tables+ends")):
        out[key] = round(timed(lambda: probe(16 + bits)), 3)
if os.environ.get("PROBE_DISPATCH"):
    nwaves = 4 * 1119

    def disp(nwg, th, lds):
        def f():
            _lib.call("ibh_set_stream", _lib.c_vp(torch.cuda.current_stream().cuda_stream))
            _lib.call("ibh_probe_dispatch", nwg, th, lds)
        return round(timed(f), 3)
    out["dispatch_shapes_us"] = {f"{nwaves * 64 // th}x{th}_lds{lds}": disp(nwaves * 64 // th, th, lds)
                                 for th in (64, 128, 256, 512, 1024) for lds in (0, min(65536, 6784 * th // 64))}
    out["dispatch_shapes_us"]["1x64_lds0"] = disp(1, 64, 0)
    out["dispatch_shapes_us"]["256x256_lds0"] = disp(256, 256, 0)
out["sweep_us"] = round(timed(lambda: ibamd.residual_advection(dpart, u, C, out=ud)), 3)
def sweep_time():
    return timed(lambda: ibamd.residual_advection(dpart, u, C, out=ud))


if os.environ.get("QUAD_TUNE"):  # same-process A/B: alternate, keep the best median of each
    best = {}
    cases = {"both_quads_first": (3, 0), "both_singles_first": (3, 1), "singles_only": (2, 0)}
    for _ in range(3):
        for name_, (parts, sf) in cases.items():
            _lib.call("ibh_set_tuning", b"quad_parts", parts)
            _lib.call("ibh_set_tuning", b"quad_singles_first", sf)
            best[name_] = min(best.get(name_, 1e9), sweep_time())
    _lib.call("ibh_set_tuning", b"quad_parts", 3)
    _lib.call("ibh_set_tuning", b"quad_singles_first", 0)
    for _ in range(3):
        for it in (1, 2, 3):
            _lib.call("ibh_set_tuning", b"quad_singles_iters", it)
            best[f"both_singles_{it}_per_wave"] = min(best.get(f"both_singles_{it}_per_wave", 1e9), sweep_time())
    _lib.call("ibh_set_tuning", b"quad_singles_iters", 1)
    _lib.call("ibh_set_tuning", b"quad_parts", 1)
    for _ in range(3):
        for name_, var in (("quads_only_paired_gathers(default)", 0), ("quads_only_seven_gathers", 126), ("quads_only_k0+ends", 85), ("quads_only_hu0_hd0_ends", 69),
                           ("quads_only_hu0_hd0", 5), ("quads_only_no_gathers", 100)):
            _lib.call("ibh_set_tuning", b"quad_variant", var)
            best[name_] = min(best.get(name_, 1e9), sweep_time())
    _lib.call("ibh_set_tuning", b"quad_variant", 0)
    out["parts_us"] = {k: round(v, 3) for k, v in best.items()}
    _lib.call("ibh_set_tuning", b"quad_parts", 3)
    _lib.call("ibh_set_tuning", b"quad_singles_first", 0)
out["sweep_per_block_kernel_us"] = round(timed(lambda: ibamd.residual_advection(dpart, u, C, out=ud,
                                                                                  flags=ibamd.IBH_NO_QUAD)), 3)
print(json.dumps(out))
