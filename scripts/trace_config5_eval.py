"""Where the ATen kernels (copies, adds) of a configs[4] V-cycle come from: one V-cycle under torch.profiler with Python stacks.
Run on the GPU box: python scripts/trace_config5_eval.py [workload]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

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


for _ in range(2):
    ibamd.FAS(f, Q, coarseners=coar, prolongators=prol, n_iter=2, rtol=1e-9)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    ibamd.FAS(f, Q, coarseners=coar, prolongators=prol, n_iter=2, rtol=1e-9)
    torch.cuda.synchronize()
rows = {}
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
        continue
    dev = getattr(ev, "device_time_total", None)
    if dev is None:
        dev = getattr(ev, "cuda_time_total", 0)
    fr = [s for s in (ev.stack or []) if "/repo/" in s and "trace_config5" not in s][:2]
    key = (ev.name, str(ev.input_shapes)[:60], " <- ".join(s.replace(ROOT + "/", "") for s in fr))
    r = rows.setdefault(key, [0, 0.0])
    r[0] += 1
    r[1] += dev
tot = sum(r[1] for r in rows.values())
print(f"top-level ATen ops of one V-cycle at {wl}: {tot / 1e3:.2f} ms of device time")
for k, r in sorted(rows.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{r[1] / 1e3:8.3f} ms {r[0]:4d} x {k[0]:22s} {k[1]:60s} {k[2]}")
