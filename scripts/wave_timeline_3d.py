"""Phase timeline of the waves of one 3-D Euler column sweep (quad_variant 4: s_memrealtime stamps per wave):
python scripts/wave_timeline_3d.py [workload]   (GPU box)"""
import json
import os
import sys

import numpy as np

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
rng = np.random.default_rng(1)
n = part.centers.shape[0]
Ph = np.empty((n, 5), dtype=np.float32)
Ph[:, 0] = 1e5 * (1 + 0.05 * rng.uniform(-1, 1, n))
Ph[:, 1] = 288.15 * (1 + 0.05 * rng.uniform(-1, 1, n))
for k in range(3):
    Ph[:, 2 + k] = 100.0 * (1 + 0.1 * rng.uniform(-1, 1, n))
P = ibamd.hip(Ph)
R = torch.zeros_like(P)
nblk = int(dpart.info["full_blocks"])
buf = torch.zeros(nblk * 8, dtype=torch.int64, device=P.device)
_lib.call("ibh_set_tuning", b"quad_variant", 4)
for _ in range(3):
    ibamd.residual_euler_hll(dpart, P, out=R)
torch.cuda.synchronize()
_lib.call("ibh_debug_buffer", _lib.c_vp(buf.data_ptr()))
for _ in range(2):   # keep the last launch
    buf.zero_()
    ibamd.residual_euler_hll(dpart, P, out=R)
    torch.cuda.synchronize()
_lib.call("ibh_debug_buffer", _lib.c_vp(None))
_lib.call("ibh_set_tuning", b"quad_variant", 0)
b = buf.cpu().numpy().reshape(-1, 8)
b = b[b[:, 0] != 0]
t0 = b[:, 0].min()
us = (b - t0) * 0.01
names = ["first loads landed", "sensor", "x fluxes", "x->y transposes", "y fluxes", "y->z transposes", "z fluxes + stores"]
dur = np.diff(us, axis=1)


def pct(a):
    return [round(float(np.percentile(a, p)), 2) for p in (10, 50, 90)]


# residency: waves alive at a few instants
alive = {str(t): int(((us[:, 0] <= t) & (us[:, 7] > t)).sum()) for t in (2, 5, 10, 20, 30, 50, 70, 90)}
out = {"workload": name, "waves": int(len(b)), "kernel_span_us": round(float(us[:, 7].max()), 2), "alive_at_us": alive,
       "wave_life_us_p10_50_90": pct(us[:, 7] - us[:, 0]),
       "phase_us_p10_50_90": {nm: pct(dur[:, k]) for k, nm in enumerate(names)},
       "start_us_p10_50_90": pct(us[:, 0])}
print(json.dumps(out))
