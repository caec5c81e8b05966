"""Wave timeline of one quad-sweep launch (quad_variant 4 stamps s_memrealtime + HW_ID per wave):
python scripts/wave_timeline.py [workload]   (GPU box)"""
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

name = sys.argv[1] if len(sys.argv) > 1 else "rae2822_0.87M"
msh = bench.build_mesh(name)
dom = ibamd.Domain(msh, max_partition_size=10 ** 9, boundaries=False)
(part,) = dom.partitions.values()
u_h, C_h = bench.synthetic_fields(part.centers)
dpart = ibamd.to_backend(part, ibamd.hip)
u, C = ibamd.hip(u_h), ibamd.hip(C_h)
ud = torch.zeros(dpart.nc, dtype=torch.float32, device=u.device)
nq, ns = int(dpart.info["quads"]), int(dpart.info["quad_singles"])
nwaves = 4 * 4096  # more than any launch of the persistent sweep has
buf = torch.zeros(nwaves * 8, dtype=torch.int64, device=u.device)
_lib.call("ibh_set_tuning", b"quad_variant", 4)
for _ in range(5):
    ibamd.residual_advection(dpart, u, C, out=ud)
torch.cuda.synchronize()
_lib.call("ibh_debug_buffer", _lib.c_vp(buf.data_ptr()))
for _ in range(3):   # keep the last launch
    buf.zero_()
    ibamd.residual_advection(dpart, u, C, out=ud)
    torch.cuda.synchronize()
_lib.call("ibh_debug_buffer", _lib.c_vp(None))
b = buf.cpu().numpy().reshape(-1, 8)
b = b[b[:, 0] != 0]
t0 = b[:, 0].min()
start, end = (b[:, 0] - t0) * 0.01, (b[:, 1] - t0) * 0.01   # us
hw = b[:, 2]
isq = (b[:, 3] & 0xffffffff) > 0   # waves that swept quads
# HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13] (+ XCC from XCC_ID reg)
simd = (hw >> 4) & 3
cu = (hw >> 8) & 15
sh = (hw >> 12) & 1
se = (hw >> 13) & 7
key = ((se * 2 + sh) * 16 + cu) * 4 + simd
out = {"waves": int(len(b)), "quad_waves": int(isq.sum()), 
       "kernel_span_us": round(float(end.max()), 2),
       "start_us_pct": [round(float(np.percentile(start, p)), 2) for p in (0, 10, 50, 90, 100)],
       "end_us_pct": [round(float(np.percentile(end, p)), 2) for p in (0, 10, 50, 90, 100)],
       "quad_dur_us_pct": [round(float(np.percentile((end - start)[isq], p)), 2) for p in (0, 10, 50, 90, 100)],
       "single_dur_us_pct": ([round(float(np.percentile((end - start)[~isq], p)), 2) for p in (0, 10, 50, 90, 100)]
                             if (~isq).any() else []),
       "distinct_simd_keys": int(len(np.unique(key))),
       "waves_per_simd_key_pct": [int(np.percentile(np.bincount(key)[np.bincount(key) > 0], p)) for p in (0, 50, 100)]}
ph = (b[isq][:, 4:8] - b[isq][:, 0:1]) * 0.01   # staged, rings, tiles written, edge fluxes  (us after the wave's start)
out["quad_phase_us_median"] = {"staged(loads+gathers arrived)": round(float(np.median(ph[:, 0])), 2),
                               "rings": round(float(np.median(ph[:, 1])), 2),
                               "own+halo S/D, interior fluxes": round(float(np.median(ph[:, 2])), 2),
                               "edge fluxes": round(float(np.median(ph[:, 3])), 2),
                               "end": round(float(np.median((end - start)[isq])), 2)}
print(json.dumps(out))
