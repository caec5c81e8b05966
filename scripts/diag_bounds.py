"""Where the per-cell error maximum of the tuned 3-D sweep sits, and the per-level / per-variable error of the config-5
residual (GPU box): python scripts/diag_bounds.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import bench  # noqa: E402
import ibamd  # noqa: E402
from oracle import residual_c as rc  # noqa: E402

f32 = np.float32
out = {}
msh = bench.build_mesh("sphere3d_1.6M")
dom = ibamd.Domain(msh, max_partition_size=10 ** 9, boundaries=False)
(part,) = dom.partitions.values()
dpart = ibamd.to_backend(part, ibamd.hip)
u, C = bench.synthetic_fields(part.centers)
exp = rc.CPart(part).residual_advection(u, C).astype(np.float64)
got = ibamd.to_host(ibamd.residual_advection(dpart, ibamd.hip(u), ibamd.hip(C))).astype(np.float64)
h = part.spacing.min(axis=1).astype(np.float64)
scale = np.abs(exp) + np.abs(u.astype(np.float64)) / h
rel = np.abs(got - exp) / scale
order = np.argsort(rel)[::-1][:8]
# the sensor of the cell and of its neighbours from the oracle's operators: is the maximum a sensor-ratio effect?
from conftest import oracle_view  # noqa: E402
from oracle import domain as od  # noqa: E402
opart = oracle_view(part)
D = od.JST_sensor(opart, u)
top = []
for c in order:
    c = int(c)
    pos = c % 512
    top.append(dict(cell=c, rel=float(rel[c]), exp=float(exp[c]), got=float(got[c]), u=float(u[c]), h=float(h[c]),
                    block=c // 512, ijk=[pos % 8, (pos // 8) % 8, pos // 64], sensor=float(D[c]),
                    abs_err_over_u_per_h=float(abs(got[c] - exp[c]) / (abs(u[c]) / h[c]))))
out["sweep3d_percell"] = dict(max=float(rel.max()), p999=float(np.percentile(rel, 99.9)),
                              norm_wise=float(np.abs(got - exp).max() / np.abs(exp).max()),
                              count_above_2e5=int((rel > 2e-5).sum()), count_above_1e5=int((rel > 1e-5).sum()), top=top)
print(json.dumps(out))


def stencil_scale(part, u, exp):
    """|ref| + (max |u| over the cell and its face neighbours) / h: the local scale of a residual that is a difference of
    neighbour values (|u| of the cell alone vanishes where u crosses zero)."""
    m = np.abs(u).astype(np.float64)
    a = m.copy()
    for d in range(1, part.ndims + 1):
        o, nb = part.face_owners_neighbors[d][0], part.face_owners_neighbors[d][1]
        np.maximum.at(m, o, a[nb])
        np.maximum.at(m, nb, a[o])
    return np.abs(exp) + m / part.spacing.min(axis=1).astype(np.float64)


rel2 = np.abs(got - exp) / stencil_scale(part, u, exp)
res = dict(sweep3d_stencil_scale=dict(max=float(rel2.max()), p999=float(np.percentile(rel2, 99.9))))
# 2-D headline
msh2 = bench.build_mesh("rae2822_0.87M")
dom2 = ibamd.Domain(msh2, max_partition_size=10 ** 9, boundaries=False)
(part2,) = dom2.partitions.values()
dpart2 = ibamd.to_backend(part2, ibamd.hip)
u2, C2 = bench.synthetic_fields(part2.centers)
exp2 = rc.CPart(part2).residual_advection(u2, C2).astype(np.float64)
got2 = ibamd.to_host(ibamd.residual_advection(dpart2, ibamd.hip(u2), ibamd.hip(C2))).astype(np.float64)
r_old = np.abs(got2 - exp2) / (np.abs(exp2) + np.abs(u2.astype(np.float64)) / part2.spacing.min(axis=1).astype(np.float64))
r_new = np.abs(got2 - exp2) / stencil_scale(part2, u2, exp2)
res["sweep2d"] = dict(old_max=float(r_old.max()), new_max=float(r_new.max()), new_p999=float(np.percentile(r_new, 99.9)))
print(json.dumps(res))
