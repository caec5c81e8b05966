"""Diagnostic: the config-4 step at sphere3d_8M -- is the residual finite with the default sweep and with the thread-per-cell
sweep (quad_variant 512), and where not."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import ibamd  # noqa: E402
from ibamd import _lib  # noqa: E402
from ibamd import cfd as gcfd  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "sphere3d_8M"
msh = bench.build_mesh(wl)
fam = [("farfield", [(d, sd) for d in (1, 2, 3) for sd in (False, True)])]
dom = ibamd.Domain(msh, hypercube_families=fam, max_partition_size=10 ** 9)
part = dom.partitions[1]
dpart = ibamd.to_backend(part, ibamd.hip)
print({k: dpart.info[k] for k in ("full_blocks", "sides_same", "sides_mirror", "sides_coarse", "sides_fine")})
rng = np.random.default_rng(12345)
n = part.centers.shape[0]
P_h = np.empty((n, 5), dtype=np.float32)
P_h[:, 0] = 1e5 * (1 + 0.05 * rng.uniform(-1, 1, n))
P_h[:, 1] = 288.15 * (1 + 0.05 * rng.uniform(-1, 1, n))
for k in range(2, 5):
    P_h[:, k] = 100.0 * (1 + 0.1 * rng.uniform(-1, 1, n))
P_h[:, 0] = 1e5 * (1 + 0.02 * rng.uniform(-1, 1, n))
P_h[:, 1] = 288.15 * (1 + 0.02 * rng.uniform(-1, 1, n))
P_h[:, 3:] = 10.0 * rng.uniform(-1, 1, (n, 2))
P = ibamd.hip(P_h)
far_bc = gcfd.FlowBC(gcfd.Fluid(), [1.0e5, 288.15, 100.0, 0.0, 0.0])
wall_bc = gcfd.FlowBC(gcfd.Fluid(), [1.0e5, 288.15, 0.0], normal_flow=True)
for it in range(140):
    ibamd.impose_bc(lambda b, ia: far_bc(ia, b.normals), dom, "farfield", P)
    ibamd.impose_bc(lambda b, ia: wall_bc(ia, b.normals), dom, "sphere", P)
    if it % 20 and it < 125:
        continue
    print("iteration", it, "P finite:", bool(torch.isfinite(P).all().item()), "min T", float(P[:, 1].min()), "min p", float(P[:, 0].min()))
    res = {}
    for var in (0, 512):
        _lib.call("ibh_set_tuning", b"quad_variant", var)
        R = ibamd.residual_euler_hll(dpart, P)
        res[var] = R.clone()
        bad = ~torch.isfinite(R).all(dim=1)
        print("  variant", var, "non-finite cells:", int(bad.sum().item()))
        if bad.any():
            ids = torch.nonzero(bad)[:12, 0].cpu().numpy()
            for c in ids:
                p = int(c) % 512
                print("    cell", int(c), "block", int(c) // 512, "xyz", (p & 7, (p >> 3) & 7, p >> 6), "P", P[int(c)].cpu().numpy(),
                      "R", R[int(c)].cpu().numpy())
    _lib.call("ibh_set_tuning", b"quad_variant", 0)
    ok = torch.isfinite(res[0]).all(dim=1) & torch.isfinite(res[512]).all(dim=1)
    d = (res[0][ok] - res[512][ok]).abs().max(dim=0).values / res[512][ok].abs().max(dim=0).values
    print("  max rel diff default vs thread-per-cell on the finite cells:", d.cpu().numpy())
