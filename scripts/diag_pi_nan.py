"""Where the NaN of the config-4 point-implicit line comes from (run on the GPU box): python scripts/diag_pi_nan.py [workload]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
import ibamd  # noqa: E402
from ibamd import cfd as gcfd  # noqa: E402
from ibamd import point_implicit as pi  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "sphere3d_8M"
msh = bench.build_mesh(name)
fam = [("farfield", [(d, s) for d in (1, 2, 3) for s in (False, True)])]
dom = ibamd.Domain(msh, max_partition_size=10 ** 9, hypercube_families=fam)
(part,) = dom.partitions.values()
dpart = ibamd.to_backend(part, ibamd.hip)
n = part.centers.shape[0]
rng = np.random.default_rng(12345)
P_h = np.empty((n, 5), dtype=np.float32)
P_h[:, 0] = 1e5 * (1 + 0.02 * rng.uniform(-1, 1, n))
P_h[:, 1] = 288.15 * (1 + 0.02 * rng.uniform(-1, 1, n))
P_h[:, 2] = 100.0 * (1 + 0.1 * rng.uniform(-1, 1, n))
P_h[:, 3:] = 10.0 * rng.uniform(-1, 1, (n, 2))
P = ibamd.hip(P_h)
far = gcfd.FlowBC(gcfd.Fluid(), [1.0e5, 288.15, 100.0, 0.0, 0.0])
wall = gcfd.FlowBC(gcfd.Fluid(), [1.0e5, 288.15, 0.0], normal_flow=True)
ibamd.impose_bc(lambda b, ia: far(ia, b.normals), dom, "farfield", P)
ibamd.impose_bc(lambda b, ia: wall(ia, b.normals), dom, "sphere", P)
print("P finite", bool(torch.isfinite(P).all()), "T min", float(P[:, 1].min()), "p min", float(P[:, 0].min()))
P0 = P.clone()
dt = 1e-5


def f(X):
    return (X - P0) / dt - ibamd.residual_euler_hll(dpart, X)


fx = f(P)
print("f(P) finite", bool(torch.isfinite(fx).all()), "nonfinite rows", int((~torch.isfinite(fx)).any(dim=1).sum()))
D = pi.hutchinson_trick(f, P, 1, h=1e-2, seed=1)
print("D finite", bool(torch.isfinite(D).all()), "nonfinite points", int((~torch.isfinite(D.reshape(n, -1))).any(dim=1).sum()))
D2 = D.clone()
pi._inverse_blocks(D2)
bad = (~torch.isfinite(D2.reshape(n, -1))).any(dim=1)
print("inverse finite", bool(torch.isfinite(D2).all()), "nonfinite points", int(bad.sum()))
if bad.any():
    i = int(torch.nonzero(bad)[0])
    print("first bad block", i, D[i].cpu().numpy(), "P", P[i].cpu().numpy())
lin, bb, prec = pi.linearize(f, P, 1, h=1e-2, seed=1)
print("b finite", bool(torch.isfinite(bb).all()))
s = prec(bb)
print("prec(b) finite", bool(torch.isfinite(s).all()), "max", float(s.abs().max()))
As = lin(s)
print("A s finite", bool(torch.isfinite(As).all()), "max", float(As.abs().max()), "nonfinite rows",
      int((~torch.isfinite(As)).any(dim=1).sum()))
if not torch.isfinite(As).all():
    i = int(torch.nonzero((~torch.isfinite(As)).any(dim=1))[0])
    X = P + 1e-2 * s
    print("row", i, "s", s[i].cpu().numpy(), "P", P[i].cpu().numpy(), "X", X[i].cpu().numpy())
x, ratio = pi.solve(lin, bb, prec, n_iter=1, rtol=1e-9)
print("ratio", ratio)
