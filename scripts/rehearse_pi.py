"""Two ranks on the one GPU of the test box (gloo process group, device arrays): the device-resident point-implicit smoother
across ranks (point_implicit.linearize / solve with distributed.RankOps: exchange before every residual sweep, zeros outside
the owned rows, all-reduced dot products / norm / max) against the one-partition device smoother of the same closure with the
same +-1 samples, on the cells each rank owns.  Run: python -m torch.distributed.run --nproc-per-node 2 scripts/rehearse_pi.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import ibamd  # noqa: E402
from conftest import advection_mesh, seeded_field  # noqa: E402
from ibamd import point_implicit as pi  # noqa: E402
from ibamd.distributed import RankOps  # noqa: E402
from ibamd.halo import HaloExchange, HaloPlan  # noqa: E402
from ibamd.hiparray import HipArray  # noqa: E402

f32 = np.float32
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
msh = advection_mesh(2e-2)
H, DT, N_SAMPLES, N_ITER = 1e-3, 2e-5, 2, 2


def residual(dpart, Xref):
    """pseudo-time residual of a nonlinear two-component problem: (X - Xref) / dt - [Laplacian(X) - 50 X^3]"""
    def f(X):
        r = torch.zeros_like(X)
        for dim in (1, 2):
            r += ibamd.green_gauss(dpart, ibamd.face_gradient(dpart, X, dim), dim)
        x, ref = HipArray(X), HipArray(Xref)
        return ((x - ref) / DT - HipArray(r) + x * x * x * 50.0).t
    return f


def samples(n_global):
    rng = np.random.default_rng(11)
    return [[rng.choice(f32([-1, 1]), n_global).astype(f32) for _ in range(N_SAMPLES)] for _ in range(2)]


# ---- across ranks
npb = msh.block_size ** msh.ndims
mps = -(-(-(-len(msh) // world)) // npb) * npb
dom = ibamd.Domain(msh, max_partition_size=mps, boundaries=False, only=[rank + 1])
part = dom.partitions[rank + 1]
dpart = ibamd.to_backend(part, ibamd.hip)
hx = HaloExchange(HaloPlan(dom, rank + 1), "cuda")
gids = np.asarray(part.domain, dtype=np.int64)
ops = RankOps(part.image_in_domain, gids.size, hx.exchange, device="cuda")
Xg = seeded_field(dom.global_centers(), nv=2)
X0 = Xg[gids].copy()
X0[ops.not_own_np] = np.nan                    # stale skirt rows: every evaluation must refresh them
f = ops.closure(residual(dpart, ibamd.hip(Xg[gids])))
samp = [[ibamd.hip(z[gids]) for z in col] for col in samples(Xg.shape[0])]
lin, b, prec = pi.linearize(f, ibamd.hip(X0), N_SAMPLES, h=H, samples=samp)
x, ratio = pi.solve(lin, b, prec, n_iter=N_ITER, rtol=0.0, atol=0.0, reduce=ops)
# ---- one partition, same device
dom1 = ibamd.Domain(msh, max_partition_size=10 ** 9, boundaries=False)
d1 = ibamd.to_backend(dom1.partitions[1], ibamd.hip)
samp1 = [[ibamd.hip(z) for z in col] for col in samples(Xg.shape[0])]
lin1, b1, prec1 = pi.linearize(residual(d1, ibamd.hip(Xg)), ibamd.hip(Xg), N_SAMPLES, h=H, samples=samp1)
x1, ratio1 = pi.solve(lin1, b1, prec1, n_iter=N_ITER, rtol=0.0, atol=0.0)
img = np.asarray(part.image_in_domain)
same_b = np.array_equal(ibamd.to_host(b)[img], ibamd.to_host(b1)[gids[img]])
same_D = np.array_equal(ibamd.to_host(prec.inverse_diagonal)[img], ibamd.to_host(prec1.inverse_diagonal)[gids[img]])
zeros_outside = not ibamd.to_host(b)[ops.not_own_np].any()
got, ref = ibamd.to_host(x)[img], ibamd.to_host(x1)[gids[img]]
err = float(np.abs(got - ref).max() / np.abs(ibamd.to_host(x1)).max())
ok = torch.tensor([int(same_b and same_D and zeros_outside and err <= 1e-5 and ratio1 < 0.5
                       and abs(ratio - ratio1) <= 1e-5 * max(1.0, ratio1) and np.isfinite(got).all())])
dist.all_reduce(ok, op=dist.ReduceOp.MIN)
if rank == 0:
    print(f"distributed device point-implicit smoother == one-partition device smoother on the owned cells: {bool(ok.item())} "
          f"(rank 0: b bit-identical {same_b}, inverse blocks bit-identical {same_D}, x max rel diff {err:.2e}, "
          f"ratios {ratio:.6f} / {ratio1:.6f})")
dist.destroy_process_group()
