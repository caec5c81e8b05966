"""Two ranks on the one GPU of the test box (gloo process group, device arrays): the device-resident FAS! across ranks
(distributed.RankLevels + solver.FAS with the exchange / norm hooks) against the one-partition device V-cycle of the same
closure, on the cells each rank owns.  Run: python -m torch.distributed.run --nproc-per-node 2 scripts/rehearse_fas.py"""
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
from ibamd.distributed import RankLevels, Reductions  # noqa: E402
from ibamd.halo import HaloExchange  # noqa: E402

f32 = np.float32
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
msh = advection_mesh(2e-2)
MAXLEV, N_ITER = 2, 3


def closure(dpart, hmin):
    def f(Q):
        r = torch.zeros_like(Q)
        for dim in (1, 2):
            r += ibamd.green_gauss(dpart, ibamd.face_gradient(dpart, Q, dim), dim)
        return r, f32(0.2) * f32(hmin) * f32(hmin)
    return f


# ---- across ranks
lv = RankLevels(msh, rank + 1, world, MAXLEV, domain_kwargs=dict(boundaries=False))
dparts = [ibamd.to_backend(p, ibamd.hip) for p in lv.parts]
fs = [closure(dp, float(p.spacing[:, 0].min())) for dp, p in zip(dparts, lv.parts)]
hxs = [HaloExchange(pl, "cuda") for pl in lv.plans]
reds = [Reductions(p.image_in_domain, device="cuda") for p in lv.parts]
ncs = [int(p.domain.size) for p in lv.parts]


def f_rank(l, Q):
    if Q.shape[0] == ncs[l]:
        return fs[l](Q)
    r = ibamd.colmajor_empty(Q.shape[0], Q.shape[1])
    r[ncs[l]:] = 0.0
    rr, om = fs[l](Q[:ncs[l]])
    r[:ncs[l]] = rr
    return r, om


Qg = seeded_field(lv.doms[0].global_centers(), nv=2)
gids = np.concatenate([lv.parts[0].domain, lv.extras[0][rank + 1]]).astype(np.int64)
Ql = Qg[gids].copy()
own = np.zeros(gids.size, dtype=bool)
own[lv.parts[0].image_in_domain] = True
Ql[~own] = np.nan
Q = ibamd.hip(Ql)
ratio = ibamd.FAS(f_rank, Q, coarseners=lv.coarseners, prolongators=lv.prolongators, n_iter=N_ITER, rtol=0.0, atol=0.0,
                  exchange=lambda l, q: hxs[l].exchange(q), level_norm=lambda l, r: reds[l].norm(r[:ncs[l]]))
# ---- one partition, same device
dom1 = ibamd.Domain(msh, max_partition_size=10 ** 9, boundaries=False)
cds, prol, coar = ibamd.multigrid(dom1, max_levels=MAXLEV)
d1 = [ibamd.to_backend(d.partitions[1], ibamd.hip) for d in [dom1] + cds]
f1 = [closure(dp, float(d.partitions[1].spacing[:, 0].min())) for dp, d in zip(d1, [dom1] + cds)]
Q1 = ibamd.hip(Qg)
ratio1 = ibamd.FAS(lambda l, q: f1[l](q), Q1, coarseners=coar, prolongators=prol, n_iter=N_ITER, rtol=0.0, atol=0.0)
img = lv.parts[0].image_in_domain
got = ibamd.to_host(Q)[img]
ref = ibamd.to_host(Q1)[gids[img]]
err = float(np.abs(got - ref).max() / np.abs(ref).max())
ok = torch.tensor([int(err <= 1e-6 and abs(ratio - ratio1) <= 1e-5 * max(1.0, ratio1) and not np.isnan(got).any())])
dist.all_reduce(ok, op=dist.ReduceOp.MIN)
if rank == 0:
    print(f"distributed device V-cycle == one-partition device V-cycle on the owned cells: {bool(ok.item())} "
          f"(rank 0: max rel diff {err:.2e}, ratios {ratio:.6f} / {ratio1:.6f})")
dist.destroy_process_group()
