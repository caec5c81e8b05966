"""Multi-process rehearsal of distributed.py on ONE GPU (every rank on cuda:0, gloo handshake, staged exchange):
`impose_bc!` with FlowBC closures on the ghosts a rank owns (donor cells beyond the skirt in the halo lists), the
image-only Euler sweep, a fixed-point update and the all-reduced residual norm -- against the same two iterations on the
one-partition domain (rank 0, same GPU).  Started by tests/test_gpu_distributed.py under torch.distributed.run."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ibamd  # noqa: E402
from conftest import RAE_FAMILIES, rae_mesh  # noqa: E402
from ibamd import cfd as gcfd  # noqa: E402
from ibamd.distributed import LocalDomain, Reductions, bc_donor_extras  # noqa: E402
from ibamd.halo import HaloExchange, HaloPlan  # noqa: E402

f32 = np.float32
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
msh = rae_mesh()
n = len(msh)
mps = -(-(-(-n // world)) // 64) * 64
far = [1.0e5, 288.15, 230.0, 10.0]
far_bc = gcfd.FlowBC(gcfd.Fluid(), far)
wall_bc = gcfd.FlowBC(gcfd.Fluid(), [1.0e5, 288.15, 0.0], normal_flow=True)
OMEGA, ITERS = 1e-7, 2


def field(ncells):
    rng = np.random.default_rng(8)
    P = np.empty((ncells, 4), dtype=f32)
    P[:, 0] = 1e5 * (1 + 0.02 * rng.uniform(-1, 1, ncells))
    P[:, 1] = 288.15 * (1 + 0.02 * rng.uniform(-1, 1, ncells))
    P[:, 2] = 230.0 * (1 + 0.05 * rng.uniform(-1, 1, ncells))
    P[:, 3] = 20.0 * rng.uniform(-1, 1, ncells)
    return P


def bcs(domlike, P):
    ibamd.impose_bc(lambda b, ia: far_bc(ia, b.normals), domlike, "farfield", P)
    ibamd.impose_bc(lambda b, ia: wall_bc(ia, b.normals), domlike, "wall", P)


Pg = field(n)
# ---- this rank's share
dom = ibamd.Domain(msh, max_partition_size=mps, hypercube_families=RAE_FAMILIES, only=[rank + 1])
part = dom.partitions[rank + 1]
extras = bc_donor_extras(dom)
ldom = LocalDomain(dom, rank + 1, extras)
plan = HaloPlan(dom, rank + 1, extra=extras)
hx = HaloExchange(plan, "cuda")
dpart = ibamd.to_backend(part, ibamd.hip)
red = Reductions(part.image_in_domain, device="cuda")
nc, nrows = dpart.nc, ldom.n_rows
gids = np.concatenate([part.domain, extras[rank + 1]]).astype(np.int64)
own = np.zeros(nrows, dtype=bool)
own[part.image_in_domain] = True
Ploc = Pg[gids].copy()
Ploc[~own] = np.nan                                   # skirt and donor rows start stale
P = ibamd.hip(Ploc)
R = ibamd.colmajor_empty(nc, 4)
img = torch.from_numpy(part.image_in_domain).long().cuda()
norms = []
for _ in range(ITERS):
    hx.exchange(P)                                    # skirt + donor cells
    bcs(ldom, P)                                      # the ghosts this rank owns, local rows
    hx.exchange(P)                                    # ghosts in the skirt were updated by their owners
    ibamd.residual_euler_hll(dpart, P[:nc], out=R, flags=ibamd.IBH_IMAGE_ONLY if dpart.info["image_blocks_all_eligible"] else 0)
    P[:nc][img] += OMEGA * R[img]
    norms.append(red.norm(R))
mine = torch.cat([torch.from_numpy(gids[part.image_in_domain]).double()[:, None], P[:nc][img].double().cpu()], dim=1)
gathered = [None] * world if rank == 0 else None
dist.gather_object(mine.numpy(), gathered, dst=0)
ok = True
if rank == 0:
    # ---- the same on the one-partition domain
    dom1 = ibamd.Domain(msh, max_partition_size=10 ** 9, hypercube_families=RAE_FAMILIES)
    (p1,) = dom1.partitions.values()
    d1 = ibamd.to_backend(p1, ibamd.hip)
    P1 = ibamd.hip(Pg)
    R1 = ibamd.colmajor_empty(n, 4)
    norms1 = []
    for _ in range(ITERS):
        bcs(dom1, P1)
        ibamd.residual_euler_hll(d1, P1, out=R1)
        P1 += OMEGA * R1
        norms1.append(float(torch.linalg.norm(R1.double())))
    ref = P1.cpu().numpy()
    got = np.full((n, 4), np.nan)
    for g in gathered:
        got[g[:, 0].astype(np.int64)] = g[:, 1:]
    scale = np.abs(ref).max(axis=0)
    err = np.abs(got - ref).max(axis=0) / scale
    nerr = max(abs(a - b) / b for a, b in zip(norms, norms1))
    ok = bool(np.isfinite(got).all() and (err <= 1e-5).all() and nerr <= 1e-5)
    print(f"ranks {world}, donor cells beyond the skirt {sum(int(v.size) for v in extras.values())}, "
          f"max rel err per variable {err}, norm err {nerr:.2e}", flush=True)
    print("distributed BC + Euler sweep + update + all-reduced norm match the one-partition run:", ok, flush=True)
t = torch.tensor([int(ok)])
dist.all_reduce(t, op=dist.ReduceOp.MIN)
dist.destroy_process_group()
sys.exit(0 if t.item() == 1 else 1)
