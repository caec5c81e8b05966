"""Multi-process rehearsal of the xGMI halo exchange on ONE GPU (every rank on cuda:0, gloo handshake):
checks XgmiHalo against the reference exchange (HaloExchange) and a graph-captured overlapped sweep."""
import os, sys
import numpy as np, torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ibamd
from conftest import ADV_FAMILIES, advection_mesh, seeded_field
from ibamd.halo import HaloExchange, HaloPlan, XgmiHalo, verify_exchangers, sweep_overlapped

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
msh = advection_mesh(1e-2)
n = len(msh)
mps = -(-(-(-n // world)) // 64) * 64
dom = ibamd.Domain(msh, max_partition_size=mps, boundaries=False, only=[rank + 1])
part = dom.partitions[rank + 1]
plan = HaloPlan(dom, rank + 1)
ref = HaloExchange(plan, "cuda")
for nv in (1, 3):
    xg = XgmiHalo(plan, dom, "cuda", nv=nv)
    ok = verify_exchangers(xg, ref, part.spacing.shape[0], nv, rounds=4)
    healthy = xg.healthy()
    if rank == 0:
        print(f"nv={nv}: xgmi == reference exchange: {ok}, healthy: {healthy}", flush=True)
    assert ok and healthy
    if nv == 3:
        xg.close()
# overlapped sweep with the xGMI exchange inside a HIP graph vs eager reference
dpart = ibamd.to_backend(part, ibamd.hip)
g = seeded_field(dom.global_centers())
u_true = ibamd.hip(np.array(g[part.domain]))
C = ibamd.hip(np.ones((dpart.nc, 2), dtype=np.float32))
ud_ref = ibamd.residual_advection(dpart, u_true, C).clone()
skirt = np.ones(dpart.nc, bool); skirt[part.image_in_domain] = False
u = u_true.clone(); u[torch.from_numpy(skirt).cuda()] = float("nan")
ud = torch.zeros_like(u)
xg1 = XgmiHalo(plan, dom, "cuda", nv=1)
comm = torch.cuda.Stream(); side = torch.cuda.Stream()
with torch.cuda.stream(side):
    sweep_overlapped(xg1, dpart, u, C, ud, comm); sweep_overlapped(xg1, dpart, u, C, ud, comm)
    torch.cuda.synchronize(); dist.barrier()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        for _ in range(4):
            sweep_overlapped(xg1, dpart, u, C, ud, comm)
    u[torch.from_numpy(skirt).cuda()] = float("nan")
    for _ in range(5):
        graph.replay()
    torch.cuda.synchronize()
img = torch.from_numpy(part.image_in_domain).long().cuda()
same = bool(torch.equal(ud[img], ud_ref[img])) and xg1.healthy()
t = torch.tensor([int(same)]); dist.all_reduce(t, op=dist.ReduceOp.MIN)
if rank == 0:
    print("graph-captured overlapped sweeps with xGMI exchange match:", bool(t.item()), flush=True)
assert t.item() == 1
xg1.close()
dist.destroy_process_group()
