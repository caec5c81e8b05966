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

# ---- the fused step (exchange + image-only quad sweep in ONE launch) on partitions of the RAE2822 mesh: against
# exchange-then-sweep, eagerly and inside a HIP graph, the skirt poisoned before every step
import bench
msh2 = bench.build_mesh("rae2822_37k")
n2 = len(msh2)
mps2 = -(-(-(-n2 // world)) // 64) * 64
dom2 = ibamd.Domain(msh2, max_partition_size=mps2, boundaries=False, only=[rank + 1])
part2 = dom2.partitions[rank + 1]
plan2 = HaloPlan(dom2, rank + 1)
dp2 = ibamd.to_backend(part2, ibamd.hip)
assert dp2.info["image_blocks_all_eligible"] and dp2.info["image_quads"] > 0
# the field every rank holds after an exchange: a function of the cell centres
xc = part2.centers.astype(np.float64)
uh = (np.sin(2 * np.pi * xc[:, 0]) * np.cos(2 * np.pi * xc[:, 1]) + 0.1 * np.sin(37.0 * xc[:, 0] + 11.0 * xc[:, 1])).astype(np.float32)
Ch = np.stack([np.ones(uh.size, np.float32), (0.5 + 0.2 * np.cos(5.0 * xc[:, 0])).astype(np.float32)], axis=1)
u_true2, C2 = ibamd.hip(uh), ibamd.hip(Ch)
skirt2 = np.ones(dp2.nc, bool); skirt2[part2.image_in_domain] = False
sk2 = torch.from_numpy(skirt2).cuda()
img2 = torch.from_numpy(part2.image_in_domain).long().cuda()
xg2 = XgmiHalo(plan2, dom2, "cuda", nv=1)
ref2 = torch.full((dp2.nc,), float("nan"), device="cuda")
u2 = u_true2.clone(); u2[sk2] = float("nan")
xg2.exchange(u2)
assert bool(torch.equal(u2, u_true2)), "synthetic field is not a function of the cell centres"
ibamd.residual_advection(dp2, u2, C2, out=ref2, flags=ibamd.IBH_IMAGE_ONLY)
ok2 = True
for _ in range(3):                                   # eager
    u2[sk2] = float("nan")
    out2 = torch.full((dp2.nc,), float("nan"), device="cuda")
    torch.cuda.synchronize(); dist.barrier()
    xg2.fused_step(dp2, u2, C2, out2)
    torch.cuda.synchronize()
    ok2 = ok2 and bool(torch.equal(out2[img2], ref2[img2])) and bool(torch.equal(u2, u_true2))
side2 = torch.cuda.Stream()
with torch.cuda.stream(side2):                        # captured: 4 steps per graph, 3 replays
    out3 = torch.full((dp2.nc,), float("nan"), device="cuda")
    xg2.fused_step(dp2, u2, C2, out3)
    torch.cuda.synchronize(); dist.barrier()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=side2):
        for _ in range(4):
            xg2.fused_step(dp2, u2, C2, out3)
    for _ in range(3):
        u2[sk2] = float("nan")
        out3.fill_(float("nan"))
        torch.cuda.synchronize(); dist.barrier()
        g2.replay()
        torch.cuda.synchronize()
        ok2 = ok2 and bool(torch.equal(out3[img2], ref2[img2])) and bool(torch.equal(u2, u_true2))
ok2 = ok2 and xg2.healthy()
t2 = torch.tensor([int(ok2)]); dist.all_reduce(t2, op=dist.ReduceOp.MIN)
if rank == 0:
    print("fused exchange + sweep step matches exchange-then-sweep:", bool(t2.item()), flush=True)
assert t2.item() == 1
xg2.close()
dist.destroy_process_group()
