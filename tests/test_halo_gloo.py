"""N>1 path on CPU: world_size-2 (and 3) gloo runs of the skirt-cell halo exchange.

Semantics under test = the reference's per-call gather (ImmersedBoundary.jl:836-841): after an
exchange every rank's local array equals global[part.domain].  Pure data movement, checked
bit-exactly.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import ibamd
from conftest import ADV_FAMILIES, advection_mesh, seeded_field
from ibamd.halo import HaloExchange, HaloPlan


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nv, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        msh = advection_mesh(2e-2)
        ncells = len(msh)
        mps = -(-ncells // world)
        mps = -(-mps // 64) * 64
        dom = ibamd.Domain(msh, max_partition_size=mps, hypercube_families=ADV_FAMILIES, boundaries=False,
                           only=[rank + 1])
        assert len(dom.images) == world
        part = dom.partitions[rank + 1]
        g = seeded_field(dom.global_centers(), nv=nv)
        local = np.array(g[part.domain])
        skirt = np.ones(local.shape[0], dtype=bool)
        skirt[part.image_in_domain] = False
        assert skirt.sum() > 0
        local[skirt] = np.nan  # stale halo
        t = torch.from_numpy(local)
        plan = HaloPlan(dom, rank + 1)
        assert plan.n_recv == int(skirt.sum())
        hx = HaloExchange(plan, "cpu")
        hx.exchange(t)
        ok = np.array_equal(t.numpy(), g[part.domain])
        # second exchange after the owners changed their values
        g2 = g * np.float32(2.0)
        t2 = torch.from_numpy(np.array(g[part.domain]))
        t2[torch.from_numpy(part.image_in_domain).long()] *= 2.0
        hx.exchange(t2)
        ok2 = np.array_equal(t2.numpy(), g2[part.domain])
        res = torch.tensor([int(ok and ok2)], dtype=torch.int32)
        dist.all_reduce(res, op=dist.ReduceOp.MIN)
        if rank == 0:
            out.put(int(res.item()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,nv", [(2, None), (2, 3), (3, None)])
def test_halo_exchange_gloo(world, nv):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nv, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    assert q.get(timeout=10) == 1


def test_halo_plan_lists_are_mirror_images():
    """send list of r towards q and recv list of q from r name the same global cells in the same order."""
    msh = advection_mesh(2e-2)
    ncells = len(msh)
    world = 4
    mps = -(-(-(-ncells // world)) // 64) * 64
    dom = ibamd.Domain(msh, max_partition_size=mps, boundaries=False)
    plans = {p: HaloPlan(dom, p) for p in dom.partitions}
    for r, pr in plans.items():
        dr = dom.partitions[r].domain
        for q, sidx in pr.send.items():
            dq = dom.partitions[q].domain
            assert np.array_equal(dr[sidx], dq[plans[q].recv[r]])
        # every skirt cell is received exactly once
        nskirt = dr.size - dom.partitions[r].image.size
        assert pr.n_recv == nskirt


def _worker_config3(rank, world, port, out):
    """BASELINE.json configs[2]: the 3.47 M-cell RAE2822 mesh in 8 block-aligned partitions (skirt growth of
    ImmersedBoundary.jl:594-621); one full exchange, every local array must equal global[part.domain]."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        msh = bench.build_mesh("rae2822_3.47M")
        ncells = len(msh)
        mps = -(-(-(-ncells // world)) // 64) * 64
        dom = ibamd.Domain(msh, max_partition_size=mps, boundaries=False, only=[rank + 1])
        part = dom.partitions[rank + 1]
        # a field every rank can evaluate for any global cell without building the other partitions: a hash of the id
        gid = np.asarray(part.domain, dtype=np.int64)
        value = lambda g: ((g * 2654435761) % 1000003).astype(np.float32) / np.float32(1000003.0)
        local = value(gid)
        expect = local.copy()
        skirt = np.ones(local.shape[0], dtype=bool)
        skirt[part.image_in_domain] = False
        local[skirt] = np.nan
        plan = HaloPlan(dom, rank + 1)
        hx = HaloExchange(plan, "cpu")
        t = torch.from_numpy(local)
        hx.exchange(t)
        ok = np.array_equal(t.numpy(), expect) and plan.n_recv == int(skirt.sum()) and part.image.size % 64 == 0
        res = torch.tensor([int(ok), int(skirt.sum())], dtype=torch.int64)
        dist.all_reduce(res, op=dist.ReduceOp.MIN)
        if rank == 0:
            out.put((int(res[0].item()), ncells, int(res[1].item())))
    finally:
        dist.destroy_process_group()


def test_halo_exchange_gloo_config3_8_ranks():
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_config3, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    ok, ncells, min_skirt = q.get(timeout=10)
    assert ok == 1 and ncells == 3469888 and min_skirt > 0
