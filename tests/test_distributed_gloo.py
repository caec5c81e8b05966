"""Distributed `impose_bc!` + `FAS!` prerequisites on CPU, world size 2 (gloo): BC donor cells in the halo lists,
boundary chunks re-indexed to a rank's local rows, all-reduced norm and minimum (SURVEY.md 8e / H6;
/root/reference/src/ImmersedBoundary.jl:1228-1245, src/solver.jl:57,84, test/advection.jl:59).

The index bookkeeping under test is the product's (immersedboundary.jl_amd/distributed.py, halo.HaloPlan); the
arithmetic on each rank is the ORACLE's numpy restatement, so the two-rank run must reproduce the one-partition oracle
run: bit for bit for the BC, to rounding for the norm."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import ibamd
from conftest import ADV_FAMILIES, RAE_FAMILIES, advection_mesh, oracle_view, rae_mesh, seeded_field
from ibamd.distributed import LocalDomain, Reductions, bc_donor_extras
from ibamd.halo import HaloExchange, HaloPlan

f32 = np.float32


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bcs(od, view, u):
    """apply_bcs! of test/advection.jl:33-46"""
    od.impose_bc(lambda b, ui: f32(1.0), view, "upper", u)
    od.impose_bc(lambda b, ui: f32(0.0), view, "lower", u)
    od.impose_bc(lambda b, ui: ui.copy(), view, "outlet", u)


def _closure(od, part, u, C):
    ud = np.zeros_like(u)
    D = od.JST_sensor(part, u)
    for dim in (1, 2):
        Cf = od.at_faces(part, np.ascontiguousarray(C[:, dim - 1]), dim)
        gu = od.cell_gradient(part, u, dim)
        uL, uR = od.MUSCL(part, u, gu, dim, D=D, high_order=True)
        ud -= od.green_gauss(part, (uL + uR) * Cf / f32(2) + np.abs(Cf) * (uL - uR) / f32(2), dim)
    return ud


def _reference_run(dom, u0, n_iter, omega):
    """One partition, global arrays: BC then `n_iter` FAS fixed-point iterations (solver.jl:76-88 without coarsening)."""
    from conftest import oracle_boundaries_view
    from oracle import domain as od
    (part,) = dom.partitions.values()
    opart, view = oracle_view(part), oracle_boundaries_view(dom)
    C = np.ones((u0.size, 2), dtype=f32)
    u = u0.copy()
    norms = []
    for _ in range(n_iter):
        _bcs(od, view, u)
        r = _closure(od, opart, u, C)
        u += f32(omega) * r
        norms.append(float(np.linalg.norm(r.astype(np.float64))))
    return u, norms


def _worker(rank, world, port, out):
    from conftest import oracle_boundaries_view
    from oracle import domain as od
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        msh = advection_mesh(2e-2)
        ncells = len(msh)
        mps = -(-(-(-ncells // world)) // 64) * 64
        dom = ibamd.Domain(msh, max_partition_size=mps, hypercube_families=ADV_FAMILIES, only=[rank + 1])
        part = dom.partitions[rank + 1]
        extras = bc_donor_extras(dom)
        ldom = LocalDomain(dom, rank + 1, extras)
        plan = HaloPlan(dom, rank + 1, extra=extras)
        hx = HaloExchange(plan, "cpu")
        red = Reductions(part.image_in_domain)
        nc, nrows = part.spacing.shape[0], ldom.n_rows
        gids = np.concatenate([part.domain, extras[rank + 1]]).astype(np.int64)
        u0g = seeded_field(dom.global_centers(), kind="step")
        u = u0g[gids].copy()                                   # extended local array: domain rows, then donor extras
        view = oracle_boundaries_view(ldom)
        opart = oracle_view(part)
        C = np.ones((nc, 2), dtype=f32)
        img = part.image_in_domain
        omega, norms = 2e-3, []
        for _ in range(3):
            t = torch.from_numpy(u)
            hx.exchange(t)                                     # skirt + donor cells from their owners
            _bcs(od, view, u)                                  # owned ghosts, local indices
            t = torch.from_numpy(u)
            hx.exchange(t)                                     # ghosts of the skirt were updated by their owners
            r = _closure(od, opart, u[:nc], C)
            u[:nc][img] += f32(omega) * r[img]
            norms.append(red.norm(r))
        dt = red.minimum(1.0 + rank)
        out.put((rank, gids[img], u[:nc][img].copy(), norms, dt, int(extras[rank + 1].size), nrows - nc))
    finally:
        dist.destroy_process_group()


def test_two_rank_bc_and_fas_match_the_one_partition_run():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    msh = advection_mesh(2e-2)
    dom1 = ibamd.Domain(msh, max_partition_size=10 ** 9, hypercube_families=ADV_FAMILIES)
    u0 = seeded_field(dom1.global_centers(), kind="step")
    u_ref, norms_ref = _reference_run(dom1, u0, 3, 2e-3)
    got = np.full(len(dom1), np.nan, dtype=f32)
    n_extra = 0
    for rank, gid, vals, norms, dt, nex, nex2 in res:
        got[gid] = vals
        assert nex == nex2
        n_extra += nex
        assert dt == 1.0                                        # the all-reduced minimum of 1 + rank
        assert np.allclose(norms, norms_ref, rtol=1e-6)         # all-reduced ||r|| = the global norm
    assert not np.isnan(got).any()
    assert np.array_equal(got, u_ref)                           # same arithmetic on the same values: bit for bit


def _worker_rae(rank, world, port, out):
    """RAE2822 case in 4 partitions at the reference's skirt depth: some image points of the wall ghosts have donor
    cells beyond the skirt (bc_donor_extras is not empty).  One exchange + `impose_bc!` on a 2-column field."""
    from conftest import oracle_boundaries_view
    from oracle import domain as od
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        msh = rae_mesh()
        ncells = len(msh)
        mps = -(-(-(-ncells // world)) // 64) * 64
        dom = ibamd.Domain(msh, max_partition_size=mps, hypercube_families=RAE_FAMILIES, only=[rank + 1])
        part = dom.partitions[rank + 1]
        extras = bc_donor_extras(dom)
        ldom = LocalDomain(dom, rank + 1, extras)
        plan = HaloPlan(dom, rank + 1, extra=extras)
        assert plan.n_extra == extras[rank + 1].size
        hx = HaloExchange(plan, "cpu")
        gids = np.concatenate([part.domain, extras[rank + 1]]).astype(np.int64)
        Ug = seeded_field(dom.global_centers(), nv=2)
        own = np.zeros(gids.size, dtype=bool)
        own[part.image_in_domain] = True
        U = Ug[gids].copy()
        U[~own] = np.nan                                        # stale skirt and donor rows
        t = torch.from_numpy(U)
        hx.exchange(t)
        assert np.array_equal(U, Ug[gids])                      # skirt AND donor extras arrived
        view = oracle_boundaries_view(ldom)
        od.impose_bc(lambda b, ia: ia * f32(0.5), view, "wall", U)
        od.impose_bc(lambda b, ia: f32(1.0), view, "farfield", U)
        img = part.image_in_domain
        out.put((rank, gids[img], U[img].copy(), int(extras[rank + 1].size)))
    finally:
        dist.destroy_process_group()


def test_four_rank_bc_with_donor_cells_beyond_the_skirt():
    from conftest import oracle_boundaries_view
    from oracle import domain as od
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_rae, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    msh = rae_mesh()
    dom1 = ibamd.Domain(msh, max_partition_size=10 ** 9, hypercube_families=RAE_FAMILIES)
    U = seeded_field(dom1.global_centers(), nv=2)
    view = oracle_boundaries_view(dom1)
    od.impose_bc(lambda b, ia: ia * f32(0.5), view, "wall", U)
    od.impose_bc(lambda b, ia: f32(1.0), view, "farfield", U)
    got = np.full(U.shape, np.nan, dtype=f32)
    n_extra = 0
    for rank, gid, vals, nex in res:
        got[gid] = vals
        n_extra += nex
    assert n_extra > 0                                          # donor cells beyond the skirt do occur here
    assert np.array_equal(got, U)
