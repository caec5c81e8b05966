"""`FAS!` (src/solver.jl:39-91) over `multigrid(dom)` (src/ImmersedBoundary.jl:1355-1407) ACROSS RANKS, rehearsed on CPU
with gloo (world 2): every rank holds its partition of every level (distributed.RankLevels: same blocks of the shared block
tree on every level), the transfer operators restricted to the rows it owns and re-indexed to local rows, a halo exchange
per level, norms all-reduced over the owned cells.  Arithmetic = the oracle's on both sides (numpy operators, the oracle's
FAS loop with the exchange / norm hooks); index bookkeeping = the product's.  The two-rank V-cycle must reproduce the
one-partition V-cycle on every owned cell BIT FOR BIT (same stencils, same weights, same order), and the all-reduced norms
the global ones."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import ibamd
from conftest import advection_mesh, oracle_view, seeded_field
from ibamd.distributed import RankLevels, Reductions
from ibamd.halo import HaloExchange

f32 = np.float32
N_ITER, MAXLEV = 3, 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oacc(acc):
    from oracle.accumulator import Accumulator as OAcc
    o = object.__new__(OAcc)
    o.n_output, o.first_index, o.stencils = acc.n_output, True, acc.stencils
    return o


def _residual(od, opart, Q):
    """pseudo-time relaxation of the Laplacian (test/dissipation.jl:69-77) on the first nc rows of a local array"""
    r = np.zeros_like(Q)
    for dim in (1, 2):
        r += od.green_gauss(opart, od.face_gradient(opart, Q, dim), dim)
    h = opart.spacing[:, 0].min()
    return r, f32(0.2) * h * h


def _bcs(od, view, Q):
    """the boundary conditions of test/dissipation.jl:30-52 on a level's own boundaries (two-column field)"""
    od.impose_bc(lambda b, a: np.broadcast_to(f32([1.0, 0.0]), a.shape).copy(), view, "upper", Q)
    od.impose_bc(lambda b, a: np.broadcast_to(f32([0.0, 1.0]), a.shape).copy(), view, "lower", Q)
    od.impose_bc(lambda b, a: a.copy(), view, "outlet", Q)


def _worker(rank, world, port, out, with_bc=False):
    from conftest import ADV_FAMILIES, oracle_boundaries_view
    from oracle import domain as od
    from oracle.solver import FAS as oFAS
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        msh = advection_mesh(2e-2)
        lv = RankLevels(msh, rank + 1, world, MAXLEV,
                        domain_kwargs=dict(hypercube_families=ADV_FAMILIES) if with_bc else dict(boundaries=False))
        assert lv.n_levels == MAXLEV + 1
        views = [oracle_boundaries_view(ld) for ld in lv.local_doms] if with_bc else None
        oparts = [oracle_view(p) for p in lv.parts]
        hxs = [HaloExchange(pl, "cpu") for pl in lv.plans]
        reds = [Reductions(p.image_in_domain) for p in lv.parts]
        norms = []
        ncs = [int(p.domain.size) for p in lv.parts]

        def exchange(l, Q):
            hxs[l].exchange(torch.from_numpy(Q))        # in place: skirt rows (and donor extras) of level l

        def f(l, Q):
            if with_bc:
                # impose_bc! on the ghosts this rank owns (the level's own boundaries, local rows), then the ghosts the
                # peers own arrive with a second exchange: the sweep sees every ghost as the one-partition run does
                _bcs(od, views[l], Q)
                exchange(l, Q)
            r = np.zeros_like(Q)
            rr, om = _residual(od, oparts[l], Q[:ncs[l]])
            r[:ncs[l]] = rr
            return r, om

        def level_norm(l, r):
            v = f32(reds[l].norm(r[:ncs[l]]))
            norms.append(float(v))
            return v
        X0 = lv.doms[0].global_centers()
        Qg = seeded_field(X0, nv=2)
        gids = np.concatenate([lv.parts[0].domain, lv.extras[0][rank + 1]]).astype(np.int64)
        Q = Qg[gids].copy()
        own = np.zeros(gids.size, dtype=bool)
        own[lv.parts[0].image_in_domain] = True
        Q[~own] = np.nan                                 # stale skirt rows: the exchange must refresh them
        ratio = oFAS(f, Q, coarseners=[_oacc(a) for a in lv.coarseners], prolongators=[_oacc(a) for a in lv.prolongators],
                     n_iter=N_ITER, rtol=f32(0.0), atol=f32(0.0), exchange=exchange, level_norm=level_norm)
        img = lv.parts[0].image_in_domain
        out.put((rank, gids[img], Q[img].copy(), norms, float(ratio),
                 [int(lv.extras[l][rank + 1].size) for l in range(lv.n_levels)], [p.n_recv for p in lv.plans]))
    finally:
        dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("with_bc", [False, True], ids=["no_ghost_cells", "impose_bc_on_every_level"])
def test_two_rank_vcycle_matches_the_one_partition_vcycle(with_bc):
    from conftest import ADV_FAMILIES, oracle_boundaries_view
    from oracle import domain as od
    from oracle.solver import FAS as oFAS
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, with_bc)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # the one-partition run: global operators of the product's multigrid(), the oracle's loop and operators
    msh = advection_mesh(2e-2)
    dom = (ibamd.Domain(msh, max_partition_size=10 ** 9, hypercube_families=ADV_FAMILIES) if with_bc
           else ibamd.Domain(msh, max_partition_size=10 ** 9, boundaries=False))
    cds, prol, coar = ibamd.multigrid(dom, max_levels=MAXLEV)
    oparts = [oracle_view(d.partitions[1]) for d in [dom] + cds]
    gviews = [oracle_boundaries_view(d) for d in [dom] + cds] if with_bc else None
    norms_ref = []

    def f(l, Q):
        if with_bc:
            _bcs(od, gviews[l], Q)
        return _residual(od, oparts[l], Q)

    def norm(r):
        v = np.sqrt(np.sum(r.astype(np.float64) ** 2)).astype(f32)
        norms_ref.append(float(v))
        return v
    Qref = seeded_field(dom.global_centers(), nv=2)
    Q0 = Qref.copy()
    ratio_ref = oFAS(f, Qref, coarseners=[_oacc(a) for a in coar], prolongators=[_oacc(a) for a in prol], n_iter=N_ITER,
                     rtol=f32(0.0), atol=f32(0.0), norm=norm)
    assert not np.array_equal(Qref, Q0)
    got = np.full(Qref.shape, np.nan, dtype=f32)
    for rank, gid, vals, norms, ratio, n_extra, n_recv in res:
        got[gid] = vals
        assert all(n > 0 for n in n_recv)                         # every level has a skirt to exchange
        assert len(norms) == len(norms_ref)
        assert np.allclose(norms, norms_ref, rtol=2e-6)           # all-reduced norms = the global norms
        assert abs(ratio - float(ratio_ref)) <= 1e-5 * max(1.0, float(ratio_ref))
    assert not np.isnan(got).any()
    assert np.array_equal(got, Qref)                              # same arithmetic on the same values: bit for bit
