"""The point-implicit smoother (orphan /root/reference/src/point_implicit.jl: ``hutchinson_trick`` :17-91, ``linearize``
:185-209, ``proj_along`` :221-236, ``solve`` :250-329) ACROSS RANKS, rehearsed on CPU with gloo (world 2): every rank holds
its partition (image + skirt rows), the residual closure refreshes the skirt rows before every sweep -- every Hutchinson
sample and every Jacobian-vector product is one -- and returns zeros outside the owned rows
(``distributed.RankOps.closure``), the dot products of ``proj_along``, the norm and ``max |r|`` of ``solve`` are all-reduced
(``RankOps.sum`` / ``.max``).  Arithmetic = the oracle's on both sides (numpy operators, the oracle's smoother with the
``reduce`` hook); index bookkeeping, exchange and reductions = the product's.  With the same +-1 samples the two-rank
linearisation reproduces the one-partition block diagonal BIT FOR BIT on every owned cell, and two relaxation steps agree to
the rounding of the all-reduced sums."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import ibamd
from conftest import advection_mesh, oracle_view, seeded_field
from ibamd.distributed import RankOps
from ibamd.halo import HaloExchange, HaloPlan

f32 = np.float32
H, DT, N_SAMPLES, N_ITER = f32(1e-3), f32(2e-5), 2, 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _residual(od, opart, X):
    """a nonlinear two-component residual on a partition: Laplacian (test/dissipation.jl:69-77) minus a cubic reaction"""
    r = np.zeros_like(X)
    for dim in (1, 2):
        r += od.green_gauss(opart, od.face_gradient(opart, X, dim), dim)
    return r - f32(50.0) * X * X * X


def _samples(n_global):
    rng = np.random.default_rng(11)
    return [[rng.choice(f32([-1, 1]), n_global).astype(f32) for _ in range(N_SAMPLES)] for _ in range(2)]


class _OneRank:
    """the reduce hooks of a one-partition run: nothing to sum over (keeps the Float64 partial sums of the hooked path)"""

    def sum(self, t):
        return t

    def max(self, t):
        return t


def _mps(msh, world):
    npb = msh.block_size ** msh.ndims
    return -(-(-(-len(msh) // world)) // npb) * npb


def _worker(rank, world, port, out):
    from oracle import domain as od
    from oracle import point_implicit as opi
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        msh = advection_mesh(2e-2)
        dom = ibamd.Domain(msh, max_partition_size=_mps(msh, world), boundaries=False, only=[rank + 1])
        part = dom.partitions[rank + 1]
        opart = oracle_view(part)
        hx = HaloExchange(HaloPlan(dom, rank + 1), "cpu")
        gids = np.asarray(part.domain, dtype=np.int64)
        n_rows = gids.size
        ops = RankOps(part.image_in_domain, n_rows, lambda X: hx.exchange(torch.from_numpy(X)))
        Xg = seeded_field(dom.global_centers(), nv=2)
        X0 = Xg[gids].copy()
        own = ops.mask_np.astype(bool)
        X0[~own] = np.nan                                    # stale skirt rows: every evaluation must refresh them
        Xref = Xg[gids].copy()                               # (the pseudo-time anchor, valid on every row)

        def f_local(X):
            return ((X - Xref) / DT - _residual(od, opart, X)).astype(f32)
        f = ops.closure(f_local)
        samp = [[z[gids] for z in col] for col in _samples(Xg.shape[0])]
        lin, b, invD = opi.linearize(f, X0, samp, h=H)
        assert np.all(b[~own] == 0) and np.all(invD[~own] == 0)
        x, ratio = opi.solve(lin, b, invD, n_iter=N_ITER, rtol=0.0, atol=0.0, reduce=ops)
        img = np.asarray(part.image_in_domain)
        out.put((rank, gids[img], invD[img].copy(), b[img].copy(), x[img].copy(), float(ratio), int(hx.plan.n_recv)))
    finally:
        dist.destroy_process_group()


def test_two_rank_point_implicit_matches_the_one_partition_smoother():
    from oracle import domain as od
    from oracle import point_implicit as opi
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # the one-partition run: same closure, same samples, the hooked loop (Float64 partial sums) without peers
    msh = advection_mesh(2e-2)
    dom = ibamd.Domain(msh, max_partition_size=10 ** 9, boundaries=False)
    opart = oracle_view(dom.partitions[1])
    Xg = seeded_field(dom.global_centers(), nv=2)

    def f(X):
        return ((X - Xg) / DT - _residual(od, opart, X)).astype(f32)
    lin, b, invD = opi.linearize(f, Xg.copy(), _samples(Xg.shape[0]), h=H)
    x, ratio = opi.solve(lin, b, invD, n_iter=N_ITER, rtol=0.0, atol=0.0, reduce=_OneRank())
    assert ratio < 0.5                                        # the relaxation does reduce the residual
    n = Xg.shape[0]
    gD, gb, gx = np.full(invD.shape, np.nan, f32), np.full(b.shape, np.nan, f32), np.full(x.shape, np.nan, f32)
    for rank, gid, Dr, br, xr, rr, n_recv in res:
        assert n_recv > 0
        gD[gid], gb[gid], gx[gid] = Dr, br, xr
        assert abs(rr - float(ratio)) <= 1e-5 * max(1.0, float(ratio))
    assert not (np.isnan(gD).any() or np.isnan(gb).any() or np.isnan(gx).any())
    assert np.array_equal(gb, b)                              # right-hand side: same sweep on the same values
    assert np.array_equal(gD, invD)                           # Hutchinson blocks + pinv: bit for bit on every owned cell
    # two relaxation steps: alpha = (As . r) / (As . As) from all-reduced Float64 sums (another summation order)
    assert np.abs(gx - x).max() <= 1e-5 * np.abs(x).max()
    assert n == gx.shape[0]
