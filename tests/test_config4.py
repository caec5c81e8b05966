"""BASELINE.json configs[3] in one piece, on one GPU at > 100 k cells: a 3-D sphere (STL) immersed in a block octree,
ghost-layer interpolation with FlowBC closures (`impose_bc!`), the Euler HLL residual sweep, and one point-implicit
linearise + solve that drives it -- each step against the oracle.  The domain comes from the product's O(N) builder
(ghost projections vectorised, domain.py::_project_3d); its ghost set is compared with the literal per-ghost
restatement on a sample in tests/test_3d.py."""
import os
import sys

import numpy as np
import pytest

import ibamd
from conftest import oracle_boundaries_view, oracle_view, rel_inf

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
f32 = np.float32
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sphere_case():
    import bench
    from ibamd.mesher import Mesh
    msh = Mesh(f32([-4, -4, -4]), f32([8, 8, 8]), ("sphere", bench.icosphere(subdiv=2), f32(0.14)), block_size=8)
    fam = [("farfield", [(d, s) for d in (1, 2, 3) for s in (False, True)])]
    dom = ibamd.Domain(msh, hypercube_families=fam, max_partition_size=10 ** 9)
    (part,) = dom.partitions.values()
    return msh, dom, part


def _field(n, seed):
    rng = np.random.default_rng(seed)
    P = np.empty((n, 5), dtype=f32)
    P[:, 0] = 1e5 * (1 + 0.02 * rng.uniform(-1, 1, n))
    P[:, 1] = 288.15 * (1 + 0.02 * rng.uniform(-1, 1, n))
    P[:, 2] = 100.0 * (1 + 0.1 * rng.uniform(-1, 1, n))
    P[:, 3] = 10.0 * rng.uniform(-1, 1, n)
    P[:, 4] = 10.0 * rng.uniform(-1, 1, n)
    return P


def test_config4_bc_sweep_point_implicit(sphere_case):
    import torch
    from ibamd import cfd as gcfd
    from ibamd import point_implicit as pi
    from oracle import cfd as ocfd
    from oracle import domain as od
    from oracle import point_implicit as opi
    from test_gpu_residual import oracle_euler_residual
    msh, dom, part = sphere_case
    n = len(dom)
    assert n >= 100_000 and dom.ndims == 3
    nghost = sum(b.ghost_indices.size for b in dom.boundaries["sphere"].values())
    assert nghost > 1000
    dpart = ibamd.to_backend(part, ibamd.hip)
    assert dpart.info["full_blocks"] * 512 == n          # the 3-D block kernels run the sweep
    view = oracle_boundaries_view(dom)
    opart = oracle_view(part)

    # ---- 1. impose_bc! with FlowBC: slip wall on the sphere, free stream on the box
    P0 = _field(n, 4)
    far = [1.0e5, 288.15, 100.0, 0.0, 0.0]
    o_far, g_far = ocfd.FlowBC(ocfd.Fluid(), f32(far)), gcfd.FlowBC(gcfd.Fluid(), far)
    o_wall = ocfd.FlowBC(ocfd.Fluid(), f32([1.0e5, 288.15, 0.0]), normal_flow=True)
    g_wall = gcfd.FlowBC(gcfd.Fluid(), [1.0e5, 288.15, 0.0], normal_flow=True)
    Po, Pg = P0.copy(), ibamd.hip(P0)
    for name, fo, fg in (("farfield", o_far, g_far), ("sphere", o_wall, g_wall)):
        od.impose_bc(lambda b, ia: fo(ia, b.normals), view, name, Po)
        ibamd.impose_bc(lambda b, ia: fg(ia, b.normals), dom, name, Pg)
    got = ibamd.to_host(Pg)
    assert not np.array_equal(got, P0)
    for v in range(5):
        assert rel_inf(got[:, v], Po[:, v]) <= 1e-5

    # ---- 2. the Euler HLL residual sweep on the BC-corrected state (one partition: local order = global order)
    assert np.array_equal(part.domain, np.arange(n))
    Ro = oracle_euler_residual(opart, Po, ocfd.Fluid())
    Rg = ibamd.to_host(ibamd.residual_euler_hll(dpart, Pg))
    for v in range(5):
        assert rel_inf(Rg[:, v], Ro[:, v]) <= 1e-5

    # ---- 3. one point-implicit linearisation + solve of a pseudo-time step (P - P0)/dt - R(P) = 0, the sweep as f.
    # Same +-1 sample vectors on both sides; the CPU side runs the oracle's smoother over the oracle's sweep.  dt is
    # small enough for the blocks to be diagonally dominant (1/dt against a/h ~ 1e4 1/s): a well-posed comparison.
    scale = f32([1e5, 288.15, 100.0, 100.0, 100.0])
    dt = f32(1e-5)
    Pg0 = Pg.clone()

    def f_dev(X):
        return (X - Pg0) / float(dt) - ibamd.residual_euler_hll(dpart, X)

    def f_np(X):
        return ((X - Po) / dt - oracle_euler_residual(opart, X, ocfd.Fluid())).astype(f32)
    rng = np.random.default_rng(3)
    samp = [[rng.choice(f32([-1, 1]), n).astype(f32) for _ in range(2)] for _ in range(5)]
    dsamp = [[ibamd.hip(z) for z in col] for col in samp]
    X0 = Po + f32(1e-3) * scale * np.random.default_rng(9).uniform(-1, 1, (n, 5)).astype(f32)
    h = 1e-2
    D_n = opi.hutchinson_trick(f_np, X0, samp, h=f32(h))
    D_g = pi.hutchinson_trick(f_dev, ibamd.hip(X0), 2, h=h, samples=dsamp).cpu().numpy()
    # Float32 finite differences on both sides, h = 1e-2 on a +-1 sample: for the pressure column that perturbation is ONE
    # ulp of 1e5 (0.0078), so a Jacobian-vector product carries ~1e-2 of rounding noise whatever computes it.  The
    # kernels behind these products are compared tightly on exact pointwise systems (tests/test_point_implicit.py:
    # block inversion 2e-4 of the block scale, block apply 1e-5); here the comparison can only show that both sides see the same operator.
    assert np.abs(D_g - D_n).max() <= 1e-2 * np.abs(D_n).max()
    lin_n, b_n, invD_n = opi.linearize(f_np, X0, samp, h=f32(h))
    lin, b, prec = pi.linearize(f_dev, ibamd.hip(X0), 2, h=h, samples=dsamp)
    for v in range(5):
        assert rel_inf(ibamd.to_host(b)[:, v], b_n[:, v]) <= 1e-4      # right-hand side = -f(X0)
    vdir = (f32(1e-2) * scale * np.sin(part.centers[:, :1] * f32(3.0))).astype(f32) * np.ones((1, 5), f32)
    Av_n, Av_g = lin_n(vdir), ibamd.to_host(lin(ibamd.hip(vdir)))
    for v in range(5):
        assert rel_inf(Av_g[:, v], Av_n[:, v]) <= 1e-2
    x_n, ratio_n = opi.solve(lin_n, b_n, invD_n, n_iter=2, rtol=1e-6)
    x, ratio = pi.solve(lin, b, prec, n_iter=2, rtol=1e-6)
    xg = ibamd.to_host(x)
    assert np.isfinite(xg).all() and np.isfinite(x_n).all()
    assert ratio < 0.9 and ratio_n < 0.9 and abs(ratio - ratio_n) <= 0.05   # both relaxations reduce the residual alike
    # the update itself: two relaxation steps through Float32 finite-difference products on both sides (noise ~1e-2
    # per product, amplified by the block inverse): agreement in the L2 sense per variable
    for v in range(5):
        err = np.linalg.norm((xg[:, v] - x_n[:, v]).astype(np.float64)) / np.linalg.norm(x_n[:, v].astype(np.float64))
        assert err <= 0.15, (v, err)
