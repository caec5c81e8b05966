"""The two CPU restatements of the scalar residual sweep against each other: the C one (oracle/csrc/residual.c,
array pass per reference broadcast) reproduces the numpy one (oracle/domain.py) bit for bit, with one thread
and with several; the cell-fused C form agrees to rounding."""
import numpy as np
import pytest

from conftest import rel_inf, seeded_field
from oracle import residual_c as rc
from test_gpu_residual import oracle_advection_residual

f32 = np.float32


@pytest.mark.parametrize("kind", ["smooth", "step"])
def test_c_restatement_equals_numpy_restatement(adv_domains, rae_domains, kind):
    for _, do in (adv_domains, rae_domains):
        for k, opart in do.partitions.items():
            u = seeded_field(opart.centers, kind=kind)
            C = np.stack([np.ones_like(u), f32(0.5) + seeded_field(opart.centers, seed=3) * f32(0.1)], axis=1)
            exp = oracle_advection_residual(opart, u, C)
            cp = rc.CPart(opart)
            for threads in (1, 3):
                got = cp.residual_advection(u, C, threads=threads)
                assert np.array_equal(got, exp), (k, threads)
            fused = cp.residual_advection(u, C, fused=True, threads=2)
            assert rel_inf(fused, exp) <= 1e-5


def test_c_restatement_3d():
    """Same check on an octree (two levels, 4^3 blocks), product host partition viewed through the oracle."""
    import ibamd
    from conftest import oracle_view
    from ibamd.mesher import Ball, Mesh
    msh = Mesh(f32([-2, -2, -2]), f32([4, 4, 4]), block_size=4,
               refinement_regions=[(Ball(np.array([0.6, 0.6, 0.6]), 0.1), f32(0.2))])
    dom = ibamd.Domain(msh, max_partition_size=8192, boundaries=False)
    part = next(iter(dom.partitions.values()))
    op = oracle_view(part)
    n = part.spacing.shape[0]
    rng = np.random.default_rng(3)
    u = rng.uniform(-1, 1, n).astype(f32)
    C = rng.uniform(-1, 1, (n, 3)).astype(f32)
    exp = oracle_advection_residual(op, u, C)
    got = rc.CPart(part).residual_advection(u, C, threads=2)
    assert np.array_equal(got, exp)


def test_c_euler_restatement_equals_numpy_restatement(adv_domains, rae_domains):
    """Euler sweep (R2): the C restatement reproduces the numpy composition (JST sensor, cell_gradient, MUSCL,
    HLL with its Float64 combine, green_gauss) bit for bit."""
    from conftest import euler_field
    from oracle import cfd as ocfd
    from test_gpu_residual import oracle_euler_residual
    for _, do in (adv_domains, rae_domains):
        for k, opart in do.partitions.items():
            P = euler_field(opart.centers)
            exp = oracle_euler_residual(opart, P, ocfd.Fluid())
            cp = rc.CPart(opart)
            for threads in (1, 3):
                got = cp.residual_euler(P, threads=threads)
                assert np.array_equal(got, exp), (k, threads)
