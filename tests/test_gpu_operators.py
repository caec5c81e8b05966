"""Parity of every grid operator (C ABI -> HIP) with the oracle, on seeded inputs.

Reference operators: /root/reference/src/ImmersedBoundary.jl:873-1157.  Tolerance: 1e-5 norm-wise
(north_star); the face-list kernels follow the reference's evaluation order, so the observed
difference is expected to be exactly 0 and is asserted at 1e-6 to catch regressions early.
"""
import numpy as np
import pytest

import ibamd
from conftest import rel_inf, seeded_field
from oracle import domain as od

pytestmark = pytest.mark.gpu
TOL = 1e-5
TIGHT = 1e-6


def _parts(domains):
    dp, do = domains
    for k in dp.partitions:
        yield ibamd.to_backend(dp.partitions[k], ibamd.hip), do.partitions[k]


@pytest.mark.parametrize("nv", [None, 2])
@pytest.mark.parametrize("kind", ["smooth", "step"])
def test_cell_and_face_operators(adv_domains, nv, kind):
    for dpart, opart in _parts(adv_domains):
        u = seeded_field(opart.centers, nv=nv, kind=kind)
        ud = ibamd.hip(u)
        for dim in (1, 2):
            for name in ("at_owners", "at_neighbors", "at_faces", "cell_gradient", "face_gradient"):
                got = ibamd.to_host(getattr(ibamd, name)(dpart, ud, dim))
                exp = getattr(od, name)(opart, u, dim)
                assert got.shape == exp.shape
                assert rel_inf(got, exp) <= TIGHT, (name, dim)
            uf = od.at_faces(opart, u, dim)
            ufd = ibamd.hip(uf)
            for name in ("green_gauss", "unsigned_green_gauss"):
                got = ibamd.to_host(getattr(ibamd, name)(dpart, ufd, dim))
                assert rel_inf(got, getattr(od, name)(opart, uf, dim)) <= TIGHT, (name, dim)
            for name in ("face_distance", "owner_distance", "neighbor_distance"):
                got = ibamd.to_host(getattr(ibamd, name)(dpart, dim))
                assert np.array_equal(got, getattr(od, name)(opart, dim)), (name, dim)
        got = ibamd.to_host(ibamd.JST_sensor(dpart, ud))
        assert rel_inf(got, od.JST_sensor(opart, u)) <= TOL
        for dim in (1, 2):
            got = ibamd.to_host(ibamd.JST_sensor(dpart, ud, dim))
            assert rel_inf(got, od.JST_sensor(opart, u, dim)) <= TOL


@pytest.mark.parametrize("nv", [None, 3])
@pytest.mark.parametrize("with_D,high_order", [(False, False), (True, False), (True, True)])
def test_muscl(adv_domains, nv, with_D, high_order):
    for dpart, opart in _parts(adv_domains):
        u = seeded_field(opart.centers, nv=nv, kind="step")
        p = seeded_field(opart.centers, seed=7)
        D = od.JST_sensor(opart, p) if with_D else None
        for dim in (1, 2):
            du = od.cell_gradient(opart, u, dim)
            eL, eR = od.MUSCL(opart, u, du, dim, D=D, high_order=high_order)
            gL, gR = ibamd.MUSCL(dpart, ibamd.hip(u), ibamd.hip(du), dim,
                                 D=None if D is None else ibamd.hip(D), high_order=high_order)
            assert rel_inf(ibamd.to_host(gL), eL) <= TIGHT
            assert rel_inf(ibamd.to_host(gR), eR) <= TIGHT


def test_divergent_and_tuple_forms(adv_domains):
    for dpart, opart in _parts(adv_domains):
        u = seeded_field(opart.centers, nv=2)
        ud = ibamd.hip(u)
        g_exp = od.cell_gradient(opart, u)
        g_got = ibamd.cell_gradient(dpart, ud)
        for a, b in zip(g_got, g_exp):
            assert rel_inf(ibamd.to_host(a), b) <= TIGHT
        ufs = tuple(od.at_faces(opart, u, d) for d in (1, 2))
        got = ibamd.divergent(dpart, tuple(ibamd.hip(x) for x in ufs))
        assert rel_inf(ibamd.to_host(got), od.divergent(opart, ufs)) <= TOL
        fg_exp = od.face_gradient(opart, u, g_exp, 2)
        fg_got = ibamd.face_gradient(dpart, ud, g_got, 2)
        for a, b in zip(fg_got, fg_exp):
            assert rel_inf(ibamd.to_host(a), b) <= TIGHT


def test_operators_reject_host_arrays(adv_domains):
    dp, _ = adv_domains
    part = dp.partitions[1]
    u = np.zeros(part.spacing.shape[0], dtype=np.float32)
    with pytest.raises(TypeError):
        ibamd.cell_gradient(part, u, 1)            # host partition: no CPU path
    dpart = ibamd.to_backend(part, ibamd.hip)
    with pytest.raises(TypeError):
        ibamd.cell_gradient(dpart, u, 1)           # host field: no CPU path


def test_accumulator_docstring_example():
    """accumulator.jl:26-33: acc([1,2,3,4]) == [3, 38]."""
    acc = ibamd.Accumulator([[0, 1], [1, 2, 3]], [[-1.0, 2.0], [3.0, 4.0, 5.0]], n_input=4)
    got = ibamd.to_host(ibamd.to_backend(acc)(ibamd.hip(np.array([1, 2, 3, 4], dtype=np.float32))))
    assert np.array_equal(got, np.array([3.0, 38.0], dtype=np.float32))


def test_tuple_cell_gradient_block_fast_path(rae_domains):
    """cell_gradient(part, u) -- the tuple form (ImmersedBoundary.jl:980-988) -- runs all dimensions in one sweep per field
    (ibh_cell_gradient_nd: pass A of the two-kernel sweeps on block-structured partitions, face-list threads for the
    cells outside blocks): against the oracle and against the operator-by-operator kernels, one and two fields."""
    dp, do = rae_domains
    rng = np.random.default_rng(3)
    used_blocks = 0
    for k in dp.partitions:
        part, opart = dp.partitions[k], do.partitions[k]
        dpart = ibamd.to_backend(part, ibamd.hip)
        used_blocks += dpart.info["full_blocks"]
        n = part.centers.shape[0]
        u1 = (np.sin(3 * part.centers[:, 0]) + 0.2 * rng.uniform(-1, 1, n)).astype(np.float32)
        u2 = np.stack([u1, (np.cos(2 * part.centers[:, 1]) + 0.1 * rng.uniform(-1, 1, n)).astype(np.float32)], axis=1)
        for u in (u1, u2):
            got = ibamd.cell_gradient(dpart, ibamd.hip(u))
            assert isinstance(got, tuple) and len(got) == 2
            for d in (1, 2):
                exp = od.cell_gradient(opart, u, d)
                one = ibamd.to_host(ibamd.cell_gradient(dpart, ibamd.hip(u), d))
                g = ibamd.to_host(got[d - 1])
                assert g.shape == exp.shape
                assert rel_inf(one, exp) <= 1e-6
                assert rel_inf(g, exp) <= 5e-6          # tuned arithmetic (reciprocal spacings)
    assert used_blocks > 0


def test_cell_gradient_fields_entry_on_both_kinds_of_partition(rae_mesh_small):
    """``ibh_cell_gradient_fields`` called directly: out (nc, nv (nd + 1)) with the gradient of field v along d in column
    v (nd + 1) + d and its JST sensor in column v (nd + 1) + nd -- on a block-structured partition (every field's block
    sweep in place) and on a partition without block structure (a coarse level of ``multigrid``: the face-list kernels field
    by field), against ``ibh_cell_gradient_nd`` / ``cell_gradient(part, u, d)`` and ``JST_sensor(part, u)``."""
    import torch
    from ibamd import _lib
    from ibamd import backend as B
    dp = ibamd.Domain(rae_mesh_small, max_partition_size=10 ** 9, boundaries=False)
    cds, _, _ = ibamd.multigrid(dp, max_levels=1)
    rng = np.random.default_rng(9)
    seen = set()
    for part in (dp.partitions[1], cds[0].partitions[1]):
        dpart = ibamd.to_backend(part, ibamd.hip)
        seen.add(dpart.info["full_blocks"] > 0)
        nc, nd, nv = dpart.nc, 2, 3
        U = ibamd.hip((np.sin(3 * part.centers[:, :1]) + 0.3 * rng.uniform(-1, 1, (nc, nv))).astype(np.float32))
        out = ibamd.colmajor_empty(nc, nv * (nd + 1))
        out.fill_(float("nan"))
        B._stream()
        _lib.call("ibh_cell_gradient_fields", dpart.handle, B._ptr(U), nv, nc, B._ptr(out))
        assert torch.isfinite(out).all()
        for v in range(nv):
            uv = U[:, v].contiguous()
            g = ibamd.cell_gradient(dpart, uv)                      # the one-field tuple form (in place as well)
            for d in range(nd):
                assert torch.equal(out[:, v * (nd + 1) + d], g[d])
            sens = ibamd.JST_sensor(dpart, uv)
            assert rel_inf(ibamd.to_host(out[:, v * (nd + 1) + nd]), ibamd.to_host(sens)) <= 5e-6
    assert seen == {True, False}
