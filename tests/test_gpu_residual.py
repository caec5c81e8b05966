"""Parity of the fused residual sweeps with the oracle's operator-by-operator closures.

R1 = closure of /root/reference/test/advection.jl:67-83; R2 = Euler HLL residual composed from
reference operators.  Tolerance 1e-5 norm-wise per variable (north_star).
"""
import numpy as np
import pytest

import ibamd
from conftest import euler_field, rel_inf, seeded_field
from oracle import cfd as ocfd
from oracle import domain as od

pytestmark = pytest.mark.gpu
TOL = 1e-5
f32 = np.float32


def oracle_advection_residual(part, u, C):
    """test/advection.jl:67-83 with ud starting from zero."""
    ud = np.zeros_like(u)
    D = od.JST_sensor(part, u)
    for dim in range(1, part.ndims + 1):
        Cf = od.at_faces(part, np.ascontiguousarray(C[:, dim - 1]), dim)
        gu = od.cell_gradient(part, u, dim)
        uL, uR = od.MUSCL(part, u, gu, dim, D=D, high_order=True)
        ud -= od.green_gauss(part, (uL + uR) * Cf / f32(2) + np.abs(Cf) * (uL - uR) / f32(2), dim)
    return ud


def oracle_euler_residual(part, P, fluid):
    R = np.zeros_like(P)
    D = od.JST_sensor(part, np.ascontiguousarray(P[:, 0]))
    for dim in range(1, part.ndims + 1):
        gP = od.cell_gradient(part, P, dim)
        PL, PR = od.MUSCL(part, P, gP, dim, D=D, high_order=True)
        F = ocfd.inviscid_fluxes(fluid, PL, PR, dim)
        R -= od.green_gauss(part, F, dim)  # Float64 flux, rounded on the in-place update
    return R


def _parts(domains):
    dp, do = domains
    for k in dp.partitions:
        yield ibamd.to_backend(dp.partitions[k], ibamd.hip), do.partitions[k]


@pytest.mark.parametrize("kind", ["smooth", "step"])
@pytest.mark.parametrize("flags", [0, 1, 16])  # tuned block path, face lists, literal block path
def test_advection_residual(adv_domains, kind, flags):
    for dpart, opart in _parts(adv_domains):
        u = seeded_field(opart.centers, kind=kind)
        C = np.stack([np.ones_like(u), f32(0.5) + seeded_field(opart.centers, seed=3) * f32(0.1)], axis=1)
        exp = oracle_advection_residual(opart, u, C)
        got = ibamd.to_host(ibamd.residual_advection(dpart, ibamd.hip(u), ibamd.hip(C), flags=flags))
        assert rel_inf(got, exp) <= TOL


def test_advection_residual_rae(rae_domains):
    dp, do = rae_domains
    for k in dp.partitions:
        dpart, opart = ibamd.to_backend(dp.partitions[k], ibamd.hip), do.partitions[k]
        u = seeded_field(opart.centers)
        C = np.ones((u.shape[0], 2), dtype=f32)
        exp = oracle_advection_residual(opart, u, C)
        fast = ibamd.to_host(ibamd.residual_advection(dpart, ibamd.hip(u), ibamd.hip(C)))
        gen = ibamd.to_host(ibamd.residual_advection(dpart, ibamd.hip(u), ibamd.hip(C), flags=1))
        lit = ibamd.to_host(ibamd.residual_advection(dpart, ibamd.hip(u), ibamd.hip(C), flags=16))
        assert rel_inf(gen, exp) <= TOL
        assert rel_inf(fast, exp) <= TOL
        # the literal block path and the face-list path share their per-face arithmetic: same bits
        assert np.array_equal(lit, gen)
        assert dpart.info["full_blocks"] > 0


def test_block_analysis_classifies_rae_mesh(rae_domains):
    dp, _ = rae_domains
    tot_blocks = 0
    for k in dp.partitions:
        info = ibamd.to_backend(dp.partitions[k], ibamd.hip).info
        tot_blocks += info["full_blocks"]
        assert info["sides_general"] <= 0.2 * max(1, 4 * info["full_blocks"])
    assert tot_blocks >= dp.mesh.nblocks  # every image block is a full block in its own partition


@pytest.mark.parametrize("flags", [0, 1])  # tuned block path (Float32 HLL combine), literal face-list path
def test_euler_residual(adv_domains, flags):
    fluid = ocfd.Fluid()
    for dpart, opart in _parts(adv_domains):
        P = euler_field(opart.centers)
        exp = oracle_euler_residual(opart, P, fluid)
        got = ibamd.to_host(ibamd.residual_euler_hll(dpart, ibamd.hip(P), flags=flags))
        assert rel_inf(got, exp) <= TOL


def test_euler_residual_rae(rae_domains):
    dp, do = rae_domains
    fluid = ocfd.Fluid()
    worst = 0.0
    for k in dp.partitions:
        dpart, opart = ibamd.to_backend(dp.partitions[k], ibamd.hip), do.partitions[k]
        P = euler_field(opart.centers)
        exp = oracle_euler_residual(opart, P, fluid)
        for flags in (0, 1):
            got = ibamd.to_host(ibamd.residual_euler_hll(dpart, ibamd.hip(P), flags=flags))
            worst = max(worst, rel_inf(got, exp))
    assert worst <= TOL


@pytest.mark.parametrize("flags", [0, 16])
def test_overlap_phases_equal_full_sweep(rae_domains, flags):
    """IBH_PHASE_INTERIOR + IBH_PHASE_BOUNDARY == one full sweep, bit for bit, and the interior phase
    does not read skirt cells (they are poisoned with NaN while it runs)."""
    import torch
    dp, _ = rae_domains
    n_int = 0
    for k in dp.partitions:
        part = dp.partitions[k]
        dpart = ibamd.to_backend(part, ibamd.hip)
        n_int += dpart.info["interior_blocks"]
        u = seeded_field(part.centers)
        C = np.ones((u.shape[0], 2), dtype=f32)
        ud_full = ibamd.to_host(ibamd.residual_advection(dpart, ibamd.hip(u), ibamd.hip(C), flags=flags))
        skirt = np.ones(u.shape[0], dtype=bool)
        skirt[part.image_in_domain] = False
        up = u.copy()
        up[skirt] = np.nan
        ud = torch.full((u.shape[0],), float("nan"), dtype=torch.float32, device="cuda")
        ibamd.residual_advection(dpart, ibamd.hip(up), ibamd.hip(C), out=ud, flags=flags | ibamd.IBH_PHASE_INTERIOR)
        got1 = ibamd.to_host(ud)
        done = ~np.isnan(got1)
        assert np.array_equal(got1[done], ud_full[done])          # interior results final and NaN-free
        ibamd.residual_advection(dpart, ibamd.hip(u), ibamd.hip(C), out=ud, flags=flags | ibamd.IBH_PHASE_BOUNDARY)
        got = ibamd.to_host(ud)
        img = part.image_in_domain
        assert np.array_equal(got[img], ud_full[img])
    assert n_int > 0


@pytest.mark.parametrize("kind", ["smooth", "step", "noise"])
def test_single_kernel_sweep(rae_mesh_small, kind):
    """One partition with no face-list cells: the whole sweep is one launch that also computes gradient and
    sensor of the halo cells (blk2::sweep_adv).  It must agree with the oracle, with the two-kernel form
    (IBH_NO_FUSE) and with itself when split in overlap phases."""
    from conftest import RAE_FAMILIES
    dom = ibamd.Domain(rae_mesh_small, hypercube_families=RAE_FAMILIES, max_partition_size=10 ** 9, boundaries=False)
    (part,) = dom.partitions.values()
    dpart = ibamd.to_backend(part, ibamd.hip)
    assert dpart.info["fusable_blocks"] == dpart.info["full_blocks"] > 0 and dpart.info["irregular_cells"] == 0
    if kind == "noise":
        u = np.random.default_rng(7).uniform(-1, 1, part.centers.shape[0]).astype(f32)
    else:
        u = seeded_field(part.centers, kind=kind)
    C = np.stack([f32(1) + seeded_field(part.centers, seed=5) * f32(0.3),
                  f32(-0.5) + seeded_field(part.centers, seed=3) * f32(0.1)], axis=1)
    from conftest import oracle_view
    exp = oracle_advection_residual(oracle_view(part), u, C)
    one = ibamd.to_host(ibamd.residual_advection(dpart, ibamd.hip(u), ibamd.hip(C)))
    two = ibamd.to_host(ibamd.residual_advection(dpart, ibamd.hip(u), ibamd.hip(C), flags=ibamd.IBH_NO_FUSE))
    assert rel_inf(two, exp) <= TOL
    assert rel_inf(one, exp) <= TOL
    assert rel_inf(one, two) <= 2e-6
    import torch
    ud = torch.full((u.shape[0],), float("nan"), dtype=torch.float32, device="cuda")
    ibamd.residual_advection(dpart, ibamd.hip(u), ibamd.hip(C), out=ud, flags=ibamd.IBH_PHASE_INTERIOR)
    ibamd.residual_advection(dpart, ibamd.hip(u), ibamd.hip(C), out=ud, flags=ibamd.IBH_PHASE_BOUNDARY)
    assert np.array_equal(ibamd.to_host(ud), one)


def test_mixed_launch_matches_two_kernel_form(rae_domains):
    """Partitions with skirt blocks: eligible blocks go through the single kernel, the rest through the
    two-kernel form with the workspace filled only where it is read.  Same results as the pure two-kernel
    form and as the oracle, on every cell."""
    dp, do = rae_domains
    mixed = 0
    for k in dp.partitions:
        dpart, opart = ibamd.to_backend(dp.partitions[k], ibamd.hip), do.partitions[k]
        info = dpart.info
        if info["fusable_blocks"] > 0 and info["irregular_cells"] > 0:   # skirt fragments: face-list cells
            mixed += 1
            assert 0 < info["workspace_blocks"] < info["full_blocks"]
        u = seeded_field(opart.centers, kind="step")
        C = np.stack([np.ones_like(u), f32(0.5) + seeded_field(opart.centers, seed=3) * f32(0.1)], axis=1)
        exp = oracle_advection_residual(opart, u, C)
        # (small partitions take the two-kernel form by default -- launch count; IBH_FORCE_MIXED overrides)
        FM = ibamd.IBH_FORCE_MIXED
        one = ibamd.to_host(ibamd.residual_advection(dpart, ibamd.hip(u), ibamd.hip(C), flags=FM))
        import torch
        out = torch.full((u.shape[0],), float("nan"), dtype=torch.float32, device="cuda")
        ibamd.residual_advection(dpart, ibamd.hip(u), ibamd.hip(C), out=out, flags=FM | ibamd.IBH_PHASE_INTERIOR)
        ibamd.residual_advection(dpart, ibamd.hip(u), ibamd.hip(C), out=out, flags=FM | ibamd.IBH_PHASE_BOUNDARY)
        assert np.array_equal(ibamd.to_host(out), one)      # phases == whole sweep, bit for bit
        two = ibamd.to_host(ibamd.residual_advection(dpart, ibamd.hip(u), ibamd.hip(C), flags=ibamd.IBH_NO_FUSE))
        assert rel_inf(one, exp) <= TOL
        assert rel_inf(one, two) <= 2e-6
    assert mixed > 0


def test_single_kernel_euler_sweep(rae_mesh_small):
    """Euler sweep on one partition with every block eligible: one launch (blk2::sweep_euler), against the oracle and
    against the two-kernel form."""
    from conftest import RAE_FAMILIES, oracle_view
    dom = ibamd.Domain(rae_mesh_small, hypercube_families=RAE_FAMILIES, max_partition_size=10 ** 9, boundaries=False)
    (part,) = dom.partitions.values()
    dpart = ibamd.to_backend(part, ibamd.hip)
    assert dpart.info["fusable_blocks"] == dpart.info["full_blocks"] > 0
    P = euler_field(part.centers)
    exp = oracle_euler_residual(oracle_view(part), P, ocfd.Fluid())
    one = ibamd.to_host(ibamd.residual_euler_hll(dpart, ibamd.hip(P)))
    two = ibamd.to_host(ibamd.residual_euler_hll(dpart, ibamd.hip(P), flags=ibamd.IBH_NO_FUSE))
    blk = ibamd.to_host(ibamd.residual_euler_hll(dpart, ibamd.hip(P), flags=ibamd.IBH_NO_QUAD))  # per-block sweep
    assert dpart.info["quads"] > 0                       # `one` is the quad form (quad2::sweep_quad_euler)
    for v in range(4):
        assert rel_inf(two[:, v], exp[:, v]) <= TOL, v
        assert rel_inf(one[:, v], exp[:, v]) <= TOL, v
        assert rel_inf(blk[:, v], exp[:, v]) <= TOL, v
        assert rel_inf(one[:, v], two[:, v]) <= 5e-6, v
        assert rel_inf(one[:, v], blk[:, v]) <= 5e-6, v


def test_image_only_sweep_on_partitions(rae_domains):
    """IBH_IMAGE_ONLY on a partition with skirt fragments: every image block is eligible for the single kernel (halo
    cells in skirt fragments included, deeper cells from the table), so the sweep is one launch per phase; image
    cells agree with the oracle and with the default sweep, phases reproduce the whole sweep bit for bit and the
    interior phase reads no skirt cell."""
    import torch
    dp, do = rae_domains
    IO = ibamd.IBH_IMAGE_ONLY
    used = 0
    for k in dp.partitions:
        part, opart = dp.partitions[k], do.partitions[k]
        dpart = ibamd.to_backend(part, ibamd.hip)
        info = dpart.info
        if not info["image_blocks_all_eligible"] or info["irregular_cells"] == 0:
            continue
        used += 1
        assert info["image_blocks"] * 64 == part.image_in_domain.size
        img = part.image_in_domain
        u = seeded_field(opart.centers, kind="step")
        C = np.stack([np.ones_like(u), f32(0.5) + seeded_field(opart.centers, seed=3) * f32(0.1)], axis=1)
        exp = oracle_advection_residual(opart, u, C)
        full = ibamd.to_host(ibamd.residual_advection(dpart, ibamd.hip(u), ibamd.hip(C)))
        out = torch.full((u.shape[0],), float("nan"), dtype=torch.float32, device="cuda")
        ibamd.residual_advection(dpart, ibamd.hip(u), ibamd.hip(C), out=out, flags=IO)
        one = ibamd.to_host(out)
        assert np.isnan(one).sum() == u.shape[0] - img.size          # nothing written outside the image
        assert rel_inf(one[img], exp[img]) <= TOL
        assert rel_inf(one[img], full[img]) <= 2e-6
        skirt = np.ones(u.shape[0], dtype=bool)
        skirt[img] = False
        up = u.copy()
        up[skirt] = np.nan
        out2 = torch.full((u.shape[0],), float("nan"), dtype=torch.float32, device="cuda")
        ibamd.residual_advection(dpart, ibamd.hip(up), ibamd.hip(C), out=out2, flags=IO | ibamd.IBH_PHASE_INTERIOR)
        got1 = ibamd.to_host(out2)
        done = ~np.isnan(got1)
        # (a 2x2 block group of the quad sweep is interior only if all four blocks are: a few interior blocks wait for
        # the boundary phase)
        assert 0 < done.sum() <= 64 * info["interior_blocks"] and done.sum() % 64 == 0
        assert done.sum() >= 32 * info["interior_blocks"] and np.array_equal(got1[done], one[done])
        ibamd.residual_advection(dpart, ibamd.hip(u), ibamd.hip(C), out=out2, flags=IO | ibamd.IBH_PHASE_BOUNDARY)
        assert np.array_equal(ibamd.to_host(out2)[img], one[img])
    assert used > 0


def test_image_only_euler_sweep_on_partitions(rae_domains):
    """Euler sweep with IBH_IMAGE_ONLY on partitions with skirt fragments: one launch per phase over the image blocks;
    image cells agree with the oracle and the default (two-kernel) sweep; phases == whole."""
    import torch
    dp, do = rae_domains
    IO = ibamd.IBH_IMAGE_ONLY
    used = 0
    for k in dp.partitions:
        part, opart = dp.partitions[k], do.partitions[k]
        dpart = ibamd.to_backend(part, ibamd.hip)
        if not dpart.info["image_blocks_all_eligible"] or dpart.info["irregular_cells"] == 0:
            continue
        used += 1
        img = part.image_in_domain
        P = euler_field(opart.centers)
        exp = oracle_euler_residual(opart, P, ocfd.Fluid())
        full = ibamd.to_host(ibamd.residual_euler_hll(dpart, ibamd.hip(P)))
        out = torch.full((4, P.shape[0]), float("nan"), dtype=torch.float32, device="cuda").T
        ibamd.residual_euler_hll(dpart, ibamd.hip(P), out=out, flags=IO)
        one = ibamd.to_host(out)
        out2 = torch.full((4, P.shape[0]), float("nan"), dtype=torch.float32, device="cuda").T
        ibamd.residual_euler_hll(dpart, ibamd.hip(P), out=out2, flags=IO | ibamd.IBH_PHASE_INTERIOR)
        ibamd.residual_euler_hll(dpart, ibamd.hip(P), out=out2, flags=IO | ibamd.IBH_PHASE_BOUNDARY)
        assert np.array_equal(ibamd.to_host(out2)[img], one[img])
        for v in range(4):
            assert np.isnan(one[:, v]).sum() == P.shape[0] - img.size
            assert rel_inf(one[img, v], exp[img, v]) <= TOL, v
            assert rel_inf(one[img, v], full[img, v]) <= 5e-6, v
    assert used > 0


@pytest.mark.parametrize("case", ["advection", "rae"])
@pytest.mark.parametrize("kind", ["smooth", "step"])
def test_quad_sweep(adv_mesh, rae_mesh_small, case, kind):
    """Quad sweep (one wavefront per 2x2 group of sibling blocks, csrc/ibh_quad2d.h) on one-partition domains: against
    the oracle, against the per-block single kernel (IBH_NO_QUAD), and split in overlap phases."""
    import torch
    from conftest import ADV_FAMILIES, RAE_FAMILIES
    from oracle import residual_c as rc
    msh, fam = (adv_mesh, ADV_FAMILIES) if case == "advection" else (rae_mesh_small, RAE_FAMILIES)
    dom = ibamd.Domain(msh, hypercube_families=fam, max_partition_size=10 ** 9)
    (part,) = dom.partitions.values()
    dpart = ibamd.to_backend(part, ibamd.hip)
    assert dpart.info["quads"] > 0 and dpart.info["quads"] * 4 + dpart.info["quad_singles"] == dpart.info["full_blocks"]
    assert dpart.info["quad_pairs"] * 2 <= dpart.info["quad_singles"]
    u = seeded_field(part.centers, kind=kind)
    C = np.stack([np.ones_like(u), f32(0.5) + seeded_field(part.centers, seed=3) * f32(0.1)], axis=1)
    exp = rc.CPart(part).residual_advection(u, C)
    ud = torch.full((u.shape[0],), float("nan"), dtype=torch.float32, device="cuda")
    ibamd.residual_advection(dpart, ibamd.hip(u), ibamd.hip(C), out=ud)
    got = ibamd.to_host(ud)
    assert not np.isnan(got).any()
    assert rel_inf(got, exp) <= TOL
    per_block = ibamd.to_host(ibamd.residual_advection(dpart, ibamd.hip(u), ibamd.hip(C), flags=ibamd.IBH_NO_QUAD))
    assert rel_inf(got, per_block) <= 2e-6
    ud2 = torch.full((u.shape[0],), float("nan"), dtype=torch.float32, device="cuda")
    ibamd.residual_advection(dpart, ibamd.hip(u), ibamd.hip(C), out=ud2, flags=ibamd.IBH_PHASE_INTERIOR)
    ibamd.residual_advection(dpart, ibamd.hip(u), ibamd.hip(C), out=ud2, flags=ibamd.IBH_PHASE_BOUNDARY)
    assert np.array_equal(ibamd.to_host(ud2), got)


@pytest.mark.parametrize("case", ["advection", "rae"])
@pytest.mark.parametrize("kind", ["smooth", "step"])
def test_row_sweep(adv_mesh, rae_mesh_small, case, kind):
    """Row / column sweep (one wavefront per eight blocks, a lane per row then per column, arithmetic halo ids:
    csrc/ibh_rows2d.h; opt-in) on one-partition domains: against the oracle and against the quad sweep."""
    import torch
    from conftest import ADV_FAMILIES, RAE_FAMILIES
    from ibamd import _lib
    from oracle import residual_c as rc
    msh, fam = (adv_mesh, ADV_FAMILIES) if case == "advection" else (rae_mesh_small, RAE_FAMILIES)
    dom = ibamd.Domain(msh, hypercube_families=fam, max_partition_size=10 ** 9)
    (part,) = dom.partitions.values()
    dpart = ibamd.to_backend(part, ibamd.hip)
    assert dpart.info["row_sweep"]   # the library checked its id arithmetic against the halo table of every block
    u = seeded_field(part.centers, kind=kind)
    C = np.stack([np.ones_like(u), f32(0.5) + seeded_field(part.centers, seed=3) * f32(0.1)], axis=1)
    exp = rc.CPart(part).residual_advection(u, C)
    quad = ibamd.to_host(ibamd.residual_advection(dpart, ibamd.hip(u), ibamd.hip(C)))
    ud = torch.full((u.shape[0],), float("nan"), dtype=torch.float32, device="cuda")
    _lib.call("ibh_set_tuning", b"rows", 1)
    try:
        ibamd.residual_advection(dpart, ibamd.hip(u), ibamd.hip(C), out=ud)
        got = ibamd.to_host(ud)
        ud2 = torch.full((u.shape[0],), float("nan"), dtype=torch.float32, device="cuda")
        ibamd.residual_advection(dpart, ibamd.hip(u), ibamd.hip(C), out=ud2, flags=ibamd.IBH_PHASE_INTERIOR)
        ibamd.residual_advection(dpart, ibamd.hip(u), ibamd.hip(C), out=ud2, flags=ibamd.IBH_PHASE_BOUNDARY)
    finally:
        _lib.call("ibh_set_tuning", b"rows", 0)
    assert not np.isnan(got).any()
    assert rel_inf(got, exp) <= TOL
    assert rel_inf(got, quad) <= 2e-6
    assert np.array_equal(ibamd.to_host(ud2), got)
