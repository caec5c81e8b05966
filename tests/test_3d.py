"""3-D (octree, 8^3 blocks): builder vs literal restatement, and the face-list HIP kernels vs the oracle.
Small sphere-in-a-box case in the spirit of BASELINE.json configs[3]."""
import numpy as np
import pytest

import ibamd
from conftest import rel_inf
from ibamd.mesher import Ball, Mesh, Stereolitography

f32 = np.float32


def icosphere(radius=0.5, center=(0.0, 0.0, 0.0), subdiv=1):
    t = (1.0 + 5 ** 0.5) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
                  [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5],
                  [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]])
    for _ in range(subdiv):
        nv, nf, cache = list(v), [], {}

        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                nv.append((nv[a] + nv[b]) / 2)
                cache[k] = len(nv) - 1
            return cache[k]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        v, f = np.array(nv), np.array(nf)
    v = v / np.linalg.norm(v, axis=1, keepdims=True) * radius + np.asarray(center)
    return Stereolitography(v.T.astype(f32), (f.T + 1).astype(np.int64))


@pytest.fixture(scope="module")
def sphere_mesh():
    """253 blocks of 4^3 cells on 2 octree levels (16 192 cells), refined towards a ball.  No immersed
    surface: the reference's per-ghost triangle projection (mesher.jl:778-801 + :544-596) is a cold-path
    Python loop here and takes minutes in 3-D; hypercube boundaries are built and compared."""
    return Mesh(f32([-2, -2, -2]), f32([4, 4, 4]), block_size=4,
                refinement_regions=[(Ball(np.array([0.6, 0.6, 0.6]), 0.1), f32(0.2))])


@pytest.fixture(scope="module")
def sphere_domains(sphere_mesh):
    from oracle import domain as od
    kw = dict(hypercube_families=[("farfield", [(1, False), (1, True)])], max_partition_size=8192)
    return ibamd.Domain(sphere_mesh, **kw), od.Domain(sphere_mesh, **kw)


def test_3d_builder_matches_oracle(sphere_domains):
    dp, do = sphere_domains
    assert dp.ndims == 3 and len(dp) == 16192 and len(dp.partitions) == 2
    fd, fo, fn = dp.faces
    of = np.array(do.faces)
    assert np.array_equal(of[:, 0], fd) and np.array_equal(of[:, 1], fo) and np.array_equal(of[:, 2], fn)
    for k in do.partitions:
        a, b = do.partitions[k], dp.partitions[k]
        assert np.array_equal(a.domain, b.domain) and np.array_equal(a.image_in_domain, b.image_in_domain)
        for dim in (1, 2, 3):
            assert np.array_equal(a.face_owners_neighbors[dim][0], b.face_owners_neighbors[dim][0])
            assert np.array_equal(a.face_owners_neighbors[dim][1], b.face_owners_neighbors[dim][1])
            for r in (False, True):
                ia, _ = a.face_accumulators[(dim, r)].decompose()
                assert np.array_equal(np.concatenate([x for x in ia if x is not None and len(x)]),
                                      b.face_accumulators[(dim, r)].idx)
    for name in do.boundaries:
        for k in do.boundaries[name]:
            assert np.array_equal(do.boundaries[name][k].ghost_indices, dp.boundaries[name][k].ghost_indices)


@pytest.mark.gpu
def test_3d_residuals_match_oracle(sphere_domains):
    from oracle import cfd as ocfd
    from oracle import domain as od
    dp, do = sphere_domains
    rng = np.random.default_rng(5)
    fluid = ocfd.Fluid()
    for k in list(dp.partitions)[:2]:
        opart = do.partitions[k]
        dpart = ibamd.to_backend(dp.partitions[k], ibamd.hip)
        n = opart.spacing.shape[0]
        x = opart.centers
        u = (np.sin(2 * x[:, 0]) * np.cos(3 * x[:, 1]) + 0.3 * x[:, 2] + 0.1 * rng.uniform(-1, 1, n)).astype(f32)
        C = np.stack([np.ones(n, f32), f32(0.5) * np.ones(n, f32), f32(-0.25) * np.ones(n, f32)], axis=1)
        ud = np.zeros(n, f32)
        D = od.JST_sensor(opart, u)
        assert rel_inf(ibamd.to_host(ibamd.JST_sensor(dpart, ibamd.hip(u))), D) <= 1e-5
        for dim in (1, 2, 3):
            Cf = od.at_faces(opart, np.ascontiguousarray(C[:, dim - 1]), dim)
            gu = od.cell_gradient(opart, u, dim)
            assert rel_inf(ibamd.to_host(ibamd.cell_gradient(dpart, ibamd.hip(u), dim)), gu) <= 1e-5
            uL, uR = od.MUSCL(opart, u, gu, dim, D=D, high_order=True)
            ud -= od.green_gauss(opart, (uL + uR) * Cf / f32(2) + np.abs(Cf) * (uL - uR) / f32(2), dim)
        got = ibamd.to_host(ibamd.residual_advection(dpart, ibamd.hip(u), ibamd.hip(C)))
        assert rel_inf(got, ud) <= 1e-5
        P = np.empty((n, 5), dtype=f32)
        P[:, 0] = 1e5 * (1 + 0.05 * rng.uniform(-1, 1, n))
        P[:, 1] = 288.15 * (1 + 0.05 * rng.uniform(-1, 1, n))
        for d in range(3):
            P[:, 2 + d] = 100.0 * (1 + 0.1 * rng.uniform(-1, 1, n))
        R = np.zeros_like(P)
        Dp = od.JST_sensor(opart, np.ascontiguousarray(P[:, 0]))
        for dim in (1, 2, 3):
            gP = od.cell_gradient(opart, P, dim)
            PL, PR = od.MUSCL(opart, P, gP, dim, D=Dp, high_order=True)
            R -= od.green_gauss(opart, ocfd.inviscid_fluxes(fluid, PL, PR, dim), dim)
        got = ibamd.to_host(ibamd.residual_euler_hll(dpart, ibamd.hip(P)))
        assert rel_inf(got, R) <= 1e-5


from conftest import oracle_view as _oracle_view  # noqa: E402


@pytest.fixture(scope="module")
def octree8_mesh():
    """183 blocks of 8^3 cells on 2 levels (93 696 cells): exercises SAME, MIRROR, COARSE and (as face-list
    sides) FINE block sides of the 3-D block path."""
    return Mesh(f32([-2, -2, -2]), f32([4, 4, 4]), block_size=8,
                refinement_regions=[(Ball(np.array([1.2, 1.2, 1.2]), 0.1), f32(0.1))])


@pytest.mark.gpu
@pytest.mark.parametrize("nparts", [1, 2])
def test_3d_block_path(octree8_mesh, nparts):
    from oracle import domain as od
    msh = octree8_mesh
    n = len(msh)
    assert n == 183 * 512
    mps = -(-(-(-n // nparts)) // 512) * 512
    dom = ibamd.Domain(msh, max_partition_size=mps, boundaries=False)
    assert len(dom.partitions) == nparts
    rng = np.random.default_rng(11)
    for pid, part in dom.partitions.items():
        dpart = ibamd.to_backend(part, ibamd.hip)
        info = dpart.info
        assert info["full_blocks"] > 0 and info["sides_coarse"] > 0 and info["sides_mirror"] > 0
        assert info["sides_fine"] > 0
        if nparts == 1:
            assert info["sides_general"] == 0 and info["irregular_cells"] == 0
        x = part.centers
        nc = x.shape[0]
        u = (np.sin(2 * x[:, 0]) * np.cos(3 * x[:, 1]) + 0.3 * x[:, 2] + 0.1 * rng.uniform(-1, 1, nc)).astype(f32)
        C = np.stack([np.ones(nc, f32), f32(0.5) + f32(0.1) * rng.uniform(-1, 1, nc).astype(f32),
                      f32(-0.25) * np.ones(nc, f32)], axis=1)
        ud_, Cd = ibamd.hip(u), ibamd.hip(C)
        fast = ibamd.to_host(ibamd.residual_advection(dpart, ud_, Cd))
        gen = ibamd.to_host(ibamd.residual_advection(dpart, ud_, Cd, flags=ibamd.IBH_FORCE_GENERAL))
        assert rel_inf(fast, gen) <= 1e-5
        # cell_gradient(part, u), the tuple form: one block sweep for the three dimensions (ibh_cell_gradient_nd)
        g3 = ibamd.cell_gradient(dpart, ud_)
        assert len(g3) == 3
        for d in (1, 2, 3):
            assert rel_inf(ibamd.to_host(g3[d - 1]), ibamd.to_host(ibamd.cell_gradient(dpart, ud_, d))) <= 5e-6
        # Euler sweep (5 primitives): block path vs face-list path (Float64 HLL combine there), and vs the oracle
        from oracle import cfd as ocfd
        P = np.stack([f32(1e5) * (1 + f32(0.05) * rng.uniform(-1, 1, nc)), f32(288.15) * (1 + f32(0.05) * rng.uniform(-1, 1, nc)),
                      f32(100.0) * (1 + f32(0.1) * rng.uniform(-1, 1, nc)), f32(60.0) * (1 + f32(0.1) * rng.uniform(-1, 1, nc)),
                      f32(-40.0) * (1 + f32(0.1) * rng.uniform(-1, 1, nc))], axis=1).astype(f32)
        efast = ibamd.to_host(ibamd.residual_euler_hll(dpart, ibamd.hip(P)))
        egen = ibamd.to_host(ibamd.residual_euler_hll(dpart, ibamd.hip(P), flags=ibamd.IBH_FORCE_GENERAL))
        for v in range(5):
            assert rel_inf(efast[:, v], egen[:, v]) <= 1e-5, v
        if nparts == 1:   # `efast` is the single kernel (blk3::sweep_euler); the two-kernel form agrees with it
            etwo = ibamd.to_host(ibamd.residual_euler_hll(dpart, ibamd.hip(P), flags=ibamd.IBH_NO_FUSE))
            for v in range(5):
                assert rel_inf(etwo[:, v], egen[:, v]) <= 1e-5, v
                assert rel_inf(efast[:, v], etwo[:, v]) <= 5e-6, v
        if nparts == 1:
            op = _oracle_view(part)
            fluid = ocfd.Fluid()
            R = np.zeros_like(P)
            Dp = od.JST_sensor(op, np.ascontiguousarray(P[:, 0]))
            for dim in (1, 2, 3):
                gP = od.cell_gradient(op, P, dim)
                PL, PR = od.MUSCL(op, P, gP, dim, D=Dp, high_order=True)
                R -= od.green_gauss(op, ocfd.inviscid_fluxes(fluid, PL, PR, dim), dim)
            for v in range(5):
                assert rel_inf(efast[:, v], R[:, v]) <= 1e-5, v
            exp = np.zeros(nc, f32)
            D = od.JST_sensor(op, u)
            for dim in (1, 2, 3):
                Cf = od.at_faces(op, np.ascontiguousarray(C[:, dim - 1]), dim)
                gu = od.cell_gradient(op, u, dim)
                uL, uR = od.MUSCL(op, u, gu, dim, D=D, high_order=True)
                exp -= od.green_gauss(op, (uL + uR) * Cf / f32(2) + np.abs(Cf) * (uL - uR) / f32(2), dim)
            assert rel_inf(gen, exp) <= 1e-5
            assert rel_inf(fast, exp) <= 1e-5
            # `fast` is the single-kernel sweep (blk3::sweep_adv) on the eligible blocks + the two-kernel form on
            # the rest; the two-kernel form alone agrees with it
            assert info["fusable_blocks"] == info["full_blocks"] and info["workspace_blocks"] == 0
            assert info["rim4_rows"] > 0      # halo cells whose lateral neighbour is four finer cells are exercised
            two = ibamd.to_host(ibamd.residual_advection(dpart, ud_, Cd, flags=ibamd.IBH_NO_FUSE))
            assert rel_inf(two, exp) <= 1e-5
            assert rel_inf(fast, two) <= 5e-6
        else:
            # overlap phases reproduce the single sweep bit for bit, interior phase reads no skirt cell
            import torch
            skirt = np.ones(nc, bool)
            skirt[part.image_in_domain] = False
            up = u.copy()
            up[skirt] = np.nan
            out = torch.full((nc,), float("nan"), dtype=torch.float32, device="cuda")
            ibamd.residual_advection(dpart, ibamd.hip(up), Cd, out=out, flags=ibamd.IBH_PHASE_INTERIOR)
            got1 = ibamd.to_host(out)
            done = ~np.isnan(got1)
            assert np.array_equal(got1[done], fast[done])
            ibamd.residual_advection(dpart, ud_, Cd, out=out, flags=ibamd.IBH_PHASE_BOUNDARY)
            img = part.image_in_domain
            assert np.array_equal(ibamd.to_host(out)[img], fast[img])


def test_vectorised_surface_projection_matches_the_per_ghost_loop():
    """domain._project_3d (all ghost x candidate-triangle pairs at once, pinv once per triangle, exact pruning) against
    the literal per-ghost ``projection`` (mesher.jl:778-801 over proj2simplex :544-596) on an immersed sphere: same
    projections to the last bit or two of the pinv product, same ghost set after the distance test."""
    from ibamd import domain as D
    from ibamd.mesher import get_cells
    msh = Mesh(f32([-2, -2, -2]), f32([4, 4, 4]), ("sphere", icosphere(1.0, subdiv=2), f32(0.25)), block_size=4)
    c, w = get_cells(msh)
    df = msh.distance_fields["sphere"]
    ratio = f32(1.5)
    diams = np.sqrt(D._colsum(w * w))
    _, dists = df.nn(c)
    cand = np.nonzero(dists <= diams * ratio * f32(2))[0]
    assert cand.size > 500
    pick = cand[np.random.default_rng(1).choice(cand.size, 120, replace=False)]
    Xg, Rg = c[:, pick], diams[pick] * ratio * f32(2)
    Pv = D._project_3d(df, Xg, Rg)
    Pl = np.stack([df.projection(Xg[:, k], Rg[k]) for k in range(pick.size)], axis=1)
    assert np.abs(Pv - Pl).max() <= 4 * np.finfo(f32).eps * max(1.0, np.abs(Pl).max())
    dv, dl = D._dist_cols(Pv, Xg), D._dist_cols(Pl, Xg)
    assert np.array_equal(dv <= diams[pick] * ratio, dl <= diams[pick] * ratio)
    assert (dv <= diams[pick] * ratio).sum() > 10


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["one_block", "eight_blocks", "corner_refined"])
def test_3d_single_kernels_on_small_octrees(kind):
    """Edge cases of the 3-D single-kernel sweeps: a domain of ONE 8^3 block (six MIRROR sides), 2 x 2 x 2 uniform blocks
    (SAME + MIRROR), and a refinement in a corner of the box (FINE / COARSE sides on domain-boundary blocks, rim cells on
    the boundary).  Strip form, thread-per-cell form and the Euler kernel against the face-list kernels."""
    from ibamd import _lib
    f32_ = np.float32
    regions = {"one_block": [],
               "eight_blocks": [(Ball(np.array([0.0, 0.0, 0.0]), 5.0), f32_(0.3))],
               "corner_refined": [(Ball(np.array([-2.0, -2.0, -2.0]), 0.1), f32_(0.1))]}[kind]
    msh = Mesh(f32_([-2, -2, -2]), f32_([4, 4, 4]), block_size=8, refinement_regions=regions)
    assert len(msh) == {"one_block": 512, "eight_blocks": 4096, "corner_refined": 32768}[kind]
    dom = ibamd.Domain(msh, max_partition_size=10 ** 9, boundaries=False)
    (part,) = dom.partitions.values()
    dpart = ibamd.to_backend(part, ibamd.hip)
    info = dpart.info
    assert info["fusable_blocks"] == info["full_blocks"] == len(msh) // 512 and info["irregular_cells"] == 0
    if kind == "one_block":
        assert info["sides_mirror"] == 6
    if kind == "corner_refined":
        assert info["sides_fine"] > 0 and info["sides_coarse"] > 0
    rng = np.random.default_rng(2)
    x = part.centers
    n = x.shape[0]
    u = (np.sin(2 * x[:, 0]) * np.cos(3 * x[:, 1]) + 0.3 * x[:, 2] + 0.1 * rng.uniform(-1, 1, n)).astype(f32_)
    C = np.stack([np.ones(n, f32_), f32_(0.5) + f32_(0.1) * rng.uniform(-1, 1, n).astype(f32_), f32_(-0.25) * np.ones(n, f32_)],
                 axis=1)
    du, dC = ibamd.hip(u), ibamd.hip(C)
    gen = ibamd.to_host(ibamd.residual_advection(dpart, du, dC, flags=ibamd.IBH_FORCE_GENERAL))
    strip = ibamd.to_host(ibamd.residual_advection(dpart, du, dC))
    _lib.call("ibh_set_tuning", b"quad_variant", 512)
    try:
        cellk = ibamd.to_host(ibamd.residual_advection(dpart, du, dC))
    finally:
        _lib.call("ibh_set_tuning", b"quad_variant", 0)
    assert rel_inf(strip, gen) <= 1e-5 and rel_inf(cellk, gen) <= 1e-5 and rel_inf(strip, cellk) <= 2e-6
    P = np.stack([f32_(1e5) * (1 + f32_(0.05) * rng.uniform(-1, 1, n)), f32_(288.15) * (1 + f32_(0.05) * rng.uniform(-1, 1, n)),
                  f32_(100.0) * (1 + f32_(0.1) * rng.uniform(-1, 1, n)), f32_(60.0) * (1 + f32_(0.1) * rng.uniform(-1, 1, n)),
                  f32_(-40.0) * (1 + f32_(0.1) * rng.uniform(-1, 1, n))], axis=1).astype(f32_)
    one = ibamd.to_host(ibamd.residual_euler_hll(dpart, ibamd.hip(P)))
    egen = ibamd.to_host(ibamd.residual_euler_hll(dpart, ibamd.hip(P), flags=ibamd.IBH_FORCE_GENERAL))
    for v in range(5):
        assert rel_inf(one[:, v], egen[:, v]) <= 1e-5, v
    g3 = ibamd.cell_gradient(dpart, du)
    for d in (1, 2, 3):
        assert rel_inf(ibamd.to_host(g3[d - 1]), ibamd.to_host(ibamd.cell_gradient(dpart, du, d))) <= 5e-6


@pytest.mark.gpu
def test_3d_image_only_sweeps_on_partitions():
    """IBH_IMAGE_ONLY on 3-D partitions with skirt fragments (what a rank of a multi-GPU run sweeps,
    ImmersedBoundary.jl:610-619, 842-845): every image block qualifies for the single-kernel column sweeps -- halo, deeper
    and rim cells inside skirt fragments come from tables (csrc/ibh_analyze3_image.cpp) -- so the scalar and the Euler sweep
    are ONE launch over the image blocks, nothing is written outside the image, and the image cells agree with the C
    restatement of the oracle (oracle/residual_c.py) and with the two-kernel form over the whole partition."""
    import torch
    import bench
    from oracle import residual_c as rc
    msh = Mesh(f32([-4, -4, -4]), f32([8, 8, 8]), ("sphere", bench.icosphere(subdiv=2), f32(0.2)), block_size=8)
    msh.distance_fields = {}
    n = len(msh)
    nparts = 4
    mps = -(-(-(-n // nparts)) // 512) * 512
    IO = ibamd.IBH_IMAGE_ONLY
    rng = np.random.default_rng(7)
    used = 0
    for pid in (1, 3):
        dom = ibamd.Domain(msh, max_partition_size=mps, boundaries=False, only=[pid])
        part = dom.partitions[pid]
        dpart = ibamd.to_backend(part, ibamd.hip)
        info = dpart.info
        img = part.image_in_domain
        nc = part.centers.shape[0]
        assert nc > img.size                                   # there is a skirt
        assert info["image_blocks_all_eligible"] and info["image_blocks"] * 512 == img.size
        assert info["fusable_blocks"] == 0                     # (the all-cells sweep of this partition is the two-kernel form)
        used += 1
        x = part.centers
        u = (np.sin(2 * x[:, 0]) * np.cos(3 * x[:, 1]) + 0.3 * x[:, 2] + 0.1 * rng.uniform(-1, 1, nc)).astype(f32)
        C = np.stack([np.ones(nc, f32), f32(0.5) + f32(0.1) * rng.uniform(-1, 1, nc).astype(f32),
                      f32(-0.25) * np.ones(nc, f32)], axis=1)
        cpart = rc.CPart(part)
        exp = cpart.residual_advection(u, C)
        out = torch.full((nc,), float("nan"), dtype=torch.float32, device="cuda")
        ibamd.residual_advection(dpart, ibamd.hip(u), ibamd.hip(C), out=out, flags=IO)
        one = ibamd.to_host(out)
        assert np.isnan(one).sum() == nc - img.size            # nothing written outside the image
        assert rel_inf(one[img], exp[img]) <= 1e-5
        full = ibamd.to_host(ibamd.residual_advection(dpart, ibamd.hip(u), ibamd.hip(C)))
        assert rel_inf(one[img], full[img]) <= 2e-6
        P = np.stack([f32(1e5) * (1 + f32(0.05) * rng.uniform(-1, 1, nc)), f32(288.15) * (1 + f32(0.05) * rng.uniform(-1, 1, nc)),
                      f32(100.0) * (1 + f32(0.1) * rng.uniform(-1, 1, nc)), f32(60.0) * (1 + f32(0.1) * rng.uniform(-1, 1, nc)),
                      f32(-40.0) * (1 + f32(0.1) * rng.uniform(-1, 1, nc))], axis=1).astype(f32)
        expE = cpart.residual_euler(P)
        outE = torch.full((5, nc), float("nan"), dtype=torch.float32, device="cuda").T
        ibamd.residual_euler_hll(dpart, ibamd.hip(P), out=outE, flags=IO)
        oneE = ibamd.to_host(outE)
        assert np.isnan(oneE).sum() == 5 * (nc - img.size)
        for v in range(5):
            assert rel_inf(oneE[img, v], expE[img, v]) <= 1e-5, v
    assert used == 2
