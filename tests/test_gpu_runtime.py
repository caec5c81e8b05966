"""Partition runtime and ghost-cell BC through the GPU backend, against the oracle running the same
user closures: (dom::Domain)(f, args...) ImmersedBoundary.jl:820-864, impose_bc! :1197-1247, and the
explicit step of /root/reference/test/advection.jl:30-89 and test/dissipation.jl:52-83."""
import numpy as np
import pytest
import torch

import ibamd
from conftest import rel_inf, seeded_field
from oracle import domain as od

pytestmark = pytest.mark.gpu
f32 = np.float32
KW = dict(conv_to_backend=None, conv_from_backend=None)


def _gpu_kw():
    return dict(conv_to_backend=ibamd.hip, conv_from_backend=ibamd.to_host)


def test_domain_call_requires_both_converters(adv_domains):
    dp, _ = adv_domains
    u = np.zeros(len(dp), dtype=f32)
    with pytest.raises(AssertionError):
        dp(lambda part, u: None, u, conv_to_backend=ibamd.hip)
    with pytest.raises(TypeError):
        dp(lambda part, u: None, u)  # no CPU path


def test_advection_march_matches_oracle(adv_domains):
    """Three explicit steps of test/advection.jl (operator-granularity closure + BCs), 3 partitions."""
    dp, do = adv_domains
    n = len(dp)
    X = dp.global_centers()
    u0 = seeded_field(X, kind="step")
    C = np.ones((n, 2), dtype=f32)

    # ---- oracle
    def o_dt(part):
        return f32(0.5) / np.max(np.maximum(
            od.unsigned_green_gauss(part, od.at_faces(part, np.ones(part.spacing.shape[0], f32), 1), 1),
            od.unsigned_green_gauss(part, od.at_faces(part, np.ones(part.spacing.shape[0], f32), 2), 2)))

    def o_closure(part, u, ud, Cl):
        D = od.JST_sensor(part, u)
        for dim in (1, 2):
            Cf = od.at_faces(part, np.ascontiguousarray(Cl[:, dim - 1]), dim)
            gu = od.cell_gradient(part, u, dim)
            uL, uR = od.MUSCL(part, u, gu, dim, D=D, high_order=True)
            ud -= od.green_gauss(part, (uL + uR) * Cf / f32(2) + np.abs(Cf) * (uL - uR) / f32(2), dim)

    def o_bcs(u):
        od.impose_bc(lambda b, ui: f32(1.0), do, "upper", u)
        od.impose_bc(lambda b, ui: f32(0.0), do, "lower", u)
        od.impose_bc(lambda b, ui: ui.copy(), do, "outlet", u)

    # ---- GPU backend, same closures written against ibamd
    def g_dt(part):
        one = torch.ones(part.nc, dtype=torch.float32, device=part.spacing.device)
        a = ibamd.unsigned_green_gauss(part, ibamd.at_faces(part, one, 1), 1)
        b = ibamd.unsigned_green_gauss(part, ibamd.at_faces(part, one, 2), 2)
        return f32(0.5) / float(torch.maximum(a, b).max())

    def g_closure(part, u, ud, Cl):
        D = ibamd.JST_sensor(part, u)
        for dim in (1, 2):
            Cf = ibamd.at_faces(part, Cl[:, dim - 1].contiguous(), dim)
            gu = ibamd.cell_gradient(part, u, dim)
            uL, uR = ibamd.MUSCL(part, u, gu, dim, D=D, high_order=True)
            ud -= ibamd.green_gauss(part, (uL + uR) * Cf / 2 + torch.abs(Cf) * (uL - uR) / 2, dim)

    def g_bcs(u):
        ibamd.impose_bc(lambda b, ui: 1.0, dp, "upper", u, **_gpu_kw())
        ibamd.impose_bc(lambda b, ui: 0.0, dp, "lower", u, **_gpu_kw())
        ibamd.impose_bc(lambda b, ui: ui.clone(), dp, "outlet", u, **_gpu_kw())

    dt_o = min(do(o_dt)) * f32(0.75)
    dt_g = min(dp(g_dt, **_gpu_kw())) * f32(0.75)
    assert abs(dt_o - dt_g) <= 1e-6 * dt_o
    uo, ug = u0.copy(), u0.copy()
    for _ in range(3):
        udo, udg = np.zeros(n, f32), np.zeros(n, f32)
        do(o_closure, uo, udo, C)
        dp(g_closure, ug, udg, C, **_gpu_kw())
        assert rel_inf(udg, udo) <= 1e-5
        uo += udo * dt_o
        ug += udg * dt_o
        o_bcs(uo)
        g_bcs(ug)
        assert rel_inf(ug, uo) <= 1e-5


def test_dissipation_closure_nd_field(adv_domains):
    """test/dissipation.jl:69-77: 2-component field, Laplacian via green_gauss(face_gradient)."""
    dp, do = adv_domains
    n = len(dp)
    uv = seeded_field(dp.global_centers(), nv=2)

    def o_closure(part, uv, uvd):
        for dim in (1, 2):
            uvd += od.green_gauss(part, od.face_gradient(part, uv, dim), dim)

    def g_closure(part, uv, uvd):
        for dim in (1, 2):
            uvd += ibamd.green_gauss(part, ibamd.face_gradient(part, uv, dim), dim)

    a, b = np.zeros((n, 2), f32), np.zeros((n, 2), f32)
    do(o_closure, uv.copy(), a)
    dp(g_closure, uv.copy(), b, **_gpu_kw())
    assert rel_inf(b, a) <= 1e-5

    # BC closures returning arrays (dissipation.jl:30-50) on a device-resident global array
    uvo = uv.copy()
    uvg = ibamd.hip(uv)

    def o_bc(bdry, ia):
        r = np.empty_like(ia)
        r[:] = np.array([1.0, 0.0], dtype=f32)
        return r

    def g_bc(bdry, ia):
        r = torch.empty_like(ia)
        r[:] = torch.tensor([1.0, 0.0], device=ia.device)
        return r
    od.impose_bc(o_bc, do, "upper", uvo)
    ibamd.impose_bc(g_bc, dp, "upper", uvg)
    od.impose_bc(lambda bd, ia: ia.copy(), do, "outlet", uvo)
    ibamd.impose_bc(lambda bd, ia: ia.clone(), dp, "outlet", uvg)
    assert rel_inf(ibamd.to_host(uvg), uvo) <= 1e-5


def test_fused_bc_modes(adv_domains):
    """ibh_bc_apply: fused interp+blend for the Dirichlet-constant and copy closures."""
    import ctypes as C
    from ibamd import backend as B
    dp, do = adv_domains
    u = seeded_field(dp.global_centers())
    for name, mode, const in (("upper", 0, 1.0), ("outlet", 1, 0.0)):
        uo = u.copy()
        od.impose_bc((lambda b, ia: f32(const)) if mode == 0 else (lambda b, ia: ia.copy()), do, name, uo)
        ug = ibamd.hip(u)
        for k in dp.boundaries[name]:
            bd = ibamd.to_backend(dp.boundaries[name][k])
            c = np.array([const], dtype=f32)
            B._stream()
            B.call("ibh_bc_apply", bd.handle, B._ptr(ug), 1, ug.shape[0], mode, B._hptr(c))
        assert rel_inf(ibamd.to_host(ug), uo) <= 1e-6


def test_multigrid_transfer_operators(adv_mesh_coarse):
    """Coarsener / prolongator application = Accumulator SpMV (ImmersedBoundary.jl:1391-1392)."""
    from oracle import nninterp
    dom = ibamd.Domain(adv_mesh_coarse, hypercube_families=[("outlet", [(1, True), (2, True)])])
    coarse_doms, prolongators, coarseners = ibamd.multigrid(dom, max_levels=2)
    u = seeded_field(dom.global_centers(), nv=3)
    uc = ibamd.to_backend(coarseners[0])(ibamd.hip(u))
    bs = dom.mesh.block_size
    mean = u.reshape(-1, bs // 2, 2, bs // 2, 2, 3).mean(axis=(2, 4)).reshape(-1, 3)
    assert rel_inf(ibamd.to_host(uc), mean) <= 1e-6
    up = ibamd.to_backend(prolongators[0])(uc)
    assert up.shape == (len(dom), 3)
    # prolongation of a constant is that constant (IDW weights sum to 1)
    one = torch.ones(len(coarse_doms[0]), dtype=torch.float32, device=uc.device)
    assert float((ibamd.to_backend(prolongators[0])(one) - 1).abs().max()) <= 1e-6


def test_fas_multigrid_matches_oracle(adv_mesh_coarse):
    """Solver.FAS! (solver.jl:39-91) on a 3-level hierarchy: device-resident loop vs the oracle's loop with
    the same residual closure (pseudo-time relaxation of the Laplacian, test/dissipation.jl:69-77)."""
    from oracle.solver import FAS as oFAS
    fam = [("neumann", [(1, True), (2, True)])]
    dp = ibamd.Domain(adv_mesh_coarse, hypercube_families=fam, boundaries=False)
    do = od.Domain(adv_mesh_coarse, hypercube_families=fam)
    cd_p, prol_p, coar_p = ibamd.multigrid(dp, max_levels=2)
    cd_o, prol_o, coar_o = od.multigrid(do, max_levels=2)
    levels_p = [dp] + cd_p
    levels_o = [do] + cd_o
    Q0 = seeded_field(dp.global_centers(), nv=2)

    def o_f(l, Q):
        part = levels_o[l].partitions[1]
        r = np.zeros_like(Q)
        for dim in (1, 2):
            r += od.green_gauss(part, od.face_gradient(part, Q, dim), dim)
        h = part.spacing[:, 0].min()
        return r, f32(0.2) * h * h

    dparts = [ibamd.to_backend(d.partitions[1], ibamd.hip) for d in levels_p]

    def g_f(l, Q):
        part = dparts[l]
        r = torch.zeros_like(Q)
        for dim in (1, 2):
            r += ibamd.green_gauss(part, ibamd.face_gradient(part, Q, dim), dim)
        h = float(levels_p[l].partitions[1].spacing[:, 0].min())
        return r, f32(0.2) * f32(h) * f32(h)

    Qo = Q0.copy()
    ro = oFAS(o_f, Qo, coarseners=coar_o, prolongators=prol_o, n_iter=8, rtol=f32(1e-3))
    Qg = ibamd.hip(Q0)
    rg = ibamd.FAS(g_f, Qg, coarseners=coar_p, prolongators=prol_p, n_iter=8, rtol=1e-3)
    assert rel_inf(ibamd.to_host(Qg), Qo) <= 1e-5
    assert abs(rg - float(ro)) <= 1e-4 * max(1.0, float(ro))
    assert not np.array_equal(Qo, Q0)
    # check_every: the convergence test (the host round trip of an iteration) every k iterations; with a tolerance that
    # is never met the iterates are the same and the norm is read 1 + 1 times per level instead of 1 + n_iter
    from ibamd import solver as _solver
    calls = {"n": 0}

    def counting_norm(r):
        calls["n"] += 1
        return _solver._norm(r)
    Q1, Qk = ibamd.hip(Q0), ibamd.hip(Q0)
    r1 = ibamd.FAS(g_f, Q1, coarseners=coar_p, prolongators=prol_p, n_iter=8, rtol=0.0, atol=0.0, norm=counting_norm)
    n1, calls["n"] = calls["n"], 0
    rk = ibamd.FAS(g_f, Qk, coarseners=coar_p, prolongators=prol_p, n_iter=8, rtol=0.0, atol=0.0, norm=counting_norm,
                   check_every=8)
    assert np.array_equal(ibamd.to_host(Q1), ibamd.to_host(Qk)) and abs(r1 - rk) <= 1e-9 * r1   # (atomic sum order)
    assert n1 == 2 * 9 and calls["n"] == 2 * 2      # two levels are visited (the last supplied level never is)


def test_flow_bc_in_impose_bc(rae_domains):
    """FlowBC closures (cfd.jl:160-300) inside impose_bc! on the immersed wall and the far field of the RAE2822
    case, device-resident, against the oracle's impose_bc with the oracle's FlowBC."""
    from ibamd import cfd as gcfd
    from oracle import cfd as ocfd
    dp, do = rae_domains
    n = len(dp)
    rng = np.random.default_rng(8)
    P = np.empty((n, 4), dtype=f32)
    P[:, 0] = 1e5 * (1 + 0.05 * rng.uniform(-1, 1, n))
    P[:, 1] = 288.15 * (1 + 0.05 * rng.uniform(-1, 1, n))
    P[:, 2] = 230.0 * (1 + 0.1 * rng.uniform(-1, 1, n))
    P[:, 3] = 20.0 * rng.uniform(-1, 1, n)
    Po, Pg = P.copy(), ibamd.hip(P)
    far = [1.0e5, 288.15, 230.0, 10.0]
    o_far, g_far = ocfd.FlowBC(ocfd.Fluid(), f32(far)), gcfd.FlowBC(gcfd.Fluid(), far)
    o_wall = ocfd.FlowBC(ocfd.Fluid(), f32([1.0e5, 288.15, 0.0]), normal_flow=True)
    g_wall = gcfd.FlowBC(gcfd.Fluid(), [1.0e5, 288.15, 0.0], normal_flow=True)
    od.impose_bc(lambda b, ia: o_far(ia, b.normals), do, "farfield", Po)
    ibamd.impose_bc(lambda b, ia: g_far(ia, b.normals), dp, "farfield", Pg)
    od.impose_bc(lambda b, ia: o_wall(ia, b.normals), do, "wall", Po)
    ibamd.impose_bc(lambda b, ia: g_wall(ia, b.normals), dp, "wall", Pg)
    got = ibamd.to_host(Pg)
    assert not np.array_equal(got, P)
    for v in range(4):
        assert rel_inf(got[:, v], Po[:, v]) <= 1e-5


def test_partition_pack_feeds_the_library(adv_domains, tmp_path):
    """A Partition loaded from its pack (ibamd.pack) gives the same block analysis and the same sweep, bit for bit."""
    from ibamd.pack import load_partition, save_partition
    dp, _ = adv_domains
    part = dp.partitions[2]
    path = str(tmp_path / "p.npz")
    save_partition(path, part)
    back, _ = load_partition(path)
    a, b = ibamd.to_backend(part, ibamd.hip), ibamd.to_backend(back, ibamd.hip)
    assert a.info == b.info
    u = seeded_field(part.centers)
    C = np.ones((u.shape[0], 2), dtype=f32)
    ra = ibamd.to_host(ibamd.residual_advection(a, ibamd.hip(u), ibamd.hip(C)))
    rb = ibamd.to_host(ibamd.residual_advection(b, ibamd.hip(u), ibamd.hip(C)))
    assert np.array_equal(ra, rb)


def test_surface_and_volume_integrals(rae_domains):
    """``surf(u)``, ``at_offset``, ``surface_integral`` and ``volume_integral`` (ImmersedBoundary.jl:345-376,
    :1415-1431) on the device against the oracle."""
    dp, do = rae_domains
    so, sp = do.surfaces["wall"], dp.surfaces["wall"]
    X = dp.global_centers()
    n = len(dp)
    rng = np.random.default_rng(5)
    U = np.stack([np.sin(3 * X[:, 0]) * np.cos(2 * X[:, 1]), 1e5 * (1 + 0.1 * rng.uniform(-1, 1, n))], axis=1).astype(f32)
    ds = ibamd.to_backend(sp, ibamd.hip)
    got = ibamd.to_host(ds(ibamd.hip(U)))
    exp = so(U)
    assert got.shape == exp.shape and rel_inf(got, exp) <= 1e-5
    got_off = ibamd.to_host(ibamd.at_offset(sp, ibamd.hip(U)))
    assert rel_inf(got_off, od.at_offset(so, U)) <= 1e-5
    si = ibamd.surface_integral(sp, ds(ibamd.hip(U)))
    sio = od.surface_integral(so, exp)
    assert si.shape == (2,) and np.abs(si - sio).max() <= 1e-5 * np.abs(sio).max()
    s1 = ibamd.surface_integral(sp, ibamd.hip(np.ones(so.points.shape[0], f32)))
    assert abs(float(s1) - float(od.surface_integral(so, np.ones(so.points.shape[0], f32)))) <= 1e-5 * float(s1)
    vi = ibamd.volume_integral(dp, ibamd.hip(U))
    vio = od.volume_integral(do, U)
    from ibamd.mesher import get_cells
    _, widths = get_cells(dp.mesh)
    exact = (U.astype(np.float64) * np.prod(widths.astype(np.float64), axis=0)[:, None]).sum(axis=0)
    # Float32 sums of 3.7e4 terms: the device's tree reduction is the closer one to the Float64 value; the oracle's
    # running Float32 sum (like the reference's) carries ~1e-4
    assert np.abs(vi - exact).max() <= 1e-5 * np.abs(exact).max()
    assert np.abs(vio - exact).max() <= 1e-3 * np.abs(exact).max()
    assert abs(float(ibamd.volume_integral(dp, ibamd.hip(np.ones(n, f32)))) - 2500.0) <= 0.5   # test/rae2822.jl:24-29


def test_advection_march_through_the_fused_step(adv_mesh):
    """``march!`` of test/advection.jl:61-89 through ``ibh_step_advection``: time step by a device reduction
    (``ibh_timestep_advection``), sweep + ``u .+= ud .* dt`` in ONE launch, the three ``impose_bc!`` calls of :30-46 as a
    ``BCSet`` -- no host read-back in the loop -- against the oracle's operator-by-operator march on the same one-partition
    domain; the BC set alone against ``impose_bc``; the unfused fallback (``IBH_QUAD=0`` path: sweep, update) agrees."""
    from conftest import ADV_FAMILIES
    from ibamd import _lib
    kw = dict(hypercube_families=ADV_FAMILIES, max_partition_size=10 ** 9)
    dp, do = ibamd.Domain(adv_mesh, **kw), od.Domain(adv_mesh, **kw)
    (part,) = dp.partitions.values()
    (opart,) = do.partitions.values()
    dpart = ibamd.to_backend(part, ibamd.hip)
    n = len(dp)
    u0 = seeded_field(dp.global_centers(), kind="step")
    Ch = np.ones((n, 2), dtype=f32)

    def o_closure(p, u, ud, Cl):
        D = od.JST_sensor(p, u)
        for dim in (1, 2):
            Cf = od.at_faces(p, np.ascontiguousarray(Cl[:, dim - 1]), dim)
            gu = od.cell_gradient(p, u, dim)
            uL, uR = od.MUSCL(p, u, gu, dim, D=D, high_order=True)
            ud -= od.green_gauss(p, (uL + uR) * Cf / f32(2) + np.abs(Cf) * (uL - uR) / f32(2), dim)
    one = np.ones(n, f32)
    dt_o = f32(0.5) / np.max(np.maximum(od.unsigned_green_gauss(opart, od.at_faces(opart, one, 1), 1),
                                        od.unsigned_green_gauss(opart, od.at_faces(opart, one, 2), 2))) * f32(0.75)
    C = ibamd.hip(Ch)
    dt = ibamd.timestep_advection(dpart, C, scale=0.75)
    assert abs(float(dt.item()) - float(dt_o)) <= 1e-6 * float(dt_o)
    bcs = ibamd.BCSet(dp, [("upper", 1.0), ("lower", 0.0), ("outlet", "copy")])
    assert bcs.n_ghost > 0
    # the BC set alone: the three sequential impose_bc calls
    a1, a2 = ibamd.hip(u0), ibamd.hip(u0)
    bcs.apply(a1)
    ibamd.impose_bc(lambda b, ui: 1.0, dp, "upper", a2)
    ibamd.impose_bc(lambda b, ui: 0.0, dp, "lower", a2)
    ibamd.impose_bc(lambda b, ui: ui.clone(), dp, "outlet", a2)
    assert torch.equal(a1, a2)
    assert bcs.healthy()
    # the march
    uo = u0.copy()
    ua, ub = ibamd.hip(u0), torch.empty(n, dtype=torch.float32, device="cuda")
    for _ in range(4):
        udo = np.zeros(n, f32)
        do(o_closure, uo, udo, Ch)
        uo += udo * dt_o
        od.impose_bc(lambda b, ui: f32(1.0), do, "upper", uo)
        od.impose_bc(lambda b, ui: f32(0.0), do, "lower", uo)
        od.impose_bc(lambda b, ui: ui.copy(), do, "outlet", uo)
        ibamd.timestep_advection(dpart, C, scale=0.75, out=dt)      # (advection.jl recomputes it every step)
        ibamd.step_advection(dpart, ua, C, dt, bcs, out=ub)
        ua, ub = ub, ua
        assert rel_inf(ibamd.to_host(ua), uo) <= 1e-5
    assert not np.array_equal(uo, u0)
    # unfused form of the same step: sweep, update with the device dt, BC set
    v = ibamd.hip(u0)
    r = ibamd.residual_advection(dpart, v, C)
    w = torch.empty_like(v)
    _lib.call("ibh_update_dev", n, _lib.c_vp(dt.data_ptr()), _lib.c_vp(v.data_ptr()), _lib.c_vp(r.data_ptr()),
              _lib.c_vp(w.data_ptr()))
    bcs.apply(w)
    f = ibamd.step_advection(dpart, ibamd.hip(u0), C, dt, bcs)
    assert rel_inf(ibamd.to_host(f), ibamd.to_host(w)) <= 1e-6
    assert bcs.healthy() and bcs.n_levels >= 1


@pytest.mark.parametrize("sets", ["three", "one_direct", "none"])
def test_step_with_the_next_time_step_beside_the_bc_set(adv_mesh, sets):
    """``ibh_step_advection_dt``: the step with the time step of the NEXT step evaluated by extra workgroups of the BC set's
    own launches (it depends on ``C`` alone) -- against ``timestep_advection`` + ``step_advection`` as separate launches: the
    same ``dt`` and the same field, bit for bit, over several steps with a non-uniform ``C``; BC sets with several launches,
    with a single launch (the final reduction falls back to its own launch) and without boundary conditions."""
    from conftest import ADV_FAMILIES
    kw = dict(hypercube_families=ADV_FAMILIES, max_partition_size=10 ** 9)
    dp = ibamd.Domain(adv_mesh, **kw)
    (part,) = dp.partitions.values()
    dpart = ibamd.to_backend(part, ibamd.hip)
    n = len(dp)
    X = dp.global_centers()
    u0 = seeded_field(X, kind="step")
    Ch = np.stack([1.0 + 0.3 * np.sin(3 * X[:, 1]), 0.8 + 0.2 * np.cos(2 * X[:, 0])], axis=1).astype(f32)
    C = ibamd.hip(Ch)
    bcs = {"three": lambda: ibamd.BCSet(dp, [("upper", 1.0), ("lower", 0.0), ("outlet", "copy")]),
           "one_direct": lambda: ibamd.BCSet(dp, [("upper", 1.0)]), "none": lambda: None}[sets]()
    # separate launches
    dt_a = ibamd.timestep_advection(dpart, C, scale=0.75)
    ua, ub = ibamd.hip(u0), torch.empty(n, dtype=torch.float32, device="cuda")
    # dt beside the BC set: the first dt from the plain call, every later one from the step before
    dt_b = ibamd.timestep_advection(dpart, C, scale=0.75)
    va, vb = ibamd.hip(u0), torch.empty(n, dtype=torch.float32, device="cuda")
    for k in range(4):
        ibamd.timestep_advection(dpart, C, scale=0.75, out=dt_a)
        ibamd.step_advection(dpart, ua, C, dt_a, bcs, out=ub)
        ua, ub = ub, ua
        dt_b_before = float(dt_b.item())
        ibamd.step_advection(dpart, va, C, dt_b, bcs, out=vb, next_dt=dt_b, scale=0.75)
        va, vb = vb, va
        assert float(dt_b.item()) == float(dt_a.item()) == dt_b_before
        assert torch.equal(ua, va), k
    assert not torch.equal(ua, ibamd.hip(u0))
    if bcs is not None:
        assert bcs.healthy()
