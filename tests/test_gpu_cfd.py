"""CFD pointwise physics on the GPU vs the oracle (cfd.jl:62-151, 459-554, 664-736)."""
import numpy as np
import pytest

import ibamd
from conftest import rel_inf
from ibamd import cfd as gcfd
from oracle import cfd as ocfd

pytestmark = pytest.mark.gpu
f32 = np.float32


def _P(n, nd, seed=0):
    rng = np.random.default_rng(seed)
    P = np.empty((n, nd + 2), dtype=f32)
    P[:, 0] = 1e5 * (1 + 0.2 * rng.uniform(-1, 1, n))
    P[:, 1] = 288.15 * (1 + 0.2 * rng.uniform(-1, 1, n))
    P[:5, 1] = 5.0  # below the 10 K clamp
    for d in range(nd):
        P[:, 2 + d] = 300.0 * rng.uniform(-1, 1, n)  # sub- and supersonic, both signs
    return P


@pytest.mark.parametrize("nd", [2, 3])
def test_state_conversions_and_properties(nd):
    of, gf = ocfd.Fluid(), gcfd.Fluid()
    P = _P(4097, nd)
    Q = ocfd.primitive2state(of, P)
    assert rel_inf(ibamd.to_host(gcfd.primitive2state(gf, ibamd.hip(P))), Q) <= 1e-6
    assert rel_inf(ibamd.to_host(gcfd.state2primitive(gf, ibamd.hip(Q))), ocfd.state2primitive(of, Q)) <= 1e-6
    T = np.ascontiguousarray(P[:, 1])
    assert rel_inf(ibamd.to_host(gcfd.speed_of_sound(gf, ibamd.hip(T))), ocfd.speed_of_sound(of, T)) <= 1e-6
    assert rel_inf(ibamd.to_host(gcfd.dynamic_viscosity(gf, ibamd.hip(T))), ocfd.dynamic_viscosity(of, T)) <= 1e-5
    assert rel_inf(ibamd.to_host(gcfd.heat_conductivity(gf, ibamd.hip(T))), ocfd.heat_conductivity(of, T)) <= 1e-6


@pytest.mark.parametrize("nd", [2, 3])
def test_inviscid_fluxes(nd):
    of, gf = ocfd.Fluid(), gcfd.Fluid()
    PL, PR = _P(3000, nd, 1), _P(3000, nd, 2)
    rng = np.random.default_rng(3)
    nuL, nuR = rng.uniform(0, 1, 3000).astype(f32), rng.uniform(0, 1, 3000).astype(f32)
    for dim in range(1, nd + 1):
        exp = ocfd.inviscid_fluxes(of, PL, PR, dim)
        got = ibamd.to_host(gcfd.inviscid_fluxes(gf, ibamd.hip(PL), ibamd.hip(PR), dim))
        ok = np.isfinite(exp).all(axis=1)  # SL = SR = 0 gives 0/0 in the reference too
        assert ok.mean() > 0.95
        assert rel_inf(got[ok], exp[ok]) <= 1e-6
        exp = ocfd.inviscid_fluxes_sensor(of, PL, PR, nuL, nuR, dim)
        got = ibamd.to_host(gcfd.inviscid_fluxes(gf, ibamd.hip(PL), ibamd.hip(PR), ibamd.hip(nuL), ibamd.hip(nuR), dim))
        assert rel_inf(got, exp) <= 1e-6


@pytest.mark.parametrize("nd", [2, 3])
def test_viscous_fluxes(nd):
    of, gf = ocfd.Fluid(), gcfd.Fluid()
    P = _P(2000, nd, 4)
    rng = np.random.default_rng(5)
    Pgrad = tuple((rng.uniform(-1, 1, P.shape) * 1e3).astype(f32) for _ in range(nd))
    mu_t = rng.uniform(0, 1e-4, 2000).astype(f32)
    for dim in range(1, nd + 1):
        exp = ocfd.viscous_fluxes(of, P, Pgrad, dim, mu_t=mu_t)
        got = ibamd.to_host(gcfd.viscous_fluxes(gf, ibamd.hip(P), tuple(ibamd.hip(g) for g in Pgrad), dim,
                                                mu_t=ibamd.hip(mu_t)))
        assert rel_inf(got, exp) <= 1e-5
        exp0 = ocfd.viscous_fluxes(of, P, Pgrad, dim)
        got0 = ibamd.to_host(gcfd.viscous_fluxes(gf, ibamd.hip(P), tuple(ibamd.hip(g) for g in Pgrad), dim))
        assert rel_inf(got0, exp0) <= 1e-5


@pytest.mark.parametrize("nd", [2, 3])
def test_flow_bc(nd):
    """FlowBC call (cfd.jl:243-300): far field (sub/supersonic in/outflow), slip wall with transpiration, no-slip wall
    with the wall-function slip scaling -- exact selects, so bit-identical to the oracle."""
    of, gf = ocfd.Fluid(), gcfd.Fluid()
    n = 3001
    rng = np.random.default_rng(5)
    P = _P(n, nd, seed=3)
    nrm = rng.normal(size=(n, nd)).astype(f32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True).astype(f32)
    dP, dn = ibamd.hip(P), ibamd.hip(nrm)
    for uinf in ([250.0, 30.0, -10.0][:nd], [420.0, 0.0, 0.0][:nd]):   # subsonic and supersonic free stream
        state = [1.0e5, 288.15] + list(uinf)
        exp = ocfd.FlowBC(of, f32(state))(P, nrm)
        got = ibamd.to_host(gcfd.FlowBC(gf, state)(dP, dn))
        assert np.array_equal(got, exp)
    # slip wall: normal velocity imposed, with and without transpiration (scalar and per ghost)
    tr = (0.1 * rng.uniform(-1, 1, n)).astype(f32)
    for transp, dtr in ((f32(0.0), 0.0), (f32(0.25), 0.25), (tr, ibamd.hip(tr))):
        exp = ocfd.FlowBC(of, f32([1.0e5, 288.15, 0.0]), normal_flow=True)(P, nrm, transpiration=transp)
        got = ibamd.to_host(gcfd.FlowBC(gf, [1.0e5, 288.15, 0.0], normal_flow=True)(dP, dn, transpiration=dtr))
        assert rel_inf(got, exp) <= 1e-6
    # wall function: du/dn and image distances together
    imd = (1e-3 * (1 + rng.uniform(0, 1, n))).astype(f32)
    dudn = (1e4 * rng.uniform(0, 1, n)).astype(f32)
    exp = ocfd.FlowBC(of, f32([1.0e5, 288.15, 0.0]), normal_flow=True)(P, nrm, image_distances=imd, dudn=dudn)
    got = ibamd.to_host(gcfd.FlowBC(gf, [1.0e5, 288.15, 0.0], normal_flow=True)(
        dP, dn, image_distances=ibamd.hip(imd), du_dn=ibamd.hip(dudn)))
    assert rel_inf(got, exp) <= 1e-6
    with pytest.raises(ValueError):
        gcfd.FlowBC(gf, [1.0e5, 288.15, 0.0], normal_flow=True)(dP, dn, du_dn=ibamd.hip(dudn))


@pytest.mark.parametrize("nd", [2, 3])
def test_three_point_jst_and_shock_sensor(nd):
    """CFD.JST_sensor(Pim1, Pi, Pip1) and CFD.shock_sensor (cfd.jl:563-617) against the oracle."""
    from ibamd import cfd as gcfd
    rng = np.random.default_rng(nd)
    n = 5000
    a, b, c = (rng.uniform(0.5, 2, (n, 3)).astype(f32) for _ in range(3))
    b[::11] = a[::11]
    c[::11] = a[::11]          # flat spots: eps / eps = 1
    got = ibamd.to_host(gcfd.JST_sensor(ibamd.hip(a), ibamd.hip(b), ibamd.hip(c)))
    assert np.allclose(got, ocfd.JST_sensor3(a, b, c), rtol=2e-6, atol=0)
    g = [[rng.normal(size=n).astype(f32) for _ in range(nd)] for _ in range(nd)]
    got = ibamd.to_host(gcfd.shock_sensor([[ibamd.hip(x) for x in row] for row in g]))
    exp = ocfd.shock_sensor(g)
    assert np.allclose(got, exp, rtol=2e-6, atol=0) and (exp > 0).all() and (exp <= 1).all()


@pytest.mark.gpu
def test_viscous_residual_is_the_operator_composition(rae_domains):
    """``ibh_viscous_residual`` -- R .+= sum_d green_gauss(viscous_fluxes(fluid, at_faces(P, d), face_gradient(P, grad P, d), d;
    mu_t = at_faces(mu_t, d)), d) in one launch -- against the same expression composed from the operator kernels, on 2-D
    partitions with skirts (single faces through the side table, coarse-fine sides through the face lists) and on a 3-D
    octree partition: bit for bit."""
    import torch
    import bench
    from ibamd import cfd as gcfd
    from ibamd.mesher import Mesh
    from conftest import euler_field
    dp, _ = rae_domains
    cases = [(dp.partitions[k], 2) for k in (1, 3)]
    msh3 = Mesh(f32([-4, -4, -4]), f32([8, 8, 8]), ("sphere", bench.icosphere(subdiv=2), f32(0.4)), block_size=8)
    msh3.distance_fields = {}
    dom3 = ibamd.Domain(msh3, max_partition_size=10 ** 9)
    cases.append((dom3.partitions[1], 3))
    fluid = gcfd.Fluid()
    rng = np.random.default_rng(8)
    for part, nd in cases:
        dpart = ibamd.to_backend(part, ibamd.hip)
        nc = part.centers.shape[0]
        P = ibamd.hip(euler_field(part.centers, seed=3))
        mut = ibamd.hip((1e-4 * rng.uniform(0, 1, nc)).astype(f32))
        R0 = ibamd.hip(rng.uniform(-1, 1, (nc, nd + 2)).astype(f32))
        gP = ibamd.cell_gradient(dpart, P)
        ref = R0.clone()
        for d in range(1, nd + 1):
            Fv = gcfd.viscous_fluxes(fluid, ibamd.at_faces(dpart, P, d), ibamd.face_gradient(dpart, P, gP, d), d,
                                     mu_t=ibamd.at_faces(dpart, mut, d))
            ref += ibamd.green_gauss(dpart, Fv, d)
        got = R0.clone()
        gcfd.viscous_residual(dpart, fluid, P, gP, mut, got)
        assert not torch.equal(got, R0)
        assert torch.equal(got, ref), float((got - ref).abs().max())
        got2 = R0.clone()                       # the same from the gradients of the velocity columns alone
        gcfd.viscous_residual(dpart, fluid, P, ibamd.cell_gradient(dpart, P[:, 2:]), mut, got2, velocity_gradients_only=True)
        assert torch.equal(got2, ref)
        # the default kernel shares faces between the cells of a workgroup through LDS; one thread per cell with both faces
        # of every direction evaluated by it (the round-4 first form) must give the same bits
        from ibamd import _lib
        got3 = R0.clone()
        _lib.call("ibh_set_tuning", b"viscous_per_cell", 1)
        try:
            gcfd.viscous_residual(dpart, fluid, P, gP, mut, got3)
        finally:
            _lib.call("ibh_set_tuning", b"viscous_per_cell", 0)
        assert torch.equal(got3, ref)


def test_viscous_residual_of_a_parabolic_shear_flow(adv_mesh):
    """The analytic pin of tests/test_oracle_known_answers.py on the device: ``u = a y^2``, ``v = 0``, uniform ``p`` and ``T``
    through ``ibh_viscous_residual`` (the fused sum, with the gradients of the velocities from the shear-rate kernel's tuple
    form where it applies) -> ``2 a mu`` in the x-momentum row, 0 in the y-momentum row, wherever the spacing is uniform."""
    from ibamd import cfd as gcfd
    dom = ibamd.Domain(adv_mesh, max_partition_size=10 ** 9)
    (part,) = dom.partitions.values()
    dpart = ibamd.to_backend(part, ibamd.hip)
    X = part.centers
    n = X.shape[0]
    a = f32(3.0)
    P_h = np.empty((n, 4), dtype=f32)
    P_h[:, 0], P_h[:, 1] = 1.0e5, 300.0
    P_h[:, 2] = a * X[:, 1] * X[:, 1]
    P_h[:, 3] = 0.0
    fluid = gcfd.Fluid()
    mu = float(ibamd.to_host(gcfd.dynamic_viscosity(fluid, ibamd.hip(f32([300.0]))))[0])
    P = ibamd.hip(P_h)
    R = ibamd.hip(np.zeros((n, 4), dtype=f32))
    gcfd.viscous_residual(dpart, fluid, P, ibamd.cell_gradient(dpart, P[:, 2:]), ibamd.hip(np.zeros(n, f32)), R,
                          velocity_gradients_only=True)
    r = ibamd.to_host(R)
    exact = 2.0 * float(a) * mu
    inner = (np.abs(X[:, 0] - X[:, 0].mean()) < 0.35 * np.ptp(X[:, 0])) & (np.abs(X[:, 1] - X[:, 1].mean()) < 0.35 * np.ptp(X[:, 1]))
    rel = np.abs(r[inner, 2] - exact) / exact
    assert np.median(rel) <= 1e-5 and np.percentile(rel, 90) <= 1e-4, (np.median(rel), np.percentile(rel, 90))
    assert np.percentile(np.abs(r[inner, 3]), 90) <= 1e-4 * exact
    assert np.all(r[:, 0] == 0)
