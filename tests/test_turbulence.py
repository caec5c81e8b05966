"""Turbulence closures (reference: /root/reference/src/turbulence.jl): oracle known answers on the CPU, device
kernels against the oracle on the GPU."""
import numpy as np
import pytest

from oracle import turbulence as ot

f32 = np.float32


def _grads(n, nd, seed=0):
    rng = np.random.default_rng(seed)
    return [[rng.uniform(-50, 50, n).astype(f32) for _ in range(nd)] for _ in range(nd)]


def test_oracle_known_answers():
    # law of the wall: viscous sublayer u+ = y+ (Rey = y+^2), log layer u+ = ln(y+)/kappa + C
    lam = ot.wall_function_rey(f32([1.0, 4.0, 9.0]))
    assert np.allclose(lam["yplus"], [1.0, 2.0, 3.0], rtol=1e-5) and np.allclose(lam["uplus"], lam["yplus"], rtol=1e-5)
    yp = f32(200.0)
    up = np.log(yp) / f32(0.41) + f32(4.9)
    log = ot.wall_function_rey(f32([yp * up]), n_iter=200)
    assert abs(log["yplus"][0] - yp) <= 1e-2 * yp
    # pure shear du/dy = g: S = |g|; solid-body rotation: S = 0 and the Ducros sensor -> 0; pure dilatation -> 1
    n = 7
    g = f32(3.0) * np.ones(n, f32)
    z = np.zeros(n, f32)
    assert np.allclose(ot.shear_rate([[z, g], [z, z]]), 3.0)
    assert np.allclose(ot.shear_rate([[z, g], [-g, z]]), 0.0)
    assert np.all(ot.Ducros_sensor([[z, g], [-g, z]]) < 1e-6)
    assert np.allclose(ot.Ducros_sensor([[g, z], [z, g]]), 1.0)
    # k-epsilon: nu_t = Cmu k^2 / eps, equilibrium Pk = eps gives Sk = 0
    k, e = f32([2.0]), f32([0.5])
    ke = ot.standard_k_epsilon(k, e, np.sqrt(e / (f32(0.09) * k ** 2 / e)))
    assert np.allclose(ke["nut"], 0.09 * 4 / 0.5) and abs(ke["Sk"][0]) < 1e-5
    # Wray-Agarwal: no gradients -> S = C1 R S, capped at 10 R
    wa = ot.Wray_Agarwal(f32([1.0, 1.0]), f32([2.0, 1000.0]), np.zeros((2, 2), f32), np.zeros((2, 2), f32))
    assert np.allclose(wa["S"], [0.0829 * 2.0, 10.0], rtol=1e-6) and np.allclose(wa["nuR"], 0.72)


@pytest.mark.gpu
@pytest.mark.parametrize("nd", [2, 3])
def test_gpu_closures_match_oracle(nd):
    import ibamd
    from ibamd import turbulence as gt
    from conftest import rel_inf
    n = 5003
    rng = np.random.default_rng(nd)
    g = _grads(n, nd, seed=nd)
    dg = [[ibamd.hip(a) for a in row] for row in g]
    assert rel_inf(ibamd.to_host(gt.shear_rate(dg)), ot.shear_rate(g)) <= 1e-6
    assert rel_inf(ibamd.to_host(gt.Ducros_sensor(dg)), ot.Ducros_sensor(g)) <= 1e-5
    Delta = rng.uniform(1e-3, 1e-1, n).astype(f32)
    S = ot.shear_rate(g)
    assert rel_inf(ibamd.to_host(gt.Smagorinsky_nuSGS(ibamd.hip(Delta), ibamd.hip(S))), ot.Smagorinsky_nuSGS(Delta, S)) <= 1e-6
    if nd == 3:
        assert rel_inf(ibamd.to_host(gt.WALE_nuSGS(ibamd.hip(Delta), dg)), ot.WALE_nuSGS(Delta, g)) <= 1e-5
    k = rng.uniform(0.1, 5, n).astype(f32)
    e = rng.uniform(0.1, 5, n).astype(f32)
    gke, oke = gt.standard_k_epsilon(ibamd.hip(k), ibamd.hip(e), ibamd.hip(S)), ot.standard_k_epsilon(k, e, S)
    for key in oke:
        assert rel_inf(ibamd.to_host(gke[key]), oke[key]) <= 1e-5, key
    R = rng.uniform(1e-6, 1e-3, n).astype(f32)
    gR = rng.uniform(-1, 1, (n, nd)).astype(f32)
    gS = rng.uniform(-100, 100, (n, nd)).astype(f32)
    gwa = gt.Wray_Agarwal(ibamd.hip(R), ibamd.hip(S), ibamd.hip(gR), ibamd.hip(gS))
    owa = ot.Wray_Agarwal(R, S, gR, gS)
    for key in owa:
        assert rel_inf(ibamd.to_host(gwa[key]), owa[key]) <= 1e-5, key


@pytest.mark.gpu
def test_gpu_wall_function_matches_oracle():
    import ibamd
    from ibamd import turbulence as gt
    from conftest import rel_inf
    n = 4099
    rng = np.random.default_rng(4)
    Rey = (10.0 ** rng.uniform(-3, 6, n)).astype(f32)
    Rey[:3] = [0.0, -5.0, 1e-12]
    gw, ow = gt.wall_function(ibamd.hip(Rey)), ot.wall_function_rey(Rey)
    for key in ow:
        assert rel_inf(ibamd.to_host(gw[key]), ow[key]) <= 1e-5, key
    y = (10.0 ** rng.uniform(-6, -2, n)).astype(f32)
    u = rng.uniform(0.1, 300, n).astype(f32)
    nu = np.full(n, 1.5e-5, f32)
    gw3, ow3 = gt.wall_function(ibamd.hip(y), ibamd.hip(u), ibamd.hip(nu)), ot.wall_function(y, u, nu)
    for key in ow3:
        assert rel_inf(ibamd.to_host(gw3[key]), ow3[key]) <= 2e-5, key


@pytest.mark.gpu
def test_scalar_transport_is_the_operator_composition(rae_domains):
    """``ibh_scalar_transport`` -- S + sum_d green_gauss(at_faces(nu + nuR) .* face_gradient(R) .- at_faces(u_d .* R)) in
    one launch -- against the same expression composed from the operator kernels (2-D partitions with skirts; the 3-D
    case is tests/test_config5.py through the closure): bit for bit."""
    import torch
    import ibamd
    from ibamd import turbulence as T
    dp, _ = rae_domains
    rng = np.random.default_rng(4)
    for k in (1, 3):
        part = dp.partitions[k]
        dpart = ibamd.to_backend(part, ibamd.hip)
        nc = part.centers.shape[0]
        R = ibamd.hip((4.5e-5 * (1 + 0.5 * rng.uniform(0, 1, nc))).astype(f32))
        nuR = ibamd.hip((1e-5 * rng.uniform(0.5, 2, nc)).astype(f32))
        S = ibamd.hip(rng.uniform(-1, 1, nc).astype(f32))
        vel = ibamd.hip(np.stack([100 * (1 + 0.1 * rng.uniform(-1, 1, nc)), 10 * rng.uniform(-1, 1, nc)], axis=1).astype(f32))
        nu = 1.5e-5
        got = T.scalar_transport(dpart, R, nuR, vel, nu, S)
        rt = S.clone()
        for d in range(2):
            conv = ibamd.at_faces(dpart, vel[:, d].contiguous() * R, d + 1)
            diff = ibamd.at_faces(dpart, float(nu) + nuR, d + 1) * ibamd.face_gradient(dpart, R, d + 1)
            rt += ibamd.green_gauss(dpart, diff - conv, d + 1)
        assert torch.equal(got, rt)


@pytest.mark.gpu
def test_fused_closures_on_an_all_block_octree_are_the_composition():
    """``ibh_shear_rate_of_velocity`` / ``ibh_wray_agarwal_of`` (gradients consumed inside the block sweep, 3-D partition of
    complete 8^3 blocks with SAME / MIRROR / COARSE / FINE sides) against the same closures composed from
    ``cell_gradient`` + the pointwise kernels -- bit for bit -- and, on a partition with skirt fragments (where the fused
    kernels do not apply), that the wrappers take the composition."""
    import torch
    import ibamd
    from ibamd import Ball, Mesh
    from ibamd import turbulence as T
    msh = Mesh(f32([-2, -2, -2]), f32([4, 4, 4]), block_size=8,
               refinement_regions=[(Ball(np.array([1.2, 1.2, 1.2]), 0.1), f32(0.1))])
    n = len(msh)
    rng = np.random.default_rng(21)
    for nparts in (1, 2):
        mps = -(-(-(-n // nparts)) // 512) * 512
        dom = ibamd.Domain(msh, max_partition_size=mps, boundaries=False)
        part = dom.partitions[1]
        dpart = ibamd.to_backend(part, ibamd.hip)
        assert T.all_blocks(dpart) == (nparts == 1)
        nc = part.centers.shape[0]
        X = part.centers
        vel_h = np.stack([np.sin(2 * X[:, 0]) * np.cos(X[:, 1]), np.cos(X[:, 2] + X[:, 0]), X[:, 1] * X[:, 2]],
                         axis=1).astype(f32) + f32(0.05) * rng.standard_normal((nc, 3)).astype(f32)
        vel = ibamd.hip(vel_h)
        R = ibamd.hip((4.5e-5 * (1 + 0.5 * rng.uniform(0, 1, nc))).astype(f32))
        S_fused = T.shear_rate_of_velocity(dpart, vel)
        gu = [list(ibamd.cell_gradient(dpart, vel[:, i].contiguous())) for i in range(3)]
        S_comp = T.shear_rate(gu)
        assert torch.equal(S_fused, S_comp)
        assert float(S_comp.abs().max()) > 0
        # gradients=True: the same S and cell_gradient(part, vel) -- the tuple over the dimensions -- made on the way
        # (ibh_shear_rate_of_velocity_grad on the all-block partition, the composition on the other)
        S_g, gV = T.shear_rate_of_velocity(dpart, vel, gradients=True)
        assert torch.equal(S_g, S_comp) and len(gV) == 3
        for j in range(3):
            for i in range(3):
                assert torch.equal(gV[j][:, i], gu[i][j]), (i, j)
        wa_f = T.Wray_Agarwal_of(dpart, R, S_comp)
        wa_c = T.Wray_Agarwal(R, S_comp, ibamd.cell_gradient_array(dpart, R), ibamd.cell_gradient_array(dpart, S_comp))
        for k in ("nut", "nuR", "S"):
            assert torch.equal(wa_f[k], wa_c[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("nd", [2, 3])
def test_fused_closures_on_face_list_partitions_are_the_composition(nd):
    """The same two closures on partitions WITHOUT block structure (4-cell blocks: the coarse levels of ``multigrid()``):
    thread-per-cell kernels over the side table (``k_shear_of_velocity_cells`` / ``k_wray_agarwal_of_cells``) against the
    composition of ``cell_gradient`` and the pointwise kernels, bit for bit, 2-D and 3-D with 2:1 interfaces."""
    import torch
    import ibamd
    from ibamd import Ball, Mesh
    from ibamd import turbulence as T
    o, w = f32([-2] * nd), f32([4] * nd)
    msh = Mesh(o, w, block_size=4, refinement_regions=[(Ball(np.array([1.2] * nd), 0.1), f32(0.05 if nd == 2 else 0.12))])
    dom = ibamd.Domain(msh, max_partition_size=10 ** 9, boundaries=False)
    (part,) = dom.partitions.values()
    dpart = ibamd.to_backend(part, ibamd.hip)
    assert dpart.info["full_blocks"] == 0 and T.fused_closures_apply(dpart) and not T.all_blocks(dpart)
    assert 0 < dpart.info["direct_sides"] < 2 * nd * dpart.nc          # some sides take the CSR walk (2:1 interfaces)
    nc = part.centers.shape[0]
    X = part.centers
    rng = np.random.default_rng(33)
    vel_h = np.stack([np.sin(2 * X[:, (i + 1) % nd]) * np.cos(X[:, i]) for i in range(nd)], axis=1).astype(f32)
    vel_h += f32(0.05) * rng.standard_normal((nc, nd)).astype(f32)
    vel = ibamd.hip(vel_h)
    R = ibamd.hip((4.5e-5 * (1 + 0.5 * rng.uniform(0, 1, nc))).astype(f32))
    S_fused = T.shear_rate_of_velocity(dpart, vel)
    S_comp = T.shear_rate([list(ibamd.cell_gradient(dpart, vel[:, i].contiguous())) for i in range(nd)])
    assert torch.equal(S_fused, S_comp) and float(S_comp.abs().max()) > 0
    S_g, gV = T.shear_rate_of_velocity(dpart, vel, gradients=True)      # the gradients kept on the way (same kernel)
    assert torch.equal(S_g, S_comp) and len(gV) == nd
    for j in range(nd):
        for i in range(nd):
            assert torch.equal(gV[j][:, i], ibamd.cell_gradient(dpart, vel[:, i].contiguous())[j]), (i, j)
    wa_f = T.Wray_Agarwal_of(dpart, R, S_comp)
    wa_c = T.Wray_Agarwal(R, S_comp, ibamd.cell_gradient_array(dpart, R), ibamd.cell_gradient_array(dpart, S_comp))
    for k in ("nut", "nuR", "S"):
        assert torch.equal(wa_f[k], wa_c[k]), k


@pytest.mark.gpu
def test_block_transport_kernel_against_the_face_list_kernel():
    """``ibh_scalar_transport`` on a 3-D partition of complete blocks runs the block kernel (wave per 8^3 block, side fluxes
    by the slot lanes): against the face-list kernel of the same entry (tuning key ``transport_blocks`` 0) -- the same
    expressions in the same order, so equal bit for bit wherever a side has one face; behind a FINE side the four face
    fluxes may be summed in another order (last-bit differences)."""
    import torch
    import ibamd
    from ibamd import Ball, Mesh, _lib
    from ibamd import turbulence as T
    msh = Mesh(f32([-2, -2, -2]), f32([4, 4, 4]), block_size=8,
               refinement_regions=[(Ball(np.array([1.2, 1.2, 1.2]), 0.1), f32(0.1))])
    dom = ibamd.Domain(msh, max_partition_size=10 ** 9, boundaries=False)
    (part,) = dom.partitions.values()
    dpart = ibamd.to_backend(part, ibamd.hip)
    assert T.all_blocks(dpart) and dpart.info["sides_fine"] > 0 and dpart.info["sides_coarse"] > 0
    nc = part.centers.shape[0]
    X = part.centers
    rng = np.random.default_rng(8)
    R = ibamd.hip((4.5e-5 * (1 + 0.5 * rng.uniform(0, 1, nc))).astype(f32))
    nuR = ibamd.hip((1e-5 * rng.uniform(0.5, 2, nc)).astype(f32))
    S = ibamd.hip(rng.uniform(-1, 1, nc).astype(f32))
    vel = ibamd.hip(np.stack([100 * (1 + 0.1 * np.sin(X[:, 1])), 10 * np.cos(X[:, 0] + X[:, 2]),
                              5 * rng.uniform(-1, 1, nc)], axis=1).astype(f32))
    got = T.scalar_transport(dpart, R, nuR, vel, 1.5e-5, S)
    _lib.call("ibh_set_tuning", b"transport_blocks", 0)
    try:
        ref = T.scalar_transport(dpart, R, nuR, vel, 1.5e-5, S)
    finally:
        _lib.call("ibh_set_tuning", b"transport_blocks", 1)
    same = (got == ref)
    frac = float(same.float().mean())
    err = float((got - ref).abs().max() / ref.abs().max())
    print(f"block transport: {100 * frac:.2f} % of the cells bit-identical, max difference {err:.2e} of max |ref|")
    assert frac > 0.97 and err <= 1e-6
    assert float((ref - S).abs().max()) > 0
