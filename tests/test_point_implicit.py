"""Point-implicit smoother (reference: orphan /root/reference/src/point_implicit.jl): oracle known answers on the
CPU, device kernels and the device-resident host mirror against the oracle on the GPU."""
import numpy as np
import pytest

from oracle import point_implicit as opi

f32 = np.float32


def _blocks(n, nv, seed=0, shift=3.0):
    rng = np.random.default_rng(seed)
    A = rng.uniform(-1, 1, (n, nv, nv)).astype(f32)
    A += f32(shift) * np.eye(nv, dtype=f32)[None]
    return A


def _pointwise(A):
    """f(X)[p] = A_p X[p]: a function whose Jacobian IS block diagonal."""
    return lambda X: np.einsum("pki,pi->pk", A, X).astype(f32)


def _samples(n, nv, k, seed=1):
    rng = np.random.default_rng(seed)
    return [[rng.choice(f32([-1, 1]), n).astype(f32) for _ in range(k)] for _ in range(nv)]


def test_oracle_hutchinson_recovers_pointwise_blocks():
    n, nv = 200, 4
    A = _blocks(n, nv)
    X = np.random.default_rng(2).uniform(-1, 1, (n, nv)).astype(f32)
    D = opi.hutchinson_trick(_pointwise(A), X, _samples(n, nv, 3), h=1e-2)
    assert np.abs(D - A).max() <= 2e-3  # finite differences in Float32
    # vector form: diagonal of a diagonal map
    d = np.random.default_rng(3).uniform(1, 2, n).astype(f32)
    s = opi.hutchinson_trick(lambda x: d * x, X[:, 0].copy(), _samples(n, 1, 4)[0], h=1e-2)
    assert np.abs(s - d).max() <= 2e-3


def test_oracle_solve_converges_on_pointwise_system():
    n, nv = 300, 3
    A = _blocks(n, nv)
    f = _pointwise(A)
    X = np.zeros((n, nv), dtype=f32)
    b_true = np.random.default_rng(5).uniform(-1, 1, (n, nv)).astype(f32)
    g = lambda Y: f(Y) - b_true                     # root: A x = b_true
    lin, b, invD = opi.linearize(g, X, _samples(n, nv, 2), h=1e-2)
    x, ratio = opi.solve(lin, b, invD, n_iter=20, rtol=1e-3)
    assert ratio <= 2e-3
    exact = np.linalg.solve(A.astype(np.float64), b_true.astype(np.float64)[..., None])[..., 0]
    assert np.abs(x - exact).max() <= 5e-3 * np.abs(exact).max()


# ------------------------------------------------------------------------------------------------------------
gpu = pytest.mark.gpu


@gpu
@pytest.mark.parametrize("nv", [1, 2, 3, 4, 5, 6, 8])
def test_gpu_invert_and_apply_blocks(nv):
    import ibamd
    from ibamd import point_implicit as pi
    import torch
    n = 1000
    rng = np.random.default_rng(nv)
    if nv == 1:
        D = rng.uniform(0.5, 2, n).astype(f32)
        dD = ibamd.hip(D)
        pi._inverse_blocks(dD)
        assert np.allclose(ibamd.to_host(dD), opi.inverse_blocks(D), rtol=1e-6)
        v = rng.uniform(-1, 1, n).astype(f32)
        got = ibamd.to_host(pi.PIPreconditioner(dD)(ibamd.hip(v)))
        assert np.allclose(got, opi.apply_prec(opi.inverse_blocks(D), v), rtol=1e-6)
        return
    A = _blocks(n, nv, seed=nv)
    A[::7, :, 0] = A[::7, :, 1]        # rank-deficient blocks: two equal columns -> pinv, not inv
    A[5] = 0                            # zero block -> zero pseudo-inverse
    dD = torch.empty((nv, nv, n), dtype=torch.float32, device="cuda").permute(2, 1, 0)
    dD.copy_(torch.from_numpy(A))
    pi._inverse_blocks(dD)
    got = dD.cpu().numpy()
    exp = opi.inverse_blocks(A)
    scale = np.abs(exp).max(axis=(1, 2), keepdims=True) + 1e-30
    assert (np.abs(got - exp) / scale).max() <= 2e-4
    v = rng.uniform(-1, 1, (n, nv)).astype(f32)
    out = ibamd.to_host(pi.PIPreconditioner(dD)(ibamd.hip(v)))
    ref = opi.apply_prec(got, v)
    assert np.abs(out - ref).max() <= 1e-5 * np.abs(ref).max()


@gpu
def test_gpu_hutchinson_and_solve_match_oracle():
    import ibamd
    from ibamd import point_implicit as pi
    import torch
    n, nv = 2000, 4
    A = _blocks(n, nv, seed=11)
    dA = torch.from_numpy(A).cuda()
    f_np = _pointwise(A)
    b_true = np.random.default_rng(5).uniform(-1, 1, (n, nv)).astype(f32)
    g_np = lambda Y: f_np(Y) - b_true
    db = ibamd.hip(b_true)

    def g_dev(Y):
        out = ibamd.colmajor_empty(n, nv)
        out.copy_(torch.einsum("pki,pi->pk", dA, Y) - db)
        return out
    X = np.random.default_rng(2).uniform(-1, 1, (n, nv)).astype(f32)
    samp = _samples(n, nv, 3)
    dsamp = [[ibamd.hip(z) for z in col] for col in samp]
    D = pi.hutchinson_trick(g_dev, ibamd.hip(X), 3, h=1e-2, samples=dsamp)
    Dn = opi.hutchinson_trick(g_np, X, samp, h=1e-2)
    assert np.abs(D.cpu().numpy() - Dn).max() <= 2e-3 and np.abs(Dn - A).max() <= 2e-3
    lin, b, prec = pi.linearize(g_dev, ibamd.hip(X), 3, h=1e-2, samples=dsamp)
    x, ratio = pi.solve(lin, b, prec, n_iter=20, rtol=1e-3)
    lin_n, b_n, invD_n = opi.linearize(g_np, X, samp, h=1e-2)
    x_n, ratio_n = opi.solve(lin_n, b_n, invD_n, n_iter=20, rtol=1e-3)
    assert ratio <= 2e-3 and ratio_n <= 2e-3
    exact = np.linalg.solve(A.astype(np.float64), (b_true - f_np(X)).astype(np.float64)[..., None])[..., 0]
    assert np.abs(ibamd.to_host(x) - exact).max() <= 5e-3 * np.abs(exact).max()
    assert np.abs(ibamd.to_host(x) - x_n).max() <= 5e-3 * np.abs(exact).max()
    # default sampling on the device: reproducible +-1 vectors
    z = ibamd.to_host(pi.rademacher(10000, 42))
    assert set(np.unique(z)) == {-1.0, 1.0} and abs(z.mean()) < 0.05
    assert np.array_equal(z, ibamd.to_host(pi.rademacher(10000, 42)))


@gpu
def test_gpu_point_implicit_on_the_residual_sweep(adv_domains):
    """The smoother driving the real hot path: one pseudo-time step (u - u0)/dt + R(u) = 0 of the advection
    residual on a partition, solved by linearize + solve with the fused GPU sweep as ``f``."""
    import ibamd
    from ibamd import point_implicit as pi
    from conftest import seeded_field
    dp, _ = adv_domains
    part = dp.partitions[1]
    dpart = ibamd.to_backend(part, ibamd.hip)
    u0 = ibamd.hip(seeded_field(part.centers))
    Cf = ibamd.hip(np.ones((part.centers.shape[0], 2), dtype=f32))
    dt = f32(2e-3)

    def f(u):
        r = ibamd.residual_advection(dpart, u, Cf)      # ud = -div(flux)
        return (u - u0) / dt - r
    u = u0 + 0.01 * ibamd.hip(seeded_field(part.centers, seed=9))
    lin, b, prec = pi.linearize(f, u, 4, h=1e-3, seed=3)
    dx, ratio = pi.solve(lin, b, prec, n_iter=30, rtol=1e-2)
    assert ratio <= 1e-2
    import torch
    r0 = float(torch.linalg.norm(f(u)))
    r1 = float(torch.linalg.norm(f(u + dx)))
    assert r1 <= 0.1 * r0   # Float32 finite-difference linearisation of a limited (non-smooth) residual
