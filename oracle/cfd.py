"""ORACLE (test infrastructure only; never imported by the product path).

CPU restatement (numpy, same dtypes and evaluation order) of the pointwise
physics the residual closures call: /root/reference/src/cfd.jl:14-151
(Fluid, speed_of_sound, dynamic_viscosity, heat_conductivity,
primitive2state, state2primitive), :243-300 (FlowBC), :459-554
(inviscid_fluxes HLL and Rusanov/sensor), :563-617 (JST 3-point sensor,
shock_sensor), :664-736 (viscous_fluxes).  ``dim`` is 1-based.

Reference quirk kept literally: in the HLL flux ``min(uR - aR, 0.0)`` /
``max(uL + aL, 0.0)`` use a Float64 literal, so the wave speeds and the
returned flux are Float64 (cfd.jl:504-507).
PARITY PINNING: no reference test asserts these; pinned by analytic checks in
tests/test_oracle_known_answers.py only.
"""
import numpy as np

f32 = np.float32


class Fluid:
    """cfd.jl:14-53 (defaults are Float32)."""

    def __init__(self, R=f32(283.0), gamma=f32(1.4), k=(f32(0.00646), f32(6.468e-5)),
                 mu_ref=f32(1.716e-5), Tref=f32(273.15), S=f32(110.4)):
        self.R = R
        self.gamma = gamma
        self.k = [k] if np.isscalar(k) else list(k)
        self.mu_ref = mu_ref
        self.Tref = Tref
        self.S = S


def speed_of_sound(fld, T):
    """cfd.jl:62-64"""
    return np.sqrt(fld.gamma * fld.R * np.maximum(T, f32(10.0)))


def dynamic_viscosity(fld, T):
    """cfd.jl:71-77 (Sutherland with the reference's exponent 2/3)."""
    T = np.maximum(T, f32(10.0))
    return fld.mu_ref * ((T / fld.Tref) ** (f32(2.0) / 3)) * (fld.Tref + fld.S) / (T + fld.S)


def heat_conductivity(fld, T):
    """cfd.jl:84-90"""
    k = 0 * T
    for i, ki in enumerate(fld.k):
        k = k + ki * T ** i
    return k


def _ke(u):
    s = u[:, 0] * u[:, 0]
    for j in range(1, u.shape[1]):
        s = s + u[:, j] * u[:, j]
    return s / f32(2)


def primitive2state(fluid, P):
    """cfd.jl:106-123: [p T u v (w)] -> [rho E rho*u ...]"""
    p = P[:, 0]
    T = np.maximum(P[:, 1], f32(10.0))
    u = P[:, 2:]
    k = _ke(u)
    rho = p / (fluid.R * T)
    E = rho * (fluid.R / (fluid.gamma - f32(1.0)) * T + k)
    return np.concatenate([rho[:, None], E[:, None], rho[:, None] * u], axis=1)


def state2primitive(fluid, Q):
    """cfd.jl:137-151"""
    rho = Q[:, 0]
    E = Q[:, 1]
    u = Q[:, 2:] / rho[:, None]
    k = _ke(u)
    p = (fluid.gamma - f32(1.0)) * (E - rho * k)
    T = np.maximum(p / (rho * fluid.R), f32(10.0))
    return np.concatenate([p[:, None], T[:, None], u], axis=1)


def _side_flux(fluid, P, dim):
    Q = primitive2state(fluid, P)
    F = Q.copy()
    p = P[:, 0]
    F[:, 1] += p
    u = P[:, 1 + dim]
    a = speed_of_sound(fluid, P[:, 1])
    F *= u[:, None]
    F[:, 1 + dim] += p
    return Q, F, u, a


def inviscid_fluxes(fluid, PL, PR, dim):
    """cfd.jl:459-508: HLL flux, Cartesian ``dim`` (1-based).  Returns Float64 (see header)."""
    QL, FL, uL, aL = _side_flux(fluid, PL, dim)
    QR, FR, uR, aR = _side_flux(fluid, PR, dim)
    SR = np.minimum((uR - aR).astype(np.float64), 0.0)[:, None]
    SL = np.maximum((uL + aL).astype(np.float64), 0.0)[:, None]
    return (SL * FL - SR * FR + SR * SL * (QR - QL)) / (SL - SR)


def inviscid_fluxes_sensor(fluid, PL, PR, nuL, nuR, dim):
    """cfd.jl:516-554: central flux + Rusanov dissipation scaled by sensors."""
    UcL = primitive2state(fluid, PL)
    UcL[:, 1] += PL[:, 0]
    UcR = primitive2state(fluid, PR)
    UcR[:, 1] += PR[:, 0]
    P = (PL + PR) / f32(2)
    p = P[:, 0]
    u = P[:, 1 + dim]
    a = speed_of_sound(fluid, P[:, 1])
    F = (UcL + UcR) * u[:, None] / f32(2)
    F[:, 1 + dim] += p
    nu = np.maximum(nuL, nuR)
    if nu.ndim == 1:
        nu = nu[:, None]
    F = F + (UcL - UcR) * (nu * (a + np.abs(u))[:, None] / f32(2))
    return F


def JST_sensor3(Pim1, Pi, Pip1):
    """cfd.jl:563-573"""
    e = f32(1e-14)
    return (np.abs(Pim1 + Pip1 - 2 * Pi) + e) / (np.abs(Pim1 - Pi) + np.abs(Pip1 - Pi) + e)


def shock_sensor(velocity_gradients):
    """cfd.jl:575-617; ``velocity_gradients[i][j]`` = d u_i / d x_j (0-based lists of arrays)."""
    e = f32(1e-14)
    nd = len(velocity_gradients)
    vort2 = np.zeros_like(velocity_gradients[0][0])
    divu = np.zeros_like(velocity_gradients[0][0])
    for i in range(nd):
        i_n = (i + 1) % nd
        i_nn = (i_n + 1) % nd
        divu = divu + velocity_gradients[i][i]
        vort2 = vort2 + (velocity_gradients[i_nn][i_n] - velocity_gradients[i_n][i_nn]) ** 2
    divu = divu ** 2
    return (divu + e) / (divu + vort2 + e)


def viscous_fluxes(fluid, P, Pgrad, dim, mu_t=f32(0.0)):
    """cfd.jl:664-736, Cartesian ``dim`` (1-based); ``Pgrad[j]`` = gradient along axis j+1."""
    T = P[:, 1]
    mu = dynamic_viscosity(fluid, T) + mu_t
    k = heat_conductivity(fluid, T)
    nd = P.shape[1] - 2

    def vg(i, j):
        return Pgrad[j - 1][:, 1 + i]

    divu = np.zeros_like(T)
    for i in range(1, nd + 1):
        divu = divu + vg(i, i)

    def tau(i, j):
        return ((vg(i, j) + vg(j, i)) - (f32(2.0) / 3 if i == j else f32(0.0)) * divu) * mu

    F = np.zeros_like(P)
    F[:, 1] += Pgrad[dim - 1][:, 1] * k
    for j in range(1, nd + 1):
        F[:, 1] += tau(dim, j) * P[:, 1 + j]
    for j in range(1, nd + 1):
        F[:, 1 + j] += tau(dim, j)
    return F


class FlowBC:
    """cfd.jl:160-300 (characteristic-style far-field / slip / no-slip BC at image points)."""

    def __init__(self, fluid, P, normal_flow=False):
        self.fluid = fluid
        self.p_inf = P[0]
        self.T_inf = P[1]
        self.u_inf = np.asarray(P[2:], dtype=f32)
        self.normal_flow = normal_flow

    def __call__(self, P, normals, image_distances=None, dudn=None, transpiration=f32(0.0)):
        p_inf, T_inf, u_inf = self.p_inf, self.T_inf, self.u_inf
        if self.normal_flow:
            assert len(u_inf) == 1
            un = np.full(P.shape[0], u_inf[0], dtype=P.dtype)
        else:
            un = normals @ u_inf
        p, T, u = P[:, 0], P[:, 1], P[:, 2:]
        cur = u[:, 0] * normals[:, 0]
        for j in range(1, u.shape[1]):
            cur = cur + u[:, j] * normals[:, j]
        a = speed_of_sound(self.fluid, T)
        M = np.abs(un) / a
        pb = (un >= 0.0) * ((M > 1.0) * p_inf + (M <= 1.0) * p) + (un < 0.0) * ((M > 1.0) * p + (M <= 1.0) * p_inf)
        Tb = (un > 0.0) * T_inf + (un <= 0.0) * T
        if self.normal_flow:
            ub = u + normals * (un - cur + transpiration)[:, None]
        else:
            ub = (un < 0.0)[:, None] * u + (un >= 0.0)[:, None] * u_inf[None, :]
        if (dudn is None) != (image_distances is None):
            raise ValueError("du!dn and image_distances must be passed together for BC imposition")
        if dudn is not None:
            e = np.finfo(ub.dtype).eps
            V = np.sqrt((ub * ub).sum(axis=1)) + e
            ub = ub * ((V - dudn * image_distances) / V)[:, None]
        return np.concatenate([pb[:, None], Tb[:, None], ub], axis=1).astype(P.dtype)
