"""``CFD`` pointwise physics on device arrays (mirror of /root/reference/src/cfd.jl for the functions residual
closures call): ``Fluid``, ``speed_of_sound``, ``dynamic_viscosity``, ``heat_conductivity``,
``primitive2state``, ``state2primitive``, ``inviscid_fluxes`` (HLL and sensor/Rusanov methods),
``viscous_fluxes``.  Same names and argument order; ``dim`` is the 1-based Cartesian direction (the
matrix-normal form of the reference is a curvilinear extension outside this hot path).
Arithmetic runs in libibhip kernels (csrc/ibh_cfd.hip); device arrays only.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from . import backend as B


class Fluid:
    """cfd.jl:14-53 (defaults = air, Float32)."""

    def __init__(self, R=283.0, gamma=1.4, k=(0.00646, 6.468e-5), mu_ref=1.716e-5, Tref=273.15, S=110.4):
        self.R, self.gamma = float(R), float(gamma)
        self.k = [float(k)] if np.isscalar(k) else [float(x) for x in k]
        if len(self.k) > 4:
            raise ValueError("at most 4 heat-conductivity coefficients are supported")
        self.mu_ref, self.Tref, self.S = float(mu_ref), float(Tref), float(S)

    def _c(self):
        kk = (C.c_float * 4)(*(self.k + [0.0] * (4 - len(self.k))))
        return _lib.ibh_fluid(self.R, self.gamma, self.mu_ref, self.Tref, self.S, len(self.k), kk)


def _pointwise(name, fld, T):
    T, nv, _ = B._field(T)
    flat = T if T.ndim == 1 else T.T.contiguous().T
    out = B._like(flat, flat.shape[0])
    f = fld._c()
    B._stream()
    B.call(name, C.byref(f), int(flat.numel()), B._ptr(flat), B._ptr(out))
    return out


def speed_of_sound(fld, T):
    """cfd.jl:62-64"""
    return _pointwise("ibh_cfd_speed_of_sound", fld, T)


def dynamic_viscosity(fld, T):
    """cfd.jl:71-77"""
    return _pointwise("ibh_cfd_dynamic_viscosity", fld, T)


def JST_sensor(Pim1, Pi, Pip1):
    """``CFD.JST_sensor(Pim1, Pi, Pip1)`` (cfd.jl:563-573): the three-point form, elementwise."""
    a, _, _ = B._field(Pim1)
    b, _, _ = B._field(Pi)
    c, _, _ = B._field(Pip1)
    if not (a.shape == b.shape == c.shape):
        raise ValueError("JST_sensor: the three arrays must have one shape")
    a, b, c = (x if x.ndim == 1 or x.stride(1) == x.shape[0] else x.T.contiguous().T for x in (a, b, c))
    out = B._like(a, a.shape[0])
    B._stream()
    B.call("ibh_cfd_jst_sensor3", int(a.numel()), B._ptr(a), B._ptr(b), B._ptr(c), B._ptr(out))
    return out


def shock_sensor(velocity_gradients):
    """``CFD.shock_sensor`` (cfd.jl:575-617): ``velocity_gradients[i][j]`` = device array of d u_i / d x_j."""
    nd = len(velocity_gradients)
    arrs = [B._field(velocity_gradients[i][j])[0] for i in range(nd) for j in range(nd)]
    n = arrs[0].shape[0]
    if any(a.ndim != 1 or a.shape[0] != n for a in arrs):
        raise ValueError("shock_sensor: gradients are vectors of one length")
    ptrs = (B.c_vp * (nd * nd))(*[a.data_ptr() for a in arrs])
    out = B.colmajor_empty(n)
    B._stream()
    B.call("ibh_cfd_shock_sensor", nd, n, ptrs, B._ptr(out))
    return out


def heat_conductivity(fld, T):
    """cfd.jl:84-90"""
    return _pointwise("ibh_cfd_heat_conductivity", fld, T)


def _nd(P):
    nd = P.shape[1] - 2
    if P.ndim != 2 or nd not in (2, 3):
        raise ValueError("expected (n, nd+2) with nd = 2 or 3")
    return nd


def primitive2state(fluid, P):
    """cfd.jl:106-123"""
    P, _, ldp = B._field(P)
    nd = _nd(P)
    Q = B._like(P, P.shape[0])
    f = fluid._c()
    B._stream()
    B.call("ibh_cfd_primitive2state", C.byref(f), nd, P.shape[0], B._ptr(P), ldp, B._ptr(Q), P.shape[0])
    return Q


def state2primitive(fluid, Q):
    """cfd.jl:137-151"""
    Q, _, ldq = B._field(Q)
    nd = _nd(Q)
    P = B._like(Q, Q.shape[0])
    f = fluid._c()
    B._stream()
    B.call("ibh_cfd_state2primitive", C.byref(f), nd, Q.shape[0], B._ptr(Q), ldq, B._ptr(P), Q.shape[0])
    return P


def inviscid_fluxes(fluid, PL, PR, *args):
    """``inviscid_fluxes(fluid, PL, PR, dim)`` (HLL, cfd.jl:459-508) or
    ``inviscid_fluxes(fluid, PL, PR, nuL, nuR, dim)`` (sensor/Rusanov, cfd.jl:516-554)."""
    PL, _, ld = B._field(PL)
    PR, _, ld2 = B._field(PR)
    if ld2 != ld:
        PR = PR.T.contiguous().T
        PL = PL.T.contiguous().T
        ld = PL.shape[0]
    nd = _nd(PL)
    n = PL.shape[0]
    F = B._like(PL, n)
    f = fluid._c()
    B._stream()
    if len(args) == 1:
        B.call("ibh_cfd_inviscid_fluxes_hll", C.byref(f), nd, int(args[0]), n, B._ptr(PL), B._ptr(PR), ld, B._ptr(F), n)
    elif len(args) == 3:
        nuL, _, _ = B._field(args[0], n)
        nuR, _, _ = B._field(args[1], n)
        B.call("ibh_cfd_inviscid_fluxes_sensor", C.byref(f), nd, int(args[2]), n, B._ptr(PL), B._ptr(PR), ld,
               B._ptr(nuL), B._ptr(nuR), B._ptr(F), n)
    else:
        raise TypeError("inviscid_fluxes(fluid, PL, PR, dim) or inviscid_fluxes(fluid, PL, PR, nuL, nuR, dim)")
    return F


def viscous_fluxes(fluid, P, Pgrad, dim, mu_t=0.0):
    """cfd.jl:664-736 (Cartesian ``dim``); ``Pgrad`` = tuple of the gradients of P along each axis."""
    P, _, ldp = B._field(P)
    nd = _nd(P)
    n = P.shape[0]
    grads = [B._field(g, n)[0].T.contiguous().T for g in Pgrad]
    if len(grads) != nd:
        raise ValueError("Pgrad needs one array per dimension")
    ptrs = (B.c_vp * nd)(*[g.data_ptr() for g in grads])
    F = B._like(P, n)
    f = fluid._c()
    mt_arr, mt_const = None, 0.0
    if hasattr(mu_t, "data_ptr"):
        mt_arr, _, _ = B._field(mu_t, n)
    else:
        mt_const = float(mu_t)
    B._stream()
    B.call("ibh_cfd_viscous_fluxes", C.byref(f), nd, int(dim), n, B._ptr(P), ldp, ptrs, n, B._ptr(mt_arr),
           C.c_float(mt_const), B._ptr(F), n)
    return F


def viscous_residual(part, fluid, P, Pgrad, mu_t, R, velocity_gradients_only=False):
    """``R[:, 1:] .+= sum_d green_gauss(part, viscous_fluxes(fluid, at_faces(part, P, d), face_gradient(part, P, Pgrad, d), d;
    mu_t = at_faces(part, mu_t, d)), d)`` in ONE launch (``ibh_viscous_residual``), bit-identical to that composition
    (cfd.jl:664-736 over ImmersedBoundary.jl:899-926, 1039-1069).  ``Pgrad`` = the tuple ``cell_gradient(part, P)`` -- or, with
    ``velocity_gradients_only``, ``cell_gradient(part, P[:, 3:end])``: the viscous fluxes read the gradients of the velocities
    and the NORMAL derivative of T only, which is a ``face_gradient`` --, ``mu_t`` a cell array, ``R`` the (nc, nd + 2)
    residual updated in place."""
    part = B._part(part)
    P, _, ldp = B._field(P, part.nc)
    nd = _nd(P)
    grads = [B._field(g, part.nc)[0] for g in Pgrad]
    if len(grads) != nd:
        raise ValueError("Pgrad needs one array per dimension")
    ldg = {g.stride(1) for g in grads}
    if len(ldg) != 1:
        grads = [g.T.contiguous().T for g in grads]
        ldg = {part.nc}
    ptrs = (B.c_vp * nd)(*[g.data_ptr() for g in grads])
    mt = B._field(mu_t, part.nc)[0]
    R, nvr, ldr = B._field_inplace(R, part.nc, "R")
    if nvr != nd + 2:
        raise ValueError("R must be (nc, nd + 2)")
    f = fluid._c()
    B._stream()
    B.call("ibh_viscous_residual", part.handle, C.byref(f), B._ptr(P), ldp, ptrs, int(ldg.pop()),
           0 if velocity_gradients_only else 2, B._ptr(mt), B._ptr(R), ldr)
    return R


class FlowBC:
    """cfd.jl:160-300 -- generic flow boundary condition, called on device arrays inside ``impose_bc`` closures:
    ``bc(P, bdry.normals)``, ``bc(P, bdry.normals, du_dn=..., image_distances=bdry.image_distances, transpiration=...)``.
    ``FlowBC(fluid, [p, T, u, v(, w)])`` is a Dirichlet state (inlet/outlet/no-slip wall), ``FlowBC(fluid, [p, T, un],
    normal_flow=True)`` imposes the normal velocity only (slip wall)."""

    def __init__(self, fluid, P, normal_flow=False):
        self.fluid = fluid
        self.p_inf, self.T_inf = float(P[0]), float(P[1])
        self.u_inf = np.ascontiguousarray(np.asarray(P[2:], dtype=np.float32))
        self.normal_flow = bool(normal_flow)

    def __call__(self, P, normals, image_distances=None, du_dn=None, transpiration=0.0):
        P, _, ldp = B._field(P)
        nd = _nd(P)
        n = P.shape[0]
        nrm, nnv, ldn = B._field(normals, n)
        if nnv != nd:
            raise ValueError("normals must be (n, nd)")
        if self.normal_flow:
            assert self.u_inf.size == 1, "Only 3 parcels in P (p, T and normal flow) allowed for normal_flow = true BC"
        elif self.u_inf.size != nd:
            raise ValueError("FlowBC needs one free-stream velocity component per dimension")
        if (du_dn is None) != (image_distances is None):
            raise ValueError("du!dn and image_distances must be passed together for BC imposition")
        imd = None if image_distances is None else B._field(image_distances, n)[0]
        dn = None if du_dn is None else B._field(du_dn, n)[0]
        tv, tc = None, 0.0
        if hasattr(transpiration, "data_ptr"):
            tv = B._field(transpiration, n)[0]
        else:
            tc = float(transpiration)
        out = B._like(P, n)
        f = self.fluid._c()
        B._stream()
        B.call("ibh_cfd_flow_bc", C.byref(f), nd, n, B._ptr(P), ldp, B._ptr(nrm), ldn, C.c_float(self.p_inf),
               C.c_float(self.T_inf), self.u_inf.ctypes.data_as(B.c_vp), int(self.normal_flow), B._ptr(imd), B._ptr(dn),
               C.c_float(tc), B._ptr(tv), B._ptr(out), n)
        return out
