"""Host mirror of the reference's Turbulence module (/root/reference/src/turbulence.jl) on device arrays: same
function names, keyword arguments and return fields (named tuples become dicts; the non-ASCII field names
``y⁺ u⁺ μ⁺ k⁺ du⁺!dy⁺ uτ νₜ ω ϵ du!dn`` are spelled ``yplus uplus muplus kplus duplus_dyplus utau nut omega
epsilon du_dn``).  Every function is one kernel of libibhip (``ibh_turb_*``); there is no CPU path.
``velocity_gradient[i][j]`` = device vector of d u_i / d x_j, as in the reference (Matrix of vectors)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import backend as B

_f = C.c_float


def _vec(t, n=None):
    t, nv, _ = B._field(t, n)
    if t.ndim != 1:
        raise TypeError("expected a device vector")
    return t


def _params(*vals):
    return (C.c_float * len(vals))(*[float(v) for v in vals])


def _grad_table(g):
    nd = len(g)
    if nd not in (2, 3) or any(len(r) != nd for r in g):
        raise ValueError("velocity_gradient must be nd x nd, nd = 2 or 3")
    n = g[0][0].shape[0]
    vs = [_vec(g[i][j], n) for i in range(nd) for j in range(nd)]
    return nd, n, vs, (B.c_vp * (nd * nd))(*[v.data_ptr() for v in vs])


def wall_function(*args, kappa=0.41, C_=4.9, A=19.0, beta=0.075, betastar=0.09, D=4.2, Aplus=360.0,
                  omega_fixed_point=0.5, n_iter=20):
    """``wall_function(Rey)`` (:27-70) or ``wall_function(y, u, nu)`` (:72-100)."""
    par = _params(kappa, C_, A, beta, betastar, D, Aplus, omega_fixed_point)
    B._stream()
    if len(args) == 1:
        Rey = _vec(args[0])
        n = Rey.shape[0]
        out = [B.colmajor_empty(n) for _ in range(5)]
        B.call("ibh_turb_wall_function_rey", n, B._ptr(Rey), C.cast(par, B.c_vp), int(n_iter), *[B._ptr(o) for o in out])
        return dict(zip(("yplus", "uplus", "muplus", "kplus", "duplus_dyplus"), out))
    if len(args) == 3:
        y = _vec(args[0])
        n = y.shape[0]
        u, nu = _vec(args[1], n), _vec(args[2], n)
        out = [B.colmajor_empty(n) for _ in range(6)]
        B.call("ibh_turb_wall_function", n, B._ptr(y), B._ptr(u), B._ptr(nu), C.cast(par, B.c_vp), int(n_iter),
               *[B._ptr(o) for o in out])
        return dict(zip(("utau", "nut", "k", "omega", "epsilon", "du_dn"), out))
    raise TypeError("wall_function(Rey) or wall_function(y, u, nu)")


def shear_rate(velocity_gradient):
    """:110-124 -- sqrt(2 Sij Sij)."""
    nd, n, keep, tab = _grad_table(velocity_gradient)
    S = B.colmajor_empty(n)
    B._stream()
    B.call("ibh_turb_shear_rate", nd, n, C.cast(tab, B.c_vp), B._ptr(S))
    return S


def Smagorinsky_nuSGS(Delta, S, Cs=0.17):
    """:135-138"""
    Delta = _vec(Delta)
    n = Delta.shape[0]
    S = _vec(S, n)
    out = B.colmajor_empty(n)
    B._stream()
    B.call("ibh_turb_smagorinsky", n, B._ptr(Delta), B._ptr(S), _f(Cs), B._ptr(out))
    return out


def standard_k_epsilon(k, eps, S, Cmu=0.09, sigma_k=1.0, sigma_eps=1.3, C1eps=1.44, C2eps=1.92):
    """:176-196 -- returns dict(nuk, nueps, Sk, Seps, nut)."""
    k = _vec(k)
    n = k.shape[0]
    eps, S = _vec(eps, n), _vec(S, n)
    out = [B.colmajor_empty(n) for _ in range(5)]
    par = _params(Cmu, sigma_k, sigma_eps, C1eps, C2eps)
    B._stream()
    B.call("ibh_turb_k_epsilon", n, B._ptr(k), B._ptr(eps), B._ptr(S), C.cast(par, B.c_vp), *[B._ptr(o) for o in out])
    return dict(zip(("nuk", "nueps", "Sk", "Seps", "nut"), out))


def Wray_Agarwal(R, S, gradR, gradS, sigmaR=0.72, C1=0.0829, kappa=0.41):
    """:222-241 -- returns dict(nut, nuR, S); gradR, gradS are (n, nd) device arrays."""
    R = _vec(R)
    n = R.shape[0]
    S = _vec(S, n)
    gR, nd, ldr = B._field(gradR, n)
    gS, nd2, lds = B._field(gradS, n)
    if nd != nd2 or nd not in (2, 3):
        raise ValueError("gradR, gradS must be (n, nd), nd = 2 or 3")
    out = [B.colmajor_empty(n) for _ in range(3)]
    B._stream()
    B.call("ibh_turb_wray_agarwal", nd, n, B._ptr(R), B._ptr(S), B._ptr(gR), ldr, B._ptr(gS), lds, _f(sigmaR), _f(C1),
           _f(kappa), *[B._ptr(o) for o in out])
    return dict(zip(("nut", "nuR", "S"), out))


def all_blocks(part):
    """True where every cell of the partition lies in a complete 8^3 block without a GENERAL side."""
    part = B._part(part)
    i = part.info
    return (part.nd == 3 and i["full_blocks"] > 0 and i["full_blocks"] * 512 == part.nc and i["irregular_cells"] == 0
            and i["sides_general"] == 0)


def fused_closures_apply(part):
    """The fused closures below (gradients consumed where they are made) apply on partitions made of complete 3-D blocks and
    on partitions without block structure; elsewhere (blocks + face-list cells) they compose the operators, same result."""
    part = B._part(part)
    return all_blocks(part) or part.info["full_blocks"] == 0


def shear_rate_of_velocity(part, vel, gradients=False):
    """``shear_rate([cell_gradient(part, vel[:, i]) ...])`` (:110-124 over the tuple ``cell_gradient``,
    ImmersedBoundary.jl:980-988): ONE launch on an all-block 3-D partition (``ibh_shear_rate_of_velocity``), the composition
    elsewhere -- bit-identical.  ``gradients=True`` returns ``(S, cell_gradient(part, vel))`` -- the tuple over the
    dimensions of ``(nc, nd)`` arrays -- with the gradients made once (``ibh_shear_rate_of_velocity_grad``): a Navier-Stokes
    closure needs them again for its viscous fluxes."""
    part = B._part(part)
    v, nd, ldv = B._field(vel, part.nc)
    if nd != part.nd:
        raise ValueError("vel must be (nc, nd)")
    if fused_closures_apply(part):
        S = B.colmajor_empty(part.nc)
        B._stream()
        if gradients:
            G = B.colmajor_empty(part.nc, nd * nd)
            B.call("ibh_shear_rate_of_velocity_grad", part.handle, B._ptr(v), ldv, B._ptr(S), B._ptr(G), part.nc)
            return S, tuple(G[:, j * nd:(j + 1) * nd] for j in range(nd))
        B.call("ibh_shear_rate_of_velocity", part.handle, B._ptr(v), ldv, B._ptr(S))
        return S
    S = shear_rate([list(B.cell_gradient(part, vel[:, i].contiguous())) for i in range(nd)])
    return (S, B.cell_gradient(part, vel)) if gradients else S


def Wray_Agarwal_of(part, R, S, sigmaR=0.72, C1=0.0829, kappa=0.41):
    """``Wray_Agarwal(R, S, cell_gradient(part, R), cell_gradient(part, S))`` (:222-241): ONE launch on an all-block 3-D
    partition (``ibh_wray_agarwal_of``), the composition elsewhere -- bit-identical."""
    part = B._part(part)
    R = _vec(R, part.nc)
    S = _vec(S, part.nc)
    if fused_closures_apply(part):
        out = [B.colmajor_empty(part.nc) for _ in range(3)]
        B._stream()
        B.call("ibh_wray_agarwal_of", part.handle, B._ptr(R), B._ptr(S), _f(sigmaR), _f(C1), _f(kappa),
               *[B._ptr(o) for o in out])
        return dict(zip(("nut", "nuR", "S"), out))
    return Wray_Agarwal(R, S, B.cell_gradient_array(part, R), B.cell_gradient_array(part, S), sigmaR=sigmaR, C1=C1,
                        kappa=kappa)


def scalar_transport(part, R, nuR, vel, nu, S, out=None):
    """``S + sum_d green_gauss(part, at_faces(part, nu .+ nuR, d) .* face_gradient(part, R, d) .- at_faces(part, vel[:, d] .* R, d), d)``
    in one launch (``ibh_scalar_transport``): the transport terms of a one-equation turbulence model, bit-identical to the
    operator-by-operator composition."""
    part = B._part(part)
    R = _vec(R, part.nc)
    nuR = _vec(nuR, part.nc)
    S = _vec(S, part.nc)
    v, nd, ldv = B._field(vel, part.nc)
    if nd != part.nd:
        raise ValueError("vel must be (nc, nd)")
    if out is None:
        out = B.colmajor_empty(part.nc)
    else:
        out, nvo, _ = B._field_inplace(out, part.nc, "out")
        if nvo != 1:
            raise ValueError("out must be a vector")
    B._stream()
    B.call("ibh_scalar_transport", part.handle, B._ptr(R), B._ptr(nuR), _f(nu), B._ptr(v), ldv, B._ptr(S), B._ptr(out))
    return out


def Ducros_sensor(velocity_gradient):
    """:252-282"""
    nd, n, keep, tab = _grad_table(velocity_gradient)
    out = B.colmajor_empty(n)
    B._stream()
    B.call("ibh_turb_ducros", nd, n, C.cast(tab, B.c_vp), B._ptr(out))
    return out


def WALE_nuSGS(Delta, velocity_gradient, Cw=0.325):
    """:291-337 (3-D only, like the reference's @assert)."""
    nd, n, keep, tab = _grad_table(velocity_gradient)
    assert nd == 3, "WALE model only implemented for 3D"
    Delta = _vec(Delta, n)
    out = B.colmajor_empty(n)
    B._stream()
    B.call("ibh_turb_wale", n, B._ptr(Delta), C.cast(tab, B.c_vp), _f(Cw), B._ptr(out))
    return out
