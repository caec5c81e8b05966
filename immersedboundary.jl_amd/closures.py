"""User-level residual closures composed from the operator API -- what a solver script written against the reference
would contain, kept here so that the benchmark and the tests drive the same code.

``euler_wray_agarwal_residual`` is the residual of BASELINE.json configs[4]: compressible Euler (JST + MUSCL + HLL,
fused sweep) plus a one-equation turbulence scalar, closed with the model the reference really has
(``Wray_Agarwal``, /root/reference/src/turbulence.jl:222-241):

    R_t = -div(u R) + div[(nu + nu_R) grad R] + S,   (nu_t, nu_R, S) = Wray_Agarwal(R, shear_rate(grad u), grad R, grad S)
"""
from __future__ import annotations

import torch

from . import backend as B
from . import turbulence as T


def euler_wray_agarwal_residual(part, Q, nu=1.5e-5, out=None):
    """``Q = [p T u v (w) R]`` on a device partition -> residual array of the same shape."""
    nd = part.nd
    nvp = nd + 2
    if Q.shape[1] != nvp + 1:
        raise ValueError(f"Q must be (nc, {nvp + 1}) = [p T u v (w) R]")
    r = out if out is not None else B.colmajor_empty(Q.shape[0], nvp + 1)
    B.residual_euler_hll(part, Q[:, :nvp], out=r[:, :nvp])
    R = Q[:, nvp].contiguous()
    # S = shear_rate(cell_gradient(part, u_i) ...); Wray_Agarwal(R, S, cell_gradient(part, R), cell_gradient(part, S)):
    # one launch each where the partition is made of complete blocks (the gradients are consumed where they are made),
    # the operator-by-operator composition elsewhere
    S = T.shear_rate_of_velocity(part, Q[:, 2:2 + nd])
    wa = T.Wray_Agarwal_of(part, R, S)
    # S + sum_d green_gauss(at_faces(nu + nuR) .* face_gradient(R) .- at_faces(u_d .* R)) in one launch, straight into r
    T.scalar_transport(part, R, wa["nuR"], Q[:, 2:2 + nd], float(nu), wa["S"], out=r[:, nvp])
    return r
