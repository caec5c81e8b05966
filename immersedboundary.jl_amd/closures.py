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
    vel = [Q[:, 2 + i].contiguous() for i in range(nd)]
    # cell_gradient(part, u): the tuple form, one sweep per field for all dimensions
    gu = [list(B.cell_gradient(part, vel[i])) for i in range(nd)]
    S = T.shear_rate(gu)
    gR = B.cell_gradient_array(part, R)      # (nc, nd): the buffer of the tuple form, no copies
    gS = B.cell_gradient_array(part, S)
    wa = T.Wray_Agarwal(R, S, gR, gS)
    # S + sum_d green_gauss(at_faces(nu + nuR) .* face_gradient(R) .- at_faces(u_d .* R)) in one launch, straight into r
    T.scalar_transport(part, R, wa["nuR"], Q[:, 2:2 + nd], float(nu), wa["S"], out=r[:, nvp])
    return r
