"""User-level residual closures composed from the operator API -- what a solver script written against the reference
would contain, kept here so that the benchmark and the tests drive the same code.

``euler_wray_agarwal_residual`` is the residual of BASELINE.json configs[4]: compressible Euler (JST + MUSCL + HLL,
fused sweep) plus a one-equation turbulence scalar, closed with the model the reference really has
(``Wray_Agarwal``, /root/reference/src/turbulence.jl:222-241):

    R_t = -div(u R) + div[(nu + nu_R) grad R] + S,   (nu_t, nu_R, S) = Wray_Agarwal(R, shear_rate(grad u), grad R, grad S)
"""
from __future__ import annotations

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


def config5_boundary_conditions(dom, Q, far, wall_name="sphere", far_name="farfield", fluid=None, R_inf=None):
    """The boundary conditions of a level of BASELINE.json configs[4], ``impose_bc!`` on THAT level's own ``Boundary``
    structs (``multigrid`` builds every coarse ``Domain`` with its boundaries, ImmersedBoundary.jl:1381-1382), in the order a
    solver script would write them:

    * far field (hypercube family): ``FlowBC(fluid, [p, T, u, v, w])`` on the primitives (cfd.jl:243-300), the free-stream
      value on the turbulence scalar;
    * immersed wall: slip wall ``FlowBC(fluid, [p, T, 0]; normal_flow = true)`` with ``du!dn`` from
      ``wall_function(y, u, nu)`` (turbulence.jl:72-98) at the image points -- ``y`` = image distance, ``u`` = tangential
      speed at the image point, ``nu = mu(T) / rho`` -- and the wall function's ``nu_t`` as the value of the scalar.

    ``Q = [p T u v w R]`` (global device array of the level, updated in place)."""
    from . import cfd
    fluid = fluid or cfd.Fluid()
    nd = dom.ndims
    P, R = Q[:, :nd + 2], Q[:, nd + 2]
    free = cfd.FlowBC(fluid, far)
    wall = cfd.FlowBC(fluid, [far[0], far[1], 0.0], normal_flow=True)
    R_inf = float(R_inf if R_inf is not None else 3 * 1.5e-5)
    B.impose_bc(lambda b, Pi, Ri: (free(Pi, b.normals), R_inf), dom, far_name, P, R)

    def wall_bc(b, Pi, Ri):
        # (boundary-sized arrays: a dozen small ATen kernels.  The same lines as three broadcast launches of HipArray
        # expressions were measured SLOWER -- V-cycle at 7.9 M cells 6.41 against 6.09 ms, same box, alternating --: these
        # stretches of the V-cycle are bound by the host's launch rate, and an expression launch costs more host time)
        import torch
        rho = Pi[:, 0] / (fluid.R * Pi[:, 1])
        nu = cfd.dynamic_viscosity(fluid, Pi[:, 1].contiguous()) / rho
        un = (Pi[:, 2:] * b.normals).sum(dim=1)
        ut = torch.sqrt(((Pi[:, 2:] - un[:, None] * b.normals) ** 2).sum(dim=1))
        wf = T.wall_function(b.image_distances, ut.contiguous(), nu.contiguous())
        return wall(Pi, b.normals, du_dn=wf["du_dn"], image_distances=b.image_distances), wf["nut"]
    B.impose_bc(wall_bc, dom, wall_name, P, R)


def navier_stokes_wray_agarwal_residual(part, Q, nu=1.5e-5, fluid=None, out=None, fused_viscous=True):
    """``Q = [p T u v (w) R]`` -> residual: Euler HLL sweep (``ibh_residual_euler_hll``) + the viscous fluxes with the eddy
    viscosity, ``sum_d green_gauss(viscous_fluxes(fluid, at_faces(P), face_gradient(P, grad P, d), d; mu_t = at_faces(rho nu_t)), d)``
    (cfd.jl:664-736 over ImmersedBoundary.jl:899-1069) + the Wray-Agarwal transport equation of ``euler_wray_agarwal_residual``.
    ``fused_viscous``: the viscous sum in one launch (``ibh_viscous_residual``, bit-identical); False = operator by operator."""
    from . import cfd
    fluid = fluid or cfd.Fluid()
    nd = part.nd
    nvp = nd + 2
    if Q.shape[1] != nvp + 1:
        raise ValueError(f"Q must be (nc, {nvp + 1}) = [p T u v (w) R]")
    r = out if out is not None else B.colmajor_empty(Q.shape[0], nvp + 1)
    P = Q[:, :nvp]
    B.residual_euler_hll(part, P, out=r[:, :nvp])
    R = Q[:, nvp].contiguous()
    # the velocity gradients are made once: the shear rate consumes them on the way, the viscous sum reads them again
    if fused_viscous:
        S, gV = T.shear_rate_of_velocity(part, Q[:, 2:2 + nd], gradients=True)
    else:
        S = T.shear_rate_of_velocity(part, Q[:, 2:2 + nd])
    wa = T.Wray_Agarwal_of(part, R, S)
    T.scalar_transport(part, R, wa["nuR"], Q[:, 2:2 + nd], float(nu), wa["S"], out=r[:, nvp])
    from .hiparray import HipArray
    mut = (HipArray(Q[:, 0]) / (HipArray(Q[:, 1]) * fluid.R) * HipArray(wa["nut"])).t   # mu_t = rho nu_t, one launch
    if fused_viscous:
        # sum_d green_gauss(viscous_fluxes(at_faces(P), face_gradient(P, gP, d), d; mu_t = at_faces(mu_t)), d) in one launch;
        # of cell_gradient(part, P) it reads the velocity columns only (the gradients of p and T are not formed)
        cfd.viscous_residual(part, fluid, P, gV, mut, r[:, :nvp], velocity_gradients_only=True)
        return r
    gP = B.cell_gradient(part, P)                                     # tuple over the dimensions of (nc, nd + 2)
    for d in range(1, nd + 1):
        Fv = cfd.viscous_fluxes(fluid, B.at_faces(part, P, d), B.face_gradient(part, P, gP, d), d,
                                mu_t=B.at_faces(part, mut.contiguous(), d))
        r[:, :nvp] += B.green_gauss(part, Fv, d)
    return r
