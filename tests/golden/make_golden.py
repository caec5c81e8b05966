"""Generates tests/golden/advection_partition.npz: one partition of the (coarse) advection case of
/root/reference/test/dissipation.jl's mesh as a flat "partition pack" + seeded inputs + the outputs
of every operator and of the two fused residuals as evaluated by the ORACLE (oracle/*.py).

The Julia reference cannot run in the build container (no julia), so these vectors are NOT outputs of
the reference itself: they pin the oracle restatement (itself pinned by the reference's known answers,
tests/test_oracle_known_answers.py) against regressions and let the GPU tests check the HIP path against
committed numbers.  Re-run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import ibamd  # noqa: E402
from conftest import ADV_FAMILIES, advection_mesh, euler_field, seeded_field  # noqa: E402
from oracle import cfd as ocfd  # noqa: E402
from oracle import domain as od  # noqa: E402
from test_gpu_residual import oracle_advection_residual, oracle_euler_residual  # noqa: E402


def main():
    msh = advection_mesh(2e-2)
    kw = dict(hypercube_families=ADV_FAMILIES, max_partition_size=1536)
    dp, do = ibamd.Domain(msh, **kw), od.Domain(msh, **kw)
    pid = 2  # a middle partition: skirts on both sides, level jumps, mirror faces
    pp, op = dp.partitions[pid], do.partitions[pid]
    out = dict(nd=np.int32(2), block_size=np.int32(msh.block_size), spacing=pp.spacing, centers=pp.centers,
               domain=pp.domain, image=pp.image, image_in_domain=pp.image_in_domain)
    for d in (1, 2):
        out[f"owners{d}"], out[f"neighbors{d}"] = pp.face_owners_neighbors[d]
        for r, nm in ((False, "left"), (True, "right")):
            acc = pp.face_accumulators[(d, r)]
            out[f"{nm}_off{d}"], out[f"{nm}_idx{d}"], out[f"{nm}_w{d}"] = acc.off, acc.idx, acc.w
    u = seeded_field(op.centers, kind="step")
    u2 = seeded_field(op.centers, nv=2)
    C = np.stack([np.ones_like(u), np.float32(0.5) + seeded_field(op.centers, seed=3) * np.float32(0.1)], axis=1)
    P = euler_field(op.centers)
    out.update(u=u, u2=u2, C=C, P=P)
    D = od.JST_sensor(op, u)
    out["jst"] = D
    for d in (1, 2):
        out[f"at_faces{d}"] = od.at_faces(op, u2, d)
        out[f"cell_gradient{d}"] = od.cell_gradient(op, u2, d)
        out[f"face_gradient{d}"] = od.face_gradient(op, u2, d)
        out[f"green_gauss{d}"] = od.green_gauss(op, out[f"at_faces{d}"], d)
        out[f"ugg{d}"] = od.unsigned_green_gauss(op, out[f"at_faces{d}"], d)
        gu = od.cell_gradient(op, u, d)
        out[f"musclL{d}"], out[f"musclR{d}"] = od.MUSCL(op, u, gu, d, D=D, high_order=True)
    out["res_adv"] = oracle_advection_residual(op, u, C)
    out["res_euler"] = oracle_euler_residual(op, P, ocfd.Fluid())
    np.savez_compressed(os.path.join(HERE, "advection_partition.npz"), **out)
    print("wrote", os.path.join(HERE, "advection_partition.npz"), "cells", pp.spacing.shape[0])


def main3d():
    """tests/golden/octree_partition.npz: a 15-block 3-D octree (two refinement levels in a corner of the box: SAME,
    COARSE, FINE and MIRROR block sides, 7 680 cells) as a flat partition pack + seeded inputs + the oracle's scalar and
    Euler residuals and cell gradients."""
    from ibamd.mesher import Ball, Mesh
    f32 = np.float32
    msh = Mesh(f32([-2, -2, -2]), f32([4, 4, 4]), block_size=8,
               refinement_regions=[(Ball(np.array([-2.0, -2.0, -2.0]), 0.1), f32(0.2))])
    dp, do = ibamd.Domain(msh, max_partition_size=10 ** 9, boundaries=False), od.Domain(msh, max_partition_size=10 ** 9)
    pp, op = dp.partitions[1], do.partitions[1]
    out = dict(nd=np.int32(3), block_size=np.int32(msh.block_size), spacing=pp.spacing, centers=pp.centers,
               domain=pp.domain, image=pp.image, image_in_domain=pp.image_in_domain)
    for d in (1, 2, 3):
        out[f"owners{d}"], out[f"neighbors{d}"] = pp.face_owners_neighbors[d]
        for r, nm in ((False, "left"), (True, "right")):
            acc = pp.face_accumulators[(d, r)]
            out[f"{nm}_off{d}"], out[f"{nm}_idx{d}"], out[f"{nm}_w{d}"] = acc.off, acc.idx, acc.w
    rng = np.random.default_rng(2026)
    x = op.centers.astype(np.float64)
    n = x.shape[0]
    u = (np.sin(2 * x[:, 0]) * np.cos(3 * x[:, 1]) + 0.3 * x[:, 2] + 0.1 * rng.uniform(-1, 1, n)).astype(f32)
    C = np.stack([np.ones(n, f32), (0.5 + 0.1 * rng.uniform(-1, 1, n)).astype(f32), f32(-0.25) * np.ones(n, f32)], axis=1)
    P = np.stack([1e5 * (1 + 0.05 * rng.uniform(-1, 1, n)), 288.15 * (1 + 0.05 * rng.uniform(-1, 1, n)),
                  100.0 * (1 + 0.1 * rng.uniform(-1, 1, n)), 60.0 * (1 + 0.1 * rng.uniform(-1, 1, n)),
                  -40.0 * (1 + 0.1 * rng.uniform(-1, 1, n))], axis=1).astype(f32)
    out.update(u=u, C=C, P=P)
    D = od.JST_sensor(op, u)
    out["jst"] = D
    res = np.zeros(n, f32)
    for d in (1, 2, 3):
        Cf = od.at_faces(op, np.ascontiguousarray(C[:, d - 1]), d)
        gu = od.cell_gradient(op, u, d)
        out[f"cell_gradient{d}"] = gu
        uL, uR = od.MUSCL(op, u, gu, d, D=D, high_order=True)
        res -= od.green_gauss(op, (uL + uR) * Cf / f32(2) + np.abs(Cf) * (uL - uR) / f32(2), d)
    out["res_adv"] = res
    fluid = ocfd.Fluid()
    R = np.zeros_like(P)
    Dp = od.JST_sensor(op, np.ascontiguousarray(P[:, 0]))
    for d in (1, 2, 3):
        gP = od.cell_gradient(op, P, d)
        PL, PR = od.MUSCL(op, P, gP, d, D=Dp, high_order=True)
        R -= od.green_gauss(op, ocfd.inviscid_fluxes(fluid, PL, PR, d), d)
    out["res_euler"] = R
    np.savez_compressed(os.path.join(HERE, "octree_partition.npz"), **out)
    print("wrote", os.path.join(HERE, "octree_partition.npz"), "cells", n)


if __name__ == "__main__":
    main()
    main3d()
