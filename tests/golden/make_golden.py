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


if __name__ == "__main__":
    main()
