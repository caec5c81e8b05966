"""Committed golden vectors (tests/golden/advection_partition.npz, made by tests/golden/make_golden.py).

CPU: the oracle reproduces them bit for bit (regression pin of the restatement).
GPU: the HIP path, fed from the flat partition pack alone through the C ABI, matches them (1e-5 norm-wise).
"""
import os

import numpy as np
import pytest

import ibamd
from conftest import GOLDEN, rel_inf
from ibamd.accumulator import Accumulator
from ibamd.domain import Partition

f32 = np.float32


@pytest.fixture(scope="module")
def pack():
    return dict(np.load(os.path.join(GOLDEN, "advection_partition.npz")))


def _partition(pk):
    fa, fon = {}, {}
    nf = {}
    for d in (1, 2):
        fon[d] = (pk[f"owners{d}"], pk[f"neighbors{d}"])
        nf[d] = pk[f"owners{d}"].size
        for r, nm in ((False, "left"), (True, "right")):
            fa[(d, r)] = Accumulator(csr=(pk[f"{nm}_off{d}"], pk[f"{nm}_idx{d}"], pk[f"{nm}_w{d}"]),
                                     first_index=True, n_input=nf[d])
    return Partition(1, pk["centers"], pk["spacing"], fa, fon, pk["domain"], pk["image"], pk["image_in_domain"],
                     block_size=int(pk["block_size"]))


def test_oracle_reproduces_golden(pack):
    from oracle import domain as od
    from oracle.accumulator import Accumulator as OAcc
    part = _partition(pack)

    class P:
        pass
    op = P()
    op.ndims, op.spacing, op.centers = 2, part.spacing, part.centers
    op.face_owners_neighbors = part.face_owners_neighbors
    op.face_accumulators = {}
    for k, acc in part.face_accumulators.items():
        o = object.__new__(OAcc)
        o.n_output, o.first_index, o.stencils = acc.n_output, True, acc.stencils
        op.face_accumulators[k] = o
    u, u2 = pack["u"], pack["u2"]
    assert np.array_equal(od.JST_sensor(op, u), pack["jst"])
    for d in (1, 2):
        assert np.array_equal(od.at_faces(op, u2, d), pack[f"at_faces{d}"])
        assert np.array_equal(od.cell_gradient(op, u2, d), pack[f"cell_gradient{d}"])
        assert np.array_equal(od.green_gauss(op, pack[f"at_faces{d}"], d), pack[f"green_gauss{d}"])
        uL, uR = od.MUSCL(op, u, od.cell_gradient(op, u, d), d, D=pack["jst"], high_order=True)
        assert np.array_equal(uL, pack[f"musclL{d}"]) and np.array_equal(uR, pack[f"musclR{d}"])


@pytest.mark.gpu
def test_hip_matches_golden(pack):
    dpart = ibamd.to_backend(_partition(pack), ibamd.hip)
    u, u2 = ibamd.hip(pack["u"]), ibamd.hip(pack["u2"])
    assert rel_inf(ibamd.to_host(ibamd.JST_sensor(dpart, u)), pack["jst"]) <= 1e-5
    D = ibamd.hip(pack["jst"])
    for d in (1, 2):
        assert rel_inf(ibamd.to_host(ibamd.at_faces(dpart, u2, d)), pack[f"at_faces{d}"]) <= 1e-6
        assert rel_inf(ibamd.to_host(ibamd.cell_gradient(dpart, u2, d)), pack[f"cell_gradient{d}"]) <= 1e-6
        assert rel_inf(ibamd.to_host(ibamd.face_gradient(dpart, u2, d)), pack[f"face_gradient{d}"]) <= 1e-6
        uf = ibamd.hip(pack[f"at_faces{d}"])
        assert rel_inf(ibamd.to_host(ibamd.green_gauss(dpart, uf, d)), pack[f"green_gauss{d}"]) <= 1e-6
        assert rel_inf(ibamd.to_host(ibamd.unsigned_green_gauss(dpart, uf, d)), pack[f"ugg{d}"]) <= 1e-6
        gu = ibamd.cell_gradient(dpart, u, d)
        uL, uR = ibamd.MUSCL(dpart, u, gu, d, D=D, high_order=True)
        assert rel_inf(ibamd.to_host(uL), pack[f"musclL{d}"]) <= 1e-6
        assert rel_inf(ibamd.to_host(uR), pack[f"musclR{d}"]) <= 1e-6
    C = ibamd.hip(pack["C"])
    for flags in (0, 1, 16):
        got = ibamd.to_host(ibamd.residual_advection(dpart, u, C, flags=flags))
        assert rel_inf(got, pack["res_adv"]) <= 1e-5, flags
    got = ibamd.to_host(ibamd.residual_euler_hll(dpart, ibamd.hip(pack["P"])))
    assert rel_inf(got, pack["res_euler"]) <= 1e-5
