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
    for d in range(1, int(pk["nd"]) + 1):
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


# ---- 3-D: tests/golden/octree_partition.npz (36 blocks on three levels: SAME / COARSE / FINE / MIRROR sides)
@pytest.fixture(scope="module")
def pack3():
    return dict(np.load(os.path.join(GOLDEN, "octree_partition.npz")))


def _oracle_view3(part):
    from oracle.accumulator import Accumulator as OAcc

    class P:
        pass
    op = P()
    op.ndims, op.spacing, op.centers = 3, part.spacing, part.centers
    op.face_owners_neighbors = part.face_owners_neighbors
    op.face_accumulators = {}
    for k, acc in part.face_accumulators.items():
        o = object.__new__(OAcc)
        o.n_output, o.first_index, o.stencils = acc.n_output, True, acc.stencils
        op.face_accumulators[k] = o
    return op


def test_oracle_reproduces_golden_3d(pack3):
    from oracle import cfd as ocfd
    from oracle import domain as od
    op = _oracle_view3(_partition(pack3))
    u, C, P = pack3["u"], pack3["C"], pack3["P"]
    D = od.JST_sensor(op, u)
    assert np.array_equal(D, pack3["jst"])
    res = np.zeros(u.size, f32)
    for d in (1, 2, 3):
        Cf = od.at_faces(op, np.ascontiguousarray(C[:, d - 1]), d)
        gu = od.cell_gradient(op, u, d)
        assert np.array_equal(gu, pack3[f"cell_gradient{d}"])
        uL, uR = od.MUSCL(op, u, gu, d, D=D, high_order=True)
        res -= od.green_gauss(op, (uL + uR) * Cf / f32(2) + np.abs(Cf) * (uL - uR) / f32(2), d)
    assert np.array_equal(res, pack3["res_adv"])
    R = np.zeros_like(P)
    Dp = od.JST_sensor(op, np.ascontiguousarray(P[:, 0]))
    for d in (1, 2, 3):
        PL, PR = od.MUSCL(op, P, od.cell_gradient(op, P, d), d, D=Dp, high_order=True)
        R -= od.green_gauss(op, ocfd.inviscid_fluxes(ocfd.Fluid(), PL, PR, d), d)
    assert np.array_equal(R, pack3["res_euler"])


@pytest.mark.gpu
def test_hip_matches_golden_3d(pack3):
    """The 3-D kernels, fed from the flat partition pack alone through the C ABI, against the committed vectors: the strip
    sweep, the thread-per-cell single kernel, the two-kernel form, the face-list kernels; the Euler single kernel and its
    two-kernel form; the tuple cell_gradient."""
    from ibamd import _lib
    dpart = ibamd.to_backend(_partition(pack3), ibamd.hip)
    info = dpart.info
    assert info["fusable_blocks"] == info["full_blocks"] == 36 and info["sides_fine"] > 0 and info["sides_coarse"] > 0
    u, C, P = ibamd.hip(pack3["u"]), ibamd.hip(pack3["C"]), ibamd.hip(pack3["P"])
    strip = ibamd.to_host(ibamd.residual_advection(dpart, u, C))
    _lib.call("ibh_set_tuning", b"quad_variant", 512)
    try:
        cellk = ibamd.to_host(ibamd.residual_advection(dpart, u, C))
    finally:
        _lib.call("ibh_set_tuning", b"quad_variant", 0)
    two = ibamd.to_host(ibamd.residual_advection(dpart, u, C, flags=ibamd.IBH_NO_FUSE))
    gen = ibamd.to_host(ibamd.residual_advection(dpart, u, C, flags=ibamd.IBH_FORCE_GENERAL))
    assert np.array_equal(gen, pack3["res_adv"])             # literal arithmetic: the committed bits
    for got in (strip, cellk, two):
        assert rel_inf(got, pack3["res_adv"]) <= 1e-5
    one = ibamd.to_host(ibamd.residual_euler_hll(dpart, P))
    twoE = ibamd.to_host(ibamd.residual_euler_hll(dpart, P, flags=ibamd.IBH_NO_FUSE))
    for v in range(5):
        assert rel_inf(one[:, v], pack3["res_euler"][:, v]) <= 1e-5, v
        assert rel_inf(twoE[:, v], pack3["res_euler"][:, v]) <= 1e-5, v
    g3 = ibamd.cell_gradient(dpart, u)
    for d in (1, 2, 3):
        assert np.array_equal(ibamd.to_host(ibamd.cell_gradient(dpart, u, d)), pack3[f"cell_gradient{d}"])
        assert rel_inf(ibamd.to_host(g3[d - 1]), pack3[f"cell_gradient{d}"]) <= 5e-6
