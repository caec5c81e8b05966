"""Host logic of the quad sweep: the 2x2 block groups libibhip builds (ibh_build_quads2) and the lane-level model of
the kernel (tests/quad_model.py) against the oracle's operator-by-operator closure (test/advection.jl:67-83)."""
import numpy as np
import pytest

import ibamd
from conftest import ADV_FAMILIES, RAE_FAMILIES, rel_inf, seeded_field
from ibamd import hostview
from oracle import residual_c as rc
from quad_model import quad_sweep

f32 = np.float32


def _single_partition(msh, families):
    dom = ibamd.Domain(msh, hypercube_families=families, max_partition_size=10 ** 9)
    (part,) = dom.partitions.values()
    return part


@pytest.mark.parametrize("case", ["advection", "rae"])
@pytest.mark.parametrize("kind", ["smooth", "step"])
def test_quad_model_matches_oracle(adv_mesh, rae_mesh_small, case, kind):
    part = _single_partition(adv_mesh, ADV_FAMILIES) if case == "advection" else _single_partition(rae_mesh_small,
                                                                                                   RAE_FAMILIES)
    A = hostview.analyze2(part)
    assert A["fuse_all"]
    Q = A["quads_all"]
    nq = len(Q["desc"])
    blocks = A["blocks"]
    # every block is in exactly one quad or in the singles list
    covered = np.zeros(len(blocks), dtype=int)
    covered[Q["singles"]] += 1
    bybase = {int(b): i for i, b in enumerate(blocks["base"])}
    for d in Q["desc"]:
        for k in range(4):
            covered[bybase[int(d["base"]) + 64 * k]] += 1
    assert np.all(covered == 1)
    assert nq > 0.5 * len(blocks) / 4
    # pair tiles: two single blocks side by side with consecutive bases, each single block in at most one pair; what is
    # left is singles2
    inpair = np.zeros(len(blocks), dtype=int)
    for d in Q["pair_desc"]:
        for k in range(2):
            inpair[bybase[int(d["base"]) + 64 * k]] += 1
    assert inpair.max() <= 1 and np.all(inpair[Q["singles"]] + np.isin(Q["singles"], Q["singles2"]) == 1)
    assert len(Q["singles2"]) + 2 * len(Q["pair_desc"]) == len(Q["singles"])
    pcls = np.array([[(int(c) >> (4 * l)) & 15 for l in range(8)] for c in Q["pair_desc"]["cls"]]).reshape(-1, 8)
    assert np.all((pcls == 0) | (pcls == 2) | (pcls == 3))
    classes = np.array([[(int(c) >> (4 * l)) & 15 for l in range(8)] for c in Q["desc"]["cls"]])
    assert ({2, 3} if case == "rae" else {2}) <= set(np.unique(classes))  # coarse (and fine) half-sides are exercised
    # companion rows: where a half-side carries an origin, origin + stride * t (t >> 1 on COARSE half-sides) IS the row's
    # halo id of both sub-face slots; FINE half-sides never do; the end ids are the row's
    for tab, aux, cls in ((Q["tab"], Q["aux"], classes), (Q["pair_tab"], Q["pair_aux"], pcls)):
        assert aux.shape[0] == tab.shape[0]
        if not len(tab):
            continue
        assert np.array_equal(aux[:, :32], tab[:, 128:160])
        ids = tab[:, :128].reshape(-1, 4, 2, 8, 2)                       # [quad][g][half][t][k]
        orig = aux[:, 32:40].reshape(-1, 4, 2)
        ty = cls.reshape(-1, 4, 2)
        stride = np.array([8, 1, 1, 8])[None, :, None, None]
        tt = np.arange(8)[None, None, None, :]
        step = np.where((ty == 2)[..., None], tt >> 1, tt)
        want = orig[..., None] + stride * step
        ok = orig >= 0
        assert np.all(ty[ok] != 3)
        assert np.array_equal(ids[..., 0][ok], want[ok]) and np.array_equal(ids[..., 1][ok], want[ok])
        assert ok.mean() > 0.5
    u = seeded_field(part.centers, kind=kind)
    C = np.stack([np.ones_like(u), f32(0.5) + seeded_field(part.centers, seed=3) * f32(0.1)], axis=1)
    exp = rc.CPart(part).residual_advection(u, C)
    cells, got = quad_sweep(Q["desc"], Q["tab"], u, C)
    assert len(np.unique(cells)) == cells.size == nq * 256
    err = np.abs(got.astype(np.float64) - exp[cells]).max() / np.abs(exp).max()
    assert err <= 1e-5, err
