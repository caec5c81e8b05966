"""Bit-exact indexing: the product's vectorised Domain builder against the literal restatement of
Domain(msh) in oracle/domain.py (ImmersedBoundary.jl:536-786), on meshes with 2:1 level jumps,
several partitions (skirts, mirror faces) and immersed + hypercube boundaries."""
import numpy as np
import pytest

import ibamd


def _cmp_partition(a, b):
    assert np.array_equal(a.domain, b.domain)
    assert np.array_equal(a.image, b.image)
    assert np.array_equal(a.image_in_domain, b.image_in_domain)
    assert np.array_equal(a.spacing, b.spacing)
    assert np.array_equal(a.centers, b.centers)
    for dim in range(1, a.ndims + 1):
        for k in (0, 1):
            assert np.array_equal(a.face_owners_neighbors[dim][k], b.face_owners_neighbors[dim][k])
        for r in (False, True):
            ia, wa = a.face_accumulators[(dim, r)].decompose()
            acc = b.face_accumulators[(dim, r)]
            assert acc.n_output == len(ia)
            flat_i = np.concatenate([x for x in ia if x is not None and len(x)])
            flat_w = np.concatenate([x for x in wa if x is not None and len(x)])
            assert np.array_equal(flat_i, acc.idx)
            assert np.array_equal(flat_w, acc.w)
            lens = np.array([0 if x is None else len(x) for x in ia])
            assert np.array_equal(lens, np.diff(acc.off))


def _cmp_boundaries(do, dp):
    assert set(do.boundaries) == set(dp.boundaries)
    for name in do.boundaries:
        assert set(do.boundaries[name]) == set(dp.boundaries[name])
        for k in do.boundaries[name]:
            a, b = do.boundaries[name][k], dp.boundaries[name][k]
            assert np.array_equal(a.ghost_indices, b.ghost_indices)          # ghost ids: bit-exact
            assert np.array_equal(a.image_domain, b.image_domain)
            assert np.allclose(a.projections, b.projections, atol=1e-6)
            assert np.allclose(a.normals, b.normals, atol=1e-5)
            assert np.array_equal(a.image_distances, b.image_distances)
            assert np.allclose(a.ghost_distances, b.ghost_distances, atol=1e-7)
            ia, wa = a.image_interpolator.decompose()
            assert np.array_equal(np.concatenate(ia), b.image_interpolator.idx)
            assert np.allclose(np.concatenate(wa), b.image_interpolator.w, atol=1e-5)


def test_advection_domain(adv_domains):
    dp, do = adv_domains
    fd, fo, fn = dp.faces
    of = np.array(do.faces)
    assert np.array_equal(of[:, 0], fd) and np.array_equal(of[:, 1], fo) and np.array_equal(of[:, 2], fn)
    assert set(dp.partitions) == set(do.partitions) and len(dp.partitions) == 3
    for k in do.partitions:
        _cmp_partition(do.partitions[k], dp.partitions[k])
    _cmp_boundaries(do, dp)


def test_rae2822_domain(rae_domains):
    dp, do = rae_domains
    assert len(dp) == 37120  # shipped test/rae2822.jl settings
    fd, fo, fn = dp.faces
    of = np.array(do.faces)
    assert np.array_equal(of[:, 0], fd) and np.array_equal(of[:, 1], fo) and np.array_equal(of[:, 2], fn)
    for k in do.partitions:
        _cmp_partition(do.partitions[k], dp.partitions[k])
    _cmp_boundaries(do, dp)


def test_only_builds_requested_partition(adv_mesh):
    full = ibamd.Domain(adv_mesh, max_partition_size=4096, boundaries=False)
    one = ibamd.Domain(adv_mesh, max_partition_size=4096, boundaries=False, only=[2])
    assert list(one.partitions) == [2]
    for k in full.partitions:
        assert np.array_equal(full.domains[k], one.domains[k])
    assert np.array_equal(full.partitions[2].face_owners_neighbors[1][0], one.partitions[2].face_owners_neighbors[1][0])


def test_empty_and_ragged_accumulators():
    acc = ibamd.Accumulator([[0, 1], [], [2]], [[0.5, 0.5], [], [1.0]], n_input=3)
    assert np.array_equal(acc.off, [0, 2, 2, 3])
    st = acc.stencils
    assert set(st) == {2, 0, 1}
    assert st[0][1].shape == (0, 1)
    with pytest.raises(TypeError):
        acc(np.zeros(3))  # no CPU evaluation path


def test_multigrid_structure(adv_mesh_coarse):
    dom = ibamd.Domain(adv_mesh_coarse, hypercube_families=[("outlet", [(1, True), (2, True)])])
    coarse_doms, prolongators, coarseners = ibamd.multigrid(dom)   # reference return order (:1406)
    assert [d.mesh.block_size for d in coarse_doms] == [4, 2, 1]
    n = len(dom)
    for lvl, (cd, pr, co) in enumerate(zip(coarse_doms, prolongators, coarseners)):
        assert len(cd) * 4 ** (lvl + 1) == n
        assert co.n_output == len(cd) and pr.n_input == len(cd)
        # IDW coarsener = mean of the 4 children: 4 equal weights
        assert np.all(np.diff(co.off) == 4) and np.allclose(co.w, 0.25, atol=1e-6)


def test_partition_pack_round_trip(tmp_path):
    """save_partition / load_partition (ibamd.pack): every array of the Partition and of its Boundary chunks comes
    back identical, accumulators included (weights 1/len are implicit in the file)."""
    from ibamd.pack import load_partition, save_partition
    from conftest import ADV_FAMILIES, advection_mesh
    dom = ibamd.Domain(advection_mesh(2e-2), hypercube_families=ADV_FAMILIES, max_partition_size=2048)
    part = dom.partitions[2]
    path = str(tmp_path / "p2.npz")
    save_partition(path, part, dom.boundaries)
    back, bnd = load_partition(path)
    assert back.id == part.id and back.block_size == part.block_size
    for name in ("centers", "spacing", "domain", "image", "image_in_domain"):
        assert np.array_equal(getattr(back, name), getattr(part, name)), name
    for d in (1, 2):
        for a, b in zip(back.face_owners_neighbors[d], part.face_owners_neighbors[d]):
            assert np.array_equal(a, b)
        for right in (False, True):
            x, y = back.face_accumulators[(d, right)], part.face_accumulators[(d, right)]
            assert np.array_equal(x.off, y.off) and np.array_equal(x.idx, y.idx) and np.array_equal(x.w, y.w)
    assert set(bnd) == set(dom.boundaries)
    for name, chunks in dom.boundaries.items():
        assert set(bnd[name]) == set(chunks)
        for x, y in ((bnd[name][c], chunks[c]) for c in chunks):
            for f in ("ghost_indices", "projections", "normals", "image_distances", "ghost_distances", "image_domain"):
                assert np.array_equal(getattr(x, f), getattr(y, f)), f
            assert np.array_equal(x.image_interpolator.idx, y.image_interpolator.idx)
            assert np.array_equal(x.image_interpolator.w, y.image_interpolator.w)
