"""Edge cases through the C ABI: empty and ragged inputs, strided / non-column-major fields, wide fields,
error reporting instead of crashes, partitions without block information, index_base = 1 (Julia callers)."""
import ctypes as C

import numpy as np
import pytest
import torch

import ibamd
from conftest import rel_inf, seeded_field
from ibamd import _lib
from ibamd import backend as B
from oracle import domain as od

pytestmark = pytest.mark.gpu
f32 = np.float32


def test_empty_accumulator_rows_and_zero_sized():
    acc = ibamd.Accumulator([[], [0, 2], []], [[], [0.25, 0.75], []], n_input=3)
    out = ibamd.to_host(ibamd.to_backend(acc)(ibamd.hip(np.array([4.0, 5.0, 8.0], dtype=f32))))
    assert np.array_equal(out, np.array([0.0, 7.0, 0.0], dtype=f32))      # zero-length bucket -> 0 (accumulator.jl:83)
    empty = ibamd.Accumulator([], [], n_input=3)
    assert ibamd.to_backend(empty)(ibamd.hip(np.zeros(3, f32))).shape[0] == 0


def test_unweighted_accumulator():
    acc = ibamd.Accumulator([[0, 1, 2], [2]], None, n_input=3)
    out = ibamd.to_host(ibamd.to_backend(acc)(ibamd.hip(np.array([[1, 10], [2, 20], [3, 30]], dtype=f32))))
    assert np.array_equal(out, np.array([[6, 60], [3, 30]], dtype=f32))


@pytest.mark.parametrize("nv", [1, 3, 6, 8, 11])
@pytest.mark.parametrize("ragged", [False, True])
def test_accumulate_diff_add_is_the_three_operations(nv, ragged):
    """``out .+= acc(a .- b)`` in one call (``ibh_accumulate_diff_add``: the prolongation step of FAS!, solver.jl:76) against
    ``out += acc(a - b)`` through the separate kernels, bit for bit: rows of eight entries (the packed-donor form for 2..8
    fields: many rows per donor), ragged rows incl. empty ones, and more fields than one pass takes."""
    rng = np.random.default_rng(100 + nv)
    n_in, n_out = 500, 4000
    cnt = rng.integers(0, 12, n_out) if ragged else np.full(n_out, 8)
    idx = [rng.integers(0, n_in, c).tolist() for c in cnt]
    w = [rng.uniform(0.05, 1.0, c).astype(f32).tolist() for c in cnt]
    acc = ibamd.to_backend(ibamd.Accumulator(idx, w, n_input=n_in))
    a, b = rng.standard_normal((n_in, nv)).astype(f32), rng.standard_normal((n_in, nv)).astype(f32)
    o0 = rng.standard_normal((n_out, nv)).astype(f32)
    if nv == 1:
        a, b, o0 = a[:, 0], b[:, 0], o0[:, 0]
    da, db = ibamd.hip(a), ibamd.hip(b)
    got = ibamd.hip(o0)
    acc.diff_add(got, da, db)
    ref = ibamd.hip(o0)
    ref += acc(da - db)
    assert torch.equal(got, ref)
    assert not torch.equal(got, ibamd.hip(o0))


def test_row_major_and_strided_fields_are_accepted(adv_domains):
    dp, do = adv_domains
    part, opart = dp.partitions[2], do.partitions[2]
    dpart = ibamd.to_backend(part, ibamd.hip)
    u = seeded_field(opart.centers, nv=3)
    exp = od.cell_gradient(opart, u, 1)
    rowmajor = torch.from_numpy(u).cuda()                       # (n, 3) C-order: stride (3, 1)
    assert rel_inf(ibamd.to_host(ibamd.cell_gradient(dpart, rowmajor, 1)), exp) <= 1e-6
    wide = ibamd.hip(np.concatenate([u, u], axis=1))[:, 1:4]    # column slice of a wider column-major array
    exp2 = od.cell_gradient(opart, np.ascontiguousarray(np.concatenate([u, u], axis=1)[:, 1:4]), 1)
    assert rel_inf(ibamd.to_host(ibamd.cell_gradient(dpart, wide, 1)), exp2) <= 1e-6
    with pytest.raises(ValueError):
        ibamd.cell_gradient(dpart, rowmajor[:-1], 1)           # wrong number of rows
    with pytest.raises(TypeError):
        ibamd.cell_gradient(dpart, rowmajor.double(), 1)       # Float32 only
    with pytest.raises(_lib.IbhError):
        ibamd.cell_gradient(dpart, ibamd.hip(u), 3)            # dim out of range -> error code, not a crash


def test_wide_field(adv_domains):
    dp, do = adv_domains
    part, opart = dp.partitions[1], do.partitions[1]
    dpart = ibamd.to_backend(part, ibamd.hip)
    u = seeded_field(opart.centers, nv=17)
    assert rel_inf(ibamd.to_host(ibamd.at_faces(dpart, ibamd.hip(u), 2)), od.at_faces(opart, u, 2)) <= 1e-6


def _raw_create(part, block_size, index_base, with_domain=True):
    nd, nc = 2, part.spacing.shape[0]
    keep = []

    def parr(arrs):
        arrs = [np.ascontiguousarray(a + index_base, dtype=np.int32) for a in arrs]
        keep.append(arrs)
        return (B.c_vp * nd)(*[a.ctypes.data for a in arrs])
    fon, fa = part.face_owners_neighbors, part.face_accumulators
    owners, neigh = parr([fon[d][0] for d in (1, 2)]), parr([fon[d][1] for d in (1, 2)])
    loff, lidx = parr([fa[(d, False)].off for d in (1, 2)]), parr([fa[(d, False)].idx for d in (1, 2)])
    roff, ridx = parr([fa[(d, True)].off for d in (1, 2)]), parr([fa[(d, True)].idx for d in (1, 2)])
    nf = np.array([fon[1][0].size, fon[2][0].size], dtype=np.int32)
    sp = np.asfortranarray(part.spacing)
    iid = np.ascontiguousarray(part.image_in_domain + index_base, dtype=np.int32)
    dom = np.ascontiguousarray(part.domain + index_base, dtype=np.int32)
    h = B.c_vp()
    B._dev()
    B.call("ibh_partition_create", C.byref(h), nd, nc, sp.ctypes.data_as(B.c_vp), B.c_vp(None), nf.ctypes.data_as(B.c_vp),
           owners, neigh, loff, lidx, roff, ridx, int(iid.size), B._hptr(iid),
           B._hptr(dom) if with_domain else B.c_vp(None), block_size, index_base)
    return h


def test_one_based_indices_and_no_block_info(adv_domains):
    """Julia callers pass 1-based arrays; a partition created without `domain` has no block path."""
    dp, do = adv_domains
    part, opart = dp.partitions[2], do.partitions[2]
    u = seeded_field(opart.centers)
    Cc = np.ones((u.shape[0], 2), dtype=f32)
    ref = ibamd.to_host(ibamd.residual_advection(ibamd.to_backend(part, ibamd.hip), ibamd.hip(u), ibamd.hip(Cc), flags=1))
    ud, Cd = ibamd.hip(u), ibamd.hip(Cc)
    for base, with_dom in ((1, True), (0, False)):
        h = _raw_create(part, 8, base, with_dom)
        info = (C.c_int64 * 8)()
        B.call("ibh_partition_info", h, info, 8)
        assert (info[0] > 0) == with_dom
        out = torch.zeros(u.shape[0], dtype=torch.float32, device="cuda")
        B._stream()
        B.call("ibh_residual_advection", h, B._ptr(ud), B._ptr(Cd), u.shape[0], B._ptr(out), 16 if with_dom else 0)
        assert np.array_equal(ibamd.to_host(out), ref)       # literal block path / face lists: same bits
        with pytest.raises(_lib.IbhError):
            B.call("ibh_residual_advection", h, B._ptr(ud), B._ptr(Cd), u.shape[0], B._ptr(out), 32 | 64)
        B.call("ibh_partition_destroy", h)


def test_bad_indices_are_rejected(adv_domains):
    dp, _ = adv_domains
    part = dp.partitions[1]
    bad = ibamd.Accumulator(csr=(np.array([0, 1], np.int32), np.array([7], np.int32), np.array([1.0], f32)), n_input=3)
    with pytest.raises(_lib.IbhError):
        ibamd.to_backend(bad)
    lib = _lib.load()
    assert lib.ibh_accumulate(None, None, 1, 1, None, 1) != 0 and b"null" in lib.ibh_last_error()


def test_image_only_flag(adv_domains):
    dp, do = adv_domains
    part = dp.partitions[2]
    dpart = ibamd.to_backend(part, ibamd.hip)
    u = seeded_field(part.centers)
    Cc = np.ones((u.shape[0], 2), dtype=f32)
    full = ibamd.to_host(ibamd.residual_advection(dpart, ibamd.hip(u), ibamd.hip(Cc), flags=1))
    out = torch.full((u.shape[0],), 123.0, dtype=torch.float32, device="cuda")
    ibamd.residual_advection(dpart, ibamd.hip(u), ibamd.hip(Cc), out=out, flags=1 | 2)
    got = ibamd.to_host(out)
    img = part.image_in_domain
    assert np.array_equal(got[img], full[img])
    skirt = np.ones(u.shape[0], bool)
    skirt[img] = False
    assert np.all(got[skirt] == 123.0)                         # skirt rows untouched (ImmersedBoundary.jl:857-859)


def test_inplace_arguments_must_be_column_major(adv_domains):
    """A row-major (n, nv) tensor handed to something that writes it (out=, impose_bc args, halo fields) raises instead
    of being copied silently (the update would go to the temporary); a wrong-sized `out` is rejected too."""
    import torch
    dp, _ = adv_domains
    part = next(iter(dp.partitions.values()))
    dpart = ibamd.to_backend(part, ibamd.hip)
    n = dpart.nc
    u = ibamd.hip(np.zeros(n, dtype=np.float32))
    Cc = ibamd.hip(np.ones((n, 2), dtype=np.float32))
    with pytest.raises(ValueError):
        ibamd.residual_advection(dpart, u, Cc, out=torch.zeros(n - 1, dtype=torch.float32, device="cuda"))
    with pytest.raises(TypeError):
        ibamd.residual_advection(dpart, u, Cc, out=torch.zeros(2 * n, dtype=torch.float32, device="cuda")[::2])
    P = ibamd.hip(np.ones((n, 4), dtype=np.float32))
    with pytest.raises(TypeError):
        ibamd.residual_euler_hll(dpart, P, out=torch.zeros((n, 4), dtype=torch.float32, device="cuda"))  # row-major
    row_major = torch.zeros((len(dp), 3), dtype=torch.float32, device="cuda")
    with pytest.raises(TypeError):
        ibamd.impose_bc(lambda bdry, a: a, dp, "outlet", row_major)
