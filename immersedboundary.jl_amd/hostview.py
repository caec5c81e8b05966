"""Host-only view of the block analysis libibhip runs in ``ibh_partition_create`` (2-D, 8x8 blocks).

``analyze2(part)`` returns the block table, the halo / end tables and the 2x2 block groups ("quads") of the
quad sweep as numpy arrays -- the same code path as the device upload, without a device.  Used by the CPU
tests of the library's host logic.
"""
import ctypes as C

import numpy as np

from ._lib import c_vp, call

BLOCK_DTYPE = np.dtype([("base", "<i4"), ("type", "<i4", (4,)), ("nb", "<i4", (4, 2)), ("sub", "<i4", (4,)),
                        ("h", "<f4", (2,)), ("rh", "<f4", (2,)), ("q", "<f4", (4,)), ("rt", "<f4", (4,)),
                        ("dt", "<i4")])
QUAD_DTYPE = np.dtype([("base", "<i4"), ("cls", "<u4"), ("rh", "<f4", (2,))])
(BLOCKS, HTAB, ETAB, FUSABLE, QUAD_DESC, QUAD_TAB, SINGLES, COUNTS, INFO, PAIR_DESC, PAIR_TAB, SINGLES2, QUAD_AUX,
 PAIR_AUX) = range(14)
SIDE_SAME, SIDE_MIRROR, SIDE_COARSE, SIDE_FINE, SIDE_GENERAL = range(5)


def analyze2(part):
    nd = part.ndims
    assert nd == 2
    nc = part.spacing.shape[0]
    spacing = np.asfortranarray(part.spacing, dtype=np.float32)
    nf = np.array([part.face_owners_neighbors[d + 1][0].shape[0] for d in range(nd)], dtype=np.int32)
    keep = []

    def parr(arrs):
        arrs = [np.ascontiguousarray(a, dtype=np.int32) for a in arrs]
        keep.append(arrs)
        return (c_vp * nd)(*[a.ctypes.data for a in arrs])

    owners = parr([part.face_owners_neighbors[d + 1][0] for d in range(nd)])
    neighbors = parr([part.face_owners_neighbors[d + 1][1] for d in range(nd)])
    loff = parr([part.face_accumulators[(d + 1, False)].off for d in range(nd)])
    lidx = parr([part.face_accumulators[(d + 1, False)].idx for d in range(nd)])
    roff = parr([part.face_accumulators[(d + 1, True)].off for d in range(nd)])
    ridx = parr([part.face_accumulators[(d + 1, True)].idx for d in range(nd)])
    iid = np.ascontiguousarray(part.image_in_domain, dtype=np.int32)
    dom = np.ascontiguousarray(part.domain, dtype=np.int32)
    h = c_vp()
    call("ibh_analyze2_host", C.byref(h), nc, spacing.ctypes.data_as(c_vp), nf.ctypes.data_as(c_vp), owners, neighbors,
         loff, lidx, roff, ridx, int(iid.size), iid.ctypes.data_as(c_vp), dom.ctypes.data_as(c_vp), 0)

    def get(what, dtype, set_=0):
        n = C.c_int64(0)
        call("ibh_host2d_get", h, what, set_, c_vp(None), 0, C.byref(n))
        buf = np.empty(n.value, dtype=np.uint8)
        if n.value:
            call("ibh_host2d_get", h, what, set_, buf.ctypes.data_as(c_vp), n.value, C.byref(n))
        return buf.view(dtype)

    try:
        out = dict(blocks=get(BLOCKS, BLOCK_DTYPE), htab=get(HTAB, np.int32).reshape(-1, 64),
                   etab=get(ETAB, np.int32).reshape(-1, 16), fusable=get(FUSABLE, np.uint8).astype(bool),
                   info=get(INFO, np.int64))
        for k, name in ((0, "all"), (1, "image")):
            cnt = get(COUNTS, np.int64, k)
            out[f"quads_{name}"] = dict(desc=get(QUAD_DESC, QUAD_DTYPE, k), tab=get(QUAD_TAB, np.int32, k).reshape(-1, 160),
                                        singles=get(SINGLES, np.int32, k), n_interior=int(cnt[2]),
                                        n_singles_interior=int(cnt[4]),
                                        # pair tiles among the single blocks (set 0 of one-partition domains) and what is left
                                        pair_desc=get(PAIR_DESC, QUAD_DTYPE, k),
                                        pair_tab=get(PAIR_TAB, np.int32, k).reshape(-1, 160),
                                        singles2=get(SINGLES2, np.int32, k),
                                        # companion rows: 32 end ids + per half-side the origin of arithmetic halo ids / -1
                                        aux=get(QUAD_AUX, np.int32, k).reshape(-1, 40),
                                        pair_aux=get(PAIR_AUX, np.int32, k).reshape(-1, 40))
            out.update(fuse_all=bool(cnt[5]), img_all_fz=bool(cnt[6]), nB1=int(cnt[7]))
    finally:
        call("ibh_host2d_destroy", h)
    return out
