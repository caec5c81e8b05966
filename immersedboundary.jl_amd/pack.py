"""Partition pack: a flat on-disk form of one ``Partition`` (+ its ``Boundary`` chunks), so that the cold-path
``Domain(msh)`` construction (ImmersedBoundary.jl:534-790) runs once and every rank of a later run loads only what
it owns (the reference itself has no wire format).

One ``.npz`` per partition, arrays only (no pickled objects):
  meta          int64 [id, nd, nc, n_image, block_size, n_boundaries]
  centers, spacing              float32 (nc, nd)
  domain, image, image_in_domain int (global ids / local positions, 0-based like the host objects)
  owners_d, neighbors_d          int32 per dim d = 1..nd
  lo_off_d, lo_idx_d, hi_off_d, hi_idx_d   CSR of the left / right face accumulators (weights are 1/len, implicit)
  b<k>_name (uint8 bytes), b<k>_ghost, b<k>_proj, b<k>_normals, b<k>_idist, b<k>_gdist, b<k>_idom,
  b<k>_off, b<k>_idx, b<k>_w    boundary chunk k: fields of ImmersedBoundary.jl:406-414, interpolator as CSR
"""
from __future__ import annotations

import numpy as np

from .accumulator import Accumulator
from .domain import Boundary, Partition

FORMAT_VERSION = 1


def save_partition(path, part, boundaries=None):
    """``boundaries``: ``dom.boundaries`` layout, {name: {chunk id: Boundary}} -- the chunks stored with this partition."""
    nd = part.ndims
    out = dict(meta=np.array([part.id, nd, part.spacing.shape[0], len(part.image), part.block_size, 0, FORMAT_VERSION],
                             dtype=np.int64),
               centers=np.ascontiguousarray(part.centers, dtype=np.float32),
               spacing=np.ascontiguousarray(part.spacing, dtype=np.float32),
               domain=np.asarray(part.domain), image=np.asarray(part.image),
               image_in_domain=np.asarray(part.image_in_domain))
    for d in range(1, nd + 1):
        o, n = part.face_owners_neighbors[d]
        out[f"owners_{d}"], out[f"neighbors_{d}"] = np.asarray(o, np.int32), np.asarray(n, np.int32)
        for tag, right in (("lo", False), ("hi", True)):
            acc = part.face_accumulators[(d, right)]
            out[f"{tag}_off_{d}"], out[f"{tag}_idx_{d}"] = acc.off, acc.idx
    k = 0
    for name, chunks in (boundaries or {}).items():
        for cid, b in chunks.items():
            acc = b.image_interpolator
            out.update({f"b{k}_cid": np.array([cid], dtype=np.int64), f"b{k}_name": np.frombuffer(name.encode(), dtype=np.uint8), f"b{k}_ghost": np.asarray(b.ghost_indices),
                        f"b{k}_proj": b.projections, f"b{k}_normals": b.normals, f"b{k}_idist": b.image_distances,
                        f"b{k}_gdist": b.ghost_distances, f"b{k}_idom": np.asarray(b.image_domain),
                        f"b{k}_off": acc.off, f"b{k}_idx": acc.idx,
                        f"b{k}_w": acc.w if acc.w is not None else np.zeros(0, np.float32)})
            k += 1
    out["meta"][5] = k
    np.savez(path, **out)


def load_partition(path):
    """Returns ``(Partition, {name: {chunk id: Boundary}})``."""
    z = np.load(path)
    pid, nd, nc, n_image, bs, nb, ver = [int(v) for v in z["meta"]]
    if ver != FORMAT_VERSION:
        raise ValueError(f"partition pack version {ver}, expected {FORMAT_VERSION}")
    fon, accs = {}, {}
    for d in range(1, nd + 1):
        fon[d] = (z[f"owners_{d}"], z[f"neighbors_{d}"])
        nf = fon[d][0].size
        for tag, right in (("lo", False), ("hi", True)):
            off, idx = z[f"{tag}_off_{d}"], z[f"{tag}_idx_{d}"]
            ls = np.diff(off)
            w = np.repeat(np.where(ls > 0, np.float32(1.0) / np.maximum(ls, 1).astype(np.float32), np.float32(0)), ls)
            accs[(d, right)] = Accumulator(csr=(off, idx, w.astype(np.float32)), n_input=nf, first_index=True)
    part = Partition(pid, z["centers"], z["spacing"], accs, fon, z["domain"], z["image"], z["image_in_domain"],
                     block_size=bs)
    assert part.spacing.shape[0] == nc and len(part.image) == n_image
    bnd = {}
    for k in range(nb):
        name = bytes(z[f"b{k}_name"]).decode()
        w = z[f"b{k}_w"]
        acc = Accumulator(csr=(z[f"b{k}_off"], z[f"b{k}_idx"], w if w.size else None),
                          n_input=int(z[f"b{k}_idom"].size), first_index=True)
        bnd.setdefault(name, {})[int(z[f"b{k}_cid"][0])] = (Boundary(z[f"b{k}_ghost"], z[f"b{k}_proj"], z[f"b{k}_normals"], z[f"b{k}_idist"],
                                                  z[f"b{k}_gdist"], acc, z[f"b{k}_idom"]))
    return part, bnd
