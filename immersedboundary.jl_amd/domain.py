"""Host-side ``Domain`` / ``Partition`` / ``Boundary`` (cold path, builds the hot path's inputs).

Mirror of the reference's data contracts and of ``Domain(msh; ...)``
(/root/reference/src/ImmersedBoundary.jl:383-490, :536-786) with the same
keyword names and meaning.  Where the reference walks KD-trees, ``Dict``s and
``Set``s cell by cell, this builder works on the block lattice with vectorised
numpy (O(N)); tests check it index-for-index against the literal restatement
in oracle/domain.py.  Canonical orders (the reference's global face order is
thread-dependent): faces are registered in single-thread order
with candidates visited in ascending cell index.

Indices are 0-based, "no cell" = -1; ``dim`` in dict keys and operator calls is
1-based like the reference.
"""
from __future__ import annotations

import numpy as np
from scipy.spatial import cKDTree

from .accumulator import Accumulator
from .mesher import Mesh, get_cells, proj2simplex

f32 = np.float32


class Partition:
    """ImmersedBoundary.jl:383-392."""

    def __init__(self, id, centers, spacing, face_accumulators, face_owners_neighbors, domain, image,
                 image_in_domain, block_size=0):
        self.id = id
        self.centers = centers
        self.spacing = spacing
        self.face_accumulators = face_accumulators
        self.face_owners_neighbors = face_owners_neighbors
        self.domain = domain
        self.image = image
        self.image_in_domain = image_in_domain
        self.block_size = block_size  # not in the reference struct: mesh.block_size, lets libibhip find blocks

    @property
    def ndims(self):
        return self.centers.shape[1]


class Boundary:
    """ImmersedBoundary.jl:406-414 (fields) -- built by ``_make_boundary`` (:422-448)."""

    def __init__(self, ghost_indices, projections, normals, image_distances, ghost_distances,
                 image_interpolator, image_domain):
        self.ghost_indices = ghost_indices
        self.projections = projections
        self.normals = normals
        self.image_distances = image_distances
        self.ghost_distances = ghost_distances
        self.image_interpolator = image_interpolator
        self.image_domain = image_domain


def _colsum(a):
    s = a[0].copy()
    for k in range(1, a.shape[0]):
        s = s + a[k]
    return s


def _rowsum(a):
    s = a[:, 0].copy()
    for k in range(1, a.shape[1]):
        s = s + a[:, k]
    return s


# ---------------------------------------------------------------------------
# faces on the block lattice
# ---------------------------------------------------------------------------
def _block_lattice(msh):
    """Integer (level, coords) of every leaf block; requires a cubic root split by 2 per level."""
    bo = msh.block_origins.astype(np.float64)
    bw = msh.block_widths.astype(np.float64)
    rw = msh.widths.astype(np.float64)
    ro = msh.origin.astype(np.float64)
    lev = np.rint(np.log2(rw[:, None] / bw)).astype(np.int64)
    if not np.all(lev == lev[0:1]) or not np.allclose(rw[:, None] / bw, 2.0 ** lev):
        raise NotImplementedError("block lattice builder needs a root hypercube split dyadically and isotropically")
    lev = lev[0]
    coords = np.rint((bo - ro[:, None]) / bw).astype(np.int64)
    return lev, coords


def _neighbor_blocks(lev, coords):
    """For every block and dim: leaf blocks across its HIGH side, plus root-boundary flags.

    Returns arrays (left_block, right_block, dim0) of abutting block pairs, and boolean
    arrays low_bdry[nd, nb], high_bdry[nd, nb].
    """
    nd, nb = coords.shape
    table = {}
    for b in range(nb):
        table[(int(lev[b]),) + tuple(int(x) for x in coords[:, b])] = b
    L, R, Dm = [], [], []
    low_b = np.zeros((nd, nb), dtype=bool)
    high_b = np.zeros((nd, nb), dtype=bool)
    ntan = 2 ** (nd - 1)
    for b in range(nb):
        l = int(lev[b])
        X = [int(x) for x in coords[:, b]]
        n = 1 << l
        for d in range(nd):
            if X[d] == 0:
                low_b[d, b] = True
            if X[d] + 1 >= n:
                high_b[d, b] = True
                continue
            Y = list(X)
            Y[d] += 1
            found = None
            for k in range(l + 1):
                key = (l - k,) + tuple(y >> k for y in Y)
                if key in table:
                    found = table[key]
                    break
            if found is not None:
                L.append(b); R.append(found); Dm.append(d)
                continue
            # finer: descend into the children touching our face
            stack = [(l, tuple(Y))]
            while stack:
                ll, YY = stack.pop()
                for m in range(ntan):
                    child = []
                    bit = 0
                    for dd in range(nd):
                        if dd == d:
                            child.append(2 * YY[dd])
                        else:
                            child.append(2 * YY[dd] + ((m >> bit) & 1))
                            bit += 1
                    key = (ll + 1,) + tuple(child)
                    if key in table:
                        L.append(b); R.append(table[key]); Dm.append(d)
                    else:
                        stack.append((ll + 1, tuple(child)))
    return (np.array(L, dtype=np.int64), np.array(R, dtype=np.int64), np.array(Dm, dtype=np.int64), low_b, high_b)


def build_faces(msh, centers, widths):
    """All faces ``(dim 1-based, owner, neighbor)`` in canonical order.

    Reference rules: ImmersedBoundary.jl:63-132 (``octree2faces``: overlap test with 1 % tolerance,
    candidates within 3.1 half-diagonals of the LEFT cell, registered from the left cell) and
    :150-184 (``hcube_faces``).
    """
    nd = centers.shape[0]
    bs = msh.block_size
    npb = bs ** nd
    nb = msh.nblocks
    lev, coords = _block_lattice(msh)
    strides = np.array([bs ** k for k in range(nd)], dtype=np.int64)

    # intra-block faces: cell (.., i_d, ..) -> (.., i_d + 1, ..)
    loc = np.arange(npb, dtype=np.int64)
    fo_parts, fn_parts, fd_parts = [], [], []
    for d in range(nd):
        i_d = (loc // strides[d]) % bs
        src = loc[i_d < bs - 1]
        o = (np.arange(nb, dtype=np.int64)[:, None] * npb + src[None, :]).ravel()
        fo_parts.append(o)
        fn_parts.append(o + strides[d])
        fd_parts.append(np.full(o.size, d + 1, dtype=np.int8))

    # inter-block candidate pairs
    Lb, Rb, Dm, _, _ = _neighbor_blocks(lev, coords)
    nt = bs ** (nd - 1)
    tt = np.arange(nt, dtype=np.int64)
    cand_o, cand_n, cand_d = [], [], []
    for d in range(nd):
        sel = Dm == d
        if not np.any(sel):
            continue
        lb, rb = Lb[sel], Rb[sel]
        ll, lr = lev[lb], lev[rb]
        fine_is_left = ll >= lr
        fb = np.where(fine_is_left, lb, rb)
        cb = np.where(fine_is_left, rb, lb)
        delta = np.abs(ll - lr)
        tdims = [k for k in range(nd) if k != d]
        floc = np.zeros((lb.size, nt), dtype=np.int64)
        cloc = np.zeros((lb.size, nt), dtype=np.int64)
        for q, k in enumerate(tdims):
            t_k = (tt // bs ** q) % bs
            gF = coords[k, fb][:, None] * bs + t_k[None, :]
            cl = (gF >> delta[:, None]) - coords[k, cb][:, None] * bs
            floc += t_k[None, :] * strides[k]
            cloc += cl * strides[k]
        f_norm = np.where(fine_is_left, bs - 1, 0)[:, None] * strides[d]
        c_norm = np.where(fine_is_left, 0, bs - 1)[:, None] * strides[d]
        fcell = fb[:, None] * npb + floc + f_norm
        ccell = cb[:, None] * npb + cloc + c_norm
        left = np.where(fine_is_left[:, None], fcell, ccell).ravel()
        right = np.where(fine_is_left[:, None], ccell, fcell).ravel()
        cand_o.append(left)
        cand_n.append(right)
        cand_d.append(np.full(left.size, d, dtype=np.int64))
    if cand_o:
        co = np.concatenate(cand_o)
        cn = np.concatenate(cand_n)
        cd = np.concatenate(cand_d)
        # the reference's float tests on each candidate (i = left cell, j = right cell)
        origins = centers - widths / f32(2)
        maxs = origins + widths
        fo_ = np.maximum(origins[:, co], origins[:, cn])
        fw = np.minimum(maxs[:, co], maxs[:, cn]) - fo_
        tol = f32(0.01) * fw.max(axis=0)
        ok = ((fw < tol).sum(axis=0) == 1) & ((fw < -tol).sum(axis=0) == 0)
        ok &= np.argmin(fw, axis=0) == cd
        ok &= ~(origins[cd, cn] < origins[cd, co])
        radii = np.sqrt(_colsum(widths * widths)) / f32(2)
        diff = centers[:, co].astype(np.float64) - centers[:, cn].astype(np.float64)
        dist = np.sqrt((diff * diff).sum(axis=0))
        ok &= dist <= (radii[co] * f32(3.1)).astype(np.float64)
        fo_parts.append(co[ok])
        fn_parts.append(cn[ok])
        fd_parts.append((cd[ok] + 1).astype(np.int8))
    fo = np.concatenate(fo_parts)
    fn = np.concatenate(fn_parts)
    fd = np.concatenate(fd_parts)
    order = np.lexsort((fn, fo))
    fo, fn, fd = fo[order], fn[order], fd[order]

    # hypercube faces (:150-184), appended after the interior faces
    origins = centers - widths / f32(2)
    ho, hn, hd = [], [], []
    for dim in range(nd):
        idxs = np.nonzero(np.abs(origins[dim] - msh.origin[dim]) < widths[dim] * f32(0.01))[0]
        ho.append(np.full(idxs.size, -1, dtype=np.int64)); hn.append(idxs); hd.append(np.full(idxs.size, dim + 1, np.int8))
        idxs = np.nonzero(np.abs(origins[dim] + widths[dim] - msh.origin[dim] - msh.widths[dim])
                          < widths[dim] * f32(0.01))[0]
        ho.append(idxs); hn.append(np.full(idxs.size, -1, dtype=np.int64)); hd.append(np.full(idxs.size, dim + 1, np.int8))
    fo = np.concatenate([fo] + ho).astype(np.int32)
    fn = np.concatenate([fn] + hn).astype(np.int32)
    fd = np.concatenate([fd] + hd)
    return fd, fo, fn


# ---------------------------------------------------------------------------
# partitions (ImmersedBoundary.jl:594-703)
# ---------------------------------------------------------------------------
def partition_domain_mask(image, ncells, fo, fn, skirt_depth):
    """Skirt growth (ImmersedBoundary.jl:609-621): image cells + `skirt_depth` rings of face neighbours."""
    mask = np.zeros(ncells + 1, dtype=bool)  # slot ncells = "no cell"
    mask[image] = True
    fo_ = np.where(fo < 0, ncells, fo)
    fn_ = np.where(fn < 0, ncells, fn)
    for _ in range(skirt_depth):
        touch = mask[fo_] | mask[fn_]
        mask[fo_[touch]] = True
        mask[fn_[touch]] = True
        mask[ncells] = False
    return mask, fo_, fn_


def _build_partition(ipart, image, ncells, fd, fo, fn, centers, widths, skirt_depth, block_size):
    nd = centers.shape[0]
    mask, fo_, fn_ = partition_domain_mask(image, ncells, fo, fn, skirt_depth)
    domain = np.nonzero(mask[:ncells])[0].astype(np.int32)
    g2l = np.full(ncells + 1, -1, dtype=np.int32)
    g2l[domain] = np.arange(domain.size, dtype=np.int32)
    touch = mask[fo_] | mask[fn_]
    fids = np.nonzero(touch)[0]
    lo = g2l[fo_[fids]]
    ln = g2l[fn_[fids]]
    big = np.int64(domain.size)
    first = np.minimum(np.where(lo < 0, big, lo), np.where(ln < 0, big, ln))
    order = np.lexsort((fids, first))  # (first referencing cell, global face id)
    fids, lo, ln = fids[order], lo[order], ln[order]
    fdim = fd[fids]
    face_accumulators = {}
    face_owners_neighbors = {}
    nc = domain.size
    for dim in range(1, nd + 1):
        sel = fdim == dim
        o = lo[sel].copy()
        n = ln[sel].copy()
        add_right = o >= 0
        o = np.where(o < 0, n, o)
        add_left = n >= 0
        n = np.where(n < 0, o, n)
        # NB: as in the reference, `add_left` is evaluated after the o==0 substitution (:653-660)
        k = np.arange(o.size, dtype=np.int32)
        face_owners_neighbors[dim] = (o.astype(np.int32), n.astype(np.int32))
        for isright, rows, add in ((False, n, add_left), (True, o, add_right)):
            r = rows[add]
            kk = k[add]
            srt = np.argsort(r, kind="stable")
            cnt = np.bincount(r, minlength=nc)
            off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
            w = (f32(1.0) / cnt.astype(f32))[r[srt]] if r.size else np.zeros(0, f32)
            face_accumulators[(dim, isright)] = Accumulator(
                csr=(off, kk[srt], w.astype(f32)), first_index=True, n_input=o.size)
    image_in_domain = g2l[image].astype(np.int32)
    return Partition(ipart, np.ascontiguousarray(centers[:, domain].T), np.ascontiguousarray(widths[:, domain].T),
                     face_accumulators, face_owners_neighbors, domain, image.astype(np.int32), image_in_domain,
                     block_size=block_size)


# ---------------------------------------------------------------------------
# ghosts / boundaries (ImmersedBoundary.jl:194-326, :422-476)
# ---------------------------------------------------------------------------
def _project_2d(dfield, X, R):
    """Vectorised ``projection(dfield, x, R)`` for segment surfaces (mesher.jl:778-801, :549-567)."""
    n = X.shape[1]
    idx, d = dfield.nn(X)
    P = dfield.centers[:, idx].copy()
    dcur = d.copy()
    cand = dfield.tree.query_ball_point(np.ascontiguousarray(X.T, dtype=np.float64), r=R.astype(np.float64))
    lens = np.array([len(c) for c in cand])
    use = R > d
    lens = np.where(use, lens, 0)
    if lens.sum() == 0:
        return P
    rows = np.repeat(np.arange(n), lens)
    sidx = np.concatenate([np.sort(np.asarray(c, dtype=np.int64)) for c, u_ in zip(cand, use) if u_ and len(c)])
    pts = dfield.stl.points
    simp = dfield.stl.simplices - 1
    p0 = pts[:, simp[0, sidx]]
    p1 = pts[:, simp[1, sidx]]
    x = X[:, rows]
    eps_ = f32(1e-14)
    u = p1 - p0
    num = ((x - p0) * u)
    num = num[0] + num[1]
    den = (u * u)
    den = den[0] + den[1] + eps_
    xi = num / den
    proj = p0 + u * xi
    proj = np.where(xi < -eps_, p0, proj)
    proj = np.where(xi > 1.0 + eps_, p1, proj)
    dd = proj - x
    dist = np.sqrt(dd[0] * dd[0] + dd[1] * dd[1])
    # sequential "first strict improvement" per row == first minimum below the nn distance
    starts = np.concatenate([[0], np.cumsum(lens)])
    for r in np.nonzero(lens)[0]:
        s, e = starts[r], starts[r + 1]
        seg = dist[s:e]
        k = int(np.argmin(seg))
        if seg[k] < dcur[r]:
            P[:, r] = proj[:, s + k]
    return P


def _seg_proj(p0, p1, x):
    """``proj2simplex`` for 2-vertex simplices, column-wise (mesher.jl:549-567): arrays ``(nd, m)``."""
    eps_ = f32(1e-14)
    u = p1 - p0
    a = (x - p0) * u
    b = u * u
    num, den = a[0], b[0]
    for q in range(1, a.shape[0]):
        num = num + a[q]
        den = den + b[q]
    xi = num / (den + eps_)
    proj = p0 + u * xi
    proj = np.where(xi < -eps_, p0, proj)
    proj = np.where(xi > 1.0 + eps_, p1, proj)
    return proj


def _dist_cols(a, b):
    d = a - b
    s = d[0] * d[0]
    for q in range(1, d.shape[0]):
        s = s + d[q] * d[q]
    return np.sqrt(s)


def _tri_cache(dfield):
    """Per triangle of a surface: the vertices, the 3 x 2 edge matrix, its pseudo-inverse (one SVD per triangle) and
    the largest vertex distance from the centre (with a safety margin, Float64)."""
    pts = dfield.stl.points
    simp = dfield.stl.simplices - 1
    T0, T1, T2 = pts[:, simp[0]], pts[:, simp[1]], pts[:, simp[2]]
    cache = getattr(dfield, "_tri_cache", None)
    if cache is None:
        M = np.stack([T1 - T0, T2 - T0], axis=2).transpose(1, 0, 2)          # (nt, 3, 2)
        pinvM = np.linalg.pinv(M)                                            # (nt, 2, 3)
        ctr = dfield.centers
        rmax = np.maximum(np.maximum(_dist_cols(T0, ctr), _dist_cols(T1, ctr)), _dist_cols(T2, ctr))
        cache = dfield._tri_cache = (M, pinvM, rmax.astype(np.float64) * 1.0001 + 1e-12)
    return T0, T1, T2, cache


def _project_3d(dfield, X, R, chunk=20_000):
    """Vectorised ``projection(dfield, x, R)`` for triangle surfaces (mesher.jl:778-801 with ``proj2simplex``
    :544-596): the (ghost, candidate triangle) pairs of a chunk of ghosts at once -- ``pinv`` of the 3 x 2 edge
    matrix once per triangle, the three edge projections where the foot point falls outside the triangle (first
    strict minimum, faces in the order of ``simplex_faces``), then per ghost the first candidate, in ascending simplex
    order, that strictly improves on the nearest-centre distance.  A candidate is dropped beforehand when it cannot
    win: |x - centre| - (largest vertex distance from the centre) >= nearest-centre distance -- the reference replaces
    the nearest centre only by a strictly closer point, so the result is the same.  Agrees with the per-ghost loop to
    the last bit or two of the ``pinv`` product (tests/test_3d.py checks the ghost sets and the projections)."""
    n = X.shape[1]
    idx, d = dfield.nn(X)
    P = dfield.centers[:, idx].copy()
    if n == 0:
        return P
    eps_ = f32(1e-14)
    T0, T1, T2, cache = _tri_cache(dfield)
    M, pinvM, rmax = cache
    rmax_all = float(rmax.max())
    use = R > d
    Xt = np.ascontiguousarray(X.T, dtype=np.float64)
    for c0 in range(0, n, chunk):
        c1 = min(c0 + chunk, n)
        sel = np.nonzero(use[c0:c1])[0] + c0
        if sel.size == 0:
            continue
        # a triangle can only win if |x - centre| < d + rmax (see the docstring): search no farther than that
        rq = np.minimum(R[sel].astype(np.float64), d[sel].astype(np.float64) + rmax_all)
        cand = dfield.tree.query_ball_point(Xt[sel], r=rq, return_sorted=True, workers=-1)
        lens = np.fromiter((len(c) for c in cand), dtype=np.int64, count=sel.size)
        if lens.sum() == 0:
            continue
        rows = np.repeat(sel, lens)
        tri = np.concatenate([np.asarray(c, dtype=np.int64) for c in cand if len(c)])
        x = X[:, rows]
        # exact pruning (see docstring); the bound is evaluated in Float64 with a safety margin on rmax
        dc = _dist_cols(x.astype(np.float64), dfield.centers[:, tri].astype(np.float64))
        keep = dc - rmax[tri] < d[rows].astype(np.float64)
        rows, tri, x = rows[keep], tri[keep], x[:, keep]
        if rows.size == 0:
            continue
        p0, p1, p2 = T0[:, tri], T1[:, tri], T2[:, tri]
        rhs = (x - p0).T
        Pv = pinvM[tri]
        xi = Pv[:, :, 0] * rhs[:, 0:1] + Pv[:, :, 1] * rhs[:, 1:2] + Pv[:, :, 2] * rhs[:, 2:3]   # (m, 2)
        Mt = M[tri]
        pr = p0 + (Mt[:, :, 0] * xi[:, 0:1] + Mt[:, :, 1] * xi[:, 1:2]).T
        outside = (xi[:, 0] < -eps_) | (xi[:, 1] < -eps_) | ((xi[:, 0] + xi[:, 1]) > 1.0 + eps_)
        if outside.any():
            o = np.nonzero(outside)[0]
            xo = x[:, o]
            best = np.empty((3, o.size), dtype=X.dtype)
            bd = np.full(o.size, np.inf, dtype=np.float64)
            # simplex_faces: the simplex without vertex i, i = 1, 2, 3
            for (a, b) in ((p1, p2), (p0, p2), (p0, p1)):
                q = _seg_proj(a[:, o], b[:, o], xo)
                dq = _dist_cols(q, xo)
                take = dq < bd
                best[:, take] = q[:, take]
                bd = np.where(take, dq, bd)
            pr[:, o] = best
        dist = _dist_cols(pr, x)
        # per ghost: candidates in ascending simplex order, the first one at the minimum distance, if below d
        order = np.lexsort((tri, dist, rows))
        rs = rows[order]
        first = order[np.concatenate([[True], rs[1:] != rs[:-1]])]
        better = dist[first] < d[rows[first]]
        P[:, rows[first][better]] = pr[:, first[better]]
    return P


def ghosts_and_projections(dfield, centers, widths, ghost_layer_ratio=f32(1.5)):
    """ImmersedBoundary.jl:194-230."""
    ratio = f32(ghost_layer_ratio)
    diams = np.sqrt(_colsum(widths * widths))
    _, dists = dfield.nn(centers)
    ghosts = np.nonzero(dists <= diams * ratio * f32(2))[0].astype(np.int32)
    Xg = centers[:, ghosts]
    Rg = diams[ghosts] * ratio * f32(2)
    if centers.shape[0] == 2:
        projs = _project_2d(dfield, Xg, Rg).astype(centers.dtype)
    else:
        # A candidate whose nearest simplex centre is farther than the ghost-layer width plus the largest simplex
        # radius cannot pass the distance test below whatever its projection is: it keeps the nearest centre.
        rmax_all = float(_tri_cache(dfield)[3][2].max())
        can = dists[ghosts].astype(np.float64) - rmax_all <= (diams[ghosts] * ratio).astype(np.float64)
        idx_nn, _ = dfield.nn(Xg)
        projs = dfield.centers[:, idx_nn].astype(centers.dtype)
        if can.any():
            projs[:, can] = _project_3d(dfield, Xg[:, can], Rg[can]).astype(centers.dtype)
    diff = projs - Xg
    d = np.sqrt(_colsum(diff * diff))
    m = d <= diams[ghosts] * ratio
    return ghosts[m], projs[:, m]


def ghosts_and_projections_hcube(hfaces, hc_origin, hc_widths, centers, widths, ghost_layer_ratio=f32(1.5)):
    """ImmersedBoundary.jl:258-305; ``hfaces`` = [(dim 1-based, front)]."""
    ratio = f32(ghost_layer_ratio)
    nc = centers.shape[1]
    diams = np.sqrt(_colsum(widths * widths))
    mask = np.zeros(nc, dtype=bool)
    projs = np.empty_like(centers)
    dists = np.full(nc, np.inf, dtype=f32)
    for dim, front in hfaces:
        ps = centers.copy()
        ps[dim - 1, :] = (hc_origin[dim - 1] + hc_widths[dim - 1]) if front else hc_origin[dim - 1]
        diff = ps - centers
        ds = np.sqrt(_colsum(diff * diff))
        closer = ds < dists
        dists = np.where(closer, ds, dists)
        projs[:, closer] = ps[:, closer]
        mask |= ds < diams * ratio
    ghosts = np.nonzero(mask)[0].astype(np.int32)
    return ghosts, projs[:, ghosts]


def knn_sorted(tree, Xq, k):
    """k nearest neighbours of the rows of ``Xq (n, nd)`` ordered by (distance, index)."""
    d, idx = tree.query(np.ascontiguousarray(Xq, dtype=np.float64), k=k, workers=-1)
    if k == 1:
        d, idx = d[:, None], idx[:, None]
    order = np.lexsort((idx, d), axis=1)
    return np.take_along_axis(idx, order, axis=1)


def interpolator(X, Xc, tree, linear=True, k=0, bias=None):
    """Batched ``Interpolator(X, Xc, tree; first_index=true, linear, bias)`` (nninterp.jl:86-138): with ``bias`` the
    donors are the nearest neighbours of ``Xc + bias``, the weights those of ``Xc`` (:80-82, :121-126).

    ``X (n, nd)`` donors, ``Xc (m, nd)`` targets.  Weights: weighted least-squares linear
    (nninterp.jl:16-42) or inverse distance (:47-69); entries with |w| <= eps (sqrt(eps) for IDW)
    are dropped like the reference does.
    """
    nd = X.shape[1]
    if k == 0:
        k = 2 ** nd
    eps_ = np.finfo(f32).eps
    nb = knn_sorted(tree, Xc if bias is None else Xc + bias, k)  # (m, k)
    dX = X[nb] - Xc[:, None, :]   # (m, k, nd)
    s = dX[..., 0] * dX[..., 0]
    for q in range(1, nd):
        s = s + dX[..., q] * dX[..., q]
    dist = np.sqrt(s) + eps_
    w = f32(1.0) / dist
    if linear:
        A = np.concatenate([dX, np.ones(dX.shape[:2] + (1,), dtype=f32)], axis=2)
        Aw = A * w[..., None]
        rtol = eps_ * min(k, nd + 1)
        W = np.linalg.pinv(Aw, rcond=rtol)[:, -1, :] * w
        keep = np.abs(W) > eps_
    else:
        tot = w[:, 0].copy()
        for q in range(1, k):
            tot = tot + w[:, q]
        W = w / tot[:, None]
        keep = np.abs(W) > np.sqrt(eps_)
    cnt = keep.sum(axis=1)
    off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    return Accumulator(csr=(off, nb[keep], W[keep].astype(f32)), first_index=True, n_input=X.shape[0])


def _make_boundary(cT, wT, tree, ghost_indices, projs, ghost_ratio):
    """ImmersedBoundary.jl:422-448."""
    eps_ = np.finfo(f32).eps
    ghosts = cT[ghost_indices, :]
    normals = ghosts - projs
    gd = np.sqrt(_rowsum(normals * normals))
    normals = normals / (gd + eps_)[:, None]
    w = wT[ghost_indices, :]
    image_distances = np.sqrt(_rowsum(w * w)) * f32(ghost_ratio) + eps_
    images = projs + normals * image_distances[:, None]
    interp = interpolator(cT, images, tree, linear=True)
    dom = np.unique(interp.idx)
    remap = np.searchsorted(dom, interp.idx).astype(np.int32)
    interp = Accumulator(csr=(interp.off, remap, interp.w), first_index=True, n_input=dom.size)
    return Boundary(ghost_indices, projs, normals, image_distances, gd, interp, dom.astype(np.int32))


def boundary_partitions(cT, wT, tree, ghosts, projs, max_partition_size, ghost_ratio):
    """ImmersedBoundary.jl:456-476."""
    bd = {}
    n = len(ghosts)
    for ipart, s in enumerate(range(0, n, max_partition_size)):
        sl = slice(s, min(n, s + max_partition_size))
        bd[ipart + 1] = _make_boundary(cT, wT, tree, ghosts[sl], projs[sl, :], ghost_ratio)
    return bd


class Surface:
    """Post-processing surface of an immersed boundary (ImmersedBoundary.jl:328-343, built at :744-766): control
    points = simplex centres of the (refined) STL, ``offsets`` = 1.01 x diameter of the nearest cell, unit ``normals``,
    ``areas`` = norm of the area-weighted normals, ``interpolator`` (stencil searched at points + normals*offset,
    evaluated AT the points) and ``offset_interpolator`` (at points + normals*offset*ghost_layer_ratio)."""

    def __init__(self, points, offsets, normals, areas, interpolator, offset_interpolator, stl):
        self.points, self.offsets, self.normals, self.areas = points, offsets, normals, areas
        self.interpolator, self.offset_interpolator, self.stl = interpolator, offset_interpolator, stl


def make_surface(dfield, cT, tree, diams, ghost_layer_ratio):
    """ImmersedBoundary.jl:744-766."""
    from .mesher import centers_and_normals
    eps_ = np.finfo(f32).eps
    fcenters, fnormals = centers_and_normals(dfield.stl)            # (nd, ns) each
    _, idx = tree.query(np.ascontiguousarray(fcenters.T, dtype=np.float64))
    h = (diams[idx] * f32(1.01)).astype(f32)
    fnT = np.ascontiguousarray(fnormals.T)
    A = (np.sqrt(_rowsum(fnT * fnT)) + eps_).astype(f32)
    fnT = (fnT / A[:, None]).astype(f32)
    fcT = np.ascontiguousarray(fcenters.T).astype(f32)
    bias = (fnT * h[:, None]).astype(f32)
    return Surface(fcT, h, fnT, A, interpolator(cT, fcT, tree, bias=bias),
                   interpolator(cT, (fcT + bias * f32(ghost_layer_ratio)).astype(f32), tree), dfield.stl)


class Domain:
    """``Domain(msh; max_partition_size, partition_skirt_depth, ghost_layer_ratio, hypercube_families)``.

    ImmersedBoundary.jl:483-490 (struct) and :536-786 (constructor); ``boundaries=False`` skips the
    ghost-cell/boundary construction (not needed by the residual sweep itself).
    """

    def __init__(self, msh, max_partition_size=100_000, partition_skirt_depth=2, ghost_layer_ratio=f32(1.5),
                 hypercube_families=(), verbose=False, boundaries=True, only=None):
        """``only``: iterable of partition ids to build fully (one-partition-per-GPU runs build their own
        partition only); ``self.domains[id]`` (global cell ids of image + skirt) is kept for every partition."""
        nd = msh.ndims
        ncells = len(msh)
        centers, widths = get_cells(msh)
        fd, fo, fn = build_faces(msh, centers, widths)
        self.faces = (fd, fo, fn)
        self.partitions = {}
        self.domains = {}
        self.images = {}
        for ipart, start in enumerate(range(0, ncells, max_partition_size)):
            image = np.arange(start, min(ncells, start + max_partition_size), dtype=np.int32)
            self.images[ipart + 1] = (start, int(image[-1]) + 1)
            if only is not None and (ipart + 1) not in only:
                mask, _, _ = partition_domain_mask(image, ncells, fo, fn, partition_skirt_depth)
                self.domains[ipart + 1] = np.nonzero(mask[:ncells])[0].astype(np.int32)
                continue
            self.partitions[ipart + 1] = _build_partition(
                ipart + 1, image, ncells, fd, fo, fn, centers, widths, partition_skirt_depth, msh.block_size)
            self.domains[ipart + 1] = self.partitions[ipart + 1].domain
        self.boundaries = {}
        self.surfaces = {}
        if boundaries:
            cT = np.ascontiguousarray(centers.T)
            wT = np.ascontiguousarray(widths.T)
            tree = cKDTree(cT.astype(np.float64))
            for bname, hfaces in hypercube_families:
                g, p = ghosts_and_projections_hcube(hfaces, msh.origin, msh.widths, centers, widths, ghost_layer_ratio)
                self.boundaries[bname] = boundary_partitions(cT, wT, tree, g, np.ascontiguousarray(p.T),
                                                             max_partition_size, ghost_layer_ratio)
            for bname, dfield in msh.distance_fields.items():
                g, p = ghosts_and_projections(dfield, centers, widths, ghost_layer_ratio)
                self.boundaries[bname] = boundary_partitions(cT, wT, tree, g, np.ascontiguousarray(p.T),
                                                             max_partition_size, ghost_layer_ratio)
                self.surfaces[bname] = make_surface(dfield, cT, tree, np.sqrt(_colsum(widths * widths)),
                                                    ghost_layer_ratio)
        self.ncells = ncells
        self.mesh = msh
        self.reconstruction_kwargs = dict(
            max_partition_size=max_partition_size, partition_skirt_depth=partition_skirt_depth,
            ghost_layer_ratio=ghost_layer_ratio, hypercube_families=list(hypercube_families))
        self._centers = centers

    @property
    def ndims(self):
        return self.mesh.ndims

    def __len__(self):
        return self.ncells

    def global_centers(self):
        """``dom(X) do part, X; X .= part.centers end`` (used by multigrid, :1370-1373)."""
        return np.ascontiguousarray(self._centers.T)

    def __call__(self, f, *args, conv_to_backend=None, conv_from_backend=None, n_threads=0, **kwargs):
        """``(dom::Domain)(f, args...; conv_to_backend, conv_from_backend)`` (:820-864).

        The per-partition compute runs on the GPU only: both converters are required
        (use ``ibamd.hip`` / ``ibamd.to_host``); see backend.py for the resident form that
        avoids the per-call gather/upload altogether.
        """
        from . import backend
        return backend.domain_call(self, f, args, conv_to_backend, conv_from_backend, kwargs)


def multigrid(dom, max_levels=0, factor=2):
    """ImmersedBoundary.jl:1355-1407.  Returns ``(coarse_doms, prolongators, coarseners)`` -- the
    reference's actual return order (its docstring says otherwise)."""
    msh = dom.mesh
    mdepth = int(np.floor(np.log2(msh.block_size)))
    max_levels = mdepth if max_levels == 0 else max_levels
    coarse_doms, coarseners, prolongators = [], [], []
    Xold = dom.global_centers()
    tree_old = cKDTree(Xold.astype(np.float64))
    bsize = msh.block_size
    for _ in range(max_levels):
        bsize //= factor
        cmsh = Mesh(msh.origin, msh.widths, block_size=bsize, block_origins=msh.block_origins,
                    block_widths=msh.block_widths, distance_fields=msh.distance_fields)
        cdom = Domain(cmsh, **dom.reconstruction_kwargs)
        X = cdom.global_centers()
        tree = cKDTree(X.astype(np.float64))
        coarseners.append(interpolator(Xold, X, tree_old, linear=False))
        prolongators.append(interpolator(X, Xold, tree, linear=False))
        coarse_doms.append(cdom)
        tree_old, Xold = tree, X
    return coarse_doms, prolongators, coarseners
