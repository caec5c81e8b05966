"""Block quadtree/octree mesher (cold path, host side).

Host-side mirror of the reference's ``BlockMesher`` module
(/root/reference/src/mesher.jl).  It exists to *produce inputs* for the
residual hot path: the Julia original cannot run in
the build container, so the workloads of bench.py and the tests are generated
here with the reference's semantics (same names, same argument meaning).

Arrays follow the reference convention ``(ndims, npoints)`` (column = point).
Float widths follow the reference: meshes are Float32 (mesher.jl:988-989),
distance functors keep the dtype of the data they were built from.
"""
from __future__ import annotations

import math
import sys

import numpy as np
from scipy.spatial import cKDTree

f32 = np.float32
f64 = np.float64


def _norm(v):
    """Euclidean norm in the dtype of ``v`` (sequential sum, like generic_norm2)."""
    v = np.asarray(v)
    s = v.dtype.type(0)
    for x in v:
        s = s + x * x
    return np.sqrt(s)


# ---------------------------------------------------------------------------
# distance functors (mesher.jl:27-122)
# ---------------------------------------------------------------------------
class Box:
    """Refinement box; call gives distance to it (mesher.jl:27-46)."""

    def __init__(self, origin, widths):
        self.origin = np.asarray(origin)
        self.widths = np.asarray(widths)

    def __call__(self, pt):
        pt = np.asarray(pt)
        d = pt - self.origin
        outside = (d > self.widths) | (pt < self.origin)
        v = np.minimum(np.abs(d), np.abs(d - self.widths)) * outside
        return _norm(v)


class Ball:
    """Ball; distance = max(0, |c - pt| - r) (mesher.jl:58-76)."""

    def __init__(self, center, radius):
        self.center = np.asarray(center)
        self.radius = radius

    def __call__(self, pt):
        return max(0.0, float(_norm(self.center - np.asarray(pt)) - self.radius))


class Line:
    """Segment p1-p2; distance to it (mesher.jl:94-122)."""

    def __init__(self, p1, p2):
        self.p1 = np.asarray(p1)
        self.p2 = np.asarray(p2)
        self.m = self.p2 - self.p1

    def __call__(self, pt):
        pt = np.asarray(pt)
        v = pt - self.p1
        # m \ v for vectors = least-squares scalar (mesher.jl:112)
        xi = np.dot(self.m, v) / np.dot(self.m, self.m)
        if xi < 0.0:
            return _norm(pt - self.p1)
        elif xi > 1.0:
            return _norm(pt - self.p2)
        return _norm(pt - (self.p1 + self.m * xi))


# ---------------------------------------------------------------------------
# STL I/O (mesher.jl:124-228)
# ---------------------------------------------------------------------------
def read_stl(filename):
    """Points ``(3, n)`` Float32 and simplices ``(3, ntri)`` (1-based), un-merged."""
    with open(filename, "rb") as f:
        head = f.read(5)
    if head == b"solid":
        verts, faces, face = [], [], []
        with open(filename, "r") as f:
            for raw in f:
                line = raw.strip()
                if line.startswith("vertex"):
                    c = line.split()
                    verts.append([f32(c[1]), f32(c[2]), f32(c[3])])
                    face.append(len(verts))
                elif line.startswith("facet normal"):
                    face = []
                elif line.startswith("endloop"):
                    faces.append(face)
        return np.array(verts, dtype=f32).T.copy(), np.array(faces, dtype=np.int64).T.copy()
    with open(filename, "rb") as f:
        raw = f.read()
    ntri = int(np.frombuffer(raw, dtype="<u4", count=1, offset=80)[0])
    rec = np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")])
    tris = np.frombuffer(raw, dtype=rec, count=ntri, offset=84)
    points = tris["v"].reshape(-1, 3).T.astype(f32).copy()
    simplices = np.arange(1, 3 * ntri + 1, dtype=np.int64).reshape(ntri, 3).T.copy()
    return points, simplices


def write_stl_binary(filename, points, simplices):
    """Write a binary STL (used to synthesise the 3-D sphere workload)."""
    pts = np.asarray(points, dtype=f32)
    simp = np.asarray(simplices, dtype=np.int64) - 1
    ntri = simp.shape[1]
    rec = np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")])
    out = np.zeros(ntri, dtype=rec)
    out["v"] = pts[:, simp].transpose(2, 1, 0)
    with open(filename, "wb") as f:
        f.write(b"\0" * 80)
        f.write(np.uint32(ntri).tobytes())
        f.write(out.tobytes())


class Stereolitography:
    """Surface: ``points (nd, np)``, ``simplices (nd, ns)`` 1-based (mesher.jl:238-296)."""

    def __init__(self, points, simplices=None, closed=True):
        if isinstance(points, str):
            fname = points
            if fname[-4:] in (".dat", ".DAT"):
                pts = np.loadtxt(fname, dtype=np.float64).T.astype(f32)
                self._from_polyline(pts, True)
                return
            self.points, self.simplices = read_stl(fname)
            return
        points = np.asarray(points)
        if simplices is None:
            self._from_polyline(points, closed)
        else:
            self.points = points
            self.simplices = np.asarray(simplices, dtype=np.int64)

    def _from_polyline(self, points, closed):
        n = points.shape[1]
        inds = np.arange(1, n + 1, dtype=np.int64)
        if closed:
            simp = np.stack([inds, np.roll(inds, -1)])
        else:
            simp = np.stack([inds[:-1], inds[1:]])
        self.points = points
        self.simplices = simp


def merge_points(*stls, tolerance=1e-7, clean_degenerate=True):
    """Merge coincident points by rounded tag (mesher.jl:351-407)."""
    tag2ind = {}
    new_points = []
    new_simplices = []
    for stl in stls:
        pts = stl.points
        q = pts / tolerance  # dtype promotion as in Julia (f32/f32 or f32/f64)
        tags = np.rint(q).astype(np.int64)
        idx = np.empty(pts.shape[1], dtype=np.int64)
        for k in range(pts.shape[1]):
            t = tuple(tags[:, k])
            i = tag2ind.get(t)
            if i is None:
                i = len(new_points) + 1
                tag2ind[t] = i
                new_points.append(pts[:, k])
            idx[k] = i
        new_simplices.append(idx[stl.simplices - 1])
    P = np.stack(new_points, axis=1)
    S = np.concatenate(new_simplices, axis=1)
    if clean_degenerate:
        srt = np.sort(S, axis=0)
        ok = np.all(srt[1:] != srt[:-1], axis=0)
        S = S[:, ok]
    return Stereolitography(P, S)


def cat(*stls):
    """Concatenate surfaces (mesher.jl:415-431)."""
    pts = np.concatenate([s.points for s in stls], axis=1)
    simp, n = [], 0
    for s in stls:
        simp.append(s.simplices + n)
        n += s.points.shape[1]
    return Stereolitography(pts, np.concatenate(simp, axis=1))


def _refine_simplex(simplex, h, growth_ratio, refinement_regions):
    """Iterative form of the recursive ``refine_to_length!`` (mesher.jl:438-495).

    Returns the refined simplices in the reference's depth-first order.
    """
    out = []
    stack = [simplex]
    gr1 = growth_ratio - 1.0
    while stack:
        s = stack.pop()
        nv = s.shape[1]
        max_violation = 0.0
        index = -1
        for i in range(nv):
            inext = 0 if i == nv - 1 else i + 1
            p1 = s[:, i]
            p2 = s[:, inext]
            phalf = (p1 + p2) / 2
            L = _norm(p2 - p1)
            hloc = h
            for df, href in refinement_regions:
                hloc = min(hloc, max((df(phalf) - L) * gr1, href))
            violation = L - hloc
            if max_violation < violation:
                max_violation = violation
                index = i
        if index < 0:
            out.append(s)
            continue
        inext = 0 if index == nv - 1 else index + 1
        pnew = (s[:, index] + s[:, inext]) / 2
        new_s = s.copy()
        s = s.copy()
        s[:, inext] = pnew
        new_s[:, index] = pnew
        # depth-first: first child fully before the second
        stack.append(new_s)
        stack.append(s)
    return out


def refine_to_length(stl, h, tolerance=1e-7, growth_ratio=1.1, refinement_regions=()):
    """Split simplices until every edge is <= local length (mesher.jl:503-528)."""
    pieces = []
    for k in range(stl.simplices.shape[1]):
        simp = stl.points[:, stl.simplices[:, k] - 1].copy()
        pieces.extend(_refine_simplex(simp, h, growth_ratio, list(refinement_regions)))
    nd = pieces[0].shape[1]
    points = np.concatenate(pieces, axis=1)
    simplices = np.arange(1, points.shape[1] + 1, dtype=np.int64).reshape(-1, nd).T
    return merge_points(Stereolitography(points, simplices), tolerance=tolerance)


def proj2simplex(simplex, pt):
    """Projection of ``pt`` on a simplex (mesher.jl:544-596)."""
    eps_ = f32(1e-14)
    ns = simplex.shape[1]
    if ns == 1:
        return simplex[:, 0].copy()
    if ns == 2:
        p0 = simplex[:, 0]
        p1 = simplex[:, 1]
        u = p1 - p0
        a = (pt - p0) * u
        b = u * u
        num, den = a[0], b[0]
        for q in range(1, a.shape[0]):
            num = num + a[q]
            den = den + b[q]
        xi = num / (den + eps_)
        if xi < -eps_:
            return p0.copy()
        elif xi > 1.0 + eps_:
            return p1.copy()
        return p0 + u * xi
    p0 = simplex[:, 0]
    M = simplex[:, 1:] - p0[:, None]
    xi = np.linalg.pinv(M) @ (pt - p0)
    if np.any(xi < -eps_) or (np.sum(xi) > 1.0 + eps_):
        best, d = None, np.inf
        for i in range(ns):
            face = np.delete(simplex, i, axis=1)
            _p = proj2simplex(face, pt)
            _d = _norm(_p - pt)
            if _d < d:
                d, best = _d, _p
        return best
    return p0 + M @ xi


def _simplex_normal(simplex, normalize=True):
    eps_ = f32(1e-14)
    if simplex.shape[0] == 2:
        v = simplex[:, 1] - simplex[:, 0]
        n = np.array([v[1], -v[0]], dtype=simplex.dtype)
        return n / (_norm(v) + eps_) if normalize else n
    p0 = simplex[:, 0]
    n = np.cross(simplex[:, 1] - p0, simplex[:, 2] - p0)
    return n / (_norm(n) + eps_) if normalize else n


def centers_and_normals(stl):
    """Simplex centres and area-weighted normals (mesher.jl:639-660)."""
    P = stl.points[:, stl.simplices - 1]  # (nd, nvert, ns)
    nvert = P.shape[1]
    c = P[:, 0, :].copy()
    for k in range(1, nvert):
        c = c + P[:, k, :]
    centers = c / nvert
    if stl.points.shape[0] == 2:
        v = P[:, 1, :] - P[:, 0, :]
        normals = np.stack([v[1], -v[0]])
    else:
        a = P[:, 1, :] - P[:, 0, :]
        b = P[:, 2, :] - P[:, 0, :]
        normals = np.cross(a.T, b.T).T
    return centers, normals


def feature_regions(stl, angle=15.0, radius=np.inf, include_boundaries=False):
    """Simplices violating radius / angle criteria (mesher.jl:670-728)."""
    eps_ = np.finfo(f32).eps
    angle = math.radians(max(angle, 1.0))
    max_cos = math.cos(math.radians(0.05))
    edges = []
    registry = {}
    ns = stl.simplices.shape[1]
    for i in range(ns):
        simp = stl.simplices[:, i]
        for pivot in simp:
            face = tuple(sorted(int(x) for x in simp if x != pivot))
            j = registry.pop(face, None)
            if j is not None:
                edges.append((j, i))
            else:
                registry[face] = i
    for ind in registry.values():
        edges.append((ind, ind))
    centers, normals = centers_and_normals(stl)
    included = np.zeros(ns, dtype=bool)
    for i, j in edges:
        ni = normals[:, i]
        nj = normals[:, j]
        ni = ni / (_norm(ni) + eps_)
        nj = nj / (_norm(nj) + eps_)
        theta = math.acos(min(float(np.dot(ni, nj)), max_cos))
        d = float(_norm(centers[:, i] - centers[:, j]))
        if (i == j and include_boundaries) or (d / theta < radius) or (theta > angle):
            included[i] = True
            included[j] = True
    return Stereolitography(stl.points, stl.simplices[:, included])


class DistanceField:
    """Approximate distance field = distance to nearest simplex centre (mesher.jl:736-801)."""

    def __init__(self, stl, leaf_size=25, h=0.0):
        if h > 0.0:
            stl = refine_to_length(stl, h)
        self.stl = stl
        self.centers, _ = centers_and_normals(stl)
        self.tree = cKDTree(np.ascontiguousarray(self.centers.T, dtype=np.float64), leafsize=leaf_size)

    def _dist(self, x, idx):
        d = x - self.centers[:, idx]
        return _norm(d)

    def __call__(self, x):
        x = np.asarray(x)
        _, idx = self.tree.query(x.astype(np.float64))
        return self._dist(x, idx)

    def nn(self, X):
        """Vectorised nearest-centre query: ``X (nd, n)`` -> (idx 0-based, dist)."""
        _, idx = self.tree.query(np.ascontiguousarray(X.T, dtype=np.float64), workers=-1)
        diff = X - self.centers[:, idx]
        s = diff[0] * diff[0]
        for k in range(1, diff.shape[0]):
            s = s + diff[k] * diff[k]
        return idx, np.sqrt(s)

    def projection(self, x, R=0.0):
        """Projection on the surface, candidates within ``R`` (mesher.jl:778-801)."""
        x = np.asarray(x)
        _, idx = self.tree.query(x.astype(np.float64))
        d = self._dist(x, idx)
        p = self.centers[:, idx].copy()
        if R > d:
            cand = self.tree.query_ball_point(x.astype(np.float64), float(R))
            for i in sorted(cand):
                simp = self.stl.points[:, self.stl.simplices[:, i] - 1]
                _p = proj2simplex(simp, x)
                _d = _norm(_p - x)
                if _d < d:
                    d = _d
                    p = np.asarray(_p, dtype=p.dtype).copy()
        return p


# ---------------------------------------------------------------------------
# octree (mesher.jl:811-862)
# ---------------------------------------------------------------------------
def refine_octree(refinement_criteria, origin, widths, growth_ratio=1.1):
    """Depth-first block refinement; children x-fastest (mesher.jl:811-862).

    ``origin``/``widths`` are Float32 vectors.  A criterion ``(df, h)`` is
    active for a cell iff ``max((growth_ratio-1)*(df(center)-R), h) < L``;
    inactive criteria are dropped for the whole subtree.
    Returns a list of ``(origin, widths)``.
    """
    origin = np.asarray(origin, dtype=f32)
    widths = np.asarray(widths, dtype=f32)
    gr1 = float(growth_ratio) - 1.0
    nd = origin.shape[0]
    out = []
    stack = [(list(refinement_criteria), origin, widths)]
    while stack:
        crit, o, w = stack.pop()
        L = np.max(w)
        R = _norm(w) / f32(2)
        center = o + w / f32(2)
        active = []
        for df, h in crit:
            Lmax = max(gr1 * float(df(center) - R), float(h))
            if Lmax < float(L):
                active.append((df, h))
        if not active:
            out.append((o, w))
            continue
        wmin = np.min(w)
        split = np.rint(w / wmin).astype(np.int64) + 1
        new_w = (w / split.astype(f32)).astype(f32)
        axes = []
        for d in range(nd):
            s = int(split[d])
            a, b = float(o[d]), float(f32(o[d] + w[d]))
            # LinRange lerp in Float64, rounded to Float32 (Base lerpi)
            axes.append([f32((1.0 - j / s) * a + (j / s) * b) for j in range(s)])
        children = []
        # Iterators.product: first axis fastest
        idx = [0] * nd
        while True:
            children.append(np.array([axes[d][idx[d]] for d in range(nd)], dtype=f32))
            d = 0
            while d < nd:
                idx[d] += 1
                if idx[d] < len(axes[d]):
                    break
                idx[d] = 0
                d += 1
            if d == nd:
                break
        for c in reversed(children):
            stack.append((active, c, new_w))
    return out


def refine_orderly(*surfaces, refinement_regions=(), ratio=f32(0.5), growth_ratio=f32(2.0),
                   tolerance=f32(1e-7)):
    """Refine STLs in ascending-h order into DistanceFields (mesher.jl:878-918)."""
    hs = [s[1] for s in surfaces]
    order = sorted(range(len(surfaces)), key=lambda i: hs[i])
    regions = [(t[0], t[1] * ratio) for t in refinement_regions]
    result = {}
    for i in order:
        stl, h = surfaces[i]
        h = h * ratio
        stl = refine_to_length(stl, h, tolerance=tolerance, refinement_regions=regions,
                               growth_ratio=growth_ratio)
        dfield = DistanceField(stl)
        result[i] = dfield
        regions.append((dfield, h))
    return [result[i] for i in range(len(surfaces))]


class Mesh:
    """Block mesh; matrices are ``(ndims, nblocks)`` Float32 (mesher.jl:926-1046).

    ``Mesh(origin, widths, (name, stl, h), ...; growth_ratio, tolerance,
    block_size, refinement_regions)`` generates; passing ``block_origins``
    builds the struct directly (the reference's positional constructor, used by
    ``multigrid`` to re-block the same tree).
    """

    def __init__(self, origin, widths, *surfaces, growth_ratio=f32(2.0), tolerance=f32(1e-7),
                 block_size=8, refinement_regions=(), verbose=False,
                 block_origins=None, block_widths=None, distance_fields=None):
        self.origin = np.asarray(origin, dtype=f32)
        self.widths = np.asarray(widths, dtype=f32)
        self.block_size = int(block_size)
        if block_origins is not None:
            self.block_origins = np.asarray(block_origins, dtype=f32)
            self.block_widths = np.asarray(block_widths, dtype=f32)
            self.distance_fields = dict(distance_fields or {})
            return
        hs = {name: h for (name, _, h) in surfaces}
        dfl = refine_orderly(*[(stl, h) for (_, stl, h) in surfaces],
                             refinement_regions=refinement_regions,
                             growth_ratio=growth_ratio, tolerance=tolerance)
        dfields = {t[0]: df for t, df in zip(surfaces, dfl)}
        bs = np.int32(self.block_size)
        ref_regions = [(t[0], t[1] * bs) for t in refinement_regions]
        for name in dfields:
            ref_regions.append((dfields[name], hs[name] * bs))
        if verbose:
            print("Refining region tree...", file=sys.stderr)
        self.ref_regions, self.growth_ratio = ref_regions, growth_ratio   # (kept: what refine_octree was called with)
        ows = refine_octree(ref_regions, self.origin, self.widths, growth_ratio)
        self.block_origins = np.stack([t[0] for t in ows], axis=1)
        self.block_widths = np.stack([t[1] for t in ows], axis=1)
        self.distance_fields = dfields

    @property
    def ndims(self):
        return self.block_origins.shape[0]

    @property
    def nblocks(self):
        return self.block_origins.shape[1]

    def __len__(self):
        return self.block_size ** self.ndims * self.nblocks


def get_cells(msh, rng=None):
    """Cell centres and widths ``(nd, ncells)`` (mesher.jl:1064-1112, margin=0).

    Global cell id = block*bs^nd + local, local index x-fastest.
    """
    bo = msh.block_origins if rng is None else msh.block_origins[:, rng]
    bw = msh.block_widths if rng is None else msh.block_widths[:, rng]
    nd = bo.shape[0]
    bs = msh.block_size
    npb = bs ** nd
    r = (np.arange(bs, dtype=f32) + f32(0.5)) / f32(bs)
    inner = np.empty((nd, npb), dtype=f32)
    loc = np.arange(npb)
    for d in range(nd):
        inner[d] = r[(loc // bs ** d) % bs]
    centers = (inner[:, None, :] * bw[:, :, None] + bo[:, :, None]).astype(f32)
    centers = centers.reshape(nd, -1)
    widths = np.repeat((bw / f32(bs)).astype(f32), npb, axis=1)
    return centers, widths
