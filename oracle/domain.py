"""ORACLE (test infrastructure only; never imported by the product path).

Literal CPU restatement (numpy, Float32, array-at-a-time in the same
evaluation order as the Julia broadcasts) of the reference's partition
runtime, grid operators, ghost-cell BC and ``Domain`` construction:
/root/reference/src/ImmersedBoundary.jl:63-326 (faces, ghosts),
:383-476 (Partition/Boundary), :536-786 (Domain ctor), :820-864 (Domain call),
:873-1157 (operators), :1197-1247 (impose_bc!), :1355-1431 (multigrid,
volume_integral).

PARITY PINNING: the reference ships no asserted tests or golden vectors for
this path and Julia is not installed, so this restatement is pinned by the
reference's own known answers only (Accumulator docstring example, rae2822
centroid, analytic invariants; tests/test_oracle_known_answers.py).  kNN
tie-breaks / pinv cut-offs / thread-dependent global face order are UNPINNED
 and canonicalised: faces are registered in single-thread order
with each cell's candidates visited in ascending index.

Conventions: cell/face indices 0-based (reference 1-based), "no cell" = -1
(reference 0); ``dim`` arguments stay 1-based (``dim=0`` = all dims in
``JST_sensor``) so signatures read like the reference's.
"""
import numpy as np
from scipy.spatial import cKDTree

from .accumulator import Accumulator
from . import nninterp

f32 = np.float32


# ---------------------------------------------------------------------------
# structs
# ---------------------------------------------------------------------------
class Partition:
    """ImmersedBoundary.jl:383-392."""

    def __init__(self, id, centers, spacing, face_accumulators, face_owners_neighbors,
                 domain, image, image_in_domain):
        self.id = id
        self.centers = centers
        self.spacing = spacing
        self.face_accumulators = face_accumulators
        self.face_owners_neighbors = face_owners_neighbors
        self.domain = domain
        self.image = image
        self.image_in_domain = image_in_domain

    @property
    def ndims(self):
        return self.centers.shape[1]


class Boundary:
    """ImmersedBoundary.jl:406-448."""

    def __init__(self, centers, widths, tree, ghost_indices, projs, ghost_ratio=f32(1.5)):
        eps_ = np.finfo(f32).eps
        ghosts = centers[ghost_indices, :]
        normals = ghosts - projs
        gd = np.sqrt(_rowsum(normals * normals))
        normals = normals / (gd + eps_)[:, None]
        w = widths[ghost_indices, :]
        image_distances = np.sqrt(_rowsum(w * w)) * f32(ghost_ratio) + eps_
        images = projs + normals * image_distances[:, None]
        interp = nninterp.Interpolator(centers, images, tree, first_index=True, linear=True)
        dom, hmap = nninterp.domain(interp)
        nninterp.re_index(interp, hmap)
        self.ghost_indices = ghost_indices
        self.projections = projs
        self.normals = normals
        self.image_distances = image_distances
        self.ghost_distances = gd
        self.image_interpolator = interp
        self.image_domain = dom.astype(np.int32)


class Surface:
    """ImmersedBoundary.jl:328-376 (struct, surface_integral, call, at_offset), built at :744-766."""

    def __init__(self, points, offsets, normals, areas, interpolator, offset_interpolator, stl):
        self.points, self.offsets, self.normals, self.areas = points, offsets, normals, areas
        self.interpolator, self.offset_interpolator, self.stl = interpolator, offset_interpolator, stl

    @staticmethod
    def build(dfield, cT, tree, diams, ghost_layer_ratio):
        from ibamd.mesher import centers_and_normals
        eps_ = np.finfo(f32).eps
        fcenters, fnormals = centers_and_normals(dfield.stl)
        _, idx = tree.query(np.ascontiguousarray(fcenters.T, dtype=np.float64))
        h = diams[idx] * f32(1.01)
        fnT = np.ascontiguousarray(fnormals.T)
        A = np.sqrt(_rowsum(fnT * fnT)) + eps_
        fnT = fnT / A[:, None]
        fcT = np.ascontiguousarray(fcenters.T)
        bias = fnT * h[:, None]
        return Surface(fcT, h, fnT, A,
                       nninterp.Interpolator(cT, fcT, tree, bias=bias, first_index=True),
                       nninterp.Interpolator(cT, fcT + bias * f32(ghost_layer_ratio), tree, first_index=True), dfield.stl)

    def __call__(self, u):
        """:364"""
        return self.interpolator(u)


def at_offset(surf, u):
    """:372"""
    return surf.offset_interpolator(u)


def surface_integral(surf, u):
    """:345-357"""
    if u.ndim == 1:
        return (surf.areas * u).sum(dtype=f32)
    return (surf.areas[:, None] * u).sum(axis=0, dtype=f32)


def _rowsum(a):
    s = a[:, 0].copy()
    for k in range(1, a.shape[1]):
        s = s + a[:, k]
    return s


def _colsum(a):
    s = a[0].copy()
    for k in range(1, a.shape[0]):
        s = s + a[k]
    return s


def boundary_partitions(centers, widths, tree, ghost_indices, projs, max_partition_size=100_000,
                        ghost_ratio=f32(1.5)):
    """ImmersedBoundary.jl:456-476."""
    bd = {}
    n = len(ghost_indices)
    for ipart, s in enumerate(range(0, n, max_partition_size)):
        sl = slice(s, min(n, s + max_partition_size))
        bd[ipart + 1] = Boundary(centers, widths, tree, ghost_indices[sl], projs[sl, :], ghost_ratio)
    return bd


class Domain:
    """ImmersedBoundary.jl:483-490 + ctor :536-786."""

    def __init__(self, msh, max_partition_size=100_000, partition_skirt_depth=2,
                 ghost_layer_ratio=f32(1.5), hypercube_families=(), verbose=False):
        from .mesher import get_cells   # the independent restatement (tests/test_mesher_restatement.py: bit-identical cells)

        nd = msh.block_origins.shape[0]
        ncells = len(msh)
        centers, widths = get_cells(msh.block_origins, msh.block_widths, msh.block_size)
        origins = centers - widths / f32(2)

        faces = octree2faces(origins, widths) + hcube_faces(msh.origin, msh.widths, origins, widths)
        self.faces = faces
        cells2faces = [[] for _ in range(ncells)]
        for ifc, (_, o, n) in enumerate(faces):
            if o >= 0:
                cells2faces[o].append(ifc)
            if n >= 0:
                cells2faces[n].append(ifc)

        partitions = {}
        for ipart, start in enumerate(range(0, ncells, max_partition_size)):
            image = np.arange(start, min(ncells, start + max_partition_size), dtype=np.int32)
            domain = set(int(c) for c in image)
            for _ in range(partition_skirt_depth):
                for c in list(domain):
                    for f in cells2faces[c]:
                        _, o, n = faces[f]
                        if o >= 0:
                            domain.add(o)
                        if n >= 0:
                            domain.add(n)
            domain = np.array(sorted(domain), dtype=np.int32)
            idx2domain = {int(d): k for k, d in enumerate(domain)}
            seen = set()
            face_indices = []
            for c in domain:
                for f in cells2faces[c]:
                    if f not in seen:
                        seen.add(f)
                        face_indices.append(f)
            face_accumulators = {}
            face_owners_neighbors = {}
            for dim in range(1, nd + 1):
                owners, neighbors = [], []
                right_faces = [[] for _ in range(len(domain))]
                left_faces = [[] for _ in range(len(domain))]
                k = -1
                for f in face_indices:
                    ndim, o, n = faces[f]
                    if ndim != dim:
                        continue
                    o = idx2domain.get(o, -1)
                    n = idx2domain.get(n, -1)
                    add_left = add_right = True
                    if o < 0:
                        o = n
                        add_right = False
                    if n < 0:
                        n = o
                        add_left = False
                    owners.append(o)
                    neighbors.append(n)
                    k += 1
                    if add_left:
                        left_faces[n].append(k)
                    if add_right:
                        right_faces[o].append(k)
                face_owners_neighbors[dim] = (np.array(owners, dtype=np.int32),
                                              np.array(neighbors, dtype=np.int32))
                face_accumulators[(dim, False)] = Accumulator(
                    left_faces, _averaging_weights(left_faces), first_index=True)
                face_accumulators[(dim, True)] = Accumulator(
                    right_faces, _averaging_weights(right_faces), first_index=True)
            image_in_domain = np.array([idx2domain[int(i)] for i in image], dtype=np.int32)
            partitions[ipart + 1] = Partition(
                ipart + 1,
                np.ascontiguousarray(centers[:, domain].T),
                np.ascontiguousarray(widths[:, domain].T),
                face_accumulators, face_owners_neighbors, domain, image, image_in_domain)

        boundaries = {}
        surfaces = {}
        cT = np.ascontiguousarray(centers.T)
        wT = np.ascontiguousarray(widths.T)
        tree = cKDTree(cT.astype(np.float64))
        for bname, hfaces in hypercube_families:
            ghosts, projs = ghosts_and_projections_hcube(
                hfaces, msh.origin, msh.widths, centers, widths, ghost_layer_ratio)
            boundaries[bname] = boundary_partitions(
                cT, wT, tree, ghosts, np.ascontiguousarray(projs.T), max_partition_size, ghost_layer_ratio)
        for bname, dfield in msh.distance_fields.items():
            ghosts, projs = ghosts_and_projections(dfield, centers, widths, ghost_layer_ratio)
            boundaries[bname] = boundary_partitions(
                cT, wT, tree, ghosts, np.ascontiguousarray(projs.T), max_partition_size, ghost_layer_ratio)
            surfaces[bname] = Surface.build(dfield, cT, tree, np.sqrt(_colsum(widths * widths)), ghost_layer_ratio)

        self.ncells = ncells
        self.mesh = msh
        self.partitions = partitions
        self.boundaries = boundaries
        self.surfaces = surfaces
        self.reconstruction_kwargs = dict(
            max_partition_size=max_partition_size, partition_skirt_depth=partition_skirt_depth,
            ghost_layer_ratio=ghost_layer_ratio, hypercube_families=list(hypercube_families))

    @property
    def ndims(self):
        return self.partitions[1].ndims

    def __len__(self):
        return self.ncells

    def __call__(self, f, *args, **kwargs):
        """ImmersedBoundary.jl:820-864: gather domain rows, call f, scatter image rows."""
        results = []
        for i in self.partitions:
            part = self.partitions[i]
            dargs = [np.array(a[part.domain]) for a in args]
            r = f(part, *dargs, **kwargs)
            for a, da in zip(args, dargs):
                a[part.image] = da[part.image_in_domain]
            results.append(r)
        return results


def _averaging_weights(stencils):
    """ImmersedBoundary.jl:501-506."""
    return [np.full(len(s), f32(1.0) / f32(len(s)) if len(s) else f32(0), dtype=f32) for s in stencils]


# ---------------------------------------------------------------------------
# faces (ImmersedBoundary.jl:63-184)
# ---------------------------------------------------------------------------
def octree2faces(origins, widths):
    """ImmersedBoundary.jl:63-132; returns list of (dim 1-based, owner, neighbor) 0-based."""
    nd, nc = origins.shape
    centers = origins + widths / f32(2)
    tree = cKDTree(np.ascontiguousarray(centers.T, dtype=np.float64))
    s = widths[0] * widths[0]
    for k in range(1, nd):
        s = s + widths[k] * widths[k]
    radii = np.sqrt(s) / f32(2)
    cand = tree.query_ball_point(np.ascontiguousarray(centers.T, dtype=np.float64),
                                 r=(radii * f32(3.1)).astype(np.float64))
    faces = []
    maxs_all = origins + widths
    for i in range(nc):
        js = np.array(sorted(cand[i]), dtype=np.int64)
        js = js[js != i]
        if js.size == 0:
            continue
        fo = np.maximum(origins[:, i:i + 1], origins[:, js])
        fw = np.minimum(maxs_all[:, i:i + 1], maxs_all[:, js]) - fo
        tol = f32(0.01) * fw.max(axis=0)
        n = (fw < tol).sum(axis=0)
        nz = (fw < -tol).sum(axis=0)
        ok = (n == 1) & (nz == 0)
        ndim = np.argmin(fw, axis=0)
        for t in np.nonzero(ok)[0]:
            j = int(js[t])
            d = int(ndim[t])
            if origins[d, j] < origins[d, i]:
                continue
            faces.append((d + 1, i, j))
    return faces


def hcube_faces(hc_origin, hc_widths, origins, widths):
    """ImmersedBoundary.jl:150-184."""
    faces = []
    for dim in range(len(hc_origin)):
        idxs = np.nonzero(np.abs(origins[dim] - hc_origin[dim]) < widths[dim] * f32(0.01))[0]
        for i in idxs:
            faces.append((dim + 1, -1, int(i)))
        idxs = np.nonzero(
            np.abs(origins[dim] + widths[dim] - hc_origin[dim] - hc_widths[dim]) < widths[dim] * f32(0.01))[0]
        for i in idxs:
            faces.append((dim + 1, int(i), -1))
    return faces


# ---------------------------------------------------------------------------
# ghosts (ImmersedBoundary.jl:194-326)
# ---------------------------------------------------------------------------
def ghosts_and_projections(dfield, centers, widths, ghost_layer_ratio=f32(1.5)):
    """ImmersedBoundary.jl:194-230 (distance-field surfaces)."""
    ratio = f32(ghost_layer_ratio)
    diams = np.sqrt(_colsum(widths * widths))
    _, dists = dfield.nn(centers)
    ghosts = np.nonzero(dists <= diams * ratio * f32(2))[0].astype(np.int32)
    projs = np.empty((centers.shape[0], ghosts.size), dtype=centers.dtype)
    for k, g in enumerate(ghosts):
        projs[:, k] = dfield.projection(centers[:, g], diams[g] * ratio * f32(2))
    diff = projs - centers[:, ghosts]
    d = np.sqrt(_colsum(diff * diff))
    mask = d <= diams[ghosts] * ratio
    return ghosts[mask], projs[:, mask]


def ghosts_and_projections_hcube(hfaces, hc_origin, hc_widths, centers, widths, ghost_layer_ratio=f32(1.5)):
    """ImmersedBoundary.jl:258-305 (hypercube families); ``hfaces`` = [(dim 1-based, front bool)]."""
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


# ---------------------------------------------------------------------------
# grid operators (ImmersedBoundary.jl:873-1157).  u: (nc,) or (nc, nv)
# ---------------------------------------------------------------------------
def _b(v, like):
    """Broadcast a per-row vector against (n,) or (n, nv...) arrays."""
    return v.reshape(v.shape + (1,) * (like.ndim - 1))


def at_owners(part, u, dim):
    """:879-881"""
    return u[part.face_owners_neighbors[dim][0]]


def at_neighbors(part, u, dim):
    """:889-891"""
    return u[part.face_owners_neighbors[dim][1]]


def at_faces(part, u, dim):
    """:899-910"""
    spown = at_owners(part, part.spacing, dim)[:, dim - 1]
    spneigh = at_neighbors(part, part.spacing, dim)[:, dim - 1]
    uown = at_owners(part, u, dim)
    uneigh = at_neighbors(part, u, dim)
    return (uown * _b(spneigh, uown) + uneigh * _b(spown, uown)) / _b(spneigh + spown, uown)


def green_gauss(part, uf, dim):
    """:918-926"""
    accl = part.face_accumulators[(dim, False)]
    accr = part.face_accumulators[(dim, True)]
    r = accr(uf) - accl(uf)
    return r / _b(part.spacing[:, dim - 1], r)


def unsigned_green_gauss(part, uf, dim):
    """:934-942"""
    accl = part.face_accumulators[(dim, False)]
    accr = part.face_accumulators[(dim, True)]
    r = accr(uf) + accl(uf)
    return r / _b(part.spacing[:, dim - 1], r)


def divergent(part, uf):
    """:950-956"""
    s = green_gauss(part, uf[0], 1)
    for d in range(2, part.ndims + 1):
        s = s + green_gauss(part, uf[d - 1], d)
    return s


def cell_gradient(part, u, dim=None):
    """:965-987"""
    if dim is None:
        return tuple(cell_gradient(part, u, d) for d in range(1, part.ndims + 1))
    return green_gauss(part, at_faces(part, u, dim), dim)


def face_distance(part, dim):
    """:995-1002"""
    spown = at_owners(part, part.spacing, dim)[:, dim - 1]
    spneigh = at_neighbors(part, part.spacing, dim)[:, dim - 1]
    return (spown + spneigh) / f32(2)


def owner_distance(part, dim):
    """:1010-1016"""
    return at_owners(part, part.spacing, dim)[:, dim - 1] / f32(2)


def neighbor_distance(part, dim):
    """:1024-1030"""
    return at_neighbors(part, part.spacing, dim)[:, dim - 1] / f32(2)


def face_gradient(part, u, a, b=None):
    """:1039-1043 ``face_gradient(part,u,dim)`` and :1051-1069 ``face_gradient(part,u,grad_u,dim)``."""
    if b is None:
        dim = a
        d = at_neighbors(part, u, dim) - at_owners(part, u, dim)
        return d / _b(face_distance(part, dim), d)
    gu, dim = a, b
    out = []
    for i in range(1, part.ndims + 1):
        out.append(face_gradient(part, u, dim) if i == dim else at_faces(part, gu[i - 1], dim))
    return tuple(out)


def JST_sensor(part, p, dim=0):
    """:1077-1097"""
    if dim == 0:
        nu = np.full_like(p, f32(1e-7))
        for d in range(1, part.ndims + 1):
            nu = np.maximum(nu, JST_sensor(part, p, d))
        return nu
    face_diff = at_neighbors(part, p, dim) - at_owners(part, p, dim)
    return (f32(1e-7) + np.abs(green_gauss(part, face_diff, dim))) / (
        f32(1e-7) + unsigned_green_gauss(part, np.abs(face_diff), dim))


def minmod(u1, u2):
    """:1099"""
    return np.minimum(np.abs(u1), np.abs(u2)) * (np.sign(u1) + np.sign(u2)) / f32(2)


def MUSCL(part, u, du, dim, D=None, high_order=False):
    """:1113-1157"""
    down = owner_distance(part, dim)
    dneigh = neighbor_distance(part, dim)
    uown = at_owners(part, u, dim)
    uneigh = at_neighbors(part, u, dim)
    down_b, dneigh_b = _b(down, uown), _b(dneigh, uown)
    guf = (uneigh - uown) / (down_b + dneigh_b)
    duo = at_owners(part, du, dim)
    dun = at_neighbors(part, du, dim)
    gu = (f32(2) * duo - guf) * down_b
    Du = (f32(2) * dun - guf) * dneigh_b
    guf = minmod(Du, gu)
    uL, uR = uown + guf, uneigh - guf
    if D is not None:
        Df = np.maximum(np.maximum(at_owners(part, D, dim), at_neighbors(part, D, dim)), f32(1e-7))
        Df = _b(Df, uown)
        uf = (uown * dneigh_b + uneigh * down_b) / (down_b + dneigh_b)
        if high_order:
            uf = uf + (duo * down_b - dun * dneigh_b) / f32(8)
        uL = uL * Df + (f32(1.0) - Df) * uf
        uR = uR * Df + (f32(1.0) - Df) * uf
    return uL, uR


# ---------------------------------------------------------------------------
# ghost-cell BC (ImmersedBoundary.jl:1197-1247)
# ---------------------------------------------------------------------------
def impose_bc(f, dom, bname, *args, **kwargs):
    parts = dom.boundaries[bname]
    for ipart in parts:
        bdry = parts[ipart]
        ginds = bdry.ghost_indices
        eta = bdry.ghost_distances / bdry.image_distances
        iargs = [bdry.image_interpolator(a[bdry.image_domain]) for a in args]
        r = f(bdry, *iargs, **kwargs)
        if not isinstance(r, tuple):
            r = (r,)
        for a, ba, ia in zip(args, r, iargs):
            e = _b(eta, ia)
            a[ginds] = e * ia + (f32(1.0) - e) * ba


# ---------------------------------------------------------------------------
# multigrid / integrals (ImmersedBoundary.jl:1355-1431)
# ---------------------------------------------------------------------------
def multigrid(dom, max_levels=0, factor=2):
    """Returns ``(coarse_doms, prolongators, coarseners)`` -- the reference's actual order (:1406)."""
    from .mesher import BlockTree   # (the oracle's own mesh container: nothing of the product's mesher is used here)

    msh = dom.mesh
    mdepth = int(np.floor(np.log2(msh.block_size)))
    max_levels = mdepth if max_levels == 0 else max_levels
    coarse_doms, coarseners, prolongators = [], [], []

    def global_centers(d):
        X = np.zeros((len(d), d.ndims), dtype=f32)

        def fill(part, X):
            X[...] = part.centers
        d(fill, X)
        return X

    Xold = global_centers(dom)
    tree_old = cKDTree(Xold.astype(np.float64))
    bsize = msh.block_size
    for _ in range(max_levels):
        bsize //= factor
        cmsh = BlockTree(msh.origin, msh.widths, msh.block_origins, msh.block_widths, bsize, msh.distance_fields)
        cdom = Domain(cmsh, **dom.reconstruction_kwargs)
        X = global_centers(cdom)
        tree = cKDTree(X.astype(np.float64))
        coarsener = nninterp.Interpolator(Xold, X, tree_old, first_index=True, linear=False)
        prolongator = nninterp.Interpolator(X, Xold, tree, first_index=True, linear=False)
        coarse_doms.append(cdom)
        prolongators.append(prolongator)
        coarseners.append(coarsener)
        tree_old, Xold = tree, X
    return coarse_doms, prolongators, coarseners


def volume_integral(dom, A):
    """:1415-1431"""
    Ai = A.copy()

    def mul(part, Ai):
        for dim in range(part.ndims):
            Ai *= _b(part.spacing[:, dim], Ai)
    dom(mul, Ai)
    return Ai.sum(axis=0, dtype=Ai.dtype)
