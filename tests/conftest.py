import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import ibamd  # noqa: E402  (registers the package directory `immersedboundary.jl_amd` as `ibamd`)

f32 = np.float32
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def advection_mesh(h=1e-2):
    """Mesh of /root/reference/test/advection.jl:4-20 (h = 1e-2) or dissipation.jl (h = 2e-2)."""
    from ibamd.mesher import Line, Mesh, Stereolitography
    lower = Stereolitography(np.array([[0.0, 1.0], [0.0, 0.0]]))
    upper = Stereolitography(np.array([[0.0, 0.0], [0.0, 1.0]]))
    return Mesh([0.0, 0.0], [1.0, 1.0], ("lower", lower, f32(h)), ("upper", upper, f32(h)),
                refinement_regions=[(Line([0.0, 0.0], [1.0, 1.0]), f32(2 * h)),
                                    (Line([0.0, 0.0], [0.5, 0.5]), f32(h))])


def rae_mesh(h_wall=1e-2, h_feat=5e-3):
    """Mesh of /root/reference/test/rae2822.jl:4-14 (shipped settings by default)."""
    from ibamd.mesher import DistanceField, Mesh, Stereolitography, feature_regions, merge_points
    stl = merge_points(Stereolitography(os.path.join(GOLDEN, "rae2822.dat")))
    features = DistanceField(feature_regions(stl, radius=0.05))
    return Mesh(f32([-25.0, -25.0]), f32([50.0, 50.0]), ("wall", stl, f32(h_wall)),
                refinement_regions=[(features, f32(h_feat))])


@pytest.fixture(scope="session")
def adv_mesh():
    return advection_mesh()


@pytest.fixture(scope="session")
def adv_mesh_coarse():
    return advection_mesh(2e-2)


@pytest.fixture(scope="session")
def rae_mesh_small():
    return rae_mesh()


ADV_FAMILIES = [("outlet", [(1, True), (2, True)])]
RAE_FAMILIES = [("farfield", [(1, False), (1, True), (2, False), (2, True)])]


@pytest.fixture(scope="session")
def adv_domains(adv_mesh):
    """(product Domain, oracle Domain) of the advection case split in 3 partitions."""
    from oracle import domain as od
    kw = dict(hypercube_families=ADV_FAMILIES, max_partition_size=4096)
    return ibamd.Domain(adv_mesh, **kw), od.Domain(adv_mesh, **kw)


@pytest.fixture(scope="session")
def rae_domains(rae_mesh_small):
    from oracle import domain as od
    kw = dict(hypercube_families=RAE_FAMILIES, max_partition_size=16384)
    return ibamd.Domain(rae_mesh_small, **kw), od.Domain(rae_mesh_small, **kw)


def seeded_field(centers, seed=12345, nv=None, kind="smooth"):
    """Synthetic fields of SURVEY.md 8d: sin*cos + 0.1*noise, or a step across the diagonal."""
    rng = np.random.default_rng(seed)
    x, y = centers[:, 0], centers[:, 1]
    n = centers.shape[0]
    cols = 1 if nv is None else nv
    out = np.empty((n, cols), dtype=f32)
    for v in range(cols):
        if kind == "smooth":
            base = np.sin(2 * np.pi * x * (1 + 0.3 * v)) * np.cos(2 * np.pi * y)
            out[:, v] = (base + 0.1 * rng.uniform(-1, 1, n)).astype(f32)
        else:
            out[:, v] = ((y > x + 0.05 * v).astype(f32) + f32(0.01) * rng.uniform(-1, 1, n)).astype(f32)
    return out[:, 0].copy() if nv is None else out


def euler_field(centers, seed=12345):
    """P = [p T u v] of SURVEY.md 8d."""
    rng = np.random.default_rng(seed)
    n = centers.shape[0]
    nd = centers.shape[1]
    P = np.empty((n, nd + 2), dtype=f32)
    P[:, 0] = 1e5 * (1 + 0.05 * rng.uniform(-1, 1, n))
    P[:, 1] = 288.15 * (1 + 0.05 * rng.uniform(-1, 1, n))
    for d in range(nd):
        P[:, 2 + d] = 100.0 * (1 + 0.1 * rng.uniform(-1, 1, n))
    return P


def rel_inf(a, b):
    """Norm-wise relative error per variable: max|a-b| / max|b| (SURVEY.md 8d tolerance)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.ndim == 1:
        a, b = a[:, None], b[:, None]
    den = np.abs(b).max(axis=0)
    den = np.where(den == 0, 1.0, den)
    return (np.abs(a - b).max(axis=0) / den).max()


def oracle_view(part):
    """Oracle-side view of a product partition (same arrays, oracle Accumulator objects)."""
    from oracle.accumulator import Accumulator as OAcc

    class P:
        pass
    op = P()
    op.ndims, op.spacing, op.centers = part.ndims, part.spacing, part.centers
    op.face_owners_neighbors = part.face_owners_neighbors
    op.face_accumulators = {}
    for k, acc in part.face_accumulators.items():
        o = object.__new__(OAcc)
        o.n_output, o.first_index, o.stencils = acc.n_output, True, acc.stencils
        op.face_accumulators[k] = o
    return op


def oracle_boundaries_view(dom):
    """Oracle-side view of the boundaries of a product Domain (same arrays, oracle Accumulator objects): lets the
    oracle's ``impose_bc`` run on domains too large for the literal ``Domain(msh)`` restatement."""
    from oracle.accumulator import Accumulator as OAcc

    class V:
        pass
    view = V()
    view.boundaries = {}
    for name, parts in dom.boundaries.items():
        view.boundaries[name] = {}
        for ipart, b in parts.items():
            ob = V()
            for k in ("ghost_indices", "projections", "normals", "image_distances", "ghost_distances", "image_domain"):
                setattr(ob, k, getattr(b, k))
            acc = b.image_interpolator
            o = object.__new__(OAcc)
            o.n_output, o.first_index, o.stencils = acc.n_output, True, acc.stencils
            ob.image_interpolator = o
            view.boundaries[name][ipart] = ob
    return view


def stencil_scale(part, u, ref):
    """Local scale of a residual value for per-cell error bounds: |ref| + (max |u| over the cell AND its face neighbours) / h.
    The residual is a difference of neighbour values divided by h; |u| of the cell alone vanishes where u crosses zero
    (scripts/diag_bounds.py: the eight worst cells of the 3-D sweep under the cell-only scale all have |u| ~ 1e-3 between
    neighbours of ~1e-1), so the cell-only scale reads rounding errors of ~3e-7 of the stencil as 2e-5."""
    import numpy as _np
    a = _np.abs(u).astype(_np.float64)
    m = a.copy()
    for d in range(1, part.ndims + 1):
        o, nb = part.face_owners_neighbors[d][0], part.face_owners_neighbors[d][1]
        _np.maximum.at(m, o, a[nb])
        _np.maximum.at(m, nb, a[o])
    return _np.abs(ref).astype(_np.float64) + m / part.spacing.min(axis=1).astype(_np.float64)
