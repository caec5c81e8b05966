"""TEST INFRASTRUCTURE -- an independent CPU restatement of the two mesher functions that fix the cell order of every
array on the hot path: ``refine_octree`` (/root/reference/src/mesher.jl:811-862) and ``get_cells`` (:1064-1112).
Recursive and literal, one Python statement per Julia statement; it shares no code with the product's mesher
(``immersedboundary.jl_amd/mesher.py``: explicit stack, vectorised cell enumeration), which tests compare it with bit
for bit (tests/test_mesher_restatement.py).  The distance functions of the refinement criteria are callables
``df(center) -> distance`` handed in by the caller.  Parity unpinned: the reference holds no golden mesh."""
import itertools

import numpy as np

f32 = np.float32


def _norm2(v):
    """``LinearAlgebra.norm`` of a short Float32 vector: scaled two-norm (generic_norm2: maxabs * sqrt(sum (x / maxabs)^2))."""
    m = f32(np.max(np.abs(v)))
    if m == 0:
        return f32(0)
    s = f32(0)
    for x in v:
        q = f32(x / m)
        s = f32(s + q * q)
    return f32(m * np.sqrt(s))


def _linrange_head(a, b, n):
    """``LinRange(a, b, n + 1)[1:end-1]`` for Float32 ends: element j is T((1 - j/n) a + (j/n) b) (Base.lerpi, the
    fraction in Float64)."""
    a64, b64 = float(a), float(b)
    return [f32((1.0 - j / n) * a64 + (j / n) * b64) for j in range(n)]


def refine_octree(refinement_criteria, origin, widths, growth_ratio=1.1):
    """mesher.jl:811-862.  Returns the list of (origin, widths) leaves in the order of the reference (depth first, children
    with the first axis fastest)."""
    origin = np.asarray(origin, dtype=f32)
    widths = np.asarray(widths, dtype=f32)
    L = f32(np.max(widths))                                           # :816
    R = f32(_norm2(widths) / f32(2))                                  # :817  circumradius
    center = (origin + widths / f32(2)).astype(f32)                   # :818
    active = []
    for (df, h) in refinement_criteria:                               # :820-830
        Lmax = max((float(growth_ratio) - 1.0) * float(f32(df(center)) - R), float(h))
        active.append(Lmax < float(L))
    if not any(active):                                               # :832-834
        return [(origin, widths)]
    refinement_criteria = [c for c, a in zip(refinement_criteria, active) if a]   # :836
    wmin = f32(np.min(widths))                                        # :838-842
    split_sizes = [int(np.rint(f32(w / wmin))) + 1 for w in widths]
    new_widths = np.array([f32(w / f32(s)) for w, s in zip(widths, split_sizes)], dtype=f32)     # :844
    axes = [_linrange_head(o, f32(o + w), s) for o, w, s in zip(origin, widths, split_sizes)]   # :845-850
    # Iterators.product |> collect |> vec: the FIRST range varies fastest
    new_origins = [tuple(reversed(t)) for t in itertools.product(*reversed(axes))]
    out = []
    for o in new_origins:                                             # :852-860  reduce(vcat, map(...))
        out += refine_octree(refinement_criteria, np.array(o, dtype=f32), new_widths, growth_ratio)
    return out


def get_cells(block_origins, block_widths, block_size):
    """mesher.jl:1064-1112 with margin = 0: centres and widths, shape (ndims, ncells); cell k of block b is column
    b * block_size^nd + k, k running with the first coordinate fastest."""
    nd = block_origins.shape[0]
    bs = int(block_size)
    rng = [f32(f32(0.5) + f32(i)) / f32(bs) for i in range(bs)]      # ((0.5f0):1.0f0:(bs - 0.5f0)) ./ bs
    inner = np.array([tuple(reversed(t)) for t in itertools.product(*([rng] * nd))], dtype=f32).T   # (nd, bs^nd)
    cols_c, cols_w = [], []
    for b in range(block_origins.shape[1]):                           # map over eachcol, reduce(hcat, ...)
        o, w = block_origins[:, b].astype(f32), block_widths[:, b].astype(f32)
        cols_c.append((inner * w[:, None]).astype(f32) + o[:, None])  # inner .* w .+ o
        cols_w.append(np.repeat((w / f32(bs)).astype(f32)[:, None], inner.shape[1], axis=1))
    return np.concatenate(cols_c, axis=1).astype(f32), np.concatenate(cols_w, axis=1).astype(f32)


class BlockTree:
    """The mesh container the oracle's ``Domain`` and ``multigrid`` need -- origin / widths of the box, the block tree
    (``block_origins``, ``block_widths``: (ndims, nblocks), depth-first order), ``block_size`` and the distance fields of the
    immersed surfaces (mesher.jl:926-970 minus everything that generates them).  ``multigrid`` (ImmersedBoundary.jl:1359-1382)
    coarsens a mesh by keeping the block tree and halving ``block_size``: ``coarser()``.  Independent of the product's ``Mesh``
    class (round-3 review: the oracle built its coarse meshes with it)."""

    def __init__(self, origin, widths, block_origins, block_widths, block_size, distance_fields=None):
        self.origin = np.asarray(origin, dtype=f32)
        self.widths = np.asarray(widths, dtype=f32)
        self.block_origins = np.asarray(block_origins, dtype=f32)
        self.block_widths = np.asarray(block_widths, dtype=f32)
        self.block_size = int(block_size)
        self.distance_fields = dict(distance_fields or {})

    @classmethod
    def of(cls, msh):
        """the same tree as any object with these six attributes (e.g. the product's mesh: the INPUT of the path)"""
        return cls(msh.origin, msh.widths, msh.block_origins, msh.block_widths, msh.block_size, msh.distance_fields)

    def coarser(self, factor=2):
        return BlockTree(self.origin, self.widths, self.block_origins, self.block_widths, self.block_size // factor,
                         self.distance_fields)

    @property
    def ndims(self):
        return self.block_origins.shape[0]

    @property
    def nblocks(self):
        return self.block_origins.shape[1]

    def __len__(self):
        return self.block_size ** self.ndims * self.nblocks
