"""One-partition-per-GPU pieces beyond the skirt exchange: what a rank needs to run
``impose_bc!`` and ``FAS!`` on device-resident local arrays.

The reference works on GLOBAL arrays: ``impose_bc!`` interpolates the image points of a boundary chunk from
``a[bdry.image_domain]`` -- donor cells anywhere in the domain -- and writes ``a[bdry.ghost_indices]``
(/root/reference/src/ImmersedBoundary.jl:1228-1245); ``FAS!`` takes ``norm(r)`` over all cells (src/solver.jl:57,84)
and test/advection.jl:59 a global minimum of the time step.  With one partition per rank:

* a rank updates the ghosts its ``image`` owns (skirt copies of ghosts arrive with the next skirt exchange);
* donor cells outside ``part.domain`` are appended to the local arrays as extra rows ``nc .. nc + n_extra`` and
  added to the halo lists (``HaloPlan(..., extra=...)``): one exchange refreshes skirt and donor cells together;
* the boundary chunks are re-indexed to those local rows (``local_boundaries``) -- same interpolation weights, same
  blending, so the result on the owned ghosts is the reference's bit for bit;
* norms and minima are all-reduced over the ranks' image cells (``Reductions``).
Everything here is index bookkeeping; the arithmetic stays in libibhip (or, in the CPU tests, in the oracle).
"""
from __future__ import annotations

import numpy as np

from .accumulator import Accumulator
from .domain import Boundary


def _owned_rows(b, lo, hi):
    gi = np.asarray(b.ghost_indices)
    return np.nonzero((gi >= lo) & (gi < hi))[0]


def _donors_of(b, rows):
    """Global ids of the donor cells of the image points of ghosts ``rows`` of boundary chunk ``b``."""
    acc = b.image_interpolator
    off, idx = acc.off, acc.idx
    if rows.size == 0:
        return np.zeros(0, dtype=np.int64)
    take = np.concatenate([np.arange(off[r], off[r + 1]) for r in rows]) if rows.size else np.zeros(0, np.int64)
    return np.asarray(b.image_domain, dtype=np.int64)[idx[take]]


def bc_donor_extras(dom):
    """For every partition: sorted global ids of the ``impose_bc!`` donor cells of the ghosts it owns that lie outside
    its ``domain`` (image + skirt).  Every rank computes the table of every partition: it needs the others' to know what
    to send."""
    out = {}
    for q, (lo, hi) in dom.images.items():
        dq = np.asarray(dom.domains[q], dtype=np.int64)
        need = [np.zeros(0, dtype=np.int64)]
        for parts in dom.boundaries.values():
            for b in parts.values():
                need.append(_donors_of(b, _owned_rows(b, lo, hi)))
        need = np.unique(np.concatenate(need))
        out[q] = need[~np.isin(need, dq)]
    return out


def local_boundaries(dom, pid, extras=None):
    """Boundary chunks of ``dom`` restricted to the ghosts partition ``pid`` owns, re-indexed to its local rows
    (``part.domain`` order, donor extras appended).  Returns ``({name: {chunk: Boundary}}, n_rows)``."""
    extras = bc_donor_extras(dom) if extras is None else extras
    lo, hi = dom.images[pid]
    domain = np.asarray(dom.domains[pid], dtype=np.int64)
    ext = extras[pid]
    nc = domain.size

    def to_local(g):
        pos = np.searchsorted(domain, g)
        pos = np.minimum(pos, nc - 1)
        in_dom = domain[pos] == g
        epos = np.searchsorted(ext, g)
        epos = np.minimum(epos, max(ext.size - 1, 0))
        if ext.size:
            assert np.all(in_dom | (ext[epos] == g)), "donor cell neither in the domain nor in the extras"
        else:
            assert np.all(in_dom)
        return np.where(in_dom, pos, nc + epos).astype(np.int64)
    out = {}
    for name, parts in dom.boundaries.items():
        out[name] = {}
        for ichunk, b in parts.items():
            rows = _owned_rows(b, lo, hi)
            if rows.size == 0:
                continue
            acc = b.image_interpolator
            lens = (acc.off[rows + 1] - acc.off[rows]).astype(np.int64)
            take = np.concatenate([np.arange(acc.off[r], acc.off[r + 1]) for r in rows])
            donors_local = to_local(np.asarray(b.image_domain, dtype=np.int64)[acc.idx[take]])
            image_domain = np.unique(donors_local)
            new_idx = np.searchsorted(image_domain, donors_local)
            off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
            interp = Accumulator(csr=(off, new_idx.astype(np.int32), acc.w[take]), first_index=True,
                                 n_input=image_domain.size)
            out[name][ichunk] = Boundary(to_local(np.asarray(b.ghost_indices, dtype=np.int64)[rows]).astype(np.int32),
                                         b.projections[rows], b.normals[rows], b.image_distances[rows],
                                         b.ghost_distances[rows], interp, image_domain.astype(np.int32))
    return out, nc + ext.size


class LocalDomain:
    """Stand-in for ``dom`` in ``impose_bc(f, dom, bname, args...)`` on a rank: ``boundaries`` hold local indices, the
    arrays passed are the rank's extended local arrays."""

    def __init__(self, dom, pid, extras=None):
        self.extras = bc_donor_extras(dom) if extras is None else extras
        self.boundaries, self.n_rows = local_boundaries(dom, pid, self.extras)
        self.pid = pid
        self.nc = int(np.asarray(dom.domains[pid]).size)
        self.ndims = dom.ndims

    def __len__(self):
        return self.n_rows


class Reductions:
    """``norm(r)`` (solver.jl:57,84) and ``minimum`` (test/advection.jl:59) over the cells the ranks own: local partial
    on the device (or with numpy in the CPU tests), one all-reduce of a scalar."""

    def __init__(self, image_in_domain, group=None, device="cpu"):
        import torch
        import torch.distributed as dist
        self.dist, self.group, self.torch = dist, group, torch
        self.device = torch.device(device)
        self.rows = torch.from_numpy(np.ascontiguousarray(image_in_domain, dtype=np.int64)).to(self.device)
        self._on_cpu = (not dist.is_initialized()) or dist.get_backend(group) != "nccl"

    def _allreduce(self, t, op):
        if self.dist.is_initialized() and self.dist.get_world_size(self.group) > 1:
            if self._on_cpu and t.is_cuda:
                c = t.cpu()
                self.dist.all_reduce(c, op=op, group=self.group)
                return c
            self.dist.all_reduce(t, op=op, group=self.group)
        return t

    def norm(self, r):
        """||r||_2 over the owned cells of all ranks; ``r``: local array (device tensor, HipArray or numpy)."""
        torch = self.torch
        if isinstance(r, np.ndarray):
            part = torch.tensor([float((r[self.rows.numpy()].astype(np.float64) ** 2).sum())], dtype=torch.float64)
        else:
            from . import backend as B
            t = getattr(r, "t", r) if not isinstance(r, torch.Tensor) else r
            own = t.index_select(0, self.rows)
            flat = own if own.ndim == 1 else own.T.contiguous().T
            part = torch.zeros(1, dtype=torch.float64, device=t.device)
            B._stream()
            B.call("ibh_sumsq", int(flat.numel()), B._ptr(flat), B._ptr(part))
        return float(self._allreduce(part, self.dist.ReduceOp.SUM).item()) ** 0.5

    def minimum(self, x):
        """Global minimum of a per-rank scalar (the time step of test/advection.jl:52-59)."""
        t = self.torch.tensor([float(x)], dtype=self.torch.float64)
        return float(self._allreduce(t, self.dist.ReduceOp.MIN).item())


class RankOps(Reductions):
    """What the point-implicit smoother (point_implicit.py; /root/reference/src/point_implicit.jl:65-114, 221-236, 250-329)
    needs of a rank beyond ``Reductions``:

    * ``closure(f_local)``: the residual closure of the rank -- refreshes the skirt (and donor) rows of its argument with
      ``exchange`` (every Hutchinson sample and every Jacobian-vector product is a residual sweep, and needs them), calls
      ``f_local`` and zeroes the rows the rank does not own, so that every vector the smoother forms vanishes there;
    * ``sum(t)`` / ``max(t)``: all-reduce of a small device (or numpy) array in place -- the two dot products of
      ``proj_along``, the norm and ``max |r|`` of ``solve``;
    * ``own(a)``: the rows of a local array that this rank owns, zero elsewhere (for ``b`` and initial vectors).

    ``n_rows``: rows of the local arrays (domain + extras); ``exchange(X)``: in-place halo exchange of a local array."""

    def __init__(self, image_in_domain, n_rows, exchange, group=None, device="cpu"):
        super().__init__(image_in_domain, group=group, device=device)
        self.n_rows = int(n_rows)
        self.exchange = exchange
        m = np.zeros(self.n_rows, dtype=np.float32)
        m[np.asarray(image_in_domain, dtype=np.int64)] = 1.0
        self.mask_np = m
        self.mask = self.torch.from_numpy(m).to(self.device)
        self.not_own_np = m == 0
        self.not_own = self.torch.from_numpy(self.not_own_np).to(self.device)

    def own(self, a):
        """``a`` with the rows this rank does not own set to zero -- assigned, not multiplied: an image-only sweep leaves
        whatever was in memory (possibly NaN) on the skirt rows of its output."""
        if isinstance(a, np.ndarray):
            a = a.copy()
            a[self.not_own_np] = 0
            return a
        a[self.not_own] = 0
        return a

    def closure(self, f_local):
        def f(X):
            self.exchange(X)
            return self.own(f_local(X))
        return f

    def _reduce_inplace(self, t, op):
        if isinstance(t, np.ndarray):
            c = self.torch.from_numpy(t)
            if self.dist.is_initialized() and self.dist.get_world_size(self.group) > 1:
                self.dist.all_reduce(c, op=op, group=self.group)
            return t
        r = self._allreduce(t, op)
        if r is not t:
            t.copy_(r)
        return t

    def sum(self, t):
        return self._reduce_inplace(t, self.dist.ReduceOp.SUM)

    def max(self, t):
        return self._reduce_inplace(t, self.dist.ReduceOp.MAX)


# ---------------------------------------------------------------------------
# FAS! across ranks: the levels of multigrid(dom) with one partition of every level per rank
# ---------------------------------------------------------------------------
def _local_rows(gids, domain, extra):
    """Local rows of global cells ``gids``: position in ``domain`` (sorted), else ``len(domain)`` + position in ``extra``."""
    gids = np.asarray(gids, dtype=np.int64)
    nc = domain.size
    pos = np.minimum(np.searchsorted(domain, gids), max(nc - 1, 0))
    in_dom = domain[pos] == gids if nc else np.zeros(gids.shape, bool)
    if extra is None or extra.size == 0:
        assert np.all(in_dom), "transfer donor cell outside the local domain and no extras given"
        return pos.astype(np.int64)
    epos = np.minimum(np.searchsorted(extra, gids), extra.size - 1)
    assert np.all(in_dom | (extra[epos] == gids)), "transfer donor cell neither in the domain nor in the extras"
    return np.where(in_dom, pos, nc + epos).astype(np.int64)


def _rows_accumulator(acc_rows, n_out_local, out_rows_local, col_local, n_in_local):
    """Local Accumulator with ``n_out_local`` rows: row ``out_rows_local[i]`` holds stencil ``i`` of ``acc_rows`` (columns
    re-indexed by ``col_local``); every other row is empty (its output is zero)."""
    lens = np.zeros(n_out_local, dtype=np.int64)
    lens[out_rows_local] = np.diff(acc_rows.off)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    # out_rows_local is increasing (image cells in domain order): the stencils are already in CSR order
    assert np.all(np.diff(out_rows_local) > 0)
    return Accumulator(csr=(off, col_local.astype(np.int32), acc_rows.w), first_index=True, n_input=n_in_local)


class RankLevels:
    """What a rank holds of ``multigrid(dom)`` (ImmersedBoundary.jl:1355-1407) when every level is cut the same way
    (partition ``pid`` of every level = the same blocks of the shared block tree, ``max_partition_size`` divided by
    ``factor^nd`` per level): its partition of every level with a halo plan, and the transfer operators restricted to
    the rows it owns, re-indexed to local rows --

    * ``coarseners[l]``: level l -> l + 1, rows = the rank's image cells of level l + 1; the IDW donors of a coarse cell are
      its own children, local by construction;
    * ``prolongators[l]``: level l + 1 -> l, rows = the rank's image cells of level l; IDW donors are coarse cells around
      the fine cell, in the coarse partition's skirt or -- rarely -- beyond it: those are appended to the coarse level's
      local arrays as extra rows and to its halo lists (as ``bc_donor_extras`` does for ``impose_bc!``), gathered from
      all ranks at set-up.

    Same stencils and weights as the global operators (the interpolator is called on the same points with the same
    trees), so a V-cycle over these reproduces the one-partition V-cycle on the owned cells.

    With boundaries (``domain_kwargs`` without ``boundaries=False``): ``local_doms[l]`` is the ``LocalDomain`` of level l --
    its boundary chunks restricted to the ghosts the rank owns, re-indexed to local rows -- for ``impose_bc`` inside the
    level closure; the BC donor cells beyond the skirt are part of the level's extra rows (one exchange refreshes all)."""

    def __init__(self, msh, pid, world, max_levels, factor=2, group=None, domain_kwargs=None):
        from scipy.spatial import cKDTree
        from .domain import Domain, interpolator
        from .halo import HaloPlan
        from .mesher import Mesh
        nd = msh.ndims
        kw = dict(domain_kwargs or {})
        npb = msh.block_size ** nd
        ncells = len(msh)
        mps = -(-(-(-ncells // world)) // npb) * npb
        self.pid, self.world = pid, world
        self.doms, self.parts, self.plans, self.extras, self.nrows = [], [], [], [], []
        self.coarseners, self.prolongators = [], []
        meshes = [msh]
        bs = msh.block_size
        for _ in range(max_levels):
            bs //= factor
            meshes.append(Mesh(msh.origin, msh.widths, block_size=bs, block_origins=msh.block_origins,
                               block_widths=msh.block_widths, distance_fields=msh.distance_fields))
        for l, m in enumerate(meshes):
            dom = Domain(m, max_partition_size=mps // (factor ** nd) ** l, only=[pid], **kw)
            assert len(dom.images) == world, "every level must be cut into one partition per rank"
            self.doms.append(dom)
            self.parts.append(dom.partitions[pid])
        X = [d.global_centers() for d in self.doms]
        trees = [cKDTree(x.astype(np.float64)) for x in X]
        need_extra = []
        pro_rows = []
        for l in range(max_levels):
            fine, coarse = self.parts[l], self.parts[l + 1]
            flo, fhi = self.doms[l].images[pid]
            clo, chi = self.doms[l + 1].images[pid]
            # the rank's rows of the global operators: same points, same trees, same weights
            co = interpolator(X[l], X[l + 1][clo:chi], trees[l], linear=False)        # coarse image rows <- fine cells
            pr = interpolator(X[l + 1], X[l][flo:fhi], trees[l + 1], linear=False)    # fine image rows <- coarse cells
            fdom = np.asarray(fine.domain, dtype=np.int64)
            cdom = np.asarray(coarse.domain, dtype=np.int64)
            self.coarseners.append((co, _local_rows(co.idx, fdom, None)))
            donors = np.asarray(pr.idx, dtype=np.int64)
            miss = np.unique(donors[~np.isin(donors, cdom)])
            need_extra.append(miss)
            pro_rows.append(pr)
        # extras of every partition of every coarse level: gathered (a rank must know what the others need of it)
        import torch.distributed as dist
        tables = []
        for l in range(max_levels + 1):
            mine = need_extra[l - 1] if l >= 1 else np.zeros(0, dtype=np.int64)
            if dist.is_initialized() and world > 1:
                every = [None] * world
                dist.all_gather_object(every, mine, group=group)
            else:
                every = [mine]
            tables.append({q + 1: np.asarray(every[q], dtype=np.int64) for q in range(world)})
        # impose_bc! on every level's own boundaries (multigrid builds each coarse Domain with them, :1381-1382): the donor
        # cells of the ghosts a rank owns that lie beyond its skirt join the level's extra rows and halo lists
        self.local_doms = [None] * (max_levels + 1)
        with_bc = bool(kw.get("boundaries", True)) and any(len(v) for d in self.doms for v in d.boundaries.values())
        if with_bc:
            for l in range(max_levels + 1):
                bx = bc_donor_extras(self.doms[l])
                tables[l] = {q: np.union1d(tables[l][q], bx.get(q, np.zeros(0, dtype=np.int64))).astype(np.int64)
                             for q in tables[l]}
        for l in range(max_levels + 1):
            self.extras.append(tables[l])
            self.plans.append(HaloPlan(self.doms[l], pid, extra=tables[l]))
            self.nrows.append(int(self.parts[l].domain.size + tables[l][pid].size))
            if with_bc:
                self.local_doms[l] = LocalDomain(self.doms[l], pid, tables[l])
                assert self.local_doms[l].n_rows == self.nrows[l]
        # local operators
        co_local, pr_local = [], []
        for l in range(max_levels):
            co, cols = self.coarseners[l]
            co_local.append(_rows_accumulator(co, self.nrows[l + 1], np.asarray(self.parts[l + 1].image_in_domain),
                                              cols, self.nrows[l]))
            pr = pro_rows[l]
            cdom = np.asarray(self.parts[l + 1].domain, dtype=np.int64)
            cols = _local_rows(pr.idx, cdom, tables[l + 1][pid])
            pr_local.append(_rows_accumulator(pr, self.nrows[l], np.asarray(self.parts[l].image_in_domain), cols,
                                              self.nrows[l + 1]))
        self.coarseners, self.prolongators = co_local, pr_local

    @property
    def n_levels(self):
        return len(self.parts)
