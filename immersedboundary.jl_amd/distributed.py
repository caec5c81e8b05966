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
