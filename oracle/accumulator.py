"""ORACLE (test infrastructure only; never imported by the product path).

CPU restatement of ``ArrayAccumulator.Accumulator``
(/root/reference/src/accumulator.jl:12-130): variable-length weighted stencil
reduce, bucketed by stencil length.

Indices are 0-based here (the reference is 1-based); everything else follows
the reference: ``stencils[len] = (rows, idx[len x n], w[len x n] | None)``,
zero-length bucket gives 0, N-d arrays are reduced slice by slice along the
first (``first_index``) or last dimension.
"""
import numpy as np


class Accumulator:
    def __init__(self, inds, weights=None, first_index=False):
        """accumulator.jl:39-65.  ``inds``: list of index lists; ``weights``: same shape or None."""
        ls = np.array([len(s) for s in inds], dtype=np.int64)
        self.n_output = len(inds)
        self.first_index = first_index
        self.stencils = {}
        # `unique(ls)`: first-appearance order
        seen = []
        for l in ls:
            if int(l) not in seen:
                seen.append(int(l))
        for l in seen:
            rows = np.nonzero(ls == l)[0]
            if l == 0:
                idx = np.zeros((0, rows.size), dtype=np.int64)
                ws = None if weights is None else np.zeros((0, rows.size), dtype=np.float32)
            else:
                idx = np.stack([np.asarray(inds[r], dtype=np.int64) for r in rows], axis=1)
                ws = None
                if weights is not None:
                    ws = np.stack([np.asarray(weights[r]) for r in rows], axis=1)
            self.stencils[l] = (rows, idx, ws)

    def _vec(self, v, delta=False, f=None, op=None):
        """accumulator.jl:78-111 (vector method)."""
        vnew = np.zeros(self.n_output, dtype=v.dtype)
        for rows, stencil, weights in self.stencils.values():
            if stencil.shape[0] == 0:
                continue
            g = v[stencil]  # (len, n)
            if weights is None:
                t = g if f is None else f(g)
            else:
                if delta:
                    g = g - v[rows][None, :]
                if f is not None:
                    g = f(g)
                t = g * weights
            # reduce(op, ...; dims=1): sequential along the stencil axis
            acc = t[0]
            for k in range(1, t.shape[0]):
                acc = (acc + t[k]) if op is None else op(acc, t[k])
            vnew[rows] = acc
        return vnew

    def __call__(self, v, delta=False, f=None, op=None):
        """accumulator.jl:126-130: mapslices over the non-summation dims."""
        v = np.asarray(v)
        if v.ndim == 1:
            return self._vec(v, delta, f, op)
        if self.first_index:
            flat = v.reshape(v.shape[0], -1)
            cols = [self._vec(np.ascontiguousarray(flat[:, j]), delta, f, op) for j in range(flat.shape[1])]
            return np.stack(cols, axis=1).reshape((self.n_output,) + v.shape[1:])
        flat = v.reshape(-1, v.shape[-1])
        rows_ = [self._vec(np.ascontiguousarray(flat[j]), delta, f, op) for j in range(flat.shape[0])]
        return np.stack(rows_, axis=0).reshape(v.shape[:-1] + (self.n_output,))

    def decompose(self):
        """accumulator.jl:137-165: back to list-of-lists (indices, weights)."""
        indices = [None] * self.n_output
        weights = [None] * self.n_output
        has_w = False
        for rows, st, ws in self.stencils.values():
            for k, i in enumerate(rows):
                indices[i] = st[:, k].copy()
                if ws is not None:
                    has_w = True
                    weights[i] = ws[:, k].copy()
        return (indices, weights) if has_w else (indices, None)
