"""Host-side ``Accumulator`` container (mirror of /root/reference/src/accumulator.jl:12-65).

Holds the variable-length weighted stencils in CSR form (what libibhip
consumes) and offers the reference's bucketed-by-length view on demand.
It does NOT evaluate anything on the CPU: application happens on the GPU
through ``ibh_accumulate`` (see backend.DeviceAccumulator).
Indices are 0-based.
"""
import numpy as np


class Accumulator:
    def __init__(self, inds=None, weights=None, first_index=False, *, csr=None, n_input=None):
        """``Accumulator(inds, weights; first_index)`` like the reference, or ``csr=(off, idx, w)``."""
        self.first_index = first_index
        if csr is not None:
            off, idx, w = csr
            self.off = np.ascontiguousarray(off, dtype=np.int32)
            self.idx = np.ascontiguousarray(idx, dtype=np.int32)
            self.w = None if w is None else np.ascontiguousarray(w, dtype=np.float32)
        else:
            ls = np.array([len(s) for s in inds], dtype=np.int64)
            self.off = np.concatenate([[0], np.cumsum(ls)]).astype(np.int32)
            self.idx = (np.concatenate([np.asarray(s, dtype=np.int64) for s in inds]) if ls.sum()
                        else np.zeros(0, np.int64)).astype(np.int32)
            self.w = None
            if weights is not None:
                self.w = (np.concatenate([np.asarray(s, dtype=np.float32) for s in weights]) if ls.sum()
                          else np.zeros(0, np.float32)).astype(np.float32)
        self.n_output = len(self.off) - 1
        self.n_input = int(n_input) if n_input is not None else (int(self.idx.max()) + 1 if self.idx.size else 0)

    @property
    def lengths(self):
        return np.diff(self.off)

    @property
    def stencils(self):
        """Reference layout: ``{len: (rows, idx[len, n], w[len, n] | None)}`` (accumulator.jl:46-61)."""
        ls = self.lengths
        out = {}
        seen = []
        for l in ls:
            if int(l) not in seen:
                seen.append(int(l))
        for l in seen:
            rows = np.nonzero(ls == l)[0]
            gather = self.off[rows][None, :] + np.arange(l)[:, None]
            idx = self.idx[gather] if l else np.zeros((0, rows.size), np.int32)
            w = None if self.w is None else (self.w[gather] if l else np.zeros((0, rows.size), np.float32))
            out[l] = (rows, idx, w)
        return out

    def __call__(self, *a, **k):
        raise TypeError("host Accumulator is a container; convert with to_backend(acc, hip) and call it on the GPU "
                        "(there is no CPU evaluation path)")
