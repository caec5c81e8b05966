"""``HipArray``: the Python mirror of the device array type of the reference-side binding (julia/IBHip.jl).

In Julia a user closure such as /root/reference/test/advection.jl:67-83 mixes the grid operators with broadcast
arithmetic (``ud .-= green_gauss(part, @. (uL + uR) * Cf / 2 + abs(Cf) * (uL - uR) / 2, dim)``).  The binding gives the
device array a ``Base.Broadcast`` style whose nodes are the elementwise kernels of libibhip (``ibh_ew_*``); this class
does the same through Python's operator protocol, so that the very same expression tree runs through the very same C
entry points.  torch only owns the memory (column-major Float32).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import c_vp, call

ADD, SUB, MUL, DIV, MAX, MIN, SUM = range(7)
ABS, NEG, SQRT, COPY = 16, 17, 18, 19


def _B():
    from . import backend
    return backend


class HipArray:
    """Device array ``(n,)`` or ``(n, nv)``, column-major Float32; arithmetic = one ``ibh_ew_*`` launch per node.
    The class itself is the ``conv_to_backend`` converter: ``dom(f, args..., conv_to_backend=ibamd.HipArray, ...)``."""

    __array_priority__ = 1000

    def __init__(self, t):
        B = _B()
        if isinstance(t, HipArray):
            t = t.t
        elif not isinstance(t, torch.Tensor):
            t = B.hip(np.asarray(t))
        t, _, ld = B._field(t)
        if t.ndim == 2 and t.shape[1] > 1 and ld != t.shape[0]:
            t = t.T.contiguous().T  # broadcast kernels want the columns back to back
        self.t = t

    # ---- array protocol
    @property
    def shape(self):
        return tuple(self.t.shape)

    @property
    def ndim(self):
        return self.t.ndim

    def __len__(self):
        return self.t.shape[0]

    @property
    def n(self):
        return int(self.t.shape[0])

    @property
    def nv(self):
        return 1 if self.t.ndim == 1 else int(self.t.shape[1])

    def similar(self):
        """``similar(a)``."""
        return HipArray(_B()._like(self.t, self.n))

    def copy(self):
        out = self.similar()
        _B()._stream()
        call("ibh_ew_unary", COPY, self.t.numel(), c_vp(self.t.data_ptr()), c_vp(out.t.data_ptr()))
        return out

    def col(self, j):
        """``@view a[:, j]`` (1-based like the reference): aliases the parent's memory."""
        if self.ndim != 2:
            raise IndexError("col() of a vector")
        return HipArray(self.t[:, j - 1])

    def fill(self, value):
        """``a .= value``."""
        _B()._stream()
        call("ibh_ew_fill", self.t.numel(), C.c_float(float(value)), c_vp(self.t.data_ptr()))
        return self

    def to_host(self):
        return _B().to_host(self.t)

    def __getitem__(self, i):
        raise TypeError("scalar indexing of a HipArray; copy it back with to_host()")

    # ---- broadcast nodes
    def _binary(self, op, other, reverse=False, out=None):
        B = _B()
        a, b = (other, self) if reverse else (self, other)

        def operand(x):
            if isinstance(x, HipArray):
                return x, 0.0
            if isinstance(x, (int, float, np.floating, np.integer)):
                return None, float(x)
            raise TypeError(f"cannot broadcast a HipArray with {type(x).__name__} (convert with HipArray(...))")
        (fa, sa), (fb, sb) = operand(a), operand(b)
        fields = [f for f in (fa, fb) if f is not None]
        n = fields[0].n
        nv = max(f.nv for f in fields)
        for f in fields:
            if f.n != n or f.nv not in (1, nv):
                raise ValueError(f"shapes {tuple(x.shape for x in fields)} do not broadcast")
        if out is None:
            out = HipArray(B.colmajor_empty(n) if nv == 1 and all(f.ndim == 1 for f in fields) else B.colmajor_empty(n, nv))
        elif out.n != n or out.nv != nv:
            raise ValueError("in-place broadcast changes the shape")
        B._stream()
        call("ibh_ew_binary", op, n, nv, c_vp(fa.t.data_ptr()) if fa is not None else c_vp(None),
             fa.nv if fa is not None else 0, C.c_float(sa), c_vp(fb.t.data_ptr()) if fb is not None else c_vp(None),
             fb.nv if fb is not None else 0, C.c_float(sb), c_vp(out.t.data_ptr()))
        return out

    def _unary(self, op):
        out = self.similar()
        _B()._stream()
        call("ibh_ew_unary", op, self.t.numel(), c_vp(self.t.data_ptr()), c_vp(out.t.data_ptr()))
        return out

    def __add__(self, o): return self._binary(ADD, o)
    def __radd__(self, o): return self._binary(ADD, o, reverse=True)
    def __sub__(self, o): return self._binary(SUB, o)
    def __rsub__(self, o): return self._binary(SUB, o, reverse=True)
    def __mul__(self, o): return self._binary(MUL, o)
    def __rmul__(self, o): return self._binary(MUL, o, reverse=True)
    def __truediv__(self, o): return self._binary(DIV, o)
    def __rtruediv__(self, o): return self._binary(DIV, o, reverse=True)
    def __iadd__(self, o): return self._binary(ADD, o, out=self)   # `a .+= o`
    def __isub__(self, o): return self._binary(SUB, o, out=self)   # `a .-= o`
    def __imul__(self, o): return self._binary(MUL, o, out=self)
    def __itruediv__(self, o): return self._binary(DIV, o, out=self)
    def __neg__(self): return self._unary(NEG)
    def __abs__(self): return self._unary(ABS)

    def maximum_with(self, o):
        """``max.(a, o)``."""
        return self._binary(MAX, o)

    def minimum_with(self, o):
        """``min.(a, o)``."""
        return self._binary(MIN, o)

    def sqrt(self):
        return self._unary(SQRT)

    def _reduce(self, op):
        out = torch.empty(1, dtype=torch.float32, device=self.t.device)
        _B()._stream()
        call("ibh_ew_reduce", op, self.t.numel(), c_vp(self.t.data_ptr()), c_vp(out.data_ptr()))
        return float(out.item())

    def maximum(self):
        """``maximum(a)``."""
        return self._reduce(MAX)

    def minimum(self):
        return self._reduce(MIN)

    def sum(self):
        return self._reduce(SUM)


def unwrap(x):
    return x.t if isinstance(x, HipArray) else x


def rewrap(result, like_hiparray):
    """Wrap the tensor(s) an operator returns when it was called with HipArray operands."""
    if not like_hiparray:
        return result
    if isinstance(result, torch.Tensor):
        return HipArray(result)
    if isinstance(result, tuple):
        return tuple(rewrap(r, True) for r in result)
    return result
