"""``HipArray``: the Python mirror of the device array type of the reference-side binding (julia/IBHip.jl).

In Julia a user closure such as /root/reference/test/advection.jl:67-83 mixes the grid operators with broadcast
arithmetic (``ud .-= green_gauss(part, @. (uL + uR) * Cf / 2 + abs(Cf) * (uL - uR) / 2, dim)``).  The binding gives the
device array a ``Base.Broadcast`` style whose nodes are the elementwise kernels of libibhip (``ibh_ew_*``); this class
does the same through Python's operator protocol, so that the very same expression tree runs through the very same C
entry points.  torch only owns the memory (column-major Float32).
"""
import ctypes as C
import weakref

import numpy as np
import torch

from . import _lib
from ._lib import c_vp, call

ADD, SUB, MUL, DIV, MAX, MIN, SUM = range(7)
ABS, NEG, SQRT, COPY = 16, 17, 18, 19


_backend = None


def _B():
    global _backend
    if _backend is None:
        from . import backend
        _backend = backend
    return _backend


PUSH_ARRAY, PUSH_SCALAR = 32, 33
_MAX_PROG, _MAX_ARR, _MAX_SCAL, _MAX_DEPTH = 48, 8, 8, 8


class HipArray:
    """Device array ``(n,)`` or ``(n, nv)``, column-major Float32.  Arithmetic between HipArrays and scalars builds an
    expression (as a Julia broadcast does before it is materialised); the whole expression runs as ONE launch
    (``ibh_ew_eval``: Julia's broadcast fusion) when its value is needed -- by an operator, ``to_host``, an in-place
    update (``ud -= expr`` evaluates ``ud - expr`` straight into ``ud``).  ``HipArray.fuse = False`` evaluates node by
    node (``ibh_ew_binary`` / ``ibh_ew_unary``), bit-identical.
    The class itself is the ``conv_to_backend`` converter: ``dom(f, args..., conv_to_backend=ibamd.HipArray, ...)``."""

    __array_priority__ = 1000
    fuse = True

    def __init__(self, t):
        B = _B()
        if isinstance(t, HipArray):
            t = t.t
        elif not isinstance(t, torch.Tensor):
            t = B.hip(np.asarray(t))
        t, _, ld = B._field(t)
        if t.ndim == 2 and t.shape[1] > 1 and ld != t.shape[0]:
            t = t.T.contiguous().T  # broadcast kernels want the columns back to back
        self._t = t
        self._expr = None               # pending (op, a, b) / (op, a): operands HipArray or float
        self._meta = (int(t.shape[0]), 1 if t.ndim == 1 else int(t.shape[1]), t.ndim)
        self._deps = weakref.WeakSet()  # pending expressions that read this array (materialised before it is written)

    @classmethod
    def _pending(cls, expr, n, nv, ndim):
        self = cls.__new__(cls)
        self._t = None
        self._expr = expr
        self._meta = (n, nv, ndim)
        self._deps = weakref.WeakSet()
        for leaf in self._leaves():
            leaf._deps.add(self)
        return self

    # ---- the value: materialises a pending expression (one launch)
    @property
    def t(self):
        if self._t is None:
            n, nv, ndim = self._meta
            out = _B().colmajor_empty(n) if ndim == 1 else _B().colmajor_empty(n, nv)
            self._evaluate_into(out)
            self._t, self._expr = out, None
        return self._t

    @t.setter
    def t(self, value):
        self._t, self._expr = value, None

    def _leaves(self):
        """Materialised arrays a pending expression reads."""
        if self._expr is None:
            return [self]
        out = []
        for x in self._expr[1:]:
            if isinstance(x, HipArray):
                out += x._leaves()
        return out

    def _flush_readers(self):
        """Before this array's memory is written: evaluate the pending expressions that read it."""
        for d in list(self._deps):
            if d._t is None:
                d.t  # noqa: B018 (materialises)
        self._deps.clear()

    def _emit(self, prog, arrs, scal):
        """Postfix program of this node; returns the stack depth it needs (None: does not fit one launch)."""
        if self._expr is None:
            key = self._t.data_ptr()
            for k, a in enumerate(arrs):
                if a._t.data_ptr() == key and a._meta == self._meta:
                    break
            else:
                if len(arrs) == _MAX_ARR:
                    return None
                arrs.append(self)
                k = len(arrs) - 1
            prog.append(PUSH_ARRAY | (k << 8))
            return 1
        op, depth, width = self._expr[0], 0, 0
        for x in self._expr[1:]:
            if isinstance(x, HipArray):
                d = x._emit(prog, arrs, scal)
                if d is None:
                    return None
            else:
                if x in scal:
                    k = scal.index(x)
                else:
                    if len(scal) == _MAX_SCAL:
                        return None
                    scal.append(x)
                    k = len(scal) - 1
                prog.append(PUSH_SCALAR | (k << 8))
                d = 1
            depth = max(depth, width + d)
            width += 1
        prog.append(op)
        if len(prog) > _MAX_PROG or depth > _MAX_DEPTH:
            return None
        return depth

    def _reads_other_shape(self, out):
        """True if an array this pending expression reads overlaps the tensor ``out`` without being the same elements
        in the same layout (elementwise evaluation in place is then not safe)."""
        lo, hi = out.data_ptr(), out.data_ptr() + out.numel() * out.element_size()
        seen, stack = set(), [self]
        while stack:
            x = stack.pop()
            if id(x) in seen:
                continue
            seen.add(id(x))
            if x._t is None:
                stack.extend(y for y in x._expr[1:] if isinstance(y, HipArray))
                continue
            t = x._t
            a, b = t.data_ptr(), t.data_ptr() + t.numel() * t.element_size()
            if a < hi and lo < b and not (a == lo and t.shape == out.shape and t.stride() == out.stride()):
                return True
        return False

    def _evaluate_into(self, out):
        """Value of the pending expression into the tensor ``out`` (may alias an operand: the kernel is elementwise)."""
        B = _B()
        prog, arrs, scal = [], [], []
        if self._emit(prog, arrs, scal) is None:
            # too large for one launch: materialise the operands that are expressions themselves, then this node
            for x in self._expr[1:]:
                if isinstance(x, HipArray):
                    x.t  # noqa: B018
            prog, arrs, scal = [], [], []
            if self._emit(prog, arrs, scal) is None:
                raise RuntimeError("broadcast expression does not fit ibh_ew_eval")
        n, nv, _ = self._meta
        P = (C.c_int32 * len(prog))(*prog)
        A = (C.c_void_p * max(len(arrs), 1))(*[a._t.data_ptr() for a in arrs])
        V = (C.c_int32 * max(len(arrs), 1))(*[a._meta[1] for a in arrs])
        S = (C.c_float * max(len(scal), 1))(*scal)
        B._stream()
        call("ibh_ew_eval", n, nv, len(prog), P, len(arrs), A, V, len(scal), S, c_vp(out.data_ptr()))

    # ---- array protocol
    @property
    def shape(self):
        n, nv, ndim = self._meta
        return (n,) if ndim == 1 else (n, nv)

    @property
    def ndim(self):
        return self._meta[2]

    def __len__(self):
        return self._meta[0]

    @property
    def n(self):
        return self._meta[0]

    @property
    def nv(self):
        return self._meta[1]

    def similar(self):
        """``similar(a)``."""
        n, nv, ndim = self._meta
        return HipArray(_B().colmajor_empty(n) if ndim == 1 else _B().colmajor_empty(n, nv))

    def copy(self):
        out = self.similar()
        _B()._stream()
        call("ibh_ew_unary", COPY, self.t.numel(), c_vp(self.t.data_ptr()), c_vp(out.t.data_ptr()))
        return out

    def col(self, j):
        """``@view a[:, j]`` (1-based like the reference): aliases the parent's memory."""
        if self.ndim != 2:
            raise IndexError("col() of a vector")
        v = HipArray(self.t[:, j - 1])
        v._deps = self._deps   # a write through either name flushes the readers of both
        return v

    def fill(self, value):
        """``a .= value``."""
        self._flush_readers()
        _B()._stream()
        call("ibh_ew_fill", self.t.numel(), C.c_float(float(value)), c_vp(self.t.data_ptr()))
        return self

    def to_host(self):
        return _B().to_host(self.t)

    def __getitem__(self, i):
        raise TypeError("scalar indexing of a HipArray; copy it back with to_host()")

    # ---- broadcast nodes
    @staticmethod
    def _operand(x):
        if isinstance(x, HipArray):
            return x
        if isinstance(x, (int, float, np.floating, np.integer)):
            return float(np.float32(x))
        raise TypeError(f"cannot broadcast a HipArray with {type(x).__name__} (convert with HipArray(...))")

    def _binary(self, op, other, reverse=False, out=None):
        a, b = (other, self) if reverse else (self, other)
        a, b = self._operand(a), self._operand(b)
        fields = [f for f in (a, b) if isinstance(f, HipArray)]
        n = fields[0].n
        nv = max(f.nv for f in fields)
        for f in fields:
            if f.n != n or f.nv not in (1, nv):
                raise ValueError(f"shapes {tuple(x.shape for x in fields)} do not broadcast")
        ndim = 1 if nv == 1 and all(f.ndim == 1 for f in fields) else 2
        if out is not None and (out.n != n or out.nv != nv):
            raise ValueError("in-place broadcast changes the shape")
        if HipArray.fuse:
            node = HipArray._pending((op, a, b), n, nv, ndim)
            if out is None:
                return node
            # `out .= out op other`: the fused expression straight into out's memory
            out.t  # noqa: B018 (out is an operand: it must hold a value)
            readers = [d for d in out._deps if d is not node]
            for d in readers:
                if d._t is None:
                    d.t  # noqa: B018
            if node._reads_other_shape(out._t):
                # a leaf aliases out's memory with another shape (`P ./= P[:, 1]`): threads of other columns would race
                # with the write; Julia's broadcast_unalias copies in that case -- so does this
                tmp = torch.empty_like(out._t)
                node._evaluate_into(tmp)
                out._t.copy_(tmp)
            else:
                node._evaluate_into(out._t)
            node._t, node._expr = out._t, None
            out._deps.clear()
            return out
        B = _B()
        if out is None:
            out = HipArray(B.colmajor_empty(n) if ndim == 1 else B.colmajor_empty(n, nv))
        else:
            out._flush_readers()
        fa, sa = (a, 0.0) if isinstance(a, HipArray) else (None, a)
        fb, sb = (b, 0.0) if isinstance(b, HipArray) else (None, b)
        B._stream()
        call("ibh_ew_binary", op, n, nv, c_vp(fa.t.data_ptr()) if fa is not None else c_vp(None),
             fa.nv if fa is not None else 0, C.c_float(sa), c_vp(fb.t.data_ptr()) if fb is not None else c_vp(None),
             fb.nv if fb is not None else 0, C.c_float(sb), c_vp(out.t.data_ptr()))
        return out

    def _unary(self, op):
        if HipArray.fuse:
            n, nv, ndim = self._meta
            return HipArray._pending((op, self), n, nv, ndim)
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
