"""ORACLE (test infrastructure only; never imported by the product path).

ctypes binding of oracle/csrc/residual.c -- the plain C (OpenMP) restatement of the scalar residual sweep
(/root/reference/test/advection.jl:67-83 over the operators of /root/reference/src/ImmersedBoundary.jl:899-1157).
`CPart(part)` flattens a Partition (oracle or product host object: same fields) into the arrays the C code
reads; the bucketed face accumulators (src/accumulator.jl:12-16) become CSR in stencil order.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_here = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_here, "_build", "libiboracle.so")
_i32p = C.POINTER(C.c_int32)
_f32p = C.POINTER(C.c_float)


class _Part(C.Structure):
    _fields_ = [("nd", C.c_int32), ("nc", C.c_int32), ("spacing", _f32p), ("nf", C.c_int32 * 3),
                ("owners", _i32p * 3), ("neighbors", _i32p * 3),
                ("loff", _i32p * 3), ("lidx", _i32p * 3), ("roff", _i32p * 3), ("ridx", _i32p * 3)]


def build():
    subprocess.run(["make", "-C", _here], check=True, capture_output=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        _lib.ibo_max_threads.restype = C.c_int
        for f in (_lib.ibo_residual_advection_faithful, _lib.ibo_residual_advection_fused):
            f.restype = C.c_int
            f.argtypes = [C.POINTER(_Part), _f32p, _f32p, C.c_int64, _f32p]
        _lib.ibo_residual_euler_faithful.restype = C.c_int
        _lib.ibo_residual_euler_faithful.argtypes = [C.POINTER(_Part), _f32p, C.c_int64, C.c_float, C.c_float, _f32p,
                                                     C.c_int64]
    return _lib


def _csr(acc, n_rows):
    """Bucketed stencils {len: (rows, idx[len x n], w)} -> CSR (offsets, face ids) in stencil order."""
    cnt = np.zeros(n_rows, dtype=np.int64)
    for l, (rows, idx, _w) in acc.stencils.items():
        cnt[rows] = l
    off = np.zeros(n_rows + 1, dtype=np.int32)
    np.cumsum(cnt, out=off[1:])
    out = np.zeros(int(off[-1]), dtype=np.int32)
    for l, (rows, idx, _w) in acc.stencils.items():
        for k in range(int(l)):
            out[off[rows] + k] = idx[k]
    return off, out


class CPart:
    def __init__(self, part):
        self.nd = int(part.ndims)
        self.nc = int(part.spacing.shape[0])
        self.keep = []
        p = _Part()
        p.nd, p.nc = self.nd, self.nc
        sp = np.asfortranarray(part.spacing, dtype=np.float32)
        self.keep.append(sp)
        p.spacing = sp.ctypes.data_as(_f32p)
        for d in range(self.nd):
            o, n = part.face_owners_neighbors[d + 1]
            o = np.ascontiguousarray(o, dtype=np.int32)
            n = np.ascontiguousarray(n, dtype=np.int32)
            lo, li = _csr(part.face_accumulators[(d + 1, False)], self.nc)
            ro, ri = _csr(part.face_accumulators[(d + 1, True)], self.nc)
            self.keep += [o, n, lo, li, ro, ri]
            p.nf[d] = o.size
            p.owners[d], p.neighbors[d] = o.ctypes.data_as(_i32p), n.ctypes.data_as(_i32p)
            p.loff[d], p.lidx[d] = lo.ctypes.data_as(_i32p), li.ctypes.data_as(_i32p)
            p.roff[d], p.ridx[d] = ro.ctypes.data_as(_i32p), ri.ctypes.data_as(_i32p)
        self.c = p

    def residual_advection(self, u, Cv, fused=False, threads=None):
        """ud of test/advection.jl:67-83 (ud starting from zero).  u: (nc,), Cv: (nc, nd)."""
        L = lib()
        if threads:
            L.ibo_set_threads(int(threads))
        u = np.ascontiguousarray(u, dtype=np.float32)
        Cf = np.asfortranarray(Cv, dtype=np.float32)
        ud = np.empty(self.nc, dtype=np.float32)
        fn = L.ibo_residual_advection_fused if fused else L.ibo_residual_advection_faithful
        rc = fn(C.byref(self.c), u.ctypes.data_as(_f32p), Cf.ctypes.data_as(_f32p), self.nc, ud.ctypes.data_as(_f32p))
        if rc:
            raise MemoryError("oracle C sweep failed")
        return ud


    def residual_euler(self, P, R=283.0, gamma=1.4, threads=None):
        """R of the Euler sweep: P (nc, nd+2) = [p T u v (w)]; returns (nc, nd+2)."""
        L = lib()
        if threads:
            L.ibo_set_threads(int(threads))
        Pf = np.asfortranarray(P, dtype=np.float32)
        out = np.empty(Pf.shape, dtype=np.float32, order="F")
        rc = L.ibo_residual_euler_faithful(C.byref(self.c), Pf.ctypes.data_as(_f32p), self.nc, C.c_float(R),
                                           C.c_float(gamma), out.ctypes.data_as(_f32p), self.nc)
        if rc:
            raise MemoryError("oracle C sweep failed")
        return out


def max_threads():
    return int(lib().ibo_max_threads())
