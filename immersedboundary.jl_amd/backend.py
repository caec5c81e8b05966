"""GPU backend: the reference's plug-in surface re-implemented over libibhip.

Mirrors, for the hot path only,
  * ``conv_to_backend`` / ``conv_from_backend`` converters and ``to_backend`` for the structs
    (/root/reference/src/arraybends.jl:14-77, src/ImmersedBoundary.jl:788-790, :846-855);
  * the grid operators on a Partition (src/ImmersedBoundary.jl:873-1157) -- same names,
    argument order and return shapes (fresh arrays), ``dim`` 1-based;
  * ``(dom::Domain)(f, args...)`` (:820-864) and ``impose_bc!`` (:1197-1247).

Device memory, streams and (elsewhere) torch.distributed come from PyTorch -- plumbing only;
all arithmetic is in the HIP kernels of libibhip.so, reached through the C ABI with raw device
pointers.  Nothing here computes on the CPU: operators raise if handed host arrays, and loading
fails loudly if the library is missing.

Device field layout = the reference's column-major ``(ncells, nvars)``: a torch tensor of shape
``(n, nv)`` with strides ``(1, ld)``.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import call, c_vp
from .accumulator import Accumulator
from .domain import Boundary, Partition, Surface

IBH_FORCE_GENERAL = 1
IBH_IMAGE_ONLY = 2
IBH_PASS_A_ONLY = 4
IBH_PASS_B_ONLY = 8
IBH_EXACT = 16
IBH_PHASE_INTERIOR = 32
IBH_PHASE_BOUNDARY = 64
IBH_NO_FUSE = 128
IBH_SWEEP_ONLY = 256
IBH_FORCE_MIXED = 512
IBH_NO_QUAD = 1024

_initialised = {}


_have_gpu = None


def _dev():
    global _have_gpu
    if _have_gpu is None:
        _have_gpu = bool(torch.cuda.is_available())
    if not _have_gpu:
        raise _lib.IbhError("no HIP device visible: the ImmersedBoundary hot path runs on the GPU only "
                            "(there is no CPU fallback)")
    d = torch.cuda.current_device()
    dev = _initialised.get(d)
    if dev is None:
        call("ibh_init", d)
        dev = _initialised[d] = torch.device("cuda", d)
    return dev


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """The library launches on torch's current stream (asked for through the raw-handle query: the operators of a user
    closure are host-bound, and `torch.cuda.current_stream()` alone costs as much as a launch)."""
    if _raw_stream is not None:
        call("ibh_set_stream", c_vp(_raw_stream(torch.cuda.current_device())))
    else:
        call("ibh_set_stream", c_vp(torch.cuda.current_stream().cuda_stream))


def _ptr(t):
    return c_vp(t.data_ptr()) if t is not None else c_vp(None)


def _hptr(a):
    return a.ctypes.data_as(c_vp)


# ---------------------------------------------------------------------------
# converters (conv_to_backend / conv_from_backend)
# ---------------------------------------------------------------------------
def colmajor_empty(n, nv=None, dtype=torch.float32):
    """Uninitialised device array ``(n,)`` or ``(n, nv)`` in column-major layout."""
    dev = _dev()
    if nv is None:
        return torch.empty(n, dtype=dtype, device=dev)
    return torch.empty((nv, n), dtype=dtype, device=dev).T


def hip(a):
    """``conv_to_backend``: host array -> device array (column-major)."""
    dev = _dev()
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    a = np.asarray(a)
    if a.dtype == np.float64:
        a = a.astype(np.float32)
    if a.ndim == 1:
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    if a.ndim == 2:
        return torch.from_numpy(np.ascontiguousarray(a.T)).to(dev).T
    raise ValueError("hip(): only 1-D and 2-D fields are supported")


def to_host(t):
    """``conv_from_backend``: device array -> numpy array."""
    if not isinstance(t, torch.Tensor):
        t = t.t  # HipArray
    return np.ascontiguousarray(t.detach().cpu().numpy())


def _hipaware(fn):
    """Operators take and return ``HipArray`` (hiparray.py: broadcast arithmetic through ``ibh_ew_*``) when they are
    called with one: the Python form of Julia's dispatch on the array type (julia/IBHip.jl)."""
    import functools

    H = {}

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        if not H:
            from . import hiparray
            H["A"], H["rewrap"] = hiparray.HipArray, hiparray.rewrap
        HipArray = H["A"]
        hit = False

        def un(x):
            nonlocal hit
            if isinstance(x, HipArray):
                hit = True
                return x.t
            if isinstance(x, tuple):
                return tuple(un(v) for v in x)
            return x
        # an operator that writes into a HipArray (out=...): pending broadcasts that read it are evaluated first, so
        # that they see the old values like Julia's eager broadcast would
        if kwargs and isinstance(kwargs.get("out"), HipArray):
            kwargs["out"]._flush_readers()
        a = [un(x) for x in args]
        k = {key: un(v) for key, v in kwargs.items()} if kwargs else kwargs
        out = fn(*a, **k)
        return H["rewrap"](out, True) if hit else out
    return wrapper


def _field(t, n=None):
    """Validate a device field and return (tensor, nv, ld)."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError("expected a device array (convert with ibamd.hip); there is no CPU path")
    if t.dtype != torch.float32:
        raise TypeError("fields must be Float32")
    if t.ndim == 1:
        if t.stride(0) != 1:
            t = t.contiguous()
        nv, ld = 1, t.shape[0]
    elif t.ndim == 2:
        if t.stride(0) != 1 or (t.shape[1] > 1 and t.stride(1) < t.shape[0]):
            t = t.T.contiguous().T
        nv, ld = t.shape[1], (t.stride(1) if t.shape[1] > 1 else t.shape[0])
    else:
        raise ValueError("fields are (n,) or (n, nv)")
    if n is not None and t.shape[0] != n:
        raise ValueError(f"field has {t.shape[0]} rows, expected {n}")
    return t, nv, ld


def _field_inplace(t, n=None, what="destination"):
    """A device field that a kernel WRITES: it must already be Float32 with the cell index fastest (column-major,
    ``ibamd.hip`` / ``colmajor_empty`` give that), because a silent layout copy would take the update and drop it."""
    f, nv, ld = _field(t, n)
    if f.data_ptr() != t.data_ptr() or f.stride() != t.stride():
        raise TypeError(f"{what} must be a column-major Float32 device array (cell index fastest, e.g. ibamd.hip(a) or "
                        f"colmajor_empty(n, nv)); got strides {tuple(t.stride())} for shape {tuple(t.shape)}: it would "
                        "be copied and the update lost")
    return f, nv, ld


def _like(t, n):
    return colmajor_empty(n) if t.ndim == 1 else colmajor_empty(n, t.shape[1])


# ---------------------------------------------------------------------------
# structs on the device
# ---------------------------------------------------------------------------
class DevicePartition:
    """Device-resident Partition: ``to_backend(part, hip)`` (uploads once, caches the handle)."""

    def __init__(self, part: Partition):
        dev = _dev()
        self.host = part
        self.id = part.id
        self.nd = part.ndims
        self.nc = part.spacing.shape[0]
        nd, nc = self.nd, self.nc
        spacing = np.asfortranarray(part.spacing, dtype=np.float32)
        centers = np.asfortranarray(part.centers, dtype=np.float32)
        self.nf = [int(part.face_owners_neighbors[d + 1][0].shape[0]) for d in range(nd)]
        nf = np.array(self.nf, dtype=np.int32)
        keep = []

        def parr(arrs):
            arrs = [np.ascontiguousarray(a, dtype=np.int32) for a in arrs]
            keep.append(arrs)
            return (c_vp * nd)(*[a.ctypes.data for a in arrs])

        owners = parr([part.face_owners_neighbors[d + 1][0] for d in range(nd)])
        neighbors = parr([part.face_owners_neighbors[d + 1][1] for d in range(nd)])
        loff = parr([part.face_accumulators[(d + 1, False)].off for d in range(nd)])
        lidx = parr([part.face_accumulators[(d + 1, False)].idx for d in range(nd)])
        roff = parr([part.face_accumulators[(d + 1, True)].off for d in range(nd)])
        ridx = parr([part.face_accumulators[(d + 1, True)].idx for d in range(nd)])
        iid = np.ascontiguousarray(part.image_in_domain, dtype=np.int32)
        dom = np.ascontiguousarray(part.domain, dtype=np.int32)
        h = c_vp()
        call("ibh_partition_create", C.byref(h), nd, nc, spacing.ctypes.data_as(c_vp), centers.ctypes.data_as(c_vp),
             nf.ctypes.data_as(c_vp), owners, neighbors, loff, lidx, roff, ridx, int(iid.size), _hptr(iid),
             _hptr(dom), int(getattr(part, "block_size", 0)), 0)
        self.handle = h
        # device copies of what user closures read (part.spacing[:, dim], part.centers)
        self.spacing = hip(part.spacing)
        self.centers = hip(part.centers)
        self.domain = torch.from_numpy(dom).to(dev)
        self.image = torch.from_numpy(np.ascontiguousarray(part.image, dtype=np.int32)).to(dev)
        self.image_in_domain = torch.from_numpy(iid).to(dev)
        info = (C.c_int64 * 24)()
        call("ibh_partition_info", h, info, 24)
        self.info = dict(full_blocks=info[0], irregular_cells=info[1], sides_same=info[2], sides_mirror=info[3],
                         sides_coarse=info[4], sides_fine=info[5], sides_general=info[6], interior_blocks=info[7],
                         fusable_blocks=info[8], workspace_blocks=info[9],
                         image_blocks_all_eligible=bool(info[10]), image_blocks=info[11],
                         quads=info[12] if self.nd == 2 else 0, quad_singles=info[13], image_quads=info[14],
                         image_quad_singles=info[15], row_sweep=bool(info[16]), direct_sides=info[17],
                         quad_pairs=info[18] if self.nd == 2 else 0,   # pair tiles among the blocks outside quads
                         quad_arith_half_sides=info[19] if self.nd == 2 else 0,   # of 8 per quad: halo ids computed, not read
                         # 3-D single-kernel sweeps: rim neighbours of halo cells that are four finer cells
                         rim4_rows=info[12] if self.nd == 3 else 0)

    @property
    def ndims(self):
        return self.nd

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.load().ibh_partition_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class DeviceAccumulator:
    """Device-resident Accumulator; callable like the reference's (accumulator.jl:78-130)."""

    def __init__(self, acc: Accumulator):
        _dev()
        self.n_output, self.n_input = acc.n_output, acc.n_input
        self.first_index = acc.first_index
        h = c_vp()
        w = None if acc.w is None else np.ascontiguousarray(acc.w, dtype=np.float32)
        call("ibh_acc_create", C.byref(h), acc.n_output, acc.n_input, _hptr(acc.off), _hptr(acc.idx),
             _hptr(w) if w is not None else c_vp(None), 0)
        self.handle = h

    def __call__(self, v):
        from .hiparray import HipArray
        if isinstance(v, HipArray):
            return HipArray(self(v.t))
        v, nv, ld = _field(v, self.n_input)
        out = _like(v, self.n_output)
        _stream()
        call("ibh_accumulate", self.handle, _ptr(v), nv, ld, _ptr(out), self.n_output)
        return out

    def diff_add(self, out, a, b):
        """``out .+= acc(a .- b)`` in one launch (``ibh_accumulate_diff_add``); same arithmetic as the separate operations."""
        a, nv, ld = _field(a, self.n_input)
        b, nvb, ldb = _field(b, self.n_input)
        o, nvo, ldo = _field_inplace(out, self.n_output, "out")
        if nvb != nv or nvo != nv or ldb != ld:
            out += self(a - b)
            return out
        _stream()
        call("ibh_accumulate_diff_add", self.handle, _ptr(a), _ptr(b), nv, ld, _ptr(o), ldo)
        return out

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.load().ibh_acc_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class DeviceBoundary:
    """Device-resident Boundary (ImmersedBoundary.jl:406-414)."""

    def __init__(self, b: Boundary):
        self.host = b
        self.ng = int(b.ghost_indices.shape[0])
        acc = b.image_interpolator
        h = c_vp()
        gi = np.ascontiguousarray(b.ghost_indices, dtype=np.int32)
        idm = np.ascontiguousarray(b.image_domain, dtype=np.int32)
        gd = np.ascontiguousarray(b.ghost_distances, dtype=np.float32)
        idist = np.ascontiguousarray(b.image_distances, dtype=np.float32)
        _dev()
        call("ibh_bc_create", C.byref(h), self.ng, _hptr(gi), _hptr(gd), _hptr(idist), int(idm.size), _hptr(idm),
             _hptr(acc.off), _hptr(acc.idx), _hptr(np.ascontiguousarray(acc.w, dtype=np.float32)), 0)
        self.handle = h
        self.ghost_indices = torch.from_numpy(gi).to(_dev())
        self.projections = hip(b.projections)
        self.normals = hip(b.normals)
        self.image_distances = hip(b.image_distances)
        self.ghost_distances = hip(b.ghost_distances)

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.load().ibh_bc_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class DeviceSurface:
    """Device-resident Surface (ImmersedBoundary.jl:328-376): two Accumulators + the areas for the integrals."""

    def __init__(self, s: Surface):
        self.host = s
        self.n = int(s.points.shape[0])
        self.interpolator = DeviceAccumulator(s.interpolator)
        self.offset_interpolator = DeviceAccumulator(s.offset_interpolator)
        self.points, self.normals = hip(s.points), hip(s.normals)
        self.offsets, self.areas = hip(s.offsets), hip(s.areas)

    def __call__(self, u):
        """``surf(u)``: values of the field array ``u`` at the surface control points (:364)."""
        return self.interpolator(u)


def at_offset(surf, u):
    """``at_offset(surf, u)`` (:372): values at the offset sampling points."""
    return to_backend(surf).offset_interpolator(u)


def surface_integral(surf, u):
    """``surface_integral(surf, u)`` (:345-357): sum over the control points of ``areas .* u`` per column --
    one elementwise product and one device reduction per column (``ibh_ew_binary`` / ``ibh_ew_reduce``)."""
    from .hiparray import HipArray
    surf = to_backend(surf)
    U = u if isinstance(u, HipArray) else HipArray(u)
    if U.n != surf.n:
        raise ValueError(f"field has {U.n} rows, the surface {surf.n} control points")
    prod = U * HipArray(surf.areas)
    if U.ndim == 1:
        return np.float32(prod.sum())
    return np.array([HipArray(prod.t[:, v]).sum() for v in range(U.nv)], dtype=np.float32)


def volume_integral(dom, A):
    """``volume_integral(dom, A)`` (:1415-1431) on a device-resident global array: A times the cell volumes (the
    widths come from the Domain's cell table), summed per column on the device."""
    from .hiparray import HipArray
    U = A if isinstance(A, HipArray) else HipArray(A)
    if U.n != len(dom):
        raise ValueError("volume_integral takes a global array (one row per cell of the domain)")
    wd = getattr(dom, "_device_widths", None)
    if wd is None:
        from .mesher import get_cells
        _, widths = get_cells(dom.mesh)
        wd = dom._device_widths = [HipArray(np.ascontiguousarray(widths[d], dtype=np.float32))
                                   for d in range(widths.shape[0])]
    prod = U * wd[0]          # `Ai .*= part.spacing[:, dim]` dim by dim, like the reference
    for w in wd[1:]:
        prod *= w
    if U.ndim == 1:
        return np.float32(prod.sum())
    return np.array([HipArray(prod.t[:, v]).sum() for v in range(U.nv)], dtype=np.float32)


def to_backend(x, converter=hip):
    """``ArrayBackends.to_backend`` (arraybends.jl:14-77) for the structs on the hot path."""
    if isinstance(x, (DevicePartition, DeviceAccumulator, DeviceBoundary, DeviceSurface)):
        return x
    if isinstance(x, Surface):
        if getattr(x, "_device", None) is None:
            x._device = DeviceSurface(x)
        return x._device
    if isinstance(x, Partition):
        if getattr(x, "_device", None) is None:
            x._device = DevicePartition(x)
        return x._device
    if isinstance(x, Accumulator):
        if getattr(x, "_device", None) is None:
            x._device = DeviceAccumulator(x)
        return x._device
    if isinstance(x, Boundary):
        if getattr(x, "_device", None) is None:
            x._device = DeviceBoundary(x)
        return x._device
    if isinstance(x, tuple):
        return tuple(to_backend(v, converter) for v in x)
    if isinstance(x, dict):
        return {k: to_backend(v, converter) for k, v in x.items()}
    if isinstance(x, (np.ndarray, torch.Tensor)):
        return converter(x)
    return x


def _part(p):
    if not isinstance(p, DevicePartition):
        raise TypeError("operators run on a device Partition (to_backend(part, hip)); there is no CPU path")
    return p


# ---------------------------------------------------------------------------
# grid operators (ImmersedBoundary.jl:873-1157)
# ---------------------------------------------------------------------------
def _cell_to_face(name, part, u, dim):
    part = _part(part)
    u, nv, ld = _field(u, part.nc)
    out = _like(u, part.nf[dim - 1])
    _stream()
    call(name, part.handle, dim, _ptr(u), nv, ld, _ptr(out), part.nf[dim - 1])
    return out


@_hipaware
def at_owners(part, u, dim):
    """:879"""
    return _cell_to_face("ibh_at_owners", part, u, dim)


@_hipaware
def at_neighbors(part, u, dim):
    """:889"""
    return _cell_to_face("ibh_at_neighbors", part, u, dim)


@_hipaware
def at_faces(part, u, dim):
    """:899"""
    return _cell_to_face("ibh_at_faces", part, u, dim)


@_hipaware
def face_gradient(part, u, a, b=None):
    """:1039 ``face_gradient(part,u,dim)`` / :1051 ``face_gradient(part,u,grad_u,dim)``."""
    if b is None:
        return _cell_to_face("ibh_face_gradient", part, u, a)
    gu, dim = a, b
    return tuple(face_gradient(part, u, dim) if i == dim else at_faces(part, gu[i - 1], dim)
                 for i in range(1, _part(part).nd + 1))


def _gg(part, uf, dim, uns):
    part = _part(part)
    uf, nv, ld = _field(uf, part.nf[dim - 1])
    out = _like(uf, part.nc)
    _stream()
    call("ibh_green_gauss", part.handle, dim, _ptr(uf), nv, ld, _ptr(out), part.nc, uns)
    return out


@_hipaware
def green_gauss(part, uf, dim):
    """:918"""
    return _gg(part, uf, dim, 0)


@_hipaware
def unsigned_green_gauss(part, uf, dim):
    """:934"""
    return _gg(part, uf, dim, 1)


@_hipaware
def divergent(part, uf):
    """:950"""
    s = green_gauss(part, uf[0], 1)
    for d in range(2, _part(part).nd + 1):
        s = s + green_gauss(part, uf[d - 1], d)
    return s


@_hipaware
def cell_gradient(part, u, dim=None):
    """:965 / :980"""
    part = _part(part)
    if dim is None:
        # the tuple form: all dimensions in one sweep per field (ibh_cell_gradient_nd)
        u, nv, ld = _field(u, part.nc)
        nd, nc = part.nd, part.nc
        _stream()
        if nv == 1 and part.info["full_blocks"] > 0:
            # gradients + sensor back to back: the block sweep writes them in place (no copy out of a workspace)
            buf = colmajor_empty(nc, nd + 1)
            call("ibh_cell_gradient_nd", part.handle, _ptr(u), 1, ld, _ptr(buf), nc, c_vp(buf.data_ptr() + 4 * nd * nc), nc)
        elif nv > 1 and part.info["full_blocks"] > 0:
            # several fields: every field's block sweep writes [grad_1 .. grad_nd, sensor] in place, the tuple's arrays are
            # strided views (columns d, d + nd + 1, ...: leading dimension (nd + 1) nc) -- no copies out of a workspace
            buf = colmajor_empty(nc, nv * (nd + 1))
            call("ibh_cell_gradient_fields", part.handle, _ptr(u), nv, ld, _ptr(buf))
            return tuple(buf[:, d::nd + 1] for d in range(nd))
        else:
            buf = colmajor_empty(nc, nd * nv)
            call("ibh_cell_gradient_nd", part.handle, _ptr(u), nv, ld, _ptr(buf), nc, None, 0)
        return tuple(buf[:, d * nv] if u.ndim == 1 else buf[:, d * nv:(d + 1) * nv] for d in range(nd))
    u, nv, ld = _field(u, part.nc)
    out = _like(u, part.nc)
    _stream()
    call("ibh_cell_gradient", part.handle, dim, _ptr(u), nv, ld, _ptr(out), part.nc)
    return out


def cell_gradient_array(part, u):
    """``cell_gradient(part, u)`` of a scalar field as ONE ``(nc, nd)`` array (the tuple form's buffer: no copies)."""
    g = cell_gradient(part, u)
    base = g[0]
    nd = len(g)
    return torch.as_strided(base, (base.shape[0], nd), (1, base.shape[0]), base.storage_offset())


def _dist(name, part, dim):
    part = _part(part)
    out = colmajor_empty(part.nf[dim - 1])
    _stream()
    call(name, part.handle, dim, _ptr(out))
    return out


def face_distance(part, dim):
    """:995"""
    return _dist("ibh_face_distance", part, dim)


def owner_distance(part, dim):
    """:1010"""
    return _dist("ibh_owner_distance", part, dim)


def neighbor_distance(part, dim):
    """:1024"""
    return _dist("ibh_neighbor_distance", part, dim)


@_hipaware
def JST_sensor(part, p, dim=0):
    """:1077"""
    part = _part(part)
    p, nv, ld = _field(p, part.nc)
    out = _like(p, part.nc)
    _stream()
    call("ibh_jst_sensor", part.handle, dim, _ptr(p), nv, ld, _ptr(out), part.nc)
    return out


@_hipaware
def MUSCL(part, u, du, dim, D=None, high_order=False):
    """:1113"""
    part = _part(part)
    u, nv, ld = _field(u, part.nc)
    du, nv2, ld2 = _field(du, part.nc)
    if nv2 != nv:
        raise ValueError("u and du must have the same shape")
    if ld2 != ld and u.ndim == 2:
        du = du.T.contiguous().T
        u = u.T.contiguous().T
        ld = part.nc
    if D is not None:
        D, nvd, _ = _field(D, part.nc)
        if nvd != 1:
            raise ValueError("D must be a vector")
    nf = part.nf[dim - 1]
    uL, uR = _like(u, nf), _like(u, nf)
    _stream()
    call("ibh_muscl", part.handle, dim, _ptr(u), _ptr(du), nv, ld, _ptr(D), int(bool(high_order)), _ptr(uL),
         _ptr(uR), nf)
    return uL, uR


# ---------------------------------------------------------------------------
# fused residual sweeps
# ---------------------------------------------------------------------------
@_hipaware
def residual_advection(part, u, C_, out=None, flags=0):
    """Fused closure of test/advection.jl:67-83 (``ud`` from zero): returns ``ud``."""
    part = _part(part)
    u, nv, _ = _field(u, part.nc)
    if nv != 1:
        raise ValueError("u must be a scalar field")
    C_, nvc, ldc = _field(C_, part.nc)
    if nvc != part.nd:
        raise ValueError("C must be (nc, nd)")
    if out is not None:
        ud, nvo, _ = _field_inplace(out, part.nc, "out")
        if nvo != 1 or out.ndim != 1:
            raise ValueError("out must be a scalar field (nc,)")
    else:
        ud = torch.zeros(part.nc, dtype=torch.float32, device=u.device)
    _stream()
    call("ibh_residual_advection", part.handle, _ptr(u), _ptr(C_), ldc, _ptr(ud), flags)
    return ud


def residual_advection_repeat(part, u, C_, out, n, flags=0):
    """``n`` sweeps of ``residual_advection`` launched back to back by one C call (``ibh_residual_advection_n``): the step
    loop as a compiled host runs it.  ``out`` is written ``n`` times."""
    part = _part(part)
    u, nv, _ = _field(u, part.nc)
    C_, nvc, ldc = _field(C_, part.nc)
    ud, nvo, _ = _field_inplace(out, part.nc, "out")
    if nv != 1 or nvc != part.nd or nvo != 1:
        raise ValueError("u (nc,), C (nc, nd), out (nc,)")
    _stream()
    call("ibh_residual_advection_n", part.handle, _ptr(u), _ptr(C_), ldc, _ptr(ud), flags, int(n))
    return ud


# ---------------------------------------------------------------------------
# an explicit solver step, device resident (test/advection.jl:30-89)
# ---------------------------------------------------------------------------
class BCSet:
    """An ordered list of ``impose_bc!`` calls (ImmersedBoundary.jl:1197-1247) whose closures the library knows --
    ``(name, value)``: ``do bdry, u; value end``; ``(name, "copy")``: ``do bdry, u; copy(u) end`` (the three calls of
    test/advection.jl:30-46) -- on the one partition ``ipart`` of ``dom`` whose local cell order is the global one.
    ``apply(u)`` has the semantics of the sequential calls; boundaries that do not read each other's ghost cells share
    their two launches (``n_levels`` pairs in all)."""

    def __init__(self, dom, specs, ipart=1):
        part = dom.partitions[ipart]
        if not np.array_equal(part.domain, np.arange(part.domain.size)):
            raise ValueError("BCSet: the partition's local cell order must be the global one (a one-partition domain)")
        self._bcs = []
        modes, values = [], []
        for name, spec in specs:
            self._bcs.append(to_backend(dom.boundaries[name][ipart]))
            modes.append(1 if isinstance(spec, str) and spec == "copy" else 0)
            values.append(0.0 if modes[-1] else float(spec))
        n = len(self._bcs)
        arr = (c_vp * n)(*[bd.handle for bd in self._bcs])
        m = np.asarray(modes, dtype=np.int32)
        v = np.asarray(values, dtype=np.float32)
        h = c_vp()
        _dev()
        call("ibh_bcset_create", C.byref(h), n, arr, _hptr(m), _hptr(v))
        self.handle = h
        ng, nl = C.c_int32(0), C.c_int32(0)
        call("ibh_bcset_info", h, C.byref(ng), C.byref(nl))
        self.n_ghost, self.n_levels = int(ng.value), int(nl.value) & 0xffff
        self.n_direct_levels = int(nl.value) >> 16      # levels blended straight into the field (one launch each)

    def healthy(self):
        """False if a barrier of the one-launch form ever gave up (its bound is far beyond anything a healthy launch needs)."""
        ng, nl = C.c_int32(0), C.c_int32(0)
        call("ibh_bcset_info", self.handle, C.byref(ng), C.byref(nl))
        return ng.value >= 0

    def apply(self, u):
        """The boundary conditions on the device field ``u`` (in place)."""
        from .hiparray import HipArray
        if isinstance(u, HipArray):
            u._flush_readers()
            u = u.t
        f, nv, _ = _field_inplace(u, what="BCSet field")
        if nv != 1:
            raise ValueError("BCSet applies to a scalar field")
        _stream()
        call("ibh_bcset_apply", self.handle, _ptr(f))
        return u

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.load().ibh_bcset_destroy(self.handle)
        except Exception:  # noqa: BLE001
            pass


def timestep_advection(part, C_, scale=1.0, out=None):
    """``scale * 0.5 / maximum(max.(unsigned_green_gauss(part, at_faces(part, C[:, d], d), d) ...))`` (test/advection.jl:
    52-59, :65) as a one-element device tensor: the step loop never reads it back."""
    part = _part(part)
    C_, nvc, ldc = _field(C_, part.nc)
    if nvc != part.nd:
        raise ValueError("C must be (nc, nd)")
    dt = out if out is not None else torch.empty(1, dtype=torch.float32, device=C_.device)
    _stream()
    call("ibh_timestep_advection", part.handle, _ptr(C_), ldc, C.c_float(scale), _ptr(dt))
    return dt


def step_advection(part, u, C_, dt, bcs=None, out=None, next_dt=None, scale=1.0):
    """One ``march!`` of test/advection.jl:61-89 without the host: ``out = u + dt * R(u)`` (sweep and update in one launch
    where the quad sweep applies), then the boundary conditions ``bcs`` (a ``BCSet``) on ``out``.  ``dt``: one-element
    device tensor (``timestep_advection``).  ``u`` and ``out`` must be different arrays (ping-pong).
    ``next_dt`` (a one-element device tensor, may be ``dt`` itself): ``timestep_advection(part, C, scale)`` for the NEXT step
    is evaluated on the way -- it depends on ``C`` alone --, by extra workgroups of the BC set's own launches
    (``ibh_step_advection_dt``) instead of two launches in front of the next sweep; same value."""
    part = _part(part)
    u, nv, _ = _field(u, part.nc)
    if nv != 1:
        raise ValueError("u must be a scalar field")
    C_, nvc, ldc = _field(C_, part.nc)
    if nvc != part.nd:
        raise ValueError("C must be (nc, nd)")
    if out is None:
        out = torch.empty(part.nc, dtype=torch.float32, device=u.device)
    o, nvo, _ = _field_inplace(out, part.nc, "out")
    if nvo != 1 or o.data_ptr() == u.data_ptr():
        raise ValueError("out must be a scalar field other than u")
    _stream()
    if next_dt is not None:
        call("ibh_step_advection_dt", part.handle, _ptr(u), _ptr(o), _ptr(C_), ldc, _ptr(dt),
             bcs.handle if bcs is not None else c_vp(None), C.c_float(scale), _ptr(next_dt))
        return out
    call("ibh_step_advection", part.handle, _ptr(u), _ptr(o), _ptr(C_), ldc, _ptr(dt),
         bcs.handle if bcs is not None else c_vp(None))
    return out


@_hipaware
def residual_euler_hll(part, P, fluid_R=283.0, fluid_gamma=1.4, out=None, flags=0, fluid=None):
    """Fused Euler residual R2: JST(p) + cell_gradient + MUSCL(high_order) + HLL + green_gauss."""
    part = _part(part)
    if fluid is not None:
        fluid_R, fluid_gamma = fluid.R, fluid.gamma
    P, nv, ldp = _field(P, part.nc)
    if nv != part.nd + 2:
        raise ValueError("P must be (nc, nd+2) = [p T u v (w)]")
    R = out if out is not None else torch.zeros((nv, part.nc), dtype=torch.float32, device=P.device).T
    R, nvr, ldr = _field_inplace(R, part.nc, "out")
    if nvr != nv:
        raise ValueError(f"out must be (nc, {nv})")
    fl = _lib.ibh_fluid(float(fluid_R), float(fluid_gamma), 0.0, 1.0, 0.0, 0)
    _stream()
    call("ibh_residual_euler_hll", part.handle, _ptr(P), ldp, _ptr(R), ldr, C.byref(fl), flags)
    return R


# ---------------------------------------------------------------------------
# partition runtime and ghost-cell BC
# ---------------------------------------------------------------------------
def domain_call(dom, f, args, conv_to_backend, conv_from_backend, kwargs):
    """``(dom::Domain)(f, args...)`` (ImmersedBoundary.jl:820-864) with the GPU backend."""
    if (conv_to_backend is None) != (conv_from_backend is None):
        raise AssertionError("Backend converters must be provided at the same time")
    if conv_to_backend is None:
        raise TypeError("Domain call needs conv_to_backend=ibamd.hip, conv_from_backend=ibamd.to_host: "
                        "the per-partition compute runs on the GPU only (no CPU path)")
    results = []
    for i in dom.partitions:
        part = dom.partitions[i]
        dargs = [np.array(a[part.domain]) for a in args]
        dargs = [conv_to_backend(a) for a in dargs]
        dpart = to_backend(part, conv_to_backend)
        r = f(dpart, *dargs, **kwargs)
        dargs = [conv_from_backend(a) for a in dargs]
        for a, da in zip(args, dargs):
            a[part.image] = da[part.image_in_domain]
        results.append(r)
    return results


def impose_bc(f, dom, bname, *args, conv_to_backend=None, conv_from_backend=None, **kwargs):
    """``impose_bc!(f, dom, bname, args...)`` (ImmersedBoundary.jl:1197-1247).

    ``args`` are global arrays.  Device arrays are updated in place on the GPU; host arrays need
    both converters (the reference's calling convention) and are copied back.
    """
    if (conv_to_backend is None) != (conv_from_backend is None):
        raise AssertionError("Backend converters must be provided at the same time")
    from .hiparray import HipArray

    def on_device(a):
        return isinstance(a, HipArray) or (isinstance(a, torch.Tensor) and a.is_cuda)
    host_args = None
    if not all(on_device(a) for a in args):
        if conv_to_backend is None:
            raise TypeError("impose_bc needs device arrays or conv_to_backend/conv_from_backend (no CPU path)")
        host_args = args
        args = tuple(a if on_device(a) else conv_to_backend(a) for a in args)   # (the converter may return HipArrays)
    wrapped = any(isinstance(a, HipArray) for a in args)
    for a in args:
        if isinstance(a, HipArray):
            a._flush_readers()   # the arrays are written in place: pending broadcasts that read them go first
    args = tuple(a.t if isinstance(a, HipArray) else a for a in args)
    fields = [_field_inplace(a, what="impose_bc argument") for a in args]  # updated in place (:1241-1245)
    for ipart in dom.boundaries[bname]:
        bdry = to_backend(dom.boundaries[bname][ipart])
        iargs = []
        _stream()
        for (a, nv, ld) in fields:
            ia = _like(a, bdry.ng)
            call("ibh_bc_interp", bdry.handle, _ptr(a), nv, ld, _ptr(ia), bdry.ng)
            iargs.append(ia)
        r = f(bdry, *([HipArray(ia) for ia in iargs] if wrapped else iargs), **kwargs)
        if not isinstance(r, tuple):
            r = (r,)
        r = tuple(x.t if isinstance(x, HipArray) else x for x in r)
        for (a, nv, ld), ba, ia in zip(fields, r, iargs):
            if isinstance(ba, torch.Tensor):
                ba, nvb, ldb = _field(ba.expand_as(ia) if ba.shape != ia.shape else ba, bdry.ng)
                call("ibh_bc_blend", bdry.handle, _ptr(a), nv, ld, _ptr(ia), bdry.ng, _ptr(ba), ldb, c_vp(None))
            else:
                const = np.full(nv, ba, dtype=np.float32) if np.isscalar(ba) else np.asarray(ba, dtype=np.float32)
                call("ibh_bc_blend", bdry.handle, _ptr(a), nv, ld, _ptr(ia), bdry.ng, c_vp(None), 0, _hptr(const))
    if host_args is not None:
        for h, (a, _, _) in zip(host_args, fields):
            if not on_device(h):
                h[...] = conv_from_backend(a)


class GraphedClosure:
    """A user closure at operator granularity captured ONCE in a HIP graph and replayed.

    The reference calls ``f(part, args...)`` once per partition and per iteration (ImmersedBoundary.jl:848-850); every
    operator and broadcast of such a closure is a kernel launch here, a few microseconds of GPU work behind tens of
    microseconds of host work each.  The launches of one call are fixed (same partition, same arrays), so they are
    recorded on a capture stream -- every libibhip launch goes to ``ibh_set_stream(current stream)``, temporaries come
    from the graph's private pool -- and ``g()`` replays them as one graph launch: the arrays are read and written in
    place, so new contents of the same arrays are what a replay sees.

    ``GraphedClosure(f, part, u, ud, C)`` runs ``f`` ``warmup`` times eagerly first (library workspaces are allocated on
    first use and nothing may be allocated during capture); the contents of the array arguments are saved before and
    restored after, so constructing it does not change them.  The closure must not read values back to the host
    (``.item()``, ``to_host``): that cannot be captured and raises.
    """

    def __init__(self, f, *args, warmup=2, **kwargs):
        from .hiparray import HipArray
        tens = [a.t if isinstance(a, HipArray) else a for a in args]
        tens = [t for t in tens if isinstance(t, torch.Tensor) and t.is_cuda]
        saved = [t.clone() for t in tens]
        self._keep = (f, args, kwargs)
        self.stream = torch.cuda.Stream()
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            for _ in range(max(1, int(warmup))):
                f(*args, **kwargs)
            for t, s in zip(tens, saved):
                t.copy_(s)
        self.stream.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self.stream):
            self.result = f(*args, **kwargs)
            for a in args:                      # pending broadcasts of the arguments belong to the call
                if isinstance(a, HipArray):
                    a.t
            if isinstance(self.result, HipArray):
                self.result.t
        _stream()

    def __call__(self):
        self.graph.replay()
        return self.result
