"""Device-resident ``FAS!`` (mirror of /root/reference/src/solver.jl:39-91).

Same signature and semantics as the reference (including its quirks: the recursion guard is
``length(coarseners) > 1`` so the last supplied level is never visited, and the coarse problem is solved
before the fine smoothing).  ``Q``, the residuals and the sources are device arrays that never leave the
GPU; coarseners / prolongators are device Accumulators (``to_backend(acc)``: one SpMV kernel each); the
fixed-point update and the norm are libibhip kernels.  The only host round-trip per iteration is the
scalar norm the reference's convergence test needs (solver.jl:84-87).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import backend as B

_eps32 = float(np.finfo(np.float32).eps)


def _norm(r):
    """||r||_2 of a device array (sum of squares accumulated in Float64 on the device)."""
    r, nv, ld = B._field(r)
    flat = r if r.ndim == 1 or ld == r.shape[0] else r.T.contiguous().T
    out = torch.zeros(1, dtype=torch.float64, device=r.device)
    B._stream()
    B.call("ibh_sumsq", int(flat.numel()), B._ptr(flat), B._ptr(out))
    return float(out.item()) ** 0.5


def _update(Q, omega, r):
    """Q += clamp(omega, 0, 1) * r  (solver.jl:82)."""
    if isinstance(omega, torch.Tensor):
        Q += torch.clamp(omega, 0.0, 1.0) * r
        return
    Qf, _, ldq = B._field(Q)
    rf, _, ldr = B._field(r)
    if Qf.data_ptr() != Q.data_ptr() or (Q.ndim == 2 and (ldq != Q.shape[0] or ldr != r.shape[0])):
        Q += min(max(float(omega), 0.0), 1.0) * r
        return
    B._stream()
    B.call("ibh_axpy_clamped", int(Q.numel()), C.c_float(float(omega)), B._ptr(rf), B._ptr(Q))


def _update_norm(Q, omega, r):
    """``_update`` and ``_norm(r)`` in one pass over ``r`` (``ibh_axpy_clamped_sumsq``); None if the arrays do not allow it."""
    return _fas_pass(r, None, Q, omega, True)


def _dense_pair(a, b):
    """True when two device arrays are column-major without padding and of one shape (one flat pass covers both)."""
    if a.shape != b.shape:
        return False
    for t in (a, b):
        f, _, ld = B._field(t)
        if f.data_ptr() != t.data_ptr() or (t.ndim == 2 and ld != t.shape[0]):
            return False
    return True


def _fas_pass(r, source, Q, omega, want_norm):
    """``rr = r [+ source]; [Q += clamp(omega, 0, 1) * rr]; [||rr||]`` in ONE launch (``ibh_fas_update``: solver.jl:80-84 on
    one read of ``r``).  ``source`` and ``Q`` may be None.  Returns the norm (or True when none was asked for), or None when
    the arrays do not allow the flat pass (padded layouts, a per-cell ``omega``): the caller composes the steps then."""
    if isinstance(omega, torch.Tensor):
        return None
    if (source is not None and not _dense_pair(r, source)) or (Q is not None and not _dense_pair(r, Q)):
        return None
    f, _, ld = B._field(r)
    if f.data_ptr() != r.data_ptr() or (r.ndim == 2 and ld != r.shape[0]):
        return None
    out = torch.empty(1, dtype=torch.float64, device=r.device) if want_norm else None
    B._stream()
    B.call("ibh_fas_update", int(r.numel()), C.c_float(float(omega) if omega is not None else 0.0), B._ptr(r),
           B._ptr(source), B._ptr(Q), B._ptr(out))
    return float(out.item()) ** 0.5 if want_norm else True


def _sub(a, b):
    """``a .- b`` of two device arrays as a libibhip broadcast (``ibh_ew_*``), not an ATen kernel."""
    from .hiparray import HipArray
    return (HipArray(a) - HipArray(b)).t


def _add(a, b):
    from .hiparray import HipArray
    return (HipArray(a) + HipArray(b)).t


def FAS(f, Q, coarseners=(), prolongators=(), perscribed_f=None, multigrid_level=0, n_iter=50,
        rtol=1e-1, atol=1e-7, norm=None, check_every=1, exchange=None, level_norm=None):
    """``FAS!(f, Q; coarseners, prolongators, perscribed_f, multigrid_level, n_iter, rtol, atol)``.

    ``f(level, Q) -> (r, omega)`` with device arrays; ``Q`` is updated in place.  Returns the residual-norm
    reduction ratio like the reference.  ``norm``: the norm of a residual array; on a rank of a multi-GPU run pass
    ``distributed.Reductions(...).norm`` (sum over the owned cells of all ranks, one all-reduce), default = local.
    ``check_every``: the convergence test -- the one host round trip (and all-reduce) of an iteration -- is made every
    that many iterations and after the last one; 1 = the reference's loop, larger values may run up to
    ``check_every - 1`` smoothing steps past the reference's exit.
    On a rank of a multi-GPU run (``distributed.RankLevels``: the rank's partition of every level, local transfer
    operators): ``exchange(level, Q)`` refreshes the skirt (and donor) rows of a level's local array -- called before every
    residual evaluation and around the coarse solve, where the prolongation reads coarse skirt rows -- and
    ``level_norm(level, r)`` is the norm over the owned cells of all ranks (``Reductions.norm`` of that level).
    """
    local_norm = level_norm is None and norm is None      # the library's own norm: fused with the update below
    if level_norm is not None:
        def _norm(r, _l=multigrid_level):
            return level_norm(_l, r)
    else:
        _norm = norm if norm is not None else globals()["_norm"]
    l = multigrid_level
    xch = exchange if exchange is not None else (lambda _l, _q: None)
    xch(l, Q)
    fQ, omega = f(l, Q)
    source = None
    if perscribed_f is not None:
        source = _sub(perscribed_f, fQ)
    nr0 = _fas_pass(fQ, source, None, None, True) if local_norm else None    # ||fQ + source|| on one read of fQ
    if nr0 is None:
        nr0 = _norm(fQ if source is None else _add(fQ, source))
    nr = nr0
    if len(coarseners) > 1:
        coars, prolong = B.to_backend(coarseners[0]), B.to_backend(prolongators[0])
        Qc = coars(Q)
        xch(l + 1, Qc)                       # (the prolongation below reads skirt rows of Qc - Qcold)
        Qcold = B._like(Qc, Qc.shape[0])
        Qcold.copy_(Qc)
        pfQc = coars(fQ if source is None else _add(fQ, source))
        FAS(f, Qc, coarseners=coarseners[1:], prolongators=prolongators[1:], perscribed_f=pfQc,
            multigrid_level=multigrid_level + 1, n_iter=n_iter, atol=atol, rtol=rtol, norm=norm,
            check_every=check_every, exchange=exchange, level_norm=level_norm)
        xch(l + 1, Qc)
        prolong.diff_add(Q, Qc, Qcold)       # Q .+= prolong(Qc .- Qcold), one launch
    check_every = max(1, int(check_every))
    for it in range(n_iter):
        xch(l, Q)
        r, omega = f(l, Q)
        check = (it + 1) % check_every == 0 or it == n_iter - 1
        # r .+= source; Q .+= clamp(omega, 0, 1) .* r; norm(r): one launch where the layouts allow (and the norm is local)
        fused = _fas_pass(r, source, Q, omega, check and local_norm)
        if fused is None:
            if source is not None:
                r = _add(r, source)
            _update(Q, omega, r)
            if check:
                fused = _norm(r)
        elif check and not local_norm:
            fused = _norm(r if source is None else _add(r, source))
        if check:
            nr = fused
            if nr < nr0 * rtol + atol:
                break
    return nr / (nr0 + _eps32)
