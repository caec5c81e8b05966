"""Host mirror of the reference's point-implicit module (the orphan file /root/reference/src/point_implicit.jl:
``hutchinson_trick`` :17-91, ``Linearization`` :97-114, ``PIPreconditioner`` :120-161, ``linearize`` :185-209,
``proj_along`` :221-236, ``solve`` :250-329), device-resident: every array stays on the GPU, every array operation
is a kernel of libibhip (``ibh_pi_*``, ``ibh_dot``, ``ibh_maxabs``); the step lengths are computed on the device
and only the convergence test reads a scalar back, once per inner iteration like the reference.

``f`` is the user's residual closure on device arrays, e.g. ``lambda P: ibamd.residual_euler_hll(dpart, P)``.
Arrays are ``(n,)`` or column-major ``(n, nv)`` device arrays (``ibamd.hip``).  There is no CPU path.

Across ranks (one partition per GPU): the arrays are the rank's local arrays (image + skirt rows [+ donor extras]) and
``f`` is ``distributed.RankOps.closure(f_local)`` -- it refreshes the skirt rows of its argument (the halo exchange) before
the residual sweep and returns zeros on the rows the rank does not own -- so that every vector the smoother forms
(``b``, the Hutchinson sums, ``A v``, ``r``, the search directions) vanishes outside the owned rows; what is left to do here
is to sum the dot products and norms and to take the maximum over the ranks: ``reduce=RankOps`` (``.sum(t)`` / ``.max(t)``
all-reduce a small device tensor in place).  The +-1 sample of a skirt row never matters: the exchange overwrites the
perturbed skirt rows with their owners' values.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import backend as B
from ._lib import c_vp, call

EPS32 = float(np.finfo(np.float32).eps)


def _p(t):
    return c_vp(t.data_ptr())


def _dense(t):
    """Device array with n*nv contiguous floats in column-major order (what the ibh_pi_* kernels take)."""
    t, nv, ld = B._field(t)
    n = t.shape[0]
    if t.ndim == 2 and ld != n:
        d = B.colmajor_empty(n, nv)
        d.copy_(t)
        t = d
    return t


def _like(t):
    return B.colmajor_empty(t.shape[0], None if t.ndim == 1 else t.shape[1])


def _numel(t):
    return int(t.shape[0] * (1 if t.ndim == 1 else t.shape[1]))


def rademacher(n, seed):
    """``rand(Int32[-1, 1], n)`` (:36-38): counter-based hash on the device, reproducible for a seed."""
    z = B.colmajor_empty(int(n))
    call("ibh_pi_rademacher", int(n), C.c_uint64(int(seed) & (2 ** 64 - 1)), _p(z))
    return z


def _hutchinson_vector(f, x, n_samples, h, fx, samples, seed):
    """:17-44; ``f`` may return ``(n,)`` or ``(n, nv)``; ``samples`` optionally fixes the +-1 vectors."""
    n = x.shape[0]
    fx = _dense(fx)
    nv = 1 if fx.ndim == 1 else fx.shape[1]
    s = _like(fx)
    s.zero_()
    xbuff = _like(x)
    for i in range(n_samples):
        z = samples[i] if samples is not None else rademacher(n, seed + i)
        call("ibh_pi_perturb", n, _p(x), _p(z), float(h), _p(xbuff))
        fxb = _dense(f(xbuff))
        call("ibh_pi_hutch_accum", n, nv, _p(fxb), _p(fx), _p(z), float(h), _p(s))
    call("ibh_pi_div_scalar", _numel(s), float(n_samples), _p(s))
    return s


def hutchinson_trick(f, x, n_samples, h=1e-6, pre_evaluated_fx=None, samples=None, seed=0):
    """Diagonal (vector ``x``) or per-point diagonal blocks ``(n, nv, nv)`` (matrix ``X``) of the Jacobian of
    ``f`` (:17-91).  ``samples`` (a list of device +-1 vectors; for a matrix ``X`` one list per column) pins the
    random draws; otherwise they come from ``rademacher(n, seed + k)``."""
    x = _dense(x)
    fx = f(x) if pre_evaluated_fx is None else pre_evaluated_fx
    if x.ndim == 1:
        return _hutchinson_vector(f, x, n_samples, h, fx, samples, seed)
    n, nv = x.shape
    xbuff = B.colmajor_empty(n, nv)
    xbuff.copy_(x)
    D = torch.empty((nv, nv, n), dtype=torch.float32, device=x.device).permute(2, 1, 0)  # (n, nv, nv) column-major
    for i in range(nv):
        col = xbuff[:, i]

        def fv(xc, col=col, i=i):
            col.copy_(xc)
            out = f(xbuff)
            col.copy_(x[:, i])
            return out
        Di = _hutchinson_vector(fv, x[:, i].contiguous(), n_samples, h, fx,
                                None if samples is None else samples[i], seed + 7919 * i)
        D[:, :, i].copy_(Di)
    return D


class Linearization:
    """:97-114 -- ``lin(v) = (f(x + v h) - fx) / h``."""

    def __init__(self, f, x, fx, h):
        self.f, self.x, self.fx, self.h = f, _dense(x), _dense(fx), float(h)
        self._buf = _like(self.x)

    def __call__(self, v):
        v = _dense(v)
        call("ibh_pi_perturb", _numel(v), _p(self.x), _p(v), self.h, _p(self._buf))
        fxb = _dense(self.f(self._buf))
        out = _like(fxb)
        call("ibh_pi_fd", _numel(out), _p(fxb), _p(self.fx), self.h, _p(out))
        return out


class PIPreconditioner:
    """:120-161 -- [block-]diagonal preconditioner; ``inverse_diagonal`` is ``(n,)`` or ``(n, nv, nv)``."""

    def __init__(self, inverse_diagonal):
        self.inverse_diagonal = inverse_diagonal

    def __call__(self, v):
        v = _dense(v)
        D = self.inverse_diagonal
        out = _like(v)
        if D.ndim == 1:
            if v.ndim != 1:
                raise TypeError("a diagonal preconditioner applies to vectors")
            call("ibh_pi_apply_blocks", v.shape[0], 1, _p(D), _p(v), _p(out))
        else:
            if v.ndim != 2 or v.shape[1] != D.shape[1]:
                raise TypeError("a block preconditioner applies to (n, nv) arrays")
            call("ibh_pi_apply_blocks", v.shape[0], int(v.shape[1]), _p(D), _p(v), _p(out))
        return out


def _inverse_blocks(D):
    """:124-135, in place."""
    if D.ndim == 1:
        call("ibh_pi_invert_blocks", D.shape[0], 1, _p(D))
    else:
        n, nv = D.shape[0], D.shape[1]
        assert D.stride() == (1, n, n * nv), "block diagonal must be (n, nv, nv) column-major"
        call("ibh_pi_invert_blocks", n, int(nv), _p(D))
    return D


def linearize(f, x, n_hutchinson_samples=30, pre_evaluated_fx=None, h=1e-6, samples=None, seed=0):
    """:185-209 -- returns ``(A, b, D)``: the Linearization, ``b = -f(x)`` and the preconditioner."""
    x0 = _dense(x)
    x = _like(x0)
    x.copy_(x0)
    fx = _dense(f(x) if pre_evaluated_fx is None else pre_evaluated_fx)
    fxc = _like(fx)
    fxc.copy_(fx)
    D = hutchinson_trick(f, x, n_hutchinson_samples, h=h, pre_evaluated_fx=fxc, samples=samples, seed=seed)
    _inverse_blocks(D)
    b = _like(fxc)
    b.copy_(fxc)
    b.neg_()
    return Linearization(f, x, fxc, h), b, PIPreconditioner(D)


class _Scalars:
    """Device scalars of one relaxation step: [Av.b, Av.Av] (double) and max|r| (float)."""

    def __init__(self, dev):
        self.dots = torch.zeros(2, dtype=torch.float64, device=dev)
        self.mx = torch.zeros(1, dtype=torch.float32, device=dev)
        self.nr = torch.zeros(1, dtype=torch.float64, device=dev)


def proj_along(A, v, b, reduce=None):
    """:221-236 -- ``(alpha, Av)`` with ``alpha = (Av . b) / (Av . Av + eps)``; alpha is read back (host float)."""
    Av = A(v)
    b = _dense(b)
    dots = torch.zeros(2, dtype=torch.float64, device=b.device)
    n = _numel(b)
    call("ibh_dot", n, _p(Av), _p(b), c_vp(dots.data_ptr()))
    call("ibh_dot", n, _p(Av), _p(Av), c_vp(dots.data_ptr() + 8))
    if reduce is not None:
        reduce.sum(dots)
    d = dots.cpu().numpy()
    return float(np.float32(d[0]) / (np.float32(d[1]) + np.float32(EPS32))), Av


def _relax(A, s, r, x, sc, reduce=None):
    """One minimal-residual step along ``s`` without leaving the device (:291-294 / :301-304)."""
    As = A(s)
    n = _numel(r)
    call("ibh_dot", n, _p(As), _p(r), c_vp(sc.dots.data_ptr()))
    call("ibh_dot", n, _p(As), _p(As), c_vp(sc.dots.data_ptr() + 8))
    if reduce is not None:
        reduce.sum(sc.dots)                      # (Av . b, Av . Av) over the owned rows of all ranks
    call("ibh_pi_update", n, _p(sc.dots), EPS32, _p(s), _p(As), _p(x), _p(r))


def solve(A, b, prec, n_iter=100, n_inner=1, rtol=1e-2, atol=1e-7, multigrid=None, verbose=False, check_every=1,
          reduce=None):
    """:250-329 -- block-preconditioned two-direction minimal-residual relaxation; returns ``(x, |r|/|r0|)``.
    ``multigrid``: object with ``coarseners`` / ``prolongators`` (callables on device arrays, e.g. the
    ``DeviceAccumulator``s of ``ibamd.multigrid``), cycled from the coarsest level to none like the reference.
    ``check_every``: the residual norm is read back (the one host round trip of a step) every that many inner steps and
    after the last one; 1 = the reference's loop.
    ``reduce``: across ranks (module docstring) -- dot products, norms and ``max |r|`` are all-reduced over the ranks."""
    b = _dense(b)
    n = _numel(b)
    sc = _Scalars(b.device)

    def norm(t):
        call("ibh_sumsq", n, _p(t), _p(sc.nr))
        if reduce is not None:
            reduce.sum(sc.nr)
        return float(np.sqrt(sc.nr.item()))
    nr0 = norm(b)
    nr = nr0
    x = torch.zeros_like(b)
    r = _like(b)
    r.copy_(b)
    n_levels = 0 if multigrid is None else len(multigrid.coarseners)
    n_mgrid = n_levels
    if verbose:
        print("Beginning point-implicit solution\nIteration |r|/|r0|")
    for nit in range(1, n_iter + 1):
        for nin in range(1, n_inner + 1):
            s = prec(r)
            if n_mgrid > 0:
                s = _dense(multigrid.prolongators[n_mgrid - 1](multigrid.coarseners[n_mgrid - 1](s)))
            _relax(A, s, r, x, sc, reduce)
            # second direction: the residual itself
            call("ibh_maxabs", n, _p(r), _p(sc.mx))
            if reduce is not None:
                reduce.max(sc.mx)
            call("ibh_pi_normalize", n, _p(r), _p(sc.mx), EPS32, _p(s))
            _relax(A, s, r, x, sc, reduce)
            step = (nit - 1) * n_inner + nin
            if step % max(1, int(check_every)) == 0 or step == n_iter * n_inner:
                nr = norm(r)
                if verbose:
                    print(f"{step}       {nr / (nr0 + EPS32)}")
                if nr < nr0 * rtol + atol:
                    return x, nr / (nr0 + EPS32)
        n_mgrid = n_levels if n_mgrid == 0 else n_mgrid - 1
    return x, nr / (nr0 + EPS32)
