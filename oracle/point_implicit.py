"""ORACLE (test infrastructure only; never imported by the product path).

numpy restatement of the reference's point-implicit module, the orphan file
/root/reference/src/point_implicit.jl (not included by src/ImmersedBoundary.jl; no reference test exercises it, so
parity is pinned only by the analytic properties checked in tests/test_point_implicit.py: exact block recovery for
pointwise-linear functions, pinv against LAPACK, convergence of the relaxation -- "parity unpinned" otherwise).
Float32 throughout, same operation order as the Julia broadcasts.  The +-1 samples of ``rand(Int32[-1,1], n)``
(:36-38) are an argument here: the Julia RNG stream cannot be reproduced.
"""
import numpy as np

f32 = np.float32
EPS = np.finfo(np.float32).eps


def _hutch_vec(f, x, samples, h, fx):
    """:17-44"""
    s = np.zeros_like(fx)
    for z in samples:
        xb = x + z * f32(h)
        J = (f(xb) - fx) / f32(h)
        s = s + (z.reshape((-1,) + (1,) * (fx.ndim - 1)) * J)
    return s / f32(len(samples))


def hutchinson_trick(f, x, samples, h=1e-6, pre_evaluated_fx=None):
    """:17-91.  ``samples``: list of +-1 vectors (vector x) or one such list per column (matrix X)."""
    fx = f(x) if pre_evaluated_fx is None else pre_evaluated_fx
    if x.ndim == 1:
        return _hutch_vec(f, x, samples, h, fx)
    xb = x.copy()
    cols = []
    for i in range(x.shape[1]):
        def fv(xc, i=i):
            xb[:, i] = xc
            out = f(xb)
            xb[:, i] = x[:, i]
            return out
        cols.append(_hutch_vec(fv, x[:, i].copy(), samples[i], h, fx))
    return np.stack(cols, axis=-1)  # (n, nv, nv): [p, k, i] = d f_k / d x_i


def inverse_blocks(D):
    """:124-135 (pinv tolerance of LinearAlgebra.pinv: eps * min(size) * sigma_max)."""
    if D.ndim == 1:
        return f32(1.0) / (EPS + D)
    out = np.empty_like(D)
    for p in range(D.shape[0]):
        out[p] = np.linalg.pinv(D[p], rcond=float(EPS) * D.shape[1])
    return out


def apply_prec(invD, v):
    """:141-161"""
    if invD.ndim == 1:
        return v * invD
    return (v[:, None, :] * invD).sum(axis=2, dtype=f32)


class Linearization:
    """:97-114"""

    def __init__(self, f, x, fx, h):
        self.f, self.x, self.fx, self.h = f, x, fx, f32(h)

    def __call__(self, v):
        return (self.f(self.x + v * self.h) - self.fx) / self.h


def linearize(f, x, samples, pre_evaluated_fx=None, h=1e-6):
    """:185-209"""
    fx = f(x) if pre_evaluated_fx is None else pre_evaluated_fx.copy()
    x = x.copy()
    D = inverse_blocks(hutchinson_trick(f, x, samples, h=h, pre_evaluated_fx=fx))
    return Linearization(f, x, fx, h), -fx, D


def _dots(Av, b, reduce):
    """(Av . b, Av . Av) in Float64 like the device kernels; across ranks (``reduce``: tests of the distributed smoother,
    ibamd.distributed.RankOps) the two partial sums are all-reduced."""
    d = np.array([np.vdot(Av.astype(np.float64), b.astype(np.float64)), np.vdot(Av.astype(np.float64), Av.astype(np.float64))])
    if reduce is not None:
        reduce.sum(d)
    return d


def proj_along(A, v, b, reduce=None):
    """:221-236"""
    Av = A(v)
    if reduce is None:
        return f32(np.vdot(Av, b)) / (f32(np.vdot(Av, Av)) + EPS), Av
    d = _dots(Av, b, reduce)
    return f32(d[0]) / (f32(d[1]) + EPS), Av


def solve(A, b, invD, n_iter=100, n_inner=1, rtol=1e-2, atol=1e-7, multigrid=None, reduce=None):
    """:250-329.  ``reduce``: the hooks of a rank of a distributed run (sums and maxima all-reduced, Float64 partial sums);
    None = the reference's loop."""
    if reduce is not None:
        return _solve_ranks(A, b, invD, n_iter, n_inner, rtol, atol, reduce)
    nr0 = f32(np.linalg.norm(b))
    nr = nr0
    x = np.zeros_like(b)
    r = b.copy()
    n_levels = 0 if multigrid is None else len(multigrid.coarseners)
    n_mgrid = n_levels
    for _ in range(n_iter):
        for _ in range(n_inner):
            s = apply_prec(invD, r)
            if n_mgrid > 0:
                s = multigrid.prolongators[n_mgrid - 1](multigrid.coarseners[n_mgrid - 1](s))
            a, As = proj_along(A, s, r)
            x = x + s * a
            r = r - As * a
            s = r / (EPS + np.max(np.abs(r)))
            a, As = proj_along(A, s, r)
            x = x + s * a
            r = r - As * a
            nr = f32(np.linalg.norm(r))
            if nr < nr0 * f32(rtol) + f32(atol):
                return x, nr / (nr0 + EPS)
        n_mgrid = n_levels if n_mgrid == 0 else n_mgrid - 1
    return x, nr / (nr0 + EPS)


def _solve_ranks(A, b, invD, n_iter, n_inner, rtol, atol, reduce):
    """``solve`` with the scalars of a step all-reduced over the ranks (vectors vanish outside the owned rows)."""
    def norm(t):
        d = np.array([np.sum(t.astype(np.float64) ** 2)])
        reduce.sum(d)
        return f32(np.sqrt(d[0]))

    def maxabs(t):
        d = np.array([np.max(np.abs(t))], dtype=np.float32)
        reduce.max(d)
        return d[0]
    nr0 = norm(b)
    nr = nr0
    x = np.zeros_like(b)
    r = b.copy()
    for _ in range(n_iter):
        for _ in range(n_inner):
            s = apply_prec(invD, r)
            a, As = proj_along(A, s, r, reduce)
            x = x + s * a
            r = r - As * a
            s = r / (EPS + maxabs(r))
            a, As = proj_along(A, s, r, reduce)
            x = x + s * a
            r = r - As * a
            nr = norm(r)
            if nr < nr0 * f32(rtol) + f32(atol):
                return x, nr / (nr0 + EPS)
    return x, nr / (nr0 + EPS)
