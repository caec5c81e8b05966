"""ORACLE (test infrastructure only; never imported by the product path).

CPU restatement of ``Solver.FAS!`` (/root/reference/src/solver.jl:39-91):
full-approximation-scheme multigrid driver around a user residual
``f(level, Q) -> (r, omega)``.  Quirks kept literally: the recursion guard is
``len(coarseners) > 1`` (the last supplied level is never visited) and the
coarse problem is solved before the fine smoothing (coarse-first sawtooth).
"""
import numpy as np

f32 = np.float32


def _norm(r):
    return np.sqrt(np.sum(r.astype(np.float64) ** 2)).astype(r.dtype)


def FAS(f, Q, coarseners=(), prolongators=(), perscribed_f=None, multigrid_level=0,
        n_iter=50, rtol=f32(1e-1), atol=f32(1e-7), norm=_norm, exchange=None, level_norm=None):
    """In place on ``Q``; returns the residual-norm reduction ratio.  ``exchange(level, Q)`` / ``level_norm(level, r)``:
    the hooks of a rank of a distributed run (the same places as in the product's solver.FAS); absent in the reference."""
    l = multigrid_level
    if level_norm is not None:
        def norm(r, _l=multigrid_level):
            return level_norm(_l, r)
    xch = exchange if exchange is not None else (lambda _l, _q: None)
    xch(l, Q)
    fQ, omega = f(l, Q)
    source = f32(0.0)
    if perscribed_f is not None:
        source = perscribed_f - fQ
    r = fQ + source
    nr0 = norm(r)
    nr = nr0
    if len(coarseners) > 1:
        coars, prolong = coarseners[0], prolongators[0]
        Qc = coars(Q)
        xch(l + 1, Qc)
        Qcold = Qc.copy()
        pfQc = coars(r)
        FAS(f, Qc, coarseners=coarseners[1:], prolongators=prolongators[1:], perscribed_f=pfQc,
            multigrid_level=multigrid_level + 1, n_iter=n_iter, atol=atol, rtol=rtol, norm=norm,
            exchange=exchange, level_norm=level_norm)
        xch(l + 1, Qc)
        Q += prolong(Qc - Qcold)
    for _ in range(n_iter):
        xch(l, Q)
        r, omega = f(l, Q)
        r = r + source
        Q += np.clip(omega, f32(0.0), f32(1.0)) * r
        nr = norm(r)
        if nr < nr0 * rtol + atol:
            break
    return nr / (nr0 + np.finfo(f32).eps)
