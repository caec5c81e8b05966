"""ORACLE (test infrastructure only; never imported by the product path).

CPU restatement of ``NNInterpolator`` (/root/reference/src/nninterp.jl:16-183).

Third-party arithmetic: the reference uses NearestNeighbors.jl 0.4.21
(``knn``) and LAPACK ``pinv``; neither is under /root/reference and no
reference test pins their tie-breaks / rank cut-offs -> PARITY UNPINNED for kNN
donor choice on exact ties and for ``pinv`` on degenerate donor sets.  Here
kNN = scipy cKDTree (exact), donors ordered by (distance, index).
"""
import numpy as np
from scipy.spatial import cKDTree

from .accumulator import Accumulator


def knn_sorted(tree, Xq, k):
    """k nearest neighbours of each column of ``Xq (nd, n)``; rows ordered by (distance, index)."""
    d, idx = tree.query(np.ascontiguousarray(Xq.T, dtype=np.float64), k=k)
    if k == 1:
        d, idx = d[:, None], idx[:, None]
    order = np.lexsort((idx, d), axis=1)
    return np.take_along_axis(idx, order, axis=1)


def linear_weights(X, indices, x):
    """nninterp.jl:16-42: weighted least-squares linear interpolation weights."""
    Tf = X.dtype.type
    eps_ = np.finfo(X.dtype).eps
    dX = X[:, indices] - x[:, None]
    s = dX[0] * dX[0]
    for k in range(1, dX.shape[0]):
        s = s + dX[k] * dX[k]
    distances = np.sqrt(s) + eps_
    w = Tf(1.0) / distances
    A = np.concatenate([dX.T, np.ones((dX.shape[1], 1), dtype=X.dtype)], axis=1)
    Aw = A * w[:, None]
    rtol = eps_ * min(Aw.shape)
    w = np.linalg.pinv(Aw, rcond=rtol)[-1, :] * w
    mask = np.abs(w) > eps_
    return w[mask], indices[mask]


def IDW_weights(X, indices, x):
    """nninterp.jl:47-69: inverse-distance weights."""
    Tf = X.dtype.type
    eps_ = np.finfo(X.dtype).eps
    dX = X[:, indices] - x[:, None]
    s = dX[0] * dX[0]
    for k in range(1, dX.shape[0]):
        s = s + dX[k] * dX[k]
    distances = np.sqrt(s) + eps_
    w = Tf(1.0) / distances
    tot = w[0]
    for k in range(1, w.shape[0]):
        tot = tot + w[k]
    w = w / tot
    mask = np.abs(w) > np.sqrt(eps_)
    return w[mask], indices[mask]


def Interpolator(X, Xc, tree=None, bias=None, first_index=False, linear=True, k=0):
    """nninterp.jl:86-138.  Returns an ``Accumulator`` (0-based donor ids)."""
    if first_index:
        X = np.ascontiguousarray(X.T)
        Xc = np.ascontiguousarray(Xc.T)
        if bias is not None:
            bias = np.ascontiguousarray(bias.T)
    if k == 0:
        k = 2 ** X.shape[0]
    if tree is None:
        tree = cKDTree(np.ascontiguousarray(X.T, dtype=np.float64))
    Xq = Xc if bias is None else Xc + bias
    nbrs = knn_sorted(tree, Xq, k)
    get_w = linear_weights if linear else IDW_weights
    idxs, ws = [], []
    for j in range(Xc.shape[1]):
        w, ii = get_w(X, nbrs[j], Xc[:, j])
        idxs.append(ii)
        ws.append(w)
    return Accumulator(idxs, ws, first_index=first_index)


def domain(*intps):
    """nninterp.jl:147-168: sorted unique donor ids + old->new map."""
    s = set()
    for intp in intps:
        for _, st, _ in intp.stencils.values():
            s.update(int(i) for i in st.ravel())
    idxs = np.array(sorted(s), dtype=np.int64)
    return idxs, {int(k): i for i, k in enumerate(idxs)}


def re_index(intp, hmap):
    """nninterp.jl:175-183."""
    for l, (rows, st, ws) in list(intp.stencils.items()):
        new = np.vectorize(lambda i: hmap[int(i)], otypes=[np.int64])(st) if st.size else st
        intp.stencils[l] = (rows, new, ws)
