"""The closure of /root/reference/test/advection.jl:47-83 VERBATIM -- same expression tree, the broadcast arithmetic
between the operators going through the elementwise entry points of the C ABI (``ibh_ew_*``) exactly as
``Base.Broadcast`` on a ``HipArray`` does in julia/IBHip.jl -- against the oracle's numpy statement of the same lines.
Plus the elementwise kernels on their own."""
import numpy as np
import pytest

import ibamd
from conftest import rel_inf, seeded_field
from oracle import domain as od

pytestmark = pytest.mark.gpu
f32 = np.float32
KW = dict(conv_to_backend=lambda a: ibamd.HipArray(a), conv_from_backend=lambda a: ibamd.to_host(a))


def test_elementwise_kernels():
    rng = np.random.default_rng(0)
    a = rng.uniform(-2, 2, (1000, 3)).astype(f32)
    b = rng.uniform(0.5, 2, (1000, 3)).astype(f32)
    v = rng.uniform(0.5, 2, 1000).astype(f32)
    A, Bm, V = ibamd.HipArray(a), ibamd.HipArray(b), ibamd.HipArray(v)
    assert np.array_equal((A + Bm).to_host(), a + b)
    assert np.array_equal((A - Bm).to_host(), a - b)
    assert np.array_equal((A * Bm).to_host(), a * b)
    assert np.allclose((A / Bm).to_host(), a / b, rtol=1e-6)       # IEEE division on both sides
    assert np.array_equal((A * V).to_host(), a * v[:, None])        # column vector over the columns
    assert np.array_equal((V * A).to_host(), a * v[:, None])
    assert np.array_equal((A / 2).to_host(), a / f32(2))
    assert np.array_equal((2 - A).to_host(), f32(2) - a)
    assert np.array_equal(abs(A).to_host(), np.abs(a))
    assert np.array_equal((-A).to_host(), -a)
    assert np.array_equal(A.maximum_with(Bm).to_host(), np.maximum(a, b))
    assert A.maximum() == a.max() and A.minimum() == a.min()
    assert abs(A.sum() - a.sum(dtype=np.float64)) <= 1e-3
    c = A.copy()
    c -= Bm
    c += 1
    assert np.array_equal(c.to_host(), a - b + f32(1))
    assert np.array_equal(A.col(2).to_host(), a[:, 1])
    z = A.similar().fill(0)
    assert not z.to_host().any()
    with pytest.raises(TypeError):
        A[0]
    with pytest.raises(ValueError):
        A + ibamd.HipArray(np.zeros((999, 3), f32))


def test_advection_closure_verbatim(adv_domains):
    dp, do = adv_domains
    n = len(dp)
    X = dp.global_centers()
    u0 = seeded_field(X, kind="step")
    Cx, Cy = np.ones(n, f32), np.ones(n, f32)
    C = np.stack([Cx, Cy], axis=1)

    # ---- oracle: advection.jl:52-59 and :67-83 in numpy
    def o_dt(part, cx, cy):
        return f32(0.5) / np.max(np.maximum(od.unsigned_green_gauss(part, od.at_faces(part, cx, 1), 1),
                                            od.unsigned_green_gauss(part, od.at_faces(part, cy, 2), 2)))

    def o_closure(part, u, ud, Cl):
        D = od.JST_sensor(part, u)
        for dim in (1, 2):
            Cf = od.at_faces(part, np.ascontiguousarray(Cl[:, dim - 1]), dim)
            gu = od.cell_gradient(part, u, dim)
            uL, uR = od.MUSCL(part, u, gu, dim, D=D, high_order=True)
            ud -= od.green_gauss(part, (uL + uR) * Cf / f32(2) + np.abs(Cf) * (uL - uR) / f32(2), dim)

    # ---- the same lines against the binding: every array below is a HipArray, every operator between them an ibh_ew_*
    def timestep_length(part, cx, cy):
        return 0.5 / ibamd.unsigned_green_gauss(part, ibamd.at_faces(part, cx, 1), 1).maximum_with(
            ibamd.unsigned_green_gauss(part, ibamd.at_faces(part, cy, 2), 2)).maximum()

    def closure(part, u, ud, C):
        D = ibamd.JST_sensor(part, u)
        for dim in range(1, part.ndims + 1):
            Cd = C.col(dim)                                   # Cd = @view C[:, dim]
            Cf = ibamd.at_faces(part, Cd, dim)
            gu = ibamd.cell_gradient(part, u, dim)
            uL, uR = ibamd.MUSCL(part, u, gu, dim, D=D, high_order=True)
            ud -= ibamd.green_gauss(part, (uL + uR) * Cf / 2 + abs(Cf) * (uL - uR) / 2, dim)   # ud .-= green_gauss(...)
        assert all(isinstance(x, ibamd.HipArray) for x in (D, Cf, gu, uL, uR, ud))

    dt_o = min(do(o_dt, Cx, Cy)) * f32(0.75)
    dt_g = min(dp(timestep_length, Cx, Cy, **KW)) * f32(0.75)
    assert abs(dt_o - dt_g) <= 1e-6 * dt_o
    uo, ug = u0.copy(), u0.copy()
    for _ in range(2):
        udo, udg = np.zeros(n, f32), np.zeros(n, f32)
        do(o_closure, uo, udo, C)
        dp(closure, ug, udg, C, **KW)
        assert rel_inf(udg, udo) <= 1e-5
        uo += udo * dt_o
        ug += udg * dt_o
        # impose_bc! with the closures of advection.jl:33-46 on device-resident arrays
        od.impose_bc(lambda b, ui: f32(1.0), do, "upper", uo)
        od.impose_bc(lambda b, ui: f32(0.0), do, "lower", uo)
        od.impose_bc(lambda b, ui: ui.copy(), do, "outlet", uo)
        dev = ibamd.HipArray(ug)
        ibamd.impose_bc(lambda b, ui: 1.0, dp, "upper", dev)
        ibamd.impose_bc(lambda b, ui: 0.0, dp, "lower", dev)
        ibamd.impose_bc(lambda b, ui: ui.copy(), dp, "outlet", dev)
        ug = dev.to_host()
        assert rel_inf(ug, uo) <= 1e-5


def test_fused_broadcast_equals_node_by_node():
    """A broadcast expression runs as ONE launch (ibh_ew_eval, Julia's broadcast fusion); evaluated node by node
    (HipArray.fuse = False: ibh_ew_binary / ibh_ew_unary) it gives the same bits.  In-place forms, column-vector
    operands, scalars on either side, an expression too large for one program, and a write to an operand while an
    expression that reads it is still pending."""
    rng = np.random.default_rng(0)
    n = 10007
    A, B, Cc = (rng.uniform(-2, 2, (n, 3)).astype(f32) for _ in range(3))
    v = rng.uniform(0.5, 2, n).astype(f32)

    def run():
        a, b, c, w = ibamd.HipArray(A), ibamd.HipArray(B), ibamd.HipArray(Cc), ibamd.HipArray(v)
        r1 = (a + b) * c / 2 + abs(c) * (a - b) / 2               # advection.jl:76-80
        r2 = (3.0 - a * w) / (abs(b) + 1.5) - (-c).maximum_with(w).minimum_with(2.0)
        r3 = (w * w + 1.0).sqrt()
        big = a
        for k in range(30):                                      # 60 nodes: more than one program holds
            big = big * 0.99 + b * (0.01 * (k + 1))
        acc = ibamd.HipArray(A.copy())
        acc -= r1 * 0.25
        acc += w
        acc *= 0.5
        pend = a + 1.0                                           # pending, reads a ...
        a.fill(7.0)                                              # ... a is overwritten: pend must hold the old values
        return [x.to_host() for x in (r1, r2, r3, big, acc, pend)]

    ibamd.HipArray.fuse = True
    fused = run()
    ibamd.HipArray.fuse = False
    try:
        nodes = run()
    finally:
        ibamd.HipArray.fuse = True
    for f, g in zip(fused, nodes):
        assert np.array_equal(f, g)
    ref1 = (A + B) * Cc / f32(2) + np.abs(Cc) * (A - B) / f32(2)
    assert np.array_equal(fused[0], ref1.astype(f32))
    assert np.array_equal(fused[5], A + f32(1.0))


def test_impose_bc_host_arrays_through_the_hiparray_converter(adv_domains):
    """impose_bc!(f, dom, name, host_u; conv_to_backend = HipArray, conv_from_backend) -- the reference's calling
    convention with host arrays (ImmersedBoundary.jl:1206-1212): converted first, the closure sees HipArrays, the host
    array is updated; against the oracle."""
    dp, do = adv_domains
    X = dp.global_centers()
    uo = seeded_field(X, kind="step")
    ug = uo.copy()
    od.impose_bc(lambda b, ui: f32(1.0), do, "upper", uo)
    od.impose_bc(lambda b, ui: ui * f32(0.5) + f32(0.25), do, "outlet", uo)
    seen = []

    def half(b, ui):
        seen.append(type(ui))
        return ui * 0.5 + 0.25
    ibamd.impose_bc(lambda b, ui: 1.0, dp, "upper", ug, **KW)
    ibamd.impose_bc(half, dp, "outlet", ug, **KW)
    assert seen and all(t is ibamd.HipArray for t in seen)
    assert rel_inf(ug, uo) <= 1e-6


def test_in_place_broadcast_that_reads_a_column_of_its_destination():
    """`P ./= P[:, 1]` (an operand aliases the destination with another shape: Julia's broadcast_unalias copies) and a
    pending expression whose operand an operator overwrites through out=..."""
    rng = np.random.default_rng(3)
    A = rng.uniform(0.5, 2, (5000, 3)).astype(f32)
    P = ibamd.HipArray(A.copy())
    P /= P.col(1)
    assert np.allclose(P.to_host(), A / A[:, :1], rtol=1e-6)
    Q = ibamd.HipArray(A.copy())
    Q *= Q.col(3)
    assert np.array_equal(Q.to_host(), A * A[:, 2:3])


def test_graphed_closure_replays_the_operator_closure(adv_domains):
    """ibamd.GraphedClosure: the closure of advection.jl:67-83 at operator granularity, captured once in a HIP graph;
    a replay is bit-identical to the eager call, reads the CURRENT contents of its arrays, and building it leaves the
    arrays as they were."""
    dp, _ = adv_domains
    part = dp.partitions[1]
    dpart = ibamd.to_backend(part, ibamd.hip)
    X = part.centers
    rng = np.random.default_rng(5)
    u = ibamd.HipArray(seeded_field(X, kind="smooth"))
    C = ibamd.HipArray(np.ones((X.shape[0], 2), f32))
    ud = ibamd.HipArray(np.zeros(X.shape[0], f32))

    def closure(part, u, ud, C):
        D = ibamd.JST_sensor(part, u)
        for dim in range(1, part.ndims + 1):
            Cf = ibamd.at_faces(part, C.col(dim), dim)
            gu = ibamd.cell_gradient(part, u, dim)
            uL, uR = ibamd.MUSCL(part, u, gu, dim, D=D, high_order=True)
            ud -= ibamd.green_gauss(part, (uL + uR) * Cf / 2 + abs(Cf) * (uL - uR) / 2, dim)

    u0 = u.to_host()
    g = ibamd.GraphedClosure(closure, dpart, u, ud, C)
    assert np.array_equal(u.to_host(), u0) and not ud.to_host().any()     # construction left the arrays alone
    for trial in range(2):
        un = (u0 + f32(0.1) * rng.standard_normal(u0.shape).astype(f32)) if trial else u0
        u.t.copy_(ibamd.hip(un))
        ud.t.zero_()
        closure(dpart, u, ud, C)
        eager = ud.to_host()
        ud.t.zero_()
        g()
        assert np.array_equal(ud.to_host(), eager)
        assert np.abs(eager).max() > 0


@pytest.mark.parametrize("n,nv", [(1003, 1), (1003, 5), (4096, 3), (3, 1), (257, 4)])
def test_vector_interpreter_equals_the_one_element_interpreter(n, nv):
    """``ibh_ew_eval`` runs a flat program four elements per thread (``k_ew_eval4``, 16-byte loads, the tail one by one);
    operands that are column vectors broadcast over the result, or not 16-byte aligned, take the one-element interpreter.
    All three against the one-element interpreter forced by ``ibh_set_tuning("ew_scalar", 1)``: bit for bit, odd sizes."""
    import torch
    from ibamd import _lib
    H = ibamd.HipArray
    rng = np.random.default_rng(n + nv)
    shp = (n,) if nv == 1 else (n, nv)
    a, b, c = (ibamd.hip(rng.uniform(0.5, 2.0, shp).astype(np.float32)) for _ in range(3))
    col = ibamd.hip(rng.uniform(0.5, 2.0, n).astype(np.float32))
    big = ibamd.hip(rng.uniform(0.5, 2.0, (n + 1, 1)).astype(np.float32))
    off = big[1:, 0]                                    # a view 4 bytes off a 16-byte boundary

    def exprs():
        out = [((H(a) - H(b)) / 1e-3 - H(c) * H(a)).t,
               (abs(H(a) - H(b)).maximum_with(H(c) * 0.25) + (H(a) / H(c)).sqrt()).t]
        if nv > 1:
            out.append((H(a) * H(col) - H(b)).t)        # column vector broadcast over the columns
        else:
            out.append((H(a) * H(off) - H(b)).t)        # misaligned operand
        return out
    fast = exprs()
    _lib.call("ibh_set_tuning", b"ew_scalar", 1)
    try:
        slow = exprs()
    finally:
        _lib.call("ibh_set_tuning", b"ew_scalar", 0)
    for f, s in zip(fast, slow):
        assert torch.equal(f, s)
    ref = (ibamd.to_host(a) - ibamd.to_host(b)) / np.float32(1e-3) - ibamd.to_host(c) * ibamd.to_host(a)
    assert np.array_equal(ibamd.to_host(fast[0]), ref.astype(np.float32))
