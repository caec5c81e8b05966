"""Lane-level numpy model of the quad sweep (csrc/ibh_quad2d.h): the executable specification of that kernel.

Test infrastructure.  One "wavefront" sweeps a 16x16 tile of four sibling 8x8 blocks with 4 cells per lane; every
array below is shaped (nq, 64[, ...]) = (quad, lane) and every data movement of the kernel (in-lane neighbours,
DPP row shifts, cross-row permutes, the LDS tables) has its counterpart here with the same lane maps, so that the
maps, the class logic of the halo cells and the edge-flux bookkeeping can be checked against the oracle on a CPU.

Lane L = 16*g + t.  Own cells of the lane: x = 4*g + r (r = 0..3), y = t.  The same lane owns the two sub-face
halo slots (k = 0, 1) of boundary cell t of side [left, bottom, top, right][g].
Arithmetic: float32 throughout, same formulas as blk2::flux_w / sensor2 (ibh_sweep2d.h).
"""
import numpy as np

f32 = np.float32
SAME, MIRROR, COARSE, FINE = 0, 1, 2, 3
L = np.arange(64)
G = L >> 4           # lane row: x strip of the own cells / side of the halo slots
T = L & 15           # y of the own cells / boundary cell of the halo slots
DELTA = np.array([-1, -8, 8, 1])[G]   # deeper cell of a halo cell: one step away from the quad
DN = np.array([0, 1, 1, 0])[G]        # normal dim of the lane's side
HIGH = np.array([0, 0, 1, 1])[G]      # top / right: the quad's cell is the owner of the face


def med3(a, b, c):
    return np.maximum(np.minimum(a, b), np.minimum(np.maximum(a, b), c))


def flux_w(ua, ub, Sa, Sb, Da, Db, Ca, Cb, wa):
    """blk2::flux_w: MUSCL(high_order) states blended with the sensor, upwind flux (ibh_sweep2d.h:39-53)."""
    ua, ub, Sa, Sb, Da, Db, Ca, Cb, wa = [np.asarray(v, dtype=f32) for v in (ua, ub, Sa, Sb, Da, Db, Ca, Cb, wa)]
    d = ub - ua
    gu = Sa - d * wa
    Du = Sb - d * (f32(1) - wa)
    s = med3(Du, gu, f32(0))
    t16 = (Sa - Sb) * f32(0.0625)
    uf = (ua + wa * d) + t16
    mu = d * (f32(0.5) - wa) - t16
    Df = np.maximum(np.maximum(Da, Db), f32(1e-7))
    A = uf + Df * mu
    Cf = Ca + wa * (Cb - Ca)
    B = Df * (s - f32(0.5) * d)
    return (Cf * A + np.abs(Cf) * B).astype(f32)


def quad_sweep(desc, tab, u, C):
    """ud on the cells of the quads: returns (cell ids (nq, 64, 4), ud (nq, 64, 4))."""
    eps = f32(1e-7)
    half = f32(0.5)
    nq = desc.shape[0]
    base = desc["base"].astype(np.int64)[:, None]
    rhx = desc["rh"][:, 0][:, None].astype(f32)
    rhy = desc["rh"][:, 1][:, None].astype(f32)
    cx_all, cy_all = C[:, 0], C[:, 1]
    # ---- loads
    a0 = base + 64 * ((G >> 1) + 2 * (T >> 3)) + 4 * (G & 1) + 8 * (T & 7)          # (nq, 64)
    cells = a0[:, :, None] + np.arange(4)
    U = u[cells]
    CX = cx_all[cells]
    CY = cy_all[cells]
    hid = tab[:, :128].reshape(nq, 64, 2).astype(np.int64)
    hu = u[hid]
    hdeep = u[hid + DELTA[None, :, None]]
    hc = np.where(DN[None, :, None] == 1, cy_all[hid], cx_all[hid])
    eu = u[tab[:, 128:160].astype(np.int64)]                                          # (nq, 32)
    l_of = 2 * G + (T >> 3)                                                          # half-side of the lane
    ty = (desc["cls"][:, None] >> (4 * l_of)[None, :].astype(np.uint32)) & 15
    assert not np.any((ty == MIRROR) | (ty > FINE))
    isC, isF = ty == COARSE, ty == FINE
    qs = np.where(isC, f32(1.0) / f32(3.0), np.where(isF, f32(2.0) / f32(3.0), half)).astype(f32)
    # ---- "LDS": the u tile, the lateral lines of the halo cells
    tile = np.empty((nq, 16, 16), dtype=f32)                                         # [y][x]
    for r in range(4):
        tile[:, T, 4 * G + r] = U[:, :, r]
    ext = np.empty((nq, 8, 20), dtype=f32)
    for k in range(2):
        ext[:, l_of, 2 + 2 * (T & 7) + k] = hu[:, :, k]
    E = np.arange(32)
    ext[:, E >> 2, np.where((E & 3) < 2, E & 3, 16 + (E & 3))] = eu
    hm = half * (hu[:, :, 0] + hu[:, :, 1])
    # boundary cell of the lane's halo slots, and its pair mate on a COARSE side
    bx = np.where(G == 0, 0, np.where(G == 3, 15, T))
    by = np.where((G == 0) | (G == 3), T, np.where(G == 1, 0, 15))
    bx1 = np.where((G == 0) | (G == 3), bx, T ^ 1)
    by1 = np.where((G == 0) | (G == 3), T ^ 1, by)
    m0 = tile[:, by, bx]
    m1 = np.where(isC, tile[:, by1, bx1], m0)
    # correction of the sensor's |d| sum of an own boundary cell facing two finer cells (0 unless FINE)
    fix = half * (np.abs(hu[:, :, 0] - m0) + np.abs(hu[:, :, 1] - m0)) - np.abs(hm - m0)
    # ---- own cells: undivided slopes and sensor
    lane_m16 = np.clip(L - 16, 0, 63)
    lane_p16 = np.clip(L + 16, 0, 63)
    lane_m1 = np.clip(L - 1, 0, 63)
    lane_p1 = np.clip(L + 1, 0, 63)
    uL = np.empty_like(U)
    uR = np.empty_like(U)
    uL[:, :, 1:] = U[:, :, :3]
    uR[:, :, :3] = U[:, :, 1:]
    uL[:, :, 0] = np.where(G == 0, hm, U[:, lane_m16, 3])        # ds_bpermute from lane - 16
    uR[:, :, 3] = np.where(G == 3, hm, U[:, lane_p16, 0])
    ringM = np.zeros((nq, 2, 16), dtype=f32)                    # bottom / top ring of u (mean of the sub-faces)
    ringF = np.zeros((nq, 2, 16), dtype=f32)
    ringQ = np.zeros((nq, 2, 16), dtype=f32)
    for side, g in ((0, 1), (1, 2)):
        ringM[:, side, :] = hm[:, 16 * g:16 * g + 16]
        ringF[:, side, :] = fix[:, 16 * g:16 * g + 16]
        ringQ[:, side, :] = qs[:, 16 * g:16 * g + 16]
    xs = 4 * G[:, None] + np.arange(4)[None, :]                  # (64, 4) x of the own cells
    uB = np.where((T == 0)[None, :, None], ringM[:, 0][:, xs], U[:, lane_m1, :])   # row_shr:1, old = ring
    uT = np.where((T == 15)[None, :, None], ringM[:, 1][:, xs], U[:, lane_p1, :])  # row_shl:1
    qL = np.full(U.shape, half, dtype=f32)
    qR = np.full(U.shape, half, dtype=f32)
    qL[:, :, 0] = np.where(G == 0, qs, half)
    qR[:, :, 3] = np.where(G == 3, qs, half)
    qB = np.where((T == 0)[None, :, None], ringQ[:, 0][:, xs], half).astype(f32)
    qT = np.where((T == 15)[None, :, None], ringQ[:, 1][:, xs], half).astype(f32)
    fx = np.zeros(U.shape, dtype=f32)
    fx[:, :, 0] += np.where(G == 0, fix, f32(0))
    fx[:, :, 3] += np.where(G == 3, fix, f32(0))
    fy = (np.where((T == 0)[None, :, None], ringF[:, 0][:, xs], f32(0)) +
          np.where((T == 15)[None, :, None], ringF[:, 1][:, xs], f32(0))).astype(f32)

    def slope_sensor(un, up, qn, qp, fixv, rh):
        dR, dL = up - U, U - un
        S = qp * dR + qn * dL
        g = dR - dL
        a = (np.abs(dR) + np.abs(dL)) + fixv
        nu = (eps + np.abs(g) * rh[:, :, None]) / (eps + a * rh[:, :, None])
        return S.astype(f32), nu.astype(f32)
    SX, nux = slope_sensor(uL, uR, qL, qR, fx, rhx)
    SY, nuy = slope_sensor(uB, uT, qB, qT, fy, rhy)
    D = np.maximum(np.maximum(nux, nuy), eps)
    # ---- halo cells (two slots per lane)
    irt = np.where(isC, half, np.where(isF, f32(2), f32(1))).astype(f32)
    rh_n = np.where(DN[None, :] == 1, rhy, rhx)
    rh_t = np.where(DN[None, :] == 1, rhx, rhy)
    ihn, iht = (rh_n * irt)[:, :, None], (rh_t * irt)[:, :, None]
    din = (half * (m0 + m1))[:, :, None] - hu
    dde = hdeep - hu
    xh = (f32(1) - qs)[:, :, None] * din - half * dde
    Sh = np.where(HIGH[None, :, None] == 1, -xh, xh).astype(f32)
    tl = T & 7
    q_idx = np.arange(nq)[:, None]
    Lb = np.where(isC, 2 * (tl & ~1)[None, :], 2 * tl[None, :])           # ext index of the low lateral pair
    Hb = np.where(isC, 2 * (tl | 1)[None, :] + 4, 2 * tl[None, :] + 4)
    lrow = l_of[None, :]
    L0, L1 = ext[q_idx, lrow, Lb], ext[q_idx, lrow, Lb + 1]
    H0, H1 = ext[q_idx, lrow, Hb], ext[q_idx, lrow, Hb + 1]
    t0, t7 = (tl == 0)[None, :], (tl == 7)[None, :]
    lat = np.empty((nq, 64, 2, 4), dtype=f32)                             # [k][lo0, lo1, hi0, hi1]
    lat[:, :, 0, 0] = np.where(isF & ~t0, L1, L0)
    lat[:, :, 0, 1] = L1
    lat[:, :, 0, 2] = np.where(isF, hu[:, :, 1], H0)
    lat[:, :, 0, 3] = np.where(isF, hu[:, :, 1], H1)
    lat[:, :, 1, 0] = np.where(isF, hu[:, :, 0], L0)
    lat[:, :, 1, 1] = np.where(isF, hu[:, :, 0], L1)
    lat[:, :, 1, 2] = H0
    lat[:, :, 1, 3] = np.where(isF & ~t7, H0, H1)
    # cross-check against the index formulas of blk2::sweep_adv (ibh_sweep2d.h:246-255)
    for k in range(2):
        p = 2 * tl[None, :] + k
        mask = np.where(isF, 15, np.where(isC, 12, 14))
        pm, w = p & mask, 16 - mask
        e0i = pm + 2
        single = mask == 15
        lo0 = np.where(pm == 0, 0, e0i - w)
        lo1 = lo0 + ((pm == 0) | ~single)
        hi0 = e0i + w
        hi1 = hi0 + ((hi0 == 18) | ~single)
        for j, ix in enumerate((lo0, lo1, hi0, hi1)):
            assert np.array_equal(ext[q_idx, lrow, ix], lat[:, :, k, j]), (k, j)

    def sensor4(c, d4, ih):
        d = d4 - c[..., None]
        g = (d[..., 0] + d[..., 1]) + (d[..., 2] + d[..., 3])
        a = (np.abs(d[..., 0]) + np.abs(d[..., 1])) + (np.abs(d[..., 2]) + np.abs(d[..., 3]))
        hih = half * ih
        return ((eps + np.abs(g) * hih) / (eps + a * hih)).astype(f32)
    nrm = np.stack([np.broadcast_to(m0[:, :, None], hu.shape), np.broadcast_to(m1[:, :, None], hu.shape), hdeep, hdeep],
                   axis=-1)
    Dh = np.maximum(np.maximum(sensor4(hu, nrm, ihn), sensor4(hu, lat, iht)), eps)
    # ---- interior faces
    hw = f32(0.5)
    Ub = np.concatenate([U[:, :, 1:], U[:, lane_p16, 0:1]], axis=2)
    Sb = np.concatenate([SX[:, :, 1:], SX[:, lane_p16, 0:1]], axis=2)
    Db = np.concatenate([D[:, :, 1:], D[:, lane_p16, 0:1]], axis=2)
    Cb = np.concatenate([CX[:, :, 1:], CX[:, lane_p16, 0:1]], axis=2)
    FR = flux_w(U, Ub, SX, Sb, D, Db, CX, Cb, hw)                          # right face of every cell (x = 15: unused)
    FT = flux_w(U, uT, SY, SY[:, lane_p1, :], D, D[:, lane_p1, :], CY, CY[:, lane_p1, :], hw)  # y = 15: unused
    # ---- edge faces: lane (side g, boundary cell t), both sub-faces
    rsel = np.where(G == 3, 3, 0)
    tileSy = np.empty((nq, 16, 16), dtype=f32)
    tileD = np.empty((nq, 16, 16), dtype=f32)
    tileCy = np.empty((nq, 16, 16), dtype=f32)
    for r in range(4):
        tileSy[:, T, 4 * G + r] = SY[:, :, r]
        tileD[:, T, 4 * G + r] = D[:, :, r]
        tileCy[:, T, 4 * G + r] = CY[:, :, r]
    lr = ((G == 0) | (G == 3))[None, :]
    pick = lambda A: A[:, L, rsel]                                         # in-lane boundary cell of left / right
    uo = np.where(lr, pick(U), tile[:, by, bx])
    So = np.where(lr, pick(SX), tileSy[:, by, bx])
    Do = np.where(lr, pick(D), tileD[:, by, bx])
    Co = np.where(lr, pick(CX), tileCy[:, by, bx])
    e = lambda A: A[:, :, None]
    F_low = flux_w(hu, e(uo), Sh, e(So), Dh, e(Do), hc, e(Co), e(f32(1) - qs))
    F_high = flux_w(e(uo), hu, e(So), Sh, e(Do), Dh, e(Co), hc, e(qs))
    Fk = np.where(HIGH[None, :, None] == 1, F_high, F_low)
    edgeF = (half * (Fk[:, :, 0] + Fk[:, :, 1])).astype(f32)
    # the same through the orientation symmetry the kernel uses: F_low(a, b) = -F(b, a) with S, C negated
    F_sym = -flux_w(e(uo), hu, -e(So), -Sh, e(Do), Dh, -e(Co), -hc, e(qs))
    scale = np.abs(Fk).max() + f32(1e-30)
    assert np.abs(np.where(HIGH[None, :, None] == 1, F_high, F_sym) - Fk).max() <= 2e-6 * scale
    # ---- Green-Gauss
    FL = np.empty_like(FR)
    FL[:, :, 1:] = FR[:, :, :3]
    FL[:, :, 0] = np.where(G == 0, edgeF, FR[:, lane_m16, 3])
    FRr = FR.copy()
    FRr[:, :, 3] = np.where(G == 3, edgeF, FR[:, :, 3])
    edgeB = edgeF[:, 16:32][:, xs]                                         # bottom edge flux of cell x
    edgeT = edgeF[:, 32:48][:, xs]
    FB = np.where((T == 0)[None, :, None], edgeB, FT[:, lane_m1, :])
    FTt = np.where((T == 15)[None, :, None], edgeT, FT)
    ud = -((FRr - FL) * rhx[:, :, None]) - ((FTt - FB) * rhy[:, :, None])
    return cells, ud.astype(f32)
