// Quad sweep (2-D scalar advection-JST-MUSCL residual, test/advection.jl:67-83): ONE wavefront sweeps the 16x16 tile
// of four sibling 8x8 blocks, 4 cells per lane.  Same arithmetic as blk2::sweep_adv (ibh_sweep2d.h) -- undivided
// slopes, flux_w -- but the 64 halo cells of the tile are 1/4 of its cells instead of as many as the block's own,
// and the neighbours of a cell come from registers, DPP row shifts and a few cross-row permutes instead of LDS
// tiles.  tests/quad_model.py is the lane-by-lane numpy statement of this file, checked against the oracle.
//
// Lane L = 16*g + t.
//   own cells:  x = 4*g + r (r = 0..3: one float4 per field), y = t.
//     y neighbours: lane -/+ 1 of the same 16-lane row  -> DPP row_shr:1 / row_shl:1 (lanes t = 0 / 15 keep `old`,
//                   which is preloaded with the bottom / top ring of the tile);
//     x neighbours: r -/+ 1 in registers, across strips lane -/+ 16 -> ds_bpermute (no LDS memory involved).
//   halo slots: the two sub-faces k = 0, 1 of boundary cell t of side [left, bottom, top, right][g] (arithmetic on
//     (k0, k1) pairs); left / right halo values are therefore already in the lanes that own the boundary cells,
//     bottom / top values go through a 16-entry LDS ring to the lanes t = 0 / 15.
//   edge faces: the same lane computes both sub-faces of its boundary cell and hands their mean to the cell's lane.
// A face between cell a (towards -) and cell b (towards +) seen from the other side: F(a,b) = -F(b,a) with slopes
// and velocities negated (every term of flux_w is odd or even under that swap), so low and high sides share one
// evaluation: own cell first, halo cell second, sign = +1 on top / right, -1 on left / bottom.
#pragma once
#include "ibh_sweep2d.h"

namespace quad2 {

#pragma clang fp contract(fast)

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));

// LDS per wave (floats): tile U | tile SY | tile D | tile CY (16 rows, pitch 20: conflict-free ds_write_b128) |
// lateral lines ext[8][20] | rings M, F, Q [4 rows][16] (row 0 bottom, 1 top, 2 neutral, 3 dump) | edge fluxes [4][16]
#define QUAD_PITCH 20
#define QUAD_TILE (16 * QUAD_PITCH)
#define QUAD_OFF_EXT (4 * QUAD_TILE)
#define QUAD_OFF_RING (QUAD_OFF_EXT + 160)
#define QUAD_OFF_EX (QUAD_OFF_RING + 3 * 64)
#define QUAD_LDS (QUAD_OFF_EX + 64)

__device__ __forceinline__ float dpp_shr1(float old, float v) {  // lane i <- lane i-1 in rows of 16; lane 0 keeps old
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x111, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_shl1(float old, float v) {  // lane i <- lane i+1; lane 15 keeps old
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x101, 0xf, 0xf, false));
}
__device__ __forceinline__ float bperm(int byte_addr, float v) {  // value of lane byte_addr/4
    return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_addr, __float_as_int(v)));
}
__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
// an LDS read every lane performs: keeps the compiler from sinking it into a divergent branch of a later select
__device__ __forceinline__ float lds_read(const float* p) {
    float v = *p;
    asm volatile("" : "+v"(v));
    return v;
}

// interior faces (both cells same level, wa = 1/2), four at a time
__device__ __forceinline__ v4f flux_half4(v4f ua, v4f ub, v4f Sa, v4f Sb, v4f Da, v4f Db, v4f Ca, v4f Cb) {
    const v4f d = ub - ua;
    const v4f gu = Sa - 0.5f * d;
    const v4f Du = Sb - 0.5f * d;
    v4f s, Df;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        s[c] = __builtin_amdgcn_fmed3f(Du[c], gu[c], 0.0f);
        Df[c] = max3(Da[c], Db[c], 1e-7f);
    }
    const v4f t16 = (Sa - Sb) * 0.0625f;
    const v4f uf = (ua + 0.5f * d) + t16;
    const v4f A = uf - Df * t16;
    const v4f Cf = 0.5f * (Ca + Cb);
    const v4f B = Df * (s - 0.5f * d);
    const v4f CA = Cf * A;
    v4f F;
#pragma unroll
    for (int c = 0; c < 4; ++c) F[c] = fmaf(fabsf(Cf[c]), B[c], CA[c]);
    return F;
}

// edge faces: two sub-faces at a time, a = the quad's boundary cell (broadcast), b = the halo cells
__device__ __forceinline__ v2f flux_w2(float ua, v2f ub, float Sa, v2f Sb, float Da, v2f Db, float Ca, v2f Cb, float wa) {
    const v2f d = ub - ua;
    const v2f gu = Sa - d * wa;
    const v2f Du = Sb - d * (1.0f - wa);
    const v2f s = v2f{__builtin_amdgcn_fmed3f(Du.x, gu.x, 0.0f), __builtin_amdgcn_fmed3f(Du.y, gu.y, 0.0f)};
    const v2f t16 = (Sa - Sb) * 0.0625f;
    const v2f uf = (ua + wa * d) + t16;
    const v2f mu = d * (0.5f - wa) - t16;
    const v2f Df = v2f{max3(Da, Db.x, 1e-7f), max3(Da, Db.y, 1e-7f)};
    const v2f A = uf + Df * mu;
    const v2f Cf = Ca + wa * (Cb - Ca);
    const v2f B = Df * (s - 0.5f * d);
    const v2f CA = Cf * A;
    return v2f{fmaf(fabsf(Cf.x), B.x, CA.x), fmaf(fabsf(Cf.y), B.y, CA.y)};
}

// JST ratio (1e-7 + |g| rh) / (1e-7 + a rh)  (JST_sensor :1077-1097, undivided sums)
__device__ __forceinline__ float jst(float g, float a, float rh) {
    return fmaf(fabsf(g), rh, 1e-7f) * __builtin_amdgcn_rcpf(fmaf(a, rh, 1e-7f));
}

__device__ __forceinline__ void sweep_quad(const QuadDesc2* __restrict__ qd, const int32_t* __restrict__ qtab, int32_t q,
                                           const float* __restrict__ u, const float* __restrict__ C, uint32_t ldc,
                                           float* __restrict__ ud, float* lds, int lane) {
    using blk2::ldg;
    using blk2::wave_lds_sync;
    float* tU = lds;
    float* tSY = lds + QUAD_TILE;
    float* tD = lds + 2 * QUAD_TILE;
    float* tCY = lds + 3 * QUAD_TILE;
    float* ext = lds + QUAD_OFF_EXT;
    float* ringM = lds + QUAD_OFF_RING;
    float* ringF = ringM + 64;
    float* ringQ = ringM + 128;
    float* exf = lds + QUAD_OFF_EX;
    // ---- lane-only geometry
    const int g = lane >> 4, t = lane & 15, tl = lane & 7;
    const bool g0 = g == 0, g3 = g == 3, lr = g0 || g3, t0 = t == 0, t15 = t == 15;
    const int delta = g0 ? -1 : g == 1 ? -8 : g == 2 ? 8 : 1;          // deeper cell of a halo cell
    const bool dny = g == 1 || g == 2;                                  // side normal to y
    const float sgn = g >= 2 ? 1.0f : -1.0f;                            // +1: the quad's cell is the owner of the edge face
    const int lhs = 2 * g + (t >> 3);                                   // outer half-side of this lane's slots
    const int bx = g0 ? 0 : g3 ? 15 : t, by = lr ? t : g == 1 ? 0 : 15;  // boundary cell of the slots
    const int pos_b = by * QUAD_PITCH + bx;
    const int mate = (t & 1) ? -1 : 1;
    const int pos_b1 = pos_b + mate * (lr ? QUAD_PITCH : 1);           // its pair mate t ^ 1 (COARSE sides)
    const int pos_own = t * QUAD_PITCH + 4 * g;                         // this lane's four cells in a tile
    const int myrow = t0 ? 0 : t15 ? 1 : 2;                             // ring row this lane reads
    const int rowB = t0 ? 0 : 2, rowT = t15 ? 1 : 2;
    const int wrow = g == 1 ? 0 : g == 2 ? 1 : 3;                       // ring row this lane's side writes
    const int eline = (lane & 31) >> 2, ee = lane & 3;
    const int epos = eline * 20 + (ee < 2 ? ee : 16 + ee);
    // neutral ring rows: u ring unused, sensor correction 0, weight 1/2 (read by lanes that are not on the y edges)
    ringF[32 + (lane & 15)] = 0.0f;
    ringQ[32 + (lane & 15)] = 0.5f;

    // ---- loads
    const QuadDesc2 d = qd[q];  // wave-uniform
    const int32_t* row = qtab + (size_t)q * IBH_QROW;
    const v2i hid = *(const v2i*)(row + 2 * lane);
    const uint32_t eid = (uint32_t)row[128 + (lane & 31)];
    const uint32_t a0 = (uint32_t)d.base + 64u * ((g >> 1) + 2 * (t >> 3)) + 4u * (g & 1) + 8u * (t & 7);
    const v4f U = *(const v4f*)((const char*)u + ((size_t)a0 << 2));
    const v4f CX = *(const v4f*)((const char*)C + ((size_t)a0 << 2));
    const v4f CY = *(const v4f*)((const char*)(C + ldc) + ((size_t)a0 << 2));
    const float* Cn = C + (dny ? ldc : 0u);
    const v2f hu = v2f{ldg(u, (uint32_t)hid.x), ldg(u, (uint32_t)hid.y)};
    const v2f hd = v2f{ldg(u, (uint32_t)(hid.x + delta)), ldg(u, (uint32_t)(hid.y + delta))};
    const v2f hc = v2f{ldg(Cn, (uint32_t)hid.x), ldg(Cn, (uint32_t)hid.y)};
    const float eu = ldg(u, eid);
    const uint32_t ty = (d.cls >> (4 * lhs)) & 15u;
    const bool isC = ty == SIDE_COARSE, isF = ty == SIDE_FINE;
    const float qs = isC ? (1.0f / 3.0f) : isF ? (2.0f / 3.0f) : 0.5f;  // at_faces weight of the quad's cell
    const float irt = isC ? 0.5f : isF ? 2.0f : 1.0f;                   // h / h_halo
    const float rhx = d.rh[0], rhy = d.rh[1];

    // ---- stage: u tile, lateral lines
    *(v4f*)(tU + pos_own) = U;
    *(v4f*)(tCY + pos_own) = CY;
    *(v2f*)(ext + lhs * 20 + 2 + 2 * tl) = hu;
    ext[epos] = eu;
    wave_lds_sync();
    const float m0 = lds_read(tU + pos_b);
    const float m1r = lds_read(tU + pos_b1);
    const float m1 = isC ? m1r : m0;
    const float hm = 0.5f * (hu.x + hu.y);
    // sensor correction of a boundary cell that faces two finer cells: its |d| sum counts both of them
    const float fix = 0.5f * (fabsf(hu.x - m0) + fabsf(hu.y - m0)) - fabsf(hm - m0);
    ringM[wrow * 16 + t] = hm;
    ringF[wrow * 16 + t] = fix;
    ringQ[wrow * 16 + t] = qs;
    wave_lds_sync();

    // ---- own cells: undivided slopes + sensor
    v4f SX, SY, D;
    v4f uT;
    {
        const float uLm = bperm((lane - 16) << 2, U.w), uRp = bperm((lane + 16) << 2, U.x);
        const v4f UL = v4f{g0 ? hm : uLm, U.x, U.y, U.z};
        const v4f UR = v4f{U.y, U.z, U.w, g3 ? hm : uRp};
        const v4f qL = v4f{g0 ? qs : 0.5f, 0.5f, 0.5f, 0.5f};
        const v4f qR = v4f{0.5f, 0.5f, 0.5f, g3 ? qs : 0.5f};
        const v4f dR = UR - U, dL = U - UL;
        SX = qR * dR + qL * dL;
        const v4f gx = dR - dL;
        const v4f rM = *(const v4f*)(ringM + myrow * 16 + 4 * g);
        const v4f rF = *(const v4f*)(ringF + myrow * 16 + 4 * g);
        const v4f qB = *(const v4f*)(ringQ + rowB * 16 + 4 * g);
        const v4f qT = *(const v4f*)(ringQ + rowT * 16 + 4 * g);
        v4f uB;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uB[c] = dpp_shr1(rM[c], U[c]);
            uT[c] = dpp_shl1(rM[c], U[c]);
        }
        const v4f dT = uT - U, dB = U - uB;
        SY = qT * dT + qB * dB;
        const v4f gy = dT - dB;
        const float fx0 = g0 ? fix : 0.0f, fx3 = g3 ? fix : 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float ax = fabsf(dR[c]) + fabsf(dL[c]);
            if (c == 0) ax += fx0;
            if (c == 3) ax += fx3;
            const float ay = (fabsf(dT[c]) + fabsf(dB[c])) + rF[c];
            D[c] = max3(jst(gx[c], ax, rhx), jst(gy[c], ay, rhy), 1e-7f);
        }
    }
    *(v4f*)(tSY + pos_own) = SY;
    *(v4f*)(tD + pos_own) = D;

    // ---- halo cells of this lane's two slots: slope along the side normal (towards +) and sensor
    v2f Shn, Dh;  // Shn = MINUS the slope along the axis on low sides, the slope itself on high sides ... see flux below
    {
        const float ihn = (dny ? rhy : rhx) * irt, iht = (dny ? rhx : rhy) * irt;
        const v2f dm0 = m0 - hu, dm1 = m1 - hu, dde = hd - hu;
        const v2f din = 0.5f * (dm0 + dm1);
        // slope of the halo cell seen from the quad outwards: -(1 - qs) din + dde / 2  (= -x of blk2::sweep_adv)
        Shn = 0.5f * dde - (1.0f - qs) * din;
        // lateral neighbours (ibh_sweep2d.h:246-255 in pair form; quad_model.py checks the equivalence)
        const int Lb = isC ? 2 * (tl & ~1) : 2 * tl;
        const int Hb = (isC ? 2 * (tl | 1) : 2 * tl) + 4;
        const v2f Lp = *(const v2f*)(ext + lhs * 20 + Lb);
        const v2f Hp = *(const v2f*)(ext + lhs * 20 + Hb);
        const bool t0l = tl == 0, t7l = tl == 7;
        const v2f lo0 = v2f{(isF && !t0l) ? Lp.y : Lp.x, isF ? hu.x : Lp.x};
        const v2f lo1 = v2f{Lp.y, isF ? hu.x : Lp.y};
        const v2f hi0 = v2f{isF ? hu.y : Hp.x, Hp.x};
        const v2f hi1 = v2f{isF ? hu.y : Hp.y, (isF && !t7l) ? Hp.x : Hp.y};
        const v2f e0 = lo0 - hu, e1 = lo1 - hu, e2 = hi0 - hu, e3 = hi1 - hu;
        const v2f gn = (dm0 + dm1) + (dde + dde);
        const v2f gt = (e0 + e1) + (e2 + e3);
        const float hn = 0.5f * ihn, ht = 0.5f * iht;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float an = (fabsf(dm0[k]) + fabsf(dm1[k])) + 2.0f * fabsf(dde[k]);
            const float at = (fabsf(e0[k]) + fabsf(e1[k])) + (fabsf(e2[k]) + fabsf(e3[k]));
            Dh[k] = max3(jst(gn[k], an, hn), jst(gt[k], at, ht), 1e-7f);
        }
    }

    // ---- interior faces: right (x+) and top (y+) face of every cell
    v4f FR, FT;
    {
        const int up = (lane + 16) << 2;
        const v4f Ub = v4f{U.y, U.z, U.w, bperm(up, U.x)};
        const v4f Sb = v4f{SX.y, SX.z, SX.w, bperm(up, SX.x)};
        const v4f Db = v4f{D.y, D.z, D.w, bperm(up, D.x)};
        const v4f Cb = v4f{CX.y, CX.z, CX.w, bperm(up, CX.x)};
        FR = flux_half4(U, Ub, SX, Sb, D, Db, CX, Cb);
        v4f St, Dt, Ct;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            St[c] = dpp_shl1(SY[c], SY[c]);
            Dt[c] = dpp_shl1(D[c], D[c]);
            Ct[c] = dpp_shl1(CY[c], CY[c]);
        }
        FT = flux_half4(U, uT, SY, St, D, Dt, CY, Ct);
    }

    // ---- edge faces: both sub-faces of this lane's boundary cell
    wave_lds_sync();  // SY, D tiles
    float edge;
    {
        const float So_lr = g3 ? SX.w : SX.x, Do_lr = g3 ? D.w : D.x, Co_lr = g3 ? CX.w : CX.x;
        const float So_bt = lds_read(tSY + pos_b), Do_bt = lds_read(tD + pos_b), Co_bt = lds_read(tCY + pos_b);
        const float So = lr ? So_lr : So_bt;
        const float Do = lr ? Do_lr : Do_bt;
        const float Co = lr ? Co_lr : Co_bt;
        // own cell first, halo second; on low sides everything that is odd under the swap is negated (sgn = -1)
        const v2f F = flux_w2(m0, hu, sgn * So, Shn, Do, Dh, sgn * Co, sgn * hc, qs);
        edge = sgn * (0.5f * (F.x + F.y));
    }
    exf[wrow * 16 + t] = edge;
    wave_lds_sync();

    // ---- Green-Gauss
    {
        const float FRm = bperm((lane - 16) << 2, FR.w);
        const v4f FL = v4f{g0 ? edge : FRm, FR.x, FR.y, FR.z};
        const v4f FRf = v4f{FR.x, FR.y, FR.z, g3 ? edge : FR.w};
        const v4f ex = *(const v4f*)(exf + myrow * 16 + 4 * g);
        v4f FB, FTf;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            FB[c] = dpp_shr1(ex[c], FT[c]);
            FTf[c] = t15 ? ex[c] : FT[c];
        }
        const v4f res = -((FRf - FL) * rhx) - ((FTf - FB) * rhy);
        *(v4f*)((char*)ud + ((size_t)a0 << 2)) = res;
    }
}

#pragma clang fp contract(off)

}  // namespace quad2
