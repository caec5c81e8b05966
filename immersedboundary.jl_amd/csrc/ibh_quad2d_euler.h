// Quad sweep of the 2-D Euler residual (R2 of SURVEY.md 8d: P = [p T u v], MUSCL(high_order) with the pressure JST sensor,
// CFD.inviscid_fluxes HLL (cfd.jl:459-508) and Green-Gauss, over ImmersedBoundary.jl:1077-1157): the design of
// ibh_quad2d.h -- one wavefront per 2x2 group of sibling blocks, 4 cells per lane, y neighbours by DPP row shifts,
// x neighbours in registers / ds_bpermute, two sub-face halo slots per lane -- with four variables.  Differences:
//   * the four primitives ARE the transported state (no separate velocity field to gather);
//   * only the pressure feeds the sensor: lateral lines, end cells and the FINE-side correction exist for p alone;
//   * an HLL face flux is not odd / even under a swap of its sides, so low and high edge faces pick (halo, quad cell) or
//     (quad cell, halo) as (owner, neighbour) with selects (noise against ~150 instructions per flux);
//   * LDS holds only the two boundary ROWS of the tile (the bottom / top edge lanes read them); left / right edge lanes
//     take their boundary cells from their own registers, the COARSE pair mate by a quad_perm DPP.
// Arithmetic: blk2::euler_flux_w / euler_side (ibh_sweep2d.h, ibh_block2d.h), Float32 HLL combine like every tuned path.
#pragma once
#include "ibh_quad2d.h"

namespace quad2 {

#pragma clang fp contract(fast)

#define QE_NV 4
#define QE_ROW 20                                     // floats per compact row (16 cells + padding)
#define QE_ROWS (2 * QE_ROW)                          // row 0: y = 0, row 1: y = 15
#define QE_OFF_P 0                                    // [4][2 rows]
#define QE_OFF_SY (QE_OFF_P + QE_NV * QE_ROWS)        // [4][2 rows]
#define QE_OFF_D (QE_OFF_SY + QE_NV * QE_ROWS)        // [2 rows]
#define QE_OFF_EXT (QE_OFF_D + QE_ROWS)               // lateral lines of the pressure [8][20]
#define QE_OFF_RING (QE_OFF_EXT + 160)                // rings: M[4 vars], F, Q  x [4 rows][16]
#define QE_OFF_EX (QE_OFF_RING + 6 * 64)              // edge fluxes [4 vars][4 rows][16]
#define QE_LDS (QE_OFF_EX + QE_NV * 64)

__device__ __forceinline__ float dpp_swap1(float v) {  // lane i <- lane i ^ 1 (quad_perm [1,0,3,2])
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));
}

// ---- two HLL face fluxes at a time: every add / mul / fma below is a packed instruction (v_pk_*_f32: two results per
// issue slot); the reciprocals, square roots, medians and max / min stay one per face.  Same expression trees as
// blk2::euler_flux_w / blk2::euler_side.
__device__ __forceinline__ v2f rcp2(v2f x) { return v2f{__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)}; }
__device__ __forceinline__ v2f sqrt2(v2f x) { return v2f{__builtin_amdgcn_sqrtf(x.x), __builtin_amdgcn_sqrtf(x.y)}; }
__device__ __forceinline__ v2f max2(v2f a, v2f b) { return v2f{fmaxf(a.x, b.x), fmaxf(a.y, b.y)}; }
__device__ __forceinline__ v2f min2(v2f a, v2f b) { return v2f{fminf(a.x, b.x), fminf(a.y, b.y)}; }

// conserved state, pressure, normal velocity and speed of sound of one side of a face pair
__device__ __forceinline__ void euler_state2(const v2f* P, bool dn, const blk2::Gas& gas, v2f* Q, v2f& p, v2f& un, v2f& a) {
    p = P[0];
    const v2f T = max2(P[1], v2f{10.0f, 10.0f});
    const v2f k = 0.5f * (P[2] * P[2] + P[3] * P[3]);
    const v2f rho = p * rcp2(gas.R * T);
    Q[0] = rho;
    Q[1] = rho * (gas.R / (gas.gamma - 1.0f) * T + k);
    Q[2] = rho * P[2];
    Q[3] = rho * P[3];
    un = dn ? P[3] : P[2];
    a = sqrt2((gas.gamma * gas.R) * T);
}

// density, energy per mass, pressure, normal velocity and speed of sound of one side of a face pair
__device__ __forceinline__ void euler_side_pm2(const v2f* P, bool dn, const blk2::Gas& gas, v2f& rho, v2f& e, v2f& p, v2f& un,
                                               v2f& a) {
    p = P[0];
    const v2f T = max2(P[1], v2f{10.0f, 10.0f});
    const v2f q = P[2] * P[2] + P[3] * P[3];
    rho = p * rcp2(gas.R * T);
    e = (gas.R / (gas.gamma - 1.0f)) * T + 0.5f * q;
    un = dn ? P[3] : P[2];
    a = sqrt2((gas.gamma * gas.R) * T);
}

// MUSCL states from undivided slopes, then HLL: F = (SL FL - SR FR + SL SR (QR - QL)) / (SL - SR) with FL = QL unL +
// pressure terms, regrouped by state (as strip3e::euler_flux): F = QL (wL unL - c) + QR (c - wR unR) + pressure terms,
// wL = SL / (SL - SR), wR = SR / (SL - SR), c = SL wR -- two instructions per variable instead of five, and the physical
// fluxes of the two sides are never held
__device__ __forceinline__ void euler_flux_w2(const v2f* Pa, const v2f* Pb, const v2f* Sa, const v2f* Sb, v2f Da, v2f Db,
                                              float wa, bool dn, const blk2::Gas& gas, v2f* F) {
    v2f PL[4], PR[4];
    const v2f Df = max2(max2(Da, Db), v2f{1e-7f, 1e-7f});
    const float wb = 1.0f - wa;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const v2f d = Pb[v] - Pa[v];
        const v2f gu = Sa[v] - d * wa;
        const v2f Du = Sb[v] - d * wb;
        const v2f s = v2f{__builtin_amdgcn_fmed3f(Du.x, gu.x, 0.0f), __builtin_amdgcn_fmed3f(Du.y, gu.y, 0.0f)};
        const v2f t16 = (Sa[v] - Sb[v]) * 0.0625f;
        const v2f uf = (Pa[v] + wa * d) + t16;
        PL[v] = uf + Df * ((s - wa * d) - t16);
        // PR = uf + Df ((wb d - s) - t16) = PL + Df (d - 2 s)   (wa + wb = 1; round 4, as strip3e::euler_flux)
        PR[v] = PL[v] + Df * (d - 2.0f * s);
    }
    // with Q = rho (1, e, u, v) the conserved states are never formed: F = AL (1, eL, uL, vL) + AR (1, eR, uR, vR), A = rho c
    v2f rL, eL, pL, uL, aL, rR, eR, pR, uR, aR;
    euler_side_pm2(PL, dn, gas, rL, eL, pL, uL, aL);
    euler_side_pm2(PR, dn, gas, rR, eR, pR, uR, aR);
    const v2f z = v2f{0.0f, 0.0f};
    const v2f SR = min2(uR - aR, z);
    const v2f SL = max2(uL + aL, z);
    const v2f rs = rcp2(SL - SR);
    const v2f wL = SL * rs, wR = SR * rs;
    const v2f c = SL * wR;
    const v2f AL = rL * (wL * uL - c), AR = rR * (c - wR * uR);
    F[0] = AL + AR;
    F[1] = AL * eL + AR * eR;
    F[2] = AL * PL[2] + AR * PR[2];
    F[3] = AL * PL[3] + AR * PR[3];
    const v2f mL = wL * pL, mR = wR * pR;
    const v2f m = mL - mR;
    F[2] += dn ? z : m;
    F[3] += dn ? m : z;
    F[1] += mL * uL - mR * uR;
}

__device__ __forceinline__ void sweep_quad_euler(const QuadDesc2* __restrict__ qd, const int32_t* __restrict__ qtab,
                                                 int32_t q, const float* __restrict__ P, uint32_t ldp,
                                                 float* __restrict__ Rr, uint32_t ldr, blk2::Gas gas, float* lds,
                                                 int lane) {
    using blk2::ldg;
    using blk2::wave_lds_sync;
    typedef float v2f_g __attribute__((ext_vector_type(2), aligned(4)));
    // ---- lane-only geometry
    const int g = lane >> 4, t = lane & 15, tl = lane & 7, lhs = lane >> 3;
    const bool g0 = g == 0, g3 = g == 3, lr = g0 || g3, dny = !lr, t15 = t == 15, t0 = t == 0;
    const bool high = g >= 2;                                      // top / right: the quad's cell is the owner
    const int delta = g0 ? -1 : g == 1 ? -8 : g == 2 ? 8 : 1;
    const uint32_t a0off = ((lane & 7) << 3) | ((lane & 8) << 4) | ((lane & 16) >> 2) | ((lane & 32) << 1);
    const int rowsel = t0 ? 0 : t15 ? 1 : 2;
    float* const rowP = lds + QE_OFF_P, * const rowSY = lds + QE_OFF_SY, * const rowD = lds + QE_OFF_D;
    const bool yedge = t0 || t15;                                  // only these lanes stage their cells
    const int own_row = (t15 ? QE_ROW : 0) + 4 * g;
    const int b_row = (g == 2 ? QE_ROW : 0) + t;                   // boundary cell of a bottom / top slot lane
    float* const ext = lds + QE_OFF_EXT;
    float* const ring = lds + QE_OFF_RING;                         // kind k at ring + 64 k: M[0..3], F (4), Q (5)
    float* const exf = lds + QE_OFF_EX;
    const int wrow = g == 1 ? 0 : g == 2 ? 1 : 3;
    const int my = rowsel * 16 + 4 * g;                            // ring row this lane reads (its four x positions)
    const int myB = (t0 ? 0 : 2) * 16 + 4 * g, myT = (t15 ? 1 : 2) * 16 + 4 * g;
    ring[4 * 64 + 32 + (lane & 15)] = 0.0f;                        // neutral rows: correction 0, weight 1/2
    ring[5 * 64 + 32 + (lane & 15)] = 0.5f;

    // ---- loads
    const QuadDesc2 d = qd[q];
    const int32_t* row = qtab + (size_t)q * IBH_QROW;
    const v2i hid = *(const v2i*)(row + 2 * lane);
    const uint32_t eid = (uint32_t)row[128 + (lane & 31)];
    const uint32_t a0 = (uint32_t)d.base + a0off;
    v4f U[QE_NV];
#pragma unroll
    for (int v = 0; v < QE_NV; ++v) U[v] = *(const v4f_g*)((const char*)(P + (size_t)v * ldp) + ((size_t)a0 << 2));
    const uint32_t ty = (d.cls >> (4 * lhs)) & 15u;
    const bool isC = ty == SIDE_COARSE, isF = ty == SIDE_FINE;
    const float qs = isC ? (1.0f / 3.0f) : isF ? (2.0f / 3.0f) : 0.5f;
    const float irt = isC ? 0.5f : isF ? 2.0f : 1.0f;
    const float rhx = d.rh[0], rhy = d.rh[1];
    // halo gathers, 8 bytes per lane each (quad_load_halo_paired): left / right lanes (halo, deeper) per slot, bottom /
    // top lanes (slot 0, slot 1) of the halo cells and of the deeper cells
    v2f hu[QE_NV], hd[QE_NV];
    {
        const int lo = delta < 0 ? delta : 0;
        const int iA = lr ? hid.x + lo : hid.x, iB = lr ? hid.y + lo : hid.x + delta;
#pragma unroll
        for (int v = 0; v < QE_NV; ++v) {
            const float* Pv = P + (size_t)v * ldp;
            const v2f PA = *(const v2f_g*)((const char*)Pv + ((size_t)(uint32_t)iA << 2));
            const v2f PB = *(const v2f_g*)((const char*)Pv + ((size_t)(uint32_t)iB << 2));
            const float huA = g0 ? PA.y : PA.x, hdA = g0 ? PA.x : PA.y;
            const float huB = g0 ? PB.y : PB.x, hdB = g0 ? PB.x : PB.y;
            hu[v] = v2f{lr ? huA : PA.x, lr ? huB : (isF ? PA.y : PA.x)};
            hd[v] = v2f{lr ? hdA : PB.x, lr ? hdB : (isF ? PB.y : PB.x)};
        }
    }
    const float eu = ldg(P, eid);                                  // pressure across the ends of the half-sides

    // ---- stage: boundary rows of P, lateral lines of the pressure
#pragma unroll
    for (int v = 0; v < QE_NV; ++v)
        if (yedge) *(v4f*)(rowP + v * QE_ROWS + own_row) = U[v];
    *(v2f*)(ext + lhs * 20 + 2 + 2 * tl) = hu[0];
    {
        const int eline = (lane & 31) >> 2, ee = lane & 3;
        ext[eline * 20 + (ee < 2 ? ee : 16 + ee)] = eu;
    }
    wave_lds_sync();
    // boundary cell of this lane's slots (m0) and its pair mate t ^ 1 (m1, COARSE half-sides)
    float m0[QE_NV], m1[QE_NV];
#pragma unroll
    for (int v = 0; v < QE_NV; ++v) {
        const float in_lane = g3 ? U[v].w : U[v].x;
        const float from_row = lds_read(rowP + v * QE_ROWS + b_row);
        const float mate_row = lds_read(rowP + v * QE_ROWS + (b_row ^ 1));
        const float mate_lane = dpp_swap1(in_lane);
        m0[v] = lr ? in_lane : from_row;
        const float mate = lr ? mate_lane : mate_row;
        m1[v] = isC ? mate : m0[v];
    }
    float hm[QE_NV];
#pragma unroll
    for (int v = 0; v < QE_NV; ++v) {
        hm[v] = 0.5f * (hu[v].x + hu[v].y);
        ring[v * 64 + wrow * 16 + t] = hm[v];
    }
    const float fix = 0.5f * (fabsf(hu[0].x - m0[0]) + fabsf(hu[0].y - m0[0])) - fabsf(hm[0] - m0[0]);
    ring[4 * 64 + wrow * 16 + t] = fix;
    ring[5 * 64 + wrow * 16 + t] = qs;
    wave_lds_sync();

    // ---- own cells: undivided slopes of the four primitives, pressure sensor
    v4f SX[QE_NV], SY[QE_NV], D;
    const v4f qB = *(const v4f*)(ring + 5 * 64 + myB), qT = *(const v4f*)(ring + 5 * 64 + myT);
    const v4f qL = v4f{g0 ? qs : 0.5f, 0.5f, 0.5f, 0.5f}, qR = v4f{0.5f, 0.5f, 0.5f, g3 ? qs : 0.5f};
#pragma unroll
    for (int v = 0; v < QE_NV; ++v) {
        const float uLm = bperm((lane - 16) << 2, U[v].w), uRp = bperm((lane + 16) << 2, U[v].x);
        const v4f UL = v4f{g0 ? hm[v] : uLm, U[v].x, U[v].y, U[v].z};
        const v4f UR = v4f{U[v].y, U[v].z, U[v].w, g3 ? hm[v] : uRp};
        const v4f dR = UR - U[v], dL = U[v] - UL;
        SX[v] = qR * dR + qL * dL;
        const v4f rM = *(const v4f*)(ring + v * 64 + my);
        v4f uB, uT;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uB[c] = dpp_shr1(rM[c], U[v][c]);
            uT[c] = dpp_shl1(rM[c], U[v][c]);
        }
        const v4f dT = uT - U[v], dB = U[v] - uB;
        SY[v] = qT * dT + qB * dB;
        if (v == 0) {
            const v4f gx = dR - dL, gy = dT - dB;
            const v4f rF = *(const v4f*)(ring + 4 * 64 + my);
            const float fx0 = g0 ? fix : 0.0f, fx3 = g3 ? fix : 0.0f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float ax = fabsf(dR[c]) + fabsf(dL[c]);
                if (c == 0) ax += fx0;
                if (c == 3) ax += fx3;
                const float ay = (fabsf(dT[c]) + fabsf(dB[c])) + rF[c];
                D[c] = jst_max2(gx[c], ax, rhx, gy[c], ay, rhy);
            }
        }
        if (yedge) *(v4f*)(rowSY + v * QE_ROWS + own_row) = SY[v];
    }
    if (yedge) *(v4f*)(rowD + own_row) = D;

    // ---- halo cells of this lane's two slots: slopes along the side normal (towards +), pressure sensor
    v2f Sh[QE_NV], Dh;
    {
        const float omq = 1.0f - qs;
#pragma unroll
        for (int v = 0; v < QE_NV; ++v) {
            const v2f dm0 = m0[v] - hu[v], dm1 = m1[v] - hu[v], dde = hd[v] - hu[v];
            const v2f x = omq * (0.5f * (dm0 + dm1)) - 0.5f * dde;   // u_face,in - u_face,deep
            Sh[v] = high ? -x : x;
            if (v == 0) {
                const float ihn = (dny ? rhy : rhx) * irt, iht = (dny ? rhx : rhy) * irt;
                const int Lb = isC ? 2 * (tl & ~1) : 2 * tl, Hb = (isC ? 2 * (tl | 1) : 2 * tl) + 4;
                const v2f Lp = *(const v2f*)(ext + lhs * 20 + Lb);
                const v2f Hp = *(const v2f*)(ext + lhs * 20 + Hb);
                const bool t0l = tl == 0, t7l = tl == 7;
                const v2f lo0 = v2f{(isF && !t0l) ? Lp.y : Lp.x, isF ? hu[0].x : Lp.x};
                const v2f lo1 = v2f{Lp.y, isF ? hu[0].x : Lp.y};
                const v2f hi0 = v2f{isF ? hu[0].y : Hp.x, Hp.x};
                const v2f hi1 = v2f{isF ? hu[0].y : Hp.y, (isF && !t7l) ? Hp.x : Hp.y};
                const v2f e0 = lo0 - hu[0], e1 = lo1 - hu[0], e2 = hi0 - hu[0], e3 = hi1 - hu[0];
                const v2f gn = (dm0 + dm1) + (dde + dde), gt = (e0 + e1) + (e2 + e3);
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const float an = (fabsf(dm0[k]) + fabsf(dm1[k])) + 2.0f * fabsf(dde[k]);
                    const float at = (fabsf(e0[k]) + fabsf(e1[k])) + (fabsf(e2[k]) + fabsf(e3[k]));
                    Dh[k] = jst_max2(gn[k], an, 0.5f * ihn, gt[k], at, 0.5f * iht);
                }
            }
        }
    }

    // ---- edge faces first: both sub-faces of this lane's boundary cell (everything the halo slots hold is dead after
    // this; the residual is then accumulated direction by direction, so neither the x fluxes nor the slopes of a direction
    // outlive it -- 188 -> fewer live registers than holding FR and FT of all cells for a final Green-Gauss pass)
    wave_lds_sync();  // SY, D rows
    float edge[QE_NV];
    {
        float Po[QE_NV], So[QE_NV];
#pragma unroll
        for (int v = 0; v < QE_NV; ++v) {
            const float s_lr = g3 ? SX[v].w : SX[v].x;
            const float s_bt = lds_read(rowSY + v * QE_ROWS + b_row);
            Po[v] = m0[v];
            So[v] = lr ? s_lr : s_bt;
        }
        const float d_lr = g3 ? D.w : D.x, d_bt = lds_read(rowD + b_row);
        const float Do = lr ? d_lr : d_bt;
        const float wa = high ? qs : 1.0f - qs;                    // h_owner / (h_owner + h_neighbour)
        v2f Pa[QE_NV], Pb[QE_NV], Sa[QE_NV], Sb[QE_NV], F[QE_NV];
#pragma unroll
        for (int v = 0; v < QE_NV; ++v) {
            const v2f po = v2f{Po[v], Po[v]}, so = v2f{So[v], So[v]};
            Pa[v] = high ? po : hu[v];
            Pb[v] = high ? hu[v] : po;
            Sa[v] = high ? so : Sh[v];
            Sb[v] = high ? Sh[v] : so;
        }
        const v2f dov = v2f{Do, Do};
        euler_flux_w2(Pa, Pb, Sa, Sb, high ? dov : Dh, high ? Dh : dov, wa, dny, gas, F);
#pragma unroll
        for (int v = 0; v < QE_NV; ++v) edge[v] = 0.5f * (F[v].x + F[v].y);
    }
#pragma unroll
    for (int v = 0; v < QE_NV; ++v) exf[v * 64 + wrow * 16 + t] = edge[v];
    wave_lds_sync();

    // ---- x faces: right face of every cell, two at a time (x = 15: replaced by the edge flux), Green-Gauss in x
    v4f res[QE_NV];
    {
        const int up = (lane + 16) << 2;
        float nP[QE_NV], nS[QE_NV];
#pragma unroll
        for (int v = 0; v < QE_NV; ++v) {
            nP[v] = bperm(up, U[v].x);
            nS[v] = bperm(up, SX[v].x);
        }
        const float nD = bperm(up, D.x);
        v4f FR[QE_NV];
        v2f Pa[QE_NV], Pb[QE_NV], Sa[QE_NV], Sb[QE_NV], F[QE_NV];
#pragma unroll
        for (int v = 0; v < QE_NV; ++v) {
            Pa[v] = U[v].xy;
            Sa[v] = SX[v].xy;
            Pb[v] = U[v].yz;
            Sb[v] = SX[v].yz;
        }
        euler_flux_w2(Pa, Pb, Sa, Sb, D.xy, D.yz, 0.5f, false, gas, F);
#pragma unroll
        for (int v = 0; v < QE_NV; ++v) {
            FR[v].xy = F[v];
            Pa[v] = U[v].zw;
            Sa[v] = SX[v].zw;
            Pb[v] = v2f{U[v].w, nP[v]};
            Sb[v] = v2f{SX[v].w, nS[v]};
        }
        euler_flux_w2(Pa, Pb, Sa, Sb, D.zw, v2f{D.w, nD}, 0.5f, false, gas, F);
#pragma unroll
        for (int v = 0; v < QE_NV; ++v) {
            FR[v].zw = F[v];
            const float FRm = bperm((lane - 16) << 2, FR[v].w);
            const v4f FL = v4f{g0 ? edge[v] : FRm, FR[v].x, FR[v].y, FR[v].z};
            const v4f FRf = v4f{FR[v].x, FR[v].y, FR[v].z, g3 ? edge[v] : FR[v].w};
            res[v] = -((FRf - FL) * rhx);
        }
    }
    // ---- y faces: top face of every cell (y = 15: the edge flux), Green-Gauss in y, store
    {
        v4f FT[QE_NV];
        v2f Pa[QE_NV], Pb[QE_NV], Sa[QE_NV], Sb[QE_NV], F[QE_NV];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            v2f Dbt;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int c = 2 * h + k;
#pragma unroll
                for (int v = 0; v < QE_NV; ++v) {
                    Pa[v][k] = U[v][c];
                    Sa[v][k] = SY[v][c];
                    Pb[v][k] = dpp_shl1(U[v][c], U[v][c]);   // lanes t = 15 keep their own state: a finite, unused flux
                    Sb[v][k] = dpp_shl1(SY[v][c], SY[v][c]);
                }
                Dbt[k] = dpp_shl1(D[c], D[c]);
            }
            euler_flux_w2(Pa, Pb, Sa, Sb, h ? D.zw : D.xy, Dbt, 0.5f, true, gas, F);
#pragma unroll
            for (int v = 0; v < QE_NV; ++v) {
                if (h) FT[v].zw = F[v];
                else FT[v].xy = F[v];
            }
        }
#pragma unroll
        for (int v = 0; v < QE_NV; ++v) {
            const v4f ex = *(const v4f*)(exf + v * 64 + my);
            v4f FB, FTf;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                FB[c] = dpp_shr1(ex[c], FT[v][c]);
                FTf[c] = t15 ? ex[c] : FT[v][c];
            }
            const v4f r = res[v] - ((FTf - FB) * rhy);
            // the residual is not read again by this sweep: stores that do not allocate in L2 -- 14.13 -> 12.78 us per sweep at
            // 0.87 M cells (same box, alternating builds, three times)
#ifdef Q2E_NO_NT_STORE   // (A/B)
            *(v4f_g*)((char*)(Rr + (size_t)v * ldr) + ((size_t)a0 << 2)) = r;
#else
            __builtin_nontemporal_store(r, (v4f_g*)((char*)(Rr + (size_t)v * ldr) + ((size_t)a0 << 2)));
#endif
        }
    }
}

#pragma clang fp contract(off)

}  // namespace quad2
