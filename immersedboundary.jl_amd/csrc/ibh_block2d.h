// Block fast path, 2-D, 8x8 blocks: one 64-lane wavefront per block, lane = cell (x fastest).
//
// This is the tuned form of the fast path (the literal IEEE form lives in ibh_fused.hip and is
// selected with IBH_EXACT).  Differences, all within the 1e-5 norm-wise parity tolerance:
//   * the divisions of the reference formulas are replaced by per-block reciprocals (1/hx, 1/hy
//     are wave-uniform, stored in the block descriptor) and per-side weights: inside a block and
//     across same-level / 2:1 sides the spacing ratio h_n/h_o is 1, 2 or 1/2, so
//     at_faces = (1-q)*u_o + q*u_n with q = 1/(1+ratio) in {1/2, 1/3, 2/3} and 1/(d_o+d_n) = 2*q/h;
//     the sensor ratio uses v_rcp_f32;
//   * every interior face flux is computed once, by the lane on its left/bottom, and handed to the
//     neighbour lane with a wavefront shuffle; block-boundary faces that no lane owns (left and
//     bottom sides, second sub-faces) are computed together in one extra pass;
//   * FMA contraction is allowed in this file.
// The body is BRANCH-FREE (profiles/r1_v2: the first version spent ~100 scalar instructions per wave
// on exec-mask bookkeeping and was bound by instruction issue):
//   * the halo cell of every slot comes from a precomputed per-block table (one coalesced load);
//   * every block side has TWO sub-face slots per boundary cell; single-face sides repeat sub-face 0
//     in slot 1, so "mean over the faces of the side" is always (f0 + f1)/2, exact when f1 == f0;
//   * interior-vs-halo neighbours are one LDS read with a selected address.
// LDS per wave and field: tile[64] followed by halo[64] (slot = (side*8 + t)*2 + k).
#pragma once
#include "ibh_common.h"

namespace blk2 {

#pragma clang fp contract(fast)

__device__ __forceinline__ float ldg(const float* __restrict__ p, uint32_t i) {
    return *(const float*)((const char*)p + (size_t)(i << 2));
}
__device__ __forceinline__ void stg(float* __restrict__ p, uint32_t i, float v) {
    *(float*)((char*)p + (size_t)(i << 2)) = v;
}

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float sel4(int s, float a0, float a1, float a2, float a3) {
    const float lo = (s & 1) ? a1 : a0;
    const float hi = (s & 1) ? a3 : a2;
    return (s & 2) ? hi : lo;
}

struct Lane {
    int i, j;
    bool edge[4];
    bool general;
    int nidx[4];  // index of the sub-face-0 neighbour in a [tile | halo] LDS field: tile = 0..63, halo = 64..127
    float q[4];   // 1/(1 + h_nb/h) towards direction s
    float rt[4];  // h_nb/h
};

__device__ __forceinline__ Lane lane_info(const BlockDesc2& b, int lane) {
    Lane L;
    L.i = lane & 7;
    L.j = lane >> 3;
    L.edge[0] = L.i == 0;
    L.edge[1] = L.i == 7;
    L.edge[2] = L.j == 0;
    L.edge[3] = L.j == 7;
    L.general = false;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        L.general |= L.edge[s] && b.type[s] == SIDE_GENERAL;
        const int off = (s == 0) ? -1 : (s == 1) ? 1 : (s == 2) ? -8 : 8;
        const int t = (s < 2) ? L.j : L.i;
        L.nidx[s] = L.edge[s] ? 64 + (s * 8 + t) * 2 : lane + off;
        L.q[s] = L.edge[s] ? b.q[s] : 0.5f;
        L.rt[s] = L.edge[s] ? b.rt[s] : 1.0f;
    }
    return L;
}

// the two sub-face neighbour values across direction s (equal for single faces)
__device__ __forceinline__ void nb2(const float* f, const Lane& L, int s, float& v0, float& v1) {
    v0 = f[L.nidx[s]];
    v1 = f[L.nidx[s] + (L.edge[s] ? 1 : 0)];
}
__device__ __forceinline__ float nb1(const float* f, const Lane& L, int s) { return f[L.nidx[s]]; }

// Gradient along x and y and JST sensor of ONE staged scalar field at this lane's cell
// (cell_gradient :965 + JST_sensor :1077 in weight form).
__device__ __forceinline__ void cell_G(const float* f, const Lane& L, const BlockDesc2& bb, float uc, float& gx,
                                       float& gy, float& D) {
    D = 1e-7f;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const float rh = bb.rh[d];
        const int sL = 2 * d, sR = 2 * d + 1;
        float l0, l1, r0, r1;
        nb2(f, L, sL, l0, l1);
        nb2(f, L, sR, r0, r1);
        const float uLm = 0.5f * (l0 + l1);
        const float uRm = 0.5f * (r0 + r1);
        const float fr = uc + L.q[sR] * (uRm - uc);  // at_faces: (1-q)*u_self + q*u_nb
        const float fl = uc + L.q[sL] * (uLm - uc);
        const float g = (fr - fl) * rh;
        if (d == 0) gx = g; else gy = g;
        const float dr = uRm - uc, dl = uc - uLm;
        const float ar = 0.5f * (fabsf(r0 - uc) + fabsf(r1 - uc));
        const float al = 0.5f * (fabsf(uc - l0) + fabsf(uc - l1));
        const float gg = (dr - dl) * rh;
        const float ugg = (ar + al) * rh;
        D = fmaxf(D, (1e-7f + fabsf(gg)) * __builtin_amdgcn_rcpf(1e-7f + ugg));
    }
}

// ------------------------------------------------------------------------------------------
// pass A: gradients of NV variables along x and y + JST sensor of variable 0
// G layout as in ibh_fused.hip: grad of var v along dim d at G[(d*NV+v)*nc + c], sensor at G[2*NV*nc + c]
// LDS per wave: NV * 128 floats
// ------------------------------------------------------------------------------------------
template <int NV>
__device__ __forceinline__ void passA(const BlockDesc2* __restrict__ blocks, const int32_t* __restrict__ htab,
                                      int32_t blk, uint32_t nc, const float* __restrict__ u, uint32_t ldu,
                                      float* __restrict__ G, float* lds, int lane) {
    const BlockDesc2 bb = blocks[blk];  // by value: wave-uniform, lives in SGPRs, no per-lane descriptor loads
    const uint32_t hidx = (uint32_t)htab[(size_t)blk * 64 + lane];
    const uint32_t c = (uint32_t)bb.base + lane;
    float self[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        self[v] = ldg(u + (size_t)v * ldu, c);
        const float hv = ldg(u + (size_t)v * ldu, hidx);
        lds[v * 128 + lane] = self[v];
        lds[v * 128 + 64 + lane] = hv;
    }
    const Lane L = lane_info(bb, lane);
    wave_lds_sync();
    float gx[NV], gy[NV], D = 1e-7f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        float Dv;
        cell_G(lds + v * 128, L, bb, self[v], gx[v], gy[v], Dv);
        if (v == 0) D = Dv;
    }
    if (!L.general) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            stg(G + (size_t)v * nc, c, gx[v]);
            stg(G + (size_t)(NV + v) * nc, c, gy[v]);
        }
        stg(G + (size_t)(2 * NV) * nc, c, D);
    }
}

// ------------------------------------------------------------------------------------------
// pass B, advection: MUSCL(high_order) + upwind flux + Green-Gauss (test/advection.jl:67-83)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float minmod(float a, float b) {
    // min(|a|,|b|)*(sign a + sign b)/2  ==  median(a, b, 0): one v_med3_f32
    return __builtin_amdgcn_fmed3f(a, b, 0.0f);
}

// flux at the face between cell a (left/owner, half width dA) and b (right/neighbour, dB); inv = 1/(dA+dB)
__device__ __forceinline__ float adv_flux(float ua, float ub, float ga, float gb, float Da, float Db, float Ca,
                                          float Cb, float dA, float dB, float inv) {
#ifdef IBH_ABLATE_NOMATH
    return ua + ub + ga + gb + Da + Db + Ca + Cb + dA + dB + inv;
#endif
    const float guf = (ub - ua) * inv;
    const float gu = (2.0f * ga - guf) * dA;
    const float Du = (2.0f * gb - guf) * dB;
    const float s = minmod(Du, gu);
    const float l = ua + s, r = ub - s;
    const float uf = (ua * dB + ub * dA) * inv + (ga * dA - gb * dB) * 0.125f;
    const float Df = fmaxf(fmaxf(Da, Db), 1e-7f);
    const float uL = uf + Df * (l - uf);
    const float uR = uf + Df * (r - uf);
    const float Cf = (Ca * dB + Cb * dA) * inv;
    return 0.5f * ((uL + uR) * Cf + fabsf(Cf) * (uL - uR));
}

// role of lane r in the extra pass: group g = r>>3, boundary cell t = r&7
//   g=0: left side sub-face 0   g=1: bottom sub-face 0   g=2: left sub-face 1   g=3: bottom sub-face 1
//   g=4: right side sub-face 1  g=5: top sub-face 1      (g=6,7 repeat g=4,5; unused)
struct Role {
    int side, dn, pos, slot;
    bool low;
};
__device__ __forceinline__ Role role_of(int lane) {
    Role R;
    const int g = lane >> 3, t = lane & 7;
    // side = 2*dn + high;  g odd -> y sides, g >= 4 -> high sides
    R.dn = g & 1;
    const int high = (g >> 2) & 1;
    R.side = 2 * R.dn + high;
    const int k = g >= 2 ? 1 : 0;
    R.low = high == 0;
    // boundary cell t of the side: x sides -> column 0 / 7, row t; y sides -> row 0 / 7, column t
    const int stride = R.dn ? 1 : 8;
    const int offs = high * (R.dn ? 56 : 7);
    R.pos = t * stride + offs;
    R.slot = 64 + (R.side * 8 + t) * 2 + k;
    return R;
}

// LDS per wave: 6 fields x [tile | halo] : U, D, GX, GY, CX, CY (halo of GX/CX meaningful on x sides,
// of GY/CY on y sides) + extra[64]
#define BLK2_PASSB_LDS (6 * 128 + 64)

// fluxes + Green-Gauss of one block once U, D, GX, GY, CX, CY are staged as [tile | halo] fields
__device__ __forceinline__ void passB_adv_core(const BlockDesc2& bb, const Lane& L, int lane, uint32_t c, float* lds,
                                               float uc, float Dc, float gxc, float gyc, float cxc, float cyc,
                                               float* __restrict__ ud) {
    const float* fU = lds;
    const float* fD = lds + 128;
    const float* fGX = lds + 256;
    const float* fGY = lds + 384;
    const float* fCX = lds + 512;
    const float* fCY = lds + 640;
    float* ex = lds + 768;
    const float hx = bb.h[0], hy = bb.h[1], rhx = bb.rh[0], rhy = bb.rh[1];
    // per-side half-widths of the neighbour and 1/(dA+dB), wave-uniform (scalar registers)
    const float dBs[4] = {0.5f * hx * bb.rt[0], 0.5f * hx * bb.rt[1], 0.5f * hy * bb.rt[2], 0.5f * hy * bb.rt[3]};
    const float invs[4] = {2.0f * rhx * bb.q[0], 2.0f * rhx * bb.q[1], 2.0f * rhy * bb.q[2], 2.0f * rhy * bb.q[3]};
    // ---- main pass: right (x+) and top (y+) face of every cell, sub-face 0 on block sides
    float FR = adv_flux(uc, nb1(fU, L, 1), gxc, nb1(fGX, L, 1), Dc, nb1(fD, L, 1), cxc, nb1(fCX, L, 1), 0.5f * hx,
                        L.edge[1] ? dBs[1] : 0.5f * hx, L.edge[1] ? invs[1] : rhx);
    float FT = adv_flux(uc, nb1(fU, L, 3), gyc, nb1(fGY, L, 3), Dc, nb1(fD, L, 3), cyc, nb1(fCY, L, 3), 0.5f * hy,
                        L.edge[3] ? dBs[3] : 0.5f * hy, L.edge[3] ? invs[3] : rhy);
    // ---- extra pass, low sides only (the halo cell is the owner, this cell the neighbour): role lane r < 32,
    //   g = r>>3:  0: left sub-face 0   1: bottom sub-face 0   2: left sub-face 1   3: bottom sub-face 1
    {
        const int g = (lane >> 3) & 3, t = lane & 7;
        const int dn = g & 1, k = g >> 1;
        const int pos = dn ? t : 8 * t;
        const int slot = 64 + (dn * 16 + t) * 2 + k;  // side = 2*dn
        const float* fG = dn ? fGY : fGX;
        const float* fC = dn ? fCY : fCX;
        ex[lane] = adv_flux(fU[slot], fU[pos], fG[slot], fG[pos], fD[slot], fD[pos], fC[slot], fC[pos],
                            dn ? dBs[2] : dBs[0], dn ? 0.5f * hy : 0.5f * hx, dn ? invs[2] : invs[0]);
    }
    // ---- second sub-faces of the HIGH sides exist only next to finer blocks (~5 % of the sides): wave-uniform
    float FR1 = FR, FT1 = FT;
    if (bb.type[1] == SIDE_FINE)
        FR1 = adv_flux(uc, fU[L.nidx[1] + 1], gxc, fGX[L.nidx[1] + 1], Dc, fD[L.nidx[1] + 1], cxc, fCX[L.nidx[1] + 1],
                       0.5f * hx, dBs[1], invs[1]);
    if (bb.type[3] == SIDE_FINE)
        FT1 = adv_flux(uc, fU[L.nidx[3] + 1], gyc, fGY[L.nidx[3] + 1], Dc, fD[L.nidx[3] + 1], cyc, fCY[L.nidx[3] + 1],
                       0.5f * hy, dBs[3], invs[3]);
    // interior faces: left flux = right flux of lane-1, bottom flux = top flux of lane-8
    const float FLs = __shfl_up(FR, 1, 64);
    const float FBs = __shfl_up(FT, 8, 64);
    wave_lds_sync();
    const float eL = 0.5f * (ex[L.j] + ex[16 + L.j]), eB = 0.5f * (ex[8 + L.i] + ex[24 + L.i]);
    const float FL = L.edge[0] ? eL : FLs;
    const float FB = L.edge[2] ? eB : FBs;
    FR = L.edge[1] ? 0.5f * (FR + FR1) : FR;
    FT = L.edge[3] ? 0.5f * (FT + FT1) : FT;
    const float res = -((FR - FL) * rhx) - ((FT - FB) * rhy);
    if (!L.general) stg(ud, c, res);
}

__device__ __forceinline__ void passB_adv(const BlockDesc2* __restrict__ blocks, const int32_t* __restrict__ htab,
                                          int32_t blk, uint32_t nc, const float* __restrict__ u,
                                          const float* __restrict__ C, uint32_t ldc, const float* __restrict__ G,
                                          float* __restrict__ ud, float* lds, int lane) {
    const BlockDesc2 bb = blocks[blk];  // by value: wave-uniform, lives in SGPRs, no per-lane descriptor loads
    const uint32_t c = (uint32_t)bb.base + lane;
    float* fU = lds;
    float* fD = lds + 128;
    float* fGX = lds + 256;
    float* fGY = lds + 384;
    float* fCX = lds + 512;
    float* fCY = lds + 640;
    const float* Gs = G + (size_t)2 * nc;
    // issue order: halo table (needs nothing), own-cell loads (need the descriptor), and only then the
    // gathers that wait for the table -- the scheduling barrier keeps them from being hoisted in between
    const uint32_t hidx = (uint32_t)htab[(size_t)blk * 64 + lane];
    const float uc = ldg(u, c), Dc = ldg(Gs, c), gxc = ldg(G, c), gyc = ldg(G + nc, c);
    const float cxc = ldg(C, c), cyc = ldg(C + ldc, c);
    __builtin_amdgcn_sched_barrier(0);
    // halo slot of this lane: sides 0,1 need the x-gradient / Cx of the neighbour, sides 2,3 the y ones
    const int dn = lane >> 5;
#ifdef IBH_ABLATE_NOHALO
    const float hu = 0.f, hD = 0.f, hg = 0.f, hc = 0.f;
#else
    const float hu = ldg(u, hidx), hD = ldg(Gs, hidx);
    const float hg = ldg(G + (size_t)dn * nc, hidx);
    const float hc = ldg(C + (size_t)dn * ldc, hidx);
#endif
    fU[lane] = uc;
    fD[lane] = Dc;
    fGX[lane] = gxc;
    fGY[lane] = gyc;
    fCX[lane] = cxc;
    fCY[lane] = cyc;
    fU[64 + lane] = hu;
    fD[64 + lane] = hD;
    fGX[64 + lane] = hg;  // slots of sides 2,3 hold gy here; they are only read through fGY below
    fGY[64 + lane] = hg;
    fCX[64 + lane] = hc;
    fCY[64 + lane] = hc;
    const Lane L = lane_info(bb, lane);
    wave_lds_sync();
    passB_adv_core(bb, L, lane, c, lds, uc, Dc, gxc, gyc, cxc, cyc, ud);
}

// ------------------------------------------------------------------------------------------
// pass B, Euler: MUSCL(high_order) on P = [p T u v] with the pressure sensor, HLL flux
// (cfd.jl:459-508, Float32 here -- the reference promotes the last combine to Float64), Green-Gauss.
// Own-cell gradients are recomputed from the staged P tile (8 fewer field loads than reading them back).
// ------------------------------------------------------------------------------------------
struct Gas {
    float R, gamma;
};

__device__ __forceinline__ void euler_side(const float* P, int dn, const Gas& gas, float* Q, float* F, float& un,
                                           float& a) {
    const float p = P[0];
    const float T = fmaxf(P[1], 10.0f);
    const float k = 0.5f * (P[2] * P[2] + P[3] * P[3]);
    const float rho = p * __builtin_amdgcn_rcpf(gas.R * T);
    const float E = rho * (gas.R / (gas.gamma - 1.0f) * T + k);
    Q[0] = rho;
    Q[1] = E;
    Q[2] = rho * P[2];
    Q[3] = rho * P[3];
    un = dn ? P[3] : P[2];
    a = __builtin_amdgcn_sqrtf(gas.gamma * gas.R * T);  // v_sqrt_f32 (1 ulp); the literal path keeps IEEE sqrtf
    F[0] = Q[0] * un;
    F[1] = (Q[1] + p) * un;
    F[2] = Q[2] * un + (dn ? 0.0f : p);
    F[3] = Q[3] * un + (dn ? p : 0.0f);
}

// a = left/owner cell, b = right/neighbour; Pa/Pb primitive states, ga/gb their gradients along the face normal
__device__ __forceinline__ void euler_flux(const float* Pa, const float* Pb, const float* ga, const float* gb, float Da,
                                           float Db, float dA, float dB, float inv, int dn, const Gas& gas, float* F) {
    float PL[4], PR[4];
    const float Df = fmaxf(fmaxf(Da, Db), 1e-7f);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const float guf = (Pb[v] - Pa[v]) * inv;
        const float gu = (2.0f * ga[v] - guf) * dA;
        const float Du = (2.0f * gb[v] - guf) * dB;
        const float s = minmod(Du, gu);
        const float l = Pa[v] + s, r = Pb[v] - s;
        const float uf = (Pa[v] * dB + Pb[v] * dA) * inv + (ga[v] * dA - gb[v] * dB) * 0.125f;
        PL[v] = uf + Df * (l - uf);
        PR[v] = uf + Df * (r - uf);
    }
    float QL[4], FL[4], QR[4], FR[4], uL, aL, uR, aR;
    euler_side(PL, dn, gas, QL, FL, uL, aL);
    euler_side(PR, dn, gas, QR, FR, uR, aR);
    const float SR = fminf(uR - aR, 0.0f);
    const float SL = fmaxf(uL + aL, 0.0f);
    const float rs = __builtin_amdgcn_rcpf(SL - SR);
#pragma unroll
    for (int v = 0; v < 4; ++v) F[v] = (SL * FL[v] - SR * FR[v] + SR * SL * (QR[v] - QL[v])) * rs;
}

// LDS per wave: P[4], D, GX[4], GY[4] as [tile | halo] fields (13 x 128) + extra flux[4][64]
#define BLK2_EULER_LDS (13 * 128 + 4 * 64)

__device__ __forceinline__ void passB_euler(const BlockDesc2* __restrict__ blocks, const int32_t* __restrict__ htab,
                                            int32_t blk, uint32_t nc, const float* __restrict__ P, uint32_t ldp,
                                            const float* __restrict__ G, float* __restrict__ Rr, uint32_t ldr, Gas gas,
                                            float* lds, int lane) {
    const BlockDesc2 bb = blocks[blk];  // by value: wave-uniform, lives in SGPRs, no per-lane descriptor loads
    const uint32_t hidx = (uint32_t)htab[(size_t)blk * 64 + lane];
    const uint32_t c = (uint32_t)bb.base + lane;
    float* fP = lds;             // [4][128]
    float* fD = lds + 4 * 128;   // [128]
    float* fGX = lds + 5 * 128;  // [4][128]
    float* fGY = lds + 9 * 128;  // [4][128]
    float* ex = lds + 13 * 128;  // [4][64]
    const int dnl = lane >> 5;
    float Pc[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        Pc[v] = ldg(P + (size_t)v * ldp, c);
        fP[v * 128 + lane] = Pc[v];
        fP[v * 128 + 64 + lane] = ldg(P + (size_t)v * ldp, hidx);
        const float hg = ldg(G + (size_t)(dnl * 4 + v) * nc, hidx);
        fGX[v * 128 + 64 + lane] = hg;
        fGY[v * 128 + 64 + lane] = hg;
    }
    fD[64 + lane] = ldg(G + (size_t)8 * nc, hidx);
    const Lane L = lane_info(bb, lane);
    wave_lds_sync();
    // own-cell gradients of the 4 primitives and the pressure sensor
    float gxc[4], gyc[4], Dc = 1e-7f;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        float Dv;
        cell_G(fP + v * 128, L, bb, Pc[v], gxc[v], gyc[v], Dv);
        if (v == 0) Dc = Dv;
    }
    if (L.general) {  // cells of a face-list side: their values were written by the face-list body of pass A
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            gxc[v] = ldg(G + (size_t)v * nc, c);
            gyc[v] = ldg(G + (size_t)(4 + v) * nc, c);
        }
        Dc = ldg(G + (size_t)8 * nc, c);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        fGX[v * 128 + lane] = gxc[v];
        fGY[v * 128 + lane] = gyc[v];
    }
    fD[lane] = Dc;
    wave_lds_sync();

    const float hx = bb.h[0], hy = bb.h[1], rhx = bb.rh[0], rhy = bb.rh[1];
    const float dBs[4] = {0.5f * hx * bb.rt[0], 0.5f * hx * bb.rt[1], 0.5f * hy * bb.rt[2], 0.5f * hy * bb.rt[3]};
    const float invs[4] = {2.0f * rhx * bb.q[0], 2.0f * rhx * bb.q[1], 2.0f * rhy * bb.q[2], 2.0f * rhy * bb.q[3]};
    float FR[4], FT[4], FR1[4], FT1[4];
    {
        float Pb[4], gb[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            Pb[v] = nb1(fP + v * 128, L, 1);
            gb[v] = nb1(fGX + v * 128, L, 1);
        }
        euler_flux(Pc, Pb, gxc, gb, Dc, nb1(fD, L, 1), 0.5f * hx, L.edge[1] ? dBs[1] : 0.5f * hx,
                   L.edge[1] ? invs[1] : rhx, 0, gas, FR);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            Pb[v] = nb1(fP + v * 128, L, 3);
            gb[v] = nb1(fGY + v * 128, L, 3);
        }
        euler_flux(Pc, Pb, gyc, gb, Dc, nb1(fD, L, 3), 0.5f * hy, L.edge[3] ? dBs[3] : 0.5f * hy,
                   L.edge[3] ? invs[3] : rhy, 1, gas, FT);
    }
    {   // extra pass, low sides only (halo cell = owner): role lane r < 32, g = r>>3:
        //   0: left sub-face 0   1: bottom sub-face 0   2: left sub-face 1   3: bottom sub-face 1
        const int g = (lane >> 3) & 3, t = lane & 7;
        const int dn = g & 1, k = g >> 1;
        const int pos = dn ? t : 8 * t;
        const int slot = 64 + (dn * 16 + t) * 2 + k;
        const float* fG = dn ? fGY : fGX;
        float Pa[4], Pb[4], ga[4], gb[4], X[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            Pa[v] = fP[v * 128 + slot];
            Pb[v] = fP[v * 128 + pos];
            ga[v] = fG[v * 128 + slot];
            gb[v] = fG[v * 128 + pos];
        }
        euler_flux(Pa, Pb, ga, gb, fD[slot], fD[pos], dn ? dBs[2] : dBs[0], dn ? 0.5f * hy : 0.5f * hx,
                   dn ? invs[2] : invs[0], dn, gas, X);
#pragma unroll
        for (int v = 0; v < 4; ++v) ex[v * 64 + lane] = X[v];
    }
    // second sub-faces of the HIGH sides exist only next to finer blocks: wave-uniform branches
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        FR1[v] = FR[v];
        FT1[v] = FT[v];
    }
    if (bb.type[1] == SIDE_FINE) {
        float Pb[4], gb[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            Pb[v] = fP[v * 128 + L.nidx[1] + 1];
            gb[v] = fGX[v * 128 + L.nidx[1] + 1];
        }
        euler_flux(Pc, Pb, gxc, gb, Dc, fD[L.nidx[1] + 1], 0.5f * hx, dBs[1], invs[1], 0, gas, FR1);
    }
    if (bb.type[3] == SIDE_FINE) {
        float Pb[4], gb[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            Pb[v] = fP[v * 128 + L.nidx[3] + 1];
            gb[v] = fGY[v * 128 + L.nidx[3] + 1];
        }
        euler_flux(Pc, Pb, gyc, gb, Dc, fD[L.nidx[3] + 1], 0.5f * hy, dBs[3], invs[3], 1, gas, FT1);
    }
    float FLs[4], FBs[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        FLs[v] = __shfl_up(FR[v], 1, 64);
        FBs[v] = __shfl_up(FT[v], 8, 64);
    }
    wave_lds_sync();
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const float* e = ex + v * 64;
        const float eL = 0.5f * (e[L.j] + e[16 + L.j]), eB = 0.5f * (e[8 + L.i] + e[24 + L.i]);
        const float fl = L.edge[0] ? eL : FLs[v];
        const float fb = L.edge[2] ? eB : FBs[v];
        const float fr = L.edge[1] ? 0.5f * (FR[v] + FR1[v]) : FR[v];
        const float ft = L.edge[3] ? 0.5f * (FT[v] + FT1[v]) : FT[v];
        const float res = -((fr - fl) * rhx) - ((ft - fb) * rhy);
        if (!L.general) stg(Rr + (size_t)v * ldr, c, res);
    }
}

#pragma clang fp contract(off)

}  // namespace blk2
