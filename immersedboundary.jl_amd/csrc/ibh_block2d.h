// Block fast path, 2-D, 8x8 blocks: one 64-lane wavefront per block, lane = cell (x fastest).
//
// This is the tuned form of the fast path (the literal IEEE form lives in ibh_fused.hip and is
// selected with IBH_EXACT).  Differences, all within the 1e-5 norm-wise parity tolerance:
//   * the divisions of the reference formulas are replaced by per-block reciprocals
//     (1/hx, 1/hy are wave-uniform) and per-side weight constants: inside a block and across
//     same-level / 2:1 sides the spacing ratio h_n/h_o is 1, 2 or 1/2, so
//     at_faces = (1-q)*u_o + q*u_n with q = 1/(1+ratio) in {1/2, 1/3, 2/3} and
//     1/(d_o+d_n) = 2*q/h;   the sensor ratio uses v_rcp_f32;
//   * every interior face flux is computed once, by the lane on its left/bottom, and handed
//     to the neighbour lane with a wavefront shuffle; block-boundary faces that no lane owns
//     (left and bottom sides, second sub-face on 2:1 fine sides) are computed together in one
//     extra pass by otherwise idle lanes;
//   * FMA contraction is allowed in this file.
// LDS per wave: cell tiles + one halo slot per (side, boundary cell, sub-face), filled by a
// single gather instruction per field (lane = slot).
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

// local id of the k-th neighbour cell across side s for boundary cell t, or -1
__device__ __forceinline__ int32_t halo_cell(const BlockDesc2& b, int s, int t, int k) {
    int ty = b.type[s];
    int tt;
    int32_t base;
    if (ty == SIDE_SAME) {
        if (k) return -1;
        tt = t;
        base = b.nb[s][0];
    } else if (ty == SIDE_COARSE) {
        if (k) return -1;
        tt = 4 * b.sub[s] + (t >> 1);
        base = b.nb[s][0];
    } else if (ty == SIDE_FINE) {
        tt = 2 * (t & 3) + k;
        base = b.nb[s][t >> 2];
    } else if (ty == SIDE_MIRROR) {
        if (k) return -1;
        // mirror face: owner == neighbour == the boundary cell itself (ImmersedBoundary.jl:653-660)
        return b.base + ((s == 0) ? 8 * t : (s == 1) ? 7 + 8 * t : (s == 2) ? t : t + 56);
    } else {
        return -1;
    }
    int pos = (s == 0) ? 7 + 8 * tt : (s == 1) ? 8 * tt : (s == 2) ? tt + 56 : tt;
    return base + pos;
}

// spacing ratio h_nb/h across side type, and q = 1/(1+ratio)
__device__ __forceinline__ float side_ratio(int ty) { return ty == SIDE_COARSE ? 2.0f : (ty == SIDE_FINE ? 0.5f : 1.0f); }
__device__ __forceinline__ float side_q(int ty) {
    return ty == SIDE_COARSE ? (1.0f / 3.0f) : (ty == SIDE_FINE ? (2.0f / 3.0f) : 0.5f);
}

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct Lane {
    int i, j;
    bool edge[4];
    bool general;
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
    for (int s = 0; s < 4; ++s) L.general |= L.edge[s] && b.type[s] == SIDE_GENERAL;
    return L;
}

// neighbour value(s) of a staged field across direction s
__device__ __forceinline__ void nbv(const float* tile, const float* halo, int lane, const Lane& L, int s, float& v0,
                                    float& v1) {
    if (!L.edge[s]) {
        const int off = (s == 0) ? -1 : (s == 1) ? 1 : (s == 2) ? -8 : 8;
        v0 = tile[lane + off];
        v1 = v0;
    } else {
        const int t = (s < 2) ? L.j : L.i;
        v0 = halo[(s * 8 + t) * 2];
        v1 = halo[(s * 8 + t) * 2 + 1];
    }
}

// ------------------------------------------------------------------------------------------
// Both passes process BPW blocks per wavefront: all descriptor loads, then all tile loads and halo
// gathers of the BPW blocks are issued before anything is consumed, so one wave keeps BPW times
// more memory requests in flight and the grid fits in a single residency round.
// ------------------------------------------------------------------------------------------

// Gradient along x and y and JST sensor of ONE staged scalar field at this lane's cell
// (cell_gradient :965 + JST_sensor :1077 in weight form).
__device__ __forceinline__ void cell_G(const float* tile, const float* halo, int lane, const Lane& L,
                                       const BlockDesc2& bb, float uc, float& gx, float& gy, float& D) {
    D = 1e-7f;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const float rh = __builtin_amdgcn_rcpf(bb.h[d]);
        const int sL = 2 * d, sR = 2 * d + 1;
        const int tyL = bb.type[sL], tyR = bb.type[sR];
        const float qL = L.edge[sL] ? side_q(tyL) : 0.5f;
        const float qR = L.edge[sR] ? side_q(tyR) : 0.5f;
        const bool twoL = L.edge[sL] && tyL == SIDE_FINE;
        const bool twoR = L.edge[sR] && tyR == SIDE_FINE;
        float l0, l1, r0, r1;
        nbv(tile, halo, lane, L, sL, l0, l1);
        nbv(tile, halo, lane, L, sR, r0, r1);
        const float uLm = twoL ? 0.5f * (l0 + l1) : l0;
        const float uRm = twoR ? 0.5f * (r0 + r1) : r0;
        const float fr = uc + qR * (uRm - uc);
        const float fl = uc + qL * (uLm - uc);
        const float g = (fr - fl) * rh;
        if (d == 0) gx = g; else gy = g;
        const float dr = uRm - uc, dl = uc - uLm;
        const float ar = twoR ? 0.5f * (fabsf(r0 - uc) + fabsf(r1 - uc)) : fabsf(dr);
        const float al = twoL ? 0.5f * (fabsf(uc - l0) + fabsf(uc - l1)) : fabsf(dl);
        const float gg = (dr - dl) * rh;
        const float ugg = (ar + al) * rh;
        D = fmaxf(D, (1e-7f + fabsf(gg)) * __builtin_amdgcn_rcpf(1e-7f + ugg));
    }
}

// pass A: gradients of NV variables along x and y + JST sensor of variable 0.
// (Measured, profiles/r1_v1: storing only the block rim and recomputing own-cell values in pass B is
// slower for the scalar residual -- 17.8 vs 15.6 us per sweep -- so everything is stored.)
// G layout as in ibh_fused.hip: grad of var v along dim d at G[(d*NV+v)*nc + c], sensor at G[2*NV*nc + c]
// LDS per wave: BPW * (tile[NV][64] + halo[NV][64])
template <int NV, int BPW>
__device__ __forceinline__ void passA(const BlockDesc2* __restrict__ blocks, int32_t blk0, int32_t nblk, uint32_t nc,
                                      const float* __restrict__ u, uint32_t ldu, float* __restrict__ G, float* lds,
                                      int lane) {
    const BlockDesc2* b[BPW];
    bool valid[BPW];
    float self[BPW][NV], hv[BPW][NV];
#pragma unroll
    for (int r = 0; r < BPW; ++r) {
        valid[r] = blk0 + r < nblk;
        b[r] = blocks + (valid[r] ? blk0 + r : nblk - 1);
    }
#pragma unroll
    for (int r = 0; r < BPW; ++r) {
        const int32_t hidx = halo_cell(*b[r], lane >> 4, (lane >> 1) & 7, lane & 1);
        const uint32_t c = (uint32_t)b[r]->base + lane;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            self[r][v] = ldg(u + (size_t)v * ldu, c);
            hv[r][v] = hidx >= 0 ? ldg(u + (size_t)v * ldu, (uint32_t)hidx) : 0.0f;
        }
    }
#pragma unroll
    for (int r = 0; r < BPW; ++r) {
        float* tile = lds + r * (2 * NV * 64);
        float* halo = tile + NV * 64;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            tile[v * 64 + lane] = self[r][v];
            halo[v * 64 + lane] = hv[r][v];
        }
    }
    wave_lds_sync();
#pragma unroll
    for (int r = 0; r < BPW; ++r) {
        const BlockDesc2& bb = *b[r];
        const Lane L = lane_info(bb, lane);
        const float* tile = lds + r * (2 * NV * 64);
        const float* halo = tile + NV * 64;
        const uint32_t c = (uint32_t)bb.base + lane;
        const bool store = valid[r] && !L.general;
        float D = 1e-7f;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const float rh = __builtin_amdgcn_rcpf(bb.h[d]);
            const int sL = 2 * d, sR = 2 * d + 1;
            const int tyL = bb.type[sL], tyR = bb.type[sR];
            const float qL = L.edge[sL] ? side_q(tyL) : 0.5f;
            const float qR = L.edge[sR] ? side_q(tyR) : 0.5f;
            const bool twoL = L.edge[sL] && tyL == SIDE_FINE;
            const bool twoR = L.edge[sR] && tyR == SIDE_FINE;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                float l0, l1, r0, r1;
                nbv(tile + v * 64, halo + v * 64, lane, L, sL, l0, l1);
                nbv(tile + v * 64, halo + v * 64, lane, L, sR, r0, r1);
                const float uc = self[r][v];
                const float uLm = twoL ? 0.5f * (l0 + l1) : l0;
                const float uRm = twoR ? 0.5f * (r0 + r1) : r0;
                // at_faces with weights: face = (1-q)*u_self + q*u_nb
                const float fr = uc + qR * (uRm - uc);
                const float fl = uc + qL * (uLm - uc);
                if (store) stg(G + (size_t)(d * NV + v) * nc, c, (fr - fl) * rh);
                if (v == 0) {
                    const float dr = uRm - uc, dl = uc - uLm;
                    const float ar = twoR ? 0.5f * (fabsf(r0 - uc) + fabsf(r1 - uc)) : fabsf(dr);
                    const float al = twoL ? 0.5f * (fabsf(uc - l0) + fabsf(uc - l1)) : fabsf(dl);
                    const float gg = (dr - dl) * rh;
                    const float ugg = (ar + al) * rh;
                    D = fmaxf(D, (1e-7f + fabsf(gg)) * __builtin_amdgcn_rcpf(1e-7f + ugg));
                }
            }
        }
        if (store) stg(G + (size_t)(2 * NV) * nc, c, D);
    }
}

// ------------------------------------------------------------------------------------------
// pass B, advection: MUSCL(high_order) + upwind flux + Green-Gauss (test/advection.jl:67-83)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float minmod(float a, float b) {
    // min(|a|,|b|)*(sign a + sign b)/2
    float m = fminf(fabsf(a), fabsf(b));
    return (a * b > 0.0f) ? copysignf(m, a) : 0.0f;
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

// LDS per wave and block: tile[6][64] (u, D, gx, gy, Cx, Cy), halo[4][64] (u, D, gN, CN), extra[64]
#define BLK2_PASSB_LDS (11 * 64)

template <int BPW>
__device__ __forceinline__ void passB_adv(const BlockDesc2* __restrict__ blocks, int32_t blk0, int32_t nblk,
                                          uint32_t nc, const float* __restrict__ u, const float* __restrict__ C,
                                          uint32_t ldc, const float* __restrict__ G, float* __restrict__ ud, float* lds,
                                          int lane) {
    const BlockDesc2* b[BPW];
    bool valid[BPW];
    float sv[BPW][6], hvv[BPW][4];
    const int hs = lane >> 4;  // side of this lane's halo slot
#pragma unroll
    for (int r = 0; r < BPW; ++r) {
        valid[r] = blk0 + r < nblk;
        b[r] = blocks + (valid[r] ? blk0 + r : nblk - 1);
    }
#pragma unroll
    for (int r = 0; r < BPW; ++r) {
        const int32_t hidx = halo_cell(*b[r], hs, (lane >> 1) & 7, lane & 1);
        const uint32_t c = (uint32_t)b[r]->base + lane;
        sv[r][0] = ldg(u, c);
        sv[r][1] = ldg(G + (size_t)2 * nc, c);
        sv[r][2] = ldg(G, c);
        sv[r][3] = ldg(G + nc, c);
        sv[r][4] = ldg(C, c);
        sv[r][5] = ldg(C + ldc, c);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#ifdef IBH_ABLATE_NOHALO
        if (false) {
#else
        if (hidx >= 0) {
#endif
            const uint32_t hi = (uint32_t)hidx;
            const int dn = hs >> 1;  // normal dim of the slot's side
            a0 = ldg(u, hi);
            a1 = ldg(G + (size_t)2 * nc, hi);
            a2 = ldg(G + (size_t)dn * nc, hi);
            a3 = ldg(C + (size_t)dn * ldc, hi);
        }
        hvv[r][0] = a0;
        hvv[r][1] = a1;
        hvv[r][2] = a2;
        hvv[r][3] = a3;
    }
#pragma unroll
    for (int r = 0; r < BPW; ++r) {
        float* base = lds + r * BLK2_PASSB_LDS;
#pragma unroll
        for (int q = 0; q < 6; ++q) base[q * 64 + lane] = sv[r][q];
#pragma unroll
        for (int q = 0; q < 4; ++q) base[384 + q * 64 + lane] = hvv[r][q];
    }
    wave_lds_sync();

    float FRr[BPW], FTr[BPW];
#pragma unroll
    for (int r = 0; r < BPW; ++r) {
        const BlockDesc2& bb = *b[r];
        const Lane L = lane_info(bb, lane);
        float* base = lds + r * BLK2_PASSB_LDS;
        const float *tU = base, *tD = base + 64, *tG = base + 128, *tC = base + 256;
        const float *hU = base + 384, *hD = base + 448, *hG = base + 512, *hC = base + 576;
        float* ex = base + 640;
        const float uc = sv[r][0], Dc = sv[r][1], gxc = sv[r][2], gyc = sv[r][3], cxc = sv[r][4], cyc = sv[r][5];
        const float hx = bb.h[0], hy = bb.h[1];
        const float rhx = __builtin_amdgcn_rcpf(hx), rhy = __builtin_amdgcn_rcpf(hy);
        // ---- main pass: right (x+) and top (y+) face of every cell, sub-face 0 on block sides
        {
            float ub, gb, Db, Cb, d1;
            nbv(tU, hU, lane, L, 1, ub, d1);
            nbv(tD, hD, lane, L, 1, Db, d1);
            nbv(tG, hG, lane, L, 1, gb, d1);
            nbv(tC, hC, lane, L, 1, Cb, d1);
            const int ty = bb.type[1];
            const float rt = L.edge[1] ? side_ratio(ty) : 1.0f;
            const float q = L.edge[1] ? side_q(ty) : 0.5f;
            FRr[r] = adv_flux(uc, ub, gxc, gb, Dc, Db, cxc, Cb, 0.5f * hx, 0.5f * hx * rt, 2.0f * rhx * q);
        }
        {
            float ub, gb, Db, Cb, d1;
            nbv(tU, hU, lane, L, 3, ub, d1);
            nbv(tD, hD, lane, L, 3, Db, d1);
            nbv(tG + 64, hG, lane, L, 3, gb, d1);
            nbv(tC + 64, hC, lane, L, 3, Cb, d1);
            const int ty = bb.type[3];
            const float rt = L.edge[3] ? side_ratio(ty) : 1.0f;
            const float q = L.edge[3] ? side_q(ty) : 0.5f;
            FTr[r] = adv_flux(uc, ub, gyc, gb, Dc, Db, cyc, Cb, 0.5f * hy, 0.5f * hy * rt, 2.0f * rhy * q);
        }
        // ---- extra pass: faces no lane owns.  role lane = 8*g + t
        //   g=0: left side sub-face 0      g=1: bottom side sub-face 0
        //   g=2: left side sub-face 1      g=3: bottom side sub-face 1     (FINE sides only)
        //   g=4: right side sub-face 1     g=5: top side sub-face 1        (FINE sides only)
        {
            const int g = lane >> 3, t = lane & 7;
            const int side = (g == 0 || g == 2) ? 0 : (g == 1 || g == 3) ? 2 : (g == 4) ? 1 : 3;
            const int k = g >= 2 ? 1 : 0;
            const int ty = bb.type[side & 3];
            const bool active = g < 2 || (g < 6 && ty == SIDE_FINE);
            float X = 0.0f;
            if (active) {
                const int dn = side >> 1;
                const int pos = (side == 0) ? 8 * t : (side == 1) ? 7 + 8 * t : (side == 2) ? t : t + 56;
                const int slot = (side * 8 + t) * 2 + k;
                const float us = tU[pos], Ds = tD[pos], gs = tG[dn * 64 + pos], Cs = tC[dn * 64 + pos];
                const float uh = hU[slot], Dh = hD[slot], gh = hG[slot], Ch = hC[slot];
                const float h = dn ? hy : hx, rh = dn ? rhy : rhx;
                const float rt = side_ratio(ty), q = side_q(ty);
                const float dS = 0.5f * h, dH = 0.5f * h * rt, inv = 2.0f * rh * q;
                if ((side & 1) == 0)  // low side: halo cell is the owner (left), this cell the neighbour
                    X = adv_flux(uh, us, gh, gs, Dh, Ds, Ch, Cs, dH, dS, inv);
                else
                    X = adv_flux(us, uh, gs, gh, Ds, Dh, Cs, Ch, dS, dH, inv);
            }
            ex[lane] = X;
        }
    }
    wave_lds_sync();
#pragma unroll
    for (int r = 0; r < BPW; ++r) {
        const BlockDesc2& bb = *b[r];
        const Lane L = lane_info(bb, lane);
        const float* ex = lds + r * BLK2_PASSB_LDS + 640;
        const float rhx = __builtin_amdgcn_rcpf(bb.h[0]), rhy = __builtin_amdgcn_rcpf(bb.h[1]);
        float FR = FRr[r], FT = FTr[r];
        // interior faces: left flux = right flux of lane-1, bottom flux = top flux of lane-8
        float FL = __shfl_up(FR, 1, 64);
        float FB = __shfl_up(FT, 8, 64);
        if (L.edge[0]) {
            FL = ex[L.j];
            if (bb.type[0] == SIDE_FINE) FL = 0.5f * (FL + ex[16 + L.j]);
        }
        if (L.edge[2]) {
            FB = ex[8 + L.i];
            if (bb.type[2] == SIDE_FINE) FB = 0.5f * (FB + ex[24 + L.i]);
        }
        if (L.edge[1] && bb.type[1] == SIDE_FINE) FR = 0.5f * (FR + ex[32 + L.j]);
        if (L.edge[3] && bb.type[3] == SIDE_FINE) FT = 0.5f * (FT + ex[40 + L.i]);
        const float res = -((FR - FL) * rhx) - ((FT - FB) * rhy);
        if (valid[r] && !L.general) stg(ud, (uint32_t)bb.base + lane, res);
    }
}


// ------------------------------------------------------------------------------------------
// pass B, Euler: MUSCL(high_order) on P = [p T u v] with the pressure sensor, HLL flux
// (cfd.jl:459-508, Float32 here -- the reference promotes the last combine to Float64), Green-Gauss.
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
    un = P[2 + dn];
    a = sqrtf(gas.gamma * gas.R * T);
    F[0] = Q[0] * un;
    F[1] = (Q[1] + p) * un;
    F[2] = Q[2] * un;
    F[3] = Q[3] * un;
    F[2 + dn] += p;
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

// LDS per wave: tiles P[4], D, gx[4], gy[4] (13 x 64); halos P[4], D, gN[4] (9 x 64); extra flux[4] (4 x 64)
#define BLK2_EULER_LDS (26 * 64)

__device__ __forceinline__ void passB_euler(const BlockDesc2* __restrict__ blocks, int32_t blk, uint32_t nc,
                                            const float* __restrict__ P, uint32_t ldp, const float* __restrict__ G,
                                            float* __restrict__ Rr, uint32_t ldr, Gas gas, float* lds, int lane) {
    const BlockDesc2& bb = blocks[blk];
    const Lane L = lane_info(bb, lane);
    const int hs = lane >> 4;
    const int32_t hidx = halo_cell(bb, hs, (lane >> 1) & 7, lane & 1);
    float* tP = lds;              // [4][64]
    float* tD = lds + 4 * 64;     // [64]
    float* tG = lds + 5 * 64;     // [2][4][64]
    float* hP = lds + 13 * 64;    // [4][64]
    float* hD = lds + 17 * 64;
    float* hG = lds + 18 * 64;    // [4][64] gradient along the slot's side normal
    float* ex = lds + 22 * 64;    // [4][64]
    const uint32_t c = (uint32_t)bb.base + lane;
    float Pc[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        Pc[v] = ldg(P + (size_t)v * ldp, c);
        tP[v * 64 + lane] = Pc[v];
    }
    {
        const int dn = hs >> 1;
        const bool ok = hidx >= 0;
        const uint32_t hi = ok ? (uint32_t)hidx : 0u;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            hP[v * 64 + lane] = ok ? ldg(P + (size_t)v * ldp, hi) : 0.0f;
            hG[v * 64 + lane] = ok ? ldg(G + (size_t)(dn * 4 + v) * nc, hi) : 0.0f;
        }
        hD[lane] = ok ? ldg(G + (size_t)8 * nc, hi) : 0.0f;
    }
    wave_lds_sync();
    // own-cell gradients of the 4 primitives and the pressure sensor
    float gxc[4], gyc[4], Dc = 1e-7f;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        float Dv;
        cell_G(tP + v * 64, hP + v * 64, lane, L, bb, Pc[v], gxc[v], gyc[v], Dv);
        if (v == 0) Dc = Dv;
        if (L.general) {  // face-list side: values written by the face-list body of pass A
            gxc[v] = ldg(G + (size_t)v * nc, c);
            gyc[v] = ldg(G + (size_t)(4 + v) * nc, c);
            if (v == 0) Dc = ldg(G + (size_t)8 * nc, c);
        }
        tG[v * 64 + lane] = gxc[v];
        tG[(4 + v) * 64 + lane] = gyc[v];
    }
    tD[lane] = Dc;
    wave_lds_sync();

    const float hx = bb.h[0], hy = bb.h[1];
    const float rhx = __builtin_amdgcn_rcpf(hx), rhy = __builtin_amdgcn_rcpf(hy);
    float FR[4], FT[4];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const int s = 2 * d + 1;
        float Pb[4], gb[4], Db, d1;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            nbv(tP + v * 64, hP + v * 64, lane, L, s, Pb[v], d1);
            nbv(tG + (d * 4 + v) * 64, hG + v * 64, lane, L, s, gb[v], d1);
        }
        nbv(tD, hD, lane, L, s, Db, d1);
        const int ty = bb.type[s];
        const float rt = L.edge[s] ? side_ratio(ty) : 1.0f;
        const float q = L.edge[s] ? side_q(ty) : 0.5f;
        const float h = d ? hy : hx, rh = d ? rhy : rhx;
        euler_flux(Pc, Pb, d ? gyc : gxc, gb, Dc, Db, 0.5f * h, 0.5f * h * rt, 2.0f * rh * q, d, gas, d ? FT : FR);
    }
    {   // extra pass (same role map as the advection kernel)
        const int g = lane >> 3, t = lane & 7;
        const int side = (g == 0 || g == 2) ? 0 : (g == 1 || g == 3) ? 2 : (g == 4) ? 1 : 3;
        const int k = g >= 2 ? 1 : 0;
        const int ty = bb.type[side & 3];
        const bool active = g < 2 || (g < 6 && ty == SIDE_FINE);
        float X[4] = {0.f, 0.f, 0.f, 0.f};
        if (active) {
            const int dn = side >> 1;
            const int pos = (side == 0) ? 8 * t : (side == 1) ? 7 + 8 * t : (side == 2) ? t : t + 56;
            const int slot = (side * 8 + t) * 2 + k;
            float Ps[4], Ph[4], gs[4], gh[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                Ps[v] = tP[v * 64 + pos];
                gs[v] = tG[(dn * 4 + v) * 64 + pos];
                Ph[v] = hP[v * 64 + slot];
                gh[v] = hG[v * 64 + slot];
            }
            const float Ds = tD[pos], Dh = hD[slot];
            const float h = dn ? hy : hx, rh = dn ? rhy : rhx;
            const float rt = side_ratio(ty), q = side_q(ty);
            const float dS = 0.5f * h, dH = 0.5f * h * rt, inv = 2.0f * rh * q;
            if ((side & 1) == 0)
                euler_flux(Ph, Ps, gh, gs, Dh, Ds, dH, dS, inv, dn, gas, X);
            else
                euler_flux(Ps, Ph, gs, gh, Ds, Dh, dS, dH, inv, dn, gas, X);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) ex[v * 64 + lane] = X[v];
    }
    wave_lds_sync();
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        float fr = FR[v], ft = FT[v];
        float fl = __shfl_up(fr, 1, 64);
        float fb = __shfl_up(ft, 8, 64);
        const float* e = ex + v * 64;
        if (L.edge[0]) {
            fl = e[L.j];
            if (bb.type[0] == SIDE_FINE) fl = 0.5f * (fl + e[16 + L.j]);
        }
        if (L.edge[2]) {
            fb = e[8 + L.i];
            if (bb.type[2] == SIDE_FINE) fb = 0.5f * (fb + e[24 + L.i]);
        }
        if (L.edge[1] && bb.type[1] == SIDE_FINE) fr = 0.5f * (fr + e[32 + L.j]);
        if (L.edge[3] && bb.type[3] == SIDE_FINE) ft = 0.5f * (ft + e[40 + L.i]);
        const float res = -((fr - fl) * rhx) - ((ft - fb) * rhy);
        if (!L.general) stg(Rr + (size_t)v * ldr, c, res);
    }
}

#pragma clang fp contract(off)

}  // namespace blk2
