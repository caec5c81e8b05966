// Column sweep of the 3-D scalar advection-JST-MUSCL residual (test/advection.jl:67-83 on an octree partition): the design
// of the 3-D Euler column sweep (ibh_strip3d_euler.h) with one transported variable.  ONE wavefront per 8x8x8 block; in
// every direction a lane owns a column of 8 cells along it and the halo slots at its two ends, so every face of the
// column has both cells, their slopes and sensors in the lane's registers; between directions u, the sensor and the
// residual are transposed through a 2.3 KB wave-private LDS buffer.  The velocity component normal to a pass is loaded
// from global memory directly in that pass's layout (it is used nowhere else).
//
//   order: u as z-columns -> sensor z, y, x (numerator / denominator carried: one reciprocal per cell) -> fluxes x, y, z ->
//   the z pass stores coalesced dwords.
// Same arithmetic and tables as strip3::sweep_strip (ibh_strip3d.h).
#pragma once
#include "ibh_strip3d_euler.h"

namespace cols3 {

#pragma clang fp contract(fast)

using blk2::flux_w;
using blk2::ldg;
using blk2::wave_lds_sync;
using quad2::v2f;
using quad2::v4f;
using quad2::v4f_g;
using strip3::lane_geo;
using strip3::LaneGeo;
using strip3e::cellv;
using strip3e::Col;
using strip3e::dpp_xor1;
using strip3e::dpp_xor8;
using strip3e::load_zcol;
using strip3e::set_cell;
using strip3e::Slot;
using strip3e::slot_of;
using strip3e::slot_sensor;
using strip3e::transpose;

#define C3_BUF 0
#define C3_PLANE 576
#define C3_PLANEA (C3_PLANE + 324)
#define C3_LDS (C3_PLANEA + 64)  // 964 floats = 3.8 KB per wave

struct Halo1 {
    float hu, hd, hc, rv;
};
template <int S>
__device__ __forceinline__ void halo_load(const BlockDesc3& bb, int lane, const float* __restrict__ u,
                                          const float* __restrict__ Cn, const Slot& sl, int32_t rid, Halo1& h) {
    h.hu = ldg(u, sl.hid);
    h.hd = ldg(u, strip3e::deeper_of<S>(bb, sl, lane, 0, sl.hid));
    h.hc = ldg(Cn, sl.hid);
    h.rv = ldg(u, (uint32_t)(rid >= 0 ? rid : bb.base));
}

// mean halo value behind boundary cell `lane` of side S and mean |halo - boundary cell|
template <int S>
__device__ __forceinline__ void side_mean(const BlockDesc3& bb, const int32_t* __restrict__ ftab,
                                          const float* __restrict__ u, int lane, float h0, float ub, float& hm, float& ha) {
    hm = h0;
    ha = fabsf(h0 - ub);
    if (bb.type[S] == SIDE_FINE) {  // wave-uniform
        const int32_t* ft = ftab + (((size_t)bb.fine * 6 + S) * 64 + lane) * 3;
        const float v1 = ldg(u, (uint32_t)ft[0]), v2 = ldg(u, (uint32_t)ft[1]), v3 = ldg(u, (uint32_t)ft[2]);
        hm = 0.25f * ((h0 + v1) + (v2 + v3));
        ha = 0.25f * ((fabsf(h0 - ub) + fabsf(v1 - ub)) + (fabsf(v2 - ub) + fabsf(v3 - ub)));
    }
}

// first differences of a column with its two (mean) halo ends: pairs of cells (j - 1, j + 4) like the column itself
struct Diffs {
    v2f dE[4];  // e[j + 1] - e[j]
    float d4;   // c[4] - c[3]
};
__device__ __forceinline__ Diffs col_diffs(const Col& c) {
    Diffs D;
#pragma unroll
    for (int j = 0; j < 4; ++j) D.dE[j] = c.e[j + 1] - c.e[j];
    D.d4 = c.e[0].y - c.e[4].x;
    return D;
}

// sensor pass along one direction (strip3e::sensor_pass with the differences given)
template <bool FIRST>
__device__ __forceinline__ void sensor_cols(const Diffs& D, float ha0, float ha1, float rh, Col& N, Col& Dn) {
    auto fold = [&](float g, float a, float Nq, float Dq) {
        const float n = fmaf(fabsf(g), rh, 1e-7f), d = fmaf(a, rh, 1e-7f);
        if constexpr (FIRST) return v2f{n, d};
        else return v2f{fmaxf(Nq * d, n * Dq), Dq * d};
    };
#pragma unroll
    for (int j = 1; j < 4; ++j) {  // cells (j - 1, j + 4)
        const v2f g = D.dE[j] - D.dE[j - 1];
        const float aLx = j == 1 ? ha0 : fabsf(D.dE[j - 1].x), aRy = j == 3 ? ha1 : fabsf(D.dE[j].y);
        const v2f lo = fold(g.x, fabsf(D.dE[j].x) + aLx, N.e[j].x, Dn.e[j].x);
        const v2f hi = fold(g.y, aRy + fabsf(D.dE[j - 1].y), N.e[j].y, Dn.e[j].y);
        N.e[j] = v2f{lo.x, hi.x};
        Dn.e[j] = v2f{lo.y, hi.y};
    }
    const v2f c3 = fold(D.d4 - D.dE[3].x, fabsf(D.d4) + fabsf(D.dE[3].x), N.e[4].x, Dn.e[4].x);  // cell 3
    const v2f c4 = fold(D.dE[0].y - D.d4, fabsf(D.dE[0].y) + fabsf(D.d4), N.e[0].y, Dn.e[0].y);  // cell 4
    N.e[4].x = c3.x;
    Dn.e[4].x = c3.y;
    N.e[0].y = c4.x;
    Dn.e[0].y = c4.y;
}

// ---- halo cell(s) of slot `lane` of side S: slope along the normal (towards +) and sensor; on a FINE side the mean
// flux through the four sub-faces instead (strip3::side_flux with the boundary cell in the lane's own registers)
template <int S>
__device__ __forceinline__ void side_eval(const BlockDesc3& bb, const LaneGeo& LG, const int32_t* __restrict__ ftab,
                                          const int32_t* __restrict__ r4tab, const float* __restrict__ u,
                                          const float* __restrict__ Cn, float* lds, int lane, const Slot& sl,
                                          const Halo1& h, int32_t rid, float ub, float Sb, float Db, float Cb, float& Sh,
                                          float& Dh, float& Ff) {
    constexpr int d = S >> 1;
    constexpr bool low = (S & 1) == 0;
    constexpr int da = d == 0 ? 1 : 0, db = d == 2 ? 1 : 2;
    const int ty = bb.type[S];
    const float qs = bb.q[S];
    const bool isC = ty == SIDE_COARSE, isF = ty == SIDE_FINE, mirror = ty == SIDE_MIRROR;
    const int t1 = lane & 7, t2 = lane >> 3;
    float* pl = lds + C3_PLANE;
    float* pA = lds + C3_PLANEA;
    const float irt = isC ? 0.5f : isF ? 2.0f : 1.0f;  // h / h_halo
    const float rn = bb.rh[d] * irt, ra = bb.rh[da] * irt, rb = bb.rh[db] * irt;
    float rv[4] = {h.rv, h.rv, h.rv, h.rv};
    if (rid < 0) {  // rim neighbour = four finer cells (few lanes)
        const int32_t* r4 = r4tab + (size_t)(-rid - 1) * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) rv[k] = ldg(u, (uint32_t)r4[k]);
    }
    const float rmean = 0.25f * ((rv[0] + rv[1]) + (rv[2] + rv[3]));
    Ff = 0.0f;
    if (!isF) {  // wave-uniform
        pl[LG.pc] = h.hu;
        if (lane < 32) pl[LG.rpos8] = rmean;
        wave_lds_sync();
        if (lane < 32) {
            const float ha = pl[LG.radj8];
            pA[lane] = 0.25f * ((fabsf(rv[0] - ha) + fabsf(rv[1] - ha)) + (fabsf(rv[2] - ha) + fabsf(rv[3] - ha)));
        }
        wave_lds_sync();
        float mean;
        if (!isC) {  // SAME / MIRROR
            mean = ub;
            Dh = slot_sensor<false>(pl, pA, h.hu, h.hd, ub, ub, ub, ub, LG.pc - 1, LG.pc + 1, LG.pc - 18, LG.pc + 18,
                                    t1 == 0, t1 == 7, t2 == 0, t2 == 7, t1, t2, 8, rn, ra, rb);
        } else {     // COARSE: the halo cell spans the 2 x 2 group of slots t, t ^ 1, t ^ 8, t ^ 9
            const int c1 = LG.b1, c2 = LG.b2;
            const int la = LG.pc - 1 - c1, ha = LG.pc + 2 - c1, lb = LG.pc - 18 - 18 * c2, hb = LG.pc + 36 - 18 * c2;
            const float m1 = dpp_xor1(ub), m2 = dpp_xor8(ub), m3 = dpp_xor8(m1);
            mean = 0.25f * ((ub + m1) + (m2 + m3));
            Dh = slot_sensor<true>(pl, pA, h.hu, h.hd, ub, m1, m2, m3, la, ha, lb, hb, t1 <= 1, t1 >= 6, t2 <= 1, t2 >= 6,
                                   t1, t2, 8, rn, ra, rb);
        }
        const float x = (1.0f - qs) * (mean - h.hu) - 0.5f * (h.hd - h.hu);
        Sh = mirror ? Sb : (low ? x : -x);
        Dh = mirror ? Db : Dh;
    } else {  // the 2 x 2 finer cells behind this boundary cell, one after the other (rolled loop)
        const int32_t* ft = ftab + (((size_t)bb.fine * 6 + S) * 64 + lane) * 3;
        pl[(2 * t1 + 1) + 18 * (2 * t2 + 1)] = h.hu;
#pragma unroll 1
        for (int k = 1; k < 4; ++k)
            pl[(2 * t1 + (k & 1) + 1) + 18 * (2 * t2 + (k >> 1) + 1)] = ldg(u, (uint32_t)ft[k - 1]);
        pl[LG.rpos16] = rmean;
        wave_lds_sync();
        {
            const float ha = pl[LG.radj16];
            pA[lane] = 0.25f * ((fabsf(rv[0] - ha) + fabsf(rv[1] - ha)) + (fabsf(rv[2] - ha) + fabsf(rv[3] - ha)));
        }
        wave_lds_sync();
        float acc = 0.0f;
#pragma unroll 1
        for (int k = 0; k < 4; ++k) {
            float hk = h.hu, hdk = h.hd, hck = h.hc;
            if (k > 0) {
                const uint32_t c = (uint32_t)ft[k - 1];
                hk = ldg(u, c);
                hdk = ldg(u, strip3e::deeper_of<S>(bb, sl, lane, k, c));
                hck = ldg(Cn, c);
            }
            const int f1 = 2 * t1 + (k & 1), f2 = 2 * t2 + (k >> 1);
            const float dhk = slot_sensor<false>(pl, pA, hk, hdk, ub, ub, ub, ub, f1 + 18 * (f2 + 1), f1 + 2 + 18 * (f2 + 1),
                                                 (f1 + 1) + 18 * f2, (f1 + 1) + 18 * (f2 + 2), f1 == 0, f1 == 15, f2 == 0,
                                                 f2 == 15, f1, f2, 16, rn, ra, rb);
            const float x = (1.0f - qs) * (ub - hk) - 0.5f * (hdk - hk);
            const float shk = low ? x : -x;
            acc += low ? flux_w(hk, ub, shk, Sb, dhk, Db, hck, Cb, 1.0f - qs) : flux_w(ub, hk, Sb, shk, Db, dhk, Cb, hck, qs);
        }
        Ff = 0.25f * acc;
        Sh = Sb;  // finite stand-ins: the packed evaluation of this face is replaced by Ff
        Dh = Db;
    }
}

// ---- flux pass along D: R (op)= -(F_high - F_low) / h for the 8 cells of the lane's column.
// MODE 0: R = ..., 1: R -= ..., 2: R - ... goes to global memory (z-columns: coalesced dwords)
template <int D, int MODE, class Prefetch>
__device__ __forceinline__ void flux_pass(const BlockDesc3& bb, const LaneGeo& LG, const int32_t* __restrict__ ftab,
                                          const int32_t* __restrict__ r4tab, const float* __restrict__ u,
                                          const float* __restrict__ Cn, float* lds, int lane, Col& uc, Col& Cc, Col& Dc,
                                          const Diffs& Df, const Slot* slots, const int32_t* rids, const Halo1& h0,
                                          const Halo1& h1, Prefetch&& prefetch, Col& R, float* __restrict__ ud) {
    constexpr int S0 = 2 * D, S1 = 2 * D + 1;
    const float rh = bb.rh[D];
    const float qlo = bb.q[S0], qhi = bb.q[S1];
    const bool isF0 = bb.type[S0] == SIDE_FINE, isF1 = bb.type[S1] == SIDE_FINE;
    // undivided slopes of the column's cells (column ends hold the MEAN halo values here)
    Col S;
    S.e[1] = 0.5f * Df.dE[1] + v2f{qlo, 0.5f} * Df.dE[0];
    S.e[2] = 0.5f * Df.dE[2] + 0.5f * Df.dE[1];
    S.e[3] = v2f{0.5f, qhi} * Df.dE[3] + 0.5f * Df.dE[2];
    S.e[4].x = 0.5f * Df.d4 + 0.5f * Df.dE[3].x;
    S.e[0].y = 0.5f * Df.dE[0].y + 0.5f * Df.d4;
    // halo cells
    float Sh0, Dh0, Ff0, Sh1, Dh1, Ff1;
    side_eval<S0>(bb, LG, ftab, r4tab, u, Cn, lds, lane, slots[S0], h0, rids[S0], uc.e[1].x, S.e[1].x, Dc.e[1].x, Cc.e[1].x,
                  Sh0, Dh0, Ff0);
    side_eval<S1>(bb, LG, ftab, r4tab, u, Cn, lds, lane, slots[S1], h1, rids[S1], uc.e[3].y, S.e[3].y, Dc.e[3].y, Cc.e[3].y,
                  Sh1, Dh1, Ff1);
    __builtin_amdgcn_sched_barrier(0);
    prefetch();
    __builtin_amdgcn_sched_barrier(0);
    S.e[0].x = Sh0;
    S.e[4].y = Sh1;
    Dc.e[0].x = Dh0;
    Dc.e[4].y = Dh1;
    uc.e[0].x = h0.hu;  // the halo cells themselves for the faces (FINE sides: replaced below)
    uc.e[4].y = h1.hu;
    Cc.e[0].x = h0.hc;
    Cc.e[4].y = h1.hc;
    // faces (j, j + 5), two at a time; face 4 alone
    v2f F[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const v2f wa = j == 0 ? v2f{1.0f - qlo, 0.5f} : j == 3 ? v2f{0.5f, qhi} : v2f{0.5f, 0.5f};
        F[j] = blk2::flux_w2(uc.e[j], uc.e[j + 1], S.e[j], S.e[j + 1], Dc.e[j], Dc.e[j + 1], Cc.e[j], Cc.e[j + 1], wa);
    }
    if (isF0) F[0].x = Ff0;
    if (isF1) F[3].y = Ff1;
    const float F4 = flux_w(uc.e[4].x, uc.e[0].y, S.e[4].x, S.e[0].y, Dc.e[4].x, Dc.e[0].y, Cc.e[4].x, Cc.e[0].y, 0.5f);
    // Green-Gauss: cells (j - 1, j + 4): faces (j - 1, j + 4) below, (j, j + 5) above
    Col dF;
#pragma unroll
    for (int j = 1; j < 4; ++j) dF.e[j] = F[j] - F[j - 1];
    dF.e[4].x = F4 - F[3].x;  // cell 3
    dF.e[0].y = F[0].y - F4;  // cell 4
    if constexpr (MODE == 0) {
#pragma unroll
        for (int j = 1; j < 4; ++j) R.e[j] = -(dF.e[j] * rh);
        R.e[4].x = -(dF.e[4].x * rh);
        R.e[0].y = -(dF.e[0].y * rh);
    } else {
#pragma unroll
        for (int j = 1; j < 4; ++j) R.e[j] -= dF.e[j] * rh;
        R.e[4].x -= dF.e[4].x * rh;
        R.e[0].y -= dF.e[0].y * rh;
    }
}

template <int I = 0>
__device__ __forceinline__ void store_zcol(float* __restrict__ p, const Col& c) {
    if constexpr (I < 8) {
#ifdef C3_NO_NT_STORE   // (A/B)
        p[64 * I] = cellv<I>(c);
#else   // the residual is not read again by this sweep: a store that does not allocate in L2 leaves it to the halo lines
        __builtin_nontemporal_store(cellv<I>(c), p + 64 * I);
#endif
        store_zcol<I + 1>(p, c);
    }
}
template <int I = 0>
__device__ __forceinline__ void load_ycol(const float* __restrict__ p, Col& c) {
    if constexpr (I < 8) {
        set_cell<I>(c, p[8 * I]);
        load_ycol<I + 1>(p, c);
    }
}

__device__ __forceinline__ void sweep_cols(const BlockDesc3* __restrict__ blocks, const int32_t* __restrict__ htab,
                                           const int32_t* __restrict__ ftab, const int32_t* __restrict__ rtab,
                                           const int32_t* __restrict__ r4tab, int32_t blk, const float* __restrict__ u,
                                           const float* __restrict__ C, uint32_t ldc, float* __restrict__ ud, float* lds,
                                           int lane, const int32_t* __restrict__ dtab = nullptr) {
    const BlockDesc3 bb = blocks[blk];
    const int ta = lane & 7, tb = lane >> 3;
    float* buf = lds + C3_BUF;
    const float* Cy = C + ldc;
    const float* Cz = C + 2 * (size_t)ldc;
    // ---- loads, in the order of their use.  Halo cell ids first (a table load on FINE sides only: the compiler waits for
    // every load in flight before it uses them), then u as z-columns and the halo values of the sensor, the x sides, the
    // velocity components each in the layout of its pass
    Slot slots[6];
    slots[0] = slot_of<0>(bb, htab, blk, lane, dtab);
    slots[1] = slot_of<1>(bb, htab, blk, lane, dtab);
    slots[2] = slot_of<2>(bb, htab, blk, lane, dtab);
    slots[3] = slot_of<3>(bb, htab, blk, lane, dtab);
    slots[4] = slot_of<4>(bb, htab, blk, lane, dtab);
    slots[5] = slot_of<5>(bb, htab, blk, lane, dtab);
    __builtin_amdgcn_sched_barrier(0);
    int32_t rids[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) rids[s] = rtab[(size_t)blk * 384 + s * 64 + lane];
    Col uc;
    load_zcol(u + (uint32_t)bb.base + lane, uc);
    Halo1 hA0, hA1, hB0, hB1, hC0, hC1;   // x, y, z sides
    hC0.hu = ldg(u, slots[4].hid);
    hC1.hu = ldg(u, slots[5].hid);
    hB0.hu = ldg(u, slots[2].hid);
    hB1.hu = ldg(u, slots[3].hid);
    halo_load<0>(bb, lane, u, C, slots[0], rids[0], hA0);
    halo_load<1>(bb, lane, u, C, slots[1], rids[1], hA1);
    __builtin_amdgcn_sched_barrier(0);
    Col Cc;
    {
        const float* p = C + (uint32_t)bb.base + 8u * (uint32_t)lane;
        const v4f lo = *(const v4f_g*)p, hi = *(const v4f_g*)(p + 4);
        Cc.e[1].x = lo.x; Cc.e[2].x = lo.y; Cc.e[3].x = lo.z; Cc.e[4].x = lo.w;
        Cc.e[0].y = hi.x; Cc.e[1].y = hi.y; Cc.e[2].y = hi.z; Cc.e[3].y = hi.w;
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- sensor: z, y, x
    Col N, Dn, Dc;
    float hm0, ha0, hm1, ha1;
    side_mean<4>(bb, ftab, u, lane, hC0.hu, cellv<0>(uc), hm0, ha0);
    side_mean<5>(bb, ftab, u, lane, hC1.hu, cellv<7>(uc), hm1, ha1);
    uc.e[0].x = hm0;
    uc.e[4].y = hm1;
    sensor_cols<true>(col_diffs(uc), ha0, ha1, bb.rh[2], N, Dn);
    transpose<2, 1>(buf, ta, tb, uc);
    transpose<2, 1>(buf, ta, tb, N);
    transpose<2, 1>(buf, ta, tb, Dn);
    side_mean<2>(bb, ftab, u, lane, hB0.hu, cellv<0>(uc), hm0, ha0);
    side_mean<3>(bb, ftab, u, lane, hB1.hu, cellv<7>(uc), hm1, ha1);
    uc.e[0].x = hm0;
    uc.e[4].y = hm1;
    sensor_cols<false>(col_diffs(uc), ha0, ha1, bb.rh[1], N, Dn);
    transpose<1, 0>(buf, ta, tb, uc);
    transpose<1, 0>(buf, ta, tb, N);
    transpose<1, 0>(buf, ta, tb, Dn);
    side_mean<0>(bb, ftab, u, lane, hA0.hu, cellv<0>(uc), hm0, ha0);
    side_mean<1>(bb, ftab, u, lane, hA1.hu, cellv<7>(uc), hm1, ha1);
    uc.e[0].x = hm0;
    uc.e[4].y = hm1;
    const Diffs Dx = col_diffs(uc);
    sensor_cols<false>(Dx, ha0, ha1, bb.rh[0], N, Dn);
#pragma unroll
    for (int j = 1; j < 4; ++j)
        Dc.e[j] = v2f{fmaxf(N.e[j].x * __builtin_amdgcn_rcpf(Dn.e[j].x), 1e-7f),
                      fmaxf(N.e[j].y * __builtin_amdgcn_rcpf(Dn.e[j].y), 1e-7f)};
    Dc.e[4].x = fmaxf(N.e[4].x * __builtin_amdgcn_rcpf(Dn.e[4].x), 1e-7f);
    Dc.e[0].y = fmaxf(N.e[0].y * __builtin_amdgcn_rcpf(Dn.e[0].y), 1e-7f);
    // ---- fluxes: x, y, z
    const LaneGeo LG = lane_geo(lane);
    Col R, Cn2;
    flux_pass<0, 0>(bb, LG, ftab, r4tab, u, C, lds, lane, uc, Cc, Dc, Dx, slots, rids, hA0, hA1, [&]() {
        halo_load<2>(bb, lane, u, Cy, slots[2], rids[2], hB0);
        halo_load<3>(bb, lane, u, Cy, slots[3], rids[3], hB1);
    }, R, ud);
    load_ycol(Cy + (uint32_t)bb.base + (uint32_t)(ta + 64 * tb), Cn2);  // y-columns: (ta, tb) = (x, z)
    transpose<0, 1>(buf, ta, tb, uc);
    transpose<0, 1>(buf, ta, tb, Dc);
    transpose<0, 1>(buf, ta, tb, R);
    side_mean<2>(bb, ftab, u, lane, hB0.hu, cellv<0>(uc), hm0, ha0);
    side_mean<3>(bb, ftab, u, lane, hB1.hu, cellv<7>(uc), hm1, ha1);
    uc.e[0].x = hm0;
    uc.e[4].y = hm1;
    flux_pass<1, 1>(bb, LG, ftab, r4tab, u, Cy, lds, lane, uc, Cn2, Dc, col_diffs(uc), slots, rids, hB0, hB1, [&]() {
        halo_load<4>(bb, lane, u, Cz, slots[4], rids[4], hC0);
        halo_load<5>(bb, lane, u, Cz, slots[5], rids[5], hC1);
    }, R, ud);
    load_zcol(Cz + (uint32_t)bb.base + lane, Cc);
    transpose<1, 2>(buf, ta, tb, uc);
    transpose<1, 2>(buf, ta, tb, Dc);
    transpose<1, 2>(buf, ta, tb, R);
    side_mean<4>(bb, ftab, u, lane, hC0.hu, cellv<0>(uc), hm0, ha0);
    side_mean<5>(bb, ftab, u, lane, hC1.hu, cellv<7>(uc), hm1, ha1);
    uc.e[0].x = hm0;
    uc.e[4].y = hm1;
    flux_pass<2, 1>(bb, LG, ftab, r4tab, u, Cz, lds, lane, uc, Cc, Dc, col_diffs(uc), slots, rids, hC0, hC1, []() {}, R, ud);
    store_zcol(ud + (uint32_t)bb.base + lane, R);
}

#pragma clang fp contract(off)

}  // namespace cols3
