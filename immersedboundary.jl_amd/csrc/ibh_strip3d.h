// Strip sweep (3-D scalar advection-JST-MUSCL residual, test/advection.jl:67-83 on an octree partition): ONE wavefront
// sweeps an 8x8x8 block, every lane an x-strip of 8 cells -- the 3-D counterpart of the quad sweep (ibh_quad2d.h).
// Same arithmetic and tables as blk3::sweep_adv (ibh_block3d.h: thread per cell, 360+ vector instructions per cell); here
// the x neighbours of a cell are registers, the y neighbours DPP row shifts, the z neighbours ds_bpermute, the
// descriptor / index arithmetic is paid once per 8 cells and the interior fluxes run four at a time on packed math.
//
// Lane L: y = L & 7, z = L >> 3; own cells x = 0..7 at block offset 8 L + x (two float4 per field).
// Halo slots: lane L also owns slot t = L of each of the six sides (boundary cell t1 = L & 7, t2 = L >> 3 in the side's
// tangential coordinates): it loads the halo cell(s) behind it -- one, or the 2 x 2 finer cells of a FINE side -- computes
// their slope along the normal and their sensor, the flux of the sub-face(s), and hands the boundary cell's lane
//   (a) the mean halo value and mean |halo - boundary cell| (for the cell's own slopes and sensor: Hm, Ha), and
//   (b) the mean sub-face flux (ex),
// through LDS rows indexed by t.  The boundary cells' own data reaches the slot lanes the same way (face rows).
// Lateral neighbours of the halo cells: the side's plane ((n + 2)^2, n = 8 or 16), border from the rim table, ONE plane
// buffer reused side after side (a wavefront's LDS operations execute in order).
#pragma once
#include "ibh_block3d.h"
#include "ibh_quad2d.h"

namespace strip3 {

#pragma clang fp contract(fast)

using blk2::flux_w;
using blk2::ldg;
using blk2::wave_lds_sync;
using blk3::halo_cell3s;
using blk3::jst_ratio;
using quad2::bperm;
// lane i <- lane i -/+ 1 in rows of 16, zero at the row's end: one instruction (the forms with an `old` value cost a
// copy more); the lanes on a block face (y = 0 / 7) replace or ignore what they get
__device__ __forceinline__ float dpp_shr1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_shl1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x101, 0xf, 0xf, true));
}
using quad2::flux_half4;
using quad2::v4f;
using quad2::v4f_g;

// LDS per wave (floats)
#define S3_FACE 0                     // [4: U, S_normal, D, C_normal][6 sides][64] boundary cells by slot
#define S3_HM (S3_FACE + 4 * 384)     // [6][64] mean halo value behind boundary cell t; later the edge fluxes
#define S3_HA (S3_HM + 384)           // [6][64] mean |halo - boundary cell|
#define S3_PLANE S3_HA                // [18 x 18] over Ha, which is dead after the slope phase (a wave's LDS operations
#define S3_PLANEA (S3_PLANE + 324)    // [64]      execute in order); rim neighbours: mean |difference| to the next halo cell
#define S3_LDS (S3_PLANEA + 64)       // 2 308 floats = 9.0 KB per wave

// one quantity of this lane's strip into the face rows of the sides the strip touches
template <int Q>
__device__ __forceinline__ void put_faces_x(float* lds, int lane, float x0, float x7) {
    float* f = lds + S3_FACE + Q * 384;
    f[0 * 64 + lane] = x0;
    f[1 * 64 + lane] = x7;
}
template <int Q>
__device__ __forceinline__ void put_faces_y(float* lds, int lane, const float* Y) {
    float* f = lds + S3_FACE + Q * 384;
    const int y = lane & 7, z = lane >> 3;
    if (y == 0 || y == 7) {
        float* r = f + (y == 0 ? 2 : 3) * 64 + 8 * z;
        *(v4f*)r = v4f{Y[0], Y[1], Y[2], Y[3]};
        *(v4f*)(r + 4) = v4f{Y[4], Y[5], Y[6], Y[7]};
    }
}
template <int Q>
__device__ __forceinline__ void put_faces_z(float* lds, int lane, const float* Z) {
    float* f = lds + S3_FACE + Q * 384;
    const int y = lane & 7, z = lane >> 3;
    if (z == 0 || z == 7) {
        float* r = f + (z == 0 ? 4 : 5) * 64 + 8 * y;
        *(v4f*)r = v4f{Z[0], Z[1], Z[2], Z[3]};
        *(v4f*)(r + 4) = v4f{Z[4], Z[5], Z[6], Z[7]};
    }
}
template <int Q>
__device__ __forceinline__ void put_faces(float* lds, int lane, const float* x0, const float* x7, const float* Y,
                                          const float* Z) {
    put_faces_x<Q>(lds, lane, *x0, *x7);
    put_faces_y<Q>(lds, lane, Y);
    put_faces_z<Q>(lds, lane, Z);
}

// (a): mean halo value and mean |halo - boundary cell| of slot `lane` of side S
template <int S>
__device__ __forceinline__ void side_mean(const BlockDesc3& bb, const int32_t* __restrict__ ftab,
                                          const float* __restrict__ u, float* lds, int lane, float hu0) {
    const float ub = lds[S3_FACE + S * 64 + lane];
    float vm = hu0, am = fabsf(hu0 - ub);
    if (bb.type[S] == SIDE_FINE) {  // wave-uniform
        const int32_t* ft = ftab + (((size_t)bb.fine * 6 + S) * 64 + lane) * 3;
        const float v1 = ldg(u, (uint32_t)ft[0]), v2 = ldg(u, (uint32_t)ft[1]), v3 = ldg(u, (uint32_t)ft[2]);
        vm = 0.25f * ((hu0 + v1) + (v2 + v3));
        am = 0.25f * ((fabsf(hu0 - ub) + fabsf(v1 - ub)) + (fabsf(v2 - ub) + fabsf(v3 - ub)));
    }
    lds[S3_HM + S * 64 + lane] = vm;
    lds[S3_HA + S * 64 + lane] = am;
}

// lane-only geometry of the slots, computed once per wave
struct LaneGeo {
    int pc;              // this slot's position in an 8 x 8 plane (pitch 18)
    int b1, b2;          // t1 & 1, t2 & 1: position inside the 2 x 2 group in front of a coarse cell
    int rpos8, radj8;    // lanes < 32: plane position of this lane's rim cell and of the halo cell next to it (n = 8)
    int rpos16, radj16;  // the same for the 16 x 16 plane of a FINE side (all lanes)
};
__device__ __forceinline__ LaneGeo lane_geo(int lane) {
    LaneGeo L;
    L.pc = ((lane & 7) + 1) + 18 * ((lane >> 3) + 1);
    L.b1 = lane & 1;
    L.b2 = (lane >> 3) & 1;
    auto rim = [](int r, int i, int n, int& pos, int& adj) {
        const int p1 = r == 0 ? 0 : r == 1 ? n + 1 : i + 1, p2 = r == 2 ? 0 : r == 3 ? n + 1 : i + 1;
        const int a1 = r == 0 ? 1 : r == 1 ? n : i + 1, a2 = r == 2 ? 1 : r == 3 ? n : i + 1;
        pos = p1 + 18 * p2;
        adj = a1 + 18 * a2;
    };
    rim((lane >> 3) & 3, lane & 7, 8, L.rpos8, L.radj8);
    rim(lane >> 4, lane & 15, 16, L.rpos16, L.radj16);
    return L;
}

// max of three JST ratios n_i / d_i and 1e-7 with ONE reciprocal (v_rcp_f32 is a quarter-rate instruction)
__device__ __forceinline__ float jst_max3(float g1, float a1, float r1, float g2, float a2, float r2, float g3, float a3,
                                          float r3) {
    const float n1 = fmaf(fabsf(g1), r1, 1e-7f), d1 = fmaf(a1, r1, 1e-7f);
    const float n2 = fmaf(fabsf(g2), r2, 1e-7f), d2 = fmaf(a2, r2, 1e-7f);
    const float n3 = fmaf(fabsf(g3), r3, 1e-7f), d3 = fmaf(a3, r3, 1e-7f);
    const float d12 = d1 * d2;
    const float m = fmaxf(fmaxf(n1 * d2, n2 * d1) * d3, n3 * d12);
    return fmaxf(m * __builtin_amdgcn_rcpf(d12 * d3), 1e-7f);
}

// slope + sensor of ONE halo cell and the flux of its sub-face.  GROUP: the halo cell is coarser, four of the block's
// cells (m0..m3) face it; otherwise one (ub)
template <bool GROUP>
__device__ __forceinline__ float slot_eval(const float* pl, const float* pA, float h, float hde, float hc, float m0,
                                           float m1, float m2, float m3, float ub, float Sb, float Db, float Cb, int la,
                                           int ha, int lb, int hb, bool ra0, bool ra1, bool rb0, bool rb1, int ia, int ib,
                                           int n, float qs, float rn, float ra, float rb, bool low, bool mirror) {
    using quad2::lds_read;
    float din, ain;
    if constexpr (GROUP) {
        din = 0.25f * ((m0 + m1) + (m2 + m3)) - h;
        ain = 0.25f * ((fabsf(m0 - h) + fabsf(m1 - h)) + (fabsf(m2 - h) + fabsf(m3 - h)));
    } else {
        din = ub - h;
        ain = fabsf(din);
    }
    const float dde = hde - h;
    const float x = (1.0f - qs) * din - 0.5f * dde;
    // eight LDS reads every lane performs, in flight together (ONE asm statement: the compiler may neither sink one of
    // them into a divergent branch of the selects below nor wait for them one by one)
    float pa0 = pl[la], pa1 = pl[ha], pb0 = pl[lb], pb1 = pl[hb];
    float A0 = pA[ib], A1 = pA[n + ib], B0 = pA[2 * n + ia], B1 = pA[3 * n + ia];
    asm volatile("" : "+v"(pa0), "+v"(pa1), "+v"(pb0), "+v"(pb1), "+v"(A0), "+v"(A1), "+v"(B0), "+v"(B1));
    const float ea0 = pa0 - h, ea1 = pa1 - h, eb0 = pb0 - h, eb1 = pb1 - h;
    const float aa0 = ra0 ? A0 : fabsf(ea0), aa1 = ra1 ? A1 : fabsf(ea1);
    const float ab0 = rb0 ? B0 : fabsf(eb0), ab1 = rb1 ? B1 : fabsf(eb1);
    float dh = jst_max3(din + dde, ain + fabsf(dde), rn, ea0 + ea1, aa0 + aa1, ra, eb0 + eb1, ab0 + ab1, rb);
    const float sh = mirror ? Sb : (low ? x : -x);
    dh = mirror ? Db : dh;
    // halo cell <-> boundary cell: the cell towards - is the owner
    return low ? flux_w(h, ub, sh, Sb, dh, Db, hc, Cb, 1.0f - qs) : flux_w(ub, h, Sb, sh, Db, dh, Cb, hc, qs);
}

// (b): slope + sensor of the halo cell(s) of slot `lane` of side S, flux of the sub-face(s), mean into ex
template <int S>
__device__ __forceinline__ void side_flux(const BlockDesc3& bb, const LaneGeo& LG, const int32_t* __restrict__ ftab,
                                          const int32_t* __restrict__ r4tab, const float* __restrict__ u,
                                          const float* __restrict__ Cn, float* lds, int lane, float hu0, float hde0,
                                          float hc0, int32_t rid, float rv0) {
    using quad2::lds_read;
    constexpr int d = S >> 1;
    constexpr bool low = (S & 1) == 0;
    constexpr int sd = d == 0 ? 1 : d == 1 ? 8 : 64;
    constexpr int da = d == 0 ? 1 : 0, db = d == 2 ? 1 : 2;
    const int ty = bb.type[S];
    const float qs = bb.q[S];
#ifdef S3_COUNT_NO_FINE  // (instruction counts of the path without FINE sides: scripts/isa_count.py)
    const bool isC = ty == SIDE_COARSE, isF = false, mirror = ty == SIDE_MIRROR;
#else
    const bool isC = ty == SIDE_COARSE, isF = ty == SIDE_FINE, mirror = ty == SIDE_MIRROR;
#endif
    const int t1 = lane & 7, t2 = lane >> 3;
    float* pl = lds + S3_PLANE;
    float* pA = lds + S3_PLANEA;
    const float* fU = lds + S3_FACE + S * 64;
    const float irt = isC ? 0.5f : isF ? 2.0f : 1.0f;  // h / h_halo
    const float rn = bb.rh[d] * irt, ra = bb.rh[da] * irt, rb = bb.rh[db] * irt;
    float rv[4] = {rv0, rv0, rv0, rv0};
    if (rid < 0) {  // rim neighbour = four finer cells (few lanes)
        const int32_t* r4 = r4tab + (size_t)(-rid - 1) * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) rv[k] = ldg(u, (uint32_t)r4[k]);
    }
    const float rmean = 0.25f * ((rv[0] + rv[1]) + (rv[2] + rv[3]));
    const float ub = lds_read(fU + lane), Sb = lds_read(fU + 384 + lane), Db = lds_read(fU + 768 + lane),
                Cb = lds_read(fU + 1152 + lane);
    float out;
    if (!isF) {  // wave-uniform
        pl[LG.pc] = hu0;
        if (lane < 32) pl[LG.rpos8] = rmean;
        wave_lds_sync();
        if (lane < 32) {
            const float ha = pl[LG.radj8];
            pA[lane] = 0.25f * ((fabsf(rv[0] - ha) + fabsf(rv[1] - ha)) + (fabsf(rv[2] - ha) + fabsf(rv[3] - ha)));
        }
        wave_lds_sync();
        if (!isC) {  // SAME / MIRROR: the lateral neighbours are the next slots, one cell of the block in front
            out = slot_eval<false>(pl, pA, hu0, hde0, hc0, ub, ub, ub, ub, ub, Sb, Db, Cb, LG.pc - 1, LG.pc + 1, LG.pc - 18,
                                   LG.pc + 18, t1 == 0, t1 == 7, t2 == 0, t2 == 7, t1, t2, 8, qs, rn, ra, rb, low, mirror);
        } else {     // COARSE: the halo cell spans a 2 x 2 group of slots
            const int c1 = LG.b1, c2 = LG.b2;
            const int la = LG.pc - 1 - c1, ha = LG.pc + 2 - c1, lb = LG.pc - 18 - 18 * c2, hb = LG.pc + 36 - 18 * c2;
            const int g0 = lane - c1 - 8 * c2;
            const float m0 = lds_read(fU + g0), m1 = lds_read(fU + g0 + 1), m2 = lds_read(fU + g0 + 8),
                        m3 = lds_read(fU + g0 + 9);
            out = slot_eval<true>(pl, pA, hu0, hde0, hc0, m0, m1, m2, m3, ub, Sb, Db, Cb, la, ha, lb, hb, t1 <= 1, t1 >= 6,
                                  t2 <= 1, t2 >= 6, t1, t2, 8, qs, rn, ra, rb, low, false);
        }
    } else {  // the 2 x 2 finer cells behind this boundary cell, one after the other (rolled loops: few registers)
        const int32_t* ft = ftab + (((size_t)bb.fine * 6 + S) * 64 + lane) * 3;
        const int dd = low ? -sd : sd;
        pl[(2 * t1 + 1) + 18 * (2 * t2 + 1)] = hu0;
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
            float hk = hu0, hdk = hde0, hck = hc0;
            if (k > 0) {
                const uint32_t c = (uint32_t)ft[k - 1];
                hk = ldg(u, c);
                hdk = ldg(u, (uint32_t)((int)c + dd));
                hck = ldg(Cn, c);
            }
            const int f1 = 2 * t1 + (k & 1), f2 = 2 * t2 + (k >> 1);
            acc += slot_eval<false>(pl, pA, hk, hdk, hck, ub, ub, ub, ub, ub, Sb, Db, Cb, f1 + 18 * (f2 + 1),
                                    f1 + 2 + 18 * (f2 + 1), (f1 + 1) + 18 * f2, (f1 + 1) + 18 * (f2 + 2), f1 == 0, f1 == 15,
                                    f2 == 0, f2 == 15, f1, f2, 16, qs, rn, ra, rb, low, false);
        }
        out = 0.25f * acc;
    }
    lds[S3_HM + S * 64 + lane] = out;
}

__device__ __forceinline__ void sweep_strip(const BlockDesc3* __restrict__ blocks, const int32_t* __restrict__ htab,
                                            const int32_t* __restrict__ ftab, const int32_t* __restrict__ rtab,
                                            const int32_t* __restrict__ r4tab, int32_t blk, const float* __restrict__ u,
                                            const float* __restrict__ C, uint32_t ldc, float* __restrict__ ud, float* lds,
                                            int lane) {
    const int y = lane & 7, z = lane >> 3;
    const bool y0 = y == 0, y7 = y == 7, z0 = z == 0, z7 = z == 7;
    // ---- loads: rim ids (need the block index only), descriptor, own strip, halo slots, rim values
    const int32_t* rrow = rtab + (size_t)blk * 384;
    int32_t rid[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) rid[s] = rrow[s * 64 + lane];
    const BlockDesc3 bb = blocks[blk];
    const uint32_t a0 = (uint32_t)bb.base + 8u * (uint32_t)lane;
    auto ld8 = [&](const float* p, float* o) {
        const v4f lo = *(const v4f_g*)(p + a0), hi = *(const v4f_g*)(p + a0 + 4);
        o[0] = lo.x; o[1] = lo.y; o[2] = lo.z; o[3] = lo.w;
        o[4] = hi.x; o[5] = hi.y; o[6] = hi.z; o[7] = hi.w;
    };
    float U[8];
    ld8(u, U);
    uint32_t hid[6];
    hid[0] = halo_cell3s<0>(bb, htab, blk, lane);
    hid[1] = halo_cell3s<1>(bb, htab, blk, lane);
    hid[2] = halo_cell3s<2>(bb, htab, blk, lane);
    hid[3] = halo_cell3s<3>(bb, htab, blk, lane);
    hid[4] = halo_cell3s<4>(bb, htab, blk, lane);
    hid[5] = halo_cell3s<5>(bb, htab, blk, lane);
    {
        float hu[6];
#pragma unroll
        for (int s = 0; s < 6; ++s) hu[s] = ldg(u, hid[s]);
        // ---- (a) boundary cells to the slot lanes, mean halo values back
        put_faces<0>(lds, lane, &U[0], &U[7], U, U);
        wave_lds_sync();
        side_mean<0>(bb, ftab, u, lds, lane, hu[0]);
        side_mean<1>(bb, ftab, u, lds, lane, hu[1]);
        side_mean<2>(bb, ftab, u, lds, lane, hu[2]);
        side_mean<3>(bb, ftab, u, lds, lane, hu[3]);
        side_mean<4>(bb, ftab, u, lds, lane, hu[4]);
        side_mean<5>(bb, ftab, u, lds, lane, hu[5]);
        wave_lds_sync();
    }
    // the velocity strips are requested one direction ahead of the fluxes that use them (scheduling barriers: a strip
    // in flight holds 8 registers): x now, landing during the slope phase
    __builtin_amdgcn_sched_barrier(0);
    float CX[8], CY[8], CZ[8];
    ld8(C, CX);
    __builtin_amdgcn_sched_barrier(0);

    // ---- own cells: undivided slopes and the sensor
    const float rhx = bb.rh[0], rhy = bb.rh[1], rhz = bb.rh[2];
    float SX[8], SY[8], SZ[8], D[8];
    {
        const float* HM = lds + S3_HM;
        const float* HA = lds + S3_HA;
        const float hmx0 = HM[lane], hmx1 = HM[64 + lane], hax0 = HA[lane], hax1 = HA[64 + lane];
        // a lane is on at most one y face and one z face of the block: one row of each serves both sides
        const int ry = (y7 ? 192 : 128) + 8 * z, rz = (z7 ? 320 : 256) + 8 * y;
        const float qyB = y0 ? bb.q[2] : 0.5f, qyT = y7 ? bb.q[3] : 0.5f;
        const float qzB = z0 ? bb.q[4] : 0.5f, qzT = z7 ? bb.q[5] : 0.5f;
        const int dn = (lane - 8) << 2, up = (lane + 8) << 2;
        // FINEBLK: the block has a side facing finer cells; only then the mean |difference| of a boundary cell to the
        // cells behind it (Ha) is not |mean value - cell| and has to be fetched
        auto slopes = [&](auto fineblk) {
            constexpr bool FINEBLK = decltype(fineblk)::value;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                // four cells' worth of shifted neighbours at a time (the asm pins the order at IR level)
                asm volatile("" : "+v"(U[4 * h]), "+v"(U[4 * h + 1]), "+v"(U[4 * h + 2]), "+v"(U[4 * h + 3])::"memory");
                __builtin_amdgcn_sched_barrier(0);
                v4f hY = *(const v4f*)(HM + ry + 4 * h), hZ = *(const v4f*)(HM + rz + 4 * h);
                v4f gY = hY, gZ = hZ;
                if constexpr (FINEBLK) {
                    gY = *(const v4f*)(HA + ry + 4 * h);
                    gZ = *(const v4f*)(HA + rz + 4 * h);
                }
                asm volatile("" : "+v"(hY), "+v"(gY), "+v"(hZ), "+v"(gZ));
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int x = 4 * h + k;
                    const float uc = U[x];
                    // x: registers
                    const float vL = x == 0 ? hmx0 : U[x > 0 ? x - 1 : 0], vR = x == 7 ? hmx1 : U[x < 7 ? x + 1 : 7];
                    const float dRx = vR - uc, dLx = uc - vL;
                    const float aLx = (FINEBLK && x == 0) ? hax0 : fabsf(dLx), aRx = (FINEBLK && x == 7) ? hax1 : fabsf(dRx);
                    SX[x] = (x == 7 ? bb.q[1] : 0.5f) * dRx + (x == 0 ? bb.q[0] : 0.5f) * dLx;
                    // y: DPP row shifts, the block faces from the Hm / Ha rows
                    const float sB = dpp_shr1(uc), sT = dpp_shl1(uc);
                    const float vB = y0 ? hY[k] : sB, vT = y7 ? hY[k] : sT;
                    const float dTy = vT - uc, dBy = uc - vB;
                    const float aBy = (FINEBLK && y0) ? gY[k] : fabsf(dBy), aTy = (FINEBLK && y7) ? gY[k] : fabsf(dTy);
                    SY[x] = qyT * dTy + qyB * dBy;
                    // z: ds_bpermute
                    const float sD = bperm(dn, uc), sU = bperm(up, uc);
                    const float vD = z0 ? hZ[k] : sD, vU = z7 ? hZ[k] : sU;
                    const float dUz = vU - uc, dDz = uc - vD;
                    const float aDz = (FINEBLK && z0) ? gZ[k] : fabsf(dDz), aUz = (FINEBLK && z7) ? gZ[k] : fabsf(dUz);
                    SZ[x] = qzT * dUz + qzB * dDz;
                    D[x] = jst_max3(dRx - dLx, aRx + aLx, rhx, dTy - dBy, aTy + aBy, rhy, dUz - dDz, aUz + aDz, rhz);
                }
            }
        };
        if (bb.fine >= 0) slopes(std::true_type{});
        else slopes(std::false_type{});
    }
    // boundary cells' slopes, sensor and normal velocity to the slot lanes
    put_faces<1>(lds, lane, &SX[0], &SX[7], SY, SZ);
    put_faces<2>(lds, lane, &D[0], &D[7], D, D);
    __builtin_amdgcn_sched_barrier(0);
    ld8(C + ldc, CY);
    __builtin_amdgcn_sched_barrier(0);
    put_faces_x<3>(lds, lane, CX[0], CX[7]);

    // ---- interior faces, four at a time: +x (inside the strip), +y (lane + 1), +z (lane + 8); their part of the
    // Green-Gauss sum right away (faces on the block's sides count zero here, the edge fluxes are added at the end)
    float res[8];
    {
        auto V = [](const float* a, int o) { return v4f{a[o], a[o + 1], a[o + 2], a[o + 3]}; };
        const v4f x0 = flux_half4(V(U, 0), V(U, 1), V(SX, 0), V(SX, 1), V(D, 0), V(D, 1), V(CX, 0), V(CX, 1));
        const v4f x1 = flux_half4(V(U, 4), v4f{U[5], U[6], U[7], U[7]}, V(SX, 4), v4f{SX[5], SX[6], SX[7], SX[7]}, V(D, 4),
                                  v4f{D[5], D[6], D[7], D[7]}, V(CX, 4), v4f{CX[5], CX[6], CX[7], CX[7]});
        const float fx[9] = {0.0f, x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, 0.0f};  // face x - 1/2, x = 0..8
#pragma unroll
        for (int x = 0; x < 8; ++x) res[x] = -((fx[x + 1] - fx[x]) * rhx);
        const int up = (lane + 8) << 2, dn = ((lane - 8) & 63) << 2;
        // one direction after the other (scheduling barriers): the slopes and velocities of a finished direction die
        __builtin_amdgcn_sched_barrier(0);
        ld8(C + 2 * (size_t)ldc, CZ);
        __builtin_amdgcn_sched_barrier(0);
        put_faces_y<3>(lds, lane, CY);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            asm volatile("" : "+v"(D[4 * h]), "+v"(D[4 * h + 1]), "+v"(D[4 * h + 2]), "+v"(D[4 * h + 3])::"memory");
            __builtin_amdgcn_sched_barrier(0);
            v4f Ub, Sb, Db, Cb;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int x = 4 * h + k;
                Ub[k] = dpp_shl1(U[x]);
                Sb[k] = dpp_shl1(SY[x]);
                Db[k] = dpp_shl1(D[x]);
                Cb[k] = dpp_shl1(CY[x]);
            }
            const v4f fy = flux_half4(V(U, 4 * h), Ub, V(SY, 4 * h), Sb, V(D, 4 * h), Db, V(CY, 4 * h), Cb);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // the face below is the top face of lane - 1; for y = 0 that is the (zeroed) y = 7 face of the
                // previous row or the zero fill of the shift: no select
                const float ft = y7 ? 0.0f : fy[k];
                res[4 * h + k] -= (ft - dpp_shr1(ft)) * rhy;
            }
        }
        put_faces_z<3>(lds, lane, CZ);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            asm volatile("" : "+v"(D[4 * h]), "+v"(D[4 * h + 1]), "+v"(D[4 * h + 2]), "+v"(D[4 * h + 3])::"memory");
            __builtin_amdgcn_sched_barrier(0);
            v4f Uc, Sc, Dc, Cc;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int x = 4 * h + k;
                Uc[k] = bperm(up, U[x]);
                Sc[k] = bperm(up, SZ[x]);
                Dc[k] = bperm(up, D[x]);
                Cc[k] = bperm(up, CZ[x]);
            }
            const v4f fz = flux_half4(V(U, 4 * h), Uc, V(SZ, 4 * h), Sc, V(D, 4 * h), Dc, V(CZ, 4 * h), Cc);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float fu = z7 ? 0.0f : fz[k];
                res[4 * h + k] -= (fu - bperm(dn, fu)) * rhz;  // z = 0: lane - 8 wraps to the zeroed z = 7 face
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- (b) block faces, side after side (the Hm rows become the edge fluxes)
    {
        float hu[6], hde[6], hc[6], rv[6];
        // (the asm keeps the compiler from forming the 64-bit addresses of these loads before the slope phase and
        // carrying them -- twice the registers of the ids -- through it)
        asm volatile("" : "+v"(hid[0]), "+v"(hid[1]), "+v"(hid[2]), "+v"(hid[3]), "+v"(hid[4]), "+v"(hid[5]));
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            const int d = s >> 1, sd = d == 0 ? 1 : d == 1 ? 8 : 64;
            const int dd = bb.type[s] == SIDE_MIRROR ? 0 : ((s & 1) ? sd : -sd);
            hu[s] = ldg(u, hid[s]);
            hde[s] = ldg(u, (uint32_t)((int)hid[s] + dd));
            hc[s] = ldg(C + (size_t)d * ldc, hid[s]);
            rv[s] = ldg(u, (uint32_t)(rid[s] >= 0 ? rid[s] : bb.base));
        }
        const LaneGeo LG = lane_geo(lane);
        wave_lds_sync();
        side_flux<0>(bb, LG, ftab, r4tab, u, C, lds, lane, hu[0], hde[0], hc[0], rid[0], rv[0]);
        side_flux<1>(bb, LG, ftab, r4tab, u, C, lds, lane, hu[1], hde[1], hc[1], rid[1], rv[1]);
        side_flux<2>(bb, LG, ftab, r4tab, u, C + ldc, lds, lane, hu[2], hde[2], hc[2], rid[2], rv[2]);
        side_flux<3>(bb, LG, ftab, r4tab, u, C + ldc, lds, lane, hu[3], hde[3], hc[3], rid[3], rv[3]);
        side_flux<4>(bb, LG, ftab, r4tab, u, C + 2 * (size_t)ldc, lds, lane, hu[4], hde[4], hc[4], rid[4], rv[4]);
        side_flux<5>(bb, LG, ftab, r4tab, u, C + 2 * (size_t)ldc, lds, lane, hu[5], hde[5], hc[5], rid[5], rv[5]);
        wave_lds_sync();
    }

    // ---- the block faces' part of the Green-Gauss sum
    {
        const float* EX = lds + S3_HM;
        const float ex0 = EX[lane], ex1 = EX[64 + lane];
        const int ry = (y7 ? 192 : 128) + 8 * z, rz = (z7 ? 320 : 256) + 8 * y;
        const v4f eY0 = *(const v4f*)(EX + ry), eY1 = *(const v4f*)(EX + ry + 4);
        const v4f eZ0 = *(const v4f*)(EX + rz), eZ1 = *(const v4f*)(EX + rz + 4);
        // a low face adds +flux / h, a high face -flux / h
        const float wy = y0 ? rhy : y7 ? -rhy : 0.0f, wz = z0 ? rhz : z7 ? -rhz : 0.0f;
        res[0] += ex0 * rhx;
        res[7] -= ex1 * rhx;
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            const float eY = x < 4 ? eY0[x & 3] : eY1[x & 3], eZ = x < 4 ? eZ0[x & 3] : eZ1[x & 3];
            res[x] += eY * wy + eZ * wz;
        }
        *(v4f_g*)(ud + a0) = v4f{res[0], res[1], res[2], res[3]};
        *(v4f_g*)(ud + a0 + 4) = v4f{res[4], res[5], res[6], res[7]};
    }
}

#pragma clang fp contract(off)

}  // namespace strip3
