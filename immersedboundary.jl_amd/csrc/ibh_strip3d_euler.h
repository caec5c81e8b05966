// Column sweep of the 3-D Euler residual (P = [p T u v w]: MUSCL(high_order) with the pressure JST sensor,
// CFD.inviscid_fluxes HLL, Green-Gauss; cfd.jl:459-508 over ImmersedBoundary.jl:1077-1157) on an octree partition:
// ONE wavefront sweeps an 8x8x8 block, and in every one of the three directions a lane owns a whole COLUMN of 8 cells
// along that direction -- so every face of the column, its two block faces included, has both its cells, their slopes
// and sensors in the lane's own registers, and the flux loop has no cross-lane traffic at all.  Between directions the
// block (5 primitives, sensor, residual) is transposed through a 2.3 KB wave-private LDS buffer.
//
//   layout d (columns along d): lane = ta + 8 tb, (ta, tb) = the two other coordinates in increasing dim order -- which is
//   exactly the slot numbering of the block sides normal to d (htab3 / rtab3, ibh_analyze3.cpp): the lane of a column IS
//   the lane of the halo slots at its two ends.  Halo cell(s), the cell one step deeper, slope along the normal and the
//   pressure sensor of the halo cell (lateral neighbours: the side's plane + rim table, as strip3::side_flux) are
//   computed where they are used.
//
//   column registers: e[j] = (c[j - 1], c[j + 4]), j = 0..4, c[0..7] the cells, c[-1] / c[8] the halo ends.  Faces
//   (j, j + 5), j = 0..3, are then one packed evaluation on register-aligned pairs (e[j], e[j + 1]); face 4 (between
//   e[4].x and e[0].y) is a scalar evaluation.  Slopes and residuals of cells (j - 1, j + 4) are packed the same way.
//
//   order: load z-columns (coalesced dwords) -> sensor: z, y, x partial ratios carried as (numerator, denominator) so that
//   one reciprocal serves the three directions -> D back to z -> fluxes: z, y, x -> store x-columns (two float4 per lane).
//
// Arithmetic: blk3::euler_flux_w3 / blk3::sweep_euler (ibh_block3d.h), Float32 HLL combine like every tuned path.
#pragma once
#include "ibh_strip3d.h"
#include "ibh_quad2d_euler.h"

namespace strip3e {

#pragma clang fp contract(fast)

using blk2::ldg;
using blk2::wave_lds_sync;
using blk3::Gas3;
using blk3::halo_cell3s;
using quad2::lds_read;
using quad2::v2f;
using quad2::v4f;
using quad2::v4f_g;
using strip3::jst_max3;
using strip3::lane_geo;
using strip3::LaneGeo;

// LDS per wave (floats).  The plane of side_eval lives in the transposition buffer: the two are used in different phases
// and a wavefront's LDS operations execute in order.
#define S3E_BUF 0                       // transposition buffer (8 x 72)
#define S3E_PLANE 0                     // [18 x 18] pressure of the halo cells of one side + rim
#define S3E_PLANEA (S3E_PLANE + 324)    // [64] rim: mean |difference| to the halo cell next to it
#define S3E_R 576                       // [5][576] residual
#define S3E_LDS (S3E_R + 5 * 576)       // 3 456 floats = 13.5 KB per wave (+ 5 KB for the first loads of the next block)


struct Col {
    v2f e[5];
};
// cell i of a column (compile-time i)
template <int I>
__device__ __forceinline__ void set_cell(Col& c, float v) {
    if constexpr (I < 4) c.e[I + 1].x = v;
    else c.e[I - 4].y = v;
}
template <int I>
__device__ __forceinline__ float cellv(const Col& c) {
    if constexpr (I < 4) return c.e[I + 1].x;
    else return c.e[I - 4].y;
}

// ---- transposition between the column layouts.  Two address maps, each free of bank conflicts for both of its layouts
// (32 lanes of a half-wave on 32 different banks at every column position) and affine in the column position, so every
// access is base + immediate:
//   y <-> z : A = x + 8 y + 72 z        y-columns: (ta, tb) = (x, z)   z-columns: (ta, tb) = (x, y)
//   x <-> y : A = x + 65 y + 8 z        x-columns: (ta, tb) = (y, z)   y-columns: (ta, tb) = (x, z)
template <int FROM, int TO>
struct TrMap {
    static constexpr bool yz = (FROM + TO) == 3;
    static __device__ __forceinline__ int base(int layout, int ta, int tb) {
        if (yz) return layout == 1 ? ta + 72 * tb : ta + 8 * tb;
        return layout == 0 ? 65 * ta + 8 * tb : ta + 8 * tb;
    }
    static constexpr int stride(int layout) {
        if (yz) return layout == 1 ? 8 : 72;
        return layout == 0 ? 1 : 65;
    }
};

template <int FROM, int TO, int I = 0>
__device__ __forceinline__ void tr_write(float* w, const Col& c) {
    if constexpr (I < 8) {
        w[TrMap<FROM, TO>::stride(FROM) * I] = cellv<I>(c);
        tr_write<FROM, TO, I + 1>(w, c);
    }
}
template <int FROM, int TO, int I = 0>
__device__ __forceinline__ void tr_read(const float* r, Col& c) {
    if constexpr (I < 8) {
        set_cell<I>(c, r[TrMap<FROM, TO>::stride(TO) * I]);
        tr_read<FROM, TO, I + 1>(r, c);
    }
}
// the cells of one quantity from layout FROM to layout TO (a wavefront's LDS operations execute in order: the buffer is
// reused quantity after quantity without waiting)
template <int FROM, int TO>
__device__ __forceinline__ void transpose(float* buf, int ta, int tb, Col& c) {
#ifdef S3E_ABLATE_TRANSPOSE  // (timing diagnostic, wrong results)
    return;
#endif
    float* w = buf + TrMap<FROM, TO>::base(FROM, ta, tb);
    const float* r = buf + TrMap<FROM, TO>::base(TO, ta, tb);
    tr_write<FROM, TO>(w, c);
    wave_lds_sync();
    tr_read<FROM, TO>(r, c);
    wave_lds_sync();
}

// ---- arithmetic on one face (float) or two faces (v2f) at a time
__device__ __forceinline__ float rcpT(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ v2f rcpT(v2f x) { return quad2::rcp2(x); }
__device__ __forceinline__ float sqrtT(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ v2f sqrtT(v2f x) { return quad2::sqrt2(x); }
__device__ __forceinline__ float maxT(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ v2f maxT(v2f a, v2f b) { return quad2::max2(a, b); }
__device__ __forceinline__ float minT(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ v2f minT(v2f a, v2f b) { return quad2::min2(a, b); }
__device__ __forceinline__ float med0T(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, 0.0f); }
__device__ __forceinline__ v2f med0T(v2f a, v2f b) {
    return v2f{__builtin_amdgcn_fmed3f(a.x, b.x, 0.0f), __builtin_amdgcn_fmed3f(a.y, b.y, 0.0f)};
}
template <class T>
__device__ __forceinline__ T bc(float x);
template <>
__device__ __forceinline__ float bc<float>(float x) { return x; }
template <>
__device__ __forceinline__ v2f bc<v2f>(float x) { return v2f{x, x}; }

// state, physical flux, normal velocity and speed of sound of one side of a face (blk3::euler_side3)
template <class T, int DN>
__device__ __forceinline__ void euler_side(const T* P, const Gas3& gas, T* Q, T* F, T& un, T& a) {
    const T p = P[0];
    const T Tt = maxT(P[1], bc<T>(10.0f));
    const T k = 0.5f * (P[2] * P[2] + P[3] * P[3] + P[4] * P[4]);
    const T rho = p * rcpT(gas.R * Tt);
    const T E = rho * (gas.R / (gas.gamma - 1.0f) * Tt + k);
    Q[0] = rho;
    Q[1] = E;
    Q[2] = rho * P[2];
    Q[3] = rho * P[3];
    Q[4] = rho * P[4];
    un = P[2 + DN];
    a = sqrtT((gas.gamma * gas.R) * Tt);
    F[0] = Q[0] * un;
    F[1] = (Q[1] + p) * un;
    F[2] = DN == 0 ? Q[2] * un + p : Q[2] * un;
    F[3] = DN == 1 ? Q[3] * un + p : Q[3] * un;
    F[4] = DN == 2 ? Q[4] * un + p : Q[4] * un;
}

// conserved state, pressure, normal velocity and speed of sound of one side of a face
template <class T, int DN>
__device__ __forceinline__ void euler_state(const T* P, const Gas3& gas, T* Q, T& p, T& un, T& a) {
    p = P[0];
    const T Tt = maxT(P[1], bc<T>(10.0f));
    const T k = 0.5f * (P[2] * P[2] + P[3] * P[3] + P[4] * P[4]);
    const T rho = p * rcpT(gas.R * Tt);
    Q[0] = rho;
    Q[1] = rho * (gas.R / (gas.gamma - 1.0f) * Tt + k);
    Q[2] = rho * P[2];
    Q[3] = rho * P[3];
    Q[4] = rho * P[4];
    un = P[2 + DN];
    a = sqrtT((gas.gamma * gas.R) * Tt);
}

// density, energy per mass, pressure, normal velocity and speed of sound of one side of a face
template <class T, int DN>
__device__ __forceinline__ void euler_side_pm(const T* P, const Gas3& gas, T& rho, T& e, T& p, T& un, T& a) {
    p = P[0];
    const T Tt = maxT(P[1], bc<T>(10.0f));
    const T q = P[2] * P[2] + P[3] * P[3] + P[4] * P[4];
    rho = p * rcpT(gas.R * Tt);
    e = (gas.R / (gas.gamma - 1.0f)) * Tt + 0.5f * q;
    un = P[2 + DN];
    a = sqrtT((gas.gamma * gas.R) * Tt);
}

// MUSCL states from undivided slopes, then HLL (blk3::euler_flux_w3); a = owner (towards -), b = neighbour,
// wa = h_a / (h_a + h_b)
template <class T, int DN>
__device__ __forceinline__ void euler_flux(const T* Pa, const T* Pb, const T* Sa, const T* Sb, T Da, T Db, T wa,
                                           const Gas3& gas, T* F) {
#ifdef S3E_ABLATE_FLUX  // (timing diagnostic, wrong results: the flux arithmetic removed, everything else in place)
#pragma unroll
    for (int v = 0; v < 5; ++v) F[v] = (Pa[v] + Pb[v]) * wa + (Sa[v] - Sb[v]) * (Da + Db);
    return;
#endif
    T PL[5], PR[5];
    const T Df = maxT(maxT(Da, Db), bc<T>(1e-7f));
    const T wb = 1.0f - wa;
#pragma unroll
    for (int v = 0; v < 5; ++v) {
        const T d = Pb[v] - Pa[v];
        const T gu = Sa[v] - d * wa;
        const T Du = Sb[v] - d * wb;
        const T s = med0T(Du, gu);
        const T t16 = (Sa[v] - Sb[v]) * 0.0625f;
        const T uf = (Pa[v] + wa * d) + t16;
        PL[v] = uf + Df * ((s - wa * d) - t16);
        // PR = uf + Df ((wb d - s) - t16) = PL + Df (d - 2 s)   (wa + wb = 1)
        PR[v] = PL[v] + Df * (d - 2.0f * s);
    }
    // HLL: F = (SL FL - SR FR + SL SR (QR - QL)) / (SL - SR) with FL = QL unL + pressure terms regrouped by state:
    // F = QL (wL unL - c) + QR (c - wR unR) + pressure terms, wL = SL / (SL - SR), wR = SR / (SL - SR), c = SL wR; with
    // Q = rho (1, e, u, v, w) the conserved states are never formed: F = AL (1, eL, uL..) + AR (1, eR, uR..), A = rho c
    T rL, eL, pL, uL, aL, rR, eR, pR, uR, aR;
    euler_side_pm<T, DN>(PL, gas, rL, eL, pL, uL, aL);
    euler_side_pm<T, DN>(PR, gas, rR, eR, pR, uR, aR);
    const T z = bc<T>(0.0f);
    const T SR = minT(uR - aR, z);
    const T SL = maxT(uL + aL, z);
    const T rs = rcpT(SL - SR);
    const T wL = SL * rs, wR = SR * rs;
    const T c = SL * wR;
    const T AL = rL * (wL * uL - c), AR = rR * (c - wR * uR);
    F[0] = AL + AR;
    F[1] = AL * eL + AR * eR;
#pragma unroll
    for (int v = 2; v < 5; ++v) F[v] = AL * PL[v] + AR * PR[v];
    // (the form (QL - QR) cL + QR (cL + cR), exact in the difference of the states, was measured: the same error against
    // the Float64 combine of the reference -- 9.6e-6 of max |R| on tests/test_config5.py either way -- at five more
    // instructions per face: the error is the Float32 rounding of the fluxes themselves)
    const T mL = wL * pL, mR = wR * pR;
    F[2 + DN] += mL - mR;
    F[1] += mL * uL - mR * uR;
}

__device__ __forceinline__ float dpp_xor1(float v) {  // lane i <- lane i ^ 1 (quad_perm [1,0,3,2])
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_xor8(float v) {  // lane i <- lane i ^ 8 (row_ror:8 in rows of 16)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, true));
}

// pressure sensor of ONE halo cell (strip3::slot_eval without the flux).  GROUP: the halo cell is coarser, four of the
// block's cells (m0..m3) face it; otherwise one (m0)
template <bool GROUP>
__device__ __forceinline__ float slot_sensor(const float* pl, const float* pA, float h, float hde, float m0, float m1,
                                             float m2, float m3, int la, int ha, int lb, int hb, bool ra0, bool ra1,
                                             bool rb0, bool rb1, int ia, int ib, int n, float rn, float ra, float rb) {
    float din, ain;
    if constexpr (GROUP) {
        din = 0.25f * ((m0 + m1) + (m2 + m3)) - h;
        ain = 0.25f * ((fabsf(m0 - h) + fabsf(m1 - h)) + (fabsf(m2 - h) + fabsf(m3 - h)));
    } else {
        din = m0 - h;
        ain = fabsf(din);
    }
    const float dde = hde - h;
    // eight LDS reads every lane performs, in flight together (the asm keeps the compiler from sinking one of them into a
    // divergent branch of the selects below -- and, being ONE statement, from waiting for them one by one)
    float pa0 = pl[la], pa1 = pl[ha], pb0 = pl[lb], pb1 = pl[hb];
    float A0 = pA[ib], A1 = pA[n + ib], B0 = pA[2 * n + ia], B1 = pA[3 * n + ia];
    asm volatile("" : "+v"(pa0), "+v"(pa1), "+v"(pb0), "+v"(pb1), "+v"(A0), "+v"(A1), "+v"(B0), "+v"(B1));
    const float ea0 = pa0 - h, ea1 = pa1 - h, eb0 = pb0 - h, eb1 = pb1 - h;
    const float aa0 = ra0 ? A0 : fabsf(ea0), aa1 = ra1 ? A1 : fabsf(ea1);
    const float ab0 = rb0 ? B0 : fabsf(eb0), ab1 = rb1 ? B1 : fabsf(eb1);
    return jst_max3(din + dde, ain + fabsf(dde), rn, ea0 + ea1, aa0 + aa1, ra, eb0 + eb1, ab0 + ab1, rb);
}

// what a lane holds of slot `lane` of one side
struct Slot {
    uint32_t hid;         // halo cell (the first of four on a FINE side)
    int dd;               // offset to the cell one step deeper ...
    const int32_t* drow;  // ... or (image-only sweeps of a partition with skirt fragments) the deeper-cell table row of the
                          // block, [6][64][4]; null: arithmetic everywhere
};
template <int S>
__device__ __forceinline__ Slot slot_of(const BlockDesc3& bb, const int32_t* __restrict__ htab, int32_t blk, int lane,
                                        const int32_t* __restrict__ dtab = nullptr) {
    constexpr int d = S >> 1;
    constexpr int sd = d == 0 ? 1 : d == 1 ? 8 : 64;
    Slot s;
    s.hid = halo_cell3s<S>(bb, htab, blk, lane);
#ifdef S3E_ABLATE_HALO  // (timing diagnostic, wrong results: every halo gather becomes a coalesced read of the block's own cells)
    s.hid = (uint32_t)bb.base + 64u * S + (uint32_t)lane;
#endif
    s.dd = bb.type[S] == SIDE_MIRROR ? 0 : ((S & 1) ? sd : -sd);
    s.drow = dtab ? dtab + (size_t)blk * 1536 : nullptr;
    return s;
}
// the cell one step deeper behind halo cell c = k-th cell of slot `lane` of side S
template <int S>
__device__ __forceinline__ uint32_t deeper_of(const BlockDesc3& bb, const Slot& sl, int lane, int k, uint32_t c) {
    if (sl.drow && bb.nb[S] < 0 && bb.type[S] != SIDE_MIRROR)  // wave-uniform
        return (uint32_t)sl.drow[(S * 64 + lane) * 4 + k];
#ifdef S3E_ABLATE_HALO
    return c;
#endif
    return (uint32_t)((int)c + sl.dd);
}

// mean halo pressure behind boundary cell `lane` of side S and mean |halo - boundary cell| (own-cell sensor)
template <int S>
__device__ __forceinline__ void side_mean_p(const BlockDesc3& bb, const int32_t* __restrict__ ftab,
                                            const float* __restrict__ P, int lane, float h0, float pb, float& hm,
                                            float& ha) {
    hm = h0;
    ha = fabsf(h0 - pb);
    if (bb.type[S] == SIDE_FINE) {  // wave-uniform
        const int32_t* ft = ftab + (((size_t)bb.fine * 6 + S) * 64 + lane) * 3;
        const float v1 = ldg(P, (uint32_t)ft[0]), v2 = ldg(P, (uint32_t)ft[1]), v3 = ldg(P, (uint32_t)ft[2]);
        hm = 0.25f * ((h0 + v1) + (v2 + v3));
        ha = 0.25f * ((fabsf(h0 - pb) + fabsf(v1 - pb)) + (fabsf(v2 - pb) + fabsf(v3 - pb)));
    }
}

// ---- sensor pass along D: numerator / denominator of the JST ratio of the pressure for the 8 cells of the column
// (n = |second difference| / h + 1e-7, d = sum of |first differences| / h + 1e-7), folded into the running pair (N, Dn)
// of the directions done before: max over directions of n_i / d_i = N / Dn with N = max(N d, n Dn), Dn = Dn d.
template <int D, bool FIRST>
__device__ __forceinline__ void sensor_pass(const BlockDesc3& bb, const int32_t* __restrict__ ftab,
                                            const float* __restrict__ P, int lane, float hp0, float hp1, Col& p, Col& N,
                                            Col& Dn) {
    constexpr int S0 = 2 * D, S1 = 2 * D + 1;
    const float rh = bb.rh[D];
    float hm0, ha0, hm1, ha1;
    side_mean_p<S0>(bb, ftab, P, lane, hp0, cellv<0>(p), hm0, ha0);
    side_mean_p<S1>(bb, ftab, P, lane, hp1, cellv<7>(p), hm1, ha1);
    p.e[0].x = hm0;
    p.e[4].y = hm1;
    v2f dE[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) dE[j] = p.e[j + 1] - p.e[j];
    const float d4 = p.e[0].y - p.e[4].x;
    // returns (N, Dn) of one cell
    auto fold = [&](float g, float a, float Nq, float Dq) {
        const float n = fmaf(fabsf(g), rh, 1e-7f), d = fmaf(a, rh, 1e-7f);
        if constexpr (FIRST) return v2f{n, d};
        else return v2f{fmaxf(Nq * d, n * Dq), Dq * d};
    };
#pragma unroll
    for (int j = 1; j < 4; ++j) {  // cells (j - 1, j + 4)
        const v2f g = dE[j] - dE[j - 1];
        const float aLx = j == 1 ? ha0 : fabsf(dE[j - 1].x), aRy = j == 3 ? ha1 : fabsf(dE[j].y);
        const v2f lo = fold(g.x, fabsf(dE[j].x) + aLx, N.e[j].x, Dn.e[j].x);
        const v2f hi = fold(g.y, aRy + fabsf(dE[j - 1].y), N.e[j].y, Dn.e[j].y);
        N.e[j] = v2f{lo.x, hi.x};
        Dn.e[j] = v2f{lo.y, hi.y};
    }
    const v2f c3 = fold(d4 - dE[3].x, fabsf(d4) + fabsf(dE[3].x), N.e[4].x, Dn.e[4].x);  // cell 3
    const v2f c4 = fold(dE[0].y - d4, fabsf(dE[0].y) + fabsf(d4), N.e[0].y, Dn.e[0].y);  // cell 4
    N.e[4].x = c3.x;
    Dn.e[4].x = c3.y;
    N.e[0].y = c4.x;
    Dn.e[0].y = c4.y;
}

// ---- halo cells of slot `lane` of side S: slopes of the five primitives along the side normal (towards +) and
// pressure sensor; on a FINE side instead the mean flux through the four sub-faces (Ff)
template <int S>
__device__ __forceinline__ void side_eval(const BlockDesc3& bb, const LaneGeo& LG, const int32_t* __restrict__ ftab,
                                          const int32_t* __restrict__ r4tab, const float* __restrict__ P, uint32_t ldp,
                                          float* lds, int lane, const Slot& sl, const float* hu, const float* hde,
                                          int32_t rid, float rv0, const float* Pb, const float* Sb, float Db,
                                          const Gas3& gas, float* Sh, float& Dh, float* Ff) {
    constexpr int d = S >> 1;
    constexpr bool low = (S & 1) == 0;
    constexpr int da = d == 0 ? 1 : 0, db = d == 2 ? 1 : 2;
#ifdef S3E_ABLATE_SIDE  // (timing diagnostic, wrong results: no slope / sensor work for the halo cells)
#pragma unroll
    for (int v = 0; v < 5; ++v) Sh[v] = Sb[v] + hu[v] - hde[v];
    Dh = Db + rv0;
    return;
#endif
    const int ty = bb.type[S];
    const float qs = bb.q[S];
#ifdef S3E_COUNT_SAME_ONLY  // (instruction counts of the path without FINE / COARSE sides: scripts/isa_count.py)
    const bool isC = false, isF = false, mirror = ty == SIDE_MIRROR;
#else
    const bool isC = ty == SIDE_COARSE, isF = ty == SIDE_FINE, mirror = ty == SIDE_MIRROR;
#endif
    const int t1 = lane & 7, t2 = lane >> 3;
    float* pl = lds + S3E_PLANE;
    float* pA = lds + S3E_PLANEA;
    const float irt = isC ? 0.5f : isF ? 2.0f : 1.0f;  // h / h_halo
    const float rn = bb.rh[d] * irt, ra = bb.rh[da] * irt, rb = bb.rh[db] * irt;
    float rv[4] = {rv0, rv0, rv0, rv0};
    if (rid < 0) {  // rim neighbour = four finer cells (few lanes)
        const int32_t* r4 = r4tab + (size_t)(-rid - 1) * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) rv[k] = ldg(P, (uint32_t)r4[k]);
    }
    const float rmean = 0.25f * ((rv[0] + rv[1]) + (rv[2] + rv[3]));
    if (!isF) {  // wave-uniform
        pl[LG.pc] = hu[0];
        if (lane < 32) pl[LG.rpos8] = rmean;
        wave_lds_sync();
        if (lane < 32) {
            const float ha = pl[LG.radj8];
            pA[lane] = 0.25f * ((fabsf(rv[0] - ha) + fabsf(rv[1] - ha)) + (fabsf(rv[2] - ha) + fabsf(rv[3] - ha)));
        }
        wave_lds_sync();
        float mean[5];
        if (!isC) {  // SAME / MIRROR: the lateral neighbours are the next slots, one cell of the block in front
#pragma unroll
            for (int v = 0; v < 5; ++v) mean[v] = Pb[v];
            Dh = slot_sensor<false>(pl, pA, hu[0], hde[0], Pb[0], Pb[0], Pb[0], Pb[0], LG.pc - 1, LG.pc + 1, LG.pc - 18,
                                    LG.pc + 18, t1 == 0, t1 == 7, t2 == 0, t2 == 7, t1, t2, 8, rn, ra, rb);
        } else {     // COARSE: the halo cell spans the 2 x 2 group of slots t, t ^ 1, t ^ 8, t ^ 9
            const int c1 = LG.b1, c2 = LG.b2;
            const int la = LG.pc - 1 - c1, ha = LG.pc + 2 - c1, lb = LG.pc - 18 - 18 * c2, hb = LG.pc + 36 - 18 * c2;
#pragma unroll
            for (int v = 0; v < 5; ++v) {
                const float s1 = Pb[v] + dpp_xor1(Pb[v]);
                mean[v] = 0.25f * (s1 + dpp_xor8(s1));
            }
            const float m1 = dpp_xor1(Pb[0]), m2 = dpp_xor8(Pb[0]), m3 = dpp_xor8(m1);
            Dh = slot_sensor<true>(pl, pA, hu[0], hde[0], Pb[0], m1, m2, m3, la, ha, lb, hb, t1 <= 1, t1 >= 6, t2 <= 1,
                                   t2 >= 6, t1, t2, 8, rn, ra, rb);
        }
        if (mirror) {  // wave-uniform (domain boundary): the halo cell is the boundary cell itself
#pragma unroll
            for (int v = 0; v < 5; ++v) Sh[v] = Sb[v];
            Dh = Db;
        } else {
#pragma unroll
            for (int v = 0; v < 5; ++v) {
                const float x = (1.0f - qs) * (mean[v] - hu[v]) - 0.5f * (hde[v] - hu[v]);
                Sh[v] = low ? x : -x;
            }
        }
    } else {  // the 2 x 2 finer cells behind this boundary cell, one after the other (rolled loop)
        const int32_t* ft = ftab + (((size_t)bb.fine * 6 + S) * 64 + lane) * 3;
        pl[(2 * t1 + 1) + 18 * (2 * t2 + 1)] = hu[0];
#pragma unroll 1
        for (int k = 1; k < 4; ++k)
            pl[(2 * t1 + (k & 1) + 1) + 18 * (2 * t2 + (k >> 1) + 1)] = ldg(P, (uint32_t)ft[k - 1]);
        pl[LG.rpos16] = rmean;
        wave_lds_sync();
        {
            const float ha = pl[LG.radj16];
            pA[lane] = 0.25f * ((fabsf(rv[0] - ha) + fabsf(rv[1] - ha)) + (fabsf(rv[2] - ha) + fabsf(rv[3] - ha)));
        }
        wave_lds_sync();
        float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int k = 0; k < 4; ++k) {
            float hk[5], hdk[5], shk[5], X[5];
            const uint32_t c = k == 0 ? sl.hid : (uint32_t)ft[k > 0 ? k - 1 : 0];
#pragma unroll
            for (int v = 0; v < 5; ++v) {
                hk[v] = ldg(P + (size_t)v * ldp, c);
                hdk[v] = ldg(P + (size_t)v * ldp, deeper_of<S>(bb, sl, lane, k, c));
            }
            const int f1 = 2 * t1 + (k & 1), f2 = 2 * t2 + (k >> 1);
            const float dhk = slot_sensor<false>(pl, pA, hk[0], hdk[0], Pb[0], Pb[0], Pb[0], Pb[0], f1 + 18 * (f2 + 1),
                                                 f1 + 2 + 18 * (f2 + 1), (f1 + 1) + 18 * f2, (f1 + 1) + 18 * (f2 + 2),
                                                 f1 == 0, f1 == 15, f2 == 0, f2 == 15, f1, f2, 16, rn, ra, rb);
#pragma unroll
            for (int v = 0; v < 5; ++v) {
                const float x = (1.0f - qs) * (Pb[v] - hk[v]) - 0.5f * (hdk[v] - hk[v]);
                shk[v] = low ? x : -x;
            }
            if (low) euler_flux<float, d>(hk, Pb, shk, Sb, dhk, Db, 1.0f - qs, gas, X);
            else euler_flux<float, d>(Pb, hk, Sb, shk, Db, dhk, qs, gas, X);
#pragma unroll
            for (int v = 0; v < 5; ++v) acc[v] += X[v];
        }
#pragma unroll
        for (int v = 0; v < 5; ++v) {
            Ff[v] = 0.25f * acc[v];
            Sh[v] = Sb[v];  // finite stand-ins: the packed evaluation of this face is replaced by Ff
        }
        Dh = Db;
    }
}

// what a lane holds of slot `lane` of one side for a flux pass: the five primitives of the halo cell (the first of four
// on a FINE side) and of the cell one step deeper, the pressure of the rim cell
struct HaloRegs {
    float hu[5], hd[5], rv;
};
// (Round 4 also evaluated the two sides of a direction at once where both are SAME / MIRROR sides -- the arithmetic of
// side_eval on (low, high) pairs, one plane each in LDS, one pair of LDS round trips for both: 70 fewer vector instructions
// per direction, 23 spilled registers, and slower: 128 against 107 us at 4.56 M cells on one box.  Not kept.)
template <int S>
__device__ __forceinline__ void halo_load_values(const BlockDesc3& bb, int lane, const float* __restrict__ P, uint32_t ldp,
                                                 const Slot& sl, HaloRegs& h) {
    constexpr int d = S >> 1;
    constexpr bool low = (S & 1) == 0;
    if constexpr (d == 0) {
        // x sides: the halo cell and the cell one step deeper are neighbours in memory (x = 7, 6 of the block on the left,
        // x = 0, 1 of the one on the right: an even / odd pair, blocks start at multiples of 512) -- ONE 8-byte gather per
        // field instead of two 4-byte ones over the same 16 cache lines.  Not across a mirror side (no deeper cell) and
        // not towards a skirt fragment (deeper cells from the table).
        const bool table = sl.drow && bb.nb[S] < 0 && bb.type[S] != SIDE_MIRROR;
        if (sl.dd == (low ? -1 : 1) && !table) {  // wave-uniform
            typedef float v2f_a __attribute__((ext_vector_type(2), aligned(8)));
            const uint32_t c2 = sl.hid & ~1u;
#pragma unroll
            for (int v = 0; v < 5; ++v) {
                const v2f_a t = *(const v2f_a*)((const char*)(P + (size_t)v * ldp) + (size_t)(c2 << 2));
                h.hu[v] = low ? t.y : t.x;
                h.hd[v] = low ? t.x : t.y;
            }
            return;
        }
    }
    const uint32_t hd = deeper_of<S>(bb, sl, lane, 0, sl.hid);
#pragma unroll
    for (int v = 0; v < 5; ++v) {
        const float* Pv = P + (size_t)v * ldp;
        h.hu[v] = ldg(Pv, sl.hid);
        h.hd[v] = ldg(Pv, hd);
    }
}
template <int S>
__device__ __forceinline__ void halo_load(const BlockDesc3& bb, int lane, const float* __restrict__ P, uint32_t ldp,
                                          const Slot& sl, int32_t rid, HaloRegs& h) {
    halo_load_values<S>(bb, lane, P, ldp, sl, h);
    h.rv = ldg(P, (uint32_t)(rid >= 0 ? rid : bb.base));
}

// r += v in LDS: read, add, write.  (Round 4 tried ds_add_f32 -- one LDS instruction, no wait for the read -- for the ~50
// accumulations of a pass: the sweep took 264 us instead of 107 at 4.56 M cells; LDS float atomics are that slow here.)
__device__ __forceinline__ void lds_add(float* r, float v) { *r = *r + v; }

// residual in LDS: A = x + 9 y + 72 z (x- and y-columns free of bank conflicts, z-columns two-way)
template <int D>
__device__ __forceinline__ int rbase(int ta, int tb) {
    return D == 0 ? 9 * ta + 72 * tb : D == 1 ? ta + 72 * tb : ta + 9 * tb;
}
template <int D>
constexpr int rstride() {
    return D == 0 ? 1 : D == 1 ? 9 : 72;
}

// ---- flux pass along D: the lane's column with its two halo ends; R -= (F_high - F_low) / h for its 8 cells.
// MODE 0: first pass (R = ...), 1: R += ..., 2: last pass (R + ... goes to global memory: z-columns, coalesced dwords).
// The halo registers of this pass were loaded by the caller -- after the flux loop of the pass before, so that nothing of
// a later pass is held in registers while the flux loop runs (168 VGPRs = three waves per SIMD).  Order: face 4 (needs no
// halo value: covers the halo loads), the two sides, then the packed faces.  The mean flux through a FINE side goes
// straight into the residual of its boundary cell in LDS (the packed evaluation of that face is zeroed).
template <int D, int MODE, class Hook>
__device__ __forceinline__ void flux_pass(const BlockDesc3& bb, const int32_t* __restrict__ ftab,
                                          const int32_t* __restrict__ r4tab, const float* __restrict__ P, uint32_t ldp,
                                          float* lds, int lane, const Gas3& gas, Col* Pc, Col& Dc, const Slot& sl0,
                                          const Slot& sl1, int32_t rid0, int32_t rid1, const HaloRegs& h0,
                                          const HaloRegs& h1, float* __restrict__ Rr, uint32_t ldr, Hook&& hook) {
    constexpr int S0 = 2 * D, S1 = 2 * D + 1;
    // lane-only integers (plane positions, LDS and store addresses) are derived again in every pass instead of living in
    // registers -- or in scratch -- across the flux loops: the asm hides that `lane` is the same value as before
    asm volatile("" : "+v"(lane));
    const float rh = bb.rh[D];
    const float qlo = bb.q[S0], qhi = bb.q[S1];
#ifdef S3E_COUNT_SAME_ONLY
    const bool isF0 = false, isF1 = false;
#else
    const bool isF0 = bb.type[S0] == SIDE_FINE, isF1 = bb.type[S1] == SIDE_FINE;
#endif
    float* const Rl = lds + S3E_R + rbase<D>(lane & 7, lane >> 3);
    // cell I of the column: R (op) -(Fhi - Flo) / h; `pre`: the cell already holds the flux of a FINE side (MODE 0)
    auto put = [&](int v, int i, float dF, bool pre = false) {
#ifdef S3E_ABLATE_R  // (timing diagnostic, wrong results: no residual accumulation in LDS)
        if constexpr (MODE == 2) Rr[(size_t)v * ldr + (uint32_t)bb.base + lane + 64 * i] = dF * rh;
        return;
#endif
        float* r = Rl + v * 576 + rstride<D>() * i;
        if constexpr (MODE == 0) {
            if (pre) lds_add(r, -(dF * rh));  // wave-uniform
            else *r = -(dF * rh);
        } else if constexpr (MODE == 1) lds_add(r, -(dF * rh));
#ifdef S3E_NO_NT_STORE   // (A/B)
        else Rr[(size_t)v * ldr + (uint32_t)bb.base + lane + 64 * i] = *r - dF * rh;
#else
        // the residual is not read again by this sweep: stores that do not allocate in L2 (`nt`) leave the cache to the halo
        // lines of the blocks around -- 108.7 -> 105.2 us at 4.56 M cells, 656.8 -> 643.6 at 33.6 M (same box, alternating)
        else __builtin_nontemporal_store(*r - dF * rh, &Rr[(size_t)v * ldr + (uint32_t)bb.base + lane + 64 * i]);
#endif
    };
    // ---- face 4 (between cells 3 and 4) first, straight into the residual of its two cells (nothing of it is held
    // through the flux loop)
    {
        float Pa[5], Pbb[5], Sa[5], Sbb[5], F4[5];
#pragma unroll
        for (int v = 0; v < 5; ++v) {
            const Col& p = Pc[v];
            const float d4 = p.e[0].y - p.e[4].x;
            Pa[v] = p.e[4].x;
            Pbb[v] = p.e[0].y;
            Sa[v] = 0.5f * d4 + 0.5f * (p.e[4].x - p.e[3].x);
            Sbb[v] = 0.5f * (p.e[1].y - p.e[0].y) + 0.5f * d4;
        }
        euler_flux<float, D>(Pa, Pbb, Sa, Sbb, Dc.e[4].x, Dc.e[0].y, 0.5f, gas, F4);
#ifndef S3E_ABLATE_R
#pragma unroll
        for (int v = 0; v < 5; ++v) {
            float* r3 = Rl + v * 576 + rstride<D>() * 3;
            float* r4 = Rl + v * 576 + rstride<D>() * 4;
            const float f = F4[v] * rh;
            if constexpr (MODE == 0) {
                *r3 = -f;
                *r4 = f;
            } else {
                lds_add(r3, -f);
                lds_add(r4, f);
            }
        }
#else
        Dc.e[4].x += F4[0] + F4[1] + F4[2] + F4[3] + F4[4];
#endif
    }
    __builtin_amdgcn_sched_barrier(0);
    const LaneGeo LG = lane_geo(lane);
    // mean halo value behind the boundary cells (own slopes)
    float hm0[5], hm1[5];
#pragma unroll
    for (int v = 0; v < 5; ++v) {
        hm0[v] = h0.hu[v];
        hm1[v] = h1.hu[v];
    }
    if (isF0) {  // wave-uniform
        const int32_t* ft = ftab + (((size_t)bb.fine * 6 + S0) * 64 + lane) * 3;
#pragma unroll
        for (int v = 0; v < 5; ++v) {
            const float* Pv = P + (size_t)v * ldp;
            hm0[v] = 0.25f * ((h0.hu[v] + ldg(Pv, (uint32_t)ft[0])) + (ldg(Pv, (uint32_t)ft[1]) + ldg(Pv, (uint32_t)ft[2])));
        }
    }
    if (isF1) {
        const int32_t* ft = ftab + (((size_t)bb.fine * 6 + S1) * 64 + lane) * 3;
#pragma unroll
        for (int v = 0; v < 5; ++v) {
            const float* Pv = P + (size_t)v * ldp;
            hm1[v] = 0.25f * ((h1.hu[v] + ldg(Pv, (uint32_t)ft[0])) + (ldg(Pv, (uint32_t)ft[1]) + ldg(Pv, (uint32_t)ft[2])));
        }
    }
    // ---- undivided slopes of the two boundary cells (0 and 7); halo cells: slopes along the normal, pressure sensor
    // (FINE sides: the finished face flux, added to the residual of the boundary cell here).  The mean halo values go to
    // the two free ends of the column registers: pairs e[0] = (c[-1], c[4]) and e[4] = (c[3], c[8]) are then register
    // aligned like the others, and nothing else of the halo is held through the flux loop.  (Behind a FINE side the mean
    // stands in for the halo cell in the packed evaluation of that face, whose result is not used.)
    float Sh0[5], Dh0, Dh1;
    float* const park = lds + S3E_BUF + lane;  // slopes of the high halo cell: not needed before the last pair
    {
        float Pb[5], Sb[5], Ff[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int v = 0; v < 5; ++v) {
            const Col& p = Pc[v];
            Pb[v] = p.e[1].x;
            Sb[v] = 0.5f * (p.e[2].x - p.e[1].x) + qlo * (p.e[1].x - hm0[v]);
        }
        side_eval<S0>(bb, LG, ftab, r4tab, P, ldp, lds, lane, sl0, h0.hu, h0.hd, rid0, h0.rv, Pb, Sb, Dc.e[1].x, gas, Sh0,
                      Dh0, Ff);
        if (isF0) {  // cell 0: R -= (F1 - Ff) / h  (the packed value of face 0 is taken out again in the flux loop)
#pragma unroll
            for (int v = 0; v < 5; ++v) {
                float* r = Rl + v * 576;
                if constexpr (MODE == 0) *r = Ff[v] * rh;
                else lds_add(r, Ff[v] * rh);
            }
        }
        float Sh1[5];
#pragma unroll
        for (int v = 0; v < 5; ++v) {
            const Col& p = Pc[v];
            Pb[v] = p.e[3].y;
            Sb[v] = qhi * (hm1[v] - p.e[3].y) + 0.5f * (p.e[3].y - p.e[2].y);
        }
        side_eval<S1>(bb, LG, ftab, r4tab, P, ldp, lds, lane, sl1, h1.hu, h1.hd, rid1, h1.rv, Pb, Sb, Dc.e[3].y, gas, Sh1,
                      Dh1, Ff);
        if (isF1) {  // cell 7: R -= (Ff - F7) / h
#pragma unroll
            for (int v = 0; v < 5; ++v) {
                float* r = Rl + v * 576 + rstride<D>() * 7;
                if constexpr (MODE == 0) *r = -(Ff[v] * rh);
                else lds_add(r, -(Ff[v] * rh));
            }
        }
        wave_lds_sync();  // (the plane of the side is read by other lanes)
#pragma unroll
        for (int v = 0; v < 5; ++v) park[64 * v] = Sh1[v];
    }
    __builtin_amdgcn_sched_barrier(0);
    hook();  // (the halo registers are free from here on: the first loads of the wave's next block go out in the last pass)
    __builtin_amdgcn_sched_barrier(0);
    // column ends
    Dc.e[0].x = Dh0;
    Dc.e[4].y = Dh1;
#pragma unroll
    for (int v = 0; v < 5; ++v) {
        Pc[v].e[0].x = hm0[v];
        Pc[v].e[4].y = hm1[v];
    }
    // slopes of cells (j - 1, j + 4)
    auto slope = [&](int v, int j) -> v2f {
        const Col& p = Pc[v];
        const v2f wl = j == 1 ? v2f{qlo, 0.5f} : v2f{0.5f, 0.5f}, wr = j == 3 ? v2f{0.5f, qhi} : v2f{0.5f, 0.5f};
        return wr * (p.e[j + 1] - p.e[j]) + wl * (p.e[j] - p.e[j - 1]);
    };
    v2f Sc[5];  // slopes e[j] of the pair in hand: starts as e[0] = (Sh0, S4)
#pragma unroll
    for (int v = 0; v < 5; ++v) {
        const Col& p = Pc[v];
        Sc[v] = v2f{Sh0[v], 0.5f * (p.e[1].y - p.e[0].y) + 0.5f * (p.e[0].y - p.e[4].x)};
    }
    // ---- faces (j, j + 5), two at a time
    v2f Fp[5];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v2f Sn[5], F[5];
#pragma unroll
        for (int v = 0; v < 5; ++v) {
            const Col& p = Pc[v];
            if (j < 3) Sn[v] = slope(v, j + 1);
            else {
                const float d4 = p.e[0].y - p.e[4].x;
                Sn[v] = v2f{0.5f * d4 + 0.5f * (p.e[4].x - p.e[3].x), park[64 * v]};
            }
        }
        v2f Pa[5], Pbb[5];
#pragma unroll
        for (int v = 0; v < 5; ++v) {
            Pa[v] = Pc[v].e[j];
            Pbb[v] = Pc[v].e[j + 1];
        }
        const v2f wa = j == 0 ? v2f{1.0f - qlo, 0.5f} : j == 3 ? v2f{0.5f, qhi} : v2f{0.5f, 0.5f};
        euler_flux<v2f, D>(Pa, Pbb, Sc, Sn, Dc.e[j], Dc.e[j + 1], wa, gas, F);
#pragma unroll
        for (int v = 0; v < 5; ++v) {
            if (j == 0 && isF0) lds_add(Rl + v * 576, -(F[v].x * rh));                        // wave-uniform, rare
            if (j == 3 && isF1) lds_add(Rl + v * 576 + rstride<D>() * 7, F[v].y * rh);
            if (j == 0) {
                put(v, 4, F[v].y, true);  // cell 4: faces 4 (in R already) and 5
            } else {                      // cells (j - 1, j + 4)
                const v2f dF = F[v] - Fp[v];
                put(v, j - 1, dF.x, j == 1 && isF0);
                put(v, j + 4, dF.y, j == 3 && isF1);
            }
            if (j == 3) put(v, 3, -F[v].x, true);  // cell 3: faces 3 and 4 (in R already)
            Fp[v] = F[v];
            Sc[v] = Sn[v];
        }
    }
}

template <int I = 0>
__device__ __forceinline__ void load_zcol(const float* __restrict__ p, Col& c) {
    if constexpr (I < 8) {
        set_cell<I>(c, p[64 * I]);
        load_zcol<I + 1>(p, c);
    }
}

// ---- the first loads of a block -- those its sensor needs, and the rim ids: the pressure as z-columns (8 rows of 64), the
// halo pressures (6), the rim ids of the six sides (6) -- go to a wave-private LDS buffer by LDS-DMA (global_load_lds_dword:
// a per-lane global address, row base + 4 * lane in LDS, no VGPR for the data).  In a chain of blocks they are requested
// during the z fluxes of the block before and cost that block neither registers nor a wait.
#define S3E_NEXT_ROWS 20
#define S3E_NEXT (64 * S3E_NEXT_ROWS)  // floats per wave
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
// row[lane] = base[idx] (4-byte elements; uniform base + 32-bit lane offset: the SGPR-base form of the instruction)
__device__ __forceinline__ void dma_row(const void* base, uint32_t idx, float* row) {
    __builtin_amdgcn_global_load_lds((gptr_t)((const char*)base + (size_t)(idx << 2)), (lptr_t)row, 4, 0, 0);
}
__device__ __forceinline__ void request_first(const BlockDesc3* __restrict__ blocks, const int32_t* __restrict__ htab,
                                              const int32_t* __restrict__ rtab, const float* __restrict__ P, int32_t blk,
                                              int lane, float* nextbuf) {
    const BlockDesc3 bn = blocks[blk];
    // halo cell ids first (a table load where a side is FINE or faces a fragment)
    uint32_t hid[6];
    hid[0] = halo_cell3s<0>(bn, htab, blk, lane);
    hid[1] = halo_cell3s<1>(bn, htab, blk, lane);
    hid[2] = halo_cell3s<2>(bn, htab, blk, lane);
    hid[3] = halo_cell3s<3>(bn, htab, blk, lane);
    hid[4] = halo_cell3s<4>(bn, htab, blk, lane);
    hid[5] = halo_cell3s<5>(bn, htab, blk, lane);
#ifdef S3E_ABLATE_HALO
#pragma unroll
    for (int s = 0; s < 6; ++s) hid[s] = (uint32_t)bn.base + 64u * s + (uint32_t)lane;
#endif
    const uint32_t c0 = (uint32_t)bn.base + (uint32_t)lane;
#pragma unroll
    for (int i = 0; i < 8; ++i) dma_row(P, c0 + 64u * i, nextbuf + 64 * i);
#pragma unroll
    for (int s = 0; s < 6; ++s) dma_row(P, hid[s], nextbuf + 64 * (8 + s));
    const int32_t* rt = rtab + (size_t)blk * 384;  // (uniform)
#pragma unroll
    for (int s = 0; s < 6; ++s) dma_row(rt, (uint32_t)(64 * s + lane), nextbuf + 64 * (14 + s));
}

// One block whose first loads are in (or on their way to) `nextbuf`; `next` requests those of the wave's next block
// (called in the z pass, when the buffer has long been read).
// STAMP: phase time stamps of the wave (100 MHz ticks) for scripts/wave_timeline_3d.py: 0 start, 1 first loads landed,
// 2 sensor done, 3 x fluxes, 4 transposed, 5 y fluxes, 6 transposed, 7 end
template <bool STAMP, bool DMA, class Next>
__device__ __forceinline__ void sweep_block(const BlockDesc3* __restrict__ blocks, const int32_t* __restrict__ htab,
                                            const int32_t* __restrict__ ftab, const int32_t* __restrict__ rtab,
                                            const int32_t* __restrict__ r4tab, int32_t blk, const float* __restrict__ P,
                                            uint32_t ldp, float* __restrict__ Rr, uint32_t ldr, const Gas3& gas, float* lds,
                                            int lane, unsigned long long* stamps, const int32_t* __restrict__ dtab,
                                            const float* nextbuf, Next&& next) {
    auto stamp = [&](int k) {
        if constexpr (STAMP) {
            __builtin_amdgcn_sched_barrier(0);
            if (k == 1 || k == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (stamps && lane == 0) stamps[k] = __builtin_amdgcn_s_memrealtime();
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    stamp(0);
    const BlockDesc3 bb = blocks[blk];
    const int ta = lane & 7, tb = lane >> 3;
    float* buf = lds + S3E_BUF;
    // ---- DMA = false (one block per wave: nothing to prefetch for): the first loads of the block -- rim ids, the pressure
    // as z-columns, the halo pressures -- as plain loads, ahead of everything else (the LDS-DMA form costs ~100 vector and
    // ~400 scalar instructions of address and M0 handling per block)
    int32_t ridk[6] = {0, 0, 0, 0, 0, 0};
    Col pz;
    float hp[6];
    if constexpr (!DMA) {
        uint32_t hid[6];
        hid[0] = halo_cell3s<0>(bb, htab, blk, lane);
        hid[1] = halo_cell3s<1>(bb, htab, blk, lane);
        hid[2] = halo_cell3s<2>(bb, htab, blk, lane);
        hid[3] = halo_cell3s<3>(bb, htab, blk, lane);
        hid[4] = halo_cell3s<4>(bb, htab, blk, lane);
        hid[5] = halo_cell3s<5>(bb, htab, blk, lane);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 6; ++s) ridk[s] = rtab[(size_t)blk * 384 + s * 64 + lane];
        load_zcol(P + (uint32_t)bb.base + lane, pz);
#pragma unroll
        for (int s = 0; s < 6; ++s) hp[s] = ldg(P, hid[s]);
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- the halo values of the x sides and the five primitives as x-columns (two float4 per field): requested now, in
    // flight during the sensor
    HaloRegs h0, h1;
    Slot sl0 = slot_of<0>(bb, htab, blk, lane, dtab), sl1 = slot_of<1>(bb, htab, blk, lane, dtab);
    halo_load_values<0>(bb, lane, P, ldp, sl0, h0);
    halo_load_values<1>(bb, lane, P, ldp, sl1, h1);
    __builtin_amdgcn_sched_barrier(0);
    Col Pc[5];
    {
        const uint32_t a0 = (uint32_t)bb.base + 8u * (uint32_t)lane;
#pragma unroll
        for (int v = 0; v < 5; ++v) {
            const float* Pv = P + (size_t)v * ldp + a0;
            const v4f lo = *(const v4f_g*)Pv, hi = *(const v4f_g*)(Pv + 4);
            Pc[v].e[1].x = lo.x; Pc[v].e[2].x = lo.y; Pc[v].e[3].x = lo.z; Pc[v].e[4].x = lo.w;
            Pc[v].e[0].y = hi.x; Pc[v].e[1].y = hi.y; Pc[v].e[2].y = hi.z; Pc[v].e[3].y = hi.w;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- the buffer of the first loads: the DMA rows are older than the loads just issued (vector memory operations
    // complete in order): at least 5 + 5 halo gathers and 10 x-column loads
    int32_t rid0, rid1;
    if constexpr (DMA) {
        asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
        const int32_t* nrid = (const int32_t*)(nextbuf + 64 * 14) + lane;
        rid0 = nrid[0];
        rid1 = nrid[64];
        const float* np_ = nextbuf + lane;
        set_cell<0>(pz, np_[0]); set_cell<1>(pz, np_[64]); set_cell<2>(pz, np_[128]); set_cell<3>(pz, np_[192]);
        set_cell<4>(pz, np_[256]); set_cell<5>(pz, np_[320]); set_cell<6>(pz, np_[384]); set_cell<7>(pz, np_[448]);
#pragma unroll
        for (int s = 0; s < 6; ++s) hp[s] = np_[64 * (8 + s)];
    } else {
        rid0 = ridk[0];
        rid1 = ridk[1];
    }
    h0.rv = ldg(P, (uint32_t)(rid0 >= 0 ? rid0 : bb.base));
    h1.rv = ldg(P, (uint32_t)(rid1 >= 0 ? rid1 : bb.base));
    __builtin_amdgcn_sched_barrier(0);
    stamp(1);
    // ---- pressure sensor of the block's cells: z, y, x
    Col Dc;
    {
        Col p = pz, N, Dn;
#ifdef S3E_ABLATE_SENSOR  // (timing diagnostic, wrong results: no sensor passes)
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            N.e[j] = p.e[j] * 1e-9f + hp[j];
            Dn.e[j] = v2f{1.0f, 1.0f};
        }
        if (false)
#endif
        {
        sensor_pass<2, true>(bb, ftab, P, lane, hp[4], hp[5], p, N, Dn);
        transpose<2, 1>(buf, ta, tb, p);
        transpose<2, 1>(buf, ta, tb, N);
        transpose<2, 1>(buf, ta, tb, Dn);
        sensor_pass<1, false>(bb, ftab, P, lane, hp[2], hp[3], p, N, Dn);
        transpose<1, 0>(buf, ta, tb, p);
        transpose<1, 0>(buf, ta, tb, N);
        transpose<1, 0>(buf, ta, tb, Dn);
        sensor_pass<0, false>(bb, ftab, P, lane, hp[0], hp[1], p, N, Dn);
        }
#pragma unroll
        for (int j = 1; j < 4; ++j)
            Dc.e[j] = v2f{fmaxf(N.e[j].x * __builtin_amdgcn_rcpf(Dn.e[j].x), 1e-7f),
                          fmaxf(N.e[j].y * __builtin_amdgcn_rcpf(Dn.e[j].y), 1e-7f)};
        Dc.e[4].x = fmaxf(N.e[4].x * __builtin_amdgcn_rcpf(Dn.e[4].x), 1e-7f);
        Dc.e[0].y = fmaxf(N.e[0].y * __builtin_amdgcn_rcpf(Dn.e[0].y), 1e-7f);
    }
    stamp(2);
    // ---- fluxes: x, y, z.  The halo values of a pass are requested before the flux loop of the pass before (after its
    // side evaluations, when its own halo registers are free).
    // (The block descriptor is read again for every pass -- scalar loads -- instead of holding its 38 words in SGPRs
    // throughout: with it the kernel ran out of SGPRs and kept the five field pointers in VGPRs.)
    asm volatile("" : "+s"(blk));
    const BlockDesc3 bx = blocks[blk];
    HaloRegs g0, g1;  // the halo registers of the next pass: requested before the flux loop of the pass in hand
    Slot tl0, tl1;
    int32_t tid0, tid1;
    flux_pass<0, 0>(bx, ftab, r4tab, P, ldp, lds, lane, gas, Pc, Dc, sl0, sl1, rid0, rid1, h0, h1, Rr, ldr, [&]() {
        tid0 = DMA ? ((const int32_t*)(nextbuf + 64 * 16))[lane] : ridk[2];
        tid1 = DMA ? ((const int32_t*)(nextbuf + 64 * 17))[lane] : ridk[3];
        tl0 = slot_of<2>(bx, htab, blk, lane, dtab);
        tl1 = slot_of<3>(bx, htab, blk, lane, dtab);
        halo_load<2>(bx, lane, P, ldp, tl0, tid0, g0);
        halo_load<3>(bx, lane, P, ldp, tl1, tid1, g1);
    });
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" : "+s"(blk));
    const BlockDesc3 by = blocks[blk];
    stamp(3);
#pragma unroll
    for (int v = 0; v < 5; ++v) transpose<0, 1>(buf, ta, tb, Pc[v]);
    transpose<0, 1>(buf, ta, tb, Dc);
    stamp(4);
    flux_pass<1, 1>(by, ftab, r4tab, P, ldp, lds, lane, gas, Pc, Dc, tl0, tl1, tid0, tid1, g0, g1, Rr, ldr, [&]() {
        rid0 = DMA ? ((const int32_t*)(nextbuf + 64 * 18))[lane] : ridk[4];
        rid1 = DMA ? ((const int32_t*)(nextbuf + 64 * 19))[lane] : ridk[5];
        sl0 = slot_of<4>(by, htab, blk, lane, dtab);
        sl1 = slot_of<5>(by, htab, blk, lane, dtab);
        halo_load<4>(by, lane, P, ldp, sl0, rid0, h0);
        halo_load<5>(by, lane, P, ldp, sl1, rid1, h1);
    });
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" : "+s"(blk));
    const BlockDesc3 bz = blocks[blk];
    stamp(5);
#pragma unroll
    for (int v = 0; v < 5; ++v) transpose<1, 2>(buf, ta, tb, Pc[v]);
    transpose<1, 2>(buf, ta, tb, Dc);
    stamp(6);
    flux_pass<2, 2>(bz, ftab, r4tab, P, ldp, lds, lane, gas, Pc, Dc, sl0, sl1, rid0, rid1, h0, h1, Rr, ldr, next);
    stamp(7);
}

// A chain of blocks first, first + stride, ... < end for one wave: the first loads of a block are requested while the z
// fluxes of the block before are computed.
template <bool STAMP = false, bool DMA = true>
__device__ __forceinline__ void sweep_euler_chain(const BlockDesc3* __restrict__ blocks, const int32_t* __restrict__ htab,
                                                  const int32_t* __restrict__ ftab, const int32_t* __restrict__ rtab,
                                                  const int32_t* __restrict__ r4tab, int32_t first, int32_t stride,
                                                  int32_t end, const float* __restrict__ P, uint32_t ldp,
                                                  float* __restrict__ Rr, uint32_t ldr, Gas3 gas, float* lds,
                                                  float* nextbuf, int lane, unsigned long long* stamps = nullptr,
                                                  const int32_t* __restrict__ dtab = nullptr) {
    if (first >= end) return;
    if constexpr (!DMA) {   // one block per wave
        sweep_block<STAMP, false>(blocks, htab, ftab, rtab, r4tab, first, P, ldp, Rr, ldr, gas, lds, lane,
                                  STAMP && stamps ? stamps + (size_t)first * 8 : nullptr, dtab, nextbuf, []() {});
        return;
    }
    request_first(blocks, htab, rtab, P, first, lane, nextbuf);
#pragma unroll 1
    for (int32_t blk = first; blk < end; blk += stride) {
        const int32_t nb = blk + stride;
        sweep_block<STAMP, true>(blocks, htab, ftab, rtab, r4tab, blk, P, ldp, Rr, ldr, gas, lds, lane,
                           STAMP && stamps ? stamps + (size_t)blk * 8 : nullptr, dtab, nextbuf, [&]() {
                               if (nb < end) request_first(blocks, htab, rtab, P, nb, lane, nextbuf);  // wave-uniform
                           });
    }
}

#pragma clang fp contract(off)

}  // namespace strip3e
