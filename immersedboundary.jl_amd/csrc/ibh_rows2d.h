// Row / column sweep (2-D scalar advection-JST-MUSCL residual, test/advection.jl:67-83): ONE wavefront sweeps EIGHT 8x8
// blocks; a lane owns a whole row of one block (x direction), then -- after a transposition through wave-private LDS --
// a whole column (y direction).  The 2-D counterpart of the 3-D column sweep (ibh_strip3d_euler.h):
//   * every face of a row, its two block faces included, has both cells, their slopes and sensors in the lane's own
//     registers: no LDS tiles, no cross-lane traffic in the flux loop;
//   * the lane of a row IS the lane of the halo slots at its two ends (both sub-faces k = 0, 1 of boundary cell t = row:
//     arithmetic on (k0, k1) pairs exactly as in the quad sweep, ibh_quad2d.h);
//   * row registers f[j] = (c[j], c[j + 4]): faces (j + 1, j + 5), slopes, sensors and residuals of cells (j, j + 4) are
//     packed evaluations on register-aligned pairs; face 4 (between f[3].x and f[0].y) is a scalar one;
//   * halo cell ids are ARITHMETIC on the neighbour block bases of the descriptor (SAME / COARSE / FINE / MIRROR sides
//     towards complete blocks: every block of a one-partition mesh) -- no 256-byte halo table row per block; only the
//     16 end-cell ids come from a table (ibh_analyze.cpp step 6);
//   * any complete block qualifies: there are no "single blocks" beside 2x2 groups.
// Arithmetic: blk2::sweep_adv (ibh_sweep2d.h) / quad2::quad_compute (ibh_quad2d.h): undivided slopes, flux_w, one
// reciprocal per sensor.
// STATUS: correct (tests/test_gpu_residual.py::test_row_sweep) and 15 % fewer vector instructions than quads + single
// blocks (881 per 512 cells), but measured SLOWER than the quad sweep -- 8.4 against 6.1 us at 0.87 M cells, 21.7 against
// 17.0 us at 3.47 M (scripts/probe_rows.py): 1 695 long waves at 0.87 M cells are 1.7 per SIMD, 122 VGPRs allow four, and a
// wave issues 49 global loads (24 halo gathers of 4 bytes).  Opt-in: IBH_ROWS=1 / ibh_set_tuning("rows", 1).
//
// Lane L: block slot g = L >> 3 (block first + g), i = L & 7 = row y (x pass) / column x (y pass) = slot t of the two
// sides normal to the pass direction.
// order: x rows (float4 loads) -> sensor x -> transposed -> sensor y, D -> y fluxes -> D, R transposed back -> x fluxes ->
// float4 stores.
#pragma once
#include "ibh_quad2d.h"

namespace rows2 {

#pragma clang fp contract(fast)

using blk2::ldg;
using blk2::wave_lds_sync;
using quad2::jst_max2;
using quad2::max3;
using quad2::v2f;
using quad2::v2i;
using quad2::v4f;
using quad2::v4f_g;
typedef int v4i_g __attribute__((ext_vector_type(4), aligned(4)));

// LDS per wave (floats)
#define ROWS_TR 0                     // transposition buffer: A = 72 g + 9 y + x (both layouts free of bank conflicts)
#define ROWS_EXT 576                  // lateral lines [8 blocks][4 sides][20]: [lo end k=0,1 | 16 slots | hi end k=0,1]
#define ROWS_LDS (ROWS_EXT + 8 * 80)  // 1 216 floats = 4.75 KB per wave

struct Row {
    v2f f[4];  // f[j] = (c[j], c[j + 4])
};
template <int I>
__device__ __forceinline__ float rowv(const Row& r) {
    if constexpr (I < 4) return r.f[I].x;
    else return r.f[I - 4].y;
}
template <int I>
__device__ __forceinline__ void row_set(Row& r, float v) {
    if constexpr (I < 4) r.f[I].x = v;
    else r.f[I - 4].y = v;
}
template <int STRIDE, int I = 0>
__device__ __forceinline__ void tr_write(float* w, const Row& r) {
    if constexpr (I < 8) {
        w[STRIDE * I] = rowv<I>(r);
        tr_write<STRIDE, I + 1>(w, r);
    }
}
template <int STRIDE, int I = 0>
__device__ __forceinline__ void tr_read(const float* p, Row& r) {
    if constexpr (I < 8) {
        row_set<I>(r, p[STRIDE * I]);
        tr_read<STRIDE, I + 1>(p, r);
    }
}
// rows -> columns (TO_Y) or back, inside each of the wave's eight blocks
template <bool TO_Y>
__device__ __forceinline__ void transpose(float* buf, int g, int i, Row& r) {
    float* px = buf + 72 * g + 9 * i;  // row i: stride 1
    float* py = buf + 72 * g + i;      // column i: stride 9
    if (TO_Y) tr_write<1>(px, r);
    else tr_write<9>(py, r);
    wave_lds_sync();
    if (TO_Y) tr_read<9>(py, r);
    else tr_read<1>(px, r);
    wave_lds_sync();
}

// what a lane needs of its block's descriptor (BlockDesc2, ibh_common.h), as vector registers: the eight blocks of a
// wave differ
struct Blk {
    int32_t base;
    int ty[4];
    int32_t nb[4][2];
    int sub[4];
    float rh[2];
    float q[4];
};
static_assert(offsetof(BlockDesc2, type) == 4 && offsetof(BlockDesc2, nb) == 20 && offsetof(BlockDesc2, sub) == 52 &&
                  offsetof(BlockDesc2, rh) == 76 && offsetof(BlockDesc2, q) == 84,
              "rows2::load_blk reads the descriptor by offset");
__device__ __forceinline__ Blk load_blk(const BlockDesc2* __restrict__ blocks, int32_t blk) {
    const int32_t* bw = (const int32_t*)(blocks + blk);
    const v4i_g a = *(const v4i_g*)bw, b = *(const v4i_g*)(bw + 4), c = *(const v4i_g*)(bw + 8),
                d = *(const v4i_g*)(bw + 12), f = *(const v4i_g*)(bw + 19);
    const int32_t q2 = bw[23], q3 = bw[24], s3 = bw[16];
    Blk B;
    B.base = a.x;
    B.ty[0] = a.y; B.ty[1] = a.z; B.ty[2] = a.w; B.ty[3] = b.x;
    B.nb[0][0] = b.y; B.nb[0][1] = b.z; B.nb[1][0] = b.w; B.nb[1][1] = c.x;
    B.nb[2][0] = c.y; B.nb[2][1] = c.z; B.nb[3][0] = c.w; B.nb[3][1] = d.x;
    B.sub[0] = d.y; B.sub[1] = d.z; B.sub[2] = d.w; B.sub[3] = s3;
    B.rh[0] = __int_as_float(f.x); B.rh[1] = __int_as_float(f.y);
    B.q[0] = __int_as_float(f.z); B.q[1] = __int_as_float(f.w);
    B.q[2] = __int_as_float(q2); B.q[3] = __int_as_float(q3);
    return B;
}

// halo cells behind boundary cell t of side S (sub-faces k = 0, 1; one cell twice unless the side is FINE), the
// arithmetic of ibh_analyze.cpp step 3: the neighbour's cell on the opposite edge at tangential position t (SAME),
// t / 2 + 4 sub (COARSE), 2 (t & 3) + k in block t >> 2 (FINE); the boundary cell itself on a MIRROR side
template <int S>
__device__ __forceinline__ v2i halo_ids(const Blk& B, int t) {
    constexpr int d = S >> 1;
    constexpr bool low = (S & 1) == 0;
    constexpr int sd = d == 0 ? 1 : 8, st = d == 0 ? 8 : 1;
    constexpr int opp = low ? 7 * sd : 0, own = low ? 0 : 7 * sd;
    const int ty = B.ty[S];
    const bool isC = ty == SIDE_COARSE, isF = ty == SIDE_FINE, mirror = ty == SIDE_MIRROR;
    const int tt = isC ? (t >> 1) + 4 * B.sub[S] : isF ? 2 * (t & 3) : t;
    const int32_t nbb = (isF && (t >> 2)) ? B.nb[S][1] : B.nb[S][0];
    const int32_t id0 = mirror ? B.base + own + t * st : nbb + opp + tt * st;
    return v2i{id0, isF ? id0 + st : id0};
}

// one side's slot of a lane: halo values (k0, k1), the cells one step deeper, the normal velocity there
struct Slot2 {
    v2f hu, hd, hc;
};
template <int S>
__device__ __forceinline__ Slot2 slot_load(const Blk& B, int t, const float* __restrict__ u, const float* __restrict__ Cn) {
    constexpr int d = S >> 1;
    constexpr bool low = (S & 1) == 0;
    constexpr int sd = d == 0 ? 1 : 8;
    const v2i id = halo_ids<S>(B, t);
    const int dd = B.ty[S] == SIDE_MIRROR ? 0 : (low ? -sd : sd);
    Slot2 s;
    s.hu = v2f{ldg(u, (uint32_t)id.x), ldg(u, (uint32_t)id.y)};
    s.hd = v2f{ldg(u, (uint32_t)(id.x + dd)), ldg(u, (uint32_t)(id.y + dd))};
    s.hc = v2f{ldg(Cn, (uint32_t)id.x), ldg(Cn, (uint32_t)id.y)};
    return s;
}

__device__ __forceinline__ float dpp_xor1(float v) {  // lane i <- lane i ^ 1 (quad_perm [1,0,3,2])
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));
}

// interior faces (both cells same level, wa = 1/2), two at a time (quad2::flux_half4 on pairs)
__device__ __forceinline__ v2f flux_half2(v2f ua, v2f ub, v2f Sa, v2f Sb, v2f Da, v2f Db, v2f Ca, v2f Cb) {
    const v2f d = ub - ua;
    const v2f gu = Sa - 0.5f * d;
    const v2f Du = Sb - 0.5f * d;
    const v2f s = v2f{__builtin_amdgcn_fmed3f(Du.x, gu.x, 0.0f), __builtin_amdgcn_fmed3f(Du.y, gu.y, 0.0f)};
    const v2f Df = v2f{max3(Da.x, Db.x, 1e-7f), max3(Da.y, Db.y, 1e-7f)};
    const v2f t16 = (Sa - Sb) * 0.0625f;
    const v2f uf = (ua + 0.5f * d) + t16;
    const v2f A = uf - Df * t16;
    const v2f Cf = 0.5f * (Ca + Cb);
    const v2f B = Df * (s - 0.5f * d);
    const v2f CA = Cf * A;
    return v2f{fmaf(fabsf(Cf.x), B.x, CA.x), fmaf(fabsf(Cf.y), B.y, CA.y)};
}
__device__ __forceinline__ float flux_half1(float ua, float ub, float Sa, float Sb, float Da, float Db, float Ca, float Cb) {
    const float d = ub - ua;
    const float gu = Sa - 0.5f * d;
    const float Du = Sb - 0.5f * d;
    const float s = __builtin_amdgcn_fmed3f(Du, gu, 0.0f);
    const float Df = max3(Da, Db, 1e-7f);
    const float t16 = (Sa - Sb) * 0.0625f;
    const float uf = (ua + 0.5f * d) + t16;
    const float A = uf - Df * t16;
    const float Cf = 0.5f * (Ca + Cb);
    const float B = Df * (s - 0.5f * d);
    return fmaf(fabsf(Cf), B, Cf * A);
}

// first differences of a row with its two halo ends: dL[j], dR[j] = differences of cells (j, j + 4) to their low / high
// neighbours
struct Diffs {
    v2f dL[4], dR[4];
};
__device__ __forceinline__ Diffs row_diffs(const Row& u, float hm0, float hm1) {
    Diffs D;
    const float d4 = u.f[0].y - u.f[3].x;
#pragma unroll
    for (int j = 0; j < 3; ++j) D.dR[j] = u.f[j + 1] - u.f[j];
    D.dR[3] = v2f{d4, hm1 - u.f[3].y};
    D.dL[0] = v2f{u.f[0].x - hm0, d4};
#pragma unroll
    for (int j = 1; j < 4; ++j) D.dL[j] = D.dR[j - 1];
    return D;
}

// JST ratio of the row's cells along the row: numerator n = |second difference| / h + 1e-7, denominator d = sum of
// |first differences| / h + 1e-7 (ha0 / ha1: mean |halo - cell| behind cells 0 / 7: two finer cells count both)
__device__ __forceinline__ void sensor_row(const Diffs& D, float ha0, float ha1, float rh, Row& N, Row& Dn) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const v2f g = D.dR[j] - D.dL[j];
        const float aLx = j == 0 ? ha0 : fabsf(D.dL[j].x), aRy = j == 3 ? ha1 : fabsf(D.dR[j].y);
        const float ax = fabsf(D.dR[j].x) + aLx, ay = aRy + fabsf(D.dL[j].y);
        N.f[j] = v2f{fmaf(fabsf(g.x), rh, 1e-7f), fmaf(fabsf(g.y), rh, 1e-7f)};
        Dn.f[j] = v2f{fmaf(ax, rh, 1e-7f), fmaf(ay, rh, 1e-7f)};
    }
}

// ---- halo cells of slot t of one side (both sub-faces): slope along the normal, sensor (lateral neighbours from the
// side's line: quad2::quad_compute / blk2::sweep_adv), the two sub-face fluxes; returns the mean flux through the block
// face of boundary cell t (owner towards -)
template <bool LOW>
__device__ __forceinline__ float slot_flux(int ty, float qs, float rhn, float rht, const float* line, int t,
                                           const Slot2& s, float m0, float So, float Do, float Co) {
    const bool isC = ty == SIDE_COARSE, isF = ty == SIDE_FINE, mirror = ty == SIDE_MIRROR;
    const v2f hu = s.hu, hd = s.hd;
    const float m1x = dpp_xor1(m0);                 // boundary cell of the pair mate t ^ 1 (COARSE: same halo cell)
    const float m1 = isC ? m1x : m0;
    const float irt = isC ? 0.5f : isF ? 2.0f : 1.0f;  // h / h_halo
    const float ihn = rhn * irt, iht = rht * irt;
    const v2f dm0 = m0 - hu, dm1 = m1 - hu, dde = hd - hu;
    const v2f din = 0.5f * (dm0 + dm1);
    v2f Shn = 0.5f * dde - (1.0f - qs) * din;       // slope seen from the block outwards (sign folded below)
    // lateral neighbours (ibh_sweep2d.h in pair form, as quad2::quad_compute)
    const float* pl = line + (isC ? 2 * (t & ~1) : 2 * t);
    const float* ph = line + (isC ? 2 * (t | 1) + 4 : 2 * t + 4);
    v2f Lp = *(const v2f*)pl, Hp = *(const v2f*)ph;
    asm volatile("" : "+v"(Lp), "+v"(Hp));
    const bool t0 = t == 0, t7 = t == 7;
    const v2f lo0 = v2f{(isF && !t0) ? Lp.y : Lp.x, isF ? hu.x : Lp.x};
    const v2f lo1 = v2f{Lp.y, isF ? hu.x : Lp.y};
    const v2f hi0 = v2f{isF ? hu.y : Hp.x, Hp.x};
    const v2f hi1 = v2f{isF ? hu.y : Hp.y, (isF && !t7) ? Hp.x : Hp.y};
    const v2f e0 = lo0 - hu, e1 = lo1 - hu, e2 = hi0 - hu, e3 = hi1 - hu;
    const v2f gn = (dm0 + dm1) + (dde + dde);
    const v2f gt = (e0 + e1) + (e2 + e3);
    const float hn = 0.5f * ihn, ht = 0.5f * iht;
    v2f Dh;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float an = (fabsf(dm0[k]) + fabsf(dm1[k])) + 2.0f * fabsf(dde[k]);
        const float at = (fabsf(e0[k]) + fabsf(e1[k])) + (fabsf(e2[k]) + fabsf(e3[k]));
        Dh[k] = jst_max2(gn[k], an, hn, gt[k], at, ht);
    }
    constexpr float sgn = LOW ? -1.0f : 1.0f;
    // MIRROR side (domain boundary): the halo cell is the boundary cell itself, with its slope and sensor
    Shn = mirror ? v2f{sgn * So, sgn * So} : Shn;
    Dh = mirror ? v2f{Do, Do} : Dh;
    // own cell first, halo second; on a low side everything that is odd under the swap is negated
    const v2f F = quad2::flux_w2(m0, hu, sgn * So, Shn, Do, Dh, sgn * Co, sgn * s.hc, qs);
    return sgn * (0.5f * (F.x + F.y));
}

// ---- fluxes along the rows held now (direction DIR): R = -(F_high - F_low) / h of the row's cells (ADD: R -= ...)
template <int DIR, bool ADD>
__device__ __forceinline__ void flux_rows(const Blk& B, const Row& u, const Row& Cn, const Row& Dc, const Diffs& Df,
                                          const Slot2& s0, const Slot2& s1, const float* line0, int i, Row& R) {
    constexpr int S0 = 2 * DIR, S1 = 2 * DIR + 1;
    const float rh = B.rh[DIR], rht = B.rh[1 - DIR];
    const float qlo = B.q[S0], qhi = B.q[S1];
    // undivided slopes
    Row S;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const v2f wl = j == 0 ? v2f{qlo, 0.5f} : v2f{0.5f, 0.5f}, wr = j == 3 ? v2f{0.5f, qhi} : v2f{0.5f, 0.5f};
        S.f[j] = wr * Df.dR[j] + wl * Df.dL[j];
    }
    // block faces
    const float eL = slot_flux<true>(B.ty[S0], qlo, rh, rht, line0, i, s0, u.f[0].x, S.f[0].x, Dc.f[0].x, Cn.f[0].x);
    const float eR = slot_flux<false>(B.ty[S1], qhi, rh, rht, line0 + 20, i, s1, u.f[3].y, S.f[3].y, Dc.f[3].y, Cn.f[3].y);
    // interior faces (j + 1, j + 5), face 4
    v2f Fp[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) Fp[j] = flux_half2(u.f[j], u.f[j + 1], S.f[j], S.f[j + 1], Dc.f[j], Dc.f[j + 1], Cn.f[j], Cn.f[j + 1]);
    const float F4 = flux_half1(u.f[3].x, u.f[0].y, S.f[3].x, S.f[0].y, Dc.f[3].x, Dc.f[0].y, Cn.f[3].x, Cn.f[0].y);
    // Green-Gauss: cells (j, j + 4): faces (j, j + 4) below, (j + 1, j + 5) above
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const v2f hi = j < 3 ? Fp[j] : v2f{F4, eR};
        const v2f lo = j == 0 ? v2f{eL, F4} : Fp[j - 1];
        if (ADD) R.f[j] -= (hi - lo) * rh;
        else R.f[j] = -((hi - lo) * rh);
    }
}

// mean halo value and mean |halo - boundary cell| behind a boundary cell
__device__ __forceinline__ void slot_means(const Slot2& s, float m, float& hm, float& ha) {
    hm = 0.5f * (s.hu.x + s.hu.y);
    ha = 0.5f * (fabsf(s.hu.x - m) + fabsf(s.hu.y - m));
}

// ---- eight blocks first .. first + 7 (n blocks in all; entries of `list` if given: any eight blocks) by one wave
__device__ __forceinline__ void sweep_rows(const BlockDesc2* __restrict__ blocks, const int32_t* __restrict__ etab,
                                           int32_t first, int32_t n, const float* __restrict__ u,
                                           const float* __restrict__ C, uint32_t ldc, float* __restrict__ ud, float* lds,
                                           int lane, const int32_t* __restrict__ list = nullptr) {
    const int g = lane >> 3, i = lane & 7;
    const bool valid = first + g < n;
    const int32_t pos = valid ? first + g : n - 1;  // (lanes past the end repeat the last block and store nothing)
    const int32_t blk = list ? list[pos] : pos;
    const float* Cy = C + ldc;
    const Blk B = load_blk(blocks, blk);
    const int32_t* et = etab + (size_t)blk * 16;
    const uint32_t eidx = (uint32_t)et[i], eidy = (uint32_t)et[8 + i];
    // ---- loads: rows of u and Cx (two float4 each), the columns of Cy, the four slots, the end cells
    const uint32_t ax = (uint32_t)B.base + 8u * (uint32_t)i;
    const v4f ulo = *(const v4f_g*)(u + ax), uhi = *(const v4f_g*)(u + ax + 4);
    const v4f clo = *(const v4f_g*)(C + ax), chi = *(const v4f_g*)(C + ax + 4);
    const Slot2 sL = slot_load<0>(B, i, u, C), sR = slot_load<1>(B, i, u, C);
    const float eux = ldg(u, eidx);
    Row Cyc;
    {
        const float* p = Cy + (uint32_t)B.base + i;
        Cyc.f[0] = v2f{p[0], p[32]};
        Cyc.f[1] = v2f{p[8], p[40]};
        Cyc.f[2] = v2f{p[16], p[48]};
        Cyc.f[3] = v2f{p[24], p[56]};
    }
    const Slot2 sB = slot_load<2>(B, i, u, Cy), sT = slot_load<3>(B, i, u, Cy);
    const float euy = ldg(u, eidy);
    Row Ux, Cxr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        Ux.f[j] = v2f{ulo[j], uhi[j]};
        Cxr.f[j] = v2f{clo[j], chi[j]};
    }
    // ---- lateral lines of the four sides: [lo end k=0,1 | slot entries 2 t + k | hi end k=0,1]
    float* ext = lds + ROWS_EXT + 80 * g;
    {
        const int epos = (i >> 2) * 20 + ((i >> 1) & 1) * 18 + (i & 1);
        *(v2f*)(ext + 2 + 2 * i) = sL.hu;
        *(v2f*)(ext + 22 + 2 * i) = sR.hu;
        *(v2f*)(ext + 42 + 2 * i) = sB.hu;
        *(v2f*)(ext + 62 + 2 * i) = sT.hu;
        ext[epos] = eux;
        ext[40 + epos] = euy;
    }
    // ---- sensor: x, transposed, y
    float* buf = lds + ROWS_TR;
    float hm0, ha0, hm1, ha1;
    slot_means(sL, Ux.f[0].x, hm0, ha0);
    slot_means(sR, Ux.f[3].y, hm1, ha1);
    const Diffs Dx = row_diffs(Ux, hm0, hm1);
    Row N, Dn;
    sensor_row(Dx, ha0, ha1, B.rh[0], N, Dn);
    Row Uy = Ux;
    transpose<true>(buf, g, i, Uy);
    transpose<true>(buf, g, i, N);
    transpose<true>(buf, g, i, Dn);
    slot_means(sB, Uy.f[0].x, hm0, ha0);
    slot_means(sT, Uy.f[3].y, hm1, ha1);
    const Diffs Dy = row_diffs(Uy, hm0, hm1);
    Row Dc;
    {
        Row Ny, Dny;
        sensor_row(Dy, ha0, ha1, B.rh[1], Ny, Dny);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const v2f num = v2f{fmaxf(N.f[j].x * Dny.f[j].x, Ny.f[j].x * Dn.f[j].x),
                                fmaxf(N.f[j].y * Dny.f[j].y, Ny.f[j].y * Dn.f[j].y)};
            const v2f den = Dn.f[j] * Dny.f[j];
            Dc.f[j] = v2f{fmaxf(num.x * __builtin_amdgcn_rcpf(den.x), 1e-7f), fmaxf(num.y * __builtin_amdgcn_rcpf(den.y), 1e-7f)};
        }
    }
    // ---- y fluxes; D and R back to rows; x fluxes
    Row R;
    flux_rows<1, false>(B, Uy, Cyc, Dc, Dy, sB, sT, ext + 40, i, R);
    transpose<false>(buf, g, i, Dc);
    transpose<false>(buf, g, i, R);
    flux_rows<0, true>(B, Ux, Cxr, Dc, Dx, sL, sR, ext, i, R);
    if (valid) {
        *(v4f_g*)(ud + ax) = v4f{R.f[0].x, R.f[1].x, R.f[2].x, R.f[3].x};
        *(v4f_g*)(ud + ax + 4) = v4f{R.f[0].y, R.f[1].y, R.f[2].y, R.f[3].y};
    }
}

#pragma clang fp contract(off)

}  // namespace rows2
