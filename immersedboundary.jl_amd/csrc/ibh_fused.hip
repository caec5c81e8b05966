// libibhip: fused residual sweeps (the headline hot path).
//
// One sweep = two kernels:
//   pass A  per cell : Green-Gauss gradients of every variable along every dim + JST sensor
//                      (cell_gradient :965, JST_sensor :1077)          -> workspace G
//   pass B  per cell : for each face of the cell MUSCL(high_order) states, flux, and the
//                      Green-Gauss sum of the fluxes (MUSCL :1113, green_gauss :918)
// Each pass has two bodies launched together in ONE grid:
//   * block fast path (2-D, 8x8 blocks): one 64-lane wavefront per block, lane = cell,
//     x-fastest like the reference's cell numbering (mesher.jl:1064-1112).  Cell values are
//     staged in LDS; the halo across the four block sides (same level, mirror, 2:1 coarse,
//     2:1 fine -- classified and verified against the face lists by ibh_analyze.cpp) is
//     fetched by one gather instruction per field and staged next to the tile.  No index
//     arrays are read for these cells.
//   * face-list path: one thread per cell walking the CSR left/right face lists; used for
//     cells of partial (skirt) blocks, sides the analysis could not classify, 3-D, and when
//     IBH_FORCE_GENERAL is set.
// Both bodies call the same per-face functions (ibh_flux.h) in the same order, so they agree
// bit for bit with each other and with the oracle's array-at-a-time evaluation.
#include <algorithm>

#include <string.h>

#include "ibh_common.h"
#include "ibh_flux.h"

using namespace ibhf;

namespace {

struct PartView {
    int32_t nc;
    const float* spacing;
    DimData dim[IBH_MAXD];
    const int32_t* side;  // side table (ibh_common.h): the cell across the one face of a side, -2 none, -1 walk the lists
};

// The faces of cell c on one side of dimension d.  A side with ONE face is taken from the side table -- the cell across,
// no offsets / face ids / owner and neighbour lookups (four dependent loads become one) -- with the weight 1.0f / 1 the
// walk would use; anything else walks the CSR lists.  Same faces, same order, same arithmetic.
struct SideIter {
    int32_t b, e, o, n;
    const int32_t* idx;
    bool direct;
};
__device__ __forceinline__ SideIter side_iter(const PartView& p, int d, int side, int32_t c) {
    const DimData& dd = p.dim[d];
    SideIter it;
    it.idx = side ? dd.ridx : dd.lidx;
    const int32_t t = p.side[(int64_t)(2 * d + side) * p.nc + c];
    it.direct = t >= 0;
    it.o = side ? c : t;
    it.n = side ? t : c;
    if (t >= 0) {
        it.b = 0;
        it.e = 1;
    } else if (t == -2) {
        it.b = it.e = 0;
    } else {
        const int32_t* off = side ? dd.roff : dd.loff;
        it.b = off[c];
        it.e = off[c + 1];
    }
    return it;
}
__device__ __forceinline__ void side_face(const DimData& dd, const SideIter& it, int32_t k, int32_t& o, int32_t& n) {
    if (it.direct) {
        o = it.o;
        n = it.n;
    } else {
        const int32_t f = it.idx[k];
        o = dd.owners[f];
        n = dd.neighbors[f];
    }
}

// ------------------------------------------------------------------------------------------
// face-list bodies
// ------------------------------------------------------------------------------------------
// G layout: gradient of variable v along dim d at G[(d*NV + v)*nc + c]; sensor at G[ND*NV*nc + c].
template <int ND, int NV>
__device__ __forceinline__ void passA_cell(const PartView& p, const float* __restrict__ u, int64_t ldu,
                                           float* __restrict__ G, int32_t c) {
    const int64_t nc = p.nc;
    float D = 1e-7f;
    float uc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) uc[v] = u[c + v * ldu];
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        const DimData& dd = p.dim[d];
        const float* h = p.spacing + d * nc;
        float hc = h[c];
        float s2[2][NV];
        float ds[2] = {0.f, 0.f}, as[2] = {0.f, 0.f};
#pragma unroll
        for (int side = 1; side >= 0; --side) {   // right faces, then left faces
            float* s = s2[side];
#pragma unroll
            for (int v = 0; v < NV; ++v) s[v] = 0.f;
            const SideIter it = side_iter(p, d, side, c);
            if (it.direct) {
                // one face: own values from registers, the cell across gathered (weight 1.0f / 1)
                const int32_t x = side ? it.n : it.o;
                const float hx = h[x];
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const float ux = u[x + v * ldu];
                    const float uo = side ? uc[v] : ux, un = side ? ux : uc[v];
                    s[v] = face_avg(uo, un, side ? hc : hx, side ? hx : hc) * 1.0f;
                    if (v == 0) {
                        const float df = un - uo;
                        ds[side] = df * 1.0f;
                        as[side] = fabsf(df) * 1.0f;
                    }
                }
                continue;
            }
            const int32_t b = it.b, e = it.e;
            float w = (e > b) ? 1.0f / (float)(e - b) : 0.f;
            for (int32_t k = b; k < e; ++k) {
                int32_t o, n;
                side_face(dd, it, k, o, n);
                float ho = h[o], hn = h[n];
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    float uo = u[o + v * ldu], un = u[n + v * ldu];
                    float t = face_avg(uo, un, ho, hn) * w;
                    s[v] = (k == b) ? t : s[v] + t;
                    if (v == 0) {
                        float df = un - uo;
                        float td = df * w, ta = fabsf(df) * w;
                        ds[side] = (k == b) ? td : ds[side] + td;
                        as[side] = (k == b) ? ta : as[side] + ta;
                    }
                }
            }
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) G[(int64_t)(d * NV + v) * nc + c] = (s2[1][v] - s2[0][v]) / hc;
        float gg = (ds[1] - ds[0]) / hc;
        float ugg = (as[1] + as[0]) / hc;
        D = fmaxf(D, (1e-7f + fabsf(gg)) / (1e-7f + ugg));
    }
    G[(int64_t)(ND * NV) * nc + c] = D;
}

template <int ND>
__device__ __forceinline__ void passB_adv_cell(const PartView& p, const float* __restrict__ u,
                                               const float* __restrict__ C, int64_t ldc, const float* __restrict__ G,
                                               float* __restrict__ ud, int32_t c) {
    const int64_t nc = p.nc;
    const float* Ds = G + (int64_t)ND * nc;
    float r = 0.0f;
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        const DimData& dd = p.dim[d];
        const float* h = p.spacing + d * nc;
        const float* g = G + (int64_t)d * nc;
        const float* Cd = C + (int64_t)d * ldc;
        float fr = 0.f, fl = 0.f;
        {
            const SideIter it = side_iter(p, d, 1, c);
            const int32_t b = it.b, e = it.e;
            float w = (e > b) ? 1.0f / (float)(e - b) : 0.f;
            for (int32_t k = b; k < e; ++k) {
                int32_t o, n;
                side_face(dd, it, k, o, n);
                float t = adv_flux(u[o], u[n], g[o], g[n], Ds[o], Ds[n], Cd[o], Cd[n], h[o], h[n]) * w;
                fr = (k == b) ? t : fr + t;
            }
        }
        {
            const SideIter it = side_iter(p, d, 0, c);
            const int32_t b = it.b, e = it.e;
            float w = (e > b) ? 1.0f / (float)(e - b) : 0.f;
            for (int32_t k = b; k < e; ++k) {
                int32_t o, n;
                side_face(dd, it, k, o, n);
                float t = adv_flux(u[o], u[n], g[o], g[n], Ds[o], Ds[n], Cd[o], Cd[n], h[o], h[n]) * w;
                fl = (k == b) ? t : fl + t;
            }
        }
        r = r - (fr - fl) / h[c];
    }
    ud[c] = r;
}

template <int ND>
__device__ __forceinline__ void passB_euler_cell(const PartView& p, const float* __restrict__ P, int64_t ldp,
                                                 const float* __restrict__ G, float* __restrict__ Rr, int64_t ldr,
                                                 float Rgas, float gamma, int32_t c) {
    constexpr int NV = ND + 2;
    const int64_t nc = p.nc;
    const float* Ds = G + (int64_t)(ND * NV) * nc;
    float res[NV], Pc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        res[v] = 0.0f;
        Pc[v] = P[c + v * ldp];
    }
    const float Dc = Ds[c];
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        const DimData& dd = p.dim[d];
        const float* h = p.spacing + d * nc;
        double fr[NV], fl[NV];
        float dPc[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            fr[v] = fl[v] = 0.0;
            dPc[v] = G[(int64_t)(d * NV + v) * nc + c];
        }
        const float hcf = h[c];
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            double* acc = side ? fr : fl;
            const SideIter it = side_iter(p, d, side, c);
            if (it.direct) {
                // one face: the cell's own values are in registers, only the cell across is gathered (weight 1.0f / 1)
                const int32_t x = side ? it.n : it.o;
                float Px[NV], dPx[NV];
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    Px[v] = P[x + v * ldp];
                    dPx[v] = G[(int64_t)(d * NV + v) * nc + x];
                }
                const float Dx = Ds[x], hx = h[x];
                double F[NV];
                if (side) euler_face_flux<ND>(Pc, Px, dPc, dPx, Dc, Dx, hcf, hx, d, Rgas, gamma, F);
                else euler_face_flux<ND>(Px, Pc, dPx, dPc, Dx, Dc, hx, hcf, d, Rgas, gamma, F);
#pragma unroll
                for (int v = 0; v < NV; ++v) acc[v] = F[v] * (double)1.0f;
                continue;
            }
            const int32_t b = it.b, e = it.e;
            float w = (e > b) ? 1.0f / (float)(e - b) : 0.f;
            for (int32_t k = b; k < e; ++k) {
                int32_t o, n;
                side_face(dd, it, k, o, n);
                float Po[NV], Pn[NV], dPo[NV], dPn[NV];
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    Po[v] = P[o + v * ldp];
                    Pn[v] = P[n + v * ldp];
                    dPo[v] = G[(int64_t)(d * NV + v) * nc + o];
                    dPn[v] = G[(int64_t)(d * NV + v) * nc + n];
                }
                double F[NV];
                euler_face_flux<ND>(Po, Pn, dPo, dPn, Ds[o], Ds[n], h[o], h[n], d, Rgas, gamma, F);
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    double t = F[v] * (double)w;
                    acc[v] = (k == b) ? t : acc[v] + t;
                }
            }
        }
        double hc = (double)hcf;
#pragma unroll
        for (int v = 0; v < NV; ++v) res[v] = (float)((double)res[v] - (fr[v] - fl[v]) / hc);
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) Rr[c + v * ldr] = res[v];
}

// ------------------------------------------------------------------------------------------
// face-list bodies over flattened stencil records (same arithmetic and summation order as the CSR
// walk above, two dependent memory trips instead of four)
// ------------------------------------------------------------------------------------------
struct FlatRec {
    const int32_t* rec;
    int32_t n;
};

template <int ND, int NV>
__device__ __forceinline__ void passA_flat(const PartView& p, const FlatRec& R, int32_t t, int32_t c,
                                           const float* __restrict__ u, int64_t ldu, float* __restrict__ G) {
    const int64_t nc = p.nc;
    float D = 1e-7f;
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        const float* h = p.spacing + d * nc;
        const float hc = h[c];
        float s[2][NV], sd[2] = {0.f, 0.f}, sa[2] = {0.f, 0.f};
        float uc[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            uc[v] = u[c + v * ldu];
            s[0][v] = s[1][v] = 0.f;
        }
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const int64_t q = 2 * d + side;
            const int cnt = R.rec[(q * 5) * R.n + t];
            const float w = cnt > 0 ? 1.0f / (float)cnt : 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k < cnt) {
                    const int32_t o = R.rec[(q * 5 + 1 + k) * R.n + t];
                    const float ho = h[o];
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        const float uo = u[o + v * ldu];
                        // side 1 (right face): owner = this cell, neighbour = o; side 0: owner = o
                        const float fa = side ? face_avg(uc[v], uo, hc, ho) : face_avg(uo, uc[v], ho, hc);
                        const float tt = fa * w;
                        s[side][v] = (k == 0) ? tt : s[side][v] + tt;
                        if (v == 0) {
                            const float df = side ? (uo - uc[v]) : (uc[v] - uo);
                            const float td = df * w, ta = fabsf(df) * w;
                            sd[side] = (k == 0) ? td : sd[side] + td;
                            sa[side] = (k == 0) ? ta : sa[side] + ta;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) G[(int64_t)(d * NV + v) * nc + c] = (s[1][v] - s[0][v]) / hc;
        const float gg = (sd[1] - sd[0]) / hc;
        const float ugg = (sa[1] + sa[0]) / hc;
        D = fmaxf(D, (1e-7f + fabsf(gg)) / (1e-7f + ugg));
    }
    G[(int64_t)(ND * NV) * nc + c] = D;
}

template <int ND>
__device__ __forceinline__ void passB_adv_flat(const PartView& p, const FlatRec& R, int32_t t, int32_t c,
                                               const float* __restrict__ u, const float* __restrict__ C, int64_t ldc,
                                               const float* __restrict__ G, float* __restrict__ ud) {
    const int64_t nc = p.nc;
    const float* Ds = G + (int64_t)ND * nc;
    const float uc = u[c], Dc = Ds[c];
    float r = 0.0f;
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        const float* h = p.spacing + d * nc;
        const float* g = G + (int64_t)d * nc;
        const float* Cd = C + (int64_t)d * ldc;
        const float hc = h[c], gc = g[c], Cc = Cd[c];
        float fs[2] = {0.f, 0.f};
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const int64_t q = 2 * d + side;
            const int cnt = R.rec[(q * 5) * R.n + t];
            const float w = cnt > 0 ? 1.0f / (float)cnt : 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k < cnt) {
                    const int32_t o = R.rec[(q * 5 + 1 + k) * R.n + t];
                    const float fl = side ? adv_flux(uc, u[o], gc, g[o], Dc, Ds[o], Cc, Cd[o], hc, h[o])
                                          : adv_flux(u[o], uc, g[o], gc, Ds[o], Dc, Cd[o], Cc, h[o], hc);
                    const float tt = fl * w;
                    fs[side] = (k == 0) ? tt : fs[side] + tt;
                }
            }
        }
        r = r - (fs[1] - fs[0]) / hc;
    }
    ud[c] = r;
}

// ------------------------------------------------------------------------------------------
// block fast path, 2-D, 8x8 blocks.  LDS per wave and per field: tile[64] + halo[4][8][2].
// halo slot (s, t, k): side s, boundary cell t along the side, k-th face (k = 1 only on FINE sides)
// ------------------------------------------------------------------------------------------
#ifndef WPB
#define WPB 4  // waves (= blocks) per workgroup of 64*WPB threads
#endif

// XCD-aware workgroup remap (cdna_hip_programming.md T1): workgroups are dealt round-robin over the
// 8 XCDs, each with its own non-coherent L2.  Give every XCD one CONTIGUOUS chunk of the block list
// (blocks are in depth-first/Morton order, so a chunk is a compact patch of the mesh): halo lines of
// neighbouring blocks are then served by the same L2 instead of being fetched once per XCD.
// Bijective for any nwg; placement only affects speed.
__device__ __forceinline__ int32_t xcd_remap(int32_t wg, int32_t nwg) {
#ifdef IBH_NO_XCD_REMAP
    return wg;
#else
    const int32_t q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
#endif
}

// Fetch the neighbour across direction s (0:x- 1:x+ 2:y- 3:y+) of the field staged in `tile`/`halo`.
__device__ __forceinline__ void nb_fetch(const float* tile, const float* halo, int lane, int i, int j, int s, float self,
                                         float& v0, float& v1) {
    bool edge = (s == 0) ? (i == 0) : (s == 1) ? (i == 7) : (s == 2) ? (j == 0) : (j == 7);
    int t = (s < 2) ? j : i;
    if (!edge) {
        int off = (s == 0) ? -1 : (s == 1) ? 1 : (s == 2) ? -8 : 8;
        v0 = tile[lane + off];
        v1 = v0;
    } else {
        v0 = halo[(s * 8 + t) * 2];
        v1 = halo[(s * 8 + t) * 2 + 1];
    }
    (void)self;
}

// Stage one field: tile[lane] = own value, halo slots gathered by lanes 0..63 (slot = lane).
// MIRROR sides take the boundary cell's own value (o == n faces, ImmersedBoundary.jl:653-660).
__device__ __forceinline__ float stage_field(const float* __restrict__ f, const BlockDesc2& b, int lane, float* tile,
                                             float* halo, int32_t hc_idx, int32_t mirror_idx) {
    float self = f[b.base + lane];
    tile[lane] = self;
    float hv = 0.0f;
    if (hc_idx >= 0) hv = f[hc_idx];
    else if (mirror_idx >= 0) hv = f[mirror_idx];
    halo[lane] = hv;
    return self;
}

// lane -> halo slot (s, t, k) = lane; its cell comes from the per-block table built by ibh_analyze.cpp
// (single-face sides repeat sub-face 0 in slot k=1, MIRROR sides name the boundary cell itself).
__device__ __forceinline__ void lane_halo_role(const int32_t* __restrict__ htab, int32_t blk, int lane,
                                               int32_t& hc_idx, int32_t& mirror_idx) {
    hc_idx = htab[(size_t)blk * 64 + lane];
    mirror_idx = -1;
}

template <int NV>
__device__ __forceinline__ void passA_block2(const BlockDesc2* __restrict__ blocks, const int32_t* __restrict__ htab,
                                             int32_t blk, const float* spacing,
                                             int64_t nc, const float* __restrict__ u, int64_t ldu, float* __restrict__ G,
                                             float* lds, int lane) {
    const BlockDesc2& b = blocks[blk];
    const int i = lane & 7, j = lane >> 3;
    int32_t hc_idx, mirror_idx;
    lane_halo_role(htab, blk, lane, hc_idx, mirror_idx);
    float* tile = lds;         // [NV][64]
    float* halo = lds + NV * 64;  // [NV][64]
    float self[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) self[v] = stage_field(u + v * ldu, b, lane, tile + v * 64, halo + v * 64, hc_idx, mirror_idx);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    const int32_t c = b.base + lane;
    float D = 1e-7f;
    const bool general = (i == 0 && b.type[0] == SIDE_GENERAL) || (i == 7 && b.type[1] == SIDE_GENERAL) ||
                         (j == 0 && b.type[2] == SIDE_GENERAL) || (j == 7 && b.type[3] == SIDE_GENERAL);
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const float hc = b.h[d];
        const int sL = 2 * d, sR = 2 * d + 1;
        const bool edgeL = d == 0 ? (i == 0) : (j == 0);
        const bool edgeR = d == 0 ? (i == 7) : (j == 7);
        const int tyL = b.type[sL], tyR = b.type[sR];
        // neighbour spacing: same inside the block; 2h / h/2 across a 2:1 side (exact: powers of two
        // times h would also be exact, but take the stored value to stay literal)
        float hL = hc, hR = hc;
        bool twoL = false, twoR = false;
        if (edgeL) { hL = (tyL == SIDE_COARSE) ? hc * 2.0f : (tyL == SIDE_FINE) ? hc * 0.5f : hc; twoL = tyL == SIDE_FINE; }
        if (edgeR) { hR = (tyR == SIDE_COARSE) ? hc * 2.0f : (tyR == SIDE_FINE) ? hc * 0.5f : hc; twoR = tyR == SIDE_FINE; }
        float sr[NV], sl[NV], dr = 0.f, ar = 0.f, dl = 0.f, al = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            float l0, l1, r0, r1;
            nb_fetch(tile + v * 64, halo + v * 64, lane, i, j, sL, self[v], l0, l1);
            nb_fetch(tile + v * 64, halo + v * 64, lane, i, j, sR, self[v], r0, r1);
            // right faces: owner = this cell; left faces: neighbour = this cell
            float fr0 = face_avg(self[v], r0, hc, hR);
            float fl0 = face_avg(l0, self[v], hL, hc);
            float wr = twoR ? 0.5f : 1.0f, wl = twoL ? 0.5f : 1.0f;
            float a = fr0 * wr;
            if (twoR) a = a + face_avg(self[v], r1, hc, hR) * wr;
            float bb = fl0 * wl;
            if (twoL) bb = bb + face_avg(l1, self[v], hL, hc) * wl;
            sr[v] = a;
            sl[v] = bb;
            if (v == 0) {
                float d0 = r0 - self[v];
                dr = d0 * wr;
                ar = fabsf(d0) * wr;
                if (twoR) { float d1 = r1 - self[v]; dr = dr + d1 * wr; ar = ar + fabsf(d1) * wr; }
                float e0 = self[v] - l0;
                dl = e0 * wl;
                al = fabsf(e0) * wl;
                if (twoL) { float e1 = self[v] - l1; dl = dl + e1 * wl; al = al + fabsf(e1) * wl; }
            }
        }
        if (!general) {
#pragma unroll
            for (int v = 0; v < NV; ++v) G[(int64_t)(d * NV + v) * nc + c] = (sr[v] - sl[v]) / hc;
        }
        float gg = (dr - dl) / hc;
        float ugg = (ar + al) / hc;
        D = fmaxf(D, (1e-7f + fabsf(gg)) / (1e-7f + ugg));
    }
    if (!general) G[(int64_t)(2 * NV) * nc + c] = D;
    (void)spacing;
}

__device__ __forceinline__ void passB_adv_block2(const BlockDesc2* __restrict__ blocks,
                                                 const int32_t* __restrict__ htab, int32_t blk, int64_t nc,
                                                 const float* __restrict__ u, const float* __restrict__ C, int64_t ldc,
                                                 const float* __restrict__ G, float* __restrict__ ud, float* lds,
                                                 int lane) {
    const BlockDesc2& b = blocks[blk];
    const int i = lane & 7, j = lane >> 3;
    int32_t hc_idx, mirror_idx;
    lane_halo_role(htab, blk, lane, hc_idx, mirror_idx);
    // fields: 0:u 1:D 2:gx 3:gy 4:Cx 5:Cy   (halo of gx/Cx only meaningful on x sides, gy/Cy on y sides;
    // every slot is gathered anyway: one instruction per field)
    float* tile = lds;
    float* halo = lds + 6 * 64;
    const float* fld[6] = {u, G + 2 * nc, G, G + nc, C, C + ldc};
    float self[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) self[q] = stage_field(fld[q], b, lane, tile + q * 64, halo + q * 64, hc_idx, mirror_idx);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    const int32_t c = b.base + lane;
    float r = 0.0f;
    const bool general = (i == 0 && b.type[0] == SIDE_GENERAL) || (i == 7 && b.type[1] == SIDE_GENERAL) ||
                         (j == 0 && b.type[2] == SIDE_GENERAL) || (j == 7 && b.type[3] == SIDE_GENERAL);
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const float hc = b.h[d];
        const int sL = 2 * d, sR = 2 * d + 1;
        const bool edgeL = d == 0 ? (i == 0) : (j == 0);
        const bool edgeR = d == 0 ? (i == 7) : (j == 7);
        const int tyL = b.type[sL], tyR = b.type[sR];
        float hL = hc, hR = hc;
        bool twoL = false, twoR = false;
        if (edgeL) { hL = (tyL == SIDE_COARSE) ? hc * 2.0f : (tyL == SIDE_FINE) ? hc * 0.5f : hc; twoL = tyL == SIDE_FINE; }
        if (edgeR) { hR = (tyR == SIDE_COARSE) ? hc * 2.0f : (tyR == SIDE_FINE) ? hc * 0.5f : hc; twoR = tyR == SIDE_FINE; }
        const int qg = 2 + d, qc = 4 + d;
        float uL0, uL1, uR0, uR1, gL0, gL1, gR0, gR1, DL0, DL1, DR0, DR1, CL0, CL1, CR0, CR1;
        nb_fetch(tile, halo, lane, i, j, sL, self[0], uL0, uL1);
        nb_fetch(tile, halo, lane, i, j, sR, self[0], uR0, uR1);
        nb_fetch(tile + 64, halo + 64, lane, i, j, sL, self[1], DL0, DL1);
        nb_fetch(tile + 64, halo + 64, lane, i, j, sR, self[1], DR0, DR1);
        nb_fetch(tile + qg * 64, halo + qg * 64, lane, i, j, sL, self[qg], gL0, gL1);
        nb_fetch(tile + qg * 64, halo + qg * 64, lane, i, j, sR, self[qg], gR0, gR1);
        nb_fetch(tile + qc * 64, halo + qc * 64, lane, i, j, sL, self[qc], CL0, CL1);
        nb_fetch(tile + qc * 64, halo + qc * 64, lane, i, j, sR, self[qc], CR0, CR1);
        float wr = twoR ? 0.5f : 1.0f, wl = twoL ? 0.5f : 1.0f;
        float fr = adv_flux(self[0], uR0, self[qg], gR0, self[1], DR0, self[qc], CR0, hc, hR) * wr;
        if (twoR) fr = fr + adv_flux(self[0], uR1, self[qg], gR1, self[1], DR1, self[qc], CR1, hc, hR) * wr;
        float fl = adv_flux(uL0, self[0], gL0, self[qg], DL0, self[1], CL0, self[qc], hL, hc) * wl;
        if (twoL) fl = fl + adv_flux(uL1, self[0], gL1, self[qg], DL1, self[1], CL1, self[qc], hL, hc) * wl;
        r = r - (fr - fl) / hc;
    }
    if (!general) ud[c] = r;
}

}  // namespace

#include "ibh_block2d.h"
#include "ibh_sweep2d.h"
#include "ibh_quad2d.h"
#include "ibh_quad2d_euler.h"
#include "ibh_rows2d.h"
#include "ibh_strip3d.h"
#include "ibh_strip3d_euler.h"
#include "ibh_cols3d.h"
#include "ibh_halo_dev.h"
#include "ibh_block3d.h"

namespace {

// ------------------------------------------------------------------------------------------
// kernels: grid = [fast-path workgroups | face-list workgroups]
// EXACT = literal IEEE arithmetic in the block path (bit-comparable with the face-list path);
// otherwise the tuned block path of ibh_block2d.h.
// ------------------------------------------------------------------------------------------
template <int ND, int NV, bool EXACT>
__global__ __launch_bounds__(64 * WPB) void k_passA(PartView p, const float* __restrict__ u, int64_t ldu,
                                               float* __restrict__ G, const BlockDesc2* __restrict__ blocks,
                                               const int32_t* __restrict__ htab, int32_t nblk, int32_t nwg_fast,
                                               const int32_t* __restrict__ cells, int32_t ncells, FlatRec flat,
                                               const int32_t* __restrict__ blist) {
    // `blocks`/`htab`/`nblk` describe the sub-range of the block table this launch covers, or, with `blist`,
    // the whole table and the list of the nblk block indices to take.
    // grid = [face-list workgroups | block workgroups]: the latency-bound face-list cells go first
    __shared__ float lds[WPB * NV * 128];
    const int32_t gI = (ncells + 64 * WPB - 1) / (64 * WPB);
    if ((int32_t)blockIdx.x >= gI) {
        const int32_t wg = blockIdx.x - gI;
        if constexpr (ND == 2) {
            int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
            int32_t blk = __builtin_amdgcn_readfirstlane(xcd_remap(wg, nwg_fast) * WPB + wave);
            if (blk < nblk) {
                if (blist) blk = blist[blk];
                if constexpr (EXACT)
                    passA_block2<NV>(blocks, htab, blk, p.spacing, p.nc, u, ldu, G, lds + wave * NV * 128, lane);
                else
                    blk2::passA<NV>(blocks, htab, blk, (uint32_t)p.nc, u, (uint32_t)ldu, G, lds + wave * NV * 128, lane);
            }
        }
        return;
    }
#ifdef IBH_NO_XCD_CELLS
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
#else
    int64_t t = (int64_t)ibh_xcd_chunk((int32_t)blockIdx.x, gI) * blockDim.x + threadIdx.x;   // (the face-list workgroups)
#endif
    if (t >= ncells) return;
    int32_t c = cells ? cells[t] : (int32_t)t;
    if (flat.rec) passA_flat<ND, NV>(p, flat, (int32_t)t, c, u, ldu, G);
    else passA_cell<ND, NV>(p, u, ldu, G, c);
}

template <int ND, bool EXACT>
__global__ __launch_bounds__(64 * WPB) void k_passB_adv(PartView p, const float* __restrict__ u, const float* __restrict__ C,
                                                   int64_t ldc, const float* __restrict__ G, float* __restrict__ ud,
                                                   const BlockDesc2* __restrict__ blocks,
                                                   const int32_t* __restrict__ htab, int32_t nblk, int32_t nwg_fast,
                                                   const int32_t* __restrict__ cells, int32_t ncells, FlatRec flat,
                                                   const int32_t* __restrict__ blist) {
    constexpr int LDSW = EXACT ? 6 * 128 : BLK2_PASSB_LDS;  // floats per wave
    __shared__ float lds[WPB * LDSW];
    const int32_t gI = (ncells + 64 * WPB - 1) / (64 * WPB);
    if ((int32_t)blockIdx.x >= gI) {
        const int32_t wg = blockIdx.x - gI;
        if constexpr (ND == 2) {
            int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
            int32_t blk = __builtin_amdgcn_readfirstlane(xcd_remap(wg, nwg_fast) * WPB + wave);
            if (blk < nblk) {
                if (blist) blk = blist[blk];
                if constexpr (EXACT)
                    passB_adv_block2(blocks, htab, blk, p.nc, u, C, ldc, G, ud, lds + wave * LDSW, lane);
                else
                    blk2::passB_adv(blocks, htab, blk, (uint32_t)p.nc, u, C, (uint32_t)ldc, G, ud, lds + wave * LDSW,
                                    lane);
            }
        }
        return;
    }
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ncells) return;
    int32_t c = cells ? cells[t] : (int32_t)t;
    if (flat.rec) passB_adv_flat<ND>(p, flat, (int32_t)t, c, u, C, ldc, G, ud);
    else passB_adv_cell<ND>(p, u, C, ldc, G, ud, c);
}

// Single-kernel sweep over a range of eligible blocks (blk2::sweep_adv): no workspace traffic, one launch.
// A workgroup owns WPB*iters consecutive blocks; wave w takes block (first + k*WPB + w), k = 0..iters-1, so the
// waves of a workgroup always work on adjacent blocks and the lane-only index arithmetic is paid once per wave.
#ifndef IBH_SWEEP_WAVES
#define IBH_SWEEP_WAVES 5
#endif
template <bool DT>
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(IBH_SWEEP_WAVES, IBH_SWEEP_WAVES))) void k_sweep_adv(const float* __restrict__ u, const float* __restrict__ C,
                                                        uint32_t ldc, float* __restrict__ ud,
                                                        const BlockDesc2* __restrict__ blocks,
                                                        const int32_t* __restrict__ htab,
                                                        const int32_t* __restrict__ etab,
                                                        const int32_t* __restrict__ dtab, int32_t nblk, int32_t nwg,
                                                        int32_t iters, const int32_t* __restrict__ blist) {
    __shared__ float lds[WPB * BLK2_SWEEP_LDS];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;  // uniform LDS base
    const int32_t first = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, nwg) * (WPB * iters) + wave);
    if (first >= nblk) return;
    const int32_t nb = __builtin_amdgcn_readfirstlane(min(iters, (nblk - first + WPB - 1) / WPB));
    blk2::sweep_adv<DT>(blocks, htab, etab, dtab, blist, first, WPB, nb, u, C, ldc, ud, lds + wave * BLK2_SWEEP_LDS, lane);
}

// Quad sweep (quad2::sweep_quad): one wavefront per 2x2 group of sibling blocks; the blocks outside such groups take the
// per-block single kernel (blk2::sweep_adv) in the SAME launch: grid = [quad workgroups | single-block workgroups]
// (the other order measured 0.5 us slower).  Measured and dropped (profiles/r2_*/README.md): a persistent form (about
// two waves per SIMD splitting the item list by estimated cost, next item's loads in flight: 7.2 us against 5.5 us) and
// several quads per wave with prefetch (6.1 us) -- the sweep lives on wave-level parallelism.
#define QUAD_WG_LDS (WPB * (QUAD_LDS > BLK2_SWEEP_LDS ? QUAD_LDS : BLK2_SWEEP_LDS))
// wave timeline of a launch (STAMP, ibh_debug_buffer): per wave {start, end} in 100 MHz ticks and the HW_ID register
__device__ unsigned long long* ibh_dbg_buf = nullptr;
__device__ __forceinline__ void dbg_stamp(int32_t slot, int k, unsigned long long v) {
    if (ibh_dbg_buf && threadIdx.x % 64 == 0) ibh_dbg_buf[(size_t)slot * 8 + k] = v;
}

// RS: the blocks outside quads take the row sweep, EIGHT per wave (rows2::sweep_rows over the list), instead of the per-block
// body -- `nwgs`, `siters` then count waves of eight
template <bool DT, bool STAMP, int GM = 127, bool STEP = false, bool RS = false>
#ifndef QS_WAVES
#define QS_WAVES 5
#endif
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(QS_WAVES, QS_WAVES))) void k_sweep_quad(const float* __restrict__ u, const float* __restrict__ C,
                                                         uint32_t ldc, float* __restrict__ ud,
                                                         const QuadDesc2* __restrict__ qd,
                                                         const int32_t* __restrict__ qtab, int32_t nq, int32_t nwgq,
                                                         const BlockDesc2* __restrict__ blocks,
                                                         const int32_t* __restrict__ htab,
                                                         const int32_t* __restrict__ etab,
                                                         const int32_t* __restrict__ dtab,
                                                         const int32_t* __restrict__ singles, int32_t ns, int32_t nwgs,
                                                         int32_t singles_first, int32_t siters,
                                                         const float* __restrict__ dtp = nullptr, int32_t npair = 0,
                                                         const int32_t* __restrict__ qaux = nullptr) {
    __shared__ __attribute__((aligned(16))) float lds[RS && WPB * ROWS_LDS > QUAD_WG_LDS ? WPB * ROWS_LDS : QUAD_WG_LDS];
    float dt = 0.0f;
    if constexpr (STEP) dt = *dtp;  // (scalar load: the time step lives on the device, ibh_timestep_advection)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int32_t slot = blockIdx.x * WPB + wave;
    if constexpr (STAMP) {
        dbg_stamp(slot, 0, __builtin_amdgcn_s_memrealtime());
        dbg_stamp(slot, 2, __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)));  // HW_REG_HW_ID, all 32 bits
    }
    const int32_t wgq = singles_first ? (int32_t)blockIdx.x - nwgs : (int32_t)blockIdx.x;
    const bool isq = wgq >= 0 && wgq < nwgq;
    if (isq) {
        const int32_t q = __builtin_amdgcn_readfirstlane(xcd_remap(wgq, nwgq) * WPB + wave);
        if (q < nq)
            quad2::sweep_quad<STAMP, GM, STEP>(qd, qtab, q, u, C, ldc, ud, lds + wave * QUAD_LDS, lane,
                                               STAMP && ibh_dbg_buf ? ibh_dbg_buf + (size_t)slot * 8 : nullptr, dt, qaux);
        else if (q < nq + npair)  // pair tiles: entries nq .. of the same arrays, the HALF form of the same wave code
            quad2::sweep_quad<STAMP, GM, STEP, true>(qd, qtab, q, u, C, ldc, ud, lds + wave * QUAD_LDS, lane,
                                                     STAMP && ibh_dbg_buf ? ibh_dbg_buf + (size_t)slot * 8 : nullptr, dt, qaux);
    } else {
        const int32_t wgs = singles_first ? (int32_t)blockIdx.x : (int32_t)blockIdx.x - nwgq;
        if constexpr (RS) {
            const int32_t first8 = __builtin_amdgcn_readfirstlane((xcd_remap(wgs, nwgs) * WPB + wave) * 8);
            if (first8 < ns) {
                __builtin_amdgcn_s_setprio(3);  // a row wave is the longest-lived wave of the launch
                rows2::sweep_rows(blocks, etab, first8, ns, u, C, ldc, ud, lds + wave * ROWS_LDS, lane, singles);
            }
            return;
        }
        const int32_t first = __builtin_amdgcn_readfirstlane(xcd_remap(wgs, nwgs) * (WPB * siters) + wave);
        if (first < ns) {
            const int32_t nb = __builtin_amdgcn_readfirstlane(min(siters, (ns - first + WPB - 1) / WPB));
            blk2::sweep_adv<DT, STEP>(blocks, htab, etab, dtab, singles, first, WPB, nb, u, C, ldc, ud,
                                      lds + wave * BLK2_SWEEP_LDS, lane, dt);
        }
    }
    if constexpr (STAMP) {
        __builtin_amdgcn_s_waitcnt(0);
        dbg_stamp(slot, 1, __builtin_amdgcn_s_memrealtime());
        dbg_stamp(slot, 3, (unsigned long long)isq);
    }
}

// Row / column sweep (rows2::sweep_rows): one wavefront per EIGHT blocks, every complete block of a one-partition mesh
#ifndef WPBR
#define WPBR 4
#endif
__global__ __launch_bounds__(64 * WPBR) void k_sweep_rows(const float* __restrict__ u, const float* __restrict__ C,
                                                          uint32_t ldc, float* __restrict__ ud,
                                                          const BlockDesc2* __restrict__ blocks,
                                                          const int32_t* __restrict__ etab, int32_t b0, int32_t n,
                                                          int32_t nwg, const int32_t* __restrict__ list) {
    __shared__ __attribute__((aligned(16))) float lds[WPBR * ROWS_LDS];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int32_t first = __builtin_amdgcn_readfirstlane((xcd_remap(blockIdx.x, nwg) * WPBR + wave) * 8);
    if (first < n)
        rows2::sweep_rows(blocks + b0, etab + (size_t)b0 * 16, first, n, u, C, ldc, ud, lds + wave * ROWS_LDS, lane, list);
}

// One step of a rank of a multi-GPU run in ONE launch: the xGMI halo exchange of u (ibh_halo_dev.h) and the image-only
// quad sweep.  Grid = [E exchange workgroups | interior quads | interior single blocks | boundary quads | boundary
// single blocks]: the exchange workgroups push this rank's skirt rows to the peers, wait for the peers' rows and unpack
// them while the interior waves -- which read no skirt cell -- already compute; a boundary wave first waits (bounded
// spin) until every exchange workgroup of ITS launch has unpacked.  fstate (device, zeroed once): [0] tickets of the
// boundary workgroups (launch index = ticket / boundary workgroups per launch: the grid of an exchanger never
// changes), [1] exchange workgroups done.  A time-out sets bit 1 of state[2] (XgmiHalo.healthy()).
template <bool DT>
__global__ __launch_bounds__(64 * WPB) void k_step_quad(float* __restrict__ u, const float* __restrict__ C, uint32_t ldc,
                                                        float* __restrict__ ud, const QuadDesc2* __restrict__ qd,
                                                        const int32_t* __restrict__ qtab, int32_t nq_int, int32_t nq,
                                                        const BlockDesc2* __restrict__ blocks,
                                                        const int32_t* __restrict__ htab,
                                                        const int32_t* __restrict__ etab,
                                                        const int32_t* __restrict__ dtab,
                                                        const int32_t* __restrict__ singles, int32_t ns_int, int32_t ns,
                                                        const int32_t* __restrict__ send_all,
                                                        const int32_t* __restrict__ recv_all,
                                                        const float* __restrict__ src0, const float* __restrict__ src1,
                                                        XchgArgs A, uint32_t* __restrict__ state, uint32_t max_spins,
                                                        int32_t E, unsigned long long* __restrict__ fstate) {
    __shared__ __attribute__((aligned(16))) float lds[QUAD_WG_LDS];
    const int32_t b0 = (int32_t)blockIdx.x;
    if (b0 < E) {
        halo_exchange_wg(u, 1, 0, send_all, recv_all, src0, src1, A, state, max_spins, b0, E);
        __threadfence();  // the unpacked skirt rows before the count
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(&fstate[1], 1ull);
        return;
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int32_t nwg_qi = (nq_int + WPB - 1) / WPB, nwg_si = (ns_int + WPB - 1) / WPB;
    const int32_t nwg_qb = (nq - nq_int + WPB - 1) / WPB, nwg_sb = (ns - ns_int + WPB - 1) / WPB;
    int32_t b = b0 - E;
    const bool boundary = b >= nwg_qi + nwg_si;
    if (boundary) {
        __shared__ unsigned long long want;
        if (threadIdx.x == 0) {
            const unsigned long long t = atomicAdd(&fstate[0], 1ull);
            want = (t / (unsigned long long)(nwg_qb + nwg_sb) + 1ull) * (unsigned long long)E;
        }
        __syncthreads();
        if (lane == 0) {
            const unsigned long long w = want;
            // bounded like the exchange wait, but strictly longer (4 x the spins at half the sleep): a peer that is late
            // yet inside the exchange bound must not make these waves give up first and compute on stale skirt rows
            unsigned long long spins = 0;
            const unsigned long long bound = 4ull * (unsigned long long)max_spins;
            while (__hip_atomic_load(&fstate[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < w) {
                if (++spins >= bound) {  // every wave reaches the exit
                    atomicOr(&state[2], 2u);
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
        }
        __threadfence();  // acquire: the loads below see the unpacked rows
        b -= nwg_qi + nwg_si;
    }
    // [quads | single blocks] of this phase
    const int32_t nwq = boundary ? nwg_qb : nwg_qi, q0 = boundary ? nq_int : 0, q1 = boundary ? nq : nq_int;
    const int32_t nws = boundary ? nwg_sb : nwg_si, s0 = boundary ? ns_int : 0, s1 = boundary ? ns : ns_int;
    if (b < nwq) {
        const int32_t q = __builtin_amdgcn_readfirstlane(q0 + xcd_remap(b, nwq) * WPB + wave);
        if (q < q1) quad2::sweep_quad<false, 127>(qd, qtab, q, u, C, ldc, ud, lds + wave * QUAD_LDS, lane);
    } else {
        const int32_t first = __builtin_amdgcn_readfirstlane(s0 + xcd_remap(b - nwq, nws) * WPB + wave);
        if (first < s1)
            blk2::sweep_adv<DT>(blocks, htab, etab, dtab, singles, first, WPB, 1, u, C, ldc, ud,
                                lds + wave * BLK2_SWEEP_LDS, lane);
    }
}

// Single-kernel Euler sweep (blk2::sweep_euler); 1 / 2 / 4 waves per workgroup measured equal within 2 %
#ifndef WPBE
#define WPBE 4
#endif
__global__ __launch_bounds__(64 * WPBE) void k_sweep_euler(const float* __restrict__ P, uint32_t ldp,
                                                           float* __restrict__ R, uint32_t ldr, float Rgas, float gamma,
                                                           const BlockDesc2* __restrict__ blocks,
                                                           const int32_t* __restrict__ htab,
                                                           const int32_t* __restrict__ etab,
                                                           const int32_t* __restrict__ dtab, int32_t nblk, int32_t nwg,
                                                           int32_t iters, const int32_t* __restrict__ blist) {
    __shared__ float lds[WPBE * BLK2_SWEEP_EULER_LDS];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int32_t first = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, nwg) * (WPBE * iters) + wave);
    if (first >= nblk) return;
    const int32_t nb = __builtin_amdgcn_readfirstlane(min(iters, (nblk - first + WPBE - 1) / WPBE));
    blk2::sweep_euler(blocks, htab, etab, dtab, blist, first, WPBE, nb, P, ldp, R, ldr, blk2::Gas{Rgas, gamma},
                      lds + wave * BLK2_SWEEP_EULER_LDS, lane);
}

// Quad form of the Euler sweep (quad2::sweep_quad_euler): grid = [quad workgroups | single-block workgroups], like
// k_sweep_quad
#define QUADE_WG_LDS (WPBE * (QE_LDS > BLK2_SWEEP_EULER_LDS ? QE_LDS : BLK2_SWEEP_EULER_LDS))
// Waves per SIMD (QE_WAVES).  Round 2: 206 VGPRs = 2 waves; forcing 3 spilled 25 registers (15.4 against 12.1 us for the quads
// of the 0.87 M-cell mesh).  Round 3: HLL regrouped by state (the physical fluxes of the two sides are never held), edge
// faces first, residual accumulated direction by direction -> 188 VGPRs as the compiler schedules it freely, 136 with no
// spill when asked for 3 waves, 128 with 6 spilled for 4.  Same box, whole sweep: 2 / 3 / 4 waves 15.8 / 14.8 / 15.0 us at
// 0.87 M cells, 48.1 / 41.9 / 43.2 us at 3.47 M (profiles/r3_final/euler2d_waves.json).
#ifndef QE_WAVES
#define QE_WAVES 3
#endif
__global__ __launch_bounds__(64 * WPBE) __attribute__((amdgpu_waves_per_eu(QE_WAVES, QE_WAVES))) void k_sweep_quad_euler(const float* __restrict__ P, uint32_t ldp,
                                                                float* __restrict__ R, uint32_t ldr, float Rgas,
                                                                float gamma, const QuadDesc2* __restrict__ qd,
                                                                const int32_t* __restrict__ qtab, int32_t nq,
                                                                int32_t nwgq, const BlockDesc2* __restrict__ blocks,
                                                                const int32_t* __restrict__ htab,
                                                                const int32_t* __restrict__ etab,
                                                                const int32_t* __restrict__ dtab,
                                                                const int32_t* __restrict__ singles, int32_t ns,
                                                                int32_t nwgs, int32_t singles_first) {
    __shared__ __attribute__((aligned(16))) float lds[QUADE_WG_LDS];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int32_t wg = singles_first ? ((int32_t)blockIdx.x < nwgs ? (int32_t)blockIdx.x + nwgq : (int32_t)blockIdx.x - nwgs)
                                     : (int32_t)blockIdx.x;
    if (wg < nwgq) {
        const int32_t q = __builtin_amdgcn_readfirstlane(xcd_remap(wg, nwgq) * WPBE + wave);
        if (q < nq) quad2::sweep_quad_euler(qd, qtab, q, P, ldp, R, ldr, blk2::Gas{Rgas, gamma}, lds + wave * QE_LDS, lane);
    } else {
        const int32_t first = __builtin_amdgcn_readfirstlane(xcd_remap(wg - nwgq, nwgs) * WPBE + wave);
#ifndef IBH_QE_NO_SINGLES  // (instruction counts of the quad path alone: scripts/isa_count.py)
        if (first < ns)
            blk2::sweep_euler(blocks, htab, etab, dtab, singles, first, WPBE, 1, P, ldp, R, ldr, blk2::Gas{Rgas, gamma},
                              lds + wave * BLK2_SWEEP_EULER_LDS, lane);
#endif
    }
}

// 3-D block kernels: one 512-thread workgroup per 8x8x8 block (the face-list cells get their own launch)
// grid = [face-list workgroups over `cells` | nblk block workgroups]: the face-list cells (sides facing finer
// blocks, partial skirt blocks) are few but latency-bound (a ~20 us chain of dependent loads); dispatched FIRST
// in the same launch they run underneath the block work instead of forming a tail.
__global__ __launch_bounds__(512) void k_passA3_blk(PartView p, const float* __restrict__ u, float* __restrict__ G,
                                                    const BlockDesc3* __restrict__ blocks,
                                                    const int32_t* __restrict__ htab,
                                                    const int32_t* __restrict__ ftab, int32_t nblk,
                                                    const int32_t* __restrict__ cells, int32_t ncells, FlatRec flat) {
    __shared__ float lds[896];
    const int32_t gI = (ncells + 511) / 512;
    if ((int32_t)blockIdx.x >= gI) {
        const int32_t blk = xcd_remap(blockIdx.x - gI, nblk);
        blk3::passA(blocks, htab, ftab, blk, (uint32_t)p.nc, u, G, lds, threadIdx.x);
        return;
    }
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < ncells) {
        if (flat.rec) passA_flat<3, 1>(p, flat, (int32_t)t, cells[t], u, (int64_t)p.nc, G);
        else passA_cell<3, 1>(p, u, (int64_t)p.nc, G, cells[t]);
    }
}

__global__ __launch_bounds__(512) void k_passB3_adv_blk(PartView p, const float* __restrict__ u,
                                                        const float* __restrict__ C, int64_t ldc,
                                                        const float* __restrict__ G, float* __restrict__ ud,
                                                        const BlockDesc3* __restrict__ blocks,
                                                        const int32_t* __restrict__ htab,
                                                        const int32_t* __restrict__ ftab, int32_t nblk,
                                                        const int32_t* __restrict__ cells, int32_t ncells,
                                                        FlatRec flat, const int32_t* __restrict__ blist) {
    __shared__ float lds[BLK3_PASSB_LDS];
    const int32_t gI = (ncells + 511) / 512;
    if ((int32_t)blockIdx.x >= gI) {
        int32_t blk = xcd_remap(blockIdx.x - gI, nblk);
        if (blist) blk = blist[blk];
        blk3::passB_adv(blocks, htab, ftab, blk, (uint32_t)p.nc, u, C, (uint32_t)ldc, G, ud, lds, threadIdx.x);
        return;
    }
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < ncells) {
        if (flat.rec) passB_adv_flat<3>(p, flat, (int32_t)t, cells[t], u, C, ldc, G, ud);
        else passB_adv_cell<3>(p, u, C, ldc, G, ud, cells[t]);
    }
}

// 3-D single-kernel scalar sweep (blk3::sweep_adv): one 512-thread workgroup per block, every block of the partition
__global__ __launch_bounds__(512) void k_sweep3_adv(const float* __restrict__ u, const float* __restrict__ C, uint32_t ldc,
                                                    float* __restrict__ ud, const BlockDesc3* __restrict__ blocks,
                                                    const int32_t* __restrict__ htab, const int32_t* __restrict__ ftab,
                                                    const int32_t* __restrict__ rtab, const int32_t* __restrict__ r4tab,
                                                    int32_t n) {
    __shared__ float lds[BLK3_SWEEP_LDS];
    const int32_t blk = xcd_remap(blockIdx.x, n);
    blk3::sweep_adv(blocks, htab, ftab, rtab + (size_t)blk * 384, r4tab, blk, u, C, ldc, ud, lds, threadIdx.x);
}

// 3-D single-kernel Euler sweep (blk3::sweep_euler): one 512-thread workgroup per block, every block of the partition
__global__ __launch_bounds__(512) void k_sweep3_euler(const float* __restrict__ P, uint32_t ldp, float* __restrict__ R,
                                                      uint32_t ldr, float Rgas, float gamma,
                                                      const BlockDesc3* __restrict__ blocks,
                                                      const int32_t* __restrict__ htab, const int32_t* __restrict__ ftab,
                                                      const int32_t* __restrict__ rtab, const int32_t* __restrict__ r4tab,
                                                      int32_t n) {
    __shared__ float lds[BLK3_SWEEP_EULER_LDS];
    const int32_t blk = xcd_remap(blockIdx.x, n);
    blk3::sweep_euler(blocks, htab, ftab, rtab + (size_t)blk * 384, r4tab, blk, P, ldp, R, ldr, blk3::Gas3{Rgas, gamma}, lds,
                      threadIdx.x);
}

// Strip form of the 3-D scalar sweep (strip3::sweep_strip): one wavefront per block
#ifndef WPB3S
#define WPB3S 2
#endif
template <int WAVES>
__global__ __launch_bounds__(64 * WPB3S) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_sweep3_strip(const float* __restrict__ u, const float* __restrict__ C,
                                                             uint32_t ldc, float* __restrict__ ud,
                                                             const BlockDesc3* __restrict__ blocks,
                                                             const int32_t* __restrict__ htab,
                                                             const int32_t* __restrict__ ftab,
                                                             const int32_t* __restrict__ rtab,
                                                             const int32_t* __restrict__ r4tab, int32_t n, int32_t nwg) {
    __shared__ __attribute__((aligned(16))) float lds[WPB3S * S3_LDS];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int32_t blk = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, nwg) * WPB3S + wave);
    if (blk < n) strip3::sweep_strip(blocks, htab, ftab, rtab, r4tab, blk, u, C, ldc, ud, lds + wave * S3_LDS, lane);
}

// Column form of the 3-D scalar sweep (cols3::sweep_cols): one wavefront per block
#ifndef WPB3C
#define WPB3C 2
#endif
// TAB: the block table is that of the IMAGE blocks of a partition with skirt fragments (ibh_analyze3_image.cpp): sides
// towards a fragment take halo and deeper cells from tables (dtab)
template <int WAVES, bool TAB = false>
__global__ __launch_bounds__(64 * WPB3C) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_sweep3_cols(
    const float* __restrict__ u, const float* __restrict__ C, uint32_t ldc, float* __restrict__ ud,
    const BlockDesc3* __restrict__ blocks, const int32_t* __restrict__ htab, const int32_t* __restrict__ ftab,
    const int32_t* __restrict__ rtab, const int32_t* __restrict__ r4tab, int32_t n, int32_t nwg,
    const int32_t* __restrict__ dtab = nullptr) {
    __shared__ __attribute__((aligned(16))) float lds[WPB3C * C3_LDS];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int32_t blk = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, nwg) * WPB3C + wave);
    if (blk < n)
        cols3::sweep_cols(blocks, htab, ftab, rtab, r4tab, blk, u, C, ldc, ud, lds + wave * C3_LDS, lane,
                          TAB ? dtab : nullptr);
}

// Column form of the 3-D Euler sweep (strip3e::sweep_block): one wavefront per 8^3 block.
// Round 4: PERSISTENT waves -- the grid is the 2 048 wave slots of the chip at two waves per SIMD (8 per CU), and each wave
// works through a chain of blocks with the first loads of its next block in flight during the z fluxes of the one in hand
// (strip3e::sweep_euler_chain).  Every XCD gets one contiguous chunk of the block list (depth-first order: a compact patch
// of the mesh), and the waves of an XCD walk their chunk side by side (wave i: blocks c0 + i, c0 + i + W, ...), so the
// blocks in flight on an XCD at any time are W consecutive ones.
// PERSIST = false: one block per wave (grid = blocks), the A/B reference ("quad_variant" 513).
// (Round 3 measured a persistent form slower, 143 against 107 us at 4.56 M cells: at 256 VGPRs the registers of the
// prefetch spilled.  Round 4 first freed the registers -- nothing of a later pass is held through a flux loop, lane-only
// integers and the block descriptor are derived / read again per pass: 190 VGPRs -- and tried them as a third wave per
// SIMD: 168 VGPRs with 17 spilled words, 11 waves per CU resident (wave timeline), and slower: 755 against 695 us at 33.6 M
// cells on the same box.  As prefetch registers they pay: see profiles/r4_*/README.md.)
#ifndef WPB3E
#define WPB3E 1
#endif
#define S3E_SLOTS_PER_CU 8
template <int WAVES, bool STAMP = false, bool TAB = false, bool PERSIST = true>
__global__ __launch_bounds__(64 * WPB3E) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_sweep3_euler_cols(
    const float* __restrict__ P, uint32_t ldp, float* __restrict__ R, uint32_t ldr, float Rgas, float gamma,
    const BlockDesc3* __restrict__ blocks, const int32_t* __restrict__ htab, const int32_t* __restrict__ ftab,
    const int32_t* __restrict__ rtab, const int32_t* __restrict__ r4tab, int32_t n, int32_t nwg,
    const int32_t* __restrict__ dtab = nullptr) {
    __shared__ __attribute__((aligned(16))) float lds[WPB3E * S3E_LDS];
    __shared__ __attribute__((aligned(16))) float nextbuf[PERSIST ? WPB3E * S3E_NEXT : 1];  // (the LDS-DMA rows)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int32_t first, stride, end;
    if constexpr (PERSIST) {
        // nwg = workgroups of the launch, a multiple of 8; workgroup w belongs to XCD w & 7 (placement affects speed only)
        const int32_t xcd = blockIdx.x & 7, idx = (blockIdx.x >> 3) * WPB3E + wave;
        const int32_t q = n >> 3, r = n & 7;
        const int32_t c0 = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        first = __builtin_amdgcn_readfirstlane(c0 + idx);
        stride = (nwg >> 3) * WPB3E;
        end = c0 + q + (xcd < r ? 1 : 0);
    } else {
        first = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, nwg) * WPB3E + wave);
        stride = n;
        end = n;
    }
    strip3e::sweep_euler_chain<STAMP, PERSIST>(blocks, htab, ftab, rtab, r4tab, first, stride, end, P, ldp, R, ldr,
                                      blk3::Gas3{Rgas, gamma}, lds + wave * S3E_LDS, nextbuf + (PERSIST ? wave * S3E_NEXT : 0), lane,
                                      STAMP ? ibh_dbg_buf : nullptr,
                                      TAB ? dtab : nullptr);
}
// workgroups of the persistent launch: the chip's wave slots (or fewer, for few blocks), a multiple of 8
static int32_t s3e_persistent_wgs(int32_t nblk) {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    const int32_t slots = cus * S3E_SLOTS_PER_CU / WPB3E;
    const int32_t want = ((nblk + WPB3E - 1) / WPB3E + 7) & ~7;
    return want < slots ? want : slots;
}

// wave-per-block form of the 3-D scalar pass A (blk3::passA_wave): 4 blocks per 256-thread workgroup
__global__ __launch_bounds__(256) void k_passA3_wave(PartView p, const float* __restrict__ u, float* __restrict__ G,
                                                     const BlockDesc3* __restrict__ blocks,
                                                     const int32_t* __restrict__ htab,
                                                     const int32_t* __restrict__ ftab, int32_t nblk, int32_t nwg,
                                                     const int32_t* __restrict__ cells, int32_t ncells, FlatRec flat,
                                                     const int32_t* __restrict__ blist) {
    __shared__ float lds[4 * BLK3W_PASSA_LDS];
    const int32_t gI = (ncells + 255) / 256;
    if ((int32_t)blockIdx.x >= gI) {
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
        const int32_t pos = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x - gI, nwg) * 4 + wave);
        if (pos < nblk) {
            const int32_t blk = blist ? __builtin_amdgcn_readfirstlane(blist[pos]) : pos;
            blk3::passA_wave(blocks, htab, ftab, blk, (uint32_t)p.nc, u, G, lds + wave * BLK3W_PASSA_LDS, lane);
        }
        return;
    }
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < ncells) {
        if (flat.rec) passA_flat<3, 1>(p, flat, (int32_t)t, cells[t], u, (int64_t)p.nc, G);
        else passA_cell<3, 1>(p, u, (int64_t)p.nc, G, cells[t]);
    }
}

// wave-per-block form of the 3-D Euler pass A (blk3::passA_wave_nv<5>)
__global__ __launch_bounds__(256) void k_passA3e_wave(PartView p, const float* __restrict__ P, int64_t ldp,
                                                      float* __restrict__ G, const BlockDesc3* __restrict__ blocks,
                                                      const int32_t* __restrict__ htab,
                                                      const int32_t* __restrict__ ftab, int32_t nblk, int32_t nwg,
                                                      const int32_t* __restrict__ cells, int32_t ncells, FlatRec flat) {
    __shared__ float lds[4 * BLK3W_PASSA_LDS];
    const int32_t gI = (ncells + 255) / 256;
    if ((int32_t)blockIdx.x >= gI) {
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
        const int32_t blk = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x - gI, nwg) * 4 + wave);
        if (blk < nblk)
            blk3::passA_wave_nv<5>(blocks, htab, ftab, blk, (uint32_t)p.nc, P, (uint32_t)ldp, G,
                                   lds + wave * BLK3W_PASSA_LDS, lane);
        return;
    }
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < ncells) {
        if (flat.rec) passA_flat<3, 5>(p, flat, (int32_t)t, cells[t], P, ldp, G);
        else passA_cell<3, 5>(p, P, ldp, G, cells[t]);
    }
}

// 3-D Euler block kernels (5 primitives): same launch layout as the scalar 3-D kernels
__global__ __launch_bounds__(512) void k_passA3e_blk(PartView p, const float* __restrict__ P, int64_t ldp,
                                                     float* __restrict__ G, const BlockDesc3* __restrict__ blocks,
                                                     const int32_t* __restrict__ htab,
                                                     const int32_t* __restrict__ ftab, int32_t nblk,
                                                     const int32_t* __restrict__ cells, int32_t ncells, FlatRec flat) {
    __shared__ float lds[5 * 896];
    const int32_t gI = (ncells + 511) / 512;
    if ((int32_t)blockIdx.x >= gI) {
        const int32_t blk = xcd_remap(blockIdx.x - gI, nblk);
        blk3::passA_nv<5>(blocks, htab, ftab, blk, (uint32_t)p.nc, P, (uint32_t)ldp, G, lds, threadIdx.x);
        return;
    }
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < ncells) {
        if (flat.rec) passA_flat<3, 5>(p, flat, (int32_t)t, cells[t], P, ldp, G);
        else passA_cell<3, 5>(p, P, ldp, G, cells[t]);
    }
}

__global__ __launch_bounds__(512) void k_passB3e_blk(uint32_t nc, const float* __restrict__ P, uint32_t ldp,
                                                     const float* __restrict__ G, float* __restrict__ R, uint32_t ldr,
                                                     float Rgas, float gamma, const BlockDesc3* __restrict__ blocks,
                                                     const int32_t* __restrict__ htab,
                                                     const int32_t* __restrict__ ftab, int32_t nblk) {
    __shared__ float lds[BLK3_EULER_LDS];
    const int32_t blk = xcd_remap(blockIdx.x, nblk);
    blk3::passB_euler(blocks, htab, ftab, blk, nc, P, ldp, G, R, ldr, blk3::Gas3{Rgas, gamma}, lds, threadIdx.x);
}

// Euler pass B: the block body and the face-list body are separate kernels (the Float64 flux combine of
// the literal face-list body needs ~120 VGPRs and would halve the occupancy of the block body).
__global__ __launch_bounds__(64 * WPB) void k_passB_euler_blk(uint32_t nc, const float* __restrict__ P, uint32_t ldp,
                                                         const float* __restrict__ G, float* __restrict__ R,
                                                         uint32_t ldr, float Rgas, float gamma,
                                                         const BlockDesc2* __restrict__ blocks,
                                                         const int32_t* __restrict__ htab, int32_t nblk, int32_t nwg) {
    __shared__ float lds[WPB * BLK2_EULER_LDS];
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int32_t blk = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, nwg) * WPB + wave);
    if (blk < nblk)
        blk2::passB_euler(blocks, htab, blk, nc, P, ldp, G, R, ldr, blk2::Gas{Rgas, gamma}, lds + wave * BLK2_EULER_LDS,
                          lane);
}

// (127 VGPRs = four waves per SIMD.  Six and eight waves per SIMD by capping the registers -- what paid in the viscous sum --
// were measured here: 216 and 293 against 171 us on 1.67 M cells; this pass is arithmetic in Float64, the spills cost more.)
template <int ND>
__global__ __launch_bounds__(64 * WPB) void k_passB_euler(PartView p, const float* __restrict__ P, int64_t ldp,
                                                     const float* __restrict__ G, float* __restrict__ R, int64_t ldr,
                                                     float Rgas, float gamma, const int32_t* __restrict__ cells,
                                                     int32_t ncells) {
    int64_t t = IBH_WG_X() * blockDim.x + threadIdx.x;
    if (t >= ncells) return;
    int32_t c = cells ? cells[t] : (int32_t)t;
    passB_euler_cell<ND>(p, P, ldp, G, R, ldr, Rgas, gamma, c);
}

// ---- closures of a turbulence model on an all-block 3-D partition, gradients consumed where they are made (one wavefront
// per 8^3 block, blk3::wave_gradients: the arithmetic of the tuple cell_gradient's block sweep; the pointwise formulas are
// those of ibh_turb.hip, evaluated without contraction):
//   k_shear_of_velocity3: S = shear_rate(cell_gradient(u), cell_gradient(v), cell_gradient(w))      (turbulence.jl:110-124)
//   k_wray_agarwal_of3:   (nut, nuR, S) = Wray_Agarwal(R, S, cell_gradient(R), cell_gradient(S))    (turbulence.jl:222-241)
// 12 B in + 4 B out per cell instead of 3 x (4 in + 16 out) + 36 in + 4 out; 8 in + 12 out instead of 2 x 20 + 44.
template <int NV>
struct FieldPtrs {
    const float* f[NV];
};
__global__ __launch_bounds__(256) void k_shear_of_velocity3(const BlockDesc3* __restrict__ blocks,
                                                            const int32_t* __restrict__ htab,
                                                            const int32_t* __restrict__ ftab, int32_t nblk, int32_t nwg,
                                                            FieldPtrs<3> V, float* __restrict__ S,
                                                            float* __restrict__ Gout, uint32_t ldg) {
    // Gout (or null): the nine gradients on the way, d u_i / d x_j in column 3 j + i (the tuple cell_gradient's layout) --
    // a Navier-Stokes closure needs them again for its viscous fluxes (three pass-A sweeps otherwise)
    __shared__ float lds[4 * BLK3W_PASSA_LDS];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int32_t blk = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, nwg) * 4 + wave);
    if (blk >= nblk) return;
    const BlockDesc3 bb = blocks[blk];
    float g[3][8][3];
    blk3::wave_gradients<3>(bb, htab, ftab, blk, V.f, lds + wave * BLK3W_PASSA_LDS, lane, g);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float t = (g[i][k][j] + g[j][k][i]) / 2.0f;
                s = s + t * t;
            }
        S[(uint32_t)bb.base + lane + 64 * k] = sqrtf(2.0f * s);
        if (Gout) {  // (uniform)
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    __builtin_nontemporal_store(g[i][k][j], Gout + (size_t)(3 * j + i) * ldg + (uint32_t)bb.base + lane + 64 * k);
        }
    }
}
__global__ __launch_bounds__(256) void k_wray_agarwal_of3(const BlockDesc3* __restrict__ blocks,
                                                          const int32_t* __restrict__ htab,
                                                          const int32_t* __restrict__ ftab, int32_t nblk, int32_t nwg,
                                                          FieldPtrs<2> RS, float sigmaR, float C1, float kappa,
                                                          float* __restrict__ nut, float* __restrict__ nuR,
                                                          float* __restrict__ Sout) {
    __shared__ float lds[4 * BLK3W_PASSA_LDS];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int32_t blk = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, nwg) * 4 + wave);
    if (blk >= nblk) return;
    const BlockDesc3 bb = blocks[blk];
    float g[2][8][3];
    blk3::wave_gradients<2>(bb, htab, ftab, blk, RS.f, lds + wave * BLK3W_PASSA_LDS, lane, g);
    const float C2 = sigmaR + C1 / (kappa * kappa);
    constexpr float EPS32 = 1.1920929e-07f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t c = (uint32_t)bb.base + lane + 64 * k;
        float dot = g[0][k][0] * g[1][k][0];
        dot = dot + g[0][k][1] * g[1][k][1];
        dot = dot + g[0][k][2] * g[1][k][2];
        const float r = RS.f[0][c], s = RS.f[1][c];
        const float src = C1 * r * s + C2 * dot * (r / (s + EPS32));
        nut[c] = r;
        nuR[c] = r * sigmaR;
        Sout[c] = fminf(src, 10.0f * r);
    }
}

// ---- transport of a scalar with variable diffusivity on an all-block 3-D partition (ibh_scalar_transport, ibh_turb.hip, is
// the face-list form): out = S + sum_d green_gauss(at_faces(nu + nuR, d) .* face_gradient(R, d) .- at_faces(u_d .* R, d), d).
// One wavefront per 8^3 block, lane = (i, j) with its z-column of R, nu + nuR and u_d R in registers; x / y neighbours from
// four LDS tiles; lane t also owns slot t of the six sides: it gathers the cell(s) across, evaluates the side's face flux(es)
// -- one, or the mean of four behind a FINE side -- and leaves it for the boundary cell.  The expressions and their order
// are those of the face-list kernel (no contraction): equal bit for bit wherever a side has one face.
#define TR3_LDS (4 * 512 + 384)
__device__ __forceinline__ float tr3_avg(float uo, float un, float ho, float hn) { return (uo * hn + un * ho) / (hn + ho); }
__device__ __forceinline__ float tr3_flux(float Ro, float Rn, float To, float Tn, float Ao, float An, float ho, float hn) {
    const float conv = tr3_avg(Ao, An, ho, hn);   // at_faces(u_d .* R)
    const float nuf = tr3_avg(To, Tn, ho, hn);    // at_faces(nu .+ nuR)
    const float fd = (ho + hn) / 2.0f;            // face_distance
    const float fg = (Rn - Ro) / fd;              // face_gradient(R)
    return nuf * fg - conv;
}
__global__ __launch_bounds__(256) void k_scalar_transport_blocks3(const BlockDesc3* __restrict__ blocks,
                                                                  const int32_t* __restrict__ htab,
                                                                  const int32_t* __restrict__ ftab, int32_t nblk,
                                                                  int32_t nwg, uint32_t nc, const float* __restrict__ hsp,
                                                                  const float* __restrict__ R, const float* __restrict__ nuR,
                                                                  float nu, const float* __restrict__ vel, uint32_t ldv,
                                                                  const float* __restrict__ S, float* __restrict__ out) {
    using blk2::ldg;
    __shared__ float lds_all[4 * TR3_LDS];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int32_t blk = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, nwg) * 4 + wave);
    if (blk >= nblk) return;
    float* lds = lds_all + wave * TR3_LDS;
    float *tR = lds, *tT = lds + 512, *tAx = lds + 1024, *tAy = lds + 1536, *Hf = lds + 2048;
    const BlockDesc3 bb = blocks[blk];
    const uint32_t base = (uint32_t)bb.base;
    const float h[3] = {hsp[base], hsp[nc + base], hsp[2 * (size_t)nc + base]};   // the block's spacing as the cells hold it
    float Rk[8], Tk[8], Ak[3][8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t c = base + lane + 64 * k;
        Rk[k] = ldg(R, c);
        Tk[k] = nu + ldg(nuR, c);
#pragma unroll
        for (int d = 0; d < 3; ++d) Ak[d][k] = ldg(vel + (size_t)d * ldv, c) * Rk[k];
    }
    uint32_t hid[6];
    hid[0] = blk3::halo_cell3s<0>(bb, htab, blk, lane);
    hid[1] = blk3::halo_cell3s<1>(bb, htab, blk, lane);
    hid[2] = blk3::halo_cell3s<2>(bb, htab, blk, lane);
    hid[3] = blk3::halo_cell3s<3>(bb, htab, blk, lane);
    hid[4] = blk3::halo_cell3s<4>(bb, htab, blk, lane);
    hid[5] = blk3::halo_cell3s<5>(bb, htab, blk, lane);
    float hR[6], hT[6], hA[6], hh[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int d = s >> 1;
        hR[s] = ldg(R, hid[s]);
        hT[s] = nu + ldg(nuR, hid[s]);
        hA[s] = ldg(vel + (size_t)d * ldv, hid[s]) * hR[s];
        hh[s] = ldg(hsp + (size_t)d * nc, hid[s]);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        tR[k * 64 + lane] = Rk[k];
        tT[k * 64 + lane] = Tk[k];
        tAx[k * 64 + lane] = Ak[0][k];
        tAy[k * 64 + lane] = Ak[1][k];
    }
    blk2::wave_lds_sync();
    // side fluxes: slot t = lane of side s belongs to boundary cell pos(s, t) of the tile
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int d = s >> 1;
        const bool low = (s & 1) == 0;
        const int sd = d == 0 ? 1 : d == 1 ? 8 : 64, sa = d == 0 ? 8 : 1, sb = d == 2 ? 8 : 64;
        const int pos = (low ? 0 : 7) * sd + (lane & 7) * sa + (lane >> 3) * sb;
        const float Rb = tR[pos], Tb = tT[pos];
        const float Ab = d == 0 ? tAx[pos] : d == 1 ? tAy[pos] : (low ? Ak[2][0] : Ak[2][7]);
        const float hb = h[d];
        // the halo cell is the owner on a low side, the neighbour on a high side
        float F = low ? tr3_flux(hR[s], Rb, hT[s], Tb, hA[s], Ab, hh[s], hb) : tr3_flux(Rb, hR[s], Tb, hT[s], Ab, hA[s], hb, hh[s]);
        if (bb.type[s] == SIDE_FINE) {  // wave-uniform: three more faces behind this slot, mean of the four fluxes
            const int32_t* ft = ftab + (((size_t)bb.fine * 6 + s) * 64 + lane) * 3;
            F = F * 0.25f;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const uint32_t x = (uint32_t)ft[q];
                const float Rx = ldg(R, x), Tx = nu + ldg(nuR, x), Ax = ldg(vel + (size_t)d * ldv, x) * Rx;
                const float hx = ldg(hsp + (size_t)d * nc, x);
                const float Fq = low ? tr3_flux(Rx, Rb, Tx, Tb, Ax, Ab, hx, hb) : tr3_flux(Rb, Rx, Tb, Tx, Ab, Ax, hb, hx);
                F = F + Fq * 0.25f;
            }
        } else {
            F = F * 1.0f;
        }
        Hf[s * 64 + lane] = F;
    }
    blk2::wave_lds_sync();
    const int i = lane & 7, j = lane >> 3;
    const bool e0 = i == 0, e1 = i == 7, e2 = j == 0, e3 = j == 7;
    const float fzl = Hf[4 * 64 + lane], fzh = Hf[5 * 64 + lane];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t c = base + lane + 64 * k;
        const float *r = tR + k * 64, *t = tT + k * 64, *ax = tAx + k * 64, *ay = tAy + k * 64;
        const float Rc = Rk[k], Tc = Tk[k];
        float fl[3], fr[3];
        // x and y: inside the plane or the side's flux (x sides: slot j + 8 k, y sides: slot i + 8 k)
        const int xl = e0 ? lane : lane - 1, xr = e1 ? lane : lane + 1, yl = e2 ? lane : lane - 8, yr = e3 ? lane : lane + 8;
        const float fxl = tr3_flux(r[xl], Rc, t[xl], Tc, ax[xl], Ak[0][k], h[0], h[0]) * 1.0f;
        const float fxr = tr3_flux(Rc, r[xr], Tc, t[xr], Ak[0][k], ax[xr], h[0], h[0]) * 1.0f;
        const float fyl = tr3_flux(r[yl], Rc, t[yl], Tc, ay[yl], Ak[1][k], h[1], h[1]) * 1.0f;
        const float fyr = tr3_flux(Rc, r[yr], Tc, t[yr], Ak[1][k], ay[yr], h[1], h[1]) * 1.0f;
        fl[0] = e0 ? Hf[0 * 64 + j + 8 * k] : fxl;
        fr[0] = e1 ? Hf[1 * 64 + j + 8 * k] : fxr;
        fl[1] = e2 ? Hf[2 * 64 + i + 8 * k] : fyl;
        fr[1] = e3 ? Hf[3 * 64 + i + 8 * k] : fyr;
        // z: registers
        const int kl = k > 0 ? k - 1 : 0, kh = k < 7 ? k + 1 : 7;
        const float fzl_in = tr3_flux(Rk[kl], Rc, Tk[kl], Tc, Ak[2][kl], Ak[2][k], h[2], h[2]) * 1.0f;
        const float fzr_in = tr3_flux(Rc, Rk[kh], Tc, Tk[kh], Ak[2][k], Ak[2][kh], h[2], h[2]) * 1.0f;
        fl[2] = k == 0 ? fzl : fzl_in;
        fr[2] = k == 7 ? fzh : fzr_in;
        float rt = ldg(S, c);
#pragma unroll
        for (int d = 0; d < 3; ++d) rt = rt + (fr[d] - fl[d]) / h[d];
        out[c] = rt;
    }
}

// 1: wave-per-block form of the 3-D scalar pass A (IBH_3D_WAVE=0 for the 512-thread form, A/B)
const int ibh_3d_wave = getenv("IBH_3D_WAVE") ? atoi(getenv("IBH_3D_WAVE")) : 1;
// blocks per wave of the single-kernel sweep; 0 = automatic (IBH_SWEEP_ITERS overrides, for tuning)
const int ibh_sweep_iters = getenv("IBH_SWEEP_ITERS") ? atoi(getenv("IBH_SWEEP_ITERS")) : 0;
// IBH_QUAD=0: per-block single kernel everywhere (A/B runs)
const int ibh_quad = getenv("IBH_QUAD") ? atoi(getenv("IBH_QUAD")) : 1;
// row / column sweep (ibh_rows2d.h) where the partition qualifies: OFF by default -- measured slower than the quad sweep
// (8.4 against 6.1 us at 0.87 M cells, 21.7 against 17.0 at 3.47 M: profiles/r3_final/probe_rows.json)
int ibh_rows = getenv("IBH_ROWS") ? atoi(getenv("IBH_ROWS")) : 0;
// ibh_set_tuning(key, v): "quad_variant" 4 = wave time stamps (scripts/wave_timeline.py); "quad_parts" 1 / 2 = only the
// quads / only the single blocks of a quad sweep (measurement); "quad_singles_first" = grid order
int ibh_quad_variant = getenv("IBH_QUAD_VARIANT") ? atoi(getenv("IBH_QUAD_VARIANT")) : 0;  // (profiling a variant under bench.py)
int ibh_quad_parts = 3, ibh_quad_singles_iters = 1;
int ibh_quad_singles_first = getenv("IBH_SINGLES_FIRST") ? atoi(getenv("IBH_SINGLES_FIRST")) : 0;
int ibh_pairs = getenv("IBH_PAIRS") ? atoi(getenv("IBH_PAIRS")) : 1;  // pair tiles for the blocks outside quads ("pairs")
int ibh_arith_ids = getenv("IBH_ARITH_IDS") ? atoi(getenv("IBH_ARITH_IDS")) : 1;  // quad sweep: halo ids from the companion rows
int ibh_transport_blocks = 1;  // tuning key "transport_blocks" 0: the face-list transport kernel everywhere (A/B, tests)
int ibh_rows_singles = getenv("IBH_ROWS_SINGLES") ? atoi(getenv("IBH_ROWS_SINGLES")) : -1;
// measured (profiles/r3_final/rows_for_singles.json): 1 441 single blocks 5.96 -> 10.6 us, 5 937: 15.5 -> 19.0 us (a row wave
// lives ~3 us whatever the load, and the second launch is serial), 47 272: 126.1 -> 119.0 us
#define IBH_ROWS_SINGLES_MIN 24000

PartView view(const ibh_part* p) {
    PartView v;
    v.nc = p->nc;
    v.spacing = p->spacing;
    for (int d = 0; d < IBH_MAXD; ++d) v.dim[d] = p->dim[d];
    v.side = p->side;
    return v;
}

// flattened records apply only when a launch walks exactly the partition's face-list cell list
FlatRec flat_of(const ibh_part* p, const int32_t* cells) {
    FlatRec r{nullptr, 0};
    if (cells && cells == p->irr_cells && p->irr_rec) {
        r.rec = p->irr_rec;
        r.n = p->n_irr;
    }
    return r;
}

// Gradient workspace of the two-kernel forms: allocated ONCE, on the first sweep that needs it, for the largest sweep
// of the partition ((nd (nd + 2) + 1) nc floats: the Euler sweep), and kept until ibh_partition_destroy -- a HIP
// graph captured earlier keeps the pointer, so it must never be freed or moved by a later, larger request.  The
// single-kernel / quad / image-only paths never touch it and do not allocate it.
int ensure_G(ibh_part* p) {
    if (p->G) return 0;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (ibh_stream && hipStreamIsCapturing(ibh_stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone)
        return ibh_fail(-1, "the gradient workspace is allocated on the first two-kernel sweep of a partition: run one "
                            "sweep before capturing it into a HIP graph", __FILE__, __LINE__);
    const size_t bytes = (size_t)(p->nd * (p->nd + 2) + 1) * (size_t)p->nc * sizeof(float);
    IBH_HIP(hipMalloc((void**)&p->G, bytes));
    p->G_bytes = bytes;
    return 0;
}

}  // namespace

extern "C" {

int ibh_debug_buffer(void* buf) {  // device buffer of 8 x uint64 per wave of the launch, or NULL
    unsigned long long* b = (unsigned long long*)buf;
    IBH_HIP(hipMemcpyToSymbol(HIP_SYMBOL(ibh_dbg_buf), &b, sizeof(b)));
    return 0;
}

extern int ibh_viscous_per_cell;   // ibh_cfd.hip
extern int ibh_ew_scalar_only;     // ibh_ew.hip
int ibh_set_tuning(const char* key, int value) {
    IBH_REQUIRE(key, "ibh_set_tuning: null key");
    if (!strcmp(key, "viscous_per_cell")) {
        ibh_viscous_per_cell = value;
        return 0;
    }
    if (!strcmp(key, "ew_scalar")) {
        ibh_ew_scalar_only = value;
        return 0;
    }
    if (!strcmp(key, "quad_variant")) ibh_quad_variant = value;
    else if (!strcmp(key, "quad_parts")) ibh_quad_parts = value;
    else if (!strcmp(key, "quad_singles_first")) ibh_quad_singles_first = value;
    else if (!strcmp(key, "quad_singles_iters")) ibh_quad_singles_iters = value;
    else if (!strcmp(key, "rows")) ibh_rows = value;
    else if (!strcmp(key, "rows_singles")) ibh_rows_singles = value;
    else if (!strcmp(key, "transport_blocks")) ibh_transport_blocks = value;
    else if (!strcmp(key, "pairs")) ibh_pairs = value;
    else if (!strcmp(key, "arith_ids")) ibh_arith_ids = value;
    else return ibh_fail(-1, "ibh_set_tuning: unknown key", __FILE__, __LINE__);
    return 0;
}

int ibh_residual_advection(ibh_part* p, const float* u, const float* C, int64_t ldc, float* ud, int flags);
// n sweeps launched back to back from ONE call: the step loop of a compiled host (a Julia `for` around the ccall costs tens
// of nanoseconds per iteration; from Python the interpreter and ctypes would be ten times the 6 us sweep)
int ibh_residual_advection_n(ibh_part* p, const float* u, const float* C, int64_t ldc, float* ud, int flags, int n) {
    for (int i = 0; i < n; ++i) {
        const int rc = ibh_residual_advection(p, u, C, ldc, ud, flags);
        if (rc) return rc;
    }
    return 0;
}

int ibh_residual_advection(ibh_part* p, const float* u, const float* C, int64_t ldc, float* ud, int flags) {
    IBH_REQUIRE(p && u && C && ud, "ibh_residual_advection: null argument");
    if (p->nc == 0) return 0;
    int rc = 0;
    if (p->nd == 3 && p->bs == 8 && p->blocks3 && p->nblk > 0 && !(flags & (IBH_FORCE_GENERAL | IBH_EXACT))) {
        const bool ph1 = (flags & IBH_PHASE_INTERIOR) != 0, ph2 = (flags & IBH_PHASE_BOUNDARY) != 0;
        IBH_REQUIRE(!(ph1 && ph2), "IBH_PHASE_INTERIOR and IBH_PHASE_BOUNDARY are exclusive");
        if ((flags & IBH_IMAGE_ONLY) && p->img_all3 && !ph1 && !ph2 &&
            !(flags & (IBH_NO_FUSE | IBH_PASS_A_ONLY | IBH_PASS_B_ONLY))) {
            // only the image cells are wanted (a rank of a multi-GPU run) and every image block qualifies: one launch over
            // the image blocks, no workspace, nothing for the skirt fragments
            const int32_t nwg = (p->n_img3 + WPB3C - 1) / WPB3C;
            hipLaunchKernelGGL((k_sweep3_cols<3, true>), dim3(nwg), dim3(64 * WPB3C), 0, ibh_stream, u, C, (uint32_t)ldc, ud,
                               p->iblocks3, p->ihtab3, p->iftab3, p->irtab3, p->ir4tab3, p->n_img3, nwg, p->idtab3);
            IBH_LAUNCH_CHECK();
            return 0;
        }
        if (p->sweep3 && !ph1 && !ph2 &&
            !(flags & (IBH_NO_FUSE | IBH_PASS_A_ONLY | IBH_PASS_B_ONLY | IBH_IMAGE_ONLY))) {
            // every block qualifies for the single-kernel sweep: one launch, nothing through the workspace
            if (ibh_quad_variant != 512) {
                const int32_t nwg = (p->nblk + WPB3S - 1) / WPB3S;
#define S3_LAUNCH(W)                                                                                                  \
    hipLaunchKernelGGL(k_sweep3_strip<W>, dim3(nwg), dim3(64 * WPB3S), 0, ibh_stream, u, C, (uint32_t)ldc, ud, p->blocks3, \
                       p->htab3, p->ftab3, p->rtab3, p->r4tab3, p->nblk, nwg)
#define C3_LAUNCH(W)                                                                                                  \
    hipLaunchKernelGGL(k_sweep3_cols<W>, dim3(nwg), dim3(64 * WPB3C), 0, ibh_stream, u, C, (uint32_t)ldc, ud, p->blocks3, \
                       p->htab3, p->ftab3, p->rtab3, p->r4tab3, p->nblk, nwg)
                static_assert(WPB3C == WPB3S, "one grid for both forms");
                if (ibh_quad_variant == 515) S3_LAUNCH(2);
                else if (ibh_quad_variant == 514) S3_LAUNCH(4);
                else if (ibh_quad_variant == 518) S3_LAUNCH(3);   // A/B: the strip form (round 2)
                else if (ibh_quad_variant == 519) C3_LAUNCH(4);   // (7 registers spilled: 46 against 41 us at 4.56 M cells)
                else if (ibh_quad_variant == 520) C3_LAUNCH(5);
                else C3_LAUNCH(3);
#undef C3_LAUNCH
#undef S3_LAUNCH
            } else  // A/B: thread-per-cell form
            hipLaunchKernelGGL(k_sweep3_adv, dim3(p->nblk), dim3(512), 0, ibh_stream, u, C, (uint32_t)ldc, ud, p->blocks3,
                               p->htab3, p->ftab3, p->rtab3, p->r4tab3, p->nblk);
            IBH_LAUNCH_CHECK();
            return 0;
        }
        if ((rc = ensure_G(p))) return rc;
        // 3-D block path: block kernels over the requested block range + face-list kernels over the rest
        const int32_t a0 = ph2 ? p->nA1 : 0, a1 = ph1 ? p->nA1 : p->nblk;
        const int32_t b0 = ph2 ? p->nB1 : 0, b1 = ph1 ? p->nB1 : p->nblk;
        const int32_t nI = ph1 ? 0 : p->n_irr;
        PartView v = view(p);
        const bool doA = !(flags & IBH_PASS_B_ONLY), doB = !(flags & IBH_PASS_A_ONLY);
        const int32_t gI = (nI + 511) / 512;
        if (doA && (a1 > a0 || gI) && ibh_3d_wave) {
            const int32_t nwgA = (a1 - a0 + 3) / 4, gIw = (nI + 255) / 256;
            hipLaunchKernelGGL(k_passA3_wave, dim3(nwgA + gIw), dim3(256), 0, ibh_stream, v, u, p->G, p->blocks3 + a0,
                               p->htab3 + (size_t)a0 * 384, p->ftab3, a1 - a0, nwgA, p->irr_cells, nI,
                               flat_of(p, p->irr_cells), (const int32_t*)nullptr);
        } else if (doA && (a1 > a0 || gI))
            hipLaunchKernelGGL(k_passA3_blk, dim3(a1 - a0 + gI), dim3(512), 0, ibh_stream, v, u, p->G, p->blocks3 + a0,
                               p->htab3 + (size_t)a0 * 384, p->ftab3, a1 - a0, p->irr_cells, nI, flat_of(p, p->irr_cells));
        if (doB && (b1 > b0 || gI))
            hipLaunchKernelGGL(k_passB3_adv_blk, dim3(b1 - b0 + gI), dim3(512), 0, ibh_stream, v, u, C, ldc, p->G, ud,
                               p->blocks3 + b0, p->htab3 + (size_t)b0 * 384, p->ftab3, b1 - b0, p->irr_cells, nI,
                               flat_of(p, p->irr_cells), (const int32_t*)nullptr);
        IBH_LAUNCH_CHECK();
        return 0;
    }
    const bool tuned2 = p->nd == 2 && p->bs == 8 && p->nblk > 0 &&
                        !(flags & (IBH_FORCE_GENERAL | IBH_EXACT | IBH_NO_FUSE | IBH_PASS_A_ONLY | IBH_PASS_B_ONLY));
    // single-kernel sweep (blk2::sweep_adv) over the eligible blocks: `count` list positions from `first`
    auto launch_sweep = [&](const int32_t* list, int32_t first, int32_t count) {
        if (count <= 0) return;
        // blocks per wave: keep enough waves to fill the chip before a wave takes a second block
        // (measured on 13.5 k and 54 k blocks, scripts/sweep_iters.sh: 2-3 and 4-6 blocks per wave are best)
        const int32_t iters = ibh_sweep_iters > 0 ? ibh_sweep_iters : std::min(6, std::max(1, count / 6000));
        const int32_t nwg = (count + WPB * iters - 1) / (WPB * iters);
        const BlockDesc2* bl = list ? p->blocks2 : p->blocks2 + first;
        const int32_t* ht = list ? p->htab : p->htab + (size_t)first * 64;
        const int32_t* et = list ? p->etab : p->etab + (size_t)first * 16;
        const int32_t* ls = list ? list + first : nullptr;
        if (p->n_dt > 0)  // some blocks take their deeper cells from the table (skirt fragments)
            hipLaunchKernelGGL(k_sweep_adv<true>, dim3(nwg), dim3(64 * WPB), 0, ibh_stream, u, C, (uint32_t)ldc, ud, bl, ht,
                               et, p->dtab, count, nwg, iters, ls);
        else
            hipLaunchKernelGGL(k_sweep_adv<false>, dim3(nwg), dim3(64 * WPB), 0, ibh_stream, u, C, (uint32_t)ldc, ud, bl, ht,
                               et, p->dtab, count, nwg, iters, ls);
    };
    // quad sweep over quad set `k` (0: all blocks, 1: image blocks): `ph1`/`ph2` select the interior / boundary part
    auto quads_ok = [&](int k) {
        return ibh_quad && !(flags & IBH_NO_QUAD) && p->nq[k] > 0;
    };
    auto launch_quads = [&](int k, bool ph1, bool ph2) {
        int32_t q0 = ph2 ? p->nq_int[k] : 0, q1 = ph1 ? p->nq_int[k] : p->nq[k];
        int32_t s0 = ph2 ? p->nqs_int[k] : 0, s1 = ph1 ? p->nqs_int[k] : p->nqs[k];
        if (q1 - q0 + s1 - s0 <= 0) return;
        if (ibh_quad_parts == 1) s1 = s0;  // measurement: quads only / single blocks only
        if (ibh_quad_parts == 2) q1 = q0;
        // The blocks outside quads as a SECOND launch of the row sweep (rows2::sweep_rows over the list: any eight complete
        // blocks per wave, 110 vector instructions per block against 365 in the per-block kernel) where a second launch
        // is cheap against the sweep ("rows_singles": -1 = by size, 0 / 1 = never / always)
        // pair tiles (set 0, whole sweeps or the interior phase -- they exist only where every block is interior): the
        // single blocks are then the ones outside quads AND pairs
        const int32_t npair =
            (k == 0 && ibh_pairs && p->npair > 0 && !ph2 && q0 == 0 && q1 == p->nq[k] && ibh_quad_parts == 3) ? p->npair : 0;
        const int32_t* slist = p->qsingles[k];
        if (npair) {
            slist = p->qsingles2;
            s0 = 0;
            s1 = p->nqs2;
        } else if (k == 0 && p->npair > 0 && ph2) {
            s1 = s0;  // (all blocks are interior blocks there: nothing in the boundary phase)
        }
        const bool rows_singles = k == 0 && p->rows_ok && p->n_dt == 0 && ibh_quad_variant == 0 && s1 > s0 &&
                                  (ibh_rows_singles < 0 ? s1 - s0 >= IBH_ROWS_SINGLES_MIN : ibh_rows_singles > 0);
        const int32_t rs0 = s0, rs1 = s1;
        const bool rows_inside = k == 0 && p->rows_ok && p->n_dt == 0 && ibh_quad_variant == 0 && s1 > s0 && ibh_rows_singles == 2;
        if (rows_singles && !rows_inside) s1 = s0;
        const int32_t siters = ibh_quad_singles_iters > 0 ? ibh_quad_singles_iters : 1;
        const int32_t nwgq = (q1 - q0 + npair + WPB - 1) / WPB,
                      nwgs = rows_inside ? (s1 - s0 + WPB * 8 - 1) / (WPB * 8) : (s1 - s0 + WPB * siters - 1) / (WPB * siters);
        if (nwgq + nwgs == 0 && !rows_singles) return;
#define QUAD_LAUNCH(DT, STAMP, ...)                                                                                    \
    hipLaunchKernelGGL((k_sweep_quad<DT, STAMP, ##__VA_ARGS__>), dim3(nwgq + nwgs), dim3(64 * WPB), 0, ibh_stream, u, C,              \
                       (uint32_t)ldc, ud, p->qd[k] + q0, p->qtab[k] + (size_t)q0 * IBH_QROW, q1 - q0, nwgq,            \
                       p->blocks2, p->htab, p->etab, p->dtab, slist + s0, s1 - s0, nwgs, ibh_quad_singles_first, siters,          \
                       (const float*)nullptr, npair, ibh_arith_ids ? p->qaux[k] + (size_t)q0 * IBH_QAUX : (const int32_t*)nullptr)
        if (rows_inside) QUAD_LAUNCH(false, false, 127, false, true);
        else if (p->n_dt > 0) QUAD_LAUNCH(true, false);
        else if (ibh_quad_variant == 4) QUAD_LAUNCH(false, true);
        else if (ibh_quad_variant == 126) QUAD_LAUNCH(false, false, 126);  // A/B: seven 4-byte gathers
        else if (ibh_quad_variant == 85) QUAD_LAUNCH(false, false, 85);  // measurement: subsets of the halo gathers
        else if (ibh_quad_variant == 69) QUAD_LAUNCH(false, false, 69);
        else if (ibh_quad_variant == 5) QUAD_LAUNCH(false, false, 5);
        else if (ibh_quad_variant == 100) QUAD_LAUNCH(false, false, 0);
        else if (nwgq + nwgs > 0) QUAD_LAUNCH(false, false);
#undef QUAD_LAUNCH
        if (rows_singles && !rows_inside) {
            const int32_t nw = (rs1 - rs0 + 7) / 8, nwgr = (nw + WPBR - 1) / WPBR;
            hipLaunchKernelGGL(k_sweep_rows, dim3(nwgr), dim3(64 * WPBR), 0, ibh_stream, u, C, (uint32_t)ldc, ud, p->blocks2,
                               p->etab, 0, rs1 - rs0, nwgr, slist + rs0);
        }
    };
    if (tuned2 && (flags & IBH_IMAGE_ONLY) && p->img_all_fz && !p->fuse_all) {
        // only the image cells are wanted (a rank of a multi-GPU run) and every image block is eligible: one launch
        // per phase over the image blocks, no workspace, nothing for the skirt fragments
        const bool ph1 = (flags & IBH_PHASE_INTERIOR) != 0, ph2 = (flags & IBH_PHASE_BOUNDARY) != 0;
        IBH_REQUIRE(!(ph1 && ph2), "IBH_PHASE_INTERIOR and IBH_PHASE_BOUNDARY are exclusive");
        const int32_t i0 = ph2 ? p->n_img_int : 0, i1 = ph1 ? p->n_img_int : p->n_img;
        if (quads_ok(1)) launch_quads(1, ph1, ph2);
        else launch_sweep(p->img_list, i0, i1 - i0);
        IBH_LAUNCH_CHECK();
        return 0;
    }
    if (tuned2 && p->fuse_all) {
        // every block is eligible: the whole sweep (or one overlap phase of it) is one launch
        const bool ph1 = (flags & IBH_PHASE_INTERIOR) != 0, ph2 = (flags & IBH_PHASE_BOUNDARY) != 0;
        IBH_REQUIRE(!(ph1 && ph2), "IBH_PHASE_INTERIOR and IBH_PHASE_BOUNDARY are exclusive");
        const int32_t b0 = ph2 ? p->nB1 : 0, b1 = ph1 ? p->nB1 : p->nblk;
        if (p->rows_ok && ibh_rows && !(flags & IBH_NO_QUAD) && ibh_quad_variant == 0) {
            // row / column sweep: eight blocks per wavefront, arithmetic halo ids (`quad_variant` != 0: the quad forms)
            const int32_t nw = (b1 - b0 + 7) / 8, nwg = (nw + WPBR - 1) / WPBR;
            if (nwg > 0)
                hipLaunchKernelGGL(k_sweep_rows, dim3(nwg), dim3(64 * WPBR), 0, ibh_stream, u, C, (uint32_t)ldc, ud,
                                   p->blocks2, p->etab, b0, b1 - b0, nwg, (const int32_t*)nullptr);
        } else if (quads_ok(0)) launch_quads(0, ph1, ph2);
        else launch_sweep(nullptr, b0, b1 - b0);
        IBH_LAUNCH_CHECK();
        return 0;
    }
    // A mixed launch is three kernels where the two-kernel form is two: at ~3 us per launch it only pays when the
    // single kernel saves more than that (0.26 ns per eligible block: scripts/mixed_ab.py).  The choice depends on
    // the partition only, never on the phase flags: a sweep split in phases reproduces the whole sweep bit for bit.
    const bool mixed_pays = p->n_fz >= 12000 || (flags & (IBH_FORCE_MIXED | IBH_SWEEP_ONLY));
    if (tuned2 && p->fz_list && 4 * (int64_t)p->n_fz >= p->nblk && mixed_pays) {
        // mixed: eligible blocks in one kernel; the rest (skirt blocks, blocks next to face-list cells) in the
        // two-kernel form, with the gradient workspace filled only where it is read (ng_list)
        const bool ph1 = (flags & IBH_PHASE_INTERIOR) != 0, ph2 = (flags & IBH_PHASE_BOUNDARY) != 0;
        IBH_REQUIRE(!(ph1 && ph2), "IBH_PHASE_INTERIOR and IBH_PHASE_BOUNDARY are exclusive");
        const int32_t f0 = ph2 ? p->n_fz_int : 0, f1 = ph1 ? p->n_fz_int : p->n_fz;
        // interior phase without blocks of the two-kernel form: nothing reads the workspace before the boundary
        // phase, so all of pass A is done there and the interior phase is one launch
        const bool defer = p->n_nf_int == 0;
        const int32_t g0 = ph2 ? (defer ? 0 : p->n_ng_int) : 0, g1 = ph1 ? (defer ? 0 : p->n_ng_int) : p->n_ng;
        const int32_t r0 = ph2 ? p->n_nf_int : 0, r1 = ph1 ? p->n_nf_int : p->n_nf;
        const int32_t nI = ph1 ? 0 : p->n_irr;
        const int32_t gI = (nI + 64 * WPB - 1) / (64 * WPB);
        PartView v = view(p);
        if ((rc = ensure_G(p))) return rc;
        const int32_t nwgA = (g1 - g0 + WPB - 1) / WPB, nwgB = (r1 - r0 + WPB - 1) / WPB;
        if (!(flags & IBH_SWEEP_ONLY)) {
            if (nwgA + gI)
                hipLaunchKernelGGL((k_passA<2, 1, false>), dim3(nwgA + gI), dim3(64 * WPB), 0, ibh_stream, v, u,
                                   (int64_t)p->nc, p->G, p->blocks2, p->htab, g1 - g0, nwgA, p->irr_cells, nI,
                                   flat_of(p, p->irr_cells), p->ng_list + g0);
            if (nwgB + gI)
                hipLaunchKernelGGL((k_passB_adv<2, false>), dim3(nwgB + gI), dim3(64 * WPB), 0, ibh_stream, v, u, C, ldc,
                                   p->G, ud, p->blocks2, p->htab, r1 - r0, nwgB, p->irr_cells, nI,
                                   flat_of(p, p->irr_cells), p->nf_list + r0);
        }
        launch_sweep(p->fz_list, f0, f1 - f0);
        IBH_LAUNCH_CHECK();
        return 0;
    }
    if ((rc = ensure_G(p))) return rc;
    const bool fast = p->bs == 8 && p->nd == 2 && p->nblk > 0 && !(flags & IBH_FORCE_GENERAL);
    const int bpwg = WPB;  // blocks per workgroup
    // overlap phases: INTERIOR = blocks independent of skirt data, BOUNDARY = the rest + face-list cells
    const bool ph1 = (flags & IBH_PHASE_INTERIOR) != 0, ph2 = (flags & IBH_PHASE_BOUNDARY) != 0;
    IBH_REQUIRE(!(ph1 && ph2), "IBH_PHASE_INTERIOR and IBH_PHASE_BOUNDARY are exclusive");
    IBH_REQUIRE(!(ph1 || ph2) || fast, "overlap phases need the block path (2-D, block_size 8, domain given)");
    const int32_t a0 = ph2 ? p->nA1 : 0, a1 = ph1 ? p->nA1 : p->nblk;  // pass A block range
    const int32_t b0 = ph2 ? p->nB1 : 0, b1 = ph1 ? p->nB1 : p->nblk;  // pass B block range
    const int32_t nwgA_fast = fast ? (a1 - a0 + bpwg - 1) / bpwg : 0;
    const int32_t nwgB_fast = fast ? (b1 - b0 + bpwg - 1) / bpwg : 0;
    // pass A always covers every cell of the partition (skirt cells feed the faces of image cells)
    const int32_t* cellsA = fast ? p->irr_cells : nullptr;
    const int32_t nA = fast ? (ph1 ? 0 : p->n_irr) : p->nc;
    // pass B: every cell, or image cells only
    const int32_t* cellsB = cellsA;
    int32_t nB = nA;
    if ((flags & IBH_IMAGE_ONLY) && !fast) {
        cellsB = p->image_in_domain;
        nB = p->n_image;
    }
    PartView v = view(p);
    dim3 blk(64 * WPB);
    dim3 gA(nwgA_fast + (nA + 64 * WPB - 1) / (64 * WPB)), gB(nwgB_fast + (nB + 64 * WPB - 1) / (64 * WPB));
    const bool exact = (flags & IBH_EXACT) != 0;
    const bool doA = gA.x && !(flags & IBH_PASS_B_ONLY), doB = gB.x && !(flags & IBH_PASS_A_ONLY);
    const BlockDesc2* blkA = p->blocks2 ? p->blocks2 + a0 : nullptr;
    const BlockDesc2* blkB = p->blocks2 ? p->blocks2 + b0 : nullptr;
    const int32_t* htA = p->htab ? p->htab + (size_t)a0 * 64 : nullptr;
    const int32_t* htB = p->htab ? p->htab + (size_t)b0 * 64 : nullptr;
    if (p->nd == 2 && exact) {
        if (doA)
            hipLaunchKernelGGL((k_passA<2, 1, true>), gA, blk, 0, ibh_stream, v, u, (int64_t)p->nc, p->G, blkA, htA,
                               a1 - a0, nwgA_fast, cellsA, nA, flat_of(p, cellsA), nullptr);
        if (doB)
            hipLaunchKernelGGL((k_passB_adv<2, true>), gB, blk, 0, ibh_stream, v, u, C, ldc, p->G, ud, blkB, htB,
                               b1 - b0, nwgB_fast, cellsB, nB, flat_of(p, cellsB), nullptr);
    } else if (p->nd == 2) {
        if (doA)
            hipLaunchKernelGGL((k_passA<2, 1, false>), gA, blk, 0, ibh_stream, v, u, (int64_t)p->nc, p->G, blkA, htA,
                               a1 - a0, nwgA_fast, cellsA, nA, flat_of(p, cellsA), nullptr);
        if (doB)
            hipLaunchKernelGGL((k_passB_adv<2, false>), gB, blk, 0, ibh_stream, v, u, C, ldc, p->G, ud, blkB, htB,
                               b1 - b0, nwgB_fast, cellsB, nB, flat_of(p, cellsB), nullptr);
    } else {
        if (doA)
            hipLaunchKernelGGL((k_passA<3, 1, true>), gA, blk, 0, ibh_stream, v, u, (int64_t)p->nc, p->G, p->blocks2,
                               p->htab, p->nblk, 0, cellsA, nA, flat_of(p, cellsA), nullptr);
        if (doB)
            hipLaunchKernelGGL((k_passB_adv<3, true>), gB, blk, 0, ibh_stream, v, u, C, ldc, p->G, ud, p->blocks2,
                               p->htab, p->nblk, 0, cellsB, nB, flat_of(p, cellsB), nullptr);
    }
    IBH_LAUNCH_CHECK();
    return 0;
}

// One step of a rank in one launch: xGMI halo exchange of u + image-only quad sweep (k_step_quad).  Needs a partition
// whose image blocks are all eligible and carry quads (the ranks of the benchmark meshes); otherwise the caller runs
// ibh_halo_exchange and ibh_residual_advection(IBH_IMAGE_ONLY) one after the other (same result).
int ibh_step_advection(ibh_part* p, const float* u, float* u_out, const float* C, int64_t ldc, const float* dt_dev,
                       const ibh_bcset* bcs) {
    IBH_REQUIRE(p && u && u_out && C && dt_dev && u != u_out, "ibh_step_advection: null or aliased argument");
    if (p->nc == 0) return 0;
    int rc = 0;
    if (p->nd == 2 && p->bs == 8 && p->nblk > 0 && p->fuse_all && ibh_quad && p->nq[0] > 0 && p->n_dt == 0) {
        // sweep and update in one launch: the quad sweep stores u + dt * residual (its cells of u are in registers)
        const int32_t npair = ibh_pairs ? p->npair : 0;
        const int32_t nq = p->nq[0], ns = npair ? p->nqs2 : p->nqs[0];
        const int32_t nwgq = (nq + npair + WPB - 1) / WPB, nwgs = (ns + WPB - 1) / WPB;
        hipLaunchKernelGGL((k_sweep_quad<false, false, 127, true>), dim3(nwgq + nwgs), dim3(64 * WPB), 0, ibh_stream, u, C,
                           (uint32_t)ldc, u_out, p->qd[0], p->qtab[0], nq, nwgq, p->blocks2, p->htab, p->etab, p->dtab,
                           npair ? p->qsingles2 : p->qsingles[0], ns, nwgs, ibh_quad_singles_first, 1, dt_dev, npair,
                           ibh_arith_ids ? p->qaux[0] : (const int32_t*)nullptr);
        IBH_LAUNCH_CHECK();
    } else {
        if ((rc = ibh_residual_advection(p, u, C, ldc, u_out, 0))) return rc;
        if ((rc = ibh_update_dev(p->nc, dt_dev, u, u_out, u_out))) return rc;
    }
    if (bcs) rc = ibh_bcset_apply(bcs, u_out);
    return rc;
}

// The step with the time step of the NEXT step evaluated beside its boundary conditions (ibh_bcset_apply_with_dt,
// ibh_ops.hip): dt_next = scale * 0.5 / max(...) depends on C alone, so its two launches ride in the two launches of the BC
// set instead of standing in front of the next sweep.  dt_next may be dt_dev (the sweep has read it by then).
// (Tried and dropped: the partial maxima beside the SWEEP instead -- its workgroups then carry the sweep's 27 KB of LDS and
// its register budget, and the launch takes longer than the two side by side save: 26.8 against 26.1 us per step.)
extern "C" int ibh_bcset_apply_with_dt(const ibh_bcset* s, float* a, ibh_part* p, const float* C, int64_t ldc, float scale,
                                       float* dt_next, int partials_done);
int ibh_step_advection_dt(ibh_part* p, const float* u, float* u_out, const float* C, int64_t ldc, const float* dt_dev,
                          const ibh_bcset* bcs, float scale, float* dt_next) {
    IBH_REQUIRE(dt_next, "ibh_step_advection_dt: null dt_next");
    int rc = ibh_step_advection(p, u, u_out, C, ldc, dt_dev, nullptr);
    if (rc) return rc;
    if (bcs && bcs->ng > 0) return ibh_bcset_apply_with_dt(bcs, u_out, p, C, ldc, scale, dt_next, 0);
    return ibh_timestep_advection(p, C, ldc, scale, dt_next);
}

int ibh_step_advection_xgmi(ibh_part* p, float* u, const float* C, int64_t ldc, float* ud, const int32_t* send_all,
                            int n_send_peers, const int32_t* send_seg, float* const* dst0, float* const* dst1,
                            uint32_t* const* send_flags, const int32_t* recv_all, const float* src0, const float* src1,
                            int n_recv_peers, const int32_t* recv_seg, const uint32_t* const* recv_flags,
                            uint32_t* state, uint32_t max_spins, unsigned long long* fstate) {
    IBH_REQUIRE(p && u && C && ud && state && fstate, "ibh_step_advection_xgmi: null argument");
    IBH_REQUIRE(p->nd == 2 && p->bs == 8 && p->nblk > 0 && p->img_all_fz && !p->fuse_all && p->nq[1] > 0 && ibh_quad,
                "ibh_step_advection_xgmi: needs a 2-D partition with skirt fragments whose image blocks are all eligible "
                "for the quad sweep");
    IBH_REQUIRE(n_send_peers >= 0 && n_send_peers <= IBH_MAX_PEERS && n_recv_peers >= 0 && n_recv_peers <= IBH_MAX_PEERS &&
                    n_send_peers + n_recv_peers > 0,
                "ibh_step_advection_xgmi: 1 to 16 peers");
    XchgArgs A;
    memset(&A, 0, sizeof(A));
    A.ns = n_send_peers;
    A.nr = n_recv_peers;
    if (n_send_peers) {
        IBH_REQUIRE(send_all && send_seg && dst0 && dst1 && send_flags, "ibh_step_advection_xgmi: null send argument");
        for (int q = 0; q < n_send_peers; ++q) {
            A.dst[0][q] = dst0[q];
            A.dst[1][q] = dst1[q];
            A.sflag[q] = send_flags[q];
            A.sseg[q] = send_seg[q];
        }
        A.sseg[n_send_peers] = send_seg[n_send_peers];
    }
    if (n_recv_peers) {
        IBH_REQUIRE(recv_all && src0 && src1 && recv_seg && recv_flags, "ibh_step_advection_xgmi: null receive argument");
        for (int q = 0; q < n_recv_peers; ++q) {
            A.rflag[q] = recv_flags[q];
            A.rseg[q] = recv_seg[q];
        }
        A.rseg[n_recv_peers] = recv_seg[n_recv_peers];
    }
    const int32_t big = std::max(A.ns ? A.sseg[A.ns] : 0, A.nr ? A.rseg[A.nr] : 0);
    int E = (big + 255) / 256;
    E = E < 1 ? 1 : E > 64 ? 64 : E;
    const int32_t nq = p->nq[1], nqi = p->nq_int[1], ns = p->nqs[1], nsi = p->nqs_int[1];
    const int32_t nwg = (nqi + WPB - 1) / WPB + (nsi + WPB - 1) / WPB + (nq - nqi + WPB - 1) / WPB + (ns - nsi + WPB - 1) / WPB;
    static_assert(WPB == 4, "the exchange workgroups of k_step_quad are 256 threads");
    if (p->n_dt > 0)
        hipLaunchKernelGGL(k_step_quad<true>, dim3(E + nwg), dim3(64 * WPB), 0, ibh_stream, u, C, (uint32_t)ldc, ud, p->qd[1],
                           p->qtab[1], nqi, nq, p->blocks2, p->htab, p->etab, p->dtab, p->qsingles[1], nsi, ns, send_all,
                           recv_all, src0, src1, A, state, max_spins, E, fstate);
    else
        hipLaunchKernelGGL(k_step_quad<false>, dim3(E + nwg), dim3(64 * WPB), 0, ibh_stream, u, C, (uint32_t)ldc, ud,
                           p->qd[1], p->qtab[1], nqi, nq, p->blocks2, p->htab, p->etab, p->dtab, p->qsingles[1], nsi, ns,
                           send_all, recv_all, src0, src1, A, state, max_spins, E, fstate);
    IBH_LAUNCH_CHECK();
    return 0;
}

// cell_gradient(part, u) -- the tuple form (ImmersedBoundary.jl:980-988): the gradients of `nv` fields along ALL
// dimensions in one sweep per field, and (optionally) the JST sensor of every field (JST_sensor(part, u), :1077-1097, the
// maximum over the dimensions).  On a block-structured partition this is pass A of the two-kernel sweeps (block kernels,
// face-list threads for the cells outside blocks; tuned arithmetic: reciprocal spacings, inside 5e-6 norm-wise of the
// operator-by-operator kernels); elsewhere it falls back to ibh_cell_gradient / ibh_jst_sensor per dimension.
//   out:    (nc, nd*nv) column-major, gradient of field v along dimension d in column d*nv + v
//   sensor: (nc, nv) or null
int ibh_cell_gradient_nd(ibh_part* p, const float* u, int nv, int64_t ldu, float* out, int64_t ldo, float* sensor,
                         int64_t lds) {
    IBH_REQUIRE(p && u && out && nv >= 1, "ibh_cell_gradient_nd: bad argument");
    if (p->nc == 0) return 0;
    const int nd = p->nd;
    const bool blocks = p->bs == 8 && p->nblk > 0 && (nd == 2 ? p->blocks2 != nullptr : p->blocks3 != nullptr);
    if (!blocks) {
        // no block structure (e.g. the coarse levels of multigrid()): every dimension in one face-list launch
        const int rc = ibh_cell_gradient_all(p, u, nv, ldu, out, ldo);
        if (rc) return rc;
        if (sensor) return ibh_jst_sensor(p, 0, u, nv, ldu, sensor, lds);
        return 0;
    }
    if (nv == 1 && ldo == p->nc && sensor == out + (size_t)nd * ldo && lds == ldo) {
        // one field, gradients and sensor back to back: pass A writes them in place (the output IS its workspace for the
        // duration of the call: no copy, and the partition's own workspace is not even allocated)
        float* const own = p->G;
        p->G = out;
        const int rc1 = ibh_residual_advection(p, u, u, p->nc, out, IBH_PASS_A_ONLY | IBH_NO_FUSE);
        p->G = own;
        return rc1;
    }
    int rc = ensure_G(p);
    if (rc) return rc;
    for (int v = 0; v < nv; ++v) {
        const float* uv = u + (size_t)v * ldu;
        // pass A of the scalar sweep: G = [grad_1 .. grad_nd, sensor], each nc floats (velocity / output are not touched)
        if ((rc = ibh_residual_advection(p, uv, uv, p->nc, p->G, IBH_PASS_A_ONLY | IBH_NO_FUSE))) return rc;
        for (int d = 0; d < nd; ++d)
            IBH_HIP(hipMemcpyAsync(out + (size_t)(d * nv + v) * ldo, p->G + (size_t)d * p->nc, sizeof(float) * p->nc,
                                   hipMemcpyDeviceToDevice, ibh_stream));
        if (sensor)
            IBH_HIP(hipMemcpyAsync(sensor + (size_t)v * lds, p->G + (size_t)nd * p->nc, sizeof(float) * p->nc,
                                   hipMemcpyDeviceToDevice, ibh_stream));
    }
    return 0;
}

// The same gradients FIELD by field: out is (nc, nv * (nd + 1)) column-major with leading dimension nc, the gradient of field
// v along dimension d in column v * (nd + 1) + d and the JST sensor of field v in column v * (nd + 1) + nd -- the layout pass A
// writes, so that on a block-structured partition every field's sweep writes in place (ibh_cell_gradient_nd copies nd
// columns per field out of the partition's workspace: 9 device-to-device copies for the velocity gradients of a 3-D
// closure, 0.3 ms of a configs[4] V-cycle at 7.9 M cells).  The gradient of all fields along d is the strided view
// out[:, d : nv * (nd + 1) : nd + 1] (leading dimension (nd + 1) * nc).
int ibh_cell_gradient_fields(ibh_part* p, const float* u, int nv, int64_t ldu, float* out) {
    IBH_REQUIRE(p && u && out && nv >= 1, "ibh_cell_gradient_fields: bad argument");
    if (p->nc == 0) return 0;
    const int nd = p->nd;
    const bool blocks = p->bs == 8 && p->nblk > 0 && (nd == 2 ? p->blocks2 != nullptr : p->blocks3 != nullptr);
    for (int v = 0; v < nv; ++v) {
        const float* uv = u + (size_t)v * ldu;
        float* ov = out + (size_t)v * (nd + 1) * p->nc;
        int rc;
        if (blocks) {
            float* const own = p->G;
            p->G = ov;
            rc = ibh_residual_advection(p, uv, uv, p->nc, ov, IBH_PASS_A_ONLY | IBH_NO_FUSE);
            p->G = own;
        } else {
            rc = ibh_cell_gradient_all(p, uv, 1, p->nc, ov, p->nc);
            if (!rc) rc = ibh_jst_sensor(p, 0, uv, 1, p->nc, ov + (size_t)nd * p->nc, p->nc);
        }
        if (rc) return rc;
    }
    return 0;
}

// every cell of the partition in a complete 8^3 block without a GENERAL side (what the fused closures below need)
static bool all_blocks3(const ibh_part* p) {
    return p->nd == 3 && p->bs == 8 && p->blocks3 && p->nblk > 0 && p->n_irr == 0 && (int64_t)p->nblk * 512 == p->nc &&
           p->info[6] == 0;
}
// ibh_scalar_transport on an all-block 3-D partition (dispatched from ibh_turb.hip; 0 = not applicable here)
int ibh_scalar_transport_blocks(const ibh_part* p, const float* R, const float* nuR, float nu, const float* vel, int64_t ldv,
                                const float* S, float* out, int* done) {
    *done = 0;
    if (!all_blocks3(p) || !ibh_transport_blocks) return 0;
    const int32_t nwg = (p->nblk + 3) / 4;
    hipLaunchKernelGGL(k_scalar_transport_blocks3, dim3(nwg), dim3(256), 0, ibh_stream, p->blocks3, p->htab3, p->ftab3,
                       p->nblk, nwg, (uint32_t)p->nc, p->spacing, R, nuR, nu, vel, (uint32_t)ldv, S, out);
    IBH_LAUNCH_CHECK();
    *done = 1;
    return 0;
}

// no block structure at all: the tuple cell_gradient is the face-list kernel there (ibh_cell_gradient_nd)
static bool no_blocks(const ibh_part* p) {
    return !(p->bs == 8 && p->nblk > 0 && (p->nd == 2 ? p->blocks2 != nullptr : p->blocks3 != nullptr));
}
int ibh_shear_rate_of_velocity_grad(ibh_part* p, const float* vel, int64_t ldv, float* S, float* G, int64_t ldg) {
    IBH_REQUIRE(p && vel && S, "ibh_shear_rate_of_velocity: null argument");
    IBH_REQUIRE(!G || ldg >= p->nc, "ibh_shear_rate_of_velocity_grad: ldg < nc");
    if (p->nc == 0) return 0;
    if (no_blocks(p)) return ibh_shear_rate_of_velocity_cells(p, vel, ldv, S, G, ldg);
    IBH_REQUIRE(all_blocks3(p), "ibh_shear_rate_of_velocity: needs a 3-D partition made of complete blocks or one without "
                                "block structure (compose cell_gradient and shear_rate otherwise)");
    FieldPtrs<3> V{{vel, vel + ldv, vel + 2 * ldv}};
    const int32_t nwg = (p->nblk + 3) / 4;
    hipLaunchKernelGGL(k_shear_of_velocity3, dim3(nwg), dim3(256), 0, ibh_stream, p->blocks3, p->htab3, p->ftab3, p->nblk, nwg,
                       V, S, G, (uint32_t)ldg);
    IBH_LAUNCH_CHECK();
    return 0;
}
int ibh_shear_rate_of_velocity(ibh_part* p, const float* vel, int64_t ldv, float* S) {
    return ibh_shear_rate_of_velocity_grad(p, vel, ldv, S, nullptr, 0);
}
int ibh_wray_agarwal_of(ibh_part* p, const float* R, const float* S, float sigmaR, float C1, float kappa, float* nut,
                        float* nuR, float* Sout) {
    IBH_REQUIRE(p && R && S && nut && nuR && Sout, "ibh_wray_agarwal_of: null argument");
    if (p->nc == 0) return 0;
    if (no_blocks(p)) return ibh_wray_agarwal_of_cells(p, R, S, sigmaR, C1, kappa, nut, nuR, Sout);
    IBH_REQUIRE(all_blocks3(p), "ibh_wray_agarwal_of: needs a 3-D partition made of complete blocks or one without block "
                                "structure (compose cell_gradient and Wray_Agarwal otherwise)");
    FieldPtrs<2> RS{{R, S}};
    const int32_t nwg = (p->nblk + 3) / 4;
    hipLaunchKernelGGL(k_wray_agarwal_of3, dim3(nwg), dim3(256), 0, ibh_stream, p->blocks3, p->htab3, p->ftab3, p->nblk, nwg,
                       RS, sigmaR, C1, kappa, nut, nuR, Sout);
    IBH_LAUNCH_CHECK();
    return 0;
}

int ibh_residual_euler_hll(ibh_part* p, const float* P, int64_t ldp, float* R, int64_t ldr, const ibh_fluid* fluid,
                           int flags) {
    IBH_REQUIRE(p && P && R && fluid, "ibh_residual_euler_hll: null argument");
    if (p->nc == 0) return 0;
    const bool tuned2e = p->nd == 2 && p->bs == 8 && p->nblk > 0 &&
                         !(flags & (IBH_FORCE_GENERAL | IBH_EXACT | IBH_NO_FUSE | IBH_PASS_A_ONLY | IBH_PASS_B_ONLY));
    if (tuned2e && (p->fuse_all || ((flags & IBH_IMAGE_ONLY) && p->img_all_fz))) {
        // every block eligible (or only the image blocks wanted and all of them eligible: a rank of a multi-GPU run):
        // the Euler sweep is one launch per phase, nothing goes through the workspace
        const bool ph1 = (flags & IBH_PHASE_INTERIOR) != 0, ph2 = (flags & IBH_PHASE_BOUNDARY) != 0;
        IBH_REQUIRE(!(ph1 && ph2), "IBH_PHASE_INTERIOR and IBH_PHASE_BOUNDARY are exclusive");
        const int32_t* list = p->fuse_all ? nullptr : p->img_list;
        const int32_t nall = p->fuse_all ? p->nblk : p->n_img, nint = p->fuse_all ? p->nB1 : p->n_img_int;
        const int32_t i0 = ph2 ? nint : 0, i1 = ph1 ? nint : nall;
        const int32_t count = i1 - i0;
        const int k = p->fuse_all ? 0 : 1;
        if (ibh_quad && !(flags & IBH_NO_QUAD) && p->nq[k] > 0) {
            int32_t s0 = ph2 ? p->nqs_int[k] : 0, s1 = ph1 ? p->nqs_int[k] : p->nqs[k];
            int32_t q0 = ph2 ? p->nq_int[k] : 0, q1 = ph1 ? p->nq_int[k] : p->nq[k];
            if (ibh_quad_parts == 1) s1 = s0;  // measurement: quads only / single blocks only
            if (ibh_quad_parts == 2) q1 = q0;
            const int32_t nwgq = (q1 - q0 + WPBE - 1) / WPBE, nwgs = (s1 - s0 + WPBE - 1) / WPBE;
            if (nwgq + nwgs > 0)
                hipLaunchKernelGGL(k_sweep_quad_euler, dim3(nwgq + nwgs), dim3(64 * WPBE), 0, ibh_stream, P, (uint32_t)ldp,
                                   R, (uint32_t)ldr, fluid->R, fluid->gamma, p->qd[k] + q0,
                                   p->qtab[k] + (size_t)q0 * IBH_QROW, q1 - q0, nwgq, p->blocks2, p->htab, p->etab, p->dtab,
                                   p->qsingles[k] + s0, s1 - s0, nwgs, ibh_quad_singles_first);
        } else if (count > 0) {
            const int32_t iters = ibh_sweep_iters > 0 ? ibh_sweep_iters : std::min(4, std::max(1, count / 6000));
            const int32_t nwg = (count + WPBE * iters - 1) / (WPBE * iters);
            const BlockDesc2* bl = list ? p->blocks2 : p->blocks2 + i0;
            const int32_t* ht = list ? p->htab : p->htab + (size_t)i0 * 64;
            const int32_t* et = list ? p->etab : p->etab + (size_t)i0 * 16;
            hipLaunchKernelGGL(k_sweep_euler, dim3(nwg), dim3(64 * WPBE), 0, ibh_stream, P, (uint32_t)ldp, R, (uint32_t)ldr,
                               fluid->R, fluid->gamma, bl, ht, et, p->dtab, count, nwg, iters, list ? list + i0 : nullptr);
        }
        IBH_LAUNCH_CHECK();
        return 0;
    }
    if (p->nd == 3 && p->img_all3 && (flags & IBH_IMAGE_ONLY) && p->bs == 8 &&
        !(flags & (IBH_FORCE_GENERAL | IBH_EXACT | IBH_NO_FUSE | IBH_PASS_A_ONLY | IBH_PASS_B_ONLY | IBH_PHASE_INTERIOR |
                   IBH_PHASE_BOUNDARY))) {
        // image blocks of a partition with skirt fragments: one launch, nothing through the workspace
        const int32_t nwg = (p->n_img3 + WPB3E - 1) / WPB3E;
        hipLaunchKernelGGL((k_sweep3_euler_cols<2, false, true, false>), dim3(nwg), dim3(64 * WPB3E), 0, ibh_stream, P,
                           (uint32_t)ldp, R, (uint32_t)ldr, fluid->R, fluid->gamma, p->iblocks3, p->ihtab3, p->iftab3,
                           p->irtab3, p->ir4tab3, p->n_img3, nwg, p->idtab3);
        IBH_LAUNCH_CHECK();
        return 0;
    }
    if (p->nd == 3 && p->sweep3 && p->bs == 8 && p->blocks3 && p->nblk > 0 &&
        !(flags & (IBH_FORCE_GENERAL | IBH_EXACT | IBH_NO_FUSE | IBH_PASS_A_ONLY | IBH_PASS_B_ONLY | IBH_IMAGE_ONLY |
                   IBH_PHASE_INTERIOR | IBH_PHASE_BOUNDARY))) {
        // 3-D, every block qualifies for the single-kernel sweep: one launch, nothing through the workspace
        if (ibh_quad_variant != 512) {
            const bool persist = ibh_quad_variant == 514;  // A/B: persistent waves (measured slower, see above)
            const int32_t nwg = persist ? s3e_persistent_wgs(p->nblk) : (p->nblk + WPB3E - 1) / WPB3E;
#define S3E_LAUNCH(W, ST, PE)                                                                                         \
    hipLaunchKernelGGL((k_sweep3_euler_cols<W, ST, false, PE>), dim3(nwg), dim3(64 * WPB3E), 0, ibh_stream, P,         \
                       (uint32_t)ldp, R, (uint32_t)ldr, fluid->R, fluid->gamma, p->blocks3, p->htab3, p->ftab3,        \
                       p->rtab3, p->r4tab3, p->nblk, nwg)
            if (persist) S3E_LAUNCH(2, false, true);
            else if (ibh_quad_variant == 4) S3E_LAUNCH(2, true, false);  // wave time stamps (scripts/wave_timeline_3d.py)
            else S3E_LAUNCH(2, false, false);
#undef S3E_LAUNCH
        } else  // A/B: thread-per-cell form
        hipLaunchKernelGGL(k_sweep3_euler, dim3(p->nblk), dim3(512), 0, ibh_stream, P, (uint32_t)ldp, R, (uint32_t)ldr,
                           fluid->R, fluid->gamma, p->blocks3, p->htab3, p->ftab3, p->rtab3, p->r4tab3, p->nblk);
        IBH_LAUNCH_CHECK();
        return 0;
    }
    // the forms below run the whole sweep: they have no overlap phases (the single-kernel branch above does)
    IBH_REQUIRE(!(flags & (IBH_PHASE_INTERIOR | IBH_PHASE_BOUNDARY)),
                "ibh_residual_euler_hll: overlap phases need a partition whose (image) blocks are all eligible for the "
                "single-kernel sweep; run the sweep unphased after the exchange");
    int rc = ensure_G(p);
    if (rc) return rc;
    PartView v = view(p);
    dim3 blk(64 * WPB);
    // tuned block path: 2-D only and not with IBH_EXACT (the literal arithmetic lives in the face-list body)
    const bool fast = p->bs == 8 && p->nd == 2 && p->nblk > 0 && !(flags & (IBH_FORCE_GENERAL | IBH_EXACT));
    const int32_t nwg_fast = fast ? (p->nblk + WPB - 1) / WPB : 0;
    const int32_t* cellsA = fast ? p->irr_cells : nullptr;
    const int32_t nA = fast ? p->n_irr : p->nc;
    const int32_t* cellsB = cellsA;
    int32_t nB = nA;
    if ((flags & IBH_IMAGE_ONLY) && !fast) {
        cellsB = p->image_in_domain;
        nB = p->n_image;
    }
    dim3 gA(nwg_fast + (nA + 64 * WPB - 1) / (64 * WPB)), gB((nB + 64 * WPB - 1) / (64 * WPB));
    const bool doA = gA.x && !(flags & IBH_PASS_B_ONLY), doB = !(flags & IBH_PASS_A_ONLY);
    if (p->nd == 2) {
        if (doA && fast)
            hipLaunchKernelGGL((k_passA<2, 4, false>), gA, blk, 0, ibh_stream, v, P, ldp, p->G, p->blocks2, p->htab,
                               p->nblk, nwg_fast, cellsA, nA, flat_of(p, cellsA), nullptr);
        else if (doA)
            hipLaunchKernelGGL((k_passA<2, 4, true>), gA, blk, 0, ibh_stream, v, P, ldp, p->G, p->blocks2, p->htab,
                               p->nblk, 0, cellsA, nA, flat_of(p, cellsA), nullptr);
        if (doB && nwg_fast)
            hipLaunchKernelGGL(k_passB_euler_blk, dim3(nwg_fast), blk, 0, ibh_stream, (uint32_t)p->nc, P, (uint32_t)ldp,
                               p->G, R, (uint32_t)ldr, fluid->R, fluid->gamma, p->blocks2, p->htab, p->nblk, nwg_fast);
        if (doB && gB.x)
            hipLaunchKernelGGL((k_passB_euler<2>), gB, blk, 0, ibh_stream, v, P, ldp, p->G, R, ldr, fluid->R,
                               fluid->gamma, cellsB, nB);
    } else if (p->bs == 8 && p->blocks3 && p->nblk > 0 && !(flags & (IBH_FORCE_GENERAL | IBH_EXACT))) {
        // 3-D block path: block kernels + the face-list kernels over the cells the analysis left out
        const int32_t nI = p->n_irr;
        const int32_t gI = (nI + 511) / 512;
        if (!(flags & IBH_PASS_B_ONLY) && ibh_3d_wave) {
            const int32_t nwgA = (p->nblk + 3) / 4, gIw = (nI + 255) / 256;
            hipLaunchKernelGGL(k_passA3e_wave, dim3(nwgA + gIw), dim3(256), 0, ibh_stream, v, P, ldp, p->G, p->blocks3,
                               p->htab3, p->ftab3, p->nblk, nwgA, p->irr_cells, nI, flat_of(p, p->irr_cells));
        } else if (!(flags & IBH_PASS_B_ONLY))
            hipLaunchKernelGGL(k_passA3e_blk, dim3(p->nblk + gI), dim3(512), 0, ibh_stream, v, P, ldp, p->G, p->blocks3,
                               p->htab3, p->ftab3, p->nblk, p->irr_cells, nI, flat_of(p, p->irr_cells));
        if (!(flags & IBH_PASS_A_ONLY)) {
            hipLaunchKernelGGL(k_passB3e_blk, dim3(p->nblk), dim3(512), 0, ibh_stream, (uint32_t)p->nc, P, (uint32_t)ldp,
                               p->G, R, (uint32_t)ldr, fluid->R, fluid->gamma, p->blocks3, p->htab3, p->ftab3, p->nblk);
            if (nI)
                hipLaunchKernelGGL((k_passB_euler<3>), dim3((nI + 64 * WPB - 1) / (64 * WPB)), blk, 0, ibh_stream, v, P,
                                   ldp, p->G, R, ldr, fluid->R, fluid->gamma, p->irr_cells, nI);
        }
    } else {
        if (doA)
            hipLaunchKernelGGL((k_passA<3, 5, true>), gA, blk, 0, ibh_stream, v, P, ldp, p->G, p->blocks2, p->htab,
                               p->nblk, 0, cellsA, nA, flat_of(p, cellsA), nullptr);
        if (doB && gB.x)
            hipLaunchKernelGGL((k_passB_euler<3>), gB, blk, 0, ibh_stream, v, P, ldp, p->G, R, ldr, fluid->R,
                               fluid->gamma, cellsB, nB);
    }
    IBH_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
