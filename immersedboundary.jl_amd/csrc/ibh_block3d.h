// Block fast path, 3-D, 8x8x8 blocks: one 512-thread workgroup (8 wavefronts) per block, thread = cell
// (x fastest: tid = i + 8j + 64k).  Same scheme as the 2-D path (ibh_block2d.h): branch-free, per-block
// halo-cell table, reciprocal arithmetic, every face flux computed once (the +x/+y/+z faces by their owner
// thread, the low-side block faces by 192 otherwise idle threads) and exchanged through LDS.
// Sides: SAME / MIRROR / 2:1 COARSE; sides facing finer blocks are GENERAL (face-list body), so every
// boundary cell has exactly one face.  LDS field layout: [tile 512 | halo 384], halo slot = side*64 + t.
#pragma once
#include "ibh_block2d.h"
#include "ibh_sweep2d.h"

namespace blk3 {

#pragma clang fp contract(fast)

using blk2::adv_flux;
using blk2::flux_w;
using blk2::ldg;
using blk2::stg;

struct Lane3 {
    int i, j, k;
    bool edge[6];
    bool general;
    int nidx[6];  // index of the neighbour across direction s in a [tile | halo] field
    float q[6];
};

__device__ __forceinline__ Lane3 lane_info(const BlockDesc3& b, int tid) {
    Lane3 L;
    L.i = tid & 7;
    L.j = (tid >> 3) & 7;
    L.k = tid >> 6;
    L.edge[0] = L.i == 0;
    L.edge[1] = L.i == 7;
    L.edge[2] = L.j == 0;
    L.edge[3] = L.j == 7;
    L.edge[4] = L.k == 0;
    L.edge[5] = L.k == 7;
    const int t[3] = {L.j + 8 * L.k, L.i + 8 * L.k, L.i + 8 * L.j};
    const int off[6] = {-1, 1, -8, 8, -64, 64};
    L.general = false;
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        L.general |= L.edge[s] && b.type[s] == SIDE_GENERAL;
        L.nidx[s] = L.edge[s] ? 512 + s * 64 + t[s >> 1] : tid + off[s];
        L.q[s] = L.edge[s] ? b.q[s] : 0.5f;
    }
    return L;
}

// Halo cell of slot `tid` (side = tid >> 6 = this wavefront's index, boundary cell t = tid & 63).  The 3-D kernels are
// latency-bound (load -> LDS -> barrier chains of a 512-thread workgroup), so the id is computed from the
// descriptor instead of read from the table -- one dependent memory trip less: SAME / COARSE sides are pure
// arithmetic on the neighbour block's base (ibh_analyze3.cpp step 5), MIRROR / GENERAL name the boundary cell
// itself; only sides facing finer blocks read the table (wave-uniform branch).
__device__ __forceinline__ uint32_t halo_cell3(const BlockDesc3& bb, const int32_t* __restrict__ htab, int32_t blk,
                                               int tid) {
    const int sw = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (sw >= 6) return (uint32_t)bb.base;
    int ty, nb, sub;
    switch (sw) {  // scalar registers, no dynamic indexing
        case 0: ty = bb.type[0]; nb = bb.nb[0]; sub = bb.sub[0]; break;
        case 1: ty = bb.type[1]; nb = bb.nb[1]; sub = bb.sub[1]; break;
        case 2: ty = bb.type[2]; nb = bb.nb[2]; sub = bb.sub[2]; break;
        case 3: ty = bb.type[3]; nb = bb.nb[3]; sub = bb.sub[3]; break;
        case 4: ty = bb.type[4]; nb = bb.nb[4]; sub = bb.sub[4]; break;
        default: ty = bb.type[5]; nb = bb.nb[5]; sub = bb.sub[5]; break;
    }
    if (ty == SIDE_FINE) return (uint32_t)htab[(size_t)blk * 384 + tid];
    const int d = sw >> 1;
    const bool low = (sw & 1) == 0;
    const int sd = d == 0 ? 1 : d == 1 ? 8 : 64;   // stride of the normal dim
    const int sa = d == 0 ? 8 : 1;                 // strides of the two tangential dims (increasing order)
    const int sb = d == 2 ? 8 : 64;
    const int t = tid & 63;
    int t1 = t & 7, t2 = t >> 3;
    int n, base;
    if (ty == SIDE_SAME) {
        n = low ? 7 : 0;
        base = nb;
    } else if (ty == SIDE_COARSE) {
        n = low ? 7 : 0;
        base = nb;
        t1 = 4 * (sub & 1) + (t1 >> 1);
        t2 = 4 * (sub >> 1) + (t2 >> 1);
    } else {  // MIRROR, GENERAL: the boundary cell itself
        n = low ? 0 : 7;
        base = bb.base;
    }
    return (uint32_t)(base + n * sd + t1 * sa + t2 * sb);
}

// ------------------------------------------------------------------------------------------
// pass A (scalar field): gradients along x, y, z + JST sensor.  LDS: 896 floats.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void passA(const BlockDesc3* __restrict__ blocks, const int32_t* __restrict__ htab,
                                      const int32_t* __restrict__ ftab, int32_t blk, uint32_t nc,
                                      const float* __restrict__ u, float* __restrict__ G, float* lds, int tid) {
    const BlockDesc3 bb = blocks[blk];
    const uint32_t c = (uint32_t)bb.base + tid;
    const uint32_t hidx = halo_cell3(bb, htab, blk, tid);
    const float uc = ldg(u, c);
    const float hv = ldg(u, hidx);
    lds[tid] = uc;
    if (tid < 384) lds[512 + tid] = hv;
    const Lane3 L = lane_info(bb, tid);
    __syncthreads();
    float g[3], D = 1e-7f;
    const int tt[3] = {L.j + 8 * L.k, L.i + 8 * L.k, L.i + 8 * L.j};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float rh = bb.rh[d];
        float vm[2], am[2];  // mean neighbour value and mean |difference| towards the low / high side
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const int s = 2 * d + side;
            const float v0 = lds[L.nidx[s]];
            vm[side] = v0;
            am[side] = fabsf(v0 - uc);
            if (bb.type[s] == SIDE_FINE) {  // wave-uniform: 4 fine cells behind every boundary cell of this side
                if (L.edge[s]) {
                    const int32_t* ft = ftab + (((size_t)bb.fine * 6 + s) * 64 + tt[d]) * 3;
                    const float v1 = ldg(u, (uint32_t)ft[0]), v2 = ldg(u, (uint32_t)ft[1]), v3 = ldg(u, (uint32_t)ft[2]);
                    vm[side] = 0.25f * (v0 + v1 + v2 + v3);
                    am[side] = 0.25f * (fabsf(v0 - uc) + fabsf(v1 - uc) + fabsf(v2 - uc) + fabsf(v3 - uc));
                }
            }
        }
        const float fr = uc + L.q[2 * d + 1] * (vm[1] - uc);  // at_faces: (1-q)*u_self + q*u_nb
        const float fl = uc + L.q[2 * d] * (vm[0] - uc);
        g[d] = (fr - fl) * rh;
        const float dr = vm[1] - uc, dl = uc - vm[0];
        const float gg = (dr - dl) * rh;
        const float ugg = (am[1] + am[0]) * rh;
        D = fmaxf(D, (1e-7f + fabsf(gg)) * __builtin_amdgcn_rcpf(1e-7f + ugg));
    }
    if (!L.general) {
        stg(G, c, g[0]);
        stg(G + nc, c, g[1]);
        stg(G + (size_t)2 * nc, c, g[2]);
        stg(G + (size_t)3 * nc, c, D);
    }
}

// ------------------------------------------------------------------------------------------
// pass B, advection.  LDS (floats): fU 896 | fD 896 | tG 3x512 | hG 384 | tC 3x512 | hC 384 | ex 192 | F 3x512
// ------------------------------------------------------------------------------------------
#define BLK3_PASSB_LDS (896 * 2 + 1536 + 384 + 1536 + 384 + 192 + 1536)

__device__ __forceinline__ void passB_adv(const BlockDesc3* __restrict__ blocks, const int32_t* __restrict__ htab,
                                          const int32_t* __restrict__ ftab, int32_t blk, uint32_t nc,
                                          const float* __restrict__ u,
                                          const float* __restrict__ C, uint32_t ldc, const float* __restrict__ G,
                                          float* __restrict__ ud, float* lds, int tid) {
    const BlockDesc3 bb = blocks[blk];
    const uint32_t c = (uint32_t)bb.base + tid;
    float* fU = lds;
    float* fD = lds + 896;
    float* tG = lds + 1792;          // [3][512]
    float* hG = lds + 1792 + 1536;   // [384]
    float* tC = hG + 384;            // [3][512]
    float* hC = tC + 1536;           // [384]
    float* ex = hC + 384;            // [192]
    float* FF = ex + 192;            // [3][512]
    const float* Gs = G + (size_t)3 * nc;
    const uint32_t hidx = halo_cell3(bb, htab, blk, tid);
    const float uc = ldg(u, c), Dc = ldg(Gs, c);
    float gc[3], cc[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        gc[d] = ldg(G + (size_t)d * nc, c);
        cc[d] = ldg(C + (size_t)d * ldc, c);
    }
    __builtin_amdgcn_sched_barrier(0);
    const int dnh = tid >> 7;  // normal dim of this thread's halo slot (side = tid >> 6)
    const float hu = ldg(u, hidx), hD = ldg(Gs, hidx);
    const float hg = ldg(G + (size_t)(dnh < 3 ? dnh : 0) * nc, hidx);
    const float hc = ldg(C + (size_t)(dnh < 3 ? dnh : 0) * ldc, hidx);
    fU[tid] = uc;
    fD[tid] = Dc;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        tG[d * 512 + tid] = gc[d];
        tC[d * 512 + tid] = cc[d];
    }
    if (tid < 384) {
        fU[512 + tid] = hu;
        fD[512 + tid] = hD;
        hG[tid] = hg;
        hC[tid] = hc;
    }
    const Lane3 L = lane_info(bb, tid);
    __syncthreads();

    float hnb[6];  // width of the neighbour across each side (wave-uniform)
#pragma unroll
    for (int s = 0; s < 6; ++s) hnb[s] = bb.h[s >> 1] * bb.rt[s];
    // ---- main pass: the +x, +y, +z face of every cell
    const int tth[3] = {L.j + 8 * L.k, L.i + 8 * L.k, L.i + 8 * L.j};
    float F[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int s = 2 * d + 1;
        const bool e = L.edge[s];
        const int ni = L.nidx[s];
        const float gb = e ? hG[ni - 512] : tG[d * 512 + ni];
        const float Cb = e ? hC[ni - 512] : tC[d * 512 + ni];
        // undivided slopes S = g*h and the weight wa = h_a/(h_a+h_b) = q: the 19-operation flux of ibh_sweep2d.h
        const float Sa = gc[d] * bb.h[d];
        const float hb = e ? hnb[s] : bb.h[d];
        F[d] = flux_w(uc, fU[ni], Sa, gb * hb, Dc, fD[ni], cc[d], Cb, L.q[s]);
        if (bb.type[s] == SIDE_FINE) {  // wave-uniform: 3 more sub-faces, straight from global memory
            if (e) {
                const int32_t* ft = ftab + (((size_t)bb.fine * 6 + s) * 64 + tth[d]) * 3;
                float acc = F[d];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const uint32_t cl = (uint32_t)ft[k];
                    acc += flux_w(uc, ldg(u, cl), Sa, ldg(G + (size_t)d * nc, cl) * hnb[s], Dc, ldg(Gs, cl), cc[d],
                                  ldg(C + (size_t)d * ldc, cl), bb.q[s]);
                }
                F[d] = 0.25f * acc;
            }
        }
    }
    // ---- extra pass: the 3 x 64 low-side block faces (halo cell = owner), threads 0..191
    {
        const int r = tid < 192 ? tid : 0;
        const int dn = r >> 6, t = r & 63;
        const int t1 = t & 7, t2 = t >> 3;
        const int pos = dn == 0 ? 8 * t1 + 64 * t2 : dn == 1 ? t1 + 64 * t2 : t;
        const int slot = dn * 128 + t;  // side 2*dn
        const float hme = dn == 0 ? bb.h[0] : dn == 1 ? bb.h[1] : bb.h[2];
        const float hha = dn == 0 ? hnb[0] : dn == 1 ? hnb[2] : hnb[4];          // width of the halo (owner) cell
        const float wha = 1.0f - (dn == 0 ? bb.q[0] : dn == 1 ? bb.q[2] : bb.q[4]);  // h_halo / (h_halo + h)
        const float Sme = tG[dn * 512 + pos] * hme;
        float X = flux_w(fU[512 + slot], fU[pos], hG[slot] * hha, Sme, fD[512 + slot], fD[pos], hC[slot],
                         tC[dn * 512 + pos], wha);
        const int tyl = dn == 0 ? bb.type[0] : dn == 1 ? bb.type[2] : bb.type[4];  // uniform per 64-thread role group
        if (tyl == SIDE_FINE && tid < 192) {
            const int32_t* ft = ftab + (((size_t)bb.fine * 6 + 2 * dn) * 64 + t) * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint32_t cl = (uint32_t)ft[k];
                X += flux_w(ldg(u, cl), fU[pos], ldg(G + (size_t)dn * nc, cl) * hha, Sme, ldg(Gs, cl), fD[pos],
                            ldg(C + (size_t)dn * ldc, cl), tC[dn * 512 + pos], wha);
            }
            X *= 0.25f;
        }
        if (tid < 192) ex[tid] = X;
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) FF[d * 512 + tid] = F[d];
    __syncthreads();
    // low-side fluxes: from the extra pass on block sides, from the neighbour thread inside the block
    // (ex and FF are contiguous: ex at FF - 192)
    const int tt[3] = {L.j + 8 * L.k, L.i + 8 * L.k, L.i + 8 * L.j};
    const int offm[3] = {1, 8, 64};
    float res = 0.0f;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int idx = L.edge[2 * d] ? (d * 64 + tt[d]) - 192 : d * 512 + tid - offm[d];
        const float Flow = FF[idx];
        res = res - (F[d] - Flow) * bb.rh[d];
    }
    if (!L.general) stg(ud, c, res);
}

// ------------------------------------------------------------------------------------------
// Single-kernel scalar sweep in 3-D (the closure of test/advection.jl:67-83 on an octree partition): the scheme of
// blk2::sweep_adv / quad2.  The workgroup of a block computes the undivided slopes and the JST sensor of its own cells
// AND of its halo cells (slope along the side normal only), so nothing goes through the gradient workspace: 20 B per
// cell moved instead of 60.
//   * wavefront s (0..5) owns side s: its lane t holds the halo cell(s) behind boundary cell t -- one (SAME / COARSE /
//     MIRROR) or the 2 x 2 finer cells of a FINE side ("chunks" k = 0..3) -- computes their slope and sensor, then the
//     flux of the sub-face(s) and hands the boundary cell their mean (ex).  The cells' own threads compute the three
//     high faces inside the block only.
//   * a halo cell needs the block's boundary cells, the cell one step deeper (same neighbour block: index arithmetic)
//     and its four lateral neighbours: other halo cells of the same side, staged in the side's plane
//     ((n + 2)^2, n = 8 or 16 halo cells along the side), whose border comes from the rim table (ibh_analyze3.cpp).
// LDS (floats): fU 896 | fD 512 | tS 3x512 | tC 3x512 | ex 6x64 | FF 3x512 | planes 6 x 18x18 | planeA 6x64
// ------------------------------------------------------------------------------------------
#define BLK3_SWEEP_LDS (896 + 512 + 1536 + 1536 + 384 + 1536 + 6 * 324 + 384)

__device__ __forceinline__ float jst_ratio(float g, float a, float rh) {
    return fmaf(fabsf(g), rh, 1e-7f) * __builtin_amdgcn_rcpf(fmaf(a, rh, 1e-7f));
}

__device__ __forceinline__ void sweep_adv(const BlockDesc3* __restrict__ blocks, const int32_t* __restrict__ htab,
                                          const int32_t* __restrict__ ftab, const int32_t* __restrict__ rtab,
                                          const int32_t* __restrict__ r4tab, int32_t blk, const float* __restrict__ u,
                                          const float* __restrict__ C, uint32_t ldc, float* __restrict__ ud, float* lds,
                                          int tid) {
    const BlockDesc3 bb = blocks[blk];
    const uint32_t c = (uint32_t)bb.base + tid;
    float* fU = lds;                  // [tile 512 | halo 384]: halo part for SAME / COARSE / MIRROR sides
    float* fD = fU + 896;             // [512]
    float* tS = fD + 512;             // [3][512] undivided slopes of the block's cells
    float* tC = tS + 1536;            // [3][512]
    float* ex = tC + 1536;            // [6][64]  mean flux through the sub-face(s) of every boundary cell
    float* FF = ex + 384;             // [3][512]
    float* plane = FF + 1536;         // [6][18 x 18]
    float* planeA = plane + 6 * 324;  // [6][64]  rim neighbours: mean |difference| to the halo cell next to them
    // ---- this thread's halo slot: side = wavefront index, boundary cell t
    const int sw = __builtin_amdgcn_readfirstlane(tid >> 6);
    int ty = SIDE_MIRROR;
    float qs = 0.5f;
    switch (sw) {  // scalar registers, no dynamic indexing
        case 0: ty = bb.type[0]; qs = bb.q[0]; break;
        case 1: ty = bb.type[1]; qs = bb.q[1]; break;
        case 2: ty = bb.type[2]; qs = bb.q[2]; break;
        case 3: ty = bb.type[3]; qs = bb.q[3]; break;
        case 4: ty = bb.type[4]; qs = bb.q[4]; break;
        case 5: ty = bb.type[5]; qs = bb.q[5]; break;
        default: break;
    }
    const int dnh = sw >> 1;                         // normal dim of the slot's side (3: the two idle wavefronts)
    const bool low = (sw & 1) == 0, slotw = sw < 6;
    const bool isC = ty == SIDE_COARSE, isF = ty == SIDE_FINE, mirror = ty == SIDE_MIRROR;
    const int sd = dnh == 0 ? 1 : dnh == 1 ? 8 : 64, sa = dnh == 0 ? 8 : 1, sb = dnh == 2 ? 8 : 64;
    const int t = tid & 63, t1 = t & 7, t2 = t >> 3;
    const int n = isF ? 16 : 8;                      // halo cells along the side
    const int dd = mirror ? 0 : (low ? -sd : sd);    // deeper cell of a halo cell
    const float* Cn = C + (size_t)(dnh < 3 ? dnh : 0) * ldc;
    // ---- loads: own cell; halo chunk(s): value, deeper value, normal velocity; rim cells
    const float uc = ldg(u, c);
    float cc[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) cc[d] = ldg(C + (size_t)d * ldc, c);
    float hu[4], hde[4], hc[4];
    {
        const uint32_t h0 = halo_cell3(bb, htab, blk, tid);
        hu[0] = ldg(u, h0);
        hde[0] = ldg(u, (uint32_t)((int)h0 + dd));
        hc[0] = ldg(Cn, h0);
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            hu[k] = hu[0];
            hde[k] = hde[0];
            hc[k] = hc[0];
        }
        if (isF) {  // wave-uniform
            const int32_t* ft = ftab + (((size_t)bb.fine * 6 + sw) * 64 + t) * 3;
#pragma unroll
            for (int k = 1; k < 4; ++k) {
                const uint32_t hk = (uint32_t)ft[k - 1];
                hu[k] = ldg(u, hk);
                hde[k] = ldg(u, (uint32_t)((int)hk + dd));
                hc[k] = ldg(Cn, hk);
            }
        }
    }
    float rv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const bool rimw = slotw && t < 4 * n;
    if (rimw) {
        const int32_t rid = rtab[sw * 64 + t];
        if (rid >= 0) {
            rv[0] = rv[1] = rv[2] = rv[3] = ldg(u, (uint32_t)rid);
        } else {
            const int32_t* r4 = r4tab + (size_t)(-rid - 1) * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) rv[k] = ldg(u, (uint32_t)r4[k]);
        }
    }
    // ---- stage
    fU[tid] = uc;
#pragma unroll
    for (int d = 0; d < 3; ++d) tC[d * 512 + tid] = cc[d];
    float* pl = plane + (slotw ? sw : 0) * 324;
    int radj = 0;  // plane position of the halo cell next to this lane's rim cell
    if (slotw) {
        fU[512 + tid] = hu[0];
        if (isF) {
#pragma unroll
            for (int k = 0; k < 4; ++k) pl[(2 * t1 + (k & 1) + 1) + 18 * (2 * t2 + (k >> 1) + 1)] = hu[k];
        } else {
            pl[(t1 + 1) + 18 * (t2 + 1)] = hu[0];
        }
        if (rimw) {
            const int r = isF ? t >> 4 : t >> 3, i = isF ? t & 15 : t & 7;
            const int p1 = r == 0 ? 0 : r == 1 ? n + 1 : i + 1, p2 = r == 2 ? 0 : r == 3 ? n + 1 : i + 1;
            const int a1 = r == 0 ? 1 : r == 1 ? n : i + 1, a2 = r == 2 ? 1 : r == 3 ? n : i + 1;
            pl[p1 + 18 * p2] = 0.25f * ((rv[0] + rv[1]) + (rv[2] + rv[3]));
            radj = a1 + 18 * a2;
        }
    }
    const Lane3 L = lane_info(bb, tid);
    __syncthreads();
    if (rimw) {
        const float ha = pl[radj];
        planeA[sw * 64 + t] = 0.25f * ((fabsf(rv[0] - ha) + fabsf(rv[1] - ha)) + (fabsf(rv[2] - ha) + fabsf(rv[3] - ha)));
    }
    // ---- own cells: undivided slopes S = fr - fl and the sensor
    float S[3], D = 1e-7f;
    const int tt[3] = {L.j + 8 * L.k, L.i + 8 * L.k, L.i + 8 * L.j};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float vm[2], am[2];
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const int s = 2 * d + side;
            const float v0 = fU[L.nidx[s]];
            vm[side] = v0;
            am[side] = fabsf(v0 - uc);
            if (bb.type[s] == SIDE_FINE) {  // workgroup-uniform: four finer cells behind every boundary cell
                if (L.edge[s]) {
                    const float* ps = plane + s * 324 + (2 * (tt[d] & 7) + 1) + 18 * (2 * (tt[d] >> 3) + 1);
                    const float w0 = ps[0], w1 = ps[1], w2 = ps[18], w3 = ps[19];
                    vm[side] = 0.25f * ((w0 + w1) + (w2 + w3));
                    am[side] = 0.25f * ((fabsf(w0 - uc) + fabsf(w1 - uc)) + (fabsf(w2 - uc) + fabsf(w3 - uc)));
                }
            }
        }
        const float dr = vm[1] - uc, dl = uc - vm[0];
        S[d] = L.q[2 * d + 1] * dr + L.q[2 * d] * dl;
        D = fmaxf(D, jst_ratio(dr - dl, am[1] + am[0], bb.rh[d]));
        tS[d * 512 + tid] = S[d];
    }
    fD[tid] = D;
    // ---- halo cells of this thread's slot: slope along the normal, sensor
    float Sh[4], Dh[4];
    const int nrm = (low ? 0 : 7) * sd;
    const int pos = nrm + t1 * sa + t2 * sb;  // the slot's boundary cell
    if (slotw) {
        blk2::wave_lds_sync();  // planeA of this side (written by this wavefront)
        // the block's cells behind the halo cell: one, or the 2 x 2 group in front of a coarse cell
        const int cm = isC ? 1 : 0;
        const int pa = t1 & ~cm, pb = t1 | cm, qa = t2 & ~cm, qb = t2 | cm;
        const float m0 = fU[nrm + pa * sa + qa * sb], m1 = fU[nrm + pb * sa + qa * sb];
        const float m2 = fU[nrm + pa * sa + qb * sb], m3 = fU[nrm + pb * sa + qb * sb];
        const float irt = isC ? 0.5f : isF ? 2.0f : 1.0f;  // h / h_halo
        float rn, ra, rb;  // reciprocal widths of the halo cell along the normal and the two tangential dims
        if (dnh == 0) { rn = bb.rh[0]; ra = bb.rh[1]; rb = bb.rh[2]; }
        else if (dnh == 1) { rn = bb.rh[1]; ra = bb.rh[0]; rb = bb.rh[2]; }
        else { rn = bb.rh[2]; ra = bb.rh[0]; rb = bb.rh[1]; }
        rn *= irt;
        ra *= irt;
        rb *= irt;
        const float* pA = planeA + sw * 64;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            Sh[k] = 0.0f;
            Dh[k] = 1e-7f;
            if (k == 0 || isF) {  // wave-uniform
                const float h = hu[k];
                const float din = 0.25f * ((m0 + m1) + (m2 + m3)) - h;
                const float ain = 0.25f * ((fabsf(m0 - h) + fabsf(m1 - h)) + (fabsf(m2 - h) + fabsf(m3 - h)));
                const float dde = hde[k] - h;
                const float x = (1.0f - qs) * din - 0.5f * dde;
                Sh[k] = low ? x : -x;
                // position in the plane; a coarse cell spans 2 x 2 slots
                const int f1 = isF ? 2 * t1 + (k & 1) : t1, f2 = isF ? 2 * t2 + (k >> 1) : t2;
                const int fa = f1 & ~cm, fb = f1 | cm, ga = f2 & ~cm, gb = f2 | cm;
                const float ea0 = pl[fa + 18 * (f2 + 1)] - h, ea1 = pl[fb + 2 + 18 * (f2 + 1)] - h;
                const float eb0 = pl[(f1 + 1) + 18 * ga] - h, eb1 = pl[(f1 + 1) + 18 * (gb + 2)] - h;
                const float aa0 = fa == 0 ? pA[f2] : fabsf(ea0), aa1 = fb == n - 1 ? pA[n + f2] : fabsf(ea1);
                const float ab0 = ga == 0 ? pA[2 * n + f1] : fabsf(eb0), ab1 = gb == n - 1 ? pA[3 * n + f1] : fabsf(eb1);
                float dh = jst_ratio(din + dde, ain + fabsf(dde), rn);
                dh = fmaxf(dh, jst_ratio(ea0 + ea1, aa0 + aa1, ra));
                dh = fmaxf(dh, jst_ratio(eb0 + eb1, ab0 + ab1, rb));
                Dh[k] = fmaxf(dh, 1e-7f);
            }
        }
    }
    __syncthreads();
    // ---- the +x, +y, +z face of every cell inside the block (block faces: from ex below)
    float F[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int ni = L.edge[2 * d + 1] ? tid : L.nidx[2 * d + 1];
        F[d] = flux_w(uc, fU[ni], S[d], tS[d * 512 + ni], D, fD[ni], cc[d], tC[d * 512 + ni], 0.5f);
    }
    // ---- block faces: the sub-face(s) of this thread's slot, halo cell <-> boundary cell
    if (slotw) {
        const float ub = fU[pos], Sb = tS[dnh * 512 + pos], Db = fD[pos], Cb = tC[dnh * 512 + pos];
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k == 0 || isF) {  // wave-uniform
                const float sh = mirror ? Sb : Sh[k], dh = mirror ? Db : Dh[k];
                const float ua = low ? hu[k] : ub, ubb = low ? ub : hu[k];
                const float Sa = low ? sh : Sb, Sbb = low ? Sb : sh;
                const float Da = low ? dh : Db, Dbb = low ? Db : dh;
                const float Ca = low ? hc[k] : Cb, Cbb = low ? Cb : hc[k];
                acc += flux_w(ua, ubb, Sa, Sbb, Da, Dbb, Ca, Cbb, low ? 1.0f - qs : qs);
            }
        }
        ex[tid] = isF ? 0.25f * acc : acc;
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) FF[d * 512 + tid] = F[d];
    __syncthreads();
    const int offm[3] = {1, 8, 64};
    float res = 0.0f;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float Fl = L.edge[2 * d] ? ex[2 * d * 64 + tt[d]] : FF[d * 512 + tid - offm[d]];
        const float Fh = L.edge[2 * d + 1] ? ex[(2 * d + 1) * 64 + tt[d]] : F[d];
        res = res - (Fh - Fl) * bb.rh[d];
    }
    stg(ud, c, res);
}

// ------------------------------------------------------------------------------------------
// Euler sweep in 3-D (P = [p T u v w], cfd.jl:106-151 / :459-508), the block form of the R2 residual.
// pass A: gradients of the NV primitives along x, y, z + JST sensor of the pressure.
//   G layout as the face-list kernels: grad of var v along dim d at G[(d*NV + v)*nc + c], sensor at G[3*NV*nc + c].
//   LDS: NV x 896 floats.
// ------------------------------------------------------------------------------------------
template <int NV>
__device__ __forceinline__ void passA_nv(const BlockDesc3* __restrict__ blocks, const int32_t* __restrict__ htab,
                                         const int32_t* __restrict__ ftab, int32_t blk, uint32_t nc,
                                         const float* __restrict__ P, uint32_t ldp, float* __restrict__ G, float* lds,
                                         int tid) {
    const BlockDesc3 bb = blocks[blk];
    const uint32_t c = (uint32_t)bb.base + tid;
    const uint32_t hidx = halo_cell3(bb, htab, blk, tid);
    float self[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        self[v] = ldg(P + (size_t)v * ldp, c);
        const float hv = ldg(P + (size_t)v * ldp, hidx);
        lds[v * 896 + tid] = self[v];
        if (tid < 384) lds[v * 896 + 512 + tid] = hv;
    }
    const Lane3 L = lane_info(bb, tid);
    __syncthreads();
    const int tt[3] = {L.j + 8 * L.k, L.i + 8 * L.k, L.i + 8 * L.j};
    float D = 1e-7f;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float rh = bb.rh[d];
        // the three extra fine cells behind a boundary cell of a FINE side (workgroup-uniform branch)
        uint32_t fc[2][3];
        bool fine[2];
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const int s = 2 * d + side;
            fine[side] = false;
            if (bb.type[s] == SIDE_FINE) {
                if (L.edge[s]) {
                    const int32_t* ft = ftab + (((size_t)bb.fine * 6 + s) * 64 + tt[d]) * 3;
                    fc[side][0] = (uint32_t)ft[0];
                    fc[side][1] = (uint32_t)ft[1];
                    fc[side][2] = (uint32_t)ft[2];
                    fine[side] = true;
                }
            }
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float uc = self[v];
            float vm[2], am[2];
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                const int s = 2 * d + side;
                const float v0 = lds[v * 896 + L.nidx[s]];
                vm[side] = v0;
                am[side] = fabsf(v0 - uc);
                if (bb.type[s] == SIDE_FINE) {
                    if (fine[side]) {
                        const float* Pv = P + (size_t)v * ldp;
                        const float v1 = ldg(Pv, fc[side][0]), v2 = ldg(Pv, fc[side][1]), v3 = ldg(Pv, fc[side][2]);
                        vm[side] = 0.25f * (v0 + v1 + v2 + v3);
                        am[side] = 0.25f * (fabsf(v0 - uc) + fabsf(v1 - uc) + fabsf(v2 - uc) + fabsf(v3 - uc));
                    }
                }
            }
            const float fr = uc + L.q[2 * d + 1] * (vm[1] - uc);
            const float fl = uc + L.q[2 * d] * (vm[0] - uc);
            if (!L.general) stg(G + (size_t)(d * NV + v) * nc, c, (fr - fl) * rh);
            if (v == 0) {
                const float gg = ((vm[1] - uc) - (uc - vm[0])) * rh;
                const float ugg = (am[1] + am[0]) * rh;
                D = fmaxf(D, (1e-7f + fabsf(gg)) * __builtin_amdgcn_rcpf(1e-7f + ugg));
            }
        }
    }
    if (!L.general) stg(G + (size_t)(3 * NV) * nc, c, D);
}

struct Gas3 {
    float R, gamma;
};

__device__ __forceinline__ void euler_side3(const float* P, int dn, const Gas3& gas, float* Q, float* F, float& un,
                                            float& a) {
    const float p = P[0];
    const float T = fmaxf(P[1], 10.0f);
    const float k = 0.5f * (P[2] * P[2] + P[3] * P[3] + P[4] * P[4]);
    const float rho = p * __builtin_amdgcn_rcpf(gas.R * T);
    const float E = rho * (gas.R / (gas.gamma - 1.0f) * T + k);
    Q[0] = rho;
    Q[1] = E;
    Q[2] = rho * P[2];
    Q[3] = rho * P[3];
    Q[4] = rho * P[4];
    un = dn == 0 ? P[2] : dn == 1 ? P[3] : P[4];
    a = __builtin_amdgcn_sqrtf(gas.gamma * gas.R * T);
    F[0] = Q[0] * un;
    F[1] = (Q[1] + p) * un;
    F[2] = Q[2] * un + (dn == 0 ? p : 0.0f);
    F[3] = Q[3] * un + (dn == 1 ? p : 0.0f);
    F[4] = Q[4] * un + (dn == 2 ? p : 0.0f);
}

// a = owner (towards -), b = neighbour; ga / gb the gradients of the 5 primitives along the face normal
__device__ __forceinline__ void euler_flux3(const float* Pa, const float* Pb, const float* ga, const float* gb, float Da,
                                            float Db, float dA, float dB, float inv, int dn, const Gas3& gas, float* F) {
    float PL[5], PR[5];
    const float Df = fmaxf(fmaxf(Da, Db), 1e-7f);
#pragma unroll
    for (int v = 0; v < 5; ++v) {
        const float guf = (Pb[v] - Pa[v]) * inv;
        const float gu = (2.0f * ga[v] - guf) * dA;
        const float Du = (2.0f * gb[v] - guf) * dB;
        const float s = blk2::minmod(Du, gu);
        const float l = Pa[v] + s, r = Pb[v] - s;
        const float uf = (Pa[v] * dB + Pb[v] * dA) * inv + (ga[v] * dA - gb[v] * dB) * 0.125f;
        PL[v] = uf + Df * (l - uf);
        PR[v] = uf + Df * (r - uf);
    }
    float QL[5], FL[5], QR[5], FR[5], uL, aL, uR, aR;
    euler_side3(PL, dn, gas, QL, FL, uL, aL);
    euler_side3(PR, dn, gas, QR, FR, uR, aR);
    const float SR = fminf(uR - aR, 0.0f);
    const float SL = fmaxf(uL + aL, 0.0f);
    const float rs = __builtin_amdgcn_rcpf(SL - SR);
#pragma unroll
    for (int v = 0; v < 5; ++v) F[v] = (SL * FL[v] - SR * FR[v] + SR * SL * (QR[v] - QL[v])) * rs;
}

// pass B, Euler.  LDS (floats): fP 5x896 | fD 896 | 2 x (tG 5x512 | hG 5x128) | ex 5x192 | FF 5x512
//   fP, fD: [tile | halo of all six sides]; tG / hG: gradients along ONE dim (tile / halo of its two sides), double
//   buffered: the loads of the next dim are issued before the barrier that ends the current one.
#define BLK3_EULER_LDS (5 * 896 + 896 + 2 * (5 * 512 + 5 * 128) + 5 * 192 + 5 * 512)

__device__ __forceinline__ void passB_euler(const BlockDesc3* __restrict__ blocks, const int32_t* __restrict__ htab,
                                            const int32_t* __restrict__ ftab, int32_t blk, uint32_t nc,
                                            const float* __restrict__ P, uint32_t ldp, const float* __restrict__ G,
                                            float* __restrict__ Rr, uint32_t ldr, Gas3 gas, float* lds, int tid) {
    const BlockDesc3 bb = blocks[blk];
    const uint32_t c = (uint32_t)bb.base + tid;
    float* fP = lds;                 // [5][896]
    float* fD = lds + 5 * 896;       // [896]
    float* gbuf = fD + 896;          // 2 x ([5][512] tile | [5][128] halo)
    float* ex = gbuf + 2 * 3200;     // [3][5][64]
    float* FF = ex + 5 * 192;        // [5][512]
    const float* Gs = G + (size_t)15 * nc;
    const uint32_t hidx = halo_cell3(bb, htab, blk, tid);
    float Pc[5];
#pragma unroll
    for (int v = 0; v < 5; ++v) Pc[v] = ldg(P + (size_t)v * ldp, c);
    const float Dc = ldg(Gs, c);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int v = 0; v < 5; ++v) {
        const float hv = ldg(P + (size_t)v * ldp, hidx);
        fP[v * 896 + tid] = Pc[v];
        if (tid < 384) fP[v * 896 + 512 + tid] = hv;
    }
    {
        const float hD = ldg(Gs, hidx);
        fD[tid] = Dc;
        if (tid < 384) fD[512 + tid] = hD;
    }
    const Lane3 L = lane_info(bb, tid);
    float dBs[6], invs[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        dBs[s] = 0.5f * bb.h[s >> 1] * bb.rt[s];
        invs[s] = 2.0f * bb.rh[s >> 1] * bb.q[s];
    }
    const int tt[3] = {L.j + 8 * L.k, L.i + 8 * L.k, L.i + 8 * L.j};
    __syncthreads();
    // ---- the 3 x 64 low-side block faces (halo cell = owner) by the first three wavefronts, gradients of the
    //      two cells straight from the workspace
    if (tid < 192) {
        const int dn = tid >> 6, t = tid & 63;  // wave-uniform dn
        const int t1 = t & 7, t2 = t >> 3;
        const int pos = dn == 0 ? 8 * t1 + 64 * t2 : dn == 1 ? t1 + 64 * t2 : t;
        const int slot = 512 + dn * 128 + t;  // side 2*dn
        const uint32_t ch = halo_cell3(bb, htab, blk, dn * 128 + t), cp = (uint32_t)bb.base + pos;
        const float hd = 0.5f * (dn == 0 ? bb.h[0] : dn == 1 ? bb.h[1] : bb.h[2]);
        const float dB = dn == 0 ? dBs[0] : dn == 1 ? dBs[2] : dBs[4];
        const float inv = dn == 0 ? invs[0] : dn == 1 ? invs[2] : invs[4];
        const int tyl = dn == 0 ? bb.type[0] : dn == 1 ? bb.type[2] : bb.type[4];
        float Pa[5], Pb[5], ga[5], gb[5], X[5];
#pragma unroll
        for (int v = 0; v < 5; ++v) {
            Pa[v] = fP[v * 896 + slot];
            Pb[v] = fP[v * 896 + pos];
            ga[v] = ldg(G + (size_t)(dn * 5 + v) * nc, ch);
            gb[v] = ldg(G + (size_t)(dn * 5 + v) * nc, cp);
        }
        euler_flux3(Pa, Pb, ga, gb, fD[slot], fD[pos], dB, hd, inv, dn, gas, X);
        if (tyl == SIDE_FINE) {  // wave-uniform: three more fine owners behind this face
            const int32_t* ft = ftab + (((size_t)bb.fine * 6 + 2 * dn) * 64 + t) * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint32_t cl = (uint32_t)ft[k];
                float Y[5];
#pragma unroll
                for (int v = 0; v < 5; ++v) {
                    Pa[v] = ldg(P + (size_t)v * ldp, cl);
                    ga[v] = ldg(G + (size_t)(dn * 5 + v) * nc, cl);
                }
                euler_flux3(Pa, Pb, ga, gb, ldg(Gs, cl), fD[pos], dB, hd, inv, dn, gas, Y);
#pragma unroll
                for (int v = 0; v < 5; ++v) X[v] += Y[v];
            }
#pragma unroll
            for (int v = 0; v < 5; ++v) X[v] *= 0.25f;
        }
#pragma unroll
        for (int v = 0; v < 5; ++v) ex[(dn * 5 + v) * 64 + t] = X[v];
    }
    float res[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const int offm[3] = {1, 8, 64};
    // gradients along dim 0 into buffer 0: own cells, and the halo cells of sides 2d, 2d+1 (slots 128d .. 128d+127)
    float gc[5];
#pragma unroll
    for (int v = 0; v < 5; ++v) {
        gc[v] = ldg(G + (size_t)v * nc, c);
        gbuf[v * 512 + tid] = gc[v];
        if ((tid >> 7) == 0) gbuf[2560 + v * 128 + (tid & 127)] = ldg(G + (size_t)v * nc, hidx);
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float* tG = gbuf + (d & 1) * 3200;
        const float* hG = tG + 2560;
        __syncthreads();  // the staged gradients (and, first time, fP / fD / ex) are visible
        const int s = 2 * d + 1;
        const bool e = L.edge[s];
        const int ni = L.nidx[s];
        const float hd = 0.5f * bb.h[d];
        float Pb[5], gb[5], F[5];
#pragma unroll
        for (int v = 0; v < 5; ++v) {
            Pb[v] = fP[v * 896 + ni];
            gb[v] = e ? hG[v * 128 + 64 + tt[d]] : tG[v * 512 + ni];
        }
        euler_flux3(Pc, Pb, gc, gb, Dc, fD[ni], hd, e ? dBs[s] : hd, e ? invs[s] : bb.rh[d], d, gas, F);
        if (bb.type[s] == SIDE_FINE) {  // workgroup-uniform: 3 more sub-faces, straight from global memory
            if (e) {
                const int32_t* ft = ftab + (((size_t)bb.fine * 6 + s) * 64 + tt[d]) * 3;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const uint32_t cl = (uint32_t)ft[k];
                    float Y[5];
#pragma unroll
                    for (int v = 0; v < 5; ++v) {
                        Pb[v] = ldg(P + (size_t)v * ldp, cl);
                        gb[v] = ldg(G + (size_t)(d * 5 + v) * nc, cl);
                    }
                    euler_flux3(Pc, Pb, gc, gb, Dc, ldg(Gs, cl), hd, dBs[s], invs[s], d, gas, Y);
#pragma unroll
                    for (int v = 0; v < 5; ++v) F[v] += Y[v];
                }
#pragma unroll
                for (int v = 0; v < 5; ++v) F[v] *= 0.25f;
            }
        }
        // the next dim's gradients are requested now and land during the barrier and the low-flux phase
        float gn[5], hn[5];
        if (d < 2) {
#pragma unroll
            for (int v = 0; v < 5; ++v) {
                gn[v] = ldg(G + (size_t)((d + 1) * 5 + v) * nc, c);
                hn[v] = (tid >> 7) == d + 1 ? ldg(G + (size_t)((d + 1) * 5 + v) * nc, hidx) : 0.0f;  // wave-uniform
            }
        }
#pragma unroll
        for (int v = 0; v < 5; ++v) FF[v * 512 + tid] = F[v];
        __syncthreads();
#pragma unroll
        for (int v = 0; v < 5; ++v) {
            const float Flow = L.edge[2 * d] ? ex[(d * 5 + v) * 64 + tt[d]] : FF[v * 512 + tid - offm[d]];
            res[v] = res[v] - (F[v] - Flow) * bb.rh[d];
        }
        if (d < 2) {  // into the other buffer: last read before the barrier above, one dim ago
            float* tN = gbuf + ((d + 1) & 1) * 3200;
#pragma unroll
            for (int v = 0; v < 5; ++v) {
                gc[v] = gn[v];
                tN[v * 512 + tid] = gn[v];
                if ((tid >> 7) == d + 1) tN[2560 + v * 128 + (tid & 127)] = hn[v];
            }
        }
    }
    if (!L.general) {
#pragma unroll
        for (int v = 0; v < 5; ++v) stg(Rr + (size_t)v * ldr, c, res[v]);
    }
}

// ------------------------------------------------------------------------------------------
// Single-kernel Euler sweep in 3-D (P = [p T u v w]; JST sensor of the pressure, MUSCL(high_order), HLL, Green-Gauss):
// the scheme of blk3::sweep_adv with five variables.  Nothing goes through the gradient workspace (40 B per cell moved
// instead of 40 + 128).  Wavefront s (0..5) owns side s: halo values (all five primitives), deeper values, the slope of
// every primitive along the side normal and the pressure sensor of its halo cell(s) stay in its registers; only the
// pressure needs the side's plane (lateral neighbours of the sensor).  The slopes of the block's own cells are staged
// one direction at a time.
// LDS (floats): fP 5x896 | fD 512 | tS 2 x 5x512 | ex 5x384 | FF 5x512 | planes 6 x 18x18 | planeA 6x64
// ------------------------------------------------------------------------------------------
#define BLK3_SWEEP_EULER_LDS (5 * 896 + 512 + 2 * 2560 + 1920 + 2560 + 6 * 324 + 384)

// MUSCL states from undivided slopes, then HLL (blk2::euler_flux_w with five primitives)
__device__ __forceinline__ void euler_flux_w3(const float* Pa, const float* Pb, const float* Sa, const float* Sb, float Da,
                                              float Db, float wa, int dn, const Gas3& gas, float* F) {
    float PL[5], PR[5];
    const float Df = fmaxf(fmaxf(Da, Db), 1e-7f);
    const float wb = 1.0f - wa;
#pragma unroll
    for (int v = 0; v < 5; ++v) {
        const float d = Pb[v] - Pa[v];
        const float gu = Sa[v] - d * wa;
        const float Du = Sb[v] - d * wb;
        const float s = __builtin_amdgcn_fmed3f(Du, gu, 0.0f);
        const float t16 = (Sa[v] - Sb[v]) * 0.0625f;
        const float uf = (Pa[v] + wa * d) + t16;
        PL[v] = uf + Df * ((s - wa * d) - t16);
        PR[v] = uf + Df * ((wb * d - s) - t16);
    }
    float QL[5], FL[5], QR[5], FR[5], uL, aL, uR, aR;
    euler_side3(PL, dn, gas, QL, FL, uL, aL);
    euler_side3(PR, dn, gas, QR, FR, uR, aR);
    const float SR = fminf(uR - aR, 0.0f);
    const float SL = fmaxf(uL + aL, 0.0f);
    const float rs = __builtin_amdgcn_rcpf(SL - SR);
#pragma unroll
    for (int v = 0; v < 5; ++v) F[v] = (SL * FL[v] - SR * FR[v] + SR * SL * (QR[v] - QL[v])) * rs;
}

__device__ __forceinline__ void sweep_euler(const BlockDesc3* __restrict__ blocks, const int32_t* __restrict__ htab,
                                            const int32_t* __restrict__ ftab, const int32_t* __restrict__ rtab,
                                            const int32_t* __restrict__ r4tab, int32_t blk, const float* __restrict__ P,
                                            uint32_t ldp, float* __restrict__ Rr, uint32_t ldr, Gas3 gas, float* lds,
                                            int tid) {
    const BlockDesc3 bb = blocks[blk];
    const uint32_t c = (uint32_t)bb.base + tid;
    float* fP = lds;                  // [5][tile 512 | halo 384]: halo part = mean value behind every boundary cell
    float* fD = fP + 5 * 896;         // [512]
    float* tS2 = fD + 512;            // 2 x [5][512] undivided slopes of the block's cells along the current direction
    float* ex = tS2 + 2 * 2560;       // [5][6][64] mean flux through the sub-face(s) of every boundary cell
    float* FF = ex + 1920;            // [5][512]
    float* plane = FF + 2560;         // [6][18 x 18] pressure
    float* planeA = plane + 6 * 324;  // [6][64]
    // ---- this thread's halo slot: side = wavefront index, boundary cell t
    const int sw = __builtin_amdgcn_readfirstlane(tid >> 6);
    int ty = SIDE_MIRROR;
    float qs = 0.5f;
    switch (sw) {  // scalar registers, no dynamic indexing
        case 0: ty = bb.type[0]; qs = bb.q[0]; break;
        case 1: ty = bb.type[1]; qs = bb.q[1]; break;
        case 2: ty = bb.type[2]; qs = bb.q[2]; break;
        case 3: ty = bb.type[3]; qs = bb.q[3]; break;
        case 4: ty = bb.type[4]; qs = bb.q[4]; break;
        case 5: ty = bb.type[5]; qs = bb.q[5]; break;
        default: break;
    }
    const int dnh = sw >> 1;                         // normal dim of the slot's side (3: the two idle wavefronts)
    const bool low = (sw & 1) == 0, slotw = sw < 6;
    const bool isC = ty == SIDE_COARSE, isF = ty == SIDE_FINE, mirror = ty == SIDE_MIRROR;
    const int sd = dnh == 0 ? 1 : dnh == 1 ? 8 : 64, sa = dnh == 0 ? 8 : 1, sb = dnh == 2 ? 8 : 64;
    const int t = tid & 63, t1 = t & 7, t2 = t >> 3;
    const int n = isF ? 16 : 8;
    const int dd = mirror ? 0 : (low ? -sd : sd);
    // ---- loads: own cell; halo chunk(s): the five primitives and their values one step deeper; rim cells (pressure)
    float Pc[5];
#pragma unroll
    for (int v = 0; v < 5; ++v) Pc[v] = ldg(P + (size_t)v * ldp, c);
    // chunk 0 of the slot stays in registers; the other three finer cells of a FINE side are fetched where needed
    float hu[5], hde[5], hmean[5];
    const int32_t* ft = ftab + (((size_t)(isF ? bb.fine : 0) * 6 + (slotw ? sw : 0)) * 64 + t) * 3;
    float pf[4];  // pressure of the chunks (plane)
    {
        const uint32_t h0 = halo_cell3(bb, htab, blk, tid);
#pragma unroll
        for (int v = 0; v < 5; ++v) {
            hu[v] = ldg(P + (size_t)v * ldp, h0);
            hde[v] = ldg(P + (size_t)v * ldp, (uint32_t)((int)h0 + dd));
            hmean[v] = hu[v];
        }
        pf[0] = pf[1] = pf[2] = pf[3] = hu[0];
        if (isF) {  // wave-uniform
            float acc[5];
#pragma unroll
            for (int v = 0; v < 5; ++v) acc[v] = hu[v];
#pragma unroll
            for (int k = 1; k < 4; ++k) {
                const uint32_t hk = (uint32_t)ft[k - 1];
#pragma unroll
                for (int v = 0; v < 5; ++v) {
                    const float w = ldg(P + (size_t)v * ldp, hk);
                    acc[v] += w;
                    if (v == 0) pf[k] = w;
                }
            }
#pragma unroll
            for (int v = 0; v < 5; ++v) hmean[v] = 0.25f * acc[v];
        }
    }
    float rv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const bool rimw = slotw && t < 4 * n;
    if (rimw) {
        const int32_t rid = rtab[sw * 64 + t];
        if (rid >= 0) {
            rv[0] = rv[1] = rv[2] = rv[3] = ldg(P, (uint32_t)rid);
        } else {
            const int32_t* r4 = r4tab + (size_t)(-rid - 1) * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) rv[k] = ldg(P, (uint32_t)r4[k]);
        }
    }
    // ---- stage
#pragma unroll
    for (int v = 0; v < 5; ++v) fP[v * 896 + tid] = Pc[v];
    float* pl = plane + (slotw ? sw : 0) * 324;
    int radj = 0;
    if (slotw) {
#pragma unroll
        for (int v = 0; v < 5; ++v) fP[v * 896 + 512 + tid] = hmean[v];
        if (isF) {
#pragma unroll
            for (int k = 0; k < 4; ++k) pl[(2 * t1 + (k & 1) + 1) + 18 * (2 * t2 + (k >> 1) + 1)] = pf[k];
        } else {
            pl[(t1 + 1) + 18 * (t2 + 1)] = hu[0];
        }
        if (rimw) {
            const int r = isF ? t >> 4 : t >> 3, i = isF ? t & 15 : t & 7;
            const int p1 = r == 0 ? 0 : r == 1 ? n + 1 : i + 1, p2 = r == 2 ? 0 : r == 3 ? n + 1 : i + 1;
            const int a1 = r == 0 ? 1 : r == 1 ? n : i + 1, a2 = r == 2 ? 1 : r == 3 ? n : i + 1;
            pl[p1 + 18 * p2] = 0.25f * ((rv[0] + rv[1]) + (rv[2] + rv[3]));
            radj = a1 + 18 * a2;
        }
    }
    const Lane3 L = lane_info(bb, tid);
    __syncthreads();
    if (rimw) {
        const float ha = pl[radj];
        planeA[sw * 64 + t] = 0.25f * ((fabsf(rv[0] - ha) + fabsf(rv[1] - ha)) + (fabsf(rv[2] - ha) + fabsf(rv[3] - ha)));
    }
    const int tt[3] = {L.j + 8 * L.k, L.i + 8 * L.k, L.i + 8 * L.j};
    // ---- own cells: pressure sensor
    float D = 1e-7f;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float vm[2], am[2];
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const int s = 2 * d + side;
            const float v0 = fP[L.nidx[s]];
            vm[side] = v0;
            am[side] = fabsf(v0 - Pc[0]);
            if (bb.type[s] == SIDE_FINE) {  // workgroup-uniform
                if (L.edge[s]) {
                    const float* ps = plane + s * 324 + (2 * (tt[d] & 7) + 1) + 18 * (2 * (tt[d] >> 3) + 1);
                    const float w0 = ps[0], w1 = ps[1], w2 = ps[18], w3 = ps[19];
                    am[side] = 0.25f * ((fabsf(w0 - Pc[0]) + fabsf(w1 - Pc[0])) + (fabsf(w2 - Pc[0]) + fabsf(w3 - Pc[0])));
                }
            }
        }
        const float dr = vm[1] - Pc[0], dl = Pc[0] - vm[0];
        D = fmaxf(D, jst_ratio(dr - dl, am[1] + am[0], bb.rh[d]));
    }
    fD[tid] = D;
    // ---- halo cell (chunk k) of this thread's slot: slopes of the five primitives along the normal, pressure sensor
    const int nrm = (low ? 0 : 7) * sd;
    const int pos = nrm + t1 * sa + t2 * sb;  // the slot's boundary cell
    const int cm = isC ? 1 : 0;
    const float irt = isC ? 0.5f : isF ? 2.0f : 1.0f;  // h / h_halo
    float rn, ra, rb;  // reciprocal widths of the halo cell along the normal and the two tangential dims
    if (dnh == 0) { rn = bb.rh[0]; ra = bb.rh[1]; rb = bb.rh[2]; }
    else if (dnh == 1) { rn = bb.rh[1]; ra = bb.rh[0]; rb = bb.rh[2]; }
    else { rn = bb.rh[2]; ra = bb.rh[0]; rb = bb.rh[1]; }
    rn *= irt;
    ra *= irt;
    rb *= irt;
    // mean[v]: the block's cells in front of the halo cell (one, or the 2 x 2 group of a coarse cell); m4: their pressures
    auto halo_eval = [&](int k, const float* h, const float* hd, const float* mean, const float* m4, float* Sh, float& Dh) {
#pragma unroll
        for (int v = 0; v < 5; ++v) {
            const float x = (1.0f - qs) * (mean[v] - h[v]) - 0.5f * (hd[v] - h[v]);
            Sh[v] = low ? x : -x;
        }
        const float* pA = planeA + sw * 64;
        const float p = h[0];
        const float din = mean[0] - p;
        const float ain = 0.25f * ((fabsf(m4[0] - p) + fabsf(m4[1] - p)) + (fabsf(m4[2] - p) + fabsf(m4[3] - p)));
        const float dde = hd[0] - p;
        const int f1 = isF ? 2 * t1 + (k & 1) : t1, f2 = isF ? 2 * t2 + (k >> 1) : t2;
        const int fa = f1 & ~cm, fb = f1 | cm, ga = f2 & ~cm, gb = f2 | cm;
        const float ea0 = pl[fa + 18 * (f2 + 1)] - p, ea1 = pl[fb + 2 + 18 * (f2 + 1)] - p;
        const float eb0 = pl[(f1 + 1) + 18 * ga] - p, eb1 = pl[(f1 + 1) + 18 * (gb + 2)] - p;
        const float aa0 = fa == 0 ? pA[f2] : fabsf(ea0), aa1 = fb == n - 1 ? pA[n + f2] : fabsf(ea1);
        const float ab0 = ga == 0 ? pA[2 * n + f1] : fabsf(eb0), ab1 = gb == n - 1 ? pA[3 * n + f1] : fabsf(eb1);
        float dh = jst_ratio(din + dde, ain + fabsf(dde), rn);
        dh = fmaxf(dh, jst_ratio(ea0 + ea1, aa0 + aa1, ra));
        dh = fmaxf(dh, jst_ratio(eb0 + eb1, ab0 + ab1, rb));
        Dh = fmaxf(dh, 1e-7f);
    };
    float Sh0[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, Dh0 = 1e-7f;
    if (slotw) {
        blk2::wave_lds_sync();  // planeA of this side (written by this wavefront)
        const int pa = t1 & ~cm, pb = t1 | cm, qa = t2 & ~cm, qb = t2 | cm;
        const int g0 = nrm + pa * sa + qa * sb, g1 = nrm + pb * sa + qa * sb, g2 = nrm + pa * sa + qb * sb,
                  g3 = nrm + pb * sa + qb * sb;
        float mean[5], m4[4];
#pragma unroll
        for (int v = 0; v < 5; ++v) {
            const float* f = fP + v * 896;
            const float m0 = f[g0], m1 = f[g1], m2 = f[g2], m3 = f[g3];
            mean[v] = 0.25f * ((m0 + m1) + (m2 + m3));
            if (v == 0) { m4[0] = m0; m4[1] = m1; m4[2] = m2; m4[3] = m3; }
        }
        halo_eval(0, hu, hde, mean, m4, Sh0, Dh0);
    }
    // ---- one direction after the other: own slopes -> LDS, inner faces by the cells, block faces by the two side waves
    float res[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float* tS = tS2 + (d & 1) * 2560;  // double buffered: one barrier per direction
        float S[5];
        {
            const int n0 = L.nidx[2 * d], n1 = L.nidx[2 * d + 1];
            const float q0 = L.q[2 * d], q1 = L.q[2 * d + 1];
#pragma unroll
            for (int v = 0; v < 5; ++v) {
                const float* f = fP + v * 896;
                S[v] = q1 * (f[n1] - Pc[v]) + q0 * (Pc[v] - f[n0]);
                tS[v * 512 + tid] = S[v];
            }
        }
        __syncthreads();  // slopes of this direction (first time: fD too)
        float F[5];
        {
            const int ni = L.edge[2 * d + 1] ? tid : L.nidx[2 * d + 1];
            float Pb[5], Sb[5];
#pragma unroll
            for (int v = 0; v < 5; ++v) {
                Pb[v] = fP[v * 896 + ni];
                Sb[v] = tS[v * 512 + ni];
            }
            euler_flux_w3(Pc, Pb, S, Sb, D, fD[ni], 0.5f, d, gas, F);
        }
        if (dnh == d) {  // wave-uniform: this wavefront's side is normal to d
            float Pbc[5], Sbc[5];
#pragma unroll
            for (int v = 0; v < 5; ++v) {
                Pbc[v] = fP[v * 896 + pos];
                Sbc[v] = tS[v * 512 + pos];
            }
            const float Dbc = fD[pos];
            float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
            auto face = [&](const float* h, const float* sh_, float dh_) {
                float Pa[5], Pb[5], Sa[5], Sb[5], X[5];
#pragma unroll
                for (int v = 0; v < 5; ++v) {
                    const float sh = mirror ? Sbc[v] : sh_[v];
                    Pa[v] = low ? h[v] : Pbc[v];
                    Pb[v] = low ? Pbc[v] : h[v];
                    Sa[v] = low ? sh : Sbc[v];
                    Sb[v] = low ? Sbc[v] : sh;
                }
                const float dh = mirror ? Dbc : dh_;
                euler_flux_w3(Pa, Pb, Sa, Sb, low ? dh : Dbc, low ? Dbc : dh, low ? 1.0f - qs : qs, d, gas, X);
#pragma unroll
                for (int v = 0; v < 5; ++v) acc[v] += X[v];
            };
            face(hu, Sh0, Dh0);
            if (isF) {  // wave-uniform: the other three finer cells behind this boundary cell, fetched again (L2)
                const float m4[4] = {Pbc[0], Pbc[0], Pbc[0], Pbc[0]};
#pragma unroll 1
                for (int k = 1; k < 4; ++k) {
                    const uint32_t hk = (uint32_t)ft[k - 1];
                    float hk_u[5], hk_d[5], shk[5], dhk;
#pragma unroll
                    for (int v = 0; v < 5; ++v) {
                        hk_u[v] = ldg(P + (size_t)v * ldp, hk);
                        hk_d[v] = ldg(P + (size_t)v * ldp, (uint32_t)((int)hk + dd));
                    }
                    halo_eval(k, hk_u, hk_d, Pbc, m4, shk, dhk);
                    face(hk_u, shk, dhk);
                }
            }
#pragma unroll
            for (int v = 0; v < 5; ++v) ex[v * 384 + tid] = isF ? 0.25f * acc[v] : acc[v];
        }
        // the faces inside the block now (x, y: the low face is the high face of lane - 1 / lane - 8 of this wavefront;
        // z: through LDS after the last barrier); the block faces (ex) at the end
        if (d < 2) {
#pragma unroll
            for (int v = 0; v < 5; ++v) {
                const float Fh = L.edge[2 * d + 1] ? 0.0f : F[v];
                const float Fs = __shfl_up(Fh, d == 0 ? 1 : 8, 64);
                res[v] = res[v] - (Fh - (L.edge[2 * d] ? 0.0f : Fs)) * bb.rh[d];
            }
        } else {
#pragma unroll
            for (int v = 0; v < 5; ++v) {
                FF[v * 512 + tid] = F[v];
                res[v] = res[v] - (L.edge[5] ? 0.0f : F[v]) * bb.rh[2];
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int v = 0; v < 5; ++v) {
        float r = res[v] + (L.edge[4] ? 0.0f : FF[v * 512 + tid - 64]) * bb.rh[2];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float elo = ex[v * 384 + 2 * d * 64 + tt[d]], ehi = ex[v * 384 + (2 * d + 1) * 64 + tt[d]];
            r += ((L.edge[2 * d] ? elo : 0.0f) - (L.edge[2 * d + 1] ? ehi : 0.0f)) * bb.rh[d];
        }
        res[v] = r;
    }
#pragma unroll
    for (int v = 0; v < 5; ++v) stg(Rr + (size_t)v * ldr, c, res[v]);
}

// ==========================================================================================
// Wave-per-block form ("plane marching"): ONE wavefront per 8x8x8 block, lane = (i, j), the eight z-planes of the
// block handled in turn.  Each lane keeps its whole z-column in registers; x/y neighbours go through wave-private
// LDS (no workgroup barrier anywhere), z neighbours are the registers of the planes above and below.  All loads of a
// block -- the eight own values and the six halo slots of every lane -- are issued up front: one memory trip.
// The halo slots are reduced by the lanes that load them to (mean value, mean |difference to the boundary cell|), so
// a side facing finer blocks (four cells per slot) costs the consumers nothing.
// ==========================================================================================
// halo cell of boundary cell t of side S (compile-time) -- the arithmetic of halo_cell3
template <int S>
__device__ __forceinline__ uint32_t halo_cell3s(const BlockDesc3& bb, const int32_t* __restrict__ htab, int32_t blk,
                                                int t) {
    const int ty = bb.type[S];
    // wave-uniform: FINE sides, and sides whose neighbour block is not complete in this partition (nb < 0: a skirt
    // fragment, image-only sweeps) take the ids from the table
    if (ty == SIDE_FINE || ((ty == SIDE_SAME || ty == SIDE_COARSE) && bb.nb[S] < 0))
        return (uint32_t)htab[(size_t)blk * 384 + S * 64 + t];
    constexpr int d = S >> 1;
    constexpr bool low = (S & 1) == 0;
    constexpr int sd = d == 0 ? 1 : d == 1 ? 8 : 64;
    constexpr int sa = d == 0 ? 8 : 1;
    constexpr int sb = d == 2 ? 8 : 64;
    int t1 = t & 7, t2 = t >> 3;
    int n, base;
    if (ty == SIDE_SAME) {
        n = low ? 7 : 0;
        base = bb.nb[S];
    } else if (ty == SIDE_COARSE) {
        n = low ? 7 : 0;
        base = bb.nb[S];
        t1 = 4 * (bb.sub[S] & 1) + (t1 >> 1);
        t2 = 4 * (bb.sub[S] >> 1) + (t2 >> 1);
    } else {
        n = low ? 0 : 7;
        base = bb.base;
    }
    return (uint32_t)(base + n * sd + t1 * sa + t2 * sb);
}

// pass A, scalar field.  LDS per wave: tile 512 | Hm 6x64 | Ha 6x64 = 1280 floats
#define BLK3W_PASSA_LDS (512 + 384 + 384)

__device__ __forceinline__ void passA_wave(const BlockDesc3* __restrict__ blocks, const int32_t* __restrict__ htab,
                                           const int32_t* __restrict__ ftab, int32_t blk, uint32_t nc,
                                           const float* __restrict__ u, float* __restrict__ G, float* lds, int lane) {
    const BlockDesc3 bb = blocks[blk];
    float* tile = lds;          // [k][lane]
    float* Hm = lds + 512;      // [side][t]: mean neighbour value across the side
    float* Ha = lds + 896;      // [side][t]: mean |neighbour - boundary cell|
    float uk[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) uk[k] = ldg(u, (uint32_t)bb.base + lane + 64 * k);
    // this lane loads slot t = lane of every side
    uint32_t hid[6];
    hid[0] = halo_cell3s<0>(bb, htab, blk, lane);
    hid[1] = halo_cell3s<1>(bb, htab, blk, lane);
    hid[2] = halo_cell3s<2>(bb, htab, blk, lane);
    hid[3] = halo_cell3s<3>(bb, htab, blk, lane);
    hid[4] = halo_cell3s<4>(bb, htab, blk, lane);
    hid[5] = halo_cell3s<5>(bb, htab, blk, lane);
    float hv[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) hv[s] = ldg(u, hid[s]);
#pragma unroll
    for (int k = 0; k < 8; ++k) tile[k * 64 + lane] = uk[k];
    blk2::wave_lds_sync();
    // reduce every slot to (mean value, mean |difference|) against its boundary cell: slot t of side s belongs to
    // cell pos(s, t) = n*sd + t1*sa + t2*sb of the tile
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int d = s >> 1;
        const bool low = (s & 1) == 0;
        const int sd = d == 0 ? 1 : d == 1 ? 8 : 64, sa = d == 0 ? 8 : 1, sb = d == 2 ? 8 : 64;
        const int pos = (low ? 0 : 7) * sd + (lane & 7) * sa + (lane >> 3) * sb;
        const float ub = tile[pos];
        float vm = hv[s], am = fabsf(hv[s] - ub);
        if (bb.type[s] == SIDE_FINE) {  // wave-uniform: three more fine cells behind this slot
            const int32_t* ft = ftab + (((size_t)bb.fine * 6 + s) * 64 + lane) * 3;
            const float v1 = ldg(u, (uint32_t)ft[0]), v2 = ldg(u, (uint32_t)ft[1]), v3 = ldg(u, (uint32_t)ft[2]);
            vm = 0.25f * (hv[s] + v1 + v2 + v3);
            am = 0.25f * (fabsf(hv[s] - ub) + fabsf(v1 - ub) + fabsf(v2 - ub) + fabsf(v3 - ub));
        }
        Hm[s * 64 + lane] = vm;
        Ha[s * 64 + lane] = am;
    }
    blk2::wave_lds_sync();
    const int i = lane & 7, j = lane >> 3;
    const bool e0 = i == 0, e1 = i == 7, e2 = j == 0, e3 = j == 7;
    bool general = (e0 && bb.type[0] == SIDE_GENERAL) || (e1 && bb.type[1] == SIDE_GENERAL) ||
                   (e2 && bb.type[2] == SIDE_GENERAL) || (e3 && bb.type[3] == SIDE_GENERAL);
    const float q0 = e0 ? bb.q[0] : 0.5f, q1 = e1 ? bb.q[1] : 0.5f, q2 = e2 ? bb.q[2] : 0.5f, q3 = e3 ? bb.q[3] : 0.5f;
    const float zlm = Hm[4 * 64 + lane], zla = Ha[4 * 64 + lane], zhm = Hm[5 * 64 + lane], zha = Ha[5 * 64 + lane];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float uc = uk[k];
        const float* tk = tile + k * 64;
        // x, y: inside the plane or the side's reduced slot (x sides: t = j + 8k, y sides: t = i + 8k)
        float vm[6], am[6];
        vm[0] = e0 ? Hm[0 * 64 + j + 8 * k] : tk[lane - 1];
        vm[1] = e1 ? Hm[1 * 64 + j + 8 * k] : tk[lane + 1];
        vm[2] = e2 ? Hm[2 * 64 + i + 8 * k] : tk[lane - 8];
        vm[3] = e3 ? Hm[3 * 64 + i + 8 * k] : tk[lane + 8];
        am[0] = e0 ? Ha[0 * 64 + j + 8 * k] : fabsf(vm[0] - uc);
        am[1] = e1 ? Ha[1 * 64 + j + 8 * k] : fabsf(vm[1] - uc);
        am[2] = e2 ? Ha[2 * 64 + i + 8 * k] : fabsf(vm[2] - uc);
        am[3] = e3 ? Ha[3 * 64 + i + 8 * k] : fabsf(vm[3] - uc);
        // z: registers
        vm[4] = k == 0 ? zlm : uk[k > 0 ? k - 1 : 0];
        am[4] = k == 0 ? zla : fabsf(vm[4] - uc);
        vm[5] = k == 7 ? zhm : uk[k < 7 ? k + 1 : 7];
        am[5] = k == 7 ? zha : fabsf(vm[5] - uc);
        const float ql[3] = {q0, q2, k == 0 ? bb.q[4] : 0.5f}, qh[3] = {q1, q3, k == 7 ? bb.q[5] : 0.5f};
        float D = 1e-7f, g[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float rh = bb.rh[d];
            const float fr = uc + qh[d] * (vm[2 * d + 1] - uc);
            const float fl = uc + ql[d] * (vm[2 * d] - uc);
            g[d] = (fr - fl) * rh;
            const float gg = ((vm[2 * d + 1] - uc) - (uc - vm[2 * d])) * rh;
            const float ugg = (am[2 * d + 1] + am[2 * d]) * rh;
            D = fmaxf(D, (1e-7f + fabsf(gg)) * __builtin_amdgcn_rcpf(1e-7f + ugg));
        }
        const bool gen = general || (k == 0 && bb.type[4] == SIDE_GENERAL) || (k == 7 && bb.type[5] == SIDE_GENERAL);
        if (!gen) {
            const uint32_t c = (uint32_t)bb.base + lane + 64 * k;
            stg(G, c, g[0]);
            stg(G + nc, c, g[1]);
            stg(G + (size_t)2 * nc, c, g[2]);
            stg(G + (size_t)3 * nc, c, D);
        }
    }
}

// the same for NV variables (Euler: 5 primitives), one variable after the other through the same LDS region;
// G layout of the face-list kernels: grad of var v along dim d at G[(d*NV + v)*nc + c], sensor of var 0 at G[3*NV*nc + c]
template <int NV>
__device__ __forceinline__ void passA_wave_nv(const BlockDesc3* __restrict__ blocks, const int32_t* __restrict__ htab,
                                              const int32_t* __restrict__ ftab, int32_t blk, uint32_t nc,
                                              const float* __restrict__ P, uint32_t ldp, float* __restrict__ G,
                                              float* lds, int lane) {
    const BlockDesc3 bb = blocks[blk];
    float* tile = lds;          // [k][lane]
    float* Hm = lds + 512;      // [side][t]: mean neighbour value across the side
    float* Ha = lds + 896;      // [side][t]: mean |neighbour - boundary cell|
    // this lane loads slot t = lane of every side
    uint32_t hid[6];
    hid[0] = halo_cell3s<0>(bb, htab, blk, lane);
    hid[1] = halo_cell3s<1>(bb, htab, blk, lane);
    hid[2] = halo_cell3s<2>(bb, htab, blk, lane);
    hid[3] = halo_cell3s<3>(bb, htab, blk, lane);
    hid[4] = halo_cell3s<4>(bb, htab, blk, lane);
    hid[5] = halo_cell3s<5>(bb, htab, blk, lane);
    const int i = lane & 7, j = lane >> 3;
    const bool e0 = i == 0, e1 = i == 7, e2 = j == 0, e3 = j == 7;
    const bool general = (e0 && bb.type[0] == SIDE_GENERAL) || (e1 && bb.type[1] == SIDE_GENERAL) ||
                         (e2 && bb.type[2] == SIDE_GENERAL) || (e3 && bb.type[3] == SIDE_GENERAL);
    const float q0 = e0 ? bb.q[0] : 0.5f, q1 = e1 ? bb.q[1] : 0.5f, q2 = e2 ? bb.q[2] : 0.5f, q3 = e3 ? bb.q[3] : 0.5f;
    // all loads of the block up front (one memory trip), then one variable after the other through the LDS region
    float uka[NV][8], hva[NV][6];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const float* Pv = P + (size_t)v * ldp;
#pragma unroll
        for (int k = 0; k < 8; ++k) uka[v][k] = ldg(Pv, (uint32_t)bb.base + lane + 64 * k);
#pragma unroll
        for (int s = 0; s < 6; ++s) hva[v][s] = ldg(Pv, hid[s]);
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
    const float* u = P + (size_t)v * ldp;
    float uk[8], hv[6];
#pragma unroll
    for (int k = 0; k < 8; ++k) uk[k] = uka[v][k];
#pragma unroll
    for (int s = 0; s < 6; ++s) hv[s] = hva[v][s];
    if (v) blk2::wave_lds_sync();  // the previous variable's LDS reads are done
#pragma unroll
    for (int k = 0; k < 8; ++k) tile[k * 64 + lane] = uk[k];
    blk2::wave_lds_sync();
    // reduce every slot to (mean value, mean |difference|) against its boundary cell: slot t of side s belongs to
    // cell pos(s, t) = n*sd + t1*sa + t2*sb of the tile
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int d = s >> 1;
        const bool low = (s & 1) == 0;
        const int sd = d == 0 ? 1 : d == 1 ? 8 : 64, sa = d == 0 ? 8 : 1, sb = d == 2 ? 8 : 64;
        const int pos = (low ? 0 : 7) * sd + (lane & 7) * sa + (lane >> 3) * sb;
        const float ub = tile[pos];
        float vm = hv[s], am = fabsf(hv[s] - ub);
        if (bb.type[s] == SIDE_FINE) {  // wave-uniform: three more fine cells behind this slot
            const int32_t* ft = ftab + (((size_t)bb.fine * 6 + s) * 64 + lane) * 3;
            const float v1 = ldg(u, (uint32_t)ft[0]), v2 = ldg(u, (uint32_t)ft[1]), v3 = ldg(u, (uint32_t)ft[2]);
            vm = 0.25f * (hv[s] + v1 + v2 + v3);
            am = 0.25f * (fabsf(hv[s] - ub) + fabsf(v1 - ub) + fabsf(v2 - ub) + fabsf(v3 - ub));
        }
        Hm[s * 64 + lane] = vm;
        Ha[s * 64 + lane] = am;
    }
    blk2::wave_lds_sync();
    const float zlm = Hm[4 * 64 + lane], zla = Ha[4 * 64 + lane], zhm = Hm[5 * 64 + lane], zha = Ha[5 * 64 + lane];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float uc = uk[k];
        const float* tk = tile + k * 64;
        // x, y: inside the plane or the side's reduced slot (x sides: t = j + 8k, y sides: t = i + 8k)
        float vm[6], am[6];
        vm[0] = e0 ? Hm[0 * 64 + j + 8 * k] : tk[lane - 1];
        vm[1] = e1 ? Hm[1 * 64 + j + 8 * k] : tk[lane + 1];
        vm[2] = e2 ? Hm[2 * 64 + i + 8 * k] : tk[lane - 8];
        vm[3] = e3 ? Hm[3 * 64 + i + 8 * k] : tk[lane + 8];
        am[0] = e0 ? Ha[0 * 64 + j + 8 * k] : fabsf(vm[0] - uc);
        am[1] = e1 ? Ha[1 * 64 + j + 8 * k] : fabsf(vm[1] - uc);
        am[2] = e2 ? Ha[2 * 64 + i + 8 * k] : fabsf(vm[2] - uc);
        am[3] = e3 ? Ha[3 * 64 + i + 8 * k] : fabsf(vm[3] - uc);
        // z: registers
        vm[4] = k == 0 ? zlm : uk[k > 0 ? k - 1 : 0];
        am[4] = k == 0 ? zla : fabsf(vm[4] - uc);
        vm[5] = k == 7 ? zhm : uk[k < 7 ? k + 1 : 7];
        am[5] = k == 7 ? zha : fabsf(vm[5] - uc);
        const float ql[3] = {q0, q2, k == 0 ? bb.q[4] : 0.5f}, qh[3] = {q1, q3, k == 7 ? bb.q[5] : 0.5f};
        float D = 1e-7f, g[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float rh = bb.rh[d];
            const float fr = uc + qh[d] * (vm[2 * d + 1] - uc);
            const float fl = uc + ql[d] * (vm[2 * d] - uc);
            g[d] = (fr - fl) * rh;
            const float gg = ((vm[2 * d + 1] - uc) - (uc - vm[2 * d])) * rh;
            const float ugg = (am[2 * d + 1] + am[2 * d]) * rh;
            D = fmaxf(D, (1e-7f + fabsf(gg)) * __builtin_amdgcn_rcpf(1e-7f + ugg));
        }
        const bool gen = general || (k == 0 && bb.type[4] == SIDE_GENERAL) || (k == 7 && bb.type[5] == SIDE_GENERAL);
        if (!gen) {
            const uint32_t c = (uint32_t)bb.base + lane + 64 * k;
            stg(G + (size_t)(0 * NV + v) * nc, c, g[0]);
            stg(G + (size_t)(1 * NV + v) * nc, c, g[1]);
            stg(G + (size_t)(2 * NV + v) * nc, c, g[2]);
            if (v == 0) stg(G + (size_t)(3 * NV) * nc, c, D);
        }
    }
    }
}


// gradients of NV cell fields of one block in registers, g[v][k][d] (plane k of this lane's column, dimension d): the
// arithmetic of passA_wave_nv without the stores and the sensor -- for kernels that consume the gradients where they
// are made (shear rate of a velocity field, the Wray-Agarwal closure: ibh_fused.hip).  Blocks without GENERAL sides only.
template <int NV>
__device__ __forceinline__ void wave_gradients(const BlockDesc3& bb, const int32_t* __restrict__ htab,
                                               const int32_t* __restrict__ ftab, int32_t blk, const float* const* F,
                                               float* lds, int lane, float (&g)[NV][8][3]) {
    float* tile = lds;          // [k][lane]
    float* Hm = lds + 512;      // [side][t]: mean neighbour value across the side
    uint32_t hid[6];
    hid[0] = halo_cell3s<0>(bb, htab, blk, lane);
    hid[1] = halo_cell3s<1>(bb, htab, blk, lane);
    hid[2] = halo_cell3s<2>(bb, htab, blk, lane);
    hid[3] = halo_cell3s<3>(bb, htab, blk, lane);
    hid[4] = halo_cell3s<4>(bb, htab, blk, lane);
    hid[5] = halo_cell3s<5>(bb, htab, blk, lane);
    const int i = lane & 7, j = lane >> 3;
    const bool e0 = i == 0, e1 = i == 7, e2 = j == 0, e3 = j == 7;
    const float q0 = e0 ? bb.q[0] : 0.5f, q1 = e1 ? bb.q[1] : 0.5f, q2 = e2 ? bb.q[2] : 0.5f, q3 = e3 ? bb.q[3] : 0.5f;
    float uka[NV][8], hva[NV][6];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
#pragma unroll
        for (int k = 0; k < 8; ++k) uka[v][k] = ldg(F[v], (uint32_t)bb.base + lane + 64 * k);
#pragma unroll
        for (int s = 0; s < 6; ++s) hva[v][s] = ldg(F[v], hid[s]);
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const float* u = F[v];
        if (v) blk2::wave_lds_sync();  // the previous field's LDS reads are done
#pragma unroll
        for (int k = 0; k < 8; ++k) tile[k * 64 + lane] = uka[v][k];
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            float vm = hva[v][s];
            if (bb.type[s] == SIDE_FINE) {  // wave-uniform: three more fine cells behind this slot
                const int32_t* ft = ftab + (((size_t)bb.fine * 6 + s) * 64 + lane) * 3;
                const float v1 = ldg(u, (uint32_t)ft[0]), v2 = ldg(u, (uint32_t)ft[1]), v3 = ldg(u, (uint32_t)ft[2]);
                vm = 0.25f * (hva[v][s] + v1 + v2 + v3);
            }
            Hm[s * 64 + lane] = vm;
        }
        blk2::wave_lds_sync();
        const float zlm = Hm[4 * 64 + lane], zhm = Hm[5 * 64 + lane];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float uc = uka[v][k];
            const float* tk = tile + k * 64;
            float vm[6];
            vm[0] = e0 ? Hm[0 * 64 + j + 8 * k] : tk[lane - 1];
            vm[1] = e1 ? Hm[1 * 64 + j + 8 * k] : tk[lane + 1];
            vm[2] = e2 ? Hm[2 * 64 + i + 8 * k] : tk[lane - 8];
            vm[3] = e3 ? Hm[3 * 64 + i + 8 * k] : tk[lane + 8];
            vm[4] = k == 0 ? zlm : uka[v][k > 0 ? k - 1 : 0];
            vm[5] = k == 7 ? zhm : uka[v][k < 7 ? k + 1 : 7];
            const float ql[3] = {q0, q2, k == 0 ? bb.q[4] : 0.5f}, qh[3] = {q1, q3, k == 7 ? bb.q[5] : 0.5f};
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const float fr = uc + qh[d] * (vm[2 * d + 1] - uc);
                const float fl = uc + ql[d] * (vm[2 * d] - uc);
                g[v][k][d] = (fr - fl) * bb.rh[d];
            }
        }
    }
}

#pragma clang fp contract(off)

}  // namespace blk3
