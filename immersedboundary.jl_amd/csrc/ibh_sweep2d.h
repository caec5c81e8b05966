// Single-kernel sweep (2-D scalar advection-JST-MUSCL residual, test/advection.jl:67-83): one wavefront per
// 8x8 block computes gradients + sensor of its 64 own cells AND of its 64 halo cells, then the face fluxes
// and the Green-Gauss sum.  Nothing goes through the gradient workspace and one launch does the sweep.
//
// A halo cell h across side s (normal dim n, tangential dim t) needs
//   * its faces towards this block: the one or two boundary cells of this block (tile),
//   * the cell one step deeper inside its own block: id -/+ 1 (x sides) or -/+ 8 (y sides),
//   * its lateral neighbours along the side: the adjacent halo slots, and at the two ends of the side the
//     cells of the end table (ibh_analyze.cpp step 6).
// Per side the halo values sit in a line ext[s][20] = [lo end k=0,1 | slots p = 2*t+k, 16 of them | hi end k=0,1];
// a halo "cell" covers w = 2 / 4 / 1 consecutive entries (SAME / COARSE / FINE) and its lateral neighbours are
// the entries before and after them.  MIRROR sides: the halo cell is the boundary cell itself, so its slope
// and sensor are the ones of that cell.
//
// The kernel is bound by VALU issue (profiles/r1_v4), so the arithmetic is arranged for instruction count:
//   * slopes are kept UNDIVIDED, S = g*h (cell_gradient times the cell's own width): with
//     wa = h_a/(h_a+h_b) (1/2, 1/3 or 2/3), d = u_b - u_a the MUSCL increments of MUSCL(:1113-1157) are
//     (2 g_a - d/(dA+dB)) dA = S_a - wa d  and  S_b - (1-wa) d, the high-order term is (S_a - S_b)/16;
//   * the upwind flux (uL+uR) Cf/2 + |Cf| (uL-uR)/2 is evaluated as Cf (uf + Df (m - uf)) + |Cf| Df (s - d/2)
//     with m - uf = (1/2 - wa) d - (S_a - S_b)/16: 19 operations per face.
// LDS per wave: the six [tile | halo] fields U, D, SX, SY, CX, CY + extra[64] + ext[80].
#pragma once
#include "ibh_block2d.h"

namespace blk2 {

#pragma clang fp contract(fast)

#define BLK2_SWEEP_LDS (6 * 128 + 64 + 80)

__device__ __forceinline__ int seli4(int s, int a0, int a1, int a2, int a3) {
    const int lo = (s & 1) ? a1 : a0;
    const int hi = (s & 1) ? a3 : a2;
    return (s & 2) ? hi : lo;
}

// flux through the face between cell a (owner, towards -) and cell b (neighbour, towards +);
// Sa, Sb undivided slopes along the face normal, wa = h_a/(h_a+h_b)
__device__ __forceinline__ float flux_w(float ua, float ub, float Sa, float Sb, float Da, float Db, float Ca, float Cb,
                                        float wa) {
    const float d = ub - ua;
    const float gu = Sa - d * wa;
    const float Du = Sb - d * (1.0f - wa);
    const float s = __builtin_amdgcn_fmed3f(Du, gu, 0.0f);
    const float t16 = (Sa - Sb) * 0.0625f;
    const float uf = (ua + wa * d) + t16;
    const float mu = d * (0.5f - wa) - t16;
    const float Df = fmaxf(fmaxf(Da, Db), 1e-7f);
    const float A = uf + Df * mu;
    const float Cf = Ca + wa * (Cb - Ca);
    const float B = Df * (s - 0.5f * d);
    return Cf * A + fabsf(Cf) * B;
}

// undivided slope and sensor term of one direction: l0,l1 / r0,r1 the neighbour values on the low / high side
// (equal when the side has one face), qL, qR the at_faces weights of the neighbours, rh = 1/h
__device__ __forceinline__ void slope_dir(float uc, float l0, float l1, float r0, float r1, float qL, float qR,
                                          float rh, float& S, float& nu) {
    const float dR = 0.5f * (r0 + r1) - uc;
    const float dL = uc - 0.5f * (l0 + l1);
    S = qR * dR + qL * dL;
    const float gg = dR - dL;
    const float a2 = (fabsf(r0 - uc) + fabsf(r1 - uc)) + (fabsf(uc - l0) + fabsf(uc - l1));
    nu = (1e-7f + fabsf(gg) * rh) * __builtin_amdgcn_rcpf(1e-7f + a2 * (0.5f * rh));
}

// what a wave fetches for a block before anything depends on anything: issued one block ahead
struct SweepPre {
    BlockDesc2 bb;          // wave-uniform (scalar registers)
    uint32_t hidx, eidx;    // halo table / end table entries of this lane
    float uc, cxc, cyc;     // own cell
    int ty;                 // class of this lane's side (lane >> 4)
    float qs;               // at_faces weight of this lane's side
};

__device__ __forceinline__ SweepPre sweep_prefetch(const BlockDesc2* __restrict__ blocks,
                                                   const int32_t* __restrict__ htab,
                                                   const int32_t* __restrict__ etab, int32_t blk,
                                                   const float* __restrict__ u, const float* __restrict__ C,
                                                   uint32_t ldc, int lane) {
    SweepPre P;
    P.bb = blocks[blk];
    P.hidx = (uint32_t)htab[(size_t)blk * 64 + lane];
    P.eidx = (uint32_t)etab[(size_t)blk * 16 + (lane & 15)];
    // per-side constants of this lane's halo slot straight from the descriptor (16 lanes share an address):
    // one load instead of a select chain over four scalar registers
    const int32_t* bw = (const int32_t*)(blocks + blk);
    P.ty = bw[1 + (lane >> 4)];
    P.qs = __int_as_float(bw[21 + (lane >> 4)]);
    const uint32_t c = (uint32_t)P.bb.base + lane;
    P.uc = ldg(u, c);
    P.cxc = ldg(C, c);
    P.cyc = ldg(C + ldc, c);
    return P;
}
static_assert(offsetof(BlockDesc2, type) == 4 && offsetof(BlockDesc2, q) == 84, "sweep_prefetch reads type/q by offset");

// `nb` blocks blk0, blk0 + stride, ... by this wave; the lane-only index arithmetic is shared by all of them
// and the independent loads of block k+1 are in flight while block k is computed
__device__ __forceinline__ void sweep_adv(const BlockDesc2* __restrict__ blocks, const int32_t* __restrict__ htab,
                                          const int32_t* __restrict__ etab, int32_t blk0, int32_t stride, int32_t nb,
                                          const float* __restrict__ u, const float* __restrict__ C, uint32_t ldc,
                                          float* __restrict__ ud, float* lds, int lane) {
    float* fU = lds;
    float* fD = lds + 128;
    float* fSX = lds + 256;
    float* fSY = lds + 384;
    float* fCX = lds + 512;
    float* fCY = lds + 640;
    float* ex = lds + 768;
    float* ext = lds + 832;
    // ---- lane-only geometry
    // halo slot of this lane: side s, boundary cell t, sub-face k
    const int s = lane >> 4, dn = lane >> 5, p = lane & 15, t = p >> 1;
    const int delta = seli4(s, -1, 1, -8, 8);
    const int pos0 = (dn ? t : 8 * t) + ((s & 1) ? (dn ? 56 : 7) : 0);
    const int posx = pos0 ^ (dn ? 1 : 8);
    float* extw = ext + s * 20 + 2 + p;
    float* exte = ext + ((lane >> 2) & 3) * 20 + ((lane >> 1) & 1) * 18 + (lane & 1);
    const float* el = ext + s * 20;
    const float* fSn = dn ? fSY : fSX;
    // cell of this lane inside the tile
    const int i = lane & 7, j = lane >> 3;
    const bool e0 = i == 0, e1 = i == 7, e2 = j == 0, e3 = j == 7;
    const int n0 = e0 ? 64 + j * 2 : lane - 1;
    const int n1 = e1 ? 80 + j * 2 : lane + 1;
    const int n2 = e2 ? 96 + i * 2 : lane - 8;
    const int n3 = e3 ? 112 + i * 2 : lane + 8;
    const int n0b = n0 + (e0 ? 1 : 0), n1b = n1 + (e1 ? 1 : 0), n2b = n2 + (e2 ? 1 : 0), n3b = n3 + (e3 ? 1 : 0);
    // role in the low-side pass: g = (lane>>3)&3: 0 left k=0, 1 bottom k=0, 2 left k=1, 3 bottom k=1
    const int xg = (lane >> 3) & 3, xt = lane & 7, xd = xg & 1;
    const int xpos = xd ? xt : 8 * xt;
    const int xslot = 64 + (xd * 16 + xt) * 2 + (xg >> 1);
    const float* xS = xd ? fSY : fSX;
    const float* xC = xd ? fCY : fCX;

    SweepPre N = sweep_prefetch(blocks, htab, etab, blk0, u, C, ldc, lane);
    for (int32_t it = 0; it < nb; ++it) {
        const SweepPre P = N;
        if (it + 1 < nb) N = sweep_prefetch(blocks, htab, etab, blk0 + (it + 1) * stride, u, C, ldc, lane);
        const BlockDesc2& bb = P.bb;
        const float uc = P.uc, cxc = P.cxc, cyc = P.cyc;
        const bool mirror = P.ty == SIDE_MIRROR, isC = P.ty == SIDE_COARSE, isF = P.ty == SIDE_FINE;
        const uint32_t didx = mirror ? P.hidx : P.hidx + (uint32_t)delta;
        const float hu = ldg(u, P.hidx), hdeep = ldg(u, didx);
        const float hc = ldg(C + (size_t)dn * ldc, P.hidx);
        const float eu = ldg(u, P.eidx);
        if (it) wave_lds_sync();  // the previous block's last LDS reads are done
        fU[lane] = uc;
        fCX[lane] = cxc;
        fCY[lane] = cyc;
        fU[64 + lane] = hu;
        fCX[64 + lane] = hc;
        fCY[64 + lane] = hc;
        *extw = hu;
        // every lane stores (lanes 16..63 repeat lanes 0..15: same address, same value): no branch, and the
        // end gather is issued with the other gathers
        *exte = eu;
        const float q0 = e0 ? bb.q[0] : 0.5f, q1 = e1 ? bb.q[1] : 0.5f;
        const float q2 = e2 ? bb.q[2] : 0.5f, q3 = e3 ? bb.q[3] : 0.5f;
        wave_lds_sync();
        // ---- own cells: undivided slopes + sensor
        float Sx, Sy, Dc;
        {
            float nx, ny;
            slope_dir(uc, fU[n0], fU[n0b], fU[n1], fU[n1b], q0, q1, bb.rh[0], Sx, nx);
            slope_dir(uc, fU[n2], fU[n2b], fU[n3], fU[n3b], q2, q3, bb.rh[1], Sy, ny);
            Dc = fmaxf(fmaxf(nx, ny), 1e-7f);
        }
        fSX[lane] = Sx;
        fSY[lane] = Sy;
        fD[lane] = Dc;
        // ---- halo cells
        float Sh, Dh;
        {
            const float m0 = fU[pos0], m1 = fU[isC ? posx : pos0];  // coarse halo cell: two fine cells face it
            const float irt = isC ? 0.5f : isF ? 2.0f : 1.0f;       // h / h_halo
            const float ihn = (dn ? bb.rh[1] : bb.rh[0]) * irt;
            const float iht = (dn ? bb.rh[0] : bb.rh[1]) * irt;
            const float din = 0.5f * (m0 + m1) - hu;   // towards this block
            const float dde = hdeep - hu;              // away from it
            const float x = (1.0f - P.qs) * din - 0.5f * dde;  // u_face,in - u_face,deep
            Sh = (s & 1) ? -x : x;
            const float an = (fabsf(m0 - hu) + fabsf(m1 - hu)) + 2.0f * fabsf(dde);
            const float nun = (1e-7f + fabsf(din + dde) * ihn) * __builtin_amdgcn_rcpf(1e-7f + an * (0.5f * ihn));
            const int mask = isF ? 15 : isC ? 12 : 14;
            const int pm = p & mask, w = 16 - mask, e0i = pm + 2;
            const bool single = mask == 15;
            const int lo0 = pm == 0 ? 0 : e0i - w;
            const int lo1 = lo0 + ((pm == 0 || !single) ? 1 : 0);
            const int hi0 = e0i + w;
            const int hi1 = hi0 + ((hi0 == 18 || !single) ? 1 : 0);
            const float l0 = el[lo0], l1 = el[lo1], r0 = el[hi0], r1 = el[hi1];
            const float gt = 0.5f * ((r0 + r1) + (l0 + l1)) - 2.0f * hu;
            const float at = (fabsf(r0 - hu) + fabsf(r1 - hu)) + (fabsf(hu - l0) + fabsf(hu - l1));
            const float nut = (1e-7f + fabsf(gt) * iht) * __builtin_amdgcn_rcpf(1e-7f + at * (0.5f * iht));
            Dh = fmaxf(fmaxf(nun, nut), 1e-7f);
        }
        wave_lds_sync();
        {
            const float Sm = fSn[pos0], Dm = fD[pos0];
            Sh = mirror ? Sm : Sh;
            Dh = mirror ? Dm : Dh;
        }
        fSX[64 + lane] = Sh;
        fSY[64 + lane] = Sh;
        fD[64 + lane] = Dh;
        wave_lds_sync();
        // ---- fluxes: right (x+) and top (y+) face of every cell, sub-face 0 on block sides
        float FR = flux_w(uc, fU[n1], Sx, fSX[n1], Dc, fD[n1], cxc, fCX[n1], q1);
        float FT = flux_w(uc, fU[n3], Sy, fSY[n3], Dc, fD[n3], cyc, fCY[n3], q3);
        // low sides: the halo cell is the owner, this block's cell the neighbour
        ex[lane] = flux_w(fU[xslot], fU[xpos], xS[xslot], xS[xpos], fD[xslot], fD[xpos], xC[xslot], xC[xpos],
                          1.0f - (xd ? bb.q[2] : bb.q[0]));
        // second sub-faces of the HIGH sides exist only next to finer blocks (~5 % of the sides): wave-uniform
        float FR1 = FR, FT1 = FT;
        if (bb.type[1] == SIDE_FINE)
            FR1 = flux_w(uc, fU[n1 + 1], Sx, fSX[n1 + 1], Dc, fD[n1 + 1], cxc, fCX[n1 + 1], bb.q[1]);
        if (bb.type[3] == SIDE_FINE)
            FT1 = flux_w(uc, fU[n3 + 1], Sy, fSY[n3 + 1], Dc, fD[n3 + 1], cyc, fCY[n3 + 1], bb.q[3]);
        // interior faces: left flux = right flux of lane-1, bottom flux = top flux of lane-8
        const float FLs = __shfl_up(FR, 1, 64);
        const float FBs = __shfl_up(FT, 8, 64);
        wave_lds_sync();
        const float eL = 0.5f * (ex[j] + ex[16 + j]), eB = 0.5f * (ex[8 + i] + ex[24 + i]);
        const float FL = e0 ? eL : FLs;
        const float FB = e2 ? eB : FBs;
        FR = e1 ? 0.5f * (FR + FR1) : FR;
        FT = e3 ? 0.5f * (FT + FT1) : FT;
        stg(ud, (uint32_t)bb.base + lane, -((FR - FL) * bb.rh[0]) - ((FT - FB) * bb.rh[1]));
    }
}

#pragma clang fp contract(off)

}  // namespace blk2
