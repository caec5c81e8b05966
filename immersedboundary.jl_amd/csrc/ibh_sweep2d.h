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

// ---- the same arithmetic on (x, y) pairs: v_pk_{add,mul,fma}_f32 do two lanes' worth per instruction
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f v2(float a, float b) { return v2f{a, b}; }
__device__ __forceinline__ v2f v2(float a) { return v2f{a, a}; }
__device__ __forceinline__ v2f absv(v2f a) { return v2f{fabsf(a.x), fabsf(a.y)}; }

__device__ __forceinline__ v2f flux_w2(v2f ua, v2f ub, v2f Sa, v2f Sb, v2f Da, v2f Db, v2f Ca, v2f Cb, v2f wa) {
    const v2f d = ub - ua;
    const v2f gu = Sa - d * wa;
    const v2f Du = Sb - d * (1.0f - wa);
    const v2f s = v2f{__builtin_amdgcn_fmed3f(Du.x, gu.x, 0.0f), __builtin_amdgcn_fmed3f(Du.y, gu.y, 0.0f)};
    const v2f t16 = (Sa - Sb) * 0.0625f;
    const v2f uf = (ua + wa * d) + t16;
    const v2f mu = d * (0.5f - wa) - t16;
    const v2f Df = v2f{fmaxf(fmaxf(Da.x, Db.x), 1e-7f), fmaxf(fmaxf(Da.y, Db.y), 1e-7f)};
    const v2f A = uf + Df * mu;
    const v2f Cf = Ca + wa * (Cb - Ca);
    const v2f B = Df * (s - 0.5f * d);
    return Cf * A + absv(Cf) * B;
}

// sensor terms of a cell of value uc in two directions at once: a0,a1 / b0,b1 the neighbour values on the two
// sides of each direction (equal when a side has one face), rh = 1/h per direction:
//   (1e-7 + |gg|)/(1e-7 + ugg)  of JST_sensor(:1077-1097) with gg, ugg the signed / unsigned Green-Gauss sums
__device__ __forceinline__ v2f sensor2(float uc, v2f a0, v2f a1, v2f b0, v2f b1, v2f rh) {
    // differences packed; the sums of absolute values scalar (|x| is a free source modifier there, packed
    // instructions would need a v_and per operand)
    const v2f c = v2(uc), hrh = 0.5f * rh;
    const v2f d0 = a0 - c, d1 = a1 - c, d2 = b0 - c, d3 = b1 - c;
    const v2f g = (d0 + d1) + (d2 + d3);  // = 2 * (signed sum)
    const float ax = (fabsf(d0.x) + fabsf(d1.x)) + (fabsf(d2.x) + fabsf(d3.x));
    const float ay = (fabsf(d0.y) + fabsf(d1.y)) + (fabsf(d2.y) + fabsf(d3.y));
    const float nx = (1e-7f + fabsf(g.x) * hrh.x) * __builtin_amdgcn_rcpf(1e-7f + ax * hrh.x);
    const float ny = (1e-7f + fabsf(g.y) * hrh.y) * __builtin_amdgcn_rcpf(1e-7f + ay * hrh.y);
    return v2f{nx, ny};
}

// what a wave fetches for a block before anything depends on anything: issued one block ahead
struct SweepPre {
    BlockDesc2 bb;          // wave-uniform (scalar registers)
    uint32_t hidx, eidx;    // halo table / end table entries of this lane
    float uc, cxc, cyc;     // own cell
    int ty;                 // class of this lane's side (lane >> 4)
    uint32_t dix;           // deeper cell of this lane's halo slot when the block has a table row (bb.dt >= 0)
    float qs;               // at_faces weight of this lane's side
};

template <bool DT>
__device__ __forceinline__ SweepPre sweep_prefetch(const BlockDesc2* __restrict__ blocks,
                                                   const int32_t* __restrict__ htab,
                                                   const int32_t* __restrict__ etab,
                                                   const int32_t* __restrict__ dtab, int32_t blk,
                                                   const float* __restrict__ u, const float* __restrict__ C,
                                                   uint32_t ldc, int lane) {
    SweepPre P;
    P.bb = blocks[blk];
    P.hidx = (uint32_t)htab[(size_t)blk * 64 + lane];
    P.eidx = (uint32_t)etab[(size_t)blk * 16 + (lane & 15)];
    P.dix = 0;
    if constexpr (DT)  // partitions with skirt fragments only (0.2 us of 9 on the one-partition headline otherwise)
        if (P.bb.dt >= 0) P.dix = (uint32_t)dtab[(size_t)P.bb.dt * 64 + lane];  // wave-uniform
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
// the dependent loads of a block (addresses from the halo / end tables): issued one block ahead as well
struct SweepGat {
    float hu, hdeep, hc, eu;
};
template <bool DT>
__device__ __forceinline__ SweepGat sweep_gather(const SweepPre& P, int delta, bool dn, const float* __restrict__ u,
                                                 const float* __restrict__ C, uint32_t ldc) {
    SweepGat G;
    const uint32_t didx = (DT && P.bb.dt >= 0) ? P.dix : (P.ty == SIDE_MIRROR ? P.hidx : P.hidx + (uint32_t)delta);
    G.hu = ldg(u, P.hidx);
    G.hdeep = ldg(u, didx);
    G.hc = ldg(C + (dn ? ldc : 0u), P.hidx);
    G.eu = ldg(u, P.eidx);
    return G;
}
static_assert(offsetof(BlockDesc2, type) == 4 && offsetof(BlockDesc2, q) == 84, "sweep_prefetch reads type/q by offset");

// `nb` blocks at positions blk0, blk0 + stride, ... (block indices, or entries of `blist`) by this wave; the
// lane-only index arithmetic is shared by all of them and the loads of the next blocks are in flight while a
// block is computed
// STEP: the explicit update u + dt * residual is stored instead of the residual (ibh_step_advection)
template <bool DT, bool STEP = false>
__device__ __forceinline__ void sweep_adv(const BlockDesc2* __restrict__ blocks, const int32_t* __restrict__ htab,
                                          const int32_t* __restrict__ etab, const int32_t* __restrict__ dtab,
                                          const int32_t* __restrict__ blist,
                                          int32_t blk0, int32_t stride, int32_t nb,
                                          const float* __restrict__ u, const float* __restrict__ C, uint32_t ldc,
                                          float* __restrict__ ud, float* lds, int lane, float dt = 0.0f) {
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

    // two-stage pipeline: while block k is computed, the gathers of block k+1 and the tables of block k+2 are in flight
    auto at = [&](int32_t pos) { return blist ? blist[pos] : pos; };
    SweepPre T1 = sweep_prefetch<DT>(blocks, htab, etab, dtab, at(blk0), u, C, ldc, lane);
    SweepGat G1 = sweep_gather<DT>(T1, delta, dn != 0, u, C, ldc);
    SweepPre T2 = T1;
    if (nb > 1) T2 = sweep_prefetch<DT>(blocks, htab, etab, dtab, at(blk0 + stride), u, C, ldc, lane);
    for (int32_t it = 0; it < nb; ++it) {
        const SweepPre P = T1;
        const SweepGat G = G1;
        if (it + 1 < nb) {
            T1 = T2;
            G1 = sweep_gather<DT>(T1, delta, dn != 0, u, C, ldc);
        }
        if (it + 2 < nb) T2 = sweep_prefetch<DT>(blocks, htab, etab, dtab, at(blk0 + (it + 2) * stride), u, C, ldc, lane);
        const BlockDesc2& bb = P.bb;
        const float uc = P.uc, cxc = P.cxc, cyc = P.cyc;
        const bool mirror = P.ty == SIDE_MIRROR, isC = P.ty == SIDE_COARSE, isF = P.ty == SIDE_FINE;
        const float hu = G.hu, hdeep = G.hdeep, hc = G.hc, eu = G.eu;
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
        // ---- own cells: undivided slopes + sensor, x and y together
        float Sx, Sy, Dc;
        {
            const v2f l0 = v2(fU[n0], fU[n2]), l1 = v2(fU[n0b], fU[n2b]);
            const v2f r0 = v2(fU[n1], fU[n3]), r1 = v2(fU[n1b], fU[n3b]);
            const v2f c = v2(uc);
            const v2f dR = 0.5f * (r0 + r1) - c, dL = c - 0.5f * (l0 + l1);
            const v2f S = v2(q1, q3) * dR + v2(q0, q2) * dL;
            const v2f nu = sensor2(uc, l0, l1, r0, r1, v2(bb.rh[0], bb.rh[1]));
            Sx = S.x;
            Sy = S.y;
            Dc = fmaxf(fmaxf(nu.x, nu.y), 1e-7f);
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
            const int mask = isF ? 15 : isC ? 12 : 14;
            const int pm = p & mask, w = 16 - mask, e0i = pm + 2;
            const bool single = mask == 15;
            const int lo0 = pm == 0 ? 0 : e0i - w;
            const int lo1 = lo0 + ((pm == 0 || !single) ? 1 : 0);
            const int hi0 = e0i + w;
            const int hi1 = hi0 + ((hi0 == 18 || !single) ? 1 : 0);
            // normal and tangential direction together
            const v2f nu = sensor2(hu, v2(m0, el[lo0]), v2(m1, el[lo1]), v2(hdeep, el[hi0]), v2(hdeep, el[hi1]),
                                   v2(ihn, iht));
            Dh = fmaxf(fmaxf(nu.x, nu.y), 1e-7f);
        }
        // MIRROR sides (domain boundary: a handful of blocks) take slope and sensor of the boundary cell itself
        if (bb.type[0] == SIDE_MIRROR || bb.type[1] == SIDE_MIRROR || bb.type[2] == SIDE_MIRROR ||
            bb.type[3] == SIDE_MIRROR) {  // wave-uniform
            wave_lds_sync();
            const float Sm = fSn[pos0], Dm = fD[pos0];
            Sh = mirror ? Sm : Sh;
            Dh = mirror ? Dm : Dh;
        }
        fSX[64 + lane] = Sh;
        fSY[64 + lane] = Sh;
        fD[64 + lane] = Dh;
        wave_lds_sync();
        // ---- fluxes: right (x+) and top (y+) face of every cell, sub-face 0 on block sides
        const v2f F2 = flux_w2(v2(uc), v2(fU[n1], fU[n3]), v2(Sx, Sy), v2(fSX[n1], fSY[n3]), v2(Dc), v2(fD[n1], fD[n3]),
                               v2(cxc, cyc), v2(fCX[n1], fCY[n3]), v2(q1, q3));
        float FR = F2.x, FT = F2.y;
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
        const float res = -((FR - FL) * bb.rh[0]) - ((FT - FB) * bb.rh[1]);
        stg(ud, (uint32_t)bb.base + lane, STEP ? uc + dt * res : res);
    }
}

// ------------------------------------------------------------------------------------------
// Single-kernel Euler sweep (R2: P = [p T u v], MUSCL(high_order) with the pressure sensor, HLL, Green-Gauss;
// cfd.jl:459-508 over ImmersedBoundary.jl:1077-1157): the scheme of sweep_adv with four variables.  Slopes of the
// halo cells need the value one step deeper for every variable; the sensor (pressure only) also the lateral
// neighbours / end table.  HLL combine in Float32 like the tuned two-kernel form.
// LDS per wave (floats): P 4x128 | D 128 | SX 4x128 | SY 4x128 | ex 4x64 | ext 80
// ------------------------------------------------------------------------------------------
#define BLK2_SWEEP_EULER_LDS (4 * 128 + 128 + 8 * 128 + 4 * 64 + 80)

// conserved state, pressure, normal velocity and speed of sound of one side of a face
__device__ __forceinline__ void euler_state(const float* P, int dn, const Gas& gas, float* Q, float& p, float& un, float& a) {
    p = P[0];
    const float T = fmaxf(P[1], 10.0f);
    const float k = 0.5f * (P[2] * P[2] + P[3] * P[3]);
    const float rho = p * __builtin_amdgcn_rcpf(gas.R * T);
    Q[0] = rho;
    Q[1] = rho * (gas.R / (gas.gamma - 1.0f) * T + k);
    Q[2] = rho * P[2];
    Q[3] = rho * P[3];
    un = dn ? P[3] : P[2];
    a = __builtin_amdgcn_sqrtf(gas.gamma * gas.R * T);
}

// density, energy per mass, pressure, normal velocity and speed of sound of one side of a face (quad2::euler_side_pm2)
__device__ __forceinline__ void euler_side_pm(const float* P, int dn, const Gas& gas, float& rho, float& e, float& p, float& un,
                                              float& a) {
    p = P[0];
    const float T = fmaxf(P[1], 10.0f);
    const float q = P[2] * P[2] + P[3] * P[3];
    rho = p * __builtin_amdgcn_rcpf(gas.R * T);
    e = (gas.R / (gas.gamma - 1.0f)) * T + 0.5f * q;
    un = dn ? P[3] : P[2];
    a = __builtin_amdgcn_sqrtf((gas.gamma * gas.R) * T);
}

// MUSCL states from undivided slopes (see flux_w), then the HLL flux of blk2::euler_flux
__device__ __forceinline__ void euler_flux_w(const float* Pa, const float* Pb, const float* Sa, const float* Sb, float Da,
                                             float Db, float wa, int dn, const Gas& gas, float* F) {
    float PL[4], PR[4];
    const float Df = fmaxf(fmaxf(Da, Db), 1e-7f);
    const float wb = 1.0f - wa;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const float d = Pb[v] - Pa[v];
        const float gu = Sa[v] - d * wa;
        const float Du = Sb[v] - d * wb;
        const float s = __builtin_amdgcn_fmed3f(Du, gu, 0.0f);
        const float t16 = (Sa[v] - Sb[v]) * 0.0625f;
        const float uf = (Pa[v] + wa * d) + t16;
        PL[v] = uf + Df * ((s - wa * d) - t16);   // (Pa + s) - uf
        PR[v] = PL[v] + Df * (d - 2.0f * s);      // uf + Df ((wb d - s) - t16) = PL + Df (d - 2 s)
    }
    // HLL regrouped by state, operation for operation what quad2::euler_flux_w2 does on pairs (ibh_quad2d_euler.h): a block
    // gives the same bits whether a quad wave or a single-block wave sweeps it.  Q = rho (1, e, u, v) is never formed:
    // F = AL (1, eL, uL, vL) + AR (1, eR, uR, vR), A = rho c
    float rL, eL, pL, uL, aL, rR, eR, pR, uR, aR;
    euler_side_pm(PL, dn, gas, rL, eL, pL, uL, aL);
    euler_side_pm(PR, dn, gas, rR, eR, pR, uR, aR);
    const float SR = fminf(uR - aR, 0.0f);
    const float SL = fmaxf(uL + aL, 0.0f);
    const float rs = __builtin_amdgcn_rcpf(SL - SR);
    const float wL = SL * rs, wR = SR * rs;
    const float c = SL * wR;
    const float AL = rL * (wL * uL - c), AR = rR * (c - wR * uR);
    F[0] = AL + AR;
    F[1] = AL * eL + AR * eR;
    F[2] = AL * PL[2] + AR * PR[2];
    F[3] = AL * PL[3] + AR * PR[3];
    const float mL = wL * pL, mR = wR * pR;
    const float m = mL - mR;
    F[2] += dn ? 0.0f : m;
    F[3] += dn ? m : 0.0f;
    F[1] += mL * uL - mR * uR;
}

struct SweepPreE {
    BlockDesc2 bb;
    uint32_t hidx, eidx;
    float Pc[4];
    int ty;
    float qs;
    uint32_t dix;
};
__device__ __forceinline__ SweepPreE sweep_prefetch_e(const BlockDesc2* __restrict__ blocks,
                                                      const int32_t* __restrict__ htab,
                                                      const int32_t* __restrict__ etab,
                                                      const int32_t* __restrict__ dtab, int32_t blk,
                                                      const float* __restrict__ P, uint32_t ldp, int lane) {
    SweepPreE T;
    T.bb = blocks[blk];
    T.hidx = (uint32_t)htab[(size_t)blk * 64 + lane];
    T.eidx = (uint32_t)etab[(size_t)blk * 16 + (lane & 15)];
    T.dix = 0;
    if (T.bb.dt >= 0) T.dix = (uint32_t)dtab[(size_t)T.bb.dt * 64 + lane];
    const int32_t* bw = (const int32_t*)(blocks + blk);
    T.ty = bw[1 + (lane >> 4)];
    T.qs = __int_as_float(bw[21 + (lane >> 4)]);
    const uint32_t c = (uint32_t)T.bb.base + lane;
#pragma unroll
    for (int v = 0; v < 4; ++v) T.Pc[v] = ldg(P + (size_t)v * ldp, c);
    return T;
}

__device__ __forceinline__ void sweep_euler(const BlockDesc2* __restrict__ blocks, const int32_t* __restrict__ htab,
                                            const int32_t* __restrict__ etab, const int32_t* __restrict__ dtab,
                                            const int32_t* __restrict__ blist, int32_t blk0, int32_t stride, int32_t nb,
                                            const float* __restrict__ P, uint32_t ldp, float* __restrict__ Rr,
                                            uint32_t ldr, Gas gas, float* lds, int lane) {
    float* fP = lds;           // [4][128]
    float* fD = lds + 512;     // [128]
    float* fSX = lds + 640;    // [4][128]
    float* fSY = lds + 1152;   // [4][128]
    float* ex = lds + 1664;    // [4][64]
    float* ext = lds + 1920;   // [80]
    // ---- lane-only geometry (as in sweep_adv)
    const int s = lane >> 4, dn = lane >> 5, p = lane & 15, t = p >> 1;
    const int delta = seli4(s, -1, 1, -8, 8);
    const int pos0 = (dn ? t : 8 * t) + ((s & 1) ? (dn ? 56 : 7) : 0);
    const int posx = pos0 ^ (dn ? 1 : 8);
    float* extw = ext + s * 20 + 2 + p;
    float* exte = ext + ((lane >> 2) & 3) * 20 + ((lane >> 1) & 1) * 18 + (lane & 1);
    const float* el = ext + s * 20;
    const float* fSn = dn ? fSY : fSX;
    const int i = lane & 7, j = lane >> 3;
    const bool e0 = i == 0, e1 = i == 7, e2 = j == 0, e3 = j == 7;
    const int n0 = e0 ? 64 + j * 2 : lane - 1;
    const int n1 = e1 ? 80 + j * 2 : lane + 1;
    const int n2 = e2 ? 96 + i * 2 : lane - 8;
    const int n3 = e3 ? 112 + i * 2 : lane + 8;
    const int n0b = n0 + (e0 ? 1 : 0), n1b = n1 + (e1 ? 1 : 0), n2b = n2 + (e2 ? 1 : 0), n3b = n3 + (e3 ? 1 : 0);
    const int xg = (lane >> 3) & 3, xt = lane & 7, xd = xg & 1;
    const int xpos = xd ? xt : 8 * xt;
    const int xslot = 64 + (xd * 16 + xt) * 2 + (xg >> 1);
    const float* xS = xd ? fSY : fSX;

    auto at = [&](int32_t pos) { return blist ? blist[pos] : pos; };
    SweepPreE N = sweep_prefetch_e(blocks, htab, etab, dtab, at(blk0), P, ldp, lane);
    for (int32_t it = 0; it < nb; ++it) {
        const SweepPreE T = N;
        if (it + 1 < nb) N = sweep_prefetch_e(blocks, htab, etab, dtab, at(blk0 + (it + 1) * stride), P, ldp, lane);
        const BlockDesc2& bb = T.bb;
        const bool mirror = T.ty == SIDE_MIRROR, isC = T.ty == SIDE_COARSE, isF = T.ty == SIDE_FINE;
        const uint32_t didx = bb.dt >= 0 ? T.dix : (mirror ? T.hidx : T.hidx + (uint32_t)delta);
        float hP[4], hdeep[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            hP[v] = ldg(P + (size_t)v * ldp, T.hidx);
            hdeep[v] = ldg(P + (size_t)v * ldp, didx);
        }
        const float eu = ldg(P, T.eidx);  // pressure across the ends of the sides (sensor only)
        if (it) wave_lds_sync();
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            fP[v * 128 + lane] = T.Pc[v];
            fP[v * 128 + 64 + lane] = hP[v];
        }
        *extw = hP[0];
        *exte = eu;
        const float q0 = e0 ? bb.q[0] : 0.5f, q1 = e1 ? bb.q[1] : 0.5f;
        const float q2 = e2 ? bb.q[2] : 0.5f, q3 = e3 ? bb.q[3] : 0.5f;
        wave_lds_sync();
        // ---- own cells: undivided slopes of the four primitives, pressure sensor
        float Sx[4], Sy[4], Dc;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float* f = fP + v * 128;
            const v2f l0 = v2(f[n0], f[n2]), l1 = v2(f[n0b], f[n2b]);
            const v2f r0 = v2(f[n1], f[n3]), r1 = v2(f[n1b], f[n3b]);
            const v2f c = v2(T.Pc[v]);
            const v2f dR = 0.5f * (r0 + r1) - c, dL = c - 0.5f * (l0 + l1);
            const v2f S = v2(q1, q3) * dR + v2(q0, q2) * dL;
            Sx[v] = S.x;
            Sy[v] = S.y;
            fSX[v * 128 + lane] = S.x;
            fSY[v * 128 + lane] = S.y;
            if (v == 0) {
                const v2f nu = sensor2(T.Pc[0], l0, l1, r0, r1, v2(bb.rh[0], bb.rh[1]));
                Dc = fmaxf(fmaxf(nu.x, nu.y), 1e-7f);
            }
        }
        fD[lane] = Dc;
        // ---- halo cells
        float Sh[4], Dh;
        {
            const int pm1 = isC ? posx : pos0;
            const float omq = 1.0f - T.qs;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float m0 = fP[v * 128 + pos0], m1 = fP[v * 128 + pm1];
                const float din = 0.5f * (m0 + m1) - hP[v];
                const float dde = hdeep[v] - hP[v];
                const float x = omq * din - 0.5f * dde;
                Sh[v] = (s & 1) ? -x : x;
                if (v == 0) {
                    const float irt = isC ? 0.5f : isF ? 2.0f : 1.0f;
                    const float ihn = (dn ? bb.rh[1] : bb.rh[0]) * irt;
                    const float iht = (dn ? bb.rh[0] : bb.rh[1]) * irt;
                    const int mask = isF ? 15 : isC ? 12 : 14;
                    const int pm = p & mask, w = 16 - mask, e0i = pm + 2;
                    const bool single = mask == 15;
                    const int lo0 = pm == 0 ? 0 : e0i - w;
                    const int lo1 = lo0 + ((pm == 0 || !single) ? 1 : 0);
                    const int hi0 = e0i + w;
                    const int hi1 = hi0 + ((hi0 == 18 || !single) ? 1 : 0);
                    const v2f nu = sensor2(hP[0], v2(m0, el[lo0]), v2(m1, el[lo1]), v2(hdeep[0], el[hi0]),
                                           v2(hdeep[0], el[hi1]), v2(ihn, iht));
                    Dh = fmaxf(fmaxf(nu.x, nu.y), 1e-7f);
                }
            }
        }
        wave_lds_sync();
        {
            const float Dm = fD[pos0];
            Dh = mirror ? Dm : Dh;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float Sm = fSn[v * 128 + pos0];
                Sh[v] = mirror ? Sm : Sh[v];
            }
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            fSX[v * 128 + 64 + lane] = Sh[v];
            fSY[v * 128 + 64 + lane] = Sh[v];
        }
        fD[64 + lane] = Dh;
        wave_lds_sync();
        // ---- fluxes: right (x+) and top (y+) face of every cell
        float FR[4], FT[4], FR1[4], FT1[4];
        {
            float Pb[4], Sb[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                Pb[v] = fP[v * 128 + n1];
                Sb[v] = fSX[v * 128 + n1];
            }
            euler_flux_w(T.Pc, Pb, Sx, Sb, Dc, fD[n1], q1, 0, gas, FR);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                Pb[v] = fP[v * 128 + n3];
                Sb[v] = fSY[v * 128 + n3];
            }
            euler_flux_w(T.Pc, Pb, Sy, Sb, Dc, fD[n3], q3, 1, gas, FT);
        }
        {   // low sides: the halo cell is the owner, this block's cell the neighbour
            float Pa[4], Pb[4], Sa[4], Sb[4], X[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                Pa[v] = fP[v * 128 + xslot];
                Pb[v] = fP[v * 128 + xpos];
                Sa[v] = xS[v * 128 + xslot];
                Sb[v] = xS[v * 128 + xpos];
            }
            euler_flux_w(Pa, Pb, Sa, Sb, fD[xslot], fD[xpos], 1.0f - (xd ? bb.q[2] : bb.q[0]), xd, gas, X);
#pragma unroll
            for (int v = 0; v < 4; ++v) ex[v * 64 + lane] = X[v];
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            FR1[v] = FR[v];
            FT1[v] = FT[v];
        }
        if (bb.type[1] == SIDE_FINE) {  // second sub-faces of the HIGH sides: wave-uniform
            float Pb[4], Sb[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                Pb[v] = fP[v * 128 + n1 + 1];
                Sb[v] = fSX[v * 128 + n1 + 1];
            }
            euler_flux_w(T.Pc, Pb, Sx, Sb, Dc, fD[n1 + 1], bb.q[1], 0, gas, FR1);
        }
        if (bb.type[3] == SIDE_FINE) {
            float Pb[4], Sb[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                Pb[v] = fP[v * 128 + n3 + 1];
                Sb[v] = fSY[v * 128 + n3 + 1];
            }
            euler_flux_w(T.Pc, Pb, Sy, Sb, Dc, fD[n3 + 1], bb.q[3], 1, gas, FT1);
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
            const float eL = 0.5f * (e[j] + e[16 + j]), eB = 0.5f * (e[8 + i] + e[24 + i]);
            const float fl = e0 ? eL : FLs[v];
            const float fb = e2 ? eB : FBs[v];
            const float fr = e1 ? 0.5f * (FR[v] + FR1[v]) : FR[v];
            const float ft = e3 ? 0.5f * (FT[v] + FT1[v]) : FT[v];
            stg(Rr + (size_t)v * ldr, (uint32_t)bb.base + lane, -((fr - fl) * bb.rh[0]) - ((ft - fb) * bb.rh[1]));
        }
    }
}

#pragma clang fp contract(off)

}  // namespace blk2
