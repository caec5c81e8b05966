// Quad sweep (2-D scalar advection-JST-MUSCL residual, test/advection.jl:67-83): ONE wavefront sweeps the 16x16 tile
// of four sibling 8x8 blocks, 4 cells per lane.  Same arithmetic as blk2::sweep_adv (ibh_sweep2d.h) -- undivided
// slopes, flux_w -- but the 64 halo cells of the tile are 1/4 of its cells instead of as many as the block's own,
// and the neighbours of a cell come from registers, DPP row shifts and a few cross-row permutes instead of LDS
// tiles.  tests/quad_model.py is the lane-by-lane numpy statement of this file, checked against the oracle.
//
// Lane L = 16*g + t.
//   own cells:  x = 4*g + r (r = 0..3: one float4 per field), y = t.
//     y neighbours: lane -/+ 1 of the same 16-lane row  -> DPP row_shr:1 / row_shl:1 (lanes t = 0 / 15 keep `old`,
//                   which is preloaded with the bottom / top ring of the tile);
//     x neighbours: r -/+ 1 in registers, across strips lane -/+ 16 -> ds_bpermute (no LDS memory involved).
//   halo slots: the two sub-faces k = 0, 1 of boundary cell t of side [left, bottom, top, right][g] (arithmetic on
//     (k0, k1) pairs); left / right halo values are therefore already in the lanes that own the boundary cells,
//     bottom / top values go through a 16-entry LDS ring to the lanes t = 0 / 15.
//   edge faces: the same lane computes both sub-faces of its boundary cell and hands their mean to the cell's lane.
// A face between cell a (towards -) and cell b (towards +) seen from the other side: F(a,b) = -F(b,a) with slopes
// and velocities negated (every term of flux_w is odd or even under that swap), so low and high sides share one
// evaluation: own cell first, halo cell second, sign = +1 on top / right, -1 on left / bottom.
#pragma once
#include "ibh_sweep2d.h"

namespace quad2 {

#pragma clang fp contract(fast)

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
// float4 in global memory at any 4-byte aligned address: a quad's first cell is a multiple of 4 only on partitions
// without skirt fragments in front of it (global_load/store_dwordx4 need dword alignment only)
typedef float v4f_g __attribute__((ext_vector_type(4), aligned(4)));

// LDS per wave (floats): tile U | tile SY | tile D | tile CY (16 rows, pitch 20: conflict-free ds_write_b128) |
// lateral lines ext[8][20] | rings M, F, Q [4 rows][16] (row 0 bottom, 1 top, 2 neutral, 3 dump) | edge fluxes [4][16]
#define QUAD_PITCH 20
#define QUAD_TILE (16 * QUAD_PITCH)
#define QUAD_OFF_EXT (4 * QUAD_TILE)
#define QUAD_OFF_RING (QUAD_OFF_EXT + 160)
#define QUAD_OFF_EX (QUAD_OFF_RING + 3 * 64)
#define QUAD_LDS (QUAD_OFF_EX + 64)

__device__ __forceinline__ float dpp_shr1(float old, float v) {  // lane i <- lane i-1 in rows of 16; lane 0 keeps old
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x111, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_shl1(float old, float v) {  // lane i <- lane i+1; lane 15 keeps old
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x101, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_shl1z(float v) {  // lane i <- lane i+1; lane 15 gets 0
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x101, 0xf, 0xf, true));
}
__device__ __forceinline__ float bperm(int byte_addr, float v) {  // value of lane byte_addr/4
    return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_addr, __float_as_int(v)));
}
__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
// an LDS read every lane performs: keeps the compiler from sinking it into a divergent branch of a later select
__device__ __forceinline__ float lds_read(const float* p) {
    float v = *p;
    asm volatile("" : "+v"(v));
    return v;
}

// interior faces (both cells same level, wa = 1/2), four at a time
__device__ __forceinline__ v4f flux_half4(v4f ua, v4f ub, v4f Sa, v4f Sb, v4f Da, v4f Db, v4f Ca, v4f Cb) {
    const v4f d = ub - ua;
    const v4f gu = Sa - 0.5f * d;
    const v4f Du = Sb - 0.5f * d;
    v4f s, Df;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        s[c] = __builtin_amdgcn_fmed3f(Du[c], gu[c], 0.0f);
        Df[c] = max3(Da[c], Db[c], 1e-7f);
    }
    const v4f t16 = (Sa - Sb) * 0.0625f;
    const v4f uf = (ua + 0.5f * d) + t16;
    const v4f A = uf - Df * t16;
    const v4f Cf = 0.5f * (Ca + Cb);
    const v4f B = Df * (s - 0.5f * d);
    const v4f CA = Cf * A;
    v4f F;
#pragma unroll
    for (int c = 0; c < 4; ++c) F[c] = fmaf(fabsf(Cf[c]), B[c], CA[c]);
    return F;
}

// edge faces: two sub-faces at a time, a = the quad's boundary cell (broadcast), b = the halo cells
__device__ __forceinline__ v2f flux_w2(float ua, v2f ub, float Sa, v2f Sb, float Da, v2f Db, float Ca, v2f Cb, float wa) {
    const v2f d = ub - ua;
    const v2f gu = Sa - d * wa;
    const v2f Du = Sb - d * (1.0f - wa);
    const v2f s = v2f{__builtin_amdgcn_fmed3f(Du.x, gu.x, 0.0f), __builtin_amdgcn_fmed3f(Du.y, gu.y, 0.0f)};
    const v2f t16 = (Sa - Sb) * 0.0625f;
    const v2f uf = (ua + wa * d) + t16;
    const v2f mu = d * (0.5f - wa) - t16;
    const v2f Df = v2f{max3(Da, Db.x, 1e-7f), max3(Da, Db.y, 1e-7f)};
    const v2f A = uf + Df * mu;
    const v2f Cf = Ca + wa * (Cb - Ca);
    const v2f B = Df * (s - 0.5f * d);
    const v2f CA = Cf * A;
    return v2f{fmaf(fabsf(Cf.x), B.x, CA.x), fmaf(fabsf(Cf.y), B.y, CA.y)};
}

// JST ratio (1e-7 + |g| rh) / (1e-7 + a rh)  (JST_sensor :1077-1097, undivided sums)
__device__ __forceinline__ float jst(float g, float a, float rh) {
    return fmaf(fabsf(g), rh, 1e-7f) * __builtin_amdgcn_rcpf(fmaf(a, rh, 1e-7f));
}
// max(1e-7, ratio of direction 1, ratio of direction 2) with ONE reciprocal (v_rcp_f32 is a quarter-rate instruction:
// it costs four issue slots): n1/d1 vs n2/d2 -> max(n1 d2, n2 d1) / (d1 d2); all four terms are positive
__device__ __forceinline__ float jst_max2(float g1, float a1, float rh1, float g2, float a2, float rh2) {
    const float n1 = fmaf(fabsf(g1), rh1, 1e-7f), d1 = fmaf(a1, rh1, 1e-7f);
    const float n2 = fmaf(fabsf(g2), rh2, 1e-7f), d2 = fmaf(a2, rh2, 1e-7f);
    return fmaxf(fmaxf(n1 * d2, n2 * d1) * __builtin_amdgcn_rcpf(d1 * d2), 1e-7f);
}

// Lane table of the quad sweep, built at compile time: per lane 8 words, LDS byte offsets packed two per word (see
// sweep_quad for the meaning of the fields), the lane's cell offset inside the quad and its deeper-cell step.
// HALF: the tile of a PAIR of blocks (16 x 8 cells: lower-left, lower-right), swept by the same wave code -- lanes t >= 8
// own no cells (they still own the slots of bottom / top boundary cells x = t), the top side sits on row 7
template <bool HALF>
struct QuadLaneTab {
    uint32_t w[64][8];
    constexpr QuadLaneTab() : w() {
        for (int lane = 0; lane < 64; ++lane) {
            const int g = lane >> 4, t = lane & 15, tl = lane & 7, lhs = lane >> 3;
            const int ytop = HALF ? 7 : 15;
            const bool g0 = g == 0, g3 = g == 3, lr = g0 || g3, t0 = t == 0, t15 = t == ytop;
            const int bx = g0 ? 0 : g3 ? 15 : t, by = lr ? t : g == 1 ? 0 : ytop;
            const int pos_b = by * QUAD_PITCH + bx;
            const int pos_b1 = lr ? (t ^ 1) * QUAD_PITCH + bx : by * QUAD_PITCH + (t ^ 1);
            const int pos_own = t * QUAD_PITCH + 4 * g;
            const int myrow = t0 ? 0 : t15 ? 1 : 2, rowB = t0 ? 0 : 2, rowT = t15 ? 1 : 2;
            const int wrow = g == 1 ? 0 : g == 2 ? 1 : 3;
            const int eline = (lane & 31) >> 2, ee = lane & 3;
            const int epos = QUAD_OFF_EXT + eline * 20 + (ee < 2 ? ee : 16 + ee);
            const int line = QUAD_OFF_EXT + lhs * 20;
            const uint32_t f[14] = {
                (uint32_t)pos_own, (uint32_t)pos_b, (uint32_t)pos_b1, (uint32_t)(line + 2 + 2 * tl), (uint32_t)epos,
                (uint32_t)(QUAD_OFF_RING + wrow * 16 + t), (uint32_t)(QUAD_OFF_RING + myrow * 16 + 4 * g),
                (uint32_t)(QUAD_OFF_RING + rowB * 16 + 4 * g), (uint32_t)(QUAD_OFF_RING + rowT * 16 + 4 * g),
                (uint32_t)(line + 2 * tl), (uint32_t)(line + 2 * (tl & ~1)), (uint32_t)(line + 2 * (tl | 1) + 4),
                0u, 0u};
            for (int k = 0; k < 6; ++k) w[lane][k] = (4u * f[2 * k]) | ((4u * f[2 * k + 1]) << 16);
            w[lane][6] = 64u * ((g >> 1) + 2 * (t >> 3)) + 4u * (g & 1) + 8u * (t & 7);
            w[lane][7] = (uint32_t)(g0 ? -1 : g == 1 ? -8 : g == 2 ? 8 : 1);
        }
    }
};
__device__ const QuadLaneTab<false> lane_tab = QuadLaneTab<false>();
__device__ const QuadLaneTab<true> lane_tab_half = QuadLaneTab<true>();

// ---- the wave's lane-only state: LDS pointers from a compile-time table (two 16-byte loads; computing them costs
// ~90 vector instructions per wave, a fifth of a quad's arithmetic), lane classes from the lane id
struct QuadLane {
    float *p_own;    // this lane's four cells in a tile
    float *p_b;      // boundary cell of this lane's halo slots (in tile U)
    float *p_b1;     // its pair mate t ^ 1 (COARSE sides)
    float *p_extw;   // this lane's two slots in the lateral line of its half-side
    float *p_epos;   // where lane & 31 puts its end value
    float *p_ringw;  // ring entry this lane's side writes (dump row for left / right)
    float *p_ringm;  // ring row this lane reads: bottom (t = 0), top (t = 15), neutral
    float *p_ringB;  // weight row for the low y side: bottom or neutral
    float *p_ringT;  // ... high y side: top or neutral
    float *p_LbS;    // low lateral pair of this lane's slots, SAME / FINE half-sides (high pair: + 4)
    float *p_LbC;    // ... COARSE half-sides
    float *p_HbC;    // high lateral pair, COARSE
    uint32_t a0off;  // this lane's first cell relative to the quad's
    int delta;       // deeper cell of a halo cell: -1, -8, +8, +1
    int lane, tl, lhs;
    bool g0, g3, lr, dny, t15;   // t15: this lane's cells are the top row of the tile (t = 15; pair tiles: t = 7)
    bool active;     // this lane owns cells (pair tiles: t < 8)
    float sgn;       // +1: the quad's cell is the owner of the edge face (top / right), -1 on left / bottom
};

template <bool HALF = false>
__device__ __forceinline__ QuadLane quad_lane(float* lds, int lane) {
    QuadLane G;
    char* const L = (char*)lds;
    const uint32_t* const tab = HALF ? lane_tab_half.w[lane] : lane_tab.w[lane];
    const uint4 w0 = *(const uint4*)(tab);
    const uint4 w1 = *(const uint4*)(tab + 4);
#define QL_LO(w) ((w) & 0xffffu)
#define QL_HI(w) ((w) >> 16)
    G.p_own = (float*)(L + QL_LO(w0.x));
    G.p_b = (float*)(L + QL_HI(w0.x));
    G.p_b1 = (float*)(L + QL_LO(w0.y));
    G.p_extw = (float*)(L + QL_HI(w0.y));
    G.p_epos = (float*)(L + QL_LO(w0.z));
    G.p_ringw = (float*)(L + QL_HI(w0.z));
    G.p_ringm = (float*)(L + QL_LO(w0.w));
    G.p_ringB = (float*)(L + QL_HI(w0.w));
    G.p_ringT = (float*)(L + QL_LO(w1.x));
    G.p_LbS = (float*)(L + QL_HI(w1.x));
    G.p_LbC = (float*)(L + QL_LO(w1.y));
    G.p_HbC = (float*)(L + QL_HI(w1.y));
#undef QL_LO
#undef QL_HI
    // from the lane id (the own-cell loads then wait for the descriptor only):
    // 64 ((g >> 1) + 2 (t >> 3)) + 4 (g & 1) + 8 (t & 7)
    // (pair tiles: lanes t >= 8 repeat the cells of t - 8 -- valid loads, nothing stored)
    G.a0off = ((lane & 7) << 3) | (HALF ? 0 : ((lane & 8) << 4)) | ((lane & 16) >> 2) | ((lane & 32) << 1);
    G.delta = (int)w1.w;
    G.lane = lane;
    G.tl = lane & 7;
    G.lhs = lane >> 3;  // outer half-side of this lane's slots
    G.g0 = lane < 16;
    G.g3 = lane >= 48;
    G.lr = G.g0 || G.g3;
    G.dny = !G.lr;
    G.t15 = (lane & 15) == (HALF ? 7 : 15);
    G.active = !HALF || (lane & 8) == 0;
    G.sgn = lane >= 32 ? 1.0f : -1.0f;
    // neutral ring rows: u ring unused, sensor correction 0, weight 1/2 (read by lanes that are not on the y edges)
    ((float*)(L + 4 * (QUAD_OFF_RING + 64 + 32)))[lane & 15] = 0.0f;
    ((float*)(L + 4 * (QUAD_OFF_RING + 128 + 32)))[lane & 15] = 0.5f;
    return G;
}

// ---- what a wave fetches for a quad, in three dependent steps
struct QuadTab {   // needs the quad's index only
    QuadDesc2 d;   // wave-uniform
    v2i hid;       // halo cells of this lane's two slots
    uint32_t eid;  // end cell of lane & 31
};
struct QuadOwn {   // needs the descriptor
    v4f U, CX, CY;
    uint32_t a0;
};
struct QuadHalo {  // needs the table entries
    v2f hu, hd, hc;
    float eu;
};
// qaux (or null): the compact companion rows (IBH_QAUX, ibh_common.h) -- half-sides with arithmetic halo ids are computed
// from one origin id, only FINE half-sides (and the rare others) read their part of the 640-byte row
__device__ __forceinline__ QuadTab quad_load_tab(const QuadDesc2* __restrict__ qd, const int32_t* __restrict__ qtab,
                                                 int32_t q, int lane, const int32_t* __restrict__ qaux = nullptr) {
    QuadTab T;
    T.d = qd[q];
    const int32_t* row = qtab + (size_t)q * IBH_QROW;
    if (qaux) {
        const int32_t* aux = qaux + (size_t)q * IBH_QAUX;
        const int lhs = lane >> 3, tl = lane & 7;
        const int32_t orig = aux[32 + lhs];
        T.eid = (uint32_t)aux[lane & 31];
        const bool isC = ((T.d.cls >> (4 * lhs)) & 15u) == SIDE_COARSE;
        const bool lr = lane < 16 || lane >= 48;
        const int32_t id = orig + (lr ? 8 : 1) * (isC ? tl >> 1 : tl);
        T.hid = v2i{id, id};
        if (orig < 0) T.hid = *(const v2i*)(row + 2 * lane);
        return T;
    }
    T.hid = *(const v2i*)(row + 2 * lane);
    T.eid = (uint32_t)row[128 + (lane & 31)];
    return T;
}
__device__ __forceinline__ QuadOwn quad_load_own(const QuadLane& G, const QuadTab& T, const float* __restrict__ u,
                                                 const float* __restrict__ C, uint32_t ldc) {
    QuadOwn O;
    O.a0 = (uint32_t)T.d.base + G.a0off;
    O.U = *(const v4f_g*)((const char*)u + ((size_t)O.a0 << 2));
    O.CX = *(const v4f_g*)((const char*)C + ((size_t)O.a0 << 2));
    O.CY = *(const v4f_g*)((const char*)(C + ldc) + ((size_t)O.a0 << 2));
    return O;
}
// Halo gathers: seven per quad.  Measured (scripts/probe_sweep.py, profiles/r2_final/README.md): alone each costs
// about as much as a float4 load of the whole tile; inside the sweep, dropping ALL second-slot gathers is worth 0.35 us
// of 4.95 (wrong on FINE half-sides), but every exact way of skipping them where the second slot names the same cell
// as the first -- a wave-uniform branch, a second code path for quads without a FINE side, exec-masked loads, a paired
// 8-byte load of halo + deeper cell -- measured equal or slower than gathering everything: the sweep ends with its
// slowest waves, and those are the quads with a FINE side either way.
// GM: which of the seven gathers are performed (127 = all; anything else: measurement only, wrong results)
template <int GM = 127>
__device__ __forceinline__ QuadHalo quad_load_halo(const QuadLane& G, const QuadTab& T, const float* __restrict__ u,
                                                   const float* __restrict__ C, uint32_t ldc) {
    using blk2::ldg;
    QuadHalo H;
    const float* Cn = C + (G.dny ? ldc : 0u);
    const float z = (float)(T.hid.x + T.hid.y + (int)T.eid) * 1e-30f;  // keeps the table loads alive
    H.hu.x = (GM & 1) ? ldg(u, (uint32_t)T.hid.x) : z;
    H.hu.y = (GM & 2) ? ldg(u, (uint32_t)T.hid.y) : H.hu.x;
    H.hd.x = (GM & 4) ? ldg(u, (uint32_t)(T.hid.x + G.delta)) : z;
    H.hd.y = (GM & 8) ? ldg(u, (uint32_t)(T.hid.y + G.delta)) : H.hd.x;
    H.hc.x = (GM & 16) ? ldg(Cn, (uint32_t)T.hid.x) : 1.0f + z;
    H.hc.y = (GM & 32) ? ldg(Cn, (uint32_t)T.hid.y) : H.hc.x;
    H.eu = (GM & 64) ? ldg(u, T.eid) : z;
    return H;
}

// The same seven values with FOUR full gathers and one that only the lanes of FINE left / right half-sides issue:
// every gather is 8 bytes per lane.  Left / right lanes: the halo cell and the cell one step deeper are x neighbours
// (one pair per sub-face slot).  Bottom / top lanes: the two sub-face cells of a FINE half-side are x neighbours (the
// builder checks it: ibh_build_quads2), so (slot 0, slot 1) of the halo cells, of the deeper cells and of the velocity
// come as pairs; on SAME / COARSE half-sides slot 1 repeats slot 0 and the second element is not used.
__device__ __forceinline__ QuadHalo quad_load_halo_paired(const QuadLane& G, const QuadTab& T, const float* __restrict__ u,
                                                          const float* __restrict__ C, uint32_t ldc) {
    using blk2::ldg;
    typedef float v2f_g __attribute__((ext_vector_type(2), aligned(4)));
    auto pair = [](const float* p, int i) { return *(const v2f_g*)((const char*)p + ((size_t)(uint32_t)i << 2)); };
    QuadHalo H;
    const float* Cn = C + (G.dny ? ldc : 0u);
    const bool isF = ((T.d.cls >> (4 * G.lhs)) & 15u) == SIDE_FINE;
    const int lo = G.delta < 0 ? G.delta : 0;                       // left side: the pair starts at the deeper cell
    const v2f PA = pair(u, G.lr ? T.hid.x + lo : T.hid.x);
    const v2f PB = pair(u, G.lr ? T.hid.y + lo : T.hid.x + G.delta);
    const v2f PC = pair(Cn, T.hid.x);
    H.eu = ldg(u, T.eid);
    float hc1 = 0.0f;
    if (G.lr && isF) hc1 = ldg(Cn, (uint32_t)T.hid.y);              // few lanes, few quads
    asm volatile("" : "+v"(hc1));
    const bool first = G.g0;                                         // left side: (deeper, halo); right side: (halo, deeper)
    const float huA = first ? PA.y : PA.x, hdA = first ? PA.x : PA.y;
    const float huB = first ? PB.y : PB.x, hdB = first ? PB.x : PB.y;
    H.hu.x = G.lr ? huA : PA.x;
    H.hd.x = G.lr ? hdA : PB.x;
    H.hu.y = G.lr ? huB : (isF ? PA.y : PA.x);
    H.hd.y = G.lr ? hdB : (isF ? PB.y : PB.x);
    H.hc.x = PC.x;
    H.hc.y = isF ? (G.lr ? hc1 : PC.y) : PC.x;
    return H;
}

// ---- the arithmetic of one quad.  STAMP: phase time stamps of the wave (100 MHz ticks) for scripts/wave_timeline.py
// STEP: the explicit update u + dt * residual is stored instead of the residual (ibh_step_advection)
template <bool STAMP, bool STEP = false, bool HALF = false>
__device__ __forceinline__ void quad_compute(const QuadLane& G, const QuadTab& T, const QuadOwn& O, const QuadHalo& H,
                                             float* __restrict__ ud, unsigned long long* stamps, float dt = 0.0f) {
    using blk2::wave_lds_sync;
    auto stamp = [&](int k) {
        if constexpr (STAMP) {
            __builtin_amdgcn_s_waitcnt(0);
            if (stamps && G.lane == 0) stamps[k] = __builtin_amdgcn_s_memrealtime();
        }
    };
    const v4f U = O.U, CX = O.CX, CY = O.CY;
    const v2f hu = H.hu, hd = H.hd, hc = H.hc;
    const bool g0 = G.g0, g3 = G.g3, lr = G.lr, dny = G.dny, t15 = G.t15;
    const int lane = G.lane;
    const float sgn = G.sgn;
    const uint32_t ty = (T.d.cls >> (4 * G.lhs)) & 15u;
    const bool isC = ty == SIDE_COARSE, isF = ty == SIDE_FINE;
    const float qs = isC ? (1.0f / 3.0f) : isF ? (2.0f / 3.0f) : 0.5f;  // at_faces weight of the quad's cell
    const float irt = isC ? 0.5f : isF ? 2.0f : 1.0f;                   // h / h_halo
    const float rhx = T.d.rh[0], rhy = T.d.rh[1];

    // ---- stage: u tile, lateral lines
    *(v4f*)G.p_own = U;
    *(v4f*)(G.p_own + 3 * QUAD_TILE) = CY;
    *(v2f*)G.p_extw = hu;
    *G.p_epos = H.eu;
    wave_lds_sync();
    stamp(4);
    const float m0 = lds_read(G.p_b);
    const float m1r = lds_read(G.p_b1);
    const float m1 = isC ? m1r : m0;
    const float hm = 0.5f * (hu.x + hu.y);
    // sensor correction of a boundary cell that faces two finer cells: its |d| sum counts both of them
    const float fix = 0.5f * (fabsf(hu.x - m0) + fabsf(hu.y - m0)) - fabsf(hm - m0);
    G.p_ringw[0] = hm;
    G.p_ringw[64] = fix;
    G.p_ringw[128] = qs;
    wave_lds_sync();
    stamp(5);

    // ---- own cells: undivided slopes + sensor
    v4f SX, SY, D;
    v4f uT;
    {
        const float uLm = bperm((lane - 16) << 2, U.w), uRp = bperm((lane + 16) << 2, U.x);
        const v4f UL = v4f{g0 ? hm : uLm, U.x, U.y, U.z};
        const v4f UR = v4f{U.y, U.z, U.w, g3 ? hm : uRp};
        const v4f qL = v4f{g0 ? qs : 0.5f, 0.5f, 0.5f, 0.5f};
        const v4f qR = v4f{0.5f, 0.5f, 0.5f, g3 ? qs : 0.5f};
        const v4f dR = UR - U, dL = U - UL;
        SX = qR * dR + qL * dL;
        const v4f gx = dR - dL;
        const v4f rM = *(const v4f*)G.p_ringm;
        const v4f rF = *(const v4f*)(G.p_ringm + 64);
        const v4f qB = *(const v4f*)(G.p_ringB + 128);
        const v4f qT = *(const v4f*)(G.p_ringT + 128);
        v4f uB;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uB[c] = dpp_shr1(rM[c], U[c]);
            uT[c] = dpp_shl1(rM[c], U[c]);
            if constexpr (HALF) uT[c] = t15 ? rM[c] : uT[c];  // (lane t = 8 exists: the top row takes the ring itself)
        }
        const v4f dT = uT - U, dB = U - uB;
        SY = qT * dT + qB * dB;
        const v4f gy = dT - dB;
        const float fx0 = g0 ? fix : 0.0f, fx3 = g3 ? fix : 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float ax = fabsf(dR[c]) + fabsf(dL[c]);
            if (c == 0) ax += fx0;
            if (c == 3) ax += fx3;
            const float ay = (fabsf(dT[c]) + fabsf(dB[c])) + rF[c];
            D[c] = jst_max2(gx[c], ax, rhx, gy[c], ay, rhy);
        }
    }
    *(v4f*)(G.p_own + QUAD_TILE) = SY;
    *(v4f*)(G.p_own + 2 * QUAD_TILE) = D;

    // ---- halo cells of this lane's two slots: slope along the side normal and sensor
    v2f Shn, Dh;  // Shn: slope of the halo cell seen from the quad outwards (sign folded for the edge flux below)
    {
        const float ihn = (dny ? rhy : rhx) * irt, iht = (dny ? rhx : rhy) * irt;
        const v2f dm0 = m0 - hu, dm1 = m1 - hu, dde = hd - hu;
        const v2f din = 0.5f * (dm0 + dm1);
        // -(1 - qs) din + dde / 2  (= -x of blk2::sweep_adv)
        Shn = 0.5f * dde - (1.0f - qs) * din;
        // lateral neighbours (ibh_sweep2d.h:246-255 in pair form; quad_model.py checks the equivalence)
        const v2f Lp = *(const v2f*)(isC ? G.p_LbC : G.p_LbS);
        const v2f Hp = *(const v2f*)(isC ? G.p_HbC : G.p_LbS + 4);
        const bool t0l = G.tl == 0, t7l = G.tl == 7;
        const v2f lo0 = v2f{(isF && !t0l) ? Lp.y : Lp.x, isF ? hu.x : Lp.x};
        const v2f lo1 = v2f{Lp.y, isF ? hu.x : Lp.y};
        const v2f hi0 = v2f{isF ? hu.y : Hp.x, Hp.x};
        const v2f hi1 = v2f{isF ? hu.y : Hp.y, (isF && !t7l) ? Hp.x : Hp.y};
        const v2f e0 = lo0 - hu, e1 = lo1 - hu, e2 = hi0 - hu, e3 = hi1 - hu;
        const v2f gn = (dm0 + dm1) + (dde + dde);
        const v2f gt = (e0 + e1) + (e2 + e3);
        const float hn = 0.5f * ihn, ht = 0.5f * iht;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float an = (fabsf(dm0[k]) + fabsf(dm1[k])) + 2.0f * fabsf(dde[k]);
            const float at = (fabsf(e0[k]) + fabsf(e1[k])) + (fabsf(e2[k]) + fabsf(e3[k]));
            Dh[k] = jst_max2(gn[k], an, hn, gt[k], at, ht);
        }
    }

    // ---- interior faces: right (x+) and top (y+) face of every cell
    v4f FR, FT;
    {
        const int up = (lane + 16) << 2;
        const v4f Ub = v4f{U.y, U.z, U.w, bperm(up, U.x)};
        const v4f Sb = v4f{SX.y, SX.z, SX.w, bperm(up, SX.x)};
        const v4f Db = v4f{D.y, D.z, D.w, bperm(up, D.x)};
        const v4f Cb = v4f{CX.y, CX.z, CX.w, bperm(up, CX.x)};
        FR = flux_half4(U, Ub, SX, Sb, D, Db, CX, Cb);
        v4f St, Dt, Ct;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            St[c] = dpp_shl1z(SY[c]);  // lanes t = 15: zero (their top face is an edge face, taken from `ex` below)
            Dt[c] = dpp_shl1z(D[c]);
            Ct[c] = dpp_shl1z(CY[c]);
        }
        FT = flux_half4(U, uT, SY, St, D, Dt, CY, Ct);
    }

    // ---- edge faces: both sub-faces of this lane's boundary cell
    wave_lds_sync();  // SY, D tiles
    stamp(6);
    float edge;
    {
        const float So_lr = g3 ? SX.w : SX.x, Do_lr = g3 ? D.w : D.x, Co_lr = g3 ? CX.w : CX.x;
        const float So_bt = lds_read(G.p_b + QUAD_TILE), Do_bt = lds_read(G.p_b + 2 * QUAD_TILE),
                    Co_bt = lds_read(G.p_b + 3 * QUAD_TILE);
        const float So = lr ? So_lr : So_bt;
        const float Do = lr ? Do_lr : Do_bt;
        const float Co = lr ? Co_lr : Co_bt;
        // own cell first, halo second; on low sides everything that is odd under the swap is negated (sgn = -1)
        const v2f F = flux_w2(m0, hu, sgn * So, Shn, Do, Dh, sgn * Co, sgn * hc, qs);
        edge = sgn * (0.5f * (F.x + F.y));
    }
    G.p_ringw[QUAD_OFF_EX - QUAD_OFF_RING] = edge;
    wave_lds_sync();
    stamp(7);

    // ---- Green-Gauss
    {
        const float FRm = bperm((lane - 16) << 2, FR.w);
        const v4f FL = v4f{g0 ? edge : FRm, FR.x, FR.y, FR.z};
        const v4f FRf = v4f{FR.x, FR.y, FR.z, g3 ? edge : FR.w};
        const v4f ex = *(const v4f*)(G.p_ringm + (QUAD_OFF_EX - QUAD_OFF_RING));
        v4f FB, FTf;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            FB[c] = dpp_shr1(ex[c], FT[c]);
            FTf[c] = t15 ? ex[c] : FT[c];
        }
        const v4f res = -((FRf - FL) * rhx) - ((FTf - FB) * rhy);
        // the residual is not read again by this sweep: stores that do not allocate in L2 (`nt`) -- 6.03 -> 5.86 us per sweep at
        // 0.87 M cells, 15.05 -> 14.89 at 3.47 M (same box, alternating builds).  Not for STEP, whose output the next step reads.
#ifdef Q2_NO_NT_STORE   // (A/B)
        if (!HALF || G.active) *(v4f_g*)((char*)ud + ((size_t)O.a0 << 2)) = STEP ? U + dt * res : res;
#else
        if (!HALF || G.active) {
            if constexpr (STEP) *(v4f_g*)((char*)ud + ((size_t)O.a0 << 2)) = U + dt * res;
            else __builtin_nontemporal_store(res, (v4f_g*)((char*)ud + ((size_t)O.a0 << 2)));
        }
#endif
    }
}

// ---- one quad by one wave.  (Sweeping several quads per wave with the next quad's loads in flight was measured and
// dropped: the prefetch registers cost a wave per SIMD -- 124 VGPRs against 89 -- and the sweep lives on wave-level
// parallelism: 6.1 us against 5.5 us at 0.87 M cells, no gain at 3.47 M.)
template <bool STAMP, int GM = 127, bool STEP = false, bool HALF = false>
__device__ __forceinline__ void sweep_quad(const QuadDesc2* __restrict__ qd, const int32_t* __restrict__ qtab, int32_t q,
                                           const float* __restrict__ u, const float* __restrict__ C, uint32_t ldc,
                                           float* __restrict__ ud, float* lds, int lane,
                                           unsigned long long* stamps = nullptr, float dt = 0.0f,
                                           const int32_t* __restrict__ qaux = nullptr) {
    const QuadLane G = quad_lane<HALF>(lds, lane);
    const QuadTab T = quad_load_tab(qd, qtab, q, lane, qaux);  // everything that needs the quad's index only is in flight
    const QuadOwn O = quad_load_own(G, T, u, C, ldc);
    const QuadHalo H = GM == 127 ? quad_load_halo_paired(G, T, u, C, ldc) : quad_load_halo<GM == 126 ? 127 : GM>(G, T, u, C, ldc);
    quad_compute<STAMP, STEP, HALF>(G, T, O, H, ud, stamps, dt);
}

#pragma clang fp contract(off)

}  // namespace quad2
