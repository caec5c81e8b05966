// libibhip: measurement probes for the 2-D sweep (not on the product path).  Same grid, workgroup shape and LDS
// allocation as the quad sweep launch of ibh_residual_advection, with the work stripped down in steps, so that a
// profile can say how much of the sweep's duration is launch ramp, streaming and dependent gathers:
//   mode 0: every wave returns at once                       (dispatch of the grid)
//   mode 1: + own-cell loads of u, Cx, Cy and the store of ud (the sweep's algorithmic traffic, 16 B per cell)
//   mode 2: + descriptor / halo-table loads and the dependent halo gathers (everything the sweep reads)
#include "ibh_common.h"
#include "ibh_quad2d.h"

namespace {

#ifndef WPB
#define WPB 4
#endif
#define PROBE_WG_LDS (WPB * (QUAD_LDS > BLK2_SWEEP_LDS ? QUAD_LDS : BLK2_SWEEP_LDS))

typedef float v4f __attribute__((ext_vector_type(4), aligned(4)));
typedef int v2i __attribute__((ext_vector_type(2)));

// bits of `what`: 1 own-cell loads + store, 2 table loads, 4 hu(k0), 8 hu(k1), 16 hdeep(k0), 32 hdeep(k1), 64 hc(k0),
// 128 hc(k1), 256 end cells
__global__ __launch_bounds__(64 * WPB) void k_probe_sweep(const float* __restrict__ u, const float* __restrict__ C,
                                                          uint32_t ldc, float* __restrict__ ud,
                                                          const QuadDesc2* __restrict__ qd,
                                                          const int32_t* __restrict__ qtab, int32_t nq, int32_t nwgq,
                                                          const BlockDesc2* __restrict__ blocks,
                                                          const int32_t* __restrict__ htab,
                                                          const int32_t* __restrict__ etab,
                                                          const int32_t* __restrict__ singles, int32_t ns, int32_t nc,
                                                          int what) {
    __shared__ __attribute__((aligned(16))) float lds[PROBE_WG_LDS];
    if (what == 0) {
        if (ldc == 0xffffffffu) lds[threadIdx.x] = 0.f;  // keep the allocation
        return;
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if ((int32_t)blockIdx.x < nwgq) {
        const int32_t q = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + wave);
        if (q >= nq) return;
        const QuadDesc2 d = qd[q];
        const int g = lane >> 4, t = lane & 15;
        const uint32_t a0 = (uint32_t)d.base + 64u * ((g >> 1) + 2 * (t >> 3)) + 4u * (g & 1) + 8u * (t & 7);
        v4f r = v4f{0.f, 0.f, 0.f, 0.f};
        if (what & 1) r = *(const v4f*)(u + a0) + *(const v4f*)(C + a0) + *(const v4f*)(C + ldc + a0);
        if (what & 2) {
            const int32_t* row = qtab + (size_t)q * IBH_QROW;
            const v2i hid = *(const v2i*)(row + 2 * lane);
            const int32_t eid = row[128 + (lane & 31)];
            const int delta = g == 0 ? -1 : g == 1 ? -8 : g == 2 ? 8 : 1;
            const float* Cn = C + ((g == 1 || g == 2) ? ldc : 0u);
            // (quads have no MIRROR side: the deeper cell hid + delta exists; clamped anyway, this is a probe)
            const int d0 = min(max(hid.x + delta, 0), nc - 1), d1 = min(max(hid.y + delta, 0), nc - 1);
            r.x += (float)(hid.x + hid.y + eid) * 1e-30f;
            if (what & 4) r.x += u[hid.x];
            if (what & 8) r.x += u[hid.y];
            if (what & 16) r.y += u[d0];
            if (what & 32) r.y += u[d1];
            if (what & 64) r.z += Cn[hid.x];
            if (what & 128) r.z += Cn[hid.y];
            if (what & 256) r.w += u[eid];
        }
        if (what & 1) *(v4f*)(ud + a0) = r;
        else if (r.x + r.y + r.z + r.w == 123.456f) ud[a0] = 0.f;
        return;
    }
    const int32_t pos = __builtin_amdgcn_readfirstlane((blockIdx.x - nwgq) * WPB + wave);
    if (pos >= ns) return;
    const int32_t blk = singles[pos];
    const BlockDesc2 bb = blocks[blk];
    const uint32_t c = (uint32_t)bb.base + lane;
    float r = 0.f;
    if (what & 1) r = u[c] + C[c] + C[ldc + c];
    if (what & 2) {
        const int32_t h = htab[(size_t)blk * 64 + lane], e = etab[(size_t)blk * 16 + (lane & 15)];
        const int s = lane >> 4;
        const int delta = s == 0 ? -1 : s == 1 ? 1 : s == 2 ? -8 : 8;
        const int32_t hd = min(max(h + delta, 0), nc - 1);  // MIRROR sides name the boundary cell itself: no deeper cell
        r += (float)(h + e) * 1e-30f;
        if (what & 4) r += u[h];
        if (what & 16) r += u[hd];
        if (what & 64) r += C[(s >= 2 ? ldc : 0u) + h];
        if (what & 256) r += u[e];
    }
    if (what & 1) ud[c] = r;
    else if (r == 123.456f) ud[c] = 0.f;
}

__global__ void k_probe_dispatch(int flag) {
    extern __shared__ float dyn[];
    if (flag == 12345) dyn[threadIdx.x] = 0.f;  // keeps the dynamic LDS allocation
}

}  // namespace

// dispatch cost of an empty grid: `nwg` workgroups of `threads` threads with `lds_bytes` of LDS each
extern "C" int ibh_probe_dispatch(int nwg, int threads, int lds_bytes) {
    IBH_REQUIRE(nwg > 0 && threads > 0 && threads <= 1024 && lds_bytes >= 0 && lds_bytes <= 65536, "ibh_probe_dispatch: bad shape");
    hipLaunchKernelGGL(k_probe_dispatch, dim3(nwg), dim3(threads), lds_bytes, ibh_stream, 0);
    IBH_LAUNCH_CHECK();
    return 0;
}

extern "C" int ibh_probe_sweep(ibh_part* p, const float* u, const float* C, int64_t ldc, float* ud, int mode) {
    IBH_REQUIRE(p && u && C && ud, "ibh_probe_sweep: null argument");
    IBH_REQUIRE(p->nd == 2 && p->fuse_all && p->nq[0] > 0, "ibh_probe_sweep: needs a partition the quad sweep runs on");
    const int32_t nq = p->nq[0], ns = p->nqs[0];
    const int32_t nwgq = (nq + WPB - 1) / WPB, nwgs = (ns + WPB - 1) / WPB;
    // modes 0..2 as documented; mode >= 16: `mode - 16` is the bit mask of k_probe_sweep
    const int what = mode == 0 ? 0 : mode == 1 ? 1 : mode == 2 ? 511 : mode - 16;
    hipLaunchKernelGGL(k_probe_sweep, dim3(nwgq + nwgs), dim3(64 * WPB), 0, ibh_stream, u, C, (uint32_t)ldc, ud,
                       p->qd[0], p->qtab[0], nq, nwgq, p->blocks2, p->htab, p->etab, p->qsingles[0], ns, p->nc, what);
    IBH_LAUNCH_CHECK();
    return 0;
}
