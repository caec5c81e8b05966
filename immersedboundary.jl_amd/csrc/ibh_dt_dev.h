// libibhip: device bodies of the time-step evaluation of an explicit advection step (ibh_timestep_advection, ibh_ops.hip), with
// the workgroup's index and count as arguments: the same code runs as its own launches and beside the BC-set workgroups of a
// march step (ibh_ops.hip: k_bcinterp_dt / k_bcscatter_dt).
#pragma once
#include "ibh_common.h"

#define DT_BLOCK 256

namespace dt_dev {

// at_faces (:907-909): (u_o*h_n + u_n*h_o)/(h_n + h_o)
__device__ __forceinline__ float dt_face_avg(float uo, float un, float ho, float hn) { return (uo * hn + un * ho) / (hn + ho); }
__device__ __forceinline__ float dt_csr_mean_face_avg(const int32_t* __restrict__ off, const int32_t* __restrict__ idx,
                                                      int32_t c, const int32_t* __restrict__ own,
                                                      const int32_t* __restrict__ nei, const float* __restrict__ h,
                                                      const float* __restrict__ u) {
    int32_t b = off[c], e = off[c + 1];
    if (e == b) return 0.0f;
    float w = 1.0f / (float)(e - b);
    float s = 0.0f;
    for (int32_t k = b; k < e; ++k) {
        int32_t f = idx[k];
        int32_t o = own[f], n = nei[f];
        float t = dt_face_avg(u[o], u[n], h[o], h[n]) * w;
        s = (k == b) ? t : s + t;
    }
    return s;
}

struct GradDims {
    DimData d[IBH_MAXD];
    const float* h[IBH_MAXD];
    const int32_t* side;
};

inline GradDims grad_dims(const ibh_part* p) {
    GradDims G;
    for (int d = 0; d < p->nd; ++d) {
        G.d[d] = p->dim[d];
        G.h[d] = p->spacing + (int64_t)d * p->nc;
    }
    G.side = p->side;
    return G;
}

// (device bodies with the workgroup's index and count as arguments: the same code runs as its own launch and beside the
// BC-set workgroups of a march step, k_bcinterp_dt / k_bcscatter_dt below)
template <int ND, bool TILED>
__device__ __forceinline__ void dt_partial_wg(int wg, int nwg, int32_t nc, const GradDims& G, const float* __restrict__ C,
                                              int64_t ldc, float* __restrict__ partial) {
    float m = 0.0f;
#ifdef IBH_NO_XCD_CELLS
    const int64_t first = wg;
#else
    const int64_t first = ibh_xcd_chunk(wg, nwg);
#endif
    for (int64_t c = first * (int64_t)blockDim.x + threadIdx.x; c < nc; c += (int64_t)nwg * blockDim.x) {
        int32_t sd[2 * ND];
#pragma unroll
        for (int s = 0; s < 2 * ND; ++s) {
            if (TILED) {
                const int pos = ((int)c >> (3 * (s >> 1))) & 7, st = 1 << (3 * (s >> 1));
                const bool inb = (s & 1) ? pos < 7 : pos > 0;
                sd[s] = inb ? (int32_t)c + ((s & 1) ? st : -st) : G.side[(int64_t)s * nc + c];
            } else sd[s] = G.side[(int64_t)s * nc + c];
        }
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const float* Cd = C + (int64_t)d * ldc;
            const float hc = G.h[d][c], uc = Cd[c];
            const int32_t l = sd[2 * d], r = sd[2 * d + 1];
            float ar, al;
            if (r >= 0) ar = dt_face_avg(uc, Cd[r], hc, G.h[d][r]) * 1.0f;
            else if (r == -2) ar = 0.0f;
            else ar = dt_csr_mean_face_avg(G.d[d].roff, G.d[d].ridx, (int32_t)c, G.d[d].owners, G.d[d].neighbors, G.h[d], Cd);
            if (l >= 0) al = dt_face_avg(Cd[l], uc, G.h[d][l], hc) * 1.0f;
            else if (l == -2) al = 0.0f;
            else al = dt_csr_mean_face_avg(G.d[d].loff, G.d[d].lidx, (int32_t)c, G.d[d].owners, G.d[d].neighbors, G.h[d], Cd);
            m = fmaxf(m, (ar + al) / hc);
        }
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float wm[DT_BLOCK / 64];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float b = wm[0];
        for (int w = 1; w < DT_BLOCK / 64; ++w) b = fmaxf(b, wm[w]);
        partial[wg] = b;
    }
}
__device__ __forceinline__ void dt_final_wg(int n, const float* __restrict__ partial, float scale, float* __restrict__ dt) {
    float m = 0.0f;
    for (int i = threadIdx.x; i < n; i += DT_BLOCK) m = fmaxf(m, partial[i]);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float wm[DT_BLOCK / 64];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float b = wm[0];
        for (int w = 1; w < DT_BLOCK / 64; ++w) b = fmaxf(b, wm[w]);
        *dt = (0.5f / b) * scale;  // advection.jl:53 and :65
    }
}

}  // namespace dt_dev
