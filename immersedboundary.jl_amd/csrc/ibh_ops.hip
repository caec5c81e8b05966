// libibhip: face-list ("general") grid-operator kernels -- one entry point per reference
// operator (ImmersedBoundary.jl:873-1157), accumulators, ghost-cell BC, row gather/scatter
// and the small vector ops of the FAS loop.  HBM-bound gathers: threads run over the
// contiguous row index (cell or face) so loads/stores of the row-major side are coalesced;
// the variable index is the slow grid dimension.
//
// Arithmetic follows the reference's broadcast order operation by operation (compiled
// with -ffp-contract=off, IEEE divide/sqrt) so results are bit-comparable with the oracle.
#include "ibh_bcset_dev.h"
#include "ibh_common.h"

#define OPS_BLOCK 256
#include "ibh_dt_dev.h"
using dt_dev::GradDims;
using dt_dev::grad_dims;
using dt_dev::dt_partial_wg;
using dt_dev::dt_final_wg;

namespace {

__device__ __forceinline__ float face_avg(float uo, float un, float ho, float hn) {
    // at_faces (:907-909): (u_o*h_n + u_n*h_o)/(h_n + h_o)
    return (uo * hn + un * ho) / (hn + ho);
}

__global__ void k_gather(const int32_t* __restrict__ rows, int32_t n, const float* __restrict__ src, int64_t lds,
                         float* __restrict__ dst, int64_t ldd) {
    int64_t v = blockIdx.y;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i + v * ldd] = src[rows[i] + v * lds];
}

__global__ void k_scatter(const int32_t* __restrict__ rows, int32_t n, const float* __restrict__ src, int64_t lds,
                          float* __restrict__ dst, int64_t ldd) {
    int64_t v = blockIdx.y;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[rows[i] + v * ldd] = src[i + v * lds];
}

__global__ void k_copy_rows(const int32_t* __restrict__ drows, const int32_t* __restrict__ srows, int32_t n,
                            const float* __restrict__ src, int64_t lds, float* __restrict__ dst, int64_t ldd) {
    int64_t v = blockIdx.y;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[drows[i] + v * ldd] = src[srows[i] + v * lds];
}

__global__ void k_at_faces(int32_t nf, const int32_t* __restrict__ own, const int32_t* __restrict__ nei,
                           const float* __restrict__ h, const float* __restrict__ u, int64_t ldu,
                           float* __restrict__ out, int64_t ldo) {
    int64_t v = blockIdx.y;
    for (int64_t f = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; f < nf; f += (int64_t)gridDim.x * blockDim.x) {
        int32_t o = own[f], n = nei[f];
        out[f + v * ldo] = face_avg(u[o + v * ldu], u[n + v * ldu], h[o], h[n]);
    }
}

// mode 0: face_distance, 1: owner_distance, 2: neighbor_distance
__global__ void k_distances(int32_t nf, const int32_t* __restrict__ own, const int32_t* __restrict__ nei,
                            const float* __restrict__ h, int mode, float* __restrict__ out) {
    for (int64_t f = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; f < nf; f += (int64_t)gridDim.x * blockDim.x) {
        float ho = h[own[f]], hn = h[nei[f]];
        out[f] = mode == 0 ? (ho + hn) / 2.0f : (mode == 1 ? ho / 2.0f : hn / 2.0f);
    }
}

__global__ void k_face_gradient(int32_t nf, const int32_t* __restrict__ own, const int32_t* __restrict__ nei,
                                const float* __restrict__ h, const float* __restrict__ u, int64_t ldu,
                                float* __restrict__ out, int64_t ldo) {
    int64_t v = blockIdx.y;
    for (int64_t f = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; f < nf; f += (int64_t)gridDim.x * blockDim.x) {
        int32_t o = own[f], n = nei[f];
        float fd = (h[o] + h[n]) / 2.0f;                       // face_distance (:1001)
        out[f + v * ldo] = (u[n + v * ldu] - u[o + v * ldu]) / fd;  // :1041-1043
    }
}

// mean over a CSR row of a face field (face accumulator with weights 1/len, :501-506)
__device__ __forceinline__ float csr_mean(const int32_t* __restrict__ off, const int32_t* __restrict__ idx, int32_t c,
                                          const float* __restrict__ uf) {
    int32_t b = off[c], e = off[c + 1];
    if (e == b) return 0.0f;
    float w = 1.0f / (float)(e - b);
    float s = uf[idx[b]] * w;
    for (int32_t k = b + 1; k < e; ++k) s = s + uf[idx[k]] * w;
    return s;
}

__global__ void k_green_gauss(int32_t nc, const int32_t* __restrict__ loff, const int32_t* __restrict__ lidx,
                              const int32_t* __restrict__ roff, const int32_t* __restrict__ ridx,
                              const float* __restrict__ h, const float* __restrict__ uf, int64_t ldf,
                              float* __restrict__ out, int64_t ldo, int uns) {
    int64_t v = blockIdx.y;
    const float* ufv = uf + v * ldf;
    for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < nc; c += (int64_t)gridDim.x * blockDim.x) {
        float ar = csr_mean(roff, ridx, (int32_t)c, ufv);
        float al = csr_mean(loff, lidx, (int32_t)c, ufv);
        out[c + v * ldo] = (uns ? (ar + al) : (ar - al)) / h[c];  // :925 / :941
    }
}

// cell_gradient = green_gauss(at_faces(u)) without materialising the face field (:965-972)
__device__ __forceinline__ float csr_mean_face_avg(const int32_t* __restrict__ off, const int32_t* __restrict__ idx,
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
        float t = face_avg(u[o], u[n], h[o], h[n]) * w;
        s = (k == b) ? t : s + t;
    }
    return s;
}

// cell_gradient(part, u), the tuple form, on partitions without the block structure: all dimensions in one launch
// (thread per cell and field; out[(d * nv + v) * ldo + c], the layout of ibh_cell_gradient_nd)
// Sides with ONE face (side table of the partition, ibh_common.h) are evaluated directly from the cell across -- the same
// expression the CSR walk evaluates for its single entry (weight 1.0f) -- the others walk the lists.
template <int ND>
__global__ void k_cell_gradient_all(int32_t nc, GradDims G, const float* __restrict__ u, int64_t ldu, int nv,
                                    float* __restrict__ out, int64_t ldo) {
    const int64_t v = blockIdx.y;
    const float* uv = u + v * ldu;
    for (int64_t c = IBH_WG_X() * (int64_t)blockDim.x + threadIdx.x; c < nc; c += (int64_t)gridDim.x * blockDim.x) {
        int32_t sd[2 * ND];
#pragma unroll
        for (int s = 0; s < 2 * ND; ++s) sd[s] = G.side[(int64_t)s * nc + c];
        const float uc = uv[c];
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const float hc = G.h[d][c];
            const int32_t l = sd[2 * d], r = sd[2 * d + 1];
            float ar, al;
            if (r >= 0) ar = face_avg(uc, uv[r], hc, G.h[d][r]) * 1.0f;
            else if (r == -2) ar = 0.0f;
            else ar = csr_mean_face_avg(G.d[d].roff, G.d[d].ridx, (int32_t)c, G.d[d].owners, G.d[d].neighbors, G.h[d], uv);
            if (l >= 0) al = face_avg(uv[l], uc, G.h[d][l], hc) * 1.0f;
            else if (l == -2) al = 0.0f;
            else al = csr_mean_face_avg(G.d[d].loff, G.d[d].lidx, (int32_t)c, G.d[d].owners, G.d[d].neighbors, G.h[d], uv);
            out[c + ((int64_t)d * nv + v) * ldo] = (ar - al) / hc;
        }
    }
}

// dt of an explicit advection step (test/advection.jl:52-59, :65): max over cells and dimensions of
// unsigned_green_gauss(at_faces(C_d, d), d) -- the same expressions as k_at_faces + k_green_gauss(unsigned), a face value
// evaluated by both of its cells instead of being written and read back.  Two launches and NO atomics: workgroup maxima into
// an array, then one workgroup reduces them and writes dt = (0.5 / max) * scale.  (A single launch whose workgroups meet at
// one counter -- atomicMax + last-workgroup-out -- took 39 us for 1 024 workgroups: arrivals at one address serialise at
// ~35 ns each across the XCDs; this form takes ~8.)
// TILED: the partition is made of complete 8^ND blocks in order (cell c = block c / 8^ND, position c % 8^ND, x fastest): the
// cell across an in-block face is c -+ 8^d -- what the side table holds for it -- so the table is read by the cells on the
// block faces only (a quarter of the index traffic in 2-D; round 4: 18.6 -> us per evaluation at 0.87 M cells)
template <int ND, bool TILED>
__global__ __launch_bounds__(OPS_BLOCK) void k_timestep_advection(int32_t nc, GradDims G, const float* __restrict__ C,
                                                                  int64_t ldc, float* __restrict__ partial) {
    dt_partial_wg<ND, TILED>(blockIdx.x, gridDim.x, nc, G, C, ldc, partial);
}
__global__ __launch_bounds__(OPS_BLOCK) void k_dt_from_partials(int n, const float* __restrict__ partial, float scale,
                                                                float* __restrict__ dt) {
    dt_final_wg(n, partial, scale, dt);
}
// ---- a march step's BC-set launches with the time step of the NEXT step riding beside them (horizontal fusion: the time
// step depends on C alone, the boundary conditions on the field; both are short launches bound by dependent loads, and side
// by side they cost what the longer one costs): workgroups [0, nwg_bc) run the BC-set body, the rest the dt body.
struct BcLaunch {
    int32_t g0, g1;
    const float *eta, *w, *value;
    const int32_t *off, *donor, *bidx, *mode, *ghost_direct, *ghost;
    float* gval;
};
template <int ND, bool TILED>
__global__ __launch_bounds__(OPS_BLOCK) void k_bcinterp_dt(int nwg_bc, BcLaunch B, float* a, int32_t nc, GradDims G,
                                                           const float* __restrict__ C, int64_t ldc,
                                                           float* __restrict__ partial) {
    if ((int)blockIdx.x < nwg_bc)
        bcset_dev::interp_wg(blockIdx.x, nwg_bc, B.g0, B.g1, B.eta, B.off, B.donor, B.w, B.bidx, B.mode, B.value, a, B.gval,
                             B.ghost_direct, a);
    else dt_partial_wg<ND, TILED>(blockIdx.x - nwg_bc, gridDim.x - nwg_bc, nc, G, C, ldc, partial);
}
__global__ __launch_bounds__(OPS_BLOCK) void k_bcscatter_dt(int nwg_bc, BcLaunch B, float* a, int npart,
                                                            const float* __restrict__ partial, float scale,
                                                            float* __restrict__ dt) {
    if ((int)blockIdx.x < nwg_bc) bcset_dev::scatter_wg(blockIdx.x, nwg_bc, B.g0, B.g1, B.ghost, B.gval, a);
    else dt_final_wg(npart, partial, scale, dt);
}
// a direct level (interpolation blended straight into the field) as the second launch of the set: interp body + dt final
__global__ __launch_bounds__(OPS_BLOCK) void k_bcinterp_dtfinal(int nwg_bc, BcLaunch B, float* a, int npart,
                                                                const float* __restrict__ partial, float scale,
                                                                float* __restrict__ dt) {
    if ((int)blockIdx.x < nwg_bc)
        bcset_dev::interp_wg(blockIdx.x, nwg_bc, B.g0, B.g1, B.eta, B.off, B.donor, B.w, B.bidx, B.mode, B.value, a, B.gval,
                             B.ghost_direct, a);
    else dt_final_wg(npart, partial, scale, dt);
}

// gradient of one field at cell c in every dimension: the expressions of k_cell_gradient_all
template <int ND>
__device__ __forceinline__ void cell_gradient_at(const GradDims& G, const int32_t (&sd)[2 * ND], int32_t nc, int32_t c,
                                                 const float* __restrict__ uv, float (&g)[ND]) {
    const float uc = uv[c];
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        const float hc = G.h[d][c];
        const int32_t l = sd[2 * d], r = sd[2 * d + 1];
        float ar, al;
        if (r >= 0) ar = face_avg(uc, uv[r], hc, G.h[d][r]) * 1.0f;
        else if (r == -2) ar = 0.0f;
        else ar = csr_mean_face_avg(G.d[d].roff, G.d[d].ridx, c, G.d[d].owners, G.d[d].neighbors, G.h[d], uv);
        if (l >= 0) al = face_avg(uv[l], uc, G.h[d][l], hc) * 1.0f;
        else if (l == -2) al = 0.0f;
        else al = csr_mean_face_avg(G.d[d].loff, G.d[d].lidx, c, G.d[d].owners, G.d[d].neighbors, G.h[d], uv);
        g[d] = (ar - al) / hc;
    }
}
// shear_rate of the gradients of a velocity field / Wray_Agarwal of the gradients of (R, S), face-list partitions: thread per
// cell, the gradients never leave the registers (same expressions as k_cell_gradient_all + ibh_turb.hip's k_shear /
// k_wray_agarwal: bit-identical to the composition)
template <int ND>
__global__ void k_shear_of_velocity_cells(int32_t nc, GradDims G, const float* __restrict__ vel, int64_t ldv,
                                          float* __restrict__ S, float* __restrict__ Gout, int64_t ldg) {
    for (int64_t c = IBH_WG_X() * (int64_t)blockDim.x + threadIdx.x; c < nc; c += (int64_t)gridDim.x * blockDim.x) {
        int32_t sd[2 * ND];
#pragma unroll
        for (int s = 0; s < 2 * ND; ++s) sd[s] = G.side[(int64_t)s * nc + c];
        float g[ND][ND];
#pragma unroll
        for (int i = 0; i < ND; ++i) cell_gradient_at<ND>(G, sd, nc, (int32_t)c, vel + (int64_t)i * ldv, g[i]);
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < ND; ++i)
#pragma unroll
            for (int j = 0; j < ND; ++j) {
                const float t = (g[i][j] + g[j][i]) / 2.0f;
                s = s + t * t;
            }
        S[c] = sqrtf(2.0f * s);
        if (Gout) {  // d u_i / d x_j in column ND j + i (the tuple cell_gradient's layout)
#pragma unroll
            for (int i = 0; i < ND; ++i)
#pragma unroll
                for (int j = 0; j < ND; ++j) Gout[(int64_t)(ND * j + i) * ldg + c] = g[i][j];
        }
    }
}
template <int ND>
__global__ void k_wray_agarwal_of_cells(int32_t nc, GradDims G, const float* __restrict__ R, const float* __restrict__ S,
                                        float sigmaR, float C1, float kappa, float* __restrict__ nut,
                                        float* __restrict__ nuR, float* __restrict__ Sout) {
    const float C2 = sigmaR + C1 / (kappa * kappa);
    constexpr float EPS32 = 1.1920929e-07f;
    for (int64_t c = IBH_WG_X() * (int64_t)blockDim.x + threadIdx.x; c < nc; c += (int64_t)gridDim.x * blockDim.x) {
        int32_t sd[2 * ND];
#pragma unroll
        for (int s = 0; s < 2 * ND; ++s) sd[s] = G.side[(int64_t)s * nc + c];
        float gR[ND], gS[ND];
        cell_gradient_at<ND>(G, sd, nc, (int32_t)c, R, gR);
        cell_gradient_at<ND>(G, sd, nc, (int32_t)c, S, gS);
        float dot = gR[0] * gS[0];
#pragma unroll
        for (int d = 1; d < ND; ++d) dot = dot + gR[d] * gS[d];
        const float r = R[c], s = S[c];
        const float src = C1 * r * s + C2 * dot * (r / (s + EPS32));
        nut[c] = r;
        nuR[c] = r * sigmaR;
        Sout[c] = fminf(src, 10.0f * r);
    }
}

__global__ void k_cell_gradient(int32_t nc, DimData D, const float* __restrict__ h, const float* __restrict__ u,
                                int64_t ldu, float* __restrict__ out, int64_t ldo) {
    int64_t v = blockIdx.y;
    const float* uv = u + v * ldu;
    for (int64_t c = IBH_WG_X() * (int64_t)blockDim.x + threadIdx.x; c < nc; c += (int64_t)gridDim.x * blockDim.x) {
        float ar = csr_mean_face_avg(D.roff, D.ridx, (int32_t)c, D.owners, D.neighbors, h, uv);
        float al = csr_mean_face_avg(D.loff, D.lidx, (int32_t)c, D.owners, D.neighbors, h, uv);
        out[c + v * ldo] = (ar - al) / h[c];
    }
}

// JST_sensor for one dim (:1091-1096): nu = (1e-7 + |gg(dp)|)/(1e-7 + ugg(|dp|)), dp = p_n - p_o
__device__ __forceinline__ void csr_mean_diff(const int32_t* __restrict__ off, const int32_t* __restrict__ idx, int32_t c,
                                              const int32_t* __restrict__ own, const int32_t* __restrict__ nei,
                                              const float* __restrict__ p, float& sd, float& sa) {
    int32_t b = off[c], e = off[c + 1];
    sd = 0.0f;
    sa = 0.0f;
    if (e == b) return;
    float w = 1.0f / (float)(e - b);
    for (int32_t k = b; k < e; ++k) {
        int32_t f = idx[k];
        float d = p[nei[f]] - p[own[f]];
        float td = d * w, ta = fabsf(d) * w;
        sd = (k == b) ? td : sd + td;
        sa = (k == b) ? ta : sa + ta;
    }
}

__device__ __forceinline__ float jst_dim(const DimData& D, const float* __restrict__ h, const float* __restrict__ p,
                                         int32_t c) {
    float dr, ar, dl, al;
    csr_mean_diff(D.roff, D.ridx, c, D.owners, D.neighbors, p, dr, ar);
    csr_mean_diff(D.loff, D.lidx, c, D.owners, D.neighbors, p, dl, al);
    float hc = h[c];
    float gg = (dr - dl) / hc;
    float ugg = (ar + al) / hc;
    return (1e-7f + fabsf(gg)) / (1e-7f + ugg);
}

__global__ void k_jst(int32_t nc, int nd, int dim, DimData D0, DimData D1, DimData D2,
                      const float* __restrict__ spacing, const float* __restrict__ p, int64_t ldp,
                      float* __restrict__ out, int64_t ldo) {
    int64_t v = blockIdx.y;
    const float* pv = p + v * ldp;
    for (int64_t c = IBH_WG_X() * (int64_t)blockDim.x + threadIdx.x; c < nc; c += (int64_t)gridDim.x * blockDim.x) {
        float nu;
        if (dim > 0) {
            const DimData& D = dim == 1 ? D0 : (dim == 2 ? D1 : D2);
            nu = jst_dim(D, spacing + (int64_t)(dim - 1) * nc, pv, (int32_t)c);
        } else {
            nu = 1e-7f;  // :1082-1086
            nu = fmaxf(nu, jst_dim(D0, spacing, pv, (int32_t)c));
            nu = fmaxf(nu, jst_dim(D1, spacing + (int64_t)nc, pv, (int32_t)c));
            if (nd == 3) nu = fmaxf(nu, jst_dim(D2, spacing + 2 * (int64_t)nc, pv, (int32_t)c));
        }
        out[c + v * ldo] = nu;
    }
}

__device__ __forceinline__ float sgn(float x) { return (float)((x > 0.0f) - (x < 0.0f)); }
__device__ __forceinline__ float minmod(float a, float b) {  // :1099
    return fminf(fabsf(a), fabsf(b)) * (sgn(a) + sgn(b)) / 2.0f;
}

__global__ void k_muscl(int32_t nf, const int32_t* __restrict__ own, const int32_t* __restrict__ nei,
                        const float* __restrict__ h, const float* __restrict__ u, const float* __restrict__ du,
                        int64_t ld, const float* __restrict__ Dsens, int high_order, float* __restrict__ uL,
                        float* __restrict__ uR, int64_t ldf) {
    int64_t v = blockIdx.y;
    for (int64_t f = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; f < nf; f += (int64_t)gridDim.x * blockDim.x) {
        int32_t o = own[f], n = nei[f];
        float dow = h[o] / 2.0f, dn = h[n] / 2.0f;  // owner/neighbor_distance
        float uo = u[o + v * ld], un = u[n + v * ld];
        float guf = (un - uo) / (dow + dn);
        float duo = du[o + v * ld], dun = du[n + v * ld];
        float gu = (2.0f * duo - guf) * dow;
        float Du = (2.0f * dun - guf) * dn;
        float s = minmod(Du, gu);
        float l = uo + s, r = un - s;
        if (Dsens) {
            float Df = fmaxf(fmaxf(Dsens[o], Dsens[n]), 1e-7f);
            float uf = (uo * dn + un * dow) / (dow + dn);
            if (high_order) uf = uf + (duo * dow - dun * dn) / 8.0f;
            l = l * Df + (1.0f - Df) * uf;
            r = r * Df + (1.0f - Df) * uf;
        }
        uL[f + v * ldf] = l;
        uR[f + v * ldf] = r;
    }
}

__global__ void k_accumulate(int32_t n_out, const int32_t* __restrict__ off, const int32_t* __restrict__ idx,
                             const float* __restrict__ w, const int32_t* __restrict__ remap,
                             const float* __restrict__ v, int64_t ldv, float* __restrict__ out, int64_t ldo) {
    int64_t var = blockIdx.y;
    const float* vv = v + var * ldv;
    for (int64_t r = IBH_WG_X() * (int64_t)blockDim.x + threadIdx.x; r < n_out; r += (int64_t)gridDim.x * blockDim.x) {
        int32_t b = off[r], e = off[r + 1];
        float s = 0.0f;
        for (int32_t k = b; k < e; ++k) {
            int32_t j = idx[k];
            if (remap) j = remap[j];
            float t = w ? vv[j] * w[k] : vv[j];
            s = (k == b) ? t : s + t;
        }
        out[r + var * ldo] = s;
    }
}

// The same for several fields at once: one thread per row, the row's indices and weights read once for up to NVB
// fields (the per-(row, field) form reads them once per field).  Same sum order per field.
// v2 (optional): the stencil is applied to v - v2 (elementwise difference first, like `acc(a .- b)`); ADD: out .+= result
template <int NVB, bool ADD = false>
__global__ void k_accumulate_rows(int32_t n_out, const int32_t* __restrict__ off, const int32_t* __restrict__ idx,
                                  const float* __restrict__ w, const int32_t* __restrict__ remap,
                                  const float* __restrict__ v, int64_t ldv, float* __restrict__ out, int64_t ldo, int nv,
                                  const float* __restrict__ v2 = nullptr) {
    const int v0 = blockIdx.y * NVB;
    const int nb = min(NVB, nv - v0);
    const float* vv = v + (int64_t)v0 * ldv;
    const float* vv2 = v2 ? v2 + (int64_t)v0 * ldv : nullptr;
    float* oo = out + (int64_t)v0 * ldo;
    for (int64_t r = IBH_WG_X() * (int64_t)blockDim.x + threadIdx.x; r < n_out; r += (int64_t)gridDim.x * blockDim.x) {
        const int32_t b = off[r], e = off[r + 1];
        float s[NVB];
#pragma unroll
        for (int q = 0; q < NVB; ++q) s[q] = 0.0f;
        int32_t k = b;
        // rows that start on a 16-byte boundary (the 2^nd-point transfer operators: every row) take their entries four
        // at a time: one dwordx4 of indices, one of weights, the gathers of the four entries in flight together; the
        // sum keeps the entry order
        if ((b & 3) == 0 && w)
            for (; k + 4 <= e; k += 4) {
                const int4 j4 = *reinterpret_cast<const int4*>(idx + k);
                const float4 w4 = *reinterpret_cast<const float4*>(w + k);
                int32_t jj[4] = {j4.x, j4.y, j4.z, j4.w};
                const float ww[4] = {w4.x, w4.y, w4.z, w4.w};
                if (remap) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) jj[i] = remap[jj[i]];
                }
                float x[4][NVB];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int q = 0; q < NVB; ++q)
                        if (q < nb) x[i][q] = vv[jj[i] + (int64_t)q * ldv];
                if (vv2) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int q = 0; q < NVB; ++q)
                            if (q < nb) x[i][q] = x[i][q] - vv2[jj[i] + (int64_t)q * ldv];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int q = 0; q < NVB; ++q)
                        if (q < nb) {
                            const float t = x[i][q] * ww[i];
                            s[q] = (k + i == b) ? t : s[q] + t;
                        }
            }
        for (; k < e; ++k) {
            int32_t j = idx[k];
            if (remap) j = remap[j];
            const float wk = w ? w[k] : 1.0f;
#pragma unroll
            for (int q = 0; q < NVB; ++q) {
                if (q < nb) {
                    float x = vv[j + (int64_t)q * ldv];
                    if (vv2) x = x - vv2[j + (int64_t)q * ldv];
                    const float t = w ? x * wk : x;
                    s[q] = (k == b) ? t : s[q] + t;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NVB; ++q)
            if (q < nb) oo[r + (int64_t)q * ldo] = ADD ? oo[r + (int64_t)q * ldo] + s[q] : s[q];
    }
}

// out .+= acc(a .- b) for up to 8 fields with the donors' fields side by side: a pre-pass writes d[j][0..7] = a[j, :] - b[j, :]
// (32 bytes per donor), the row kernel then takes a donor with TWO 16-byte gathers instead of two 4-byte gathers per field
// (6 fields x 8 donors: 16 gather instructions per row instead of 96) -- same differences, same products, same order.
__global__ void k_pack_diff8(int32_t n_in, int nv, const float* __restrict__ a, const float* __restrict__ b, int64_t ld,
                             float* __restrict__ d) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < (int64_t)n_in * 8; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = t >> 3;
        const int q = (int)(t & 7);
        d[t] = q < nv ? a[j + (int64_t)q * ld] - b[j + (int64_t)q * ld] : 0.0f;
    }
}
template <int NV>
__global__ void k_accumulate_packed_add(int32_t n_out, const int32_t* __restrict__ off, const int32_t* __restrict__ idx,
                                        const float* __restrict__ w, const float4* __restrict__ d, float* __restrict__ out,
                                        int64_t ldo) {
    for (int64_t r = IBH_WG_X() * (int64_t)blockDim.x + threadIdx.x; r < n_out; r += (int64_t)gridDim.x * blockDim.x) {
        const int32_t b = off[r], e = off[r + 1];
        float s[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) s[q] = 0.0f;
        int32_t k = b;
        if ((b & 3) == 0)
            for (; k + 4 <= e; k += 4) {
                const int4 j4 = *reinterpret_cast<const int4*>(idx + k);
                const float4 w4 = *reinterpret_cast<const float4*>(w + k);
                const int32_t jj[4] = {j4.x, j4.y, j4.z, j4.w};
                const float ww[4] = {w4.x, w4.y, w4.z, w4.w};
                float4 lo[4], hi[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    lo[i] = d[2 * (int64_t)jj[i]];
                    if (NV > 4) hi[i] = d[2 * (int64_t)jj[i] + 1];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float x[8] = {lo[i].x, lo[i].y, lo[i].z, lo[i].w, NV > 4 ? hi[i].x : 0.0f, NV > 4 ? hi[i].y : 0.0f,
                                        NV > 4 ? hi[i].z : 0.0f, NV > 4 ? hi[i].w : 0.0f};
#pragma unroll
                    for (int q = 0; q < NV; ++q) {
                        const float t = x[q] * ww[i];
                        s[q] = (k + i == b) ? t : s[q] + t;
                    }
                }
            }
        for (; k < e; ++k) {
            const int64_t j = idx[k];
            const float wk = w[k];
            const float4 lo = d[2 * j];
            float4 hi = float4{0.0f, 0.0f, 0.0f, 0.0f};
            if (NV > 4) hi = d[2 * j + 1];
            const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const float t = x[q] * wk;
                s[q] = (k == b) ? t : s[q] + t;
            }
        }
#pragma unroll
        for (int q = 0; q < NV; ++q) out[r + (int64_t)q * ldo] = out[r + (int64_t)q * ldo] + s[q];
    }
}

// a[ghost] = eta*ia + (1-eta)*ba (:1242-1245); ba from array, constant, or ia (copy BC)
__global__ void k_bc_blend(int32_t ng, const int32_t* __restrict__ ghost, const float* __restrict__ eta,
                           float* __restrict__ a, int64_t lda, const float* __restrict__ ia, int64_t ldi,
                           const float* __restrict__ ba, int64_t ldb, const float* __restrict__ bconst) {
    int64_t v = blockIdx.y;
    for (int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; g < ng; g += (int64_t)gridDim.x * blockDim.x) {
        float e = eta[g];
        float i = ia[g + v * ldi];
        float b = ba ? ba[g + v * ldb] : bconst[v];
        a[ghost[g] + v * lda] = e * i + (1.0f - e) * b;
    }
}

__global__ void k_axpy_clamped(int64_t n, float omega, const float* __restrict__ r, float* __restrict__ q) {
    float w = fminf(fmaxf(omega, 0.0f), 1.0f);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        q[i] = q[i] + w * r[i];
}

__global__ void k_axpy(int64_t n, float a, const float* __restrict__ x, float* __restrict__ y) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = a * x[i] + y[i];
}

// final step of the two-stage reductions below: one workgroup adds the workgroup sums (in index order: a fixed order for a
// fixed grid) -- no atomics: 1 024 - 2 048 arrivals at ONE address cost 20 - 40 us on this part (they serialise across the XCDs)
__global__ __launch_bounds__(OPS_BLOCK) void k_sum_partials(int n, const double* __restrict__ part, double* __restrict__ out) {
    __shared__ double sh[OPS_BLOCK];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += OPS_BLOCK) s += part[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = OPS_BLOCK / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = sh[0];
}
// scratch of the two-stage reductions (per host thread: one stream of reductions at a time)
static double* red_scratch() {
    static thread_local double* buf = nullptr;
    if (!buf && hipMalloc((void**)&buf, 4096 * sizeof(double)) != hipSuccess) buf = nullptr;
    return buf;
}

__global__ void k_sumsq(int64_t n, const float* __restrict__ x, double* __restrict__ out) {
    double s = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double t = (double)x[i];
        s += t * t;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    __shared__ double part[OPS_BLOCK / 64];
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) part[w] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < OPS_BLOCK / 64; ++i) t += part[i];
        out[blockIdx.x] = t;    // (workgroup sums: k_sum_partials adds them)
    }
}

// q += clamp(omega, 0, 1) * r and out += sum r^2 in one pass over r (solver.jl:82 and the norm of :84 on the same array)
__global__ void k_axpy_clamped_sumsq(int64_t n, float omega, const float* __restrict__ r, float* __restrict__ q,
                                     double* __restrict__ out) {
    const float w = fminf(fmaxf(omega, 0.0f), 1.0f);
    double s = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float ri = r[i];
        q[i] = q[i] + w * ri;
        s += (double)ri * (double)ri;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    __shared__ double part[OPS_BLOCK / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) part[wv] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < OPS_BLOCK / 64; ++i) t += part[i];
        out[blockIdx.x] = t;    // (workgroup sums: k_sum_partials adds them)
    }
}

// One pass of `FAS!` over a residual array (solver.jl:80-84): rr = r [+ source]; [q += clamp(omega, 0, 1) * rr]; [sum rr^2]
// -- `r .+= source`, the fixed-point update and the norm on ONE read of r (round 3 had the sum as an ATen kernel).
template <bool SRC, bool UPD, bool NRM>
__global__ void k_fas_update(int64_t n, float omega, const float* __restrict__ r, const float* __restrict__ src,
                             float* __restrict__ q, double* __restrict__ out) {
    const float w = fminf(fmaxf(omega, 0.0f), 1.0f);
    double s = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float ri = r[i];
        if (SRC) ri = ri + src[i];
        if (UPD) q[i] = q[i] + w * ri;
        if (NRM) s += (double)ri * (double)ri;
    }
    if (NRM) {
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        __shared__ double part[OPS_BLOCK / 64];
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        if (lane == 0) part[wv] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int i = 0; i < OPS_BLOCK / 64; ++i) t += part[i];
            out[blockIdx.x] = t;
        }
    }
}

inline dim3 grid2(int64_t n, int nv) { return dim3(ibh_grid(n, OPS_BLOCK) > 4096 ? 4096 : ibh_grid(n, OPS_BLOCK), nv); }

// launch of the accumulation kernels: per (row, field) for one field, per row over blocks of up to 4 fields otherwise
static inline void launch_accumulate(int32_t n_out, const int32_t* off, const int32_t* idx, const float* w,
                                     const int32_t* remap, const float* v, int64_t ldv, float* out, int64_t ldo, int nv) {
    if (nv == 1) {
        hipLaunchKernelGGL(k_accumulate, grid2(n_out, 1), dim3(OPS_BLOCK), 0, ibh_stream, n_out, off, idx, w, remap, v, ldv,
                           out, ldo);
    } else if (nv > 4 && nv <= 8) {
        // 5..8 fields (a state vector): indices and weights read ONCE for all of them (2.0 -> 1.3 ms per application of the
        // transfer operators of a 33.6 M-cell level to 6 fields)
        hipLaunchKernelGGL(k_accumulate_rows<8>, grid2(n_out, 1), dim3(OPS_BLOCK), 0, ibh_stream, n_out, off, idx, w, remap, v,
                           ldv, out, ldo, nv, (const float*)nullptr);
    } else {
        dim3 g = grid2(n_out, (nv + 3) / 4);
        hipLaunchKernelGGL(k_accumulate_rows<4>, g, dim3(OPS_BLOCK), 0, ibh_stream, n_out, off, idx, w, remap, v, ldv, out,
                           ldo, nv, (const float*)nullptr);
    }
}


}  // namespace

#define CHECK_DIM(p, dim) IBH_REQUIRE((p) && (dim) >= 1 && (dim) <= (p)->nd, "dim out of range or null partition")
#define CHECK_NV(nv) IBH_REQUIRE((nv) >= 1 && (nv) <= 65535, "nv out of range")

extern "C" {

int ibh_gather_rows(const int32_t* rows, int32_t n, const float* src, int nv, int64_t lds, float* dst, int64_t ldd) {
    CHECK_NV(nv);
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_gather, grid2(n, nv), dim3(OPS_BLOCK), 0, ibh_stream, rows, n, src, lds, dst, ldd);
    IBH_LAUNCH_CHECK();
    return 0;
}
int ibh_scatter_rows(const int32_t* rows, int32_t n, const float* src, int nv, int64_t lds, float* dst, int64_t ldd) {
    CHECK_NV(nv);
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_scatter, grid2(n, nv), dim3(OPS_BLOCK), 0, ibh_stream, rows, n, src, lds, dst, ldd);
    IBH_LAUNCH_CHECK();
    return 0;
}
int ibh_copy_rows(const int32_t* drows, const int32_t* srows, int32_t n, const float* src, int nv, int64_t lds,
                  float* dst, int64_t ldd) {
    CHECK_NV(nv);
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_copy_rows, grid2(n, nv), dim3(OPS_BLOCK), 0, ibh_stream, drows, srows, n, src, lds, dst, ldd);
    IBH_LAUNCH_CHECK();
    return 0;
}

int ibh_at_owners(const ibh_part* p, int dim, const float* u, int nv, int64_t ldu, float* out, int64_t ldo) {
    CHECK_DIM(p, dim);
    return ibh_gather_rows(p->dim[dim - 1].owners, p->dim[dim - 1].nf, u, nv, ldu, out, ldo);
}
int ibh_at_neighbors(const ibh_part* p, int dim, const float* u, int nv, int64_t ldu, float* out, int64_t ldo) {
    CHECK_DIM(p, dim);
    return ibh_gather_rows(p->dim[dim - 1].neighbors, p->dim[dim - 1].nf, u, nv, ldu, out, ldo);
}

int ibh_at_faces(const ibh_part* p, int dim, const float* u, int nv, int64_t ldu, float* out, int64_t ldo) {
    CHECK_DIM(p, dim);
    CHECK_NV(nv);
    const DimData& D = p->dim[dim - 1];
    if (D.nf == 0) return 0;
    hipLaunchKernelGGL(k_at_faces, grid2(D.nf, nv), dim3(OPS_BLOCK), 0, ibh_stream, D.nf, D.owners, D.neighbors,
                       p->spacing + (int64_t)(dim - 1) * p->nc, u, ldu, out, ldo);
    IBH_LAUNCH_CHECK();
    return 0;
}

int ibh_green_gauss(const ibh_part* p, int dim, const float* uf, int nv, int64_t ldf, float* out, int64_t ldo,
                    int uns) {
    CHECK_DIM(p, dim);
    CHECK_NV(nv);
    const DimData& D = p->dim[dim - 1];
    if (p->nc == 0) return 0;
    hipLaunchKernelGGL(k_green_gauss, grid2(p->nc, nv), dim3(OPS_BLOCK), 0, ibh_stream, p->nc, D.loff, D.lidx, D.roff,
                       D.ridx, p->spacing + (int64_t)(dim - 1) * p->nc, uf, ldf, out, ldo, uns);
    IBH_LAUNCH_CHECK();
    return 0;
}

int ibh_cell_gradient_all(const ibh_part* p, const float* u, int nv, int64_t ldu, float* out, int64_t ldo) {
    IBH_REQUIRE(p && u && out && (p->nd == 2 || p->nd == 3), "ibh_cell_gradient_all: bad argument");
    CHECK_NV(nv);
    if (p->nc == 0) return 0;
    GradDims G;
    for (int d = 0; d < p->nd; ++d) {
        G.d[d] = p->dim[d];
        G.h[d] = p->spacing + (int64_t)d * p->nc;
    }
    G.side = p->side;
    if (p->nd == 2)
        hipLaunchKernelGGL(k_cell_gradient_all<2>, grid2(p->nc, nv), dim3(OPS_BLOCK), 0, ibh_stream, p->nc, G, u, ldu, nv, out, ldo);
    else
        hipLaunchKernelGGL(k_cell_gradient_all<3>, grid2(p->nc, nv), dim3(OPS_BLOCK), 0, ibh_stream, p->nc, G, u, ldu, nv, out, ldo);
    IBH_LAUNCH_CHECK();
    return 0;
}

// face-list forms of ibh_shear_rate_of_velocity / ibh_wray_agarwal_of (ibh_fused.hip dispatches here on partitions
// without block structure)
int ibh_shear_rate_of_velocity_cells(const ibh_part* p, const float* vel, int64_t ldv, float* S, float* Gout, int64_t ldg) {
    const GradDims G = grad_dims(p);
    if (p->nd == 2)
        hipLaunchKernelGGL(k_shear_of_velocity_cells<2>, grid2(p->nc, 1), dim3(OPS_BLOCK), 0, ibh_stream, p->nc, G, vel, ldv, S, Gout, ldg);
    else
        hipLaunchKernelGGL(k_shear_of_velocity_cells<3>, grid2(p->nc, 1), dim3(OPS_BLOCK), 0, ibh_stream, p->nc, G, vel, ldv, S, Gout, ldg);
    IBH_LAUNCH_CHECK();
    return 0;
}
int ibh_wray_agarwal_of_cells(const ibh_part* p, const float* R, const float* S, float sigmaR, float C1, float kappa,
                              float* nut, float* nuR, float* Sout) {
    const GradDims G = grad_dims(p);
    if (p->nd == 2)
        hipLaunchKernelGGL(k_wray_agarwal_of_cells<2>, grid2(p->nc, 1), dim3(OPS_BLOCK), 0, ibh_stream, p->nc, G, R, S, sigmaR,
                           C1, kappa, nut, nuR, Sout);
    else
        hipLaunchKernelGGL(k_wray_agarwal_of_cells<3>, grid2(p->nc, 1), dim3(OPS_BLOCK), 0, ibh_stream, p->nc, G, R, S, sigmaR,
                           C1, kappa, nut, nuR, Sout);
    IBH_LAUNCH_CHECK();
    return 0;
}

int ibh_timestep_advection(ibh_part* p, const float* C, int64_t ldc, float scale, float* dt_dev) {
    IBH_REQUIRE(p && C && dt_dev && (p->nd == 2 || p->nd == 3) && p->nc > 0, "ibh_timestep_advection: bad argument");
    if (!p->march_tmp) {  // workgroup maxima
        IBH_HIP(hipMalloc((void**)&p->march_tmp, 8192 * sizeof(float)));
        p->march_tmp_n = 8192;
    }
    const GradDims G = grad_dims(p);
    // one cell per thread up to 2 M cells (every load of a cell in flight at once: the kernel is two dependent round trips,
    // not bandwidth; with 1 024 workgroups and 3-4 cells per thread it took 18.6 us at 0.87 M cells)
    const int nwg = std::min(ibh_grid(p->nc, OPS_BLOCK), 8192);
    const bool tiled = p->info[20] != 0;   // complete blocks in order (ibh_api.hip)
#define DT_LAUNCH(ND_, T_) \
    hipLaunchKernelGGL((k_timestep_advection<ND_, T_>), dim3(nwg), dim3(OPS_BLOCK), 0, ibh_stream, p->nc, G, C, ldc, p->march_tmp)
    if (p->nd == 2) {
        if (tiled) DT_LAUNCH(2, true);
        else DT_LAUNCH(2, false);
    } else {
        if (tiled) DT_LAUNCH(3, true);
        else DT_LAUNCH(3, false);
    }
#undef DT_LAUNCH
    hipLaunchKernelGGL(k_dt_from_partials, dim3(1), dim3(OPS_BLOCK), 0, ibh_stream, nwg, p->march_tmp, scale, dt_dev);
    IBH_LAUNCH_CHECK();
    return 0;
}

// ibh_bcset_apply(s, a) with ibh_timestep_advection(p, C, ldc, scale, dt_next) riding in its launches: the partial maxima
// beside the first launch of the set, the final reduction beside the second.  Same kernels' bodies, same results; sets with
// fewer than two launches take the separate launches for what is left.
// partials_done: the partial maxima are in p->march_tmp already (a caller that launched them elsewhere); the final reduction
// then rides in the FIRST launch of the set
int ibh_bcset_apply_with_dt(const ibh_bcset* s, float* a, ibh_part* p, const float* C, int64_t ldc, float scale,
                            float* dt_next, int partials_done) {
    IBH_REQUIRE(s && a && p && C && dt_next && (p->nd == 2 || p->nd == 3) && p->nc > 0, "ibh_bcset_apply_with_dt: bad argument");
    if (!p->march_tmp) {
        IBH_HIP(hipMalloc((void**)&p->march_tmp, 8192 * sizeof(float)));
        p->march_tmp_n = 8192;
    }
    const GradDims G = grad_dims(p);
    const int nwg_dt = std::min(ibh_grid(p->nc, OPS_BLOCK), 8192);
    const bool tiled = p->info[20] != 0;
    int stage = partials_done ? 1 : 0;   // 0: partial maxima not launched yet, 1: final reduction not launched yet, 2: done
    for (int lv = 0; lv < s->nlev; ++lv) {
        const int32_t g0 = s->seg[lv], g1 = s->seg[lv + 1];
        if (g1 == g0) continue;
        const int nwg = std::min(ibh_grid(g1 - g0, OPS_BLOCK), 2048);
        BcLaunch B{g0, g1, s->eta, s->w, s->value, s->off, s->donor, s->bidx, s->mode, s->direct[lv] ? s->ghost : nullptr,
                   s->ghost, s->gval};
        // the interpolation launch of the level
        if (stage == 0) {
#define BCDT_LAUNCH(ND_, T_)                                                                                              \
    hipLaunchKernelGGL((k_bcinterp_dt<ND_, T_>), dim3(nwg + nwg_dt), dim3(OPS_BLOCK), 0, ibh_stream, nwg, B, a, p->nc, G, C, \
                       ldc, p->march_tmp)
            if (p->nd == 2) {
                if (tiled) BCDT_LAUNCH(2, true);
                else BCDT_LAUNCH(2, false);
            } else {
                if (tiled) BCDT_LAUNCH(3, true);
                else BCDT_LAUNCH(3, false);
            }
#undef BCDT_LAUNCH
            stage = 1;
        } else if (stage == 1) {
            hipLaunchKernelGGL(k_bcinterp_dtfinal, dim3(nwg + 1), dim3(OPS_BLOCK), 0, ibh_stream, nwg, B, a, nwg_dt,
                               p->march_tmp, scale, dt_next);
            stage = 2;
        } else {
            hipLaunchKernelGGL(k_bcinterp_dtfinal, dim3(nwg), dim3(OPS_BLOCK), 0, ibh_stream, nwg, B, a, 0, p->march_tmp, scale,
                               dt_next);
        }
        if (s->direct[lv]) continue;
        // the scatter launch of the level
        if (stage == 1) {
            hipLaunchKernelGGL(k_bcscatter_dt, dim3(nwg + 1), dim3(OPS_BLOCK), 0, ibh_stream, nwg, B, a, nwg_dt, p->march_tmp,
                               scale, dt_next);
            stage = 2;
        } else {
            hipLaunchKernelGGL(k_bcscatter_dt, dim3(nwg), dim3(OPS_BLOCK), 0, ibh_stream, nwg, B, a, 0, p->march_tmp, scale,
                               dt_next);
        }
    }
    IBH_LAUNCH_CHECK();
    if (stage == 0) return ibh_timestep_advection(p, C, ldc, scale, dt_next);
    if (stage == 1) {
        hipLaunchKernelGGL(k_dt_from_partials, dim3(1), dim3(OPS_BLOCK), 0, ibh_stream, nwg_dt, p->march_tmp, scale, dt_next);
        IBH_LAUNCH_CHECK();
    }
    return 0;
}

int ibh_cell_gradient(const ibh_part* p, int dim, const float* u, int nv, int64_t ldu, float* out, int64_t ldo) {
    CHECK_DIM(p, dim);
    CHECK_NV(nv);
    if (p->nc == 0) return 0;
    hipLaunchKernelGGL(k_cell_gradient, grid2(p->nc, nv), dim3(OPS_BLOCK), 0, ibh_stream, p->nc, p->dim[dim - 1],
                       p->spacing + (int64_t)(dim - 1) * p->nc, u, ldu, out, ldo);
    IBH_LAUNCH_CHECK();
    return 0;
}

static int distances(const ibh_part* p, int dim, int mode, float* out) {
    CHECK_DIM(p, dim);
    const DimData& D = p->dim[dim - 1];
    if (D.nf == 0) return 0;
    hipLaunchKernelGGL(k_distances, grid2(D.nf, 1), dim3(OPS_BLOCK), 0, ibh_stream, D.nf, D.owners, D.neighbors,
                       p->spacing + (int64_t)(dim - 1) * p->nc, mode, out);
    IBH_LAUNCH_CHECK();
    return 0;
}
int ibh_face_distance(const ibh_part* p, int dim, float* out) { return distances(p, dim, 0, out); }
int ibh_owner_distance(const ibh_part* p, int dim, float* out) { return distances(p, dim, 1, out); }
int ibh_neighbor_distance(const ibh_part* p, int dim, float* out) { return distances(p, dim, 2, out); }

int ibh_face_gradient(const ibh_part* p, int dim, const float* u, int nv, int64_t ldu, float* out, int64_t ldo) {
    CHECK_DIM(p, dim);
    CHECK_NV(nv);
    const DimData& D = p->dim[dim - 1];
    if (D.nf == 0) return 0;
    hipLaunchKernelGGL(k_face_gradient, grid2(D.nf, nv), dim3(OPS_BLOCK), 0, ibh_stream, D.nf, D.owners, D.neighbors,
                       p->spacing + (int64_t)(dim - 1) * p->nc, u, ldu, out, ldo);
    IBH_LAUNCH_CHECK();
    return 0;
}

int ibh_jst_sensor(const ibh_part* p, int dim, const float* q, int nv, int64_t ldp, float* out, int64_t ldo) {
    IBH_REQUIRE(p && dim >= 0 && dim <= p->nd, "ibh_jst_sensor: dim out of range");
    CHECK_NV(nv);
    if (p->nc == 0) return 0;
    hipLaunchKernelGGL(k_jst, grid2(p->nc, nv), dim3(OPS_BLOCK), 0, ibh_stream, p->nc, p->nd, dim, p->dim[0], p->dim[1],
                       p->dim[2], p->spacing, q, ldp, out, ldo);
    IBH_LAUNCH_CHECK();
    return 0;
}

int ibh_muscl(const ibh_part* p, int dim, const float* u, const float* du, int nv, int64_t ld, const float* D,
              int high_order, float* uL, float* uR, int64_t ldf) {
    CHECK_DIM(p, dim);
    CHECK_NV(nv);
    const DimData& DD = p->dim[dim - 1];
    if (DD.nf == 0) return 0;
    hipLaunchKernelGGL(k_muscl, grid2(DD.nf, nv), dim3(OPS_BLOCK), 0, ibh_stream, DD.nf, DD.owners, DD.neighbors,
                       p->spacing + (int64_t)(dim - 1) * p->nc, u, du, ld, D, high_order, uL, uR, ldf);
    IBH_LAUNCH_CHECK();
    return 0;
}

int ibh_accumulate(const ibh_acc* a, const float* v, int nv, int64_t ldv, float* out, int64_t ldo) {
    IBH_REQUIRE(a, "ibh_accumulate: null accumulator");
    CHECK_NV(nv);
    if (a->n_out == 0) return 0;
    launch_accumulate(a->n_out, a->off, a->idx, a->w, (const int32_t*)nullptr, v, ldv, out, ldo, nv);
    IBH_LAUNCH_CHECK();
    return 0;
}

int ibh_accumulate_diff_add(const ibh_acc* a, const float* v, const float* v2, int nv, int64_t ldv, float* out,
                            int64_t ldo) {
    IBH_REQUIRE(a && v && v2 && out, "ibh_accumulate_diff_add: null argument");
    CHECK_NV(nv);
    if (a->n_out == 0) return 0;
    if (a->w && nv >= 2 && nv <= 8 && (int64_t)a->n_out >= 4 * (int64_t)a->n_in) {
        // many rows per donor (a prolongation): pack the donors' differences once, gather them 16 bytes at a time
        ibh_acc* am = const_cast<ibh_acc*>(a);
        if (!am->packed) IBH_HIP(hipMalloc((void**)&am->packed, sizeof(float) * 8 * (size_t)a->n_in));
        hipLaunchKernelGGL(k_pack_diff8, grid2((int64_t)a->n_in * 8, 1), dim3(OPS_BLOCK), 0, ibh_stream, a->n_in, nv, v, v2,
                           ldv, am->packed);
        const float4* d4 = reinterpret_cast<const float4*>(am->packed);
#define PACKED_LAUNCH(N)                                                                                               \
    hipLaunchKernelGGL(k_accumulate_packed_add<N>, grid2(a->n_out, 1), dim3(OPS_BLOCK), 0, ibh_stream, a->n_out, a->off,  \
                       a->idx, a->w, d4, out, ldo)
        switch (nv) {
            case 2: PACKED_LAUNCH(2); break;
            case 3: PACKED_LAUNCH(3); break;
            case 4: PACKED_LAUNCH(4); break;
            case 5: PACKED_LAUNCH(5); break;
            case 6: PACKED_LAUNCH(6); break;
            case 7: PACKED_LAUNCH(7); break;
            default: PACKED_LAUNCH(8); break;
        }
#undef PACKED_LAUNCH
        IBH_LAUNCH_CHECK();
        return 0;
    }
    for (int v0 = 0; v0 < nv; v0 += 8) {
        const int nb = nv - v0 < 8 ? nv - v0 : 8;
        hipLaunchKernelGGL((k_accumulate_rows<8, true>), grid2(a->n_out, 1), dim3(OPS_BLOCK), 0, ibh_stream, a->n_out, a->off,
                           a->idx, a->w, (const int32_t*)nullptr, v + (int64_t)v0 * ldv, ldv, out + (int64_t)v0 * ldo, ldo,
                           nb, v2 + (int64_t)v0 * ldv);
    }
    IBH_LAUNCH_CHECK();
    return 0;
}

int ibh_bc_interp(const ibh_bc* b, const float* a, int nv, int64_t lda, float* ia, int64_t ldi) {
    IBH_REQUIRE(b, "ibh_bc_interp: null boundary");
    CHECK_NV(nv);
    if (b->ng == 0) return 0;
    launch_accumulate(b->ng, b->interp.off, b->interp.idx, b->interp.w, b->image_domain, a, lda, ia, ldi, nv);
    IBH_LAUNCH_CHECK();
    return 0;
}

int ibh_bc_blend(const ibh_bc* b, float* a, int nv, int64_t lda, const float* ia, int64_t ldi, const float* ba,
                 int64_t ldb, const float* ba_const) {
    IBH_REQUIRE(b && (ba || ba_const), "ibh_bc_blend: need ba or ba_const");
    CHECK_NV(nv);
    if (b->ng == 0) return 0;
    float* dconst = nullptr;
    if (!ba) {
        IBH_HIP(hipMallocAsync((void**)&dconst, sizeof(float) * nv, ibh_stream));
        IBH_HIP(hipMemcpyAsync(dconst, ba_const, sizeof(float) * nv, hipMemcpyHostToDevice, ibh_stream));
    }
    hipLaunchKernelGGL(k_bc_blend, grid2(b->ng, nv), dim3(OPS_BLOCK), 0, ibh_stream, b->ng, b->ghost, b->eta, a, lda,
                       ia, ldi, ba, ldb, dconst);
    IBH_LAUNCH_CHECK();
    if (dconst) IBH_HIP(hipFreeAsync(dconst, ibh_stream));
    return 0;
}

int ibh_bc_apply(const ibh_bc* b, float* a, int nv, int64_t lda, int mode, const float* ba_const) {
    IBH_REQUIRE(b && (mode == 0 || mode == 1), "ibh_bc_apply: bad mode");
    CHECK_NV(nv);
    if (b->ng == 0) return 0;
    float* ia = nullptr;
    IBH_HIP(hipMallocAsync((void**)&ia, sizeof(float) * (size_t)b->ng * nv, ibh_stream));
    int rc = ibh_bc_interp(b, a, nv, lda, ia, b->ng);
    if (!rc) rc = (mode == 0) ? ibh_bc_blend(b, a, nv, lda, ia, b->ng, nullptr, 0, ba_const)
                              : ibh_bc_blend(b, a, nv, lda, ia, b->ng, ia, b->ng, nullptr);
    hipError_t e = hipFreeAsync(ia, ibh_stream);
    if (rc) return rc;
    IBH_HIP(e);
    return 0;
}

int ibh_axpy_clamped(int64_t n, float omega, const float* r, float* q) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_axpy_clamped, dim3(ibh_grid(n, OPS_BLOCK) > 2048 ? 2048 : ibh_grid(n, OPS_BLOCK)),
                       dim3(OPS_BLOCK), 0, ibh_stream, n, omega, r, q);
    IBH_LAUNCH_CHECK();
    return 0;
}
int ibh_axpy_clamped_sumsq(int64_t n, float omega, const float* r, float* q, double* out) {
    IBH_REQUIRE(out, "ibh_axpy_clamped_sumsq: null argument");
    double* part = red_scratch();
    IBH_REQUIRE(part, "ibh_axpy_clamped_sumsq: no scratch");
    if (n <= 0) {
        IBH_HIP(hipMemsetAsync(out, 0, sizeof(double), ibh_stream));
        return 0;
    }
    const int nwg = ibh_grid(n, OPS_BLOCK) > 2048 ? 2048 : ibh_grid(n, OPS_BLOCK);
    hipLaunchKernelGGL(k_axpy_clamped_sumsq, dim3(nwg), dim3(OPS_BLOCK), 0, ibh_stream, n, omega, r, q, part);
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(OPS_BLOCK), 0, ibh_stream, nwg, part, out);
    IBH_LAUNCH_CHECK();
    return 0;
}
int ibh_fas_update(int64_t n, float omega, const float* r, const float* source, float* q, double* out_sumsq) {
    IBH_REQUIRE(r || n <= 0, "ibh_fas_update: null residual");
    double* part = out_sumsq ? red_scratch() : nullptr;
    IBH_REQUIRE(!out_sumsq || part, "ibh_fas_update: no scratch");
    if (n <= 0) {
        if (out_sumsq) IBH_HIP(hipMemsetAsync(out_sumsq, 0, sizeof(double), ibh_stream));
        return 0;
    }
    if (!q && !out_sumsq) return 0;
    const int nwg = ibh_grid(n, OPS_BLOCK) > 2048 ? 2048 : ibh_grid(n, OPS_BLOCK);
#define FAS_LAUNCH(S, U, N) \
    hipLaunchKernelGGL((k_fas_update<S, U, N>), dim3(nwg), dim3(OPS_BLOCK), 0, ibh_stream, n, omega, r, source, q, part)
    if (source) {
        if (q && out_sumsq) FAS_LAUNCH(true, true, true);
        else if (q) FAS_LAUNCH(true, true, false);
        else FAS_LAUNCH(true, false, true);
    } else {
        if (q && out_sumsq) FAS_LAUNCH(false, true, true);
        else if (q) FAS_LAUNCH(false, true, false);
        else FAS_LAUNCH(false, false, true);
    }
#undef FAS_LAUNCH
    if (out_sumsq) hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(OPS_BLOCK), 0, ibh_stream, nwg, part, out_sumsq);
    IBH_LAUNCH_CHECK();
    return 0;
}
int ibh_axpy(int64_t n, float a, const float* x, float* y) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_axpy, dim3(ibh_grid(n, OPS_BLOCK) > 2048 ? 2048 : ibh_grid(n, OPS_BLOCK)), dim3(OPS_BLOCK), 0,
                       ibh_stream, n, a, x, y);
    IBH_LAUNCH_CHECK();
    return 0;
}
int ibh_sumsq(int64_t n, const float* x, double* out) {
    double* part = red_scratch();
    IBH_REQUIRE(out && part, "ibh_sumsq: null argument or no scratch");
    if (n <= 0) {
        IBH_HIP(hipMemsetAsync(out, 0, sizeof(double), ibh_stream));
        return 0;
    }
    const int nwg = ibh_grid(n, OPS_BLOCK) > 1024 ? 1024 : ibh_grid(n, OPS_BLOCK);
    hipLaunchKernelGGL(k_sumsq, dim3(nwg), dim3(OPS_BLOCK), 0, ibh_stream, n, x, part + 2048);
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(OPS_BLOCK), 0, ibh_stream, nwg, part + 2048, out);
    IBH_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
