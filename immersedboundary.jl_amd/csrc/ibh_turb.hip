// libibhip: turbulence closures of /root/reference/src/turbulence.jl as pointwise kernels (thread per cell / per
// ghost): wall_function :27-100, shear_rate :110-124, Smagorinsky_νSGS :135-138, standard_kϵ :176-196,
// Wray_Agarwal :222-241, Ducros_sensor :252-282, WALE_νSGS :291-337.  Float32, the reference's operation order
// (-ffp-contract=off); log/exp/pow come from the device math library (a few ulp from Julia's).
// Velocity gradients: an nd x nd table of device pointers, g[i*nd + j] = d u_i / d x_j (the reference's Matrix of vectors).
#include "ibh_common.h"

namespace {

constexpr int TB = 256;
constexpr float EPS32 = 1.1920929e-07f;

struct GradPtrs {
    const float* g[9];
};

__device__ __forceinline__ float von_karman(float yp, float kappa, float C) {
    return fminf(logf(fmaxf(yp, 1.0f)) / kappa + C, yp);  // :11-16
}

struct WallParams {
    float kappa, C, A, beta, betastar, D, Aplus, omega;
    int n_iter;
};

__device__ __forceinline__ void wall_point(float Rey, const WallParams& w, float& yp, float& up, float& mup, float& kp,
                                           float& dudy) {
    Rey = fminf(fmaxf(fabsf(Rey), EPS32), INFINITY);  // clamp(abs(Rey), eps, Inf32)
    yp = sqrtf(Rey);
    up = 0.0f;
    for (int it = 0; it < w.n_iter; ++it) {
        up = von_karman(yp, w.kappa, w.C);
        yp = w.omega * (Rey / up) + (1.0f - w.omega) * yp;
    }
    up = Rey / yp;
    const float e = 1.0f - expf(-yp / w.A);
    mup = w.kappa * yp * (e * e);
    dudy = 1.0f / (1.0f + mup);
    kp = fminf(yp * yp / (6.0f * w.betastar / w.beta - 2.0f), w.D * expf(-yp / w.Aplus));
}

__global__ void k_wall_rey(int64_t n, const float* __restrict__ Rey, WallParams w, float* __restrict__ yp,
                           float* __restrict__ up, float* __restrict__ mup, float* __restrict__ kp,
                           float* __restrict__ dudy) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float a, b, c, d, e;
        wall_point(Rey[i], w, a, b, c, d, e);
        yp[i] = a; up[i] = b; mup[i] = c; kp[i] = d; dudy[i] = e;
    }
}

__global__ void k_wall(int64_t n, const float* __restrict__ y, const float* __restrict__ u, const float* __restrict__ nu,
                       WallParams w, float* __restrict__ utau, float* __restrict__ nut, float* __restrict__ k,
                       float* __restrict__ omega, float* __restrict__ eps, float* __restrict__ dudn) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float yp, up, mup, kp, dudy;
        wall_point(u[i] * y[i] / nu[i], w, yp, up, mup, kp, dudy);
        const float ut = u[i] / up;
        const float nt = mup * nu[i];
        const float kk = kp * (ut * ut);
        const float om = kk / nt;
        utau[i] = ut;
        nut[i] = nt;
        k[i] = kk;
        omega[i] = om;
        eps[i] = w.betastar * om * kk;
        dudn[i] = dudy * (ut * ut) / nu[i];
    }
}

template <int ND>
__global__ void k_shear(int64_t n, GradPtrs G, float* __restrict__ S) {
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < ND; ++i)
#pragma unroll
            for (int j = 0; j < ND; ++j) {
                const float t = (G.g[i * ND + j][c] + G.g[j * ND + i][c]) / 2.0f;
                s = s + t * t;
            }
        S[c] = sqrtf(2.0f * s);
    }
}

__global__ void k_smagorinsky(int64_t n, const float* __restrict__ D, const float* __restrict__ S, float Cs,
                              float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float t = Cs * D[i];
        out[i] = t * t * S[i];
    }
}

__global__ void k_keps(int64_t n, const float* __restrict__ k, const float* __restrict__ e, const float* __restrict__ S,
                       float Cmu, float sk, float se, float C1, float C2, float* __restrict__ nuk, float* __restrict__ nue,
                       float* __restrict__ Sk, float* __restrict__ Se, float* __restrict__ nut) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float kk = k[i], ee = e[i];
        const float nt = Cmu * (kk * kk) / ee;
        const float Pk = nt * (S[i] * S[i]);
        nut[i] = nt;
        nuk[i] = nt / sk;
        nue[i] = nt / se;
        Sk[i] = Pk - ee;
        Se[i] = C1 * Pk * ee / kk - C2 * (ee * ee) / kk;
    }
}

template <int ND>
__global__ void k_wray_agarwal(int64_t n, const float* __restrict__ R, const float* __restrict__ S,
                               const float* __restrict__ gR, int64_t ldr, const float* __restrict__ gS, int64_t lds,
                               float sigmaR, float C1, float kappa, float* __restrict__ nut, float* __restrict__ nuR,
                               float* __restrict__ Sout) {
    const float C2 = sigmaR + C1 / (kappa * kappa);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float dot = gR[i] * gS[i];
#pragma unroll
        for (int d = 1; d < ND; ++d) dot = dot + gR[i + d * ldr] * gS[i + d * lds];
        const float r = R[i], s = S[i];
        const float src = C1 * r * s + C2 * dot * (r / (s + EPS32));
        nut[i] = r;
        nuR[i] = r * sigmaR;
        Sout[i] = fminf(src, 10.0f * r);
    }
}

template <int ND>
__global__ void k_ducros(int64_t n, GradPtrs G, float* __restrict__ out) {
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += (int64_t)gridDim.x * blockDim.x) {
        float div = 0.0f;
#pragma unroll
        for (int i = 0; i < ND; ++i) div = div + G.g[i * ND + i][c];
        const float div2 = div * div;
        float curl2;
        if (ND == 2) {
            const float w = G.g[1 * ND + 0][c] - G.g[0 * ND + 1][c];
            curl2 = w * w;
        } else {
            const float a = G.g[2 * ND + 1][c] - G.g[1 * ND + 2][c];
            const float b = G.g[0 * ND + 2][c] - G.g[2 * ND + 0][c];
            const float d = G.g[1 * ND + 0][c] - G.g[0 * ND + 1][c];
            curl2 = a * a + b * b + d * d;
        }
        out[c] = (div2 + EPS32) / (div2 + curl2 + EPS32);
    }
}

__global__ void k_wale(int64_t n, const float* __restrict__ Delta, GradPtrs G, float Cw, float* __restrict__ out) {
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += (int64_t)gridDim.x * blockDim.x) {
        float g[3][3], g2[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) g[i][j] = G.g[i * 3 + j][c];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float s = 0.0f;
#pragma unroll
                for (int k = 0; k < 3; ++k) s = s + g[i][k] * g[k][j];
                g2[i][j] = s;
            }
        float SS = 0.0f, SdSd = 0.0f;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float t = (g[i][j] + g[j][i]) / 2.0f;
                SS = SS + t * t;
                const float dlt = (i == j) ? (1.0f / 3.0f) : 0.0f;
                const float q = (g2[i][j] + g2[j][i]) / 2.0f - g2[i][j] * dlt;
                SdSd = SdSd + q * q;
            }
        const float D = Delta[c];
        out[c] = Cw * (D * D) * powf(SdSd, 1.5f) / (powf(SS, 2.5f) + powf(SdSd, 1.25f) + EPS32);
    }
}

// ---- transport of a scalar with variable diffusivity, all dimensions in one launch (thread per cell, face lists):
//   out = S + sum_d green_gauss(at_faces(nu + nuR, d) .* face_gradient(R, d) .- at_faces(vel_d .* R, d), d)
// the composition of closures.euler_wray_agarwal_residual (turbulence.jl:222-241 closes it) operation by operation -- same
// expressions, same order as the operator kernels of ibh_ops.hip (-ffp-contract=off), so the result is theirs bit for bit;
// a face's flux is evaluated by both of its cells instead of being written and read back.
struct TransportDims {
    DimData d[IBH_MAXD];
    const float* h[IBH_MAXD];
    const float* vel[IBH_MAXD];
    const int32_t* side;      // side table of the partition: sides with one face are evaluated from the cell across
};
__device__ __forceinline__ float tr_face_avg(float uo, float un, float ho, float hn) { return (uo * hn + un * ho) / (hn + ho); }
__device__ __forceinline__ float tr_flux_on(int32_t o, int32_t n, const float* __restrict__ h, const float* __restrict__ R,
                                            const float* __restrict__ nuR, const float* __restrict__ vel, float nu) {
    const float ho = h[o], hn = h[n];
    const float Ro = R[o], Rn = R[n];
    const float conv = tr_face_avg(vel[o] * Ro, vel[n] * Rn, ho, hn);       // at_faces(vel_d .* R)
    const float nuf = tr_face_avg(nu + nuR[o], nu + nuR[n], ho, hn);        // at_faces(nu .+ nuR)
    const float fd = (ho + hn) / 2.0f;                                      // face_distance
    const float fg = (Rn - Ro) / fd;                                        // face_gradient(R)
    return nuf * fg - conv;
}
__device__ __forceinline__ float tr_flux(const DimData& D, const float* __restrict__ h, const float* __restrict__ R,
                                         const float* __restrict__ nuR, const float* __restrict__ vel, float nu, int32_t f) {
    return tr_flux_on(D.owners[f], D.neighbors[f], h, R, nuR, vel, nu);
}
__device__ __forceinline__ float tr_mean(const int32_t* __restrict__ off, const int32_t* __restrict__ idx, int32_t c,
                                         const DimData& D, const float* __restrict__ h, const float* __restrict__ R,
                                         const float* __restrict__ nuR, const float* __restrict__ vel, float nu) {
    const int32_t b = off[c], e = off[c + 1];
    if (e == b) return 0.0f;
    const float w = 1.0f / (float)(e - b);
    float s = tr_flux(D, h, R, nuR, vel, nu, idx[b]) * w;
    for (int32_t k = b + 1; k < e; ++k) s = s + tr_flux(D, h, R, nuR, vel, nu, idx[k]) * w;
    return s;
}
template <int ND>
__global__ void k_scalar_transport(int32_t nc, TransportDims T, const float* __restrict__ R, const float* __restrict__ nuR,
                                   float nu, const float* __restrict__ S, float* __restrict__ out) {
    for (int64_t c = IBH_WG_X() * (int64_t)blockDim.x + threadIdx.x; c < nc; c += (int64_t)gridDim.x * blockDim.x) {
        float rt = S[c];
        int32_t sd[2 * ND];
#pragma unroll
        for (int s = 0; s < 2 * ND; ++s) sd[s] = T.side[(int64_t)s * nc + c];
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const int32_t l = sd[2 * d], r = sd[2 * d + 1];
            float ar, al;
            if (r >= 0) ar = tr_flux_on((int32_t)c, r, T.h[d], R, nuR, T.vel[d], nu) * 1.0f;
            else if (r == -2) ar = 0.0f;
            else ar = tr_mean(T.d[d].roff, T.d[d].ridx, (int32_t)c, T.d[d], T.h[d], R, nuR, T.vel[d], nu);
            if (l >= 0) al = tr_flux_on(l, (int32_t)c, T.h[d], R, nuR, T.vel[d], nu) * 1.0f;
            else if (l == -2) al = 0.0f;
            else al = tr_mean(T.d[d].loff, T.d[d].lidx, (int32_t)c, T.d[d], T.h[d], R, nuR, T.vel[d], nu);
            rt = rt + (ar - al) / T.h[d][c];
        }
        out[c] = rt;
    }
}

inline int tgrid(int64_t n) {
    int g = ibh_grid(n, TB);
    return g > 4096 ? 4096 : g;
}
inline WallParams wall_params(const float* p, int n_iter) {
    return WallParams{p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], n_iter};
}

}  // namespace

extern "C" {

int ibh_turb_wall_function_rey(int64_t n, const float* Rey, const float* params8, int n_iter, float* yplus, float* uplus,
                               float* muplus, float* kplus, float* dudy) {
    if (n <= 0) return 0;
    IBH_REQUIRE(Rey && params8 && yplus && uplus && muplus && kplus && dudy, "ibh_turb_wall_function_rey: null argument");
    hipLaunchKernelGGL(k_wall_rey, dim3(tgrid(n)), dim3(TB), 0, ibh_stream, n, Rey, wall_params(params8, n_iter), yplus,
                       uplus, muplus, kplus, dudy);
    IBH_LAUNCH_CHECK();
    return 0;
}
int ibh_turb_wall_function(int64_t n, const float* y, const float* u, const float* nu, const float* params8, int n_iter,
                           float* utau, float* nut, float* k, float* omega, float* eps, float* dudn) {
    if (n <= 0) return 0;
    IBH_REQUIRE(y && u && nu && params8 && utau && nut && k && omega && eps && dudn, "ibh_turb_wall_function: null argument");
    hipLaunchKernelGGL(k_wall, dim3(tgrid(n)), dim3(TB), 0, ibh_stream, n, y, u, nu, wall_params(params8, n_iter), utau,
                       nut, k, omega, eps, dudn);
    IBH_LAUNCH_CHECK();
    return 0;
}
static int grad_ptrs(int nd, const float* const* g, GradPtrs* G) {
    IBH_REQUIRE(g && (nd == 2 || nd == 3), "velocity gradient: nd x nd table of device pointers, nd = 2 or 3");
    for (int k = 0; k < nd * nd; ++k) {
        IBH_REQUIRE(g[k], "velocity gradient: null component");
        G->g[k] = g[k];
    }
    return 0;
}
int ibh_turb_shear_rate(int nd, int64_t n, const float* const* g, float* S) {
    if (n <= 0) return 0;
    GradPtrs G;
    int rc = grad_ptrs(nd, g, &G);
    if (rc) return rc;
    IBH_REQUIRE(S, "ibh_turb_shear_rate: null argument");
    if (nd == 2) hipLaunchKernelGGL(k_shear<2>, dim3(tgrid(n)), dim3(TB), 0, ibh_stream, n, G, S);
    else hipLaunchKernelGGL(k_shear<3>, dim3(tgrid(n)), dim3(TB), 0, ibh_stream, n, G, S);
    IBH_LAUNCH_CHECK();
    return 0;
}
int ibh_turb_smagorinsky(int64_t n, const float* Delta, const float* S, float Cs, float* out) {
    if (n <= 0) return 0;
    IBH_REQUIRE(Delta && S && out, "ibh_turb_smagorinsky: null argument");
    hipLaunchKernelGGL(k_smagorinsky, dim3(tgrid(n)), dim3(TB), 0, ibh_stream, n, Delta, S, Cs, out);
    IBH_LAUNCH_CHECK();
    return 0;
}
int ibh_turb_k_epsilon(int64_t n, const float* k, const float* eps, const float* S, const float* params5, float* nuk,
                       float* nue, float* Sk, float* Se, float* nut) {
    if (n <= 0) return 0;
    IBH_REQUIRE(k && eps && S && params5 && nuk && nue && Sk && Se && nut, "ibh_turb_k_epsilon: null argument");
    hipLaunchKernelGGL(k_keps, dim3(tgrid(n)), dim3(TB), 0, ibh_stream, n, k, eps, S, params5[0], params5[1], params5[2],
                       params5[3], params5[4], nuk, nue, Sk, Se, nut);
    IBH_LAUNCH_CHECK();
    return 0;
}
int ibh_turb_wray_agarwal(int nd, int64_t n, const float* R, const float* S, const float* gradR, int64_t ldr,
                          const float* gradS, int64_t lds, float sigmaR, float C1, float kappa, float* nut, float* nuR,
                          float* Sout) {
    if (n <= 0) return 0;
    IBH_REQUIRE(R && S && gradR && gradS && nut && nuR && Sout && (nd == 2 || nd == 3), "ibh_turb_wray_agarwal: bad argument");
    if (nd == 2)
        hipLaunchKernelGGL(k_wray_agarwal<2>, dim3(tgrid(n)), dim3(TB), 0, ibh_stream, n, R, S, gradR, ldr, gradS, lds, sigmaR,
                           C1, kappa, nut, nuR, Sout);
    else
        hipLaunchKernelGGL(k_wray_agarwal<3>, dim3(tgrid(n)), dim3(TB), 0, ibh_stream, n, R, S, gradR, ldr, gradS, lds, sigmaR,
                           C1, kappa, nut, nuR, Sout);
    IBH_LAUNCH_CHECK();
    return 0;
}
int ibh_scalar_transport(const ibh_part* p, const float* R, const float* nuR, float nu, const float* vel, int64_t ldv,
                         const float* S, float* out) {
    IBH_REQUIRE(p && R && nuR && vel && S && out && (p->nd == 2 || p->nd == 3), "ibh_scalar_transport: bad argument");
    if (p->nc == 0) return 0;
    {
        int done = 0;   // all-block 3-D partitions: the block kernel (ibh_fused.hip)
        const int rc = ibh_scalar_transport_blocks(p, R, nuR, nu, vel, ldv, S, out, &done);
        if (rc || done) return rc;
    }
    TransportDims T;
    for (int d = 0; d < p->nd; ++d) {
        T.d[d] = p->dim[d];
        T.h[d] = p->spacing + (int64_t)d * p->nc;
        T.vel[d] = vel + (int64_t)d * ldv;
    }
    T.side = p->side;
    if (p->nd == 2)
        hipLaunchKernelGGL(k_scalar_transport<2>, dim3(tgrid(p->nc)), dim3(TB), 0, ibh_stream, p->nc, T, R, nuR, nu, S, out);
    else
        hipLaunchKernelGGL(k_scalar_transport<3>, dim3(tgrid(p->nc)), dim3(TB), 0, ibh_stream, p->nc, T, R, nuR, nu, S, out);
    IBH_LAUNCH_CHECK();
    return 0;
}
int ibh_turb_ducros(int nd, int64_t n, const float* const* g, float* out) {
    if (n <= 0) return 0;
    GradPtrs G;
    int rc = grad_ptrs(nd, g, &G);
    if (rc) return rc;
    IBH_REQUIRE(out, "ibh_turb_ducros: null argument");
    if (nd == 2) hipLaunchKernelGGL(k_ducros<2>, dim3(tgrid(n)), dim3(TB), 0, ibh_stream, n, G, out);
    else hipLaunchKernelGGL(k_ducros<3>, dim3(tgrid(n)), dim3(TB), 0, ibh_stream, n, G, out);
    IBH_LAUNCH_CHECK();
    return 0;
}
int ibh_turb_wale(int64_t n, const float* Delta, const float* const* g, float Cw, float* out) {
    if (n <= 0) return 0;
    GradPtrs G;
    int rc = grad_ptrs(3, g, &G);
    if (rc) return rc;
    IBH_REQUIRE(Delta && out, "ibh_turb_wale: null argument");
    hipLaunchKernelGGL(k_wale, dim3(tgrid(n)), dim3(TB), 0, ibh_stream, n, Delta, G, Cw, out);
    IBH_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
