// libibhip: CFD pointwise physics (cfd.jl) as standalone kernels, one thread per row, literal
// Float32 arithmetic in the reference's evaluation order (-ffp-contract=off).  These exist so that
// operator-granularity closures (`CFD.inviscid_fluxes(fluid, PL, PR, dim)` etc.) run on device arrays;
// the fused sweeps of ibh_fused.hip inline the same formulas (ibh_flux.h).
#include "ibh_common.h"
#include "ibh_flux.h"

#define CFD_BLOCK 256

// ibh_set_tuning("viscous_per_cell", 1): ibh_viscous_residual as one thread per cell (A/B against the LDS-shared faces)
int ibh_viscous_per_cell = 0;
namespace {

__device__ __forceinline__ float sutherland(const ibh_fluid& f, float T) {
    T = fmaxf(T, 10.0f);
    // mu_ref * ((T/Tref)^(2/3)) * (Tref + S) / (T + S)     (cfd.jl:75, exponent as in the reference)
    // x^(2/3) = exp2(2/3 log2 x) on the transcendental unit (v_log_f32 / v_exp_f32, 1 ulp each; x = T / Tref in [0.03, 30]:
    // inside 4e-7 of the correctly rounded power) instead of the ~100 instructions of the library's powf -- a third of a
    // viscous face flux
    return f.mu_ref * __builtin_amdgcn_exp2f((2.0f / 3.0f) * __builtin_amdgcn_logf(T / f.Tref)) * (f.Tref + f.S) / (T + f.S);
}

__device__ __forceinline__ float conductivity(const ibh_fluid& f, float T) {
    float k = 0.0f * T;
    float Tp = 1.0f;
    for (int i = 0; i < f.nk; ++i) {
        k = k + f.k[i] * (i == 0 ? 1.0f : Tp);
        Tp = (i == 0) ? T : Tp * T;
    }
    return k;
}

__global__ void k_pointwise(ibh_fluid f, int mode, int64_t n, const float* __restrict__ T, float* __restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float t = T[i];
        out[i] = mode == 0 ? sqrtf(f.gamma * f.R * fmaxf(t, 10.0f)) : mode == 1 ? sutherland(f, t) : conductivity(f, t);
    }
}

template <int ND>
__global__ void k_p2s(ibh_fluid f, int64_t n, const float* __restrict__ P, int64_t ldp, float* __restrict__ Q,
                      int64_t ldq) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float p = P[i], T = fmaxf(P[i + ldp], 10.0f);
        float k = P[i + 2 * ldp] * P[i + 2 * ldp];
#pragma unroll
        for (int j = 1; j < ND; ++j) k = k + P[i + (2 + j) * ldp] * P[i + (2 + j) * ldp];
        k = k / 2.0f;
        float rho = p / (f.R * T);
        Q[i] = rho;
        Q[i + ldq] = rho * (f.R / (f.gamma - 1.0f) * T + k);
#pragma unroll
        for (int j = 0; j < ND; ++j) Q[i + (2 + j) * ldq] = rho * P[i + (2 + j) * ldp];
    }
}

template <int ND>
__global__ void k_s2p(ibh_fluid f, int64_t n, const float* __restrict__ Q, int64_t ldq, float* __restrict__ P,
                      int64_t ldp) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float rho = Q[i], E = Q[i + ldq];
        float u[ND];
        float k = 0.f;
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            u[j] = Q[i + (2 + j) * ldq] / rho;
            k = (j == 0) ? u[j] * u[j] : k + u[j] * u[j];
        }
        k = k / 2.0f;
        float p = (f.gamma - 1.0f) * (E - rho * k);
        P[i] = p;
        P[i + ldp] = fmaxf(p / (rho * f.R), 10.0f);
#pragma unroll
        for (int j = 0; j < ND; ++j) P[i + (2 + j) * ldp] = u[j];
    }
}

template <int ND>
__global__ void k_hll(ibh_fluid f, int dim0, int64_t n, const float* __restrict__ PL, const float* __restrict__ PR,
                      int64_t ld, float* __restrict__ F, int64_t ldf) {
    constexpr int NV = ND + 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float l[NV], r[NV];
        double Fd[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            l[v] = PL[i + v * ld];
            r[v] = PR[i + v * ld];
        }
        ibhf::hll_flux<ND>(l, r, dim0, f.R, f.gamma, Fd);
#pragma unroll
        for (int v = 0; v < NV; ++v) F[i + v * ldf] = (float)Fd[v];
    }
}

template <int ND>
__global__ void k_sensor_flux(ibh_fluid f, int dim0, int64_t n, const float* __restrict__ PL,
                              const float* __restrict__ PR, int64_t ld, const float* __restrict__ nuL,
                              const float* __restrict__ nuR, float* __restrict__ F, int64_t ldf) {
    constexpr int NV = ND + 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float l[NV], r[NV], UL[NV], UR[NV], dummyF[NV], un, a;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            l[v] = PL[i + v * ld];
            r[v] = PR[i + v * ld];
        }
        // UcL/UcR = primitive2state with the pressure added to the energy (cfd.jl:521-525)
        ibhf::side_state<ND>(l, dim0, f.R, f.gamma, UL, dummyF, un, a);
        ibhf::side_state<ND>(r, dim0, f.R, f.gamma, UR, dummyF, un, a);
        UL[1] = UL[1] + l[0];
        UR[1] = UR[1] + r[0];
        float Pm[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) Pm[v] = (l[v] + r[v]) / 2.0f;
        float u = Pm[2 + dim0];
        float am = sqrtf(f.gamma * f.R * fmaxf(Pm[1], 10.0f));
        float nu = fmaxf(nuL[i], nuR[i]);
        float diss = nu * (am + fabsf(u)) / 2.0f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            float Fv = (UL[v] + UR[v]) * u / 2.0f;
            if (v == 2 + dim0) Fv = Fv + Pm[0];
            F[i + v * ldf] = Fv + (UL[v] - UR[v]) * diss;
        }
    }
}

template <int ND>
struct GradPtrs {
    const float* g[ND];
};

template <int ND>
__global__ void k_viscous(ibh_fluid f, int dim0, int64_t n, const float* __restrict__ P, int64_t ldp, GradPtrs<ND> G,
                          int64_t ldg, const float* __restrict__ mu_t, float mu_t_const, float* __restrict__ F,
                          int64_t ldf) {
    constexpr int NV = ND + 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float T = P[i + ldp];
        float mu = sutherland(f, T) + (mu_t ? mu_t[i] : mu_t_const);
        float k = conductivity(f, T);
        // vel_grad(a, b) = d u_a / d x_b = Pgrad[b][:, 2 + a]   (cfd.jl:677)
        float vg[ND][ND];
        float divu = 0.f;
#pragma unroll
        for (int a = 0; a < ND; ++a)
#pragma unroll
            for (int b = 0; b < ND; ++b) vg[a][b] = G.g[b][i + (2 + a) * ldg];
#pragma unroll
        for (int a = 0; a < ND; ++a) divu = (a == 0) ? 0.f + vg[0][0] : divu + vg[a][a];
        float tau[ND];
#pragma unroll
        for (int j = 0; j < ND; ++j)
            tau[j] = ((vg[dim0][j] + vg[j][dim0]) - (dim0 == j ? 2.0f / 3.0f : 0.0f) * divu) * mu;
        float Fe = 0.0f + G.g[dim0][i + ldg] * k;
#pragma unroll
        for (int j = 0; j < ND; ++j) Fe = Fe + tau[j] * P[i + (2 + j) * ldp];
        F[i] = 0.0f;
        F[i + ldf] = Fe;
#pragma unroll
        for (int j = 0; j < ND; ++j) F[i + (2 + j) * ldf] = 0.0f + tau[j];
        (void)NV;
    }
}

// ---- the viscous part of a Navier-Stokes residual, all dimensions in one launch (thread per cell, side table / face lists):
//   R[:, v] += sum_d green_gauss(viscous_fluxes(fluid, at_faces(P, d), face_gradient(P, grad P, d), d; mu_t = at_faces(mu_t, d)), d)
// (cfd.jl:664-736 over ImmersedBoundary.jl:899-926, 1039-1069) -- the composition of closures.navier_stokes_wray_agarwal_residual
// operation by operation: same expressions, same order as k_at_faces / k_face_gradient / k_viscous / k_green_gauss and the
// three `R .+= ...` (-ffp-contract=off), so the result is theirs bit for bit; a face's flux is evaluated by both of its cells
// instead of nine face arrays per dimension being written and read back.
template <int ND>
struct ViscDims {
    DimData d[ND];
    const float* h[ND];
    const float* g[ND];       // cell gradients along each axis, column-major with leading dimension ldg; u_a in column gcol + a
    int gcol;                 // 2: gradients of P = [p T u v (w)]; 0: gradients of the velocity columns only
    const int32_t* side;
};
__device__ __forceinline__ float v_face_avg(float uo, float un, float ho, float hn) { return (uo * hn + un * ho) / (hn + ho); }
// a / d for several numerators over ONE denominator: the IEEE division the compiler expands `a / d` into (v_rcp, one
// Newton step on the reciprocal, quotient, two residual corrections -- AMDGPU's f32 fdiv lowering) with the part that depends
// on d alone done once.  Bit-identical to `a / d` whenever v_div_scale would not rescale, i.e. for |d| and |a / d| inside
// [2^-96, 2^96] (spacings, velocities, temperatures, their gradients); the operator kernels divide with `/`, and
// tests/test_gpu_cfd.py holds the two against each other bit for bit.  11 instead of 16.5 x 11 instructions per face.
struct Recip {
    float d, r;
};
__device__ __forceinline__ Recip recip_of(float d) {
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r0, 1.0f);
    return Recip{d, __builtin_fmaf(e, r0, r0)};
}
__device__ __forceinline__ float div_by(float a, const Recip& R) {
    float q = a * R.r;
    float e = __builtin_fmaf(-R.d, q, a);
    q = __builtin_fmaf(e, R.r, q);
    e = __builtin_fmaf(-R.d, q, a);
    return __builtin_fmaf(e, R.r, q);
}
// flux through the face between owner o and neighbour n normal to D0: F[0] = energy, F[1 + j] = momentum j
template <int ND, int D0>
__device__ __forceinline__ void visc_flux_on(const ibh_fluid& f, const ViscDims<ND>& V, int32_t o, int32_t n,
                                             const float* __restrict__ P, int64_t ldp, int64_t ldg,
                                             const float* __restrict__ mut, float* F) {
    const float ho = V.h[D0][o], hn = V.h[D0][n];
    const float fd = (ho + hn) / 2.0f;                                   // face_distance
    const Recip rs = recip_of(hn + ho), rf = recip_of(fd);
#define V_FACE_AVG(uo_, un_) div_by((uo_) * hn + (un_) * ho, rs)         /* v_face_avg(uo, un, ho, hn) */
    const float T = V_FACE_AVG(P[o + ldp], P[n + ldp]);                  // at_faces(P)[:, 2]
    float u[ND], vg[ND][ND];
#pragma unroll
    for (int a = 0; a < ND; ++a) {
        const float uo = P[o + (2 + a) * ldp], un = P[n + (2 + a) * ldp];
        u[a] = V_FACE_AVG(uo, un);
#pragma unroll
        for (int b = 0; b < ND; ++b)                                     // vel_grad(a, b) = Pgrad[b][:, 2 + a]
            vg[a][b] = b == D0 ? div_by(un - uo, rf)                     // face_gradient(part, P, dim)
                               : V_FACE_AVG(V.g[b][o + (V.gcol + a) * ldg], V.g[b][n + (V.gcol + a) * ldg]);  // at_faces(grad_b P)
    }
    const float gT = div_by(P[n + ldp] - P[o + ldp], rf);                // Pgrad[dim][:, 2]
    const float mu = sutherland(f, T) + V_FACE_AVG(mut[o], mut[n]);
#undef V_FACE_AVG
    const float k = conductivity(f, T);
    float divu = 0.f;
#pragma unroll
    for (int a = 0; a < ND; ++a) divu = (a == 0) ? 0.f + vg[0][0] : divu + vg[a][a];
    float Fe = 0.0f + gT * k;
#pragma unroll
    for (int j = 0; j < ND; ++j) {
        const float tau = ((vg[D0][j] + vg[j][D0]) - (D0 == j ? 2.0f / 3.0f : 0.0f) * divu) * mu;
        Fe = Fe + tau * u[j];
        F[1 + j] = 0.0f + tau;
    }
    F[0] = Fe;
}
template <int ND, int D0>
__device__ __forceinline__ void visc_mean(const ibh_fluid& f, const ViscDims<ND>& V, const int32_t* __restrict__ off,
                                          const int32_t* __restrict__ idx, int32_t c, const float* __restrict__ P,
                                          int64_t ldp, int64_t ldg, const float* __restrict__ mut, float* A) {
    const int32_t b = off[c], e = off[c + 1];
#pragma unroll
    for (int v = 0; v <= ND; ++v) A[v] = 0.0f;
    if (e == b) return;
    const float w = 1.0f / (float)(e - b);
    for (int32_t k = b; k < e; ++k) {
        const int32_t fc = idx[k];
        float F[ND + 1];
        visc_flux_on<ND, D0>(f, V, V.d[D0].owners[fc], V.d[D0].neighbors[fc], P, ldp, ldg, mut, F);
#pragma unroll
        for (int v = 0; v <= ND; ++v) A[v] = (k == b) ? F[v] * w : A[v] + F[v] * w;
    }
}
template <int ND, int D0>
__device__ __forceinline__ void visc_dim(const ibh_fluid& f, const ViscDims<ND>& V, int32_t nc, int32_t c,
                                         const float* __restrict__ P, int64_t ldp, int64_t ldg,
                                         const float* __restrict__ mut, float* acc) {
    const int32_t l = V.side[(int64_t)(2 * D0) * nc + c], r = V.side[(int64_t)(2 * D0 + 1) * nc + c];
    float ar[ND + 1], al[ND + 1];
    if (r >= 0) {
        visc_flux_on<ND, D0>(f, V, c, r, P, ldp, ldg, mut, ar);
#pragma unroll
        for (int v = 0; v <= ND; ++v) ar[v] = ar[v] * 1.0f;
    } else if (r == -2) {
#pragma unroll
        for (int v = 0; v <= ND; ++v) ar[v] = 0.0f;
    } else visc_mean<ND, D0>(f, V, V.d[D0].roff, V.d[D0].ridx, c, P, ldp, ldg, mut, ar);
    if (l >= 0) {
        visc_flux_on<ND, D0>(f, V, l, c, P, ldp, ldg, mut, al);
#pragma unroll
        for (int v = 0; v <= ND; ++v) al[v] = al[v] * 1.0f;
    } else if (l == -2) {
#pragma unroll
        for (int v = 0; v <= ND; ++v) al[v] = 0.0f;
    } else visc_mean<ND, D0>(f, V, V.d[D0].loff, V.d[D0].lidx, c, P, ldp, ldg, mut, al);
    const float hc = V.h[D0][c];
#pragma unroll
    for (int v = 0; v <= ND; ++v) acc[v] = acc[v] + (ar[v] - al[v]) / hc;   // R .+= green_gauss(...), one dimension after the other
}
template <int ND>
__global__ __launch_bounds__(CFD_BLOCK) void k_viscous_residual(ibh_fluid f, int32_t nc, ViscDims<ND> V,
                                                                const float* __restrict__ P, int64_t ldp, int64_t ldg,
                                                                const float* __restrict__ mut, float* __restrict__ R,
                                                                int64_t ldr) {
    for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < nc; c += (int64_t)gridDim.x * blockDim.x) {
        float acc[ND + 1];
#pragma unroll
        for (int v = 0; v <= ND; ++v) acc[v] = R[c + (1 + v) * ldr];
        visc_dim<ND, 0>(f, V, nc, (int32_t)c, P, ldp, ldg, mut, acc);
        visc_dim<ND, 1>(f, V, nc, (int32_t)c, P, ldp, ldg, mut, acc);
        if constexpr (ND == 3) visc_dim<ND, 2>(f, V, nc, (int32_t)c, P, ldp, ldg, mut, acc);
#pragma unroll
        for (int v = 0; v <= ND; ++v) R[c + (1 + v) * ldr] = acc[v];
    }
}

// The same sum with every face evaluated ONCE per workgroup (512 consecutive cells).  Phase A: each thread evaluates the
// flux through the "right" face of its cell in every direction and leaves it in LDS with the index of the cell on the other
// side.  The "left" face of a cell is its left neighbour's right face -- visc_flux_on(l, c) is the same call with the same
// arguments in both threads, so the sum is bit-identical to k_viscous_residual's -- and is taken from LDS when the neighbour
// is in the workgroup and names this cell.  The left faces that are not (block rims, level jumps, boundary faces: with 8^3
// blocks tiled in order one plane of 64 per direction) are compacted into one task list per direction and evaluated in phase B
// by whole waves (threads 128 d .. 128 d + 127 take direction d: wave-uniform), not by the eight scattered lanes of every
// wave that own them -- 27 wave-evaluations of the flux per 8 waves instead of 48.  The task lists are made first (from the
// side table alone), so that phase B follows phase A in the same waves without a barrier.  Tasks beyond 128 per direction
// (partitions that are not tiled) are evaluated in place, as k_viscous_residual does.
#define VISC_WG 512
#define VISC_CAP 128
template <int ND, int D0>
__device__ __forceinline__ void visc_right(const ibh_fluid& f, const ViscDims<ND>& V, int32_t nc, int32_t c,
                                           const float* __restrict__ P, int64_t ldp, int64_t ldg,
                                           const float* __restrict__ mut, float* ar, int32_t& r) {
    r = V.side[(int64_t)(2 * D0 + 1) * nc + c];
    if (r >= 0) {
        visc_flux_on<ND, D0>(f, V, c, r, P, ldp, ldg, mut, ar);
    } else if (r == -2) {
#pragma unroll
        for (int v = 0; v <= ND; ++v) ar[v] = 0.0f;
    } else visc_mean<ND, D0>(f, V, V.d[D0].roff, V.d[D0].ridx, c, P, ldp, ldg, mut, ar);
}
template <int ND, int D0>
__device__ __forceinline__ void visc_left(const ibh_fluid& f, const ViscDims<ND>& V, int32_t l, int32_t c,
                                          const float* __restrict__ P, int64_t ldp, int64_t ldg,
                                          const float* __restrict__ mut, float* al) {
    if (l >= 0) {
        visc_flux_on<ND, D0>(f, V, l, c, P, ldp, ldg, mut, al);
    } else if (l == -2) {
#pragma unroll
        for (int v = 0; v <= ND; ++v) al[v] = 0.0f;
    } else visc_mean<ND, D0>(f, V, V.d[D0].loff, V.d[D0].lidx, c, P, ldp, ldg, mut, al);
}
template <int ND>
struct ViscShared {
    float Fs[ND][ND + 1][VISC_WG];      // right-face fluxes of the workgroup's cells
    int32_t Rs[ND][VISC_WG];            // the cell on the other side of each
    float Fl[ND][ND + 1][VISC_CAP];     // left-face fluxes evaluated in phase B
    int32_t list[ND][VISC_CAP];         // their cells (thread ids)
    int32_t cnt[ND];
};
// phase B of direction D0: task i of the list
template <int ND, int D0>
__device__ __forceinline__ void visc_task(const ibh_fluid& f, const ViscDims<ND>& V, int32_t nc, int32_t base, int i,
                                          const float* __restrict__ P, int64_t ldp, int64_t ldg,
                                          const float* __restrict__ mut, ViscShared<ND>& sh) {
    const int n = sh.cnt[D0] < VISC_CAP ? sh.cnt[D0] : VISC_CAP;
    if (i >= n) return;
    const int32_t c = base + sh.list[D0][i];
    float al[ND + 1];
    visc_left<ND, D0>(f, V, V.side[(int64_t)(2 * D0) * nc + c], c, P, ldp, ldg, mut, al);
#pragma unroll
    for (int v = 0; v <= ND; ++v) sh.Fl[D0][v][i] = al[v];
}
// phase C of direction D0: slot -1 = the neighbour's right face in LDS, >= 0 = task slot, -2 = evaluate here
template <int ND, int D0>
__device__ __forceinline__ void visc_accumulate(const ibh_fluid& f, const ViscDims<ND>& V, int32_t nc, int32_t c, int32_t l,
                                                int32_t base, int slot, const float* __restrict__ P, int64_t ldp,
                                                int64_t ldg, const float* __restrict__ mut, const ViscShared<ND>& sh,
                                                const float* ar, float* acc) {
    float al[ND + 1];
    if (slot == -1) {
#pragma unroll
        for (int v = 0; v <= ND; ++v) al[v] = sh.Fs[D0][v][l - base];
    } else if (slot >= 0) {
#pragma unroll
        for (int v = 0; v <= ND; ++v) al[v] = sh.Fl[D0][v][slot];
    } else visc_left<ND, D0>(f, V, l, c, P, ldp, ldg, mut, al);
    const Recip rh = recip_of(V.h[D0][c]);
#pragma unroll
    for (int v = 0; v <= ND; ++v) acc[v] = acc[v] + div_by(ar[v] - al[v], rh);
}
// Eight waves per SIMD (64 VGPRs, 40 spilled words; four workgroups = 154 KB of LDS per CU): the phases of a workgroup wait
// on one another at barriers, and what fills the gaps is other workgroups -- 246 us with the 94 registers the compiler takes
// unconstrained (two workgroups per CU), 214 at six waves per SIMD (79 registers, no spill, three workgroups), **205 at
// eight**, at 4.56 M cells on one box, alternating builds.
#ifndef VISC_WAVES
#define VISC_WAVES 8
#endif
#define VISC_ATTR __attribute__((amdgpu_waves_per_eu(VISC_WAVES, VISC_WAVES)))
template <int ND>
__global__ __launch_bounds__(VISC_WG) VISC_ATTR void k_viscous_residual_shared(ibh_fluid f, int32_t nc, ViscDims<ND> V,
                                                                     const float* __restrict__ P, int64_t ldp, int64_t ldg,
                                                                     const float* __restrict__ mut, float* __restrict__ R,
                                                                     int64_t ldr) {
    __shared__ ViscShared<ND> sh;
    const int tid = threadIdx.x;
    {   // one chunk of 512 cells per workgroup, every XCD a contiguous run of chunks: the neighbour lines of a rim cell (a
        // block's rim gathers 99 KB of lines next to 30 KB of its own) are then in the L2 that read them as own cells --
        // round-robin placement fetched 2.3 x the algorithmic bytes from HBM
        const int32_t ch = ibh_xcd_chunk(blockIdx.x, gridDim.x);
        const int32_t base = ch * VISC_WG, c = base + tid;
        const bool on = c < nc;
        float ar[ND][ND + 1];
        int32_t r[ND], l[ND];
        int slot[ND];
        if (tid < ND) sh.cnt[tid] = 0;
        __syncthreads();
        if (on) {   // left faces whose neighbour is outside the workgroup's cells become tasks (the others are expected in LDS)
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                l[d] = V.side[(int64_t)(2 * d) * nc + c];
                const int32_t k = l[d] - base;
                if (l[d] >= 0 && k >= 0 && k < VISC_WG) slot[d] = -1;
                else {
                    const int s = atomicAdd(&sh.cnt[d], 1);
                    slot[d] = s < VISC_CAP ? s : -2;
                    if (s < VISC_CAP) sh.list[d][s] = tid;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int d = 0; d < ND; ++d) r[d] = -1;
        if (on) {   // phase A: the right faces
            visc_right<ND, 0>(f, V, nc, c, P, ldp, ldg, mut, ar[0], r[0]);
            visc_right<ND, 1>(f, V, nc, c, P, ldp, ldg, mut, ar[1], r[1]);
            if constexpr (ND == 3) visc_right<ND, 2>(f, V, nc, c, P, ldp, ldg, mut, ar[2], r[2]);
#pragma unroll
            for (int d = 0; d < ND; ++d)
#pragma unroll
                for (int v = 0; v <= ND; ++v) sh.Fs[d][v][tid] = ar[d][v];
        }
#pragma unroll
        for (int d = 0; d < ND; ++d) sh.Rs[d][tid] = r[d];
        {   // phase B, no barrier in between: wave w < ND takes the first 64 tasks of direction w, wave ND + w the next 64
            const int w = tid >> 6, d = w % ND, i = (w / ND) * 64 + (tid & 63);
            if (w < 2 * ND) {
                if (d == 0) visc_task<ND, 0>(f, V, nc, base, i, P, ldp, ldg, mut, sh);
                else if (d == 1) visc_task<ND, 1>(f, V, nc, base, i, P, ldp, ldg, mut, sh);
                else visc_task<ND, ND == 3 ? 2 : 0>(f, V, nc, base, i, P, ldp, ldg, mut, sh);
            }
        }
        __syncthreads();
        if (on) {   // phase C: R .+= green_gauss(...), one dimension after the other
            float acc[ND + 1];
#pragma unroll
            for (int v = 0; v <= ND; ++v) acc[v] = R[c + (1 + v) * ldr];
#pragma unroll
            for (int d = 0; d < ND; ++d)   // a neighbour in the workgroup that does not name this cell as its right one: here
                if (slot[d] == -1 && sh.Rs[d][l[d] - base] != c) slot[d] = -2;
            visc_accumulate<ND, 0>(f, V, nc, c, l[0], base, slot[0], P, ldp, ldg, mut, sh, ar[0], acc);
            visc_accumulate<ND, 1>(f, V, nc, c, l[1], base, slot[1], P, ldp, ldg, mut, sh, ar[1], acc);
            if constexpr (ND == 3) visc_accumulate<ND, 2>(f, V, nc, c, l[2], base, slot[2], P, ldp, ldg, mut, sh, ar[2], acc);
#pragma unroll
            for (int v = 0; v <= ND; ++v) R[c + (1 + v) * ldr] = acc[v];
        }
    }
}

inline dim3 visc_grid(int64_t n) { return dim3((unsigned)((n + VISC_WG - 1) / VISC_WG)); }
inline dim3 grid1(int64_t n) { int g = ibh_grid(n, CFD_BLOCK); return dim3(g > 4096 ? 4096 : g); }

// FlowBC call, cfd.jl:243-300: characteristic-style boundary state from the image-point primitives
template <int ND>
__global__ void k_flow_bc(ibh_fluid f, int64_t n, const float* __restrict__ P, int64_t ldp,
                          const float* __restrict__ nrm, int64_t ldn, float pinf, float Tinf, float u0, float u1, float u2,
                          int normal_flow, const float* __restrict__ imd, const float* __restrict__ dudn, float transp,
                          const float* __restrict__ transp_v, float* __restrict__ out, int64_t ldo) {
    const float uinf[3] = {u0, u1, u2};
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float u[ND], nn[ND];
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            u[j] = P[i + (2 + j) * ldp];
            nn[j] = nrm[i + j * ldn];
        }
        const float p = P[i], T = P[i + ldp];
        float un, cur = u[0] * nn[0];
#pragma unroll
        for (int j = 1; j < ND; ++j) cur = cur + u[j] * nn[j];
        if (normal_flow) {
            un = u0;
        } else {
            un = nn[0] * uinf[0];
#pragma unroll
            for (int j = 1; j < ND; ++j) un = un + nn[j] * uinf[j];
        }
        const float a = sqrtf(f.gamma * f.R * fmaxf(T, 10.0f));
        const float M = fabsf(un) / a;
        // (un >= 0) * ((M > 1) * p_inf + (M <= 1) * p) + (un < 0) * ((M > 1) * p + (M <= 1) * p_inf)
        const float pb = (un >= 0.0f) ? (M > 1.0f ? pinf : p) : (M > 1.0f ? p : pinf);
        const float Tb = (un > 0.0f) ? Tinf : T;
        float ub[ND];
        if (normal_flow) {
            const float tr = transp_v ? transp_v[i] : transp;
            const float d = un - cur + tr;
#pragma unroll
            for (int j = 0; j < ND; ++j) ub[j] = u[j] + nn[j] * d;
        } else {
#pragma unroll
            for (int j = 0; j < ND; ++j) ub[j] = (un < 0.0f) ? u[j] : uinf[j];
        }
        if (dudn) {
            float V = ub[0] * ub[0];
#pragma unroll
            for (int j = 1; j < ND; ++j) V = V + ub[j] * ub[j];
            V = sqrtf(V) + 1.1920929e-07f;
            const float sc = (V - dudn[i] * imd[i]) / V;
#pragma unroll
            for (int j = 0; j < ND; ++j) ub[j] = ub[j] * sc;
        }
        out[i] = pb;
        out[i + ldo] = Tb;
#pragma unroll
        for (int j = 0; j < ND; ++j) out[i + (2 + j) * ldo] = ub[j];
    }
}

// CFD.JST_sensor(Pim1, Pi, Pip1), cfd.jl:563-573: (|Pim1 + Pip1 - 2 Pi| + eps) / (|Pim1 - Pi| + |Pip1 - Pi| + eps), eps = 1f-14
__global__ __launch_bounds__(CFD_BLOCK) void k_jst3(int64_t n, const float* __restrict__ a, const float* __restrict__ b,
                                                    const float* __restrict__ c, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float e = 1e-14f, pm = a[i], p0 = b[i], pp = c[i];
    out[i] = (fabsf(pm + pp - 2.0f * p0) + e) / (fabsf(pm - p0) + fabsf(pp - p0) + e);
}
// CFD.shock_sensor(velocity_gradients), cfd.jl:575-617: (div^2 + eps) / (div^2 + |curl|^2 + eps); g[i * nd + j] = d u_i / d x_j
struct ShockGradPtrs {
    const float* g[9];
};
template <int ND>
__global__ __launch_bounds__(CFD_BLOCK) void k_shock(int64_t n, ShockGradPtrs G, float* __restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    float divu = 0.0f, vort2 = 0.0f;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        const int in = (i + 1) % ND, inn = (in + 1) % ND;
        divu = divu + G.g[i * ND + i][p];
        const float w = G.g[inn * ND + in][p] - G.g[in * ND + inn][p];
        vort2 = vort2 + w * w;
    }
    divu = divu * divu;
    out[p] = (divu + 1e-14f) / (divu + vort2 + 1e-14f);
}
}  // namespace

#define CHECK_ND(nd, dim) IBH_REQUIRE(((nd) == 2 || (nd) == 3) && (dim) >= 1 && (dim) <= (nd), "bad nd/dim")

extern "C" {

static int pointwise(const ibh_fluid* f, int mode, int64_t n, const float* T, float* out) {
    IBH_REQUIRE(f && T && out, "ibh_cfd: null argument");
    IBH_REQUIRE(f->nk >= 0 && f->nk <= 4, "ibh_cfd: nk out of range");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_pointwise, grid1(n), dim3(CFD_BLOCK), 0, ibh_stream, *f, mode, n, T, out);
    IBH_LAUNCH_CHECK();
    return 0;
}
int ibh_cfd_speed_of_sound(const ibh_fluid* f, int64_t n, const float* T, float* a) { return pointwise(f, 0, n, T, a); }
int ibh_cfd_dynamic_viscosity(const ibh_fluid* f, int64_t n, const float* T, float* mu) { return pointwise(f, 1, n, T, mu); }
int ibh_cfd_heat_conductivity(const ibh_fluid* f, int64_t n, const float* T, float* k) { return pointwise(f, 2, n, T, k); }

int ibh_cfd_primitive2state(const ibh_fluid* f, int nd, int64_t n, const float* P, int64_t ldp, float* Q, int64_t ldq) {
    IBH_REQUIRE(f && P && Q, "ibh_cfd_primitive2state: null argument");
    CHECK_ND(nd, 1);
    if (n <= 0) return 0;
    if (nd == 2) hipLaunchKernelGGL(k_p2s<2>, grid1(n), dim3(CFD_BLOCK), 0, ibh_stream, *f, n, P, ldp, Q, ldq);
    else hipLaunchKernelGGL(k_p2s<3>, grid1(n), dim3(CFD_BLOCK), 0, ibh_stream, *f, n, P, ldp, Q, ldq);
    IBH_LAUNCH_CHECK();
    return 0;
}

int ibh_cfd_state2primitive(const ibh_fluid* f, int nd, int64_t n, const float* Q, int64_t ldq, float* P, int64_t ldp) {
    IBH_REQUIRE(f && P && Q, "ibh_cfd_state2primitive: null argument");
    CHECK_ND(nd, 1);
    if (n <= 0) return 0;
    if (nd == 2) hipLaunchKernelGGL(k_s2p<2>, grid1(n), dim3(CFD_BLOCK), 0, ibh_stream, *f, n, Q, ldq, P, ldp);
    else hipLaunchKernelGGL(k_s2p<3>, grid1(n), dim3(CFD_BLOCK), 0, ibh_stream, *f, n, Q, ldq, P, ldp);
    IBH_LAUNCH_CHECK();
    return 0;
}

int ibh_cfd_inviscid_fluxes_hll(const ibh_fluid* f, int nd, int dim, int64_t n, const float* PL, const float* PR,
                                int64_t ld, float* F, int64_t ldf) {
    IBH_REQUIRE(f && PL && PR && F, "ibh_cfd_inviscid_fluxes_hll: null argument");
    CHECK_ND(nd, dim);
    if (n <= 0) return 0;
    if (nd == 2) hipLaunchKernelGGL(k_hll<2>, grid1(n), dim3(CFD_BLOCK), 0, ibh_stream, *f, dim - 1, n, PL, PR, ld, F, ldf);
    else hipLaunchKernelGGL(k_hll<3>, grid1(n), dim3(CFD_BLOCK), 0, ibh_stream, *f, dim - 1, n, PL, PR, ld, F, ldf);
    IBH_LAUNCH_CHECK();
    return 0;
}

int ibh_cfd_inviscid_fluxes_sensor(const ibh_fluid* f, int nd, int dim, int64_t n, const float* PL, const float* PR,
                                   int64_t ld, const float* nuL, const float* nuR, float* F, int64_t ldf) {
    IBH_REQUIRE(f && PL && PR && nuL && nuR && F, "ibh_cfd_inviscid_fluxes_sensor: null argument");
    CHECK_ND(nd, dim);
    if (n <= 0) return 0;
    if (nd == 2)
        hipLaunchKernelGGL(k_sensor_flux<2>, grid1(n), dim3(CFD_BLOCK), 0, ibh_stream, *f, dim - 1, n, PL, PR, ld, nuL,
                           nuR, F, ldf);
    else
        hipLaunchKernelGGL(k_sensor_flux<3>, grid1(n), dim3(CFD_BLOCK), 0, ibh_stream, *f, dim - 1, n, PL, PR, ld, nuL,
                           nuR, F, ldf);
    IBH_LAUNCH_CHECK();
    return 0;
}

int ibh_cfd_viscous_fluxes(const ibh_fluid* f, int nd, int dim, int64_t n, const float* P, int64_t ldp,
                           const float* const* Pgrad, int64_t ldg, const float* mu_t, float mu_t_const, float* F,
                           int64_t ldf) {
    IBH_REQUIRE(f && P && Pgrad && F, "ibh_cfd_viscous_fluxes: null argument");
    IBH_REQUIRE(f->nk >= 0 && f->nk <= 4, "ibh_cfd: nk out of range");
    CHECK_ND(nd, dim);
    if (n <= 0) return 0;
    if (nd == 2) {
        GradPtrs<2> g{{Pgrad[0], Pgrad[1]}};
        hipLaunchKernelGGL(k_viscous<2>, grid1(n), dim3(CFD_BLOCK), 0, ibh_stream, *f, dim - 1, n, P, ldp, g, ldg, mu_t,
                           mu_t_const, F, ldf);
    } else {
        GradPtrs<3> g{{Pgrad[0], Pgrad[1], Pgrad[2]}};
        hipLaunchKernelGGL(k_viscous<3>, grid1(n), dim3(CFD_BLOCK), 0, ibh_stream, *f, dim - 1, n, P, ldp, g, ldg, mu_t,
                           mu_t_const, F, ldf);
    }
    IBH_LAUNCH_CHECK();
    return 0;
}

int ibh_viscous_residual(const ibh_part* p, const ibh_fluid* f, const float* P, int64_t ldp, const float* const* Pgrad,
                         int64_t ldg, int grad_vel_col, const float* mu_t, float* R, int64_t ldr) {
    IBH_REQUIRE(p && f && P && Pgrad && mu_t && R && (p->nd == 2 || p->nd == 3), "ibh_viscous_residual: bad argument");
    IBH_REQUIRE(grad_vel_col == 0 || grad_vel_col == 2, "ibh_viscous_residual: grad_vel_col is 2 (gradients of P) or 0 (of the velocities)");
    IBH_REQUIRE(f->nk >= 0 && f->nk <= 4, "ibh_cfd: nk out of range");
    IBH_REQUIRE(p->side, "ibh_viscous_residual: the partition has no side table");
    if (p->nc == 0) return 0;
    if (p->nd == 2) {
        ViscDims<2> V;
        for (int d = 0; d < 2; ++d) {
            IBH_REQUIRE(Pgrad[d], "ibh_viscous_residual: null gradient array");
            V.d[d] = p->dim[d];
            V.h[d] = p->spacing + (int64_t)d * p->nc;
            V.g[d] = Pgrad[d];
        }
        V.side = p->side;
        V.gcol = grad_vel_col;
        if (ibh_viscous_per_cell)   // (A/B: one thread per cell, both faces of every direction evaluated by it)
            hipLaunchKernelGGL(k_viscous_residual<2>, grid1(p->nc), dim3(CFD_BLOCK), 0, ibh_stream, *f, p->nc, V, P, ldp,
                               ldg, mu_t, R, ldr);
        else
            hipLaunchKernelGGL(k_viscous_residual_shared<2>, visc_grid(p->nc), dim3(VISC_WG), 0, ibh_stream, *f, p->nc, V,
                               P, ldp, ldg, mu_t, R, ldr);
    } else {
        ViscDims<3> V;
        for (int d = 0; d < 3; ++d) {
            IBH_REQUIRE(Pgrad[d], "ibh_viscous_residual: null gradient array");
            V.d[d] = p->dim[d];
            V.h[d] = p->spacing + (int64_t)d * p->nc;
            V.g[d] = Pgrad[d];
        }
        V.side = p->side;
        V.gcol = grad_vel_col;
        if (ibh_viscous_per_cell)   // (A/B: one thread per cell, both faces of every direction evaluated by it)
            hipLaunchKernelGGL(k_viscous_residual<3>, grid1(p->nc), dim3(CFD_BLOCK), 0, ibh_stream, *f, p->nc, V, P, ldp,
                               ldg, mu_t, R, ldr);
        else
            hipLaunchKernelGGL(k_viscous_residual_shared<3>, visc_grid(p->nc), dim3(VISC_WG), 0, ibh_stream, *f, p->nc, V,
                               P, ldp, ldg, mu_t, R, ldr);
    }
    IBH_LAUNCH_CHECK();
    return 0;
}

int ibh_cfd_flow_bc(const ibh_fluid* f, int nd, int64_t n, const float* P, int64_t ldp, const float* normals, int64_t ldn,
                    float p_inf, float T_inf, const float* u_inf, int normal_flow, const float* image_distances,
                    const float* dudn, float transpiration, const float* transpiration_v, float* out, int64_t ldo) {
    IBH_REQUIRE(f && P && normals && u_inf && out, "ibh_cfd_flow_bc: null argument");
    CHECK_ND(nd, 1);
    IBH_REQUIRE((image_distances == nullptr) == (dudn == nullptr),
                "du!dn and image_distances must be passed together for BC imposition");
    if (n <= 0) return 0;
    const float u0 = u_inf[0], u1 = normal_flow ? 0.0f : u_inf[1], u2 = (!normal_flow && nd == 3) ? u_inf[2] : 0.0f;
    if (nd == 2)
        hipLaunchKernelGGL(k_flow_bc<2>, grid1(n), dim3(CFD_BLOCK), 0, ibh_stream, *f, n, P, ldp, normals, ldn, p_inf, T_inf,
                           u0, u1, u2, normal_flow, image_distances, dudn, transpiration, transpiration_v, out, ldo);
    else
        hipLaunchKernelGGL(k_flow_bc<3>, grid1(n), dim3(CFD_BLOCK), 0, ibh_stream, *f, n, P, ldp, normals, ldn, p_inf, T_inf,
                           u0, u1, u2, normal_flow, image_distances, dudn, transpiration, transpiration_v, out, ldo);
    IBH_LAUNCH_CHECK();
    return 0;
}

int ibh_cfd_jst_sensor3(int64_t n, const float* Pim1, const float* Pi, const float* Pip1, float* out) {
    IBH_REQUIRE(Pim1 && Pi && Pip1 && out, "ibh_cfd_jst_sensor3: null argument");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_jst3, grid1(n), dim3(CFD_BLOCK), 0, ibh_stream, n, Pim1, Pi, Pip1, out);
    IBH_LAUNCH_CHECK();
    return 0;
}

int ibh_cfd_shock_sensor(int nd, int64_t n, const float* const* velocity_gradients, float* out) {
    IBH_REQUIRE(velocity_gradients && out, "ibh_cfd_shock_sensor: null argument");
    CHECK_ND(nd, 1);
    if (n <= 0) return 0;
    ShockGradPtrs G;
    for (int k = 0; k < nd * nd; ++k) {
        IBH_REQUIRE(velocity_gradients[k], "ibh_cfd_shock_sensor: null gradient array");
        G.g[k] = velocity_gradients[k];
    }
    if (nd == 2) hipLaunchKernelGGL(k_shock<2>, grid1(n), dim3(CFD_BLOCK), 0, ibh_stream, n, G, out);
    else hipLaunchKernelGGL(k_shock<3>, grid1(n), dim3(CFD_BLOCK), 0, ibh_stream, n, G, out);
    IBH_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
