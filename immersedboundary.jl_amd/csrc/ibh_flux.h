// Per-face arithmetic of the fused residual sweeps, written once and used by both the
// face-list kernels and the block fast path so the two agree bit for bit.
// Every expression follows the reference's broadcast, operation by operation
// (no FMA contraction: the library is built with -ffp-contract=off).
#pragma once
#include <hip/hip_runtime.h>

namespace ibhf {

__device__ __forceinline__ float face_avg(float uo, float un, float ho, float hn) {
    return (uo * hn + un * ho) / (hn + ho);  // at_faces, ImmersedBoundary.jl:907-909
}

__device__ __forceinline__ float sgn(float x) { return (float)((x > 0.0f) - (x < 0.0f)); }

__device__ __forceinline__ float minmod(float a, float b) {  // ImmersedBoundary.jl:1099
    return fminf(fabsf(a), fabsf(b)) * (sgn(a) + sgn(b)) / 2.0f;
}

// MUSCL(part,u,du,dim; D, high_order=true) at one face (ImmersedBoundary.jl:1119-1156).
// dO/dN = owner/neighbor_distance = h/2; Df = max(D_o, D_n, 1e-7).
__device__ __forceinline__ void muscl_ho(float uo, float un, float duo, float dun, float Df, float dO, float dN,
                                         float& uL, float& uR) {
    float guf = (un - uo) / (dO + dN);
    float gu = (2.0f * duo - guf) * dO;
    float Du = (2.0f * dun - guf) * dN;
    float s = minmod(Du, gu);
    float l = uo + s, r = un - s;
    float uf = (uo * dN + un * dO) / (dO + dN);
    uf = uf + (duo * dO - dun * dN) / 8.0f;
    uL = l * Df + (1.0f - Df) * uf;
    uR = r * Df + (1.0f - Df) * uf;
}

// Upwind advective flux of test/advection.jl:78-82 at one face.
__device__ __forceinline__ float adv_flux(float uo, float un, float duo, float dun, float Do, float Dn, float Co,
                                          float Cn, float ho, float hn) {
    float uL, uR;
    float Df = fmaxf(fmaxf(Do, Dn), 1e-7f);
    muscl_ho(uo, un, duo, dun, Df, ho / 2.0f, hn / 2.0f, uL, uR);
    float Cf = face_avg(Co, Cn, ho, hn);
    return (uL + uR) * Cf / 2.0f + fabsf(Cf) * (uL - uR) / 2.0f;
}

// primitive2state + flux along `dim` for one side (cfd.jl:462-481); P = [p T u v (w)]
template <int ND>
__device__ __forceinline__ void side_state(const float* P, int dim0, float R, float gamma, float* Q, float* F,
                                           float& un, float& a) {
    constexpr int NV = ND + 2;
    float p = P[0];
    float T = fmaxf(P[1], 10.0f);
    float k = P[2] * P[2];
#pragma unroll
    for (int j = 1; j < ND; ++j) k = k + P[2 + j] * P[2 + j];
    k = k / 2.0f;
    float rho = p / (R * T);
    float E = rho * (R / (gamma - 1.0f) * T + k);
    Q[0] = rho;
    Q[1] = E;
#pragma unroll
    for (int j = 0; j < ND; ++j) Q[2 + j] = rho * P[2 + j];
#pragma unroll
    for (int v = 0; v < NV; ++v) F[v] = Q[v];
    F[1] = F[1] + p;
    un = P[2 + dim0];
    a = sqrtf(gamma * R * fmaxf(P[1], 10.0f));
#pragma unroll
    for (int v = 0; v < NV; ++v) F[v] = F[v] * un;
    F[2 + dim0] = F[2 + dim0] + p;
}

// HLL flux of cfd.jl:459-508 (the wave speeds and the result are Float64 in the reference
// because of the `0.0` literals at :504-505; kept).
template <int ND>
__device__ __forceinline__ void hll_flux(const float* PL, const float* PR, int dim0, float R, float gamma, double* F) {
    constexpr int NV = ND + 2;
    float QL[NV], FL[NV], QR[NV], FR[NV], uL, aL, uR, aR;
    side_state<ND>(PL, dim0, R, gamma, QL, FL, uL, aL);
    side_state<ND>(PR, dim0, R, gamma, QR, FR, uR, aR);
    double SR = fmin((double)(uR - aR), 0.0);
    double SL = fmax((double)(uL + aL), 0.0);
#pragma unroll
    for (int v = 0; v < NV; ++v)
        F[v] = (SL * (double)FL[v] - SR * (double)FR[v] + SR * SL * (double)(QR[v] - QL[v])) / (SL - SR);
}

// Full Euler face flux: MUSCL(high_order) on every primitive with the pressure sensor, then HLL.
template <int ND>
__device__ __forceinline__ void euler_face_flux(const float* Po, const float* Pn, const float* dPo, const float* dPn,
                                                float Do, float Dn, float ho, float hn, int dim0, float R, float gamma,
                                                double* F) {
    constexpr int NV = ND + 2;
    float PL[NV], PR[NV];
    float Df = fmaxf(fmaxf(Do, Dn), 1e-7f);
    float dO = ho / 2.0f, dN = hn / 2.0f;
#pragma unroll
    for (int v = 0; v < NV; ++v) muscl_ho(Po[v], Pn[v], dPo[v], dPn[v], Df, dO, dN, PL[v], PR[v]);
    hll_flux<ND>(PL, PR, dim0, R, gamma, F);
}

}  // namespace ibhf
