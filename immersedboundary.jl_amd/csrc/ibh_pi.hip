// libibhip: device kernels of the point-implicit smoother (reference: the orphan file
// /root/reference/src/point_implicit.jl -- Hutchinson estimate of the block diagonal :17-91, per-point pinv
// :124-135, block apply :153-161, two-direction minimal-residual relaxation :250-329).
//
// Everything here is streaming work at <= 0.5 flop/B (a 5x5 block apply reads 30 floats for 50 flops):
// bound by HBM bandwidth; MFMA is not applicable (north_star: "MFMA only if the block solve proves a dense
// contraction" -- it does not: the batched mat-vec has no reuse across points).
// Block layout: D (n, nv, nv) column-major like the reference's stack(): D[p + n*(k + nv*i)] = d f_k / d x_i at
// point p, so every access is coalesced over p.
#include "ibh_common.h"

namespace {

constexpr int PB = 256;

__device__ __forceinline__ uint64_t mix64(uint64_t x) {  // splitmix64 finaliser
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ void k_rademacher(int64_t n, uint64_t seed, float* __restrict__ z) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        z[i] = (mix64(seed * 0x2545F4914F6CDD1Dull + (uint64_t)i) >> 63) ? 1.0f : -1.0f;
}

// out = x + v*h   (:30 / :112)
__global__ void k_perturb(int64_t n, const float* __restrict__ x, const float* __restrict__ v, float h,
                          float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = x[i] + v[i] * h;
}

// out = (fxb - fx) / h   (:32, :112-114)
__global__ void k_fd(int64_t n, const float* __restrict__ fxb, const float* __restrict__ fx, float h,
                     float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (fxb[i] - fx[i]) / h;
}

// s[p, k] += z[p] * (fxb[p, k] - fx[p, k]) / h   (:40 with J of :29-33)
__global__ void k_hutch_accum(int64_t n, int nv, const float* __restrict__ fxb, const float* __restrict__ fx,
                              const float* __restrict__ z, float h, float* __restrict__ s) {
    const int64_t tot = n * nv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (int64_t)gridDim.x * blockDim.x)
        s[i] = s[i] + z[i % n] * ((fxb[i] - fx[i]) / h);
}

__global__ void k_div_scalar(int64_t n, float d, float* __restrict__ s) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        s[i] = s[i] / d;
}

// :124-126  D = 1 / (eps + D)
__global__ void k_invert_diag(int64_t n, float* __restrict__ D) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        D[i] = 1.0f / (1.1920929e-07f + D[i]);
}

// :127-135  per-point Moore-Penrose inverse of an M x M block, in place: one-sided Jacobi SVD in registers,
// singular values <= eps(Float32) * M * sigma_max are dropped (the default tolerance of LinearAlgebra.pinv).
template <int M>
__global__ void k_pinv_blocks(int64_t n, float* __restrict__ D) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    float U[M][M], V[M][M];  // U[r][c]
#pragma unroll
    for (int c = 0; c < M; ++c)
#pragma unroll
        for (int r = 0; r < M; ++r) {
            U[r][c] = D[p + n * (int64_t)(r + M * c)];
            V[r][c] = r == c ? 1.0f : 0.0f;
        }
    for (int sweep = 0; sweep < 12; ++sweep) {
        float off = 0.0f;
#pragma unroll
        for (int a = 0; a < M - 1; ++a)
#pragma unroll
            for (int b = a + 1; b < M; ++b) {
                float al = 0.f, be = 0.f, ga = 0.f;
#pragma unroll
                for (int r = 0; r < M; ++r) {
                    al += U[r][a] * U[r][a];
                    be += U[r][b] * U[r][b];
                    ga += U[r][a] * U[r][b];
                }
                const float lim = 1e-7f * sqrtf(al * be);
                if (fabsf(ga) > lim && fabsf(ga) > 0.0f) {
                    off = fmaxf(off, fabsf(ga) / fmaxf(sqrtf(al * be), 1e-37f));
                    const float zeta = (be - al) / (2.0f * ga);
                    const float t = (zeta >= 0.0f ? 1.0f : -1.0f) / (fabsf(zeta) + sqrtf(1.0f + zeta * zeta));
                    const float c = 1.0f / sqrtf(1.0f + t * t), s = c * t;
#pragma unroll
                    for (int r = 0; r < M; ++r) {
                        const float ua = U[r][a], ub = U[r][b];
                        U[r][a] = c * ua - s * ub;
                        U[r][b] = s * ua + c * ub;
                        const float va = V[r][a], vb = V[r][b];
                        V[r][a] = c * va - s * vb;
                        V[r][b] = s * va + c * vb;
                    }
                }
            }
        if (off < 1e-7f) break;
    }
    float sig[M], smax = 0.0f;
#pragma unroll
    for (int c = 0; c < M; ++c) {
        float q = 0.f;
#pragma unroll
        for (int r = 0; r < M; ++r) q += U[r][c] * U[r][c];
        sig[c] = sqrtf(q);
        smax = fmaxf(smax, sig[c]);
    }
    const float tol = 1.1920929e-07f * (float)M * smax;
    // pinv = V diag(1/sigma) (U/sigma)^T  ->  P[i][k] = sum_c V[i][c] * U[k][c] / sigma_c^2
    float w[M];
#pragma unroll
    for (int c = 0; c < M; ++c) w[c] = sig[c] > tol ? 1.0f / (sig[c] * sig[c]) : 0.0f;
#pragma unroll
    for (int k = 0; k < M; ++k)
#pragma unroll
        for (int i = 0; i < M; ++i) {
            float q = 0.f;
#pragma unroll
            for (int c = 0; c < M; ++c) q += V[i][c] * U[k][c] * w[c];
            D[p + n * (int64_t)(i + M * k)] = q;
        }
}

// :153-161  out[p, k] = sum_i v[p, i] * invD[p, k, i]
template <int M>
__global__ void k_apply_blocks(int64_t n, const float* __restrict__ invD, const float* __restrict__ v,
                               float* __restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    float x[M];
#pragma unroll
    for (int i = 0; i < M; ++i) x[i] = v[p + n * i];
#pragma unroll
    for (int k = 0; k < M; ++k) {
        float q = x[0] * invD[p + n * (int64_t)k];
#pragma unroll
        for (int i = 1; i < M; ++i) q = q + x[i] * invD[p + n * (int64_t)(k + M * i)];
        out[p + n * k] = q;
    }
}

__global__ void k_mul(int64_t n, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = a[i] * b[i];
}

// block reduction helpers: out[0] += sum a*b (double), out[0] = max |a| (float bits, non-negative: integer max)
__global__ void k_dot(int64_t n, const float* __restrict__ a, const float* __restrict__ b, double* __restrict__ out) {
    __shared__ double sh[PB / 64];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        s += (double)a[i] * (double)b[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < PB / 64; ++k) t += sh[k];
        atomicAdd(out, t);
    }
}
__global__ void k_maxabs(int64_t n, const float* __restrict__ a, uint32_t* __restrict__ out) {
    float m = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf(a[i]));
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

// x += alpha s ; r -= alpha As  with  alpha = AvB / (AvAv + eps)  read from device memory (:231-236, :291-294)
__global__ void k_pi_update(int64_t n, const double* __restrict__ dots, float eps, const float* __restrict__ s,
                            const float* __restrict__ As, float* __restrict__ x, float* __restrict__ r) {
    const float alpha = (float)dots[0] / ((float)dots[1] + eps);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        x[i] = x[i] + s[i] * alpha;
        r[i] = r[i] - As[i] * alpha;
    }
}
// s = r / (eps + max|r|)   (:297-299)
__global__ void k_pi_normalize(int64_t n, const float* __restrict__ r, const uint32_t* __restrict__ mx, float eps,
                               float* __restrict__ s) {
    const float d = eps + __uint_as_float(mx[0]);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        s[i] = r[i] / d;
}

inline int grid_for(int64_t n) {
    int g = ibh_grid(n, PB);
    return g > 4096 ? 4096 : g;
}

}  // namespace

extern "C" {

int ibh_pi_rademacher(int64_t n, uint64_t seed, float* z) {
    IBH_REQUIRE(z || n == 0, "ibh_pi_rademacher: null argument");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_rademacher, dim3(grid_for(n)), dim3(PB), 0, ibh_stream, n, seed, z);
    IBH_LAUNCH_CHECK();
    return 0;
}
int ibh_pi_perturb(int64_t n, const float* x, const float* v, float h, float* out) {
    if (n <= 0) return 0;
    IBH_REQUIRE(x && v && out, "ibh_pi_perturb: null argument");
    hipLaunchKernelGGL(k_perturb, dim3(grid_for(n)), dim3(PB), 0, ibh_stream, n, x, v, h, out);
    IBH_LAUNCH_CHECK();
    return 0;
}
int ibh_pi_fd(int64_t n, const float* fxb, const float* fx, float h, float* out) {
    if (n <= 0) return 0;
    IBH_REQUIRE(fxb && fx && out, "ibh_pi_fd: null argument");
    hipLaunchKernelGGL(k_fd, dim3(grid_for(n)), dim3(PB), 0, ibh_stream, n, fxb, fx, h, out);
    IBH_LAUNCH_CHECK();
    return 0;
}
int ibh_pi_hutch_accum(int64_t n, int nv, const float* fxb, const float* fx, const float* z, float h, float* s) {
    if (n <= 0 || nv <= 0) return 0;
    IBH_REQUIRE(fxb && fx && z && s, "ibh_pi_hutch_accum: null argument");
    hipLaunchKernelGGL(k_hutch_accum, dim3(grid_for(n * nv)), dim3(PB), 0, ibh_stream, n, nv, fxb, fx, z, h, s);
    IBH_LAUNCH_CHECK();
    return 0;
}
int ibh_pi_div_scalar(int64_t n, float d, float* s) {
    if (n <= 0) return 0;
    IBH_REQUIRE(s, "ibh_pi_div_scalar: null argument");
    hipLaunchKernelGGL(k_div_scalar, dim3(grid_for(n)), dim3(PB), 0, ibh_stream, n, d, s);
    IBH_LAUNCH_CHECK();
    return 0;
}
int ibh_pi_invert_blocks(int64_t n, int nv, float* D) {
    if (n <= 0) return 0;
    IBH_REQUIRE(D, "ibh_pi_invert_blocks: null argument");
    IBH_REQUIRE(nv >= 1 && nv <= 8, "ibh_pi_invert_blocks: 1 <= nv <= 8");
    const dim3 g((unsigned)((n + 63) / 64)), b(64);
    switch (nv) {
        case 1: hipLaunchKernelGGL(k_invert_diag, dim3(grid_for(n)), dim3(PB), 0, ibh_stream, n, D); break;
        case 2: hipLaunchKernelGGL(k_pinv_blocks<2>, g, b, 0, ibh_stream, n, D); break;
        case 3: hipLaunchKernelGGL(k_pinv_blocks<3>, g, b, 0, ibh_stream, n, D); break;
        case 4: hipLaunchKernelGGL(k_pinv_blocks<4>, g, b, 0, ibh_stream, n, D); break;
        case 5: hipLaunchKernelGGL(k_pinv_blocks<5>, g, b, 0, ibh_stream, n, D); break;
        case 6: hipLaunchKernelGGL(k_pinv_blocks<6>, g, b, 0, ibh_stream, n, D); break;
        case 7: hipLaunchKernelGGL(k_pinv_blocks<7>, g, b, 0, ibh_stream, n, D); break;
        default: hipLaunchKernelGGL(k_pinv_blocks<8>, g, b, 0, ibh_stream, n, D); break;
    }
    IBH_LAUNCH_CHECK();
    return 0;
}
int ibh_pi_apply_blocks(int64_t n, int nv, const float* invD, const float* v, float* out) {
    if (n <= 0) return 0;
    IBH_REQUIRE(invD && v && out, "ibh_pi_apply_blocks: null argument");
    IBH_REQUIRE(nv >= 1 && nv <= 8, "ibh_pi_apply_blocks: 1 <= nv <= 8");
    const dim3 g((unsigned)((n + PB - 1) / PB)), b(PB);
    switch (nv) {
        case 1: hipLaunchKernelGGL(k_mul, dim3(grid_for(n)), dim3(PB), 0, ibh_stream, n, v, invD, out); break;
        case 2: hipLaunchKernelGGL(k_apply_blocks<2>, g, b, 0, ibh_stream, n, invD, v, out); break;
        case 3: hipLaunchKernelGGL(k_apply_blocks<3>, g, b, 0, ibh_stream, n, invD, v, out); break;
        case 4: hipLaunchKernelGGL(k_apply_blocks<4>, g, b, 0, ibh_stream, n, invD, v, out); break;
        case 5: hipLaunchKernelGGL(k_apply_blocks<5>, g, b, 0, ibh_stream, n, invD, v, out); break;
        case 6: hipLaunchKernelGGL(k_apply_blocks<6>, g, b, 0, ibh_stream, n, invD, v, out); break;
        case 7: hipLaunchKernelGGL(k_apply_blocks<7>, g, b, 0, ibh_stream, n, invD, v, out); break;
        default: hipLaunchKernelGGL(k_apply_blocks<8>, g, b, 0, ibh_stream, n, invD, v, out); break;
    }
    IBH_LAUNCH_CHECK();
    return 0;
}
int ibh_dot(int64_t n, const float* a, const float* b, double* out) {
    IBH_REQUIRE(out, "ibh_dot: null argument");
    IBH_HIP(hipMemsetAsync(out, 0, sizeof(double), ibh_stream));
    if (n <= 0) return 0;
    IBH_REQUIRE(a && b, "ibh_dot: null argument");
    hipLaunchKernelGGL(k_dot, dim3(grid_for(n) > 1024 ? 1024 : grid_for(n)), dim3(PB), 0, ibh_stream, n, a, b, out);
    IBH_LAUNCH_CHECK();
    return 0;
}
int ibh_maxabs(int64_t n, const float* a, float* out) {
    IBH_REQUIRE(out, "ibh_maxabs: null argument");
    IBH_HIP(hipMemsetAsync(out, 0, sizeof(float), ibh_stream));
    if (n <= 0) return 0;
    IBH_REQUIRE(a, "ibh_maxabs: null argument");
    hipLaunchKernelGGL(k_maxabs, dim3(grid_for(n) > 1024 ? 1024 : grid_for(n)), dim3(PB), 0, ibh_stream, n, a,
                       reinterpret_cast<uint32_t*>(out));
    IBH_LAUNCH_CHECK();
    return 0;
}
int ibh_pi_update(int64_t n, const double* dots, float eps, const float* s, const float* As, float* x, float* r) {
    if (n <= 0) return 0;
    IBH_REQUIRE(dots && s && As && x && r, "ibh_pi_update: null argument");
    hipLaunchKernelGGL(k_pi_update, dim3(grid_for(n)), dim3(PB), 0, ibh_stream, n, dots, eps, s, As, x, r);
    IBH_LAUNCH_CHECK();
    return 0;
}
int ibh_pi_normalize(int64_t n, const float* r, const float* maxabs, float eps, float* s) {
    if (n <= 0) return 0;
    IBH_REQUIRE(r && maxabs && s, "ibh_pi_normalize: null argument");
    hipLaunchKernelGGL(k_pi_normalize, dim3(grid_for(n)), dim3(PB), 0, ibh_stream, n, r,
                       reinterpret_cast<const uint32_t*>(maxabs), eps, s);
    IBH_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
