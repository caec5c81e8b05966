// libibhip: device bodies of the BC-set launches (ibh_march.hip) -- shared with the launches of ibh_ops.hip that carry the
// time-step evaluation of the next step beside them (ibh_bcset_apply_with_dt).
#pragma once
#include "ibh_common.h"

namespace bcset_dev {

// interpolated value of ghost g of the set: sum over its stencil in the order of k_accumulate (ibh_ops.hip)
__device__ __forceinline__ float bc_interp1(const int32_t* __restrict__ off, const int32_t* __restrict__ donor,
                                            const float* __restrict__ w, const float* __restrict__ a, int32_t g) {
    const int32_t b = off[g], e = off[g + 1];
    float s = 0.0f;
    for (int32_t k = b; k < e; ++k) {
        const float t = a[donor[k]] * w[k];
        s = (k == b) ? t : s + t;
    }
    return s;
}

// ghosts g0 .. g1 of the set by workgroup `wg` of `nwg`: interpolate from the field, closure, blend (k_bc_blend) -- into
// `gval`, NOT into the field: every ghost cell of these boundaries is interpolated from the field as it was before any of
// them is written (ghost != null: a level without a ghost cell among its donors, blended straight into the field)
__device__ __forceinline__ void interp_wg(int wg, int nwg, int32_t g0, int32_t g1, const float* __restrict__ eta,
                                          const int32_t* __restrict__ off, const int32_t* __restrict__ donor,
                                          const float* __restrict__ w, const int32_t* __restrict__ bidx,
                                          const int32_t* __restrict__ mode, const float* __restrict__ value,
                                          const float* __restrict__ a, float* __restrict__ gval,
                                          const int32_t* __restrict__ ghost, float* a_out) {
    for (int32_t g = g0 + wg * blockDim.x + threadIdx.x; g < g1; g += nwg * blockDim.x) {
        const float i = bc_interp1(off, donor, w, a, g);
        const int32_t k = bidx[g];
        const float e = eta[g];
        const float b = mode[k] ? i : value[k];
        const float v = e * i + (1.0f - e) * b;
        if (ghost) a_out[ghost[g]] = v;
        else gval[g] = v;
    }
}
__device__ __forceinline__ void scatter_wg(int wg, int nwg, int32_t g0, int32_t g1, const int32_t* __restrict__ ghost,
                                           const float* __restrict__ gval, float* __restrict__ a) {
    for (int32_t g = g0 + wg * blockDim.x + threadIdx.x; g < g1; g += nwg * blockDim.x) a[ghost[g]] = gval[g];
}

}  // namespace bcset_dev
