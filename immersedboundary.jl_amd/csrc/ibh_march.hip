// libibhip: the pieces of an explicit solver step around the residual sweep, device resident (test/advection.jl:30-89):
//   * ibh_bcset_*: an ordered list of ghost-cell boundary conditions whose closures the library knows (a constant value,
//     copy(u): advection.jl:30-46) applied to a field with the semantics of the sequential impose_bc! calls
//     (ImmersedBoundary.jl:1197-1247: every ghost cell of a boundary is interpolated BEFORE any of them is written; a later
//     boundary sees the earlier ones): boundaries that do not read each other's ghost cells form a level, a level is two
//     launches (interpolate + closure + blend into a side buffer, scatter) whatever the number of boundaries;
//   * ibh_timestep_advection (ibh_ops.hip): dt = scale * 0.5 / max over cells and dimensions of
//     unsigned_green_gauss(at_faces(C_d, d), d) (advection.jl:52-59) left in device memory -- no host read-back in the loop.
// The step itself (sweep + u += dt ud in one launch) is ibh_step_advection in ibh_fused.hip.
// Arithmetic: the operator kernels' (ibh_ops.hip): -ffp-contract=off, the reference's evaluation order.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "ibh_bcset_dev.h"
#include "ibh_common.h"

#define MARCH_BLOCK 256

namespace {

// ghosts g0 .. g1 of the set: interpolate from the field, closure, blend (k_bc_blend) -- into `gval`, NOT into the field:
// every ghost cell of these boundaries is interpolated from the field as it was before any of them is written
__global__ void k_bcset_interp(int32_t g0, int32_t g1, const float* __restrict__ eta, const int32_t* __restrict__ off,
                               const int32_t* __restrict__ donor, const float* __restrict__ w,
                               const int32_t* __restrict__ bidx, const int32_t* __restrict__ mode,
                               const float* __restrict__ value, const float* __restrict__ a, float* __restrict__ gval,
                               const int32_t* __restrict__ ghost, float* a_out) {
    bcset_dev::interp_wg(blockIdx.x, gridDim.x, g0, g1, eta, off, donor, w, bidx, mode, value, a, gval, ghost, a_out);
}
__global__ void k_bcset_scatter(int32_t g0, int32_t g1, const int32_t* __restrict__ ghost, const float* __restrict__ gval,
                                float* __restrict__ a) {
    bcset_dev::scatter_wg(blockIdx.x, gridDim.x, g0, g1, ghost, gval, a);
}

// ---- every level of the set in ONE launch: at most 256 workgroups, all resident at once (a launch starts when the stream's
// previous kernel has ended), so a grid-wide barrier on a counter cannot dead-lock; it is bounded all the same -- a wave that
// gives up sets sync[2] and goes on (ibh_bcset_info reports it).  The last workgroup to leave resets the counters.
struct BcsetSeg {
    int32_t seg[IBH_MAX_BC + 1];
    int32_t direct[IBH_MAX_BC];
};
__device__ __forceinline__ void bcset_grid_barrier(unsigned int* sync, unsigned int target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                   // this workgroup's stores before its arrival
        atomicAdd(&sync[0], 1u);
        unsigned int spins = 0;
        while (__hip_atomic_load(&sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 22)) {
                atomicOr(&sync[2], 1u);
                break;
            }
        }
        __threadfence();                                   // the others' stores after the barrier
    }
    __syncthreads();
}
__global__ __launch_bounds__(MARCH_BLOCK) void k_bcset_all(int nlev, BcsetSeg S, const float* __restrict__ eta,
                                                           const int32_t* __restrict__ off, const int32_t* __restrict__ donor,
                                                           const float* __restrict__ w, const int32_t* __restrict__ bidx,
                                                           const int32_t* __restrict__ mode, const float* __restrict__ value,
                                                           const int32_t* __restrict__ ghost, float* a, float* gval,
                                                           unsigned int* sync) {
    unsigned int nbar = 0;
    for (int lv = 0; lv < nlev; ++lv) {
        const int32_t g0 = S.seg[lv], g1 = S.seg[lv + 1];
        for (int32_t g = g0 + blockIdx.x * blockDim.x + threadIdx.x; g < g1; g += gridDim.x * blockDim.x) {
            const int32_t b = off[g], e = off[g + 1];
            float s = 0.0f;
            for (int32_t k = b; k < e; ++k) {   // (plain loads: `a` changes between the levels of this launch)
                const float t = a[donor[k]] * w[k];
                s = (k == b) ? t : s + t;
            }
            const int32_t kb = bidx[g];
            const float et = eta[g];
            const float bv = mode[kb] ? s : value[kb];
            const float v = et * s + (1.0f - et) * bv;
            if (S.direct[lv]) a[ghost[g]] = v;
            else gval[g] = v;
        }
        if (!S.direct[lv]) {
            bcset_grid_barrier(sync, ++nbar * gridDim.x);
            for (int32_t g = g0 + blockIdx.x * blockDim.x + threadIdx.x; g < g1; g += gridDim.x * blockDim.x)
                a[ghost[g]] = gval[g];
        }
        if (lv + 1 < nlev) bcset_grid_barrier(sync, ++nbar * gridDim.x);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&sync[1], 1u) == gridDim.x - 1) {    // last one out: counters back to zero for the next launch
            __threadfence();
            sync[0] = 0u;
            sync[1] = 0u;
        }
    }
}

// out = u + dt * r with dt in device memory (the update of advection.jl:87 for partitions without the fused step kernel)
__global__ void k_update_dev(int64_t n, const float* __restrict__ dt, const float* __restrict__ u,
                             const float* __restrict__ r, float* __restrict__ out) {
    const float h = *dt;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = u[i] + r[i] * h;
}

}  // namespace

extern "C" {

int ibh_bcset_create(ibh_bcset** out, int nbc, const ibh_bc* const* bcs, const int32_t* modes, const float* values) {
    IBH_REQUIRE(out && nbc >= 1 && nbc <= IBH_MAX_BC && bcs && modes && values, "ibh_bcset_create: bad argument");
    for (int k = 0; k < nbc; ++k) {
        IBH_REQUIRE(bcs[k] && (modes[k] == 0 || modes[k] == 1), "ibh_bcset_create: null boundary or bad mode");
        IBH_REQUIRE(bcs[k]->ng == 0 || (int32_t)bcs[k]->h_off.size() == bcs[k]->ng + 1,
                    "ibh_bcset_create: boundary without host stencils");
    }
    // Levels.  Sequential impose_bc! calls: boundary k is interpolated from the field as boundaries 1..k-1 left it and
    // before any of its own ghost cells is written.  Interpolating a whole LEVEL of boundaries into a side buffer and
    // scattering afterwards is the same thing iff no boundary of the level has a donor cell that is a ghost cell of an
    // earlier boundary of the same level (its own ghost cells and those of later boundaries are read as they were: right)
    // and no ghost cell belongs to two boundaries of the level.  level(k) = 1 + max level of the earlier boundaries it
    // depends on that way.
    std::vector<std::vector<int32_t>> gs(nbc);
    for (int k = 0; k < nbc; ++k) {
        gs[k] = bcs[k]->h_ghost;
        std::sort(gs[k].begin(), gs[k].end());
    }
    int level[IBH_MAX_BC] = {0}, nlev = 0;
    for (int k = 0; k < nbc; ++k) {
        int lv = 0;
        for (int j = 0; j < k; ++j) {
            bool dep = false;
            for (int32_t c : bcs[k]->h_donor)
                if (std::binary_search(gs[j].begin(), gs[j].end(), c)) { dep = true; break; }
            for (int32_t c : gs[k])
                if (!dep && std::binary_search(gs[j].begin(), gs[j].end(), c)) { dep = true; break; }
            if (dep) lv = std::max(lv, level[j] + 1);
        }
        level[k] = lv;
        nlev = std::max(nlev, lv + 1);
    }
    ibh_bcset* s = new ibh_bcset();
    s->nbc = nbc;
    s->nlev = nlev;
    std::vector<int32_t> ghost, off{0}, donor, bidx, mode(modes, modes + nbc);
    std::vector<float> eta, w, value(values, values + nbc);
    for (int lv = 0; lv < nlev; ++lv) {  // ghosts ordered by level, boundaries of a level in their order
        s->seg[lv] = (int32_t)ghost.size();
        for (int k = 0; k < nbc; ++k) {
            if (level[k] != lv) continue;
            const ibh_bc* b = bcs[k];
            for (int32_t g = 0; g < b->ng; ++g) {
                ghost.push_back(b->h_ghost[g]);
                eta.push_back(b->h_eta[g]);
                bidx.push_back(k);
                for (int32_t j = b->h_off[g]; j < b->h_off[g + 1]; ++j) {
                    donor.push_back(b->h_donor[j]);
                    w.push_back(b->h_w[j]);
                }
                off.push_back((int32_t)donor.size());
            }
        }
    }
    s->seg[nlev] = s->ng = (int32_t)ghost.size();
    for (int lv = 0; lv < nlev; ++lv) {  // a level none of whose ghost cells is one of its donors needs no side buffer
        std::vector<int32_t> gl(ghost.begin() + s->seg[lv], ghost.begin() + s->seg[lv + 1]);
        std::sort(gl.begin(), gl.end());
        bool hazard = false;
        for (int32_t j = off[s->seg[lv]]; j < off[s->seg[lv + 1]] && !hazard; ++j)
            hazard = std::binary_search(gl.begin(), gl.end(), donor[j]);
        s->direct[lv] = !hazard;
    }
    int rc;
    if ((rc = ibh_upload(&s->ghost, ghost.data(), ghost.size()))) return rc;
    if ((rc = ibh_upload(&s->eta, eta.data(), eta.size()))) return rc;
    if ((rc = ibh_upload(&s->off, off.data(), off.size()))) return rc;
    if ((rc = ibh_upload(&s->donor, donor.data(), donor.size()))) return rc;
    if ((rc = ibh_upload(&s->w, w.data(), w.size()))) return rc;
    if ((rc = ibh_upload(&s->bidx, bidx.data(), bidx.size()))) return rc;
    if ((rc = ibh_upload(&s->mode, mode.data(), mode.size()))) return rc;
    if ((rc = ibh_upload(&s->value, value.data(), value.size()))) return rc;
    IBH_HIP(hipMalloc((void**)&s->gval, sizeof(float) * std::max<size_t>(ghost.size(), 1)));
    IBH_HIP(hipMalloc((void**)&s->sync, 4 * sizeof(unsigned int)));
    IBH_HIP(hipMemset(s->sync, 0, 4 * sizeof(unsigned int)));
    *out = s;
    return 0;
}

int ibh_bcset_destroy(ibh_bcset* s) {
    if (!s) return 0;
    hipFree(s->ghost); hipFree(s->eta); hipFree(s->off); hipFree(s->donor); hipFree(s->w); hipFree(s->bidx);
    hipFree(s->mode); hipFree(s->value); hipFree(s->gval); hipFree(s->sync);
    delete s;
    return 0;
}

int ibh_bcset_info(const ibh_bcset* s, int32_t* n_ghost, int32_t* n_levels) {
    IBH_REQUIRE(s, "ibh_bcset_info: null set");
    if (n_ghost) *n_ghost = s->ng;
    if (n_levels) {
        int nd = 0;
        for (int lv = 0; lv < s->nlev; ++lv) nd += s->direct[lv];
        *n_levels = s->nlev | (nd << 16);  // low half: levels; high half: how many of them are blended in one launch
    }
    if (n_ghost && s->sync) {   // a barrier of the one-launch form gave up (never seen): report it as a negative ghost count
        unsigned int st[4] = {0, 0, 0, 0};
        IBH_HIP(hipMemcpy(st, s->sync, sizeof(st), hipMemcpyDeviceToHost));
        if (st[2]) *n_ghost = -s->ng;
    }
    return 0;
}

// 0 (default): interpolate / scatter launches per level; 1: all levels in one launch with an in-kernel barrier -- measured
// SLOWER (march step 34.5 against 16.8 us with 182 workgroups: the arrivals at one counter serialise at ~100 ns each across
// the XCDs, a grid barrier costs several launches' worth), kept as an opt-in for the record
static int ibh_bcset_one_launch = getenv("IBH_BCSET_ONE_LAUNCH") ? atoi(getenv("IBH_BCSET_ONE_LAUNCH")) : 0;

int ibh_bcset_apply(const ibh_bcset* s, float* a) {
    IBH_REQUIRE(s && a, "ibh_bcset_apply: null argument");
    if (s->ng == 0) return 0;
    if (ibh_bcset_one_launch) {
        BcsetSeg S;
        int32_t widest = 0;
        for (int lv = 0; lv <= IBH_MAX_BC; ++lv) S.seg[lv] = lv <= s->nlev ? s->seg[lv] : s->seg[s->nlev];
        for (int lv = 0; lv < IBH_MAX_BC; ++lv) S.direct[lv] = lv < s->nlev && s->direct[lv];
        for (int lv = 0; lv < s->nlev; ++lv) widest = std::max(widest, s->seg[lv + 1] - s->seg[lv]);
        const int nwg = std::max(1, std::min(ibh_grid(widest, MARCH_BLOCK), 256));   // all resident at once
        hipLaunchKernelGGL(k_bcset_all, dim3(nwg), dim3(MARCH_BLOCK), 0, ibh_stream, s->nlev, S, s->eta, s->off, s->donor,
                           s->w, s->bidx, s->mode, s->value, s->ghost, a, s->gval, s->sync);
        IBH_LAUNCH_CHECK();
        return 0;
    }
    for (int lv = 0; lv < s->nlev; ++lv) {
        const int32_t g0 = s->seg[lv], g1 = s->seg[lv + 1];
        if (g1 == g0) continue;
        const int nwg = std::min(ibh_grid(g1 - g0, MARCH_BLOCK), 2048);
        if (s->direct[lv]) {
            hipLaunchKernelGGL(k_bcset_interp, dim3(nwg), dim3(MARCH_BLOCK), 0, ibh_stream, g0, g1, s->eta, s->off, s->donor,
                               s->w, s->bidx, s->mode, s->value, a, s->gval, s->ghost, a);
            continue;
        }
        hipLaunchKernelGGL(k_bcset_interp, dim3(nwg), dim3(MARCH_BLOCK), 0, ibh_stream, g0, g1, s->eta, s->off, s->donor,
                           s->w, s->bidx, s->mode, s->value, a, s->gval, (const int32_t*)nullptr, (float*)nullptr);
        hipLaunchKernelGGL(k_bcset_scatter, dim3(nwg), dim3(MARCH_BLOCK), 0, ibh_stream, g0, g1, s->ghost, s->gval, a);
    }
    IBH_LAUNCH_CHECK();
    return 0;
}

int ibh_update_dev(int64_t n, const float* dt_dev, const float* u, const float* r, float* out) {
    IBH_REQUIRE(dt_dev && u && r && out, "ibh_update_dev: null argument");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_update_dev, dim3(std::min(ibh_grid(n, MARCH_BLOCK), 2048)), dim3(MARCH_BLOCK), 0, ibh_stream, n,
                       dt_dev, u, r, out);
    IBH_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
