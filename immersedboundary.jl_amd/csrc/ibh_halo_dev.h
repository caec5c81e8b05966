// Device side of the one-launch xGMI halo exchange (ibh_halo.hip: k_halo_exchange; ibh_fused.hip: the fused
// exchange + sweep step).  `wg` of `nwg` workgroups of 256 threads run the body; all of them must be resident.
#pragma once
#include "ibh_common.h"

#define IBH_MAX_PEERS 16

__device__ __forceinline__ int seg_of(const int32_t* seg, int n, int32_t t) {
    int q = 0;
    while (q + 1 < n && t >= seg[q + 1]) ++q;
    return q;
}

// push and pull in ONE launch: every workgroup packs its share for the peers, the last one to finish signals;
// then every workgroup waits for the peers' flags and unpacks its share.  All workgroups are resident (<= 64) and
// the push part waits for nothing, so ranks running this kernel at the same time cannot block each other.
struct XchgArgs {
    float* dst[2][IBH_MAX_PEERS];     // peer receive buffers for the two parities (my offset applied)
    uint32_t* sflag[IBH_MAX_PEERS];   // my slot in the peers' flag arrays
    const uint32_t* rflag[IBH_MAX_PEERS];  // local flag slot of each source peer
    int32_t sseg[IBH_MAX_PEERS + 1], rseg[IBH_MAX_PEERS + 1];
    int32_t ns, nr;
};
__device__ __forceinline__ void halo_exchange_wg(float* __restrict__ f, int nv, int64_t ld,
                                                 const int32_t* __restrict__ send_all,
                                                 const int32_t* __restrict__ recv_all, const float* __restrict__ src0,
                                                 const float* __restrict__ src1, const XchgArgs& A,
                                                 uint32_t* __restrict__ state, uint32_t max_spins, int wg, int nwg) {
    __shared__ uint32_t last, seq, exp, par;
    if (threadIdx.x == 0) {
        // the parity of the double buffer comes from the device-side sequence number, so graph replays and eager
        // launches can be mixed freely; state[0] / state[1] advance only after every workgroup has read them
        par = state[0] & 1u;
        exp = state[1] + 1u;
    }
    __syncthreads();
    const uint32_t pp = par;
    const int32_t stotal = A.ns ? A.sseg[A.ns] : 0;
    for (int32_t t = wg * blockDim.x + threadIdx.x; t < stotal; t += nwg * blockDim.x) {
        const int q = seg_of(A.sseg, A.ns, t);
        const int32_t i = t - A.sseg[q], nq = A.sseg[q + 1] - A.sseg[q];
        const int32_t c = send_all[t];
        float* d = pp ? A.dst[1][q] : A.dst[0][q];
        for (int v = 0; v < nv; ++v) d[i + (int64_t)v * nq] = f[c + (int64_t)v * ld];
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(&state[3], 1u) == (uint32_t)nwg - 1 ? 1u : 0u;
    __syncthreads();
    if (last) {
        if (threadIdx.x == 0) {
            seq = state[0] + 1u;
            state[0] = seq;
            state[3] = 0u;
        }
        __syncthreads();
        __threadfence_system();
        if ((int)threadIdx.x < A.ns)
            __hip_atomic_store(A.sflag[threadIdx.x], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if ((int)threadIdx.x < A.nr) {
        const uint32_t* s = A.rflag[threadIdx.x];
        uint32_t spins = 0;
        while (__hip_atomic_load(s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < exp) {
            if (++spins >= max_spins) {  // bounded: every wave reaches the exit
                atomicOr(&state[2], 1u);
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
    __threadfence_system();
    const float* src = pp ? src1 : src0;
    const int32_t rtotal = A.nr ? A.rseg[A.nr] : 0;
    for (int32_t t = wg * blockDim.x + threadIdx.x; t < rtotal; t += nwg * blockDim.x) {
        const int q = seg_of(A.rseg, A.nr, t);
        const int32_t i = t - A.rseg[q], nq = A.rseg[q + 1] - A.rseg[q];
        const float* sq = src + (int64_t)A.rseg[q] * nv;
        const int32_t c = recv_all[t];
        for (int v = 0; v < nv; ++v) f[c + (int64_t)v * ld] = __builtin_nontemporal_load(sq + i + (int64_t)v * nq);
    }
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(&state[4], 1u) == (uint32_t)nwg - 1) {
        state[1] = exp;
        state[4] = 0u;
    }
}

