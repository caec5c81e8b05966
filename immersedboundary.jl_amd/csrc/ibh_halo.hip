// libibhip: building blocks of the direct xGMI halo exchange (see include/ibhip.h).
// Flags are monotonic sequence numbers in fine-grained memory; data visibility: the packing kernel ends
// (system-scope release at kernel end) before the signal kernel of the same stream starts; the receiver's
// unpack kernel starts (system-scope acquire at kernel start) after its wait kernel has seen the flag.
#include <string.h>

#include <algorithm>

#include "ibh_common.h"
#include "ibh_halo_dev.h"

namespace {

__global__ void k_flag_signal(uint32_t* __restrict__ counter, uint32_t* const* __restrict__ slots, int n) {
    __shared__ uint32_t seq;
    if (threadIdx.x == 0) {
        seq = *counter + 1u;
        *counter = seq;
    }
    __syncthreads();
    __threadfence_system();
    if ((int)threadIdx.x < n)
        __hip_atomic_store(slots[threadIdx.x], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void k_flag_wait(uint32_t* __restrict__ counter, const uint32_t* const* __restrict__ slots, int n,
                            uint32_t max_spins, uint32_t* __restrict__ status) {
    __shared__ uint32_t exp;
    if (threadIdx.x == 0) {
        exp = *counter + 1u;
        *counter = exp;
    }
    __syncthreads();
    if ((int)threadIdx.x < n) {
        const uint32_t* s = slots[threadIdx.x];
        uint32_t spins = 0;
        // relaxed system-scope polls (bypass the caches), bounded: every wave reaches the exit
        while (__hip_atomic_load(s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < exp) {
            if (++spins >= max_spins) {
                atomicOr(status, 1u);
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __threadfence_system();
}

// ---- the whole exchange in two launches (a step of a strong-scaled run is a handful of ~3 us kernels, so every
// launch counts): push = pack for ALL peers + signal by the last workgroup to finish; pull = bounded wait on ALL
// flags + unpack.  Per-peer segments of the concatenated index lists: seg[q] .. seg[q+1].
struct PushArgs {
    float* dst[IBH_MAX_PEERS];        // peer receive buffer (parity and my offset applied)
    uint32_t* flag[IBH_MAX_PEERS];    // my slot in the peer's flag array
    int32_t seg[IBH_MAX_PEERS + 1];
    int32_t n;
};
struct PullArgs {
    const uint32_t* flag[IBH_MAX_PEERS];  // local flag slot of each source peer
    int32_t seg[IBH_MAX_PEERS + 1];
    int32_t n;
};


// state: [0] signal sequence, [1] wait sequence, [2] status, [3] push workgroups done, [4] pull workgroups done
__global__ void k_halo_push(const float* __restrict__ f, int nv, int64_t ld, const int32_t* __restrict__ send_all,
                            PushArgs P, uint32_t* __restrict__ state) {
    const int32_t total = P.seg[P.n];
    for (int32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const int q = seg_of(P.seg, P.n, t);
        const int32_t i = t - P.seg[q], nq = P.seg[q + 1] - P.seg[q];
        const int32_t c = send_all[t];
        for (int v = 0; v < nv; ++v) P.dst[q][i + (int64_t)v * nq] = f[c + (int64_t)v * ld];
    }
    __threadfence_system();  // this workgroup's stores are visible system-wide before it counts as done
    __shared__ uint32_t last, seq;
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(&state[3], 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!last) return;
    if (threadIdx.x == 0) {
        seq = state[0] + 1u;
        state[0] = seq;
        state[3] = 0u;
    }
    __syncthreads();
    __threadfence_system();
    if ((int)threadIdx.x < P.n) __hip_atomic_store(P.flag[threadIdx.x], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void k_halo_pull(float* __restrict__ f, int nv, int64_t ld, const int32_t* __restrict__ recv_all,
                            const float* __restrict__ src, PullArgs R, uint32_t* __restrict__ state,
                            uint32_t max_spins) {
    __shared__ uint32_t exp, last;
    if (threadIdx.x == 0) exp = state[1] + 1u;  // advanced only by the last workgroup to finish (below)
    __syncthreads();
    if ((int)threadIdx.x < R.n) {
        const uint32_t* s = R.flag[threadIdx.x];
        uint32_t spins = 0;
        // relaxed system-scope polls (bypass the caches), bounded: every wave reaches the exit
        while (__hip_atomic_load(s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < exp) {
            if (++spins >= max_spins) {
                atomicOr(&state[2], 1u);
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
    __threadfence_system();  // acquire: the peers' data stores precede their flag stores
    const int32_t total = R.seg[R.n];
    for (int32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const int q = seg_of(R.seg, R.n, t);
        const int32_t i = t - R.seg[q], nq = R.seg[q + 1] - R.seg[q];
        const float* sq = src + (int64_t)R.seg[q] * nv;
        const int32_t c = recv_all[t];
        for (int v = 0; v < nv; ++v)
            f[c + (int64_t)v * ld] = __builtin_nontemporal_load(sq + i + (int64_t)v * nq);
    }
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(&state[4], 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (last && threadIdx.x == 0) {
        state[1] = exp;
        state[4] = 0u;
    }
}

// push and pull in ONE launch (body: ibh_halo_dev.h, shared with the fused step kernel of ibh_fused.hip)
__global__ void k_halo_exchange(float* __restrict__ f, int nv, int64_t ld, const int32_t* __restrict__ send_all,
                                const int32_t* __restrict__ recv_all, const float* __restrict__ src0,
                                const float* __restrict__ src1, XchgArgs A, uint32_t* __restrict__ state,
                                uint32_t max_spins) {
    halo_exchange_wg(f, nv, ld, send_all, recv_all, src0, src1, A, state, max_spins, (int)blockIdx.x, (int)gridDim.x);
}

}  // namespace

extern "C" {

int ibh_ipc_alloc(void** dptr, size_t bytes, int fine_grained) {
    IBH_REQUIRE(dptr, "ibh_ipc_alloc: null argument");
    if (bytes == 0) bytes = 4;
    if (fine_grained)
        IBH_HIP(hipExtMallocWithFlags(dptr, bytes, hipDeviceMallocFinegrained));
    else
        IBH_HIP(hipMalloc(dptr, bytes));
    IBH_HIP(hipMemset(*dptr, 0, bytes));
    IBH_HIP(hipDeviceSynchronize());
    return 0;
}

int ibh_ipc_free(void* dptr) {
    if (dptr) IBH_HIP(hipFree(dptr));
    return 0;
}

int ibh_ipc_export(void* dptr, void* handle64) {
    IBH_REQUIRE(dptr && handle64, "ibh_ipc_export: null argument");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "HIP IPC handle is 64 bytes");
    IBH_HIP(hipIpcGetMemHandle((hipIpcMemHandle_t*)handle64, dptr));
    return 0;
}

int ibh_ipc_import(const void* handle64, void** dptr) {
    IBH_REQUIRE(dptr && handle64, "ibh_ipc_import: null argument");
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, sizeof(h));
    IBH_HIP(hipIpcOpenMemHandle(dptr, h, hipIpcMemLazyEnablePeerAccess));
    return 0;
}

int ibh_ipc_close(void* dptr) {
    if (dptr) IBH_HIP(hipIpcCloseMemHandle(dptr));
    return 0;
}

int ibh_flag_signal(uint32_t* counter, uint32_t* const* slots, int n) {
    IBH_REQUIRE(counter && slots && n >= 0 && n <= 64, "ibh_flag_signal: bad argument");
    hipLaunchKernelGGL(k_flag_signal, dim3(1), dim3(64), 0, ibh_stream, counter, slots, n);
    IBH_LAUNCH_CHECK();
    return 0;
}

int ibh_flag_wait(uint32_t* counter, const uint32_t* const* slots, int n, uint32_t max_spins, uint32_t* status) {
    IBH_REQUIRE(counter && slots && status && n >= 0 && n <= 64, "ibh_flag_wait: bad argument");
    hipLaunchKernelGGL(k_flag_wait, dim3(1), dim3(64), 0, ibh_stream, counter, slots, n, max_spins, status);
    IBH_LAUNCH_CHECK();
    return 0;
}

int ibh_halo_push(const float* f, int nv, int64_t ld, const int32_t* send_all, int n_peers, const int32_t* seg,
                  float* const* dst, uint32_t* const* flags, uint32_t* state) {
    IBH_REQUIRE(n_peers >= 0 && n_peers <= IBH_MAX_PEERS, "ibh_halo_push: at most 16 peers");
    if (n_peers == 0) return 0;
    IBH_REQUIRE(f && send_all && seg && dst && flags && state && nv >= 1, "ibh_halo_push: bad argument");
    PushArgs P;
    memset(&P, 0, sizeof(P));
    P.n = n_peers;
    for (int q = 0; q < n_peers; ++q) {
        P.dst[q] = dst[q];
        P.flag[q] = flags[q];
        P.seg[q] = seg[q];
        IBH_REQUIRE(seg[q + 1] >= seg[q], "ibh_halo_push: segments must ascend");
    }
    P.seg[n_peers] = seg[n_peers];
    int g = (seg[n_peers] + 255) / 256;
    g = g < 1 ? 1 : g > 256 ? 256 : g;
    hipLaunchKernelGGL(k_halo_push, dim3(g), dim3(256), 0, ibh_stream, f, nv, ld, send_all, P, state);
    IBH_LAUNCH_CHECK();
    return 0;
}

int ibh_halo_pull(float* f, int nv, int64_t ld, const int32_t* recv_all, const float* src, int n_peers,
                  const int32_t* seg, const uint32_t* const* flags, uint32_t* state, uint32_t max_spins) {
    IBH_REQUIRE(n_peers >= 0 && n_peers <= IBH_MAX_PEERS, "ibh_halo_pull: at most 16 peers");
    if (n_peers == 0) return 0;
    IBH_REQUIRE(f && recv_all && src && seg && flags && state && nv >= 1, "ibh_halo_pull: bad argument");
    PullArgs R;
    memset(&R, 0, sizeof(R));
    R.n = n_peers;
    for (int q = 0; q < n_peers; ++q) {
        R.flag[q] = flags[q];
        R.seg[q] = seg[q];
        IBH_REQUIRE(seg[q + 1] >= seg[q], "ibh_halo_pull: segments must ascend");
    }
    R.seg[n_peers] = seg[n_peers];
    int g = (seg[n_peers] + 255) / 256;
    g = g < 1 ? 1 : g > 64 ? 64 : g;  // every workgroup polls the flags: keep them few
    hipLaunchKernelGGL(k_halo_pull, dim3(g), dim3(256), 0, ibh_stream, f, nv, ld, recv_all, src, R, state, max_spins);
    IBH_LAUNCH_CHECK();
    return 0;
}

int ibh_halo_exchange(float* f, int nv, int64_t ld, const int32_t* send_all, int n_send_peers, const int32_t* send_seg,
                      float* const* dst0, float* const* dst1, uint32_t* const* send_flags, const int32_t* recv_all,
                      const float* src0, const float* src1, int n_recv_peers, const int32_t* recv_seg,
                      const uint32_t* const* recv_flags, uint32_t* state, uint32_t max_spins) {
    IBH_REQUIRE(n_send_peers >= 0 && n_send_peers <= IBH_MAX_PEERS && n_recv_peers >= 0 && n_recv_peers <= IBH_MAX_PEERS,
                "ibh_halo_exchange: at most 16 peers");
    if (n_send_peers == 0 && n_recv_peers == 0) return 0;
    IBH_REQUIRE(f && state && nv >= 1, "ibh_halo_exchange: bad argument");
    XchgArgs A;
    memset(&A, 0, sizeof(A));
    A.ns = n_send_peers;
    A.nr = n_recv_peers;
    if (n_send_peers) {
        IBH_REQUIRE(send_all && send_seg && dst0 && dst1 && send_flags, "ibh_halo_exchange: null send argument");
        for (int q = 0; q < n_send_peers; ++q) {
            A.dst[0][q] = dst0[q];
            A.dst[1][q] = dst1[q];
            A.sflag[q] = send_flags[q];
            A.sseg[q] = send_seg[q];
            IBH_REQUIRE(send_seg[q + 1] >= send_seg[q], "ibh_halo_exchange: segments must ascend");
        }
        A.sseg[n_send_peers] = send_seg[n_send_peers];
    }
    if (n_recv_peers) {
        IBH_REQUIRE(recv_all && src0 && src1 && recv_seg && recv_flags, "ibh_halo_exchange: null receive argument");
        for (int q = 0; q < n_recv_peers; ++q) {
            A.rflag[q] = recv_flags[q];
            A.rseg[q] = recv_seg[q];
            IBH_REQUIRE(recv_seg[q + 1] >= recv_seg[q], "ibh_halo_exchange: segments must ascend");
        }
        A.rseg[n_recv_peers] = recv_seg[n_recv_peers];
    }
    const int32_t big = std::max(A.ns ? A.sseg[A.ns] : 0, A.nr ? A.rseg[A.nr] : 0);
    int g = (big + 255) / 256;
    g = g < 1 ? 1 : g > 64 ? 64 : g;  // every workgroup polls the flags: few, and all resident
    hipLaunchKernelGGL(k_halo_exchange, dim3(g), dim3(256), 0, ibh_stream, f, nv, ld, send_all, recv_all, src0, src1, A,
                       state, max_spins);
    IBH_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
