// libibhip: building blocks of the direct xGMI halo exchange (see include/ibhip.h).
// Flags are monotonic sequence numbers in fine-grained memory; data visibility: the packing kernel ends
// (system-scope release at kernel end) before the signal kernel of the same stream starts; the receiver's
// unpack kernel starts (system-scope acquire at kernel start) after its wait kernel has seen the flag.
#include <string.h>

#include "ibh_common.h"

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

}  // extern "C"
