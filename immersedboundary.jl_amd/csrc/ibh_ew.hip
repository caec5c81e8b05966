// libibhip: elementwise kernels behind `Base.Broadcast` on device arrays (julia/IBHip.jl) -- what a user closure
// writes between the operators, e.g. `@. (uL + uR) * Cf / 2 + abs(Cf) * (uL - uR) / 2` and `ud .-= ...`
// (/root/reference/test/advection.jl:67-83), `max.(a, b)` and `maximum(...)` (:52-59).  One launch per broadcast node;
// HBM-bound (8-12 B per element).  Fields are column-major (n, nv); an operand is a field of the same shape, a column
// vector (n,) broadcast over the columns, or a scalar.
#include "ibh_common.h"

namespace {

template <int OP>
__device__ __forceinline__ float ew2(float a, float b) {
    if (OP == IBH_EW_ADD) return a + b;
    if (OP == IBH_EW_SUB) return a - b;
    if (OP == IBH_EW_MUL) return a * b;
    if (OP == IBH_EW_DIV) return a / b;
    if (OP == IBH_EW_MAX) return fmaxf(a, b);
    return fminf(a, b);
}

// out[i + j n] = a(i, j) OP b(i, j); an operand with nv* == 1 is a column vector, a null pointer a scalar
template <int OP>
__global__ __launch_bounds__(256) void k_ew_binary(int64_t n, int nv, const float* __restrict__ a, int nva, float sa,
                                                   const float* __restrict__ b, int nvb, float sb, float* out) {
    const int64_t total = n * nv;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = t % n;
        const float x = a ? (nva == 1 ? a[i] : a[t]) : sa;
        const float y = b ? (nvb == 1 ? b[i] : b[t]) : sb;
        out[t] = ew2<OP>(x, y);
    }
}

template <int OP>
__global__ __launch_bounds__(256) void k_ew_unary(int64_t total, const float* __restrict__ a, float* out) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const float x = a[t];
        out[t] = OP == IBH_EW_ABS ? fabsf(x) : OP == IBH_EW_NEG ? -x : OP == IBH_EW_SQRT ? sqrtf(x) : x;
    }
}

__global__ __launch_bounds__(256) void k_ew_fill(int64_t total, float v, float* out) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x)
        out[t] = v;
}

// A postfix program evaluated per element (ibh_ew_eval): the top of the stack in a register, the rest in LDS
// (conflict-free: one column per thread); the program is wave-uniform, so the interpreter loop does not diverge.
struct EwProg {
    int32_t nprog;
    int32_t prog[48];
    const float* arr[8];
    int32_t arr_nv[8];
    float scal[8];
};
// BCAST: some operand is a column vector broadcast over the columns of the result (row index = t mod n); without one every
// operand is a flat array of the result's shape and the 64-bit modulo -- half of the interpreter's instructions -- is not made.
template <bool BCAST>
__global__ __launch_bounds__(256) void k_ew_eval(int64_t n, int nv, EwProg P, float* out) {
    __shared__ float stk[7][256];
    const int64_t total = n * nv;
    const int tid = threadIdx.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + tid; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = BCAST ? t % n : t;
        float tos = 0.0f;
        int sp = 0;  // entries on the stack, the last one in `tos`
        for (int pc = 0; pc < P.nprog; ++pc) {
            const int ins = P.prog[pc], op = ins & 255, k = ins >> 8;
            if (op >= IBH_EW_PUSH_ARRAY) {
                if (sp > 0) stk[sp - 1][tid] = tos;
                tos = op == IBH_EW_PUSH_ARRAY ? (BCAST && P.arr_nv[k] == 1 ? P.arr[k][i] : P.arr[k][t]) : P.scal[k];
                ++sp;
            } else if (op >= IBH_EW_ABS) {
                tos = op == IBH_EW_ABS ? fabsf(tos) : op == IBH_EW_NEG ? -tos : op == IBH_EW_SQRT ? sqrtf(tos) : tos;
            } else {
                const float a = stk[sp - 2][tid];
                --sp;
                switch (op) {
                    case IBH_EW_ADD: tos = a + tos; break;
                    case IBH_EW_SUB: tos = a - tos; break;
                    case IBH_EW_MUL: tos = a * tos; break;
                    case IBH_EW_DIV: tos = a / tos; break;
                    case IBH_EW_MAX: tos = fmaxf(a, tos); break;
                    default: tos = fminf(a, tos); break;
                }
            }
        }
        out[t] = tos;
    }
}
// The same program on FOUR consecutive elements per thread (flat operands, 16-byte aligned): the interpreter's decode and
// branches -- ~10 instructions per operation, more than the operation -- are paid once per four elements, the loads and the
// store are 16 bytes per lane.  Same operations on every element: bit-identical to k_ew_eval.
__global__ __launch_bounds__(256) void k_ew_eval4(int64_t total4, EwProg P, float* out) {
    __shared__ float4 stk[7][256];
    const int tid = threadIdx.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + tid; t < total4; t += (int64_t)gridDim.x * blockDim.x) {
        float4 tos = make_float4(0.f, 0.f, 0.f, 0.f);
        int sp = 0;
        for (int pc = 0; pc < P.nprog; ++pc) {
            const int ins = P.prog[pc], op = ins & 255, k = ins >> 8;
            if (op >= IBH_EW_PUSH_ARRAY) {
                if (sp > 0) stk[sp - 1][tid] = tos;
                if (op == IBH_EW_PUSH_ARRAY) tos = ((const float4*)P.arr[k])[t];
                else tos = make_float4(P.scal[k], P.scal[k], P.scal[k], P.scal[k]);
                ++sp;
            } else if (op >= IBH_EW_ABS) {
                if (op == IBH_EW_ABS) tos = make_float4(fabsf(tos.x), fabsf(tos.y), fabsf(tos.z), fabsf(tos.w));
                else if (op == IBH_EW_NEG) tos = make_float4(-tos.x, -tos.y, -tos.z, -tos.w);
                else if (op == IBH_EW_SQRT) tos = make_float4(sqrtf(tos.x), sqrtf(tos.y), sqrtf(tos.z), sqrtf(tos.w));
            } else {
                const float4 a = stk[sp - 2][tid];
                --sp;
                switch (op) {
                    case IBH_EW_ADD: tos = make_float4(a.x + tos.x, a.y + tos.y, a.z + tos.z, a.w + tos.w); break;
                    case IBH_EW_SUB: tos = make_float4(a.x - tos.x, a.y - tos.y, a.z - tos.z, a.w - tos.w); break;
                    case IBH_EW_MUL: tos = make_float4(a.x * tos.x, a.y * tos.y, a.z * tos.z, a.w * tos.w); break;
                    case IBH_EW_DIV: tos = make_float4(a.x / tos.x, a.y / tos.y, a.z / tos.z, a.w / tos.w); break;
                    case IBH_EW_MAX:
                        tos = make_float4(fmaxf(a.x, tos.x), fmaxf(a.y, tos.y), fmaxf(a.z, tos.z), fmaxf(a.w, tos.w));
                        break;
                    default:
                        tos = make_float4(fminf(a.x, tos.x), fminf(a.y, tos.y), fminf(a.z, tos.z), fminf(a.w, tos.w));
                        break;
                }
            }
        }
        ((float4*)out)[t] = tos;
    }
}

template <int OP>
__device__ __forceinline__ float red2(float a, float b) {
    return OP == IBH_EW_SUM ? a + b : OP == IBH_EW_MAX ? fmaxf(a, b) : fminf(a, b);
}
template <int OP>
__device__ __forceinline__ float red_identity() {
    return OP == IBH_EW_SUM ? 0.0f : OP == IBH_EW_MAX ? -INFINITY : INFINITY;
}
template <int OP>
__device__ __forceinline__ float block_reduce(float v, float* sm) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = red2<OP>(v, __shfl_down(v, o, 64));
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sm[w] = v;
    __syncthreads();
    v = threadIdx.x < (blockDim.x >> 6) ? sm[threadIdx.x] : red_identity<OP>();
    if (w == 0) {
#pragma unroll
        for (int o = 2; o > 0; o >>= 1) v = red2<OP>(v, __shfl_down(v, o, 64));
    }
    return v;
}
// every block reduces its stride of the array; with ONE block the result is written at once, with more (`two_stage`) only
// the partials are -- a second one-block launch over them finishes (many blocks meeting at one counter serialise: ~20 ns
// per arrival across the XCDs, 20 us for 1 024 blocks; a launch costs a tenth of that)
template <int OP>
__global__ __launch_bounds__(256) void k_ew_reduce(int64_t total, const float* __restrict__ a, float* part,
                                                   unsigned int* counter, float* out, int two_stage) {
    __shared__ float sm[4];
    __shared__ bool last;
    float v = red_identity<OP>();
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x)
        v = red2<OP>(v, a[t]);
    v = block_reduce<OP>(v, sm);
    if (two_stage) {
        if (threadIdx.x == 0) part[blockIdx.x] = v;
        return;
    }
    if (threadIdx.x == 0) {
        part[blockIdx.x] = v;
        __threadfence();
        last = atomicAdd(counter, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    v = red_identity<OP>();
    for (unsigned t = threadIdx.x; t < gridDim.x; t += blockDim.x) v = red2<OP>(v, ((volatile float*)part)[t]);
    __syncthreads();
    v = block_reduce<OP>(v, sm);
    if (threadIdx.x == 0) {
        *out = v;
        *counter = 0;  // ready for the next launch on this stream
    }
}

// scratch of the reductions: partials of up to 1024 blocks + the arrival counter (one per host thread)
struct RedScratch {
    float* part = nullptr;
    unsigned int* counter = nullptr;
};
thread_local RedScratch red_scratch;

int ensure_scratch() {
    if (red_scratch.part) return 0;
    IBH_HIP(hipMalloc((void**)&red_scratch.part, 1032 * sizeof(float)));
    IBH_HIP(hipMalloc((void**)&red_scratch.counter, sizeof(unsigned int)));
    IBH_HIP(hipMemset(red_scratch.counter, 0, sizeof(unsigned int)));
    return 0;
}

}  // namespace

extern "C" {

int ibh_ew_binary(int op, int64_t n, int nv, const float* a, int nva, float sa, const float* b, int nvb, float sb,
                  float* out) {
    IBH_REQUIRE(out && n >= 0 && nv >= 1, "ibh_ew_binary: bad argument");
    IBH_REQUIRE((!a || nva == nv || nva == 1) && (!b || nvb == nv || nvb == 1),
                "ibh_ew_binary: an operand is a field of the result's shape, a column vector (nv = 1) or a scalar (NULL)");
    if (n * nv == 0) return 0;
    const dim3 grid(ibh_grid(n * nv, 256 * 4)), blk(256);
#define EW2(OP) hipLaunchKernelGGL(k_ew_binary<OP>, grid, blk, 0, ibh_stream, n, nv, a, nva, sa, b, nvb, sb, out)
    switch (op) {
        case IBH_EW_ADD: EW2(IBH_EW_ADD); break;
        case IBH_EW_SUB: EW2(IBH_EW_SUB); break;
        case IBH_EW_MUL: EW2(IBH_EW_MUL); break;
        case IBH_EW_DIV: EW2(IBH_EW_DIV); break;
        case IBH_EW_MAX: EW2(IBH_EW_MAX); break;
        case IBH_EW_MIN: EW2(IBH_EW_MIN); break;
        default: return ibh_fail(-1, "ibh_ew_binary: unknown operation", __FILE__, __LINE__);
    }
#undef EW2
    IBH_LAUNCH_CHECK();
    return 0;
}

// ibh_set_tuning("ew_scalar", 1): the one-element-per-thread interpreter everywhere (A/B, tests)
int ibh_ew_scalar_only = 0;
int ibh_ew_eval(int64_t n, int nv, int nprog, const int32_t* prog, int narr, const float* const* arrays,
                const int32_t* arr_nv, int nscal, const float* scalars, float* out) {
    IBH_REQUIRE(out && prog && n >= 0 && nv >= 1, "ibh_ew_eval: bad argument");
    IBH_REQUIRE(nprog >= 1 && nprog <= 48 && narr >= 0 && narr <= 8 && nscal >= 0 && nscal <= 8,
                "ibh_ew_eval: at most 48 instructions, 8 arrays and 8 scalars");
    EwProg P;
    P.nprog = nprog;
    int sp = 0;
    for (int pc = 0; pc < nprog; ++pc) {
        const int op = prog[pc] & 255, k = prog[pc] >> 8;
        if (op == IBH_EW_PUSH_ARRAY) {
            IBH_REQUIRE(k >= 0 && k < narr && arrays && arrays[k] && arr_nv && (arr_nv[k] == nv || arr_nv[k] == 1),
                        "ibh_ew_eval: array operand out of range, null, or neither a field of the result's shape nor a "
                        "column vector");
            ++sp;
        } else if (op == IBH_EW_PUSH_SCALAR) {
            IBH_REQUIRE(k >= 0 && k < nscal && scalars, "ibh_ew_eval: scalar operand out of range");
            ++sp;
        } else if (op == IBH_EW_ABS || op == IBH_EW_NEG || op == IBH_EW_SQRT || op == IBH_EW_COPY) {
            IBH_REQUIRE(sp >= 1, "ibh_ew_eval: unary operation on an empty stack");
        } else {
            IBH_REQUIRE(op >= IBH_EW_ADD && op <= IBH_EW_MIN, "ibh_ew_eval: unknown operation");
            IBH_REQUIRE(sp >= 2, "ibh_ew_eval: binary operation needs two operands");
            --sp;
        }
        IBH_REQUIRE(sp <= 8, "ibh_ew_eval: stack deeper than 8");
        P.prog[pc] = prog[pc];
    }
    IBH_REQUIRE(sp == 1, "ibh_ew_eval: the program must leave exactly one value");
    for (int k = 0; k < 8; ++k) {
        P.arr[k] = k < narr ? arrays[k] : nullptr;
        P.arr_nv[k] = k < narr ? arr_nv[k] : 1;
        P.scal[k] = k < nscal ? scalars[k] : 0.0f;
    }
    if (n * nv == 0) return 0;
    const int64_t total = n * nv;
    bool bcast = false, aligned = ((uintptr_t)out & 15) == 0;
    for (int k = 0; k < narr; ++k) {
        bcast = bcast || (arr_nv[k] == 1 && nv > 1);
        aligned = aligned && ((uintptr_t)arrays[k] & 15) == 0;
    }
    if (bcast) {
        hipLaunchKernelGGL(k_ew_eval<true>, dim3(ibh_grid(total, 256 * 4)), dim3(256), 0, ibh_stream, n, nv, P, out);
    } else if (aligned && total >= 4 && !ibh_ew_scalar_only) {
        // flat operands: four elements per thread, and the last total mod 4 elements one by one
        const int64_t total4 = total / 4, tail = total - 4 * total4;
        hipLaunchKernelGGL(k_ew_eval4, dim3(ibh_grid(total4, 256)), dim3(256), 0, ibh_stream, total4, P, out);
        if (tail) {
            EwProg T = P;
            for (int k = 0; k < narr; ++k) T.arr[k] = P.arr[k] + 4 * total4;
            hipLaunchKernelGGL(k_ew_eval<false>, dim3(1), dim3(256), 0, ibh_stream, tail, 1, T, out + 4 * total4);
        }
    } else {
        hipLaunchKernelGGL(k_ew_eval<false>, dim3(ibh_grid(total, 256 * 4)), dim3(256), 0, ibh_stream, total, 1, P, out);
    }
    IBH_LAUNCH_CHECK();
    return 0;
}

int ibh_ew_unary(int op, int64_t total, const float* a, float* out) {
    IBH_REQUIRE(a && out && total >= 0, "ibh_ew_unary: bad argument");
    if (total == 0) return 0;
    const dim3 grid(ibh_grid(total, 256 * 4)), blk(256);
    switch (op) {
        case IBH_EW_ABS: hipLaunchKernelGGL(k_ew_unary<IBH_EW_ABS>, grid, blk, 0, ibh_stream, total, a, out); break;
        case IBH_EW_NEG: hipLaunchKernelGGL(k_ew_unary<IBH_EW_NEG>, grid, blk, 0, ibh_stream, total, a, out); break;
        case IBH_EW_SQRT: hipLaunchKernelGGL(k_ew_unary<IBH_EW_SQRT>, grid, blk, 0, ibh_stream, total, a, out); break;
        case IBH_EW_COPY: hipLaunchKernelGGL(k_ew_unary<IBH_EW_COPY>, grid, blk, 0, ibh_stream, total, a, out); break;
        default: return ibh_fail(-1, "ibh_ew_unary: unknown operation", __FILE__, __LINE__);
    }
    IBH_LAUNCH_CHECK();
    return 0;
}

int ibh_ew_fill(int64_t total, float value, float* out) {
    IBH_REQUIRE(out && total >= 0, "ibh_ew_fill: bad argument");
    if (total == 0) return 0;
    hipLaunchKernelGGL(k_ew_fill, dim3(ibh_grid(total, 256 * 4)), dim3(256), 0, ibh_stream, total, value, out);
    IBH_LAUNCH_CHECK();
    return 0;
}

int ibh_ew_reduce(int op, int64_t total, const float* a, float* out_device) {
    IBH_REQUIRE(a && out_device && total >= 1, "ibh_ew_reduce: bad argument (an empty reduction has no value)");
    int rc = ensure_scratch();
    if (rc) return rc;
    const dim3 grid(std::min(1024, ibh_grid(total, 256 * 8))), blk(256);
    float* part = red_scratch.part;
    unsigned int* cnt = red_scratch.counter;
    const int two = grid.x > 8;
    const int64_t nparts = (int64_t)grid.x;
    // (the second stage reads the first `nparts` partials and writes its own one partial behind them)
#define EW_REDUCE(OP)                                                                                                  \
    hipLaunchKernelGGL(k_ew_reduce<OP>, grid, blk, 0, ibh_stream, total, a, part, cnt, out_device, two);               \
    if (two) hipLaunchKernelGGL(k_ew_reduce<OP>, dim3(1), blk, 0, ibh_stream, nparts, (const float*)part, part + 1024, cnt, out_device, 0)
    switch (op) {
        case IBH_EW_SUM: EW_REDUCE(IBH_EW_SUM); break;
        case IBH_EW_MAX: EW_REDUCE(IBH_EW_MAX); break;
        case IBH_EW_MIN: EW_REDUCE(IBH_EW_MIN); break;
        default: return ibh_fail(-1, "ibh_ew_reduce: unknown operation", __FILE__, __LINE__);
    }
#undef EW_REDUCE
    IBH_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
