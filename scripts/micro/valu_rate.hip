// Issue rate of FP32 vector instructions on gfx950, one or several waves per SIMD (what the "VALU busy" counters do not
// tell: whether v_pk_*_f32 retires in the 4 cycles of a plain wave64 instruction).
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* cyc) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float c = 1.0001f, d = 0.5f;
    const v2f pc = {c, c}, pd = {d, d};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {  // 8 independent v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
        } else if (KIND == 1) {  // 8 independent v_pk_fma_f32
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc), "v"(pd));
        } else if (KIND == 2) {  // 8 v_pk_mul_f32
            asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                         "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc));
        } else if (KIND == 3) {  // 8 v_rcp_f32
            asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                         "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (KIND == 4) {  // 4 v_fma + 4 v_pk_fma interleaved
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %4, %4, %10, %11\n v_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %5, %5, %10, %11\n"
                         "v_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %6, %6, %10, %11\n v_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %7, %7, %10, %11"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(c), "v"(d), "v"(pc), "v"(pd));
        } else if (KIND == 5) {  // 8 v_med3_f32
            asm volatile("v_med3_f32 %0, %0, %8, %9\n v_med3_f32 %1, %1, %8, %9\n v_med3_f32 %2, %2, %8, %9\n v_med3_f32 %3, %3, %8, %9\n"
                         "v_med3_f32 %4, %4, %8, %9\n v_med3_f32 %5, %5, %8, %9\n v_med3_f32 %6, %6, %8, %9\n v_med3_f32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
        } else if (KIND == 6) {  // 8 v_sqrt_f32
            asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
                         "v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (KIND == 7) {  // 8 v_pk_add_f32
            asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                         "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pd));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p5.y + p6.x + p7.y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name, int wg_per_cu) {
    int ncu = 256;
    const int nwg = ncu * wg_per_cu, iters = 20000;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * nwg * 256);
    hipMalloc(&cyc, 8 * nwg);
    hipLaunchKernelGGL(k<KIND>, dim3(nwg), dim3(256), 0, 0, out, 100, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(nwg), dim3(256), 0, 0, out, iters, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(nwg);
    hipMemcpy(h.data(), cyc, 8 * nwg, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += v;
    avg /= nwg;
    // s_memtime ticks (shader clock) per instruction of ONE wave, and wall time per instruction per SIMD
    printf("%-22s waves/SIMD %d: %.2f ticks per instruction per wave, wall %.3f ms -> %.2f ns per instruction issued per SIMD\n",
           name, wg_per_cu, avg / (8.0 * iters), ms, ms * 1e6 / (8.0 * iters * wg_per_cu));
    hipFree(out);
    hipFree(cyc);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", w);
        run<1>("v_pk_fma_f32", w);
        run<2>("v_pk_mul_f32", w);
        run<7>("v_pk_add_f32", w);
        run<4>("fma/pk_fma mix", w);
        run<5>("v_med3_f32", w);
        run<3>("v_rcp_f32", w);
        run<6>("v_sqrt_f32", w);
    }
    return 0;
}
