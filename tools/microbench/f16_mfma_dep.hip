// Microbenchmark (r4): issue-to-issue ticks of v_mfma_f32_16x16x16_f16 when an accumulator is reused every D instructions (D = 1: a dependent chain),
// one wave per SIMD.  conv3x3_wino4h.hip reuses an accumulator after 2 MFMAs (both tile groups of a position, three products each).
//   hipcc -O3 -w --offload-arch=gfx950 tools/microbench/f16_mfma_dep.hip -o tools/_build/f16_mfma_dep && tools/_build/f16_mfma_dep
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <int D, bool AGPR>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* ticks) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    s16x4 a = {1, 2, 3, 4}, b = {4, 3, 2, 1};
    asm volatile("" : "+v"(a), "+v"(b));
    unsigned long long t0 = 0, t1 = 0;
    for (int rep = 0; rep < 3; ++rep) {
        t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int m = 0; m < 96; ++m) {
            if (AGPR) asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+a"(acc[m % D]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(acc[m % D]) : "v"(a), "v"(b));
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        t1 = __builtin_amdgcn_s_memtime();
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += acc[i].x;
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
template <int D, bool AGPR>
double run(float* out, unsigned long long* ticks, int cus) {
    for (int i = 0; i < 2; ++i) { hipLaunchKernelGGL((k<D, AGPR>), dim3(cus), dim3(256), 0, 0, out, ticks); hipDeviceSynchronize(); }
    unsigned long long h[1024];
    hipMemcpy(h, ticks, sizeof(unsigned long long) * cus, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < cus; ++i) s += (double)h[i];
    return s / cus / 96.0;
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount < 1024 ? p.multiProcessorCount : 1024;
    float* out; unsigned long long* ticks;
    hipMalloc(&out, sizeof(float) * 1024 * 256); hipMalloc(&ticks, sizeof(unsigned long long) * 1024);
    printf("v_mfma_f32_16x16x16_f16, ticks per MFMA by accumulator reuse distance D (AGPR accumulators): D=1 %.1f  2 %.1f  3 %.1f  4 %.1f  6 %.1f  8 %.1f\n",
           run<1, true>(out, ticks, cus), run<2, true>(out, ticks, cus), run<3, true>(out, ticks, cus), run<4, true>(out, ticks, cus), run<6, true>(out, ticks, cus), run<8, true>(out, ticks, cus));
    printf("                                                              (VGPR accumulators): D=1 %.1f  2 %.1f  3 %.1f  4 %.1f  6 %.1f  8 %.1f\n",
           run<1, false>(out, ticks, cus), run<2, false>(out, ticks, cus), run<3, false>(out, ticks, cus), run<4, false>(out, ticks, cus), run<6, false>(out, ticks, cus), run<8, false>(out, ticks, cus));
    return 0;
}
