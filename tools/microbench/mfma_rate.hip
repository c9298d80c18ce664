// Microbenchmark: sustained issue-to-issue cycles of the two fp32 MFMA shapes with independent accumulators, one wave per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/mfma_rate.hip -o tools/_build/mfma_rate && tools/_build/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// correctness of back-to-back inline-asm MFMAs: the same accumulation with and without s_nop 1, results compared on the host
template <bool NOP, bool DEP>
__global__ __launch_bounds__(256, 1) void k16check(float* res, int iters) {
    f32x4 acc[36];
    for (int p = 0; p < 36; ++p) acc[p] = f32x4{0, 0, 0, 0};
    const float a = 1.0f + (threadIdx.x & 63) * 0.001f, b = 0.5f + (threadIdx.x & 15) * 0.01f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 36; ++p) {
            f32x4& c = acc[DEP ? (p % 6) : p];            // DEP: an accumulator is touched again six MFMAs later
            if (NOP) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    for (int p = 0; p < 36; ++p) for (int r = 0; r < 4; ++r) res[((size_t)blockIdx.x * 256 + threadIdx.x) * 144 + p * 4 + r] = acc[p][r];
}

template <int NACC, bool NOP>
__global__ __launch_bounds__(256, 1) void k16(unsigned long long* out, int iters) {
    f32x4 acc[NACC];
    for (int p = 0; p < NACC; ++p) acc[p] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < NACC; ++p) {
            if (NOP) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[p]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[p]) : "v"(a), "v"(b));
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int p = 0; p < NACC; ++p) s += acc[p][0];
    if (s == 123.456f) out[1] = 1;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) atomicMax(out, t1 - t0);
}

// N packed VALU instructions (v_pk_fma_f32 on private registers) after every 16x16x4 MFMA
template <int N>
__global__ __launch_bounds__(256, 1) void k16v(unsigned long long* out, int iters) {
    f32x4 acc[36];
    for (int p = 0; p < 36; ++p) acc[p] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.0f;
    double v[8]; for (int i = 0; i < 8; ++i) v[i] = i + threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 36; ++p) {
            asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[p]) : "v"(a), "v"(b));
#pragma unroll
            for (int n = 0; n < N; ++n) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(v[n & 7]) : "v"(v[(n + 1) & 7]));
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int p = 0; p < 36; ++p) s += acc[p][0]; for (int i = 0; i < 8; ++i) s += (float)v[i];
    if (s == 123.456f) out[1] = 1;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) atomicMax(out, t1 - t0);
}

template <int NACC>
__global__ __launch_bounds__(256, 1) void k32(unsigned long long* out, int iters) {
    f32x16 acc[NACC];
    for (int p = 0; p < NACC; ++p) for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < NACC; ++p) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[p]) : "v"(a), "v"(b));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int p = 0; p < NACC; ++p) s += acc[p][0];
    if (s == 123.456f) out[1] = 1;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) atomicMax(out, t1 - t0);
}

template <typename F>
double run(F launch, int per_iter, int iters) {
    unsigned long long* d; hipMalloc(&d, 16); hipMemset(d, 0, 16);
    launch(d, iters); hipDeviceSynchronize();
    hipMemset(d, 0, 16);
    launch(d, iters); hipDeviceSynchronize();
    unsigned long long h = 0; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost); hipFree(d);
    return (double)h / ((double)iters * per_iter);
}

template <bool DEP>
void check() {
    const size_t n = (size_t)256 * 256 * 144;
    float *d0, *d1; hipMalloc(&d0, n * 4); hipMalloc(&d1, n * 4);
    hipLaunchKernelGGL((k16check<false, DEP>), dim3(256), dim3(256), 0, 0, d0, 50);
    hipLaunchKernelGGL((k16check<true, DEP>), dim3(256), dim3(256), 0, 0, d1, 50);
    hipDeviceSynchronize();
    float* h0 = new float[n]; float* h1 = new float[n];
    hipMemcpy(h0, d0, n * 4, hipMemcpyDeviceToHost); hipMemcpy(h1, d1, n * 4, hipMemcpyDeviceToHost);
    size_t bad = 0; for (size_t i = 0; i < n; ++i) bad += h0[i] != h1[i];
    printf("back-to-back asm MFMAs, %s accumulators: %zu of %zu results differ between the plain and the s_nop 1 build (sample %g vs %g)\n",
           DEP ? "re-used (distance 6)" : "independent", bad, n, h0[0], h1[0]);
    delete[] h0; delete[] h1; hipFree(d0); hipFree(d1);
}

int main() {
    check<false>();
    check<true>();
    const int iters = 2000;
    // s_memtime counts at a fixed 100 MHz-derived rate on some parts; report raw ticks per MFMA and the ratio between shapes
    printf("16x16x4  36 acc          : %.2f ticks/MFMA\n", run([](auto d, int n) { hipLaunchKernelGGL((k16<36, false>), dim3(256), dim3(256), 0, 0, d, n); }, 36, iters));
    printf("16x16x4  36 acc + s_nop 1: %.2f ticks/MFMA\n", run([](auto d, int n) { hipLaunchKernelGGL((k16<36, true>), dim3(256), dim3(256), 0, 0, d, n); }, 36, iters));
    printf("16x16x4   6 acc          : %.2f ticks/MFMA\n", run([](auto d, int n) { hipLaunchKernelGGL((k16<6, false>), dim3(256), dim3(256), 0, 0, d, n); }, 6, iters));
    printf("32x32x2  16 acc          : %.2f ticks/MFMA (2x the MACs of a 16x16x4)\n", run([](auto d, int n) { hipLaunchKernelGGL((k32<16>), dim3(256), dim3(256), 0, 0, d, n); }, 16, iters));
    printf("16x16x4 + 1 v_pk_fma_f32 : %.2f ticks/MFMA\n", run([](auto d, int n) { hipLaunchKernelGGL((k16v<1>), dim3(256), dim3(256), 0, 0, d, n); }, 36, iters));
    printf("16x16x4 + 2 v_pk_fma_f32 : %.2f ticks/MFMA\n", run([](auto d, int n) { hipLaunchKernelGGL((k16v<2>), dim3(256), dim3(256), 0, 0, d, n); }, 36, iters));
    printf("16x16x4 + 4 v_pk_fma_f32 : %.2f ticks/MFMA\n", run([](auto d, int n) { hipLaunchKernelGGL((k16v<4>), dim3(256), dim3(256), 0, 0, d, n); }, 36, iters));
    printf("16x16x4 + 6 v_pk_fma_f32 : %.2f ticks/MFMA\n", run([](auto d, int n) { hipLaunchKernelGGL((k16v<6>), dim3(256), dim3(256), 0, 0, d, n); }, 36, iters));
    return 0;
}
