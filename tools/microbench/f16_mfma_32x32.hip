// Microbenchmark (r4): issue interval of the 32 x 32 f16 matrix instructions of gfx950 -- v_mfma_f32_32x32x16_f16 (K = 16, the double-rate form the
// direct f16-split convolution issues, conv3x3_f16x3.hip), v_mfma_f32_32x32x8_f16 (K = 8: pointwise_big_kernel<.., HF>) and, as the reference point,
// v_mfma_f32_32x32x2_f32 -- one wave per SIMD, 64 MFMAs on 8 rotating accumulators (AGPRs), with N ds_read_b128 / v_and_b32 behind each.
//   hipcc -O3 -w --offload-arch=gfx950 tools/microbench/f16_mfma_32x32.hip -o tools/_build/f16_mfma_32x32 && tools/_build/f16_mfma_32x32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
enum { M_K16, M_K8, M_F32 };
enum { F_AND, F_DSR };
const char* MN[] = {"v_mfma_f32_32x32x16_f16", "v_mfma_f32_32x32x8_f16 ", "v_mfma_f32_32x32x2_f32 "};
const char* FN[] = {"v_and_b32", "ds_read_b128"};

template <int MK, int FK, int N>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* ticks) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    f32x4 a8 = {1, 2, 3, 4}, b8 = {4, 3, 2, 1};
    f32x2 a4 = {1, 2}, b4 = {2, 1};
    float fa = 1.0f + threadIdx.x, fb = 0.5f;
    f32x4 d[4]; unsigned w[8];
    for (int i = 0; i < 8; ++i) w[i] = i;
    for (int i = 0; i < 4; ++i) d[i] = f32x4{1, 2, 3, 4};
    lds[threadIdx.x] = 1.0f;
    __syncthreads();
    const unsigned lp = (threadIdx.x & 255) * 16;
    asm volatile("" : "+v"(a8), "+v"(b8), "+v"(a4), "+v"(b4), "+v"(fa), "+v"(fb));
    unsigned long long t0 = 0, t1 = 0;
    for (int rep = 0; rep < 3; ++rep) {
        t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int m = 0; m < 64; ++m) {
            if (MK == M_K16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[m & 7]) : "v"(a8), "v"(b8));
            if (MK == M_K8) asm volatile("v_mfma_f32_32x32x8_f16 %0, %1, %2, %0" : "+a"(acc[m & 7]) : "v"(a4), "v"(b4));
            if (MK == M_F32) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[m & 7]) : "v"(fa), "v"(fb));
#pragma unroll
            for (int n = 0; n < N; ++n) {
                const int i = (m * N + n) & 7;
                if (FK == F_AND) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(w[i]) : "v"(w[(i + 1) & 7]));
                if (FK == F_DSR) asm volatile("ds_read_b128 %0, %1" : "=v"(d[i & 3]) : "v"(lp));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
        t1 = __builtin_amdgcn_s_memtime();
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += acc[i][0] + (float)w[i];
    for (int i = 0; i < 4; ++i) r += d[i].x;
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MK, int FK, int N>
double run(float* out, unsigned long long* ticks, int cus) {
    for (int i = 0; i < 2; ++i) { hipLaunchKernelGGL((k<MK, FK, N>), dim3(cus), dim3(256), 0, 0, out, ticks); hipDeviceSynchronize(); }
    unsigned long long h[1024];
    hipMemcpy(h, ticks, sizeof(unsigned long long) * cus, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < cus; ++i) s += (double)h[i];
    return s / cus / 64.0;
}

template <int MK, int FK>
void row(float* out, unsigned long long* ticks, int cus) {
    printf("  %-24s + N x %-14s  N=0 %6.1f  1 %6.1f  2 %6.1f  4 %6.1f  6 %6.1f  8 %6.1f  12 %6.1f   ticks per MFMA\n", MN[MK], FN[FK],
           run<MK, FK, 0>(out, ticks, cus), run<MK, FK, 1>(out, ticks, cus), run<MK, FK, 2>(out, ticks, cus), run<MK, FK, 4>(out, ticks, cus),
           run<MK, FK, 6>(out, ticks, cus), run<MK, FK, 8>(out, ticks, cus), run<MK, FK, 12>(out, ticks, cus));
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount < 1024 ? p.multiProcessorCount : 1024;
    float* out; unsigned long long* ticks;
    hipMalloc(&out, sizeof(float) * 1024 * 256);
    hipMalloc(&ticks, sizeof(unsigned long long) * 1024);
    printf("%s, %d CUs, one wave per SIMD, 64 MFMAs per run on 8 accumulators of 16 registers, N fillers behind each MFMA\n", p.gcnArchName, cus);
    row<M_K16, F_AND>(out, ticks, cus);  row<M_K16, F_DSR>(out, ticks, cus);
    row<M_K8, F_AND>(out, ticks, cus);   row<M_K8, F_DSR>(out, ticks, cus);
    row<M_F32, F_AND>(out, ticks, cus);  row<M_F32, F_DSR>(out, ticks, cus);
    return 0;
}
