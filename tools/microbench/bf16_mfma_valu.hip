// Microbenchmark (r4): what fits beside bf16 MFMAs on gfx950 -- the arithmetic behind DESIGN section 8's "bf16-split F(4x4)" proposal.
//   hipcc -O3 -w --offload-arch=gfx950 tools/microbench/bf16_mfma_valu.hip -o tools/_build/bf16_mfma_valu && tools/_build/bf16_mfma_valu
// One wave per SIMD (256 threads per workgroup, one workgroup per CU).  A run = 64 MFMAs on 8 rotating accumulators, with N fillers of one kind behind
// every MFMA; printed: shader ticks per MFMA (s_memtime), for v_mfma_f32_16x16x32_bf16 (K = 32, the 2.5 PF/s form), v_mfma_f32_16x16x16_bf16 (K = 16) and,
// as the reference point, v_mfma_f32_16x16x4_f32 (what conv3x3_wino4 issues today: 32 ticks, and every VALU instruction next to it costs ~5 more).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

enum { M_BF32, M_BF16K, M_F32 };
enum { F_PKFMA, F_PKADD, F_CVT, F_AND, F_DSR, F_KINDS };
const char* MN[] = {"v_mfma_f32_16x16x32_bf16", "v_mfma_f32_16x16x16_bf16", "v_mfma_f32_16x16x4_f32 "};
const char* FN[] = {"v_pk_fma_f32", "v_pk_add_f32", "v_cvt_pk_bf16_f32", "v_and_b32", "ds_read_b128"};

template <int MK, int FK, int N>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* ticks) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    bf16x8 a8 = {1, 2, 3, 4, 5, 6, 7, 8}, b8 = {8, 7, 6, 5, 4, 3, 2, 1};
    bf16x4 a4 = {1, 2, 3, 4}, b4 = {4, 3, 2, 1};
    float fa = 1.0f + threadIdx.x, fb = 0.5f;
    double v[8]; float s[8]; f32x4 d[4]; unsigned w[8];
    for (int i = 0; i < 8; ++i) { v[i] = 1.0 + i; s[i] = 0.5f + i; w[i] = i; }
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
            if (MK == M_BF32) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(a8), "v"(b8));
            if (MK == M_BF16K) asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(a4), "v"(b4));
            if (MK == M_F32) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(fa), "v"(fb));
#pragma unroll
            for (int n = 0; n < N; ++n) {
                const int i = (m * N + n) & 7;
                if (FK == F_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
                if (FK == F_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
                if (FK == F_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[i]) : "v"(s[i]), "v"(s[(i + 1) & 7]));
                if (FK == F_AND) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(w[i]) : "v"(w[(i + 1) & 7]));
                if (FK == F_DSR) asm volatile("ds_read_b128 %0, %1" : "=v"(d[i & 3]) : "v"(lp));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
        t1 = __builtin_amdgcn_s_memtime();
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += acc[i].x + (float)v[i] + s[i] + (float)w[i];
    for (int i = 0; i < 4; ++i) r += d[i].x;
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MK, int FK, int N>
double run(float* out, unsigned long long* ticks, int cus) {
    hipLaunchKernelGGL((k<MK, FK, N>), dim3(cus), dim3(256), 0, 0, out, ticks);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k<MK, FK, N>), dim3(cus), dim3(256), 0, 0, out, ticks);
    hipDeviceSynchronize();
    unsigned long long h[1024];
    hipMemcpy(h, ticks, sizeof(unsigned long long) * cus, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < cus; ++i) s += (double)h[i];
    return s / cus / 64.0;
}

template <int MK, int FK>
void row(float* out, unsigned long long* ticks, int cus) {
    printf("  %-24s + N x %-18s  N=0 %6.1f  1 %6.1f  2 %6.1f  3 %6.1f  4 %6.1f  6 %6.1f  8 %6.1f   ticks per MFMA\n", MN[MK], FN[FK],
           run<MK, FK, 0>(out, ticks, cus), run<MK, FK, 1>(out, ticks, cus), run<MK, FK, 2>(out, ticks, cus), run<MK, FK, 3>(out, ticks, cus),
           run<MK, FK, 4>(out, ticks, cus), run<MK, FK, 6>(out, ticks, cus), run<MK, FK, 8>(out, ticks, cus));
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount < 1024 ? p.multiProcessorCount : 1024;
    float* out; unsigned long long* ticks;
    hipMalloc(&out, sizeof(float) * 1024 * 256);
    hipMalloc(&ticks, sizeof(unsigned long long) * 1024);
    printf("%s, %d CUs, one wave per SIMD, 64 MFMAs per run on 8 accumulators, N fillers behind each MFMA\n", p.gcnArchName, cus);
    row<M_BF32, F_PKFMA>(out, ticks, cus);  row<M_BF32, F_PKADD>(out, ticks, cus);  row<M_BF32, F_CVT>(out, ticks, cus);  row<M_BF32, F_AND>(out, ticks, cus);  row<M_BF32, F_DSR>(out, ticks, cus);
    row<M_BF16K, F_PKFMA>(out, ticks, cus); row<M_BF16K, F_CVT>(out, ticks, cus);  row<M_BF16K, F_DSR>(out, ticks, cus);
    row<M_F32, F_PKFMA>(out, ticks, cus);   row<M_F32, F_CVT>(out, ticks, cus);    row<M_F32, F_DSR>(out, ticks, cus);
    return 0;
}
