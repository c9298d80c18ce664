// Microbenchmark (r6): sustained wall-clock rate and in-kernel clock of the two bf16 MFMA shapes on RANDOM operands (MI355X_MICROARCH.md, DVFS give-back item 7:
// the chip may hold a higher clock on one shape), next to the fp32 matrix instruction -- what decides whether the split-product kernels should use
// v_mfma_f32_32x32x16_bf16 or v_mfma_f32_16x16x32_bf16.  One or two waves per SIMD, every CU busy, ~2 s of back-to-back launches per row.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/bf16_shape_clock.hip -o tools/_build/bf16_shape_clock && tools/_build/bf16_shape_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>   // 0: 32x32x16 bf16 (4 accumulators of 16), 1: 16x16x32 bf16 (16 accumulators of 4), 2: 32x32x2 f32
__global__ __launch_bounds__(256) void k(const u32x4* __restrict__ src, float* __restrict__ out, unsigned long long* stamps, int iters) {
    u32x4 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = src[(blockIdx.x * 256 + threadIdx.x) * 8 + i]; b[i] = src[(blockIdx.x * 256 + threadIdx.x) * 8 + 4 + i]; }
    f32x16 acc32[4];
    f32x4 acc16[16];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc32[i][r] = 0.f;
    for (int i = 0; i < 16; ++i) acc16[i] = f32x4{0, 0, 0, 0};
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (SHAPE == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j)            // 8 x 32x32x16 = the FLOPs of 32 x 16x16x32
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc32[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[(i + j) & 3]), __builtin_bit_cast(bf16x8, b[(i ^ j) & 3]), acc32[i], 0, 0, 0);
        } else if (SHAPE == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    acc16[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[(i + j) & 3]), __builtin_bit_cast(bf16x8, b[(i ^ j) & 3]), acc16[i], 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc32[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a[(i + j) & 3].x), __builtin_bit_cast(float, b[(i ^ j) & 3].y), acc32[i], 0, 0, 0);
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc32[i][r];
    for (int i = 0; i < 16; ++i) s += acc16[i].x + acc16[i].y + acc16[i].z + acc16[i].w;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
    const int wgs_per_cu[2] = {1, 2};
    std::vector<unsigned> h(256 * 512 * 2 * 32);
    srand(1);
    for (auto& v : h) {   // random bf16 pairs of moderate magnitude
        const unsigned short lo = (unsigned short)(0x3F00 + (rand() & 0xFF) + ((rand() & 1) << 15)), hi = (unsigned short)(0x3F00 + (rand() & 0xFF) + ((rand() & 1) << 15));
        v = lo | ((unsigned)hi << 16);
    }
    u32x4* src; float* out; unsigned long long* st;
    hipMalloc(&src, h.size() * 4); hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&st, 1024 * 16);
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const char* names[3] = {"v_mfma_f32_32x32x16_bf16", "v_mfma_f32_16x16x32_bf16", "v_mfma_f32_32x32x2_f32  "};
    for (int rep = 0; rep < 2; ++rep)
    for (int w = 0; w < 2; ++w)
        for (int shape = 0; shape < 3; ++shape) {
            const int grid = 256 * wgs_per_cu[w], iters = shape == 2 ? 4000 : 20000;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&] {
                if (shape == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, src, out, st, iters);
                else if (shape == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, src, out, st, iters);
                else hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, src, out, st, iters);
            };
            for (int i = 0; i < 20; ++i) launch();        // soak
            hipDeviceSynchronize();
            hipEventRecord(e0);
            const int n = 40;
            for (int i = 0; i < n; ++i) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> hs(2 * grid);
            hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
            double clk = 0; for (int i = 0; i < grid; ++i) clk += (double)hs[2 * i] / hs[2 * i + 1] * 100.0; clk /= grid;
            const double flop = shape == 2 ? 4.0 * 8 * 32 * 32 * 2 * 2 : 32.0 * 16 * 16 * 32 * 2;      // per wave and iteration
            const double tf = flop * iters * grid * 4 * n / (ms * 1e-3) / 1e12;
            printf("%s  %d wave(s) per SIMD: %8.1f TFLOP/s  in-kernel clock %5.0f MHz  (%.2f ms per launch)\n", names[shape], wgs_per_cu[w], tf, clk, ms / n);
        }
    return 0;
}
