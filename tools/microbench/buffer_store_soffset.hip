// Repro (r3): raw_buffer_store_b128 with the per-pixel offset in the SCALAR offset field vs in the vector offset.
//   hipcc -O3 -w --offload-arch=gfx950 tools/microbench/buffer_store_soffset.hip -o tools/_build/buffer_store_soffset && tools/_build/buffer_store_soffset
// conv3x3_wino4's epilogue (r2) "stored wrong values for one lane quad" when the pixel offset (i * W + jj) * ldo * 4 rode in soffset.  This
// program stores 16 pixels x 4 cout quads x 16 tiles per wave the same way, with the offset in (A) voffset, (B) soffset computed by scalar
// instructions from kernel arguments, (C) soffset read back from a VGPR lane (v_readlane, what an SGPR spill reload does) -- and with the
// resource covering (1) the whole tensor, (2) exactly up to the last byte the LANE offsets reach (num_records smaller than soffset + voffset
// of some lanes would be legal to drop).  Every element is then compared with the expected value on the host.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, int W, int ldo, int H, unsigned num_records) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tile = lane & 15, kq = lane >> 4;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)num_records, 0x00020000);
    int Wt = __builtin_amdgcn_readfirstlane(W), ldot = __builtin_amdgcn_readfirstlane(ldo);
    asm volatile("" : "+s"(Wt), "+s"(ldot));
    const int py0 = blockIdx.x * 16 + 4 * (tile >> 2), px0 = 4 * (tile & 3), co = wave * 16 + 4 * kq;
    const unsigned lane_off = (unsigned)((py0 * Wt + px0) * ldot + co) * 4u;
    unsigned spill = 0;                                         // MODE 2: the 16 scalar offsets parked in lanes of a VGPR
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const int i = p >> 2, jj = p & 3;
        const unsigned soff = (unsigned)((i * Wt + jj) * ldot * 4);
        const f32x4 v = {(float)(py0 + i), (float)(px0 + jj), (float)co, (float)p};
        if (MODE == 0) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, lane_off + soff, 0, 0);
        if (MODE == 1) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, lane_off, (int)soff, 0);
        if (MODE == 2) asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(spill) : "s"(soff), "n"(p));
    }
    if (MODE == 2) {
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const int i = p >> 2, jj = p & 3;
            const f32x4 v = {(float)(py0 + i), (float)(px0 + jj), (float)co, (float)p};
            const int soff = __builtin_amdgcn_readlane((int)spill, p);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, lane_off, soff, 0);
        }
    }
}

template <int MODE>
int run(const char* name, int W, int ldo, bool tight) {
    const int H = 64, blocks = H / 16;
    const size_t n = (size_t)H * W * ldo;
    float* d; hipMalloc(&d, n * 4); hipMemset(d, 0xFF, n * 4);
    // tight: num_records = one past the largest LANE offset (voffset) of the grid + 16 bytes, i.e. smaller than the tensor
    const unsigned full = (unsigned)(n * 4);
    const unsigned lane_max = (unsigned)((((blocks - 1) * 16 + 12) * W + 12) * ldo + 48 + 12) * 4u + 16u;
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, d, W, ldo, H, tight ? lane_max : full);
    hipDeviceSynchronize();
    std::vector<float> h(n); hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost); hipFree(d);
    size_t bad = 0, unwritten = 0, first = (size_t)-1;
    for (int y = 0; y < H; ++y) for (int x = 0; x < 16; ++x) for (int c = 0; c < 64; c += 4) {
        const float* p = &h[((size_t)y * W + x) * ldo + c];
        const bool nan = p[0] != p[0];
        const bool ok = p[0] == (float)y && p[1] == (float)x && p[2] == (float)c && p[3] == (float)((y & 3) * 4 + (x & 3));
        if (nan) ++unwritten; else if (!ok) { ++bad; if (first == (size_t)-1) first = ((size_t)y * W + x) * ldo + c; }
    }
    printf("%-46s W=%3d ldo=%3d %s: %zu wrong, %zu not written of %d float4s%s\n", name, W, ldo, tight ? "resource ends at the last lane offset" : "resource = whole tensor           ",
           bad, unwritten, H * 16 * 16, bad ? "  <-- WRONG VALUES" : unwritten ? "  (dropped by the range check)" : "");
    return (int)bad;
}

int main() {
    int bad = 0;
    for (int tight = 0; tight < 2; ++tight)
        for (int ldo : {64, 68}) {
            bad += run<0>("A: pixel offset in voffset", 16, ldo, tight);
            bad += run<1>("B: pixel offset in soffset (scalar arithmetic)", 16, ldo, tight);
            bad += run<2>("C: pixel offset in soffset (v_readlane)", 16, ldo, tight);
        }
    printf(bad ? "MISMATCHES FOUND\n" : "all three forms store the same values; the range check sees soffset + voffset\n");
    return 0;
}
