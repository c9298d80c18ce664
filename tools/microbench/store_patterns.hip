// Microbenchmark (r3): what one CU's store path sustains for the store patterns an NHWC conv epilogue can produce on gfx950.
//   hipcc -O3 -w --offload-arch=gfx950 tools/microbench/store_patterns.hip -o tools/_build/store_patterns && tools/_build/store_patterns
// One workgroup of 4 waves per CU (one wave per SIMD, like conv3x3_wino4), every wave streams 16 x 16 pixel tiles x 16 couts of a
// [pixels][64 couts] fp32 tensor (256 bytes per pixel; the four waves of a workgroup own the four 16-cout quarters of the same pixels):
//   P0  buffer_store_dwordx4, lane = (tile = l & 15, cout quad = l >> 4): consecutive lanes hit different pixels (what wino4 did in r2)
//   P1  buffer_store_dword,   lane = (cout = l & 15, tile row = l >> 4): 16 consecutive lanes = 64 contiguous bytes (MFMA operands swapped)
//   P2  buffer_store_dwordx4, lane * 16 bytes contiguous (eight whole 128-byte lines per instruction: the upper bound)
//   P3  as P0 through global_store_dwordx4 (64-bit lane addresses)
//   P4  buffer_store_dwordx2, lane = (cout pair = l & 7, pixel = l >> 3): 8 consecutive lanes = 64 contiguous bytes
// Reports shader cycles per KB stored per CU (4 waves together) and the implied bytes per clock per CU, for 256 / 64 / 8 active CUs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int P>
__global__ __launch_bounds__(256, 1) void k(unsigned long long* out, float* g, int tiles_per_wg, long px_per_wg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* base = g + (size_t)blockIdx.x * px_per_wg * 64;                 // this workgroup's pixels
    i32x4 rs;
    rs[0] = __builtin_amdgcn_readfirstlane((int)(size_t)base); rs[1] = __builtin_amdgcn_readfirstlane((int)((size_t)base >> 32));
    rs[2] = (int)(px_per_wg * 256); rs[3] = 0x00020000;
    const f32x4 v = {1.0f + lane, 2.0f, 3.0f, 4.0f};
    const int W = 256;                                                      // image row = 256 pixels; a tile row = 16 tiles of 4x4 pixels... (16 x 16 pixel tile)
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < tiles_per_wg; ++t) {
        // tile t: pixels [16 t, 16 t + 16) of 16 consecutive image rows (row stride W pixels)
        const unsigned tile_px = (unsigned)((t / (W / 16)) * 16 * W + (t % (W / 16)) * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                if (P == 0 || P == 3) {      // lane: 4x4-pixel tile (l & 15) of the 16x16 tile, cout quad l >> 4; pixel (i, jj) of that tile
                    const int tl = lane & 15, kq = lane >> 4;
                    const unsigned px = tile_px + (4 * (tl >> 2) + i) * W + 4 * (tl & 3) + jj;
                    const unsigned off = px * 256 + wave * 64 + kq * 16;
                    if (P == 0) asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" :: "v"(v), "v"(off), "s"(rs) : "memory");
                    else { float* p = base + off / 4; asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory"); }
                } else if (P == 1) {         // lane: cout l & 15, tile row kq = l >> 4; register r = tile column: four dword stores per (i, jj)
                    const int co = lane & 15, kq = lane >> 4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const unsigned px = tile_px + (4 * kq + i) * W + 4 * r + jj;
                        const unsigned off = px * 256 + wave * 64 + co * 4;
                        asm volatile("buffer_store_dword %0, %1, %2, 0 offen" :: "v"(v.x), "v"(off), "s"(rs) : "memory");
                    }
                } else if (P == 2) {         // contiguous: the wave's 1 KB
                    const unsigned off = (tile_px + (i * 4 + jj) * 16) * 256 / 4 + wave * 1024 + lane * 16;
                    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" :: "v"(v), "v"(off), "s"(rs) : "memory");
                } else if (P == 4) {         // lane: cout pair l & 7 (8 bytes), pixel l >> 3 (8 pixels of a row); two stores per 16 pixels...
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const unsigned px = tile_px + (4 * i + jj) * W + 8 * r + (lane >> 3);
                        const unsigned off = px * 256 + wave * 64 + (lane & 7) * 8;
                        asm volatile("buffer_store_dwordx2 %0, %1, %2, 0 offen" :: "v"(f32x2{v.x, v.y}), "v"(off), "s"(rs) : "memory");
                    }
                }
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) atomicMax(out, t1 - t0);
}

template <int P>
void row(const char* name, unsigned long long* dout, float* g, size_t gbytes) {
    for (int wgs : {256, 64, 8}) {
        const int tiles = 256;                                  // 256 tiles x 16 x 16 pixels x 256 B = 16 MB per workgroup
        const long px_per_wg = (long)tiles * 256;
        if ((size_t)wgs * px_per_wg * 256 > gbytes) { printf("buffer too small\n"); return; }
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(dout, 0, 8);
            hipLaunchKernelGGL((k<P>), dim3(wgs), dim3(256), 0, 0, dout, g, tiles, px_per_wg);
            hipDeviceSynchronize();
        }
        unsigned long long h = 0;
        hipMemcpy(&h, dout, 8, hipMemcpyDeviceToHost);
        const double kb = tiles * 256.0 * 256.0 / 1024.0;       // per workgroup (4 waves x 16 couts = the 64 couts of every pixel)
        printf("%-58s %3d CUs: %7.1f cycles per KB per CU = %5.1f B/clk/CU\n", name, wgs, (double)h / kb, 1024.0 * kb / (double)h);
    }
}

int main() {
    unsigned long long* dout; float* g;
    const size_t gbytes = (size_t)5 << 30;
    hipMalloc(&dout, 64);
    if (hipMalloc(&g, gbytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    row<0>("P0 dwordx4, lanes = 16 pixels x 4 cout quads (r2 wino4)", dout, g, gbytes);
    row<1>("P1 dword, 16 lanes = 64 contiguous bytes (operands swapped)", dout, g, gbytes);
    row<2>("P2 dwordx4, whole wave contiguous (upper bound)", dout, g, gbytes);
    row<3>("P3 as P0 with global_store (64-bit addresses)", dout, g, gbytes);
    row<4>("P4 dwordx2, 8 lanes = 64 contiguous bytes", dout, g, gbytes);
    return 0;
}
