// Microbenchmark (r3): what VALU / LDS / store work costs on gfx950 next to v_mfma_f32_16x16x4_f32 -- alone, in the same wave, and from a
// second wave of the SIMD.
//   hipcc -O3 -w --offload-arch=gfx950 tools/microbench/mfma_overlap.hip -o tools/_build/mfma_overlap && tools/_build/mfma_overlap
// Section 1 (no MFMA): ticks per instruction of a run of 64 fillers, independent (8 rotating registers) or one dependent chain, with one and
//   with two waves per SIMD -- does a single resident wave reach the VALU's issue rate?
// Section 2 (same wave): 36 MFMAs (independent accumulators, 32.0 ticks each alone) + 36 N independent fillers, one behind every MFMA
//   ("spread") or all behind the 36 MFMAs ("clump"): extra ticks per filler.
// Section 3 (split): two waves per SIMD, wave A only MFMAs, wave B only fillers until A is done: ticks per MFMA of A, fillers per MFMA B got in.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { F_NONE, F_PKFMA, F_PKADD, F_FMA, F_EXP, F_RCP, F_ACCREAD, F_ACCWRITE, F_MOV, F_CNDMASK, F_MAD24, F_DSW_A, F_DSW_V, F_DSR, F_BST, F_DSW64, F_DSW32, F_DSW64X2, F_KINDS };
const char* NAMES[] = {"none", "v_pk_fma_f32", "v_pk_add_f32", "v_fma_f32", "v_exp_f32", "v_rcp_f32", "v_accvgpr_read", "v_accvgpr_write", "v_mov_b32",
                       "v_cndmask_b32", "v_mad_u32_u24", "ds_write_b128<-a", "ds_write_b128<-v", "ds_read_b128", "buffer_store_x4", "ds_write_b64", "ds_write_b32", "ds_write2_b64"};

struct Regs { double v[8]; float s[8]; f32x4 d[4]; f32x4 av[2]; unsigned lp, goff; int rs[4]; };

// n-th filler of a run; DEP: always register 0 (a dependent chain)
template <int KIND, bool DEP>
__device__ __forceinline__ void filler(int n, Regs& r) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const int i = DEP ? 0 : n & 7;
    if (KIND == F_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(r.v[i]) : "v"(r.v[(i + 1) & 7]));
    if (KIND == F_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r.v[i]) : "v"(r.v[(i + 1) & 7]));
    if (KIND == F_FMA) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(r.s[i]) : "v"(r.s[(i + 1) & 7]));
    if (KIND == F_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(r.s[i]));
    if (KIND == F_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(r.s[i]));
    if (KIND == F_ACCREAD) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r.s[i]) : "a"(r.av[0][i & 3]));
    if (KIND == F_ACCWRITE) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(r.av[1][i & 3]) : "v"(r.s[i]));
    if (KIND == F_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(r.s[i]) : "v"(r.s[DEP ? 0 : (i + 1) & 7]));
    if (KIND == F_CNDMASK) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(r.s[i]) : "v"(r.s[DEP ? 0 : (i + 1) & 7]), "v"(r.s[(i + 2) & 7]));
    if (KIND == F_MAD24) asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r.s[i]) : "v"(r.s[DEP ? 0 : (i + 1) & 7]), "v"(r.s[(i + 2) & 7]), "v"(r.s[(i + 3) & 7]));
    if (KIND == F_DSW_A) asm volatile("ds_write_b128 %0, %1" :: "v"(r.lp), "a"(r.av[0]) : "memory");
    if (KIND == F_DSW_V) asm volatile("ds_write_b128 %0, %1" :: "v"(r.lp), "v"(r.d[n & 3]) : "memory");
    if (KIND == F_DSR) asm volatile("ds_read_b128 %0, %1" : "=v"(r.d[n & 3]) : "v"(r.lp));
    if (KIND == F_DSW64) asm volatile("ds_write_b64 %0, %1" :: "v"(r.lp), "v"(r.v[n & 3]) : "memory");              // 8 bytes per lane (the lanes' 16-byte slots: half of each)
    if (KIND == F_DSW32) asm volatile("ds_write_b32 %0, %1" :: "v"(r.lp), "v"(r.s[n & 3]) : "memory");
    if (KIND == F_DSW64X2) asm volatile("ds_write2_b64 %0, %1, %2 offset1:1" :: "v"(r.lp), "v"(r.v[n & 3]), "v"(r.v[(n + 1) & 3]) : "memory");   // the same 16 bytes as b128, as two qwords
    if (KIND == F_BST) {
        i32x4 q = {r.rs[0], r.rs[1], r.rs[2], r.rs[3]};
        asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" :: "v"(r.d[n & 3]), "v"(r.goff), "s"(q) : "memory");
    }
}

__device__ __forceinline__ void init(Regs& r, float* g) {
    for (int i = 0; i < 8; ++i) { r.v[i] = 1.0 + i + threadIdx.x; r.s[i] = 0.5f + 0.01f * i; }
    for (int i = 0; i < 4; ++i) r.d[i] = f32x4{1.5f, 0.5f, 0.25f, 2.0f};
    r.av[0] = f32x4{1, 2, 3, 4}; r.av[1] = f32x4{1, 2, 3, 4};
    asm volatile("" : "+a"(r.av[0]), "+a"(r.av[1]));
    r.lp = (threadIdx.x & 255) * 16; r.goff = (threadIdx.x & 63) * 16;
    float* gdst = g + 4096 + (size_t)(blockIdx.x * 8 + (threadIdx.x >> 6)) * 1024;
    r.rs[0] = __builtin_amdgcn_readfirstlane((int)(size_t)gdst); r.rs[1] = __builtin_amdgcn_readfirstlane((int)((size_t)gdst >> 32));
    r.rs[2] = 1 << 20; r.rs[3] = 0x00020000;
}

__device__ __forceinline__ float sink(const Regs& r) {
    float s = r.av[0].x + r.av[1].y;
    for (int i = 0; i < 8; ++i) s += (float)r.v[i] + r.s[i];
    for (int i = 0; i < 4; ++i) s += r.d[i].x;
    return s;
}

// ---- section 1: fillers alone
template <int KIND, bool DEP, int THREADS>
__global__ __launch_bounds__(THREADS, 1) void k_alone(unsigned long long* out, float* g, int iters) {
    extern __shared__ float lds[];
    Regs r; init(r, g);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < 64; ++n) filler<KIND, DEP>(n, r);
        if (KIND >= F_DSW_A) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (sink(r) == 123.456f) out[3] = 1;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) atomicMax(out, t1 - t0);
}

// ---- section 2: MFMAs + fillers in one wave.  MODE 0 spread, 1 clump
template <int KIND, int N, int MODE>
__global__ __launch_bounds__(256, 1) void k_same(unsigned long long* out, float* g, int iters) {
    extern __shared__ float lds[];
    f32x4 acc[36];
    for (int p = 0; p < 36; ++p) acc[p] = f32x4{0, 0, 0, 0};
    Regs r; init(r, g);
    float a = threadIdx.x * 1e-3f, b = 1.0f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 36; ++p) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[p]) : "v"(a), "v"(b));
            if (MODE == 0)
#pragma unroll
                for (int n = 0; n < N; ++n) filler<KIND, false>(p * N + n, r);
        }
        if (MODE == 1)
#pragma unroll
            for (int n = 0; n < 36 * N; ++n) filler<KIND, false>(n, r);
        if (KIND >= F_DSW_A) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = sink(r);
    for (int p = 0; p < 36; ++p) s += acc[p][0];
    if (s == 123.456f) out[3] = 1;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) atomicMax(out, t1 - t0);
}

// ---- section 3: two waves per SIMD, MFMA wave + filler wave (8 accumulators only: 128 + 128 registers per wave)
template <int KIND>
__global__ __launch_bounds__(512, 1) void k_split(unsigned long long* out, float* g, int iters) {
    extern __shared__ float lds[];
    f32x4 acc[8];
    for (int p = 0; p < 8; ++p) acc[p] = f32x4{0, 0, 0, 0};
    Regs r; init(r, g);
    float a = threadIdx.x * 1e-3f, b = 1.0f;
    volatile int* flag = reinterpret_cast<volatile int*>(lds + 8192);
    if (threadIdx.x == 0) *flag = 0;
    __syncthreads();
    unsigned long long fills = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x < 256) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int p = 0; p < 32; ++p) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[p & 7]) : "v"(a), "v"(b));
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        if ((threadIdx.x & 63) == 0) atomicAdd(const_cast<int*>(flag), 1);
    } else {
        while (*flag < 4) {                                  // ends when the four MFMA waves are done
#pragma unroll
            for (int n = 0; n < 64; ++n) filler<KIND, false>(n, r);
            if (KIND >= F_DSW_A) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            fills += 64;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = sink(r);
    for (int p = 0; p < 8; ++p) s += acc[p][0];
    if (s == 123.456f) out[3] = 1;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) {
        if (threadIdx.x < 256) atomicMax(out, t1 - t0);
        else atomicMax(out + 1, fills);
    }
}

template <typename K>
void launch(K kern, int threads, unsigned long long* dout, float* g, int iters, unsigned long long (&h)[2]) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 40960);
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(dout, 0, 32);
        hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 40960, 0, dout, g, iters);
        hipDeviceSynchronize();
    }
    hipMemcpy(h, dout, 16, hipMemcpyDeviceToHost);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) printf("   !! %s\n", hipGetErrorString(e));
}

template <int KIND>
void rows(unsigned long long* dout, float* g) {
    unsigned long long h[2];
    const int iters = 400;
    double v[8];
    launch(k_alone<KIND, false, 256>, 256, dout, g, iters, h); v[0] = (double)h[0] / (iters * 64.0);
    launch(k_alone<KIND, true, 256>, 256, dout, g, iters, h);  v[1] = (double)h[0] / (iters * 64.0);
    launch(k_alone<KIND, false, 512>, 512, dout, g, iters, h); v[2] = (double)h[0] / (iters * 64.0);
    launch(k_same<KIND, 1, 0>, 256, dout, g, iters, h); v[3] = (double)h[0] / (iters * 36.0) - 32.0;
    launch(k_same<KIND, 2, 0>, 256, dout, g, iters, h); v[4] = ((double)h[0] / (iters * 36.0) - 32.0) / 2;
    launch(k_same<KIND, 1, 1>, 256, dout, g, iters, h); v[5] = (double)h[0] / (iters * 36.0) - 32.0;
    launch(k_same<KIND, 4, 1>, 256, dout, g, iters, h); v[6] = ((double)h[0] / (iters * 36.0) - 32.0) / 4;
    launch(k_split<KIND>, 512, dout, g, iters, h);
    printf("%-17s alone: %5.2f indep / %5.2f dep chain / %5.2f per wave with 2 waves per SIMD | next to MFMAs, extra per filler: spread N=1 %5.2f, N=2 %5.2f; "
           "clump 36 %5.2f, 144 %5.2f | split: MFMA wave %5.2f ticks/MFMA, filler wave %5.2f fillers/MFMA\n",
           NAMES[KIND], v[0], v[1], v[2], v[3], v[4], v[5], v[6], (double)h[0] / (iters * 32.0), (double)h[1] / (iters * 32.0));
}

int main() {
    unsigned long long* dout; float* g;
    hipMalloc(&dout, 64); hipMalloc(&g, 64 << 20); hipMemset(g, 0, 64 << 20);
    unsigned long long h[2];
    launch(k_same<F_NONE, 0, 0>, 256, dout, g, 400, h);
    printf("36 MFMAs alone: %.2f ticks/MFMA\n", (double)h[0] / (400 * 36.0));
    rows<F_PKFMA>(dout, g); rows<F_PKADD>(dout, g); rows<F_FMA>(dout, g); rows<F_EXP>(dout, g); rows<F_RCP>(dout, g);
    rows<F_ACCREAD>(dout, g); rows<F_ACCWRITE>(dout, g); rows<F_MOV>(dout, g); rows<F_CNDMASK>(dout, g); rows<F_MAD24>(dout, g);
    rows<F_DSW_A>(dout, g); rows<F_DSW_V>(dout, g); rows<F_DSR>(dout, g); rows<F_BST>(dout, g);
    rows<F_DSW64>(dout, g); rows<F_DSW32>(dout, g); rows<F_DSW64X2>(dout, g);
    return 0;
}
