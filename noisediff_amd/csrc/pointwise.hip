// pointwise.hip -- per-pixel GEMM on NHWC fp32 (1x1 conv == Linear on tokens), exact-fp32 MFMA.
//
// Replaces res_conv (models/archs/Diffusion_arch.py:156), Downsample's conv1x1 behind the
// pixel-unshuffle (:80-81), Mlp.fc1/fc2 (:345-347), ResnetBlock2.mlp (:176-179),
// FeedForward's two Linears (:410-419), AttnBlock.proj_out (:432), final_conv (:554) and the
// Attention to_qkv/to_out convs (:252-253).  CrossAttention itself (:379-402) needs no kernel:
// with the 1-token ISO context its output is the per-sample vector to_out(to_v(ctx)), which
// enters here as `src.vec` (added before LayerNorm) and `vec` (residual in the epilogue).
//
// One workgroup = 4 waves = BM consecutive pixels of one sample x BN output channels.  The A
// tile (BM x 64 channels per chunk) is staged through LDS with the prologue applied on the way
// (LayerNorm / SiLU / pixel-unshuffle addressing / virtual concat); B fragments come straight
// from the packed weight [k/4][coutP][4] as coalesced 16-byte loads (see conv3x3.hip).
#include <stdlib.h>
#include <type_traits>
#include "nd_common.h"

namespace {

constexpr int KC = 64;
constexpr int LDA = KC + 4;

struct PwArgs {
    nd_pointwise d;
    int m_tiles, n_tiles, coutP, cinP, total_wg;
};

// Epilogue shared by both kernels: accumulators -> LDS -> (bias, activation, residuals, fused ResnetBlock tail) -> global.
template <int MB, int NB, int JB = 0>
__device__ __forceinline__ void pw_epilogue(const PwArgs& a, f32x16 (&acc)[MB][NB], float* As, int b, int p0, int n0, unsigned long long* ph = nullptr) {   // ph: diagnostic phase stamps
    constexpr int BM = 2 * MB * 32, BN = 2 * NB * 32;
    constexpr int LDO = BN + 4;                               // output tile row stride in LDS (floats)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int col = lane & 31;
    const int HW = a.d.HW, W = a.d.W, Cout = a.d.cout;
    // ------------------------------------------------------------ epilogue
    // Accumulators go through LDS so that every global access of the epilogue (residual reads, the store) is a
    // 16-byte-per-lane, row-contiguous access like the staging loads -- 4x fewer memory instructions than storing
    // the MFMA layout directly (one dword per lane), which capped these HBM-bound layers at ~1.7 TB/s of writes.
    if (ph) ph[0] = __builtin_amdgcn_s_memtime();
    __syncthreads();                                          // all waves are done reading the A tile
    if (ph) ph[1] = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                As[((wm * MB + mb) * 32 + nd_acc_row(r, lane)) * LDO + (wn * NB + nb) * 32 + col] = acc[mb][nb][r];
    if (ph) ph[2] = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (ph) ph[3] = __builtin_amdgcn_s_memtime();
    {
        constexpr int QPR = BN / 4;                           // quads per tile row
        constexpr int RPI = 256 / QPR;                        // rows covered per pass of the 256 threads
        const int q = tid % QPR, rbase = tid / QPR;
        const int n = n0 + q * 4;
        const bool nvalid = n < Cout;                         // Cout % 4 == 0 is not required: handled below
        const int ns = (nvalid && n + 4 <= Cout) ? n : 0;
        const bool vec_ok = nvalid && n + 4 <= Cout;          // whole quad inside the tensor -> 16-byte path
        const f32x4 zero = {0, 0, 0, 0};
        f32x4 bias4 = zero, vadd4 = zero, gM = zero, gA = {1, 1, 1, 1}, gD = zero;
        if (vec_ok) {
            if (a.d.bias) bias4 = nd_ld4(a.d.bias + ns);
            if (a.d.vec) vadd4 = nd_ld4(a.d.vec + (size_t)b * Cout + ns);
            if (a.d.gn_t) {
                const float* m = a.d.gn_mad + (size_t)b * 3 * Cout + ns;
                gM = nd_ld4(m); gA = nd_ld4(m + Cout); gD = nd_ld4(m + 2 * Cout);
            }
        }
        float* out = a.d.out;
        constexpr int NJ = BM / RPI, JBLK = JB > 0 ? JB : NJ;  // residual reads in flight together: a block of JBLK rows per thread
        static_assert(NJ % JBLK == 0, "row block");
        if (vec_ok && p0 + BM <= HW && a.d.shuffle_c == 0) {
            // whole tile inside the image (the common case): row pointers advance by a scalar stride -- no per-row multiplies, clamps
            // or bounds tests.  Every VALU instruction here waits behind the 64-cycle MFMAs of the workgroup that shares the SIMDs.
            const size_t pix0 = (size_t)b * HW + p0 + rbase;
            float* orow = out + pix0 * a.d.ldo + n;
            const float* p_r0 = a.d.res0 ? a.d.res0 + pix0 * a.d.ldr0 + ns : nullptr;
            const float* p_r1 = a.d.res1 ? a.d.res1 + pix0 * a.d.ldr1 + ns : nullptr;
            const float* p_t = a.d.gn_t ? a.d.gn_t + pix0 * a.d.ldt + ns : nullptr;
            const float* lrow = As + rbase * LDO + q * 4;
            const int so = RPI * a.d.ldo, s0 = RPI * a.d.ldr0, s1 = RPI * a.d.ldr1, st = RPI * a.d.ldt;
            const int act = a.d.act;
            constexpr int FB = JB > 0 ? 4 : 2;                    // rows per thread in flight (a float4 each of the tile, two residuals, the tail input); 2 where 168 registers must do
            static_assert(NJ % FB == 0, "row block");
#pragma unroll 1
            for (int j0 = 0; j0 < NJ; j0 += FB) {
                f32x4 r0[FB], r1[FB], rt[FB];
                if (p_r0) {
#pragma unroll
                    for (int j = 0; j < FB; ++j) r0[j] = nd_ld4(p_r0 + (size_t)(j * s0));
                    p_r0 += (size_t)(FB * s0);
                }
                if (p_r1) {
#pragma unroll
                    for (int j = 0; j < FB; ++j) r1[j] = nd_ld4(p_r1 + (size_t)(j * s1));
                    p_r1 += (size_t)(FB * s1);
                }
                if (p_t) {
#pragma unroll
                    for (int j = 0; j < FB; ++j) rt[j] = nd_ld4(p_t + (size_t)(j * st));
                    p_t += (size_t)(FB * st);
                }
                f32x4 v[FB];
#pragma unroll
                for (int j = 0; j < FB; ++j) v[j] = nd_ld4(lrow + (j0 + j) * (RPI * LDO)) + bias4;
                if (act == ND_ACT_GELU) {
#pragma unroll
                    for (int j = 0; j < FB; ++j) v[j] = nd_gelu4(v[j]);
                } else if (act == ND_ACT_SILU) {
#pragma unroll
                    for (int j = 0; j < FB; ++j) v[j] = nd_silu4(v[j]);
                }
                if (a.d.vec) {
#pragma unroll
                    for (int j = 0; j < FB; ++j) v[j] += vadd4;
                }
                if (p_r0) {
#pragma unroll
                    for (int j = 0; j < FB; ++j) v[j] += r0[j];
                }
                if (p_r1) {
#pragma unroll
                    for (int j = 0; j < FB; ++j) v[j] += r1[j];
                }
                if (p_t) {
#pragma unroll
                    for (int j = 0; j < FB; ++j) v[j] += nd_silu4((rt[j] - gM) * gA + gD);
                }
#pragma unroll
                for (int j = 0; j < FB; ++j) nd_st4(orow + (size_t)(j * so), v[j]);
                orow += (size_t)(FB * so);
            }
        } else if (vec_ok) {
#pragma unroll 1
            for (int j0 = 0; j0 < NJ; j0 += JBLK) {
                f32x4 r0[JBLK], r1[JBLK], rt[JBLK];
#pragma unroll
                for (int j = 0; j < JBLK; ++j) {
                    const size_t pix = (size_t)b * HW + min(p0 + rbase + (j0 + j) * RPI, HW - 1);
                    r0[j] = a.d.res0 ? nd_ld4(a.d.res0 + pix * a.d.ldr0 + ns) : zero;
                    r1[j] = a.d.res1 ? nd_ld4(a.d.res1 + pix * a.d.ldr1 + ns) : zero;
                    rt[j] = a.d.gn_t ? nd_ld4(a.d.gn_t + pix * a.d.ldt + ns) : zero;
                }
#pragma unroll
                for (int j = 0; j < JBLK; ++j) {
                    const int r = rbase + (j0 + j) * RPI;
                    f32x4 v = nd_ld4(&As[r * LDO + q * 4]) + bias4;
                    if (a.d.act == ND_ACT_GELU) v = nd_gelu4(v);
                    else if (a.d.act == ND_ACT_SILU) v = nd_silu4(v);
                    v += r0[j] + r1[j] + vadd4;
                    if (a.d.gn_t) v += nd_silu4((rt[j] - gM) * gA + gD);
                    if (a.d.shuffle_c > 0) {       // ConvTranspose2d(2, stride 2): scatter to pixel (2y+p1, 2x+p2), channel c
                        const int p = p0 + r, y = p / W, x = p - y * W;
                        const int sub = n / a.d.shuffle_c, cch = n - sub * a.d.shuffle_c;
                        const int oy = 2 * y + (sub >> 1), ox = 2 * x + (sub & 1);
                        if (p < HW && oy < a.d.shuffle_h && ox < a.d.shuffle_w)
                            nd_st4(out + ((size_t)(b * a.d.shuffle_h + oy) * a.d.shuffle_w + ox) * a.d.ldo + cch, v);
                    } else if (p0 + r < HW) nd_st4(out + ((size_t)b * HW + p0 + r) * a.d.ldo + n, v);
                }
            }
        } else if (nvalid) {                                  // ragged channel tail (cout % 4 != 0): scalar path
            for (int j = 0; j < BM / RPI; ++j) {
                const int r = rbase + j * RPI;
                if (p0 + r >= HW) continue;
                const size_t pix = (size_t)b * HW + p0 + r;
                for (int e = 0; e < 4 && n + e < Cout; ++e) {
                    float v = nd_act(As[r * LDO + q * 4 + e] + (a.d.bias ? a.d.bias[n + e] : 0.0f), a.d.act);
                    if (a.d.res0) v += a.d.res0[pix * a.d.ldr0 + n + e];
                    if (a.d.res1) v += a.d.res1[pix * a.d.ldr1 + n + e];
                    if (a.d.vec) v += a.d.vec[(size_t)b * Cout + n + e];
                    if (a.d.gn_t) {
                        const float* m = a.d.gn_mad + (size_t)b * 3 * Cout + n + e;
                        v += nd_silu((a.d.gn_t[pix * a.d.ldt + n + e] - m[0]) * m[Cout] + m[2 * Cout]);
                    }
                    out[pix * a.d.ldo + n + e] = v;
                }
            }
        }
    }
}

// ---- software-pipelined variant for cin % 32 == 0 (every C >= 64 layer of the net): K in chunks of 32, A tile and
// weight fragments double-buffered (LDS / registers).  The global loads of chunk c+1 are issued before the MFMAs of
// chunk c and written to the other LDS buffer after them, so a workgroup's HBM/L2 round trip rides under its own
// matrix work instead of relying on a second resident workgroup to fill the gap; one barrier per chunk.
constexpr int PKC = 32, PLDA = PKC + 4;

#ifndef PW_PIPE_OCC
#define PW_PIPE_OCC 3        // waves per SIMD the 64-pixel tiles are compiled for (168 registers; 2 -> 3: -0.2 % per step same-box, 4 spills)
#endif
template <int MB, int NB, int MODE>
__global__ __launch_bounds__(256, (MB == 1 ? PW_PIPE_OCC : 2)) void pointwise_pipe_kernel(const PwArgs a) {
    constexpr int BM = 2 * MB * 32, BN = 2 * NB * 32;
    constexpr int SIT = BM / 32;                              // staging passes: BM rows x 8 channel quads / 256 threads
    constexpr int LDO = BN + 4;
    constexpr int ABUF = BM * PLDA;
    constexpr int SMEM = 2 * ABUF > BM * LDO ? 2 * ABUF : BM * LDO;
    __shared__ __attribute__((aligned(16))) float As[SMEM];   // two A buffers during the K loop, output tile in the epilogue

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, col = lane & 31;

    int lid = nd_xcd_remap(blockIdx.x, a.total_wg);
    const int nt = lid % a.n_tiles;  lid /= a.n_tiles;
    const int mt = lid % a.m_tiles;
    const int b = lid / a.m_tiles;

    const nd_src& s = a.d.src;
    const int HW = a.d.HW, Cin = a.d.cin;
    const int p0 = mt * BM, n0 = nt * BN;

    int a_off[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) a_off[mb] = ((wm * MB + mb) * 32 + col) * PLDA + 4 * half;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.d.weight), 0, a.cinP * a.coutP * 4, 0x00020000);
    const unsigned wvoff = (unsigned)((half * a.coutP + n0 + wn * NB * 32 + col) * 16);

    f32x16 acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.0f;

    const int quad = tid & 7, prow = tid >> 3;
    // per-thread rows of the A tile: the same pixels for every chunk
    size_t pixoff0[SIT], pixoff1[SIT];
    float rmean[SIT], rrstd[SIT];
#pragma unroll
    for (int it = 0; it < SIT; ++it) {
        const size_t pix = (size_t)b * HW + min(p0 + prow + it * 32, HW - 1);
        pixoff0[it] = pix * s.ld0;
        pixoff1[it] = pix * s.ld1;
        rmean[it] = 0.0f; rrstd[it] = 1.0f;
        if (MODE == ND_PRO_LAYERNORM) {                       // host: rowstats present (rows wider than one chunk)
            rmean[it] = s.rowstats[2 * pix];
            rrstd[it] = s.rowstats[2 * pix + 1];
        }
    }

    f32x4 bq[2][4][NB], raw[2][SIT];
    f32x4 pA[2], pB[2], pC[2];                                // per-chunk channel constants of the prologue
    auto stage_load = [&](auto sel, int cb) {
        constexpr int S = decltype(sel)::value;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                bq[S][g][nb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    wrsrc, wvoff, __builtin_amdgcn_readfirstlane((((cb >> 2) + 2 * g) * a.coutP + nb * 32) * 16), 0));
        const int c = cb + quad * 4;                          // < Cin: cin % 32 == 0
        const bool sec = c >= s.c0;
        const float* base = sec ? s.p1 : s.p0;
        const int cc = sec ? c - s.c0 : c;
#pragma unroll
        for (int it = 0; it < SIT; ++it) raw[S][it] = nd_ld4(base + (sec ? pixoff1[it] : pixoff0[it]) + cc);
        if (MODE == ND_PRO_LAYERNORM) {
            const f32x4 zero = {0, 0, 0, 0};
            pA[S] = nd_ld4(s.gamma + c); pB[S] = nd_ld4(s.beta + c);
            pC[S] = s.vec ? nd_ld4(s.vec + (size_t)b * Cin + c) : zero;
        }
        if (MODE == ND_PRO_AFFINE_SILU) {
            const float* m = s.mad + (size_t)b * 3 * Cin + c;
            pA[S] = nd_ld4(m); pB[S] = nd_ld4(m + Cin); pC[S] = nd_ld4(m + 2 * Cin);
        }
    };
    auto stage_write = [&](auto sel, float* dst) {
        constexpr int S = decltype(sel)::value;
#pragma unroll
        for (int it = 0; it < SIT; ++it) {
            const int r = prow + it * 32;
            f32x4 v = raw[S][it];
            if (MODE == ND_PRO_LAYERNORM) v = ((v + pC[S]) - rmean[it]) * rrstd[it] * pA[S] + pB[S];
            else if (MODE == ND_PRO_SILU) v = nd_silu4(v);
            else if (MODE == ND_PRO_AFFINE_SILU) v = nd_silu4((v - pA[S]) * pB[S] + pC[S]);
            else if (MODE == ND_PRO_LEAKY) v = nd_leaky4(v);
            const f32x4 zero = {0, 0, 0, 0};
            v = (p0 + r < HW) ? v : zero;
            nd_st4(&dst[r * PLDA + quad * 4], v);
        }
    };
    auto mma = [&](auto sel, const float* src) {
        constexpr int S = decltype(sel)::value;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 av[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) av[mb] = nd_ld4(&src[a_off[mb] + g * 8]);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[mb][nb] = nd_mfma(av[mb][k], bq[S][g][nb][k], acc[mb][nb]);
        }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;

    const int n_chunks = a.cinP / PKC;
    stage_load(S0{}, 0);
    stage_write(S0{}, As);
    __syncthreads();
    // No branch sits between a load and the MFMAs it overlaps: hipcc counts outstanding loads (vmcnt) along the path with
    // the fewest, so a conditional prefetch in front of an MFMA block makes that block wait for the prefetch itself.
    // sched_barriers: hipcc otherwise sinks the prefetch loads to just before their LDS writes (shorter live ranges),
    // which is exactly the overlap this kernel exists for
#define PW_STEP(LD, MM, WR)                         \
    LD; __builtin_amdgcn_sched_barrier(0);          \
    MM; __builtin_amdgcn_sched_barrier(0);          \
    WR; __syncthreads()
    int c = 0;
    for (; c + 2 < n_chunks; c += 2) {
        PW_STEP(stage_load(S1{}, (c + 1) * PKC), mma(S0{}, As), stage_write(S1{}, As + ABUF));
        PW_STEP(stage_load(S0{}, (c + 2) * PKC), mma(S1{}, As + ABUF), stage_write(S0{}, As));
    }
    if (n_chunks - c == 2) {
        PW_STEP(stage_load(S1{}, (c + 1) * PKC), mma(S0{}, As), stage_write(S1{}, As + ABUF));
        mma(S1{}, As + ABUF);
    } else {
        mma(S0{}, As);
    }
#undef PW_STEP

    pw_epilogue<MB, NB>(a, acc, As, b, p0, n0);
}

// ---- large-tile variant for the wide layers (cin % 64 == 0, cout % 128 == 0): one workgroup per CU, ONE WAVE PER SIMD with the
// whole register file -- the design that carries conv3x3_wino2 / wino4.  Tile = 128 pixels x 128 or 256 couts, a wave owns 64 x 64
// or 64 x 128 of it: 4 or 8 accumulators of 32 x 32 pinned in the AGPR half by inline-asm MFMAs, so one A fragment serves NB and one
// B fragment two MFMAs (6 operand registers per 8 MFMAs; the 64 x 128 tiles of the kernel above move 3 per 2).  K in chunks of 64:
// a barrier every 16 k MFMA cycles instead of every 2 k; the next chunk's activations are requested at the start of a chunk and
// written (prologue applied) to the other LDS buffer in its middle; weight fragments run three 8-channel groups ahead in a register
// ring that crosses chunk boundaries.  Everything between two MFMA groups is loads -- the fp32 MFMA and the VALU share issue cycles.
#define PWB_MFMA(acc, av, bv) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(av), "v"(bv))
constexpr int BKC = 64, BLDA = BKC + 4;

template <int NB, int MODE>
__global__ __launch_bounds__(256, 1) void pointwise_big_kernel(const PwArgs a) {
    constexpr int RS = 4, RD = RS - 1;                        // weight ring: slots, groups ahead
    constexpr int MB = 2, BM = 2 * MB * 32;                   // 128 pixels
    constexpr int SIT = BM / 16;                              // staging passes: 128 rows x 16 channel quads / 256 threads
    constexpr int ABUF = BM * BLDA;
    extern __shared__ __attribute__((aligned(16))) float Ab[]; // two A buffers during the K loop, the output tile in the epilogue

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, col = lane & 31;

    int lid = nd_xcd_remap(blockIdx.x, a.total_wg);
    const int nt = lid % a.n_tiles;  lid /= a.n_tiles;
    const int mt = lid % a.m_tiles;
    const int b = lid / a.m_tiles;

    const nd_src& s = a.d.src;
    const int HW = a.d.HW, Cin = a.d.cin;
    const int p0 = mt * BM, n0 = nt * (2 * NB * 32);

    int a_off[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) a_off[mb] = ((wm * MB + mb) * 32 + col) * BLDA + 4 * half;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.d.weight), 0, a.cinP * a.coutP * 4, 0x00020000);
    const unsigned wvoff = (unsigned)((half * a.coutP + n0 + wn * NB * 32 + col) * 16);
    // activations through buffer resources: 32-bit row offsets (the host checks that the sources stay below 4 GiB)
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(s.p0), 0, (int)((unsigned)a.d.B * HW * s.ld0 * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(s.p1 ? s.p1 : s.p0), 0,
                                                                          (int)((unsigned)a.d.B * HW * (s.p1 ? s.ld1 : s.ld0) * 4u), 0x00020000);

    f32x16 acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = nd_zero16();

    const int quad = tid & 15, prow = tid >> 4;
    unsigned rowoff0[SIT], rowoff1[SIT];                      // byte offset of this thread's rows (+ its channel quad)
    float rmean[SIT], rrstd[SIT];
#pragma unroll
    for (int it = 0; it < SIT; ++it) {
        const unsigned pix = (unsigned)b * HW + min(p0 + prow + it * 16, HW - 1);
        rowoff0[it] = pix * (unsigned)s.ld0 * 4u + quad * 16u;
        rowoff1[it] = pix * (unsigned)s.ld1 * 4u + quad * 16u;
        rmean[it] = 0.0f; rrstd[it] = 1.0f;
        if (MODE == ND_PRO_LAYERNORM) {                       // host: rowstats present
            rmean[it] = s.rowstats[2 * (size_t)pix];
            rrstd[it] = s.rowstats[2 * (size_t)pix + 1];
        }
    }

    f32x4 bq[RS][NB], av[2][MB], raw[SIT];
    f32x4 pA, pB, pC;                                          // per-chunk channel constants of the prologue
    auto load_b = [&](int slot, int cb, int g) {               // weight fragments of channels cb + 8g .. + 7
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
            bq[slot & (RS - 1)][nb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                wrsrc, wvoff, __builtin_amdgcn_readfirstlane((((cb >> 2) + 2 * g) * a.coutP + nb * 32) * 16), 0));
    };
    auto load_a = [&](int slot, const float* src, int g) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
        av[slot & 1][mb] = nd_ld4(&src[a_off[mb] + g * 8]);
    };
    auto stage_load = [&](int cb) {
        const bool sec = cb >= s.c0;                          // wave-uniform: a 64-channel chunk never straddles the sources (host check)
        const int soff = __builtin_amdgcn_readfirstlane((sec ? cb - s.c0 : cb) * 4);
        if (sec) {
#pragma unroll
            for (int it = 0; it < SIT; ++it) raw[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, rowoff1[it], soff, 0));
        } else {
#pragma unroll
            for (int it = 0; it < SIT; ++it) raw[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0, rowoff0[it], soff, 0));
        }
        const int c = cb + quad * 4;
        if (MODE == ND_PRO_LAYERNORM) {
            const f32x4 zero = {0, 0, 0, 0};
            pA = nd_ld4(s.gamma + c); pB = nd_ld4(s.beta + c);
            pC = s.vec ? nd_ld4(s.vec + (size_t)b * Cin + c) : zero;
        }
        if (MODE == ND_PRO_AFFINE_SILU) {
            const float* m = s.mad + (size_t)b * 3 * Cin + c;
            pA = nd_ld4(m); pB = nd_ld4(m + Cin); pC = nd_ld4(m + 2 * Cin);
        }
    };
    auto stage_write = [&](float* dst) {
#pragma unroll
        for (int it = 0; it < SIT; ++it) {
            const int r = prow + it * 16;
            f32x4 v = raw[it];
            if (MODE == ND_PRO_LAYERNORM) v = ((v + pC) - rmean[it]) * rrstd[it] * pA + pB;
            else if (MODE == ND_PRO_SILU) v = nd_silu4(v);
            else if (MODE == ND_PRO_AFFINE_SILU) v = nd_silu4((v - pA) * pB + pC);
            else if (MODE == ND_PRO_LEAKY) v = nd_leaky4(v);
            const f32x4 zero = {0, 0, 0, 0};
            v = (p0 + r < HW) ? v : zero;
            nd_st4(&dst[r * BLDA + quad * 4], v);
        }
    };

    const int n_chunks = a.cinP / BKC;
#ifdef PWB_STAMP                 // diagnostic (tools/pw_clock.py): phase stamps of a workgroup, written over the first 32 bytes of its output tile
    const unsigned long long st0 = __builtin_amdgcn_s_memtime();
#endif
    stage_load(0);
#pragma unroll
    for (int g = 0; g < RD; ++g) load_b(g, 0, g);
    stage_write(Ab);
    __syncthreads();
#ifdef PWB_STAMP
    const unsigned long long st1 = __builtin_amdgcn_s_memtime();
#endif
    for (int c = 0; c < n_chunks; ++c) {
        const float* cur = Ab + (c & 1) * ABUF;
        float* nxt = Ab + ((c + 1) & 1) * ABUF;
        const int cb = c * BKC, cbn = (c + 1 < n_chunks ? c + 1 : c) * BKC;      // behind the last chunk: a harmless re-stage of it
        load_a(0, cur, 0);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            // (no branch between a load and the MFMAs it overlaps, see above; sched_barriers pin load / MFMA order)
            if (g + RD < 8) load_b(g + RD, cb, g + RD); else load_b(g + RD, cbn, g + RD - 8);
            if (g + 1 < 8) load_a(g + 1, cur, g + 1);
            if (g == 0) stage_load(cbn);
            if (g == 4) stage_write(nxt);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) PWB_MFMA(acc[mb][nb], av[g & 1][mb][k], bq[g & 3][nb][k]);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                                       // the other buffer is complete, this one has been consumed
    }
#ifdef PWB_STAMP
    const unsigned long long st2 = __builtin_amdgcn_s_memtime();
#endif
    // the MFMAs are asm statements: hipcc does not know their results are still in flight
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) asm volatile("" : "+a"(acc[mb][nb]));
#ifdef PWB_STAMP
    unsigned long long ph[4];
    pw_epilogue<MB, NB, 8>(a, acc, Ab, b, p0, n0, ph);
#else
    pw_epilogue<MB, NB, 8>(a, acc, Ab, b, p0, n0);
#endif
#ifdef PWB_STAMP
    if (tid == 0) {                                           // (row 0, columns 0-7 of the tile are this wave's own stores: program order)
        unsigned long long* o = reinterpret_cast<unsigned long long*>(a.d.out + ((size_t)b * HW + p0) * a.d.ldo + n0);
        const unsigned long long st3 = __builtin_amdgcn_s_memtime();
        o[0] = st1 - st0;  o[1] = st2 - st1;  o[2] = st3 - st2;
        o[3] = (ph[1] - ph[0]) | ((ph[2] - ph[1]) << 16) | ((ph[3] - ph[2]) << 32) | ((st3 - ph[3]) << 48);    // 16 bits each: barrier, acc -> LDS, barrier, rows -> global
    }
#endif
}

// ---- SPLIT variant (r6; nd_pointwise_gemm_split_nhwc_f32): the same per-pixel GEMM with every product on the bf16 matrix pipe at the operands' FULL fp32
// significand.  v = v1 + v2 + v3 exactly (v1 = bf16(v), v2 = bf16(v - v1), round-to-nearest-even; v3 = v - v1 - v2: at most 8 significant bits); six of the nine
// term products are kept (w1 v1, w1 v2, w2 v1, w2 v2, w1 v3, w3 v1: the dropped ones are below 2^-25 of the product), each one v_mfma_f32_32x32x16_bf16 -- 16
// channels of a 32 x 32 tile cost 6 x 32 matrix cycles against 8 x 64 on v_mfma_f32_32x32x2_f32, and the bf16 instruction leaves the vector ALU to the split,
// the prologue and the epilogue (the fp32 one issues on the VALU's own lanes).  The wide 1x1 layers then sit against HBM, so the kernel is built to keep bytes
// in flight: TWO workgroups per CU (67.6 KB of LDS, 256 registers each), each one software-pipelined over 32-channel chunks.
//   * activations: 128 rows x 32 channels per chunk, thread = (row, 8-channel segment): two 16-byte loads, the prologue, the split ONCE per element
//     (11 VALU instructions per value pair), three ds_write_b128 into the chunk's term planes [term 3][row 128][64 bytes + 16 pad] (conflict-free b128 reads);
//   * weights: split once at pack time (nd_pack_pointwise_weight_split: [K step][term][32-cout tile][lane 64] x 16 bytes, lane (cout l & 31, K group l >> 5) holding
//     channels 16 ks + 8 (l >> 5) .. + 7), straight from the L2 into a register ring one K step ahead;
//   * a wave owns 64 pixels x 64 couts: 4 accumulators, 6 + 6 operand reads (16 bytes per lane each) per 24 MFMAs.
// Epilogue: pw_epilogue, as every other kernel of this file.  Accuracy against fp64: profiles/r6_split_gemm_accuracy.txt.
typedef __bf16 pw_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 pw_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned pw_u32x4 __attribute__((ext_vector_type(4)));
constexpr int SKC = 32;                                      // channels per chunk: two 16-channel K steps
constexpr int SROW = SKC * 2 + 16;                           // bytes per row of a term plane
constexpr int SPLANE = 128 * SROW, SABUF = 3 * SPLANE;       // one term plane, one chunk buffer (30 KB)
constexpr int SPLIT_LDS = 2 * SABUF > 128 * 132 * 4 ? 2 * SABUF : 128 * 132 * 4;

// (x, y) -> one dword of each of the three bf16 terms (x in the low half)
__device__ __forceinline__ void pw_split2(float x, float y, unsigned& w1, unsigned& w2, unsigned& w3) {
    w1 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x, y}, pw_bf16x2));
    const float rx = x - __builtin_bit_cast(float, w1 << 16), ry = y - __builtin_bit_cast(float, w1 & 0xFFFF0000u);
    w2 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{rx, ry}, pw_bf16x2));
    const float sx = rx - __builtin_bit_cast(float, w2 << 16), sy = ry - __builtin_bit_cast(float, w2 & 0xFFFF0000u);
    w3 = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, sy), __builtin_bit_cast(unsigned, sx), 0x07060302u);   // the upper halves ARE the third terms
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void pointwise_split_kernel(const PwArgs a) {
    constexpr int MB = 2, NB = 2, BM = 128, BN = 128;
    extern __shared__ __attribute__((aligned(16))) float Ab[];  // two chunk buffers during the K loop, the output tile in the epilogue
    char* const As = reinterpret_cast<char*>(Ab);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, col = lane & 31;

    int lid = nd_xcd_remap(blockIdx.x, a.total_wg);
    const int nt = lid % a.n_tiles;  lid /= a.n_tiles;
    const int mt = lid % a.m_tiles;
    const int b = lid / a.m_tiles;

    const nd_src& s = a.d.src;
    const int HW = a.d.HW, Cin = a.d.cin;
    const int p0 = mt * BM, n0 = nt * BN;

    // operand addresses: activations = A (rows = pixels), weights = B (columns = couts)
    int a_off[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) a_off[mb] = ((wm * MB + mb) * 32 + col) * SROW + half * 16;
    const int n32 = a.coutP / 32;                              // 32-cout tiles of the packed weight
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.d.weight), 0, (int)((long)a.cinP * a.coutP * 6), 0x00020000);
    const unsigned wvoff = (unsigned)(lane * 16);
    const int wtile = (n0 + wn * (NB * 32)) / 32;

    f32x16 acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.0f;

    const int seg = tid & 3, prow = tid >> 2;                  // staging: 8-channel segment, row (two passes of 64 rows)
    size_t pixoff0[2], pixoff1[2];
    float rmean[2], rrstd[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const size_t pix = (size_t)b * HW + min(p0 + prow + it * 64, HW - 1);
        pixoff0[it] = pix * s.ld0;
        pixoff1[it] = pix * s.ld1;
        rmean[it] = 0.0f; rrstd[it] = 1.0f;
        if (MODE == ND_PRO_LAYERNORM) {                       // host: rowstats present
            rmean[it] = s.rowstats[2 * pix];
            rrstd[it] = s.rowstats[2 * pix + 1];
        }
    }

    pw_u32x4 bq[2][3][NB];                                     // weight ring: K step parity x term x 32-cout tile
    f32x4 raw[2][2];                                           // the next chunk's activations in flight: pass x half segment
    f32x4 pA[2], pB[2], pC[2];                                 // per-chunk channel constants of the prologue
    auto load_b = [&](int slot, int ks) {                      // weight fragments of K step ks (beyond the last one: a harmless reload of it)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                bq[slot][t][nb] = __builtin_bit_cast(pw_u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    wrsrc, wvoff, __builtin_amdgcn_readfirstlane((((ks * 3 + t) * n32 + wtile + nb) * 64) * 16), 0));
    };
    auto stage_load = [&](int cb) {
        const int c = cb + seg * 8;                            // < Cin: cin % 32 == 0
        const bool sec = c >= s.c0;                            // a chunk never straddles the sources (host check)
        const float* base = sec ? s.p1 : s.p0;
        const int cc = sec ? c - s.c0 : c;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const float* p = base + (sec ? pixoff1[it] : pixoff0[it]) + cc;
            raw[it][0] = nd_ld4(p);
            raw[it][1] = nd_ld4(p + 4);
        }
        if (MODE == ND_PRO_LAYERNORM) {
            const f32x4 zero = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                pA[j] = nd_ld4(s.gamma + c + 4 * j); pB[j] = nd_ld4(s.beta + c + 4 * j);
                pC[j] = s.vec ? nd_ld4(s.vec + (size_t)b * Cin + c + 4 * j) : zero;
            }
        }
        if (MODE == ND_PRO_AFFINE_SILU) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float* m = s.mad + (size_t)b * 3 * Cin + c + 4 * j;
                pA[j] = nd_ld4(m); pB[j] = nd_ld4(m + Cin); pC[j] = nd_ld4(m + 2 * Cin);
            }
        }
    };
    auto stage_write = [&](char* dst) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int r = prow + it * 64;
            pw_u32x4 t1, t2, t3;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 v = raw[it][j];
                if (MODE == ND_PRO_LAYERNORM) v = ((v + pC[j]) - rmean[it]) * rrstd[it] * pA[j] + pB[j];
                else if (MODE == ND_PRO_SILU) v = nd_silu4(v);
                else if (MODE == ND_PRO_AFFINE_SILU) v = nd_silu4((v - pA[j]) * pB[j] + pC[j]);
                else if (MODE == ND_PRO_LEAKY) v = nd_leaky4(v);
                const f32x4 zero = {0, 0, 0, 0};
                v = (p0 + r < HW) ? v : zero;
                unsigned u1, u2, u3;
                pw_split2(v.x, v.y, u1, u2, u3);  t1[2 * j] = u1;  t2[2 * j] = u2;  t3[2 * j] = u3;
                pw_split2(v.z, v.w, u1, u2, u3);  t1[2 * j + 1] = u1;  t2[2 * j + 1] = u2;  t3[2 * j + 1] = u3;
            }
            char* o = dst + r * SROW + seg * 16;
            *reinterpret_cast<pw_u32x4*>(o) = t1;
            *reinterpret_cast<pw_u32x4*>(o + SPLANE) = t2;
            *reinterpret_cast<pw_u32x4*>(o + 2 * SPLANE) = t3;
        }
    };
    auto mfma = [&](pw_u32x4 av, pw_u32x4 bv, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pw_bf16x8, av), __builtin_bit_cast(pw_bf16x8, bv), c, 0, 0, 0);
    };
    auto kstep = [&](auto slot_c, const char* src, int k2) {   // one 16-channel K step of the chunk in `src` (k2 = 0, 1) on ring slot `slot_c`
        constexpr int S = decltype(slot_c)::value;
        pw_u32x4 av[MB][3];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int t = 0; t < 3; ++t) av[mb][t] = *reinterpret_cast<const pw_u32x4*>(src + t * SPLANE + a_off[mb] + k2 * 32);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                f32x16 c = acc[mb][nb];
                c = mfma(av[mb][0], bq[S][0][nb], c);
                c = mfma(av[mb][1], bq[S][0][nb], c);
                c = mfma(av[mb][0], bq[S][1][nb], c);
                c = mfma(av[mb][1], bq[S][1][nb], c);
                c = mfma(av[mb][2], bq[S][0][nb], c);
                c = mfma(av[mb][0], bq[S][2][nb], c);
                acc[mb][nb] = c;
            }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;

    const int n_chunks = a.cinP / SKC, last_ks = 2 * n_chunks - 1;
#ifdef PWS_STAMP                 // diagnostic (tools/pw_clock.py FORM=split): phase stamps of a workgroup, written over the first 32 bytes of its output tile
    const unsigned long long st0 = __builtin_amdgcn_s_memtime();
#endif
    stage_load(0);
    load_b(0, 0);
    load_b(1, 1);
    stage_write(As);
    __syncthreads();
#ifdef PWS_STAMP
    const unsigned long long st1 = __builtin_amdgcn_s_memtime();
#endif
    // (no branch between a load and the MFMAs it overlaps; sched_barriers keep hipcc from sinking the prefetches to their uses: see pointwise_pipe_kernel)
    for (int c = 0; c < n_chunks; ++c) {
        const char* cur = As + (c & 1) * SABUF;
        char* nxt = As + ((c + 1) & 1) * SABUF;
        const int cbn = (c + 1 < n_chunks ? c + 1 : c) * SKC;  // behind the last chunk: a harmless re-stage of it
        stage_load(cbn);
        __builtin_amdgcn_sched_barrier(0);
        kstep(S0{}, cur, 0);
        __builtin_amdgcn_sched_barrier(0);
        load_b(0, min(2 * c + 2, last_ks));
        __builtin_amdgcn_sched_barrier(0);
        kstep(S1{}, cur, 1);
        __builtin_amdgcn_sched_barrier(0);
        load_b(1, min(2 * c + 3, last_ks));
        stage_write(nxt);
        __syncthreads();                                       // the other buffer is complete, this one has been consumed
    }
#ifdef PWS_STAMP
    const unsigned long long st2 = __builtin_amdgcn_s_memtime();
    unsigned long long ph[4];
    pw_epilogue<MB, NB, 8>(a, acc, Ab, b, p0, n0, ph);
    if (tid == 0) {                                           // (row 0, columns 0-7 of the tile are this thread's own stores: program order)
        unsigned long long* o = reinterpret_cast<unsigned long long*>(a.d.out + ((size_t)b * HW + p0) * a.d.ldo + n0);
        const unsigned long long st3 = __builtin_amdgcn_s_memtime();
        o[0] = st1 - st0;  o[1] = st2 - st1;  o[2] = st3 - st2;
        o[3] = (ph[1] - ph[0]) | ((ph[2] - ph[1]) << 16) | ((ph[3] - ph[2]) << 32) | ((st3 - ph[3]) << 48);    // 16 bits each: barrier, acc -> LDS, barrier, rows -> global
    }
#else
    pw_epilogue<MB, NB, 8>(a, acc, Ab, b, p0, n0);
#endif
}

// NOTE on code shape: every global load below is unconditional (clamped address + select) and the prologue mode is a
// template parameter.  With data-dependent branches around loads hipcc emits s_waitcnt vmcnt(0) after each one, which
// turned the staging pass and the residual reads into chains of serialized HBM round trips (measured 2-5x slower).
template <int MB, int NB, int MODE>
__global__ __launch_bounds__(256) void pointwise_kernel(const PwArgs a) {
    constexpr int WM = 2, WN = 2;
    constexpr int BM = WM * MB * 32, BN = WN * NB * 32;
    constexpr int STAGE_IT = BM / 16;
    constexpr int LDO = BN + 4;                               // output tile row stride in LDS (floats)
    constexpr int SMEM = BM * (LDA > LDO ? LDA : LDO);
    __shared__ __attribute__((aligned(16))) float As[SMEM];   // A tile during the K loop, output tile in the epilogue

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, col = lane & 31;

    int lid = nd_xcd_remap(blockIdx.x, a.total_wg);
    const int nt = lid % a.n_tiles;  lid /= a.n_tiles;
    const int mt = lid % a.m_tiles;
    const int b = lid / a.m_tiles;

    const nd_src& s = a.d.src;
    const int HW = a.d.HW, W = a.d.W, Cin = a.d.cin;
    const int p0 = mt * BM, n0 = nt * BN;

    int a_off[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) a_off[mb] = ((wm * MB + mb) * 32 + col) * LDA + 4 * half;
    // weight fragments through buffer loads: resource + scalar chunk offset + constant lane offset (cheaper to issue
    // next to the MFMAs than 64-bit per-lane addresses, tools/microbench/mfma_issue.hip)
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.d.weight), 0, a.cinP * a.coutP * 4, 0x00020000);
    const unsigned wvoff = (unsigned)((half * a.coutP + n0 + wn * NB * 32 + col) * 16);

    f32x16 acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.0f;

    const int quad = tid & 15, prow = tid >> 4;
    const int Cs = s.unshuffle ? (s.c0 >> 2) : 1;
    const int Hs2 = 2 * (HW / max(W, 1));                 // source height for the unshuffle addressing

    for (int cb = 0; cb < a.cinP; cb += KC) {
        const int ng = min(8, (a.cinP - cb) >> 3);
        // all weight fragments of this chunk first: their L2 latency hides behind the activation loads below
        f32x4 bq[8][NB];
#pragma unroll
        for (int g = 0; g < 8; ++g)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                bq[g][nb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    wrsrc, wvoff, __builtin_amdgcn_readfirstlane((((cb >> 2) + 2 * min(g, ng - 1)) * a.coutP + nb * 32) * 16), 0));
        {
            const int c = cb + quad * 4;
            const bool cvalid = c < Cin;
            const int cs = cvalid ? c : 0;
            const float* base = s.p0;
            int ld = s.ld0, cc = cs, py = 0, px = 0;
            if (s.unshuffle) {
                const int sub = cs / Cs;
                cc = cs - sub * Cs; py = sub >> 1; px = sub & 1;
            } else if (cs >= s.c0) {
                base = s.p1; ld = s.ld1; cc = cs - s.c0;
            }
            f32x4 g4 = {1, 1, 1, 1}, b4 = {0, 0, 0, 0}, vec4 = {0, 0, 0, 0};
            f32x4 tM = {0, 0, 0, 0}, tA = {1, 1, 1, 1}, tD = {0, 0, 0, 0};
            if (MODE == ND_PRO_LAYERNORM) {
                g4 = nd_ld4(s.gamma + cs); b4 = nd_ld4(s.beta + cs);
                if (s.vec) vec4 = nd_ld4(s.vec + (size_t)b * Cin + cs);
            }
            if (MODE == ND_PRO_AFFINE_SILU) {
                const float* m = s.mad + (size_t)b * 3 * Cin + cs;
                tM = nd_ld4(m); tA = nd_ld4(m + Cin); tD = nd_ld4(m + 2 * Cin);
            }
            f32x4 raw[STAGE_IT];
            float rmean[STAGE_IT], rrstd[STAGE_IT];
#pragma unroll
            for (int it = 0; it < STAGE_IT; ++it) {
                const int p = min(p0 + prow + it * 16, HW - 1);
                size_t pix = (size_t)b * HW + p;
                if (s.unshuffle) {
                    const int y = p / W, x = p - y * W;
                    pix = ((size_t)b * Hs2 + 2 * y + py) * (2 * W) + 2 * x + px;
                }
                raw[it] = nd_ld4(base + pix * ld + cc);
                if (MODE == ND_PRO_LAYERNORM && s.rowstats) {
                    rmean[it] = s.rowstats[2 * pix];
                    rrstd[it] = s.rowstats[2 * pix + 1];
                }
            }
            if (MODE == ND_PRO_LAYERNORM && !s.rowstats) {
                // C <= 64: the 16 lanes {tid & ~15} hold the whole row (one quad each) -> two-pass statistics in
                // registers with DPP row reductions, no extra memory or LDS traffic
                const f32x4 zero = {0, 0, 0, 0};
#pragma unroll
                for (int it = 0; it < STAGE_IT; ++it) {
                    const f32x4 v = cvalid ? raw[it] + vec4 : zero;
                    const float mean = nd_row16_sum(v.x + v.y + v.z + v.w) / (float)Cin;
                    const f32x4 dv = cvalid ? v - mean : zero;
                    const float m2 = nd_row16_sum(dv.x * dv.x + dv.y * dv.y + dv.z * dv.z + dv.w * dv.w);
                    rmean[it] = mean;
                    rrstd[it] = rsqrtf(m2 / (float)Cin + 1e-5f);
                }
            }
            __syncthreads();   // previous chunk consumed
#pragma unroll
            for (int it = 0; it < STAGE_IT; ++it) {
                const int r = prow + it * 16;
                f32x4 v = raw[it];
                if (MODE == ND_PRO_LAYERNORM) v = ((v + vec4) - rmean[it]) * rrstd[it] * g4 + b4;
                else if (MODE == ND_PRO_SILU) v = nd_silu4(v);
                else if (MODE == ND_PRO_AFFINE_SILU) v = nd_silu4((v - tM) * tA + tD);
                else if (MODE == ND_PRO_LEAKY) v = nd_leaky4(v);
                const f32x4 zero = {0, 0, 0, 0};
                v = (cvalid && p0 + r < HW) ? v : zero;
                nd_st4(&As[r * LDA + quad * 4], v);
            }
        }
        __syncthreads();

        if (ng == 8) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                f32x4 av[MB];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) av[mb] = nd_ld4(&As[a_off[mb] + g * 8]);
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc[mb][nb] = nd_mfma(av[mb][k], bq[g][nb][k], acc[mb][nb]);
            }
        } else {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                if (g < ng) {
                    f32x4 av[MB];
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) av[mb] = nd_ld4(&As[a_off[mb] + g * 8]);
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb)
                                acc[mb][nb] = nd_mfma(av[mb][k], bq[g][nb][k], acc[mb][nb]);
                }
            }
        }
    }

    pw_epilogue<MB, NB>(a, acc, As, b, p0, n0);
}

// ---- narrow outputs (cout <= 8: final_conv, Diffusion_arch.py:554 -- 64 -> 4 at full resolution): on the MFMA tiles 60 of 64 output columns are padding and the layer
// runs at the padded product's rate (123 us at 256 x 256 x 16, 2.3 TB/s); here it is a streaming dot product on the VALU.  Four lanes share a pixel: lane part p reads
// the channel quads p, p + 4, ... (the four lanes of a pixel read 64 contiguous bytes per step), keeps CO partial sums (fmaf chains), the four parts are added with two
// DPP steps and lane 0 of the quad applies bias / activation / residuals and stores.  Weights [cin/4][CO][4] in LDS (<= 32 KB), read as broadcasts.
template <int CO>
__global__ __launch_bounds__(256) void pointwise_narrow_kernel(const PwArgs a) {
    __shared__ __attribute__((aligned(16))) float wl[256 * 8 * 4];
    const int tid = threadIdx.x, part = tid & 3;
    const nd_src& s = a.d.src;
    const int nq = a.d.cin >> 2, cout = a.d.cout;
    for (int i = tid; i < nq * CO; i += 256) {
        const int kq = i / CO, n = i - kq * CO;
        const f32x4 zero = {0, 0, 0, 0};
        nd_st4(wl + i * 4, n < cout ? nd_ld4(a.d.weight + ((size_t)kq * a.coutP + n) * 4) : zero);
    }
    __syncthreads();
    const long P = (long)a.d.B * a.d.HW;
    for (long base = (long)blockIdx.x * 64; base < P; base += (long)gridDim.x * 64) {
        const long pix = base + (tid >> 2), pc = pix < P ? pix : P - 1;      // (clamped: every lane takes part in the DPP steps)
        float acc[CO];
#pragma unroll
        for (int n = 0; n < CO; ++n) acc[n] = 0.0f;
        const float* r0 = s.p0 + pc * s.ld0;
        const float* r1 = s.p1 ? s.p1 + pc * s.ld1 - s.c0 : r0;
#pragma unroll 4
        for (int kq = part; kq < nq; kq += 4) {
            const int c = kq * 4;
            const f32x4 x = nd_ld4((c < s.c0 ? r0 : r1) + c);
#pragma unroll
            for (int n = 0; n < CO; ++n) {
                const f32x4 w = nd_ld4(wl + (kq * CO + n) * 4);
                acc[n] = fmaf(x.w, w.w, fmaf(x.z, w.z, fmaf(x.y, w.y, fmaf(x.x, w.x, acc[n]))));
            }
        }
#pragma unroll
        for (int n = 0; n < CO; ++n) {                        // parts 0 + 1, 2 + 3, then the two pairs (quad_perm [1 0 3 2], [2 3 0 1])
            acc[n] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc[n]), 0xB1, 0xF, 0xF, true));
            acc[n] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc[n]), 0x4E, 0xF, 0xF, true));
        }
        if (part == 0 && pix < P) {
#pragma unroll
            for (int q = 0; q < CO / 4; ++q) {
                if (q * 4 >= cout) break;
                f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
                if (a.d.bias) v += nd_ld4(a.d.bias + 4 * q);
                if (a.d.act == ND_ACT_GELU) v = nd_gelu4(v);
                else if (a.d.act == ND_ACT_SILU) v = nd_silu4(v);
                if (a.d.res0) v += nd_ld4(a.d.res0 + pix * a.d.ldr0 + 4 * q);
                if (a.d.res1) v += nd_ld4(a.d.res1 + pix * a.d.ldr1 + 4 * q);
                nd_st4(a.d.out + pix * a.d.ldo + 4 * q, v);
            }
        }
    }
}

// (cout, cin) -> [cinP/4][coutP][4], optional K permutation for pixel-unshuffled inputs
// transposed: `w` is (cin, cout) row-major -- the forward weight of the Linear whose DATA gradient dx = dy @ W this packing serves
__global__ void pack_pointwise_kernel(const float* __restrict__ w, float* __restrict__ out,
                                      int cin, int cout, int cinP, int coutP, int unshuffle_c, int transposed) {
    const size_t total = (size_t)cinP * coutP;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int e = i & 3;
        size_t r = i >> 2;
        const int n = r % coutP;
        const int q = r / coutP;
        const int kp = q * 4 + e;   // packed K index
        float v = 0.0f;
        if (n < cout && kp < cin) {
            int k = kp;
            if (unshuffle_c > 0) {   // packed order (p1 p2 c)  <-  torch order (c p1 p2)
                const int sub = kp / unshuffle_c, c = kp - sub * unshuffle_c;
                k = c * 4 + sub;
            }
            v = transposed ? w[(size_t)k * cout + n] : w[(size_t)n * cin + k];
        }
        out[i] = v;
    }
}

// The same packing for MANY weights in one launch (training: every Linear / 1x1 weight and its transposed data-gradient packing once per optimizer
// step -- 356 packings of a few microseconds each at d = 64 would otherwise be 356 launches).  blockIdx.y = item; `items` lives in device memory.
__global__ void pack_pointwise_batch_kernel(const nd_pack_item* __restrict__ items) {
    const nd_pack_item it = items[blockIdx.y];
    const int cinP = (it.cin + 7) / 8 * 8, coutP = (it.cout + 63) / 64 * 64;
    const size_t total = (size_t)cinP * coutP;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int e = i & 3;
        const size_t r = i >> 2;
        const int n = r % coutP, kp = (int)(r / coutP) * 4 + e;
        float v = 0.0f;
        if (n < it.cout && kp < it.cin) v = it.transposed ? it.w[(size_t)kp * it.cout + n] : it.w[(size_t)n * it.cin + kp];
        it.packed[i] = v;
    }
}

template <int MB, int NB>
void launch_pipe(const PwArgs& a, hipStream_t st) {
    const dim3 grid(a.total_wg), block(256);
    switch (a.d.src.mode) {
        case ND_PRO_LAYERNORM: hipLaunchKernelGGL((pointwise_pipe_kernel<MB, NB, ND_PRO_LAYERNORM>), grid, block, 0, st, a); break;
        case ND_PRO_SILU: hipLaunchKernelGGL((pointwise_pipe_kernel<MB, NB, ND_PRO_SILU>), grid, block, 0, st, a); break;
        case ND_PRO_AFFINE_SILU: hipLaunchKernelGGL((pointwise_pipe_kernel<MB, NB, ND_PRO_AFFINE_SILU>), grid, block, 0, st, a); break;
        case ND_PRO_LEAKY: hipLaunchKernelGGL((pointwise_pipe_kernel<MB, NB, ND_PRO_LEAKY>), grid, block, 0, st, a); break;
        default: hipLaunchKernelGGL((pointwise_pipe_kernel<MB, NB, ND_PRO_NONE>), grid, block, 0, st, a);
    }
}

template <int NB>
int launch_big(const PwArgs& a, hipStream_t st) {
    constexpr size_t lds = sizeof(float) * (2 * 128 * BLDA > 128 * (64 * NB + 4) ? 2 * 128 * BLDA : 128 * (64 * NB + 4));
    const dim3 grid(a.total_wg), block(256);
#define PWB_LAUNCH(MODE)                                                                                                      \
    {                                                                                                                         \
        static nd_device_once configured;                                                                                     \
        if (int e = nd_reserve_lds(configured, reinterpret_cast<const void*>(pointwise_big_kernel<NB, MODE>), lds, "nd_pointwise")) return e; \
        hipLaunchKernelGGL((pointwise_big_kernel<NB, MODE>), grid, block, lds, st, a);                                    \
    }
    switch (a.d.src.mode) {
        case ND_PRO_LAYERNORM: PWB_LAUNCH(ND_PRO_LAYERNORM) break;
        case ND_PRO_SILU: PWB_LAUNCH(ND_PRO_SILU) break;
        case ND_PRO_AFFINE_SILU: PWB_LAUNCH(ND_PRO_AFFINE_SILU) break;
        case ND_PRO_LEAKY: PWB_LAUNCH(ND_PRO_LEAKY) break;
        default: PWB_LAUNCH(ND_PRO_NONE)
    }
#undef PWB_LAUNCH
    return 0;
}

template <int MB, int NB>
void launch(const PwArgs& a, hipStream_t st) {
    const dim3 grid(a.total_wg), block(256);
    switch (a.d.src.mode) {
        case ND_PRO_LAYERNORM: hipLaunchKernelGGL((pointwise_kernel<MB, NB, ND_PRO_LAYERNORM>), grid, block, 0, st, a); break;
        case ND_PRO_SILU: hipLaunchKernelGGL((pointwise_kernel<MB, NB, ND_PRO_SILU>), grid, block, 0, st, a); break;
        case ND_PRO_AFFINE_SILU: hipLaunchKernelGGL((pointwise_kernel<MB, NB, ND_PRO_AFFINE_SILU>), grid, block, 0, st, a); break;
        case ND_PRO_LEAKY: hipLaunchKernelGGL((pointwise_kernel<MB, NB, ND_PRO_LEAKY>), grid, block, 0, st, a); break;
        default: hipLaunchKernelGGL((pointwise_kernel<MB, NB, ND_PRO_NONE>), grid, block, 0, st, a);
    }
}

// (cout, cin) row-major -> three bf16 terms in [K step][term][32-cout tile][lane 64][8]: slot i of lane l = W[32 tile + (l & 31)][16 ks + 8 (l >> 5) + i]
__global__ void pack_pointwise_split_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int cin, int cout, int cinP, int coutP) {
    const size_t total = (size_t)cinP * coutP;
    const int n32 = coutP / 32;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx & 7), l = (int)((idx >> 3) & 63);
        const size_t rest = idx >> 9;
        const int tile = (int)(rest % n32), ks = (int)(rest / n32);
        const int n = 32 * tile + (l & 31), k = 16 * ks + 8 * (l >> 5) + i;
        const float v = (n < cout && k < cin) ? w[(size_t)n * cin + k] : 0.0f;
        const __bf16 t1 = (__bf16)v;
        const float r1 = v - (float)t1;                       // exact
        const __bf16 t2 = (__bf16)r1;
        const float r2 = r1 - (float)t2;                      // exact, at most 8 significant bits: its upper half is the third term
        unsigned short* o = out + (((size_t)ks * 3 * n32 + tile) * 64 + l) * 8 + i;
        const size_t term = (size_t)n32 * 512;
        o[0] = __builtin_bit_cast(unsigned short, t1);
        o[term] = __builtin_bit_cast(unsigned short, t2);
        o[2 * term] = (unsigned short)(__builtin_bit_cast(unsigned, r2) >> 16);
    }
}

int launch_split(const PwArgs& a, hipStream_t st) {
    const dim3 grid(a.total_wg), block(256);
#define PWS_LAUNCH(MODE)                                                                                                      \
    {                                                                                                                         \
        static nd_device_once configured;                                                                                     \
        if (int e = nd_reserve_lds(configured, reinterpret_cast<const void*>(pointwise_split_kernel<MODE>), SPLIT_LDS, "nd_pointwise_split")) return e; \
        hipLaunchKernelGGL((pointwise_split_kernel<MODE>), grid, block, SPLIT_LDS, st, a);                                    \
    }
    switch (a.d.src.mode) {
        case ND_PRO_LAYERNORM: PWS_LAUNCH(ND_PRO_LAYERNORM) break;
        case ND_PRO_SILU: PWS_LAUNCH(ND_PRO_SILU) break;
        case ND_PRO_AFFINE_SILU: PWS_LAUNCH(ND_PRO_AFFINE_SILU) break;
        case ND_PRO_LEAKY: PWS_LAUNCH(ND_PRO_LEAKY) break;
        default: PWS_LAUNCH(ND_PRO_NONE)
    }
#undef PWS_LAUNCH
    return 0;
}

}  // namespace

extern "C" int64_t nd_pack_pointwise_weight_floats(int cin, int cout) {
    return (int64_t)nd_round_up(cin, 8) * nd_round_up(cout, 64);
}

extern "C" int nd_pack_pointwise_weight(const float* w, float* packed, int cin, int cout, int unshuffle_c, void* stream) {
    ND_REQUIRE(w && packed, ND_E_BADARG, "nd_pack_pointwise_weight: null pointer");
    ND_REQUIRE(cin > 0 && cout > 0, ND_E_BADARG, "nd_pack_pointwise_weight: non-positive size");
    ND_REQUIRE(unshuffle_c == 0 || (unshuffle_c * 4 == cin && unshuffle_c % 4 == 0), ND_E_SHAPE,
               "nd_pack_pointwise_weight: unshuffle_c=%d must be cin/4 and a multiple of 4", unshuffle_c);
    const int cinP = nd_round_up(cin, 8), coutP = nd_round_up(cout, 64);
    const size_t total = (size_t)cinP * coutP;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(pack_pointwise_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, packed, cin, cout, cinP, coutP, unshuffle_c, 0);
    return nd_launch_status("nd_pack_pointwise_weight");
}

extern "C" int nd_pack_pointwise_weight_t(const float* w_t, float* packed, int cin, int cout, void* stream) {
    ND_REQUIRE(w_t && packed, ND_E_BADARG, "nd_pack_pointwise_weight_t: null pointer");
    ND_REQUIRE(cin > 0 && cout > 0, ND_E_BADARG, "nd_pack_pointwise_weight_t: non-positive size");
    const int cinP = nd_round_up(cin, 8), coutP = nd_round_up(cout, 64);
    const size_t total = (size_t)cinP * coutP;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(pack_pointwise_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_t, packed, cin, cout, cinP, coutP, 0, 1);
    return nd_launch_status("nd_pack_pointwise_weight_t");
}

extern "C" int nd_pack_pointwise_weights_batch(const nd_pack_item* items_dev, int n_items, void* stream) {
    ND_REQUIRE(items_dev && n_items > 0 && n_items <= 65535, ND_E_BADARG, "nd_pack_pointwise_weights_batch: needs 1 .. 65535 items in device memory");
    hipLaunchKernelGGL(pack_pointwise_batch_kernel, dim3(32, (unsigned)n_items), dim3(256), 0, (hipStream_t)stream, items_dev);
    return nd_launch_status("nd_pack_pointwise_weights_batch");
}

namespace {
bool pw_pipe_takes(const nd_pointwise* d) {
    // software-pipelined kernel: whole 32-channel chunks, LayerNorm statistics from the pre-pass, plain addressing
    static const bool use_pipe = !(getenv("ND_PW_PIPE") && atoi(getenv("ND_PW_PIPE")) == 0);   // A/B knob (tools/ only)
    const nd_src& s = d->src;
    return use_pipe && d->cin % PKC == 0 && d->cin >= 2 * PKC && !s.unshuffle && (s.mode != ND_PRO_LAYERNORM || s.rowstats) && (s.c1 == 0 || nd_aligned16(s.p1));
}
// large tiles, one wave per SIMD: wide layers with enough 128-pixel tiles to fill the chip
long pw_big_tiles(const nd_pointwise* d) {                                  // 0: another kernel takes the layer
    static const int use_big = getenv("ND_PW_BIG") ? atoi(getenv("ND_PW_BIG")) : 1;            // A/B knob (tools/ only): 0 off
    const nd_src& s = d->src;
    if (!(pw_pipe_takes(d) && use_big && d->cin % BKC == 0 && d->cin >= 2 * BKC && d->cout % 128 == 0 && d->shuffle_c == 0 &&
          (s.c1 == 0 || s.c0 % BKC == 0) && (long)d->B * d->HW * s.ld0 * 4 < (1L << 32) && (long)d->B * d->HW * s.ld1 * 4 < (1L << 32))) return 0;
    const long tiles = (long)d->B * nd_cdiv(d->HW, 128) * (d->cout / 128);      // (256-cout tiles measured slower: 2429 vs 2318 us over the workload's wide layers)
    return tiles >= nd_device_cus() ? tiles : 0;
}

// layers the SPLIT kernel takes: whole 32-channel chunks (at least two), 128-cout tiles, plain addressing, LayerNorm statistics from the pre-pass
bool pw_split_takes(const nd_pointwise* d) {
    const nd_src& s = d->src;
    return d->cin % SKC == 0 && d->cin >= 2 * SKC && d->cout % 128 == 0 && !s.unshuffle && d->shuffle_c == 0 && (s.mode != ND_PRO_LAYERNORM || s.rowstats) &&
           (s.c1 == 0 || (s.c0 % SKC == 0 && nd_aligned16(s.p1)));
}

int pw_run(const nd_pointwise* d, void* stream, bool split = false) {
    ND_REQUIRE(d, ND_E_BADARG, "nd_pointwise: null descriptor");
    const nd_src& s = d->src;
    ND_REQUIRE(s.p0 && d->weight && d->out, ND_E_BADARG, "nd_pointwise: null tensor pointer");
    ND_REQUIRE(d->B > 0 && d->HW > 0 && d->cin > 0 && d->cout > 0, ND_E_BADARG, "nd_pointwise: non-positive size");
    ND_REQUIRE(d->cin % 4 == 0, ND_E_SHAPE, "nd_pointwise: cin=%d must be a multiple of 4", d->cin);
    ND_REQUIRE(s.c0 + s.c1 == d->cin && s.c0 % 4 == 0 && s.c1 % 4 == 0 && s.c0 > 0, ND_E_SHAPE,
               "nd_pointwise: source channels %d+%d do not match cin=%d", s.c0, s.c1, d->cin);
    ND_REQUIRE((s.c1 == 0) == (s.p1 == nullptr), ND_E_BADARG, "nd_pointwise: p1/c1 mismatch");
    ND_REQUIRE(s.ld0 % 4 == 0 && (s.c1 == 0 || s.ld1 % 4 == 0), ND_E_ALIGN, "nd_pointwise: pixel strides must be multiples of 4");
    ND_REQUIRE(nd_aligned16(s.p0) && nd_aligned16(s.p1) && nd_aligned16(d->weight) && nd_aligned16(s.vec) &&
               nd_aligned16(s.gamma) && nd_aligned16(s.beta) && nd_aligned16(s.mad), ND_E_ALIGN, "nd_pointwise: pointers must be 16-byte aligned");
    ND_REQUIRE(!s.upsample, ND_E_BADARG, "nd_pointwise: upsample is a conv3x3-only addressing mode");
    if (s.unshuffle) {
        ND_REQUIRE(s.c1 == 0 && d->W > 0 && d->HW % d->W == 0 && (s.c0 / 4) % 4 == 0 && s.mode == ND_PRO_NONE, ND_E_SHAPE,
                   "nd_pointwise: unshuffle needs one source, W | HW, c0/4 %% 4 == 0, no prologue");
        ND_REQUIRE(s.ld0 >= s.c0 / 4, ND_E_SHAPE, "nd_pointwise: ld0 < source channels");
    } else {
        ND_REQUIRE(s.ld0 >= s.c0 && (s.c1 == 0 || s.ld1 >= s.c1), ND_E_SHAPE, "nd_pointwise: ld < channels");
    }
    ND_REQUIRE(s.mode == ND_PRO_NONE || s.mode == ND_PRO_LAYERNORM || s.mode == ND_PRO_SILU || s.mode == ND_PRO_AFFINE_SILU ||
               s.mode == ND_PRO_LEAKY, ND_E_BADARG, "nd_pointwise: unsupported prologue %d", s.mode);
    if (d->shuffle_c > 0)
        ND_REQUIRE(d->cout == 4 * d->shuffle_c && d->shuffle_c % 4 == 0 && d->W > 0 && d->HW % d->W == 0 && d->shuffle_h > 0 && d->shuffle_w > 0 &&
                   d->ldo >= d->shuffle_c && !d->res0 && !d->res1 && !d->gn_t && !d->vec, ND_E_SHAPE,
                   "nd_pointwise: shuffle_c=%d needs cout == 4*shuffle_c, W | HW, an output size and no residual operands", d->shuffle_c);
    if (s.mode == ND_PRO_LAYERNORM) {
        ND_REQUIRE(s.gamma && s.beta && s.c1 == 0 && d->cin <= 1024, ND_E_SHAPE, "nd_pointwise: LayerNorm needs gamma/beta, one source, C <= 1024");
        ND_REQUIRE(d->cin <= 64 || s.rowstats, ND_E_BADARG, "nd_pointwise: LayerNorm over C=%d > 64 needs src.rowstats (nd_layernorm_stats_f32)", d->cin);
    }
    if (s.mode == ND_PRO_AFFINE_SILU) ND_REQUIRE(s.mad && s.c1 == 0, ND_E_BADARG, "nd_pointwise: affine prologue needs mad, one source");
    ND_REQUIRE(d->ldo >= d->cout || d->shuffle_c > 0, ND_E_SHAPE, "nd_pointwise: ldo < cout");
    ND_REQUIRE(d->ldo % 4 == 0 && (!d->res0 || d->ldr0 % 4 == 0) && (!d->res1 || d->ldr1 % 4 == 0) && (!d->gn_t || d->ldt % 4 == 0),
               ND_E_ALIGN, "nd_pointwise: output / residual pixel strides must be multiples of 4");
    ND_REQUIRE(nd_aligned16(d->out) && nd_aligned16(d->res0) && nd_aligned16(d->res1) && nd_aligned16(d->gn_t) && nd_aligned16(d->bias) &&
               nd_aligned16(d->vec) && nd_aligned16(d->gn_mad), ND_E_ALIGN, "nd_pointwise: epilogue pointers must be 16-byte aligned");
    ND_REQUIRE(!d->res0 || d->ldr0 >= d->cout, ND_E_SHAPE, "nd_pointwise: ldr0 < cout");
    ND_REQUIRE(!d->res1 || d->ldr1 >= d->cout, ND_E_SHAPE, "nd_pointwise: ldr1 < cout");
    ND_REQUIRE(!d->gn_t || (d->gn_mad && d->ldt >= d->cout), ND_E_BADARG, "nd_pointwise: gn_t needs gn_mad and ldt >= cout");
    ND_REQUIRE(d->act >= ND_ACT_NONE && d->act <= ND_ACT_SILU, ND_E_BADARG, "nd_pointwise: bad act");

    PwArgs a;
    a.d = *d;
    if (split) {
        ND_REQUIRE(pw_split_takes(d), ND_E_SHAPE, "nd_pointwise_gemm_split: %d -> %d is not a layer of the split kernel (cin %% 32, cin >= 64, cout %% 128, plain "
                   "addressing, LayerNorm with rowstats, concat on a 32-channel boundary: nd_pointwise_gemm_split_takes)", d->cin, d->cout);
        a.cinP = d->cin;  a.coutP = d->cout;
        a.m_tiles = nd_cdiv(d->HW, 128);
        a.n_tiles = d->cout / 128;
        const long wgs = (long)d->B * a.m_tiles * a.n_tiles;
        ND_REQUIRE(wgs < (1L << 31), ND_E_SHAPE, "nd_pointwise_gemm_split: grid too large");
        a.total_wg = (int)wgs;
        if (int e = launch_split(a, (hipStream_t)stream)) return e;
        return nd_launch_status("nd_pointwise_gemm_split_nhwc_f32");
    }
    a.cinP = nd_round_up(d->cin, 8);
    a.coutP = nd_round_up(d->cout, 64);
    // tiling: 128x128, 128x64, 64x64 -- first with >= 2 workgroups per CU, else the smallest
    static const int force_mb = getenv("ND_PW_MB") ? atoi(getenv("ND_PW_MB")) : 0;   // tuning knob (tools/ only)
    // 64-pixel tiles win on every layer of the bench workload (more workgroups in flight per CU; measured)
    static const int force_nb = getenv("ND_PW_NB") ? atoi(getenv("ND_PW_NB")) : 0;
    int mb = force_mb ? force_mb : 1, nb = force_nb ? force_nb : ((d->cout % 128 == 0) ? 2 : 1);
    auto count = [&](int mb_, int nb_) { return (long)d->B * nd_cdiv(d->HW, 64 * mb_) * nd_cdiv(d->cout, 64 * nb_); };
    if (nb == 2 && count(mb, 2) < 512) nb = 1;
    if (count(mb, nb) < 512) mb = 1;
    a.m_tiles = nd_cdiv(d->HW, 64 * mb);
    a.n_tiles = nd_cdiv(d->cout, 64 * nb);
    const long wg = count(mb, nb);
    ND_REQUIRE(wg < (1L << 31), ND_E_SHAPE, "nd_pointwise: grid too large");
    a.total_wg = (int)wg;
    hipStream_t st = (hipStream_t)stream;
    // narrow outputs: a streaming dot product instead of an MFMA tile that is 15/16 padding (geometry only; A/B knob ND_PW_NARROW=0, tools/ only)
    static const bool use_narrow = !(getenv("ND_PW_NARROW") && atoi(getenv("ND_PW_NARROW")) == 0);
    if (use_narrow && d->cout <= 8 && d->cout % 4 == 0 && d->cin <= 1024 && s.mode == ND_PRO_NONE && !s.unshuffle && d->shuffle_c == 0 && !d->gn_t && !d->vec &&
        (s.c1 == 0 || nd_aligned16(s.p1))) {
        const long blocks = ((long)d->B * d->HW + 63) / 64;
        const int grid = (int)(blocks < 8L * nd_device_cus() ? blocks : 8L * nd_device_cus());
        if (d->cout <= 4) hipLaunchKernelGGL((pointwise_narrow_kernel<4>), dim3(grid), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((pointwise_narrow_kernel<8>), dim3(grid), dim3(256), 0, st, a);
        return nd_launch_status("nd_pointwise_gemm_nhwc_f32");
    }
    const bool pipe = pw_pipe_takes(d);
    const long tiles = pw_big_tiles(d);
    if (tiles > 0) {
        a.m_tiles = nd_cdiv(d->HW, 128);
        a.n_tiles = d->cout / 128;
        a.total_wg = (int)tiles;
        if (int e = launch_big<2>(a, st)) return e;
        return nd_launch_status("nd_pointwise_gemm_nhwc_f32");
    }
    if (pipe) {
        if (mb == 2 && nb == 2) launch_pipe<2, 2>(a, st);
        else if (mb == 2) launch_pipe<2, 1>(a, st);
        else if (nb == 2) launch_pipe<1, 2>(a, st);
        else launch_pipe<1, 1>(a, st);
        return nd_launch_status("nd_pointwise_gemm_nhwc_f32");
    }
    if (mb == 2 && nb == 2) launch<2, 2>(a, st);
    else if (mb == 2) launch<2, 1>(a, st);
    else if (nb == 2) launch<1, 2>(a, st);
    else launch<1, 1>(a, st);
    return nd_launch_status("nd_pointwise_gemm_nhwc_f32");
}
}  // namespace

extern "C" int nd_pointwise_gemm_nhwc_f32(const nd_pointwise* d, void* stream) { return pw_run(d, stream); }

// The same operator with the products on the bf16 matrix pipe at full fp32 significand (pointwise_split_kernel above); `weight` is an
// nd_pack_pointwise_weight_split packing.  Same prologues, epilogues and errors; takes the layers nd_pointwise_gemm_split_takes names.
extern "C" int nd_pointwise_gemm_split_nhwc_f32(const nd_pointwise* d, void* stream) { return pw_run(d, stream, true); }

extern "C" int nd_pointwise_gemm_split_takes(const nd_pointwise* d) { return d && pw_split_takes(d) ? 1 : 0; }

extern "C" int64_t nd_pack_pointwise_weight_split_floats(int cin, int cout) {
    return (int64_t)nd_round_up(cin, 32) * nd_round_up(cout, 128) * 3 / 2;       // three bf16 terms per value, counted in floats (the arena's unit)
}

extern "C" int nd_pack_pointwise_weight_split(const float* w, float* packed, int cin, int cout, void* stream) {
    ND_REQUIRE(w && packed && nd_aligned16(packed), ND_E_BADARG, "nd_pack_pointwise_weight_split: null or unaligned pointer");
    ND_REQUIRE(cin > 0 && cout > 0, ND_E_BADARG, "nd_pack_pointwise_weight_split: non-positive size");
    const int cinP = nd_round_up(cin, 32), coutP = nd_round_up(cout, 128);
    const size_t total = (size_t)cinP * coutP;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(pack_pointwise_split_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, reinterpret_cast<unsigned short*>(packed), cin, cout, cinP, coutP);
    return nd_launch_status("nd_pack_pointwise_weight_split");
}
