// pwchain.hip -- two or three chained per-pixel Linear layers (1x1 convs) in one kernel, activations never leaving
// registers:   Mlp = fc2(GELU(fc1(x)))                                   (Diffusion_arch.py:340-356)
//              AttnBlock tail = proj_out(ff.net.2(GELU(ff.net.0(LN(x + v)))) + x + v) + x        (:405-443, v = the 1-token
//              cross-attention output, a per-sample vector)
// Unfused, each layer streams its input and output through HBM (the 2C-wide hidden tensor of the FeedForward alone is
// 2 x 268 MB per call at 256x256x16); fused, a chain reads x once and writes y once and is bounded by the fp32 MFMA.
//
// The products are computed TRANSPOSED, D^T = W . X^T, one 32-pixel column tile per wave:
//   v_mfma_f32_32x32x2_f32  A = weights (lane l: W[n = l & 31][k(j, l >> 5)])   B = activations (lane l: X[pixel l & 31][k(j, l >> 5)])
//   accumulator register r of lane (pixel, half) = output channel 32*nt + (r & 3) + 8*(r >> 2) + 4*half
// With K paired as k(j, half) = 8*(j >> 2) + (j & 3) + 4*half, accumulator register r of n-tile nt IS operand j = 16*nt + r
// of the next layer: the chain flows register to register with no shuffle, no LDS round trip, and the residual / bias /
// activation are plain per-register VALU work.  The input row of a pixel is loaded in the same pairing (lane (pixel, half)
// reads channels 8q + 4*half .. +3), so LayerNorm statistics are an in-lane sum plus one cross-half exchange, and the
// output is written as 16-byte pieces.
//
// Weights (packed in operand order by nd_pack_chain_weight), biases and LayerNorm gamma/beta of all stages sit in LDS for
// the lifetime of the persistent workgroup (80 KB at C = 64): an A fragment is one ds_read_b128 per four MFMAs.  Three or
// four waves per SIMD (768 / 1024 threads) let one wave's activation VALU run under the others' MFMAs.
//
// SPLIT instances (r6; nd_pointwise_chain_split_nhwc_f32): the same chain with every product on the bf16 matrix pipe at the operands' FULL
// fp32 significand.  x = x1 + x2 + x3 exactly, x1 = bf16(x), x2 = bf16(x - x1) (round-to-nearest-even, v_cvt_pk_bf16_f32), x3 = x - x1 - x2 (both
// remainders are exact in fp32 and the second has at most 8 significant bits); weights are split once at pack time, activations per 16-channel
// K step in registers (11 VALU instructions per value pair).  Six of the nine term products are kept -- w1 x1, w1 x2, w2 x1, w2 x2, w1 x3, w3 x1;
// the dropped ones are below 2^-25 of the product, under the rounding of an fp32 FMA -- each one v_mfma_f32_32x32x16_bf16 (products of bf16 pairs
// are exact in fp32; accumulation in fp32).  16 channels cost 6 x 32 matrix cycles per 32 x 32 tile against 8 x 64 for v_mfma_f32_32x32x2_f32
// (2.67x), and -- unlike the fp32 instruction, which issues on the VALU's own lanes -- leave 24 of every 32 cycles of vector issue to the split,
// LayerNorm and GELU work of the other waves.  Same operand pairing as the fp32 form: slot (K step s, lane half h, i < 8) = channel
// 16 s + 8 (i >> 2) + 4 h + (i & 3), which is both "registers (2s, 2s+1) of the input row's float4s" and "registers 8 (s & 1) .. + 7 of n-tile
// s >> 1 of the previous stage's accumulators": the chain still flows register to register.  Weight terms sit in LDS (120 KB at C = 64).
// Accuracy against fp64: profiles/r6_split_gemm_accuracy.txt (the r5 study of the same split on the F(4x4) kernel: rms error 0.94 of fp32's).
#include <type_traits>
#include "nd_common.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// (x, y) -> one dword of each of the three bf16 terms (x in the low half): x == t1 + t2 + t3 exactly
__device__ __forceinline__ void chain_split2(float x, float y, unsigned& w1, unsigned& w2, unsigned& w3) {
    w1 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x, y}, bf16x2_t));
    const float rx = x - __builtin_bit_cast(float, w1 << 16), ry = y - __builtin_bit_cast(float, w1 & 0xFFFF0000u);
    w2 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{rx, ry}, bf16x2_t));
    const float sx = rx - __builtin_bit_cast(float, w2 << 16), sy = ry - __builtin_bit_cast(float, w2 & 0xFFFF0000u);
    w3 = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, sy), __builtin_bit_cast(unsigned, sx), 0x07060302u);   // the upper halves ARE the third terms
}

__device__ __forceinline__ f32x16 chain_mfma_bf16(u32x4_t a, u32x4_t b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

struct ChainArgs {
    nd_chain d;
    int n_tiles;            // 32-pixel tiles in total
    int tiles_per_sample;
};

template <int K, int N>
__device__ __forceinline__ void chain_gemm(const float (&in)[K / 2], const float* __restrict__ w, const float* __restrict__ bias,
                                           f32x16 (&acc)[N / 32], const int lane) {
    const int half = lane >> 5;
#pragma unroll
    for (int nt = 0; nt < N / 32; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b4 = nd_ld4(bias + 32 * nt + 8 * g + 4 * half);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[nt][4 * g + i] = b4[i];
        }
#pragma unroll
    for (int jq = 0; jq < K / 8; ++jq)
#pragma unroll
        for (int nt = 0; nt < N / 32; ++nt) {
            const f32x4 a4 = nd_ld4(w + ((nt * (K / 8) + jq) * 64 + lane) * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[nt] = nd_mfma(a4[i], in[4 * jq + i], acc[nt]);
        }
}

// SPLIT: `w` = the stage's weight terms in LDS, [n-tile][K step][term 3][lane 64] x 16 bytes (eight bf16: the lane's slots of the K step).
// One software-pipelined stream per stage: the three weight reads of group (s, nt) + 1 are issued BEFORE the six MFMAs of group (s, nt) -- left to itself
// hipcc puts every ds_read_b128 right in front of its first use, one exposed LDS round trip per pair of MFMAs (profiles/r6a_chain_wave_states.txt: half the
// wave time parked) -- and the split of K step s + 1 (44 VALU instructions) sits in the same scheduling region as the MFMAs of K step s.
template <int K, int N>
__device__ __forceinline__ void chain_gemm_split(const float (&in)[K / 2], const char* __restrict__ w, const float* __restrict__ bias,
                                                 f32x16 (&acc)[N / 32], const int lane) {
    static_assert(K % 16 == 0, "whole 16-channel K steps");
    constexpr int S = K / 16, NT = N / 32;
    const int half = lane >> 5;
    const u32x4_t* const wl = reinterpret_cast<const u32x4_t*>(w) + lane;
    u32x4_t wq[2][3], xs[2][3];
    auto load_w = [&](int slot, int s_, int nt_) {
        const u32x4_t* p = wl + ((nt_ * S + s_) * 3) * 64;
        wq[slot][0] = p[0];  wq[slot][1] = p[64];  wq[slot][2] = p[128];
    };
    auto split = [&](int slot, int s_) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned t1, t2, t3;
            chain_split2(in[8 * s_ + 2 * j], in[8 * s_ + 2 * j + 1], t1, t2, t3);
            xs[slot][0][j] = t1;  xs[slot][1][j] = t2;  xs[slot][2][j] = t3;
        }
    };
    load_w(0, 0, 0);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b4 = nd_ld4(bias + 32 * nt + 8 * g + 4 * half);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[nt][4 * g + i] = b4[i];
        }
    split(0, 0);
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int g = s * NT + nt;
            if (g + 1 < S * NT) load_w((g + 1) & 1, nt + 1 < NT ? s : s + 1, nt + 1 < NT ? nt + 1 : 0);
            __builtin_amdgcn_sched_barrier(0);
            if (nt == 0 && s + 1 < S) split((s + 1) & 1, s + 1);
            const u32x4_t w1 = wq[g & 1][0], w2 = wq[g & 1][1], w3 = wq[g & 1][2];
            const u32x4_t x1 = xs[s & 1][0], x2 = xs[s & 1][1], x3 = xs[s & 1][2];
            acc[nt] = chain_mfma_bf16(w1, x1, acc[nt]);
            acc[nt] = chain_mfma_bf16(w1, x2, acc[nt]);
            acc[nt] = chain_mfma_bf16(w2, x1, acc[nt]);
            acc[nt] = chain_mfma_bf16(w2, x2, acc[nt]);
            acc[nt] = chain_mfma_bf16(w1, x3, acc[nt]);
            acc[nt] = chain_mfma_bf16(w3, x1, acc[nt]);
            __builtin_amdgcn_sched_barrier(0);
        }
}

template <int N>
__device__ __forceinline__ void chain_act(f32x16 (&acc)[N / 32], const int act) {
    if (act == ND_ACT_NONE) return;
#pragma unroll
    for (int nt = 0; nt < N / 32; ++nt)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            if (act == ND_ACT_GELU) {                                    // two accumulator registers at a time on the packed pipe
                const f32x2 g = nd_gelu2(f32x2{acc[nt][r], acc[nt][r + 1]});
                acc[nt][r] = g.x;  acc[nt][r + 1] = g.y;
            } else { acc[nt][r] = nd_silu(acc[nt][r]);  acc[nt][r + 1] = nd_silu(acc[nt][r + 1]); }
        }
}

// acc[nt][r] += x[q = 4*nt + (r >> 2)][r & 3] (- v): the residual lives in the input registers
template <int K0, int N>
__device__ __forceinline__ void chain_res(f32x16 (&acc)[N / 32], const f32x4 (&x)[K0 / 8], const float* vstash, const int res, const int half) {
    if (res == ND_CHAIN_RES_NONE) return;
#pragma unroll
    for (int nt = 0; nt < N / 32; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int q = 4 * nt + g;
            if (q < K0 / 8) {
                f32x4 v = x[q];
                if (res == ND_CHAIN_RES_INPUT_RAW && vstash) v -= nd_ld4(vstash + 8 * q + 4 * half);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[nt][4 * g + i] += v[i];
            }
        }
}

// SPLIT: the same stores through a buffer resource, a lane beyond cout writing out of range (dropped) instead of branching around the store
template <int N>
__device__ __forceinline__ void chain_store_buf(const f32x16 (&acc)[N / 32], __amdgpu_buffer_rsrc_t rso, const unsigned row_bytes, const int cout, const int half) {
#pragma unroll
    for (int nt = 0; nt < N / 32; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n0 = 32 * nt + 8 * g + 4 * half;
            const f32x4 v = {acc[nt][4 * g], acc[nt][4 * g + 1], acc[nt][4 * g + 2], acc[nt][4 * g + 3]};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rso, n0 < cout ? row_bytes + n0 * 4u : 0xFFFFFFF0u, 0, 0);
        }
}

template <int N>
__device__ __forceinline__ void chain_store(const f32x16 (&acc)[N / 32], float* row, const int cout, const int half) {
#pragma unroll
    for (int nt = 0; nt < N / 32; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n0 = 32 * nt + 8 * g + 4 * half;
            if (n0 < cout) {
                const f32x4 v = {acc[nt][4 * g], acc[nt][4 * g + 1], acc[nt][4 * g + 2], acc[nt][4 * g + 3]};
                nd_st4(row + n0, v);
            }
        }
}

// K0: stage-0 input channels rounded up to 8; N1, N2, N3: stage widths rounded up to 32 (N3 = 0: two stages)
// waves per workgroup: as many as the register budget of the widest stage allows (3 or 4 per SIMD) -- the activation
// VALU of one wave runs under the MFMAs of the others
// SPLIT: two or three waves per SIMD -- the next tile's rows are prefetched into registers (the bf16 products leave a tile too short to hide a cold HBM
// round trip behind the other waves), and nothing may spill: a scratch reload parks the wave like any other memory wait
constexpr int chain_threads(int n1, bool split = false, int k0 = 0) { return split ? (n1 >= 128 || k0 >= 48 ? 512 : 768) : (n1 >= 128 ? 768 : 1024); }

#ifdef CHAIN_STAMP               // diagnostic build (tools/chain_clock.py): per-wave phase sums in shader cycles
__device__ unsigned long long chain_dbg[4096 * 16 * 10];
#define CH_T0() (stamp_t = __builtin_amdgcn_s_memtime())
#define CH_ACC(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); stamp[i] += now_ - stamp_t; stamp_t = now_; }
#else
#define CH_T0()
#define CH_ACC(i)
#endif

// floats of LDS a stage's weights take: fp32 operands, or three bf16 terms (6 bytes per value)
template <bool SPLIT> constexpr int chain_w_floats(int n, int k) { return SPLIT ? n * k * 3 / 2 : n * k; }

template <int K0, int N1, int N2, int N3, int MODE, bool SPLIT = false>
__global__ __launch_bounds__(chain_threads(N1, SPLIT, K0), 1) void chain_kernel(const ChainArgs a) {
    constexpr int THREADS = chain_threads(N1, SPLIT, K0), WAVES = THREADS / 64;
    static_assert(!SPLIT || K0 % 16 == 0, "SPLIT: the first stage reads whole 16-channel K steps");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* w1 = lds;
    float* w2 = w1 + chain_w_floats<SPLIT>(N1, K0);
    float* w3 = w2 + chain_w_floats<SPLIT>(N2, N1);
    float* b1 = w3 + chain_w_floats<SPLIT>(N3, N2);
    float* b2 = b1 + N1;
    float* b3 = b2 + N2;
    float* gam = b3 + (N3 > 0 ? N3 : 0);
    float* bet = gam + K0;
    float* vst = bet + K0;                                // [WAVES][K0]: this wave's per-sample vector

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const nd_src& s = a.d.src;
    const int cin = s.c0 + s.c1;

    // ---- weights, biases, gamma/beta -> LDS (once per persistent workgroup)
    {
        auto copy4 = [&](float* dst, const float* src, int n) {
            for (int i = tid * 4; i < n; i += THREADS * 4) nd_st4(dst + i, nd_ld4(src + i));
        };
        auto copy_pad = [&](float* dst, const float* src, int n, int npad, float fill) {
            for (int i = tid; i < npad; i += THREADS) dst[i] = (src && i < n) ? src[i] : fill;
        };
        copy4(w1, a.d.st[0].weight, chain_w_floats<SPLIT>(N1, K0));
        copy4(w2, a.d.st[1].weight, chain_w_floats<SPLIT>(N2, N1));
        copy_pad(b1, a.d.st[0].bias, a.d.st[0].cout, N1, 0.0f);
        copy_pad(b2, a.d.st[1].bias, a.d.st[1].cout, N2, 0.0f);
        if (N3 > 0) {
            copy4(w3, a.d.st[2].weight, chain_w_floats<SPLIT>(N3, N2));
            copy_pad(b3, a.d.st[2].bias, a.d.st[2].cout, N3, 0.0f);
        }
        if (MODE == ND_PRO_LAYERNORM) {
            copy_pad(gam, s.gamma, cin, K0, 0.0f);        // zero gamma/beta: padded channels stay exactly 0
            copy_pad(bet, s.beta, cin, K0, 0.0f);
        }
    }
    __syncthreads();

    // ---- contiguous range of 32-pixel tiles for this wave (a sample's vector is reloaded only when b changes)
    const int n_waves = gridDim.x * WAVES, wid = blockIdx.x * WAVES + wave;
    const int t_begin = (int)((long)wid * a.n_tiles / n_waves), t_end = (int)((long)(wid + 1) * a.n_tiles / n_waves);
    float* myv = vst + wave * K0;
    int b_cur = -1;
    const int cout_last = a.d.st[N3 > 0 ? 2 : 1].cout;

    // SPLIT: every global access of the tile loop is UNCONDITIONAL -- buffer loads / stores whose lane offset is out of range where the lane has nothing to
    // read or write (a channel quad of the other source or of the padding: the load returns 0, the store is dropped).  With lane-dependent branches around
    // them hipcc cannot count the operations in flight at the joins and falls back to s_waitcnt vmcnt(0): the wait for this tile's rows (requested one tile
    // ago) would also wait for the prefetch just issued and for the previous tile's stores.  Two sources only where the first stage is one K step (cin <= 16).
    constexpr bool TWO = SPLIT && K0 == 16;
    constexpr unsigned OOB = 0xFFFFFFF0u;
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(s.p0), 0, (int)((unsigned)a.d.B * a.d.HW * s.ld0 * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(s.p1 ? s.p1 : s.p0), 0,
                                                                          s.p1 ? (int)((unsigned)a.d.B * a.d.HW * s.ld1 * 4u) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(a.d.out, 0, (int)((unsigned)a.d.B * a.d.HW * a.d.ldo * 4u), 0x00020000);
    f32x4 Xn[SPLIT ? K0 / 8 : 1], Xm[TWO ? K0 / 8 : 1];     // SPLIT: the next tile's rows, in flight over this tile's stages
    auto prefetch_rows = [&](int t_) {
        const unsigned pix_ = (unsigned)t_ * 32u + (unsigned)(lane & 31);
#pragma unroll
        for (int q = 0; q < K0 / 8; ++q) {
            const int c = 8 * q + 4 * half;
            Xn[SPLIT ? q : 0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0, c < s.c0 ? (pix_ * (unsigned)s.ld0 + c) * 4u : OOB, 0, 0));
            if constexpr (TWO)
                Xm[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, (c >= s.c0 && c < cin) ? (pix_ * (unsigned)s.ld1 + (c - s.c0)) * 4u : OOB, 0, 0));
        }
    };
    if constexpr (SPLIT) {
        if (t_begin < t_end) prefetch_rows(t_begin);
    }

#ifdef CHAIN_STAMP
    unsigned long long stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_t = 0;
    const unsigned long long stamp_c0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int t = t_begin; t < t_end; ++t) {
        CH_T0();
        const int b = t / a.tiles_per_sample;             // wave-uniform
        const size_t pix = (size_t)t * 32 + (lane & 31);
        if (s.vec && b != b_cur) {                         // stash vec[b] for this wave
            for (int c = lane; c < K0; c += 64) myv[c] = c < cin ? s.vec[(size_t)b * cin + c] : 0.0f;
            b_cur = b;
        }
        // ---- this pixel's input row, channels 8q + 4*half .. +3 (two sources = virtual concat)
        f32x4 X[K0 / 8];
        if constexpr (SPLIT) {
#pragma unroll
            for (int q = 0; q < K0 / 8; ++q) {
                X[q] = Xn[q];
                if constexpr (TWO) X[q] += Xm[q];
            }
            prefetch_rows(t + 1 < t_end ? t + 1 : t);      // (behind the last tile: a harmless reload of it)
            if (s.vec) {
#pragma unroll
                for (int q = 0; q < K0 / 8; ++q) X[q] += nd_ld4(myv + 8 * q + 4 * half);
            }
        } else {
#pragma unroll
            for (int q = 0; q < K0 / 8; ++q) {
                const int c = 8 * q + 4 * half;
                const f32x4 zero = {0, 0, 0, 0};
                X[q] = zero;
                if (c < s.c0) X[q] = nd_ld4(s.p0 + pix * s.ld0 + c);
                else if (c < cin) X[q] = nd_ld4(s.p1 + pix * s.ld1 + (c - s.c0));
                if (s.vec) X[q] += nd_ld4(myv + c);
            }
        }
        // ---- prologue -> operands of stage 0
        float in0[K0 / 2];
        if (MODE == ND_PRO_LAYERNORM) {                    // nn.LayerNorm over the cin channels of (x + v): biased variance, eps 1e-5
            float sum = 0.0f;
#pragma unroll
            for (int q = 0; q < K0 / 8; ++q) sum += (X[q][0] + X[q][1]) + (X[q][2] + X[q][3]);
            sum += __shfl_xor(sum, 32);
            const float mean = sum / (float)cin;
            float m2 = 0.0f;
#pragma unroll
            for (int q = 0; q < K0 / 8; ++q) {
                if (8 * q + 4 * half < cin) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const float dv = X[q][i] - mean; m2 = fmaf(dv, dv, m2); }
                }
            }
            m2 += __shfl_xor(m2, 32);
            const float rstd = rsqrtf(m2 / (float)cin + 1e-5f);
#pragma unroll
            for (int q = 0; q < K0 / 8; ++q) {
                const f32x4 g4 = nd_ld4(gam + 8 * q + 4 * half), be4 = nd_ld4(bet + 8 * q + 4 * half);
#pragma unroll
                for (int i = 0; i < 4; ++i) in0[4 * q + i] = (X[q][i] - mean) * rstd * g4[i] + be4[i];
            }
        } else {
#pragma unroll
            for (int q = 0; q < K0 / 8; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) in0[4 * q + i] = X[q][i];
        }

        CH_ACC(0);                                         // rows + vector + prologue
        // ---- stage 0
        f32x16 H1[N1 / 32];
        if constexpr (SPLIT) chain_gemm_split<K0, N1>(in0, reinterpret_cast<const char*>(w1), b1, H1, lane);
        else chain_gemm<K0, N1>(in0, w1, b1, H1, lane);
        CH_ACC(1);
        chain_res<K0, N1>(H1, X, s.vec ? myv : nullptr, a.d.st[0].res, half);
        chain_act<N1>(H1, a.d.st[0].act);
        CH_ACC(2);
        // ---- stage 1: accumulator register r of n-tile nt is operand 16*nt + r
        float in1[N1 / 2];
#pragma unroll
        for (int nt = 0; nt < N1 / 32; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) in1[16 * nt + r] = H1[nt][r];
        f32x16 H2[N2 / 32];
        if constexpr (SPLIT) chain_gemm_split<N1, N2>(in1, reinterpret_cast<const char*>(w2), b2, H2, lane);
        else chain_gemm<N1, N2>(in1, w2, b2, H2, lane);
        CH_ACC(3);
        chain_res<K0, N2>(H2, X, s.vec ? myv : nullptr, a.d.st[1].res, half);
        chain_act<N2>(H2, a.d.st[1].act);
        CH_ACC(4);
        float* row = a.d.out + pix * a.d.ldo;
        if (N3 > 0) {
            float in2[N2 / 2];
#pragma unroll
            for (int nt = 0; nt < N2 / 32; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) in2[16 * nt + r] = H2[nt][r];
            f32x16 H3[(N3 > 0 ? N3 : 32) / 32];
            if constexpr (SPLIT) chain_gemm_split<N2, (N3 > 0 ? N3 : 32)>(in2, reinterpret_cast<const char*>(w3), b3, H3, lane);
            else chain_gemm<N2, (N3 > 0 ? N3 : 32)>(in2, w3, b3, H3, lane);
            CH_ACC(5);
            chain_res<K0, (N3 > 0 ? N3 : 32)>(H3, X, s.vec ? myv : nullptr, a.d.st[2].res, half);
            chain_act<(N3 > 0 ? N3 : 32)>(H3, a.d.st[2].act);
            if constexpr (SPLIT) chain_store_buf<(N3 > 0 ? N3 : 32)>(H3, rso, (unsigned)pix * (unsigned)a.d.ldo * 4u, cout_last, half);
            else chain_store<(N3 > 0 ? N3 : 32)>(H3, row, cout_last, half);
        } else {
            if constexpr (SPLIT) chain_store_buf<N2>(H2, rso, (unsigned)pix * (unsigned)a.d.ldo * 4u, cout_last, half);
            else chain_store<N2>(H2, row, cout_last, half);
        }
        CH_ACC(6);
    }
#ifdef CHAIN_STAMP
    if (lane == 0 && wid < 4096 * 16) {
        unsigned long long* o = chain_dbg + (size_t)wid * 10;
        for (int i = 0; i < 7; ++i) o[i] = stamp[i];
        o[7] = __builtin_amdgcn_s_memtime() - stamp_c0;
        o[8] = __builtin_amdgcn_s_memrealtime() - stamp_r0;
        o[9] = (unsigned long long)(t_end - t_begin);
    }
#endif
}

// (cout, cin) row-major -> operand order [nt][jq][lane][4]: element i of lane l = W[32*nt + (l & 31)][8*jq + i + 4*(l >> 5)]
__global__ void pack_chain_kernel(const float* __restrict__ w, float* __restrict__ out, int cin, int cout, int KP, int NP) {
    const int total = KP * NP;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int i = idx & 3, l = (idx >> 2) & 63, rest = idx >> 8;
        const int jq = rest % (KP / 8), nt = rest / (KP / 8);
        const int n = 32 * nt + (l & 31), k = 8 * jq + i + 4 * (l >> 5);
        out[idx] = (n < cout && k < cin) ? w[(size_t)n * cin + k] : 0.0f;
    }
}

// the same operand order as three bf16 terms: [nt][K step s][term][lane][8], slot i of lane l = W[32*nt + (l & 31)][16*s + 8*(i >> 2) + 4*(l >> 5) + (i & 3)]
__global__ void pack_chain_split_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int cin, int cout, int KP, int NP) {
    const int total = KP * NP;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int i = idx & 7, l = (idx >> 3) & 63, rest = idx >> 9;
        const int s = rest % (KP / 16), nt = rest / (KP / 16);
        const int n = 32 * nt + (l & 31), k = 16 * s + 8 * (i >> 2) + 4 * (l >> 5) + (i & 3);
        const float v = (n < cout && k < cin) ? w[(size_t)n * cin + k] : 0.0f;
        const __bf16 t1 = (__bf16)v;
        const float r1 = v - (float)t1;                       // exact
        const __bf16 t2 = (__bf16)r1;
        const float r2 = r1 - (float)t2;                      // exact, at most 8 significant bits: its upper half is the third term
        unsigned short* o = out + ((size_t)(nt * (KP / 16) + s) * 3 * 64 + l) * 8 + i;
        o[0] = __builtin_bit_cast(unsigned short, t1);
        o[512] = __builtin_bit_cast(unsigned short, t2);
        o[1024] = (unsigned short)(__builtin_bit_cast(unsigned, r2) >> 16);
    }
}

static inline int device_cus() { return nd_device_cus(); }

template <int K0, int N1, int N2, int N3, int MODE, bool SPLIT = false>
int launch(const ChainArgs& a, hipStream_t st) {
    static nd_device_once configured;
    constexpr int THREADS = chain_threads(N1, SPLIT, K0), WAVES = THREADS / 64;
    const size_t lds = (size_t)(chain_w_floats<SPLIT>(N1, K0) + chain_w_floats<SPLIT>(N2, N1) + chain_w_floats<SPLIT>(N3, N2) + N1 + N2 + N3 + 2 * K0 + WAVES * K0) * sizeof(float);
    static_assert((size_t)(chain_w_floats<SPLIT>(N1, K0) + chain_w_floats<SPLIT>(N2, N1) + chain_w_floats<SPLIT>(N3, N2) + N1 + N2 + N3 + 2 * K0 + 16 * K0) * 4 <= 160 * 1024, "LDS");
    if (int e = nd_reserve_lds(configured, reinterpret_cast<const void*>(chain_kernel<K0, N1, N2, N3, MODE, SPLIT>), lds, "nd_pointwise_chain")) return e;
    const int wgs = nd_cdiv(a.n_tiles, WAVES);
    const int grid = wgs < device_cus() ? wgs : device_cus();
    hipLaunchKernelGGL((chain_kernel<K0, N1, N2, N3, MODE, SPLIT>), dim3(grid), dim3(THREADS), lds, st, a);
    return 0;
}

template <int K0, int N1, int N2, int N3, bool SPLIT = false>
int launch_mode(const ChainArgs& a, hipStream_t st) {
    if (a.d.src.mode == ND_PRO_LAYERNORM) return launch<K0, N1, N2, N3, ND_PRO_LAYERNORM, SPLIT>(a, st);
    return launch<K0, N1, N2, N3, ND_PRO_NONE, SPLIT>(a, st);
}

}  // namespace

extern "C" int64_t nd_pack_chain_weight_floats(int cin, int cout, int first_stage) {
    return (int64_t)nd_round_up(cin, first_stage ? 8 : 32) * nd_round_up(cout, 32);
}

extern "C" int nd_pack_chain_weight(const float* w, float* packed, int cin, int cout, int first_stage, void* stream) {
    ND_REQUIRE(w && packed, ND_E_BADARG, "nd_pack_chain_weight: null pointer");
    ND_REQUIRE(cin > 0 && cout > 0, ND_E_SHAPE, "nd_pack_chain_weight: non-positive size");
    const int KP = nd_round_up(cin, first_stage ? 8 : 32), NP = nd_round_up(cout, 32);   // later stages read whole 32-wide n-tiles
    const int total = KP * NP;
    hipLaunchKernelGGL(pack_chain_kernel, dim3(nd_cdiv(total, 256) < 1024 ? nd_cdiv(total, 256) : 1024), dim3(256), 0, (hipStream_t)stream,
                       w, packed, cin, cout, KP, NP);
    return nd_launch_status("nd_pack_chain_weight");
}

// the (K0, N1, N2, N3) shapes this build instantiates: NoiseDiffNet's Mlp / AttnBlock chains at dim 16, 32, 48, 64
extern "C" int nd_pointwise_chain_supported(int cin, int n1, int n2, int n3) {
    const int K0 = nd_round_up(cin, 8), N1 = nd_round_up(n1, 32), N2 = nd_round_up(n2, 32), N3 = n3 > 0 ? nd_round_up(n3, 32) : 0;
    static const int table[][4] = {{8, 32, 32, 0},  {16, 32, 32, 0}, {16, 32, 32, 32}, {32, 32, 32, 0}, {32, 64, 32, 32},
                                   {8, 64, 64, 0},  {48, 64, 64, 0}, {48, 96, 64, 64}, {64, 64, 64, 0}, {64, 128, 64, 64},
                                   {64, 64, 32, 0}, {32, 32, 32, 32}, {48, 64, 32, 0}};
    for (const auto& e : table)
        if (e[0] == K0 && e[1] == N1 && e[2] == N2 && e[3] == N3) return 1;
    return 0;
}

extern "C" int64_t nd_pack_chain_weight_split_floats(int cin, int cout, int first_stage) {
    return (int64_t)nd_round_up(cin, first_stage ? 16 : 32) * nd_round_up(cout, 32) * 3 / 2;      // three bf16 terms per value, counted in floats (the arena's unit)
}

extern "C" int nd_pack_chain_weight_split(const float* w, float* packed, int cin, int cout, int first_stage, void* stream) {
    ND_REQUIRE(w && packed && nd_aligned16(packed), ND_E_BADARG, "nd_pack_chain_weight_split: null or unaligned pointer");
    ND_REQUIRE(cin > 0 && cout > 0, ND_E_SHAPE, "nd_pack_chain_weight_split: non-positive size");
    const int KP = nd_round_up(cin, first_stage ? 16 : 32), NP = nd_round_up(cout, 32);
    const int total = KP * NP;
    hipLaunchKernelGGL(pack_chain_split_kernel, dim3(nd_cdiv(total, 256) < 1024 ? nd_cdiv(total, 256) : 1024), dim3(256), 0, (hipStream_t)stream,
                       w, reinterpret_cast<unsigned short*>(packed), cin, cout, KP, NP);
    return nd_launch_status("nd_pack_chain_weight_split");
}

static int chain_run(const nd_chain* d, void* stream, bool split) {
    ND_REQUIRE(d, ND_E_BADARG, "nd_pointwise_chain: null descriptor");
    const nd_src& s = d->src;
    ND_REQUIRE(d->n_stages == 2 || d->n_stages == 3, ND_E_BADARG, "nd_pointwise_chain: n_stages=%d (2 or 3)", d->n_stages);
    ND_REQUIRE(s.p0 && d->out, ND_E_BADARG, "nd_pointwise_chain: null tensor pointer");
    ND_REQUIRE(d->B > 0 && d->HW > 0 && d->HW % 32 == 0, ND_E_SHAPE, "nd_pointwise_chain: HW=%d must be a positive multiple of 32", d->HW);
    ND_REQUIRE(s.c0 > 0 && s.c0 % 4 == 0 && s.c1 % 4 == 0 && (s.c1 == 0) == (s.p1 == nullptr), ND_E_SHAPE,
               "nd_pointwise_chain: source channels %d+%d (multiples of 4; p1 iff c1)", s.c0, s.c1);
    const int cin = s.c0 + s.c1;
    ND_REQUIRE(s.ld0 >= s.c0 && s.ld0 % 4 == 0 && (s.c1 == 0 || (s.ld1 >= s.c1 && s.ld1 % 4 == 0)), ND_E_ALIGN,
               "nd_pointwise_chain: pixel strides must be >= channels and multiples of 4");
    ND_REQUIRE(s.mode == ND_PRO_NONE || s.mode == ND_PRO_LAYERNORM, ND_E_BADARG, "nd_pointwise_chain: unsupported prologue %d", s.mode);
    ND_REQUIRE(s.mode != ND_PRO_LAYERNORM || (s.gamma && s.beta), ND_E_BADARG, "nd_pointwise_chain: LayerNorm needs gamma and beta");
    ND_REQUIRE(!s.upsample && !s.unshuffle && !s.rowstats, ND_E_BADARG, "nd_pointwise_chain: plain pixel addressing only");
    int width = cin;
    for (int i = 0; i < d->n_stages; ++i) {
        const nd_chain_stage& g = d->st[i];
        ND_REQUIRE(g.weight && g.cin == width && g.cout > 0, ND_E_SHAPE, "nd_pointwise_chain: stage %d is %d->%d, expected input width %d", i,
                   g.cin, g.cout, width);
        ND_REQUIRE(g.res == ND_CHAIN_RES_NONE || g.cout == cin, ND_E_SHAPE, "nd_pointwise_chain: stage %d residual needs cout == input width", i);
        ND_REQUIRE(nd_aligned16(g.weight), ND_E_ALIGN, "nd_pointwise_chain: weights must be 16-byte aligned");
        width = g.cout;
    }
    ND_REQUIRE(width % 4 == 0 && d->ldo >= width && d->ldo % 4 == 0, ND_E_SHAPE, "nd_pointwise_chain: output width %d, ldo %d", width, d->ldo);
    ND_REQUIRE(nd_aligned16(s.p0) && nd_aligned16(s.p1) && nd_aligned16(d->out), ND_E_ALIGN, "nd_pointwise_chain: tensors must be 16-byte aligned");
    const int n3 = d->n_stages == 3 ? d->st[2].cout : 0;
    ND_REQUIRE(nd_pointwise_chain_supported(cin, d->st[0].cout, d->st[1].cout, n3), ND_E_SHAPE,
               "nd_pointwise_chain: widths %d->%d->%d->%d are not instantiated (use nd_pointwise_gemm_nhwc_f32 per layer)", cin,
               d->st[0].cout, d->st[1].cout, n3);
    ChainArgs a;
    a.d = *d;
    a.tiles_per_sample = d->HW / 32;
    const long tiles = (long)d->B * a.tiles_per_sample;
    ND_REQUIRE(tiles < (1L << 31), ND_E_SHAPE, "nd_pointwise_chain: too many pixels");
    a.n_tiles = (int)tiles;
    hipStream_t st = (hipStream_t)stream;
    const int K0 = nd_round_up(cin, split ? 16 : 8), N1 = nd_round_up(d->st[0].cout, 32), N2 = nd_round_up(d->st[1].cout, 32), N3 = n3 ? nd_round_up(n3, 32) : 0;
    int rc = ND_E_SHAPE;
    if (split) {          // the same widths with the first stage's channels in whole 16-channel K steps
        ND_REQUIRE(s.c1 == 0 || K0 == 16, ND_E_SHAPE, "nd_pointwise_chain_split: two sources only with cin <= 16 (got %d + %d)", s.c0, s.c1);
        ND_REQUIRE((long)d->B * d->HW * s.ld0 * 4 < (1L << 32) - 65536 && (long)d->B * d->HW * (s.c1 ? s.ld1 : 0) * 4 < (1L << 32) - 65536 &&
                   (long)d->B * d->HW * d->ldo * 4 < (1L << 32) - 65536, ND_E_SHAPE, "nd_pointwise_chain_split: a tensor of 4 GiB or more");
#define ND_CHAIN_SPLIT_CASE(k0, n1, n2, n3) \
        if (K0 == k0 && N1 == n1 && N2 == n2 && N3 == n3) rc = launch_mode<k0, n1, n2, n3, true>(a, st);
        ND_CHAIN_SPLIT_CASE(16, 32, 32, 0)
        ND_CHAIN_SPLIT_CASE(16, 32, 32, 32)
        ND_CHAIN_SPLIT_CASE(32, 32, 32, 0)
        ND_CHAIN_SPLIT_CASE(32, 32, 32, 32)
        ND_CHAIN_SPLIT_CASE(32, 64, 32, 32)
        ND_CHAIN_SPLIT_CASE(16, 64, 64, 0)
        ND_CHAIN_SPLIT_CASE(48, 64, 64, 0)
        ND_CHAIN_SPLIT_CASE(48, 96, 64, 64)
        ND_CHAIN_SPLIT_CASE(64, 64, 64, 0)
        ND_CHAIN_SPLIT_CASE(64, 64, 32, 0)
        ND_CHAIN_SPLIT_CASE(48, 64, 32, 0)
        ND_CHAIN_SPLIT_CASE(64, 128, 64, 64)
#undef ND_CHAIN_SPLIT_CASE
        if (rc) return rc;
        return nd_launch_status("nd_pointwise_chain_split_nhwc_f32");
    }
#define ND_CHAIN_CASE(k0, n1, n2, n3) \
    if (K0 == k0 && N1 == n1 && N2 == n2 && N3 == n3) rc = launch_mode<k0, n1, n2, n3>(a, st);
    ND_CHAIN_CASE(8, 32, 32, 0)
    ND_CHAIN_CASE(16, 32, 32, 0)
    ND_CHAIN_CASE(16, 32, 32, 32)
    ND_CHAIN_CASE(32, 32, 32, 0)
    ND_CHAIN_CASE(32, 32, 32, 32)
    ND_CHAIN_CASE(32, 64, 32, 32)
    ND_CHAIN_CASE(8, 64, 64, 0)
    ND_CHAIN_CASE(48, 64, 64, 0)
    ND_CHAIN_CASE(48, 96, 64, 64)
    ND_CHAIN_CASE(64, 64, 64, 0)
    ND_CHAIN_CASE(64, 64, 32, 0)
    ND_CHAIN_CASE(48, 64, 32, 0)
    ND_CHAIN_CASE(64, 128, 64, 64)
#undef ND_CHAIN_CASE
    if (rc) return rc;
    return nd_launch_status("nd_pointwise_chain_nhwc_f32");
}

#ifdef CHAIN_STAMP
extern "C" int nd_chain_debug_read(unsigned long long* host, int n_words) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(chain_dbg), (size_t)n_words * 8);
}
#endif

extern "C" int nd_pointwise_chain_nhwc_f32(const nd_chain* d, void* stream) { return chain_run(d, stream, false); }

// The same chain with the products on the bf16 matrix pipe at full fp32 significand (three-term split, six products: see the head of this file);
// the stages' `weight` pointers are nd_pack_chain_weight_split packings.  Same widths, prologues, residuals, activations and errors.
extern "C" int nd_pointwise_chain_split_nhwc_f32(const nd_chain* d, void* stream) { return chain_run(d, stream, true); }
