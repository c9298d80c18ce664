// conv3x3_wino4.hip -- Winograd F(4x4,3x3) 3x3 convolution on v_mfma_f32_16x16x4_f32 (EXPERIMENTAL, r1e).
//
// F(4x4,3x3) needs 36 multiplies per 16 outputs (2.25 per output) against 4 for F(2x2,3x3): 1.78x fewer MFMAs than
// conv3x3_wino2.hip.  Its 36 position accumulators do not fit a 32x32 MFMA tile (36 x 16 registers), so the product is
// laid out on the 16x16x4 instruction instead: one wave = 16 tiles of 4x4 pixels (the workgroup's 16x16 pixels) x 16
// output channels = 36 x 4 accumulator registers; the four waves of a workgroup take four cout groups (64 couts, the
// workgroup tile of wino2).  Operand roles are swapped (A = transformed weights U, B = transformed input V) so that a
// lane ends up with FOUR CONSECUTIVE COUTS of one tile: the output transform runs on float4s and stores 16 bytes.
//
// Transform matrices (Lavin & Gray, interpolation points 0, +-1, +-2, inf):
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
// fp32 rounding of these transforms is ~10x that of F(2x2,3x3) (~1e-5 relative); the sampler's measured full-length
// error with F(2x2,3x3) is 3.8e-6 against a 1e-3 budget.
//
// LDS image of a K chunk (16 channels): [channel pair 8][patch position (a,b) 36][tile 16] float2 -- a halo pixel is
// stored once per (tile, patch position) it belongs to (1.78x duplication), which makes every operand read of the
// transform a conflict-free 128-byte row: lane (tile = l & 15, k = l >> 4) reads the pair of channels (2k, 2k+1) of
// its tile's patch entry (a, b); the two channels feed two MFMAs (k groups {0,2,4,6} and {1,3,5,7} of an 8-channel block)
// and the whole transform runs on packed float2 math.
#include <stdlib.h>
#include <type_traits>
#include "nd_common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int KC4 = 16;                              // channels per K chunk
constexpr int NPOS = 36;                             // positions (xi, nu) == patch entries (a, b)
constexpr int VD_FLOATS = (KC4 / 2) * NPOS * 16 * 2; // 9216 floats = 36 KB

struct Wino4Args {
    nd_conv3x3 d;
    int tiles_x, tiles_y, n_tiles, n_cg, n_c8, slots, total_wg;
};

// one row of B^T applied to six packed values (the same code serves the column pass)
__device__ __forceinline__ void w4_bt(const f32x2 (&d)[6], f32x2 (&t)[6]) {
    const f32x2 p = d[4] - 4.0f * d[2], q = d[3] - 4.0f * d[1];
    const f32x2 r = d[4] - d[2], s = d[3] - d[1];
    t[0] = 4.0f * d[0] - 5.0f * d[2] + d[4];
    t[1] = p + q;
    t[2] = p - q;
    t[3] = r + 2.0f * s;
    t[4] = r - 2.0f * s;
    t[5] = 4.0f * d[1] - 5.0f * d[3] + d[5];
}

// A^T applied to six float4s (four consecutive couts each)
__device__ __forceinline__ void w4_at(const f32x4 (&m)[6], f32x4 (&y)[4]) {
    const f32x4 a = m[1] + m[2], b = m[1] - m[2], c = m[3] + m[4], e = m[3] - m[4];
    y[0] = m[0] + a + c;
    y[1] = b + 2.0f * e;
    y[2] = a + 4.0f * c;
    y[3] = b + 8.0f * e + m[5];
}

// ---- the kernel.  One persistent workgroup per CU (one wave per SIMD, the whole register file):
//   * items = (tile, 16-channel chunk); the halo of item i+1 is loaded from HBM/L2 during item i's first 8-channel
//     stage (9 buffer loads per thread: the thread's tile and channel quad are fixed, the patch entry (a, b) of load `it`
//     is wave-uniform, so the big part of every address sits in the scalar offset and nothing about it is recomputed)
//     and written into the other LDS buffer at the end of that stage; the chunk's one barrier follows, and the second
//     stage already reads the next item's first patch entries behind it;
//   * a stage = 72 MFMAs (36 positions x the channel pair of this lane); the 18 weight fragments are refreshed in place
//     for the NEXT stage right after their MFMAs (a whole stage of latency cover), the next stage's 36 patch entries are
//     read from LDS at the top of the stage and transformed (B^T d B, packed float2) between the MFMA rows, each new
//     operand row replacing the row whose MFMAs have just been issued.
#ifndef W4_ABLATE
#define W4_ABLATE 0          // diagnostic builds only (tools/w4_variants.sh): 1 no staging, 2 no weight refresh, 4 no transform, 8 no epilogue
#endif
#ifndef W4_URING
#define W4_URING 18          // weight fragments in flight (18 = a whole stage ahead; 9 measured 15 % slower: L2 latency shows)
#endif
// MFMAs through inline asm (as in conv3x3_wino2.hip): "+a" keeps the 36 accumulators in place in the accumulator half of the
// register file -- with the builtin, hipcc moved them between AGPR tuples and through VGPRs (s_nop 7 + 4 v_accvgpr_write per
// move) several times per stage; operands are pinned to VGPRs.  The epilogue drains the pipe before reading them.
// `s_nop 1` in front of every MFMA: without it results are wrong in every test shape (identical code otherwise), with it the
// timing does not change -- the MFMA issue slot is not what bounds this kernel.  The hazard it covers is not one of the
// register-proximity cases (no VALU or load writes an operand within 8 instructions of its MFMA); kept until identified.
#define W4_MFMA(acc, av, bv) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(av), "v"(bv))
#define W4_MFMA_DRAIN() asm volatile("s_nop 15\n\ts_nop 15" ::: "memory")
constexpr int W4_STAGE_LOADS = 9;                    // 576 (entry, tile) slots x 4 quads / 256 threads

template <int MODE>
__global__ __launch_bounds__(256, 1) void wino4_kernel(const Wino4Args a) {
    constexpr bool AFF = MODE == ND_PRO_AFFINE_SILU;
    extern __shared__ __attribute__((aligned(16))) float Vd[];          // [2][VD_FLOATS]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = lane & 15, kq = lane >> 4;

    const int t_begin = (int)((long)blockIdx.x * a.total_wg / gridDim.x), t_end = (int)((long)(blockIdx.x + 1) * a.total_wg / gridDim.x);
    if (t_begin >= t_end) return;

    const nd_src& s = a.d.src;
    const int H = a.d.H, W = a.d.W, Cin = a.d.cin, Cout = a.d.cout;
    const int Ctot = s.c0 + s.c1;
    const int n_chunks = (Cin + KC4 - 1) / KC4;

    auto decode = [&](int t, int& b_, int& ty_, int& tx_, int& nt_) {
        int lid = t;
        nt_ = lid % a.n_tiles;  lid /= a.n_tiles;
        tx_ = lid % a.tiles_x;  lid /= a.tiles_x;
        ty_ = lid % a.tiles_y;
        b_ = lid / a.tiles_y;
    };

    // ---- staging: thread = (tile st, channel quad sq), load `it` = patch entry e = wave + 4 * it of that tile
    const int st = tid & 15, sq = (tid >> 4) & 3;                  // == (lane & 15, lane >> 4): a wave's 64 lanes fill one 1-KB patch entry
    const int sty = 4 * (st >> 2), stx = 4 * (st & 3);          // tile origin inside the 16x16 pixels (patch entry (0,0) is one up-left)
    const long npx = (long)a.d.B * H * W;
    const __amdgpu_buffer_rsrc_t rsrc0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(s.p0), 0, (int)(npx * s.ld0 * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(s.p1 ? s.p1 : s.p0), 0, (int)(npx * (s.p1 ? s.ld1 : s.ld0) * 4), 0x00020000);
    const unsigned OOB = 0x7FFFFFF0u;                            // lane offset beyond any tensor: the load returns zeros (padding)
    // LDS byte address of this thread's first write: pair 2*sq, entry `wave`, tile st
    const unsigned st_lds = (unsigned)(wave * 1024 + lane * 16);           // entry `wave`, then + 4 KB per load

    f32x4 raw[AFF ? W4_STAGE_LOADS : 1];
    f32x4 tM = {0, 0, 0, 0}, tA = {1, 1, 1, 1}, tD = {0, 0, 0, 0};
    // per staged tile: pixel offset of each of the 9 patch entries (lane part relative to a wave-uniform scalar pixel) and validity
    int sbase[W4_STAGE_LOADS];
    int* poff = reinterpret_cast<int*>(Vd + 2 * VD_FLOATS) + tid;          // [9][256], thread-private column
    unsigned tilemask = 0, okmask = 0;
    int sb_ = 0;
    auto stage_tile = [&](int b_, int ty_, int tx_) {
        sb_ = b_;
        const int y0 = ty_ * 16 - 1, x0 = tx_ * 16 - 1;
        const bool interior = ty_ > 0 && tx_ > 0 && y0 + 18 <= H && x0 + 18 <= W;      // wave-uniform
        tilemask = 0;
#pragma unroll
        for (int it = 0; it < W4_STAGE_LOADS; ++it) {
            const int e = wave + 4 * it;                         // wave-uniform patch entry
            const int ay = (e * 43) >> 8, ax = e - 6 * ay;       // e / 6, e % 6 for e < 36
            if (interior) {
                sbase[it] = __builtin_amdgcn_readfirstlane((b_ * H + y0 + ay) * W + x0 + ax);
                poff[it * 256] = sty * W + stx;
                tilemask |= 1u << it;
            } else {
                const int gy = y0 + sty + ay, gx = x0 + stx + ax;
                const bool ok = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
                const int sy = min(max(y0 + ay, 0), H - 1), sx = min(max(x0 + ax, 0), W - 1);
                sbase[it] = __builtin_amdgcn_readfirstlane((b_ * H + sy) * W + sx);
                poff[it * 256] = ok ? (gy - sy) * W + (gx - sx) : 0;
                tilemask |= (ok ? 1u : 0u) << it;
            }
        }
    };
    auto stage_issue = [&](int cb_, float* dma_dst) {
        const bool sec = cb_ >= s.c0;                            // wave-uniform: a chunk never straddles the sources (host check)
        const __amdgpu_buffer_rsrc_t rs = sec ? rsrc1 : rsrc0;
        const int ld = sec ? s.ld1 : s.ld0;
        const int cbase = sec ? cb_ - s.c0 : cb_;
        const int c = cb_ + 4 * sq;
        const bool cvalid = c < Cin;
        if (AFF) {
            const float* m = s.mad + (size_t)sb_ * 3 * Ctot + (cvalid ? c : 0);
            tM = nd_ld4(m); tA = nd_ld4(m + Ctot); tD = nd_ld4(m + 2 * Ctot);
            tD = tD - tM * tA;
        }
        okmask = cvalid ? tilemask : 0u;
        const unsigned ld4 = (unsigned)ld * 4u;
#pragma unroll
        for (int it = 0; it < W4_STAGE_LOADS; ++it) {
            const int soff = __builtin_amdgcn_readfirstlane((sbase[it] * ld + cbase) * 4);
            const unsigned voff = ((okmask >> it) & 1u) ? (unsigned)poff[it * 256] * ld4 + 16u * sq : OOB;
#if !(W4_ABLATE & 1)
            if (AFF) raw[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
            else     // LDS-DMA: 64 lanes x 16 bytes land as one contiguous patch entry, out-of-range lanes as zeros; no registers
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dma_dst + (wave + 4 * it) * 256),
                                                         16, voff, soff, 0, 0);
#endif
        }
    };
    auto stage_commit = [&](float* dst) {
        if (!AFF) return;                                        // plain inputs went straight to LDS (LDS-DMA), nothing to commit
        char* base = reinterpret_cast<char*>(dst) + st_lds;
#pragma unroll
        for (int it = 0; it < W4_STAGE_LOADS; ++it) {
            f32x4 v = nd_silu4(raw[it] * tA + tD);
            const f32x4 zero = {0, 0, 0, 0};
            v = ((okmask >> it) & 1u) ? v : zero;                // silu(affine(0)) != 0: padding is applied after the activation
            *reinterpret_cast<f32x4*>(base + it * 4096) = v;
        }
    };

    // ---- MFMA side
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.d.weight), 0, (int)((long)a.n_c8 * a.n_cg * 18 * 256 * 4), 0x00020000);
    const unsigned wvoff = (unsigned)(lane * 16);
    auto wblock = [&](int c8_, int cg_) { return __builtin_amdgcn_readfirstlane(((c8_ * a.n_cg + cg_) * 18) * 1024); };
    const unsigned d_lds = (unsigned)((kq >> 1) * 256 + tile * 16 + (kq & 1) * 8);   // + stage * 512, + entry * 1024

    f32x4 acc[NPOS];
#pragma unroll
    for (int p = 0; p < NPOS; ++p) acc[p] = f32x4{0, 0, 0, 0};
    f32x4 U[W4_URING];                                          // ring: fragment pp lives in U[pp % W4_URING], refreshed that far ahead
    f32x2 V[6][6], T[6][6];

    auto load_u = [&](int pp, int wb) { U[pp % W4_URING] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, wb + pp * 1024, 0)); };
    auto read_d = [&](const float* buf, int g2) {                // next stage's patch entries -> T (raw values for now)
        const char* base = reinterpret_cast<const char*>(buf) + d_lds + g2 * 512;
#pragma unroll
        for (int e = 0; e < NPOS; ++e) T[e / 6][e % 6] = *reinterpret_cast<const f32x2*>(base + e * 1024);
    };
    auto col_pass = [&]() {                                      // T <- B^T T (over the patch rows, every column)
#pragma unroll
        for (int bx = 0; bx < 6; ++bx) {
            f32x2 col[6], t[6];
#pragma unroll
            for (int ay = 0; ay < 6; ++ay) col[ay] = T[ay][bx];
            w4_bt(col, t);
#pragma unroll
            for (int xi = 0; xi < 6; ++xi) T[xi][bx] = t[xi];
        }
    };
    auto mfma_row = [&](int xi, int wb_cur, int wb_next) {       // 12 MFMAs of operand row xi; the ring refilled behind them
        // first channel of the pair for all six positions, then the second: an accumulator is touched again six MFMAs later
        // (the asm hides the instruction from hipcc's hazard recognizer, so the dependent-accumulate distance is kept long)
#pragma unroll
        for (int h = 0; h < 3; ++h) {
            const int pp = xi * 3 + h;
            const f32x4 u = U[pp % W4_URING];
            W4_MFMA(acc[2 * pp], u.x, V[xi][2 * h].x);
            W4_MFMA(acc[2 * pp + 1], u.z, V[xi][2 * h + 1].x);
        }
#pragma unroll
        for (int h = 0; h < 3; ++h) {
            const int pp = xi * 3 + h;
            const f32x4 u = U[pp % W4_URING];
            W4_MFMA(acc[2 * pp], u.y, V[xi][2 * h].y);
            W4_MFMA(acc[2 * pp + 1], u.w, V[xi][2 * h + 1].y);
#if !(W4_ABLATE & 2)
            if (pp + W4_URING < 18) load_u(pp + W4_URING, wb_cur);
            else load_u(pp + W4_URING - 18, wb_next);
#endif
        }
    };
    // one 8-channel stage: MFMAs with (U, V) of this stage, operands of the next stage produced on the way
    auto stage = [&](const float* next_buf, int next_g2, int wb_cur, int wb_next, auto&& mid) {
        read_d(next_buf, next_g2);
        __builtin_amdgcn_sched_barrier(0);
        mfma_row(0, wb_cur, wb_next);
        mfma_row(1, wb_cur, wb_next);
        __builtin_amdgcn_sched_barrier(0);
#if !(W4_ABLATE & 4)
        col_pass();
#endif
#pragma unroll
        for (int xi = 2; xi < 6; ++xi) {
            mfma_row(xi, wb_cur, wb_next);
#if !(W4_ABLATE & 4)
            w4_bt(T[xi - 2], V[xi - 2]);                         // rows whose MFMAs are issued take their next values
#else
            for (int q = 0; q < 6; ++q) V[xi - 2][q] = T[xi - 2][q];
#endif
        }
#if !(W4_ABLATE & 4)
        w4_bt(T[4], V[4]);
        w4_bt(T[5], V[5]);
#else
        for (int q = 0; q < 6; ++q) { V[4][q] = T[4][q]; V[5][q] = T[5][q]; }
#endif
        __builtin_amdgcn_sched_barrier(0);
        mid();
    };

    // ---- prologue: first item staged synchronously, first operands built
    int b, ty, tx, nt;
    decode(t_begin, b, ty, tx, nt);
    int cur = 0;
    stage_tile(b, ty, tx);
    stage_issue(0, Vd);
    stage_commit(Vd);
    __syncthreads();                                             // (plain inputs: the fence in front of it waits for the LDS-DMA)
    if (!AFF) stage_issue(KC4, Vd + VD_FLOATS);                  // item 1 (n_chunks >= 2, host check) is on its way before chunk 0 starts
    {
        const int wb0 = wblock(0, nt * 4 + wave);
#pragma unroll
        for (int pp = 0; pp < W4_URING; ++pp) load_u(pp, wb0);
        read_d(Vd, 0);
        col_pass();
#pragma unroll
        for (int xi = 0; xi < 6; ++xi) w4_bt(T[xi], V[xi]);
    }

    for (int t = t_begin; t < t_end; ++t) {
        int b1 = b, ty1 = ty, tx1 = tx, nt1 = nt;
        const bool more = t + 1 < t_end;
        if (more) decode(t + 1, b1, ty1, tx1, nt1);
        const int cg = nt * 4 + wave, cg1 = nt1 * 4 + wave;
        for (int ch = 0; ch < n_chunks; ++ch) {
            const float* src = Vd + cur * VD_FLOATS;
            float* dst = Vd + (cur ^ 1) * VD_FLOATS;
            const bool last = ch + 1 == n_chunks;
            const int c8 = 2 * ch;
            if (AFF) {
                // activation on the way in: the halo of the NEXT item goes through registers, loaded at the top of stage 0,
                // transformed and written at its end (after the very last item: a harmless re-stage of this tile's first chunk)
                if (last) {
                    if (more) stage_tile(b1, ty1, tx1);
                    stage_issue(0, dst);
                } else {
                    stage_issue((ch + 1) * KC4, dst);
                }
            }
            // stage 0: channels 0-7 of the chunk; next operands = channels 8-15 of the same buffer
            stage(src, 1, wblock(c8, cg), wblock(c8 + 1, cg), [&]() {
                stage_commit(dst);
                if (AFF) {
                    __syncthreads();                             // next buffer complete; every read of this one has been issued
                } else {
                    // plain inputs: the next item's 9 LDS-DMA loads were issued at the top of the PREVIOUS stage, before that
                    // stage's and this stage's 18 + 18 weight-fragment loads, and memory operations retire in order: at most
                    // 36 outstanding == the halo has landed.  No fence (it would drain the weight stream as well).
                    asm volatile("s_waitcnt vmcnt(36) lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
            });
            if (!AFF) {
                // the buffer this chunk read is free behind the barrier: the item after next goes into it, two stages ahead of
                // its barrier (HBM latency under load is longer than one stage)
                if (ch + 2 < n_chunks) {
                    stage_issue((ch + 2) * KC4, const_cast<float*>(src));
                } else {
                    if (ch + 2 == n_chunks && more) stage_tile(b1, ty1, tx1);       // from here on the next tile is staged
                    stage_issue((ch + 2 - n_chunks) * KC4, const_cast<float*>(src));
                }
            }
            // stage 1: channels 8-15; next operands = first 8 channels of the next item (other buffer)
            stage(dst, 0, wblock(c8 + 1, cg), last ? wblock(0, cg1) : wblock(c8 + 2, cg), [&]() {});
            cur ^= 1;
        }

        // ---- output transform Y = A^T M A on float4s (couts co .. co+3 of tile `tile`), bias, 16-byte stores, GN partials
        W4_MFMA_DRAIN();
        {
            const int co = cg * 16 + 4 * kq;
            const bool cok = co + 3 < Cout;
            f32x4 bias4 = {0, 0, 0, 0};
            if (a.d.bias && cok) bias4 = nd_ld4(a.d.bias + co);
            const int py0 = ty * 16 + 4 * (tile >> 2), px0 = tx * 16 + 4 * (tile & 3);
            f32x4 sum4 = {0, 0, 0, 0}, sq4 = {0, 0, 0, 0}, pivot4 = {0, 0, 0, 0};
            int cnt = 0;
            f32x4 Z[4][6];
#pragma unroll
            for (int nu = 0; nu < 6; ++nu) {
                f32x4 m[6], y[4];
#pragma unroll
                for (int xi = 0; xi < 6; ++xi) m[xi] = acc[xi * 6 + nu];
                w4_at(m, y);
#pragma unroll
                for (int i = 0; i < 4; ++i) Z[i][nu] = y[i];
            }
#pragma unroll
            for (int p = 0; p < NPOS; ++p) acc[p] = f32x4{0, 0, 0, 0};
            const bool full = ty * 16 + 16 <= H && tx * 16 + 16 <= W && cg * 16 + 16 <= Cout;      // wave-uniform
            const bool want_stats = a.d.stats != nullptr;
            float* lane_out = a.d.out + (((size_t)b * H + py0) * W + px0) * a.d.ldo + co;
            auto emit = [&](auto full_c, auto stats_c) {
                constexpr bool FULL = decltype(full_c)::value, STATS = decltype(stats_c)::value;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x4 y[4];
                    w4_at(Z[i], y);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 v = y[j] + bias4;
                        if (STATS && i == 0 && j == 0) {         // one pivot per cout for the whole workgroup tile: tile 0's first pixel
                            pivot4.x = __shfl(v.x, lane & 48); pivot4.y = __shfl(v.y, lane & 48);
                            pivot4.z = __shfl(v.z, lane & 48); pivot4.w = __shfl(v.w, lane & 48);
                        }
                        const bool inside = FULL || (py0 + i < H && px0 + j < W);
                        if (inside) {
                            if (STATS) {
                                const f32x4 dv = v - pivot4;
                                sum4 += dv;
                                sq4 += dv * dv;
                                ++cnt;
                            }
#if !(W4_ABLATE & 8)
                            if (FULL || cok) nd_st4(lane_out + (size_t)((i * W + j) * a.d.ldo), v);
#endif
                        }
                    }
                }
            };
            if (full) { if (want_stats) emit(std::true_type{}, std::true_type{}); else emit(std::true_type{}, std::false_type{}); }
            else { if (want_stats) emit(std::false_type{}, std::true_type{}); else emit(std::false_type{}, std::false_type{}); }
            if (a.d.stats) {
                // pool over the 16 tiles (the 16 lanes of a DPP row share their couts): sum = S + n p, M2 = Q - S^2 / n
                float fc = nd_row16_sum((float)cnt);
                f32x4 S, Q;
                S.x = nd_row16_sum(sum4.x); S.y = nd_row16_sum(sum4.y); S.z = nd_row16_sum(sum4.z); S.w = nd_row16_sum(sum4.w);
                Q.x = nd_row16_sum(sq4.x); Q.y = nd_row16_sum(sq4.y); Q.z = nd_row16_sum(sq4.z); Q.w = nd_row16_sum(sq4.w);
                const int slot = (ty * a.tiles_x + tx) * 2;
                if (tile == 0 && cok) {
                    fc = fmaxf(fc, 1.0f);
                    float* o = a.d.stats + (((size_t)b * a.slots + slot) * Cout + co) * 2;
                    const f32x4 sm = S + fc * pivot4;
                    const f32x4 m2 = Q - S * S / fc;
                    nd_st4(o, f32x4{sm.x, fmaxf(m2.x, 0.0f), sm.y, fmaxf(m2.y, 0.0f)});
                    nd_st4(o + 4, f32x4{sm.z, fmaxf(m2.z, 0.0f), sm.w, fmaxf(m2.w, 0.0f)});
                    const f32x4 zero = {0, 0, 0, 0};             // the second slot of the tile (F(2x2) kernels: lower half) stays empty
                    nd_st4(o + (size_t)Cout * 2, zero);
                    nd_st4(o + (size_t)Cout * 2 + 4, zero);
                }
                if (b == 0 && nt == 0 && wave == 0 && lane == 0) {
                    a.d.slot_count[slot] = (float)(min(16, H - ty * 16) * min(16, W - tx * 16));
                    a.d.slot_count[slot + 1] = 0.0f;
                }
            }
        }
        b = b1; ty = ty1; tx = tx1; nt = nt1;
    }
}

// OIHW (cout, cin, 3, 3) -> U = G g G^T in blocks [cin/8][cout/16][18 position pairs][64 lanes][4]:
// lane (cout = l & 15, k = l >> 4) holds {U[2pp][ch 2k], U[2pp][ch 2k+1], U[2pp+1][ch 2k], U[2pp+1][ch 2k+1]} of its block
__global__ void pack_wino4_kernel(const float* __restrict__ w, float* __restrict__ out, int cin, int cout, int n_c8, int n_cg) {
    const float G[6][3] = {{0.25f, 0.f, 0.f}, {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                           {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
    const size_t total = (size_t)n_c8 * n_cg * 18 * 64 * 4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int e = i & 3, l = (i >> 2) & 63;
        size_t r = i >> 8;
        const int pp = r % 18;  r /= 18;
        const int cg = r % n_cg;
        const int c8 = r / n_cg;
        const int pos = 2 * pp + (e >> 1), xi = pos / 6, nu = pos % 6;
        const int ch = c8 * 8 + 2 * (l >> 4) + (e & 1), co = cg * 16 + (l & 15);
        float u = 0.0f;
        if (ch < cin && co < cout) {
            const float* g = w + ((size_t)co * cin + ch) * 9;
            // fp64 accumulation of the 9 products: the packed weights are exact-rounded once
            double acc = 0.0;
            for (int rr = 0; rr < 3; ++rr)
                for (int ss = 0; ss < 3; ++ss) acc += (double)G[xi][rr] * (double)g[rr * 3 + ss] * (double)G[nu][ss];
            u = (float)acc;
        }
        out[i] = u;
    }
}

static inline int w4_cus() { return nd_device_cus(); }

template <int MODE>
int launch4(const Wino4Args& a, hipStream_t st) {
    static nd_device_once configured;
    const size_t lds = ((size_t)2 * VD_FLOATS + W4_STAGE_LOADS * 256) * sizeof(float);
    if (int e = nd_reserve_lds(configured, reinterpret_cast<const void*>(wino4_kernel<MODE>), lds, "nd_conv3x3_wino4")) return e;
    const long resident = w4_cus();                       // one workgroup per CU (registers)
    hipLaunchKernelGGL((wino4_kernel<MODE>), dim3((unsigned)(a.total_wg < resident ? a.total_wg : resident)), dim3(256), lds, st, a);
    return 0;
}

}  // namespace

extern "C" int64_t nd_pack_conv3x3_wino4_weight_floats(int cin, int cout) {
    return (int64_t)nd_round_up(nd_cdiv(cin, 8), 2) * nd_cdiv(nd_round_up(cout, 64), 16) * 18 * 256;     // whole 16-channel chunks
}

extern "C" int nd_pack_conv3x3_wino4_weight(const float* oihw, float* packed, int cin, int cout, void* stream) {
    ND_REQUIRE(oihw && packed, ND_E_BADARG, "nd_pack_conv3x3_wino4_weight: null pointer");
    ND_REQUIRE(cin > 0 && cout > 0, ND_E_BADARG, "nd_pack_conv3x3_wino4_weight: non-positive size");
    const int n_c8 = nd_round_up(nd_cdiv(cin, 8), 2), n_cg = nd_round_up(cout, 64) / 16;
    const size_t total = (size_t)n_c8 * n_cg * 18 * 256;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_wino4_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, oihw, packed, cin, cout, n_c8, n_cg);
    return nd_launch_status("nd_pack_conv3x3_wino4_weight");
}

extern "C" int nd_conv3x3_wino4_nhwc_f32(const nd_conv3x3* d, void* stream) {
    ND_REQUIRE(d, ND_E_BADARG, "nd_conv3x3_wino4: null descriptor");
    const nd_src& s = d->src;
    ND_REQUIRE(s.p0 && d->weight && d->out, ND_E_BADARG, "nd_conv3x3_wino4: null tensor pointer");
    ND_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->cin > 0 && d->cout > 0, ND_E_BADARG, "nd_conv3x3_wino4: non-positive size");
    ND_REQUIRE(d->cin % 4 == 0 && d->cout % 4 == 0 && d->cin > KC4, ND_E_SHAPE,
               "nd_conv3x3_wino4: cin=%d and cout=%d must be multiples of 4, cin > 16 (two K chunks in flight)", d->cin, d->cout);
    ND_REQUIRE(s.c0 + s.c1 == d->cin && s.c0 % 4 == 0 && s.c1 % 4 == 0 && s.c0 > 0, ND_E_SHAPE,
               "nd_conv3x3_wino4: source channels %d+%d do not match cin=%d (multiples of 4)", s.c0, s.c1, d->cin);
    ND_REQUIRE((s.c1 == 0) == (s.p1 == nullptr), ND_E_BADARG, "nd_conv3x3_wino4: p1/c1 mismatch");
    ND_REQUIRE(s.ld0 >= s.c0 && s.ld0 % 4 == 0 && (s.c1 == 0 || (s.ld1 >= s.c1 && s.ld1 % 4 == 0)), ND_E_ALIGN,
               "nd_conv3x3_wino4: pixel strides must be >= channels and multiples of 4");
    ND_REQUIRE(nd_aligned16(s.p0) && nd_aligned16(s.p1) && nd_aligned16(d->weight) && nd_aligned16(s.mad) && nd_aligned16(d->out) &&
               nd_aligned16(d->bias), ND_E_ALIGN, "nd_conv3x3_wino4: pointers must be 16-byte aligned");
    ND_REQUIRE(d->ldo >= d->cout && d->ldo % 4 == 0, ND_E_SHAPE, "nd_conv3x3_wino4: ldo must be >= cout and a multiple of 4");
    ND_REQUIRE(s.mode == ND_PRO_NONE || s.mode == ND_PRO_AFFINE_SILU, ND_E_BADARG,
               "nd_conv3x3_wino4: unsupported prologue %d (use nd_conv3x3_wino2_nhwc_f32)", s.mode);
    ND_REQUIRE(s.mode != ND_PRO_AFFINE_SILU || s.mad, ND_E_BADARG, "nd_conv3x3_wino4: affine prologue needs mad");
    ND_REQUIRE(!s.upsample && !s.unshuffle, ND_E_BADARG, "nd_conv3x3_wino4: no upsample / unshuffle addressing (use nd_conv3x3_wino2_nhwc_f32)");
    ND_REQUIRE((d->stats == nullptr) == (d->slot_count == nullptr), ND_E_BADARG, "nd_conv3x3_wino4: stats and slot_count go together");
    ND_REQUIRE(!d->stats || d->cout % 4 == 0, ND_E_SHAPE, "nd_conv3x3_wino4: statistics need cout %% 4 == 0");
    ND_REQUIRE(s.c1 == 0 || s.c0 % KC4 == 0, ND_E_SHAPE,
               "nd_conv3x3_wino4: first concat source has %d channels; a 16-channel K chunk must not straddle the sources", s.c0);
    {
        const long px = (long)d->B * d->H * d->W;
        ND_REQUIRE(px * s.ld0 * 4 < (1L << 31) && px * s.ld1 * 4 < (1L << 31), ND_E_SHAPE, "nd_conv3x3_wino4: a source tensor of 2 GiB or more");
    }

    Wino4Args a;
    a.d = *d;
    a.tiles_x = nd_cdiv(d->W, 16);
    a.tiles_y = nd_cdiv(d->H, 16);
    a.n_tiles = nd_cdiv(d->cout, 64);
    a.n_cg = nd_round_up(d->cout, 64) / 16;
    a.n_c8 = nd_round_up(nd_cdiv(d->cin, 8), 2);
    a.slots = a.tiles_x * a.tiles_y * 2;
    const long wg = (long)d->B * a.tiles_x * a.tiles_y * a.n_tiles;
    ND_REQUIRE(wg < (1L << 31), ND_E_SHAPE, "nd_conv3x3_wino4: grid too large");
    a.total_wg = (int)wg;
    hipStream_t st = (hipStream_t)stream;
    const int rc = s.mode == ND_PRO_AFFINE_SILU ? launch4<ND_PRO_AFFINE_SILU>(a, st) : launch4<ND_PRO_NONE>(a, st);
    if (rc) return rc;
    return nd_launch_status("nd_conv3x3_wino4_nhwc_f32");
}
