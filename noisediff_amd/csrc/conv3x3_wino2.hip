// conv3x3_wino2.hip -- Winograd F(2x2,3x3) 3x3 convolution, one-wave-per-SIMD structure.
//
// Same operator, descriptor, packed weights (nd_pack_conv3x3_wino_weight) and epilogue as conv3x3_wino.hip; what
// changes is where the state lives.  conv3x3_wino.hip keeps the four 2x2-output accumulators in registers and folds
// every position's partial product into them with VALU adds each K chunk -- 8 VALU instructions per MFMA, 256 VGPRs,
// two waves per SIMD, matrix pipe ~44 % busy.  Here one workgroup owns a CU (4 waves = one per SIMD, the whole
// 512-entry register file each):
//   * the SIXTEEN position accumulators M[xi][nu] (256 registers, the accumulator half of the file) stay resident for
//     the whole K loop, so the output transform Y = A^T M A runs ONCE per tile instead of once per chunk;
//   * a single in-order wave keeps the matrix pipe fed because nothing in the loop depends on MFMA results: per step
//     (8 MFMAs = 512 pipe cycles) it issues 8 LDS reads for the NEXT step, ~24 VALU for the input transform and two
//     weight-fragment loads -- all of which retire while the MFMAs execute;
//   * the raw halo tile is double-buffered in LDS (2 x 62 KB): the global loads of the next (tile, chunk) item are
//     issued before the chunk's MFMAs and their prologue transform + LDS writes are interleaved between the steps, so
//     there is one barrier per chunk and no exposed HBM round trip.
#include <stdlib.h>
#include <type_traits>
#include "nd_common.h"

namespace {

constexpr int KC = 32, LDA = KC + 4;
constexpr int HT = 18, PW = 12, NPIX = HT * HT, PLANE = HT * PW;
constexpr int WBLOCK = 16 * 8 * 64 * 4;              // floats per packed weight block (conv3x3_wino.hip)
constexpr int BUF = (2 * PLANE + 1) * LDA;          // floats per LDS buffer (+1 scratch pixel)
constexpr int STAGE_IT = (NPIX * 8 + 255) / 256;    // 11

#ifndef W2_CLUMP
#define W2_CLUMP 1           // all VALU of a step in one slice (0: spread over ten slices, the r1d arrangement; A/B builds)
#endif
#ifndef W2_XCD_REMAP
#define W2_XCD_REMAP 1
#endif
#ifndef W2_LAZY_AFFINE
#define W2_LAZY_AFFINE 1     // GroupNorm-affine constants folded at their first use instead of right behind their loads
#endif
#ifndef W2_LAG
#define W2_LAG 3             // steps between a staging load and its commit to LDS
#endif
#ifndef W2_WD
#define W2_WD 3              // steps a weight fragment is loaded ahead of its MFMAs (2 -> 3: -0.5 % per step in a same-box A/B at r1e; bw[] holds four)
#endif
#ifndef W2_ZERO_C
#define W2_ZERO_C 1          // peeled first chunk whose first MFMAs take C = 0 (0: accumulators re-zeroed in the epilogue; A/B builds)
#endif
#ifndef W2_ABLATE
#define W2_ABLATE 0          // diagnostic builds only (tools/w2_variants.sh): drop parts of the loop to time the rest
#endif
// The MFMAs go through inline asm: (1) "+a" pins the 16 accumulators to the accumulator half of the register file,
// (2) a volatile asm keeps its place between the sched_barriers (a builtin MFMA is a pure value and instruction
// selection clumps them), which is what lets the source dictate the MFMA / VALU / LDS / VMEM interleave.  The compiler
// cannot see that this is an MFMA, so the VALU->MFMA operand wait states ride along (hidden behind the matrix pipe's
// 64-cycle cadence) and the epilogue drains the pipe before it reads the accumulators.
// Wait states in front of the asm MFMAs.  The ISA rule at stake: a VGPR written by a VALU instruction may be read as an MFMA
// A/B operand only two wait states later, and hipcc's hazard recognizer pads nothing inside (or on behalf of) an asm string.
// r1 carried a blanket `s_nop 1` in front of every MFMA ("wrong without it").  r2 probed it with eleven side builds that keep
// the pad only for one class of MFMA at a time -- behind LDS reads, behind the VALU clump, behind buffer loads, behind another
// MFMA, for the zero-C MFMAs of a tile's first chunk, or nowhere (tools/w2_nop_probe.sh, profiles/r2_wino2_nop_probe.txt): every
// build is bit-identical to the padded one.  That is what the schedule implies: every B operand is loaded (hipcc counts and waits
// for its own loads in front of the asm), every A operand (Vc, packed adds) is written a whole step -- sixteen MFMAs -- before its
// first use, except at a chunk's start where six more packed adds stand between the last write of Vc[nu] and MFMA nu.  The pad is
// therefore kept only where VALU work directly precedes an MFMA, as insurance against a reordering of that work: the MFMA behind
// the VALU slice (i == 3) and the first MFMA of a step (i == 0; step 0 follows the chunk prologue's transforms).  It is free there
// (the matrix pipe holds the issue slot far longer than two wait states, tools/microbench/mfma_rate.hip).
#ifndef W2_NOP_MASK
#define W2_NOP_MASK 0x0009       // bit i: MFMA i of a step is padded (0xFFFF: the r1 blanket pad; tools/w2_nop_probe.sh builds others)
#endif
#define W2_MFMA_P(acc, av, bv) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(av), "v"(bv))
#define W2_MFMA_N(acc, av, bv) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(av), "v"(bv))
#define W2_MFMA_ZP(acc, av, bv) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=a"(acc) : "v"(av), "v"(bv))
#define W2_MFMA_ZN(acc, av, bv) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=a"(acc) : "v"(av), "v"(bv))
#define W2_MFMA_DRAIN() asm volatile("s_nop 15\n\ts_nop 15" ::: "memory")
#define W2_PIN(x) asm volatile("" : "+v"(x))          // value is complete here: keeps pure VALU work in its slice

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) char* gchar_p;        // global (not flat) pointers for the weight stream
typedef const __attribute__((address_space(1))) f32x4* gf32x4_p;
// s1 * a + s2 * b on four floats as two packed adds (a VALU slot costs the single in-order wave ~5 cycles of matrix
// time whether it carries one float or two); volatile so the pair stays in the slice it was written in
template <int S1, int S2>
__device__ __forceinline__ f32x4 w2_pk(const f32x4 a, const f32x4 b) {
    static_assert(S1 > 0 || S2 > 0, "no (-,-) combination in B^T");
    const f32x4 x = S1 > 0 ? a : b, y = S1 > 0 ? b : a;        // x + y or x - y
    f32x2 lo, hi;
    const f32x2 xl = {x[0], x[1]}, xh = {x[2], x[3]}, yl = {y[0], y[1]}, yh = {y[2], y[3]};
    if (S1 > 0 && S2 > 0) {
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(lo) : "v"(xl), "v"(yl));
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(hi) : "v"(xh), "v"(yh));
    } else {
        asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(lo) : "v"(xl), "v"(yl));
        asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(hi) : "v"(xh), "v"(yh));
    }
    return f32x4{lo[0], lo[1], hi[0], hi[1]};
}
template <int XI>
__device__ __forceinline__ f32x4 w2_bt(const f32x4 r1, const f32x4 r2) {   // row XI of B^T applied to (d[r1(XI)], d[r2(XI)])
    return w2_pk<(XI == 2 ? -1 : 1), ((XI == 0 || XI == 3) ? -1 : 1)>(r1, r2);
}

struct Wino2Args {
    nd_conv3x3 d;
    int tiles_x, tiles_y, n_tiles, coutP, slots, total_wg;
};

__device__ __forceinline__ constexpr int bt_r1(int xi) { return xi == 0 ? 0 : 1; }
__device__ __forceinline__ constexpr int bt_r2(int xi) { return xi == 3 ? 3 : 2; }
__device__ __forceinline__ constexpr float bt_s1(int xi) { return xi == 2 ? -1.0f : 1.0f; }
__device__ __forceinline__ constexpr float bt_s2(int xi) { return (xi == 0 || xi == 3) ? -1.0f : 1.0f; }
__device__ __forceinline__ constexpr int at_coef(int a, int xi) {
    return a == 0 ? (xi <= 2 ? 1 : 0) : (xi == 0 ? 0 : (xi == 1 ? 1 : -1));
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void wino2_kernel(const Wino2Args a) {
    constexpr bool MAP = MODE == ND_PRO_AFFINE_MAP_SILU;
    constexpr bool AFF = MODE == ND_PRO_AFFINE_SILU || MAP;
    extern __shared__ __attribute__((aligned(16))) float As[];          // [2][BUF]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, col = lane & 31;

    const nd_src& s = a.d.src;
    const int H = a.d.H, W = a.d.W, Cin = a.d.cin, Cout = a.d.cout;
    const int up = s.upsample ? 1 : 0;
    const int sH = H >> up, sW = W >> up;
    const int Ctot = s.c0 + s.c1;
    const int quad = tid & 7;
    const int by = 4 * wm + (col >> 3), bx = col & 7;
    const int a_base = ((2 * by) * PW + bx) * LDA + 4 * half;

    // ---- (tile, chunk) items of this persistent workgroup
    // consecutive workgroup ids go round-robin over the 8 XCDs: give every XCD one contiguous run of tile ranges, so that
    // neighbouring tiles (shared halo rows) and the weight blocks of one cout tile meet in the same L2
    const int wgid = W2_XCD_REMAP ? nd_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int t_begin = (int)((long)wgid * a.total_wg / gridDim.x), t_end = (int)((long)(wgid + 1) * a.total_wg / gridDim.x);
    if (t_begin >= t_end) return;
#ifdef W2_STAMP                 // diagnostic: shader-clock and 100 MHz wall stamps per workgroup -> clock under load
    const unsigned long long stamp_c0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long stamp_epi = 0;
#endif
    auto decode = [&](int t, int& b_, int& ty_, int& tx_, int& nt_) {
        int lid = t;
        nt_ = lid % a.n_tiles;  lid /= a.n_tiles;
        tx_ = lid % a.tiles_x;  lid /= a.tiles_x;
        ty_ = lid % a.tiles_y;
        b_ = lid / a.tiles_y;
    };

    // ---- staging state: loads of the NEXT item, committed into the other LDS buffer during the current chunk.
    //      VALU slots are what this kernel is short of (one in-order wave: every VALU instruction costs ~5 cycles of
    //      matrix time), so everything about a halo pixel that does not change from chunk to chunk lives in a
    //      thread-private LDS table (LDS instructions are nearly free next to an MFMA): its clamped source pixel index
    //      (per tile) and its LDS slot (per kernel).  Per chunk an iteration is then one 64-bit mad, the load, the
    //      activation, four selects and the LDS write.
    unsigned* tab = reinterpret_cast<unsigned*>(As + 2 * BUF) + tid;     // [1 + 2][STAGE_IT][256]: constants, pixel index x 2 tiles
    f32x4 raw[STAGE_IT], msc[MAP ? STAGE_IT : 1], msh[MAP ? STAGE_IT : 1];
    f32x4 tM = {0, 0, 0, 0}, tA = {1, 1, 1, 1}, tD = {0, 0, 0, 0};
    unsigned okmask = 0;                                   // bit it: halo pixel inside the image and channel valid
    bool second = false;
    const int stid = tid >> 3;                             // staging pixel of iteration 0
    unsigned ild4 = 0;
#pragma unroll
    for (int it = 0; it < STAGE_IT; ++it) {
        const int p = stid + it * 32;
        const int hy = p < NPIX ? p / HT : 31, hx = p < NPIX ? p - hy * HT : 31;      // 31: never inside an image tile
        const unsigned slot = ((p < NPIX ? ((hx & 1) * PLANE + hy * PW + (hx >> 1)) : 2 * PLANE) * LDA + quad * 4) * 4;
        tab[it * 256] = slot | ((unsigned)hy << 20) | ((unsigned)hx << 26);
    }
    // per tile: clamped source pixel of every halo pixel of this thread -> table `par`; returns the inside-image mask
    const unsigned px_oob = (unsigned)a.d.B * (unsigned)sH * (unsigned)sW;   // < 2^24 (host check): offset == num_records, out of range
    unsigned mask_all = 0;                                 // bit it: iteration it of this thread is a real halo pixel
#pragma unroll
    for (int it = 0; it < STAGE_IT; ++it) mask_all |= (stid + it * 32 < NPIX ? 1u : 0u) << it;
    auto tile_table = [&](int par, int b_, int ty_, int tx_) -> unsigned {
        const int y0 = ty_ * 16 - 1, x0 = tx_ * 16 - 1;
        unsigned* dstt = tab + (1 + par) * STAGE_IT * 256;
        if (ty_ > 0 && tx_ > 0 && y0 + HT <= H && x0 + HT <= W) {      // halo entirely inside the image (wave-uniform): no clamps, no tests
#pragma unroll
            for (int it = 0; it < STAGE_IT; ++it) {
                const unsigned c = tab[it * 256];
                const int y = y0 + (int)((c >> 20) & 31u), x = x0 + (int)(c >> 26);      // dummy lanes (31, 31): some pixel of the image, masked
                dstt[it * 256] = (unsigned)((b_ * sH + (min(y, H - 1) >> up)) * sW + (min(x, W - 1) >> up));
            }
            return mask_all;
        }
        unsigned mask = 0;
#pragma unroll
        for (int it = 0; it < STAGE_IT; ++it) {
            const unsigned c = tab[it * 256];
            const int y = y0 + (int)((c >> 20) & 31u), x = x0 + (int)(c >> 26);
            const bool ok = (c >> 26) != 31u && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
            mask |= (ok ? 1u : 0u) << it;
            const int yc = min(max(y, 0), H - 1), xc = min(max(x, 0), W - 1);
            // outside the image: the pixel index one past the tensor, so the buffer load itself returns the zero padding
            dstt[it * 256] = ok ? (unsigned)((b_ * sH + (yc >> up)) * sW + (xc >> up)) : px_oob;
        }
        return mask;
    };
    unsigned mask_cur = 0, mask_next = 0;                  // inside-image masks of this tile and the next one
    int par = 0;                                           // table of this tile; the next tile's is par ^ 1
    const unsigned* itab = tab;                            // pixel table of the item being staged
    // activations, like the weights, come through buffer loads (SGPR resource + scalar channel offset, 32-bit lane
    // offset = pixel * stride + quad): cheap to issue, and a lane whose channels lie beyond the tensor reads zeros.
    // A K chunk never straddles the two concat sources (host check), so the source is wave-uniform per chunk.
    const long src_px = (long)a.d.B * sH * sW;
    const __amdgpu_buffer_rsrc_t rsrc0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(s.p0), 0, (int)(src_px * s.ld0 * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(s.p1 ? s.p1 : s.p0), 0, (int)(src_px * (s.p1 ? s.ld1 : s.ld0) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcm = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(MAP ? s.map : s.p0), 0, MAP ? (int)(src_px * 2 * Ctot * 4) : 0, 0x00020000);
    __amdgpu_buffer_rsrc_t irsrc = rsrc0;
    int isoff = 0, imoff = 0;                              // scalar byte offsets: channel base inside the source / the map
    auto issue_begin = [&](int b_, int cb_, int par_, unsigned tilemask) {   // per chunk: channel quad of this thread
        itab = tab + (1 + par_) * STAGE_IT * 256;
        const bool sec = cb_ >= s.c0;                      // wave-uniform
        second = sec;
        irsrc = sec ? rsrc1 : rsrc0;
        ild4 = (unsigned)(sec ? s.ld1 : s.ld0) * 4u;
        isoff = (sec ? cb_ - s.c0 : cb_) * 4;
        const int c = cb_ + quad * 4;
        const bool cvalid = c < Cin;
        const int cs = cvalid ? c : 0;
        if (AFF) {
            const float* m = s.mad + (size_t)b_ * 3 * Ctot + cs;
            tM = nd_ld4(m); tA = nd_ld4(m + Ctot); tD = nd_ld4(m + 2 * Ctot);
#if !W2_LAZY_AFFINE
            tD = tD - tM * tA;                             // (v - M) * A + D = v * A + (D - M * A)
#endif                                                     // (lazy: folded in the first commit, three steps later -- here the single
                                                           //  in-order wave would sit out the whole L2 round trip of the three loads)
        }
        if (MAP) imoff = cb_ * 4;
        okmask = cvalid ? tilemask : 0u;
    };
    unsigned pix_r = 0, slot_r = 0;                        // table entries read one slice ahead of their use (W2_CLUMP)
    auto issue_pre = [&](int it) { pix_r = itab[it * 256]; };
    auto commit_pre = [&](int it) { slot_r = tab[it * 256]; };      // (masked in commit: no lone VALU instruction in this slice)
    auto issue_one = [&](int it) {
        const unsigned pix = W2_CLUMP ? pix_r : itab[it * 256];
        // the scalar offsets are wave-uniform by construction; saying so keeps hipcc from wrapping each load in a waterfall loop
        raw[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(irsrc, (unsigned)(__umul24(pix, ild4) + quad * 16),
                                                                                  __builtin_amdgcn_readfirstlane(isoff), 0));
        if (MAP) {                                         // the map has the conv's resolution (host: no upsample with MAP)
            const unsigned mo = (unsigned)(__umul24(pix, 2 * Ctot * 4) + quad * 16);
            msc[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcm, mo, __builtin_amdgcn_readfirstlane(imoff), 0));
            msh[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcm, mo, __builtin_amdgcn_readfirstlane(imoff + Ctot * 4), 0));
        }
    };
    auto commit = [&](int it, float* dst) {                // prologue transform + zero padding, registers -> LDS
        f32x4 v = raw[it];
        if (AFF) {
#if W2_LAZY_AFFINE
            if (it == 0) tD = tD - tM * tA;                // (v - M) * A + D = v * A + (D - M * A), once per staged item
#endif
            v = v * tA + tD;
            if (MAP) v = v * (msc[it] + 1.0f) + msh[it];
            v = nd_silu4(v);
        }
        if (MODE == ND_PRO_LEAKY || (MODE == ND_PRO_LEAKY_SECOND && second)) v = nd_leaky4(v);
        if (AFF) {                                         // silu(affine(0)) != 0: the padding is applied after the activation
            const f32x4 zero = {0, 0, 0, 0};
            v = ((okmask >> it) & 1u) ? v : zero;
        }   // otherwise out-of-image pixels were loaded as zeros (px_oob) and channels beyond cin meet zero weights
        const unsigned slot = (W2_CLUMP ? slot_r : tab[it * 256]) & 0xFFFFFu;
        nd_st4(reinterpret_cast<float*>(reinterpret_cast<char*>(dst) + slot), v);
    };

    f32x16 M[16];                                         // position accumulators, resident across the K loop
#if !W2_ZERO_C
#pragma unroll
    for (int p = 0; p < 16; ++p) M[p] = nd_zero16();
#endif

    // tile coordinates of this tile, the next one and the one after: decoded once, then advanced with carries
    // (the three integer divisions of decode() cost the in-order wave several hundred cycles per tile)
    auto advance = [&](int& b_, int& ty_, int& tx_, int& nt_) {
        if (++nt_ == a.n_tiles) {
            nt_ = 0;
            if (++tx_ == a.tiles_x) {
                tx_ = 0;
                if (++ty_ == a.tiles_y) { ty_ = 0; ++b_; }
            }
        }
    };
    int b, ty, tx, nt;
    decode(t_begin, b, ty, tx, nt);
    int b1 = b, ty1 = ty, tx1 = tx, nt1 = nt;
    advance(b1, ty1, tx1, nt1);
    int b2 = b1, ty2 = ty1, tx2 = tx1, nt2 = nt1;
    advance(b2, ty2, tx2, nt2);
    mask_cur = tile_table(0, b, ty, tx);
    if (t_begin + 1 < t_end) mask_next = tile_table(1, b1, ty1, tx1);
    issue_begin(b, 0, 0, mask_cur);
#pragma unroll
    for (int it = 0; it < STAGE_IT; ++it) { issue_pre(it); issue_one(it); }
#pragma unroll
    for (int it = 0; it < STAGE_IT; ++it) { commit_pre(it); commit(it, As); }
    __syncthreads();
    int cur = 0;
    const unsigned voff = (half * 64 + wn * 32 + col) * 16;                 // per-lane byte offset inside a weight block
    const int n_chunks = (Cin + KC - 1) / KC;
    // the weight stream goes through buffer loads: one resource for the whole packed tensor, the per-lane offset in a
    // VGPR (constant), block and fragment offsets in the scalar offset -- an SGPR-addressed load costs the in-order
    // wave ~5 issue cycles, a 64-bit per-lane address ~16 plus the adds (tools/microbench/mfma_issue.hip)
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.d.weight), 0, (int)(((Cin + 31) & ~31) * 16 * a.coutP * 4), 0x00020000);
    auto block_off = [&](int ch_, int nt_) {              // byte offset of weight block (chunk, n tile); wave-uniform
        return __builtin_amdgcn_readfirstlane((ch_ * a.n_tiles + nt_) * (WBLOCK * 4));
    };
    // ---- the K loop.  A chunk is 16 steps (xi, channel group g) of 16 MFMAs; the 8 patch-row reads and the 4 column
    //      terms of a step serve all four nu positions.  Software pipeline, one in-order wave per SIMD: LDS rows are
    //      read two steps ahead, the transformed operands V of step s+1 are built during step s, weight fragments are
    //      fetched WD steps ahead (across chunk and tile boundaries), staging iteration `it` of the next item is issued
    //      at step is(it) and committed LAG steps later.  Nothing in a step depends on that step's MFMAs; the source
    //      is written slice by slice (one MFMA + its share of the other work, pinned by sched_barrier) so that at
    //      most ~60 cycles of other instructions sit between two MFMAs and the matrix pipe never waits for the wave.
    constexpr int S = 16, LAG = W2_LAG, WD = W2_WD;
    f32x4 bw[4][4], dr[2][8], Vc[4], Vn[4], T[4];
    auto load_w1 = [&](int wb, int step, int nu) {
        const int g = step & 3, xi = step >> 2;
        bw[step & 3][nu] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, voff, wb + ((xi * 4 + nu) * 8 + 2 * g) * 1024, 0));
    };
    auto load_d1 = [&](const float* src, int step, int c, int which) {
        const int g = step & 3, xi = step >> 2;
        const int off = ((c & 1) * PLANE + (c >> 1)) * LDA + g * 8;
        if (which == 0) dr[step & 1][c] = nd_ld4(&src[a_base + off + bt_r1(xi) * PW * LDA]);
        else dr[step & 1][4 + c] = nd_ld4(&src[a_base + off + bt_r2(xi) * PW * LDA]);
    };
    auto make_t = [&](int step, int c) {
        const int xi = step >> 2;
        const f32x4 r1 = dr[step & 1][c], r2 = dr[step & 1][4 + c];
        T[c] = xi == 0 ? w2_bt<0>(r1, r2) : xi == 1 ? w2_bt<1>(r1, r2) : xi == 2 ? w2_bt<2>(r1, r2) : w2_bt<3>(r1, r2);
    };
    auto make_v = [&](int nu, f32x4 (&V)[4]) {
#if W2_ABLATE & 8
        V[nu] = T[nu];
#else
        const f32x4 r1 = T[bt_r1(nu)], r2 = T[bt_r2(nu)];
        V[nu] = nu == 0 ? w2_bt<0>(r1, r2) : nu == 1 ? w2_bt<1>(r1, r2) : nu == 2 ? w2_bt<2>(r1, r2) : w2_bt<3>(r1, r2);
#endif
    };
    int wblock = block_off(0, nt);
#pragma unroll
    for (int w = 0; w < WD; ++w)
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) load_w1(wblock, w, nu);

    for (int t = t_begin; t < t_end; ++t) {
        const bool more_tiles = t + 1 < t_end;
        const int nb_ = more_tiles ? b1 : b, nnt = more_tiles ? nt1 : nt;   // the tile whose first chunk is staged during this tile's last one

        auto chunk = [&](int ch, auto first_c) {
            constexpr bool FIRST = W2_ZERO_C && decltype(first_c)::value;     // first chunk of a tile: accumulators start from C = 0
            const float* src = As + cur * BUF;
            float* dst = As + (cur ^ 1) * BUF;
            const bool last_chunk = ch + 1 == n_chunks;
            // the item after the very last one is a harmless re-stage of this tile's first chunk: no branch in the loop
            if (last_chunk) issue_begin(nb_, 0, more_tiles ? par ^ 1 : par, more_tiles ? mask_next : mask_cur);
            else issue_begin(b, (ch + 1) * KC, par, mask_cur);
            const int wnext = last_chunk ? block_off(0, nnt) : block_off(ch + 1, nt);

#pragma unroll
            for (int c = 0; c < 8; ++c) load_d1(src, 0, c >> 1, c & 1);
#pragma unroll
            for (int c = 0; c < 8; ++c) load_d1(src, 1, c >> 1, c & 1);
#pragma unroll
            for (int c = 0; c < 4; ++c) make_t(0, c);
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) make_v(nu, Vc);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int step = 0; step < S; ++step) {
                const int xi = step >> 2;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int nu = i & 3, k = i >> 2;
                    {
                        const bool pad = (W2_NOP_MASK >> i) & 1;
                        if (FIRST && (step & 3) == 0 && k == 0) {         // first touch of the accumulator in this tile: C = 0
                            if (pad) W2_MFMA_ZP(M[xi * 4 + nu], Vc[nu][k], bw[step & 3][nu][k]);
                            else W2_MFMA_ZN(M[xi * 4 + nu], Vc[nu][k], bw[step & 3][nu][k]);
                        } else if (pad) W2_MFMA_P(M[xi * 4 + nu], Vc[nu][k], bw[step & 3][nu][k]);
                        else W2_MFMA_N(M[xi * 4 + nu], Vc[nu][k], bw[step & 3][nu][k]);
                    }
                    // the slice's share of the other work (<= ~40 issue cycles each)
#if !(W2_ABLATE & 4)
                    if (i < 8 && step + 2 < S) load_d1(src, step + 2, i >> 1, i & 1);
#endif
#if W2_CLUMP
                    // Every switch between the fp32 MFMA and ordinary VALU work costs ~17-20 cycles on top of the instructions
                    // themselves (tools/microbench/mfma_rate.hip: +20 for the first VALU instruction behind an MFMA, +4 for each
                    // further one), so ALL of a step's VALU -- both transform passes, staging addresses, the prologue
                    // activation -- sits in ONE slice; the other 15 carry only LDS reads and buffer loads.
                    if (i == 2 && step + 1 < S) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) make_t(step + 1, c);
#pragma unroll
                        for (int q = 0; q < 4; ++q) make_v(q, Vn);
                    }
                    if (i >= 8 && i < 12) {
#if !(W2_ABLATE & 2)
                        if (step + WD < S) load_w1(wblock, step + WD, i - 8);
                        else load_w1(wnext, step + WD - S, i - 8);
#endif
                    }
#if !(W2_ABLATE & 1)
#pragma unroll
                    for (int it = 0; it < STAGE_IT; ++it) {
                        if (it * (S - LAG) / STAGE_IT == step) { if (i == 0) issue_pre(it); if (i == 2) issue_one(it); }
                        if (it * (S - LAG) / STAGE_IT + LAG == step) { if (i == 1) commit_pre(it); if (i == 2) commit(it, dst); }
                    }
#endif
#else
                    if (i < 4) {
                        if (step + 1 < S) make_t(step + 1, i);
                    } else if (i < 8) {
                        if (step + 1 < S) make_v(i - 4, Vn);
                    } else if (i < 12) {
#if !(W2_ABLATE & 2)
                        if (step + WD < S) load_w1(wblock, step + WD, i - 8);
                        else load_w1(wnext, step + WD - S, i - 8);
#endif
                    }
#if !(W2_ABLATE & 1)
#pragma unroll
                    for (int it = 0; it < STAGE_IT; ++it) {
                        if (it * (S - LAG) / STAGE_IT == step && i == 12) issue_one(it);
                        if (it * (S - LAG) / STAGE_IT + LAG == step && i == 14) commit(it, dst);
                    }
#endif
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int nu = 0; nu < 4; ++nu) Vc[nu] = Vn[nu];
            }
            __syncthreads();                               // next buffer complete, current buffer free
            cur ^= 1;
            wblock = wnext;
        };
        // the first chunk is its own instance of the loop body (no branch inside the unrolled slices): its first MFMA on each
        // accumulator takes the inline constant 0 as C, so the epilogue does not spend 256 v_accvgpr_write on re-zeroing
        chunk(0, std::true_type{});
        for (int ch = 1; ch < n_chunks; ++ch) chunk(ch, std::false_type{});

        // ------------------------------------------------------------ output transform + epilogue of tile (b, ty, tx, nt)
        W2_MFMA_DRAIN();
        // this tile's pixel table is free now: fill it for the tile after next (its staging starts one tile from now)
        mask_cur = mask_next;
        if (t + 2 < t_end) mask_next = tile_table(par, b2, ty2, tx2);
        par ^= 1;
#ifdef W2_STAMP
        const unsigned long long stamp_e0 = __builtin_amdgcn_s_memtime();
#endif
        // accumulator register r of lane (col, half): block row r>>2, block column (r&3) + 4*half of this wave's 4 x 8
        // blocks.  Y = A^T M A one output row parity i at a time (Z_i[nu] = sum_xi A^T[i][xi] M[xi][nu], then the two
        // column parities), so that only ~64 registers are live; statistics are accumulated about a pivot value
        // (shifted sums: sum = S + cnt*p, M2 = Q - S^2/cnt) so that no second pass over the outputs is needed.
        {
            const int wrow0 = ty * 16 + wm * 8;
            const int rows_valid = max(0, min(8, H - wrow0));
            const int cols_valid = max(0, min(16, W - tx * 16));
            const int cnt = rows_valid * cols_valid;
            const int slot = (ty * a.tiles_x + tx) * 2 + wm;
            const int n = nt * 64 + wn * 32 + col;
            const bool nvalid = n < Cout;
            const float bias = (nvalid && a.d.bias) ? a.d.bias[n] : 0.0f;
            float* lane_out = a.d.out + ((size_t)(b * H + wrow0) * W + tx * 16 + 8 * half) * a.d.ldo + n;
            const bool full = rows_valid == 8 && cols_valid == 16 && nt * 64 + 64 <= Cout;   // wave-uniform
            float pivot = 0.0f, sS = 0.0f, sQ = 0.0f;
            f32x2 sS2 = {0.0f, 0.0f}, sQ2 = {0.0f, 0.0f};
            int Wt = W, ldot = a.d.ldo;
            asm volatile("" : "+s"(Wt), "+s"(ldot));      // per tile: keeps 64 store offsets from being hoisted into (spilled) SGPRs
            auto emit = [&](auto full_c) {
                constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    f32x16 Y0, Y1;
#pragma unroll
                    for (int r = 0; r < 16; ++r) { Y0[r] = bias; Y1[r] = bias; }
#pragma unroll
                    for (int nu = 0; nu < 4; ++nu) {
                        f32x16 z;                          // i = 0: M0 + M1 + M2;  i = 1: M1 - M2 - M3
                        if (i == 0) { z = M[nu]; z += M[4 + nu]; z += M[8 + nu]; }
                        else { z = M[4 + nu]; z -= M[8 + nu]; z -= M[12 + nu]; }
                        if (nu <= 2) Y0 += z;              // A^T[0] = (1, 1, 1, 0)
                        if (nu == 1) Y1 += z;              // A^T[1] = (0, 1, -1, -1)
                        if (nu >= 2) Y1 -= z;
                    }
                    if (i == 0) pivot = __shfl(Y0[0], col);   // the same pivot for both lanes (halves) of a column
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const f32x16& y = j == 0 ? Y0 : Y1;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int dy = 2 * (r >> 2) + i, dx = 2 * (r & 3) + j + 8 * half;
                            float* dstp = lane_out + (size_t)(((2 * (r >> 2) + i) * Wt + 2 * (r & 3) + j) * ldot);
                            if (FULL) {
                                if ((r & 1) == 0) {        // shifted sums two values per VALU slot
                                    const f32x2 dv = f32x2{y[r], y[r + 1]} - pivot;
                                    sS2 += dv;
                                    sQ2 += dv * dv;
                                }
                                *dstp = y[r];
                            } else if (dy < rows_valid && dx < cols_valid) {
                                const float dv = y[r] - pivot;
                                sS += dv;
                                sQ = fmaf(dv, dv, sQ);
                                if (nvalid) *dstp = y[r];
                            }
                        }
                    }
                }
            };
#if !(W2_ABLATE & 32)
            if (full) emit(std::true_type{});
            else emit(std::false_type{});
#endif
#if !W2_ZERO_C
#pragma unroll
            for (int p = 0; p < 16; ++p) M[p] = nd_zero16();
#endif
            if (a.d.stats) {
                sS += sS2[0] + sS2[1];
                sQ += sQ2[0] + sQ2[1];
                sS += __shfl_xor(sS, 32);
                sQ += __shfl_xor(sQ, 32);
                if (half == 0 && nvalid) {
                    const float fc = (float)max(cnt, 1);
                    float* st = a.d.stats + (((size_t)b * a.slots + slot) * Cout + n) * 2;
                    st[0] = fmaf(fc, pivot, sS);
                    st[1] = fmaxf(sQ - sS * sS / fc, 0.0f);
                }
            }
#ifndef W2_STAMP
            if (a.d.slot_count && b == 0 && nt == 0 && wn == 0 && lane == 0) a.d.slot_count[slot] = (float)cnt;
#endif
        }
#ifdef W2_STAMP
        stamp_epi += __builtin_amdgcn_s_memtime() - stamp_e0;
#endif
        b = b1; ty = ty1; tx = tx1; nt = nt1;
        b1 = b2; ty1 = ty2; tx1 = tx2; nt1 = nt2;
        advance(b2, ty2, tx2, nt2);
    }
#ifdef W2_STAMP
    if (tid == 0) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(a.d.slot_count) + 4 * blockIdx.x;
        dbg[0] = __builtin_amdgcn_s_memtime() - stamp_c0;
        dbg[1] = __builtin_amdgcn_s_memrealtime() - stamp_r0;
        dbg[2] = (unsigned long long)(t_end - t_begin) * n_chunks;
        dbg[3] = stamp_epi;
    }
#endif
}

static inline int device_cus() { return nd_device_cus(); }

template <int MODE>
int launch_mode(const Wino2Args& a, hipStream_t st) {
    static nd_device_once configured;
    const size_t lds = ((size_t)2 * BUF + 3 * STAGE_IT * 256) * sizeof(float);
    if (int e = nd_reserve_lds(configured, reinterpret_cast<const void*>(wino2_kernel<MODE>), lds, "nd_conv3x3_wino2")) return e;
    const long resident = device_cus();                   // one workgroup per CU by construction (LDS + registers)
    const dim3 grid((unsigned)(a.total_wg < resident ? a.total_wg : resident)), block(256);
    hipLaunchKernelGGL((wino2_kernel<MODE>), grid, block, lds, st, a);
    return 0;
}

}  // namespace

#ifndef W2_ENTRY
#define W2_ENTRY nd_conv3x3_wino2_nhwc_f32
#endif
extern "C" int W2_ENTRY(const nd_conv3x3* d, void* stream) {
    ND_REQUIRE(d, ND_E_BADARG, "nd_conv3x3_wino2: null descriptor");
    const nd_src& s = d->src;
    ND_REQUIRE(s.p0 && d->weight && d->out, ND_E_BADARG, "nd_conv3x3_wino2: null tensor pointer");
    ND_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->cin > 0 && d->cout > 0, ND_E_BADARG, "nd_conv3x3_wino2: non-positive size");
    ND_REQUIRE(d->cin % 8 == 0, ND_E_SHAPE, "nd_conv3x3_wino2: cin=%d must be a multiple of 8", d->cin);
    ND_REQUIRE(s.c0 + s.c1 == d->cin && s.c0 % 4 == 0 && s.c1 % 4 == 0 && s.c0 > 0, ND_E_SHAPE,
               "nd_conv3x3_wino2: source channels %d+%d do not match cin=%d (multiples of 4)", s.c0, s.c1, d->cin);
    ND_REQUIRE((s.c1 == 0) == (s.p1 == nullptr), ND_E_BADARG, "nd_conv3x3_wino2: p1/c1 mismatch");
    ND_REQUIRE(s.ld0 >= s.c0 && s.ld0 % 4 == 0 && (s.c1 == 0 || (s.ld1 >= s.c1 && s.ld1 % 4 == 0)), ND_E_ALIGN,
               "nd_conv3x3_wino2: pixel strides must be >= channels and multiples of 4");
    ND_REQUIRE(nd_aligned16(s.p0) && nd_aligned16(s.p1) && nd_aligned16(d->weight) && nd_aligned16(s.mad) && nd_aligned16(s.map),
               ND_E_ALIGN, "nd_conv3x3_wino2: pointers must be 16-byte aligned");
    ND_REQUIRE(d->ldo >= d->cout, ND_E_SHAPE, "nd_conv3x3_wino2: ldo < cout");
    const bool affine = s.mode == ND_PRO_AFFINE_SILU || s.mode == ND_PRO_AFFINE_MAP_SILU;
    ND_REQUIRE(s.mode == ND_PRO_NONE || affine || s.mode == ND_PRO_LEAKY || s.mode == ND_PRO_LEAKY_SECOND, ND_E_BADARG,
               "nd_conv3x3_wino2: unsupported prologue %d", s.mode);
    ND_REQUIRE(!affine || s.mad, ND_E_BADARG, "nd_conv3x3_wino2: affine prologue needs mad");
    ND_REQUIRE(s.mode != ND_PRO_AFFINE_MAP_SILU || s.map, ND_E_BADARG, "nd_conv3x3_wino2: map prologue needs map");
    ND_REQUIRE(!s.map_blocked, ND_E_BADARG, "nd_conv3x3_wino2: the blocked map layout is read by nd_conv3x3_wino4_nhwc_f32 only");
    ND_REQUIRE(!s.upsample || (d->H % 2 == 0 && d->W % 2 == 0 && s.c1 == 0), ND_E_SHAPE, "nd_conv3x3_wino2: upsample needs even H, W and one source");
    ND_REQUIRE(!s.unshuffle, ND_E_BADARG, "nd_conv3x3_wino2: unshuffle is a pointwise-only addressing mode");
    ND_REQUIRE(!(s.mode == ND_PRO_AFFINE_MAP_SILU && s.upsample), ND_E_BADARG,
               "nd_conv3x3_wino2: map prologue with upsample is not supported here (use nd_conv3x3_wino_nhwc_f32)");
    ND_REQUIRE((long)d->B * d->H * d->W < (1L << 24) && (long)(7 * d->W + 7) * d->ldo < (1L << 31), ND_E_SHAPE,
               "nd_conv3x3_wino2: more than 2^24 pixels (use nd_conv3x3_wino_nhwc_f32)");
    {
        const long px = (long)d->B * (d->H >> (s.upsample ? 1 : 0)) * (d->W >> (s.upsample ? 1 : 0));
        const long widest = s.mode == ND_PRO_AFFINE_MAP_SILU ? 2L * (s.c0 + s.c1) : 0;
        ND_REQUIRE(px * s.ld0 * 4 < (1L << 31) && px * s.ld1 * 4 < (1L << 31) && px * widest * 4 < (1L << 31), ND_E_SHAPE,
                   "nd_conv3x3_wino2: a source tensor of 2 GiB or more (use nd_conv3x3_wino_nhwc_f32)");
    }
    ND_REQUIRE(s.c1 == 0 || s.c0 % 32 == 0, ND_E_SHAPE,
               "nd_conv3x3_wino2: first concat source has %d channels; a 32-channel K chunk must not straddle the sources "
               "(use nd_conv3x3_wino_nhwc_f32)", s.c0);
#ifndef W2_STAMP
    ND_REQUIRE((d->stats == nullptr) == (d->slot_count == nullptr), ND_E_BADARG, "nd_conv3x3_wino2: stats and slot_count go together");
#endif

    Wino2Args a;
    a.d = *d;
    a.tiles_x = nd_cdiv(d->W, 16);
    a.tiles_y = nd_cdiv(d->H, 16);
    a.coutP = nd_round_up(d->cout, 64);
    a.n_tiles = nd_cdiv(d->cout, 64);
    a.slots = a.tiles_x * a.tiles_y * 2;
    const long wg = (long)d->B * a.tiles_x * a.tiles_y * a.n_tiles;
    ND_REQUIRE(wg < (1L << 31), ND_E_SHAPE, "nd_conv3x3_wino2: grid too large");
    a.total_wg = (int)wg;
    hipStream_t st = (hipStream_t)stream;
    int rc = 0;
    switch (s.mode) {
        case ND_PRO_AFFINE_SILU: rc = launch_mode<ND_PRO_AFFINE_SILU>(a, st); break;
        case ND_PRO_AFFINE_MAP_SILU: rc = launch_mode<ND_PRO_AFFINE_MAP_SILU>(a, st); break;
        case ND_PRO_LEAKY: rc = launch_mode<ND_PRO_LEAKY>(a, st); break;
        case ND_PRO_LEAKY_SECOND: rc = launch_mode<ND_PRO_LEAKY_SECOND>(a, st); break;
        default: rc = launch_mode<ND_PRO_NONE>(a, st);
    }
    if (rc) return rc;
    return nd_launch_status("nd_conv3x3_wino2_nhwc_f32");
}
