// conv3x3_wino4.hip -- Winograd F(4x4,3x3) 3x3 convolution on v_mfma_f32_16x16x4_f32.
//
// F(4x4,3x3) needs 36 multiplies per 16 outputs (2.25 per output) against 4 for F(2x2,3x3): 1.78x fewer MFMAs than
// conv3x3_wino2.hip, 4x fewer than the direct form.  On this chip the fp32 MFMA runs on the VALU's own lanes
// (tools/microbench/mfma_overlap.hip, r3: a second wave on the SIMD that issues only VALU work gets 0.01 instructions in per
// MFMA of the first; in the same wave a packed VALU instruction costs ~5 cycles of matrix time, v_accvgpr_read / v_exp / v_rcp
// ~8, every MFMA <-> VALU switch ~8 more), so the structure is built around ONE rule: no VALU instruction inside the MFMA
// loop, and every transform computed once per workgroup.
//
//   * workgroup = 4 waves, one per SIMD, the whole register file each; its tile = a 16 x 32 pixel region (two "tile
//     groups" of 16 tiles of 4x4 pixels) x 64 output channels.  Wave w owns the 16 couts [16 w, 16 w + 16) of BOTH tile groups:
//     36 positions x 2 tile groups = 72 accumulators of 4 registers (288: the first 64 live in the accumulator half of the
//     file, the last 8 in ordinary registers), so every weight fragment is loaded once per workgroup and serves two MFMAs.
//     Operands: A = transformed input V (rows = the 16 tiles of a tile group), B = transformed weights U (columns = the
//     wave's 16 couts): lane (cout = l & 15, kq = l >> 4) ends up with the four tiles of tile row kq for ONE cout, the output
//     transform runs on float4s over those four tiles and a store of one register is a dword per lane with 16 consecutive
//     lanes = 64 contiguous bytes (r3; r2 had the roles the other way round: four couts of one tile per lane, 16-byte stores
//     that the store path takes lane by lane).
//   * K is processed in chunks of 16 channels.  The 18 x 34 pixel halo of the region is fetched ONCE per chunk (10 x 16 bytes
//     per thread, registers) into a swizzled raw LDS image; GroupNorm-affine (+ per-pixel map) + SiLU / LeakyReLU prologues are
//     applied once per raw pixel when it is written to LDS.
//   * the input transform V = B^T d B runs once per workgroup on packed float2 (lane = (tile, channel pair): 36 raw reads, 168
//     packed VALU instructions, 18 16-byte writes into the tile group's V image [position pair][half][tile][pair]).
//   * the MFMA loop of a wave is then nothing but, per position pair: 2 ds_read_b128 (V of both tile groups), 1 buffer_load_b128
//     (U, a ring a whole 8-channel stage ahead), 8 MFMAs, and the staging of one halo item (stage 0: its address, one
//     v_mad_u32_u24, and the load; stage 1: its LDS write) whose LDS table entry is read BEFORE the eight MFMAs -- one wave per
//     SIMD means nobody else covers an LDS round trip.
//   * LDS (160 KB): raw image 45 KB at address 0, two V images 72 KB, per-thread tables 31 KB, the bias of every cout 8 KB.
//
// Transform matrices (Lavin & Gray, interpolation points 0, +-1, +-2, inf):
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
// fp32 rounding of these transforms is ~10x that of F(2x2,3x3) (~1e-5 relative; parity tests at 5e-5).
//
#include <stdlib.h>
#include <type_traits>
#include "nd_common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef __attribute__((address_space(3))) short* lds_short_ptr;
typedef __attribute__((address_space(3))) unsigned* lds_u32_ptr;
typedef __attribute__((address_space(3))) unsigned short* lds_u16_ptr;
typedef __attribute__((address_space(3))) f32x4* lds_f32x4_ptr;

constexpr int KC4 = 16;                              // channels per K chunk
constexpr int NPOS = 36;                             // positions (xi, nu) == patch entries (a, b)
constexpr int VD_FLOATS = NPOS * 256;                // one tile group's chunk image: 36 entries x 1 KB = 36 KB

// LDS map (bytes): [raw halo image][V image of every tile group: NTG x 36864][per-thread tables: source pixel u32 x RAW_IT, (NTG == 2: (row, column) u32 x RAW_IT,)
// raw-image address of a staged item u16 x RAW_IT, raw-image address of the transform lane's patch columns u16 x 12][NTG == 2: bias of every cout].  The raw
// image sits at address 0 so that its addresses fit 16 bits.
//
// NTG = tile groups (16 x 16-pixel tiles of 16 F(4x4) tiles) per workgroup:
//   NTG == 2 (r2/r3): a 16 x 32-pixel region, ONE workgroup per CU, 72 accumulators per wave, every weight fragment serves two MFMAs, 156 KB of LDS.
//   NTG == 1 (r4):    a 16 x 16-pixel region, TWO co-resident workgroups per CU (two waves per SIMD, 256 registers each: 36 accumulators = 128 AGPRs + 16 VGPRs),
//                     78 KB of LDS each.  Nothing overlaps a wave's own LDS round trips, barriers, store queue and transform with its MFMAs any more -- the
//                     OTHER workgroup's MFMAs do (the fp32 MFMA shares the VALU lanes, so only waits can be hidden, never VALU work); the price is that
//                     a weight fragment serves one MFMA quad instead of two (32 B/clk per CU from the L2 instead of 16).  It is also the F(4x4) path of
//                     images narrower than 32 pixels.  Same arithmetic in the same order: the two forms agree bit for bit.
constexpr int MAX_COUT = 2048;                                           // NTG == 2: the whole bias vector (padded to cout tiles) lives in LDS
template <int NTG, int NW = 4>
struct W4Geo {
    static constexpr int NT = 64 * NW;                                    // threads: NW waves (4: one per SIMD and workgroup; 8: two per SIMD in ONE workgroup)
    static constexpr int TGW = NTG * 4 / NW;                              // tile groups a wave multiplies: 2 (72 accumulators, 512 registers) or 1 (36, 256 registers)
    static constexpr int REG_W = 16 * NTG, HALO_W = REG_W + 2;            // region width, halo columns (18 halo rows)
    static constexpr int RAW_ROWP = NTG == 2 ? 40 : 24;                   // records of 64 bytes per halo row (row stride = 0 mod 256 bytes: the bank pattern of the swizzle)
    static constexpr int RAW_ITEMS = 18 * HALO_W * 4, RAW_IT = (RAW_ITEMS + NT - 1) / NT;      // (pixel, channel quad) items per chunk: 10 / 6 / 5 per thread
    static constexpr int RAW_FLOATS = 18 * RAW_ROWP * 16;
    static constexpr bool RTAB = TGW == 2;                                // the (row, column) table of the items (the 256-register forms recompute it per border tile: LDS)
    static constexpr bool BIAS_LDS = TGW == 2;                            // the whole bias vector in LDS (the 256-register forms: a register per tile)
    static constexpr int TAB_BYTES = (RAW_IT + (RTAB ? RAW_IT : 0)) * NT * 4 + (RAW_IT + 12) * NT * 2;
    static constexpr int BIAS_OFF_BYTES = RAW_FLOATS * 4 + NTG * VD_FLOATS * 4 + TAB_BYTES;
    static constexpr int LDS_BYTES = BIAS_OFF_BYTES + (BIAS_LDS ? MAX_COUT * 4 : 0);
    static constexpr int WG_PER_CU = NTG == 2 ? 1 : 2;
    static constexpr int ACC_AGPR = TGW == 2 ? 64 : 32;                   // accumulators [0, ACC_AGPR) in the AGPR half (256 / 128 registers), the rest in VGPRs
    static_assert(NW == 4 || (NW == 8 && NTG == 2), "four waves, or eight on the 16 x 32 region");
    static_assert(LDS_BYTES * WG_PER_CU <= 160 * 1024, "LDS map");
};

#ifndef W4_UR
#define W4_UR 18             // weight fragments in flight per wave in the K loop (x 4 registers); a stage consumes 18
#endif
#ifndef W4_UR_AFF
#define W4_UR_AFF 18         // ... of the GroupNorm-affine + SiLU variant (its transform needs more registers)
#endif
#ifndef W4_UR_MAP
#define W4_UR_MAP 6          // ... of the map variant (20 more staging registers per in-flight halo item): no spill in any fp32 instance; 9 measures the same
#endif
#ifndef W4_UR_GEN
#define W4_UR_GEN 6          // ... of the variant that forms the maps in the kernel (246 registers; 9: 256 and +9 us per launch, 12 spills; reading the pixel table ahead of the clump as the
                             // plain affine instance does: no gain)
#endif
#ifndef W4_UR_EPI
#define W4_UR_EPI 6          // ... across a tile's epilogue (its output transform needs the registers): the ring runs down in the
#endif                       // last stage of a tile and is refilled in one burst at the start of the next tile's first stage
#ifndef W4_VR
#define W4_VR 3              // V operand pairs read ahead
#endif
#ifndef W4_COMMIT_AT
#define W4_COMMIT_AT 8       // AFF / LEAKY: position pair of stage 1 behind which the 10 staged halo items are activated and written to LDS
#endif                       // (plain sources: item pp behind position pair pp, no VALU work)
#ifndef W4_XF_AT
#define W4_XF_AT 10          // position pair of stage 1 behind which the raw image is complete (barrier); the transform's 36 raw reads follow,
#endif                       // six per position pair, under the MFMAs of the stage's tail
#ifndef W4_U_AUX
#define W4_U_AUX 0           // cache-policy bits of the weight-fragment loads (experiment: 2 = nt)
#endif
#ifndef W4_STORE_AUX
#define W4_STORE_AUX 19      // cache-policy bits of the STREAMING output stores (sc0 | nt | sc1): see wino4_kernel's STREAM parameter
#endif
#ifndef W4_STAGGER
#define W4_STAGGER 12        // s_sleep units (64 cycles) between the 16 start phases of the workgroups; 0 = all start together
#endif
#ifndef W4_XF_SPLIT
#define W4_XF_SPLIT 1        // the "every wave has read its V operands" barrier between the input transform's two passes (0: behind both)
#endif
#ifndef W4_UR1
#define W4_UR1 9             // NTG == 1 (two workgroups per CU, 128 VGPRs per wave): weight fragments in flight per wave; a stage consumes 18
#endif
#ifndef W4_UR1_AFF
#define W4_UR1_AFF 6         // ... of its GroupNorm-affine + SiLU variant
#endif
#ifndef W4_UR1_EPI
#define W4_UR1_EPI 0         // ... across its epilogue (none: with any, the output transform spills -- and a scratch reload drains the stores issued before it)
#endif
#ifndef W4_UR1_XF
#define W4_UR1_XF 6          // ... across the input transform between two chunks (its 18 + 6 packed values need the registers)
#endif
#ifndef W4_PAIR_SKEW
#define W4_PAIR_SKEW 2       // NTG == 1: s_sleep(32) units (2048 cycles) by which the second workgroup of a CU starts later
#endif
#ifndef W4_PRIO
#define W4_PRIO 0            // NTG == 1: s_setprio level of a wave inside its transform / epilogue (VALU phases; the other workgroup's waves sit in MFMA stages at level 0)
#endif
#ifndef W4_ABLATE
#define W4_ABLATE 0          // diagnostic builds only: 1 no staging, 2 no weight loads, 4 no transform, 8 no epilogue stores,
                             // 16 halo loaded but not written to LDS, 32 written but not loaded, 64 every halo load from the same pixels
#endif

struct Wino4Args {
    nd_conv3x3 d;
    int tiles_x, tiles_y, regions_x, n_tiles, n_cg, n_c8, slots, total_wg;
    int splits, chunks_per_split;            // SPLIT instances: K (cin) in `splits` ranges of `chunks_per_split` 16-channel chunks, partial outputs [split][B][H][W][ldo]
};

// MFMAs through inline asm: the constraint pins each accumulator to its half of the register file for the whole kernel
// (with the builtin, hipcc moved accumulators between AGPR tuples and through VGPRs several times per stage) and a
// volatile asm keeps its place between the sched_barriers.  No wait states are needed in front of them: the ISA's
// VALU-write -> MFMA-operand rule cannot apply here, because no MFMA operand of this kernel is ever written by a VALU
// instruction -- A comes from buffer loads, B from LDS reads (hipcc waits for both in front of the asm), C from the
// previous MFMA on the accumulator or the inline constant 0.  (-DW4_NOP='"s_nop 1\n\t"' restores r1's pad: same
// bits, same time.)
#ifndef W4_NOP
#define W4_NOP ""
#endif
// (uv = a weight-fragment register: lane (cout l & 15, channel l >> 4) -> the B operand; vv = a V register: lane (tile l & 15, channel l >> 4) -> A)
#define W4_MFMA_A(acc, uv, vv)  asm volatile(W4_NOP "v_mfma_f32_16x16x4_f32 %0, %2, %1, %0" : "+a"(acc) : "v"(uv), "v"(vv))
#define W4_MFMA_V(acc, uv, vv)  asm volatile(W4_NOP "v_mfma_f32_16x16x4_f32 %0, %2, %1, %0" : "+v"(acc) : "v"(uv), "v"(vv))
#define W4_MFMA_AZ(acc, uv, vv) asm volatile(W4_NOP "v_mfma_f32_16x16x4_f32 %0, %2, %1, 0" : "=&a"(acc) : "v"(uv), "v"(vv))
#define W4_MFMA_VZ(acc, uv, vv) asm volatile(W4_NOP "v_mfma_f32_16x16x4_f32 %0, %2, %1, 0" : "=&v"(acc) : "v"(uv), "v"(vv))
#define W4_MFMA_DRAIN() asm volatile("s_nop 15\n\ts_nop 15" ::: "memory")

// one row of B^T applied to six packed values (the same code serves the column pass)
__device__ __forceinline__ void w4_bt(const f32x2 (&d)[6], f32x2 (&t)[6]) {
    // twelve packed instructions: every multiply rides in an fma (written out: left to itself hipcc spends fourteen)
    const f32x2 c4 = {4.0f, 4.0f}, cm4 = {-4.0f, -4.0f}, cm5 = {-5.0f, -5.0f}, c2 = {2.0f, 2.0f}, cm2 = {-2.0f, -2.0f};
    const f32x2 p = __builtin_elementwise_fma(cm4, d[2], d[4]), q = __builtin_elementwise_fma(cm4, d[1], d[3]);
    const f32x2 r = d[4] - d[2], s = d[3] - d[1];
    t[0] = __builtin_elementwise_fma(c4, d[0], __builtin_elementwise_fma(cm5, d[2], d[4]));
    t[1] = p + q;
    t[2] = p - q;
    t[3] = __builtin_elementwise_fma(c2, s, r);
    t[4] = __builtin_elementwise_fma(cm2, s, r);
    t[5] = __builtin_elementwise_fma(c4, d[1], __builtin_elementwise_fma(cm5, d[3], d[5]));
}

// rows 3 HALF .. 3 HALF + 2 of B^T d (six packed instructions each way: the split costs nothing)
template <int HALF>
__device__ __forceinline__ void w4_bt_half(const f32x2 (&d)[6], f32x2 (&t)[3]) {
    const f32x2 c4 = {4.0f, 4.0f}, cm4 = {-4.0f, -4.0f}, cm5 = {-5.0f, -5.0f}, c2 = {2.0f, 2.0f}, cm2 = {-2.0f, -2.0f};
    if (HALF == 0) {
        const f32x2 p = __builtin_elementwise_fma(cm4, d[2], d[4]), q = __builtin_elementwise_fma(cm4, d[1], d[3]);
        t[0] = __builtin_elementwise_fma(c4, d[0], __builtin_elementwise_fma(cm5, d[2], d[4]));
        t[1] = p + q;
        t[2] = p - q;
    } else {
        const f32x2 r = d[4] - d[2], s = d[3] - d[1];
        t[0] = __builtin_elementwise_fma(c2, s, r);
        t[1] = __builtin_elementwise_fma(cm2, s, r);
        t[2] = __builtin_elementwise_fma(c4, d[1], __builtin_elementwise_fma(cm5, d[3], d[5]));
    }
}

// STREAM: the output tensor is far larger than the L2s (host: >= ND_W4_STREAM_MB, default 48 MB): its stores carry the non-temporal
// system-scope policy bits, so the L2s stream them out instead of allocating lines for them.  With the default policy a 64 -> 64 layer at
// 256 x 256 spends 2.3 k cycles more in the K chunk that follows an epilogue (its halo reads queue behind the output's write-back) and
// 4.6 k more per workgroup in the prologue: -4 % per region tile (profiles/r3_w4_store_policy.txt); small outputs keep the default (the next
// kernel finds them in the L2 / Infinity Cache).
// SPLIT (split-K, plain or affine + SiLU sources; a statistics epilogue moves into the reduction): layers whose (sample, region, cout tile) items fill a fraction of the chip -- 512 -> 512 at
// 32 x 32 with 4 samples: 64 items for 256 CUs, each walking 32 K chunks -- are cut along cin: item (split, sample, region, cout tile) walks
// `chunks_per_split` chunks starting at chunk split * chunks_per_split and writes its partial sums to out[split] (the host passes a workspace
// and no bias); w4_splitk_reduce_kernel adds the partials in split order and the bias.  The split count is fixed by the shape alone.
template <int MODE, bool STREAM, bool SPLIT = false, int NTG = 2, int NW = 4>
__global__ __launch_bounds__((64 * NW), (W4Geo<NTG, NW>::WG_PER_CU)) void wino4_kernel(const Wino4Args a) {
    using Geo = W4Geo<NTG, NW>;
    constexpr int NT = Geo::NT, TGW = Geo::TGW;
    constexpr int RAW_FLOATS = Geo::RAW_FLOATS, BIAS_OFF_BYTES = Geo::BIAS_OFF_BYTES, ACC_AGPR = Geo::ACC_AGPR, REG_W = Geo::REG_W, HALO_W = Geo::HALO_W;
    constexpr bool MAP = MODE == ND_PRO_AFFINE_MAP_SILU;                  // + per-pixel scale / shift maps (ResnetBlock2)
    constexpr bool GEN = MODE == ND_PRO_AFFINE_GENMAP_SILU;               // ... the maps formed here, per chunk, from silu(pos_emb) (8 channels) on the matrix pipe: see gen_maps
    constexpr bool MAPX = MAP || GEN;
    constexpr bool AFF = MODE == ND_PRO_AFFINE_SILU || MAPX;              // GroupNorm-affine + SiLU applied while the halo is written to LDS
    constexpr bool LEAKY = MODE == ND_PRO_LEAKY || MODE == ND_PRO_LEAKY_SECOND;      // LSID: LeakyReLU(0.2) of the producer, applied by the consumer
    constexpr int UR = TGW == 1 ? (AFF ? W4_UR1_AFF : W4_UR1) : MAP ? W4_UR_MAP : GEN ? W4_UR_GEN : AFF ? W4_UR_AFF : W4_UR, UR_EPI = TGW == 1 ? W4_UR1_EPI : W4_UR_EPI;      // weight ring depth in the K loop / across the epilogue
    static_assert(TGW == 2 || !MAPX, "the map prologues (20 more staging registers per halo item in flight) stay on the 512-register form");
    extern __shared__ __attribute__((aligned(16))) float lds_[];        // the LDS map above (dynamic shared memory starts at LDS address 0)
    float* const Vd = lds_ + RAW_FLOATS;                                // [tg NTG][VD_FLOATS]: the V images

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // transform roles: NTG == 2: wave = (tile group, channel half): a lane owns a whole 6 x 6 patch of a channel pair; NTG == 1: wave = (row half, channel half):
    // a lane produces rows xi = 3 rh .. 3 rh + 2 of V for its (tile, channel pair) -- 256 work items either way
    // NW == 8: waves 0-3 multiply tile group 0, waves 4-7 tile group 1 (mtg), both on the same weight fragments (the second request hits the L1); the
    // transform is the half-row form over 512 lanes.
    const int mtg = NW == 8 ? wave >> 2 : 0;                             // TGW == 1: the tile group this wave multiplies
    const int tg = TGW == 2 ? wave >> 1 : mtg, rh = (wave >> 1) & 1, ch2 = wave & 1;      // transform role: tile group, row half (half form), channel half
    const int tile = lane & 15, kq = lane >> 4;

    const int wgid = nd_xcd_remap(blockIdx.x, gridDim.x);
    const int t_begin = (int)((long)wgid * a.total_wg / gridDim.x), t_end = (int)((long)(wgid + 1) * a.total_wg / gridDim.x);
    if (t_begin >= t_end) return;

#ifdef W4_STAMP                  // diagnostic (tools/w4_clock.py): shader-clock and 100 MHz wall stamps per workgroup -> clock under load, phase split
    const unsigned long long stamp_c0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long stamp_epi = 0, stamp_xf = 0, stamp_wait = 0, stamp_t = 0, stamp_drain = 0, stamp_first = 0, stamp_last = 0, stamp_second = 0, stamp_third = 0, stamp_pro = 0, stamp_top = 0, stamp_tile = 0;
#define W4_T0() (stamp_t = __builtin_amdgcn_s_memtime())
#define W4_ACC(x) (x += __builtin_amdgcn_s_memtime() - stamp_t)
#else
#define W4_T0()
#define W4_ACC(x)
#endif
#if W4_STAGGER
    // persistent workgroups with equal work run in lockstep: their halo requests and output stores would hit memory as chip-wide
    // bursts.  Spread the start over ~one chunk period (16 phases x W4_STAGGER x 64 cycles).
    // Only where a workgroup walks four tiles or more (the full-resolution layers: -3 %); with one or two tiles each the wait itself shows (+1.5 %).
    if (a.total_wg >= 4 * (int)gridDim.x)
        for (int k = (int)(blockIdx.x >> 3) & 15; k > 0; --k) __builtin_amdgcn_s_sleep(W4_STAGGER);
#endif
#if W4_PAIR_SKEW
    // NTG == 1: the two workgroups of a CU are dispatched together; the second one starts about half a K chunk later, so that its transform / epilogue phases
    // (no MFMAs) fall under the other's stage loops from the first chunk on
    if (NTG == 1 && 2 * blockIdx.x >= gridDim.x)
        for (int k = 0; k < W4_PAIR_SKEW; ++k) __builtin_amdgcn_s_sleep(32);
#endif
    const nd_src& s = a.d.src;
    const int H = a.d.H, W = a.d.W, Cin = a.d.cin, Cout = a.d.cout;
    const int up = s.upsample ? 1 : 0;
    const int sH = H >> up, sW = W >> up;
    const int Ctot = s.c0 + s.c1;
    const int n_chunks = SPLIT ? a.chunks_per_split : (Cin + KC4 - 1) / KC4;   // chunks an item walks

    auto decode = [&](int t, int& b_, int& ty_, int& rx_, int& nt_, int& sp_) {
        int lid = t;
        nt_ = lid % a.n_tiles;  lid /= a.n_tiles;
        rx_ = lid % a.regions_x;  lid /= a.regions_x;
        ty_ = lid % a.tiles_y;
        b_ = lid / a.tiles_y;
        sp_ = 0;
        if (SPLIT) { sp_ = b_ / a.d.B;  b_ -= sp_ * a.d.B; }             // the split is the slowest index: neighbouring workgroups share a K range's weights
    };

    // ---- staging.  The 18 x 34 pixel halo of the region (both tile groups) x 16 channels is fetched ONCE per chunk: item
    //      (pixel, channel quad) = 16 bytes, 2448 items over 256 threads (10 per thread), a wave instruction covers 16 pixels
    //      x 64 contiguous bytes.  (Fetching per (tile, patch entry) instead asks for every pixel 2.25 times in 32-byte pieces:
    //      4 x the cache-line fills, and the L1 fill path -- one 128-byte line per two cycles -- then bounds the kernel.)
    //      The raw image in LDS: record (row r, column c) of 64 bytes at index r * 40 + cperm(c), cperm swapping column bits
    //      0-1 with bits 2-3, and the channel pair P of a pixel in 8-byte slot P ^ swz(r, c): the transform's reads -- tiles 4 pixels
    //      apart in x and y -- then fall on different banks.
    constexpr int RAW_ROWP = Geo::RAW_ROWP, RAW_IT = Geo::RAW_IT;         // 40 records per row, 10 items per thread (NTG == 1: 24, 6)
    // Source addressing: byte address = resource base + soffset (SGPR: the region's base pixel and the chunk's channel base) +
    // voffset (VGPR: item pixel relative to the region x pixel stride + channel quad).  The resources start one row + one pixel in
    // front of the tensors, so that soffset is never negative, and end with the tensors.  Halo entries outside the image carry
    // the relative pixel PX_MARK = one past the last pixel: x any source's stride that is beyond its resource whatever the region
    // (gfx950 range-checks soffset + voffset; voffset alone is already out of range), and the load returns the zero padding -- as
    // does every item of an invalid channel quad (bias 0x7FFFFFF0).  No product or sum wraps: tensors stay below 1 GiB (host check).
    const long npx = (long)a.d.B * sH * sW;
    const unsigned PX_MARK = (unsigned)(npx + sW + 2);
    auto src_rsrc = [&](const float* p, long px_stride_floats) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p) - (long)(sW + 1) * px_stride_floats, 0,
                                                 (int)((npx + sW + 1) * px_stride_floats * 4), 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t rsrc0 = src_rsrc(s.p0, s.ld0);
    const __amdgpu_buffer_rsrc_t rsrc1 = s.p1 ? src_rsrc(s.p1, s.ld1) : rsrc0;
    const __amdgpu_buffer_rsrc_t rsrcm = MAP ? src_rsrc(s.map, 2 * Ctot) : GEN ? src_rsrc(s.map, 8) : rsrc0;      // GEN: silu(pos_emb), 8 floats per pixel
    const int map_shift = s.map_blocked ? 64 : Ctot * 4;                 // bytes from a channel's scale to its shift (blocked layout: [chunk][scale 16 | shift 16])
    float* const vd_tg = Vd + tg * VD_FLOATS;                            // the V image this wave's transform lanes write
    char* const rawbuf = reinterpret_cast<char*>(lds_);                  // [18][40] records of 64 bytes
    lds_u32_ptr const ptab = (lds_u32_ptr)(Vd + NTG * VD_FLOATS) + tid;  // [RAW_IT][256] source pixel of this thread's items
    auto cperm = [](int c) { return ((c >> 2) & 3) | ((c & 3) << 2) | (c & 48); };
    auto swz = [](int r, int c) { return (2 * ((r >> 2) & 3)) ^ (4 * ((c >> 3) & 1)); };       // slot swizzle of a pixel's 8 channel pairs (even: quads stay 16 contiguous bytes)
    // A wave's staging instruction covers 16 pixels x 4 channel quads = 64 contiguous bytes per pixel: lane = (pixel l >> 2, quad l & 3).  GEN: lane = (pixel l & 15,
    // quad l >> 4) -- the same items in the lane layout of a 16 x 16 MFMA result (column l & 15, rows 4 (l >> 4) .. + 3), so that the maps formed on the matrix pipe
    // land in the registers of the lane that stages the item (same cache lines per instruction; measured: the plain instance does not care, profiles/r6_w4_map_traffic_probe.txt)
    auto pixl_of = [](int t) { return GEN ? 16 * (t >> 6) + (t & 15) : t >> 2; };
    const int sq = GEN ? (tid >> 4) & 3 : tid & 3;                       // channel quad of this thread's items
    // per-thread constants live in LDS tables (thread-private columns), not in registers: [10] LDS address of staged item k,
    // [12] raw-image address of the transform lane's patch column b for patch rows 0-3 / 4-5 (the slot swizzle changes where the
    // patch crosses a multiple-of-4 row): entry (a, b) is at ttab[(a >> 2) * 6 + b] + a * RAW_ROWP * 64
    lds_u32_ptr const rtab = ptab + RAW_IT * NT;                       // [RAW_IT] item k: pixel relative to the region | halo row << 16 | column << 24 (stage_tile, once per tile; NTG == 2 only)
    lds_u16_ptr const dtab = (lds_u16_ptr)(ptab - tid + (Geo::RTAB ? 2 : 1) * RAW_IT * NT) + tid;    // [RAW_IT] LDS address of staged item k (16 bits: the raw image starts at 0)
    lds_u16_ptr const ttab = dtab + RAW_IT * NT;
    float* const bias_lds = reinterpret_cast<float*>(reinterpret_cast<char*>(lds_) + BIAS_OFF_BYTES);
    auto item_rc = [&](int k, int tid_) {                                // item k of this thread: pixel relative to the region | halo row << 16 | column << 24
        const int pix = pixl_of(tid_) + (NT / 4) * k, r = pix / HALO_W, c = pix - HALO_W * r;
        // pixel of halo entry (r, c) relative to the region's base pixel (one source row above, one pixel left of the halo origin):
        // nearest-x2 upsample addressing halves the coordinates -- (16 ty - 1 + r) >> 1 = 8 ty - 1 + ((r + 1) >> 1)
        const int dy = up ? (r + 1) >> 1 : r, dx = up ? (c + 1) >> 1 : c;
        return (unsigned)(dy * sW + dx) | ((unsigned)r << 16) | ((unsigned)c << 24);
    };
#pragma unroll
    for (int k = 0; k < RAW_IT; ++k) {
        const int pix = pixl_of(tid) + (NT / 4) * k, r = pix / HALO_W, c = pix - HALO_W * r;
        // (the per-item values are tables, not registers: kept in registers they get spilled, and a scratch reload in the K loop
        //  drains the weight ring; any per-item VALU arithmetic in the stage loops costs an MFMA <-> VALU switch)
        dtab[k * NT] = (unsigned short)((r * RAW_ROWP + cperm(c)) * 64 + (((2 * sq) ^ swz(r, c)) * 8));     // byte address in LDS (items beyond pixel 611 are never written)
        if (Geo::RTAB) rtab[k * NT] = item_rc(k, tid);
    }
    // the transform's own lane mapping (any lane may produce any V element): 16 consecutive lanes = 8 tiles x the two channel
    // pairs of a quad, so that the compiler's paired LDS accesses (ds_read2 / ds_write2: 16-lane groups, 32 banks) are conflict-free
    // on the raw image (with the swizzle above) and on the V image (8 tiles x 16 contiguous bytes)
    const int t_tile = ((lane >> 1) & 7) | ((lane >> 4) & 1) << 3, t_kq = (lane & 1) | ((lane >> 5) << 1);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int bx = 0; bx < 6; ++bx) {
            const int r0 = 4 * (t_tile >> 2) + 4 * h, c = 16 * tg + 4 * (t_tile & 3) + bx;
            ttab[(h * 6 + bx) * NT] = (unsigned short)((4 * (t_tile >> 2) * RAW_ROWP + cperm(c)) * 64 + (((4 * ch2 + t_kq) ^ swz(r0, c)) * 8));
        }
    const unsigned t_lds = (unsigned)(ch2 * 1024 + (t_kq >> 1) * 512 + t_tile * 32 + (t_kq & 1) * 16);  // V image address of the transform lane
    // the bias of every cout (zero beyond cout / without a bias) -> LDS, once per workgroup: the epilogues read it with an LDS load.  (A global
    // load there shares the in-order vmcnt counter with the output stores: waiting for it drained every store issued before it.)
    // NTG == 1 has no LDS left for it: the lane's bias (cout = cg * 16 + (l & 15)) is loaded into a register at the start of every tile, long before the
    // epilogue (a resource of zero records without a bias: the load returns 0, as it does for a padded cout)
    if (Geo::BIAS_LDS)
        for (int i = tid; i < a.n_tiles * 64; i += NT) bias_lds[i] = (a.d.bias && i < Cout) ? a.d.bias[i] : 0.0f;
    const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.d.bias ? a.d.bias : a.d.weight), 0, a.d.bias ? Cout * 4 : 0, 0x00020000);
    float bias_r = 0.0f;

    const unsigned OOB = 0x7FFFFFF0u;                                    // byte offset beyond any tensor: the load returns zeros (padding)
    int sb_ = 0;
    unsigned spx_ = 0;                                                   // the staged region's base pixel (scalar)
    bool tab_clean = false;                                              // ptab holds the unmasked table of an interior region
    f32x4 tA4 = {1, 1, 1, 1}, tD4 = {0, 0, 0, 0};                        // AFF: loaded with a chunk's halo, applied when it is written to LDS
    auto stage_tile = [&](int b_, int ty_, int rx_) {
        sb_ = b_;
        spx_ = (unsigned)((b_ * sH + ((ty_ * 16) >> up)) * sW + ((rx_ * REG_W) >> up));
        // halo rows / columns inside the image (scalars); a region away from the border keeps the table of the one before it
        const int r_lo = ty_ == 0 ? 1 : 0, r_n = min(17, H - ty_ * 16) - r_lo;
        const int c_lo = rx_ == 0 ? 1 : 0, c_n = min(HALO_W - 1, W - rx_ * REG_W) - c_lo;
        const bool interior = r_lo == 0 && r_n == 17 && c_lo == 0 && c_n == HALO_W - 1;
        if (!(interior && tab_clean)) {
            int tid_ = tid;                                              // NTG == 1 recomputes the items' (row, column) here: kept visible, hipcc computes the 18 values once,
            if (!Geo::RTAB) asm volatile("" : "+v"(tid_));               // spills them and reloads them from scratch -- a vmcnt(0) drain of the weight ring per border tile
#pragma unroll
            for (int k = 0; k < RAW_IT; ++k) {
                const unsigned rc = Geo::RTAB ? rtab[k * NT] : item_rc(k, tid_);
                const unsigned r = (rc >> 16) & 255u, c = rc >> 24;
                const bool ok = r - (unsigned)r_lo <= (unsigned)r_n && c - (unsigned)c_lo <= (unsigned)c_n;   // (items beyond the halo's last pixel have r >= 18)
                ptab[k * NT] = ok ? (rc & 0xFFFFu) : PX_MARK;            // outside the image: beyond the resource, the load returns the zero padding
            }
        }
        tab_clean = interior;
    };
    // the chunk being staged: source, channel base, affine constants of the transform lane
    f32x4 raw[RAW_IT];                                                   // the halo in flight: loaded during a chunk's stage 0, written to LDS during its stage 1
    constexpr int GEN_H = RAW_IT / 2;                                    // GEN: the maps of half the items at a time (registers)
    f32x4 msc[MAP ? RAW_IT : GEN ? GEN_H : 1], msh[MAP ? RAW_IT : GEN ? GEN_H : 1];     // MAP: the items' scale / shift map values (GEN: between gen_maps and the clump)
    float ev0[GEN ? RAW_IT : 1], ev1[GEN ? RAW_IT : 1];                  // GEN: silu(pos_emb)[pixel][sq], [sq + 4] of the items: the B operand (k = l >> 4, pixel = l & 15)
    float gw_sc0 = 0, gw_sc1 = 0, gw_sh0 = 0, gw_sh1 = 0;                // GEN: mlp[1].weight[channel cb + (l & 15)][sq], [sq + 4], scale and shift rows: the A operand
    f32x4 gb_sc = {0, 0, 0, 0}, gb_sh = {0, 0, 0, 0};                    // GEN: mlp[1].bias of this lane's quad: the C operand of the first MFMA
    int i_eoff = 0;
    bool i_second = false;                                               // wave-uniform: the chunk comes from the second concat source
    __amdgpu_buffer_rsrc_t i_rs = rsrc0;
    int i_soff = 0, i_mapoff = 0;
    unsigned i_ld4 = 0, i_bias = 0;
    auto stage_issue_begin = [&](int cb_) {
        const bool sec = cb_ >= s.c0;                                    // wave-uniform: a chunk never straddles the sources (host check)
        i_second = sec;
        i_mapoff = (int)(cb_ * (s.map_blocked ? 8u : 4u) + spx_ * (unsigned)(2 * Ctot * 4));   // blocked: chunk cb / 16 at 128 bytes each
        i_rs = sec ? rsrc1 : rsrc0;
        i_ld4 = (unsigned)(sec ? s.ld1 : s.ld0) * 4u;
        i_soff = (int)((sec ? cb_ - s.c0 : cb_) * 4u + spx_ * i_ld4);
        i_bias = cb_ + 4 * sq < Cin ? 16u * sq : OOB;                    // invalid channel quad: every item out of range
        if (GEN) {                                                       // (host: one source, whole chunks)
            i_eoff = (int)(spx_ * 32u);
            const float* wr = s.gamma + (size_t)(cb_ + (lane & 15)) * 8 + sq;
            gw_sc0 = wr[0];  gw_sc1 = wr[4];  gw_sh0 = wr[(size_t)Ctot * 8];  gw_sh1 = wr[(size_t)Ctot * 8 + 4];
            gb_sc = nd_ld4(s.beta + cb_ + 4 * sq);  gb_sh = nd_ld4(s.beta + Ctot + cb_ + 4 * sq);
        }
        if (AFF) {                                                       // this thread's channel quad: cb + 4 sq .. + 3
            const int c = cb_ + 4 * sq;
            const float* m = s.mad + (size_t)sb_ * 3 * Ctot + (c < Cin ? c : 0);
            const f32x4 M = nd_ld4(m), A = nd_ld4(m + Ctot), D = nd_ld4(m + 2 * Ctot);
            tA4 = A;
            tD4 = D - M * A;                                             // (v - M) * A + D = v * A + (D - M * A)
        }
    };
    auto stage_issue_one = [&](int k, unsigned px, auto tile_first_c) {  // px = ptab[k * NT], read by the caller one step ahead; tile_first_c: the staged chunk is its tile's first
#if !(W4_ABLATE & 1)
        // ONE VALU instruction per item (no branch, no masking: every per-item instruction in a stage loop costs an MFMA <-> VALU
        // switch): pixel x stride + this lane's channel-quad offset, the latter out of range for a quad beyond cin.  The table entry
        // is read from LDS BEFORE the position pair's eight MFMAs (`pre`): read behind them, its ~100 cycles of LDS latency stood between
        // the last MFMA of one position pair and the first of the next, twenty times per chunk.
        const unsigned voff = __umul24(px, i_ld4) + i_bias;
        raw[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(i_rs, voff, i_soff, 0));
        if (GEN && decltype(tile_first_c)::value) {                      // 8 bytes of the pixel's 32 per lane (the four quads of a pixel read all of them), ONCE PER TILE: the
            // items' pixels are the same in every chunk of a tile, and 32 bytes per pixel and chunk next to the chunk's own 64 cost 48 us of 350 (r6_w4_map_traffic_probe.txt)
            const unsigned eo = __umul24(px, 32u) + 4u * sq;
            ev0[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrcm, eo, i_eoff, 0));
            ev1[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrcm, eo, i_eoff + 16, 0));
        }
        if (MAP) {                                                       // the maps have the conv's resolution (host: no upsample with MAP): [pixel][scale C | shift C]
            const unsigned mo = __umul24(px, (unsigned)(2 * Ctot * 4)) + i_bias;
            msc[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcm, mo, i_mapoff, 0));
            msh[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcm, mo, i_mapoff + map_shift, 0));
        }
#else
        raw[k] = f32x4{(float)k, 1.0f, 0.5f, 0.25f};
#endif
    };
    // GEN: scale | shift of the staged chunk's ten items on the matrix pipe: D[16 channels][16 pixels] = W[16][8] E[8][16] + bias, two K steps of v_mfma_f32_16x16x4_f32
    // each -- lane (pixel l & 15, quad l >> 4) receives rows 4 (l >> 4) .. + 3 of column l & 15: its own item's four channels.  40 MFMAs per chunk and wave (+ 14 %)
    // instead of 2 x 16 bytes of map per item from HBM (537 MB per launch: 80 us of 361, profiles/r6_w4_map_traffic_probe.txt).  No operand is written by a vector
    // instruction (loads only); dependent MFMAs are ten apart; the results are read by the clump behind W4_MFMA_DRAIN.
    auto gen_maps = [&](int k0) {                                        // items k0 .. k0 + GEN_H - 1
#pragma unroll
        for (int k = 0; k < GEN_H; ++k) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %3" : "=&v"(msc[GEN ? k : 0]) : "v"(gw_sc0), "v"(ev0[GEN ? k0 + k : 0]), "v"(gb_sc));
#pragma unroll
        for (int k = 0; k < GEN_H; ++k) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %3" : "=&v"(msh[GEN ? k : 0]) : "v"(gw_sh0), "v"(ev0[GEN ? k0 + k : 0]), "v"(gb_sh));
#pragma unroll
        for (int k = 0; k < GEN_H; ++k) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(msc[GEN ? k : 0]) : "v"(gw_sc1), "v"(ev1[GEN ? k0 + k : 0]));
#pragma unroll
        for (int k = 0; k < GEN_H; ++k) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(msh[GEN ? k : 0]) : "v"(gw_sh1), "v"(ev1[GEN ? k0 + k : 0]));
        W4_MFMA_DRAIN();
    };
    auto stage_commit_one = [&](int k, unsigned daddr, unsigned pxk) {   // daddr = dtab[k * NT], pxk = ptab[k * NT] (AFF only): read ahead by the caller
        if (k == RAW_IT - 1 && pixl_of(tid) + (NT / 4) * k >= 18 * HALO_W) return;   // the last round covers 36 (NTG == 1: 4) pixels only
        f32x4 v = raw[k];
        if (AFF) {
            // GroupNorm-affine + SiLU on the raw halo (each pixel once: 40 values per thread and chunk); silu(x) = x / (1 + 2^(-x log2 e)).
            // The zero padding is applied after the activation (silu(affine(0)) != 0): items outside the image carry PX_MARK
            f32x4 x = v * tA4 + tD4;
            if (MAPX) x = x * (msc[MAP ? k : GEN ? k % GEN_H : 0] + 1.0f) + msh[MAP ? k : GEN ? k % GEN_H : 0];                   // ResnetBlock2: x * (scale + 1) + shift per pixel (Diffusion_arch.py:188-192)
            const f32x4 t = x * -1.44269504088896340736f;
            f32x4 e;
            e.x = __builtin_amdgcn_exp2f(t.x); e.y = __builtin_amdgcn_exp2f(t.y); e.z = __builtin_amdgcn_exp2f(t.z); e.w = __builtin_amdgcn_exp2f(t.w);
            e = e + 1.0f;
            f32x4 r;
            r.x = __builtin_amdgcn_rcpf(e.x); r.y = __builtin_amdgcn_rcpf(e.y); r.z = __builtin_amdgcn_rcpf(e.z); r.w = __builtin_amdgcn_rcpf(e.w);
            const bool inside = pxk != PX_MARK && i_bias != OOB;
            const f32x4 zero = {0, 0, 0, 0};
            v = inside ? x * r : zero;
        }
        if (MODE == ND_PRO_LEAKY || (MODE == ND_PRO_LEAKY_SECOND && i_second)) v = nd_leaky4(v);      // keeps zeros: the padding needs no mask
        *reinterpret_cast<lds_f32x4_ptr>(daddr) = v;
    };

    // ---- V image of a tile group (36 KB): [position pair 18][8-channel half g2][kq >> 1][tile 16][kq & 1] x 16 bytes, the 16 bytes
    //      being positions (2pp, 2pp+1) x channels (2kq, 2kq+1) -- what one MFMA lane needs for a position pair, one conflict-free
    //      ds_read_b128 (twice the bytes per LDS cycle of the 8-byte forms).  MFMA lane: + g2 * 1024 + pp * 2048 bytes
    const unsigned d_lds = (unsigned)((kq >> 1) * 512 + tile * 32 + (kq & 1) * 16);

    // input transform of the staged halo (raw image) into this tile group's V image (this wave: half ch2), in three steps so that
    // the raw reads can run under the tail of a stage: xf_addr (the lane's 12 raw addresses), xf_read (patch row `ay`), xf_finish
    // (B^T d B in registers, then -- behind the caller's barrier: every wave has read its last V operands -- the V image)
    auto xf_addr = [&](unsigned (&t_addr)[12]) {
#pragma unroll
        for (int i = 0; i < 12; ++i) t_addr[i] = ttab[i * NT];
    };
    auto xf_read = [&](f32x2 (&T)[6][6], const unsigned (&t_addr)[12], int ay) {
#if !(W4_ABLATE & 4)
#pragma unroll
        for (int bx = 0; bx < 6; ++bx)
            T[ay][bx] = *reinterpret_cast<const f32x2*>(rawbuf + t_addr[(ay >> 2) * 6 + bx] + ay * (RAW_ROWP * 64));
#endif
    };
    auto xf_finish = [&](f32x2 (&T)[6][6], float* buf, auto&& before_write) {
#if !(W4_ABLATE & 4)
        char* base = reinterpret_cast<char*>(buf) + t_lds;
#pragma unroll
        for (int bx = 0; bx < 6; ++bx) {                                 // T <- B^T T (over the patch rows, every column)
            f32x2 col[6], t[6];
#pragma unroll
            for (int ay = 0; ay < 6; ++ay) col[ay] = T[ay][bx];
            w4_bt(col, t);
#pragma unroll
            for (int xi = 0; xi < 6; ++xi) T[xi][bx] = t[xi];
        }
#if W4_XF_SPLIT
        before_write();                                                  // between the passes: the 18 writes below leave row by row, under the second pass' VALU
#endif                                                                   // (the LDS takes 16-byte writes at ~77 B/clk: the 72 KB of a chunk's V images are ~940 cycles)
#pragma unroll
        for (int xi = 0; xi < 6; ++xi) {                                 // V[xi] = T[xi] B
            f32x2 v[6];
            w4_bt(T[xi], v);
#if W4_XF_SPLIT
#pragma unroll
            for (int h = 0; h < 3; ++h)                                  // 16 bytes = positions (xi, 2h), (xi, 2h + 1) of this lane's channel pair
                *reinterpret_cast<f32x4*>(base + (xi * 3 + h) * 2048) = f32x4{v[2 * h].x, v[2 * h].y, v[2 * h + 1].x, v[2 * h + 1].y};
            __builtin_amdgcn_sched_barrier(0);
#else
#pragma unroll
            for (int bx = 0; bx < 6; ++bx) T[xi][bx] = v[bx];
#endif
        }
#if !W4_XF_SPLIT
        before_write();
#pragma unroll
        for (int xi = 0; xi < 6; ++xi)
#pragma unroll
            for (int h = 0; h < 3; ++h)
                *reinterpret_cast<f32x4*>(base + (xi * 3 + h) * 2048) = f32x4{T[xi][2 * h].x, T[xi][2 * h].y, T[xi][2 * h + 1].x, T[xi][2 * h + 1].y};
#endif
#else
        before_write();
#endif
    };

    // NTG == 1: the same transform with a lane producing HALF the rows of V for its (tile, channel pair) -- waves 0, 1: xi = 0..2, waves 2, 3: xi = 3..5 --
    // so that all 256 threads take part with 18 + 6 live packed values instead of 36 (128 VGPRs per wave).  Column by column: six raw reads, three rows
    // of B^T d; then the three rows times B and their nine 16-byte writes.  It runs between two barriers of its own (the raw image is complete and the V
    // image is free / the V image is complete); the other workgroup of the CU has the matrix pipe meanwhile.
    auto xf_half = [&](float* buf) {
#if !(W4_ABLATE & 4)
        unsigned t_addr[12];
        xf_addr(t_addr);
        f32x2 T[3][6];
        auto pass1 = [&](auto half_c) {
#pragma unroll
            for (int bx = 0; bx < 6; ++bx) {
                f32x2 col[6], t[3];
#pragma unroll
                for (int ay = 0; ay < 6; ++ay) col[ay] = *reinterpret_cast<const f32x2*>(rawbuf + t_addr[(ay >> 2) * 6 + bx] + ay * (RAW_ROWP * 64));
                w4_bt_half<decltype(half_c)::value>(col, t);
#pragma unroll
                for (int i = 0; i < 3; ++i) T[i][bx] = t[i];
            }
        };
        if (rh == 0) pass1(std::integral_constant<int, 0>{}); else pass1(std::integral_constant<int, 1>{});
        char* base = reinterpret_cast<char*>(buf) + t_lds + rh * (9 * 2048);
#pragma unroll
        for (int i = 0; i < 3; ++i) {                                    // V[3 rh + i] = T[i] B
            f32x2 v[6];
            w4_bt(T[i], v);
#pragma unroll
            for (int h = 0; h < 3; ++h)
                *reinterpret_cast<f32x4*>(base + (i * 3 + h) * 2048) = f32x4{v[2 * h].x, v[2 * h].y, v[2 * h + 1].x, v[2 * h + 1].y};
        }
#endif
    };

    // ---- MFMA side
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.d.weight), 0, (int)((long)a.n_c8 * a.n_cg * 18 * 256 * 4), 0x00020000);
    const unsigned wvoff = (unsigned)(lane * 16);
    auto wblock = [&](int c8_, int cg_) { return __builtin_amdgcn_readfirstlane(((c8_ * a.n_cg + cg_) * 18) * 1024); };

    f32x4 acc[TGW * NPOS];                                               // [position][tile group j]: TGW * pos + j
    f32x4 U[UR];                                                         // ring: fragment pp of a stage (positions 2pp, 2pp+1) lives in U[(OFF + pp) % UR]
    f32x4 Vr[W4_VR][TGW];                                                // ring: V of position pair pp, one per tile group, in Vr[pp % W4_VR]
    static_assert(36 % UR == 0, "the ring must close over a chunk's two stages of 18 fragments");
    auto load_u = [&](int slot, int q, int wb) {                         // q in [0, 18): position pair
#if !(W4_ABLATE & 2)
        U[slot % UR] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, wb + q * 1024, W4_U_AUX));
#endif
    };
    auto read_v = [&](const char* v0base, const char* v1base, int pp) {
        Vr[pp % W4_VR][0] = *reinterpret_cast<const f32x4*>(v0base + pp * 2048);
        if (TGW == 2) Vr[pp % W4_VR][TGW - 1] = *reinterpret_cast<const f32x4*>(v1base + pp * 2048);
    };
    auto mfma = [&](auto first_c, int idx, float av, float bv) {
        constexpr bool FIRST = decltype(first_c)::value;
        if (idx < ACC_AGPR) { if (FIRST) W4_MFMA_AZ(acc[idx], av, bv); else W4_MFMA_A(acc[idx], av, bv); }
        else                { if (FIRST) W4_MFMA_VZ(acc[idx], av, bv); else W4_MFMA_V(acc[idx], av, bv); }
    };
    // one 8-channel stage: 18 position pairs x 8 MFMAs.  V operands come from the two tile groups' current images (+ half g2);
    // weight fragments of this stage from block `wb`, the ring is refilled from the next stage's `nb` once this stage's 18 are
    // requested.  `mid(pp)` runs behind position pair pp (halo loads of stage 0, their LDS writes in stage 1).
    // `off_c`: ring slot of the stage's fragment 0 (stage 0 of a chunk: 0, stage 1: 18 % UR; a chunk's 36 fragments close the ring).
    // `din_c` fragments of this stage are in flight on entry, `dout_c` of the next stage on exit (UR in the steady state).
    auto stage = [&](auto first_c, auto off_c, auto din_c, auto dout_c, const char* v0base, const char* v1base, int wb, int nb, auto&& pre, auto&& mid) {
        constexpr int OFF = decltype(off_c)::value, DIN = decltype(din_c)::value, DOUT = decltype(dout_c)::value;
        auto hi = [](int pp) { const int h = pp + UR; return h < 18 + DOUT ? h : 18 + DOUT; };   // fragments requested before position pair pp
#pragma unroll
        for (int pp = 0; pp < W4_VR - 1; ++pp) read_v(v0base, v1base, pp);
#pragma unroll
        for (int q = DIN; q < hi(0); ++q) {                              // (only behind an epilogue: the burst that refills the ring)
            if (q < 18) load_u(OFF + q, q, wb);
            else load_u(OFF + q, q - 18, nb);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pp = 0; pp < 18; ++pp) {
            if (pp + W4_VR - 1 < 18) read_v(v0base, v1base, pp + W4_VR - 1);
            pre(pp);                                                     // LDS table reads of mid(pp): their latency hides under the MFMAs
            const f32x4 u = U[(OFF + pp) % UR];
            const f32x4 va = Vr[pp % W4_VR][0], vb = Vr[pp % W4_VR][TGW - 1];     // {pos 2pp: ch even, odd; pos 2pp+1: ch even, odd}
            __builtin_amdgcn_sched_barrier(0);
            // even channels of the pair first (the first touch of every accumulator in a tile's first stage), then the odd ones:
            // an accumulator is used again four (TGW == 1: two) MFMAs later (dependent latency 40 cycles, issue 32).  TGW == 2: each weight
            // fragment serves both tile groups: it is loaded once per workgroup.
            if (TGW == 2) {
                mfma(first_c, 4 * pp + 0, u.x, va.x);
                mfma(first_c, 4 * pp + 1, u.x, vb.x);
                mfma(first_c, 4 * pp + 2, u.z, va.z);
                mfma(first_c, 4 * pp + 3, u.z, vb.z);
                mfma(std::false_type{}, 4 * pp + 0, u.y, va.y);
                mfma(std::false_type{}, 4 * pp + 1, u.y, vb.y);
                mfma(std::false_type{}, 4 * pp + 2, u.w, va.w);
                mfma(std::false_type{}, 4 * pp + 3, u.w, vb.w);
            } else {
                mfma(first_c, 2 * pp + 0, u.x, va.x);
                mfma(first_c, 2 * pp + 1, u.z, va.z);
                mfma(std::false_type{}, 2 * pp + 0, u.y, va.y);
                mfma(std::false_type{}, 2 * pp + 1, u.w, va.w);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = (hi(pp) > DIN ? hi(pp) : DIN); q < hi(pp + 1); ++q) {     // the slot just consumed takes the fragment UR ahead
                if (q < 18) load_u(OFF + q, q, wb);
                else load_u(OFF + q, q - 18, nb);
            }
            mid(pp);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // accumulators are read in program order through volatile asm: left to itself hipcc hoists ~200 v_accvgpr_read to the top of
    // the epilogue and spills what they produce
    auto read_acc2 = [&](int idx, int h) -> f32x2 {                     // registers (2 h, 2 h + 1) of an accumulator
        if (idx >= ACC_AGPR) {
            // an accumulator in ordinary registers: the empty volatile asm keeps hipcc from scheduling its (plain VALU) readers above
            // W4_MFMA_DRAIN -- it does not know the asm statements that produced it are MFMAs still in flight
            asm volatile("" : "+v"(acc[idx]));
            return h ? f32x2{acc[idx].z, acc[idx].w} : f32x2{acc[idx].x, acc[idx].y};
        }
        f32x2 r;
        if (h == 0) {
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r.x) : "a"(acc[idx].x));
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r.y) : "a"(acc[idx].y));
        } else {
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r.x) : "a"(acc[idx].z));
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r.y) : "a"(acc[idx].w));
        }
        return r;
    };

    // ---- prologue: first item staged and transformed synchronously, weight ring primed
    int b, ty, rx, nt, sp;
    decode(t_begin, b, ty, rx, nt, sp);
    int b1 = b, ty1 = ty, rx1 = rx, nt1 = nt, sp1 = sp;
    stage_tile(b, ty, rx);
    stage_issue_begin(SPLIT ? sp * n_chunks * KC4 : 0);
#pragma unroll
    for (int k = 0; k < RAW_IT; ++k) stage_issue_one(k, ptab[k * NT], std::true_type{});
    {
        const int wb = wblock(SPLIT ? 2 * sp * n_chunks : 0, nt * 4 + (wave & 3));
#pragma unroll
        for (int q = 0; q < UR_EPI; ++q) load_u(q, q, wb);
    }
#pragma unroll
    for (int k = 0; k < RAW_IT; ++k) {
        if (GEN && k % GEN_H == 0) gen_maps(k);
        stage_commit_one(k, dtab[k * NT], AFF ? ptab[k * NT] : 0u);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (TGW == 2) {
        f32x2 T[6][6];
        unsigned t_addr[12];
        xf_addr(t_addr);
#pragma unroll
        for (int ay = 0; ay < 6; ++ay) xf_read(T, t_addr, ay);
        xf_finish(T, vd_tg, [] {});
    } else xf_half(vd_tg);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

#ifdef W4_STAMP
    stamp_pro = __builtin_amdgcn_s_memtime() - stamp_c0;
#endif
    for (int t = t_begin; t < t_end; ++t) {
        const bool more = t + 1 < t_end;
        W4_T0();
        if (more) {                                                      // the next item: decode(t + 1) without the divisions
            nt1 = nt + 1;  rx1 = rx;  ty1 = ty;  b1 = b;  sp1 = sp;
            if (nt1 == a.n_tiles) { nt1 = 0;  if (++rx1 == a.regions_x) { rx1 = 0;  if (++ty1 == a.tiles_y) { ty1 = 0;  ++b1;  if (SPLIT && b1 == a.d.B) { b1 = 0;  ++sp1; } } } }
        }
        const int ch0 = SPLIT ? sp * n_chunks : 0, ch0n = SPLIT ? sp1 * n_chunks : 0;      // first chunk of this item's / the next item's K range
        const int cg = nt * 4 + (wave & 3), cg_next = (more ? nt1 : nt) * 4 + (wave & 3);          // this wave's 16 output channels

        auto chunk = [&](int ch, auto first_c, auto last_c) {
            const char* v0cur = reinterpret_cast<const char*>(Vd + mtg * VD_FLOATS) + d_lds;                     // tile group 0 / 1 (TGW == 1: the wave's own): V images
            const char* v1cur = reinterpret_cast<const char*>(Vd + (NTG - 1) * VD_FLOATS) + d_lds;
            constexpr bool last = decltype(last_c)::value;               // last chunk of the tile (n_chunks >= 2: never also the first)
            const int c8 = 2 * (ch0 + ch);
            // weight blocks: this chunk's two stages, then the next item's first stage (after the very last item: a harmless reload)
            const int w0 = wblock(c8, cg), w1 = wblock(c8 + 1, cg), wn = last ? wblock(2 * ch0n, cg_next) : wblock(c8 + 2, cg);
#ifdef W4_STAMP
            if (decltype(first_c)::value) W4_ACC(stamp_top);
            W4_T0();
#endif
            if (last && more) stage_tile(b1, ty1, rx1);                  // from here on the next tile is staged
#ifdef W4_STAMP
            if (last) W4_ACC(stamp_tile);
#endif
            using I0 = std::integral_constant<int, 0>;
            using IOFF1 = std::integral_constant<int, 18 % UR>;
            using IUR = std::integral_constant<int, UR>;
            using IEPI = std::integral_constant<int, UR_EPI>;
            using IXF = std::integral_constant<int, TGW == 1 ? (W4_UR1_XF < UR ? W4_UR1_XF : UR) : UR>;      // fragments in flight across the transform between two chunks
            constexpr bool FIRST = decltype(first_c)::value;              // first chunk of a tile: the ring comes out of an epilogue
            // the next item's halo: requested over the first position pairs of stage 0, written to the raw image over stage 1
            // (unconditional: behind the very last item this is a harmless re-stage of the tile's first chunk -- a conditional load /
            // store pair would keep the staging registers alive everywhere)
            unsigned tab1 = 0;                                            // the table entry of position pair pp's item, read ahead of its MFMAs
            constexpr bool PRE_P = AFF && !MAPX;                           // (the map variant has no registers to spare: it reads ptab inside the clump)
            unsigned tabd[(AFF || LEAKY) ? RAW_IT : 1], tabp[PRE_P ? RAW_IT : 1];   // AFF / LEAKY: the clump's ten items
            auto issue_pre = [&](int pp) { if (pp < RAW_IT) tab1 = ptab[pp * NT]; };
            auto issue = [&](int pp) {
                if (pp == 0) stage_issue_begin(last ? ch0n * KC4 : (ch0 + ch + 1) * KC4);
                if (pp < RAW_IT) stage_issue_one(pp, tab1, last_c);      // (a tile's last chunk stages the next tile's first)
            };
            auto commit_pre = [&](int pp) {
                if (AFF || LEAKY) {
                    if (pp == W4_COMMIT_AT) {
#pragma unroll
                        for (int k = 0; k < RAW_IT; ++k) {
                            tabd[k] = dtab[k * NT];
                            if (PRE_P) tabp[k] = ptab[k * NT];
                        }
                    }
                } else if (pp < RAW_IT) tab1 = dtab[pp * NT];
            };
            f32x2 T[TGW == 2 ? 6 : 1][6];                                 // TGW == 2: the next item's patch of this transform lane: read under the tail of stage 1
            unsigned t_addr[12];
            auto commit = [&](int pp) {
                if (AFF || LEAKY) {     // the activation is VALU work: one clump (every MFMA <-> VALU switch costs ~18 cycles)
                    if (pp == W4_COMMIT_AT) {
#pragma unroll
                        for (int k = 0; k < RAW_IT; ++k) {
                            if (GEN && k % GEN_H == 0) gen_maps(k);
                            stage_commit_one(k, tabd[(AFF || LEAKY) ? k : 0], PRE_P ? tabp[PRE_P ? k : 0] : MAPX ? ptab[k * NT] : 0u);
                        }
                    }
                } else if (pp < RAW_IT) stage_commit_one(pp, tab1, 0u);
                if constexpr (TGW == 2) {
                    if (pp == W4_XF_AT) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // this thread's share of the raw image is written ...
                        __builtin_amdgcn_s_barrier();                        // ... and so is every other wave's
                        xf_addr(t_addr);
                    }
                    if (pp > W4_XF_AT && pp <= W4_XF_AT + 6) xf_read(T, t_addr, pp - W4_XF_AT - 1);
                }
            };
            W4_T0();
            if (TGW == 1 && FIRST)                                        // this tile's bias (lane = cout): in flight over the whole first stage
                bias_r = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brsrc, (unsigned)(cg * 16 + (lane & 15)) * 4u, 0, 0));
            stage(first_c, I0{}, std::conditional_t<FIRST, IEPI, IXF>{}, IUR{}, v0cur, v1cur, w0, w1, issue_pre, issue);
            if (TGW == 1 && FIRST) asm volatile("" : "+v"(bias_r));      // waited for HERE (every older load has been consumed), not behind the ring in the epilogue
            stage(std::false_type{}, IOFF1{}, IUR{}, std::conditional_t<last, IEPI, IXF>{}, v0cur + 1024, v1cur + 1024, w1, wn, commit_pre, commit);
#ifdef W4_STAMP
            if (FIRST) W4_ACC(stamp_first);
            if (last) W4_ACC(stamp_last);
            if (!FIRST && !last && ch == 1) W4_ACC(stamp_second);
            if (!FIRST && !last && ch == 2) W4_ACC(stamp_third);
#endif
            W4_T0();
            if constexpr (TGW == 2) {
                xf_finish(T, vd_tg, [&] {
                    W4_ACC(stamp_xf);
                    W4_T0();
                    __builtin_amdgcn_s_barrier();                        // every wave has read its last V operands of this chunk
                    W4_ACC(stamp_wait);
                    W4_T0();
                });
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // this thread's share of the raw image is written ...
                __builtin_amdgcn_s_barrier();                            // ... so is every other wave's, and every wave has read its last V operands of this chunk
                W4_ACC(stamp_wait);
                W4_T0();
                if (W4_PRIO) __builtin_amdgcn_s_setprio(W4_PRIO);
                xf_half(vd_tg);
                if (W4_PRIO) __builtin_amdgcn_s_setprio(0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                // the next item's V images are complete
            W4_ACC(stamp_xf);
        };
        // three instances of the chunk body, no branch between alternatives (accumulators pinned by asm constraints do not survive an
        // if / else of two instances without copies): first (ring refilled behind the epilogue), middle, last (ring runs down)
        chunk(0, std::true_type{}, std::false_type{});
        for (int ch = 1; ch + 1 < n_chunks; ++ch) chunk(ch, std::false_type{}, std::false_type{});
        chunk(n_chunks - 1, std::false_type{}, std::true_type{});

        // ---- output transform Y = A^T M A, bias, stores, GN partials.  The MFMA operands are A = V (rows = the tile group's 16 tiles),
        //      B = U (columns = the wave's 16 couts): lane (cout = l & 15, kq = l >> 4) holds in the four registers of an accumulator
        //      the four tiles 4 kq .. 4 kq + 3 = the four 4x4-pixel blocks of tile-row kq, i.e. pixels x = 4 r + jj (r = register, jj < 4),
        //      y = 4 kq + i of the 16x16 tile, for ONE cout.  The float4 arithmetic below runs over those four tiles, and a store of
        //      register r is a dword per lane with 16 consecutive lanes = 16 consecutive couts = 64 contiguous bytes: the CU's store
        //      path takes such a wave store in ~4 requests (58 B/clk).  With the roles the other way round (r2: lane = four couts of
        //      one tile, 16-byte stores) consecutive lanes hit different pixels, every lane is a request of its own and the path runs
        //      at 16 B/clk -- 8 k cycles per 128 KB region tile (tools/microbench/store_patterns.hip).
        W4_T0();
        W4_MFMA_DRAIN();
        if (TGW == 1 && W4_PRIO) __builtin_amdgcn_s_setprio(W4_PRIO);

        const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.d.out, 0, (int)((unsigned)(SPLIT ? a.splits : 1) * a.d.B * H * W * a.d.ldo * 4u), 0x00020000);
        int Wt = __builtin_amdgcn_readfirstlane(W), ldot = __builtin_amdgcn_readfirstlane(a.d.ldo);
        asm volatile("" : "+s"(Wt), "+s"(ldot));                         // per tile: keeps the store offsets from being hoisted into (spilled) SGPRs
#pragma unroll
        for (int j = 0; j < TGW; ++j) {                                  // the 16x16-pixel tiles (tile groups) this wave holds
            const int tx = NTG * rx + (TGW == 2 ? j : mtg);
            if (tx < a.tiles_x) {
                int l15 = lane;
                asm volatile("" : "+v"(l15));                              // (lane-derived values are recomputed per tile: hoisted to the kernel's start they get spilled)
                l15 &= 15;
                const int co = cg * 16 + l15;                              // this lane's cout
                const bool cok = co < Cout;
                float z0 = 0.0f;
                asm volatile("" : "+v"(z0));                               // a fresh zero per tile: hipcc otherwise keeps one zero float4 alive (and spilled) for the whole kernel
                const float* const bias_p = bias_lds + co;                 // read where it is used (an LDS load: held in registers it gets spilled)
                const int py0 = ty * 16 + 4 * kq, px0 = tx * 16;           // the lane's tile row; register r covers columns px0 + 4 r .. + 3
                f32x2 sum2 = {z0, z0}, sq2 = {z0, z0}, cnt2 = {z0, z0};
                float pivot = z0;
                const bool full = ty * 16 + 16 <= H && tx * 16 + 16 <= W && cg * 16 + 16 <= Cout;      // wave-uniform
                const bool want_stats = a.d.stats != nullptr;
                // stores: buffer addressing -- one 32-bit lane offset per tile, the pixel's offset (uniform) in the scalar offset
                // field (the output stays below 4 GiB, host check)
                const unsigned lane_off = (unsigned)((((unsigned)(SPLIT ? sp * a.d.B + b : b) * H + py0) * Wt + px0) * ldot + co) * 4u;   // (SPLIT: the split's partial tensor)
                // two of the lane's four tiles at a time (Z[4][6] of float2 = 48 registers): every accumulator register is read ONCE
                // (v_accvgpr_read_b32 issues every 8 cycles: 288 instead of the 480 of a float4 pass over two output rows at a time)
                auto emit = [&](auto full_c, auto stats_c) {
                    constexpr bool FULL = decltype(full_c)::value, STATS = decltype(stats_c)::value;
                    int step4 = ldot * 16, back3 = ldot * -12, rowadv = (Wt - 7) * ldot * 4;   // pixel (i, 4 r + jj): r -> r + 1, (jj, r + 1) -> (jj + 1, r), next row
                    asm volatile("" : "+s"(step4), "+s"(back3), "+s"(rowadv));
                    const f32x2 c2 = {2.0f, 2.0f}, c4 = {4.0f, 4.0f}, c8 = {8.0f, 8.0f};
                    const int rowlim = FULL ? 4 : H - py0, collim = FULL ? 16 : W - px0;     // !FULL: rows / columns of the lane's tile row inside the image
#pragma unroll
                    for (int h = 0; h < 2; ++h) {                        // tiles (2 h, 2 h + 1) of the lane's tile row = registers 2 h, 2 h + 1 of every accumulator
                        int soff = h * 32 * ldot;                        // byte offset of pixel (0, 8 h): one running scalar, see below
                        asm volatile("" : "+s"(soff));
                        f32x2 Z[4][6];
#pragma unroll
                        for (int nu = 0; nu < 6; ++nu) {                 // Z = A^T M, one column of positions at a time: short live ranges
                            auto M = [&](int xi) { return read_acc2(TGW * (xi * 6 + nu) + j, h); };
                            f32x2 m1 = M(1);
                            const f32x2 m2 = M(2);
                            if (nu == 1) m1 += Geo::BIAS_LDS ? *bias_p : bias_r;   // a constant on all 16 outputs of a tile == that constant on position (1, 1): A^T e1 = (1, 1, 1, 1)
                            const f32x2 p = m1 + m2, q = m1 - m2;
                            const f32x2 m3 = M(3), m4 = M(4);
                            const f32x2 r = m3 + m4, u = m3 - m4;
                            Z[0][nu] = M(0) + p + r;
                            Z[1][nu] = __builtin_elementwise_fma(c2, u, q);
                            Z[2][nu] = __builtin_elementwise_fma(c4, r, p);
                            Z[3][nu] = __builtin_elementwise_fma(c8, u, q) + M(5);
                            __builtin_amdgcn_sched_barrier(0);
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) {                    // Y = Z A, one output row at a time
                            const f32x2 (&z)[6] = Z[i];
                            const f32x2 ta = z[1] + z[2], tb = z[1] - z[2], tc = z[3] + z[4], te = z[3] - z[4];
                            f32x2 y[4];
                            y[0] = z[0] + ta + tc;
                            y[1] = __builtin_elementwise_fma(c2, te, tb);
                            y[2] = __builtin_elementwise_fma(c4, tc, ta);
                            y[3] = __builtin_elementwise_fma(c8, te, tb) + z[5];
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) {
                                const f32x2 v = y[jj];                   // pixels (4 kq + i, 4 (2 h + e) + jj) of the tile, e = component
                                if (STATS && h == 0 && i == 0 && jj == 0)   // one pivot per cout for the whole 16x16 tile: its first pixel (lane l & 15, register 0)
                                    pivot = __shfl(v.x, l15);
                                f32x2 in2 = {1.0f, 1.0f};                // !FULL: which of the two pixels are inside the image
                                if (!FULL) {
                                    // (the row limit is made opaque at every use: left visible, hipcc computes the 64 (row, column) lane masks of a tile up front,
                                    //  128 SGPRs that push the kernel's long-lived scalars into VGPR lanes -- v_readlane in every K chunk)
                                    int rl = rowlim;
                                    asm volatile("" : "+v"(rl));
                                    const bool row_in = i < rl;
                                    in2.x = (row_in && 8 * h + jj < collim) ? 1.0f : 0.0f;
                                    in2.y = (row_in && 8 * h + 4 + jj < collim) ? 1.0f : 0.0f;
                                }
                                if (STATS) {
                                    f32x2 dv = v - f32x2{pivot, pivot};
                                    if (!FULL) { dv *= in2;  cnt2 += in2; }
                                    sum2 += dv;
                                    sq2 += dv * dv;
                                }
#if !(W4_ABLATE & 8)
                                // EVERY path issues the same store instructions: a pixel / cout outside the tensor gets an offset beyond the
                                // resource (the store is dropped by the range check) instead of a branch around the store.  vmcnt counts loads
                                // and stores in one in-order queue; with a store-free path through the epilogue hipcc sizes the next tile's first
                                // weight-fragment waits as if NO store were in flight.
#pragma unroll
                                for (int e = 0; e < 2; ++e) {
                                    const float vr = e ? v.y : v.x;      // (by value: hipcc 7.2 evaluates __builtin_bit_cast(unsigned, v[e]) on an element
                                                                         //  reference as element 0 for every e -- stores of the same register)
                                    unsigned off = lane_off;
                                    if (!FULL) off = ((e ? in2.y : in2.x) != 0.0f && cok) ? off : 0xFFFFFFF0u;
                                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vr), orsrc, off, soff, STREAM ? W4_STORE_AUX : 0);
                                    // the pixel's offset ((i W + 4 r + jj) ldo 4 bytes, uniform) is ONE running scalar, advanced by a scalar add behind
                                    // every store.  Written as 64 expressions of W and ldo, hipcc computes them all at the kernel's start, spills them
                                    // to VGPR lanes and reloads one with v_readlane_b32 (+ the VALU-writes-SGPR -> VMEM wait states) in front of every store
                                    asm volatile("s_add_i32 %0, %0, %1" : "+s"(soff) : "s"(e == 0 ? step4 : jj < 3 ? back3 : rowadv) : "scc");
                                }
#else
                                asm volatile("" :: "v"(v));              // (the output transform stays: only the stores are gone)
#endif
                            }
                        }
                    }
                };
                if (full) { if (want_stats) emit(std::true_type{}, std::true_type{}); else emit(std::true_type{}, std::false_type{}); }
                else { if (want_stats) emit(std::false_type{}, std::true_type{}); else emit(std::false_type{}, std::false_type{}); }
                if (a.d.stats) {
                    // pool the lane's four tiles, then the four tile rows (lanes l, l + 16, l + 32, l + 48 share the cout): sum = S + n p, M2 = Q - S^2 / n
                    float fc = full ? 64.0f : cnt2.x + cnt2.y;
                    float S = sum2.x + sum2.y, Q = sq2.x + sq2.y;
                    fc += __shfl_xor(fc, 16);  S += __shfl_xor(S, 16);  Q += __shfl_xor(Q, 16);
                    fc += __shfl_xor(fc, 32);  S += __shfl_xor(S, 32);  Q += __shfl_xor(Q, 32);
                    const int slot = ty * a.tiles_x + tx;             // one statistics slot per 16 x 16 tile (nd_conv3x3_wino4_stat_slots)
                    if (kq == 0 && cok) {
                        fc = fmaxf(fc, 1.0f);
                        float* o = a.d.stats + (((size_t)b * a.slots + slot) * Cout + co) * 2;
                        *reinterpret_cast<f32x2*>(o) = f32x2{S + fc * pivot, fmaxf(Q - S * S / fc, 0.0f)};
                    }
#ifndef W4_STAMP
                    if (b == 0 && nt == 0 && (wave & 3) == 0 && lane == 0) {
                        a.d.slot_count[slot] = (float)(min(16, H - ty * 16) * min(16, W - tx * 16));
                    }
#endif
                }
            }
        }
        if (TGW == 1 && W4_PRIO) __builtin_amdgcn_s_setprio(0);
        W4_ACC(stamp_epi);
#ifdef W4_STAMP_DRAIN          // diagnostic: how long do the epilogue's stores (and the weight fragments in flight) take to complete
        W4_T0();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        W4_ACC(stamp_drain);
#endif
        b = b1; ty = ty1; rx = rx1; nt = nt1; sp = sp1;
    }
#ifdef W4_STAMP
    if (tid == 0) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(a.d.slot_count) + 16 * blockIdx.x;
        dbg[0] = __builtin_amdgcn_s_memtime() - stamp_c0;
        dbg[1] = __builtin_amdgcn_s_memrealtime() - stamp_r0;
        dbg[2] = (unsigned long long)(t_end - t_begin) * n_chunks;
        dbg[3] = stamp_epi;
        dbg[4] = stamp_xf;
        dbg[5] = stamp_second;      // ... second chunks
        dbg[6] = stamp_first;       // stage time of the tiles' first chunks (behind an epilogue) ...
        dbg[7] = stamp_last;        // ... and of their last chunks
        dbg[8] = stamp_third;
        dbg[9] = stamp_wait;        // barrier in front of the transform
        dbg[10] = stamp_pro;        // stagger + prologue
        dbg[11] = stamp_top;        // loop top: decode of the next tile
        dbg[12] = stamp_tile;       // stage_tile
    }
#endif
}

// ---- split-K: the reduction kernels and the launcher of the SPLIT instances
// out[n][c] = sum over the splits (in split order) of part[s][n][c] + bias[c]; n = pixel (B * H * W), float4 per thread
__global__ __launch_bounds__(256) void w4_splitk_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias, float* __restrict__ out,
                                                               int splits, long npix, int cout, int ldo) {
    const int cq = cout >> 2;
    const long total = npix * cq;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long n = i / cq;
        const int c = (int)(i - n * cq) * 4;
        f32x4 acc = nd_ld4(part + n * cout + c);
        for (int s = 1; s < splits; ++s) acc += nd_ld4(part + ((long)s * npix + n) * cout + c);
        if (bias) acc += nd_ld4(bias + c);
        nd_st4(out + n * ldo + c, acc);
    }
}

// The same reduction for layers with a statistics epilogue (Block.proj: the GroupNorm that follows pools them): one workgroup per (sample, 16 x 16-pixel
// tile) -- the tile is one statistics slot of nd_conv3x3_wino4_stat_slots -- adds the partial tensors' tile (+ bias), stores it, and leaves the slot's
// per-channel {sum, M2 about the tile's mean} in `stats` exactly as wino4_kernel's own epilogue does (pivot = the tile's first pixel; M2 = Q - S^2 / n).
// Thread = (channel quad, pixel row group); the row groups meet through LDS in a fixed order.
__global__ __launch_bounds__(256) void w4_splitk_reduce_stats_kernel(const float* __restrict__ part, const float* __restrict__ bias, float* __restrict__ out,
                                                                     float* __restrict__ stats, float* __restrict__ slot_count, int splits, int B, int H, int W,
                                                                     int cout, int ldo, int tiles_x, int tiles_y, int n_cgrp) {
    __shared__ __attribute__((aligned(16))) float red[2][256][4];
    int bid = blockIdx.x;
    const int cgrp = bid % n_cgrp;  bid /= n_cgrp;                 // 64 couts (16 channel quads) per workgroup: small images still give the chip work
    const int slot = bid % (tiles_x * tiles_y), b = bid / (tiles_x * tiles_y);
    const int ty = slot / tiles_x, tx = slot % tiles_x;
    const int y0 = ty * 16, x0 = tx * 16, th = min(16, H - y0), tw = min(16, W - x0), npx = th * tw;
    const long npix = (long)B * H * W;
    const int q = cgrp * 16 + (threadIdx.x & 15), rg = threadIdx.x >> 4;      // channel quad, pixel row group (16 of them)
    const bool act = 4 * q < cout;
    f32x4 s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0}, pv = {0, 0, 0, 0};
    if (act) {
        const f32x4 bq = bias ? nd_ld4(bias + 4 * q) : f32x4{0, 0, 0, 0};
        {   // pivot: the tile's first pixel, summed the same way
            const long n0 = ((long)b * H + y0) * W + x0;
            pv = nd_ld4(part + n0 * cout + 4 * q);
            for (int sp = 1; sp < splits; ++sp) pv += nd_ld4(part + ((long)sp * npix + n0) * cout + 4 * q);
            pv += bq;
        }
        for (int p = rg; p < npx; p += 16) {
            const int py = p / tw, px = p - py * tw;
            const long n = ((long)b * H + y0 + py) * W + x0 + px;
            f32x4 v = nd_ld4(part + n * cout + 4 * q);
            for (int sp = 1; sp < splits; ++sp) v += nd_ld4(part + ((long)sp * npix + n) * cout + 4 * q);
            v += bq;
            nd_st4(out + n * ldo + 4 * q, v);
            const f32x4 dv = v - pv;
            s1 += dv;  s2 += dv * dv;
        }
    }
    *reinterpret_cast<f32x4*>(red[0][threadIdx.x]) = s1;
    *reinterpret_cast<f32x4*>(red[1][threadIdx.x]) = s2;
    __syncthreads();
    if (act && rg == 0) {
        for (int r = 1; r < 16; ++r) {
            s1 += *reinterpret_cast<const f32x4*>(red[0][r * 16 + (threadIdx.x & 15)]);
            s2 += *reinterpret_cast<const f32x4*>(red[1][r * 16 + (threadIdx.x & 15)]);
        }
        const float fn = (float)npx;
        float* o = stats + (((size_t)b * tiles_x * tiles_y + slot) * cout + 4 * q) * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[2 * i] = s1[i] + fn * pv[i];
            o[2 * i + 1] = fmaxf(s2[i] - s1[i] * s1[i] / fn, 0.0f);
        }
    }
    if (b == 0 && cgrp == 0 && threadIdx.x == 0) slot_count[slot] = (float)npx;
}

template <int MODE, int NTG = 2>
int launch4_split(const Wino4Args& a, hipStream_t st) {
    static nd_device_once configured;
    constexpr int LDS_BYTES = W4Geo<NTG>::LDS_BYTES;
    if (int e = nd_reserve_lds(configured, reinterpret_cast<const void*>(wino4_kernel<MODE, false, true, NTG>), LDS_BYTES, "nd_conv3x3_wino4_splitk")) return e;
    const long resident = (long)nd_device_cus() * W4Geo<NTG>::WG_PER_CU;
    hipLaunchKernelGGL((wino4_kernel<MODE, false, true, NTG>), dim3((unsigned)(a.total_wg < resident ? a.total_wg : resident)), dim3(256), LDS_BYTES, st, a);
    return 0;
}

// OIHW (cout, cin, 3, 3) -> U = G g G^T in blocks [cin/8][cout/16][18 position pairs][64 lanes][4]:
// lane (cout = l & 15, k = l >> 4) holds {U[2pp][ch 2k], U[2pp][ch 2k+1], U[2pp+1][ch 2k], U[2pp+1][ch 2k+1]} of its block
// dgrad: `w` is the FORWARD layer's OIHW weight (cin_fwd = cout, cout_fwd = cin) and the packed operator is the data gradient's --
// taps flipped, channel roles swapped: g'[co][ch][r][s] = w[ch][co][2 - r][2 - s] -- read in place (no flipped / transposed copy).
template <bool DGRAD>
__device__ __forceinline__ void pack_wino4_body(const float* __restrict__ w, float* __restrict__ out, int cin, int cout, int n_c8, int n_cg) {
    // One thread = one lane slot (block (c8, cg), lane l): the two channels' nine taps are read once and all 18 position pairs leave as
    // float4s -- 64 lanes x 16 bytes contiguous per store.  (One thread per output float re-read every 3 x 3 filter 36 times from
    // addresses a whole filter row apart: 150 us for a 512 -> 512 layer, 1.3 ms of a training step's 98 packings.)
    constexpr float G[6][3] = {{0.25f, 0.f, 0.f}, {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                               {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
    const size_t total = (size_t)n_c8 * n_cg * 64;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int l = (int)(i & 63);
        const size_t blk = i >> 6;
        const int cg = (int)(blk % n_cg), c8 = (int)(blk / n_cg);
        const int co = cg * 16 + (l & 15), ch0 = c8 * 8 + 2 * (l >> 4);
        float g[2][9];
        bool ok[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int ch = ch0 + c;
            ok[c] = ch < cin && co < cout;
            const float* gp = w + (ok[c] ? (DGRAD ? (size_t)ch * cout + co : (size_t)co * cin + ch) * 9 : 0);
#pragma unroll
            for (int j = 0; j < 9; ++j) g[c][j] = gp[DGRAD ? 8 - j : j];
        }
        float* o = out + blk * (18 * 256) + l * 4;
#pragma unroll
        for (int pp = 0; pp < 18; ++pp) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int pos = 2 * pp + (e >> 1), xi = pos / 6, nu = pos % 6, c = e & 1;
                // fp64 accumulation of the 9 products: the packed weights are exact-rounded once
                double acc = 0.0;
#pragma unroll
                for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                    for (int ss = 0; ss < 3; ++ss) acc += (double)G[xi][rr] * (double)g[c][rr * 3 + ss] * (double)G[nu][ss];
                v[e] = ok[c] ? (float)acc : 0.0f;
            }
            nd_st4(o + pp * 256, v);
        }
    }
}

template <bool DGRAD>
__global__ void pack_wino4_kernel(const float* __restrict__ w, float* __restrict__ out, int cin, int cout, int n_c8, int n_cg) {
    pack_wino4_body<DGRAD>(w, out, cin, cout, n_c8, n_cg);
}

// The same packing for MANY weights in one launch (training: every Block.proj weight and its data-gradient packing once per optimizer step --
// 98 launches of 4-17 microseconds at d = 64 otherwise).  blockIdx.y = item; `items` lives in device memory; item.transposed = data-gradient form.
__global__ void pack_wino4_batch_kernel(const nd_pack_item* __restrict__ items) {
    const nd_pack_item it = items[blockIdx.y];
    const int n_c8 = (((it.cin + 7) / 8) + 1) / 2 * 2, n_cg = (it.cout + 63) / 64 * 4;
    if (it.transposed) pack_wino4_body<true>(it.w, it.packed, it.cin, it.cout, n_c8, n_cg);
    else pack_wino4_body<false>(it.w, it.packed, it.cin, it.cout, n_c8, n_cg);
}

template <int MODE, bool STREAM, int NTG = 2, int NW = 4>
int launch4s(const Wino4Args& a, hipStream_t st) {
    static nd_device_once configured;
    constexpr int LDS_BYTES = W4Geo<NTG, NW>::LDS_BYTES;
    if (int e = nd_reserve_lds(configured, reinterpret_cast<const void*>(wino4_kernel<MODE, STREAM, false, NTG, NW>), LDS_BYTES, "nd_conv3x3_wino4")) return e;
    const long resident = (long)nd_device_cus() * W4Geo<NTG, NW>::WG_PER_CU;  // one (NTG == 1: two) workgroup(s) per CU (registers, LDS)
    hipLaunchKernelGGL((wino4_kernel<MODE, STREAM, false, NTG, NW>), dim3((unsigned)(a.total_wg < resident ? a.total_wg : resident)), dim3(64 * NW), LDS_BYTES, st, a);
    return 0;
}

template <int MODE, int NTG = 2, int NW = 4>
int launch4(const Wino4Args& a, hipStream_t st) {
    static const long stream_min = (getenv("ND_W4_STREAM_MB") ? atol(getenv("ND_W4_STREAM_MB")) : 48) << 20;     // A/B knob (tools/ only)
    static const int stream_kinds = getenv("ND_W4_STREAM_KINDS") ? atoi(getenv("ND_W4_STREAM_KINDS")) : 7;           // A/B knob (tools/ only)
    const int kind = (MODE == ND_PRO_AFFINE_SILU || MODE == ND_PRO_AFFINE_MAP_SILU || MODE == ND_PRO_AFFINE_GENMAP_SILU) ? 2 : a.d.stats ? 1 : 4;   // block2 / block1 / resampling convs
    const long out_bytes = (long)a.d.B * a.d.H * a.d.W * a.d.ldo * 4;
    return (out_bytes >= stream_min && (stream_kinds & kind)) ? launch4s<MODE, true, NTG, NW>(a, st) : launch4s<MODE, false, NTG, NW>(a, st);
}

}  // namespace

extern "C" int64_t nd_pack_conv3x3_wino4_weight_floats(int cin, int cout) {
    return (int64_t)nd_round_up(nd_cdiv(cin, 8), 2) * nd_cdiv(nd_round_up(cout, 64), 16) * 18 * 256;     // whole 16-channel chunks
}

static int pack_wino4(const float* oihw, float* packed, int cin, int cout, int dgrad, void* stream) {
    ND_REQUIRE(oihw && packed, ND_E_BADARG, "nd_pack_conv3x3_wino4_weight: null pointer");
    ND_REQUIRE(cin > 0 && cout > 0, ND_E_BADARG, "nd_pack_conv3x3_wino4_weight: non-positive size");
    const int n_c8 = nd_round_up(nd_cdiv(cin, 8), 2), n_cg = nd_round_up(cout, 64) / 16;
    const size_t total = (size_t)n_c8 * n_cg * 64;                  // one thread per (block, lane): 72 outputs each
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (dgrad) hipLaunchKernelGGL(pack_wino4_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, oihw, packed, cin, cout, n_c8, n_cg);
    else hipLaunchKernelGGL(pack_wino4_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, oihw, packed, cin, cout, n_c8, n_cg);
    return nd_launch_status("nd_pack_conv3x3_wino4_weight");
}

extern "C" int nd_pack_conv3x3_wino4_weight(const float* oihw, float* packed, int cin, int cout, void* stream) {
    return pack_wino4(oihw, packed, cin, cout, 0, stream);
}

extern "C" int nd_pack_conv3x3_wino4_weight_dgrad(const float* oihw_fwd, float* packed, int cin, int cout, void* stream) {
    return pack_wino4(oihw_fwd, packed, cin, cout, 1, stream);
}

extern "C" int nd_pack_conv3x3_wino4_weights_batch(const nd_pack_item* items_dev, int n_items, void* stream) {
    ND_REQUIRE(items_dev && n_items > 0 && n_items <= 65535, ND_E_BADARG, "nd_pack_conv3x3_wino4_weights_batch: needs 1 .. 65535 items in device memory");
    hipLaunchKernelGGL(pack_wino4_batch_kernel, dim3(64, (unsigned)n_items), dim3(256), 0, (hipStream_t)stream, items_dev);
    return nd_launch_status("nd_pack_conv3x3_wino4_weights_batch");
}

extern "C" int nd_conv3x3_wino4_stat_slots(int H, int W) { return nd_cdiv(W, 16) * nd_cdiv(H, 16); }

// descriptor checks shared by the entry points; fills the launch arguments
static int w4_prepare(const nd_conv3x3* d, Wino4Args& a, int ntg = 2) {
    ND_REQUIRE(d, ND_E_BADARG, "nd_conv3x3_wino4: null descriptor");
    const nd_src& s = d->src;
    ND_REQUIRE(s.p0 && d->weight && d->out, ND_E_BADARG, "nd_conv3x3_wino4: null tensor pointer");
    ND_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->cin > 0 && d->cout > 0, ND_E_BADARG, "nd_conv3x3_wino4: non-positive size");
    ND_REQUIRE(d->cin % 4 == 0 && d->cout % 4 == 0 && d->cin > KC4 && d->cout <= MAX_COUT, ND_E_SHAPE,
               "nd_conv3x3_wino4: cin=%d and cout=%d must be multiples of 4, cin > 16 (at least two K chunks), cout <= %d", d->cin, d->cout, MAX_COUT);
    ND_REQUIRE(s.c0 + s.c1 == d->cin && s.c0 % 4 == 0 && s.c1 % 4 == 0 && s.c0 > 0, ND_E_SHAPE,
               "nd_conv3x3_wino4: source channels %d+%d do not match cin=%d (multiples of 4)", s.c0, s.c1, d->cin);
    ND_REQUIRE((s.c1 == 0) == (s.p1 == nullptr), ND_E_BADARG, "nd_conv3x3_wino4: p1/c1 mismatch");
    ND_REQUIRE(s.ld0 >= s.c0 && s.ld0 % 4 == 0 && (s.c1 == 0 || (s.ld1 >= s.c1 && s.ld1 % 4 == 0)), ND_E_ALIGN,
               "nd_conv3x3_wino4: pixel strides must be >= channels and multiples of 4");
    ND_REQUIRE(nd_aligned16(s.p0) && nd_aligned16(s.p1) && nd_aligned16(d->weight) && nd_aligned16(s.mad) && nd_aligned16(d->out) &&
               nd_aligned16(d->bias), ND_E_ALIGN, "nd_conv3x3_wino4: pointers must be 16-byte aligned");
    ND_REQUIRE(d->ldo >= d->cout && d->ldo % 4 == 0, ND_E_SHAPE, "nd_conv3x3_wino4: ldo must be >= cout and a multiple of 4");
    const bool aff = s.mode == ND_PRO_AFFINE_SILU || s.mode == ND_PRO_AFFINE_MAP_SILU || s.mode == ND_PRO_AFFINE_GENMAP_SILU;
    ND_REQUIRE(s.mode == ND_PRO_NONE || aff || s.mode == ND_PRO_LEAKY || s.mode == ND_PRO_LEAKY_SECOND, ND_E_BADARG,
               "nd_conv3x3_wino4: unsupported prologue %d", s.mode);
    ND_REQUIRE(s.mode != ND_PRO_AFFINE_GENMAP_SILU || (ntg == 2 && s.map && s.gamma && s.beta && !s.upsample && !s.map_blocked && s.c1 == 0 && s.c0 % KC4 == 0 &&
                                                     nd_aligned16(s.map) && nd_aligned16(s.gamma) && nd_aligned16(s.beta)), ND_E_BADARG,
               "nd_conv3x3_wino4: the in-kernel map prologue needs silu(pos_emb) (map), mlp[1].weight (gamma) and mlp[1].bias (beta), 16-byte aligned, one source of whole "
               "16-channel chunks, no upsample addressing, the 16 x 32-region form");
    ND_REQUIRE(!aff || s.mad, ND_E_BADARG, "nd_conv3x3_wino4: affine prologue needs mad");
    ND_REQUIRE(s.mode != ND_PRO_AFFINE_MAP_SILU || (s.map && !s.upsample && nd_aligned16(s.map)), ND_E_BADARG,
               "nd_conv3x3_wino4: the map prologue needs a 16-byte aligned map and no upsample addressing");
    ND_REQUIRE(s.mode != ND_PRO_AFFINE_MAP_SILU || (long)(d->B * (long)d->H + 2) * d->W * 2 * (s.c0 + s.c1) * 4 < (1L << 30) - 65536, ND_E_SHAPE,
               "nd_conv3x3_wino4: a scale / shift map of 1 GiB or more");
    ND_REQUIRE(s.mode != ND_PRO_LEAKY_SECOND || s.p1, ND_E_BADARG, "nd_conv3x3_wino4: LEAKY_SECOND needs a second source");
    ND_REQUIRE(!s.map_blocked || (s.mode == ND_PRO_AFFINE_MAP_SILU && (s.c0 + s.c1) % 16 == 0), ND_E_BADARG,
               "nd_conv3x3_wino4: map_blocked needs the map prologue and a channel count that is a multiple of 16");
    ND_REQUIRE(!s.unshuffle, ND_E_BADARG, "nd_conv3x3_wino4: no unshuffle addressing");
    ND_REQUIRE(!s.upsample || (d->H % 2 == 0 && d->W % 2 == 0 && s.c1 == 0), ND_E_SHAPE,
               "nd_conv3x3_wino4: nearest-x2 upsample addressing needs even H, W and a single source");
    ND_REQUIRE((d->stats == nullptr) == (d->slot_count == nullptr), ND_E_BADARG, "nd_conv3x3_wino4: stats and slot_count go together");
    ND_REQUIRE(s.c1 == 0 || s.c0 % KC4 == 0, ND_E_SHAPE,
               "nd_conv3x3_wino4: first concat source has %d channels; a 16-channel K chunk must not straddle the sources", s.c0);
    ND_REQUIRE(d->W <= 2048 && d->H <= 32768, ND_E_SHAPE, "nd_conv3x3_wino4: image wider than 2048 (16-bit border table)");
    {
        const long px = (long)(d->B) * (d->H >> (s.upsample ? 1 : 0)) * (d->W >> (s.upsample ? 1 : 0));
        // byte offsets (region base + relative pixel x stride + the out-of-range bias 0x7FFFFFF0) stay below 2^32, pixels below 2^24
        const long ext = px + (d->W >> (s.upsample ? 1 : 0)) + 2;
        ND_REQUIRE(ext * s.ld0 * 4 < (1L << 30) - 65536 && ext * s.ld1 * 4 < (1L << 30) - 65536 && ext < (1L << 24), ND_E_SHAPE,
                   "nd_conv3x3_wino4: a source tensor of 1 GiB or 16 M pixels or more");
    }

    ND_REQUIRE((long)d->B * d->H * d->W * d->ldo * 4 < (1L << 32) - 65536, ND_E_SHAPE, "nd_conv3x3_wino4: an output tensor of 4 GiB or more");

    a.d = *d;
    a.tiles_x = nd_cdiv(d->W, 16);
    a.tiles_y = nd_cdiv(d->H, 16);
    a.regions_x = nd_cdiv(d->W, 16 * ntg);
    a.n_tiles = nd_cdiv(d->cout, 64);
    a.n_cg = nd_round_up(d->cout, 64) / 16;
    a.n_c8 = nd_round_up(nd_cdiv(d->cin, 8), 2);
    a.slots = a.tiles_x * a.tiles_y;
    const long wg = (long)d->B * a.regions_x * a.tiles_y * a.n_tiles;
    ND_REQUIRE(wg < (1L << 31), ND_E_SHAPE, "nd_conv3x3_wino4: grid too large");
    a.total_wg = (int)wg;
    a.splits = 1;
    a.chunks_per_split = nd_cdiv(d->cin, KC4);
    return 0;
}

// split-K launch shared by the entry points of both product forms: SPLIT instances into `workspace`, then the reduction (+ bias, + statistics)
static int w4_splitk(const nd_conv3x3* d, float* workspace, int splits, void* stream, int ntg, const char* who) {
    Wino4Args a;
    if (int e = w4_prepare(d, a, ntg)) return e;
    ND_REQUIRE(workspace && nd_aligned16(workspace), ND_E_BADARG, "%s: the workspace must be a 16-byte aligned pointer", who);
    ND_REQUIRE(d->src.mode == ND_PRO_NONE || d->src.mode == ND_PRO_AFFINE_SILU, ND_E_BADARG,
               "%s: plain or GroupNorm-affine + SiLU sources (no map / LeakyReLU prologue)", who);
    const int n_chunks = nd_cdiv(d->cin, KC4);
    ND_REQUIRE((splits == 2 || splits == 4 || splits == 8) && d->cin % KC4 == 0 && n_chunks % splits == 0 && n_chunks / splits >= 2, ND_E_SHAPE,
               "%s: splits=%d must be 2, 4 or 8 and divide cin=%d into ranges of at least two whole 16-channel chunks", who, splits, d->cin);
    const long npix = (long)d->B * d->H * d->W;
    ND_REQUIRE((long)splits * npix * d->cout * 4 < (1L << 31), ND_E_SHAPE, "%s: partial sums of 2 GiB or more", who);
    ND_REQUIRE(d->cout % 4 == 0, ND_E_SHAPE, "%s: cout must be a multiple of 4", who);
    const float* bias = d->bias;
    float* out = d->out;
    const int ldo = d->ldo;
    a.d.out = workspace;  a.d.ldo = d->cout;  a.d.bias = nullptr;       // partial sums [split][B][H][W][cout]; the bias joins in the reduction
    a.d.stats = nullptr;  a.d.slot_count = nullptr;                      // ... and so do the statistics (of the SUMMED output)
    a.splits = splits;
    a.chunks_per_split = n_chunks / splits;
    a.total_wg *= splits;
    hipStream_t st = (hipStream_t)stream;
    const bool aff = d->src.mode == ND_PRO_AFFINE_SILU;
    if (int rc = ntg == 1 ? (aff ? launch4_split<ND_PRO_AFFINE_SILU, 1>(a, st) : launch4_split<ND_PRO_NONE, 1>(a, st))
                          : (aff ? launch4_split<ND_PRO_AFFINE_SILU>(a, st) : launch4_split<ND_PRO_NONE>(a, st))) return rc;
    if (int e = nd_launch_status(who)) return e;
    if (d->stats) {
        const int n_cgrp = nd_cdiv(d->cout, 64);
        hipLaunchKernelGGL(w4_splitk_reduce_stats_kernel, dim3((unsigned)(d->B * a.tiles_x * a.tiles_y * n_cgrp)), dim3(256), 0, st, workspace, bias, out,
                           d->stats, d->slot_count, splits, d->B, d->H, d->W, d->cout, ldo, a.tiles_x, a.tiles_y, n_cgrp);
        return nd_launch_status(who);
    }
    const long total = npix * (d->cout / 4);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(w4_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, workspace, bias, out, splits, npix, d->cout, ldo);
    return nd_launch_status(who);
}

extern "C" int nd_conv3x3_wino4_nhwc_f32(const nd_conv3x3* d, void* stream) {
    Wino4Args a;
    if (int e = w4_prepare(d, a)) return e;
    const nd_src& s = d->src;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    switch (s.mode) {
        case ND_PRO_AFFINE_SILU: rc = launch4<ND_PRO_AFFINE_SILU>(a, st); break;
        case ND_PRO_AFFINE_MAP_SILU: rc = launch4<ND_PRO_AFFINE_MAP_SILU>(a, st); break;
        case ND_PRO_AFFINE_GENMAP_SILU: rc = launch4<ND_PRO_AFFINE_GENMAP_SILU>(a, st); break;
        case ND_PRO_LEAKY: rc = launch4<ND_PRO_LEAKY>(a, st); break;
        case ND_PRO_LEAKY_SECOND: rc = launch4<ND_PRO_LEAKY_SECOND>(a, st); break;
        default: rc = launch4<ND_PRO_NONE>(a, st); break;
    }
    if (rc) return rc;
    return nd_launch_status("nd_conv3x3_wino4_nhwc_f32");
}

// The same operator on 16 x 16-pixel regions with two co-resident workgroups per CU (wino4_kernel<..., NTG = 1>): same packed weights, same statistics
// slots, same descriptor checks, the same bits as nd_conv3x3_wino4_nhwc_f32.  Plain and GroupNorm-affine + SiLU sources (the map / LeakyReLU prologues stay
// on the 16 x 32 form).  It is the F(4x4) path of images narrower than 32 pixels.
extern "C" int nd_conv3x3_wino4_16_nhwc_f32(const nd_conv3x3* d, void* stream) {
    Wino4Args a;
    if (int e = w4_prepare(d, a, 1)) return e;
    ND_REQUIRE(d->src.mode == ND_PRO_NONE || d->src.mode == ND_PRO_AFFINE_SILU, ND_E_BADARG,
               "nd_conv3x3_wino4_16: plain or GroupNorm-affine + SiLU sources (no map / LeakyReLU prologue)");
    hipStream_t st = (hipStream_t)stream;
    if (int rc = d->src.mode == ND_PRO_AFFINE_SILU ? launch4<ND_PRO_AFFINE_SILU, 1>(a, st) : launch4<ND_PRO_NONE, 1>(a, st)) return rc;
    return nd_launch_status("nd_conv3x3_wino4_16_nhwc_f32");
}

// ---- split-K (training at small batch, and any plain layer without a statistics epilogue whose items fill a fraction of the chip)
//
// The layer's items are (sample, 16 x 32-pixel region, 64-cout tile): 512 -> 512 at 32 x 32 with 4 samples has 64 of them for 256 CUs, each
// walking 32 K chunks.  nd_conv3x3_wino4_splitk_plan picks -- from the shape alone, so that the summation order never depends on the device --
// the number of K ranges (1, 2, 4 or 8) that brings the item count to about one per CU while leaving every range at least four chunks.
static int w4_splits(long items, int cin) {
    const int n_chunks = nd_cdiv(cin, KC4);
    if (cin % KC4) return 1;
    static const long cap = getenv("ND_W4_SPLIT_ITEMS") ? atol(getenv("ND_W4_SPLIT_ITEMS")) : 256;      // A/B knob (tools/ only)
    static const int min_chunks = getenv("ND_W4_SPLIT_MIN_CHUNKS") ? atoi(getenv("ND_W4_SPLIT_MIN_CHUNKS")) : 4;
    int splits = 1;
    while (splits < 8 && items * splits * 2 <= cap && n_chunks % (splits * 2) == 0 && n_chunks / (splits * 2) >= min_chunks) splits *= 2;
    return splits;
}

extern "C" int nd_conv3x3_wino4_splitk_plan(int B, int H, int W, int cin, int cout) {
    if (B <= 0 || H <= 0 || W <= 0 || cin <= 0 || cout <= 0) return 1;
    return w4_splits((long)B * nd_cdiv(W, 32) * nd_cdiv(H, 16) * nd_cdiv(cout, 64), cin);
}

extern "C" int64_t nd_conv3x3_wino4_splitk_workspace_floats(int B, int H, int W, int cout, int splits) {
    if (B <= 0 || H <= 0 || W <= 0 || cout <= 0 || splits <= 0) return -1;
    return (int64_t)splits * B * H * W * cout;
}

extern "C" int nd_conv3x3_wino4_splitk_nhwc_f32(const nd_conv3x3* d, float* workspace, int splits, void* stream) {
    return w4_splitk(d, workspace, splits, stream, 2, "nd_conv3x3_wino4_splitk_nhwc_f32");
}

// Split-K on the 16 x 16-region form for SAMPLING: layers with few items per sample (BASELINE config 2's 16 x 16 and 32 x 32 stages: 512 -> 512 at 16 x 16 is 8
// items per sample, 128 for 512 workgroup slots at 16 patches per GPU, each walking 32 K chunks).  The split count is a function of the SAMPLE's geometry alone
// (never of the batch), so a sample's bits do not depend on the batch it is sharded into: 2, 4 or 8 ranges of cin so that a sample has about 32 items, at
// least four 16-channel chunks per range.
extern "C" int nd_conv3x3_wino4_16_splitk_plan(int H, int W, int cin, int cout) {
    if (H <= 0 || W <= 0 || cin <= 0 || cout <= 0 || cin % KC4) return 1;
    const long items = (long)nd_cdiv(W, 16) * nd_cdiv(H, 16) * nd_cdiv(cout, 64);
    const int n_chunks = cin / KC4;
    int splits = 1;
    while (splits < 8 && items * splits < 32 && n_chunks % (splits * 2) == 0 && n_chunks / (splits * 2) >= 4) splits *= 2;
    return splits;
}

extern "C" int nd_conv3x3_wino4_16_splitk_nhwc_f32(const nd_conv3x3* d, float* workspace, int splits, void* stream) {
    return w4_splitk(d, workspace, splits, stream, 1, "nd_conv3x3_wino4_16_splitk_nhwc_f32");
}
