// conv3x3_f16x3.hip -- nn.Conv2d(cin, cout, 3, padding=1) on NHWC fp32 as a DIRECT convolution on the f16 matrix instruction of gfx950
// (v_mfma_f32_32x32x16_f16, the double-rate form: 16 channels per instruction), every fp32 product as three f16 products of two-term operands
// with fp32 accumulation -- the opt-in product form of conv3x3_wino4h.hip (ND_CONV_F16X3=1) without the Winograd transforms.
//
// Replaces Block.proj (models/archs/Diffusion_arch.py:131,136) for the layers the engine gives it (nd_conv3x3_f16x3_takes).  Why a direct form
// next to F(4x4,3x3): at 64 channels the Winograd kernel spends 30 % of a tile on matrix instructions and the rest on the input / output transforms,
// the V image and the epilogue (tools/w4_clock.py: 52 k cycles per 16 x 32-pixel tile, 15.6 k of them MFMA); a direct product has 4 x the
// multiplies of F(4x4) -- but no transform at all, the double-rate instruction (the F(4x4) position products would be LDS-bound with it), and an
// epilogue that is a transpose: 27.6 k MFMA cycles per tile at 64 channels, everything else issued in the gaps between them (an f16 MFMA occupies the
// matrix pipe for 32 cycles and the wave's issue slot for 4: tools/microbench/bf16_mfma_valu.hip).
//
// An activation a = A1 + A2 and a weight 2^11 w = U1 + U2 as two f16 terms each (A1 = rtz(a), A2 = rtz(a - A1): exact remainder; U from
// nd_pack_conv3x3_f16x3_weight, round-to-nearest); a w ~ 2^-11 (A1 U1 + A1 U2 + A2 U1), the dropped A2 U2 is 2^-22 of the product.  No
// Winograd transform amplifies the operands' rounding: against an fp64 convolution the error is at or below the fp32 kernels'
// (tests/test_hip_kernels.py::test_conv3x3_f16x3_*).
//
// One workgroup = 4 waves (one per SIMD) = a 16 x 32-pixel output region x 64 couts; wave w owns rows 4w .. 4w+3 of the region and both
// 32-cout blocks: 8 accumulators of 32 x 32 (128 AGPRs), pixels on the M side, couts on the N side.  K runs in chunks of 16 channels;
// the chunk's 18 x 34 halo lives in LDS already split: 80 bytes per pixel {A1 ch0..15 | A2 ch0..15 | pad} (the stride keeps the 16-byte
// operand reads of 16 consecutive pixels on distinct banks), double-buffered; the next chunk (of this tile or the next one) is requested
// from HBM behind the chunk's first MFMAs and written -- prologue, zero padding, split -- in slices of a few instructions behind its last
// ones, one barrier per chunk.  Per tap and chunk a wave reads 8 operands of 16 bytes from LDS (4 rows x 2 terms, shifted by the tap) and
// 4 weight fragments from L2 (register ring two taps ahead) for 24 MFMAs; every one of those requests sits behind an MFMA of its own.
// Epilogue: a lane holds 16 pixels of ONE cout per accumulator block and consecutive lanes consecutive couts, so the blocks are stored as they are
// (a dword per lane: 128 contiguous bytes per pixel and store); the GroupNorm statistics (two 16 x 16 slots per region, nd_conv3x3_wino4_stat_slots'
// geometry) are summed in the lane about the lane's first pixel and the eight partials of a (slot, cout) -- 4 waves x 2 lane halves, 32 pixels
// each -- combined by Chan's formula through LDS.  A workgroup takes a contiguous range of the (sample, region row, region, cout tile) items.
#include <stdlib.h>
#include <type_traits>
#include "nd_common.h"

namespace {

constexpr int DKC = 16;                                       // channels per K chunk = K of one MFMA
constexpr int RH = 16, RW = 32;                               // output region of a workgroup
constexpr int HALO_W = RW + 2, HALO_H = RH + 2, HALO_PX = HALO_W * HALO_H;      // 34 x 18 = 612 pixels
constexpr int PXB = 80;                                       // bytes per halo pixel in LDS
constexpr int ITEMS = 10;                                     // staging passes: 640 pixel slots x 4 channel quads / 256 threads
constexpr int ABUF = 64 * ITEMS * PXB;                        // 51,200 bytes per chunk buffer (slots 612..639 are never read)
constexpr int SP_OFF = 2 * ABUF;                              // statistics partials [wave 4][lane half 2][nb 2][slot 2][cout 32]{sum, M2}
constexpr int SP_BYTES = 4 * 2 * 2 * 2 * 32 * 2 * 4;          // 8 KB
constexpr int LDS_BYTES = SP_OFF + SP_BYTES;                  // 110,592 bytes
constexpr float WSCALE = 2048.0f, WSCALE_INV = 1.0f / 2048.0f;
constexpr int STORE_AUX = 19;                                 // cache-policy bits of the streaming output stores (sc0 | nt | sc1), as conv3x3_wino4.hip

struct DArgs {
    nd_conv3x3 d;
    int regions_x, regions_y, tiles_x, n_ct, n_chunks, slots, total_items;
};

#define D16_MFMA(acc, av, bv) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(av), "v"(bv))

template <int MODE, bool STREAM>
__global__ __launch_bounds__(256, 1) void conv3x3_f16x3_kernel(const DArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, col = lane & 31;
    const nd_src& s = a.d.src;
    const int H = a.d.H, W = a.d.W, Cout = a.d.cout, ldo = a.d.ldo;
    const int up = s.upsample ? 1 : 0;
    const int Hs = H >> up, Ws = W >> up;
    const int n_chunks = a.n_chunks;

    // ---- staging role: item j of this thread = halo pixel slot (tid >> 2) + 64 j, channel quad tid & 3 of the chunk
    const int quad = tid & 3;
    int hyx[ITEMS];                                           // hy | hx << 8 of the slot; 0xFFFF: a slot behind the halo
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int pi = (tid >> 2) + 64 * j, hy = pi / HALO_W, hx = pi - hy * HALO_W;
        hyx[j] = pi < HALO_PX ? (hy | (hx << 8)) : 0xFFFF;
    }
    const unsigned st_lds = (unsigned)((tid >> 2) * PXB + quad * 8);
    // ---- MFMA role: operand of pixel (row 4 wave + i + dy, column col + dx), channels 8 half .. + 7
    const unsigned a_lds = (unsigned)(((4 * wave) * HALO_W + col) * PXB + half * 16);

    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(s.p0), 0, (int)((unsigned)a.d.B * Hs * Ws * s.ld0 * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(s.p1 ? s.p1 : s.p0), 0,
                                                                          (int)((unsigned)a.d.B * Hs * Ws * (s.p1 ? s.ld1 : s.ld0) * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.d.weight), 0, a.n_ct * n_chunks * 9 * 4096, 0x00020000);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.d.out, 0, (int)((unsigned)a.d.B * H * W * ldo * 4u), 0x00020000);
    const unsigned wvoff = (unsigned)lane * 16u;

    f32x16 acc[4][2];
    f32x4 bq[3][4];                                           // weight fragments: ring over taps, [nb * 2 + term]
    f32x4 av[2][8];                                           // activation operands of a tap: [row * 2 + term]
    f32x4 raw[ITEMS];
    f32x4 pM = {0, 0, 0, 0}, pA = {1, 1, 1, 1}, pD = {0, 0, 0, 0};
    int pixoff[ITEMS];                                        // source pixel index of the staging target tile's items
    unsigned inmask = 0;                                      // bit j: item j lies inside the image

    // items (sample, region row, region column, cout tile), the cout tile fastest; a workgroup takes a contiguous range of them (the two cout tiles of a
    // region back to back: the second pass over the halo hits the L2), one division chain per kernel and carries from then on
    int item, item_end;
    {
        const int G = gridDim.x, wg = nd_xcd_remap(blockIdx.x, G);
        const int per = a.total_items / G, rem = a.total_items - per * G;
        item = wg * per + min(wg, rem);
        item_end = item + per + (wg < rem ? 1 : 0);
    }
    auto decode = [&](int it, int& b, int& ry, int& rx, int& ct) {
        ct = it % a.n_ct;  it /= a.n_ct;
        rx = it % a.regions_x;  it /= a.regions_x;
        ry = it % a.regions_y;
        b = it / a.regions_y;
    };
    auto advance = [&](int& b, int& ry, int& rx, int& ct) {
        if (++ct == a.n_ct) { ct = 0;  if (++rx == a.regions_x) { rx = 0;  if (++ry == a.regions_y) { ry = 0;  ++b; } } }
    };
    auto tile_offsets = [&](int b, int ry, int rx) {
        inmask = 0;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int y = ry * RH - 1 + (hyx[j] & 0xFF), x = rx * RW - 1 + (hyx[j] >> 8);
            const bool in = hyx[j] != 0xFFFF && y >= 0 && y < H && x >= 0 && x < W;
            pixoff[j] = in ? ((b * Hs + (y >> up)) * Ws + (x >> up)) : 0;
            inmask |= in ? (1u << j) : 0u;
        }
    };
    auto stage_load = [&](int b, int c) {                     // chunk c of the staging target: requests
        const int cb = c * DKC;
        const bool sec = cb >= s.c0;                          // uniform: a chunk never straddles the two sources (host check)
        const int soff = __builtin_amdgcn_readfirstlane((sec ? cb - s.c0 : cb) * 4);
        const unsigned ldb = (unsigned)(sec ? s.ld1 : s.ld0) * 4u;
        if (sec) {
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) raw[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, (unsigned)pixoff[j] * ldb + quad * 16u, soff, 0));
        } else {
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) raw[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0, (unsigned)pixoff[j] * ldb + quad * 16u, soff, 0));
        }
        if (MODE == ND_PRO_AFFINE_SILU) {
            const float* m = s.mad + (size_t)b * 3 * a.d.cin + cb + quad * 4;
            pM = nd_ld4(m);  pA = nd_ld4(m + a.d.cin);  pD = nd_ld4(m + 2 * a.d.cin);
        }
    };
    // The staging of item j in NS slices of at most ~6 VALU instructions (one slice goes behind one MFMA); the value stays in raw[j]
    constexpr int NS = MODE == ND_PRO_AFFINE_SILU ? 12 : 3;
    constexpr int G0 = 216 - ITEMS * NS;                       // first MFMA slot of a chunk (9 taps x 24) that carries a slice
    float se = 0.0f;                                           // exp(-v) between the two slices of a SiLU
    auto stage_slice = [&](unsigned char* dst, int j, int sl) {
        f32x4& v = raw[j];
        const int fin = NS - 2;                                // the last two slices: mask + split, the LDS writes
        if (MODE == ND_PRO_AFFINE_SILU) {
            if (sl == 0) v = v - pM;
            else if (sl == 1) v = __builtin_elementwise_fma(v, pA, pD);
            else if (sl < fin) {                               // nd_silu of component (sl - 2) / 2 in two halves
                const int e = (sl - 2) >> 1;
                if (((sl - 2) & 1) == 0) se = __expf(-v[e]);
                else v[e] = v[e] * __builtin_amdgcn_rcpf(1.0f + se);
            }
        }
        if (sl == fin) {
            const f32x4 zero = {0, 0, 0, 0};
            v = ((inmask >> j) & 1u) ? v : zero;               // the convolution's zero padding comes behind the activation
            v = nd_split4_f16(v);
        }
        if (sl == fin + 1) {
            unsigned char* p = dst + st_lds + j * (64 * PXB);
            *reinterpret_cast<f32x2*>(p) = f32x2{v.x, v.y};
            *reinterpret_cast<f32x2*>(p + 32) = f32x2{v.z, v.w};
        }
    };
    auto load_b1 = [&](int slot, int wbase, int tap, int f) {  // fragment f = nb * 2 + term of a tap: 1 KB, 16 bytes per lane
        bq[slot % 3][f] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, __builtin_amdgcn_readfirstlane(wbase + (tap * 4 + f) * 1024), 0));
    };
    auto load_a1 = [&](int slot, const unsigned char* src, int tap, int o) {      // operand o = row * 2 + term of a tap
        const int dy = tap / 3, dx = tap - 3 * dy, i = o >> 1, term = o & 1;
        av[slot & 1][o] = *reinterpret_cast<const f32x4*>(src + a_lds + ((i + dy) * HALO_W + dx) * PXB + term * 32);
    };

#ifdef D16_STAMP             // diagnostic (tools/d16_clock.py): cycle split of a workgroup, written to slot_count as 8 x u64 per workgroup
    unsigned long long sk_t0 = __builtin_amdgcn_s_memtime(), sk_r0 = __builtin_amdgcn_s_memrealtime(), sk_mma = 0, sk_bar = 0, sk_epi = 0, sk_tiles = 0, sk_t;
#define D16_T0() sk_t = __builtin_amdgcn_s_memtime()
#define D16_ACC(x) x += __builtin_amdgcn_s_memtime() - sk_t
#else
#define D16_T0()
#define D16_ACC(x)
#endif
    // ---- pipeline prologue: the first tile's first chunk, the weight fragments of its first two taps
    int b = 0, ry = 0, rx = 0, ct = 0;
    if (item < item_end) {
        decode(item, b, ry, rx, ct);
        tile_offsets(b, ry, rx);
        stage_load(b, 0);
#pragma unroll
        for (int f = 0; f < 4; ++f) { load_b1(0, (ct * n_chunks) * 9 * 4096, 0, f);  load_b1(1, (ct * n_chunks) * 9 * 4096, 1, f); }
#pragma unroll
        for (int j = 0; j < ITEMS; ++j)
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) stage_slice(lds, j, sl);
    }
    __syncthreads();
    int par = 0;

    for (; item < item_end; ++item) {
        int nb_ = b, nry = ry, nrx = rx, nct = ct;            // the tile behind this one (or this one again: a harmless re-stage)
        if (item + 1 < item_end) advance(nb_, nry, nrx, nct);
        float biasv[2];                                        // this lane's cout of either block: requested here, used in the epilogue
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) biasv[nb] = a.d.bias ? a.d.bias[ct * 64 + nb * 32 + col] : 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[i][nb] = nd_zero16();

        for (int c = 0; c < n_chunks; ++c) {
            const unsigned char* cur = lds + par * ABUF;
            unsigned char* nxt = lds + (par ^ 1) * ABUF;
            const bool last = c + 1 == n_chunks;
            const int wcur = ((ct * n_chunks + c) * 9) * 4096;
            const int wnxt = last ? ((nct * n_chunks) * 9) * 4096 : wcur + 9 * 4096;
            if (last) tile_offsets(nb_, nry, nrx);            // the staging target moves on to the next tile
            D16_T0();
#pragma unroll
            for (int o = 0; o < 8; ++o) load_a1(0, cur, 0, o);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
#pragma unroll
                for (int k = 0; k < 24; ++k) {
                    const int prod = k >> 3, i = (k >> 1) & 3, nb = k & 1;       // A1 U1, A1 U2, A2 U1
                    D16_MFMA(acc[i][nb], av[t & 1][i * 2 + (prod == 2 ? 1 : 0)], bq[t % 3][nb * 2 + (prod == 1 ? 1 : 0)]);
                    // ---- behind this MFMA (its 32 cycles of matrix pipe leave ~6 issue slots): one request or one staging slice
                    const int g = t * 24 + k;
                    if (k < 8) { if (t + 1 < 9) load_a1(t + 1, cur, t + 1, k); }
                    else if (k < 12) { if (t + 2 < 9) load_b1(t + 2, wcur, t + 2, k - 8); else load_b1(t + 2, wnxt, t + 2 - 9, k - 8); }
                    else if (g == 12) stage_load(last ? nb_ : b, last ? 0 : c + 1);
                    if (g >= G0) stage_slice(nxt, (g - G0) / NS, (g - G0) % NS);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            D16_ACC(sk_mma);
            D16_T0();
            __syncthreads();                                   // the other buffer is complete, this one has been consumed
            D16_ACC(sk_bar);
            par ^= 1;
        }

        D16_T0();
        // ---- epilogue.  The MFMAs are asm statements: hipcc does not know their results are still in flight
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) asm volatile("" : "+a"(acc[i][nb]));
        {
            // accumulator register r of a block = pixel column 8 (r / 4) + 4 half + r % 4 of the row, cout col: slot half r / 8
            const bool want_stats = a.d.stats != nullptr;
            float S[2][2], Q[2][2], P[2][2];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) { S[nb][hf] = 0.0f;  Q[nb][hf] = 0.0f;  P[nb][hf] = 0.0f; }
            const unsigned ovoff = (unsigned)((4 * half * ldo + col) * 4);
            const int ldo4 = ldo * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int y = ry * RH + 4 * wave + i;
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const int sbase = __builtin_amdgcn_readfirstlane((((b * H + y) * W + rx * RW) * ldo + ct * 64 + nb * 32) * 4);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = __builtin_fmaf(acc[i][nb][r], WSCALE_INV, biasv[nb]);
                        const int hf = r >> 3;
                        if (want_stats) {
                            if (i == 0 && (r & 7) == 0) P[nb][hf] = v;      // the lane's pivot: its first pixel of the slot
                            const float dv = v - P[nb][hf];
                            S[nb][hf] += dv;
                            Q[nb][hf] = __builtin_fmaf(dv, dv, Q[nb][hf]);
                        }
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), orsrc, ovoff, sbase + (8 * (r >> 2) + (r & 3)) * ldo4, STREAM ? STORE_AUX : 0);
                    }
                }
            }
            if (want_stats) {                                  // uniform
                float* sp = reinterpret_cast<float*>(lds + SP_OFF);
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf)             // 32 pixels about the lane's pivot -> {sum, M2 about the partial's mean}
                        *reinterpret_cast<f32x2*>(sp + (((((wave * 2 + half) * 2 + nb) * 2 + hf) * 32 + col) * 2)) =
                            f32x2{__builtin_fmaf(32.0f, P[nb][hf], S[nb][hf]), fmaxf(Q[nb][hf] - S[nb][hf] * S[nb][hf] * (1.0f / 32.0f), 0.0f)};
                __syncthreads();
                if (tid < 128) {                               // (slot half, cout): the eight partials by Chan's formula
                    const int hf = tid >> 6, co = tid & 63, nb = co >> 5, c32 = co & 31;
                    float sum_p[8], m2 = 0.0f, sum = 0.0f;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const f32x2 t = *reinterpret_cast<const f32x2*>(sp + ((((q * 2 + nb) * 2 + hf) * 32 + c32) * 2));
                        sum_p[q] = t.x;  sum += t.x;  m2 += t.y;
                    }
                    const float mean = sum * (1.0f / 256.0f);
#pragma unroll
                    for (int q = 0; q < 8; ++q) { const float dm = sum_p[q] * (1.0f / 32.0f) - mean;  m2 = __builtin_fmaf(32.0f * dm, dm, m2); }
                    const int slot = ry * a.tiles_x + 2 * rx + hf;    // one statistics slot per 16 x 16 tile (nd_conv3x3_wino4_stat_slots)
                    float* o = a.d.stats + (((size_t)b * a.slots + slot) * Cout + ct * 64 + co) * 2;
                    *reinterpret_cast<f32x2*>(o) = f32x2{sum, m2};
#ifndef D16_STAMP
                    if (b == 0 && ct == 0 && co == 0) a.d.slot_count[slot] = 256.0f;
#endif
                }
            }
        }
        D16_ACC(sk_epi);
#ifdef D16_STAMP
        ++sk_tiles;
#endif
        b = nb_;  ry = nry;  rx = nrx;  ct = nct;
    }
#ifdef D16_STAMP
    if (tid == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(a.d.slot_count) + 8 * blockIdx.x;
        o[0] = __builtin_amdgcn_s_memtime() - sk_t0;  o[1] = __builtin_amdgcn_s_memrealtime() - sk_r0;
        o[2] = sk_mma;  o[3] = sk_bar;  o[4] = sk_epi;  o[5] = sk_tiles;  o[6] = sk_r0;  o[7] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// OIHW (cout, cin, 3, 3) -> 2^11 w as two f16 terms in the operand order of v_mfma_f32_32x32x16_f16:
// [cout / 64][cin / 16][tap 9][nb 2][term 2][lane 64] x 16 bytes; lane (n = l & 31, h = l >> 5) holds channels 8 h .. 8 h + 7 of its chunk for cout 32 nb + n
__global__ void pack_f16x3_kernel(const float* __restrict__ w, float* __restrict__ out, int cin, int cout, int n_ct, int n_chunks) {
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    const size_t total = (size_t)n_ct * n_chunks * 9 * 2 * 64;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int l = i & 63, nb = (i >> 6) & 1;
        size_t r = i >> 7;
        const int tap = r % 9;  r /= 9;
        const int ch = r % n_chunks, ct = r / n_chunks;
        const int co = ct * 64 + nb * 32 + (l & 31), c0 = ch * 16 + 8 * (l >> 5);
        _Float16 h1[8], h2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float u = (co < cout && c0 + e < cin) ? w[((size_t)co * cin + c0 + e) * 9 + tap] * WSCALE : 0.0f;
            h1[e] = (_Float16)fminf(fmaxf(u, -65504.0f), 65504.0f);       // (a weight beyond 32 saturates the first term; the remainder carries on)
            h2[e] = (_Float16)fminf(fmaxf(u - (float)h1[e], -65504.0f), 65504.0f);
        }
        f32x4 t1, t2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            t1[e] = __builtin_bit_cast(float, h2v{h1[2 * e], h1[2 * e + 1]});
            t2[e] = __builtin_bit_cast(float, h2v{h2[2 * e], h2[2 * e + 1]});
        }
        float* o = out + ((((size_t)(ct * n_chunks + ch) * 9 + tap) * 2 + nb) * 2) * 256 + l * 4;
        nd_st4(o, t1);
        nd_st4(o + 256, t2);
    }
}

template <int MODE, bool STREAM>
int launch_d(const DArgs& a, hipStream_t st) {
    static nd_device_once configured;
    if (int e = nd_reserve_lds(configured, reinterpret_cast<const void*>(conv3x3_f16x3_kernel<MODE, STREAM>), LDS_BYTES, "nd_conv3x3_f16x3")) return e;
    const int cus = nd_device_cus();
    hipLaunchKernelGGL((conv3x3_f16x3_kernel<MODE, STREAM>), dim3((unsigned)(a.total_items < cus ? a.total_items : cus)), dim3(256), LDS_BYTES, st, a);
    return 0;
}

// what the kernel covers: whole 16 x 32 regions, whole 16-channel chunks, whole 64-cout tiles; plain and GroupNorm-affine + SiLU sources
bool d16_takes(const nd_conv3x3* d) {
    const nd_src& s = d->src;
    const int up = s.upsample ? 1 : 0;
    return d->B > 0 && d->H > 0 && d->W > 0 && d->H % RH == 0 && d->W % RW == 0 && d->cin % DKC == 0 && d->cin >= DKC && d->cout % 64 == 0 &&
           (s.mode == ND_PRO_NONE || s.mode == ND_PRO_AFFINE_SILU) && !s.unshuffle && !s.map_blocked && s.c0 + s.c1 == d->cin && s.c0 > 0 &&
           (s.c1 == 0 || (s.c0 % DKC == 0 && !up)) &&
           (long)d->B * (d->H >> up) * (d->W >> up) * s.ld0 * 4 < (1L << 31) && (long)d->B * (d->H >> up) * (d->W >> up) * s.ld1 * 4 < (1L << 31) &&
           (long)d->B * d->H * d->W * d->ldo * 4 < (1L << 31);
}

}  // namespace

extern "C" int64_t nd_pack_conv3x3_f16x3_weight_floats(int cin, int cout) {
    return (int64_t)nd_cdiv(cout, 64) * nd_cdiv(cin, DKC) * 9 * 1024;
}

extern "C" int nd_pack_conv3x3_f16x3_weight(const float* oihw, float* packed, int cin, int cout, void* stream) {
    ND_REQUIRE(oihw && packed, ND_E_BADARG, "nd_pack_conv3x3_f16x3_weight: null pointer");
    ND_REQUIRE(cin > 0 && cout > 0, ND_E_BADARG, "nd_pack_conv3x3_f16x3_weight: non-positive size");
    const int n_ct = nd_cdiv(cout, 64), n_chunks = nd_cdiv(cin, DKC);
    const size_t total = (size_t)n_ct * n_chunks * 9 * 2 * 64;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(pack_f16x3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, oihw, packed, cin, cout, n_ct, n_chunks);
    return nd_launch_status("nd_pack_conv3x3_f16x3_weight");
}

extern "C" int nd_conv3x3_f16x3_takes(const nd_conv3x3* d) { return d && d16_takes(d) ? 1 : 0; }

extern "C" int nd_conv3x3_f16x3_nhwc_f32(const nd_conv3x3* d, void* stream) {
    ND_REQUIRE(d, ND_E_BADARG, "nd_conv3x3_f16x3: null descriptor");
    const nd_src& s = d->src;
    ND_REQUIRE(s.p0 && d->weight && d->out, ND_E_BADARG, "nd_conv3x3_f16x3: null tensor pointer");
    ND_REQUIRE(d16_takes(d), ND_E_SHAPE,
               "nd_conv3x3_f16x3: the layer is not one of the direct f16-split kernel's (H %% 16, W %% 32, cin %% 16, cout %% 64 == 0; plain or affine + SiLU "
               "source; tensors below 2 GiB): ask nd_conv3x3_f16x3_takes first (H=%d W=%d cin=%d cout=%d mode=%d)", d->H, d->W, d->cin, d->cout, s.mode);
    ND_REQUIRE((s.c1 == 0) == (s.p1 == nullptr), ND_E_BADARG, "nd_conv3x3_f16x3: p1/c1 mismatch");
    ND_REQUIRE(s.ld0 >= s.c0 && s.ld0 % 4 == 0 && s.c0 % 4 == 0 && (s.c1 == 0 || (s.ld1 >= s.c1 && s.ld1 % 4 == 0)), ND_E_ALIGN,
               "nd_conv3x3_f16x3: pixel strides must be >= channels and multiples of 4");
    ND_REQUIRE(nd_aligned16(s.p0) && nd_aligned16(s.p1) && nd_aligned16(d->weight) && nd_aligned16(s.mad) && nd_aligned16(d->out) && nd_aligned16(d->bias),
               ND_E_ALIGN, "nd_conv3x3_f16x3: pointers must be 16-byte aligned");
    ND_REQUIRE(d->ldo >= d->cout && d->ldo % 4 == 0, ND_E_SHAPE, "nd_conv3x3_f16x3: ldo must be >= cout and a multiple of 4");
    ND_REQUIRE(s.mode != ND_PRO_AFFINE_SILU || s.mad, ND_E_BADARG, "nd_conv3x3_f16x3: affine prologue needs mad");
    ND_REQUIRE((d->stats == nullptr) == (d->slot_count == nullptr), ND_E_BADARG, "nd_conv3x3_f16x3: stats and slot_count go together");

    DArgs a;
    a.d = *d;
    a.regions_x = d->W / RW;
    a.regions_y = d->H / RH;
    a.tiles_x = d->W / 16;
    a.n_ct = d->cout / 64;
    a.n_chunks = d->cin / DKC;
    a.slots = (d->W / 16) * (d->H / 16);
    const long items = (long)d->B * a.regions_x * a.regions_y * a.n_ct;
    ND_REQUIRE(items < (1L << 31), ND_E_SHAPE, "nd_conv3x3_f16x3: grid too large");
    a.total_items = (int)items;
    hipStream_t st = (hipStream_t)stream;
    static const long stream_min = (getenv("ND_W4_STREAM_MB") ? atol(getenv("ND_W4_STREAM_MB")) : 48) << 20;     // the streaming-store rule of conv3x3_wino4.hip
    const bool stream_out = (long)d->B * d->H * d->W * d->ldo * 4 >= stream_min;
    int rc;
    if (s.mode == ND_PRO_AFFINE_SILU) rc = stream_out ? launch_d<ND_PRO_AFFINE_SILU, true>(a, st) : launch_d<ND_PRO_AFFINE_SILU, false>(a, st);
    else rc = stream_out ? launch_d<ND_PRO_NONE, true>(a, st) : launch_d<ND_PRO_NONE, false>(a, st);
    if (rc) return rc;
    return nd_launch_status("nd_conv3x3_f16x3_nhwc_f32");
}
