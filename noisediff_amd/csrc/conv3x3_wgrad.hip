// conv3x3_wgrad.hip -- weight gradient of nn.Conv2d(cin, cout, 3, padding=1) on NHWC fp32 (first training kernel).
//
//   dW[co][ci][r][s] = sum over (b, y, x) of dY[b][y][x][co] * X[b][y + r - 1][x + s - 1][ci]          (zero padding)
//
// the backward of Block.proj / the up- and down-sampling convs under GaussianDiffusion.p_losses
// (models/denoising_diffusion_pytorch.py:481-531 -> loss.backward(), models/trainer_diffusion.py:187).  The data gradient of
// the same layer needs no kernel of its own: it is the forward convolution with the taps flipped and the channel roles
// swapped, i.e. nd_conv3x3_*_nhwc_f32 on weights packed from w.flip(2, 3).transpose(0, 1) (noisediff_amd/train.py).
//
// Structure: nine 64 x 64 GEMMs (one per tap) that share their A operand, K = pixels, on the exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32: D(co, ci) += dY(pixel pair, co)^T X(pixel pair, ci)).  A workgroup owns a (64 couts x 64 cins) block
// of all nine taps -- 4 waves x 9 accumulators of 32 x 32 -- and walks its share of the 16 x 16-pixel tiles: per tile the dY tile
// and the 18 x 18 X halo (zero padded) are staged in LDS (145 KB), then 128 pixel pairs x 9 MFMAs per wave with both operands
// read as one conflict-free ds_read_b32 each.  The split over pixel tiles is fixed by the shape alone, partial sums go to a
// workspace and a second kernel adds them in a fixed order: bitwise repeatable, no atomics.  The bias gradient (sum of dY over the
// pixels) falls out of the staged dY tiles of the first cin block's workgroups.
#include <type_traits>
#include <stdlib.h>
#include "nd_common.h"

namespace {

#ifndef WG_TILE_ROWS
#define WG_TILE_ROWS 8        // 16 x 8-pixel tiles: 78 KB of LDS, so TWO workgroups share a CU and one's staging runs under the other's MFMAs
#endif                        // (16 x 16: 145 KB, one workgroup per CU, every tile's staging exposed)
constexpr int WG_TILE = 16, TILE_H = WG_TILE_ROWS, HALO = 18, HALO_H = TILE_H + 2, CB = 64;   // tile width / height, halo width / height, channel block
constexpr int WGRAD_TARGET_WGS = (WG_TILE_ROWS <= 8 ? 512 : 256);                                  // one workgroup per CU of an MI355X (LDS: one fits); fixed: the summation order must not depend on the device

struct WgradArgs {
    const float* x; const float* dy; float* ws; float* wsb;          // wsb: bias-gradient partials [S][coP] (null: no bias gradient)
    int ldx, ldy, B, H, W, cin, cout;
    int n_co, n_ci, S, tiles_x, tiles_y, n_tiles, fast;
    unsigned dy_bytes, x_bytes;
};

__global__ __launch_bounds__(256, (WG_TILE_ROWS <= 8 ? 2 : 1)) void wgrad_kernel(const WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* dYs = sm;                                                 // [256 pixels][64 couts]
    float* Xs = sm + WG_TILE * TILE_H * CB;                          // [halo pixels][64 cins]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int co_w = 32 * (wave >> 1), ci_w = 32 * (wave & 1);       // this wave's 32 x 32 block of the 64 x 64
    int bid = blockIdx.x;
    const int s = bid % a.S;  bid /= a.S;
    const int cib = bid % a.n_ci, cob = bid / a.n_ci;
    const int co0 = cob * CB, ci0 = cib * CB;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = nd_zero16();
    // bias gradient db[co] = sum over the pixels of dY[.][co]: falls out of the staged dY tile in the workgroups of the first cin block
    // (thread = cout tid & 63, pixel quarter tid >> 6) -- no second pass over dY (ATen's sum over a channels_last tensor took 30 us per layer)
    float bsum = 0.0f;
    const bool do_bias = a.wsb != nullptr && cib == 0;

    // Interior tiles (no image border inside the halo, whole channel blocks) take a path without per-item vector arithmetic: buffer loads
    // with the tile's base in the scalar offset and per-thread item offsets computed once -- every VALU instruction of the staging pass
    // waits behind a 64-cycle MFMA of the workgroup that shares the SIMDs (12 % of the kernel with the clamped / select form below).
    constexpr int DY_IT = WG_TILE * TILE_H * (CB / 4) / 256;             // 8 (16)
    constexpr int X_IT = (HALO * HALO_H * (CB / 4) + 255) / 256;          // 12 (21)
    const bool fast_ok = a.fast && co0 + CB <= a.cout && ci0 + CB <= a.cin;
    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, fast_ok ? (int)a.dy_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, fast_ok ? (int)a.x_bytes : 0, 0x00020000);
    const unsigned dy_voff = (unsigned)(((tid >> 4) * a.ldy + co0 + 4 * (tid & 15)) * 4);          // item i: + i rows
    // X halo (18 wide): columns 0-15 of halo row i are item i of thread row tid >> 4 (a scalar row stride apart); the 2 x HALO_H edge pixels
    // (columns 16, 17) take two more items: edge pixel e = row e >> 1, column 16 + (e & 1)
    const int tr = tid >> 4, e1 = min(16 + tr, 2 * HALO_H - 1);
    const unsigned xm_voff = (unsigned)((tr * a.ldx + ci0 + 4 * (tid & 15)) * 4);
    const unsigned xe0_voff = (unsigned)((((tr >> 1) * a.W + 16 + (tr & 1)) * a.ldx + ci0 + 4 * (tid & 15)) * 4);
    const unsigned xe1_voff = (unsigned)((((e1 >> 1) * a.W + 16 + (e1 & 1)) * a.ldx + ci0 + 4 * (tid & 15)) * 4);
    static_assert(2 * HALO_H <= 32, "edge pixels fit two passes of 16 thread rows");

    for (int tile = s; tile < a.n_tiles; tile += a.S) {
        int lid = tile;
        const int tx = lid % a.tiles_x;  lid /= a.tiles_x;
        const int ty = lid % a.tiles_y;
        const int b = lid / a.tiles_y;
        const int y0 = ty * TILE_H, x0 = tx * WG_TILE;
        __syncthreads();                                             // the previous tile's operands have been consumed
        // ---- stage dY (16 x 16 x 64 couts) and X (18 x 18 x 64 cins); outside the image / beyond the channels: zeros.
        //      Every load is unconditional (clamped address, select afterwards) and a batch of them is in flight before the first LDS
        //      write: with a branch around each load hipcc waits for it on the spot, and the ~37 serialized memory round trips per tile
        //      took as long as the tile's MFMAs.
        const bool interior = fast_ok && y0 >= 1 && y0 + TILE_H + 1 <= a.H && x0 >= 1 && x0 + WG_TILE + 1 <= a.W;      // workgroup-uniform
#ifdef WG_ABLATE_STAGE          // diagnostic: operands staged for the first tile only
        if (tile == s)
#endif
        if (interior) {
            const int q = tid & 15;
            const unsigned dy_base = (unsigned)(((b * a.H + y0) * a.W + x0) * a.ldy) * 4u;
            const unsigned x_base = (unsigned)(((b * a.H + y0 - 1) * a.W + x0 - 1) * a.ldx) * 4u;
            const unsigned dy_row = (unsigned)(a.W * a.ldy) * 4u;
            const unsigned x_row = (unsigned)(a.W * a.ldx) * 4u;
            float* xdst = Xs + tr * CB + 4 * q;
            {   // two batches (registers: 144 accumulators live): dY + the first halo rows, then the rest and the edge columns
                constexpr int XA = HALO_H / 2;
                f32x4 vy[DY_IT], vx[XA];
#pragma unroll
                for (int i = 0; i < DY_IT; ++i)
                    vy[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, dy_voff, dy_base + i * dy_row, 0));
#pragma unroll
                for (int i = 0; i < XA; ++i)
                    vx[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, xm_voff, x_base + i * x_row, 0));
#pragma unroll
                for (int i = 0; i < DY_IT; ++i) nd_st4(dYs + (tr + 16 * i) * CB + 4 * q, vy[i]);
#pragma unroll
                for (int i = 0; i < XA; ++i) nd_st4(xdst + i * HALO * CB, vx[i]);
            }
            {
                constexpr int XA = HALO_H / 2, XB = HALO_H - XA;
                f32x4 vx[XB], ve0, ve1;
#pragma unroll
                for (int i = 0; i < XB; ++i)
                    vx[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, xm_voff, x_base + (XA + i) * x_row, 0));
                ve0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, xe0_voff, x_base, 0));
                ve1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, xe1_voff, x_base, 0));
#pragma unroll
                for (int i = 0; i < XB; ++i) nd_st4(xdst + (XA + i) * HALO * CB, vx[i]);
                nd_st4(Xs + ((tr >> 1) * HALO + 16 + (tr & 1)) * CB + 4 * q, ve0);
                if (16 + tr < 2 * HALO_H) nd_st4(Xs + ((e1 >> 1) * HALO + 16 + (e1 & 1)) * CB + 4 * q, ve1);
            }
        } else
#ifdef WG_ABLATE_STAGE
        if (tile == s)
#endif
        {
            const f32x4 zero = {0, 0, 0, 0};
            const int q = tid & 15;
            constexpr int BATCH = 11;
            const int cy = co0 + 4 * q, cx = ci0 + 4 * q;
            const bool cy_ok = cy < a.cout, cx_ok = cx < a.cin;
            const float* dyb = a.dy + (cy_ok ? cy : 0);
            const float* xb = a.x + (cx_ok ? cx : 0);
#pragma unroll
            for (int i0 = 0; i0 < DY_IT; i0 += BATCH) {
                f32x4 v[BATCH];
#pragma unroll
                for (int j = 0; j < BATCH; ++j) {
                    if (i0 + j < DY_IT) {
                        const int p = (tid >> 4) + 16 * (i0 + j), py = p >> 4, px = p & 15;
                        const int gy = min(y0 + py, a.H - 1), gx = min(x0 + px, a.W - 1);
                        v[j] = nd_ld4(dyb + ((size_t)(b * a.H + gy) * a.W + gx) * a.ldy);
                    }
                }
#pragma unroll
                for (int j = 0; j < BATCH; ++j) {
                    if (i0 + j < DY_IT) {
                        const int p = (tid >> 4) + 16 * (i0 + j), py = p >> 4, px = p & 15;
                        const bool ok = cy_ok && y0 + py < a.H && x0 + px < a.W;
                        nd_st4(dYs + p * CB + 4 * q, ok ? v[j] : zero);
                    }
                }
            }
#pragma unroll
            for (int i0 = 0; i0 < X_IT; i0 += BATCH) {
                f32x4 v[BATCH];
#pragma unroll
                for (int j = 0; j < BATCH; ++j) {
                    if (i0 + j < X_IT) {
                        const int p = min((tid >> 4) + 16 * (i0 + j), HALO * HALO_H - 1), py = p / HALO, px = p - py * HALO;
                        const int gy = min(max(y0 + py - 1, 0), a.H - 1), gx = min(max(x0 + px - 1, 0), a.W - 1);
                        v[j] = nd_ld4(xb + ((size_t)(b * a.H + gy) * a.W + gx) * a.ldx);
                    }
                }
#pragma unroll
                for (int j = 0; j < BATCH; ++j) {
                    if (i0 + j < X_IT) {
                        const int p = (tid >> 4) + 16 * (i0 + j), py = p / HALO, px = p - py * HALO;
                        const int gy = y0 + py - 1, gx = x0 + px - 1;
                        const bool ok = cx_ok && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
                        if (p < HALO * HALO_H) nd_st4(Xs + p * CB + 4 * q, ok ? v[j] : zero);
                    }
                }
            }
        }
        __syncthreads();
        // ---- 128 pixel pairs x 9 taps: lane (col, half) feeds dY[pixel + half][co_w + col] and X[pixel + half + tap][ci_w + col]
        const float* ap = dYs + half * CB + co_w + col;
        const float* bp = Xs + half * CB + ci_w + col;
#pragma unroll 2
        for (int py = 0; py < TILE_H; ++py) {
#pragma unroll
            for (int px = 0; px < WG_TILE; px += 2) {
                const float av = ap[(py * WG_TILE + px) * CB];
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const float bv = bp[((py + t / 3) * HALO + px + t % 3) * CB];
                    acc[t] = nd_mfma(av, bv, acc[t]);
                }
            }
        }
        if (do_bias) {
            constexpr int QR = WG_TILE * TILE_H / 4;
            const float* cp = dYs + (tid >> 6) * QR * CB + (tid & 63);
#pragma unroll 8
            for (int r = 0; r < QR; ++r) bsum += cp[r * CB];
        }
    }
    // ---- this workgroup's partial sums: ws[s][tap][co][ci], rows of 32 consecutive cins per lane group
    const int coP = a.n_co * CB, ciP = a.n_ci * CB;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + co_w + nd_acc_row(r, lane), ci = ci0 + ci_w + col;
            a.ws[(((size_t)s * 9 + t) * coP + co) * ciP + ci] = acc[t][r];
        }
    if (do_bias) {                                                   // the four pixel quarters meet in a fixed order
        __syncthreads();
        dYs[tid] = bsum;
        __syncthreads();
        if (tid < 64) a.wsb[(size_t)s * coP + co0 + tid] = dYs[tid] + dYs[64 + tid] + dYs[128 + tid] + dYs[192 + tid];
    }
}

// dW (OIHW, torch layout) = sum over the S partials in a fixed order.  A thread owns one element (tap, co, ci) of the partial
// blocks -- consecutive threads read consecutive cins, every load of the S-deep sum is coalesced -- and scatters its one result.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, const float* __restrict__ wsb, float* __restrict__ dw,
                                                           float* __restrict__ db, int S, int cin, int cout, int coP, int ciP) {
    const size_t block = (size_t)9 * coP * ciP, total = block + (db ? coP : 0);     // the bias partials follow the weight partials element-wise
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < total; j += (size_t)gridDim.x * blockDim.x) {
        const bool bias = j >= block;
        const int ci = (int)(j % ciP);
        const int co = bias ? (int)(j - block) : (int)((j / ciP) % coP);
        const int t = (int)(j / ((size_t)ciP * coP));
        if (co >= cout || (!bias && ci >= cin)) continue;
        const float* p = bias ? wsb + co : ws + j;
        const size_t stride = bias ? (size_t)coP : block;
        float sum = 0.0f;
        int s = 0;
        for (; s + 8 <= S; s += 8) {                                     // eight loads in flight, added in slot order
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = p[(size_t)(s + k) * stride];
#pragma unroll
            for (int k = 0; k < 8; ++k) sum += v[k];
        }
        for (; s < S; ++s) sum += p[(size_t)s * stride];
        if (bias) db[co] = sum;
        else dw[((size_t)co * cin + ci) * 9 + t] = sum;
    }
}

void plan(int B, int H, int W, int cin, int cout, WgradArgs& a) {
    a.B = B; a.H = H; a.W = W; a.cin = cin; a.cout = cout;
    a.n_co = nd_cdiv(cout, CB);
    a.n_ci = nd_cdiv(cin, CB);
    a.tiles_x = nd_cdiv(W, WG_TILE);
    a.tiles_y = nd_cdiv(H, TILE_H);
    a.n_tiles = B * a.tiles_x * a.tiles_y;
    const int blocks = a.n_co * a.n_ci;
    int S = WGRAD_TARGET_WGS / blocks;                               // fixed by the shape: the summation order never depends on the device
    if (S < 1) S = 1;
    if (S > a.n_tiles) S = a.n_tiles;
    a.S = S;
}

// ====================================================================================================================================
// The same gradient in the WINOGRAD domain, F(4x4,3x3) (r5): with Y = A^T [(G g G^T) o (B^T d B)] A per 4 x 4 output tile,
//
//   dg = G^T [ sum over the tiles of (A dY A^T) o (B^T d B) ] G
//
// i.e. per tile the 6 x 6 transforms of the input patch (V = B^T d B, the forward kernel's) and of the 4 x 4 block of dY (D = A dY A^T),
// 36 position-wise GEMMs M[pos](co, ci) += D[pos](tile, co)^T V[pos](tile, ci) with K = tiles, and ONE back-transform G^T M G at the end:
// 36 multiplies per 16 pixels instead of 144 -- a quarter of the nine-tap form's MFMAs.  fp32 error against an fp64 sum: ~3-5e-6 of max|dW|
// (the nine-tap form: 4e-7; tests/test_train_gpu.py holds both to 2e-5).
//
// A workgroup owns a (32 couts x 32 cins) block of all 36 positions -- wave w the positions 9 w .. 9 w + 8, nine 32 x 32 accumulators pinned in
// AGPRs -- and walks its share of the tile groups (one row of four tiles = 4 x 16 pixels).  Per group: the 6 x 18 x 32 input halo and the
// 4 x 16 x 32 dY block go to LDS (requested two groups ahead, in registers), thread (tile, channel pair) transforms half a patch of each into
// V[pos][tile][ch] / D[pos][tile][ch] (the MFMA operands: lane (ch, tile parity) reads two dwords, conflict-free), then 18 MFMAs per wave
// (v_mfma_f32_32x32x2_f32, K = two tiles) -- all of it ONE software-pipelined instruction stream per wave (ww_wave below).  The split over tile
// groups is fixed by the shape alone; partials [split][pos][co][ci] and the bias partials go to the workspace, wgrad_wino_finish_kernel adds
// them in split order and applies G^T . G.  Bitwise repeatable, no atomics.
// Taken when H % 4 == 0, W % 16 == 0 and both channel counts are multiples of 32 (every Block.proj of the d = 64 network at the training sizes);
// anything else keeps the nine-tap kernel above.  ND_WGRAD_WINO=0: A/B knob.  Cycle split and what was tried: profiles/r5_wgrad_wino.txt.
constexpr int WW_CB = 32;                                            // channels per block, both sides
constexpr int WW_NT = 4;                                             // tile group: one row of four 4 x 4 tiles
constexpr int WW_GH = 4, WW_GW = 16;                                 // = 4 x 16 pixels of dY
constexpr int WW_HR = WW_GH + 2, WW_HC = WW_GW + 2;                  // input halo 6 x 18
constexpr int WW_PS = 40;                                            // floats per staged pixel: 32 channels + 8 (tiles t, t + 1 of a half wave's 8-byte reads on opposite bank halves)
constexpr int WW_XF = WW_HR * WW_HC * WW_PS, WW_YF = WW_GH * WW_GW * WW_PS, WW_RAWF = WW_XF + WW_YF;     // one raw buffer: halo, then the dY block
constexpr int WW_VDF = 36 * WW_NT * WW_CB;                           // one operand image [pos][tile][channel]
constexpr size_t WW_LDS = (size_t)(2 * WW_RAWF + 4 * WW_VDF) * sizeof(float);      // two raw buffers + two (V, D) pairs: 128768 bytes
constexpr int WW_TARGET_WGS = 256;                                   // fixed: the summation order must not depend on the device
constexpr int WW_X_IT = 7;                                           // float4s of the halo per thread of the two D waves: six rows of 16 columns, the two last columns
constexpr int WW_Y_IT = 4;                                           // float4s of the dY block per thread of the two V waves: its four rows

struct WwArgs {
    const float* x; const float* dy; float* ws; float* wsb;
    const float* x1;                                                 // second source of a virtual concatenation along cin: channels c0 .. cin - 1 (c0 = cin: none)
    int ldx1, c0;
    int ldx, ldy, B, H, W, cin, cout;
    int n_co, n_ci, S, gx, gy, n_groups, wide, coP, ciP;
};

typedef float ww_f2 __attribute__((ext_vector_type(2)));
// (s_nop 1: hipcc places no wait states between a VALU / v_accvgpr_write it put in front of an asm statement and the MFMA reading that register; free next to a 64-cycle MFMA)
#define WW_MFMA_ASM(k) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(accr[(k) % 9]) : "v"(a_), "v"(b_));
#define WW_LDRAW(p) (*reinterpret_cast<const ww_f2*>(p))
#define WW_ST2(p, v) *reinterpret_cast<ww_f2*>(p) = (v)
// Rows 3 RH .. 3 RH + 2 of B^T applied to six values / all six rows (float2: two channels at once, packed instructions)
template <int RH>
__device__ __forceinline__ void ww_bt3(const ww_f2 (&d)[6], ww_f2 (&t)[3]) {
    if constexpr (RH == 0) {
        const ww_f2 p = d[4] - 4.0f * d[2], q = d[3] - 4.0f * d[1];
        t[0] = 4.0f * d[0] - 5.0f * d[2] + d[4];
        t[1] = p + q;
        t[2] = p - q;
    } else {
        const ww_f2 p = d[4] - d[2], q = d[3] - d[1];
        t[0] = p + 2.0f * q;
        t[1] = p - 2.0f * q;
        t[2] = 4.0f * d[1] - 5.0f * d[3] + d[5];
    }
}
__device__ __forceinline__ void ww_bt6(const ww_f2 (&d)[6], ww_f2 (&t)[6]) {
    const ww_f2 p = d[4] - 4.0f * d[2], q = d[3] - 4.0f * d[1], r = d[4] - d[2], s = d[3] - d[1];
    t[0] = 4.0f * d[0] - 5.0f * d[2] + d[4];
    t[1] = p + q;
    t[2] = p - q;
    t[3] = r + 2.0f * s;
    t[4] = r - 2.0f * s;
    t[5] = 4.0f * d[1] - 5.0f * d[3] + d[5];
}
// Rows 3 RH .. 3 RH + 2 of A (6 x 4) applied to four values / all six rows
template <int RH>
__device__ __forceinline__ void ww_a3(const ww_f2 (&y)[4], ww_f2 (&o)[3]) {
    if constexpr (RH == 0) {
        const ww_f2 e = y[0] + y[2], f = y[1] + y[3];
        o[0] = y[0];
        o[1] = e + f;
        o[2] = e - f;
    } else {
        const ww_f2 g4 = y[0] + 4.0f * y[2], h = 2.0f * y[1] + 8.0f * y[3];
        o[0] = g4 + h;
        o[1] = g4 - h;
        o[2] = y[3];
    }
}
__device__ __forceinline__ void ww_a6(const ww_f2 (&y)[4], ww_f2 (&o)[6]) {
    const ww_f2 e = y[0] + y[2], f = y[1] + y[3], g4 = y[0] + 4.0f * y[2], h = 2.0f * y[1] + 8.0f * y[3];
    o[0] = y[0];
    o[1] = e + f;
    o[2] = e - f;
    o[3] = g4 + h;
    o[4] = g4 - h;
    o[5] = y[3];
}

// One wave of the Winograd-domain kernel.  ISV: the wave transforms input patches (V) and stages the dY block; otherwise it transforms dY blocks (D)
// and stages the input halo.  RH: which three of the six transform rows it produces.  Every wave also owns nine of the 36 positions' accumulators.
// The pipeline runs in PHASES from barrier to barrier; phase p holds, as ONE instruction stream of 18 slots -- an MFMA (64 cycles of the matrix
// pipe) followed by the few VALU / LDS instructions that fit beside it (an LDS write issued right behind an MFMA costs ~2 cycles instead of its 32;
// on their own the transforms' LDS writes alone take as long as the MFMAs: profiles/r5_wgrad_wino.txt):
//     MFMAs 15..17 of group p - 1 (operand registers of set (p - 1) & 1), then MFMAs 0..14 of group p (set p & 1)
//     R(p)    the operands of group p: VD[p & 1] -> register set p & 1, three slots ahead of their first MFMA
//     X(p+1)  this thread's half transform of group p + 1: raw[(p + 1) & 1] -> VD[(p + 1) & 1]
//     W(p+2)  its staged float4s of group p + 2: staging set p & 1 -> raw[p & 1]
//     L(p+3)  the requests for group p + 3 into staging set (p + 1) & 1 (a whole phase in flight)
// Everything indexed by the parity of p is a compile-time constant of the two copies of the phase (register sets, LDS offsets as immediates).
// Phases -2 and -1 fill the pipeline (no MFMAs); past the last group X / W / L run on clamped, unused data.
template <bool ISV, int RH>
__device__ __forceinline__ void ww_wave(const WwArgs& a, float* sm, const int wave, const int lane) {
    constexpr int NIT = ISV ? WW_Y_IT : WW_X_IT;
    const int col = lane & 31, half = lane >> 5;
    int bid = blockIdx.x;
    const int s = bid % a.S;  bid /= a.S;
    const int cib = bid % a.n_ci, cob = bid / a.n_ci;
    const int co0 = cob * WW_CB, ci0 = cib * WW_CB;
    const int g_lo = (int)((long)s * a.n_groups / a.S), g_hi = (int)((long)(s + 1) * a.n_groups / a.S), n = g_hi - g_lo;

    f32x16 acc[9];
#pragma unroll
    for (int p = 0; p < 9; ++p) acc[p] = nd_zero16();
    // the bias gradient rides on the operands: position (1, 1) of D = A dY A^T is the plain sum of the tile's 16 values (row 1 of A is all ones), and
    // wave 0 reads D[7][cout][its two tiles] as an MFMA operand anyway
    const bool do_bias = ISV && RH == 0 && a.wsb != nullptr && cib == 0;
    float bsum = 0.0f;

    // ---- staging: a thread of the V waves = (column c of the 4 x 16 dY block, channel quad), its items the four rows; a thread of the D waves =
    // (column c < 16 of the 6 x 18 halo, quad), items 0..5 the rows, item 6 = the two last columns as (row, column, quad) on 96 lanes.  The row goes
    // into the scalar offset of the request, the group into the resource's base, so the lane offsets are constants; what lies outside the image is
    // requested out of range (-> zeros): one select for the left edge, one each for the top and the bottom row, a mask test for item 6.
    const int u = (wave & 1) * 64 + lane;                            // 0..127 within the wave pair
    const bool second = !ISV && ci0 >= a.c0;                         // the workgroup's cin block lies in the second source (blocks never straddle: c0 % 32 == 0)
    const int ld = ISV ? a.ldy : second ? a.ldx1 : a.ldx;
    const float* const src = ISV ? a.dy : second ? a.x1 : a.x;
    const int cs0 = ISV ? co0 : second ? ci0 - a.c0 : ci0;           // first channel of the block within its source
    const int sc = u >> 3, sq = u & 7;
    const int OOB = (int)0x80000000;
    // channel counts that are multiples of 16 only (d = 48: 48 / 96 / ...): the quads past the tensor's last channel are requested out of range (zeros)
    const int c_end = ISV ? a.cout : second ? a.cin - a.c0 : a.c0;   // channels of this source
    const int vl = cs0 + 4 * sq < c_end ? (sc * ld + cs0 + 4 * sq) * 4 : OOB;
    const int vl_left = (!ISV && sc == 0) ? OOB : vl;                // left image edge: halo column 0 is padding
    const int r6 = u >> 4, c6 = 16 + ((u >> 3) & 1);
    const int v6 = ((r6 * a.W + c6) * ld + cs0 + 4 * sq) * 4;
    const int code6 = (c6 == 17 ? 1 : 0) | (r6 == 0 ? 2 : 0) | (r6 == 5 ? 4 : 0) | ((u >= 96 || cs0 + 4 * sq >= c_end) ? 8 : 0);
    const int lds_c = (ISV ? WW_XF : 0) + sc * WW_PS + 4 * sq;       // + row * (16 | 18) * WW_PS
    const int lds_6 = (r6 * WW_HC + c6) * WW_PS + 4 * sq;
    const int row_bytes = a.W * ld * 4;
    int lg = g_lo, lb, lgy, lgx;                                     // the group the next request is for (scalar state; past the last group it stays there)
    {
        const int gxy = a.gx * a.gy;
        lb = lg / gxy;  const int r_ = lg - lb * gxy;  lgy = r_ / a.gx;  lgx = r_ - lgy * a.gx;
    }
    f32x4 sr[2][NIT];
    auto request = [&](auto set_) {
        constexpr int SET = decltype(set_)::value;
        const int y0 = lgy * WW_GH, x0 = lgx * WW_GW;
        const long org = ((long)(lb * a.H + y0 - (ISV ? 0 : 1)) * a.W + x0 - (ISV ? 0 : 1)) * ld;      // (for the halo: may lie in front of the tensor; never dereferenced there)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src) + org, 0, 0x7fffffff, 0x00020000);
        if constexpr (ISV) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) sr[SET][it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vl, it * row_bytes, 0));
        } else {
            const bool left = x0 == 0, right = x0 + WW_GW == a.W, top = y0 == 0, bottom = y0 + WW_GH == a.H;
            const int vm = left ? vl_left : vl;
#pragma unroll
            for (int it = 0; it < 6; ++it) {
                const int vo = (it == 0 && top) || (it == 5 && bottom) ? OOB : vm;
                sr[SET][it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, it * row_bytes, 0));
            }
            const int edge = (right ? 1 : 0) | (top ? 2 : 0) | (bottom ? 4 : 0) | 8;
            sr[SET][6] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (code6 & edge) ? OOB : v6, 0, 0));
        }
        if (lg + 1 < g_hi) {
            ++lg;
            if (++lgx == a.gx) { lgx = 0;  if (++lgy == a.gy) { lgy = 0;  ++lb; } }
        }
    };

    // ---- transform role: thread (tile t, channel pair cp) of the wave's row half
    const int t = lane >> 4, cp = lane & 15;
    const int rd_off = (ISV ? 0 : WW_XF) + 4 * t * WW_PS + 2 * cp;   // in a raw buffer
    const int wr_off = (ISV ? 0 : WW_VDF) + t * WW_CB + 2 * cp;      // in a (V, D) pair: + pos * 128
    const int op_off = (9 * wave * WW_NT + half) * WW_CB + col;      // operands of this wave's positions: + p * 128 (+ 64: the second tile of the lane's K slot)

    f32x2 av[2][9], bv[2][9];
#pragma unroll
    for (int p = 0; p < 9; ++p) { av[1][p] = f32x2{0.0f, 0.0f};  bv[1][p] = f32x2{0.0f, 0.0f}; }       // phase 0 runs "MFMAs 15..17 of group -1" on these
    f32x16 (&accr)[9] = acc;                                         // (an asm operand alone does not capture)
#ifdef WW_STAMP                  // diagnostic (tools/wgrad_clock.py): shader-clock stamps per wave -> slots / barrier wait per phase, clock under load
    const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long st_t0 = 0, st_t2 = 0, st_b = 0, st_c = 0;
#endif

#define WW_MFMA(set, k)                                                                                                        \
        if constexpr (MF) {                                                                                                    \
            const float a_ = av[set][(k) % 9][(k) / 9], b_ = bv[set][(k) % 9][(k) / 9];                                        \
            WW_MFMA_ASM(k)                                                                                                     \
        }
#define WW_SB() __builtin_amdgcn_sched_barrier(0)
    auto phase = [&](auto par_, auto with_mfma) {
        constexpr int P = decltype(par_)::value, Q = P ^ 1;
        constexpr bool MF = decltype(with_mfma)::value;
        const float* const rbuf = sm + Q * WW_RAWF + rd_off;                                         // X(p+1) reads raw[(p+1) & 1]
        float* const wbuf = sm + P * WW_RAWF;                                                         // W(p+2) writes raw[p & 1]
        const float* const opv = sm + 2 * WW_RAWF + P * (2 * WW_VDF) + op_off;                        // R(p) reads VD[p & 1]
        float* const vdw = sm + 2 * WW_RAWF + Q * (2 * WW_VDF) + wr_off + (3 * RH * 6) * 128;        // X(p+1) writes its rows of VD[(p+1) & 1]
        auto commit = [&](auto ii, auto hh) {                        // W: half of one staged float4 -> the raw buffer.  8-byte writes: behind an MFMA a ds_write_b64
            constexpr int it = decltype(ii)::value, h = decltype(hh)::value;      // costs the wave ~2 cycles, a ds_write_b128 ~80 (tools/wgrad_clock.py)
            const ww_f2 v = h ? ww_f2{sr[P][it][2], sr[P][it][3]} : ww_f2{sr[P][it][0], sr[P][it][1]};
            float* d;
            if constexpr (ISV) d = wbuf + lds_c + it * WW_GW * WW_PS;
            else if constexpr (it < 6) d = wbuf + lds_c + it * WW_HC * WW_PS;
            else d = wbuf + lds_6;
            if (ISV || it < 6 || u < 96) *reinterpret_cast<ww_f2*>(d + 2 * h) = v;
        };
#define WW_I(k) std::integral_constant<int, (k)>{}
#define WW_C(it, h) commit(WW_I(it), WW_I(h))
#ifdef WW_STAMP
        if constexpr (MF) st_t0 = __builtin_amdgcn_s_memtime();
#endif
        // ---- slots 15..17 of the previous group: this group's operands, the first staged float4s, the first raw reads, the next requests
        WW_MFMA(Q, 15)
        if constexpr (MF) {
#pragma unroll
            for (int p = 0; p < 9; ++p) {
                av[P][p][0] = opv[WW_VDF + p * 128];  av[P][p][1] = opv[WW_VDF + p * 128 + 64];
                bv[P][p][0] = opv[p * 128];           bv[P][p][1] = opv[p * 128 + 64];
            }
        }
        if constexpr (ISV) { WW_C(0, 0); } else { WW_C(0, 0);  WW_C(3, 1); }
        WW_SB();
        if constexpr (ISV) {
            ww_f2 X6[6][6], T[3][6];
            WW_MFMA(Q, 16)  WW_C(0, 1);
#pragma unroll
            for (int r = 0; r < 6; ++r) X6[0][r] = WW_LDRAW(rbuf + (r * WW_HC) * WW_PS);
            WW_SB();
            WW_MFMA(Q, 17)  WW_C(1, 0);  request(WW_I(Q));  WW_SB();
            // slots 0..5: column k of the patch through the wave's three rows of B^T (the next column's six reads in flight)
#define WW_COL(k, E)                                                                                                           \
            WW_MFMA(P, k)                                                                                                      \
            if constexpr ((k) < 5) {                                                                                           \
                _Pragma("unroll") for (int r = 0; r < 6; ++r) X6[(k) + 1][r] = WW_LDRAW(rbuf + (r * WW_HC + (k) + 1) * WW_PS); \
            }                                                                                                                  \
            { ww_f2 t3[3];  ww_bt3<RH>(X6[k], t3);  T[0][k] = t3[0];  T[1][k] = t3[1];  T[2][k] = t3[2]; }                      \
            E;  WW_SB();
            WW_COL(0, WW_C(1, 1)) WW_COL(1, WW_C(2, 0)) WW_COL(2, WW_C(2, 1)) WW_COL(3, WW_C(3, 0)) WW_COL(4, WW_C(3, 1))
            WW_COL(5, if (MF && do_bias) bsum += av[P][7][0] + av[P][7][1])
#undef WW_COL
            // slots 6..14: row r through all of B^T, its six 8-byte writes two per slot
#define WW_ROW(r)                                                                                                              \
            {                                                                                                                  \
                ww_f2 v[6];                                                                                                    \
                float* const o = vdw + ((r) * 6) * 128;                                                                        \
                WW_MFMA(P, 6 + 3 * (r))  ww_bt6(T[r], v);  WW_ST2(o, v[0]);  WW_ST2(o + 128, v[1]);  WW_SB();                  \
                WW_MFMA(P, 7 + 3 * (r))  WW_ST2(o + 256, v[2]);  WW_ST2(o + 384, v[3]);  WW_SB();                              \
                WW_MFMA(P, 8 + 3 * (r))  WW_ST2(o + 512, v[4]);  WW_ST2(o + 640, v[5]);  WW_SB();                              \
            }
            WW_ROW(0) WW_ROW(1) WW_ROW(2)
#undef WW_ROW
        } else {
            ww_f2 Y4[4][4], U[3][4];
            WW_MFMA(Q, 16)  WW_C(1, 0);  WW_C(4, 1);
#pragma unroll
            for (int m = 0; m < 4; ++m) Y4[0][m] = WW_LDRAW(rbuf + (m * WW_GW) * WW_PS);
            WW_SB();
            WW_MFMA(Q, 17)  WW_C(2, 0);  WW_C(5, 1);  request(WW_I(Q));  WW_SB();
            // slots 0..3: column k of the dY tile through the wave's three rows of A, one staged float4 each
#define WW_COL(k)                                                                                                              \
            WW_MFMA(P, k)                                                                                                      \
            if constexpr ((k) < 3) {                                                                                           \
                _Pragma("unroll") for (int m = 0; m < 4; ++m) Y4[(k) + 1][m] = WW_LDRAW(rbuf + (m * WW_GW + (k) + 1) * WW_PS); \
            }                                                                                                                  \
            { ww_f2 o3[3];  ww_a3<RH>(Y4[k], o3);  U[0][k] = o3[0];  U[1][k] = o3[1];  U[2][k] = o3[2]; }                       \
            WW_C(3 + (k), 0);  WW_C(((k) + 6) % 7, 1);  WW_SB();
            WW_COL(0) WW_COL(1) WW_COL(2) WW_COL(3)
#undef WW_COL
            // slots 4..12: row r through all of A, its six 8-byte writes two per slot; 13, 14: MFMAs only
#define WW_ROW(r)                                                                                                              \
            {                                                                                                                  \
                ww_f2 v[6];                                                                                                    \
                float* const o = vdw + ((r) * 6) * 128;                                                                        \
                WW_MFMA(P, 4 + 3 * (r))  ww_a6(U[r], v);  WW_ST2(o, v[0]);  WW_ST2(o + 128, v[1]);  WW_SB();                   \
                WW_MFMA(P, 5 + 3 * (r))  WW_ST2(o + 256, v[2]);  WW_ST2(o + 384, v[3]);  WW_SB();                              \
                WW_MFMA(P, 6 + 3 * (r))  WW_ST2(o + 512, v[4]);  WW_ST2(o + 640, v[5]);  WW_SB();                              \
            }
            WW_ROW(0) WW_ROW(1) WW_ROW(2)
#undef WW_ROW
            WW_MFMA(P, 13)  WW_SB();
            WW_MFMA(P, 14)  WW_SB();
        }
#undef WW_I
#undef WW_C
#ifdef WW_STAMP
        if constexpr (MF) st_t2 = __builtin_amdgcn_s_memtime();
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // this wave's LDS writes have landed (its global requests stay in flight) ...
        __builtin_amdgcn_s_barrier();                                // ... and so have every other wave's; all LDS reads of the phase are done
#ifdef WW_STAMP
        if constexpr (MF) { const unsigned long long t3 = __builtin_amdgcn_s_memtime();  st_b += st_t2 - st_t0;  st_c += t3 - st_t2; }
#endif
    };

    request(std::integral_constant<int, 0>{});                       // group 0
    phase(std::integral_constant<int, 0>{}, std::false_type{});
    phase(std::integral_constant<int, 1>{}, std::false_type{});
    int p = 0;
    for (; p + 1 < n; p += 2) {
        phase(std::integral_constant<int, 0>{}, std::true_type{});
        phase(std::integral_constant<int, 1>{}, std::true_type{});
        // leaving the loop hipcc may move accumulators between register files for the code behind it -- and it does not know that the asm MFMAs'
        // results are still in flight (n = 3 read stale accumulators there): let them land first
        if (p + 3 >= n) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    }
    {
        constexpr bool MF = true;
        if (p < n) {
            phase(std::integral_constant<int, 0>{}, std::true_type{});
            WW_MFMA(0, 15) WW_MFMA(0, 16) WW_MFMA(0, 17)
        } else {
            WW_MFMA(1, 15) WW_MFMA(1, 16) WW_MFMA(1, 17)
        }
    }
#undef WW_MFMA
#undef WW_SB

#ifdef WW_STAMP
    if (lane == 0) {
        float* dbg = a.ws + (size_t)a.S * 36 * a.cout * a.cin + (size_t)a.S * a.cout + ((size_t)blockIdx.x * 4 + wave) * 8;
        dbg[0] = (float)(__builtin_amdgcn_s_memtime() - st_c0);  dbg[1] = (float)(__builtin_amdgcn_s_memrealtime() - st_r0);
        dbg[2] = (float)n;  dbg[3] = 0.0f;  dbg[4] = (float)st_b;  dbg[5] = (float)st_c;
    }
#endif
    // ---- this workgroup's partial sums: ws[s][pos][co][ci]  (the MFMAs are asm statements: hipcc does not know their results are still in flight)
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int p9 = 0; p9 < 9; ++p9) asm volatile("" : "+a"(acc[p9]));
    const int coP = a.n_co * WW_CB, ciP = a.n_ci * WW_CB;
#pragma unroll
    for (int p9 = 0; p9 < 9; ++p9)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + nd_acc_row(r, lane), ci = ci0 + col;
            a.ws[(((size_t)s * 36 + wave * 9 + p9) * coP + co) * ciP + ci] = acc[p9][r];
        }
    if (do_bias) {                                                   // lane (cout, half) holds the sum over its two tiles of every group: the halves meet in lane order
        const float other = __shfl_down(bsum, 32);
        if (lane < WW_CB) a.wsb[(size_t)s * coP + co0 + lane] = bsum + other;
    }
}

__global__ __launch_bounds__(256, 1) void wgrad_wino_kernel(const WwArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63;     // scalar: the four roles are wave-uniform branches
    if (wave == 0) ww_wave<true, 0>(a, sm, wave, lane);
    else if (wave == 1) ww_wave<true, 1>(a, sm, wave, lane);
    else if (wave == 2) ww_wave<false, 0>(a, sm, wave, lane);
    else ww_wave<false, 1>(a, sm, wave, lane);
}

// ---- the same pipeline for a (64 couts x 32 cins) block on EIGHT waves, two per SIMD (taken when cout % 64 == 0).  What bounds the four-wave form
// is the LDS pipe (profiles/r5_wgrad_wino.txt): per tile group ~1.16 k cycles of LDS transfers that a lone wave per SIMD cannot overlap with its own
// 1.15 k cycles of MFMAs.  Here a group carries twice the MFMAs for 1.5x the LDS bytes (the V side is shared by both cout halves) and a wave's LDS
// waits are covered by its SIMD partner's MFMAs.  Wave w multiplies positions 9 (w & 3) .. + 8 of cout half w >> 2; roles besides: waves 0, 1
// transform the input patches (rows 0-2 / 3-5), waves 2, 3, 6, 7 the dY blocks (cout half x row half) and stage dY, waves 4, 5 stage the input halo.
// The accumulator volume per CU -- and with it the split and the partial sums -- is the four-wave form's.  256 registers per wave: eight
// accumulators in AGPRs, the ninth in ordinary registers (hipcc splits the file 128 / 128 when asm statements use AGPRs).  LDS: two raw buffers +
// ONE (V, D) pair (126720 bytes), so a phase p has two barriers:
//   [A]  the two MFMAs of position 8 of group p - 1 (operands still in registers), then all operands of group p -> registers
//   [B]  (the images may be overwritten) the sixteen MFMAs of positions 0..7 of group p; behind them the transform of group p + 1 -> the images,
//        the staged float4s of group p + 2 -> raw[p & 1], the requests for group p + 3
constexpr int W8_CO = 64;
constexpr int W8_PSY = 72;                                           // floats per staged dY pixel: 64 channels + 8
constexpr int W8_XF = WW_HR * WW_HC * WW_PS, W8_YF = WW_GH * WW_GW * W8_PSY, W8_RAWF = W8_XF + W8_YF;
constexpr int W8_VF = 36 * WW_NT * WW_CB, W8_DF = 36 * WW_NT * W8_CO, W8_VDF = W8_VF + W8_DF;
constexpr size_t W8_LDS = (size_t)(2 * W8_RAWF + W8_VDF) * sizeof(float);

enum { W8_V = 0, W8_D = 1, W8_S = 2 };

template <int ROLE, int RH>
__device__ __forceinline__ void ww8_wave(const WwArgs& a, float* sm, const int wave, const int lane) {
    constexpr int NIT = ROLE == W8_S ? WW_X_IT : ROLE == W8_D ? WW_Y_IT : 1;
    const int col = lane & 31, half = lane >> 5;
    const int pa = wave & 3, cb = wave >> 2;                         // MFMA role: positions 9 pa .., cout half cb
    int bid = blockIdx.x;
    const int s = bid % a.S;  bid /= a.S;
    const int cib = bid % a.n_ci, cob = bid / a.n_ci;
    const int co0 = cob * W8_CO, ci0 = cib * WW_CB;
    const int g_lo = (int)((long)s * a.n_groups / a.S), g_hi = (int)((long)(s + 1) * a.n_groups / a.S), n = g_hi - g_lo;

    f32x16 acc[9];
#pragma unroll
    for (int p = 0; p < 9; ++p) acc[p] = nd_zero16();
    // the bias gradient rides on the operands (position (1, 1) of D = the plain sum of the tile): the waves that own position 7, one per cout half
    const bool do_bias = pa == 0 && a.wsb != nullptr && cib == 0;
    float bsum = 0.0f;

    // ---- staging (ROLE S: the halo, as the D waves of the four-wave form; ROLE D: the 4 x 16 x 64 dY block, thread = (column, quad), items = rows)
    const int OOB = (int)0x80000000;
    const int u = ROLE == W8_D ? ((wave >> 2) * 2 + (wave & 1)) * 64 + lane : (wave & 1) * 64 + lane;      // 0..255 over the D waves, 0..127 over the S waves
    const bool second = ROLE != W8_D && ci0 >= a.c0;                 // (see ww_wave)
    const int ld = ROLE == W8_D ? a.ldy : second ? a.ldx1 : a.ldx;
    const float* const src = ROLE == W8_D ? a.dy : second ? a.x1 : a.x;
    const int cs0 = ROLE == W8_D ? co0 : second ? ci0 - a.c0 : ci0;
    const int sc = ROLE == W8_D ? u >> 4 : u >> 3, sq = ROLE == W8_D ? u & 15 : u & 7;
    const int vl = (sc * ld + cs0 + 4 * sq) * 4;
    const int vl_left = (ROLE == W8_S && sc == 0) ? OOB : vl;
    const int r6 = u >> 4, c6 = 16 + ((u >> 3) & 1);
    const int v6 = ((r6 * a.W + c6) * ld + cs0 + 4 * (u & 7)) * 4;
    const int code6 = (c6 == 17 ? 1 : 0) | (r6 == 0 ? 2 : 0) | (r6 == 5 ? 4 : 0) | (u >= 96 ? 8 : 0);
    const int st_c = ROLE == W8_D ? W8_XF + sc * W8_PSY + 4 * sq : sc * WW_PS + 4 * sq;      // staged float4s in a raw buffer: + row * (16 PSY | 18 PS)
    const int st_6 = (r6 * WW_HC + c6) * WW_PS + 4 * (u & 7);
    const int row_bytes = a.W * ld * 4;
    int lg = g_lo, lb, lgy, lgx;
    {
        const int gxy = a.gx * a.gy;
        lb = lg / gxy;  const int r_ = lg - lb * gxy;  lgy = r_ / a.gx;  lgx = r_ - lgy * a.gx;
    }
    f32x4 sr[NIT];
    auto request = [&]() {
        if constexpr (ROLE != W8_V) {
            const int y0 = lgy * WW_GH, x0 = lgx * WW_GW;
            const long org = ((long)(lb * a.H + y0 - (ROLE == W8_S ? 1 : 0)) * a.W + x0 - (ROLE == W8_S ? 1 : 0)) * ld;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src) + org, 0, 0x7fffffff, 0x00020000);
            if constexpr (ROLE == W8_D) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) sr[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vl, it * row_bytes, 0));
            } else {
                const bool left = x0 == 0, right = x0 + WW_GW == a.W, top = y0 == 0, bottom = y0 + WW_GH == a.H;
                const int vm = left ? vl_left : vl;
#pragma unroll
                for (int it = 0; it < 6; ++it) {
                    const int vo = (it == 0 && top) || (it == 5 && bottom) ? OOB : vm;
                    sr[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, it * row_bytes, 0));
                }
                const int edge = (right ? 1 : 0) | (top ? 2 : 0) | (bottom ? 4 : 0) | 8;
                sr[6] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (code6 & edge) ? OOB : v6, 0, 0));
            }
            if (lg + 1 < g_hi) {
                ++lg;
                if (++lgx == a.gx) { lgx = 0;  if (++lgy == a.gy) { lgy = 0;  ++lb; } }
            }
        }
    };
    // ---- transform role: thread (tile t, channel pair cp) of the wave's row half (D: of its cout half as well)
    const int t = lane >> 4, cp = lane & 15;
    const int rd_off = ROLE == W8_D ? W8_XF + 4 * t * W8_PSY + 32 * (wave >> 2) + 2 * cp : 4 * t * WW_PS + 2 * cp;      // its patch in a raw buffer
    constexpr int WSTEP = WW_NT * (ROLE == W8_D ? W8_CO : WW_CB);    // floats between two positions of the image this wave writes
    const int wr_off = (ROLE == W8_D ? W8_VF + t * W8_CO + 32 * (wave >> 2) : t * WW_CB) + 2 * cp + (3 * RH * 6) * WSTEP;
    const int opv_off = (9 * pa * WW_NT + half) * WW_CB + col;                     // V operands of this wave's positions: + p * 128 (+ 64)
    const int opd_off = W8_VF + (9 * pa * WW_NT + half) * W8_CO + 32 * cb + col;   // D operands: + p * 256 (+ 128)

    f32x2 oa[9], ob[9];                                              // the group's operands
    oa[7] = oa[8] = f32x2{0.0f, 0.0f};  ob[7] = ob[8] = f32x2{0.0f, 0.0f};      // phase 0 runs "positions 7, 8 of group -1" on these
    f32x16 (&accr)[9] = acc;                                         // (an asm operand alone does not capture)

#define W8_MFMA(p, j)                                                                                                          \
        if constexpr (MF) {                                                                                                    \
            const float a_ = oa[p][j], b_ = ob[p][j];                                                                          \
            if constexpr ((p) == 8) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(accr[8]) : "v"(a_), "v"(b_));     \
            else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(accr[p]) : "v"(a_), "v"(b_));         \
        }
#define W8_LDOP(p)                                                                                                             \
        if constexpr (MF) {                                                                                                    \
            oa[p][0] = sm[opd_off + (p) * 256];  oa[p][1] = sm[opd_off + (p) * 256 + 128];                                     \
            ob[p][0] = sm[opv_off + (p) * 128];  ob[p][1] = sm[opv_off + (p) * 128 + 64];                                      \
        }
#define WW_SB() __builtin_amdgcn_sched_barrier(0)
#define W8_BARRIER() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  __builtin_amdgcn_s_barrier(); }
#define W8_I(k) std::integral_constant<int, (k)>{}
#ifdef WW_STAMP                  // diagnostic (tools/wgrad_clock.py): cycles per phase = segment 1 / wait at B / segment 2 / wait at A
    const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long st_t0 = 0, st_t1 = 0, st_t2 = 0, st_t3 = 0, st_s1 = 0, st_wb = 0, st_s2 = 0, st_wa = 0;
#endif
    int par = 0;
    auto phase = [&](auto with_mfma) {
        constexpr bool MF = decltype(with_mfma)::value;
#ifdef WW_STAMP
        if constexpr (MF) st_t0 = __builtin_amdgcn_s_memtime();
#endif
        const float* const rd = sm + W8_VDF + (par ^ 1) * W8_RAWF + rd_off;      // X(p+1) reads raw[(p+1) & 1]  (the images sit at LDS address 0: 16-bit offsets reach all of them)
        float* const wbuf = sm + W8_VDF + par * W8_RAWF;             // W(p+2) writes raw[p & 1]
        float* const vdw = sm + wr_off;                              // X(p+1) writes its rows of the images
        auto commit = [&](auto ii) {                                 // one staged float4 -> the raw buffer
            constexpr int it = decltype(ii)::value;
            if constexpr (ROLE == W8_D) nd_st4(wbuf + st_c + it * WW_GW * W8_PSY, sr[it]);
            else if constexpr (ROLE == W8_S && it < 6) nd_st4(wbuf + st_c + it * WW_HC * WW_PS, sr[it]);
            else if constexpr (ROLE == W8_S) { if (u < 96) nd_st4(wbuf + st_6, sr[6]); }
        };
        // ================= segment 1 (behind barrier A): positions 7, 8 of the previous group, then this group's operands
        W8_MFMA(7, 0)  W8_MFMA(8, 0)  W8_MFMA(7, 1)  W8_MFMA(8, 1)
        W8_LDOP(0) W8_LDOP(1) W8_LDOP(2) W8_LDOP(3) W8_LDOP(4) W8_LDOP(5) W8_LDOP(6) W8_LDOP(7) W8_LDOP(8)
#ifdef WW_STAMP
        if constexpr (MF) st_t1 = __builtin_amdgcn_s_memtime();
#endif
        W8_BARRIER()                                                 // B: every wave holds its operands -- the images may be overwritten
#ifdef WW_STAMP
        if constexpr (MF) st_t2 = __builtin_amdgcn_s_memtime();
#endif
        if (MF && do_bias) bsum += oa[7][0] + oa[7][1];
        // ================= segment 2: fourteen MFMAs (K step 0 of positions 0..6, then K step 1); slot m = MFMA m + what fits behind it
#define W8_M(m) if constexpr ((m) < 14) { W8_MFMA((m) % 7, (m) / 7) }
        ww_f2 T[3][ROLE == W8_V ? 6 : 4];
        if constexpr (ROLE == W8_V) {
            ww_f2 X6[6][6];
#pragma unroll
            for (int r = 0; r < 6; ++r) X6[0][r] = WW_LDRAW(rd + (r * WW_HC) * WW_PS);
            WW_SB();
#define W8_COL(k)                                                                                                              \
            W8_M(k)                                                                                                            \
            { ww_f2 t3[3];  ww_bt3<RH>(X6[k], t3);  T[0][k] = t3[0];  T[1][k] = t3[1];  T[2][k] = t3[2]; }                      \
            asm volatile("" ::: "memory");                                                                                     \
            if constexpr ((k) < 5) {                                                                                           \
                _Pragma("unroll") for (int r = 0; r < 6; ++r) X6[(k) + 1][r] = WW_LDRAW(rd + (r * WW_HC + (k) + 1) * WW_PS);   \
            }                                                                                                                  \
            WW_SB();
            W8_COL(0) W8_COL(1) W8_COL(2) W8_COL(3) W8_COL(4) W8_COL(5)
#undef W8_COL
#define W8_ROW(r, m0)                                                                                                          \
            {                                                                                                                  \
                ww_f2 v[6];                                                                                                    \
                float* const o = vdw + ((r) * 6) * WSTEP;                                                                      \
                W8_M(m0)  ww_bt6(T[r], v);  WW_ST2(o, v[0]);  WW_ST2(o + WSTEP, v[1]);  WW_SB();                               \
                W8_M((m0) + 1)  WW_ST2(o + 2 * WSTEP, v[2]);  WW_ST2(o + 3 * WSTEP, v[3]);  WW_SB();                           \
                W8_M((m0) + 2)  WW_ST2(o + 4 * WSTEP, v[4]);  WW_ST2(o + 5 * WSTEP, v[5]);  WW_SB();                           \
            }
            W8_ROW(0, 6) W8_ROW(1, 9) W8_ROW(2, 12)
#undef W8_ROW
        } else if constexpr (ROLE == W8_D) {
            ww_f2 Y4[4][4];
#pragma unroll
            for (int m = 0; m < 4; ++m) Y4[0][m] = WW_LDRAW(rd + (m * WW_GW) * W8_PSY);
            WW_SB();
#define W8_COL(k)                                                                                                              \
            W8_M(k)                                                                                                            \
            if constexpr ((k) < 3) {                                                                                           \
                _Pragma("unroll") for (int m = 0; m < 4; ++m) Y4[(k) + 1][m] = WW_LDRAW(rd + (m * WW_GW + (k) + 1) * W8_PSY);  \
            }                                                                                                                  \
            { ww_f2 o3[3];  ww_a3<RH>(Y4[k], o3);  T[0][k] = o3[0];  T[1][k] = o3[1];  T[2][k] = o3[2]; }                       \
            commit(W8_I(k));  WW_SB();
            W8_COL(0) W8_COL(1) W8_COL(2) W8_COL(3)
#undef W8_COL
#define W8_ROW(r, m0)                                                                                                          \
            {                                                                                                                  \
                ww_f2 v[6];                                                                                                    \
                float* const o = vdw + ((r) * 6) * WSTEP;                                                                      \
                W8_M(m0)  ww_a6(T[r], v);  WW_ST2(o, v[0]);  WW_ST2(o + WSTEP, v[1]);  WW_SB();                                \
                W8_M((m0) + 1)  WW_ST2(o + 2 * WSTEP, v[2]);  WW_ST2(o + 3 * WSTEP, v[3]);  WW_SB();                           \
                W8_M((m0) + 2)  WW_ST2(o + 4 * WSTEP, v[4]);  WW_ST2(o + 5 * WSTEP, v[5]);  WW_SB();                           \
            }
            W8_ROW(0, 4) W8_ROW(1, 7) W8_ROW(2, 10)
#undef W8_ROW
            W8_M(13)  request();  WW_SB();
        } else {
            W8_M(0)  commit(W8_I(0));  WW_SB();
            W8_M(1)  commit(W8_I(1));  WW_SB();
            W8_M(2)  commit(W8_I(2));  WW_SB();
            W8_M(3)  commit(W8_I(3));  WW_SB();
            W8_M(4)  commit(W8_I(4));  WW_SB();
            W8_M(5)  commit(W8_I(5));  WW_SB();
            W8_M(6)  commit(W8_I(6));  WW_SB();
            W8_M(7)  request();  WW_SB();
            W8_M(8) WW_SB(); W8_M(9) WW_SB(); W8_M(10) WW_SB(); W8_M(11) WW_SB(); W8_M(12) WW_SB(); W8_M(13) WW_SB();
        }
#undef W8_M
#ifdef WW_STAMP
        if constexpr (MF) st_t3 = __builtin_amdgcn_s_memtime();
#endif
        W8_BARRIER()                                                 // A: the images of the next group and the raw buffer of the one after it are complete
#ifdef WW_STAMP
        if constexpr (MF) { const unsigned long long t4 = __builtin_amdgcn_s_memtime();  st_s1 += st_t1 - st_t0;  st_wb += st_t2 - st_t1;  st_s2 += st_t3 - st_t2;  st_wa += t4 - st_t3; }
#endif
        par ^= 1;
    };

    request();                                                       // group 0
    phase(std::false_type{});                                        // phase -2: stages group 0, requests group 1 (its transform works on nothing yet)
    phase(std::false_type{});                                        // phase -1: transforms group 0, stages group 1, requests group 2
    for (int p = 0; p < n; ++p) {
        phase(std::true_type{});
        // leaving the loop hipcc may move accumulators between register files -- and it does not know that the asm MFMAs' results are still in flight
        if (p + 1 >= n) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    }
    {
        constexpr bool MF = true;
        W8_MFMA(7, 0) W8_MFMA(8, 0) W8_MFMA(7, 1) W8_MFMA(8, 1)
    }
#undef W8_MFMA
#undef W8_LDOP
#undef WW_SB
#undef W8_BARRIER
#undef W8_I

#ifdef WW_STAMP
    if (lane == 0) {
        float* dbg = a.ws + (size_t)a.S * 36 * a.cout * a.cin + (size_t)a.S * a.cout + ((size_t)blockIdx.x * 8 + wave) * 8;
        dbg[0] = (float)(__builtin_amdgcn_s_memtime() - st_c0);  dbg[1] = (float)(__builtin_amdgcn_s_memrealtime() - st_r0);
        dbg[2] = (float)n;  dbg[3] = (float)st_s1;  dbg[4] = (float)st_wb;  dbg[5] = (float)st_s2;  dbg[6] = (float)st_wa;
    }
#endif
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int p9 = 0; p9 < 8; ++p9) asm volatile("" : "+a"(acc[p9]));
    asm volatile("" : "+v"(acc[8]));
#pragma unroll
    for (int p9 = 0; p9 < 9; ++p9)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + 32 * cb + nd_acc_row(r, lane), ci = ci0 + col;
            a.ws[(((size_t)s * 36 + pa * 9 + p9) * a.cout + co) * a.cin + ci] = acc[p9][r];
        }
    if (do_bias) {
        const float other = __shfl_down(bsum, 32);
        if (lane < WW_CB) a.wsb[(size_t)s * a.cout + co0 + 32 * cb + lane] = bsum + other;
    }
}

__global__ __launch_bounds__(512, 1) void wgrad_wino8_kernel(const WwArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    switch (wave) {
        case 0: ww8_wave<W8_V, 0>(a, sm, wave, lane); break;
        case 1: ww8_wave<W8_V, 1>(a, sm, wave, lane); break;
        case 4: case 5: ww8_wave<W8_S, 0>(a, sm, wave, lane); break;
        case 2: case 6: ww8_wave<W8_D, 0>(a, sm, wave, lane); break;
        default: ww8_wave<W8_D, 1>(a, sm, wave, lane); break;
    }
}

// The S partials [s][pos][co][ci] (+ the bias partials [s][co]) summed in split order into the split-0 slot: a thread owns one element -- 36 cout cin
// of them, every load coalesced.  (One thread per (co, ci) doing all 36 S-deep sums had 4096 threads reading 38 MB for a 64 -> 64 layer.)
__global__ __launch_bounds__(256) void wgrad_wino_sum_kernel(float* __restrict__ ws, float* __restrict__ wsb, int S, size_t n_w, int n_b) {
    const size_t total = n_w + (wsb ? (size_t)n_b : 0);
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < total; j += (size_t)gridDim.x * blockDim.x) {
        float* p = j < n_w ? ws + j : wsb + (j - n_w);
        const size_t stride = j < n_w ? n_w : (size_t)n_b;
        float sum = p[0];
        int s_ = 1;
        for (; s_ + 8 <= S; s_ += 8) {                                   // eight loads in flight, added in split order
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = p[(size_t)(s_ + k) * stride];
#pragma unroll
            for (int k = 0; k < 8; ++k) sum += v[k];
        }
        for (; s_ < S; ++s_) sum += p[(size_t)s_ * stride];
        p[0] = sum;
    }
}

// dW (OIHW) = G^T M G from the summed positions M[pos][co][ci] (split-0 slot); a thread owns one (co, ci): 36 coalesced reads, 9 results; db from the summed bias row
__global__ __launch_bounds__(256) void wgrad_wino_reduce_kernel(const float* __restrict__ ws, const float* __restrict__ wsb, float* __restrict__ dw,
                                                                float* __restrict__ db, int cin, int cout, int coP, int ciP) {
    constexpr float G[6][3] = {{0.25f, 0.f, 0.f}, {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                               {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
    const size_t plane = (size_t)cout * cin, total = plane + (db ? cout : 0), planeP = (size_t)coP * ciP;     // (the workspace planes are padded to whole blocks)
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < total; j += (size_t)gridDim.x * blockDim.x) {
        if (j >= plane) { db[j - plane] = wsb[j - plane];  continue; }
        const size_t jp = (j / cin) * ciP + j % cin;
        float M[36];
#pragma unroll
        for (int pos = 0; pos < 36; ++pos) M[pos] = ws[(size_t)pos * planeP + jp];
        float R[3][6];                                               // G^T M: (3 x 6)(6 x 6)
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int jj = 0; jj < 6; ++jj) {
                float v = 0.0f;
#pragma unroll
                for (int i = 0; i < 6; ++i) v += G[i][r] * M[i * 6 + jj];
                R[r][jj] = v;
            }
        float* o = dw + j * 9;                                       // j = co * cin + ci
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float v = 0.0f;
#pragma unroll
                for (int jj = 0; jj < 6; ++jj) v += R[r][jj] * G[jj][c];
                o[r * 3 + c] = v;
            }
    }
}

// Both steps in one launch, for the planes wide enough to fill the chip that way (cout cin / 32 >= 512 workgroups): a workgroup owns one cout x 32 cins, thread
// (position triple, cin) adds the S partials of its three positions (eight loads in flight, every load a 128-byte row) into LDS, then thread
// (tap, cin) forms its tap of G^T M G.  The last workgroups add the bias partials [s][co].  (r5 first form: a sum launch + a transform launch,
// 14 us of a 128 -> 128 layer's 89.  Same additions in the same order: the same bits as the two launches.)
__global__ __launch_bounds__(384) void wgrad_wino_finish_kernel(const float* __restrict__ ws, const float* __restrict__ wsb, float* __restrict__ dw,
                                                                float* __restrict__ db, int S, int cin, int cout, int coP, int ciP) {
    constexpr float G[6][3] = {{0.25f, 0.f, 0.f}, {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                               {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
    __shared__ float M[36][32];
    const int tid = threadIdx.x, n_cb = ciP / 32, n_main = cout * n_cb;
    auto sum_splits = [&](const float* p, size_t stride) {
        float sum = p[0];
        int s_ = 1;
        for (; s_ + 8 <= S; s_ += 8) {                                   // eight loads in flight, added in split order
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = p[(size_t)(s_ + k) * stride];
#pragma unroll
            for (int k = 0; k < 8; ++k) sum += v[k];
        }
        for (; s_ < S; ++s_) sum += p[(size_t)s_ * stride];
        return sum;
    };
    if ((int)blockIdx.x >= n_main) {                                 // bias rows
        const int co = ((int)blockIdx.x - n_main) * 384 + tid;
        if (db && co < cout) db[co] = sum_splits(wsb + co, (size_t)coP);
        return;
    }
    const int co = blockIdx.x / n_cb, ci0 = (blockIdx.x % n_cb) * 32, cl = tid & 31, pg = tid >> 5;
    const size_t plane = (size_t)coP * ciP;                          // (padded to whole blocks)
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int pos = 3 * pg + q;
        M[pos][cl] = sum_splits(ws + (size_t)pos * plane + (size_t)co * ciP + ci0 + cl, 36 * plane);
    }
    __syncthreads();
    if (tid < 288) {
        const int r = pg / 3, c = pg % 3;
        float v = 0.0f;
#pragma unroll
        for (int jj = 0; jj < 6; ++jj) {
            float rj = 0.0f;                                             // (G^T M)[r][jj]
#pragma unroll
            for (int i = 0; i < 6; ++i) rj += G[i][r] * M[i * 6 + jj][cl];
            v += rj * G[jj][c];
        }
        if (ci0 + cl < cin) dw[((size_t)co * cin + ci0 + cl) * 9 + pg] = v;
    }
}

// Which form runs: 0 = by the shape (the product), 1 = nine taps everywhere, 2 = Winograd domain on four waves wherever it applies, 3 = on eight waves
// wherever cout % 64 == 0 (else four).  Set by nd_conv3x3_wgrad_form (tests, tools); the environment gives the start value: ND_WGRAD_WINO=0 -> 1,
// ND_WGRAD_WINO8=0 -> 2.  The result of a form never depends on anything but the shape.
int& ww_form() {
    static int form = (getenv("ND_WGRAD_WINO") && atoi(getenv("ND_WGRAD_WINO")) == 0) ? 1 : (getenv("ND_WGRAD_WINO8") && atoi(getenv("ND_WGRAD_WINO8")) == 0) ? 2 : 0;
    return form;
}

bool ww_takes(int B, int H, int W, int cin, int cout) {
    return ww_form() != 1 && H % WW_GH == 0 && W % WW_GW == 0 && cin % 16 == 0 && cout % 16 == 0 && (long)B * H * W < (1L << 30);      // (blocks of 32 channels, the last one may be half empty)
}

void ww_plan(int B, int H, int W, int cin, int cout, WwArgs& a) {
    a.B = B; a.H = H; a.W = W; a.cin = cin; a.cout = cout;
    // the eight-wave form ((64 couts x 32 cins) blocks): 20 % fewer cycles per tile group and block, but twice the accumulators per CU -- twice the
    // partial sums to write and add.  It pays where the groups x blocks product is large (measured over the d = 64 layers: profiles/r5_wgrad_wino.txt)
    a.wide = ww_form() != 2 && cout % W8_CO == 0 && cin % WW_CB == 0 && (ww_form() == 3 || (long)B * (H / WW_GH) * (W / WW_GW) * (cin / WW_CB) * (cout / WW_CB) >= 20000);
    a.n_co = a.wide ? cout / W8_CO : nd_cdiv(cout, WW_CB);  a.n_ci = nd_cdiv(cin, WW_CB);
    a.coP = a.n_co * (a.wide ? W8_CO : WW_CB);  a.ciP = a.n_ci * WW_CB;      // the workspace planes: whole blocks
    a.gx = W / WW_GW;  a.gy = H / WW_GH;
    a.n_groups = B * a.gx * a.gy;
    // The split over the tile groups: fixed by the shape (the summation order never depends on the device).  Workgroups run in rounds of
    // WW_TARGET_WGS, one per CU; a round costs its groups + ~14 groups' worth of pipeline fill, partial-sum stores and their summation: the cheapest split wins
    // (768 -> 512 at 32 x 32: 384 blocks, two rounds unsplit, three rounds of half the length split in two).
    const int blocks = a.n_co * a.n_ci;
    long best = -1;
    int S = 1;
    for (int c = 1; c <= a.n_groups && c <= 4 * WW_TARGET_WGS; ++c) {
        const long rounds = ((long)blocks * c + WW_TARGET_WGS - 1) / WW_TARGET_WGS, cost = rounds * ((a.n_groups + c - 1) / c + 14);
        if (best < 0 || cost < best) { best = cost;  S = c; }
        if (rounds > 8) break;
    }
    a.S = S;
}

}  // namespace

extern "C" int nd_conv3x3_wgrad_form(int form) {
    const int was = ww_form();
    if (form >= 0 && form <= 3) ww_form() = form;
    return was;
}

extern "C" int64_t nd_conv3x3_wgrad_workspace_floats(int B, int H, int W, int cin, int cout) {
    if (B <= 0 || H <= 0 || W <= 0 || cin <= 0 || cout <= 0) return -1;
    WgradArgs a;
    plan(B, H, W, cin, cout, a);
    int64_t n = (int64_t)a.S * a.n_co * CB * (9 * a.n_ci * CB + 1);      // weight partials, then the bias partials
    if (ww_takes(B, H, W, cin, cout)) {                                  // the Winograd-domain form: [S][36][co][ci] + [S][co]
        WwArgs w;
        ww_plan(B, H, W, cin, cout, w);
        const int64_t m = (int64_t)w.S * w.coP * (36 * (int64_t)w.ciP + 1);
        if (m > n) n = m;
    }
    return n;
}

static int wgrad_run(const float* x, int ldx, int c0, const float* x1, int ldx1, const float* dy, int ldy, float* dw_oihw, float* dbias, float* workspace,
                     int B, int H, int W, int cin, int cout, void* stream) {
    ND_REQUIRE(x && dy && dw_oihw && workspace, ND_E_BADARG, "nd_conv3x3_wgrad: null pointer");
    ND_REQUIRE(B > 0 && H > 0 && W > 0 && cin > 0 && cout > 0, ND_E_BADARG, "nd_conv3x3_wgrad: non-positive size");
    ND_REQUIRE(cin % 4 == 0 && cout % 4 == 0 && ldx >= c0 && ldy >= cout && ldx % 4 == 0 && ldy % 4 == 0, ND_E_SHAPE,
               "nd_conv3x3_wgrad: cin=%d, cout=%d and the pixel strides must be multiples of 4", cin, cout);
    ND_REQUIRE(nd_aligned16(x) && nd_aligned16(dy), ND_E_ALIGN, "nd_conv3x3_wgrad: x and dy must be 16-byte aligned");
    if (ww_takes(B, H, W, cin, cout) && (long)H * W * ldx * 4 < (1L << 30) && (long)H * W * ldy * 4 < (1L << 30)) {     // F(4x4) Winograd-domain form (a quarter of the MFMAs); 32-bit offsets within a sample
        WwArgs w;
        ww_plan(B, H, W, cin, cout, w);
        w.x = x; w.dy = dy; w.ws = workspace; w.ldx = ldx; w.ldy = ldy;
        w.x1 = x1; w.ldx1 = ldx1; w.c0 = c0;
        w.wsb = dbias ? workspace + (size_t)w.S * 36 * w.coP * w.ciP : nullptr;
        static nd_device_once configured_w;
        if (int e = nd_reserve_lds(configured_w, reinterpret_cast<const void*>(wgrad_wino_kernel), WW_LDS, "nd_conv3x3_wgrad (Winograd domain)")) return e;
        hipStream_t st = (hipStream_t)stream;
        if (w.wide) {
            static nd_device_once configured_w8;
            if (int e = nd_reserve_lds(configured_w8, reinterpret_cast<const void*>(wgrad_wino8_kernel), W8_LDS, "nd_conv3x3_wgrad (Winograd domain, eight waves)")) return e;
            hipLaunchKernelGGL(wgrad_wino8_kernel, dim3((unsigned)(w.n_co * w.n_ci * w.S)), dim3(512), W8_LDS, st, w);
        } else
        hipLaunchKernelGGL(wgrad_wino_kernel, dim3((unsigned)(w.n_co * w.n_ci * w.S)), dim3(256), WW_LDS, st, w);
        if (int e = nd_launch_status("nd_conv3x3_wgrad_nhwc_f32 (Winograd domain)")) return e;
        const int n_main = cout * (w.ciP / 32), n_bias = dbias ? (cout + 383) / 384 : 0;
        if (w.S > 1 && n_main >= 512) {
            hipLaunchKernelGGL(wgrad_wino_finish_kernel, dim3((unsigned)(n_main + n_bias)), dim3(384), 0, st, workspace, w.wsb, dw_oihw, dbias, w.S, cin, cout, w.coP, w.ciP);
            return nd_launch_status("nd_conv3x3_wgrad_nhwc_f32 (Winograd-domain sum + reduce)");
        }
        const size_t n_w = (size_t)36 * w.coP * w.ciP;
        if (w.S > 1) {
            const size_t tot = n_w + (dbias ? w.coP : 0);
            hipLaunchKernelGGL(wgrad_wino_sum_kernel, dim3((unsigned)((tot + 255) / 256 < 8192 ? (tot + 255) / 256 : 8192)), dim3(256), 0, st, workspace, w.wsb, w.S, n_w, w.coP);
        }
        const size_t total = (size_t)cout * (cin + 1);
        hipLaunchKernelGGL(wgrad_wino_reduce_kernel, dim3((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), dim3(256), 0, st, workspace, w.wsb, dw_oihw, dbias, cin, cout, w.coP, w.ciP);
        return nd_launch_status("nd_conv3x3_wgrad_nhwc_f32 (Winograd-domain reduce)");
    }
    ND_REQUIRE(c0 == cin, ND_E_SHAPE, "nd_conv3x3_wgrad_cat_nhwc_f32: two sources need the Winograd-domain form (H %% 4 == 0, W %% 16 == 0, channel counts %% 32 == 0)");
    WgradArgs a;
    plan(B, H, W, cin, cout, a);
    a.x = x; a.dy = dy; a.ws = workspace; a.ldx = ldx; a.ldy = ldy;
    a.wsb = dbias ? workspace + (size_t)a.S * 9 * a.n_co * CB * a.n_ci * CB : nullptr;
    {   // the interior-tile path addresses both tensors with 32-bit byte offsets
        const long xb = (long)B * H * W * ldx * 4, dyb = (long)B * H * W * ldy * 4;
        a.fast = xb < (1L << 31) && dyb < (1L << 31) && nd_aligned16(x) && nd_aligned16(dy);
        a.x_bytes = a.fast ? (unsigned)xb : 0u;
        a.dy_bytes = a.fast ? (unsigned)dyb : 0u;
    }
    const long wgs = (long)a.n_co * a.n_ci * a.S;
    ND_REQUIRE(wgs < (1L << 31), ND_E_SHAPE, "nd_conv3x3_wgrad: grid too large");
    const size_t lds = (size_t)(WG_TILE * TILE_H + HALO * HALO_H) * CB * sizeof(float);
    static nd_device_once configured;
    if (int e = nd_reserve_lds(configured, reinterpret_cast<const void*>(wgrad_kernel), lds, "nd_conv3x3_wgrad")) return e;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(wgrad_kernel, dim3((unsigned)wgs), dim3(256), lds, st, a);
    if (int e = nd_launch_status("nd_conv3x3_wgrad_nhwc_f32")) return e;
    const size_t total = (size_t)a.n_co * CB * (9 * a.n_ci * CB + 1);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, workspace, a.wsb, dw_oihw, dbias, a.S, cin, cout, a.n_co * CB, a.n_ci * CB);
    return nd_launch_status("nd_conv3x3_wgrad_nhwc_f32 (reduce)");
}

extern "C" int nd_conv3x3_wgrad_nhwc_f32(const float* x, int ldx, const float* dy, int ldy, float* dw_oihw, float* dbias, float* workspace,
                                         int B, int H, int W, int cin, int cout, void* stream) {
    return wgrad_run(x, ldx, cin, nullptr, 0, dy, ldy, dw_oihw, dbias, workspace, B, H, W, cin, cout, stream);
}

extern "C" int nd_conv3x3_wgrad_cat_nhwc_f32(const float* x0, int ldx0, int c0, const float* x1, int ldx1, int c1, const float* dy, int ldy, float* dw_oihw,
                                             float* dbias, float* workspace, int B, int H, int W, int cout, void* stream) {
    ND_REQUIRE(x0 && x1 && c0 > 0 && c1 > 0 && c0 % 32 == 0 && c1 % 32 == 0 && ldx0 >= c0 && ldx1 >= c1 && ldx1 % 4 == 0 && nd_aligned16(x1), ND_E_SHAPE,
               "nd_conv3x3_wgrad_cat_nhwc_f32: two sources of whole 32-channel blocks (c0=%d, c1=%d)", c0, c1);
    ND_REQUIRE((long)H * W * ldx1 * 4 < (1L << 30), ND_E_SHAPE, "nd_conv3x3_wgrad_cat_nhwc_f32: second source too large for 32-bit offsets within a sample");
    return wgrad_run(x0, ldx0, c0, x1, ldx1, dy, ldy, dw_oihw, dbias, workspace, B, H, W, c0 + c1, cout, stream);
}
