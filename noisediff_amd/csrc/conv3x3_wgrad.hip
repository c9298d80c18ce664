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
// A workgroup owns a (32 couts x 32 cins) block of all 36 positions -- wave w the positions 9 w .. 9 w + 8, nine 32 x 32 accumulators -- and walks
// its share of the tile groups (2 x 4 tiles = 8 x 16 pixels).  Per group: the 10 x 18 x 32 input halo and the 8 x 16 x 32 dY block go to LDS
// (requested one group ahead, in registers), thread (tile, channel) transforms one patch of each into V[pos][tile][ch] / D[pos][tile][ch]
// (the MFMA operands: lane (ch, tile parity) reads one dword each, conflict-free), then 36 MFMAs per wave (v_mfma_f32_32x32x2_f32, K = two
// tiles).  The split over tile groups is fixed by the shape alone; partials [split][pos][co][ci] and the bias partials go to the workspace and
// wgrad_wino_reduce_kernel adds them in split order and applies G^T . G.  Bitwise repeatable, no atomics.
// Taken when H % 8 == 0, W % 16 == 0 and both channel counts are multiples of 32 (every Block.proj of the d = 64 network at the training sizes);
// anything else keeps the nine-tap kernel above.  ND_WGRAD_WINO=0: A/B knob.
constexpr int WW_CB = 32;                                            // channels per block, both sides
constexpr int WW_TY = 2, WW_TX = 4, WW_NT = WW_TY * WW_TX;           // tile group: 2 x 4 tiles of 4 x 4 pixels
constexpr int WW_GH = 4 * WW_TY, WW_GW = 4 * WW_TX;                  // 8 x 16 pixels
constexpr int WW_HR = WW_GH + 2, WW_HC = WW_GW + 2;                  // halo 10 x 18
constexpr int WW_TARGET_WGS = 256;                                   // fixed: the summation order must not depend on the device
constexpr int WW_X_IT = (WW_HR * WW_HC * (WW_CB / 4) + 255) / 256;   // 6 float4 per thread
constexpr int WW_Y_IT = WW_GH * WW_GW * (WW_CB / 4) / 256;           // 4 float4 per thread
constexpr int WW_LDT = 12;                                           // floats per (position, channel) row of the operand images: 8 tiles + pad (16-byte reads, conflict-free)
constexpr size_t WW_LDS = (size_t)(WW_HR * WW_HC + WW_GH * WW_GW + 2 * 36 * WW_LDT) * WW_CB * sizeof(float);   // 150016 bytes

struct WwArgs {
    const float* x; const float* dy; float* ws; float* wsb;
    int ldx, ldy, B, H, W, cin, cout;
    int n_co, n_ci, S, gx, gy, n_groups;
};

typedef float ww_f2 __attribute__((ext_vector_type(2)));
template <typename T>
__device__ __forceinline__ void ww_bt6(const T (&d)[6], T (&t)[6]) {      // one row of B^T applied to six values (float2: two channels at once, packed instructions)
    t[0] = 4.0f * d[0] - 5.0f * d[2] + d[4];
    t[1] = -4.0f * (d[1] + d[2]) + d[3] + d[4];
    t[2] = 4.0f * (d[1] - d[2]) - d[3] + d[4];
    t[3] = 2.0f * (d[3] - d[1]) + d[4] - d[2];
    t[4] = 2.0f * (d[1] - d[3]) + d[4] - d[2];
    t[5] = 4.0f * d[1] - 5.0f * d[3] + d[5];
}
template <typename T>
__device__ __forceinline__ void ww_a4(const T (&y)[4], T (&o)[6]) {       // A (6 x 4) applied to four values
    const T e = y[0] + y[2], f = y[1] + y[3], g4 = y[0] + 4.0f * y[2], h = 2.0f * y[1] + 8.0f * y[3];
    o[0] = y[0];
    o[1] = e + f;
    o[2] = e - f;
    o[3] = g4 + h;
    o[4] = g4 - h;
    o[5] = y[3];
}

__global__ __launch_bounds__(256, 1) void wgrad_wino_kernel(const WwArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Xs = sm;                                                  // [10][18][32]
    float* Ys = Xs + WW_HR * WW_HC * WW_CB;                          // [8][16][32]
    float* Vs = Ys + WW_GH * WW_GW * WW_CB;                          // [36][32 cins][8 tiles + 4]
    float* Ds = Vs + 36 * WW_CB * WW_LDT;                            // [36][32 couts][8 tiles + 4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, half = lane >> 5;
    int bid = blockIdx.x;
    const int s = bid % a.S;  bid /= a.S;
    const int cib = bid % a.n_ci, cob = bid / a.n_ci;
    const int co0 = cob * WW_CB, ci0 = cib * WW_CB;
    const int g_lo = (int)((long)s * a.n_groups / a.S), g_hi = (int)((long)(s + 1) * a.n_groups / a.S);

    f32x16 acc[9];
#pragma unroll
    for (int p = 0; p < 9; ++p) acc[p] = nd_zero16();
    const bool do_bias = a.wsb != nullptr && cib == 0;
    ww_f2 bsum2 = {0.0f, 0.0f};                                      // D-transform threads: their (tile, cout pair) share of the bias gradient
    // transform roles: waves 0, 1 the input patches (V), waves 2, 3 the dY blocks (D); a thread = (tile, channel PAIR), packed float2 arithmetic
    const int t_tile = (tid & 127) >> 4, t_cp = tid & 15, t_ch = 2 * t_cp;
    const int t_ty = t_tile / WW_TX, t_tx = t_tile % WW_TX;
    const bool t_isV = tid < 128;

    f32x4 xr[WW_X_IT], yr[WW_Y_IT];
    auto issue = [&](int g) {                                        // the group's halo and dY block -> registers (zero outside the image)
        const int gxy = a.gx * a.gy;
        const int b = g / gxy, r_ = g - b * gxy, gyi = r_ / a.gx, gxi = r_ - gyi * a.gx;
        const int y0 = gyi * WW_GH, x0 = gxi * WW_GW;
        const f32x4 zero = {0, 0, 0, 0};
#pragma unroll
        for (int it = 0; it < WW_X_IT; ++it) {
            const int idx = tid + 256 * it, q = idx & 7, px = idx >> 3, r = px / WW_HC, c = px - r * WW_HC;
            const int y = y0 - 1 + r, x = x0 - 1 + c;
            const bool ok = px < WW_HR * WW_HC && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
            xr[it] = ok ? nd_ld4(a.x + ((size_t)(b * a.H + y) * a.W + x) * a.ldx + ci0 + 4 * q) : zero;
        }
#pragma unroll
        for (int it = 0; it < WW_Y_IT; ++it) {
            const int idx = tid + 256 * it, q = idx & 7, px = idx >> 3, r = px / WW_GW, c = px - r * WW_GW;
            yr[it] = nd_ld4(a.dy + ((size_t)(b * a.H + y0 + r) * a.W + x0 + c) * a.ldy + co0 + 4 * q);
        }
    };
    if (g_lo < g_hi) issue(g_lo);
    for (int g = g_lo; g < g_hi; ++g) {
#pragma unroll
        for (int it = 0; it < WW_X_IT; ++it) {
            const int idx = tid + 256 * it;
            if (idx < WW_HR * WW_HC * 8) nd_st4(Xs + (idx >> 3) * WW_CB + 4 * (idx & 7), xr[it]);
        }
#pragma unroll
        for (int it = 0; it < WW_Y_IT; ++it) {
            const int idx = tid + 256 * it;
            nd_st4(Ys + (idx >> 3) * WW_CB + 4 * (idx & 7), yr[it]);
        }
        __syncthreads();                                             // the group's raw data is in LDS (and every wave is done with the previous group's MFMAs)
        if (g + 1 < g_hi) issue(g + 1);                              // the next group's loads fly over this group's transforms and MFMAs
        if (t_isV) {    // V = B^T d B of this thread's (tile, cin pair): 36 8-byte reads, two packed passes, 72 writes
            ww_f2 T[6][6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                ww_f2 d[6], t[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) d[i] = *reinterpret_cast<const ww_f2*>(Xs + ((4 * t_ty + i) * WW_HC + 4 * t_tx + j) * WW_CB + t_ch);
                ww_bt6(d, t);
#pragma unroll
                for (int i = 0; i < 6; ++i) T[i][j] = t[i];
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                ww_f2 v[6];
                ww_bt6(T[i], v);
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    float* o = Vs + ((i * 6 + j) * WW_CB + t_ch) * WW_LDT + t_tile;
                    o[0] = v[j].x;  o[WW_LDT] = v[j].y;
                }
            }
        } else {        // D = A dY A^T of this thread's (tile, cout pair): 16 reads, 72 writes; the bias gradient's share on the way
            ww_f2 U[6][4];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                ww_f2 y[4], o[6];
#pragma unroll
                for (int m = 0; m < 4; ++m) y[m] = *reinterpret_cast<const ww_f2*>(Ys + ((4 * t_ty + m) * WW_GW + 4 * t_tx + n) * WW_CB + t_ch);
                if (do_bias) bsum2 += (y[0] + y[1]) + (y[2] + y[3]);
                ww_a4(y, o);
#pragma unroll
                for (int i = 0; i < 6; ++i) U[i][n] = o[i];
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                ww_f2 o6[6];
                ww_a4(U[i], o6);
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    float* o = Ds + ((i * 6 + j) * WW_CB + t_ch) * WW_LDT + t_tile;
                    o[0] = o6[j].x;  o[WW_LDT] = o6[j].y;
                }
            }
        }
        __syncthreads();                                             // V and D of the group are complete
        // all operands of the wave's nine positions first (one LDS round trip for the lot: left next to their MFMAs, hipcc waits for every read in
        // front of its MFMA -- 36 exposed LDS latencies per group), then 36 MFMAs back to back.  K step k of an instruction = tiles {k, 4 + k}: lane
        // (channel, half) holds tiles 4 half .. 4 half + 3 of its channel -- 16 contiguous bytes per operand and position.
        f32x4 av[9], bv[9];
#pragma unroll
        for (int p = 0; p < 9; ++p) {
            const int o = ((wave * 9 + p) * WW_CB + col) * WW_LDT + 4 * half;
            av[p] = *reinterpret_cast<const f32x4*>(Ds + o);
            bv[p] = *reinterpret_cast<const f32x4*>(Vs + o);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < 9; ++p)
#pragma unroll
            for (int k = 0; k < WW_NT / 2; ++k) {
                const float a_ = av[p][k], b_ = bv[p][k];                // (element k of the float4s)
                // accumulators pinned to the AGPR half by the constraint (the file is built with hipcc's VGPR-form switch for the nine-tap kernel: left to the
                // builtin, 144 accumulator registers + the transforms' values overflow the VGPR half and hipcc shuttles them through v_accvgpr_mov / read / write)
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[p]) : "v"(a_), "v"(b_));
            }
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- this workgroup's partial sums: ws[s][pos][co][ci]  (the MFMAs are asm statements: hipcc does not know their results are still in flight)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int p = 0; p < 9; ++p) asm volatile("" : "+a"(acc[p]));
    const int coP = a.n_co * WW_CB, ciP = a.n_ci * WW_CB;
#pragma unroll
    for (int p = 0; p < 9; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + nd_acc_row(r, lane), ci = ci0 + col;
            a.ws[(((size_t)s * 36 + wave * 9 + p) * coP + co) * ciP + ci] = acc[p][r];
        }
    if (do_bias) {                                                   // the eight tile threads of a cout pair meet in a fixed order
        __syncthreads();
        if (!t_isV) { Xs[(t_tile * 16 + t_cp) * 2] = bsum2.x;  Xs[(t_tile * 16 + t_cp) * 2 + 1] = bsum2.y; }
        __syncthreads();
        if (tid < WW_CB) {
            float v = Xs[tid];
#pragma unroll
            for (int t = 1; t < WW_NT; ++t) v += Xs[t * WW_CB + tid];
            a.wsb[(size_t)s * coP + co0 + tid] = v;
        }
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
                                                                float* __restrict__ db, int cin, int cout) {
    constexpr float G[6][3] = {{0.25f, 0.f, 0.f}, {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                               {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
    const size_t plane = (size_t)cout * cin, total = plane + (db ? cout : 0);
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < total; j += (size_t)gridDim.x * blockDim.x) {
        if (j >= plane) { db[j - plane] = wsb[j - plane];  continue; }
        float M[36];
#pragma unroll
        for (int pos = 0; pos < 36; ++pos) M[pos] = ws[(size_t)pos * plane + j];
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

bool ww_takes(int B, int H, int W, int cin, int cout) {
    static const bool on = !(getenv("ND_WGRAD_WINO") && atoi(getenv("ND_WGRAD_WINO")) == 0);       // A/B knob
    return on && H % WW_GH == 0 && W % WW_GW == 0 && cin % WW_CB == 0 && cout % WW_CB == 0 && (long)B * H * W < (1L << 30);
}

void ww_plan(int B, int H, int W, int cin, int cout, WwArgs& a) {
    a.B = B; a.H = H; a.W = W; a.cin = cin; a.cout = cout;
    a.n_co = cout / WW_CB;  a.n_ci = cin / WW_CB;
    a.gx = W / WW_GW;  a.gy = H / WW_GH;
    a.n_groups = B * a.gx * a.gy;
    int S = WW_TARGET_WGS / (a.n_co * a.n_ci);                       // fixed by the shape: the summation order never depends on the device
    if (S < 1) S = 1;
    if (S > a.n_groups) S = a.n_groups;
    a.S = S;
}

}  // namespace

extern "C" int64_t nd_conv3x3_wgrad_workspace_floats(int B, int H, int W, int cin, int cout) {
    if (B <= 0 || H <= 0 || W <= 0 || cin <= 0 || cout <= 0) return -1;
    WgradArgs a;
    plan(B, H, W, cin, cout, a);
    int64_t n = (int64_t)a.S * a.n_co * CB * (9 * a.n_ci * CB + 1);      // weight partials, then the bias partials
    if (ww_takes(B, H, W, cin, cout)) {                                  // the Winograd-domain form: [S][36][co][ci] + [S][co]
        WwArgs w;
        ww_plan(B, H, W, cin, cout, w);
        const int64_t m = (int64_t)w.S * cout * (36 * (int64_t)cin + 1);
        if (m > n) n = m;
    }
    return n;
}

extern "C" int nd_conv3x3_wgrad_nhwc_f32(const float* x, int ldx, const float* dy, int ldy, float* dw_oihw, float* dbias, float* workspace,
                                         int B, int H, int W, int cin, int cout, void* stream) {
    ND_REQUIRE(x && dy && dw_oihw && workspace, ND_E_BADARG, "nd_conv3x3_wgrad: null pointer");
    ND_REQUIRE(B > 0 && H > 0 && W > 0 && cin > 0 && cout > 0, ND_E_BADARG, "nd_conv3x3_wgrad: non-positive size");
    ND_REQUIRE(cin % 4 == 0 && cout % 4 == 0 && ldx >= cin && ldy >= cout && ldx % 4 == 0 && ldy % 4 == 0, ND_E_SHAPE,
               "nd_conv3x3_wgrad: cin=%d, cout=%d and the pixel strides must be multiples of 4", cin, cout);
    ND_REQUIRE(nd_aligned16(x) && nd_aligned16(dy), ND_E_ALIGN, "nd_conv3x3_wgrad: x and dy must be 16-byte aligned");
    if (ww_takes(B, H, W, cin, cout)) {                                  // F(4x4) Winograd-domain form (a quarter of the MFMAs)
        WwArgs w;
        ww_plan(B, H, W, cin, cout, w);
        w.x = x; w.dy = dy; w.ws = workspace; w.ldx = ldx; w.ldy = ldy;
        w.wsb = dbias ? workspace + (size_t)w.S * 36 * cout * cin : nullptr;
        static nd_device_once configured_w;
        if (int e = nd_reserve_lds(configured_w, reinterpret_cast<const void*>(wgrad_wino_kernel), WW_LDS, "nd_conv3x3_wgrad (Winograd domain)")) return e;
        hipStream_t st = (hipStream_t)stream;
        hipLaunchKernelGGL(wgrad_wino_kernel, dim3((unsigned)(w.n_co * w.n_ci * w.S)), dim3(256), WW_LDS, st, w);
        if (int e = nd_launch_status("nd_conv3x3_wgrad_nhwc_f32 (Winograd domain)")) return e;
        const size_t n_w = (size_t)36 * cout * cin;
        if (w.S > 1) {
            const size_t tot = n_w + (dbias ? cout : 0);
            hipLaunchKernelGGL(wgrad_wino_sum_kernel, dim3((unsigned)((tot + 255) / 256 < 8192 ? (tot + 255) / 256 : 8192)), dim3(256), 0, st, workspace, w.wsb, w.S, n_w, cout);
        }
        const size_t total = (size_t)cout * (cin + 1);
        hipLaunchKernelGGL(wgrad_wino_reduce_kernel, dim3((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), dim3(256), 0, st, workspace, w.wsb, dw_oihw, dbias, cin, cout);
        return nd_launch_status("nd_conv3x3_wgrad_nhwc_f32 (Winograd-domain reduce)");
    }
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
