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

}  // namespace

extern "C" int64_t nd_conv3x3_wgrad_workspace_floats(int B, int H, int W, int cin, int cout) {
    if (B <= 0 || H <= 0 || W <= 0 || cin <= 0 || cout <= 0) return -1;
    WgradArgs a;
    plan(B, H, W, cin, cout, a);
    return (int64_t)a.S * a.n_co * CB * (9 * a.n_ci * CB + 1);           // weight partials, then the bias partials
}

extern "C" int nd_conv3x3_wgrad_nhwc_f32(const float* x, int ldx, const float* dy, int ldy, float* dw_oihw, float* dbias, float* workspace,
                                         int B, int H, int W, int cin, int cout, void* stream) {
    ND_REQUIRE(x && dy && dw_oihw && workspace, ND_E_BADARG, "nd_conv3x3_wgrad: null pointer");
    ND_REQUIRE(B > 0 && H > 0 && W > 0 && cin > 0 && cout > 0, ND_E_BADARG, "nd_conv3x3_wgrad: non-positive size");
    ND_REQUIRE(cin % 4 == 0 && cout % 4 == 0 && ldx >= cin && ldy >= cout && ldx % 4 == 0 && ldy % 4 == 0, ND_E_SHAPE,
               "nd_conv3x3_wgrad: cin=%d, cout=%d and the pixel strides must be multiples of 4", cin, cout);
    ND_REQUIRE(nd_aligned16(x) && nd_aligned16(dy), ND_E_ALIGN, "nd_conv3x3_wgrad: x and dy must be 16-byte aligned");
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
