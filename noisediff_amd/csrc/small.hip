// small.hip -- the conditioning path (tiny dense ops on (B, K) rows), the two special
// full-resolution layers with <= 4 input channels, and NCHW<->NHWC plumbing for the API tensors.
#include "nd_common.h"

namespace {

// out[b, n] = act_out(sum_k act_in(in[b, k]) * W[n, k] + bias[n]); one wave per output column n,
// the weight row lives in registers and is reused for every b.  K <= 2048.
// time_mlp (models/archs/Diffusion_arch.py:502-507), all ResnetBlock.mlp projections stacked into
// one tall matrix (:149-152,:161-164), CrossAttention.to_v/to_out on the ISO token (:385,:402).
__global__ __launch_bounds__(256) void linear_rows_kernel(const float* __restrict__ in, int ld_in, const float* __restrict__ W,
                                                          const float* __restrict__ bias, float* __restrict__ out, int ld_out,
                                                          int B, int K, int N, int act_in, int act_out) {
    const int lane = threadIdx.x & 63;
    const int n = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (n >= N) return;
    float w[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const int k = lane + j * 64;
        w[j] = k < K ? W[(size_t)n * K + k] : 0.0f;
    }
    const float bv = bias ? bias[n] : 0.0f;
    for (int b = 0; b < B; ++b) {
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int k = lane + j * 64;
            if (k < K) s += nd_act(in[(size_t)b * ld_in + k], act_in) * w[j];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) out[(size_t)b * ld_out + n] = nd_act(s + bv, act_out);
    }
}

// SinusoidalPosEmb.forward (:100-107)
__global__ void sinusoidal_kernel(const int64_t* __restrict__ time, const float* __restrict__ freqs, float* __restrict__ emb, int B, int half) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * half) return;
    const int b = i / half, j = i - b * half;
    const float ang = (float)time[b] * freqs[j];
    emb[(size_t)b * 2 * half + j] = sinf(ang);
    emb[(size_t)b * 2 * half + half + j] = cosf(ang);
}

__global__ void embedding_kernel(const int64_t* __restrict__ idx, const float* __restrict__ table, float* __restrict__ out, int B, int rows, int dim) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * dim) return;
    const int b = i / dim, j = i - b * dim;
    int64_t r = idx[b];
    r = r < 0 ? 0 : (r >= rows ? rows - 1 : r);   // host validates; clamp keeps a bad index from faulting
    out[i] = table[r * dim + j];
}

// LearnedSinusoidalPosEmb.forward (:331-337): w = conv1x1(position); out = cat(w, sin(2 pi w), cos(2 pi w))
__global__ __launch_bounds__(256) void pos_enc_kernel(const float* __restrict__ pos, const float* __restrict__ w, const float* __restrict__ bias,
                                                      float* __restrict__ out, int B, int HW, int hid) {
    const size_t total = (size_t)B * HW * hid;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int h = (int)(i % hid);
        const size_t pix = i / hid;
        const int b = (int)(pix / HW);
        const size_t p = pix - (size_t)b * HW;
        const float p0 = pos[((size_t)b * 2 + 0) * HW + p], p1 = pos[((size_t)b * 2 + 1) * HW + p];
        const float v = fmaf(w[h * 2 + 1], p1, fmaf(w[h * 2], p0, bias[h]));
        const float f = v * 2.0f * 3.14159265358979323846f;
        float* o = out + pix * (3 * hid);
        o[h] = v;
        o[hid + h] = sinf(f);
        o[2 * hid + h] = cosf(f);
    }
}

__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int C, int HW) {
    const size_t total = (size_t)B * HW * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t pix = i / C;
        const int b = (int)(pix / HW);
        const size_t p = pix - (size_t)b * HW;
        out[i] = in[((size_t)b * C + c) * HW + p];
    }
}

// NCHW -> NHWC with the channel tail zero-filled up to Cpad
__global__ __launch_bounds__(256) void nchw_to_nhwc_pad_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int C, int HW, int Cpad) {
    const size_t total = (size_t)B * HW * Cpad;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cpad);
        const size_t pix = i / Cpad;
        const int b = (int)(pix / HW);
        const size_t p = pix - (size_t)b * HW;
        out[i] = c < C ? in[((size_t)b * C + c) * HW + p] : 0.0f;
    }
}

// nn.MaxPool2d(2, 2, ceil_mode=True): windows hanging over the border use the in-bounds elements only
__global__ __launch_bounds__(256) void maxpool2x2_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W, int C) {
    const int Ho = (H + 1) >> 1, Wo = (W + 1) >> 1, cq = C >> 2;
    const size_t total = (size_t)B * Ho * Wo * cq;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cq) * 4;
        size_t r = i / cq;
        const int x = (int)(r % Wo); r /= Wo;
        const int y = (int)(r % Ho);
        const int b = (int)(r / Ho);
        const int y1 = min(2 * y + 1, H - 1), x1 = min(2 * x + 1, W - 1);
        const float* base = in + (size_t)b * H * W * C + c;
        const f32x4 v00 = nd_ld4(base + ((size_t)(2 * y) * W + 2 * x) * C), v01 = nd_ld4(base + ((size_t)(2 * y) * W + x1) * C);
        const f32x4 v10 = nd_ld4(base + ((size_t)y1 * W + 2 * x) * C), v11 = nd_ld4(base + ((size_t)y1 * W + x1) * C);
        f32x4 m;
        m.x = fmaxf(fmaxf(v00.x, v01.x), fmaxf(v10.x, v11.x)); m.y = fmaxf(fmaxf(v00.y, v01.y), fmaxf(v10.y, v11.y));
        m.z = fmaxf(fmaxf(v00.z, v01.z), fmaxf(v10.z, v11.z)); m.w = fmaxf(fmaxf(v00.w, v01.w), fmaxf(v10.w, v11.w));
        nd_st4(out + i * 4, m);
    }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int C, int HW) {
    const size_t total = (size_t)B * HW * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i % HW;
        const size_t bc = i / HW;
        const int c = (int)(bc % C);
        const int b = (int)(bc / C);
        out[i] = in[((size_t)b * HW + p) * C + c];
    }
}

// init_conv: Conv2d(4, cout, 7, padding=3) (:478) as an implicit GEMM on the fp32 MFMA, K = 49 taps x 4 channels.
// Transposed product (rows = output channels, columns = 32 pixels of one image row) so that a lane ends up with 16
// channels of ONE pixel and stores 16-byte pieces.  A wave owns one 32-channel n-tile: its 98 weight operands
// (lane (n, half): W[n][tap][c], instruction 2*tap + jj pairs channels (jj, jj + 2)) stay in registers for the whole
// persistent kernel; the pixel operand of a tap is one ds_read_b64 from the zero-padded 22 x 38 x 4 halo tile in LDS
// (lane (p, half) reads channels 2*half, 2*half + 1 of pixel p + dx: 512 contiguous bytes per wave, conflict-free).
// Per 32 pixels x 32 channels: 49 LDS reads, 98 MFMAs, no VALU in the loop.
constexpr int C7_TH = 16, C7_TW = 32, C7_HH = C7_TH + 6, C7_HW = C7_TW + 6;

template <int NT>   // n-tiles of 32 output channels (power of two, <= 8); 8 waves = NT n-tiles x 8/NT row groups
__global__ __launch_bounds__(512, 1) void conv7x7_kernel(const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias,
                                                         float* __restrict__ out, int ldo, int B, int H, int W, int cout) {
    __shared__ __attribute__((aligned(16))) float tile[C7_HH * C7_HW * 4];
    constexpr int RG = 8 / NT;                           // row groups; a wave takes rows rg, rg + RG, ...
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
    const int nt = wave % NT, rg = wave / NT;
    const int n = nt * 32 + col;
    // ---- this wave's weights: packed [tap*4 + c][cout]
    float wr[98];
#pragma unroll
    for (int t = 0; t < 49; ++t)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) wr[2 * t + jj] = n < cout ? wp[(size_t)(t * 4 + jj + 2 * half) * cout + n] : 0.0f;
    f32x4 bias4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int n0 = nt * 32 + 8 * g + 4 * half;
#pragma unroll
        for (int i = 0; i < 4; ++i) bias4[g][i] = n0 + i < cout ? bias[n0 + i] : 0.0f;
    }
    const int tiles_x = (W + C7_TW - 1) / C7_TW, tiles_y = (H + C7_TH - 1) / C7_TH;
    const int n_tiles = B * tiles_y * tiles_x;
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
        const int ty0 = ty * C7_TH, tx0 = tx * C7_TW;
        __syncthreads();                                   // previous tile consumed
        for (int i = tid; i < C7_HH * C7_HW; i += 512) {
            const int hy = i / C7_HW, hx = i - hy * C7_HW;
            const int y = ty0 + hy - 3, xx = tx0 + hx - 3;
            f32x4 v = {0, 0, 0, 0};
            if (y >= 0 && y < H && xx >= 0 && xx < W) v = nd_ld4(x + ((size_t)(b * H + y) * W + xx) * 4);
            nd_st4(&tile[i * 4], v);
        }
        __syncthreads();
        for (int ly = rg; ly < C7_TH; ly += RG) {
            f32x16 acc;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[4 * g + i] = bias4[g][i];
            const float* base = &tile[(ly * C7_HW + col) * 4 + 2 * half];
#pragma unroll
            for (int ky = 0; ky < 7; ++ky)
#pragma unroll
                for (int kx = 0; kx < 7; ++kx) {
                    const float2 v = *reinterpret_cast<const float2*>(base + (ky * C7_HW + kx) * 4);
                    acc = nd_mfma(wr[2 * (ky * 7 + kx)], v.x, acc);
                    acc = nd_mfma(wr[2 * (ky * 7 + kx) + 1], v.y, acc);
                }
            const int y = ty0 + ly, xx = tx0 + col;
            if (y < H && xx < W) {
                float* o = out + ((size_t)(b * H + y) * W + xx) * ldo;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n0 = nt * 32 + 8 * g + 4 * half;
                    if (n0 + 4 <= cout) {
                        nd_st4(o + n0, f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]});
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (n0 + i < cout) o[n0 + i] = acc[4 * g + i];
                    }
                }
            }
        }
    }
}

// ---- the same stem on the bf16 matrix cores at full fp32 significand (r6: the split-product form of pointwise.hip / pwchain.hip).  K = 49 taps x 4 channels in
// 13 K steps of 16 (4 taps x 4 channels; taps 49-51 carry zero weights): slot i of lane (column, half h) in K step s = tap 4 s + 2 h + (i >> 2), channel i & 3.
// The weights of a wave's 32 couts sit in registers as three bf16 terms (13 x 3 x 4 = 156 registers, two waves per SIMD); the halo tile is split ONCE per element when
// it is staged into LDS term planes [term][pixel][4 x bf16]; a lane's operand of a K step is two 8-byte reads per term.  13 x 6 MFMAs of 32 cycles per 32-pixel row and
// 32 couts against 98 of 64.
typedef __bf16 c7_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 c7_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned c7_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned c7_u32x2 __attribute__((ext_vector_type(2)));
constexpr int C7_KS = 13;

__device__ __forceinline__ void c7_split2(float x, float y, unsigned& w1, unsigned& w2, unsigned& w3) {
    w1 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x, y}, c7_bf16x2));
    const float rx = x - __builtin_bit_cast(float, w1 << 16), ry = y - __builtin_bit_cast(float, w1 & 0xFFFF0000u);
    w2 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{rx, ry}, c7_bf16x2));
    const float sx = rx - __builtin_bit_cast(float, w2 << 16), sy = ry - __builtin_bit_cast(float, w2 & 0xFFFF0000u);
    w3 = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, sy), __builtin_bit_cast(unsigned, sx), 0x07060302u);
}

template <int NT>
__global__ __launch_bounds__(512, 1) void conv7x7_split_kernel(const float* __restrict__ x, const c7_u32x4* __restrict__ wp, const float* __restrict__ bias,
                                                               float* __restrict__ out, int ldo, int B, int H, int W, int cout) {
    constexpr int NPX = C7_HH * C7_HW;
    __shared__ __attribute__((aligned(16))) c7_u32x2 tile[3][NPX];                 // term planes: a pixel's four channels as bf16 = 8 bytes
    constexpr int RG = 8 / NT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
    const int nt = wave % NT, rg = wave / NT;
    // ---- this wave's weight terms: [cout tile][K step][term][lane] x 16 bytes
    c7_u32x4 wq[C7_KS][3];
#pragma unroll
    for (int s = 0; s < C7_KS; ++s)
#pragma unroll
        for (int t = 0; t < 3; ++t) wq[s][t] = wp[((nt * C7_KS + s) * 3 + t) * 64 + lane];
    f32x4 bias4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int n0 = nt * 32 + 8 * g + 4 * half;
#pragma unroll
        for (int i = 0; i < 4; ++i) bias4[g][i] = n0 + i < cout ? bias[n0 + i] : 0.0f;
    }
    // tile offsets (pixels) of this lane's two taps of every K step: tap 4 s + 2 half + j (the zero-weight padding taps read tap 48's pixel)
    int toff[C7_KS][2];
#pragma unroll
    for (int s = 0; s < C7_KS; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int tp = min(4 * s + 2 * half + j, 48);
            toff[s][j] = (tp / 7) * C7_HW + tp % 7;
        }
    const int tiles_x = (W + C7_TW - 1) / C7_TW, tiles_y = (H + C7_TH - 1) / C7_TH;
    const int n_tiles = B * tiles_y * tiles_x;
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
        const int ty0 = ty * C7_TH, tx0 = tx * C7_TW;
        __syncthreads();                                   // previous tile consumed
        for (int i = tid; i < NPX; i += 512) {
            const int hy = i / C7_HW, hx = i - hy * C7_HW;
            const int y = ty0 + hy - 3, xx = tx0 + hx - 3;
            f32x4 v = {0, 0, 0, 0};
            if (y >= 0 && y < H && xx >= 0 && xx < W) v = nd_ld4(x + ((size_t)(b * H + y) * W + xx) * 4);
            unsigned a1, a2, a3, b1, b2, b3;
            c7_split2(v.x, v.y, a1, a2, a3);
            c7_split2(v.z, v.w, b1, b2, b3);
            tile[0][i] = c7_u32x2{a1, b1};
            tile[1][i] = c7_u32x2{a2, b2};
            tile[2][i] = c7_u32x2{a3, b3};
        }
        __syncthreads();
        for (int ly = rg; ly < C7_TH; ly += RG) {
            f32x16 acc;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[4 * g + i] = bias4[g][i];
            const int p = ly * C7_HW + col;
#pragma unroll
            for (int s = 0; s < C7_KS; ++s) {
                c7_u32x4 xv[3];
#pragma unroll
                for (int tm = 0; tm < 3; ++tm) {
                    const c7_u32x2 lo = tile[tm][p + toff[s][0]], hi = tile[tm][p + toff[s][1]];
                    xv[tm] = c7_u32x4{lo.x, lo.y, hi.x, hi.y};
                }
#define C7_MFMA(wt, xt) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(c7_bf16x8, wq[s][wt]), __builtin_bit_cast(c7_bf16x8, xv[xt]), acc, 0, 0, 0)
                C7_MFMA(0, 0);  C7_MFMA(0, 1);  C7_MFMA(1, 0);  C7_MFMA(1, 1);  C7_MFMA(0, 2);  C7_MFMA(2, 0);
#undef C7_MFMA
            }
            const int y = ty0 + ly, xx = tx0 + col;
            if (y < H && xx < W) {
                float* o = out + ((size_t)(b * H + y) * W + xx) * ldo;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n0 = nt * 32 + 8 * g + 4 * half;
                    if (n0 + 4 <= cout) {
                        nd_st4(o + n0, f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]});
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (n0 + i < cout) o[n0 + i] = acc[4 * g + i];
                    }
                }
            }
        }
    }
}

// OIHW (cout, 4, 7, 7) -> [cout tile 32][K step 13][term 3][lane 64][8 bf16]: slot i of lane l = W[32 tile + (l & 31)][channel i & 3][tap 4 s + 2 (l >> 5) + (i >> 2)]
__global__ void pack_conv7x7_split_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int cout, int ntiles) {
    const int total = ntiles * C7_KS * 64 * 8;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int i = idx & 7, l = (idx >> 3) & 63, rest = idx >> 9;
        const int s = rest % C7_KS, tile = rest / C7_KS;
        const int n = 32 * tile + (l & 31), tap = 4 * s + 2 * (l >> 5) + (i >> 2), c = i & 3;
        const float v = (n < cout && tap < 49) ? w[((size_t)n * 4 + c) * 49 + tap] : 0.0f;
        const __bf16 t1 = (__bf16)v;
        const float r1 = v - (float)t1;
        const __bf16 t2 = (__bf16)r1;
        const float r2 = r1 - (float)t2;
        unsigned short* o = out + ((size_t)((tile * C7_KS + s) * 3) * 64 + l) * 8 + i;
        o[0] = __builtin_bit_cast(unsigned short, t1);
        o[512] = __builtin_bit_cast(unsigned short, t2);
        o[1024] = (unsigned short)(__builtin_bit_cast(unsigned, r2) >> 16);
    }
}

__global__ void pack_conv7x7_kernel(const float* __restrict__ w, float* __restrict__ out, int cout) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // over [49*4][cout]
    if (i >= 196 * cout) return;
    const int n = i % cout, k = i / cout, ci = k & 3, tap = k >> 2;
    out[i] = w[((size_t)n * 4 + ci) * 49 + tap];
}

inline int grid_for(size_t total, int cap = 4096) {
    const size_t g = (total + 255) / 256;
    return (int)(g < (size_t)cap ? (g ? g : 1) : cap);
}


// ---- nd_cond_step_f32: the whole time conditioning of one diffusion step in ONE launch:
// SinusoidalPosEmb (:100-107) -> time_mlp = Linear(d, 4d), GELU, Linear(4d, 4d) (:502-507) -> the SiLU every ResnetBlock.mlp
// applies first (:149) -> all ResnetBlock.mlp Linears stacked into one (J, 4d) matrix (:150-152).  Every workgroup recomputes the
// small head (emb, t1, st: 0.3 MFLOP per sample at d = 64) into LDS -- cheaper than a grid-wide hand-off -- and then produces its own
// slice of the J output rows, one wave per row as nd_linear_rows_f32 does.
__global__ __launch_bounds__(1024) void cond_step_kernel(const int64_t* __restrict__ time, const float* __restrict__ freqs,
                                                         const float* __restrict__ W1, const float* __restrict__ b1,
                                                         const float* __restrict__ W2, const float* __restrict__ b2,
                                                         const float* __restrict__ Wp, const float* __restrict__ bp,
                                                         float* __restrict__ out, int ld_out, int B, int d, int J,
                                                         const float* __restrict__ table, int table_rows, float* __restrict__ table_out,
                                                         const float* __restrict__ ptable = nullptr) {
    // `ptable` (rows x J): the WHOLE result for timestep = row index (nd_cond_step_ptable_f32; built by nd_cond_proj_table_build_f32 from this kernel's own rows, so a
    // lookup is the very same bits): when every sample's timestep is inside it, the step's conditioning is a copy of B rows (48 -> a few us at the start of every step);
    // any timestep outside it sends the whole batch down the computing path.
    if (ptable && !table_out) {
        bool all_in = true;
        for (int b = 0; b < B; ++b) all_in = all_in && time[b] >= 0 && time[b] < table_rows;     // (uniform: every thread reads the same B values)
        if (all_in) {
            const int J4 = J >> 2;
            for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)B * J4; i += (long)gridDim.x * blockDim.x) {
                const int b = (int)(i / J4), j = (int)(i - (long)b * J4) * 4;
                nd_st4(out + (size_t)b * ld_out + j, nd_ld4(ptable + (size_t)time[b] * J + j));
            }
            return;
        }
    }
    // `table` (rows x 4d): the head's result st for timestep = row index, built once per weight set by this kernel itself (`table_out`
    // mode: workgroup w computes the head for timesteps 16 w .. 16 w + 15 and writes it instead of projecting).  With a table whose rows
    // cover every sample's timestep the head -- two dependent small Linears behind barriers, ~50 us of latency at the start of EVERY diffusion
    // step -- is a lookup of the very same bits.
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int D4 = 4 * d, half = d / 2, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    float* emb = sm;                    // [B][d]
    float* t1 = emb + B * d;            // [B][4d]
    float* st = t1 + B * D4;            // [B][4d]
    const int t_first = table_out ? (int)blockIdx.x * B : 0;             // build mode: this workgroup's timesteps
    if (table_out) { if (t_first >= table_rows) return;  B = min(B, table_rows - t_first); }
    bool lookup = table != nullptr && !table_out;
    if (lookup)
        for (int b = 0; b < B; ++b) lookup = lookup && time[b] >= 0 && time[b] < table_rows;     // (uniform: every thread reads the same B values)
    if (lookup) {
        for (int i = tid; i < B * D4; i += blockDim.x) {
            const int b = i / D4;
            st[i] = table[(size_t)time[b] * D4 + (i - b * D4)];
        }
        __syncthreads();
    } else {
    for (int i = tid; i < B * half; i += blockDim.x) {
        const int b = i / half, j = i - b * half;
        const float ang = (float)(table_out ? (int64_t)(t_first + b) : time[b]) * freqs[j];
        emb[b * d + j] = sinf(ang);
        emb[b * d + half + j] = cosf(ang);
    }
    __syncthreads();
    // the two small Linears: 16 lanes per output column (coalesced 256-byte pieces of its weight row, the activations from LDS),
    // four columns per wave, the batch in registers (B <= 16 per pass); the 16 partial dot products meet in a DPP row sum
    const int grp = lane >> 4, l16 = lane & 15;
    auto linear = [&](const float* x, int K, const float* W, const float* bias, float* y, int act) {
        for (int n = wave * 4 + grp; n < D4; n += nwaves * 4) {
            for (int b0 = 0; b0 < B; b0 += 16) {
                float acc[16];
#pragma unroll
                for (int b = 0; b < 16; ++b) acc[b] = 0.0f;
                for (int k = l16 * 4; k < K; k += 64) {
                    const f32x4 w = nd_ld4(W + (size_t)n * K + k);
#pragma unroll
                    for (int b = 0; b < 16; ++b) {
                        if (b0 + b < B) {
                            const f32x4 xv = nd_ld4(x + (b0 + b) * K + k);
                            acc[b] = fmaf(xv.w, w.w, fmaf(xv.z, w.z, fmaf(xv.y, w.y, fmaf(xv.x, w.x, acc[b]))));
                        }
                    }
                }
                const float bv = bias[n];
#pragma unroll
                for (int b = 0; b < 16; ++b) {
                    const float sum = nd_row16_sum(acc[b]);       // (every lane of the wave takes part: the loop bounds are wave-uniform)
                    if (l16 == 0 && b0 + b < B) y[(b0 + b) * D4 + n] = nd_act(sum + bv, act);
                }
            }
        }
    };
    linear(emb, d, W1, b1, t1, ND_ACT_GELU);
    __syncthreads();
    linear(t1, D4, W2, b2, st, ND_ACT_SILU);
    __syncthreads();
    }
    if (table_out) {
        for (int i = tid; i < B * D4; i += blockDim.x) table_out[(size_t)t_first * D4 + i] = st[i];
        return;
    }
    // this workgroup's rows of the stacked projection: one wave per row, the weight row in registers (4d <= 2048)
    for (int n = blockIdx.x * nwaves + wave; n < J; n += gridDim.x * nwaves) {
        float w[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int k = lane + j * 64;
            w[j] = k < D4 ? Wp[(size_t)n * D4 + k] : 0.0f;
        }
        const float bv = bp[n];
        for (int b = 0; b < B; ++b) {
            float sum = 0.0f;
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const int k = lane + j * 64;
                if (k < D4) sum += st[b * D4 + k] * w[j];
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
            if (lane == 0) out[(size_t)b * ld_out + n] = sum + bv;
        }
    }
}

}  // namespace

extern "C" int nd_linear_rows_f32(const float* in, int ld_in, const float* W, const float* bias, float* out, int ld_out, int B,
                                  int K, int N, int act_in, int act_out, void* stream) {
    ND_REQUIRE(in && W && out, ND_E_BADARG, "nd_linear_rows: null pointer");
    ND_REQUIRE(B > 0 && K > 0 && N > 0 && K <= 2048 && ld_in >= K && ld_out >= N, ND_E_SHAPE, "nd_linear_rows: B=%d K=%d N=%d (K <= 2048)", B, K, N);
    hipLaunchKernelGGL(linear_rows_kernel, dim3(nd_cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, in, ld_in, W, bias, out, ld_out, B, K, N, act_in, act_out);
    return nd_launch_status("nd_linear_rows_f32");
}

extern "C" int64_t nd_cond_step_lds_bytes(int B, int dim) { return (int64_t)B * dim * 9 * (int64_t)sizeof(float); }

extern "C" int nd_cond_step_f32(const int64_t* time, const float* freqs, const float* W1, const float* b1, const float* W2, const float* b2,
                                const float* Wp, const float* bp, float* out, int ld_out, int B, int dim, int J, void* stream) {
    ND_REQUIRE(time && freqs && W1 && b1 && W2 && b2 && Wp && bp && out, ND_E_BADARG, "nd_cond_step: null pointer");
    ND_REQUIRE(B > 0 && dim >= 8 && dim % 8 == 0 && 4 * dim <= 2048 && J > 0 && ld_out >= J, ND_E_SHAPE,
               "nd_cond_step: B=%d dim=%d J=%d (dim a multiple of 8, 4 dim <= 2048)", B, dim, J);
    ND_REQUIRE(nd_aligned16(W1) && nd_aligned16(W2), ND_E_ALIGN, "nd_cond_step: time_mlp weights must be 16-byte aligned");
    const int64_t lds = nd_cond_step_lds_bytes(B, dim);
    ND_REQUIRE(lds <= 160 * 1024, ND_E_SHAPE, "nd_cond_step: B * dim = %d needs %lld bytes of LDS (use the separate launches)", B * dim, (long long)lds);
    static nd_device_once configured;
    if (lds > 64 * 1024)
        if (int e = nd_reserve_lds(configured, reinterpret_cast<const void*>(cond_step_kernel), 160 * 1024, "nd_cond_step")) return e;
    const int rows16 = nd_cdiv(J, 16), cus = nd_device_cus();
    hipLaunchKernelGGL(cond_step_kernel, dim3(rows16 < cus ? rows16 : cus), dim3(1024), (size_t)lds, (hipStream_t)stream, time, freqs, W1, b1, W2, b2, Wp, bp,
                       out, ld_out, B, dim, J, (const float*)nullptr, 0, (float*)nullptr);
    return nd_launch_status("nd_cond_step_f32");
}

extern "C" int nd_cond_table_build_f32(const float* freqs, const float* W1, const float* b1, const float* W2, const float* b2, float* table, int rows,
                                       int dim, void* stream) {
    ND_REQUIRE(freqs && W1 && b1 && W2 && b2 && table, ND_E_BADARG, "nd_cond_table_build: null pointer");
    ND_REQUIRE(rows > 0 && dim >= 8 && dim % 8 == 0 && 4 * dim <= 2048, ND_E_SHAPE, "nd_cond_table_build: rows=%d dim=%d", rows, dim);
    ND_REQUIRE(nd_aligned16(W1) && nd_aligned16(W2), ND_E_ALIGN, "nd_cond_table_build: time_mlp weights must be 16-byte aligned");
    const int Bc = 16;                                                   // timesteps per workgroup: the head's batch size in the step kernel
    const int64_t lds = nd_cond_step_lds_bytes(Bc, dim);
    ND_REQUIRE(lds <= 160 * 1024, ND_E_SHAPE, "nd_cond_table_build: dim=%d needs %lld bytes of LDS (no table for this width: use nd_cond_step_f32 or the separate launches)",
               dim, (long long)lds);
    static nd_device_once configured;
    if (lds > 64 * 1024)
        if (int e = nd_reserve_lds(configured, reinterpret_cast<const void*>(cond_step_kernel), 160 * 1024, "nd_cond_table_build")) return e;
    hipLaunchKernelGGL(cond_step_kernel, dim3(nd_cdiv(rows, Bc)), dim3(1024), (size_t)lds, (hipStream_t)stream, (const int64_t*)nullptr, freqs, W1, b1, W2, b2,
                       (const float*)nullptr, (const float*)nullptr, (float*)nullptr, 0, Bc, dim, 0, (const float*)nullptr, rows, table);
    return nd_launch_status("nd_cond_table_build_f32");
}

extern "C" int nd_cond_step_table_f32(const int64_t* time, const float* freqs, const float* W1, const float* b1, const float* W2, const float* b2,
                                      const float* Wp, const float* bp, float* out, int ld_out, int B, int dim, int J, const float* table, int table_rows,
                                      void* stream) {
    ND_REQUIRE(time && freqs && W1 && b1 && W2 && b2 && Wp && bp && out && table, ND_E_BADARG, "nd_cond_step_table: null pointer");
    ND_REQUIRE(B > 0 && dim >= 8 && dim % 8 == 0 && 4 * dim <= 2048 && J > 0 && ld_out >= J && table_rows > 0, ND_E_SHAPE,
               "nd_cond_step_table: B=%d dim=%d J=%d rows=%d (dim a multiple of 8, 4 dim <= 2048)", B, dim, J, table_rows);
    ND_REQUIRE(nd_aligned16(W1) && nd_aligned16(W2), ND_E_ALIGN, "nd_cond_step_table: time_mlp weights must be 16-byte aligned");
    const int64_t lds = nd_cond_step_lds_bytes(B, dim);
    ND_REQUIRE(lds <= 160 * 1024, ND_E_SHAPE, "nd_cond_step_table: B * dim = %d needs %lld bytes of LDS", B * dim, (long long)lds);
    static nd_device_once configured;
    if (lds > 64 * 1024)
        if (int e = nd_reserve_lds(configured, reinterpret_cast<const void*>(cond_step_kernel), 160 * 1024, "nd_cond_step_table")) return e;
    const int rows16 = nd_cdiv(J, 16), cus = nd_device_cus();
    hipLaunchKernelGGL(cond_step_kernel, dim3(rows16 < cus ? rows16 : cus), dim3(1024), (size_t)lds, (hipStream_t)stream, time, freqs, W1, b1, W2, b2, Wp, bp,
                       out, ld_out, B, dim, J, table, table_rows, (float*)nullptr);
    return nd_launch_status("nd_cond_step_table_f32");
}

// ... with the projection itself tabulated (`ptable`: rows x J, row t = what this entry computes for timestep t; J and ld_out multiples of 4)
extern "C" int nd_cond_step_ptable_f32(const int64_t* time, const float* freqs, const float* W1, const float* b1, const float* W2, const float* b2,
                                       const float* Wp, const float* bp, float* out, int ld_out, int B, int dim, int J, const float* table, int table_rows,
                                       const float* ptable, void* stream) {
    ND_REQUIRE(time && freqs && W1 && b1 && W2 && b2 && Wp && bp && out && table && ptable, ND_E_BADARG, "nd_cond_step_ptable: null pointer");
    ND_REQUIRE(B > 0 && dim >= 8 && dim % 8 == 0 && 4 * dim <= 2048 && J > 0 && J % 4 == 0 && ld_out >= J && ld_out % 4 == 0 && table_rows > 0, ND_E_SHAPE,
               "nd_cond_step_ptable: B=%d dim=%d J=%d ld_out=%d rows=%d (dim a multiple of 8, 4 dim <= 2048, J and ld_out multiples of 4)", B, dim, J, ld_out, table_rows);
    ND_REQUIRE(nd_aligned16(W1) && nd_aligned16(W2) && nd_aligned16(out) && nd_aligned16(ptable), ND_E_ALIGN, "nd_cond_step_ptable: weights, out and ptable must be 16-byte aligned");
    const int64_t lds = nd_cond_step_lds_bytes(B, dim);
    ND_REQUIRE(lds <= 160 * 1024, ND_E_SHAPE, "nd_cond_step_ptable: B * dim = %d needs %lld bytes of LDS", B * dim, (long long)lds);
    static nd_device_once configured;
    if (lds > 64 * 1024)
        if (int e = nd_reserve_lds(configured, reinterpret_cast<const void*>(cond_step_kernel), 160 * 1024, "nd_cond_step_ptable")) return e;
    const int rows16 = nd_cdiv(J, 16), cus = nd_device_cus();
    hipLaunchKernelGGL(cond_step_kernel, dim3(rows16 < cus ? rows16 : cus), dim3(1024), (size_t)lds, (hipStream_t)stream, time, freqs, W1, b1, W2, b2, Wp, bp,
                       out, ld_out, B, dim, J, table, table_rows, (float*)nullptr, ptable);
    return nd_launch_status("nd_cond_step_ptable_f32");
}

extern "C" int nd_sinusoidal_time_emb_f32(const int64_t* time, const float* freqs, float* emb, int B, int half, void* stream) {
    ND_REQUIRE(time && freqs && emb && B > 0 && half > 0, ND_E_BADARG, "nd_sinusoidal_time_emb: bad argument");
    hipLaunchKernelGGL(sinusoidal_kernel, dim3(nd_cdiv(B * half, 256)), dim3(256), 0, (hipStream_t)stream, time, freqs, emb, B, half);
    return nd_launch_status("nd_sinusoidal_time_emb_f32");
}

extern "C" int nd_embedding_rows_f32(const int64_t* idx, const float* table, float* out, int B, int rows, int dim, void* stream) {
    ND_REQUIRE(idx && table && out && B > 0 && rows > 0 && dim > 0, ND_E_BADARG, "nd_embedding_rows: bad argument");
    hipLaunchKernelGGL(embedding_kernel, dim3(nd_cdiv(B * dim, 256)), dim3(256), 0, (hipStream_t)stream, idx, table, out, B, rows, dim);
    return nd_launch_status("nd_embedding_rows_f32");
}

extern "C" int nd_pos_enc_f32(const float* position_nchw, const float* w, const float* bias, float* out, int B, int H, int W, int hid, void* stream) {
    ND_REQUIRE(position_nchw && w && bias && out && B > 0 && H > 0 && W > 0 && hid > 0, ND_E_BADARG, "nd_pos_enc: bad argument");
    const size_t total = (size_t)B * H * W * hid;
    hipLaunchKernelGGL(pos_enc_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, position_nchw, w, bias, out, B, H * W, hid);
    return nd_launch_status("nd_pos_enc_f32");
}

extern "C" int nd_nchw_to_nhwc_f32(const float* in, float* out, int B, int C, int H, int W, void* stream) {
    ND_REQUIRE(in && out && B > 0 && C > 0 && H > 0 && W > 0, ND_E_BADARG, "nd_nchw_to_nhwc: bad argument");
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for((size_t)B * C * H * W)), dim3(256), 0, (hipStream_t)stream, in, out, B, C, H * W);
    return nd_launch_status("nd_nchw_to_nhwc_f32");
}

extern "C" int nd_nchw_to_nhwc_pad_f32(const float* in, float* out, int B, int C, int H, int W, int Cpad, void* stream) {
    ND_REQUIRE(in && out && B > 0 && C > 0 && H > 0 && W > 0 && Cpad >= C, ND_E_BADARG, "nd_nchw_to_nhwc_pad: bad argument");
    hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel, dim3(grid_for((size_t)B * Cpad * H * W)), dim3(256), 0, (hipStream_t)stream, in, out, B, C, H * W, Cpad);
    return nd_launch_status("nd_nchw_to_nhwc_pad_f32");
}

extern "C" int nd_maxpool2x2_nhwc_f32(const float* in, float* out, int B, int H, int W, int C, void* stream) {
    ND_REQUIRE(in && out && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, ND_E_BADARG, "nd_maxpool2x2: bad argument (C %% 4 == 0)");
    ND_REQUIRE(nd_aligned16(in) && nd_aligned16(out), ND_E_ALIGN, "nd_maxpool2x2: alignment");
    const size_t total = (size_t)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
    hipLaunchKernelGGL(maxpool2x2_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, out, B, H, W, C);
    return nd_launch_status("nd_maxpool2x2_nhwc_f32");
}

extern "C" int nd_nhwc_to_nchw_f32(const float* in, float* out, int B, int C, int H, int W, void* stream) {
    ND_REQUIRE(in && out && B > 0 && C > 0 && H > 0 && W > 0, ND_E_BADARG, "nd_nhwc_to_nchw: bad argument");
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for((size_t)B * C * H * W)), dim3(256), 0, (hipStream_t)stream, in, out, B, C, H * W);
    return nd_launch_status("nd_nhwc_to_nchw_f32");
}

extern "C" int nd_pack_conv7x7_weight(const float* oihw, float* packed, int cout, void* stream) {
    ND_REQUIRE(oihw && packed && cout > 0, ND_E_BADARG, "nd_pack_conv7x7_weight: bad argument");
    hipLaunchKernelGGL(pack_conv7x7_kernel, dim3(nd_cdiv(196 * cout, 256)), dim3(256), 0, (hipStream_t)stream, oihw, packed, cout);
    return nd_launch_status("nd_pack_conv7x7_weight");
}

extern "C" int nd_conv7x7_c4_f32(const float* x, const float* wpacked, const float* bias, float* out, int ldo, int B, int H, int W,
                                 int cout, void* stream) {
    ND_REQUIRE(x && wpacked && bias && out, ND_E_BADARG, "nd_conv7x7_c4: null pointer");
    ND_REQUIRE(B > 0 && H > 0 && W > 0 && cout > 0 && ldo >= cout, ND_E_SHAPE, "nd_conv7x7_c4: bad shape");
    ND_REQUIRE(nd_aligned16(x), ND_E_ALIGN, "nd_conv7x7_c4: x must be 16-byte aligned");
    ND_REQUIRE(cout <= 256 && ldo % 4 == 0 && nd_aligned16(out), ND_E_SHAPE, "nd_conv7x7_c4: cout <= 256, ldo %% 4 == 0, out 16-byte aligned");
    const long tiles = (long)B * nd_cdiv(H, C7_TH) * nd_cdiv(W, C7_TW);
    ND_REQUIRE(tiles < (1L << 31), ND_E_SHAPE, "nd_conv7x7_c4: grid too large");
    const int cus = nd_device_cus();
    const dim3 grid((unsigned)(tiles < cus ? tiles : cus)), block(512);
    const int ntiles = nd_cdiv(cout, 32);
#define ND_C7_LAUNCH(NT) hipLaunchKernelGGL(conv7x7_kernel<NT>, grid, block, 0, (hipStream_t)stream, x, wpacked, bias, out, ldo, B, H, W, cout)
    if (ntiles <= 1) ND_C7_LAUNCH(1);
    else if (ntiles <= 2) ND_C7_LAUNCH(2);
    else if (ntiles <= 4) ND_C7_LAUNCH(4);
    else ND_C7_LAUNCH(8);
#undef ND_C7_LAUNCH
    return nd_launch_status("nd_conv7x7_c4_f32");
}

extern "C" int64_t nd_pack_conv7x7_weight_split_floats(int cout) { return (int64_t)nd_cdiv(cout, 32) * C7_KS * 3 * 64 * 4; }

extern "C" int nd_pack_conv7x7_weight_split(const float* oihw, float* packed, int cout, void* stream) {
    ND_REQUIRE(oihw && packed && cout > 0 && nd_aligned16(packed), ND_E_BADARG, "nd_pack_conv7x7_weight_split: bad argument");
    const int ntiles = nd_cdiv(cout, 32);
    hipLaunchKernelGGL(pack_conv7x7_split_kernel, dim3(nd_cdiv(ntiles * C7_KS * 512, 256)), dim3(256), 0, (hipStream_t)stream, oihw,
                       reinterpret_cast<unsigned short*>(packed), cout, ntiles);
    return nd_launch_status("nd_pack_conv7x7_weight_split");
}

// nd_conv7x7_c4_f32 with the products on the bf16 matrix cores at full fp32 significand (three-term split, six products, fp32 accumulation);
// `wsplit` from nd_pack_conv7x7_weight_split.  Same shapes and errors.
extern "C" int nd_conv7x7_c4_split_f32(const float* x, const float* wsplit, const float* bias, float* out, int ldo, int B, int H, int W,
                                       int cout, void* stream) {
    ND_REQUIRE(x && wsplit && bias && out, ND_E_BADARG, "nd_conv7x7_c4_split: null pointer");
    ND_REQUIRE(B > 0 && H > 0 && W > 0 && cout > 0 && ldo >= cout, ND_E_SHAPE, "nd_conv7x7_c4_split: bad shape");
    ND_REQUIRE(nd_aligned16(x) && nd_aligned16(wsplit), ND_E_ALIGN, "nd_conv7x7_c4_split: x and the weights must be 16-byte aligned");
    ND_REQUIRE(cout <= 256 && ldo % 4 == 0 && nd_aligned16(out), ND_E_SHAPE, "nd_conv7x7_c4_split: cout <= 256, ldo %% 4 == 0, out 16-byte aligned");
    const long tiles = (long)B * nd_cdiv(H, C7_TH) * nd_cdiv(W, C7_TW);
    ND_REQUIRE(tiles < (1L << 31), ND_E_SHAPE, "nd_conv7x7_c4_split: grid too large");
    const int cus = nd_device_cus();
    const dim3 grid((unsigned)(tiles < cus ? tiles : cus)), block(512);
    const int ntiles = nd_cdiv(cout, 32);
    const c7_u32x4* wq = reinterpret_cast<const c7_u32x4*>(wsplit);
#define ND_C7S_LAUNCH(NT) hipLaunchKernelGGL(conv7x7_split_kernel<NT>, grid, block, 0, (hipStream_t)stream, x, wq, bias, out, ldo, B, H, W, cout)
    if (ntiles <= 1) ND_C7S_LAUNCH(1);
    else if (ntiles <= 2) ND_C7S_LAUNCH(2);
    else if (ntiles <= 4) ND_C7S_LAUNCH(4);
    else ND_C7S_LAUNCH(8);
#undef ND_C7S_LAUNCH
    return nd_launch_status("nd_conv7x7_c4_split_f32");
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Weight (and bias) gradient of the 7x7 stem over a 4-channel image (init_conv / cond_init_conv: Diffusion_arch.py:478; training path, SURVEY
// 8f-4): dw[co][ci][ky][kx] = sum over the pixels of dy[p][co] x[p + (ky - 3, kx - 3)][ci], zero padding.  A GEMM with K = pixels, M = cout,
// N = 196 = (tap, ci): the patch matrix is never built (r5 first form: F.unfold + a transposed copy, 205 MB and 330 us per step for the one layer).
// A workgroup walks tiles of 4 x 32 pixels: the 10 x 38 x 4 halo of x and the 128 x cout block of dy in LDS, v_mfma_f32_16x16x4_f32 with K = four
// consecutive pixels of a row -- A = dy (lane (cout, pixel)), B = x[pixel + tap][ci] gathered from the halo (lane (n, pixel); n >= 196 reads zeros).
// The 13 column blocks of 16 go to the four waves as 4 + 3 + 3 + 3, the heavy share rotating with the workgroup; accumulators stay in registers over
// all tiles of the workgroup, partial sums [workgroup][cout][208] + [workgroup][cout] to the workspace, summed in workgroup order by the second kernel.
namespace {
constexpr int C7W_TH = 4, C7W_TW = 32, C7W_PX = C7W_TH * C7W_TW;     // pixels per tile
constexpr int C7W_HR = C7W_TH + 6, C7W_HC = C7W_TW + 6;              // halo 10 x 38
constexpr int C7W_NB = 13, C7W_N = 16 * C7W_NB;                      // column blocks, padded N
constexpr int C7W_WGS = 512;                                         // fixed: the summation order must not depend on the device

struct C7wArgs {
    const float* x; const float* dy; float* ws; float* wsb;
    int ldy, B, H, W, cout, tiles_x, tiles_y, n_tiles, n_wg;
};

__host__ __device__ constexpr int c7w_stride(int MB) { return (16 * MB + 63) / 64 * 64 + 16; }      // floats per dy pixel in LDS: = 16 mod 64 (the four pixels of a K step on distinct banks)

template <int MB>
__global__ __launch_bounds__(256) void c7_wgrad_kernel(const C7wArgs a) {
    constexpr int YS = c7w_stride(MB);
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Xs = sm;                                                  // [10][38][4] (+ 8 zeros: what a lane of a padded column reads)
    float* Ys = sm + C7W_HR * C7W_HC * 4 + 8;                        // [128][YS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, k = lane >> 4;
    const int role = (wave + (int)blockIdx.x / 256) & 3;             // which share of the column blocks: role, role + 4, role + 8 (+ 12 for role 0)
    const int nblk = role == 0 ? 4 : 3;
    int boff[4];                                                     // halo offset of this lane's column n = 16 (role + 4 j) + m: ((ky 38 + kx) 4 + ci) floats
    bool bok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = 16 * (role + 4 * j) + m, tap = n >> 2, ci = n & 3;
        bok[j] = j < nblk && tap < 49;
        boff[j] = bok[j] ? ((tap / 7) * C7W_HC + tap % 7) * 4 + ci : 0;
    }
    f32x4 acc[MB][4];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[mb][j] = f32x4{0, 0, 0, 0};
    float bs[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) bs[mb] = 0.0f;
    const bool do_bias = a.wsb != nullptr && wave == 0;

    const int t_lo = (int)((long)blockIdx.x * a.n_tiles / a.n_wg), t_hi = (int)((long)(blockIdx.x + 1) * a.n_tiles / a.n_wg);
    for (int tile = t_lo; tile < t_hi; ++tile) {
        const int per = a.tiles_x * a.tiles_y, b = tile / per, r_ = tile - b * per, ty = r_ / a.tiles_x, tx = r_ - ty * a.tiles_x;
        const int y0 = ty * C7W_TH, x0 = tx * C7W_TW;
        __syncthreads();                                             // the previous tile's operands are consumed
        for (int idx = tid; idx < C7W_HR * C7W_HC; idx += 256) {     // halo: one float4 (the pixel's four channels) per item, zero outside the image
            const int r = idx / C7W_HC, c = idx - r * C7W_HC, y = y0 - 3 + r, xx = x0 - 3 + c;
            const f32x4 zero = {0, 0, 0, 0};
            nd_st4(Xs + idx * 4, ((unsigned)y < (unsigned)a.H && (unsigned)xx < (unsigned)a.W) ? nd_ld4(a.x + ((size_t)(b * a.H + y) * a.W + xx) * 4) : zero);
        }
        for (int idx = tid; idx < C7W_PX * 4 * MB; idx += 256) {     // dy block: float4 = (pixel, channel quad)
            const int p = idx / (4 * MB), q = idx - p * (4 * MB), py = p >> 5, px = p & 31;
            nd_st4(Ys + p * YS + 4 * q, nd_ld4(a.dy + ((size_t)(b * a.H + y0 + py) * a.W + x0 + px) * a.ldy + 4 * q));
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < C7W_PX / 4; ++kk) {                    // K step = pixels 4 kk .. 4 kk + 3 of the tile (one row segment)
            const int pbase = ((kk >> 3) * C7W_HC + 4 * (kk & 7)) * 4;       // halo offset of pixel 4 kk's patch origin
            float av[MB], bv[4];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) av[mb] = Ys[(4 * kk + k) * YS + 16 * mb + m];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = Xs[pbase + 4 * k + boff[j]];
                bv[j] = bok[j] ? v : 0.0f;
            }
            if (do_bias) {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) bs[mb] += av[mb];
            }
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < 3 || role == 0) acc[mb][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mb], bv[j], acc[mb][j], 0, 0, 0);
        }
    }
    // partial sums: ws[wg][co][n] (n = 16 (role + 4 j) + (lane & 15); rows 4 (lane >> 4) + r of the block), wsb[wg][co]
    float* o = a.ws + (size_t)blockIdx.x * a.cout * C7W_N;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < nblk)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[(size_t)(16 * mb + 4 * k + r) * C7W_N + 16 * (role + 4 * j) + m] = acc[mb][j][r];
    if (do_bias) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            float v = bs[mb];                                        // lanes (m, k): the four pixel slots of a K step meet in a fixed order
            v += __shfl_down(v, 32);
            v += __shfl_down(v, 16);
            if (lane < 16) a.wsb[(size_t)blockIdx.x * a.cout + 16 * mb + m] = v;
        }
    }
}

// dw (cout, 4, 7, 7) and db from the partial sums: four lanes share an output, lane q adds workgroups q, q + 4, ... in order, the four meet in lane order
__global__ __launch_bounds__(256) void c7_wgrad_reduce_kernel(const float* __restrict__ ws, const float* __restrict__ wsb, float* __restrict__ dw, float* __restrict__ db,
                                                              int n_wg, int cout) {
    const int total = cout * 196 + (db ? cout : 0);
    const int i = (blockIdx.x * 256 + threadIdx.x) >> 2, q = threadIdx.x & 3;
    const bool live = i < total;
    const int ii = live ? i : 0;
    const bool bias = ii >= cout * 196;
    const int co = bias ? ii - cout * 196 : ii / 196, e = bias ? 0 : ii - co * 196;                   // e = ci 49 + tap in dw
    const int ci = e / 49, tap = e - ci * 49;
    const float* p = bias ? wsb + co : ws + (size_t)co * C7W_N + tap * 4 + ci;
    const size_t stride = bias ? (size_t)cout : (size_t)cout * C7W_N;
    float sum = 0.0f;
    int s = q;
    for (; s + 28 < n_wg; s += 32) {                                     // eight loads in flight
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = p[(size_t)(s + 4 * k) * stride];
#pragma unroll
        for (int k = 0; k < 8; ++k) sum += v[k];
    }
    for (; s < n_wg; s += 4) sum += p[(size_t)s * stride];
    sum += __shfl_down(sum, 1, 4) ;                                      // (q, q + 1), then (0 + 1) + (2 + 3)
    sum += __shfl_down(sum, 2, 4);
    if (live && q == 0) { if (bias) db[co] = sum; else dw[i] = sum; }
}

bool c7w_takes(int B, int H, int W, int cout) {
    return B > 0 && H % C7W_TH == 0 && W % C7W_TW == 0 && (cout == 32 || cout == 48 || cout == 64 || cout == 96 || cout == 128) && (long)B * H * W < (1L << 28);
}
int c7w_wgs(int B, int H, int W) {
    const int tiles = B * (H / C7W_TH) * (W / C7W_TW);
    return tiles < C7W_WGS ? tiles : C7W_WGS;
}

template <int MB>
int c7w_launch(const C7wArgs& a, hipStream_t st) {
    const size_t lds = (size_t)(C7W_HR * C7W_HC * 4 + 8 + C7W_PX * c7w_stride(MB)) * sizeof(float);
    static nd_device_once configured;
    if (int e = nd_reserve_lds(configured, reinterpret_cast<const void*>(c7_wgrad_kernel<MB>), lds, "nd_conv7x7_c4_wgrad_f32")) return e;
    hipLaunchKernelGGL(c7_wgrad_kernel<MB>, dim3((unsigned)a.n_wg), dim3(256), lds, st, a);
    return nd_launch_status("nd_conv7x7_c4_wgrad_f32");
}
}  // namespace

extern "C" int64_t nd_conv7x7_c4_wgrad_workspace_floats(int B, int H, int W, int cout) {
    if (!c7w_takes(B, H, W, cout)) return -1;                        // the shape is not taken (callers unfold the image and use nd_linear_wgrad_f32)
    return (int64_t)c7w_wgs(B, H, W) * cout * (C7W_N + 1);
}

extern "C" int nd_conv7x7_c4_wgrad_f32(const float* x, const float* dy, int ldy, float* dw_oihw, float* dbias, float* workspace, int B, int H, int W,
                                       int cout, void* stream) {
    ND_REQUIRE(x && dy && dw_oihw && workspace, ND_E_BADARG, "nd_conv7x7_c4_wgrad_f32: null pointer");
    ND_REQUIRE(c7w_takes(B, H, W, cout), ND_E_SHAPE, "nd_conv7x7_c4_wgrad_f32: needs H %% 4 == 0, W %% 32 == 0 and cout in {32, 48, 64, 96, 128} (H=%d W=%d cout=%d): "
               "unfold the image and use nd_linear_wgrad_f32", H, W, cout);
    ND_REQUIRE(ldy >= cout && ldy % 4 == 0 && nd_aligned16(x) && nd_aligned16(dy), ND_E_ALIGN, "nd_conv7x7_c4_wgrad_f32: dy rows must be 16-byte aligned, ldy >= cout");
    C7wArgs a;
    a.x = x; a.dy = dy; a.ldy = ldy; a.B = B; a.H = H; a.W = W; a.cout = cout;
    a.tiles_x = W / C7W_TW; a.tiles_y = H / C7W_TH; a.n_tiles = B * a.tiles_x * a.tiles_y; a.n_wg = c7w_wgs(B, H, W);
    a.ws = workspace; a.wsb = dbias ? workspace + (size_t)a.n_wg * cout * C7W_N : nullptr;
    hipStream_t st = (hipStream_t)stream;
    int e;
    switch (cout / 16) {
        case 2: e = c7w_launch<2>(a, st); break;
        case 3: e = c7w_launch<3>(a, st); break;
        case 4: e = c7w_launch<4>(a, st); break;
        case 6: e = c7w_launch<6>(a, st); break;
        default: e = c7w_launch<8>(a, st); break;
    }
    if (e) return e;
    const int total = cout * 196 + (dbias ? cout : 0);
    hipLaunchKernelGGL(c7_wgrad_reduce_kernel, dim3((unsigned)((4 * total + 255) / 256)), dim3(256), 0, st, workspace, a.wsb, dw_oihw, dbias, a.n_wg, cout);
    return nd_launch_status("nd_conv7x7_c4_wgrad_f32 (reduce)");
}
