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
#include "nd_common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int KC4 = 16;                              // channels per K chunk
constexpr int NPOS = 36;                             // positions (xi, nu) == patch entries (a, b)
constexpr int VD_FLOATS = (KC4 / 2) * NPOS * 16 * 2; // 9216 floats = 36 KB

struct Wino4Args {
    nd_conv3x3 d;
    int tiles_x, tiles_y, n_tiles, n_cg, total_wg;
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

template <int MODE>
__global__ __launch_bounds__(256, 1) void wino4_kernel(const Wino4Args a) {
    constexpr bool AFF = MODE == ND_PRO_AFFINE_SILU;
    __shared__ __attribute__((aligned(16))) float Vd[VD_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = lane & 15, kq = lane >> 4;

    int lid = blockIdx.x;
    const int nt = lid % a.n_tiles;  lid /= a.n_tiles;
    const int tx = lid % a.tiles_x;  lid /= a.tiles_x;
    const int ty = lid % a.tiles_y;
    const int b = lid / a.tiles_y;

    const nd_src& s = a.d.src;
    const int H = a.d.H, W = a.d.W, Cin = a.d.cin, Cout = a.d.cout;
    const int Ctot = s.c0 + s.c1;
    const int y0 = ty * 16 - 1, x0 = tx * 16 - 1;
    const int cg = nt * 4 + wave;                              // this wave's 16-cout group

    f32x4 acc[NPOS];
#pragma unroll
    for (int p = 0; p < NPOS; ++p) acc[p] = f32x4{0, 0, 0, 0};

    const int n_c8 = (Cin + 7) >> 3;
    for (int cb = 0; cb < Cin; cb += KC4) {
        __syncthreads();                                       // previous chunk consumed
        // ---- stage: 18x18 halo pixels x 4 channel quads, each written to every (tile, patch entry) it belongs to
        for (int i = tid; i < 324 * 4; i += 256) {
            const int p = i >> 2, q = i & 3;
            const int hy = p / 18, hx = p - hy * 18;
            const int gy = y0 + hy, gx = x0 + hx;
            const int c = cb + 4 * q;
            f32x4 v = {0, 0, 0, 0};
            const bool inside = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W && c < Cin;
            if (inside) {
                const size_t pix = ((size_t)b * H + gy) * W + gx;
                const bool sec = c >= s.c0;
                v = sec ? nd_ld4(s.p1 + pix * s.ld1 + (c - s.c0)) : nd_ld4(s.p0 + pix * s.ld0 + c);
                if (AFF) {
                    const float* m = s.mad + (size_t)b * 3 * Ctot + c;
                    v = nd_silu4((v - nd_ld4(m)) * nd_ld4(m + Ctot) + nd_ld4(m + 2 * Ctot));
                }
            }
            const int kp = (q >> 1) * 4 + (q & 1) * 2;         // channel pair index of (v.x, v.y); (v.z, v.w) is kp + 1
#pragma unroll
            for (int ey = 0; ey < 2; ++ey) {
                const int tyi = (hy >> 2) - ey, ay = hy - 4 * tyi;
                if (tyi < 0 || tyi > 3 || ay > 5) continue;
#pragma unroll
                for (int ex = 0; ex < 2; ++ex) {
                    const int txi = (hx >> 2) - ex, ax = hx - 4 * txi;
                    if (txi < 0 || txi > 3 || ax > 5) continue;
                    const int slot = (ay * 6 + ax) * 16 + tyi * 4 + txi;
                    *reinterpret_cast<f32x2*>(&Vd[((kp * NPOS * 16) + slot) * 2]) = f32x2{v.x, v.y};
                    *reinterpret_cast<f32x2*>(&Vd[(((kp + 1) * NPOS * 16) + slot) * 2]) = f32x2{v.z, v.w};
                }
            }
        }
        __syncthreads();

#pragma unroll 1
        for (int g2 = 0; g2 < 2; ++g2) {                       // 8 channels: two MFMA k groups, packed side by side
            const int c8 = (cb >> 3) + g2;
            if (c8 >= n_c8) break;
            const f32x2* dsrc = reinterpret_cast<const f32x2*>(Vd) + ((g2 * 4 + kq) * NPOS) * 16 + tile;
            f32x2 V[6][6];
            {   // B^T d B: rows (over a) for every column b, then columns
                f32x2 T[6][6];
#pragma unroll
                for (int bx = 0; bx < 6; ++bx) {
                    f32x2 col[6], t[6];
#pragma unroll
                    for (int ay = 0; ay < 6; ++ay) col[ay] = dsrc[(ay * 6 + bx) * 16];
                    w4_bt(col, t);
#pragma unroll
                    for (int xi = 0; xi < 6; ++xi) T[xi][bx] = t[xi];
                }
#pragma unroll
                for (int xi = 0; xi < 6; ++xi) w4_bt(T[xi], V[xi]);
            }
            const f32x4* wsrc = reinterpret_cast<const f32x4*>(a.d.weight) + ((size_t)(c8 * a.n_cg + cg) * 18) * 64 + lane;
#pragma unroll
            for (int pp = 0; pp < 18; ++pp) {
                const f32x4 u = wsrc[pp * 64];
                const int p0 = 2 * pp, p1 = 2 * pp + 1;
                acc[p0] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.x, V[p0 / 6][p0 % 6].x, acc[p0], 0, 0, 0);
                acc[p0] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.y, V[p0 / 6][p0 % 6].y, acc[p0], 0, 0, 0);
                acc[p1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.z, V[p1 / 6][p1 % 6].x, acc[p1], 0, 0, 0);
                acc[p1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.w, V[p1 / 6][p1 % 6].y, acc[p1], 0, 0, 0);
            }
        }
    }

    // ---- output transform Y = A^T M A on float4s (couts co .. co+3 of tile `tile`), bias, 16-byte stores
    const int co = cg * 16 + 4 * kq;
    f32x4 bias4 = {0, 0, 0, 0};
    if (a.d.bias && co + 3 < Cout) bias4 = nd_ld4(a.d.bias + co);
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
    const int py0 = ty * 16 + 4 * (tile >> 2), px0 = tx * 16 + 4 * (tile & 3);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f32x4 y[4];
        w4_at(Z[i], y);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int py = py0 + i, px = px0 + j;
            if (py < H && px < W && co + 3 < Cout)
                nd_st4(a.d.out + (((size_t)b * H + py) * W + px) * a.d.ldo + co, y[j] + bias4);
        }
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

template <int MODE>
int launch4(const Wino4Args& a, hipStream_t st) {
    hipLaunchKernelGGL((wino4_kernel<MODE>), dim3((unsigned)a.total_wg), dim3(256), 0, st, a);
    return 0;
}

}  // namespace

extern "C" int64_t nd_pack_conv3x3_wino4_weight_floats(int cin, int cout) {
    return (int64_t)nd_cdiv(cin, 8) * nd_cdiv(nd_round_up(cout, 64), 16) * 18 * 256;
}

extern "C" int nd_pack_conv3x3_wino4_weight(const float* oihw, float* packed, int cin, int cout, void* stream) {
    ND_REQUIRE(oihw && packed, ND_E_BADARG, "nd_pack_conv3x3_wino4_weight: null pointer");
    ND_REQUIRE(cin > 0 && cout > 0, ND_E_BADARG, "nd_pack_conv3x3_wino4_weight: non-positive size");
    const int n_c8 = nd_cdiv(cin, 8), n_cg = nd_round_up(cout, 64) / 16;
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
    ND_REQUIRE(d->cin % 4 == 0 && d->cout % 4 == 0, ND_E_SHAPE, "nd_conv3x3_wino4: cin=%d and cout=%d must be multiples of 4", d->cin, d->cout);
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
    ND_REQUIRE(!d->stats && !d->slot_count, ND_E_BADARG, "nd_conv3x3_wino4: GroupNorm statistics are not produced by this kernel yet");

    Wino4Args a;
    a.d = *d;
    a.tiles_x = nd_cdiv(d->W, 16);
    a.tiles_y = nd_cdiv(d->H, 16);
    a.n_tiles = nd_cdiv(d->cout, 64);
    a.n_cg = nd_round_up(d->cout, 64) / 16;
    const long wg = (long)d->B * a.tiles_x * a.tiles_y * a.n_tiles;
    ND_REQUIRE(wg < (1L << 31), ND_E_SHAPE, "nd_conv3x3_wino4: grid too large");
    a.total_wg = (int)wg;
    hipStream_t st = (hipStream_t)stream;
    if (s.mode == ND_PRO_AFFINE_SILU) launch4<ND_PRO_AFFINE_SILU>(a, st);
    else launch4<ND_PRO_NONE>(a, st);
    return nd_launch_status("nd_conv3x3_wino4_nhwc_f32");
}
