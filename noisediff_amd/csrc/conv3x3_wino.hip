// conv3x3_wino.hip -- the same 3x3 convolution (pad 1, stride 1, NHWC fp32) as conv3x3.hip, computed with the
// Winograd minimal-filtering algorithm F(2x2, 3x3) on the exact-fp32 matrix pipe: 16 multiplies per 2x2 output
// block and (cin, cout) pair instead of 36, i.e. 2.25x fewer MFMA instructions for the same result
// (Y = A^T [ (G g G^T) .* (B^T d B) ] A; fp32 transforms with +-1 / +-1/2 coefficients: error ~1e-6 relative, far
// inside the 1e-3 budget and pinned by the same parity tests as the direct kernel).
//
// Replaces Block.proj and the up/down/last-stage convs (models/archs/Diffusion_arch.py:131,136,533,547,75) wherever
// the image is at least one 16x16 tile; conv3x3.hip stays the path for smaller images.
//
// One workgroup = 4 waves = a 16x16-pixel tile (8x8 blocks of 2x2 outputs) x BN output channels of one sample:
//   * the 18x18 input halo tile, 32 channels at a time, is staged global -> registers -> LDS with the same fused
//     prologue as the direct kernel (previous GroupNorm + scale/shift + SiLU, virtual concat, nearest-x2 addressing,
//     zero padding after the activation).  LDS holds RAW pixels, split into even-x / odd-x planes so that the
//     stride-2 block reads of a 16-lane group fall on 16 different 16-byte slots;
//   * the input transform B^T d B is separable and done on the fly in registers: per position row xi and channel
//     group, 8 LDS reads give the 4 column terms T_c, each of the 4 positions nu is one more add -> MFMA A operand;
//   * weights are pre-transformed at pack time to U = G g G^T, laid out [16 positions][cin/4][coutP][4] so a lane's
//     B fragment for 4 k-steps is one coalesced 16-byte load (prefetched one step ahead);
//   * the output transform is linear, so each position's K-chunk partial product (4 accumulators in flight) is
//     folded straight into the four Y accumulators with +-1 coefficients -- no 16-position accumulator file;
//   * epilogue identical to the direct kernel: +bias, store, per-(wave, channel) GroupNorm partials {sum, M2}.
#include <stdlib.h>
#include "nd_common.h"

namespace {

// KC = channels per staged chunk (template parameter: 32 or 16); LDA = KC + 4 floats per staged pixel
// (9 or 5 sixteen-byte slots: odd, so consecutive pixels of a plane row rotate through the banks)
constexpr int HT = 18;              // halo tile side (16 + 2)
constexpr int PW = 12;              // pixels per plane row: 9 used, padded so that 2 plane rows = 8 (mod 16) slots
constexpr int NPIX = HT * HT;       // 324
constexpr int PLANE = HT * PW;      // pixel slots per plane
constexpr int WBLOCK = 16 * 8 * 64 * 4;   // floats per packed weight block (one K chunk of 32 x 64 output channels)

struct WinoArgs {
    nd_conv3x3 d;
    int tiles_x, tiles_y, n_tiles, coutP, slots, total_wg;
};

// B^T rows as (index, sign) pairs: xi -> s1*d[r1] + s2*d[r2]
__device__ __forceinline__ constexpr int bt_r1(int xi) { return xi == 0 ? 0 : 1; }
__device__ __forceinline__ constexpr int bt_r2(int xi) { return xi == 3 ? 3 : 2; }
__device__ __forceinline__ constexpr float bt_s1(int xi) { return xi == 2 ? -1.0f : 1.0f; }
__device__ __forceinline__ constexpr float bt_s2(int xi) { return (xi == 0 || xi == 3) ? -1.0f : 1.0f; }
// A^T[a][xi]
__device__ __forceinline__ constexpr int at_coef(int a, int xi) {
    return a == 0 ? (xi <= 2 ? 1 : 0) : (xi == 0 ? 0 : (xi == 1 ? 1 : -1));
}

template <int NB, int NG, int LDA>
__device__ __forceinline__ void wino_chunk(f32x16 (&Y)[2][2][NB], const float* As, const int a_base, const float* wchunk) {
    static_assert(NB == 1, "the packed weight block is 64 output channels wide");
    // step = ((xi * 2 + np) * NG + g): positions (xi, nu = 2*np, 2*np+1), channel group g.  Two position accumulators
    // in flight keep the register budget at 3 waves per SIMD; the price is that the column terms T_c are rebuilt for
    // each nu pair (1 LDS b128 read per MFMA -- still only ~25 % of the LDS rate at the fp32 MFMA issue rate).
    f32x4 bw[2][2][NB];     // [buffer][nu & 1][nb]
    auto load_b = [&](int buf, int step) {
        const int g = step % NG, pp = step / NG, xi = pp >> 1, np = pp & 1;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                bw[buf][j][nb] = nd_ld4(wchunk + ((xi * 4 + np * 2 + j) * 8 + 2 * g) * 256);
    };
    load_b(0, 0);
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
#pragma unroll
        for (int np = 0; np < 2; ++np) {
            f32x16 M[2][NB];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) M[j][nb][r] = 0.0f;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int step = (xi * 2 + np) * NG + g;
                if (step + 1 < 8 * NG) {
                    load_b((step + 1) & 1, step + 1);
                    __builtin_amdgcn_sched_barrier(0);      // keep the prefetch one step ahead (see conv3x3.hip)
                }
                // column terms T_c = s1 * d[r1][c] + s2 * d[r2][c], c = 0..3 (plane = c & 1, plane column += c >> 1)
                f32x4 T[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int off = ((c & 1) * PLANE + (c >> 1)) * LDA + g * 8;
                    const f32x4 d1 = nd_ld4(&As[a_base + off + bt_r1(xi) * PW * LDA]);
                    const f32x4 d2 = nd_ld4(&As[a_base + off + bt_r2(xi) * PW * LDA]);
                    T[c] = bt_s1(xi) * d1 + bt_s2(xi) * d2;
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int nu = np * 2 + j;
#if defined(ND_WINO_ABLATE) && ND_WINO_ABLATE == 2      // diagnostic: no column transform
                    const f32x4 V = T[nu];
#else
                    const f32x4 V = bt_s1(nu) * T[bt_r1(nu)] + bt_s2(nu) * T[bt_r2(nu)];
#endif
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            M[j][nb] = nd_mfma(V[k], bw[step & 1][j][nb][k], M[j][nb]);
                }
            }
            // fold into the 2x2 outputs: Y[a][b] += A^T[a][xi] * A^T[b][nu] * M[nu]
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
#if defined(ND_WINO_ABLATE) && ND_WINO_ABLATE == 1      // diagnostic: one fold per position instead of ~2.25
                        const int cf = (a == (xi & 1) && b == (j & 1)) ? 1 : 0;
#else
                        const int cf = at_coef(a, xi) * at_coef(b, np * 2 + j);
#endif
                        if (cf != 0) {
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb)
                                Y[a][b][nb] = cf > 0 ? Y[a][b][nb] + M[j][nb] : Y[a][b][nb] - M[j][nb];
                        }
                    }
        }
    }
}

template <int NB, int MODE, int KC>
__global__ __launch_bounds__(256, KC == 16 ? 3 : 2) void wino_kernel(const WinoArgs a) {
    constexpr int LDA = KC + 4, QPP = KC / 4;               // quads per pixel
    constexpr int STAGE_IT = (NPIX * QPP + 255) / 256;
    constexpr int BN = 64 * NB;
    constexpr bool MAP = MODE == ND_PRO_AFFINE_MAP_SILU;
    constexpr bool AFF = MODE == ND_PRO_AFFINE_SILU || MAP;
    __shared__ __attribute__((aligned(16))) float As[(2 * PLANE + 1) * LDA];   // +1: scratch pixel for the staging tail

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, col = lane & 31;

    const nd_src& s = a.d.src;
    const int H = a.d.H, W = a.d.W, Cin = a.d.cin, Cout = a.d.cout;
    const int up = s.upsample ? 1 : 0;
    const int sH = H >> up, sW = W >> up;
    const int Ctot = s.c0 + s.c1;
    const int quad = tid % QPP;

    // lane (block row 4*wm + col/8, block column col%8): halo-tile pixel (2*by, 2*bx) -> plane 0, plane column bx
    const int by = 4 * wm + (col >> 3), bx = col & 7;
    const int a_base = ((2 * by) * PW + bx) * LDA + 4 * half;

    const int t_begin = (int)((long)blockIdx.x * a.total_wg / gridDim.x), t_end = (int)((long)(blockIdx.x + 1) * a.total_wg / gridDim.x);
    for (int t = t_begin; t < t_end; ++t) {
        int lid = t;
        const int nt = lid % a.n_tiles;  lid /= a.n_tiles;
        const int tx = lid % a.tiles_x;  lid /= a.tiles_x;
        const int ty = lid % a.tiles_y;
        const int b = lid / a.tiles_y;
        const int n0 = nt * BN;
        const int y0 = ty * 16 - 1, x0 = tx * 16 - 1;
        const float* wbase = a.d.weight + (size_t)nt * WBLOCK + (half * 64 + wn * 32 + col) * 4;   // this lane's column of block (chunk 0, nt)

        f32x16 Y[2][2][NB];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) Y[i][j][nb][r] = 0.0f;

        for (int cb = 0; cb < Cin; cb += KC) {
            const int ng = min(KC / 8, (Cin - cb) >> 3);
            {   // ---- stage the 18x18 halo tile of 32 channels (all loads in flight, no branches)
                const int c = cb + quad * 4;
                const bool cvalid = c < Cin;
                const int cs = cvalid ? c : 0;
                const bool second = cs >= s.c0;
                const float* base = second ? s.p1 : s.p0;
                const int ld = second ? s.ld1 : s.ld0, cc = second ? cs - s.c0 : cs;
                f32x4 tM = {0, 0, 0, 0}, tA = {1, 1, 1, 1}, tD = {0, 0, 0, 0};
                if (AFF) {
                    const float* m = s.mad + (size_t)b * 3 * Ctot + cs;
                    tM = nd_ld4(m); tA = nd_ld4(m + Ctot); tD = nd_ld4(m + 2 * Ctot);
                }
                f32x4 raw[STAGE_IT], msc[MAP ? STAGE_IT : 1], msh[MAP ? STAGE_IT : 1];
                unsigned okmask = 0;
#pragma unroll
                for (int it = 0; it < STAGE_IT; ++it) {
                    const int p = tid / QPP + it * (256 / QPP);
                    const int hy = p / HT, hx = p - hy * HT;
                    const int y = y0 + hy, x = x0 + hx;
                    const bool ok = cvalid && p < NPIX && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
                    okmask |= (ok ? 1u : 0u) << it;
                    const int yc = min(max(y, 0), H - 1), xc = min(max(x, 0), W - 1);
                    raw[it] = nd_ld4(base + ((size_t)(b * sH + (yc >> up)) * sW + (xc >> up)) * ld + cc);
                    if (MAP) {
                        const float* mp = s.map + ((size_t)(b * H + yc) * W + xc) * (2 * Ctot) + cs;
                        msc[it] = nd_ld4(mp);
                        msh[it] = nd_ld4(mp + Ctot);
                    }
                }
                __syncthreads();   // everyone is done reading the previous chunk / tile
#pragma unroll
                for (int it = 0; it < STAGE_IT; ++it) {
                    const int p = tid / QPP + it * (256 / QPP);
                    const int hy = p / HT, hx = p - hy * HT;
                    f32x4 v = raw[it];
                    if (AFF) {
                        v = (v - tM) * tA + tD;
                        if (MAP) v = v * (msc[it] + 1.0f) + msh[it];
                        v = nd_silu4(v);
                    }
                    if (MODE == ND_PRO_LEAKY || (MODE == ND_PRO_LEAKY_SECOND && second)) v = nd_leaky4(v);
                    const f32x4 zero = {0, 0, 0, 0};
                    v = ((okmask >> it) & 1u) ? v : zero;
                    const int slot = p < NPIX ? ((hx & 1) * PLANE + hy * PW + (hx >> 1)) : 2 * PLANE;
                    nd_st4(&As[slot * LDA + quad * 4], v);
                }
            }
            __syncthreads();

            const float* wchunk = wbase + (size_t)(cb >> 5) * a.n_tiles * WBLOCK + ((cb & 31) >> 2) * 256;
            if (KC == 32 && ng == 4) wino_chunk<NB, 4, LDA>(Y, As, a_base, wchunk);
            else if (ng == 2) wino_chunk<NB, 2, LDA>(Y, As, a_base, wchunk);
            else if (KC == 32 && ng == 3) wino_chunk<NB, 3, LDA>(Y, As, a_base, wchunk);
            else wino_chunk<NB, 1, LDA>(Y, As, a_base, wchunk);
        }

        // ------------------------------------------------------------ epilogue
        // accumulator register r of lane (col, half) is block i = acc_row(r): block row 4*wm + i/8, block column i%8
        const int wrow0 = ty * 16 + wm * 8;                     // first image row of this wave (8 pixel rows x 16 columns)
        const int rows_valid = max(0, min(8, H - wrow0));
        const int cols_valid = max(0, min(16, W - tx * 16));
        const int cnt = rows_valid * cols_valid;
        const int slot = (ty * a.tiles_x + tx) * 2 + wm;
        float* out = a.d.out;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int n = n0 + (wn * NB + nb) * 32 + col;
            const bool nvalid = n < Cout;
            const float bias = (nvalid && a.d.bias) ? a.d.bias[n] : 0.0f;
            float s1 = 0.0f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int blk = nd_acc_row(r, lane);
                        const int y = wrow0 + 2 * (blk >> 3) + i, x = tx * 16 + 2 * (blk & 7) + j;
                        const float v = Y[i][j][nb][r] + bias;
                        Y[i][j][nb][r] = v;
                        if (y < H && x < W) {
                            s1 += v;
                            if (nvalid) out[((size_t)(b * H + y) * W + x) * a.d.ldo + n] = v;
                        }
                    }
            if (a.d.stats) {
                s1 += __shfl_xor(s1, 32);
                const float mean = s1 / (float)max(cnt, 1);
                float m2 = 0.0f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int blk = nd_acc_row(r, lane);
                            const int y = wrow0 + 2 * (blk >> 3) + i, x = tx * 16 + 2 * (blk & 7) + j;
                            const float dv = Y[i][j][nb][r] - mean;
                            m2 += (y < H && x < W) ? dv * dv : 0.0f;
                        }
                m2 += __shfl_xor(m2, 32);
                if (half == 0 && nvalid) {
                    float* st = a.d.stats + (((size_t)b * a.slots + slot) * Cout + n) * 2;
                    st[0] = s1;
                    st[1] = m2;
                }
            }
        }
        if (a.d.slot_count && b == 0 && nt == 0 && wn == 0 && lane == 0) a.d.slot_count[slot] = (float)cnt;
    }
}

// OIHW -> U = G g G^T, packed as 128 KB blocks [cinP/32][coutP/64] of [pos = xi*4 + nu][8 channel quads][64 n][4]:
// everything one workgroup reads for one K chunk is contiguous and every fragment address is block base + a
// compile-time offset.  coutP = cout rounded up to 64, cinP = cin rounded up to 32 (zero fill).
// dgrad: `w` is the forward layer's weight (cout x cin there = cin x cout here); taps flipped, channel roles swapped, read in place.
template <bool DGRAD>
__global__ void pack_wino_kernel(const float* __restrict__ w, float* __restrict__ out, int cin, int cinP, int cout, int coutP) {
    const float G[4][3] = {{1.0f, 0.0f, 0.0f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.0f, 0.0f, 1.0f}};
    const size_t total = (size_t)16 * cinP * coutP;
    const int n_tiles = coutP >> 6;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int e = i & 3, n64 = (i >> 2) & 63, q = (i >> 8) & 7, pos = (i >> 11) & 15;
        const size_t blk = i >> 15;
        const int n = (int)(blk % n_tiles) * 64 + n64, ci = (int)(blk / n_tiles) * 32 + q * 4 + e;
        const int xi = pos >> 2, nu = pos & 3;
        float u = 0.0f;
        if (n < cout && ci < cin) {
            const float* gp = w + (DGRAD ? (size_t)ci * cout + n : (size_t)n * cin + ci) * 9;
            float g[9];
#pragma unroll
            for (int j = 0; j < 9; ++j) g[j] = gp[DGRAD ? 8 - j : j];
            float tmp[3];                                   // row xi of G g
#pragma unroll
            for (int j = 0; j < 3; ++j) tmp[j] = G[xi][0] * g[j] + G[xi][1] * g[3 + j] + G[xi][2] * g[6 + j];
            u = tmp[0] * G[nu][0] + tmp[1] * G[nu][1] + tmp[2] * G[nu][2];
        }
        out[i] = u;
    }
}

static inline int device_cus() { return nd_device_cus(); }

template <int NB, int MODE, int KC>
void launch_kc(const WinoArgs& a, hipStream_t st) {
    static std::atomic<int> per_cu_cache{0};
    int per_cu = per_cu_cache.load(std::memory_order_relaxed);
    if (!per_cu) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, wino_kernel<NB, MODE, KC>, 256, 0) != hipSuccess || per_cu <= 0) per_cu = 2;
        per_cu_cache.store(per_cu, std::memory_order_relaxed);
    }
    const long resident = (long)device_cus() * per_cu;
    const dim3 grid((unsigned)(a.total_wg < resident ? a.total_wg : resident)), block(256);
    hipLaunchKernelGGL((wino_kernel<NB, MODE, KC>), grid, block, 0, st, a);
}

template <int NB, int MODE>
void launch_mode(const WinoArgs& a, hipStream_t st) {
    static const int kc = getenv("ND_WINO_KC") ? atoi(getenv("ND_WINO_KC")) : 32;   // tuning knob
    if (kc == 16) launch_kc<NB, MODE, 16>(a, st);
    else launch_kc<NB, MODE, 32>(a, st);
}

template <int NB>
void launch(const WinoArgs& a, hipStream_t st) {
    switch (a.d.src.mode) {
        case ND_PRO_AFFINE_SILU: launch_mode<NB, ND_PRO_AFFINE_SILU>(a, st); break;
        case ND_PRO_AFFINE_MAP_SILU: launch_mode<NB, ND_PRO_AFFINE_MAP_SILU>(a, st); break;
        case ND_PRO_LEAKY: launch_mode<NB, ND_PRO_LEAKY>(a, st); break;
        case ND_PRO_LEAKY_SECOND: launch_mode<NB, ND_PRO_LEAKY_SECOND>(a, st); break;
        default: launch_mode<NB, ND_PRO_NONE>(a, st);
    }
}

}  // namespace

extern "C" int nd_conv3x3_wino_stat_slots(int H, int W) {
    if (H <= 0 || W <= 0) return ND_E_BADARG;
    return nd_cdiv(H, 16) * nd_cdiv(W, 16) * 2;
}

extern "C" int64_t nd_pack_conv3x3_wino_weight_floats(int cin, int cout) {
    return (int64_t)16 * nd_round_up(cin, 32) * nd_round_up(cout, 64);
}

static int pack_wino(const float* oihw, float* packed, int cin, int cout, int dgrad, void* stream) {
    ND_REQUIRE(oihw && packed, ND_E_BADARG, "nd_pack_conv3x3_wino_weight: null pointer");
    ND_REQUIRE(cin > 0 && cout > 0 && cin % 8 == 0, ND_E_SHAPE, "nd_pack_conv3x3_wino_weight: cin=%d must be a positive multiple of 8", cin);
    const int coutP = nd_round_up(cout, 64), cinP = nd_round_up(cin, 32);
    const size_t total = (size_t)16 * cinP * coutP;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    if (dgrad) hipLaunchKernelGGL(pack_wino_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, oihw, packed, cin, cinP, cout, coutP);
    else hipLaunchKernelGGL(pack_wino_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, oihw, packed, cin, cinP, cout, coutP);
    return nd_launch_status("nd_pack_conv3x3_wino_weight");
}

extern "C" int nd_pack_conv3x3_wino_weight(const float* oihw, float* packed, int cin, int cout, void* stream) {
    return pack_wino(oihw, packed, cin, cout, 0, stream);
}

extern "C" int nd_pack_conv3x3_wino_weight_dgrad(const float* oihw_fwd, float* packed, int cin, int cout, void* stream) {
    return pack_wino(oihw_fwd, packed, cin, cout, 1, stream);
}

extern "C" int nd_conv3x3_wino_nhwc_f32(const nd_conv3x3* d, void* stream) {
    ND_REQUIRE(d, ND_E_BADARG, "nd_conv3x3_wino: null descriptor");
    const nd_src& s = d->src;
    ND_REQUIRE(s.p0 && d->weight && d->out, ND_E_BADARG, "nd_conv3x3_wino: null tensor pointer");
    ND_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->cin > 0 && d->cout > 0, ND_E_BADARG, "nd_conv3x3_wino: non-positive size");
    ND_REQUIRE(d->cin % 8 == 0, ND_E_SHAPE, "nd_conv3x3_wino: cin=%d must be a multiple of 8", d->cin);
    ND_REQUIRE(s.c0 + s.c1 == d->cin && s.c0 % 4 == 0 && s.c1 % 4 == 0 && s.c0 > 0, ND_E_SHAPE,
               "nd_conv3x3_wino: source channels %d+%d do not match cin=%d (multiples of 4)", s.c0, s.c1, d->cin);
    ND_REQUIRE((s.c1 == 0) == (s.p1 == nullptr), ND_E_BADARG, "nd_conv3x3_wino: p1/c1 mismatch");
    ND_REQUIRE(s.ld0 >= s.c0 && s.ld0 % 4 == 0 && (s.c1 == 0 || (s.ld1 >= s.c1 && s.ld1 % 4 == 0)), ND_E_ALIGN,
               "nd_conv3x3_wino: pixel strides must be >= channels and multiples of 4");
    ND_REQUIRE(nd_aligned16(s.p0) && nd_aligned16(s.p1) && nd_aligned16(d->weight) && nd_aligned16(s.mad) && nd_aligned16(s.map),
               ND_E_ALIGN, "nd_conv3x3_wino: pointers must be 16-byte aligned");
    ND_REQUIRE(d->ldo >= d->cout, ND_E_SHAPE, "nd_conv3x3_wino: ldo < cout");
    const bool affine = s.mode == ND_PRO_AFFINE_SILU || s.mode == ND_PRO_AFFINE_MAP_SILU;
    ND_REQUIRE(s.mode == ND_PRO_NONE || affine || s.mode == ND_PRO_LEAKY || s.mode == ND_PRO_LEAKY_SECOND, ND_E_BADARG,
               "nd_conv3x3_wino: unsupported prologue %d", s.mode);
    ND_REQUIRE(!affine || s.mad, ND_E_BADARG, "nd_conv3x3_wino: affine prologue needs mad");
    ND_REQUIRE(s.mode != ND_PRO_AFFINE_MAP_SILU || s.map, ND_E_BADARG, "nd_conv3x3_wino: map prologue needs map");
    ND_REQUIRE(!s.map_blocked, ND_E_BADARG, "nd_conv3x3_wino: the blocked map layout is read by nd_conv3x3_wino4_nhwc_f32 only");
    ND_REQUIRE(!s.upsample || (d->H % 2 == 0 && d->W % 2 == 0 && s.c1 == 0), ND_E_SHAPE, "nd_conv3x3_wino: upsample needs even H, W and one source");
    ND_REQUIRE(!s.unshuffle, ND_E_BADARG, "nd_conv3x3_wino: unshuffle is a pointwise-only addressing mode");
    ND_REQUIRE((d->stats == nullptr) == (d->slot_count == nullptr), ND_E_BADARG, "nd_conv3x3_wino: stats and slot_count go together");

    WinoArgs a;
    a.d = *d;
    a.tiles_x = nd_cdiv(d->W, 16);
    a.tiles_y = nd_cdiv(d->H, 16);
    a.coutP = nd_round_up(d->cout, 64);
    // NB = 2 (128 output channels per workgroup) needs more than the 256 architectural VGPRs hipcc will use for
    // MFMA-in-VGPR code and spills; one 64-channel N-tile per workgroup it is.
    const int nb = 1;
    a.n_tiles = nd_cdiv(d->cout, 64 * nb);
    a.slots = a.tiles_x * a.tiles_y * 2;
    const long wg = (long)d->B * a.tiles_x * a.tiles_y * a.n_tiles;
    ND_REQUIRE(wg < (1L << 31), ND_E_SHAPE, "nd_conv3x3_wino: grid too large");
    a.total_wg = (int)wg;
    hipStream_t st = (hipStream_t)stream;
    launch<1>(a, st);
    return nd_launch_status("nd_conv3x3_wino_nhwc_f32");
}
