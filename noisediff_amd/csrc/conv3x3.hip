// conv3x3.hip -- NHWC fp32 3x3 convolution (pad 1, stride 1) as an implicit GEMM on the
// exact-fp32 matrix pipe of gfx950 (v_mfma_f32_32x32x2_f32: bitwise an fmaf chain).
//
// Replaces every nn.Conv2d(.., 3, padding=1) of NoiseDiffNet: Block.proj
// (models/archs/Diffusion_arch.py:131,136), the last-stage down/up convs (:533,:547) and the
// conv behind nn.Upsample (:74-75).  One workgroup = 4 waves = one (TH x TW pixels) x (BN couts)
// output tile of one sample:
//   * A operand (activations): the tile plus its 1-pixel halo, KC=32 channels at a time, is
//     staged global -> registers -> LDS.  The staging pass is where the *previous* layer's
//     GroupNorm + scale/shift + SiLU is applied (ND_PRO_AFFINE_*), where torch.cat is resolved
//     (two base pointers) and where nearest-x2 upsampling is resolved (index >> 1); zero padding
//     is applied after the transform, exactly like padding the activated tensor.  Each staged
//     value is reused by 9 taps x BN output channels.
//   * B operand (weights): pre-packed [tap][cin/4][coutP][4] so that the fragment a lane needs
//     for four consecutive k-steps is ONE coalesced 16-byte global load (32 lanes x 16 B = 512 B
//     contiguous); weights are shared by every workgroup, so these hit L2/MALL.  Fragments for
//     tap t+1 are fetched while tap t is multiplied (two register sets).
//   * Epilogue: +bias, store, and per-(wave, channel) {sum, M2} partials of the output for the
//     following GroupNorm (two-pass inside registers, so no E[x^2]-E[x]^2 cancellation).
// K order inside a group of 8 channels: lane half h supplies channel 4h+s at k-step s.
#include <stdlib.h>
#include "nd_common.h"

namespace {

constexpr int KC = 32;        // channels per staged chunk
constexpr int LDA = KC + 4;   // LDS floats per staged pixel (16-byte pad keeps b128 reads spread over banks)

#ifdef ND_STAMP
// diagnostic build only (tools/conv_phases.py): per-workgroup phase timestamps, 100 MHz s_memrealtime ticks
__device__ unsigned long long nd_dbg_stamps[16384 * 8];
#define ND_STAMP_AT(i) do { if (threadIdx.x == 0 && blockIdx.x < 16384) nd_dbg_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ND_STAMP_AT(i) do {} while (0)
#endif

struct ConvArgs {
    nd_conv3x3 d;
    int tiles_x, tiles_y, n_tiles, coutP, slots, total_wg;
};

// One K-chunk of the implicit GEMM with NG (compile-time) groups of 8 channels: 9 taps, weight fragments of
// tap t+1 in flight while tap t is multiplied.  Straight-line code on purpose: with any branch inside, hipcc
// falls back to s_waitcnt vmcnt(0) and every tap then waits for the *next* tap's prefetch (measured).
template <int TWH, int MB, int NB, int NG>
__device__ __forceinline__ void conv_chunk(f32x16 (&acc)[MB][NB], f32x4 (&bq)[2][NB][4], const float* As, const int (&a_off)[MB],
                                           const float* wchunk, const size_t tap_stride, const int coutP) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        if (tap < 8) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int g = 0; g < NG; ++g)
                    bq[(tap + 1) & 1][nb][g] = nd_ld4(wchunk + (tap + 1) * tap_stride + ((size_t)(2 * g) * coutP + nb * 32) * 4);
            // keep the prefetch HERE: left alone, the scheduler sinks these loads to just before their first use one tap
            // later and every tap then opens with an exposed L2 round trip (~600 cycles per 2048 cycles of MFMA)
            __builtin_amdgcn_sched_barrier(0);
        }
        const int toff = ((tap / 3) * TWH + (tap % 3)) * LDA;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            f32x4 av[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) av[mb] = nd_ld4(&As[a_off[mb] + toff + g * 8]);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[mb][nb] = nd_mfma(av[mb][k], bq[tap & 1][nb][g][k], acc[mb][nb]);
        }
    }
}

template <int TW, int MB, int NB, int MODE>
__global__ __launch_bounds__(256) void conv3x3_kernel(const ConvArgs a) {
    constexpr int WM = 2, WN = 2;
    constexpr int RB = 32 / TW;            // image rows covered by one 32-pixel M-block
    constexpr int TH = WM * MB * RB;
    constexpr int TWH = TW + 2, THH = TH + 2;
    constexpr int NPIX = TWH * THH;
    constexpr int BN = WN * NB * 32;
    constexpr int STAGE_IT = (NPIX + 31) / 32;
    // rows NPIX .. STAGE_IT*32-1 are scratch so that the staging pass needs no bounds branch
    __shared__ __attribute__((aligned(16))) float As[STAGE_IT * 32 * LDA];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, col = lane & 31;

    ND_STAMP_AT(0);
#ifdef ND_STAMP
    if (threadIdx.x == 0 && blockIdx.x < 16384) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        nd_dbg_stamps[blockIdx.x * 8 + 6] = hw; nd_dbg_stamps[blockIdx.x * 8 + 7] = xcc;
    }
    int stamp_i = 1;
#endif
    const nd_src& s = a.d.src;
    const int H = a.d.H, W = a.d.W, Cin = a.d.cin, Cout = a.d.cout;
    const int up = s.upsample ? 1 : 0;
    const int sH = H >> up, sW = W >> up;
    const int Ctot = s.c0 + s.c1;
    const int quad = tid & 7, prow = tid >> 3;      // staging role: 8 channel-quads x 32 pixel slots

    // A-fragment LDS offsets (floats) of this lane for tap (0,0), group 0
    int a_off[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int ly = (wm * MB + mb) * RB + col / TW, lx = col % TW;
        a_off[mb] = (ly * TWH + lx) * LDA + 4 * half;
    }
    const int Q = Cin >> 2;
    const size_t tap_stride = (size_t)Q * a.coutP * 4;
    f32x4 bq[2][NB][4];

    // Persistent workgroup: a CONTIGUOUS range of tiles (n-tile fastest, then x, y, sample).  Consecutive tiles of
    // a workgroup sit next to each other in the image, so their rows are the same pages (address translation
    // stays warm: the first staging of a tile that starts cold costs ~13 us against ~3 us warm, measured with
    // tools/conv_phases.py) and their halo columns are L2 hits.
    const int t_begin = (int)((long)blockIdx.x * a.total_wg / gridDim.x), t_end = (int)((long)(blockIdx.x + 1) * a.total_wg / gridDim.x);
    for (int t = t_begin; t < t_end; ++t) {
    int lid = t;
    const int nt = lid % a.n_tiles;  lid /= a.n_tiles;
    const int tx = lid % a.tiles_x;  lid /= a.tiles_x;
    const int ty = lid % a.tiles_y;
    const int b = lid / a.tiles_y;
    const int n0 = nt * BN;
    const int y0 = ty * TH - 1, x0 = tx * TW - 1;
    // B-fragment base: Wp[((tap*Q + q) * coutP + n) * 4]
    const float* wbase = a.d.weight + ((size_t)half * a.coutP + n0 + wn * NB * 32 + col) * 4;

    f32x16 acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.0f;

    for (int cb = 0; cb < Cin; cb += KC) {
        const int ng = min(4, (Cin - cb) >> 3);   // groups of 8 channels in this chunk
        const float* wchunk = wbase + (size_t)(cb >> 2) * a.coutP * 4;
        // tap-0 weight fragments first: their latency hides behind the activation loads below
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                bq[0][nb][g] = nd_ld4(wchunk + ((size_t)(2 * min(g, ng - 1)) * a.coutP + nb * 32) * 4);

        {   // ---- stage the tile + halo of 32 channels: all loads issued back to back, no branches
            const int c = cb + quad * 4;
            const bool cvalid = c < Cin;
            const int cs = cvalid ? c : 0;
            const bool second = cs >= s.c0;
            const float* base = second ? s.p1 : s.p0;
            const int ld = second ? s.ld1 : s.ld0, cc = second ? cs - s.c0 : cs;
            f32x4 tM = {0, 0, 0, 0}, tA = {1, 1, 1, 1}, tD = {0, 0, 0, 0};
            constexpr bool AFF = MODE == ND_PRO_AFFINE_SILU || MODE == ND_PRO_AFFINE_MAP_SILU;
            if (AFF) {
                const float* m = s.mad + (size_t)b * 3 * Ctot + cs;
                tM = nd_ld4(m); tA = nd_ld4(m + Ctot); tD = nd_ld4(m + 2 * Ctot);
            }
            f32x4 raw[STAGE_IT], msc[MODE == ND_PRO_AFFINE_MAP_SILU ? STAGE_IT : 1], msh[MODE == ND_PRO_AFFINE_MAP_SILU ? STAGE_IT : 1];
            unsigned okmask = 0;
#pragma unroll
            for (int it = 0; it < STAGE_IT; ++it) {
                const int p = prow + it * 32;
                const int hy = p / TWH, hx = p - hy * TWH;
                const int y = y0 + hy, x = x0 + hx;
                const bool ok = cvalid && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
                okmask |= (ok ? 1u : 0u) << it;
                const int yc = min(max(y, 0), H - 1), xc = min(max(x, 0), W - 1);
                raw[it] = nd_ld4(base + ((size_t)(b * sH + (yc >> up)) * sW + (xc >> up)) * ld + cc);
                if (MODE == ND_PRO_AFFINE_MAP_SILU) {
                    const float* mp = s.map + ((size_t)(b * H + yc) * W + xc) * (2 * Ctot) + cs;
                    msc[it] = nd_ld4(mp);
                    msh[it] = nd_ld4(mp + Ctot);
                }
            }
            __syncthreads();   // everyone is done reading the previous chunk
#pragma unroll
            for (int it = 0; it < STAGE_IT; ++it) {
                f32x4 v = raw[it];
                if (AFF) {
                    v = (v - tM) * tA + tD;
                    if (MODE == ND_PRO_AFFINE_MAP_SILU) v = v * (msc[it] + 1.0f) + msh[it];
                    v = nd_silu4(v);
                }
                if (MODE == ND_PRO_LEAKY || (MODE == ND_PRO_LEAKY_SECOND && second)) v = nd_leaky4(v);
                const f32x4 zero = {0, 0, 0, 0};
                v = ((okmask >> it) & 1u) ? v : zero;           // zero padding applies to the activated tensor
                nd_st4(&As[(prow + it * 32) * LDA + quad * 4], v);
            }
        }
        __syncthreads();
#ifdef ND_STAMP
        if (stamp_i < 5) { ND_STAMP_AT(stamp_i); ++stamp_i; }
#endif

        if (ng == 4) conv_chunk<TWH, MB, NB, 4>(acc, bq, As, a_off, wchunk, tap_stride, a.coutP);
        else if (ng == 2) conv_chunk<TWH, MB, NB, 2>(acc, bq, As, a_off, wchunk, tap_stride, a.coutP);
        else if (ng == 3) conv_chunk<TWH, MB, NB, 3>(acc, bq, As, a_off, wchunk, tap_stride, a.coutP);
        else conv_chunk<TWH, MB, NB, 1>(acc, bq, As, a_off, wchunk, tap_stride, a.coutP);
#ifdef ND_STAMP
        if (stamp_i < 5) { ND_STAMP_AT(stamp_i); ++stamp_i; }
#endif
    }

    // ------------------------------------------------------------ epilogue
    const int wrow0 = ty * TH + wm * MB * RB;                       // first image row of this wave
    const int rows_valid = max(0, min(MB * RB, H - wrow0));
    const int cols_valid = max(0, min(TW, W - tx * TW));
    const int cnt = rows_valid * cols_valid;
    const int slot = (ty * a.tiles_x + tx) * WM + wm;
    float* out = a.d.out;
    const bool interior = rows_valid == MB * RB && cols_valid == TW && n0 + BN <= Cout;   // uniform per wave

#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int n = n0 + (wn * NB + nb) * 32 + col;
        const bool nvalid = n < Cout;
        const float bias = (nvalid && a.d.bias) ? a.d.bias[n] : 0.0f;
        float s1 = 0.0f;
        if (interior) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = nd_acc_row(r, lane);
                    const int y = wrow0 + mb * RB + rr / TW, x = tx * TW + rr % TW;
                    const float v = acc[mb][nb][r] + bias;
                    acc[mb][nb][r] = v;
                    s1 += v;
                    out[((size_t)(b * H + y) * W + x) * a.d.ldo + n] = v;
                }
        } else {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = nd_acc_row(r, lane);
                    const int y = wrow0 + mb * RB + rr / TW, x = tx * TW + rr % TW;
                    const float v = acc[mb][nb][r] + bias;
                    acc[mb][nb][r] = v;
                    if (y < H && x < W) {
                        s1 += v;
                        if (nvalid) out[((size_t)(b * H + y) * W + x) * a.d.ldo + n] = v;
                    }
                }
        }
        if (a.d.stats) {
            s1 += __shfl_xor(s1, 32);
            const float mean = s1 / (float)max(cnt, 1);
            float m2 = 0.0f;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = nd_acc_row(r, lane);
                    const int y = wrow0 + mb * RB + rr / TW, x = tx * TW + rr % TW;
                    const float dv = acc[mb][nb][r] - mean;
                    m2 += (interior || (y < H && x < W)) ? dv * dv : 0.0f;
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
#ifdef ND_STAMP
    if (t == t_begin) ND_STAMP_AT(5);
#endif
    }   // tile loop
}

// ---------------------------------------------------------------- tiling choice (host)
struct Tiling { int tw, mb, nb, th, bn; };

Tiling choose_tiling(int B, int H, int W, int cout) {
    // candidates in order of preference (bigger tiles = more operand reuse); take the first that
    // still yields >= 2 workgroups per CU, otherwise the one with the most workgroups.
    const Tiling cand[4] = {{16, 2, 2, 8, 128}, {16, 2, 1, 8, 64}, {8, 1, 2, 8, 128}, {8, 1, 1, 8, 64}};
    static const int force = getenv("ND_CONV_TILING") ? atoi(getenv("ND_CONV_TILING")) : 0;   // tuning knob, e.g. 811
    if (force)
        for (int i = 0; i < 4; ++i)
            if (cand[i].tw * 100 + cand[i].mb * 10 + cand[i].nb == force && !(cand[i].tw == 16 && W < 16) && !(cand[i].bn == 128 && cout % 128)) return cand[i];
    int best = -1;
    long best_wg = -1;
    for (int i = 0; i < 4; ++i) {
        const Tiling& t = cand[i];
        if (t.tw == 16 && W < 16) continue;
        if (t.bn == 128 && cout % 128 != 0) continue;   // packed weights are padded to 64 columns
        const long wg = (long)B * nd_cdiv(H, t.th) * nd_cdiv(W, t.tw) * nd_cdiv(cout, t.bn);
        if (wg >= 512) return t;
        if (wg > best_wg) { best_wg = wg; best = i; }
    }
    return cand[best];
}

static inline int device_cus() { return nd_device_cus(); }

template <int TW, int MB, int NB, int MODE>
void launch_mode(const ConvArgs& a, hipStream_t st) {
    static std::atomic<int> per_cu_cache{0};   // resident workgroups per CU of this instantiation (same on every gfx950 device)
    int per_cu = per_cu_cache.load(std::memory_order_relaxed);
    if (!per_cu) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv3x3_kernel<TW, MB, NB, MODE>, 256, 0) != hipSuccess || per_cu <= 0) per_cu = 2;
        static const int force = getenv("ND_CONV_WG_PER_CU") ? atoi(getenv("ND_CONV_WG_PER_CU")) : 0;   // tuning knob
        if (force > 0) per_cu = force;
        per_cu_cache.store(per_cu, std::memory_order_relaxed);
    }
    const long resident = (long)device_cus() * per_cu;
    const dim3 grid((unsigned)(a.total_wg < resident ? a.total_wg : resident)), block(256);
    hipLaunchKernelGGL((conv3x3_kernel<TW, MB, NB, MODE>), grid, block, 0, st, a);
}

template <int TW, int MB, int NB>
void launch(const ConvArgs& a, hipStream_t st) {
    switch (a.d.src.mode) {
        case ND_PRO_AFFINE_SILU: launch_mode<TW, MB, NB, ND_PRO_AFFINE_SILU>(a, st); break;
        case ND_PRO_AFFINE_MAP_SILU: launch_mode<TW, MB, NB, ND_PRO_AFFINE_MAP_SILU>(a, st); break;
        case ND_PRO_LEAKY: launch_mode<TW, MB, NB, ND_PRO_LEAKY>(a, st); break;
        case ND_PRO_LEAKY_SECOND: launch_mode<TW, MB, NB, ND_PRO_LEAKY_SECOND>(a, st); break;
        default: launch_mode<TW, MB, NB, ND_PRO_NONE>(a, st);
    }
}

// OIHW -> [tap][cin/4][coutP][4], coutP = cout rounded up to 64
// dgrad: `w` is the forward layer's weight; taps flipped, channel roles swapped, read in place.
template <bool DGRAD>
__global__ void pack_conv3x3_kernel(const float* __restrict__ w, float* __restrict__ out, int cin, int cout, int coutP) {
    const size_t total = (size_t)9 * cin * coutP;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int e = i & 3;
        size_t r = i >> 2;
        const int n = r % coutP; r /= coutP;
        const int q = r % (cin >> 2);
        const int tap = r / (cin >> 2);
        const int ci = q * 4 + e;
        out[i] = n < cout ? (DGRAD ? w[((size_t)ci * cout + n) * 9 + 8 - tap] : w[((size_t)n * cin + ci) * 9 + tap]) : 0.0f;
    }
}

}  // namespace

#ifdef ND_STAMP
extern "C" int nd_dbg_read_stamps(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(nd_dbg_stamps), sizeof(unsigned long long) * n);
}
#endif

extern "C" int nd_conv3x3_stat_slots(int H, int W, int cout, int B) {
    if (H <= 0 || W <= 0 || cout <= 0 || B <= 0) return ND_E_BADARG;
    const Tiling t = choose_tiling(B, H, W, cout);
    return nd_cdiv(H, t.th) * nd_cdiv(W, t.tw) * 2;
}

extern "C" int nd_conv3x3_tiling_id(int B, int H, int W, int cout) {
    if (H <= 0 || W <= 0 || cout <= 0 || B <= 0) return ND_E_BADARG;
    const Tiling t = choose_tiling(B, H, W, cout);
    return t.tw * 100 + t.mb * 10 + t.nb;   // matches the kernel's template arguments <TW, MB, NB>
}

extern "C" int64_t nd_pack_conv3x3_weight_floats(int cin, int cout) {
    return (int64_t)9 * cin * nd_round_up(cout, 64);
}

static int pack_direct(const float* oihw, float* packed, int cin, int cout, int dgrad, void* stream) {
    ND_REQUIRE(oihw && packed, ND_E_BADARG, "nd_pack_conv3x3_weight: null pointer");
    ND_REQUIRE(cin > 0 && cout > 0 && cin % 8 == 0, ND_E_SHAPE, "nd_pack_conv3x3_weight: cin=%d must be a positive multiple of 8", cin);
    const int coutP = nd_round_up(cout, 64);
    const size_t total = (size_t)9 * cin * coutP;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    if (dgrad) hipLaunchKernelGGL(pack_conv3x3_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, oihw, packed, cin, cout, coutP);
    else hipLaunchKernelGGL(pack_conv3x3_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, oihw, packed, cin, cout, coutP);
    return nd_launch_status("nd_pack_conv3x3_weight");
}

extern "C" int nd_pack_conv3x3_weight(const float* oihw, float* packed, int cin, int cout, void* stream) {
    return pack_direct(oihw, packed, cin, cout, 0, stream);
}

extern "C" int nd_pack_conv3x3_weight_dgrad(const float* oihw_fwd, float* packed, int cin, int cout, void* stream) {
    return pack_direct(oihw_fwd, packed, cin, cout, 1, stream);
}

extern "C" int nd_conv3x3_nhwc_f32(const nd_conv3x3* d, void* stream) {
    ND_REQUIRE(d, ND_E_BADARG, "nd_conv3x3: null descriptor");
    const nd_src& s = d->src;
    ND_REQUIRE(s.p0 && d->weight && d->out, ND_E_BADARG, "nd_conv3x3: null tensor pointer");
    ND_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->cin > 0 && d->cout > 0, ND_E_BADARG, "nd_conv3x3: non-positive size");
    ND_REQUIRE(d->cin % 8 == 0, ND_E_SHAPE, "nd_conv3x3: cin=%d must be a multiple of 8", d->cin);
    ND_REQUIRE(s.c0 + s.c1 == d->cin && s.c0 % 4 == 0 && s.c1 % 4 == 0 && s.c0 > 0, ND_E_SHAPE,
               "nd_conv3x3: source channels %d+%d do not match cin=%d (multiples of 4)", s.c0, s.c1, d->cin);
    ND_REQUIRE((s.c1 == 0) == (s.p1 == nullptr), ND_E_BADARG, "nd_conv3x3: p1/c1 mismatch");
    ND_REQUIRE(s.ld0 >= s.c0 && s.ld0 % 4 == 0 && (s.c1 == 0 || (s.ld1 >= s.c1 && s.ld1 % 4 == 0)), ND_E_ALIGN,
               "nd_conv3x3: pixel strides must be >= channels and multiples of 4");
    ND_REQUIRE(nd_aligned16(s.p0) && nd_aligned16(s.p1) && nd_aligned16(d->weight) && nd_aligned16(s.mad) && nd_aligned16(s.map),
               ND_E_ALIGN, "nd_conv3x3: pointers must be 16-byte aligned");
    ND_REQUIRE(d->ldo >= d->cout, ND_E_SHAPE, "nd_conv3x3: ldo < cout");
    const bool affine = s.mode == ND_PRO_AFFINE_SILU || s.mode == ND_PRO_AFFINE_MAP_SILU;
    ND_REQUIRE(s.mode == ND_PRO_NONE || affine || s.mode == ND_PRO_LEAKY || s.mode == ND_PRO_LEAKY_SECOND, ND_E_BADARG,
               "nd_conv3x3: unsupported prologue %d", s.mode);
    ND_REQUIRE(!affine || s.mad, ND_E_BADARG, "nd_conv3x3: affine prologue needs mad");
    ND_REQUIRE(s.mode != ND_PRO_AFFINE_MAP_SILU || s.map, ND_E_BADARG, "nd_conv3x3: map prologue needs map");
    ND_REQUIRE(!s.map_blocked, ND_E_BADARG, "nd_conv3x3: the blocked map layout is read by nd_conv3x3_wino4_nhwc_f32 only");
    ND_REQUIRE(!s.upsample || (d->H % 2 == 0 && d->W % 2 == 0 && s.c1 == 0), ND_E_SHAPE, "nd_conv3x3: upsample needs even H, W and one source");
    ND_REQUIRE(!s.unshuffle, ND_E_BADARG, "nd_conv3x3: unshuffle is a pointwise-only addressing mode");
    ND_REQUIRE((d->stats == nullptr) == (d->slot_count == nullptr), ND_E_BADARG, "nd_conv3x3: stats and slot_count go together");

    const Tiling t = choose_tiling(d->B, d->H, d->W, d->cout);
    ConvArgs a;
    a.d = *d;
    a.tiles_x = nd_cdiv(d->W, t.tw);
    a.tiles_y = nd_cdiv(d->H, t.th);
    a.n_tiles = nd_cdiv(d->cout, t.bn);
    a.coutP = nd_round_up(d->cout, 64);
    a.slots = a.tiles_x * a.tiles_y * 2;
    const long wg = (long)d->B * a.tiles_x * a.tiles_y * a.n_tiles;
    ND_REQUIRE(wg < (1L << 31), ND_E_SHAPE, "nd_conv3x3: grid too large");
    a.total_wg = (int)wg;
    ND_REQUIRE(a.n_tiles * t.bn <= a.coutP, ND_E_SHAPE, "nd_conv3x3: internal tiling error");   // weight columns read
    hipStream_t st = (hipStream_t)stream;
    if (t.tw == 16 && t.nb == 2) launch<16, 2, 2>(a, st);
    else if (t.tw == 16) launch<16, 2, 1>(a, st);
    else if (t.nb == 2) launch<8, 1, 2>(a, st);
    else launch<8, 1, 1>(a, st);
    return nd_launch_status("nd_conv3x3_nhwc_f32");
}
