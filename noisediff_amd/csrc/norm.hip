// norm.hip -- GroupNorm statistics plumbing, the fused ResnetBlock tail, RMSNorm.
#include "nd_common.h"

namespace {

// One workgroup (4 waves) per (sample, group): pool the conv epilogue's per-(slot, channel) {sum, M2} partials
// in fp64 (see below), then fold gamma/beta and the
// time-embedding scale/shift so that  GN(x)*(scale+1)+shift == (x - M)*A + D
// (Block.forward, models/archs/Diffusion_arch.py:137-141; nn.GroupNorm: biased variance, eps inside sqrt).
// Per item i (one wave-slot of one channel): n_i pixels, sum_i, M2_i (centred second moment).  Pooled over
// the group:  N = sum n_i,  S = sum sum_i,  Q = sum (M2_i + sum_i^2 / n_i)   =>  mean = S/N,  var = Q/N - mean^2.
// The per-item M2 was centred in fp32 registers; the pooling runs in fp64 (53 bits against fp32 inputs), so the
// final subtraction loses nothing unless mean^2/var exceeds ~1e8.  Plain sums => independent loads, no serial
// Welford chain, fixed reduction tree => bitwise reproducible.
#ifndef GNF_WAVES
#define GNF_WAVES 4
#endif
constexpr int GNF_THREADS = 64 * GNF_WAVES;
__global__ __launch_bounds__(GNF_THREADS) void gn_finalize_kernel(const float* __restrict__ stats, const float* __restrict__ slot_count,
                                                          int slots, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ ss, int ld_ss, float* __restrict__ mad,
                                                          int C, int G, float eps, float* __restrict__ mean_rstd = nullptr) {
    __shared__ double part[GNF_WAVES][3];
    const int b = blockIdx.x / G, g = blockIdx.x % G;
    const int cpg = C / G, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double N = 0.0, S = 0.0, Q = 0.0;
    const int items = slots * cpg;
    const float2* st2 = reinterpret_cast<const float2*>(stats) + (size_t)b * slots * C + g * cpg;
    for (int i0 = tid; i0 < items; i0 += GNF_THREADS * 8) {   // 8 independent loads in flight per thread, no branches
        float ni[8];
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = min(i0 + u * GNF_THREADS, items - 1);
            const int slot = i / cpg, j = i - slot * cpg;
            ni[u] = (i0 + u * GNF_THREADS < items) ? slot_count[slot] : 0.0f;
            v[u] = st2[(size_t)slot * C + j];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool ok = ni[u] > 0.0f;
            const double n = ok ? (double)ni[u] : 1.0;
            N += ok ? n : 0.0;
            S += ok ? (double)v[u].x : 0.0;
            Q += ok ? (double)v[u].y + (double)v[u].x * (double)v[u].x / n : 0.0;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        N += __shfl_xor(N, o); S += __shfl_xor(S, o); Q += __shfl_xor(Q, o);
    }
    if (lane == 0) { part[wave][0] = N; part[wave][1] = S; part[wave][2] = Q; }
    __syncthreads();
    double tot[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {                           // fixed pairwise tree over the waves: bitwise reproducible
        double t[GNF_WAVES];
#pragma unroll
        for (int w = 0; w < GNF_WAVES; ++w) t[w] = part[w][q];
#pragma unroll
        for (int n = GNF_WAVES / 2; n > 0; n >>= 1)
#pragma unroll
            for (int w = 0; w < n; ++w) t[w] = t[2 * w] + t[2 * w + 1];
        tot[q] = t[0];
    }
    N = tot[0]; S = tot[1]; Q = tot[2];
    const double mean = N > 0.0 ? S / N : 0.0;
    double var = N > 0.0 ? Q / N - mean * mean : 0.0;
    var = var > 0.0 ? var : 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float fmean = (float)mean;
    if (mean_rstd && tid == 0) {                                // training: the group's moments, saved for the backward pass
        mean_rstd[((size_t)b * G + g) * 2] = fmean;
        mean_rstd[((size_t)b * G + g) * 2 + 1] = rstd;
    }
    for (int i = tid; i < cpg; i += GNF_THREADS) {
        const int ch = g * cpg + i;
        const float sc = ss ? ss[(size_t)b * ld_ss + ch] : 0.0f;
        const float sh = ss ? ss[(size_t)b * ld_ss + C + ch] : 0.0f;
        float* o = mad + (size_t)b * 3 * C + ch;
        o[0] = fmean;
        o[C] = rstd * gamma[ch] * (sc + 1.0f);
        o[2 * C] = beta[ch] * (sc + 1.0f) + sh;
    }
}

// out = silu((t - M)*A + D) + res0 + res1   (ResnetBlock.forward tail, Diffusion_arch.py:168-170,
// when res_conv is nn.Identity; res1 carries `shot_emb + r`, :603).  Pure HBM streaming.
__global__ __launch_bounds__(256) void affine_silu_add_kernel(const float* __restrict__ t, int ldt, const float* __restrict__ mad,
                                                              const float* __restrict__ r0, int ld0, const float* __restrict__ r1, int ld1,
                                                              float* __restrict__ out, int ldo, int B, int HW, int C) {
    const int cq = C >> 2;
    const size_t total = (size_t)B * HW * cq;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cq) * 4;
        const size_t pix = i / cq;
        const int b = (int)(pix / HW);
        const float* m = mad + (size_t)b * 3 * C + c;
        f32x4 v = nd_silu4((nd_ld4(t + pix * ldt + c) - nd_ld4(m)) * nd_ld4(m + C) + nd_ld4(m + 2 * C));
        if (r0) v += nd_ld4(r0 + pix * ld0 + c);
        if (r1) v += nd_ld4(r1 + pix * ld1 + c);
        nd_st4(out + pix * ldo + c, v);
    }
}

// Per-pixel LayerNorm statistics of (x + vec[b]) over C channels.  A row's C/4 quads are spread over `lpr`
// lanes (power of two <= 64, up to 4 quads per lane => C <= 1024); a wave handles 64/lpr rows per pass.
template <int NJ>      // float4s of a row per lane: C = 4 * lpr * NJ exactly (host: C a multiple of 64 above 256, lpr = 64 there)
__global__ __launch_bounds__(256) void ln_stats_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ vec,
                                                       float* __restrict__ stats, int B, int HW, int C, float eps) {
    const int lane = threadIdx.x & 63;
    int lpr = 1;
    while (lpr < 64 && lpr * 4 < C) lpr <<= 1;
    const int rpp = 64 / lpr, sub = lane / lpr, ql = lane - sub * lpr;
    const size_t npix = (size_t)B * HW;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    constexpr int UN = NJ == 1 ? 4 : NJ == 2 ? 2 : 1;                // row groups per pass: four independent 16-byte loads in flight per lane
    const float inv_c = 1.0f / (float)C;
    for (size_t r0 = wave * (rpp * UN); r0 < npix; r0 += nwaves * (rpp * UN)) {
        f32x4 v[UN][NJ];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const size_t r = r0 + u * rpp + sub, rs = r < npix ? r : npix - 1;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int c = (ql + j * lpr) * 4, cs = c < C ? c : 0;
                v[u][j] = nd_ld4(x + rs * ldx + cs);
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const size_t r = r0 + u * rpp + sub, rs = r < npix ? r : npix - 1;
            const int b = (int)(rs / HW);
            float sum = 0.0f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int c = (ql + j * lpr) * 4;
                if (vec) v[u][j] += nd_ld4(vec + (size_t)b * C + (c < C ? c : 0));
                const f32x4 zero = {0, 0, 0, 0};
                v[u][j] = c < C ? v[u][j] : zero;
                sum += (v[u][j].x + v[u][j].y) + (v[u][j].z + v[u][j].w);
            }
            for (int o = lpr >> 1; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
            const float mean = sum * inv_c;
            float m2 = 0.0f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int c = (ql + j * lpr) * 4;
                const f32x4 dv = v[u][j] - mean;
                m2 += c < C ? (dv.x * dv.x + dv.y * dv.y) + (dv.z * dv.z + dv.w * dv.w) : 0.0f;
            }
            for (int o = lpr >> 1; o > 0; o >>= 1) m2 += __shfl_xor(m2, o);
            if (ql == 0 && r < npix) {
                stats[2 * r] = mean;
                stats[2 * r + 1] = rsqrtf(m2 * inv_c + eps);
            }
        }
    }
}

// RMSNorm.forward (Diffusion_arch.py:89-90): F.normalize(x, dim=channel) * g * sqrt(C); one wave per pixel.
// res != null: out = RMSNorm(x) + res (the residual around LinearAttention, whose last layer is an RMSNorm: Diffusion_arch.py:213-216).
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ g, const float* __restrict__ res, int ldr,
                                                      float* __restrict__ out, int ldo, size_t npix, int C) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const float rootc = sqrtf((float)C);
    for (size_t p = wave; p < npix; p += nwaves) {
        const float* row = x + p * ldx;
        float ssq = 0.0f;
        for (int c = lane * 4; c < C; c += 256) {
            const f32x4 v = nd_ld4(row + c);
            ssq += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ssq += __shfl_xor(ssq, o);
        const float inv = rootc / fmaxf(sqrtf(ssq), 1e-12f);
        for (int c = lane * 4; c < C; c += 256) {
            f32x4 v = nd_ld4(row + c) * inv * nd_ld4(g + c);
            if (res) v += nd_ld4(res + p * ldr + c);
            nd_st4(out + p * ldo + c, v);
        }
    }
}

}  // namespace

extern "C" int nd_groupnorm_finalize_f32(const float* stats, const float* slot_count, int slots, const float* gamma,
                                         const float* beta, const float* scale_shift, int ld_ss, float* mad, int B, int C,
                                         int groups, float eps, void* stream) {
    ND_REQUIRE(stats && slot_count && gamma && beta && mad, ND_E_BADARG, "nd_groupnorm_finalize: null pointer");
    ND_REQUIRE(B > 0 && C > 0 && groups > 0 && slots > 0 && C % groups == 0, ND_E_SHAPE, "nd_groupnorm_finalize: C=%d groups=%d", C, groups);
    ND_REQUIRE(!scale_shift || ld_ss >= 2 * C, ND_E_SHAPE, "nd_groupnorm_finalize: ld_ss < 2C");
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B * groups), dim3(GNF_THREADS), 0, (hipStream_t)stream, stats, slot_count, slots, gamma,
                       beta, scale_shift, ld_ss, mad, C, groups, eps);
    return nd_launch_status("nd_groupnorm_finalize_f32");
}

extern "C" int nd_groupnorm_finalize_train_f32(const float* stats, const float* slot_count, int slots, const float* gamma, const float* beta,
                                               const float* scale_shift, int ld_ss, float* mad, float* mean_rstd, int B, int C, int groups, float eps,
                                               void* stream) {
    ND_REQUIRE(stats && slot_count && gamma && beta && mad && mean_rstd, ND_E_BADARG, "nd_groupnorm_finalize_train: null pointer");
    ND_REQUIRE(B > 0 && C > 0 && groups > 0 && slots > 0 && C % groups == 0, ND_E_SHAPE, "nd_groupnorm_finalize_train: C=%d groups=%d", C, groups);
    ND_REQUIRE(!scale_shift || ld_ss >= 2 * C, ND_E_SHAPE, "nd_groupnorm_finalize_train: ld_ss < 2C");
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B * groups), dim3(GNF_THREADS), 0, (hipStream_t)stream, stats, slot_count, slots, gamma,
                       beta, scale_shift, ld_ss, mad, C, groups, eps, mean_rstd);
    return nd_launch_status("nd_groupnorm_finalize_train_f32");
}

extern "C" int nd_layernorm_stats_f32(const float* x, int ldx, const float* vec, float* stats, int B, int HW, int C, float eps, void* stream) {
    ND_REQUIRE(x && stats, ND_E_BADARG, "nd_layernorm_stats: null pointer");
    ND_REQUIRE(B > 0 && HW > 0 && C > 0 && C % 4 == 0 && C <= 1024 && ldx % 4 == 0 && ldx >= C, ND_E_SHAPE, "nd_layernorm_stats: C=%d (multiple of 4, <= 1024)", C);
    ND_REQUIRE(nd_aligned16(x) && nd_aligned16(vec), ND_E_ALIGN, "nd_layernorm_stats: alignment");
    const size_t npix = (size_t)B * HW;
    const size_t want = (npix + 15) / 16;
    const int blocks = (int)(want < 4096 ? (want ? want : 1) : 4096);
    const int nj = C <= 256 ? 1 : (C + 255) / 256;                    // 16-byte loads of a row per lane (the row's quads over up to 64 lanes)
    if (nj == 1) hipLaunchKernelGGL(ln_stats_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, vec, stats, B, HW, C, eps);
    else if (nj == 2) hipLaunchKernelGGL(ln_stats_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, vec, stats, B, HW, C, eps);
    else hipLaunchKernelGGL(ln_stats_kernel<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, vec, stats, B, HW, C, eps);
    return nd_launch_status("nd_layernorm_stats_f32");
}

extern "C" int nd_affine_silu_add_f32(const float* t, int ldt, const float* mad, const float* res0, int ldr0, const float* res1,
                                      int ldr1, float* out, int ldo, int B, int HW, int C, void* stream) {
    ND_REQUIRE(t && mad && out, ND_E_BADARG, "nd_affine_silu_add: null pointer");
    ND_REQUIRE(B > 0 && HW > 0 && C > 0 && C % 4 == 0, ND_E_SHAPE, "nd_affine_silu_add: C=%d must be a multiple of 4", C);
    ND_REQUIRE(ldt % 4 == 0 && ldo % 4 == 0 && (!res0 || ldr0 % 4 == 0) && (!res1 || ldr1 % 4 == 0), ND_E_ALIGN, "nd_affine_silu_add: strides");
    ND_REQUIRE(nd_aligned16(t) && nd_aligned16(mad) && nd_aligned16(res0) && nd_aligned16(res1) && nd_aligned16(out), ND_E_ALIGN,
               "nd_affine_silu_add: pointers must be 16-byte aligned");
    const size_t total = (size_t)B * HW * (C / 4);
#ifndef ASA_BLOCKS
#define ASA_BLOCKS 16384     // grid cap of the streaming pass (4096 -> 16384: 5.3 -> 5.6 TB/s in the per-launch measurement, same-box)
#endif
    const int blocks = (int)((total + 255) / 256 < ASA_BLOCKS ? (total + 255) / 256 : ASA_BLOCKS);
    hipLaunchKernelGGL(affine_silu_add_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, ldt, mad, res0, ldr0, res1, ldr1,
                       out, ldo, B, HW, C);
    return nd_launch_status("nd_affine_silu_add_f32");
}

extern "C" int nd_rmsnorm_nhwc_f32(const float* x, int ldx, const float* g, float* out, int ldo, int B, int HW, int C, void* stream) {
    ND_REQUIRE(x && g && out, ND_E_BADARG, "nd_rmsnorm: null pointer");
    ND_REQUIRE(B > 0 && HW > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0, ND_E_SHAPE, "nd_rmsnorm: C and strides must be multiples of 4");
    ND_REQUIRE(nd_aligned16(x) && nd_aligned16(g) && nd_aligned16(out), ND_E_ALIGN, "nd_rmsnorm: alignment");
    const size_t npix = (size_t)B * HW;
    const int blocks = (int)((npix + 3) / 4 < 4096 ? (npix + 3) / 4 : 4096);
    hipLaunchKernelGGL(rmsnorm_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, g, (const float*)nullptr, 0, out, ldo, npix, C);
    return nd_launch_status("nd_rmsnorm_nhwc_f32");
}

extern "C" int nd_rmsnorm_add_nhwc_f32(const float* x, int ldx, const float* g, const float* res, int ldr, float* out, int ldo, int B, int HW, int C,
                                       void* stream) {
    ND_REQUIRE(x && g && res && out, ND_E_BADARG, "nd_rmsnorm_add: null pointer");
    ND_REQUIRE(B > 0 && HW > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && ldr % 4 == 0 && ldx >= C && ldo >= C && ldr >= C, ND_E_SHAPE,
               "nd_rmsnorm_add: C and strides must be multiples of 4, strides >= C");
    ND_REQUIRE(nd_aligned16(x) && nd_aligned16(g) && nd_aligned16(res) && nd_aligned16(out), ND_E_ALIGN, "nd_rmsnorm_add: alignment");
    const size_t npix = (size_t)B * HW;
    const int blocks = (int)((npix + 3) / 4 < 4096 ? (npix + 3) / 4 : 4096);
    hipLaunchKernelGGL(rmsnorm_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, g, res, ldr, out, ldo, npix, C);
    return nd_launch_status("nd_rmsnorm_add_nhwc_f32");
}
