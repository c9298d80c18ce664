// norm_train.hip -- nn.GroupNorm forward AND backward on NHWC fp32 (second training slice): Block.norm under
// GaussianDiffusion.p_losses -> loss.backward() (models/archs/Diffusion_arch.py:132,138; models/denoising_diffusion_pytorch.py:481-531).
// PyTorch's native_group_norm is written for NCHW: on the channels_last tensors the HIP convolutions produce it first copies to
// NCHW (and the gradient back), and its row-moment kernel then runs at a fraction of HBM speed -- at 256 x 256 x 64 the norms cost
// more than the convolutions (tools/train_step_bench.py).  Here every pass streams NHWC once:
//
//   forward    partials  {sum(x - p), sum(x - p)^2} per (sample, pixel slot, channel), p = the sample's first pixel     [gn_partials_kernel, mode 0]
//              finalize  fp64 over slots and the group's channels -> {mean, rstd} per (sample, group), (M, A, D) per channel  [gn_fwd_finalize_kernel]
//              apply     y = (x - M) A + D                                                                               [affine3_kernel, mode 0]
//   backward   partials  {sum(dy x), sum(dy)} per (sample, slot, channel)                                                 [gn_partials_kernel, mode 1]
//              finalize  per (sample, channel) c0 = rstd gamma, c1 = -rstd^2 S1 / N, c2 = mean rstd^2 S1 / N - rstd S2 / N
//                        with S1 = sum_c gamma_c rstd (sum dy x - mean sum dy), S2 = sum_c gamma_c sum dy over the group  [gn_bwd_finalize_kernel];
//                        dgamma, dbeta = the same per-channel sums added over the samples                                [gn_dparam_kernel]
//              apply     dx = dy c0 + x c1 + c2                                                                          [affine3_kernel, mode 1]
//
// All sums have a fixed order (no atomics): bitwise repeatable.
#include "nd_common.h"

namespace {

#ifndef GT_SLOTS_MAX_N
#define GT_SLOTS_MAX_N 256       // pixel slots per sample of the partial-sum passes (64 in r2: 4 x 64 workgroups at B = 4 streamed a 256 x 256 x 64 tensor at 1.7 TB/s)
#endif
constexpr int GT_SLOTS_MAX = GT_SLOTS_MAX_N;

__host__ __device__ inline int gt_slots(int HW) { return HW >= 64 * GT_SLOTS_MAX ? GT_SLOTS_MAX : (HW + 63) / 64; }

// partial sums over a slot's pixels, one float4 of channels per thread column, the rows of the block strided over the pixels
__global__ __launch_bounds__(256) void gn_partials_kernel(const float* __restrict__ u, int ldu, const float* __restrict__ v, int ldv, int mode,
                                                         float* __restrict__ part, int HW, int C, int slots) {
    __shared__ __attribute__((aligned(16))) float red[2][256][4];
    const int Q = C >> 2, R = 256 / Q;                         // channel quads, pixel rows per pass (host: Q <= 256)
    const int tid = threadIdx.x, q = tid % Q, rq = tid / Q;
    const int b = blockIdx.x / slots, slot = blockIdx.x % slots;
    const int p_begin = (int)((long)slot * HW / slots), p_end = (int)((long)(slot + 1) * HW / slots);
    const float* ub = u + (size_t)b * HW * ldu + 4 * q;
    const float* vb = v + (size_t)b * HW * ldv + 4 * q;
    f32x4 s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
    if (rq < R) {
        const f32x4 pv = mode == 0 ? nd_ld4(ub) : f32x4{0, 0, 0, 0};   // pivot: the sample's first pixel (keeps sum of squares well conditioned)
        int p = p_begin + rq;
        for (; p + 3 * R < p_end; p += 4 * R) {                // four independent loads in flight
            f32x4 a[4], c[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                a[k] = nd_ld4(ub + (size_t)(p + k * R) * ldu);
                c[k] = mode == 0 ? a[k] : nd_ld4(vb + (size_t)(p + k * R) * ldv);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (mode == 0) { const f32x4 d = a[k] - pv; s1 += d; s2 += d * d; }
                else { s1 += a[k] * c[k]; s2 += a[k]; }
            }
        }
        for (; p < p_end; p += R) {
            const f32x4 a = nd_ld4(ub + (size_t)p * ldu);
            if (mode == 0) { const f32x4 d = a - pv; s1 += d; s2 += d * d; }
            else { s1 += a * nd_ld4(vb + (size_t)p * ldv); s2 += a; }
        }
    }
    *reinterpret_cast<f32x4*>(red[0][tid]) = s1;
    *reinterpret_cast<f32x4*>(red[1][tid]) = s2;
    __syncthreads();
    if (rq == 0) {                                             // fixed order over the rows
        for (int r = 1; r < R; ++r) {
            s1 += *reinterpret_cast<const f32x4*>(red[0][r * Q + q]);
            s2 += *reinterpret_cast<const f32x4*>(red[1][r * Q + q]);
        }
        float* o = part + (((size_t)b * slots + slot) * C + 4 * q) * 2;
        nd_st4(o, f32x4{s1.x, s2.x, s1.y, s2.y});
        nd_st4(o + 4, f32x4{s1.z, s2.z, s1.w, s2.w});
    }
}

// Per-channel totals of a (sample, group) over the slots, fp64, fixed order: thread (channel i, stripe) adds its slots, the stripes
// of a channel are then added in order.  Result in tot[0][i], tot[1][i] (i < cpg <= 512); 256 threads.
__device__ __forceinline__ void gt_channel_totals(const float* __restrict__ part, int slots, int C, int b, int g, int cpg, double (&tot)[2][512],
                                                  double (&red)[2][256]) {
    const int tid = threadIdx.x;
    if (256 % cpg == 0) {
        const int i = tid % cpg, stripe = tid / cpg, nstr = 256 / cpg, c = g * cpg + i;
        double s1 = 0.0, s2 = 0.0;
        for (int s = stripe; s < slots; s += nstr) {
            const float* o = part + (((size_t)b * slots + s) * C + c) * 2;
            s1 += (double)o[0];  s2 += (double)o[1];
        }
        red[0][tid] = s1;  red[1][tid] = s2;
        __syncthreads();
        if (stripe == 0) {
            for (int r = 1; r < nstr; ++r) { s1 += red[0][r * cpg + i]; s2 += red[1][r * cpg + i]; }
            tot[0][i] = s1;  tot[1][i] = s2;
        }
    } else {                                                   // group widths that do not divide 256: a channel per thread, all its slots
        for (int i = tid; i < cpg; i += 256) {
            const int c = g * cpg + i;
            double s1 = 0.0, s2 = 0.0;
            for (int s = 0; s < slots; ++s) {
                const float* o = part + (((size_t)b * slots + s) * C + c) * 2;
                s1 += (double)o[0];  s2 += (double)o[1];
            }
            tot[0][i] = s1;  tot[1][i] = s2;
        }
    }
    __syncthreads();
}

// one workgroup per (sample, group)
// (ss != null: the time embedding's per-(sample, channel) scale | shift folded into A and D: what gs_mad_kernel computed in a launch of its own)
__global__ __launch_bounds__(256) void gn_fwd_finalize_kernel(const float* __restrict__ part, int slots, int HW, const float* __restrict__ x, int ldx,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ mean_rstd, float* __restrict__ mad, int C, int G, float eps,
                                                             const float* __restrict__ ss = nullptr) {
    __shared__ double tot[2][512], red[2][256], stat[2];
    const int b = blockIdx.x / G, g = blockIdx.x % G, cpg = C / G, tid = threadIdx.x;
    gt_channel_totals(part, slots, C, b, g, cpg, tot, red);
    if (tid == 0) {                                            // un-shift every channel by its pivot, then the group's moments (in channel order)
        double S = 0.0, Q2 = 0.0;
        const double n = (double)HW;
        for (int i = 0; i < cpg; ++i) {
            const double p = (double)x[(size_t)b * HW * ldx + g * cpg + i];
            S += tot[0][i] + n * p;
            Q2 += tot[1][i] + 2.0 * p * tot[0][i] + n * p * p;
        }
        const double N = (double)cpg * n, mean = S / N;
        double var = Q2 / N - mean * mean;
        var = var > 0.0 ? var : 0.0;
        stat[0] = mean;  stat[1] = 1.0 / sqrt(var + (double)eps);
        mean_rstd[((size_t)b * G + g) * 2] = (float)mean;
        mean_rstd[((size_t)b * G + g) * 2 + 1] = (float)stat[1];
    }
    __syncthreads();
    const float fmean = (float)stat[0], rstd = (float)stat[1];
    for (int i = tid; i < cpg; i += 256) {
        const int c = g * cpg + i;
        float* o = mad + (size_t)b * 3 * C + c;
        const float s1 = ss ? ss[(size_t)b * 2 * C + c] + 1.0f : 1.0f, h = ss ? ss[(size_t)b * 2 * C + C + c] : 0.0f;
        o[0] = fmean;  o[C] = rstd * gamma[c] * s1;  o[2 * C] = beta[c] * s1 + h;
    }
}

// one workgroup per (sample, group): the coefficients of dx and this sample's terms {sum dy xhat, sum dy} of dgamma / dbeta
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const float* __restrict__ part, int slots, int HW, const float* __restrict__ mean_rstd,
                                                             const float* __restrict__ gamma, float* __restrict__ coef, float* __restrict__ ab,
                                                             int C, int G) {
    __shared__ double tot[2][512], red[2][256], sums[2];
    const int b = blockIdx.x / G, g = blockIdx.x % G, cpg = C / G, tid = threadIdx.x;
    gt_channel_totals(part, slots, C, b, g, cpg, tot, red);
    const double mean = (double)mean_rstd[((size_t)b * G + g) * 2], rstd = (double)mean_rstd[((size_t)b * G + g) * 2 + 1];
    for (int i = tid; i < cpg; i += 256) {                     // tot <- {A = sum dy xhat, B = sum dy} per channel
        const double sdx = tot[0][i], sd = tot[1][i];
        tot[0][i] = rstd * (sdx - mean * sd);
        tot[1][i] = sd;
    }
    __syncthreads();
    if (tid == 0) {
        double S1 = 0.0, S2 = 0.0;
        for (int i = 0; i < cpg; ++i) { const double gm = (double)gamma[g * cpg + i]; S1 += gm * tot[0][i]; S2 += gm * tot[1][i]; }
        sums[0] = S1;  sums[1] = S2;
    }
    __syncthreads();
    const double N = (double)cpg * (double)HW;
    const float c1 = (float)(-rstd * rstd * sums[0] / N), c2 = (float)(mean * rstd * rstd * sums[0] / N - rstd * sums[1] / N);
    for (int i = tid; i < cpg; i += 256) {
        const int c = g * cpg + i;
        float* o = coef + (size_t)b * 3 * C + c;
        o[0] = (float)rstd * gamma[c];  o[C] = c1;  o[2 * C] = c2;
        ab[((size_t)b * C + c) * 2] = (float)tot[0][i];
        ab[((size_t)b * C + c) * 2 + 1] = (float)tot[1][i];
    }
}

// dgamma[c] = sum over the samples of sum dy xhat, dbeta[c] = ... of sum dy (in sample order)
__global__ __launch_bounds__(256) void gn_dparam_kernel(const float* __restrict__ ab, float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double dg = 0.0, db = 0.0;
    for (int b = 0; b < B; ++b) { dg += (double)ab[((size_t)b * C + c) * 2]; db += (double)ab[((size_t)b * C + c) * 2 + 1]; }
    dgamma[c] = (float)dg;  dbeta[c] = (float)db;
}

// mode 0: out = (u - c0) c1 + c2;  mode 1: out = u c0 + v c1 + c2;  coefficients per (sample, channel).  Pure HBM streaming.
__global__ __launch_bounds__(256) void affine3_kernel(const float* __restrict__ u, int ldu, const float* __restrict__ v, int ldv,
                                                      const float* __restrict__ coef, float* __restrict__ out, int ldo, int B, int HW, int C, int mode) {
    const int cq = C >> 2;
    const size_t total = (size_t)B * HW * cq;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % cq);
        const size_t pix = i / cq;
        const int b = (int)(pix / HW);
        const float* m = coef + (size_t)b * 3 * C + 4 * q;
        const f32x4 c0 = nd_ld4(m), c1 = nd_ld4(m + C), c2 = nd_ld4(m + 2 * C);
        const f32x4 a = nd_ld4(u + pix * ldu + 4 * q);
        f32x4 r;
        if (mode == 0) r = (a - c0) * c1 + c2;
        else r = a * c0 + nd_ld4(v + pix * ldv + 4 * q) * c1 + c2;
        nd_st4(out + pix * ldo + 4 * q, r);
    }
}

}  // namespace

extern "C" int nd_groupnorm_train_forward_f32(const float* x, int ldx, const float* gamma, const float* beta, float* y, int ldy, float* mean_rstd,
                                              float* workspace, int B, int HW, int C, int groups, float eps, void* stream) {
    ND_REQUIRE(x && gamma && beta && y && mean_rstd && workspace, ND_E_BADARG, "nd_groupnorm_train_forward: null pointer");
    ND_REQUIRE(B > 0 && HW > 0 && C > 0 && groups > 0 && C % groups == 0 && C % 4 == 0 && C <= 1024 && C / groups <= 512, ND_E_SHAPE,
               "nd_groupnorm_train_forward: C=%d groups=%d (C a multiple of 4 and of groups, <= 1024)", C, groups);
    ND_REQUIRE(ldx >= C && ldy >= C && ldx % 4 == 0 && ldy % 4 == 0 && nd_aligned16(x) && nd_aligned16(y) && nd_aligned16(workspace), ND_E_ALIGN,
               "nd_groupnorm_train_forward: strides must be multiples of 4 floats >= C, pointers 16-byte aligned");
    const int slots = gt_slots(HW);
    float* part = workspace;                                   // [B][slots][C][2]
    float* mad = workspace + (size_t)B * slots * C * 2;        // [B][3][C]
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gn_partials_kernel, dim3(B * slots), dim3(256), 0, st, x, ldx, x, ldx, 0, part, HW, C, slots);
    hipLaunchKernelGGL(gn_fwd_finalize_kernel, dim3(B * groups), dim3(256), 0, st, part, slots, HW, x, ldx, gamma, beta, mean_rstd, mad, C, groups, eps);
    const size_t total = (size_t)B * HW * (C / 4);
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(affine3_kernel, dim3(blocks), dim3(256), 0, st, x, ldx, x, ldx, mad, y, ldy, B, HW, C, 0);
    return nd_launch_status("nd_groupnorm_train_forward_f32");
}

extern "C" int64_t nd_groupnorm_train_workspace_floats(int B, int HW, int C) {
    if (B <= 0 || HW <= 0 || C <= 0) return -1;
    return (int64_t)B * gt_slots(HW) * C * 2 + (int64_t)B * 3 * C + (int64_t)B * C * 2;     // slot partials, per-channel coefficients, per-sample dgamma / dbeta terms
}

extern "C" int nd_groupnorm_train_backward_f32(const float* dy, int lddy, const float* x, int ldx, const float* gamma, const float* mean_rstd,
                                               float* dx, int lddx, float* dgamma, float* dbeta, float* workspace, int B, int HW, int C, int groups,
                                               void* stream) {
    ND_REQUIRE(dy && x && gamma && mean_rstd && dx && dgamma && dbeta && workspace, ND_E_BADARG, "nd_groupnorm_train_backward: null pointer");
    ND_REQUIRE(B > 0 && HW > 0 && C > 0 && groups > 0 && C % groups == 0 && C % 4 == 0 && C <= 1024 && C / groups <= 512, ND_E_SHAPE,
               "nd_groupnorm_train_backward: C=%d groups=%d (C a multiple of 4 and of groups, <= 1024)", C, groups);
    ND_REQUIRE(lddy >= C && ldx >= C && lddx >= C && lddy % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0 && nd_aligned16(dy) && nd_aligned16(x) &&
               nd_aligned16(dx) && nd_aligned16(workspace), ND_E_ALIGN,
               "nd_groupnorm_train_backward: strides must be multiples of 4 floats >= C, pointers 16-byte aligned");
    const int slots = gt_slots(HW);
    float* part = workspace;
    float* coef = workspace + (size_t)B * slots * C * 2;
    float* ab = coef + (size_t)B * 3 * C;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gn_partials_kernel, dim3(B * slots), dim3(256), 0, st, dy, lddy, x, ldx, 1, part, HW, C, slots);
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(B * groups), dim3(256), 0, st, part, slots, HW, mean_rstd, gamma, coef, ab, C, groups);
    hipLaunchKernelGGL(gn_dparam_kernel, dim3(nd_cdiv(C, 256)), dim3(256), 0, st, ab, dgamma, dbeta, B, C);
    const size_t total = (size_t)B * HW * (C / 4);
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(affine3_kernel, dim3(blocks), dim3(256), 0, st, dy, lddy, x, ldx, coef, dx, lddx, B, HW, C, 1);
    return nd_launch_status("nd_groupnorm_train_backward_f32");
}

// ====================================================================================================================================
// nn.LayerNorm over the channels of NHWC tokens, forward and backward (AttnBlock.norm1 / norm2, Diffusion_arch.py:427-428,438-439):
// a row of C channels is held by G = min(64, C / 4) lanes (one float4 each, C / 256 of them beyond C = 256), so a wave works on
// 64 / G rows at once and every row reduction is a DPP / shuffle sum inside the lane group.  PyTorch's kernels take 100-170 us for
// the 268 MB of a full-resolution token tensor; one streaming pass is 50-90.
//   forward    y = (x - mean) rstd gamma + beta, {mean, rstd} per row saved
//   backward   a = dy gamma, dx = rstd (a - mean_c(a) - xhat mean_c(a xhat)); per-lane running sums of dy xhat / dy over the workgroup's
//              rows -> partials [workgroup][C][2] -> dgamma / dbeta in a second kernel (fixed order)
namespace {

template <int G>
__device__ __forceinline__ float lt_group_sum(float v) {
    v = nd_row16_sum(v);
    if (G >= 32) v += __shfl_xor(v, 16);
    if (G >= 64) v += __shfl_xor(v, 32);
    return v;
}

constexpr int LT_MAXQ = 4;                                               // float4s per lane: C <= 1024

template <int G>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    float* __restrict__ y, int ldy, float* __restrict__ stats, long N, int C, float eps) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // a row's C / 4 channel quads over the G lanes of its group: quad g + k G for k < QL; a width that is not 4 G QL (C = 48, 96, 192, 384: the d = 48 network)
    // leaves the last lanes of the group without a quad in the last round -- they load nothing and add zeros to the row sums
    const int g = lane % G, rw = lane / G, RPW = 64 / G, CQ = C / 4, QL = (CQ + G - 1) / G;
    f32x4 gm[LT_MAXQ], bt[LT_MAXQ];
#pragma unroll
    for (int k = 0; k < LT_MAXQ; ++k)
        if (k < QL && g + k * G < CQ) { gm[k] = nd_ld4(gamma + 4 * (g + k * G)); bt[k] = nd_ld4(beta + 4 * (g + k * G)); }
    const float inv = 1.0f / (float)C;
    for (long row = ((long)blockIdx.x * 4 + wave) * RPW + rw; row < N + rw; row += (long)gridDim.x * 4 * RPW) {
        const bool ok = row < N;                                         // (all lanes of a wave stay in the loop together: the sums are wave-wide instructions)
        const long rr = ok ? row : N - 1;
        f32x4 v[LT_MAXQ];
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < LT_MAXQ; ++k)
            if (k < QL && g + k * G < CQ) { v[k] = nd_ld4(x + (size_t)rr * ldx + 4 * (g + k * G)); s += v[k].x + v[k].y + v[k].z + v[k].w; }
        const float mean = lt_group_sum<G>(s) * inv;
        float q = 0.0f;
#pragma unroll
        for (int k = 0; k < LT_MAXQ; ++k)
            if (k < QL && g + k * G < CQ) { v[k] = v[k] - mean; q += v[k].x * v[k].x + v[k].y * v[k].y + v[k].z * v[k].z + v[k].w * v[k].w; }
        const float rstd = rsqrtf(lt_group_sum<G>(q) * inv + eps);
        if (ok) {
#pragma unroll
            for (int k = 0; k < LT_MAXQ; ++k)
                if (k < QL && g + k * G < CQ) nd_st4(y + (size_t)row * ldy + 4 * (g + k * G), v[k] * rstd * gm[k] + bt[k]);
            if (g == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
        }
    }
}

template <int G>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ x, int ldx,
                                                    const float* __restrict__ gamma, const float* __restrict__ stats, float* __restrict__ dx, int lddx,
                                                    float* __restrict__ part, long N, int C, long rows_per_wg) {
    __shared__ __attribute__((aligned(16))) float red[2][256][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane % G, rw = lane / G, RPW = 64 / G, CQ = C / 4, QL = (CQ + G - 1) / G;
    f32x4 gm[LT_MAXQ], dg[LT_MAXQ], db[LT_MAXQ];
#pragma unroll
    for (int k = 0; k < LT_MAXQ; ++k) {
        dg[k] = f32x4{0, 0, 0, 0};  db[k] = f32x4{0, 0, 0, 0};
        if (k < QL && g + k * G < CQ) gm[k] = nd_ld4(gamma + 4 * (g + k * G));
    }
    const float inv = 1.0f / (float)C;
    const long r_begin = (long)blockIdx.x * rows_per_wg, r_end = min(r_begin + rows_per_wg, N);
    for (long row = r_begin + wave * RPW + rw; row < r_end + rw; row += 4 * RPW) {
        const bool ok = row < r_end;
        const long rr = ok ? row : max(r_end - 1, 0L);
        const float mean = stats[2 * rr], rstd = stats[2 * rr + 1];
        f32x4 a[LT_MAXQ], xh[LT_MAXQ];
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int k = 0; k < LT_MAXQ; ++k)
            if (k < QL && g + k * G < CQ) {
                const f32x4 d = nd_ld4(dy + (size_t)rr * lddy + 4 * (g + k * G));
                xh[k] = (nd_ld4(x + (size_t)rr * ldx + 4 * (g + k * G)) - mean) * rstd;
                a[k] = d * gm[k];
                if (ok) { dg[k] += d * xh[k];  db[k] += d; }
                const f32x4 ax = a[k] * xh[k];
                s1 += ax.x + ax.y + ax.z + ax.w;
                s2 += a[k].x + a[k].y + a[k].z + a[k].w;
            }
        s1 = lt_group_sum<G>(s1) * inv;
        s2 = lt_group_sum<G>(s2) * inv;
        if (ok) {
#pragma unroll
            for (int k = 0; k < LT_MAXQ; ++k)
                if (k < QL && g + k * G < CQ) nd_st4(dx + (size_t)row * lddx + 4 * (g + k * G), (a[k] - s2 - xh[k] * s1) * rstd);
        }
    }
    // the workgroup's column sums: the 4 * RPW row lanes of every channel quad meet in LDS, in lane order
    float* o = part + (size_t)blockIdx.x * C * 2;
#pragma unroll
    for (int k = 0; k < LT_MAXQ; ++k) {
        if (k < QL) {                                                    // (QL is uniform over the workgroup: the barriers below are reached by everyone)
            __syncthreads();
            *reinterpret_cast<f32x4*>(red[0][tid]) = dg[k];
            *reinterpret_cast<f32x4*>(red[1][tid]) = db[k];
            __syncthreads();
            if (tid < G && tid + k * G < CQ) {
                f32x4 sg = {0, 0, 0, 0}, sb = {0, 0, 0, 0};
                for (int w = 0; w < 4; ++w)
                    for (int r = 0; r < RPW; ++r) {
                        sg += *reinterpret_cast<const f32x4*>(red[0][w * 64 + r * G + tid]);
                        sb += *reinterpret_cast<const f32x4*>(red[1][w * 64 + r * G + tid]);
                    }
                const int c = 4 * (tid + k * G);
                nd_st4(o + 2 * c, f32x4{sg.x, sb.x, sg.y, sb.y});
                nd_st4(o + 2 * c + 4, f32x4{sg.z, sb.z, sg.w, sb.w});
            }
        }
    }
}

// dgamma[c], dbeta[c] = sum over the workgroups' partials: 32 channels per workgroup, eight stripes of partials (w = stripe, stripe + 8, ...,
// eight loads in flight each) that meet in stripe order -- fixed order, fp64
__global__ __launch_bounds__(256) void ln_dparam_kernel(const float* __restrict__ part, float* __restrict__ dgamma, float* __restrict__ dbeta, int wgs, int C) {
    __shared__ double red[2][8][32];
    const int o = threadIdx.x & 31, stripe = threadIdx.x >> 5, c = blockIdx.x * 32 + o;
    double sg = 0.0, sb = 0.0;
    if (c < C) {
        const float2* p = reinterpret_cast<const float2*>(part) + c;
        int w = stripe;
        for (; w + 56 < wgs; w += 64) {
            float2 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = p[(size_t)(w + 8 * k) * C];
#pragma unroll
            for (int k = 0; k < 8; ++k) { sg += (double)v[k].x; sb += (double)v[k].y; }
        }
        for (; w < wgs; w += 8) { const float2 v = p[(size_t)w * C]; sg += (double)v.x; sb += (double)v.y; }
    }
    red[0][stripe][o] = sg;  red[1][stripe][o] = sb;
    __syncthreads();
    if (stripe == 0 && c < C) {
#pragma unroll
        for (int k = 1; k < 8; ++k) { sg += red[0][k][o]; sb += red[1][k][o]; }
        dgamma[c] = (float)sg;  dbeta[c] = (float)sb;
    }
}

constexpr int LT_BWD_WGS = 1024;                                         // fixed: the summation order must not depend on the device

inline int lt_group(int C) { return C > 128 ? 64 : C > 64 ? 32 : 16; }             // lanes per row: 16, 32 or 64 (C / 4 quads over them, up to LT_MAXQ rounds)
inline bool lt_ok(int C) { return C % 4 == 0 && C >= 16 && C <= 1024; }

}  // namespace

extern "C" int64_t nd_layernorm_train_workspace_floats(int64_t N, int C) { return N > 0 && C > 0 ? (int64_t)LT_BWD_WGS * C * 2 : -1; }

extern "C" int nd_layernorm_train_forward_f32(const float* x, int ldx, const float* gamma, const float* beta, float* y, int ldy, float* stats,
                                              int64_t N, int C, float eps, void* stream) {
    ND_REQUIRE(x && gamma && beta && y && stats, ND_E_BADARG, "nd_layernorm_train_forward: null pointer");
    ND_REQUIRE(N > 0 && lt_ok(C), ND_E_SHAPE, "nd_layernorm_train_forward: C=%d (a multiple of 4 in 16 .. 1024)", C);
    ND_REQUIRE(ldx >= C && ldy >= C && ldx % 4 == 0 && ldy % 4 == 0 && nd_aligned16(x) && nd_aligned16(y) && nd_aligned16(gamma) && nd_aligned16(beta),
               ND_E_ALIGN, "nd_layernorm_train_forward: strides must be multiples of 4 floats >= C, pointers 16-byte aligned");
    const int G = lt_group(C), rows_per_pass = 4 * (64 / G);
    const long want = (N + rows_per_pass - 1) / rows_per_pass;
    const dim3 grid((unsigned)(want < 8192 ? want : 8192)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (G == 16) hipLaunchKernelGGL(ln_fwd_kernel<16>, grid, block, 0, st, x, ldx, gamma, beta, y, ldy, stats, (long)N, C, eps);
    else if (G == 32) hipLaunchKernelGGL(ln_fwd_kernel<32>, grid, block, 0, st, x, ldx, gamma, beta, y, ldy, stats, (long)N, C, eps);
    else hipLaunchKernelGGL(ln_fwd_kernel<64>, grid, block, 0, st, x, ldx, gamma, beta, y, ldy, stats, (long)N, C, eps);
    return nd_launch_status("nd_layernorm_train_forward_f32");
}

extern "C" int nd_layernorm_train_backward_f32(const float* dy, int lddy, const float* x, int ldx, const float* gamma, const float* stats,
                                               float* dx, int lddx, float* dgamma, float* dbeta, float* workspace, int64_t N, int C, void* stream) {
    ND_REQUIRE(dy && x && gamma && stats && dx && dgamma && dbeta && workspace, ND_E_BADARG, "nd_layernorm_train_backward: null pointer");
    ND_REQUIRE(N > 0 && lt_ok(C), ND_E_SHAPE, "nd_layernorm_train_backward: C=%d (a multiple of 4 in 16 .. 1024)", C);
    ND_REQUIRE(lddy >= C && ldx >= C && lddx >= C && lddy % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0 && nd_aligned16(dy) && nd_aligned16(x) &&
               nd_aligned16(dx) && nd_aligned16(gamma) && nd_aligned16(workspace), ND_E_ALIGN,
               "nd_layernorm_train_backward: strides must be multiples of 4 floats >= C, pointers 16-byte aligned");
    const int G = lt_group(C);
    const long rows_per_wg = (N + LT_BWD_WGS - 1) / LT_BWD_WGS;
    const int wgs = (int)((N + rows_per_wg - 1) / rows_per_wg);
    const dim3 grid(wgs), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (G == 16) hipLaunchKernelGGL(ln_bwd_kernel<16>, grid, block, 0, st, dy, lddy, x, ldx, gamma, stats, dx, lddx, workspace, (long)N, C, rows_per_wg);
    else if (G == 32) hipLaunchKernelGGL(ln_bwd_kernel<32>, grid, block, 0, st, dy, lddy, x, ldx, gamma, stats, dx, lddx, workspace, (long)N, C, rows_per_wg);
    else hipLaunchKernelGGL(ln_bwd_kernel<64>, grid, block, 0, st, dy, lddy, x, ldx, gamma, stats, dx, lddx, workspace, (long)N, C, rows_per_wg);
    hipLaunchKernelGGL(ln_dparam_kernel, dim3(nd_cdiv(C, 32)), dim3(256), 0, st, workspace, dgamma, dbeta, wgs, C);
    return nd_launch_status("nd_layernorm_train_backward_f32");
}

// ====================================================================================================================================
// Block's tail as ONE operator for training: y = silu(GroupNorm(x) * (scale + 1) + shift) with per-(sample, channel) scale / shift from the
// time embedding (Block.forward, Diffusion_arch.py:137-143; ResnetBlock.mlp :150-152,162-164), forward and backward.  As separate PyTorch
// ops this is the norm plus three elementwise passes forward and five backward over full-resolution tensors; here the forward is the two
// passes of the norm (the activation rides on its apply pass) and the backward its two passes with d(silu) recomputed from x on the fly.
//   forward    m = (x - M) A + D with A = rstd gamma (1 + s), D = beta (1 + s) + h, M = mean;  y = m sigmoid(m);  (M, A, D) saved
//   backward   dm = dy sigmoid(m) (1 + m (1 - sigmoid(m)));  P1 = rstd sum_p dm (x - M), P2 = sum_p dm per (sample, channel)
//              dshift = P2, dscale = gamma P1 + beta P2, dgamma = sum_b (1 + s) P1, dbeta = sum_b (1 + s) P2
//              dx = dm A + x c1 + c2, c1 = -rstd^2 S1 / N, c2 = mean rstd^2 S1 / N - rstd S2 / N, S1 = sum_c gamma (1 + s) P1, S2 = sum_c gamma (1 + s) P2
namespace {

__device__ __forceinline__ f32x4 gs_dsilu(f32x4 dy, f32x4 m) {           // dy * d/dm (m sigmoid(m))
    f32x4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-m[k]));
        r[k] = dy[k] * sg * (1.0f + m[k] * (1.0f - sg));
    }
    return r;
}

// {sum dm (x - M), sum dm} per (sample, slot, channel); same layout and reduction as gn_partials_kernel
__global__ __launch_bounds__(256) void gs_partials_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ x, int ldx,
                                                         const float* __restrict__ mad, float* __restrict__ part, int HW, int C, int slots) {
    __shared__ __attribute__((aligned(16))) float red[2][256][4];
    const int Q = C >> 2, R = 256 / Q;
    const int tid = threadIdx.x, q = tid % Q, rq = tid / Q;
    const int b = blockIdx.x / slots, slot = blockIdx.x % slots;
    const int p_begin = (int)((long)slot * HW / slots), p_end = (int)((long)(slot + 1) * HW / slots);
    const float* db = dy + (size_t)b * HW * lddy + 4 * q;
    const float* xb = x + (size_t)b * HW * ldx + 4 * q;
    f32x4 s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
    if (rq < R) {
        const float* m = mad + (size_t)b * 3 * C + 4 * q;
        const f32x4 M = nd_ld4(m), A = nd_ld4(m + C), D = nd_ld4(m + 2 * C);
        int p = p_begin + rq;
        for (; p + 3 * R < p_end; p += 4 * R) {
            f32x4 g[4], v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { g[k] = nd_ld4(db + (size_t)(p + k * R) * lddy); v[k] = nd_ld4(xb + (size_t)(p + k * R) * ldx); }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 xc = v[k] - M, dm = gs_dsilu(g[k], xc * A + D);
                s1 += dm * xc;  s2 += dm;
            }
        }
        for (; p < p_end; p += R) {
            const f32x4 xc = nd_ld4(xb + (size_t)p * ldx) - M, dm = gs_dsilu(nd_ld4(db + (size_t)p * lddy), xc * A + D);
            s1 += dm * xc;  s2 += dm;
        }
    }
    *reinterpret_cast<f32x4*>(red[0][tid]) = s1;
    *reinterpret_cast<f32x4*>(red[1][tid]) = s2;
    __syncthreads();
    if (rq == 0) {
        for (int r = 1; r < R; ++r) {
            s1 += *reinterpret_cast<const f32x4*>(red[0][r * Q + q]);
            s2 += *reinterpret_cast<const f32x4*>(red[1][r * Q + q]);
        }
        float* o = part + (((size_t)b * slots + slot) * C + 4 * q) * 2;
        nd_st4(o, f32x4{s1.x, s2.x, s1.y, s2.y});
        nd_st4(o + 4, f32x4{s1.z, s2.z, s1.w, s2.w});
    }
}

// (M, A, D) of the fused forward from the group statistics: one thread per (sample, channel)
__global__ __launch_bounds__(256) void gs_mad_kernel(const float* __restrict__ mean_rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    const float* __restrict__ ss, float* __restrict__ mad, int B, int C, int G) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i - b * C, g = c / (C / G);
    const float mean = mean_rstd[((size_t)b * G + g) * 2], rstd = mean_rstd[((size_t)b * G + g) * 2 + 1];
    const float s1 = ss ? ss[(size_t)b * 2 * C + c] + 1.0f : 1.0f, h = ss ? ss[(size_t)b * 2 * C + C + c] : 0.0f;
    float* o = mad + (size_t)b * 3 * C + c;
    o[0] = mean;  o[C] = rstd * gamma[c] * s1;  o[2 * C] = beta[c] * s1 + h;
}

// one workgroup per (sample, group): c1 / c2 of dx, dscale / dshift, this sample's terms of dgamma / dbeta
__global__ __launch_bounds__(256) void gs_bwd_finalize_kernel(const float* __restrict__ part, int slots, int HW, const float* __restrict__ mean_rstd,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ ss,
                                                             float* __restrict__ coef, float* __restrict__ ab, float* __restrict__ dss, int C, int G) {
    __shared__ double tot[2][512], red[2][256], sums[2];
    const int b = blockIdx.x / G, g = blockIdx.x % G, cpg = C / G, tid = threadIdx.x;
    gt_channel_totals(part, slots, C, b, g, cpg, tot, red);
    const double mean = (double)mean_rstd[((size_t)b * G + g) * 2], rstd = (double)mean_rstd[((size_t)b * G + g) * 2 + 1];
    for (int i = tid; i < cpg; i += 256) {
        const int c = g * cpg + i;
        const double P1 = rstd * tot[0][i], P2 = tot[1][i];
        const double s1 = ss ? (double)ss[(size_t)b * 2 * C + c] + 1.0 : 1.0;
        if (dss) {
            dss[(size_t)b * 2 * C + c] = (float)((double)gamma[c] * P1 + (double)beta[c] * P2);     // d scale = sum_p dm n
            dss[(size_t)b * 2 * C + C + c] = (float)P2;                                           // d shift
        }
        tot[0][i] = s1 * P1;  tot[1][i] = s1 * P2;                                                // sum_p dn xhat, sum_p dn
    }
    __syncthreads();
    if (tid == 0) {
        double S1 = 0.0, S2 = 0.0;
        for (int i = 0; i < cpg; ++i) { const double gm = (double)gamma[g * cpg + i]; S1 += gm * tot[0][i]; S2 += gm * tot[1][i]; }
        sums[0] = S1;  sums[1] = S2;
    }
    __syncthreads();
    const double N = (double)cpg * (double)HW;
    const float c1 = (float)(-rstd * rstd * sums[0] / N), c2 = (float)(mean * rstd * rstd * sums[0] / N - rstd * sums[1] / N);
    for (int i = tid; i < cpg; i += 256) {
        const int c = g * cpg + i;
        coef[(size_t)b * 2 * C + c] = c1;  coef[(size_t)b * 2 * C + C + c] = c2;
        ab[((size_t)b * C + c) * 2] = (float)tot[0][i];
        ab[((size_t)b * C + c) * 2 + 1] = (float)tot[1][i];
    }
}

// mode 0: y = silu((x - M) A + D) (+ res: the ResnetBlock's shortcut, Diffusion_arch.py:170);  mode 1: dx = dm A + x c1 + c2 with dm recomputed from (dy, x)
__global__ __launch_bounds__(256) void gs_apply_kernel(const float* __restrict__ u, int ldu, const float* __restrict__ x, int ldx, const float* __restrict__ mad,
                                                      const float* __restrict__ coef, float* __restrict__ out, int ldo, int B, int HW, int C, int mode,
                                                      const float* __restrict__ res = nullptr, int ldr = 0, const float* __restrict__ ab = nullptr,
                                                      float* __restrict__ dgamma = nullptr, float* __restrict__ dbeta = nullptr) {
    if (ab && blockIdx.x == 0) {      // dgamma / dbeta = the per-sample terms added in sample order (gn_dparam_kernel's work, without its launch)
        for (int c = threadIdx.x; c < C; c += 256) {
            double dg = 0.0, db = 0.0;
            for (int b = 0; b < B; ++b) { dg += (double)ab[((size_t)b * C + c) * 2]; db += (double)ab[((size_t)b * C + c) * 2 + 1]; }
            dgamma[c] = (float)dg;  dbeta[c] = (float)db;
        }
    }
    const int cq = C >> 2;
    const size_t total = (size_t)B * HW * cq;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % cq);
        const size_t pix = i / cq;
        const int b = (int)(pix / HW);
        const float* m = mad + (size_t)b * 3 * C + 4 * q;
        const f32x4 M = nd_ld4(m), A = nd_ld4(m + C), D = nd_ld4(m + 2 * C);
        const f32x4 xv = nd_ld4(x + pix * ldx + 4 * q), xc = xv - M, mm = xc * A + D;
        f32x4 r;
        if (mode == 0) {
            r = nd_silu4(mm);
            if (res) r += nd_ld4(res + pix * ldr + 4 * q);
        } else {
            const float* cf = coef + (size_t)b * 2 * C + 4 * q;
            r = gs_dsilu(nd_ld4(u + pix * ldu + 4 * q), mm) * A + xv * nd_ld4(cf) + nd_ld4(cf + C);
        }
        nd_st4(out + pix * ldo + 4 * q, r);
    }
}

int gs_check(const char* who, int B, int HW, int C, int groups) {
    ND_REQUIRE(B > 0 && HW > 0 && C > 0 && groups > 0 && C % groups == 0 && C % 4 == 0 && C <= 1024 && C / groups <= 512, ND_E_SHAPE,
               "%s: C=%d groups=%d (C a multiple of 4 and of groups, <= 1024)", who, C, groups);
    return 0;
}

}  // namespace

extern "C" int64_t nd_groupnorm_silu_train_workspace_floats(int B, int HW, int C) {
    if (B <= 0 || HW <= 0 || C <= 0) return -1;
    return (int64_t)B * gt_slots(HW) * C * 2 + (int64_t)B * 3 * C + (int64_t)B * 2 * C + (int64_t)B * C * 2;
}

extern "C" int nd_groupnorm_silu_train_forward_f32(const float* x, int ldx, const float* gamma, const float* beta, const float* scale_shift, const float* res, int ldr,
                                                   float* y, int ldy, float* mean_rstd, float* mad, float* workspace, int B, int HW, int C, int groups, float eps,
                                                   void* stream) {
    ND_REQUIRE(x && gamma && beta && y && mean_rstd && mad && workspace, ND_E_BADARG, "nd_groupnorm_silu_train_forward: null pointer");
    if (int e = gs_check("nd_groupnorm_silu_train_forward", B, HW, C, groups)) return e;
    ND_REQUIRE(ldx >= C && ldy >= C && ldx % 4 == 0 && ldy % 4 == 0 && nd_aligned16(x) && nd_aligned16(y) && nd_aligned16(workspace) && nd_aligned16(mad),
               ND_E_ALIGN, "nd_groupnorm_silu_train_forward: strides must be multiples of 4 floats >= C, pointers 16-byte aligned");
    ND_REQUIRE(!res || (ldr >= C && ldr % 4 == 0 && nd_aligned16(res)), ND_E_ALIGN, "nd_groupnorm_silu_train_forward: the residual's stride / alignment");
    const int slots = gt_slots(HW);
    float* part = workspace;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gn_partials_kernel, dim3(B * slots), dim3(256), 0, st, x, ldx, x, ldx, 0, part, HW, C, slots);
    hipLaunchKernelGGL(gn_fwd_finalize_kernel, dim3(B * groups), dim3(256), 0, st, part, slots, HW, x, ldx, gamma, beta, mean_rstd, mad, C, groups, eps, scale_shift);
    const size_t total = (size_t)B * HW * (C / 4);
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(gs_apply_kernel, dim3(blocks), dim3(256), 0, st, x, ldx, x, ldx, mad, mad, y, ldy, B, HW, C, 0, res, ldr);
    return nd_launch_status("nd_groupnorm_silu_train_forward_f32");
}

extern "C" int nd_groupnorm_silu_train_backward_f32(const float* dy, int lddy, const float* x, int ldx, const float* gamma, const float* beta,
                                                    const float* scale_shift, const float* mean_rstd, const float* mad, float* dx, int lddx,
                                                    float* dgamma, float* dbeta, float* dscale_shift, float* workspace, int B, int HW, int C, int groups,
                                                    void* stream) {
    ND_REQUIRE(dy && x && gamma && beta && mean_rstd && mad && dx && dgamma && dbeta && workspace, ND_E_BADARG, "nd_groupnorm_silu_train_backward: null pointer");
    ND_REQUIRE((scale_shift == nullptr) == (dscale_shift == nullptr), ND_E_BADARG, "nd_groupnorm_silu_train_backward: scale_shift and dscale_shift go together");
    if (int e = gs_check("nd_groupnorm_silu_train_backward", B, HW, C, groups)) return e;
    ND_REQUIRE(lddy >= C && ldx >= C && lddx >= C && lddy % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0 && nd_aligned16(dy) && nd_aligned16(x) &&
               nd_aligned16(dx) && nd_aligned16(workspace) && nd_aligned16(mad), ND_E_ALIGN,
               "nd_groupnorm_silu_train_backward: strides must be multiples of 4 floats >= C, pointers 16-byte aligned");
    const int slots = gt_slots(HW);
    float* part = workspace;
    float* coef = workspace + (size_t)B * slots * C * 2 + (size_t)B * 3 * C;       // [B][2][C]: c1, c2
    float* ab = coef + (size_t)B * 2 * C;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gs_partials_kernel, dim3(B * slots), dim3(256), 0, st, dy, lddy, x, ldx, mad, part, HW, C, slots);
    hipLaunchKernelGGL(gs_bwd_finalize_kernel, dim3(B * groups), dim3(256), 0, st, part, slots, HW, mean_rstd, gamma, beta, scale_shift, coef, ab, dscale_shift,
                       C, groups);
    const size_t total = (size_t)B * HW * (C / 4);
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(gs_apply_kernel, dim3(blocks), dim3(256), 0, st, dy, lddy, x, ldx, mad, coef, dx, lddx, B, HW, C, 1, (const float*)nullptr, 0, ab, dgamma,
                       dbeta);
    return nd_launch_status("nd_groupnorm_silu_train_backward_f32");
}


// ---- per-pixel modulation + SiLU (ResnetBlock2, Diffusion_arch.py:173-196: scale / shift are MAPS from the position embedding):
//      y = silu(n (s + 1) + t) with n [N][C] (the GroupNorm's output) and map [N][2C] = s | t.  Backward: g = dy dsilu(m), dn = g (s + 1), ds = g n, dt = g.
namespace {

__global__ __launch_bounds__(256) void modsilu_kernel(const float* __restrict__ n, int ldn, const float* __restrict__ map, int ldm, const float* __restrict__ dy, int lddy,
                                                     float* __restrict__ out, int ldo, float* __restrict__ dmap, int lddm, size_t N, int C) {
    const int cq = C >> 2;
    const size_t total = N * cq;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % cq);
        const size_t p = i / cq;
        const f32x4 nv = nd_ld4(n + p * ldn + 4 * q), s1 = nd_ld4(map + p * ldm + 4 * q) + 1.0f, t = nd_ld4(map + p * ldm + C + 4 * q);
        const f32x4 m = nv * s1 + t;
        if (!dy) nd_st4(out + p * ldo + 4 * q, nd_silu4(m));
        else {
            const f32x4 g = gs_dsilu(nd_ld4(dy + p * lddy + 4 * q), m);
            nd_st4(out + p * ldo + 4 * q, g * s1);
            nd_st4(dmap + p * lddm + 4 * q, g * nv);
            nd_st4(dmap + p * lddm + C + 4 * q, g);
        }
    }
}

int modsilu_check(const char* who, const void* a, const void* b, const void* c, int ld0, int ld1, int ld2, int64_t N, int C) {
    ND_REQUIRE(a && b && c, ND_E_BADARG, "%s: null pointer", who);
    ND_REQUIRE(N > 0 && C > 0 && C % 4 == 0 && ld0 >= C && ld2 >= C && ld1 >= 2 * C && ld0 % 4 == 0 && ld1 % 4 == 0 && ld2 % 4 == 0, ND_E_SHAPE,
               "%s: C=%d (multiple of 4), strides multiples of 4 (tensor >= C, map >= 2C)", who, C);
    ND_REQUIRE(nd_aligned16(a) && nd_aligned16(b) && nd_aligned16(c), ND_E_ALIGN, "%s: pointers must be 16-byte aligned", who);
    return 0;
}

}  // namespace

extern "C" int nd_modulate_silu_forward_f32(const float* n, int ldn, const float* map, int ldm, float* y, int ldy, int64_t N, int C, void* stream) {
    if (int e = modsilu_check("nd_modulate_silu_forward", n, map, y, ldn, ldm, ldy, N, C)) return e;
    const size_t total = (size_t)N * (C / 4);
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(modsilu_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, ldn, map, ldm, (const float*)nullptr, 0, y, ldy, (float*)nullptr, 0, (size_t)N, C);
    return nd_launch_status("nd_modulate_silu_forward_f32");
}

extern "C" int nd_modulate_silu_backward_f32(const float* dy, int lddy, const float* n, int ldn, const float* map, int ldm, float* dn, int lddn, float* dmap, int lddm,
                                             int64_t N, int C, void* stream) {
    if (int e = modsilu_check("nd_modulate_silu_backward", n, map, dn, ldn, ldm, lddn, N, C)) return e;
    ND_REQUIRE(dy && dmap && lddy >= C && lddy % 4 == 0 && lddm >= 2 * C && lddm % 4 == 0 && nd_aligned16(dy) && nd_aligned16(dmap), ND_E_BADARG,
               "nd_modulate_silu_backward: dy / dmap pointer, stride or alignment");
    const size_t total = (size_t)N * (C / 4);
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(modsilu_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, ldn, map, ldm, dy, lddy, dn, lddn, dmap, lddm, (size_t)N, C);
    return nd_launch_status("nd_modulate_silu_backward_f32");
}

// ---- out[b][c] = sum over the HW tokens of x[b][p][c]: the gradient of a per-sample vector that was broadcast over the tokens (the one-token ISO
//      cross attention's output, AttnBlock: Diffusion_arch.py:435-437).  gn_partials_kernel's sum (mode 1, second component) per (sample, slot, channel),
//      then the slots in order (fp64): bitwise repeatable.  ATen's reduction of a (4, 65536, 64) tensor over its middle dimension takes 93 us.
namespace {
// block = 16 channels of a sample x 16 stripes of slots; the stripes meet in order (one thread per output and all slots: 40 us of dependent loads)
__global__ __launch_bounds__(256) void token_sum_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int slots, int B, int C) {
    __shared__ double red[16][16];
    const int cl = threadIdx.x & 15, stripe = threadIdx.x >> 4;
    const int per = (C + 15) / 16, b = blockIdx.x / per, c = (blockIdx.x % per) * 16 + cl;
    double acc = 0.0;
    if (c < C)
        for (int s = stripe; s < slots; s += 16) acc += (double)part[(((size_t)b * slots + s) * C + c) * 2 + 1];
    red[stripe][cl] = acc;
    __syncthreads();
    if (stripe == 0 && c < C) {
        for (int k = 1; k < 16; ++k) acc += red[k][cl];
        out[(size_t)b * C + c] = (float)acc;
    }
}
}  // namespace

extern "C" int64_t nd_token_sum_workspace_floats(int B, int HW, int C) {
    if (B <= 0 || HW <= 0 || C <= 0) return -1;
    return (int64_t)B * gt_slots(HW) * C * 2;
}

extern "C" int nd_token_sum_f32(const float* x, int ldx, float* out, float* workspace, int B, int HW, int C, void* stream) {
    ND_REQUIRE(x && out && workspace, ND_E_BADARG, "nd_token_sum: null pointer");
    ND_REQUIRE(B > 0 && HW > 0 && C > 0 && C % 4 == 0 && C <= 1024 && ldx >= C && ldx % 4 == 0, ND_E_SHAPE, "nd_token_sum: C=%d (multiple of 4, <= 1024), stride", C);
    ND_REQUIRE(nd_aligned16(x) && nd_aligned16(workspace), ND_E_ALIGN, "nd_token_sum: x and the workspace must be 16-byte aligned");
    const int slots = gt_slots(HW);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gn_partials_kernel, dim3(B * slots), dim3(256), 0, st, x, ldx, x, ldx, 1, workspace, HW, C, slots);
    hipLaunchKernelGGL(token_sum_reduce_kernel, dim3(B * nd_cdiv(C, 16)), dim3(256), 0, st, workspace, out, slots, B, C);
    return nd_launch_status("nd_token_sum_f32");
}
