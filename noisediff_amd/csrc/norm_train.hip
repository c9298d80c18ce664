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

constexpr int GT_SLOTS_MAX = 64;

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
__global__ __launch_bounds__(256) void gn_fwd_finalize_kernel(const float* __restrict__ part, int slots, int HW, const float* __restrict__ x, int ldx,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ mean_rstd, float* __restrict__ mad, int C, int G, float eps) {
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
        o[0] = fmean;  o[C] = rstd * gamma[c];  o[2 * C] = beta[c];
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
