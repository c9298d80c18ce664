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
//              finalize  dgamma, dbeta; per (sample, channel) c0 = rstd gamma, c1 = -rstd^2 S1 / N, c2 = mean rstd^2 S1 / N - rstd S2 / N
//                        with S1 = sum_c gamma_c rstd (sum dy x - mean sum dy), S2 = sum_c gamma_c sum dy over the group  [gn_bwd_finalize_kernel]
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

// one workgroup (64 threads) per (sample, group)
__global__ __launch_bounds__(64) void gn_fwd_finalize_kernel(const float* __restrict__ part, int slots, int HW, const float* __restrict__ x, int ldx,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ mean_rstd, float* __restrict__ mad, int C, int G, float eps) {
    const int b = blockIdx.x / G, g = blockIdx.x % G, cpg = C / G, lane = threadIdx.x;
    double S = 0.0, Q2 = 0.0;
    for (int i = lane; i < cpg; i += 64) {                     // a channel per lane: its slots in order, then un-shift by its pivot
        const int c = g * cpg + i;
        double s1 = 0.0, s2 = 0.0;
        for (int s = 0; s < slots; ++s) {
            const float* o = part + (((size_t)b * slots + s) * C + c) * 2;
            s1 += (double)o[0];  s2 += (double)o[1];
        }
        const double p = (double)x[(size_t)b * HW * ldx + c], n = (double)HW;
        S += s1 + n * p;
        Q2 += s2 + 2.0 * p * s1 + n * p * p;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { S += __shfl_xor(S, o); Q2 += __shfl_xor(Q2, o); }
    const double N = (double)cpg * (double)HW;
    const double mean = S / N;
    double var = Q2 / N - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps)), fmean = (float)mean;
    if (lane == 0) { mean_rstd[((size_t)b * G + g) * 2] = fmean; mean_rstd[((size_t)b * G + g) * 2 + 1] = rstd; }
    for (int i = lane; i < cpg; i += 64) {
        const int c = g * cpg + i;
        float* o = mad + (size_t)b * 3 * C + c;
        o[0] = fmean;  o[C] = rstd * gamma[c];  o[2 * C] = beta[c];
    }
}

// one workgroup (64 threads) per group; the samples in order (dgamma / dbeta sum over them)
__global__ __launch_bounds__(64) void gn_bwd_finalize_kernel(const float* __restrict__ part, int slots, int HW, const float* __restrict__ mean_rstd,
                                                            const float* __restrict__ gamma, float* __restrict__ coef,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int C, int G) {
    const int g = blockIdx.x, cpg = C / G, lane = threadIdx.x;
    const double N = (double)cpg * (double)HW;
    // lane i owns channels i, i + 64, ... of the group (cpg <= 64 * 8 per the host check)
    double dg[8], db[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) dg[k] = db[k] = 0.0;
    for (int b = 0; b < B; ++b) {
        const double mean = (double)mean_rstd[((size_t)b * G + g) * 2], rstd = (double)mean_rstd[((size_t)b * G + g) * 2 + 1];
        double A[8], Bs[8], S1 = 0.0, S2 = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = lane + 64 * k;
            A[k] = Bs[k] = 0.0;
            if (i < cpg) {
                const int c = g * cpg + i;
                double sdx = 0.0, sd = 0.0;
                for (int s = 0; s < slots; ++s) {
                    const float* o = part + (((size_t)b * slots + s) * C + c) * 2;
                    sdx += (double)o[0];  sd += (double)o[1];
                }
                A[k] = rstd * (sdx - mean * sd);               // sum over pixels of dy * xhat
                Bs[k] = sd;
                dg[k] += A[k];  db[k] += Bs[k];
                S1 += (double)gamma[c] * A[k];  S2 += (double)gamma[c] * Bs[k];
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { S1 += __shfl_xor(S1, o); S2 += __shfl_xor(S2, o); }
        const float c1 = (float)(-rstd * rstd * S1 / N), c2 = (float)(mean * rstd * rstd * S1 / N - rstd * S2 / N);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = lane + 64 * k;
            if (i < cpg) {
                const int c = g * cpg + i;
                float* o = coef + (size_t)b * 3 * C + c;
                o[0] = (float)rstd * gamma[c];  o[C] = c1;  o[2 * C] = c2;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = lane + 64 * k;
        if (i < cpg) { dgamma[g * cpg + i] = (float)dg[k]; dbeta[g * cpg + i] = (float)db[k]; }
    }
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
    hipLaunchKernelGGL(gn_fwd_finalize_kernel, dim3(B * groups), dim3(64), 0, st, part, slots, HW, x, ldx, gamma, beta, mean_rstd, mad, C, groups, eps);
    const size_t total = (size_t)B * HW * (C / 4);
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(affine3_kernel, dim3(blocks), dim3(256), 0, st, x, ldx, x, ldx, mad, y, ldy, B, HW, C, 0);
    return nd_launch_status("nd_groupnorm_train_forward_f32");
}

extern "C" int64_t nd_groupnorm_train_workspace_floats(int B, int HW, int C) {
    if (B <= 0 || HW <= 0 || C <= 0) return -1;
    return (int64_t)B * gt_slots(HW) * C * 2 + (int64_t)B * 3 * C;
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
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gn_partials_kernel, dim3(B * slots), dim3(256), 0, st, dy, lddy, x, ldx, 1, part, HW, C, slots);
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(groups), dim3(64), 0, st, part, slots, HW, mean_rstd, gamma, coef, dgamma, dbeta, B, C, groups);
    const size_t total = (size_t)B * HW * (C / 4);
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(affine3_kernel, dim3(blocks), dim3(256), 0, st, dy, lddy, x, ldx, coef, dx, lddx, B, HW, C, 1);
    return nd_launch_status("nd_groupnorm_train_backward_f32");
}
