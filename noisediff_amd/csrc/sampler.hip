// sampler.hip -- the reverse-diffusion update and its device-resident loop state.
//
// GaussianDiffusion.p_sample / p_mean_variance / q_posterior / model_predictions / ddim_sample
// (models/denoising_diffusion_pytorch.py:322-444) are ~12 elementwise ATen kernels and 7 tensor
// streams per step in the reference; here they are ONE pass: read x_t and the network output,
// (optionally) generate the Gaussian noise in registers, write x_{t-1} in place.
// All per-timestep scalars come from a table built on the host from the fp32 schedule buffers and
// indexed by a DEVICE step counter, so the captured step graph is identical for every step.
//
// coef[step][8]:
//   DDPM  {sqrt_ac, sqrt_1m_ac, sqrt_recip_ac, sqrt_recipm1_ac, post_c1, post_c2, exp(.5*post_logvar), t>0}
//   DDIM  {sqrt_ac, sqrt_1m_ac, sqrt_recip_ac, sqrt_recipm1_ac, sqrt(ac_next), c, sigma, last(time_next<0)}
#include "nd_common.h"

namespace {

struct Philox {
    // Philox4x32-10 (Salmon et al. 2011); restated in oracle/noisediff_oracle.py::philox4x32_10
    static __device__ __forceinline__ void round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
    }
    static __device__ __forceinline__ void gen(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            round(c, k0, k1);
            k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
    }
};

// four N(0,1) for quad `q` of sample `sample` at noise draw `step1` (0 = x_T, i+1 = i-th step)
__device__ __forceinline__ f32x4 philox_normal4(uint64_t seed, uint32_t sample, uint32_t step1, uint32_t q) {
    uint32_t c[4] = {q, sample, step1, 0u};
    Philox::gen(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float k = 1.0f / 4294967296.0f;
    const float u0 = ((float)c[0] + 0.5f) * k, u1 = ((float)c[1] + 0.5f) * k;
    const float u2 = ((float)c[2] + 0.5f) * k, u3 = ((float)c[3] + 0.5f) * k;
    const float r0 = sqrtf(-2.0f * logf(fminf(u0, 1.0f))), r1 = sqrtf(-2.0f * logf(fminf(u2, 1.0f)));
    float s0, c0, s1, c1;
    sincosf(6.283185307179586f * u1, &s0, &c0);
    sincosf(6.283185307179586f * u3, &s1, &c1);
    return (f32x4){r0 * c0, r0 * s0, r1 * c1, r1 * s1};
}

__global__ void begin_step_kernel(nd_sampler_state s) {
    const int st = *s.step;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < s.B) s.time_out[i] = st < s.n_steps ? (int64_t)s.t_cur[st] : 0;
}
__global__ void advance_kernel(nd_sampler_state s) { *s.step = *s.step + 1; }

__device__ __forceinline__ f32x4 clamp4(f32x4 v) {
    f32x4 r;
    r.x = fminf(fmaxf(v.x, -1.0f), 1.0f); r.y = fminf(fmaxf(v.y, -1.0f), 1.0f);
    r.z = fminf(fmaxf(v.z, -1.0f), 1.0f); r.w = fminf(fmaxf(v.w, -1.0f), 1.0f);
    return r;
}

template <bool DDIM>
__global__ __launch_bounds__(256) void step_kernel(float* __restrict__ x, const float* __restrict__ mo, const float* __restrict__ noise_base,
                                                   int64_t noise_stride, nd_sampler_state s, int objective, uint64_t seed, int64_t first_sample,
                                                   int B, int per_sample_q) {
    const int st = *s.step;
    if (st >= s.n_steps) return;
    const float* cf = s.coef + (size_t)st * 8;
    const float* noise = noise_base ? noise_base + (size_t)st * noise_stride : nullptr;
    const float c0 = cf[0], c1 = cf[1], c2 = cf[2], c3 = cf[3], c4 = cf[4], c5 = cf[5], c6 = cf[6], c7 = cf[7];
    if (s.rng) { seed = (uint64_t)s.rng[0]; first_sample = s.rng[1]; }
    const size_t total = (size_t)B * per_sample_q;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 xt = nd_ld4(x + i * 4), o = nd_ld4(mo + i * 4);
        f32x4 x0;
        if (objective == 2) x0 = c0 * xt - c1 * o;          // predict_start_from_v      :316-320
        else if (objective == 0) x0 = c2 * xt - c3 * o;     // predict_start_from_noise  :298-302
        else x0 = o;
        x0 = clamp4(x0);                                    // :361 (DDPM) / :351 (DDIM, clip_x_start)
        f32x4 z = {0, 0, 0, 0};
        const bool need_noise = DDIM ? (c7 == 0.0f && c6 != 0.0f) : (c7 != 0.0f);
        if (need_noise) {
            if (noise) z = nd_ld4(noise + i * 4);
            else {
                const uint32_t b = (uint32_t)(i / per_sample_q), q = (uint32_t)(i % per_sample_q);
                z = philox_normal4(seed, (uint32_t)(first_sample + b), (uint32_t)(st + 1), q);
            }
        }
        f32x4 r;
        if (DDIM) {
            const f32x4 eps = (c2 * xt - x0) / c3;          // predict_noise_from_start  :304-308
            r = (c7 != 0.0f) ? x0 : (x0 * c4 + c5 * eps + c6 * z);   // :422-437
        } else {
            r = (c4 * x0 + c5 * xt) + c6 * z;               // q_posterior :323-326, p_sample :372
        }
        nd_st4(x + i * 4, r);
    }
}

__global__ __launch_bounds__(256) void normal_kernel(float* __restrict__ out, uint64_t seed, int64_t first_sample, uint32_t step1, int B, int per_sample_q) {
    const size_t total = (size_t)B * per_sample_q;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t b = (uint32_t)(i / per_sample_q), q = (uint32_t)(i % per_sample_q);
        nd_st4(out + i * 4, philox_normal4(seed, (uint32_t)(first_sample + b), step1, q));
    }
}

int check_state(const nd_sampler_state* s, const char* who) {
    ND_REQUIRE(s && s->step && s->t_cur && s->coef && s->time_out, ND_E_BADARG, "%s: incomplete sampler state", who);
    ND_REQUIRE(s->n_steps > 0 && s->B > 0, ND_E_BADARG, "%s: n_steps/B must be positive", who);
    return 0;
}

template <bool DDIM>
int step_impl(float* x, const float* model_out, const float* noise, int64_t noise_stride, const nd_sampler_state* s, int objective, uint64_t seed,
              int64_t first_sample, int B, int HW, int C, void* stream, const char* who) {
    if (int e = check_state(s, who)) return e;
    ND_REQUIRE(x && model_out, ND_E_BADARG, "%s: null tensor", who);
    ND_REQUIRE(B > 0 && HW > 0 && C > 0 && ((size_t)HW * C) % 4 == 0, ND_E_SHAPE, "%s: HW*C must be a multiple of 4", who);
    ND_REQUIRE(objective >= 0 && objective <= 2, ND_E_BADARG, "%s: objective %d", who, objective);
    ND_REQUIRE(nd_aligned16(x) && nd_aligned16(model_out) && nd_aligned16(noise) && noise_stride % 4 == 0 && noise_stride >= 0, ND_E_ALIGN, "%s: alignment", who);
    const int psq = (int)(((size_t)HW * C) / 4);
    const size_t total = (size_t)B * psq;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(step_kernel<DDIM>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, model_out, noise, noise_stride, *s, objective, seed,
                       first_sample, B, psq);
    return nd_launch_status(who);
}

}  // namespace

extern "C" int nd_sampler_begin_step(const nd_sampler_state* s, void* stream) {
    if (int e = check_state(s, "nd_sampler_begin_step")) return e;
    hipLaunchKernelGGL(begin_step_kernel, dim3(nd_cdiv(s->B, 256)), dim3(256), 0, (hipStream_t)stream, *s);
    return nd_launch_status("nd_sampler_begin_step");
}

extern "C" int nd_sampler_advance(const nd_sampler_state* s, void* stream) {
    if (int e = check_state(s, "nd_sampler_advance")) return e;
    hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, *s);
    return nd_launch_status("nd_sampler_advance");
}

extern "C" int nd_sampler_step_ddpm_f32(float* x, const float* model_out, const float* noise, int64_t noise_step_stride,
                                        const nd_sampler_state* s, int objective,
                                        uint64_t seed, int64_t first_sample, int B, int HW, int C, void* stream) {
    return step_impl<false>(x, model_out, noise, noise_step_stride, s, objective, seed, first_sample, B, HW, C, stream, "nd_sampler_step_ddpm_f32");
}

extern "C" int nd_sampler_step_ddim_f32(float* x, const float* model_out, const float* noise, int64_t noise_step_stride,
                                        const nd_sampler_state* s, int objective,
                                        uint64_t seed, int64_t first_sample, int B, int HW, int C, void* stream) {
    ND_REQUIRE(s && s->t_next, ND_E_BADARG, "nd_sampler_step_ddim_f32: t_next table missing");
    return step_impl<true>(x, model_out, noise, noise_step_stride, s, objective, seed, first_sample, B, HW, C, stream, "nd_sampler_step_ddim_f32");
}

extern "C" int nd_philox_normal_f32(float* out, uint64_t seed, int64_t first_sample, int32_t step, int B, int HW, int C, void* stream) {
    ND_REQUIRE(out && B > 0 && HW > 0 && C > 0 && ((size_t)HW * C) % 4 == 0 && step >= -1, ND_E_BADARG, "nd_philox_normal: bad argument");
    ND_REQUIRE(nd_aligned16(out), ND_E_ALIGN, "nd_philox_normal: alignment");
    const int psq = (int)(((size_t)HW * C) / 4);
    const size_t total = (size_t)B * psq;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(normal_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, seed, first_sample, (uint32_t)(step + 1), B, psq);
    return nd_launch_status("nd_philox_normal_f32");
}
