// adam.hip -- torch.optim.Adam's update (the reference's optimizer: models/trainer_diffusion.py:94, models/denoising_diffusion_pytorch.py:22) for
// every parameter of a group in ONE launch (the optimizer step of the training path).  PyTorch's default on a GPU is the foreach
// form: ~10 element-wise passes over the 35 M parameters of the d = 64 network, 47 launches, 1.14 ms of a 28 ms step; this is one pass --
// read p, g, m, v, write p, m, v: 28 bytes per parameter, HBM-bound.
//
//   g' = g + weight_decay p                      (L2 form, as torch.optim.Adam; not AdamW)
//   m  = m + (g' - m)(1 - beta1)                 (Tensor.lerp_)
//   v  = v beta2 + (1 - beta2) g' g'
//   p  = p - step_size m / (sqrt(v) / sqrt(1 - beta2^t) + eps),   step_size = lr / (1 - beta1^t)
//
// step_size and sqrt(1 - beta2^t) come per parameter from the host (float64 there, as PyTorch computes them from the CPU step counters).
// Work is cut into chunks of ADAM_CHUNK elements of one parameter; the (parameter, offset) list is a device table the caller builds once.
// No reductions: bitwise repeatable.
#include "nd_common.h"

namespace {

constexpr int ADAM_CHUNK = 8192;                                     // elements per workgroup: 256 threads x 8 float4

__global__ __launch_bounds__(256) void adam_kernel(const nd_adam_item* __restrict__ items, const int2* __restrict__ chunks, float beta1, float beta2,
                                                   float eps, float weight_decay) {
    const int2 ck = chunks[blockIdx.x];
    const nd_adam_item it = items[ck.x];
    const long lo = (long)ck.y * ADAM_CHUNK;
    const long hi = lo + ADAM_CHUNK < it.n ? lo + ADAM_CHUNK : it.n;
    const float w1 = 1.0f - beta1, w2 = 1.0f - beta2;
    auto one = [&](float& p, float g, float& m, float& v) {
        g = fmaf(weight_decay, p, g);
        m = fmaf(g - m, w1, m);
        v = fmaf(w2 * g, g, v * beta2);
        const float denom = sqrtf(v) / it.bias2_sqrt + eps;
        p = p - it.step_size * (m / denom);
    };
    if (it.vec4 && hi - lo == ADAM_CHUNK) {                          // whole chunk, 16-byte aligned tensors
#pragma unroll 2
        for (long j = lo + 4 * threadIdx.x; j < hi; j += 4 * 256) {
            f32x4 p = nd_ld4(it.p + j), m = nd_ld4(it.m + j), v = nd_ld4(it.v + j);
            const f32x4 g = nd_ld4(it.g + j);
#pragma unroll
            for (int k = 0; k < 4; ++k) { float pk = p[k], mk = m[k], vk = v[k];  one(pk, g[k], mk, vk);  p[k] = pk;  m[k] = mk;  v[k] = vk; }
            nd_st4(it.p + j, p);  nd_st4(it.m + j, m);  nd_st4(it.v + j, v);
        }
    } else {
        for (long j = lo + threadIdx.x; j < hi; j += 256) {
            float p = it.p[j], m = it.m[j], v = it.v[j];
            one(p, it.g[j], m, v);
            it.p[j] = p;  it.m[j] = m;  it.v[j] = v;
        }
    }
}

// Capturable form (a whole training step as one graph: no host arithmetic per step).  Every parameter's step counter lives in device memory (PyTorch's
// capturable=True keeps it there too): adam_count_kernel adds one to each, adam_kernel<true> derives step_size and sqrt(1 - beta2^t) from it.
__global__ void adam_count_kernel(const nd_adam_item* __restrict__ items, int n_items) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_items) *items[i].step += 1.0f;
}

__global__ __launch_bounds__(256) void adam_cap_kernel(const nd_adam_item* __restrict__ items, const int2* __restrict__ chunks, float lr, float beta1, float beta2,
                                                       float eps, float weight_decay) {
    const int2 ck = chunks[blockIdx.x];
    nd_adam_item it = items[ck.x];
    const double t = (double)*it.step;                               // (already counted for this step)
    const float step_size = (float)((double)lr / (1.0 - pow((double)beta1, t))), bias2_sqrt = (float)sqrt(1.0 - pow((double)beta2, t));
    const long lo = (long)ck.y * ADAM_CHUNK;
    const long hi = lo + ADAM_CHUNK < it.n ? lo + ADAM_CHUNK : it.n;
    const float w1 = 1.0f - beta1, w2 = 1.0f - beta2;
    auto one = [&](float& p, float g, float& m, float& v) {
        g = fmaf(weight_decay, p, g);
        m = fmaf(g - m, w1, m);
        v = fmaf(w2 * g, g, v * beta2);
        const float denom = sqrtf(v) / bias2_sqrt + eps;
        p = p - step_size * (m / denom);
    };
    if (it.vec4 && hi - lo == ADAM_CHUNK) {
#pragma unroll 2
        for (long j = lo + 4 * threadIdx.x; j < hi; j += 4 * 256) {
            f32x4 p = nd_ld4(it.p + j), m = nd_ld4(it.m + j), v = nd_ld4(it.v + j);
            const f32x4 g = nd_ld4(it.g + j);
#pragma unroll
            for (int k = 0; k < 4; ++k) { float pk = p[k], mk = m[k], vk = v[k];  one(pk, g[k], mk, vk);  p[k] = pk;  m[k] = mk;  v[k] = vk; }
            nd_st4(it.p + j, p);  nd_st4(it.m + j, m);  nd_st4(it.v + j, v);
        }
    } else {
        for (long j = lo + threadIdx.x; j < hi; j += 256) {
            float p = it.p[j], m = it.m[j], v = it.v[j];
            one(p, it.g[j], m, v);
            it.p[j] = p;  it.m[j] = m;  it.v[j] = v;
        }
    }
}

}  // namespace

extern "C" int nd_adam_chunk_elements(void) { return ADAM_CHUNK; }

extern "C" int nd_adam_step_capturable_f32(const nd_adam_item* items_dev, int n_items, const int32_t* chunks_dev, int n_chunks, float lr, float beta1, float beta2,
                                           float eps, float weight_decay, void* stream) {
    ND_REQUIRE(items_dev && chunks_dev && n_items > 0 && n_chunks > 0, ND_E_BADARG, "nd_adam_step_capturable_f32: needs an item table and a chunk table in device memory");
    ND_REQUIRE(beta1 >= 0.0f && beta1 < 1.0f && beta2 >= 0.0f && beta2 < 1.0f && eps >= 0.0f && weight_decay >= 0.0f, ND_E_BADARG,
               "nd_adam_step_capturable_f32: beta1=%g, beta2=%g must lie in [0, 1), eps=%g and weight_decay=%g must not be negative", beta1, beta2, eps, weight_decay);
    hipLaunchKernelGGL(adam_count_kernel, dim3((unsigned)((n_items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, items_dev, n_items);
    hipLaunchKernelGGL(adam_cap_kernel, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream, items_dev, reinterpret_cast<const int2*>(chunks_dev), lr,
                       beta1, beta2, eps, weight_decay);
    return nd_launch_status("nd_adam_step_capturable_f32");
}

extern "C" int nd_adam_step_f32(const nd_adam_item* items_dev, int n_items, const int32_t* chunks_dev, int n_chunks, float beta1, float beta2, float eps,
                                float weight_decay, void* stream) {
    ND_REQUIRE(items_dev && chunks_dev && n_items > 0 && n_chunks > 0, ND_E_BADARG, "nd_adam_step_f32: needs an item table and a chunk table in device memory");
    ND_REQUIRE(beta1 >= 0.0f && beta1 < 1.0f && beta2 >= 0.0f && beta2 < 1.0f && eps >= 0.0f && weight_decay >= 0.0f, ND_E_BADARG,
               "nd_adam_step_f32: beta1=%g, beta2=%g must lie in [0, 1), eps=%g and weight_decay=%g must not be negative", beta1, beta2, eps, weight_decay);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream, items_dev, reinterpret_cast<const int2*>(chunks_dev), beta1,
                       beta2, eps, weight_decay);
    return nd_launch_status("nd_adam_step_f32");
}
