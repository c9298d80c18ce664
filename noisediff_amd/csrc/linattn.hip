// linattn.hip -- LinearAttention core (models/archs/Diffusion_arch.py:218-235), O(N * dh^2) per head.
//
//   q = softmax_d(q) * dh^-0.5      (over the 32 channels of a head, per pixel)
//   k = softmax_n(k)                (over all N pixels, per channel)
//   ctx[d][e] = sum_n k[d][n] v[e][n]        out[e][n] = sum_d ctx[d][e] q[d][n]
//
// The class is defined but not wired into NoiseDiffNet; it is provided as a standalone operator with
// its own golden (tests/golden modules.npz: mod.linear_attention).  Three small launches on NHWC qkv [B][N][3*heads*32]:
//   1. per (b, head, channel) column statistics of k over N (online max / sum-exp, fixed reduction tree);
//   2. per (b, head, chunk of N): partial ctx on the fp32 matrix pipe (K dimension = pixels);
//   3. per 32 pixels: softmax of q in registers, ctx reduced over chunks, out = ctx^T q on the matrix pipe.
#include "nd_common.h"

namespace {

constexpr int DH = 32;
constexpr int CHUNK = 2048;      // pixels per partial-context workgroup

// 1. kstat[b][h][d] = {max_n k, sum_n exp(k - max)}
__global__ __launch_bounds__(256) void la_kstat_kernel(const float* __restrict__ qkv, int ldq, float* __restrict__ kstat, int N, int heads) {
    __shared__ float sm[32][DH], ss[32][DH];
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int q4 = (tid & 7) * 4, r0 = tid >> 3;
    const int hid = heads * DH;
    const float* base = qkv + (size_t)b * N * ldq + hid + h * DH + q4;
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, s = {0, 0, 0, 0};
    for (int n = r0; n < N; n += 32) {
        const f32x4 v = nd_ld4(base + (size_t)n * ldq);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float mn = fmaxf(m[e], v[e]);
            s[e] = s[e] * __expf(m[e] - mn) + __expf(v[e] - mn);
            m[e] = mn;
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { sm[r0][q4 + e] = m[e]; ss[r0][q4 + e] = s[e]; }
    __syncthreads();
    if (tid < DH) {
        float M = -INFINITY, S = 0.0f;
        for (int r = 0; r < 32; ++r) {
            const float mr = sm[r][tid], sr = ss[r][tid];
            if (sr > 0.0f) {
                const float mn = fmaxf(M, mr);
                S = S * __expf(M - mn) + sr * __expf(mr - mn);
                M = mn;
            }
        }
        float* o = kstat + (((size_t)b * heads + h) * DH + tid) * 2;
        o[0] = M; o[1] = S;
    }
}

// 2. partial[b][h][chunk][d][e] = sum_{n in chunk} softmax_n(k)[d][n] * v[e][n]
__global__ __launch_bounds__(256) void la_context_kernel(const float* __restrict__ qkv, int ldq, const float* __restrict__ kstat,
                                                         float* __restrict__ partial, int N, int heads, int chunks) {
    __shared__ float red[4][DH * DH];
    const int c = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
    const int hid = heads * DH;
    const float* ks = kstat + (((size_t)b * heads + h) * DH + col) * 2;
    const float kmax = ks[0], kinv = 1.0f / ks[1];
    const float* kb = qkv + (size_t)b * N * ldq + hid + h * DH + col;         // lane = channel d (A operand rows)
    const float* vb = qkv + (size_t)b * N * ldq + 2 * hid + h * DH + col;     // lane = channel e (B operand cols)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int n_end = min(N, (c + 1) * CHUNK);
    for (int n0 = c * CHUNK + wave * 2; n0 < n_end; n0 += 8) {                // k-step = 2 pixels (lane half picks one); wave-uniform trip count
        const int n = n0 + half;
        const bool ok = n < n_end;
        const size_t nn = ok ? n : n0;
        const float kv = ok ? __expf(kb[nn * ldq] - kmax) * kinv : 0.0f;
        const float vv = ok ? vb[nn * ldq] : 0.0f;
        acc = nd_mfma(kv, vv, acc);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][nd_acc_row(r, lane) * DH + col] = acc[r];
    __syncthreads();
    float* o = partial + (((size_t)b * heads + h) * chunks + c) * DH * DH;
    for (int i = tid; i < DH * DH; i += 256) o[i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}

// 3. out[n][h*32 + e] = sum_d ctx[d][e] * softmax_d(q)[n][d] * scale
__global__ __launch_bounds__(256) void la_output_kernel(const float* __restrict__ qkv, int ldq, const float* __restrict__ partial,
                                                        float* __restrict__ out, int ldo, int N, int heads, int chunks, float scale) {
    __shared__ float ctx[DH * (DH + 1)];
    const int h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
    const float* pb = partial + ((size_t)b * heads + h) * chunks * DH * DH;
    for (int i = tid; i < DH * DH; i += 256) {
        float s = 0.0f;
        for (int c = 0; c < chunks; ++c) s += pb[(size_t)c * DH * DH + i];     // fixed order over chunks
        ctx[(i / DH) * (DH + 1) + (i % DH)] = s;
    }
    __syncthreads();
    const int n = (blockIdx.x * 4 + wave) * 32 + col;        // A operand row = pixel
    const bool ok = n < N;
    const float* qp = qkv + ((size_t)b * N + (ok ? n : 0)) * ldq + h * DH + 16 * half;
    float q[16];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4 v = nd_ld4(qp + 4 * j);
        q[4 * j] = v.x; q[4 * j + 1] = v.y; q[4 * j + 2] = v.z; q[4 * j + 3] = v.w;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) mx = fmaxf(mx, q[j]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j) { q[j] = __expf(q[j] - mx); sum += q[j]; }
    sum += __shfl_xor(sum, 32);
    const float norm = scale / sum;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j)        // k-step j pairs channels d = j (half 0) and 16 + j (half 1)
        acc = nd_mfma(q[j] * norm, ctx[(16 * half + j) * (DH + 1) + col], acc);
    // acc: rows = pixels of this wave's block, cols = e
    const int nb = (blockIdx.x * 4 + wave) * 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int p = nb + nd_acc_row(r, lane);
        if (p < N) out[((size_t)b * N + p) * ldo + h * DH + col] = acc[r];
    }
}

}  // namespace

extern "C" int64_t nd_linear_attention_workspace_floats(int B, int N, int heads) {
    if (B <= 0 || N <= 0 || heads <= 0) return ND_E_BADARG;
    const int chunks = nd_cdiv(N, CHUNK);
    return (int64_t)B * heads * (DH * 2 + (int64_t)chunks * DH * DH);
}

extern "C" int nd_linear_attention_f32(const float* qkv, int ld_qkv, float* out, int ld_out, float* workspace, int B, int N, int heads,
                                       int dh, void* stream) {
    ND_REQUIRE(qkv && out && workspace, ND_E_BADARG, "nd_linear_attention: null pointer");
    ND_REQUIRE(B > 0 && N > 0 && heads > 0, ND_E_BADARG, "nd_linear_attention: non-positive size");
    ND_REQUIRE(dh == DH, ND_E_SHAPE, "nd_linear_attention: dim_head=%d (only 32 is built)", dh);
    ND_REQUIRE(ld_qkv >= 3 * heads * dh && ld_qkv % 4 == 0 && ld_out >= heads * dh, ND_E_SHAPE, "nd_linear_attention: strides");
    ND_REQUIRE(nd_aligned16(qkv), ND_E_ALIGN, "nd_linear_attention: alignment");
    ND_REQUIRE(B <= 65535 && heads <= 65535, ND_E_SHAPE, "nd_linear_attention: grid too large");
    const int chunks = nd_cdiv(N, CHUNK);
    float* kstat = workspace;
    float* partial = workspace + (size_t)B * heads * DH * 2;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(la_kstat_kernel, dim3(heads, B), dim3(256), 0, st, qkv, ld_qkv, kstat, N, heads);
    hipLaunchKernelGGL(la_context_kernel, dim3(chunks, heads, B), dim3(256), 0, st, qkv, ld_qkv, kstat, partial, N, heads, chunks);
    hipLaunchKernelGGL(la_output_kernel, dim3(nd_cdiv(N, 128), heads, B), dim3(256), 0, st, qkv, ld_qkv, partial, out, ld_out, N, heads,
                       chunks, 1.0f / sqrtf((float)dh));
    return nd_launch_status("nd_linear_attention_f32");
}
