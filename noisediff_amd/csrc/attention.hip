// attention.hip -- full self-attention core for the optional mid-block Attention
// (models/archs/Diffusion_arch.py:237-266 + models/attend.py:101-116; BASELINE config 4).
//
// out[b, i, h*dh + :] = softmax_j(q_i . k_j / sqrt(dh)) v_j, flash-style (no N x N matrix in memory),
// on the exact-fp32 matrix pipe.  One workgroup = 4 waves = 128 queries of one (sample, head);
// each wave owns 32 queries and walks the keys in tiles of 32:
//   S^T = K Q^T   (keys on accumulator rows, queries on the lane column)  -> the softmax reduction
//                 over keys is a register reduction plus ONE cross-half shuffle;
//   O^T += V^T P^T: the probabilities are consumed straight from the accumulator registers as the
//                 B operand (register r of lane half h is key row (r&3)+8(r>>2)+4h: one k-step), so
//                 P never moves through LDS.
// K and V tiles are staged once per workgroup in LDS (K rows padded to 36 floats for b128 reads).
#include "nd_common.h"

namespace {

constexpr int DH = 32, KT = 64, LDK = DH + 4;

__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ qkv, int ldq, float* __restrict__ out, int ldo,
                                                        int N, int heads, float scale) {
    __shared__ __attribute__((aligned(16))) float Ks[KT * LDK];
    __shared__ __attribute__((aligned(16))) float Vs[KT * DH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
    const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int hid = heads * DH;
    const float* base = qkv + (size_t)b * N * ldq;
    const int qi = qt * 128 + wave * 32 + col;          // this lane's query
    // Q fragment: lane (query, half) holds q[16*half + s], s = 0..15, pre-scaled
    float qf[16];
    {
        const bool qv = qi < N;
        const float* qp = base + (size_t)(qv ? qi : 0) * ldq + h * DH + 16 * half;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 v = qv ? nd_ld4(qp + 4 * j) : (f32x4){0, 0, 0, 0};
            qf[4 * j] = v.x * scale; qf[4 * j + 1] = v.y * scale; qf[4 * j + 2] = v.z * scale; qf[4 * j + 3] = v.w * scale;
        }
    }
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;

    for (int k0 = 0; k0 < N; k0 += KT) {
        __syncthreads();
        // stage K and V tiles: 64 keys x 32 floats each = 512 quads per tensor, 256 threads x 2
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = tid + it * 256, key = idx >> 3, q4 = (idx & 7) * 4;
            f32x4 kv = {0, 0, 0, 0}, vv = {0, 0, 0, 0};
            if (k0 + key < N) {
                const float* rp = base + (size_t)(k0 + key) * ldq + h * DH + q4;
                kv = nd_ld4(rp + hid);
                vv = nd_ld4(rp + 2 * hid);
            }
            nd_st4(&Ks[key * LDK + q4], kv);
            nd_st4(&Vs[key * DH + q4], vv);
        }
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < KT / 32; ++sub) {
            if (k0 + sub * 32 >= N) break;
            // S^T tile: rows = keys, cols = queries
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.0f;
            const float* kp = &Ks[(sub * 32 + col) * LDK + 16 * half];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 kq = nd_ld4(kp + 4 * j);
                st = nd_mfma(kq.x, qf[4 * j], st);
                st = nd_mfma(kq.y, qf[4 * j + 1], st);
                st = nd_mfma(kq.z, qf[4 * j + 2], st);
                st = nd_mfma(kq.w, qf[4 * j + 3], st);
            }
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + sub * 32 + nd_acc_row(r, lane);
                if (key >= N) st[r] = -INFINITY;
                mx = fmaxf(mx, st[r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run, mx);          // finite: every tile has >= 1 valid key
            const float alpha = __expf(m_run - m_new);
            float ps = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                st[r] = __expf(st[r] - m_new);
                ps += st[r];
            }
            ps += __shfl_xor(ps, 32);
            l_run = l_run * alpha + ps;
            m_run = m_new;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] *= alpha;
            // O^T[dh][query] += sum_key V[key][dh] * P^T[key][query]; k-step r pairs keys row(r,0), row(r,1)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float vf = Vs[(sub * 32 + nd_acc_row(r, lane)) * DH + col];
                o = nd_mfma(vf, st[r], o);
            }
        }
    }
    if (qi < N) {
        const float inv = 1.0f / l_run;
        float* op = out + ((size_t)b * N + qi) * ldo + h * DH;
#pragma unroll
        for (int r = 0; r < 16; ++r) op[nd_acc_row(r, lane)] = o[r] * inv;
    }
}

}  // namespace

extern "C" int nd_attention_mfma_f32(const float* qkv, int ld_qkv, float* out, int ld_out, int B, int N, int heads, int dh, void* stream) {
    ND_REQUIRE(qkv && out, ND_E_BADARG, "nd_attention_mfma: null pointer");
    ND_REQUIRE(B > 0 && N > 0 && heads > 0, ND_E_BADARG, "nd_attention_mfma: non-positive size");
    ND_REQUIRE(dh == 32, ND_E_SHAPE, "nd_attention_mfma: dim_head=%d (only 32 is built)", dh);
    ND_REQUIRE(ld_qkv >= 3 * heads * dh && ld_qkv % 4 == 0 && ld_out >= heads * dh, ND_E_SHAPE, "nd_attention_mfma: strides");
    ND_REQUIRE(nd_aligned16(qkv), ND_E_ALIGN, "nd_attention_mfma: alignment");
    ND_REQUIRE(B <= 65535 && heads <= 65535, ND_E_SHAPE, "nd_attention_mfma: grid too large");
    hipLaunchKernelGGL(attention_kernel, dim3(nd_cdiv(N, 128), heads, B), dim3(256), 0, (hipStream_t)stream, qkv, ld_qkv, out, ld_out,
                       N, heads, 1.0f / sqrtf((float)dh));
    return nd_launch_status("nd_attention_mfma_f32");
}
