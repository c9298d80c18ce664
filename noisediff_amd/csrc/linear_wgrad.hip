// linear_wgrad.hip -- weight and bias gradient of a token Linear / 1x1 convolution on NHWC fp32 (third training slice):
//
//   dW[co][ci] = sum over the N = B*H*W tokens of dY[p][co] * X[p][ci],      db[co] = sum over the tokens of dY[p][co]
//
// the backward of res_conv, Mlp.fc1 / fc2, FeedForward's Linears, proj_out and the attention projections under GaussianDiffusion.p_losses
// (models/archs/Diffusion_arch.py:156,345-347,410-419,432; models/denoising_diffusion_pytorch.py:481-531).  A GEMM whose reduction runs over
// 10^5..10^6 tokens with a 64..1024-wide output: library GEMMs take 0.4-0.5 ms for what is 0.2 GB of HBM traffic (rocBLAS picks 16x16 /
// 32x64 macro tiles, torch profiler in tools/train_step_bench.py).  Here it is the one-tap form of conv3x3_wgrad.hip: a workgroup owns a
// 64 x 64 block of dW -- four waves x one 32 x 32 accumulator of the exact-fp32 MFMA -- and walks its share of the 128-token chunks,
// both operand chunks staged in LDS (64 KB; two workgroups per CU hide each other's staging); the bias gradient falls out of the staged
// dY chunk.  The split over chunks depends on the shape only, partial sums meet in a second kernel in a fixed order: bitwise repeatable.
#include "nd_common.h"

namespace {

constexpr int LW_PK = 128, LW_CB = 64;                                   // tokens per chunk, channel block
constexpr int LW_TARGET_WGS = 512;                                       // two per CU of an MI355X; fixed: the summation order must not depend on the device

struct LwArgs {
    const float* x; const float* dy; float* ws; float* wsb;
    int ldx, ldy, cin, cout, n_co, n_ci, S, want_bias;
    long N, chunks;
};

__global__ __launch_bounds__(256, 2) void linear_wgrad_kernel(const LwArgs a) {
    __shared__ __attribute__((aligned(16))) float dYs[LW_PK * LW_CB];
    __shared__ __attribute__((aligned(16))) float Xs[LW_PK * LW_CB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int co_w = 32 * (wave >> 1), ci_w = 32 * (wave & 1);
    int bid = blockIdx.x;
    const int s = bid % a.S;  bid /= a.S;
    const int cib = bid % a.n_ci, cob = bid / a.n_ci;
    const int co0 = cob * LW_CB, ci0 = cib * LW_CB;

    f32x16 acc = nd_zero16();
    float bsum = 0.0f;                                                   // bias gradient: thread (cout tid & 63, token quarter tid >> 6) of the cib == 0 workgroups
    const bool do_bias = a.want_bias && cib == 0;
    const int q = tid & 15, r0 = tid >> 4;                               // staging: channel quad, first row (16 rows per pass, 8 passes)
    const int cy = co0 + 4 * q, cx = ci0 + 4 * q;
    const bool cy_ok = cy < a.cout, cx_ok = cx < a.cin;
    const float* dyb = a.dy + (cy_ok ? cy : 0);
    const float* xb = a.x + (cx_ok ? cx : 0);
    const f32x4 zero = {0, 0, 0, 0};

    for (long chunk = s; chunk < a.chunks; chunk += a.S) {
        const long p0 = chunk * LW_PK;
        __syncthreads();                                                 // the previous chunk's operands have been consumed
        {   // every load unconditional (clamped row, select afterwards), all 16 in flight before the first LDS write
            f32x4 vy[8], vx[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const long p = min(p0 + r0 + 16 * j, a.N - 1);
                vy[j] = nd_ld4(dyb + (size_t)p * a.ldy);
                vx[j] = nd_ld4(xb + (size_t)p * a.ldx);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = r0 + 16 * j;
                const bool in = p0 + r < a.N;
                nd_st4(dYs + r * LW_CB + 4 * q, in && cy_ok ? vy[j] : zero);
                nd_st4(Xs + r * LW_CB + 4 * q, in && cx_ok ? vx[j] : zero);
            }
        }
        __syncthreads();
        const float* ap = dYs + half * LW_CB + co_w + col;
        const float* bp = Xs + half * LW_CB + ci_w + col;
#pragma unroll 16
        for (int pp = 0; pp < LW_PK / 2; ++pp) acc = nd_mfma(ap[2 * pp * LW_CB], bp[2 * pp * LW_CB], acc);
        if (do_bias) {
            const float* cp = dYs + (tid >> 6) * (LW_PK / 4) * LW_CB + (tid & 63);
#pragma unroll 8
            for (int r = 0; r < LW_PK / 4; ++r) bsum += cp[r * LW_CB];
        }
    }
    // ---- this workgroup's partial sums: ws[s][co][ci]; the bias partial wsb[s][co] through LDS (four token quarters, fixed order)
    const int coP = a.n_co * LW_CB, ciP = a.n_ci * LW_CB;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        a.ws[((size_t)s * coP + co0 + co_w + nd_acc_row(r, lane)) * ciP + ci0 + ci_w + col] = acc[r];
    if (do_bias) {
        __syncthreads();
        dYs[tid] = bsum;
        __syncthreads();
        if (tid < 64) a.wsb[(size_t)s * coP + co0 + tid] = dYs[tid] + dYs[64 + tid] + dYs[128 + tid] + dYs[192 + tid];
    }
}

// dW[co][ci] = sum over the S partials, db[co] likewise.  A workgroup owns 32 consecutive elements of the partial blocks; its eight
// 32-thread stripes add the slots s = stripe, stripe + 8, ... (eight loads in flight each) and the stripes then meet in stripe order:
// a fixed order, eight times shallower than one thread per element (which made this pass slower than the gradient kernel itself).
__global__ __launch_bounds__(256) void linear_wgrad_reduce_kernel(const float* __restrict__ ws, const float* __restrict__ wsb, float* __restrict__ dw,
                                                                  float* __restrict__ db, int S, int cin, int cout, int coP, int ciP) {
    __shared__ float red[8][32];
    const size_t block = (size_t)coP * ciP, total = block + coP;         // the bias partials follow the weight partials element-wise
    const int o = threadIdx.x & 31, stripe = threadIdx.x >> 5;
    for (size_t j0 = (size_t)blockIdx.x * 32; j0 < total; j0 += (size_t)gridDim.x * 32) {
        const size_t j = j0 + o;
        const bool bias = j >= block;
        const float* p = bias ? wsb + (j - block) : ws + j;
        const size_t stride = bias ? (size_t)coP : block;
        float sum = 0.0f;
        if (j < total && (!bias || db)) {
            int s = stripe;
            for (; s + 56 < S; s += 64) {
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = p[(size_t)(s + 8 * k) * stride];
#pragma unroll
                for (int k = 0; k < 8; ++k) sum += v[k];
            }
            for (; s < S; s += 8) sum += p[(size_t)s * stride];
        }
        __syncthreads();
        red[stripe][o] = sum;
        __syncthreads();
        if (stripe == 0 && j < total) {
#pragma unroll
            for (int k = 1; k < 8; ++k) sum += red[k][o];
            if (bias) {
                const int co = (int)(j - block);
                if (db && co < cout) db[co] = sum;
            } else {
                const int ci = (int)(j % ciP), co = (int)(j / ciP);
                if (co < cout && ci < cin) dw[(size_t)co * cin + ci] = sum;
            }
        }
    }
}

void lw_plan(long N, int cin, int cout, LwArgs& a) {
    a.N = N; a.cin = cin; a.cout = cout;
    a.n_co = nd_cdiv(cout, LW_CB);
    a.n_ci = nd_cdiv(cin, LW_CB);
    a.chunks = (N + LW_PK - 1) / LW_PK;
    long S = LW_TARGET_WGS / ((long)a.n_co * a.n_ci);
    if (S < 1) S = 1;
    if (S > a.chunks) S = a.chunks;
    a.S = (int)S;
}

}  // namespace

extern "C" int64_t nd_linear_wgrad_workspace_floats(int64_t N, int cin, int cout) {
    if (N <= 0 || cin <= 0 || cout <= 0) return -1;
    LwArgs a;
    lw_plan(N, cin, cout, a);
    return (int64_t)a.S * a.n_co * LW_CB * (a.n_ci * LW_CB + 1);
}

extern "C" int nd_linear_wgrad_f32(const float* x, int ldx, const float* dy, int ldy, float* dw, float* dbias, float* workspace,
                                   int64_t N, int cin, int cout, void* stream) {
    ND_REQUIRE(x && dy && dw && workspace, ND_E_BADARG, "nd_linear_wgrad: null pointer");
    ND_REQUIRE(N > 0 && cin > 0 && cout > 0, ND_E_BADARG, "nd_linear_wgrad: non-positive size");
    ND_REQUIRE(cin % 4 == 0 && cout % 4 == 0 && ldx >= cin && ldy >= cout && ldx % 4 == 0 && ldy % 4 == 0, ND_E_SHAPE,
               "nd_linear_wgrad: cin=%d, cout=%d and the token strides must be multiples of 4", cin, cout);
    ND_REQUIRE(nd_aligned16(x) && nd_aligned16(dy) && nd_aligned16(workspace), ND_E_ALIGN, "nd_linear_wgrad: x, dy and the workspace must be 16-byte aligned");
    LwArgs a;
    lw_plan(N, cin, cout, a);
    a.x = x; a.dy = dy; a.ldx = ldx; a.ldy = ldy; a.want_bias = dbias != nullptr;
    a.ws = workspace;
    a.wsb = workspace + (size_t)a.S * a.n_co * LW_CB * a.n_ci * LW_CB;
    const long wgs = (long)a.n_co * a.n_ci * a.S;
    ND_REQUIRE(wgs < (1L << 31), ND_E_SHAPE, "nd_linear_wgrad: grid too large");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(linear_wgrad_kernel, dim3((unsigned)wgs), dim3(256), 0, st, a);
    if (int e = nd_launch_status("nd_linear_wgrad_f32")) return e;
    const size_t total = (size_t)a.n_co * LW_CB * (a.n_ci * LW_CB + 1);
    const int blocks = (int)((total + 31) / 32 < 8192 ? (total + 31) / 32 : 8192);
    hipLaunchKernelGGL(linear_wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, a.ws, a.wsb, dw, dbias, a.S, cin, cout, a.n_co * LW_CB, a.n_ci * LW_CB);
    return nd_launch_status("nd_linear_wgrad_f32 (reduce)");
}
