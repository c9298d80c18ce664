// Shared host/device helpers for libnoisediff_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include "../../include/noisediff_hip.h"

#define ND_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------- host side
void nd_set_error(const char* fmt, ...);

#define ND_REQUIRE(cond, code, ...)            \
    do {                                       \
        if (!(cond)) {                         \
            nd_set_error(__VA_ARGS__);         \
            return (code);                     \
        }                                      \
    } while (0)

static inline int nd_launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        nd_set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

// Launch-time facts are cached PER DEVICE ORDINAL and race-free: the net may be driven from several host threads on
// several devices at once (nn.DataParallel replicas, models/modules.py:81).
int nd_device_cus();                       // CU count of the CURRENT device (runtime.hip)
int nd_current_device();                   // hipGetDevice, 0 on failure
struct nd_device_once {                    // one bit per device: "this kernel's function attribute is set on that device"
    std::atomic<uint64_t> done{0};
    bool is_done(int dev) const { return dev >= 0 && dev < 64 && ((done.load(std::memory_order_acquire) >> dev) & 1u); }
    void mark(int dev) { if (dev >= 0 && dev < 64) done.fetch_or(1ull << dev, std::memory_order_release); }
};
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device and idempotent: two racing threads both set it, harmlessly
static inline int nd_reserve_lds(nd_device_once& once, const void* func, size_t lds, const char* who) {
    const int dev = nd_current_device();
    if (once.is_done(dev)) return 0;
    hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
        nd_set_error("%s: cannot reserve %zu bytes of LDS: %s", who, lds, hipGetErrorString(e));
        return (int)e;
    }
    once.mark(dev);
    return 0;
}

static inline bool nd_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
static inline int nd_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int nd_round_up(int a, int b) { return nd_cdiv(a, b) * b; }

// ---------------------------------------------------------------- device side
#ifdef __HIPCC__

// XCD-aware block remap (bijective for any grid size): hardware deals consecutive
// workgroup ids round-robin over the 8 XCDs, so ids {k, k+8, ...} share an L2.  Give
// each XCD one contiguous range of logical tiles so neighbouring tiles (shared conv
// halos, shared activation rows across N-tiles) hit the same L2.  Speed only.
__device__ __forceinline__ int nd_xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + idx;
}

// SiLU / exact-erf GELU with hardware reciprocal and exp (v_rcp_f32 / v_exp_f32, ~1 ulp) instead of the IEEE division
// sequence and libm erff (~10 and ~40 instructions, the latter with divergent range branches): these run once per
// activation in the conv staging pass and the pointwise epilogue and showed up as 25-45 % of those HBM-bound kernels.
// erf: Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7 absolute -- two orders below the fp32 parity budget.
__device__ __forceinline__ float nd_silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ float nd_erf(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float e = 1.0f - poly * __expf(-ax * ax);
    return copysignf(e, x);
}
__device__ __forceinline__ float nd_gelu(float v) { return 0.5f * v * (1.0f + nd_erf(v * 0.70710678118654752440f)); }

// GELU of two values on the packed fp32 pipe: the same Abramowitz-Stegun 7.1.26 erf as nd_erf (|error| < 1.5e-7), written as
// gelu(v) = v/2 + |v|/2 - |v|/2 * poly(t) * exp(-v^2/2), t = 1 / (1 + p |v| / sqrt 2): 12 packed instructions + 2 v_rcp + 2 v_exp for the pair
// against ~30 scalar ones (the fp32 MFMA shares the VALU: every instruction of an activation is matrix time).  For v < 0 the first two
// terms cancel exactly, so the tail keeps its relative accuracy.
__device__ __forceinline__ f32x2 nd_gelu2(f32x2 v) {
    const f32x2 av = {fabsf(v.x), fabsf(v.y)};
    const f32x2 one = {1.0f, 1.0f}, half = {0.5f, 0.5f};
    const f32x2 pk = {0.3275911f * 0.70710678118654752440f, 0.3275911f * 0.70710678118654752440f};
    const f32x2 d = __builtin_elementwise_fma(av, pk, one);
    const f32x2 t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const f32x2 a1 = {0.254829592f, 0.254829592f}, a2 = {-0.284496736f, -0.284496736f}, a3 = {1.421413741f, 1.421413741f},
                a4 = {-1.453152027f, -1.453152027f}, a5 = {1.061405429f, 1.061405429f};
    const f32x2 poly = t * __builtin_elementwise_fma(t, __builtin_elementwise_fma(t, __builtin_elementwise_fma(t, __builtin_elementwise_fma(t, a5, a4), a3), a2), a1);
    const f32x2 arg = (v * v) * f32x2{-0.5f * 1.44269504088896340736f, -0.5f * 1.44269504088896340736f};     // exp(-v^2 / 2) = 2^arg
    const f32x2 e = {__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)};
    const f32x2 h = av * half;
    return __builtin_elementwise_fma(-h, poly * e, __builtin_elementwise_fma(v, half, h));
}
__device__ __forceinline__ f32x4 nd_gelu4(f32x4 v) {
    const f32x2 lo = nd_gelu2(f32x2{v.x, v.y}), hi = nd_gelu2(f32x2{v.z, v.w});
    return f32x4{lo.x, lo.y, hi.x, hi.y};
}

__device__ __forceinline__ float nd_act(float v, int act) {
    if (act == ND_ACT_GELU) return nd_gelu(v);
    if (act == ND_ACT_SILU) return nd_silu(v);
    return v;
}

__device__ __forceinline__ f32x4 nd_ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void nd_st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

__device__ __forceinline__ f32x4 nd_silu4(f32x4 v) {
    f32x4 r;
    r.x = nd_silu(v.x); r.y = nd_silu(v.y); r.z = nd_silu(v.z); r.w = nd_silu(v.w);
    return r;
}

__device__ __forceinline__ f32x4 nd_leaky4(f32x4 v) {      // nn.LeakyReLU(negative_slope=0.2) == max(v, 0.2 v)
    const f32x4 w = v * 0.2f;
    return f32x4{fmaxf(v[0], w[0]), fmaxf(v[1], w[1]), fmaxf(v[2], w[2]), fmaxf(v[3], w[3])};
}

// Sum over the 16 lanes of a DPP row (lanes 16k..16k+15), result in every lane.  Four VALU adds with DPP
// operand swizzles (quad_perm xor1, quad_perm xor2, row_half_mirror, row_mirror) -- no LDS crossbar traffic,
// unlike __shfl_xor which lowers to ds_bpermute_b32 + a full lgkmcnt wait per step.
__device__ __forceinline__ float nd_row16_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));
    return v;
}

// The value of the first lane of each DPP row (lane 16k) in all 16 lanes of the row: one v_mov_b32 with row_newbcast:0 (gfx90a+), no LDS
// crossbar and no lane-index register (unlike __shfl(v, lane & 48) = ds_bpermute_b32).
__device__ __forceinline__ float nd_row16_first(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150, 0xF, 0xF, true));
}

// exact-fp32 matrix FMA: D(32x32) += A(32x2) * B(2x32); lane l gives A[l&31][l>>5], B[l>>5][l&31]
__device__ __forceinline__ f32x16 nd_zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.0f;
    return z;
}
__device__ __forceinline__ f32x16 nd_mfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// accumulator register r of lane l holds D[row][col]: col = l & 31, row = (r&3) + 8*(r>>2) + 4*(l>>5)
__device__ __forceinline__ int nd_acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

#endif  // __HIPCC__
