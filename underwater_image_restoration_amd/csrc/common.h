// Shared device/host helpers for libuwir_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/uwir.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define UWIR_WAVE 64

int uwir_set_error(int code, const char* fmt, ...);

#define UWIR_CHECK(cond, ...)                                   \
    do {                                                        \
        if (!(cond)) return uwir_set_error(UWIR_EINVAL, __VA_ARGS__); \
    } while (0)

#define UWIR_LAUNCH_CHECK(name)                                                          \
    do {                                                                                 \
        hipError_t e_ = hipGetLastError();                                               \
        if (e_ != hipSuccess) return uwir_set_error(UWIR_ELAUNCH, "%s: %s", name, hipGetErrorString(e_)); \
    } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t ceil_div_sz(size_t a, size_t b) { return (a + b - 1) / b; }

// erf to 3e-7 absolute in ~16 branch-free VALU ops (libm's erff is ~3x that and divergent; GELU sits in the epilogue
// of the widest FFN GEMMs, where it measured +0.12 ms on a 0.15 ms launch).  Abramowitz-Stegun 7.1.26:
// erfc(|x|) = exp(-x^2) * poly(1 / (1 + 0.3275911 |x|)), |error| <= 1.5e-7 in exact arithmetic.  GELU multiplies by
// x/2, so the absolute error of the activation is <= 1.5e-7 |x|: far inside the 1e-4 fp32 parity budget.
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    const float k = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
    float q = 1.061405429f;
    q = fmaf(q, k, -1.453152027f);
    q = fmaf(q, k, 1.421413741f);
    q = fmaf(q, k, -0.284496736f);
    q = fmaf(q, k, 0.254829592f);
    const float r = fmaf(-q * k, __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax), 1.f);
    return copysignf(r, x);
}

// 1 / (1 + e^-v): e^-v overflows to +inf for v < -88 -> rcp(inf) = 0, the correct limit; v > 88 -> 1 / (1 + 0) = 1.
__device__ __forceinline__ float sigmoid_fast(float v) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
}

__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case UWIR_ACT_RELU: return v > 0.f ? v : 0.f;
        case UWIR_ACT_GELU: return 0.5f * v * (1.f + erf_fast(v * 0.70710678118654752440f));  // exact (erf) GELU
        // v_exp_f32 / v_rcp_f32 (1 ulp each) instead of libm expf + an IEEE division: ~6 VALU instructions per element
        // against ~30 -- the SiLU epilogue of a 3x3 conv with 64 output channels cost more issue cycles than its MFMAs
        case UWIR_ACT_SIGMOID: return sigmoid_fast(v);
        case UWIR_ACT_SILU: return v * sigmoid_fast(v);
        default: return v;
    }
}

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return (float)v; }

// 4 consecutive elements -> float4, for either storage type
__device__ __forceinline__ float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 load4(const bf16_t* p) {
    bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
__device__ __forceinline__ void store4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void store4(bf16_t* p, float4 v) {
    bf16x4 o;
    o[0] = (bf16_t)v.x; o[1] = (bf16_t)v.y; o[2] = (bf16_t)v.z; o[3] = (bf16_t)v.w;
    *reinterpret_cast<bf16x4*>(p) = o;
}
