// lina_common.h -- host-side helpers shared by the C-ABI launchers (error text, checks)
// and tiny device-side dtype adaptors.  Included after <lina_dev.h>.
#pragma once
#include <lina_dev.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/lina_gla.h"

namespace lina {

char* last_error_buf();  // thread-local, defined in abi.hip

static inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

static inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(LINA_ERR_LAUNCH, "%s: launch failed: %s", what, hipGetErrorString(e));
    return LINA_OK;
}

#define LINA_REQUIRE(cond, ...) \
    do {                        \
        if (!(cond)) return ::lina::fail(LINA_ERR_ARG, __VA_ARGS__); \
    } while (0)

// ---- element load/store adaptors: T = float or unsigned short (bf16 bits) ----
typedef unsigned short bf16_t;

__device__ __forceinline__ float ld(const float* p) { return *p; }
__device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
__device__ __forceinline__ void st(float* p, float v) { *p = v; }
__device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }

// 4 consecutive elements (pointer must be 16-B aligned for float, 8-B for bf16)
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const bf16_t* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(bf2f((bf16_t)(u.x & 0xffff)), bf2f((bf16_t)(u.x >> 16)),
                       bf2f((bf16_t)(u.y & 0xffff)), bf2f((bf16_t)(u.y >> 16)));
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(bf16_t* p, float4 v) {
    uint2 u;
    u.x = pack_bf16x2(v.x, v.y);
    u.y = pack_bf16x2(v.z, v.w);
    *reinterpret_cast<uint2*>(p) = u;
}

// 2 consecutive elements (pointer 8-B aligned for float, 4-B for bf16)
__device__ __forceinline__ void st_pair(float* p, float a, float b) { *reinterpret_cast<float2*>(p) = make_float2(a, b); }
__device__ __forceinline__ void st_pair(bf16_t* p, float a, float b) { *reinterpret_cast<uint32_t*>(p) = pack_bf16x2(a, b); }

// RAW forms of the 4-element load: a load whose value is only needed much later must not be converted where it is issued --
// the bf16 -> fp32 conversion is a USE of the loaded register, i.e. an `s_waitcnt` for this load and for every load issued
// before it.  Keep the raw bits, convert at the point of use.
template <typename T> struct raw4;
template <> struct raw4<float> { typedef float4 type; };
template <> struct raw4<bf16_t> { typedef uint2 type; };
__device__ __forceinline__ float4 ld4_raw(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ uint2 ld4_raw(const bf16_t* p) { return *reinterpret_cast<const uint2*>(p); }
__device__ __forceinline__ float4 cvt4(float4 v) { return v; }
__device__ __forceinline__ float4 cvt4(uint2 u) {
    return make_float4(bf2f((bf16_t)(u.x & 0xffff)), bf2f((bf16_t)(u.x >> 16)), bf2f((bf16_t)(u.y & 0xffff)),
                       bf2f((bf16_t)(u.y >> 16)));
}
__device__ __forceinline__ float ld_raw(const float* p) { return *p; }
__device__ __forceinline__ bf16_t ld_raw(const bf16_t* p) { return *p; }
__device__ __forceinline__ float cvt1(float v) { return v; }
__device__ __forceinline__ float cvt1(bf16_t v) { return bf2f(v); }

// sigmoid via v_exp_f32 + v_rcp_f32 (1 ulp): a `/` here expands to the ~10-instruction IEEE division sequence, which
// made the per-element epilogues (short conv, norm-gate, SwiGLU) VALU-bound
__device__ __forceinline__ float sigmoidf(float x) { return fast_rcp(1.0f + __expf(-x)); }
__device__ __forceinline__ float silu(float x) { return x * sigmoidf(x); }
// log(sigmoid(x)) = min(x,0) - log1p(exp(-|x|))   (the form torch's CPU kernel uses), on the hardware transcendentals:
// e = v_exp_f32, log1p(e) = v_log_f32(1 + e) above 1/64 and the series e - e^2/2 + e^3/3 below (where 1 + e would round
// e away): absolute error < 1e-7 everywhere, relative error of the small values < 2e-6.  The precise libm forms (log1pf,
// expf: ~100 instructions per call) made the gate tiles the slowest workgroups of the in-projection launch -- 8 calls per
// thread, a ~2.5 us tail on a 10.5 us kernel (time stamps of tools/probe_skinny_prof.py).
__device__ __forceinline__ float logsigmoidf(float x) {
    const float e = __expf(-fabsf(x));
    const float l = e < 0.015625f ? e * (1.0f - e * (0.5f - e * 0.33333334f)) : __logf(1.0f + e);
    return fminf(x, 0.0f) - l;
}

static inline bool valid_dtype(int d) { return d == LINA_F32 || d == LINA_BF16; }

}  // namespace lina
