// short_conv.hip -- K3/K4: causal depthwise short convolution (+SiLU) and its decode step.
//
// Replaces fla.modules.ShortConvolution (ctor reference model/gla.py:106-108, calls :161-163;
// SURVEY.md 8(a) a-4, Appendix A.2):  y_t[c] = act(sum_j w[c,j] * x_{t-(W-1)+j}[c] + bias[c]),
// x_{<0} = 0; cache[b,c,:] <- last W (masked) inputs.  Pure streaming op: lanes run along the
// contiguous channel dimension, each thread slides a W-wide register window over TT steps.
#include <lina_dev.h>
#include "lina_common.h"

namespace lina {

constexpr int kConvTT = 16;  // time steps per thread in the prefill kernel

// VEC channels per thread: 1, or 4 (8-byte bf16 / 16-byte fp32 accesses) when D and the strides allow it
template <int VEC, typename T> struct VecIO;
template <typename T> struct VecIO<1, T> {
    static __device__ __forceinline__ void load(const T* p, float (&v)[1]) { v[0] = ld(p); }
    static __device__ __forceinline__ void store(T* p, const float (&v)[1]) { st(p, v[0]); }
};
template <typename T> struct VecIO<4, T> {
    static __device__ __forceinline__ void load(const T* p, float (&v)[4]) {
        const float4 f = ld4(p);
        v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
    }
    static __device__ __forceinline__ void store(T* p, const float (&v)[4]) { st4(p, make_float4(v[0], v[1], v[2], v[3])); }
};

// RAW loads (bits only, converted where they are used): a block of time steps is requested in one go -- a load that is
// converted (or sits under a per-step branch) where it is issued is waited for on the spot, one memory round trip per step.
template <int VEC, typename T> struct RawIO;
template <typename T> struct RawIO<1, T> {
    typedef T type;
    static __device__ __forceinline__ type load(const T* p) { return ld_raw(p); }
    static __device__ __forceinline__ void cvt(type r, float (&v)[1]) { v[0] = cvt1(r); }
};
template <typename T> struct RawIO<4, T> {
    typedef typename raw4<T>::type type;
    static __device__ __forceinline__ type load(const T* p) { return ld4_raw(p); }
    static __device__ __forceinline__ void cvt(type r, float (&v)[4]) {
        const float4 f = cvt4(r);
        v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
    }
};

template <int W, typename T, int VEC>
__global__ __launch_bounds__(256) void short_conv_fwd_kernel(
    const T* __restrict__ x, const T* __restrict__ w, const T* __restrict__ bias, const float* __restrict__ mask,
    T* cache, T* __restrict__ y, int Tn, int D, int64_t x_sb, int64_t x_st, int64_t y_sb, int64_t y_st, int act) {
    const int c = (blockIdx.x * 256 + threadIdx.x) * VEC;
    const int t0 = blockIdx.y * kConvTT;
    const int b = blockIdx.z;
    if (c >= D) return;
    float wv[W][VEC], bv[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
#pragma unroll
        for (int j = 0; j < W; ++j) wv[j][i] = ld(w + (int64_t)(c + i) * W + j);
        bv[i] = bias ? ld(bias + c + i) : 0.0f;
    }
    const T* xb = x + b * x_sb + c;
    const float* mb = mask ? mask + (int64_t)b * Tn : nullptr;
    float win[W][VEC];  // win[j] = x_{t-(W-1)+j}
#pragma unroll
    for (int i = 0; i < VEC; ++i) win[0][i] = 0.0f;
#pragma unroll
    for (int j = 0; j < W - 1; ++j) {
        const int tt = t0 - (W - 1) + j;
        float v[VEC];
        VecIO<VEC, T>::load(xb + max(tt, 0) * x_st, v);
        const float m = (tt >= 0) ? (mb ? mb[tt] : 1.0f) : 0.0f;
#pragma unroll
        for (int i = 0; i < VEC; ++i) win[j + 1][i] = v[i] * m;
    }
    const int t1 = min(t0 + kConvTT, Tn);
    typename RawIO<VEC, T>::type xr[kConvTT];               // all of this thread's steps in flight (rows past the end: row Tn-1)
    float mr[kConvTT];
#pragma unroll
    for (int k = 0; k < kConvTT; ++k) {
        const int tc = min(t0 + k, Tn - 1);
        xr[k] = RawIO<VEC, T>::load(xb + tc * x_st);
        mr[k] = mb ? mb[tc] : 1.0f;
    }
#pragma unroll
    for (int k = 0; k < kConvTT; ++k) {
        const int t = t0 + k;
        if (t >= t1) break;
        float v[VEC], o[VEC];
        RawIO<VEC, T>::cvt(xr[k], v);
        const float m = mr[k];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
#pragma unroll
            for (int j = 0; j < W - 1; ++j) win[j][i] = win[j + 1][i];
            win[W - 1][i] = v[i] * m;
            float acc = bv[i];
#pragma unroll
            for (int j = 0; j < W; ++j) acc = fmaf(wv[j][i], win[j][i], acc);
            o[i] = act ? silu(acc) : acc;
        }
        VecIO<VEC, T>::store(y + b * y_sb + t * y_st + c, o);
    }
    if (cache && t1 == Tn) {  // this thread saw the tail: win holds x_{Tn-W..Tn-1} (zeros left of 0)
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            T* cb = cache + ((int64_t)b * D + c + i) * W;
#pragma unroll
            for (int j = 0; j < W; ++j) st(cb + j, win[j][i]);
        }
    }
}

template <int W, typename T>
__global__ __launch_bounds__(256) void short_conv_step_kernel(
    const T* __restrict__ x, const T* __restrict__ w, const T* __restrict__ bias, T* cache, T* __restrict__ y,
    int D, int64_t x_sb, int64_t y_sb, int act) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (c >= D) return;
    T* cb = cache + ((int64_t)b * D + c) * W;
    float win[W];
#pragma unroll
    for (int j = 0; j < W - 1; ++j) win[j] = ld(cb + j + 1);
    win[W - 1] = ld(x + b * x_sb + c);
    float acc = bias ? ld(bias + c) : 0.0f;
#pragma unroll
    for (int j = 0; j < W; ++j) {
        acc = fmaf(ld(w + (int64_t)c * W + j), win[j], acc);
        st(cb + j, win[j]);
    }
    st(y + b * y_sb + c, act ? silu(acc) : acc);
}

template <typename T>
static int conv_fwd_dispatch(const void* x, const void* w, const void* bias, const float* mask, void* cache, void* y,
                             int B, int Tn, int D, int W, int64_t x_sb, int64_t x_st, int64_t y_sb, int64_t y_st,
                             int act, lina_stream_t stream) {
    const bool vec = D % 4 == 0 && x_sb % 4 == 0 && x_st % 4 == 0 && y_sb % 4 == 0 && y_st % 4 == 0 &&
                     ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0);
    const int per = vec ? 4 : 1;
    dim3 grid((unsigned)((D + 256 * per - 1) / (256 * per)), (unsigned)((Tn + kConvTT - 1) / kConvTT), (unsigned)B);
#define LINA_CONV_LAUNCH(WW, VV)                                                                                    \
    LINA_LAUNCH((short_conv_fwd_kernel<WW, T, VV>), grid, dim3(256), 0, stream, (const T*)x, (const T*)w,          \
                (const T*)bias, mask, (T*)cache, (T*)y, Tn, D, x_sb, x_st, y_sb, y_st, act)
#define LINA_CONV_CASE(WW)                                                                                          \
    case WW:                                                                                                        \
        if (vec) LINA_CONV_LAUNCH(WW, 4); else LINA_CONV_LAUNCH(WW, 1);                                             \
        break;
    switch (W) {
        LINA_CONV_CASE(2) LINA_CONV_CASE(3) LINA_CONV_CASE(4) LINA_CONV_CASE(5)
        LINA_CONV_CASE(6) LINA_CONV_CASE(7) LINA_CONV_CASE(8)
        default: return fail(LINA_ERR_UNSUPPORTED, "lina_short_conv_fwd: W=%d not in 2..8", W);
    }
#undef LINA_CONV_CASE
#undef LINA_CONV_LAUNCH
    return check_launch("lina_short_conv_fwd");
}

template <typename T>
static int conv_step_dispatch(const void* x, const void* w, const void* bias, void* cache, void* y, int B, int D, int W,
                              int64_t x_sb, int64_t y_sb, int act, lina_stream_t stream) {
    dim3 grid((unsigned)((D + 255) / 256), (unsigned)B);
#define LINA_CONV_CASE(WW)                                                                                     \
    case WW:                                                                                                   \
        LINA_LAUNCH((short_conv_step_kernel<WW, T>), grid, dim3(256), 0, stream, (const T*)x, (const T*)w,     \
                    (const T*)bias, (T*)cache, (T*)y, D, x_sb, y_sb, act);                                     \
        break;
    switch (W) {
        LINA_CONV_CASE(2) LINA_CONV_CASE(3) LINA_CONV_CASE(4) LINA_CONV_CASE(5)
        LINA_CONV_CASE(6) LINA_CONV_CASE(7) LINA_CONV_CASE(8)
        default: return fail(LINA_ERR_UNSUPPORTED, "lina_short_conv_step: W=%d not in 2..8", W);
    }
#undef LINA_CONV_CASE
    return check_launch("lina_short_conv_step");
}

}  // namespace lina

extern "C" int lina_short_conv_fwd(const void* x, const void* w, const void* bias, const float* mask, void* cache,
                                   void* y, int B, int T, int D, int W, int64_t x_sb, int64_t x_st, int64_t y_sb,
                                   int64_t y_st, int activation, int dtype, lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(x && w && y, "lina_short_conv_fwd: null pointer");
    LINA_REQUIRE(B > 0 && T > 0 && D > 0, "lina_short_conv_fwd: B,T,D must be positive (got %d,%d,%d)", B, T, D);
    LINA_REQUIRE(valid_dtype(dtype), "lina_short_conv_fwd: bad dtype %d", dtype);
    LINA_REQUIRE(activation == 0 || activation == 1, "lina_short_conv_fwd: activation must be 0 or 1");
    if (dtype == LINA_F32)
        return conv_fwd_dispatch<float>(x, w, bias, mask, cache, y, B, T, D, W, x_sb, x_st, y_sb, y_st, activation, stream);
    return conv_fwd_dispatch<bf16_t>(x, w, bias, mask, cache, y, B, T, D, W, x_sb, x_st, y_sb, y_st, activation, stream);
}

extern "C" int lina_short_conv_step(const void* x, const void* w, const void* bias, void* cache, void* y, int B, int D,
                                    int W, int64_t x_sb, int64_t y_sb, int activation, int dtype,
                                    lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(x && w && y && cache, "lina_short_conv_step: null pointer");
    LINA_REQUIRE(B > 0 && D > 0, "lina_short_conv_step: B,D must be positive (got %d,%d)", B, D);
    LINA_REQUIRE(valid_dtype(dtype), "lina_short_conv_step: bad dtype %d", dtype);
    LINA_REQUIRE(activation == 0 || activation == 1, "lina_short_conv_step: activation must be 0 or 1");
    if (dtype == LINA_F32) return conv_step_dispatch<float>(x, w, bias, cache, y, B, D, W, x_sb, y_sb, activation, stream);
    return conv_step_dispatch<bf16_t>(x, w, bias, cache, y, B, D, W, x_sb, y_sb, activation, stream);
}

// ================================================================================================
// K3b -- backward of the prefill convolution (no cache): recompute z, dz = dy * act'(z),
//   dx_t = mask_t * sum_j w[j] dz_{t+(W-1)-j},   dw[j] = sum_{b,t} dz_t xm_{t-(W-1)+j},   dbias = sum dz.
// Same thread map as the forward (lanes along channels, one thread slides over kConvBwdTT steps, plus W-1
// look-ahead steps whose dz it needs for dx).  Weight / bias gradients leave as fp32 partials
// [B * ceil(T/kConvBwdTT)][D][W+1] (slot W = bias) that the caller sums: deterministic, no atomics.
// ================================================================================================
namespace lina {

constexpr int kConvBwdTT = LINA_CONV_BWD_TT;

template <int W, typename T, int VEC>
__global__ __launch_bounds__(256) void short_conv_bwd_kernel(
    const T* __restrict__ x, const T* __restrict__ w, const T* __restrict__ bias, const float* __restrict__ mask,
    const T* __restrict__ dy, T* __restrict__ dx, float* __restrict__ part, int Tn, int D, int64_t x_sb, int64_t x_st,
    int64_t dy_sb, int64_t dy_st, int64_t dx_sb, int64_t dx_st, int act) {
    const int c = (blockIdx.x * 256 + threadIdx.x) * VEC;
    const int t0 = blockIdx.y * kConvBwdTT;
    const int b = blockIdx.z;
    if (c >= D) return;
    float wv[W][VEC], dw[W][VEC], dzw[W][VEC], db[VEC], bv[VEC];   // dzw[j] = dz_{u-j}
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
#pragma unroll
        for (int j = 0; j < W; ++j) {
            wv[j][i] = ld(w + (int64_t)(c + i) * W + j);
            dw[j][i] = 0.0f;
            dzw[j][i] = 0.0f;
        }
        db[i] = 0.0f;
        bv[i] = bias ? ld(bias + c + i) : 0.0f;
    }
    const T* xb = x + b * x_sb + c;
    const T* dyb = dy + b * dy_sb + c;
    const float* mb = mask ? mask + (int64_t)b * Tn : nullptr;
    float win[W][VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) win[0][i] = 0.0f;
#pragma unroll
    for (int j = 0; j < W - 1; ++j) {
        const int tt = t0 - (W - 1) + j;
        float v[VEC];
        VecIO<VEC, T>::load(xb + max(tt, 0) * x_st, v);
        const float m = (tt >= 0) ? (mb ? mb[tt] : 1.0f) : 0.0f;
#pragma unroll
        for (int i = 0; i < VEC; ++i) win[j + 1][i] = v[i] * m;
    }
    const int t1 = min(t0 + kConvBwdTT, Tn);
    constexpr int KB = 8;                                    // steps requested per block (x and dy: 2 KB loads in flight)
    for (int u0 = t0; u0 < t1 + W - 1; u0 += KB) {
    typename RawIO<VEC, T>::type xr[KB], dr[KB];
    float mr[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) {
        const int uc = min(u0 + k, Tn - 1);
        xr[k] = RawIO<VEC, T>::load(xb + uc * x_st);
        dr[k] = RawIO<VEC, T>::load(dyb + uc * dy_st);
        mr[k] = mb ? mb[uc] : 1.0f;
    }
#pragma unroll
    for (int k = 0; k < KB; ++k) {
        const int u = u0 + k;
        if (u >= t1 + W - 1) break;
        float dz[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) dz[i] = 0.0f;
        if (u < Tn) {
            float xv[VEC], dv[VEC];
            RawIO<VEC, T>::cvt(xr[k], xv);
            RawIO<VEC, T>::cvt(dr[k], dv);
            const float m = mr[k];
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
#pragma unroll
                for (int j = 0; j < W - 1; ++j) win[j][i] = win[j + 1][i];
                win[W - 1][i] = xv[i] * m;
                float z = bv[i];
#pragma unroll
                for (int j = 0; j < W; ++j) z = fmaf(wv[j][i], win[j][i], z);
                float d = dv[i];
                if (act) {
                    const float sg = sigmoidf(z);
                    d *= sg * (1.0f + z * (1.0f - sg));
                }
                dz[i] = d;
                if (u < t1) {
#pragma unroll
                    for (int j = 0; j < W; ++j) dw[j][i] = fmaf(d, win[j][i], dw[j][i]);
                    db[i] += d;
                }
            }
        }
        const int t = u - (W - 1);
        float a[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
#pragma unroll
            for (int j = W - 1; j > 0; --j) dzw[j][i] = dzw[j - 1][i];
            dzw[0][i] = dz[i];
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < W; ++j) acc = fmaf(wv[j][i], dzw[j][i], acc);
            a[i] = acc * ((mb && t >= 0) ? mb[t] : 1.0f);
        }
        if (t >= t0) VecIO<VEC, T>::store(dx + b * dx_sb + t * dx_st + c, a);
    }
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        float* pp = part + (((int64_t)b * gridDim.y + blockIdx.y) * D + c + i) * (W + 1);
#pragma unroll
        for (int j = 0; j < W; ++j) pp[j] = dw[j][i];
        pp[W] = db[i];
    }
}

template <typename T>
static int conv_bwd_dispatch(const void* x, const void* w, const void* bias, const float* mask, const void* dy,
                             void* dx, float* part, int B, int Tn, int D, int W, int64_t x_sb, int64_t x_st,
                             int64_t dy_sb, int64_t dy_st, int64_t dx_sb, int64_t dx_st, int act, lina_stream_t stream) {
    const bool vec = D % 4 == 0 && x_sb % 4 == 0 && x_st % 4 == 0 && dy_sb % 4 == 0 && dy_st % 4 == 0 && dx_sb % 4 == 0 &&
                     dx_st % 4 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)dy % 16 == 0) && ((uintptr_t)dx % 16 == 0);
    const int per = vec ? 4 : 1;
    dim3 grid((unsigned)((D + 256 * per - 1) / (256 * per)), (unsigned)((Tn + kConvBwdTT - 1) / kConvBwdTT), (unsigned)B);
#define LINA_CONV_LAUNCH(WW, VV)                                                                                    \
    LINA_LAUNCH((short_conv_bwd_kernel<WW, T, VV>), grid, dim3(256), 0, stream, (const T*)x, (const T*)w,          \
                (const T*)bias, mask, (const T*)dy, (T*)dx, part, Tn, D, x_sb, x_st, dy_sb, dy_st, dx_sb, dx_st, act)
#define LINA_CONV_CASE(WW)                                                                                          \
    case WW:                                                                                                        \
        if (vec) LINA_CONV_LAUNCH(WW, 4); else LINA_CONV_LAUNCH(WW, 1);                                             \
        break;
    switch (W) {
        LINA_CONV_CASE(2) LINA_CONV_CASE(3) LINA_CONV_CASE(4) LINA_CONV_CASE(5)
        LINA_CONV_CASE(6) LINA_CONV_CASE(7) LINA_CONV_CASE(8)
        default: return fail(LINA_ERR_UNSUPPORTED, "lina_short_conv_bwd: W=%d not in 2..8", W);
    }
#undef LINA_CONV_CASE
#undef LINA_CONV_LAUNCH
    return check_launch("lina_short_conv_bwd");
}

}  // namespace lina

extern "C" int lina_short_conv_bwd(const void* x, const void* w, const void* bias, const float* mask, const void* dy,
                                   void* dx, float* dwb_partial, int B, int T, int D, int W, int64_t x_sb,
                                   int64_t x_st, int64_t dy_sb, int64_t dy_st, int64_t dx_sb, int64_t dx_st,
                                   int activation, int dtype, lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(x && w && dy && dx && dwb_partial, "lina_short_conv_bwd: null pointer");
    LINA_REQUIRE(B > 0 && T > 0 && D > 0, "lina_short_conv_bwd: B,T,D must be positive (got %d,%d,%d)", B, T, D);
    LINA_REQUIRE(valid_dtype(dtype), "lina_short_conv_bwd: bad dtype %d", dtype);
    LINA_REQUIRE(activation == 0 || activation == 1, "lina_short_conv_bwd: activation must be 0 or 1");
    if (dtype == LINA_F32)
        return conv_bwd_dispatch<float>(x, w, bias, mask, dy, dx, dwb_partial, B, T, D, W, x_sb, x_st, dy_sb, dy_st,
                                        dx_sb, dx_st, activation, stream);
    return conv_bwd_dispatch<bf16_t>(x, w, bias, mask, dy, dx, dwb_partial, B, T, D, W, x_sb, x_st, dy_sb, dy_st, dx_sb,
                                     dx_st, activation, stream);
}
