// rmsnorm_gate.hip -- K5: y = x * rsqrt(mean(x^2)+eps) * w [* g*sigmoid(g)] over the last dim.
//
// Replaces fla.modules.FusedRMSNormSwishGate / RMSNorm (reference model/gla.py:111,115,219,222;
// SURVEY.md 8(a) a-5, Appendix A.6).  One wave64 per row, 4 elements per lane per trip (8/16-byte
// loads), the row kept in registers between the sum-of-squares pass and the scale pass; the
// mean is reduced with wave64 xor-shuffles only (no LDS, no barrier).  Optional: the row arrives
// as `n_partial` fp32 partial sums (decode path).
#include <lina_dev.h>
#include "lina_common.h"

namespace lina {

constexpr int kNormMaxTrips = 8;  // D <= 8 * 256 = 2048

template <typename TX, typename T>
__global__ __launch_bounds__(256) void rmsnorm_gate_kernel(
    const TX* __restrict__ x, const T* __restrict__ g, const T* __restrict__ w, T* __restrict__ y,
    int64_t rows, int rows_inner, int D, int64_t x_outer, int64_t x_inner, int64_t g_outer, int64_t g_inner,
    int64_t y_outer, int64_t y_inner, int n_partial, int64_t x_part, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool live = row < rows;  // whole wave uniform
    const int64_t ro = row / rows_inner, ri = row % rows_inner;
    const int64_t x_off = ro * x_outer + ri * x_inner, g_off = ro * g_outer + ri * g_inner,
                  y_off = ro * y_outer + ri * y_inner;
    float4 xv[kNormMaxTrips];
    float ss = 0.0f;
    const int trips = (D + 255) / 256;
#pragma unroll
    for (int i = 0; i < kNormMaxTrips; ++i) {
        const int e = i * 256 + lane * 4;
        xv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live && i < trips && e < D) {
            const TX* xp = x + x_off + e;
            float4 a = ld4(xp);
            for (int p = 1; p < n_partial; ++p) {
                const float4 c = ld4(xp + p * x_part);
                a.x += c.x; a.y += c.y; a.z += c.z; a.w += c.w;
            }
            xv[i] = a;
            ss += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
        }
    }
    ss += shfl_xor(ss, 1); ss += shfl_xor(ss, 2); ss += shfl_xor(ss, 4);
    ss += shfl_xor(ss, 8); ss += shfl_xor(ss, 16); ss += shfl_xor(ss, 32);
    const float rs = rsqrtf(ss / (float)D + eps);
#pragma unroll
    for (int i = 0; i < kNormMaxTrips; ++i) {
        const int e = i * 256 + lane * 4;
        if (live && i < trips && e < D) {
            float4 a = xv[i];
            a.x *= rs; a.y *= rs; a.z *= rs; a.w *= rs;
            if (w) {
                const float4 ww = ld4(w + e);
                a.x *= ww.x; a.y *= ww.y; a.z *= ww.z; a.w *= ww.w;
            }
            if (g) {
                const float4 gg = ld4(g + g_off + e);
                a.x *= gg.x * sigmoidf(gg.x); a.y *= gg.y * sigmoidf(gg.y);
                a.z *= gg.z * sigmoidf(gg.z); a.w *= gg.w * sigmoidf(gg.w);
            }
            st4(y + y_off + e, a);
        }
    }
}

}  // namespace lina

extern "C" int lina_rmsnorm_gate_fwd(const void* x, const void* g, const void* w, void* y, int64_t rows,
                                     int rows_inner, int D, int64_t x_outer, int64_t x_inner, int64_t g_outer,
                                     int64_t g_inner, int64_t y_outer, int64_t y_inner, int n_partial,
                                     int64_t x_part_stride, float eps, int x_dtype, int dtype, lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(x && y, "lina_rmsnorm_gate_fwd: null pointer");
    LINA_REQUIRE(rows > 0, "lina_rmsnorm_gate_fwd: rows must be positive");
    LINA_REQUIRE(D > 0 && D % 4 == 0 && D <= kNormMaxTrips * 256,
                 "lina_rmsnorm_gate_fwd: D=%d must be a multiple of 4 and <= %d", D, kNormMaxTrips * 256);
    LINA_REQUIRE(rows_inner >= 1, "lina_rmsnorm_gate_fwd: rows_inner must be >= 1");
    LINA_REQUIRE(x_outer % 4 == 0 && x_inner % 4 == 0 && y_outer % 4 == 0 && y_inner % 4 == 0 &&
                     (!g || (g_outer % 4 == 0 && g_inner % 4 == 0)) && x_part_stride % 4 == 0,
                 "lina_rmsnorm_gate_fwd: row strides must be multiples of 4 elements");
    LINA_REQUIRE(valid_dtype(dtype) && valid_dtype(x_dtype), "lina_rmsnorm_gate_fwd: bad dtype");
    LINA_REQUIRE(n_partial >= 1, "lina_rmsnorm_gate_fwd: n_partial must be >= 1");
    dim3 grid((unsigned)((rows + 3) / 4));
    if (x_dtype == LINA_F32 && dtype == LINA_F32) {
        LINA_LAUNCH((rmsnorm_gate_kernel<float, float>), grid, dim3(256), 0, stream, (const float*)x, (const float*)g,
                    (const float*)w, (float*)y, rows, rows_inner, D, x_outer, x_inner, g_outer, g_inner, y_outer, y_inner, n_partial, x_part_stride, eps);
    } else if (x_dtype == LINA_F32 && dtype == LINA_BF16) {
        LINA_LAUNCH((rmsnorm_gate_kernel<float, bf16_t>), grid, dim3(256), 0, stream, (const float*)x, (const bf16_t*)g,
                    (const bf16_t*)w, (bf16_t*)y, rows, rows_inner, D, x_outer, x_inner, g_outer, g_inner, y_outer, y_inner, n_partial, x_part_stride, eps);
    } else if (x_dtype == LINA_BF16 && dtype == LINA_BF16) {
        LINA_LAUNCH((rmsnorm_gate_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, stream, (const bf16_t*)x,
                    (const bf16_t*)g, (const bf16_t*)w, (bf16_t*)y, rows, rows_inner, D, x_outer, x_inner, g_outer,
                    g_inner, y_outer, y_inner, n_partial, x_part_stride, eps);
    } else {
        return fail(LINA_ERR_UNSUPPORTED, "lina_rmsnorm_gate_fwd: x bf16 with f32 output is not built");
    }
    return check_launch("lina_rmsnorm_gate_fwd");
}
