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

// ================================================================================================
// K5b -- backward of K5 over contiguous rows [rows][D] (training path):
//   n = x rs, u = n w, s = g sigmoid(g), y = u s
//   dg = dy u sigmoid(g) (1 + g (1 - sigmoid(g)))      dn = dy s w      dw += dy s n
//   dx = rs (dn - n mean(dn n))
// One wave per row as in the forward; a wave walks rows with stride 4*gridDim.x and keeps its share of dw
// in registers; the 4 waves of a workgroup are summed through LDS into dw_partial[blockIdx.x][D] (fp32),
// which the caller sums (deterministic, no atomics).
// ================================================================================================
namespace lina {

constexpr int kNormBwdMaxWG = LINA_NORM_BWD_MAX_WG;

template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_gate_bwd_kernel(
    const T* __restrict__ x, const T* __restrict__ g, const T* __restrict__ w, const T* __restrict__ dy,
    T* __restrict__ dx, T* __restrict__ dg, float* __restrict__ dw_partial, int64_t rows, int D, float eps) {
    __shared__ float4 s_dw[3][kNormMaxTrips * 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int trips = (D + 255) / 256;
    float4 dwa[kNormMaxTrips];
#pragma unroll
    for (int i = 0; i < kNormMaxTrips; ++i) dwa[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t row = (int64_t)blockIdx.x * 4 + wv; row < rows; row += (int64_t)gridDim.x * 4) {
        const int64_t off = row * D;
        float4 xv[kNormMaxTrips], dn[kNormMaxTrips];
        float ss = 0.0f;
#pragma unroll
        for (int i = 0; i < kNormMaxTrips; ++i) {
            const int e = i * 256 + lane * 4;
            xv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < trips && e < D) {
                xv[i] = ld4(x + off + e);
                ss += xv[i].x * xv[i].x + xv[i].y * xv[i].y + xv[i].z * xv[i].z + xv[i].w * xv[i].w;
            }
        }
        ss += shfl_xor(ss, 1); ss += shfl_xor(ss, 2); ss += shfl_xor(ss, 4);
        ss += shfl_xor(ss, 8); ss += shfl_xor(ss, 16); ss += shfl_xor(ss, 32);
        const float rs = rsqrtf(ss / (float)D + eps);
        float dot = 0.0f;   // sum dn n
#pragma unroll
        for (int i = 0; i < kNormMaxTrips; ++i) {
            const int e = i * 256 + lane * 4;
            dn[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < trips && e < D) {
                const float4 d4 = ld4(dy + off + e);
                const float4 w4 = w ? ld4(w + e) : make_float4(1.f, 1.f, 1.f, 1.f);
                const float nn[4] = {xv[i].x * rs, xv[i].y * rs, xv[i].z * rs, xv[i].w * rs};
                const float dd[4] = {d4.x, d4.y, d4.z, d4.w};
                const float ww[4] = {w4.x, w4.y, w4.z, w4.w};
                float sv[4] = {1.f, 1.f, 1.f, 1.f}, dgo[4];
                if (g) {
                    const float4 g4 = ld4(g + off + e);
                    const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float sg = sigmoidf(gg[c]);
                        sv[c] = gg[c] * sg;
                        dgo[c] = dd[c] * nn[c] * ww[c] * sg * (1.0f + gg[c] * (1.0f - sg));
                    }
                    st4(dg + off + e, make_float4(dgo[0], dgo[1], dgo[2], dgo[3]));
                }
                float dnn[4], dwv[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float du = dd[c] * sv[c];
                    dwv[c] = du * nn[c];
                    dnn[c] = du * ww[c];
                    dot += dnn[c] * nn[c];
                }
                dwa[i].x += dwv[0]; dwa[i].y += dwv[1]; dwa[i].z += dwv[2]; dwa[i].w += dwv[3];
                dn[i] = make_float4(dnn[0], dnn[1], dnn[2], dnn[3]);
            }
        }
        dot += shfl_xor(dot, 1); dot += shfl_xor(dot, 2); dot += shfl_xor(dot, 4);
        dot += shfl_xor(dot, 8); dot += shfl_xor(dot, 16); dot += shfl_xor(dot, 32);
        const float m = dot / (float)D;
#pragma unroll
        for (int i = 0; i < kNormMaxTrips; ++i) {
            const int e = i * 256 + lane * 4;
            if (i < trips && e < D) {
                float4 a;
                a.x = rs * (dn[i].x - xv[i].x * rs * m); a.y = rs * (dn[i].y - xv[i].y * rs * m);
                a.z = rs * (dn[i].z - xv[i].z * rs * m); a.w = rs * (dn[i].w - xv[i].w * rs * m);
                st4(dx + off + e, a);
            }
        }
    }
    if (wv > 0) {
#pragma unroll
        for (int i = 0; i < kNormMaxTrips; ++i) s_dw[wv - 1][i * 64 + lane] = dwa[i];
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
        for (int i = 0; i < kNormMaxTrips; ++i) {
            const int e = i * 256 + lane * 4;
            if (i < trips && e < D) {
                float4 a = dwa[i];
#pragma unroll
                for (int o = 0; o < 3; ++o) {
                    const float4 c = s_dw[o][i * 64 + lane];
                    a.x += c.x; a.y += c.y; a.z += c.z; a.w += c.w;
                }
                *reinterpret_cast<float4*>(dw_partial + (int64_t)blockIdx.x * D + e) = a;
            }
        }
    }
}

}  // namespace lina

extern "C" int lina_rmsnorm_gate_bwd_partials(int64_t rows) {
    if (rows <= 0) return 0;
    const int64_t wg = (rows + 3) / 4;
    return (int)(wg < lina::kNormBwdMaxWG ? wg : lina::kNormBwdMaxWG);
}

extern "C" int lina_rmsnorm_gate_bwd(const void* x, const void* g, const void* w, const void* dy, void* dx, void* dg,
                                     float* dw_partial, int64_t rows, int D, float eps, int dtype,
                                     lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(x && dy && dx && dw_partial, "lina_rmsnorm_gate_bwd: null pointer");
    LINA_REQUIRE(!g == !dg, "lina_rmsnorm_gate_bwd: g and dg must both be given or both be NULL");
    LINA_REQUIRE(rows > 0, "lina_rmsnorm_gate_bwd: rows must be positive");
    LINA_REQUIRE(D > 0 && D % 4 == 0 && D <= kNormMaxTrips * 256,
                 "lina_rmsnorm_gate_bwd: D=%d must be a multiple of 4 and <= %d", D, kNormMaxTrips * 256);
    LINA_REQUIRE(valid_dtype(dtype), "lina_rmsnorm_gate_bwd: bad dtype");
    dim3 grid((unsigned)lina_rmsnorm_gate_bwd_partials(rows));
    if (dtype == LINA_F32) {
        LINA_LAUNCH((rmsnorm_gate_bwd_kernel<float>), grid, dim3(256), 0, stream, (const float*)x, (const float*)g,
                    (const float*)w, (const float*)dy, (float*)dx, (float*)dg, dw_partial, rows, D, eps);
    } else {
        LINA_LAUNCH((rmsnorm_gate_bwd_kernel<bf16_t>), grid, dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)g,
                    (const bf16_t*)w, (const bf16_t*)dy, (bf16_t*)dx, (bf16_t*)dg, dw_partial, rows, D, eps);
    }
    return check_launch("lina_rmsnorm_gate_bwd");
}
