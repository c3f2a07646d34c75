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

// TR = pieces of 256 elements per row (D <= 256 TR); a wave works on RG = 8 / TR rows at once: its 8 register slots are
// (row r, piece p) pairs.  EVERY load of the group -- x (all partials), g, w -- is unconditional on a clamped offset and
// issued before anything uses it: with one row per wave and the gate / weight loads behind the reduction a wave had a single
// 8-byte load in flight per memory round trip (L169, D = 256: 1.9 TB/s in the train step's profile).
template <typename TX, typename T, int TR>
__global__ __launch_bounds__(256) void rmsnorm_gate_kernel(
    const TX* __restrict__ x, const T* __restrict__ g, const T* __restrict__ w, T* __restrict__ y,
    int64_t rows, int rows_inner, int D, int64_t x_outer, int64_t x_inner, int64_t g_outer, int64_t g_inner,
    int64_t y_outer, int64_t y_inner, int n_partial, int64_t x_part, float eps) {
    constexpr int RG = kNormMaxTrips / TR;
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave_uniform(threadIdx.x >> 6)) * RG;
    if (row0 >= rows) return;                                // wave-uniform
    int64_t x_off[RG], g_off[RG], y_off[RG];
#pragma unroll
    for (int r = 0; r < RG; ++r) {
        const int64_t row = min(row0 + r, rows - 1);         // rows past the end re-read the last row (not stored)
        const int64_t ro = row / rows_inner, ri = row % rows_inner;
        x_off[r] = ro * x_outer + ri * x_inner;
        g_off[r] = ro * g_outer + ri * g_inner;
        y_off[r] = ro * y_outer + ri * y_inner;
    }
    typename raw4<TX>::type xr[kNormMaxTrips];
    typename raw4<T>::type gr[kNormMaxTrips], wr[TR];
#pragma unroll
    for (int p = 0; p < TR; ++p) {
        const int e = p * 256 + lane * 4, ec = e < D ? e : D - 4;
        if (w) wr[p] = ld4_raw(w + ec);
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            xr[r * TR + p] = ld4_raw(x + x_off[r] + ec);
            if (g) gr[r * TR + p] = ld4_raw(g + g_off[r] + ec);
        }
    }
    float4 xv[kNormMaxTrips];
    float ss[RG];
#pragma unroll
    for (int r = 0; r < RG; ++r) {
        ss[r] = 0.0f;
#pragma unroll
        for (int p = 0; p < TR; ++p) {
            const int e = p * 256 + lane * 4, ec = e < D ? e : D - 4;
            float4 a = cvt4(xr[r * TR + p]);
            for (int q = 1; q < n_partial; ++q) {            // decode path: fp32 partial sums of the row
                const float4 c = ld4(x + x_off[r] + ec + q * x_part);
                a.x += c.x; a.y += c.y; a.z += c.z; a.w += c.w;
            }
            xv[r * TR + p] = a;
            ss[r] += e < D ? a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w : 0.0f;
        }
    }
#pragma unroll
    for (int r = 0; r < RG; ++r) {
        float t = ss[r];
        t += shfl_xor(t, 1); t += shfl_xor(t, 2); t += shfl_xor(t, 4);
        t += shfl_xor(t, 8); t += shfl_xor(t, 16); t += shfl_xor(t, 32);
        const float rs = rsqrtf(t / (float)D + eps);
        if (row0 + r >= rows) break;                         // wave-uniform
#pragma unroll
        for (int p = 0; p < TR; ++p) {
            const int e = p * 256 + lane * 4;
            float4 a = xv[r * TR + p];
            a.x *= rs; a.y *= rs; a.z *= rs; a.w *= rs;
            if (w) {
                const float4 ww = cvt4(wr[p]);
                a.x *= ww.x; a.y *= ww.y; a.z *= ww.z; a.w *= ww.w;
            }
            if (g) {
                const float4 gg = cvt4(gr[r * TR + p]);
                a.x *= gg.x * sigmoidf(gg.x); a.y *= gg.y * sigmoidf(gg.y);
                a.z *= gg.z * sigmoidf(gg.z); a.w *= gg.w * sigmoidf(gg.w);
            }
            if (e < D) st4(y + y_off[r] + e, a);
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
    const int tr = D <= 256 ? 1 : D <= 512 ? 2 : D <= 1024 ? 4 : 8;       // pieces per row; a wave takes 8 / tr rows at a time
    const int64_t groups = (rows + (kNormMaxTrips / tr) - 1) / (kNormMaxTrips / tr);
    dim3 grid((unsigned)((groups + 3) / 4));
#define LINA_RN_F(TXX, TT, TRR)                                                                                        \
    LINA_LAUNCH((rmsnorm_gate_kernel<TXX, TT, TRR>), grid, dim3(256), 0, stream, (const TXX*)x, (const TT*)g, (const TT*)w, \
                (TT*)y, rows, rows_inner, D, x_outer, x_inner, g_outer, g_inner, y_outer, y_inner, n_partial, x_part_stride, eps)
#define LINA_RN_FT(TXX, TT)                                                                                            \
    do {                                                                                                               \
        if (tr == 1) LINA_RN_F(TXX, TT, 1); else if (tr == 2) LINA_RN_F(TXX, TT, 2);                                   \
        else if (tr == 4) LINA_RN_F(TXX, TT, 4); else LINA_RN_F(TXX, TT, 8);                                           \
    } while (0)
    if (x_dtype == LINA_F32 && dtype == LINA_F32) LINA_RN_FT(float, float);
    else if (x_dtype == LINA_F32 && dtype == LINA_BF16) LINA_RN_FT(float, bf16_t);
    else if (x_dtype == LINA_BF16 && dtype == LINA_BF16) LINA_RN_FT(bf16_t, bf16_t);
    else return fail(LINA_ERR_UNSUPPORTED, "lina_rmsnorm_gate_fwd: x bf16 with f32 output is not built");
#undef LINA_RN_FT
#undef LINA_RN_F
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

template <typename T, int TR>
__global__ __launch_bounds__(256) void rmsnorm_gate_bwd_kernel(
    const T* __restrict__ x, const T* __restrict__ g, const T* __restrict__ w, const T* __restrict__ dy,
    T* __restrict__ dx, T* __restrict__ dg, float* __restrict__ dw_partial, int64_t rows, int D, float eps,
    int rows_inner, int64_t g_outer, int64_t g_inner, int64_t dg_outer, int64_t dg_inner) {
    // slots (row r, piece p) as in the forward: RG = 8 / TR rows per wave and iteration, every load of the group first.
    // g / dg row r sits at (r / rows_inner) * outer + (r % rows_inner) * inner: head slices of wider rows, in place
    constexpr int RG = kNormMaxTrips / TR;
    __shared__ float4 s_dw[3][TR * 64];
    const int lane = threadIdx.x & 63, wv = wave_uniform(threadIdx.x >> 6);
    float4 dwa[TR], w4[TR];
#pragma unroll
    for (int p = 0; p < TR; ++p) {
        const int e = p * 256 + lane * 4;
        dwa[p] = make_float4(0.f, 0.f, 0.f, 0.f);
        w4[p] = w ? ld4(w + (e < D ? e : D - 4)) : make_float4(1.f, 1.f, 1.f, 1.f);
    }
    for (int64_t row0 = ((int64_t)blockIdx.x * 4 + wv) * RG; row0 < rows; row0 += (int64_t)gridDim.x * 4 * RG) {
        typename raw4<T>::type xr[kNormMaxTrips], dr[kNormMaxTrips], gr[kNormMaxTrips];
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            const int64_t rr = min(row0 + r, rows - 1), off = rr * D;
            const int64_t goff = (rr / rows_inner) * g_outer + (rr % rows_inner) * g_inner;
#pragma unroll
            for (int p = 0; p < TR; ++p) {
                const int e = p * 256 + lane * 4, ec = e < D ? e : D - 4;
                xr[r * TR + p] = ld4_raw(x + off + ec);
                dr[r * TR + p] = ld4_raw(dy + off + ec);
                if (g) gr[r * TR + p] = ld4_raw(g + goff + ec);
            }
        }
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            if (row0 + r >= rows) break;                     // wave-uniform
            const int64_t off = (row0 + r) * D;
            const int64_t dgoff = ((row0 + r) / rows_inner) * dg_outer + ((row0 + r) % rows_inner) * dg_inner;
            float4 xv[TR], dn[TR];
            float ss = 0.0f;
#pragma unroll
            for (int p = 0; p < TR; ++p) {
                xv[p] = cvt4(xr[r * TR + p]);
                ss += (p * 256 + lane * 4 < D) ? xv[p].x * xv[p].x + xv[p].y * xv[p].y + xv[p].z * xv[p].z + xv[p].w * xv[p].w : 0.0f;
            }
            ss += shfl_xor(ss, 1); ss += shfl_xor(ss, 2); ss += shfl_xor(ss, 4);
            ss += shfl_xor(ss, 8); ss += shfl_xor(ss, 16); ss += shfl_xor(ss, 32);
            const float rs = rsqrtf(ss / (float)D + eps);
            float dot = 0.0f;   // sum dn n
#pragma unroll
            for (int p = 0; p < TR; ++p) {
                const int e = p * 256 + lane * 4;
                const bool ok = e < D;
                const float4 d4 = cvt4(dr[r * TR + p]);
                const float nn[4] = {xv[p].x * rs, xv[p].y * rs, xv[p].z * rs, xv[p].w * rs};
                const float dd[4] = {ok ? d4.x : 0.f, ok ? d4.y : 0.f, ok ? d4.z : 0.f, ok ? d4.w : 0.f};
                const float ww[4] = {w4[p].x, w4[p].y, w4[p].z, w4[p].w};
                float sv[4] = {1.f, 1.f, 1.f, 1.f}, dgo[4];
                if (g) {
                    const float4 g4 = cvt4(gr[r * TR + p]);
                    const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float sg = sigmoidf(gg[c]);
                        sv[c] = gg[c] * sg;
                        dgo[c] = dd[c] * nn[c] * ww[c] * sg * (1.0f + gg[c] * (1.0f - sg));
                    }
                    if (ok) st4(dg + dgoff + e, make_float4(dgo[0], dgo[1], dgo[2], dgo[3]));
                }
                float dnn[4], dwv[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float du = dd[c] * sv[c];
                    dwv[c] = du * nn[c];
                    dnn[c] = du * ww[c];
                    dot += dnn[c] * nn[c];
                }
                dwa[p].x += dwv[0]; dwa[p].y += dwv[1]; dwa[p].z += dwv[2]; dwa[p].w += dwv[3];
                dn[p] = make_float4(dnn[0], dnn[1], dnn[2], dnn[3]);
            }
            dot += shfl_xor(dot, 1); dot += shfl_xor(dot, 2); dot += shfl_xor(dot, 4);
            dot += shfl_xor(dot, 8); dot += shfl_xor(dot, 16); dot += shfl_xor(dot, 32);
            const float m = dot / (float)D;
#pragma unroll
            for (int p = 0; p < TR; ++p) {
                const int e = p * 256 + lane * 4;
                if (e < D) {
                    float4 a;
                    a.x = rs * (dn[p].x - xv[p].x * rs * m); a.y = rs * (dn[p].y - xv[p].y * rs * m);
                    a.z = rs * (dn[p].z - xv[p].z * rs * m); a.w = rs * (dn[p].w - xv[p].w * rs * m);
                    st4(dx + off + e, a);
                }
            }
        }
    }
    if (wv > 0) {
#pragma unroll
        for (int p = 0; p < TR; ++p) s_dw[wv - 1][p * 64 + lane] = dwa[p];
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
        for (int p = 0; p < TR; ++p) {
            const int e = p * 256 + lane * 4;
            if (e < D) {
                float4 a = dwa[p];
#pragma unroll
                for (int o = 0; o < 3; ++o) {
                    const float4 c = s_dw[o][p * 64 + lane];
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
                                     float* dw_partial, int64_t rows, int rows_inner, int D, int64_t g_outer, int64_t g_inner,
                                     int64_t dg_outer, int64_t dg_inner, float eps, int dtype, lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(x && dy && dx && dw_partial, "lina_rmsnorm_gate_bwd: null pointer");
    LINA_REQUIRE(!g == !dg, "lina_rmsnorm_gate_bwd: g and dg must both be given or both be NULL");
    LINA_REQUIRE(rows > 0, "lina_rmsnorm_gate_bwd: rows must be positive");
    LINA_REQUIRE(rows_inner >= 1 && rows % rows_inner == 0, "lina_rmsnorm_gate_bwd: rows=%lld is not a multiple of rows_inner=%d",
                 (long long)rows, rows_inner);
    LINA_REQUIRE(!g || ((g_outer | g_inner | dg_outer | dg_inner) % 4 == 0 && g_outer >= 0 && g_inner >= 0 && dg_outer > 0 &&
                        (rows_inner == 1 || dg_inner >= D)),
                 "lina_rmsnorm_gate_bwd: gate strides must be multiples of 4 elements and dg rows must not overlap");
    LINA_REQUIRE(D > 0 && D % 4 == 0 && D <= kNormMaxTrips * 256,
                 "lina_rmsnorm_gate_bwd: D=%d must be a multiple of 4 and <= %d", D, kNormMaxTrips * 256);
    LINA_REQUIRE(valid_dtype(dtype), "lina_rmsnorm_gate_bwd: bad dtype");
    dim3 grid((unsigned)lina_rmsnorm_gate_bwd_partials(rows));
    const int tr = D <= 256 ? 1 : D <= 512 ? 2 : D <= 1024 ? 4 : 8;
#define LINA_RN_B(TT, TRR)                                                                                           \
    LINA_LAUNCH((rmsnorm_gate_bwd_kernel<TT, TRR>), grid, dim3(256), 0, stream, (const TT*)x, (const TT*)g, (const TT*)w, \
                (const TT*)dy, (TT*)dx, (TT*)dg, dw_partial, rows, D, eps, rows_inner, g_outer, g_inner, dg_outer, dg_inner)
#define LINA_RN_BT(TT)                                                                                               \
    do {                                                                                                             \
        if (tr == 1) LINA_RN_B(TT, 1); else if (tr == 2) LINA_RN_B(TT, 2); else if (tr == 4) LINA_RN_B(TT, 4);       \
        else LINA_RN_B(TT, 8);                                                                                       \
    } while (0)
    if (dtype == LINA_F32) LINA_RN_BT(float); else LINA_RN_BT(bf16_t);
#undef LINA_RN_BT
#undef LINA_RN_B
    return check_launch("lina_rmsnorm_gate_bwd");
}
