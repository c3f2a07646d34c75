// decode_fused.hip -- fused elementwise stretches of the single-token decode step.
//
//  * lina_gla_decode_prologue: the three conv steps (K4) on the q/k/v slices of the fused
//    projection row and the gate prologue (K7) incl. the rank-R up-projection, one launch
//    instead of ~12 (reference model/gla.py:158-163 and :174-180 at T = 1).
//  * lina_swiglu: SwiGLU gate of the channel mixer (reference model/base_blocks.py:48-50; a-8),
//    writing a K-padded row whose extra column carries the constant 1 that folds p_out's bias
//    into the down-projection GEMM.
// Both are pure streaming kernels: lanes along the contiguous channel dimension.
#include <lina_dev.h>
#include "lina_common.h"

namespace lina {

template <typename T>
__global__ __launch_bounds__(256) void decode_prologue_kernel(
    const T* __restrict__ z, int64_t ldz, int off_q, int off_k, int off_v, int off_lr,
    const T* __restrict__ wq, const T* __restrict__ wk, const T* __restrict__ wv, T* cq, T* ck, T* cv,
    const T* __restrict__ w2, const T* __restrict__ b2, T* __restrict__ qkv, float* __restrict__ gk,
    int Kd, int Vd, int R, float inv_norm, float clamp_min, int has_clamp) {
    constexpr int W = 4;
    const int b = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int nconv = 2 * Kd + Vd;
    const T* zb = z + (int64_t)b * ldz;
    if (idx < nconv) {
        const T* wsel; T* csel; int c, D, off;
        if (idx < Kd) { c = idx; D = Kd; wsel = wq; csel = cq; off = off_q; }
        else if (idx < 2 * Kd) { c = idx - Kd; D = Kd; wsel = wk; csel = ck; off = off_k; }
        else { c = idx - 2 * Kd; D = Vd; wsel = wv; csel = cv; off = off_v; }
        T* cb = csel + ((int64_t)b * D + c) * W;
        const float4 old = ld4(cb);
        const float4 wj = ld4(wsel + (int64_t)c * W);
        const float xn = ld(zb + off + c);
        const float4 nw = make_float4(old.y, old.z, old.w, xn);
        st4(cb, nw);
        const float acc = fmaf(wj.w, nw.w, fmaf(wj.z, nw.z, fmaf(wj.y, nw.y, wj.x * nw.x)));
        st(qkv + (int64_t)b * nconv + idx, silu(acc));
    } else if (idx < nconv + Kd) {
        const int c = idx - nconv;
        float acc = ld(b2 + c);
        if ((R & 3) == 0 && (off_lr & 3) == 0 && (ldz & 3) == 0) {   // 8/16-byte loads of the rank-R row and activations
            for (int r = 0; r < R; r += 4) {
                const float4 zv = ld4(zb + off_lr + r), wv4 = ld4(w2 + (int64_t)c * R + r);
                acc = fmaf(zv.x, wv4.x, acc); acc = fmaf(zv.y, wv4.y, acc);
                acc = fmaf(zv.z, wv4.z, acc); acc = fmaf(zv.w, wv4.w, acc);
            }
        } else {
            for (int r = 0; r < R; ++r) acc = fmaf(ld(zb + off_lr + r), ld(w2 + (int64_t)c * R + r), acc);
        }
        float gv = logsigmoidf(acc) * inv_norm;
        if (has_clamp) gv = fmaxf(gv, clamp_min);
        gk[(int64_t)b * Kd + c] = gv;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void swiglu_kernel(const T* __restrict__ u, T* __restrict__ y, int64_t rows, int Hd,
                                                     int64_t ld_u, int64_t ld_y) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= ld_y) return;
    for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {        // (grid.y is capped: training calls have > 65535 rows)
        float out;
        if (j < Hd) {
            const float gate = ld(u + r * ld_u + j), val = ld(u + r * ld_u + Hd + j);
            out = silu(gate) * val;
        } else {
            out = (j == Hd) ? 1.0f : 0.0f;
        }
        st(y + r * ld_y + j, out);
    }
}

// The same gate for a whole sequence of rows (the train path: 32768 rows x 1408 at config 5): a workgroup takes 128 rows x
// 256 gate columns, its 4 waves a quarter of the rows each (4 at a time, all loads first), a lane 4 columns -- 8-byte loads
// and stores where the kernel above moves one element per thread (95 us for 277 MB there).  Hd and the strides must be
// multiples of 4 and ld_y == Hd (no bias column).
template <typename T>
__global__ __launch_bounds__(256) void swiglu_rows_kernel(const T* __restrict__ u, T* __restrict__ y, int64_t rows, int Hd,
                                                          int64_t ld_u, int64_t ld_y) {
    constexpr int U = 4, RW = 32;                                    // rows per round, rows per wave
    const int lane = threadIdx.x & 63, wv = wave_uniform(threadIdx.x >> 6);
    const int j = (blockIdx.y * 64 + lane) * 4;
    const bool ok = j < Hd;
    const int jc = ok ? j : Hd - 4;
    const int64_t r_begin = ((int64_t)blockIdx.x * 4 + wv) * RW;
    const int64_t r_end = r_begin + RW < rows ? r_begin + RW : rows;
    for (int64_t r0 = r_begin; r0 < r_end; r0 += U) {
        typename raw4<T>::type ar[U], br[U];
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int64_t r = r0 + q < rows ? r0 + q : rows - 1;
            ar[q] = ld4_raw(u + r * ld_u + jc);
            br[q] = ld4_raw(u + r * ld_u + Hd + jc);
        }
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const float4 a = cvt4(ar[q]), b = cvt4(br[q]);
            const float4 o = make_float4(silu(a.x) * b.x, silu(a.y) * b.y, silu(a.z) * b.z, silu(a.w) * b.w);
            if (ok && r0 + q < r_end) st4(y + (r0 + q) * ld_y + j, o);
        }
    }
}

}  // namespace lina

extern "C" int lina_gla_decode_prologue(const void* z, int64_t ldz, int off_q, int off_k, int off_v, int off_lr,
                                        const void* wq, const void* wk, const void* wv, void* cq, void* ck, void* cv,
                                        const void* w2, const void* b2, void* qkv, float* gk, int B, int Kd, int Vd,
                                        int W, int R, float normalizer, float clamp_min, int dtype,
                                        lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(z && wq && wk && wv && cq && ck && cv && w2 && b2 && qkv && gk, "lina_gla_decode_prologue: null pointer");
    LINA_REQUIRE(B > 0 && Kd > 0 && Vd > 0, "lina_gla_decode_prologue: B,Kd,Vd must be positive");
    LINA_REQUIRE(R > 0 && R <= 32, "lina_gla_decode_prologue: R=%d not in 1..32", R);
    LINA_REQUIRE(valid_dtype(dtype), "lina_gla_decode_prologue: bad dtype %d", dtype);
    if (W != 4) return fail(LINA_ERR_UNSUPPORTED, "lina_gla_decode_prologue: conv width W=%d (only 4 is built)", W);
    LINA_REQUIRE(normalizer != 0.0f, "lina_gla_decode_prologue: normalizer must be non-zero");
    const int has_clamp = (clamp_min == clamp_min) ? 1 : 0;  // NaN = no clamp
    dim3 grid((unsigned)((2 * Kd + Vd + Kd + 255) / 256), (unsigned)B);
    if (dtype == LINA_F32)
        LINA_LAUNCH((decode_prologue_kernel<float>), grid, dim3(256), 0, stream, (const float*)z, ldz, off_q, off_k,
                    off_v, off_lr, (const float*)wq, (const float*)wk, (const float*)wv, (float*)cq, (float*)ck,
                    (float*)cv, (const float*)w2, (const float*)b2, (float*)qkv, gk, Kd, Vd, R, 1.0f / normalizer,
                    clamp_min, has_clamp);
    else
        LINA_LAUNCH((decode_prologue_kernel<bf16_t>), grid, dim3(256), 0, stream, (const bf16_t*)z, ldz, off_q, off_k,
                    off_v, off_lr, (const bf16_t*)wq, (const bf16_t*)wk, (const bf16_t*)wv, (bf16_t*)cq, (bf16_t*)ck,
                    (bf16_t*)cv, (const bf16_t*)w2, (const bf16_t*)b2, (bf16_t*)qkv, gk, Kd, Vd, R, 1.0f / normalizer,
                    clamp_min, has_clamp);
    return check_launch("lina_gla_decode_prologue");
}

extern "C" int lina_swiglu(const void* u, void* y, int64_t rows, int Hd, int64_t ld_u, int64_t ld_y, int dtype,
                           lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(u && y, "lina_swiglu: null pointer");
    LINA_REQUIRE(rows > 0 && Hd > 0 && ld_u >= 2 * (int64_t)Hd && ld_y >= Hd, "lina_swiglu: bad shape");
    LINA_REQUIRE(valid_dtype(dtype), "lina_swiglu: bad dtype %d", dtype);
    if (rows >= 1024 && Hd % 4 == 0 && ld_u % 4 == 0 && ld_y == Hd && rows <= (int64_t)65535 * 128) {   // sequence form
        dim3 grid_r((unsigned)((rows + 127) / 128), (unsigned)((Hd + 255) / 256));
        if (dtype == LINA_F32)
            LINA_LAUNCH((swiglu_rows_kernel<float>), grid_r, dim3(256), 0, stream, (const float*)u, (float*)y, rows, Hd, ld_u, ld_y);
        else
            LINA_LAUNCH((swiglu_rows_kernel<bf16_t>), grid_r, dim3(256), 0, stream, (const bf16_t*)u, (bf16_t*)y, rows, Hd, ld_u,
                        ld_y);
        return check_launch("lina_swiglu");
    }
    dim3 grid((unsigned)((ld_y + 255) / 256), (unsigned)(rows < 32768 ? rows : 32768));
    if (dtype == LINA_F32)
        LINA_LAUNCH((swiglu_kernel<float>), grid, dim3(256), 0, stream, (const float*)u, (float*)y, rows, Hd, ld_u, ld_y);
    else
        LINA_LAUNCH((swiglu_kernel<bf16_t>), grid, dim3(256), 0, stream, (const bf16_t*)u, (bf16_t*)y, rows, Hd, ld_u, ld_y);
    return check_launch("lina_swiglu");
}
