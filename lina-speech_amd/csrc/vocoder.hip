// vocoder.hip -- the two streaming kernels of the codes -> waveform step that follows the generation path
// (reference 3rdparty/decoder/modules.py:44-55, spectral_ops.py:36-75).  Channels-last
// activations [B, L, C] throughout (the reference transposes between [B,C,L] convolutions and [B,L,C] norms).
//
// K8  dwconv7 + LayerNorm: y[b,t,:] = LN_C( bias + sum_{j<7} w[:,j] * x[b, t-3+j, :] ) * scale[b or 0] + shift
//     -- ConvNeXtBlock.dwconv ("same" zero padding) fused with the LayerNorm / AdaLayerNorm that follows it
//     (modules.py:44-50, 62-82).  One wave per (b,t) row, lanes along the contiguous channel dimension, the 7
//     neighbouring rows are read straight from global memory (L2-resident re-reads), mean/variance by wave
//     shuffles.  HBM-bound streaming: (1 read + 1 write) * C * e bytes per row.
// K9  ISTFT overlap-add with "same" padding: every output sample gathers its <= ceil(win/hop) overlapping
//     windowed frames and divides by the window envelope (spectral_ops.py:56-75: two torch `fold`s and a
//     divide) -- no atomics, one pass, one thread per sample.
#include <lina_dev.h>
#include "lina_common.h"

namespace lina {

constexpr int kDwMaxPer = 16;   // channels per lane: C <= 1024

template <typename T>
__global__ __launch_bounds__(256) void dwconv7_ln_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                         const T* __restrict__ bias, const T* __restrict__ scale,
                                                         const T* __restrict__ shift, T* __restrict__ y, int64_t rows,
                                                         int L, int C, int64_t scale_sb, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;                                  // whole wave
    const int64_t b = row / L;
    const int t = (int)(row % L);
    const int per = (C + 63) / 64;
    float z[kDwMaxPer];
    float s1 = 0.0f;
#pragma unroll
    for (int i = 0; i < kDwMaxPer; ++i) {
        const int c = lane + 64 * i;
        z[i] = 0.0f;
        if (i < per && c < C) {
            float acc = bias ? ld(bias + c) : 0.0f;
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const int tt = t - 3 + j;
                const float xv = (tt >= 0 && tt < L) ? ld(x + ((int64_t)b * L + tt) * C + c) : 0.0f;
                acc = fmaf(ld(w + (int64_t)c * 7 + j), xv, acc);
            }
            z[i] = acc;
            s1 += acc;
        }
    }
    s1 += shfl_xor(s1, 1); s1 += shfl_xor(s1, 2); s1 += shfl_xor(s1, 4);
    s1 += shfl_xor(s1, 8); s1 += shfl_xor(s1, 16); s1 += shfl_xor(s1, 32);
    const float mu = s1 / (float)C;
    float s2 = 0.0f;
#pragma unroll
    for (int i = 0; i < kDwMaxPer; ++i) {
        const int c = lane + 64 * i;
        if (i < per && c < C) { const float d = z[i] - mu; s2 += d * d; }
    }
    s2 += shfl_xor(s2, 1); s2 += shfl_xor(s2, 2); s2 += shfl_xor(s2, 4);
    s2 += shfl_xor(s2, 8); s2 += shfl_xor(s2, 16); s2 += shfl_xor(s2, 32);
    const float rstd = rsqrtf(s2 / (float)C + eps);
#pragma unroll
    for (int i = 0; i < kDwMaxPer; ++i) {
        const int c = lane + 64 * i;
        if (i < per && c < C) {
            float v = (z[i] - mu) * rstd;
            if (scale) v *= ld(scale + b * scale_sb + c);
            if (shift) v += ld(shift + b * scale_sb + c);
            st(y + row * C + c, v);
        }
    }
}

// frames [B, T, win] (already inverse-transformed, NOT yet windowed), window [win] fp32, y [B, n_out] fp32,
// n_out = (T-1)*hop + win - 2*pad
__global__ __launch_bounds__(256) void istft_ola_kernel(const float* __restrict__ frames, const float* __restrict__ window,
                                                        float* __restrict__ y, int T, int win, int hop, int pad,
                                                        int n_out) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (n >= n_out) return;
    const int p = n + pad;                                    // position in the un-trimmed overlap-add buffer
    int t_hi = p / hop;
    if (t_hi > T - 1) t_hi = T - 1;
    float acc = 0.0f, env = 0.0f;
    for (int t = t_hi; t >= 0; --t) {
        const int k = p - t * hop;
        if (k >= win) break;
        const float wv = window[k];
        acc = fmaf(frames[((int64_t)b * T + t) * win + k], wv, acc);
        env = fmaf(wv, wv, env);
    }
    y[(int64_t)b * n_out + n] = acc / env;
}

}  // namespace lina

extern "C" int lina_dwconv7_ln(const void* x, const void* w, const void* bias, const void* scale, const void* shift,
                               void* y, int B, int L, int C, int64_t scale_sb, float eps, int dtype,
                               lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(x && w && y, "lina_dwconv7_ln: null pointer");
    LINA_REQUIRE(B > 0 && L > 0 && C > 0, "lina_dwconv7_ln: B,L,C must be positive (got %d,%d,%d)", B, L, C);
    LINA_REQUIRE(valid_dtype(dtype), "lina_dwconv7_ln: bad dtype %d", dtype);
    if (C > 64 * kDwMaxPer) return fail(LINA_ERR_UNSUPPORTED, "lina_dwconv7_ln: C=%d exceeds %d", C, 64 * kDwMaxPer);
    const int64_t rows = (int64_t)B * L;
    dim3 grid((unsigned)((rows + 3) / 4));
    if (dtype == LINA_F32)
        LINA_LAUNCH((dwconv7_ln_kernel<float>), grid, dim3(256), 0, stream, (const float*)x, (const float*)w,
                    (const float*)bias, (const float*)scale, (const float*)shift, (float*)y, rows, L, C, scale_sb, eps);
    else
        LINA_LAUNCH((dwconv7_ln_kernel<bf16_t>), grid, dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)w,
                    (const bf16_t*)bias, (const bf16_t*)scale, (const bf16_t*)shift, (bf16_t*)y, rows, L, C, scale_sb,
                    eps);
    return check_launch("lina_dwconv7_ln");
}

extern "C" int lina_istft_ola(const float* frames, const float* window, float* y, int B, int T, int win, int hop,
                              lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(frames && window && y, "lina_istft_ola: null pointer");
    LINA_REQUIRE(B > 0 && T > 0 && win > 0 && hop > 0 && hop <= win, "lina_istft_ola: bad shape");
    const int pad = (win - hop) / 2;
    const int n_out = (T - 1) * hop + win - 2 * pad;
    LINA_REQUIRE(n_out > 0, "lina_istft_ola: empty output");
    dim3 grid((unsigned)((n_out + 255) / 256), (unsigned)B);
    LINA_LAUNCH(istft_ola_kernel, grid, dim3(256), 0, stream, frames, window, y, T, win, hop, pad, n_out);
    return check_launch("lina_istft_ola");
}
