// vocoder.hip -- the two streaming kernels of the codes -> waveform step that follows the generation path
// (reference 3rdparty/decoder/modules.py:44-55, spectral_ops.py:36-75).  Channels-last
// activations [B, L, C] throughout (the reference transposes between [B,C,L] convolutions and [B,L,C] norms).
//
// K8  dwconv7 + LayerNorm: y[b,t,:] = LN_C( bias + sum_{j<7} w[:,j] * x[b, t-3+j, :] ) * scale[b or 0] + shift
//     -- ConvNeXtBlock.dwconv ("same" zero padding) fused with the LayerNorm / AdaLayerNorm that follows it
//     (modules.py:44-50, 62-82).  One wave per run of 8 consecutive rows of one b, lanes along the contiguous
//     channel dimension, a 7-row sliding window in registers (each x row is loaded once per wave), mean/variance
//     by wave shuffles.  HBM-bound streaming: (1 read + 1 write) * C * e bytes per row.
// K9  ISTFT overlap-add with "same" padding: every output sample gathers its <= ceil(win/hop) overlapping
//     windowed frames and divides by the window envelope (spectral_ops.py:56-75: two torch `fold`s and a
//     divide) -- no atomics, one pass, one thread per sample.
#include <lina_dev.h>
#include "lina_common.h"

namespace lina {

constexpr int kDwMaxPer = 16;   // channels per lane: C <= 1024

constexpr int kDwRows = 8;      // consecutive time steps per wave: every x row is loaded once per wave (7 -> 1.75 reads/row)

template <typename T, int PER>     // PER = channel groups of 64 per lane (C <= 64 * PER)
__global__ __launch_bounds__(256) void dwconv7_ln_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                         const T* __restrict__ bias, const T* __restrict__ scale,
                                                         const T* __restrict__ shift, T* __restrict__ y, int B, int L,
                                                         int C, int64_t scale_sb, float eps) {
    const int lane = threadIdx.x & 63;
    const int nblk = (L + kDwRows - 1) / kDwRows;             // row blocks per batch element
    const int64_t blk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blk >= (int64_t)B * nblk) return;                     // whole wave
    const int b = (int)(blk / nblk);
    const int t0 = (int)(blk % nblk) * kDwRows;
    const T* xb = x + (int64_t)b * L * C;
    // sliding window win[j][i] = x[t - 3 + j][lane + 64 i], taps and the affine rows stay in registers
    float win[7][PER], wv[7][PER], bv[PER], sc[PER], sh[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = lane + 64 * i;
        const bool ok = c < C;
        bv[i] = (ok && bias) ? ld(bias + c) : 0.0f;
        sc[i] = (ok && scale) ? ld(scale + b * scale_sb + c) : 1.0f;
        sh[i] = (ok && shift) ? ld(shift + b * scale_sb + c) : 0.0f;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            wv[j][i] = ok ? ld(w + (int64_t)c * 7 + j) : 0.0f;
            const int tt = t0 - 4 + j;                        // one step behind: the loop shifts before it uses
            win[j][i] = (ok && j > 0 && tt >= 0 && tt < L) ? ld(xb + (int64_t)tt * C + c) : 0.0f;
        }
    }
    const int t1 = min(t0 + kDwRows, L);
    for (int t = t0; t < t1; ++t) {
        float z[PER];
        float s1 = 0.0f;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = lane + 64 * i;
            const bool ok = c < C;
#pragma unroll
            for (int j = 0; j < 6; ++j) win[j][i] = win[j + 1][i];
            win[6][i] = (ok && t + 3 < L) ? ld(xb + (int64_t)(t + 3) * C + c) : 0.0f;
            float acc = bv[i];
#pragma unroll
            for (int j = 0; j < 7; ++j) acc = fmaf(wv[j][i], win[j][i], acc);
            z[i] = ok ? acc : 0.0f;
            s1 += z[i];
        }
        s1 += shfl_xor(s1, 1); s1 += shfl_xor(s1, 2); s1 += shfl_xor(s1, 4);
        s1 += shfl_xor(s1, 8); s1 += shfl_xor(s1, 16); s1 += shfl_xor(s1, 32);
        const float mu = s1 / (float)C;
        float s2 = 0.0f;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = lane + 64 * i;
            if (c < C) { const float d = z[i] - mu; s2 += d * d; }
        }
        s2 += shfl_xor(s2, 1); s2 += shfl_xor(s2, 2); s2 += shfl_xor(s2, 4);
        s2 += shfl_xor(s2, 8); s2 += shfl_xor(s2, 16); s2 += shfl_xor(s2, 32);
        const float rstd = rsqrtf(s2 / (float)C + eps);
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = lane + 64 * i;
            if (c < C) st(y + ((int64_t)b * L + t) * C + c, (z[i] - mu) * rstd * sc[i] + sh[i]);
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
    const int64_t blocks = (int64_t)B * ((L + kDwRows - 1) / kDwRows);
    dim3 grid((unsigned)((blocks + 3) / 4));
#define LINA_DW(TT, PP)                                                                                              \
    LINA_LAUNCH((dwconv7_ln_kernel<TT, PP>), grid, dim3(256), 0, stream, (const TT*)x, (const TT*)w, (const TT*)bias, \
                (const TT*)scale, (const TT*)shift, (TT*)y, B, L, C, scale_sb, eps)
#define LINA_DW_T(TT)                                                                          \
    do {                                                                                       \
        if (C <= 256) LINA_DW(TT, 4); else if (C <= 512) LINA_DW(TT, 8);                       \
        else if (C <= 768) LINA_DW(TT, 12); else LINA_DW(TT, 16);                              \
    } while (0)
    if (dtype == LINA_F32) LINA_DW_T(float); else LINA_DW_T(bf16_t);
#undef LINA_DW_T
#undef LINA_DW
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
