// gla_inproj.hip -- the WHOLE input side of one GLA mixer at T = 1 in a single launch:
//   LayerNorm-1 (folded) -> fused projection q|k|v|g|gate-low-rank -> per-channel epilogues:
//     q,k,v tiles : causal short-conv step on the rolled cache + SiLU          (reference model/gla.py:158-163)
//     g tiles     : stored for the output gate                                  (model/gla.py:216)
//     gate tiles  : the 16 low-rank activations of the tile's rows are exchanged through LDS and the
//                   rank-16 up-projection + bias + logsigmoid / normaliser (+clamp) is applied (:174-180)
// i.e. lina_linear_skinny + lina_gla_decode_prologue without the z round trip or the second launch.
// Same work split and main loop as linear_skinny.hip (64 x 16 output tile per 256-thread workgroup,
// in-workgroup split-K over 4 waves, operands loaded in MFMA fragment layout).  Gate tiles multiply by
// the low-rank weight rows (one n-tile) instead of a 16-column slice of the big matrix.
#include <lina_dev.h>
#include "lina_common.h"
#include "skinny_frag.h"

namespace lina {

template <typename T>
__global__ __launch_bounds__(256) void gla_inproj_kernel(
    const T* __restrict__ A, int64_t lda, const T* __restrict__ W, int64_t ldw, const float* __restrict__ c1,
    const float* __restrict__ c2, const T* __restrict__ wq, const T* __restrict__ wk, const T* __restrict__ wv,
    T* cq, T* ck, T* cv, const T* __restrict__ w2, const T* __restrict__ b2, T* __restrict__ qkv,
    T* __restrict__ g_out, float* __restrict__ gk, int M, int K, int Kd, int Vd, float ln_eps, float inv_norm,
    float clamp_min, int has_clamp) {
    using F = Frag<T>;
    constexpr int R = 16;
    __shared__ __attribute__((aligned(16))) float s_acc[4][4][64][4];
    __shared__ float s_st[4][64][2];
    __shared__ float s_lr[64][R + 1];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int m0 = blockIdx.y * 64;
    const int n_direct = 2 * Kd + 2 * Vd;               // q | k | v | g columns; low-rank rows follow in W
    const int tile0 = blockIdx.x * 16;
    const bool gate_tile = tile0 >= n_direct;           // block-uniform
    const int wrow = gate_tile ? n_direct + li : tile0 + li;

    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    const T* wp = W + (int64_t)wrow * ldw + F::KL * lg;
    const T* ap[4];
    bool m_ok[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = m0 + 16 * mt + li;
        m_ok[mt] = m < M;
        ap[mt] = A + (int64_t)(m_ok[mt] ? m : 0) * lda + F::KL * lg;
    }
    const int nsteps = K / F::KSTEP;
    constexpr int U = 8;
    int ks = w;
    for (; ks + 4 * (U - 1) < nsteps; ks += 4 * U) {
        F fb[U], fa[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k0 = (ks + 4 * u) * F::KSTEP;
            fb[u].load(wp + k0);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) { if (m_ok[mt]) fa[u][mt].load(ap[mt] + k0); else fa[u][mt].zero(); }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                fa[u][mt].stats(s1[mt], s2[mt]);
                acc[mt] = F::mma(fa[u][mt], fb[u], acc[mt]);
            }
    }
    for (; ks < nsteps; ks += 4) {
        const int k0 = ks * F::KSTEP;
        F fb, fa[4];
        fb.load(wp + k0);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) { if (m_ok[mt]) fa[mt].load(ap[mt] + k0); else fa[mt].zero(); }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            fa[mt].stats(s1[mt], s2[mt]);
            acc[mt] = F::mma(fa[mt], fb, acc[mt]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        *reinterpret_cast<float4*>(&s_acc[w][i][lane][0]) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        float a = s1[mt], b = s2[mt];
        a += shfl_xor(a, 16); b += shfl_xor(b, 16);
        a += shfl_xor(a, 32); b += shfl_xor(b, 32);
        if (lg == 0) { s_st[w][16 * mt + li][0] = a; s_st[w][16 * mt + li][1] = b; }
    }
    __syncthreads();

    // wave w finalises m-tile w (rows m0 + 16w + 4lg + r, column li of the tile)
    float val[4];
    {
        float4 t = *reinterpret_cast<const float4*>(&s_acc[0][w][lane][0]);
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) {
            const float4 u = *reinterpret_cast<const float4*>(&s_acc[ww][w][lane][0]);
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        val[0] = t.x; val[1] = t.y; val[2] = t.z; val[3] = t.w;
    }
    const float cc1 = c1[wrow], cc2 = c2[wrow];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 16 * w + 4 * lg + r;
        const float a = (s_st[0][row][0] + s_st[1][row][0]) + (s_st[2][row][0] + s_st[3][row][0]);
        const float b = (s_st[0][row][1] + s_st[1][row][1]) + (s_st[2][row][1] + s_st[3][row][1]);
        const float mu = a / (float)K;
        const float rstd = rsqrtf(fmaxf(b / (float)K - mu * mu, 0.f) + ln_eps);
        val[r] = rstd * (val[r] - mu * cc1) + cc2;      // the projected value z[m, wrow]
    }

    if (gate_tile) {
#pragma unroll
        for (int r = 0; r < 4; ++r) s_lr[16 * w + 4 * lg + r][li] = val[r];
        __syncthreads();
        const int c = (tile0 - n_direct) + li;          // gate channel
        float w2r[R];
#pragma unroll
        for (int j = 0; j < R; ++j) w2r[j] = ld(w2 + (int64_t)c * R + j);
        const float bias = ld(b2 + c);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * w + 4 * lg + r, m = m0 + row;
            float accg = bias;
#pragma unroll
            for (int j = 0; j < R; ++j) accg = fmaf(s_lr[row][j], w2r[j], accg);
            float gv = logsigmoidf(accg) * inv_norm;
            if (has_clamp) gv = fmaxf(gv, clamp_min);
            if (m < M) gk[(int64_t)m * Kd + c] = gv;
        }
        return;
    }
    const int n = tile0 + li;
    if (n >= 2 * Kd + Vd) {                              // g columns
        const int c = n - (2 * Kd + Vd);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 16 * w + 4 * lg + r;
            if (m < M) st(g_out + (int64_t)m * Vd + c, val[r]);
        }
        return;
    }
    // q / k / v columns: conv step on the rolled cache (W = 4) + SiLU
    const T* wsel; T* csel; int c, D;
    if (n < Kd) { c = n; D = Kd; wsel = wq; csel = cq; }
    else if (n < 2 * Kd) { c = n - Kd; D = Kd; wsel = wk; csel = ck; }
    else { c = n - 2 * Kd; D = Vd; wsel = wv; csel = cv; }
    const float4 wj = ld4(wsel + (int64_t)c * 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + 16 * w + 4 * lg + r;
        if (m < M) {
            T* cb = csel + ((int64_t)m * D + c) * 4;
            const float4 old = ld4(cb);
            float xn = val[r];
            T tmp;                                       // the conv sees the projection in the model dtype
            st(&tmp, xn);
            xn = ld(&tmp);
            const float4 nw = make_float4(old.y, old.z, old.w, xn);
            st4(cb, nw);
            const float y = fmaf(wj.w, nw.w, fmaf(wj.z, nw.z, fmaf(wj.y, nw.y, wj.x * nw.x)));
            st(qkv + (int64_t)m * (2 * Kd + Vd) + n, silu(y));
        }
    }
}

}  // namespace lina

extern "C" int lina_gla_decode_inproj(const void* x, int64_t ldx, const void* w_in, int64_t ldw, const float* c1,
                                      const float* c2, const void* wq, const void* wk, const void* wv, void* cq,
                                      void* ck, void* cv, const void* w2, const void* b2, void* qkv, void* g_out,
                                      float* gk, int B, int K, int Kd, int Vd, int W, int R, float ln_eps,
                                      float normalizer, float clamp_min, int dtype, lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(x && w_in && c1 && c2 && wq && wk && wv && cq && ck && cv && w2 && b2 && qkv && g_out && gk,
                 "lina_gla_decode_inproj: null pointer");
    LINA_REQUIRE(B > 0 && K > 0, "lina_gla_decode_inproj: B,K must be positive");
    LINA_REQUIRE(valid_dtype(dtype), "lina_gla_decode_inproj: bad dtype %d", dtype);
    if (W != 4 || R != 16) return fail(LINA_ERR_UNSUPPORTED, "lina_gla_decode_inproj: needs conv width 4 and gate rank 16 (got %d, %d)", W, R);
    if (Kd <= 0 || Vd <= 0 || Kd % 16 || Vd % 16) return fail(LINA_ERR_UNSUPPORTED, "lina_gla_decode_inproj: Kd,Vd must be positive multiples of 16");
    const int kstep = dtype == LINA_BF16 ? 32 : 16, al = dtype == LINA_BF16 ? 8 : 4;
    LINA_REQUIRE(K % kstep == 0 && ldx % al == 0 && ldw % al == 0, "lina_gla_decode_inproj: K/ldx/ldw alignment");
    LINA_REQUIRE(normalizer != 0.0f, "lina_gla_decode_inproj: normalizer must be non-zero");
    const int has_clamp = (clamp_min == clamp_min) ? 1 : 0;
    dim3 grid((unsigned)((2 * Kd + 2 * Vd + Kd) / 16), (unsigned)((B + 63) / 64));
    if (dtype == LINA_F32)
        LINA_LAUNCH((gla_inproj_kernel<float>), grid, dim3(256), 0, stream, (const float*)x, ldx, (const float*)w_in, ldw,
                    c1, c2, (const float*)wq, (const float*)wk, (const float*)wv, (float*)cq, (float*)ck, (float*)cv,
                    (const float*)w2, (const float*)b2, (float*)qkv, (float*)g_out, gk, B, K, Kd, Vd, ln_eps,
                    1.0f / normalizer, clamp_min, has_clamp);
    else
        LINA_LAUNCH((gla_inproj_kernel<bf16_t>), grid, dim3(256), 0, stream, (const bf16_t*)x, ldx, (const bf16_t*)w_in,
                    ldw, c1, c2, (const bf16_t*)wq, (const bf16_t*)wk, (const bf16_t*)wv, (bf16_t*)cq, (bf16_t*)ck,
                    (bf16_t*)cv, (const bf16_t*)w2, (const bf16_t*)b2, (bf16_t*)qkv, (bf16_t*)g_out, gk, B, K, Kd, Vd,
                    ln_eps, 1.0f / normalizer, clamp_min, has_clamp);
    return check_launch("lina_gla_decode_inproj");
}
