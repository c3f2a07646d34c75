// linear_skinny.hip -- decode-step projections: out[M,N] = epilogue(A[M,K] . W[N,K]^T), M ~ batch (64).
//
// The decode step of SURVEY.md 8(a) a-7/a-8 runs ~56 projections per token with M = batch rows;
// at M = 64 they are weight-streaming, launch-latency-bound kernels (rocprof r01a: 10-12 us each
// through the library GEMM, 43% of the step).  This kernel streams each weight row exactly once
// and folds the surrounding elementwise work into the same launch:
//   * LayerNorm of A folded in algebraically:  LN(a).W^T = rstd*(a.W'^T - mu*c1) + c2 with
//     W' = gamma (*) W, c1[n] = sum_k W'[n,k], c2[n] = sum_k beta_k W[n,k] (+ bias) precomputed once;
//     mu / rstd of every row come out of the same pass over A (reference base_blocks.py:65-69);
//   * bias, residual add (out = resid + ...), and the SwiGLU gate silu(a)*b of base_blocks.py:48-50
//     (the workgroup computes column n of both halves), incl. the constant-1 bias column.
// Work split: grid = (ceil(N/16), ceil(M/64)); 1024 threads = 16 waves; a workgroup owns a 64 x 16
// output tile; wave (mt, kq) takes the 16-row m-tile mt and the k-steps kq, kq+4, ... (in-workgroup
// split-K over 4 quarters), loading its operands straight from global memory into MFMA fragment layout
// (16 B per lane) with all of its loads (8 k-steps) in flight at once -- the kernel is pure load latency,
// so the point is ONE round trip per workgroup.  The four partial accumulators per m-tile are reduced
// through LDS; waves with kq = 0 run the epilogue.
// bf16: v_mfma_f32_16x16x32_bf16 (k-step 32);  f32: v_mfma_f32_16x16x4_f32 x4 (k-step 16).
#include <lina_dev.h>
#include "lina_common.h"

namespace lina {

template <typename T> struct Frag;
template <> struct Frag<bf16_t> {
    static constexpr int KSTEP = 32, KL = 8;  // k per step / k per lane
    uint4 v;
    __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const uint4*>(p); }
    __device__ __forceinline__ void zero() { v = make_uint4(0u, 0u, 0u, 0u); }
    __device__ __forceinline__ void stats(float& s1, float& s2) const {
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = bf2f((bf16_t)(w[j] & 0xffff)), b = bf2f((bf16_t)(w[j] >> 16));
            s1 += a + b;
            s2 += a * a + b * b;
        }
    }
    __device__ __forceinline__ bf16x8 pack() const {
        bf16x8 r;
        r[0] = (short)(v.x & 0xffff); r[1] = (short)(v.x >> 16); r[2] = (short)(v.y & 0xffff); r[3] = (short)(v.y >> 16);
        r[4] = (short)(v.z & 0xffff); r[5] = (short)(v.z >> 16); r[6] = (short)(v.w & 0xffff); r[7] = (short)(v.w >> 16);
        return r;
    }
    static __device__ __forceinline__ f32x4 mma(const Frag& a, const Frag& b, f32x4 c) {
        return mfma_bf16_16x16x32(a.pack(), b.pack(), c);
    }
};
template <> struct Frag<float> {
    static constexpr int KSTEP = 16, KL = 4;
    float4 v;
    __device__ __forceinline__ void load(const float* p) { v = *reinterpret_cast<const float4*>(p); }
    __device__ __forceinline__ void zero() { v = make_float4(0.f, 0.f, 0.f, 0.f); }
    __device__ __forceinline__ void stats(float& s1, float& s2) const {
        s1 += (v.x + v.y) + (v.z + v.w);
        s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    // k-slot (step s, lane group g) <-> k0 + 4g + s on both operands: any bijection is valid
    static __device__ __forceinline__ f32x4 mma(const Frag& a, const Frag& b, f32x4 c) {
        c = mfma_f32_16x16x4(a.v.x, b.v.x, c);
        c = mfma_f32_16x16x4(a.v.y, b.v.y, c);
        c = mfma_f32_16x16x4(a.v.z, b.v.z, c);
        return mfma_f32_16x16x4(a.v.w, b.v.w, c);
    }
};

template <typename T, bool SWIGLU, bool LN>
__global__ __launch_bounds__(1024) void linear_skinny_kernel(
    const T* __restrict__ A, int64_t lda, const T* __restrict__ W, int64_t ldw, const float* __restrict__ c1,
    const float* __restrict__ c2, const T* resid, int64_t ldr, T* out, int64_t ldo, int M, int N, int K, int Hd,
    int ln_dim, float ln_eps) {
    using F = Frag<T>;
    constexpr int NB = SWIGLU ? 2 : 1;
    constexpr int U = 8;   // k-steps whose loads are in flight together: (1 + NB) * U 16-byte loads per lane
    __shared__ __attribute__((aligned(16))) float s_acc[16][NB][64][4];   // [wave][half][lane][reg]
    __shared__ float s_st[4][64][2];                                       // [k-quarter][row][sum, sumsq]

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int mt = w & 3, kq = w >> 2;          // wave = (16-row m-tile, K quarter)
    const int li = lane & 15, lg = lane >> 4;
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 64;
    const int n = n0 + li;
    const int n_rows = SWIGLU ? Hd : N;          // weight rows per half
    const bool n_ok = n < n_rows;
    const int m = m0 + 16 * mt + li;
    const bool m_ok = m < M;

    f32x4 acc[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float s1 = 0.f, s2 = 0.f;

    const T* wp = W + (int64_t)(n_ok ? n : 0) * ldw + F::KL * lg;
    const T* wp2 = SWIGLU ? W + (int64_t)(n_ok ? Hd + n : 0) * ldw + F::KL * lg : nullptr;
    const T* ap = A + (int64_t)(m_ok ? m : 0) * lda + F::KL * lg;

    const int nsteps = K / F::KSTEP;
    for (int ks0 = kq; ks0 < nsteps; ks0 += 4 * U) {
        F fb[U], fb2[U], fa[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ks = ks0 + 4 * u;
            const int k0 = ks * F::KSTEP;
            const bool k_ok = ks < nsteps;
            if (k_ok && n_ok) fb[u].load(wp + k0); else fb[u].zero();
            if (SWIGLU) { if (k_ok && n_ok) fb2[u].load(wp2 + k0); else fb2[u].zero(); }
            if (k_ok && m_ok) fa[u].load(ap + k0); else fa[u].zero();
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (LN) fa[u].stats(s1, s2);
            acc[0] = F::mma(fa[u], fb[u], acc[0]);
            if (SWIGLU) acc[1] = F::mma(fa[u], fb2[u], acc[1]);
        }
    }

    // ---- in-workgroup split-K reduction over the 4 K quarters ----
#pragma unroll
    for (int i = 0; i < NB; ++i)
        *reinterpret_cast<float4*>(&s_acc[w][i][lane][0]) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
    if (LN) {   // sum over the 4 lane groups that share row li
        s1 += shfl_xor(s1, 16); s2 += shfl_xor(s2, 16);
        s1 += shfl_xor(s1, 32); s2 += shfl_xor(s2, 32);
        if (lg == 0) { s_st[kq][16 * mt + li][0] = s1; s_st[kq][16 * mt + li][1] = s2; }
    }
    __syncthreads();
    if (kq != 0) return;   // waves 0..3 finalise m-tile mt: D layout -> rows m0 + 16mt + 4*lg + r, column n

    float val[NB][4];
#pragma unroll
    for (int hb = 0; hb < NB; ++hb) {
        float4 t = *reinterpret_cast<const float4*>(&s_acc[mt][hb][lane][0]);
#pragma unroll
        for (int q4 = 1; q4 < 4; ++q4) {
            const float4 u = *reinterpret_cast<const float4*>(&s_acc[mt + 4 * q4][hb][lane][0]);
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        val[hb][0] = t.x; val[hb][1] = t.y; val[hb][2] = t.z; val[hb][3] = t.w;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 16 * mt + 4 * lg + r;
        const int mo = m0 + row;
        float mu = 0.f, rstd = 1.f;
        if (LN) {
            const float a = (s_st[0][row][0] + s_st[1][row][0]) + (s_st[2][row][0] + s_st[3][row][0]);
            const float b = (s_st[0][row][1] + s_st[1][row][1]) + (s_st[2][row][1] + s_st[3][row][1]);
            mu = a / (float)ln_dim;
            rstd = rsqrtf(fmaxf(b / (float)ln_dim - mu * mu, 0.f) + ln_eps);
        }
        float res;
        if (SWIGLU) {
            float ga = val[0][r], gb = val[1][r];
            if (n_ok) {
                if (LN) { ga = rstd * (ga - mu * c1[n]); gb = rstd * (gb - mu * c1[Hd + n]); }
                if (c2) { ga += c2[n]; gb += c2[Hd + n]; }
            }
            res = n_ok ? silu(ga) * gb : ((n == Hd) ? 1.0f : 0.0f);   // bias column of the K-padded row
        } else {
            res = val[0][r];
            if (n_ok) {
                if (LN) res = rstd * (res - mu * c1[n]);
                if (c2) res += c2[n];
            }
        }
        if (mo < M && n < N) {
            if (resid) res += ld(resid + (int64_t)mo * ldr + n);
            st(out + (int64_t)mo * ldo + n, res);
        }
    }
}

}  // namespace lina

extern "C" int lina_linear_skinny(const void* A, int64_t lda, const void* W, int64_t ldw, const float* c1,
                                  const float* c2, const void* resid, int64_t ldr, void* out, int64_t ldo, int M,
                                  int N, int K, int swiglu_hidden, int ln_dim, float ln_eps, int dtype,
                                  lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(A && W && out, "lina_linear_skinny: null pointer");
    LINA_REQUIRE(M > 0 && N > 0 && K > 0, "lina_linear_skinny: M,N,K must be positive");
    LINA_REQUIRE(valid_dtype(dtype), "lina_linear_skinny: bad dtype %d", dtype);
    const int kstep = dtype == LINA_BF16 ? 32 : 16, al = dtype == LINA_BF16 ? 8 : 4;
    LINA_REQUIRE(K % kstep == 0, "lina_linear_skinny: K=%d must be a multiple of %d (pad the operands)", K, kstep);
    LINA_REQUIRE(lda % al == 0 && ldw % al == 0, "lina_linear_skinny: lda/ldw must keep rows 16-byte aligned");
    LINA_REQUIRE(ln_dim >= 0 && (ln_dim == 0 || c1), "lina_linear_skinny: LayerNorm folding needs c1");
    LINA_REQUIRE(swiglu_hidden >= 0 && swiglu_hidden <= N, "lina_linear_skinny: bad swiglu_hidden");
    dim3 grid((unsigned)((N + 15) / 16), (unsigned)((M + 63) / 64));
    const bool sw = swiglu_hidden > 0, ln = ln_dim > 0;
#define LINA_LS(TT, SW, LNN)                                                                                        \
    LINA_LAUNCH((linear_skinny_kernel<TT, SW, LNN>), grid, dim3(1024), 0, stream, (const TT*)A, lda, (const TT*)W,   \
                ldw, c1, c2, (const TT*)resid, ldr, (TT*)out, ldo, M, N, K, swiglu_hidden, ln_dim, ln_eps)
    if (dtype == LINA_BF16) {
        if (sw && ln) LINA_LS(bf16_t, true, true); else if (sw) LINA_LS(bf16_t, true, false);
        else if (ln) LINA_LS(bf16_t, false, true); else LINA_LS(bf16_t, false, false);
    } else {
        if (sw && ln) LINA_LS(float, true, true); else if (sw) LINA_LS(float, true, false);
        else if (ln) LINA_LS(float, false, true); else LINA_LS(float, false, false);
    }
#undef LINA_LS
    return check_launch("lina_linear_skinny");
}
