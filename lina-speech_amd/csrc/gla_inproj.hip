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
#include <stdlib.h>

#ifdef LINA_SKINNY_PROF
// tools-only build (tools/skinny_prof.sh): time stamps of thread 0 of every workgroup, [workgroup][slot] (slots as in
// linear_skinny.hip).  NOT part of the product library.
__device__ unsigned long long lina_inproj_prof[1024 * 8];
#define IP_PROF(i, expr) do { if (threadIdx.x == 0) pr_[i] = (expr); } while (0)
#define IP_PROF_FLUSH() do { IP_PROF(5, clock64()); IP_PROF(6, wall_clock64()); if (threadIdx.x == 0 && blockIdx.x < 1024) \
        for (int i_ = 0; i_ < 8; ++i_) lina_inproj_prof[blockIdx.x * 8 + i_] = pr_[i_]; } while (0)
#else
#define IP_PROF(i, expr) do { } while (0)
#define IP_PROF_FLUSH() do { } while (0)
#endif
#include "gla_inproj_body.h"

namespace lina {

template <typename T, int NT, bool PK, bool WNT, int NW = 4>
__global__ __launch_bounds__(64 * NW) void gla_inproj_kernel(
    const T* __restrict__ A, int64_t lda, const T* __restrict__ W, int64_t ldw, const float* __restrict__ c1,
    const float* __restrict__ c2, const T* __restrict__ wq, const T* __restrict__ wk, const T* __restrict__ wv,
    T* cq, T* ck, T* cv, const T* __restrict__ w2, const T* __restrict__ b2, T* __restrict__ qkv,
    T* __restrict__ g_out, float* __restrict__ gk, int M, int K, int Kd, int Vd, float ln_eps, float inv_norm,
    float clamp_min, int has_clamp) {
    __shared__ InprojSmem<NT, NW> sm;
    gla_inproj_body<T, NT, PK, WNT, NW, 0, false>(sm, (int)blockIdx.x, (int)blockIdx.y, A, lda, W, ldw, c1, c2, wq, wk, wv, cq, ck,
                                                  cv, w2, b2, qkv, g_out, gk, M, K, Kd, Vd, ln_eps, inv_norm, clamp_min,
                                                  has_clamp, 1, nullptr);
}

}  // namespace lina


static int inproj_impl(const void* x, int64_t ldx, const void* w_in, int64_t ldw, int packed, const float* c1,
                       const float* c2, const void* wq, const void* wk, const void* wv, void* cq, void* ck, void* cv,
                       const void* w2, const void* b2, void* qkv, void* g_out, float* gk, int B, int K, int Kd, int Vd,
                       int W, int R, float ln_eps, float normalizer, float clamp_min, int dtype, lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(x && w_in && c1 && c2 && wq && wk && wv && cq && ck && cv && w2 && b2 && qkv && g_out && gk,
                 "lina_gla_decode_inproj: null pointer");
    LINA_REQUIRE(B > 0 && K > 0, "lina_gla_decode_inproj: B,K must be positive");
    LINA_REQUIRE(valid_dtype(dtype), "lina_gla_decode_inproj: bad dtype %d", dtype);
    if (W != 4 || R != 16) return fail(LINA_ERR_UNSUPPORTED, "lina_gla_decode_inproj: needs conv width 4 and gate rank 16 (got %d, %d)", W, R);
    if (Kd <= 0 || Vd <= 0 || Kd % 16 || Vd % 16) return fail(LINA_ERR_UNSUPPORTED, "lina_gla_decode_inproj: Kd,Vd must be positive multiples of 16");
    const int kstep = dtype == LINA_BF16 ? 32 : 16, al = dtype == LINA_BF16 ? 8 : 4;
    LINA_REQUIRE(K % kstep == 0 && ((packed & 1) || (ldx % al == 0 && ldw % al == 0)), "lina_gla_decode_inproj: K/ldx/ldw alignment");
    LINA_REQUIRE(normalizer != 0.0f, "lina_gla_decode_inproj: normalizer must be non-zero");
    const int has_clamp = (clamp_min == clamp_min) ? 1 : 0;
    // 64 rows x 32 columns per workgroup when the q|k|v|g regions allow it (fewer, fatter workgroups: one per CU at
    // L169 -- the busiest CU's byte count sets the time, see linear_skinny.hip); else 64 x 16
    static const bool narrow = getenv("LINA_INPROJ_NARROW") != nullptr;      // tuning knob (tools/probe_decode.py)
    const bool wide = Kd % 32 == 0 && Vd % 32 == 0 && !narrow;
    const int cols = wide ? 32 : 16;
    dim3 grid((unsigned)((2 * Kd + 2 * Vd) / cols + Kd / 16), (unsigned)((B + 63) / 64));
    // waves per workgroup (split-K width) of the packed kernel, see linear_skinny.hip; LINA_SKINNY_WAVES overrides
    // Measured in the L169 decode step (tests/gpu_r03e.sh, ms per token): 4 waves 0.660, 8 waves 0.624, 16 waves on the plain
    // 16-column projections + 8 elsewhere 0.617.
    int nw = 16;
    {
        const char* forced_nw = getenv("LINA_SKINNY_WAVES");   // (read per call: tests switch it)
        if (forced_nw) nw = atoi(forced_nw);
        while (nw > 4 && K / kstep < 2 * nw) nw /= 2;
        if (nw == 16 && (wide || dtype == LINA_F32)) nw = 8;   // 16 waves only on the bf16 16-column tiles (registers)
        if (nw != 8 && nw != 16) nw = 4;
    }
#define LINA_INPROJ_PK(TT, NTT, PKK, WNTT, NWW)                                                                      \
    LINA_LAUNCH((gla_inproj_kernel<TT, NTT, PKK, WNTT, NWW>), grid, dim3(64 * NWW), 0, stream, (const TT*)x, ldx,      \
                (const TT*)w_in, ldw, c1, c2, (const TT*)wq, (const TT*)wk, (const TT*)wv, (TT*)cq, (TT*)ck, (TT*)cv,   \
                (const TT*)w2, (const TT*)b2, (TT*)qkv, (TT*)g_out, gk, B, K, Kd, Vd, ln_eps, 1.0f / normalizer,        \
                clamp_min, has_clamp)
#define LINA_INPROJ_NW(TT, NTT, WNTT)                                                                                \
    do {                                                                                                             \
        if (nw == 16 && NTT == 1) LINA_INPROJ_PK(TT, 1, true, WNTT, 16);                                             \
        else if (nw >= 8) LINA_INPROJ_PK(TT, NTT, true, WNTT, 8);                                                    \
        else LINA_INPROJ_PK(TT, NTT, true, WNTT, 4);                                                                 \
    } while (0)
#define LINA_INPROJ(TT, NTT)                                                                                         \
    do {                                                                                                             \
        if (packed == 3) LINA_INPROJ_NW(TT, NTT, true);                                                              \
        else if (packed & 1) LINA_INPROJ_NW(TT, NTT, false);                                                         \
        else LINA_INPROJ_PK(TT, NTT, false, false, 4);                                                               \
    } while (0)
    if (dtype == LINA_F32) { if (wide) LINA_INPROJ(float, 2); else LINA_INPROJ(float, 1); }
    else { if (wide) LINA_INPROJ(bf16_t, 2); else LINA_INPROJ(bf16_t, 1); }
#undef LINA_INPROJ
#undef LINA_INPROJ_NW
#undef LINA_INPROJ_PK
    return check_launch("lina_gla_decode_inproj");
}

extern "C" int lina_gla_decode_inproj(const void* x, int64_t ldx, const void* w_in, int64_t ldw, const float* c1,
                                      const float* c2, const void* wq, const void* wk, const void* wv, void* cq,
                                      void* ck, void* cv, const void* w2, const void* b2, void* qkv, void* g_out,
                                      float* gk, int B, int K, int Kd, int Vd, int W, int R, float ln_eps,
                                      float normalizer, float clamp_min, int dtype, lina_stream_t stream) {
    return inproj_impl(x, ldx, w_in, ldw, 0, c1, c2, wq, wk, wv, cq, ck, cv, w2, b2, qkv, g_out, gk, B, K, Kd, Vd, W, R,
                       ln_eps, normalizer, clamp_min, dtype, stream);
}

extern "C" int lina_gla_decode_inproj_packed(const void* x_packed, const void* w_in_packed, const float* c1,
                                             const float* c2, const void* wq, const void* wk, const void* wv, void* cq,
                                             void* ck, void* cv, const void* w2, const void* b2, void* qkv, void* g_out,
                                             float* gk, int B, int K, int Kd, int Vd, int W, int R, float ln_eps,
                                             float normalizer, float clamp_min, int w_stream, int dtype,
                                             lina_stream_t stream) {
    return inproj_impl(x_packed, 0, w_in_packed, 0, w_stream ? 3 : 1, c1, c2, wq, wk, wv, cq, ck, cv, w2, b2, qkv, g_out, gk, B, K, Kd,
                       Vd, W, R, ln_eps, normalizer, clamp_min, dtype, stream);
}

#ifdef LINA_SKINNY_PROF
extern "C" int lina_inproj_prof_read(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(lina_inproj_prof), sizeof(unsigned long long) * 1024 * 8);
}
#endif
