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
// Work split: grid = (ceil(N/16), ceil(M/64)); 256 threads = 4 waves; a workgroup owns a 64 x 16
// output tile; wave w takes k-steps {2w,2w+1}+8j (whole 128-B lines) for ALL 16-row m-tiles (in-workgroup split-K),
// so every byte of W and of A is loaded once per workgroup, straight from global memory into MFMA
// fragment layout (16 B per lane, no LDS staging: each operand byte is used once per wave).  The four
// partial accumulator sets are reduced through LDS; wave w finalises m-tile w.
// bf16: v_mfma_f32_16x16x32_bf16 (k-step 32);  f32: v_mfma_f32_16x16x4_f32 x4 (k-step 16).
#include <lina_dev.h>
#include "lina_common.h"
#include "skinny_frag.h"
#include "linear_tall.h"
#include <stdlib.h>

#ifdef LINA_SKINNY_PROF
// tools-only build (tools/skinny_prof.sh): time stamps of thread 0 of every workgroup, [kind][workgroup][slot]; kind 0 =
// SwiGLU up-projection, 1 = K = 1024 with residual (o-projection), 2 = other with residual (down), 3 = the rest.
// slots: 0 wall clock (100 MHz) at entry, 1 shader clock at entry, 2 after the first round of loads was consumed, 3 after
// the main loop, 4 after the reduction barrier, 5 at the end, 6 wall clock at the end.  NOT part of the product library.
__device__ unsigned long long lina_skinny_prof[4 * 1024 * 8];
#define SK_PROF(i, expr) do { if (threadIdx.x == 0) pr_[i] = (expr); } while (0)
#else
#define SK_PROF(i, expr) do { } while (0)
#endif

namespace lina {

template <typename T, bool SWIGLU, bool LN, int MT, int NT, bool PK, bool WNT, int NW = 4>
__global__ __launch_bounds__(64 * NW) void linear_skinny_kernel(
    const T* __restrict__ A, int64_t lda, const T* __restrict__ W, int64_t ldw, const float* __restrict__ c1,
    const float* __restrict__ c2, const T* resid, int64_t ldr, T* out, int64_t ldo, int M, int N, int K, int Hd,
    int ln_dim, float ln_eps, T* outp, int Kp, int Hp) {
    // WNT: weight fragments with the non-temporal load hint (packed operands only)
    // NW: waves per workgroup = width of the in-workgroup split-K.  The operands arrive at ~0.09 us per 1 KiB load instruction
    // PER WAVE (time stamps of tools/probe_skinny_prof.py: the first load round of a wave with 16 / 32 / 48 loads in flight
    // lands 2.3 / 4.4 / 5.6 us after entry), so the same bytes per workgroup spread over more waves arrive sooner.
    // PK: A and W are fragment-major (skinny_frag.h); Hp = padded rows per weight half.  outp (optional, any PK): a packed
    // copy of the output for the next projection, Kp = its padded width.
    using F = Frag<T>;
#ifdef LINA_SKINNY_PROF
    unsigned long long pr_[8] = {};
    SK_PROF(0, wall_clock64());
    SK_PROF(1, clock64());
#endif
    constexpr int NB = SWIGLU ? 2 : 1;      // weight-row halves (gate | value) per output column
    constexpr int G = NB * NT;              // 16-row groups of W per workgroup
    constexpr int U = NW > 8 ? 2 : (NW > 4 || (G + MT) * 8 > 48) ? 4 : 8;   // k-steps in flight: (G + MT) * U 16-byte loads per lane
    __shared__ __attribute__((aligned(16))) float s_acc[NW][G * MT][64][4];  // [wave][group x m-tile][lane][reg]
    __shared__ float s_st[NW][16 * MT][2];
    __shared__ float s_fin[NW > 4 ? 16 * MT : 1][2];   // NW > 4: the rows' LayerNorm sums, added up once (see the reduction)

    const int tid = threadIdx.x, lane = tid & 63;
    // the wave index in an SGPR: everything that depends on it (k-step ownership, remainder rounds, who finalises what) is
    // then a SCALAR branch.  With w in a VGPR the compiler predicates such code with EXEC -- and an MFMA issued under
    // EXEC = 0 still executes, on whatever its (unwritten) operand registers hold: non-finite sums on the hardware.
    const int w = wave_uniform(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int n0 = blockIdx.x * (16 * NT), m0 = blockIdx.y * (16 * MT);
    const int n_rows = SWIGLU ? Hd : N;     // weight rows per half

    f32x4 acc[G * MT];
#pragma unroll
    for (int i = 0; i < G * MT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // LayerNorm statistics without VALU work: sum_k a = (A . 1)[i][*], sum_k a^2 = diag(A . A^T) -- the A
    // fragment is also a valid B fragment (same lane layout with i <-> n), so both are two more MFMAs per step
    // (16-wave workgroups have 128 registers per lane: there the two statistics are per-lane dot products of the lane's own
    // fragment elements -- 2 registers per m-tile instead of 8 -- summed over the four lane groups of a row at the end)
    constexpr bool VST = LN && NW == 16;
    // Finalisation: MT * RS waves, each takes 4 / RS of a lane's four output rows of one m-tile.  With one wave per m-tile
    // the epilogue of the 16-wave kernels was a single wave adding 16 partial tiles and storing four rows with every
    // latency exposed (0.4 + 0.6 us of the o-projection's 3.0 us, 0.8 + 0.9 us of the up-projection's 5.4: time stamps).
    constexpr int RS = NW >= 4 * MT ? 4 : NW >= 2 * MT ? 2 : 1;
    constexpr int RPW = 4 / RS;
    f32x4 st1[MT], st2[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) { st1[i] = f32x4{0.f, 0.f, 0.f, 0.f}; st2[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    F f_ones;
    f_ones.ones();

    const int nks_all = K / F::KSTEP;
    const int64_t kstr = PK ? 64 * F::KL : F::KSTEP;   // elements between consecutive k-steps of one fragment row block
    const T* wp[G];
    bool g_ok[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int n = n0 + 16 * (g % NT) + li;
        if (PK) {                                       // padded with zero rows: every tile of the grid is readable
            g_ok[g] = true;
            wp[g] = W + ((int64_t)(((g / NT) * Hp + n0 + 16 * (g % NT)) >> 4) * nks_all * 64 + lane) * F::KL;
        } else {
            g_ok[g] = n < n_rows;
            wp[g] = W + (int64_t)(g_ok[g] ? (g / NT) * Hd + n : 0) * ldw + F::KL * lg;
        }
    }
    const T* ap[MT];
    bool m_ok[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = m0 + 16 * mt + li;
        if (PK) {                                       // rows padded to 64: rows >= M feed only output rows >= M
            m_ok[mt] = true;
            ap[mt] = A + ((int64_t)((m0 >> 4) + mt) * nks_all * 64 + lane) * F::KL;
        } else {
            m_ok[mt] = m < M;
            ap[mt] = A + (int64_t)(m_ok[mt] ? m : 0) * lda + F::KL * lg;
        }
    }

    // epilogue operands that do not depend on the GEMM (residual, fold constants) are requested before the main loop:
    // their latency hides under it instead of sitting exposed after the reduction
    T pre_res[NT][4];                     // RAW residual values (converted where they are used: lina_common.h raw4)
    float pre_c1[G], pre_c2[G];
    // Every load below is UNCONDITIONAL on a clamped (always readable) address and the value is selected afterwards; what
    // decides whether a tensor exists at all is a kernel argument (a scalar branch).  The predicated form
    // `ok ? ld(p) : 0` compiles to an EXEC-masked region per load with `s_waitcnt vmcnt(0)` behind it -- the loads ran one
    // memory round trip after the other (4 + 2 of them in front of the main loop's first fragment load).
    // They are requested BEHIND the first round of fragment loads: a wave's loads return in order, and the fragments are
    // needed first (A/B in the L169 step: 0.609 -> 0.583 ms per token, tests/gpu_r03m.sh).
    auto preload = [&]() {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int n = n0 + 16 * (g % NT) + li;
            const bool ok = n < n_rows;
            const int idx = (g / NT) * Hd + (ok ? n : n_rows - 1);
            // (no select on the loaded value here: it would be a USE, i.e. a wait for this load -- and for every fragment load
            //  in front of it; columns past the end get a neighbour's constant and are never stored)
            pre_c1[g] = 0.0f;
            pre_c2[g] = 0.0f;
            if (LN) pre_c1[g] = c1[idx];
            if (c2) pre_c2[g] = c2[idx];
        }
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) pre_res[j][r] = T();
        if (resid && w < MT * RS) {                          // (only the finalising waves; w is wave-uniform)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    const int m = m0 + 16 * (w / RS) + 4 * lg + (w % RS) * RPW + r, n = n0 + 16 * j + li;
                    const int mc = m < M ? m : M - 1, nc = n < N ? n : N - 1;
                    // resid == outp: the residual stream lives ONLY in the packed buffer (read-modify-write by the same thread)
                    pre_res[j][r] = ld_raw(resid == outp ? resid + packed_off<T>(mc, nc, Kp) : resid + (int64_t)mc * ldr + nc);
                }
        }
    };
    bool pre_done = false;

    const int nsteps = K / F::KSTEP;
    // wave w takes k-steps {2w, 2w+1} + 8j: its two consecutive 64-byte (bf16) loads of a row are the two halves
    // of ONE 128-byte line, so every line is pulled into this CU's L1 by a single wave, back to back
    int ks = 0;                                  // per-wave step counter; global k-step = kstep_of<NW>(w, ks)
    for (; kstep_of<NW>(w, ks + U - 1) < nsteps; ks += U) {
        F fb[U][G], fa[U][MT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t k0 = kstep_of<NW>(w, ks + u) * kstr;
#pragma unroll
            for (int g = 0; g < G; ++g) { if (g_ok[g]) fb[u][g].template load_stream<WNT>(wp[g] + k0); else fb[u][g].zero(); }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) { if (m_ok[mt]) fa[u][mt].load(ap[mt] + k0); else fa[u][mt].zero(); }
        }
        if (ks == 0) { sched_fence(); preload(); sched_fence(); pre_done = true; }   // behind the first round's fragments
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (LN && VST) {
                    float s1_ = st1[mt][0], s2_ = st2[mt][0];
                    fa[u][mt].stats(s1_, s2_);
                    st1[mt][0] = s1_; st2[mt][0] = s2_;
                }
                if (LN && !VST) {
                    st1[mt] = F::mma(fa[u][mt], f_ones, st1[mt]);
                    st2[mt] = F::mma(fa[u][mt], fa[u][mt], st2[mt]);
                }
#pragma unroll
                for (int g = 0; g < G; ++g) acc[g * MT + mt] = F::mma(fa[u][mt], fb[u][g], acc[g * MT + mt]);
            }
#ifdef LINA_SKINNY_PROF
        if (ks == 0) SK_PROF(2, clock64());
#endif
    }
    SK_PROF(3, clock64());
    if (!pre_done) preload();                                     // (a K shorter than one round)
    if (kstep_of<NW>(w, ks) < nsteps) {
        // remainder (fewer than U steps for this wave, e.g. K = 1376: 43 k-steps): ONE more round with every load in flight --
        // a loop of single steps pays a full memory round trip per step (+1.1 us on the down-projection, time stamps of
        // tools/probe_skinny_prof.py).  Steps past the end contribute zero fragments: the sums are unchanged bit for bit.
        F fb[U][G], fa[U][MT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool on = kstep_of<NW>(w, ks + u) < nsteps;           // wave-uniform
            const int64_t k0 = kstep_of<NW>(w, ks + u) * kstr;
#pragma unroll
            for (int g = 0; g < G; ++g) { if (on && g_ok[g]) fb[u][g].template load_stream<WNT>(wp[g] + k0); else fb[u][g].zero(); }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) { if (on && m_ok[mt]) fa[u][mt].load(ap[mt] + k0); else fa[u][mt].zero(); }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (kstep_of<NW>(w, ks + u) >= nsteps) break;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (LN && VST) {
                    float s1_ = st1[mt][0], s2_ = st2[mt][0];
                    fa[u][mt].stats(s1_, s2_);
                    st1[mt][0] = s1_; st2[mt][0] = s2_;
                }
                if (LN && !VST) {
                    st1[mt] = F::mma(fa[u][mt], f_ones, st1[mt]);
                    st2[mt] = F::mma(fa[u][mt], fa[u][mt], st2[mt]);
                }
#pragma unroll
                for (int g = 0; g < G; ++g) acc[g * MT + mt] = F::mma(fa[u][mt], fb[u][g], acc[g * MT + mt]);
            }
        }
    }

    // ---- in-workgroup split-K reduction ----
#pragma unroll
    for (int i = 0; i < G * MT; ++i)
        *reinterpret_cast<float4*>(&s_acc[w][i][lane][0]) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
    if (LN) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {   // D layout: lane (col li, rows 4lg+r); the diagonal sits at li == 4lg+r
            if (VST) {                           // lane (li, lg) holds row li's sums over its own 8 (4) k-slots of every step
                float a = st1[mt][0], b = st2[mt][0];
                a += shfl_xor(a, 16); b += shfl_xor(b, 16);
                a += shfl_xor(a, 32); b += shfl_xor(b, 32);
                if (lg == 0) { s_st[w][16 * mt + li][0] = a; s_st[w][16 * mt + li][1] = b; }
                continue;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (li == 4 * lg + r) { s_st[w][16 * mt + li][0] = st1[mt][r]; s_st[w][16 * mt + li][1] = st2[mt][r]; }
        }
    }
    __syncthreads();
    if (LN && NW > 4) {
        // with 8 / 16 partial sets a finalising lane would add 2 x 4 x NW dependent LDS values for its four rows (~1000 clocks
        // of the epilogue): one thread per (row, statistic) adds them once instead, in the same pairwise order
        if (tid < 32 * MT) {
            const int row = tid >> 1, c = tid & 1;
            float a = (s_st[0][row][c] + s_st[1][row][c]) + (s_st[2][row][c] + s_st[3][row][c]);
#pragma unroll
            for (int ww = 4; ww < NW; ww += 4)
                a += (s_st[ww][row][c] + s_st[ww + 1][row][c]) + (s_st[ww + 2][row][c] + s_st[ww + 3][row][c]);
            s_fin[row][c] = a;
        }
        __syncthreads();
    }
    SK_PROF(4, clock64());
    if (w >= MT * RS) return;

    // wave w finalises rows [rq RPW, rq RPW + RPW) of every lane's four rows of m-tile mtf, for every group:
    // D layout -> rows m0 + 16 mtf + 4 lg + r, column li of the tile.  Same sums in the same order as with RS = 1.
    const int mtf = w / RS, rq = w % RS;
    float val[G][RPW];
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int r = 0; r < RPW; ++r) val[g][r] = s_acc[0][g * MT + mtf][lane][rq * RPW + r];
#pragma unroll 4                          // (all NW reads in flight at once would cost RPW NW registers per group)
        for (int ww = 1; ww < NW; ++ww) {
#pragma unroll
            for (int r = 0; r < RPW; ++r) val[g][r] += s_acc[ww][g * MT + mtf][lane][rq * RPW + r];
        }
    }
#ifdef LINA_SKINNY_PROF
    if (val[0][0] == 1.2345e30f) s_fin[0][0] = 1.0f;             // (consume the sums before the stamp)
    SK_PROF(7, clock64());
#endif
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = 16 * mtf + 4 * lg + rq * RPW + r;
        const int m = m0 + row;
        float mu = 0.f, rstd = 1.f;
        if (LN) {
            float a, b;
            if (NW > 4) { a = s_fin[row][0]; b = s_fin[row][1]; }
            else {
                a = (s_st[0][row][0] + s_st[1][row][0]) + (s_st[2][row][0] + s_st[3][row][0]);
                b = (s_st[0][row][1] + s_st[1][row][1]) + (s_st[2][row][1] + s_st[3][row][1]);
            }
            const float inv_d = fast_rcp((float)ln_dim);
            mu = a * inv_d;
            rstd = rsqrtf(fmaxf(b * inv_d - mu * mu, 0.f) + ln_eps);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = n0 + 16 * j + li;
            const bool n_ok = n < n_rows;
            float res;
            if (SWIGLU) {
                float ga = val[j][r], gb = val[NT + j][r];
                if (n_ok) {
                    if (LN) { ga = rstd * (ga - mu * pre_c1[j]); gb = rstd * (gb - mu * pre_c1[NT + j]); }
                    if (c2) { ga += pre_c2[j]; gb += pre_c2[NT + j]; }
                }
                res = n_ok ? silu(ga) * gb : ((n == Hd) ? 1.0f : 0.0f);   // bias column of the K-padded row
            } else {
                res = val[j][r];
                if (n_ok) {
                    if (LN) res = rstd * (res - mu * pre_c1[j]);
                    if (c2) res += pre_c2[j];
                }
            }
            if (m < M && n < N) {
                if (resid) res += cvt1(pre_res[j][r]);
                if (out) st(out + (int64_t)m * ldo + n, res);
                if (outp) st(outp + packed_off<T>(m, n, Kp), res);
            }
        }
    }
#ifdef LINA_SKINNY_PROF
    SK_PROF(5, clock64());
    SK_PROF(6, wall_clock64());
    if (threadIdx.x == 0) {
        const int kind = SWIGLU ? 0 : (resid ? (K == 1024 ? 1 : 2) : 3);
        const int wg = blockIdx.y * gridDim.x + blockIdx.x;
        if (wg < 1024)
            for (int i = 0; i < 8; ++i) lina_skinny_prof[(kind * 1024 + wg) * 8 + i] = pr_[i];
    }
#endif
}

// ---- M >= kTallMinRows (160): the tall tiling (linear_tall.h) with the same epilogues.  Packed operands only; XCD-aware 1-D grid over
// ceil(N / 16 NT) column blocks x ceil(M / 64) row blocks (tall_tile_of).
template <typename T, bool SWIGLU, bool LN, int NT, int V>
__global__ __launch_bounds__(64 * TallShape<V>::NWV) void linear_tall_kernel(
    const T* __restrict__ A, const T* __restrict__ W, const float* __restrict__ c1, const float* __restrict__ c2,
    const T* resid, int64_t ldr, T* out, int64_t ldo, int M, int N, int K, int Hd, int ln_dim, float ln_eps, T* outp, int Kp,
    int Hp) {
    using F = Frag<T>;
    constexpr int NB = SWIGLU ? 2 : 1, G = NB * NT, MTW = TallShape<V>::MTW;
    __shared__ __attribute__((aligned(16))) unsigned char s_w[TallShape<V>::LDS];
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    const int w = wave_uniform(threadIdx.x >> 6);
    int cblk, rblk;
    if (!tall_tile_of((int)blockIdx.x, (N + 16 * NT - 1) / (16 * NT), (M + TallShape<V>::ROWS - 1) / TallShape<V>::ROWS, cblk, rblk))
        return;
    const int n0 = cblk * (16 * NT), m0 = rblk * TallShape<V>::ROWS + 16 * MTW * w;
    const int n_rows = SWIGLU ? Hd : N;                    // weight rows per half
    int nb[G];
#pragma unroll
    for (int g = 0; g < G; ++g) nb[g] = ((g / NT) * Hp + n0 + 16 * (g % NT)) >> 4;
    // fold constants of this lane's columns (clamped address; columns past the end are never stored)
    float pre_c1[G], pre_c2[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int n = n0 + 16 * (g % NT) + li;
        const int idx = (g / NT) * Hd + (n < n_rows ? n : n_rows - 1);
        pre_c1[g] = LN ? c1[idx] : 0.0f;
        pre_c2[g] = c2 ? c2[idx] : 0.0f;
    }
    f32x4 acc[G][MTW];
    float s1[MTW][4], s2[MTW][4];
    tall_core_v<V, T, G, LN, MTW>(A, W, nb, K / F::KSTEP, m0 >> 4, m0 < (M + 63) / 64 * 64, s_w, acc, s1, s2);

#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        float mu[4] = {0.f, 0.f, 0.f, 0.f}, rstd[4] = {1.f, 1.f, 1.f, 1.f};
        if (LN) tall_row_stats(s1[mt], s2[mt], fast_rcp((float)ln_dim), ln_eps, mu, rstd);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 16 * mt + 4 * lg + r;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = n0 + 16 * j + li;
                const bool n_ok = n < n_rows;
                float res;
                if (SWIGLU) {
                    float ga = acc[j][mt][r], gb = acc[NT + j][mt][r];
                    if (n_ok) {
                        if (LN) { ga = rstd[r] * (ga - mu[r] * pre_c1[j]); gb = rstd[r] * (gb - mu[r] * pre_c1[NT + j]); }
                        if (c2) { ga += pre_c2[j]; gb += pre_c2[NT + j]; }
                    }
                    res = n_ok ? silu(ga) * gb : ((n == Hd) ? 1.0f : 0.0f);   // bias column of the K-padded row
                } else {
                    res = acc[j][mt][r];
                    if (n_ok) {
                        if (LN) res = rstd[r] * (res - mu[r] * pre_c1[j]);
                        if (c2) res += pre_c2[j];
                    }
                }
                if (m < M && n < N) {
                    if (resid) res += ld(resid == outp ? resid + packed_off<T>(m, n, Kp) : resid + (int64_t)m * ldr + n);
                    if (out) st(out + (int64_t)m * ldo + n, res);
                    if (outp) st(outp + packed_off<T>(m, n, Kp), res);
                }
            }
        }
    }
}

}  // namespace lina

// packed kernels: one launch helper per tiling that picks the split-K width.  16 waves only where a wave's accumulators and
// in-flight fragments fit 128 registers (the plain N x K projections on 16-column tiles); the other shapes stop at 8.
template <typename TT, bool SW, bool LNN, int MTT, int NTT, bool WNTT>
static void launch_skinny_packed(int nw, dim3 grid, lina_stream_t stream, const void* A, int64_t lda, const void* W, int64_t ldw,
                                 const float* c1, const float* c2, const void* resid, int64_t ldr, void* out, int64_t ldo,
                                 int M, int N, int K, int swiglu_hidden, int ln_dim, float ln_eps, void* outp, int Kp, int Hp) {
    using namespace lina;
#define LINA_LS_GO(NWW)                                                                                              \
    LINA_LAUNCH((linear_skinny_kernel<TT, SW, LNN, MTT, NTT, true, WNTT, NWW>), grid, dim3(64 * NWW), 0, stream,      \
                (const TT*)A, lda, (const TT*)W, ldw, c1, c2, (const TT*)resid, ldr, (TT*)out, ldo, M, N, K,          \
                swiglu_hidden, ln_dim, ln_eps, (TT*)outp, Kp, Hp)
    if constexpr (NTT == 1 && MTT <= 2) {
        if (nw == 16) { LINA_LS_GO(16); return; }
    }
    if constexpr (SW && MTT == 4 && NTT == 2) {       // 16 accumulator tiles per wave: 8 waves would spill
        LINA_LS_GO(4);
    } else {
        if (nw >= 8) LINA_LS_GO(8);
        else LINA_LS_GO(4);
    }
#undef LINA_LS_GO
}

static int linear_skinny_impl(const void* A, int64_t lda, const void* W, int64_t ldw, const float* c1, const float* c2,
                              const void* resid, int64_t ldr, void* out, int64_t ldo, int M, int N, int K,
                              int swiglu_hidden, int ln_dim, float ln_eps, int packed, void* outp, int Kp, int Hp,
                              int dtype, lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(A && W && (out || outp), "lina_linear_skinny: null pointer");
    LINA_REQUIRE(M > 0 && N > 0 && K > 0, "lina_linear_skinny: M,N,K must be positive");
    LINA_REQUIRE(valid_dtype(dtype), "lina_linear_skinny: bad dtype %d", dtype);
    const int kstep = dtype == LINA_BF16 ? 32 : 16, al = dtype == LINA_BF16 ? 8 : 4;
    LINA_REQUIRE(K % kstep == 0, "lina_linear_skinny: K=%d must be a multiple of %d (pad the operands)", K, kstep);
    LINA_REQUIRE((packed & 1) || (lda % al == 0 && ldw % al == 0), "lina_linear_skinny: lda/ldw must keep rows 16-byte aligned");
    LINA_REQUIRE(!outp || (Kp > 0 && Kp % kstep == 0 && Kp >= N), "lina_linear_skinny: packed output width must be a multiple of %d and >= N", kstep);
    LINA_REQUIRE(!(packed & 1) || Hp % 64 == 0, "lina_linear_skinny: packed weights need rows padded to a multiple of 64");
    LINA_REQUIRE(ln_dim >= 0 && (ln_dim == 0 || c1), "lina_linear_skinny: LayerNorm folding needs c1");
    LINA_REQUIRE(swiglu_hidden >= 0 && swiglu_hidden <= N, "lina_linear_skinny: bad swiglu_hidden");
    // Tiling.  Measured on MI355X (tools/perf_skinny2.py): one CU ingests only ~30 GB/s on this access pattern,
    // so the kernel time is (bytes pulled by the busiest CU) / 30 GB/s + ~2.4 us.  A workgroup pulls
    // (16*MT rows of A + 16*NT*NB rows of W) * K elements; pick (MT, NT) minimising rounds * bytes-per-workgroup
    // with rounds = ceil(workgroups / 256 CUs).
    const bool sw = swiglu_hidden > 0, ln = ln_dim > 0;
    const int nb = sw ? 2 : 1;
    {   // M >= 160 on packed operands, wide outputs: the tall tiling (linear_tall.h).  Measured per launch at L169
        // (tools/perf_tall.py, profiles/r05_tall_perf.txt; 64-row split-K kernel -> tall), up-projection / head:
        // M = 512 21.9 -> 12.0 / 22.4 -> 12.6 us, 384: 20.4 -> 11.5 / 18.0 -> 11.8, 256: 11.3 -> 9.8 / 13.1 -> 10.4,
        // 192: 10.9 -> 9.7 / 9.6 -> 9.8, 128: 8.8 -> 9.6 / 9.1 -> 9.5 (hence the row threshold).
        // LINA_TALL=0 / 1: never / whenever the operands allow it (test hook, like LINA_SKINNY_WAVES: read per call)
        const char* tall_env = getenv("LINA_TALL");
        const int tall_mode = tall_env ? atoi(tall_env) : -1;
        const bool can = (packed & 1) && Hp % 64 == 0;
        const bool want = tall_mode == 1 || (tall_mode != 0 && M >= kTallMinRows && (sw || N >= 2048));
        if (can && want) {
            const int nt = sw ? 2 : 4;                       // 32 gate + 32 value weight rows / 64 plain weight rows per workgroup
            LINA_REQUIRE(Hp >= (N + 16 * nt - 1) / (16 * nt) * (16 * nt),
                         "lina_linear_skinny: packed weights must be zero-padded to whole %d-row blocks covering N", 16 * nt);
            const char* v_env = getenv("LINA_TALL_V");       // variant (linear_tall.h): 0 = LDS ring, 1 = register ring, 2 = W in LDS + A in registers
            int tv = v_env ? atoi(v_env) : 0;         // measured (profiles/r05_tall_perf.txt): head 12.5 / 14.8 / 13.1 us as variant 0 / 1 / 2
            if (tv < 0 || tv > 2) tv = 0;             // (3 / 4 name variants of the fused in-projection only: gla_inproj.hip)
            const int rows = tv == 1 ? TallShape<1>::ROWS : TallShape<0>::ROWS;
            dim3 tgrid(tall_grid((N + 16 * nt - 1) / (16 * nt), (M + rows - 1) / rows));
#define LINA_LT(TT, SW, LNN, NTT, VV)                                                                                \
    LINA_LAUNCH((linear_tall_kernel<TT, SW, LNN, NTT, VV>), tgrid, dim3(64 * TallShape<VV>::NWV), 0, stream,          \
                (const TT*)A, (const TT*)W, c1, c2, (const TT*)resid, ldr, (TT*)out, ldo, M, N, K, swiglu_hidden,       \
                ln_dim, ln_eps, (TT*)outp, Kp, Hp)
#define LINA_LT_V(TT, VV)                                                                                            \
    do {                                                                                                             \
        if (sw && ln) LINA_LT(TT, true, true, 2, VV); else if (sw) LINA_LT(TT, true, false, 2, VV);                  \
        else if (ln) LINA_LT(TT, false, true, 4, VV); else LINA_LT(TT, false, false, 4, VV);                         \
    } while (0)
#define LINA_LT_T(TT) do { if (tv == 2) LINA_LT_V(TT, 2); else if (tv) LINA_LT_V(TT, 1); else LINA_LT_V(TT, 0); } while (0)
            if (dtype == LINA_BF16) LINA_LT_T(bf16_t); else LINA_LT_T(float);
#undef LINA_LT_T
#undef LINA_LT_V
#undef LINA_LT
            return check_launch("lina_linear_skinny (tall)");
        }
    }
    int best_mt = 4, best_nt = 1;
    long best_cost = -1;
    const int cand[5][2] = {{1, 1}, {2, 1}, {4, 1}, {2, 2}, {4, 2}};
    for (int c = 0; c < 5; ++c) {
        const int mt = cand[c][0], nt = cand[c][1];
        const long wgs = (long)((N + 16 * nt - 1) / (16 * nt)) * ((M + 16 * mt - 1) / (16 * mt));
        const long cost = ((wgs + 255) / 256) * (16L * mt + 16L * nt * nb);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_mt = mt; best_nt = nt; }
    }
    {   // tuning knob (tools/perf_skinny3.py): LINA_SKINNY_TILE="mt,nt" forces one of the built tilings
        static const char* forced = getenv("LINA_SKINNY_TILE");
        int fm = 0, fn = 0;
        if (forced && sscanf(forced, "%d,%d", &fm, &fn) == 2)
            for (int c = 0; c < 5; ++c)
                if (cand[c][0] == fm && cand[c][1] == fn) { best_mt = fm; best_nt = fn; }
    }
    dim3 grid((unsigned)((N + 16 * best_nt - 1) / (16 * best_nt)), (unsigned)((M + 16 * best_mt - 1) / (16 * best_mt)));
    // waves per workgroup (split-K width) of the packed kernels: more waves = fewer loads per wave for the same bytes per
    // workgroup (see the kernel's NW note); each wave needs at least one pair of k-steps.  LINA_SKINNY_WAVES overrides.
    // Measured in the L169 decode step (tests/gpu_r03e.sh, ms per token): 4 waves 0.660, 8 waves 0.624, 16 waves on the plain
    // 16-column projections + 8 elsewhere 0.617.
    int nw = 16;
    {
        const char* forced_nw = getenv("LINA_SKINNY_WAVES");   // (read per call: tests switch it)
        if (forced_nw) nw = atoi(forced_nw);
        while (nw > 4 && K / kstep < 2 * nw) nw /= 2;
        if (nw != 8 && nw != 16) nw = 4;
    }
#define LINA_LS_PK(TT, SW, LNN, MTT, NTT, PKK, WNTT, NWW)                                                          \
    LINA_LAUNCH((linear_skinny_kernel<TT, SW, LNN, MTT, NTT, PKK, WNTT, NWW>), grid, dim3(64 * NWW), 0, stream,      \
                (const TT*)A, lda, (const TT*)W, ldw, c1, c2, (const TT*)resid, ldr, (TT*)out, ldo, M, N, K,         \
                swiglu_hidden, ln_dim, ln_eps, (TT*)outp, Kp, Hp)
#define LINA_LS_NW(TT, SW, LNN, MTT, NTT, WNTT)                                                                     \
    launch_skinny_packed<TT, SW, LNN, MTT, NTT, WNTT>(nw, grid, stream, A, lda, W, ldw, c1, c2, resid, ldr, out, ldo, M, N, \
                                                      K, swiglu_hidden, ln_dim, ln_eps, outp, Kp, Hp)
#define LINA_LS_ONE(TT, SW, LNN, MTT, NTT)                                                                          \
    do {                                                                                                            \
        if (packed == 3) LINA_LS_NW(TT, SW, LNN, MTT, NTT, true);                                                   \
        else if (packed & 1) LINA_LS_NW(TT, SW, LNN, MTT, NTT, false);                                              \
        else LINA_LS_PK(TT, SW, LNN, MTT, NTT, false, false, 4);                                                    \
    } while (0)
#define LINA_LS(TT, SW, LNN)                                                                                        \
    do {                                                                                                            \
        if (best_nt == 1 && best_mt == 1) LINA_LS_ONE(TT, SW, LNN, 1, 1);                                           \
        else if (best_nt == 1 && best_mt == 2) LINA_LS_ONE(TT, SW, LNN, 2, 1);                                      \
        else if (best_nt == 1) LINA_LS_ONE(TT, SW, LNN, 4, 1);                                                      \
        else if (best_mt == 2) LINA_LS_ONE(TT, SW, LNN, 2, 2);                                                      \
        else LINA_LS_ONE(TT, SW, LNN, 4, 2);                                                                        \
    } while (0)
    if (dtype == LINA_BF16) {
        if (sw && ln) LINA_LS(bf16_t, true, true); else if (sw) LINA_LS(bf16_t, true, false);
        else if (ln) LINA_LS(bf16_t, false, true); else LINA_LS(bf16_t, false, false);
    } else {
        if (sw && ln) LINA_LS(float, true, true); else if (sw) LINA_LS(float, true, false);
        else if (ln) LINA_LS(float, false, true); else LINA_LS(float, false, false);
    }
#undef LINA_LS
#undef LINA_LS_ONE
#undef LINA_LS_NW
#undef LINA_LS_PK
    return check_launch("lina_linear_skinny");
}

extern "C" int lina_linear_skinny(const void* A, int64_t lda, const void* W, int64_t ldw, const float* c1,
                                  const float* c2, const void* resid, int64_t ldr, void* out, int64_t ldo, int M,
                                  int N, int K, int swiglu_hidden, int ln_dim, float ln_eps, int dtype,
                                  lina_stream_t stream) {
    return linear_skinny_impl(A, lda, W, ldw, c1, c2, resid, ldr, out, ldo, M, N, K, swiglu_hidden, ln_dim, ln_eps, 0,
                              nullptr, 0, 0, dtype, stream);
}

extern "C" int lina_linear_skinny_ex(const void* A, int64_t lda, const void* W, int64_t ldw, int in_packed,
                                     int w_half_rows, const float* c1, const float* c2, const void* resid, int64_t ldr,
                                     void* out, int64_t ldo, void* out_packed, int out_packed_width, int M, int N, int K,
                                     int swiglu_hidden, int ln_dim, float ln_eps, int dtype, lina_stream_t stream) {
    return linear_skinny_impl(A, lda, W, ldw, c1, c2, resid, ldr, out, ldo, M, N, K, swiglu_hidden, ln_dim, ln_eps,
                              in_packed & 3, out_packed, out_packed_width, w_half_rows, dtype, stream);
}

#ifdef LINA_SKINNY_PROF
extern "C" int lina_skinny_prof_read(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(lina_skinny_prof), sizeof(unsigned long long) * 4 * 1024 * 8);
}
#endif
