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
#include "linear_tall.h"
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

namespace lina {

template <typename T, int NT, bool PK, bool WNT, int NW = 4>
__global__ __launch_bounds__(64 * NW) void gla_inproj_kernel(
    const T* __restrict__ A, int64_t lda, const T* __restrict__ W, int64_t ldw, const float* __restrict__ c1,
    const float* __restrict__ c2, const T* __restrict__ wq, const T* __restrict__ wk, const T* __restrict__ wv,
    T* cq, T* ck, T* cv, const T* __restrict__ w2, const T* __restrict__ b2, T* __restrict__ qkv,
    T* __restrict__ g_out, float* __restrict__ gk, int M, int K, int Kd, int Vd, float ln_eps, float inv_norm,
    float clamp_min, int has_clamp) {
    using F = Frag<T>;              // WNT: weight fragments with the non-temporal load hint
#ifdef LINA_SKINNY_PROF
    unsigned long long pr_[8] = {};
    IP_PROF(0, wall_clock64());
    IP_PROF(1, clock64());
#endif
    constexpr int R = 16, MT = 4;
    constexpr int U = NW > 8 ? 2 : (NW > 4 || (NT + MT) * 8 > 48) ? 4 : 8;   // NW: split-K width, see linear_skinny.hip
    constexpr int RS = NW >= 4 * MT ? 4 : NW >= 2 * MT ? 2 : 1;              // finalising waves per m-tile (row split), see there
    constexpr int RPW = 4 / RS;
    __shared__ __attribute__((aligned(16))) float s_acc[NW][NT * MT][64][4];
    __shared__ float s_st[NW][64][2];
    __shared__ float s_fin[NW > 4 ? 64 : 1][2];           // NW > 4: the rows' LayerNorm sums, added up once (linear_skinny.hip)
    __shared__ float s_lr[64][R + 1];

    const int tid = threadIdx.x, lane = tid & 63;
    // the wave index in an SGPR: everything that depends on it (k-step ownership, remainder rounds, who finalises what) is
    // then a SCALAR branch.  With w in a VGPR the compiler predicates such code with EXEC -- and an MFMA issued under
    // EXEC = 0 still executes, on whatever its (unwritten) operand registers hold: non-finite sums on the hardware.
    const int w = wave_uniform(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int m0 = blockIdx.y * 64;
    const int n_direct = 2 * Kd + 2 * Vd;               // q | k | v | g columns; low-rank rows follow in W
    // q | k | v | g workgroups take 16 NT columns; a GATE workgroup takes 16 gate channels (one low-rank tile to multiply, but
    // the heaviest epilogue -- 16 FMAs + a log-sigmoid per element: with 32 channels the gate workgroups finished ~1.5 us
    // after everybody else, time stamps of tools/probe_skinny_prof.py; there are idle CUs for the extra workgroups)
    const int nb_direct = n_direct / (16 * NT);
    const bool gate_wg = (int)blockIdx.x >= nb_direct;  // block-uniform
    const int tile0 = gate_wg ? n_direct + ((int)blockIdx.x - nb_direct) * 16 : (int)blockIdx.x * (16 * NT);

    f32x4 acc[NT * MT], st1[MT], st2[MT];
#pragma unroll
    for (int i = 0; i < NT * MT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MT; ++i) { st1[i] = f32x4{0.f, 0.f, 0.f, 0.f}; st2[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    F f_ones;
    f_ones.ones();
    // PK: A and W fragment-major (skinny_frag.h; rows padded to 64): one contiguous 1 KiB per fragment load
    const int nks_all = K / F::KSTEP;
    const int64_t kstr = PK ? 64 * F::KL : F::KSTEP;
    const T* wp[NT];
    bool g_on[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        g_on[j] = !gate_wg || j == 0;
        const int wrow0 = gate_wg ? n_direct : tile0 + 16 * j;
        wp[j] = PK ? W + ((int64_t)(wrow0 >> 4) * nks_all * 64 + lane) * F::KL
                   : W + (int64_t)(wrow0 + li) * ldw + F::KL * lg;
    }
    const T* ap[MT];
    bool m_ok[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = m0 + 16 * mt + li;
        m_ok[mt] = PK || m < M;
        ap[mt] = PK ? A + ((int64_t)((m0 >> 4) + mt) * nks_all * 64 + lane) * F::KL
                    : A + (int64_t)(m < M ? m : 0) * lda + F::KL * lg;
    }
    // Epilogue operands that do not depend on the GEMM -- the rolled conv caches of this wave's 4 rows (wave w
    // finalises m-tile w), the conv taps and the LayerNorm-fold constants -- are requested NOW, so their
    // (HBM-cold) latency is hidden under the main loop instead of sitting exposed after the reduction.
    typedef typename raw4<T>::type raw_t;   // RAW bits of the epilogue operands, converted where they are used (lina_common.h)
    raw_t pre_old[NT][4], pre_wj[NT];
    T pre_b2[NT];
    float pre_c1[NT], pre_c2[NT];
    // Which tensor a TILE needs is decided per tile (Kd, Vd are multiples of 16: a 16-column tile never straddles q|k|v|g):
    // scalar branches, and inside them every load is unconditional on a clamped address -- a per-lane `ok ? ld(p) : 0`
    // compiles to one EXEC-masked region + `s_waitcnt vmcnt(0)` per load, i.e. one memory round trip after the other.
    // Requested BEHIND the first round of fragment loads (loads return in order; the fragments are needed first).
    auto preload = [&]() {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = gate_wg ? n_direct + li : tile0 + 16 * j + li;
            pre_c1[j] = c1[n];
            pre_c2[j] = c2[n];
            pre_wj[j] = raw_t();
            pre_b2[j] = T();
#pragma unroll
            for (int r = 0; r < 4; ++r) pre_old[j][r] = raw_t();
            if (w >= MT * RS) continue;                         // the other waves only feed the split-K sum
            const int tn0 = tile0 + 16 * j;                     // first column of this tile: workgroup-uniform
            if (gate_wg) {
                // gate tiles: the rank-16 up-projection row of this lane's channel (16 contiguous elements) and its bias are
                // epilogue operands too -- they travel in the registers the q/k/v tiles use for the conv cache
                if (j == 0 && tn0 - n_direct < Kd) {            // (Kd % 16 == 0: the whole tile is inside)
                    const int c = (tn0 - n_direct) + li;
#pragma unroll
                    for (int r = 0; r < 4; ++r) pre_old[j][r] = ld4_raw(w2 + (int64_t)c * R + 4 * r);
                    pre_b2[j] = ld_raw(b2 + c);
                }
            } else if (tn0 < 2 * Kd + Vd) {
                const T* wsel; const T* csel; int c, D;
                if (tn0 < Kd) { c = tn0 + li; D = Kd; wsel = wq; csel = cq; }
                else if (tn0 < 2 * Kd) { c = tn0 - Kd + li; D = Kd; wsel = wk; csel = ck; }
                else { c = tn0 - 2 * Kd + li; D = Vd; wsel = wv; csel = cv; }
                pre_wj[j] = ld4_raw(wsel + (int64_t)c * 4);
#pragma unroll
                for (int r = 0; r < RPW; ++r) {                  // this wave's rows of m-tile w / RS
                    const int m = m0 + 16 * (w / RS) + 4 * lg + (w % RS) * RPW + r;
                    pre_old[j][r] = ld4_raw(csel + ((int64_t)(m < M ? m : 0) * D + c) * 4);
                }
            }
        }
    };
    bool pre_done = false;

    const int nsteps = K / F::KSTEP;
    // wave w takes k-steps {2w, 2w+1} + 8j: its two consecutive 64-byte (bf16) loads of a row are the two halves
    // of ONE 128-byte line, so every line is pulled into this CU's L1 by a single wave, back to back
    int ks = 0;                                  // per-wave step counter; global k-step = kstep_of<NW>(w, ks)
    for (; kstep_of<NW>(w, ks + U - 1) < nsteps; ks += U) {
        F fb[U][NT], fa[U][MT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t k0 = kstep_of<NW>(w, ks + u) * kstr;
#pragma unroll
            for (int j = 0; j < NT; ++j) { if (g_on[j]) fb[u][j].template load_stream<WNT>(wp[j] + k0); else fb[u][j].zero(); }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) { if (m_ok[mt]) fa[u][mt].load(ap[mt] + k0); else fa[u][mt].zero(); }
        }
        if (ks == 0) { sched_fence(); preload(); sched_fence(); pre_done = true; }   // behind the first round's fragments
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                st1[mt] = F::mma(fa[u][mt], f_ones, st1[mt]);          // row sums      (LayerNorm mean)
                st2[mt] = F::mma(fa[u][mt], fa[u][mt], st2[mt]);       // Gram diagonal (LayerNorm variance)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[j * MT + mt] = F::mma(fa[u][mt], fb[u][j], acc[j * MT + mt]);
            }
#ifdef LINA_SKINNY_PROF
        if (ks == 0) IP_PROF(2, clock64());
#endif
    }
    IP_PROF(3, clock64());
    if (!pre_done) preload();                                   // (a K shorter than one round)
    for (; kstep_of<NW>(w, ks) < nsteps; ++ks) {
        const int64_t k0 = kstep_of<NW>(w, ks) * kstr;
        F fb[NT], fa[MT];
#pragma unroll
        for (int j = 0; j < NT; ++j) { if (g_on[j]) fb[j].template load_stream<WNT>(wp[j] + k0); else fb[j].zero(); }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { if (m_ok[mt]) fa[mt].load(ap[mt] + k0); else fa[mt].zero(); }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            st1[mt] = F::mma(fa[mt], f_ones, st1[mt]);
            st2[mt] = F::mma(fa[mt], fa[mt], st2[mt]);
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j * MT + mt] = F::mma(fa[mt], fb[j], acc[j * MT + mt]);
        }
    }
#pragma unroll
    for (int i = 0; i < NT * MT; ++i)
        *reinterpret_cast<float4*>(&s_acc[w][i][lane][0]) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (li == 4 * lg + r) { s_st[w][16 * mt + li][0] = st1[mt][r]; s_st[w][16 * mt + li][1] = st2[mt][r]; }
    __syncthreads();
    if (NW > 4) {
        if (tid < 128) {
            const int row = tid >> 1, c = tid & 1;
            float a = (s_st[0][row][c] + s_st[1][row][c]) + (s_st[2][row][c] + s_st[3][row][c]);
#pragma unroll
            for (int ww = 4; ww < NW; ww += 4)
                a += (s_st[ww][row][c] + s_st[ww + 1][row][c]) + (s_st[ww + 2][row][c] + s_st[ww + 3][row][c]);
            s_fin[row][c] = a;
        }
        __syncthreads();
    }
    IP_PROF(4, clock64());
    if (w >= MT * RS) {                                     // the extra waves have delivered their partial sums
        if (gate_wg) __syncthreads();                       // (the gate tiles' low-rank exchange below has one more barrier)
        return;
    }

    // wave w finalises rows [rq RPW, rq RPW + RPW) of every lane's four rows of m-tile mtf (rows m0 + 16 mtf + 4 lg + r,
    // column li of each tile): the same sums in the same order as one wave per m-tile
    const int mtf = w / RS, rq = w % RS;
    float val[NT][RPW], mu[RPW], rstd[RPW];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
        for (int r = 0; r < RPW; ++r) val[j][r] = s_acc[0][j * MT + mtf][lane][rq * RPW + r];
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) {
#pragma unroll
            for (int r = 0; r < RPW; ++r) val[j][r] += s_acc[ww][j * MT + mtf][lane][rq * RPW + r];
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = 16 * mtf + 4 * lg + rq * RPW + r;
        float a, b;
        if (NW > 4) { a = s_fin[row][0]; b = s_fin[row][1]; }
        else {
            a = (s_st[0][row][0] + s_st[1][row][0]) + (s_st[2][row][0] + s_st[3][row][0]);
            b = (s_st[0][row][1] + s_st[1][row][1]) + (s_st[2][row][1] + s_st[3][row][1]);
        }
        const float inv_k = fast_rcp((float)K);
        mu[r] = a * inv_k;
        rstd[r] = rsqrtf(fmaxf(b * inv_k - mu[r] * mu[r], 0.f) + ln_eps);
    }

    if (gate_wg) {
        const float cc1 = pre_c1[0], cc2 = pre_c2[0];
#pragma unroll
        for (int r = 0; r < RPW; ++r) s_lr[16 * mtf + 4 * lg + rq * RPW + r][li] = rstd[r] * (val[0][r] - mu[r] * cc1) + cc2;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 1; ++j) {                              // (one 16-channel tile per gate workgroup)
            const int c = (tile0 - n_direct) + li;                   // gate channel
            if (c >= Kd) continue;
            const float4 q0 = cvt4(pre_old[j][0]), q1 = cvt4(pre_old[j][1]), q2 = cvt4(pre_old[j][2]), q3 = cvt4(pre_old[j][3]);
            const float w2r[R] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
            const float bias = cvt1(pre_b2[j]);
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int row = 16 * mtf + 4 * lg + rq * RPW + r, m = m0 + row;
                float accg = bias;
#pragma unroll
                for (int jj = 0; jj < R; ++jj) accg = fmaf(s_lr[row][jj], w2r[jj], accg);
                float gv = logsigmoidf(accg) * inv_norm;
                if (has_clamp) gv = fmaxf(gv, clamp_min);
                if (m < M) gk[(int64_t)m * Kd + c] = gv;
            }
        }
        IP_PROF_FLUSH();
        return;
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = tile0 + 16 * j + li;
        const float cc1 = pre_c1[j], cc2 = pre_c2[j];
        float z[RPW];
#pragma unroll
        for (int r = 0; r < RPW; ++r) z[r] = rstd[r] * (val[j][r] - mu[r] * cc1) + cc2;   // projected value z[m, n]
        if (n >= 2 * Kd + Vd) {                              // g columns
            const int c = n - (2 * Kd + Vd);
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int m = m0 + 16 * mtf + 4 * lg + rq * RPW + r;
                if (m < M) st(g_out + (int64_t)m * Vd + c, z[r]);
            }
            continue;
        }
        // q / k / v columns: conv step on the rolled cache (W = 4) + SiLU
        T* csel; int c, D;
        if (n < Kd) { c = n; D = Kd; csel = cq; }
        else if (n < 2 * Kd) { c = n - Kd; D = Kd; csel = ck; }
        else { c = n - 2 * Kd; D = Vd; csel = cv; }
        const float4 wj = cvt4(pre_wj[j]);
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int m = m0 + 16 * mtf + 4 * lg + rq * RPW + r;
            if (m < M) {
                T* cb = csel + ((int64_t)m * D + c) * 4;
                const float4 old = cvt4(pre_old[j][r]);
                T tmp;                                       // the conv sees the projection in the model dtype
                st(&tmp, z[r]);
                const float xn = ld(&tmp);
                const float4 nw = make_float4(old.y, old.z, old.w, xn);
                st4(cb, nw);
                const float y = fmaf(wj.w, nw.w, fmaf(wj.z, nw.z, fmaf(wj.y, nw.y, wj.x * nw.x)));
                st(qkv + (int64_t)m * (2 * Kd + Vd) + n, silu(y));
            }
        }
    }
    IP_PROF_FLUSH();
}

// ---- B >= 160 (kTallMinRows): the same launch on the tall tiling (linear_tall.h: 64 rows x 64 weight rows per workgroup, the weights of a
// k-step staged through LDS once for the four waves, every wave's accumulators final).  grid.x = (2 Kd + 2 Vd) / 64 workgroups
// on q | k | v | g columns (Kd, Vd multiples of 64: a workgroup's columns lie in ONE region) + Kd / 64 GATE workgroups, each
// projecting its 64 rows on the 16 low-rank weight rows (one weight fragment per k-step) and applying the rank-16
// up-projection + bias + log-sigmoid for 64 gate channels; ceil(B / 64) row blocks (XCD-aware 1-D grid).  Epilogues as gla_inproj_kernel's.
template <typename T, int V>
__global__ __launch_bounds__(64 * TallShape<V>::NWV) void gla_inproj_tall_kernel(
    const T* __restrict__ A, const T* __restrict__ W, const float* __restrict__ c1, const float* __restrict__ c2,
    const T* __restrict__ wq, const T* __restrict__ wk, const T* __restrict__ wv, T* cq, T* ck, T* cv,
    const T* __restrict__ w2, const T* __restrict__ b2, T* __restrict__ qkv, T* __restrict__ g_out, float* __restrict__ gk,
    int M, int K, int Kd, int Vd, float ln_eps, float inv_norm, float clamp_min, int has_clamp) {
    using F = Frag<T>;
#ifdef LINA_SKINNY_PROF
    unsigned long long pr_[8] = {};
    IP_PROF(0, wall_clock64());
    IP_PROF(1, clock64());
#endif
    constexpr int R = 16, NT = 4, MTW = TallShape<V>::MTW;
    typedef typename raw4<T>::type raw_t;
    __shared__ __attribute__((aligned(16))) unsigned char s_w[TallShape<V>::LDS];   // the ONLY LDS object (see linear_tall.h)
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    const int w = wave_uniform(threadIdx.x >> 6);
    const int n_direct = 2 * Kd + 2 * Vd;
    const int nb_direct = n_direct / (16 * NT);
    int cblk, rblk;
    if (!tall_tile_of((int)blockIdx.x, nb_direct + Kd / 64, (M + TallShape<V>::ROWS - 1) / TallShape<V>::ROWS, cblk, rblk)) return;
    const int m0 = rblk * TallShape<V>::ROWS + 16 * MTW * w;       // this wave's first row
    const bool gate_wg = cblk >= nb_direct;                       // block-uniform
    const int nks = K / F::KSTEP;
    const bool wave_on = m0 < (M + 63) / 64 * 64;
    const float inv_k = fast_rcp((float)K);
    float s1[MTW][4], s2[MTW][4];

    if (gate_wg) {
        const int c0 = (cblk - nb_direct) * 64;                   // first gate channel of this workgroup
        const int nb[1] = {n_direct >> 4};
        const float cc1 = c1[n_direct + li], cc2 = c2[n_direct + li];
        f32x4 acc[1][MTW];
        tall_core_v<V, T, 1, true, MTW>(A, W, nb, nks, m0 >> 4, wave_on, s_w, acc, s1, s2);
        // the 16 low-rank activations of a row sit in 16 lanes: exchanged through this wave's slice of the (now idle) stage
        float (*s_lr)[R + 1] = reinterpret_cast<float (*)[R + 1]>(s_w) + 16 * MTW * w;
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
            float mu[4], rstd[4];
            tall_row_stats(s1[mt], s2[mt], inv_k, ln_eps, mu, rstd);
#pragma unroll
            for (int r = 0; r < 4; ++r) s_lr[16 * mt + 4 * lg + r][li] = rstd[r] * (acc[0][mt][r] - mu[r] * cc1) + cc2;
        }
        __syncthreads();
#pragma unroll 1
        for (int jj = 0; jj < 4; ++jj) {
            const int c = c0 + 16 * jj + li;                       // (Kd % 64 == 0: always a real channel)
            const float4 q0 = cvt4(ld4_raw(w2 + (int64_t)c * R)), q1 = cvt4(ld4_raw(w2 + (int64_t)c * R + 4)),
                         q2 = cvt4(ld4_raw(w2 + (int64_t)c * R + 8)), q3 = cvt4(ld4_raw(w2 + (int64_t)c * R + 12));
            const float w2r[R] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
            const float bias = ld(b2 + c);
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * mt + 4 * lg + r, m = m0 + row;
                    float accg = bias;
#pragma unroll
                    for (int k = 0; k < R; ++k) accg = fmaf(s_lr[row][k], w2r[k], accg);
                    float gv = logsigmoidf(accg) * inv_norm;
                    if (has_clamp) gv = fmaxf(gv, clamp_min);
                    if (m < M) gk[(int64_t)m * Kd + c] = gv;
                }
        }
        IP_PROF_FLUSH();
        return;
    }

    const int n0 = cblk * (16 * NT);
    int nb[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) nb[j] = (n0 >> 4) + j;
    float pre_c1[NT], pre_c2[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) { pre_c1[j] = c1[n0 + 16 * j + li]; pre_c2[j] = c2[n0 + 16 * j + li]; }
    f32x4 acc[NT][MTW];
    tall_core_v<V, T, NT, true, MTW>(A, W, nb, nks, m0 >> 4, wave_on, s_w, acc, s1, s2);
    IP_PROF(2, clock64());

    // the workgroup's 64 columns lie in one region (block-uniform): q | k | v -> conv step + SiLU, g -> stored as is
    const bool is_g = n0 >= 2 * Kd + Vd;
    const T* wsel; T* csel; int cb0, D;
    if (n0 < Kd) { cb0 = n0; D = Kd; wsel = wq; csel = cq; }
    else if (n0 < 2 * Kd) { cb0 = n0 - Kd; D = Kd; wsel = wk; csel = ck; }
    else if (!is_g) { cb0 = n0 - 2 * Kd; D = Vd; wsel = wv; csel = cv; }
    else { cb0 = n0 - (2 * Kd + Vd); D = Vd; wsel = wv; csel = cv; }
    raw_t pre_wj[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) pre_wj[j] = is_g ? raw_t() : ld4_raw(wsel + (int64_t)(cb0 + 16 * j + li) * 4);
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        float mu[4], rstd[4];
        tall_row_stats(s1[mt], s2[mt], inv_k, ln_eps, mu, rstd);
        if (is_g) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 16 * mt + 4 * lg + r;
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    if (m < M) st(g_out + (int64_t)m * Vd + cb0 + 16 * j + li, rstd[r] * (acc[j][mt][r] - mu[r] * pre_c1[j]) + pre_c2[j]);
            }
            continue;
        }
        // the rolled conv caches of this m-tile's 4 x NT (row, channel) pairs: all requested before any is used / rewritten
        raw_t old[4][NT];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 16 * mt + 4 * lg + r;
#pragma unroll
            for (int j = 0; j < NT; ++j) old[r][j] = ld4_raw(csel + ((int64_t)(m < M ? m : 0) * D + cb0 + 16 * j + li) * 4);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 16 * mt + 4 * lg + r;
            if (m >= M) continue;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int c = cb0 + 16 * j + li;
                const float z = rstd[r] * (acc[j][mt][r] - mu[r] * pre_c1[j]) + pre_c2[j];
                const float4 o4 = cvt4(old[r][j]), wj = cvt4(pre_wj[j]);
                T tmp;                                       // the conv sees the projection in the model dtype
                st(&tmp, z);
                const float xn = ld(&tmp);
                const float4 nw = make_float4(o4.y, o4.z, o4.w, xn);
                st4(csel + ((int64_t)m * D + c) * 4, nw);
                const float y = fmaf(wj.w, nw.w, fmaf(wj.z, nw.z, fmaf(wj.y, nw.y, wj.x * nw.x)));
                st(qkv + (int64_t)m * (2 * Kd + Vd) + n0 + 16 * j + li, silu(y));
            }
        }
    }
    IP_PROF_FLUSH();
}

// ---- opt-in variants 3 / 4 of the tall launch (LINA_TALL_V; round 6, measured: no faster than variant 0 -- kept as A/B builds,
// see DESIGN.md 4.3): 128 (variant 4: 64) rows x 64 columns per workgroup on EXACTLY (2 Kd + 2 Vd) / 64 x ceil(B / 128) workgroups -- 256 at
// L169 / B = 512, one per CU -- with the register-ring main loop (linear_tall.h, variant 3) and two changes to the launch itself:
//   * no gate workgroups.  The kernel above spends Kd / 64 of its column blocks (128 of 640 workgroups at B = 512) on the 16
//     low-rank gate rows: each streams its rows of A over the WHOLE contraction -- the bytes and the time of a regular workgroup --
//     for 1/64 of the flops, sixteen times over per row block.  Here the low-rank rows ride along as a FIFTH weight fragment in
//     every workgroup (+ 8 % bytes per workgroup), and a workgroup applies the rank-16 up-projection + bias + log-sigmoid to ITS
//     share of the gate channels (Kd / 64 ... : 16 of them at L169: 16 fmas + one log-sigmoid for 8 elements per thread);
//   * the rolled conv caches are requested BEFORE the main loop in the epilogue's own lane mapping (a tile row is 64 channels x
//     4 taps = one contiguous 512-byte run of the cache: 16 bytes per lane), the tile goes through LDS once, and the cache is
//     rewritten / q|k|v stored with 16- / 4-byte accesses instead of 8- / 2-byte ones.
// Bytes a CU pulls: (128 + 80) rows x K x e = 416 KB against 640 KB (2.5 workgroups of 256 KB per CU) -- the loop time of these
// kernels is that figure / ~30 B per clock (linear_tall.h).  Contract (checked by the launcher): K a multiple of kTallRwD k-steps,
// Kd, Vd multiples of 64.  Results: the same sums in the same order as gla_inproj_tall_kernel (bit-identical outputs).
template <typename T, bool WNT, int MT>
__global__ __launch_bounds__(256) void gla_inproj_tall3_kernel(
    const T* A, const T* W, const float* __restrict__ c1, const float* __restrict__ c2,
    const T* __restrict__ wq, const T* __restrict__ wk, const T* __restrict__ wv, T* cq, T* ck, T* cv,
    const T* __restrict__ w2, const T* __restrict__ b2, T* __restrict__ qkv, T* __restrict__ g_out, float* __restrict__ gk,
    int M, int K, int Kd, int Vd, float ln_eps, float inv_norm, float clamp_min, int has_clamp) {
    using F = Frag<T>;
#ifdef LINA_SKINNY_PROF
    unsigned long long pr_[8] = {};
    IP_PROF(0, wall_clock64());
    IP_PROF(1, clock64());
#endif
    constexpr int R = 16, NT = 4, D = kTallRwD, G = NT + 1, ROWS = 64 * MT;
    constexpr int ZS = 68;                                   // staging row stride in floats (272 B: rows 4 apart sit 16 banks apart)
    constexpr int WR = 16 * MT;                              // rows per wave
    constexpr int CPL = 16 / (int)sizeof(T);                 // columns per 16-byte piece of an output row
    typedef typename raw4<T>::type raw_t;
    __shared__ __attribute__((aligned(16))) unsigned char s_w[tall_rw_lds_bytes(G)];
    __shared__ __attribute__((aligned(16))) float s_z[4][16 * MT][ZS];
    __shared__ float s_lr[ROWS][R + 1];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const int w = wave_uniform(tid >> 6);
    const int n_direct = 2 * Kd + 2 * Vd;
    const int nb_direct = n_direct / (16 * NT);
    int cblk, rblk;
    if (!tall_tile_of((int)blockIdx.x, nb_direct, (M + ROWS - 1) / ROWS, cblk, rblk)) return;
    const int mw = rblk * ROWS + WR * w;                     // this wave's first row
    const int n0 = cblk * (16 * NT);
    const int nks = K / F::KSTEP;
    const bool wave_on = mw < (M + 63) / 64 * 64;            // (the packed operand is padded to whole 64-row blocks)
    const float inv_k = fast_rcp((float)K);

    // the workgroup's 64 columns lie in one region (block-uniform): q | k | v -> conv step + SiLU, g -> stored as is
    const bool is_g = n0 >= 2 * Kd + Vd;
    const T* wsel; T* csel; int cb0, Dch;
    if (n0 < Kd) { cb0 = n0; Dch = Kd; wsel = wq; csel = cq; }
    else if (n0 < 2 * Kd) { cb0 = n0 - Kd; Dch = Kd; wsel = wk; csel = ck; }
    else if (!is_g) { cb0 = n0 - 2 * Kd; Dch = Vd; wsel = wv; csel = cv; }
    else { cb0 = n0 - (2 * Kd + Vd); Dch = Vd; wsel = wv; csel = cv; }

    int nb[G];
#pragma unroll
    for (int j = 0; j < NT; ++j) nb[j] = (n0 >> 4) + j;
    nb[NT] = n_direct >> 4;
    float pre_c1[G], pre_c2[G];
#pragma unroll
    for (int j = 0; j < NT; ++j) { pre_c1[j] = c1[n0 + 16 * j + li]; pre_c2[j] = c2[n0 + 16 * j + li]; }
    pre_c1[NT] = c1[n_direct + li];
    pre_c2[NT] = c2[n_direct + li];

    // conv epilogue mapping of a wave's WR x 64 tile: item i of a lane = (row 2 i + lane / 32, channel pair lane % 32); the pair's
    // 2 x 4 cached taps are 16 contiguous bytes (bf16), a row of the tile one contiguous run of the cache
    constexpr int NI = WR / 2;
    const int cp = lane & 31, rsub = lane >> 5;
    raw_t old[NI][2], wj[2];
    // this workgroup's share of the gate channels, GC of them from c_lo on.  256 % GC == 0 (GC = 16 at Kd == Vd): a thread's
    // channel is the same for all of its elements, and its rank-16 weight row + bias are requested here, in front of the main
    // loop -- in the epilogue each element's row was a memory round trip of its own (8 in a row: 3.4 us, profiles/r06_tall_prof.txt)
    const int GC = (Kd + nb_direct - 1) / nb_direct;
    const int c_lo = cblk * GC;
    const bool gate_fixed = 256 % GC == 0;
    raw_t w2q[4];
    T b2q;
    auto prefetch = [&]() {                                  // called by the main loop behind its first D k-steps of loads
        const int c = c_lo + tid % GC;
        const int cc = c < Kd ? c : Kd - 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) w2q[q] = ld4_raw(w2 + (int64_t)cc * R + 4 * q);
        b2q = ld_raw(b2 + cc);
        if (!is_g) {
            wj[0] = ld4_raw(wsel + (int64_t)(cb0 + 2 * cp) * 4);
            wj[1] = ld4_raw(wsel + (int64_t)(cb0 + 2 * cp + 1) * 4);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int m = mw + 2 * i + rsub;
                const T* src = csel + ((int64_t)(m < M ? m : 0) * Dch + cb0 + 2 * cp) * 4;
                old[i][0] = ld4_raw(src);
                old[i][1] = ld4_raw(src + 4);
            }
        }
    };

    f32x4 acc[G][MT];
    float s1[MT][4], s2[MT][4];
    IP_PROF(7, clock64());
    tall_core_rw<T, G, true, WNT, MT, D>(A, W, nb, nks, mw >> 4, wave_on, s_w, acc, s1, s2, prefetch);
    IP_PROF(2, clock64());
#pragma unroll
    for (int q = 0; q < 4; ++q) opaque_raw(w2q[q]);          // the prefetched bits are first USED behind this point (lina_dev.h)
    if (!is_g) {
        opaque_raw(wj[0]);
        opaque_raw(wj[1]);
#pragma unroll
        for (int i = 0; i < NI; ++i) { opaque_raw(old[i][0]); opaque_raw(old[i][1]); }
    }

    // LayerNorm fold; the tile (rounded to the model dtype: the conv / the stored g see the projection in that dtype) and the
    // low-rank activations go through LDS
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        float mu[4], rstd[4];
        tall_row_stats(s1[mt], s2[mt], inv_k, ln_eps, mu, rstd);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * mt + 4 * lg + r;
            s_lr[WR * w + row][li] = rstd[r] * (acc[NT][mt][r] - mu[r] * pre_c1[NT]) + pre_c2[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                T tmp;
                st(&tmp, rstd[r] * (acc[j][mt][r] - mu[r] * pre_c1[j]) + pre_c2[j]);
                s_z[w][row][16 * j + li] = ld(&tmp);
            }
        }
    }
    __syncthreads();
    IP_PROF(3, clock64());

    if (is_g) {
        // rows of 64 columns as 16-byte pieces: item -> (row, piece)
        constexpr int PPR = 64 / CPL, NP = WR * PPR / 64;    // pieces per row, pieces per lane
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int idx = lane + 64 * i, row = idx / PPR, pc = idx % PPR;
            const int m = mw + row;
            if (m >= M) continue;
            T* dst = g_out + (int64_t)m * Vd + cb0 + CPL * pc;
            const float* zr = &s_z[w][row][CPL * pc];
#pragma unroll
            for (int q = 0; q < CPL / 4; ++q) st4(dst + 4 * q, make_float4(zr[4 * q], zr[4 * q + 1], zr[4 * q + 2], zr[4 * q + 3]));
        }
    } else {
        const float4 wa = cvt4(wj[0]), wb = cvt4(wj[1]);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int row = 2 * i + rsub, m = mw + row;
            if (m >= M) continue;
            const float z0 = s_z[w][row][2 * cp], z1 = s_z[w][row][2 * cp + 1];
            const float4 o0 = cvt4(old[i][0]), o1 = cvt4(old[i][1]);
            const float4 n0v = make_float4(o0.y, o0.z, o0.w, z0), n1v = make_float4(o1.y, o1.z, o1.w, z1);
            T* cb = csel + ((int64_t)m * Dch + cb0 + 2 * cp) * 4;
            st4(cb, n0v);
            st4(cb + 4, n1v);
            const float y0 = fmaf(wa.w, n0v.w, fmaf(wa.z, n0v.z, fmaf(wa.y, n0v.y, wa.x * n0v.x)));
            const float y1 = fmaf(wb.w, n1v.w, fmaf(wb.z, n1v.z, fmaf(wb.y, n1v.y, wb.x * n1v.x)));
            T* qd = qkv + (int64_t)m * (2 * Kd + Vd) + n0 + 2 * cp;
            st_pair(qd, silu(y0), silu(y1));
        }
    }

    IP_PROF(4, clock64());
    // this workgroup's share of the gate channels: rank-16 up-projection + bias + log-sigmoid / normaliser (+ clamp)
    for (int idx = tid; idx < ROWS * GC; idx += 256) {
        const int row = idx / GC, c = c_lo + idx % GC;
        const int m = rblk * ROWS + row;
        if (c >= Kd || m >= M) continue;
        if (!gate_fixed) {                                   // (a thread's channel changes from element to element)
#pragma unroll
            for (int q = 0; q < 4; ++q) w2q[q] = ld4_raw(w2 + (int64_t)c * R + 4 * q);
            b2q = ld_raw(b2 + c);
        }
        const float4 q0 = cvt4(w2q[0]), q1 = cvt4(w2q[1]), q2 = cvt4(w2q[2]), q3 = cvt4(w2q[3]);
        const float w2r[R] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
        float accg = cvt1(b2q);
#pragma unroll
        for (int k = 0; k < R; ++k) accg = fmaf(s_lr[row][k], w2r[k], accg);
        float gv = logsigmoidf(accg) * inv_norm;
        if (has_clamp) gv = fmaxf(gv, clamp_min);
        gk[(int64_t)m * Kd + c] = gv;
    }
    IP_PROF_FLUSH();
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
    {   // B >= 160 on packed operands: the tall tiling (see gla_inproj_tall_kernel).  Per launch at L169 (profiles/r05_tall_perf.txt;
        // 64-row split-K kernel -> tall): B = 512 31.5 -> 19.0 us, 384: 25.9 -> 16.0, 256: 17.1 -> 14.3, 192: 16.1 -> 11.9,
        // 128: 12.0 -> 12.0.  LINA_TALL=0 / 1: never / whenever the operands allow it (test hook, read per call)
        const char* tall_env = getenv("LINA_TALL");
        const int tall_mode = tall_env ? atoi(tall_env) : -1;
        const bool can = (packed & 1) && Kd % 64 == 0 && Vd % 64 == 0;
        if (can && (tall_mode == 1 || (tall_mode != 0 && B >= kTallMinRows))) {
            // variant (linear_tall.h): 0 = LDS-DMA ring, 64-row workgroups + gate workgroups (21.3 us at B = 512 in situ; 24.3 / 28.2
            // as 1 / 2); 3 = register ring, 128-row workgroups, gate folded in (gla_inproj_tall3_kernel) -- the default from 257
            // rows up when K is a whole number of its 8-k-step groups.  LINA_TALL_V forces one (test / A-B hook, read per call).
            const char* v_env = getenv("LINA_TALL_V");
            const bool v3_ok = (K / kstep) % kTallRwD == 0;
            int tv = v_env ? atoi(v_env) : LINA_TALL_DEFAULT_V;
            if ((tv == 3 || tv == 4) && !v3_ok) tv = LINA_TALL_DEFAULT_V;
            if (tv == 3 || tv == 4) {                        // 4: the same kernel on 64-row workgroups (twice as many: A/B)
                const int mt = tv == 3 ? kTallRwMT : 1;
                dim3 tgrid(tall_grid((2 * Kd + 2 * Vd) / 64, (B + 64 * mt - 1) / (64 * mt)));
#define LINA_INPROJ_T3(TT, WNTT, MTT)                                                                                 \
    LINA_LAUNCH((gla_inproj_tall3_kernel<TT, WNTT, MTT>), tgrid, dim3(256), 0, stream, (const TT*)x, (const TT*)w_in, c1, c2, \
                (const TT*)wq, (const TT*)wk, (const TT*)wv, (TT*)cq, (TT*)ck, (TT*)cv, (const TT*)w2, (const TT*)b2,   \
                (TT*)qkv, (TT*)g_out, gk, B, K, Kd, Vd, ln_eps, 1.0f / normalizer, clamp_min, has_clamp)
#define LINA_INPROJ_T3M(TT, WNTT) do { if (mt == 1) LINA_INPROJ_T3(TT, WNTT, 1); else LINA_INPROJ_T3(TT, WNTT, kTallRwMT); } while (0)
                if (dtype == LINA_F32) { if (packed == 3) LINA_INPROJ_T3M(float, true); else LINA_INPROJ_T3M(float, false); }
                else { if (packed == 3) LINA_INPROJ_T3M(bf16_t, true); else LINA_INPROJ_T3M(bf16_t, false); }
#undef LINA_INPROJ_T3M
#undef LINA_INPROJ_T3
                return check_launch("lina_gla_decode_inproj (tall, register ring)");
            }
            const int rows = tv == 1 ? TallShape<1>::ROWS : TallShape<0>::ROWS;
            dim3 tgrid(tall_grid((2 * Kd + 2 * Vd) / 64 + Kd / 64, (B + rows - 1) / rows));
#define LINA_INPROJ_TALL(TT, VV)                                                                                      \
    LINA_LAUNCH((gla_inproj_tall_kernel<TT, VV>), tgrid, dim3(64 * TallShape<VV>::NWV), 0, stream, (const TT*)x,        \
                (const TT*)w_in, c1, c2, (const TT*)wq, (const TT*)wk, (const TT*)wv, (TT*)cq, (TT*)ck, (TT*)cv,       \
                (const TT*)w2, (const TT*)b2, (TT*)qkv, (TT*)g_out, gk, B, K, Kd, Vd, ln_eps, 1.0f / normalizer,        \
                clamp_min, has_clamp)
#define LINA_INPROJ_TALL_T(TT) do { if (tv == 2) LINA_INPROJ_TALL(TT, 2); else if (tv) LINA_INPROJ_TALL(TT, 1); else LINA_INPROJ_TALL(TT, 0); } while (0)
            if (dtype == LINA_F32) LINA_INPROJ_TALL_T(float); else LINA_INPROJ_TALL_T(bf16_t);
#undef LINA_INPROJ_TALL_T
#undef LINA_INPROJ_TALL
            return check_launch("lina_gla_decode_inproj (tall)");
        }
    }
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
