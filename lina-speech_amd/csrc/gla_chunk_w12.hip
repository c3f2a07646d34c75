// gla_chunk_w12.hip -- K2 forward (bf16, one 256 x 256 head per workgroup) with SPECIALISED waves: the forward, key-gated form of
// gla_chunk_full.hip (same formulation, same LDS tiles and swizzles, same MFMA shapes; reference model/gla.py:193,195) on
// 768 threads = twelve waves = three per SIMD at <= 168 registers:
//   waves 0..7   STATE waves: 32 state columns each (2 x 16 accumulator tiles = 128 registers), steps (1) o^T = S'^T q~^T,
//                (4) S' += k~^T v and (3) o^T += v^T mask(A)^T for their two column tiles; o leaves right after step (3).  They
//                run no phase A and issue no DMA.
//   waves 8..11  UTILITY waves, one per SIMD: the next chunk's global -> LDS prefetch (16 pieces each -- an LDS-DMA
//                instruction blocks its wave until the memory system has accepted it: ~2200 clocks per chunk that sixteen
//                equal waves could only put in front of somebody's MFMAs), phase A for 64 channels each (four passes of
//                gla_chunk_full.hip's thread map, two at a time), mask(A).
// In the sixteen-wave kernel the critical path of a chunk is  phase A -> the loaders' DMA issue -> the loaders' own MFMAs;
// here the DMA issue and mask(A) run beside the state waves' MFMAs and nobody waits for a blocked wave that still has
// matrix work to do.  Timing probe of this split (tools/probes/k2_w12_probe.hip, round 4): 0.552 ms against 0.583 for the
// sixteen-wave kernel at B=64, H=4, T=4096.  Chunk cuts, renormalisation, partial chunks, h0 / ht and sequence segments as
// in gla_chunk_full.hip; every other form (state-only pass, head groups, the backward's sweeps) stays there.
#include <type_traits>
#define LINA_DMA_NT 1   // the q,k,g,v prefetch is read once: non-temporal DMA
#include <lina_dev.h>
#include "lina_common.h"

#ifdef LINA_W12_PROF
// tools-only build (tools/w12_prof.sh): per-phase shader-clock totals of workgroup 0, [wave][slot]; NOT part of the product library
__device__ unsigned long long lina_w12_prof[12 * 8];
#define W12_PROF(i) do { const unsigned long long now_ = clock64(); pacc[i] += now_ - plast; plast = now_; } while (0)
#define W12_PROF_INIT unsigned long long pacc[8] = {}, plast = clock64()
#define W12_PROF_FLUSH do { if (blockIdx.x == 0 && lane_id() == 0) for (int i_ = 0; i_ < 8; ++i_) lina_w12_prof[w * 8 + i_] = pacc[i_]; } while (0)
#else
#define W12_PROF(i) do { } while (0)
#define W12_PROF_INIT do { } while (0)
#define W12_PROF_FLUSH do { } while (0)
#endif

namespace lina {

namespace w12 {
constexpr int C = 32;
constexpr float kMaxDecay = 60.0f;
constexpr float kRenorm = 20.0f;
__device__ __forceinline__ void unpack4(const uint2 u, float (&f)[4]) {
    f[0] = bf2f((bf16_t)(u.x & 0xffff)); f[1] = bf2f((bf16_t)(u.x >> 16));
    f[2] = bf2f((bf16_t)(u.y & 0xffff)); f[3] = bf2f((bf16_t)(u.y >> 16));
}
__device__ __forceinline__ bf16x8 frag16(const bf16_t* p) { return as_bf16x8(*reinterpret_cast<const uint4*>(p)); }
}  // namespace w12

__global__ __launch_bounds__(768) void gla_chunk_bf16_h256_w12_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    const bf16_t* __restrict__ gk, bf16_t* __restrict__ o, const float* h0, float* ht, int H, int T_total, int nseg,
    int Tseg, lina_bht_strides sq, lina_bht_strides sk, lina_bht_strides sv, lina_bht_strides sg, lina_bht_strides so,
    float scale, float h0_scale) {
    using namespace w12;
    constexpr int DK = 256, DV = 256;
    // tile geometry of gla_chunk_full.hip (see there for the swizzles)
    constexpr int SQ = DK + 16, SK = DK + 16, ST = C + 16, PE = 2 * DK + 8, RAWT = (C / 2) * PE;
    constexpr int SA = 40;            // bf16 row stride of mask(A): [t block][t][32 s], 80-byte rows (16-byte reads at 8 lg)
    __shared__ __attribute__((aligned(16))) bf16_t s_qk[2 * C * SQ];    // q~ | row-major k~
    bf16_t* const s_q = s_qk;
    bf16_t* const s_k = s_qk + C * SQ;
    __shared__ __attribute__((aligned(16))) bf16_t s_A[2 * 16 * SA];    // mask(A)[t][s], rows = output tokens
    __shared__ __attribute__((aligned(16))) bf16_t s_T[(DK + DV) * ST];  // k~^T | v^T
    bf16_t* const s_kT = s_T;
    bf16_t* const s_vT = s_T + DK * ST;
    __shared__ __attribute__((aligned(16))) bf16_t s_raw[4 * RAWT];     // next chunk's q, k, g, v (DMA)
    bf16_t* const s_rg = s_raw + 2 * RAWT;
    __shared__ __attribute__((aligned(16))) float s_Rs[2 * DK];
    float* const s_R = s_Rs;              // R (log2 units) used by this chunk's phase A
    float* const s_Rn = s_Rs + DK;        // R after this chunk
    __shared__ __attribute__((aligned(8))) unsigned s_flags[4];         // {cut needed, renormalise} per chunk parity
    __shared__ int s_cut;
    __shared__ unsigned s_acnt;                                          // mask(A) tiles written (monotone: 4 per chunk)
    __shared__ unsigned s_rawcnt;                                        // waves that have finished reading the raw tiles (monotone: 12 per chunk)

    int tid = threadIdx.x, lane = tid & 63;
    const int w = wave_uniform(tid >> 6);
    const bool util = w >= 8;
    const int u = w - 8;
    int li = lane & 15, lg = lane >> 4, rp = lane & 15;
    const int slot = blockIdx.x;
    const int bh = slot / nseg, b = bh / H, h = bh % H;
    const int t_begin = (slot % nseg) * Tseg;
    const int T = min(Tseg, T_total - t_begin);

    const bf16_t* gsrc[4] = {q + b * sq.b + h * sq.h + t_begin * sq.t, k + b * sk.b + h * sk.h + t_begin * sk.t,
                             gk + b * sg.b + h * sg.h + t_begin * sg.t, v + b * sv.b + h * sv.h + t_begin * sv.t};
    const unsigned gst[4] = {(unsigned)sq.t, (unsigned)sk.t, (unsigned)sg.t, (unsigned)sv.t};
    bf16_t* ob = o + b * so.b + h * so.h + t_begin * so.t;

    for (int c = tid; c < 2 * DK; c += 768) s_Rs[c] = 0.0f;
    if (tid < 4) s_flags[tid] = 0;
    if (tid == 4) s_cut = 0;
    if (tid == 5) s_rawcnt = 0;
    if (tid == 6) s_acnt = 0;

    if (!util) {

    // =========================================================================================================
    // state waves: columns [32 w, 32 w + 32) as two tiles of 16; tile (c2, p) = rows [16p, 16p + 16) in C/D layout
    // (col = lane & 15, row = 4 (lane >> 4) + reg)
    // =========================================================================================================
    lane = lane_id(); opaque(lane); tid = w * 64 + lane; li = lane & 15; lg = lane >> 4;
    f32x4 S[2][16];
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int p = 0; p < 16; ++p) S[c2][p] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (h0) {
        const float* hp = h0 + ((int64_t)slot * DK + 4 * lg) * DV + 32 * w + li;
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int p = 0; p < 16; ++p)
#pragma unroll
                for (int r = 0; r < 4; ++r) S[c2][p][r] = hp[(16 * p + r) * DV + 16 * c2] * h0_scale;
    }
    __syncthreads();   // DMA of chunk 0 landed
    int t0 = 0, par = 0;
    // v^T of this wave's 32 columns (channel blocks 2w, 2w + 1; lane = (channel quad, row pair) as in phase A): v enters the
    // products unscaled, so the state waves -- idle during phase A -- transpose it; rows >= nv are zeroed
    auto write_vT = [&](int nv) {
        const int rp_ = lane & 15;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const int ch0 = 16 * (2 * w + blk) + 4 * (lane >> 4);
            const bf16_t* const rawp = &s_raw[3 * RAWT + rp_ * PE + ch0];
            uint2 vv[2];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                vv[rr] = *reinterpret_cast<const uint2*>(rawp + rr * DK);
                vv[rr] = (2 * rp_ + rr < nv) ? vv[rr] : make_uint2(0u, 0u);
            }
            bf16_t* const tp = &s_T[(DK + ch0) * ST + 2 * rp_];
            *reinterpret_cast<unsigned*>(tp) = byte_perm(vv[1].x, vv[0].x, 0x05040100u);
            *reinterpret_cast<unsigned*>(tp + ST) = byte_perm(vv[1].x, vv[0].x, 0x07060302u);
            *reinterpret_cast<unsigned*>(tp + 2 * ST) = byte_perm(vv[1].y, vv[0].y, 0x05040100u);
            *reinterpret_cast<unsigned*>(tp + 3 * ST) = byte_perm(vv[1].y, vv[0].y, 0x07060302u);
        }
    };
    // the prefetch of the rows from t_first on: state wave w issues row pairs 2w, 2w + 1 of q, k, g, v (8 pieces).  An LDS-DMA
    // instruction blocks its wave until the memory system has accepted it -- here, in the phase-A window, the state waves
    // have nothing else to do
    auto dma_pairs = [&](int t_first) {                         // (g and v; q and k come from the utility waves after barrier (2))
#pragma unroll
        for (int a = 2; a < 4; ++a)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int pair = 2 * w + j;
                const unsigned t = (unsigned)min(t_first + 2 * pair + (lane >> 5), T - 1);
                const unsigned boff = 2u * (t * gst[a] + 8u * (unsigned)(lane & 31));
                dma16_to_lds_async(gsrc[a], boff, &s_raw[a * RAWT + pair * PE]);
            }
    };
    // o of the chunk [tp, tp + np), packed after step (3): stored in the next window (a state wave has nothing else to do then)
    uint2 opk[2][2] = {};
    auto store_prev = [&](int tp, int np) {
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int row = 16 * nt + li;
                if (row < np) {
                    const unsigned boff = 2u * ((unsigned)(tp + row) * (unsigned)so.t + 32u * (unsigned)w + 16u * (unsigned)c2 +
                                                4u * (unsigned)lg);
                    *reinterpret_cast<uint2*>(reinterpret_cast<char*>(ob) + boff) = opk[c2][nt];
                }
            }
    };
    W12_PROF_INIT;
    int tp = 0, np = 0;                                        // the previous chunk, finished at the top of the next window
    unsigned chunk_no = 0;
    while (t0 < T) {
        const int nrem = T - t0;
        int n = min(C, nrem);
        lane = lane_id(); opaque(lane); li = lane & 15; lg = lane >> 4;
        // ---------------- the phase-A window of the state waves ----------------
        if (np > 0) store_prev(tp, np);
        write_vT(n);                                           // reads this chunk's raw v rows
        lds_wait();
        if (lane == 0) lds_atomic_add(reinterpret_cast<int*>(&s_rawcnt), 1);
        W12_PROF(0);       // step (3) + stores of the previous chunk, v^T
        if (t0 + C < T) {
            // every wave has read what it needs of the raw tiles (12 arrivals per chunk): the next chunk's rows may overwrite them
            const unsigned target = 12u * (chunk_no + 1u);
            unsigned seen;
            do { seen = (unsigned)shfl_i((int)*reinterpret_cast<volatile unsigned*>(&s_rawcnt), 0); } while (seen < target);
            dma_pairs(t0 + C);
        }
        W12_PROF(4);       // wait for the raw tiles + DMA issue
        __syncthreads();   // (2)
        W12_PROF(1);       // wait at (2)
        lane = lane_id(); opaque(lane); tid = w * 64 + lane; li = lane & 15; lg = lane >> 4;
        uint2 fl = *reinterpret_cast<const uint2*>(&s_flags[2 * par]);   // workgroup-uniform
        float rn = tid < DK ? s_Rn[tid] : 0.0f;
        if (fl.x) {
            wait_vmem();                                       // this wave's (now useless) prefetch pieces have landed ...
            __syncthreads();   // (c0) ... everybody's have: the utility waves fetch this chunk's rows again
            __syncthreads();   // (c1)
            __syncthreads();   // (c2)
            n = max(min(n, C - s_cut), 1);
            __syncthreads();   // (c3)
            if (tid == 0) { int z = 0; opaque(z); s_cut = z; }
            write_vT(n);                                       // the cut chunk's rows >= n are zero
            __syncthreads();   // (c4)
            fl.y = s_flags[2 * par + 1];                       // the rewritten tiles may have changed both
            rn = tid < DK ? s_Rn[tid] : 0.0f;
        }
        const bool renorm = fl.y != 0;
        // next chunk's R: s_R is read by phase A only (before (2) / after (3)), s_Rn is stable between (2) and (3)
        if (tid < DK) s_R[tid] = renorm ? 0.0f : rn;

        // ---------------- phase B ----------------
        const bf16_t* ktp = &s_kT[li * ST + 8 * lg];
        const bf16_t* qp = &s_q[li * SQ + 8 * (lg ^ ((li >> 2) & 3))];
        // (1) o^T = S'^T q~^T: one K = 32 MFMA per pair of 16-row state tiles (converted to bf16 in registers), then (3)
        //     o^T += v^T mask(A)^T and the packed o -- one column tile after the other, so that only one tile's accumulators are
        //     live.  mask(A) of THIS chunk is complete when its four tiles have been counted in (the utility waves write them
        //     first thing after barrier (2)); the k-slots of both operands of (3) are tokens s = 8 lg .. 8 lg + 7
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
            // (the q~ fragment of token tile 1 of pair pp is requested behind the MFMA of token tile 0 and so on: a ring of ONE pair --
            //  this wave has 168 registers and 128 of them are the state)
            bf16x8 q0 = frag16(qp), q1 = frag16(qp + 16 * SQ);
#pragma unroll
            for (int pp = 0; pp < 8; ++pp) {
                bf16x8 bb;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    bb[r] = (short)f2bf(S[c2][2 * pp][r]);
                    bb[4 + r] = (short)f2bf(S[c2][2 * pp + 1][r]);
                }
                acc0 = mfma_bf16_16x16x32(bb, q0, acc0);
                if (pp + 1 < 8) q0 = frag16(qp + 32 * (pp + 1));
                sched_fence();
                acc1 = mfma_bf16_16x16x32(bb, q1, acc1);
                if (pp + 1 < 8) q1 = frag16(qp + 16 * SQ + 32 * (pp + 1));
                sched_fence();
            }
            if (c2 == 0) {
                const unsigned target = 4u * (chunk_no + 1u);
                unsigned seen;
                do { seen = (unsigned)shfl_i((int)*reinterpret_cast<volatile unsigned*>(&s_acnt), 0); } while (seen < target);
            }
            const bf16x8 vbc = frag16(&s_vT[(32 * w + 16 * c2 + li) * ST + 8 * lg]);
            acc0 = mfma_bf16_16x16x32(vbc, frag16(&s_A[li * SA + 8 * lg]), acc0);
            acc1 = mfma_bf16_16x16x32(vbc, frag16(&s_A[(16 + li) * SA + 8 * lg]), acc1);
            opk[c2][0].x = pack_bf16x2(acc0[0] * scale, acc0[1] * scale);
            opk[c2][0].y = pack_bf16x2(acc0[2] * scale, acc0[3] * scale);
            opk[c2][1].x = pack_bf16x2(acc1[0] * scale, acc1[1] * scale);
            opk[c2][1].y = pack_bf16x2(acc1[2] * scale, acc1[3] * scale);
        }
        W12_PROF(2);       // steps (1) + (3)
        // (4) S' += k~^T v: the k~^T fragments are shared by the two column tiles
        {
            bf16x8 vb[2];
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) vb[c2] = frag16(&s_vT[(32 * w + 16 * c2 + li) * ST + 8 * lg]);
            bf16x8 tf[2];
            tf[0] = frag16(ktp);
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                if (p + 1 < 16) tf[(p + 1) & 1] = frag16(ktp + 16 * (p + 1) * ST);
                sched_fence();
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) S[c2][p] = mfma_bf16_16x16x32(tf[p & 1], vb[c2], S[c2][p]);
                sched_fence();
            }
        }
        if (renorm) {                                          // rare: S' <- e^{R} S' (R = s_Rn, the value after this chunk)
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const float4 r4 = *reinterpret_cast<const float4*>(&s_Rn[16 * p + 4 * lg]);
                const float f0 = fast_exp2(r4.x), f1 = fast_exp2(r4.y), f2 = fast_exp2(r4.z), f3 = fast_exp2(r4.w);
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) { S[c2][p][0] *= f0; S[c2][p][1] *= f1; S[c2][p][2] *= f2; S[c2][p][3] *= f3; }
            }
        }
        W12_PROF(3);       // step (4)
        wait_vmem();       // this wave's prefetch pieces (issued in the window above) have long landed
        __syncthreads();   // (3) operand tiles dead; mask(A) complete; next raw tiles landed
        W12_PROF(5);       // wait at (3)
        lane = lane_id(); opaque(lane); tid = w * 64 + lane; li = lane & 15; lg = lane >> 4;
        if (tid == 0) {                                        // read by all before (3); set again two chunks later
            int z = 0;
            opaque(z);
            s_flags[2 * par] = (unsigned)z; s_flags[2 * par + 1] = (unsigned)z;
        }
        tp = t0; np = n;
        par ^= 1;
        t0 += n;
        ++chunk_no;
    }
    lane = lane_id(); opaque(lane); li = lane & 15; lg = lane >> 4;
    store_prev(tp, np);                                        // the last chunk (T >= 1)
    W12_PROF_FLUSH;
    if (ht) {
        __syncthreads();                                       // s_R of the last chunk is visible
        lane = lane_id(); opaque(lane); li = lane & 15; lg = lane >> 4;
        float* hp = ht + ((int64_t)slot * DK + 4 * lg) * DV + 32 * w + li;
#pragma unroll
        for (int p = 0; p < 16; ++p) {                          // S = diag(e^{R}) S'
            const float4 r4 = *reinterpret_cast<const float4*>(&s_R[16 * p + 4 * lg]);
            const float f[4] = {fast_exp2(r4.x), fast_exp2(r4.y), fast_exp2(r4.z), fast_exp2(r4.w)};
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int r = 0; r < 4; ++r) hp[(16 * p + r) * DV + 16 * c2] = S[c2][p][r] * f[r];
        }
    }
    } else {
        // =====================================================================================================
        // utility waves
        // =====================================================================================================
        lane = lane_id(); opaque(lane); li = lane & 15; lg = lane >> 4; rp = lane & 15;   // (nothing per-lane is shared with the state waves' code: a common subexpression hoisted above the role split lives in a register through BOTH loops)
        auto dma_chunk = [&](int t_first, int a_lo, int a_hi) {   // row pairs 4u .. 4u + 3 of tensors [a_lo, a_hi) of q, k, g, v; rows past the end re-read row T-1
#pragma unroll
            for (int a = a_lo; a < a_hi; ++a)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int pair = 4 * u + j;
                    const unsigned t = (unsigned)min(t_first + 2 * pair + (lane >> 5), T - 1);
                    const unsigned boff = 2u * (t * gst[a] + 8u * (unsigned)(lane & 31));
                    dma16_to_lds_async(gsrc[a], boff, &s_raw[a * RAWT + pair * PE]);
                }
        };
        // ---- phase A of gla_chunk_full.hip for the channel block wv (16 channels): lane = (channel quad, row pair)
        auto read_raw = [&](int wv, uint2 (&hq)[2], uint2 (&hk)[2], uint2 (&hv)[2]) {
            const bf16_t* const rawp = &s_raw[rp * PE + 16 * wv + 4 * (lane >> 4)];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                hq[rr] = *reinterpret_cast<const uint2*>(rawp + rr * DK);
                hk[rr] = *reinterpret_cast<const uint2*>(rawp + RAWT + rr * DK);
                hv[rr] = *reinterpret_cast<const uint2*>(rawp + 3 * RAWT + rr * DK);
            }
        };
        // inclusive gate cumsum of this thread's 2 rows x 4 channels (rows >= nrem count as 0); true if the chunk's total decay
        // is too large for one chunk.  CLAMP (a single gate below -60 clamped to -60): only the cut path needs it.
        auto gate_scan = [&](auto full_tag, auto clamp_tag, int wv, float (&bc)[2][4], int nrem) {
            constexpr bool FULL = decltype(full_tag)::value, CLAMP = decltype(clamp_tag)::value;
            float g0[4], g1[4];
            const bf16_t* gp = &s_rg[rp * PE + 16 * wv + 4 * (lane >> 4)];
            unpack4(*reinterpret_cast<const uint2*>(gp), g0);
            unpack4(*reinterpret_cast<const uint2*>(gp + DK), g1);
            const bool in0 = FULL || 2 * rp < nrem, in1 = FULL || 2 * rp + 1 < nrem;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if constexpr (CLAMP) {
                    g0[c] = vmax_raw(g0[c], -kMaxDecay);
                    g1[c] = vmax_raw(g1[c], -kMaxDecay);
                }
                g1[c] = in1 ? g1[c] : 0.0f;
                bc[1][c] = (in0 ? g0[c] : 0.0f) + g1[c];
            }
            row_scan4(bc[1][0], bc[1][1], bc[1][2], bc[1][3]);
            bool viol = false;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                bc[0][c] = bc[1][c] - g1[c];
                viol |= (-bc[1][c] > kMaxDecay);
            }
            return viol;
        };
        auto write_tiles = [&](auto full_tag, int wv, const float (&bc)[2][4], int nv, int par, const uint2 (&hq)[2],
                               const uint2 (&hk)[2]) {
            constexpr bool FULL = decltype(full_tag)::value;
            constexpr float kLog2e = 1.4426950408889634f;
            const int ch0 = 16 * wv + 4 * (lane >> 4);
            uint2 kk[2];
            bf16_t* const qkp = &s_qk[2 * rp * SQ + 32 * (wv >> 1) + 8 * ((lane >> 4) ^ ((rp >> 1) & 3)) + 4 * (wv & 1)];
            bf16_t* const tp = &s_T[ch0 * ST + 2 * rp];
            const float4 R4 = *reinterpret_cast<const float4*>(&s_R[ch0]);
            const float Rc[4] = {R4.x, R4.y, R4.z, R4.w};
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int row = 2 * rp + rr;
                const bool valid = FULL || row < nv;
                float f[4], x[4], e[4], ri[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    x[c] = __builtin_fmaf(bc[rr][c], kLog2e, Rc[c]);   // (b + R) log2 e
                    e[c] = fast_exp2(x[c]);
                    ri[c] = fast_rcp(e[c]);
                }
                uint2 pq;
                unpack4(hq[rr], f);
                pq.x = pack_bf16x2(f[0] * e[0], f[1] * e[1]);
                pq.y = pack_bf16x2(f[2] * e[2], f[3] * e[3]);
                pq.x = valid ? pq.x : 0u;
                pq.y = valid ? pq.y : 0u;
                *reinterpret_cast<uint2*>(qkp + rr * SQ) = pq;
                unpack4(hk[rr], f);
                kk[rr].x = pack_bf16x2(f[0] * ri[0], f[1] * ri[1]);
                kk[rr].y = pack_bf16x2(f[2] * ri[2], f[3] * ri[3]);
                kk[rr].x = valid ? kk[rr].x : 0u;
                kk[rr].y = valid ? kk[rr].y : 0u;
                *reinterpret_cast<uint2*>(qkp + C * SQ + rr * SQ) = kk[rr];
                if (FULL ? (rr == 1 && rp == C / 2 - 1) : (row == nv - 1)) {   // owner of the chunk's last row: R after the chunk
                    bool need = false;
#pragma unroll
                    for (int c = 0; c < 4; ++c) need |= x[c] < -kRenorm * kLog2e;
                    *reinterpret_cast<float4*>(&s_Rn[ch0]) = make_float4(x[0], x[1], x[2], x[3]);
                    if (need) s_flags[2 * par + 1] = 1;
                }
            }
            *reinterpret_cast<unsigned*>(tp) = byte_perm(kk[1].x, kk[0].x, 0x05040100u);
            *reinterpret_cast<unsigned*>(tp + ST) = byte_perm(kk[1].x, kk[0].x, 0x07060302u);
            *reinterpret_cast<unsigned*>(tp + 2 * ST) = byte_perm(kk[1].y, kk[0].y, 0x05040100u);
            *reinterpret_cast<unsigned*>(tp + 3 * ST) = byte_perm(kk[1].y, kk[0].y, 0x07060302u);
        };
        using FullT = std::true_type;
        using PartT = std::false_type;

        dma_chunk(0, 0, 4);
        wait_vmem();
        __syncthreads();   // DMA of chunk 0 landed
        int t0 = 0, par = 0;
        W12_PROF_INIT;
        while (t0 < T) {
            lane = lane_id(); opaque(lane); li = lane & 15; lg = lane >> 4; rp = lane & 15;
            const int nrem = T - t0;
            int n = min(C, nrem);
            // ---------------- phase A: 64 channels = four passes of the sixteen-wave kernel's thread map, interleaved (one wave per
            // SIMD runs them: the four independent passes are what hides its LDS / transcendental latencies); v^T is written by
            // the state waves ----------------
            bool viol = false;
            {
                uint2 hq[4][2], hk[4][2], hv[2];
                float bc[4][2][4];
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const bf16_t* const rawp = &s_raw[rp * PE + 16 * (4 * u + it) + 4 * (lane >> 4)];
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr) {
                        hq[it][rr] = *reinterpret_cast<const uint2*>(rawp + rr * DK);
                        hk[it][rr] = *reinterpret_cast<const uint2*>(rawp + RAWT + rr * DK);
                    }
                }
                (void)hv;
                const bool full = nrem >= C;                    // workgroup-uniform: the mask-free form
                if (full) {
#pragma unroll
                    for (int it = 0; it < 4; ++it) viol |= gate_scan(FullT{}, PartT{}, 4 * u + it, bc[it], nrem);
                } else {
#pragma unroll
                    for (int it = 0; it < 4; ++it) viol |= gate_scan(PartT{}, PartT{}, 4 * u + it, bc[it], nrem);
                }
                // this wave has READ everything it needs of the raw tiles (q, k rows in registers, gates scanned): tell the state
                // waves, which issue the next chunk's prefetch into the same tiles while phase A goes on
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    consume4(hq[it][0].x, hq[it][0].y, hq[it][1].x, hq[it][1].y);
                    consume4(hk[it][0].x, hk[it][0].y, hk[it][1].x, hk[it][1].y);
                }
                lds_wait();
                if (lane == 0) lds_atomic_add(reinterpret_cast<int*>(&s_rawcnt), 1);
                if (full) {
#pragma unroll
                    for (int it = 0; it < 4; ++it) write_tiles(FullT{}, 4 * u + it, bc[it], C, par, hq[it], hk[it]);
                } else {
#pragma unroll
                    for (int it = 0; it < 4; ++it) write_tiles(PartT{}, 4 * u + it, bc[it], n, par, hq[it], hk[it]);
                }
            }
            if (viol) s_flags[2 * par] = 1;
            W12_PROF(0);       // phase A
            __syncthreads();   // (2) operand tiles ready; raw q,k,g,v consumed
            W12_PROF(1);       // wait at (2)
            lane = lane_id(); opaque(lane); li = lane & 15; lg = lane >> 4; rp = lane & 15;
            const unsigned cut_needed = s_flags[2 * par];       // workgroup-uniform
            bool own_prefetch = false;
            if (cut_needed) {
                // ---- rare: the decay inside this chunk exceeds e^-60 -> cut the chunk at the first such row.  The state waves
                // have already prefetched rows t0 + C .. into the raw tiles: once that has landed (barrier 0) this chunk's rows are
                // fetched again, the chunk is redone with n rows, and the prefetch of rows t0 + n .. is this wave's (below).
                __syncthreads();   // (c0) the state waves' prefetch has landed
                dma_chunk(t0, 0, 4);
                wait_vmem();
                __syncthreads();   // (c1) this chunk's raw rows are back
                int nc = C;
#pragma unroll 1
                for (int it = 0; it < 4; ++it) {
                    float bc[2][4];
                    gate_scan(PartT{}, FullT{}, 4 * u + it, bc, nrem);
#pragma unroll
                    for (int rr = 1; rr >= 0; --rr) {
                        bool bad = false;
#pragma unroll
                        for (int c = 0; c < 4; ++c) bad |= (-bc[rr][c] > kMaxDecay);
                        if (bad) nc = min(nc, 2 * rp + rr);
                    }
                }
                if (nc < C) lds_atomic_max(&s_cut, C - nc);
                __syncthreads();   // (c2)
                n = max(min(n, C - s_cut), 1);
                __syncthreads();   // (c3) everyone has read s_cut; the optimistic tiles are dead
#pragma unroll 1
                for (int it = 0; it < 4; ++it) {
                    const int wa = 4 * u + it;
                    float bc[2][4];
                    uint2 hq[2], hk[2], hv[2];
                    gate_scan(PartT{}, FullT{}, wa, bc, nrem);
                    read_raw(wa, hq, hk, hv);
                    write_tiles(PartT{}, wa, bc, n, par, hq, hk);
                }
                __syncthreads();   // (c4)
                own_prefetch = t0 + n < T;
                if (own_prefetch) dma_chunk(t0 + n, 0, 4);      // (beside the state waves' phase B, waited for before (3))
            }
            W12_PROF(2);       // DMA issue
            {
                // (2) A^T[s][t] = k~_s . q~_t, one 16 x 16 tile per utility wave (mt = u & 1: s block, nt = u >> 1: t block); its C/D
                //     layout (col t = li, rows s = 4 lg + r) goes, masked (s <= t), to row t of mask(A) as one 8-byte piece
                const int mt = u & 1, nt = u >> 1;
                const int pc = 8 * (lg ^ ((li >> 2) & 3));
                const bf16_t* kp = &s_k[(16 * mt + li) * SK + pc];
                const bf16_t* qp = &s_q[(16 * nt + li) * SQ + pc];
                bf16x8 kf[4], qf[4];
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) { kf[ks] = frag16(kp + 32 * ks); qf[ks] = frag16(qp + 32 * ks); }
                f32x4 at = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    if (ks + 3 < 8) { kf[(ks + 3) & 3] = frag16(kp + 32 * (ks + 3)); qf[(ks + 3) & 3] = frag16(qp + 32 * (ks + 3)); }
                    sched_fence();
                    at = mfma_bf16_16x16x32(kf[ks & 3], qf[ks & 3], at);
                    sched_fence();
                }
                const int t = 16 * nt + li, sb = 16 * mt + 4 * lg;
                uint2 pa;
                pa.x = pack_bf16x2(sb <= t ? at[0] : 0.0f, sb + 1 <= t ? at[1] : 0.0f);
                pa.y = pack_bf16x2(sb + 2 <= t ? at[2] : 0.0f, sb + 3 <= t ? at[3] : 0.0f);
                *reinterpret_cast<uint2*>(&s_A[(16 * nt + li) * SA + sb]) = pa;
                lds_wait();
                if (lane == 0) lds_atomic_add(reinterpret_cast<int*>(&s_acnt), 1);   // the state waves take step (3) before barrier (3)
            }
            W12_PROF(3);       // mask(A)
            if (!cut_needed && t0 + C < T) {
                dma_chunk(t0 + C, 0, 2);                        // q, k of the next chunk (g, v: the state waves, in their window)
                own_prefetch = true;
            }
            if (own_prefetch) wait_vmem();
            W12_PROF(4);       // wait_vmem
            __syncthreads();   // (3) the next raw tiles have landed (the state waves waited for theirs); operand tiles dead; mask(A) complete
            W12_PROF(5);       // wait at (3)
            par ^= 1;
            t0 += n;
        }
        W12_PROF_FLUSH;
        if (ht) __syncthreads();                                // (pairs with the state waves' barrier in front of the final state)
    }
}

// one workgroup per (head, segment) slot; the caller has checked the shapes (full_ok of gla_chunk_full.hip, Dk = Dv = 256)
int launch_chunk_w12(const void* q, const void* k, const void* v, const void* gk, void* o, const float* h0, float* ht,
                     int64_t slots, int H, int T, int nseg, int Tseg, lina_bht_strides sq, lina_bht_strides sk,
                     lina_bht_strides sv, lina_bht_strides sg, lina_bht_strides so, float scale, float h0_scale,
                     lina_stream_t stream) {
    LINA_LAUNCH(gla_chunk_bf16_h256_w12_kernel, dim3((unsigned)slots), dim3(768), 0, stream, (const bf16_t*)q,
                (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)gk, (bf16_t*)o, h0, ht, H, T, nseg, Tseg, sq, sk, sv, sg, so,
                scale, h0_scale);
    return LINA_OK;
}

}  // namespace lina

#ifdef LINA_W12_PROF
extern "C" int lina_w12_prof_read(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(lina_w12_prof), sizeof(unsigned long long) * 12 * 8);
}
#endif
