// gla_chunk_pipe.hip -- K2p (bf16, Dk = Dv = 256): the chunk-wise GLA forward as a SOFTWARE PIPELINE over 16-token chunks.
// One workgroup per (b,h), the 256 x 256 fp32 state resident in MFMA accumulators for the whole sequence (as in
// gla_chunk_full.hip); HBM traffic is exactly the algorithmic q,k,g,v in + o out.
//
// Replaces fla.ops.gla.chunk_gla / fused_chunk_gla (reference model/gla.py:193,195) for the L169 head shape.
//
// Why a second kernel: in gla_chunk_full.hip (C = 32, 155.6 of 160 KB LDS) the raw q,k,g,v tiles
// are single-buffered, so the prefetch of chunk n+1 can only be issued after phase A of chunk n has consumed them and must
// land before phase A of chunk n+1 starts: DMA issue (~2200 clocks on the four loader waves) + HBM latency sit on the
// critical path of every chunk and twelve waves wait 45 % of the time at two barriers.  With C = 16 every tile halves and
// BOTH the raw tiles and the operand tiles fit twice (138 KB):
//
//   iteration v (ONE barrier per 16 tokens); the sixteen waves split into two sets of eight (two waves of either set on each
//   SIMD), whose roles alternate every iteration:
//     A-waves   fin: o(v-1) += v^T mask(A)(v-1), store;  A(v+1): gate scan + scaled operands of the NEXT chunk, raw[(v+1)&1] ->
//               ops[(v+1)&1], with the C = 32 kernel's thread map (a thread owns 2 rows x 4 channels: the same VALU work per
//               token as there; the first version, one row per thread on all sixteen waves, was VALU-bound at 0.76 ms);  B(v)
//     B-waves   DMA: raw chunk v+2 -> raw[v&1], four pieces per wave (a whole iteration to land);  fin;  mask(A)(v) (one wave);  B(v)
//     B(v)      MFMAs from ops[v&1]:  o^T = S'^T q^^T (8 x K=32),  S' += k^^T v (16 x K=16)
//   so a SIMD's VALU (phase A) and its matrix pipe (phase B of the other set) are busy at the same time.
//
// Same formulation as gla_chunk_full.hip: UN-normalised state S' with S = diag(e^R) S', q^ = q e^{b+R}, k^ = k e^{-(b+R)},
// o = scale (q^ S' + mask(q^ k^^T) v), S' += k^^T v, R += b_last, rows rescaled when R < -20; a chunk whose in-chunk decay
// exceeds e^-60 is cut at the first such row ("virtual chunks": the rest of the raw chunk is processed by the next
// iteration, exact for reset gates).  Token contractions use v_mfma_f32_16x16x16_bf16 (K = 16 = the chunk).
#include <type_traits>
#ifndef LINA_DMA_NT
#define LINA_DMA_NT 1   // the q,k,g,v prefetch is read once: non-temporal DMA (0.582 -> 0.572 ms at B=64,H=4,T=4096, round 4)
#endif
#include <lina_dev.h>
#include "lina_common.h"

#ifdef LINA_K2_PROF
// tools-only build (tools/k2_tune.sh): per-phase shader-clock totals of workgroup 0, [wave][slot] + per-workgroup totals
__device__ unsigned long long lina_k2p_prof[16 * 16 + 1024];
#define K2P_PROF(i) do { const unsigned long long now_ = clock64(); pacc[i] += now_ - plast; plast = now_; } while (0)
#else
#define K2P_PROF(i) do { } while (0)
#endif

namespace lina {

constexpr int kPipeC = 16;
constexpr float kPipeMaxDecay = 60.0f;
constexpr float kPipeRenorm = 20.0f;

namespace pipe_detail {
__device__ __forceinline__ void unpack4(const uint2 u, float (&f)[4]) {
    f[0] = bf2f((bf16_t)(u.x & 0xffff)); f[1] = bf2f((bf16_t)(u.x >> 16));
    f[2] = bf2f((bf16_t)(u.y & 0xffff)); f[3] = bf2f((bf16_t)(u.y >> 16));
}
__device__ __forceinline__ bf16x8 frag16(const bf16_t* p) { return as_bf16x8(*reinterpret_cast<const uint4*>(p)); }
__device__ __forceinline__ bf16x4 frag8(const bf16_t* p) { return as_bf16x4(*reinterpret_cast<const uint2*>(p)); }
}  // namespace pipe_detail

// blockIdx.x = (b*H + h) * nseg + seg handles tokens [seg*Tseg, min(T_total, (seg+1)*Tseg)); h0 / ht are indexed by blockIdx.x
// (nseg = 1: one workgroup per head over the whole sequence; nseg > 1: pass 2 of the segment-parallel forward).
__global__ __launch_bounds__(1024) void gla_chunk_pipe_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    const bf16_t* __restrict__ gk, bf16_t* __restrict__ o, const float* h0, float* ht, int H, int T_total, int nseg,
    int Tseg, lina_bht_strides sq, lina_bht_strides sk, lina_bht_strides sv, lina_bht_strides sg, lina_bht_strides so,
    float scale, float h0_scale) {
    using namespace pipe_detail;
    constexpr int DK = 256, C = kPipeC;
    // q^ / k^ row-major tiles exactly as in gla_chunk_full.hip: 544-byte rows, channels of each group of 32 in the order
    // [0-3,16-19,4-7,20-23,...] (what the state tiles' C/D layout gives the k-slots), 16-byte piece index XOR (row>>2)&3:
    // every operand read is one conflict-free ds_read_b128.
    constexpr int SQ = DK + 16;
    constexpr int OPQ = 2 * C * SQ;                 // elements of one {q^ | k^} buffer
    // k^^T | v^T: [512 rows = channels | columns][16 tokens], 32-byte rows, token quad t>>2 stored at position (t>>2) ^ 2*((row>>3)&1):
    // the 16 rows x 2 quads of one half-wave ds_read_b64 cover all 64 banks exactly once.
    constexpr int OPT = 2 * DK * C;                 // elements of one {k^^T | v^T} buffer
    // raw tiles: one DMA piece = one row pair (1 KiB) with the two rows INTERLEAVED at 16-byte granularity (lane -> (piece
    // l>>1 of row l&1)); pair p starts at byte 1024 p + 64 (p>>1) + 16 (p&1): the 8 row pairs x 4 channel quads of one phase-A
    // half-wave read (8 bytes each) hit 64 distinct banks.
    constexpr int RAWT = (C / 2) * 2 * DK + 128;    // per tensor (elements)
    constexpr int RAWB = 4 * RAWT;                  // per buffer {q, k, g, v}
    __shared__ __attribute__((aligned(16))) bf16_t s_qk[2 * OPQ];
    __shared__ __attribute__((aligned(16))) bf16_t s_T[2 * OPT];
    __shared__ __attribute__((aligned(16))) bf16_t s_A[2 * 64 * 4];   // mask(A)^T of chunk v as the B operand of the K=16 MFMA: lane l -> 4 values
    __shared__ __attribute__((aligned(16))) bf16_t s_raw[2 * RAWB];
    __shared__ __attribute__((aligned(16))) float s_Rn[2 * DK];       // R (log2 units) AFTER virtual chunk v, [v&1][channel]
    // {cut needed, renormalise} of virtual chunk v at [2(v&1)], [2(v&1)+1], valid iff == v + 1 (generation tag: no reset, no race
    // between the reset and the next writer two chunks later)
    __shared__ __attribute__((aligned(8))) int s_flag[4];
    __shared__ int s_cut;

    int lane = threadIdx.x & 63;
    const int w = wave_uniform((int)threadIdx.x >> 6);        // wave index in an SGPR for the whole kernel
    int li = lane & 15, lg = lane >> 4;
    const int slot = blockIdx.x;
    const int bh = slot / nseg, b = bh / H, h = bh % H;
    const int t_begin = (slot % nseg) * Tseg;
    const int T = min(Tseg, T_total - t_begin);               // tokens of this segment (>= 1 by construction)
    const int NJ = (T + C - 1) / C;                           // raw chunks
    const int wset = (w >> 2) & 1;                            // the two wave sets: {0-3, 8-11} and {4-7, 12-15} (two waves of each per SIMD)
    const int wa = 4 * (w >> 3) + (w & 3);                    // index of this wave inside its set, 0..7
    auto pair_base = [](int p) { return p * (2 * DK) + 32 * (p >> 1) + 8 * (p & 1); };   // elements

    // ---- state: wave w owns columns [16w, 16w+16); tile p = rows [16p, 16p+16) in C/D layout (col = li, row = 4 lg + reg) ----
    f32x4 S[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) S[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (h0) {
        const float* hp = h0 + ((int64_t)slot * DK + 4 * lg) * DK + 16 * w + li;
#pragma unroll
        for (int p = 0; p < 16; ++p)
#pragma unroll
            for (int r = 0; r < 4; ++r) S[p][r] = hp[(16 * p + r) * DK] * h0_scale;
    }

    const bf16_t* const gq = q + b * sq.b + h * sq.h + t_begin * sq.t;
    const bf16_t* const gkk = k + b * sk.b + h * sk.h + t_begin * sk.t;
    const bf16_t* const gg = gk + b * sg.b + h * sg.h + t_begin * sg.t;
    const bf16_t* const gv = v + b * sv.b + h * sv.h + t_begin * sv.t;
    bf16_t* const ob = o + b * so.b + h * so.h + t_begin * so.t;

    // Prefetch of raw chunk j into raw[j&1]: 32 pieces (4 tensors x 8 row pairs), four per wave of the set whose turn it is NOT to
    // run phase A (wave wa: tensor wa>>1, pairs 4 (wa&1) .. +3).  Lane l fetches the 16-byte piece l>>1 of row 2*pair + (l&1);
    // rows past the end re-read row T-1 (phase A masks them).  Addressed as (uniform 64-bit base) + (32-bit byte offset per lane),
    // issued through inline assembly, waited for by hand.
    const unsigned stq = (unsigned)sq.t, stk = (unsigned)sk.t, stg = (unsigned)sg.t, stv = (unsigned)sv.t;   // < 2^20 (launcher)
    auto dma_pieces = [&](const bf16_t* src, unsigned st, bf16_t* dst_tensor, int j) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pair = 4 * (wa & 1) + i;
            const unsigned t = (unsigned)min(C * j + 2 * pair + (lane & 1), T - 1);
            const unsigned boff = 2u * (t * st + 8u * (unsigned)(lane >> 1));
            dma16_to_lds_async(src, boff, &dst_tensor[pair_base(pair)]);
        }
    };
    auto dma_chunk = [&](int j) {
        bf16_t* const dst = &s_raw[(j & 1) * RAWB];
        const int a = wa >> 1;                                 // wave-uniform; the tensors are compile-time in each arm
        if (a == 0) dma_pieces(gq, stq, dst, j);
        else if (a == 1) dma_pieces(gkk, stk, dst + RAWT, j);
        else if (a == 2) dma_pieces(gg, stg, dst + 2 * RAWT, j);
        else dma_pieces(gv, stv, dst + 3 * RAWT, j);
    };

    // ---- phase A thread map (eight waves): wave wa <-> channels [32 wa, 32 wa + 32); lane = (channel quad cq = lane>>3, row pair
    // rp = lane&7): a 16-lane DPP row holds two channel quads x 8 row pairs, so the gate scan is the 16-lane scan (four fused DPP
    // adds per value) + one masked subtraction of the first half's total in the second half; the thread owns two ADJACENT tokens
    // and writes k^^T / v^T as 4-byte pieces.
    // inclusive gate cumsum over the rows [lo, .] of raw[rbuf] for this thread's 2 rows x 4 channels (rows outside [lo, end) count
    // as 0); true if the in-chunk decay at this thread's second row is too large for one chunk (monotone in the row)
    auto gate_scan = [&](float (&bc)[2][4], int rbuf, int lo, int end) -> bool {
        const int rp = lane & 7, cq = lane >> 3;
        const bf16_t* gp = &s_raw[rbuf * RAWB + 2 * RAWT + pair_base(rp) + 16 * (4 * wa + (cq >> 1)) + 4 * (cq & 1)];
        float g0[4], g1[4];
        unpack4(*reinterpret_cast<const uint2*>(gp), g0);
        unpack4(*reinterpret_cast<const uint2*>(gp + 8), g1);
        const bool in0 = 2 * rp >= lo && 2 * rp < end, in1 = 2 * rp + 1 >= lo && 2 * rp + 1 < end;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            g0[c] = in0 ? vmax_raw(g0[c], -kPipeMaxDecay) : 0.0f;
            g1[c] = in1 ? vmax_raw(g1[c], -kPipeMaxDecay) : 0.0f;
            bc[1][c] = g0[c] + g1[c];                         // the row pair's sum
        }
        row_scan4(bc[1][0], bc[1][1], bc[1][2], bc[1][3]);
        row_half_fix4(bc[1][0], bc[1][1], bc[1][2], bc[1][3]);
        bool viol = false;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            bc[0][c] = bc[1][c] - g1[c];                      // the pair's first row
            viol |= (-bc[1][c] > kPipeMaxDecay);
        }
        return viol;
    };
    // first row of this thread's pair whose in-chunk decay is too large (C if none): the rare path's cut position
    auto first_bad = [&](const float (&bc)[2][4]) -> int {
        int nc = C;
#pragma unroll
        for (int rr = 1; rr >= 0; --rr) {
            bool bad = false;
#pragma unroll
            for (int c = 0; c < 4; ++c) bad |= (-bc[rr][c] > kPipeMaxDecay);
            if (bad) nc = 2 * (lane & 7) + rr;
        }
        return nc;
    };
    // operand tiles of a virtual chunk = rows [lo, hi) of raw[rbuf] -> ops[pn]; rows outside are zeroed.  The owner of row
    // hi-1 publishes R after the chunk (s_Rn[pn]) and the renormalisation flag.  q^ carries NO 1/sqrt(Dk) (applied to o).
    auto write_tiles = [&](const float (&bc)[2][4], int pn, int rbuf, int lo, int hi, bool zero_r, int rpar, int gen) {
        constexpr float kLog2e = 1.4426950408889634f;
        const int rp = lane & 7, cq = lane >> 3, ch0 = 32 * wa + 4 * cq;
        const bf16_t* const rawp = &s_raw[rbuf * RAWB + pair_base(rp) + 16 * (4 * wa + (cq >> 1)) + 4 * (cq & 1)];
        float4 R4 = *reinterpret_cast<const float4*>(&s_Rn[rpar * DK + ch0]);
        if (zero_r) R4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const float Rc[4] = {R4.x, R4.y, R4.z, R4.w};
        uint2 kk[2], vv[2];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int row = 2 * rp + rr;
            const bool valid = row >= lo && row < hi;
            float x[4], e[4], ri[4], f[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                x[c] = __builtin_fmaf(bc[rr][c], kLog2e, Rc[c]);   // (b + R) log2 e: |b| <= 60 in a legal chunk, |R| <= kPipeRenorm + 60
                e[c] = fast_exp2(x[c]);
                ri[c] = fast_rcp(e[c]);
            }
            // column of this thread's channel quad in the q^ / k^ tiles: group wa (32 channels), piece (cq&3) ^ ((row>>2)&3), half cq>>2
            bf16_t* const qkp = &s_qk[pn * OPQ + row * SQ + 32 * wa + 8 * ((cq & 3) ^ ((row >> 2) & 3)) + 4 * (cq >> 2)];
            uint2 pq;
            unpack4(*reinterpret_cast<const uint2*>(rawp + 8 * rr), f);
            pq.x = pack_bf16x2(f[0] * e[0], f[1] * e[1]);      // rows outside [lo, hi): zeroed as packed words
            pq.y = pack_bf16x2(f[2] * e[2], f[3] * e[3]);
            pq.x = valid ? pq.x : 0u; pq.y = valid ? pq.y : 0u;
            *reinterpret_cast<uint2*>(qkp) = pq;
            unpack4(*reinterpret_cast<const uint2*>(rawp + RAWT + 8 * rr), f);
            kk[rr].x = pack_bf16x2(f[0] * ri[0], f[1] * ri[1]);
            kk[rr].y = pack_bf16x2(f[2] * ri[2], f[3] * ri[3]);
            kk[rr].x = valid ? kk[rr].x : 0u; kk[rr].y = valid ? kk[rr].y : 0u;
            *reinterpret_cast<uint2*>(qkp + C * SQ) = kk[rr];
            const uint2 rv = *reinterpret_cast<const uint2*>(rawp + 3 * RAWT + 8 * rr);
            vv[rr].x = valid ? rv.x : 0u; vv[rr].y = valid ? rv.y : 0u;
            if (row == hi - 1) {                               // owner of the chunk's last row: R after the chunk
                bool need = false;
#pragma unroll
                for (int c = 0; c < 4; ++c) need |= x[c] < -kPipeRenorm * kLog2e;
                *reinterpret_cast<float4*>(&s_Rn[pn * DK + ch0]) = make_float4(x[0], x[1], x[2], x[3]);
                if (need) s_flag[2 * pn + 1] = gen;
            }
        }
        // transposed tiles: element (channel ch0+i, tokens 2rp, 2rp+1) = one 4-byte word at row ch0+i, token-quad position
        // (rp>>1) ^ 2*((row>>3)&1) with (row>>3)&1 = (cq>>1)&1, word (rp&1) of the quad
        bf16_t* const tp = &s_T[pn * OPT + ch0 * C + 4 * ((rp >> 1) ^ (2 * ((cq >> 1) & 1))) + 2 * (rp & 1)];
        *reinterpret_cast<unsigned*>(tp) = byte_perm(kk[1].x, kk[0].x, 0x05040100u);
        *reinterpret_cast<unsigned*>(tp + C) = byte_perm(kk[1].x, kk[0].x, 0x07060302u);
        *reinterpret_cast<unsigned*>(tp + 2 * C) = byte_perm(kk[1].y, kk[0].y, 0x05040100u);
        *reinterpret_cast<unsigned*>(tp + 3 * C) = byte_perm(kk[1].y, kk[0].y, 0x07060302u);
        *reinterpret_cast<unsigned*>(tp + DK * C) = byte_perm(vv[1].x, vv[0].x, 0x05040100u);
        *reinterpret_cast<unsigned*>(tp + DK * C + C) = byte_perm(vv[1].x, vv[0].x, 0x07060302u);
        *reinterpret_cast<unsigned*>(tp + DK * C + 2 * C) = byte_perm(vv[1].y, vv[0].y, 0x05040100u);
        *reinterpret_cast<unsigned*>(tp + DK * C + 3 * C) = byte_perm(vv[1].y, vv[0].y, 0x07060302u);
    };
    // phase A of virtual chunk (raw chunk j, first row lo), optimistic (all rows up to the raw chunk's end): sets the cut
    // flag (generation gen) when the chunk must be cut.  Run by the eight waves of one set.
    auto phase_a = [&](int pn, int j, int lo, bool zero_r, int rpar, int gen) {
        const int end = min(C, T - C * j);
        float bc[2][4];
        if (gate_scan(bc, j & 1, lo, end)) s_flag[2 * pn] = gen;
        write_tiles(bc, pn, j & 1, lo, end, zero_r, rpar, gen);
    };

    if (threadIdx.x < 4) s_flag[threadIdx.x] = 0;
    if (threadIdx.x == 4) s_cut = 0;
    if (wset == 0) dma_chunk(0);
    else if (NJ > 1) dma_chunk(1);
    wait_vmem();
    __syncthreads();                                           // raw chunks 0 (and 1) landed; flags initialised
    if (wset == 1) phase_a(0, 0, 0, true, 0, 1);               // (iteration v's phase A belongs to set v&1; chunk 0's to set 1)
    __syncthreads();                                           // ops[0] complete

    int vj = 0, vlo = 0;                                       // current virtual chunk: raw chunk, first row
    int ptok = 0, plo = 0, phi = 0;                            // previous virtual chunk: first token of its raw chunk, rows [plo, phi)
    f32x4 accp = {0.f, 0.f, 0.f, 0.f};                         // o^T of the previous chunk before its intra-chunk term
    bf16x4 vbp = as_bf16x4(make_uint2(0u, 0u));                // v^T fragment of the previous chunk (A operand of the intra-chunk MFMA)
    // o^T += v^T mask(A)^T of the previous chunk, then o straight from the accumulators: a lane holds 4 consecutive columns
    // (16w + 4lg ..) of token li = one 8-byte store; the 16 waves' 32-byte pieces of a 512-byte row meet in L2.
    auto finish_prev = [&](int ppar) {
        accp = mfma_bf16_16x16x16(vbp, frag8(&s_A[ppar * 256 + lane * 4]), accp);
        uint2 po;
        po.x = pack_bf16x2(accp[0] * scale, accp[1] * scale);
        po.y = pack_bf16x2(accp[2] * scale, accp[3] * scale);
        if (li >= plo && li < phi) {
            const unsigned boff = 2u * ((unsigned)(ptok + li) * (unsigned)so.t + 16u * (unsigned)w + 4u * (unsigned)lg);
            *reinterpret_cast<uint2*>(reinterpret_cast<char*>(ob) + boff) = po;
        }
    };

#ifdef LINA_K2_PROF
    unsigned long long pacc[8] = {}, plast = clock64();
    const unsigned long long pstart = plast;
#endif
    for (int vi = 0;; ++vi) {
        // nothing per-lane is carried across iterations except the accumulators: the lane index comes from v_mbcnt
        lane = lane_id();
        opaque(lane);
        li = lane & 15; lg = lane >> 4;
        const int par = vi & 1;
        const int end = min(C, T - C * vj);
        int hi = end;
        if (s_flag[2 * par] == vi + 1) {
            // ---- rare: the decay inside this chunk exceeds e^-60 -> cut it at the first such row; the rest of the raw chunk
            //      becomes the next virtual chunk.  Workgroup-uniform branch.
            // (the rewrite is the work of the set that ran this chunk's optimistic phase A; everybody meets at the barriers)
            const bool mine = wset == ((vi & 1) ^ 1);
            float bc[2][4];
            if (mine) {
                gate_scan(bc, vj & 1, vlo, end);
                const int nc = first_bad(bc);
                if (nc < C) lds_atomic_max(&s_cut, C - nc);    // first bad row of the workgroup = C - max
            }
            __syncthreads();
            hi = max(min(end, C - s_cut), vlo + 1);            // row vlo itself is never bad (one clamped gate)
            __syncthreads();                                   // everyone has read s_cut; the optimistic tiles are dead
            if (threadIdx.x == 0) { int z = 0; opaque(z); s_cut = z; }
            const bool zr = vi == 0 || s_flag[2 * (par ^ 1) + 1] == vi;   // R this chunk started from (intact until phase A below)
            if (mine) write_tiles(bc, par, vj & 1, vlo, hi, zr, par ^ 1, vi + 1);
            __syncthreads();
        }
        const bool renorm = s_flag[2 * par + 1] == vi + 1;     // workgroup-uniform
        const bool split = hi < end;
        const int nj = split ? vj : vj + 1, nlo = split ? hi : 0;
        const bool more = nj < NJ;
        const bool a_wave = wset == (vi & 1);                  // this iteration's phase-A set (wave-uniform)
        // raw chunk vj is consumed (phase A of its last virtual chunk ran in the previous iteration): its buffer takes chunk vj+2
        if (!a_wave && !split && vj + 2 < NJ) dma_chunk(vj + 2);
        K2P_PROF(0);
        if (vi > 0) finish_prev(par ^ 1);
        K2P_PROF(1);

        if (a_wave && more) phase_a(par ^ 1, nj, nlo, renorm, par, vi + 2);
        K2P_PROF(2);

        // ---------------- phase B of virtual chunk vi ----------------
        if (!a_wave && wa == 0) {
            // mask(A)^T[s][t] = k^_s . q^_t (s <= t), once per workgroup: C/D layout (col t = li, rows s = 4lg + r) IS the B
            // operand layout of the K=16 MFMA of the intra-chunk term -> each lane masks and stores its 4 values as 8 bytes
            wave_priority<2>();
            const int pc = 8 * (lg ^ ((li >> 2) & 3));
            const bf16_t* kp = &s_qk[par * OPQ + C * SQ + li * SQ + pc];
            const bf16_t* qp = &s_qk[par * OPQ + li * SQ + pc];
            bf16x8 kf[4], qf[4];
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) { kf[ks] = frag16(kp + 32 * ks); qf[ks] = frag16(qp + 32 * ks); }
            f32x4 at[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                if (ks + 3 < 8) { kf[(ks + 3) & 3] = frag16(kp + 32 * (ks + 3)); qf[(ks + 3) & 3] = frag16(qp + 32 * (ks + 3)); }
                sched_fence();
                at[ks & 1] = mfma_bf16_16x16x32(kf[ks & 3], qf[ks & 3], at[ks & 1]);
                sched_fence();
            }
            const int sb = 4 * lg;
            uint2 pa;
            pa.x = pack_bf16x2(sb <= li ? at[0][0] + at[1][0] : 0.0f, sb + 1 <= li ? at[0][1] + at[1][1] : 0.0f);
            pa.y = pack_bf16x2(sb + 2 <= li ? at[0][2] + at[1][2] : 0.0f, sb + 3 <= li ? at[0][3] + at[1][3] : 0.0f);
            *reinterpret_cast<uint2*>(&s_A[par * 256 + lane * 4]) = pa;
            wave_priority<0>();
        }
        K2P_PROF(3);
        // (1) o^T = S'_old^T-tiles . q^^T: one K = 32 MFMA per pair of 16-row state tiles (converted to bf16 in registers);
        //     two accumulators alternate (independent chains), operand reads issued QA tile pairs ahead by hand
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        {
            const bf16_t* qp = &s_qk[par * OPQ + li * SQ + 8 * (lg ^ ((li >> 2) & 3))];
            constexpr int QA = 3;
            bf16x8 qf[4];
#pragma unroll
            for (int pp = 0; pp < QA; ++pp) qf[pp] = frag16(qp + 32 * pp);
#pragma unroll
            for (int pp = 0; pp < 8; ++pp) {
                if (pp + QA < 8) qf[(pp + QA) & 3] = frag16(qp + 32 * (pp + QA));
                sched_fence();
                bf16x8 bb;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    bb[r] = (short)f2bf(S[2 * pp][r]);
                    bb[4 + r] = (short)f2bf(S[2 * pp + 1][r]);
                }
                acc[pp & 1] = mfma_bf16_16x16x32(bb, qf[pp & 3], acc[pp & 1]);
                sched_fence();
            }
        }
        K2P_PROF(4);
        // (4) S' += k^^T v: 16 K=16 MFMAs, A = k^^T of state row tile p (tokens 4lg..4lg+3 of row 16p+li), B = v^T of this wave's
        //     columns -- the same fragment is the A operand of the intra-chunk term after the barrier
        {
            const int sw = 4 * (lg ^ (2 * ((li >> 3) & 1)));
            const bf16_t* ktp = &s_T[par * OPT + li * C + sw];
            const bf16x4 vb = frag8(&s_T[par * OPT + (DK + 16 * w + li) * C + sw]);
            constexpr int TA = 5;
            bf16x4 tf[8];
#pragma unroll
            for (int p = 0; p < TA; ++p) tf[p] = frag8(ktp + 16 * p * C);
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                if (p + TA < 16) tf[(p + TA) & 7] = frag8(ktp + 16 * (p + TA) * C);
                sched_fence();
                S[p] = mfma_bf16_16x16x16(tf[p & 7], vb, S[p]);
                sched_fence();
            }
            vbp = vb;
        }
        if (renorm) {                                          // rare: S' <- e^{R} S' (R = s_Rn[par], the value after this chunk)
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const float4 r4 = *reinterpret_cast<const float4*>(&s_Rn[par * DK + 16 * p + 4 * lg]);
                S[p][0] *= fast_exp2(r4.x); S[p][1] *= fast_exp2(r4.y); S[p][2] *= fast_exp2(r4.z); S[p][3] *= fast_exp2(r4.w);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) accp[r] = acc[0][r] + acc[1][r];
        ptok = C * vj; plo = vlo; phi = hi;
        K2P_PROF(5);

        wait_vmem();                                           // this wave's prefetch pieces have landed ...
        K2P_PROF(6);
        __syncthreads();                                       // ... and everybody's; ops[par^1], mask(A)(vi), R, flags visible
        K2P_PROF(7);
        if (!more) {
            lane = lane_id(); opaque(lane); li = lane & 15; lg = lane >> 4;
            finish_prev(par);
            if (ht) {                                          // S = diag(e^{R}) S'
                float* hp = ht + ((int64_t)slot * DK + 4 * lg) * DK + 16 * w + li;
#pragma unroll
                for (int p = 0; p < 16; ++p) {
                    float4 r4 = *reinterpret_cast<const float4*>(&s_Rn[par * DK + 16 * p + 4 * lg]);
                    if (renorm) r4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    hp[(16 * p + 0) * DK] = S[p][0] * fast_exp2(r4.x);
                    hp[(16 * p + 1) * DK] = S[p][1] * fast_exp2(r4.y);
                    hp[(16 * p + 2) * DK] = S[p][2] * fast_exp2(r4.z);
                    hp[(16 * p + 3) * DK] = S[p][3] * fast_exp2(r4.w);
                }
            }
            break;
        }
        vj = nj; vlo = nlo;
    }
#ifdef LINA_K2_PROF
    if (blockIdx.x == 0 && lane_id() == 0)
        for (int i = 0; i < 8; ++i) lina_k2p_prof[w * 16 + i] = pacc[i];
    if (blockIdx.x < 1024 && w == 0 && lane_id() == 0) lina_k2p_prof[256 + blockIdx.x] = clock64() - pstart;
#endif
}

// Launcher used by launch_chunk_full (gla_chunk_full.hip) for one head per workgroup (Dk = Dv = 256).
int launch_chunk_pipe(const void* q, const void* k, const void* v, const void* gk, void* o, const float* h0, float* ht,
                      int slots, int H, int T, int nseg, int Tseg, lina_bht_strides sq, lina_bht_strides sk,
                      lina_bht_strides sv, lina_bht_strides sg, lina_bht_strides so, float scale, lina_stream_t stream) {
    LINA_LAUNCH(gla_chunk_pipe_kernel, dim3((unsigned)slots), dim3(1024), 0, stream, (const bf16_t*)q, (const bf16_t*)k,
                (const bf16_t*)v, (const bf16_t*)gk, (bf16_t*)o, h0, ht, H, T, nseg, Tseg, sq, sk, sv, sg, so, scale, 1.0f);
    return check_launch("lina_gla_chunk_fwd(pipe)");
}

}  // namespace lina

#ifdef LINA_K2_PROF
extern "C" int lina_k2p_prof_read(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(lina_k2p_prof), sizeof(unsigned long long) * (256 + 1024));
}
#endif
