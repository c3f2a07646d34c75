// gla_chunk_reg.hip -- K2r (bf16, Dk = Dv = 256): the chunk-wise GLA forward with the NEXT chunk's q,k,g,v prefetched into
// REGISTERS by plain global loads and the operand tiles double-buffered: one barrier per 32-token chunk.
// One workgroup per (b,h), the 256 x 256 fp32 state resident in MFMA accumulators for the whole sequence (as in
// gla_chunk_full.hip, whose phase A / phase B arithmetic and tile layouts this kernel keeps); HBM traffic is exactly the
// algorithmic q,k,g,v in + o out.
//
// Replaces fla.ops.gla.chunk_gla / fused_chunk_gla (reference model/gla.py:193,195) for the L169 head shape.
//
// Why (DESIGN_HISTORY.md R4.1): a global -> LDS DMA instruction does not retire from issue until the memory system has accepted
// it, so in gla_chunk_full.hip the four loader waves are blocked for the whole transfer of a chunk (~2200 clocks) on the critical
// path, and the raw tiles they fill (66 KB) leave no room to double-buffer anything at C = 32.  But the phase-A inputs of a thread
// are exactly its own 2 rows x 4 channels of q, k, g, v = 8 loads of 8 bytes = 16 registers: a plain global load is
// fire-and-forget, needs no LDS, and its data is waited for only where phase A uses it.  Without raw tiles the operand tiles fit
// TWICE (141 KB), so
//
//   iteration v (ONE barrier per 32 tokens; every wave the same program):
//     fin     o(v-1) += v^T mask(A)(v-1) (needs mask(A) from iteration v-1), store o(v-1)
//     B(v)    MFMAs from ops[v&1]:  mask(A)(v) (waves 0-3), o^T = S'^T q^^T, S' += k^^T v        (raw(v+1) lands underneath)
//     A(v+1)  gate scan + scaled operands of the next chunk from the raw REGISTERS -> ops[(v+1)&1]
//     load    raw(v+2) -> registers (8 loads per thread), no wait
//   A chunk whose in-chunk decay exceeds e^-60 is cut at the first such row ("virtual chunks": the rest of the raw chunk is the
//   next iteration's chunk; rare, re-loads the raw rows).  Flags carry generation tags (no reset, no race two chunks later).
//
// Same formulation as gla_chunk_full.hip: UN-normalised state S' with S = diag(e^R) S', q^ = q e^{b+R}, k^ = k e^{-(b+R)},
// o = scale (q^ S' + mask(q^ k^^T) v), S' += k^^T v, R += b_last, rows rescaled when R < -20.
#include <type_traits>
#include <lina_dev.h>
#include "lina_common.h"

#ifdef LINA_K2_PROF
// tools-only build: per-phase shader-clock totals of workgroup 0, [wave][slot] + per-workgroup totals
__device__ unsigned long long lina_k2r_prof[16 * 16 + 1024];
#define K2R_PROF(i) do { const unsigned long long now_ = clock64(); pacc[i] += now_ - plast; plast = now_; } while (0)
#else
#define K2R_PROF(i) do { } while (0)
#endif
#ifndef LINA_K2R_ORDER
#define LINA_K2R_ORDER 0       // 0: every wave runs B then A; 1: waves 8..15 run A then B (their raw loads are waited for first)
#endif

namespace lina {

constexpr int kRegC = 32;
constexpr float kRegMaxDecay = 60.0f;
constexpr float kRegRenorm = 20.0f;

namespace reg_detail {
__device__ __forceinline__ void unpack4(const uint2 u, float (&f)[4]) {
    f[0] = bf2f((bf16_t)(u.x & 0xffff)); f[1] = bf2f((bf16_t)(u.x >> 16));
    f[2] = bf2f((bf16_t)(u.y & 0xffff)); f[3] = bf2f((bf16_t)(u.y >> 16));
}
__device__ __forceinline__ bf16x8 frag16(const bf16_t* p) { return as_bf16x8(*reinterpret_cast<const uint4*>(p)); }
__device__ __forceinline__ bf16x8 frag8x2(const bf16_t* p_lo, const bf16_t* p_hi) {
    return as_bf16x8(*reinterpret_cast<const uint2*>(p_lo), *reinterpret_cast<const uint2*>(p_hi));
}
}  // namespace reg_detail

// blockIdx.x = (b*H + h) * nseg + seg handles tokens [seg*Tseg, min(T_total, (seg+1)*Tseg)); h0 / ht are indexed by blockIdx.x
// (nseg = 1: one workgroup per head over the whole sequence; nseg > 1: pass 2 of the segment-parallel forward).
__global__ __launch_bounds__(1024) void gla_chunk_reg_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    const bf16_t* __restrict__ gk, bf16_t* __restrict__ o, const float* h0, float* ht, int H, int T_total, int nseg,
    int Tseg, lina_bht_strides sq, lina_bht_strides sk, lina_bht_strides sv, lina_bht_strides sg, lina_bht_strides so,
    float scale, float h0_scale) {
    using namespace reg_detail;
    constexpr int DK = 256, C = kRegC;
    // q^ / k^ row-major tiles as in gla_chunk_full.hip: 544-byte rows, channels of each group of 32 in the order
    // [0-3,16-19,4-7,20-23,...] (what the state tiles' C/D layout gives the k-slots), 16-byte piece index XOR (row>>2)&3:
    // every operand read is one conflict-free ds_read_b128.
    constexpr int SQ = DK + 16;
    constexpr int OPQ = 2 * C * SQ;                 // elements of one {q^ | k^} buffer
    // k^^T | v^T: [512 rows = channels | columns][32 tokens], UNPADDED 64-byte rows; the 16-byte piece p (tokens 8p..8p+7) of row r
    // sits at position p ^ f(r), f = (-(r>>2)) & 3 over the row's index inside its 16-row tile: with the real ds_read_b128 lane
    // groups ({0-3,12-15,20-27}, ...) the 16 rows x 4 pieces of a fragment read hit 64 distinct banks (the 96-byte padded rows of
    // gla_chunk_full.hip would not fit twice).
    constexpr int ST = C;
    constexpr int OPT = 2 * DK * ST;                // elements of one {k^^T | v^T} buffer
    __shared__ __attribute__((aligned(16))) bf16_t s_qk[2 * OPQ];
    __shared__ __attribute__((aligned(16))) bf16_t s_T[2 * OPT];
    __shared__ __attribute__((aligned(16))) bf16_t s_A[2 * 2 * 64 * 8];   // [chunk parity][nt][lane][8]: mask(A) as ready-made operands
    __shared__ __attribute__((aligned(16))) float s_Rn[2 * DK];           // R (log2 units) AFTER virtual chunk v, [v&1][channel]
    // {cut needed, renormalise} of virtual chunk v at [2(v&1)], [2(v&1)+1], valid iff == v + 1
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
    const bool a_first = LINA_K2R_ORDER == 1 && ((w >> 3) & 1);

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
    const unsigned stq = (unsigned)sq.t, stk = (unsigned)sk.t, stg = (unsigned)sg.t, stv = (unsigned)sv.t;   // < 2^20 (launcher)

    // ---- phase A thread map (gla_chunk_full.hip's): wave w <-> channels [16w, 16w+16); lane = (channel quad c4 = lane>>4, row
    // pair rp = lane&15): the gate scan over the 16 row pairs is four fused DPP adds per value inside one 16-lane row; the thread
    // owns two ADJACENT tokens and writes k^^T / v^T as 4-byte pieces.
    // The thread's raw inputs of one chunk: rows 2rp, 2rp+1 x channels 16w + 4c4 .. +3 of q, k, g, v -- 8 loads of 8 bytes.  Rows past
    // the end of the sequence re-read row T-1 (always mapped); phase A masks them.
    struct Raw { uint2 q[2], k[2], g[2], v[2]; };
    auto load_raw = [&](Raw& r, int j) {
        // (uniform 64-bit base) + ONE 32-bit byte offset per lane: the SGPR-base addressing form, no 64-bit per-lane arithmetic
#if defined(LINA_K2R_PROBE_LINES)
        // TIMING PROBE ONLY (wrong results): the addresses a line-friendly ownership (64 channels x 8 rows per wave) would load
        const unsigned col = 2u * (64u * (unsigned)(w & 3) + 4u * (unsigned)(lane & 15));
#else
        const unsigned col = 2u * (16u * (unsigned)w + 4u * (unsigned)(lane >> 4));          // byte offset of the channel quad
#endif
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
#if defined(LINA_K2R_PROBE_LINES)
            const unsigned t = (unsigned)min(C * j + 8 * (w >> 2) + 2 * (lane >> 4) + rr, T - 1);
#else
            const unsigned t = (unsigned)min(C * j + 2 * (lane & 15) + rr, T - 1);
#endif
            // (g and v first: the scan needs g first, v^T is written before the scan -- a wave's loads return in order)
            r.g[rr] = ld_nt8(reinterpret_cast<const char*>(gg) + (2u * t * stg + col));
            r.v[rr] = ld_nt8(reinterpret_cast<const char*>(gv) + (2u * t * stv + col));
            r.q[rr] = ld_nt8(reinterpret_cast<const char*>(gq) + (2u * t * stq + col));
            r.k[rr] = ld_nt8(reinterpret_cast<const char*>(gkk) + (2u * t * stk + col));
        }
    };
    // inclusive gate cumsum of this thread's 2 rows x 4 channels over the rows [lo, .] (rows outside [lo, end) count as 0); true if
    // the in-chunk decay at the thread's second row is too large for one chunk (monotone in the row)
    auto gate_scan = [&](const Raw& r, float (&bc)[2][4], int lo, int end) -> bool {
        const int rp = lane & 15;
        float g0[4], g1[4];
        unpack4(r.g[0], g0);
        unpack4(r.g[1], g1);
        const bool in0 = 2 * rp >= lo && 2 * rp < end, in1 = 2 * rp + 1 >= lo && 2 * rp + 1 < end;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            g0[c] = in0 ? vmax_raw(g0[c], -kRegMaxDecay) : 0.0f;
            g1[c] = in1 ? vmax_raw(g1[c], -kRegMaxDecay) : 0.0f;
            bc[1][c] = g0[c] + g1[c];                         // the row pair's sum
        }
        row_scan4(bc[1][0], bc[1][1], bc[1][2], bc[1][3]);
        bool viol = false;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            bc[0][c] = bc[1][c] - g1[c];                      // the pair's first row
            viol |= (-bc[1][c] > kRegMaxDecay);
        }
        return viol;
    };
    // v enters the products unscaled: its transposed pieces need nothing from the scan (written first, registers freed)
    auto write_vT = [&](const Raw& r, int pn, int lo, int hi) {
        const int rp = lane & 15, c4 = lane >> 4, ch0 = 16 * w + 4 * c4;
        uint2 vv[2];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const bool valid = 2 * rp + rr >= lo && 2 * rp + rr < hi;
            vv[rr].x = valid ? r.v[rr].x : 0u; vv[rr].y = valid ? r.v[rr].y : 0u;
        }
        // (channel row ch0+i, tokens 2rp, 2rp+1) = one 4-byte word: piece (rp>>2) ^ f, word rp&3;  (ch0+i) & 15 = 4 c4 + i -> f = (-c4) & 3
        bf16_t* const tp = &s_T[pn * OPT + (DK + ch0) * ST + 8 * ((rp >> 2) ^ ((4 - c4) & 3)) + 2 * (rp & 3)];
        *reinterpret_cast<unsigned*>(tp) = byte_perm(vv[1].x, vv[0].x, 0x05040100u);
        *reinterpret_cast<unsigned*>(tp + ST) = byte_perm(vv[1].x, vv[0].x, 0x07060302u);
        *reinterpret_cast<unsigned*>(tp + 2 * ST) = byte_perm(vv[1].y, vv[0].y, 0x05040100u);
        *reinterpret_cast<unsigned*>(tp + 3 * ST) = byte_perm(vv[1].y, vv[0].y, 0x07060302u);
    };
    // q^, k^ (row-major) and k^^T of a virtual chunk = rows [lo, hi) -> ops[pn]; rows outside are zeroed.  The owner of row hi-1
    // publishes R after the chunk (s_Rn[pn]) and the renormalisation flag.  q^ carries NO 1/sqrt(Dk) (applied to o).
    auto write_tiles = [&](const Raw& r, const float (&bc)[2][4], int pn, int lo, int hi, bool zero_r, int rpar, int gen) {
        constexpr float kLog2e = 1.4426950408889634f;
        const int rp = lane & 15, c4 = lane >> 4, ch0 = 16 * w + 4 * c4;
        float4 R4 = *reinterpret_cast<const float4*>(&s_Rn[rpar * DK + ch0]);
        if (zero_r) R4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const float Rc[4] = {R4.x, R4.y, R4.z, R4.w};
        // column of this thread's channel quad in the q^ / k^ tiles: group w/2, piece c4 ^ ((row>>2)&3) = c4 ^ ((rp>>1)&3), half w&1
        bf16_t* const qkp = &s_qk[pn * OPQ + 2 * rp * SQ + 32 * (w >> 1) + 8 * (c4 ^ ((rp >> 1) & 3)) + 4 * (w & 1)];
        uint2 kk[2];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int row = 2 * rp + rr;
            const bool valid = row >= lo && row < hi;
            float x[4], e[4], ri[4], f[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                x[c] = __builtin_fmaf(bc[rr][c], kLog2e, Rc[c]);   // (b + R) log2 e: |b| <= 60 in a legal chunk, |R| <= kRegRenorm + 60
                e[c] = fast_exp2(x[c]);
                ri[c] = fast_rcp(e[c]);
            }
            uint2 pq;
            unpack4(r.q[rr], f);
            pq.x = pack_bf16x2(f[0] * e[0], f[1] * e[1]);      // rows outside [lo, hi): zeroed as packed words
            pq.y = pack_bf16x2(f[2] * e[2], f[3] * e[3]);
            pq.x = valid ? pq.x : 0u; pq.y = valid ? pq.y : 0u;
            *reinterpret_cast<uint2*>(qkp + rr * SQ) = pq;
            unpack4(r.k[rr], f);
            kk[rr].x = pack_bf16x2(f[0] * ri[0], f[1] * ri[1]);
            kk[rr].y = pack_bf16x2(f[2] * ri[2], f[3] * ri[3]);
            kk[rr].x = valid ? kk[rr].x : 0u; kk[rr].y = valid ? kk[rr].y : 0u;
            *reinterpret_cast<uint2*>(qkp + C * SQ + rr * SQ) = kk[rr];
            if (row == hi - 1) {                               // owner of the chunk's last row: R after the chunk
                bool need = false;
#pragma unroll
                for (int c = 0; c < 4; ++c) need |= x[c] < -kRegRenorm * kLog2e;
                *reinterpret_cast<float4*>(&s_Rn[pn * DK + ch0]) = make_float4(x[0], x[1], x[2], x[3]);
                if (need) s_flag[2 * pn + 1] = gen;
            }
        }
        bf16_t* const tp = &s_T[pn * OPT + ch0 * ST + 8 * ((rp >> 2) ^ ((4 - c4) & 3)) + 2 * (rp & 3)];
        *reinterpret_cast<unsigned*>(tp) = byte_perm(kk[1].x, kk[0].x, 0x05040100u);
        *reinterpret_cast<unsigned*>(tp + ST) = byte_perm(kk[1].x, kk[0].x, 0x07060302u);
        *reinterpret_cast<unsigned*>(tp + 2 * ST) = byte_perm(kk[1].y, kk[0].y, 0x05040100u);
        *reinterpret_cast<unsigned*>(tp + 3 * ST) = byte_perm(kk[1].y, kk[0].y, 0x07060302u);
    };
    // phase A of virtual chunk (raw chunk j in r, first row lo), optimistic (all rows up to the raw chunk's end): sets the cut flag
    // (generation gen) when the chunk must be cut
    auto phase_a = [&](const Raw& r, int pn, int j, int lo, bool zero_r, int rpar, int gen) {
        const int end = min(C, T - C * j);
        write_vT(r, pn, lo, end);
        sched_fence();
        float bc[2][4];
        if (gate_scan(r, bc, lo, end)) s_flag[2 * pn] = gen;
        write_tiles(r, bc, pn, lo, end, zero_r, rpar, gen);
    };

    if (threadIdx.x < 4) s_flag[threadIdx.x] = 0;
    if (threadIdx.x == 4) s_cut = 0;
    Raw raw;                                                   // the raw registers: chunk `jr`
    load_raw(raw, 0);
    int jr = 0;
    __syncthreads();                                           // flags initialised
    phase_a(raw, 0, 0, 0, true, 0, 1);
    if (NJ > 1) { load_raw(raw, 1); jr = 1; }
    lds_barrier();                                             // ops[0] complete (the loads of chunk 1 stay in flight)

    int vj = 0, vlo = 0;                                       // current virtual chunk: raw chunk, first row
    int ptok = 0, plo = 0, phi = 0;                            // previous virtual chunk: first token of its raw chunk, rows [plo, phi)
    f32x4 accp[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};   // o^T of the previous chunk before its intra-chunk term
    bf16x8 vbp = as_bf16x8(make_uint4(0u, 0u, 0u, 0u));        // v^T fragment of the previous chunk (same token order as mask(A)'s C/D rows)
    // o^T += v^T mask(A)^T of the previous chunk, then o straight from the accumulators: a lane holds 4 consecutive columns
    // (16w + 4lg ..) of token 16 nt + li = one 8-byte store; the 16 waves' 32-byte pieces of a 512-byte row meet in L2.
    auto finish_prev = [&](int ppar) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            accp[nt] = mfma_bf16_16x16x32(vbp, frag16(&s_A[ppar * 1024 + (nt * 64 + lane) * 8]), accp[nt]);
            const int row = 16 * nt + li;
            uint2 po;
            po.x = pack_bf16x2(accp[nt][0] * scale, accp[nt][1] * scale);
            po.y = pack_bf16x2(accp[nt][2] * scale, accp[nt][3] * scale);
#if defined(LINA_K2R_PROBE_NOSTORE)
            if (row >= plo && row < phi && po.x == 0x12345678u) {   // TIMING PROBE ONLY: (almost) never stores
#else
            if (row >= plo && row < phi) {
#endif
                const unsigned boff = 2u * ((unsigned)(ptok + row) * (unsigned)so.t + 16u * (unsigned)w + 4u * (unsigned)lg);
                *reinterpret_cast<uint2*>(reinterpret_cast<char*>(ob) + boff) = po;
            }
        }
    };

#ifdef LINA_K2_PROF
    unsigned long long pacc[8] = {}, plast = clock64();
    const unsigned long long pstart = plast;
#endif
    int last_par = 0;
    bool last_renorm = false;
    for (int vi = 0;; ++vi) {
        // nothing per-lane is carried across iterations except the accumulators and the raw registers: the lane index comes from v_mbcnt
        lane = lane_id();
        opaque(lane);
        li = lane & 15; lg = lane >> 4;
        const int par = vi & 1;
        const int end = min(C, T - C * vj);
        int hi = end;
        if (s_flag[2 * par] == vi + 1) {
            // ---- rare: the decay inside this chunk exceeds e^-60 -> cut it at the first such row; the rest of the raw chunk
            //      becomes the next virtual chunk.  Workgroup-uniform branch.  The raw registers hold a LATER chunk: fetch this
            //      one again (and the one they held afterwards, below).
            load_raw(raw, vj);
            float bc[2][4];
            gate_scan(raw, bc, vlo, end);
            int nc = C;
#pragma unroll
            for (int rr = 1; rr >= 0; --rr) {
                bool bad = false;
#pragma unroll
                for (int c = 0; c < 4; ++c) bad |= (-bc[rr][c] > kRegMaxDecay);
                if (bad) nc = 2 * (lane & 15) + rr;
            }
            if (nc < C) lds_atomic_max(&s_cut, C - nc);        // first bad row of the workgroup = C - max
            __syncthreads();
            hi = max(min(end, C - s_cut), vlo + 1);            // row vlo itself is never bad (one clamped gate)
            __syncthreads();                                   // everyone has read s_cut; the optimistic tiles are dead
            if (threadIdx.x == 0) { int z = 0; opaque(z); s_cut = z; }
            const bool zr = vi == 0 || s_flag[2 * (par ^ 1) + 1] == vi;   // R this chunk started from (intact until phase A below)
            write_vT(raw, par, vlo, hi);
            write_tiles(raw, bc, par, vlo, hi, zr, par ^ 1, vi + 1);
            jr = -1;                                           // the registers no longer hold what phase A below needs
            __syncthreads();
        }
        const bool renorm = s_flag[2 * par + 1] == vi + 1;     // workgroup-uniform
        const bool split = hi < end;
        const int nj = split ? vj : vj + 1, nlo = split ? hi : 0;
        const bool more = nj < NJ;
        if (more && jr != nj) { load_raw(raw, nj); jr = nj; }  // rare (after a cut): the next virtual chunk's rows, waited for in phase A
        K2R_PROF(0);
        if (vi > 0) finish_prev(par ^ 1);
        K2R_PROF(1);

        if (a_first && more) phase_a(raw, par ^ 1, nj, nlo, renorm, par, vi + 2);
        K2R_PROF(2);

        // ---------------- phase B of virtual chunk vi ----------------
        if (w < 4) {
            // (2) A^T[s][t] = k^_s . q^_t, computed ONCE per workgroup by waves 0..3 (one per SIMD): wave w takes the whole 16 x 16 tile
            //     (mt = w&1: s block, nt = w>>1: t block), 8 MFMAs over the 256 channels; each lane masks (s <= t) and stores its 4
            //     values as one 8-byte piece of the ready-made operand of the intra-chunk MFMA
            wave_priority<2>();
            const int mt = w & 1, nt = w >> 1;
            const int pc = 8 * (lg ^ ((li >> 2) & 3));
            const bf16_t* kp = &s_qk[par * OPQ + C * SQ + (16 * mt + li) * SQ + pc];
            const bf16_t* qp = &s_qk[par * OPQ + (16 * nt + li) * SQ + pc];
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
            *reinterpret_cast<uint2*>(&s_A[par * 1024 + (nt * 64 + lane) * 8 + 4 * mt]) = pa;
            wave_priority<0>();
        }
        K2R_PROF(3);
        // k^^T / v^T fragments: row r of a 16-row tile, piece p (tokens 8p..8p+7) at position p ^ ((-(r>>2)) & 3)
        const int fsw = (4 - (li >> 2)) & 3;
        const bf16_t* ktp = &s_T[par * OPT + li * ST];
        auto ld_kt = [&](int p) -> bf16x8 { return frag16(ktp + 16 * p * ST + 8 * (lg ^ fsw)); };
        const bf16_t* vrow = &s_T[par * OPT + (DK + 16 * w + li) * ST];
        // (1) o^T = S'_old^T-tiles . q^^T: one K = 32 MFMA per pair of 16-row state tiles and token tile (converted to bf16 in registers)
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#ifndef LINA_K2R_TA
#define LINA_K2R_TA 3
#endif
#ifndef LINA_K2R_QA
#define LINA_K2R_QA 2
#endif
        // operand rings: QA q^ tile pairs / TA k^^T tiles requested ahead of their MFMAs.  Shallow on purpose: the raw registers of the
        // next chunk (16) are live through this phase, and a spill here would be a scratch round trip behind the loads in flight
        constexpr int TA = LINA_K2R_TA, QA = LINA_K2R_QA;
        bf16x8 tf[4];
        bf16x8 vb2;
        {
            const bf16_t* qp = &s_qk[par * OPQ + li * SQ + 8 * (lg ^ ((li >> 2) & 3))];
            bf16x8 qf[4][2];
#pragma unroll
            for (int pp = 0; pp < QA; ++pp)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) qf[pp][nt] = frag16(qp + 16 * nt * SQ + 32 * pp);
#pragma unroll
            for (int pp = 0; pp < 8; ++pp) {
                if (pp + QA < 8) {
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) qf[(pp + QA) & 3][nt] = frag16(qp + 16 * nt * SQ + 32 * (pp + QA));
                } else if (pp == 8 - QA) {                     // the ring's free slots take step (4)'s first operands
                    vb2 = frag16(vrow + 8 * (lg ^ fsw));
                    tf[0] = ld_kt(0);
                } else if (pp == 9 - QA) {
                    tf[1] = ld_kt(1);
                    if constexpr (TA > 2) tf[2] = ld_kt(2);
                }
                sched_fence();
                bf16x8 bb;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    bb[r] = (short)f2bf(S[2 * pp][r]);
                    bb[4 + r] = (short)f2bf(S[2 * pp + 1][r]);
                }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[nt] = mfma_bf16_16x16x32(bb, qf[pp & 3][nt], acc[nt]);
                sched_fence();
            }
            if constexpr (QA == 1) {                           // (what a one-deep ring had no free slot for)
                tf[1] = ld_kt(1);
                if constexpr (TA > 2) tf[2] = ld_kt(2);
            }
        }
        K2R_PROF(4);
        // (4) S' += k^^T v
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            if (p + TA < 16) tf[(p + TA) & 3] = ld_kt(p + TA);
            sched_fence();
            S[p] = mfma_bf16_16x16x32(tf[p & 3], vb2, S[p]);
            sched_fence();
        }
        if (renorm) {                                          // rare: S' <- e^{R} S' (R = s_Rn[par], the value after this chunk)
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const float4 r4 = *reinterpret_cast<const float4*>(&s_Rn[par * DK + 16 * p + 4 * lg]);
                S[p][0] *= fast_exp2(r4.x); S[p][1] *= fast_exp2(r4.y); S[p][2] *= fast_exp2(r4.z); S[p][3] *= fast_exp2(r4.w);
            }
        }
        // v fragment of the intra-chunk term (tokens 4lg..4lg+3 and 16+4lg..+3: the order of mask(A)'s C/D rows), read before the
        // tiles die at the barrier: token quad u = lg (first half) / 4 + lg (second half) -> piece u>>1, half u&1
        {
            const bf16_t* p_lo = vrow + 8 * ((lg >> 1) ^ fsw) + 4 * (lg & 1);
            const bf16_t* p_hi = vrow + 8 * ((2 + (lg >> 1)) ^ fsw) + 4 * (lg & 1);
            vbp = frag8x2(p_lo, p_hi);
        }
        accp[0] = acc[0]; accp[1] = acc[1];
        ptok = C * vj; plo = vlo; phi = hi;
        K2R_PROF(5);

        if (!a_first && more) phase_a(raw, par ^ 1, nj, nlo, renorm, par, vi + 2);
        K2R_PROF(2);
        // the raw registers are free: request the rows of the chunk after the next one (optimistic: the next chunk is not cut)
        if (more && nj + 1 < NJ) { load_raw(raw, nj + 1); jr = nj + 1; }
        K2R_PROF(6);
        lds_barrier();                                         // ops[par^1], mask(A)(vi), R, flags visible; the loads stay in flight
        K2R_PROF(7);
        if (!more) { last_par = par; last_renorm = renorm; break; }
        vj = nj; vlo = nlo;
    }
    lane = lane_id(); opaque(lane); li = lane & 15; lg = lane >> 4;
    finish_prev(last_par);                                     // the last chunk's intra-chunk term and its store
    if (ht) {                                                  // S = diag(e^{R}) S'
        float* hp = ht + ((int64_t)slot * DK + 4 * lg) * DK + 16 * w + li;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            float4 r4 = *reinterpret_cast<const float4*>(&s_Rn[last_par * DK + 16 * p + 4 * lg]);
            if (last_renorm) r4 = make_float4(0.f, 0.f, 0.f, 0.f);
            hp[(16 * p + 0) * DK] = S[p][0] * fast_exp2(r4.x);
            hp[(16 * p + 1) * DK] = S[p][1] * fast_exp2(r4.y);
            hp[(16 * p + 2) * DK] = S[p][2] * fast_exp2(r4.z);
            hp[(16 * p + 3) * DK] = S[p][3] * fast_exp2(r4.w);
        }
    }
#ifdef LINA_K2_PROF
    if (blockIdx.x == 0 && lane_id() == 0)
        for (int i = 0; i < 8; ++i) lina_k2r_prof[w * 16 + i] = pacc[i];
    if (blockIdx.x < 1024 && w == 0 && lane_id() == 0) lina_k2r_prof[256 + blockIdx.x] = clock64() - pstart;
#endif
}

// Launcher used by launch_chunk_full (gla_chunk_full.hip) for one head per workgroup (Dk = Dv = 256).
int launch_chunk_reg(const void* q, const void* k, const void* v, const void* gk, void* o, const float* h0, float* ht,
                     int slots, int H, int T, int nseg, int Tseg, lina_bht_strides sq, lina_bht_strides sk,
                     lina_bht_strides sv, lina_bht_strides sg, lina_bht_strides so, float scale, lina_stream_t stream) {
    LINA_LAUNCH(gla_chunk_reg_kernel, dim3((unsigned)slots), dim3(1024), 0, stream, (const bf16_t*)q, (const bf16_t*)k,
                (const bf16_t*)v, (const bf16_t*)gk, (bf16_t*)o, h0, ht, H, T, nseg, Tseg, sq, sk, sv, sg, so, scale, 1.0f);
    return check_launch("lina_gla_chunk_fwd(reg)");
}

}  // namespace lina

#ifdef LINA_K2_PROF
extern "C" int lina_k2r_prof_read(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(lina_k2r_prof), sizeof(unsigned long long) * (256 + 1024));
}
#endif
