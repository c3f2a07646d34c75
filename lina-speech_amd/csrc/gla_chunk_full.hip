// gla_chunk_full.hip -- K2 (bf16, Dk = Dv = 256): chunk-wise GLA forward, ONE workgroup per (b,h),
// the whole 256 x 256 fp32 state resident in MFMA accumulators (16 waves x 16 tiles of 16 x 16 =
// 64 accumulator VGPRs per lane) for the entire sequence.  HBM traffic is exactly the algorithmic
// q,k,g,v in + o out (e*(3Dk+2Dv) per (row, head, token)); nothing else touches HBM.
//
// Replaces fla.ops.gla.chunk_gla / fused_chunk_gla (reference model/gla.py:193,195) for the L169
// head shape; other shapes / fp32 use gla_chunk.hip.  Same formulation as gla_chunk.hip:
//     q~ = scale q e^{b}, k~ = k e^{-b}, o = q~ S + mask(q~ k~^T) v,  S <- e^{b_last} (S + k~^T v)
// with chunks of C = 32 tokens, cut adaptively when the in-chunk decay would exceed e^-60.
//
// 1024 threads = 16 waves = 4 per SIMD at <= 128 VGPRs (64 of them the state).  Per chunk, two barriers:
//   A  the chunk's raw q,k,g,v tiles are already in LDS (asynchronous global->LDS DMA issued one chunk ahead by the last
//      four waves, no staging registers).  Wave w <-> channels [16w, 16w+16); lane = (channel quad, row pair): the gate
//      scan over the 16 row pairs is four fused DPP adds per value inside one 16-lane row; the thread scales its two
//      tokens x four channels (q~ = q e^{b+R}, k~ = k e^{-b-R}; R = log-decay carried by the UN-normalised state) and
//      writes q~ / k~ row-major (channel-permuted, XOR-swizzled: every later operand read is one conflict-free
//      ds_read_b128) and k~^T / v^T (the operands whose K dimension is the token axis) as 4-byte pieces.  The previous
//      chunk's o is stored at the end of this phase.                                             -- barrier (2) --
//   B  wave w owns state columns [16w, 16w+16), tiles p = rows [16p,16p+16) in C/D layout (col = lane&15,
//      row = 4*(lane>>4)+reg):  (2) A^T = k~ . q~^T (32 x 32 as 2x2 tiles) once per workgroup (waves 0..3: one whole
//      tile each, stored masked straight into the A-operand layout of step (3)); (1) o^T = S'^T-tiles . q~^T with the
//      state tiles consumed DIRECTLY as the A operand (converted to bf16 in registers); (4) S' += k~^T . v.  All LDS
//      operand reads are issued 3-5 MFMAs ahead by hand (fragment rings + sched_fence()).  36 x
//      v_mfma_f32_16x16x32_bf16 per wave per chunk.                                              -- barrier (3) --
//      (3) o^T += v^T . mask(A)^T after the barrier (mask(A) has its own buffer), o carried in registers to the next
//      phase A's end.  1/sqrt(Dk) is applied to o.  History and measurements: DESIGN.md 4.2.
#include <type_traits>
#ifndef LINA_DMA_NT
#define LINA_DMA_NT 1   // the q,k,g,v prefetch is read once: non-temporal DMA (0.582 -> 0.572 ms at B=64,H=4,T=4096, round 4).  Per
#endif                  // instantiation since round 6 (template argument NTD, default = this): the segment-parallel FORWARD reads k, g, v
                        // twice within microseconds (state-only pass, then the full pass) and runs 0.150 instead of 0.162 ms at b = 8 with
                        // the plain policy; the backward's sweeps and the one-workgroup-per-head form keep nt (profiles/r06_k2_dma_nt_ab.txt)
#include <lina_dev.h>
#include "lina_common.h"

#ifdef LINA_K2_PROF
// tools-only build (tools/k2_prof.sh): per-phase shader-clock totals of workgroup 0, [wave][slot]; NOT part of the product library
__device__ unsigned long long lina_k2_prof[16 * 16 + 3 * 1024];   // + per workgroup: total, wait_vmem, bar(3) of wave 0
#define K2_PROF(i) do { const unsigned long long now_ = clock64(); pacc[i] += now_ - plast; plast = now_; } while (0)
#else
#define K2_PROF(i) do { } while (0)
#endif



namespace lina {

constexpr int kFullC = 32;
constexpr float kFullMaxDecay = 60.0f;
constexpr float kRenorm = 20.0f;        // renormalise the state when a channel's accumulated log-decay passes -kRenorm

__device__ __forceinline__ void unpack4(const uint2 u, float (&f)[4]) {
    f[0] = bf2f((bf16_t)(u.x & 0xffff)); f[1] = bf2f((bf16_t)(u.x >> 16));
    f[2] = bf2f((bf16_t)(u.y & 0xffff)); f[3] = bf2f((bf16_t)(u.y >> 16));
}
__device__ __forceinline__ bf16x8 frag16(const bf16_t* p) {   // one 16-byte LDS read
    return as_bf16x8(*reinterpret_cast<const uint4*>(p));
}
__device__ __forceinline__ bf16x8 frag8x2(const bf16_t* p_lo, const bf16_t* p_hi) {   // two 8-byte LDS reads
    return as_bf16x8(*reinterpret_cast<const uint2*>(p_lo), *reinterpret_cast<const uint2*>(p_hi));
}

// Sequence segments: blockIdx.x = (b*H + h) * nseg + seg handles tokens [seg*Tseg, min(T, (seg+1)*Tseg)) with
// h0 / ht / dec_out indexed by blockIdx.x (nseg = 1: the plain one-workgroup-per-head form).  STATE_ONLY: no output,
// only the segment's state transition -- final state from the given start (zero if h0 == NULL) and the product of
// the chunk decays dec_out[blockIdx.x][Dk] -- used by the two-pass segment-parallel forward below.
// G heads of dimension D = 256 / G side by side in one workgroup (G = 1: the L169 head, 256 x 256; G = 2: 128 x 128; G = 4:
// 64 x 64): in a [B,T,H,D] tensor the G heads of a group are one contiguous 512-byte row per token, so the DMA, phase A and
// every tile are unchanged; the state is block-diagonal -- wave w belongs to head w / (16/G) and keeps only that head's
// 16/G row tiles, the token contractions run over its head's channels only, mask(A) exists once per head.
//
// The same kernel body is the BACKWARD's three sweeps (K2b, lina_gla_chunk_bwd_full below):
//   REV     the segment's tokens are visited last to first (row r of a chunk = token T-1-t0-r) and the gate of reversed
//           row r is the gate of the token AFTER it in time (the decay between two tokens belongs to the later one), zero
//           past the end of the sequence.  With (q,k,v) := (k,q,do) the forward body then yields dv and the state dS.
//   MODE 1  "value-gated": the state is held TRANSPOSED (S'^T: rows = the contraction index of the sweep, columns = the
//           gated channels c), out[t][c] = scale e^{b_t+R}[c] ( X_t . S'^T + mask(X Y^T) Z^ )[c],  S'^T += Y^T Z^,
//           Z^ = Z e^{-b-R}: X, Y enter the MFMAs raw, Z is gated like k, and the gate factor is applied to the OUTPUT.
//           (X,Y,Z) = (do,v,k) forward gives dq, (v,do,q) with REV gives dk.  The token columns of the products are
//           permuted (column li of tile nt = token 2 li + nt) so that an output lane holds exactly the two tokens x four
//           channels whose factors e^{b+R} it computed itself in phase A: nothing is exchanged.
//   DG      (sweep K) d = z (.) aux2 - aux (.) out, running sum of d over the visited tokens (+ carry[slot]) -> dg: with
//           z = q (this sweep's own Z rows: in MODE 1 a phase-A thread reads exactly the (token, channel) values whose output
//           it will hold), aux2 = dq (sweep Q's output) and aux = k that is dg = reverse-cumsum(q dq - k dk), formed while dk
//           is still fp32 in registers.
//
// NCB = 2 (round 6: Dv = 2 Dk = 512, the reference's default expand_v = 2 at the L169 key width -- model/gla.py:52,267): the
// head's value columns are split over TWO workgroups of ONE launch (the recurrence is independent per value column; the state
// of a 256 x 512 head does not fit one CU's accumulators).  The two halves of a head get block ids i and i + 8: the dispatcher
// deals consecutive ids round-robin over the 8 XCDs, so both land on the SAME XCD in the same dispatch round, run in step
// and the second reader of a q / k / g row pair finds it in that XCD's L2 (plain cache policy for this form) -- HBM sees
// q, k, g once.  h0 / ht keep their natural [B, H, 256, 512] layout (row stride 512, column offset 256 cb).
template <bool STATE_ONLY, int G, int MODE = 0, bool REV = false, bool DG = false, bool NTD = (LINA_DMA_NT != 0), int NCB = 1>
__global__ __launch_bounds__(1024) void gla_chunk_bf16_h256_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    const bf16_t* __restrict__ gk, bf16_t* __restrict__ o, const float* h0, float* ht, float* dec_out, int H,
    int T_total, int nseg, int Tseg, lina_bht_strides sq, lina_bht_strides sk, lina_bht_strides sv,
    lina_bht_strides sg, lina_bht_strides so, float scale, float h0_scale, const bf16_t* aux, lina_bht_strides saux,
    const bf16_t* aux2, lina_bht_strides saux2, bf16_t* dg, lina_bht_strides sdg, const float* carry) {
    static_assert(MODE == 0 || !STATE_ONLY, "the state-only pass exists in the key-gated form only");
    static_assert(!DG || MODE == 1, "dg is formed by a value-gated sweep");
    static_assert(NCB == 1 || (NCB == 2 && G == 1 && MODE == 0 && !STATE_ONLY && !REV), "column blocks: the plain forward of a 256 x 512 head");
    constexpr int DK = 256, DV = 256, C = kFullC;       // the WORKGROUP's channel / column width: G heads of D each
    constexpr int D = 256 / G;                            // head dimension
    constexpr int NTL = 16 / G;                           // waves per head = state row tiles per wave
    // q~ / k~ row-major tiles, 544-byte rows.  Two twists make EVERY operand read of them one conflict-free ds_read_b128
    // (the two 8-byte reads step (1) needed before were merged by the compiler into ds_read2_b64: half rate, 32 banks,
    // 2-way conflicted -- 4096 LDS cycles per chunk, the largest single cost of the kernel):
    //  * channel order inside each group of 32: [0-3, 16-19, 4-7, 20-23, 8-11, 24-27, 12-15, 28-31], so that 16-byte piece
    //    lg of a group holds exactly the channels the state tiles (C/D layout, rows 4lg+r of tiles 2pp, 2pp+1) give lane
    //    group lg as k-slots 0..7; mask(A) reads both tiles the same way, so its contraction is unaffected;
    //  * the piece index is XOR-ed with (row >> 2) & 3: 16 rows x 4 pieces of one read land on 64 distinct banks, and the
    //    row-strided 8-byte writes of phase A are 2-way instead of 4-way conflicted (tools: bank model in DESIGN.md 4.2).
    constexpr int SQ = DK + 16;
    constexpr int SK = DK + 16;
    constexpr int ST = C + 16;   // bf16 row stride of the transposed tiles (96 B): conflict-free 16-byte fragment reads under the
                                 // real ds_read_b128 lane grouping ({0-3,12-15,20-27}, ...); 80 B was 2-way there
    constexpr int PE = 2 * DK + 8;   // elements per ROW PAIR of a raw tile: one DMA instruction (2 rows, 1 KiB) + 16 B pad, so
                                     // that the 16 row pairs read by one phase-A instruction start in different banks
    // Tiles that phase A addresses together live in ONE object each, so that a thread needs one address register per
    // object and every other tile / row is an immediate offset (the per-chunk address arithmetic was ~30 of phase A's
    // VALU instructions, and phase A is VALU-issue bound).
    __shared__ __attribute__((aligned(16))) bf16_t s_qk[2 * C * SQ];    // q~ | row-major k~ (for mask(A))
    bf16_t* const s_q = s_qk;
    bf16_t* const s_k = s_qk + C * SQ;
    __shared__ __attribute__((aligned(16))) bf16_t s_A[G * 2 * 64 * 8]; // mask(A) as ready-made operands [head][nt][lane][8]
    __shared__ __attribute__((aligned(16))) bf16_t s_T[(DK + DV) * ST];  // k~^T | v^T
    bf16_t* const s_kT = s_T;
    bf16_t* const s_vT = s_T + DK * ST;
    // next chunk's q, k, g, v, filled by DMA (issued through inline assembly and waited for by hand: the compiler does not
    // see these writes, so reads of the object are never held back by them)
    constexpr int RAWT = (C / 2) * PE;
    __shared__ __attribute__((aligned(16))) bf16_t s_raw[4 * RAWT];
    bf16_t* const s_rq = s_raw;
    bf16_t* const s_rk = s_raw + RAWT;
    bf16_t* const s_rg = s_raw + 2 * RAWT;
    bf16_t* const s_rv = s_raw + 3 * RAWT;
    // UN-NORMALISED state: the accumulators hold S' with S = diag(e^{R}) S', R[c] <= 0 the log-decay accumulated since the
    // last renormalisation.  q~ and k~ absorb R (q^ = q~ e^{R}, k^ = k~ e^{-R}), so a chunk needs NO pass over the state
    // for its decay; when some R drops below -kRenorm the state rows are rescaled once (S' <- e^{R} S', R <- 0).
    // R is kept in LOG2 units (R log2 e): the exponent argument is then one fma per value.
    __shared__ __attribute__((aligned(16))) float s_Rs[3 * DK];
    float* const s_R = s_Rs;              // R used by this chunk's phase A
    float* const s_Rn = s_Rs + DK;        // R after this chunk (written by the owners of the last row)
    float* const s_dec = s_Rs + 2 * DK;   // e^{b_last} of this chunk (STATE_ONLY: segment decay product)
    // flags {cut needed, renormalise} per chunk parity (reset one chunk later): adjacent, so that ONE 8-byte read right after
    // barrier (2) fetches both (three dependent LDS round trips -- flag, flag, R -- sat in front of every wave's MFMAs)
    __shared__ __attribute__((aligned(8))) unsigned s_flags[4];
    __shared__ int s_cut;
    __shared__ __attribute__((aligned(16))) float s_carry[DG ? DK : 4];   // DG: running sum of d per channel (wave-private quads)

    int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int w_s = wave_uniform(w);                         // wave index in an SGPR for the whole kernel
    int li = lane & 15, lg = lane >> 4;
    int slot_ = blockIdx.x, cb_ = 0;                         // NCB = 2: (head, column block) from the block id
    if constexpr (NCB == 2) {
        if (gridDim.x & 15) { cb_ = slot_ & 1; slot_ >>= 1; }                                       // heads % 8 != 0: no pairing
        else { cb_ = (slot_ >> 3) & 1; slot_ = ((slot_ >> 4) << 3) | (slot_ & 7); }                  // ids i, i + 8 = one head
    }
    const int slot = slot_;                                  // state slot: (head group) * nseg + segment
    const int cb = wave_uniform(cb_);
    constexpr int HS = D * NCB;                              // row stride of h0 / ht
    const int bh = (slot / nseg) * G, b = bh / H, h = bh % H;   // first head of the group (H % G == 0: one batch row)
    const int hw = w_s / NTL, wl = w_s % NTL;                // this wave's head inside the group, its index inside the head
    const int seg_t = REV ? nseg - 1 - slot % nseg : slot % nseg;   // REV: slot order = visiting order = last segment first
    const int t_begin = seg_t * Tseg;
    const int T = min(Tseg, T_total - t_begin);              // tokens of this segment (>= 1 by construction)
    const bool rev_tail = REV && t_begin + T >= T_total;     // the first visited row has no later token: its gate is 0

    // ---- state: wave w owns columns [16w, 16w+16), tile p = rows [16p, 16p+16) ----
    auto tile_row = [&](int t) { return 16 * t; };                       // first row inside the head
    auto tile_col = [&](int t) { return 16 * wl; };                      // first column inside the head
    auto tile_ch = [&](int t) { return 16 * (NTL * hw + t); };           // first channel of the workgroup
    f32x4 S[NTL];                                            // local tile p <-> rows [16p, 16p+16) of this wave's head
#pragma unroll
    for (int p = 0; p < NTL; ++p) S[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (h0) {                                                // state slot layout [G][D][D] (G = 1: [256][256])
        if constexpr (MODE == 1) {                           // transposed: tile element (row j, column c) = state[c][j]
            const float* hp = h0 + (((int64_t)slot * G + hw) * D + li) * D + 4 * lg;
#pragma unroll
            for (int p = 0; p < NTL; ++p) {
                const float4 t4 = *reinterpret_cast<const float4*>(hp + tile_col(p) * D + tile_row(p));
                S[p] = f32x4{t4.x * h0_scale, t4.y * h0_scale, t4.z * h0_scale, t4.w * h0_scale};
            }
        } else {
            const float* hp = h0 + (((int64_t)slot * G + hw) * D + 4 * lg) * HS + (NCB > 1 ? cb * D : 0) + li;
#pragma unroll
            for (int p = 0; p < NTL; ++p)
#pragma unroll
                for (int r = 0; r < 4; ++r) S[p][r] = hp[(tile_row(p) + r) * HS + tile_col(p)] * h0_scale;
        }
    }

    const bf16_t* gsrc[4] = {q + b * sq.b + h * sq.h + t_begin * sq.t, k + b * sk.b + h * sk.h + t_begin * sk.t,
                             gk + b * sg.b + h * sg.h + t_begin * sg.t,
                             v + b * sv.b + h * sv.h + t_begin * sv.t + (NCB > 1 ? cb * DV : 0)};
    const unsigned gst[4] = {(unsigned)sq.t, (unsigned)sk.t, (unsigned)sg.t, (unsigned)sv.t};   // < 2^31 (launcher)
    bf16_t* ob = STATE_ONLY ? nullptr : o + b * so.b + h * so.h + t_begin * so.t + (NCB > 1 ? cb * DV : 0);
    float4 decp = make_float4(1.f, 1.f, 1.f, 1.f);            // STATE_ONLY, wave 0: product of the chunk decays, channels 4*lane..+3

    // One DMA instruction = one row pair (2 rows x 512 B, 16 B per lane) of one raw tile.  A DMA instruction blocks its wave
    // for 30+ clocks and the 64 of a chunk queue up in the address unit (~1800 clocks): when every wave issued its own four
    // after barrier (2), the waves at the end of that queue started their MFMAs ~1300 clocks late and everybody waited for
    // them at (3).  So ONLY the last four waves (one per SIMD, the lowest issue priority) issue DMAs, 16 each (row pairs
    // 4(w-12) .. +3); the other twelve go straight to their MFMAs and the four catch up on SIMDs the others have left.
    // Rows past the end of the sequence re-read row T-1 (always mapped); phase A masks them.
    constexpr int kLoaders = 4;   // loader waves (measured on the final kernel: 8 loaders 0.598 ms, 2 loaders 0.612, 4 loaders 0.592)
    constexpr int kDmaWave0 = 16 - kLoaders, kPairsPer = 16 / kLoaders;
    // memory row (relative to the segment's first token) of visited row tl <= T-1 of tensor a; REV: the gate (a == 2) of a
    // visited row is the gate of the token after it, clamped to the sequence (the one row past it is zeroed in gate_scan)
    auto src_row = [&](int a, int tl) -> unsigned {
        if constexpr (!REV) return (unsigned)tl;
        else return (unsigned)(a == 2 ? min(T - tl, T_total - 1 - t_begin) : T - 1 - tl);
    };
    auto dma_chunk = [&](int t_first, int a_lo, int a_hi) {
        if (w < kDmaWave0) return;
#pragma unroll
        for (int a = a_lo; a < a_hi; ++a) {
#pragma unroll
            for (int j = 0; j < kPairsPer; ++j) {
                const int pair = kPairsPer * (w - kDmaWave0) + j;
                const unsigned t = src_row(a, min(t_first + 2 * pair + (lane >> 5), T - 1));
                bf16_t* dst = a == 0 ? s_rq : a == 1 ? s_rk : a == 2 ? s_rg : s_rv;                 // a is a compile-time index
                // uniform base + 32-bit BYTE offset per lane (< 2^32: launcher guard): selects the SGPR-base addressing form,
                // no 64-bit per-lane address arithmetic (whose zero high word the compiler kept in -- and spilled from -- a VGPR)
                const unsigned boff = 2u * (t * gst[a] + 8u * (unsigned)(lane & 31));
                dma16_to_lds_async_p<NTD>(gsrc[a], boff, &dst[pair * PE]);
            }
        }
    };

    // ---- phase A thread map: wave w <-> channels [16w, 16w+16); lane = (channel quad c4 = lane>>4, row pair rp = lane&15).
    // The 16 row pairs of a channel quad sit in ONE 16-lane row, so the gate scan is four DPP row_shr adds per value:
    // no cross-wave totals, no barrier, no serial prefix loop; the thread owns two ADJACENT tokens, so it writes
    // k~^T / v^T directly (4-byte pieces).
    int rp = lane & 15, ch0 = 16 * w + 4 * (lane >> 4);
    bool gate_zero0 = rev_tail;                               // REV, first chunk of the sequence's last segment: row 0 has no gate
    // inclusive gate cumsum of this thread's 2 rows x 4 channels (rows >= nrem count as 0); true if the chunk's total
    // decay is too large for one chunk.  FULL: all C rows are in the sequence (no masks).  CLAMP: a single gate below -60 is
    // clamped to -60 -- only the cut path needs it: an unclamped gate below -60 makes the optimistic scan report a violation
    // by itself, and the cut path rescans with the clamp (8 v_max per thread and chunk less in the common path, round 4).
    auto gate_scan = [&](auto full_tag, auto clamp_tag, float (&bc)[2][4], int nrem) {
        constexpr bool FULL = decltype(full_tag)::value, CLAMP = decltype(clamp_tag)::value;
        float g0[4], g1[4];
        const bf16_t* gp = &s_rg[rp * PE + ch0];
        unpack4(*reinterpret_cast<const uint2*>(gp), g0);
        unpack4(*reinterpret_cast<const uint2*>(gp + DK), g1);
        const bool in0 = FULL || 2 * rp < nrem, in1 = FULL || 2 * rp + 1 < nrem;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if constexpr (CLAMP) {
                g0[c] = vmax_raw(g0[c], -kFullMaxDecay);
                g1[c] = vmax_raw(g1[c], -kFullMaxDecay);
            }
            if constexpr (REV) g0[c] = (gate_zero0 && rp == 0) ? 0.0f : g0[c];
            g1[c] = in1 ? g1[c] : 0.0f;
            bc[1][c] = (in0 ? g0[c] : 0.0f) + g1[c];         // the row pair's sum
        }
        // inclusive scan over the 16 row pairs of this 16-lane row, in place; the pair's first row is the pair's inclusive
        // value minus the second row's gate
        row_scan4(bc[1][0], bc[1][1], bc[1][2], bc[1][3]);
        bool viol = false;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            bc[0][c] = bc[1][c] - g1[c];
            viol |= (-bc[1][c] > kFullMaxDecay);              // b is monotone: the last row pair sees the chunk total
        }
        return viol;
    };
    // rows >= nv are zeroed; the thread that owns row nv-1 publishes R after the chunk.  FULL: nv == C.
    // q~ carries NO 1/sqrt(Dk): the scale is applied to o (linear), one multiply per output instead of one per q element.
    // MODE 1: slot 0 (X) and slot 1 (Y) are copied raw, slot 3 (Z) is gated like k; X goes to tile row 16 rr + rp (token
    // 2 rp + rr: the column permutation of the products, see the kernel header); Eo = the output factors e^{b+R} of this thread's
    // two tokens x four channels.
    // MODE 0: the raw q / k / v rows of this thread (two rows x four channels each), requested in front of the gate scan: they
    // arrive under the scan instead of being waited for one by one inside the tile writes
    auto read_raw = [&](uint2 (&hq)[2], uint2 (&hk)[2], uint2 (&hv)[2]) {
        const bf16_t* const rawp = &s_raw[rp * PE + ch0];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            if constexpr (!STATE_ONLY) hq[rr] = *reinterpret_cast<const uint2*>(rawp + rr * DK);
            hk[rr] = *reinterpret_cast<const uint2*>(rawp + RAWT + rr * DK);
            hv[rr] = *reinterpret_cast<const uint2*>(rawp + 3 * RAWT + rr * DK);
        }
    };
    auto write_tiles = [&](auto full_tag, const float (&bc)[2][4], int nv, int par, float (&Eo)[2][4], uint2 (&Zq)[2],
                           const uint2 (&hq)[2], const uint2 (&hk)[2], const uint2 (&hv)[2]) {
        constexpr bool FULL = decltype(full_tag)::value;
        constexpr float kLog2e = 1.4426950408889634f;
        uint2 kk[2], vv[2];                                   // packed k~ / v of the two rows, for the transposed pieces
        // column (element index) of this thread's channel quad in the q~ / k~ tiles: group w/2, piece c4 ^ ((row>>2)&3) =
        // c4 ^ ((rp>>1)&3) for both rows 2rp, 2rp+1, half w&1
        bf16_t* const qkp = &s_qk[2 * rp * SQ + 32 * (w >> 1) + 8 * ((lane >> 4) ^ ((rp >> 1) & 3)) + 4 * (w & 1)];
        // MODE 1, X tile: rows rp and 16 + rp, piece c4 ^ ((row>>2)&3) = c4 ^ ((rp>>2)&3)
        bf16_t* const xp = &s_qk[rp * SQ + 32 * (w >> 1) + 8 * ((lane >> 4) ^ ((rp >> 2) & 3)) + 4 * (w & 1)];
        const bf16_t* const rawp = &s_raw[rp * PE + ch0];
        bf16_t* const tp = &s_T[ch0 * ST + 2 * rp];          // transposed pieces: (channel ch0+i, tokens 2rp, 2rp+1) = one word
        if constexpr (MODE == 1) {
            // the raw operands first (copy X, copy + transpose Y): their registers are free before the gate arithmetic starts
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const bool valid = FULL || 2 * rp + rr < nv;
                const uint2 rx = *reinterpret_cast<const uint2*>(rawp + rr * DK);
                *reinterpret_cast<uint2*>(xp + 16 * rr * SQ) = valid ? rx : make_uint2(0u, 0u);
                const uint2 ry = *reinterpret_cast<const uint2*>(rawp + RAWT + rr * DK);
                kk[rr] = valid ? ry : make_uint2(0u, 0u);
                *reinterpret_cast<uint2*>(qkp + C * SQ + rr * SQ) = kk[rr];
            }
            *reinterpret_cast<unsigned*>(tp) = byte_perm(kk[1].x, kk[0].x, 0x05040100u);
            *reinterpret_cast<unsigned*>(tp + ST) = byte_perm(kk[1].x, kk[0].x, 0x07060302u);
            *reinterpret_cast<unsigned*>(tp + 2 * ST) = byte_perm(kk[1].y, kk[0].y, 0x05040100u);
            *reinterpret_cast<unsigned*>(tp + 3 * ST) = byte_perm(kk[1].y, kk[0].y, 0x07060302u);
            sched_fence();
        }
        const float4 R4 = *reinterpret_cast<const float4*>(&s_R[ch0]);
        const float Rc[4] = {R4.x, R4.y, R4.z, R4.w};
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int row = 2 * rp + rr;
            const bool valid = FULL || row < nv;
            float f[4], x[4], e[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                x[c] = __builtin_fmaf(bc[rr][c], kLog2e, Rc[c]);   // (b + R) log2 e, one fma: |b| <= 60, |R| <= kRenorm + 60
                if constexpr (!(STATE_ONLY && MODE == 0)) e[c] = fast_exp2(x[c]);
            }
            // k e^{-b} = k * rcp(e^{b}): v_rcp_f32 (1 ulp) + multiply; `/` and __fdividef both expand to the ~10-instruction
            // IEEE division sequence here (160 VALU instructions per thread and chunk).  The state-only pass has no q~: e^{-b}
            // directly, ONE transcendental per element instead of two (quarter-rate instructions: a third of phase A's issue time)
            float ri[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) ri[c] = (STATE_ONLY && MODE == 0) ? fast_exp2(-x[c]) : fast_rcp(e[c]);
            if constexpr (MODE == 1) {
#pragma unroll
                for (int c = 0; c < 4; ++c) Eo[rr][c] = e[c];
                Zq[rr] = *reinterpret_cast<const uint2*>(rawp + 3 * RAWT + rr * DK);   // raw Z rows (DG: q of these tokens)
                unpack4(Zq[rr], f);
                vv[rr].x = pack_bf16x2(f[0] * ri[0], f[1] * ri[1]);
                vv[rr].y = pack_bf16x2(f[2] * ri[2], f[3] * ri[3]);
                vv[rr].x = valid ? vv[rr].x : 0u;
                vv[rr].y = valid ? vv[rr].y : 0u;
            } else {
                if constexpr (!STATE_ONLY) {
                    uint2 pq;
                    unpack4(hq[rr], f);
                    pq.x = pack_bf16x2(f[0] * e[0], f[1] * e[1]);   // rows >= nv: finite values, zeroed as packed words
                    pq.y = pack_bf16x2(f[2] * e[2], f[3] * e[3]);
                    pq.x = valid ? pq.x : 0u;
                    pq.y = valid ? pq.y : 0u;
                    *reinterpret_cast<uint2*>(qkp + rr * SQ) = pq;
                }
                unpack4(hk[rr], f);
                kk[rr].x = pack_bf16x2(f[0] * ri[0], f[1] * ri[1]);
                kk[rr].y = pack_bf16x2(f[2] * ri[2], f[3] * ri[3]);
                kk[rr].x = valid ? kk[rr].x : 0u;
                kk[rr].y = valid ? kk[rr].y : 0u;
                if constexpr (!STATE_ONLY) *reinterpret_cast<uint2*>(qkp + C * SQ + rr * SQ) = kk[rr];
                (void)hv;                                       // v^T is written by write_vT, before the gate scan
            }
            if (FULL ? (rr == 1 && rp == C / 2 - 1) : (row == nv - 1)) {   // owner of the chunk's last row: R after the chunk
                bool need = false;
#pragma unroll
                for (int c = 0; c < 4; ++c) need |= x[c] < -kRenorm * kLog2e;
                *reinterpret_cast<float4*>(&s_Rn[ch0]) = make_float4(x[0], x[1], x[2], x[3]);
                if (need) s_flags[2 * par + 1] = 1;
                if constexpr (STATE_ONLY)
                    *reinterpret_cast<float4*>(&s_dec[ch0]) =
                        make_float4(__expf(bc[rr][0]), __expf(bc[rr][1]), __expf(bc[rr][2]), __expf(bc[rr][3]));
            }
        }
        // transposed pieces: element (channel ch0+i, tokens 2rp, 2rp+1) = one 4-byte word = one byte permute of the two rows
        if constexpr (MODE == 0) {
            *reinterpret_cast<unsigned*>(tp) = byte_perm(kk[1].x, kk[0].x, 0x05040100u);
            *reinterpret_cast<unsigned*>(tp + ST) = byte_perm(kk[1].x, kk[0].x, 0x07060302u);
            *reinterpret_cast<unsigned*>(tp + 2 * ST) = byte_perm(kk[1].y, kk[0].y, 0x05040100u);
            *reinterpret_cast<unsigned*>(tp + 3 * ST) = byte_perm(kk[1].y, kk[0].y, 0x07060302u);
        }
        if constexpr (MODE == 1) {
            *reinterpret_cast<unsigned*>(tp + DK * ST) = byte_perm(vv[1].x, vv[0].x, 0x05040100u);
            *reinterpret_cast<unsigned*>(tp + DK * ST + ST) = byte_perm(vv[1].x, vv[0].x, 0x07060302u);
            *reinterpret_cast<unsigned*>(tp + DK * ST + 2 * ST) = byte_perm(vv[1].y, vv[0].y, 0x05040100u);
            *reinterpret_cast<unsigned*>(tp + DK * ST + 3 * ST) = byte_perm(vv[1].y, vv[0].y, 0x07060302u);
        }
    };
    // MODE 0: v enters the products unscaled, so its transposed pieces need nothing from the gate scan: written first, their
    // registers are free before the scan starts (the hoisted raw rows are what the early prefetch costs in registers)
    auto write_vT = [&](auto full_tag, const uint2 (&hv)[2], int nv) {
        constexpr bool FULL = decltype(full_tag)::value;
        uint2 vv[2];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) vv[rr] = (FULL || 2 * rp + rr < nv) ? hv[rr] : make_uint2(0u, 0u);
        bf16_t* const tp = &s_T[(DK + ch0) * ST + 2 * rp];
        *reinterpret_cast<unsigned*>(tp) = byte_perm(vv[1].x, vv[0].x, 0x05040100u);
        *reinterpret_cast<unsigned*>(tp + ST) = byte_perm(vv[1].x, vv[0].x, 0x07060302u);
        *reinterpret_cast<unsigned*>(tp + 2 * ST) = byte_perm(vv[1].y, vv[0].y, 0x05040100u);
        *reinterpret_cast<unsigned*>(tp + 3 * ST) = byte_perm(vv[1].y, vv[0].y, 0x07060302u);
    };
    using FullT = std::true_type;
    using PartT = std::false_type;

    for (int c = tid; c < DK; c += 1024) s_R[c] = 0.0f;
    if constexpr (DG)
        for (int c = tid; c < DK; c += 1024) s_carry[c] = carry ? carry[(int64_t)slot * DK + c] : 0.0f;
    if (tid < 4) s_flags[tid] = 0;
    if (tid == 2) s_cut = 0;
    dma_chunk(0, STATE_ONLY ? 1 : 0, 4);
    wait_vmem();
    __syncthreads();   // DMA of chunk 0 landed
    int t0 = 0, par = 0;                                      // par: chunk parity
    int tp = 0, np = 0;                                       // previous chunk: its o is stored at the END of the next phase A
    f32x4 acc[2] = {};       // o^T: this wave's 16 columns (rows 4lg + r) x tokens [16nt, 16nt+16) (column li)
    // o straight from the accumulators.  The products were taken TRANSPOSED (state / v as the A operand), so a lane holds 4
    // consecutive columns of ONE token = one 8-byte store; the 16 waves' 32-byte pieces of a 512-byte row meet in L2.  No LDS
    // staging, no read-back.  Issued at the end of the NEXT chunk's phase A: a store blocks its wave while the address unit
    // is busy, and right after barrier (3) all 32 of them queued up in front of phase A; a wave that finishes phase A early
    // stores while the others still compute.
    // memory row (relative to the segment's first token) of visited row tl
    auto out_row = [&](int tl) -> unsigned { return (unsigned)(REV ? T - 1 - tl : tl); };
    uint2 opk[2] = {};                                         // MODE 0: the finished o of the previous chunk, packed (4 registers carried through phase A, not 8)
    uint2 auxr[2], aux2r[2];                                   // DG: aux / aux2 rows of this chunk's tokens (requested early)
    uint2 zq[2];                                               // DG: this thread's raw Z rows of the chunk (kept from phase A)
    const bf16_t* auxb = DG ? aux + b * saux.b + h * saux.h + t_begin * saux.t : nullptr;
    const bf16_t* aux2b = DG ? aux2 + b * saux2.b + h * saux2.h + t_begin * saux2.t : nullptr;
    bf16_t* dgb = DG ? dg + b * sdg.b + h * sdg.h + t_begin * sdg.t : nullptr;
    auto prefetch_prev = [&]() {                               // issued before step (4), consumed by store_prev in the next phase A
        if constexpr (DG) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const unsigned mr = out_row(min(tp + 2 * li + nt, T - 1));
                auxr[nt] = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(auxb) +
                                                           2u * (mr * (unsigned)saux.t + 16u * (unsigned)w + 4u * (unsigned)lg));
                aux2r[nt] = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(aux2b) +
                                                            2u * (mr * (unsigned)saux2.t + 16u * (unsigned)w + 4u * (unsigned)lg));
            }
        }
    };
    auto store_prev = [&]() {
        if constexpr (!STATE_ONLY && MODE == 0) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int row = 16 * nt + li;
                if (row < np) {
                    const unsigned boff = 2u * (out_row(tp + row) * (unsigned)so.t + 16u * (unsigned)w + 4u * (unsigned)lg);
                    *reinterpret_cast<uint2*>(reinterpret_cast<char*>(ob) + boff) = opk[nt];
                }
            }
        }
        if constexpr (MODE == 1) {
            float ov[2][4], dd[2][4];                          // this lane: tokens 2 li + nt, channels 16 w + 4 lg + r
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const bool valid = 2 * li + nt < np;
                float a4[4], p4[4], z4[4];
                if constexpr (DG) { unpack4(auxr[nt], a4); unpack4(aux2r[nt], p4); unpack4(zq[nt], z4); }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ov[nt][r] = acc[nt][r];                    // already scaled by e^{b+R} / sqrt(Dk) (end of its iteration)
                    if constexpr (DG) dd[nt][r] = valid ? z4[r] * p4[r] - a4[r] * ov[nt][r] : 0.0f;
                }
                if constexpr (DG) sched_fence();               // one token's unpacked operands at a time (register budget)
                if (valid) {
                    const unsigned mr = out_row(tp + 2 * li + nt);
                    uint2 po;
                    po.x = pack_bf16x2(ov[nt][0], ov[nt][1]);
                    po.y = pack_bf16x2(ov[nt][2], ov[nt][3]);
                    *reinterpret_cast<uint2*>(reinterpret_cast<char*>(ob) +
                                              2u * (mr * (unsigned)so.t + 16u * (unsigned)w + 4u * (unsigned)lg)) = po;
                }
            }
            if constexpr (DG) {
                // running sum over the visited rows: pair sums, inclusive scan over the 16 row pairs (one 16-lane row per
                // channel quad), + the carry of the earlier chunks; the last lane of the row publishes the new carry
                float ps[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) ps[r] = dd[0][r] + dd[1][r];
                // (read before the scan: the scan is where the emulator's lanes meet, so no lane sees lane 15's new value)
                const float4 c4v = *reinterpret_cast<const float4*>(&s_carry[16 * w + 4 * lg]);
                row_scan4(ps[0], ps[1], ps[2], ps[3]);
                const float cr[4] = {c4v.x, c4v.y, c4v.z, c4v.w};
                float g1v[4], g0v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { g1v[r] = cr[r] + ps[r]; g0v[r] = g1v[r] - dd[1][r]; }
                if (li == 15) *reinterpret_cast<float4*>(&s_carry[16 * w + 4 * lg]) = make_float4(g1v[0], g1v[1], g1v[2], g1v[3]);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    if (2 * li + nt < np) {
                        const unsigned mr = out_row(tp + 2 * li + nt);
                        uint2 po;
                        po.x = nt ? pack_bf16x2(g1v[0], g1v[1]) : pack_bf16x2(g0v[0], g0v[1]);
                        po.y = nt ? pack_bf16x2(g1v[2], g1v[3]) : pack_bf16x2(g0v[2], g0v[3]);
                        *reinterpret_cast<uint2*>(reinterpret_cast<char*>(dgb) +
                                                  2u * (mr * (unsigned)sdg.t + 16u * (unsigned)w + 4u * (unsigned)lg)) = po;
                    }
                }
            }
        }
    };
#ifdef LINA_K2_PROF
    unsigned long long pacc[16] = {}, plast = clock64();
    const unsigned long long pstart = plast;
#endif
    while (t0 < T) {
        // nothing per-lane is carried across iterations: the wave index sits in an SGPR, the lane index comes from v_mbcnt
        // (a spilled index would be reloaded through vmcnt, the counter the in-flight DMA also uses, and stall on it)
        lane = lane_id();
        opaque(lane);
        w = w_s; tid = w * 64 + lane; li = lane & 15; lg = lane >> 4; rp = lane & 15; ch0 = 16 * w + 4 * (lane >> 4);
        const int nrem = T - t0;
        int n = min(C, nrem);
        // ---------------- phase A: gate scan, scaled operands, transposed operands ----------------
        // MODE 1: the previous chunk's output leaves BETWEEN the gate scan and the tile writes -- its accumulators, factors
        // (and, in sweep K, the aux rows requested before step (4)) are dead before the tile writes need their registers
        float En[2][4];                                        // MODE 1: this chunk's output factors
        if (DG && np > 0) store_prev();                        // sweep K: before the gate scan (its registers: aux, aux2, q rows)
        {
            float bc[2][4];
            uint2 hq[2], hk[2], hv[2];
            if constexpr (MODE == 0) {
                read_raw(hq, hk, hv);
                if (nrem >= C) write_vT(FullT{}, hv, C); else write_vT(PartT{}, hv, n);
                sched_fence();
            }
            bool viol;
#ifdef LINA_K2_CLAMP_ALWAYS   // tools-only A/B build: the round-3 form
            using OptClamp = FullT;
#else
            using OptClamp = PartT;
#endif
            if (nrem >= C) viol = gate_scan(FullT{}, OptClamp{}, bc, nrem);   // workgroup-uniform: the mask-free form
            else viol = gate_scan(PartT{}, OptClamp{}, bc, nrem);
            if (viol) s_flags[2 * par] = 1;
            if (MODE == 1 && !DG && np > 0) store_prev();
            if (nrem >= C) write_tiles(FullT{}, bc, C, par, En, zq, hq, hk, hv);
            else write_tiles(PartT{}, bc, n, par, En, zq, hq, hk, hv);
        }
        K2_PROF(0);
        if (MODE == 0 && np > 0) store_prev();
        K2_PROF(10);
        __syncthreads();   // (2) operand tiles ready; raw q,k,g,v consumed
        K2_PROF(1);
        lane = lane_id(); opaque(lane); tid = w * 64 + lane; li = lane & 15; lg = lane >> 4; rp = lane & 15; ch0 = 16 * w + 4 * (lane >> 4);
        uint2 fl = *reinterpret_cast<const uint2*>(&s_flags[2 * par]);   // workgroup-uniform
        float rn = tid < DK ? s_Rn[tid] : 0.0f;                 // for the roll of R below, fetched in the same round trip
        if (fl.x) {
            // ---- rare: the decay inside this chunk exceeds e^-60 -> cut the chunk at the first such row ----
            float bc[2][4];
            gate_scan(PartT{}, FullT{}, bc, nrem);             // (with the clamp)
            int nc = C;
#pragma unroll
            for (int rr = 1; rr >= 0; --rr) {
                bool bad = false;
#pragma unroll
                for (int c = 0; c < 4; ++c) bad |= (-bc[rr][c] > kFullMaxDecay);
                if (bad) nc = 2 * rp + rr;
            }
            // first bad row of the workgroup through ONE LDS atomic (a shuffle reduction here made the compiler keep its six
            // lane-permutation addresses live across the whole main loop)
            if (nc < C) lds_atomic_max(&s_cut, C - nc);
            __syncthreads();
            n = max(min(n, C - s_cut), 1);
            __syncthreads();   // everyone has read s_cut; the optimistic tiles are dead
            if (tid == 0) { int z = 0; opaque(z); s_cut = z; }
            {
                uint2 hq[2], hk[2], hv[2];
                if constexpr (MODE == 0) { read_raw(hq, hk, hv); write_vT(PartT{}, hv, n); }
                write_tiles(PartT{}, bc, n, par, En, zq, hq, hk, hv);
            }
            __syncthreads();
            fl.y = s_flags[2 * par + 1];                       // the rewritten tiles may have changed both
            rn = tid < DK ? s_Rn[tid] : 0.0f;
        }
        const bool renorm = fl.y != 0;                         // workgroup-uniform: set in phase A, reset one chunk later
        const bool more = t0 + n < T;
        // (the loader waves interleaving their 16 DMA instructions with their own step-(1) MFMAs instead: 0.644 ms vs 0.592 --
        //  the pieces land later and everybody waits at (3); measured round 2, tests/gpu_k2var.sh)
        if (more) dma_chunk(t0 + n, STATE_ONLY ? 1 : 0, 4);   // next chunk's raw q,k,g,v fly under phase B
        if (STATE_ONLY && w == 0) {                                    // s_dec is stable between barriers (2) and (3)
            const float4 d = *reinterpret_cast<const float4*>(&s_dec[4 * lane]);
            decp.x *= d.x; decp.y *= d.y; decp.z *= d.z; decp.w *= d.w;
        }
        // next chunk's R: s_R is read by phase A only (before (2) / after (3)), s_Rn is stable between (2) and (3)
        if (tid < DK) s_R[tid] = renorm ? 0.0f : rn;
        K2_PROF(2);

        // ---------------- phase B ----------------
        // Every LDS operand read below is issued SEVERAL MFMAs ahead of its use by hand (rings of fragment registers +
        // sched_fence()): left to itself the compiler issues each read right before its MFMA under this register
        // pressure and every MFMA then eats a full LDS round trip (3400 of 12400 clocks per chunk in step (1) alone).
        acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (!STATE_ONLY) {
            if (w < 4) {
                // (2) A^T[s][t] = k~_s . q~_t, computed ONCE per workgroup by waves 0..3 (one per SIMD): wave w takes the
                //     whole 16 x 16 tile (mt = w&1: s block, nt = w>>1: t block), 8 MFMAs over the 256 channels.  Its C/D
                //     layout (col t = li, rows s = 4lg + r) is exactly where step (3)'s A operand wants the values: lane
                //     (li = t&15, lg) slot j of tile nt holds A[t][s] with s = 4lg + j (s block 0) or 16 + 4lg + (j-4)
                //     (s block 1) -- so each lane masks (s <= t) and stores its 4 values as one 8-byte piece.
                wave_priority<2>();                            // the SIMD's one wave with extra work goes first
                const int mt = w & 1, nt = w >> 1;
                const int pc = 8 * (lg ^ ((li >> 2) & 3));    // swizzled piece of this lane's row
                const bf16_t* kp = &s_k[(16 * mt + li) * SK + pc];
                const bf16_t* qp = &s_q[(16 * nt + li) * SQ + pc];
                bf16x8 kf[4], qf[4];                           // ring, 3 k-steps ahead
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) { kf[ks] = frag16(kp + 32 * ks); qf[ks] = frag16(qp + 32 * ks); }
                f32x4 at[G];                                   // one A per head: k-steps [8g/G, 8(g+1)/G) are head g's channels
#pragma unroll
                for (int g2 = 0; g2 < G; ++g2) at[g2] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    if (ks + 3 < 8) { kf[(ks + 3) & 3] = frag16(kp + 32 * (ks + 3)); qf[(ks + 3) & 3] = frag16(qp + 32 * (ks + 3)); }
                    sched_fence();
                    at[ks / (8 / G)] = mfma_bf16_16x16x32(kf[ks & 3], qf[ks & 3], at[ks / (8 / G)]);
                    sched_fence();
                }
                const int t = MODE == 1 ? 2 * li + nt : 16 * nt + li, sb = 16 * mt + 4 * lg;   // MODE 1: permuted token columns
#pragma unroll
                for (int g2 = 0; g2 < G; ++g2) {
                    uint2 pa;
                    pa.x = pack_bf16x2(sb <= t ? at[g2][0] : 0.0f, sb + 1 <= t ? at[g2][1] : 0.0f);
                    pa.y = pack_bf16x2(sb + 2 <= t ? at[g2][2] : 0.0f, sb + 3 <= t ? at[g2][3] : 0.0f);
                    *reinterpret_cast<uint2*>(&s_A[g2 * 1024 + (nt * 64 + lane) * 8 + 4 * mt]) = pa;
                }
                wave_priority<0>();
            }
            K2_PROF(3);
        }
        const bf16_t* ktp = &s_kT[(16 * NTL * hw + li) * ST + 8 * lg];   // k~^T fragment of this head's state tile p: + 16 p ST
        // operand fragments whose K dimension is the token axis, from the transposed tiles (one ds_read_b128 each).
        // (Round 3 measured the alternative built in round 2 -- NO transposed tiles, the fragments read with ds_read_b64_tr_b16
        //  from the row-major tiles, optionally with a 128 x 32 state block per wave that halves the q~ / k~^T operand reads:
        //  0.586 / 0.584 ms against 0.577 ms for this form at B=64,H=4,T=4096 (tests/gpu_k2tr.sh, profiles/r03_k2_variants.txt);
        //  removed.)
        auto ld_kt = [&](int p) -> bf16x8 { return frag16(ktp + 16 * p * ST); };   // k~^T of this head's state row tile p, tokens 8 lg .. + 7
        auto ld_v8 = [&]() -> bf16x8 { return frag16(&s_vT[(16 * w + li) * ST + 8 * lg]); };   // v^T of this wave's 16 columns, tokens 8 lg .. + 7
        auto ld_v4 = [&]() -> bf16x8 {                         // ... tokens 4 lg .. 4 lg + 3 and 16 + 4 lg .. 16 + 4 lg + 3
            const bf16_t* vp = &s_vT[(16 * w + li) * ST + 4 * lg];
            return frag8x2(vp, vp + 16);
        };
        // ring of k~^T fragments, TA tiles ahead (MODE 1 carries its 8 output factors across this phase: a shallower ring
        // instead of spills, whose reloads would wait on the loader waves' in-flight DMA)
        constexpr int TA = MODE == 1 ? 3 : 5;
        bf16x8 tf[8];
        bf16x8 vb2;                                            // v^T fragment of step (4) (tokens as k-slots 8lg..8lg+7)
        bf16x8 vb;                                             // v^T fragment of step (3), read before the tiles die at (3)
        if constexpr (!STATE_ONLY && G > 1) {
            // (1) for G heads per workgroup: the same products over this head's NTL/2 tile pairs (channels
            //     [D hw, D hw + D)); everything is requested up front (the state is only 64/G registers)
            constexpr int NPP = 8 / G;
            const bf16_t* qp = &s_q[li * SQ + 32 * NPP * hw + 8 * (lg ^ ((li >> 2) & 3))];
            bf16x8 qf[NPP][2];
#pragma unroll
            for (int pp = 0; pp < NPP; ++pp)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) qf[pp][nt] = frag16(qp + 16 * nt * SQ + 32 * pp);
            vb2 = ld_v8();
#pragma unroll
            for (int p = 0; p < (NTL < TA ? NTL : TA); ++p) tf[p] = ld_kt(p);
            sched_fence();
#pragma unroll
            for (int pp = 0; pp < NPP; ++pp) {
                bf16x8 bb;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    bb[r] = (short)f2bf(S[2 * pp][r]);
                    bb[4 + r] = (short)f2bf(S[2 * pp + 1][r]);
                }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[nt] = mfma_bf16_16x16x32(bb, qf[pp][nt], acc[nt]);
            }
        } else if constexpr (!STATE_ONLY) {
            // (1) o = q~ . S_old : one K = 32 MFMA per pair of 16-row state tiles (converted to bf16 in registers)
            const bf16_t* qp = &s_q[li * SQ + 8 * (lg ^ ((li >> 2) & 3))];
            constexpr int QA = MODE == 1 ? 2 : 3;                               // ring, QA tile pairs ahead (MODE 1: see TA)
            bf16x8 qf[4][2];
#pragma unroll
            for (int pp = 0; pp < QA; ++pp)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) qf[pp][nt] = frag16(qp + 16 * nt * SQ + 32 * pp);
#pragma unroll
            for (int pp = 0; pp < 8; ++pp) {
                if (pp + QA < 8) {
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        qf[(pp + QA) & 3][nt] = frag16(qp + 16 * nt * SQ + 32 * (pp + QA));
                } else if (DG) {                               // (sweep K carries 4 more registers: step (4)'s operands after step (1))
                } else if (pp == 8 - QA) {                     // the ring's free slots take step (4)'s first operands
                    vb2 = ld_v8();
                    tf[0] = ld_kt(0);
                } else if (pp == 9 - QA) {
                    tf[1] = ld_kt(1);
                    tf[2] = ld_kt(2);
                } else {
                    if constexpr (TA > 3) {
                        tf[3] = ld_kt(3);
                        tf[4] = ld_kt(4);
                    }
                }
                sched_fence();
                bf16x8 bb;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    bb[r] = (short)f2bf(S[2 * pp][r]);
                    bb[4 + r] = (short)f2bf(S[2 * pp + 1][r]);
                }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[nt] = mfma_bf16_16x16x32(bb, qf[pp & 3][nt], acc[nt]);   // o^T: rows = state columns, cols = tokens
                sched_fence();
            }
        } else {
            vb2 = ld_v8();
#pragma unroll
            for (int p = 0; p < (NTL < TA ? NTL : TA); ++p) tf[p] = ld_kt(p);
        }
        K2_PROF(4);
        // DG: the aux (k) / aux2 (dq) rows of THIS chunk's tokens are requested here -- the q~ ring's registers are free from now on, and
        // the loads have step (4), barrier (3), step (3) and the next gate scan to land (requested after barrier (3) they were
        // waited for with most of their latency exposed: SQ_WAIT_ANY 65 % of the wave cycles against 47 % for sweep V)
        tp = t0; np = n;
        if constexpr (DG) {
            if constexpr (!STATE_ONLY && G == 1) {
                vb2 = ld_v8();
#pragma unroll
                for (int p = 0; p < TA; ++p) tf[p] = ld_kt(p);
            }
            prefetch_prev();
            sched_fence();
        }
        // (4) S' += k^^T v
#pragma unroll
        for (int p = 0; p < NTL; ++p) {
            if (p + TA < NTL) tf[(p + TA) & 7] = ld_kt(p + TA);
            sched_fence();
            S[p] = mfma_bf16_16x16x32(tf[p & 7], vb2, S[p]);
            sched_fence();
        }
        if (renorm && MODE == 1) {                       // rare: S'^T <- S'^T diag(e^{R}): the gated channel is the tile COLUMN
            const float f = fast_exp2(s_Rn[16 * w + li]);
#pragma unroll
            for (int p = 0; p < NTL; ++p) { S[p][0] *= f; S[p][1] *= f; S[p][2] *= f; S[p][3] *= f; }
        } else if (renorm) {                             // rare: S' <- e^{R} S' (R = s_Rn, the value after this chunk)
#pragma unroll
            for (int p = 0; p < NTL; ++p) {
                const float4 r4 = *reinterpret_cast<const float4*>(&s_Rn[tile_ch(p) + 4 * lg]);
                S[p][0] *= fast_exp2(r4.x); S[p][1] *= fast_exp2(r4.y); S[p][2] *= fast_exp2(r4.z); S[p][3] *= fast_exp2(r4.w);
            }
        }
        // v fragment of step (3) (same token order as the C/D rows of mask(A)): read before the tiles die at (3)
        if constexpr (!STATE_ONLY) vb = ld_v4();
        K2_PROF(5);
        // the DMA was issued through inline assembly: this wave's part has landed (the DG rows requested after it may not) ...
        wait_vmem_but<DG ? 4 : 0>();
        K2_PROF(8);
        __syncthreads();   // (3) ... and so has everybody's; operand tiles dead; mask(A) complete (its own buffer)
        K2_PROF(9);
        lane = lane_id(); opaque(lane); tid = w * 64 + lane; li = lane & 15; lg = lane >> 4;   // re-derive, do not carry
        if (tid == 0) {                                        // read by all before (3); set again two chunks later, after (2) of the next
            int z = 0;
            opaque(z);                                         // materialised here: hoisted out of the loop the constant was SPILLED
            s_flags[2 * par] = (unsigned)z; s_flags[2 * par + 1] = (unsigned)z;
        }
        if constexpr (!STATE_ONLY) {
            // (3) o += mask(A) . v -- AFTER the barrier: no barrier of its own for mask(A); s_A is rewritten only after the
            //     next chunk's barrier (2)
            acc[0] = mfma_bf16_16x16x32(vb, frag16(&s_A[hw * 1024 + (0 * 64 + lane) * 8]), acc[0]);   // o^T += v^T . mask(A)^T
            acc[1] = mfma_bf16_16x16x32(vb, frag16(&s_A[hw * 1024 + (1 * 64 + lane) * 8]), acc[1]);
            K2_PROF(7);
            if constexpr (MODE == 0) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    opk[nt].x = pack_bf16x2(acc[nt][0] * scale, acc[nt][1] * scale);
                    opk[nt].y = pack_bf16x2(acc[nt][2] * scale, acc[nt][3] * scale);
                }
            }
            if constexpr (MODE == 1) {                         // the output factors die here, not in the next phase A
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[nt][r] *= En[nt][r] * scale;
            }
        }
        par ^= 1;
        t0 += n;
        if constexpr (REV) gate_zero0 = false;
    }
    lane = lane_id(); opaque(lane); li = lane & 15; lg = lane >> 4;
    store_prev();                                              // the last chunk (T >= 1)
#ifdef LINA_K2_PROF
#ifndef LINA_K2_PROF_WHICH
#define LINA_K2_PROF_WHICH 0      // tools-only: 0 = every instantiation records (the last launch wins), 1 = only the forward STATE_ONLY pass
#endif
    constexpr bool prof_me = LINA_K2_PROF_WHICH == 0 || (LINA_K2_PROF_WHICH == 1 && STATE_ONLY && !REV);
    if (prof_me && blockIdx.x == 0 && lane_id() == 0)
        for (int i = 0; i < 16; ++i) lina_k2_prof[w_s * 16 + i] = pacc[i];
    if (prof_me && blockIdx.x < 1024 && w_s == 0 && lane_id() == 0) {
        lina_k2_prof[256 + 3 * blockIdx.x] = clock64() - pstart;
        lina_k2_prof[256 + 3 * blockIdx.x + 1] = pacc[8];
        lina_k2_prof[256 + 3 * blockIdx.x + 2] = pacc[9];
    }
#endif

    if (STATE_ONLY && dec_out && w == 0)
        *reinterpret_cast<float4*>(dec_out + (int64_t)slot * DK + 4 * lane) = decp;
    if (ht) {
        __syncthreads();                                     // s_R of the last chunk is visible (STATE_ONLY has no barrier (4))
#pragma unroll
        for (int p = 0; p < NTL; ++p) {                      // S = diag(e^{R}) S'
            const float4 r4 = *reinterpret_cast<const float4*>(&s_R[tile_ch(p) + 4 * lg]);
            S[p][0] *= fast_exp2(r4.x); S[p][1] *= fast_exp2(r4.y); S[p][2] *= fast_exp2(r4.z); S[p][3] *= fast_exp2(r4.w);
        }
        float* hp = ht + (((int64_t)slot * G + hw) * D + 4 * lg) * HS + (NCB > 1 ? cb * D : 0) + li;
#pragma unroll
        for (int p = 0; p < NTL; ++p)
#pragma unroll
            for (int r = 0; r < 4; ++r) hp[(tile_row(p) + r) * HS + tile_col(p)] = S[p][r];
    }
}

// true when the full-head kernel can take this call (16-byte aligned rows everywhere); D = 128 / 64: groups of 2 / 4 heads
// per workgroup, which must be adjacent in memory (head stride == D, the [B,T,H,D] layout) and inside one batch row
// kernel arguments that only the backward's sweeps use
#define LINA_FWD_ONLY 1.0f, (const bf16_t*)nullptr, lina_bht_strides{}, (const bf16_t*)nullptr, lina_bht_strides{},        \
                      (bf16_t*)nullptr, lina_bht_strides{}, (const float*)nullptr

static bool full_ok(int H, int Dk, int Dv, int dtype, const void* q, const void* k, const void* v, const void* gk,
                    const void* o, int g_dtype, lina_bht_strides sq, lina_bht_strides sk, lina_bht_strides sv,
                    lina_bht_strides sg, lina_bht_strides so, bool two_blocks = false) {
    // two_blocks: also Dk = 256, Dv = 512 (two value column blocks in one launch, NCB = 2 of the kernel)
    if (dtype != LINA_BF16 || g_dtype != LINA_BF16 || (Dk != 256 && Dk != 128 && Dk != 64)) return false;
    if (Dk != Dv && !(two_blocks && Dk == 256 && Dv == 512)) return false;
    const int G = 256 / Dk;
    if (G > 1) {
        if (H % G) return false;
        auto adj = [Dk](lina_bht_strides s) { return s.h == Dk; };
        if (!adj(sq) || !adj(sk) || !adj(sv) || !adj(sg) || !adj(so)) return false;
    }
    auto al = [](lina_bht_strides s, int m) { return s.b % m == 0 && s.h % m == 0 && s.t % m == 0; };
    if (!al(sq, 8) || !al(sk, 8) || !al(sv, 8) || !al(so, 8) || !al(sg, 8)) return false;
    auto small = [](lina_bht_strides s) { return s.t >= 0 && s.t < (1LL << 20); };   // 32-bit in-sequence offsets
    if (!small(sq) || !small(sk) || !small(sv) || !small(so) || !small(sg)) return false;
    auto p16 = [](const void* p) { return ((uintptr_t)p & 15u) == 0; };
    return p16(q) && p16(k) && p16(v) && p16(gk) && p16(o);
}

int launch_chunk_full(const void* q, const void* k, const void* v, const void* gk, void* o, const float* h0,
                      float* ht, int B, int H, int T, int Dk, int Dv, lina_bht_strides sq, lina_bht_strides sk,
                      lina_bht_strides sv, lina_bht_strides sg, lina_bht_strides so, int dtype, int g_dtype,
                      float scale, lina_stream_t stream, bool* taken) {
    auto fits32 = [T](lina_bht_strides st) { return (int64_t)T * st.t < (1LL << 31); };
    *taken = full_ok(H, Dk, Dv, dtype, q, k, v, gk, o, g_dtype, sq, sk, sv, sg, so, true) && fits32(sq) && fits32(sk) &&
             fits32(sv) && fits32(sg) && fits32(so);
    if (!*taken) return LINA_OK;
    if (Dv == 2 * Dk) {                                       // 256 x 512 heads: both value column blocks in ONE launch
        LINA_LAUNCH((gla_chunk_bf16_h256_kernel<false, 1, 0, false, false, false, 2>), dim3((unsigned)(2 * B * H)), dim3(1024), 0,
                    stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)gk, (bf16_t*)o, h0, ht,
                    (float*)nullptr, H, T, 1, T, sq, sk, sv, sg, so, scale, LINA_FWD_ONLY);
        return check_launch("lina_gla_chunk_fwd(full, two value blocks)");
    }
    const int G = 256 / Dk;
    dim3 grid((unsigned)(B * H / G));
#define LINA_FULL(GG)                                                                                                  \
    LINA_LAUNCH((gla_chunk_bf16_h256_kernel<false, GG>), grid, dim3(1024), 0, stream, (const bf16_t*)q, (const bf16_t*)k, \
                (const bf16_t*)v, (const bf16_t*)gk, (bf16_t*)o, h0, ht, (float*)nullptr, H, T, 1, T, sq, sk, sv, sg, so, \
                scale, LINA_FWD_ONLY)
    if (G == 1) LINA_FULL(1); else if (G == 2) LINA_FULL(2); else LINA_FULL(4);
#undef LINA_FULL
    return check_launch("lina_gla_chunk_fwd(full)");
}

// ------------------------------------------------------------------------------------------------------------
// Segment-parallel forward for SMALL B*H (a training micro-batch gives only B*H workgroups to the kernel above):
// the sequence is cut into nseg segments that run concurrently.
//   pass 1  STATE_ONLY, zero start: local end state L_s and decay product P_s of every segment
//   combine S_start[0] = h0, S_start[s+1] = diag(P_s) S_start[s] + L_s   (elementwise, sequential over nseg)
//   pass 2  the full kernel on every segment with h0 = S_start[s]
// Exact (the recurrence is linear in the state); costs one extra pass over k, g, v and 2 x nseg state tiles of
// workspace traffic per head, buys nseg x the workgroups.
// ------------------------------------------------------------------------------------------------------------
// A slot's state block is [G][D][D] = 256 * D floats (G = 256 / D heads side by side); row c of that block (0 <= c < 256)
// decays by P[slot][c].
// CARRY (the backward's reverse pass: slots in visiting order, s = 0 is the LAST segment of the sequence; S = dS / scale):
// also the gate gradient that enters segment s from all later tokens, the telescoped form of reverse-cumsum(q dq - k dk):
//     carry[slot][c] = scale e^{g_t[c]} sum_j S_{t-1}[c][j] dS_t[c][j],   t = first token after the segment
// with S_{t-1} = SstartF of the following segment (forward pass) and dS_t = the state entering this slot; the last segment's
// carry is dg_tail (the caller's sum_j final_state (.) dht) or 0.
template <bool CARRY>
// L and Sstart may be ONE buffer (every thread reads its element of L[slot] before it overwrites it with Sstart[slot]).
__global__ __launch_bounds__(256) void gla_seg_combine_kernel(const float* L, const float* __restrict__ P,
                                                              const float* h0, float h0_scale, float* Sstart,
                                                              float* ht, int nseg, int D, const float* __restrict__ SstartF,
                                                              const bf16_t* __restrict__ gk, lina_bht_strides sg,
                                                              const float* dg_tail, float* __restrict__ carry, int H,
                                                              int Tseg, float scale) {
    constexpr int DK = 256;
    const int64_t blk = (int64_t)DK * D;                      // floats per slot
    const int bh = blockIdx.x;                                // head group
    const int e = (blockIdx.y * 256 + threadIdx.x) * 4;       // element of the slot's state block, 4 columns per thread
    const int c = e / D;
    float4 S = h0 ? *reinterpret_cast<const float4*>(h0 + (int64_t)bh * blk + e) : make_float4(0.f, 0.f, 0.f, 0.f);
    S.x *= h0_scale; S.y *= h0_scale; S.z *= h0_scale; S.w *= h0_scale;
    for (int s = 0; s < nseg; ++s) {
        const int64_t slot = (int64_t)bh * nseg + s;
        const float4 l = *reinterpret_cast<const float4*>(L + slot * blk + e);
        *reinterpret_cast<float4*>(Sstart + slot * blk + e) = S;
        if constexpr (CARRY) {
            const bool lead = (e % D) == 0;                   // D/4 consecutive lanes hold one row
            if (s == 0) {
                if (lead) carry[slot * DK + c] = dg_tail ? dg_tail[(int64_t)bh * DK + c] : 0.0f;
            } else {
                const int seg_t = nseg - 1 - s;               // this slot's segment in time order; the next one starts at t
                const float4 f = *reinterpret_cast<const float4*>(SstartF + ((int64_t)bh * nseg + seg_t + 1) * blk + e);
                float dot = f.x * S.x + f.y * S.y + f.z * S.z + f.w * S.w;
                for (int m = 1; m < D / 4; m <<= 1) dot += shfl_xor(dot, m);
                if (lead) {
                    const int G = DK / D, first = bh * G, t = (seg_t + 1) * Tseg;
                    const float g = fmaxf(bf2f(gk[(int64_t)(first / H) * sg.b + (int64_t)(first % H + c / D) * sg.h +
                                                 (int64_t)t * sg.t + c % D]), -kFullMaxDecay);
                    carry[slot * DK + c] = scale * __expf(g) * dot;
                }
            }
        }
        const float p = P[slot * DK + c];
        S.x = p * S.x + l.x; S.y = p * S.y + l.y; S.z = p * S.z + l.z; S.w = p * S.w + l.w;
    }
    if (ht) *reinterpret_cast<float4*>(ht + (int64_t)bh * blk + e) = S;
}

}  // namespace lina

extern "C" int64_t lina_gla_chunk_fwd_seg_workspace(int B, int H, int Dk, int Dv, int nseg) {
    if (B <= 0 || H <= 0 || Dk <= 0 || Dv <= 0 || nseg <= 0) return 0;
    return (int64_t)sizeof(float) * B * H * nseg * ((int64_t)Dk * Dv + Dk);
}

extern "C" int lina_gla_chunk_fwd_seg(const void* q, const void* k, const void* v, const void* gk, void* o,
                                      const float* h0, float* ht, float* workspace, int nseg, int B, int H, int T,
                                      int Dk, int Dv, lina_bht_strides sq, lina_bht_strides sk, lina_bht_strides sv,
                                      lina_bht_strides sg, lina_bht_strides so, int dtype, int g_dtype, float scale,
                                      lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(q && k && v && gk && o && workspace, "lina_gla_chunk_fwd_seg: null pointer");
    LINA_REQUIRE(B > 0 && H > 0 && T > 0, "lina_gla_chunk_fwd_seg: B,H,T must be positive");
    LINA_REQUIRE(nseg >= 1, "lina_gla_chunk_fwd_seg: nseg must be >= 1");
    auto fits32 = [T](lina_bht_strides st) { return (int64_t)T * st.t < (1LL << 31); };
    if (!(full_ok(H, Dk, Dv, dtype, q, k, v, gk, o, g_dtype, sq, sk, sv, sg, so) && fits32(sq) && fits32(sk) && fits32(sv) &&
          fits32(sg) && fits32(so)))
        return fail(LINA_ERR_UNSUPPORTED, "lina_gla_chunk_fwd_seg: needs bf16 tensors and gates, Dk = Dv in {64,128,256}, "
                                          "adjacent heads, 16-byte aligned rows (use lina_gla_chunk_fwd)");
    const int G = 256 / Dk;
    const int Tseg = ((T + nseg - 1) / nseg + kFullC - 1) / kFullC * kFullC;   // whole 32-token chunks per segment
    const int ns = (T + Tseg - 1) / Tseg;                                       // segments that hold tokens
    const int64_t slots = (int64_t)(B * H / G) * ns;          // one slot = G heads x one segment: 256 * Dk state floats
    // workspace = [ states | decay products ]: the local end states of pass 1 become the segment START states in place, and
    // stay there after the call -- lina_gla_chunk_bwd_full takes them as seg_states (same T and nseg) instead of recomputing
    float* L = workspace;
    float* Sstart = workspace;
    float* P = Sstart + slots * 256 * Dk;
    dim3 grid((unsigned)slots);
    // (plain cache policy for both passes: the second one finds k, g, v where the first one left them)
#define LINA_SEG(SO, GG, OO, H0, HT, PP)                                                                               \
    LINA_LAUNCH((gla_chunk_bf16_h256_kernel<SO, GG, 0, false, false, false>), grid, dim3(1024), 0, stream, (const bf16_t*)q, (const bf16_t*)k,  \
                (const bf16_t*)v, (const bf16_t*)gk, (bf16_t*)(OO), (const float*)(H0), (float*)(HT), (float*)(PP), H, T, \
                ns, Tseg, sq, sk, sv, sg, so, scale, LINA_FWD_ONLY)
    if (G == 1) LINA_SEG(true, 1, nullptr, nullptr, L, P); else if (G == 2) LINA_SEG(true, 2, nullptr, nullptr, L, P);
    else LINA_SEG(true, 4, nullptr, nullptr, L, P);
    LINA_LAUNCH(gla_seg_combine_kernel<false>, dim3((unsigned)(B * H / G), (unsigned)(Dk / 4)), dim3(256), 0, stream,
                (const float*)L, (const float*)P, h0, 1.0f, Sstart, ht, ns, Dk, (const float*)nullptr,
                (const bf16_t*)nullptr, sg, (const float*)nullptr, (float*)nullptr, H, Tseg, scale);
    if (G == 1) LINA_SEG(false, 1, o, Sstart, nullptr, nullptr); else if (G == 2) LINA_SEG(false, 2, o, Sstart, nullptr, nullptr);
    else LINA_SEG(false, 4, o, Sstart, nullptr, nullptr);
#undef LINA_SEG
    return check_launch("lina_gla_chunk_fwd_seg");
}

// ------------------------------------------------------------------------------------------------------------
// K2b on the full-head kernel (bf16, Dk = Dv in {64,128,256}): three sweeps of the body above, each at the forward's cost,
//     V  (REV; q,k,v := k,q,do)            dv, dS            key-gated   = the forward kernel on the reversed sequence
//     Q  (MODE 1; X,Y,Z := do,v,k)         dq                value-gated, state S^T
//     K  (MODE 1, REV; X,Y,Z := v,do,q)    dk, dg            value-gated, state dS^T; dg = running sum of q dq - k dk
// With nseg > 1 every sweep runs on all segments concurrently from boundary states: a state-only forward pass + combine
// gives S at every segment start, a state-only reverse pass + combine gives dS at every segment end and the gate-gradient
// carry of every segment (gla_seg_combine_kernel<true>).  The state is kept as dS / scale (the sweeps apply scale to
// their outputs), so dht enters as dht / scale and dh0 leaves as scale e^{g_0} (.) state.
// ------------------------------------------------------------------------------------------------------------
namespace lina {
// dh0[c][j] = scale e^{g_0[c]} R[c][j]  (R = the reverse sweep's end state: dS_0 / scale)
__global__ __launch_bounds__(256) void gla_bwd_dh0_kernel(const float* __restrict__ R, const bf16_t* __restrict__ gk,
                                                          lina_bht_strides sg, float* __restrict__ dh0, int H, int D,
                                                          float scale) {
    constexpr int DK = 256;
    const int bh = blockIdx.x, G = DK / D, first = bh * G;
    const int e = (blockIdx.y * 256 + threadIdx.x) * 4, c = e / D;
    const float g = fmaxf(bf2f(gk[(int64_t)(first / H) * sg.b + (int64_t)(first % H + c / D) * sg.h + c % D]), -kFullMaxDecay);
    const float f = scale * __expf(g);
    const float4 r = *reinterpret_cast<const float4*>(R + (int64_t)bh * DK * D + e);
    *reinterpret_cast<float4*>(dh0 + (int64_t)bh * DK * D + e) = make_float4(f * r.x, f * r.y, f * r.z, f * r.w);
}
}  // namespace lina

static int64_t bwd_full_ws_floats(int B, int H, int T, int Dk, int nseg) {
    const int64_t G = 256 / Dk, groups = (int64_t)B * H / G, slots = groups * nseg, blk = 256 * (int64_t)Dk;
    // SstartF | SstartR | P | carry | end state of the reverse pass
    return 2 * slots * blk + 2 * slots * 256 + groups * blk;
}

extern "C" int64_t lina_gla_chunk_bwd_full_workspace(int B, int H, int T, int Dk, int Dv, int nseg) {
    if (B <= 0 || H <= 0 || T <= 0 || nseg <= 0 || Dk != Dv || (Dk != 64 && Dk != 128 && Dk != 256) || H % (256 / Dk)) return 0;
    return (int64_t)sizeof(float) * bwd_full_ws_floats(B, H, T, Dk, nseg);
}

extern "C" int lina_gla_chunk_bwd_full(const void* q, const void* k, const void* v, const void* gk, const void* d_o,
                                       const float* h0, const float* dht, const float* dg_tail, void* dq, void* dk,
                                       void* dv, void* dg, float* dh0, float* workspace, const float* seg_states,
                                       int nseg, int B, int H, int T,
                                       int Dk, int Dv, lina_bht_strides sq, lina_bht_strides sk, lina_bht_strides sv,
                                       lina_bht_strides sg, lina_bht_strides sdo, lina_bht_strides sdq,
                                       lina_bht_strides sdk, lina_bht_strides sdv, lina_bht_strides sdg, int dtype,
                                       int g_dtype, float scale, lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(q && k && v && gk && d_o && dq && dk && dv && dg && workspace, "lina_gla_chunk_bwd_full: null pointer");
    LINA_REQUIRE(B > 0 && H > 0 && T > 0 && nseg >= 1, "lina_gla_chunk_bwd_full: B,H,T,nseg must be positive");
    LINA_REQUIRE(scale != 0.0f, "lina_gla_chunk_bwd_full: scale must be non-zero");
    LINA_REQUIRE(!dht == !dg_tail, "lina_gla_chunk_bwd_full: dht and dg_tail come together");
    auto fits32 = [T](lina_bht_strides st) { return (int64_t)T * st.t < (1LL << 31); };
    const bool ok = full_ok(H, Dk, Dv, dtype, q, k, v, gk, dv, g_dtype, sq, sk, sv, sg, sdv) &&
                    full_ok(H, Dk, Dv, dtype, d_o, dq, dk, dg, dq, g_dtype, sdo, sdq, sdk, sdg, sdq) && fits32(sq) &&
                    fits32(sk) && fits32(sv) && fits32(sg) && fits32(sdo) && fits32(sdq) && fits32(sdk) && fits32(sdv) &&
                    fits32(sdg);
    if (!ok)
        return fail(LINA_ERR_UNSUPPORTED, "lina_gla_chunk_bwd_full: needs bf16 tensors and gates, Dk = Dv in {64,128,256}, "
                                          "adjacent heads, 16-byte aligned rows (use lina_gla_chunk_bwd)");
    const int G = 256 / Dk;
    const int Tseg = ((T + nseg - 1) / nseg + kFullC - 1) / kFullC * kFullC;   // whole 32-token chunks per segment
    const int ns = (T + Tseg - 1) / Tseg;
    const int64_t groups = (int64_t)B * H / G, slots = groups * ns, blk = 256 * (int64_t)Dk;
    float* SF = workspace;            // local end states of a state-only pass, then the segment start states (in place)
    float* SR = SF + slots * blk;
    float* P = SR + slots * blk;
    float* carry = P + slots * 256;
    float* endR = carry + slots * 256;
    const float inv = 1.0f / scale;
    const lina_bht_strides z{};
    dim3 grid((unsigned)slots);
    const bf16_t *Q = (const bf16_t*)q, *K = (const bf16_t*)k, *V = (const bf16_t*)v, *GK = (const bf16_t*)gk,
                 *DO = (const bf16_t*)d_o;
#define LINA_BW(SO, GG, MODE, REV, DGM, A0, A1, A3, OO, H0, HT, PP, S0, S1, S3, SOO, H0S, AUX, SAUX, CARRY)               \
    LINA_LAUNCH((gla_chunk_bf16_h256_kernel<SO, GG, MODE, REV, DGM>), grid, dim3(1024), 0, stream, A0, A1, A3, GK,       \
                (bf16_t*)(OO), (const float*)(H0), (float*)(HT), (float*)(PP), H, T, ns, Tseg, S0, S1, S3, sg, SOO, scale, \
                H0S, (const bf16_t*)(AUX), SAUX, (const bf16_t*)dq, sdq, (bf16_t*)dg, sdg, (const float*)(CARRY))
#define LINA_BW_G(...)                                                                                                  \
    do { if (G == 1) LINA_BW(__VA_ARGS__); } while (0)
    const float* startF = h0;         // per-slot start states of sweep Q / of sweeps V, K
    const float* startR = dht;
    const float* carry_in = dg_tail;
    float startR_scale = inv;
    float* endV = dh0 ? endR : nullptr;
#define LINA_BW_ALL(GG)                                                                                                 \
    do {                                                                                                                \
        if (ns > 1) {                                                                                                   \
            if (!seg_states) {                                                                                          \
                LINA_BW(true, GG, 0, false, false, Q, K, V, nullptr, nullptr, SF, P, sq, sk, sv, z, 1.0f, nullptr, z, nullptr); \
                LINA_LAUNCH(gla_seg_combine_kernel<false>, dim3((unsigned)groups, (unsigned)(Dk / 4)), dim3(256), 0, stream, \
                            (const float*)SF, (const float*)P, h0, 1.0f, SF, (float*)nullptr, ns, Dk, (const float*)nullptr, \
                            (const bf16_t*)nullptr, sg, (const float*)nullptr, (float*)nullptr, H, Tseg, scale);        \
            }                                                                                                           \
            const float* sf = seg_states ? seg_states : SF;                                                             \
            LINA_BW(true, GG, 0, true, false, K, Q, DO, nullptr, nullptr, SR, P, sk, sq, sdo, z, 1.0f, nullptr, z, nullptr); \
            LINA_LAUNCH(gla_seg_combine_kernel<true>, dim3((unsigned)groups, (unsigned)(Dk / 4)), dim3(256), 0, stream,  \
                        (const float*)SR, (const float*)P, dht, inv, SR, dh0 ? endR : (float*)nullptr, ns, Dk, sf, GK,   \
                        sg, dg_tail, carry, H, Tseg, scale);                                                            \
            startF = sf; startR = SR; carry_in = carry; startR_scale = 1.0f; endV = nullptr;                            \
        }                                                                                                               \
        LINA_BW(false, GG, 0, true, false, K, Q, DO, dv, startR, endV, nullptr, sk, sq, sdo, sdv, startR_scale, nullptr, z,  \
                nullptr);                                                                                               \
        LINA_BW(false, GG, 1, false, false, DO, V, K, dq, startF, nullptr, nullptr, sdo, sv, sk, sdq, 1.0f, nullptr, z,  \
                nullptr);                                                                                               \
        LINA_BW(false, GG, 1, true, true, V, DO, Q, dk, startR, nullptr, nullptr, sv, sdo, sq, sdk, startR_scale, K, sk,    \
                carry_in);                                                                                              \
    } while (0)
    if (G == 1) LINA_BW_ALL(1); else if (G == 2) LINA_BW_ALL(2); else LINA_BW_ALL(4);
#undef LINA_BW_ALL
#undef LINA_BW_G
#undef LINA_BW
    if (dh0)
        LINA_LAUNCH(gla_bwd_dh0_kernel, dim3((unsigned)groups, (unsigned)(Dk / 4)), dim3(256), 0, stream,
                    (const float*)endR, GK, sg, dh0, H, Dk, scale);
    return check_launch("lina_gla_chunk_bwd_full");
}

#ifdef LINA_K2_PROF
extern "C" int lina_k2_prof_read(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(lina_k2_prof), sizeof(unsigned long long) * (256 + 3 * 1024));
}
#endif
