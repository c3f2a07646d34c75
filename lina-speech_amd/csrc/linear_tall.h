// linear_tall.h -- the decode-step projections at a LARGE utterance batch (M >= kTallMinRows = 160 rows): main loop shared by the plain /
// LayerNorm-folded / SwiGLU projection (linear_skinny.hip) and by the fused input side of a GLA mixer (gla_inproj.hip).
//
// The "skinny" kernels of those files are built for M <= 64: a workgroup owns 64 rows x 16-32 columns and splits K over its
// waves, every operand byte goes straight from L2 into MFMA fragments once per workgroup.  At M = 512 (the batch the headline
// metric is quoted on) that tiling re-reads the activations once per 32 columns and the weights once per 64 rows: the
// in-projection pulled 1.15 MB through every CU's vector-memory path and ran 38 us for 4.3 GFLOP (rocprofv3,
// profiles/r05_step_b512_before_timeline.txt) -- 19 % of the step, the up-projection another 13 %.
//
// Tall tiling: a workgroup of NWV = 4 waves owns 16 MTW NWV rows x (16 G) weight rows -- 64 rows since LINA_TALL_MTW = 1 became the
// default (128 in the first versions, which the byte count below is written for); wave w owns rows [16 MTW w, 16 MTW (w + 1)) for the
// WHOLE contraction (no split-K, no reduction through LDS: the accumulators are final), so
//   * its A fragments (fragment-major: 1 KiB contiguous per load instruction, skinny_frag.h) are nobody else's;
//   * the weight fragments of a k-step are needed by all four waves: fetched once per workgroup;
//   * both go global -> LDS by DMA (global_load_lds_dwordx4, lane-linear = exactly the fragment-major image, so the consumer's
//     ds_read_b128 is conflict-free) through a three-stage ring with counted waits (tall_core below), one barrier per stage.
// Bytes through a CU per 128 x 64 output tile: (128 + 64) x K x e instead of 4 x (64 + 32) x 2 x K x e -- 2.7 x fewer for the
// same flops, and the k-loop overlaps loads with MFMAs instead of "everything in flight, then compute, then reduce".
// LayerNorm statistics: per-lane sums over the lane's own fragment elements (2 registers per m-tile), reduced over the four
// lane groups of a row at the end and fetched by the lanes that hold that row's outputs with one shuffle per row.
#pragma once
#include <lina_dev.h>
#include <type_traits>
#include "lina_common.h"
#include "skinny_frag.h"

namespace lina {

#ifndef LINA_TALL_MTW
#define LINA_TALL_MTW 1            // 1: 64-row workgroups (twice as many, two to three per CU): in-projection 21.4 -> 18.5 us, up 14.7 -> 11.8,
#endif                            // head 15.2 -> 12.5 at M = 512 against 2 (128 rows); profiles/r05_tall_perf.txt
constexpr int kTallMTW = LINA_TALL_MTW;   // 16-row m-tiles per wave
constexpr int kTallNWV = 4;       // waves per workgroup (rows per workgroup = 16 * MTW * NWV = 64 at MTW = 1)
#ifndef LINA_TALL_KB
#define LINA_TALL_KB 2            // (tools/tall_variants.sh builds other ring shapes for A/B)
#endif
#ifndef LINA_TALL_NS
#define LINA_TALL_NS 3
#endif
constexpr int kTallKB = LINA_TALL_KB;   // k-steps per LDS stage
constexpr int kTallNS = LINA_TALL_NS;   // stages in the LDS ring (NS - 1 of them in flight while one is consumed)
constexpr int kTallRows = 16 * kTallMTW * kTallNWV;
constexpr int kTallMinRows = 160; // the launchers' own rule: from M = 192 up the tall kernels win (M = 128: equal or slightly slower)
constexpr int kTallSlots = kTallMTW * kTallNWV + kTallNWV;    // fragments per k-step: 8 of A (two per wave) + 4 of W

// LDS of the ring: NS stages x KB k-steps x 12 fragments of 1 KiB (72 KiB: two workgroups per CU)
constexpr int tall_lds_bytes() { return kTallNS * kTallKB * kTallSlots * 1024; }

// one fragment (1 KiB) out of the staged image: lane-linear, 16 bytes per lane
template <typename T>
__device__ __forceinline__ void frag_from_lds(Frag<T>& f, const unsigned char* stage, int lane) {
    f.load(reinterpret_cast<const T*>(stage + 16 * lane));
}

// acc[g][mt] += A[rows of this wave's m-tile mt, :] . W[16-row block nb[g], :]^T over all nks k-steps.
//   A, W: fragment-major (packed) operands; mtile0: this wave's first 16-row block of A; wave_on: the block exists (a wave
//   beyond the padded batch streams block 0 instead -- the launch's operation counts stay uniform -- and its results are never
//   stored); s_w: tall_lds_bytes() of LDS, 16-byte aligned, the workgroup's ONLY LDS use while the loop runs.  G = 4 or 1
//   weight fragments per k-step (G = 1: the four waves all fetch that one fragment -- again uniform counts).  Every wave of
//   the workgroup must call this (barriers inside).
//
// BOTH operands travel global -> LDS by DMA (inline asm: the compiler neither waits for it nor counts it), in a ring of NS
// stages of KB k-steps: the first version of this loop kept A in a compiler-managed register double buffer beside a
// two-stage DMA of W, and every stage ended in `vmcnt(0)` -- one full memory round trip per 128 columns of K, 8 per workgroup
// at K = 1024: 24 us for the 4.3 GFLOP in-projection (profiles/r05_tall_v1_timeline.txt).  Here a wave issues 3 DMA pieces
// per k-step (its own two A fragments, one W fragment), waits with a COUNTED vmcnt for the stage it is about to consume
// (NS - 2 younger stages stay in flight), meets the others at a raw barrier (LDS counter only), refills the stage consumed
// last, and multiplies.  The last NS - 2 stages wait with vmcnt(0) (their successors may be short or absent).
template <typename T, int G, bool LN>
__device__ __forceinline__ void tall_core(const T* __restrict__ A, const T* __restrict__ W, const int (&nb)[G], int nks,
                                          int mtile0, bool wave_on, unsigned char* s_w, f32x4 (&acc)[G][kTallMTW],
                                          float (&rs1)[kTallMTW][4], float (&rs2)[kTallMTW][4]) {
    // rs1 / rs2 (LN): sum_k a and sum_k a^2 of the rows this lane holds OUTPUTS for (rows 4 lg + r of m-tile mt).  They come
    // off the matrix pipe -- A . 1 and the diagonal of A . A^T, two more MFMAs per m-tile and k-step, as in the 64-row kernels
    // -- not off the VALU: the SQ counters of the first version showed ten VALU instructions per MFMA and the matrix pipe
    // 7 % busy (profiles/r05_tall_sq.json); the per-lane dot products of the statistics were a fifth of them.
    using F = Frag<T>;
    constexpr int MTW = kTallMTW, NWV = kTallNWV, KB = kTallKB, NS = kTallNS, SL = kTallSlots;
    constexpr int OPS = KB * (MTW + 1);                     // DMA pieces per wave and (full) stage
    static_assert(G == 4 || G == 1, "one weight fragment per wave and k-step (G = 4), or the same one for all (G = 1)");
    const int lane = threadIdx.x & 63;
    const int w = wave_uniform(threadIdx.x >> 6);
    const int64_t fstr = 64 * F::KL;                        // elements per fragment
    const int mt_src = wave_on ? mtile0 : 0;
    const int wb = nb[G == 4 ? w : 0];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) acc[g][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 st1[MTW], st2[MTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) { st1[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; st2[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    F f_ones;
    f_ones.ones();
    // Full stages run in a loop whose body has NO data-dependent exit (the first version broke out of the unrolled k-step loop
    // of a partial stage: the compiler then kept the accumulators in AGPRs and copied them at every loop edge -- 91 of the 98
    // VALU instructions of a stage were v_accvgpr moves); a last partial stage (K not a multiple of KB k-steps) is handled after.
    const int nfull = nks / KB, rem = nks - nfull * KB;    // rem < KB k-steps in a last, partial stage
    const lds_addr_t lds0 = lds_addr_of(s_w);
    const unsigned my_a = (unsigned)(MTW * w) * 1024u, my_w = (unsigned)(MTW * NWV + w) * 1024u;
    const T* a_src = A + (int64_t)mt_src * nks * fstr;
    const T* w_src = W + (int64_t)wb * nks * fstr;

    auto issue_steps = [&](int st, int nu) {               // this wave's pieces of k-steps [st KB, st KB + nu) -> ring slot st % NS
        const lds_addr_t base = lds_addr_add(lds0, (unsigned)(st % NS) * (unsigned)(KB * SL * 1024));
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            if (u < nu) {                                   // (nu == KB in the loop: folded)
                const int64_t ks = (int64_t)st * KB + u;
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
                    dma16_to_lds_at(a_src + ((int64_t)mt * nks + ks) * fstr, 16u * (unsigned)lane,
                                    lds_addr_add(base, (unsigned)(u * SL + mt) * 1024u + my_a));
                dma16_to_lds_at(w_src + ks * fstr, 16u * (unsigned)lane, lds_addr_add(base, (unsigned)(u * SL) * 1024u + my_w));
            }
        }
    };
    auto issue = [&](int st) {
        if (st < nfull) issue_steps(st, KB);
        else if (st == nfull && rem) issue_steps(st, rem);
    };
    auto compute_steps = [&](int st, int nu) {
        const unsigned char* base = s_w + (st % NS) * (KB * SL * 1024);
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            if (u < nu) {
                F fa[MTW], fb[G];
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) frag_from_lds<T>(fa[mt], base + (u * SL + MTW * w + mt) * 1024, lane);
#pragma unroll
                for (int g = 0; g < G; ++g) frag_from_lds<T>(fb[g], base + (u * SL + MTW * NWV + g) * 1024, lane);
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) {
                    if (LN) {
                        st1[mt] = F::mma(fa[mt], f_ones, st1[mt]);       // row sums (every column)
                        st2[mt] = F::mma(fa[mt], fa[mt], st2[mt]);       // Gram matrix: the diagonal holds sum a^2
                    }
#pragma unroll
                    for (int g = 0; g < G; ++g) acc[g][mt] = F::mma(fa[mt], fb[g], acc[g][mt]);
                }
            }
        }
    };

#pragma unroll
    for (int p = 0; p < NS - 1; ++p) issue(p);
    for (int st = 0; st < nfull; ++st) {
        if (st + NS - 2 <= nfull - 1) wait_vmem_but<(NS - 2) * OPS>();   // stage st has landed; NS - 2 FULL stages stay in flight
        else wait_vmem();
        lds_barrier();                                      // ... everybody's pieces; and stage st - 1 has been consumed by all
        issue(st + NS - 1);                                 // -> into the slot of stage st - 1
        compute_steps(st, KB);
    }
    if (rem) {
        wait_vmem();
        lds_barrier();
        compute_steps(nfull, rem);
    }
    __syncthreads();                                        // the ring is free: callers reuse it for their epilogue exchange
    // D layout: lane (li, lg) holds column li, rows 4 lg + r.  A . 1 has the row sum in every column; the diagonal element
    // of row 4 lg + r sits in the lane with li = 4 lg + r, register r: one shuffle per row
    const int lg = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            rs1[mt][r] = LN ? st1[mt][r] : 0.f;
            rs2[mt][r] = LN ? shfl(st2[mt][r], 16 * lg + 4 * lg + r) : 0.f;
        }
}

// ---- variant 1: no LDS at all.  ONE wave per workgroup owns a 64 x (16 G) tile for the whole contraction and streams BOTH
// operands straight from L2 into VGPRs through a register ring D k-steps deep (plain loads: asynchronous until the compiler's
// counted wait in front of the MFMAs that consume them; an LDS-DMA piece blocks the issuing wave until the memory system has
// taken it, ~34 clocks per KiB, which is what kept the ring of variant 0 from overlapping its own refill with its MFMAs).
// 4 + G fragments per k-step feed 4 G MFMAs (variant 0: 3 DMA pieces + 6 LDS reads for 2 G); nothing is shared inside a
// workgroup, tiles that share an operand meet in L2 (the launchers map them to one XCD).
#ifndef LINA_TALL_DEFAULT_V
#define LINA_TALL_DEFAULT_V 0     // which variant the launchers pick (LINA_TALL_V overrides per call: test / A-B hook)
#endif
constexpr int kTallRegMT = 4;     // m-tiles of a variant-1 wave (64 rows)
#ifndef LINA_TALL_D
#define LINA_TALL_D 4
#endif
constexpr int kTallRegD = LINA_TALL_D;

template <typename T, int G, bool LN>
__device__ __forceinline__ void tall_core_reg(const T* __restrict__ A, const T* __restrict__ W, const int (&nb)[G], int nks,
                                              int mtile0, bool wave_on, f32x4 (&acc)[G][kTallRegMT], float (&rs1)[kTallRegMT][4],
                                              float (&rs2)[kTallRegMT][4]) {
    using F = Frag<T>;
    constexpr int MT = kTallRegMT, D = kTallRegD;
    float s1[MT], s2[MT];              // (this variant has no registers for two more accumulator tiles per m-tile: per-lane sums)
    const int lane = threadIdx.x & 63;
    const int64_t fstr = 64 * F::KL;
    const int mt_src = wave_on ? mtile0 : 0;
    const T* ap[MT];
    const T* wp[G];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) ap[mt] = A + ((int64_t)(mt_src + mt) * nks * 64 + lane) * F::KL;
#pragma unroll
    for (int g = 0; g < G; ++g) wp[g] = W + ((int64_t)nb[g] * nks * 64 + lane) * F::KL;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[g][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { s1[mt] = 0.f; s2[mt] = 0.f; }
    F fa[D][MT], fb[D][G];
    auto load = [&](int slot, int ks) {                     // (slot: a compile-time constant after unrolling)
        const int kc = ks < nks ? ks : nks - 1;             // past the end: a harmless re-read, never multiplied
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) fa[slot][mt].load(ap[mt] + (int64_t)kc * fstr);
#pragma unroll
        for (int g = 0; g < G; ++g) fb[slot][g].load(wp[g] + (int64_t)kc * fstr);
    };
#pragma unroll
    for (int d = 0; d < D - 1; ++d) load(d, d);
    for (int k0 = 0; k0 < nks; k0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            load((d + D - 1) % D, k0 + d + D - 1);           // refill the slot consumed one step ago
            if (k0 + d < nks) {                              // wave-uniform
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    if (LN) fa[d][mt].stats(s1[mt], s2[mt]);
#pragma unroll
                    for (int g = 0; g < G; ++g) acc[g][mt] = F::mma(fa[d][mt], fb[d][g], acc[g][mt]);
                }
            }
        }
    }
    // per-lane partial sums of the row the lane holds INPUTS for (A layout: row li) -> reduce over the four lane groups,
    // then one shuffle per OUTPUT row (D layout: rows 4 lg + r)
    const int lg = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        float a = s1[mt], b = s2[mt];
        a += shfl_xor(a, 16); b += shfl_xor(b, 16);
        a += shfl_xor(a, 32); b += shfl_xor(b, 32);
#pragma unroll
        for (int r = 0; r < 4; ++r) { rs1[mt][r] = shfl(a, 4 * lg + r); rs2[mt][r] = shfl(b, 4 * lg + r); }
    }
}

// ---- variant 2: the LDS ring carries only what the four waves SHARE (the weight fragments, by DMA); every wave's own A
// fragments go straight to its registers with plain loads, through a register ring of the same depth.  An LDS-DMA piece
// occupies the issuing wave until the memory system has taken it (K2's loader waves measured ~34 clocks per KiB, twice that
// beside LDS reads), and variant 0 pushes 24 KiB per stage and CU through that path: its stages take ~1800 clocks for 24
// MFMAs per wave whatever the ring depth (profiles/r05_tall_perf.txt).  Here the DMA path carries 8 KiB per stage; the A loads
// are asynchronous until they are used.  Same stage bookkeeping as variant 0: per wave and stage KB x (MTW loads + 1 DMA
// piece) vector-memory operations, so the counted wait is the same expression; the compiler adds its own (earlier) wait for
// the A registers -- it does not see the DMA pieces between its loads, so that wait also covers part of the next stage.
template <typename T, int G, bool LN>
__device__ __forceinline__ void tall_core_hyb(const T* __restrict__ A, const T* __restrict__ W, const int (&nb)[G], int nks,
                                              int mtile0, bool wave_on, unsigned char* s_w, f32x4 (&acc)[G][kTallMTW],
                                              float (&rs1)[kTallMTW][4], float (&rs2)[kTallMTW][4]) {
    using F = Frag<T>;
    constexpr int MTW = kTallMTW, NWV = kTallNWV, KB = kTallKB, NS = kTallNS;
    constexpr int OPS = KB * (MTW + 1);
    static_assert(G == 4 || G == 1, "one weight fragment per wave and k-step (G = 4), or the same one for all (G = 1)");
    static_assert(NS == 3, "the stage loop is unrolled by hand over a ring of three");
    const int lane = threadIdx.x & 63;
    const int w = wave_uniform(threadIdx.x >> 6);
    const int64_t fstr = 64 * F::KL;
    const int mt_src = wave_on ? mtile0 : 0;
    const int wb = nb[G == 4 ? w : 0];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) acc[g][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 st1[MTW], st2[MTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) { st1[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; st2[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    F f_ones;
    f_ones.ones();
    const int nfull = nks / KB, rem = nks - nfull * KB;
    const lds_addr_t lds0 = lds_addr_of(s_w);
    const T* a_src = A + ((int64_t)mt_src * nks * 64 + lane) * F::KL;
    const T* w_src = W + (int64_t)wb * nks * fstr;
    F fa[NS][KB][MTW];

    // stage st -> ring slot S (a compile-time constant: the register ring is indexed with it)
    auto issue = [&](auto slot, int st) {
        constexpr int S = decltype(slot)::value;
        const int nu = st < nfull ? KB : (st == nfull ? rem : 0);           // wave-uniform
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            if (u < nu) {
                const int64_t ks = (int64_t)st * KB + u;
                dma16_to_lds_at(w_src + ks * fstr, 16u * (unsigned)lane,
                                lds_addr_add(lds0, (unsigned)((S * KB + u) * NWV + w) * 1024u));
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) fa[S][u][mt].load(a_src + ((int64_t)mt * nks + ks) * fstr);
            }
        }
    };
    auto compute = [&](auto slot, int nu) {
        constexpr int S = decltype(slot)::value;
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            if (u < nu) {
                F fb[G];
#pragma unroll
                for (int g = 0; g < G; ++g) frag_from_lds<T>(fb[g], s_w + ((S * KB + u) * NWV + g) * 1024, lane);
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) {
                    if (LN) {
                        st1[mt] = F::mma(fa[S][u][mt], f_ones, st1[mt]);
                        st2[mt] = F::mma(fa[S][u][mt], fa[S][u][mt], st2[mt]);
                    }
#pragma unroll
                    for (int g = 0; g < G; ++g) acc[g][mt] = F::mma(fa[S][u][mt], fb[g], acc[g][mt]);
                }
            }
        }
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;
    auto body = [&](auto slot, auto refill, int st) {       // one FULL stage: wait, meet, refill the slot consumed last, multiply
        if (st + NS - 2 <= nfull - 1) wait_vmem_but<(NS - 2) * OPS>();
        else wait_vmem();
        lds_barrier();
        issue(refill, st + NS - 1);
        compute(slot, KB);
    };
    issue(I0{}, 0);
    issue(I1{}, 1);
    int st = 0;
    for (; st + NS <= nfull; st += NS) {
        body(I0{}, I2{}, st);
        body(I1{}, I0{}, st + 1);
        body(I2{}, I1{}, st + 2);
    }
    const int left = nfull - st;                            // 0, 1 or 2 full stages, then the partial one in slot `left`
    if (left >= 1) body(I0{}, I2{}, st);
    if (left >= 2) body(I1{}, I0{}, st + 1);
    if (rem) {
        wait_vmem();
        lds_barrier();
        if (left == 0) compute(I0{}, rem);
        else if (left == 1) compute(I1{}, rem);
        else compute(I2{}, rem);
    }
    __syncthreads();
    const int lg = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            rs1[mt][r] = LN ? st1[mt][r] : 0.f;
            rs2[mt][r] = LN ? shfl(st2[mt][r], 16 * lg + 4 * lg + r) : 0.f;
        }
}

// ---- variant 3 ("rw"): NO LDS-DMA.  Every wave's own A fragments go straight into a register ring D k-steps deep (plain 16-byte
// loads: asynchronous until the compiler's counted wait in front of the MFMAs that use them); the weight fragments the four waves
// share go global -> registers (the same ring) -> ds_write_b128 -> a two-slot LDS buffer of ONE k-step each, one barrier per
// k-step.  Round 6 measured every main loop of this file on its own (tools/micro/tall_gemm.hip, profiles/r06_tall_gemm.txt): a
// workgroup ingests ~30 bytes per clock whatever the transport (LDS-DMA ring, register ring, 4 or 8 waves per CU, ring depth 4-16,
// with or without the LayerNorm MFMAs) -- the CU's vector-memory path, not the matrix pipe or the LDS, sets the loop time, so the
// loop time is (bytes a CU pulls) / 30 B/clk and the tile shape that minimises it at M = 512 is 128 rows x 64 columns on exactly
// 256 workgroups.  At 128 rows (two m-tiles per wave) this loop takes 14.4 k clocks against 19.8 k for the LDS-DMA ring, whose
// issuing waves block 60-185 clocks per piece.  G = 5: a fifth weight fragment (the 16 low-rank gate rows of the fused
// in-projection) rides along, fetched in quarters (4 bytes per lane and wave) so that the four waves stay uniform.
// The loop is the ONLY code (no tail with its own register assignment: the accumulators then stay put instead of being
// permuted through v_accvgpr moves at the loop edge): nks % D == 0 and nks >= D are the CALLER's contract; the prefetch past the
// end re-reads the last k-step (clamped address, never multiplied) and the last LDS write fills a slot nobody reads.
// (element-wise: an aggregate copy out of the register ring makes the compiler keep the whole ring in scratch)
__device__ __forceinline__ void st16_lds(unsigned char* p, const uint4& v) { *reinterpret_cast<uint4*>(p) = make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void st16_lds(unsigned char* p, const float4& v) { *reinterpret_cast<float4*>(p) = make_float4(v.x, v.y, v.z, v.w); }
constexpr int kTallRwMT = 2;      // m-tiles per wave (32 rows; 128 rows per workgroup)
constexpr int kTallRwD = 8;       // k-steps in flight per wave
constexpr int tall_rw_lds_bytes(int G) { return 2 * G * 1024; }

// (A / W deliberately NOT __restrict__, here and in the calling kernels: loads from a noalias read-only pointer may be moved
// across the compiler fences below, and the compiler then sinks the prologue's A loads behind the first barrier -- two serial
// memory round trips before the first MFMA instead of one)
// ``prefetch()`` is called once, BEHIND the ring's first D k-steps of loads and in front of the loop: the place for the caller's
// epilogue operands (loads return in order: requested first they would hold up k-step 0, requested here they land while the
// first D k-steps are being multiplied).
template <typename T, int G, bool LN, bool WNT, int MT, int D, class Prefetch>
__device__ __forceinline__ void tall_core_rw(const T* A, const T* W, const int (&nb)[G], int nks,
                                             int mtile0, bool wave_on, unsigned char* s_w, f32x4 (&acc)[G][MT],
                                             float (&rs1)[MT][4], float (&rs2)[MT][4], Prefetch&& prefetch) {
    using F = Frag<T>;
    static_assert(G == 4 || G == 5, "four shared weight fragments per k-step (one per wave), optionally a fifth fetched in quarters");
    const int lane = threadIdx.x & 63;
    const int w = wave_uniform(threadIdx.x >> 6);
    const int64_t fstr = 64 * F::KL;                        // elements per fragment (1 KiB)
    const int mt_src = wave_on ? mtile0 : 0;
    const T* ap[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) ap[mt] = A + ((int64_t)(mt_src + mt) * nks * 64 + lane) * F::KL;
    const int wb = w == 0 ? nb[0] : w == 1 ? nb[1] : w == 2 ? nb[2] : nb[3];
    const T* wp = W + ((int64_t)wb * nks * 64 + lane) * F::KL;
    // quarter w of the fifth fragment: 4 bytes per lane
    const float* xp = reinterpret_cast<const float*>(W + (int64_t)nb[G - 1] * nks * fstr) + 64 * w + lane;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[g][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 st1[MT], st2[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { st1[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; st2[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    F f_ones;
    f_ones.ones();
    F fa[D][MT], fw[D];
    float fx[D];
    auto load = [&](int slot, int ks) {                     // (slot: a compile-time constant after unrolling)
        const int kc = ks < nks ? ks : nks - 1;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) fa[slot][mt].load(ap[mt] + (int64_t)kc * fstr);
        fw[slot].template load_stream<WNT>(wp + (int64_t)kc * fstr);
        if (G == 5) fx[slot] = WNT ? ld_nt1(xp + (int64_t)kc * 256) : xp[(int64_t)kc * 256];
    };
    auto put = [&](int slot, int lds_slot) {                // this wave's share of a k-step's weight fragments -> LDS
        st16_lds(s_w + (lds_slot * G + w) * 1024 + 16 * lane, fw[slot].v);
        if (G == 5) *reinterpret_cast<float*>(s_w + (lds_slot * G + 4) * 1024 + 256 * w + 4 * lane) = fx[slot];
    };
    auto step = [&](int d, int ks) {                        // one k-step; ring slot d == ks % D
        put((d + 1) % D, (ks + 1) & 1);
        F fb[G];
#pragma unroll
        for (int g = 0; g < G; ++g) frag_from_lds<T>(fb[g], s_w + ((ks & 1) * G + g) * 1024, lane);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (LN) {
                st1[mt] = F::mma(fa[d][mt], f_ones, st1[mt]);       // row sums (every column)
                st2[mt] = F::mma(fa[d][mt], fa[d][mt], st2[mt]);    // Gram matrix: the diagonal holds sum a^2
            }
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g][mt] = F::mma(fa[d][mt], fb[g], acc[g][mt]);
        }
        load(d, ks + D);                                    // refill the slot just consumed
        lds_barrier();                                      // slot (ks + 1) & 1 is complete; slot ks & 1 has been read by all
    };
#pragma unroll
    for (int d = 0; d < D; ++d) {
        load(d, d);
        cfence();                                           // in k order: the first LDS write waits for k-step 0 only, not for the whole ring
    }
    prefetch();
    cfence();
    put(0, 0);
    lds_barrier();
    for (int k0 = 0; k0 < nks; k0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) step(d, k0 + d);
    }
    __syncthreads();                                        // (also drains the prefetch past the end)
    const int lg = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            rs1[mt][r] = LN ? st1[mt][r] : 0.f;
            rs2[mt][r] = LN ? shfl(st2[mt][r], 16 * lg + 4 * lg + r) : 0.f;
        }
}

// shape of a tall workgroup by variant: 0 = LDS ring (4 waves x 32 rows), 1 = register ring (1 wave x 64 rows)
template <int V> struct TallShape {
    static constexpr int MTW = V == 1 ? kTallRegMT : kTallMTW, NWV = V == 1 ? 1 : kTallNWV, ROWS = 16 * MTW * NWV;
    // V = 1: only the gate tiles' 64 x 16 exchange; V = 2: the weight ring (and, after it, the gate tiles' 128 x 16 exchange)
    static constexpr int LDS = V == 1 ? 64 * 17 * 4 : V == 2 ? kTallNS * kTallKB * kTallNWV * 1024 : tall_lds_bytes();
};
// XCD-aware tile order: consecutive workgroup ids go round the 8 XCDs (each with its own L2), so the row blocks that share a
// weight tile must have ids that are EQUAL mod 8 to meet in one L2 -- with the natural (column, row) order of a 43- or
// 65-column grid every weight tile was fetched from the fabric by four XCDs (PMC: FETCH_SIZE 4.4 x the unique bytes, L2 hit
// rate 0.6; profiles/r05_tall_pmc.txt).  1-D grid of 8 * ceil(ncol / 8) * nrow ids: id -> (column block, row block) with the
// column fixed by (id mod 8, id / (8 nrow)) and the row blocks of a column on consecutive slots of that XCD.  Returns false
// for the padding ids (column >= ncol): the workgroup exits before any barrier.
__device__ __forceinline__ bool tall_tile_of(int id, int ncol, int nrow, int& col, int& row) {
    const int xcd = id & 7, j = id >> 3;
    col = (j / nrow) * 8 + xcd;
    row = j % nrow;
    return col < ncol;
}
static inline unsigned tall_grid(int ncol, int nrow) { return 8u * (unsigned)((ncol + 7) / 8) * (unsigned)nrow; }

template <int V, typename T, int G, bool LN, int MTW>
__device__ __forceinline__ void tall_core_v(const T* A, const T* W, const int (&nb)[G], int nks, int mtile0, bool wave_on,
                                            unsigned char* s_w, f32x4 (&acc)[G][MTW], float (&rs1)[MTW][4], float (&rs2)[MTW][4]) {
    if constexpr (V == 0) tall_core<T, G, LN>(A, W, nb, nks, mtile0, wave_on, s_w, acc, rs1, rs2);
    else if constexpr (V == 2) tall_core_hyb<T, G, LN>(A, W, nb, nks, mtile0, wave_on, s_w, acc, rs1, rs2);
    else tall_core_reg<T, G, LN>(A, W, nb, nks, mtile0, wave_on, acc, rs1, rs2);
}

// mean / reciprocal standard deviation of the lane's four output rows from their sums (rs1, rs2 of the cores above)
__device__ __forceinline__ void tall_row_stats(const float (&rs1)[4], const float (&rs2)[4], float inv_d, float eps,
                                               float (&mu)[4], float (&rstd)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        mu[r] = rs1[r] * inv_d;
        rstd[r] = rsqrtf(fmaxf(rs2[r] * inv_d - mu[r] * mu[r], 0.f) + eps);
    }
}

}  // namespace lina
