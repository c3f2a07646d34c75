// gla_inproj_window.hip -- the in-projection AND the windowed recurrent update (K1w + K5) of one GLA mixer at T = 1 in ONE launch.
//
// Why: the decode step is a chain of launches; the in-projection (reference model/gla.py:158-180 at T = 1) is latency-bound
// (~8 us: a launch, two memory round trips, a reduction, an epilogue) and K1w (model/gla.py:186-219) is bound by how fast a
// CU can ingest its 256 KiB of fp32 state (~13 us) -- but the state does not depend on the projection.  Here the K1w
// workgroups start streaming their state at launch while the in-projection workgroups run on the SAME CUs (both kinds are
// 512 threads x <= 128 VGPRs x <= 80 KB LDS: one of each fits a CU), and q | k | v | g | gk are handed over inside the launch:
//   producer (gla_inproj_body<PUB>): write-through 8-byte stores, every wave drains, one lane counts the tile in on arrive[head];
//   consumer (K1w): one lane polls arrive[head] (relaxed agent-scope loads, s_sleep), workgroup barrier that leaves the state
//   loads in flight, then relaxed agent-scope (sc1) loads of its 3 x 256 + 256 + 256 values
// (programming guide, Guideline 16, form R1: placement-independent, no fence).  The last K1w workgroup of a head to pass the
// wait re-arms the head's two words for the next launch.  The counters must be zero before the first launch.
//
// Residency: nothing here waits for a workgroup that might not be resident unless the whole grid fits the chip -- the entry
// point refuses B*H + tiles > 2 x CU count (the K1w workgroups come FIRST in the grid so that an empty chip gives every CU one
// of them; the in-projection workgroups take the second slot of the CUs).  Every spin is bounded: on a timeout the kernel sets
// sync[32] and carries on with whatever it finds (the host checks the word when it synchronises).
//
// K1w here is the 512-thread form of gla_decode_window.hip: wave w accumulates the SAME two row-group partials (2w, 2w + 1) in
// the same order, the partials meet in the same s_red layout and wave 0 finishes the head with the same code -- the outputs
// are bit-identical to lina_gla_decode_inproj_packed + lina_gla_decode_window.  Of a thread's 32 state vectors NPRE are
// requested at launch (registers: NPRE x 4 of the 128); the rest once the first 32 - NPRE have been consumed.
#include <lina_dev.h>
#include "lina_common.h"
#include "skinny_frag.h"
#ifdef LINA_IW_PROF
// tools-only build (tools/iw_prof.sh): wall-clock stamps (100 MHz) of thread 0 of every workgroup, [block][slot].  In-projection
// blocks: 0 entry, 2 first load round consumed, 3 main loop done, 4 reduction done, 5 tile handed over; K1w blocks: 0 entry,
// 1 prefetch issued, 2 this head's tiles seen, 3 bookkeeping done, 4 late loads issued, 5 state pass done, 6 end.  NOT in the product.
__device__ unsigned long long lina_iw_prof[1024 * 8];
#define LINA_SKINNY_PROF 1
#define IP_PROF(i, expr) do { if (threadIdx.x == 0) pr_[i] = wall_clock64(); } while (0)
#define IP_PROF_FLUSH() do { if (threadIdx.x == 0 && blockIdx.x < 1024) { pr_[5] = wall_clock64(); \
        for (int i_ = 0; i_ < 8; ++i_) lina_iw_prof[blockIdx.x * 8 + i_] = pr_[i_]; } } while (0)
#define KW_PROF(i) do { if (threadIdx.x == 0) kw_[i] = wall_clock64(); } while (0)
#define KW_FLUSH() do { if (threadIdx.x == 0 && blockIdx.x < 1024) for (int i_ = 0; i_ < 8; ++i_) lina_iw_prof[blockIdx.x * 8 + i_] = kw_[i_]; } while (0)
#else
#define KW_PROF(i) do { } while (0)
#define KW_FLUSH() do { } while (0)
#endif
#include "gla_inproj_body.h"
#include <stdlib.h>

namespace lina {

struct InprojWinArgs {
    // in-projection (lina_gla_decode_inproj_packed)
    const bf16_t* A; const bf16_t* W; const float* c1; const float* c2;
    const bf16_t* wq; const bf16_t* wk; const bf16_t* wv; bf16_t* cq; bf16_t* ck; bf16_t* cv;
    const bf16_t* w2; const bf16_t* b2; bf16_t* qkv; bf16_t* g_out; float* gk;
    int M, K, Kd, Vd; float ln_eps, inv_norm, clamp_min; int has_clamp;
    // K1w + K5 (lina_gla_decode_window)
    float* S; float* hist_k; float* hist_c; float* hist_v; const int64_t* step; const int64_t* origin;
    int window, H; float scale; const bf16_t* nw; float eps; bf16_t* og; int og_packed;
    int* sync;                   // [0, H): tiles arrived per head; [16, 16 + H): K1w workgroups past the wait; [32]: timeout flag
    int n_k1w, k1w_first, target;
    int delay_ticks;             // the K1w workgroups wait this many 10 ns ticks before they request their state (0 = at once)
};

constexpr int kFW = 8;            // window positions the fused form is built for
constexpr int kFD = 256;          // Dk = Dv
constexpr int kSpinMax = 1 << 16;   // polls (~0.1 s): the wait is tens of microseconds when all is well

// LDS of a K1w workgroup (floats)
struct alignas(16) K1wSmem {
    float q[kFD], e[kFD], a[kFW][4];
    float w[kFW][kFD];            // e^{c_j - c_s} k_s per row
    float v[kFW][kFD];
    float h1[kFW][kFD], h2[kFW][kFD];   // the window's c_s, k_s as loaded (consumed after the hand-off)
    float red[16 * kFD];
};

template <int NPRE, int PACE>
__device__ __forceinline__ void k1w512_body(K1wSmem& sm, const int bh, const InprojWinArgs& p) {
    constexpr int DK = kFD, DV = kFD, NTOT = 32, NLATE = NTOT - NPRE;
    static_assert(NPRE >= 16 && NPRE < 32 && NPRE % 2 == 0, "NPRE");
    const int tid = threadIdx.x, lane = tid & 63;
#ifdef LINA_IW_PROF
    unsigned long long kw_[8] = {};
#endif
    KW_PROF(0);
    const int w = wave_uniform(tid >> 6);
    const int H = p.H, b = bh / H, h = bh % H;
    const int64_t BH = p.n_k1w;
    const int j = (int)(p.step[0] - p.origin[0]) & (p.window - 1);        // window position: workgroup-uniform
    const bool write_back = j == p.window - 1;
    const bool row_thr = w < 4;                    // waves 0..3: lane = row (c_s, k_s); waves 4..7: lane = v column
    const int rc = tid & 255;                      // this thread's row / column
    const int64_t hoff = (int64_t)bh * DK + rc;

    // ---- the window's history (written by earlier launches): global -> LDS copies, no registers; a wave moves the 256 bytes
    // of its own 64 rows (c_s, k_s) or columns (v_s) of every past step.  Issued IN FRONT of the state (loads return in order).
    {
        const unsigned lane_b = 4u * (unsigned)lane;
        if (row_thr) {
#pragma unroll
            for (int s = 0; s < kFW; ++s)
                if (s < j) {                                             // workgroup-uniform
                    dma4_to_lds_async(p.hist_c + ((int64_t)s * BH + bh) * DK + 64 * w, lane_b, &sm.h1[s][64 * w]);
                    dma4_to_lds_async(p.hist_k + ((int64_t)s * BH + bh) * DK + 64 * w, lane_b, &sm.h2[s][64 * w]);
                }
        } else {
#pragma unroll
            for (int s = 0; s < kFW; ++s)
                if (s < j) dma4_to_lds_async(p.hist_v + ((int64_t)s * BH + bh) * DV + 64 * (w - 4), lane_b, &sm.v[s][64 * (w - 4)]);
        }
    }
    // ---- state: wave w owns the row groups (rb = w / 2, rg in {2 (w % 2), 2 (w % 2) + 1}) of gla_decode_window.hip, i.e. rows
    // rb 64 + rg + 4 i; vector n = 2 i + (rg & 1); lane = 4 columns
    if (p.delay_ticks > 0) {                       // let the in-projection's load rounds through before the stream saturates HBM
        const long long t_in = wall_clock64();
        while (wall_clock64() - t_in < p.delay_ticks) poll_sleep();
    }
    // (addresses as wave-uniform base + compile-time row offset + one 32-bit lane offset: no per-load address registers)
    const int row0 = (w >> 1) * 64 + 2 * (w & 1);
    float* const tile_w = p.S + ((int64_t)bh * DK + row0) * DV;           // wave-uniform
    const unsigned lane4 = 4u * (unsigned)lane;
    auto rel_of = [](int n) { return (n & 1) + 4 * (n >> 1); };           // row of vector n relative to row0
    auto row_of = [&](int n) { return row0 + rel_of(n); };
    // PACE > 0: in batches of PACE with at most two batches outstanding per wave.  All at once (PACE = 0) the 192 KiB queue of this
    // CU's memory pipe stands in front of every load of the in-projection workgroup beside it (7-8 us per round trip of a
    // latency-bound kernel: the one-launch form then LOSES 10 us per block); thinned, the stream still arrives at the CU's
    // ingest rate and a foreign load waits for <= 2 PACE KiB per wave.
    float4 St[NPRE];
#pragma unroll
    for (int n = 0; n < NPRE; ++n) {
        St[n] = ld_nt4((tile_w + rel_of(n) * DV) + lane4);
        if (PACE > 0 && (n + 1) % PACE == 0 && n + 1 < NPRE) wait_vmem_but<PACE>();
    }

    KW_PROF(1);
    // ---- wait for this head's tiles of the in-projection
    if (tid == 0) {
        int spins = 0;
        while (ld_agent_i32(p.sync + h) < p.target) {
            if (++spins > kSpinMax) { st_agent_i32(p.sync + 32, 1); break; }
            poll_sleep();
        }
    }
    KW_PROF(2);
    lds_barrier();                                 // (leaves the state loads in flight)

    const int64_t nqkv = 2 * p.Kd + p.Vd;
    // wave 0 finishes the head at the very end: its output-gate and norm-weight values are requested NOW (one round trip less
    // in the tail); 4 columns per lane
    uint2 gate_raw = make_uint2(0u, 0u), nw_raw = make_uint2(0u, 0u);
    if (w == 0) {
        gate_raw = ld_agent_u64(p.g_out + (int64_t)b * p.Vd + h * DV + 4 * lane);
        nw_raw = ld4_raw(p.nw + 4 * lane);
    }
    if (row_thr) {
        const float gj = ld_agent_f32(p.gk + (int64_t)b * p.Kd + h * DK + rc);
        const float kj = cvt1(ld_agent_u16(p.qkv + b * nqkv + p.Kd + h * DK + rc));
        const float qj = cvt1(ld_agent_u16(p.qkv + b * nqkv + h * DK + rc)) * p.scale;
        wait_vmem();                               // this wave's history copies have landed (and, in order, everything since)
        float cprev = 0.0f;
#pragma unroll
        for (int s = 0; s < kFW; ++s) cprev = (s == j - 1) ? sm.h1[s][rc] : cprev;
        const float cj = cprev + gj;
        p.hist_c[(int64_t)j * BH * DK + hoff] = cj;
        p.hist_k[(int64_t)j * BH * DK + hoff] = kj;
        sm.q[rc] = qj;
        sm.e[rc] = __expf(cj);
#pragma unroll
        for (int s = 0; s < kFW; ++s) {
            if (s <= j) {                                                // workgroup-uniform
                const float ws = s == j ? kj : __expf(cj - sm.h1[s][rc]) * sm.h2[s][rc];
                sm.w[s][rc] = ws;
                float a = qj * ws;                                       // <q (.) e^{c_j - c_s}, k_s> over this wave's 64 rows
                a += shfl_xor(a, 1); a += shfl_xor(a, 2); a += shfl_xor(a, 4);
                a += shfl_xor(a, 8); a += shfl_xor(a, 16); a += shfl_xor(a, 32);
                if (lane == 0) sm.a[s][w] = a;
            }
        }
    } else {
        const float vj = cvt1(ld_agent_u16(p.qkv + b * nqkv + 2 * p.Kd + h * DV + rc));
        wait_vmem();                               // the v history this wave copied is read by every wave after the barrier
        sm.v[j][rc] = vj;
        p.hist_v[((int64_t)j * BH + bh) * DV + rc] = vj;
    }
    lds_barrier();
    KW_PROF(3);

    // ---- the state pass: vectors n = 0 .. 31 in order (the order of gla_decode_window.hip's i loop per partial)
    float4 acc[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    auto consume = [&](float4& s4, int n) {        // one state vector: (write-back: update + store,) accumulate
        const int r = row_of(n);
        float4& ac = acc[n & 1];
        if (write_back) {
            const float d = sm.e[r];
            s4.x *= d; s4.y *= d; s4.z *= d; s4.w *= d;
#pragma unroll 1
            for (int s = 0; s <= j; ++s) {        // (not unrolled: eight v vectors at once do not fit beside the state)
                const float4 vv = *reinterpret_cast<const float4*>(&sm.v[s][4 * lane]);
                const float ws = sm.w[s][r];
                s4.x = fmaf(ws, vv.x, s4.x); s4.y = fmaf(ws, vv.y, s4.y); s4.z = fmaf(ws, vv.z, s4.z); s4.w = fmaf(ws, vv.w, s4.w);
            }
            int l4 = (int)lane4;
            opaque(l4);                           // (address formed here, not hoisted into -- and spilled from -- 64 registers)
            st_nt4((tile_w + rel_of(n) * DV) + (unsigned)l4, s4);
            const float qq = sm.q[r];
            ac.x = fmaf(qq, s4.x, ac.x); ac.y = fmaf(qq, s4.y, ac.y); ac.z = fmaf(qq, s4.z, ac.z); ac.w = fmaf(qq, s4.w, ac.w);
        } else {
            const float qe = sm.q[r] * sm.e[r];
            ac.x = fmaf(qe, s4.x, ac.x); ac.y = fmaf(qe, s4.y, ac.y); ac.z = fmaf(qe, s4.z, ac.z); ac.w = fmaf(qe, s4.w, ac.w);
        }
    };
#pragma unroll
    for (int n = 0; n < NLATE; ++n) consume(St[n], n);
#pragma unroll
    for (int n = 0; n < NLATE; ++n) {              // the rest, into the freed registers
        int l4 = (int)lane4;
        opaque(l4);
        St[n] = ld_nt4((tile_w + rel_of(NPRE + n) * DV) + (unsigned)l4);
    }
    KW_PROF(4);
#pragma unroll
    for (int n = NLATE; n < NPRE; ++n) consume(St[n], n);
#pragma unroll
    for (int n = 0; n < NLATE; ++n) consume(St[n], NPRE + n);
    KW_PROF(5);
    *reinterpret_cast<float4*>(&sm.red[(2 * w) * DV + 4 * lane]) = acc[0];
    *reinterpret_cast<float4*>(&sm.red[(2 * w + 1) * DV + 4 * lane]) = acc[1];
    __syncthreads();
    if (tid < 64) {
        // ---- wave 0 finishes the head as in gla_decode_window.hip: sum of the 16 row-group partials (+ the pending window
        // terms), then K5: RMS-normalise over Dv, weight, swish gate (reference model/gla.py:219)
        float4 r = *reinterpret_cast<const float4*>(&sm.red[4 * tid]);
#pragma unroll
        for (int jj = 1; jj < 16; ++jj) {
            const float4 t = *reinterpret_cast<const float4*>(&sm.red[jj * DV + 4 * tid]);
            r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w;
        }
        if (!write_back)
            for (int s = 0; s <= j; ++s) {
                float a = sm.a[s][0];
#pragma unroll
                for (int g2 = 1; g2 < 4; ++g2) a += sm.a[s][g2];
                const float4 vv = *reinterpret_cast<const float4*>(&sm.v[s][4 * tid]);
                r.x = fmaf(a, vv.x, r.x); r.y = fmaf(a, vv.y, r.y); r.z = fmaf(a, vv.z, r.z); r.w = fmaf(a, vv.w, r.w);
            }
        float ss = r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w;
        ss += shfl_xor(ss, 1); ss += shfl_xor(ss, 2); ss += shfl_xor(ss, 4);
        ss += shfl_xor(ss, 8); ss += shfl_xor(ss, 16); ss += shfl_xor(ss, 32);
        const float rs = rsqrtf(ss / (float)DV + p.eps);
        r.x *= rs; r.y *= rs; r.z *= rs; r.w *= rs;
        const float4 ww = cvt4(nw_raw);
        const float4 gg = cvt4(gate_raw);                          // this launch's output gate
        r.x *= ww.x; r.y *= ww.y; r.z *= ww.z; r.w *= ww.w;
        r.x *= gg.x * sigmoidf(gg.x); r.y *= gg.y * sigmoidf(gg.y);
        r.z *= gg.z * sigmoidf(gg.z); r.w *= gg.w * sigmoidf(gg.w);
        if (p.og_packed) st4(p.og + packed_off<bf16_t>(b, h * DV + 4 * tid, H * DV), r);
        else st4(p.og + (int64_t)bh * DV + 4 * tid, r);
    }
    // the last workgroup of the head to get here re-arms the head's words for the next launch (every other one has long read
    // `arrive` by then); at the END: the returning atomic is a memory-side round trip nobody should wait for
    if (tid == 64 * 7) {
        if (ticket_agent(p.sync + 16 + h) == p.n_k1w / H - 1) { st_agent_i32(p.sync + h, 0); st_agent_i32(p.sync + 16 + h, 0); }
    }
    KW_PROF(6);
    KW_FLUSH();
}

union InprojWinSmem {
    InprojSmem<2, 8> in;
    K1wSmem k1w;
};

template <int NPRE, int PACE, bool WNT>
__global__ __launch_bounds__(512, 4) void gla_inproj_window_kernel(const InprojWinArgs p) {
    __shared__ InprojWinSmem sm;
    const int bid = (int)blockIdx.x, n_in = (int)gridDim.x - p.n_k1w;
    const bool is_k1w = p.k1w_first ? bid < p.n_k1w : bid >= n_in;
    if (is_k1w) {
        k1w512_body<NPRE, PACE>(sm.k1w, p.k1w_first ? bid : bid - n_in, p);
    } else {
        gla_inproj_body<bf16_t, 2, true, WNT, 8, 2, true>(sm.in, p.k1w_first ? bid - p.n_k1w : bid, 0, p.A, 0, p.W, 0, p.c1, p.c2,
                                                           p.wq, p.wk, p.wv, p.cq, p.ck, p.cv, p.w2, p.b2, p.qkv, p.g_out, p.gk,
                                                           p.M, p.K, p.Kd, p.Vd, p.ln_eps, p.inv_norm, p.clamp_min, p.has_clamp,
                                                           p.H, p.sync);
    }
}

}  // namespace lina

extern "C" int lina_gla_decode_inproj_window(
    const void* x_packed, const void* w_in_packed, const float* c1, const float* c2, const void* wq, const void* wk,
    const void* wv, void* cq, void* ck, void* cv, const void* w2, const void* b2, void* qkv, void* g_out, float* gk,
    float* state, const void* norm_weight, void* og, float* hist_k, float* hist_c, float* hist_v, const int64_t* step,
    const int64_t* origin, int* sync, int window, int B, int K, int H, int Dk, int Dv, int W, int R, float ln_eps,
    float normalizer, float clamp_min, float eps, float scale, int og_packed, int w_stream, int n_pre, int pace, int dtype,
    lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(x_packed && w_in_packed && c1 && c2 && wq && wk && wv && cq && ck && cv && w2 && b2 && qkv && g_out && gk &&
                 state && norm_weight && og && hist_k && hist_c && hist_v && step && origin && sync,
                 "lina_gla_decode_inproj_window: null pointer");
    LINA_REQUIRE(B > 0 && K > 0 && H > 0, "lina_gla_decode_inproj_window: B, K, H must be positive");
    LINA_REQUIRE(normalizer != 0.0f, "lina_gla_decode_inproj_window: normalizer must be non-zero");
    LINA_REQUIRE(window >= 1 && window <= kFW && (window & (window - 1)) == 0,
                 "lina_gla_decode_inproj_window: window must be a power of two in [1, %d]", kFW);
    // what the one-launch form is built for; anything else runs as lina_gla_decode_inproj_packed + lina_gla_decode_window
    if (dtype != LINA_BF16 || W != 4 || R != 16 || Dk != kFD || Dv != kFD || B > 64 || H > 16 || K % 64 != 0)
        return fail(LINA_ERR_UNSUPPORTED, "lina_gla_decode_inproj_window: needs bf16, Dk = Dv = 256, B <= 64, H <= 16, conv width 4, "
                                          "gate rank 16 (got dtype %d, Dk %d, Dv %d, B %d, H %d, W %d, R %d)", dtype, Dk, Dv, B, H, W, R);
    const int Kd = H * Dk, Vd = H * Dv;
    const int n_in = (2 * Kd + 2 * Vd) / 32 + Kd / 16, n_k1w = B * H;
    int cus = 256;
#ifndef LINA_EMU
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return fail(LINA_ERR_LAUNCH, "lina_gla_decode_inproj_window: device query failed");
#endif
    if (n_k1w > cus || n_in + n_k1w > 2 * cus)     // every workgroup must be resident: the K1w workgroups wait inside the launch
        return fail(LINA_ERR_UNSUPPORTED, "lina_gla_decode_inproj_window: %d + %d workgroups do not fit %d CUs at two per CU",
                    n_k1w, n_in, cus);
    InprojWinArgs p;
    p.A = (const bf16_t*)x_packed; p.W = (const bf16_t*)w_in_packed; p.c1 = c1; p.c2 = c2;
    p.wq = (const bf16_t*)wq; p.wk = (const bf16_t*)wk; p.wv = (const bf16_t*)wv;
    p.cq = (bf16_t*)cq; p.ck = (bf16_t*)ck; p.cv = (bf16_t*)cv; p.w2 = (const bf16_t*)w2; p.b2 = (const bf16_t*)b2;
    p.qkv = (bf16_t*)qkv; p.g_out = (bf16_t*)g_out; p.gk = gk;
    p.M = B; p.K = K; p.Kd = Kd; p.Vd = Vd; p.ln_eps = ln_eps; p.inv_norm = 1.0f / normalizer; p.clamp_min = clamp_min;
    p.has_clamp = (clamp_min == clamp_min) ? 1 : 0;
    p.S = state; p.hist_k = hist_k; p.hist_c = hist_c; p.hist_v = hist_v; p.step = step; p.origin = origin;
    p.window = window; p.H = H; p.scale = scale; p.nw = (const bf16_t*)norm_weight; p.eps = eps; p.og = (bf16_t*)og;
    p.og_packed = og_packed; p.sync = sync; p.n_k1w = n_k1w;
    p.target = (2 * Dk + 2 * Dv) / 32 + Dk / 16;
    {
        const char* dl = getenv("LINA_IW_DELAY");                 // tuning knob (tools/probe_one_launch.py), 10 ns ticks; read per call
        p.delay_ticks = dl ? atoi(dl) : 0;
    }
#ifdef LINA_EMU
    p.k1w_first = 0;          // the emulator runs the workgroups one after the other in grid order: producers first
#else
    p.k1w_first = 1;
#endif
    const dim3 grid((unsigned)(n_in + n_k1w));
#define LINA_IW(NPREE, PACEE)                                                                                          \
    do {                                                                                                               \
        if (w_stream) LINA_LAUNCH((gla_inproj_window_kernel<NPREE, PACEE, true>), grid, dim3(512), 0, stream, p);      \
        else LINA_LAUNCH((gla_inproj_window_kernel<NPREE, PACEE, false>), grid, dim3(512), 0, stream, p);              \
    } while (0)
#define LINA_IW_P(NPREE)                                                                                               \
    do {                                                                                                               \
        if (pace == 0) LINA_IW(NPREE, 0); else if (pace == 2) LINA_IW(NPREE, 2); else LINA_IW(NPREE, 4);               \
    } while (0)
    if (pace < 0) pace = 4;
    if (pace != 0 && pace != 2 && pace != 4)
        return fail(LINA_ERR_ARG, "lina_gla_decode_inproj_window: pace must be 0, 2 or 4 (got %d)", pace);
    if (n_pre == 16) LINA_IW_P(16); else if (n_pre == 20) LINA_IW_P(20); else if (n_pre == 24 || n_pre <= 0) LINA_IW_P(24);
    else return fail(LINA_ERR_ARG, "lina_gla_decode_inproj_window: n_pre must be 16, 20 or 24 (got %d)", n_pre);
#undef LINA_IW_P
#undef LINA_IW
    return check_launch("lina_gla_decode_inproj_window");
}

#ifdef LINA_IW_PROF
extern "C" int lina_iw_prof_read(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(lina_iw_prof), sizeof(unsigned long long) * 1024 * 8);
}
#endif
