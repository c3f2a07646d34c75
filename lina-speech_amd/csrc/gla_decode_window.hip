// gla_decode_window.hip -- K1w: the decode-step recurrent update with a WINDOWED (lazily written) state.
//
// The recurrence of reference model/gla.py:186-213 at T = 1,  S_t = diag(e^{g_t}) S_{t-1} + k_t^T v_t,  o_t = q_t S_t,
// makes K1d stream the whole fp32 state in AND out of HBM for every token (2 x 67 MB per block at L169).  The output
// needs every element of S once per token -- the read cannot be avoided -- but not the write: with the chunk algebra of
// K2 applied to a window of W decode steps,
//     c_j   = g_0 + ... + g_j                                   (cumulative log-gate inside the window, <= 0)
//     S_j   = diag(e^{c_j}) S_base + sum_{s<=j} diag(e^{c_j - c_s}) k_s^T v_s
//     o_j   = (q_j (.) e^{c_j}) S_base + sum_{s<=j} <q_j (.) e^{c_j - c_s}, k_s> v_s
// S_base is only READ for j < W-1 and rewritten (S_base <- S_{W-1}) on the last step of the window.  Per token and
// (row, head) the kernel moves  4 Dk Dv (1 + 1/W)  bytes of state plus the window history
// (k_s, c_s: Dk floats, v_s: Dv floats per past step; ~5 % of the state bytes at W = 8) instead of  8 Dk Dv.
// Every exponent is a difference c_j - c_s <= 0 or c_j <= 0: reset gates (-20) are harmless.
//
// Work split: ONE workgroup per (b, h) with 256 threads per 64-row block of the state (Dk/64 <= 4 blocks: 1024 threads at
// Dk = 256); thread group rb streams the contiguous 64 x Dv block rb with non-temporal 16-byte accesses (all loads of the
// workgroup in flight at once: 256 KiB at L169), the partial outputs of the row groups meet in LDS and wave 0 finishes the
// head: sum, RMSNorm (x) swish gate (K5), store.  No inter-workgroup hand-off, no atomics, no drained stores in the tail
// (K1d+K5's row-block workgroups exchange partials through L2 with a ticket: a ~5 us dependent-latency tail that a
// read-only pass cannot hide behind its own writes).  The window position comes from a DEVICE step counter
// (j = (step - origin) mod W), so one captured hipGraph serves every position of the window.
#include <lina_dev.h>
#include "lina_common.h"
#include "skinny_frag.h"

#ifndef LINA_K1W_HIST_NT
#define LINA_K1W_HIST_NT 0     // experiment: window-history loads / stores with the non-temporal hint
#endif

// experiments (tools/probe_decode.py builds them into tools/abl/; the product build leaves both at 0)
#ifndef LINA_K1W_STATE_PLAIN
#define LINA_K1W_STATE_PLAIN 0  // state loads WITHOUT the non-temporal hint (is a cache-resident state any faster?)
#endif
#ifndef LINA_K1W_NO_TAIL_LOADS
#define LINA_K1W_NO_TAIL_LOADS 0  // WRONG RESULTS: the K5 tail without its norm-weight / gate loads (what does their latency cost?)
#endif
#if LINA_K1W_STATE_PLAIN
#define LINA_K1W_STATE_LOAD(p) (*reinterpret_cast<const float4*>(p))
#else
#define LINA_K1W_STATE_LOAD(p) ld_nt4(p)
#endif

namespace lina {

constexpr int kWinMax = 16;

__device__ __forceinline__ float ld_hist(const float* p) {
#if LINA_K1W_HIST_NT
    return ld_nt1(p);
#else
    return *p;
#endif
}
__device__ __forceinline__ void st_hist(float* p, float v) {
#if LINA_K1W_HIST_NT
    st_nt1(p, v);
#else
    *p = v;
#endif
}

// Four consecutive state elements of a row: fp32 (the product's default) or bf16 -- the OPT-IN state dtype of round 6: the
// reference keeps the recurrent state of a bf16 model in bf16 between decode steps (model/gla.py:229-240 `param.new_zeros` +
// Cache.update's copy_): every step upcasts it, updates in fp32 and rounds the result back.  With a bf16 state the arithmetic
// below is unchanged (fp32 registers); only what is read / written back differs -- at window 1 exactly the reference's
// per-step rounding, at window W a rounding every W-th step.
// A lane's piece of a state row is ONE 16-byte access whatever the state dtype: four fp32 or (round 6, late) EIGHT bf16 elements.
// (The first bf16-state build kept four elements per lane = 8-byte accesses and ran the read-only launch at 4.1 TB/s where the fp32
// state's 16-byte accesses reach 5.5: the CU's memory path is priced per instruction as much as per byte.)
template <typename TS> struct state_piece { static constexpr int n = 4; };
template <> struct state_piece<bf16_t> { static constexpr int n = 8; };
__device__ __forceinline__ void ld_state(const float* p, float (&x)[4]) {
    const float4 t = LINA_K1W_STATE_LOAD(p);
    x[0] = t.x; x[1] = t.y; x[2] = t.z; x[3] = t.w;
}
__device__ __forceinline__ void ld_state(const bf16_t* p, float (&x)[8]) {
    const uint4 u = ld_nt16(p);
    const float4 lo = cvt4(make_uint2(u.x, u.y)), hi = cvt4(make_uint2(u.z, u.w));
    x[0] = lo.x; x[1] = lo.y; x[2] = lo.z; x[3] = lo.w; x[4] = hi.x; x[5] = hi.y; x[6] = hi.z; x[7] = hi.w;
}
__device__ __forceinline__ void st_state(float* p, const float (&x)[4]) { st_nt4(p, make_float4(x[0], x[1], x[2], x[3])); }
__device__ __forceinline__ void st_state(bf16_t* p, const float (&x)[8]) {
    st_nt16(p, make_uint4(pack_bf16x2(x[0], x[1]), pack_bf16x2(x[2], x[3]), pack_bf16x2(x[4], x[5]), pack_bf16x2(x[6], x[7])));
}

template <int DV, int NRB, int CS, typename TIO, typename TG, typename TS = float>
__global__ __launch_bounds__(256 * NRB) void gla_decode_window_kernel(
    const TIO* __restrict__ q, const TIO* __restrict__ k, const TIO* __restrict__ v, const TG* __restrict__ gk, TS* S,
    float* hist_k, float* hist_c, float* hist_v, const int64_t* step, const int64_t* origin, int window, int flush_n,
    int H, int64_t q_sb, int64_t q_sh, int64_t k_sb, int64_t k_sh, int64_t v_sb, int64_t v_sh, int64_t g_sb,
    int64_t g_sh, float scale, const TIO* __restrict__ gate, int64_t gate_sb, int64_t gate_sh,
    const TIO* __restrict__ nw, float eps, TIO* __restrict__ og, int og_packed, float* o_x, int* counters) {
    // og_packed: og is written fragment-major (skinny_frag.h) as the [B, H*Dv] A operand of the output projection.
    // CS > 1 (Dv = CS * DV, e.g. expand_v = 2: Dv = 512): the head's columns are split over CS workgroups (blockIdx.y);
    // the recurrence is independent per column, only the RMS norm needs all of them: the halves' outputs meet in o_x
    // (fp32 [B*H][Dv], 8-byte agent-scope atomics + one ticket per head, as in K1d + K5) and the last arriver normalises.
    constexpr int DVT = DV * CS;      // the head's full value width
    constexpr int RB = 64;            // rows per thread group
    constexpr int DK = RB * NRB;
    constexpr int EPL = state_piece<TS>::n;   // state elements per lane and access (16 bytes)
    constexpr int CG = DV / EPL;      // lanes per row
    constexpr int RPI = 256 / CG;     // rows per pass of a thread group
    constexpr int NP = RB / RPI;      // 16-byte pieces per thread
    __shared__ float s_q[DK], s_e[DK], s_a[kWinMax][NRB];
    __shared__ float s_w[kWinMax][DK];                                   // e^{c_j - c_s} k_s per row
    __shared__ __attribute__((aligned(16))) float s_v[kWinMax][DV];
    __shared__ __attribute__((aligned(16))) float s_red[NRB * RPI * DV];

    const int tid = threadIdx.x;
    const int rb = tid >> 8, t256 = tid & 255;                          // thread group (row block), index inside it
    const int cg = t256 % CG, rg = t256 / CG;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int r0 = rb * RB;
    const int64_t BH = gridDim.x;
    const int col0 = CS > 1 ? (int)blockIdx.y * DV : 0;
    TS* tile = S + ((int64_t)bh * DK + r0) * DVT + col0 + EPL * cg;

    // window position: workgroup-uniform.  flush_n >= 0: apply the first flush_n history entries to the state, no output
    const bool flush_only = flush_n >= 0;
    const int j = flush_only ? flush_n - 1 : ((int)(step[0] - origin[0]) & (window - 1));   // window is a power of two
    const bool write_back = flush_only || j == window - 1;
    if (flush_only && flush_n == 0) return;

    // ---- small loads FIRST (a wave's loads return in order: issued behind the 64 KiB tile they would wait for it), and
    // ALL of the window's history entries at once (independent loads, one latency)
    const int n_hist = flush_only ? j + 1 : j;              // entries written by earlier launches
    // The first wave of each thread group owns the group's 64 rows (lane = row: c_s, k_s of the window in h1, h2); the
    // OTHER waves own the v columns (v_s in h1) -- disjoint waves, so the two histories share registers.
    constexpr int NV = 192 * NRB;                           // threads that are not in a row wave
    const bool row_wave = t256 < RB;
    float h1[kWinMax], h2[kWinMax];
    float gj = 0.f, kj = 0.f, qj = 0.f, vj = 0.f;
    const int row = r0 + (t256 & (RB - 1));
    const int64_t hoff = (int64_t)bh * DK + row;                         // [.][B*H][Dk]
    const int vc = rb * 192 + t256 - RB;                                 // v column of a non-row thread (first pass)
    if (row_wave) {
#pragma unroll
        for (int s = 0; s < kWinMax; ++s) {
            h1[s] = s < n_hist ? ld_hist(&hist_c[(int64_t)s * BH * DK + hoff]) : 0.0f;
            h2[s] = s < n_hist ? ld_hist(&hist_k[(int64_t)s * BH * DK + hoff]) : 0.0f;
        }
        if (!flush_only) {
            gj = ld(gk + b * g_sb + h * g_sh + row);
            kj = ld(k + b * k_sb + h * k_sh + row);
            qj = ld(q + b * q_sb + h * q_sh + row) * scale;
        }
    } else if (vc < DV) {
#pragma unroll
        for (int s = 0; s < kWinMax; ++s) h1[s] = s < n_hist ? ld_hist(&hist_v[((int64_t)s * BH + bh) * DVT + col0 + vc]) : 0.0f;
        if (!flush_only) vj = ld(v + b * v_sb + h * v_sh + col0 + vc);
    }

    float St[NP][EPL];
#pragma unroll
    for (int i = 0; i < NP; ++i) ld_state(tile + (int64_t)(rg + RPI * i) * DVT, St[i]);

    // ---- per-row gate bookkeeping and the window's v rows
    if (row_wave) {
        float cj;
        if (flush_only) {
            cj = h1[kWinMax - 1];
#pragma unroll
            for (int s = 0; s < kWinMax - 1; ++s) cj = (s == j) ? h1[s] : cj;
        } else {
            float cprev = 0.0f;
#pragma unroll
            for (int s = 0; s < kWinMax; ++s) cprev = (s == j - 1) ? h1[s] : cprev;
            cj = cprev + gj;
            if (CS == 1 || blockIdx.y == 0) {                            // the column halves compute the same values
                st_hist(&hist_c[(int64_t)j * BH * DK + hoff], cj);
                st_hist(&hist_k[(int64_t)j * BH * DK + hoff], kj);
            }
        }
        s_q[row] = qj;
        s_e[row] = __expf(cj);
#pragma unroll
        for (int s = 0; s < kWinMax; ++s) {
            if (s <= j) {                                                // workgroup-uniform
                const float ws = (!flush_only && s == j) ? kj : __expf(cj - h1[s]) * h2[s];
                s_w[s][row] = ws;
                float a = qj * ws;                                       // <q (.) e^{c_j - c_s}, k_s> over this row block
                a += shfl_xor(a, 1); a += shfl_xor(a, 2); a += shfl_xor(a, 4);
                a += shfl_xor(a, 8); a += shfl_xor(a, 16); a += shfl_xor(a, 32);
                if (t256 == 0) s_a[s][rb] = a;
            }
        }
    } else {
        if (vc < DV) {
#pragma unroll
            for (int s = 0; s < kWinMax; ++s)
                if (s <= j) s_v[s][vc] = (!flush_only && s == j) ? vj : h1[s];
            if (!flush_only) st_hist(&hist_v[((int64_t)j * BH + bh) * DVT + col0 + vc], vj);
        }
        for (int c = vc + NV; c < DV; c += NV) {            // only when Dv > 192 * Dk/64 (Dk = 64, Dv = 256)
            for (int s = 0; s <= j; ++s) {
                float vs;
                if (!flush_only && s == j) {
                    vs = ld(v + b * v_sb + h * v_sh + col0 + c);
                    hist_v[((int64_t)j * BH + bh) * DVT + col0 + c] = vs;
                } else {
                    vs = hist_v[((int64_t)s * BH + bh) * DVT + col0 + c];
                }
                s_v[s][c] = vs;
            }
        }
    }
    __syncthreads();

    float acc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = 0.f;
    if (write_back) {
        // S <- e^{c_j} S + sum_s w_s (x) v_s  (the window's rank-(j+1) update), o from the UPDATED rows
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const float d = s_e[r0 + rg + RPI * i];
#pragma unroll
            for (int e = 0; e < EPL; ++e) St[i][e] *= d;
        }
        for (int s = 0; s <= j; ++s) {
            float vv[EPL];
#pragma unroll
            for (int e4 = 0; e4 < EPL; e4 += 4) {
                const float4 t = *reinterpret_cast<const float4*>(&s_v[s][EPL * cg + e4]);
                vv[e4] = t.x; vv[e4 + 1] = t.y; vv[e4 + 2] = t.z; vv[e4 + 3] = t.w;
            }
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const float ws = s_w[s][r0 + rg + RPI * i];
#pragma unroll
                for (int e = 0; e < EPL; ++e) St[i][e] = fmaf(ws, vv[e], St[i][e]);
            }
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) st_state(tile + (int64_t)(rg + RPI * i) * DVT, St[i]);
        if (flush_only) return;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const float qq = s_q[r0 + rg + RPI * i];
#pragma unroll
            for (int e = 0; e < EPL; ++e) acc[e] = fmaf(qq, St[i][e], acc[e]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int rr = r0 + rg + RPI * i;
            const float qe = s_q[rr] * s_e[rr];
#pragma unroll
            for (int e = 0; e < EPL; ++e) acc[e] = fmaf(qe, St[i][e], acc[e]);
        }
    }
#pragma unroll
    for (int e4 = 0; e4 < EPL; e4 += 4)
        *reinterpret_cast<float4*>(&s_red[(rb * RPI + rg) * DV + EPL * cg + e4]) = make_float4(acc[e4], acc[e4 + 1], acc[e4 + 2], acc[e4 + 3]);
    __syncthreads();
    if (tid < 64) {
        // ---- wave 0 finishes the head: sum of the NRB*RPI row-group partials (+ the pending window terms), then K5:
        // RMS-normalise over Dv, weight, swish gate (reference model/gla.py:219)
        constexpr int CG4 = DV / 4;                               // the tail's lanes: four columns each, whatever EPL
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tid < CG4) {
            r = *reinterpret_cast<const float4*>(&s_red[4 * tid]);
#pragma unroll
            for (int jj = 1; jj < NRB * RPI; ++jj) {
                const float4 t = *reinterpret_cast<const float4*>(&s_red[jj * DV + 4 * tid]);
                r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w;
            }
            if (!write_back)            // pending window terms: sum_s <q (.) e^{c_j - c_s}, k_s> v_s
                for (int s = 0; s <= j; ++s) {
                    float a = s_a[s][0];
#pragma unroll
                    for (int g2 = 1; g2 < NRB; ++g2) a += s_a[s][g2];
                    const float4 vv = *reinterpret_cast<const float4*>(&s_v[s][4 * tid]);
                    r.x = fmaf(a, vv.x, r.x); r.y = fmaf(a, vv.y, r.y); r.z = fmaf(a, vv.z, r.z); r.w = fmaf(a, vv.w, r.w);
                }
        }
        float ss = r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w;
        ss += shfl_xor(ss, 1); ss += shfl_xor(ss, 2); ss += shfl_xor(ss, 4);
        ss += shfl_xor(ss, 8); ss += shfl_xor(ss, 16); ss += shfl_xor(ss, 32);
        // finish columns [c0, c0 + DV) of the head from the un-normalised values r (lane = 4 columns)
        auto finish = [&](float4 x, int c0, float rs) {
            if (tid < CG4) {
                x.x *= rs; x.y *= rs; x.z *= rs; x.w *= rs;
#if LINA_K1W_NO_TAIL_LOADS
                const float4 ww = make_float4(1.f, 1.f, 1.f, 1.f), gg = make_float4(eps, scale, eps, scale);
#else
                const float4 ww = ld4(nw + c0 + 4 * tid);
                const float4 gg = ld4(gate + b * gate_sb + h * gate_sh + c0 + 4 * tid);
#endif
                x.x *= ww.x; x.y *= ww.y; x.z *= ww.z; x.w *= ww.w;
                x.x *= gg.x * sigmoidf(gg.x); x.y *= gg.y * sigmoidf(gg.y);
                x.z *= gg.z * sigmoidf(gg.z); x.w *= gg.w * sigmoidf(gg.w);
                if (og_packed) st4(og + packed_off<TIO>(b, h * DVT + c0 + 4 * tid, H * DVT), x);   // 4 | KL: one piece
                else st4(og + (int64_t)bh * DVT + c0 + 4 * tid, x);
            }
        };
        if constexpr (CS == 1) {
            finish(r, 0, rsqrtf(ss / (float)DV + eps));
        } else {
            // publish this half, take a ticket; the LAST arriver reads the other halves and normalises the whole head
            float* ox = o_x + (int64_t)bh * DVT;
            if (tid < CG4) { st_agent8(ox + col0 + 4 * tid, r.x, r.y); st_agent8(ox + col0 + 4 * tid + 2, r.z, r.w); }
            drain_stores();
            int t = 0;
            if (tid == 0) t = ticket_agent(&counters[bh]);
            t = shfl_i(t, 0);
            if (t == CS - 1) {
                if (tid == 0) counters[bh] = 0;                     // re-armed for the next launch
                float4 oth[CS];
                float tot = ss;
#pragma unroll
                for (int c = 0; c < CS; ++c) {
                    oth[c] = r;
                    if (c != (int)blockIdx.y) {
                        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (tid < CG4) {
                            const float2 lo = ld_agent8(ox + c * DV + 4 * tid), hi = ld_agent8(ox + c * DV + 4 * tid + 2);
                            x = make_float4(lo.x, lo.y, hi.x, hi.y);
                        }
                        float s2 = x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
                        s2 += shfl_xor(s2, 1); s2 += shfl_xor(s2, 2); s2 += shfl_xor(s2, 4);
                        s2 += shfl_xor(s2, 8); s2 += shfl_xor(s2, 16); s2 += shfl_xor(s2, 32);
                        tot += s2;
                        oth[c] = x;
                    }
                }
                const float rs = rsqrtf(tot / (float)DVT + eps);
#pragma unroll
                for (int c = 0; c < CS; ++c) finish(oth[c], c * DV, rs);
            }
        }
    }
}

template <typename TIO, typename TG, typename TS = float>
static int launch_window(const void* q, const void* k, const void* v, const void* gk, TS* S, float* hk, float* hc,
                         float* hv, const int64_t* step, const int64_t* origin, int window, int flush_n, int B, int H,
                         int Dk, int Dv, const int64_t* st, float scale, lina_stream_t stream, const void* gate,
                         int64_t gate_sb, int64_t gate_sh, const void* nw, float eps, void* og, int og_packed = 0,
                         float* o_x = nullptr, int* counters = nullptr) {
    const int cs = Dv == 512 ? 2 : 1;                        // column splits: one workgroup streams <= 256 columns
    dim3 grid((unsigned)(B * H), (unsigned)cs);
#define LINA_WIN_ONE(DVV, NRBB, CSS)                                                                                   \
    LINA_LAUNCH((gla_decode_window_kernel<DVV, NRBB, CSS, TIO, TG, TS>), grid, dim3(256 * NRBB), 0, stream, (const TIO*)q, \
                (const TIO*)k, (const TIO*)v, (const TG*)gk, S, hk, hc, hv, step, origin, window, flush_n, H, st[0],   \
                st[1], st[2], st[3], st[4], st[5], st[6], st[7], scale, (const TIO*)gate, gate_sb, gate_sh,            \
                (const TIO*)nw, eps, (TIO*)og, og_packed, o_x, counters)
#define LINA_WIN_CASE(DVV)                                                                                             \
    case DVV:                                                                                                          \
        if (Dk == 64) LINA_WIN_ONE(DVV, 1, 1); else if (Dk == 128) LINA_WIN_ONE(DVV, 2, 1); else LINA_WIN_ONE(DVV, 4, 1); \
        break;
    if (Dk != 64 && Dk != 128 && Dk != 256)
        return fail(LINA_ERR_UNSUPPORTED, "lina_gla_decode_window: Dk=%d not in {64,128,256}", Dk);
    if (cs == 2 && flush_n < 0 && !(o_x && counters))
        return fail(LINA_ERR_ARG, "lina_gla_decode_window: Dv=512 needs the o_exchange buffer and the counters");
    switch (Dv) {
        LINA_WIN_CASE(64) LINA_WIN_CASE(128) LINA_WIN_CASE(256)
        case 512:
            if (Dk == 64) LINA_WIN_ONE(256, 1, 2); else if (Dk == 128) LINA_WIN_ONE(256, 2, 2); else LINA_WIN_ONE(256, 4, 2);
            break;
        default: return fail(LINA_ERR_UNSUPPORTED, "lina_gla_decode_window: Dv=%d not in {64,128,256,512}", Dv);
    }
#undef LINA_WIN_CASE
#undef LINA_WIN_ONE
    return check_launch("lina_gla_decode_window");
}

}  // namespace lina

extern "C" int lina_gla_decode_window_max(void) { return lina::kWinMax; }

extern "C" int lina_gla_decode_window(const void* q, const void* k, const void* v, const void* gk,
                                      float* state, const void* gate, const void* norm_weight, void* og,
                                      float* o_exchange, int* counters, float* hist_k, float* hist_c, float* hist_v, const int64_t* step,
                                      const int64_t* origin, int window, int B, int H, int Dk, int Dv, int64_t q_sb,
                                      int64_t q_sh, int64_t k_sb, int64_t k_sh, int64_t v_sb, int64_t v_sh, int64_t g_sb,
                                      int64_t g_sh, int64_t gate_sb, int64_t gate_sh, float eps, int og_packed, int dtype,
                                      int g_dtype, float scale, lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(q && k && v && gk && state && gate && norm_weight && og && hist_k && hist_c && hist_v && step && origin,
                 "lina_gla_decode_window: null pointer");
    LINA_REQUIRE(B > 0 && H > 0, "lina_gla_decode_window: B,H must be positive");
    LINA_REQUIRE(window >= 1 && window <= kWinMax && (window & (window - 1)) == 0,
                 "lina_gla_decode_window: window must be a power of two in [1, %d]", kWinMax);
    LINA_REQUIRE(valid_dtype(dtype) && valid_dtype(g_dtype), "lina_gla_decode_window: bad dtype enum");
    LINA_REQUIRE(gate_sb % 4 == 0 && gate_sh % 4 == 0, "lina_gla_decode_window: gate strides must be multiples of 4");
    const int64_t st[8] = {q_sb, q_sh, k_sb, k_sh, v_sb, v_sh, g_sb, g_sh};
    if (dtype == LINA_F32 && g_dtype == LINA_F32)
        return launch_window<float, float>(q, k, v, gk, state, hist_k, hist_c, hist_v, step, origin, window, -1, B, H,
                                           Dk, Dv, st, scale, stream, gate, gate_sb, gate_sh, norm_weight, eps, og, og_packed,
                                           o_exchange, counters);
    if (dtype == LINA_BF16 && g_dtype == LINA_F32)
        return launch_window<bf16_t, float>(q, k, v, gk, state, hist_k, hist_c, hist_v, step, origin, window, -1, B,
                                            H, Dk, Dv, st, scale, stream, gate, gate_sb, gate_sh, norm_weight, eps, og, og_packed,
                                           o_exchange, counters);
    if (dtype == LINA_BF16 && g_dtype == LINA_BF16)
        return launch_window<bf16_t, bf16_t>(q, k, v, gk, state, hist_k, hist_c, hist_v, step, origin, window, -1, B,
                                             H, Dk, Dv, st, scale, stream, gate, gate_sb, gate_sh, norm_weight, eps, og, og_packed,
                                           o_exchange, counters);
    return fail(LINA_ERR_UNSUPPORTED, "lina_gla_decode_window: dtype=f32 with bf16 gates is not built");
}

extern "C" int lina_gla_decode_window_flush(float* state, const float* hist_k, const float* hist_c, const float* hist_v,
                                            int n_pending, int B, int H, int Dk, int Dv, lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(state && hist_k && hist_c && hist_v, "lina_gla_decode_window_flush: null pointer");
    LINA_REQUIRE(B > 0 && H > 0, "lina_gla_decode_window_flush: B,H must be positive");
    LINA_REQUIRE(n_pending >= 0 && n_pending <= kWinMax, "lina_gla_decode_window_flush: n_pending must be in [0, %d]", kWinMax);
    if (n_pending == 0) return LINA_OK;
    const int64_t st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    return launch_window<float, float>(nullptr, nullptr, nullptr, nullptr, state, (float*)hist_k, (float*)hist_c,
                                       (float*)hist_v, nullptr, nullptr, kWinMax, n_pending, B, H, Dk, Dv, st, 1.0f, stream,
                                       nullptr, 0, 0, nullptr, 0.f, nullptr);
}

// ---- opt-in state dtype (round 6): the same two entry points with the dtype of `state` as an argument (LINA_F32: identical to
// the entries above; LINA_BF16: the reference's own arithmetic for a bf16 model at window 1, see ld_state4).  bf16 state is
// built for bf16 activations (dtype == LINA_BF16), gates in either dtype.
extern "C" int lina_gla_decode_window_s(const void* q, const void* k, const void* v, const void* gk, void* state, int state_dtype,
                                        const void* gate, const void* norm_weight, void* og, float* o_exchange, int* counters,
                                        float* hist_k, float* hist_c, float* hist_v, const int64_t* step, const int64_t* origin,
                                        int window, int B, int H, int Dk, int Dv, int64_t q_sb, int64_t q_sh, int64_t k_sb,
                                        int64_t k_sh, int64_t v_sb, int64_t v_sh, int64_t g_sb, int64_t g_sh, int64_t gate_sb,
                                        int64_t gate_sh, float eps, int og_packed, int dtype, int g_dtype, float scale,
                                        lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(valid_dtype(state_dtype), "lina_gla_decode_window_s: bad state dtype %d", state_dtype);
    if (state_dtype == LINA_F32)
        return lina_gla_decode_window(q, k, v, gk, (float*)state, gate, norm_weight, og, o_exchange, counters, hist_k, hist_c, hist_v,
                                      step, origin, window, B, H, Dk, Dv, q_sb, q_sh, k_sb, k_sh, v_sb, v_sh, g_sb, g_sh, gate_sb,
                                      gate_sh, eps, og_packed, dtype, g_dtype, scale, stream);
    LINA_REQUIRE(q && k && v && gk && state && gate && norm_weight && og && hist_k && hist_c && hist_v && step && origin,
                 "lina_gla_decode_window_s: null pointer");
    LINA_REQUIRE(B > 0 && H > 0, "lina_gla_decode_window_s: B,H must be positive");
    LINA_REQUIRE(window >= 1 && window <= kWinMax && (window & (window - 1)) == 0,
                 "lina_gla_decode_window_s: window must be a power of two in [1, %d]", kWinMax);
    LINA_REQUIRE(valid_dtype(dtype) && valid_dtype(g_dtype), "lina_gla_decode_window_s: bad dtype enum");
    LINA_REQUIRE(gate_sb % 4 == 0 && gate_sh % 4 == 0, "lina_gla_decode_window_s: gate strides must be multiples of 4");
    if (dtype != LINA_BF16) return fail(LINA_ERR_UNSUPPORTED, "lina_gla_decode_window_s: a bf16 state is built for bf16 activations");
    const int64_t st[8] = {q_sb, q_sh, k_sb, k_sh, v_sb, v_sh, g_sb, g_sh};
    if (g_dtype == LINA_F32)
        return launch_window<bf16_t, float, bf16_t>(q, k, v, gk, (bf16_t*)state, hist_k, hist_c, hist_v, step, origin, window, -1,
                                                    B, H, Dk, Dv, st, scale, stream, gate, gate_sb, gate_sh, norm_weight, eps, og,
                                                    og_packed, o_exchange, counters);
    return launch_window<bf16_t, bf16_t, bf16_t>(q, k, v, gk, (bf16_t*)state, hist_k, hist_c, hist_v, step, origin, window, -1, B,
                                                 H, Dk, Dv, st, scale, stream, gate, gate_sb, gate_sh, norm_weight, eps, og,
                                                 og_packed, o_exchange, counters);
}

extern "C" int lina_gla_decode_window_flush_s(void* state, int state_dtype, const float* hist_k, const float* hist_c,
                                              const float* hist_v, int n_pending, int B, int H, int Dk, int Dv,
                                              lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(valid_dtype(state_dtype), "lina_gla_decode_window_flush_s: bad state dtype %d", state_dtype);
    if (state_dtype == LINA_F32)
        return lina_gla_decode_window_flush((float*)state, hist_k, hist_c, hist_v, n_pending, B, H, Dk, Dv, stream);
    LINA_REQUIRE(state && hist_k && hist_c && hist_v, "lina_gla_decode_window_flush_s: null pointer");
    LINA_REQUIRE(B > 0 && H > 0, "lina_gla_decode_window_flush_s: B,H must be positive");
    LINA_REQUIRE(n_pending >= 0 && n_pending <= kWinMax, "lina_gla_decode_window_flush_s: n_pending must be in [0, %d]", kWinMax);
    if (n_pending == 0) return LINA_OK;
    const int64_t st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    return launch_window<bf16_t, float, bf16_t>(nullptr, nullptr, nullptr, nullptr, (bf16_t*)state, (float*)hist_k, (float*)hist_c,
                                                (float*)hist_v, nullptr, nullptr, kWinMax, n_pending, B, H, Dk, Dv, st, 1.0f,
                                                stream, nullptr, 0, 0, nullptr, 0.f, nullptr);
}
