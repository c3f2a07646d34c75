// gla_recurrent.hip -- K1: GLA recurrence with the (Dk x BV) fp32 state tile held in VGPRs.
//
// Replaces fla.ops.gla.fused_recurrent_gla / naive_recurrent_gla at the reference call sites
// model/gla.py:188,190,197,201.  T = 1 is the decode step, which is
// HBM-bound on the state: 8*Dk*Dv bytes (read + write) per (row, head) -- DESIGN.md.
//
// Work split: grid = (B*H, Dv/BV); one 256-thread workgroup owns S[b,h, 0:Dk, v0:v0+BV].
//   thread (rg = tid / CG, cg = tid % CG), CG = BV/4:  columns v0+4cg..+3 (one float4, so a
//   16-lane group reads 256 contiguous bytes of a state row), rows rg + RG*i, i < Dk/RG.
//   All Dk/RG float4 state loads of a thread are independent and issued up front, i.e. the
//   whole 64 KiB tile of a workgroup is in flight at once; <=128 VGPRs -> 4 workgroups per CU.
//   Per step: q*scale, k, exp(gk) (Dk values) and v (BV values) are staged in LDS, each thread
//   updates its rows and accumulates q.S for its 4 columns; the row-group partials are reduced
//   with two wave64 xor-shuffles (lanes +16, +32) and across the 4 waves through LDS.
#include <lina_dev.h>
#include "lina_common.h"

namespace lina {

template <int DK, int BV, typename TIO, typename TG>
__global__ __launch_bounds__(256) void gla_recurrent_kernel(
    const TIO* __restrict__ q, const TIO* __restrict__ k, const TIO* __restrict__ v,
    const TG* __restrict__ gk, TIO* __restrict__ o, const float* h0, float* ht,
    int H, int T, int Dv, lina_bht_strides sq, lina_bht_strides sk, lina_bht_strides sv,
    lina_bht_strides sg, lina_bht_strides so, float scale) {
    constexpr int CG = BV / 4;    // float4 column groups per row
    constexpr int RG = 256 / CG;  // row groups
    constexpr int NR = DK / RG;   // rows per thread
    constexpr int NW = 4;         // waves per workgroup
    static_assert(CG == 16 && DK % RG == 0, "tile shape");

    __shared__ float s_q[DK], s_k[DK], s_d[DK];
    __shared__ __attribute__((aligned(16))) float s_v[BV];
    __shared__ __attribute__((aligned(16))) float s_red[NW][BV];

    const int tid = threadIdx.x;
    const int cg = tid % CG, rg = tid / CG;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int v0 = blockIdx.y * BV;

    const int64_t tile = ((int64_t)bh * DK) * Dv + v0 + 4 * cg;
    float4 S[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int row = rg + RG * i;
        S[i] = h0 ? *reinterpret_cast<const float4*>(h0 + tile + (int64_t)row * Dv) : make_float4(0.f, 0.f, 0.f, 0.f);
    }

    const TIO* qb = q + b * sq.b + h * sq.h;
    const TIO* kb = k + b * sk.b + h * sk.h;
    const TG* gb = gk + b * sg.b + h * sg.h;
    const TIO* vb = v + b * sv.b + h * sv.h + v0;
    TIO* ob = o + b * so.b + h * so.h + v0;

    for (int t = 0; t < T; ++t) {
        for (int c = tid; c < DK; c += 256) {
            s_q[c] = ld(qb + t * sq.t + c) * scale;
            s_k[c] = ld(kb + t * sk.t + c);
            s_d[c] = expf(ld(gb + t * sg.t + c));
        }
        if (tid < BV) s_v[tid] = ld(vb + t * sv.t + tid);
        __syncthreads();

        const float4 vv = *reinterpret_cast<const float4*>(&s_v[4 * cg]);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int row = rg + RG * i;
            const float d = s_d[row], kk = s_k[row], qq = s_q[row];
            S[i].x = fmaf(S[i].x, d, kk * vv.x);
            S[i].y = fmaf(S[i].y, d, kk * vv.y);
            S[i].z = fmaf(S[i].z, d, kk * vv.z);
            S[i].w = fmaf(S[i].w, d, kk * vv.w);
            acc.x = fmaf(qq, S[i].x, acc.x);
            acc.y = fmaf(qq, S[i].y, acc.y);
            acc.z = fmaf(qq, S[i].z, acc.z);
            acc.w = fmaf(qq, S[i].w, acc.w);
        }
        // reduce over the 4 row groups that share a wave (lanes l, l^16, l^32, l^48)
        acc.x += shfl_xor(acc.x, 16); acc.y += shfl_xor(acc.y, 16);
        acc.z += shfl_xor(acc.z, 16); acc.w += shfl_xor(acc.w, 16);
        acc.x += shfl_xor(acc.x, 32); acc.y += shfl_xor(acc.y, 32);
        acc.z += shfl_xor(acc.z, 32); acc.w += shfl_xor(acc.w, 32);
        if ((tid & 63) < CG) *reinterpret_cast<float4*>(&s_red[tid >> 6][4 * cg]) = acc;
        __syncthreads();
        if (tid < BV) {
            const float r = (s_red[0][tid] + s_red[1][tid]) + (s_red[2][tid] + s_red[3][tid]);
            st(ob + t * so.t + tid, r);
        }
    }

    if (ht) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int row = rg + RG * i;
            *reinterpret_cast<float4*>(ht + tile + (int64_t)row * Dv) = S[i];
        }
    }
}

template <int DK, typename TIO, typename TG>
static int launch_recurrent(const void* q, const void* k, const void* v, const void* gk, void* o,
                            const float* h0, float* ht, int B, int H, int T, int Dv,
                            lina_bht_strides sq, lina_bht_strides sk, lina_bht_strides sv,
                            lina_bht_strides sg, lina_bht_strides so, float scale, lina_stream_t stream) {
    constexpr int BV = 64;
    dim3 grid((unsigned)(B * H), (unsigned)(Dv / BV));
    LINA_LAUNCH((gla_recurrent_kernel<DK, BV, TIO, TG>), grid, dim3(256), 0, stream,
                (const TIO*)q, (const TIO*)k, (const TIO*)v, (const TG*)gk, (TIO*)o, h0, ht,
                H, T, Dv, sq, sk, sv, sg, so, scale);
    return check_launch("lina_gla_recurrent_fwd");
}

template <typename TIO, typename TG>
static int dispatch_dk(int Dk, const void* q, const void* k, const void* v, const void* gk, void* o,
                       const float* h0, float* ht, int B, int H, int T, int Dv,
                       lina_bht_strides sq, lina_bht_strides sk, lina_bht_strides sv,
                       lina_bht_strides sg, lina_bht_strides so, float scale, lina_stream_t stream) {
    switch (Dk) {
        case 64: return launch_recurrent<64, TIO, TG>(q, k, v, gk, o, h0, ht, B, H, T, Dv, sq, sk, sv, sg, so, scale, stream);
        case 128: return launch_recurrent<128, TIO, TG>(q, k, v, gk, o, h0, ht, B, H, T, Dv, sq, sk, sv, sg, so, scale, stream);
        case 256: return launch_recurrent<256, TIO, TG>(q, k, v, gk, o, h0, ht, B, H, T, Dv, sq, sk, sv, sg, so, scale, stream);
    }
    return fail(LINA_ERR_UNSUPPORTED, "lina_gla_recurrent_fwd: Dk=%d not in {64,128,256}", Dk);
}

// shared argument validation of K1 / K2
int check_gla_args(const char* fn, const void* q, const void* k, const void* v, const void* gk, const void* o,
                   int B, int H, int T, int Dk, int Dv, int dtype, int g_dtype) {
    if (!q || !k || !v || !gk || !o) return fail(LINA_ERR_ARG, "%s: null tensor pointer", fn);
    if (B <= 0 || H <= 0 || T <= 0) return fail(LINA_ERR_ARG, "%s: B,H,T must be positive (got %d,%d,%d)", fn, B, H, T);
    if (!valid_dtype(dtype) || !valid_dtype(g_dtype)) return fail(LINA_ERR_ARG, "%s: bad dtype enum", fn);
    if (Dk != 64 && Dk != 128 && Dk != 256) return fail(LINA_ERR_UNSUPPORTED, "%s: Dk=%d not in {64,128,256}", fn, Dk);
    if (Dv <= 0 || Dv % 64 != 0) return fail(LINA_ERR_UNSUPPORTED, "%s: Dv=%d must be a positive multiple of 64", fn, Dv);
    return LINA_OK;
}

}  // namespace lina

extern "C" int lina_gla_recurrent_fwd(const void* q, const void* k, const void* v, const void* gk, void* o,
                                      const float* h0, float* ht, int B, int H, int T, int Dk, int Dv,
                                      lina_bht_strides sq, lina_bht_strides sk, lina_bht_strides sv,
                                      lina_bht_strides sg, lina_bht_strides so,
                                      int dtype, int g_dtype, float scale, lina_stream_t stream) {
    using namespace lina;
    int rc = check_gla_args("lina_gla_recurrent_fwd", q, k, v, gk, o, B, H, T, Dk, Dv, dtype, g_dtype);
    if (rc) return rc;
    if (dtype == LINA_F32 && g_dtype == LINA_F32)
        return dispatch_dk<float, float>(Dk, q, k, v, gk, o, h0, ht, B, H, T, Dv, sq, sk, sv, sg, so, scale, stream);
    if (dtype == LINA_BF16 && g_dtype == LINA_BF16)
        return dispatch_dk<bf16_t, bf16_t>(Dk, q, k, v, gk, o, h0, ht, B, H, T, Dv, sq, sk, sv, sg, so, scale, stream);
    if (dtype == LINA_BF16 && g_dtype == LINA_F32)
        return dispatch_dk<bf16_t, float>(Dk, q, k, v, gk, o, h0, ht, B, H, T, Dv, sq, sk, sv, sg, so, scale, stream);
    return fail(LINA_ERR_UNSUPPORTED, "lina_gla_recurrent_fwd: dtype=f32 with bf16 gates is not built");
}

// ================================================================================================
// K1d -- decode-step state update, ROW-split: one workgroup owns 64 full state rows of one (b,h),
// i.e. a CONTIGUOUS 64*Dv*4-byte block of HBM that it streams in once and out once with
// non-temporal 16-byte accesses (a wave instruction covers whole 1 KiB rows).  The q.S products of
// the Dk/64 row blocks are written as fp32 partials o_part[Dk/64][B*H][Dv]; the norm-gate kernel K5
// adds them (n_partial) -- no atomics, deterministic.  T = 1 only.
// ================================================================================================
namespace lina {

template <int DV, typename TIO, typename TG, bool FUSE>
__global__ __launch_bounds__(256) void gla_decode_rowsplit_kernel(
    const TIO* __restrict__ q, const TIO* __restrict__ k, const TIO* __restrict__ v, const TG* __restrict__ gk,
    float* o_part, float* S, int H, int Dk, int64_t q_sb, int64_t q_sh, int64_t k_sb, int64_t k_sh,
    int64_t v_sb, int64_t v_sh, int64_t g_sb, int64_t g_sh, float scale,
    const TIO* __restrict__ gate, int64_t gate_sb, int64_t gate_sh, const TIO* __restrict__ nw, float eps,
    TIO* __restrict__ og, int* counters) {
    constexpr int RB = 64;            // rows per workgroup
    constexpr int CG = DV / 4;        // lanes per row
    constexpr int RPI = 256 / CG;     // rows per pass of the workgroup
    constexpr int NP = RB / RPI;      // float4 per thread
    __shared__ float s_q[RB], s_k[RB], s_d[RB];
    __shared__ __attribute__((aligned(16))) float s_v[DV];
    __shared__ __attribute__((aligned(16))) float s_red[RPI * DV];

    const int tid = threadIdx.x;
    const int cg = tid % CG, rg = tid / CG;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int r0 = blockIdx.y * RB;
    float* tile = S + ((int64_t)bh * Dk + r0) * DV + 4 * cg;

    float4 St[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) St[i] = ld_nt4(tile + (int64_t)(rg + RPI * i) * DV);

    if (tid < RB) {
        const int c = r0 + tid;
        s_q[tid] = ld(q + b * q_sb + h * q_sh + c) * scale;
        s_k[tid] = ld(k + b * k_sb + h * k_sh + c);
        s_d[tid] = expf(ld(gk + b * g_sb + h * g_sh + c));
    }
    for (int c = tid; c < DV; c += 256) s_v[c] = ld(v + b * v_sb + h * v_sh + c);
    __syncthreads();

    const float4 vv = *reinterpret_cast<const float4*>(&s_v[4 * cg]);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int row = rg + RPI * i;
        const float d = s_d[row], kk = s_k[row], qq = s_q[row];
        St[i].x = fmaf(St[i].x, d, kk * vv.x);
        St[i].y = fmaf(St[i].y, d, kk * vv.y);
        St[i].z = fmaf(St[i].z, d, kk * vv.z);
        St[i].w = fmaf(St[i].w, d, kk * vv.w);
        acc.x = fmaf(qq, St[i].x, acc.x);
        acc.y = fmaf(qq, St[i].y, acc.y);
        acc.z = fmaf(qq, St[i].z, acc.z);
        acc.w = fmaf(qq, St[i].w, acc.w);
    }
    *reinterpret_cast<float4*>(&s_red[rg * DV + 4 * cg]) = acc;
    __syncthreads();
    const int64_t BH = gridDim.x;
    if (tid < CG) {                     // wave 0 (CG <= 64): this row block's partial q.S for 4 columns
        float4 r = *reinterpret_cast<const float4*>(&s_red[4 * tid]);
#pragma unroll
        for (int j = 1; j < RPI; ++j) {
            const float4 t = *reinterpret_cast<const float4*>(&s_red[j * DV + 4 * tid]);
            r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w;
        }
        float* op = o_part + ((int64_t)blockIdx.y * BH + bh) * DV + 4 * tid;
        if (FUSE) { st_agent8(op, r.x, r.y); st_agent8(op + 2, r.z, r.w); }
        else *reinterpret_cast<float4*>(op) = r;
    }
    if (FUSE && tid < 64) {
        // ---- K5 fused: the LAST of this head's Dk/64 row blocks adds the partials, RMS-normalises over Dv and
        // applies weight and swish gate.  Hand-off = 8-byte agent-scope atomics on both sides (lina_dev.h); the
        // state stores below are issued only afterwards so that this drain waits for 1 KiB, not for 64 KiB.
        drain_stores();
        int t = 0;
        if (tid == 0) t = ticket_agent(&counters[bh]);
        t = shfl_i(t, 0);
        if (t == (int)gridDim.y - 1) {
            if (tid == 0) counters[bh] = 0;                     // re-armed for the next launch
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tid < CG) {
                const float* xp = o_part + (int64_t)bh * DV + 4 * tid;
                for (int p = 0; p < (int)gridDim.y; ++p) {      // same order as K5's n_partial loop
                    const float2 lo = ld_agent8(xp + (int64_t)p * BH * DV), hi = ld_agent8(xp + (int64_t)p * BH * DV + 2);
                    if (p == 0) a = make_float4(lo.x, lo.y, hi.x, hi.y);
                    else { a.x += lo.x; a.y += lo.y; a.z += hi.x; a.w += hi.y; }
                }
            }
            float ss = 0.0f;
            ss += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
            ss += shfl_xor(ss, 1); ss += shfl_xor(ss, 2); ss += shfl_xor(ss, 4);
            ss += shfl_xor(ss, 8); ss += shfl_xor(ss, 16); ss += shfl_xor(ss, 32);
            const float rs = rsqrtf(ss / (float)DV + eps);
            if (tid < CG) {
                a.x *= rs; a.y *= rs; a.z *= rs; a.w *= rs;
                const float4 ww = ld4(nw + 4 * tid);
                a.x *= ww.x; a.y *= ww.y; a.z *= ww.z; a.w *= ww.w;
                const float4 gg = ld4(gate + b * gate_sb + h * gate_sh + 4 * tid);
                a.x *= gg.x * sigmoidf(gg.x); a.y *= gg.y * sigmoidf(gg.y);
                a.z *= gg.z * sigmoidf(gg.z); a.w *= gg.w * sigmoidf(gg.w);
                st4(og + (int64_t)bh * DV + 4 * tid, a);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) st_nt4(tile + (int64_t)(rg + RPI * i) * DV, St[i]);
}

template <typename TIO, typename TG>
static int launch_rowsplit(const void* q, const void* k, const void* v, const void* gk, float* o_part, float* S,
                           int B, int H, int Dk, int Dv, const int64_t* st, float scale, lina_stream_t stream,
                           const void* gate = nullptr, int64_t gate_sb = 0, int64_t gate_sh = 0,
                           const void* nw = nullptr, float eps = 0.f, void* og = nullptr, int* counters = nullptr) {
    dim3 grid((unsigned)(B * H), (unsigned)(Dk / 64));
    const bool fuse = og != nullptr;
#define LINA_RS_CASE(DVV)                                                                                            \
    case DVV:                                                                                                        \
        if (fuse)                                                                                                    \
            LINA_LAUNCH((gla_decode_rowsplit_kernel<DVV, TIO, TG, true>), grid, dim3(256), 0, stream, (const TIO*)q, \
                        (const TIO*)k, (const TIO*)v, (const TG*)gk, o_part, S, H, Dk, st[0], st[1], st[2], st[3],   \
                        st[4], st[5], st[6], st[7], scale, (const TIO*)gate, gate_sb, gate_sh, (const TIO*)nw, eps,  \
                        (TIO*)og, counters);                                                                         \
        else                                                                                                         \
            LINA_LAUNCH((gla_decode_rowsplit_kernel<DVV, TIO, TG, false>), grid, dim3(256), 0, stream, (const TIO*)q, \
                        (const TIO*)k, (const TIO*)v, (const TG*)gk, o_part, S, H, Dk, st[0], st[1], st[2], st[3],   \
                        st[4], st[5], st[6], st[7], scale, (const TIO*)nullptr, (int64_t)0, (int64_t)0,              \
                        (const TIO*)nullptr, 0.f, (TIO*)nullptr, (int*)nullptr);                                     \
        break;
    switch (Dv) {
        LINA_RS_CASE(64) LINA_RS_CASE(128) LINA_RS_CASE(256)
        default: return fail(LINA_ERR_UNSUPPORTED, "lina_gla_decode_update: Dv=%d not in {64,128,256}", Dv);
    }
#undef LINA_RS_CASE
    return check_launch("lina_gla_decode_update");
}

}  // namespace lina

extern "C" int lina_gla_decode_update(const void* q, const void* k, const void* v, const void* gk, float* o_part,
                                      float* state, int B, int H, int Dk, int Dv, int64_t q_sb, int64_t q_sh,
                                      int64_t k_sb, int64_t k_sh, int64_t v_sb, int64_t v_sh, int64_t g_sb,
                                      int64_t g_sh, int dtype, int g_dtype, float scale, lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(q && k && v && gk && o_part && state, "lina_gla_decode_update: null pointer");
    LINA_REQUIRE(B > 0 && H > 0, "lina_gla_decode_update: B,H must be positive");
    LINA_REQUIRE(valid_dtype(dtype) && valid_dtype(g_dtype), "lina_gla_decode_update: bad dtype enum");
    if (Dk <= 0 || Dk % 64 != 0) return fail(LINA_ERR_UNSUPPORTED, "lina_gla_decode_update: Dk=%d must be a multiple of 64", Dk);
    const int64_t st[8] = {q_sb, q_sh, k_sb, k_sh, v_sb, v_sh, g_sb, g_sh};
    if (dtype == LINA_F32 && g_dtype == LINA_F32)
        return launch_rowsplit<float, float>(q, k, v, gk, o_part, state, B, H, Dk, Dv, st, scale, stream);
    if (dtype == LINA_BF16 && g_dtype == LINA_F32)
        return launch_rowsplit<bf16_t, float>(q, k, v, gk, o_part, state, B, H, Dk, Dv, st, scale, stream);
    if (dtype == LINA_BF16 && g_dtype == LINA_BF16)
        return launch_rowsplit<bf16_t, bf16_t>(q, k, v, gk, o_part, state, B, H, Dk, Dv, st, scale, stream);
    return fail(LINA_ERR_UNSUPPORTED, "lina_gla_decode_update: dtype=f32 with bf16 gates is not built");
}

extern "C" int lina_gla_decode_update_norm(const void* q, const void* k, const void* v, const void* gk, float* o_part,
                                           float* state, const void* gate, const void* norm_weight, void* og,
                                           int* counters, int B, int H, int Dk, int Dv, int64_t q_sb, int64_t q_sh,
                                           int64_t k_sb, int64_t k_sh, int64_t v_sb, int64_t v_sh, int64_t g_sb,
                                           int64_t g_sh, int64_t gate_sb, int64_t gate_sh, float eps, int dtype,
                                           int g_dtype, float scale, lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(q && k && v && gk && o_part && state && gate && norm_weight && og && counters,
                 "lina_gla_decode_update_norm: null pointer");
    LINA_REQUIRE(B > 0 && H > 0, "lina_gla_decode_update_norm: B,H must be positive");
    LINA_REQUIRE(valid_dtype(dtype) && valid_dtype(g_dtype), "lina_gla_decode_update_norm: bad dtype enum");
    if (Dk <= 0 || Dk % 64 != 0) return fail(LINA_ERR_UNSUPPORTED, "lina_gla_decode_update_norm: Dk=%d must be a multiple of 64", Dk);
    LINA_REQUIRE(gate_sb % 4 == 0 && gate_sh % 4 == 0, "lina_gla_decode_update_norm: gate strides must be multiples of 4");
    const int64_t st[8] = {q_sb, q_sh, k_sb, k_sh, v_sb, v_sh, g_sb, g_sh};
    if (dtype == LINA_F32 && g_dtype == LINA_F32)
        return launch_rowsplit<float, float>(q, k, v, gk, o_part, state, B, H, Dk, Dv, st, scale, stream, gate, gate_sb,
                                             gate_sh, norm_weight, eps, og, counters);
    if (dtype == LINA_BF16 && g_dtype == LINA_F32)
        return launch_rowsplit<bf16_t, float>(q, k, v, gk, o_part, state, B, H, Dk, Dv, st, scale, stream, gate, gate_sb,
                                              gate_sh, norm_weight, eps, og, counters);
    if (dtype == LINA_BF16 && g_dtype == LINA_BF16)
        return launch_rowsplit<bf16_t, bf16_t>(q, k, v, gk, o_part, state, B, H, Dk, Dv, st, scale, stream, gate,
                                               gate_sb, gate_sh, norm_weight, eps, og, counters);
    return fail(LINA_ERR_UNSUPPORTED, "lina_gla_decode_update_norm: dtype=f32 with bf16 gates is not built");
}
