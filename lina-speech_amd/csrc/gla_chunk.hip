// gla_chunk.hip -- K2: chunk-wise GLA forward on the matrix cores, state tile resident in
// MFMA accumulators for the whole sequence (no per-chunk state ever touches HBM).
//
// Replaces fla.ops.gla.chunk_gla / fused_chunk_gla at the reference call sites model/gla.py:193,195
// (SURVEY.md 8(a) a-2, Appendix A.3/A.4).  Same contract and same results as K1.
//
// Formulation (per (b,h); b_t = inclusive cumsum of the log-gates inside the current chunk):
//     q~_t = scale * q_t * exp(b_t)        k~_s = k_s * exp(-b_s)
//     o_t  = q~_t S  +  sum_{s<=t} (q~_t . k~_s) v_s
//     S   <- diag(exp(b_last)) (S + k~^T v)
// exp(-b_s) grows with the decay inside a chunk, so the chunk is cut ADAPTIVELY: a chunk takes
// up to C = 16 tokens but ends early at the first row whose accumulated |b| would exceed
// kMaxDecay (and a single gate is clamped to >= -kMaxDecay, i.e. a decay of e^-60 ~ 9e-27).  With
// ordinary gates (logsigmoid/16) every chunk is full; the reference's reset gates of -20
// (model/gla.py:136,183) shorten a chunk only when >= 3 resets fall inside 16 tokens.  All factors
// therefore stay inside fp32/bf16 range and the result equals the exact-difference form of A.3.
//
// Work split: grid = (B*H, Dv/64), 256 threads = 4 waves.  Wave w owns state columns
// [16w, 16w+16) of the 64-wide tile for all Dk rows: Dk/16 accumulator tiles (f32x4 each) in the
// MFMA C/D layout (col = lane&15, row = 4*(lane>>4)+reg).  That layout is consumed directly as the
// B operand of q~.S by giving the A operand the matching k-slot -> channel map, so the state never
// leaves registers.  Per chunk: (A) thread c<Dk loads its channel's 16 gate/q/k values (coalesced
// along c), scans the gates, scales q,k and stages them in LDS; (B) MFMA: o_inter = q~.S,
// A = q~ k~^T (split over the waves along Dk, reduced through LDS), o += mask(A) v, S update.
// f32 inputs use v_mfma_f32_16x16x4_f32 (exact fp32), bf16 inputs v_mfma_f32_16x16x32_bf16.
#include <lina_dev.h>
#include "lina_common.h"

namespace lina {

constexpr int kChunk = 16;
constexpr float kMaxDecay = 60.0f;

int check_gla_args(const char* fn, const void* q, const void* k, const void* v, const void* gk, const void* o,
                   int B, int H, int T, int Dk, int Dv, int dtype, int g_dtype);
// gla_chunk_full.hip: bf16, Dk = Dv = 256, one workgroup per head
int launch_chunk_full(const void* q, const void* k, const void* v, const void* gk, void* o, const float* h0,
                      float* ht, int B, int H, int T, int Dk, int Dv, lina_bht_strides sq, lina_bht_strides sk,
                      lina_bht_strides sv, lina_bht_strides sg, lina_bht_strides so, int dtype, int g_dtype,
                      float scale, lina_stream_t stream, bool* taken);

// ----------------------------------------------------------------------------------------------
// phase A helper: scan the gates of one channel, return the chunk length this channel allows
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ int scan_gates(float (&bv)[kChunk], const float (&gv)[kChunk]) {
    float b = 0.0f;
    int nc = kChunk;
#pragma unroll
    for (int r = 0; r < kChunk; ++r) {
        b += fmaxf(gv[r], -kMaxDecay);
        if (r > 0 && nc == kChunk && -b > kMaxDecay) nc = r;
        bv[r] = b;
    }
    return nc;
}

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v = min(v, shfl_xor_i(v, m));
    return v;
}

// ================================== f32 path ====================================================
template <int DK, typename TG>
__global__ __launch_bounds__(256) void gla_chunk_f32_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const TG* __restrict__ gk, float* __restrict__ o, const float* h0, float* ht, int H, int T, int Dv,
    lina_bht_strides sq, lina_bht_strides sk, lina_bht_strides sv, lina_bht_strides sg, lina_bht_strides so,
    float scale) {
    constexpr int C = kChunk, BV = 64, NT = DK / 16;
    constexpr int SQ = DK + 2;   // row stride == 2 (mod 32): conflict-free (row=lane&15, col+=lane>>4) b32 reads
    constexpr int SV = BV + 16;  // row stride == 16 (mod 32)
    __shared__ float s_q[C * SQ], s_k[C * SQ];
    __shared__ float s_v[C * SV];
    __shared__ float s_dec[DK];
    __shared__ float s_A[4][C][C + 1];
    __shared__ int s_nw[4];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int v0 = blockIdx.y * BV;

    f32x4 S[NT];
    {
        const float* hp = h0 ? h0 + ((int64_t)bh * DK) * Dv + v0 + 16 * w + li : nullptr;
#pragma unroll
        for (int p = 0; p < NT; ++p)
#pragma unroll
            for (int r = 0; r < 4; ++r) S[p][r] = hp ? hp[(int64_t)(16 * p + 4 * lg + r) * Dv] : 0.0f;
    }

    const float* qb = q + b * sq.b + h * sq.h;
    const float* kb = k + b * sk.b + h * sk.h;
    const TG* gb = gk + b * sg.b + h * sg.h;
    const float* vb = v + b * sv.b + h * sv.h + v0;
    float* ob = o + b * so.b + h * so.h + v0 + 16 * w + li;

    const bool chan = tid < DK;
    const int ch = chan ? tid : 0;
    const int vr = tid >> 4, vc = (tid & 15) * 4;

    int t0 = 0;
    while (t0 < T) {
        // ---------------- phase A1: loads, gate scan ----------------
        float gv[C], qv[C], kv[C], bv[C];
#pragma unroll
        for (int r = 0; r < C; ++r) {
            // unconditional loads from a clamped address + select: no branch, all loads in flight together
            const int t = t0 + r, tc = min(t, T - 1);
            const bool in = chan && t < T;
            const float g_ = ld(gb + tc * sg.t + ch), q_ = qb[tc * sq.t + ch], k_ = kb[tc * sk.t + ch];
            gv[r] = in ? g_ : 0.0f;
            qv[r] = in ? q_ : 0.0f;
            kv[r] = in ? k_ : 0.0f;
        }
        float4 vv = *reinterpret_cast<const float4*>(vb + min(t0 + vr, T - 1) * sv.t + vc);
        if (t0 + vr >= T) vv = make_float4(0.f, 0.f, 0.f, 0.f);
        int nc = scan_gates(bv, gv);
        nc = wave_min_i(nc);
        if (lane == 0) s_nw[w] = nc;
        __syncthreads();  // (1) also: every wave is done with last chunk's s_q/s_k/s_v/s_dec
        const int n = min(min(min(s_nw[0], s_nw[1]), min(s_nw[2], s_nw[3])), T - t0);

        // ---------------- phase A2: scaled operands -> LDS ----------------
        if (chan) {
            float blast = 0.0f;
#pragma unroll
            for (int r = 0; r < C; ++r) {
                const bool valid = r < n;
                const float e = __expf(bv[r]);
                s_q[r * SQ + tid] = valid ? qv[r] * scale * e : 0.0f;
                s_k[r * SQ + tid] = valid ? kv[r] * __expf(-bv[r]) : 0.0f;
                if (r == n - 1) blast = bv[r];
            }
            s_dec[tid] = __expf(blast);
        }
        {
            const bool valid = vr < n;
            float* d = &s_v[vr * SV + vc];
            d[0] = valid ? vv.x : 0.0f; d[1] = valid ? vv.y : 0.0f;
            d[2] = valid ? vv.z : 0.0f; d[3] = valid ? vv.w : 0.0f;
        }
        __syncthreads();  // (2)

        // ---------------- phase B: matrix-core work ----------------
        // o_inter = q~ . S_old  (two accumulators: the f32 MFMA has a 40-cycle dependent latency)
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < NT; ++p) {
            const float* qp = &s_q[li * SQ + 16 * p + 4 * lg];
            acc0 = mfma_f32_16x16x4(qp[0], S[p][0], acc0);
            acc1 = mfma_f32_16x16x4(qp[1], S[p][1], acc1);
            acc0 = mfma_f32_16x16x4(qp[2], S[p][2], acc0);
            acc1 = mfma_f32_16x16x4(qp[3], S[p][3], acc1);
        }
        // partial A = q~ k~^T over this wave's quarter of the channels
        {
            f32x4 pa = {0.f, 0.f, 0.f, 0.f};
            const int c0 = w * (DK / 4);
#pragma unroll
            for (int kk = 0; kk < DK / 16; ++kk) {
                const int cc = c0 + 4 * kk + lg;
                pa = mfma_f32_16x16x4(s_q[li * SQ + cc], s_k[li * SQ + cc], pa);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) s_A[w][4 * lg + r][li] = pa[r];
        }
        __syncthreads();  // (3)
        // o += mask(A) . v ; the same v fragments feed the state update
        float bvf[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int s = 4 * kk + lg;
            bvf[kk] = s_v[s * SV + 16 * w + li];
            float a = (s_A[0][li][s] + s_A[1][li][s]) + (s_A[2][li][s] + s_A[3][li][s]);
            a = (s <= li) ? a : 0.0f;
            if (kk & 1) acc1 = mfma_f32_16x16x4(a, bvf[kk], acc1);
            else acc0 = mfma_f32_16x16x4(a, bvf[kk], acc0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * lg + r;
            if (row < n) ob[(t0 + row) * so.t] = acc0[r] + acc1[r];
        }
        // S <- diag(exp(b_last)) (S + k~^T v)
#pragma unroll
        for (int p = 0; p < NT; ++p) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) S[p] = mfma_f32_16x16x4(s_k[(4 * kk + lg) * SQ + 16 * p + li], bvf[kk], S[p]);
#pragma unroll
            for (int r = 0; r < 4; ++r) S[p][r] *= s_dec[16 * p + 4 * lg + r];
        }
        t0 += n;
    }

    if (ht) {
        float* hp = ht + ((int64_t)bh * DK) * Dv + v0 + 16 * w + li;
#pragma unroll
        for (int p = 0; p < NT; ++p)
#pragma unroll
            for (int r = 0; r < 4; ++r) hp[(int64_t)(16 * p + 4 * lg + r) * Dv] = S[p][r];
    }
}

// ================================== bf16 path ===================================================
__device__ __forceinline__ bf16x8 ld8(const bf16_t* p) {  // 16-byte aligned LDS read
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    bf16x8 r;
    r[0] = (short)(u.x & 0xffff); r[1] = (short)(u.x >> 16); r[2] = (short)(u.y & 0xffff); r[3] = (short)(u.y >> 16);
    r[4] = (short)(u.z & 0xffff); r[5] = (short)(u.z >> 16); r[6] = (short)(u.w & 0xffff); r[7] = (short)(u.w >> 16);
    return r;
}

template <int DK, typename TG>
__global__ __launch_bounds__(256) void gla_chunk_bf16_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    const TG* __restrict__ gk, bf16_t* __restrict__ o, const float* h0, float* ht, int H, int T, int Dv,
    lina_bht_strides sq, lina_bht_strides sk, lina_bht_strides sv, lina_bht_strides sg, lina_bht_strides so,
    float scale) {
    constexpr int C = kChunk, BV = 64, NT = DK / 16;
    constexpr int NWA = DK / 64;  // waves that take part in the split-K of A (64 channels each)
    constexpr int SQ = DK + 8;    // bf16 elements; rows stay 16-byte aligned
    constexpr int ST = C + 8;     // transposed tiles: [channel or column][token]
    __shared__ __attribute__((aligned(16))) bf16_t s_q[C * SQ];
    __shared__ __attribute__((aligned(16))) bf16_t s_k[C * SQ];
    __shared__ __attribute__((aligned(16))) bf16_t s_kT[DK * ST];
    __shared__ __attribute__((aligned(16))) bf16_t s_vT[BV * ST];
    __shared__ float s_dec[DK];
    __shared__ float s_A[NWA][C][C + 1];
    __shared__ int s_nw[4];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int v0 = blockIdx.y * BV;

    f32x4 S[NT];
    {
        const float* hp = h0 ? h0 + ((int64_t)bh * DK) * Dv + v0 + 16 * w + li : nullptr;
#pragma unroll
        for (int p = 0; p < NT; ++p)
#pragma unroll
            for (int r = 0; r < 4; ++r) S[p][r] = hp ? hp[(int64_t)(16 * p + 4 * lg + r) * Dv] : 0.0f;
    }

    const bf16_t* qb = q + b * sq.b + h * sq.h;
    const bf16_t* kb = k + b * sk.b + h * sk.h;
    const TG* gb = gk + b * sg.b + h * sg.h;
    const bf16_t* vb = v + b * sv.b + h * sv.h + v0;
    bf16_t* ob = o + b * so.b + h * so.h + v0 + 16 * w + li;

    const bool chan = tid < DK;
    const int ch = chan ? tid : 0;
    const int vr = tid >> 4, vc = (tid & 15) * 4;

    int t0 = 0;
    while (t0 < T) {
        float gv[C], qv[C], kv[C], bv[C];
#pragma unroll
        for (int r = 0; r < C; ++r) {
            const int t = t0 + r, tc = min(t, T - 1);
            const bool in = chan && t < T;
            const float g_ = ld(gb + tc * sg.t + ch), q_ = ld(qb + tc * sq.t + ch), k_ = ld(kb + tc * sk.t + ch);
            gv[r] = in ? g_ : 0.0f;
            qv[r] = in ? q_ : 0.0f;
            kv[r] = in ? k_ : 0.0f;
        }
        float4 vv = ld4(vb + min(t0 + vr, T - 1) * sv.t + vc);
        if (t0 + vr >= T) vv = make_float4(0.f, 0.f, 0.f, 0.f);
        int nc = scan_gates(bv, gv);
        nc = wave_min_i(nc);
        if (lane == 0) s_nw[w] = nc;
        __syncthreads();  // (1)
        const int n = min(min(min(s_nw[0], s_nw[1]), min(s_nw[2], s_nw[3])), T - t0);

        if (chan) {
            float blast = 0.0f;
#pragma unroll
            for (int r = 0; r < C; ++r) {
                const bool valid = r < n;
                const bf16_t qt = f2bf(valid ? qv[r] * scale * __expf(bv[r]) : 0.0f);
                const bf16_t kt = f2bf(valid ? kv[r] * __expf(-bv[r]) : 0.0f);
                s_q[r * SQ + tid] = qt;
                s_k[r * SQ + tid] = kt;
                s_kT[tid * ST + r] = kt;
                if (r == n - 1) blast = bv[r];
            }
            s_dec[tid] = __expf(blast);
        }
        {
            const bool valid = vr < n;
            s_vT[(vc + 0) * ST + vr] = f2bf(valid ? vv.x : 0.0f);
            s_vT[(vc + 1) * ST + vr] = f2bf(valid ? vv.y : 0.0f);
            s_vT[(vc + 2) * ST + vr] = f2bf(valid ? vv.z : 0.0f);
            s_vT[(vc + 3) * ST + vr] = f2bf(valid ? vv.w : 0.0f);
        }
        __syncthreads();  // (2)

        // o_inter = q~ . S_old : one K=32 MFMA per pair of 16-row state tiles
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pp = 0; pp < NT / 2; ++pp) {
            const bf16_t* qp = &s_q[li * SQ + 32 * pp + 4 * lg];
            const uint2 a_lo = *reinterpret_cast<const uint2*>(qp);
            const uint2 a_hi = *reinterpret_cast<const uint2*>(qp + 16);
            bf16x8 a, bb;
            a[0] = (short)(a_lo.x & 0xffff); a[1] = (short)(a_lo.x >> 16); a[2] = (short)(a_lo.y & 0xffff); a[3] = (short)(a_lo.y >> 16);
            a[4] = (short)(a_hi.x & 0xffff); a[5] = (short)(a_hi.x >> 16); a[6] = (short)(a_hi.y & 0xffff); a[7] = (short)(a_hi.y >> 16);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                bb[r] = (short)f2bf(S[2 * pp][r]);
                bb[4 + r] = (short)f2bf(S[2 * pp + 1][r]);
            }
            acc = mfma_bf16_16x16x32(a, bb, acc);
        }
        if (w < NWA) {
            f32x4 pa = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int cc = 64 * w + 32 * kk + 8 * lg;
                pa = mfma_bf16_16x16x32(ld8(&s_q[li * SQ + cc]), ld8(&s_k[li * SQ + cc]), pa);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) s_A[w][4 * lg + r][li] = pa[r];
        }
        __syncthreads();  // (3)
        bf16x8 a_in, b_v;
#pragma unroll
        for (int j = 0; j < 8; ++j) { a_in[j] = 0; b_v[j] = 0; }
        if (lg < 2) {
            b_v = ld8(&s_vT[(16 * w + li) * ST + 8 * lg]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int s = 8 * lg + j;
                float a = 0.0f;
#pragma unroll
                for (int ww = 0; ww < NWA; ++ww) a += s_A[ww][li][s];
                a_in[j] = (short)f2bf((s <= li) ? a : 0.0f);
            }
        }
        acc = mfma_bf16_16x16x32(a_in, b_v, acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * lg + r;
            if (row < n) ob[(t0 + row) * so.t] = f2bf(acc[r]);
        }
#pragma unroll
        for (int p = 0; p < NT; ++p) {
            bf16x8 a_k;
#pragma unroll
            for (int j = 0; j < 8; ++j) a_k[j] = 0;
            if (lg < 2) a_k = ld8(&s_kT[(16 * p + li) * ST + 8 * lg]);
            S[p] = mfma_bf16_16x16x32(a_k, b_v, S[p]);
#pragma unroll
            for (int r = 0; r < 4; ++r) S[p][r] *= s_dec[16 * p + 4 * lg + r];
        }
        t0 += n;
    }

    if (ht) {
        float* hp = ht + ((int64_t)bh * DK) * Dv + v0 + 16 * w + li;
#pragma unroll
        for (int p = 0; p < NT; ++p)
#pragma unroll
            for (int r = 0; r < 4; ++r) hp[(int64_t)(16 * p + 4 * lg + r) * Dv] = S[p][r];
    }
}

template <int DK>
static int launch_chunk(int dtype, int g_dtype, const void* q, const void* k, const void* v, const void* gk, void* o,
                        const float* h0, float* ht, int B, int H, int T, int Dv, lina_bht_strides sq,
                        lina_bht_strides sk, lina_bht_strides sv, lina_bht_strides sg, lina_bht_strides so, float scale,
                        lina_stream_t stream) {
    dim3 grid((unsigned)(B * H), (unsigned)(Dv / 64));
    if (dtype == LINA_F32 && g_dtype == LINA_F32) {
        LINA_LAUNCH((gla_chunk_f32_kernel<DK, float>), grid, dim3(256), 0, stream, (const float*)q, (const float*)k,
                    (const float*)v, (const float*)gk, (float*)o, h0, ht, H, T, Dv, sq, sk, sv, sg, so, scale);
    } else if (dtype == LINA_BF16 && g_dtype == LINA_BF16) {
        LINA_LAUNCH((gla_chunk_bf16_kernel<DK, bf16_t>), grid, dim3(256), 0, stream, (const bf16_t*)q, (const bf16_t*)k,
                    (const bf16_t*)v, (const bf16_t*)gk, (bf16_t*)o, h0, ht, H, T, Dv, sq, sk, sv, sg, so, scale);
    } else if (dtype == LINA_BF16 && g_dtype == LINA_F32) {
        LINA_LAUNCH((gla_chunk_bf16_kernel<DK, float>), grid, dim3(256), 0, stream, (const bf16_t*)q, (const bf16_t*)k,
                    (const bf16_t*)v, (const float*)gk, (bf16_t*)o, h0, ht, H, T, Dv, sq, sk, sv, sg, so, scale);
    } else {
        return fail(LINA_ERR_UNSUPPORTED, "lina_gla_chunk_fwd: dtype=f32 with bf16 gates is not built");
    }
    return check_launch("lina_gla_chunk_fwd");
}

}  // namespace lina

extern "C" int lina_gla_chunk_fwd(const void* q, const void* k, const void* v, const void* gk, void* o,
                                  const float* h0, float* ht, int B, int H, int T, int Dk, int Dv,
                                  lina_bht_strides sq, lina_bht_strides sk, lina_bht_strides sv,
                                  lina_bht_strides sg, lina_bht_strides so,
                                  int dtype, int g_dtype, float scale, lina_stream_t stream) {
    using namespace lina;
    int rc = check_gla_args("lina_gla_chunk_fwd", q, k, v, gk, o, B, H, T, Dk, Dv, dtype, g_dtype);
    if (rc) return rc;
    bool taken = false;
    rc = launch_chunk_full(q, k, v, gk, o, h0, ht, B, H, T, Dk, Dv, sq, sk, sv, sg, so, dtype, g_dtype, scale, stream,
                           &taken);
    if (taken) return rc;
    // v rows are read 4 elements at a time
    LINA_REQUIRE(sv.t % 4 == 0 && sv.b % 4 == 0 && sv.h % 4 == 0, "lina_gla_chunk_fwd: v strides must be multiples of 4");
    switch (Dk) {
        case 64: return launch_chunk<64>(dtype, g_dtype, q, k, v, gk, o, h0, ht, B, H, T, Dv, sq, sk, sv, sg, so, scale, stream);
        case 128: return launch_chunk<128>(dtype, g_dtype, q, k, v, gk, o, h0, ht, B, H, T, Dv, sq, sk, sv, sg, so, scale, stream);
        case 256: return launch_chunk<256>(dtype, g_dtype, q, k, v, gk, o, h0, ht, B, H, T, Dv, sq, sk, sv, sg, so, scale, stream);
    }
    return fail(LINA_ERR_UNSUPPORTED, "lina_gla_chunk_fwd: Dk=%d", Dk);
}
