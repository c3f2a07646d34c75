// cross_att.hip -- the two attention steps of the blind cross-attention at T = 1 (SURVEY.md 8(f) f-1;
// reference model/crossatt.py:105-155 with the eager softmax(q k^T / sqrt d) v of :13-19), one workgroup
// per utterance row, text-side tensors precomputed once per utterance (BlindCrossAttention.prepare):
//   step 1: q = LayerNorm(q_lin) ; att1 = softmax(q . K_b^T * scale) ; xp = att1 . PE          (:114,143)
//   step 2: att2 = softmax(xp' . PE^T * scale) ; x += att2 . V_b                                   (:149, gla.py:362)
// (xp' = xp after the pos_net GLA block, :145).  Replaces ~11 small torch launches per token.
// Scores live in LDS; the d-long dot products are split over the 256 threads (4 elements per lane per trip,
// wave64 shuffle + LDS reduction), the weighted sums over the T_txt rows are column-parallel.
#include <lina_dev.h>
#include "lina_common.h"
#include "skinny_frag.h"

namespace lina {

constexpr int kCaMaxT = 1024;   // text positions held in LDS

__device__ __forceinline__ float block_sum(float v, float* s_red) {   // 256 threads
    v += shfl_xor(v, 1); v += shfl_xor(v, 2); v += shfl_xor(v, 4);
    v += shfl_xor(v, 8); v += shfl_xor(v, 16); v += shfl_xor(v, 32);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}
__device__ __forceinline__ float block_max(float v, float* s_red) {
    v = fmaxf(v, shfl_xor(v, 1)); v = fmaxf(v, shfl_xor(v, 2)); v = fmaxf(v, shfl_xor(v, 4));
    v = fmaxf(v, shfl_xor(v, 8)); v = fmaxf(v, shfl_xor(v, 16)); v = fmaxf(v, shfl_xor(v, 32));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
}

// scores[t] = scale * <vec, M[t,:]> for t < T, then softmax in place (s_sc); every thread returns with s_sc valid
template <typename T>
__device__ __forceinline__ void scores_softmax(const float* s_vec, const T* __restrict__ Mrows, int64_t row_stride,
                                               int Tn, int d, float scale, float* s_sc, float* s_red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int t = w; t < Tn; t += 4) {          // one wave per text row
        float acc = 0.0f;
        for (int e = lane * 4; e < d; e += 256) {
            const float4 m = ld4(Mrows + (int64_t)t * row_stride + e);
            acc = fmaf(m.x, s_vec[e], fmaf(m.y, s_vec[e + 1], fmaf(m.z, s_vec[e + 2], fmaf(m.w, s_vec[e + 3], acc))));
        }
        acc += shfl_xor(acc, 1); acc += shfl_xor(acc, 2); acc += shfl_xor(acc, 4);
        acc += shfl_xor(acc, 8); acc += shfl_xor(acc, 16); acc += shfl_xor(acc, 32);
        if (lane == 0) s_sc[t] = acc * scale;
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int t = threadIdx.x; t < Tn; t += 256) mx = fmaxf(mx, s_sc[t]);
    mx = block_max(mx, s_red);
    float sum = 0.0f;
    for (int t = threadIdx.x; t < Tn; t += 256) { const float e = expf(s_sc[t] - mx); s_sc[t] = e; sum += e; }
    sum = block_sum(sum, s_red);
    const float inv = 1.0f / sum;
    for (int t = threadIdx.x; t < Tn; t += 256) s_sc[t] *= inv;
    __syncthreads();
}

// ---- spread versions (one workgroup per utterance row pulls 256 KiB through a single CU: 30-40 us; these
// ---- put >= 256 workgroups on the text-side tensors) ----------------------------------------------------------

#ifdef LINA_SKINNY_PROF
// tools-only build (tools/skinny_prof.sh): time stamps of thread 0 of every cross_scores workgroup, [workgroup][slot]:
// 0 wall clock at entry, 1 shader clock at entry, 2 query + parameters arrived, 3 LayerNorm done (barrier), 4 dots done, 5 end,
// 6 wall clock at the end.  NOT part of the product library.
__device__ unsigned long long lina_cross_prof[1024 * 8];
#define CS_PROF(i, expr) do { if (threadIdx.x == 0) pr_[i] = (expr); } while (0)
#else
#define CS_PROF(i, expr) do { } while (0)
#endif

// scores[b,t] = scale * <LN(q_lin[b]), kk[b,t,:]>   grid (ceil(Tn/16), B): 4 waves x 4 text rows each
template <typename T>
__global__ __launch_bounds__(256) void cross_scores_kernel(
    const T* __restrict__ qlin, const T* __restrict__ ln_w, const T* __restrict__ ln_b, float ln_eps,
    const T* __restrict__ kk, float* __restrict__ scores, int Tn, int d, float scale) {
    LINA_DYN_SMEM(smem);
    float* s_q = reinterpret_cast<float*>(smem);            // [d]
    __shared__ float s_red[4];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
#ifdef LINA_SKINNY_PROF
    unsigned long long pr_[8] = {};
    CS_PROF(0, wall_clock64());
    CS_PROF(1, clock64());
#endif
    // the text rows of this wave are requested FIRST (d <= 1024: 4 rows x 4 pieces of 4 elements per lane), so their
    // HBM latency runs under the LayerNorm of the query below
    constexpr int kIt = 4;
    const bool pre = d <= 256 * kIt;
    // A wave's loads RETURN IN ORDER: the query and the LayerNorm parameters are requested BEFORE the (HBM-cold) text rows, so
    // the LayerNorm runs while the rows stream in.  Requested behind them (as this kernel did until round 3) they arrived
    // only after the whole 32 KiB burst, and the LayerNorm chain -- two workgroup reductions and the parameter loads -- ran
    // exposed after it: 11.7 us per launch against 4.7 us for the weighted-row kernel that moves the same bytes.
    float qv[kIt], gw[kIt], gb[kIt];
#pragma unroll
    for (int i = 0; i < kIt; ++i) {
        const int e = tid + 256 * i;
        const bool ok = pre && e < d;
        const int ec = e < d ? e : d - 1;                    // unconditional loads on clamped indices (a predicated load
        const float a0 = ld(qlin + (int64_t)b * d + ec), a1 = ld(ln_w + ec), a2 = ld(ln_b + ec);   // is waited for at once)
        qv[i] = ok ? a0 : 0.0f;
        gw[i] = ok ? a1 : 0.0f;
        gb[i] = ok ? a2 : 0.0f;
    }
    float4 m[4][kIt];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = min((int)blockIdx.x * 16 + 4 * w + i, Tn - 1);
        const T* row = kk + ((int64_t)b * Tn + t) * d;
#pragma unroll
        for (int it = 0; it < kIt; ++it) {
            const int e = lane * 4 + 256 * it;
            const float4 v4 = ld4(row + (e < d ? e : d - 4));
            m[i][it] = (pre && e < d) ? v4 : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    float s = 0.0f;
    if (pre) {
#pragma unroll
        for (int i = 0; i < kIt; ++i) s += qv[i];
#ifdef LINA_SKINNY_PROF
        if (s == 1.2345e30f) s_q[0] = s;                     // (consume the loads before the stamp)
        CS_PROF(2, clock64());
#endif
    } else {
        for (int e = tid; e < d; e += 256) { const float x = ld(qlin + (int64_t)b * d + e); s_q[e] = x; s += x; }
    }
    const float mu = block_sum(s, s_red) / (float)d;
    float vs = 0.0f;
    if (pre) {
#pragma unroll
        for (int i = 0; i < kIt; ++i) { const float c = qv[i] - mu; vs += (tid + 256 * i < d) ? c * c : 0.0f; }
    } else {
        for (int e = tid; e < d; e += 256) { const float c = s_q[e] - mu; vs += c * c; }
    }
    const float rstd = rsqrtf(block_sum(vs, s_red) / (float)d + ln_eps);
    if (pre) {
#pragma unroll
        for (int i = 0; i < kIt; ++i) {
            const int e = tid + 256 * i;
            if (e < d) {
                T tmp;                                       // the reference rounds the LN output to the model dtype
                st(&tmp, (qv[i] - mu) * rstd * gw[i] + gb[i]);
                s_q[e] = ld(&tmp);
            }
        }
    } else {
        for (int e = tid; e < d; e += 256) {
            T tmp;
            st(&tmp, (s_q[e] - mu) * rstd * ld(ln_w + e) + ld(ln_b + e));
            s_q[e] = ld(&tmp);
        }
    }
    __syncthreads();
    CS_PROF(3, clock64());
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = blockIdx.x * 16 + 4 * w + i;           // wave-uniform
        if (t >= Tn) break;
        float acc = 0.0f;
        if (pre) {
#pragma unroll
            for (int it = 0; it < kIt; ++it) {
                const int e = lane * 4 + 256 * it;
                if (e < d) {
                    const float4 mm = m[i][it];
                    acc = fmaf(mm.x, s_q[e], fmaf(mm.y, s_q[e + 1], fmaf(mm.z, s_q[e + 2], fmaf(mm.w, s_q[e + 3], acc))));
                }
            }
        } else {
            const T* row = kk + ((int64_t)b * Tn + t) * d;
            for (int e = lane * 4; e < d; e += 256) {
                const float4 mm = ld4(row + e);
                acc = fmaf(mm.x, s_q[e], fmaf(mm.y, s_q[e + 1], fmaf(mm.z, s_q[e + 2], fmaf(mm.w, s_q[e + 3], acc))));
            }
        }
        acc += shfl_xor(acc, 1); acc += shfl_xor(acc, 2); acc += shfl_xor(acc, 4);
        acc += shfl_xor(acc, 8); acc += shfl_xor(acc, 16); acc += shfl_xor(acc, 32);
        if (lane == 0) scores[(int64_t)b * Tn + t] = acc * scale;
    }
#ifdef LINA_SKINNY_PROF
    CS_PROF(4, clock64());
    CS_PROF(5, clock64());
    CS_PROF(6, wall_clock64());
    if (threadIdx.x == 0) {
        const int wg = blockIdx.y * gridDim.x + blockIdx.x;
        if (wg < 1024)
            for (int i = 0; i < 8; ++i) lina_cross_prof[wg * 8 + i] = pr_[i];
    }
#endif
}

// row softmax of x[b, 0:Tn] * scale; written (model dtype) to the strided attention buffer AND to a contiguous
// zero-padded [B, Tp] copy that feeds the following projection.  One wave per row.
template <typename TX, typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const TX* __restrict__ x, int64_t x_sb, float scale,
                                                           T* __restrict__ att, int64_t att_sb, T* __restrict__ attc,
                                                           int B, int Tn, int Tp) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;                                      // whole wave
    float mx = -INFINITY;
    for (int t = lane; t < Tn; t += 64) mx = fmaxf(mx, ld(x + b * x_sb + t) * scale);
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) mx = fmaxf(mx, shfl_xor(mx, m));
    float sum = 0.0f;
    for (int t = lane; t < Tn; t += 64) sum += expf(ld(x + b * x_sb + t) * scale - mx);
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) sum += shfl_xor(sum, m);
    const float inv = 1.0f / sum;
    for (int t = lane; t < Tp; t += 64) {
        const float p = t < Tn ? expf(ld(x + b * x_sb + t) * scale - mx) * inv : 0.0f;
        if (t < Tn) st(att + b * att_sb + t, p);
        st(attc + (int64_t)b * Tp + t, p);
    }
}

// x[b, e] += sum_t att[b,t] * vv[b,t,e]    grid (ceil(d/256), B): workgroup = (b, 256-column slab); lane = 4 columns,
// wave w takes text rows w, w+4, ... (8 row loads in flight per lane), the 4 partial sums meet in LDS
template <typename T>
__global__ __launch_bounds__(256) void weighted_rows_kernel(const T* __restrict__ attc, int Tp, const T* __restrict__ vv,
                                                            T* x, int Tn, int d, T* xpk) {
    __shared__ float s_a[kCaMaxT];
    __shared__ __attribute__((aligned(16))) float s_p[3][64][4];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int t = tid; t < Tn; t += 256) s_a[t] = ld(attc + (int64_t)b * Tp + t);
    __syncthreads();
    const int e = blockIdx.x * 256 + lane * 4;
    const bool live = e < d;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const T* base = vv + (int64_t)b * Tn * d + (live ? e : 0);
    for (int t0 = w; t0 < Tn; t0 += 32) {
        float4 p[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) p[u] = ld4(base + (int64_t)min(t0 + 4 * u, Tn - 1) * d);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float a = (t0 + 4 * u < Tn) ? s_a[min(t0 + 4 * u, Tn - 1)] : 0.0f;
            acc.x = fmaf(a, p[u].x, acc.x); acc.y = fmaf(a, p[u].y, acc.y);
            acc.z = fmaf(a, p[u].z, acc.z); acc.w = fmaf(a, p[u].w, acc.w);
        }
    }
    if (w > 0) *reinterpret_cast<float4*>(&s_p[w - 1][lane][0]) = acc;
    __syncthreads();
    if (w > 0 || !live) return;
#pragma unroll
    for (int ww = 0; ww < 3; ++ww) {
        const float4 o = *reinterpret_cast<const float4*>(&s_p[ww][lane][0]);
        acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
    }
    T tmp4[4];
    st4(tmp4, acc);                                          // bmm result in the model dtype, then the residual add
    // xpk given: the residual stream lives in the fragment-major buffer (read-modify-write there; x is not touched)
    T* const xa = xpk ? xpk + packed_off<T>(b, e, d) : x + (int64_t)b * d + e;
    const float4 o = ld4(tmp4), r = ld4(xa);
    st4(xa, make_float4(r.x + o.x, r.y + o.y, r.z + o.z, r.w + o.w));
}

// ---- round-2 fusions: fewer launches on the serial chain of the decode step (a launch slot costs ~5 us there) ---------
// scores + softmax in ONE launch: one workgroup of 1024 threads per utterance row (the whole row's T_txt scores are
// needed for the softmax); wave w takes text rows w, w+16, ... -- a row's d elements are one contiguous run, read 8 or 16
// bytes per lane -- and the first row of every wave is requested before the query's LayerNorm.
//   att[b, 0:Tn] = softmax(scale * <LN(q_lin[b]), kk[b,t,:]>)  -> strided att rows + zero-padded contiguous attc [B,Tp]
template <typename T>
__global__ __launch_bounds__(1024) void cross_scores_softmax_kernel(
    const T* __restrict__ qlin, const T* __restrict__ ln_w, const T* __restrict__ ln_b, float ln_eps,
    const T* __restrict__ kk, T* __restrict__ att, int64_t att_sb, T* __restrict__ attc, int Tn, int Tp, int d,
    float scale) {
    LINA_DYN_SMEM(smem);
    float* s_q = reinterpret_cast<float*>(smem);            // [d]
    float* s_sc = s_q + d;                                    // [Tn]
    __shared__ float s_red[16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    auto wg_sum = [&](float v) {
        v += shfl_xor(v, 1); v += shfl_xor(v, 2); v += shfl_xor(v, 4);
        v += shfl_xor(v, 8); v += shfl_xor(v, 16); v += shfl_xor(v, 32);
        __syncthreads();
        if (lane == 0) s_red[w] = v;
        __syncthreads();
        float t = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += s_red[i];
        return t;
    };
    constexpr int kIt = 4;                                   // d <= 1024: a row = 4 pieces of 4 elements per lane
    constexpr int kPre = 4;                                  // text rows per wave requested up front (T_txt <= 64: all of them)
    const bool pre = d <= 256 * kIt;
    // every row a wave will need is requested NOW (the text keys are HBM-cold: 16.8 MB stream between two tokens): a loop
    // that fetches row t + 16 after finishing row t pays a full memory round trip per row (3 of them on top of the first
    // at T_txt = 64: 11.7 us per launch in the step's timeline, profiles/r03c_step_timeline.txt)
    float4 m[kPre][kIt];
#pragma unroll
    for (int r = 0; r < kPre; ++r) {
        const int t = min(w + 16 * r, Tn - 1);
        const T* row = kk + ((int64_t)b * Tn + t) * d;
#pragma unroll
        for (int it = 0; it < kIt; ++it) {
            const int e = lane * 4 + 256 * it;
            const float4 v4 = ld4(row + (e < d ? e : d - 4));        // unconditional, clamped (see cross_scores_kernel)
            m[r][it] = (pre && e < d) ? v4 : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    float s = 0.0f;
    for (int e = tid; e < d; e += 1024) { const float x = ld(qlin + (int64_t)b * d + e); s_q[e] = x; s += x; }
    const float mu = wg_sum(s) / (float)d;
    float vs = 0.0f;
    for (int e = tid; e < d; e += 1024) { const float c = s_q[e] - mu; vs += c * c; }
    const float rstd = rsqrtf(wg_sum(vs) / (float)d + ln_eps);
    for (int e = tid; e < d; e += 1024) {
        T tmp;                                               // the reference rounds the LN output to the model dtype
        st(&tmp, (s_q[e] - mu) * rstd * ld(ln_w + e) + ld(ln_b + e));
        s_q[e] = ld(&tmp);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kPre; ++r) {                         // the prefetched rows (same arithmetic as the loop below)
        const int t = w + 16 * r;
        if (!pre || t >= Tn) break;                          // wave-uniform
        float acc = 0.0f;
#pragma unroll
        for (int it = 0; it < kIt; ++it) {
            const int e = lane * 4 + 256 * it;
            if (e < d)
                acc = fmaf(m[r][it].x, s_q[e], fmaf(m[r][it].y, s_q[e + 1], fmaf(m[r][it].z, s_q[e + 2], fmaf(m[r][it].w, s_q[e + 3], acc))));
        }
        acc += shfl_xor(acc, 1); acc += shfl_xor(acc, 2); acc += shfl_xor(acc, 4);
        acc += shfl_xor(acc, 8); acc += shfl_xor(acc, 16); acc += shfl_xor(acc, 32);
        if (lane == 0) s_sc[t] = acc * scale;
    }
    for (int t = pre ? w + 16 * kPre : w; t < Tn; t += 16) { // wave-uniform: rows beyond the prefetch window / wide models
        float acc = 0.0f;
        {
            const T* row = kk + ((int64_t)b * Tn + t) * d;
            for (int e = lane * 4; e < d; e += 256) {
                const float4 mm = ld4(row + e);
                acc = fmaf(mm.x, s_q[e], fmaf(mm.y, s_q[e + 1], fmaf(mm.z, s_q[e + 2], fmaf(mm.w, s_q[e + 3], acc))));
            }
        }
        acc += shfl_xor(acc, 1); acc += shfl_xor(acc, 2); acc += shfl_xor(acc, 4);
        acc += shfl_xor(acc, 8); acc += shfl_xor(acc, 16); acc += shfl_xor(acc, 32);
        if (lane == 0) s_sc[t] = acc * scale;
    }
    __syncthreads();
    if (w == 0) {                                            // softmax of the row by one wave (same arithmetic as softmax_rows)
        float mx = -INFINITY;
        for (int t = lane; t < Tn; t += 64) mx = fmaxf(mx, s_sc[t]);
#pragma unroll
        for (int mm = 1; mm < 64; mm <<= 1) mx = fmaxf(mx, shfl_xor(mx, mm));
        float sum = 0.0f;
        for (int t = lane; t < Tn; t += 64) sum += expf(s_sc[t] - mx);
#pragma unroll
        for (int mm = 1; mm < 64; mm <<= 1) sum += shfl_xor(sum, mm);
        const float inv = 1.0f / sum;
        for (int t = lane; t < Tp; t += 64) {
            const float p = t < Tn ? expf(s_sc[t] - mx) * inv : 0.0f;
            if (t < Tn) st(att + b * att_sb + t, p);
            st(attc + (int64_t)b * Tp + t, p);
        }
    }
}

// softmax + weighted row sum + residual in ONE launch: as weighted_rows_kernel, but every workgroup first turns the
// row's raw scores (model dtype, from the x_pos . pe^T projection) into the softmax weights itself (T_txt values: nothing
// next to the 2 KiB it then pulls per text row); the slab-0 workgroup of a row also stores them as the attention output.
template <typename T>
__global__ __launch_bounds__(256) void softmax_weighted_rows_kernel(const T* __restrict__ scores, int64_t sc_sb, float scale,
                                                                    T* __restrict__ att, int64_t att_sb,
                                                                    const T* __restrict__ vv, T* x, int Tn, int d, T* xpk) {
    __shared__ float s_a[kCaMaxT];
    __shared__ float s_red[4];
    __shared__ __attribute__((aligned(16))) float s_p[3][64][4];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    float mx = -INFINITY;
    for (int t = tid; t < Tn; t += 256) { const float v = ld(scores + b * sc_sb + t) * scale; s_a[t] = v; mx = fmaxf(mx, v); }
    mx = block_max(mx, s_red);
    float sum = 0.0f;
    for (int t = tid; t < Tn; t += 256) sum += expf(s_a[t] - mx);
    sum = block_sum(sum, s_red);
    const float inv = 1.0f / sum;
    for (int t = tid; t < Tn; t += 256) {
        T tmp;                                               // the weights in the model dtype, as softmax_rows stores them
        st(&tmp, expf(s_a[t] - mx) * inv);
        s_a[t] = ld(&tmp);
        if (blockIdx.x == 0) att[b * att_sb + t] = tmp;
    }
    __syncthreads();
    const int e = blockIdx.x * 256 + lane * 4;
    const bool live = e < d;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const T* base = vv + (int64_t)b * Tn * d + (live ? e : 0);
    for (int t0 = w; t0 < Tn; t0 += 32) {
        float4 p[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) p[u] = ld4(base + (int64_t)min(t0 + 4 * u, Tn - 1) * d);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float a = (t0 + 4 * u < Tn) ? s_a[min(t0 + 4 * u, Tn - 1)] : 0.0f;
            acc.x = fmaf(a, p[u].x, acc.x); acc.y = fmaf(a, p[u].y, acc.y);
            acc.z = fmaf(a, p[u].z, acc.z); acc.w = fmaf(a, p[u].w, acc.w);
        }
    }
    if (w > 0) *reinterpret_cast<float4*>(&s_p[w - 1][lane][0]) = acc;
    __syncthreads();
    if (w > 0 || !live) return;
#pragma unroll
    for (int ww = 0; ww < 3; ++ww) {
        const float4 o = *reinterpret_cast<const float4*>(&s_p[ww][lane][0]);
        acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
    }
    T tmp4[4];
    st4(tmp4, acc);
    T* const xa = xpk ? xpk + packed_off<T>(b, e, d) : x + (int64_t)b * d + e;
    const float4 o = ld4(tmp4), r = ld4(xa);
    st4(xa, make_float4(r.x + o.x, r.y + o.y, r.z + o.z, r.w + o.w));
}

// x_pos . pe^T + softmax + weighted row sum + residual in ONE launch (round 4: the last two launches of the cross-attention
// step as one): as softmax_weighted_rows_kernel, but every workgroup first computes the row's T_txt raw scores
//   sc[t] = < xp[b, :], pe[t, :] >          (reference model/crossatt.py:125-127: the second sdpa's q . k^T with q = x_pos,
//                                            k = the positional table; fp32 accumulate, ROUNDED to the model dtype as the
//                                            projection GEMM this replaces stored them)
// itself -- T_txt x d multiply-adds per workgroup (65 k at L169: nothing), pe's T_txt rows come from L2 (shared by every
// workgroup) -- instead of a launch of its own that 256 CUs wait for.  xp is read row-major or fragment-major (xp_packed).
template <typename T>
__global__ __launch_bounds__(256) void pe_softmax_weighted_rows_kernel(const T* __restrict__ xp, int xp_packed,
                                                                       const T* __restrict__ pe, float scale,
                                                                       T* __restrict__ att, int64_t att_sb,
                                                                       const T* __restrict__ vv, T* x, int Tn, int d, T* xpk,
                                                                       const int64_t* __restrict__ att_step, int64_t att_ss,
                                                                       int64_t att_ns) {
    LINA_DYN_SMEM(smem);
    float* s_x = reinterpret_cast<float*>(smem);              // [d]: the row of x_pos in fp32
    __shared__ float s_a[kCaMaxT];
    __shared__ float s_red[4];
    __shared__ __attribute__((aligned(16))) float s_p[3][64][4];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int e = 4 * tid; e < d; e += 1024) {
        const float4 v4 = ld4(xp_packed ? xp + packed_off<T>(b, e, d) : xp + (int64_t)b * d + e);
        *reinterpret_cast<float4*>(&s_x[e]) = v4;
    }
    __syncthreads();
    // wave w takes the text positions w, w+4, ...: a lane covers the columns 4 lane + 256 c; four positions in flight
    const int nc = d / 256;                                   // (launcher: d % 256 == 0)
    for (int t0 = w; t0 < Tn; t0 += 16) {
        float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = min(t0 + 4 * u, Tn - 1);
            const T* row = pe + (int64_t)t * d + 4 * lane;
            for (int c = 0; c < nc; ++c) {
                const float4 pv = ld4(row + 256 * c);
                const float4 xv = *reinterpret_cast<const float4*>(&s_x[4 * lane + 256 * c]);
                part[u] = fmaf(pv.x, xv.x, part[u]); part[u] = fmaf(pv.y, xv.y, part[u]);
                part[u] = fmaf(pv.z, xv.z, part[u]); part[u] = fmaf(pv.w, xv.w, part[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float a = part[u];
            a += shfl_xor(a, 1); a += shfl_xor(a, 2); a += shfl_xor(a, 4);
            a += shfl_xor(a, 8); a += shfl_xor(a, 16); a += shfl_xor(a, 32);
            if (lane == 0 && t0 + 4 * u < Tn) {
                T tmp;                                        // the projection's output dtype
                st(&tmp, a);
                s_a[t0 + 4 * u] = ld(&tmp) * scale;
            }
        }
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int t = tid; t < Tn; t += 256) mx = fmaxf(mx, s_a[t]);
    mx = block_max(mx, s_red);
    float sum = 0.0f;
    for (int t = tid; t < Tn; t += 256) sum += expf(s_a[t] - mx);
    sum = block_sum(sum, s_red);
    const float inv = 1.0f / sum;
    // att log of the device-side decode loop: the row goes to att + step[0] * att_ss (dropped outside [0, att_ns))
    const int64_t a_t = att_step ? att_step[0] : 0;
    const bool a_ok = blockIdx.x == 0 && (!att_step || (a_t >= 0 && a_t < att_ns));
    for (int t = tid; t < Tn; t += 256) {
        T tmp;                                               // the weights in the model dtype, as softmax_rows stores them
        st(&tmp, expf(s_a[t] - mx) * inv);
        s_a[t] = ld(&tmp);
        if (a_ok) att[b * att_sb + a_t * att_ss + t] = tmp;
    }
    __syncthreads();
    // the row's 256-column slabs: one per workgroup (grid.x = d / 256: small batches, more workgroups) or all of them in turn
    // (grid.x = 1: at B >= 256 the T_txt x d scores above -- pe's 128 KB from L2 and 65 k multiply-adds per workgroup -- were
    // recomputed by four workgroups per row and were most of the launch's 37 us at B = 512; profiles/r05_step_timeline.txt)
    const int nslab = (d + 255) / 256;
    for (int sb = blockIdx.x; sb < nslab; sb += gridDim.x) {
        const int e = sb * 256 + lane * 4;
        const bool live = e < d;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const T* base = vv + (int64_t)b * Tn * d + (live ? e : 0);
        for (int t0 = w; t0 < Tn; t0 += 32) {
            float4 p[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) p[u] = ld4(base + (int64_t)min(t0 + 4 * u, Tn - 1) * d);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float a = (t0 + 4 * u < Tn) ? s_a[min(t0 + 4 * u, Tn - 1)] : 0.0f;
                acc.x = fmaf(a, p[u].x, acc.x); acc.y = fmaf(a, p[u].y, acc.y);
                acc.z = fmaf(a, p[u].z, acc.z); acc.w = fmaf(a, p[u].w, acc.w);
            }
        }
        if (w > 0) *reinterpret_cast<float4*>(&s_p[w - 1][lane][0]) = acc;
        __syncthreads();
        if (w == 0 && live) {
#pragma unroll
            for (int ww = 0; ww < 3; ++ww) {
                const float4 o = *reinterpret_cast<const float4*>(&s_p[ww][lane][0]);
                acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
            }
            T tmp4[4];
            st4(tmp4, acc);
            T* const xa = xpk ? xpk + packed_off<T>(b, e, d) : x + (int64_t)b * d + e;
            const float4 o = ld4(tmp4), r = ld4(xa);
            st4(xa, make_float4(r.x + o.x, r.y + o.y, r.z + o.z, r.w + o.w));
        }
        __syncthreads();                                      // the partial sums have been read: the next slab may write them
    }
}

// softmax + att . pe in ONE launch (the first half of the blind cross-attention, reference model/crossatt.py:117-127):
//   att[b, :Tn] = softmax(scores[b, :Tn])  (scores: fp32, already scaled -- lina_cross_scores' output);
//   xp[b, :]    = att[b, :] . pe[:Tn, :]    (pe shared by all rows; att rounded to the model dtype first, as the reference's
//                                            bmm sees it)  -> row-major xp and, optionally, its fragment-major copy.
// grid (ceil(d / 256), B) like softmax_weighted_rows_kernel: with lina_cross_scores in front (256 workgroups on the text
// keys) this replaces {scores + softmax in ONE 1024-thread workgroup per row, then a K = T_txt skinny GEMM}: the row-wide
// workgroups of that form put 64 CUs on 8.4 MB of HBM-cold keys (11.5 us per launch in the step's timeline).
template <typename T>
__global__ __launch_bounds__(256) void softmax_pe_rows_kernel(const float* __restrict__ scores, int64_t sc_sb,
                                                              T* __restrict__ att, int64_t att_sb, const T* __restrict__ pe,
                                                              T* __restrict__ xp, T* __restrict__ xpk, int Tn, int d,
                                                              const int64_t* __restrict__ att_step, int64_t att_ss,
                                                              int64_t att_ns) {
    __shared__ float s_a[kCaMaxT];
    __shared__ float s_red[4];
    __shared__ __attribute__((aligned(16))) float s_p[3][64][4];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int e = blockIdx.x * 256 + lane * 4;
    const bool live = e < d;
    // the scores first (a wave's loads return in order), then the first pe rows of this wave: they do not depend on the softmax
    const T* base = pe + (live ? e : 0);
    const float sc0 = tid < Tn ? scores[b * sc_sb + tid] : -INFINITY;
    float4 p0[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) p0[u] = ld4(base + (int64_t)min(w + 4 * u, Tn - 1) * d);
    float mx = -INFINITY;
    for (int t = tid; t < Tn; t += 256) { const float v = t == tid ? sc0 : scores[b * sc_sb + t]; s_a[t] = v; mx = fmaxf(mx, v); }
    mx = block_max(mx, s_red);
    float sum = 0.0f;
    for (int t = tid; t < Tn; t += 256) sum += expf(s_a[t] - mx);
    sum = block_sum(sum, s_red);
    const float inv = 1.0f / sum;
    // att log of the device-side decode loop: the row goes to att + step[0] * att_ss (dropped outside [0, att_ns))
    const int64_t a_t = att_step ? att_step[0] : 0;
    const bool a_ok = blockIdx.x == 0 && (!att_step || (a_t >= 0 && a_t < att_ns));
    for (int t = tid; t < Tn; t += 256) {
        T tmp;                                               // the weights in the model dtype, as softmax_rows stores them
        st(&tmp, expf(s_a[t] - mx) * inv);
        s_a[t] = ld(&tmp);
        if (a_ok) att[b * att_sb + a_t * att_ss + t] = tmp;
    }
    __syncthreads();
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t0 = w; t0 < Tn; t0 += 32) {
        float4 p[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) p[u] = t0 == w ? p0[u] : ld4(base + (int64_t)min(t0 + 4 * u, Tn - 1) * d);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float a = (t0 + 4 * u < Tn) ? s_a[min(t0 + 4 * u, Tn - 1)] : 0.0f;
            acc.x = fmaf(a, p[u].x, acc.x); acc.y = fmaf(a, p[u].y, acc.y);
            acc.z = fmaf(a, p[u].z, acc.z); acc.w = fmaf(a, p[u].w, acc.w);
        }
    }
    if (w > 0) *reinterpret_cast<float4*>(&s_p[w - 1][lane][0]) = acc;
    __syncthreads();
    if (w > 0 || !live) return;
#pragma unroll
    for (int ww = 0; ww < 3; ++ww) {
        const float4 o = *reinterpret_cast<const float4*>(&s_p[ww][lane][0]);
        acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
    }
    st4(xp + (int64_t)b * d + e, acc);
    if (xpk) {                                               // fragment-major copy (the model-dtype values just stored)
        T tmp4[4];
        st4(tmp4, acc);
        st4(xpk + packed_off<T>(b, e, d), ld4(tmp4));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void cross_att_step1_kernel(
    const T* __restrict__ qlin, const T* __restrict__ ln_w, const T* __restrict__ ln_b, float ln_eps,
    const T* __restrict__ kk, const T* __restrict__ pe, T* __restrict__ att1, int64_t att_sb, T* __restrict__ xp,
    int Tn, int d, float scale) {
    LINA_DYN_SMEM(smem);
    float* s_q = reinterpret_cast<float*>(smem);            // [d]
    float* s_sc = s_q + d;                                  // [Tn]
    __shared__ float s_red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    // LayerNorm of the projected query (two-pass, fp32)
    float s = 0.0f;
    for (int e = tid; e < d; e += 256) { const float x = ld(qlin + (int64_t)b * d + e); s_q[e] = x; s += x; }
    const float mu = block_sum(s, s_red) / (float)d;
    float vs = 0.0f;
    for (int e = tid; e < d; e += 256) { const float c = s_q[e] - mu; vs += c * c; }
    const float rstd = rsqrtf(block_sum(vs, s_red) / (float)d + ln_eps);
    for (int e = tid; e < d; e += 256) {
        T tmp;                                               // the reference rounds the LN output to the model dtype
        st(&tmp, (s_q[e] - mu) * rstd * ld(ln_w + e) + ld(ln_b + e));
        s_q[e] = ld(&tmp);
    }
    __syncthreads();
    scores_softmax<T>(s_q, kk + (int64_t)b * Tn * d, d, Tn, d, scale, s_sc, s_red);
    for (int t = tid; t < Tn; t += 256) {
        T tmp;
        st(&tmp, s_sc[t]);
        att1[(int64_t)b * att_sb + t] = tmp;
        s_sc[t] = ld(&tmp);                                  // att1 is used in the model dtype downstream
    }
    __syncthreads();
    for (int e = tid * 4; e < d; e += 1024) {               // xp = att1 . PE   (column-parallel)
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int t = 0; t < Tn; ++t) {
            const float a = s_sc[t];
            const float4 p = ld4(pe + (int64_t)t * d + e);
            acc.x = fmaf(a, p.x, acc.x); acc.y = fmaf(a, p.y, acc.y); acc.z = fmaf(a, p.z, acc.z); acc.w = fmaf(a, p.w, acc.w);
        }
        st4(xp + (int64_t)b * d + e, acc);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void cross_att_step2_kernel(
    const T* __restrict__ xp, const T* __restrict__ pe, const T* __restrict__ vv, T* __restrict__ att2,
    int64_t att_sb, T* x, int Tn, int d, float scale) {
    LINA_DYN_SMEM(smem);
    float* s_q = reinterpret_cast<float*>(smem);
    float* s_sc = s_q + d;
    __shared__ float s_red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int e = tid; e < d; e += 256) s_q[e] = ld(xp + (int64_t)b * d + e);
    __syncthreads();
    scores_softmax<T>(s_q, pe, d, Tn, d, scale, s_sc, s_red);
    for (int t = tid; t < Tn; t += 256) {
        T tmp;
        st(&tmp, s_sc[t]);
        att2[(int64_t)b * att_sb + t] = tmp;
        s_sc[t] = ld(&tmp);
    }
    __syncthreads();
    for (int e = tid * 4; e < d; e += 1024) {               // x += att2 . V_b
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int t = 0; t < Tn; ++t) {
            const float a = s_sc[t];
            const float4 p = ld4(vv + ((int64_t)b * Tn + t) * d + e);
            acc.x = fmaf(a, p.x, acc.x); acc.y = fmaf(a, p.y, acc.y); acc.z = fmaf(a, p.z, acc.z); acc.w = fmaf(a, p.w, acc.w);
        }
        T tmp4[4];
        st4(tmp4, acc);                                      // bmm result in the model dtype, then the residual add
        const float4 o = ld4(tmp4), r = ld4(x + (int64_t)b * d + e);
        st4(x + (int64_t)b * d + e, make_float4(r.x + o.x, r.y + o.y, r.z + o.z, r.w + o.w));
    }
}

}  // namespace lina

extern "C" int lina_cross_att_step1(const void* q_lin, const void* ln_w, const void* ln_b, float ln_eps, const void* kk,
                                    const void* pe, void* att1, int64_t att_sb, void* xp, int B, int Tn, int d,
                                    float scale, int dtype, lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(q_lin && ln_w && ln_b && kk && pe && att1 && xp, "lina_cross_att_step1: null pointer");
    LINA_REQUIRE(B > 0 && Tn > 0 && Tn <= kCaMaxT, "lina_cross_att_step1: B > 0, 0 < T_txt <= %d", kCaMaxT);
    LINA_REQUIRE(d > 0 && d % 4 == 0 && d <= 8192, "lina_cross_att_step1: d must be a multiple of 4, <= 8192");
    LINA_REQUIRE(valid_dtype(dtype), "lina_cross_att_step1: bad dtype %d", dtype);
    const size_t smem = sizeof(float) * (size_t)(d + Tn);
    if (dtype == LINA_F32)
        LINA_LAUNCH((cross_att_step1_kernel<float>), dim3((unsigned)B), dim3(256), smem, stream, (const float*)q_lin,
                    (const float*)ln_w, (const float*)ln_b, ln_eps, (const float*)kk, (const float*)pe, (float*)att1,
                    att_sb, (float*)xp, Tn, d, scale);
    else
        LINA_LAUNCH((cross_att_step1_kernel<bf16_t>), dim3((unsigned)B), dim3(256), smem, stream, (const bf16_t*)q_lin,
                    (const bf16_t*)ln_w, (const bf16_t*)ln_b, ln_eps, (const bf16_t*)kk, (const bf16_t*)pe,
                    (bf16_t*)att1, att_sb, (bf16_t*)xp, Tn, d, scale);
    return check_launch("lina_cross_att_step1");
}

extern "C" int lina_cross_att_step2(const void* xp, const void* pe, const void* vv, void* att2, int64_t att_sb, void* x,
                                    int B, int Tn, int d, float scale, int dtype, lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(xp && pe && vv && att2 && x, "lina_cross_att_step2: null pointer");
    LINA_REQUIRE(B > 0 && Tn > 0 && Tn <= kCaMaxT, "lina_cross_att_step2: B > 0, 0 < T_txt <= %d", kCaMaxT);
    LINA_REQUIRE(d > 0 && d % 4 == 0 && d <= 8192, "lina_cross_att_step2: d must be a multiple of 4, <= 8192");
    LINA_REQUIRE(valid_dtype(dtype), "lina_cross_att_step2: bad dtype %d", dtype);
    const size_t smem = sizeof(float) * (size_t)(d + Tn);
    if (dtype == LINA_F32)
        LINA_LAUNCH((cross_att_step2_kernel<float>), dim3((unsigned)B), dim3(256), smem, stream, (const float*)xp,
                    (const float*)pe, (const float*)vv, (float*)att2, att_sb, (float*)x, Tn, d, scale);
    else
        LINA_LAUNCH((cross_att_step2_kernel<bf16_t>), dim3((unsigned)B), dim3(256), smem, stream, (const bf16_t*)xp,
                    (const bf16_t*)pe, (const bf16_t*)vv, (bf16_t*)att2, att_sb, (bf16_t*)x, Tn, d, scale);
    return check_launch("lina_cross_att_step2");
}

extern "C" int lina_cross_scores(const void* q_lin, const void* ln_w, const void* ln_b, float ln_eps, const void* kk,
                                 float* scores, int B, int Tn, int d, float scale, int dtype, lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(q_lin && ln_w && ln_b && kk && scores, "lina_cross_scores: null pointer");
    LINA_REQUIRE(B > 0 && Tn > 0, "lina_cross_scores: B, T_txt must be positive");
    LINA_REQUIRE(d > 0 && d % 4 == 0 && d <= 16384, "lina_cross_scores: d must be a multiple of 4, <= 16384");
    LINA_REQUIRE(valid_dtype(dtype), "lina_cross_scores: bad dtype %d", dtype);
    dim3 grid((unsigned)((Tn + 15) / 16), (unsigned)B);
    const size_t smem = sizeof(float) * (size_t)d;
    if (dtype == LINA_F32)
        LINA_LAUNCH((cross_scores_kernel<float>), grid, dim3(256), smem, stream, (const float*)q_lin, (const float*)ln_w,
                    (const float*)ln_b, ln_eps, (const float*)kk, scores, Tn, d, scale);
    else
        LINA_LAUNCH((cross_scores_kernel<bf16_t>), grid, dim3(256), smem, stream, (const bf16_t*)q_lin,
                    (const bf16_t*)ln_w, (const bf16_t*)ln_b, ln_eps, (const bf16_t*)kk, scores, Tn, d, scale);
    return check_launch("lina_cross_scores");
}

extern "C" int lina_softmax_rows(const void* x, int64_t x_sb, int x_dtype, float scale, void* att, int64_t att_sb,
                                 void* attc, int B, int Tn, int Tp, int dtype, lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(x && att && attc, "lina_softmax_rows: null pointer");
    LINA_REQUIRE(B > 0 && Tn > 0 && Tp >= Tn, "lina_softmax_rows: bad shape");
    LINA_REQUIRE(valid_dtype(dtype) && valid_dtype(x_dtype), "lina_softmax_rows: bad dtype");
    dim3 grid((unsigned)((B + 3) / 4));
    if (x_dtype == LINA_F32 && dtype == LINA_F32)
        LINA_LAUNCH((softmax_rows_kernel<float, float>), grid, dim3(256), 0, stream, (const float*)x, x_sb, scale,
                    (float*)att, att_sb, (float*)attc, B, Tn, Tp);
    else if (x_dtype == LINA_F32 && dtype == LINA_BF16)
        LINA_LAUNCH((softmax_rows_kernel<float, bf16_t>), grid, dim3(256), 0, stream, (const float*)x, x_sb, scale,
                    (bf16_t*)att, att_sb, (bf16_t*)attc, B, Tn, Tp);
    else if (x_dtype == LINA_BF16 && dtype == LINA_BF16)
        LINA_LAUNCH((softmax_rows_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, stream, (const bf16_t*)x, x_sb, scale,
                    (bf16_t*)att, att_sb, (bf16_t*)attc, B, Tn, Tp);
    else
        return fail(LINA_ERR_UNSUPPORTED, "lina_softmax_rows: bf16 input with f32 output is not built");
    return check_launch("lina_softmax_rows");
}

static int weighted_rows_impl(const void* attc, int Tp, const void* vv, void* x, void* x_packed, int B, int Tn, int d,
                              int dtype, lina_stream_t stream);
extern "C" int lina_weighted_rows_add(const void* attc, int Tp, const void* vv, void* x, int B, int Tn, int d, int dtype,
                                      lina_stream_t stream) {
    return weighted_rows_impl(attc, Tp, vv, x, nullptr, B, Tn, d, dtype, stream);
}
extern "C" int lina_weighted_rows_add_packed(const void* attc, int Tp, const void* vv, void* x, void* x_packed, int B,
                                             int Tn, int d, int dtype, lina_stream_t stream) {
    return weighted_rows_impl(attc, Tp, vv, x, x_packed, B, Tn, d, dtype, stream);
}
static int weighted_rows_impl(const void* attc, int Tp, const void* vv, void* x, void* x_packed, int B, int Tn, int d,
                              int dtype, lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(!x_packed || d % (dtype == LINA_BF16 ? 32 : 16) == 0, "lina_weighted_rows_add: packed copy needs whole k-steps");
    LINA_REQUIRE(attc && vv && x, "lina_weighted_rows_add: null pointer");
    LINA_REQUIRE(B > 0 && Tn > 0 && Tn <= kCaMaxT && Tp >= Tn, "lina_weighted_rows_add: 0 < T_txt <= %d", kCaMaxT);
    LINA_REQUIRE(d > 0 && d % 4 == 0, "lina_weighted_rows_add: d must be a multiple of 4");
    LINA_REQUIRE(valid_dtype(dtype), "lina_weighted_rows_add: bad dtype %d", dtype);
    dim3 grid((unsigned)((d + 255) / 256), (unsigned)B);
    if (dtype == LINA_F32)
        LINA_LAUNCH((weighted_rows_kernel<float>), grid, dim3(256), 0, stream, (const float*)attc, Tp, (const float*)vv,
                    (float*)x, Tn, d, (float*)x_packed);
    else
        LINA_LAUNCH((weighted_rows_kernel<bf16_t>), grid, dim3(256), 0, stream, (const bf16_t*)attc, Tp,
                    (const bf16_t*)vv, (bf16_t*)x, Tn, d, (bf16_t*)x_packed);
    return check_launch("lina_weighted_rows_add");
}

extern "C" int lina_cross_scores_softmax(const void* q_lin, const void* ln_w, const void* ln_b, float ln_eps,
                                         const void* kk, void* att, int64_t att_sb, void* attc, int B, int Tn, int Tp,
                                         int d, float scale, int dtype, lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(q_lin && ln_w && ln_b && kk && att && attc, "lina_cross_scores_softmax: null pointer");
    LINA_REQUIRE(B > 0 && Tn > 0 && Tn <= kCaMaxT && Tp >= Tn, "lina_cross_scores_softmax: 0 < T_txt <= %d", kCaMaxT);
    LINA_REQUIRE(d > 0 && d % 4 == 0 && d <= 8192, "lina_cross_scores_softmax: d must be a multiple of 4, <= 8192");
    LINA_REQUIRE(valid_dtype(dtype), "lina_cross_scores_softmax: bad dtype %d", dtype);
    const size_t smem = sizeof(float) * ((size_t)d + (size_t)Tn);
    if (dtype == LINA_F32)
        LINA_LAUNCH((cross_scores_softmax_kernel<float>), dim3((unsigned)B), dim3(1024), smem, stream, (const float*)q_lin,
                    (const float*)ln_w, (const float*)ln_b, ln_eps, (const float*)kk, (float*)att, att_sb, (float*)attc, Tn,
                    Tp, d, scale);
    else
        LINA_LAUNCH((cross_scores_softmax_kernel<bf16_t>), dim3((unsigned)B), dim3(1024), smem, stream, (const bf16_t*)q_lin,
                    (const bf16_t*)ln_w, (const bf16_t*)ln_b, ln_eps, (const bf16_t*)kk, (bf16_t*)att, att_sb,
                    (bf16_t*)attc, Tn, Tp, d, scale);
    return check_launch("lina_cross_scores_softmax");
}

extern "C" int lina_softmax_pe_rows(const float* scores, int64_t scores_sb, void* att, int64_t att_sb,
                                    const int64_t* att_step, int64_t att_step_stride, int64_t att_steps, const void* pe, void* xp,
                                   void* xp_packed, int B, int Tn, int d, int dtype, lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(scores && att && pe && xp, "lina_softmax_pe_rows: null pointer");
    LINA_REQUIRE(B > 0 && Tn > 0 && Tn <= kCaMaxT, "lina_softmax_pe_rows: 0 < T_txt <= %d", kCaMaxT);
    LINA_REQUIRE(d > 0 && d % 4 == 0, "lina_softmax_pe_rows: d must be a multiple of 4");
    LINA_REQUIRE(valid_dtype(dtype), "lina_softmax_pe_rows: bad dtype %d", dtype);
    LINA_REQUIRE(!xp_packed || d % (dtype == LINA_BF16 ? 32 : 16) == 0, "lina_softmax_pe_rows: packed copy needs whole k-steps");
    dim3 grid((unsigned)((d + 255) / 256), (unsigned)B);
    if (dtype == LINA_F32)
        LINA_LAUNCH((softmax_pe_rows_kernel<float>), grid, dim3(256), 0, stream, scores, scores_sb, (float*)att, att_sb,
                    (const float*)pe, (float*)xp, (float*)xp_packed, Tn, d, att_step, att_step_stride, att_steps);
    else
        LINA_LAUNCH((softmax_pe_rows_kernel<bf16_t>), grid, dim3(256), 0, stream, scores, scores_sb, (bf16_t*)att, att_sb,
                    (const bf16_t*)pe, (bf16_t*)xp, (bf16_t*)xp_packed, Tn, d, att_step, att_step_stride, att_steps);
    return check_launch("lina_softmax_pe_rows");
}

extern "C" int lina_softmax_weighted_rows_add(const void* scores, int64_t scores_sb, float scale, void* att, int64_t att_sb,
                                              const void* vv, void* x, void* x_packed, int B, int Tn, int d, int dtype,
                                              lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(scores && att && vv && (x || x_packed), "lina_softmax_weighted_rows_add: null pointer");
    LINA_REQUIRE(B > 0 && Tn > 0 && Tn <= kCaMaxT, "lina_softmax_weighted_rows_add: 0 < T_txt <= %d", kCaMaxT);
    LINA_REQUIRE(d > 0 && d % 4 == 0, "lina_softmax_weighted_rows_add: d must be a multiple of 4");
    LINA_REQUIRE(valid_dtype(dtype), "lina_softmax_weighted_rows_add: bad dtype %d", dtype);
    LINA_REQUIRE(!x_packed || d % (dtype == LINA_BF16 ? 32 : 16) == 0, "lina_softmax_weighted_rows_add: packed x needs whole k-steps");
    dim3 grid((unsigned)((d + 255) / 256), (unsigned)B);
    if (dtype == LINA_F32)
        LINA_LAUNCH((softmax_weighted_rows_kernel<float>), grid, dim3(256), 0, stream, (const float*)scores, scores_sb, scale,
                    (float*)att, att_sb, (const float*)vv, (float*)x, Tn, d, (float*)x_packed);
    else
        LINA_LAUNCH((softmax_weighted_rows_kernel<bf16_t>), grid, dim3(256), 0, stream, (const bf16_t*)scores, scores_sb,
                    scale, (bf16_t*)att, att_sb, (const bf16_t*)vv, (bf16_t*)x, Tn, d, (bf16_t*)x_packed);
    return check_launch("lina_softmax_weighted_rows_add");
}

extern "C" int lina_pe_softmax_weighted_rows_add(const void* xp, int xp_packed, const void* pe, float scale, void* att,
                                                 int64_t att_sb, const int64_t* att_step, int64_t att_step_stride,
                                                 int64_t att_steps, const void* vv, void* x, void* x_packed, int B, int Tn, int d,
                                                 int dtype, lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(xp && pe && att && vv && (x || x_packed), "lina_pe_softmax_weighted_rows_add: null pointer");
    LINA_REQUIRE(B > 0 && Tn > 0 && Tn <= kCaMaxT, "lina_pe_softmax_weighted_rows_add: 0 < T_txt <= %d", kCaMaxT);
    LINA_REQUIRE(d > 0 && d % 256 == 0 && d <= 8192, "lina_pe_softmax_weighted_rows_add: d must be a multiple of 256 (<= 8192)");
    LINA_REQUIRE(valid_dtype(dtype), "lina_pe_softmax_weighted_rows_add: bad dtype %d", dtype);
    dim3 grid((unsigned)(B >= 256 ? 1 : (d + 255) / 256), (unsigned)B);   // B >= 256: one workgroup per row takes every slab
    const size_t smem = sizeof(float) * (size_t)d;
    if (dtype == LINA_F32)
        LINA_LAUNCH((pe_softmax_weighted_rows_kernel<float>), grid, dim3(256), smem, stream, (const float*)xp, xp_packed,
                    (const float*)pe, scale, (float*)att, att_sb, (const float*)vv, (float*)x, Tn, d, (float*)x_packed,
                    att_step, att_step_stride, att_steps);
    else
        LINA_LAUNCH((pe_softmax_weighted_rows_kernel<bf16_t>), grid, dim3(256), smem, stream, (const bf16_t*)xp, xp_packed,
                    (const bf16_t*)pe, scale, (bf16_t*)att, att_sb, (const bf16_t*)vv, (bf16_t*)x, Tn, d,
                    (bf16_t*)x_packed, att_step, att_step_stride, att_steps);
    return check_launch("lina_pe_softmax_weighted_rows_add");
}

#ifdef LINA_SKINNY_PROF
extern "C" int lina_cross_prof_read(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(lina::lina_cross_prof), sizeof(unsigned long long) * 1024 * 8);
}
#endif
