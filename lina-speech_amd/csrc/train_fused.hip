// train_fused.hip -- K10 / K11: the elementwise glue of the TRAINING step around the mixer:
//   K10   x' = x (+ r);  y = LayerNorm(x') * gamma + beta        forward and backward   (reference model/base_blocks.py:65-69:
//         `x = tmix(norm1(x)) + x; x = cmix(norm2(x)) + x` -- the residual add that precedes a norm rides in the norm's pass)
//   K11b  backward of the SwiGLU gate  s = silu(a) * b                                    (reference model/base_blocks.py:48-50)
// Why: in the L169 train step (b = 8 x 4096 tokens, bf16 autocast) the vendor LayerNorm runs in fp32 with separate cast
// and residual-add passes, 35 % of the step was torch elementwise / LayerNorm / cast kernels
// (profiles/r02r_train_step_kernel_stats.csv).  These kernels read the fp32 residual stream once and write the model-dtype
// operand of the next GEMM directly; the backward forms dx, the residual's pass-through gradient and per-workgroup
// partials of dgamma / dbeta (summed by the caller: deterministic, no atomics) in one pass.
// Work split: one wave per row (a row's D elements are contiguous: 16 B per lane and piece), 4 rows per 256-thread
// workgroup, grid-stride over the rows; all reductions are wave shuffles.  HBM-bound streaming kernels.
#include <type_traits>
#include <lina_dev.h>
#include "lina_common.h"

namespace lina {

constexpr int kLnMaxPieces = 8;        // D <= 64 lanes * 4 elements * 8 pieces = 2048

__device__ __forceinline__ float wave_sum(float v) {
    v += shfl_xor(v, 1); v += shfl_xor(v, 2); v += shfl_xor(v, 4);
    v += shfl_xor(v, 8); v += shfl_xor(v, 16); v += shfl_xor(v, 32);
    return v;
}

// TX: residual-stream dtype (x, xsum, dx);  TR: dtype of the added branch r (and of its gradient);  TY: output dtype
// All loads of a row are UNCONDITIONAL on clamped element offsets and issued before anything uses them (a load under a
// per-lane `if (e < D)` is waited for on the spot: one memory round trip per piece); the tail pieces are masked at the sums
// and at the stores.
template <typename TX, typename TR, typename TY, int NP>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const TX* __restrict__ x, const TR* __restrict__ r,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            TX* __restrict__ xsum, TY* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd, int64_t N,
                                                            int D, float eps) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float inv_d = 1.0f / (float)D;
    float4 gm[NP], bt[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int e = 4 * lane + 256 * i, ec = e < D ? e : D - 4;
        gm[i] = *reinterpret_cast<const float4*>(gamma + ec);
        bt[i] = *reinterpret_cast<const float4*>(beta + ec);
    }
    for (int64_t row = (int64_t)blockIdx.x * 4 + w; row < N; row += (int64_t)gridDim.x * 4) {
        typename raw4<TX>::type xr[NP];
        typename raw4<TR>::type rr[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int e = 4 * lane + 256 * i, ec = e < D ? e : D - 4;
            xr[i] = ld4_raw(x + row * D + ec);
            if (r) rr[i] = ld4_raw(r + row * D + ec);            // (r: a kernel argument -- scalar branch)
        }
        float4 v[NP];
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int e = 4 * lane + 256 * i;
            const bool ok = e < D;
            v[i] = cvt4(xr[i]);
            if (r) {
                const float4 a = cvt4(rr[i]);
                v[i].x += a.x; v[i].y += a.y; v[i].z += a.z; v[i].w += a.w;
                if (xsum) {                                  // the value the stream carries on (rounded to its dtype first)
                    if (ok) st4(xsum + row * D + e, v[i]);
                    if (!std::is_same<TX, float>::value) {
                        TX t4[4];
                        st4(t4, v[i]);
                        v[i] = ld4(t4);
                    }
                }
            }
            s += ok ? (v[i].x + v[i].y) + (v[i].z + v[i].w) : 0.0f;
        }
        const float mu = wave_sum(s) * inv_d;
        float q = 0.0f;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const float a = v[i].x - mu, b = v[i].y - mu, c = v[i].z - mu, d = v[i].w - mu;
            q += (4 * lane + 256 * i < D) ? (a * a + b * b) + (c * c + d * d) : 0.0f;
        }
        const float rs = rsqrtf(wave_sum(q) * inv_d + eps);
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int e = 4 * lane + 256 * i;
            if (e < D)
                st4(y + row * D + e, make_float4((v[i].x - mu) * rs * gm[i].x + bt[i].x, (v[i].y - mu) * rs * gm[i].y + bt[i].y,
                                                 (v[i].z - mu) * rs * gm[i].z + bt[i].z, (v[i].w - mu) * rs * gm[i].w + bt[i].w));
        }
    }
}

// dx = dpass + rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma,  xhat = (x - mean) * rstd
// dpass (optional): the gradient that reaches x' through the residual path; dr (optional): the same dx in the branch's
// dtype (the gradient of the added branch r).  dgamma / dbeta partials per workgroup: [gridDim.x][D].
template <typename TX, typename TR, typename TY, int NP>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const TY* __restrict__ dy, const TX* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const TX* __restrict__ dpass,
                                                            TX* __restrict__ dx, TR* __restrict__ dr,
                                                            float* __restrict__ dg_part, float* __restrict__ db_part,
                                                            int64_t N, int D) {
    __shared__ __attribute__((aligned(16))) float s_part[2][3][64][4];       // [dgamma|dbeta][waves 1..3][lane][4]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float inv_d = 1.0f / (float)D;
    float4 ag[NP], ab[NP], gm[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int e = 4 * lane + 256 * i;
        ag[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        ab[i] = ag[i];
        gm[i] = *reinterpret_cast<const float4*>(gamma + (e < D ? e : D - 4));
    }
    for (int64_t row = (int64_t)blockIdx.x * 4 + w; row < N; row += (int64_t)gridDim.x * 4) {
        typename raw4<TY>::type dr_[NP];
        typename raw4<TX>::type xr[NP], pr[NP];
        const float mu = mean[row], rs = rstd[row];
#pragma unroll
        for (int i = 0; i < NP; ++i) {                       // every load of the row first (unconditional, clamped)
            const int e = 4 * lane + 256 * i, ec = e < D ? e : D - 4;
            dr_[i] = ld4_raw(dy + row * D + ec);
            xr[i] = ld4_raw(x + row * D + ec);
            if (dpass) pr[i] = ld4_raw(dpass + row * D + ec);
        }
        float4 g[NP], xh[NP];
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const bool ok = 4 * lane + 256 * i < D;
            const float4 d0 = cvt4(dr_[i]), xv = cvt4(xr[i]);
            const float4 d = ok ? d0 : make_float4(0.f, 0.f, 0.f, 0.f);
            xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
            ag[i].x += d.x * xh[i].x; ag[i].y += d.y * xh[i].y; ag[i].z += d.z * xh[i].z; ag[i].w += d.w * xh[i].w;
            ab[i].x += d.x; ab[i].y += d.y; ab[i].z += d.z; ab[i].w += d.w;
            g[i] = make_float4(d.x * gm[i].x, d.y * gm[i].y, d.z * gm[i].z, d.w * gm[i].w);
            s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
            s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
        }
        const float m1 = wave_sum(s1) * inv_d, m2 = wave_sum(s2) * inv_d;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int e = 4 * lane + 256 * i;
            float4 o = make_float4(rs * (g[i].x - m1 - xh[i].x * m2), rs * (g[i].y - m1 - xh[i].y * m2),
                                   rs * (g[i].z - m1 - xh[i].z * m2), rs * (g[i].w - m1 - xh[i].w * m2));
            if (dpass) {
                const float4 p = cvt4(pr[i]);
                o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
            }
            if (e < D) {
                st4(dx + row * D + e, o);
                if (dr) st4(dr + row * D + e, o);
            }
        }
    }
    // dgamma / dbeta of this workgroup's rows: waves 1..3 hand their sums to wave 0 through LDS, piece by piece
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        if (256 * i >= D) break;                             // workgroup-uniform (the barriers below)
        __syncthreads();
        if (w > 0) {
            *reinterpret_cast<float4*>(&s_part[0][w - 1][lane][0]) = ag[i];
            *reinterpret_cast<float4*>(&s_part[1][w - 1][lane][0]) = ab[i];
        }
        __syncthreads();
        if (w == 0) {
            const int e = 4 * lane + 256 * i;
            float4 a = ag[i], b = ab[i];
#pragma unroll
            for (int ww = 0; ww < 3; ++ww) {
                const float4 pa = *reinterpret_cast<const float4*>(&s_part[0][ww][lane][0]);
                const float4 pb = *reinterpret_cast<const float4*>(&s_part[1][ww][lane][0]);
                a.x += pa.x; a.y += pa.y; a.z += pa.z; a.w += pa.w;
                b.x += pb.x; b.y += pb.y; b.z += pb.z; b.w += pb.w;
            }
            if (e < D) {
                *reinterpret_cast<float4*>(dg_part + (int64_t)blockIdx.x * D + e) = a;
                *reinterpret_cast<float4*>(db_part + (int64_t)blockIdx.x * D + e) = b;
            }
        }
    }
}

// the value a store of v to T leaves in memory
template <typename T> __device__ __forceinline__ float stored_as(float v);
template <> __device__ __forceinline__ float stored_as<float>(float v) { return v; }
template <> __device__ __forceinline__ float stored_as<bf16_t>(float v) { return bf2f(f2bf(v)); }

// scalar form for widths / strides that are not multiples of 4 (L169: Hd = 1365): one element per thread
template <typename T>
__global__ __launch_bounds__(256) void swiglu_bwd_scalar_kernel(const T* __restrict__ ds, const T* __restrict__ u,
                                                                T* __restrict__ du, int64_t rows, int Hd, int64_t ld_u,
                                                                int64_t ld_ds, int64_t ld_du) {
    const int64_t n = rows * Hd;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / Hd;
        const int j = (int)(i % Hd);
        const float a = ld(u + row * ld_u + j), b = ld(u + row * ld_u + Hd + j), d = ld(ds + row * ld_ds + j);
        const float sg = sigmoidf(a);
        st(du + row * ld_du + j, d * b * sg * (1.0f + a * (1.0f - sg)));
        st(du + row * ld_du + Hd + j, d * a * sg);
    }
}

// ds [rows, Hd], u [rows, 2 Hd] (row stride ld_u)  ->  du [rows, 2 Hd]:  d/da (silu(a) b) = b sig(a) (1 + a (1 - sig(a))),
// d/db = silu(a)
template <typename T>
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const T* __restrict__ ds, const T* __restrict__ u, T* __restrict__ du,
                                                         int64_t rows, int Hd, int64_t ld_u, int64_t ld_ds, int64_t ld_du) {
    const int64_t n4 = (int64_t)rows * (Hd / 4);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / (Hd / 4);
        const int j = (int)(i % (Hd / 4)) * 4;
        const float4 a = ld4(u + row * ld_u + j), b = ld4(u + row * ld_u + Hd + j), d = ld4(ds + row * ld_ds + j);
        float4 da, db;
        const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w}, dv[4] = {d.x, d.y, d.z, d.w};
        float oa[4], ob[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float sg = sigmoidf(av[c]);
            oa[c] = dv[c] * bv[c] * sg * (1.0f + av[c] * (1.0f - sg));
            ob[c] = dv[c] * av[c] * sg;
        }
        da = make_float4(oa[0], oa[1], oa[2], oa[3]);
        db = make_float4(ob[0], ob[1], ob[2], ob[3]);
        st4(du + row * ld_du + j, da);
        st4(du + row * ld_du + Hd + j, db);
    }
}

// K11c -- K11b plus the column sums of du (the bias gradient of the up-projection) in the same pass: a workgroup takes
// kColsumRows rows x 256 gate columns, its 4 waves a quarter of the rows each (4 at a time, all loads first), a lane 4
// columns of each half; the waves' sums meet in LDS and the workgroup writes one fp32 partial row [2 Hd] per row slab
// (summed over slabs by the caller: 256 x 2 Hd floats at L169 against the 2 x 90 MB of du a separate column sum re-read).
constexpr int kColsumRows = LINA_SWIGLU_COLSUM_ROWS;
template <typename T>
__global__ __launch_bounds__(256) void swiglu_bwd_colsum_kernel(const T* __restrict__ ds, const T* __restrict__ u, T* __restrict__ du,
                                                                float* __restrict__ colsum_partial, int64_t rows, int Hd,
                                                                int64_t ld_u, int64_t ld_ds, int64_t ld_du) {
    constexpr int U = 4;
    __shared__ float s_red[4][8][64];
    const int lane = threadIdx.x & 63, wv = wave_uniform(threadIdx.x >> 6);
    const int j = (blockIdx.y * 64 + lane) * 4;
    const bool ok = j < Hd;
    const int jc = ok ? j : Hd - 4;
    float sa[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
    const int64_t r_begin = (int64_t)blockIdx.x * kColsumRows + wv * (kColsumRows / 4);
    const int64_t r_end = r_begin + kColsumRows / 4 < rows ? r_begin + kColsumRows / 4 : rows;
    for (int64_t r0 = r_begin; r0 < r_end; r0 += U) {
        typename raw4<T>::type ar[U], br[U], dr[U];
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int64_t r = r0 + q < rows ? r0 + q : rows - 1;
            ar[q] = ld4_raw(u + r * ld_u + jc);
            br[q] = ld4_raw(u + r * ld_u + Hd + jc);
            dr[q] = ld4_raw(ds + r * ld_ds + jc);
        }
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int64_t r = r0 + q;
            if (r >= r_end) break;                                   // wave-uniform
            const float4 a = cvt4(ar[q]), b = cvt4(br[q]), d = cvt4(dr[q]);
            const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w}, dv[4] = {d.x, d.y, d.z, d.w};
            float oa[4], ob[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float sg = sigmoidf(av[c]);
                oa[c] = dv[c] * bv[c] * sg * (1.0f + av[c] * (1.0f - sg));
                ob[c] = dv[c] * av[c] * sg;
            }
            if (ok) {
                st4(du + r * ld_du + j, make_float4(oa[0], oa[1], oa[2], oa[3]));
                st4(du + r * ld_du + Hd + j, make_float4(ob[0], ob[1], ob[2], ob[3]));
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {                            // sums of the values as stored (rounded to T)
                sa[c] += stored_as<T>(oa[c]);
                sb[c] += stored_as<T>(ob[c]);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        s_red[wv][c][lane] = sa[c];
        s_red[wv][4 + c][lane] = sb[c];
    }
    __syncthreads();
    // 512 sums of this workgroup (256 columns x 2 halves): thread t -> half t / 128 ... two per thread
    for (int i = threadIdx.x; i < 512; i += 256) {
        const int half = i >> 8, col = i & 255, jj = blockIdx.y * 256 + col;
        if (jj < Hd) {
            const int k = half * 4 + (col & 3), l = col >> 2;
            colsum_partial[(int64_t)blockIdx.x * 2 * Hd + half * Hd + jj] =
                s_red[0][k][l] + s_red[1][k][l] + s_red[2][k][l] + s_red[3][k][l];
        }
    }
}

// K13a -- first level for a plain matrix: per-slab column sums of x [M, N] (the bias gradient of a projection: the column
// sum of its output gradient), fp32 partials [ceil(M / kColsumRows)][N] for K13.  Same mapping as K11c: a workgroup takes
// kColsumRows rows x 256 columns, its 4 waves a quarter of the rows each (8 at a time, all loads first), a lane 4 columns.
// (torch's two-stage reduction for this shape keeps semaphores in global memory, and on ROCm 7.2 a hipGraph REPLAY of it
// returns garbage from the second replay on -- tools/probe_graph_memset.py -- which is what broke the captured train step.)
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, float* __restrict__ partial, int64_t M, int N,
                                                     int64_t ld) {
    constexpr int U = 8;
    __shared__ float4 s_red[3][64];
    const int lane = threadIdx.x & 63, wv = wave_uniform(threadIdx.x >> 6);
    const int j = (blockIdx.y * 64 + lane) * 4;
    const bool ok = j < N;
    const int jc = ok ? j : N - 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t r_begin = (int64_t)blockIdx.x * kColsumRows + wv * (kColsumRows / 4);
    const int64_t r_end = r_begin + kColsumRows / 4 < M ? r_begin + kColsumRows / 4 : M;
    for (int64_t r0 = r_begin; r0 < r_end; r0 += U) {
        typename raw4<T>::type v[U];
#pragma unroll
        for (int q = 0; q < U; ++q) v[q] = ld4_raw(x + (r0 + q < M ? r0 + q : M - 1) * ld + jc);
        sched_fence();
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const float m = r0 + q < r_end ? 1.0f : 0.0f;               // branch-free (see K13)
            const float4 f = cvt4(v[q]);
            acc.x = fmaf(m, f.x, acc.x); acc.y = fmaf(m, f.y, acc.y); acc.z = fmaf(m, f.z, acc.z); acc.w = fmaf(m, f.w, acc.w);
        }
    }
    if (wv > 0) s_red[wv - 1][lane] = acc;
    __syncthreads();
    if (wv == 0 && ok) {
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            const float4 c = s_red[o][lane];
            acc.x += c.x; acc.y += c.y; acc.z += c.z; acc.w += c.w;
        }
        *reinterpret_cast<float4*>(partial + (int64_t)blockIdx.x * N + j) = acc;
    }
}

// K13 -- second level of the two-level parameter-gradient sums: out[o][n] = sum_p part[o][p][n] (fp32 partials written by
// K3b / K5b / K10b / K11c / K12b, one row per workgroup of those kernels).  These inputs are small (1-30 MB) and the sum is
// LATENCY-bound: what matters is how many dependent load rounds a wave makes and how many CUs take part.  A workgroup of 16
// waves takes CL x 4 columns (CL = 64, 32, 16 or 8 lanes along the columns -- the launcher narrows the workgroup until the grid
// has ~64 of them: at CL = 64 the LayerNorm's [2][1024][1024] partials ran on 8 CUs, the norm-gate's [1024][256] on one); the
// other 64 / CL lane groups and the 16 waves split the partial rows, QR rows requested per round.  The lane groups meet by
// shuffles (fixed order), the waves in LDS.  (First form: 4 waves, 8 rows per round -- 16 rounds at P = 512, 40-100 us, slower
// than torch's generic reduction at 12-23 us; second form: CL = 64 only, 9-20 us per launch, 125 launches per train step.)
constexpr int kSumWaves = 16;
template <typename TO, int CL, int QR>
__global__ __launch_bounds__(64 * kSumWaves) void sum_partials_kernel(const float* __restrict__ part, TO* __restrict__ out, int P,
                                                                      int64_t N) {
    constexpr int RL = 64 / CL;                                       // lane groups along the partial rows
    __shared__ float4 s_red[kSumWaves - 1][CL];
    const int lane = threadIdx.x & 63, wv = wave_uniform(threadIdx.x >> 6);
    const int cl = lane % CL, rl = lane / CL;
    const int64_t n = ((int64_t)blockIdx.x * CL + cl) * 4;
    const bool ok = n < N;
    const float* src = part + (int64_t)blockIdx.y * P * N + (ok ? n : N - 4);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int kStride = kSumWaves * RL;                           // row lanes of the workgroup
    for (int p0 = wv * RL + rl; p0 < P; p0 += kStride * QR) {
        float4 v[QR];
#pragma unroll
        for (int q = 0; q < QR; ++q) {
            const int p = p0 + kStride * q;
            v[q] = *reinterpret_cast<const float4*>(src + (int64_t)(p < P ? p : P - 1) * N);
        }
        sched_fence();                                               // every load of the round before the first use
#pragma unroll
        for (int q = 0; q < QR; ++q) {
            // no branch here: a `break` made the compiler fold each load into its own load / wait / add / branch chain
            const float m = p0 + kStride * q < P ? 1.0f : 0.0f;
            acc.x = fmaf(m, v[q].x, acc.x); acc.y = fmaf(m, v[q].y, acc.y);
            acc.z = fmaf(m, v[q].z, acc.z); acc.w = fmaf(m, v[q].w, acc.w);
        }
    }
#pragma unroll
    for (int off = CL; off < 64; off <<= 1) {                         // the RL lane groups of the wave
        acc.x += shfl_xor(acc.x, off); acc.y += shfl_xor(acc.y, off);
        acc.z += shfl_xor(acc.z, off); acc.w += shfl_xor(acc.w, off);
    }
    if (wv > 0 && rl == 0) s_red[wv - 1][cl] = acc;
    __syncthreads();
    if (wv == 0 && rl == 0 && ok) {
#pragma unroll
        for (int o = 0; o < kSumWaves - 1; ++o) {
            const float4 c = s_red[o][cl];
            acc.x += c.x; acc.y += c.y; acc.z += c.z; acc.w += c.w;
        }
        st4(out + (int64_t)blockIdx.y * N + n, acc);
    }
}

// K14 -- cross-entropy of the codec head's logits (reference modeling_lina.py:106, F.cross_entropy(..., ignore_index=1)): one
// wave per row; the row (V = 4099 logits: 8 KB of bf16) is loaded ONCE into registers as NG aligned 4-element groups per lane
// (rows of an odd width start on any 2-byte boundary: the groups are laid on the 8-byte grid below the row start and the
// elements outside the row masked), then max, sum of exponentials and the target's logit come from registers.  The backward
// rebuilds softmax from the row and the saved log-sum-exp: d logits = (softmax - onehot) * scale for rows that count, 0 for
// ignored rows.  fp32 arithmetic from the stored dtype, as autocast's fp32 log_softmax -- without its fp32 copy of the logits
// (537 MB at config 5), the two casts and the separate log_softmax / nll kernels and their backwards.
__device__ __forceinline__ float4 pack4(const float (&e)[4]) { return make_float4(e[0], e[1], e[2], e[3]); }
__device__ __forceinline__ uint2 pack4(const bf16_t (&e)[4]) {
    return make_uint2((uint32_t)e[0] | ((uint32_t)e[1] << 16), (uint32_t)e[2] | ((uint32_t)e[3] << 16));
}

template <typename T, int NG> struct CeRow {
    typename raw4<T>::type raw[NG];
    int64_t a;          // element index of group 0 (multiple of 4, <= row start)
    int lo, hi;         // the row's elements are the grid positions [lo, hi) relative to a
    __device__ __forceinline__ void load(const T* __restrict__ logits, int64_t row, int64_t ld, int V, int lane, int64_t total) {
        const int64_t b = row * ld;
        a = b & ~(int64_t)3;
        lo = (int)(b - a);
        hi = lo + V;
        // groups past the row are clamped back to the row's last readable group (their elements are masked in values()); the
        // one group that may straddle the END OF THE TENSOR (last row only) is assembled from single-element loads
        const int64_t last_grp = (b + V - 1) & ~(int64_t)3;           // the last group that holds elements of the row
        const bool tail = last_grp + 4 > total;                       // wave-uniform: it reaches past the tensor
        const int64_t last_ok = tail ? last_grp - 4 : last_grp;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int64_t e = a + 4 * ((int64_t)g * 64 + lane);
            raw[g] = ld4_raw(logits + (e <= last_ok ? e : last_ok));
        }
        if (tail) {                                                   // scalar branch: the last row of the tensor only
            const int64_t e0 = last_grp;                              // the straddling group
            const int64_t q = (e0 - a) / 4;                           // its grid index: lane q % 64 of group q / 64
            T el[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) el[c] = logits[e0 + c < total ? e0 + c : total - 1];
#pragma unroll
            for (int g = 0; g < NG; ++g)
                if (g == (int)(q / 64) && lane == (int)(q % 64)) raw[g] = pack4(el);
        }
    }
    __device__ __forceinline__ void values(int g, int lane, float (&v)[4], bool (&in)[4], int64_t total) const {
        const float4 f = cvt4(raw[g]);
        const float fv[4] = {f.x, f.y, f.z, f.w};
        const int p = 4 * (g * 64 + lane);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            in[c] = p + c >= lo && p + c < hi;                        // (clamped groups lie wholly past hi)
            v[c] = fv[c];
        }
    }
};

__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, shfl_xor(v, 1)); v = fmaxf(v, shfl_xor(v, 2)); v = fmaxf(v, shfl_xor(v, 4));
    v = fmaxf(v, shfl_xor(v, 8)); v = fmaxf(v, shfl_xor(v, 16)); return fmaxf(v, shfl_xor(v, 32));
}
template <typename T, int NG, bool BWD>
__global__ __launch_bounds__(256) void cross_entropy_kernel(const T* __restrict__ logits, const int64_t* __restrict__ target,
                                                            float* __restrict__ lse, float* __restrict__ loss_row,
                                                            const float* __restrict__ scale, T* __restrict__ dlogits, int64_t N,
                                                            int V, int64_t ld, int64_t ld_d, int64_t ignore_index) {
    const int lane = threadIdx.x & 63, wv = wave_uniform(threadIdx.x >> 6);
    const int64_t total = (N - 1) * ld + V;                           // elements of the logits tensor that may be touched
    for (int64_t row = (int64_t)blockIdx.x * 4 + wv; row < N; row += (int64_t)gridDim.x * 4) {
        CeRow<T, NG> r;
        r.load(logits, row, ld, V, lane, total);
        const int64_t tg = target[row];
        const bool counts = tg != ignore_index && tg >= 0 && tg < V;
        if (!BWD) {
            float m = -INFINITY;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                float v[4]; bool in[4];
                r.values(g, lane, v, in, total);
#pragma unroll
                for (int c = 0; c < 4; ++c) m = fmaxf(m, in[c] ? v[c] : -INFINITY);
            }
            m = wave_max(m);
            float se = 0.0f, xt = 0.0f;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                float v[4]; bool in[4];
                r.values(g, lane, v, in, total);
                const int p = 4 * (g * 64 + lane) - r.lo;             // column of element 0 of this group
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    se += in[c] ? __expf(v[c] - m) : 0.0f;
                    xt += (in[c] && p + c == tg) ? v[c] : 0.0f;
                }
            }
            se = wave_sum(se);
            xt = wave_sum(xt);
            const float l = m + __logf(se);
            if (lane == 0) {
                lse[row] = l;
                loss_row[row] = counts ? l - xt : 0.0f;
            }
        } else {
            const float l = lse[row];
            const float sc = counts ? scale[0] : 0.0f;
            T* drow = dlogits + row * ld_d;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                float v[4]; bool in[4];
                r.values(g, lane, v, in, total);
                const int p = 4 * (g * 64 + lane) - r.lo;
                float o[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] = (__expf(v[c] - l) - (p + c == tg ? 1.0f : 0.0f)) * sc;
                if (in[0] && in[3] && ((row * ld_d + p) & 3) == 0) { // whole group inside the row, 8-byte aligned in dlogits
                    st4(drow + p, make_float4(o[0], o[1], o[2], o[3]));
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (in[c]) st(drow + p + c, o[c]);
                }
            }
        }
    }
}

// K12 -- the gate of the mixer for a whole sequence (reference model/gla.py:174-180): y = logsigmoid(x) / normalizer
// (optionally clamped from below), and its gradient dx = dy (1 - sigmoid(x)) / normalizer (0 where the clamp is active).
// Elementwise over n4 groups of 4 elements; torch's chain (log_sigmoid with its second output, the division, their two
// backward kernels) moved 4.5x the bytes.
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void gate_logsigmoid_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ out,
                                                              int64_t n4, float inv_norm, float clamp_min, int has_clamp) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 a = ld4(x + 4 * i);
        const float av[4] = {a.x, a.y, a.z, a.w};
        float o[4];
        if (BWD) {
            const float4 d = ld4(dy + 4 * i);
            const float dv[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool clamped = has_clamp && logsigmoidf(av[c]) * inv_norm < clamp_min;
                o[c] = clamped ? 0.0f : dv[c] * inv_norm * sigmoidf(-av[c]);
            }
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                o[c] = logsigmoidf(av[c]) * inv_norm;
                if (has_clamp) o[c] = fmaxf(o[c], clamp_min);
            }
        }
        st4(out + 4 * i, make_float4(o[0], o[1], o[2], o[3]));
    }
}

// K12b -- the gate projection and the gate in one pass (reference model/gla.py:107-109,174-180): the second factor of the
// low-rank gate projection has an inner dimension of 16, so pre = lr W^T + b is 16 FMAs per element from a 32-byte row of lr
// (wave-uniform: scalar loads) and the 4 x 16 weights a thread keeps in registers -- the [R, C] pre-activation never exists
// in memory, forward or backward.  A workgroup takes kGateRows rows x 256 columns: each of its 4 waves a quarter of the rows
// (8 at a time, all loads first), each lane 4 columns; backward a lane also carries dW [4][16] and db [4] of its columns,
// the 4 waves add theirs through LDS and the workgroup writes ONE fp32 partial [C-block][L + 1] (slot L = bias) per row slab,
// which the caller sums -- torch's chain for the same work was a 10-TFLOP/s [C, R] x [R, 16] GEMM, two column sums and the
// bias add on top of the elementwise kernels.  (First form: 4 waves side by side over 1024 columns, 256 workgroups = one
// wave per SIMD: 120 / 134 us, latency-bound; this form has 4 per SIMD.)
// Rounding follows the autocast chain it replaces: weights and bias rounded to T, fp32 accumulate, pre rounded to T.
constexpr int kGateRows = LINA_GATE_LOWRANK_ROWS;
constexpr int kGateL = 16;
constexpr int kGateUnroll = 8;

template <typename T> __device__ __forceinline__ float round_to(float v);
template <> __device__ __forceinline__ float round_to<float>(float v) { return v; }
template <> __device__ __forceinline__ float round_to<bf16_t>(float v) { return bf2f(f2bf(v)); }

// one row of lr (wave-uniform address): 32-bit words, so that bf16 rows go through the scalar cache as well
__device__ __forceinline__ void load_lr_row(const float* p, int L, bool full, float (&lv)[kGateL]) {
#pragma unroll
    for (int j = 0; j < kGateL; ++j) lv[j] = (full || j < L) ? p[j] : 0.0f;
}
__device__ __forceinline__ void load_lr_row(const bf16_t* p, int L, bool full, float (&lv)[kGateL]) {
    if (full) {
        const uint32_t* q = reinterpret_cast<const uint32_t*>(p);
#pragma unroll
        for (int j = 0; j < kGateL / 2; ++j) {
            const uint32_t u = q[j];
            lv[2 * j] = bf2f((bf16_t)(u & 0xffff));
            lv[2 * j + 1] = bf2f((bf16_t)(u >> 16));
        }
    } else {
#pragma unroll
        for (int j = 0; j < kGateL; ++j) lv[j] = j < L ? bf2f(p[j]) : 0.0f;
    }
}

// FULL: L == kGateL and rows of lr 4-byte aligned (the reference's gate_low_rank_dim = 16): no per-element bounds
template <typename T, bool BWD, bool FULL>
__global__ __launch_bounds__(256) void gate_lowrank_kernel(const T* __restrict__ lr, int64_t lr_stride, const float* __restrict__ w,
                                                           const float* __restrict__ b, const T* __restrict__ dy, T* __restrict__ out,
                                                           float* __restrict__ dwb_partial, int64_t R, int C, int L,
                                                           float inv_norm, float clamp_min, int has_clamp) {
    constexpr int kAcc = 4 * (kGateL + 1), kPad = kAcc + 1;           // per-lane accumulators; odd LDS stride
    __shared__ float s_red[BWD ? 4 * 64 * kPad : 1];
    const int lane = threadIdx.x & 63, wv = wave_uniform(threadIdx.x >> 6);
    const int c0 = (blockIdx.y * 64 + lane) * 4;
    const bool col_ok = c0 < C;
    const int cc = col_ok ? c0 : C - 4;
    float wr[4][kGateL], br[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        br[c] = b ? round_to<T>(b[cc + c]) : 0.0f;
#pragma unroll
        for (int j = 0; j < kGateL; ++j) wr[c][j] = (FULL || j < L) ? round_to<T>(w[(int64_t)(cc + c) * L + j]) : 0.0f;
    }
    float dwa[BWD ? 4 : 1][BWD ? kGateL : 1], dba[4] = {0.f, 0.f, 0.f, 0.f};
    if (BWD) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int j = 0; j < kGateL; ++j) dwa[c][j] = 0.0f;
    }
    const int64_t r_begin = (int64_t)blockIdx.x * kGateRows + wv * (kGateRows / 4);
    const int64_t r_end = r_begin + kGateRows / 4 < R ? r_begin + kGateRows / 4 : R;
    for (int64_t r0 = r_begin; r0 < r_end; r0 += kGateUnroll) {
        typename raw4<T>::type dr[kGateUnroll];
        if (BWD) {
#pragma unroll
            for (int u = 0; u < kGateUnroll; ++u) {
                const int64_t r = r0 + u < R ? r0 + u : R - 1;
                dr[u] = ld4_raw(dy + r * C + cc);
            }
        }
#pragma unroll
        for (int u = 0; u < kGateUnroll; ++u) {
            const int64_t r = r0 + u;
            if (r >= r_end) break;                                   // wave-uniform
            float lv[kGateL];
            load_lr_row(lr + r * lr_stride, L, FULL, lv);
            float pre[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float a = 0.0f;
#pragma unroll
                for (int j = 0; j < kGateL; ++j) a = fmaf(lv[j], wr[c][j], a);
                pre[c] = round_to<T>(a + br[c]);
            }
            float o[4];
            if (BWD) {
                const float4 d4 = cvt4(dr[u]);
                const float dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const bool clamped = has_clamp && logsigmoidf(pre[c]) * inv_norm < clamp_min;
                    o[c] = round_to<T>(clamped ? 0.0f : dv[c] * inv_norm * sigmoidf(-pre[c]));
                    dba[c] += o[c];
#pragma unroll
                    for (int j = 0; j < kGateL; ++j) dwa[c][j] = fmaf(o[c], lv[j], dwa[c][j]);
                }
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    o[c] = logsigmoidf(pre[c]) * inv_norm;
                    if (has_clamp) o[c] = fmaxf(o[c], clamp_min);
                }
            }
            if (col_ok) st4(out + r * C + c0, make_float4(o[0], o[1], o[2], o[3]));
        }
    }
    if (BWD) {
        // the 4 waves' sums -> LDS [wave][lane][kAcc], added and written as one contiguous run of 64 x kAcc floats
        float* mine = s_red + (wv * 64 + lane) * kPad;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int j = 0; j < kGateL; ++j) mine[c * (kGateL + 1) + j] = dwa[c][j];
            mine[c * (kGateL + 1) + kGateL] = dba[c];
        }
        __syncthreads();
        const int ncol = C - blockIdx.y * 256 < 256 ? C - blockIdx.y * 256 : 256;          // columns of this block
        float* dst = dwb_partial + ((int64_t)blockIdx.x * C + blockIdx.y * 256) * (L + 1);
        for (int i = threadIdx.x; i < ncol * (L + 1); i += 256) {
            const int col = i / (L + 1), k = i % (L + 1);               // column within the block, slot (L = bias)
            const int src = (col >> 2) * kPad + (col & 3) * (kGateL + 1) + (k == L ? kGateL : k);
            dst[i] = s_red[src] + s_red[64 * kPad + src] + s_red[128 * kPad + src] + s_red[192 * kPad + src];
        }
    }
}

// K12c -- the bf16 form of K12b with both rank-16 contractions on the matrix core (L = 16, C a multiple of 64, 16-byte
// aligned rows).  K12b spends 16 FMAs per element on lr W^T and, backward, 16 more on dW += dpre^T lr: ~50 VALU instructions
// per element against 4 bytes of traffic -- the forward wrote its 67 MB at 1.25 TB/s, the backward moved 134 MB at 1.65 TB/s.
// Here a wave owns 64 columns and walks the rows 16 at a time:
//   pre^T [cols x rows] = W [cols x 16] . lr^T [16 x rows]   one v_mfma_f32_16x16x32_bf16 per 16 columns (k >= 16: zeros);
//     the column each A row stands for is permuted so that lane (row = l & 15, g = l >> 4) ends up with 8 CONSECUTIVE columns
//     per pair of tiles: dy comes in and the result goes out as one 16-byte access per lane and tile pair;
//   backward, per 32 rows: dpre (rounded to bf16, as K12b rounds it) goes to global memory AND row-major into the wave's
//     own LDS tile; dW^T [16 x cols] += lr^T [16 x rows] . dpre [rows x cols] takes both operands from LDS with the
//     transposing read ds_read_b64_tr_b16 (a lane needs 8 rows of one column); dbias is summed per lane and folded over the
//     16 row lanes once at the end.  No workgroup barrier: the four waves share rows, not data.
// Same values as K12b up to the summation order inside a 16-term dot product (fp32) / a 128-row partial sum.
constexpr int kGate2Pad = 8;                                          // LDS row = 64 + 8 bf16: the four row groups of a
                                                                      // transposing read land on different banks
template <bool BWD>
__global__ __launch_bounds__(256) void gate_lowrank_mfma_kernel(const bf16_t* __restrict__ lr, int64_t lr_stride,
                                                                const float* __restrict__ w, const float* __restrict__ b,
                                                                const bf16_t* __restrict__ dy, bf16_t* __restrict__ out,
                                                                float* __restrict__ dwb_partial, int64_t R, int C,
                                                                float inv_norm, float clamp_min, int has_clamp) {
    constexpr int L = kGateL, kRow = 64 + kGate2Pad;
    __shared__ __attribute__((aligned(16))) bf16_t s_dp[BWD ? 4 : 1][BWD ? 32 * kRow : 8];
    __shared__ __attribute__((aligned(16))) bf16_t s_lr[BWD ? 4 : 1][BWD ? 32 * L : 8];
    const int lane = threadIdx.x & 63, wv = wave_uniform(threadIdx.x >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int cw = blockIdx.y * 256 + wv * 64;                        // first column of this wave
    if (cw >= C) return;
    // A operands: tile t, MFMA row m = li stands for column cw + 32 (t / 2) + 8 (m / 4) + 4 (t % 2) + m % 4
    bf16x8 wa[4];
    float br[2][8];                                                   // bias of this lane's 8 columns per tile pair
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int col = cw + 32 * (t >> 1) + 8 * (li >> 2) + 4 * (t & 1) + (li & 3);
        uint32_t u[4] = {0u, 0u, 0u, 0u};
        if (lg < 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                u[j] = pack_bf16x2(w[(int64_t)col * L + 8 * lg + 2 * j], w[(int64_t)col * L + 8 * lg + 2 * j + 1]);
        }
        wa[t] = as_bf16x8(make_uint4(u[0], u[1], u[2], u[3]));
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) br[s][e] = b ? bf2f(f2bf(b[cw + 32 * s + 8 * lg + e])) : 0.0f;
    f32x4 dwt[4];                                                     // dW^T tiles: lane (col = 16 t + li, j = 4 lg + i)
    float dba[2][8];
    if (BWD) {
#pragma unroll
        for (int t = 0; t < 4; ++t) dwt[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) dba[s][e] = 0.0f;
    }
    const int64_t r_begin = (int64_t)blockIdx.x * kGateRows;
    const int64_t r_end = r_begin + kGateRows < R ? r_begin + kGateRows : R;
    for (int64_t r0 = r_begin; r0 < r_end; r0 += 32) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {                                 // two groups of 16 rows
            const int64_t row = r0 + 16 * h + li;
            const bool row_ok = row < r_end;
            const int64_t rc = row_ok ? row : r_end - 1;
            uint4 lb = make_uint4(0u, 0u, 0u, 0u);
            if (lg < 2) lb = *reinterpret_cast<const uint4*>(lr + rc * lr_stride + 8 * lg);
            uint4 dr[2];
            if (BWD) {
#pragma unroll
                for (int s = 0; s < 2; ++s) dr[s] = *reinterpret_cast<const uint4*>(dy + rc * C + cw + 32 * s + 8 * lg);
                if (lg < 2) *reinterpret_cast<uint4*>(&s_lr[wv][(16 * h + li) * L + 8 * lg]) = lb;
            }
            const bf16x8 bop = as_bf16x8(lb);
            f32x4 acc[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mfma_bf16_16x16x32(wa[t], bop, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float o[8];
                const uint32_t du[4] = {dr[s].x, dr[s].y, dr[s].z, dr[s].w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float pre = bf2f(f2bf(acc[2 * s + (e >> 2)][e & 3] + br[s][e]));
                    if (BWD) {
                        const float dv = bf2f((bf16_t)((e & 1) ? du[e >> 1] >> 16 : du[e >> 1] & 0xffff));
                        const bool clamped = has_clamp && logsigmoidf(pre) * inv_norm < clamp_min;
                        o[e] = (clamped || !row_ok) ? 0.0f : bf2f(f2bf(dv * inv_norm * sigmoidf(-pre)));
                        dba[s][e] += o[e];
                    } else {
                        o[e] = logsigmoidf(pre) * inv_norm;
                        if (has_clamp) o[e] = fmaxf(o[e], clamp_min);
                    }
                }
                const uint4 pk = make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]),
                                            pack_bf16x2(o[6], o[7]));
                if (row_ok) *reinterpret_cast<uint4*>(out + row * C + cw + 32 * s + 8 * lg) = pk;
                if (BWD) *reinterpret_cast<uint4*>(&s_dp[wv][(16 * h + li) * kRow + 32 * s + 8 * lg]) = pk;
            }
        }
        if (BWD) {
            // lane group lg takes rows 8 lg .. 8 lg + 7 of the 32 as its eight k values; inside the group lane p passes the
            // piece (row p / 4 of four, elements 4 (p % 4) ..): lane i receives column i of those four rows
            const int pr = 8 * lg + (li >> 2), pc = 4 * (li & 3);
            const bf16x8 la = as_bf16x8(lds_read_tr16_b64(&s_lr[wv][pr * L + pc]), lds_read_tr16_b64(&s_lr[wv][(pr + 4) * L + pc]));
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bf16x8 db = as_bf16x8(lds_read_tr16_b64(&s_dp[wv][pr * kRow + 16 * t + pc]),
                                            lds_read_tr16_b64(&s_dp[wv][(pr + 4) * kRow + 16 * t + pc]));
                dwt[t] = mfma_bf16_16x16x32(la, db, dwt[t]);
            }
        }
    }
    if (BWD) {
        float* dst = dwb_partial + ((int64_t)blockIdx.x * C + cw) * (L + 1);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) dst[(16 * t + li) * (L + 1) + 4 * lg + i] = dwt[t][i];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = dba[s][e];
                v += shfl_xor(v, 1); v += shfl_xor(v, 2); v += shfl_xor(v, 4); v += shfl_xor(v, 8);
                if (li == 0) dst[(32 * s + 8 * lg + e) * (L + 1) + L] = v;
            }
    }
}

}  // namespace lina

extern "C" int lina_swiglu_bwd_partials(int64_t rows);

extern "C" int lina_colsum(const void* x, float* partial, int64_t M, int N, int64_t ld, int dtype, lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(x && partial, "lina_colsum: null pointer");
    LINA_REQUIRE(M > 0 && M <= (int64_t)65535 * LINA_SWIGLU_COLSUM_ROWS && N >= 4 && N % 4 == 0 && ld >= N && ld % 4 == 0,
                 "lina_colsum: bad shape M=%lld N=%d ld=%lld (N, ld multiples of 4)", (long long)M, N, (long long)ld);
    LINA_REQUIRE(valid_dtype(dtype), "lina_colsum: bad dtype %d", dtype);
    dim3 grid((unsigned)lina_swiglu_bwd_partials(M), (unsigned)((N + 255) / 256));
    if (dtype == LINA_F32)
        LINA_LAUNCH((colsum_kernel<float>), grid, dim3(256), 0, stream, (const float*)x, partial, M, N, ld);
    else
        LINA_LAUNCH((colsum_kernel<bf16_t>), grid, dim3(256), 0, stream, (const bf16_t*)x, partial, M, N, ld);
    return check_launch("lina_colsum");
}

extern "C" int lina_sum_partials(const float* part, void* out, int outer, int P, int64_t N, int out_dtype, lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(part && out, "lina_sum_partials: null pointer");
    LINA_REQUIRE(outer >= 1 && outer <= 65535 && P >= 1 && N >= 4 && N % 4 == 0,
                 "lina_sum_partials: bad shape outer=%d P=%d N=%lld (N must be a multiple of 4)", outer, P, (long long)N);
    LINA_REQUIRE(valid_dtype(out_dtype), "lina_sum_partials: bad dtype %d", out_dtype);
    // lanes along the columns: the widest workgroup that still gives the grid ~64 of them (>= 128-byte row pieces)
    const int64_t cols4 = N / 4;
    int cl = 64;
    while (cl > 8 && ((cols4 + cl - 1) / cl) * outer < 64) cl >>= 1;
    const int per_lane = (P + kSumWaves * (64 / cl) - 1) / (kSumWaves * (64 / cl));   // partial rows per (wave, lane group)
    const int qr = per_lane <= 4 ? 4 : per_lane <= 8 ? 8 : 16;
    dim3 grid((unsigned)((cols4 + cl - 1) / cl), (unsigned)outer);
#define LINA_SUMP(TT, CLL, QRR)                                                                                      \
    LINA_LAUNCH((sum_partials_kernel<TT, CLL, QRR>), grid, dim3(64 * kSumWaves), 0, stream, part, (TT*)out, P, N)
#define LINA_SUMP_Q(TT, CLL) do { if (qr == 4) LINA_SUMP(TT, CLL, 4); else if (qr == 8) LINA_SUMP(TT, CLL, 8); else LINA_SUMP(TT, CLL, 16); } while (0)
#define LINA_SUMP_C(TT)                                                                                              \
    do {                                                                                                             \
        if (cl == 64) LINA_SUMP_Q(TT, 64); else if (cl == 32) LINA_SUMP_Q(TT, 32);                                   \
        else if (cl == 16) LINA_SUMP_Q(TT, 16); else LINA_SUMP_Q(TT, 8);                                             \
    } while (0)
    if (out_dtype == LINA_F32) LINA_SUMP_C(float); else LINA_SUMP_C(bf16_t);
#undef LINA_SUMP_C
#undef LINA_SUMP_Q
#undef LINA_SUMP
    return check_launch("lina_sum_partials");
}

extern "C" int lina_cross_entropy(const void* logits, const int64_t* target, float* lse, float* loss_row, const float* scale,
                                  void* dlogits, int64_t N, int V, int64_t ld, int64_t ld_d, int64_t ignore_index, int dtype,
                                  lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(logits && target && lse, "lina_cross_entropy: null pointer");
    LINA_REQUIRE((loss_row != nullptr) != (dlogits != nullptr), "lina_cross_entropy: give loss_row (forward) or dlogits (backward)");
    LINA_REQUIRE(!dlogits || scale, "lina_cross_entropy: the backward needs the scale");
    LINA_REQUIRE(N > 0 && V >= 4 && ld >= V && (!dlogits || ld_d >= V), "lina_cross_entropy: bad shape N=%lld V=%d ld=%lld ld_d=%lld",
                 (long long)N, V, (long long)ld, (long long)ld_d);
    LINA_REQUIRE(V + 3 <= 256 * 33, "lina_cross_entropy: V=%d above the register-resident row limit %d", V, 256 * 33 - 3);
    LINA_REQUIRE(valid_dtype(dtype), "lina_cross_entropy: bad dtype %d", dtype);
    LINA_REQUIRE((reinterpret_cast<uintptr_t>(logits) & 15) == 0 && (!dlogits || (reinterpret_cast<uintptr_t>(dlogits) & 15) == 0),
                 "lina_cross_entropy: logits / dlogits must be 16-byte aligned");
    const int64_t wgs = (N + 3) / 4;
    dim3 grid((unsigned)(wgs < 4096 ? wgs : 4096));
    const int ng = (V + 3 + 255) / 256;                               // 4-element groups per lane that cover lo + V
#define LINA_CE(TT, NGG, BB)                                                                                         \
    LINA_LAUNCH((cross_entropy_kernel<TT, NGG, BB>), grid, dim3(256), 0, stream, (const TT*)logits, target, lse, loss_row, scale, \
                (TT*)dlogits, N, V, ld, ld_d, ignore_index)
#define LINA_CE_N(TT, BB)                                                                                            \
    do {                                                                                                             \
        if (ng <= 2) LINA_CE(TT, 2, BB); else if (ng <= 5) LINA_CE(TT, 5, BB); else if (ng <= 9) LINA_CE(TT, 9, BB);   \
        else if (ng <= 17) LINA_CE(TT, 17, BB); else LINA_CE(TT, 33, BB);                                            \
    } while (0)
    if (dtype == LINA_F32) { if (dlogits) LINA_CE_N(float, true); else LINA_CE_N(float, false); }
    else { if (dlogits) LINA_CE_N(bf16_t, true); else LINA_CE_N(bf16_t, false); }
#undef LINA_CE_N
#undef LINA_CE
    return check_launch("lina_cross_entropy");
}

extern "C" int lina_swiglu_bwd_partials(int64_t rows) {
    return rows <= 0 ? 0 : (int)((rows + LINA_SWIGLU_COLSUM_ROWS - 1) / LINA_SWIGLU_COLSUM_ROWS);
}

extern "C" int lina_swiglu_bwd_colsum(const void* ds, const void* u, void* du, float* colsum_partial, int64_t rows, int Hd,
                                      int64_t ld_u, int64_t ld_ds, int64_t ld_du, int dtype, lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(ds && u && du && colsum_partial, "lina_swiglu_bwd_colsum: null pointer");
    LINA_REQUIRE(rows > 0 && rows <= (int64_t)65535 * LINA_SWIGLU_COLSUM_ROWS, "lina_swiglu_bwd_colsum: bad row count");
    LINA_REQUIRE(Hd >= 4 && Hd % 4 == 0 && ld_u % 4 == 0 && ld_ds % 4 == 0 && ld_du % 4 == 0,
                 "lina_swiglu_bwd_colsum: Hd and the row strides must be multiples of 4");
    LINA_REQUIRE(valid_dtype(dtype), "lina_swiglu_bwd_colsum: bad dtype %d", dtype);
    dim3 grid((unsigned)lina_swiglu_bwd_partials(rows), (unsigned)((Hd + 255) / 256));
    if (dtype == LINA_F32)
        LINA_LAUNCH((swiglu_bwd_colsum_kernel<float>), grid, dim3(256), 0, stream, (const float*)ds, (const float*)u, (float*)du,
                    colsum_partial, rows, Hd, ld_u, ld_ds, ld_du);
    else
        LINA_LAUNCH((swiglu_bwd_colsum_kernel<bf16_t>), grid, dim3(256), 0, stream, (const bf16_t*)ds, (const bf16_t*)u,
                    (bf16_t*)du, colsum_partial, rows, Hd, ld_u, ld_ds, ld_du);
    return check_launch("lina_swiglu_bwd_colsum");
}

extern "C" int lina_gate_lowrank_partials(int64_t rows) {
    return rows <= 0 ? 0 : (int)((rows + LINA_GATE_LOWRANK_ROWS - 1) / LINA_GATE_LOWRANK_ROWS);
}

extern "C" int lina_gate_lowrank(const void* lr, int64_t lr_stride, const float* w, const float* b, const void* dy, void* out,
                                 float* dwb_partial, int64_t rows, int C, int L, float normalizer, float clamp_min, int dtype,
                                 lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(lr && w && out, "lina_gate_lowrank: null pointer");
    LINA_REQUIRE(!dy == !dwb_partial, "lina_gate_lowrank: dy and dwb_partial must both be given (backward) or both be NULL");
    LINA_REQUIRE(rows > 0 && rows <= (int64_t)65535 * LINA_GATE_LOWRANK_ROWS, "lina_gate_lowrank: bad row count %lld", (long long)rows);
    LINA_REQUIRE(C >= 4 && C % 4 == 0, "lina_gate_lowrank: C=%d must be a positive multiple of 4", C);
    LINA_REQUIRE(L >= 1 && L <= kGateL, "lina_gate_lowrank: inner dimension L=%d must be in 1..%d", L, kGateL);
    LINA_REQUIRE(lr_stride >= L, "lina_gate_lowrank: lr_stride=%lld < L", (long long)lr_stride);
    LINA_REQUIRE(normalizer != 0.0f, "lina_gate_lowrank: normalizer must be non-zero");
    LINA_REQUIRE(valid_dtype(dtype), "lina_gate_lowrank: bad dtype %d", dtype);
    const int has_clamp = (clamp_min == clamp_min) ? 1 : 0;          // NaN = no clamp
    dim3 grid((unsigned)lina_gate_lowrank_partials(rows), (unsigned)((C + 255) / 256));
    const bool full = L == kGateL && lr_stride % 2 == 0 && (reinterpret_cast<uintptr_t>(lr) & 3) == 0;
    if (dtype == LINA_BF16 && L == kGateL && C % 64 == 0 && lr_stride % 8 == 0 && (reinterpret_cast<uintptr_t>(lr) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(out) & 15) == 0 && (!dy || (reinterpret_cast<uintptr_t>(dy) & 15) == 0)) {
        // K12c: both rank-16 contractions on the matrix core
        if (dy)
            LINA_LAUNCH((gate_lowrank_mfma_kernel<true>), grid, dim3(256), 0, stream, (const bf16_t*)lr, lr_stride, w, b,
                        (const bf16_t*)dy, (bf16_t*)out, dwb_partial, rows, C, 1.0f / normalizer, clamp_min, has_clamp);
        else
            LINA_LAUNCH((gate_lowrank_mfma_kernel<false>), grid, dim3(256), 0, stream, (const bf16_t*)lr, lr_stride, w, b,
                        (const bf16_t*)dy, (bf16_t*)out, dwb_partial, rows, C, 1.0f / normalizer, clamp_min, has_clamp);
        return check_launch("lina_gate_lowrank");
    }
#define LINA_GLR(TT, BB, FF)                                                                                         \
    LINA_LAUNCH((gate_lowrank_kernel<TT, BB, FF>), grid, dim3(256), 0, stream, (const TT*)lr, lr_stride, w, b, (const TT*)dy, \
                (TT*)out, dwb_partial, rows, C, L, 1.0f / normalizer, clamp_min, has_clamp)
#define LINA_GLR_F(TT, BB) do { if (full) LINA_GLR(TT, BB, true); else LINA_GLR(TT, BB, false); } while (0)
    if (dtype == LINA_F32) { if (dy) LINA_GLR_F(float, true); else LINA_GLR_F(float, false); }
    else { if (dy) LINA_GLR_F(bf16_t, true); else LINA_GLR_F(bf16_t, false); }
#undef LINA_GLR_F
#undef LINA_GLR
    return check_launch("lina_gate_lowrank");
}

extern "C" int lina_gate_logsigmoid(const void* x, const void* dy, void* out, int64_t n, float normalizer, float clamp_min,
                                    int dtype, lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(x && out, "lina_gate_logsigmoid: null pointer");
    LINA_REQUIRE(n > 0 && n % 4 == 0, "lina_gate_logsigmoid: n must be a positive multiple of 4");
    LINA_REQUIRE(normalizer != 0.0f, "lina_gate_logsigmoid: normalizer must be non-zero");
    LINA_REQUIRE(valid_dtype(dtype), "lina_gate_logsigmoid: bad dtype %d", dtype);
    const int has_clamp = (clamp_min == clamp_min) ? 1 : 0;          // NaN = no clamp
    const int64_t n4 = n / 4, wgs = (n4 + 255) / 256;
    dim3 grid((unsigned)(wgs < 65536 ? wgs : 65536));
#define LINA_GL(TT, BB)                                                                                              \
    LINA_LAUNCH((gate_logsigmoid_kernel<TT, BB>), grid, dim3(256), 0, stream, (const TT*)x, (const TT*)dy, (TT*)out, n4, \
                1.0f / normalizer, clamp_min, has_clamp)
    if (dtype == LINA_F32) { if (dy) LINA_GL(float, true); else LINA_GL(float, false); }
    else { if (dy) LINA_GL(bf16_t, true); else LINA_GL(bf16_t, false); }
#undef LINA_GL
    return check_launch("lina_gate_logsigmoid");
}

extern "C" int lina_layernorm_bwd_partials(int64_t rows) {
    const int64_t wgs = (rows + 3) / 4;
    return (int)(wgs < 1024 ? wgs : 1024);
}

namespace {
template <typename TX, typename TR, typename TY>
int ln_fwd_launch(int np, dim3 grid, lina_stream_t stream, const void* x, const void* r, const float* gamma, const float* beta,
                  void* xsum, void* y, float* mean, float* rstd, int64_t N, int D, float eps) {
    using namespace lina;
#define LINA_LN_F(NPP)                                                                                               \
    LINA_LAUNCH((layernorm_fwd_kernel<TX, TR, TY, NPP>), grid, dim3(256), 0, stream, (const TX*)x, (const TR*)r, gamma, beta, \
                (TX*)xsum, (TY*)y, mean, rstd, N, D, eps)
    if (np <= 1) LINA_LN_F(1); else if (np <= 2) LINA_LN_F(2); else if (np <= 4) LINA_LN_F(4); else LINA_LN_F(8);
#undef LINA_LN_F
    return check_launch("lina_layernorm_fwd");
}
template <typename TX, typename TR, typename TY>
int ln_bwd_launch(int np, dim3 grid, lina_stream_t stream, const void* dy, const void* x, const float* mean, const float* rstd,
                  const float* gamma, const void* dpass, void* dx, void* dr, float* dg, float* db, int64_t N, int D) {
    using namespace lina;
#define LINA_LN_B(NPP)                                                                                               \
    LINA_LAUNCH((layernorm_bwd_kernel<TX, TR, TY, NPP>), grid, dim3(256), 0, stream, (const TY*)dy, (const TX*)x, mean, rstd, \
                gamma, (const TX*)dpass, (TX*)dx, (TR*)dr, dg, db, N, D)
    if (np <= 1) LINA_LN_B(1); else if (np <= 2) LINA_LN_B(2); else if (np <= 4) LINA_LN_B(4); else LINA_LN_B(8);
#undef LINA_LN_B
    return check_launch("lina_layernorm_bwd");
}
}  // namespace

// dtype triple (x_dtype, r_dtype, y_dtype): the combinations the train step produces -- fp32 stream with bf16 or fp32
// branches and outputs, and the all-bf16 stream of a model cast to bf16
#define LINA_LN_DISPATCH(CALL)                                                                                        \
    do {                                                                                                              \
        if (x_dtype == LINA_F32 && r_dtype == LINA_F32 && y_dtype == LINA_F32) return CALL(float, float, float);      \
        if (x_dtype == LINA_F32 && r_dtype == LINA_BF16 && y_dtype == LINA_BF16) return CALL(float, bf16_t, bf16_t);  \
        if (x_dtype == LINA_F32 && r_dtype == LINA_F32 && y_dtype == LINA_BF16) return CALL(float, float, bf16_t);    \
        if (x_dtype == LINA_BF16 && r_dtype == LINA_BF16 && y_dtype == LINA_BF16) return CALL(bf16_t, bf16_t, bf16_t); \
        return fail(LINA_ERR_UNSUPPORTED, "lina_layernorm: dtype combination (%d, %d, %d) is not built", x_dtype, r_dtype, y_dtype); \
    } while (0)

extern "C" int lina_layernorm_fwd(const void* x, const void* r, const float* gamma, const float* beta, void* xsum, void* y,
                                  float* mean, float* rstd, int64_t N, int D, float eps, int x_dtype, int r_dtype,
                                  int y_dtype, lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(x && gamma && beta && y && mean && rstd, "lina_layernorm_fwd: null pointer");
    LINA_REQUIRE(N > 0 && D > 0 && D % 4 == 0, "lina_layernorm_fwd: N, D must be positive, D a multiple of 4");
    LINA_REQUIRE(!xsum || r, "lina_layernorm_fwd: xsum needs the added branch r");
    LINA_REQUIRE(valid_dtype(x_dtype) && valid_dtype(r_dtype) && valid_dtype(y_dtype), "lina_layernorm_fwd: bad dtype enum");
    if (D > 256 * kLnMaxPieces) return fail(LINA_ERR_UNSUPPORTED, "lina_layernorm_fwd: D=%d exceeds %d", D, 256 * kLnMaxPieces);
    const int np = (D + 255) / 256;
    const int64_t wgs = (N + 3) / 4;
    dim3 grid((unsigned)(wgs < 16384 ? wgs : 16384));
#define LINA_LN_CALL(A, B, C) ln_fwd_launch<A, B, C>(np, grid, stream, x, r, gamma, beta, xsum, y, mean, rstd, N, D, eps)
    LINA_LN_DISPATCH(LINA_LN_CALL);
#undef LINA_LN_CALL
}

extern "C" int lina_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                                  const void* dpass, void* dx, void* dr, float* dgamma_part, float* dbeta_part, int64_t N,
                                  int D, int x_dtype, int r_dtype, int y_dtype, lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(dy && x && mean && rstd && gamma && dx && dgamma_part && dbeta_part, "lina_layernorm_bwd: null pointer");
    LINA_REQUIRE(N > 0 && D > 0 && D % 4 == 0, "lina_layernorm_bwd: N, D must be positive, D a multiple of 4");
    LINA_REQUIRE(valid_dtype(x_dtype) && valid_dtype(r_dtype) && valid_dtype(y_dtype), "lina_layernorm_bwd: bad dtype enum");
    if (D > 256 * kLnMaxPieces) return fail(LINA_ERR_UNSUPPORTED, "lina_layernorm_bwd: D=%d exceeds %d", D, 256 * kLnMaxPieces);
    const int np = (D + 255) / 256;
    dim3 grid((unsigned)lina_layernorm_bwd_partials(N));
#define LINA_LN_CALL(A, B, C) ln_bwd_launch<A, B, C>(np, grid, stream, dy, x, mean, rstd, gamma, dpass, dx, dr, dgamma_part, dbeta_part, N, D)
    LINA_LN_DISPATCH(LINA_LN_CALL);
#undef LINA_LN_CALL
}

extern "C" int lina_swiglu_bwd(const void* ds, const void* u, void* du, int64_t rows, int Hd, int64_t ld_u, int64_t ld_ds,
                               int64_t ld_du, int dtype, lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(ds && u && du, "lina_swiglu_bwd: null pointer");
    LINA_REQUIRE(rows > 0 && Hd > 0, "lina_swiglu_bwd: rows, Hd must be positive");
    LINA_REQUIRE(valid_dtype(dtype), "lina_swiglu_bwd: bad dtype %d", dtype);
    if (Hd % 4 || ld_u % 4 || ld_ds % 4 || ld_du % 4) {           // e.g. L169: Hd = 1365
        const int64_t wgs1 = (rows * Hd + 255) / 256;
        dim3 grid1((unsigned)(wgs1 < 262144 ? wgs1 : 262144));
        if (dtype == LINA_F32)
            LINA_LAUNCH((swiglu_bwd_scalar_kernel<float>), grid1, dim3(256), 0, stream, (const float*)ds, (const float*)u,
                        (float*)du, rows, Hd, ld_u, ld_ds, ld_du);
        else
            LINA_LAUNCH((swiglu_bwd_scalar_kernel<bf16_t>), grid1, dim3(256), 0, stream, (const bf16_t*)ds, (const bf16_t*)u,
                        (bf16_t*)du, rows, Hd, ld_u, ld_ds, ld_du);
        return check_launch("lina_swiglu_bwd");
    }
    const int64_t n4 = rows * (Hd / 4);
    const int64_t wgs = (n4 + 255) / 256;
    dim3 grid((unsigned)(wgs < 65536 ? wgs : 65536));
    if (dtype == LINA_F32)
        LINA_LAUNCH((swiglu_bwd_kernel<float>), grid, dim3(256), 0, stream, (const float*)ds, (const float*)u, (float*)du, rows,
                    Hd, ld_u, ld_ds, ld_du);
    else
        LINA_LAUNCH((swiglu_bwd_kernel<bf16_t>), grid, dim3(256), 0, stream, (const bf16_t*)ds, (const bf16_t*)u, (bf16_t*)du,
                    rows, Hd, ld_u, ld_ds, ld_du);
    return check_launch("lina_swiglu_bwd");
}
