// gla_chunk_full.hip -- K2 (bf16, Dk = Dv = 256): chunk-wise GLA forward, ONE workgroup per (b,h),
// the whole 256 x 256 fp32 state resident in MFMA accumulators (8 waves x 8 tiles of 32 x 32 =
// 128 accumulator VGPRs per lane) for the entire sequence.  HBM traffic is exactly the algorithmic
// q,k,g,v in + o out (e*(3Dk+2Dv) per (row, head, token)); nothing else touches HBM.
//
// Replaces fla.ops.gla.chunk_gla / fused_chunk_gla (reference model/gla.py:193,195) for the L169
// head shape; other shapes / fp32 use gla_chunk.hip.  Same formulation as gla_chunk.hip:
//     q~ = scale q e^{b}, k~ = k e^{-b}, o = q~ S + mask(q~ k~^T) v,  S <- e^{b_last} (S + k~^T v)
// with chunks of C = 32 tokens, cut adaptively when the in-chunk decay would exceed e^-60.
//
// Per chunk (512 threads = 8 waves; thread (rg = tid>>5, co = tid&31) owns rows {2rg,2rg+1} x
// channels/columns 8co..8co+7 for the load/scan phase, i.e. 16-byte global loads, 512 B per row):
//   A  the chunk's raw q,k,g,v tiles are already in LDS (asynchronous global->LDS DMA issued one chunk ahead,
//      no staging registers: the state leaves only 128 VGPRs per lane); gate scan: per-thread 2-row sums ->
//      LDS -> exclusive scan over the 16 row groups; q~,k~ are computed in registers and written to LDS as
//      bf16 tiles q~[t][c], k~[t][c] (row-major) and k~^T[c][t], v^T[col][t]; then the NEXT chunk's DMA
//      (64 KiB, 8 x 1 KiB instructions per wave) is issued and flies under phase B.
//   B  wave w owns state columns [32w, 32w+32):  (1) o = q~ . S  -- the state tile in C/D layout is used
//      directly as the B operand by giving the A operand the matching k-slot -> channel map;
//      (2) A^T = k~ . q~^T (32 x 32, every wave, in registers; lane t ends up holding A[t][.] in exactly
//      the k-slot order the C/D layout dictates, so mask + bf16 convert makes it the next A operand);
//      (3) o += mask(A) . v;  (4) S += k~^T . v, then rows scaled by e^{b_last}.   50 MFMAs
//      (v_mfma_f32_32x32x16_bf16) per wave per chunk; o is staged through LDS and stored 16 B per lane.
#include <lina_dev.h>
#include "lina_common.h"

namespace lina {

constexpr int kFullC = 32;
constexpr float kFullMaxDecay = 60.0f;

__device__ __forceinline__ int drow(int reg, int hi) { return (reg & 3) + 8 * (reg >> 2) + 4 * hi; }

__device__ __forceinline__ void unpack8(const uint4 u, float (&f)[8]) {
    f[0] = bf2f((bf16_t)(u.x & 0xffff)); f[1] = bf2f((bf16_t)(u.x >> 16));
    f[2] = bf2f((bf16_t)(u.y & 0xffff)); f[3] = bf2f((bf16_t)(u.y >> 16));
    f[4] = bf2f((bf16_t)(u.z & 0xffff)); f[5] = bf2f((bf16_t)(u.z >> 16));
    f[6] = bf2f((bf16_t)(u.w & 0xffff)); f[7] = bf2f((bf16_t)(u.w >> 16));
}
__device__ __forceinline__ bf16x8 frag16(const bf16_t* p) {   // one 16-byte LDS read
    return as_bf16x8(*reinterpret_cast<const uint4*>(p));
}
__device__ __forceinline__ bf16x8 frag8x2(const bf16_t* p_lo, const bf16_t* p_hi) {   // two 8-byte LDS reads
    return as_bf16x8(*reinterpret_cast<const uint2*>(p_lo), *reinterpret_cast<const uint2*>(p_hi));
}

__global__ __launch_bounds__(512) void gla_chunk_bf16_h256_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    const bf16_t* __restrict__ gk, bf16_t* __restrict__ o, const float* h0, float* ht, int H, int T,
    lina_bht_strides sq, lina_bht_strides sk, lina_bht_strides sv, lina_bht_strides sg, lina_bht_strides so,
    float scale) {
    constexpr int DK = 256, DV = 256, C = kFullC;
    constexpr int SQ = DK + 8;   // bf16 row stride of the row-major tiles (528 B: 16-byte aligned rows)
    constexpr int ST = C + 8;    // bf16 row stride of the transposed tiles (80 B)
    __shared__ __attribute__((aligned(16))) bf16_t s_q[C * SQ];
    __shared__ __attribute__((aligned(16))) bf16_t s_k[C * SQ];
    __shared__ __attribute__((aligned(16))) bf16_t s_kT[DK * ST];
    __shared__ __attribute__((aligned(16))) bf16_t s_vT[DV * ST];
    __shared__ __attribute__((aligned(16))) bf16_t s_raw[4][C * DK];     // next chunk's q,k,g,v, filled by DMA
    __shared__ __attribute__((aligned(16))) float s_ot[C * SQ / 2];      // gate-scan totals, later the o tile
    __shared__ __attribute__((aligned(16))) float s_dec[DK];
    __shared__ int s_flag, s_nw[8];
    float* s_tot = s_ot;                                  // [16][DK] fp32   (phase A)
    bf16_t* s_o = reinterpret_cast<bf16_t*>(s_ot);        // [C][SQ] bf16    (phase B)

    int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int li = lane & 31, hi = lane >> 5;
    int co = tid & 31, rg = tid >> 5;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;

    // ---- state: wave w owns columns [32w, 32w+32), tile p = rows [32p, 32p+32) ----
    f32x16 S[8];
    {
        const float* hp = h0 ? h0 + ((int64_t)bh * DK + 4 * hi) * DV + 32 * w + li : nullptr;
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) S[p][r] = 0.0f;
        if (hp) {
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const float* tp = hp + (32 * p) * DV;
#pragma unroll
                for (int r = 0; r < 16; ++r) S[p][r] = tp[((r & 3) + 8 * (r >> 2)) * DV];
            }
        }
    }

    const bf16_t* gsrc[4] = {q + b * sq.b + h * sq.h, k + b * sk.b + h * sk.h, gk + b * sg.b + h * sg.h,
                             v + b * sv.b + h * sv.h};
    const int64_t gst[4] = {sq.t, sk.t, sg.t, sv.t};
    bf16_t* ob = o + b * so.b + h * so.h;

    // wave w DMAs rows 4w..4w+3 of each raw tile: one instruction = 2 rows x 512 B, 16 B per lane.
    // Rows past the end of the sequence re-read row T-1 (always mapped); phase A masks them.
    auto dma_chunk = [&](int t_first, int a_lo, int a_hi) {
#pragma unroll
        for (int a = a_lo; a < a_hi; ++a)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int row = 4 * w + 2 * u;
                const int t = min(t_first + row + hi, T - 1);
                dma16_to_lds(gsrc[a] + t * gst[a] + 8 * li, &s_raw[a][row * DK]);
            }
    };

    // rows >= nv are zeroed; the thread that owns row nv-1 publishes exp(b_last)
    auto write_tiles = [&](const float (&bc)[2][8], int nv) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int row = 2 * rg + rr;
            const bool valid = row < nv;
            float f[8];
            unsigned pk[8];
            unpack8(*reinterpret_cast<const uint4*>(&s_raw[0][row * DK + 8 * co]), f);
            float e[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                e[c] = __expf(bc[rr][c]);
                pk[c] = f2bf(valid ? f[c] * scale * e[c] : 0.0f);
            }
            *reinterpret_cast<uint4*>(&s_q[row * SQ + 8 * co]) =
                make_uint4(pk[0] | (pk[1] << 16), pk[2] | (pk[3] << 16), pk[4] | (pk[5] << 16), pk[6] | (pk[7] << 16));
            unpack8(*reinterpret_cast<const uint4*>(&s_raw[1][row * DK + 8 * co]), f);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                pk[c] = f2bf(valid ? __fdividef(f[c], e[c]) : 0.0f);
            }
            *reinterpret_cast<uint4*>(&s_k[row * SQ + 8 * co]) =
                make_uint4(pk[0] | (pk[1] << 16), pk[2] | (pk[3] << 16), pk[4] | (pk[5] << 16), pk[6] | (pk[7] << 16));
            if (row == nv - 1) {
#pragma unroll
                for (int c = 0; c < 8; ++c) s_dec[8 * co + c] = e[c];
            }
        }
    };

    // this thread's 2 rows x 8 channels of clamped gates, summed down the 2 rows (rows >= nrem count as 0)
    auto local_gates = [&](float (&bc)[2][8], int nrem) {
        float g0[8], g1[8];
        unpack8(*reinterpret_cast<const uint4*>(&s_raw[2][(2 * rg) * DK + 8 * co]), g0);
        unpack8(*reinterpret_cast<const uint4*>(&s_raw[2][(2 * rg + 1) * DK + 8 * co]), g1);
        const bool in0 = 2 * rg < nrem, in1 = 2 * rg + 1 < nrem;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            bc[0][c] = in0 ? fmaxf(g0[c], -kFullMaxDecay) : 0.0f;
            bc[1][c] = bc[0][c] + (in1 ? fmaxf(g1[c], -kFullMaxDecay) : 0.0f);
        }
    };
    // add the exclusive prefix over the 16 row groups (from s_tot); true if the chunk's total decay is too large
    auto add_prefix = [&](float (&bc)[2][8]) {
        float pre[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) pre[c] = 0.f;
#pragma unroll 1
        for (int r = 0; r < rg; ++r) {     // rg takes two values per wave: at most one partially masked trip
            const float4 x0 = *reinterpret_cast<const float4*>(&s_tot[r * DK + 8 * co]);
            const float4 x1 = *reinterpret_cast<const float4*>(&s_tot[r * DK + 8 * co + 4]);
            pre[0] += x0.x; pre[1] += x0.y; pre[2] += x0.z; pre[3] += x0.w;
            pre[4] += x1.x; pre[5] += x1.y; pre[6] += x1.z; pre[7] += x1.w;
        }
        bool viol = false;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            bc[0][c] += pre[c];
            bc[1][c] += pre[c];
            viol |= (-bc[1][c] > kFullMaxDecay);   // b is monotone: the last row group sees the chunk total
        }
        return viol;
    };

    dma_chunk(0, 0, 4);
    __syncthreads();   // DMA of chunk 0 landed (hipcc drains vmcnt before the barrier)
    int t0 = 0;
    while (t0 < T) {
        // keep the per-lane index arithmetic INSIDE the loop: hoisted, it would need ~100 more VGPRs than the
        // 128 the state leaves free and be spilled to scratch
        opaque(tid); opaque(lane); opaque(li); opaque(hi); opaque(co); opaque(rg);
        const int nrem = T - t0;
        // ---------------- phase A: gate scan ----------------
        float bc[2][8];
        local_gates(bc, nrem);
        *reinterpret_cast<float4*>(&s_tot[rg * DK + 8 * co]) = make_float4(bc[1][0], bc[1][1], bc[1][2], bc[1][3]);
        *reinterpret_cast<float4*>(&s_tot[rg * DK + 8 * co + 4]) = make_float4(bc[1][4], bc[1][5], bc[1][6], bc[1][7]);
        if (tid == 0) s_flag = 0;
        __syncthreads();   // (1)
        if (add_prefix(bc)) s_flag = 1;
        int n = min(C, nrem);
        write_tiles(bc, n);
        __syncthreads();   // (2) operand tiles ready; raw q,k,g consumed
        if (s_flag) {
            // ---- rare: the decay inside this chunk exceeds e^-60 -> cut the chunk at the first such row ----
            float bc[2][8];                 // recomputed (s_tot is intact until the o tile is staged)
            local_gates(bc, nrem);
            add_prefix(bc);
            int nc = C;
#pragma unroll
            for (int rr = 1; rr >= 0; --rr) {
                bool bad = false;
#pragma unroll
                for (int c = 0; c < 8; ++c) bad |= (-bc[rr][c] > kFullMaxDecay);
                if (bad) nc = 2 * rg + rr;
            }
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) nc = min(nc, shfl_xor_i(nc, m));
            if (lane == 0) s_nw[w] = nc;
            __syncthreads();
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) n = min(n, s_nw[ww]);
            n = max(n, 1);
            __syncthreads();   // everyone has read s_nw; the optimistic tiles are dead
            write_tiles(bc, n);
            __syncthreads();
        }
        if (t0 + n < T) dma_chunk(t0 + n, 0, 3);   // next chunk's raw q,k,g fly under phase B (v: see below)

        // ---------------- transposed operands: k~^T[c][t], v^T[col][t] ----------------
        // thread (ch = tid & 255, half = tid >> 8) gathers 16 tokens of one channel/column (2-byte LDS reads,
        // lanes along ch: conflict-free) and writes them as two 16-byte pieces (row stride 80 B: conflict-free)
        {
            const int ch = tid & 255, r0 = 16 * (tid >> 8);
            unsigned w8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                w8[j] = (unsigned)s_k[(r0 + 2 * j) * SQ + ch] | ((unsigned)s_k[(r0 + 2 * j + 1) * SQ + ch] << 16);
            *reinterpret_cast<uint4*>(&s_kT[ch * ST + r0]) = make_uint4(w8[0], w8[1], w8[2], w8[3]);
            *reinterpret_cast<uint4*>(&s_kT[ch * ST + r0 + 8]) = make_uint4(w8[4], w8[5], w8[6], w8[7]);
            cfence();
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ra = r0 + 2 * j, rb = ra + 1;
                const unsigned va = ra < n ? (unsigned)s_raw[3][ra * DK + ch] : 0u;
                const unsigned vb_ = rb < n ? (unsigned)s_raw[3][rb * DK + ch] : 0u;
                w8[j] = va | (vb_ << 16);
            }
            *reinterpret_cast<uint4*>(&s_vT[ch * ST + r0]) = make_uint4(w8[0], w8[1], w8[2], w8[3]);
            *reinterpret_cast<uint4*>(&s_vT[ch * ST + r0 + 8]) = make_uint4(w8[4], w8[5], w8[6], w8[7]);
            cfence();
        }

        // ---------------- phase B ----------------
        f32x16 acc;
        bf16x8 afr[2];   // mask(A)[t = li][k-slots] as the A operand of (3)
        {
            f32x16 at;
#pragma unroll
            for (int r = 0; r < 16; ++r) { at[r] = 0.0f; acc[r] = 0.0f; }
            // (2) A^T[s][t] = k~_s . q~_t  (lane t = li holds A[t][drow(reg,hi)])
#pragma unroll
            for (int ks = 0; ks < DK / 16; ++ks) {
                const int cc = 16 * ks + 8 * hi;
                at = mfma_bf16_32x32x16(frag16(&s_k[li * SQ + cc]), frag16(&s_q[li * SQ + cc]), at);
                if ((ks & 1) == 1) cfence();
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int stok = drow(8 * s2 + j, hi);
                    afr[s2][j] = (short)f2bf((stok <= li) ? at[8 * s2 + j] : 0.0f);
                }
        }
        // (1) o = q~ . S_old   (B operand = this wave's state tiles, converted to bf16 in registers)
#pragma unroll
        for (int p = 0; p < 8; ++p) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16_t* qp = &s_q[li * SQ + 32 * p + 16 * s + 4 * hi];
                const bf16x8 a = frag8x2(qp, qp + 8);
                bf16x8 bb;
#pragma unroll
                for (int j = 0; j < 8; ++j) bb[j] = (short)f2bf(S[p][8 * s + j]);
                acc = mfma_bf16_32x32x16(a, bb, acc);
            }
            cfence();
        }
        __syncthreads();   // (2b) k~^T / v^T complete; raw v consumed
        if (t0 + n < T) dma_chunk(t0 + n, 3, 4);
        // (3) o += mask(A) . v ; v fragments in the same token order as the C/D rows
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16_t* vp = &s_vT[(32 * w + li) * ST + 16 * s2 + 4 * hi];
            acc = mfma_bf16_32x32x16(afr[s2], frag8x2(vp, vp + 8), acc);
        }
        // stage o (this wave's 32 x 32 block)
#pragma unroll
        for (int r = 0; r < 16; ++r) s_o[drow(r, hi) * SQ + 32 * w + li] = f2bf(acc[r]);
        // (4) S <- e^{b_last} (S + k~^T v)
        {
            bf16x8 vb2[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) vb2[s2] = frag16(&s_vT[(32 * w + li) * ST + 16 * s2 + 8 * hi]);
#pragma unroll
            for (int p = 0; p < 8; ++p) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
                    S[p] = mfma_bf16_32x32x16(frag16(&s_kT[(32 * p + li) * ST + 16 * s2 + 8 * hi]), vb2[s2], S[p]);
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const float4 d = *reinterpret_cast<const float4*>(&s_dec[32 * p + 8 * r4 + 4 * hi]);
                    S[p][4 * r4 + 0] *= d.x; S[p][4 * r4 + 1] *= d.y;
                    S[p][4 * r4 + 2] *= d.z; S[p][4 * r4 + 3] *= d.w;
                }
                cfence();
            }
        }
        __syncthreads();   // (3) o tile complete, operand tiles dead, next chunk's DMA landed
        {
            const int row = tid >> 4, seg = tid & 15;      // 16 lanes x 32 B = one 512-byte output row
            if (row < n) {
                const uint4 u0 = *reinterpret_cast<const uint4*>(&s_o[row * SQ + 16 * seg]);
                const uint4 u1 = *reinterpret_cast<const uint4*>(&s_o[row * SQ + 16 * seg + 8]);
                bf16_t* op = ob + (t0 + row) * so.t + 16 * seg;
                *reinterpret_cast<uint4*>(op) = u0;
                *reinterpret_cast<uint4*>(op + 8) = u1;
            }
        }
        __syncthreads();   // (4) s_o (aliases the scan totals) has been read
        t0 += n;
    }

    if (ht) {
        float* hp = ht + ((int64_t)bh * DK + 4 * hi) * DV + 32 * w + li;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            float* tp = hp + (32 * p) * DV;
#pragma unroll
            for (int r = 0; r < 16; ++r) tp[((r & 3) + 8 * (r >> 2)) * DV] = S[p][r];
        }
    }
}

// true when the full-head kernel can take this call (16-byte aligned rows everywhere)
static bool full_ok(int Dk, int Dv, int dtype, const void* q, const void* k, const void* v, const void* gk,
                    const void* o, int g_dtype, lina_bht_strides sq, lina_bht_strides sk, lina_bht_strides sv,
                    lina_bht_strides sg, lina_bht_strides so) {
    if (dtype != LINA_BF16 || g_dtype != LINA_BF16 || Dk != 256 || Dv != 256) return false;
    auto al = [](lina_bht_strides s, int m) { return s.b % m == 0 && s.h % m == 0 && s.t % m == 0; };
    if (!al(sq, 8) || !al(sk, 8) || !al(sv, 8) || !al(so, 8) || !al(sg, 8)) return false;
    auto p16 = [](const void* p) { return ((uintptr_t)p & 15u) == 0; };
    return p16(q) && p16(k) && p16(v) && p16(gk) && p16(o);
}

int launch_chunk_full(const void* q, const void* k, const void* v, const void* gk, void* o, const float* h0,
                      float* ht, int B, int H, int T, int Dk, int Dv, lina_bht_strides sq, lina_bht_strides sk,
                      lina_bht_strides sv, lina_bht_strides sg, lina_bht_strides so, int dtype, int g_dtype,
                      float scale, lina_stream_t stream, bool* taken) {
    *taken = full_ok(Dk, Dv, dtype, q, k, v, gk, o, g_dtype, sq, sk, sv, sg, so);
    if (!*taken) return LINA_OK;
    dim3 grid((unsigned)(B * H));
    LINA_LAUNCH(gla_chunk_bf16_h256_kernel, grid, dim3(512), 0, stream, (const bf16_t*)q, (const bf16_t*)k,
                (const bf16_t*)v, (const bf16_t*)gk, (bf16_t*)o, h0, ht, H, T, sq, sk, sv, sg, so, scale);
    return check_launch("lina_gla_chunk_fwd(full)");
}

}  // namespace lina
