// tools/probes/k2_w12_probe.hip -- TIMING PROBE, WRONG RESULTS BY CONSTRUCTION IN PLACES, NOT PART OF THE PRODUCT.
// Question: what does a chunk of K2 at H = 4 (256 x 256 head) cost when the DMA is taken off the MFMA waves?
//   768 threads = 12 waves = 3 per SIMD (<= 168 registers):
//     waves 0..7   STATE waves: 32 state columns each (2 x 16 tiles = 128 accumulator registers), steps (1), (4), (3) of
//                  gla_chunk_full.hip for two column tiles, o stored right after step (3); NO phase A, NO DMA;
//     waves 8..11  UTILITY waves (one per SIMD): the prefetch DMA (16 pieces each), phase A of 64 channels each (four
//                  iterations of gla_chunk_full.hip's thread map, two at a time), mask(A).
// Same LDS tiles, swizzles and MFMA shapes as the product kernel; no chunk cuts, no renormalisation, no h0 / ht (the probe's
// inputs never need them).  The instruction mix per chunk is what a real 12-wave kernel would issue; the outputs are checked
// only for being finite.
#define LINA_DMA_NT 1
#include <lina_dev.h>
#include "lina_common.h"

namespace lina {

__device__ __forceinline__ void unpack4p(const uint2 u, float (&f)[4]) {
    f[0] = bf2f((bf16_t)(u.x & 0xffff)); f[1] = bf2f((bf16_t)(u.x >> 16));
    f[2] = bf2f((bf16_t)(u.y & 0xffff)); f[3] = bf2f((bf16_t)(u.y >> 16));
}
__device__ __forceinline__ bf16x8 frag16p(const bf16_t* p) { return as_bf16x8(*reinterpret_cast<const uint4*>(p)); }
__device__ __forceinline__ bf16x8 frag8x2p(const bf16_t* a, const bf16_t* b) {
    return as_bf16x8(*reinterpret_cast<const uint2*>(a), *reinterpret_cast<const uint2*>(b));
}

__global__ __launch_bounds__(768) void k2_w12_probe_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                           const bf16_t* __restrict__ v, const bf16_t* __restrict__ gk,
                                                           bf16_t* __restrict__ o, int H, int T, lina_bht_strides sq,
                                                           lina_bht_strides sk, lina_bht_strides sv, lina_bht_strides sg,
                                                           lina_bht_strides so, float scale, int ilp) {
    constexpr int DK = 256, DV = 256, C = 32;
    constexpr int SQ = DK + 16, SK = DK + 16, ST = C + 16, PE = 2 * DK + 8, RAWT = (C / 2) * PE;
    __shared__ __attribute__((aligned(16))) bf16_t s_qk[2 * C * SQ];
    bf16_t* const s_q = s_qk;
    bf16_t* const s_k = s_qk + C * SQ;
    __shared__ __attribute__((aligned(16))) bf16_t s_A[2 * 64 * 8];
    __shared__ __attribute__((aligned(16))) bf16_t s_T[(DK + DV) * ST];
    bf16_t* const s_kT = s_T;
    bf16_t* const s_vT = s_T + DK * ST;
    __shared__ __attribute__((aligned(16))) bf16_t s_raw[4 * RAWT];
    bf16_t* const s_rg = s_raw + 2 * RAWT;
    __shared__ __attribute__((aligned(16))) float s_Rs[2 * DK];
    float* const s_R = s_Rs;
    float* const s_Rn = s_Rs + DK;

    int lane = threadIdx.x & 63;
    const int w = wave_uniform(threadIdx.x >> 6);
    const bool util = w >= 8;
    const int u = w - 8;
    int li = lane & 15, lg = lane >> 4, rp = lane & 15;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const bf16_t* gsrc[4] = {q + b * sq.b + h * sq.h, k + b * sk.b + h * sk.h, gk + b * sg.b + h * sg.h, v + b * sv.b + h * sv.h};
    const unsigned gst[4] = {(unsigned)sq.t, (unsigned)sk.t, (unsigned)sg.t, (unsigned)sv.t};
    bf16_t* ob = o + b * so.b + h * so.h;

    auto dma_chunk = [&](int t_first) {                       // utility wave u: row pairs 4u .. 4u + 3 of q, k, g, v
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int pair = 4 * u + j;
                const unsigned t = (unsigned)min(t_first + 2 * pair + (lane >> 5), T - 1);
                const unsigned boff = 2u * (t * gst[a] + 8u * (unsigned)(lane & 31));
                dma16_to_lds_async(gsrc[a], boff, &s_raw[a * RAWT + pair * PE]);
            }
    };
    // ---- phase A of gla_chunk_full.hip (MODE 0, full chunks) for the channel block wv (16 channels)
    auto read_raw = [&](int wv, uint2 (&hq)[2], uint2 (&hk)[2], uint2 (&hv)[2]) {
        const bf16_t* const rawp = &s_raw[rp * PE + 16 * wv + 4 * (lane >> 4)];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            hq[rr] = *reinterpret_cast<const uint2*>(rawp + rr * DK);
            hk[rr] = *reinterpret_cast<const uint2*>(rawp + RAWT + rr * DK);
            hv[rr] = *reinterpret_cast<const uint2*>(rawp + 3 * RAWT + rr * DK);
        }
    };
    auto write_vT = [&](int wv, const uint2 (&hv)[2]) {
        bf16_t* const tp = &s_T[(DK + 16 * wv + 4 * (lane >> 4)) * ST + 2 * rp];
        *reinterpret_cast<unsigned*>(tp) = byte_perm(hv[1].x, hv[0].x, 0x05040100u);
        *reinterpret_cast<unsigned*>(tp + ST) = byte_perm(hv[1].x, hv[0].x, 0x07060302u);
        *reinterpret_cast<unsigned*>(tp + 2 * ST) = byte_perm(hv[1].y, hv[0].y, 0x05040100u);
        *reinterpret_cast<unsigned*>(tp + 3 * ST) = byte_perm(hv[1].y, hv[0].y, 0x07060302u);
    };
    auto gate_scan = [&](int wv, float (&bc)[2][4]) {
        float g0[4], g1[4];
        const bf16_t* gp = &s_rg[rp * PE + 16 * wv + 4 * (lane >> 4)];
        unpack4p(*reinterpret_cast<const uint2*>(gp), g0);
        unpack4p(*reinterpret_cast<const uint2*>(gp + DK), g1);
#pragma unroll
        for (int c = 0; c < 4; ++c) bc[1][c] = g0[c] + g1[c];
        row_scan4(bc[1][0], bc[1][1], bc[1][2], bc[1][3]);
#pragma unroll
        for (int c = 0; c < 4; ++c) bc[0][c] = bc[1][c] - g1[c];
    };
    auto write_tiles = [&](int wv, const float (&bc)[2][4], const uint2 (&hq)[2], const uint2 (&hk)[2]) {
        constexpr float kLog2e = 1.4426950408889634f;
        const int ch0 = 16 * wv + 4 * (lane >> 4);
        uint2 kk[2];
        bf16_t* const qkp = &s_qk[2 * rp * SQ + 32 * (wv >> 1) + 8 * ((lane >> 4) ^ ((rp >> 1) & 3)) + 4 * (wv & 1)];
        bf16_t* const tp = &s_T[ch0 * ST + 2 * rp];
        const float4 R4 = *reinterpret_cast<const float4*>(&s_R[ch0]);
        const float Rc[4] = {R4.x, R4.y, R4.z, R4.w};
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            float f[4], x[4], e[4], ri[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                x[c] = __builtin_fmaf(bc[rr][c], kLog2e, Rc[c]);
                e[c] = fast_exp2(x[c]);
                ri[c] = fast_rcp(e[c]);
            }
            uint2 pq;
            unpack4p(hq[rr], f);
            pq.x = pack_bf16x2(f[0] * e[0], f[1] * e[1]);
            pq.y = pack_bf16x2(f[2] * e[2], f[3] * e[3]);
            *reinterpret_cast<uint2*>(qkp + rr * SQ) = pq;
            unpack4p(hk[rr], f);
            kk[rr].x = pack_bf16x2(f[0] * ri[0], f[1] * ri[1]);
            kk[rr].y = pack_bf16x2(f[2] * ri[2], f[3] * ri[3]);
            *reinterpret_cast<uint2*>(qkp + C * SQ + rr * SQ) = kk[rr];
            if (rr == 1 && rp == C / 2 - 1) *reinterpret_cast<float4*>(&s_Rn[ch0]) = make_float4(x[0], x[1], x[2], x[3]);
        }
        *reinterpret_cast<unsigned*>(tp) = byte_perm(kk[1].x, kk[0].x, 0x05040100u);
        *reinterpret_cast<unsigned*>(tp + ST) = byte_perm(kk[1].x, kk[0].x, 0x07060302u);
        *reinterpret_cast<unsigned*>(tp + 2 * ST) = byte_perm(kk[1].y, kk[0].y, 0x05040100u);
        *reinterpret_cast<unsigned*>(tp + 3 * ST) = byte_perm(kk[1].y, kk[0].y, 0x07060302u);
    };

    for (int c = threadIdx.x; c < 2 * DK; c += 768) s_Rs[c] = 0.0f;
    if (util) { dma_chunk(0); wait_vmem(); }
    __syncthreads();
    if (util) {
        // ================= utility waves: phase A (64 channels), prefetch, mask(A) =================
        for (int t0 = 0; t0 < T; t0 += C) {
            lane = lane_id(); opaque(lane); li = lane & 15; lg = lane >> 4; rp = lane & 15;
            if (ilp == 2) {
#pragma unroll
                for (int it = 0; it < 4; it += 2) {
                    const int wa = 4 * u + it, wb = wa + 1;
                    uint2 hqa[2], hka[2], hva[2], hqb[2], hkb[2], hvb[2];
                    float bca[2][4], bcb[2][4];
                    read_raw(wa, hqa, hka, hva);
                    read_raw(wb, hqb, hkb, hvb);
                    write_vT(wa, hva);
                    write_vT(wb, hvb);
                    gate_scan(wa, bca);
                    gate_scan(wb, bcb);
                    write_tiles(wa, bca, hqa, hka);
                    write_tiles(wb, bcb, hqb, hkb);
                }
            } else {
#pragma unroll 1
                for (int it = 0; it < 4; ++it) {
                    const int wa = 4 * u + it;
                    uint2 hqa[2], hka[2], hva[2];
                    float bca[2][4];
                    read_raw(wa, hqa, hka, hva);
                    write_vT(wa, hva);
                    gate_scan(wa, bca);
                    write_tiles(wa, bca, hqa, hka);
                }
            }
            __syncthreads();   // (2) operand tiles ready; raw q,k,g,v consumed
            lane = lane_id(); opaque(lane); li = lane & 15; lg = lane >> 4; rp = lane & 15;
            if (t0 + C < T) dma_chunk(t0 + C);                 // next chunk's raw tiles: the issuing wave is blocked, nobody waits for it
            // (2) A^T tiles: utility wave u takes tile (mt = u & 1, nt = u >> 1)
            const int mt = u & 1, nt = u >> 1;
            const int pc = 8 * (lg ^ ((li >> 2) & 3));
            const bf16_t* kp = &s_k[(16 * mt + li) * SK + pc];
            const bf16_t* qp = &s_q[(16 * nt + li) * SQ + pc];
            bf16x8 kf[4], qf[4];
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) { kf[ks] = frag16p(kp + 32 * ks); qf[ks] = frag16p(qp + 32 * ks); }
            f32x4 at = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                if (ks + 3 < 8) { kf[(ks + 3) & 3] = frag16p(kp + 32 * (ks + 3)); qf[(ks + 3) & 3] = frag16p(qp + 32 * (ks + 3)); }
                sched_fence();
                at = mfma_bf16_16x16x32(kf[ks & 3], qf[ks & 3], at);
                sched_fence();
            }
            const int t = 16 * nt + li, sb = 16 * mt + 4 * lg;
            uint2 pa;
            pa.x = pack_bf16x2(sb <= t ? at[0] : 0.0f, sb + 1 <= t ? at[1] : 0.0f);
            pa.y = pack_bf16x2(sb + 2 <= t ? at[2] : 0.0f, sb + 3 <= t ? at[3] : 0.0f);
            *reinterpret_cast<uint2*>(&s_A[(nt * 64 + lane) * 8 + 4 * mt]) = pa;
            wait_vmem();
            __syncthreads();   // (3) DMA landed; operand tiles dead; mask(A) complete
        }
        return;
    }
    // ================= state waves: 32 columns each =================
    f32x4 S[2][16];
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int p = 0; p < 16; ++p) S[c2][p] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 acc[2][2] = {};
    bf16x8 vb2[2] = {};                                        // v^T fragments: step (4), and (probe: in step (4)'s token order) step (3)
    for (int t0 = 0; t0 < T; t0 += C) {
        __syncthreads();   // (2)
        lane = lane_id(); opaque(lane); li = lane & 15; lg = lane >> 4;
        if (threadIdx.x < DK) s_R[threadIdx.x] = s_Rn[threadIdx.x];
        const bf16_t* ktp = &s_kT[li * ST + 8 * lg];
        const bf16_t* qp = &s_q[li * SQ + 8 * (lg ^ ((li >> 2) & 3))];
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) { acc[c2][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[c2][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        // (1) o^T = S'^T q~^T: one K = 32 MFMA per pair of row tiles; one column tile after the other
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            bf16x8 qf[2][2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) qf[0][nt] = frag16p(qp + 16 * nt * SQ);
#pragma unroll
            for (int pp = 0; pp < 8; ++pp) {
                if (pp + 1 < 8) {
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) qf[(pp + 1) & 1][nt] = frag16p(qp + 16 * nt * SQ + 32 * (pp + 1));
                }
                sched_fence();
                bf16x8 bb;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    bb[r] = (short)f2bf(S[c2][2 * pp][r]);
                    bb[4 + r] = (short)f2bf(S[c2][2 * pp + 1][r]);
                }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[c2][nt] = mfma_bf16_16x16x32(bb, qf[pp & 1][nt], acc[c2][nt]);
                sched_fence();
            }
        }
        // (4) S' += k~^T v: the k~^T fragments shared by the two column tiles
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) vb2[c2] = frag16p(&s_vT[(32 * w + 16 * c2 + li) * ST + 8 * lg]);
        bf16x8 tf[2];
        tf[0] = frag16p(ktp);
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            if (p + 1 < 16) tf[(p + 1) & 1] = frag16p(ktp + 16 * (p + 1) * ST);
            sched_fence();
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) S[c2][p] = mfma_bf16_16x16x32(tf[p & 1], vb2[c2], S[c2][p]);
            sched_fence();
        }
        __syncthreads();   // (3)
        lane = lane_id(); opaque(lane); li = lane & 15; lg = lane >> 4;
        // (3) o += mask(A) v, then straight out (a state wave has nothing else to do until the next barrier (2))
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            acc[c2][0] = mfma_bf16_16x16x32(vb2[c2], frag16p(&s_A[(0 * 64 + lane) * 8]), acc[c2][0]);
            acc[c2][1] = mfma_bf16_16x16x32(vb2[c2], frag16p(&s_A[(1 * 64 + lane) * 8]), acc[c2][1]);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                uint2 po;
                po.x = pack_bf16x2(acc[c2][nt][0] * scale, acc[c2][nt][1] * scale);
                po.y = pack_bf16x2(acc[c2][nt][2] * scale, acc[c2][nt][3] * scale);
                const int row = t0 + 16 * nt + li;
                if (row < T) {
                    const unsigned boff = 2u * ((unsigned)row * (unsigned)so.t + 32u * (unsigned)w + 16u * (unsigned)c2 + 4u * (unsigned)lg);
                    *reinterpret_cast<uint2*>(reinterpret_cast<char*>(ob) + boff) = po;
                }
            }
        }
    }
    // keep the state alive (a real kernel returns it on request)
    float keep = 0.f;
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int p = 0; p < 16; ++p) keep += S[c2][p][0] + S[c2][p][1] + S[c2][p][2] + S[c2][p][3];
    if (keep == 1234.5f) ob[0] = 0;
}

}  // namespace lina

extern "C" int lina_k2_w12_probe(const void* q, const void* k, const void* v, const void* gk, void* o, int B, int H, int T,
                                 const int64_t* strides /* 5 x (b, h, t) */, float scale, int ilp, void* stream) {
    using namespace lina;
    lina_bht_strides s[5];
    for (int i = 0; i < 5; ++i) s[i] = lina_bht_strides{strides[3 * i], strides[3 * i + 1], strides[3 * i + 2]};
    hipLaunchKernelGGL(k2_w12_probe_kernel, dim3((unsigned)(B * H)), dim3(768), 0, (hipStream_t)stream, (const bf16_t*)q,
                       (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)gk, (bf16_t*)o, H, T, s[0], s[1], s[2], s[3], s[4], scale, ilp);
    return (int)hipGetLastError();
}
