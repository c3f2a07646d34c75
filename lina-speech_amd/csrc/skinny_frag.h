// skinny_frag.h -- MFMA operand fragments loaded straight from global memory (16 B per lane), shared by the
// decode-step projection kernels (linear_skinny.hip, gla_inproj.hip).
#pragma once
#include <lina_dev.h>
#include "lina_common.h"

namespace lina {

// split-K over the NW waves of a workgroup: the i-th k-step of wave w (pairs of adjacent steps per wave)
template <int NW = 4>
__device__ __forceinline__ int kstep_of(int w, int i) { return (i >> 1) * (2 * NW) + 2 * w + (i & 1); }

#ifndef LINA_SKINNY_W_NT
#define LINA_SKINNY_W_NT 0      // experiment (tools/skinny_variants.sh): weight fragments with the non-temporal load hint
#endif
template <typename T> struct Frag;
template <> struct Frag<bf16_t> {
    static constexpr int KSTEP = 32, KL = 8;  // k per step / k per lane
    uint4 v;
    __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const uint4*>(p); }
    // weight fragment, optionally with the non-temporal hint: a streamed weight matrix then does
    // not displace the matrices that are meant to stay in the 256 MB Infinity Cache between two tokens
    // (NT is a COMPILE-TIME choice: a launch-uniform runtime flag here cost the in-projection its load scheduling)
    template <bool NT>
    __device__ __forceinline__ void load_stream(const bf16_t* p) {
        if constexpr (NT || LINA_SKINNY_W_NT) v = ld_nt16(p);
        else load(p);
    }
    __device__ __forceinline__ void zero() { v = make_uint4(0u, 0u, 0u, 0u); }
    __device__ __forceinline__ void ones() { v = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u); }
    __device__ __forceinline__ void stats(float& s1, float& s2) const {
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s1 = dot2_bf16(w[j], 0x3f803f80u, s1);   // (1.0, 1.0) in bf16
            s2 = dot2_bf16(w[j], w[j], s2);
        }
    }
    __device__ __forceinline__ bf16x8 pack() const { return as_bf16x8(v); }   // a register reinterpretation
    static __device__ __forceinline__ f32x4 mma(const Frag& a, const Frag& b, f32x4 c) {
        return mfma_bf16_16x16x32(a.pack(), b.pack(), c);
    }
};
template <> struct Frag<float> {
    static constexpr int KSTEP = 16, KL = 4;
    float4 v;
    __device__ __forceinline__ void load(const float* p) { v = *reinterpret_cast<const float4*>(p); }
    template <bool NT>
    __device__ __forceinline__ void load_stream(const float* p) {
        if constexpr (NT || LINA_SKINNY_W_NT) v = ld_nt4(p);
        else load(p);
    }
    __device__ __forceinline__ void zero() { v = make_float4(0.f, 0.f, 0.f, 0.f); }
    __device__ __forceinline__ void ones() { v = make_float4(1.f, 1.f, 1.f, 1.f); }
    __device__ __forceinline__ void stats(float& s1, float& s2) const {
        s1 += (v.x + v.y) + (v.z + v.w);
        s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    // k-slot (step s, lane group g) <-> k0 + 4g + s on both operands: any bijection is valid
    static __device__ __forceinline__ f32x4 mma(const Frag& a, const Frag& b, f32x4 c) {
        c = mfma_f32_16x16x4(a.v.x, b.v.x, c);
        c = mfma_f32_16x16x4(a.v.y, b.v.y, c);
        c = mfma_f32_16x16x4(a.v.z, b.v.z, c);
        return mfma_f32_16x16x4(a.v.w, b.v.w, c);
    }
};

// ---- fragment-major ("packed") operand layout ---------------------------------------------------------------------
// A row-major operand makes a fragment load touch 16 rows x 64 B: sixteen different cache lines per quarter wave, and the
// vector memory pipe then moves ~16 B per clock per CU (measured: ~30 GB/s per CU, tools/micro/load_pattern2.hip).  In
// the packed layout the 16 rows x KSTEP columns of one fragment are ONE contiguous 1 KiB block, lane-linear:
//     element (m, k)  ->  ((m/16 * K/KSTEP + k/KSTEP) * 64 + (m%16) + 16*((k%KSTEP)/KL)) * KL + k%KL
// so a wave instruction reads 1 KiB contiguous (2x the rate for L2-resident operands).  Weights are packed once at
// engine construction; activations are written packed by the epilogue that produces them.  Rows are padded to 64.
template <typename T>
__device__ __forceinline__ int64_t packed_off(int m, int k, int K) {
    constexpr int KL = Frag<T>::KL, KS = Frag<T>::KSTEP;
    return (((int64_t)(m >> 4) * (K / KS) + (k / KS)) * 64 + (m & 15) + 16 * ((k % KS) / KL)) * KL + (k % KL);
}

}  // namespace lina
