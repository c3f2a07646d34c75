// lina_dev.h -- gfx950 (CDNA4) device primitives used by the kernels in this directory.
// wave = 64 lanes; MFMA fragment layouts per /opt/skills/guides/cdna_hip_programming.md s3:
//   v_mfma_f32_16x16x4_f32   : A[i=l&15][k=l>>4]        B[k=l>>4][n=l&15]
//   v_mfma_f32_16x16x32_bf16 : A[i=l&15][k=8*(l>>4)+j]  B[k=8*(l>>4)+j][n=l&15]   (j<8)
//   C/D (both)               : col = l&15, row = 4*(l>>4) + reg
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define LINA_LAUNCH(kernel, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kernel, grid, block, smem, (hipStream_t)(stream), __VA_ARGS__)
#define LINA_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]

namespace lina {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((uint32_t)h << 16); }
// fp32 -> bf16, round-to-nearest-even, via the native __bf16 conversion (v_cvt_pk_bf16_f32 on gfx950;
// branch-free, NaN stays NaN)
__device__ __forceinline__ unsigned short f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}

__device__ __forceinline__ float shfl_xor(float v, int mask) { return __shfl_xor(v, mask, 64); }
__device__ __forceinline__ float shfl(float v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ int shfl_xor_i(int v, int mask) { return __shfl_xor(v, mask, 64); }
__device__ __forceinline__ int shfl_i(int v, int src) { return __shfl(v, src, 64); }
// value of lane+d / lane-d; a lane whose source is outside the wave gets its own value back
__device__ __forceinline__ int shfl_down_i(int v, int d) { return __shfl_down(v, d, 64); }
__device__ __forceinline__ float shfl_up(float v, int d) { return __shfl_up(v, d, 64); }
// LDS atomics on a workgroup-shared int
__device__ __forceinline__ void lds_atomic_add(int* p, int v) { atomicAdd(p, v); }
__device__ __forceinline__ void lds_atomic_max(int* p, int v) { atomicMax(p, v); }

__device__ __forceinline__ f32x4 mfma_f32_16x16x4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_bf16_16x16x32(bf16x8 a, bf16x8 b, f32x4 c) {
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                   __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

//   v_mfma_f32_32x32x16_bf16 : A[i=l&31][k=8*(l>>5)+j]  B[k=8*(l>>5)+j][n=l&31]  (j<8)
//   C/D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5), reg < 16
__device__ __forceinline__ f32x16 mfma_bf16_32x32x16(bf16x8 a, bf16x8 b, f32x16 c) {
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                   __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// Asynchronous 16-byte-per-lane global -> LDS copy (global_load_lds_dwordx4): the LDS destination is
// `lds_wave_base + lane*16` (wave-uniform base), the global source is per lane.  Completion: vmcnt;
// hipcc drains it at the next __syncthreads() (a pending LDS write), which is what the kernels rely on.
__device__ __forceinline__ void dma16_to_lds(const void* gsrc_lane, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// The same copy issued through inline assembly, addressed as (wave-uniform 64-bit base) + (32-bit byte offset per
// lane).  The compiler does not see an LDS-DMA, so it does NOT make later LDS reads or barriers wait for it (with the
// builtin every LDS read it cannot disambiguate drains vmcnt, i.e. stalls on the prefetch).  The CALLER owns the
// ordering: wait_vmem() + a workgroup barrier before anyone reads the destination.
__device__ __forceinline__ void dma16_to_lds_async(const void* base_uniform, unsigned lane_byte_off, void* lds_wave_base) {
    const unsigned lds = (unsigned)__builtin_amdgcn_readfirstlane(
        (int)(unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds_wave_base);
#if defined(LINA_DMA_NT) && LINA_DMA_NT   // non-temporal: the bytes are read once (set per kernel file before this header)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt"
#else
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
#endif
                 :
                 : "s"(lds), "v"(lane_byte_off), "s"(base_uniform)
                 : "memory", "m0");
}
// The same with the cache policy as a template argument (a kernel whose instantiations want different policies: K2's plain passes
// read every byte once -- non-temporal -- while the segment-parallel passes of a small batch read k, g, v twice within
// microseconds and are 7.5 % faster when the first pass leaves them in the caches; profiles/r06_k2_dma_nt_ab.txt)
template <bool NT>
__device__ __forceinline__ void dma16_to_lds_async_p(const void* base_uniform, unsigned lane_byte_off, void* lds_wave_base) {
    const unsigned lds = (unsigned)__builtin_amdgcn_readfirstlane(
        (int)(unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds_wave_base);
    if constexpr (NT)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt" : : "s"(lds), "v"(lane_byte_off), "s"(base_uniform) : "memory", "m0");
    else
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(lds), "v"(lane_byte_off), "s"(base_uniform) : "memory", "m0");
}
// The same, addressed by an LDS BYTE ADDRESS held as an integer: a kernel that walks a ring of LDS stages converts its base pointer
// once (lds_addr_of) and adds plain integers -- every generic -> LDS pointer conversion in a loop is a null check + select on the
// scalar pipe (s_cmp_lg_u64 / s_cselect per DMA piece in the first version of linear_tall.h).
typedef unsigned lds_addr_t;
__device__ __forceinline__ lds_addr_t lds_addr_of(void* lds_ptr) {
    return (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds_ptr);
}
__device__ __forceinline__ lds_addr_t lds_addr_add(lds_addr_t a, unsigned bytes) { return a + bytes; }
__device__ __forceinline__ void dma16_to_lds_at(const void* base_uniform, unsigned lane_byte_off, lds_addr_t lds_wave_base) {
#if defined(LINA_DMA_NT) && LINA_DMA_NT
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt"
#else
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
#endif
                 :
                 : "s"(lds_wave_base), "v"(lane_byte_off), "s"(base_uniform)
                 : "memory", "m0");
}
__device__ __forceinline__ void wait_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// ds_read_b64_tr_b16: transposing LDS read of 16-bit elements.  Every lane passes the address of an 8-byte piece; inside
// each group of 16 lanes  result[lane i][j] = piece[lane 4 j + i / 4][element i % 4]  (probed: tools/micro/tr_read.hip,
// profiles/r02_tr_read_probe.txt; the emulator's model is checked against that table): the building block for feeding
// token-contraction MFMA operands from ROW-MAJOR tiles.
__device__ __forceinline__ uint2 lds_read_tr16_b64(const void* piece) {
    typedef short v4s_ __attribute__((ext_vector_type(4)));
    const v4s_ r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_*)piece);
    return __builtin_bit_cast(uint2, r);                // (a builtin: the compiler tracks lgkmcnt for it like any LDS read)
}
// ... all but the N most recently issued vector-memory LOADS of this wave (loads return in order among themselves, so
// everything issued before the last N -- e.g. the DMA above -- has landed, whatever stores are still in flight)
template <int N>
__device__ __forceinline__ void wait_vmem_but() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// reinterpret 16 bytes of packed bf16 as an MFMA operand (no instructions)
__device__ __forceinline__ bf16x8 as_bf16x8(uint4 u) { return __builtin_bit_cast(bf16x8, u); }
__device__ __forceinline__ bf16x8 as_bf16x8(uint2 lo, uint2 hi) {
    return __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
}

// c + a.lo*b.lo + a.hi*b.hi on packed bf16 pairs, fp32 accumulate (v_dot2c_f32_bf16)
__device__ __forceinline__ float dot2_bf16(uint32_t a, uint32_t b, float c) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
}

// ---- inter-workgroup hand-off inside one launch (programming guide, Guideline 16, "8-byte agent-scope
// atomics on both sides"): payload written with relaxed agent-scope 8-byte atomic stores (sc1, write-through),
// the storing wave drains them (vmcnt(0)), ONE lane takes a relaxed agent-scope ticket; the last arriver reads
// the payload with relaxed agent-scope 8-byte atomic loads.  Placement-independent; nothing spins.
__device__ __forceinline__ void st_agent8(float* p, float a, float b) {
    const unsigned long long v = (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float2 ld_agent8(const float* p) {
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float((unsigned)(v & 0xffffffffu)), __uint_as_float((unsigned)(v >> 32)));
}
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ int ticket_agent(int* counter) {
    return __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// lane index from the hardware (v_mbcnt: no register has to stay live for it) and a wave-uniform value moved to an
// SGPR: together they let a kernel re-derive threadIdx.x anywhere instead of carrying it in (or spilling it from) a VGPR
__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Workgroup barrier that orders LDS traffic only (ds_* via lgkmcnt) and leaves global / LDS-DMA operations in flight:
// __syncthreads() also drains vmcnt, i.e. it would wait for an asynchronous global->LDS prefetch that nobody reads yet.
// Use it only where no wave touches the DMA destination before the next full __syncthreads().
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// value of lane (l - n) of the same 16-lane row, 0 for the first n lanes of a row (v_*_dpp row_shr:n, bound_ctrl): one VALU
// modifier instead of a ds_bpermute -- the building block of a 16-wide scan
template <int N>
__device__ __forceinline__ float dpp_row_shr(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x110 + N, 0xf, 0xf, true));
}

// inclusive scan of FOUR values over the 16 lanes of a DPP row, every step ONE fused add per value (v_add_f32_dpp: the
// compiler emits v_mov_dpp + add).  Lanes whose source is outside the row add 0.  The leading s_nop covers the
// VALU-write -> DPP-read hazard against whatever instruction precedes the block (the compiler does not look inside);
// within a block the four values are interleaved, so a register is re-read three instructions after its write.
#define LINA_DPP_STEP4(N)                                                                                       \
    asm("s_nop 1\n\t"                                                                                           \
        "v_add_f32_dpp %0, %0, %0 row_shr:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                  \
        "v_add_f32_dpp %1, %1, %1 row_shr:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                  \
        "v_add_f32_dpp %2, %2, %2 row_shr:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                  \
        "v_add_f32_dpp %3, %3, %3 row_shr:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1"                       \
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
__device__ __forceinline__ void row_scan4(float& a, float& b, float& c, float& d) {
    LINA_DPP_STEP4(1);
    LINA_DPP_STEP4(2);
    LINA_DPP_STEP4(4);
    LINA_DPP_STEP4(8);
}
#undef LINA_DPP_STEP4
// max as ONE v_max_f32 (fmaxf first canonicalises an operand the compiler cannot prove quiet: two instructions)
__device__ __forceinline__ float vmax_raw(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// v_perm_b32: result byte i = byte sel[i] of the 8-byte value {hi:lo} (sel 0..3 -> lo, 4..7 -> hi)
__device__ __forceinline__ unsigned byte_perm(unsigned hi, unsigned lo, unsigned sel) {
    return __builtin_amdgcn_perm(hi, lo, sel);
}
// 2^x as one v_exp_f32
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// 1/x as one v_rcp_f32 (1 ulp)
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// scheduling fence: the compiler moves NO instruction across it (hand-placed software pipelines: LDS reads issued
// several MFMAs ahead stay there); generates no code
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// instruction-issue priority of this wave among the waves of its SIMD (0 = default ... 3)
template <int P> __device__ __forceinline__ void wave_priority() { __builtin_amdgcn_s_setprio(P); }

// compiler-only fence: memory operations are not moved across it (keeps LDS-read hoisting, and with it
// register pressure, bounded in the fully unrolled MFMA loops)
__device__ __forceinline__ void cfence() { asm volatile("" ::: "memory"); }

// make a per-lane value opaque to the optimiser (same value): expressions built from it are not
// hoisted out of a loop, so loop-invariant address arithmetic is recomputed instead of living in
// (and being spilled from) VGPRs across the whole loop
__device__ __forceinline__ void opaque(int& x) { asm volatile("" : "+v"(x)); }

// the same for the raw bits of a prefetched operand (uint2 / float4 / uint4): placed AFTER a main loop it pins the bf16 -> fp32
// conversion (a USE of the loaded registers, i.e. a wait for the prefetch) behind the loop -- the compiler otherwise hoists the
// conversion to right behind the loads and the kernel waits for its epilogue operands before it starts
__device__ __forceinline__ void opaque_raw(uint2& v) { asm volatile("" : "+v"(v.x), "+v"(v.y)); }
__device__ __forceinline__ void opaque_raw(uint4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
__device__ __forceinline__ void opaque_raw(float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }

// streaming (read-once / write-once) 16-byte accesses: keep the state tile out of the caches
__device__ __forceinline__ float4 ld_nt4(const float* p) {
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
}
// the same hint on one 16-byte piece of anything / on one float (streamed weights, window history)
__device__ __forceinline__ uint4 ld_nt16(const void* p) {
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    const u32x4_t t = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return make_uint4(t[0], t[1], t[2], t[3]);
}
// ... and on one 8-byte piece (four bf16 state elements)
__device__ __forceinline__ uint2 ld_nt8(const void* p) {
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t t = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(p));
    return make_uint2(t[0], t[1]);
}
__device__ __forceinline__ void st_nt8(void* p, uint2 v) {
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t t = {v.x, v.y};
    __builtin_nontemporal_store(t, reinterpret_cast<u32x2_t*>(p));
}
__device__ __forceinline__ void st_nt16(void* p, uint4 v) {
    typedef unsigned u32x4s_t __attribute__((ext_vector_type(4)));
    u32x4s_t t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<u32x4s_t*>(p));
}
__device__ __forceinline__ float ld_nt1(const float* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void st_nt1(float* p, float v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void st_nt4(float* p, float4 v) {
    f32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(p));
}

}  // namespace lina
