// tests/emu/lina_dev.h -- TEST INFRASTRUCTURE: a CPU emulation of the small set of
// gfx950 device primitives the kernels in lina-speech_amd/csrc use (<lina_dev.h>).
//
// The product build includes lina-speech_amd/csrc/lina_dev.h (real HIP builtins) and is
// the only thing that ships.  This header shadows it when tests/emu/build_emu.py compiles
// the very same kernel sources with g++, so that the index arithmetic, LDS staging, wave64
// shuffles and MFMA fragment layouts of every kernel can be checked against the CPU oracle
// in the GPU-less build container.  Each GPU thread is a ucontext fiber; __syncthreads and
// the wave-collective primitives are cooperative yield points.  MFMA lane layouts follow
// /opt/skills/guides/cdna_hip_programming.md section 3.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <algorithm>
using std::min;
using std::max;

#define LINA_EMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) float2 { float x, y; };
struct alignas(8) ushort4 { unsigned short x, y, z, w; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(8) uint2 { unsigned x, y; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }

#define __expf(x) expf(x)
#define __logf(x) logf(x)
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __fdividef(float a, float b) { return a / b; }

namespace lina_emu {
struct Fiber;
extern Fiber* cur;
extern dim3 g_blockIdx, g_blockDim, g_gridDim;
extern unsigned char* g_dyn_smem;
const dim3& cur_tid();
int cur_lane();
void syncthreads();
bool dma_late();
void dma_defer(void* dst, const void* src);
void dma_flush_mine();
// wave-collective exchange: every lane of the wave deposits `n` 32-bit words, then reads
// the 64 x n table `out` (out[lane*n + i]).
void wave_exchange(const uint32_t* mine, int n, uint32_t* out);
void launch_impl(const std::function<void()>& body, dim3 grid, dim3 block, size_t smem);
template <class K, class... A>
void launch(K kernel, dim3 grid, dim3 block, size_t smem, void*, A... args) {
    launch_impl([=]() { kernel(args...); }, grid, block, smem);
}
}  // namespace lina_emu

#define threadIdx (lina_emu::cur_tid())
#define blockIdx (lina_emu::g_blockIdx)
#define blockDim (lina_emu::g_blockDim)
#define gridDim (lina_emu::g_gridDim)
static inline void __syncthreads() { lina_emu::syncthreads(); }

#define LINA_LAUNCH(kernel, grid, block, smem, stream, ...) \
    lina_emu::launch(kernel, grid, block, smem, stream, __VA_ARGS__)
#define LINA_DYN_SMEM(name) unsigned char* name = lina_emu::g_dyn_smem

namespace lina {

struct f32x4 {
    float v[4];
    float& operator[](int i) { return v[i]; }
    const float& operator[](int i) const { return v[i]; }
};
struct f32x16 {
    float v[16];
    float& operator[](int i) { return v[i]; }
    const float& operator[](int i) const { return v[i]; }
};
struct bf16x8 {
    short v[8];
    short& operator[](int i) { return v[i]; }
    const short& operator[](int i) const { return v[i]; }
};

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static inline float bf2f(unsigned short h) { return u2f((uint32_t)h << 16); }
static inline unsigned short f2bf(float f) {  // round-to-nearest-even, NaN kept quiet
    uint32_t u = f2u(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

static inline uint32_t pack_bf16x2(float lo, float hi) { return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16); }

static inline float shfl_xor(float v, int mask) {
    uint32_t mine = f2u(v), tab[64];
    lina_emu::wave_exchange(&mine, 1, tab);
    return u2f(tab[lina_emu::cur_lane() ^ mask]);
}
static inline float shfl(float v, int src) {
    uint32_t mine = f2u(v), tab[64];
    lina_emu::wave_exchange(&mine, 1, tab);
    return u2f(tab[src & 63]);
}
static inline int shfl_i(int v, int src) {
    uint32_t mine = (uint32_t)v, tab[64];
    lina_emu::wave_exchange(&mine, 1, tab);
    return (int)tab[src & 63];
}
static inline int shfl_down_i(int v, int d) {
    uint32_t mine = (uint32_t)v, tab[64];
    lina_emu::wave_exchange(&mine, 1, tab);
    const int src = lina_emu::cur_lane() + d;
    return src < 64 ? (int)tab[src] : v;
}
static inline float shfl_up(float v, int d) {
    uint32_t mine = f2u(v), tab[64];
    lina_emu::wave_exchange(&mine, 1, tab);
    const int src = lina_emu::cur_lane() - d;
    return src >= 0 ? u2f(tab[src]) : v;
}
// fibers are cooperative: a plain read-modify-write is atomic
static inline void lds_atomic_add(int* p, int v) { *p += v; }
static inline void lds_atomic_max(int* p, int v) { if (v > *p) *p = v; }
static inline int shfl_xor_i(int v, int mask) {
    uint32_t mine = (uint32_t)v, tab[64];
    lina_emu::wave_exchange(&mine, 1, tab);
    return (int)tab[lina_emu::cur_lane() ^ mask];
}

// v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D: col=l&15, row=4*(l>>4)+r.
static inline f32x4 mfma_f32_16x16x4(float a, float b, f32x4 c) {
    uint32_t mine[2] = {f2u(a), f2u(b)}, tab[128];
    lina_emu::wave_exchange(mine, 2, tab);
    const int l = lina_emu::cur_lane(), col = l & 15;
    f32x4 d;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(u2f(tab[(row + 16 * k) * 2]), u2f(tab[(col + 16 * k) * 2 + 1]), acc);
        d[r] = acc;
    }
    return d;
}

// v_mfma_f32_16x16x32_bf16: lane l holds A[i=l&15][k=8*(l>>4)+j], B[k=8*(l>>4)+j][n=l&15], j<8.
static inline f32x4 mfma_bf16_16x16x32(bf16x8 a, bf16x8 b, f32x4 c) {
    uint32_t mine[8], tab[512];
    for (int j = 0; j < 4; ++j) {
        mine[j] = (uint32_t)(uint16_t)a[2 * j] | ((uint32_t)(uint16_t)a[2 * j + 1] << 16);
        mine[4 + j] = (uint32_t)(uint16_t)b[2 * j] | ((uint32_t)(uint16_t)b[2 * j + 1] << 16);
    }
    lina_emu::wave_exchange(mine, 8, tab);
    const int l = lina_emu::cur_lane(), col = l & 15;
    auto A = [&](int i, int k) {
        const uint32_t w = tab[(i + 16 * (k >> 3)) * 8 + ((k & 7) >> 1)];
        return bf2f((unsigned short)((k & 1) ? (w >> 16) : (w & 0xffff)));
    };
    auto Bm = [&](int k, int n) {
        const uint32_t w = tab[(n + 16 * (k >> 3)) * 8 + 4 + ((k & 7) >> 1)];
        return bf2f((unsigned short)((k & 1) ? (w >> 16) : (w & 0xffff)));
    };
    f32x4 d;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int k = 0; k < 32; ++k) acc = fmaf(A(row, k), Bm(k, col), acc);
        d[r] = acc;
    }
    return d;
}

// v_mfma_f32_32x32x16_bf16: A[i=l&31][k=8*(l>>5)+j], B[k=8*(l>>5)+j][n=l&31];
// D: col=l&31, row=(reg&3)+8*(reg>>2)+4*(l>>5).
static inline f32x16 mfma_bf16_32x32x16(bf16x8 a, bf16x8 b, f32x16 c) {
    uint32_t mine[8], tab[512];
    for (int j = 0; j < 4; ++j) {
        mine[j] = (uint32_t)(uint16_t)a[2 * j] | ((uint32_t)(uint16_t)a[2 * j + 1] << 16);
        mine[4 + j] = (uint32_t)(uint16_t)b[2 * j] | ((uint32_t)(uint16_t)b[2 * j + 1] << 16);
    }
    lina_emu::wave_exchange(mine, 8, tab);
    const int l = lina_emu::cur_lane(), col = l & 31, hi = l >> 5;
    auto A = [&](int i, int k) {
        const uint32_t w = tab[(i + 32 * (k >> 3)) * 8 + ((k & 7) >> 1)];
        return bf2f((unsigned short)((k & 1) ? (w >> 16) : (w & 0xffff)));
    };
    auto Bm = [&](int k, int n) {
        const uint32_t w = tab[(n + 32 * (k >> 3)) * 8 + 4 + ((k & 7) >> 1)];
        return bf2f((unsigned short)((k & 1) ? (w >> 16) : (w & 0xffff)));
    };
    f32x16 d;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int k = 0; k < 16; ++k) acc = fmaf(A(row, k), Bm(k, col), acc);
        d[r] = acc;
    }
    return d;
}

static inline void dma16_to_lds(const void* gsrc_lane, void* lds_wave_base) {
    memcpy((unsigned char*)lds_wave_base + 16 * lina_emu::cur_lane(), gsrc_lane, 16);
}

// LINA_EMU_DMA_LATE=1: the copy is performed when the issuing lane reaches its wait_vmem() -- the LATEST moment the hardware
// may land it (the default, at issue, is the earliest): a kernel that reads the destination before its wait, or relies on the
// data not yet having landed, fails under one of the two.
static inline void dma16_to_lds_async(const void* base_uniform, unsigned lane_byte_off, void* lds_wave_base) {
    unsigned char* dst = (unsigned char*)lds_wave_base + 16 * lina_emu::cur_lane();
    const unsigned char* src = (const unsigned char*)base_uniform + lane_byte_off;
    if (lina_emu::dma_late()) lina_emu::dma_defer(dst, src);
    else memcpy(dst, src, 16);
}
template <bool NT>
static inline void dma16_to_lds_async_p(const void* base_uniform, unsigned lane_byte_off, void* lds_wave_base) {
    dma16_to_lds_async(base_uniform, lane_byte_off, lds_wave_base);      // (the cache policy has no meaning here)
}
// LDS byte address as a value (lina_dev.h of the product: an unsigned; here the host pointer itself)
typedef unsigned char* lds_addr_t;
static inline lds_addr_t lds_addr_of(void* lds_ptr) { return (unsigned char*)lds_ptr; }
static inline lds_addr_t lds_addr_add(lds_addr_t a, unsigned bytes) { return a + bytes; }
static inline void dma16_to_lds_at(const void* base_uniform, unsigned lane_byte_off, lds_addr_t lds_wave_base) {
    dma16_to_lds_async(base_uniform, lane_byte_off, lds_wave_base);
}
// the wave's DMA pieces have landed: on the emulator every lane copies its own 16 bytes when it runs, so this is a
// wave-wide meeting point (all lanes of a wave call it together, as on the hardware)
static inline void wait_vmem() {
    lina_emu::dma_flush_mine();
    uint32_t mine = 0, tab[64];
    lina_emu::wave_exchange(&mine, 1, tab);
}

template <int N> static inline void wait_vmem_but() { wait_vmem(); }
// ds_read_b64_tr_b16 as measured on the hardware (tools/micro/tr_read.hip, profiles/r02_tr_read_probe.txt): every lane
// supplies the LDS address of an 8-byte piece (4 x 16-bit); inside each group of 16 lanes
//     result[lane i][j] = piece[lane 4 j + i / 4][element i % 4]            (a 4 x 16 -> 16 x 4 transpose per group)
static inline uint2 lds_read_tr16_b64(const void* piece) {
    const uint64_t a = (uint64_t)(uintptr_t)piece;
    uint32_t mine[2] = {(uint32_t)a, (uint32_t)(a >> 32)}, tab[128];
    lina_emu::wave_exchange(mine, 2, tab);
    const int l = lina_emu::cur_lane(), g0 = l & ~15, i = l & 15;
    unsigned short e[4];
    for (int j = 0; j < 4; ++j) {
        const int src = g0 + 4 * j + i / 4;
        const unsigned short* p = (const unsigned short*)(uintptr_t)((uint64_t)tab[2 * src] | ((uint64_t)tab[2 * src + 1] << 32));
        e[j] = p[i % 4];
    }
    return make_uint2((uint32_t)e[0] | ((uint32_t)e[1] << 16), (uint32_t)e[2] | ((uint32_t)e[3] << 16));
}
static inline bf16x8 as_bf16x8(uint4 u) { bf16x8 r; memcpy(&r, &u, 16); return r; }
static inline bf16x8 as_bf16x8(uint2 lo, uint2 hi) { bf16x8 r; memcpy(&r.v[0], &lo, 8); memcpy(&r.v[4], &hi, 8); return r; }

static inline float dot2_bf16(uint32_t a, uint32_t b, float c) {
    return fmaf(bf2f((unsigned short)(a >> 16)), bf2f((unsigned short)(b >> 16)),
                fmaf(bf2f((unsigned short)(a & 0xffff)), bf2f((unsigned short)(b & 0xffff)), c));
}

static inline void st_agent8(float* p, float a, float b) { p[0] = a; p[1] = b; }
static inline float2 ld_agent8(const float* p) { return make_float2(p[0], p[1]); }
static inline void drain_stores() {}
static inline int ticket_agent(int* counter) { return (*counter)++; }

static inline void lds_barrier() { lina_emu::syncthreads(); }
static inline int lane_id() { return lina_emu::cur_lane(); }
static inline int wave_uniform(int v) { return v; }

template <int N>
static inline float dpp_row_shr(float v) {
    uint32_t mine = f2u(v), tab[64];
    lina_emu::wave_exchange(&mine, 1, tab);
    const int l = lina_emu::cur_lane();
    return (l & 15) >= N ? u2f(tab[l - N]) : 0.0f;
}
static inline void row_scan4(float& a, float& b, float& c, float& d) {
    a += dpp_row_shr<1>(a); b += dpp_row_shr<1>(b); c += dpp_row_shr<1>(c); d += dpp_row_shr<1>(d);
    a += dpp_row_shr<2>(a); b += dpp_row_shr<2>(b); c += dpp_row_shr<2>(c); d += dpp_row_shr<2>(d);
    a += dpp_row_shr<4>(a); b += dpp_row_shr<4>(b); c += dpp_row_shr<4>(c); d += dpp_row_shr<4>(d);
    a += dpp_row_shr<8>(a); b += dpp_row_shr<8>(b); c += dpp_row_shr<8>(c); d += dpp_row_shr<8>(d);
}
static inline float vmax_raw(float a, float b) { return a > b ? a : b; }
static inline unsigned byte_perm(unsigned hi, unsigned lo, unsigned sel) {
    const uint64_t v = ((uint64_t)hi << 32) | lo;
    unsigned r = 0;
    for (int i = 0; i < 4; ++i) r |= (unsigned)((v >> (8 * ((sel >> (8 * i)) & 7))) & 0xff) << (8 * i);
    return r;
}
static inline float fast_exp2(float x) { return exp2f(x); }
static inline float fast_rcp(float x) { return 1.0f / x; }
static inline void sched_fence() {}
template <int P> static inline void wave_priority() {}
static inline void cfence() { asm volatile("" ::: "memory"); }

static inline void opaque(int& x) { asm volatile("" : "+r"(x)); }

static inline float4 ld_nt4(const float* p) { return *reinterpret_cast<const float4*>(p); }
static inline void st_nt4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
static inline void opaque_raw(uint2&) {}
static inline void opaque_raw(uint4&) {}
static inline void opaque_raw(float4&) {}
static inline uint4 ld_nt16(const void* p) { return *reinterpret_cast<const uint4*>(p); }
static inline void st_nt16(void* p, uint4 v) { *reinterpret_cast<uint4*>(p) = v; }
static inline float ld_nt1(const float* p) { return *p; }
static inline uint2 ld_nt8(const void* p) { return *reinterpret_cast<const uint2*>(p); }
static inline void st_nt8(void* p, uint2 v) { *reinterpret_cast<uint2*>(p) = v; }
static inline void st_nt1(float* p, float v) { *p = v; }

}  // namespace lina
