// Microbenchmark of the main loop of the tall projection kernels (csrc/linear_tall.h) at the decode step's shapes:
//   C[M, N] = LN?(A[M, K]) . W[N, K]^T   bf16, fragment-major ("packed") operands, fp32 accumulate, bf16 row-major output,
// timed as a CHAIN of launches inside one hipGraph (as the decode step runs them), over enough weight sets that no launch
// finds its weights in the 256 MB Infinity Cache, with shader-clock stamps per workgroup (entry / first data / loop end / exit).
//   core 0 / 2: the product's tall_core / tall_core_hyb (whatever LINA_TALL_MTW this file is compiled with)
//   core 3    : A fragments by plain loads into a register ring, W fragments global -> registers -> ds_write_b128 -> LDS
//               (no LDS-DMA: an LDS-DMA piece blocks the issuing wave 60-185 clocks, a plain 16-byte load does not)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I lina-speech_amd/csrc -I include [-DLINA_TALL_MTW=2] tools/micro/tall_gemm.hip -o tools/micro/tall_gemm
//   tools/micro/tall_gemm [M] [N] [K]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
#include "linear_tall.h"

namespace lina { char* last_error_buf() { static thread_local char b[512]; return b; } }
using namespace lina;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned long long now() { return __builtin_readcyclecounter(); }

// ---- core 3: register rings + ds_write for the shared operand
// NWV waves, wave w owns MT m-tiles (16 MT rows) for the whole contraction; G = 4 weight fragments per k-step shared by the
// workgroup, fragment (g) of a k-step fetched by wave g % NWV.  D = k-steps in flight per wave (register ring); LDS: 2 slots of
// one k-step (G KiB each), one barrier per k-step.
template <int MT, int NWV, int D, bool LN>
__device__ __forceinline__ void core_rw(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, int nb0, int nks, int mtile0, int phase,
                                        unsigned char* s_w, f32x4 (&acc)[4][MT], float (&rs1)[MT][4], float (&rs2)[MT][4],
                                        unsigned long long& t_first) {
    using F = Frag<bf16_t>;
    constexpr int G = 4, WPW = G / NWV;                   // weight fragments fetched per wave and k-step
    static_assert(G % NWV == 0, "waves share the four weight fragments evenly");
    const int lane = threadIdx.x & 63;
    const int w = wave_uniform(threadIdx.x >> 6);
    const int64_t fstr = 512;
    const bf16_t* ap[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) ap[mt] = A + ((int64_t)(mtile0 + mt) * nks * 64 + lane) * 8;
    const bf16_t* wp[WPW];
#pragma unroll
    for (int j = 0; j < WPW; ++j) wp[j] = W + ((int64_t)(nb0 + w * WPW + j) * nks * 64 + lane) * 8;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[g][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 st1[MT], st2[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { st1[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; st2[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    F f_ones;
    f_ones.ones();
    F fa[D][MT], fw[D][WPW];
    auto load = [&](int slot, int ks) {
        int kc = (ks < nks ? ks : nks - 1) + phase;         // past the end: a harmless re-read, never used
        kc = kc >= nks ? kc - nks : kc;                     // every workgroup sweeps K from its own starting point
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) fa[slot][mt].load(ap[mt] + (int64_t)kc * fstr);
#pragma unroll
        for (int j = 0; j < WPW; ++j) fw[slot][j].template load_stream<true>(wp[j] + (int64_t)kc * fstr);
    };
    auto put = [&](int slot, int lds_slot) {               // this wave's weight fragments of a k-step -> LDS
#pragma unroll
        for (int j = 0; j < WPW; ++j)
            *reinterpret_cast<uint4*>(s_w + (lds_slot * G + w * WPW + j) * 1024 + 16 * lane) = fw[slot][j].v;
    };
#pragma unroll
    for (int d = 0; d < D; ++d) load(d, d);
    put(0, 0);
    t_first = now();
    lds_barrier();
    auto step = [&](int d, int ks, bool more, bool refill) {   // one k-step; ring slot d == ks % D
        if (more) put((d + 1) % D, (ks + 1) & 1);
        F fb[G];
#pragma unroll
        for (int g = 0; g < G; ++g) fb[g].load(reinterpret_cast<const bf16_t*>(s_w + ((ks & 1) * G + g) * 1024 + 16 * lane));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (LN) {
                st1[mt] = F::mma(fa[d][mt], f_ones, st1[mt]);
                st2[mt] = F::mma(fa[d][mt], fa[d][mt], st2[mt]);
            }
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g][mt] = F::mma(fa[d][mt], fb[g], acc[g][mt]);
        }
        if (refill) load(d, ks + D);                        // refill the slot just consumed
        lds_barrier();
    };
    // exit-free main loop over whole groups of D k-steps that all have a successor group to prefetch (no condition inside: a
    // data-dependent branch in the unrolled body makes the compiler merge its wait counters to vmcnt(0) at every join)
    // (experiment: nks % D == 0 -- the loop is the ONLY code: the prefetch past the end re-reads the last k-step (clamped address,
    // never multiplied) and the last put() writes a slot nobody reads, so no tail with its own register assignment exists)
    for (int k0 = 0; k0 < nks; k0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) step(d, k0 + d, true, true);
    }
    const int lg = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            rs1[mt][r] = LN ? st1[mt][r] : 0.f;
            rs2[mt][r] = LN ? shfl(st2[mt][r], 16 * lg + 4 * lg + r) : 0.f;
        }
}

// ---- the test kernel: one workgroup = (16 MT NWV) rows x 64 columns
template <int CORE, int MT, int NWV, int D, bool LN, int EPI, int PH = 0>
__global__ __launch_bounds__(64 * NWV) void gemm_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                         const float* __restrict__ c1, const float* __restrict__ c2,
                                                         bf16_t* __restrict__ C, int M, int N, int K, int ldc,
                                                         unsigned long long* stamps) {
    using F = Frag<bf16_t>;
    constexpr int ROWS = 16 * MT * NWV;
    constexpr int LDSB = CORE == 3 ? 2 * 4 * 1024 : (CORE == 2 ? kTallNS * kTallKB * kTallNWV * 1024 : tall_lds_bytes());
    __shared__ __attribute__((aligned(16))) unsigned char s_w[LDSB < 64 * 65 * 4 ? 64 * 65 * 4 : LDSB];
    const unsigned long long t0 = now();
    unsigned long long t_first = t0;
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    const int w = wave_uniform(threadIdx.x >> 6);
    int cblk, rblk;
    if (!tall_tile_of((int)blockIdx.x, (N + 63) / 64, (M + ROWS - 1) / ROWS, cblk, rblk)) return;
    const int n0 = cblk * 64, m0 = rblk * ROWS + 16 * MT * w;
    const int nks = K / 32;
    f32x4 acc[4][MT];
    float s1[MT][4], s2[MT][4];
    if constexpr (CORE == 3) {
        const int phase = PH == 0 ? 0 : (PH == 1 ? (cblk * 5 + rblk * 11) % nks : (PH == 2 ? (rblk * (nks / 8)) % nks : ((cblk >> 3) * 4 + rblk * (nks / 8)) % nks));
        core_rw<MT, NWV, D, LN>(A, W, n0 >> 4, nks, m0 >> 4, phase, s_w, acc, s1, s2, t_first);
    } else {
        static_assert(CORE == 3 || (MT == kTallMTW && NWV == kTallNWV), "product cores: compile with the matching LINA_TALL_MTW");
        int nb[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) nb[g] = (n0 >> 4) + g;
        if constexpr (CORE == 0) tall_core<bf16_t, 4, LN>(A, W, nb, nks, m0 >> 4, true, s_w, acc, s1, s2);
        else tall_core_hyb<bf16_t, 4, LN>(A, W, nb, nks, m0 >> 4, true, s_w, acc, s1, s2);
    }
    const unsigned long long t2 = now();
    float pc1[4], pc2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { pc1[j] = c1[n0 + 16 * j + li]; pc2[j] = c2[n0 + 16 * j + li]; }
    if constexpr (EPI == 0) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float mu[4] = {0.f, 0.f, 0.f, 0.f}, rstd[4] = {1.f, 1.f, 1.f, 1.f};
            if (LN) tall_row_stats(s1[mt], s2[mt], 1.0f / (float)K, 1e-5f, mu, rstd);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 16 * mt + 4 * lg + r;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = n0 + 16 * j + li;
                    float res = acc[j][mt][r];
                    if (LN) res = rstd[r] * (res - mu[r] * pc1[j]) + pc2[j];
                    if (m < M && n < N) st(C + (int64_t)m * ldc + n, res);
                }
            }
        }
    } else {
        // rows through LDS (one 16 x 64 m-tile per wave at a time, fp32, padded), then 16-byte row-major stores
        __syncthreads();
        float (*tile)[65] = reinterpret_cast<float (*)[65]>(s_w) + 16 * w;      // 16 rows of this wave
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float mu[4] = {0.f, 0.f, 0.f, 0.f}, rstd[4] = {1.f, 1.f, 1.f, 1.f};
            if (LN) tall_row_stats(s1[mt], s2[mt], 1.0f / (float)K, 1e-5f, mu, rstd);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float res = acc[j][mt][r];
                    if (LN) res = rstd[r] * (res - mu[r] * pc1[j]) + pc2[j];
                    tile[4 * lg + r][16 * j + li] = res;
                }
            // lane -> (row = lane / 4, 16 columns = 16 * (lane % 4)): two 16-byte stores
            const int row = lane >> 2, cq = (lane & 3) * 16;
            const int m = m0 + 16 * mt + row;
            uint4 o0, o1;
            o0.x = pack_bf16x2(tile[row][cq + 0], tile[row][cq + 1]);   o0.y = pack_bf16x2(tile[row][cq + 2], tile[row][cq + 3]);
            o0.z = pack_bf16x2(tile[row][cq + 4], tile[row][cq + 5]);   o0.w = pack_bf16x2(tile[row][cq + 6], tile[row][cq + 7]);
            o1.x = pack_bf16x2(tile[row][cq + 8], tile[row][cq + 9]);   o1.y = pack_bf16x2(tile[row][cq + 10], tile[row][cq + 11]);
            o1.z = pack_bf16x2(tile[row][cq + 12], tile[row][cq + 13]); o1.w = pack_bf16x2(tile[row][cq + 14], tile[row][cq + 15]);
            if (m < M && n0 + cq + 15 < N) {
                *reinterpret_cast<uint4*>(C + (int64_t)m * ldc + n0 + cq) = o0;
                *reinterpret_cast<uint4*>(C + (int64_t)m * ldc + n0 + cq + 8) = o1;
            } else if (m < M) {
                for (int c = 0; c < 16; ++c)
                    if (n0 + cq + c < N) st(C + (int64_t)m * ldc + n0 + cq + c, tile[row][cq + c]);
            }
        }
    }
    const unsigned long long t3 = now();
    if (stamps && threadIdx.x == 0) {
        unsigned long long* s = stamps + 4 * (size_t)blockIdx.x;
        s[0] = t0; s[1] = t_first; s[2] = t2; s[3] = t3;
    }
}

// reference: one thread per output element, packed operands
__global__ void ref_kernel(const bf16_t* A, const bf16_t* W, const float* c1, const float* c2, float* C, int M, int N, int K, int ldc, int ln) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N || m >= M) return;
    double acc = 0, s1 = 0, s2 = 0;
    for (int k = 0; k < K; ++k) {
        const double a = bf2f(A[packed_off<bf16_t>(m, k, K)]), b = bf2f(W[packed_off<bf16_t>(n, k, K)]);
        acc += a * b; s1 += a; s2 += a * a;
    }
    if (ln) {
        const double mu = s1 / K, var = s2 / K - mu * mu;
        acc = (acc - mu * c1[n]) / sqrt((var > 0 ? var : 0) + 1e-5) + c2[n];
    }
    C[(int64_t)m * ldc + n] = (float)acc;
}

struct Variant { const char* name; void (*launch)(hipStream_t, const bf16_t*, const bf16_t*, const float*, const float*, bf16_t*, int, int, int, int, unsigned long long*); int rows; int ln; };

template <int CORE, int MT, int NWV, int D, bool LN, int EPI, int PH = 0>
static void launch_v(hipStream_t st, const bf16_t* A, const bf16_t* W, const float* c1, const float* c2, bf16_t* C, int M, int N, int K, int ldc, unsigned long long* stamps) {
    constexpr int ROWS = 16 * MT * NWV;
    const unsigned grid = tall_grid((N + 63) / 64, (M + ROWS - 1) / ROWS);
    hipLaunchKernelGGL((gemm_kernel<CORE, MT, NWV, D, LN, EPI, PH>), dim3(grid), dim3(64 * NWV), 0, st, A, W, c1, c2, C, M, N, K, ldc, stamps);
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 512, N = argc > 2 ? atoi(argv[2]) : 4112, K = argc > 3 ? atoi(argv[3]) : 1024;
    const int Mp = (M + 255) / 256 * 256, Np = (N + 63) / 64 * 64, ldc = Np;
    const int NSETS = 40;                                  // 40 x 8.5 MB of weights: more than the Infinity Cache holds
    const size_t a_el = (size_t)Mp * K, w_el = (size_t)Np * K;
    bf16_t *dA, *dW, *dC; float *dc1, *dc2, *dRef; unsigned long long* dSt;
    CK(hipMalloc(&dA, a_el * 2)); CK(hipMalloc(&dW, w_el * 2 * NSETS)); CK(hipMalloc(&dC, (size_t)Mp * ldc * 2));
    CK(hipMalloc(&dc1, Np * 4)); CK(hipMalloc(&dc2, Np * 4)); CK(hipMalloc(&dRef, (size_t)Mp * ldc * 4));
    CK(hipMalloc(&dSt, 4 * 8 * 4096));
    {
        std::vector<bf16_t> h(a_el > w_el ? a_el : w_el);
        unsigned s = 12345u;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
        auto tobf = [](float f) { unsigned u; memcpy(&u, &f, 4); return (bf16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); };
        for (size_t i = 0; i < a_el; ++i) h[i] = tobf(rnd());
        CK(hipMemcpy(dA, h.data(), a_el * 2, hipMemcpyHostToDevice));
        for (int sset = 0; sset < NSETS; ++sset) {
            if (sset < 2) for (size_t i = 0; i < w_el; ++i) h[i] = tobf(rnd() * 0.1f);
            CK(hipMemcpy(dW + (size_t)sset * w_el, h.data(), w_el * 2, hipMemcpyHostToDevice));
        }
        std::vector<float> c(Np);
        for (int i = 0; i < Np; ++i) c[i] = rnd();
        CK(hipMemcpy(dc1, c.data(), Np * 4, hipMemcpyHostToDevice));
        for (int i = 0; i < Np; ++i) c[i] = rnd();
        CK(hipMemcpy(dc2, c.data(), Np * 4, hipMemcpyHostToDevice));
    }
    std::vector<Variant> vs;
#define V(name, CORE, MT, NWV, D, LN, EPI) vs.push_back({name, launch_v<CORE, MT, NWV, D, LN, EPI>, 16 * MT * NWV, LN})
#define VP(name, CORE, MT, NWV, D, LN, EPI, PH) vs.push_back({name, launch_v<CORE, MT, NWV, D, LN, EPI, PH>, 16 * MT * NWV, LN})
    constexpr int PM = kTallMTW;
    V("product core0 (LDS-DMA ring)        LN", 0, PM, 4, 0, true, 0);
    V("product core0                    no-LN", 0, PM, 4, 0, false, 0);
    V("product core2 (W ring + A regs)     LN", 2, PM, 4, 0, true, 0);
    V("product core0, 16-byte-store epilog LN", 0, PM, 4, 0, true, 1);
#if LINA_TALL_MTW == 1
    V("rw  64x64  4 waves MT1 D4           LN", 3, 1, 4, 4, true, 0);
    V("rw  64x64  4 waves MT1 D8           LN", 3, 1, 4, 8, true, 0);
    V("rw  64x64  4 waves MT1 D8        no-LN", 3, 1, 4, 8, false, 0);
    V("rw 128x64  4 waves MT2 D4           LN", 3, 2, 4, 4, true, 0);
    V("rw 128x64  4 waves MT2 D8           LN", 3, 2, 4, 8, true, 0);
    V("rw 128x64  4 waves MT2 D8        no-LN", 3, 2, 4, 8, false, 0);
    V("rw 128x64  4 waves MT2 D8 epi16     LN", 3, 2, 4, 8, true, 1);
    VP("rw  64x64  MT1 D8 phase(c,r)        LN", 3, 1, 4, 8, true, 0, 1);
    VP("rw  64x64  MT1 D8 phase(r)          LN", 3, 1, 4, 8, true, 0, 2);
    VP("rw  64x64  MT1 D8 phase(c/8,r)      LN", 3, 1, 4, 8, true, 0, 3);
    VP("rw 128x64  MT2 D8 phase(c,r)        LN", 3, 2, 4, 8, true, 0, 1);
    VP("rw 128x64  MT2 D8 phase(r)          LN", 3, 2, 4, 8, true, 0, 2);
    VP("rw 128x64  MT2 D8 phase(c/8,r)      LN", 3, 2, 4, 8, true, 0, 3);
    VP("rw 128x64  MT2 D8 phase(c,r)     no-LN", 3, 2, 4, 8, false, 0, 1);
    V("rw 128x64  4 waves MT2 D16          LN", 3, 2, 4, 16, true, 0);
    V("rw 256x64  4 waves MT4 D4           LN", 3, 4, 4, 4, true, 0);
    V("rw 256x64  4 waves MT4 D8           LN", 3, 4, 4, 8, true, 0);
    V("rw  64x64  2 waves MT2 D8           LN", 3, 2, 2, 8, true, 0);
    V("rw 128x64  2 waves MT4 D8           LN", 3, 4, 2, 8, true, 0);
    V("rw 128x64  2 waves MT4 D8        no-LN", 3, 4, 2, 8, false, 0);
    V("rw  64x64  1 wave  MT4 D8           LN", 3, 4, 1, 8, true, 0);
#endif
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("M=%d N=%d K=%d  (product cores compiled with LINA_TALL_MTW=%d)\n", M, N, K, (int)kTallMTW);
    for (int ln = 0; ln < 2; ++ln) {      // references
        hipLaunchKernelGGL(ref_kernel, dim3((N + 255) / 256, M), dim3(256), 0, st, dA, dW, dc1, dc2, dRef + 0, M, N, K, ldc, ln);
        CK(hipStreamSynchronize(st));
        std::vector<float> href((size_t)Mp * ldc);
        CK(hipMemcpy(href.data(), dRef, href.size() * 4, hipMemcpyDeviceToHost));
        for (auto& v : vs) {
            if (v.ln != ln) continue;
            CK(hipMemset(dC, 0, (size_t)Mp * ldc * 2));
            v.launch(st, dA, dW, dc1, dc2, dC, M, N, K, ldc, nullptr);
            CK(hipStreamSynchronize(st));
            std::vector<bf16_t> hc((size_t)Mp * ldc);
            CK(hipMemcpy(hc.data(), dC, hc.size() * 2, hipMemcpyDeviceToHost));
            double worst = 0, big = 0;
            for (int m = 0; m < M; ++m)
                for (int n = 0; n < N; ++n) {
                    unsigned u = (unsigned)hc[(size_t)m * ldc + n] << 16; float f; memcpy(&f, &u, 4);
                    const double d = fabs(f - href[(size_t)m * ldc + n]);
                    if (d > worst) worst = d;
                    if (fabs(href[(size_t)m * ldc + n]) > big) big = fabs(href[(size_t)m * ldc + n]);
                }
            // timed: a chain of 26 launches (two per "layer") over rotating weight sets in one graph, replayed
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < NSETS; ++i) v.launch(st, dA, dW + (size_t)i * w_el, dc1, dc2, dC, M, N, K, ldc, nullptr);
            CK(hipStreamEndCapture(st, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
            CK(hipStreamSynchronize(st));
            const int reps = 10;
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / (reps * NSETS);
            // stamps of one launch
            CK(hipMemset(dSt, 0, 4 * 8 * 4096));
            v.launch(st, dA, dW + 7 * w_el, dc1, dc2, dC, M, N, K, ldc, dSt);
            CK(hipStreamSynchronize(st));
            std::vector<unsigned long long> hs(4 * 4096);
            CK(hipMemcpy(hs.data(), dSt, hs.size() * 8, hipMemcpyDeviceToHost));
            double a = 0, b = 0, c = 0; int cnt = 0; unsigned long long tmin = ~0ull, tmax = 0;
            for (int i = 0; i < 4096; ++i) {
                if (!hs[4 * i + 3]) continue;
                a += (double)(hs[4 * i + 1] - hs[4 * i]); b += (double)(hs[4 * i + 2] - hs[4 * i + 1]); c += (double)(hs[4 * i + 3] - hs[4 * i + 2]);
                if (hs[4 * i] < tmin) tmin = hs[4 * i];
                if (hs[4 * i + 3] > tmax) tmax = hs[4 * i + 3];
                ++cnt;
            }
            printf("%-42s %7.2f us  %6.1f TF/s | wgs %4d  clocks: to-first %6.0f  loop %6.0f  epilogue %6.0f  span %7llu | err %.3g (max|ref| %.3g)\n",
                   v.name, us, 2.0 * M * N * K / us * 1e-6, cnt, cnt ? a / cnt : 0, cnt ? b / cnt : 0, cnt ? c / cnt : 0,
                   cnt ? tmax - tmin : 0ull, worst, big);
            fflush(stdout);
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
    }
    return 0;
}
