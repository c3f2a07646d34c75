// Microbenchmark: read-only streaming of a 67 MB fp32 tensor (the K1w state pass) -- how do the load flavour
// (plain vs non-temporal), the workgroup shape (1024 x 64 KiB vs 256 x 256 KiB persistent) and the loads in flight
// change the achieved HBM read rate?  13 distinct 67 MB buffers are cycled (0.87 GB > MALL) like the decode step.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ f32x4 ld16(const float* p) {
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return *reinterpret_cast<const f32x4*>(p);
}
// V0: K1w-like: grid (256 bh, 4 row blocks), 256 threads, 16 loads of 16 B per thread, rows of 1 KiB
template <bool NT>
__global__ __launch_bounds__(256) void k_tile(const float* S, float* out) {
    const int tid = threadIdx.x, cg = tid % 64, rg = tid / 64;
    const float* tile = S + ((size_t)blockIdx.x * 256 + blockIdx.y * 64) * 256 + 4 * cg;
    f32x4 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = ld16<NT>(tile + (size_t)(rg + 4 * i) * 256);
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) a += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    if (a == 1.2345f) out[blockIdx.x] = a;
}
// V1: persistent: 256 (or 512) workgroups, each streams a contiguous slab with U loads in flight per thread
template <bool NT, int U>
__global__ __launch_bounds__(256) void k_slab(const float* S, float* out, size_t floats_per_wg) {
    const int tid = threadIdx.x;
    const float* base = S + (size_t)blockIdx.x * floats_per_wg + 4 * tid;
    float a = 0.f;
    for (size_t off = 0; off < floats_per_wg; off += (size_t)U * 1024) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld16<NT>(base + off + (size_t)u * 1024);
#pragma unroll
        for (int u = 0; u < U; ++u) a += v[u][0] + v[u][1] + v[u][2] + v[u][3];
    }
    if (a == 1.2345f) out[blockIdx.x] = a;
}
int main() {
    const size_t N = (size_t)64 * 4 * 256 * 256;           // floats per layer (67 MB)
    float* S; float* out;
    hipMalloc(&S, 13 * N * 4); hipMalloc(&out, 1 << 20);
    hipMemset(S, 0, 13 * N * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 26; ++i) launch(S + (size_t)(i % 13) * N);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 130; ++i) launch(S + (size_t)(i % 13) * N);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / 130;
        printf("%-44s %7.2f us/launch  %6.2f TB/s\n", name, us, N * 4.0 / us / 1e6);
    };
    run("tile 1024 WG x 64 KiB, plain", [&](float* p) { hipLaunchKernelGGL((k_tile<false>), dim3(256, 4), dim3(256), 0, 0, p, out); });
    run("tile 1024 WG x 64 KiB, nt", [&](float* p) { hipLaunchKernelGGL((k_tile<true>), dim3(256, 4), dim3(256), 0, 0, p, out); });
    run("slab 256 WG, U=8, plain", [&](float* p) { hipLaunchKernelGGL((k_slab<false, 8>), dim3(256), dim3(256), 0, 0, p, out, N / 256); });
    run("slab 256 WG, U=8, nt", [&](float* p) { hipLaunchKernelGGL((k_slab<true, 8>), dim3(256), dim3(256), 0, 0, p, out, N / 256); });
    run("slab 256 WG, U=16, plain", [&](float* p) { hipLaunchKernelGGL((k_slab<false, 16>), dim3(256), dim3(256), 0, 0, p, out, N / 256); });
    run("slab 512 WG, U=8, plain", [&](float* p) { hipLaunchKernelGGL((k_slab<false, 8>), dim3(512), dim3(256), 0, 0, p, out, N / 512); });
    run("slab 512 WG, U=8, nt", [&](float* p) { hipLaunchKernelGGL((k_slab<true, 8>), dim3(512), dim3(256), 0, 0, p, out, N / 512); });
    run("slab 1024 WG, U=8, plain", [&](float* p) { hipLaunchKernelGGL((k_slab<false, 8>), dim3(1024), dim3(256), 0, 0, p, out, N / 1024); });
    run("slab 1024 WG, U=16, plain", [&](float* p) { hipLaunchKernelGGL((k_slab<false, 16>), dim3(1024), dim3(256), 0, 0, p, out, N / 1024); });
    run("slab 2048 WG, U=8, plain", [&](float* p) { hipLaunchKernelGGL((k_slab<false, 8>), dim3(2048), dim3(256), 0, 0, p, out, N / 2048); });
    run("slab 2048 WG, U=8, nt", [&](float* p) { hipLaunchKernelGGL((k_slab<true, 8>), dim3(2048), dim3(256), 0, 0, p, out, N / 2048); });
    run("slab 4096 WG, U=4, plain", [&](float* p) { hipLaunchKernelGGL((k_slab<false, 4>), dim3(4096), dim3(256), 0, 0, p, out, N / 4096); });
    return 0;
}
