// Microbenchmark: per-CU ingest rate of 16-byte-per-lane global loads by ADDRESS PATTERN (decides the operand layout of
// the decode-step projection kernels).  256 workgroups x 256 threads, one per CU; each workgroup pulls `kb` KiB:
//   P1  MFMA-fragment shaped from a ROW-MAJOR matrix: a wave instruction = 16 rows x 64 B (lane (i, g) -> row i, 16-B piece g)
//   P2  fragment-MAJOR packed matrix: a wave instruction = 1 KiB contiguous (lane l -> 16-B piece l)
// hot = every workgroup reads the same 128 KiB (L2), cold = disjoint regions of a 1 GiB buffer (HBM).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int P, int U>
__global__ __launch_bounds__(256) void pull(const uint4* __restrict__ A, size_t wg_stride16, int n_inst, unsigned* out) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint4* base = A + (size_t)blockIdx.x * wg_stride16;
    unsigned acc = 0;
    // n_inst wave-instructions per wave, U in flight
    for (int i0 = 0; i0 < n_inst; i0 += U) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = (i0 + u) * 4 + w;          // global instruction index (1 KiB each)
            size_t idx;
            if (P == 1) {  // row-major [rows][128 pieces]: instruction j covers 16 rows x 4 pieces: rows 16*(j%4).., pieces 4*(j/4)..
                const int mt = j & 3, ks = j >> 2;
                idx = (size_t)(16 * mt + (lane & 15)) * 128 + (size_t)(ks % 32) * 4 + (lane >> 4) + (size_t)(ks / 32) * 64 * 128;
            } else {
                idx = (size_t)j * 64 + lane;
            }
            v[u] = base[idx];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}
int main() {
    uint4* A; unsigned* out;
    const size_t big = (size_t)1 << 30;
    hipMalloc(&A, big); hipMalloc(&out, 1 << 20);
    hipMemset(A, 1, big);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int kb : {64, 192, 1024}) for (int cold = 0; cold <= 1; ++cold) for (int P = 1; P <= 2; ++P) for (int U : {8, 16}) {
        const int n_inst = kb / 4;                     // per wave: kb KiB / 4 waves / 1 KiB
        const size_t stride = cold ? (size_t)kb * 1024 / 16 : 0;
        const int wgs = 256;
        auto launch = [&]() {
            if (P == 1 && U == 8) hipLaunchKernelGGL((pull<1, 8>), dim3(wgs), dim3(256), 0, 0, A, stride, n_inst, out);
            if (P == 1 && U == 16) hipLaunchKernelGGL((pull<1, 16>), dim3(wgs), dim3(256), 0, 0, A, stride, n_inst, out);
            if (P == 2 && U == 8) hipLaunchKernelGGL((pull<2, 8>), dim3(wgs), dim3(256), 0, 0, A, stride, n_inst, out);
            if (P == 2 && U == 16) hipLaunchKernelGGL((pull<2, 16>), dim3(wgs), dim3(256), 0, 0, A, stride, n_inst, out);
        };
        for (int i = 0; i < 50; ++i) launch();
        hipDeviceSynchronize();
        hipEventRecord(e0); for (int i = 0; i < 200; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / 200;
        printf("kb=%4d %s P%d U=%2d: %7.2f us/launch  -> %6.1f GB/s per CU (incl. ~1.5 us boundary), chip %.2f TB/s\n", kb,
               cold ? "cold" : "hot ", P, U, us, kb * 1024.0 / us / 1e3, wgs * kb * 1024.0 / us / 1e6);
    }
    return 0;
}
