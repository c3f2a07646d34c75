// Throwaway microbenchmark: how fast can a workgroup pull a 64 x 1024 bf16 tile (128 KiB, L2-resident, shared by
// all workgroups) when the 16-byte-per-lane loads are (P1) MFMA-fragment shaped (16 rows x 64 B per wave
// instruction) vs (P2) row-contiguous (1 KiB per wave instruction)?  Decides how linear_skinny should fetch A.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int P>
__global__ __launch_bounds__(256) void pull(const uint4* __restrict__ A, unsigned* out, int reps) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    unsigned acc = 0;
    for (int r = 0; r < reps; ++r) {
        uint4 v[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            size_t idx;   // in units of 16 B; a row = 128 units
            if (P == 1) {  // fragment: wave w takes k-steps w, w+4, ...: step ks=j>>2... mt=j&3 ; lane (i=lane&15, g=lane>>4)
                const int mt = j & 3, ks = w + 4 * (j >> 2);
                idx = (size_t)(16 * mt + (lane & 15)) * 128 + ks * 4 + (lane >> 4);
            } else {       // contiguous: each instruction = 1 KiB of one row-half: row = 8*w*... cover all 64 rows x 2 halves
                const int chunk = w * 32 + j;            // 128 chunks of 1 KiB
                idx = (size_t)chunk * 64 + lane;
            }
            v[j] = A[idx + (size_t)(r & 1) * 0];
        }
#pragma unroll
        for (int j = 0; j < 32; ++j) acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}
int main() {
    uint4* A; unsigned* out;
    hipMalloc(&A, 128 * 1024); hipMalloc(&out, 1 << 20);
    hipMemset(A, 1, 128 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs : {320, 2560}) for (int P = 1; P <= 2; ++P) {
        const int reps = 8;
        auto launch = [&]() { if (P == 1) hipLaunchKernelGGL(pull<1>, dim3(wgs), dim3(256), 0, 0, A, out, reps);
                              else hipLaunchKernelGGL(pull<2>, dim3(wgs), dim3(256), 0, 0, A, out, reps); };
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int i = 0; i < 20; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / 20, bytes = (double)wgs * reps * 128 * 1024;
        printf("P%d wgs=%d: %.2f us/launch, %.2f TB/s L2->CU, %.2f us per 128KiB tile per WG-slot\n", P, wgs, us, bytes / us / 1e6,
               us / reps / ((wgs + 255) / 256));
    }
    return 0;
}
