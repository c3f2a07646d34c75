// Microbenchmark: what does a kernel boundary cost in a dependent chain when the kernels WRITE (dirty L2 lines to write
// back at the release) -- and does the store flavour change it?  256 workgroups x 256 threads, each pulls `kb` KiB of an
// L2-resident operand and writes 512 B (a 64 x 1024 bf16 activation in total); kernel i+1 reads what kernel i wrote.
//   W0: no output   W1: plain 8-byte stores   W2: non-temporal stores   W3: plain 2-byte stores   W4: nt 2-byte stores
#include <hip/hip_runtime.h>
#include <cstdio>
template <int W>
__global__ __launch_bounds__(256) void step(const uint4* __restrict__ A, int n_inst, const unsigned short* in, unsigned short* out) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    unsigned acc = in[blockIdx.x * 256 + tid];                       // depends on the previous kernel's output
    for (int i0 = 0; i0 < n_inst; i0 += 8) {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = A[(size_t)((i0 + u) * 4 + w) * 64 + lane];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    unsigned short* o = out + blockIdx.x * 256 + tid;
    if (W == 3) *o = (unsigned short)acc;
    if (W == 4) __builtin_nontemporal_store((unsigned short)acc, o);
    if (W == 1 || W == 2) {
        if ((tid & 3) == 0) {
            unsigned long long val = acc * 0x100010001ull;
            unsigned long long* o8 = reinterpret_cast<unsigned long long*>(out + blockIdx.x * 256 + tid);
            if (W == 1) *o8 = val; else __builtin_nontemporal_store(val, o8);
        }
    }
    if (W == 0 && acc == 0x12345678u) *o = 1;
}
int main() {
    uint4* A; unsigned short *b0, *b1;
    hipMalloc(&A, 1 << 20); hipMalloc(&b0, 256 * 256 * 2); hipMalloc(&b1, 256 * 256 * 2);
    hipMemset(A, 1, 1 << 20); hipMemset(b0, 0, 256 * 256 * 2); hipMemset(b1, 0, 256 * 256 * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int kb : {16, 64}) for (int W = 0; W <= 4; ++W) {
        const int n_inst = kb / 4;
        auto launch = [&](int i) {
            const unsigned short* in = (i & 1) ? b1 : b0; unsigned short* out = (i & 1) ? b0 : b1;
            switch (W) {
                case 0: hipLaunchKernelGGL(step<0>, dim3(256), dim3(256), 0, 0, A, n_inst, in, out); break;
                case 1: hipLaunchKernelGGL(step<1>, dim3(256), dim3(256), 0, 0, A, n_inst, in, out); break;
                case 2: hipLaunchKernelGGL(step<2>, dim3(256), dim3(256), 0, 0, A, n_inst, in, out); break;
                case 3: hipLaunchKernelGGL(step<3>, dim3(256), dim3(256), 0, 0, A, n_inst, in, out); break;
                default: hipLaunchKernelGGL(step<4>, dim3(256), dim3(256), 0, 0, A, n_inst, in, out); break;
            }
        };
        // timed inside a graph (as the decode step runs)
        hipStream_t st; hipStreamCreate(&st);
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        for (int i = 0; i < 64; ++i) {
            const unsigned short* in = (i & 1) ? b1 : b0; unsigned short* out = (i & 1) ? b0 : b1;
            switch (W) {
                case 0: hipLaunchKernelGGL(step<0>, dim3(256), dim3(256), 0, st, A, n_inst, in, out); break;
                case 1: hipLaunchKernelGGL(step<1>, dim3(256), dim3(256), 0, st, A, n_inst, in, out); break;
                case 2: hipLaunchKernelGGL(step<2>, dim3(256), dim3(256), 0, st, A, n_inst, in, out); break;
                case 3: hipLaunchKernelGGL(step<3>, dim3(256), dim3(256), 0, st, A, n_inst, in, out); break;
                default: hipLaunchKernelGGL(step<4>, dim3(256), dim3(256), 0, st, A, n_inst, in, out); break;
            }
        }
        hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        for (int i = 0; i < 5; ++i) hipGraphLaunch(ge, st);
        hipStreamSynchronize(st);
        hipEventRecord(e0, st); for (int i = 0; i < 20; ++i) hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // plain launches
        for (int i = 0; i < 64; ++i) launch(i);
        hipDeviceSynchronize();
        hipEventRecord(e0); for (int i = 0; i < 640; ++i) launch(i); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms2; hipEventElapsedTime(&ms2, e0, e1);
        printf("kb=%2d W%d: %6.2f us per kernel in a graph chain, %6.2f us as plain launches\n", kb, W, ms * 1e3 / (20 * 64), ms2 * 1e3 / 640);
        hipGraphExecDestroy(ge); hipGraphDestroy(g); hipStreamDestroy(st);
    }
    return 0;
}
