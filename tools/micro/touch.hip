// Probe kernel for tools/probe_decode.py: read `bytes` of memory (16 B per lane, whole 1 KiB per wave instruction) and
// throw the data away -- a cache warm-up ("prefetch") of a buffer another kernel is about to stream.  Not product code.
#include <hip/hip_runtime.h>
extern "C" __global__ __launch_bounds__(256) void touch_kernel(const uint4* __restrict__ p, long n16, unsigned* sink) {
    const long stride = (long)gridDim.x * 256;
    unsigned acc = 0;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 7 * stride < n16; i += 8 * stride) {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n16; i += stride) { uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x9e3779b9u) *sink = acc;
}
extern "C" int touch_launch(const void* p, long bytes, int blocks, void* sink, void* stream) {
    hipLaunchKernelGGL(touch_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)p, bytes / 16, (unsigned*)sink);
    return (int)hipGetLastError();
}
