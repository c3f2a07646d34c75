// Probe of gfx950's ds_read_b64_tr_b16 (transposing LDS read of 16-bit elements): which LDS element does (lane, j) get for
// a given per-lane address pattern?  LDS holds lds[i] = i (u16); every lane passes its own byte address; the table that
// comes back decides whether K2's row-major k~ / v tiles can feed the token-contraction MFMAs directly.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/tr_read.hip -o tools/micro/tr_read && tools/micro/tr_read
// Patterns (row stride RS elements, all in bf16 units):
//   0: addr = lane * 4                     (lane-linear 8-byte pieces: what a plain ds_read_b64 would read)
//   1: addr = (lane & 15) * RS + (lane >> 4) * 4      (16 rows of a row-major tile, 4 consecutive columns per lane group)
//   2: addr = (lane >> 4) * RS * 4 ... see code: 4 rows per lane group, 16 column quads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void probe(int pattern, int RS, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int lane = threadIdx.x;
    int el;
    if (pattern == 0) el = lane * 4;
    else if (pattern == 1) el = (lane & 15) * RS + (lane >> 4) * 4;
    else el = ((lane >> 4) * 4 + (lane & 3)) * RS + ((lane & 15) >> 2) * 4;   // 16 rows as 4 groups x 4, 4 column quads
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)&lds[el];
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[lane * 4 + 0] = (unsigned short)(v.x & 0xffff);
    out[lane * 4 + 1] = (unsigned short)(v.x >> 16);
    out[lane * 4 + 2] = (unsigned short)(v.y & 0xffff);
    out[lane * 4 + 3] = (unsigned short)(v.y >> 16);
}
int main() {
    unsigned short* d;
    hipMalloc(&d, 64 * 4 * sizeof(unsigned short));
    std::vector<unsigned short> h(256);
    const int RS = 64;
    for (int pattern = 0; pattern < 3; ++pattern) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, pattern, RS, d);
        hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
        printf("pattern %d (row stride %d elements): lane -> the 4 elements it received, as (row, col) of the row-major tile\n", pattern, RS);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d:", l);
            for (int j = 0; j < 4; ++j) printf(" (%3d,%2d)", h[l * 4 + j] / RS, h[l * 4 + j] % RS);
            printf("%s", (l & 3) == 3 ? "\n" : "");
        }
    }
    return 0;
}
