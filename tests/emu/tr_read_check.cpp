// TEST INFRASTRUCTURE: the probe kernel of tools/micro/tr_read.hip run on the CPU wave64 emulator; prints the same table
// as the hardware run committed in profiles/r02_tr_read_probe.txt (tests/test_emu_primitives.py compares the two).
#include <lina_dev.h>
#include <cstdio>
#include <vector>
using namespace lina;
static void probe(int pattern, int RS, unsigned short* out) {
    __shared__ unsigned short lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int lane = threadIdx.x;
    int el;
    if (pattern == 0) el = lane * 4;
    else if (pattern == 1) el = (lane & 15) * RS + (lane >> 4) * 4;
    else el = ((lane >> 4) * 4 + (lane & 3)) * RS + ((lane & 15) >> 2) * 4;
    const uint2 v = lds_read_tr16_b64(&lds[el]);
    out[lane * 4 + 0] = (unsigned short)(v.x & 0xffff);
    out[lane * 4 + 1] = (unsigned short)(v.x >> 16);
    out[lane * 4 + 2] = (unsigned short)(v.y & 0xffff);
    out[lane * 4 + 3] = (unsigned short)(v.y >> 16);
}
int main() {
    std::vector<unsigned short> h(256);
    const int RS = 64;
    for (int pattern = 0; pattern < 3; ++pattern) {
        LINA_LAUNCH(probe, dim3(1), dim3(64), 0, nullptr, pattern, RS, h.data());
        printf("pattern %d\n", pattern);
        for (int l = 0; l < 64; ++l) {
            printf("lane %d:", l);
            for (int j = 0; j < 4; ++j) printf(" (%d,%d)", h[l * 4 + j] / RS, h[l * 4 + j] % RS);
            printf("\n");
        }
    }
    return 0;
}
