// tools/micro/census.hip -- where does the dispatcher put the workgroups of a grid whose blocks fit TWO per CU (512 threads,
// <= 128 VGPRs, ~74 KB LDS: the shape of lina_gla_decode_inproj_window)?  Every block records its XCC id, HW id and start
// time and stays resident for ~20 us.  Prints, per role split (first N blocks = role A), how many CUs host 0 / 1 / 2 role-A
// blocks.  NOT part of the product library.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ __launch_bounds__(512, 4) void census(unsigned* out, long long spin) {
    __shared__ float pad[74 * 256];
    pad[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        const long long t0 = wall_clock64();
        out[blockIdx.x * 4 + 0] = xcc;
        out[blockIdx.x * 4 + 1] = hw;
        out[blockIdx.x * 4 + 2] = (unsigned)(t0 & 0xffffffffu);
        while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
        out[blockIdx.x * 4 + 3] = (unsigned)pad[5];
    }
    __syncthreads();
}

int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 448, nA = argc > 2 ? atoi(argv[2]) : 256;
    unsigned* d;
    hipMalloc(&d, grid * 16);
    std::vector<unsigned> h(grid * 4);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(d, 0, grid * 16);
        hipLaunchKernelGGL(census, dim3(grid), dim3(512), 0, 0, d, 2000LL);   // wall clock: 100 MHz -> 20 us
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, grid * 16, hipMemcpyDeviceToHost);
        std::map<unsigned, std::vector<int>> cu;     // key: xcc | se | sh | cu
        for (int b = 0; b < grid; ++b) {
            const unsigned xcc = h[b * 4] & 0xf, hw = h[b * 4 + 1];
            const unsigned cu_id = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            cu[(xcc << 12) | (se << 8) | (sh << 4) | cu_id].push_back(b);
        }
        int hist[4] = {0, 0, 0, 0}, histAll[5] = {0, 0, 0, 0, 0};
        for (auto& kv : cu) {
            int a = 0;
            for (int b : kv.second) a += b < nA;
            hist[a < 3 ? a : 3]++;
            histAll[kv.second.size() < 4 ? kv.second.size() : 4]++;
        }
        printf("rep %d: grid %d (first %d = role A): %zu CUs used; CUs with 0/1/2/3+ role-A blocks: %d %d %d %d; CUs with 1/2/3/4+ blocks: %d %d %d %d\n",
               rep, grid, nA, cu.size(), hist[0], hist[1], hist[2], hist[3], histAll[1], histAll[2], histAll[3], histAll[4]);
        if (rep == 0) {
            printf("  first 40 blocks -> (xcc, se, sh, cu), start tick:\n");
            for (int b = 0; b < 40 && b < grid; ++b) {
                const unsigned hw = h[b * 4 + 1];
                printf("   b%3d: xcc %u se %u sh %u cu %2u  t %u\n", b, h[b * 4] & 0xf, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf, h[b * 4 + 2]);
            }
            printf("  blocks on the CU of block 0, 8, 16:");
            for (int q : {0, 8, 16}) {
                const unsigned hw = h[q * 4 + 1];
                const unsigned key = ((h[q * 4] & 0xf) << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf);
                printf("  [");
                for (int b : cu[key]) printf(" %d", b);
                printf(" ]");
            }
            printf("\n");
        }
    }
    return 0;
}
