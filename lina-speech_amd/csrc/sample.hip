// sample.hip -- K6c: device-side top-k / temperature sampling of one token per logits row
// (SURVEY.md 8(f) f-2).  Replaces topk_sampling(seq, k, temp) for k > 1 -- torch.topk + masked softmax +
// torch.multinomial, reference model/tools.py:38-44 called from model/modeling_lina.py:159-164 -- without
// leaving the device or the captured graph.  Same distribution as the reference, including its quirk that the
// TEMPERED logits are compared with the UNTEMPERED k-th largest value (tools.py:39-41):
//     keep_j = (x_j / temp >= kth(x)),   p_j = softmax over kept of x_j / temp,   token ~ p.
// The draw is the inverse CDF in index order of ONE uniform number per row, which either the caller supplies
// (u_ext, fp32 [rows]) or the kernel derives from (seed, *step, row) with a counter-based hash, so that a graph
// replay whose step counter lives in device memory draws fresh numbers.  (torch.multinomial's own random stream
// is not reproducible outside torch; the distribution is what is matched.)
// One 256-thread workgroup per row: the row sits in LDS as fp32; the k-th largest value comes from a 4-pass
// 8-bit radix select on order-preserving keys (exact, value-based like torch.topk(...).values[:, -1]).
#include <lina_dev.h>
#include "lina_common.h"

namespace lina {

constexpr int kSampleMaxN = 8192;

__device__ __forceinline__ uint32_t order_key(float f) {           // larger float <-> larger key (NaN last)
    const uint32_t u = __builtin_bit_cast(uint32_t, f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_value(uint32_t k) {
    return __builtin_bit_cast(float, (uint32_t)((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k));
}
// splitmix64 finaliser over (seed, step, row): 24 uniform bits -> [0, 1)
__device__ __forceinline__ float hash_uniform(uint64_t seed, uint64_t step, uint64_t row, uint64_t rows) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (step * rows + row + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}

// inclusive suffix sum over the 256 threads of the workgroup (thread t gets sum_{j >= t} v_j)
__device__ __forceinline__ int wg_suffix_sum(int v, int* s_part) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int s = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = shfl_down_i(s, d);
        if (lane + d < 64) s += o;
    }
    if (lane == 0) s_part[w] = s;
    __syncthreads();
    int add = 0;
    for (int ww = w + 1; ww < 4; ++ww) add += s_part[ww];
    __syncthreads();
    return s + add;
}
__device__ __forceinline__ float wg_prefix_sum_excl(float v, float* s_part, float* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float s = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float o = shfl_up(s, d);
        if (lane >= d) s += o;
    }
    if (lane == 63) s_part[w] = s;
    __syncthreads();
    float add = 0.0f;
    for (int ww = 0; ww < w; ++ww) add += s_part[ww];
    *total = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
    __syncthreads();
    return add + s - v;
}

template <typename T>
__global__ __launch_bounds__(256) void topk_sample_rows_kernel(const T* __restrict__ logits, int64_t row_stride,
                                                               int64_t* __restrict__ out, int n, int k, float inv_temp,
                                                               const float* __restrict__ u_ext, uint64_t seed,
                                                               const int64_t* __restrict__ step) {
    LINA_DYN_SMEM(smem_raw);
    float* s_x = reinterpret_cast<float*>(smem_raw);            // [n] the row
    __shared__ int s_hist[256];
    __shared__ int s_ipart[4];
    __shared__ float s_fpart[4];
    __shared__ uint32_t s_prefix;
    __shared__ int s_rank, s_pick;

    const int tid = threadIdx.x;
    const int64_t row = blockIdx.x;
    const T* x = logits + row * row_stride;
    for (int j = tid; j < n; j += 256) s_x[j] = ld(x + j);
    if (tid == 0) { s_prefix = 0u; s_rank = min(k, n); s_pick = -1; }
    __syncthreads();

    // ---- k-th largest value: radix select, 8 bits per pass from the top ----
#pragma unroll 1
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        const uint32_t prefix = s_prefix;
        const uint32_t himask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
        s_hist[tid] = 0;
        __syncthreads();
        for (int j = tid; j < n; j += 256) {
            const uint32_t key = order_key(s_x[j]);
            if ((key & himask) == prefix) lds_atomic_add(&s_hist[(key >> shift) & 255u], 1);
        }
        __syncthreads();
        const int cnt = s_hist[tid];
        const int rank = s_rank;
        const int suf = wg_suffix_sum(cnt, s_ipart);                 // elements in bins >= tid
        if (suf >= rank && suf - cnt < rank) {                       // exactly one bin holds the rank-th largest
            s_prefix = prefix | ((uint32_t)tid << shift);
            s_rank = rank - (suf - cnt);
        }
        __syncthreads();
    }
    const float kth = key_value(s_prefix);

    // ---- masked softmax over the kept entries; thread t owns the contiguous block [t*ept, (t+1)*ept) ----
    const int ept = (n + 255) / 256;
    const int j0 = tid * ept, j1 = min(n, j0 + ept);
    float mx = -INFINITY;
    for (int j = j0; j < j1; ++j) {
        const float l = s_x[j] * inv_temp;
        if (l >= kth) mx = fmaxf(mx, l);
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) mx = fmaxf(mx, shfl_xor(mx, m));
    if ((tid & 63) == 0) s_fpart[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(s_fpart[0], s_fpart[1]), fmaxf(s_fpart[2], s_fpart[3]));
    __syncthreads();
    float local = 0.0f;
    for (int j = j0; j < j1; ++j) {
        const float l = s_x[j] * inv_temp;
        if (l >= kth) local += __expf(l - mx);
    }
    float total;
    const float before = wg_prefix_sum_excl(local, s_fpart, &total);
    const float u = u_ext ? u_ext[row] : hash_uniform(seed, step ? (uint64_t)step[0] : 0ull, (uint64_t)row, gridDim.x);
    const float target = u * total;
    // the owner of the crossing walks its block; "before <= target < before + local"
    if (local > 0.0f && target >= before && target < before + local) {
        float c = before;
        int pick = -1;
        for (int j = j0; j < j1; ++j) {
            const float l = s_x[j] * inv_temp;
            if (l >= kth) {
                c += __expf(l - mx);
                pick = j;
                if (target < c) break;
            }
        }
        s_pick = pick;
    }
    __syncthreads();
    const bool miss = s_pick < 0;      // snapshot: every thread reads the flag BEFORE anybody's atomic may change it
    __syncthreads();
    if (miss) {                        // workgroup-uniform (all threads read the same value between two barriers)
        // rounding put the target at/after the total: take the LAST kept entry (thread order = index order)
        int last = -1;
        for (int j = j0; j < j1; ++j)
            if (s_x[j] * inv_temp >= kth) last = j;
        lds_atomic_max(&s_pick, last);
        __syncthreads();
    }
    if (tid == 0) out[row] = s_pick < 0 ? 0 : s_pick;
}

}  // namespace lina

extern "C" int lina_topk_sample_rows(const void* logits, int64_t* out, int64_t rows, int n, int64_t row_stride, int k,
                                     float temp, const float* u_ext, uint64_t seed, const int64_t* step, int dtype,
                                     lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(logits && out, "lina_topk_sample_rows: null pointer");
    LINA_REQUIRE(rows > 0 && n > 0, "lina_topk_sample_rows: rows,n must be positive");
    LINA_REQUIRE(k >= 1, "lina_topk_sample_rows: k must be >= 1 (got %d)", k);
    LINA_REQUIRE(temp > 0.0f, "lina_topk_sample_rows: temp must be positive");
    LINA_REQUIRE(valid_dtype(dtype), "lina_topk_sample_rows: bad dtype %d", dtype);
    if (n > kSampleMaxN) return fail(LINA_ERR_UNSUPPORTED, "lina_topk_sample_rows: n=%d exceeds %d", n, kSampleMaxN);
    dim3 grid((unsigned)rows);
    const size_t smem = (size_t)n * sizeof(float);
    if (dtype == LINA_F32)
        LINA_LAUNCH((topk_sample_rows_kernel<float>), grid, dim3(256), smem, stream, (const float*)logits, row_stride, out, n,
                    k, 1.0f / temp, u_ext, seed, step);
    else
        LINA_LAUNCH((topk_sample_rows_kernel<bf16_t>), grid, dim3(256), smem, stream, (const bf16_t*)logits, row_stride, out,
                    n, k, 1.0f / temp, u_ext, seed, step);
    return check_launch("lina_topk_sample_rows");
}
