// sample_dev.h -- block-level pieces of the device-side token picks (K6b arg-max, K6c top-k / temperature sampling),
// shared by the standalone row kernels (sample.hip, codec_head.hip) and the one-launch token epilogues (codec_head.hip).
// Semantics: reference model/tools.py:38-44 (topk_sampling) as called from model/modeling_lina.py:159-164.
#pragma once
#include <lina_dev.h>
#include "lina_common.h"

namespace lina {

constexpr int kSampleMaxN = 8192;

struct SampleScratch {       // static LDS of one 256-thread workgroup
    int hist[256];
    int ipart[4];
    float fpart[4];
    uint32_t prefix;
    int rank, pick;
};

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


// One top-k / temperature draw from the row s_x[0..n) (fp32, in LDS, filled by the caller BEFORE a barrier or by this
// workgroup's own threads -- the function starts with a barrier) by the 256 threads of the workgroup: the k-th largest
// value by a 4-pass 8-bit radix select on order-preserving keys (exact, value-based like torch.topk(...).values[:, -1]),
// a masked softmax over the kept entries and the inverse CDF of the uniform number u in index order.  Every thread
// returns the picked index.  Ends with a barrier: s_x and the scratch may be reused right away.
__device__ __forceinline__ int topk_sample_block(const float* s_x, int n, int k, float inv_temp, float u, SampleScratch& sc) {
    const int tid = threadIdx.x;
    __syncthreads();                                            // s_x complete, scratch of an earlier draw consumed
    if (tid == 0) { sc.prefix = 0u; sc.rank = min(k, n); sc.pick = -1; }
    __syncthreads();
#pragma unroll 1
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        const uint32_t prefix = sc.prefix;
        const uint32_t himask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
        sc.hist[tid] = 0;
        __syncthreads();
        for (int j = tid; j < n; j += 256) {
            const uint32_t key = order_key(s_x[j]);
            if ((key & himask) == prefix) lds_atomic_add(&sc.hist[(key >> shift) & 255u], 1);
        }
        __syncthreads();
        const int cnt = sc.hist[tid];
        const int rank = sc.rank;
        const int suf = wg_suffix_sum(cnt, sc.ipart);                // elements in bins >= tid
        if (suf >= rank && suf - cnt < rank) {                       // exactly one bin holds the rank-th largest
            sc.prefix = prefix | ((uint32_t)tid << shift);
            sc.rank = rank - (suf - cnt);
        }
        __syncthreads();
    }
    const float kth = key_value(sc.prefix);

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
    if ((tid & 63) == 0) sc.fpart[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(sc.fpart[0], sc.fpart[1]), fmaxf(sc.fpart[2], sc.fpart[3]));
    __syncthreads();
    float local = 0.0f;
    for (int j = j0; j < j1; ++j) {
        const float l = s_x[j] * inv_temp;
        if (l >= kth) local += __expf(l - mx);
    }
    float total;
    const float before = wg_prefix_sum_excl(local, sc.fpart, &total);
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
        sc.pick = pick;
    }
    __syncthreads();
    const bool miss = sc.pick < 0;     // snapshot: every thread reads the flag BEFORE anybody's atomic may change it
    __syncthreads();
    if (miss) {                        // workgroup-uniform (all threads read the same value between two barriers)
        // rounding put the target at/after the total: take the LAST kept entry (thread order = index order)
        int last = -1;
        for (int j = j0; j < j1; ++j)
            if (s_x[j] * inv_temp >= kth) last = j;
        lds_atomic_max(&sc.pick, last);
        __syncthreads();
    }
    const int pick = sc.pick < 0 ? 0 : sc.pick;
    __syncthreads();
    return pick;
}

// thread-local part of a row arg-max: elements tid, tid + 256, ... in ascending order (lowest index wins a tie).  The
// loads go out in rounds of kArgmaxInFlight per thread -- 20 x 256 covers the 4099 codec logits in ONE round trip; the plain
// loop (one 2-byte load per iteration, the compare chain behind it) paid one L2 round trip per 256 elements.
constexpr int kArgmaxInFlight = 20;
template <typename T>
__device__ __forceinline__ void argmax_scan(const T* row, int n, float& best, int& bi) {
    const int tid = threadIdx.x;
    for (int j0 = tid; j0 < n; j0 += 256 * kArgmaxInFlight) {
        float v[kArgmaxInFlight];
#pragma unroll
        for (int u = 0; u < kArgmaxInFlight; ++u) {          // unconditional loads on clamped indices: a predicated load
            const int j = j0 + 256 * u;                        // (`j < n ? ld : x`) waits for its data before the next is issued
            v[u] = ld(row + (j < n ? j : n - 1));
        }
#pragma unroll
        for (int u = 0; u < kArgmaxInFlight; ++u) {
            const int j = j0 + 256 * u;
            if (j < n && (v[u] > best || (v[u] == best && j < bi))) { best = v[u]; bi = j; }
        }
    }
}

// row[0..n) -> s_x (fp32, LDS) by the 256 threads, every load in flight at once like argmax_scan
template <typename T>
__device__ __forceinline__ void row_to_lds(const T* row, int n, float* s_x) {
    const int tid = threadIdx.x;
    for (int j0 = tid; j0 < n; j0 += 256 * kArgmaxInFlight) {
        float v[kArgmaxInFlight];
#pragma unroll
        for (int u = 0; u < kArgmaxInFlight; ++u) {
            const int j = j0 + 256 * u;
            v[u] = ld(row + (j < n ? j : n - 1));
        }
#pragma unroll
        for (int u = 0; u < kArgmaxInFlight; ++u) {
            const int j = j0 + 256 * u;
            if (j < n) s_x[j] = v[u];
        }
    }
}

// arg-max of row[0..n) (first index on ties; 0 when every entry is -inf / NaN) by the 256 threads of the workgroup;
// s_val / s_idx: 4-entry LDS scratch.  Every thread returns the index.  Ends with a barrier.
template <typename T>
__device__ __forceinline__ int argmax_block(const T* row, int n, float* s_val, int* s_idx) {
    const int tid = threadIdx.x;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    argmax_scan(row, n, best, bi);
    if (bi == 0x7fffffff && tid < n) bi = tid;      // all -inf/NaN in this thread's slice
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const float ov = shfl_xor(best, m);
        const int oi = shfl_xor_i(bi, m);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    __syncthreads();                                // scratch of an earlier call consumed
    if ((tid & 63) == 0) { s_val[tid >> 6] = best; s_idx[tid >> 6] = bi; }
    __syncthreads();
    best = s_val[0]; bi = s_idx[0];
    for (int wv = 1; wv < 4; ++wv)
        if (s_val[wv] > best || (s_val[wv] == best && s_idx[wv] < bi)) { best = s_val[wv]; bi = s_idx[wv]; }
    __syncthreads();
    return bi == 0x7fffffff ? 0 : bi;
}

}  // namespace lina
