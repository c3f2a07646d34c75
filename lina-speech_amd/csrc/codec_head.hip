// codec_head.hip -- K6: codec-token embedding gather-sum and the greedy arg-max pick.
//
// K6a replaces MultiEmbedding.forward + sum over quantizers (reference model/multiembed.py:21-23,
// model/modeling_lina.py:131,178-179); K6b replaces topk_sampling(k=1) (reference model/tools.py:38-44,
// model/modeling_lina.py:159-164).  SURVEY.md 8(a) a-10.  Both are tiny, latency-bound device-side
// steps of the decode loop; they exist so the loop needs no host round trip.
#include <lina_dev.h>
#include "lina_common.h"

namespace lina {

template <typename T>
__global__ __launch_bounds__(256) void embed_sum_kernel(const int64_t* __restrict__ idx, const T* __restrict__ table,
                                                        T* __restrict__ out, int Q, int64_t N, int n_emb, int d) {
    const int64_t n = blockIdx.x;
    for (int e = threadIdx.x * 4; e < d; e += 256 * 4) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int qi = 0; qi < Q; ++qi) {
            int64_t tok = idx[(int64_t)qi * N + n];
            tok = tok < 0 ? 0 : (tok >= n_emb ? n_emb - 1 : tok);  // never read out of bounds
            const float4 r = ld4(table + ((int64_t)qi * n_emb + tok) * d + e);
            acc.x += r.x; acc.y += r.y; acc.z += r.z; acc.w += r.w;
        }
        st4(out + n * d + e, acc);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void argmax_rows_kernel(const T* __restrict__ logits, int64_t* __restrict__ out,
                                                          int n, int64_t row_stride) {
    __shared__ float s_val[4];
    __shared__ int s_idx[4];
    const T* row = logits + (int64_t)blockIdx.x * row_stride;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int j = threadIdx.x; j < n; j += 256) {
        const float vj = ld(row + j);
        if (vj > best || (vj == best && j < bi)) { best = vj; bi = j; }
    }
    if (bi == 0x7fffffff && threadIdx.x < n) bi = threadIdx.x;  // all -inf/NaN in this thread's slice
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const float ov = shfl_xor(best, m);
        const int oi = shfl_xor_i(bi, m);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { s_val[threadIdx.x >> 6] = best; s_idx[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int wv = 1; wv < 4; ++wv)
            if (s_val[wv] > best || (s_val[wv] == best && s_idx[wv] < bi)) { best = s_val[wv]; bi = s_idx[wv]; }
        out[blockIdx.x] = (bi == 0x7fffffff) ? 0 : bi;
    }
}

}  // namespace lina

extern "C" int lina_embed_sum(const int64_t* idx, const void* table, void* out, int Q, int64_t N, int n_emb, int d,
                              int dtype, lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(idx && table && out, "lina_embed_sum: null pointer");
    LINA_REQUIRE(Q > 0 && N > 0 && n_emb > 0, "lina_embed_sum: Q,N,n_emb must be positive");
    LINA_REQUIRE(d > 0 && d % 4 == 0, "lina_embed_sum: d=%d must be a positive multiple of 4", d);
    LINA_REQUIRE(valid_dtype(dtype), "lina_embed_sum: bad dtype %d", dtype);
    dim3 grid((unsigned)N);
    if (dtype == LINA_F32)
        LINA_LAUNCH((embed_sum_kernel<float>), grid, dim3(256), 0, stream, idx, (const float*)table, (float*)out, Q, N, n_emb, d);
    else
        LINA_LAUNCH((embed_sum_kernel<bf16_t>), grid, dim3(256), 0, stream, idx, (const bf16_t*)table, (bf16_t*)out, Q, N, n_emb, d);
    return check_launch("lina_embed_sum");
}

extern "C" int lina_argmax_rows(const void* logits, int64_t* out, int64_t rows, int n, int64_t row_stride, int dtype,
                                lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(logits && out, "lina_argmax_rows: null pointer");
    LINA_REQUIRE(rows > 0 && n > 0, "lina_argmax_rows: rows,n must be positive");
    LINA_REQUIRE(valid_dtype(dtype), "lina_argmax_rows: bad dtype %d", dtype);
    dim3 grid((unsigned)rows);
    if (dtype == LINA_F32)
        LINA_LAUNCH((argmax_rows_kernel<float>), grid, dim3(256), 0, stream, (const float*)logits, out, n, row_stride);
    else
        LINA_LAUNCH((argmax_rows_kernel<bf16_t>), grid, dim3(256), 0, stream, (const bf16_t*)logits, out, n, row_stride);
    return check_launch("lina_argmax_rows");
}
