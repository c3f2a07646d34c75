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
#include "sample_dev.h"

namespace lina {

template <typename T>
__global__ __launch_bounds__(256) void topk_sample_rows_kernel(const T* __restrict__ logits, int64_t row_stride,
                                                               int64_t* __restrict__ out, int n, int k, float inv_temp,
                                                               const float* __restrict__ u_ext, uint64_t seed,
                                                               const int64_t* __restrict__ step) {
    LINA_DYN_SMEM(smem_raw);
    float* s_x = reinterpret_cast<float*>(smem_raw);            // [n] the row
    __shared__ SampleScratch sc;
    const int tid = threadIdx.x;
    const int64_t row = blockIdx.x;
    const T* x = logits + row * row_stride;
    row_to_lds(x, n, s_x);
    const float u = u_ext ? u_ext[row] : hash_uniform(seed, step ? (uint64_t)step[0] : 0ull, (uint64_t)row, gridDim.x);
    const int pick = topk_sample_block(s_x, n, k, inv_temp, u, sc);
    if (tid == 0) out[row] = pick;
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
