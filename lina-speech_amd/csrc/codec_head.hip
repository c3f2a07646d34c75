// codec_head.hip -- K6: codec-token embedding gather-sum and the greedy arg-max pick.
//
// K6a replaces MultiEmbedding.forward + sum over quantizers (reference model/multiembed.py:21-23,
// model/modeling_lina.py:131,178-179); K6b replaces topk_sampling(k=1) (reference model/tools.py:38-44,
// model/modeling_lina.py:159-164).  SURVEY.md 8(a) a-10.  Both are tiny, latency-bound device-side
// steps of the decode loop; they exist so the loop needs no host round trip.
#include "sample_dev.h"
#include "skinny_frag.h"

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
    argmax_scan(row, n, best, bi);
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

// Loop-control block of the device-side decode loop (include/lina_gla.h LINA_LOOP_CTL_*): int32 words
//   [0] rows that have emitted the stop token so far   [1] first step at which ALL rows had (or -1)
//   [2],[3] per-call seed word (lo, hi)                [4 + b] row b has stopped
// -- the reference's `is_stop_token = (q_sampled == stop_token).prod(0)`, `all_stop_token |= ...`, `if all_stop_token.prod(): break`
// (model/modeling_lina.py:168-173) kept on the device: the host reads word [1] every few steps instead of syncing every step.
constexpr int kCtlCount = 0, kCtlStopAt = 1, kCtlSeedLo = 2, kCtlSeedHi = 3, kCtlRows = 4, kStopToken = 2;
__device__ __forceinline__ void note_stop(int* ctl, const int* toks, int Q, int b, int B, int64_t t) {
    bool stop = true;
    for (int qi = 0; qi < Q; ++qi) stop = stop && toks[qi] == kStopToken;
    if (!stop || ctl[kCtlRows + b]) return;
    ctl[kCtlRows + b] = 1;
    if (ticket_agent(ctl + kCtlCount) == B - 1) ctl[kCtlStopAt] = (int)t;   // this row was the last one still running
}

// K6d -- the whole token epilogue of a greedy decode step in ONE launch (one workgroup per batch row): arg-max of every
// quantizer's logits (K6b), the pick appended to the device-side token log at position step[0], the next step's input
// embedding sum_q table[q, pick_q] (K6a) written into the residual-stream buffer, and -- by the LAST workgroup to finish,
// when every workgroup has read it -- step[0] += 1.  Replaces five launches (arg-max, transpose copy, index_copy_, add_,
// embedding gather) of ~5 us each on the serial chain of the step (reference model/modeling_lina.py:159-179).
template <typename T>
__global__ __launch_bounds__(256) void greedy_pick_embed_kernel(const T* __restrict__ logits, int64_t row_stride,
                                                                const T* __restrict__ table, T* __restrict__ x_out,
                                                                int64_t* __restrict__ tok_log, int64_t* step, int* counter,
                                                                int Q, int L, int n_emb, int d, int max_steps,
                                                                T* __restrict__ x_pk, int* ctl) {
    __shared__ float s_val[4];
    __shared__ int s_idx[4];
    __shared__ int s_tok[16];
    __shared__ int64_t s_step;
    const int b = blockIdx.x, B = gridDim.x, tid = threadIdx.x;
    if (tid == 0) s_step = step[0];
    for (int qi = 0; qi < Q; ++qi) {
        const T* row = logits + (int64_t)b * row_stride + (int64_t)qi * L;
        float best = -INFINITY;
        int bi = 0x7fffffff;
        argmax_scan(row, L, best, bi);              // every load of the row in flight at once (sample_dev.h)
        if (bi == 0x7fffffff && tid < L) bi = tid;  // all -inf/NaN in this thread's slice
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const float ov = shfl_xor(best, m);
            const int oi = shfl_xor_i(bi, m);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        __syncthreads();                            // s_val / s_idx of the previous quantizer consumed
        if ((tid & 63) == 0) { s_val[tid >> 6] = best; s_idx[tid >> 6] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int wv = 1; wv < 4; ++wv)
                if (s_val[wv] > best || (s_val[wv] == best && s_idx[wv] < bi)) { best = s_val[wv]; bi = s_idx[wv]; }
            bi = (bi == 0x7fffffff) ? 0 : bi;
            s_tok[qi] = bi;
            const int64_t t = s_step;
            if (t >= 0 && t < max_steps) tok_log[(t * Q + qi) * B + b] = bi;
        }
    }
    __syncthreads();
    for (int e = tid * 4; e < d; e += 256 * 4) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int qi = 0; qi < Q; ++qi) {
            int tok = s_tok[qi];
            tok = tok >= n_emb ? n_emb - 1 : tok;
            const float4 r = ld4(table + ((int64_t)qi * n_emb + tok) * d + e);
            acc.x += r.x; acc.y += r.y; acc.z += r.z; acc.w += r.w;
        }
        st4(x_out + (int64_t)b * d + e, acc);
        if (x_pk) {                                  // fragment-major copy (the model-dtype values just stored)
            T tmp4[4];
            st4(tmp4, acc);
            st4(x_pk + packed_off<T>(b, e, d), ld4(tmp4));
        }
    }
    if (tid == 0) {
        if (ctl) note_stop(ctl, s_tok, Q, b, B, s_step);
        const int tk = ticket_agent(counter);       // taken AFTER this workgroup has read step[0]
        if (tk == B - 1) {
            *counter = 0;                            // re-armed for the next launch
            step[0] = s_step + 1;
        }
    }
}

// K6e -- the token epilogue of a decode step in the reference's DEFAULT generation mode (model/modeling_lina.py:119-121,
// 159-164: quantizers q < first_greedy_quant are SAMPLED with top-k / temperature, the others take the arg-max), in one
// launch like K6d: per batch row the Q picks (K6c on the sampled quantizers with the uniform number of row b*Q + q of a
// [B*Q]-row lina_topk_sample_rows call at the same (seed, step) -- the tokens are the ones the separate launches give --,
// K6b on the others), the token log, the next-input embedding (row-major and, optionally, fragment-major) and step[0] += 1.
template <typename T>
__global__ __launch_bounds__(256) void sample_pick_embed_kernel(const T* __restrict__ logits, int64_t row_stride,
                                                                const T* __restrict__ table, T* __restrict__ x_out,
                                                                int64_t* __restrict__ tok_log, int64_t* step, int* counter,
                                                                int Q, int L, int n_emb, int d, int max_steps,
                                                                T* __restrict__ x_pk, int n_sampled, int k, float inv_temp,
                                                                uint64_t seed, int* ctl) {
    LINA_DYN_SMEM(smem_raw);
    float* s_x = reinterpret_cast<float*>(smem_raw);            // [L] the row being sampled
    __shared__ SampleScratch sc;
    __shared__ float s_val[4];
    __shared__ int s_idx[4];
    __shared__ int s_tok[16];
    __shared__ int64_t s_step;
    const int b = blockIdx.x, B = gridDim.x, tid = threadIdx.x;
    if (tid == 0) s_step = step[0];
    __syncthreads();
    const int64_t t = s_step;
    // the loop-control block carries a per-call seed word (XORed in) so that a captured launch can be re-seeded
    if (ctl) seed ^= ((uint64_t)(uint32_t)ctl[kCtlSeedLo]) | ((uint64_t)(uint32_t)ctl[kCtlSeedHi] << 32);
    for (int qi = 0; qi < Q; ++qi) {
        const T* row = logits + (int64_t)b * row_stride + (int64_t)qi * L;
        int pick;
        if (qi < n_sampled) {
            row_to_lds(row, L, s_x);
            const float u = hash_uniform(seed, (uint64_t)t, (uint64_t)b * Q + qi, (uint64_t)B * Q);
            pick = topk_sample_block(s_x, L, k, inv_temp, u, sc);
        } else {
            pick = argmax_block(row, L, s_val, s_idx);
        }
        if (tid == 0) {
            s_tok[qi] = pick;
            if (t >= 0 && t < max_steps) tok_log[(t * Q + qi) * B + b] = pick;
        }
    }
    __syncthreads();
    for (int e = tid * 4; e < d; e += 256 * 4) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int qi = 0; qi < Q; ++qi) {
            int tok = s_tok[qi];
            tok = tok >= n_emb ? n_emb - 1 : tok;
            const float4 r = ld4(table + ((int64_t)qi * n_emb + tok) * d + e);
            acc.x += r.x; acc.y += r.y; acc.z += r.z; acc.w += r.w;
        }
        st4(x_out + (int64_t)b * d + e, acc);
        if (x_pk) {                                  // fragment-major copy (the model-dtype values just stored)
            T tmp4[4];
            st4(tmp4, acc);
            st4(x_pk + packed_off<T>(b, e, d), ld4(tmp4));
        }
    }
    if (tid == 0) {
        if (ctl) note_stop(ctl, s_tok, Q, b, B, t);
        const int tk = ticket_agent(counter);       // taken AFTER this workgroup has read step[0]
        if (tk == B - 1) {
            *counter = 0;                            // re-armed for the next launch
            step[0] = t + 1;
        }
    }
}

}  // namespace lina

extern "C" int lina_greedy_pick_embed(const void* logits, int64_t row_stride, const void* table, void* x_out,
                                      void* x_out_packed, int64_t* tok_log, int64_t* step, int* counter, int* loop_ctl, int B,
                                      int Q, int L, int n_emb, int d, int max_steps, int dtype, lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(logits && table && x_out && tok_log && step && counter, "lina_greedy_pick_embed: null pointer");
    LINA_REQUIRE(B > 0 && Q > 0 && Q <= 16 && L > 0 && n_emb > 0 && max_steps > 0,
                 "lina_greedy_pick_embed: B,Q (<= 16),L,n_emb,max_steps must be positive");
    LINA_REQUIRE(d > 0 && d % 4 == 0, "lina_greedy_pick_embed: d=%d must be a positive multiple of 4", d);
    LINA_REQUIRE(valid_dtype(dtype), "lina_greedy_pick_embed: bad dtype %d", dtype);
    LINA_REQUIRE(!x_out_packed || d % (dtype == LINA_BF16 ? 32 : 16) == 0, "lina_greedy_pick_embed: packed copy needs whole k-steps");
    dim3 grid((unsigned)B);
    if (dtype == LINA_F32)
        LINA_LAUNCH((greedy_pick_embed_kernel<float>), grid, dim3(256), 0, stream, (const float*)logits, row_stride,
                    (const float*)table, (float*)x_out, tok_log, step, counter, Q, L, n_emb, d, max_steps,
                    (float*)x_out_packed, loop_ctl);
    else
        LINA_LAUNCH((greedy_pick_embed_kernel<bf16_t>), grid, dim3(256), 0, stream, (const bf16_t*)logits, row_stride,
                    (const bf16_t*)table, (bf16_t*)x_out, tok_log, step, counter, Q, L, n_emb, d, max_steps,
                    (bf16_t*)x_out_packed, loop_ctl);
    return check_launch("lina_greedy_pick_embed");
}

extern "C" int lina_sample_pick_embed(const void* logits, int64_t row_stride, const void* table, void* x_out,
                                      void* x_out_packed, int64_t* tok_log, int64_t* step, int* counter, int* loop_ctl, int B,
                                      int Q, int L, int n_emb, int d, int max_steps, int n_sampled, int k, float temp,
                                      uint64_t seed, int dtype, lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(logits && table && x_out && tok_log && step && counter, "lina_sample_pick_embed: null pointer");
    LINA_REQUIRE(B > 0 && Q > 0 && Q <= 16 && L > 0 && n_emb > 0 && max_steps > 0,
                 "lina_sample_pick_embed: B,Q (<= 16),L,n_emb,max_steps must be positive");
    LINA_REQUIRE(d > 0 && d % 4 == 0, "lina_sample_pick_embed: d=%d must be a positive multiple of 4", d);
    LINA_REQUIRE(n_sampled >= 0 && n_sampled <= Q, "lina_sample_pick_embed: n_sampled must be in [0, Q]");
    LINA_REQUIRE(k >= 1 && temp > 0.0f, "lina_sample_pick_embed: k must be >= 1 and temp positive");
    LINA_REQUIRE(valid_dtype(dtype), "lina_sample_pick_embed: bad dtype %d", dtype);
    LINA_REQUIRE(!x_out_packed || d % (dtype == LINA_BF16 ? 32 : 16) == 0, "lina_sample_pick_embed: packed copy needs whole k-steps");
    if (L > kSampleMaxN) return fail(LINA_ERR_UNSUPPORTED, "lina_sample_pick_embed: L=%d exceeds %d", L, kSampleMaxN);
    dim3 grid((unsigned)B);
    const size_t smem = n_sampled > 0 ? (size_t)L * sizeof(float) : 0;
    if (dtype == LINA_F32)
        LINA_LAUNCH((sample_pick_embed_kernel<float>), grid, dim3(256), smem, stream, (const float*)logits, row_stride,
                    (const float*)table, (float*)x_out, tok_log, step, counter, Q, L, n_emb, d, max_steps,
                    (float*)x_out_packed, n_sampled, k, 1.0f / temp, seed, loop_ctl);
    else
        LINA_LAUNCH((sample_pick_embed_kernel<bf16_t>), grid, dim3(256), smem, stream, (const bf16_t*)logits, row_stride,
                    (const bf16_t*)table, (bf16_t*)x_out, tok_log, step, counter, Q, L, n_emb, d, max_steps,
                    (bf16_t*)x_out_packed, n_sampled, k, 1.0f / temp, seed, loop_ctl);
    return check_launch("lina_sample_pick_embed");
}

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
