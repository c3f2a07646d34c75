/* lina_gla.h -- C ABI of the MI355X-native Lina-Speech codec-token generation path.
 *
 * Drop-in boundary.  The reference has no FFI of its own: its
 * hot path is reached through Python operator names imported from the external
 * `fla` package (reference model/gla.py:19-23).  Each entry point below replaces
 * one of those operators (or one in-tree torch-eager stretch of the decode loop)
 * and is what a ctypes stub on the reference side binds.
 *
 * Conventions (all entry points):
 *   - raw DEVICE pointers + explicit shapes/strides (in ELEMENTS) + dtype enum;
 *     no torch types; the caller owns every buffer; nothing is allocated;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*) and never
 *     synchronised -> every call is hipGraph-capturable and re-entrant;
 *   - return 0 on success, a negative LINA_ERR_* on argument errors (nothing was
 *     launched) or when the launch itself failed; lina_last_error() then holds a
 *     thread-local message.  Never throws.
 *   - recurrent state is ALWAYS fp32 [B,H,Dk,Dv] contiguous (the reference keeps
 *     it in the model dtype between decode steps; fp32 is >= that).
 *   - "model dtype" tensors (q,k,v,o,x,y,weights,conv caches) are LINA_F32 or
 *     LINA_BF16; the innermost dimension is always contiguous.
 */
#ifndef LINA_GLA_H
#define LINA_GLA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* lina_stream_t; /* hipStream_t */

enum { LINA_F32 = 0, LINA_BF16 = 1 };
enum {
    LINA_OK = 0,
    LINA_ERR_ARG = -1,         /* bad shape / null pointer / bad enum                  */
    LINA_ERR_UNSUPPORTED = -2, /* legal request this build has no kernel for           */
    LINA_ERR_LAUNCH = -3       /* hipLaunchKernel reported an error                    */
};

/* ABI version (major*10000 + minor*100 + patch). */
int lina_version(void);
/* Thread-local text of the last error returned on this thread ("" if none). */
const char* lina_last_error(void);

/* Strides of one head-first view [B,H,T,D] (D contiguous), in elements.
 * A `rearrange(x,'b l (h d) -> b h l d')` view of [B,L,H*D] is (L*H*D, D, H*D). */
typedef struct lina_bht_strides {
    int64_t b, h, t;
} lina_bht_strides;

/* K1 -- GLA recurrence, T sequential steps (T = 1 is the decode step).
 *   S_t = diag(exp(gk_t)) S_{t-1} + k_t^T v_t ;  o_t = scale * q_t S_t     (fp32 math)
 * Replaces fla.ops.gla.fused_recurrent_gla and fla.ops.gla.naive.naive_recurrent_gla
 * (reference model/gla.py:188,190,197,201).
 *   q,k,gk: [B,H,T,Dk]   v,o: [B,H,T,Dv]   (views described by the stride structs)
 *   h0: initial state or NULL (= zeros);  ht: final state or NULL (not written).
 *   ht may alias h0 (in-place decode update: the state is read once, written once).
 *   dtype: dtype of q,k,v,o;  g_dtype: dtype of gk.
 *   Dk in {64,128,256}, Dv a multiple of 64. */
int lina_gla_recurrent_fwd(const void* q, const void* k, const void* v, const void* gk, void* o,
                           const float* h0, float* ht,
                           int B, int H, int T, int Dk, int Dv,
                           lina_bht_strides sq, lina_bht_strides sk, lina_bht_strides sv,
                           lina_bht_strides sg, lina_bht_strides so,
                           int dtype, int g_dtype, float scale, lina_stream_t stream);

/* K2 -- the same recurrence evaluated chunk-wise on the matrix cores (state tile kept in
 * MFMA accumulators, no per-chunk state in HBM).  Same contract as K1.
 * Replaces fla.ops.gla.chunk_gla and fla.ops.gla.fused_chunk_gla
 * (reference model/gla.py:193,195; 'fused_chunk' is the mixer's default mode, :48). */
int lina_gla_chunk_fwd(const void* q, const void* k, const void* v, const void* gk, void* o,
                       const float* h0, float* ht,
                       int B, int H, int T, int Dk, int Dv,
                       lina_bht_strides sq, lina_bht_strides sk, lina_bht_strides sv,
                       lina_bht_strides sg, lina_bht_strides so,
                       int dtype, int g_dtype, float scale, lina_stream_t stream);

/* K2 for SMALL B*H (training micro-batches): the same forward with the sequence cut into `nseg` segments that
 * run concurrently (state-only pass, elementwise combine of the segment states, full pass) -- exact, nseg x the
 * workgroups of lina_gla_chunk_fwd.  bf16 tensors and gates, Dk = Dv = 256, 16-byte aligned rows only
 * (LINA_ERR_UNSUPPORTED otherwise: call lina_gla_chunk_fwd).
 *   workspace: fp32 scratch of lina_gla_chunk_fwd_seg_workspace(B, H, Dk, Dv, nseg) BYTES; after the call its head holds
 *              the state at the start of every segment ([B*H*Dk/256][segments][256][Dk]), which lina_gla_chunk_bwd_full
 *              accepts as seg_states. */
int64_t lina_gla_chunk_fwd_seg_workspace(int B, int H, int Dk, int Dv, int nseg);
int lina_gla_chunk_fwd_seg(const void* q, const void* k, const void* v, const void* gk, void* o,
                           const float* h0, float* ht, float* workspace, int nseg,
                           int B, int H, int T, int Dk, int Dv,
                           lina_bht_strides sq, lina_bht_strides sk, lina_bht_strides sv,
                           lina_bht_strides sg, lina_bht_strides so,
                           int dtype, int g_dtype, float scale, lina_stream_t stream);

/* K2b -- backward of K2: given d_o = dL/do (and optionally
 * dht = dL/d final_state) produce dq, dk, dv, dg and optionally dh0 = dL/d initial_state.
 * Replaces the autograd backward of fla.ops.gla.chunk_gla / fused_chunk_gla that the reference
 * reaches through loss.backward() (train_lina.py:88-94 over the call sites model/gla.py:193,195).
 *   q,k,v,gk,d_o and the outputs dq,dk,dv,dg are head-first [B,H,T,D] views with explicit strides;
 *   h0, dht, dh0: fp32 [B,H,Dk,Dv] contiguous or NULL;
 *   dg_tail: fp32 [B,H,Dk] or NULL -- sum_v final_state (.) dht, the gate gradient that enters
 *            through the final state (the caller holds final_state; NULL when dht is NULL);
 *   workspace: fp32 scratch of lina_gla_chunk_bwd_workspace(...) BYTES (dq, dk stay fp32 there until
 *            dg = reverse-cumsum(q dq - k dk) has been formed).  Nothing is allocated inside. */
int64_t lina_gla_chunk_bwd_workspace(int B, int H, int T, int Dk, int Dv);
int lina_gla_chunk_bwd(const void* q, const void* k, const void* v, const void* gk, const void* d_o,
                       const float* h0, const float* dht, const float* dg_tail,
                       void* dq, void* dk, void* dv, void* dg, float* dh0, float* workspace,
                       int B, int H, int T, int Dk, int Dv,
                       lina_bht_strides sq, lina_bht_strides sk, lina_bht_strides sv,
                       lina_bht_strides sg, lina_bht_strides sdo,
                       lina_bht_strides sdq, lina_bht_strides sdk, lina_bht_strides sdv, lina_bht_strides sdg,
                       int dtype, int g_dtype, float scale, lina_stream_t stream);

/* K2b on the full-head kernel (bf16 tensors and gates, Dk = Dv in {64,128,256}, heads adjacent in memory for Dk < 256):
 * the same contract and results as lina_gla_chunk_bwd, computed as three sweeps of K2's own kernel body
 * (reverse key-gated sweep -> dv, dS; value-gated forward sweep -> dq; value-gated reverse sweep -> dk and, from its own q
 * rows, the dq just written and k, dg), each sweep on
 * all of nseg sequence segments concurrently from boundary states when nseg > 1 (small B*H).  Returns
 * LINA_ERR_UNSUPPORTED for layouts the kernel does not take (call lina_gla_chunk_bwd then).
 *   workspace: lina_gla_chunk_bwd_full_workspace(...) BYTES of fp32 scratch (boundary states of the segments);
 *   seg_states: NULL, or the segment start states that lina_gla_chunk_fwd_seg left at the head of ITS workspace for the
 *               same inputs, T and nseg (the forward of this backward): the state-only forward pass is then skipped. */
int64_t lina_gla_chunk_bwd_full_workspace(int B, int H, int T, int Dk, int Dv, int nseg);
int lina_gla_chunk_bwd_full(const void* q, const void* k, const void* v, const void* gk, const void* d_o,
                            const float* h0, const float* dht, const float* dg_tail,
                            void* dq, void* dk, void* dv, void* dg, float* dh0, float* workspace,
                            const float* seg_states, int nseg, int B, int H, int T, int Dk, int Dv,
                            lina_bht_strides sq, lina_bht_strides sk, lina_bht_strides sv,
                            lina_bht_strides sg, lina_bht_strides sdo,
                            lina_bht_strides sdq, lina_bht_strides sdk, lina_bht_strides sdv, lina_bht_strides sdg,
                            int dtype, int g_dtype, float scale, lina_stream_t stream);

/* K3 -- causal depthwise short convolution (+ optional SiLU), prefill form.
 * Replaces fla.modules.ShortConvolution.forward for T > 1 or cache == NULL
 * (ctor reference model/gla.py:106-108, calls :161-163).
 *   x,y: [B,T,D] with batch/time strides (D contiguous); w: [D,W]; bias: [D] or NULL;
 *   mask: fp32 [B,T] multiplied into x, or NULL;
 *   cache: [B,D,W] (model dtype) or NULL -- receives the last W (masked) inputs,
 *   left zero-padded when T < W.  W <= 8.  activation: 0 none, 1 SiLU. */
int lina_short_conv_fwd(const void* x, const void* w, const void* bias, const float* mask,
                        void* cache, void* y, int B, int T, int D, int W,
                        int64_t x_sb, int64_t x_st, int64_t y_sb, int64_t y_st,
                        int activation, int dtype, lina_stream_t stream);

/* K4 -- single-token step of the same convolution: cache <- roll(cache,-1);
 * cache[..,-1] <- x; y = act(sum_j cache[..,j] w[..,j] + bias).  x,y: [B,D] with row strides.
 * Replaces ShortConvolution.forward when T == 1 and a cache is given. */
int lina_short_conv_step(const void* x, const void* w, const void* bias, void* cache, void* y,
                         int B, int D, int W, int64_t x_sb, int64_t y_sb,
                         int activation, int dtype, lina_stream_t stream);

/* K3b -- backward of K3 (cache == NULL): dx [B,T,D] (strided like x) and fp32 partial sums of the
 * weight / bias gradient, dwb_partial [B * ceil(T / LINA_CONV_BWD_TT)][D][W+1] (slot W = bias), which
 * the caller sums over dim 0.  Replaces the autograd backward of ShortConvolution.forward
 * (reference model/gla.py:161-163 under loss.backward()). */
#ifndef LINA_CONV_BWD_TT                /* (A/B builds of the library override it) */
#define LINA_CONV_BWD_TT 64
#endif
int lina_short_conv_bwd(const void* x, const void* w, const void* bias, const float* mask, const void* dy,
                        void* dx, float* dwb_partial, int B, int T, int D, int W,
                        int64_t x_sb, int64_t x_st, int64_t dy_sb, int64_t dy_st, int64_t dx_sb, int64_t dx_st,
                        int activation, int dtype, lina_stream_t stream);

/* K5 -- y = x * rsqrt(mean(x^2) + eps) * w  [ * g * sigmoid(g) ]   over the last dim D.
 * g == NULL -> plain RMSNorm.  w == NULL -> no affine.
 * Rows are addressed two-level: row r -> (ro, ri) = (r / rows_inner, r % rows_inner) and the row
 * starts at  ro * outer + ri * inner  (elements) of x / g / y -- e.g. rows = B*H, rows_inner = H
 * reads the gate straight out of a wider fused-projection row.  rows_inner = 1 for flat rows.
 * `n_partial` > 1: x holds n_partial fp32 partial sums per row, `x_part_stride` apart, which are
 * added first (pass 1 / 0 otherwise).  x_dtype: dtype of x; dtype: dtype of g, w, y.
 * Replaces fla.modules.FusedRMSNormSwishGate / fla.modules.RMSNorm
 * (reference model/gla.py:111,115,219,222). */
int lina_rmsnorm_gate_fwd(const void* x, const void* g, const void* w, void* y,
                          int64_t rows, int rows_inner, int D,
                          int64_t x_outer, int64_t x_inner, int64_t g_outer, int64_t g_inner,
                          int64_t y_outer, int64_t y_inner,
                          int n_partial, int64_t x_part_stride,
                          float eps, int x_dtype, int dtype, lina_stream_t stream);

/* K5b -- backward of K5 over contiguous rows x, dy, dx [rows][D], all tensors of `dtype` (training path).
 * Replaces the autograd backward of FusedRMSNormSwishGate / RMSNorm (reference model/gla.py:219,222).
 *   g, dg: both given (swish gate) or both NULL; w may be NULL;
 *   row r of g / dg sits at (r / rows_inner) * outer + (r % rows_inner) * inner elements (strides multiples of 4): the
 *   gate is read from, and its gradient written into, head slices of wider rows in place (the stacked projection's
 *   output and the gradient slab of that projection);
 *   dw_partial: fp32 [lina_rmsnorm_gate_bwd_partials(rows)][D], summed over dim 0 by the caller. */
#ifndef LINA_NORM_BWD_MAX_WG            /* (A/B builds of the library override it; callers size by the function below) */
#define LINA_NORM_BWD_MAX_WG 1024
#endif
int lina_rmsnorm_gate_bwd_partials(int64_t rows);
int lina_rmsnorm_gate_bwd(const void* x, const void* g, const void* w, const void* dy, void* dx, void* dg,
                          float* dw_partial, int64_t rows, int rows_inner, int D, int64_t g_outer, int64_t g_inner,
                          int64_t dg_outer, int64_t dg_inner, float eps, int dtype, lina_stream_t stream);

/* K6a -- codec-token embedding gather-sum: out[n,:] = sum_q table[q, idx[q,n], :].
 * Replaces MultiEmbedding.forward + reduce('q b n d -> b n d','sum')
 * (reference model/multiembed.py:21-23, model/modeling_lina.py:131,178-179).
 *   idx: int64 [Q,N];  table: [Q,n_emb,d];  out: [N,d]. */
int lina_embed_sum(const int64_t* idx, const void* table, void* out,
                   int Q, int64_t N, int n_emb, int d, int dtype, lina_stream_t stream);

/* K6d -- the token epilogue of a GREEDY decode step in one launch (one workgroup per batch row): arg-max of each of the
 * Q quantizers' logits (lowest index on ties, as K6b), tok_log[step[0]][q][b] = pick (int64 [max_steps][Q][B]; skipped
 * when step[0] >= max_steps), x_out[b,:] = sum_q table[q, pick_q, :] (K6a), and step[0] += 1 by the last workgroup to
 * finish.  logits: [B, Q*L] with a row stride; counter: one int32, zero before the first call (left zero).
 * x_out_packed (optional): the same rows in the fragment-major layout of the packed projections (see below).
 * loop_ctl (optional, int32 [LINA_LOOP_CTL_ROWS + B], zero with word [1] = -1 before the loop): the reference's stop
 * bookkeeping (model/modeling_lina.py:126,168-173) kept on the device --
 *     word [0] = rows that have emitted the stop token (id 2 on every quantizer) at some step so far,
 *     word [1] = the first step at which ALL rows had (the step the reference breaks at), -1 until then,
 *     words [2],[3] = a per-call seed word (lo, hi) XORed into `seed` of lina_sample_pick_embed (a captured launch is
 *                     re-seeded by writing it),   words [4 + b] = row b has stopped.
 * The host reads word [1] whenever it likes (every 16 steps in LinaModel.generate_batch) instead of syncing per step.
 * Replaces reference model/modeling_lina.py:159-179 (k = 1 picks, token list append, stop flags, next-input embedding). */
#define LINA_LOOP_CTL_ROWS 4
int lina_greedy_pick_embed(const void* logits, int64_t row_stride, const void* table, void* x_out,
                           void* x_out_packed, int64_t* tok_log, int64_t* step, int* counter, int* loop_ctl, int B, int Q,
                           int L, int n_emb, int d, int max_steps, int dtype, lina_stream_t stream);

/* K6e -- K6d for the reference's DEFAULT generation mode (model/modeling_lina.py:119-121,159-164, tools.py:38-44):
 * quantizers q < n_sampled are SAMPLED (top-k / temperature, K6c) and the others take the arg-max (K6b); everything else
 * as lina_greedy_pick_embed.  The uniform number of (row b, quantizer q) is the one lina_topk_sample_rows hashes for row
 * b*Q + q of a [B*Q]-row call at the same (seed, step[0]): the tokens equal those of the separate launches. */
int lina_sample_pick_embed(const void* logits, int64_t row_stride, const void* table, void* x_out,
                           void* x_out_packed, int64_t* tok_log, int64_t* step, int* counter, int* loop_ctl, int B, int Q,
                           int L, int n_emb, int d, int max_steps, int n_sampled, int k, float temp, uint64_t seed,
                           int dtype, lina_stream_t stream);

/* K6b -- greedy pick: out[r] = argmax_j logits[r,j], lowest index on exact ties.
 * Replaces topk_sampling(k=1) (reference model/tools.py:38-44, modeling_lina.py:159-164);
 * identical except on exact ties, where the reference draws uniformly among them. */
int lina_argmax_rows(const void* logits, int64_t* out, int64_t rows, int n, int64_t row_stride,
                     int dtype, lina_stream_t stream);

/* K6c -- top-k / temperature sampling, one token per logits row, on the device.
 * Replaces topk_sampling(seq, k, temp) for k > 1 (reference model/tools.py:38-44, called from
 * model/modeling_lina.py:159-164): keep_j = (x_j / temp >= k-th largest x), p = softmax of the kept
 * x_j / temp, token = inverse CDF (index order) of ONE uniform number per row.
 *   u_ext: fp32 [rows] uniforms in [0,1) supplied by the caller, or NULL: then the kernel hashes
 *          (seed, step[0], row) -- `step` is a DEVICE int64 the caller advances, so a replayed graph
 *          draws fresh numbers (NULL step = 0).  n <= 8192.  out: int64 [rows]. */
int lina_topk_sample_rows(const void* logits, int64_t* out, int64_t rows, int n, int64_t row_stride,
                          int k, float temp, const float* u_ext, uint64_t seed, const int64_t* step,
                          int dtype, lina_stream_t stream);

/* K4x3 + K7 -- decode-step prologue of one GLA mixer (reference model/gla.py:158-163,174-180
 * at T = 1): three conv steps on the q/k/v slices of the fused projection row `z`, and the
 * gate  gk = logsigmoid(W2 * z_lowrank + b2) / normalizer  (optionally clamped from below).
 *   z: [B, ldz] model dtype; q_pre at column off_q (Kd wide), k_pre at off_k (Kd),
 *      v_pre at off_v (Vd), gate low-rank activations at off_lr (R wide, R <= 32).
 *   wq,wk: [Kd,W]; wv: [Vd,W];  cq,ck: [B,Kd,W]; cv: [B,Vd,W] (updated in place).
 *   w2: [Kd,R]; b2: [Kd] (model dtype).
 *   qkv: [B, 2*Kd+Vd] model dtype, receives silu(conv(q)) | silu(conv(k)) | silu(conv(v));
 *   gk: fp32 [B,Kd].   clamp_min: NaN = no clamp.  W == 4 only. */
int lina_gla_decode_prologue(const void* z, int64_t ldz, int off_q, int off_k, int off_v, int off_lr,
                             const void* wq, const void* wk, const void* wv,
                             void* cq, void* ck, void* cv,
                             const void* w2, const void* b2,
                             void* qkv, float* gk,
                             int B, int Kd, int Vd, int W, int R,
                             float normalizer, float clamp_min, int dtype, lina_stream_t stream);

/* SwiGLU gate of the channel mixer (reference model/base_blocks.py:48-50):
 *   y[r, j] = silu(u[r, j]) * u[r, Hd + j]   for j < Hd;   y[r, Hd] = 1 if ld_y > Hd (bias
 *   column of a K-padded down-projection), y[r, Hd+1 ..] = 0.   u: [rows, 2*Hd] (+ row stride). */
int lina_swiglu(const void* u, void* y, int64_t rows, int Hd, int64_t ld_u, int64_t ld_y,
                int dtype, lina_stream_t stream);

/* K11b -- backward of the SwiGLU gate:  du[r, j] = ds[r, j] u[r, Hd + j] sig(a)(1 + a(1 - sig(a))),  a = u[r, j];
 *   du[r, Hd + j] = ds[r, j] silu(a).   ds: [rows, Hd], u / du: [rows, 2 Hd] (row strides in elements, multiples of 4).
 * Replaces autograd through `F.silu(a) * b` (reference model/base_blocks.py:48-50) in the training step. */
int lina_swiglu_bwd(const void* ds, const void* u, void* du, int64_t rows, int Hd, int64_t ld_u, int64_t ld_ds,
                    int64_t ld_du, int dtype, lina_stream_t stream);

/* K12 -- the mixer's gate for a whole sequence (reference model/gla.py:174-180): out = logsigmoid(x) / normalizer, clamped
 * from below when clamp_min is not NaN (dy == NULL); with dy: out = dy (1 - sigmoid(x)) / normalizer, 0 where the clamp is
 * active -- the gradient w.r.t. x.  n elements (multiple of 4), contiguous. */
int lina_gate_logsigmoid(const void* x, const void* dy, void* out, int64_t n, float normalizer, float clamp_min, int dtype,
                         lina_stream_t stream);

/* K14 -- cross-entropy over the rows of logits [N, V] (row stride ld, any V >= 4 up to 8445) against int64 targets,
 * fp32 arithmetic (reference modeling_lina.py:106 `F.cross_entropy(flat_logits, flat_target, ignore_index=1)`):
 *   forward  (loss_row given, dlogits NULL): lse[r] = logsumexp(logits[r]); loss_row[r] = lse[r] - logits[r, target[r]],
 *            0 for rows whose target is ignore_index (the caller forms sum(loss_row) / count);
 *   backward (dlogits given, loss_row NULL): dlogits[r] = (softmax(logits[r]) - onehot(target[r])) * scale[0] for rows
 *            that count, 0 for ignored rows (row stride ld_d); scale is a DEVICE scalar (d loss / count). */
int lina_cross_entropy(const void* logits, const int64_t* target, float* lse, float* loss_row, const float* scale,
                       void* dlogits, int64_t N, int V, int64_t ld, int64_t ld_d, int64_t ignore_index, int dtype,
                       lina_stream_t stream);

/* K13a -- per-slab column sums of x [M, N] (row stride ld; N, ld multiples of 4): partial fp32
 * [lina_swiglu_bwd_partials(M)][N], summed over dim 0 by lina_sum_partials.  The bias gradient of a projection
 * (`grad_output.sum(0)` of nn.Linear's autograd) as a deterministic two-level sum without global semaphores. */
int lina_colsum(const void* x, float* partial, int64_t M, int N, int64_t ld, int dtype, lina_stream_t stream);

/* K13 -- second level of the parameter-gradient sums: out[o][n] = sum_p part[o][p][n], part fp32 [outer][P][N] (the
 * `*_partial` outputs of K3b / K5b / K10b / K11c / K12b), out [outer][N] of out_dtype; N a multiple of 4. */
int lina_sum_partials(const float* part, void* out, int outer, int P, int64_t N, int out_dtype, lina_stream_t stream);

/* K15 -- the channel mixer's GEMM operands from its fp32 master weights, ONE pass (the cast torch.autocast makes of every
 * parameter each step -- reference train_lina.py:72-120 -- written straight into the padded layout of the train path:
 * reference model/base_blocks.py:42-50 `p_in` / `p_out`):
 *   Wi [2][Hp][d_in]: rows [0, H) of half s = w_in rows [s H, s H + H), the rest 0;   bi [2][Hp]: b_in likewise, and with b_out
 *   given (32, 1/32) at row H (the gate's column H is then exactly 1);   Wo [d_out][Hq]: w_out in columns [0, H), b_out in
 *   column H, the rest 0 (Hq >= Hp: the row length the down-projection's dX GEMM wants; its other uses take the first Hp
 *   columns).   w_in fp32 [2 H][d_in], b_in fp32 [2 H] or NULL, w_out fp32 [d_out][H], b_out fp32 [d_out] or NULL;
 *   Hp > H; Hp, Hq and d_in multiples of 4; the three outputs are of out_dtype. */
int lina_mlp_pack(const float* w_in, const float* b_in, const float* w_out, const float* b_out, void* Wi, void* bi, void* Wo,
                  int H, int Hp, int Hq, int d_in, int d_out, int out_dtype, lina_stream_t stream);

/* K16 -- row blocks stacked into one GEMM operand, ONE pass: out [total_rows][cols] of out_dtype = the n_src <= 8 fp32 blocks
 * srcs[i] [rows[i]][cols] one under the other, zero rows after them (the q | k | v | g | low-rank projections of reference
 * model/gla.py:158-160,216 as one weight; replaces torch.cat + the autocast cast).  srcs / rows are HOST arrays; cols a
 * multiple of 4, every block 16-byte aligned. */
int lina_stack_rows(const float* const* srcs, const int* rows, int n_src, int cols, int total_rows, void* out, int out_dtype,
                    lina_stream_t stream);

/* K17 -- AdamW (decoupled weight decay) over n_tensors fp32 tensors, 48 per launch (reference train_lina.py:104-118:
 * torch.optim.AdamW; the operation order of torch's `_fused_adamw_` in fp32):
 *   p -= lr wd p;  m += (1 - beta1)(g - m);  v = beta2 v + (1 - beta2) g g;  p -= (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps)
 * params / grads / exp_avg / exp_avg_sq / numel are HOST arrays of n_tensors device pointers / element counts;
 * bias_correction1 = 1 - beta1^t, bias_correction2 = 1 - beta2^t for the step t >= 1 being taken. */
int lina_adamw_multi(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                     const int64_t* numel, int n_tensors, double lr, double beta1, double beta2, double eps, double weight_decay,
                     double bias_correction1, double bias_correction2, lina_stream_t stream);
int lina_adamw_multi_max(void);   /* tensors per launch */

/* K11c -- K11b that also leaves the column sums of du (the bias gradient of the up-projection):
 *   colsum_partial fp32 [lina_swiglu_bwd_partials(rows)][2 Hd], summed over dim 0 by the caller (sums of the values as
 *   stored in `dtype`).  Hd and the row strides must be multiples of 4. */
#define LINA_SWIGLU_COLSUM_ROWS 128
int lina_swiglu_bwd_partials(int64_t rows);
int lina_swiglu_bwd_colsum(const void* ds, const void* u, void* du, float* colsum_partial, int64_t rows, int Hd,
                           int64_t ld_u, int64_t ld_ds, int64_t ld_du, int dtype, lina_stream_t stream);

/* K12b -- gate projection + gate in one pass (reference model/gla.py:107-109 `gk_proj[1]`, :174-180): with lr [rows, L]
 * (row stride lr_stride, L <= 16), w fp32 [C, L], b fp32 [C] or NULL (rounded to `dtype` in the kernel, as autocast would):
 *   dy == NULL:  out [rows, C] = logsigmoid(lr w^T + b) / normalizer, clamped from below unless clamp_min is NaN;
 *   dy given:    out [rows, C] = d(pre) = dy (1 - sigmoid(pre)) / normalizer (0 where clamped), and
 *                dwb_partial fp32 [lina_gate_lowrank_partials(rows)][C][L + 1] = per-workgroup sums of d(pre)^T [lr | 1]
 *                (slot L = bias gradient), summed over dim 0 by the caller.  d(lr) = d(pre) w is the caller's GEMM.
 * bf16 with L == 16, C % 64 == 0 and 16-byte aligned rows (the mixer's layout) takes K12c: both rank-16 contractions on the
 * matrix core; same contract, sums in a different order. */
#define LINA_GATE_LOWRANK_ROWS 128
int lina_gate_lowrank_partials(int64_t rows);
int lina_gate_lowrank(const void* lr, int64_t lr_stride, const float* w, const float* b, const void* dy, void* out,
                      float* dwb_partial, int64_t rows, int C, int L, float normalizer, float clamp_min, int dtype,
                      lina_stream_t stream);

/* K10 -- LayerNorm over the last dimension with the residual add that precedes it in a pre-norm block
 * (reference model/base_blocks.py:65-69: `x = tmix(norm1(x)) + x; x = cmix(norm2(x)) + x`), for the TRAINING step:
 *   forward:  x' = x + r (r optional; x' written to xsum when given);  y = (x' - mean) rstd gamma + beta;  mean / rstd
 *             (fp32 [N]) are kept for the backward.   x, xsum, dx, dpass: x_dtype;  r, dr: r_dtype;  y, dy: y_dtype.
 *   backward: dx = dpass + rstd (g - mean(g) - xhat mean(g xhat)),  g = dy gamma  (dpass optional: the gradient reaching x'
 *             through the residual path);  dr (optional) = dx in r_dtype;  dgamma_part / dbeta_part: fp32
 *             [lina_layernorm_bwd_partials(N)][D] per-workgroup sums, added up by the caller (deterministic, no atomics).
 * [N, D] contiguous, D % 4 == 0, D <= 2048.  Built dtype triples (x, r, y): (f32,f32,f32), (f32,bf16,bf16), (f32,f32,bf16),
 * (bf16,bf16,bf16).  Replaces nn.LayerNorm + the elementwise add / cast passes around it under bf16 autocast. */
int lina_layernorm_fwd(const void* x, const void* r, const float* gamma, const float* beta, void* xsum, void* y,
                       float* mean, float* rstd, int64_t N, int D, float eps, int x_dtype, int r_dtype, int y_dtype,
                       lina_stream_t stream);
int lina_layernorm_bwd_partials(int64_t rows);
int lina_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                       const void* dpass, void* dx, void* dr, float* dgamma_part, float* dbeta_part, int64_t N, int D,
                       int x_dtype, int r_dtype, int y_dtype, lina_stream_t stream);

/* K1d -- decode-step (T = 1) state update, row-split: same arithmetic as K1, but each workgroup
 * streams a contiguous 64-row block of the fp32 state (in place) and the q.S products of the Dk/64
 * row blocks are returned as fp32 PARTIALS  o_part[Dk/64][B*H][Dv]  for K5 to add (n_partial).
 * q,k,gk: [B,H,Dk], v: [B,H,Dv] addressed by (batch, head) strides.  Dv in {64,128,256}, Dk % 64 == 0.
 * Same reference call sites as K1 (model/gla.py:188-201 at T = 1). */
int lina_gla_decode_update(const void* q, const void* k, const void* v, const void* gk,
                           float* o_part, float* state, int B, int H, int Dk, int Dv,
                           int64_t q_sb, int64_t q_sh, int64_t k_sb, int64_t k_sh,
                           int64_t v_sb, int64_t v_sh, int64_t g_sb, int64_t g_sh,
                           int dtype, int g_dtype, float scale, lina_stream_t stream);

/* K1d + K5 in one launch: as lina_gla_decode_update, and the LAST of a head's Dk/64 row-block workgroups
 * adds the partials, RMS-normalises over Dv, applies `norm_weight` and the swish gate and writes
 * og [B,H,Dv] (model dtype).  gate: [B,H,Dv] addressed by (batch, head) strides.  counters: int32 [B*H],
 * zero before the first call (each launch leaves them zero).  The inter-workgroup hand-off uses 8-byte
 * agent-scope atomics on both sides and never spins (programming guide G16); results are bit-identical to
 * lina_gla_decode_update followed by lina_rmsnorm_gate_fwd(n_partial = Dk/64).
 * Replaces reference model/gla.py:186-220 at T = 1. */
int lina_gla_decode_update_norm(const void* q, const void* k, const void* v, const void* gk,
                                float* o_part, float* state, const void* gate, const void* norm_weight,
                                void* og, int* counters, int B, int H, int Dk, int Dv,
                                int64_t q_sb, int64_t q_sh, int64_t k_sb, int64_t k_sh,
                                int64_t v_sb, int64_t v_sh, int64_t g_sb, int64_t g_sh,
                                int64_t gate_sb, int64_t gate_sh, float eps,
                                int dtype, int g_dtype, float scale, lina_stream_t stream);

/* Fragment-major ("packed") operands of the decode-step projections.  A row-major MFMA operand costs sixteen cache
 * lines per quarter wave (16 rows x 64 B per load instruction): ~30 GB/s per CU.  Packed, the 16 rows x KSTEP columns of
 * one fragment (KSTEP = 32 bf16 / 16 fp32, KL = 8 / 4 elements per lane) are one contiguous 1 KiB block:
 *     element (m, k) -> ((m/16 * K/KSTEP + k/KSTEP) * 64 + m%16 + 16*((k%KSTEP)/KL)) * KL + k%KL
 * with the rows padded to a multiple of 64 (zero rows).  Weights are packed once; activations are written packed by the
 * producing epilogue (out_packed below, og_packed of lina_gla_decode_window, x_out_packed of lina_greedy_pick_embed,
 * lina_weighted_rows_add_packed).
 *   lina_linear_skinny_ex = lina_linear_skinny with  in_packed != 0: A and W are packed (lda / ldw ignored; for SwiGLU
 *   the two weight halves are packed separately, w_half_rows = padded rows of one half, else the padded row count);
 *   out (row-major, may be NULL) and / or out_packed (packed copy, width out_packed_width >= N, whole k-steps);
 *   resid == out_packed (the same pointer): the residual is read from the packed buffer (in-place update of a
 *   residual stream that exists only in packed form).
 *   lina_gla_decode_inproj_packed = lina_gla_decode_inproj with packed x and w_in.  Same arithmetic, bit-identical.
 *   in_packed bit 1 (value 2) / w_stream != 0: the WEIGHT fragments are loaded with the non-temporal hint -- a matrix
 *   that is streamed this way does not displace the operands meant to stay in the 256 MB Infinity Cache between two
 *   tokens (the decode engine streams the largest matrices and keeps the rest resident). */
int lina_linear_skinny_ex(const void* A, int64_t lda, const void* W, int64_t ldw, int in_packed, int w_half_rows,
                          const float* c1, const float* c2, const void* resid, int64_t ldr, void* out, int64_t ldo,
                          void* out_packed, int out_packed_width, int M, int N, int K, int swiglu_hidden, int ln_dim,
                          float ln_eps, int dtype, lina_stream_t stream);
int lina_gla_decode_inproj_packed(const void* x_packed, const void* w_in_packed, const float* c1, const float* c2,
                                  const void* wq, const void* wk, const void* wv, void* cq, void* ck, void* cv,
                                  const void* w2, const void* b2, void* qkv, void* g_out, float* gk, int B, int K,
                                  int Kd, int Vd, int W, int R, float ln_eps, float normalizer, float clamp_min,
                                  int w_stream, int dtype, lina_stream_t stream);
int lina_weighted_rows_add_packed(const void* attc, int Tp, const void* vv, void* x, void* x_packed, int B, int Tn, int d,
                                  int dtype, lina_stream_t stream);

/* K1w + K5 -- decode-step state update with a WINDOWED (lazily written) state; inputs / output og as
 * lina_gla_decode_update_norm (one workgroup per (b,h): no partial buffer, no counters), same reference lines
 * (model/gla.py:186-220 at T = 1), Dk in {64,128,256}, Dv in {64,128,256,512}; Dv = 512 (expand_v = 2) splits a head's
 * columns over two workgroups that meet in o_exchange (fp32 [B*H*Dv]) + counters (int32 [B*H], zero; left zero) for the
 * norm -- both may be NULL for Dv <= 256.  `state` is the state at the
 * START of the current window of `window` (1, 2, 4 or 8) steps: it is only READ on steps 0 .. window-2 and rewritten on
 * step window-1, the steps in between live in the history buffers
 *     hist_k, hist_c: fp32 [window][B*H][Dk]  (k_s and the cumulative log-gate c_s of step s of the window)
 *     hist_v:         fp32 [window][B*H][Dv]
 * (chunk algebra of SURVEY App. A.3; every exponent is a difference <= 0).  The window position is
 * (step[0] - origin[0]) mod window, read from DEVICE memory, so one captured graph serves all positions.
 * lina_gla_decode_window_flush applies the first n_pending history entries to `state` (call it before anybody else reads
 * the state, then restart the window: origin <- step).  HBM bytes per token and (row, head): 4 Dk Dv (1 + 1/window)
 * + history instead of 8 Dk Dv.  Returns the state of the immediate form up to fp32 rounding.
 * og_packed != 0: og is written fragment-major as the [B, H*Dv] A operand of lina_linear_skinny_ex (see there). */
int lina_gla_decode_window_max(void);
int lina_gla_decode_window(const void* q, const void* k, const void* v, const void* gk,
                           float* state, const void* gate, const void* norm_weight,
                           void* og, float* o_exchange, int* counters,
                           float* hist_k, float* hist_c, float* hist_v,
                           const int64_t* step, const int64_t* origin, int window,
                           int B, int H, int Dk, int Dv,
                           int64_t q_sb, int64_t q_sh, int64_t k_sb, int64_t k_sh,
                           int64_t v_sb, int64_t v_sh, int64_t g_sb, int64_t g_sh,
                           int64_t gate_sb, int64_t gate_sh, float eps, int og_packed,
                           int dtype, int g_dtype, float scale, lina_stream_t stream);
int lina_gla_decode_window_flush(float* state, const float* hist_k, const float* hist_c, const float* hist_v,
                                 int n_pending, int B, int H, int Dk, int Dv, lina_stream_t stream);
/* The same two entry points with the dtype of `state` as an argument (round 6, opt-in): LINA_F32 = the entries above;
 * LINA_BF16 = a bf16 state [B,H,Dk,Dv] -- what the reference keeps between the decode steps of a bf16 model
 * (model/gla.py:229-240: init_state allocates with param.new_zeros, Cache.update copy_-s the fp32 final state into it): read,
 * updated in fp32 registers, rounded (nearest-even) when written back -- every step at window 1 (the reference's arithmetic),
 * every window-th step otherwise.  bf16 state needs dtype == LINA_BF16; the window history stays fp32. */
int lina_gla_decode_window_s(const void* q, const void* k, const void* v, const void* gk,
                             void* state, int state_dtype, const void* gate, const void* norm_weight,
                             void* og, float* o_exchange, int* counters,
                             float* hist_k, float* hist_c, float* hist_v,
                             const int64_t* step, const int64_t* origin, int window,
                             int B, int H, int Dk, int Dv,
                             int64_t q_sb, int64_t q_sh, int64_t k_sb, int64_t k_sh,
                             int64_t v_sb, int64_t v_sh, int64_t g_sb, int64_t g_sh,
                             int64_t gate_sb, int64_t gate_sh, float eps, int og_packed,
                             int dtype, int g_dtype, float scale, lina_stream_t stream);
int lina_gla_decode_window_flush_s(void* state, int state_dtype, const float* hist_k, const float* hist_c,
                                   const float* hist_v, int n_pending, int B, int H, int Dk, int Dv, lina_stream_t stream);

/* Decode-step projection with fused neighbours: out[M,N] = epi(A[M,K] . W[N,K]^T), M ~ batch rows.
 *   ln_dim > 0 : A is layer-normalised over its ln_dim features first, folded algebraically:
 *                out = rstd*(A.W^T - mu*c1) + c2   with W pre-scaled by the LN gamma,
 *                c1[n] = sum_k W[n,k] (fp32), c2[n] = sum_k beta_k W0[n,k] + bias[n] (fp32);
 *   ln_dim == 0: out = A.W^T + c2 (c2 may be NULL);
 *   swiglu_hidden = Hd > 0: W holds 2*Hd rows (gate half, value half; c1,c2 2*Hd long); out column n
 *                < Hd is silu(gate_n)*value_n, column Hd is the constant 1, columns > Hd are 0;
 *   resid != NULL: out += resid (out may alias resid).
 * Replaces the nn.Linear / nn.LayerNorm / SwiGLU glue of the reference decode step
 * (model/gla.py:158-160,216,225; model/base_blocks.py:48-50,65-69; model/modeling_lina.py:155).
 * K must be a multiple of 32 (bf16) / 16 (f32); lda, ldw multiples of 8 / 4 elements. */
int lina_linear_skinny(const void* A, int64_t lda, const void* W, int64_t ldw,
                       const float* c1, const float* c2, const void* resid, int64_t ldr,
                       void* out, int64_t ldo, int M, int N, int K,
                       int swiglu_hidden, int ln_dim, float ln_eps, int dtype, lina_stream_t stream);

/* The whole input side of one GLA mixer at T = 1 in ONE launch: LayerNorm-1 (folded as in
 * lina_linear_skinny) -> fused projection -> q,k,v conv step (+SiLU, caches rolled in place) | g | gate
 * (rank-16 up-projection + bias + logsigmoid / normalizer [+clamp]).  Equivalent to
 * lina_linear_skinny(ln) + lina_gla_decode_prologue without the intermediate row.
 *   x: [B,K];  w_in: [2Kd+2Vd+R, K] rows q|k|v|g|low-rank, LayerNorm gamma folded in; c1,c2 fp32 (same length);
 *   outputs qkv [B,2Kd+Vd], g_out [B,Vd] (model dtype), gk fp32 [B,Kd].  W = 4, R = 16, Kd,Vd % 16 == 0.
 * Replaces reference model/base_blocks.py:66 (norm1) + model/gla.py:158-163,174-180,216 at T = 1. */
int lina_gla_decode_inproj(const void* x, int64_t ldx, const void* w_in, int64_t ldw,
                           const float* c1, const float* c2,
                           const void* wq, const void* wk, const void* wv,
                           void* cq, void* ck, void* cv, const void* w2, const void* b2,
                           void* qkv, void* g_out, float* gk,
                           int B, int K, int Kd, int Vd, int W, int R,
                           float ln_eps, float normalizer, float clamp_min, int dtype, lina_stream_t stream);

/* Blind cross-attention at T = 1, text side precomputed (reference model/crossatt.py:105-155).
 * step1: q = LayerNorm(q_lin[b]); att1[b] = softmax(q . kk[b]^T * scale); xp[b] = att1[b] . pe
 * step2: att2[b] = softmax(xp[b] . pe^T * scale); x[b] += att2[b] . vv[b]      (xp = pos_net output)
 *   q_lin, xp, x: [B,d];  kk, vv: [B,T_txt,d];  pe: [T_txt,d];  att1/att2: rows of length T_txt, att_sb apart. */
int lina_cross_att_step1(const void* q_lin, const void* ln_w, const void* ln_b, float ln_eps,
                         const void* kk, const void* pe, void* att1, int64_t att_sb, void* xp,
                         int B, int T_txt, int d, float scale, int dtype, lina_stream_t stream);
int lina_cross_att_step2(const void* xp, const void* pe, const void* vv, void* att2, int64_t att_sb, void* x,
                         int B, int T_txt, int d, float scale, int dtype, lina_stream_t stream);

/* The same two attention steps as spread kernels (>= 256 workgroups on the text-side tensors; the engine uses
 * these with lina_linear_skinny for  xp = att1 . pe  and  scores2 = xp . pe^T):
 *   lina_cross_scores     : scores[b,t] = scale * <LayerNorm(q_lin[b]), kk[b,t,:]>            (fp32 [B,T_txt])
 *   lina_softmax_rows     : att[b, 0:Tn] = softmax(x[b, 0:Tn] * scale)  -> strided `att` rows AND a contiguous
 *                           zero-padded copy attc [B, Tp] (Tp >= Tn) that feeds the next projection
 *   lina_weighted_rows_add: x[b,:] += sum_t attc[b,t] * vv[b,t,:]                                              */
/* Blind cross-attention step, first half after the scores (reference model/crossatt.py:117-127) in one launch:
 *   att[b,:Tn] = softmax(scores[b,:Tn])  (scores fp32, already scaled: lina_cross_scores);  xp[b,:] = att[b,:] . pe[:Tn,:]
 *   (pe [Tn,d] shared by all rows) -> xp [B,d] row-major and, when xp_packed is given, its fragment-major copy.
 * The att LOG of the device-side decode loop (reference model/modeling_lina.py:157,180: `atts.append(att)` / torch.cat over
 * steps): with att_step != NULL row b is stored at  att + b*att_sb + att_step[0]*att_step_stride  (a device step counter,
 * so one captured launch is valid at every step) and dropped when att_step[0] is outside [0, att_steps); att_step == NULL:
 * att + b*att_sb as before.  Same three arguments on lina_pe_softmax_weighted_rows_add. */
int lina_softmax_pe_rows(const float* scores, int64_t scores_sb, void* att, int64_t att_sb,
                         const int64_t* att_step, int64_t att_step_stride, int64_t att_steps, const void* pe, void* xp,
                         void* xp_packed, int B, int Tn, int d, int dtype, lina_stream_t stream);

/* Round-2 fusions of the same step (fewer launches on the serial chain):
 *   lina_cross_scores_softmax      = lina_cross_scores + lina_softmax_rows(scale 1) in one launch (one 1024-thread
 *                                    workgroup per utterance row): att rows (strided) + zero-padded attc [B,Tp];
 *   lina_softmax_weighted_rows_add = lina_softmax_rows(scores * scale) + lina_weighted_rows_add in one launch: scores
 *                                    [B, >= T_txt] in the model dtype; the weights are rounded to the model dtype as
 *                                    lina_softmax_rows stores them; att rows written by one workgroup per row; x (or, if
 *                                    x_packed != NULL, the fragment-major residual stream) += att . vv.
 * Same reference lines: model/crossatt.py:114-155 (eager softmax(q k^T / sqrt d) v of :13-19). */
int lina_cross_scores_softmax(const void* q_lin, const void* ln_w, const void* ln_b, float ln_eps, const void* kk,
                              void* att, int64_t att_sb, void* attc, int B, int T_txt, int Tp, int d, float scale,
                              int dtype, lina_stream_t stream);
int lina_softmax_weighted_rows_add(const void* scores, int64_t scores_sb, float scale, void* att, int64_t att_sb,
                                   const void* vv, void* x, void* x_packed, int B, int T_txt, int d, int dtype,
                                   lina_stream_t stream);
/* Round 4: the LAST TWO launches of the cross-attention step as one -- the raw scores of the second attention,
 *     sc[b, t] = < xp[b, :], pe[t, :] >      (reference model/crossatt.py:125-127: q = x_pos, k = the positional table;
 *                                             fp32 accumulate, rounded to `dtype` as the projection GEMM stored them),
 * computed by every workgroup for its own row (T_txt x d multiply-adds), then exactly lina_softmax_weighted_rows_add.
 * xp: [B, d] row-major, or (xp_packed != 0) the fragment-major layout of lina_linear_skinny_ex; pe: [>= T_txt, d] row-major;
 * d % 256 == 0. */
int lina_pe_softmax_weighted_rows_add(const void* xp, int xp_packed, const void* pe, float scale, void* att, int64_t att_sb,
                                      const int64_t* att_step, int64_t att_step_stride, int64_t att_steps, const void* vv, void* x, void* x_packed, int B, int T_txt, int d, int dtype,
                                      lina_stream_t stream);
int lina_cross_scores(const void* q_lin, const void* ln_w, const void* ln_b, float ln_eps, const void* kk,
                      float* scores, int B, int T_txt, int d, float scale, int dtype, lina_stream_t stream);
int lina_softmax_rows(const void* x, int64_t x_sb, int x_dtype, float scale, void* att, int64_t att_sb,
                      void* attc, int B, int T_txt, int Tp, int dtype, lina_stream_t stream);
int lina_weighted_rows_add(const void* attc, int Tp, const void* vv, void* x, int B, int T_txt, int d,
                           int dtype, lina_stream_t stream);

/* ---- f-3: the codes -> waveform step after the generation path (reference 3rdparty/decoder) ---- */

/* K8 -- depthwise conv (kernel 7, "same" zero padding along L) fused with the LayerNorm / AdaLayerNorm over C
 * that follows it in ConvNeXtBlock (reference 3rdparty/decoder/modules.py:44-50, 62-82), channels-last:
 *   x, y: [B, L, C] contiguous;  w: [C, 7];  bias: [C] or NULL;
 *   scale / shift: NULL, or per-batch rows (row stride scale_sb elements; 0 = one shared row), applied as
 *   LN(z) * scale + shift (LayerNorm weight/bias, or the AdaLayerNorm embeddings already looked up). C <= 1024. */
int lina_dwconv7_ln(const void* x, const void* w, const void* bias, const void* scale, const void* shift,
                    void* y, int B, int L, int C, int64_t scale_sb, float eps, int dtype, lina_stream_t stream);

/* K9 -- ISTFT overlap-add with "same" padding (reference 3rdparty/decoder/spectral_ops.py:56-75):
 *   frames: fp32 [B, T, win] inverse-transformed frames (NOT yet windowed); window: fp32 [win];
 *   y: fp32 [B, (T-1)*hop + win - 2*((win-hop)/2)] = sum of the windowed frames / window envelope, trimmed. */
int lina_istft_ola(const float* frames, const float* window, float* y, int B, int T, int win, int hop,
                   lina_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LINA_GLA_H */
