// train_weights.hip -- the per-step OPERAND forms of the train path's master weights, one launch each (the
// reference's step casts every fp32 parameter to bf16 under autocast, train_lina.py:72-120 through torch.autocast).
//
// The train path feeds its GEMMs operands that are not the parameters' own layout: the five projections of a mixer input stacked
// into one [4160, 1024] weight (mixer.py), the channel mixer's two weights zero-padded to an aligned hidden width with the biases
// folded in (autograd._SwiGLUMLPFunction).  Built with torch ops that took ~11 launches per block and step (cat, cast, three
// fills, five strided copies: ~190 launches and ~0.9 ms of a 51 ms step, profiles/r06_train_step_kernel_stats.csv); here each
// operand set is ONE pass: read the fp32 master weights once, write the GEMM-dtype operand in its final layout.
//   K15  lina_mlp_pack     w_in [2H, d_in], b_in [2H], w_out [d_out, H], b_out [d_out]  ->  Wi [2, Hp, d_in], bi [2, Hp], Wo [d_out, Hq >= Hp]
//   K16  lina_stack_rows   up to 8 row blocks [r_i, cols]  ->  one [R, cols] operand (rows past the blocks: zero)
#include <lina_dev.h>
#include "lina_common.h"

namespace lina {

constexpr int kStackMax = 8;
struct stack_srcs {
    const float* p[kStackMax];
    int end[kStackMax];          // end[i] = first row AFTER block i in the stacked operand
};

// one thread = four consecutive output elements of one of the three operands (all three row widths are multiples of 4)
template <typename TO>
__global__ __launch_bounds__(256) void mlp_pack_kernel(const float* __restrict__ w_in, const float* __restrict__ b_in,
                                                       const float* __restrict__ w_out, const float* __restrict__ b_out,
                                                       TO* __restrict__ Wi, TO* __restrict__ bi, TO* __restrict__ Wo, int H,
                                                       int Hp, int Hq, int d_in, int d_out) {
    // Hq >= Hp: row length of Wo (columns past H + 1 are zero): the down-projection's dX GEMM runs on the whole [d_out, Hq]
    // operand when Hq is the width the GEMM library prefers (1536 for Hp = 1408), everything else on its first Hp columns
    const int64_t n_wi = (int64_t)2 * Hp * d_in, n_wo = (int64_t)d_out * Hq, n_bi = (int64_t)2 * Hp;
    const int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e < n_wi) {                                          // Wi[s][r][c] = w_in[s H + r][c] for r < H, else 0
        const int64_t row = e / d_in;
        const int c = (int)(e - row * d_in), s = (int)(row / Hp), r = (int)(row - (int64_t)s * Hp);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < H) v = *reinterpret_cast<const float4*>(w_in + ((int64_t)s * H + r) * d_in + c);
        st4(Wi + e, v);
        return;
    }
    int64_t f = e - n_wi;
    if (f < n_wo) {                                          // Wo[r][c] = w_out[r][c] for c < H; column H = b_out (the bias rides in the GEMM)
        const int r = (int)(f / Hq), c0 = (int)(f - (int64_t)r * Hq);
        float x[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = c0 + i;
            x[i] = c < H ? w_out[(int64_t)r * H + c] : (c == H && b_out) ? b_out[r] : 0.0f;
        }
        st4(Wo + f, make_float4(x[0], x[1], x[2], x[3]));
        return;
    }
    f -= n_wo;
    if (f < n_bi) {                                          // bi[s][r] = b_in[s H + r]; (32, 1/32) at r = H makes the gate's column H exactly 1
        const int s = (int)(f / Hp), r0 = (int)(f - (int64_t)s * Hp);
        float x[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + i;
            x[i] = r < H ? (b_in ? b_in[(int64_t)s * H + r] : 0.0f) : (r == H && b_out) ? (s == 0 ? 32.0f : 0.03125f) : 0.0f;
        }
        st4(bi + f, make_float4(x[0], x[1], x[2], x[3]));
    }
}

template <typename TO>
__global__ __launch_bounds__(256) void stack_rows_kernel(stack_srcs src, int n_src, int cols, int total_rows, TO* __restrict__ out) {
    const int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= (int64_t)total_rows * cols) return;
    const int row = (int)(e / cols), c = (int)(e - (int64_t)row * cols);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    int begin = 0;
#pragma unroll
    for (int i = 0; i < kStackMax; ++i) {
        if (i < n_src) {
            if (row >= begin && row < src.end[i]) v = *reinterpret_cast<const float4*>(src.p[i] + (int64_t)(row - begin) * cols + c);
            begin = src.end[i];
        }
    }
    st4(out + e, v);
}

}  // namespace lina

extern "C" int lina_mlp_pack(const float* w_in, const float* b_in, const float* w_out, const float* b_out, void* Wi, void* bi,
                             void* Wo, int H, int Hp, int Hq, int d_in, int d_out, int out_dtype, lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(w_in && w_out && Wi && bi && Wo, "lina_mlp_pack: null pointer");
    LINA_REQUIRE(H > 0 && Hp > H && d_in > 0 && d_out > 0, "lina_mlp_pack: needs 0 < H < Hp and positive widths");
    LINA_REQUIRE(Hp % 4 == 0 && d_in % 4 == 0, "lina_mlp_pack: Hp and d_in must be multiples of 4");
    LINA_REQUIRE(Hq >= Hp && Hq % 4 == 0, "lina_mlp_pack: Hq must be a multiple of 4, >= Hp");
    LINA_REQUIRE(valid_dtype(out_dtype), "lina_mlp_pack: bad dtype %d", out_dtype);
    const int64_t n = (int64_t)2 * Hp * d_in + (int64_t)d_out * Hq + (int64_t)2 * Hp;
    dim3 grid((unsigned)((n / 4 + 255) / 256));
    if (out_dtype == LINA_F32)
        LINA_LAUNCH((mlp_pack_kernel<float>), grid, dim3(256), 0, stream, w_in, b_in, w_out, b_out, (float*)Wi, (float*)bi, (float*)Wo,
                    H, Hp, Hq, d_in, d_out);
    else
        LINA_LAUNCH((mlp_pack_kernel<bf16_t>), grid, dim3(256), 0, stream, w_in, b_in, w_out, b_out, (bf16_t*)Wi, (bf16_t*)bi,
                    (bf16_t*)Wo, H, Hp, Hq, d_in, d_out);
    return check_launch("lina_mlp_pack");
}

extern "C" int lina_stack_rows(const float* const* srcs, const int* rows, int n_src, int cols, int total_rows, void* out,
                               int out_dtype, lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(srcs && rows && out, "lina_stack_rows: null pointer");
    LINA_REQUIRE(n_src >= 1 && n_src <= kStackMax, "lina_stack_rows: n_src=%d not in 1..%d", n_src, kStackMax);
    LINA_REQUIRE(cols > 0 && cols % 4 == 0, "lina_stack_rows: cols must be a positive multiple of 4");
    LINA_REQUIRE(valid_dtype(out_dtype), "lina_stack_rows: bad dtype %d", out_dtype);
    stack_srcs s{};
    int64_t end = 0;
    for (int i = 0; i < n_src; ++i) {
        LINA_REQUIRE(srcs[i] && rows[i] > 0, "lina_stack_rows: block %d is empty", i);
        LINA_REQUIRE(((uintptr_t)srcs[i] & 15u) == 0, "lina_stack_rows: block %d is not 16-byte aligned", i);
        end += rows[i];
        s.p[i] = srcs[i];
        s.end[i] = (int)end;
    }
    LINA_REQUIRE(end <= total_rows, "lina_stack_rows: the blocks hold %lld rows, the operand %d", (long long)end, total_rows);
    const int64_t n = (int64_t)total_rows * cols;
    dim3 grid((unsigned)((n / 4 + 255) / 256));
    if (out_dtype == LINA_F32) LINA_LAUNCH((stack_rows_kernel<float>), grid, dim3(256), 0, stream, s, n_src, cols, total_rows, (float*)out);
    else LINA_LAUNCH((stack_rows_kernel<bf16_t>), grid, dim3(256), 0, stream, s, n_src, cols, total_rows, (bf16_t*)out);
    return check_launch("lina_stack_rows");
}

// ------------------------------------------------------------------------------------------------------------
// K17 -- AdamW over MANY tensors per launch (reference train_lina.py:104-118: torch.optim.AdamW, lr 5e-4, betas (0.9, 0.999),
// weight decay 0.1).  The decoupled-weight-decay update of torch's own `_fused_adamw_`, same operation order in fp32:
//     p -= lr wd p;   m += (1 - b1)(g - m);   v = b2 v + (1 - b2) g g;   p -= (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps)
// with bc1 = 1 - b1^t, bc2 = 1 - b2^t formed on the host in double.  One launch takes up to kAdamMax tensors by value (no
// device-side table to keep current: gradient tensors are new allocations every step); a block owns 4096 consecutive elements of
// one tensor and streams p, g, m, v with 16-byte accesses (m, v, g non-temporal: touched once per step).
namespace lina {

constexpr int kAdamMax = 48;
constexpr int kAdamBlock = 4096;          // elements per block: 256 threads x 4 pieces of 4
struct adam_tensors {
    float* p[kAdamMax];
    const float* g[kAdamMax];
    float* m[kAdamMax];
    float* v[kAdamMax];
    int64_t n[kAdamMax];
    int blk_end[kAdamMax];                // first block AFTER tensor i's blocks
};

__global__ __launch_bounds__(256) void adamw_multi_kernel(adam_tensors T, int n_tensors, float step_size, float decay, float omb1,
                                                          float beta2, float omb2, float eps, float bc2_sqrt) {
    int t = 0, b0 = 0;
    const int blk = blockIdx.x;
#pragma unroll 1
    while (t < n_tensors - 1 && blk >= T.blk_end[t]) { b0 = T.blk_end[t]; ++t; }
    if (t > 0) b0 = T.blk_end[t - 1];
    float* __restrict__ p = T.p[t];
    const float* __restrict__ g = T.g[t];
    float* __restrict__ m = T.m[t];
    float* __restrict__ v = T.v[t];
    const int64_t n = T.n[t];
    const int64_t base = (int64_t)(blk - b0) * kAdamBlock;
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        pp -= decay * pp;
        mm = mm + omb1 * (gg - mm);
        vv = beta2 * vv + omb2 * gg * gg;
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        pp -= step_size * mm / denom;
    };
    const bool vec = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15u) == 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t e = base + (int64_t)(i * 256 + threadIdx.x) * 4;
        if (e >= n) break;
        if (vec && e + 4 <= n) {
            float4 pp = *reinterpret_cast<const float4*>(p + e);
            const float4 gg = ld_nt4(g + e);
            float4 mm = ld_nt4(m + e), vv = ld_nt4(v + e);
            upd(pp.x, gg.x, mm.x, vv.x); upd(pp.y, gg.y, mm.y, vv.y); upd(pp.z, gg.z, mm.z, vv.z); upd(pp.w, gg.w, mm.w, vv.w);
            *reinterpret_cast<float4*>(p + e) = pp;
            st_nt4(m + e, mm);
            st_nt4(v + e, vv);
        } else {
            for (int64_t j = e; j < n && j < e + 4; ++j) {
                float pp = p[j], mm = m[j], vv = v[j];
                upd(pp, g[j], mm, vv);
                p[j] = pp; m[j] = mm; v[j] = vv;
            }
        }
    }
}

}  // namespace lina

extern "C" int lina_adamw_multi_max(void) { return lina::kAdamMax; }

extern "C" int lina_adamw_multi(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                                const int64_t* numel, int n_tensors, double lr, double beta1, double beta2, double eps,
                                double weight_decay, double bias_correction1, double bias_correction2, lina_stream_t stream) {
    using namespace lina;
    LINA_REQUIRE(params && grads && exp_avg && exp_avg_sq && numel, "lina_adamw_multi: null pointer");
    LINA_REQUIRE(n_tensors >= 0, "lina_adamw_multi: n_tensors must be >= 0");
    LINA_REQUIRE(bias_correction1 > 0.0 && bias_correction2 > 0.0, "lina_adamw_multi: bias corrections must be positive (step >= 1)");
    for (int i0 = 0; i0 < n_tensors; i0 += kAdamMax) {
        adam_tensors T{};
        const int nt = n_tensors - i0 < kAdamMax ? n_tensors - i0 : kAdamMax;
        int64_t blocks = 0;
        for (int i = 0; i < nt; ++i) {
            const int j = i0 + i;
            LINA_REQUIRE(params[j] && grads[j] && exp_avg[j] && exp_avg_sq[j] && numel[j] > 0, "lina_adamw_multi: tensor %d is empty", j);
            T.p[i] = params[j]; T.g[i] = grads[j]; T.m[i] = exp_avg[j]; T.v[i] = exp_avg_sq[j]; T.n[i] = numel[j];
            blocks += (numel[j] + kAdamBlock - 1) / kAdamBlock;
            LINA_REQUIRE(blocks < (1LL << 31), "lina_adamw_multi: too many elements in one launch");
            T.blk_end[i] = (int)blocks;
        }
        // the scalars are formed in double and rounded once (1 - 0.999 taken in fp32 is off by 1.3e-5 of itself)
        LINA_LAUNCH(adamw_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, T, nt, (float)(lr / bias_correction1),
                    (float)(lr * weight_decay), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
                    (float)sqrt(bias_correction2));
    }
    return check_launch("lina_adamw_multi");
}
