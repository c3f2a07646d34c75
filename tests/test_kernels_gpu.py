"""HIP kernels on a real MI355X vs the CPU oracle, through the C ABI (ops -> ctypes -> liblina_gla.so)."""
import pytest
import torch

from kernel_cases import (check_chunk_bwd_full, check_chunk_segmented, check_topk_sample, assert_close, check_argmax, check_chunk, check_chunk_bwd, check_conv_bwd, check_embed_bwd, check_rmsnorm_bwd, check_conv, check_cross_att, check_cross_spread, check_decode_update, check_decode_update_norm, check_embed, check_inproj,
                          check_linear_skinny, check_prologue, check_recurrent, check_rmsnorm, check_swiglu,
                          make_gla_inputs, oracle_gla)
from lina_speech_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("Dk,Dv,T,dtype", [(64, 64, 1, torch.float32), (128, 64, 3, torch.float32),
                                           (256, 128, 1, torch.float32), (256, 256, 5, torch.float32),
                                           (64, 64, 2, torch.bfloat16), (256, 256, 1, torch.bfloat16)])
def test_recurrent(hip, Dk, Dv, T, dtype):
    check_recurrent(DEV, B=3, H=2, T=T, Dk=Dk, Dv=Dv, dtype=dtype)


@pytest.mark.parametrize("Dk,Dv,T,dtype", [(64, 64, 37, torch.float32), (128, 64, 20, torch.float32),
                                           (256, 64, 18, torch.float32), (256, 256, 100, torch.float32),
                                           (64, 64, 33, torch.bfloat16), (128, 128, 17, torch.bfloat16),
                                           (256, 256, 130, torch.bfloat16)])
def test_chunk(hip, Dk, Dv, T, dtype):
    check_chunk(DEV, B=2, H=2, T=T, Dk=Dk, Dv=Dv, dtype=dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_chunk_reset_gates(hip, dtype):
    check_chunk(DEV, B=2, H=2, T=70, Dk=128, Dv=64, dtype=dtype, resets=True)


@pytest.mark.parametrize("T,resets", [(5, False), (32, False), (33, False), (200, False), (100, True)])
def test_chunk_full_head_kernel(hip, T, resets):
    # bf16, Dk = Dv = 256: the one-workgroup-per-head kernel (gla_chunk_full.hip), incl. the adaptive cut
    check_chunk(DEV, B=3, H=4, T=T, Dk=256, Dv=256, dtype=torch.bfloat16, resets=resets)


@pytest.mark.parametrize("T,W,dtype", [(1, 4, torch.float32), (3, 4, torch.float32), (37, 4, torch.float32),
                                       (300, 4, torch.float32), (19, 3, torch.bfloat16), (64, 4, torch.bfloat16)])
def test_conv(hip, T, W, dtype):
    check_conv(DEV, B=3, T=T, D=1000, W=W, dtype=dtype)


@pytest.mark.parametrize("D,dtype", [(64, torch.float32), (256, torch.float32), (1024, torch.float32),
                                     (512, torch.bfloat16)])
def test_rmsnorm(hip, D, dtype):
    check_rmsnorm(DEV, rows=37, D=D, dtype=dtype)


def test_embed_argmax_swiglu_prologue(hip):
    for dt in (torch.float32, torch.bfloat16):
        check_embed(DEV, Q=2, B=5, n=3, n_emb=4099, d=1024, dtype=dt)
        check_argmax(DEV, rows=64, n=4099, dtype=dt)
        check_swiglu(DEV, rows=64, hidden=1365, dtype=dt)
        check_prologue(DEV, B=64, Kd=1024, Vd=1024, dtype=dt)
    check_prologue(DEV, B=3, Kd=64, Vd=128, dtype=torch.float32, clamp_min=-0.05)


@pytest.mark.parametrize("Dk,Dv,dtype", [(64, 64, torch.float32), (128, 256, torch.float32),
                                         (256, 256, torch.float32), (256, 256, torch.bfloat16)])
def test_decode_update_rowsplit(hip, Dk, Dv, dtype):
    check_decode_update(DEV, B=5, H=4, Dk=Dk, Dv=Dv, dtype=dtype)


@pytest.mark.parametrize("kw", [
    dict(M=5, N=20, K=64, dtype=torch.float32),
    dict(M=64, N=4112, K=1024, dtype=torch.float32, ln=True),
    dict(M=64, N=4112, K=1024, dtype=torch.bfloat16, ln=True),
    dict(M=70, N=1024, K=1024, dtype=torch.bfloat16, resid=True),
    dict(M=64, N=1376, K=1024, dtype=torch.bfloat16, ln=True, bias=True, swiglu=1365),
    dict(M=64, N=1376, K=1024, dtype=torch.float32, ln=True, bias=True, swiglu=1365),
    dict(M=64, N=1024, K=1376, dtype=torch.bfloat16, resid=True),
    dict(M=64, N=4099, K=1024, dtype=torch.bfloat16),
    dict(M=130, N=100, K=96, dtype=torch.float32, resid=True, bias=True),
])
def test_linear_skinny(hip, kw):
    check_linear_skinny(DEV, **kw)


@pytest.mark.parametrize("B,K,Kd,Vd,dtype", [(5, 64, 32, 48, torch.float32), (64, 1024, 1024, 1024, torch.float32),
                                              (64, 1024, 1024, 1024, torch.bfloat16), (70, 256, 128, 256, torch.bfloat16)])
def test_inproj_fused(hip, B, K, Kd, Vd, dtype):
    check_inproj(DEV, B=B, K=K, Kd=Kd, Vd=Vd, dtype=dtype)


@pytest.mark.parametrize("B,H,Dk,Dv,dtype,rep", [(3, 2, 64, 64, torch.float32, 3), (5, 4, 256, 128, torch.bfloat16, 3),
                                                  (64, 4, 256, 256, torch.bfloat16, 40), (64, 4, 256, 256, torch.float32, 20),
                                                  (200, 4, 256, 256, torch.bfloat16, 10)])
def test_decode_update_norm_fused(hip, B, H, Dk, Dv, dtype, rep):
    check_decode_update_norm(DEV, B=B, H=H, Dk=Dk, Dv=Dv, dtype=dtype, repeats=rep)


@pytest.mark.parametrize("B,Tn,d,dtype", [(3, 9, 64, torch.float32), (64, 64, 1024, torch.float32),
                                           (64, 64, 1024, torch.bfloat16), (5, 300, 1024, torch.bfloat16)])
def test_cross_att_fused(hip, B, Tn, d, dtype):
    check_cross_att(DEV, B=B, Tn=Tn, d=d, dtype=dtype)


@pytest.mark.parametrize("B,Tn,d,dtype", [(3, 9, 64, torch.float32), (64, 64, 1024, torch.float32),
                                           (64, 64, 1024, torch.bfloat16), (5, 300, 1024, torch.bfloat16)])
def test_cross_att_spread(hip, B, Tn, d, dtype):
    check_cross_spread(DEV, B=B, Tn=Tn, d=d, dtype=dtype)


# ---------------------------------------------------------------------------------------------
# BASELINE.json sizes (B=64, H=4, Dk=Dv=256): size-independent properties, no CPU oracle needed
# ---------------------------------------------------------------------------------------------
def test_full_size_decode_step_properties(hip):
    B, H, Dk, Dv = 64, 4, 256, 256
    q, k, v, gk, h0 = make_gla_inputs(B, H, 1, Dk, Dv, torch.float32, DEV, seed=3)
    o, S = ops.fused_recurrent_gla(q, k, v, gk, initial_state=h0, output_final_state=True)
    # (1) in-place == out-of-place, bit for bit
    h_in = h0.clone()
    o_i, S_i = ops.fused_recurrent_gla(q, k, v, gk, initial_state=h_in, output_final_state=True, inplace_state=True)
    assert S_i.data_ptr() == h_in.data_ptr() and torch.equal(S_i, S) and torch.equal(o_i, o)
    # (2) closed form of one step: S' = exp(g) S + k^T v ; o = scale q S'
    S_ref = h0 * gk[:, :, 0].exp().unsqueeze(-1) + k[:, :, 0].unsqueeze(-1) * v[:, :, 0].unsqueeze(-2)
    o_ref = torch.einsum("bhk,bhkv->bhv", q[:, :, 0] * Dk ** -0.5, S_ref)
    assert_close(S, S_ref, 1e-6, "full-size S'")
    assert_close(o[:, :, 0], o_ref, 1e-5, "full-size o")
    # (3) linearity in v with a zero state
    v2 = torch.randn_like(v)
    oa, _ = ops.fused_recurrent_gla(q, k, v, gk)
    ob, _ = ops.fused_recurrent_gla(q, k, v2, gk)
    oc, _ = ops.fused_recurrent_gla(q, k, v + v2, gk)
    assert_close(oc, oa + ob, 1e-5, "linearity in v")
    # (4) the chunk kernel agrees on the same single step
    o2, S2 = ops.chunk_gla(q, k, v, gk, initial_state=h0, output_final_state=True)
    assert_close(o2, o, 1e-4, "K2 == K1 (o)")
    assert_close(S2, S, 1e-4, "K2 == K1 (S)")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_full_length_chunk_equals_recurrent_kernel(hip, dtype):
    # seqlen 4096 (config 5): K2's final state / outputs == K1 run sequentially over the same tokens
    B, H, T, Dk, Dv = 2, 4, 4096, 256, 256
    q, k, v, gk, h0 = make_gla_inputs(B, H, T, Dk, Dv, dtype, DEV, seed=4)
    gk = (gk.float() / 4).to(dtype)             # logsigmoid/16 scale, as in the model
    o1, S1 = ops.fused_recurrent_gla(q, k, v, gk, initial_state=h0, output_final_state=True)
    o2, S2 = ops.chunk_gla(q, k, v, gk, initial_state=h0, output_final_state=True)
    tol = 2e-4 if dtype == torch.float32 else 3e-2
    assert_close(o2, o1.float(), tol, "T=4096 K2 vs K1 (o)")
    assert_close(S2, S1, 2e-4 if dtype == torch.float32 else 2e-2, "T=4096 K2 vs K1 (S)")
    # split the sequence: prefill 3000 tokens then 1096 more from the carried state == one pass
    oa, Sa = ops.chunk_gla(q[:, :, :3000], k[:, :, :3000], v[:, :, :3000], gk[:, :, :3000], initial_state=h0,
                           output_final_state=True)
    ob, Sb = ops.chunk_gla(q[:, :, 3000:], k[:, :, 3000:], v[:, :, 3000:], gk[:, :, 3000:], initial_state=Sa,
                           output_final_state=True)
    assert_close(torch.cat([oa, ob], 2), o2, tol, "split prefill (o)")
    assert_close(Sb, S2, 1e-3 if dtype == torch.float32 else 2e-2, "split prefill (S)")


def test_missing_device_tensor_is_an_error(hip):
    x = torch.randn(2, 2, 1, 64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.fused_recurrent_gla(x, x, x, x)


# ----------------------------------------------------------------------------- backward kernels
@pytest.mark.parametrize("Dk,Dv,T,dtype", [(64, 64, 37, torch.float32), (128, 256, 50, torch.float32),
                                           (256, 256, 150, torch.float32), (256, 256, 150, torch.bfloat16),
                                           (64, 128, 33, torch.bfloat16)])
def test_chunk_bwd(hip, Dk, Dv, T, dtype):
    check_chunk_bwd(DEV, B=2, H=2, T=T, Dk=Dk, Dv=Dv, dtype=dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_chunk_bwd_reset_gates(hip, dtype):
    check_chunk_bwd(DEV, B=2, H=2, T=70, Dk=128, Dv=64, dtype=dtype, resets=True)
    check_chunk_bwd(DEV, B=1, H=2, T=70, Dk=64, Dv=64, dtype=dtype, resets=True, with_h0=False, with_dht=False,
                    via="fused_chunk_gla")


@pytest.mark.parametrize("T,nseg,resets,h0,dht", [(40, 1, False, False, False), (150, 1, True, True, True),
                                                   (300, 4, True, True, True), (1000, 8, False, True, False)])
def test_chunk_bwd_full_head_sweeps(hip, T, nseg, resets, h0, dht):
    check_chunk_bwd_full(DEV, 2, 2, T, 256, nseg, resets=resets, with_h0=h0, with_dht=dht)


@pytest.mark.parametrize("D,H,T,nseg", [(128, 2, 150, 1), (64, 4, 150, 2), (128, 4, 300, 4)])
def test_chunk_bwd_full_head_sweeps_head_groups(hip, D, H, T, nseg):
    check_chunk_bwd_full(DEV, 2, H, T, D, nseg, resets=True)


def test_chunk_bwd_full_head_sweeps_at_the_training_sequence_length(hip):
    # one segment per head against the fp64 oracle; 16 segments: test_chunk_bwd_at_the_training_sequence_length_with_resets
    # (through autograd), 8 segments: test_chunk_bwd_segments_agree_with_the_single_pass
    check_chunk_bwd_full(DEV, 1, 2, 4096, 256, 1, resets=True, seed=43)


def test_chunk_bwd_segments_agree_with_the_single_pass(hip):
    # size-independent property at the training shape: the segment-parallel sweeps equal the one-segment sweeps
    B, H, T, D = 2, 4, 4096, 256
    q, k, v, gk, h0 = make_gla_inputs(B, H, T, D, D, torch.bfloat16, DEV, seed=19)
    d_o = torch.randn(B, H, T, D, generator=torch.Generator().manual_seed(20)).to(torch.bfloat16).to(DEV)
    a = ops.gla_chunk_bwd(q, k, v, gk, d_o, D ** -0.5, h0, need_dh0=True, nseg=1, path="full")
    b = ops.gla_chunk_bwd(q, k, v, gk, d_o, D ** -0.5, h0, need_dh0=True, nseg=8, path="full")
    for name, x, y in zip(("dq", "dk", "dv", "dg", "dh0"), a, b):
        # dg: two differently-ordered suffix sums of cancelling bf16-noisy terms over 4096 tokens (see check_chunk_bwd_long)
        assert_close(y.float(), x.float(), 4e-2 if name == "dg" else 2e-2, f"K2b segments vs single pass {name}")


def test_chunk_bwd_generic_kernel_for_bf16(hip, monkeypatch):
    monkeypatch.setattr(ops.POLICY, "k2b_path", "sweeps")
    check_chunk_bwd(DEV, B=2, H=2, T=150, Dk=256, Dv=256, dtype=torch.bfloat16)
    check_chunk_bwd(DEV, B=2, H=2, T=70, Dk=64, Dv=64, dtype=torch.bfloat16, resets=True)


def test_chunk_two_value_blocks_in_one_launch_equals_two_launches(hip, monkeypatch):
    from kernel_cases import check_chunk_dv512_one_launch
    check_chunk_dv512_one_launch(DEV, monkeypatch, B=16, H=4, T=1000)  # 64 heads: the XCD-paired block-id mapping, ragged T
    check_chunk_dv512_one_launch(DEV, monkeypatch, B=3, H=3, T=300)    # heads % 8 != 0: the plain mapping
    check_chunk_dv512_one_launch(DEV, monkeypatch, B=64, H=4, T=4096, oracle=False)   # the benchmark shape: 512 workgroups, two rounds


def test_chunk_and_its_backward_for_value_column_blocks(hip):
    # expand_v = 2 heads (256 x 512): the forward as ONE launch of two workgroups per head, the backward as two full-head calls
    # on column blocks; the long case also runs segment-parallel
    check_chunk(DEV, B=2, H=2, T=300, Dk=256, Dv=512, dtype=torch.bfloat16, resets=True)
    check_chunk_bwd(DEV, B=2, H=2, T=150, Dk=256, Dv=512, dtype=torch.bfloat16, resets=True)
    from kernel_cases import check_chunk_bwd_long
    check_chunk_bwd_long(DEV, B=1, H=1, T=2048, Dk=256, Dv=512, dtype=torch.bfloat16, reset_every=600)


def test_chunk_bwd_is_linear_in_the_output_gradient(hip):
    # size-independent property at the training shape: grads(do1 + do2) == grads(do1) + grads(do2)
    B, H, T, D = 2, 4, 1024, 256
    q, k, v, gk, _ = make_gla_inputs(B, H, T, D, D, torch.bfloat16, DEV, seed=9)
    g = torch.Generator().manual_seed(10)
    d1 = torch.randn(B, H, T, D, generator=g).to(torch.bfloat16).to(DEV)
    d2 = torch.randn(B, H, T, D, generator=g).to(torch.bfloat16).to(DEV)
    scale = D ** -0.5
    a = ops.gla_chunk_bwd(q, k, v, gk, d1, scale)
    b = ops.gla_chunk_bwd(q, k, v, gk, d2, scale)
    c = ops.gla_chunk_bwd(q, k, v, gk, (d1.float() + d2.float()).to(torch.bfloat16), scale)
    for name, x, y, z in zip(("dq", "dk", "dv", "dg"), a, b, c):
        assert_close(z.float(), x.float() + y.float(), 3e-2, f"K2b linearity {name}")


@pytest.mark.parametrize("T,W,dtype,bias,act", [(70, 4, torch.float32, False, "silu"), (5, 4, torch.float32, True, None),
                                                (300, 4, torch.bfloat16, False, "silu"), (130, 3, torch.bfloat16, True, "silu")])
def test_conv_bwd(hip, T, W, dtype, bias, act):
    check_conv_bwd(DEV, B=3, T=T, D=1024, W=W, dtype=dtype, use_bias=bias, activation=act)


@pytest.mark.parametrize("D,dtype,gate,affine", [(256, torch.float32, True, True), (64, torch.float32, False, True),
                                                 (1024, torch.float32, True, False), (256, torch.bfloat16, True, True)])
def test_rmsnorm_bwd(hip, D, dtype, gate, affine):
    check_rmsnorm_bwd(DEV, rows=2500, D=D, dtype=dtype, gate=gate, affine=affine)


def test_embed_bwd(hip):
    check_embed_bwd(DEV, Q=2, B=3, n=50, n_emb=4099, d=1024, dtype=torch.float32)


@pytest.mark.parametrize("n,k,temp,dtype", [(4099, 100, 1.0, torch.float32), (300, 7, 0.7, torch.float32),
                                            (50, 100, 1.3, torch.float32), (4099, 100, 0.9, torch.bfloat16),
                                            (8192, 1000, 1.0, torch.float32)])
def test_topk_sample(hip, n, k, temp, dtype):
    check_topk_sample(DEV, rows=64, n=n, k=k, temp=temp, dtype=dtype)


@pytest.mark.parametrize("T,nseg,resets", [(100, 3, False), (300, 4, True), (1024, 8, False), (257, 16, False)])
def test_chunk_segment_parallel(hip, T, nseg, resets):
    check_chunk_segmented(DEV, B=2, H=2, T=T, nseg=nseg, resets=resets)


def test_chunk_default_policy_uses_segments_for_small_batches(hip):
    from lina_speech_amd.ops import chunk_segments
    assert chunk_segments(32, 4096) == 8 and chunk_segments(256, 4096) == 1 and chunk_segments(4, 512) == 1
    assert chunk_segments(8, 4096) == 16


# ----------------------------------------------------------------------------- codes -> waveform (f-3)
@pytest.mark.parametrize("C,dtype,ada", [(64, torch.float32, False), (768, torch.float32, True), (768, torch.bfloat16, True),
                                         (1024, torch.bfloat16, False)])
def test_dwconv7_ln(hip, C, dtype, ada):
    from kernel_cases import check_dwconv7_ln
    check_dwconv7_ln(DEV, B=3, L=200, C=C, dtype=dtype, ada=ada)


@pytest.mark.parametrize("T,win,hop", [(20, 64, 16), (5, 40, 10), (750, 1280, 320)])
def test_istft_ola(hip, T, win, hop):
    from kernel_cases import check_istft_ola
    check_istft_ola(DEV, B=3, T=T, win=win, hop=hop)


# ----------------------------------------------------------------------------- config 1: scalar-gate GLA (a-12)
@pytest.mark.parametrize("Dk,Dv,T,dtype,h0", [(64, 64, 256, torch.float32, True), (64, 64, 70, torch.float32, False),
                                              (256, 256, 300, torch.bfloat16, True), (128, 128, 257, torch.bfloat16, False)])
def test_chunk_simple_gla(hip, Dk, Dv, T, dtype, h0):
    from kernel_cases import check_chunk_simple
    check_chunk_simple(DEV, B=4, H=4, T=T, Dk=Dk, Dv=Dv, dtype=dtype, with_h0=h0)


# ----------------------------------------------------------------------------- config 5 sequence length (T = 4096)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_chunk_bwd_at_the_training_sequence_length_with_resets(hip, dtype):
    """K2b vs fp64 autograd through the oracle at T=4096, b=1, H=4, 256x256, resets every 512 tokens."""
    from kernel_cases import check_chunk_bwd_long
    check_chunk_bwd_long(DEV, B=1, H=4, T=4096, Dk=256, Dv=256, dtype=dtype, reset_every=512)


def test_chunk_bwd_long_without_boundary_states(hip):
    from kernel_cases import check_chunk_bwd_long
    check_chunk_bwd_long(DEV, B=1, H=2, T=2048 + 37, Dk=256, Dv=256, dtype=torch.bfloat16, reset_every=700,
                         with_h0=False, with_dht=False)


@pytest.mark.parametrize("nseg,resets", [(8, False), (16, True)])
def test_chunk_segment_parallel_at_the_training_sequence_length(hip, nseg, resets):
    """Segment-parallel K2 (the form config 5's micro-batch takes) at T=4096 vs the fp64 recurrent oracle."""
    check_chunk_segmented(DEV, B=1, H=4, T=4096, nseg=nseg, resets=resets)


# ----------------------------------------------------------------------------- K1w: windowed decode-step update
@pytest.mark.parametrize("B,H,Dk,Dv,dtype,window,n", [(64, 4, 256, 256, torch.bfloat16, 8, 21), (3, 2, 256, 256, torch.float32, 8, 17),
                                                      (5, 8, 128, 128, torch.bfloat16, 4, 10), (2, 16, 64, 64, torch.float32, 8, 9),
                                                      (64, 4, 256, 512, torch.bfloat16, 8, 18), (3, 2, 128, 512, torch.float32, 8, 9),
                                                      (64, 4, 256, 256, torch.bfloat16, 16, 37), (3, 2, 64, 64, torch.float32, 16, 33)])
def test_decode_window(hip, B, H, Dk, Dv, dtype, window, n):
    from kernel_cases import check_decode_window
    check_decode_window(DEV, B=B, H=H, Dk=Dk, Dv=Dv, dtype=dtype, window=window, n_steps=n)


@pytest.mark.parametrize("B,H,Dk,Dv,window,n", [(64, 4, 256, 256, 1, 9), (64, 4, 256, 256, 8, 21), (5, 8, 128, 128, 4, 10),
                                                (3, 4, 256, 512, 8, 10)])
def test_decode_window_bf16_state(hip, B, H, Dk, Dv, window, n):
    """Opt-in bf16 recurrent state (reference model/gla.py:229-240 + Cache.update for a bf16 model): window 1 = rounded after
    every step, as the reference does; window W = every W-th step."""
    from kernel_cases import check_decode_window
    check_decode_window(DEV, B=B, H=H, Dk=Dk, Dv=Dv, dtype=torch.bfloat16, window=window, n_steps=n, state_dtype=torch.bfloat16)


@pytest.mark.parametrize("B,Q,L,d,dtype", [(64, 1, 4099, 1024, torch.bfloat16), (7, 4, 1027, 256, torch.float32)])
def test_greedy_pick_embed(hip, B, Q, L, d, dtype):
    from kernel_cases import check_greedy_pick_embed
    check_greedy_pick_embed(DEV, B=B, Q=Q, L=L, d=d, dtype=dtype, steps=4)


@pytest.mark.parametrize("M,N,K,dtype,ln,bias,resid,sw,force", [
    (512, 4099, 1024, torch.bfloat16, False, False, False, 0, False),        # the codec head at the metric's batch
    (512, 1376, 1024, torch.bfloat16, True, True, False, 1365, False),       # LN-2 + up-projection + SwiGLU (+ bias column)
    (400, 1376, 1024, torch.bfloat16, True, True, False, 1365, False),
    (130, 1024, 1376, torch.bfloat16, False, False, True, 0, True),          # down-projection shape, ragged rows, 43 k-steps
    (200, 4099, 1024, torch.float32, False, True, False, 0, True),
    (129, 56, 288, torch.float32, True, True, True, 40, True)])
@pytest.mark.parametrize("variant", [0, 1, 2])
def test_linear_tall(hip, M, N, K, dtype, ln, bias, resid, sw, force, variant):
    from kernel_cases import check_linear_tall
    check_linear_tall(DEV, M, N, K, dtype, ln=ln, bias=bias, resid=resid, swiglu=sw, force=force, variant=variant)


@pytest.mark.parametrize("B,K,Kd,Vd,dtype,force", [(512, 1024, 1024, 1024, torch.bfloat16, False),
                                                   (130, 1024, 1024, 2048, torch.bfloat16, True),
                                                   (70, 160, 128, 64, torch.bfloat16, True),
                                                   (200, 256, 256, 256, torch.float32, True)])
@pytest.mark.parametrize("variant", [0, 1, 2, 3])        # 3 (the default above 256 rows) falls back to 0 where K is not whole 8-k-step groups
def test_inproj_tall(hip, B, K, Kd, Vd, dtype, force, variant):
    from kernel_cases import check_inproj_tall
    check_inproj_tall(DEV, B, K, Kd, Vd, dtype, force=force, variant=variant, same_as_variant=0 if variant == 3 else None)


@pytest.mark.parametrize("B,Q,L,d,dtype,sampled", [(64, 1, 4099, 1024, torch.bfloat16, False), (7, 4, 1027, 256, torch.float32, False),
                                                   (64, 1, 4099, 1024, torch.bfloat16, True), (9, 3, 513, 64, torch.float32, True)])
def test_pick_loop_control_block(hip, B, Q, L, d, dtype, sampled):
    from kernel_cases import check_pick_loop_ctl
    check_pick_loop_ctl(DEV, B=B, Q=Q, L=L, d=d, dtype=dtype, sampled=sampled)


@pytest.mark.parametrize("B,Q,L,d,dtype,ns,k", [(64, 1, 4099, 1024, torch.bfloat16, 1, 100),
                                                (7, 4, 1027, 256, torch.float32, 2, 5),
                                                (9, 3, 513, 64, torch.bfloat16, 0, 3)])
def test_sample_pick_embed(hip, B, Q, L, d, dtype, ns, k):
    from kernel_cases import check_sample_pick_embed
    check_sample_pick_embed(DEV, B=B, Q=Q, L=L, d=d, dtype=dtype, n_sampled=ns, k=k, temp=1.0)


@pytest.mark.parametrize("N,D,xd,rd,yd", [(32768, 1024, torch.float32, torch.bfloat16, torch.bfloat16),
                                           (4099, 1024, torch.float32, None, torch.bfloat16),
                                           (1000, 256, torch.float32, torch.float32, torch.float32),
                                           (777, 1024, torch.bfloat16, torch.bfloat16, torch.bfloat16),
                                           (130, 2048, torch.float32, torch.float32, torch.bfloat16)])
def test_layer_norm(hip, N, D, xd, rd, yd):
    from kernel_cases import check_layer_norm
    check_layer_norm(DEV, N, D, xd, rd, yd)


@pytest.mark.parametrize("N,H,dtype", [(32768, 1365, torch.bfloat16), (70000, 64, torch.float32), (513, 1365, torch.float32),
                                       (32768, 1408, torch.bfloat16), (4099, 260, torch.float32)])
def test_swiglu_gate(hip, N, H, dtype):
    from kernel_cases import check_swiglu_gate
    check_swiglu_gate(DEV, N, H, dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_swiglu_gate_unit_column_is_exactly_one(hip, dtype):
    from kernel_cases import check_swiglu_unit_column
    check_swiglu_unit_column(DEV, dtype)


# ----------------------------------------------------------------------------- fragment-major (packed) projections
@pytest.mark.parametrize("M,N,K,dtype,ln,bias,resid,sw", [(64, 1024, 1024, torch.bfloat16, False, False, True, 0),
                                                          (64, 1376, 1024, torch.bfloat16, True, True, False, 1365),
                                                          (64, 2080, 768, torch.bfloat16, True, True, False, 2048),
                                                          (64, 1024, 1376, torch.bfloat16, False, False, True, 0),
                                                          (64, 4099, 1024, torch.bfloat16, False, False, False, 0),
                                                          (33, 300, 256, torch.float32, True, True, True, 0),
                                                          (64, 64, 1024, torch.bfloat16, False, False, False, 0)])
def test_linear_skinny_packed(hip, M, N, K, dtype, ln, bias, resid, sw):
    from kernel_cases import check_linear_skinny_packed
    check_linear_skinny_packed(DEV, M, N, K, dtype, ln=ln, bias=bias, resid=resid, swiglu=sw)


@pytest.mark.parametrize("B,dtype", [(64, torch.bfloat16), (9, torch.float32)])
def test_inproj_packed(hip, B, dtype):
    from kernel_cases import check_inproj_packed
    check_inproj_packed(DEV, B=B, K=1024, Kd=1024, Vd=1024, dtype=dtype)


# ----------------------------------------------------------------------------- full-head K2 for D = 128 / 64 (2 / 4 heads per workgroup)
@pytest.mark.parametrize("B,H,D,T,resets", [(2, 8, 128, 300, True), (1, 16, 64, 257, False), (3, 4, 64, 100, True),
                                            (2, 2, 128, 1024, False)])
def test_chunk_full_head_kernel_head_groups(hip, B, H, D, T, resets):
    check_chunk(DEV, B=B, H=H, T=T, Dk=D, Dv=D, dtype=torch.bfloat16, resets=resets)


@pytest.mark.parametrize("H,D,T,nseg", [(8, 128, 1024, 8), (16, 64, 300, 4)])
def test_chunk_segment_parallel_head_groups(hip, H, D, T, nseg):
    check_chunk_segmented(DEV, B=1, H=H, T=T, nseg=nseg, resets=True, D=D)


@pytest.mark.parametrize("B,Tn,d,dtype", [(64, 64, 1024, torch.bfloat16), (5, 100, 256, torch.float32), (64, 20, 1024, torch.float32),
                                          (512, 64, 1024, torch.bfloat16), (300, 33, 512, torch.float32)])
def test_cross_attention_fusions(hip, B, Tn, d, dtype):
    from kernel_cases import check_cross_fused, check_softmax_pe_rows
    check_cross_fused(DEV, B, Tn, d, dtype)
    check_softmax_pe_rows(DEV, B, Tn, d, dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("clamp", [None, -0.2])
@pytest.mark.parametrize("n", [4096, 64 * 1024 * 512 + 4])
def test_gate_logsigmoid(hip, n, dtype, clamp):
    from kernel_cases import check_gate_logsigmoid
    check_gate_logsigmoid(DEV, n, dtype, clamp)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("in_place", [True, False])
@pytest.mark.parametrize("B,T,H,D", [(2, 70, 2, 64), (3, 515, 4, 256)])
def test_split_slab(hip, B, T, H, D, dtype, in_place):
    from kernel_cases import check_split_slab
    check_split_slab(DEV, B, T, H, D, dtype, in_place)


@pytest.mark.parametrize("dtype,bias,through_gla", [(torch.float32, True, True), (torch.bfloat16, False, True), (torch.bfloat16, True, False)])
@pytest.mark.parametrize("B,T,H,D", [(2, 70, 2, 64), (2, 515, 4, 256)])
def test_short_conv3_fused_qkv(hip, B, T, H, D, dtype, bias, through_gla):
    from kernel_cases import check_short_conv3
    check_short_conv3(DEV, B, T, H, D, dtype, bias, through_gla)


@pytest.mark.parametrize("n_out,n_in,bias", [(1024, 1365, True), (2730, 1024, True), (1024, 1024, False), (1024, 16, True),
                                             (4112, 1024, False)])
def test_linear_train_path_under_autocast(hip, n_out, n_in, bias):
    """ops.linear under bf16 autocast with fp32 master weights (the train step's setting) against F.linear under the same
    autocast: same bf16 y; dX equal; dW / db in fp32 and CLOSER to the fp64 product than autograd's bf16-rounded ones
    (token-split batched GEMM with fp32 partial products)."""
    rows = 16384
    g = torch.Generator().manual_seed(8)
    x = torch.randn(4, rows // 4, n_in, generator=g).to(DEV)
    w = (torch.randn(n_out, n_in, generator=g) * 0.05).to(DEV)
    b = torch.randn(n_out, generator=g).to(DEV) if bias else None
    dy = torch.randn(4, rows // 4, n_out, generator=g).to(torch.bfloat16).to(DEV)
    res = []
    for fn in (ops.linear, torch.nn.functional.linear):
        xx, ww = x.clone().requires_grad_(), w.clone().requires_grad_()
        bb = None if b is None else b.clone().requires_grad_()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = fn(xx, ww, bb)
        assert y.dtype == torch.bfloat16
        (y.float() * dy.float()).sum().backward()
        assert ww.grad.dtype == torch.float32 and xx.grad.dtype == torch.float32
        res.append((y, xx.grad, ww.grad, None if bb is None else bb.grad))
    assert torch.equal(res[0][0], res[1][0])
    assert torch.allclose(res[0][1], res[1][1], rtol=2e-2, atol=2e-2)
    xb, db = x.to(torch.bfloat16).double().view(rows, n_in), dy.double().view(rows, n_out)
    dw64 = db.t() @ xb
    e_mine = (res[0][2].double() - dw64).abs().max().item()
    e_auto = (res[1][2].double() - dw64).abs().max().item()
    assert e_mine <= max(e_auto, 1e-3 * dw64.abs().max().item()), (e_mine, e_auto)
    if bias:
        db64 = db.sum(0)
        assert (res[0][3].double() - db64).abs().max().item() <= max((res[1][3].double() - db64).abs().max().item(), 1e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,T,C,L,clamp,bias,strided", [(2, 70, 64, 16, None, True, True), (8, 4096, 1024, 16, None, True, True), (3, 1000, 1024, 16, -0.2, True, False), (2, 333, 2048, 7, None, False, False), (2, 70, 64, 16, None, True, False), (1, 150, 128, 16, -0.2, True, "aligned"), (3, 515, 1024, 16, None, True, "aligned"), (2, 300, 320, 16, None, False, "aligned")])
def test_gate_lowrank(hip, B, T, C, L, clamp, bias, strided, dtype):
    from kernel_cases import check_gate_lowrank
    check_gate_lowrank(DEV, B, T, C, L, dtype, clamp, bias, strided)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,T,d,H,bias", [(2, 70, 48, 21, True), (4, 2048, 1024, 1365, True), (1, 300, 256, 128, False)])
def test_swiglu_mlp(hip, B, T, d, H, bias, dtype):
    from kernel_cases import check_swiglu_mlp
    check_swiglu_mlp(DEV, B, T, d, H, dtype, bias)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_block_chain_with_pending_branch(hip, dtype):
    from kernel_cases import check_block_chain
    check_block_chain(DEV, dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,V,ld", [(37, 131, None), (4096, 4099, None), (777, 2050, None), (64, 8445, None), (100, 64, 72)])
def test_cross_entropy(hip, N, V, ld, dtype):
    from kernel_cases import check_cross_entropy
    check_cross_entropy(DEV, N, V, dtype, ld)


def test_sum_partials(hip):
    from kernel_cases import check_sum_partials
    check_sum_partials(DEV)


@pytest.mark.parametrize("N,V,ld,dtype", [(1, 4, None, torch.float32), (3, 5, 9, torch.bfloat16), (2, 257, 261, torch.bfloat16),
                                          (5, 1023, None, torch.bfloat16)])
def test_cross_entropy_edge_shapes(hip, N, V, ld, dtype):
    from kernel_cases import check_cross_entropy
    check_cross_entropy(DEV, N, V, dtype, ld)


def test_column_sum(hip):
    from kernel_cases import check_column_sum
    check_column_sum(DEV)


def test_train_weight_operands_in_one_pass(hip):
    from kernel_cases import check_mlp_pack, check_stacked_linear
    check_mlp_pack(DEV)
    check_mlp_pack(DEV, out_dtype=torch.float32, bias=False)
    check_mlp_pack(DEV, H=127, d_in=8, d_out=3)              # H + 1 = Hp: the bias column is the last one
    check_stacked_linear(DEV)
    check_stacked_linear(DEV, rows=(5,), pad=0)
    check_stacked_linear(DEV, rows=(512, 512, 16), n_in=16, pad=48, B=1, T=9, expect_split=True)    # main 1024 + tail 64
    check_stacked_linear(DEV, rows=(512, 500, 28), n_in=16, pad=48, B=1, T=9, expect_split=False)   # a block straddles the cut
    check_stacked_linear(DEV, rows=(1024, 1024, 1024, 1024, 16), n_in=1024, pad=48, B=2, T=4096, autocast=True, expect_split=True)   # the L169 mixer's stack under autocast
    check_mlp_pack(DEV, H=1365, d_in=1024, d_out=1024)


def test_fused_adamw_matches_torch_adamw(hip):
    from kernel_cases import check_fused_adamw
    check_fused_adamw(DEV)
