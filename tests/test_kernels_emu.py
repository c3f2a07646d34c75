"""Kernel SOURCES vs the CPU oracle, run on the wave64 emulator (no GPU needed).
The same checks run on the real device in tests/test_kernels_gpu.py."""
import pytest
import torch

from oracle import gla_oracle as O
from kernel_cases import (make_gla_inputs, check_chunk_segmented, check_topk_sample, check_argmax, check_chunk, check_chunk_bwd, check_conv_bwd, check_rmsnorm_bwd, check_embed_bwd, check_conv, check_cross_att, check_cross_spread, check_decode_update, check_decode_update_norm, check_embed, check_inproj, check_linear_skinny,
                          check_prologue, check_recurrent, check_rmsnorm, check_swiglu, check_chunk_simple, check_chunk_bwd_full)

from lina_speech_amd import ops

DEV = "cpu"


@pytest.mark.parametrize("Dk,Dv,T,dtype", [(64, 64, 1, torch.float32), (128, 64, 3, torch.float32),
                                           (256, 128, 1, torch.float32), (64, 64, 2, torch.bfloat16)])
def test_recurrent(emu, Dk, Dv, T, dtype):
    check_recurrent(DEV, B=2, H=2, T=T, Dk=Dk, Dv=Dv, dtype=dtype)


@pytest.mark.parametrize("Dk,Dv,T,dtype", [(64, 64, 37, torch.float32), (128, 64, 20, torch.float32),
                                           (256, 64, 18, torch.float32), (64, 64, 33, torch.bfloat16),
                                           (128, 128, 17, torch.bfloat16)])
def test_chunk(emu, Dk, Dv, T, dtype):
    check_chunk(DEV, B=1, H=2, T=T, Dk=Dk, Dv=Dv, dtype=dtype)


# T = 200: seven chunks with a ragged tail; the un-normalised state is renormalised several times (gates ~ -0.28 per token)
@pytest.mark.parametrize("T,resets", [(1, False), (5, False), (32, False), (33, False), (40, False), (64, False), (65, True),
                                      (70, True), (200, False)])
def test_chunk_full_head_kernel(emu, T, resets):
    check_chunk(DEV, B=1, H=1, T=T, Dk=256, Dv=256, dtype=torch.bfloat16, resets=resets)


@pytest.mark.parametrize("Dk,Dv,T,dtype", [(64, 64, 37, torch.float32), (128, 64, 21, torch.float32),
                                            (64, 128, 33, torch.bfloat16)])
def test_chunk_bwd(emu, Dk, Dv, T, dtype):
    check_chunk_bwd(DEV, B=1, H=2, T=T, Dk=Dk, Dv=Dv, dtype=dtype)


@pytest.mark.parametrize("T", [70, 101])
def test_chunk_single_saturated_gate_full_and_partial_chunk(emu, T):
    """one gate below -60 among zeros, in a full and in the last partial chunk: forward (MODE 0) on the full-head
    kernel and the three backward sweeps (MODE 1) vs the fp64 oracle / fp64 autograd."""
    from kernel_cases import check_chunk_bwd_full
    check_chunk(DEV, B=1, H=1, T=T, Dk=256, Dv=256, dtype=torch.bfloat16, resets="saturated")
    check_chunk_bwd_full(DEV, B=1, H=1, T=T, D=256, nseg=1, resets="saturated")


def test_chunk_bwd_reset_gates_no_state(emu):
    check_chunk_bwd(DEV, B=1, H=1, T=40, Dk=64, Dv=64, dtype=torch.float32, resets=True, with_h0=False, with_dht=False)


def test_chunk_bwd_reset_gates(emu):
    check_chunk_bwd(DEV, B=1, H=1, T=40, Dk=64, Dv=64, dtype=torch.float32, resets=True, via="fused_chunk_gla")


def test_chunk_reset_gates(emu):
    # adversarial gates: runs of -20 resets force the adaptive chunk cut
    check_chunk(DEV, B=1, H=1, T=40, Dk=64, Dv=64, dtype=torch.float32, resets=True)


@pytest.mark.parametrize("T,W,dtype", [(1, 4, torch.float32), (3, 4, torch.float32), (37, 4, torch.float32),
                                       (19, 3, torch.bfloat16)])
def test_conv(emu, T, W, dtype):
    check_conv(DEV, B=2, T=T, D=96, W=W, dtype=dtype)


@pytest.mark.parametrize("D,dtype", [(64, torch.float32), (256, torch.float32), (512, torch.bfloat16)])
def test_rmsnorm(emu, D, dtype):
    check_rmsnorm(DEV, rows=7, D=D, dtype=dtype)


def test_embed_argmax_swiglu_prologue(emu):
    check_embed(DEV, Q=2, B=3, n=2, n_emb=37, d=64, dtype=torch.float32)
    check_argmax(DEV, rows=5, n=4099, dtype=torch.float32)
    check_swiglu(DEV, rows=3, hidden=85, dtype=torch.float32)
    check_prologue(DEV, B=3, Kd=64, Vd=128, dtype=torch.float32)
    check_prologue(DEV, B=2, Kd=64, Vd=64, dtype=torch.bfloat16)


@pytest.mark.parametrize("Dk,Dv,dtype", [(64, 64, torch.float32), (128, 256, torch.float32), (256, 128, torch.bfloat16)])
def test_decode_update_rowsplit(emu, Dk, Dv, dtype):
    check_decode_update(DEV, B=2, H=2, Dk=Dk, Dv=Dv, dtype=dtype)


@pytest.mark.parametrize("kw", [
    dict(M=5, N=20, K=64, dtype=torch.float32),
    dict(M=64, N=48, K=128, dtype=torch.float32, ln=True, bias=True),
    dict(M=70, N=33, K=96, dtype=torch.float32, resid=True, bias=True),
    dict(M=9, N=96, K=64, dtype=torch.float32, ln=True, bias=True, swiglu=85),
    dict(M=7, N=40, K=160, dtype=torch.bfloat16, ln=True),
    dict(M=66, N=64, K=64, dtype=torch.bfloat16, ln=True, bias=True, swiglu=37),
    dict(M=3, N=17, K=352, dtype=torch.bfloat16, resid=True),
    dict(M=40, N=2080, K=32, dtype=torch.bfloat16, ln=True, bias=True),      # tiling (MT,NT) = (4,1)/(2,2) territory
    dict(M=64, N=4112, K=32, dtype=torch.bfloat16, ln=True),                 # (4,2): 64 rows x 32 columns per workgroup
    dict(M=32, N=4112, K=32, dtype=torch.float32, resid=True),               # (2,2)
    dict(M=64, N=1376, K=64, dtype=torch.bfloat16, ln=True, bias=True, swiglu=1365),   # (2,1) with SwiGLU halves
])
def test_linear_skinny(emu, kw):
    check_linear_skinny(DEV, **kw)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_inproj_fused(emu, dtype):
    check_inproj(DEV, B=5, K=64, Kd=32, Vd=48, dtype=dtype)       # 16-column workgroups
    check_inproj(DEV, B=70, K=64, Kd=64, Vd=32, dtype=dtype)      # 32-column workgroups, two row blocks


@pytest.mark.parametrize("Dk,Dv,dtype", [(64, 64, torch.float32), (256, 128, torch.bfloat16), (128, 256, torch.float32)])
def test_decode_update_norm_fused(emu, Dk, Dv, dtype):
    check_decode_update_norm(DEV, B=2, H=2, Dk=Dk, Dv=Dv, dtype=dtype, repeats=2)


@pytest.mark.parametrize("Tn,d,dtype", [(9, 64, torch.float32), (70, 128, torch.bfloat16)])
def test_cross_att_fused(emu, Tn, d, dtype):
    check_cross_att(DEV, B=3, Tn=Tn, d=d, dtype=dtype)


@pytest.mark.parametrize("Tn,d,dtype", [(9, 64, torch.float32), (70, 128, torch.bfloat16)])
def test_cross_att_spread(emu, Tn, d, dtype):
    check_cross_spread(DEV, B=5, Tn=Tn, d=d, dtype=dtype)


@pytest.mark.parametrize("T,W,dtype,bias,act", [(70, 4, torch.float32, False, "silu"), (5, 4, torch.float32, True, None),
                                                (130, 3, torch.bfloat16, True, "silu")])
def test_conv_bwd(emu, T, W, dtype, bias, act):
    check_conv_bwd(DEV, B=2, T=T, D=96, W=W, dtype=dtype, use_bias=bias, activation=act)


@pytest.mark.parametrize("D,dtype,gate,affine", [(256, torch.float32, True, True), (64, torch.float32, False, True),
                                                 (128, torch.float32, True, False), (256, torch.bfloat16, True, True)])
def test_rmsnorm_bwd(emu, D, dtype, gate, affine):
    check_rmsnorm_bwd(DEV, rows=9, D=D, dtype=dtype, gate=gate, affine=affine)


def test_embed_bwd(emu):
    check_embed_bwd(DEV, Q=2, B=3, n=5, n_emb=11, d=64, dtype=torch.float32)


@pytest.mark.parametrize("n,k,temp,dtype", [(4099, 100, 1.0, torch.float32), (300, 7, 0.7, torch.float32),
                                            (50, 100, 1.3, torch.float32), (1030, 100, 0.9, torch.bfloat16)])
def test_topk_sample(emu, n, k, temp, dtype):
    check_topk_sample(DEV, rows=6, n=n, k=k, temp=temp, dtype=dtype, draws=60)


@pytest.mark.parametrize("T,nseg,resets", [(100, 3, False), (70, 2, True), (64, 2, False), (33, 4, False)])
def test_chunk_segment_parallel(emu, T, nseg, resets):
    check_chunk_segmented(DEV, B=1, H=1, T=T, nseg=nseg, resets=resets)


@pytest.mark.parametrize("C,dtype,ada", [(64, torch.float32, False), (768, torch.float32, True), (96, torch.bfloat16, True)])
def test_dwconv7_ln(emu, C, dtype, ada):
    from kernel_cases import check_dwconv7_ln
    check_dwconv7_ln(DEV, B=2, L=11, C=C, dtype=dtype, ada=ada)


@pytest.mark.parametrize("T,win,hop", [(20, 64, 16), (5, 40, 10), (3, 1280, 320)])
def test_istft_ola(emu, T, win, hop):
    from kernel_cases import check_istft_ola
    check_istft_ola(DEV, B=2, T=T, win=win, hop=hop)


def test_chunk_bwd_through_the_final_state_only(emu):
    # loss depends on the final state alone: the output gradient arrives as None
    from kernel_cases import make_gla_inputs, assert_close
    q, k, v, gk, h0 = make_gla_inputs(1, 1, 20, 64, 64, torch.float32, DEV, seed=8)
    leaves = [x.detach().clone().requires_grad_(True) for x in (q, k, v, gk)]
    _, S = O.naive_recurrent_gla(*[x.double() for x in leaves], initial_state=h0.double(), output_final_state=True,
                                 compute_dtype=torch.float64)
    S.sum().backward()
    ref = [torch.zeros_like(x) if x.grad is None else x.grad.clone() for x in leaves]   # q does not reach the state
    for x in leaves:
        x.grad = None
    from lina_speech_amd import ops
    _, S2 = ops.chunk_gla(*leaves, initial_state=h0, output_final_state=True)
    S2.sum().backward()
    for name, x, r in zip(("dq", "dk", "dv", "dg"), leaves, ref):
        if float(r.abs().max()) == 0.0:
            assert float(x.grad.abs().max()) < 1e-6, name
        else:
            assert_close(x.grad, r, 2e-4, f"K2b {name} (final state only)")


@pytest.mark.parametrize("Dk,Dv,T,dtype,h0", [(64, 64, 37, torch.float32, True), (64, 128, 21, torch.bfloat16, False)])
def test_chunk_simple_gla(emu, Dk, Dv, T, dtype, h0):
    check_chunk_simple(DEV, B=1, H=2, T=T, Dk=Dk, Dv=Dv, dtype=dtype, with_h0=h0)


@pytest.mark.parametrize("Dk,Dv,dtype,window,n", [(64, 64, torch.float32, 8, 19), (128, 128, torch.bfloat16, 4, 9),
                                                  (64, 256, torch.float32, 2, 5), (64, 64, torch.float32, 1, 3),
                                                  (256, 256, torch.bfloat16, 8, 10), (256, 128, torch.float32, 8, 9),
                                                  (64, 512, torch.float32, 4, 6), (128, 512, torch.bfloat16, 8, 9),
                                                  (64, 64, torch.float32, 16, 35), (256, 256, torch.bfloat16, 16, 18)])
def test_decode_window(emu, Dk, Dv, dtype, window, n):
    from kernel_cases import check_decode_window
    check_decode_window(DEV, B=2, H=2, Dk=Dk, Dv=Dv, dtype=dtype, window=window, n_steps=n)


# the opt-in bf16 recurrent state (the reference's state dtype for a bf16 model): rounded at every write-back -- every step at
# window 1 (the reference's arithmetic), every window-th step otherwise
@pytest.mark.parametrize("Dk,Dv,window,n", [(256, 256, 1, 6), (128, 128, 8, 11), (64, 512, 4, 6)])
def test_decode_window_bf16_state(emu, Dk, Dv, window, n):
    from kernel_cases import check_decode_window
    check_decode_window(DEV, B=2, H=2, Dk=Dk, Dv=Dv, dtype=torch.bfloat16, window=window, n_steps=n, state_dtype=torch.bfloat16)


@pytest.mark.parametrize("Q,L,d,dtype", [(1, 300, 64, torch.float32), (3, 70, 32, torch.bfloat16)])
def test_greedy_pick_embed(emu, Q, L, d, dtype):
    from kernel_cases import check_greedy_pick_embed
    check_greedy_pick_embed(DEV, B=5, Q=Q, L=L, d=d, dtype=dtype)


@pytest.mark.parametrize("N,D,xd,rd,yd", [(9, 64, torch.float32, None, torch.float32),
                                           (13, 320, torch.float32, torch.float32, torch.float32),
                                           (7, 256, torch.float32, torch.bfloat16, torch.bfloat16),
                                           (5, 1024, torch.float32, torch.float32, torch.bfloat16),
                                           (6, 128, torch.bfloat16, torch.bfloat16, torch.bfloat16),
                                           (4, 2048, torch.float32, None, torch.bfloat16)])
def test_layer_norm(emu, N, D, xd, rd, yd):
    from kernel_cases import check_layer_norm
    check_layer_norm(DEV, N, D, xd, rd, yd)


@pytest.mark.parametrize("N,H,dtype", [(5, 64, torch.float32), (7, 85, torch.float32), (3, 85, torch.bfloat16),
                                       (1100, 132, torch.bfloat16), (1027, 260, torch.float32),   # sequence form (rows >= 1024)
                                       (9, 128, torch.bfloat16)])
def test_swiglu_gate(emu, N, H, dtype):
    from kernel_cases import check_swiglu_gate
    check_swiglu_gate(DEV, N, H, dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_swiglu_gate_unit_column_is_exactly_one(emu, dtype):
    from kernel_cases import check_swiglu_unit_column
    check_swiglu_unit_column(DEV, dtype)


@pytest.mark.parametrize("nw", [8, 16])
def test_projections_wider_split_k(emu, monkeypatch, nw):
    """The packed projection kernels with 8 / 16 waves per workgroup (K = 1024: 32 k-steps): same products, another
    summation order -- against the 4-wave row-major kernel within fp32 rounding of the partial sums."""
    from kernel_cases import check_linear_skinny_packed, check_inproj_packed
    monkeypatch.setenv("LINA_SKINNY_WAVES", str(nw))
    check_linear_skinny_packed(DEV, 20, 48, 1024, torch.bfloat16, ln=False, bias=False, resid=True)
    check_linear_skinny_packed(DEV, 33, 64, 1024, torch.bfloat16, ln=True, bias=True, swiglu=40)
    check_linear_skinny_packed(DEV, 7, 16, 512, torch.float32, ln=True, bias=True, resid=True)
    check_inproj_packed(DEV, 9, 1024, 32, 32, torch.bfloat16)
    check_inproj_packed(DEV, 5, 1024, 16, 48, torch.bfloat16)


_TALL_CASES = [(130, 70, 160, torch.bfloat16, False, True, True, 0), (70, 100, 96, torch.bfloat16, True, True, False, 0),
               (129, 56, 288, torch.bfloat16, True, True, False, 40), (33, 64, 128, torch.bfloat16, False, False, False, 64),
               (140, 48, 80, torch.float32, True, True, True, 0), (20, 40, 48, torch.float32, True, False, False, 21)]


# variant 0 (the launchers' default) on every case; variants 1 and 2 (A/B builds of the operand path) on two cases each
@pytest.mark.parametrize("variant,case", [(0, c) for c in _TALL_CASES] + [(v, _TALL_CASES[i]) for v in (1, 2) for i in (2, 4)])
def test_linear_tall(emu, variant, case):
    from kernel_cases import check_linear_tall
    M, N, K, dtype, ln, bias, resid, sw = case
    check_linear_tall(DEV, M, N, K, dtype, ln=ln, bias=bias, resid=resid, swiglu=sw, variant=variant)


@pytest.mark.parametrize("B,K,Kd,Vd,dtype,variant", [(130, 96, 64, 128, torch.bfloat16, 0), (70, 160, 128, 64, torch.bfloat16, 0),
                                                     (129, 48, 64, 64, torch.float32, 0), (70, 160, 128, 64, torch.bfloat16, 1),
                                                     (129, 48, 64, 64, torch.float32, 2),
                                                     # variant 3 (128-row workgroups, gate folded in; K = whole groups of 8 k-steps):
                                                     # ragged gate-channel shares (64 channels over 6 column blocks), a row block
                                                     # with idle waves, two groups of k-steps
                                                     (130, 256, 64, 128, torch.bfloat16, 3), (70, 512, 128, 64, torch.bfloat16, 3),
                                                     (200, 128, 64, 64, torch.float32, 3)])
def test_inproj_tall(emu, B, K, Kd, Vd, dtype, variant):
    from kernel_cases import check_inproj_tall
    check_inproj_tall(DEV, B, K, Kd, Vd, dtype, variant=variant, same_as_variant=0 if variant == 3 else None)


@pytest.mark.parametrize("Q,L,d,dtype,sampled", [(1, 70, 32, torch.float32, False), (3, 70, 32, torch.bfloat16, False),
                                                 (2, 70, 32, torch.float32, True)])
def test_pick_loop_control_block(emu, Q, L, d, dtype, sampled):
    from kernel_cases import check_pick_loop_ctl
    check_pick_loop_ctl(DEV, B=5, Q=Q, L=L, d=d, dtype=dtype, sampled=sampled)


@pytest.mark.parametrize("Q,L,d,dtype,ns", [(1, 300, 64, torch.float32, 1), (3, 70, 32, torch.bfloat16, 1),
                                            (3, 70, 32, torch.bfloat16, 3), (2, 50, 20, torch.float32, 0)])
def test_sample_pick_embed(emu, Q, L, d, dtype, ns):
    from kernel_cases import check_sample_pick_embed
    check_sample_pick_embed(DEV, B=5, Q=Q, L=L, d=d, dtype=dtype, n_sampled=ns)


@pytest.mark.parametrize("M,N,K,dtype,ln,bias,resid,sw", [(5, 40, 64, torch.float32, False, True, True, 0),
                                                          (20, 48, 64, torch.bfloat16, True, True, False, 0),
                                                          (7, 32, 64, torch.bfloat16, True, True, False, 21),
                                                          (9, 96, 64, torch.bfloat16, True, True, False, 64),
                                                          (64, 100, 96, torch.float32, False, False, True, 0)])
def test_linear_skinny_packed(emu, M, N, K, dtype, ln, bias, resid, sw):
    from kernel_cases import check_linear_skinny_packed
    check_linear_skinny_packed(DEV, M, N, K, dtype, ln=ln, bias=bias, resid=resid, swiglu=sw)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_inproj_packed(emu, dtype):
    from kernel_cases import check_inproj_packed
    check_inproj_packed(DEV, B=5, K=64, Kd=32, Vd=32, dtype=dtype)


# full-head K2 with 2 / 4 heads per workgroup (D = 128 / 64): block-diagonal state, one mask(A) per head
@pytest.mark.parametrize("H,D,T,resets", [(2, 128, 70, True), (4, 64, 45, False), (4, 128, 33, False), (8, 64, 100, True),
                                          (2, 128, 1, False)])
def test_chunk_full_head_kernel_head_groups(emu, H, D, T, resets):
    check_chunk(DEV, B=1, H=H, T=T, Dk=D, Dv=D, dtype=torch.bfloat16, resets=resets)


@pytest.mark.parametrize("H,D,T,nseg", [(2, 128, 100, 3), (4, 64, 70, 2)])
def test_chunk_segment_parallel_head_groups(emu, H, D, T, nseg):
    check_chunk_segmented(DEV, B=1, H=H, T=T, nseg=nseg, resets=True, D=D)


@pytest.mark.parametrize("B,Tn,d,dtype", [(3, 11, 64, torch.float32), (2, 40, 128, torch.bfloat16),
                                          (2, 13, 256, torch.float32), (3, 33, 256, torch.bfloat16),
                                          (257, 9, 512, torch.bfloat16)])      # B >= 256: one workgroup per row loops the slabs
def test_cross_attention_fusions(emu, B, Tn, d, dtype):
    from kernel_cases import check_cross_fused, check_softmax_pe_rows
    check_cross_fused(DEV, B, Tn, d, dtype)
    check_softmax_pe_rows(DEV, B, Tn, d, dtype)


# ----------------------------------------------------------------------------- K2b on the full-head kernel (three sweeps)
@pytest.mark.parametrize("T,nseg,resets,h0,dht", [(40, 1, False, False, False), (33, 1, False, True, True),
                                                   (70, 1, True, True, True), (100, 3, False, True, True),
                                                   (96, 3, True, False, True), (33, 2, True, True, False)])
def test_chunk_bwd_full_head_sweeps(emu, T, nseg, resets, h0, dht):
    check_chunk_bwd_full(DEV, 1, 1, T, 256, nseg, resets=resets, with_h0=h0, with_dht=dht)


@pytest.mark.parametrize("T,nseg", [(1, 1), (2, 1), (5, 4), (32, 1), (65, 2)])
def test_chunk_bwd_full_head_sweeps_edge_lengths(emu, T, nseg):
    # one token, fewer tokens than segments asked for, exactly one chunk, one token past a segment boundary
    check_chunk_bwd_full(DEV, 1, 1, T, 256, nseg)


@pytest.mark.parametrize("D,H,T,nseg", [(128, 2, 70, 1), (64, 4, 70, 2), (128, 4, 65, 2)])
def test_chunk_bwd_full_head_sweeps_head_groups(emu, D, H, T, nseg):
    check_chunk_bwd_full(DEV, 1, H, T, D, nseg, resets=True)


def test_chunk_bwd_generic_kernel_still_reachable_for_bf16(emu, monkeypatch):
    monkeypatch.setattr(ops.POLICY, "k2b_path", "sweeps")
    check_chunk_bwd(DEV, B=1, H=1, T=40, Dk=64, Dv=64, dtype=torch.bfloat16)


def test_chunk_autograd_hands_the_forward_segment_states_to_the_backward(emu):
    # ops.chunk_gla under autograd with forced segments: the backward takes the forward's boundary states (seg_states)
    B, H, T, D = 1, 1, 100, 256
    q, k, v, gk, h0 = make_gla_inputs(B, H, T, D, D, torch.bfloat16, DEV, seed=12, resets=True)
    d_o = torch.randn(B, H, T, D, generator=torch.Generator().manual_seed(13)).to(torch.bfloat16)
    leaves = [x.detach().clone().requires_grad_(True) for x in (q, k, v, gk)]
    o, _ = ops.chunk_gla(*leaves, initial_state=h0, nseg=3)
    (o.float() * d_o.float()).sum().backward()
    ref = ops.gla_chunk_bwd(q, k, v, gk, d_o, D ** -0.5, h0, nseg=3, path="full")      # recomputes the states itself
    for name, a, r in zip(("dq", "dk", "dv", "dg"), leaves, ref):
        assert torch.equal(a.grad, r), name


def test_chunk_and_its_backward_for_value_column_blocks(emu):
    # expand_v = 2 heads (256 x 512): the forward runs both value column blocks in ONE launch (two workgroups per head), K2b
    # as two 256 x 256 calls on column blocks of v, o and the states
    check_chunk(DEV, B=1, H=1, T=40, Dk=256, Dv=512, dtype=torch.bfloat16, resets=True)
    check_chunk_bwd(DEV, B=1, H=1, T=40, Dk=256, Dv=512, dtype=torch.bfloat16, resets=True)


def test_chunk_two_value_blocks_in_one_launch_equals_two_launches(emu, monkeypatch):
    from kernel_cases import check_chunk_dv512_one_launch
    check_chunk_dv512_one_launch(DEV, monkeypatch, B=1, H=8, T=33)     # 8 heads: the XCD-paired block-id mapping
    check_chunk_dv512_one_launch(DEV, monkeypatch, B=1, H=3, T=36)     # heads % 8 != 0: the plain mapping


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("clamp", [None, -0.2])
@pytest.mark.parametrize("n", [1024, 4100])
def test_gate_logsigmoid(emu, n, dtype, clamp):
    from kernel_cases import check_gate_logsigmoid
    check_gate_logsigmoid(DEV, n, dtype, clamp)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("in_place", [True, False])
@pytest.mark.parametrize("B,T,H,D", [(2, 70, 2, 64)])
def test_split_slab(emu, B, T, H, D, dtype, in_place):
    from kernel_cases import check_split_slab
    check_split_slab(DEV, B, T, H, D, dtype, in_place)


@pytest.mark.parametrize("dtype,bias,through_gla", [(torch.float32, True, True), (torch.bfloat16, False, True), (torch.bfloat16, True, False)])
def test_short_conv3_fused_qkv(emu, dtype, bias, through_gla):
    from kernel_cases import check_short_conv3
    check_short_conv3(DEV, 2, 70, 2, 64, dtype, bias, through_gla)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,T,C,L,clamp,bias,strided", [(2, 70, 64, 16, None, True, True), (1, 130, 40, 16, -0.2, True, False), (2, 33, 64, 7, None, False, False), (2, 70, 64, 16, None, True, False), (1, 150, 128, 16, -0.2, True, "aligned"), (2, 45, 320, 16, None, False, "aligned")])
def test_gate_lowrank(emu, B, T, C, L, clamp, bias, strided, dtype):
    from kernel_cases import check_gate_lowrank
    check_gate_lowrank(DEV, B, T, C, L, dtype, clamp, bias, strided)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,T,d,H,bias", [(2, 70, 48, 21, True), (1, 300, 32, 128, False)])
def test_swiglu_mlp(emu, B, T, d, H, bias, dtype):
    from kernel_cases import check_swiglu_mlp
    check_swiglu_mlp(DEV, B, T, d, H, dtype, bias)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_block_chain_with_pending_branch(emu, dtype):
    from kernel_cases import check_block_chain
    check_block_chain(DEV, dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,V,ld", [(37, 131, None), (20, 1027, None), (9, 64, 72)])
def test_cross_entropy(emu, N, V, ld, dtype):
    from kernel_cases import check_cross_entropy
    check_cross_entropy(DEV, N, V, dtype, ld)


def test_sum_partials(emu):
    from kernel_cases import check_sum_partials
    check_sum_partials(DEV)


@pytest.mark.parametrize("N,V,ld,dtype", [(1, 4, None, torch.float32), (3, 5, 9, torch.bfloat16), (2, 257, 261, torch.bfloat16),
                                          (5, 1023, None, torch.bfloat16)])
def test_cross_entropy_edge_shapes(emu, N, V, ld, dtype):
    from kernel_cases import check_cross_entropy
    check_cross_entropy(DEV, N, V, dtype, ld)


def test_column_sum(emu):
    from kernel_cases import check_column_sum
    check_column_sum(DEV)


def test_train_weight_operands_in_one_pass(emu):
    from kernel_cases import check_mlp_pack, check_stacked_linear
    check_mlp_pack(DEV)
    check_mlp_pack(DEV, out_dtype=torch.float32, bias=False)
    check_mlp_pack(DEV, H=127, d_in=8, d_out=3)              # H + 1 = Hp: the bias column is the last one
    check_mlp_pack(DEV, H=1365, d_in=8, d_out=5)             # Hp = 1408 in rows of Hq = 1536 (the wide dX operand)
    check_stacked_linear(DEV)
    check_stacked_linear(DEV, rows=(5,), pad=0)
    check_stacked_linear(DEV, rows=(512, 512, 16), n_in=16, pad=48, B=1, T=9, expect_split=True)    # main 1024 + tail 64
    check_stacked_linear(DEV, rows=(512, 500, 28), n_in=16, pad=48, B=1, T=9, expect_split=False)   # a block straddles the cut


def test_fused_adamw_matches_torch_adamw(emu):
    from kernel_cases import check_fused_adamw
    check_fused_adamw(DEV)
