"""Operator-level drop-in surface: the names the reference imports from ``fla``
(/root/reference/model/gla.py:19-23), served by the hand-written HIP kernels through
the C ABI of include/lina_gla.h.

Every function takes/returns torch tensors exactly like the fla operator it replaces
(head-first ``[B,H,T,D]`` views, ``(o, final_state)`` returns) and enqueues on the
CURRENT torch HIP stream without synchronising, so callers can graph-capture.
There is no CPU path: tensors must live on a ROCm device (``Backend.require``).
GLA ops are differentiable (K2b); the other custom ops raise on tensors that require grad
unless stated otherwise.

This module is the facade: the implementation lives in backend.py (provider of the C ABI, argument helpers), kernels.py
(thin launchers), autograd.py (differentiable operators) and policy.py (launch policy, ``POLICY``).
"""
from __future__ import annotations

from . import backend, policy
from .backend import (
    HipBackend, set_backend, get_backend, _dt, _ptr, _no_grad, _inner_contig, _bht, _check, _WORKSPACES, _workspace,
    clear_workspaces, fused_ops_available)
from .policy import (
    POLICY, Policy, _value_blocks, chunk_segments, _LINEAR_SPLIT_MAX_OUT, _LINEAR_SPLIT_MIN_ROWS, _linear_split,
    _MLP_PAD)
from .kernels import (
    _gla_prepare, _head_first_empty, _gla_launch, gla_chunk_bwd, _short_conv_launch, _sum_partials, _sum_partials2,
    column_sum, _sum_vector, _embed_sum_launch, argmax_rows, new_loop_ctl, LOOP_CTL_ROWS, greedy_pick_embed, sample_pick_embed, topk_sample_rows,
    gla_decode_prologue, swiglu, gla_decode_update, _kstep, packed_numel, pack_rows, unpack_rows,
    linear_skinny_packed, linear_skinny, gla_decode_inproj, gla_decode_inproj_packed, gla_decode_update_norm,
    gla_decode_window, gla_decode_window_flush, cross_att_step1, cross_att_step2, cross_scores,
    cross_scores_softmax, softmax_weighted_rows_add, pe_softmax_weighted_rows_add, softmax_pe_rows, softmax_rows, weighted_rows_add, dwconv7_ln,
    istft_ola)
from .autograd import (
    _GLAFunction, _needs_grad, _gla, fused_recurrent_gla, naive_recurrent_gla, chunk_gla, fused_chunk_gla,
    chunk_simple_gla, GradSlab, _slab_part, _SplitSlabFunction, split_slab, _ShortConvFunction, short_conv, _ShortConv3Function, short_conv3,
    _RMSNormGateFunction, _gate_rows_view, rmsnorm_swish_gate, rmsnorm, _LayerNormFunction, _LN_TRIPLES, layer_norm,
    _SwiGLUFunction, swiglu_gate, linear_weight_grad, _LinearFunction, linear, _StackedLinearFunction, stacked_linear, _SwiGLUMLPFunction, _MLP_ONE,
    _mlp_one, swiglu_mlp, clear_mlp_pack, _GateLogSigmoidFunction, gate_logsigmoid, _GateLowRankFunction, gate_lowrank,
    _CrossEntropyFunction, cross_entropy, _EmbedSumFunction, embed_sum)
