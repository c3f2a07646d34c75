"""DecodeEngine: the decode loop of ``LinaModel.generate_batch`` (reference model/modeling_lina.py:111-192) and its
single-token step (``AttentiveGLA.step`` + logits head, model/gla.py:358-365) restructured for MI355X.

Per GLA block FIVE launches instead of ~30 --
  1 LayerNorm-1 (folded) + fused projection q|k|v|g|gate-low-rank + 3 conv steps + gate   (lina_gla_decode_inproj[_packed])
  2 recurrent-state update + RMSNorm (x) swish gate: K1w, the windowed (lazily written) state inside the device loop
    (lina_gla_decode_window), or the immediate row-split update K1d + K5 for a single step  (lina_gla_decode_update_norm)
  3 o_proj + residual                                                                     (lina_linear_skinny[_ex])
  4 up-projection with LayerNorm-2 folded in + bias + SwiGLU                              (lina_linear_skinny[_ex])
  5 down-projection (bias as a constant-1 column) + residual                              (lina_linear_skinny[_ex])
-- the blind cross-attention is 4 more launches around its pos_net block, the head one, the token epilogue one (K6d / K6e:
picks, token log, stop bookkeeping, next-token embedding, step counter): 71 launches per token, captured in hipGraphs of 1 and 8
tokens.  From 160 utterance rows up the projections of 1 / 4 / head run on the tall tiling (csrc/linear_tall.h).

``generate()`` is what ``LinaModel.generate_batch`` runs by default: tokens, attention log and stop flags stay on the device, 8
bytes of the loop-control block are read back every ``stop_check_every`` steps through pinned memory behind the queued work.
The text side of the cross-attention is projected once (BlindCrossAttention.prepare); state lives in the caller-visible
``Cache`` tensors (reference layout; ``state`` / ``sync_state()`` materialise the pending window steps on demand);
``reset()`` re-arms a built engine for the next utterance batch.
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch
import torch.nn.functional as F

from . import ops
from .mixer import GatedLinearAttention
from .modules import Cache


def _fold_layernorm(weight, ln, bias=None):
    """LN(x) @ W^T == rstd * (x @ W'^T - mu * c1) + c2  with  W' = gamma (*) W  (lina_linear_skinny)."""
    dt = weight.dtype
    w_ln = (weight.float() * ln.weight.float()[None, :]).to(dt).contiguous()
    c1 = w_ln.float().sum(1).contiguous()
    c2 = weight.float() @ ln.bias.float()
    if bias is not None:
        c2 = c2 + bias.float()
    return w_ln, c1, c2.contiguous()


# hipGraph capture next to a live process group (bench.py --gpus N, DDP): RCCL's watchdog thread polls its events while
# this thread captures -- legal only when the capture does not claim the whole process ("thread_local": other threads' runtime
# calls neither join nor invalidate the capture).  Nothing the engine captures depends on another thread's work.
_CAPTURE_MODE = "thread_local"


class _capture_guard:
    """No garbage collection while a stream is capturing.  An engine is a reference cycle (its loop bodies close over it), so a
    dropped engine -- and the hipGraphs it owns -- is destroyed whenever Python's cyclic collector happens to run; if that moment
    falls inside another engine's stream capture, ``hipGraphDestroy`` fails with "operation not permitted when stream is
    capturing" inside a C++ destructor and the process aborts (seen in round 5: bench.py building its fifth engine).  Collect
    first, outside the capture, then keep the collector off until the capture has ended."""

    def __enter__(self):
        import gc
        gc.collect()
        self._was = gc.isenabled()
        gc.disable()
        return self

    def __exit__(self, *exc):
        import gc
        if self._was:
            gc.enable()
        return False



class _BlockPack:
    """Decode-time weights of one MixingBlock(GatedLinearAttention, SwiGLU, LayerNorm)."""

    _SHARED = ("H", "Dk", "Dv", "Kd", "Vd", "d", "R", "normalizer", "clamp_min", "eps_gate", "n1_eps", "n2_eps",
               "w_in", "c1_in", "c2_in", "ldz", "off_q", "off_k", "off_v", "off_g", "off_lr", "wq", "wk", "wv",
               "w2", "b2", "gnw", "w_o", "hid", "hid_pad", "w_up", "c1_up", "c2_up", "w_down",
               "w_in_p", "w_o_p", "w_up_p", "w_down_p", "up_half_rows")

    def __init__(self, blk, state, lo=0, hi=None, shared=None, window=1):
        self.window = window
        if shared is not None:                     # same block, another row range: reuse the packed weights
            for name in self._SHARED:
                setattr(self, name, getattr(shared, name))
            self._buffers(state, lo, hi, shared.w_in.dtype, shared.w_in.device)
            return
        m: GatedLinearAttention = blk.tmix
        if not (m.use_short_conv and not m.share_conv_kernel and m.conv_size == 4 and not m.conv_bias
                and m.fuse_norm_and_gate):
            raise NotImplementedError("DecodeEngine needs use_short_conv=True, conv_size=4, no conv bias, "
                                      "fused swish norm gate (the 'convblind_shortconv' configuration)")
        dt = m.q_proj.weight.dtype
        dev = m.q_proj.weight.device
        kq = 32 if dt == torch.bfloat16 else 16                         # k-step of lina_linear_skinny
        self.H, self.Dk, self.Dv, self.Kd, self.Vd = m.num_heads, m.head_qk_dim, m.head_v_dim, m.key_dim, m.value_dim
        self.d = m.hidden_size
        if self.d % kq or self.Vd % kq:
            raise NotImplementedError(f"hidden/value dims must be multiples of {kq}")
        self.R = m.gk_proj[0].weight.shape[0]
        self.normalizer, self.clamp_min = float(m.gate_logit_normalizer), m.clamp_min
        self.eps_gate = m.g_norm_swish_gate.eps
        self.n1_eps, self.n2_eps = blk.norm1.eps, blk.norm2.eps
        # fused projection: rows = q | k | v | g | low-rank gate, LayerNorm-1 folded in
        w_cat = torch.cat([m.q_proj.weight, m.k_proj.weight, m.v_proj.weight, m.g_proj.weight,
                           m.gk_proj[0].weight], dim=0)
        self.w_in, self.c1_in, self.c2_in = _fold_layernorm(w_cat, blk.norm1)
        self.ldz = w_cat.shape[0]
        self.off_q, self.off_k, self.off_v = 0, self.Kd, 2 * self.Kd
        self.off_g, self.off_lr = 2 * self.Kd + self.Vd, 2 * self.Kd + 2 * self.Vd
        self.wq = m.q_conv1d.weight.reshape(self.Kd, 4).contiguous()
        self.wk = m.k_conv1d.weight.reshape(self.Kd, 4).contiguous()
        self.wv = m.v_conv1d.weight.reshape(self.Vd, 4).contiguous()
        self.w2, self.b2 = m.gk_proj[1].weight.contiguous(), m.gk_proj[1].bias.contiguous()
        self.gnw = m.g_norm_swish_gate.weight.contiguous()
        self.w_o = m.o_proj.weight.contiguous()                         # [d, Vd]
        c = blk.cmix
        self.hid = c.hidden
        self.hid_pad = (self.hid + 1 + kq - 1) // kq * kq               # + bias column, whole k-steps
        self.w_up, self.c1_up, self.c2_up = _fold_layernorm(c.p_in.weight, blk.norm2, c.p_in.bias)
        w_down = torch.zeros(c.p_out.weight.shape[0], self.hid_pad, dtype=dt, device=dev)
        w_down[:, :self.hid] = c.p_out.weight
        w_down[:, self.hid] = c.p_out.bias                              # multiplied by the constant-1 column
        self.w_down = w_down
        # fragment-major copies for the device-side loop (include/lina_gla.h "packed operands"): one contiguous 1 KiB
        # per MFMA fragment load instead of 16 rows x 64 B
        self.w_in_p = ops.pack_rows(self.w_in)
        self.w_o_p = ops.pack_rows(self.w_o)
        # each half zero-padded to whole 64-row blocks covering the hid_pad output columns the launch sweeps (the K-padding
        # columns beyond `hid` read weight rows too: with hid % 64 == 0 they would lie past a half padded to ceil64(hid))
        self.up_half_rows = (self.hid_pad + 63) // 64 * 64
        halves = []
        for h in (self.w_up[:self.hid], self.w_up[self.hid:]):
            hp = torch.zeros(self.up_half_rows, h.shape[1], dtype=dt, device=dev)
            hp[:self.hid] = h
            halves.append(ops.pack_rows(hp))
        self.w_up_p = torch.cat(halves)
        self.w_down_p = ops.pack_rows(self.w_down)
        self._buffers(state, lo, hi, dt, dev)

    def _buffers(self, state, lo, hi, dt, dev):
        if state[3].dtype not in (torch.float32, torch.bfloat16) or not state[3].is_contiguous():
            raise ValueError("recurrent state must be a contiguous fp32 (opt-in: bf16) tensor (see GatedLinearAttention.init_state)")
        hi = state[3].shape[0] if hi is None else hi
        self.cq, self.ck, self.cv, self.S = (t[lo:hi] for t in state)    # row slices stay contiguous
        B = hi - lo
        self.row_split = self.Dk % 64 == 0 and self.Dv in (64, 128, 256)
        self.fused_in = self.R == 16 and self.Kd % 16 == 0 and self.Vd % 16 == 0
        self.z = torch.empty(B, self.ldz, dtype=dt, device=dev)
        self.g = torch.empty(B, self.Vd, dtype=dt, device=dev)
        self.qkv = torch.empty(B, 2 * self.Kd + self.Vd, dtype=dt, device=dev)
        self.gk = torch.empty(B, self.Kd, dtype=torch.float32, device=dev)
        self.o_part = torch.empty(max(self.Dk // 64, 1), B, self.H, self.Dv, dtype=torch.float32, device=dev)
        self.og = torch.empty(B, self.H, self.Dv, dtype=dt, device=dev)
        self.counters = torch.zeros(B * self.H, dtype=torch.int32, device=dev)
        self.s = torch.empty(B, self.hid_pad, dtype=dt, device=dev)
        # (a bf16 state always takes K1w, at window 1 too: the immediate kernels K1d / K1 are built for an fp32 state)
        self.lazy = ((self.window > 1 or state[3].dtype == torch.bfloat16) and self.Dk in (64, 128, 256)
                     and self.Dv in (64, 128, 256, 512))
        if state[3].dtype == torch.bfloat16 and not self.lazy:
            raise NotImplementedError("bf16 recurrent state: head shapes of K1w only")
        self.o_x = torch.zeros(B * self.H * self.Dv, dtype=torch.float32, device=dev) if self.Dv > 256 else None
        self.packed = self.lazy and self.fused_in
        if self.packed:
            self.og_p = torch.zeros(ops.packed_numel(B, self.Vd), dtype=dt, device=dev)
            self.s_p = torch.zeros(ops.packed_numel(B, self.hid_pad), dtype=dt, device=dev)
        if self.lazy:      # K1w: k_s, cumulative log-gate c_s and v_s of the steps of the current window
            self.hk = torch.zeros(self.window, B * self.H, self.Dk, dtype=torch.float32, device=dev)
            self.hc = torch.zeros(self.window, B * self.H, self.Dk, dtype=torch.float32, device=dev)
            self.hv = torch.zeros(self.window, B * self.H, self.Dv, dtype=torch.float32, device=dev)


class _Part:
    """Rows [lo, hi) of the batch: own residual/workspace buffers and state slices, shared weights."""

    def __init__(self, lo, hi, packs, kk, vv, d, dtype, dev):
        self.lo, self.hi, self.packs = lo, hi, packs
        self.kk, self.vv = kk, vv
        self.x = torch.zeros(hi - lo, d, dtype=dtype, device=dev)       # residual stream
        self.xp = torch.zeros(hi - lo, d, dtype=dtype, device=dev)      # pos_net stream
        self.x_p = torch.zeros(ops.packed_numel(hi - lo, d), dtype=dtype, device=dev)    # fragment-major copies
        self.xp_p = torch.zeros(ops.packed_numel(hi - lo, d), dtype=dtype, device=dev)
        self.q_lin = torch.zeros(hi - lo, d, dtype=dtype, device=dev)   # projected cross-attention query
        Tn = kk.shape[1]
        Tp = (Tn + 31) // 32 * 32
        self.scores = torch.zeros(hi - lo, Tn, dtype=torch.float32, device=dev)
        self.sc2 = torch.zeros(hi - lo, Tp, dtype=dtype, device=dev)
        self.attc = torch.zeros(hi - lo, Tp, dtype=dtype, device=dev)  # softmax rows, zero-padded to whole k-steps


class _Loop:
    """One captured configuration of the device-side decode loop: its logs, control block and hipGraphs."""
    __slots__ = ("cap", "tok_log", "att_log", "att_direct", "ctl", "body", "graph1", "graphN", "att", "hid_log")

    def __init__(self):
        self.cap, self.tok_log, self.att_log, self.att_direct, self.ctl, self.hid_log = 0, None, None, False, None, None
        self.body = self.graph1 = self.graphN = self.att = None


class DecodeEngine:
    def __init__(self, model, x_enc: torch.Tensor, batch_size: int, state: Optional[Cache] = None,
                 use_graph: Optional[bool] = None, n_split: Optional[int] = None, fuse_norm: bool = True,
                 window: Optional[int] = None, stream_weights=("in", "up"), cross: str = "spread",
                 fused_pick: bool = True, packed: bool = True, cross_tail_fused: bool = True,
                 share_weights_with: Optional["DecodeEngine"] = None, state_dtype: Optional[torch.dtype] = None):
        """``share_weights_with``: another engine of the same model whose packed decode-time weights this one reuses
        (DecodeEngineGroup: several engines on row ranges of one batch).
        Every variant of the step is a constructor argument (rounds 2-3 read ``LINA_DECODE_*`` environment switches here).
        ``stream_weights``: which weight matrices of the device loop ("in", "o", "up", "down", "head") are loaded with the
        non-temporal hint instead of competing for the 256 MB Infinity Cache (measured optimum: in + up).
        ``cross``: first half of the cross-attention -- "spread" = scores on 256 workgroups + {softmax, att1 . pe} in one launch,
        "fused" = the round-2 form {scores + softmax in one workgroup per row} + skinny GEMM.  ``fused_pick`` / ``packed``: the
        one-launch sampled epilogue K6e / the fragment-major operand path of the device loop (False = the unfused forms, kept
        for A/B measurements and as the parity reference of the fused ones).  ``cross_tail_fused``: the second attention's
        scores x_pos . pe^T computed inside the softmax + att2 . V + residual launch (one launch less per token).
        ``window`` (1, 2, 4, 8 or 16; default 8): the device-side decode loop keeps the recurrent state of every block
        LAZILY WRITTEN -- read every token, rewritten every ``window``-th token (K1w, lina_gla_decode_window); the
        steps in between live in small history buffers.  ``engine.state`` / ``sync_state()`` materialise the exact
        state on demand.  1 = the immediate in-place update K1d on every token.
        ``state_dtype`` (opt-in; default fp32): ``torch.bfloat16`` keeps the recurrent state of every block in bf16, as the
        REFERENCE does for a bf16 model (model/gla.py:229-240: ``init_state`` allocates with ``param.new_zeros`` and
        ``Cache.update`` copies the fp32 final state of every step into it): read, updated in fp32 registers, rounded when
        written back -- on every token with ``window=1`` (the reference's arithmetic; the default window for this dtype), on
        every ``window``-th token otherwise (less rounding than the reference, half of K1w's bytes).  bf16 models only; the
        fp32 state stays the product's default and the headline's.
        ``n_split`` > 1 cuts the batch into independent row ranges that run on parallel HIP streams inside
        the same graph: the step is a chain of ~100 short dependent launches, so two (or four) independent
        chains in flight hide each other's launch/drain latency; rows never interact."""
        rnn = model.attentive_rnn
        self.model = model
        self.fuse_norm = fuse_norm
        self.B = batch_size
        self.dev = x_enc.device
        if state_dtype not in (None, torch.float32, torch.bfloat16):
            raise ValueError("state_dtype must be torch.float32 or torch.bfloat16")
        self.state_dtype = state_dtype or torch.float32
        bf16_state = self.state_dtype == torch.bfloat16
        if bf16_state and (model.logits_head.weight.dtype != torch.bfloat16 or not fuse_norm):
            raise ValueError("a bf16 recurrent state is the reference's arithmetic for a bf16 model: bf16 weights only")
        if window is None:
            window = 1 if bf16_state else 8
        if window not in (1, 2, 4, 8, 16):
            raise ValueError("window must be 1, 2, 4, 8 or 16")
        self.window = window if self.fuse_norm else 1
        self._state = state if state is not None else rnn.init_state(batch_size=batch_size)
        if bf16_state:                                 # the Cache the caller sees then holds bf16 states, like the reference's
            conv = Cache()
            for li, st in enumerate(self._state.states):
                conv.update(tuple(st[:-1]) + (st[-1].to(torch.bfloat16).contiguous(),), li, offset=0)
            self._state = conv
        self._t_idx = torch.zeros(1, dtype=torch.long, device=self.dev)      # device step counter of the greedy loop
        self._origin = torch.zeros(1, dtype=torch.long, device=self.dev)     # step at which the current window began
        self._origin_host, self._n_done, self._lazy_live = 0, 0, False
        self._skip_update = False
        self._loop_packed = False
        self._loops, self._loop, self._att_direct, self._t0 = {}, None, None, 0
        self._pin = self._pin_ev = None
        self._pick_counter = torch.zeros(1, dtype=torch.int32, device=self.dev)
        # which weight matrices of the device loop are STREAMED (non-temporal loads) instead of competing for the 256 MB
        # Infinity Cache: per token the loop touches 0.87 GB of state (always streamed) + 0.26 GB of weights; streaming
        # the largest matrices lets the others stay resident between two tokens
        self._stream = set(stream_weights)
        if not self._stream <= {"in", "o", "up", "down", "head"}:
            raise ValueError(f"stream_weights: unknown weight set(s) {sorted(self._stream - {'in', 'o', 'up', 'down', 'head'})}")
        if cross not in ("spread", "fused"):
            raise ValueError("cross must be 'spread' or 'fused'")
        self._cross_spread = cross == "spread"
        self._fused_pick, self._packed_ok = bool(fused_pick), bool(packed)
        self._cross_tail_fused = bool(cross_tail_fused)   # x_pos . pe^T folded into the softmax + att2 . V launch (d % 256 == 0)
        blocks = list(rnn.encoder) + list(rnn.decoder) + [rnn.cross_att.pos_net]
        self.n_enc = len(rnn.encoder)
        ca = rnn.cross_att
        self.ca = ca
        kk, vv, pe = ca.prepare(x_enc)                                   # [B,1,Ttxt,d] x2, [1,1,Ttxt,d]
        kk, vv = kk.squeeze(1).contiguous(), vv.squeeze(1).contiguous()
        if kk.shape[0] != batch_size:                                    # one text for every row
            kk, vv = kk.expand(batch_size, -1, -1).contiguous(), vv.expand(batch_size, -1, -1).contiguous()
        self._kk, self._vv = kk, vv                                      # static: reset(x_enc) rewrites them in place
        self.pe = pe.squeeze(1).squeeze(0).contiguous()                  # [Ttxt, d]
        self.Tn = self.pe.shape[0]
        Tp = (self.Tn + 31) // 32 * 32
        self.pe_pad = torch.zeros(Tp, self.pe.shape[1], dtype=self.pe.dtype, device=self.pe.device)
        self.pe_pad[:self.Tn] = self.pe                                   # weight rows of  scores2 = xp . pe^T
        self.peT = self.pe_pad.t().contiguous()                           # [d, Tp]: weight rows of  xp = att1 . pe
        self.att_scale = 1.0 / math.sqrt(kk.shape[-1])
        self.ca_qw, self.ca_qb = ca.q.weight.contiguous(), ca.q.bias.float().contiguous()
        hw = model.logits_head.weight
        self.Q, self.L, self.d = hw.shape
        self.w_head = hw.reshape(self.Q * self.L, self.d).contiguous()
        self.w_head_p = ops.pack_rows(self.w_head)
        self.ca_qw_p = ops.pack_rows(self.ca_qw)
        self.pe_pad_p = ops.pack_rows(self.pe_pad)
        self.use_graph = (self.dev.type == "cuda") if use_graph is None else use_graph
        if n_split is None:
            # measured on MI355X (B=64): 1 range 1.035 ms/step, 2 ranges 1.013 ms, 4 ranges 1.64 ms -- the forked
            # branches of a hipGraph barely overlap, so one range stays the default
            n_split = 1
        n_split = max(1, min(n_split, batch_size))
        from .shard import shard_rows
        self.parts = []
        first = None if share_weights_with is None else share_weights_with.packs
        for i in range(n_split):
            lo, hi = shard_rows(batch_size, i, n_split)
            packs = [_BlockPack(b, self._state[j], lo, hi, shared=None if first is None else first[j],
                                window=self.window) for j, b in enumerate(blocks)]
            first = first or packs
            self.parts.append(_Part(lo, hi, packs, kk[lo:hi], vv[lo:hi], self.d, hw.dtype, self.dev))
        self.packs = self.parts[0].packs
        self._streams = ([torch.cuda.Stream(device=self.dev) for _ in self.parts]
                         if (len(self.parts) > 1 and self.dev.type == "cuda") else None)
        self._graph = None
        self._y_in = torch.zeros(batch_size, self.d, dtype=hw.dtype, device=self.dev)
        self._logits = torch.zeros(batch_size, self.Q * self.L, dtype=hw.dtype, device=self.dev)
        self._att = torch.zeros(batch_size, 2, 1, kk.shape[1], dtype=hw.dtype, device=self.dev)

    # ------------------------------------------------------------------ one GLA block, T = 1 (7 launches)
    def _block(self, x, P: _BlockPack, lazy: bool = False, x_p=None):
        """x [B,d] is the residual stream and is UPDATED IN PLACE.  ``lazy``: windowed state update K1w (device loop);
        ``x_p``: the fragment-major copy of x -- the projections then run on packed operands and keep it current."""
        B = x.shape[0]
        packed = x_p is not None
        if packed:
            ops.gla_decode_inproj_packed(x_p, P.w_in_p, B, P.d, P.c1_in, P.c2_in, P.wq, P.wk, P.wv, P.cq, P.ck, P.cv,
                                         P.w2, P.b2, P.qkv, P.g, P.gk, P.n1_eps, P.normalizer, P.clamp_min,
                                         w_stream="in" in self._stream)
            gate = P.g.view(B, P.H, P.Dv)
        elif P.fused_in:
            ops.gla_decode_inproj(x, P.w_in, P.c1_in, P.c2_in, P.wq, P.wk, P.wv, P.cq, P.ck, P.cv, P.w2, P.b2,
                                  P.qkv, P.g, P.gk, P.n1_eps, P.normalizer, P.clamp_min)
            gate = P.g.view(B, P.H, P.Dv)
        else:
            z = ops.linear_skinny(x, P.w_in, P.c1_in, P.c2_in, out=P.z, ln_dim=P.d, ln_eps=P.n1_eps)
            ops.gla_decode_prologue(z, P.off_q, P.off_k, P.off_v, P.off_lr, P.wq, P.wk, P.wv, P.cq, P.ck, P.cv,
                                    P.w2, P.b2, P.qkv, P.gk, P.normalizer, P.clamp_min)
            gate = z[:, P.off_g:P.off_g + P.Vd].view(B, P.H, P.Dv)
        q = P.qkv[:, :P.Kd].view(B, P.H, P.Dk)
        k = P.qkv[:, P.Kd:2 * P.Kd].view(B, P.H, P.Dk)
        v = P.qkv[:, 2 * P.Kd:].view(B, P.H, P.Dv)
        if self._skip_update:
            pass                                  # measurement only (time_update_kernel): the step without K1w / K1d
        elif (lazy and P.lazy) or P.S.dtype == torch.bfloat16:
            # (a bf16 state outside the device loop: the same kernel as an immediate update, window 1)
            ops.gla_decode_window(q, k, v, P.gk.view(B, P.H, P.Dk), P.S, gate, P.gnw, P.og_p if packed else P.og,
                                  P.hk, P.hc, P.hv, self._t_idx, self._origin, P.window if lazy else 1, P.eps_gate,
                                  og_packed=packed, o_exchange=P.o_x, counters=P.counters)
        elif P.row_split and self.fuse_norm:
            ops.gla_decode_update_norm(q, k, v, P.gk.view(B, P.H, P.Dk), P.o_part, P.S, gate, P.gnw, P.og,
                                       P.counters, P.eps_gate)
        elif P.row_split:
            ops.gla_decode_update(q, k, v, P.gk.view(B, P.H, P.Dk), P.o_part, P.S)
            ops.rmsnorm_swish_gate(P.o_part, gate, P.gnw, P.eps_gate, n_partial=P.o_part.shape[0], out=P.og)
        else:
            o, _ = ops.fused_recurrent_gla(q.unsqueeze(2), k.unsqueeze(2), v.unsqueeze(2),
                                           P.gk.view(B, P.H, 1, P.Dk), initial_state=P.S,
                                           output_final_state=True, inplace_state=True)
            ops.rmsnorm_swish_gate(o.reshape(B, P.H, P.Dv), gate, P.gnw, P.eps_gate, out=P.og)
        if packed:
            # the residual stream lives in x_p only (read-modify-write there); the row-major x is stale inside the loop
            ops.linear_skinny_packed(P.og_p, P.w_o_p, B, P.d, P.Vd, resid=x_p, out_packed=x_p, out_packed_width=P.d,
                                     w_stream="o" in self._stream)
            ops.linear_skinny_packed(x_p, P.w_up_p, B, P.hid_pad, P.d, P.c1_up, P.c2_up, out_packed=P.s_p,
                                     out_packed_width=P.hid_pad, swiglu_hidden=P.hid, ln_dim=P.d, ln_eps=P.n2_eps,
                                     w_half_rows=P.up_half_rows, w_stream="up" in self._stream)
            ops.linear_skinny_packed(P.s_p, P.w_down_p, B, P.d, P.hid_pad, resid=x_p, out_packed=x_p,
                                     out_packed_width=P.d, w_stream="down" in self._stream)
            return x
        ops.linear_skinny(P.og.view(B, P.Vd), P.w_o, resid=x, out=x)
        ops.linear_skinny(x, P.w_up, P.c1_up, P.c2_up, out=P.s, swiglu_hidden=P.hid, ln_dim=P.d, ln_eps=P.n2_eps,
                          n_out=P.hid_pad)
        ops.linear_skinny(P.s, P.w_down, resid=x, out=x)
        return x

    def _cross(self, part, x, lazy=False, packed=False):
        """x += blind cross-attention: 5 short launches around the pos_net block (query projection, scores + softmax,
        att1.pe | xp.pe^T, softmax + att2.V + residual)."""
        ca = self.ca
        att = self._att[part.lo:part.hi]
        B = x.shape[0]
        log = self._att_direct                 # the loop's att log [B,2,cap,Ttxt]: rows filed at the device step index
        if log is not None:
            att = log.att_log[part.lo:part.hi]
            log_kw = dict(att_step=self._t_idx, att_step_stride=log.att_log.stride(2), att_steps=log.cap)
        else:
            log_kw = {}
        if packed:
            q_lin = ops.linear_skinny_packed(part.x_p, self.ca_qw_p, B, self.d, self.d, c2=self.ca_qb, out=part.q_lin)
        else:
            q_lin = ops.linear_skinny(x, self.ca_qw, c2=self.ca_qb, out=part.q_lin)
        if self._cross_spread:
            # scores on 256 workgroups (4 per row), then softmax + att1 . pe in one launch (K = T_txt: no MFMA needed)
            ops.cross_scores(q_lin, ca.ln_q.weight, ca.ln_q.bias, ca.ln_q.eps, part.kk, part.scores, self.att_scale)
            ops.softmax_pe_rows(part.scores, att[:, 0, 0], self.pe, part.xp, part.xp_p if packed else None, **log_kw)
        else:
            ops.cross_scores_softmax(q_lin, ca.ln_q.weight, ca.ln_q.bias, ca.ln_q.eps, part.kk, att[:, 0, 0], part.attc,
                                     self.att_scale)
        fuse2 = self._cross_tail_fused and self.d % 256 == 0
        if packed:
            if not self._cross_spread:
                ops.linear_skinny(part.attc, self.peT, out=part.xp, out_packed=part.xp_p, out_packed_width=self.d)  # xp = att1 . pe
            self._block(part.xp, part.packs[-1], lazy, part.xp_p)
            if not fuse2:
                ops.linear_skinny_packed(part.xp_p, self.pe_pad_p, B, self.pe_pad.shape[0], self.d, out=part.sc2)
        else:
            if not self._cross_spread:
                ops.linear_skinny(part.attc, self.peT, out=part.xp)               # xp = att1 . pe
            self._block(part.xp, part.packs[-1], lazy)
            if not fuse2:
                ops.linear_skinny(part.xp, self.pe_pad, out=part.sc2)             # scores2 = xp . pe^T
        if fuse2:
            # scores2 = xp . pe^T, softmax, att2 . V and the residual add in ONE launch (round 4; was a projection launch + this)
            ops.pe_softmax_weighted_rows_add(part.xp_p if packed else part.xp, self.pe_pad, self.att_scale, att[:, 1, 0],
                                             part.vv, x, x_packed=part.x_p if packed else None, xp_is_packed=packed,
                                             **log_kw)
        else:
            ops.softmax_weighted_rows_add(part.sc2, self.att_scale, att[:, 1, 0], part.vv, x,
                                          x_packed=part.x_p if packed else None)

    def _core_part(self, part, y, lazy=False, packed=False):
        x = part.x
        if y is not x:                       # the device-side loop embeds the next token straight into part.x
            x.copy_(y[part.lo:part.hi])
        x_p = part.x_p if packed else None   # packed: part.x_p already holds x (pick kernel / begin_greedy)
        for P in part.packs[:self.n_enc]:
            self._block(x, P, lazy, x_p)
        self._cross(part, x, lazy, packed)
        for P in part.packs[self.n_enc:-1]:
            self._block(x, P, lazy, x_p)
        if packed:
            ops.linear_skinny_packed(x_p, self.w_head_p, x.shape[0], self.Q * self.L, self.d,
                                     out=self._logits[part.lo:part.hi], w_stream="head" in self._stream)
        else:
            ops.linear_skinny(x, self.w_head, out=self._logits[part.lo:part.hi])

    def _core(self, y, lazy=False, packed=False):
        """y [B,d] -> (logits [B,1,Q,L], att [B,2,1,Ttxt]) written into the engine's static buffers."""
        if self._streams is None:
            for part in self.parts:
                self._core_part(part, y, lazy, packed)
        else:
            main = torch.cuda.current_stream(self.dev)
            for part, st in zip(self.parts, self._streams):       # fork
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    self._core_part(part, y, lazy, packed)
            for st in self._streams:                                # join
                main.wait_stream(st)
        return self._logits.view(self.B, 1, self.Q, self.L), self._att

    # ------------------------------------------------------------------ lazily written state (K1w)
    @property
    def state(self):
        """The per-layer Cache (reference layout).  Reading it materialises the pending window steps first."""
        self.sync_state()
        return self._state

    def sync_state(self):
        """Apply the pending steps of the current window to the recurrent states (no-op when nothing is pending)
        and start a new window at the current step."""
        if not self._lazy_live:
            return
        pending = (self._n_done - self._origin_host) % self.window
        if pending:
            for P in self._all_packs():
                if P.lazy:
                    ops.gla_decode_window_flush(P.S, P.hk, P.hc, P.hv, pending)
        self._origin_host = self._n_done
        self._origin.fill_(self._n_done)

    # ------------------------------------------------------------------ measurement
    @torch.inference_mode()
    def time_update_kernel(self, reps: int = 160):
        """Duration of the recurrent-update kernel (K1w + K5, or K1d + K5) IN SITU: the captured step graph is timed
        with and without its update launches (everything else identical); the difference divided by the number of
        update launches is what one launch adds to the step -- its own run time between the kernels that really
        surround it.  (Timing the kernel back to back with itself measures something else: every launch then starts
        streaming while its predecessor's tail is still draining.)  Leaves the state untouched.
        Returns (seconds per launch, launches per step, ms with, ms without)."""
        if not self.use_graph:
            raise RuntimeError("time_update_kernel needs the hipGraph path (a ROCm device)")
        self.sync_state()
        lazy = self.window > 1 or self.state_dtype == torch.bfloat16
        snap = self._snapshot()
        x_keep = [part.x.clone() for part in self.parts]
        t_keep, o_keep, live = self._t_idx.clone(), self._origin.clone(), self._lazy_live
        y = self.parts[0].x if len(self.parts) == 1 else self._y_in
        times = []
        for skip in (False, True):
            self._skip_update = skip
            self._t_idx.zero_()
            self._origin.zero_()
            side = torch.cuda.Stream(device=self.dev)
            side.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(side):
                self._core(y, lazy, self._loop_packed)
            torch.cuda.current_stream(self.dev).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with _capture_guard(), torch.cuda.graph(g, capture_error_mode=_CAPTURE_MODE):
                self._core(y, lazy, self._loop_packed)
                self._t_idx.add_(1)                       # walks through the window positions
            for _ in range(16):
                g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) * 1e-3 / reps)
        self._skip_update = False
        self._restore(snap)
        for part, xk in zip(self.parts, x_keep):
            part.x.copy_(xk)
        self._t_idx.copy_(t_keep)
        self._origin.copy_(o_keep)
        self._lazy_live = live
        n = len(self._all_packs())
        return (times[0] - times[1]) / n, n, times[0] * 1e3, times[1] * 1e3

    # ------------------------------------------------------------------ graph capture
    def _all_packs(self):
        return [P for part in self.parts for P in part.packs]

    @staticmethod
    def _live_buffers(P):
        return (P.cq, P.ck, P.cv, P.S) + ((P.hk, P.hc, P.hv) if P.lazy else ())   # + the window history of K1w

    def _snapshot(self):
        return [[t.clone() for t in self._live_buffers(P)] for P in self._all_packs()]

    def _restore(self, snap):
        for P, saved in zip(self._all_packs(), snap):
            for dst, src in zip(self._live_buffers(P), saved):
                dst.copy_(src)

    def _capture(self):
        snap = self._snapshot()
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            for _ in range(2):                                           # warm hipBLASLt workspaces / autotune
                self._core(self._y_in)
        torch.cuda.current_stream(self.dev).wait_stream(side)
        self._restore(snap)
        g = torch.cuda.CUDAGraph()
        with _capture_guard(), torch.cuda.graph(g, capture_error_mode=_CAPTURE_MODE):
            self._core(self._y_in)
        self._graph = g

    def __call__(self, y_embd: torch.Tensor, t: int = 0):
        """One token for every row: y_embd [B,1,d] -> (logits [B,1,Q,L], att [B,2,1,Ttxt])."""
        y = y_embd.reshape(self.B, self.d)
        self.sync_state()                     # the generic step updates the state immediately (K1d)
        if not self.use_graph:
            logits, att = self._core(y)
            return logits, att.clone()
        if self._graph is None:
            self._capture()
        self._y_in.copy_(y)
        self._graph.replay()
        return self._logits.view(self.B, 1, self.Q, self.L), self._att.clone()

    step = __call__

    # ------------------------------------------------------------------ fully device-side decode loop
    def begin_greedy(self, max_steps: int, y0: Optional[torch.Tensor] = None, k: int = 1, temp: float = 1.0,
                     seed: int = 0, first_greedy_quant: int = 0, log_att: bool = False, t0: int = 0,
                     log_hidden: bool = False):
        """Arm the device-side decode loop: token picks, the stop bookkeeping, the next-token embedding (K6a), the token
        log and -- with ``log_att`` -- the attention log are part of the captured step, so one token == one graph replay
        and nothing is read back until ``greedy_tokens()``.  Quantizers ``i < first_greedy_quant`` are SAMPLED (top-``k``,
        temperature ``temp``, K6c with uniforms hashed from (seed, device step counter, row)) like the reference's
        default generation mode (modeling_lina.py:159-164); the others -- all of them by default -- take the arg-max
        (K6b).  ``log_hidden`` (parity tests): the residual stream the codec head reads -- the pre-head hidden state [B,d] of
        every step -- is filed in a log as well (one more copy per token; ``logged_hidden()``): a check on it is sensitive to
        recurrent-state error where the logits of a peaked head are dominated by the embedding -> head shortcut.
        ``t0``: the step index the loop starts at (a prompt prefill has produced steps 0 .. t0-1; ``preload``
        puts their tokens / attention rows into the logs).  The recurrent state is NOT reset (``reset()`` does that).

        A configuration (sampling mode, att log, operand layout) is captured ONCE per engine and kept (``self._loops``):
        arming it again only rewrites the control block (stop flags, per-call seed word), the step counters and the
        first input -- no new graph capture."""
        emb = self.model.rvq_embed
        self.sync_state()                     # pending window steps of an earlier loop
        if y0 is None:
            y0 = emb.embed_sum(torch.ones(self.Q, self.B, 1, dtype=torch.long, device=self.dev))
        self._y_in.copy_(y0.reshape(self.B, self.d))
        lazy = self.window > 1 or self.state_dtype == torch.bfloat16     # (K1w: also the bf16 state's immediate update, window 1)
        n_sampled = min(max(first_greedy_quant, 0), self.Q) if k > 1 else 0
        # fragment-major operands: the device loop on one row range with the one-launch token epilogue (K6d / K6e keep
        # x_p current; the unfused sampled epilogue, fused_pick=False, has no packed output)
        fused_pick = self.Q <= 16 and (n_sampled == 0 or self._fused_pick)
        packed = (lazy and len(self.parts) == 1 and fused_pick
                  and all(P.packed for P in self.packs) and self._packed_ok)
        if n_sampled == 0:
            k, temp = 1, 1.0
        key = (n_sampled, int(k), float(temp), bool(log_att), packed, fused_pick, 0 if fused_pick else int(seed),
               bool(log_hidden))
        loop = self._loops.pop(key, None)
        if loop is None or loop.cap < max_steps:
            # a configuration owns its logs (att log: B x 2 x cap x T_txt, ~134 MB at B = 512) and two hipGraphs; callers that
            # sweep k / temp (captured kernel arguments) or, on the unfused pick path, the seed would otherwise grow the set
            # without bound: keep the MAX_LOOPS most recently used (dropped here, outside any stream capture)
            if loop is not None:
                self._drop_loop(loop)                        # (too short: rebuilt with longer logs)
            while len(self._loops) >= self.MAX_LOOPS:
                self._drop_loop(self._loops.pop(next(iter(self._loops))))
            loop = self._build_loop(key, max_steps, lazy)
        self._loops[key] = loop                              # most recently used last
        self._loop = loop
        self._loop_packed = packed
        self._lazy_live = lazy
        # ---- arm: control block, counters, first input
        loop.tok_log.zero_()
        loop.ctl.copy_(ops.new_loop_ctl(self.B, "cpu", int(seed) if fused_pick else 0))
        self._pick_counter.zero_()
        self._t_idx.fill_(t0)
        self._origin.fill_(t0)
        self._n_done, self._origin_host, self._t0 = t0, t0, t0
        y_buf = self._loop_input()
        y_buf.copy_(self._y_in)
        if packed:
            ops.pack_rows(y_buf, out=self.parts[0].x_p)

    def _loop_input(self):
        """One row range: the residual-stream buffer itself is the step's input (no y -> x copy, no embed -> y copy)."""
        return self.parts[0].x if len(self.parts) == 1 else self._y_in

    def _build_loop(self, key, max_steps: int, lazy: bool):
        n_sampled, k, temp, log_att, packed, fused_pick, seed, log_hidden = key
        emb = self.model.rvq_embed
        L = _Loop()
        L.cap = (max(int(max_steps), 1) + 63) // 64 * 64
        L.tok_log = torch.zeros(L.cap, self.Q, self.B, dtype=torch.long, device=self.dev)
        L.ctl = ops.new_loop_ctl(self.B, self.dev)
        hw_dt = self._att.dtype
        L.att_log = torch.zeros(self.B, 2, L.cap, self.Tn, dtype=hw_dt, device=self.dev) if log_att else None
        # the two cross-attention launches of the default step write their rows straight into the log at the device step
        # index; the other forms of the step write the engine's static [B,2,1,Ttxt] buffer and one index_copy_ files it
        L.att_direct = bool(log_att and self._cross_spread and self._cross_tail_fused and self.d % 256 == 0)
        if log_hidden:
            if len(self.parts) != 1:
                raise ValueError("log_hidden needs the engine on one row range")
            hid_src = self.parts[0].x_p if packed else self.parts[0].x      # the operand of the head projection
            L.hid_log = torch.zeros(L.cap, hid_src.numel(), dtype=hid_src.dtype, device=self.dev)
        is_sampled = (torch.arange(self.Q, device=self.dev) < n_sampled).unsqueeze(0)        # [1,Q]
        y_buf = self._loop_input()
        R = ops.LOOP_CTL_ROWS

        def body():
            self._att_direct = L if L.att_direct else None
            try:
                logits, att = self._core(y_buf, lazy, packed)
            finally:
                self._att_direct = None
            if L.att_log is not None and not L.att_direct:
                L.att_log.index_copy_(2, self._t_idx, att)
            if L.hid_log is not None:          # before the token epilogue overwrites the stream with the next token's embedding
                L.hid_log.index_copy_(0, self._t_idx, (self.parts[0].x_p if packed else self.parts[0].x).reshape(1, -1))
            lg = logits.view(self.B, self.Q, self.L)
            if n_sampled == 0 and self.Q <= 16:
                # K6d: picks, token log, stop flags, next-token embedding and the step counter in ONE launch
                ops.greedy_pick_embed(lg, emb.weight, y_buf, L.tok_log, self._t_idx, self._pick_counter,
                                      x_packed=self.parts[0].x_p if packed else None, loop_ctl=L.ctl)
                return att
            if fused_pick:
                # K6e: the same epilogue with the first n_sampled quantizers drawn by top-k / temperature sampling
                ops.sample_pick_embed(lg, emb.weight, y_buf, L.tok_log, self._t_idx, self._pick_counter, n_sampled, k,
                                      temp, seed=0, x_packed=self.parts[0].x_p if packed else None, loop_ctl=L.ctl)
                return att
            if n_sampled == 0:
                pick = ops.argmax_rows(lg)
            elif n_sampled == self.Q:
                pick = ops.topk_sample_rows(lg, k, temp, seed=seed, step=self._t_idx)
            else:
                pick = torch.where(is_sampled, ops.topk_sample_rows(lg, k, temp, seed=seed, step=self._t_idx),
                                   ops.argmax_rows(lg))
            pick = pick.t().contiguous()                                                      # [Q,B]
            L.tok_log.index_copy_(0, self._t_idx, pick.unsqueeze(0))
            # the stop bookkeeping of K6d / K6e as torch ops on the same control block (reference modeling_lina.py:168-173)
            rows = L.ctl[R:R + self.B]
            rows.copy_(torch.maximum(rows, (pick == 2).all(dim=0).to(torch.int32)))
            L.ctl[0:1].copy_(rows.sum().to(torch.int32).view(1))
            first = (L.ctl[0:1] >= self.B) & (L.ctl[1:2] < 0)
            L.ctl[1:2].copy_(torch.where(first, self._t_idx.to(torch.int32), L.ctl[1:2]))
            self._t_idx.add_(1)
            ops.embed_sum(emb.weight, pick, out=y_buf)              # next step's input, written in place
            return att

        L.body = body
        if self.use_graph:
            snap, y_keep = self._snapshot(), y_buf.clone()
            xp_keep = self.parts[0].x_p.clone() if packed else None
            t_keep, o_keep = self._t_idx.clone(), self._origin.clone()
            self._t_idx.zero_()
            self._origin.zero_()
            side = torch.cuda.Stream(device=self.dev)
            side.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(side):
                for _ in range(2):
                    body()
            torch.cuda.current_stream(self.dev).wait_stream(side)
            self._restore(snap)
            y_buf.copy_(y_keep)
            if packed:
                self.parts[0].x_p.copy_(xp_keep)
            self._t_idx.copy_(t_keep)
            self._origin.copy_(o_keep)
            self._pick_counter.zero_()
            g = torch.cuda.CUDAGraph()
            with _capture_guard(), torch.cuda.graph(g, capture_error_mode=_CAPTURE_MODE):
                L.att = body()
            L.graph1 = g
        return L

    MAX_LOOPS = 4            # captured loop configurations kept per engine (least recently used dropped first)

    @staticmethod
    def _drop_loop(L):
        """A loop's body closes over the loop object (a reference cycle): cut it, so that its logs and hipGraphs go NOW -- outside
        any stream capture -- and not whenever the cyclic collector runs."""
        L.body = L.graph1 = L.graphN = L.att = L.tok_log = L.att_log = L.ctl = L.hid_log = None

    def close(self):
        """Release the engine's device memory and hipGraphs now.  An engine is a reference cycle (its loop bodies close over
        it), so dropping the last reference frees nothing until a later pass of Python's cyclic collector -- at B = 512 that
        is ~7 GB of state and logs lingering after a cache eviction.  The engine is unusable afterwards."""
        for L in self._loops.values():
            self._drop_loop(L)
        self._loops.clear()
        self._loop = self._graph = self._att_direct = None
        for part in self.parts:
            part.packs = []
        self.parts, self.packs, self._state = [], [], None
        self._kk = self._vv = self._logits = self._att = self._y_in = None
        self.model = None

    # names older callers / tests look at
    @property
    def _greedy_graph(self):
        return None if self._loop is None else self._loop.graph1

    @property
    def _greedy_graph_n(self):
        return None if self._loop is None else self._loop.graphN

    @property
    def _tok_log(self):
        return self._loop.tok_log

    def preload(self, tokens: torch.Tensor, atts: Optional[torch.Tensor] = None):
        """File the steps a prefill produced in front of the loop: ``tokens [Q,B,t0]`` (and ``atts [B,2,t0,Ttxt]``) go
        into the logs at 0 .. t0-1 and the control block learns which rows have already emitted the stop token --
        everything ``begin_greedy(t0=...)`` needs to continue as if it had run those steps itself."""
        L = self._loop
        t0 = tokens.shape[2]
        if t0 != self._t0:
            raise ValueError("preload: the number of steps must equal begin_greedy's t0")
        L.tok_log[:t0].copy_(tokens.permute(2, 0, 1))
        if L.att_log is not None and atts is not None:
            L.att_log[:, :, :t0].copy_(atts)
        is_stop = (tokens == 2).all(dim=0)                                     # [B,t0]
        seen = is_stop.cummax(dim=1).values                                     # row has stopped at or before step t
        all_seen = seen.all(dim=0)                                              # [t0]
        R = ops.LOOP_CTL_ROWS
        L.ctl[R:R + self.B].copy_(seen[:, -1].to(torch.int32))
        L.ctl[0:1].copy_(seen[:, -1].sum().to(torch.int32).view(1))
        first = torch.where(all_seen.any(), all_seen.to(torch.int32).argmax().to(torch.int32),
                            torch.full((), -1, dtype=torch.int32, device=tokens.device))
        L.ctl[1:2].copy_(first.view(1))

    def stop_step(self) -> int:
        """First step at which every row had emitted the stop token (the step the reference's loop breaks at), -1 if
        that has not happened.  Reads 4 bytes back (host sync)."""
        return int(self._loop.ctl[1])

    def greedy_step(self):
        """Enqueue one token for every row (no host sync). Returns the step's attention weights
        (a static buffer: clone to keep; stale when the loop files them into its att log itself)."""
        self._n_done += 1
        L = self._loop
        if L.graph1 is not None:
            L.graph1.replay()
            return L.att
        return L.body()

    GRAPH_STEPS = 8          # tokens per replay of the multi-step graph (greedy_steps)

    def greedy_steps(self, n: int):
        """Enqueue ``n`` tokens for every row.  Whole groups of GRAPH_STEPS tokens go as ONE graph replay (a replay has a
        fixed cost of its own -- a ~9 us kernel-argument copy on the stream plus the host launch -- that a one-token
        graph pays per token); the remainder token by token.  The window position of K1w and the token log index come
        from the device step counter, so the same captured graph is valid at any position."""
        N = self.GRAPH_STEPS
        L = self._loop
        if L.graph1 is not None and n >= N:
            if L.graphN is None:                             # captured on first use (stream capture executes nothing)
                g = torch.cuda.CUDAGraph()
                with _capture_guard(), torch.cuda.graph(g, capture_error_mode=_CAPTURE_MODE):
                    for _ in range(N):
                        L.body()
                L.graphN = g
            while n >= N:
                L.graphN.replay()
                self._n_done += N
                n -= N
        for _ in range(n):
            self.greedy_step()

    def greedy_tokens(self):
        """Tokens produced so far: [Q,B,n]."""
        return self._loop.tok_log[:self._n_done].permute(1, 2, 0).clone(memory_format=torch.contiguous_format)

    def logged_atts(self, n: Optional[int] = None):
        """The attention log of a loop armed with ``log_att``: [B,2,n,Ttxt]."""
        n = self._n_done if n is None else n
        return self._loop.att_log[:, :, :n].clone(memory_format=torch.contiguous_format)   # never the static log itself (n == cap)

    def logged_hidden(self, n: Optional[int] = None):
        """The pre-head hidden states of a loop armed with ``log_hidden``: [n, B, d] (row-major, whatever the loop's layout)."""
        n = self._n_done if n is None else n
        L = self._loop
        if self._loop_packed:
            return torch.stack([ops.unpack_rows(L.hid_log[t], self.B, self.d) for t in range(n)])
        return L.hid_log[:n].view(n, self.B, self.d).clone()

    # ------------------------------------------------------------------ engine reuse
    def reset(self, x_enc: Optional[torch.Tensor] = None, state: Optional[Cache] = None):
        """Back to the start of an utterance batch: recurrent states and conv caches zeroed (or copied from ``state``, a
        Cache of the reference layout), and -- with ``x_enc`` -- the text side of the cross-attention recomputed into the
        engine's static buffers.  The captured graphs stay valid: they only know buffer addresses."""
        if x_enc is not None:
            if x_enc.shape[1] != self.Tn:
                raise ValueError("reset: the text length is part of the captured step (build another engine)")
            kk, vv, _ = self.ca.prepare(x_enc)
            self._kk.copy_(kk.squeeze(1))
            self._vv.copy_(vv.squeeze(1))
        self._lazy_live = False                       # pending window steps of the previous utterances are dropped
        for li, st in enumerate(self._state.states):
            for j, dst in enumerate(st):
                if state is None:
                    dst.zero_()
                else:
                    dst.copy_(state.states[li][j])
        for P in self._all_packs():
            P.counters.zero_()
        self._n_done = self._origin_host = 0
        self._t_idx.zero_()
        self._origin.zero_()

    @torch.inference_mode()
    def generate(self, max_seqlen: int, y0: Optional[torch.Tensor] = None, k: int = 1, temp: float = 1.0,
                 first_greedy_quant: int = 0, seed: int = 0, force_max_seqlen: bool = False,
                 stop_check_every: int = 16, log_att: bool = True, preload=None):
        """The loop of the reference's ``generate_batch`` (model/modeling_lina.py:152-179) on the device: up to
        ``max_seqlen`` steps in replays of GRAPH_STEPS tokens; every ``stop_check_every`` steps 8 bytes of the control
        block are copied to pinned host memory BEHIND the queued work, and the copy issued one check earlier is looked at
        -- the GPU always has the next group of steps queued, the host never waits for the step it has just enqueued.
        When the block says that every row has stopped, the loop ends; up to two groups of steps may have run past that
        point, and the logs are trimmed to the exact length the reference's per-step check produces.
        ``preload`` = (tokens [Q,B,t0], atts [B,2,t0,Ttxt]) of a prompt prefill.  Returns (qs [Q,B,n], atts [B,2,n,Ttxt]
        or None, n)."""
        t0 = 0 if preload is None else int(preload[0].shape[2])
        self.begin_greedy(max_seqlen, y0, k=k, temp=temp, seed=seed, first_greedy_quant=first_greedy_quant,
                          log_att=log_att, t0=t0)
        if preload is not None:
            self.preload(*preload)
        L = self._loop
        total = max(max_seqlen - t0, 0)
        if force_max_seqlen:
            self.greedy_steps(total)
            n = t0 + total
        else:
            N = self.GRAPH_STEPS
            every = max(int(stop_check_every), 1)
            every = (every + N - 1) // N * N if L.graph1 is not None else every
            stop_at, done, prev = -1, 0, None
            if preload is not None:
                stop_at = self.stop_step()                  # the prefill may already have seen every row stop
            cuda = self.dev.type == "cuda"
            if cuda and self._pin is None:
                self._pin = [torch.empty(2, dtype=torch.int32).pin_memory() for _ in range(2)]
                self._pin_ev = [torch.cuda.Event() for _ in range(2)]
            while stop_at < 0 and done < total:
                n_now = min(every, total - done)
                self.greedy_steps(n_now)
                done += n_now
                if not cuda:
                    stop_at = self.stop_step()
                    continue
                if prev is not None:                        # the check issued BEFORE the group just enqueued
                    self._pin_ev[prev].synchronize()
                    stop_at = int(self._pin[prev][1])
                    if stop_at >= 0:
                        break
                if done < total:
                    slot = 1 if prev == 0 else 0
                    self._pin[slot].copy_(L.ctl[:2], non_blocking=True)
                    self._pin_ev[slot].record()
                    prev = slot
            if stop_at < 0:
                stop_at = self.stop_step()
            n = min(stop_at + 1, t0 + done) if stop_at >= 0 else t0 + done
        # fresh tensors, like the reference's: `.contiguous()` is a no-op -- i.e. a view of the engine's static log that the next
        # call on the cached engine overwrites -- whenever n == cap (max_seqlen a multiple of 64 and no early stop)
        qs = L.tok_log[:n].permute(1, 2, 0).clone(memory_format=torch.contiguous_format)
        atts = L.att_log[:, :, :n].clone(memory_format=torch.contiguous_format) if L.att_log is not None else None
        return qs, atts, n

    def poll_stop(self, slot: int):
        """Queue a copy of the control block's first two words to pinned slot ``slot`` behind the work enqueued so far."""
        if self._pin is None:
            self._pin = [torch.empty(2, dtype=torch.int32).pin_memory() for _ in range(2)]
            self._pin_ev = [torch.cuda.Event() for _ in range(2)]
        self._pin[slot].copy_(self._loop.ctl[:2], non_blocking=True)
        self._pin_ev[slot].record()

    def polled_stop(self, slot: int) -> int:
        """Wait for the copy queued by ``poll_stop(slot)`` and return the stop step it carried (-1: not every row has stopped)."""
        self._pin_ev[slot].synchronize()
        return int(self._pin[slot][1])

    @torch.inference_mode()
    def run_greedy(self, n_steps: int, y0: Optional[torch.Tensor] = None, record_att: bool = False, **sampling):
        """Decode ``n_steps`` tokens on the device (greedy unless ``k``/``temp``/``seed``/``first_greedy_quant``
        say otherwise, see begin_greedy). Returns tokens [Q,B,n_steps] (and atts [B,2,n,Ttxt])."""
        self.begin_greedy(n_steps, y0, log_att=record_att, **sampling)
        for _ in range(n_steps):
            self.greedy_step()
        toks = self.greedy_tokens()
        self.sync_state()
        return (toks, self.logged_atts(n_steps)) if record_att else toks


class DecodeEngineGroup:
    """Several DecodeEngines on contiguous row ranges of ONE utterance batch, each driven on its own HIP stream.

    The rows of a batch never interact (reference model/modeling_lina.py:125,152-179: one state and one token stream per row), and
    at a large batch the step of one engine is two thirds K1w -- HBM-bound -- and one third projections that leave HBM idle: two
    half-batch engines whose graph replays sit in different hardware queues run one half's projections under the other half's
    state update.  Measured at L169, 512 rows (tools/probe_two_engines.py, profiles/r05_two_engines.txt): one engine 2.34 ms per
    token, 2 x 256 rows on two streams 2.22 ms (sequentially 2.80), 4 x 128 rows 2.43.  The packed weights are shared; every
    engine has its own state, logs, control block and captured graphs.  ``generate`` is DecodeEngine.generate over the group: the
    loop ends when EVERY engine's control block says all of its rows have stopped, and the logs are trimmed at the last of those
    steps -- the step the reference's single loop breaks at.  No codec-prompt preload (the caller uses one engine for that)."""

    def __init__(self, model, x_enc: torch.Tensor, batch_size: int, n_engines: int = 2, **engine_args):
        from .shard import shard_rows
        if n_engines < 2 or n_engines > batch_size:
            raise ValueError("DecodeEngineGroup needs 2 <= n_engines <= batch_size")
        self.B, self.dev = batch_size, x_enc.device
        self.ranges = [shard_rows(batch_size, i, n_engines) for i in range(n_engines)]
        self.engines = []
        for lo, hi in self.ranges:
            xe = x_enc[lo:hi] if x_enc.shape[0] == batch_size else x_enc
            self.engines.append(DecodeEngine(model, xe, batch_size=hi - lo,
                                             share_weights_with=self.engines[0] if self.engines else None, **engine_args))
        self.Q, self.Tn = self.engines[0].Q, self.engines[0].Tn
        self.streams = ([torch.cuda.Stream(device=self.dev) for _ in self.engines] if self.dev.type == "cuda" else None)
        self._n_done = 0

    def close(self):
        for e in self.engines:
            e.close()
        self.engines = []

    def _each(self, fn):
        """fn(engine, lo, hi) for every engine, each on its own stream (forked from / before the caller's current stream)."""
        if self.streams is None:
            return [fn(e, lo, hi) for e, (lo, hi) in zip(self.engines, self.ranges)]
        main = torch.cuda.current_stream(self.dev)
        out = []
        for e, (lo, hi), st in zip(self.engines, self.ranges, self.streams):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                out.append(fn(e, lo, hi))
        return out

    def _join(self):
        if self.streams is not None:
            main = torch.cuda.current_stream(self.dev)
            for st in self.streams:
                main.wait_stream(st)

    def reset(self, x_enc: Optional[torch.Tensor] = None):
        self._each(lambda e, lo, hi: e.reset(None if x_enc is None else (x_enc[lo:hi] if x_enc.shape[0] == self.B else x_enc)))
        self._join()

    def begin_greedy(self, max_steps: int, y0: Optional[torch.Tensor] = None, seed: int = 0, **kw):
        # (another seed word per engine: the sampler hashes (seed, step, LOCAL row))
        self._each(lambda e, lo, hi: e.begin_greedy(max_steps, None if y0 is None else y0[lo:hi],
                                                    seed=(seed + 0x9E3779B97F4A7C15 * self.engines.index(e)) & (2 ** 63 - 1), **kw))
        self._n_done = 0

    def greedy_steps(self, n: int):
        """``n`` tokens on every engine, enqueued ALTERNATELY in groups of GRAPH_STEPS: a hipGraph launch costs the host a
        fraction of a millisecond, so enqueueing one engine's whole run first would start the other one that much later (the
        first form of generate_batch(n_engines=2) ran 750 steps 15 % slower than the bare loop for that reason)."""
        N = DecodeEngine.GRAPH_STEPS
        if self.streams is None:
            for e in self.engines:
                e.greedy_steps(n)
        else:
            main = torch.cuda.current_stream(self.dev)
            for st in self.streams:
                st.wait_stream(main)
            k = 0
            while k < n:
                m = min(N, n - k)
                for e, st in zip(self.engines, self.streams):
                    with torch.cuda.stream(st):
                        e.greedy_steps(m)
                k += m
        self._n_done += n

    def greedy_tokens(self):
        self._join()
        return torch.cat([e._loop.tok_log[:self._n_done].permute(1, 2, 0) for e in self.engines], dim=1).contiguous()

    @torch.inference_mode()
    def generate(self, max_seqlen: int, y0: Optional[torch.Tensor] = None, k: int = 1, temp: float = 1.0,
                 first_greedy_quant: int = 0, seed: int = 0, force_max_seqlen: bool = False, stop_check_every: int = 16,
                 log_att: bool = True):
        """As DecodeEngine.generate (no preload).  Returns (qs [Q,B,n], atts [B,2,n,Ttxt] or None, n)."""
        self.begin_greedy(max_seqlen, y0, seed=seed, k=k, temp=temp, first_greedy_quant=first_greedy_quant, log_att=log_att)
        total, done = max(int(max_seqlen), 0), 0
        stops = [-1] * len(self.engines)
        if force_max_seqlen:
            self.greedy_steps(total)
            done = total
        else:
            N = DecodeEngine.GRAPH_STEPS
            every = max(int(stop_check_every), 1)
            every = (every + N - 1) // N * N if self.engines[0]._loop.graph1 is not None else every
            prev = None
            while min(stops) < 0 and done < total:
                n_now = min(every, total - done)
                self.greedy_steps(n_now)
                done += n_now
                if self.streams is None:
                    stops = [e.stop_step() for e in self.engines]
                    continue
                if prev is not None:                        # the checks issued BEFORE the group of steps just enqueued
                    stops = [e.polled_stop(prev) for e in self.engines]
                    if min(stops) >= 0:
                        break
                if done < total:
                    slot = 1 if prev == 0 else 0
                    self._each(lambda e, lo, hi: e.poll_stop(slot))
                    prev = slot
        self._join()
        if not force_max_seqlen:
            stops = [e.stop_step() for e in self.engines]
        n = done
        if not force_max_seqlen and min(stops) >= 0:
            n = min(max(stops) + 1, done)                   # every engine's rows have stopped: the last of those steps
        qs = torch.cat([e._loop.tok_log[:n].permute(1, 2, 0) for e in self.engines], dim=1).contiguous()
        atts = (torch.cat([e._loop.att_log[:, :, :n] for e in self.engines], dim=0).contiguous()
                if self.engines[0]._loop.att_log is not None else None)
        return qs, atts, n
