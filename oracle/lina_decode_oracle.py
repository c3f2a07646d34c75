"""CPU restatement of the reference's batched decode loop -- TEST INFRASTRUCTURE ONLY.

Plain-PyTorch functions over a *state dict* with the reference's key names; nothing from
lina-speech_amd is imported.  It restates, at T = 1 per step:
    LinaModel.generate_batch            /root/reference/model/modeling_lina.py:111-192
    AttentiveGLA.step / init_state      /root/reference/model/gla.py:302-313,358-365
    MixingBlock / SwiGLU                /root/reference/model/base_blocks.py:42-69
    GatedLinearAttention.forward        /root/reference/model/gla.py:131-227   (mode='naive': the
                                        pure-PyTorch recurrent path, gla.py:196-197)
    BlindCrossAttention.forward, ConvPos /root/reference/model/crossatt.py:21-32,105-155
    TextEncoder / SelfAttention (rotary=False)  model/encoder.py:14-43, base_blocks.py:9-40
It is (a) an independent checker for the product's host glue and fused decode engine and
(b) the ``cpu_baseline`` ("port") that bench.py times on the GPU box's host cores.
PARITY UNPINNED at the fla boundary (see gla_oracle.py); the glue itself is pinned by
tests/golden/lina_d64.npz, which test_oracle.py replays through this file.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import gla_oracle as O


class OracleLina:
    def __init__(self, sd: dict, n_layer: int, heads: int, txt_heads: int = None, n_quant: int = 1,
                 normalizer: float = 16.0, eps: float = 1e-5, dtype=torch.float32, state_dtype=None):
        self.sd = {k: v.detach().to("cpu", dtype if v.is_floating_point() else v.dtype) for k, v in sd.items()}
        self.n_layer, self.H, self.n_quant = n_layer, heads, n_quant
        self.txt_heads = txt_heads or heads
        self.normalizer, self.eps = normalizer, eps
        self.dtype = dtype
        # the dtype the recurrent state is STORED in between decode steps.  The reference allocates it with param.new_zeros
        # (model/gla.py:229-240) and Cache.update copy_-s every step's fp32 final state into it: a bf16 model rounds its state
        # to bf16 after EVERY step.  None = keep it in the arithmetic dtype (what the product's fp32 state does).
        self.state_dtype = state_dtype

    # ---- small pieces -----------------------------------------------------------------
    def _lin(self, x, name, bias=True):
        b = self.sd.get(name + ".bias") if bias else None
        return F.linear(x, self.sd[name + ".weight"], b)

    def _ln(self, x, name):
        return F.layer_norm(x, (x.shape[-1],), self.sd[name + ".weight"], self.sd[name + ".bias"], 1e-5)

    def _swiglu(self, x, p):                                        # base_blocks.py:48-50
        a, b = self._lin(x, p + ".p_in").chunk(2, dim=-1)
        return self._lin(F.silu(a) * b, p + ".p_out")

    def mixer(self, x, p, state):                                   # gla.py:131-227, use_cache=True
        """x [B,T,d]; state = [conv_q, conv_k, conv_v, S] (mutated) or None."""
        B, T, _ = x.shape
        H = self.H
        q, k, v = (self._lin(x, f"{p}.{n}_proj", bias=False) for n in "qkv")
        cq, ck, cv = (state[0], state[1], state[2]) if state is not None else (None, None, None)
        q = O.short_conv(q, self.sd[p + ".q_conv1d.weight"], None, cq)
        k = O.short_conv(k, self.sd[p + ".k_conv1d.weight"], None, ck)
        v = O.short_conv(v, self.sd[p + ".v_conv1d.weight"], None, cv)
        hf = lambda t: t.view(B, T, H, -1).transpose(1, 2)
        gk = self._lin(self._lin(x, p + ".gk_proj.0", bias=False), p + ".gk_proj.1")
        gk = O.gate_logsigmoid(hf(gk), self.normalizer)
        o, S = O.naive_recurrent_gla(hf(q), hf(k), hf(v), gk, initial_state=None if state is None else state[3],
                                     output_final_state=state is not None)
        if state is not None:
            state[3].copy_(S if self.state_dtype is None else S.to(self.state_dtype))
        g = self._lin(x, p + ".g_proj", bias=False).view(B, T, H, -1)
        o = O.rmsnorm_swish_gate(o.transpose(1, 2), g, self.sd[p + ".g_norm_swish_gate.weight"], self.eps)
        return self._lin(o.reshape(B, T, -1), p + ".o_proj", bias=False)

    def block(self, x, p, state):                                   # base_blocks.py:65-69
        x = self.mixer(self._ln(x, p + ".norm1"), p + ".tmix", state) + x
        return self._swiglu(self._ln(x, p + ".norm2"), p + ".cmix") + x

    def text_encoder(self, x_ids):                                  # encoder.py:14-43 (rotary=False, no mask)
        x = self.sd["txt_embed.weight"][x_ids]
        i = 0
        while f"txt_encoder.sa.{i}.norm1.weight" in self.sd:
            p = f"txt_encoder.sa.{i}"
            h = self._ln(x, p + ".norm1")
            B, N, D = h.shape
            q, k, v = self._lin(h, p + ".tmix.qkv").chunk(3, dim=-1)
            hf = lambda t: t.view(B, N, self.txt_heads, -1).transpose(1, 2)
            y = F.scaled_dot_product_attention(hf(q), hf(k), hf(v)).transpose(1, 2).reshape(B, N, D)
            x = y + x
            x = self._swiglu(self._ln(x, p + ".norm2"), p + ".cmix") + x
            i += 1
        return x

    def pos_emb(self, n):                                           # crossatt.py:21-32 ConvPos
        p = "attentive_rnn.cross_att.pos_embed"
        e = self.sd[p + ".embed.weight"][:n].unsqueeze(0).transpose(1, 2)
        w, b = self.sd[p + ".dw_conv.weight"], self.sd[p + ".dw_conv.bias"]
        return F.conv1d(e, w, b, groups=w.shape[0], padding="same").transpose(1, 2)   # [1,n,d]

    def cross_att(self, y, x_enc, state):                           # crossatt.py:105-155 (eval branch)
        p = "attentive_rnn.cross_att"
        q = self._ln(self._lin(y, p + ".q"), p + ".ln_q")
        v = self._ln(self._lin(x_enc, p + ".v"), p + ".ln_v")
        k = self._ln(self._lin(x_enc, p + ".k"), p + ".ln_k")
        pe = self.pos_emb(x_enc.shape[1])
        sc = 1.0 / math.sqrt(q.shape[-1])
        a1 = torch.softmax(q @ k.transpose(-2, -1) * sc, dim=-1)
        x = self.block(a1 @ pe, p + ".pos_net", state)
        a2 = torch.softmax(x @ pe.transpose(-2, -1) * sc, dim=-1)
        return a2 @ v, torch.stack((a1, a2), dim=1)                 # [B,T,d], [B,2,T,Ttxt]

    # ---- state ------------------------------------------------------------------------
    def init_state(self, B):                                        # gla.py:229-240,302-313
        st = []
        for i in range(2 * self.n_layer + 1):
            p = self._block_prefix(i) + ".tmix"
            Kd = self.sd[p + ".q_proj.weight"].shape[0]
            Vd = self.sd[p + ".v_proj.weight"].shape[0]
            W = self.sd[p + ".q_conv1d.weight"].shape[-1]
            z = lambda *s: torch.zeros(*s, dtype=self.dtype)
            st.append([z(B, Kd, W), z(B, Kd, W), z(B, Vd, W), z(B, self.H, Kd // self.H, Vd // self.H)])
        return st

    def _block_prefix(self, i):
        n = self.n_layer
        if i < n:
            return f"attentive_rnn.encoder.{i}"
        if i < 2 * n:
            return f"attentive_rnn.decoder.{i - n}"
        return "attentive_rnn.cross_att.pos_net"

    # ---- decode -----------------------------------------------------------------------
    def step(self, y, x_enc, state):                                # gla.py:358-365 + logits head
        n = self.n_layer
        for i in range(n):
            y = self.block(y, self._block_prefix(i), state[i])
        v, att = self.cross_att(y, x_enc, state[2 * n])
        y = y + v
        for i in range(n, 2 * n):
            y = self.block(y, self._block_prefix(i), state[i])
        W = self.sd["logits_head.weight"]                           # [q,l,d]
        self.last_hidden = y                                        # what the head reads (tests compare it as well as the logits)
        return torch.einsum("bnd,qld->bnql", y, W), att

    def embed(self, tok):                                           # tok [q,B,n] -> [B,n,d]
        return O.embed_sum(self.sd["rvq_embed.weight"], tok)

    @torch.no_grad()
    def generate_greedy(self, x_ids, n_steps, teacher=None):
        """x_ids [B,Ttxt].  Greedy (k=1, first_greedy_quant=0, force_max_seqlen) decode, modeling_lina.py:111-179.
        ``teacher`` [q,B,n_steps]: feed these tokens instead of the picks (teacher forcing).
        Returns tokens [q,B,n], logits [B,n,q,l], atts [B,2,n,Ttxt], margins [B,n] (top-1 minus top-2)."""
        B = x_ids.shape[0]
        x_enc = self.text_encoder(x_ids)
        state = self.init_state(B)
        y = self.embed(torch.ones(self.n_quant, B, 1, dtype=torch.long))
        toks, logits_all, atts, margins = [], [], [], []
        self.hiddens = []                                           # pre-head hidden state of every step, [B,1,d] each
        for t in range(n_steps):
            logits, att = self.step(y, x_enc, state)
            self.hiddens.append(self.last_hidden)
            pick = O.argmax_lowest(logits[:, 0].transpose(0, 1)).unsqueeze(-1)        # [q,B,1]
            top2 = logits[:, 0, 0].float().topk(2, dim=-1).values
            margins.append(top2[:, 0] - top2[:, 1])
            toks.append(pick)
            logits_all.append(logits)
            atts.append(att)
            y = self.embed(pick if teacher is None else teacher[:, :, t:t + 1])
        self.final_state = state
        return torch.cat(toks, 2), torch.cat(logits_all, 1), torch.cat(atts, 2), torch.stack(margins, 1)
