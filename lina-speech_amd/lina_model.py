"""LinaModel: text embed + codec embed -> AttentiveRNN -> codec logits, with the
teacher-forced ``forward`` and the batched autoregressive ``generate_batch`` of the
reference (model/modeling_lina.py:14-192; same constructor, argument names, returns and
state-dict keys ``txt_embed / rvq_embed / logits_head / attentive_rnn / txt_encoder``).

MI355X-first differences that do not change results:
  * the text side of the cross-attention is projected once per utterance (prepare());
  * ``generate_batch`` runs the device-side loop by default (decode.DecodeEngine.generate: fused HIP step, 8 tokens per
    hipGraph replay): picks, stop bookkeeping, the attention log and the next-token embedding stay on the device, and the
    "all rows stopped" test is read back every ``stop_check_every`` steps instead of every step (the loop may run up to two
    such groups of extra steps; the returned tensors are trimmed to the exact length the reference would have produced);
  * the engine is cached per (batch, text length, weights) and re-armed for later calls (``clear_decode_cache``).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn
from torch import Tensor

from . import ops
from .attentive import AttentiveRNN
from .codec import CodecHead, MultiEmbedding, topk_sampling, undelay_rvq


class LinaModel(nn.Module):
    def __init__(self, attentive_rnn: AttentiveRNN, d_model: int, n_quant: int, n_codebook: int,
                 n_special_token_in: int, n_special_token_out: int, n_txt_vocab: int, tie_embed: bool = False,
                 txt_encoder: Optional[nn.Module] = None, spk_encoder: Optional[nn.Module] = None,
                 mask_text_p: float = 0.0):
        super().__init__()
        if mask_text_p > 0.0:
            raise NotImplementedError("mask_text_p > 0 raises in the reference too (SURVEY App. D)")
        self.n_quant, self.n_codebook = n_quant, n_codebook
        self.n_special_token_in, self.n_special_token_out = n_special_token_in, n_special_token_out
        self.mask_text_p = mask_text_p
        self.n_txt_vocab = n_txt_vocab
        self.n_target_vocab = n_codebook + n_special_token_out
        self.txt_encoder, self.spk_encoder, self.attentive_rnn = txt_encoder, spk_encoder, attentive_rnn
        self.txt_embed = nn.Embedding(n_txt_vocab, d_model, padding_idx=0)
        self.rvq_embed = MultiEmbedding(n_quant, n_codebook + n_special_token_in, d_model, padding_idx=0)
        self.logits_head = CodecHead(n_quant, self.n_target_vocab, d_model)
        if tie_embed:
            self.logits_head.weight = self.rvq_embed.weight

    # ------------------------------------------------------------------ teacher-forced
    def forward(self, x, y, encoder_mask, crossatt_mask, logits_mask=None, attention_only=False,
                forced_attention=None, init_state=None, crossatt_pos=None, return_masked: bool = True):
        """Reference modeling_lina.py:72-108.  ``return_masked=False`` (ours): skip the two gathered outputs -- their
        boolean indexing has a data-dependent shape, i.e. a host sync, which a captured (hipGraph) train step cannot
        contain; the loss does not need them."""
        x_embd = self.txt_embed(x)
        # the LAST token's embedding is only ever read by the speaker encoder: without one, embed y[:, :-1] -- the values the
        # reference slices out of the full embedding (modeling_lina.py:80-84), minus the slice's copy forward and its
        # zero-fill + copy backward over [b, n, d]
        n_in = y.shape[1] - 1
        y_src = y if self.spk_encoder is not None else y[:, :-1]
        y_embd = self.rvq_embed(y_src.permute(2, 0, 1))               # 'b n q -> q b n' -> sum over q
        y_embd = y_embd.squeeze(0) if y_embd.shape[0] == 1 else y_embd.sum(0)   # (one quantizer: the sum of one term is a copy -- skipped)
        x_enc = self.txt_encoder(x_embd, mask=encoder_mask)
        if self.spk_encoder is not None:
            y_embd[:, 0] = self.spk_encoder(y_embd)
            y_embd = y_embd[:, :-1, :]
        y_hat, att = self.attentive_rnn(
            y_embd, x_enc, mask=crossatt_mask[:, :-1],
            forced_attention=None if forced_attention is None else forced_attention[:, :, :n_in],
            attention_only=attention_only, init_state=init_state, crossatt_pos=crossatt_pos)
        if attention_only:
            return att
        logits = self.logits_head(y_hat)
        target = y[:, 1:]
        if logits_mask is not None:
            keep = logits_mask[:, 1:]
            masked_logits = masked_target = None
            if return_masked:
                masked_logits, masked_target = logits[keep, :, :], target[keep, :]  # returned, as the reference does
            # the loss itself: masked positions carry the ignored class instead of being gathered out -- the same mean
            # over the same terms (reference modeling_lina.py:97-107), without the scatter of the gather in backward
            target = torch.where(keep.unsqueeze(-1), target, torch.ones_like(target))
        else:
            masked_logits, masked_target = logits, target
        loss = ops.cross_entropy(logits.reshape(-1, logits.shape[-1]), target.reshape(-1), ignore_index=1)   # K14
        return logits, loss, att, masked_logits, masked_target

    # ------------------------------------------------------------------ batched decode
    _ENGINE_CACHE_SIZE = 2
    # generate_batch(n_engines=None): from this many rows up the batch is decoded by TWO engines on two HIP streams (rows never
    # interact -- reference modeling_lina.py:125,152-179; measured at L169, ms per token as one / two engines: 512 rows 2.31 /
    # 2.16, 384 rows 1.82 / 1.74, 256 rows 1.38 / 1.36, 128 rows 0.875 / 0.852; 4 x 128 rows and every split of 64 rows are slower
    # than one engine: profiles/r06_two_engines.txt, r06_b64_engines.txt)
    AUTO_TWO_ENGINES_ROWS = 384

    def __getstate__(self):
        """copy.deepcopy / pickling: the cached decode engines (hipGraphs, static buffers) stay with the original."""
        st = self.__dict__.copy()
        st.pop("_decode_engines", None)
        return st

    def clear_decode_cache(self):
        """Drop the cached decode engines (packed weights, static buffers, captured hipGraphs) NOW.  Rarely needed for
        correctness: the cache key carries every parameter's (storage, version) AND a content fingerprint
        (``_weights_fingerprint``), so optimizer steps, ``load_state_dict``, writes through ``param.data`` and in-place
        writes to inference-mode parameters build a new engine by themselves."""
        for eng in self.__dict__.pop("_decode_engines", {}).values():
            eng.close()

    _FINGERPRINT_ELEMS = 4096

    def _weights_fingerprint(self):
        """Two numbers per parameter -- the L2 norm of its first ``_FINGERPRINT_ELEMS`` elements and of their positive part (the
        second one tells a sign flip apart) -- from multi-tensor kernels over views and one 8-byte-per-parameter read-back: the
        (storage, version) part of the cache key does not see ``param.data`` writes (EMA swaps, hand-edited weights) or in-place
        writes to parameters made under ``torch.inference_mode`` (no version counter), and a stale engine would silently decode
        with the old packed weights.  (A write that leaves the first 4096 elements of EVERY tensor, their storage and their
        version counters untouched is still missed: ``clear_decode_cache()`` after such surgery.)"""
        ps = [p.detach().reshape(-1)[:self._FINGERPRINT_ELEMS] for p in self.parameters()]
        if not ps:
            return ()
        stats = torch._foreach_norm(ps) + torch._foreach_norm(torch._foreach_clamp_min(ps, 0))
        return tuple(torch.stack(stats).float().cpu().tolist())

    def _decode_engine(self, x_enc: Tensor, B: int, init_state, n_engines: int = 1, state_dtype=None):
        """The DecodeEngine of (batch size, text length, dtype, device, current weights), built once and re-armed for
        every later ``generate_batch`` call of the same shape: construction packs 0.3 GB of weights and captures two
        hipGraphs (~0.6 k kernel nodes), far more than a call at B = 64 should pay.  "Current weights" = every
        parameter's (storage, version) AND a content fingerprint (see ``_weights_fingerprint``)."""
        from .decode import DecodeEngine, DecodeEngineGroup
        w = self.logits_head.weight
        key = (B, int(n_engines), int(x_enc.shape[1]), w.dtype, str(w.device), str(state_dtype),
               tuple((p.data_ptr(), -1 if p.is_inference() else p._version) for p in self.parameters()),
               self._weights_fingerprint())
        cache = self.__dict__.setdefault("_decode_engines", {})
        eng = cache.pop(key, None)
        if eng is None:
            if n_engines > 1:                                        # (no init_state / prompt in this form: the caller checked)
                eng = DecodeEngineGroup(self, x_enc, batch_size=B, n_engines=n_engines, state_dtype=state_dtype)
            else:
                eng = DecodeEngine(self, x_enc, batch_size=B, state_dtype=state_dtype)   # NotImplementedError: architecture not covered
                if init_state is not None:
                    eng.reset(state=init_state)
        elif n_engines > 1:
            eng.reset(x_enc)
        else:
            eng.reset(x_enc, state=init_state)
        cache[key] = eng                                             # most recently used last
        while len(cache) > self._ENGINE_CACHE_SIZE:
            cache.pop(next(iter(cache))).close()                     # an engine is a reference cycle: free its device memory now
        return eng

    @torch.inference_mode()
    def generate_batch(self, x: Tensor, batch_size: int = 3, prompt: Optional[Tensor] = None, device: str = "cpu",
                       max_seqlen: int = 1000, k: int = 100, first_greedy_quant: int = 1, temp: float = 1.0,
                       init_state=None, force_max_seqlen: bool = False, stop_check_every: int = 16,
                       engine: Optional[str] = None, seed: Optional[int] = None, n_engines: Optional[int] = None,
                       state_dtype: Optional[torch.dtype] = None):
        """Reference model/modeling_lina.py:111-192 (same arguments, same four returns).  ``engine``:
          None / "auto" -- the device-side loop (decode.DecodeEngine.generate: one hipGraph replay per 8 tokens, picks /
                           stop flags / attention log / next-token embedding inside the graph) when the architecture is
                           one it covers, else the module path;
          "loop"        -- the device-side loop or an error;   "fused" -- the fused step, one graph replay per token, picks
                           on the host side;   "module" -- ``AttentiveGLA.step`` + logits head per token (unfused).
        ``n_engines`` > 1 (device loop, no codec prompt, no init_state): the batch is cut into that many row ranges, one
        engine and one HIP stream each (decode.DecodeEngineGroup: at B = 512 two engines decode 6-8 % faster than one: one
        half's projections run under the other half's HBM-bound state update).  None (default): 2 from
        ``AUTO_TWO_ENGINES_ROWS`` rows up, else 1.  Greedy tokens do not depend on it (rows never interact, every kernel's
        per-row sums are independent of the row count); the sampled quantizers draw from a per-engine seed word.
        ``state_dtype`` (device loop, bf16 models; opt-in): ``torch.bfloat16`` keeps the recurrent state in bf16 and rounds it
        after every step, as the reference itself does for a bf16 model (model/gla.py:229-240 + Cache.update); default fp32.
        ``seed`` feeds the device-side sampler of the loop (default: drawn from torch's generator, so
        ``torch.manual_seed`` makes a call reproducible, like the reference's multinomial)."""
        B, Q = batch_size, self.n_quant
        x = (x.unsqueeze(0).expand(B, -1) if x.dim() == 1 else x).to(device)   # 1-D: one text for every row
        x_enc = self.txt_encoder(self.txt_embed(x))
        y_embd = self.rvq_embed.embed_sum(torch.ones(Q, B, 1, dtype=torch.long, device=device))

        p_len = -1
        if prompt is not None:
            if prompt.shape[1] != B:
                prompt = prompt.expand(Q, B, -1) + 3
            prompt = self.rvq_embed.embed_sum(prompt.to(device))
            p_len = prompt.shape[1]
            if self.spk_encoder is not None:
                prompt[:, 0] = self.spk_encoder(prompt)

        mode = engine or "auto"
        if mode not in ("auto", "loop", "fused", "module"):
            raise ValueError("engine must be None, 'auto', 'loop', 'fused' or 'module'")
        eng = None
        if mode in ("auto", "loop"):
            if n_engines is None:
                n_engines = 2 if B >= self.AUTO_TWO_ENGINES_ROWS else 1
            n_eng = n_engines if (n_engines > 1 and prompt is None and init_state is None and B >= 2 * n_engines) else 1
            try:
                eng = self._decode_engine(x_enc, B, init_state, n_eng, state_dtype)
            except NotImplementedError:
                if mode == "loop":
                    raise
                mode = "module"
        if eng is not None and not hasattr(eng, "state"):            # a group of engines: no prompt, nothing to prefill
            if seed is None:
                seed = int(torch.randint(0, 2 ** 62, (1,)))
            qs, atts, n = eng.generate(max_seqlen, y_embd, k=k, temp=temp, first_greedy_quant=first_greedy_quant, seed=seed,
                                       force_max_seqlen=force_max_seqlen, stop_check_every=stop_check_every, log_att=True)
            return self._finish_generate(qs, atts, (qs == 2).all(dim=0), B, device)
        if eng is not None:
            state, prepared, step_fn = eng.state, None, None
        elif mode == "fused":
            from .decode import DecodeEngine
            step_fn = DecodeEngine(self, x_enc, batch_size=B, state=init_state)
            state = step_fn.state
            prepared = None
        else:
            state = init_state if init_state is not None else self.attentive_rnn.init_state(
                max_seqlen=max_seqlen, batch_size=B)
            prepared = self.attentive_rnn.cross_att.prepare(x_enc)

            def step_fn(y, t):
                h, att, _ = self.attentive_rnn.step(y, x_enc, t, state, prepared=prepared)
                return self.logits_head(h), att

        # ---- prompt prefill: the reference feeds the start token and the p_len prompt tokens one step at a time
        # (modeling_lina.py:152-176); their inputs are known in advance, so all p_len + 1 positions go through the stack in
        # ONE teacher-forced pass on the cached path (K2 chunk scan + conv prefill + cached pos_net block, the same
        # recurrence as p_len + 1 single steps) and the per-position bookkeeping below consumes its logits.
        pre_logits = pre_att = None
        n_pre = 0
        if prompt is not None and p_len > 0 and hasattr(self.attentive_rnn, "step"):
            n_pre = min(p_len + 1, max_seqlen)
            y_seq = torch.cat([y_embd, prompt[:, :n_pre - 1]], dim=1)
            h, pre_att, _ = self.attentive_rnn.step(y_seq, x_enc, 0, state, prepared=prepared)
            pre_logits = self.logits_head(h)                        # [B,n_pre,Q,L]

        def pick_tokens(logits):                                    # [B,1,Q,L] -> [Q,B,1]
            per_q = logits.squeeze(1).transpose(0, 1)               # [Q,B,L]
            return torch.stack([topk_sampling(per_q[i], k=k, temp=temp) if i < first_greedy_quant
                                else topk_sampling(per_q[i], k=1) for i in range(Q)])

        if eng is not None:
            # ---- the device-side loop: the prefill's positions are picked here and filed in front of it
            preload = None
            y0 = y_embd
            if n_pre > 0:
                pre_q = torch.cat([pick_tokens(pre_logits[:, t:t + 1]) for t in range(n_pre)], dim=2)   # [Q,B,n_pre]
                preload = (pre_q, pre_att)
                y0 = prompt[:, [n_pre - 1]] if n_pre - 1 < p_len else self.rvq_embed.embed_sum(pre_q[:, :, -1:])
            if seed is None:
                seed = int(torch.randint(0, 2 ** 62, (1,)))
            qs, atts, n = eng.generate(max_seqlen, y0, k=k, temp=temp, first_greedy_quant=first_greedy_quant, seed=seed,
                                       force_max_seqlen=force_max_seqlen, stop_check_every=stop_check_every,
                                       log_att=True, preload=preload)
            return self._finish_generate(qs, atts, (qs == 2).all(dim=0), B, device)

        all_stop = torch.zeros(B, 1, dtype=torch.bool, device=device)
        qs, atts, stop_tokens = [], [], []
        stop_at = None                      # first step index at which every row had stopped
        for t in range(max_seqlen):
            if t < n_pre:
                logits, att = pre_logits[:, t:t + 1], pre_att[:, :, t:t + 1]
            else:
                logits, att = step_fn(y_embd, t)                # [B,1,Q,L], [B,2,1,Ttxt]
            atts.append(att)
            q_sampled = pick_tokens(logits)                     # [Q,B,1]
            qs.append(q_sampled)
            is_stop = (q_sampled == 2).all(dim=0)               # every quantizer emitted the stop token
            stop_tokens.append(is_stop)
            all_stop |= is_stop
            if not force_max_seqlen and ((t + 1) % stop_check_every == 0 or t == max_seqlen - 1):
                flags = torch.stack([s.squeeze(-1) for s in stop_tokens]).cumsum(0).bool().all(dim=1)  # [t+1]
                if bool(flags.any()):
                    stop_at = int(torch.nonzero(flags)[0])
                    break
            y_embd = prompt[:, [t]] if (prompt is not None and t < p_len) else self.rvq_embed.embed_sum(q_sampled)

        if stop_at is not None:             # trim to what a per-step check would have produced
            qs, atts, stop_tokens = qs[:stop_at + 1], atts[:stop_at + 1], stop_tokens[:stop_at + 1]
        atts = torch.cat(atts, dim=2) if atts[0] is not None else None
        qs = torch.stack(qs, dim=2).squeeze(-1)                                  # [Q,B,n]
        return self._finish_generate(qs, atts, torch.cat(stop_tokens, dim=1), B, device)

    def _finish_generate(self, qs, atts, is_stop, B, device):
        """Post-processing of the reference (modeling_lina.py:180-192) from the [B,n] stop flags: the stop-flag matrix
        with the closing column of ones, the un-delayed codes and the per-row cuts.  The reference takes
        ``torch.unique(stop_idx[i])[1]`` row by row (a sort and a host read per row); the set it sorts is {0} plus the
        positions of the row's stop flags, so element [1] is the first position >= 1 that carries a flag -- computed here
        for all rows at once and read back once."""
        stop_tokens = torch.cat([is_stop.float(), torch.ones(B, 1, device=device)], dim=1)    # [B,n+1]
        n = stop_tokens.shape[1]
        rvq = (undelay_rvq(qs) - self.n_special_token_in).clamp_min(0)
        pos = torch.arange(n, device=device)[None, :].expand(B, -1)
        flagged = (stop_tokens != 0) & (pos >= 1)
        if n < 2:
            raise IndexError("generate_batch: no step was decoded (max_seqlen == 0)")   # the reference's unique()[1] raises too
        first = torch.where(flagged, pos, torch.full_like(pos, n)).min(dim=1).values.tolist()
        cuts = [(rvq[:, i:i + 1, :idx - self.n_quant], None if atts is None else atts[i, :, :idx])
                for i, idx in enumerate(first)]
        return qs, atts, stop_tokens, cuts
