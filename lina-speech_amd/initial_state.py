"""Initial-state tuning ("speaker states"): learn the start state of every GLA block for a voice
while the model stays frozen (reference initial_state.py:85-160), plus the speaker-state file format
(reference initial_state.py:20-48: safetensors with keys ``layer{i}_k`` / ``layer{i}_v`` for rank-r states,
``layer{i}`` for full ones).

The kernel side is K2/K2b: the forward runs with ``initial_state`` tensors that require grad and K2b returns ``dh0``
(sweep V's final accumulator).  The reference's data pipeline (HF datasets, tokenizer, collate) is not rebuilt: the
loop takes ready ``train.Batch`` objects.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence, Tuple

import torch

from .lina_model import LinaModel
from .train import Batch


def speaker_state_dict(params: Sequence) -> dict:
    """[(k, v) | state] per block -> flat dict (reference initial_state.py:20-30)."""
    sd = {}
    for i, layer in enumerate(params):
        if isinstance(layer, (tuple, list)) and len(layer) == 2:
            sd[f"layer{i}_k"], sd[f"layer{i}_v"] = layer[0].detach(), layer[1].detach()
        else:
            sd[f"layer{i}"] = (layer[0] if isinstance(layer, (tuple, list)) else layer).detach()
    return sd


def save_speaker_state(params: Sequence, path: str) -> None:
    from safetensors.torch import save_file
    save_file({k: v.contiguous().cpu() for k, v in speaker_state_dict(params).items()}, path)


def parse_speaker_state(path: str, device: str = "cpu") -> List[Tuple[torch.Tensor, torch.Tensor]]:
    """Read the rank-r (k, v) pairs of a speaker-state file in block order (reference initial_state.py:38-48)."""
    from safetensors import safe_open
    with safe_open(path, framework="pt", device=device) as st:
        keys = [k for k in st.keys() if k.endswith("_k")]
        keys.sort(key=lambda name: int("".join(c for c in name if c.isdigit())))
        return [(st.get_tensor(k), st.get_tensor(k[:-2] + "_v")) for k in keys]


def tuning_loss(model: LinaModel, batch: Batch, params: Sequence, scale: float = 0.02) -> torch.Tensor:
    """Teacher-forced loss with the start states built from ``params`` (reference model_step, initial_state.py:111-124)."""
    init_state = model.attentive_rnn.get_state_from_params(params, batch.x.shape[0], scale=scale)
    return model(batch.x, batch.y, batch.encoder_mask, batch.crossatt_mask, logits_mask=batch.logits_mask,
                 init_state=init_state)[1]


def train_initial_state(model: LinaModel, batches: Iterable[Batch], n_steps: int, lr: float = 0.1, grad_acc: int = 4,
                        scale: float = 0.02, rank: Optional[int] = 1, seed: int = 123, device=None):
    """Adam on the state parameters only (reference train_initial_state): returns (params, losses)."""
    device = device if device is not None else next(model.parameters()).device
    torch.manual_seed(seed)
    model.attentive_rnn.to_mode("fused_recurrent")
    model.train()
    frozen = [(p, p.requires_grad) for p in model.parameters()]
    for p, _ in frozen:
        p.requires_grad_(False)
    params = model.attentive_rnn.get_init_state_tuning_params(lora=rank, device=device)
    flat = [t for layer in params for t in (layer if isinstance(layer, tuple) else (layer,))]
    opt = torch.optim.Adam(flat, lr=lr)
    losses = []
    it = iter(batches)
    for i in range(n_steps):
        loss = tuning_loss(model, next(it).to(device), params, scale)
        losses.append(float(loss.detach()))
        loss.backward()
        if i % grad_acc == grad_acc - 1:
            opt.step()
            opt.zero_grad()
    model.eval()
    for p, was in frozen:                  # the reference leaves the model's requires_grad flags as they were
        p.requires_grad_(was)
    return params, losses
