"""Minimal training step for the teacher-forced path.

Mirrors what the reference's Lightning module does per step (train_lina.py:72-120): build the masks,
``LinaModel.forward`` -> cross-entropy (``ignore_index=1``), ``loss.backward()``, AdamW.  Nothing of
Lightning / the CLI / the data pipeline is rebuilt (out of scope).

MI355X-first:
  * the sequence kernels on the path are the HIP ones (K2/K2b chunk scan, K3/K3b short conv, K5/K5b
    norm-gate, K6 embedding gather) reached through ``ops``; GEMMs, LayerNorm, softmax attention of the
    text encoder and the loss stay on the vendor libraries through torch;
  * data parallel only: one process per GPU, gradients all-reduced by RCCL over xGMI through
    ``DistributedDataParallel`` with LARGE buckets (xGMI rings are per-link bound: few big all-reduces,
    overlapped with the rest of backward) and ``gradient_as_bucket_view`` (no extra gradient copy);
  * bf16 autocast for the GEMMs, fp32 master weights, fp32 recurrent state / gate cumsum inside K2.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import Optional

import torch
import torch.distributed as dist

from . import ops
from .lina_model import LinaModel

DDP_BUCKET_MB = 128        # ~0.67 GB of fp32 gradients at L169 -> 6 buckets


class FusedAdamW(torch.optim.AdamW):
    """``torch.optim.AdamW`` (the reference's optimizer, train_lina.py:104-118) with the update on K17 ``lina_adamw_multi``:
    same hyper-parameters, ``param_groups`` and per-parameter state (``step`` / ``exp_avg`` / ``exp_avg_sq`` -- ``state_dict``s
    move freely between the two), the operation order of torch's fused kernel; 48 tensors per launch with the pointer table
    passed by value.  Falls back to torch's implementation for a step that holds anything K17 is not built for (non-fp32 or
    non-contiguous parameters / gradients, sparse gradients, amsgrad, maximize, tensors off the fused-op device)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False, maximize=False):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, maximize=maximize,
                         foreach=False, fused=False)

    def _eligible(self) -> bool:
        for group in self.param_groups:
            if group["amsgrad"] or group["maximize"] or group.get("capturable") or group.get("differentiable"):
                return False
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if (g.is_sparse or p.dtype != torch.float32 or g.dtype != torch.float32 or not p.is_contiguous()
                        or not g.is_contiguous() or p.device != g.device or not ops.fused_ops_available(p)):
                    return False
                st = self.state.get(p)
                if st:                                        # a state loaded from a fused / capturable run: torch's path
                    if (st["step"].is_cuda or not st["exp_avg"].is_contiguous() or not st["exp_avg_sq"].is_contiguous()
                            or st["exp_avg"].dtype != torch.float32 or st["exp_avg"].device != p.device):
                        return False
        return True

    @torch.no_grad()
    def step(self, closure=None):
        if not self._eligible():
            return super().step(closure)
        import ctypes as C
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        be = ops.get_backend()
        for group in self.param_groups:
            by_step = {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:                              # torch.optim.AdamW's own lazy state (non-capturable form)
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                if p.numel():
                    by_step.setdefault((p.device, float(st["step"])), []).append((p, st))
            beta1, beta2 = group["betas"]
            lr = float(group["lr"])
            for (_, t), items in by_step.items():
                n = len(items)
                arr = lambda f: (C.c_void_p * n)(*[f(p, st) for p, st in items])
                numel = (C.c_int64 * n)(*[p.numel() for p, _ in items])
                with torch.cuda.device_of(items[0][0]):       # (no-op for the emulator's host tensors)
                    ops._check(be.lib.lina_adamw_multi(arr(lambda p, st: p.data_ptr()), arr(lambda p, st: p.grad.data_ptr()),
                                                       arr(lambda p, st: st["exp_avg"].data_ptr()),
                                                       arr(lambda p, st: st["exp_avg_sq"].data_ptr()), numel, n, lr, beta1,
                                                       beta2, group["eps"], group["weight_decay"], 1.0 - beta1 ** t,
                                                       1.0 - beta2 ** t, be.stream(items[0][0])))
        ops.clear_mlp_pack()          # K17 writes the parameters behind their version counters: drop the cached padded operands
        return loss


@dataclass
class Batch:
    x: torch.Tensor                 # [b, Ttxt] text ids
    y: torch.Tensor                 # [b, n, Q] codec ids (y[:,0] = BOS 1)
    encoder_mask: torch.Tensor      # [b, Ttxt, Ttxt] bool
    crossatt_mask: torch.Tensor     # [b, n, Ttxt] bool
    logits_mask: Optional[torch.Tensor] = None   # [b, n] bool

    def to(self, device):
        mv = lambda t: None if t is None else t.to(device, non_blocking=True)
        return Batch(mv(self.x), mv(self.y), mv(self.encoder_mask), mv(self.crossatt_mask), mv(self.logits_mask))


def synthetic_batch(b: int, n: int, t_txt: int = 64, n_codebook: int = 4096, n_quant: int = 1, n_txt_vocab: int = 256,
                    seed: int = 0, ragged: bool = False) -> Batch:
    """SURVEY.md 8(d) config-5 inputs: text ids randint(3, vocab), codec ids randint(3, codebook+3) with
    y[:,0] = 1, full masks (``ragged`` shortens some texts / targets to exercise the masks)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randint(3, n_txt_vocab, (b, t_txt), generator=g)
    y = torch.randint(3, n_codebook + 3, (b, n, n_quant), generator=g)
    y[:, 0] = 1
    txt_len = torch.full((b,), t_txt)
    y_len = torch.full((b,), n)
    if ragged:
        txt_len = torch.randint(max(1, t_txt // 2), t_txt + 1, (b,), generator=g)
        y_len = torch.randint(max(2, n // 2), n + 1, (b,), generator=g)
    em = torch.arange(t_txt)[None, :] < txt_len[:, None]
    encoder_mask = em[:, None, :] & em[:, :, None]
    crossatt_mask = em[:, None, :].expand(b, n, t_txt).contiguous()
    logits_mask = torch.arange(n)[None, :] < y_len[:, None]
    return Batch(x, y, encoder_mask, crossatt_mask, logits_mask)


class TrainStep:
    """forward + loss + backward + AdamW for one micro-batch per rank; data parallel when a process group is up."""

    def __init__(self, model: LinaModel, lr: float = 5e-4, weight_decay: float = 0.1, betas=(0.9, 0.999),
                 n_warmup_steps: int = 500, n_training_steps: int = 300000,
                 autocast_dtype: Optional[torch.dtype] = torch.bfloat16, device: Optional[torch.device] = None,
                 grad_clip: Optional[float] = None, ddp: Optional[bool] = None):
        """Optimiser defaults are the reference's (train_lina.py:25-29,104-118): AdamW lr 5e-4, betas (0.9, 0.999),
        weight decay 0.1, cosine schedule with 500 warm-up steps over 300 000 steps, no gradient clipping.
        (Rounds 2-3 carried an opt-in ``graph=True`` -- the whole step as one hipGraph.  It replayed correctly only on some
        runs -- torch's two-stage reductions read stale partials under replay on ROCm 7.2, profiles/r03_graph_replay_reduction.txt --
        and gained 0.9 % when it did (56.8 vs 57.3 ms, profiles/r03_train_graph_probe.txt): removed in round 4)"""
        self.device = device if device is not None else next(model.parameters()).device
        self.model = model.to(self.device).train()
        self.autocast_dtype = autocast_dtype
        self.grad_clip = grad_clip
        use_ddp = (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1) if ddp is None else ddp
        self.net = self.model
        if use_ddp:
            from torch.nn.parallel import DistributedDataParallel as DDP
            ids = [self.device.index] if self.device.type == "cuda" else None
            self.net = DDP(self.model, device_ids=ids, bucket_cap_mb=DDP_BUCKET_MB, gradient_as_bucket_view=True,
                           broadcast_buffers=False)
        # K17 where the fused ops run (a ROCm device, or the test emulator); elsewhere torch's own AdamW
        if ops.fused_ops_available(next(self.model.parameters())):
            self.opt = FusedAdamW(self.model.parameters(), lr=lr, weight_decay=weight_decay, betas=betas)
        else:
            self.opt = torch.optim.AdamW(self.model.parameters(), lr=lr, weight_decay=weight_decay, betas=betas)

        def cosine_with_warmup(step: int) -> float:          # transformers.get_cosine_schedule_with_warmup, half a cycle
            if step < n_warmup_steps:
                return step / max(1, n_warmup_steps)
            progress = (step - n_warmup_steps) / max(1, n_training_steps - n_warmup_steps)
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * progress)))

        self.sched = (torch.optim.lr_scheduler.LambdaLR(self.opt, cosine_with_warmup)
                      if n_training_steps and n_training_steps > 0 else None)

    def loss(self, batch: Batch) -> torch.Tensor:
        kw = dict(logits_mask=batch.logits_mask, return_masked=False)
        if self.autocast_dtype is not None and self.device.type == "cuda":
            with torch.autocast("cuda", dtype=self.autocast_dtype):
                out = self.net(batch.x, batch.y, batch.encoder_mask, batch.crossatt_mask, **kw)
        else:
            out = self.net(batch.x, batch.y, batch.encoder_mask, batch.crossatt_mask, **kw)
        return out[1]

    def step(self, batch: Batch) -> torch.Tensor:
        """One optimizer step; returns the (detached) loss of this rank's micro-batch."""
        self.opt.zero_grad(set_to_none=True)
        loss = self.loss(batch)
        loss.backward()                       # DDP: RCCL all-reduce (mean) of the buckets overlaps with backward
        if self.grad_clip is not None:
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.grad_clip)
        self.opt.step()
        ops.clear_mlp_pack()                  # padded SwiGLU weights of this step (also covers optimizers that write .data)
        if self.sched is not None:
            self.sched.step()
        return loss.detach()


def init_distributed(backend: Optional[str] = None) -> tuple[int, int, torch.device]:
    """Join the process group torch.distributed.run describes in the environment (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_*); backend 'nccl' is RCCL on ROCm.  Returns (rank, world, device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    gpu = torch.cuda.is_available()
    device = torch.device("cuda", local) if gpu else torch.device("cpu")
    if gpu:
        torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend or ("nccl" if gpu else "gloo"), rank=rank, world_size=world)
    return rank, world, device
