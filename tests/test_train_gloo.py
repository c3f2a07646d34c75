"""Training step, N > 1 on CPU: two gloo ranks (127.0.0.1), each running the real host code and the kernel
sources (K2/K2b, K3/K3b, K5/K5b on the emulator) on its micro-batch under DistributedDataParallel; the
all-reduced gradients must equal the mean of the two single-process micro-batch gradients, and the
parameters after one AdamW step must agree across ranks."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup_emu():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import EmuBackend
    from emu import build_emu
    from lina_speech_amd import _lib, ops
    ops.set_backend(EmuBackend(_lib.bind(build_emu.build(), hip_runtime=False)))


def _model():
    from model_cases import build_lina, golden_state_dict, load_golden
    model = build_lina()
    model.load_state_dict(golden_state_dict(load_golden("lina_d64.npz")))
    return model


def _micro(rank):
    from lina_speech_amd.train import synthetic_batch
    return synthetic_batch(b=2, n=11, t_txt=9, n_codebook=253, seed=100 + rank, ragged=True)


def _worker(rank, world, port, out_path):
    import torch.distributed as dist
    _setup_emu()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lina_speech_amd.train import TrainStep
    ts = TrainStep(_model(), autocast_dtype=None, grad_clip=None, lr=1e-3, n_warmup_steps=0)
    assert ts.net is not ts.model            # wrapped in DDP
    loss = ts.loss(_micro(rank))
    loss.backward()
    grads = {n: p.grad.clone() for n, p in ts.model.named_parameters() if p.grad is not None}
    ts.opt.step()
    params = {n: p.detach().clone() for n, p in ts.model.named_parameters()}
    torch.save({"grads": grads, "params": params, "loss": loss.detach()}, f"{out_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_ddp_gradients_equal_mean_of_microbatch_gradients(tmp_path, emu):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "ddp")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    # single-process reference: mean of the two micro-batch gradients
    from lina_speech_amd.train import TrainStep
    ref = {}
    for rank in range(2):
        ts = TrainStep(_model(), autocast_dtype=None, grad_clip=None, ddp=False)
        loss = ts.loss(_micro(rank))
        assert torch.allclose(loss.detach(), (r0, r1)[rank]["loss"], rtol=1e-6)
        loss.backward()
        for n, p in ts.model.named_parameters():
            if p.grad is not None:
                ref[n] = ref.get(n, 0) + 0.5 * p.grad
    assert set(ref) == set(r0["grads"])
    for n, g in ref.items():
        scale = g.abs().max().clamp_min(1e-8)
        assert (r0["grads"][n] - g).abs().max() <= 1e-5 * scale + 1e-9, n
        assert torch.equal(r0["grads"][n], r1["grads"][n]), n
    for n in r0["params"]:
        assert torch.equal(r0["params"][n], r1["params"][n]), n


def test_train_step_decreases_loss(emu):
    from lina_speech_amd.train import TrainStep
    ts = TrainStep(_model(), autocast_dtype=None, lr=3e-3, weight_decay=0.0, n_warmup_steps=0)
    batch = _micro(0)
    losses = [float(ts.step(batch)) for _ in range(4)]
    assert losses[-1] < losses[0], losses


def test_initial_state_tuning_loop_lowers_the_loss_with_frozen_weights(emu):
    from lina_speech_amd.initial_state import train_initial_state
    model = _model()
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    batch = _micro(0)
    params, losses = train_initial_state(model, iter([batch] * 8), n_steps=8, lr=0.2, grad_acc=1, rank=1, device="cpu")
    assert losses[-1] < losses[0], losses
    assert len(params) == 2 and all(len(p) == 2 for p in params)          # one (k, v) pair per GLA block (encoder+decoder)
    for n, p in model.named_parameters():
        assert torch.equal(p, before[n]), f"{n} changed: the model must stay frozen"
