"""Step-by-step diagnostics of the captured train step (TrainStep(graph=True)) at the config-5 shape: loss, time and the
finiteness of the parameters after every replay, printed as it goes (a run that degrades shows where)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lina_speech_amd import configs  # noqa: E402
from lina_speech_amd.train import TrainStep, synthetic_batch  # noqa: E402

if os.environ.get("PROBE_BLAS"):                     # e.g. hipblas (= rocBLAS path: no hipBLASLt stream-K kernels / memsets)
    torch.backends.cuda.preferred_blas_library(os.environ["PROBE_BLAS"])
# PROBE_PLAIN=mlp,linear,slab,gate,ce,ln,chain,conv,norm: replace that fused path by the plain torch ops (monkeypatched here, no
# product switch) -- to bisect which node a failing replay depends on
import torch.nn.functional as F  # noqa: E402
from lina_speech_amd import blocks, ops  # noqa: E402
plain = set(filter(None, os.environ.get("PROBE_PLAIN", "").split(",")))
if "mlp" in plain:
    ops.swiglu_mlp = lambda x, wi, bi, wo, bo: ops.linear(ops.swiglu_gate(ops.linear(x, wi, bi)), wo, bo)
if "linear" in plain:
    ops.linear = lambda x, w, b=None: F.linear(x, w, b)
if "slab" in plain:
    ops.split_slab = lambda z, sizes: (z.split(list(sizes), dim=-1), None)
if "gate" in plain:
    ops.gate_lowrank = lambda lr, w, b=None, normalizer=16.0, clamp_min=None: (
        F.logsigmoid(F.linear(lr, w, b)) / normalizer if clamp_min is None
        else torch.clamp_min(F.logsigmoid(F.linear(lr, w, b)) / normalizer, clamp_min))
if "ce" in plain:
    ops.cross_entropy = lambda lg, t, ignore_index=-100: F.cross_entropy(lg, t, ignore_index=ignore_index)
if "ln" in plain:
    ops.fused_ops_available_orig = ops.fused_ops_available
    _ln = ops.layer_norm

    def _plain_ln(x, weight, bias, eps=1e-5, residual=None, out_dtype=None):
        xs = x if residual is None else x + residual
        y = F.layer_norm(xs, (x.shape[-1],), weight, bias, eps)
        return y if residual is None else (y, xs)
    ops.layer_norm = _plain_ln
if "chain" in plain:
    blocks.MixingBlock.can_defer = lambda self, x: False
if "conv" in plain:
    from oracle import gla_oracle as O  # noqa: E402  (diagnostics only: the CPU restatement's conv is plain torch)
    ops.short_conv = lambda x, w, bias=None, mask=None, cache=None, activation="silu", grad_slab=None: O.short_conv(
        x, w, mask, cache, activation, bias).to(x.dtype)
if "norm" in plain:
    from oracle import gla_oracle as O2  # noqa: E402
    ops.rmsnorm_swish_gate = lambda x, g=None, weight=None, eps=1e-5, n_partial=1, out_dtype=None, out=None, grad_slab=None: (
        O2.rmsnorm_swish_gate(x, g, weight, eps) if g is not None else O2.rmsnorm(x, weight, eps)).to(x.dtype)
print(json.dumps({"plain": sorted(plain)}), flush=True)

dev = torch.device("cuda", 0)
torch.manual_seed(0)
b, T = int(os.environ.get("PROBE_B", "8")), int(os.environ.get("PROBE_T", "4096"))
ts = TrainStep(configs.l169(), device=dev, ddp=False, graph=True)
batch = synthetic_batch(b=b, n=T + 1, t_txt=64, seed=1).to(dev)
for i in range(int(os.environ.get("PROBE_STEPS", "14"))):
    t0 = time.perf_counter()
    loss = float(ts.step(batch))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    bad = [n for n, p in ts.model.named_parameters() if not bool(torch.isfinite(p).all())]
    badg = [n for n, p in ts.model.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
    n_par = sum(1 for _ in ts.model.parameters())
    kinds = sorted({n.rsplit(".", 1)[-1] for n in bad}), sorted({str(tuple(p.shape)) for n, p in ts.model.named_parameters() if n in set(bad)})[:8]
    print(json.dumps({"step": i, "ms": round(dt, 1), "loss": loss, "n_params": n_par, "n_bad_params": len(bad), "n_bad_grads": len(badg),
                      "bad_param_kinds": kinds[0], "bad_param_shapes": kinds[1], "nonfinite_params": bad[:10],
                      "nonfinite_grads": badg[:10]}), flush=True)
    if bad or loss != loss:
        break
