"""Does a hipGraph replay re-run the memset nodes of a captured region?  torch's column sum of a tall matrix is a two-stage
reduction whose semaphores are zeroed by a captured cudaMemsetAsync; a replay that skips (or mis-orders) that memset leaves
the semaphores at the previous replay's count and the sum wrong.  Prints one JSON line per replay."""
import json

import torch

dev = torch.device("cuda", 0)
torch.manual_seed(0)
x = torch.randn(32768, 2730, device=dev, dtype=torch.bfloat16)
z = torch.zeros(4096, device=dev)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        y = x.sum(0)
        z2 = z.clone().zero_()
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    y = x.sum(0)
    y32 = x.float().sum(0)
    w = torch.zeros(1000, device=dev) + y32[:1000]
for i in range(5):
    x.copy_(torch.randn(32768, 2730, device=dev, dtype=torch.bfloat16) * (i + 1))
    g.replay()
    torch.cuda.synchronize()
    ref = x.float().sum(0)
    e1 = ((y.float() - ref).abs().max() / ref.abs().max()).item()
    e2 = ((y32 - ref).abs().max() / ref.abs().max()).item()
    e3 = ((w - ref[:1000]).abs().max() / ref.abs().max()).item()
    print(json.dumps({"replay": i, "bf16_sum_rel_err": e1, "f32_sum_rel_err": e2, "zeros_plus_rel_err": e3,
                      "finite": bool(torch.isfinite(y).all() and torch.isfinite(y32).all())}), flush=True)
