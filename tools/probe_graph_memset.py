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

# ---- second probe: a memset node proper (hipMemsetAsync captured into the graph) followed by `buf += 1`: every replay must
# leave 1.0 everywhere; a replay that skips the memset leaves 2, 3, ...
import ctypes  # noqa: E402

hip = None
for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
    try:
        hip = ctypes.CDLL(name)
        break
    except OSError:
        continue
if hip is not None:
    hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    buf = torch.full((1 << 20,), 7.0, device=dev)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, capture_error_mode="thread_local"):
        st = torch.cuda.current_stream().cuda_stream
        rc = hip.hipMemsetAsync(buf.data_ptr(), 0, buf.numel() * 4, ctypes.c_void_p(st))
        buf += 1.0
    for i in range(4):
        g2.replay()
        torch.cuda.synchronize()
        print(json.dumps({"memset_node_replay": i, "rc": rc, "min": float(buf.min()), "max": float(buf.max())}), flush=True)
