"""Premise check: the mixer's stacked in-projection (N = 4160 = q | k | v | g | low-rank 16 | pad 48: 16.25 column tiles of 256 -> nine
rounds of 256 CUs at M = 32768) against the same product as N = 4096 (eight rounds) + a 64-column GEMM, both written into ONE
[M, 4160] buffer through out= views.   python tools/perf_inproj_split.py"""
import torch

dev = torch.device("cuda", 0)
M, K, N = 32768, 1024, 4160
g = torch.Generator().manual_seed(0)
x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
w = (torch.randn(N, K, generator=g) * 0.03).to(torch.bfloat16).to(dev)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
side = torch.cuda.Stream()


def timed(fn, iters=30, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def one():
    torch.mm(x, w.t(), out=out)


def two_serial():
    torch.mm(x, w[:4096].t(), out=out[:, :4096])
    torch.mm(x, w[4096:].t(), out=out[:, 4096:])


def two_streams():
    ev = torch.cuda.current_stream().record_event()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        torch.mm(x, w[4096:].t(), out=out[:, 4096:])
        ev2 = side.record_event()
    torch.mm(x, w[:4096].t(), out=out[:, :4096])
    torch.cuda.current_stream().wait_event(ev2)


ref = (x.float() @ w.float().t())
for name, fn in (("one GEMM, N = 4160", one), ("N = 4096 then N = 64, one stream", two_serial), ("N = 4096 beside N = 64 on a side stream", two_streams),
                 ("one GEMM, N = 4160 (again)", one)):
    out.zero_()
    fn()
    torch.cuda.synchronize()
    err = ((out.float() - ref).abs().max() / ref.abs().max()).item()
    print(f"{name:45s} {timed(fn):8.1f} us   rel err {err:.2e}   out data_ptr stable: {out.data_ptr() == out.data_ptr()}")
print("N = 4096 alone:", round(timed(lambda: torch.mm(x, w[:4096].t(), out=out[:, :4096])), 1), "us;  N = 64 alone:",
      round(timed(lambda: torch.mm(x, w[4096:].t(), out=out[:, 4096:])), 1), "us")

# ---- backward products of the same projection: dW = dY^T X (fp32 result) and dX = dY W
dy = torch.randn(M, N, generator=g).to(torch.bfloat16).to(dev)
dw = torch.empty(N, K, dtype=torch.float32, device=dev)
dx = torch.empty(M, K, dtype=torch.bfloat16, device=dev)
print("dW one GEMM [4160, 32768] x [32768, 1024] (fp32 out):", round(timed(lambda: torch.mm(dy.t(), x, out_dtype=torch.float32)), 1), "us")
print("dW rows [:4096]:", round(timed(lambda: torch.mm(dy[:, :4096].t(), x, out_dtype=torch.float32)), 1), "us;  rows [4096:]:",
      round(timed(lambda: torch.mm(dy[:, 4096:].t(), x, out_dtype=torch.float32)), 1), "us")
for S in (2, 4, 8):
    f = lambda S=S: torch.bmm(dy[:, :4096].reshape(S, M // S, 4096).transpose(1, 2), x.view(S, M // S, K), out_dtype=torch.float32).sum(0)
    try:
        print(f"dW rows [:4096] token-split S = {S} (bmm + sum):", round(timed(f), 1), "us")
    except Exception as e:  # noqa: BLE001
        print("split", S, "failed:", str(e)[:100])
print("dX one GEMM [32768, 4160] x [4160, 1024]:", round(timed(lambda: torch.mm(dy, w, out=dx)), 1), "us")
print("dX K = 4096 only:", round(timed(lambda: torch.mm(dy[:, :4096], w[:4096], out=dx)), 1), "us;  + K = 64 accumulate (addmm_):",
      round(timed(lambda: dx.addmm_(dy[:, 4096:], w[4096:])), 1), "us")
