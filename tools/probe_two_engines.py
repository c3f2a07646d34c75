"""Do two independent decode engines on two HIP streams overlap?  The B = 512 step is 2/3 K1w (HBM-bound) and 1/3 projections
(L2- / latency-bound, HBM idle): two half-batch engines whose graph replays are enqueued on different streams could run one
half's projections under the other half's K1w -- if the hardware co-schedules the two queues.
    python tools/probe_two_engines.py [rows per engine] [engines]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lina_speech_amd.configs import l169  # noqa: E402
from lina_speech_amd.decode import DecodeEngine  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
NE = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = l169().eval().to(dev, torch.bfloat16)
texts = torch.randint(3, 256, (R * NE, 64), generator=torch.Generator().manual_seed(1234)).to(dev)
steps, N = 400, 8
with torch.inference_mode():
    x_enc = m.txt_encoder(m.txt_embed(texts))
    engs = [DecodeEngine(m, x_enc[i * R:(i + 1) * R], batch_size=R) for i in range(NE)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(NE)]
    for e in engs:
        e.begin_greedy(2 * steps + 64, log_att=True)
        e.greedy_steps(16)
    torch.cuda.synchronize()

    def run(concurrent):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        main = torch.cuda.current_stream(dev)
        for s in streams:
            s.wait_stream(main)
        for _ in range(steps // N):
            for e, s in zip(engs, streams):
                if concurrent:
                    with torch.cuda.stream(s):
                        e.greedy_steps(N)
                else:
                    e.greedy_steps(N)
        for s in streams:
            main.wait_stream(s)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    seq = run(False)
    con = run(True)
print(f"{NE} engines x {R} rows: sequential {seq:.4f} ms per token of all rows ({R * NE / seq:.1f} k tok/s), "
      f"two streams {con:.4f} ms ({R * NE / con:.1f} k tok/s)")
