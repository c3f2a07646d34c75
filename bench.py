#!/usr/bin/env python
"""bench.py -- codec tokens/s of the 169M (d1024 x l12) batched greedy decode on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one decode token for every row of the batch: the whole hot path (13 GLA blocks with
the fused HIP step kernels, blind cross-attention, 4099-way head, device-side pick + stop flags +
attention log + next-token embedding) -- the loop `LinaModel.generate_batch` runs, 8 tokens per
hipGraph replay.  The utterance batch is the one BASELINE.json's metric is quoted on, B_total = 512
(`--total-batch`), sharded over the N GPUs with NO data-path collective: 512 rows on one GPU at
N = 1, 64 per GPU at N = 8 ("scaling": "strong").  `--batch-per-gpu R` keeps R rows per GPU instead
(weak scaling; R = 64 is BASELINE configs[1]); at N = 1 the per-GPU batches 64 / 128 / 256 are
measured beside the headline (`per_gpu_batch`), and the reference's decode ENTRY POINT end to end
(`generate_batch`).  Inputs and state are resident in HBM when the timed region starts.  Synthetic
data, seeded random-init weights from the reference initialisers.

Prints ONE JSON line (rank 0) with the driver contract plus
  roofline      the dominant kernel of the step -- K1w + K5 (lina::gla_decode_window_kernel), the windowed recurrent-state
                update: its bytes per launch / its launch duration measured IN SITU (the captured step graph
                with and without its 13 update launches, HIP events around the replays), the back-to-back figure and the
                immediate-form bytes beside it; traffic from the committed PMC passes
  step_roofline the whole step's HBM bytes / ms_per_step (also priced with the immediate-form state bytes)
  chunk_kernel  the same accounting for K2 (lina::gla_chunk_*_kernel) at the training shape (+ b=8, K2b, vocoder, train step)
  other_head_shapes / decode_f32   the same engine at H=8, H=16, expand_v=2 and in fp32
  cpu_baseline  the CPU oracle (pure-PyTorch recurrent port of the reference path) on a bounded sample, host CPU named
`--gpus N` without a launcher starts N ranks under torch.distributed.run; `--train` measures config 5 (DDP train step).
The timed region is max(--steps, 0.25 s of steps); "steps" reports what was timed.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
TOTAL_BATCH = 512            # default of --total-batch: the batch BASELINE.json's metric is quoted on (8 x configs[1]'s 64)
T_TXT = 64


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def k1_algorithmic_bytes(B, H, Dk, Dv, e_io, e_g):
    """Per launch (one layer, one token): state read + write in fp32, q,k (e_io), gk (e_g) in, v in, o out.
    SURVEY 8(d): 2*4*H*Dk*Dv + e*(3*H*Dk + 2*H*Dv) per (row, layer, token)."""
    return B * (8 * H * Dk * Dv + e_io * (2 * H * Dk + 2 * H * Dv) + e_g * H * Dk)


def k1w_algorithmic_bytes(B, H, Dk, Dv, e_io, e_g, W):
    """K1w per launch, averaged over the W positions of a window: the state is read every step and written every W-th,
    q,k,v,gk in and o out as for K1, plus the window history (fp32): one (k, c, v) entry written per step and the
    j earlier entries read at position j."""
    state = 4 * H * Dk * Dv * (1.0 + 1.0 / W)
    io = e_io * (2 * H * Dk + 2 * H * Dv) + e_g * H * Dk
    hist_w = 4 * H * (2 * Dk + Dv)
    hist_r = 4 * H * (2 * Dk + Dv) * (W - 1) / 2.0
    return int(B * (state + io + hist_w + hist_r))


def measure_k1(engine, reps=20):
    """Average duration of one launch of the kernel the hipGraph step actually runs for the recurrent update, at the
    decode shape, on the engine's real buffers, cycling through the 13 layers (1.7 GB of state > L3, so every launch
    streams from HBM):  K1w + K5 (lina_gla_decode_window, all W window positions in turn) when the engine keeps the state
    lazily written, else K1d + K5 (lina_gla_decode_update_norm), else plain K1d.  HIP events on torch's current stream --
    the stream the C-ABI launches are enqueued on.  Returns (seconds per launch, entry point name)."""
    from lina_speech_amd import ops
    packs = engine.packs
    B = packs[0].S.shape[0]          # rows per launch (the engine may split the batch into parallel row ranges)
    fused = engine.fuse_norm and packs[0].row_split
    lazy = fused and packs[0].lazy
    W = engine.window
    steps = [torch.full((1,), j, dtype=torch.long, device=packs[0].S.device) for j in range(W)]
    origin = torch.zeros(1, dtype=torch.long, device=packs[0].S.device)

    def one_pass(j=0):
        for P in packs:
            q = P.qkv[:, :P.Kd].view(B, P.H, P.Dk)
            k = P.qkv[:, P.Kd:2 * P.Kd].view(B, P.H, P.Dk)
            v = P.qkv[:, 2 * P.Kd:].view(B, P.H, P.Dv)
            if lazy:
                ops.gla_decode_window(q, k, v, P.gk.view(B, P.H, P.Dk), P.S, P.g.view(B, P.H, P.Dv), P.gnw,
                                      P.og, P.hk, P.hc, P.hv, steps[j], origin, W, P.eps_gate, o_exchange=P.o_x,
                                      counters=P.counters)
            elif fused:
                ops.gla_decode_update_norm(q, k, v, P.gk.view(B, P.H, P.Dk), P.o_part, P.S, P.g.view(B, P.H, P.Dv),
                                           P.gnw, P.og, P.counters, P.eps_gate)
            else:
                ops.gla_decode_update(q, k, v, P.gk.view(B, P.H, P.Dk), P.o_part, P.S)

    engine.sync_state()
    for j in range(W):
        one_pass(j)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = (reps + W - 1) // W * W
    ev0.record()
    for r in range(reps):
        one_pass(r % W)
    ev1.record()
    torch.cuda.synchronize()
    return (ev0.elapsed_time(ev1) * 1e-3 / (reps * len(packs)),
            "lina_gla_decode_window" if lazy else "lina_gla_decode_update_norm" if fused else "lina_gla_decode_update")


def host_cpu():
    """CPU model and physical core count of the host the cpu_baseline runs on."""
    model, phys, logical = None, set(), 0
    try:
        pid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model is None:
                model = line.split(":", 1)[1].strip()
            elif line.startswith("processor"):
                logical += 1
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                phys.add((pid, line.split(":", 1)[1].strip()))
    except OSError:
        pass
    return {"model": model, "physical_cores": len(phys) or None, "logical_cpus": logical or os.cpu_count()}


def sustained(fn, warm_s=1.5, reps=200):
    """Average duration of fn() launched back to back with the engine clock SETTLED: the chip is power-managed, the clock
    ramps up from idle over the first ~0.5 s of load and then settles at its sustained value (tools/clk_sample.sh: K2 runs
    at ~1.8 GHz), so a burst of three launches after a pause measures the ramp, not the kernel.  Returns (sustained s,
    burst-of-3 s)."""
    fn()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(3):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    burst = ev0.elapsed_time(ev1) * 1e-3 / 3
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < warm_s:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
    ev0.record()
    for _ in range(reps):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) * 1e-3 / reps, burst


def measure_chunk(dev, B=64, H=4, T=4096, Dk=256, Dv=256, reps=200):
    """K2 at the training shape (config 5: seqlen 4096), bf16 I/O."""
    from lina_speech_amd import ops
    g = torch.Generator(device="cpu").manual_seed(0)
    mk = lambda D: torch.randn(B, T, H * D, generator=g).to(torch.bfloat16).to(dev).view(B, T, H, D).transpose(1, 2)
    q, k, v = mk(Dk), mk(Dk), mk(Dv)
    gk = (torch.nn.functional.logsigmoid(torch.randn(B, T, H * Dk, generator=g)) / 16).to(torch.bfloat16).to(dev)
    gk = gk.view(B, T, H, Dk).transpose(1, 2)
    # the training call (reference model/gla.py:193-195: output_final_state = use_cache = False); the prefill form that also
    # returns the final state (67 MB more at B = 64, the shape of the committed PMC traffic figure) is timed beside it
    dt, burst = sustained(lambda: ops.chunk_gla(q, k, v, gk, output_final_state=False), reps=reps)
    dt_state, _ = sustained(lambda: ops.chunk_gla(q, k, v, gk, output_final_state=True), warm_s=0.5, reps=reps)
    nbytes = B * H * T * 2 * (3 * Dk + 2 * Dv)                       # e*(3Dk+2Dv) per (row,head,token)
    flops = B * H * T * (2 * 64 * (Dk + Dv) + 4 * Dk * Dv)            # nominal C=64 count of SURVEY 8(d)
    traffic, traffic_src = None, None
    for name, note in (("r02_k2_b8_traffic.json", "; state-only pass + combine + full pass of the segment-parallel form"),
                       ("r05_k2_b8_traffic.json", "; state-only pass + combine + full pass of the segment-parallel form"),
                       ("r04_k2_h4_traffic.json", "; the training call"), ("r04_k2_h8_traffic.json", "; the training call"),
                       ("r04_k2_h16_traffic.json", "; the training call"),
                       # round 6: every figure re-measured in one session on the final tree (tests/gpu_r06_evidence.sh); the LAST match wins
                       ("r06_k2_b8_traffic.json", "; state-only pass + combine + full pass of the segment-parallel form"),
                       ("r06_k2_h4_traffic.json", "; the training call"), ("r06_k2_h8_traffic.json", "; the training call"),
                       ("r06_k2_h16_traffic.json", "; the training call"),
                       ("r06_k2_dv512_traffic.json", "; the training call, ONE launch of two XCD-paired workgroups per head: the second "
                                                     "reader of q, k, g hits the XCD's L2 (rounds 2-5, two launches: 1.43 x algorithmic)")):
        tpath = os.path.join(ROOT, "profiles", name)                 # PMC passes are separate runs; their committed summaries
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get("shape") == {"B": B, "H": H, "T": T, "Dk": Dk, "Dv": Dv}:
                traffic = tj["traffic_bytes_per_launch"]
                traffic_src = f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, gfx950-corrected{note})"
    sq = None                                                        # SQ counter passes (separate rocprofv3 runs): committed summary
    sname = "r06_k2_sq.json" if os.path.exists(os.path.join(ROOT, "profiles", "r06_k2_sq.json")) else "r05_k2_sq.json"
    spath = os.path.join(ROOT, "profiles", sname)
    if os.path.exists(spath) and (B, H, T, Dk, Dv) == (64, 4, 4096, 256, 256):
        der = next(iter(json.load(open(spath))["kernels"].values()))["derived"]
        sq = {"source": f"profiles/{sname} (rocprofv3 --pmc, SQ counters; tools/pmc_sq.py)",
              **{k_: der[k_] for k_ in ("mfma_util", "wait_share", "issue_stall", "active_share", "valu_share", "lds_conflict")
                 if k_ in der}}
    return {"kernel": "lina::gla_chunk_bf16_h256_kernel", "shape": {"B": B, "H": H, "T": T, "Dk": Dk, "Dv": Dv},
            "bytes_per_launch": nbytes, "traffic": traffic, "traffic_source": traffic_src,
            "mfma_util": None if sq is None else sq["mfma_util"], "sq_counters": sq,
            "dtype": "bf16", "ms": dt * 1e3, "ms_burst_of_3": burst * 1e3, "bound": "hbm", "achieved": nbytes / dt / 1e9,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nbytes / dt / 1e9 / HBM_PEAK_GBS, "tflops": flops / dt / 1e12,
            "call": "output_final_state=False (the training call)",
            "with_final_state": {"ms": dt_state * 1e3, "frac": nbytes / dt_state / 1e9 / HBM_PEAK_GBS},
            "tokens_per_s": B * T / dt, "timing": f"{reps} back-to-back launches after 1.5 s of the same (settled clock)"}


def measure_chunk_bwd(dev, B=8, H=4, T=4096, Dk=256, Dv=256, reps=100):
    """K2b (three sweeps + dg) at the training shape, bf16 I/O.  Algorithmic bytes: q,k,v,g,do in, dq,dk,dv,dg out."""
    from lina_speech_amd import ops
    g = torch.Generator(device="cpu").manual_seed(0)
    mk = lambda D: torch.randn(B, T, H * D, generator=g).to(torch.bfloat16).to(dev).view(B, T, H, D).transpose(1, 2)
    q, k, v, do = mk(Dk), mk(Dk), mk(Dv), mk(Dv)
    gk = (torch.nn.functional.logsigmoid(torch.randn(B, T, H * Dk, generator=g)) / 16).to(torch.bfloat16).to(dev)
    gk = gk.view(B, T, H, Dk).transpose(1, 2)
    scale = Dk ** -0.5
    nseg = ops.chunk_segments(B * H, T)
    kept = []                    # as in a training step: the segment-parallel forward leaves its boundary states for the backward
    if nseg > 1:
        ops._gla_launch("lina_gla_chunk_fwd", q, k, v, gk, scale, None, False, nseg=nseg, keep_seg_states=kept)
    seg_ws = kept[0][0] if kept else None
    dt, burst = sustained(lambda: ops.gla_chunk_bwd(q, k, v, gk, do, scale, nseg=nseg, seg_states=seg_ws), warm_s=1.0,
                          reps=reps)
    nbytes = B * H * T * 2 * (5 * Dk + 4 * Dv)
    traffic, traffic_src = None, None
    for tname in ("r04_k2b_traffic.json", "r05_k2b_b8_traffic.json", "r06_k2b_traffic.json", "r06_k2b_b8_traffic.json"):   # PMC passes are separate runs; their committed summaries (the last match wins)
        tpath = os.path.join(ROOT, "profiles", tname)
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get("shape") == {"B": B, "H": H, "T": T, "Dk": Dk, "Dv": Dv}:
                traffic = tj["traffic_bytes_per_launch"]
                traffic_src = (f"profiles/{tname} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, gfx950-corrected; per backward "
                               "call = the three sweeps: 18 tensor passes for the 9 algorithmic ones"
                               + (" + the boundary states of the segment-parallel form" if B * H < 256 else "") + ")")
    return {"kernel": "lina::gla_chunk_bf16_h256_kernel<MODE, REV, DG>: reverse sweep (dv, dS) + value-gated sweeps (dq | dk, dg)"
                      + (f", {nseg} sequence segments from boundary states (forward's S, one state-only reverse pass for dS)"
                         if nseg > 1 else ""),
            "shape": {"B": B, "H": H, "T": T, "Dk": Dk, "Dv": Dv},
            "bytes_per_launch": nbytes, "traffic": traffic, "traffic_source": traffic_src,
            "dtype": "bf16 I/O, bf16 MFMA, fp32 accumulate", "ms": dt * 1e3, "ms_burst_of_3": burst * 1e3, "bound": "hbm",
            "achieved": nbytes / dt / 1e9,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nbytes / dt / 1e9 / HBM_PEAK_GBS}


def measure_train_step(dev, b=8, T=4096, steps=10):
    """a-11: one L169 training step (teacher-forced forward, CE loss, backward through K2b/K3b/K5b, fused AdamW) in
    bf16 autocast on one GPU; synthetic config-5 batch."""
    from lina_speech_amd.configs import l169
    from lina_speech_amd.train import TrainStep, synthetic_batch
    torch.manual_seed(0)
    ts = TrainStep(l169(), device=dev, ddp=False)
    batch = synthetic_batch(b=b, n=T + 1, t_txt=T_TXT, seed=1).to(dev)
    import gc
    for _ in range(3):
        ts.step(batch)
    torch.cuda.synchronize()
    # The bench process holds the decode engines' object graphs by now: ONE full (generation-2) collection over them inside the
    # timed region stalled the host for 139 ms -- a 5-step mean of 59.5 ms against 52.3 with those objects frozen and 51.4 for
    # the same step in a fresh process (tools/perf_train_step.py), same box, same session (profiles/r06_train_gc.txt).  Collect
    # now and keep the collector off the OLD objects (young-generation collections still run, and are counted below).
    if os.environ.get("BENCH_TRAIN_GC", "freeze") == "freeze":
        gc.collect()
        gc.freeze()
    gc0 = [g["collections"] for g in gc.get_stats()]
    host = []
    t0 = time.perf_counter()
    for _ in range(steps):
        h0 = time.perf_counter()
        loss = ts.step(batch)
        host.append(time.perf_counter() - h0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    res = {"what": "L169 train step: fwd + CE + bwd + AdamW, bf16 autocast, fp32 master weights", "micro_batch": b,
           "seq_len": T, "ms_per_step": dt * 1e3, "tokens_per_s": b * T / dt, "loss": float(loss),
           "mode": "eager launches", "steps_timed": steps, "max_mem_GB": torch.cuda.max_memory_allocated(dev) / 1e9,
           "host_issue_ms_per_step": sum(host) / steps * 1e3, "host_issue_ms_min_max": [min(host) * 1e3, max(host) * 1e3],
           "gc_collections_in_timed_region": [g["collections"] - a for g, a in zip(gc.get_stats(), gc0)]}
    gc.unfreeze()
    del ts, batch
    torch.cuda.empty_cache()
    return res


def measure_vocoder(dev, B=64, L=750, reps=3):
    """f-3: codes -> 24 kHz waveform for the decode batch (B utterances x 750 tokens = 10 s each), bf16 backbone,
    fp32 ISTFT head; random-init weights of the WavTokenizer-small shape."""
    from lina_speech_amd.vocoder import WavTokenizerDecoder
    torch.manual_seed(0)
    voc = WavTokenizerDecoder().eval().to(dev)
    voc.backbone.to(torch.bfloat16)
    voc.codebook.data = voc.codebook.data.to(torch.bfloat16)
    codes = torch.randint(0, 4096, (1, B, L), device=dev)
    bw = torch.zeros(1, dtype=torch.long, device=dev)
    run = lambda: voc.head(voc.backbone(voc.codes_to_features(codes), bandwidth_id=bw).float())
    with torch.inference_mode():
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            audio = run()
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    assert audio.shape == (B, L * 320) and bool(torch.isfinite(audio).all())
    return {"what": "WavTokenizer-small decode (12 ConvNeXt blocks, dim 768, ISTFT 1280/320), bf16 backbone", "batch": B,
            "tokens_each": L, "ms": dt * 1e3, "codec_tokens_per_s": B * L / dt, "audio_seconds_per_s": B * L / 75.0 / dt}


def measure_config3(eng, dev, B, L=750):
    """BASELINE configs[2] as ONE pipeline: L greedy codec tokens for B utterances on the decode engine -> undelay_rvq - 3
    (clamped) -> WavTokenizer-small decoder (random-init weights, bf16 backbone, fp32 ISTFT head) -> 24 kHz waveform
    (reference model/modeling_lina.py:181-192 -> 3rdparty/decoder/pretrained.py:193-239); wall time of the whole chain."""
    from lina_speech_amd.codec import undelay_rvq
    from lina_speech_amd.vocoder import WavTokenizerDecoder
    torch.manual_seed(0)
    voc = WavTokenizerDecoder().eval().to(dev)
    voc.backbone.to(torch.bfloat16)
    voc.codebook.data = voc.codebook.data.to(torch.bfloat16)
    bw = torch.zeros(1, dtype=torch.long, device=dev)

    def run():
        eng.begin_greedy(L)
        eng.greedy_steps(L)
        codes = (undelay_rvq(eng.greedy_tokens()) - 3).clamp_min(0)
        return voc.head(voc.backbone(voc.codes_to_features(codes), bandwidth_id=bw).float())

    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    audio = run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n_tok = audio.shape[1] // 320
    assert audio.shape[0] == B and bool(torch.isfinite(audio).all())
    del voc
    return {"what": f"decode {L} tokens x {B} utterances -> undelay -> WavTokenizer decoder -> waveform, one chain",
            "seconds": dt, "codec_tokens_per_s": B * L / dt, "audio_seconds_per_s": B * n_tok / 75.0 / dt,
            "real_time_factor_per_utterance": (n_tok / 75.0) / dt}


def physical_cores_per_socket() -> int:
    """Physical cores of ONE socket of this host (/proc/cpuinfo: distinct core ids of physical id 0); os.cpu_count() / 2 if the
    file does not say."""
    try:
        cores, pid = set(), None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pid = int(line.split(":")[1])
            elif line.startswith("core id") and pid == 0:
                cores.add(int(line.split(":")[1]))
        if cores:
            return len(cores)
    except (OSError, ValueError):
        pass
    return max((os.cpu_count() or 2) // 2, 1)


def cpu_baseline(model, B=8, steps=200, repeats=3):
    """The reference's pure-PyTorch recurrent path (mode='naive') restated in oracle/, timed on the host cores on a bounded
    sample of the same workload: same 166.7M weights (fp32), B = 8 rows, T_txt = 64, greedy, a FIXED number of steps, run
    ``repeats`` times from a fresh state; `value` is the MEDIAN run (min and max beside it).  Threads: FIXED at min(32, physical
    cores of one socket) -- the setting that is fastest for these small per-token ops on the GPU box's host (measured there: 32
    threads 156-178 tok/s, 64 threads = one socket's cores 58 tok/s, os.cpu_count() = 256 threads 50 s for ONE step); rounds 1-5
    probed two settings per run and took the faster -- a third of the bench's wall time for a figure with a 4 x spread.
    Also times BASELINE config 1 (d256 x l2 simple-GLA forward, B=4, T=256, pure-PyTorch recurrent) on the same cores."""
    from oracle.lina_decode_oracle import OracleLina
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    orc = OracleLina(sd, n_layer=6, heads=4, txt_heads=4)
    g = torch.Generator().manual_seed(0)
    x = torch.randint(3, 256, (B, T_TXT), generator=g)
    threads = min(32, physical_cores_per_socket(), os.cpu_count() or 1)
    torch.set_num_threads(threads)
    runs = []
    with torch.no_grad():
        x_enc = orc.text_encoder(x)
        for rep in range(repeats):
            state = orc.init_state(B)
            y = orc.embed(torch.ones(1, B, 1, dtype=torch.long))
            for _ in range(40 if rep == 0 else 1):                      # untimed warm-up steps (the first run of a session read 46 tok/s
                                                                        # against 147 / 153 after THREE warm-up steps: thread pool + clocks still cold)
                logits, _ = orc.step(y, x_enc, state)
                y = orc.embed(logits[:, 0].argmax(-1).t().unsqueeze(-1))
            n, t0 = 0, time.time()
            while n < steps and (n < 8 or time.time() - t0 < 40.0):     # (bounded: a loaded host still ends the run)
                logits, _ = orc.step(y, x_enc, state)
                y = orc.embed(logits[:, 0].argmax(-1).t().unsqueeze(-1))
                n += 1
            dt = time.time() - t0
            runs.append({"threads": threads, "tokens_per_s": B * n / dt, "steps": n, "seconds": dt})
    rates = sorted(r["tokens_per_s"] for r in runs)
    med = rates[len(rates) // 2]
    cfg1 = cpu_config1()
    return {"value": med, "unit": "codec tokens/s", "cores": threads, "kind": "port", "min": rates[0], "max": rates[-1],
            "spread": rates[-1] / max(rates[0], 1e-9), "host": host_cpu(), "runs": runs,
            "sample": f"oracle/lina_decode_oracle.py (pure-PyTorch recurrent, fp32), 166.7M model, B={B}, T_txt={T_TXT}, "
                      f"{runs[0]['steps']} greedy steps x {repeats} runs at {threads} threads (fixed: min(32, the {physical_cores_per_socket()} physical "
                      f"cores of one socket); os.cpu_count() = {os.cpu_count()}); value = the median run",
            "config1_cpu": cfg1}


def cpu_config1(B=4, T=256, d=256, heads=4, n_blocks=3, reps=3):
    """BASELINE configs[0] on the host: d256 x l2 simple-GLA forward (2 GLA blocks + the pos_net block = 3 scalar-gate
    mixers with short convolutions), B=4, T=256, through the oracle's pure-PyTorch recurrent layer (the stand-in the goldens
    of tests/golden/simple_gla_d256.npz were generated with).  Mixer stack only: the timing check BASELINE.md section 2 names."""
    from oracle.fla_standin import SimpleGatedLinearAttention
    torch.manual_seed(0)
    layers = [SimpleGatedLinearAttention(hidden_size=d, num_heads=heads, use_short_conv=True, layer_idx=i).eval()
              for i in range(n_blocks)]
    x = torch.randn(B, T, d)
    with torch.no_grad():
        def fwd():
            h = x
            for lay in layers:
                h = h + lay(h)[0]
            return h
        fwd()
        t0 = time.time()
        for _ in range(reps):
            fwd()
        dt = (time.time() - t0) / reps
    return {"what": f"{n_blocks} simple-GLA mixers d={d} H={heads}, B={B}, T={T}, pure-PyTorch recurrent (oracle/fla_standin.py)",
            "ms_per_forward": dt * 1e3, "tokens_per_s": B * T / dt, "threads": torch.get_num_threads()}


SUSTAINED_S = 0.25       # secondary "sustained" figure: the same loop over at least this much wall time (not `value`)


def relaunch_under_torchrun(n: int, argv) -> int:
    """``python bench.py --gpus N`` with N > 1 and no launcher around it: start N ranks of this script under
    torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1) and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
    log("bench.py: --gpus", n, "without a launcher -> ", " ".join(cmd))
    return subprocess.call(cmd, env=env)


def init_world(n_gpus: int):
    """Join the process group the launcher describes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*): backend "nccl" (= RCCL)
    on GPUs, "gloo" on a CPU-only host (launch check only).  The world MUST be --gpus ranks."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != n_gpus:
        raise SystemExit(f"bench.py: --gpus {n_gpus} but the launcher started WORLD_SIZE={world} ranks")
    gpu = torch.cuda.is_available()
    dev = torch.device("cuda", local) if gpu else torch.device("cpu")
    if gpu:
        if local >= torch.cuda.device_count():
            raise SystemExit(f"bench.py: rank {rank} needs GPU {local}, the node shows {torch.cuda.device_count()}")
        torch.cuda.set_device(dev)
    dist = None
    if world > 1 or os.environ.get("LINA_BENCH_FORCE_PG"):            # (forced at world 1: RCCL init next to graph capture)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if gpu:
            # RCCL's own account of what it set up / chose (channels, transports, algorithm and protocol per collective):
            # INFO lines of this rank go to a file that rank 0 summarises into the JSON line (ranks.rccl)
            os.environ.setdefault("NCCL_DEBUG", "INFO")
            os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,COLL,TUNING,GRAPH")
            os.environ.setdefault("NCCL_DEBUG_FILE", f"/tmp/lina_rccl_{os.getpid()}_rank{rank}.log")
            dist.init_process_group("nccl", device_id=dev)           # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group("gloo")
        assert dist.get_world_size() == n_gpus and dist.get_rank() == rank
    return rank, local, world, dev, dist


def rccl_summary(max_lines=12):
    """What RCCL logged for THIS rank (NCCL_DEBUG=INFO file set up in init_world): version, channel / transport lines and
    the algorithm + protocol it picked per collective -- so a multi-GPU line says which ring / tree actually ran."""
    import re
    try:
        lib_version = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        lib_version = None
    path = os.environ.get("NCCL_DEBUG_FILE", "")
    if not path or not os.path.exists(path):
        # (a world of one rank never builds a communicator: nothing is logged)
        return {"library_version": lib_version, "log_lines": 0, "note": "no RCCL log for this rank (no communicator was built)"}
    txt = open(path, errors="replace").read().splitlines()
    pick = lambda pat: [ln.split("NCCL INFO", 1)[-1].strip() for ln in txt if re.search(pat, ln)]
    algo = pick(r"[Aa]lgo(rithm)?\b.*[Pp]roto|AllReduce.*(Ring|Tree|RING|TREE)|-> algo")
    out = {"library_version": lib_version, "log_lines": len(txt),
           "version": (pick(r"RCCL version|NCCL version") or [None])[0],
           "channels": (pick(r"coll channels|nChannels|Channel 00") or [None])[0],
           "transports": sorted({m.group(0) for ln in txt for m in [re.search(r"via [A-Za-z0-9/_\-]+", ln)] if m})[:8],
           "algo_proto": sorted(set(algo))[:max_lines]}
    counts = {}
    for ln in algo:
        m = re.search(r"(Ring|Tree|CollNet|NVLS|RING|TREE)[^A-Za-z]*(LL128|LL|Simple|SIMPLE)?", ln)
        if m:
            key = "/".join(x for x in m.groups() if x)
            counts[key] = counts.get(key, 0) + 1
    out["algo_proto_counts"] = counts
    return out


def per_rank_ms(elapsed_local, k, dist, dev):
    """Each rank's own wall time per step over the timed region (ms): min / max / all -- the spread between ranks."""
    if dist is None:
        return None
    t = torch.tensor([elapsed_local / k * 1e3], device=dev, dtype=torch.float64)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    v = [float(x.item()) for x in out]
    return {"min": min(v), "max": max(v), "all": v}


def check_launch(args):
    """--check-launch: only the launch / rendezvous logic (runs on a CPU-only host with gloo): every rank joins the
    group, contributes its rank to an all-reduce and rank 0 prints what it saw."""
    rank, local, world, dev, dist = init_world(args.gpus)
    seen = world
    total = args.total_batch if args.batch_per_gpu is None else args.batch_per_gpu * world
    if dist is not None:
        t = torch.ones(1, device=dev)
        dist.all_reduce(t)
        seen = int(t.item())
        from lina_speech_amd.shard import shard_rows
        lo, hi = shard_rows(total, rank, world)
        rows = torch.tensor([float(hi - lo)], device=dev)
        dist.all_reduce(rows)
        assert int(rows.item()) == total
    if rank == 0:
        print(json.dumps({"check_launch": True, "n_gpus": args.gpus, "world_size": world, "ranks_seen": seen,
                          "backend": (dist.get_backend() if dist is not None else None),
                          "rows_total": total, "scaling": "strong" if args.batch_per_gpu is None else "weak"}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def timed_steps(run_step, k_req, dist, dev, est_steps=None, batched=False):
    """Time EXACTLY k_req steps (or est_steps if that is larger: the callers pass it only for --min-timed-s > 0) bracketed
    by barrier + synchronize on both sides; returns (elapsed seconds = MAX over ranks, steps timed)."""
    k = k_req
    if est_steps is not None:
        k = max(k_req, est_steps)
    if dist is not None:
        kk = torch.tensor([k], device=dev, dtype=torch.int64)
        dist.all_reduce(kk, op=dist.ReduceOp.MAX)
        k = int(kk.item())
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if batched:
        run_step(k)                           # exactly k steps, enqueued as multi-token graph replays + a remainder
    else:
        for _ in range(k):
            run_step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timed_steps.local_elapsed = elapsed          # this rank's own clock (per_rank_ms)
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    return elapsed, k


def bench_train(args):
    """--train: config 5 -- the L169 training step (teacher-forced forward, CE, backward through K2b/K3b/K5b, AdamW;
    bf16 autocast, fp32 master weights) with b = 8 sequences of 4096 tokens PER GPU, data parallel over RCCL
    (DistributedDataParallel, 128 MB buckets, gradient all-reduce overlapped with backward).  One JSON line."""
    rank, local, world, dev, dist = init_world(args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (there is no CPU fallback for the product path)")
    from lina_speech_amd import ops
    from lina_speech_amd.configs import l169, n_params
    from lina_speech_amd.train import TrainStep, synthetic_batch
    ops.get_backend().lib
    b, T = args.train_batch, 4096
    torch.manual_seed(0)
    model = l169()
    nparam = n_params(model)
    ts = TrainStep(model, device=dev, ddp=world > 1)
    batch = synthetic_batch(b=b, n=T + 1, t_txt=T_TXT, seed=1 + rank).to(dev)
    for _ in range(max(args.warmup, 2)):
        loss = ts.step(batch)
    torch.cuda.synchronize()
    t_est = time.perf_counter()
    ts.step(batch)
    torch.cuda.synchronize()
    est = time.perf_counter() - t_est
    elapsed, k = timed_steps(lambda: ts.step(batch), args.steps, dist, dev,
                             est_steps=(int(args.min_timed_s / est) + 1) if args.min_timed_s > 0 else None)
    ranks_ms = per_rank_ms(timed_steps.local_elapsed, k, dist, dev)
    loss = float(ts.step(batch))
    if rank == 0:
        ranks = None
        if dist is not None:
            ranks = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "per_rank_ms": ranks_ms,
                     "collectives_in_timed_region": "gradient all-reduce (DDP buckets) per step + barrier",
                     "rccl": rccl_summary()}
        print(json.dumps({
            "metric": "training tokens/sec (whole node), 169M d1024xl12 train step, seqlen 4096",
            "value": world * b * T * k / elapsed, "unit": "tokens/s", "n_gpus": world, "steps": k,
            "steps_requested": args.steps, "warmup": max(args.warmup, 2), "ms_per_step": elapsed / k * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"L169 train step (fwd + CE + bwd + AdamW), micro-batch {b} x {T} tokens per GPU, "
                                   f"{nparam / 1e6:.1f}M params, bf16 autocast, fp32 master weights and state",
                       "global_batch": b * world, "seq_len": T,
                       "parallelism": f"dp{world} (DistributedDataParallel over RCCL, 128 MB buckets)" if world > 1
                       else "single GPU"},
            "min_timed_s": args.min_timed_s, "ranks": ranks,
            "loss": loss, "max_mem_GB": torch.cuda.max_memory_allocated(dev) / 1e9}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def step_hbm_bytes(eng):
    """HBM bytes of one decode step of `eng`: recurrent state read every token (+ written every W-th with the windowed
    update), every decode-time weight read once, text-side K/V rows read once.  Returns (state, weights, text K/V,
    state bytes of the immediate form = read AND written every token)."""
    w_bytes = sum(t.numel() * t.element_size() for P_ in eng.packs for t in
                  (P_.w_in, P_.w_o, P_.w_up, P_.w_down)) + eng.w_head.numel() * eng.w_head.element_size() \
        + eng.ca_qw.numel() * eng.ca_qw.element_size()
    s_bytes = int(sum((1.0 + 1.0 / (eng.window if P_.lazy else 1)) * P_.S.numel() * P_.S.element_size() for part in eng.parts
                      for P_ in part.packs))
    kv_bytes = sum(part.kk.numel() * part.kk.element_size() + part.vv.numel() * part.vv.element_size()
                   for part in eng.parts)
    s_bytes_imm = sum(2 * P_.S.numel() * P_.S.element_size() for part in eng.parts for P_ in part.packs)
    return s_bytes, w_bytes, kv_bytes, s_bytes_imm


def measure_batch(model_dev, dev, B=512, steps=120):
    """The same engine and device loop at another per-GPU batch (secondary blocks beside the headline): ms per token, tok/s and
    the step against its HBM bytes."""
    from lina_speech_amd.decode import DecodeEngine
    g = torch.Generator().manual_seed(4321)
    texts = torch.randint(3, 256, (B, T_TXT), generator=g).to(dev)
    with torch.inference_mode():
        eng = DecodeEngine(model_dev, model_dev.txt_encoder(model_dev.txt_embed(texts)), batch_size=B)
        eng.begin_greedy(steps + 80, log_att=True)
        eng.greedy_steps(16)                     # captures the multi-token graph
        eng.greedy_steps(48)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.greedy_steps(steps)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        toks = eng.greedy_tokens()
        assert int(toks.min()) >= 0 and int(toks.max()) < 4099
        s_b, w_b, kv_b, s_imm = step_hbm_bytes(eng)
        mem = torch.cuda.max_memory_allocated(dev) / 1e9
        del eng
    import gc
    gc.collect()                                 # (an engine is a reference cycle: free its graphs and buffers now)
    torch.cuda.empty_cache()
    nb = s_b + w_b + kv_b
    return {"what": f"the same decode loop with B = {B} rows on ONE GPU (secondary, not `value`)",
            "batch": B, "steps_timed": steps, "ms_per_step": dt * 1e3, "tokens_per_s": B / dt,
            "step_roofline": {"state_bytes": s_b, "weight_bytes": w_b, "text_kv_bytes": kv_b, "bytes_per_step": nb,
                              "bound": "hbm", "achieved": nb / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": nb / dt / 1e9 / HBM_PEAK_GBS,
                              "immediate_form_frac": (w_b + s_imm + kv_b) / dt / 1e9 / HBM_PEAK_GBS},
            "max_mem_GB": mem}


def measure_bf16_state(model_dev, texts, dev, steps=160):
    """SECONDARY line (never `value`): the opt-in reference-dtype recurrent state, ``DecodeEngine(state_dtype=torch.bfloat16)``.
    The reference keeps the state of a bf16 model in bf16 between decode steps (model/gla.py:229-240 + Cache.update: rounded
    after every step); the product's default -- and the headline -- is an fp32 state.  window 1 = the reference's arithmetic
    (state read AND written every token, 2 + 2 bytes per element); window 8 = bf16 storage rounded every 8th token (2 (1 + 1/8)
    bytes per element per token: half of the fp32 K1w's).  One engine of all the rows."""
    from lina_speech_amd.decode import DecodeEngine
    B = texts.shape[0]
    res = {"what": "opt-in bf16 recurrent state (the reference's state dtype for a bf16 model); secondary, the headline keeps fp32 state",
           "batch": B}
    with torch.inference_mode():
        x_enc = model_dev.txt_encoder(model_dev.txt_embed(texts))
        for window in (1, 8):
            eng = DecodeEngine(model_dev, x_enc, batch_size=B, state_dtype=torch.bfloat16, window=window)
            eng.begin_greedy(steps + 80, log_att=True)
            eng.greedy_steps(64)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.greedy_steps(steps)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            toks = eng.greedy_tokens()
            assert int(toks.min()) >= 0 and int(toks.max()) < 4099
            s_b, w_b, kv_b, s_imm = step_hbm_bytes(eng)
            nb = s_b + w_b + kv_b
            res[f"window_{window}"] = {
                "what": ("state rounded to bf16 after every token: the reference's own arithmetic for a bf16 model" if window == 1
                         else "state stored in bf16, rounded every 8th token (K1w window)"),
                "ms_per_step": dt * 1e3, "tokens_per_s": B / dt,
                "step_roofline": {"state_bytes": s_b, "weight_bytes": w_b, "text_kv_bytes": kv_b, "bytes_per_step": nb, "bound": "hbm",
                                  "achieved": nb / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nb / dt / 1e9 / HBM_PEAK_GBS}}
            eng.close()
            del eng
            torch.cuda.empty_cache()
    return res


def measure_generate_batch(model_dev, texts, dev, loop_ms, max_seqlen=750):
    """The reference's decode ENTRY POINT end to end (model/modeling_lina.py:111-192): ``LinaModel.generate_batch(texts,
    max_seqlen=750, force_max_seqlen=True)`` with no engine argument -- text embedding + text encoder, the text side of the
    cross-attention, state reset, the device loop (what `value` times), token / attention-log read-out, stop-flag matrix,
    un-delay and the per-row cuts -- wall time of the whole call on a warm engine cache (the first call of a (batch, text
    length) builds the engine: packs the weights, captures two hipGraphs; reported as `first_call_s`).  Greedy and the
    reference's default sampled mode (k = 100, first quantizer sampled); plus one EARLY-STOPPING call on a copy of the model
    whose stop-token head row is scaled up (x 4) so that every row emits token 2 within ~100 steps (random-init weights never do)."""
    import copy
    B = texts.shape[0]
    res = {"what": f"LinaModel.generate_batch(x, batch_size={B}, max_seqlen={max_seqlen}, force_max_seqlen=True, device=...) "
                   "end to end, engine argument left at its default", "batch": B, "max_seqlen": max_seqlen}
    with torch.inference_mode():
        kw = dict(batch_size=B, max_seqlen=max_seqlen, device=dev, force_max_seqlen=True)
        t0 = time.perf_counter()
        model_dev.generate_batch(texts, k=1, first_greedy_quant=0, **{**kw, "max_seqlen": 64})
        torch.cuda.synchronize()
        res["first_call_s"] = time.perf_counter() - t0
        for name, mode in (("greedy", dict(k=1, first_greedy_quant=0)), ("sampled_k100", dict(k=100, first_greedy_quant=1))):
            model_dev.generate_batch(texts, **mode, **{**kw, "max_seqlen": 64})          # this mode's graphs (untimed)
            torch.cuda.synchronize()
            best = None
            for _ in range(3):
                t0 = time.perf_counter()
                qs, atts, stops, cuts = model_dev.generate_batch(texts, **mode, **kw)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            assert qs.shape == (1, B, max_seqlen) and atts.shape[:3] == (B, 2, max_seqlen) and stops.shape == (B, max_seqlen + 1)
            res[name] = {"seconds": best, "tokens_per_s": B * max_seqlen / best, "ms_per_step_end_to_end": best / max_seqlen * 1e3,
                         "loop_ms_per_step": loop_ms, "vs_loop": (best / max_seqlen * 1e3) / loop_ms if loop_ms else None}
        # early stop: scale the stop token's head row (logit_2 = s * <h, w_2>: positive and dominant at random steps)
        m2 = copy.deepcopy(model_dev)
        m2.logits_head.weight[0, 2] *= 4.0
        m2.generate_batch(texts, batch_size=B, max_seqlen=64, k=1, first_greedy_quant=0, device=dev, force_max_seqlen=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        qs, atts, stops, cuts = m2.generate_batch(texts, batch_size=B, max_seqlen=max_seqlen, k=1, first_greedy_quant=0, device=dev)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        eng = next(reversed(m2._decode_engines.values()))
        n = qs.shape[-1]
        res["early_stop"] = {"what": "stop-token head row x 4; force_max_seqlen=False, stop_check_every=16 (default)",
                             "steps_returned": n, "steps_executed": eng._n_done, "seconds": dt,
                             "ms_per_returned_step": dt / n * 1e3, "stopped_early": n < max_seqlen,
                             "cut_lengths_min_max": [min(c[0].shape[-1] for c in cuts), max(c[0].shape[-1] for c in cuts)]}
        m2.clear_decode_cache()
        del m2
    model_dev.clear_decode_cache()
    torch.cuda.empty_cache()
    return res


def measure_two_engines(model_dev, texts, dev, steps=240, max_seqlen=750):
    """The same utterance batch on TWO engines (half the rows each, shared packed weights) driven on two HIP streams
    (decode.DecodeEngineGroup; ``generate_batch(..., n_engines=2)``): one half's projections run under the other half's
    HBM-bound state update.  Secondary figure (opt-in: the default entry point uses one engine)."""
    from lina_speech_amd.decode import DecodeEngineGroup
    B = texts.shape[0]
    with torch.inference_mode():
        grp = DecodeEngineGroup(model_dev, model_dev.txt_encoder(model_dev.txt_embed(texts)), batch_size=B, n_engines=2)
        grp.begin_greedy(steps + 80, log_att=True)
        grp.greedy_steps(64)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        grp.greedy_steps(steps)
        toks = grp.greedy_tokens()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        assert toks.shape == (1, B, steps + 64) and int(toks.min()) >= 0 and int(toks.max()) < 4099
        del grp
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        kw = dict(batch_size=B, max_seqlen=max_seqlen, device=dev, force_max_seqlen=True, k=1, first_greedy_quant=0, n_engines=2)
        model_dev.generate_batch(texts, **{**kw, "max_seqlen": 64})
        torch.cuda.synchronize()
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            qs, atts, stops, cuts = model_dev.generate_batch(texts, **kw)
            torch.cuda.synchronize()
            g = time.perf_counter() - t0
            best = g if best is None else min(best, g)
        assert qs.shape == (1, B, max_seqlen) and atts.shape[:3] == (B, 2, max_seqlen)
    model_dev.clear_decode_cache()
    torch.cuda.empty_cache()
    return {"what": f"{B} rows on 2 engines x {B // 2} rows, one HIP stream each (DecodeEngineGroup); secondary, not `value`",
            "loop_ms_per_step": dt * 1e3, "loop_tokens_per_s": B / dt,
            "generate_batch_tokens_per_s": B * max_seqlen / best, "generate_batch_seconds": best}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--preheat-s", type=float, default=1.0, help="seconds of untimed decode before the W warm-up steps")
    ap.add_argument("--min-timed-s", type=float, default=0.0,
                    help="extend the timed region beyond --steps to at least this much wall time (default 0: EXACTLY --steps "
                         "steps are timed; a >= 0.25 s figure is always reported beside it as `sustained`)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--train", action="store_true", help="measure config 5 (the DDP training step) instead of decode")
    ap.add_argument("--train-batch", type=int, default=8, help="--train: sequences of 4096 tokens per GPU")
    ap.add_argument("--check-launch", action="store_true", help="only exercise the N-rank launch / rendezvous logic")
    ap.add_argument("--total-batch", type=int, default=TOTAL_BATCH,
                    help="utterance rows of the whole job, sharded over the GPUs (default 512 = the batch BASELINE's metric is "
                         "quoted on: strong scaling, 512 / N rows per GPU)")
    ap.add_argument("--batch-per-gpu", type=int, default=None,
                    help="instead of --total-batch: this many rows on EVERY GPU (weak scaling; 64 = BASELINE configs[1])")
    ap.add_argument("--engines", type=int, default=0,
                    help="engines (row ranges, one HIP stream each) per GPU; 0 = what generate_batch picks for the batch (2 from 512 rows up)")
    ap.add_argument("--window", type=int, default=None, help="state window of the decode loop (1 = immediate update K1d; default 8)")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the timed decode loop (no secondary blocks): what a rocprofv3 trace of the headline should contain")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-chunk", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the secondary training-step measurement")
    args = ap.parse_args()
    if args.headline_only:
        args.no_cpu_baseline = args.no_chunk = args.no_train = True

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(relaunch_under_torchrun(args.gpus, sys.argv[1:]))
    if args.check_launch:
        return check_launch(args)
    if args.train:
        return bench_train(args)

    rank, local, world, dev, dist = init_world(args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (there is no CPU fallback for the product path)")

    from lina_speech_amd import ops
    from lina_speech_amd.configs import l169, n_params
    from lina_speech_amd.decode import DecodeEngine
    ops.get_backend().lib                                            # fail loudly if the HIP library is missing

    from lina_speech_amd.shard import shard_rows
    weak = args.batch_per_gpu is not None
    total_rows = args.batch_per_gpu * world if weak else args.total_batch
    if total_rows < world:
        raise SystemExit(f"bench.py: {total_rows} utterance rows cannot be sharded over {world} GPUs")
    lo, hi = shard_rows(total_rows, rank, world)                     # this rank's utterances
    B = hi - lo
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32

    torch.manual_seed(0)
    model = l169().eval()
    nparam = n_params(model)
    model_dev = model.to(dev, dtype)
    g = torch.Generator().manual_seed(1234)
    texts = torch.randint(3, 256, (total_rows, T_TXT), generator=g)[lo:hi].to(dev)   # one distinct text per row

    with torch.inference_mode():
        x_enc = model_dev.txt_encoder(model_dev.txt_embed(texts))
        # the loop LinaModel.generate_batch runs for this many rows: from LinaModel.AUTO_TWO_ENGINES_ROWS (384) rows up the batch is
        # cut into two row ranges, one engine and one HIP stream each (decode.DecodeEngineGroup; --engines overrides)
        n_eng = args.engines if args.engines else (2 if B >= model_dev.AUTO_TWO_ENGINES_ROWS else 1)
        if n_eng > 1:
            from lina_speech_amd.decode import DecodeEngineGroup
            eng = DecodeEngineGroup(model_dev, x_enc, batch_size=B, n_engines=n_eng, window=args.window)
            eng1 = eng.engines[0]                 # (attribute reads, and the engine the update kernel is timed in)
        else:
            eng = eng1 = DecodeEngine(model_dev, x_enc, batch_size=B, window=args.window)
        # settle the engine clock first (untimed, outside the W warm-up steps): the chip ramps up from idle over the first
        # fraction of a second of load
        t_pre = time.perf_counter()
        n_pre = 0
        while time.perf_counter() - t_pre < args.preheat_s or n_pre == 0:
            eng.begin_greedy(150, log_att=True)
            eng.greedy_steps(96)
            torch.cuda.synchronize()
            n_pre += 96
        t_est = time.perf_counter()                                  # step time estimate on the captured graph
        eng.greedy_steps(48)
        torch.cuda.synchronize()
        est = (time.perf_counter() - t_est) / 48
        k_run = max(args.steps, int(args.min_timed_s / est) + 1) if args.min_timed_s > 0 else args.steps
        k_sus = max(args.steps, int(SUSTAINED_S / est) + 1)
        if dist is not None:
            kk = torch.tensor([k_run, k_sus], device=dev, dtype=torch.int64)   # every rank sizes its token log for the same counts
            dist.all_reduce(kk, op=dist.ReduceOp.MAX)
            k_run, k_sus = int(kk[0].item()), int(kk[1].item())
        # the loop generate_batch runs: picks, stop flags, the attention log and the next-token embedding inside the step
        eng.begin_greedy(k_run + k_sus + args.warmup + 8, log_att=True)
        eng.greedy_steps(8)                                           # captures the multi-token graph (untimed)
        eng.greedy_steps(args.warmup)
        elapsed, k_run = timed_steps(lambda n: eng.greedy_steps(n), k_run, dist, dev, batched=True)
        ranks_ms = per_rank_ms(timed_steps.local_elapsed, k_run, dist, dev)
        el_sus, k_sus = timed_steps(lambda n: eng.greedy_steps(n), k_sus, dist, dev, batched=True)   # secondary figure
        toks = eng.greedy_tokens()
        assert toks.shape == (1, B, k_run + k_sus + args.warmup + 8)
        assert int(toks.min()) >= 0 and int(toks.max()) < 4099

        out = None
        if rank == 0:
            P = eng1.packs[0]
            k1_b2b, k1_entry = measure_k1(eng1)                    # back to back with itself (conservative)
            k1_dt, k1_n, ms_with, ms_without = eng1.time_update_kernel()   # (two engines: in ONE engine's step graph, the other idle)
            e_io = 2 if dtype == torch.bfloat16 else 4
            k1_rows = P.S.shape[0]
            lazy = k1_entry == "lina_gla_decode_window"
            k1d_bytes = k1_algorithmic_bytes(k1_rows, P.H, P.Dk, P.Dv, e_io, 4)
            k1_bytes = k1w_algorithmic_bytes(k1_rows, P.H, P.Dk, P.Dv, e_io, 4, eng1.window) if lazy else k1d_bytes
            traffic, traffic_src = None, None
            names = ("r06_k1w_traffic_b%d.json" % k1_rows, "r05_k1w_traffic_b%d.json" % k1_rows, "r04_k1w_traffic.json", "r02_k1w_traffic.json") if lazy \
                else ("r02_k1d_traffic.json", "r01_k1d_traffic.json")
            for name in names:                                # PMC passes are separate runs; their committed summary
                tpath = os.path.join(ROOT, "profiles", name)
                if os.path.exists(tpath) and dtype == torch.bfloat16:
                    tj = json.load(open(tpath))
                    if tj.get("rows_per_launch", 64) != k1_rows:
                        continue
                    traffic = tj["traffic_bytes_per_launch"]
                    traffic_src = f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, gfx950-corrected)"
                    break
            kname = {"lina_gla_decode_window": "lina::gla_decode_window_kernel<256, 4> (K1w + K5, one workgroup per head, lina_gla_decode_window)",
                     "lina_gla_decode_update_norm": "lina::gla_decode_rowsplit_kernel<256, FUSE=true> (lina_gla_decode_update_norm)",
                     "lina_gla_decode_update": "lina::gla_decode_rowsplit_kernel<256, FUSE=false> (lina_gla_decode_update)"}[k1_entry]
            roof = {"kernel": kname, "bound": "hbm",
                    "achieved": k1_bytes / k1_dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": k1_bytes / k1_dt / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                    "bytes_per_launch": k1_bytes, "us_per_launch": k1_dt * 1e6, "rows_per_launch": k1_rows,
                    "launches_per_step": len(eng1.packs) * len(eng1.parts) * n_eng,
                    "timing": f"in situ: (step graph {ms_with:.4f} ms - the same graph without its {k1_n} update launches "
                              f"{ms_without:.4f} ms) / {k1_n}, HIP events around 160 replays each",
                    "us_per_launch_back_to_back": k1_b2b * 1e6,
                    "frac_back_to_back": k1_bytes / k1_b2b / 1e9 / HBM_PEAK_GBS}
            if lazy:
                roof["window"] = eng1.window
                roof["bytes_definition"] = ("state read every step + written every W-th (4 H Dk Dv (1 + 1/W)) + q,k,v,gk,o "
                                            "+ window history; averaged over the W window positions (DESIGN 4.1)")
                roof["immediate_form"] = {"what": "SURVEY 8(d) bytes of the immediate update K1 (state read AND written "
                                                  "every step) over the same time: the traffic this kernel avoids",
                                          "bytes_per_launch": k1d_bytes,
                                          "equivalent_GBs": k1d_bytes / k1_dt / 1e9}
            # the step as a whole against HBM: recurrent state read + written once per block, every decode-time weight
            # read once (the bytes the engine's packs actually hold), text-side K/V rows read once
            ms_step = elapsed / k_run * 1e3
            engs = eng.engines if n_eng > 1 else [eng]
            parts_b = [step_hbm_bytes(e_) for e_ in engs]
            s_bytes, kv_bytes, s_bytes_imm = (sum(p_[i] for p_ in parts_b) for i in (0, 2, 3))
            w_bytes = parts_b[0][1] * (n_eng if B < 256 else 1)      # (packed weights are shared; at a large batch the second engine's reads hit the Infinity Cache or not: priced ONCE)
            step_bytes = w_bytes + s_bytes + kv_bytes
            step_roof = {"what": "whole decode step vs HBM: state read (+ write every W-th step) + decode-time weights + text K/V, per step per GPU",
                         "immediate_form": {"what": "the same step priced with the state read AND written every token "
                                                    "(SURVEY 8(d) / VERDICT r01 accounting: the bytes K1w avoids still counted)",
                                            "bytes_per_step": w_bytes + s_bytes_imm + kv_bytes,
                                            "frac": (w_bytes + s_bytes_imm + kv_bytes) / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
                         "state_bytes": s_bytes, "weight_bytes": w_bytes, "text_kv_bytes": kv_bytes,
                         "bytes_per_step": step_bytes, "ms_per_step": ms_step, "bound": "hbm",
                         "achieved": step_bytes / (ms_step * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": step_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS}
            out = {
                "metric": "codec tokens/sec (whole node), 169M d1024xl12 batched greedy decode",
                "value": total_rows * k_run / elapsed, "unit": "codec tokens/s", "n_gpus": world,
                "steps": k_run, "steps_requested": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
                "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None,
                "dtype": "bf16" if dtype == torch.bfloat16 else "f32", "data": "synthetic",
                "config": {"workload": f"L169 greedy codec-token decode, B_total={total_rows} ({B} rows on this GPU" + (", fixed per GPU" if weak else f" = B_total / {world}") + "), "
                                       f"T_txt={T_TXT}, H=4 Dk=Dv=256, 12+1 GLA blocks, fp32 recurrent state, "
                                       f"{nparam / 1e6:.1f}M params, one hipGraph replay per {eng1.GRAPH_STEPS} tokens, state window {eng1.window}, "
                                       + (f"{n_eng} engines of {B // n_eng} rows on {n_eng} HIP streams (shared packed weights)" if n_eng > 1
                                          else "one engine"),
                           "global_batch": total_rows, "rows_per_gpu": B, "engines_per_gpu": n_eng,
                           "loop": "the device loop LinaModel.generate_batch runs (picks, stop flags, attention log, next-token "
                                   "embedding inside the captured step)",
                           "parallelism": f"batch-shard x{world} (no collective)",
                           "timed_region": f"exactly {k_run} steps" + (f" (--min-timed-s {args.min_timed_s})" if args.min_timed_s > 0 else "")},
                "min_timed_s": args.min_timed_s,
                "sustained": {"what": f"the same loop over >= {SUSTAINED_S} s right after the timed region (secondary; not `value`)",
                              "steps": k_sus, "ms_per_step": el_sus / k_sus * 1e3, "tokens_per_s": total_rows * k_sus / el_sus},
                "roofline": roof, "step_roofline": step_roof,
            }
            if dist is not None:
                out["ranks"] = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "per_rank_ms": ranks_ms,
                                "collectives_in_timed_region": "barrier only (batch shard, no data-path collective)",
                                "rccl": rccl_summary()}
            # secondary, untimed-region measurements (rank 0 only): the default sampling mode of the reference
            # (top-k 100, temperature) through the same graph, K2 / K2b at the training shape
            if world == 1 and not args.headline_only:
                eng.begin_greedy(80, k=100, temp=1.0, seed=1, first_greedy_quant=1)
                eng.greedy_steps(16)
                torch.cuda.synchronize()
                ts0 = time.perf_counter()
                eng.greedy_steps(48)
                torch.cuda.synchronize()
                out["sampled_decode"] = {"k": 100, "temp": 1.0, "ms_per_step": (time.perf_counter() - ts0) / 48 * 1e3,
                                         "tokens_per_s": B * 48 / (time.perf_counter() - ts0)}
            per = {}
            if not args.no_chunk and world == 1 and dtype == torch.bfloat16:
                for bb in (64, 128, 256, 512):                   # the same loop at the other per-GPU batches
                    if bb == B and n_eng == 1:           # (two engines: the one-engine loop of the same rows stays beside the headline)
                        continue
                    try:
                        per[f"B={bb}"] = measure_batch(model_dev, dev, B=bb)
                    except Exception as e:                       # (a secondary block must not cost the headline line)
                        per[f"B={bb}"] = {"error": repr(e)}
                out["per_gpu_batch"] = per
            if world == 1 and dtype == torch.bfloat16 and not args.headline_only:
                # the reference's entry point end to end, on the headline batch and on configs[1]'s 64 rows
                gb = {}
                for bb in sorted({B, 64}):
                    loop_ms = ms_step if bb == B else per.get(f"B={bb}", {}).get("ms_per_step")
                    try:
                        gb[f"B={bb}"] = measure_generate_batch(model_dev, texts[:bb] if bb <= B else texts, dev, loop_ms)
                    except Exception as e:
                        gb[f"B={bb}"] = {"error": repr(e)}
                out["generate_batch"] = gb
                if B >= 256 and n_eng == 1:
                    try:
                        out["two_engines"] = measure_two_engines(model_dev, texts, dev)
                    except Exception as e:
                        out["two_engines"] = {"error": repr(e)}
                try:
                    out["decode_bf16_state"] = measure_bf16_state(model_dev, texts, dev)
                except Exception as e:
                    out["decode_bf16_state"] = {"error": repr(e)}
            if not args.no_chunk and world == 1:
                out["config3_pipeline"] = measure_config3(eng, dev, B)
                out["chunk_kernel"] = measure_chunk(dev)
                for hh in (8, 16):                                   # the same width as 8 / 16 heads: 2 / 4 heads per workgroup
                    ck = measure_chunk(dev, B=64, H=hh, Dk=1024 // hh, Dv=1024 // hh, reps=100)
                    ck["kernel"] = f"lina::gla_chunk_bf16_h256_kernel<false, G={256 * hh // 1024}> ({256 * hh // 1024} heads per workgroup)"
                    out[f"chunk_kernel_h{hh}"] = ck
                ck = measure_chunk(dev, B=64, H=4, Dk=256, Dv=512, reps=60)   # expand_v = 2 (the reference's default, model/gla.py:52,267)
                ck["kernel"] = ("lina::gla_chunk_bf16_h256_kernel<false, 1, NCB = 2>: both 256-column blocks of v / o in ONE launch, "
                                "the two workgroups of a head on one XCD (block ids i, i + 8)")
                from lina_speech_amd import ops as _ops
                _ops.POLICY.dv512_one_launch = False                  # rounds 2-5: one launch per column block, same session
                try:
                    two = measure_chunk(dev, B=64, H=4, Dk=256, Dv=512, reps=30)
                finally:
                    _ops.POLICY.dv512_one_launch = True
                ck["two_launches"] = {"ms": two["ms"], "frac": two["frac"]}
                out["chunk_kernel_dv512"] = ck
                small = measure_chunk(dev, B=8)                      # training micro-batch: segment-parallel form
                small["kernel"] = "lina_gla_chunk_fwd_seg (state-only pass + combine + full pass, 8 segments)"
                out["chunk_kernel_b8"] = small
                out["vocoder"] = measure_vocoder(dev)
                out["chunk_bwd_kernel"] = measure_chunk_bwd(dev)
                out["chunk_bwd_kernel_b64"] = measure_chunk_bwd(dev, B=64, reps=20)   # one workgroup per head, no segments
    if rank == 0:
        if world == 1 and dtype == torch.bfloat16 and not args.no_chunk:
            # other head shapes of the same width (the 169M hyper-parameters are inferred)
            eng.close()
            del eng, eng1
            torch.cuda.empty_cache()
            shapes = {}
            for heads, ev in ((8, 1.0), (16, 1.0), (4, 2.0)):
                with torch.inference_mode():
                    torch.manual_seed(0)
                    mh = l169(heads=heads, expand_v=ev).eval().to(dev, dtype)
                    eh = DecodeEngine(mh, mh.txt_encoder(mh.txt_embed(texts)), batch_size=B)
                    eh.begin_greedy(320)
                    eh.greedy_steps(64)
                    torch.cuda.synchronize()
                    th = time.perf_counter()
                    eh.greedy_steps(240)
                    torch.cuda.synchronize()
                    th = (time.perf_counter() - th) / 240
                    shapes[f"H={heads},Dk={1024 // heads},Dv={int(1024 * ev) // heads}"] = {
                        "ms_per_step": th * 1e3, "tokens_per_s": B / th, "params_M": n_params(mh) / 1e6,
                        "update_kernel": "K1w (windowed)" if eh.packs[0].lazy else "generic K1 (head shape outside K1w)"}
                    del eh, mh
                    torch.cuda.empty_cache()
            out["other_head_shapes"] = shapes
            eng = None
            # the parity dtype beside the headline dtype: the same engine with fp32 weights / activations
            with torch.inference_mode():
                m32 = model.to(dev, torch.float32)
                eng32 = DecodeEngine(m32, m32.txt_encoder(m32.txt_embed(texts)), batch_size=B)
                eng32.begin_greedy(260)
                for _ in range(60):
                    eng32.greedy_step()
                torch.cuda.synchronize()
                t32 = time.perf_counter()
                for _ in range(200):
                    eng32.greedy_step()
                torch.cuda.synchronize()
                t32 = (time.perf_counter() - t32) / 200
            out["decode_f32"] = {"dtype": "f32", "ms_per_step": t32 * 1e3, "tokens_per_s": B / t32,
                                 "what": "same engine, fp32 weights and activations (the dtype the token-exact parity "
                                         "tests run in)"}
            del eng32
            model.to("cpu")
            torch.cuda.empty_cache()
        if not args.no_train and world == 1:
            eng = None
            torch.cuda.empty_cache()
            out["train_step"] = measure_train_step(dev)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(model)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
