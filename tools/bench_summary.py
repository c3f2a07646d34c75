#!/usr/bin/env python
"""Print the headline figures of a bench.py JSON line (one per line, for a session log)."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
g = lambda *ks: (lambda v: v)(__import__("functools").reduce(lambda a, k: a.get(k, {}) if isinstance(a, dict) else {}, ks, d))
print(f"decode: {d['value']:.0f} tok/s  {d['ms_per_step']:.4f} ms/step  step_roofline {g('step_roofline','frac'):.3f}  K1w frac {g('roofline','frac'):.3f} ({g('roofline','us_per_launch'):.2f} us)")
for k in ("chunk_kernel", "chunk_kernel_h8", "chunk_kernel_h16", "chunk_kernel_dv512", "chunk_kernel_b8", "chunk_bwd_kernel", "chunk_bwd_kernel_b64"):
    if k in d:
        print(f"{k}: {d[k]['ms']:.4f} ms  frac {d[k]['frac']:.3f}")
for k, b in d.get("per_gpu_batch", {}).items():
    if "error" in b:
        print(f"per_gpu_batch {k}: {b['error']}")
    else:
        print(f"per_gpu_batch {k}: {b['tokens_per_s']:.0f} tok/s  {b['ms_per_step']:.4f} ms/step  step_roofline {b['step_roofline']['frac']:.3f}")
for k, b in d.get("generate_batch", {}).items():
    if "error" in b:
        print(f"generate_batch {k}: {b['error']}")
    else:
        print(f"generate_batch {k}: greedy {b['greedy']['tokens_per_s']:.0f} tok/s ({b['greedy']['vs_loop']} x loop), sampled "
              f"{b['sampled_k100']['tokens_per_s']:.0f}, first call {b['first_call_s']:.2f} s, early stop returned "
              f"{b['early_stop']['steps_returned']} of {b['early_stop']['steps_executed']} executed steps")
for k in ("sampled_decode", "train_step", "decode_f32"):
    if k in d:
        print(f"{k}: {d[k].get('ms_per_step'):.4f} ms/step  {d[k].get('tokens_per_s'):.0f} tok/s")
if "other_head_shapes" in d:
    for k, v in d["other_head_shapes"].items():
        print(f"{k}: {v['ms_per_step']:.4f} ms/step {v['tokens_per_s']:.0f} tok/s")
if "cpu_baseline" in d:
    print(f"cpu_baseline: {d['cpu_baseline']['value']:.1f} tok/s on {d['cpu_baseline']['cores']} threads")
if "decode_bf16_state" in d:
    t = d["decode_bf16_state"]
    print("decode_bf16_state:", t.get("error") or "  ".join(
        f"window {w}: {t[f'window_{w}']['tokens_per_s']:.0f} tok/s ({t[f'window_{w}']['ms_per_step']:.4f} ms, step {t[f'window_{w}']['step_roofline']['frac']:.3f} of HBM)"
        for w in (1, 8)))
if "per_gpu_batch" in d:
    print("per_gpu_batch:", "  ".join(f"{k} {v.get('ms_per_step', float('nan')):.4f} ms ({v.get('tokens_per_s', 0):.0f} tok/s)" for k, v in d["per_gpu_batch"].items()))
cb = d.get("cpu_baseline")
if cb:
    print(f"cpu_baseline: {cb['value']:.1f} tok/s at {cb['cores']} threads (min {cb.get('min', 0):.1f}, max {cb.get('max', 0):.1f})")
if "two_engines" in d:
    t = d["two_engines"]
    print("two_engines:", t.get("error") or f"{t['loop_tokens_per_s']:.0f} tok/s ({t['loop_ms_per_step']:.4f} ms/step), generate_batch {t['generate_batch_tokens_per_s']:.0f} tok/s")
