"""bench.py --gpus N must run N ranks: launched WITHOUT a launcher it re-executes itself under
torch.distributed.run; launched BY one (the driver's command) it refuses a world that is not N.  --check-launch stops
after the rendezvous, so this runs on a CPU-only host over gloo."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, *argv], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_gpus_2_relaunches_itself_with_two_ranks():
    r = _run(["bench.py", "--gpus", "2", "--check-launch"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["world_size"] == 2 and j["ranks_seen"] == 2 and j["n_gpus"] == 2
    assert j["rows_total"] == 512 and j["scaling"] == "strong"        # B_total = 512 (the metric's batch) split over the ranks
    r = _run(["bench.py", "--gpus", "2", "--check-launch", "--batch-per-gpu", "64"])
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["rows_total"] == 128 and j["scaling"] == "weak"


def test_bench_under_the_drivers_launcher_sees_world_2():
    r = _run(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", "29731", "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--check-launch"])
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["world_size"] == 2 and j["ranks_seen"] == 2


def test_bench_refuses_a_world_that_is_not_gpus():
    r = _run(["bench.py", "--gpus", "4", "--check-launch"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 4" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_decode_bench_under_a_live_rccl_process_group_on_one_gpu():
    """The multi-GPU launch path as far as ONE GPU can take it: bench.py under torch.distributed.run
    with --nproc-per-node 1 and a FORCED process group (backend "nccl" = RCCL): RCCL initialises, the barrier / max-over-ranks
    collectives of the timed region run, and the decode loop's hipGraph is captured and replayed next to the live
    communicator.  The JSON line must carry the ranks block (per-rank times, RCCL's own log summary)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no ROCm device")
    r = _run(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
              "--master-port", "29741", "bench.py", "--gpus", "1", "--steps", "24", "--warmup", "4", "--preheat-s", "0.2",
              "--batch-per-gpu", "64", "--no-chunk", "--no-train", "--no-cpu-baseline"],
             {"LINA_BENCH_FORCE_PG": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 1 and j["steps"] == 24 and j["value"] > 0
    rk = j["ranks"]
    assert rk["backend"] == "nccl" and rk["world_size"] == 1 and len(rk["per_rank_ms"]["all"]) == 1
    assert rk["rccl"] is not None and rk["rccl"]["library_version"], rk      # (one rank: RCCL may build no communicator -> no log)
