"""N > 1 path on CPU: two processes (gloo, 127.0.0.1) each decode their shard of the utterance batch
with the real host code + kernel sources on the emulator; the gathered tokens must equal a
single-process run of the whole batch (the decode shards with no data-path collective)."""
import os
import socket
import sys

import pytest
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


def _decode(rows_lo, rows_hi, n_steps):
    from model_cases import build_lina, golden_state_dict, load_golden
    from lina_speech_amd.decode import DecodeEngine
    g = load_golden("lina_d64.npz")
    model = build_lina()
    model.load_state_dict(golden_state_dict(g))
    model.eval()
    gen = torch.Generator().manual_seed(7)
    texts = torch.randint(3, 256, (5, 9), generator=gen)[rows_lo:rows_hi]
    with torch.no_grad():
        x_enc = model.txt_encoder(model.txt_embed(texts))
        return DecodeEngine(model, x_enc, batch_size=rows_hi - rows_lo).run_greedy(n_steps)


def _worker(rank, world, port, out_path):
    import torch.distributed as dist
    _setup_emu()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lina_speech_amd.shard import gather_tokens, shard_rows
    lo, hi = shard_rows(5, rank, world)
    toks = _decode(lo, hi, 6)
    dist.barrier()
    full = gather_tokens(toks, 5)
    if rank == 0:
        torch.save(full, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_rows_partition():
    from lina_speech_amd.shard import shard_rows
    for total in (0, 1, 5, 64, 512, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_rows(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert [shard_rows(512, r, 8) for r in (0, 7)] == [(0, 64), (448, 512)]
    with pytest.raises(ValueError):
        shard_rows(4, 2, 2)


def test_two_rank_sharded_decode_equals_single_process(tmp_path, emu):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "gathered.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    gathered = torch.load(out)
    single = _decode(0, 5, 6)
    assert gathered.shape == (1, 5, 6)
    assert torch.equal(gathered, single)
