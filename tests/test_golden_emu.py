"""Product modules (real host code + kernel sources on the emulator) vs vectors captured from the
reference's own model/*.py (tests/golden/make_golden.py)."""
import numpy as np
import torch

from conftest import golden
from lina_speech_amd import codec
from model_cases import check_lina_golden, check_mixer_golden
from oracle import gla_oracle as O


def test_mixer_matches_reference_module(emu):
    check_mixer_golden("cpu")


def test_lina_forward_and_greedy_decode_match_reference(emu):
    check_lina_golden("cpu")


def test_tools_known_answers(emu):
    g = golden("tools.npz")
    for mod in (codec, O):   # product helpers and oracle helpers both reproduce reference tools.py
        d = mod.delay_rvq(torch.from_numpy(g["delay_in"]), head_token=1, tail_token=2)
        assert torch.equal(d, torch.from_numpy(g["delay_out"]))
        assert d.tolist() == [[1, 13, 14, 15, 16, 2]]
        d2 = mod.delay_rvq(torch.from_numpy(g["delay2_in"]), head_token=1, tail_token=2)
        assert torch.equal(d2, torch.from_numpy(g["delay2_out"]))
        assert torch.equal(mod.undelay_rvq(d2.unsqueeze(1)), torch.from_numpy(g["undelay2"]))
    lg = torch.from_numpy(g["topk_logits"])
    assert torch.equal(codec.topk_sampling(lg, k=1), torch.from_numpy(g["topk_k1"]))
    assert torch.equal(O.topk_sampling(lg, k=1), torch.from_numpy(g["topk_k1"]))
    assert torch.equal(codec.sequence_mask(torch.tensor([3, 1, 4])), torch.from_numpy(g["seqmask"]))


def test_fused_decode_engine_matches_reference_tokens(emu):
    # DecodeEngine (fused projection + prologue + in-place K1 + K5 + padded SwiGLU) == reference decode
    check_lina_golden("cpu", engine="fused")


def test_generate_batch_default_is_the_device_loop(emu):
    from model_cases import check_generate_batch_loop
    check_generate_batch_loop("cpu", full=False)


def test_generate_batch_outputs_are_fresh(emu):
    from model_cases import check_generate_batch_outputs_are_fresh
    check_generate_batch_outputs_are_fresh("cpu")


def test_engine_hidden_state_log(emu):
    from model_cases import check_engine_hidden_log
    check_engine_hidden_log("cpu")


def test_engine_bf16_state(emu):
    from model_cases import check_engine_bf16_state
    check_engine_bf16_state("cpu")


def test_generate_batch_with_two_engines(emu):
    from model_cases import check_generate_batch_group
    check_generate_batch_group("cpu")


def test_engine_device_side_greedy_loop(emu):
    import torch
    from model_cases import build_lina, golden_state_dict, load_golden
    from lina_speech_amd.decode import DecodeEngine
    g = load_golden("lina_d64.npz")
    model = build_lina()
    model.load_state_dict(golden_state_dict(g))
    model.eval()
    with torch.no_grad():
        x = torch.from_numpy(g["gen_x"]).unsqueeze(0).expand(3, -1)
        x_enc = model.txt_encoder(model.txt_embed(x))
        eng = DecodeEngine(model, x_enc, batch_size=3, n_split=2)     # two row ranges (parallel streams on a GPU)
        assert [(p.lo, p.hi) for p in eng.parts] == [(0, 2), (2, 3)]
        toks, atts = eng.run_greedy(12, record_att=True)
    assert torch.equal(toks, torch.from_numpy(g["gen_qs"]))
    assert atts.shape == g["gen_atts"].shape
    assert (atts - torch.from_numpy(g["gen_atts"])).abs().max() < 2e-4


def test_train_step_loss_and_gradients_match_reference(emu):
    from model_cases import check_lina_train_golden
    check_lina_train_golden("cpu")


def test_engine_device_side_sampling_loop(emu):
    from model_cases import check_engine_sampling
    check_engine_sampling("cpu")


def test_init_state_tuning_gradients_match_reference(emu):
    from model_cases import check_init_state_tuning_golden
    check_init_state_tuning_golden("cpu")


def test_vocoder_matches_reference_modules(emu):
    from model_cases import check_vocoder_golden
    check_vocoder_golden("cpu")


def test_config1_simple_gla_stack_matches_reference_wrapper(emu):
    """a-12: AttentiveSimpleGLA.forward (reference model/simple_gla.py:152-165) -- short ragged case on the emulator
    (the B=4, T=256 case of BASELINE configs[0] runs in the -m gpu suite and, on the oracle, in test_oracle.py)."""
    from model_cases import check_simple_gla_golden
    check_simple_gla_golden("cpu", full=False)


def test_engine_windowed_state_equals_immediate_state(emu):
    """K1w in the device-side loop (state rewritten every 8th / 4th token) vs the immediate update (window=1): same
    tokens, and after sync_state() the same recurrent states and conv caches; the loop can be continued after a sync
    (the window restarts) and re-armed (begin_greedy flushes what is pending)."""
    import torch
    from model_cases import build_lina, golden_state_dict, load_golden
    from lina_speech_amd.decode import DecodeEngine
    g = load_golden("lina_d64.npz")
    model = build_lina()
    model.load_state_dict(golden_state_dict(g))
    model.eval()
    with torch.no_grad():
        x = torch.from_numpy(g["gen_x"]).unsqueeze(0).expand(3, -1)
        x_enc = model.txt_encoder(model.txt_embed(x))
        ref = DecodeEngine(model, x_enc, batch_size=3, window=1)
        toks_ref = ref.run_greedy(12)
        assert torch.equal(toks_ref, torch.from_numpy(g["gen_qs"]))
        for window in (8,):           # other windows: kernel-level tests (test_decode_window)
            eng = DecodeEngine(model, x_enc, batch_size=3, window=window)
            assert eng.packs[0].lazy
            eng.begin_greedy(12)
            assert eng._loop_packed, "the device loop must run on fragment-major operands"
            for _ in range(5):
                eng.greedy_step()
            mid = [s[3].clone() for s in eng.state.states]           # sync in the middle of a window ...
            for _ in range(7):
                eng.greedy_step()                                    # ... and keep going
            assert torch.equal(eng.greedy_tokens(), toks_ref)
            for li, (a, b) in enumerate(zip(eng.state.states, ref.state.states)):
                for j, (ta, tb) in enumerate(zip(a, b)):
                    err = (ta - tb).abs().max() / tb.abs().max().clamp_min(1e-30)
                    assert err < 2e-5, (window, li, j, float(err))
            ref5 = DecodeEngine(model, x_enc, batch_size=3, window=1)
            ref5.run_greedy(5)
            for a, b in zip(mid, ref5.state.states):
                assert (a - b[3]).abs().max() / b[3].abs().max() < 2e-5
            eng.begin_greedy(12)
            eng.greedy_steps(12)                                     # groups of GRAPH_STEPS + remainder (no graph on CPU)
            assert torch.equal(eng.greedy_tokens(), toks_ref)
            # re-arming after a partial window: pending steps are flushed, the continuation matches a fresh run
            eng.begin_greedy(3, y0=None)
            eng.greedy_step()
            eng.begin_greedy(2)
            assert eng._n_done == 0 and int(eng._origin) == 0
