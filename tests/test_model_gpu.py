"""Module / model level parity on a real MI355X: reference goldens, fused engine + hipGraph,
and the 166.7M model's greedy tokens against the CPU oracle."""
import pytest
import torch
from kernel_cases import record_parity

from model_cases import check_lina_golden, check_mixer_golden

pytestmark = pytest.mark.gpu


def test_mixer_matches_reference_module(hip):
    check_mixer_golden("cuda")


def test_lina_forward_and_greedy_decode_match_reference(hip):
    check_lina_golden("cuda")


def test_fused_engine_with_hipgraph_matches_reference_tokens(hip):
    check_lina_golden("cuda", engine="fused")


def test_generate_batch_default_is_the_device_loop(hip):
    """generate_batch with no engine argument == the per-token module path with the reference's per-step stop test (tokens,
    stop flags, cuts exact; attention rows 2e-4), early stops at different steps, prompt, engine reuse / invalidation."""
    from model_cases import check_generate_batch_loop
    check_generate_batch_loop("cuda")


def test_generate_batch_outputs_are_fresh(hip):
    """returned tensors are the caller's; the engine cache sees param.data writes; bounded loop cache."""
    from model_cases import check_generate_batch_outputs_are_fresh
    check_generate_batch_outputs_are_fresh("cuda")


def test_generate_batch_with_two_engines_on_two_streams(hip):
    """decode.DecodeEngineGroup: two engines on two HIP streams behind generate_batch(n_engines=2) == the per-token module path
    (early stops at different steps in the two halves, forced length, reproducible sampling)."""
    from model_cases import check_generate_batch_group
    check_generate_batch_group("cuda")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_generate_batch_early_stop_over_many_checks(hip, dtype):
    """The hipGraph loop (8 tokens per replay, attention rows filed by the cross-attention kernels at the device step index,
    stop flags in the token epilogue, control block polled through pinned memory behind the queued work) stops where the
    reference's per-step test stops."""
    from model_cases import check_generate_batch_early_stop
    check_generate_batch_early_stop("cuda", dtype)


def test_l169_greedy_tokens_match_cpu_oracle(hip):
    """Full 166.7M model, fp32: device-side greedy loop (graph replay) vs oracle/lina_decode_oracle.py.
    Token ids must be identical wherever the oracle's top-2 logit margin exceeds 1e-3 (SURVEY A.8:
    equality is not meaningful at near-ties); teacher forcing keeps both sides on the oracle's tokens."""
    from lina_speech_amd.configs import l169
    from lina_speech_amd.decode import DecodeEngine
    from oracle.lina_decode_oracle import OracleLina
    torch.manual_seed(0)
    model = l169().eval()
    B, n = 4, 6
    x = torch.randint(3, 256, (B, 16))
    orc = OracleLina(model.state_dict(), n_layer=6, heads=4, txt_heads=4)
    ref_toks, ref_logits, ref_atts, margins = orc.generate_greedy(x, n)
    with torch.inference_mode():
        m = model.to("cuda")
        x_enc = m.txt_encoder(m.txt_embed(x.cuda()))
        eng = DecodeEngine(m, x_enc, batch_size=B)
        # teacher-forced logits through the generic step API
        y = m.rvq_embed.embed_sum(torch.ones(1, B, 1, dtype=torch.long, device="cuda"))
        worst = 0.0
        for t in range(n):
            logits, att = eng(y, t)
            ref = ref_logits[:, t:t + 1]
            worst = max(worst, float((logits.cpu() - ref).abs().max() / ref.abs().max()))
            pick = logits[:, 0, 0].argmax(-1).cpu()
            safe = margins[:, t] > 1e-3
            assert torch.equal(pick[safe], ref_toks[0, :, t][safe]), f"token mismatch at step {t}"
            y = m.rvq_embed.embed_sum(ref_toks[:, :, t:t + 1].cuda())
        record_parity("L169 fp32 engine (generic step API), teacher-forced logits vs fp64-checked oracle", worst, 5e-4)
        assert worst < 5e-4, f"logits rel err {worst:.2e}"
        # free-running device-side loop on a fresh state
        eng2 = DecodeEngine(m, x_enc, batch_size=B)
        toks = eng2.run_greedy(n).cpu()
    # rows stay comparable until their first near-tie
    for b in range(B):
        ok = (margins[b] > 1e-3).long().cumprod(0).bool()
        assert torch.equal(toks[0, b][ok], ref_toks[0, b][ok])


def test_bf16_engine_runs_and_tracks_fp32(hip):
    from lina_speech_amd.configs import tiny
    from lina_speech_amd.decode import DecodeEngine
    torch.manual_seed(1)
    model = tiny(d=256, heads=2, n_layer=2, n_codebook=4096).eval().cuda()
    x = torch.randint(3, 256, (8, 20), device="cuda")
    with torch.inference_mode():
        x_enc = model.txt_encoder(model.txt_embed(x))
        y = model.rvq_embed.embed_sum(torch.ones(1, 8, 1, dtype=torch.long, device="cuda"))
        l32, _ = DecodeEngine(model, x_enc, batch_size=8)(y, 0)
        l32 = l32.clone()
        mb = model.to(torch.bfloat16)
        lb, _ = DecodeEngine(mb, x_enc.bfloat16(), batch_size=8)(y.bfloat16(), 0)
    err = (lb.float() - l32).abs().max() / l32.abs().max()
    record_parity("tiny model: bf16 engine logits vs fp32 engine logits", err, 5e-2)
    assert torch.isfinite(lb.float()).all() and err < 5e-2, f"bf16 vs fp32 logits rel err {err:.3e}"


def test_train_step_loss_and_gradients_match_reference(hip):
    from model_cases import check_lina_train_golden
    check_lina_train_golden("cuda")


def test_l169_train_step_runs_in_bf16_autocast_and_learns(hip):
    """a-11 at the real width: d=1024, H=4 (Dk=Dv=256) -> K2 full-head forward + K2b backward in bf16."""
    from lina_speech_amd import configs
    from lina_speech_amd.train import TrainStep, synthetic_batch
    torch.manual_seed(0)
    model = configs.l169()
    ts = TrainStep(model, device=torch.device("cuda", 0), lr=1e-3, ddp=False, n_warmup_steps=0, grad_clip=1.0)
    batch = synthetic_batch(b=2, n=257, t_txt=32, seed=3).to("cuda")
    losses = [float(ts.step(batch)) for _ in range(5)]
    assert all(l == l and l < 1e4 for l in losses), losses
    assert losses[-1] < losses[0], losses


def test_engine_device_side_sampling_loop_in_hipgraph(hip):
    from model_cases import check_engine_sampling
    check_engine_sampling("cuda", n_steps=12)


def test_init_state_tuning_gradients_match_reference(hip):
    from model_cases import check_init_state_tuning_golden
    check_init_state_tuning_golden("cuda")


def test_vocoder_matches_reference_modules(hip):
    from model_cases import check_vocoder_golden
    check_vocoder_golden("cuda")


def test_full_size_vocoder_matches_cpu_oracle(hip):
    """WavTokenizer-sized decoder (dim 768, 12 ConvNeXt blocks, n_fft 1280, hop 320), fp32, vs the fp64 oracle."""
    from lina_speech_amd.vocoder import WavTokenizerDecoder
    from oracle.vocoder_oracle import OracleVocoder
    torch.manual_seed(3)
    voc = WavTokenizerDecoder().eval()
    with torch.no_grad():
        for name, p in voc.named_parameters():
            if "gamma" in name or "scale" in name or "shift" in name:
                p.add_(torch.randn_like(p) * 0.1)
        voc.head.out.weight.mul_(0.2)
    codes = torch.randint(0, 4096, (1, 2, 60))
    bw = torch.tensor([0])
    sd = {k: v.detach().clone() for k, v in voc.state_dict().items()}
    feats = voc.codebook[0][codes[0]].transpose(1, 2)
    ref = OracleVocoder(sd, 12, 1280, 320).decode(feats, bw)
    voc = voc.cuda()
    audio = voc(codes.cuda(), bandwidth_id=bw.cuda())
    assert audio.shape == (2, 60 * 320)
    err = (audio.cpu().double() - ref).abs().max() / ref.abs().max()
    record_parity("WavTokenizer decode: waveform vs fp64 oracle", err, 5e-4)
    assert err < 5e-4, float(err)


def test_config1_simple_gla_stack_matches_reference_wrapper(hip):
    """BASELINE configs[0] at its named shape (d=256, 2 GLA blocks, B=4, T=256): product on the GPU vs the output the
    reference's AttentiveSimpleGLA.forward produced on the CPU (pure-PyTorch recurrent)."""
    from model_cases import check_simple_gla_golden
    check_simple_gla_golden("cuda", full=True)


def test_l169_bf16_engine_b64_free_running_vs_fp32_oracle(hip):
    """The HEADLINE configuration (BASELINE configs[1]: L169, bf16, B=64, device-side hipGraph loop with the windowed
    state update) against the fp32 CPU oracle of the reference loop (model/modeling_lina.py:152-179), on weights whose
    logits are PEAKED (model_cases.peak_logits: random-init weights give flat logits and a third of the positions were
    near-ties, where the arg-max claim cannot be checked):
      1. the engine decodes 32 tokens free-running;
      2. the oracle (fp32 arithmetic on the SAME bf16-rounded weights) is teacher-forced with the engine's tokens ->
         reference logits and top-2 margins for the same history at every position;
      3. the engine's teacher-forced logits (generic step API, immediate state update) and the windowed loop's OWN logits
         must be within 1e-2 of max|oracle logits| (bf16 activations vs fp32; achieved 3.3e-3);
      4. every free-running token must be the oracle's arg-max wherever the oracle's margin exceeds TWICE the measured
         logit error (if |dlogit| <= e everywhere the arg-max cannot differ at a margin > 2e); fewer than 5 % of the
         B x n positions may fall below that margin, and there must be NO raw token difference outside them."""
    from lina_speech_amd.configs import l169
    from lina_speech_amd.decode import DecodeEngine
    from oracle.lina_decode_oracle import OracleLina
    torch.manual_seed(0)
    from model_cases import peak_logits
    model = peak_logits(l169().eval())
    B, n, REL, REL_ATT = 64, 32, 1e-2, 2e-2     # logits: 1e-2 of max|oracle logit| (achieved 3.3e-3; ADVICE r04: the peaked head
                                                # makes max|logit| several times larger, so the round-4 bound of 2e-2 was loose)
    MASK_CAP = 0.05          # peaked logits: (almost) every position is comparable (round 3, flat logits: 30 % were not)
    x = torch.randint(3, 256, (B, 24), generator=torch.Generator().manual_seed(7))
    mb = model.to(torch.bfloat16)
    sd = {k: v.float() for k, v in mb.state_dict().items()}          # the oracle sees the SAME (bf16-rounded) weights
    with torch.inference_mode():
        m = mb.to("cuda")
        x_enc = m.txt_encoder(m.txt_embed(x.cuda()))
        eng = DecodeEngine(m, x_enc, batch_size=B)
        assert eng.window == 8 and eng.packs[0].lazy
        # the bench's own loop (K1w windowed state, packed projections, K6d epilogue inside the hipGraph), one replay per
        # token; the head's output buffer is copied out after every step: these ARE the logits the loop picked from
        eng.begin_greedy(n, log_hidden=True)         # (+ the pre-head hidden state of every step: VERDICT r05 item 3 (c))
        assert eng._loop_packed and eng._greedy_graph is not None
        loop_logits = []
        for _ in range(n):
            eng.greedy_step()
            loop_logits.append(eng._logits.view(B, 1, eng.Q, eng.L).float().cpu())
        toks = eng.greedy_tokens().cpu()                                                    # [1,B,n]
        loop_hidden = eng.logged_hidden(n).float().cpu()                                    # [n,B,d]
        eng.sync_state()
        loop_logits = torch.cat(loop_logits, dim=1)                                         # [B,n,Q,L]
        # the reference's ENTRY POINT with no engine argument runs this very loop (8 tokens per replay, its own engine):
        # same tokens bit for bit, so everything established below about `toks` holds for what generate_batch returns
        gb_qs, gb_atts, gb_stops, gb_cuts = m.generate_batch(x.cuda(), batch_size=B, max_seqlen=n, k=1, first_greedy_quant=0,
                                                             force_max_seqlen=True, device="cuda")
        assert torch.equal(gb_qs.cpu(), toks), "generate_batch does not run the loop whose logits are checked here"
        assert next(reversed(m._decode_engines.values()))._loop.att_direct, "att rows must be filed by the cross-attention launches"
        gb_atts = gb_atts.float().cpu()
    orc = OracleLina(sd, n_layer=6, heads=4, txt_heads=4)
    n_thr = torch.get_num_threads()
    torch.set_num_threads(min(n_thr, 32))        # small-op decode on a 256-thread host: more threads only add sync cost
    try:
        ref_toks, ref_logits, ref_atts, margins = orc.generate_greedy(x, n, teacher=toks)  # teacher-forced on OUR tokens
    finally:
        torch.set_num_threads(n_thr)
    ref_hidden = torch.cat(orc.hiddens, dim=1).transpose(0, 1)
    hid_err = (loop_hidden - ref_hidden).abs().amax(dim=(1, 2)) / ref_hidden.abs().max()
    record_parity("L169 bf16 B=64 windowed device loop: pre-head hidden state vs fp32 oracle (teacher-forced on the loop's tokens)",
                  float(hid_err.max()), 2e-2, steps=n, first_step=float(hid_err[0]), last_step=float(hid_err[-1]))
    assert float(hid_err.max()) < 2e-2, f"bf16 pre-head hidden state rel err {float(hid_err.max()):.3e}"
    att_err = float((gb_atts - ref_atts).abs().max() / ref_atts.abs().max())
    record_parity("L169 bf16 B=64 generate_batch: attention log [B,2,n,Ttxt] vs fp32 oracle", att_err, REL_ATT)
    assert gb_atts.shape == ref_atts.shape and att_err < REL_ATT, f"attention log rel err {att_err:.3e}"
    assert gb_stops.shape == (B, n + 1) and len(gb_cuts) == B
    scale = float(ref_logits.abs().max())
    with torch.inference_mode():
        eng2 = DecodeEngine(m, x_enc, batch_size=B)
        y = m.rvq_embed.embed_sum(torch.ones(1, B, 1, dtype=torch.long, device="cuda"))
        worst = 0.0
        for t in range(n):
            logits, _ = eng2(y, t)
            worst = max(worst, float((logits.float().cpu() - ref_logits[:, t:t + 1]).abs().max()))
            y = m.rvq_embed.embed_sum(toks[:, :, t:t + 1].cuda())
    # the windowed device loop's OWN logits (free-running on its own tokens == the oracle's teacher-forced inputs)
    worst_loop = float((loop_logits - ref_logits).abs().max())
    per_step = (loop_logits - ref_logits).abs().amax(dim=(0, 2, 3))
    worst = max(worst, worst_loop)
    safe = margins > 2.0 * worst
    n_masked, n_diff = int((~safe).sum()), int((toks[0] != ref_toks[0]).sum())
    print(f"\nbf16 B=64 engine vs fp32 oracle over {n} free-running steps: max |logit error| of the windowed loop = "
          f"{worst_loop:.4f} = {worst_loop / scale:.2e} of max|logit| (generic step API {worst / scale:.2e}); {n_masked} of "
          f"{B * n} positions have a top-2 margin <= {2 * worst:.4f} (not comparable); {n_diff} raw token differences "
          f"(all of them must be at such positions); oracle top-2 margin / max|logit|: min {float(margins.min()) / scale:.3f}, "
          f"median {float(margins.median()) / scale:.3f}; distinct tokens decoded: {int(toks.unique().numel())}")
    record_parity("L169 bf16 B=64 windowed device loop: its own logits vs fp32 oracle (teacher-forced on the loop's tokens)",
                  worst_loop / scale, REL, steps=n, first_step=float(per_step[0] / scale), last_step=float(per_step[-1] / scale))
    record_parity("L169 bf16 B=64: positions with a top-2 margin <= 2 x max logit error (excluded from the token comparison)",
                  n_masked / (B * n), MASK_CAP, n_masked=n_masked, positions=B * n, raw_token_differences=n_diff)
    assert worst < REL * scale, f"logits rel err {worst / scale:.3e}"
    assert n_masked < MASK_CAP * B * n
    assert torch.equal(toks[0][safe], ref_toks[0][safe]), "bf16 engine token != oracle arg-max at a clear margin"


def test_generate_batch_picks_two_engines_at_512_rows_same_tokens(hip):
    """The entry point's default at the metric's batch: from ``LinaModel.AUTO_TWO_ENGINES_ROWS`` rows up ``generate_batch`` runs
    decode.DecodeEngineGroup (two engines of 256 rows on two HIP streams).  Rows never interact (reference
    model/modeling_lina.py:125,152-179) and no kernel's per-row sums depend on the row count, so the greedy tokens, stop flags
    and attention rows must be BIT-IDENTICAL to the one-engine loop of the same 512 rows (L169, bf16, peaked logits)."""
    from lina_speech_amd.configs import l169
    from lina_speech_amd.decode import DecodeEngine, DecodeEngineGroup
    from model_cases import peak_logits
    torch.manual_seed(0)
    m = peak_logits(l169().eval()).to("cuda", torch.bfloat16)
    B, n = 512, 24
    assert B >= m.AUTO_TWO_ENGINES_ROWS
    x = torch.randint(3, 256, (B, 24), generator=torch.Generator().manual_seed(21)).cuda()
    kw = dict(batch_size=B, max_seqlen=n, k=1, first_greedy_quant=0, force_max_seqlen=True, device="cuda")
    two = m.generate_batch(x, **kw)
    grp = next(reversed(m._decode_engines.values()))
    assert isinstance(grp, DecodeEngineGroup) and [hi - lo for lo, hi in grp.ranges] == [B // 2, B // 2]
    one = m.generate_batch(x, n_engines=1, **kw)
    assert isinstance(next(reversed(m._decode_engines.values())), DecodeEngine)
    assert torch.equal(two[0], one[0]) and torch.equal(two[2], one[2]), "two engines decode other tokens than one"
    assert torch.equal(two[1], one[1]), "attention rows differ between one and two engines"
    assert int(one[0].unique().numel()) > 100                       # (rows decode different sequences)
    m.generate_batch(x[:64], **{**kw, "batch_size": 64})
    assert isinstance(next(reversed(m._decode_engines.values())), DecodeEngine)      # below the threshold: one engine
    m.clear_decode_cache()


@pytest.mark.parametrize("window", [1, 8])
def test_l169_bf16_state_engine_vs_oracle_with_bf16_rounded_state(hip, window):
    """VERDICT r05 item 9, the opt-in reference-dtype state: ``DecodeEngine(state_dtype=torch.bfloat16)`` against an oracle that
    keeps the recurrent state in bf16 between steps -- rounds it after EVERY step, as the reference does for a bf16 model
    (model/gla.py:229-240 `param.new_zeros` + Cache.update's copy_).  window = 1 is that arithmetic (the kernel rounds at every
    write-back = every step); window = 8 stores the state in bf16 but rounds it every 8th step only.  L169, bf16, B = 8, peaked
    logits, 24 free-running steps of the device loop; the oracle is teacher-forced on the loop's tokens: pre-head hidden state
    and logits within the bf16 bounds of the fp32-state test, tokens equal at margins beyond twice the logit error."""
    from lina_speech_amd.configs import l169
    from lina_speech_amd.decode import DecodeEngine
    from oracle.lina_decode_oracle import OracleLina
    from model_cases import peak_logits
    torch.manual_seed(0)
    mb = peak_logits(l169().eval()).to(torch.bfloat16)
    B, n = 8, 24
    x = torch.randint(3, 256, (B, 24), generator=torch.Generator().manual_seed(17))
    sd = {k: v.float() for k, v in mb.state_dict().items()}
    with torch.inference_mode():
        m = mb.to("cuda")
        eng = DecodeEngine(m, m.txt_encoder(m.txt_embed(x.cuda())), batch_size=B, state_dtype=torch.bfloat16, window=window)
        assert eng.packs[0].S.dtype == torch.bfloat16 and eng.packs[0].lazy and eng.window == window
        eng.begin_greedy(n, log_hidden=True)
        assert eng._loop_packed
        eng.greedy_steps(n)
        toks = eng.greedy_tokens().cpu()
        hid = eng.logged_hidden(n).float().cpu()
        S_eng = eng.state.states[0][3].float().cpu()          # (sync_state: pending window steps applied, rounded)
        gb = m.generate_batch(x.cuda(), batch_size=B, max_seqlen=n, k=1, first_greedy_quant=0, force_max_seqlen=True,
                              device="cuda", state_dtype=torch.bfloat16) if window == 1 else None
        eng.close()
    if gb is not None:                                        # the entry point's opt-in runs this very loop (window 1)
        assert torch.equal(gb[0].cpu(), toks)
    orc = OracleLina(sd, n_layer=6, heads=4, txt_heads=4, state_dtype=torch.bfloat16)
    n_thr = _oracle_threads()
    try:
        ref_toks, ref_logits, ref_atts, margins = orc.generate_greedy(x, n, teacher=toks)
    finally:
        torch.set_num_threads(n_thr)
    ref_hid = torch.cat(orc.hiddens, dim=1).transpose(0, 1)
    hid_err = float((hid - ref_hid).abs().max() / ref_hid.abs().max())
    W = sd["logits_head.weight"][0].double()
    lg = (hid.double() @ W.t()).transpose(0, 1)
    lg_err = float((lg - ref_logits[:, :, 0].double()).abs().max())
    scale = float(ref_logits.abs().max())
    st_err = float((S_eng - orc.final_state[0][3]).abs().max() / orc.final_state[0][3].abs().max())
    safe = margins > 2.0 * lg_err
    tag = f"L169 bf16 activations + bf16 recurrent state (window {window}) B=8 x 24 steps vs the oracle with a bf16-rounded state"
    record_parity(tag + ": pre-head hidden state", hid_err, 2e-2)
    record_parity(tag + ": logits implied by the hidden states", lg_err / scale, 1e-2)
    record_parity(tag + ": first block's state after the run", st_err, 3e-2, masked_positions=int((~safe).sum()), positions=B * n)
    assert hid_err < 2e-2 and lg_err < 1e-2 * scale and st_err < 3e-2
    assert int((~safe).sum()) < 0.1 * B * n
    assert torch.equal(toks[0][safe], ref_toks[0][safe])


def _oracle_threads():
    n = torch.get_num_threads()
    torch.set_num_threads(min(n, 32))            # small-op decode on a 256-thread host: more threads only add sync cost
    return n


def test_l169_fp32_b64_generate_batch_64_steps_tokens_and_hidden_vs_oracle(hip):
    """VERDICT r05 item 3 (a) + (c): the reference's entry point, L169 in fp32, B = 64 (BASELINE configs[1]) x 64 steps = 8 K1w
    windows, peaked logits, against the fp32 CPU oracle teacher-forced on the loop's own tokens:
      * token ids EXACTLY the oracle's arg-max wherever the oracle's top-2 margin exceeds 1e-3 -- the masked
        fraction is recorded;
      * the PRE-HEAD hidden state of every step (what the codec head reads: sensitive to recurrent-state error, where the logits
        of a peaked head are dominated by the embedding -> head shortcut) within 2e-5 of max|hidden|, and the logits the hidden
        states imply within 2e-5 of max|logit|."""
    from lina_speech_amd.configs import l169
    from lina_speech_amd.decode import DecodeEngine
    from oracle.lina_decode_oracle import OracleLina
    from model_cases import peak_logits
    torch.manual_seed(0)
    model = peak_logits(l169().eval())
    B, n, REL = 64, 64, 2e-5              # (achieved 1.2e-6, profiles/r06_parity.json)
    x = torch.randint(3, 256, (B, 24), generator=torch.Generator().manual_seed(7))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    with torch.inference_mode():
        m = model.to("cuda")
        qs, atts, stops, cuts = m.generate_batch(x.cuda(), batch_size=B, max_seqlen=n, k=1, first_greedy_quant=0,
                                                 force_max_seqlen=True, device="cuda")
        ge = next(reversed(m._decode_engines.values()))
        assert isinstance(ge, DecodeEngine) and ge.window == 8 and ge._loop.graphN is not None      # the hipGraph loop, 8 tokens per replay
        toks = qs.cpu()
        # the same loop once more with the hidden-state log armed (its own engine): the same tokens bit for bit
        eng = DecodeEngine(m, m.txt_encoder(m.txt_embed(x.cuda())), batch_size=B)
        eng.begin_greedy(n, log_hidden=True)
        eng.greedy_steps(n)
        assert torch.equal(eng.greedy_tokens().cpu(), toks), "the hidden-state log changed the tokens"
        hid = eng.logged_hidden(n).float().cpu()                                   # [n,B,d]
        eng.close()
    orc = OracleLina(sd, n_layer=6, heads=4, txt_heads=4)
    n_thr = _oracle_threads()
    try:
        ref_toks, ref_logits, ref_atts, margins = orc.generate_greedy(x, n, teacher=toks)
    finally:
        torch.set_num_threads(n_thr)
    ref_hid = torch.cat(orc.hiddens, dim=1).transpose(0, 1)                        # [n,B,d]
    hid_err = (hid - ref_hid).abs().amax(dim=(1, 2)) / ref_hid.abs().max()
    W = sd["logits_head.weight"][0].double()
    lg_err = float(((hid.double() @ W.t()).transpose(0, 1) - ref_logits[:, :, 0].double()).abs().max() / ref_logits.abs().max())
    safe = margins > 1e-3
    n_masked, n_diff = int((~safe).sum()), int((toks[0] != ref_toks[0]).sum())
    print(f"\nfp32 L169 B=64 x {n} steps through generate_batch: pre-head hidden state {float(hid_err.max()):.2e} of max|hidden| "
          f"(step 0 {float(hid_err[0]):.2e}, step {n - 1} {float(hid_err[-1]):.2e}), implied logits {lg_err:.2e}; {n_masked} of {B * n} "
          f"positions at a top-2 margin <= 1e-3, {n_diff} raw token differences; distinct tokens {int(toks.unique().numel())}")
    record_parity("L169 fp32 B=64 x 64 steps, generate_batch device loop: pre-head hidden state vs fp32 oracle (teacher-forced on the loop's tokens)",
                  float(hid_err.max()), REL, steps=n, first_step=float(hid_err[0]), last_step=float(hid_err[-1]))
    record_parity("L169 fp32 B=64 x 64 steps, generate_batch device loop: logits implied by the hidden states vs fp32 oracle", lg_err, REL)
    record_parity("L169 fp32 B=64 x 64 steps: positions with an oracle top-2 margin <= 1e-3 (excluded from the exact token comparison)",
                  n_masked / (B * n), 0.05, n_masked=n_masked, positions=B * n, raw_token_differences=n_diff)
    assert float(hid_err.max()) < REL and lg_err < REL
    assert n_masked < 0.05 * B * n
    assert torch.equal(toks[0][safe], ref_toks[0][safe]), "fp32 token != oracle arg-max at a margin > 1e-3"
    att_err = float((atts.float().cpu() - ref_atts).abs().max() / ref_atts.abs().max())
    record_parity("L169 fp32 B=64 x 64 steps, generate_batch: attention log vs fp32 oracle", att_err, 2e-5)
    assert att_err < 2e-5 and stops.shape == (B, n + 1) and len(cuts) == B


def test_l169_fp32_long_horizon_512_steps_vs_oracle(hip):
    """VERDICT r05 item 3 (b) + (c): 512 free-running steps (64 K1w windows; the bench runs 750) at B = 8, fp32, peaked logits,
    device loop with 8 tokens per hipGraph replay.  The fp32 oracle is teacher-forced on the loop's tokens over all 512 steps;
    the pre-head hidden state is compared at EVERY step (the error must not grow with the horizon: recorded at steps 0, 63, ...,
    511), the tokens wherever the oracle's margin exceeds 1e-3."""
    from lina_speech_amd.configs import l169
    from lina_speech_amd.decode import DecodeEngine
    from oracle.lina_decode_oracle import OracleLina
    from model_cases import peak_logits
    torch.manual_seed(0)
    model = peak_logits(l169().eval())
    B, n, REL = 8, 512, 2e-5              # (achieved 1.2e-6)
    x = torch.randint(3, 256, (B, 24), generator=torch.Generator().manual_seed(9))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    with torch.inference_mode():
        m = model.to("cuda")
        eng = DecodeEngine(m, m.txt_encoder(m.txt_embed(x.cuda())), batch_size=B)
        eng.begin_greedy(n, log_hidden=True, log_att=True)
        eng.greedy_steps(n)
        assert eng._loop.graphN is not None and eng.window == 8
        toks = eng.greedy_tokens().cpu()
        hid = eng.logged_hidden(n).float().cpu()
        final_logits = eng._logits.view(B, eng.Q, eng.L).float().cpu()
        eng.close()
    orc = OracleLina(sd, n_layer=6, heads=4, txt_heads=4)
    n_thr = _oracle_threads()
    try:
        ref_toks, ref_logits, ref_atts, margins = orc.generate_greedy(x, n, teacher=toks)
    finally:
        torch.set_num_threads(n_thr)
    ref_hid = torch.cat(orc.hiddens, dim=1).transpose(0, 1)
    hid_err = (hid - ref_hid).abs().amax(dim=(1, 2)) / ref_hid.abs().max()
    marks = {f"step_{t}": float(hid_err[t]) for t in [0] + list(range(63, n, 64))}
    fin_err = float((final_logits - ref_logits[:, -1]).abs().max() / ref_logits.abs().max())
    safe = margins > 1e-3
    n_masked, n_diff = int((~safe).sum()), int((toks[0] != ref_toks[0]).sum())
    print(f"\nfp32 L169 B=8 x {n} free-running steps: pre-head hidden state max {float(hid_err.max()):.2e} of max|hidden|, at every "
          f"64th step {[f'{v:.1e}' for v in marks.values()]}; last step's logits {fin_err:.2e}; {n_masked} of {B * n} positions masked, "
          f"{n_diff} raw token differences; distinct tokens {int(toks.unique().numel())}")
    record_parity("L169 fp32 B=8 x 512 free-running steps (64 K1w windows): pre-head hidden state vs fp32 oracle, worst step",
                  float(hid_err.max()), REL, steps=n, **marks)
    record_parity("L169 fp32 B=8 x 512 steps: logits of the last step vs fp32 oracle", fin_err, REL)
    record_parity("L169 fp32 B=8 x 512 steps: positions with an oracle top-2 margin <= 1e-3", n_masked / (B * n), 0.05,
                  n_masked=n_masked, positions=B * n, raw_token_differences=n_diff)
    assert float(hid_err.max()) < REL and fin_err < REL
    assert float(hid_err[-64:].max()) < 4 * max(float(hid_err[:64].max()), 2e-5), "the hidden-state error grows with the horizon"
    assert n_masked < 0.05 * B * n
    assert torch.equal(toks[0][safe], ref_toks[0][safe])


def test_config5_slice_at_sequence_length_4096_matches_reference_autograd(hip):
    """BASELINE configs[4] at its named length, fp32: loss and EVERY parameter gradient of a b=1, T=4096 teacher-forced train
    step on a one-layer slice of L169 (3 GLA blocks at d=1024, H=4 + text encoder + 4099-way head) against the golden made
    from the REFERENCE's modules under torch autograd (tests/golden/make_golden.py config5_slice)."""
    from model_cases import check_config5_slice_golden
    check_config5_slice_golden("cuda", torch.float32)


def test_config5_slice_at_sequence_length_4096_bf16_autocast_tracks_reference_autograd(hip):
    """The same step the way training runs it (bf16 autocast: K2 / K2b full-head kernels, bf16 MFMA): as close to the fp32
    reference gradients as the REFERENCE's own modules are under torch.autocast(bfloat16) (measured when the golden was made)."""
    from model_cases import check_config5_slice_golden
    check_config5_slice_golden("cuda", torch.bfloat16, rel_loss=5e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_config5_slice_bigram_targets_cosine_with_reference_gradients(hip, dtype):
    """A bf16 train check that bites: the T = 4096 slice on coherent (bigram) targets -- fp32 entries within
    5e-3, bf16-autocast per-tensor cosine >= 0.97 and norm error <= 10 % against the reference's fp32 autograd."""
    from model_cases import check_config5_structured_golden
    check_config5_structured_golden("cuda", dtype)


def test_config4_rows_sharded_equal_unsharded(hip):
    """BASELINE configs[3] (169M decode, B = 512 batch-sharded over 8 GPUs, no collective) on ONE GPU: the eight
    `shard_rows(512, r, 8)` engines, run one after another on cuda:0 as rank r of the 8-GPU job would run them (its rows of
    the text batch, its own state), against the same rows of a single B = 512 engine -- rows never interact (reference
    model/modeling_lina.py:125,152-179: one state and one token stream per row).  bf16, the headline dtype; peaked logits
    (model_cases.peak_logits).  The B = 512 engine decodes free-running through the graph loop (2048 K1w workgroups, 8 row tiles
    per projection: config 4's row count through the HIP path); every shard engine is then teacher-forced with those tokens
    through the generic step API and must reproduce the big engine's LOGITS at every step within bf16 rounding (the projection
    kernels pick their tiling and split-K width by the row count, so the fp32 sums are taken in another order -- same values
    up to rounding, not the same bits), and its arg-max wherever the top-2 margin exceeds twice that tolerance."""
    from lina_speech_amd.configs import l169
    from lina_speech_amd.decode import DecodeEngine
    from lina_speech_amd.shard import shard_rows
    from model_cases import peak_logits
    torch.manual_seed(0)
    TOTAL, WORLD, n, TOL = 512, 8, 24, 1e-2
    model = peak_logits(l169().eval()).to("cuda", torch.bfloat16)
    texts = torch.randint(3, 256, (TOTAL, 32), generator=torch.Generator().manual_seed(1234)).cuda()
    with torch.inference_mode():
        x_enc = model.txt_encoder(model.txt_embed(texts))
        full_eng = DecodeEngine(model, x_enc, batch_size=TOTAL)
        full_eng.begin_greedy(n)
        full_logits = []
        for _ in range(n):
            full_eng.greedy_step()
            full_logits.append(full_eng._logits.view(TOTAL, full_eng.Q, full_eng.L).clone())
        full = full_eng.greedy_tokens().clone()                                                 # [1, 512, n]
        assert full.shape == (1, TOTAL, n)
        scale = max(float(l.float().abs().max()) for l in full_logits)
        worst, n_mask, n_diff = 0.0, 0, 0
        for r in range(WORLD):
            lo, hi = shard_rows(TOTAL, r, WORLD)
            assert hi - lo == 64
            eng = DecodeEngine(model, x_enc[lo:hi], batch_size=hi - lo)
            y = model.rvq_embed.embed_sum(torch.ones(1, hi - lo, 1, dtype=torch.long, device="cuda"))
            for t in range(n):
                logits, _ = eng(y, t)                                                           # [64, 1, Q, L]
                got, ref = logits[:, 0].float(), full_logits[t][lo:hi].float()
                worst = max(worst, float((got - ref).abs().max()))
                top2 = ref.topk(2, dim=-1).values
                safe = (top2[..., 0] - top2[..., 1]) > 2 * TOL * scale
                n_mask += int((~safe).sum())
                n_diff += int((got.argmax(-1)[safe] != ref.argmax(-1)[safe]).sum())
                y = model.rvq_embed.embed_sum(full[:, lo:hi, t:t + 1])
            del eng
    record_parity("config 4: logits of 8 x 64-row shard engines (teacher-forced) vs the same rows of one B=512 engine, "
                  "max |difference| / max|logit|", worst / scale, TOL, rows=TOTAL, steps=n,
                  distinct_tokens=int(full.unique().numel()), distinct_rows=len({tuple(r_.tolist()) for r_ in full[0].cpu()}))
    record_parity("config 4: positions with a top-2 margin <= 2 x tolerance (excluded from the arg-max comparison)",
                  n_mask / (TOTAL * n), 0.08, arg_max_differences_elsewhere=n_diff)
    assert worst <= TOL * scale, (worst, scale)
    assert n_diff == 0, f"{n_diff} arg-max differences at clear margins"
    assert n_mask < 0.08 * TOTAL * n
    assert len({tuple(r_.tolist()) for r_ in full[0].cpu()}) > 50       # the rows decode different sequences
    # ---- and against the ORACLE (round 5): rows 192..255 of the B = 512 engine -- the tall projection kernels (128-row
    # workgroups), 2048 K1w workgroups -- vs the fp32 CPU restatement of the reference loop teacher-forced with the engine's
    # own tokens: logits within 2e-2 of max|logit| (bf16 activations vs fp32), arg-max equal at clear margins
    from oracle.lina_decode_oracle import OracleLina
    lo, hi = shard_rows(TOTAL, 3, WORLD)
    sd = {k: v.float().cpu() for k, v in model.state_dict().items()}          # the SAME (bf16-rounded) weights
    orc = OracleLina(sd, n_layer=6, heads=4, txt_heads=4)
    toks = full[:, lo:hi].cpu()
    n_thr = torch.get_num_threads()
    torch.set_num_threads(min(n_thr, 32))
    try:
        ref_toks, ref_logits, _, margins = orc.generate_greedy(texts[lo:hi].cpu(), n, teacher=toks)
    finally:
        torch.set_num_threads(n_thr)
    got = torch.stack([l[lo:hi].float().cpu() for l in full_logits], dim=1)               # [64, n, Q, L]
    o_scale = float(ref_logits.abs().max())
    o_err = float((got - ref_logits).abs().max())
    safe = margins > 2.0 * o_err
    record_parity("config 4: rows 192..255 of the B=512 engine (free-running) vs the fp32 oracle teacher-forced on its tokens, "
                  "max |logit difference| / max|logit|", o_err / o_scale, 2e-2, steps=n,
                  positions_below_margin=int((~safe).sum()), raw_token_differences=int((toks[0] != ref_toks[0]).sum()))
    assert o_err < 2e-2 * o_scale, (o_err, o_scale)
    assert int((~safe).sum()) < 0.08 * 64 * n
    assert torch.equal(toks[0][safe], ref_toks[0][safe]), "B=512 engine token != oracle arg-max at a clear margin"


def test_config3_decode_to_waveform_chain_vs_oracle(hip):
    """BASELINE configs[2] as ONE pipeline (reference model/modeling_lina.py:181-192 ->
    3rdparty/decoder/pretrained.py:193-239): L169 greedy decode (fused engine, B=64) -> undelay_rvq - 3 (clamped)
    -> codes -> WavTokenizer decoder -> 24 kHz waveform, against the oracle chain (OracleLina tokens ->
    oracle undelay -> OracleVocoder in fp64).  fp32 end to end; waveform rows are compared for the rows whose tokens
    are all decided at a clear margin (a flipped near-tie token legitimately changes that row's audio)."""
    from lina_speech_amd.codec import undelay_rvq
    from lina_speech_amd.configs import l169
    from lina_speech_amd.vocoder import WavTokenizerDecoder
    from oracle import gla_oracle as O
    from oracle.lina_decode_oracle import OracleLina
    from oracle.vocoder_oracle import OracleVocoder
    torch.manual_seed(0)
    model = l169().eval()
    B, n = 64, 14
    x = torch.randint(3, 256, (B, 16), generator=torch.Generator().manual_seed(11))
    orc = OracleLina(model.state_dict(), n_layer=6, heads=4, txt_heads=4)
    ref_toks, _, _, margins = orc.generate_greedy(x, n)
    torch.manual_seed(1)
    voc = WavTokenizerDecoder(n_codes=4096, dim=256, intermediate_dim=512, num_layers=3).eval()
    with torch.no_grad():
        for name, p in voc.named_parameters():          # the decoder's default init leaves many tensors constant
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.2)
    ovoc = OracleVocoder({k: v.clone() for k, v in voc.state_dict().items() if k != "codebook"}, 3, 1280, 320)
    with torch.inference_mode():
        m = model.to("cuda")
        qs, atts, stops, cuts = m.generate_batch(x.cuda(), batch_size=B, max_seqlen=n, k=1, first_greedy_quant=0,
                                                 force_max_seqlen=True, device="cuda", engine="fused")
        codes = (undelay_rvq(qs) - m.n_special_token_in).clamp_min(0)                   # [q,B,n-q-1]
        vg = voc.to("cuda")
        bw = torch.zeros(1, dtype=torch.long, device="cuda")
        audio = vg(codes, bandwidth_id=bw).cpu()
    clear = (margins > 1e-3).all(dim=1)
    assert int(clear.sum()) >= B // 2
    assert torch.equal(qs.cpu()[0][clear], ref_toks[0][clear]), "decode tokens differ from the oracle"
    rcodes = (O.undelay_rvq(ref_toks) - 3).clamp_min(0)
    assert torch.equal(codes.cpu()[:, clear], rcodes[:, clear])
    feats = voc.codebook.detach().cpu()[0][rcodes[0]].transpose(1, 2).double()           # [B,C,L]
    ref_audio = ovoc.decode(feats, torch.zeros(1, dtype=torch.long))
    assert audio.shape == ref_audio.shape == (B, codes.shape[-1] * 320)
    err = (audio[clear].double() - ref_audio[clear]).abs().max() / ref_audio[clear].abs().max()
    record_parity("config 3 chain: waveform of the clear-margin rows vs oracle chain", err, 5e-4, clear_rows=int(clear.sum()), rows=B)
    assert err < 5e-4, f"waveform rel err {err:.3e}"


def test_multi_token_graph_replay_equals_single_steps(hip):
    """greedy_steps(n): groups of GRAPH_STEPS tokens per hipGraph replay + a remainder == n single-token replays
    (tokens and, after sync_state, the recurrent states)."""
    from lina_speech_amd.configs import tiny
    from lina_speech_amd.decode import DecodeEngine
    torch.manual_seed(2)
    model = tiny(d=256, heads=2, n_layer=1, n_codebook=500).eval().cuda()
    x = torch.randint(3, 256, (5, 13), device="cuda")
    with torch.inference_mode():
        x_enc = model.txt_encoder(model.txt_embed(x))
        a = DecodeEngine(model, x_enc, batch_size=5)
        a.begin_greedy(29)
        for _ in range(29):
            a.greedy_step()
        b = DecodeEngine(model, x_enc, batch_size=5)
        b.begin_greedy(29)
        b.greedy_steps(21)                         # 2 x 8 + 5
        b.greedy_steps(8)
        assert b._greedy_graph_n is not None
        assert torch.equal(a.greedy_tokens(), b.greedy_tokens())
        for sa, sb in zip(a.state.states, b.state.states):
            for ta, tb in zip(sa, sb):
                assert torch.equal(ta, tb)


@pytest.mark.parametrize("heads,expand_v", [(8, 1.0), (16, 1.0), (4, 2.0)])
def test_l169_other_head_shapes_match_cpu_oracle(hip, heads, expand_v):
    """The 169M hyper-parameters are inferred, not known: the same width with H = 8 (Dk = Dv = 128),
    H = 16 (64) and the mixer's default expand_v = 2 (Dv = 512; reference model/gla.py:50).  Device-side greedy loop
    (windowed K1w; Dv = 512 as two column halves per head) vs the fp32 CPU oracle:
    tokens identical wherever the oracle's top-2 margin exceeds 1e-3, rows comparable until their first near-tie."""
    from lina_speech_amd.configs import l169
    from lina_speech_amd.decode import DecodeEngine
    from oracle.lina_decode_oracle import OracleLina
    torch.manual_seed(0)
    model = l169(heads=heads, expand_v=expand_v).eval()
    B, n = 4, 10
    x = torch.randint(3, 256, (B, 16))
    orc = OracleLina(model.state_dict(), n_layer=6, heads=heads, txt_heads=4)
    with torch.inference_mode():
        m = model.to("cuda")
        x_enc = m.txt_encoder(m.txt_embed(x.cuda()))
        eng = DecodeEngine(m, x_enc, batch_size=B)
        assert eng.packs[0].lazy                             # every one of these head shapes runs the windowed K1w
        toks = eng.run_greedy(n).cpu()
    # the oracle teacher-forced on the engine's tokens: same history at every position, so every position is comparable
    ref_toks, _, _, margins = orc.generate_greedy(x, n, teacher=toks)
    safe = margins > 1e-3
    assert int(safe.sum()) >= B * n // 2, "too many near-ties for the comparison to mean anything"
    assert torch.equal(toks[0][safe], ref_toks[0][safe])


def test_config5_train_step_at_sequence_length_4096(hip):
    """BASELINE configs[4] at its sequence length (one rank's share, micro-batch 1): L169, 4096 codec tokens, bf16 autocast
    -- K2 in its segment-parallel form, K2b, K3b, K5b, fused AdamW; with GRAD_CKPT the same loss (reference switch
    model/gla.py:26-33).  The loss must be finite, reproducible under checkpointing and fall over three steps."""
    import os
    from lina_speech_amd import configs
    from lina_speech_amd.train import TrainStep, synthetic_batch
    batch = synthetic_batch(b=1, n=4097, t_txt=64, seed=3).to("cuda")
    losses = {}
    for ckpt in (False, True):
        if ckpt:
            os.environ["GRAD_CKPT"] = "1"
        try:
            torch.manual_seed(0)
            ts = TrainStep(configs.l169(), device=torch.device("cuda", 0), lr=1e-3, ddp=False, n_warmup_steps=0, grad_clip=1.0)
            losses[ckpt] = [float(ts.step(batch)) for _ in range(3)]
            del ts
            torch.cuda.empty_cache()
        finally:
            os.environ.pop("GRAD_CKPT", None)
    for l in losses.values():
        assert all(v == v and v < 1e4 for v in l), l
        assert l[-1] < l[0], l
    assert abs(losses[True][0] - losses[False][0]) < 1e-3 * abs(losses[False][0]), losses
