"""Admission of new utterances into a running decode batch (`gpt.DecodeSession`, `itts_gpt_admit_rows`; design reference: the in-flight batching of
the reference's serving path, backends/trt/serving/triton_server.py:96-305, backends/trt/pipeline/pipeline.py:459-548).  The contract: a row
admitted at ANY step k produces, bit for bit, the ids the same row produces decoded ALONE (every slot keeps its own cache position and its own
step), the rows that were already running are not disturbed, and a session outlives max_new_tokens / the mel position table -- in the f32 engine and
in the bf16 engine."""
import os

import numpy as np
import pytest
import torch

from oracle import gpt_oracle as G

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _engine(cfg, sd, prec):
    from indextts_amd import gpt
    m = gpt.UnifiedVoice(spk_cond_mode="campplus", layers=cfg.layers, model_dim=cfg.model_dim, heads=cfg.heads, max_text_tokens=cfg.max_text_tokens,
                         max_mel_tokens=cfg.max_mel_tokens, number_text_tokens=cfg.number_text_tokens, precision=prec, device=DEV)
    m.load_state_dict(sd)
    m.post_init_gpt2_config(kv_cache=True, half=prec == "bf16")
    return m


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_admitted_row_equals_the_row_alone(golden_dir, prec):
    from indextts_amd import gpt
    z = np.load(os.path.join(golden_dir, "gpt_greedy.npz"))
    c = z["cfg"]
    cfg = G.GPTConfig(layers=int(c[0]), model_dim=int(c[1]), heads=int(c[2]), max_text_tokens=int(c[3]), max_mel_tokens=int(c[4]),
                      number_text_tokens=int(c[5]))
    sd = G.synth_weights(cfg, seed=int(z["seed"]))
    sd["mel_head.bias"][cfg.stop_mel_token] += float(z["eos_bias"])          # rows stop at ragged steps
    m = _engine(cfg, sd, prec)
    style, emo = torch.from_numpy(z["style"]), torch.from_numpy(z["emo_vec"])
    text = torch.from_numpy(z["text"])
    B = text.shape[0]
    assert B >= 3
    gen = z["gen"]                                               # the fixture's own generation settings (greedy): its rows stop at steps 0 / 21 / 6
    kw = dict(do_sample=bool(gen[0]), num_beams=1, top_p=float(gen[2]), top_k=int(gen[3]), temperature=float(gen[4]), repetition_penalty=float(gen[5]))
    langs = torch.from_numpy(z["langs"])
    max_new = min(96, cfg.max_mel_tokens - 2)                    # the fixture model's mel position table is short

    def prep(t, lg):
        return m.inference_speech_stream(None, t, langs=lg, emo_vec=emo, campplus_embedding=style, max_generate_length=max_new, **kw)

    emb, mask, mn, hf = prep(text, langs)
    # the late utterance: the text of the fixture's longest-running row once more (random texts stop at once under this model's EOS bias)
    ref_codes = z["codes"]
    stop = int(ref_codes.max())
    ref_lens = [int((r == stop).argmax()) if (r == stop).any() else r.shape[0] for r in ref_codes]
    long_row = int(np.argmax(ref_lens))
    assert ref_lens[long_row] >= 6, ref_lens
    new_text = text[long_row:long_row + 1].contiguous()          # (trailing pad ids are stripped by prepare_gpt_inputs)
    emb_n, mask_n, _, _ = prep(new_text, langs[long_row:long_row + 1])

    # (1) the batch alone, to its end
    with gpt.DecodeSession(m, emb, mask, mn, **hf) as s0:
        while s0.steps < max_new and len(s0.finished()) < B:
            s0.run(8)
        alone_batch = [s0.codes(b).cpu() for b in range(B)]
    lens = [int(v.numel()) for v in alone_batch]
    assert min(lens) < max_new - 12, f"no row finishes early enough to free a slot: {lens}"

    # (0) the new utterance alone, exactly as it will be handed to admit(): the reference ids of every admission below
    with gpt.DecodeSession(m, emb_n, mask_n, mn, **hf) as s2:
        while s2.steps < max_new and not s2.finished():
            s2.run(8)
        solo = s2.codes(0).cpu()
    assert solo.numel() >= 6

    # (2) the same batch; as soon as a row has finished the new utterance takes its slot
    with gpt.DecodeSession(m, emb, mask, mn, **hf) as s1:
        while not s1.finished():
            s1.run(8)
        slot = s1.finished()[0]
        before = s1.codes(slot).cpu()
        k = s1.steps
        s1.admit([slot], emb_n, mask_n)
        assert s1.step0[slot] == k - 1 and s1.codes(slot).numel() == 1          # its first id sits in column 0 of its code row
        while len(s1.finished()) < B and s1.steps < 4 * max_new:
            s1.run(8)
        admitted = s1.codes(slot).cpu()
        others = {b: s1.codes(b).cpu() for b in range(B) if b != slot}
        # (3) the session keeps running: the same utterance once more, far past the first batch's max_new_tokens (the old shared position counter
        # had no room left there), into another slot
        while s1.steps < max_new + 9:
            s1.run(8)
        late_slot = [b for b in s1.finished() if b != slot][0]
        k2 = s1.steps
        s1.admit([late_slot], emb_n, mask_n)
        while late_slot not in s1.finished() and s1.steps < k2 + 2 * max_new:
            s1.run(8)
        late = s1.codes(late_slot).cpu()
    assert torch.equal(before, alone_batch[slot])
    for b, v in others.items():
        assert torch.equal(v, alone_batch[b]), f"row {b} was disturbed by the admission"
    print(f"{prec}: admitted at step {k} into slot {slot} and at step {k2} (max_new_tokens {max_new}) into slot {late_slot}: {admitted.numel()} / "
          f"{late.numel()} codes, alone {solo.numel()}; first ids {admitted[:8].tolist()}")
    assert k2 > max_new
    assert torch.equal(admitted, solo), "a row admitted into a running batch must generate the ids it generates alone"
    assert torch.equal(late, solo), "... also when it joins after the session's step counter has passed max_new_tokens"


def test_rejected_admission_leaves_the_running_rows_caps_alone(golden_dir):
    """`itts_gpt_admit_rows` writes the new utterances' token caps itself, after its checks: a call it rejects (slot still generating) must not
    have touched the cap of the row that is running there."""
    from indextts_amd import gpt, _lib
    z = np.load(os.path.join(golden_dir, "gpt_greedy.npz"))
    c = z["cfg"]
    cfg = G.GPTConfig(layers=int(c[0]), model_dim=int(c[1]), heads=int(c[2]), max_text_tokens=int(c[3]), max_mel_tokens=int(c[4]),
                      number_text_tokens=int(c[5]))
    sd = G.synth_weights(cfg, seed=int(z["seed"]))                          # no EOS bias: the rows run to their caps
    m = _engine(cfg, sd, "fp32")
    style, emo = torch.from_numpy(z["style"]), torch.from_numpy(z["emo_vec"])
    text, langs = torch.from_numpy(z["text"]), torch.from_numpy(z["langs"])
    max_new = min(40, cfg.max_mel_tokens - 2)
    emb, mask, mn, hf = m.inference_speech_stream(None, text, langs=langs, emo_vec=emo, campplus_embedding=style, max_generate_length=max_new,
                                                  do_sample=False, num_beams=1, repetition_penalty=10.0)
    B = text.shape[0]
    with gpt.DecodeSession(m, emb, mask, mn, **hf) as s:                       # uncapped: how far the rows run on their own
        while len(s.finished()) < B and s.steps < max_new:
            s.run(8)
        natural = [int(s.codes(b).numel()) for b in range(B)]
    assert natural[0] > 9 and natural[1] > 12, natural
    caps = [6] + [30] * (B - 1)
    with gpt.DecodeSession(m, emb, mask, mn, row_max_new=caps, **hf) as s:
        s.run(8)
        assert 0 in s.finished() and 1 not in s.finished()
        with pytest.raises(_lib.HipEngineError):
            s.admit([1], emb[:1], mask[:1], row_max_new=[3])            # slot 1 is still generating
        assert s._lim.tolist() == caps
        s.admit([0], emb[:1], mask[:1], row_max_new=[9])
        assert s._lim.tolist() == [9] + caps[1:]
        while len(s.finished()) < B and s.steps < 3 * max_new:
            s.run(8)
        assert s.codes(1).numel() == min(30, natural[1]) and s.codes(0).numel() == 9


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_inflight_equals_one_batch(golden_dir, prec):
    """`inference_speech_inflight` (2 decode slots, freed slots refilled from the waiting utterances) returns, bit for bit, the ids of
    `inference_speech` over all utterances in one batch (greedy)."""
    z = np.load(os.path.join(golden_dir, "gpt_greedy.npz"))
    c = z["cfg"]
    cfg = G.GPTConfig(layers=int(c[0]), model_dim=int(c[1]), heads=int(c[2]), max_text_tokens=int(c[3]), max_mel_tokens=int(c[4]),
                      number_text_tokens=int(c[5]))
    sd = G.synth_weights(cfg, seed=int(z["seed"]))
    sd["mel_head.bias"][cfg.stop_mel_token] += float(z["eos_bias"])
    m = _engine(cfg, sd, prec)
    style, emo = torch.from_numpy(z["style"]), torch.from_numpy(z["emo_vec"])
    order = [1, 0, 2, 1, 2, 0, 1]                                # the fixture's rows stop after 0 / 21 / 6 ids
    text = torch.from_numpy(z["text"])[order].contiguous()
    langs = torch.from_numpy(z["langs"])[order].contiguous()
    gen = z["gen"]
    kw = dict(do_sample=bool(gen[0]), num_beams=1, top_p=float(gen[2]), top_k=int(gen[3]), temperature=float(gen[4]), repetition_penalty=float(gen[5]))
    max_new = min(48, cfg.max_mel_tokens - 2)
    n = len(order)
    assert style.shape[0] == 1 and emo.shape[0] == 1 and n == 7              # one voice / emotion vector for every row, as in the fixture
    ref, _ = m.inference_speech(None, text, langs, emo_vec=emo, campplus_embedding=style, max_generate_length=max_new, **kw)
    got, _ = m.inference_speech_inflight(None, text, langs, emo_vec=emo, campplus_embedding=style, max_generate_length=max_new, slots=2,
                                         chunk_tokens=4, **kw)
    st = m.last_inflight
    print(f"{prec}: in-flight schedule {st}; one batch {tuple(ref.shape)}, in flight {tuple(got.shape)}")
    assert st["admitted"] == n - 2 and st["sessions"] == 1 and st["truncated"] == 0
    assert got.shape == ref.shape and torch.equal(got.cpu(), ref.cpu())
    again, _ = m.inference_speech(None, text[:3], langs[:3], emo_vec=emo, campplus_embedding=style, max_generate_length=max_new, **kw)
    assert torch.equal(again.cpu()[:, : ref.shape[1]], ref.cpu()[:3, : again.shape[1]])          # the engine is idle and usable again


def test_inflight_at_production_widths_equals_one_batch():
    """The in-flight schedule at the production GPT widths (24 x 1280, bf16, the benchmark's engine): 20 ragged utterances on 8 decode slots,
    greedy, lengths given as per-utterance caps (the synthetic weights never emit the stop token) -- bit for bit the ids of one 20-row batch, with
    every waiting utterance admitted into ONE session whose step counter passes the longest cap."""
    from indextts_amd import gpt, synth
    cfg = dict(synth.GPT_V25)
    m = gpt.UnifiedVoice(**cfg, spk_cond_mode="campplus", precision="bf16", device=DEV)
    m.load_state_dict(synth.gpt_weights(cfg, seed=1234, suppress_eos=True))
    m.post_init_gpt2_config(kv_cache=True, half=True)
    g = torch.Generator().manual_seed(77)
    n, slots, hi = 20, 8, 40
    caps = torch.randint(6, hi + 1, (n,), generator=g).tolist()
    lens = torch.randint(20, 65, (n,), generator=g).tolist()                  # ragged texts: the prompts are left-padded among themselves
    text = torch.ones(n, 65, dtype=torch.int32)
    for i, L in enumerate(lens):
        text[i, :L] = torch.randint(2, 12000, (L,), generator=g).to(torch.int32)
    text = text.to(DEV)
    langs = torch.full((n,), 3, dtype=torch.long, device=DEV)
    style = (torch.randn(1, 192, generator=g) * 0.1).to(DEV)
    emo = (torch.randn(1, cfg["model_dim"], generator=g) * 0.1).to(DEV)
    kw = dict(emo_vec=emo, campplus_embedding=style, max_generate_length=hi, do_sample=False, num_beams=1, repetition_penalty=10.0)
    ref, _ = m.inference_speech(None, text, langs=langs, row_max_new=caps, **kw)
    got, _ = m.inference_speech_inflight(None, text, langs=langs, slots=slots, chunk_tokens=16, min_free=2, row_max_new=caps, **kw)
    st = m.last_inflight
    print(f"production widths: in-flight schedule {st}; caps {caps}")
    assert st["sessions"] == 1 and st["admitted"] == n - slots and st["steps"] > hi
    stop = m.stop_mel_token
    for i in range(n):
        a, b = ref[i].tolist(), got[i].tolist()
        la = a.index(stop) if stop in a else len(a)
        lb = b.index(stop) if stop in b else len(b)
        assert la == lb == caps[i] and a[:la] == b[:lb], f"utterance {i}: one batch {a[:la]} vs in flight {b[:lb]}"


def test_chunk_call_returns_at_the_flag_check_once_a_slot_can_be_refilled(golden_dir):
    """`DecodeSession.run(n, return_when_finished=k)` (itts_gpt_set_chunk_return): the chunk comes back at one of the engine's own flag checks (every 8
    steps) as soon as k utterances have finished, instead of running all n steps; without it the same call runs to n.  Ids are those of an undisturbed run."""
    from indextts_amd import gpt
    z = np.load(os.path.join(golden_dir, "gpt_greedy.npz"))
    c = z["cfg"]
    cfg = G.GPTConfig(layers=int(c[0]), model_dim=int(c[1]), heads=int(c[2]), max_text_tokens=int(c[3]), max_mel_tokens=int(c[4]),
                      number_text_tokens=int(c[5]))
    sd = G.synth_weights(cfg, seed=int(z["seed"]))
    m = _engine(cfg, sd, "fp32")
    style, emo = torch.from_numpy(z["style"]), torch.from_numpy(z["emo_vec"])
    text, langs = torch.from_numpy(z["text"]), torch.from_numpy(z["langs"])
    max_new = min(48, cfg.max_mel_tokens - 2)
    emb, mask, mn, hf = m.inference_speech_stream(None, text, langs=langs, emo_vec=emo, campplus_embedding=style, max_generate_length=max_new,
                                                  do_sample=False, num_beams=1, repetition_penalty=10.0)
    B = text.shape[0]
    caps = [5] + [40] * (B - 1)                                   # utterance 0 stops after 5 codes: finished at the flag check of step 8
    with gpt.DecodeSession(m, emb, mask, mn, row_max_new=caps, **hf) as s:
        assert s.run(40) == 40
        full = [s.codes(b).cpu() for b in range(B)]
    with gpt.DecodeSession(m, emb, mask, mn, row_max_new=caps, **hf) as s:
        n = s.run(40, return_when_finished=1)
        assert n == 8 and s.finished() == [0], (n, s.finished())
        n = s.run(40, return_when_finished=2)                     # nobody else finishes before its cap
        assert n == 48 or n % 8 == 0
        while s.steps < 40:
            s.run(40 - s.steps)
        again = [s.codes(b).cpu()[:full[b].numel()] for b in range(B)]
    for b in range(B):
        assert torch.equal(again[b], full[b])
