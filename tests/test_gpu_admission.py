"""Admission of new utterances into a running decode batch (`gpt.DecodeSession`, `itts_gpt_admit_rows`; design reference: the in-flight batching of
the reference's serving path, backends/trt/serving/triton_server.py:96-305, backends/trt/pipeline/pipeline.py:459-548).  The contract: a row
admitted at step k produces, bit for bit, the ids the same row produces decoded ALONE (with the same left padding), and the rows that were already
running are not disturbed -- in the f32 engine and in the bf16 engine."""
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

    # (2) the same batch; as soon as a row has finished the new utterance takes its slot
    with gpt.DecodeSession(m, emb, mask, mn, **hf) as s1:
        while not s1.finished():
            s1.run(8)
        slot = s1.finished()[0]
        before = s1.codes(slot).cpu()
        k, S_new = s1.steps, s1.position()
        s1.admit([slot], emb_n, mask_n)
        assert s1.col0[slot] == k - 1
        while s1.steps < max_new and len(s1.finished()) < B:
            s1.run(8)
        admitted = s1.codes(slot).cpu()
        others = {b: s1.codes(b).cpu() for b in range(B) if b != slot}
    assert torch.equal(before, alone_batch[slot])
    for b, v in others.items():
        assert torch.equal(v, alone_batch[b]), f"row {b} was disturbed by the admission"

    # (3) the new utterance alone, left-padded to the position it joined at: same ids, bit for bit
    extra = S_new - (emb_n.shape[1] + 1)
    emb_p = torch.cat([torch.zeros(1, extra, emb_n.shape[2], device=emb_n.device), emb_n.to(torch.float32)], dim=1)
    mask_p = torch.cat([torch.zeros(1, extra, dtype=mask_n.dtype, device=mask_n.device), mask_n], dim=1)
    with gpt.DecodeSession(m, emb_p, mask_p, mn, **hf) as s2:
        while s2.steps < max_new - (k - 1) and not s2.finished():
            s2.run(8)
        solo = s2.codes(0).cpu()
    n = min(int(solo.numel()), int(admitted.numel()))
    print(f"{prec}: admitted at step {k} into slot {slot} (position {S_new}): {admitted.numel()} codes, alone {solo.numel()}; first ids {admitted[:8].tolist()}")
    assert n >= 6 and torch.equal(admitted[:n], solo[:n])
    assert admitted.numel() == solo.numel() or admitted.numel() >= max_new - k     # (the admitted row may run into the batch's token budget)


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
                                         chunk_tokens=4, admit_room=24, **kw)
    st = m.last_inflight
    print(f"{prec}: in-flight schedule {st}; one batch {tuple(ref.shape)}, in flight {tuple(got.shape)}")
    assert st["admitted"] >= 2 and st["truncated"] == 0
    assert got.shape == ref.shape and torch.equal(got.cpu(), ref.cpu())
    again, _ = m.inference_speech(None, text[:3], langs[:3], emo_vec=emo, campplus_embedding=style, max_generate_length=max_new, **kw)
    assert torch.equal(again.cpu()[:, : ref.shape[1]], ref.cpu()[:3, : again.shape[1]])          # the engine is idle and usable again
