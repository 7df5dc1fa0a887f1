"""GPU: row compaction of ragged decode batches (itts_gpt_set_compaction) and per-row token caps (itts_gpt_set_row_limits).

Finished utterances leave the running batch in buckets of rows; survivors keep their KV-cache rows behind a slot map.  A row's
arithmetic does not depend on the batch it runs in, so the ids must be IDENTICAL with compaction on and off -- greedy and sampled (a
fixed uniform stream, indexed by utterance), f32 and bf16 engines, rows finishing at scattered steps, the last survivors alone."""
import numpy as np
import pytest
import torch

from oracle import gpt_oracle as G

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _engine(cfg, sd, precision):
    from indextts_amd import gpt
    m = gpt.UnifiedVoice(spk_cond_mode="campplus", layers=cfg.layers, model_dim=cfg.model_dim, heads=cfg.heads,
                         max_text_tokens=cfg.max_text_tokens, max_mel_tokens=cfg.max_mel_tokens, number_text_tokens=cfg.number_text_tokens,
                         precision=precision, device=DEV)
    m.load_state_dict(sd)
    m.post_init_gpt2_config(kv_cache=True)
    return m


def _case(B=40, L=12, seed=5):
    cfg = G.GPTConfig(layers=3, model_dim=128, heads=2, max_text_tokens=40, max_mel_tokens=120, number_text_tokens=200)
    sd = G.synth_weights(cfg, seed=seed)
    sd["mel_head.bias"][cfg.stop_mel_token] -= 1e4                 # rows end at their own caps (set below), not at a sampled EOS
    g = torch.Generator().manual_seed(seed + 1)
    text = torch.randint(2, cfg.number_text_tokens, (B, L), generator=g)
    for b in range(0, B, 3):
        text[b, L - 1 - (b % 5):] = 1                               # ragged prompts as well
    langs = torch.randint(0, cfg.n_langs, (B,), generator=g)
    style = torch.randn(1, 192, generator=g)
    emo = torch.randn(1, cfg.model_dim, generator=g) * 0.1
    limits = torch.randint(9, 97, (B,), generator=g).tolist()
    limits[7], limits[B - 1] = 100, 3                               # one row outlives all the others by a margin; one stops at once
    return cfg, sd, text, langs, style, emo, limits


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("sample", [False, True])
def test_compaction_leaves_ids_unchanged(precision, sample):
    cfg, sd, text, langs, style, emo, limits = _case()
    B, n = text.shape[0], 100
    m = _engine(cfg, sd, precision)
    kw = dict(max_generate_length=n, num_beams=1, repetition_penalty=10.0, row_max_new=limits)
    if sample:
        kw.update(do_sample=True, top_k=30, top_p=0.8, temperature=1.1,
                  uniforms=torch.rand(n, B, generator=torch.Generator().manual_seed(2), dtype=torch.float64))
    else:
        kw.update(do_sample=False)
    outs, stats = {}, {}
    for name, on in (("off", False), ("on", True), ("on4", True)):
        m.set_compaction(on, 4 if name == "on4" else 8)
        ids, _ = m.inference_speech(None, text, langs=langs, emo_vec=emo, campplus_embedding=style, **kw)
        outs[name], stats[name] = ids.cpu().numpy(), dict(m.last_timing)
    m.set_compaction(True, 8)
    assert np.array_equal(outs["off"], outs["on"]) and np.array_equal(outs["off"], outs["on4"])
    # the caps were honoured: row b holds exactly limits[b] tokens before its first stop token
    got = outs["on"]
    for b in range(B):
        stop = np.nonzero(got[b] == cfg.stop_mel_token)[0]
        assert (int(stop[0]) if stop.size else got.shape[1]) == min(limits[b], got.shape[1]), (b, limits[b])
    steps = stats["off"]["steps"]
    assert stats["off"]["compactions"] == 0 and stats["off"]["row_steps"] == B * (steps - 1)
    assert stats["on"]["compactions"] >= 3 and stats["on4"]["compactions"] >= stats["on"]["compactions"]
    assert stats["on4"]["row_steps"] <= stats["on"]["row_steps"] < 0.8 * stats["off"]["row_steps"]
    print(f"{precision} {'sampled' if sample else 'greedy'}: {steps} steps, row-steps {stats['off']['row_steps']} -> {stats['on']['row_steps']} "
          f"(8-row buckets, {stats['on']['compactions']} compactions) / {stats['on4']['row_steps']} (4-row buckets)")


def test_compacted_rows_match_oracle_ids():
    """f32 engine, ragged caps, compaction on: every row's ids up to its cap equal the CPU oracle's greedy ids for that row."""
    cfg, sd, text, langs, style, emo, limits = _case(B=12, seed=9)
    m = _engine(cfg, sd, "fp32")
    n = 60
    limits = [min(v, n) for v in limits]
    ids, _ = m.inference_speech(None, text, langs=langs, emo_vec=emo, campplus_embedding=style, max_generate_length=n, do_sample=False,
                                num_beams=1, repetition_penalty=10.0, row_max_new=limits)
    assert m.last_timing["compactions"] >= 1
    with torch.no_grad():
        ref = G.inference_speech(sd, cfg, G.conds_latent_campplus(sd, style, emo), text, langs, G.GenParams(max_generate_length=n)).numpy()
    got = ids.cpu().numpy()
    for b in range(text.shape[0]):
        k = min(limits[b], got.shape[1])
        assert np.array_equal(got[b, :k], ref[b, :k]), b
