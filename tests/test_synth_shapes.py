"""The product-side random-weight generators (indextts_amd/synth.py: what bench.py times) carry the same tensor names and shapes as the oracle's
generators, which are pinned to the reference classes -- so a bench line cannot run on a mis-shaped architecture."""
import torch

from indextts_amd import synth
from oracle import bigvgan_oracle as BO
from oracle import ecapa_oracle as EO


def _shapes(sd, skip=()):
    return {k: tuple(v.shape) for k, v in sd.items() if not k.endswith("num_batches_tracked") and not k.startswith(skip)}


def test_ecapa_generator_matches_the_oracle_names_and_shapes():
    assert _shapes(synth.ecapa_weights()) == _shapes(EO.synth_weights(EO.EcapaCfg()))


def test_v1_vocoder_generator_matches_the_oracle_names_and_shapes():
    h = dict(synth.BIGVGAN_V1_24K)
    hb = dict(BO.V2_HPARAMS, **{k: h[k] for k in ("upsample_rates", "upsample_kernel_sizes", "use_tanh_at_final", "use_bias_at_final",
                                                  "upsample_initial_channel")})
    ours = synth.bigvgan_v1_weights(h)
    ref = BO.synth_weights(hb, seed=1, cond_dim=h["speaker_embedding_dim"], in_dim=h["gpt_dim"])
    assert _shapes(ours, skip=("speaker_encoder.",)) == _shapes(ref)
    enc = {k[len("speaker_encoder."):]: v for k, v in ours.items() if k.startswith("speaker_encoder.")}
    assert _shapes(enc) == _shapes(EO.synth_weights(EO.EcapaCfg(input_size=h["num_mels"], lin_neurons=h["speaker_embedding_dim"])))
    total_up = 1
    for u in h["upsample_rates"]:
        total_up *= u
    assert total_up == 1024                                   # 96 latent frames -> 98 304 samples


def test_v1_gpt_generator_has_the_v1_host_tensors_only():
    cfg = dict(synth.GPT_V15, layers=1, model_dim=64, heads=2)
    sd = synth.gpt_v1_weights(cfg)
    assert "lang_embedding.weight" not in sd and "spk_emb_proj.weight" not in sd
    assert sd["mel_pos_embedding.emb.weight"].shape == (cfg["max_mel_tokens"] + 2 + cfg["max_conditioning_inputs"], 64)
    assert sd["text_pos_embedding.emb.weight"].shape == (cfg["max_text_tokens"] + 2, 64)
