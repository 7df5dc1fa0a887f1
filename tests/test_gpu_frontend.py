"""GPU test of `EngineFrontend` (indextts_amd/frontend.py): the prompt side of `IndexTTS2` built from a checkpoint DIRECTORY without the
reference package -- w2v-bert-2.0 (config.json + safetensors), its statistics, CAMPPlus, the s2mel checkpoint's length regulator, the
emotion / speaker matrices, a WAV prompt -- every network and DSP step on the engine.  A synthetic directory with the reference's file
layout and names is written here (small w2v-bert / regulator widths, the real CAMPPlus architecture, oracle-seeded weights) and the speaker
bundle is compared with the ORACLE chain on the same file: resample -> SeamlessM4T features -> w2v-bert hidden state -> (x - mean) / std;
log-mel; Kaldi fbank -> CAMPPlus; length regulator."""
import json

import numpy as np
import pytest
import torch

from oracle import audio_oracle as AO
from oracle import campplus_oracle as CO
from oracle import codec_oracle as KO
from oracle import w2vbert_oracle as WO
from tools.make_golden_audio import speechlike

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
WCFG = WO.W2VBertCfg(hidden_size=64, num_hidden_layers=4, num_attention_heads=2, intermediate_size=128, feature_projection_input_dim=160,
                     left_max_position_embeddings=6, right_max_position_embeddings=2, conv_depthwise_kernel_size=7)
RCFG = KO.RegulatorConfig(channels=64, in_channels=64, n_layers=4, groups=1, codebook_size=64)


def write_checkpoint_dir(d):
    from safetensors.torch import save_file
    from scipy.io import wavfile
    g = torch.Generator().manual_seed(5)
    wdir = d / "hf_cache" / "w2v-bert-2.0"
    wdir.mkdir(parents=True)
    (wdir / "config.json").write_text(json.dumps(dict(
        hidden_size=WCFG.hidden_size, num_hidden_layers=WCFG.num_hidden_layers, num_attention_heads=WCFG.num_attention_heads,
        intermediate_size=WCFG.intermediate_size, feature_projection_input_dim=WCFG.feature_projection_input_dim,
        position_embeddings_type="relative_key", left_max_position_embeddings=WCFG.left_max_position_embeddings,
        right_max_position_embeddings=WCFG.right_max_position_embeddings, conv_depthwise_kernel_size=WCFG.conv_depthwise_kernel_size,
        hidden_act="swish", layer_norm_eps=WCFG.layer_norm_eps, add_adapter=False, model_type="wav2vec2-bert", vocab_size=None)))
    wsd = WO.synth_weights(WCFG)
    save_file({k: v.contiguous() for k, v in wsd.items()}, str(wdir / "model.safetensors"))
    stats = {"mean": 0.3 * torch.randn(WCFG.hidden_size, generator=g), "var": 0.5 + torch.rand(WCFG.hidden_size, generator=g)}
    torch.save(stats, d / "wav2vec2bert_stats.pt")
    csd = CO.synth_weights()
    torch.save(csd, d / "hf_cache" / "campplus_cn_common.bin")
    rsd = KO.synth_regulator_weights(RCFG, 9)
    torch.save({"net": {"cfm": {"module.placeholder": torch.zeros(1)}, "length_regulator": {"module." + k: v for k, v in rsd.items()}}}, d / "s2mel.pth")
    emo, spk = torch.randn(5, 32, generator=g), torch.randn(5, 192, generator=g)
    torch.save(emo, d / "emo.pt")
    torch.save({"model": {"up.bias": torch.zeros(4)}}, d / "codec.pth")
    torch.save(spk, d / "spk.pt")
    wave = speechlike(48000, 24000, 61)                                   # 2 s at 24 kHz -> 198 Kaldi frames at 16 kHz (even: no padded pair)
    pcm = np.round(wave * 32767.0).astype(np.int16)
    wavfile.write(str(d / "prompt.wav"), 24000, pcm)
    cfg = {"w2v_stat": "wav2vec2bert_stats.pt", "s2mel_checkpoint": "s2mel.pth", "emo_matrix": "emo.pt", "spk_matrix": "spk.pt", "emo_num": [2, 3],
           "s2mel": {"length_regulator": {"channels": RCFG.channels, "sampling_ratios": [1, 1, 1, 1], "is_discrete": False,
                                          "in_channels": RCFG.in_channels, "content_codebook_size": RCFG.codebook_size},
                     "preprocess_params": {"sr": 22050, "spect_params": {"n_fft": 1024, "win_length": 1024, "hop_length": 256, "n_mels": 80,
                                                                         "fmin": 0, "fmax": "None"}}}}
    return cfg, dict(wsd=wsd, stats=stats, csd=csd, rsd=rsd, emo=emo, spk=spk, pcm=pcm)


def test_speaker_bundle_from_checkpoint_directory_vs_oracle_chain(tmp_path):
    from indextts_amd.frontend import EngineFrontend, load_wav
    d = tmp_path / "ckpt"
    cfg, w = write_checkpoint_dir(d)
    fe = EngineFrontend(cfg, str(d), DEV)
    x, sr = load_wav(str(d / "prompt.wav"))
    assert sr == 24000 and x.shape == (1, 48000) and torch.equal(x[0], torch.from_numpy(w["pcm"].astype(np.float32) / 32768.0))
    b = fe.speaker_bundle(str(d / "prompt.wav"))
    # the oracle chain on the same samples (infer_v2_5.py:626-656; the file arrives at 22.05 kHz like librosa.load's default)
    a22 = AO.resample(x, 24000, 22050)
    a16 = AO.resample(a22, 22050, 16000)
    feats, mask = AO.seamless_features(a16[0].numpy())
    assert feats.shape[1] == 99 and int(mask.sum()) == 99
    emb = WO.get_emb(w["wsd"], WCFG, torch.from_numpy(feats), torch.from_numpy(mask).long(), w["stats"]["mean"], torch.sqrt(w["stats"]["var"]),
                     layer=WCFG.num_hidden_layers)
    mel = AO.mel_spectrogram(a22)
    fb = AO.kaldi_fbank(a16)
    style = CO.campplus(w["csd"], (fb - fb.mean(dim=0, keepdim=True))[None])
    cond, _ = KO.length_regulator(w["rsd"], RCFG, emb, torch.tensor([mel.shape[2]]))
    err = {k: float((b[k].cpu() - v).abs().max()) for k, v in (("spk_cond_emb", emb), ("ref_mel", mel), ("style", style), ("prompt_condition", cond))}
    print("EngineFrontend.speaker_bundle vs the oracle chain, max|d|:", {k: f"{v:.2e}" for k, v in err.items()},
          "| scales:", f"emb {float(emb.abs().max()):.1f}, style {float(style.abs().max()):.1f}, cond {float(cond.abs().max()):.1f}")
    assert b["spk_cond_emb"].shape == emb.shape == (1, 99, 64) and b["ref_mel"].shape == mel.shape and b["style"].shape == (1, 192)
    assert b["prompt_condition"].shape == cond.shape == (1, mel.shape[2], 64)
    # every stage sees the float32 DSP's rounding of the stage before it (quiet fbank bins differ by up to 1e-3, test_gpu_audio.py): the bars are
    # relative to each output's scale -- 1e-4 of the largest value for the three network outputs (measured 1e-6 .. 1e-5), 5e-4 absolute on the log-mel (1e-5)
    assert err["ref_mel"] <= 5e-4
    assert err["spk_cond_emb"] <= 1e-4 * float(emb.abs().max()) and err["style"] <= 1e-4 * float(style.abs().max())
    assert err["prompt_condition"] <= 1e-4 * float(cond.abs().max())
    # emotion prompt: the same file at 16 kHz directly (librosa.load(path, sr=16000), :687)
    e = fe.emo_cond(str(d / "prompt.wav"))
    f16, m16 = AO.seamless_features(AO.resample(x, 24000, 16000)[0].numpy())
    e_o = WO.get_emb(w["wsd"], WCFG, torch.from_numpy(f16), torch.from_numpy(m16).long(), w["stats"]["mean"], torch.sqrt(w["stats"]["var"]),
                     layer=WCFG.num_hidden_layers)
    valid = torch.from_numpy(m16[0]).bool()
    assert e.shape == e_o.shape and float((e.cpu() - e_o)[0, valid].abs().max()) <= 1e-4 * float(e_o.abs().max())
    # emotion-vector mixing (:669-680): nearest speaker row per emotion group by cosine similarity, weighted sum of the emotion rows
    vec = [0.3, 0.5]
    mat, wsum = fe.emo_vector_mix(vec, b["style"], use_random=False)
    q = b["style"].cpu().float()
    groups_s, groups_e = torch.split(w["spk"], [2, 3]), torch.split(w["emo"], [2, 3])
    idx = [int(torch.argmax(torch.nn.functional.cosine_similarity(q, m, dim=1))) for m in groups_s]
    ref = sum(v * ge[i] for v, ge, i in zip(vec, groups_e, idx))[None]
    assert float((mat.cpu() - ref).abs().max()) <= 1e-6 and abs(float(wsum) - 0.8) <= 1e-6
    sds = fe.engine_state_dicts()
    assert set(sds) == {"semantic_codec", "cfm", "length_regulator"} and "placeholder" in sds["cfm"] and "up.bias" in sds["semantic_codec"]
    with pytest.raises(RuntimeError):
        fe.text_segments("no text front end was injected", "en", 120, True, 600)
    with pytest.raises(RuntimeError):
        fe.merge_emovec(b["spk_cond_emb"], e, 1.0)


def test_v1_conditioning_mel_from_a_wav_file(tmp_path):
    """`EngineFrontendV1.cond_mel` (indextts/infer.py:303-323): WAV -> mono -> 24 kHz -> truncation -> MelSpectrogramFeatures, vs the oracle chain."""
    from scipy.io import wavfile
    from indextts_amd.frontend import EngineFrontendV1
    wave = speechlike(30000, 22050, 62)
    stereo = np.stack([wave, 0.5 * wave[::-1]], axis=1)
    wavfile.write(str(tmp_path / "v1.wav"), 22050, stereo.astype(np.float32))
    fe = EngineFrontendV1(DEV)
    mono = torch.from_numpy(stereo.astype(np.float32).mean(axis=1))[None]
    a24 = AO.resample(mono, 22050, 24000)
    for trunc in (None, 1.0):
        m = fe.cond_mel(str(tmp_path / "v1.wav"), truncate_seconds=trunc)
        ref = AO.mel_spectrogram_features(a24 if trunc is None else a24[:, :24000])
        assert m.shape == ref.shape and m.shape[1] == 100 and float((m.cpu() - ref).abs().max()) <= 5e-4
    with pytest.raises(RuntimeError):
        fe.conditioning(m, torch.tensor([m.shape[-1]]))


def test_v2_frontend_prompt_condition_goes_through_quantize(tmp_path):
    """`EngineFrontendV2` (indextts/infer_v2.py:138-142,465-479): the semantic codec read from safetensors with its encoder half; prompt_condition =
    length_regulator(quantize(spk_cond_emb)) vs the oracle chain; the other bundle entries equal the v2.5 frontend's."""
    from safetensors.torch import save_file
    from indextts_amd.frontend import EngineFrontend, EngineFrontendV2
    from tools.make_golden_codec_quantize import CFG as QCFG, SEED as QSEED
    d = tmp_path / "ckpt"
    cfg, w = write_checkpoint_dir(d)
    csd = KO.synth_codec_weights(QCFG, QSEED)
    csd.update(KO.synth_codec_encoder_weights(QCFG, QSEED + 1))
    (d / "hf_cache" / "semantic_codec").mkdir(parents=True)
    save_file({k: v.contiguous() for k, v in csd.items()}, str(d / "hf_cache" / "semantic_codec" / "model.safetensors"))
    cfg = dict(cfg, semantic_codec=dict(codebook_size=QCFG.codebook_size, hidden_size=QCFG.hidden_size, codebook_dim=QCFG.codebook_dim,
                                        vocos_dim=QCFG.vocos_dim, vocos_intermediate_dim=QCFG.vocos_intermediate_dim,
                                        vocos_num_layers=QCFG.vocos_num_layers))
    assert QCFG.hidden_size == WCFG.hidden_size == RCFG.in_channels
    fe2 = EngineFrontendV2(cfg, str(d), DEV)
    b2 = fe2.speaker_bundle(str(d / "prompt.wav"))
    b = EngineFrontend(cfg, str(d), DEV).speaker_bundle(str(d / "prompt.wav"))
    for k in ("spk_cond_emb", "ref_mel", "style"):
        assert torch.equal(b2[k], b[k])
    idx_o, q_o, margin = KO.codec_quantize(csd, QCFG, b["spk_cond_emb"].cpu())
    idx_e, q_e = fe2.codec.quantize(b["spk_cond_emb"])
    same = idx_e.cpu() == idx_o
    assert bool(same[margin > 1e-4].all()) and same.float().mean() > 0.9
    cond_o, _ = KO.length_regulator(w["rsd"], RCFG, q_e.cpu(), torch.tensor([b["ref_mel"].shape[2]]))       # regulator on the engine's own codes
    err = float((b2["prompt_condition"].cpu() - cond_o).abs().max())
    print(f"v2 prompt_condition: {int(same.sum())}/{same.numel()} codes equal the oracle's (smallest margin {float(margin.min()):.1e}), "
          f"regulator max|d| {err:.2e}")
    assert b2["prompt_condition"].shape == b["prompt_condition"].shape and err <= 1e-4 * float(cond_o.abs().max()) + 1e-5
    assert float((b2["prompt_condition"] - b["prompt_condition"]).abs().max()) > 1e-3           # not the v2.5 path
    assert set(fe2.engine_state_dicts()) == {"semantic_codec", "cfm", "length_regulator"}
