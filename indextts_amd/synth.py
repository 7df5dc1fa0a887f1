"""Random-init weights with the reference checkpoint names and shapes, for benchmarking without checkpoints.

There are no model checkpoints in the build/bench environment (SURVEY.md header), so `bench.py` times the engine on
seeded random weights of the IndexTTS-2.5 architecture.  This is NOT the test oracle's generator (tests use
oracle.*.synth_weights); it lives in the package so that the product/bench path never imports `oracle/`.
"""
import math
from typing import Dict

import torch

GPT_V25 = dict(layers=24, model_dim=1280, heads=20, max_text_tokens=600, max_mel_tokens=1815, number_text_tokens=12000,
               number_mel_codes=8194, start_mel_token=8192, stop_mel_token=8193, start_text_token=0, stop_text_token=1,
               max_conditioning_inputs=1, types=1)
N_LANGS = 101

BIGVGAN_V2_22K = dict(num_mels=80, upsample_rates=[4, 4, 2, 2, 2, 2], upsample_kernel_sizes=[8, 8, 4, 4, 4, 4],
                      upsample_initial_channel=1536, resblock_kernel_sizes=[3, 7, 11],
                      resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]], activation="snakebeta",
                      snake_logscale=True, use_tanh_at_final=False, use_bias_at_final=False, resblock="1",
                      sampling_rate=22050, hop_size=256)


def gpt_weights(cfg: dict = GPT_V25, seed: int = 1234, suppress_eos: bool = False) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    D, L, V = cfg["model_dim"], cfg["layers"], cfg["number_mel_codes"]
    sd = {}
    rn = lambda *s, std=0.02: torch.randn(*s, generator=g) * std
    for i in range(L):
        p = f"gpt.h.{i}."
        for ln in ("ln_1", "ln_2"):
            sd[p + ln + ".weight"] = 1.0 + rn(D, std=0.05)
            sd[p + ln + ".bias"] = rn(D)
        sd[p + "attn.c_attn.weight"] = rn(D, 3 * D, std=0.05)
        sd[p + "attn.c_attn.bias"] = rn(3 * D)
        sd[p + "attn.c_proj.weight"] = rn(D, D, std=0.05 / math.sqrt(2 * L))
        sd[p + "attn.c_proj.bias"] = rn(D)
        sd[p + "mlp.c_fc.weight"] = rn(D, 4 * D, std=0.05)
        sd[p + "mlp.c_fc.bias"] = rn(4 * D)
        sd[p + "mlp.c_proj.weight"] = rn(4 * D, D, std=0.05 / math.sqrt(2 * L))
        sd[p + "mlp.c_proj.bias"] = rn(D)
    for ln in ("gpt.ln_f", "final_norm"):
        sd[ln + ".weight"] = 1.0 + rn(D, std=0.05)
        sd[ln + ".bias"] = rn(D)
    sd["mel_head.weight"] = rn(V, D, std=0.08)
    sd["mel_head.bias"] = rn(V)
    if suppress_eos:
        sd["mel_head.bias"][cfg["stop_mel_token"]] = -1e4     # fixed-length decode for timing
    sd["mel_embedding.weight"] = rn(V, D, std=0.5)
    sd["mel_pos_embedding.emb.weight"] = rn(cfg["max_mel_tokens"] + 2 + cfg["max_conditioning_inputs"], D, std=0.3)
    sd["text_embedding.weight"] = rn(cfg["number_text_tokens"] * cfg["types"] + 1, D, std=0.5)
    sd["text_pos_embedding.emb.weight"] = rn(cfg["max_text_tokens"] + 2, D, std=0.3)
    sd["lang_embedding.weight"] = rn(N_LANGS, D, std=0.3)
    sd["spk_emb_proj.weight"] = rn(D, 192, std=0.05)
    sd["spk_emb_proj.bias"] = rn(D)
    return sd


def kaiser_sinc_filter12() -> torch.Tensor:
    """12-tap Kaiser-windowed sinc, cutoff 0.25, half-width 0.3 (the buffer UpSample1d/DownSample1d register)."""
    half, K, cutoff, hw = 6, 12, 0.25, 0.3
    A = 2.285 * (half - 1) * math.pi * 4 * hw + 7.95
    beta = 0.1102 * (A - 8.7) if A > 50 else (0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21) if A >= 21 else 0.0)
    win = torch.kaiser_window(K, beta=beta, periodic=False)
    t = torch.arange(-half, half) + 0.5
    f = 2 * cutoff * win * torch.sinc(2 * cutoff * t)
    return (f / f.sum()).float()


def bigvgan_weights(h: dict = BIGVGAN_V2_22K, seed: int = 1234, post_gain: float = 0.04) -> Dict[str, torch.Tensor]:
    """Variance-preserving init (std 1/sqrt(C_in*k)) so the waveform has RMS ~0.2 instead of collapsing to zero."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    filt = kaiser_sinc_filter12().view(1, 1, 12)

    def conv(name, cout, cin, k, bias=True, gain=1.0):
        sd[name + ".weight"] = torch.randn(cout, cin, k, generator=g) * (gain / math.sqrt(cin * k))
        if bias:
            sd[name + ".bias"] = torch.randn(cout, generator=g) * 0.02

    def act(name, ch):
        sd[name + ".act.alpha"] = torch.rand(ch, generator=g) - 0.5
        sd[name + ".act.beta"] = torch.rand(ch, generator=g) - 0.5
        sd[name + ".upsample.filter"] = filt.clone()
        sd[name + ".downsample.lowpass.filter"] = filt.clone()

    c0 = h["upsample_initial_channel"]
    conv("conv_pre", c0, h["num_mels"], 7)
    nk = len(h["resblock_kernel_sizes"])
    ch = c0
    for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
        cin, ch = c0 // 2 ** i, c0 // 2 ** (i + 1)
        sd[f"ups.{i}.0.weight"] = torch.randn(cin, ch, k, generator=g) / math.sqrt(cin * k / u)
        sd[f"ups.{i}.0.bias"] = torch.randn(ch, generator=g) * 0.02
        for j, kk in enumerate(h["resblock_kernel_sizes"]):
            n = i * nk + j
            nd = len(h["resblock_dilation_sizes"][j])
            for d in range(nd):
                conv(f"resblocks.{n}.convs1.{d}", ch, ch, kk, gain=0.7)
                conv(f"resblocks.{n}.convs2.{d}", ch, ch, kk, gain=0.5)
            for m in range(2 * nd):
                act(f"resblocks.{n}.activations.{m}", ch)
    act("activation_post", ch)
    conv("conv_post", 1, ch, 7, bias=h.get("use_bias_at_final", True), gain=post_gain)
    return sd


# ---- IndexTTS-1.5 (BASELINE.json configs[0]): the same 24 x 1280 GPT-2 stack behind 32 Conformer / Perceiver latents (no language
# embedding, no style projection; 800 mel positions), and the v1 vocoder: BigVGAN on the GPT latent (gpt_dim 1280) with 1024 samples per
# latent frame at 24 kHz, a 512-d ECAPA-TDNN speaker embedding (100-band reference mel) added after conv_pre and every upsampler
# (indextts/BigVGAN/models.py:139-250; checkpoints/config.yaml is not in the repository: rates as SURVEY.md section 8 config 1) ----------
GPT_V15 = dict(layers=24, model_dim=1280, heads=20, max_text_tokens=600, max_mel_tokens=800, number_text_tokens=12000,
               number_mel_codes=8194, start_mel_token=8192, stop_mel_token=8193, start_text_token=0, stop_text_token=1,
               max_conditioning_inputs=1, types=1)
BIGVGAN_V1_24K = dict(num_mels=100, upsample_rates=[4, 4, 4, 4, 2, 2], upsample_kernel_sizes=[8, 8, 4, 4, 4, 4],
                      upsample_initial_channel=1536, resblock_kernel_sizes=[3, 7, 11],
                      resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]], activation="snakebeta",
                      snake_logscale=True, use_tanh_at_final=True, use_bias_at_final=True, resblock="1",
                      sampling_rate=24000, hop_size=256, gpt_dim=1280, speaker_embedding_dim=512)


def gpt_v1_weights(cfg: dict = GPT_V15, seed: int = 1234, suppress_eos: bool = False) -> Dict[str, torch.Tensor]:
    """`UnifiedVoiceV1` tensors: the GPT-2 stack, heads and embeddings of `gpt_weights` without the v2.5-only host tensors."""
    sd = gpt_weights(cfg, seed=seed, suppress_eos=suppress_eos)
    for k in ("lang_embedding.weight", "spk_emb_proj.weight", "spk_emb_proj.bias"):
        sd.pop(k, None)
    return sd


def ecapa_weights(input_size: int = 100, lin_neurons: int = 512, C: int = 512, attention_channels: int = 128, se_channels: int = 128,
                  scale: int = 8, seed: int = 41, prefix: str = "") -> Dict[str, torch.Tensor]:
    """ECAPA-TDNN (indextts/BigVGAN/ECAPA_TDNN.py) under the reference class's parameter names, eval-mode BatchNorm statistics included."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    kernels = (5, 3, 3, 3, 1)

    def conv(p, cout, cin, k):
        sd[prefix + p + "conv.weight"] = torch.randn(cout, cin, k, generator=g) * math.sqrt(2.0 / (cin * k))
        sd[prefix + p + "conv.bias"] = 0.1 * torch.randn(cout, generator=g)

    def bn(p, ch):
        sd[prefix + p + "weight"] = 1 + 0.2 * torch.randn(ch, generator=g)
        sd[prefix + p + "bias"] = 0.1 * torch.randn(ch, generator=g)
        sd[prefix + p + "running_mean"] = 0.2 * torch.randn(ch, generator=g)
        sd[prefix + p + "running_var"] = 0.5 + torch.rand(ch, generator=g)

    def tdnn(p, cin, cout, k):
        conv(p + "conv.", cout, cin, k)
        bn(p + "norm.norm.", cout)

    tdnn("blocks.0.", input_size, C, kernels[0])
    for i in (1, 2, 3):
        p = f"blocks.{i}."
        tdnn(p + "tdnn1.", C, C, 1)
        for j in range(scale - 1):
            tdnn(p + f"res2net_block.blocks.{j}.", C // scale, C // scale, kernels[i])
        tdnn(p + "tdnn2.", C, C, 1)
        conv(p + "se_block.conv1.", se_channels, C, 1)
        conv(p + "se_block.conv2.", C, se_channels, 1)
    tdnn("mfa.", 3 * C, 3 * C, 1)
    tdnn("asp.tdnn.", 9 * C, attention_channels, 1)
    conv("asp.conv.", 3 * C, attention_channels, 1)
    bn("asp_bn.norm.", 6 * C)
    sd[prefix + "fc.conv.weight"] = torch.randn(lin_neurons, 6 * C, 1, generator=g) * math.sqrt(2.0 / (6 * C))
    sd[prefix + "fc.conv.bias"] = 0.1 * torch.randn(lin_neurons, generator=g)
    return sd


def bigvgan_v1_weights(h: dict = BIGVGAN_V1_24K, seed: int = 1234, post_gain: float = 0.2) -> Dict[str, torch.Tensor]:
    """v1 / v1.5 vocoder: `bigvgan_weights` on the GPT latent instead of a mel + the speaker-conditioning 1x1 convs + `speaker_encoder.*`."""
    cond, gpt_dim = h["speaker_embedding_dim"], h["gpt_dim"]
    sd = bigvgan_weights(dict(h, num_mels=gpt_dim), seed=seed, post_gain=post_gain)
    g = torch.Generator().manual_seed(seed + 1)
    c0 = h["upsample_initial_channel"]

    def conv1(name, cout):
        sd[name + ".weight"] = torch.randn(cout, cond, 1, generator=g) * (0.3 / math.sqrt(cond))
        sd[name + ".bias"] = torch.randn(cout, generator=g) * 0.02

    conv1("cond_layer", c0)
    for i in range(len(h["upsample_rates"])):
        conv1(f"conds.{i}", c0 // 2 ** (i + 1))
    sd.update(ecapa_weights(input_size=h["num_mels"], lin_neurons=cond, seed=seed + 2, prefix="speaker_encoder."))
    return sd


# ---- s2mel flow-matching decoder (DiT + WaveNet head) at the shipped IndexTTS-2 / 2.5 widths (Seed-VC style configuration;
# checkpoints/config.yaml is not in the repository, the widths follow backends/trt/export/export_dit_onnx.py:88-96 and the
# module defaults: hidden 512, 13 layers, 8 heads, content 512, style 192, WaveNet 512 x 8, k = 5, dilation rate 1) ----------
S2MEL_V2 = dict(DiT=dict(hidden_dim=512, num_heads=8, depth=13, in_channels=80, content_dim=512, content_codebook_size=1024,
                         style_condition=True, final_layer_type="wavenet", is_causal=False, long_skip_connection=True,
                         uvit_skip_connection=True, time_as_token=False, style_as_token=False),
                wavenet=dict(hidden_dim=512, num_layers=8, kernel_size=5, dilation_rate=1, style_condition=True),
                style_encoder=dict(dim=192))


def s2mel_weights(args: dict = S2MEL_V2, seed: int = 1234) -> Dict[str, torch.Tensor]:
    """Seeded, roughly variance-preserving weights under the reference's `estimator.*` names (weight-norm already folded)."""
    g = torch.Generator().manual_seed(seed)
    d, w = args["DiT"], args["wavenet"]
    H, C, cd, sd_ = d["hidden_dim"], d["in_channels"], d["content_dim"], args["style_encoder"]["dim"]
    W, L, k = w["hidden_dim"], w["num_layers"], w["kernel_size"]
    n = int(2 * (4 * H) / 3)
    I = n if n % 256 == 0 else n + 256 - (n % 256)
    sd: Dict[str, torch.Tensor] = {}

    def lin(name, out, inp, bias=True, scale=1.0, ones=0):
        sd[name + ".weight"] = torch.randn(out, inp, generator=g) * (scale / math.sqrt(inp))
        if bias:
            b = torch.randn(out, generator=g) * 0.02
            if ones:
                b[:ones] += 1.0
            sd[name + ".bias"] = b

    P = "estimator."
    for i in range(d["depth"]):
        Lp = f"{P}transformer.layers.{i}."
        lin(Lp + "attention.wqkv", 3 * H, H, bias=False)
        lin(Lp + "attention.wo", H, H, bias=False, scale=0.5)
        lin(Lp + "feed_forward.w1", I, H, bias=False)
        lin(Lp + "feed_forward.w3", I, H, bias=False)
        lin(Lp + "feed_forward.w2", H, I, bias=False, scale=0.5)
        for nm in ("attention_norm", "ffn_norm"):
            lin(Lp + nm + ".project_layer", 2 * H, H, scale=0.5, ones=H)
            sd[Lp + nm + ".norm.weight"] = 1.0 + 0.1 * torch.randn(H, generator=g)
        if i > d["depth"] // 2:
            lin(Lp + "skip_in_linear", H, 2 * H)
    lin(P + "transformer.norm.project_layer", 2 * H, H, scale=0.5, ones=H)
    sd[P + "transformer.norm.norm.weight"] = 1.0 + 0.1 * torch.randn(H, generator=g)
    lin(P + "cond_projection", H, cd)
    lin(P + "cond_x_merge_linear", H, H + 2 * C + sd_)
    lin(P + "skip_linear", H, H + C)
    for te, wd in (("t_embedder", H), ("t_embedder2", W)):
        sd[P + te + ".freqs"] = torch.exp(-math.log(10000) * torch.arange(128, dtype=torch.float32) / 128)
        lin(P + te + ".mlp.0", wd, 256)
        lin(P + te + ".mlp.2", wd, wd)
    lin(P + "conv1", W, H)
    lin(P + "res_projection", W, H)
    lin(P + "final_layer.linear", W, W)
    lin(P + "final_layer.adaLN_modulation.1", 2 * W, W, scale=0.5)
    sd[P + "conv2.weight"] = torch.randn(C, W, 1, generator=g) / math.sqrt(W)
    sd[P + "conv2.bias"] = torch.randn(C, generator=g) * 0.02
    for i in range(L):
        ro = 2 * W if i < L - 1 else W
        sd[f"{P}wavenet.in_layers.{i}.conv.conv.weight"] = torch.randn(2 * W, W, k, generator=g) / math.sqrt(W * k)
        sd[f"{P}wavenet.in_layers.{i}.conv.conv.bias"] = torch.randn(2 * W, generator=g) * 0.02
        sd[f"{P}wavenet.res_skip_layers.{i}.conv.conv.weight"] = torch.randn(ro, W, 1, generator=g) * (0.5 / math.sqrt(W))
        sd[f"{P}wavenet.res_skip_layers.{i}.conv.conv.bias"] = torch.randn(ro, generator=g) * 0.02
    sd[P + "wavenet.cond_layer.conv.conv.weight"] = torch.randn(2 * W * L, W, 1, generator=g) / math.sqrt(W)
    sd[P + "wavenet.cond_layer.conv.conv.bias"] = torch.randn(2 * W * L, generator=g) * 0.02
    return sd


# ---- semantic codec (decode half) and s2mel length regulator at the reference's constructor defaults
# (indextts/codec/models.py EnhancedCodec: codebook 8192 x 8, hidden 1024, Vocos 384 / 2048 x 12;
#  InterpolateRegulator: channels 512, in_channels 1024, sampling_ratios [1, 1, 1, 1]) ----------------------------------------
CODEC_V2 = dict(codebook_size=8192, hidden_size=1024, codebook_dim=8, vocos_dim=384, vocos_intermediate_dim=2048, vocos_num_layers=12)
REGULATOR_V2 = dict(channels=512, sampling_ratios=(1, 1, 1, 1), is_discrete=False, in_channels=1024, content_codebook_size=1024)


def codec_weights(c: dict = CODEC_V2, seed: int = 1234) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    D, H, I = c["vocos_dim"], c["hidden_size"], c["vocos_intermediate_dim"]
    rn = lambda *s_, fan: torch.randn(*s_, generator=g) / math.sqrt(fan)
    sd = {"up.weight": rn(H, H, 3, fan=3 * H), "up.bias": torch.randn(H, generator=g) * 0.05,
          "decoder.0.embed.weight": rn(D, H, 7, fan=7 * H), "decoder.0.embed.bias": torch.randn(D, generator=g) * 0.05,
          "decoder.0.norm.weight": 1 + 0.1 * torch.randn(D, generator=g), "decoder.0.norm.bias": 0.05 * torch.randn(D, generator=g)}
    for i in range(c["vocos_num_layers"]):
        p = f"decoder.0.convnext.{i}."
        sd[p + "gamma"] = 0.3 + 0.1 * torch.randn(D, generator=g)
        sd[p + "dwconv.weight"] = rn(D, 1, 7, fan=7)
        sd[p + "dwconv.bias"] = 0.05 * torch.randn(D, generator=g)
        sd[p + "norm.weight"] = 1 + 0.1 * torch.randn(D, generator=g)
        sd[p + "norm.bias"] = 0.05 * torch.randn(D, generator=g)
        sd[p + "pwconv1.weight"], sd[p + "pwconv1.bias"] = rn(I, D, fan=D), 0.05 * torch.randn(I, generator=g)
        sd[p + "pwconv2.weight"], sd[p + "pwconv2.bias"] = rn(D, I, fan=I), 0.05 * torch.randn(D, generator=g)
    sd["decoder.0.final_layer_norm.weight"] = 1 + 0.1 * torch.randn(D, generator=g)
    sd["decoder.0.final_layer_norm.bias"] = 0.05 * torch.randn(D, generator=g)
    sd["decoder.1.weight"], sd["decoder.1.bias"] = rn(H, D, fan=D), 0.05 * torch.randn(H, generator=g)
    Q = "quantizer.quantizers.0."
    sd[Q + "codebook.weight"] = torch.randn(c["codebook_size"], c["codebook_dim"], generator=g)
    sd[Q + "out_project.weight"] = rn(H, c["codebook_dim"], 1, fan=c["codebook_dim"])          # weight-norm already folded
    sd[Q + "out_project.bias"] = 0.05 * torch.randn(H, generator=g)
    return sd


def codec_encoder_weights(c: dict = CODEC_V2, seed: int = 4321) -> Dict[str, torch.Tensor]:
    """The `quantize` half of the semantic codec (stride-2 down conv, Vocos encoder, FVQ in_project) under the reference names;
    merge into `codec_weights(...)` to get a state dict `EnhancedCodec.quantize` accepts."""
    g = torch.Generator().manual_seed(seed)
    D, H, I = c["vocos_dim"], c["hidden_size"], c["vocos_intermediate_dim"]
    rn = lambda *s_, fan: torch.randn(*s_, generator=g) / math.sqrt(fan)
    sd = {"down.weight": rn(H, H, 3, fan=3 * H), "down.bias": torch.randn(H, generator=g) * 0.05,
          "encoder.0.embed.weight": rn(D, H, 7, fan=7 * H), "encoder.0.embed.bias": torch.randn(D, generator=g) * 0.05,
          "encoder.0.norm.weight": 1 + 0.1 * torch.randn(D, generator=g), "encoder.0.norm.bias": 0.05 * torch.randn(D, generator=g)}
    for i in range(c["vocos_num_layers"]):
        p = f"encoder.0.convnext.{i}."
        sd[p + "gamma"] = 0.3 + 0.1 * torch.randn(D, generator=g)
        sd[p + "dwconv.weight"] = rn(D, 1, 7, fan=7)
        sd[p + "dwconv.bias"] = 0.05 * torch.randn(D, generator=g)
        sd[p + "norm.weight"] = 1 + 0.1 * torch.randn(D, generator=g)
        sd[p + "norm.bias"] = 0.05 * torch.randn(D, generator=g)
        sd[p + "pwconv1.weight"], sd[p + "pwconv1.bias"] = rn(I, D, fan=D), 0.05 * torch.randn(I, generator=g)
        sd[p + "pwconv2.weight"], sd[p + "pwconv2.bias"] = rn(D, I, fan=I), 0.05 * torch.randn(D, generator=g)
    sd["encoder.0.final_layer_norm.weight"] = 1 + 0.1 * torch.randn(D, generator=g)
    sd["encoder.0.final_layer_norm.bias"] = 0.05 * torch.randn(D, generator=g)
    sd["encoder.1.weight"], sd["encoder.1.bias"] = rn(H, D, fan=D), 0.05 * torch.randn(H, generator=g)
    Q = "quantizer.quantizers.0."
    sd[Q + "in_project.weight"] = rn(c["codebook_dim"], H, 1, fan=H)                            # weight-norm already folded
    sd[Q + "in_project.bias"] = 0.05 * torch.randn(c["codebook_dim"], generator=g)
    return sd


def regulator_weights(c: dict = REGULATOR_V2, seed: int = 1234) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    Cc, n = c["channels"], len(c["sampling_ratios"])
    sd = {"content_in_proj.weight": torch.randn(Cc, c["in_channels"], generator=g) / math.sqrt(c["in_channels"]),
          "content_in_proj.bias": 0.05 * torch.randn(Cc, generator=g)}
    for i in range(n):
        sd[f"model.{3 * i}.weight"] = torch.randn(Cc, Cc, 3, generator=g) / math.sqrt(3 * Cc)
        sd[f"model.{3 * i}.bias"] = 0.05 * torch.randn(Cc, generator=g)
        sd[f"model.{3 * i + 1}.weight"] = 1 + 0.1 * torch.randn(Cc, generator=g)
        sd[f"model.{3 * i + 1}.bias"] = 0.05 * torch.randn(Cc, generator=g)
    sd[f"model.{3 * n}.weight"] = torch.randn(Cc, Cc, 1, generator=g) / math.sqrt(Cc)
    sd[f"model.{3 * n}.bias"] = 0.05 * torch.randn(Cc, generator=g)
    return sd


# ---- IndexTTS-2 (BASELINE.json configs[3]): conditioning encoders, speed embedding, GPT-latent projector --------------------------
# the published IndexTTS-2 config's sections (checkpoints/config.yaml `gpt.condition_module` / `gpt.emo_condition_module`)
COND_V2 = dict(output_size=512, linear_units=2048, attention_heads=8, num_blocks=6, input_layer="conv2d2", perceiver_mult=2)
EMO_COND_V2 = dict(output_size=512, linear_units=1024, attention_heads=4, num_blocks=4, input_layer="conv2d2", perceiver_mult=2)
GPT_V2 = dict(GPT_V25, condition_type="conformer_perceiver", condition_module=COND_V2, emo_condition_module=EMO_COND_V2)


def _conformer_weights(pre: str, cm: dict, g, input_size: int = 1024, cnn_kernel: int = 15) -> Dict[str, torch.Tensor]:
    """ConformerEncoder parameters (indextts/gpt/conformer_encoder.py) by name and shape, fan-in scaled."""
    D, H, U = cm["output_size"], cm["attention_heads"], cm["linear_units"]
    f_out = (input_size - 1) // 2
    w = lambda *shape: torch.randn(*shape, generator=g) / math.sqrt(max(1, int(torch.tensor(shape[1:]).prod())))
    b = lambda n: 0.05 * torch.randn(n, generator=g)
    gain = lambda n: 1.0 + 0.1 * torch.randn(n, generator=g)
    sd = {pre + "embed.conv.0.weight": w(D, 1, 3, 3), pre + "embed.conv.0.bias": b(D), pre + "embed.out.0.weight": 2.0 * w(D, D * f_out),
          pre + "embed.out.0.bias": b(D), pre + "after_norm.weight": gain(D), pre + "after_norm.bias": b(D)}
    shapes = {"self_attn.linear_q": (D, D), "self_attn.linear_k": (D, D), "self_attn.linear_v": (D, D), "self_attn.linear_out": (D, D),
              "feed_forward.w_1": (U, D), "feed_forward.w_2": (D, U), "conv_module.pointwise_conv1": (2 * D, D, 1),
              "conv_module.depthwise_conv": (D, 1, cnn_kernel), "conv_module.pointwise_conv2": (D, D, 1)}
    for i in range(cm["num_blocks"]):
        p = f"{pre}encoders.{i}."
        for name, shape in shapes.items():
            sd[p + name + ".weight"], sd[p + name + ".bias"] = w(*shape), b(shape[0])
        sd[p + "self_attn.linear_pos.weight"] = w(D, D)
        sd[p + "self_attn.pos_bias_u"] = 0.3 * torch.randn(H, D // H, generator=g)
        sd[p + "self_attn.pos_bias_v"] = 0.3 * torch.randn(H, D // H, generator=g)
        for n in ("conv_module.norm", "norm_ff", "norm_mha", "norm_conv", "norm_final"):
            sd[p + n + ".weight"], sd[p + n + ".bias"] = gain(D), b(D)
    return sd


def _perceiver_weights(pre: str, dim: int, dim_context: int, heads: int, ff_mult: float, num_latents: int, g, depth: int = 2,
                       dim_head: int = 64) -> Dict[str, torch.Tensor]:
    """PerceiverResampler parameters (indextts/gpt/perceiver.py) by name and shape."""
    inner, ffi = dim_head * heads, int(dim * ff_mult * 2 / 3)
    w = lambda o, i: torch.randn(o, i, generator=g) / math.sqrt(i)
    sd = {pre + "latents": 0.5 * torch.randn(num_latents, dim, generator=g), pre + "norm.gamma": 1.0 + 0.1 * torch.randn(dim, generator=g)}
    if dim_context != dim:
        sd[pre + "proj_context.weight"], sd[pre + "proj_context.bias"] = w(dim, dim_context), 0.05 * torch.randn(dim, generator=g)
    for i in range(depth):
        p = f"{pre}layers.{i}."
        sd[p + "0.to_q.weight"], sd[p + "0.to_kv.weight"], sd[p + "0.to_out.weight"] = w(inner, dim), w(2 * inner, dim), w(dim, inner)
        sd[p + "1.0.weight"], sd[p + "1.0.bias"] = w(2 * ffi, dim), 0.05 * torch.randn(2 * ffi, generator=g)
        sd[p + "1.2.weight"], sd[p + "1.2.bias"] = w(dim, ffi), 0.05 * torch.randn(dim, generator=g)
    return sd


def cond_weights(cfg: dict = GPT_V2, seed: int = 4242, cond_num: int = 32) -> Dict[str, torch.Tensor]:
    """What an IndexTTS-2 `gpt.pth` holds beside the GPT-2 stack: the speaker / emotion Conformer + Perceiver encoders
    (model_v2.py:358-384), `emovec_layer`, `emo_layer` and the speed embedding."""
    g = torch.Generator().manual_seed(seed)
    D, cm, em = cfg["model_dim"], cfg["condition_module"], cfg["emo_condition_module"]
    sd = {}
    sd.update(_conformer_weights("conditioning_encoder.", cm, g))
    sd.update(_perceiver_weights("perceiver_encoder.", D, cm["output_size"], cm["attention_heads"], cm["perceiver_mult"], cond_num, g))
    sd.update(_conformer_weights("emo_conditioning_encoder.", em, g))
    sd.update(_perceiver_weights("emo_perceiver_encoder.", 1024, em["output_size"], em["attention_heads"], em["perceiver_mult"], 1, g))
    sd["emovec_layer.weight"], sd["emovec_layer.bias"] = torch.randn(D, 1024, generator=g) / 32.0, 0.02 * torch.randn(D, generator=g)
    sd["emo_layer.weight"], sd["emo_layer.bias"] = torch.randn(D, D, generator=g) / math.sqrt(D) * 0.3, 0.02 * torch.randn(D, generator=g)
    sd["speed_emb.weight"] = 0.3 * torch.randn(2, D, generator=g)
    return sd


def gpt_layer_weights(dims=(1280, 256, 128, 1024), seed: int = 77) -> Dict[str, torch.Tensor]:
    """`s2mel.models['gpt_layer']` of IndexTTS-2 (s2mel/modules/commons.py:413): three Linear layers on the GPT latents."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for i in range(len(dims) - 1):
        sd[f"{i}.weight"] = torch.randn(dims[i + 1], dims[i], generator=g) / math.sqrt(dims[i])
        sd[f"{i}.bias"] = 0.02 * torch.randn(dims[i + 1], generator=g)
    return sd
