/* indextts_hip.h -- C ABI of the MI355X (gfx950) IndexTTS hot-path engine.
 *
 * The reference (index-tts/index-tts) has no FFI for this path: the replaceable seams are Python call sites
 * (SURVEY.md section 8b).  Each entry point below names the reference call it stands in for.  All device
 * pointers are raw HIP device addresses owned by the caller (PyTorch's allocator in the shipped host code);
 * the library never frees or retains them past the call, except the weights it copies at load time.
 * Every function returns 0 on success or an ITTS_ERR_* code; itts_last_error() gives the message.
 * Nothing throws across this boundary.  `stream` is a hipStream_t passed as void*.
 * Devices: a model handle belongs to the HIP device that was current when its *_create ran (the analogue of
 * `module.to(device)` in the reference, indextts/infer_v2_5.py:146,232); every call taking the handle switches to that
 * device for its duration and rejects tensors that live on another one.
 */
#ifndef INDEXTTS_HIP_H
#define INDEXTTS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ITTS_ABI_VERSION 12

int itts_abi_version(void);
const char* itts_last_error(void);
/* number of HIP devices visible, or <0 with the HIP error recorded (the library itself loads without a GPU) */
int itts_device_count(void);

/* ------------------------------------------------------------------------------------------------------------
 * Run-time options.  The library reads NO environment variable; every switch between kernels that tests hold to the same
 * results (most of them bitwise) is a named integer option, process-wide, read by the launchers at launch time.  The defaults
 * are the measured-best paths.  Changing a value retires the decode hipGraphs cached in GPT handles (they bake kernel choices
 * in); options marked "at create" are sampled when a model handle is created.  The reference has no counterpart (its kernels are
 * picked by PyTorch's dispatcher); the table exists for A/B tests and measurement tools.
 *   name              default  range    meaning
 *   decode_fuse_ln       1     0..2    GPT decode: LayerNorm inside the consuming GEMM, 1: at 1-8 rows, 2: up to 16 rows, 0: ln_kernel launches
 *   decode_gemm          1     0..1    bf16 decode GEMMs on the LDS-DMA slab kernel (0: register-path kernel)
 *   decode_rot           1     0..1    per-block rotation of the slab DMA issue order
 *   decode_wnt           0     0..1    non-temporal policy on the decode weight stream
 *   decode_nt            0     0..4    force the n-tiles per block of the 64-row decode GEMM
 *   prefill_gemm         1     0..1    bf16 prefill GEMMs on the LDS-DMA tile kernels (0: register-path kernel)
 *   tile256             -1    -1..2    bf16 tile GEMM: -1 by shape, 0 128x128, 1 256x256, 2 256x128
 *   f32_tile             1     0..1    f32 GEMMs with plain epilogues on the f32-MFMA tile kernel
 *   x3_products          6     6..8    plane products per f32 product of the fp32x3 GEMMs / attention (6 or 8)
 *   x3_sched             1     0..1    fp32x3 GEMM: operand split interleaved with the MFMAs
 *   x3_attn              1     0..1    fp32x3 s2mel: attention products on bf16 planes as well (0: the f32-MFMA flash kernel)
 *   sample_radix        -1    -1..1    top-k threshold: -1 per-kernel default, 0 ballot bisection, 1 radix select
 *   gpt_compact          1     0..1    row compaction of ragged decode batches
 *   attn_waves           0     0..16   waves per block of the KV-cache attention kernel (0: by shape; 4 / 8 / 16)
 *   s2mel_fused          1     0..2    s2mel: fused GEMM epilogues (at create); 1: fused, 0: separate element-wise kernels (2 = 1, round 4's spelling)
 *   fa_qs                0     0..4    bf16 flash attention: query sub-tiles per wave (0: by shape)
 *   f32_attn_scalar      0     0..1    f32 s2mel attention on the one-wave-per-query reference kernel
 *   fa32_qs              2     1..2    f32 flash attention: query sub-tiles per wave
 *   aa_act               2     0..2    anti-aliased activation kernel variant
 *   conv_bm              0     0..128  force the co-tile height of the vocoder conv kernel
 *   h3_kernel            1     0..1    f16x3 vocoder conv: window kernel / two-stage kernel
 *   decode_ln_nt         2     2..4    LayerNorm-fused decode GEMM at 5-16 rows: n-tiles per block (2 or 4)
 *   x3_aplanes           0     0..1    fp32x3 s2mel: adaptive-norm outputs as bf16 planes, wqkv / w1|w3 GEMMs without an operand split
 *   x3_pin               0     0..1    fp32x3 GEMM: 1 = non-default variants on one block per CU (round 4's workaround, diagnostic only)
 *   prefill_attn        -1    -1..1    GPT attention of S > 1 passes: -1 causal MFMA kernel (bf16 mode) / canonical streams (f32 mode), 0 canonical, 1 MFMA
 *   voc_act_planes       1     0..1    vocoder bf16x3 mode: activation writes the x3 conv's operand planes (0: f32 activation + split pass; same bits)
 *   x3_waves             8     4..8    fp32x3 GEMM: waves per 128 x 128 block, 8 (wave tile 32 x 64, weights through LDS) or 4 (64 x 64, round 5's kernel); same bits
 * itts_option_count / _name / _doc / _default enumerate the table (index 0 .. count-1).
 * ---------------------------------------------------------------------------------------------------------- */
int itts_set_option(const char* name, int value);      /* ITTS_ERR_ARG: unknown name or value out of range */
int itts_get_option(const char* name, int* value);
int itts_reset_options(void);                          /* every option back to its default */
int itts_option_count(void);
const char* itts_option_name(int index);
const char* itts_option_doc(int index);
int itts_option_default(int index);

/* ------------------------------------------------------------------------------------------------------------
 * BigVGAN vocoder
 * ---------------------------------------------------------------------------------------------------------- */

/* replaces: anti_alias_activation_cuda.forward(input, up_ftr, down_ftr, alpha, beta)
 *   indextts/s2mel/modules/bigvgan/alias_free_activation/cuda/anti_alias_activation.cpp:19-23 (fwd_cuda,
 *   anti_alias_activation_cuda.cu:212) and the torch Activation1d it shadows (.../torch/act.py:25-30).
 * x, y: [B][C][T] f32; alpha, beta: [C] (log scale iff logscale != 0); up_filter, down_filter: [12].
 * lens: optional [B] int32 row lengths (in units of T / len_mult); rows are bounded at their own length. */
int itts_aa_act_forward(const float* x, float* y, const float* alpha, const float* beta, const float* up_filter,
                        const float* down_filter, int B, int C, int T, const int32_t* lens, int len_mult,
                        int logscale, void* stream);

/* host-side weight packing into MFMA A-fragment order (pure CPU; no device needed).
 * conv1d: w [Cout][Cin][k] -> out[itts_packed_conv_floats(Cout, Cin, k)]
 * convT : w [Cin][Cout][k] (torch ConvTranspose1d layout), stride u, padding (k-u)/2: phase r in [0,u) -> a (k/u)-tap
 *         conv, out as above with k/u taps (requires k a multiple of u and k-u even: 8/4, 4/2 in BigVGAN-v2, also 4/4
 *         as in the IndexTTS-1.5 vocoder config). */
size_t itts_packed_conv_floats(int Cout, int Cin, int k);
int itts_pack_conv1d_weight(const float* w, int Cout, int Cin, int k, float* out);
int itts_pack_convT_weight(const float* w, int Cin, int Cout, int k, int u, int phase, float* out);

/* replaces: torch.nn.Conv1d forward inside AMPBlock1 / conv_pre (bigvgan.py:132-141,362), "same" padding.
 * y = epi(conv(x) + bias + bias_b[b] + res), epi by acc_mode: 0 store, 1 y += v, 2 y = (y + v) / div. */
int itts_conv1d_forward(const float* x, const float* wpk, const float* bias, const float* bias_b, const float* res,
                        float* y, int B, int Cin, int Cout, int T, int k, int dilation, const int32_t* lens,
                        int len_mult, int acc_mode, float div, void* stream);

/* Opt-in second mode of the resblock Conv1d's (same arithmetic contract as itts_conv1d_forward, `same` zero padding, odd k):
 * the f32 operands are split into two f16 parts each (x = xh + 2^-11 xl, 22 significand bits) and the conv runs as three
 * v_mfma_f32_16x16x32_f16 products per fragment pair with exact f32 accumulation (the xl*wl term, 2^-22 relative, is dropped).
 * C_in % 32 == 0.  wp3 = itts_pack_conv1d_h3_weight output on the device; scratch = itts_conv1d_h3_scratch_bytes device bytes.
 * replaces: the same nn.Conv1d calls of AMPBlock1 (bigvgan.py:96-141) as itts_conv1d_forward. */
size_t itts_conv1d_h3_packed_bytes(int Cout, int Cin, int k);
int itts_pack_conv1d_h3_weight(const float* w, int Cout, int Cin, int k, void* out);        /* host -> host */
size_t itts_conv1d_h3_scratch_bytes(int B, int Cin, int T);
int itts_conv1d_h3_forward(const float* x, const void* wp3, const float* bias, const float* res, float* y, int B, int Cin, int Cout,
                           int T, int k, int dilation, const int32_t* lens, int len_mult, int acc_mode, float div, void* scratch,
                           void* stream);

/* Third mode of the resblock Conv1d's, the one that may carry the benchmark: every f32 operand EXACTLY as three bf16 planes (x = xh + xm + xl, 8 + 8 +
 * 8 significand bits), six v_mfma_f32_16x16x32_bf16 plane products per fragment pair (the three dropped ones are <= 2^-24 |x w| each), exact f32
 * accumulation -- the arithmetic of the flow-matching stage's fp32x3 GEMMs; error against an f64 convolution not above the f32-MFMA kernel's.
 * C_in % 32 == 0, odd k >= 3, (k - 1) * dilation <= 64.  wp3 = itts_pack_conv1d_x3_weight output on the device; scratch =
 * itts_conv1d_x3_scratch_bytes device bytes.
 * replaces: the same nn.Conv1d calls of AMPBlock1 (bigvgan.py:96-141) as itts_conv1d_forward. */
size_t itts_conv1d_x3_packed_bytes(int Cout, int Cin, int k);
int itts_pack_conv1d_x3_weight(const float* w, int Cout, int Cin, int k, void* out);        /* host -> host */
size_t itts_conv1d_x3_scratch_bytes(int B, int Cin, int T);
int itts_conv1d_x3_forward(const float* x, const void* wp3, const float* bias, const float* res, float* y, int B, int Cin, int Cout,
                           int T, int k, int dilation, const int32_t* lens, int len_mult, int acc_mode, float div, void* scratch,
                           void* stream);
/* replaces: torch.nn.ConvTranspose1d forward of the upsamplers (bigvgan.py:300-316,366-367); k == 2*u,
 * padding (k-u)/2.  wpk_phases: u packed 2-tap weights back to back (itts_pack_convT_weight, phase 0..u-1). */
int itts_conv_transpose1d_forward(const float* x, const float* wpk_phases, const float* bias, const float* bias_b,
                                  float* y, int B, int Cin, int Cout, int Tin, int k, int u, const int32_t* lens,
                                  int len_mult_in, void* stream);

typedef struct {
    int32_t in_channels;               /* num_mels (v2: 80) or gpt latent dim (v1) */
    int32_t upsample_initial_channel;
    int32_t num_upsamples;
    int32_t upsample_rates[8];
    int32_t upsample_kernel_sizes[8];
    int32_t num_kernels;
    int32_t resblock_kernel_sizes[4];
    int32_t num_dilations;
    int32_t resblock_dilations[4][4];
    int32_t snake_logscale;
    int32_t activation;                /* 0 snakebeta, 1 snake */
    int32_t use_tanh_at_final;         /* v2: 0 (clamp), v1: 1 */
    int32_t use_bias_at_final;
    int32_t cond_dim;                  /* 0 = v2 (no conditioning); >0 = v1 speaker embedding width */
    int32_t cond_in_each_up_layer;
} itts_bigvgan_config;

typedef struct itts_bigvgan itts_bigvgan;

/* replaces: BigVGAN.__init__ + from_pretrained/load_state_dict + remove_weight_norm
 *   (indextts/s2mel/modules/bigvgan/bigvgan.py:266-358,388-492; v1 indextts/BigVGAN/models.py).
 * Tensors are given by their reference state-dict names (weight-norm already folded), host f32 pointers. */
int itts_bigvgan_create(const itts_bigvgan_config* cfg, itts_bigvgan** out);
int itts_bigvgan_device(const itts_bigvgan* h);   /* Conv mode of the generator's resblock convs, to be chosen BEFORE the weights are loaded: 0 = exact f32 MFMA (default; the parity
 * mode), 1 = f16 x 3 split operands (itts_conv1d_h3_forward), 2 = bf16 x 3 plane operands (itts_conv1d_x3_forward; exact operands, the mode
 * the benchmark runs) for the resblocks with >= min_channels channels (0 = default 96);
 * everything else (conv_pre / upsamplers / conv_post / activations) is unchanged. */
int itts_bigvgan_set_conv_mode(itts_bigvgan* h, int mode, int min_channels);
/* f16 x 3 mode only: 1 if a forward since the last call met an activation that is not finite or not below 65504 in magnitude (that
 * forward's output is invalid -- use mode 0), else 0; synchronises the device and clears the flag; < 0 on error. */
int itts_bigvgan_range_check(itts_bigvgan* h);
/* device index the handle is bound to */
int itts_bigvgan_load_tensor(itts_bigvgan* h, const char* name, const float* host_data, const int64_t* shape, int ndim);
int itts_bigvgan_finalize(itts_bigvgan* h);       /* checks every required tensor arrived */
void itts_bigvgan_destroy(itts_bigvgan* h);
size_t itts_bigvgan_workspace_bytes(const itts_bigvgan* h, int B, int T);

/* replaces: BigVGAN.forward(mel) (bigvgan.py:360-386; call site indextts/infer_v2_5.py:850) and the v1
 *   BigVGAN.forward(latent, mel_ref) generator part (indextts/BigVGAN/models.py:216-250; infer.py:647) with the
 *   speaker embedding passed in (spk [B][cond_dim], or NULL for v2).
 * x [B][in_channels][T] f32; lens optional [B] int32 (frames); wav [B][T*prod(upsample_rates)] f32. */
int itts_bigvgan_forward(itts_bigvgan* h, const float* x, const int32_t* lens, const float* spk, float* wav, int B,
                         int T, void* workspace, size_t workspace_bytes, void* stream);

/* replaces: the vocoder stage of the reference's streaming pipeline (backends/trt/pipeline/streaming.py:57-68,140-172:
 *   overlapping chunks re-synthesised and Hann-crossfaded) by an exact overlap-save stream: one utterance, mel frames
 *   [in_channels][n_frames] (row stride ld_mel floats, device) pushed in chunks of at most chunk_frames; each push writes
 *   the samples that became final to wav_out (capacity (chunk_frames + halo_frames) * prod(upsample_rates) floats) and
 *   their count to *n_samples_out.  Output trails the input by halo_frames until the push with is_last != 0.  The
 *   concatenation equals itts_bigvgan_forward over the whole mel when halo_frames covers the receptive field (43 frames
 *   for the 22 kHz v2 generator).  spk: v1 speaker embedding [cond_dim] or NULL. */
typedef struct itts_bigvgan_stream itts_bigvgan_stream;
int itts_bigvgan_stream_open(itts_bigvgan* h, int chunk_frames, int halo_frames, itts_bigvgan_stream** out);
size_t itts_bigvgan_stream_workspace_bytes(const itts_bigvgan_stream* s);
int itts_bigvgan_stream_push(itts_bigvgan_stream* s, const float* mel, int ld_mel, int n_frames, int is_last,
                             const float* spk, float* wav_out, int32_t* n_samples_out, void* workspace,
                             size_t workspace_bytes, void* stream);
void itts_bigvgan_stream_close(itts_bigvgan_stream* s);

/* measurement hooks (no reference counterpart: the reference only has unsynchronised perf_counter stage timers,
 *   indextts/infer_v2_5.py:744-746,871-876).  With profiling enabled every kernel launch of the forward is bracketed
 *   by HIP events on the launch stream; profile_read returns, for the LAST forward and per kernel class
 *   {0: Conv1d MFMA, 1: ConvTranspose1d MFMA, 2: anti-aliased activation, 3: conv_post}, the summed GPU ms, launch
 *   count, algorithmic FLOPs and algorithmic tensor bytes (arrays of 4 doubles). */
int itts_bigvgan_set_profiling(itts_bigvgan* h, int enable);
int itts_bigvgan_profile_read(itts_bigvgan* h, double* ms, double* launches, double* flops, double* bytes);
/* per-launch records of the last forward, launch order: out[4*i..] = {class, ms, flops, bytes}; returns the count */
int itts_bigvgan_profile_records(itts_bigvgan* h, double* out, int max_records);

/* ------------------------------------------------------------------------------------------------------------
 * GPT speech-token decoder
 * ---------------------------------------------------------------------------------------------------------- */
#define ITTS_PREC_F32 0   /* parity mode: f32 weights / KV / MFMA (v_mfma_f32_16x16x4_f32, exact f32) */
#define ITTS_PREC_BF16 1  /* bf16 weights / KV / GEMM inputs, f32 accumulate and residual stream */
#define ITTS_PREC_F32X3 2 /* s2mel and itts_gemm_forward only: f32 activations; GEMMs on the bf16 matrix pipe with every f32 operand carried
                           * exactly as three bf16 planes (x = h + m + l), 8 plane products per f32 product, f32 accumulate.  Weights are
                           * packed for it by itts_pack_gemm_weight(..., precision = 2); attention, norms, gates run their f32 code. */

typedef struct {
    int32_t layers, model_dim, heads;   /* head_dim = model_dim / heads must be 64 */
    int32_t vocab;                      /* number_mel_codes (8194) */
    int32_t n_mel_pos;                  /* rows of mel_pos_embedding (max_mel_tokens + 2 + max_conditioning_inputs) */
    int32_t precision;                  /* ITTS_PREC_* */
    int32_t start_mel_token, stop_mel_token;
    float ln_eps;                       /* 1e-5 (GPT2Config.layer_norm_epsilon / nn.LayerNorm default) */
} itts_gpt_config;

typedef struct {
    int32_t do_sample;                  /* 0: argmax (HF greedy), 1: multinomial after the warpers */
    int32_t num_beams;                  /* 1 for itts_gpt_generate, 2..4 for itts_gpt_generate_beam */
    int32_t top_k;                      /* 1..64 when do_sample */
    int32_t min_tokens_to_keep;         /* 1 (2 under beams) */
    int32_t max_new_tokens;             /* max_generate_length */
    int32_t pos_offset;                 /* 2: HF kv-cache position rule (k-th token at mel position k+1, 1-based k);
                                           1: kv_cache=False rule (positions 0..n-1), model_v2.py:145-161 */
    float top_p, temperature, repetition_penalty, length_penalty;
    float typical_mass;                 /* 0: off; else typical_sampling=True with this typical_mass in (0,1)
                                           (model_v2.py:794-799, indextts/utils/typical_sampling.py:4-30) */
    int32_t reserved;                   /* must be 0 */
    uint64_t seed;                      /* device RNG seed when no uniform stream is supplied */
} itts_gen_params;

typedef struct itts_gpt itts_gpt;

/* host-side packing of a [K][N] (transposed=0, HF Conv1D) or [N][K] (transposed=1, nn.Linear) f32 matrix into
 * MFMA B-fragment order for the given precision (pure CPU). */
size_t itts_packed_gemm_bytes(int K, int N, int precision);
int itts_pack_gemm_weight(const float* w, int K, int N, int transposed, int precision, void* out);

/* replaces: UnifiedVoice.__init__/load_checkpoint/post_init_gpt2_config for the decoder stack
 *   (indextts/gpt/model_v2.py:305-493, indextts/utils/checkpoint.py:22-35).  Tensors by reference state-dict name:
 *   gpt.h.{i}.{ln_1,ln_2}.{weight,bias}, gpt.h.{i}.attn.{c_attn,c_proj}.{weight,bias}, gpt.h.{i}.mlp.{c_fc,c_proj}.*,
 *   gpt.ln_f.*, final_norm.*, mel_head.*, mel_embedding.weight, mel_pos_embedding.emb.weight (host f32 pointers). */
int itts_gpt_create(const itts_gpt_config* cfg, itts_gpt** out);
int itts_gpt_device(const itts_gpt* h);           /* device index the handle is bound to */
int itts_gpt_load_tensor(itts_gpt* h, const char* name, const float* host_data, const int64_t* shape, int ndim);
int itts_gpt_finalize(itts_gpt* h);
void itts_gpt_destroy(itts_gpt* h);
size_t itts_gpt_workspace_bytes(const itts_gpt* h, int nseq, int S, int Tmax);

/* replaces: GPT2InferenceModel.generate(...) as called by UnifiedVoice.inference_speech
 *   (indextts/gpt/model_v2.py:815-820 -> vendored GenerationMixin._sample, transformers_generation_utils.py:3123-3297;
 *    per-step forward model_v2.py:121-198; KV cache transformers_gpt2.py:325-328).
 * prefix_embeds [nseq][S][D] f32 device: rows = [left pad][cond][text] embeddings followed by the start-mel row
 *   (mel_embedding[start]+mel_pos[0]) -- what prepare_gpt_inputs + the prefill branch of forward build.
 * pad_lens [nseq] int32 device (left-pad length per row) or NULL.  penalty_ids: host ints already "in input_ids"
 *   for the repetition penalty (the fake prefix id 1 and start_mel).
 * uniforms: optional device f64 [max_new_tokens][nseq] stream for sampled modes (else seeded device RNG).
 * codes_out [nseq][max_new_tokens] int64 device, pre-filled with stop_mel; *n_steps_out = columns generated before
 *   every row finished (HF returns sequences of that length).  Runs on an internal stream ordered after/before
 *   `stream`; the per-token step is one hipGraph replay when use_graph != 0. */
int itts_gpt_generate(itts_gpt* h, const float* prefix_embeds, const int32_t* pad_lens, int nseq, int S,
                      const itts_gen_params* params, const int32_t* penalty_ids, int n_penalty_ids,
                      const double* uniforms, int64_t* codes_out, int32_t* n_steps_out, void* workspace,
                      size_t workspace_bytes, int use_graph, void* stream);
/* Chunked form of itts_gpt_generate for streaming (replaces: GPTTRTEngine.generate_chunks, backends/trt/runtime/
 *   gpt_trtllm_runtime.py:381-520, whose TRT-LLM session yields tokens from inside its decode loop).  First call: prefix_embeds
 *   non-NULL -- prefill + decode until `step_limit` tokens exist (or every row finished).  Later calls: prefix_embeds NULL --
 *   the decode loop continues from the device state left in the SAME workspace (KV cache, step / position, finished flags,
 *   repetition set) up to the new step_limit; nseq, S, params->max_new_tokens, codes_out, uniforms must be those of the first
 *   call.  *n_steps_out = steps run so far (< step_limit only when every row has finished).  The first call clamps step_limit to
 *   params->max_new_tokens; a later call may run past it (sessions with itts_gpt_admit_rows): each row is bounded by its OWN step -- from its step
 *   max_new_tokens on it emits the stop token and stores nothing. */
int itts_gpt_generate_chunk(itts_gpt* h, const float* prefix_embeds, const int32_t* pad_lens, int nseq, int S,
                            const itts_gen_params* params, const int32_t* penalty_ids, int n_penalty_ids,
                            const double* uniforms, int64_t* codes_out, int32_t step_limit, int32_t* n_steps_out,
                            void* workspace, size_t workspace_bytes, int use_graph, void* stream);
/* Admission of new utterances into a suspended itts_gpt_generate_chunk loop (in-flight batching; design reference: backends/trt/serving/
 *   triton_server.py:96-305, backends/trt/pipeline/pipeline.py:459-548 -- the HF loop itself has no counterpart).  prefix_embeds [n_new][S_new][D]
 *   f32 device: the new prompts at their OWN length (left-padded among themselves, pad_lens [n_new] device: pad positions per row; 1 <= S_new <= the
 *   session's prompt bucket), ending with the start-mel row as for itts_gpt_generate.  slots [n_new] host: the utterances (rows of the first call)
 *   whose cache rows / code rows the new ones take over -- they must have FINISHED.  row_limits_new [n_new] host: the new utterances' token caps,
 *   given exactly when limits are installed (itts_gpt_set_row_limits over the first call's rows; the call writes them into that device array once
 *   every check has passed).  The new rows are prefilled on admit_workspace (itts_gpt_admit_workspace_bytes), keep their keys at their own cache
 *   positions (a per-row shift under the batch's position counter), their code row is refilled with the stop token and their first token lands in
 *   its column 0; the following chunk calls decode them with the rest -- token column, position embedding, uniform / RNG stream and row limit follow
 *   the row's own step; the session's step counter may run past params->max_new_tokens (itts_gpt_generate_chunk), each ROW stops at its own
 *   step max_new_tokens.  params, penalty ids, uniforms, codes_out,
 *   workspace: those of the chunk calls.  The running batch is un-compacted by the call.  An admitted row produces bit for bit the ids it produces
 *   decoded alone, whatever step it joins at (tests/test_gpu_admission.py). */
size_t itts_gpt_admit_workspace_bytes(const itts_gpt* h, int n_new, int S_new);
int itts_gpt_admit_rows(itts_gpt* h, const float* prefix_embeds, const int32_t* pad_lens, const int32_t* slots, int n_new, int S_new,
                        const int32_t* row_limits_new, const itts_gen_params* params, const int32_t* penalty_ids, int n_penalty_ids,
                        const double* uniforms, int64_t* codes_out, void* workspace, size_t workspace_bytes, void* admit_workspace,
                        size_t admit_bytes, void* stream);
/* replaces: the same generate() call in beam mode, num_beams > 1 (the reference default is 3-beam beam-sample,
 *   indextts/infer_v2_5.py:732-740) -> vendored GenerationMixin._beam_search (transformers_generation_utils.py:3325-3609),
 *   BeamSearchScorer.process (3rd-party; mirror indextts/gpt/transformers_beam_search.py:215-305,930-1013) and
 *   _reorder_cache (model_v2.py:200-213, done here as a per-position row map instead of copying the KV cache).
 * prefix_embeds / pad_lens are given per SEQUENCE row (n_utts*num_beams rows, beams of one utterance adjacent, i.e.
 *   repeat_interleave as _expand_inputs_for_generation does); the beams of an utterance must therefore carry identical
 *   prefix rows -- the engine prefills row n*num_beams of each utterance once and all its beams read that prompt K/V
 *   through the row map.  uniforms: optional f64 [max_new][n_utts][2*num_beams].
 * Outputs (device): per step the chosen token and parent row of every sequence row (hist_*: [max_new][n_utts*num_beams]
 *   int32), the running beam scores, the finished-hypothesis records hyps_out [n_utts][4]{f32 score, i32 step, i32 row,
 *   i32 pad}, their count and the per-utterance done flags; BeamSearchScorer.finalize (:320-408) is a host-side walk over
 *   these (indextts_amd/gpt.py).  Max 4 beams. */
size_t itts_gpt_beam_workspace_bytes(const itts_gpt* h, int n_utts, int num_beams, int S, int Tmax);
int itts_gpt_generate_beam(itts_gpt* h, const float* prefix_embeds, const int32_t* pad_lens, int n_utts, int num_beams,
                           int S, const itts_gen_params* params, const int32_t* penalty_ids, int n_penalty_ids,
                           const double* uniforms, int32_t* hist_tok_out, int32_t* hist_par_out, float* beam_scores_out,
                           float* hyps_out, int32_t* n_hyps_out, uint8_t* done_out, int32_t* n_steps_out,
                           void* workspace, size_t workspace_bytes, int use_graph, void* stream);
/* HIP-event timings of the last itts_gpt_generate call on its internal stream */
int itts_gpt_last_timing(const itts_gpt* h, float* prefill_ms, float* decode_ms, int32_t* steps);
/* The instantiated decode-step graph is kept in the handle and reused by later generate calls with the same workspace base,
 * batch rows, prompt-length bucket (multiples of 32), max_new_tokens, generation parameters and codes / uniforms pointers (small
 * LRU).  Counters since create: graphs captured, generate calls that reused one. */
int itts_gpt_graph_stats(const itts_gpt* h, int32_t* captures, int32_t* hits);
/* Ragged batches (continuous batching of the decode loop; design reference: backends/trt/pipeline/pipeline.py:459-548,
 * backends/trt/runtime/gpt_trtllm_runtime.py:381-519 -- the HF loop itself keeps finished rows in the batch, generation_utils.py:3256).
 * Utterances that have emitted their stop token leave the running batch: every 8 tokens the survivors are compacted to the front
 * (cache rows stay in place behind a row map) and the step is replayed from the graph of the smaller batch, in buckets of
 * `granularity` rows (default on, 8).  Results are unchanged: a row's arithmetic does not depend on the batch it runs in. */
int itts_gpt_set_compaction(itts_gpt* h, int enable, int granularity);
/* In-flight batching (with itts_gpt_admit_rows): the following itts_gpt_generate_chunk calls return early -- at one of the finished-flag checks the
 * loop makes every 8 steps anyway -- once at least `finished_rows` utterances of the batch have finished (those finished before the call count),
 * so the caller refills their slots without polling in short chunks.  0 = off.  No counterpart in the HF loop; TRT-LLM's in-flight batcher does
 * this inside its executor (backends/trt/serving/triton_server.py:96-305). */
int itts_gpt_set_chunk_return(itts_gpt* h, int finished_rows);
/* Per-utterance caps on generated tokens for the following generate / generate_chunk calls of exactly n utterances (one batch merges
 * requests carrying their own `max_mel_tokens`, infer_v2_5.py:740): utterance u emits the stop token from token index limits[u] on.
 * limits: DEVICE int32 [n], caller-owned, must outlive those calls; NULL clears. */
int itts_gpt_set_row_limits(itts_gpt* h, const int32_t* limits, int n);
/* Of the last generate call: sum over its decode steps of the rows each step ran (= steps x utterances without compaction), and
 * the number of compactions. */
int itts_gpt_compaction_stats(const itts_gpt* h, int64_t* row_steps, int32_t* compactions);

/* replaces: UnifiedVoice.forward(..., return_latent=True) transformer pass (model_v2.py:596-646, get_logits
 *   :528-554; call site indextts/infer_v2.py:636-651): x [nseq][S][D] -> final_norm(ln_f(blocks(x))) [nseq][S][D]. */
int itts_gpt_forward_latent(itts_gpt* h, const float* x, int nseq, int S, float* out, void* workspace,
                            size_t workspace_bytes, void* stream);

/* diagnostics: resident blocks per CU the HIP runtime predicts for the 128 x 128 tile GEMM kernel of a precision (0 f32, 1 bf16, 2 f32x3) */
int itts_gemm_tile_occupancy(int precision, int32_t* blocks_per_cu);
/* unit-level ops for the parity tests: C[M,N] = A[M,K] * W + bias (A in the precision's activation dtype), LayerNorm */
int itts_gemm_forward(const void* A, const void* Wp, const float* bias, float* out, int M, int N, int K, int precision,
                      int prefill_tiles, int gelu, void* stream);
int itts_layernorm_forward(const float* x, const float* gamma, const float* beta, const float* gamma2,
                           const float* beta2, float* out, int rows, int D, float eps, void* stream);
/* unit-level form of the LayerNorm-fused decode GEMM (what a decode step of 1-4 rows runs instead of a LayerNorm launch + a GEMM launch;
 * replaces GPT2Block's ln_1 -> c_attn and ln_2 -> c_fc pairs, indextts/gpt/transformers_gpt2.py:616-618,652-654, for a single new position):
 *   x' = x + (p0 + p1) + (p2 + p3) + bias_prev  when `partial` ([4][M][K] f32 split-K partials of the previous GEMM) is given, else x' = x;
 *   out [M][N] f32 = bf16(LayerNorm(x'; ln_gamma, ln_beta, eps)) * W (bf16-packed, precision 1) + bias;  x_out [M][K] = x' (required with
 *   `partial`, and a different buffer than x).  M <= 4, K in {256, 512, 1280}.  Bitwise itts_layernorm_forward -> bf16 -> itts_gemm_forward. */
int itts_gemm_ln_forward(const float* x, const float* partial, const float* bias_prev, const float* ln_gamma, const float* ln_beta, float eps,
                         const void* Wp, const float* bias, float* out, float* x_out, int M, int N, int K, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * s2mel flow-matching decoder (DiT estimator + classifier-free-guidance Euler solver)
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t hidden_dim, num_heads, depth;          /* DiT (head_dim must be 64)                                   */
    int32_t in_channels;                           /* mel bands (80)                                              */
    int32_t wavenet_hidden, wavenet_layers, wavenet_kernel, wavenet_dilation_rate;
    int32_t precision;                             /* ITTS_PREC_F32 (what the reference computes, infer_v2_5.py:827-828), _BF16 (GEMM
                                                    * operands / Q K V P in bf16, f32 accumulate) or _F32X3 */
    float norm_eps;
} itts_s2mel_config;

typedef struct itts_s2mel itts_s2mel;

/* replaces: CFM.__init__ / DiT.__init__ + load_checkpoint2 (indextts/s2mel/modules/flow_matching.py:117-135,
 *   diffusion_transformer.py:103-184, commons.py load_checkpoint2; call site indextts/infer_v2_5.py:190-206).
 * Tensors by reference state-dict name under `estimator.` with weight-norm folded into `.weight` (host f32): per layer
 *   transformer.layers.{i}.attention.{wqkv,wo}.weight, feed_forward.{w1,w2,w3}.weight, {attention_norm,ffn_norm}.norm.weight,
 *   skip_in_linear.{weight,bias} (layers past the middle); transformer.norm.norm.weight; cond_x_merge_linear.weight (its mel
 *   columns are used); skip_linear, conv1, res_projection, final_layer.linear, conv2 {weight,bias};
 *   wavenet.{in_layers,res_skip_layers}.{i}.conv.conv.{weight,bias}. */
int itts_s2mel_create(const itts_s2mel_config* cfg, itts_s2mel** out);
int itts_s2mel_device(const itts_s2mel* h);
int itts_s2mel_load_tensor(itts_s2mel* h, const char* name, const float* host_data, const int64_t* shape, int ndim);
int itts_s2mel_finalize(itts_s2mel* h);
void itts_s2mel_destroy(itts_s2mel* h);
size_t itts_s2mel_workspace_bytes(const itts_s2mel* h, int n_tok, int n_seq, int t_max);
/* floats of per-step modulation vectors the host supplies (everything that depends on the timestep only):
 *   per layer [attention_norm (weight|bias) 2H][ffn_norm 2H], transformer.norm 2H, WaveNet cond_layer output L*2W,
 *   final-layer (shift|scale) 2W -- AdaptiveLayerNorm.project_layer / WN.cond_layer / FinalLayer.adaLN_modulation applied
 *   to the timestep embeddings (gpt_fast/model.py:31-37, wavenet.py:150, diffusion_transformer.py:98). */
int itts_s2mel_mods_per_step(const itts_s2mel* h);

/* Token layout of both calls: the sequences (CFG branch major: all utterances of the conditional branch, then the null
 * branch) are packed back to back into [n_tok][channels] matrices.  Device int32 tables: tok_seq / tok_t [n_tok] (sequence and
 * frame of a row), seq_start / seq_T / seq_len [n_seq] (first row, frames processed, valid frames = the reference's x_lens).
 * rope [t_max][32][2] f32 (cos, sin) = precompute_freqs_cis rows (gpt_fast/model.py:336-345).
 * const_in [n_tok][hidden] f32 = cond_x_merge_linear applied to the step-invariant columns (prompt mel, projected content,
 * style) plus its bias (diffusion_transformer.py:205-216).
 *
 * replaces: DiT.forward (diffusion_transformer.py:186-257), one call: x [n_tok][in_channels] f32 -> d_out, same shape. */
int itts_s2mel_estimator(itts_s2mel* h, const float* x, const float* const_in, const float* mods, const float* rope,
                         const int32_t* tok_seq, const int32_t* tok_t, const int32_t* seq_start, const int32_t* seq_T,
                         const int32_t* seq_len, int n_seq, int n_tok, int t_max, float* d_out, void* workspace,
                         size_t workspace_bytes, void* stream);
/* replaces: BASECFM.solve_euler (flow_matching.py:57-115; call site indextts/infer_v2_5.py:841-845).  x_state
 * [n_tok / n_branch][in_channels] f32 in/out (noise in, mel out; prompt frames are held at 0), n_branch = 2 with
 * classifier-free guidance (the estimator sees [cond ; null]) or 1 without; mods [n_steps][mods_per_step] device;
 * t_span [n_steps + 1] HOST floats; prompt_len [n_seq] device. */
int itts_s2mel_solve(itts_s2mel* h, float* x_state, const float* const_in, const float* mods, const float* rope,
                     const int32_t* tok_seq, const int32_t* tok_t, const int32_t* seq_start, const int32_t* seq_T,
                     const int32_t* seq_len, const int32_t* prompt_len, int n_seq, int n_tok, int t_max, int n_branch, int n_steps,
                     const float* t_span, float cfg_rate, void* workspace, size_t workspace_bytes, void* stream);

/* HIP-event timing of the last estimator / solve call per kernel class: [0] GEMMs on the MFMA tile kernels (algorithmic FLOPs
 * reported), [1] attention, [2] the wall span of every estimator call (element-wise time = [2] - [0] - [1]). */
int itts_s2mel_set_profiling(itts_s2mel* h, int enable);
int itts_s2mel_profile_read(itts_s2mel* h, double* ms, double* launches, double* flops);

/* Diagnostics: a 64-bit order-independent checksum of every stage's output buffer of the following estimator / solve calls, in launch order, into
 * dev_u64 (capacity 64-bit words on the handle's device; NULL clears).  Each call clears the words and restarts at entry 0; itts_s2mel_trace_count = entries
 * written by the last call, itts_s2mel_trace_label = what entry i is.  Two runs on the same inputs compared entry by entry name the first stage
 * that is not bit-stable (tools/s2mel_trace.py).  The reference has no counterpart. */
int itts_s2mel_set_trace(itts_s2mel* h, void* dev_u64, int capacity);
int itts_s2mel_trace_count(const itts_s2mel* h);
int itts_s2mel_trace_wanted(const itts_s2mel* h);   /* entries the last call asked for; > trace_count: the trace stopped at its capacity */
const char* itts_s2mel_trace_label(const itts_s2mel* h, int index);
/* ... and a COPY of the output of every traced stage whose label starts with label_prefix, packed into dev_buf in launch order (256-byte aligned;
 * stages that no longer fit are skipped): the checksums name the stage that differed between two runs, the images say which elements and how
 * (tools/s2mel_capture.py).  itts_s2mel_capture_offset = byte offset of trace entry `index` of the last call in dev_buf (-1: not captured). */
int itts_s2mel_set_capture(itts_s2mel* h, void* dev_buf, size_t bytes, const char* label_prefix);
long long itts_s2mel_capture_offset(const itts_s2mel* h, int index, size_t* bytes);

/* Dead-row elimination for the following itts_s2mel_solve calls.  The Euler step never reads the estimator's output at prompt frames
 * (flow_matching.py:107 zeroes them) and everything after the DiT's last attention is row-wise except the WaveNet's few frames of
 * context, so that part can run on the TAIL of every sequence only (frames >= prompt_len - receptive field).  Device tables of the
 * tail layout: per tail row its sequence and its frame relative to the sequence's cut; per sequence the first tail row, the tail
 * frames and the valid tail frames; tail_src [n_tail] = full-layout row of a tail row; tail_base [n_seq]: frame t of sequence s is
 * tail row tail_base[s] + t.  Caller-owned, must outlive the solves; tok_seq = NULL clears.  Results of the solve are bit-identical. */
int itts_s2mel_set_tail(itts_s2mel* h, const int32_t* tok_seq, const int32_t* tok_t, const int32_t* seq_start, const int32_t* seq_T,
                        const int32_t* seq_len, const int32_t* tail_src, const int32_t* tail_base, int n_seq, int n_tail, int t_max);

/* unit-level (parity tests): one layer's RoPE + split + non-causal attention.  replaces: Attention.forward between wqkv and wo
 * (gpt_fast/model.py:262-307): qkv f32 [n_tok][3 * heads * 64] -> out [n_tok][heads * 64] in the precision's activation type. */
size_t itts_s2mel_attention_scratch_bytes(int n_tok, int n_seq, int heads, int t_max, int precision);
int itts_s2mel_attention_forward(const float* qkv, const float* rope, const int32_t* tok_seq, const int32_t* tok_t,
                                 const int32_t* seq_start, const int32_t* seq_T, const int32_t* seq_len, int n_seq, int n_tok,
                                 int t_max, int heads, int precision, void* out, void* scratch, size_t scratch_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * codes -> content features: EnhancedCodec.decode and the s2mel InterpolateRegulator, as token-major f32 ops on packed
 * sequences ([n_tok][C]; int32 tables tok_seq / tok_t per row, start / T per sequence).  The dense layers between them
 * run on itts_gemm_forward (precision 0) and itts_layernorm_forward; indextts_amd/codec.py sequences the calls.
 * ---------------------------------------------------------------------------------------------------------- */
/* replaces: FVQ.vq2emb = codebook lookup + weight-normed 1x1 out_project
 *   (indextts/codec/amphion_codec/quantize/factorized_vector_quantize.py:99-127); codes int64 [n] -> out [n][H] */
int itts_vq_project_forward(const int64_t* codes, const float* codebook, const float* w, const float* bias, float* out, int n,
                            int n_codes, int cd, int H, void* stream);
/* replaces: FVQ.forward up to the indices (eval mode; indextts/codec/amphion_codec/quantize/factorized_vector_quantize.py:52-118, reached
 *   from EnhancedCodec.quantize, indextts/codec/models.py:179-199 <- indextts/infer_v2.py:465): the weight-normed 1x1 in_project
 *   (w_in [cd][H], b_in [cd]), L2 normalisation, nearest row of the L2-normalised codebook cb_norm [n_codes][cd] (cb_sq [n_codes] = its
 *   squared row norms) by the reference's distance expression; h [n][H] -> idx int64 [n] (first index on ties); cd <= 16.
 *   itts_vq_project_forward on idx then gives the quantized features. */
int itts_vq_search_forward(const float* h, const float* w_in, const float* b_in, const float* cb_norm, const float* cb_sq, int64_t* idx,
                           int n, int H, int n_codes, int cd, void* stream);
/* replaces: F.interpolate(mode="nearest") + the im2col of a "same" zero-padded Conv1d that follows it
 *   (indextts/codec/models.py:226-229 `up`; indextts/s2mel/modules/length_regulator.py:121-125; vocos.py:770 `embed`):
 *   col [n_dst][k*C], the GEMM with the [k*C][C_out] matrix of the conv weight finishes the conv. */
int itts_tok_gather_conv_forward(const float* x, float* col, const int32_t* tok_seq, const int32_t* tok_t, const int32_t* src_start,
                                 const int32_t* src_T, const int32_t* dst_T, int n_dst, int C, int k, void* stream);
/* replaces: ConvNeXtBlock.dwconv (depthwise Conv1d k=7, vocos.py:490-491,509); w [C][k] */
int itts_tok_dwconv_forward(const float* x, const float* w, const float* b, float* y, const int32_t* tok_seq, const int32_t* tok_t,
                            const int32_t* seq_T, int n, int C, int k, void* stream);
/* replaces: nn.GELU() (erf form) of ConvNeXtBlock (vocos.py:496,515), in place */
int itts_tok_gelu_forward(float* x, size_t n, void* stream);
/* replaces: x = residual + gamma * x of ConvNeXtBlock (vocos.py:517-521): x += gamma[c] * y */
int itts_tok_scale_residual_forward(float* x, const float* y, const float* gamma, int n, int C, void* stream);
/* replaces: nn.GroupNorm(groups=1, C) + nn.Mish() of the regulator stack (length_regulator.py:52-57), in place */
int itts_tok_groupnorm_mish_forward(float* x, const float* gamma, const float* beta, const int32_t* seq_start, const int32_t* seq_T,
                                    int n_seq, int C, float eps, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * conditioning encoders: the Conformer encoder + Perceiver resampler behind UnifiedVoice.get_conditioning /
 * get_emo_conditioning / get_emovec (indextts/gpt/model_v2.py:556-593,827-838), as f32 unit ops on packed rows; the dense layers run
 * on itts_gemm_forward, LayerNorms on itts_layernorm_forward, the depthwise conv on itts_tok_dwconv_forward;
 * indextts_amd/cond.py sequences the calls.
 * ---------------------------------------------------------------------------------------------------------- */
/* replaces: softmax(Q K^T * scale + key mask) V of RelPositionMultiHeadedAttention.forward (matrix_ac + matrix_bd as ONE product of
 *   [q + u | q + v] with [k | p], indextts/gpt/conformer/attention.py:232-312,85-120) and of perceiver.py Attend.forward.
 *   q [n_q][heads][dq], k [n_k][heads][dq], v [n_k][heads][dv], out [n_q][heads][dv]; query row m attends key rows
 *   kstart[m] .. kstart[m] + klen[m] - 1 (klen 0 -> zeros, as the reference's masked softmax). */
int itts_attention_forward(const float* q, const float* k, const float* v, float* out, const int32_t* kstart, const int32_t* klen,
                           int n_q, int heads, int dq, int dv, float scale, void* stream);
/* replaces: Wav2Vec2BertSelfAttention.forward with position_embeddings_type "relative_key" (transformers modeling_wav2vec2_bert.py, the
 *   `semantic_model` of indextts/infer_v2_5.py:171-176,282-290): scores = (q . k_j + q . dist_emb[clamp(j - qpos, -left, right) + left]) * scale
 *   over the query's key range, softmax, . V.  dist_emb [left + right + 1][dq] (shared by the heads), qpos[m] = the query's index inside
 *   its own key range; the rest as itts_attention_forward. */
int itts_attention_relkey_forward(const float* q, const float* k, const float* v, float* out, const int32_t* kstart, const int32_t* klen,
                                  const int32_t* qpos, const float* dist_emb, int left, int right, int n_q, int heads, int dq, int dv,
                                  float scale, void* stream);
/* replaces: the causal depthwise Conv1d of Wav2Vec2BertConvolutionModule (left pad k - 1, no bias: b may be NULL); layout as
 *   itts_tok_dwconv_forward */
int itts_tok_dwconv_causal_forward(const float* x, const float* w, const float* b, float* y, const int32_t* tok_seq, const int32_t* tok_t,
                                   const int32_t* seq_T, int n, int C, int k, void* stream);
/* replaces: F.glu (mode 0, conformer_encoder.py:147) / GEGLU (mode 1, perceiver.py): x [n][2C] -> out [n][C] */
int itts_tok_glu_forward(const float* x, float* out, int n, int C, int mode, void* stream);
/* replaces: ReLU (mode 0, subsampling.py:150) / SiLU (mode 1, conformer activation) / tanh (mode 2, ECAPA attention), in place */
int itts_tok_act_forward(float* x, size_t n, int mode, void* stream);
/* replaces: perceiver.py RMSNorm (F.normalize(x) * sqrt(dim) * gamma), in place */
int itts_tok_l2norm_forward(float* x, const float* gamma, int n, int C, float scale, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * CAMPPlus speaker encoder (indextts/s2mel/modules/campplus/{DTDNN,layers}.py, eval mode): BatchNorm folded
 * into the adjacent conv where it follows one, else applied by itts_tok_affine_forward; convs are row gathers + itts_gemm_forward;
 * indextts_amd/campplus.py sequences the calls.
 * ---------------------------------------------------------------------------------------------------------- */
/* replaces: get_nonlinear('batchnorm-relu') in eval mode (layers.py): out[m][c] = act(x[m][c] * scale[c] + shift[c]); x rows may be
 *   wider than C (row stride ld_x): the dense blocks' growing feature matrix */
int itts_tok_affine_forward(const float* x, int ld_x, const float* scale, const float* shift, float* out, int n, int C, int relu,
                            void* stream);
/* replaces: CAMLayer context = x.mean(-1) + seg_pooling(x, 100) (layers.py CAMLayer.forward / seg_pooling), one sequence of n rows */
int itts_tok_ctxpool_forward(const float* h, float* out, int n, int C, int seg_len, void* stream);
/* replaces: y * sigmoid(linear2(...)) of CAMLayer.forward, in place on y */
int itts_tok_gate_forward(float* y, const float* g, size_t n, void* stream);
/* replaces: AttentiveStatisticsPooling._compute_statistics (indextts/BigVGAN/ECAPA_TDNN.py:282-338, the speaker encoder inside the v1 / v1.5
 *   vocoder, models.py:191,202): per channel, weights softmax over the n frames of logit [n][C] (NULL: uniform, the global-context statistics);
 *   x [n][C] -> out [2C] = weighted mean | sqrt(max(weighted variance, eps)) */
int itts_tok_attnstats_forward(const float* x, const float* logit, float* out, int n, int C, float eps, void* stream);
/* replaces: StatsPool (mean | unbiased std over time, layers.py statistics_pooling): x [n][C] -> out [2C] */
int itts_tok_statspool_forward(const float* x, float* out, int n, int C, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * prompt-audio front end (the DSP half; once per speaker prompt): resampling and the three framed-spectrum
 * feature kinds of indextts/infer_v2_5.py:626-648; indextts_amd/audio.py builds the filter banks / windows / twiddles and mirrors the
 * reference callables (mel_spectrogram, kaldi.fbank, SeamlessM4TFeatureExtractor, Resample).
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t frame_length;   /* samples per frame (<= n_fft; the frame sits at the start of the zero-padded FFT buffer) */
    int32_t hop;            /* frame shift */
    int32_t n_fft;          /* power of two, 64 .. 2048 */
    int32_t n_mels;
    int32_t pad;            /* > 0: the waveform is reflect-padded by this much on both sides (mel_spectrogram: (n_fft - hop) / 2);
                               0: Kaldi snip_edges framing */
    int32_t remove_dc;      /* subtract the frame mean (Kaldi remove_dc_offset) */
    int32_t power;          /* 2: |X|^2;  1: sqrt(|X|^2 + mag_eps) */
    int32_t take_log;       /* natural log after the floor */
    int32_t layout;         /* 0: out [frames][ld_out >= n_mels];  1: out [n_mels][ld_out >= frames] */
    float preemphasis;      /* y[i] = x[i] - c x[i-1], y[0] = (1 - c) x[0];  0 = none */
    float mag_eps;
    float floor;            /* mel energies are floored here before the log (Kaldi: FLT_EPSILON; mel_spectrogram: 1e-5) */
    float scale;            /* multiplies the samples first (SeamlessM4T: 32768) */
} itts_fbank_config;
/* frames the configuration yields for n_samples (0 when shorter than one frame; < 0 on a bad configuration) */
int itts_fbank_frames(const itts_fbank_config* cfg, int n_samples);
/* replaces: mel_spectrogram (indextts/s2mel/modules/audio.py:43-83: reflect pad + torch.stft + magnitude + mel basis + log clamp),
 *   torchaudio.compliance.kaldi.fbank (indextts/infer_v2_5.py:644-647) and transformers' SeamlessM4TFeatureExtractor._extract_fbank_features
 *   (infer_v2_5.py:174,631), by configuration.  wave [B][n_samples] (row stride wave_stride), window [frame_length],
 *   twiddle [n_fft / 2][2] = (cos, -sin)(2 pi m / n_fft), mel [n_mels][n_fft / 2 + 1], out per `layout` (batch stride out_stride). */
int itts_fbank_forward(const float* wave, int B, int n_samples, int64_t wave_stride, const itts_fbank_config* cfg, const float* window,
                       const float* twiddle, const float* mel, float* out, int ld_out, int64_t out_stride, void* stream);
/* replaces: torchaudio.transforms.Resample.forward (infer_v2_5.py:627-628,642; functional._apply_sinc_resample_kernel): x [B][L_in],
 *   kernel [new_rate][2 width + orig] (rates already divided by their gcd), y [B][L_out], L_out = ceil(new_rate L_in / orig) */
int itts_resample_forward(const float* x, const float* kernel, float* y, int B, int L_in, int64_t x_stride, int L_out, int64_t y_stride,
                          int orig, int new_rate, int width, void* stream);
/* replaces: `feat - feat.mean(dim=0)` (mode 0, infer_v2_5.py:648) and the per-mel-bin normalisation of SeamlessM4TFeatureExtractor
 *   (mode 1: (x - mean) / sqrt(var(ddof) + eps)); x [n][C] -> out [n][ld_out >= C] */
int itts_tok_colnorm_forward(const float* x, float* out, int n, int C, int ld_out, int mode, int ddof, float eps, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* INDEXTTS_HIP_H */
