/* indextts_hip.h -- C ABI of the MI355X (gfx950) IndexTTS hot-path engine.
 *
 * The reference (index-tts/index-tts) has no FFI for this path: the replaceable seams are Python call sites
 * (SURVEY.md section 8b).  Each entry point below names the reference call it stands in for.  All device
 * pointers are raw HIP device addresses owned by the caller (PyTorch's allocator in the shipped host code);
 * the library never frees or retains them past the call, except the weights it copies at load time.
 * Every function returns 0 on success or an ITTS_ERR_* code; itts_last_error() gives the message.
 * Nothing throws across this boundary.  `stream` is a hipStream_t passed as void*.
 */
#ifndef INDEXTTS_HIP_H
#define INDEXTTS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ITTS_ABI_VERSION 1

int itts_abi_version(void);
const char* itts_last_error(void);
/* number of HIP devices visible, or <0 with the HIP error recorded (the library itself loads without a GPU) */
int itts_device_count(void);

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
 * convT : w [Cin][Cout][k] (torch ConvTranspose1d layout), stride u: phase r in [0,u) -> 2-tap conv, out as above
 *         with k = 2 (requires k == 2*u). */
size_t itts_packed_conv_floats(int Cout, int Cin, int k);
int itts_pack_conv1d_weight(const float* w, int Cout, int Cin, int k, float* out);
int itts_pack_convT_weight(const float* w, int Cin, int Cout, int k, int u, int phase, float* out);

/* replaces: torch.nn.Conv1d forward inside AMPBlock1 / conv_pre (bigvgan.py:132-141,362), "same" padding.
 * y = epi(conv(x) + bias + bias_b[b] + res), epi by acc_mode: 0 store, 1 y += v, 2 y = (y + v) / div. */
int itts_conv1d_forward(const float* x, const float* wpk, const float* bias, const float* bias_b, const float* res,
                        float* y, int B, int Cin, int Cout, int T, int k, int dilation, const int32_t* lens,
                        int len_mult, int acc_mode, float div, void* stream);

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

#ifdef __cplusplus
}
#endif
#endif /* INDEXTTS_HIP_H */
