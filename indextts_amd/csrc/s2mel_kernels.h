// Launcher prototypes for the s2mel flow-matching decoder kernels (s2mel_kernels.hip), shared with capi_s2mel.hip.
//
// Token layout: every sequence (one CFG branch of one utterance) occupies T_s consecutive rows of the packed token matrices
// ([n_tok][channels], row-major); seq_start[s] is its first row.  A sequence has two lengths: T_s frames are processed
// (reflect padding of the WaveNet convs happens at T_s) and the first len_s <= T_s of them are valid (attention key limit,
// WaveNet mask) -- the reference's tensor length and `x_lens` (diffusion_transformer.py:219, wavenet.py:143-166).
#pragma once
#include "common.h"

struct SeqTab {
    const int* tok_seq;     // [n_tok] sequence of a token row
    const int* tok_t;       // [n_tok] frame index inside its sequence
    const int* seq_start;   // [n_seq]
    const int* seq_T;       // [n_seq]
    const int* seq_len;     // [n_seq]
    int n_seq, n_tok, t_max;
};

// AdaptiveLayerNorm of gpt_fast (weight * RMSNorm(x) * g + bias) / FinalLayer norm (LayerNorm without affine, modulated)
// row_map (optional): output row m reads input row row_map[m] (the rows that survive dead-row elimination, capi_s2mel.hip)
int launch_ada_rmsnorm(const float* x, const float* g, const float* wb, void* out, int n_tok, int H, float eps, int prec, hipStream_t st,
                       const int* row_map = nullptr);
int launch_final_ln_mod(const float* wn_out, const float* rp, const float* mod, void* out, const SeqTab& tab, int W, int prec, hipStream_t st);
// f32 [rows][C_in] -> act dtype [rows][C_out >= C_in] (zero padded); src row of output row m is m % src_rows (CFG stacking)
int launch_cast_pad(const float* in, void* out, int rows, int src_rows, int C_in, int C_out, int prec, hipStream_t st, const int* row_map = nullptr);
// RoPE + split of a fused QKV GEMM output: qkv f32 [n_tok][3H] -> Q act [n_tok][H] (rotated), K [n_seq][heads][t_pad][64]
// (rotated), V^T [n_seq][heads][64][t_pad] (bf16 mode) or V [n_seq][heads][t_pad][64] (f32 mode)
int launch_rope_split(const float* qkv, const float* rope, void* q, void* k, void* v, const SeqTab& tab, int heads, int t_pad, int prec, hipStream_t st);
// non-causal attention over the first len_s keys of the query's sequence
int launch_s2mel_attention(const void* q, const void* k, const void* v, void* out, const SeqTab& tab, int heads, int t_pad, int prec, hipStream_t st);
// the fp32x3 mode's attention: q f32, K / V^T as three bf16 planes each (plane stride n_seq * heads * t_pad * 64 elements), out f32;
// plane products per f32 product from option x3_products
int launch_s2mel_attention_x3(const void* q, const void* kp, const void* vp, void* out, const SeqTab& tab, int heads, int t_pad, hipStream_t st);
// f32 [n] -> three bf16 planes [3][n] (h + m + l == x exactly)
int launch_split_planes(const float* in, void* out, size_t n, hipStream_t st);
int launch_ada_rmsnorm_planes(const float* x, const float* g, const float* wb, void* out, size_t plane_stride, int n_tok, int H, float eps, hipStream_t st);
int launch_trace_hash(const void* p, size_t bytes, unsigned long long* out, hipStream_t st);
// SwiGLU combine: in f32 [n][2I] = [w1 x | w3 x] -> act [n][I] = silu(a) * b
int launch_swiglu(const float* in, void* out, int n_tok, int I, int prec, hipStream_t st);
// reflect-padded im2col for the WaveNet dilated convs: x f32 [n_tok][W] -> col act [n_tok][k*W], col[m][j*W + c] = x[src(m, j)][c]
int launch_im2col_reflect(const float* x, void* col, const SeqTab& tab, int W, int k, int dilation, int prec, hipStream_t st);
// WaveNet gate: in f32 [n][2W] (+ g [2W]) -> act [n][W] = tanh(a + ga) * sigmoid(b + gb)
int launch_wn_gate(const float* in, const float* g, void* out, int n_tok, int W, int prec, hipStream_t st);
// WaveNet residual/skip update: rs f32 [n][2W] (last layer: [n][W]); x = (x + rs[:, :W]) * mask; out (=|+=) rs[:, W:] (last: rs)
int launch_wn_update(const float* rs, float* x, float* out, const SeqTab& tab, int W, int first, int last, hipStream_t st);
// CFG combine + Euler step on the solver state xs [n_tok / n_branch][C]: xs += dt * ((1 + r) * d_cond - r * d_null); prompt frames 0
int launch_euler_update(float* xs, const float* d, const SeqTab& tab, const int* prompt_len, int C, int n_branch, float dt, float cfg_rate, hipStream_t st,
                        const int* tail_base = nullptr, int tail_half = 0);
