// Launcher prototypes shared by the kernel and C-ABI translation units.
#pragma once
#include "common.h"

struct ConvArgs {
    const float* x;
    float* y;
    const float* wpk;
    const float* bias;     // [Cout] or null
    const float* bias_b;   // [B][Cout] or null (v1 speaker conditioning, constant over time)
    const float* res;      // [B][Cout][Tout] or null
    const int* lens;       // [B] or null
    int len_mult_in, len_mult_out;
    int Cin, Cout, Tin, Tout;
    int k, tap_base, tap_step, ostride, ooff, m_extra;
    int acc_mode;          // 0 store, 1 y += v, 2 y = (y + v) / div
    float div;
    int B, n_mt, n_co;     // filled by the launcher: batch rows, time tiles, co tiles (XCD-aware 1-D grid)
};

int launch_aa_act(const float* x, float* y, const float* alpha, const float* beta, const float* fu, const float* fd,
                  int B, int C, int T, const int* lens, int len_mult, int logscale, hipStream_t st);
int launch_conv(const ConvArgs& a, int B, hipStream_t st);
int launch_conv_post(const float* x, float* y, const float* w, const float* bias, int B, int Cin, int T, int k,
                     const int* lens, int len_mult, int use_tanh, hipStream_t st);
int launch_cond_bias(const float* spk, const float* w, const float* bias, float* out, int B, int Cout, int cond_dim,
                     hipStream_t st);

// ---- opt-in f16 x 3 split-operand mode of the resblock convs (bigvgan_h3.hip) ----------------------------------------------
struct ConvH3Args {
    const void* xh; const void* xl;   // [B][T][Cin] f16: f16(x) and f16(2^11 (x - f16(x))), frames >= the row length zeroed
    const void* wp;                   // conv_h3_pack output
    const float* bias;                // [Cout] or null
    const float* res;                 // [B][Cout][T] or null
    float* y;                         // [B][Cout][T]
    const void* zero_row;             // >= 16 zero bytes
    const int* lens; int len_mult;    // row b is min(lens[b] * len_mult, T) frames long (null: T)
    int B, Cin, Cout, T, k, dil;
    int acc_mode; float div;          // as ConvArgs
    int n_mt, n_co;                   // filled by the launcher
};
// bf16 x 3 mode (bigvgan_x3.hip): three bf16 planes per f32 operand, six plane products
struct ConvX3Args {
    const void* xp;                   // three planes [3][B][T][Cin] bf16 (h, m, l: h + m + l == x), frames >= the row length zeroed
    const void* wp;                   // conv_x3_pack output
    const float* bias;                // [Cout] or null
    const float* res;                 // [B][Cout][T] or null
    float* y;                         // [B][Cout][T]
    const void* zero_row;             // >= 16 zero bytes
    const int* lens; int len_mult;    // row b is min(lens[b] * len_mult, T) frames long (null: T)
    int B, Cin, Cout, T, k, dil;
    int acc_mode; float div;          // as ConvArgs
    int n_mt, n_co;                   // filled by the launcher
};
bool conv_x3_supported(int Cin, int Cout, int k, int dil);
size_t conv_x3_packed_bytes(int Cout, int Cin, int k);
int conv_x3_pack(const float* w, int Cout, int Cin, int k, void* out);
int launch_split_tm3(const float* x, void* xp, int B, int C, int T, const int* lens, int len_mult, hipStream_t st);
int launch_aa_act_planes(const float* x, void* xp, const float* alpha, const float* beta, const float* fu, const float* fd,
                         int B, int C, int T, const int* lens, int len_mult, int logscale, hipStream_t st);
int launch_conv_x3(const ConvX3Args& a, hipStream_t st);
size_t conv_h3_packed_bytes(int Cout, int Cin, int k);
int conv_h3_pack(const float* w, int Cout, int Cin, int k, void* out);
int launch_split_tm(const float* x, void* xh, void* xl, int B, int C, int T, const int* lens, int len_mult, int* range_flag, hipStream_t st);
int launch_conv_h3(const ConvH3Args& a, hipStream_t st);
