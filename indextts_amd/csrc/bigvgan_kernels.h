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
