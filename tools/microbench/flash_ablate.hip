// Ablation microbench of the s2mel flash-attention kernel (rule: ablate before optimising).  Builds against the product source with
// -DFA_ABL=<mask> (pieces of the key-tile loop removed) and times one attention call at the bench shape.
//   mask bits: 1 no per-tile global fetch / LDS staging (every tile re-reads stage 0), 2 no barrier, 4 no exp2, 8 no max / permlane,
//              16 no PV MFMAs, 32 no QK MFMAs
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I indextts_amd/csrc -mllvm -amdgpu-mfma-vgpr-form=1 -DFA_ABL=0 -DQS_=1 \
//         tools/microbench/flash_ablate.hip indextts_amd/csrc/options.hip -o /tmp/fa_0 && /tmp/fa_0 64 2443
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../indextts_amd/csrc/common.h"
void itts_set_error(const char* fmt, ...) { (void)fmt; }
#include "ablate_src/s2mel_kernels_r05_ablation.hip"     // frozen copy of the product source that still carries the PF_ABL / FA_ABL / FA_OPT branches
#ifndef QS_
#define QS_ 1
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
    const int n_seq = argc > 1 ? atoi(argv[1]) : 64, T = argc > 2 ? atoi(argv[2]) : 2443, heads = 8;
    const int t_pad = (T + 63) / 64 * 64 + 64;
    const size_t n_tok = (size_t)n_seq * T;
    std::vector<int> tok_seq(n_tok), tok_t(n_tok), seq_start(n_seq), seq_T(n_seq), seq_len(n_seq);
    for (int s = 0; s < n_seq; ++s) { seq_start[s] = s * T; seq_T[s] = T; seq_len[s] = T; for (int t = 0; t < T; ++t) { tok_seq[(size_t)s * T + t] = s; tok_t[(size_t)s * T + t] = t; } }
    int *d_ts, *d_tt, *d_ss, *d_sT, *d_sl;
    CK(hipMalloc(&d_ts, n_tok * 4)); CK(hipMalloc(&d_tt, n_tok * 4)); CK(hipMalloc(&d_ss, n_seq * 4)); CK(hipMalloc(&d_sT, n_seq * 4)); CK(hipMalloc(&d_sl, n_seq * 4));
    CK(hipMemcpy(d_ts, tok_seq.data(), n_tok * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_tt, tok_t.data(), n_tok * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_ss, seq_start.data(), n_seq * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_sT, seq_T.data(), n_seq * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_sl, seq_len.data(), n_seq * 4, hipMemcpyHostToDevice));
    const size_t qn = n_tok * heads * 64, kn = (size_t)n_seq * heads * t_pad * 64;
    std::vector<unsigned short> hq(qn), hk(kn);
    unsigned x = 12345;
    auto rnd = [&]() { x = x * 1664525u + 1013904223u; const float f = ((x >> 8) & 0xffff) / 65536.0f - 0.5f; unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); };
    for (auto& v : hq) v = rnd();
    for (auto& v : hk) v = rnd();
    unsigned short *q, *k, *v, *o;
    CK(hipMalloc(&q, qn * 2)); CK(hipMalloc(&k, kn * 2)); CK(hipMalloc(&v, kn * 2)); CK(hipMalloc(&o, qn * 2));
    CK(hipMemcpy(q, hq.data(), qn * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(k, hk.data(), kn * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(v, hk.data(), kn * 2, hipMemcpyHostToDevice));
    SeqTab tab{d_ts, d_tt, d_ss, d_sT, d_sl, n_seq, (int)n_tok, T};
    const float scale_log2e = 0.125f * 1.4426950408889634f;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch = [&]() { hipLaunchKernelGGL(flash_attn_bf16_kernel<QS_>, dim3((T + 64 * QS_ - 1) / (64 * QS_), heads, n_seq), dim3(256), 0, 0, q, k, v, o, tab, heads, t_pad, scale_log2e); };
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    const int reps = 10;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = 4.0 * 64 * (double)T * T * heads * n_seq;
    printf("FA_ABL=%d QS=%d n_seq=%d T=%d: %.1f us per call, %.0f TFLOP/s (nominal)\n", FA_ABL, QS_, n_seq, T, ms * 1000 / reps, flops / (ms / reps * 1e-3) / 1e12);
    return 0;
}
