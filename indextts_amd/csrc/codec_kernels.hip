// Token-major f32 kernels for the step between the GPT codes and the flow-matching decoder: `EnhancedCodec.decode` (codebook
// lookup + projection, Vocos ConvNeXt backbone, nearest 2x upsampling + conv) and the s2mel `InterpolateRegulator`
// (nearest interpolation to the mel length, conv / GroupNorm / Mish stack).
//
// Reference arithmetic replaced (paths relative to the reference repo root):
//   FVQ decode_code / vq2emb, weight-normed 1x1 out_project   indextts/codec/amphion_codec/quantize/factorized_vector_quantize.py:99-127
//   ConvNeXtBlock, VocosBackbone                              indextts/codec/kmeans/vocos.py:468-526,719-782
//   EnhancedCodec.decode (interpolate x2 + up conv)           indextts/codec/models.py:205-231
//   InterpolateRegulator.forward                              indextts/s2mel/modules/length_regulator.py:90-141
//
// The stage is ~40 GFLOP per utterance (the flow-matching decoder after it is ~37 TFLOP), so everything runs in f32: the dense
// layers on the exact-f32 MFMA GEMM of gpt_kernels.hip, the rest here.  Sequences are packed back to back ([n_tok][C]); a
// sequence is addressed through (start, T) tables, so zero padding / interpolation happen at each utterance's own ends.
#include "../../include/indextts_hip.h"
#include "common.h"

static inline unsigned grid_for(size_t total) {
    const size_t b = (total + 255) / 256;
    return (unsigned)(b < (size_t)65536 * 8 ? (b ? b : 1) : (size_t)65536 * 8);
}

// out[m][h] = bias[h] + sum_d W[h][d] * codebook[codes[m]][d]
__global__ __launch_bounds__(256) void vq_project_kernel(const long long* __restrict__ codes, const float* __restrict__ cb,
                                                         const float* __restrict__ W, const float* __restrict__ bias, float* __restrict__ out,
                                                         int n, int n_codes, int cd, int H) {
    const size_t total = (size_t)n * H;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int m = (int)(i / H), h = (int)(i - (size_t)m * H);
        long long c = codes[m];
        c = c < 0 ? 0 : (c >= n_codes ? n_codes - 1 : c);
        const float* e = cb + (size_t)c * cd;
        const float* w = W + (size_t)h * cd;
        float acc = 0.f;
        for (int d = 0; d < cd; ++d) acc = fmaf(w[d], e[d], acc);
        out[i] = acc + bias[h];
    }
}

// im2col of a k-tap conv ("same" zero padding) whose input is the source sequence nearest-interpolated to the destination
// length: col[m][j*C + c] = x[src_start[s] + idx(t + j - (k-1)/2)][c], idx(u) = min(floor(u * T_src / T_dst), T_src - 1)
// (torch nearest: float scale), 0 outside [0, T_dst).  T_src == T_dst gives a plain conv; T_dst == 2 T_src the x2 upsampling.
__global__ __launch_bounds__(256) void gather_conv_kernel(const float* __restrict__ x, float* __restrict__ col, const int* __restrict__ tok_seq,
                                                          const int* __restrict__ tok_t, const int* __restrict__ src_start,
                                                          const int* __restrict__ src_T, const int* __restrict__ dst_T, int n_dst, int C, int k) {
    const int c4n = C >> 2;
    const size_t total = (size_t)n_dst * k * c4n;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % c4n) * 4;
        const size_t mj = i / c4n;
        const int j = (int)(mj % k), m = (int)(mj / k);
        const int s = tok_seq[m], Ts = src_T[s], Td = dst_T[s];
        const int u = tok_t[m] + j - (k - 1) / 2;
        f32x4 v{0.f, 0.f, 0.f, 0.f};
        if (u >= 0 && u < Td) {
            int si = u;
            if (Ts != Td) {
                const float scale = (float)Ts / (float)Td;
                si = (int)floorf((float)u * scale);
                si = si < Ts - 1 ? si : Ts - 1;
            }
            v = *(const f32x4*)(x + (size_t)(src_start[s] + si) * C + c);
        }
        *(f32x4*)(col + ((size_t)m * k + j) * C + c) = v;
    }
}

// depthwise conv, k taps, zero padding at the sequence's own ends: y[m][c] = b[c] + sum_j w[c][j] * x[m + j - (k-1)/2][c]
__global__ __launch_bounds__(256) void dwconv_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                     float* __restrict__ y, const int* __restrict__ tok_seq, const int* __restrict__ tok_t,
                                                     const int* __restrict__ seq_T, int n, int C, int k) {
    const size_t total = (size_t)n * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int m = (int)(i / C), c = (int)(i - (size_t)m * C);
        const int t = tok_t[m], T = seq_T[tok_seq[m]];
        float acc = b[c];
        for (int j = 0; j < k; ++j) {
            const int u = t + j - (k - 1) / 2;
            if (u >= 0 && u < T) acc = fmaf(w[(size_t)c * k + j], x[(size_t)(m + u - t) * C + c], acc);
        }
        y[i] = acc;
    }
}

__global__ __launch_bounds__(256) void gelu_erf_kernel(float* __restrict__ x, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = x[i];
        x[i] = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    }
}

__global__ __launch_bounds__(256) void scale_residual_kernel(float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ gamma,
                                                             size_t n, int C) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) x[i] += gamma[i % C] * y[i];
}

// GroupNorm with ONE group (statistics over all channels and frames of a sequence) + affine + Mish, in place.
// One block per sequence: fixed-order double accumulation (deterministic), then the apply pass.
__global__ __launch_bounds__(1024) void groupnorm1_mish_kernel(float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const int* __restrict__ seq_start, const int* __restrict__ seq_T, int C, float eps) {
    __shared__ double sh_s[16], sh_q[16];
    __shared__ float sh_mean, sh_rstd;
    const int s = blockIdx.x, tid = threadIdx.x;
    const size_t n = (size_t)seq_T[s] * C;
    float* xs = x + (size_t)seq_start[s] * C;
    double sum = 0.0;
    for (size_t i = tid; i < n; i += 1024) sum += (double)xs[i];
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if ((tid & 63) == 0) sh_s[tid >> 6] = sum;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += sh_s[i];
        sh_mean = n ? (float)(t / (double)n) : 0.f;
    }
    __syncthreads();
    const float mean = sh_mean;
    double sq = 0.0;
    for (size_t i = tid; i < n; i += 1024) { const double d = (double)xs[i] - (double)mean; sq += d * d; }
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
    if ((tid & 63) == 0) sh_q[tid >> 6] = sq;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += sh_q[i];
        sh_rstd = n ? (float)(1.0 / sqrt(t / (double)n + (double)eps)) : 0.f;
    }
    __syncthreads();
    const float rstd = sh_rstd;
    for (size_t i = tid; i < n; i += 1024) {
        const int c = (int)(i % C);
        const float v = (xs[i] - mean) * rstd * gamma[c] + beta[c];
        const float sp = v > 20.f ? v : log1pf(expf(v));        // softplus with torch's threshold
        xs[i] = v * tanhf(sp);
    }
}

// ---- C ABI (unit-level ops; the host classes of indextts_amd/codec.py sequence them) ------------------------------------------
extern "C" int itts_vq_project_forward(const int64_t* codes, const float* codebook, const float* w, const float* bias, float* out, int n,
                                       int n_codes, int cd, int H, void* stream) {
    if (!codes || !codebook || !w || !bias || !out || n < 0 || cd < 1 || H < 1) { itts_set_error("vq_project: bad args"); return ITTS_ERR_ARG; }
    if (n == 0) return ITTS_OK;
    hipLaunchKernelGGL(vq_project_kernel, dim3(grid_for((size_t)n * H)), dim3(256), 0, (hipStream_t)stream, (const long long*)codes, codebook, w,
                       bias, out, n, n_codes, cd, H);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

extern "C" int itts_tok_gather_conv_forward(const float* x, float* col, const int32_t* tok_seq, const int32_t* tok_t, const int32_t* src_start,
                                            const int32_t* src_T, const int32_t* dst_T, int n_dst, int C, int k, void* stream) {
    if (!x || !col || !tok_seq || !tok_t || !src_start || !src_T || !dst_T || C % 4 || k < 1 || (k & 1) == 0) {
        itts_set_error("tok_gather_conv: bad args (C %% 4 == 0, odd k)");
        return ITTS_ERR_ARG;
    }
    if (n_dst <= 0) return ITTS_OK;
    hipLaunchKernelGGL(gather_conv_kernel, dim3(grid_for((size_t)n_dst * k * (C >> 2))), dim3(256), 0, (hipStream_t)stream, x, col, tok_seq, tok_t,
                       src_start, src_T, dst_T, n_dst, C, k);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

extern "C" int itts_tok_dwconv_forward(const float* x, const float* w, const float* b, float* y, const int32_t* tok_seq, const int32_t* tok_t,
                                       const int32_t* seq_T, int n, int C, int k, void* stream) {
    if (!x || !w || !b || !y || !tok_seq || !tok_t || !seq_T || k < 1 || (k & 1) == 0) { itts_set_error("tok_dwconv: bad args"); return ITTS_ERR_ARG; }
    if (n <= 0) return ITTS_OK;
    hipLaunchKernelGGL(dwconv_kernel, dim3(grid_for((size_t)n * C)), dim3(256), 0, (hipStream_t)stream, x, w, b, y, tok_seq, tok_t, seq_T, n, C, k);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

extern "C" int itts_tok_gelu_forward(float* x, size_t n, void* stream) {
    if (!x) { itts_set_error("tok_gelu: null"); return ITTS_ERR_ARG; }
    if (n == 0) return ITTS_OK;
    hipLaunchKernelGGL(gelu_erf_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, n);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

extern "C" int itts_tok_scale_residual_forward(float* x, const float* y, const float* gamma, int n, int C, void* stream) {
    if (!x || !y || !gamma || C < 1) { itts_set_error("tok_scale_residual: bad args"); return ITTS_ERR_ARG; }
    if (n <= 0) return ITTS_OK;
    hipLaunchKernelGGL(scale_residual_kernel, dim3(grid_for((size_t)n * C)), dim3(256), 0, (hipStream_t)stream, x, y, gamma, (size_t)n * C, C);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

extern "C" int itts_tok_groupnorm_mish_forward(float* x, const float* gamma, const float* beta, const int32_t* seq_start, const int32_t* seq_T,
                                               int n_seq, int C, float eps, void* stream) {
    if (!x || !gamma || !beta || !seq_start || !seq_T || C < 1) { itts_set_error("tok_groupnorm_mish: bad args"); return ITTS_ERR_ARG; }
    if (n_seq <= 0) return ITTS_OK;
    hipLaunchKernelGGL(groupnorm1_mish_kernel, dim3(n_seq), dim3(1024), 0, (hipStream_t)stream, x, gamma, beta, seq_start, seq_T, C, eps);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}
