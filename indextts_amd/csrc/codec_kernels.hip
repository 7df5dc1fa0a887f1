// Token-major f32 kernels for the step between the GPT codes and the flow-matching decoder: `EnhancedCodec.decode` (codebook
// lookup + projection, Vocos ConvNeXt backbone, nearest 2x upsampling + conv) and the s2mel `InterpolateRegulator`
// (nearest interpolation to the mel length, conv / GroupNorm / Mish stack).
//
// Reference arithmetic replaced (paths relative to the reference repo root):
//   FVQ decode_code / vq2emb, weight-normed 1x1 out_project   indextts/codec/amphion_codec/quantize/factorized_vector_quantize.py:99-127
//   ConvNeXtBlock, VocosBackbone                              indextts/codec/kmeans/vocos.py:468-526,719-782
//   EnhancedCodec.decode (interpolate x2 + up conv)           indextts/codec/models.py:205-231
//   EnhancedCodec.quantize: FVQ in_project + nearest code      indextts/codec/models.py:179-199, factorized_vector_quantize.py:52-118
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

// FVQ search of one row per block (factorized_vector_quantize.py:52-118, eval): z_e = W_in h + b (the weight-normed 1x1 in_project,
// H -> cd <= 16), e = z_e / max(|z_e|, 1e-12), index = argmax_k -((|e|^2 - 2 e . c_k) + |c_k|^2) over the L2-normalised codebook rows c_k
// (normalised once at load time, |c_k|^2 passed alongside); the first index wins a tie, as torch.max does.
#define VQ_MAX_CD 16
__global__ __launch_bounds__(256) void vq_search_kernel(const float* __restrict__ h, const float* __restrict__ w_in, const float* __restrict__ b_in,
                                                        const float* __restrict__ cbn, const float* __restrict__ c2, long long* __restrict__ idx,
                                                        int H, int K, int cd) {
    __shared__ float part[4][VQ_MAX_CD];
    __shared__ float ze[VQ_MAX_CD];
    __shared__ float bv[4];
    __shared__ int bi[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* row = h + (size_t)blockIdx.x * H;
    float acc[VQ_MAX_CD];
#pragma unroll
    for (int d = 0; d < VQ_MAX_CD; ++d) acc[d] = 0.f;
    for (int c = tid; c < H; c += 256) {
        const float x = row[c];
#pragma unroll
        for (int d = 0; d < VQ_MAX_CD; ++d)
            if (d < cd) acc[d] = fmaf(w_in[(size_t)d * H + c], x, acc[d]);
    }
#pragma unroll
    for (int d = 0; d < VQ_MAX_CD; ++d) {
        const float s = wave_sum(acc[d]);
        if (lane == 0) part[wv][d] = s;
    }
    __syncthreads();
    if (tid < cd) ze[tid] = ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid])) + b_in[tid];
    __syncthreads();
    float en[VQ_MAX_CD];
    float n2 = 0.f;
#pragma unroll
    for (int d = 0; d < VQ_MAX_CD; ++d) { en[d] = d < cd ? ze[d] : 0.f; n2 = fmaf(en[d], en[d], n2); }
    const float inv = 1.f / fmaxf(sqrtf(n2), 1e-12f);
    float e2 = 0.f;
#pragma unroll
    for (int d = 0; d < VQ_MAX_CD; ++d) { en[d] *= inv; e2 = fmaf(en[d], en[d], e2); }
    float best = -INFINITY;
    int besti = 0x7fffffff;
    for (int k = tid; k < K; k += 256) {
        const float* c = cbn + (size_t)k * cd;
        float dot = 0.f;
#pragma unroll
        for (int d = 0; d < VQ_MAX_CD; ++d)
            if (d < cd) dot = fmaf(en[d], c[d], dot);
        const float v = -((e2 - 2.f * dot) + c2[k]);
        if (v > best) { best = v; besti = k; }                 // k ascends per thread: the first maximum stays
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(besti, o, 64);
        if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
    }
    if (lane == 0) { bv[wv] = best; bi[wv] = besti; }
    __syncthreads();
    if (tid == 0) {
        for (int q = 1; q < 4; ++q)
            if (bv[q] > best || (bv[q] == best && bi[q] < besti)) { best = bv[q]; besti = bi[q]; }
        idx[blockIdx.x] = besti;
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
                                                     const int* __restrict__ seq_T, int n, int C, int k, int pad_left) {
    const size_t total = (size_t)n * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int m = (int)(i / C), c = (int)(i - (size_t)m * C);
        const int t = tok_t[m], T = seq_T[tok_seq[m]];
        float acc = b ? b[c] : 0.f;
        for (int j = 0; j < k; ++j) {
            const int u = t + j - pad_left;
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

extern "C" int itts_vq_search_forward(const float* h, const float* w_in, const float* b_in, const float* cb_norm, const float* cb_sq, int64_t* idx,
                                      int n, int H, int n_codes, int cd, void* stream) {
    if (!h || !w_in || !b_in || !cb_norm || !cb_sq || !idx || n < 0 || H < 1 || n_codes < 1 || cd < 1 || cd > VQ_MAX_CD) {
        itts_set_error("vq_search: bad args (1 <= codebook_dim <= %d)", VQ_MAX_CD);
        return ITTS_ERR_ARG;
    }
    if (n == 0) return ITTS_OK;
    hipLaunchKernelGGL(vq_search_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, h, w_in, b_in, cb_norm, cb_sq, (long long*)idx, H, n_codes, cd);
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
    hipLaunchKernelGGL(dwconv_kernel, dim3(grid_for((size_t)n * C)), dim3(256), 0, (hipStream_t)stream, x, w, b, y, tok_seq, tok_t, seq_T, n, C, k, (k - 1) / 2);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

extern "C" int itts_tok_dwconv_causal_forward(const float* x, const float* w, const float* b, float* y, const int32_t* tok_seq, const int32_t* tok_t,
                                              const int32_t* seq_T, int n, int C, int k, void* stream) {
    if (!x || !w || !y || !tok_seq || !tok_t || !seq_T || C < 1 || k < 1) { itts_set_error("tok_dwconv_causal: bad args"); return ITTS_ERR_ARG; }
    if (n <= 0) return ITTS_OK;
    hipLaunchKernelGGL(dwconv_kernel, dim3(grid_for((size_t)n * C)), dim3(256), 0, (hipStream_t)stream, x, w, b, y, tok_seq, tok_t, seq_T, n, C, k, k - 1);
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

// ---- conditioning encoders (Conformer + Perceiver): generic f32 attention and gated activations ----------
// One wave per (query row, head): lanes take keys j = lane, lane + 64, ... of the query's key range, keep an online-softmax state
// and a Dv-wide accumulator each, and merge at the end.  Q [n_q][H][Dq], K [n_k][H][Dq], V [n_k][H][Dv] (Dq, Dv <= 128, % 4 == 0).
#define ATTN_REL_MAX 512
template <int DV4>       // Dv / 4 rounded up to 16 or 32
__global__ __launch_bounds__(64) void attn_generic_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                          float* __restrict__ out, const int* __restrict__ kstart, const int* __restrict__ klen,
                                                          int H, int Dq, int Dv, float scale, const float* __restrict__ rel,
                                                          const int* __restrict__ qpos, int left, int right) {
    __shared__ float qpe[ATTN_REL_MAX];                           // relative_key: q . distance_embedding[r] for this (query, head)
    const int m = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
    const int ks = kstart[m], kl = klen[m];
    float* o = out + ((size_t)m * H + h) * Dv;
    if (kl <= 0) {                                                // every key masked: the reference's softmax row is filled with 0
        for (int d = lane; d < Dv; d += 64) o[d] = 0.f;
        return;
    }
    const float* qr = q + ((size_t)m * H + h) * Dq;
    int qp = 0;
    if (rel) {
        qp = qpos[m];
        for (int r = lane; r <= left + right; r += 64) {
            const float* er = rel + (size_t)r * Dq;
            float s = 0.f;
            for (int d = 0; d < Dq; d += 4) {
                const f32x4 a = *(const f32x4*)(qr + d), b = *(const f32x4*)(er + d);
                s += (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]);
            }
            qpe[r] = s;
        }
        __syncthreads();
    }
    float m_run = -INFINITY, l_run = 0.f;
    f32x4 acc[DV4];
#pragma unroll
    for (int i = 0; i < DV4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int j = lane; j < kl; j += 64) {
        const float* kr = k + ((size_t)(ks + j) * H + h) * Dq;
        float s = 0.f;
        for (int d = 0; d < Dq; d += 4) {
            const f32x4 a = *(const f32x4*)(qr + d), b = *(const f32x4*)(kr + d);
            s += (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]);
        }
        if (rel) s += qpe[min(max(j - qp, -left), right) + left];
        s *= scale;
        const float nm = fmaxf(m_run, s);
        const float al = expf(m_run - nm), p = expf(s - nm);
        l_run = l_run * al + p;
        const float* vr = v + ((size_t)(ks + j) * H + h) * Dv;
#pragma unroll
        for (int i = 0; i < DV4; ++i) {
            if (4 * i < Dv) {
                const f32x4 vv = *(const f32x4*)(vr + 4 * i);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[i][c] = acc[i][c] * al + p * vv[c];
            }
        }
        m_run = nm;
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float om = __shfl_xor(m_run, off, 64), ol = __shfl_xor(l_run, off, 64);
        const float nm = fmaxf(m_run, om);
        const float sa = (m_run == -INFINITY) ? 0.f : expf(m_run - nm), sb = (om == -INFINITY) ? 0.f : expf(om - nm);
        l_run = l_run * sa + ol * sb;
#pragma unroll
        for (int i = 0; i < DV4; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[i][c] = acc[i][c] * sa + __shfl_xor(acc[i][c], off, 64) * sb;
        m_run = nm;
    }
    if (lane == 0) {
        const float inv = 1.0f / l_run;
#pragma unroll
        for (int i = 0; i < DV4; ++i)
            if (4 * i < Dv) {
                f32x4 r;
#pragma unroll
                for (int c = 0; c < 4; ++c) r[c] = acc[i][c] * inv;
                *(f32x4*)(o + 4 * i) = r;
            }
    }
}

// x [n][2C] -> out [n][C].  mode 0: GLU  a * sigmoid(b) (F.glu, conformer conv module); mode 1: GEGLU  a * gelu_erf(b) (perceiver FF)
__global__ __launch_bounds__(256) void glu_kernel(const float* __restrict__ x, float* __restrict__ out, size_t n, int C, int mode) {
    const size_t total = n * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t m = i / C;
        const int c = (int)(i - m * C);
        const float a = x[m * 2 * C + c], b = x[m * 2 * C + C + c];
        out[i] = mode == 0 ? a * (1.0f / (1.0f + expf(-b))) : a * (0.5f * b * (1.0f + erff(b * 0.70710678118654752440f)));
    }
}

// in place.  mode 0: ReLU, 1: SiLU
__global__ __launch_bounds__(256) void act_kernel(float* __restrict__ x, size_t n, int mode) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = x[i];
        x[i] = mode == 0 ? fmaxf(v, 0.f) : (mode == 1 ? v / (1.0f + expf(-v)) : tanhf(v));
    }
}

// perceiver.py RMSNorm: out = x / max(||x||_2, 1e-12) * scale * gamma, one wave per row, in place
__global__ __launch_bounds__(64) void l2norm_kernel(float* __restrict__ x, const float* __restrict__ gamma, int C, float scale) {
    float* r = x + (size_t)blockIdx.x * C;
    float ss = 0.f;
    for (int c = threadIdx.x; c < C; c += 64) ss += r[c] * r[c];
    ss = wave_sum(ss);
    const float inv = scale / fmaxf(sqrtf(ss), 1e-12f);
    for (int c = threadIdx.x; c < C; c += 64) r[c] = r[c] * inv * gamma[c];
}

// ---- CAMPPlus speaker encoder: eval-mode BatchNorm + ReLU, context pooling, gate, statistics pooling -----
// out[m][c] = act(x[m * ld_x + c] * scale[c] + shift[c])   (BatchNorm in eval mode = a per-channel affine map; x may be the first C
// columns of a wider row: the dense blocks grow their feature matrix in place)
__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ x, int ld_x, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, float* __restrict__ out, size_t n, int C, int relu) {
    const size_t total = n * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t m = i / C;
        const int c = (int)(i - m * C);
        const float v = fmaf(x[m * ld_x + c], scale[c], shift[c]);
        out[i] = relu ? fmaxf(v, 0.f) : v;
    }
}

// CAMLayer context (layers.py): out[t][c] = mean_t'(h[t'][c]) + mean over t's segment of seg_len frames (ceil_mode: the last
// segment averages the frames it has).  One sequence of n rows; one wave per channel.
__global__ __launch_bounds__(64) void ctxpool_kernel(const float* __restrict__ h, float* __restrict__ out, int n, int C, int seg_len) {
    const int c = blockIdx.x, lane = threadIdx.x;
    float tot = 0.f;
    for (int t = lane; t < n; t += 64) tot += h[(size_t)t * C + c];
    tot = wave_sum(tot) / (float)n;
    for (int s0 = 0; s0 < n; s0 += seg_len) {
        const int s1 = s0 + seg_len < n ? s0 + seg_len : n;
        float ss = 0.f;
        for (int t = s0 + lane; t < s1; t += 64) ss += h[(size_t)t * C + c];
        ss = wave_sum(ss) / (float)(s1 - s0);
        for (int t = s0 + lane; t < s1; t += 64) out[(size_t)t * C + c] = tot + ss;
    }
}

// y *= sigmoid(g)
__global__ __launch_bounds__(256) void gate_kernel(float* __restrict__ y, const float* __restrict__ g, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] *= 1.0f / (1.0f + expf(-g[i]));
}

// StatsPool: out[c] = mean over the n rows, out[C + c] = unbiased standard deviation; one wave per channel, two passes
__global__ __launch_bounds__(64) void statspool_kernel(const float* __restrict__ x, float* __restrict__ out, int n, int C) {
    const int c = blockIdx.x, lane = threadIdx.x;
    float s = 0.f;
    for (int t = lane; t < n; t += 64) s += x[(size_t)t * C + c];
    const float mean = wave_sum(s) / (float)n;
    float q = 0.f;
    for (int t = lane; t < n; t += 64) { const float d = x[(size_t)t * C + c] - mean; q += d * d; }
    q = wave_sum(q);
    if (lane == 0) { out[c] = mean; out[C + c] = sqrtf(q / (float)(n - 1)); }
}

// Attentive statistics pooling (ECAPA_TDNN.py AttentiveStatisticsPooling._compute_statistics): per channel c over the n frames, weights
// a_t = softmax_t(logit[t][c]) (logit == NULL: uniform 1 / n -- the global-context statistics): out[c] = sum_t a_t x[t][c],
// out[C + c] = sqrt(max(sum_t a_t (x[t][c] - mean)^2, eps)).  One wave per channel, like statspool_kernel.
__global__ __launch_bounds__(64) void attnstats_kernel(const float* __restrict__ x, const float* __restrict__ logit, float* __restrict__ out, int n,
                                                       int C, float eps) {
    const int c = blockIdx.x, lane = threadIdx.x;
    float mx = 0.f, den = (float)n;
    if (logit) {
        mx = -INFINITY;
        for (int t = lane; t < n; t += 64) mx = fmaxf(mx, logit[(size_t)t * C + c]);
        mx = wave_max(mx);
        float s = 0.f;
        for (int t = lane; t < n; t += 64) s += expf(logit[(size_t)t * C + c] - mx);
        den = wave_sum(s);
    }
    float m = 0.f;
    for (int t = lane; t < n; t += 64) {
        const float a = logit ? expf(logit[(size_t)t * C + c] - mx) : 1.f;
        m = fmaf(a, x[(size_t)t * C + c], m);
    }
    const float mean = wave_sum(m) / den;
    float q = 0.f;
    for (int t = lane; t < n; t += 64) {
        const float a = logit ? expf(logit[(size_t)t * C + c] - mx) : 1.f;
        const float d = x[(size_t)t * C + c] - mean;
        q = fmaf(a * d, d, q);
    }
    q = wave_sum(q) / den;
    if (lane == 0) { out[c] = mean; out[C + c] = sqrtf(fmaxf(q, eps)); }
}

extern "C" int itts_tok_attnstats_forward(const float* x, const float* logit, float* out, int n, int C, float eps, void* stream) {
    if (!x || !out || C < 1 || n < 1) { itts_set_error("tok_attnstats: bad args"); return ITTS_ERR_ARG; }
    hipLaunchKernelGGL(attnstats_kernel, dim3(C), dim3(64), 0, (hipStream_t)stream, x, logit, out, n, C, eps);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

extern "C" int itts_tok_affine_forward(const float* x, int ld_x, const float* scale, const float* shift, float* out, int n, int C, int relu,
                                       void* stream) {
    if (!x || !scale || !shift || !out || C < 1 || ld_x < C) { itts_set_error("tok_affine: bad args"); return ITTS_ERR_ARG; }
    if (n <= 0) return ITTS_OK;
    hipLaunchKernelGGL(affine_act_kernel, dim3(grid_for((size_t)n * C)), dim3(256), 0, (hipStream_t)stream, x, ld_x, scale, shift, out, (size_t)n, C, relu);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

extern "C" int itts_tok_ctxpool_forward(const float* h, float* out, int n, int C, int seg_len, void* stream) {
    if (!h || !out || C < 1 || seg_len < 1) { itts_set_error("tok_ctxpool: bad args"); return ITTS_ERR_ARG; }
    if (n <= 0) return ITTS_OK;
    hipLaunchKernelGGL(ctxpool_kernel, dim3(C), dim3(64), 0, (hipStream_t)stream, h, out, n, C, seg_len);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

extern "C" int itts_tok_gate_forward(float* y, const float* g, size_t n, void* stream) {
    if (!y || !g) { itts_set_error("tok_gate: null"); return ITTS_ERR_ARG; }
    if (n == 0) return ITTS_OK;
    hipLaunchKernelGGL(gate_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, y, g, n);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

extern "C" int itts_tok_statspool_forward(const float* x, float* out, int n, int C, void* stream) {
    if (!x || !out || C < 1 || n < 2) { itts_set_error("tok_statspool: need n >= 2 rows"); return ITTS_ERR_ARG; }
    hipLaunchKernelGGL(statspool_kernel, dim3(C), dim3(64), 0, (hipStream_t)stream, x, out, n, C);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

static int attention_launch(const float* q, const float* k, const float* v, float* out, const int32_t* kstart, const int32_t* klen, int n_q,
                            int heads, int dq, int dv, float scale, const float* rel, const int32_t* qpos, int left, int right, void* stream) {
    if (!q || !k || !v || !out || !kstart || !klen || heads < 1 || dq < 4 || dv < 4 || (dq & 3) || (dv & 3) || dv > 128) {
        itts_set_error("attention_forward: need non-null tensors, dq %% 4 == 0, dv %% 4 == 0, dv <= 128");
        return ITTS_ERR_ARG;
    }
    if (rel && (!qpos || left < 0 || right < 0 || left + right + 1 > ATTN_REL_MAX)) {
        itts_set_error("attention_relkey_forward: need qpos and 0 <= left + right < %d", ATTN_REL_MAX);
        return ITTS_ERR_ARG;
    }
    if (n_q <= 0) return ITTS_OK;
    if (dv <= 64)
        hipLaunchKernelGGL(attn_generic_kernel<16>, dim3(n_q, heads), dim3(64), 0, (hipStream_t)stream, q, k, v, out, kstart, klen, heads, dq, dv, scale,
                           rel, qpos, left, right);
    else
        hipLaunchKernelGGL(attn_generic_kernel<32>, dim3(n_q, heads), dim3(64), 0, (hipStream_t)stream, q, k, v, out, kstart, klen, heads, dq, dv, scale,
                           rel, qpos, left, right);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

extern "C" int itts_attention_forward(const float* q, const float* k, const float* v, float* out, const int32_t* kstart, const int32_t* klen,
                                      int n_q, int heads, int dq, int dv, float scale, void* stream) {
    return attention_launch(q, k, v, out, kstart, klen, n_q, heads, dq, dv, scale, nullptr, nullptr, 0, 0, stream);
}

extern "C" int itts_attention_relkey_forward(const float* q, const float* k, const float* v, float* out, const int32_t* kstart, const int32_t* klen,
                                             const int32_t* qpos, const float* dist_emb, int left, int right, int n_q, int heads, int dq, int dv,
                                             float scale, void* stream) {
    if (!dist_emb) { itts_set_error("attention_relkey_forward: null distance embedding"); return ITTS_ERR_ARG; }
    return attention_launch(q, k, v, out, kstart, klen, n_q, heads, dq, dv, scale, dist_emb, qpos, left, right, stream);
}

extern "C" int itts_tok_glu_forward(const float* x, float* out, int n, int C, int mode, void* stream) {
    if (!x || !out || C < 1 || mode < 0 || mode > 1) { itts_set_error("tok_glu: bad args"); return ITTS_ERR_ARG; }
    if (n <= 0) return ITTS_OK;
    hipLaunchKernelGGL(glu_kernel, dim3(grid_for((size_t)n * C)), dim3(256), 0, (hipStream_t)stream, x, out, (size_t)n, C, mode);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

extern "C" int itts_tok_act_forward(float* x, size_t n, int mode, void* stream) {
    if (!x || mode < 0 || mode > 2) { itts_set_error("tok_act: bad args"); return ITTS_ERR_ARG; }
    if (n == 0) return ITTS_OK;
    hipLaunchKernelGGL(act_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, n, mode);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

extern "C" int itts_tok_l2norm_forward(float* x, const float* gamma, int n, int C, float scale, void* stream) {
    if (!x || !gamma || C < 1) { itts_set_error("tok_l2norm: bad args"); return ITTS_ERR_ARG; }
    if (n <= 0) return ITTS_OK;
    hipLaunchKernelGGL(l2norm_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, x, gamma, C, scale);
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
