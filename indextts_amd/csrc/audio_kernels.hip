// Prompt-audio front end (the DSP half): what turns the speaker prompt's waveform into the three feature
// matrices the prompt encoders read, once per speaker (indextts/infer_v2_5.py:626-648).
//
// Reference arithmetic replaced (paths relative to the reference repo root / its pinned third-party packages):
//   torchaudio.transforms.Resample(sr, 22050 | 16000)                                        infer_v2_5.py:627-628,642
//   mel_spectrogram: reflect pad, STFT (Hann 1024 / hop 256), magnitude, Slaney mel, log      s2mel/modules/audio.py:43-83
//   torchaudio.compliance.kaldi.fbank(num_mel_bins=80, dither=0) + mean subtraction            infer_v2_5.py:644-648
//   SeamlessM4TFeatureExtractor: Kaldi fbank of the 2^15-scaled waveform, per-bin normalisation  infer_v2_5.py:174,631
//
// One framed-spectrum kernel serves all three feature kinds: a workgroup owns one frame -- load (with the reflect padding or the
// snip-edges framing the caller chose), DC removal, pre-emphasis, window, zero padding to the FFT length in LDS; a radix-2 Stockham
// FFT between two LDS buffers (twiddles from a host-built table, so the kernel's rounding does not depend on device sin / cos); power
// or magnitude spectrum; the mel bank as wave-per-filter dot products with coalesced filter rows; floor + log.  15 s of prompt are
// ~1500 frames of 512 .. 1024 points: the stage is latency-sized (tens of microseconds), HBM traffic = the waveform once + the
// filter bank from L2.  All f32; the float64 CPU reference differs from it only in bins ~80 dB under the frame's peak (tests).
#include "../../include/indextts_hip.h"
#include "common.h"

#define FB_MAX_FFT 2048
#define FB_THREADS 256

struct FbankArgs {
    const float* wave; const float* window; const float* tw; const float* mel; float* out;
    long long wave_stride, out_stride;
    int n_samples, frame_length, hop, n_fft, n_freq, n_mels, pad, remove_dc, power, take_log, layout, ld_out, n_frames;
    float preemph, mag_eps, floor_v, scale;
};

__global__ __launch_bounds__(FB_THREADS) void fbank_kernel(FbankArgs a) {
    __shared__ float2 bufA[FB_MAX_FFT];
    __shared__ float2 bufB[FB_MAX_FFT];
    __shared__ float red[FB_THREADS / 64];
    const int tid = threadIdx.x, t = blockIdx.x, b = blockIdx.y;
    const int N = a.n_fft, FL = a.frame_length;
    const float* w = a.wave + (size_t)b * a.wave_stride;
    float* raw = (float*)bufB;                                   // the frame before windowing lives in the second buffer

    // 1. frame samples; `pad` > 0: the waveform is reflect-padded by `pad` on both sides (single reflection: pad < n_samples)
    float part = 0.f;
    for (int i = tid; i < FL; i += FB_THREADS) {
        int s = t * a.hop - a.pad + i;
        s = s < 0 ? -s : s;
        s = s >= a.n_samples ? 2 * (a.n_samples - 1) - s : s;
        const float v = w[s] * a.scale;
        raw[i] = v;
        part += v;
    }
    float mean = 0.f;
    if (a.remove_dc) {
        part = wave_sum(part);
        if ((tid & 63) == 0) red[tid >> 6] = part;
        __syncthreads();
        mean = (red[0] + red[1] + red[2] + red[3]) / (float)FL;
    } else {
        __syncthreads();
    }
    // 2. DC removal, pre-emphasis (first sample against itself: Kaldi's replicated edge), window, zero padding to N
    for (int i = tid; i < N; i += FB_THREADS) {
        float v = 0.f;
        if (i < FL) {
            const float cur = raw[i] - mean;
            const float prev = raw[i > 0 ? i - 1 : 0] - mean;
            v = (cur - a.preemph * prev) * a.window[i];
        }
        bufA[i] = make_float2(v, 0.f);
    }
    __syncthreads();
    // 3. radix-2 Stockham autosort FFT, N / 2 butterflies per stage; tw[m] = exp(-2 pi i m / N), m < N / 2
    float2* src = bufA;
    float2* dst = bufB;
    const int half = N >> 1;
    for (int Ns = 1; Ns < N; Ns <<= 1) {
        const int tstep = half / Ns;
        for (int j = tid; j < half; j += FB_THREADS) {
            const int k = j & (Ns - 1);
            const float2 wv = *(const float2*)(a.tw + 2 * (size_t)(k * tstep));
            const float2 u0 = src[j], x1 = src[j + half];
            const float2 u1 = make_float2(x1.x * wv.x - x1.y * wv.y, x1.x * wv.y + x1.y * wv.x);
            const int j0 = ((j - k) << 1) + k;
            dst[j0] = make_float2(u0.x + u1.x, u0.y + u1.y);
            dst[j0 + Ns] = make_float2(u0.x - u1.x, u0.y - u1.y);
        }
        __syncthreads();
        float2* sw = src; src = dst; dst = sw;
    }
    // 4. one-sided power / magnitude spectrum into the free buffer
    float* P = (float*)dst;
    for (int f = tid; f < a.n_freq; f += FB_THREADS) {
        const float2 c = src[f];
        const float p = c.x * c.x + c.y * c.y;
        P[f] = a.power == 2 ? p : sqrtf(p + a.mag_eps);
    }
    __syncthreads();
    // 5. mel bank: one wave per filter row, lanes across the frequency bins; floor, log
    const int wv = tid >> 6, lane = tid & 63;
    float* o = a.out + (size_t)b * a.out_stride;
    for (int m = wv; m < a.n_mels; m += FB_THREADS / 64) {
        const float* row = a.mel + (size_t)m * a.n_freq;
        float acc = 0.f;
        for (int f = lane; f < a.n_freq; f += 64) acc = fmaf(row[f], P[f], acc);
        acc = wave_sum(acc);
        if (lane == 0) {
            acc = fmaxf(acc, a.floor_v);
            if (a.take_log) acc = logf(acc);
            if (a.layout == 0) o[(size_t)t * a.ld_out + m] = acc;
            else o[(size_t)m * a.ld_out + t] = acc;
        }
    }
}

// y[b][i * new + p] = sum_j kernel[p][j] * x[b][i * orig - width + j]   (zero outside the row): the strided conv1d of
// torchaudio's _apply_sinc_resample_kernel with the (width, width + orig) zero padding folded into the bounds check
__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ x, const float* __restrict__ kern, float* __restrict__ y,
                                                       int L_in, int L_out, int orig, int nw, int width, int taps, long long xs, long long ys) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= L_out) return;
    const float* xr = x + (size_t)blockIdx.y * xs;
    const int blk = idx / nw, ph = idx - blk * nw;
    const int base = blk * orig - width;
    const float* kr = kern + (size_t)ph * taps;
    float acc = 0.f;
    for (int j = 0; j < taps; ++j) {
        const int s = base + j;
        if (s >= 0 && s < L_in) acc = fmaf(kr[j], xr[s], acc);
    }
    y[(size_t)blockIdx.y * ys + idx] = acc;
}

// per-column statistics over the n rows of x [n][C] (two passes, like numpy / torch): mode 0 out = x - mean; mode 1 out =
// (x - mean) / sqrt(var + eps) with var over (n - ddof)
__global__ __launch_bounds__(256) void colnorm_kernel(const float* __restrict__ x, float* __restrict__ out, int n, int C, int ld_out, int mode,
                                                      int ddof, float eps) {
    __shared__ float red[4];
    __shared__ float bc;
    const int c = blockIdx.x, tid = threadIdx.x;
    float s = 0.f;
    for (int r = tid; r < n; r += 256) s += x[(size_t)r * C + c];
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) bc = (red[0] + red[1] + red[2] + red[3]) / (float)n;
    __syncthreads();
    const float mean = bc;
    float inv = 1.f;
    if (mode == 1) {
        float q = 0.f;
        for (int r = tid; r < n; r += 256) { const float d = x[(size_t)r * C + c] - mean; q = fmaf(d, d, q); }
        q = wave_sum(q);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = q;
        __syncthreads();
        inv = 1.f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)(n - ddof) + eps);
    }
    for (int r = tid; r < n; r += 256) out[(size_t)r * ld_out + c] = (x[(size_t)r * C + c] - mean) * inv;
}

extern "C" int itts_fbank_frames(const itts_fbank_config* cfg, int n_samples) {
    if (!cfg || cfg->frame_length < 1 || cfg->hop < 1 || cfg->pad < 0) return -1;
    const long long padded = (long long)n_samples + 2LL * cfg->pad;
    if (padded < cfg->frame_length) return 0;
    return (int)(1 + (padded - cfg->frame_length) / cfg->hop);
}

extern "C" int itts_fbank_forward(const float* wave, int B, int n_samples, int64_t wave_stride, const itts_fbank_config* cfg,
                                  const float* window, const float* twiddle, const float* mel, float* out, int ld_out, int64_t out_stride,
                                  void* stream) {
    if (!wave || !cfg || !window || !twiddle || !mel || !out || B < 1 || n_samples < 1) {
        itts_set_error("fbank_forward: null tensor or empty input");
        return ITTS_ERR_ARG;
    }
    const int N = cfg->n_fft;
    if (N < 64 || N > FB_MAX_FFT || (N & (N - 1)) || cfg->frame_length < 2 || cfg->frame_length > N || cfg->hop < 1 || cfg->n_mels < 1 ||
        (cfg->power != 1 && cfg->power != 2) || (cfg->layout != 0 && cfg->layout != 1)) {
        itts_set_error("fbank_forward: need a power-of-two n_fft in 64..%d, 2 <= frame_length <= n_fft, hop >= 1, power 1|2, layout 0|1", FB_MAX_FFT);
        return ITTS_ERR_ARG;
    }
    if (cfg->pad < 0 || cfg->pad >= n_samples) {
        itts_set_error("fbank_forward: reflect padding %d needs more than %d samples", cfg->pad, cfg->pad);
        return ITTS_ERR_ARG;
    }
    const int frames = itts_fbank_frames(cfg, n_samples);
    if (frames <= 0) return ITTS_OK;                       // shorter than one frame: nothing to write (kaldi.fbank returns an empty matrix)
    if (ld_out < (cfg->layout == 0 ? cfg->n_mels : frames)) {
        itts_set_error("fbank_forward: ld_out %d too small", ld_out);
        return ITTS_ERR_ARG;
    }
    FbankArgs a;
    a.wave = wave; a.window = window; a.tw = twiddle; a.mel = mel; a.out = out;
    a.wave_stride = wave_stride; a.out_stride = out_stride;
    a.n_samples = n_samples; a.frame_length = cfg->frame_length; a.hop = cfg->hop; a.n_fft = N; a.n_freq = N / 2 + 1;
    a.n_mels = cfg->n_mels; a.pad = cfg->pad; a.remove_dc = cfg->remove_dc; a.power = cfg->power; a.take_log = cfg->take_log;
    a.layout = cfg->layout; a.ld_out = ld_out; a.n_frames = frames;
    a.preemph = cfg->preemphasis; a.mag_eps = cfg->mag_eps; a.floor_v = cfg->floor; a.scale = cfg->scale;
    hipLaunchKernelGGL(fbank_kernel, dim3(frames, B), dim3(FB_THREADS), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

extern "C" int itts_resample_forward(const float* x, const float* kernel, float* y, int B, int L_in, int64_t x_stride, int L_out,
                                     int64_t y_stride, int orig, int new_rate, int width, void* stream) {
    if (!x || !kernel || !y || B < 1 || L_in < 1 || orig < 1 || new_rate < 1 || width < 0) {
        itts_set_error("resample_forward: bad arguments");
        return ITTS_ERR_ARG;
    }
    const long long full = ((long long)L_in + orig - 1) / orig * new_rate + new_rate;       // outputs the padded conv can produce
    if (L_out < 1 || L_out > full) {
        itts_set_error("resample_forward: L_out %d outside 1..%lld", L_out, full);
        return ITTS_ERR_ARG;
    }
    hipLaunchKernelGGL(resample_kernel, dim3((L_out + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, x, kernel, y, L_in, L_out, orig,
                       new_rate, width, 2 * width + orig, (long long)x_stride, (long long)y_stride);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

extern "C" int itts_tok_colnorm_forward(const float* x, float* out, int n, int C, int ld_out, int mode, int ddof, float eps, void* stream) {
    if (!x || !out || C < 1 || n < 1 || ld_out < C || (mode != 0 && mode != 1) || ddof < 0 || (mode == 1 && n - ddof < 1)) {
        itts_set_error("tok_colnorm: need n >= 1 (n > ddof for mode 1), ld_out >= C, mode 0|1");
        return ITTS_ERR_ARG;
    }
    hipLaunchKernelGGL(colnorm_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, x, out, n, C, ld_out, mode, ddof, eps);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}
