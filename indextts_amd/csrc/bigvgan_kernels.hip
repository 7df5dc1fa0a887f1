// BigVGAN vocoder kernels for gfx950 (MI355X).  fp32 in / fp32 out, exact-f32 MFMA for the contractions.
//
// Reference arithmetic being replaced (paths relative to the reference repo root):
//   Activation1d  up(12-tap, x2) -> SnakeBeta -> down(12-tap, /2)
//       indextts/s2mel/modules/bigvgan/alias_free_activation/torch/{act.py:25-30,resample.py:29-58,filter.py:93-101}
//       fused CUDA form: .../alias_free_activation/cuda/anti_alias_activation_cuda.cu:43-179
//   Conv1d / ConvTranspose1d of the generator: indextts/s2mel/modules/bigvgan/bigvgan.py:132-141,300-316,360-386
//
// Design notes (MI355X-first):
//   * A Conv1d with C_in*k = 2k..17k terms per output IS a dense contraction: it runs as an implicit GEMM on
//     v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: bitwise an fmaf chain, so the 1e-4 RMS gate holds; the
//     f32 MFMA rate equals the f32 VALU peak, 157 TF, but needs 1 VGPR per operand and no LDS bandwidth).
//     M = C_out (weights, A operand, pre-packed in fragment order -> one coalesced 1 KiB float4 load per wave
//     feeds 4 K-steps), N = time (B operand read from an LDS tile [ci][t] with the dilation halo), K = (tap, ci).
//   * ConvTranspose1d(stride u, k = 2u) is u phase convolutions with 2 taps each over the same kernel.
//   * Every kernel takes per-row lengths so a ragged batch gives the B=1 result for each row: zero padding for
//     convs, replicate padding for the activation at the row's OWN end (ragged batches).
#include "bigvgan_kernels.h"
#include <stdlib.h>

#define AA_TILE 1024

// --------------------------------------------------------------------------------------------------------------
// Anti-aliased SnakeBeta activation:  y = down2(snake(up2(x)))
//   u[2q]   = 2*sum_j fu[1+2j]*x[clamp(q+2-j)],  u[2q+1] = 2*sum_j fu[2j]*x[clamp(q+3-j)]      (j = 0..5)
//   v[i]    = u[i] + 1/(exp(beta)+1e-9) * sin(u[i]*exp(alpha))^2
//   y[t]    = sum_{j<12} fd[j] * v[clamp(2t+j-5, 0, 2T-1)]
// grid (ceil(T/AA_TILE), C, B), 256 threads.  LDS: x tile (+6 halo each side), v tile (2x rate, +6 halo).
// --------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void aa_act_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                     const float* __restrict__ alpha, const float* __restrict__ beta,
                                                     const float* __restrict__ fu, const float* __restrict__ fd,
                                                     int C, int T, const int* __restrict__ lens, int len_mult,
                                                     int logscale) {
    __shared__ float xs[AA_TILE + 16];
    __shared__ float vs[2 * AA_TILE + 16];
    __shared__ float fus[12], fds[12];
    const int b = blockIdx.z, c = blockIdx.y;
    const int t0 = blockIdx.x * AA_TILE;
    const int len = lens ? min(lens[b] * len_mult, T) : T;
    if (t0 >= len) return;
    const int tid = threadIdx.x;
    if (tid < 12) { fus[tid] = fu[tid]; fds[tid] = fd[tid]; }
    const float* xr = x + ((size_t)b * C + c) * T;
    float* yr = y + ((size_t)b * C + c) * T;
    float a_e = alpha[c], b_e = beta[c];
    if (logscale) { a_e = expf(a_e); b_e = expf(b_e); }
    const float inv_b = 1.0f / (b_e + 1e-9f);
    const int n_out = min(AA_TILE, len - t0);
    // x window: x[clamp(t0-6+cidx)], cidx in [0, n_out+12)
    for (int i = tid; i < n_out + 12; i += 256) {
        int t = t0 - 6 + i;
        t = t < 0 ? 0 : (t > len - 1 ? len - 1 : t);
        xs[i] = xr[t];
    }
    __syncthreads();
    // v window: i = 2*t0-6+vi, vi in [0, 2*n_out+12)
    const int two_len_m1 = 2 * len - 1;
    for (int vi = tid; vi < 2 * n_out + 12; vi += 256) {
        int i = 2 * t0 - 6 + vi;
        i = i < 0 ? 0 : (i > two_len_m1 ? two_len_m1 : i);
        const int q = i >> 1;
        const int base = q - (t0 - 6);          // xs index of x[q]
        float u = 0.f;
        if (i & 1) {
#pragma unroll
            for (int j = 0; j < 6; ++j) u = fmaf(fus[2 * j], xs[base + 3 - j], u);
        } else {
#pragma unroll
            for (int j = 0; j < 6; ++j) u = fmaf(fus[1 + 2 * j], xs[base + 2 - j], u);
        }
        u *= 2.0f;
        const float s = sinf(u * a_e);
        vs[vi] = u + inv_b * s * s;
    }
    __syncthreads();
    for (int i = tid; i < n_out; i += 256) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 12; ++j) acc = fmaf(fds[j], vs[2 * i + j + 1], acc);
        yr[t0 + i] = acc;
    }
}

// --------------------------------------------------------------------------------------------------------------
// Register-tiled version of the same activation (the one launched): each thread owns 4 outputs = 8 consecutive
// 2x-rate samples; it reads its 10 input samples from LDS once (instead of 6 per 2x-rate sample), keeps the 8
// snake values in registers, shares them through LDS, and reads only the 11 neighbour values it does not own.
// LDS traffic per output drops from ~27 to ~7 dwords.  sin: two-term Cody-Waite reduction by 2*pi followed by the
// hardware v_sin_f32 (FAST_SIN, absolute error vs sinf measured in tests/test_gpu_bigvgan.py) or libm sinf.
// --------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sin_reduced(float x) {
    const float inv2pi = 0.15915494309189535f;
    const float k = rintf(x * inv2pi);
    float r = fmaf(-k, 6.2831855f, x);            // 2*pi rounded to f32
    r = fmaf(-k, -1.7484555e-7f, r);              // 2*pi - f32(2*pi)
    return __builtin_amdgcn_sinf(r * inv2pi);     // v_sin_f32: sin(2*pi*arg), arg in [-0.5, 0.5]
}

// LDS image of the 2x-rate window: a thread touches 16-byte chunks 2*tid + k, so within one ds_*_b128 lane group (16 lanes, e.g.
// {0-3, 12-15, 20-27}) chunks 16 apart meet on the same banks (2-way conflicts on every access; PMC: 78 % of the kernel's LDS
// cycles were bank conflicts, LDS array busy 74 % -- profiles/r02p).  Flipping the low chunk bit in odd groups of 16 chunks makes
// the stride-2-chunk pattern conflict-free.  i = dword index of a 16-byte aligned chunk or of a single element.
__device__ __forceinline__ int aa_sw(int i) { return i ^ (((i >> 6) & 1) << 2); }

template <int OPT, bool FAST_SIN>     // OPT = outputs per thread (4 or 8); tile = 256 * OPT outputs per block
__global__ __launch_bounds__(256) void aa_act_kernel_v2(const float* __restrict__ x, float* __restrict__ y,
                                                        const float* __restrict__ alpha, const float* __restrict__ beta,
                                                        const float* __restrict__ fu, const float* __restrict__ fd,
                                                        int C, int T, const int* __restrict__ lens, int len_mult,
                                                        int logscale) {
    constexpr int TILE = 256 * OPT;
    __shared__ __attribute__((aligned(16))) float xs[TILE + 16];
    __shared__ __attribute__((aligned(16))) float vs[2 * TILE + 64];
    const int b = blockIdx.z, c = blockIdx.y;
    const int t0 = blockIdx.x * TILE;
    const int len = lens ? min(lens[b] * len_mult, T) : T;
    if (t0 >= len) return;
    const int tid = threadIdx.x;
    const float* xr = x + ((size_t)b * C + c) * T;
    float* yr = y + ((size_t)b * C + c) * T;
    float fus[12], fds[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) { fus[j] = 2.0f * fu[j]; fds[j] = fd[j]; }     // fold the x2 up-sampling gain
    float a_e = alpha[c], b_e = beta[c];
    if (logscale) { a_e = expf(a_e); b_e = expf(b_e); }
    const float inv_b = 1.0f / (b_e + 1e-9f);
    const int n_out = min(TILE, len - t0);
    // x window: xs[i] = x[clamp(t0 - 6 + i)], i in [0, TILE + 12): unconditional clamped loads, all in flight at once
    float xl[OPT + 1];
#pragma unroll
    for (int k = 0; k < OPT + 1; ++k) {
        int t = t0 - 6 + tid + 256 * k;
        t = t < 0 ? 0 : (t > len - 1 ? len - 1 : t);
        xl[k] = xr[t];
    }
#pragma unroll
    for (int k = 0; k < OPT + 1; ++k)
        if (tid + 256 * k < TILE + 12) xs[tid + 256 * k] = xl[k];
    __syncthreads();
    // 2x-rate window: vs[vi] <-> i = 2*t0 - 6 + vi.  Thread owns vi = 2*OPT*tid .. +2*OPT-1; threads 0..1 also the tail.
    auto snake8 = [&](int vb, float* v) {
        float xw[12];
        *(f32x4*)&xw[0] = *(const f32x4*)&xs[(vb >> 1)];                  // (vb >> 1) is a multiple of 4: 16-byte chunks, lane stride 1 chunk
        *(f32x4*)&xw[4] = *(const f32x4*)&xs[(vb >> 1) + 4];
        *(f32x4*)&xw[8] = *(const f32x4*)&xs[(vb >> 1) + 8];
#pragma unroll
        for (int p = 0; p < 4; ++p) {            // q = q0 + p ; xw[p + 3] = x[q]
            float ue = 0.f, uo = 0.f;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                ue = fmaf(fus[1 + 2 * j], xw[p + 5 - j], ue);     // x[q + 2 - j]
                uo = fmaf(fus[2 * j], xw[p + 6 - j], uo);         // x[q + 3 - j]
            }
            const float se = FAST_SIN ? sin_reduced(ue * a_e) : sinf(ue * a_e);
            const float so = FAST_SIN ? sin_reduced(uo * a_e) : sinf(uo * a_e);
            v[2 * p] = fmaf(inv_b * se, se, ue);
            v[2 * p + 1] = fmaf(inv_b * so, so, uo);
        }
    };
#pragma unroll
    for (int g = 0; g < OPT / 4; ++g) {
        float v[8];
        const int vb = 2 * OPT * tid + 8 * g;
        snake8(vb, v);
        *(f32x4*)&vs[aa_sw(vb)] = *(const f32x4*)&v[0];
        *(f32x4*)&vs[aa_sw(vb + 4)] = *(const f32x4*)&v[4];
    }
    if (tid < 2) {                               // tail: vi = 2*TILE .. 2*TILE+15 (only the first 12 are used)
        float vt[8];
        snake8(2 * TILE + 8 * tid, vt);
        *(f32x4*)&vs[aa_sw(2 * TILE + 8 * tid)] = *(const f32x4*)&vt[0];
        *(f32x4*)&vs[aa_sw(2 * TILE + 8 * tid + 4)] = *(const f32x4*)&vt[4];
    }
    __syncthreads();
    // replicate padding at the 2x rate (5 left / 6 right): entries outside [0, 2*len-1] take the edge value
    if (t0 == 0 && tid < 6) vs[aa_sw(tid)] = vs[aa_sw(6)];
    const int vi_end = (2 * len - 1) - (2 * t0 - 6);          // window index of the last real 2x-rate sample
    if (vi_end < 2 * TILE + 12 - 1) {
        const int vi = vi_end + 1 + tid;
        if (tid < 8 && vi < 2 * TILE + 16) vs[aa_sw(vi)] = vs[aa_sw(vi_end)];
    }
    __syncthreads();
    // outputs OPT*tid .. OPT*tid+OPT-1: y[t] = sum_j fd[j] * vs[2*(t - t0) + j + 1]
#pragma unroll
    for (int g = 0; g < OPT / 4; ++g) {
        float vw[20];
        const int base = 2 * (OPT * tid + 4 * g);                 // multiple of 8
#pragma unroll
        for (int k = 0; k < 5; ++k) *(f32x4*)&vw[4 * k] = *(const f32x4*)&vs[aa_sw(base + 4 * k)];
        f32x4 o;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 12; ++j) acc = fmaf(fds[j], vw[2 * p + j + 1], acc);
            o[p] = acc;
        }
        const int i0 = OPT * tid + 4 * g;
        if (i0 + 3 < n_out && ((t0 + i0) & 3) == 0 && (T & 3) == 0) {
            *(f32x4*)&yr[t0 + i0] = o;                             // rows start 16-byte aligned when T % 4 == 0
        } else {
#pragma unroll
            for (int p = 0; p < 4; ++p)
                if (i0 + p < n_out) yr[t0 + i0 + p] = o[p];
        }
    }
}

// --------------------------------------------------------------------------------------------------------------
// The same activation written straight into the bf16 x 3 conv's operand: three bf16 planes [B][T][C] (h, m, l: h + m + l == the f32 activation
// exactly; frames >= the row's length are zeros) instead of f32 [B][C][T] -- the producer side of the x3 window kernel (bigvgan_x3.hip), so that the
// f32 activation tensor and the transposing split pass (split_tm3_kernel: 4 B read + 6 B written per element, one more launch per conv) disappear.
// Arithmetic per element = aa_act_kernel_v2<4, true> (same windows, same fmaf order, same reduced v_sin) followed by split_tm3_kernel's split:
// the planes are bit-identical to the two-kernel path (tests/test_gpu_bigvgan_x3.py).
// Block = 32 channels x 128 frames, 4 waves; a wave runs 2 channels at a time on 32 lanes each (a lane owns 4 consecutive frames of one channel, as
// in the v2 kernel) through windows of its own -- no block barrier inside the four rounds, the next round's input is requested before this round's
// arithmetic --; the block's result is transposed through an LDS image [plane][frame][channel] and leaves as 64-byte rows.
// Measured (profiles/r05m, 16 x 1926 frames): 0.78 ms per launch against 0.36 (v2 kernel) + 0.49 (split pass as it was) -- the 27 KiB image holds the
// kernel to 3 blocks per CU.  Two other forms lost: a 64 x 64 tile with block barriers (0.83 ms), and keeping the split results in registers and
// storing each lane's 8-byte pieces directly (no LDS image, 4 blocks per CU: 1.45 ms -- scattered 8-byte stores).
// --------------------------------------------------------------------------------------------------------------
#define AAP_TT 128
#define AAP_CB 32
#define AAP_RS 36           // u16 per frame row of the LDS image (18 dwords: 32 consecutive row slots fall on 32 different even banks)
__device__ __forceinline__ uint32_t aap_cvt2(float a, float b) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{a, b}, bf2));
}
// LDS traffic between the lanes of ONE wave: the LDS executes a wave's instructions in order, the fence keeps the compiler from moving them
__device__ __forceinline__ void aap_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// row slot of frame tt in the LDS image: the four frames of a lane are 32 slots apart, the 32 lanes of a channel take consecutive slots
__device__ __forceinline__ int aap_slot(int tt) { return (tt & 3) * 32 + (tt >> 2); }

__global__ __launch_bounds__(256) void aa_act_planes_kernel(const float* __restrict__ x, u16* __restrict__ xh, u16* __restrict__ xm, u16* __restrict__ xl,
                                                            const float* __restrict__ alpha, const float* __restrict__ beta,
                                                            const float* __restrict__ fu, const float* __restrict__ fd,
                                                            int C, int T, const int* __restrict__ lens, int len_mult, int logscale) {
    // windows start 64 dwords apart modulo the bank row, so the 16 lanes of a ds_*_b128 group meet the v2 kernel's conflict-free pattern (its 2x-rate
    // window swizzle aa_sw included)
    __shared__ __attribute__((aligned(256))) float xs[4][2][192];
    __shared__ __attribute__((aligned(256))) float vs[4][2][320];
    __shared__ __attribute__((aligned(16))) u16 ot[3][AAP_TT][AAP_RS];
    const int b = blockIdx.z, c0 = blockIdx.y * AAP_CB, t0 = blockIdx.x * AAP_TT;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, sub = lane >> 5, pos = lane & 31;
    const int len = lens ? min(lens[b] * len_mult, T) : T;
    const int n_out = len - t0 < AAP_TT ? (len - t0 < 0 ? 0 : len - t0) : AAP_TT;       // frames of this tile inside the row
    if (n_out > 0) {
        float fus[12], fds[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) { fus[j] = 2.0f * fu[j]; fds[j] = fd[j]; }
        float* xw_ = xs[w][sub];
        float* vw_ = vs[w][sub];
        // x window of a round: xw_[i] = x[clamp(t0 - 6 + i)], i in [0, TT + 12): five values per lane
        int toff[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            int t = t0 - 6 + pos + 32 * k;
            toff[k] = t < 0 ? 0 : (t > len - 1 ? len - 1 : t);
        }
        auto chan = [&](int it) { const int c = c0 + w * 8 + it * 2 + sub; return c < C ? c : C - 1; };
        float xn[5];
        {
            const float* xr = x + ((size_t)b * C + chan(0)) * T;
#pragma unroll
            for (int k = 0; k < 5; ++k) xn[k] = xr[toff[k]];
        }
#pragma unroll 1
        for (int it = 0; it < 4; ++it) {
            const int cl = w * 8 + it * 2 + sub;
            const bool c_ok = c0 + cl < C;
            float a_e = alpha[chan(it)], b_e = beta[chan(it)];
            if (logscale) { a_e = expf(a_e); b_e = expf(b_e); }
            const float inv_b = 1.0f / (b_e + 1e-9f);
            aap_wave_sync();                         // the previous round's reads of both windows are done
#pragma unroll
            for (int k = 0; k < 5; ++k) xw_[pos + 32 * k] = xn[k];
            if (it < 3) {
                const float* xr = x + ((size_t)b * C + chan(it + 1)) * T;
#pragma unroll
                for (int k = 0; k < 5; ++k) xn[k] = xr[toff[k]];
            }
            aap_wave_sync();
            auto snake8 = [&](int vb, float* v) {
                float xw[12];
                *(f32x4*)&xw[0] = *(const f32x4*)&xw_[(vb >> 1)];
                *(f32x4*)&xw[4] = *(const f32x4*)&xw_[(vb >> 1) + 4];
                *(f32x4*)&xw[8] = *(const f32x4*)&xw_[(vb >> 1) + 8];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    float ue = 0.f, uo = 0.f;
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        ue = fmaf(fus[1 + 2 * j], xw[p + 5 - j], ue);
                        uo = fmaf(fus[2 * j], xw[p + 6 - j], uo);
                    }
                    const float se = sin_reduced(ue * a_e);
                    const float so = sin_reduced(uo * a_e);
                    v[2 * p] = fmaf(inv_b * se, se, ue);
                    v[2 * p + 1] = fmaf(inv_b * so, so, uo);
                }
            };
            {
                float v[8];
                const int vb = 8 * pos;
                snake8(vb, v);
                *(f32x4*)&vw_[aa_sw(vb)] = *(const f32x4*)&v[0];
                *(f32x4*)&vw_[aa_sw(vb + 4)] = *(const f32x4*)&v[4];
            }
            if (pos < 2) {                       // tail: 2x-rate samples 2 TT .. 2 TT + 15
                float vt[8];
                snake8(2 * AAP_TT + 8 * pos, vt);
                *(f32x4*)&vw_[aa_sw(2 * AAP_TT + 8 * pos)] = *(const f32x4*)&vt[0];
                *(f32x4*)&vw_[aa_sw(2 * AAP_TT + 8 * pos + 4)] = *(const f32x4*)&vt[4];
            }
            aap_wave_sync();
            // replicate padding at the 2x rate (5 left / 6 right)
            if (t0 == 0 && pos < 6) vw_[aa_sw(pos)] = vw_[aa_sw(6)];
            const int vi_end = (2 * len - 1) - (2 * t0 - 6);
            if (vi_end < 2 * AAP_TT + 12 - 1) {
                const int vi = vi_end + 1 + pos;
                if (pos < 8 && vi < 2 * AAP_TT + 16) vw_[aa_sw(vi)] = vw_[aa_sw(vi_end)];
            }
            aap_wave_sync();
            float vw[20];
            const int base = 8 * pos;
#pragma unroll
            for (int k = 0; k < 5; ++k) *(f32x4*)&vw[4 * k] = *(const f32x4*)&vw_[aa_sw(base + 4 * k)];
            float o[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < 12; ++j) acc = fmaf(fds[j], vw[2 * p + j + 1], acc);
                o[p] = (4 * pos + p < n_out && c_ok) ? acc : 0.f;
            }
            // split (split_tm3_kernel's arithmetic) into the LDS image
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float f0 = o[2 * e], f1 = o[2 * e + 1];
                const uint32_t h = aap_cvt2(f0, f1);
                const float r0 = f0 - __uint_as_float(h << 16), r1 = f1 - __uint_as_float(h & 0xffff0000u);          // exact
                const uint32_t m = aap_cvt2(r0, r1);
                const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);          // exact
                const uint32_t l = aap_cvt2(s0, s1);
                const int sa = aap_slot(4 * pos + 2 * e), sb = aap_slot(4 * pos + 2 * e + 1);
                ot[0][sa][cl] = (u16)h; ot[0][sb][cl] = (u16)(h >> 16);
                ot[1][sa][cl] = (u16)m; ot[1][sb][cl] = (u16)(m >> 16);
                ot[2][sa][cl] = (u16)l; ot[2][sb][cl] = (u16)(l >> 16);
            }
        }
    }
    __syncthreads();
    // rows of 32 channels x 2 B: 4 lanes x 16 B per (plane, frame); 64 rows per pass
    const size_t plane_row = (size_t)b * T;
    const int q = tid & 3;
#pragma unroll 1
    for (int r = tid >> 2; r < 3 * AAP_TT; r += 64) {
        const int p = r / AAP_TT, tt = r - p * AAP_TT;
        const int t = t0 + tt;
        if (t >= T || c0 + 8 * q >= C) continue;
        uint2 lo = {0u, 0u}, hi = {0u, 0u};
        if (n_out > 0) {
            const u16* src = &ot[p][aap_slot(tt)][8 * q];
            lo = *(const uint2*)src;
            hi = *(const uint2*)(src + 4);
        }
        u16* dst = (p == 0 ? xh : (p == 1 ? xm : xl)) + (plane_row + t) * C + c0 + 8 * q;
        *(uint4*)dst = uint4{lo.x, lo.y, hi.x, hi.y};
    }
}

// --------------------------------------------------------------------------------------------------------------
// Implicit-GEMM convolution on v_mfma_f32_32x32x2_f32.
//   out[b, co, n(m)] = epi( sum_{j<k} sum_{ci} w[co][ci][j] * x[b, ci, m + tap_base + j*tap_step] )
//   n(m) = m*ostride + ooff.   Regular Conv1d: tap_base = -(k-1)/2*d, tap_step = d, ostride 1, ooff 0.
//   ConvTranspose1d phase r (stride u, pad p): taps {r, r+u} read x[m], x[m-1]; n = m*u + r - p.
// Packed weights (host side, itts_pack_conv_weight): wpk[co_sub][j][ci/8][lane][s] =
//   w[co_sub*32 + (lane&31)][ (ci/8)*8 + 2*s + (lane>>5) ][j]      (zero rows for co >= C_out)
// so one float4 per lane = the A fragments of 4 consecutive K-steps (K-step = 2 input channels of one tap).
// Block = 256 threads = WM x WN waves, each wave MT x NT tiles of 32x32.
// --------------------------------------------------------------------------------------------------------------
#define CI_CHUNK 32
#define CONV_HALO 64   // >= (k-1)*|tap_step| : (11-1)*5 = 50


template <int WM, int WN, int MT, int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((WM == 1 && MT == 1) ? 4 : 1, (WM == 1 && MT == 1) ? 4 : 8)))
void conv_mfma_kernel(ConvArgs a) {
    constexpr int BM = WM * MT * 32;
    constexpr int BN = WN * NT * 32;
    constexpr int LDW = BN + CONV_HALO;
    extern __shared__ __attribute__((aligned(16))) float xs[];   // [CI_CHUNK][LDW]

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w / WN, wn = w % WN;
    // XCD-aware 1-D block mapping: the dispatcher places block L on XCD L % 8 (observed, used for speed only).  All
    // co-tiles of one (batch row, time tile) get consecutive slots of the SAME XCD, so the x tile they share is fetched
    // from HBM once and then served by that XCD's L2 (with co-tile as the slowest grid axis it was re-streamed from HBM
    // once per co-tile: measured FETCH_SIZE 6x the tensor).
    const int L = blockIdx.x;
    const int xcd = L & 7, slot = L >> 3;
    const int tile = (slot / a.n_co) * 8 + xcd;
    if (tile >= a.n_mt * a.B) return;
    const int b = tile / a.n_mt;
    const int m0 = (tile - b * a.n_mt) * BN;
    const int co0 = (slot % a.n_co) * BM;
    const int len_in = a.lens ? min(a.lens[b] * a.len_mult_in, a.Tin) : a.Tin;
    const int len_out = a.lens ? min(a.lens[b] * a.len_mult_out, a.Tout) : a.Tout;
    const int m_count = len_in + a.m_extra;
    if (m0 >= m_count) return;

    const int span = (a.k - 1) * (a.tap_step < 0 ? -a.tap_step : a.tap_step);
    const int last_off = a.tap_base + (a.k - 1) * a.tap_step;
    const int min_off = a.tap_base < last_off ? a.tap_base : last_off;
    const int W = BN + span;
    const int n_cosub = (a.Cout + 31) >> 5;
    const int cin8 = a.Cin >> 3;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const float* xb = a.x + (size_t)b * a.Cin * a.Tin;
    const int lane_row = lane >> 5;          // which of the 2 input channels of a K-step
    const int lane_col = lane & 31;

    // x-tile staging is split (issue-early / write-late): the global loads of chunk c+1 are issued into registers
    // before the MFMA loop of chunk c and written to LDS after it, so their HBM/L2 latency hides under the MFMAs.
    constexpr int NCOL = LDW / 64;             // 64-lane column groups per row (covers BN + halo)
    float stage[CI_CHUNK / 4][NCOL];
    // Rows are fetched through a buffer descriptor per row (base = row start, size = the row's valid length):
    // the hardware bounds check returns 0 for t < 0 (offset wraps to a huge unsigned) and t >= len, and for rows past
    // the channel count (size 0) -- no per-lane guard in the code, so hipcc emits straight back-to-back loads (a
    // guarded load becomes a branch + vmcnt(0) per element and serialises the tile fetch).
    const int w_u = __builtin_amdgcn_readfirstlane(w);
    auto stage_load = [&](int ci0s) {
        const int cnt = min(CI_CHUNK, a.Cin - ci0s);
#pragma unroll
        for (int rr = 0; rr < CI_CHUNK / 4; ++rr) {
            const int r = w_u + 4 * rr;
            const bool row_ok = r < cnt;
            const float* xr = xb + (size_t)(ci0s + (row_ok ? r : 0)) * a.Tin;
            const __amdgpu_buffer_rsrc_t rsrc =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xr), 0, row_ok ? len_in * 4 : 0, 0x00020000);
#pragma unroll
            for (int i = 0; i < NCOL; ++i) {
                const int t = m0 + min_off + lane + 64 * i;
                stage[rr][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, t * 4, 0, 0));
            }
        }
    };
    auto stage_write = [&]() {
#pragma unroll
        for (int rr = 0; rr < CI_CHUNK / 4; ++rr)
#pragma unroll
            for (int i = 0; i < NCOL; ++i) xs[(w + 4 * rr) * LDW + lane + 64 * i] = stage[rr][i];
    };
    stage_load(0);

    // Weight (A) fragments: one tap = CP4 x MT float4 per lane, fetched ONE TAP AHEAD (4 K-groups = 64 MFMAs ~ 1.7 us).
    // vmcnt retires in order, so a fragment load issued behind the 24 staging loads of the next x tile would stall on
    // their HBM latency; issuing the next tap's fragments first and the staging loads second keeps every wait on data
    // that was requested at least one tap earlier.  Loads are unconditional (clamped index + select) -> straight-line.
    constexpr int CP4 = CI_CHUNK / 8;
    f32x4 a_cur[CP4][MT], a_nxt[CP4][MT];
    const int nchunks = (a.Cin + CI_CHUNK - 1) / CI_CHUNK;
    const int G = nchunks * a.k;                       // stream of (chunk, tap) steps
    // packed weights through a buffer descriptor: fragments of padded channel groups / co sub-tiles get an out-of-range
    // offset and read as 0 in hardware (a `cond ? load : 0` select is compiled to a branch + vmcnt(0) per load)
    const unsigned wpk_bytes = (unsigned)n_cosub * a.k * cin8 * 64u * 16u;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wpk), 0, (int)wpk_bytes, 0x00020000);
    auto load_tap = [&](int g, f32x4 (&dst)[CP4][MT]) {
        const int c = g / a.k, j = g - c * a.k;
        const int cbase = c * CI_CHUNK;
        const int cp4n = min(CI_CHUNK, a.Cin - cbase) >> 3;
#pragma unroll
        for (int cp4 = 0; cp4 < CP4; ++cp4) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int co_sub = (co0 >> 5) + wm * MT + mt;
                const bool ok = (cp4 < cp4n) && (co_sub < n_cosub);            // wave-uniform
                const unsigned off = ok ? ((unsigned)((co_sub * a.k + j) * cin8 + ((cbase >> 3) + cp4)) * 64u + lane) * 16u
                                        : 0xFFFFFFF0u;
                dst[cp4][mt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, off, 0, 0));
            }
        }
    };
    auto compute_tap = [&](int j) {
        const int colb = wn * NT * 32 + lane_col + (a.tap_base + j * a.tap_step - min_off);
        const float* xcol = xs + lane_row * LDW + colb;
        // B fragments (LDS) are read one K-group ahead of the MFMAs that consume them
        float bcur[4][NT], bnxt[4][NT];
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bcur[sidx][nt] = xcol[(2 * sidx) * LDW + nt * 32];
#pragma unroll
        for (int cp4 = 0; cp4 < CP4; ++cp4) {
            if (cp4 + 1 < CP4) {
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bnxt[sidx][nt] = xcol[((cp4 + 1) * 8 + 2 * sidx) * LDW + nt * 32];
            }
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[cp4][mt][sidx], bcur[sidx][nt], acc[mt][nt], 0, 0, 0);
            if (cp4 + 1 < CP4) {
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bcur[sidx][nt] = bnxt[sidx][nt];
            }
        }
#pragma unroll
        for (int cp4 = 0; cp4 < CP4; ++cp4)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a_cur[cp4][mt] = a_nxt[cp4][mt];
    };
    load_tap(0, a_cur);
    int g = 0;
    for (int ci0 = 0; ci0 < a.Cin; ci0 += CI_CHUNK) {
        __syncthreads();   // previous chunk fully consumed
        stage_write();
        __syncthreads();
        // Tap 0 is straight-line: every load below is issued unconditionally (the last step re-fetches its own tap, a
        // chunk past the end gets zero-size descriptors) so the compiler knows the exact number of loads in flight and
        // can wait with counted vmcnt(N) instead of draining the queue.
        load_tap(g + 1 < G ? g + 1 : G - 1, a_nxt);    // next tap's weights first ...
        stage_load(ci0 + CI_CHUNK);                     // ... then the next x tile (behind them in vmcnt order)
        compute_tap(0);
        ++g;
        for (int j = 1; j < a.k; ++j, ++g) {
            load_tap(g + 1 < G ? g + 1 : G - 1, a_nxt);
            compute_tap(j);
        }
    }

    // epilogue: C/D layout col(N) = lane&31, row(M) = (r&3) + 8*(r>>2) + 4*(lane>>5)
    // Branch-free: residual / accumulate reads and the stores go through one buffer descriptor per batch row
    // ([C_out][T_out] of row b); an invalid element gets an out-of-range offset, for which the hardware returns 0 on
    // loads and drops stores.  (Per-element `if (valid) { load; ...; store }` compiles to a branch + vmcnt(0) per
    // element, i.e. 64 serialised memory round trips per lane per tile.)
    const int plane_bytes = a.Cout * a.Tout * 4;
    const __amdgpu_buffer_rsrc_t yrs =
        __builtin_amdgcn_make_buffer_rsrc(a.y + (size_t)b * a.Cout * a.Tout, 0, plane_bytes, 0x00020000);
    const float* resb = a.res ? a.res + (size_t)b * a.Cout * a.Tout : a.y;
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(resb), 0, plane_bytes, 0x00020000);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {          // 8 accumulator rows at a time keeps the register footprint small
            int corow[8];
            float badd[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int r = hf * 8 + q;
                corow[q] = co0 + (wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lane_row;
                badd[q] = 0.f;
            }
            if (a.bias) {
#pragma unroll
                for (int q = 0; q < 8; ++q) badd[q] = a.bias[corow[q] < a.Cout ? corow[q] : a.Cout - 1];
            }
            if (a.bias_b) {
                const float* bb = a.bias_b + (size_t)b * a.Cout;
#pragma unroll
                for (int q = 0; q < 8; ++q) badd[q] += bb[corow[q] < a.Cout ? corow[q] : a.Cout - 1];
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int m = m0 + (wn * NT + nt) * 32 + lane_col;
                const int n = m * a.ostride + a.ooff;
                const bool n_ok = (m < m_count) && (n >= 0) && (n < len_out);
                unsigned voff[8];
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const bool ok = n_ok && corow[q] < a.Cout;
                    voff[q] = ok ? (unsigned)(corow[q] * a.Tout + n) * 4u : 0xFFFFFFF0u;
                    v[q] = acc[mt][nt][hf * 8 + q] + badd[q];
                }
                if (a.res) {
                    float rv[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) rv[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, voff[q], 0, 0));
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] += rv[q];
                }
                if (a.acc_mode != 0) {
                    float yo[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) yo[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yrs, voff[q], 0, 0));
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = a.acc_mode == 1 ? yo[q] + v[q] : (yo[q] + v[q]) / a.div;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[q]), yrs, voff[q], 0, 0);
            }
        }
    }
}

// --------------------------------------------------------------------------------------------------------------
// conv_post: C_in -> 1 channel, k taps, optional bias, then clamp(-1,1) or tanh; writes 0 beyond the row length.
// --------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_post_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        int Cin, int T, int k, const int* __restrict__ lens,
                                                        int len_mult, int use_tanh) {
    extern __shared__ float ws[];   // [Cin*k]
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < Cin * k; i += 256) ws[i] = w[i];
    __syncthreads();
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const int len = lens ? min(lens[b] * len_mult, T) : T;
    float acc = 0.f;
    if (t < len) {
        const int pad = (k - 1) / 2;
        acc = bias ? bias[0] : 0.f;
        const float* xb = x + (size_t)b * Cin * T;
        for (int ci = 0; ci < Cin; ++ci) {
            const float* xr = xb + (size_t)ci * T;
            for (int j = 0; j < k; ++j) {
                const int tt = t + j - pad;
                if (tt >= 0 && tt < len) acc = fmaf(ws[ci * k + j], xr[tt], acc);
            }
        }
        acc = use_tanh ? tanhf(acc) : fminf(1.0f, fmaxf(-1.0f, acc));
    }
    y[(size_t)b * T + t] = acc;
}

// --------------------------------------------------------------------------------------------------------------
// v1 speaker conditioning: out[b][co] = bias[co] + sum_d w[co][d] * spk[b][d]   (1x1 conv of a length-1 signal,
// indextts/BigVGAN/models.py:191-197,226,236)
// --------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cond_bias_kernel(const float* __restrict__ spk, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ out,
                                                        int Cout, int cond_dim) {
    const int b = blockIdx.y;
    const int co = blockIdx.x * 256 + threadIdx.x;
    if (co >= Cout) return;
    float acc = bias ? bias[co] : 0.f;
    const float* wr = w + (size_t)co * cond_dim;
    const float* sr = spk + (size_t)b * cond_dim;
    for (int d = 0; d < cond_dim; ++d) acc = fmaf(wr[d], sr[d], acc);
    out[(size_t)b * Cout + co] = acc;
}

// --------------------------------------------------------------------------------------------------------------
// host launchers (called from capi_bigvgan.hip)
// --------------------------------------------------------------------------------------------------------------
int launch_aa_act(const float* x, float* y, const float* alpha, const float* beta, const float* fu, const float* fd,
                  int B, int C, int T, const int* lens, int len_mult, int logscale, hipStream_t st) {
    if (B <= 0 || C <= 0 || T <= 0) return ITTS_OK;
    // option aa_act: 0 = v1 (LDS per tap, libm sinf), 1 = register-tiled + sinf, 2 = register-tiled + reduced v_sin (4/thread),
    //              3 = same with 8 outputs per thread.  Measured (B=16, T=1926): 4.17 / 5.4 / 3.66 / 4.98 ms per 6 launches
    //              -> default 2.
    const int mode = itts_opt(OPT_AA_ACT);
    if (mode == 0) hipLaunchKernelGGL(aa_act_kernel, dim3(ceil_div(T, AA_TILE), C, B), dim3(256), 0, st, x, y, alpha, beta, fu, fd, C, T, lens, len_mult, logscale);
    else if (mode == 1) hipLaunchKernelGGL((aa_act_kernel_v2<4, false>), dim3(ceil_div(T, 1024), C, B), dim3(256), 0, st, x, y, alpha, beta, fu, fd, C, T, lens, len_mult, logscale);
    else if (mode == 2) hipLaunchKernelGGL((aa_act_kernel_v2<4, true>), dim3(ceil_div(T, 1024), C, B), dim3(256), 0, st, x, y, alpha, beta, fu, fd, C, T, lens, len_mult, logscale);
    else hipLaunchKernelGGL((aa_act_kernel_v2<8, true>), dim3(ceil_div(T, 2048), C, B), dim3(256), 0, st, x, y, alpha, beta, fu, fd, C, T, lens, len_mult, logscale);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

// the activation as the x3 conv's operand planes (aa_act_planes_kernel); xp = [3][B][T][C] bf16
int launch_aa_act_planes(const float* x, void* xp, const float* alpha, const float* beta, const float* fu, const float* fd,
                         int B, int C, int T, const int* lens, int len_mult, int logscale, hipStream_t st) {
    if (B <= 0 || C <= 0 || T <= 0) return ITTS_OK;
    if (C % 8) { itts_set_error("aa_act_planes: C %% 8 != 0"); return ITTS_ERR_ARG; }
    u16* p = (u16*)xp;
    const size_t plane = (size_t)B * T * C;
    hipLaunchKernelGGL(aa_act_planes_kernel, dim3(ceil_div(T, AAP_TT), ceil_div(C, AAP_CB), B), dim3(256), 0, st, x, p, p + plane, p + 2 * plane,
                       alpha, beta, fu, fd, C, T, lens, len_mult, logscale);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

template <int WM, int WN, int MT, int NT>
static int launch_conv_cfg(const ConvArgs& a, int B, int m_total, hipStream_t st) {
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    const size_t lds = (size_t)CI_CHUNK * (BN + CONV_HALO) * sizeof(float);
    ConvArgs a2 = a;
    a2.B = B;
    a2.n_mt = ceil_div(m_total, BN);
    a2.n_co = ceil_div(a.Cout, BM);
    const long long tiles8 = ((long long)a2.n_mt * B + 7) / 8 * 8;
    const long long nblocks = tiles8 * a2.n_co;
    if (nblocks > 2147483647ll) { itts_set_error("conv: grid too large"); return ITTS_ERR_ARG; }
    hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, MT, NT>), dim3((unsigned)nblocks), dim3(256), lds, st, a2);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

int launch_conv(const ConvArgs& a, int B, hipStream_t st) {
    if (B <= 0) return ITTS_OK;
    if (a.Cin % 8 != 0) { itts_set_error("conv: C_in=%d must be a multiple of 8", a.Cin); return ITTS_ERR_ARG; }
    if ((long long)a.Cout * a.Tout >= (1ll << 29)) { itts_set_error("conv: C_out*T_out = %lld exceeds the 2 GiB per-row plane limit", (long long)a.Cout * a.Tout); return ITTS_ERR_ARG; }
    if ((long long)a.Tin >= (1ll << 29)) { itts_set_error("conv: T too large"); return ITTS_ERR_ARG; }
    const int span = (a.k - 1) * (a.tap_step < 0 ? -a.tap_step : a.tap_step);
    if (span > CONV_HALO) { itts_set_error("conv: tap span %d exceeds halo %d", span, CONV_HALO); return ITTS_ERR_ARG; }
    const int m_total = a.Tin + a.m_extra;
    const int n_cosub = (a.Cout + 31) / 32;
    // co-tile choice (measured, profiles/r01_conv_tiles.txt): the 32-row tile <1,4,1,2> fits 118 registers without spilling,
    // i.e. 4 waves/SIMD with its 40 KiB x tile (4 blocks = the CU's 160 KiB of LDS), and beats or ties the taller tiles
    // (occupancy 2 or 1) at every channel count of the generator.  Option conv_bm (32/64/96/128) forces a height for experiments.
    const int force_bm = itts_opt(OPT_CONV_BM);
    const int bm_sub = force_bm ? force_bm / 32 : 1;               // co sub-tiles (32 rows each) per block
    (void)n_cosub;
    switch (bm_sub) {
        case 1: return launch_conv_cfg<1, 4, 1, 2>(a, B, m_total, st);
        case 2: return launch_conv_cfg<1, 4, 2, 2>(a, B, m_total, st);
        case 3: return launch_conv_cfg<1, 4, 3, 2>(a, B, m_total, st);
        default: return launch_conv_cfg<2, 2, 2, 2>(a, B, m_total, st);
    }
}

int launch_conv_post(const float* x, float* y, const float* w, const float* bias, int B, int Cin, int T, int k,
                     const int* lens, int len_mult, int use_tanh, hipStream_t st) {
    if (B <= 0 || T <= 0) return ITTS_OK;
    dim3 grid(ceil_div(T, 256), B);
    hipLaunchKernelGGL(conv_post_kernel, grid, dim3(256), (size_t)Cin * k * sizeof(float), st, x, y, w, bias, Cin, T, k,
                       lens, len_mult, use_tanh);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

int launch_cond_bias(const float* spk, const float* w, const float* bias, float* out, int B, int Cout, int cond_dim,
                     hipStream_t st) {
    if (B <= 0 || Cout <= 0) return ITTS_OK;
    dim3 grid(ceil_div(Cout, 256), B);
    hipLaunchKernelGGL(cond_bias_kernel, grid, dim3(256), 0, st, spk, w, bias, out, Cout, cond_dim);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}
