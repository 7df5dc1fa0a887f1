// Vocoder conv mode "bf16 x 3": the resblock Conv1d's of BigVGAN (bigvgan.py:132-141, AMPBlock1) on the bf16 matrix cores with every f32 operand
// carried EXACTLY as three bf16 planes -- the arithmetic of the fp32x3 GEMMs of the flow-matching stage (gpt_kernels.hip, gemm_x3_kernel):
//
//   x = xh + xm + xl,  xh = bf16(x), xm = bf16(x - xh), xl = bf16(x - xh - xm)         (8 + 8 + 8 significand bits, nothing dropped; bf16 has the
//   w = wh + wm + wl   (packed once at load time)                                       f32 exponent range, so there is no range condition either)
//   y = sum of the six plane products  xl wh, xh wl, xm wm, xm wh, xh wm, xh wh          (smallest first; the dropped xm wl / xl wm / xl wl terms are
//                                                                                         <= 2^-24 |x w| each)
//
// Every plane product is exact in the f32 accumulator of v_mfma_f32_16x16x32_bf16, accumulation stays f32: the result is held to an error against
// an f64 convolution that is not above the exact-f32 MFMA kernel's (tests/test_gpu_bigvgan_x3.py), which is why this mode -- unlike the 22-bit
// f16 x 3 mode of bigvgan_h3.hip -- may carry the benchmark's headline.  Six bf16 MFMAs per f32-equivalent MFMA: 2.5 PFLOP/s / 6 = 417 TFLOP/s
// f32-equivalent against 157 for the f32 matrix pipe.
//
// Data flow per conv (the structure of bigvgan_h3.hip's window kernel): the activation's f32 [B][C][T] output is split into three token-major bf16
// tensors [B][T][C] (split_tm3_kernel, rows beyond a row's length zeroed); the conv runs as a GEMM with M = frames, N = output channels,
// K = (tap, 32 input channels).  Block = 256 frames x 96 output channels (every channel count of the generator from 1536 down to 96 is a multiple
// of 96), 8 waves as 4 (frames) x 2 (channels), wave tile 64 x 48.  For one 32-channel chunk every tap reads the same frames shifted by
// tap x dilation, so the three plane images of frames m0 - pad .. m0 + 255 + pad are staged ONCE per chunk (double-buffered LDS-DMA,
// k-group-major 16-row chunks: conflict-free ds_read_b128 at any row offset) and only the weight fragments stream per (tap, chunk) K tile through
// a two-deep ring -- with six products per fragment pair a K tile is ~2 300 matrix-pipe cycles per SIMD, so one tile of lookahead covers the L2
// latency that needed three tiles in the 3-product kernel.  LDS: 2 x 60 KiB (windows of 256 + 64 frames x 3 planes) + 2 x 18 KiB = 156 KiB, one
// block (8 waves) per CU.  The accumulator fragment is 4 consecutive frames of one output channel = one 16-byte piece of the channel-major f32
// output, so bias / residual / MRF accumulate work on the [B][C][T] tensors of the f32 path unchanged.
#include <string.h>

#include "bigvgan_kernels.h"
#include "common.h"

typedef __bf16 x3_bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t x3_v4u __attribute__((ext_vector_type(4)));

#define CX3_BK 32            // input channels per K tile (one MFMA k-step)
#define CX3_MT 4             // m-tiles (16 frames) per wave: block = 4 waves x 64 = 256 frames
#define CX3_NT 3             // n-tiles (16 channels) per wave: block = 2 waves x 48 = 96 channels
#define CX3_BM 256
#define CX3_BN 96
#define CX3_SLACK 64         // window rows beyond the block's frames: needs (k - 1) * dil <= 64
#define CX3_ACH ((CX3_BM + CX3_SLACK) / 16)          // 20 chunks of 16 rows
#define CX3_ABUF (3 * CX3_ACH * 1024)                // 60 KiB: three planes
#define CX3_NWT 6                                    // n-tiles of a weight stage
#define CX3_WST (3 * CX3_NWT * 1024)                 // 18 KiB
#define CX3_LDS (2 * CX3_ABUF + 2 * CX3_WST)

__device__ __forceinline__ uint32_t cx3_cvt2(float a, float b) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{a, b}, bf2));
}

// f32 [B][C][T] -> three bf16 planes [B][T][C] (h, m, l: h + m + l == x exactly); frames >= the row's length are written as zeros (the conv's zero
// padding on the right)
__global__ __launch_bounds__(256) void split_tm3_kernel(const float* __restrict__ x, u16* __restrict__ xh, u16* __restrict__ xm, u16* __restrict__ xl,
                                                        int C, int T, const int* __restrict__ lens, int len_mult) {
    __shared__ float tile[64][65];
    const int b = blockIdx.z, c0 = blockIdx.y * 64, t0 = blockIdx.x * 64;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int len = lens ? min(lens[b] * len_mult, T) : T;
    const float* xb = x + (size_t)b * C * T;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + w * 16 + i, t = t0 + lane;
        tile[w * 16 + i][lane] = (c < C && t < len) ? xb[(size_t)c * T + t] : 0.f;
    }
    __syncthreads();
    // 8 lanes x 16 B = the 128 contiguous bytes of a frame's 64 channels in a plane; 32 frames per pass.  (First version: 4 lanes per frame, each
    // writing channels [16 q, 16 q + 8) and then [16 q + 8, 16 q + 16): every store instruction left 16-byte holes in its 32-byte sectors.)
    const int cb = (tid & 7) * 8;
    if (c0 + cb >= C) return;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        const int tl = (tid >> 3) + 32 * hf, t = t0 + tl;
        if (t >= T) break;
        x3_v4u ph, pm, pl;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float f0 = tile[cb + 2 * e][tl], f1 = tile[cb + 2 * e + 1][tl];
            const uint32_t h = cx3_cvt2(f0, f1);
            const float r0 = f0 - __uint_as_float(h << 16), r1 = f1 - __uint_as_float(h & 0xffff0000u);          // exact
            const uint32_t m = cx3_cvt2(r0, r1);
            const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);          // exact
            ph[e] = h; pm[e] = m; pl[e] = cx3_cvt2(s0, s1);
        }
        const size_t o = ((size_t)b * T + t) * C + c0 + cb;
        *(x3_v4u*)(xh + o) = ph;
        *(x3_v4u*)(xm + o) = pm;
        *(x3_v4u*)(xl + o) = pl;
    }
}

template <int N>
__device__ __forceinline__ void cx3_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// NH = co tiles of 96 channels a block runs against ONE staged window (1 or 2): the window of a 32-channel chunk (3 planes x up to 20 KiB) is the larger
// part of a K tile's LDS-DMA traffic -- at k = 3 a 96-wide block moved 60 + 54 KiB per 6 900 matrix-pipe cycles per SIMD and was bound by the
// CU's fill rate (~12 B / clk), profiles/r05d/voc_bench.log -- so with NH = 2 the block walks (chunk, co half, tap): two accumulator sets, the same
// weight bytes as two blocks, half the window bytes.
template <int NH>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_x3w_kernel(ConvX3Args a) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = w >> 1, wc = w & 1;
    // XCD-aware 1-D mapping as in conv_mfma_kernel: the co tiles of one (row, frame tile) sit on one XCD and share its L2 copy of x
    const int L = blockIdx.x;
    const int xcd = L & 7, slot = L >> 3;
    const int tile = (slot / a.n_co) * 8 + xcd;
    if (tile >= a.n_mt * a.B) return;
    const int b = tile / a.n_mt;
    const int m0 = (tile - b * a.n_mt) * CX3_BM;
    const int co0 = (slot % a.n_co) * (CX3_BN * NH);
    const int len = a.lens ? min(a.lens[b] * a.len_mult, a.T) : a.T;
    if (m0 >= len) return;
    const int nkc = a.Cin / CX3_BK, G = a.k * nkc * NH;
    const int ntiles = (a.Cout + 15) >> 4;
    const int pad = (a.k - 1) / 2 * a.dil;
    const int nach = (CX3_BM + 2 * pad + 15) >> 4;                      // row chunks of the window actually needed (17 .. CX3_ACH)

    // ---- staging sources.  A chunk = 16 frames x 64 bytes of one plane, held K-GROUP MAJOR: piece (frame r, k-group q) at q * 256 + r * 16; the
    // DMA writes lane l to byte 16 l of the chunk, so lane l fetches (frame l & 15, k-group l >> 4).
    const int srow = lane & 15, skp = lane >> 4;
    const size_t plane = (size_t)a.B * a.T * a.Cin * 2;                 // bytes of one token-major plane
    const char* x_b = (const char*)a.xp + ((size_t)b * a.T * a.Cin + skp * 8) * 2;
    const char* zr = (const char*)a.zero_row;
    // wave w stages window chunks w, w + 8, w + 16 (< nach): 2 or 3 chunks x 3 planes -- the count is wave-uniform and known up front, so the
    // counted wait below (how many of this wave's own LDS-DMA instructions may stay in flight) is exact per wave
    const int my_ach = (nach - w + 7) >> 3;
    auto issue_a = [&](int kc_, int buf) {
        char* base = sm + buf * CX3_ABUF;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int c = w + 8 * i;
            if (c < nach) {                                              // wave-uniform
                const int t = m0 - pad + c * 16 + srow;
                const bool ok = (unsigned)t < (unsigned)a.T;
                const size_t o = ((size_t)(ok ? t : 0) * a.Cin + kc_ * CX3_BK) * 2;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ok ? x_b + pl * plane + o : zr),
                                                     (__attribute__((address_space(3))) void*)(base + (pl * CX3_ACH + c) * 1024), 16, 0, 0);
            }
        }
    };
    // weight stage: 6 n-tiles x 3 planes; waves 0 .. 5 stage one n-tile each (3 DMA instructions), waves 6 and 7 none
    const bool w_on = w < CX3_NWT;
    auto issue_w = [&](int g) {                                         // K tile g = (chunk, co half, tap); packed index tap * nkc + chunk
        if (!w_on) return;
        const int kc_ = g / (a.k * NH), r_ = g - kc_ * a.k * NH;
        const int h_ = r_ / a.k, j_ = r_ - h_ * a.k;
        int ntg = (co0 >> 4) + h_ * CX3_NWT + w;
        ntg = ntg < ntiles ? ntg : ntiles - 1;
        const char* wsrc = (const char*)a.wp + ((size_t)ntg * a.k * nkc + (size_t)j_ * nkc + kc_) * 1024 + lane * 16;
        const size_t wstream = (size_t)ntiles * a.k * nkc * 1024;       // bytes of one weight plane stream
        char* base = sm + 2 * CX3_ABUF + (g & 1) * CX3_WST;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + pl * wstream),
                                             (__attribute__((address_space(3))) void*)(base + (pl * CX3_NWT + w) * 1024), 16, 0, 0);
    };

    f32x4 acc[NH][CX3_MT][CX3_NT];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int i = 0; i < CX3_MT; ++i)
#pragma unroll
            for (int jn = 0; jn < CX3_NT; ++jn) acc[h][i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int row16 = lane & 15, kg = lane >> 4;
    const int b_off = wc * CX3_NT * 1024 + lane * 16;

    issue_a(0, 0);
    issue_w(0);
    // window fragments of one K tile: tap j of chunk kc, three planes per m-tile
    auto read_a = [&](int kc, int j, x3_v4u (&af)[CX3_MT][3]) {
        const char* abase = sm + (kc & 1) * CX3_ABUF;
        const int rbase = wr * CX3_MT * 16 + row16 + j * a.dil;         // window row of this lane's first frame for tap j
#pragma unroll
        for (int mt = 0; mt < CX3_MT; ++mt) {
            const int rr = rbase + mt * 16;
            const char* ap = abase + (rr >> 4) * 1024 + kg * 256 + (rr & 15) * 16;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) af[mt][pl] = *(const x3_v4u*)(ap + pl * CX3_ACH * 1024);
        }
    };
    // One K tile: wait, barrier, request the next weight tile (and, at a chunk's first tile, the next chunk's window BEHIND it: a wait for the
    // weights then never drains the window), 72 MFMAs per wave.  a_recent: the window request of the previous iteration may stay in flight.
    // The window fragments of the NEXT K tile are read during this tile's MFMAs (m-tile by m-tile, into the registers that m-tile's MFMAs have just
    // consumed): all eight waves leave the barrier together, and with the
    // 12 window reads + 9 weight reads per wave issued behind it the LDS was busy ~1 300 cycles per K tile before the first MFMA could start
    // (profiles/r05d: 213 TFLOP/s on the 768-channel stage); the next tile's window is resident by then -- the same chunk's, or the next chunk's,
    // which was requested k NH >= 3 tiles earlier and waited for (vmcnt(0) + barrier) two tiles after its request.
    bool a_recent = false;
    x3_v4u afc[CX3_MT][3];                                              // this tile's window fragments (read during the previous tile)
    auto ktile = [&](int g, int kc, int kc_n, int j_n, f32x4 (&ac)[CX3_MT][CX3_NT], bool first_of_chunk) {
        if (a_recent) { if (my_ach == 3) cx3_wait_vm<9>(); else cx3_wait_vm<6>(); } else cx3_wait_vm<0>();
        __builtin_amdgcn_s_barrier();                                   // ... for every wave; the other weight stage and window buffer are free again
        asm volatile("" ::: "memory");                                  // (raw barrier: __syncthreads' fence would drain the window request in flight)
        if (g + 1 < G) issue_w(g + 1);
        a_recent = false;
        if (first_of_chunk && kc + 1 < nkc) { issue_a(kc + 1, (kc + 1) & 1); a_recent = true; }
        const char* wbase = sm + 2 * CX3_ABUF + (g & 1) * CX3_WST;
        x3_v4u bf[CX3_NT][3];
#pragma unroll
        for (int nt = 0; nt < CX3_NT; ++nt)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) bf[nt][pl] = *(const x3_v4u*)(wbase + pl * CX3_NWT * 1024 + b_off + nt * 1024);
        if (g == 0) read_a(0, 0, afc);                                  // the first tile's own fragments (its window landed with this barrier)
        const char* abase_n = sm + (kc_n & 1) * CX3_ABUF;
        const int rbase_n = wr * CX3_MT * 16 + row16 + j_n * a.dil;
        const bool more = g + 1 < G;
#pragma unroll
        for (int mt = 0; mt < CX3_MT; ++mt) {
            // plane pairs, smallest terms first (x plane, w plane): l h, h l, m m, m h, h m, h h
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
            constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int nt = 0; nt < CX3_NT; ++nt)
                    ac[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(x3_bf16x8, afc[mt][PA[q]]),
                                                                         __builtin_bit_cast(x3_bf16x8, bf[nt][PB[q]]), ac[mt][nt], 0, 0, 0);
            if (more) {                                                 // this m-tile's fragments of the NEXT K tile, into the registers just consumed
                const int rr = rbase_n + mt * 16;
                const char* ap = abase_n + (rr >> 4) * 1024 + kg * 256 + (rr & 15) * 16;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) afc[mt][pl] = *(const x3_v4u*)(ap + pl * CX3_ACH * 1024);
            }
        }
    };
    {
        // K tiles in (chunk, co half, tap) order
        int g = 0;
        for (int kc = 0; kc < nkc; ++kc) {
#pragma unroll
            for (int h = 0; h < NH; ++h)
                for (int j = 0; j < a.k; ++j, ++g) {
                    int j_n = j + 1, kc_n = kc;                          // the successor tile's tap and chunk (the co half does not matter to the window)
                    if (j_n == a.k) { j_n = 0; if (h == NH - 1) kc_n = kc + 1; }
                    ktile(g, kc, kc_n, j_n, acc[h], h == 0 && j == 0);
                }
        }
    }

    const bool vec = (a.T & 3) == 0;
    float* yb = a.y + (size_t)b * a.Cout * a.T;
    const float* rb = a.res ? a.res + (size_t)b * a.Cout * a.T : nullptr;
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int nt = 0; nt < CX3_NT; ++nt) {
        const int co = co0 + h * CX3_BN + (wc * CX3_NT + nt) * 16 + (lane & 15);
        if (co >= a.Cout) continue;
        const float bias = a.bias ? a.bias[co] : 0.f;
#pragma unroll
        for (int mt = 0; mt < CX3_MT; ++mt) {
            const int t = m0 + (wr * CX3_MT + mt) * 16 + (lane >> 4) * 4;
            if (t >= len) continue;
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[h][mt][nt][r] + bias;
            const size_t o = (size_t)co * a.T + t;
            if (vec && t + 3 < len) {
                if (rb) { const f32x4 rv = *(const f32x4*)(rb + o); v += rv; }
                if (a.acc_mode != 0) {
                    const f32x4 yo = *(const f32x4*)(yb + o);
                    v = yo + v;
                    if (a.acc_mode == 2) v = v / a.div;
                }
                *(f32x4*)(yb + o) = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (t + r >= len) break;
                    float s = v[r];
                    if (rb) s += rb[o + r];
                    if (a.acc_mode != 0) { s = yb[o + r] + s; if (a.acc_mode == 2) s = s / a.div; }
                    yb[o + r] = s;
                }
            }
        }
    }
}

bool conv_x3_supported(int Cin, int Cout, int k, int dil) {
    return Cin >= CX3_BK && Cin % CX3_BK == 0 && Cout >= 1 && (k & 1) && k >= 3 && (k - 1) * dil <= CX3_SLACK;
}

size_t conv_x3_packed_bytes(int Cout, int Cin, int k) {
    return (size_t)3 * ((Cout + 15) / 16) * ((size_t)k * (Cin / CX3_BK)) * 1024;
}

// w [Cout][Cin][k] f32 -> three streams of MFMA B fragments: out[s][nt][kt = j * Cin/32 + kc][lane = kg*16 + n][e] =
//   plane_s( w[nt*16 + n][kc*32 + kg*8 + e][j] ),  plane_0 = bf16(w), plane_1 = bf16(w - plane_0), plane_2 = bf16(w - plane_0 - plane_1); zero rows beyond Cout
static uint16_t cx3_bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7f800000u) == 0x7f800000u) return (uint16_t)(u >> 16);          // inf / nan: truncate (the packer rejects them below)
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float cx3_bf16_f(uint16_t h) { const uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

int conv_x3_pack(const float* w, int Cout, int Cin, int k, void* out) {
    if (!w || !out || Cout < 1 || Cin < CX3_BK || Cin % CX3_BK || k < 1 || !(k & 1)) {
        itts_set_error("conv_x3_pack: need C_in %% 32 == 0 and an odd kernel size (Cout=%d Cin=%d k=%d)", Cout, Cin, k);
        return ITTS_ERR_ARG;
    }
    const int ntiles = (Cout + 15) / 16, nkc = Cin / CX3_BK;
    const size_t nk = (size_t)k * nkc, stream = (size_t)ntiles * nk * 512;
    u16* o0 = (u16*)out;
    for (int nt = 0; nt < ntiles; ++nt)
        for (int j = 0; j < k; ++j)
            for (int kc = 0; kc < nkc; ++kc)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int co = nt * 16 + (lane & 15), ci = kc * CX3_BK + (lane >> 4) * 8 + e;
                        const float f = co < Cout ? w[((size_t)co * Cin + ci) * k + j] : 0.f;
                        if (!(f - f == 0.f)) { itts_set_error("conv_x3_pack: weight %g is not finite", (double)f); return ITTS_ERR_ARG; }
                        const uint16_t h = cx3_bf16_rne(f);
                        const float r = f - cx3_bf16_f(h);
                        const uint16_t m = cx3_bf16_rne(r);
                        const uint16_t l = cx3_bf16_rne(r - cx3_bf16_f(m));
                        const size_t o = (((size_t)nt * nk + (size_t)j * nkc + kc) * 64 + lane) * 8 + e;
                        o0[o] = h; o0[stream + o] = m; o0[2 * stream + o] = l;
                    }
    return ITTS_OK;
}

int launch_split_tm3(const float* x, void* xp, int B, int C, int T, const int* lens, int len_mult, hipStream_t st) {
    if (B <= 0 || C <= 0 || T <= 0) return ITTS_OK;
    if (C % 8) { itts_set_error("split_tm3: C %% 8 != 0"); return ITTS_ERR_ARG; }
    u16* p = (u16*)xp;
    const size_t plane = (size_t)B * T * C;
    hipLaunchKernelGGL(split_tm3_kernel, dim3(ceil_div(T, 64), ceil_div(C, 64), B), dim3(256), 0, st, x, p, p + plane, p + 2 * plane, C, T, lens, len_mult);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

int launch_conv_x3(const ConvX3Args& a0, hipStream_t st) {
    if (a0.B <= 0 || a0.T <= 0) return ITTS_OK;
    if (!conv_x3_supported(a0.Cin, a0.Cout, a0.k, a0.dil)) {
        itts_set_error("conv_x3: need C_in %% 32 == 0, odd k >= 3 and (k - 1) * dilation <= %d (Cin=%d k=%d dil=%d)", CX3_SLACK, a0.Cin, a0.k, a0.dil);
        return ITTS_ERR_ARG;
    }
    if ((long long)a0.Cout * a0.T >= (1ll << 31) || (long long)a0.Cin * a0.T >= (1ll << 31)) { itts_set_error("conv_x3: row plane too large"); return ITTS_ERR_ARG; }
    ConvX3Args a = a0;
    const int nh = a.Cout > CX3_BN ? 2 : 1;                            // two co tiles per staged window where there are two
    a.n_mt = ceil_div(a.T, CX3_BM);
    a.n_co = ceil_div(a.Cout, CX3_BN * nh);
    const long long tiles8 = ((long long)a.n_mt * a.B + 7) / 8 * 8;
    const long long nblocks = tiles8 * a.n_co;
    if (nblocks > 2147483647ll) { itts_set_error("conv_x3: grid too large"); return ITTS_ERR_ARG; }
    static ItPerDevice<bool> attr_set_pd;
    bool& attr_set = attr_set_pd.cur();
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)conv_x3w_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, CX3_LDS));
        HIP_TRY(hipFuncSetAttribute((const void*)conv_x3w_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, CX3_LDS));
        attr_set = true;
    }
    if (nh == 2) hipLaunchKernelGGL(conv_x3w_kernel<2>, dim3((unsigned)nblocks), dim3(512), CX3_LDS, st, a);
    else hipLaunchKernelGGL(conv_x3w_kernel<1>, dim3((unsigned)nblocks), dim3(512), CX3_LDS, st, a);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}
