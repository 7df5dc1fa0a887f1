// Opt-in second vocoder mode: the resblock Conv1d's on the f16 matrix cores with SPLIT f32 operands ("f16 x 3").
//
//   x = xh + 2^-11 xl,  xh = f16(x) (0 where that would be an f16 subnormal),  xl = f16(2^11 (x - xh))
//                                      (22 significand bits; the scale keeps xl a normal f16 down to |x| = 3e-8)
//   w = wh + 2^-11 wl,  likewise (packed once at load time)
//   y = sum xh wh  +  2^-11 ( sum xh wl + sum xl wh )                   (the xl wl term, 2^-22 relative, is dropped)
//
// Every product of two f16 values is exact in the f32 accumulator, so the result differs from the f32-MFMA path only by the
// 2^-22 representation error of the split -- measured on the reference's BigVGAN goldens it is indistinguishable from the f32
// path (CPU emulation of this arithmetic: 3.3e-7 .. 1.2e-6 RMS against 1.5e-7 .. 7.4e-7 for plain f32).
// Three v_mfma_f32_16x16x32_f16 per fragment pair against the f32 matrix pipe: 2.5 PFLOP/s / 3 = 833 TFLOP/s f32-equivalent peak
// vs 157.  The exact-f32 kernel (bigvgan_kernels.hip) stays the default and the parity mode.
//
// Data flow per conv: the activation kernel's f32 [B][C][T] output goes through `split_tm_kernel` (-> two token-major f16
// tensors [B][T][C], rows beyond a row's length zeroed), then `conv_h3_kernel` runs the conv as a GEMM with M = frames,
// N = output channels, K = (tap, input channel): the A operand of K tile (tap j, channels 32 kc ..) is the 128-row window of
// frames shifted by the tap offset, LDS-DMA'd straight from the token-major tensors (rows outside [0, T) come from a zero row);
// the accumulator fragment holds 4 consecutive frames of one output channel per lane, i.e. one 16-byte piece of the
// channel-major f32 output, so bias / residual / MRF accumulate work on the [B][C][T] tensors of the f32 path unchanged.
#include "bigvgan_kernels.h"
#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));

#define H3_BM 128          // frames per block
#define H3_BN 128          // output channels per block
#define H3_BK 32           // input channels per K tile (one MFMA k-step)
#define H3_STAGE 32768     // A_hi 8 KiB | A_lo 8 KiB | W_hi 8 KiB | W_lo 8 KiB
#define H3_LDS (2 * H3_STAGE)

// f32 [B][C][T] -> (hi, lo) f16 [B][T][C]; frames >= the row's length are written as zeros (the conv's zero padding on the right)
// *range_flag (optional) is set when a value is not finite or outside the f16 range (the mode cannot represent it: the caller fails)
__global__ __launch_bounds__(256) void split_tm_kernel(const float* __restrict__ x, u16* __restrict__ xh, u16* __restrict__ xl, int C, int T,
                                                       const int* __restrict__ lens, int len_mult, int* __restrict__ range_flag) {
    __shared__ float tile[64][65];
    const int b = blockIdx.z, c0 = blockIdx.y * 64, t0 = blockIdx.x * 64;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int len = lens ? min(lens[b] * len_mult, T) : T;
    const float* xb = x + (size_t)b * C * T;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + w * 16 + i, t = t0 + lane;
        v[i] = (c < C && t < len) ? xb[(size_t)c * T + t] : 0.f;
    }
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        tile[w * 16 + i][lane] = v[i];
        bad |= !(fabsf(v[i]) < 65504.0f);
    }
    if (range_flag && __any(bad) && lane == 0) atomicOr(range_flag, 1);
    __syncthreads();
    const int t = t0 + (tid >> 2), cq = tid & 3;
    if (t >= T) return;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        const int cb = cq * 16 + hf * 8;
        if (c0 + cb >= C) break;
        v4u32 ph, pl;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            uint32_t hh = 0, ll = 0;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const float f = tile[cb + 2 * e + s][tid >> 2];
                const _Float16 h = fabsf(f) < 6.103515625e-05f ? (_Float16)0.0f : (_Float16)f;     // no subnormal high part: xl carries it
                const _Float16 l = (_Float16)((f - (float)h) * 2048.0f);
                hh |= (uint32_t)__builtin_bit_cast(u16, h) << (16 * s);
                ll |= (uint32_t)__builtin_bit_cast(u16, l) << (16 * s);
            }
            ph[e] = hh; pl[e] = ll;
        }
        const size_t o = ((size_t)b * T + t) * C + c0 + cb;
        *(v4u32*)(xh + o) = ph;
        *(v4u32*)(xl + o) = pl;
    }
}

__global__ __launch_bounds__(256, 2) void conv_h3_kernel(ConvH3Args a) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = w >> 1, wc = w & 1;
    // XCD-aware 1-D mapping as in conv_mfma_kernel: the co tiles of one (row, frame tile) sit on one XCD and share its L2 copy of x
    const int L = blockIdx.x;
    const int xcd = L & 7, slot = L >> 3;
    const int tile = (slot / a.n_co) * 8 + xcd;
    if (tile >= a.n_mt * a.B) return;
    const int b = tile / a.n_mt;
    const int m0 = (tile - b * a.n_mt) * H3_BM;
    const int co0 = (slot % a.n_co) * H3_BN;
    const int len = a.lens ? min(a.lens[b] * a.len_mult, a.T) : a.T;
    if (m0 >= len) return;
    const int nkc = a.Cin / H3_BK, nk = a.k * nkc;
    const int ntiles = (a.Cout + 15) >> 4;
    const int pad = (a.k - 1) / 2 * a.dil;

    // ---- staging sources.  A chunk = 16 frames x 64 bytes, held K-GROUP MAJOR: piece (frame r, k-group q) at q * 256 + r * 16.
    // A ds_read_b128 lane group takes 8 frames of one k-group and the other 8 frames of its neighbour ({0-3, 12-15} / {4-11}),
    // and bank = frame mod 16 -> conflict-free, also when the 16 frames start at any row offset (the window kernel's taps).
    // The DMA writes lane l to byte 16 l of the chunk, so lane l fetches (frame l & 15, k-group l >> 4).
    const int srow = lane & 15, skp = lane >> 4;
    int arow[2];
    const char* bsrc_h[2];
    const char* bsrc_l[2];
    const size_t wstream = (size_t)ntiles * nk * 1024;                  // bytes of one weight stream
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        arow[i] = m0 + (w * 2 + i) * 16 + srow;
        int nt = (co0 >> 4) + w * 2 + i;
        nt = nt < ntiles ? nt : ntiles - 1;
        bsrc_h[i] = (const char*)a.wp + (size_t)nt * nk * 1024 + lane * 16;
        bsrc_l[i] = bsrc_h[i] + wstream;
    }
    const char* xh_b = (const char*)a.xh + ((size_t)b * a.T * a.Cin + skp * 8) * 2;
    const char* xl_b = (const char*)a.xl + ((size_t)b * a.T * a.Cin + skp * 8) * 2;
    const char* zr = (const char*)a.zero_row;
    auto issue = [&](int kt, int buf) {
        char* base = sm + buf * H3_STAGE;
        const int j = kt / nkc, kc = kt - j * nkc;
        const int off = j * a.dil - pad;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int t = arow[i] + off;
            const bool ok = (unsigned)t < (unsigned)a.T;
            const size_t o = ((size_t)(ok ? t : 0) * a.Cin + kc * H3_BK) * 2;
            const char* sh = ok ? xh_b + o : zr;
            const char* sl = ok ? xl_b + o : zr;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sh,
                                             (__attribute__((address_space(3))) void*)(base + (w * 2 + i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sl,
                                             (__attribute__((address_space(3))) void*)(base + 8192 + (w * 2 + i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc_h[i] + (size_t)kt * 1024),
                                             (__attribute__((address_space(3))) void*)(base + 16384 + (w * 2 + i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc_l[i] + (size_t)kt * 1024),
                                             (__attribute__((address_space(3))) void*)(base + 24576 + (w * 2 + i) * 1024), 16, 0, 0);
        }
    };

    f32x4 acc_h[4][4], acc_l[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) { acc_h[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_l[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    // fragment read offsets
    const int row16 = lane & 15, kg = lane >> 4;
    const int a_off = wr * 4 * 1024 + kg * 256 + row16 * 16;
    const int b_off = 16384 + wc * 4 * 1024 + lane * 16;

    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
        const char* base = sm + (kt & 1) * H3_STAGE;
        v4u32 ah[4], al[4], bh[4], bl[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) ah[mt] = *(const v4u32*)(base + a_off + mt * 1024);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bh[nt] = *(const v4u32*)(base + b_off + nt * 1024);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bl[nt] = *(const v4u32*)(base + b_off + 8192 + nt * 1024);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) al[mt] = *(const v4u32*)(base + a_off + 8192 + mt * 1024);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc_h[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah[mt]), __builtin_bit_cast(f16x8, bh[nt]),
                                                                       acc_h[mt][nt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc_l[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah[mt]), __builtin_bit_cast(f16x8, bl[nt]),
                                                                       acc_l[mt][nt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc_l[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, al[mt]), __builtin_bit_cast(f16x8, bh[nt]),
                                                                       acc_l[mt][nt], 0, 0, 0);
    }

    // epilogue: lane (n = lane & 15, q = lane >> 4) of tile (mt, nt) holds frames m0 + (wr*4 + mt)*16 + 4q .. +3 of channel
    // co0 + (wc*4 + nt)*16 + n: one 16-byte piece of the channel-major output row
    const bool vec = (a.T & 3) == 0;
    float* yb = a.y + (size_t)b * a.Cout * a.T;
    const float* rb = a.res ? a.res + (size_t)b * a.Cout * a.T : nullptr;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int co = co0 + (wc * 4 + nt) * 16 + (lane & 15);
        if (co >= a.Cout) continue;
        const float bias = a.bias ? a.bias[co] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int t = m0 + (wr * 4 + mt) * 16 + (lane >> 4) * 4;
            if (t >= len) continue;
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc_h[mt][nt][r] + acc_l[mt][nt][r] * (1.0f / 2048.0f) + bias;
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

// ---- second kernel: (64 MT) frames x (32 NT) channels per block (8 waves as 4 x 2), the A operand held as a WINDOW ----------------
// For one 32-channel chunk kc every tap reads the same frames shifted by j * dil, so the (hi, lo) rows of frames
// m0 - pad .. m0 + BM - 1 + pad are staged ONCE per chunk (double-buffered) and a tap's fragments are read at a row offset; only the
// weights stream per (tap, chunk) K tile, through a 4-deep ring so a tile is requested three K tiles (~4 600 cycles) before it is
// used -- the two-stage pipeline of conv_h3_kernel had one tile of lookahead (~770 cycles) against ~1 500 of L2 latency.
// DMA bytes per 128 frames and chunk at k = 11 (256 x 128 tile): 107 KB against 352 KB.  Needs 3 <= k and (k - 1) * dil <= 48.
// Tile shapes: <4,4> 256 x 128 (the wide stages), <3,6> 192 x 192 and <4,3> 256 x 96 so that 192 / 96 output channels fill the tile.
template <int N>
__device__ __forceinline__ void h3_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int WI>     // DMA instructions per wave and weight stage
__device__ __forceinline__ void h3_wait_allow(int later_w, bool a_recent) {
    if (!a_recent) {
        if (later_w == 0) h3_wait_vm<0>(); else if (later_w == 1) h3_wait_vm<WI>(); else h3_wait_vm<2 * WI>();
    } else {
        if (later_w == 0) h3_wait_vm<6>(); else if (later_w == 1) h3_wait_vm<WI + 6>(); else h3_wait_vm<2 * WI + 6>();
    }
}

template <int MT, int NT>
struct H3W {
    static constexpr int BM = 64 * MT, BN = 32 * NT;
    static constexpr int ACH = (BM + 48) / 16;              // 16-row chunks of the window
    static constexpr int ABUF = 2 * ACH * 1024;             // hi + lo
    static constexpr int NWT = 2 * NT;                      // n-tiles of a weight stage
    static constexpr int WPW = (NWT + 7) / 8;               // n-tiles a wave stages per stream
    static constexpr int WST = 2 * NWT * 1024;              // hi | lo
    static constexpr int LDS = 2 * ABUF + 4 * WST;
};

template <int MT, int NT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_h3w_kernel(ConvH3Args a) {
    using S = H3W<MT, NT>;
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = w >> 1, wc = w & 1;
    const int L = blockIdx.x;
    const int xcd = L & 7, slot = L >> 3;
    const int tile = (slot / a.n_co) * 8 + xcd;
    if (tile >= a.n_mt * a.B) return;
    const int b = tile / a.n_mt;
    const int m0 = (tile - b * a.n_mt) * S::BM;
    const int co0 = (slot % a.n_co) * S::BN;
    const int len = a.lens ? min(a.lens[b] * a.len_mult, a.T) : a.T;
    if (m0 >= len) return;
    const int nkc = a.Cin / H3_BK, G = a.k * nkc;
    const int ntiles = (a.Cout + 15) >> 4;
    const int pad = (a.k - 1) / 2 * a.dil;
    const int nach = (S::BM + 2 * pad + 15) >> 4;                       // row chunks of the window actually needed

    const int srow = lane & 15, skp = lane >> 4;                        // k-group-major chunks, as in conv_h3_kernel
    const char* xh_b = (const char*)a.xh + ((size_t)b * a.T * a.Cin + skp * 8) * 2;
    const char* xl_b = (const char*)a.xl + ((size_t)b * a.T * a.Cin + skp * 8) * 2;
    const char* zr = (const char*)a.zero_row;
    // three window chunks per wave and stream (a wave past the end repeats the last chunk: same bytes to the same place), so every
    // wave issues the same number of DMA instructions and the counted waits below hold for all of them
    auto issue_a = [&](int kc_, int buf) {
        char* base = sm + buf * S::ABUF;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            int c = w + 8 * i;
            c = c < nach ? c : nach - 1;
            const int t = m0 - pad + c * 16 + srow;
            const bool ok = (unsigned)t < (unsigned)a.T;
            const size_t o = ((size_t)(ok ? t : 0) * a.Cin + kc_ * H3_BK) * 2;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ok ? xh_b + o : zr),
                                             (__attribute__((address_space(3))) void*)(base + c * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ok ? xl_b + o : zr),
                                             (__attribute__((address_space(3))) void*)(base + (S::ACH + c) * 1024), 16, 0, 0);
        }
    };
    const char* wsrc_h[S::WPW];
    int wslot[S::WPW];
#pragma unroll
    for (int i = 0; i < S::WPW; ++i) {
        int ntl = w + 8 * i;                                            // n-tile of the stage (a wave past the end repeats the last one)
        ntl = ntl < S::NWT ? ntl : S::NWT - 1;
        int ntg = (co0 >> 4) + ntl;
        ntg = ntg < ntiles ? ntg : ntiles - 1;
        wslot[i] = ntl * 1024;
        wsrc_h[i] = (const char*)a.wp + (size_t)ntg * G * 1024 + lane * 16;
    }
    const size_t wstream = (size_t)ntiles * G * 1024;
    auto issue_w = [&](int g) {                                         // K tile g = (chunk g / k, tap g % k); packed index tap * nkc + chunk
        const int kc_ = g / a.k, j_ = g - kc_ * a.k;
        const size_t kt = (size_t)j_ * nkc + kc_;
        char* base = sm + 2 * S::ABUF + (g & 3) * S::WST;
#pragma unroll
        for (int i = 0; i < S::WPW; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc_h[i] + kt * 1024),
                                             (__attribute__((address_space(3))) void*)(base + wslot[i]), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc_h[i] + wstream + kt * 1024),
                                             (__attribute__((address_space(3))) void*)(base + S::NWT * 1024 + wslot[i]), 16, 0, 0);
        }
    };

    f32x4 acc_h[MT][NT], acc_l[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int jn = 0; jn < NT; ++jn) { acc_h[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_l[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    const int row16 = lane & 15, kg = lane >> 4;
    const int b_off = wc * NT * 1024 + lane * 16;

    issue_a(0, 0);
    issue_w(0);
    if (G > 1) issue_w(1);
    if (G > 2) issue_w(2);
    // (kc, j) of K tile g, and how many iterations ago a window was requested (3 = not within the last two), kept incrementally
    int kc = 0, j = 0, a_age = 3;
    for (int g = 0; g < G; ++g) {
        // wait for weight tile g (and the window of its chunk): everything issued after it may stay in flight
        h3_wait_allow<2 * S::WPW>((G - 1 - g) < 2 ? (G - 1 - g) : 2, a_age <= 2);
        __syncthreads();
        const char* abase = sm + (kc & 1) * S::ABUF;
        const char* wbase = sm + 2 * S::ABUF + (g & 3) * S::WST;
        v4u32 ah[MT], al[MT], bh[NT], bl[NT];
        const int rbase = wr * MT * 16 + row16 + j * a.dil;             // window row of this lane's first frame for tap j
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int rr = rbase + mt * 16;
            ah[mt] = *(const v4u32*)(abase + (rr >> 4) * 1024 + kg * 256 + (rr & 15) * 16);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bh[nt] = *(const v4u32*)(wbase + b_off + nt * 1024);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bl[nt] = *(const v4u32*)(wbase + S::NWT * 1024 + b_off + nt * 1024);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int rr = rbase + mt * 16;
            al[mt] = *(const v4u32*)(abase + S::ACH * 1024 + (rr >> 4) * 1024 + kg * 256 + (rr & 15) * 16);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc_h[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah[mt]), __builtin_bit_cast(f16x8, bh[nt]),
                                                                       acc_h[mt][nt], 0, 0, 0);
        // the prefetches go out behind the first MFMA group: their address arithmetic runs in the matrix pipe's shadow instead of
        // between the barrier and the fragment reads
        if (g + 3 < G) issue_w(g + 3);
        a_age = a_age < 3 ? a_age + 1 : 3;
        if (j == 0 && kc + 1 < nkc) { issue_a(kc + 1, (kc + 1) & 1); a_age = 1; }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc_l[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah[mt]), __builtin_bit_cast(f16x8, bl[nt]),
                                                                       acc_l[mt][nt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc_l[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, al[mt]), __builtin_bit_cast(f16x8, bh[nt]),
                                                                       acc_l[mt][nt], 0, 0, 0);
        if (++j == a.k) { j = 0; ++kc; }
    }

    const bool vec = (a.T & 3) == 0;
    float* yb = a.y + (size_t)b * a.Cout * a.T;
    const float* rb = a.res ? a.res + (size_t)b * a.Cout * a.T : nullptr;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = co0 + (wc * NT + nt) * 16 + (lane & 15);
        if (co >= a.Cout) continue;
        const float bias = a.bias ? a.bias[co] : 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int t = m0 + (wr * MT + mt) * 16 + (lane >> 4) * 4;
            if (t >= len) continue;
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc_h[mt][nt][r] + acc_l[mt][nt][r] * (1.0f / 2048.0f) + bias;
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

template <int MT, int NT>
static int launch_h3w(ConvH3Args a, hipStream_t st) {
    using S = H3W<MT, NT>;
    a.n_mt = ceil_div(a.T, S::BM);
    a.n_co = ceil_div(a.Cout, S::BN);
    const long long tiles8 = ((long long)a.n_mt * a.B + 7) / 8 * 8;
    const long long nblocks = tiles8 * a.n_co;
    if (nblocks > 2147483647ll) { itts_set_error("conv_h3: grid too large"); return ITTS_ERR_ARG; }
    HIP_TRY(hipFuncSetAttribute((const void*)conv_h3w_kernel<MT, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, S::LDS));
    hipLaunchKernelGGL((conv_h3w_kernel<MT, NT>), dim3((unsigned)nblocks), dim3(512), S::LDS, st, a);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

size_t conv_h3_packed_bytes(int Cout, int Cin, int k) {
    return (size_t)2 * ((Cout + 15) / 16) * ((size_t)k * (Cin / H3_BK)) * 1024;
}

// w [Cout][Cin][k] f32 -> two streams of MFMA B fragments: out[s][nt][kt = j * Cin/32 + kc][lane = kg*16 + n][e] =
//   part_s( w[nt*16 + n][kc*32 + kg*8 + e][j] ),  part_0 = f16(w), part_1 = f16(2^11 (w - part_0)); zero rows beyond Cout
int conv_h3_pack(const float* w, int Cout, int Cin, int k, void* out) {
    if (!w || !out || Cout < 1 || Cin < H3_BK || Cin % H3_BK || k < 1 || !(k & 1)) {
        itts_set_error("conv_h3_pack: need C_in %% 32 == 0 and an odd kernel size (Cout=%d Cin=%d k=%d)", Cout, Cin, k);
        return ITTS_ERR_ARG;
    }
    const int ntiles = (Cout + 15) / 16, nkc = Cin / H3_BK;
    const size_t nk = (size_t)k * nkc;
    u16* oh = (u16*)out;
    u16* ol = oh + (size_t)ntiles * nk * 512;
    for (int nt = 0; nt < ntiles; ++nt)
        for (int j = 0; j < k; ++j)
            for (int kc = 0; kc < nkc; ++kc)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int co = nt * 16 + (lane & 15), ci = kc * H3_BK + (lane >> 4) * 8 + e;
                        const float f = co < Cout ? w[((size_t)co * Cin + ci) * k + j] : 0.f;
                        const _Float16 h = (f < 6.103515625e-05f && f > -6.103515625e-05f) ? (_Float16)0.0f : (_Float16)f;
                        const _Float16 l = (_Float16)((f - (float)h) * 2048.0f);
                        if (!(f - f == 0.f) || !((float)h - (float)h == 0.f)) {
                            itts_set_error("conv_h3_pack: weight %g is outside the f16 range", (double)f);
                            return ITTS_ERR_ARG;
                        }
                        const size_t o = (((size_t)nt * nk + (size_t)j * nkc + kc) * 64 + lane) * 8 + e;
                        oh[o] = __builtin_bit_cast(u16, h);
                        ol[o] = __builtin_bit_cast(u16, l);
                    }
    return ITTS_OK;
}

int launch_split_tm(const float* x, void* xh, void* xl, int B, int C, int T, const int* lens, int len_mult, int* range_flag, hipStream_t st) {
    if (B <= 0 || C <= 0 || T <= 0) return ITTS_OK;
    if (C % 8) { itts_set_error("split_tm: C %% 8 != 0"); return ITTS_ERR_ARG; }
    hipLaunchKernelGGL(split_tm_kernel, dim3(ceil_div(T, 64), ceil_div(C, 64), B), dim3(256), 0, st, x, (u16*)xh, (u16*)xl, C, T, lens, len_mult, range_flag);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

int launch_conv_h3(const ConvH3Args& a0, hipStream_t st) {
    if (a0.B <= 0 || a0.T <= 0) return ITTS_OK;
    if (a0.Cin % H3_BK || !(a0.k & 1) || a0.Cout < 1) { itts_set_error("conv_h3: need C_in %% 32 == 0 and odd k"); return ITTS_ERR_ARG; }
    if ((long long)a0.Cout * a0.T >= (1ll << 31) || (long long)a0.Cin * a0.T >= (1ll << 31)) { itts_set_error("conv_h3: row plane too large"); return ITTS_ERR_ARG; }
    // option h3_kernel: 0 = the 128-frame two-stage kernel everywhere, 1 (default) = the window kernel where the taps fit, with the
    // co tile (128 / 192 / 96 wide) that wastes the fewest columns
    const int which = itts_opt(OPT_H3_KERNEL);
    const bool win = which != 0 && a0.k >= 3 && (a0.k - 1) * a0.dil <= 48;     // the counted waits need the window requested >= 3 K tiles ahead
    if (win) {
        auto padded = [&](int bn) { return (long long)ceil_div(a0.Cout, bn) * bn; };
        int best = 128;
        if (which == 1) {
            if (padded(192) < padded(best)) best = 192;
            if (padded(96) < padded(best)) best = 96;
        }
        if (best == 192) return launch_h3w<3, 6>(a0, st);
        if (best == 96) return launch_h3w<4, 3>(a0, st);
        return launch_h3w<4, 4>(a0, st);
    }
    ConvH3Args a = a0;
    a.n_mt = ceil_div(a.T, H3_BM);
    a.n_co = ceil_div(a.Cout, H3_BN);
    const long long tiles8 = ((long long)a.n_mt * a.B + 7) / 8 * 8;
    const long long nblocks = tiles8 * a.n_co;
    if (nblocks > 2147483647ll) { itts_set_error("conv_h3: grid too large"); return ITTS_ERR_ARG; }
    HIP_TRY(hipFuncSetAttribute((const void*)conv_h3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, H3_LDS));
    hipLaunchKernelGGL(conv_h3_kernel, dim3((unsigned)nblocks), dim3(256), H3_LDS, st, a);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}
