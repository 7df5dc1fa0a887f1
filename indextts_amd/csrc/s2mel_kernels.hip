// s2mel flow-matching decoder kernels for gfx950 (MI355X): everything of the DiT estimator that is not a plain GEMM.
//
// Reference arithmetic replaced (paths relative to the reference repo root):
//   AdaptiveLayerNorm / RMSNorm            indextts/s2mel/modules/gpt_fast/model.py:20-38,322-333
//   apply_rotary_emb + Attention.forward   indextts/s2mel/modules/gpt_fast/model.py:262-307,348-360
//   FeedForward (SwiGLU)                   indextts/s2mel/modules/gpt_fast/model.py:311-319
//   WN (gated dilated convs)               indextts/s2mel/modules/wavenet.py:143-166, commons.py:133-141
//   SConv1d reflect padding                indextts/s2mel/modules/encodec.py:96-113,212-228
//   FinalLayer (LayerNorm + modulate)      indextts/s2mel/modules/diffusion_transformer.py:85-101
//   solve_euler CFG combine + Euler step   indextts/s2mel/modules/flow_matching.py:84-113
//
// The GEMMs run on the kernels of gpt_kernels.hip (bf16 MFMA 128x128 LDS-DMA tiles, or the exact-f32 MFMA path in parity
// mode).  Tokens of all sequences (CFG branches x utterances) are packed into one [n_tok][channels] matrix, so the GEMMs see
// no padding; attention, RoPE, the reflect-padded convs and the masks find a row's sequence through SeqTab.
// head_dim is 64 (hidden 512 / 8 heads in the shipped configuration).
#include <stdlib.h>
#include <type_traits>
#include "s2mel_kernels.h"
#include "gpt_kernels.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// v_cvt_pk_bf16_f32 (gfx950): two f32 -> packed bf16, round-to-nearest-even in hardware (one instruction instead of ~10)
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}

template <bool BF16>
__device__ __forceinline__ void store_act4(void* base, size_t idx, f32x4 v) {      // 4 consecutive elements at element index idx
    if constexpr (BF16) {
        v2u pk{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        *(v2u*)((u16*)base + idx) = pk;
    } else {
        *(f32x4*)((float*)base + idx) = v;
    }
}

// ================================================================================================================
// AdaptiveLayerNorm: out = w * (x * rsqrt(mean(x^2) + eps) * g) + b,  (w | b) = wb[0:H] | wb[H:2H]  (one vector per step:
// the conditioning is the timestep embedding, identical for every row).  One wave per row.
// ================================================================================================================
template <bool BF16>
__global__ __launch_bounds__(256) void ada_rmsnorm_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                          const float* __restrict__ wb, void* __restrict__ out, int n, int H, float eps,
                                                          const int* __restrict__ row_map) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int m = blockIdx.x * 4 + w;
    if (m >= n) return;
    const float* xr = x + (size_t)(row_map ? row_map[m] : m) * H;          // output row m normalises input row row_map[m]
    float ss = 0.f;
    for (int c = lane * 4; c < H; c += 256) {
        const f32x4 v = *(const f32x4*)(xr + c);
        ss += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    ss = wave_sum(ss);
    const float rstd = 1.0f / sqrtf(ss / (float)H + eps);
    for (int c = lane * 4; c < H; c += 256) {
        const f32x4 v = *(const f32x4*)(xr + c), gg = *(const f32x4*)(g + c), ww = *(const f32x4*)(wb + c), bb = *(const f32x4*)(wb + H + c);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = ww[j] * ((v[j] * rstd) * gg[j]) + bb[j];
        store_act4<BF16>(out, (size_t)m * H + c, o);
    }
}

// The same normalisation with the output written as the THREE bf16 planes of every f32 value (h = bf16(y), m = bf16(y - h), l = bf16(y - h - m):
// exact), in the order the f32x3 GEMM's A fragments want them: inside every 32-column group the eight values of k-group kg -- columns 4 kg .. 4 kg + 3
// and 16 + 4 kg .. + 3, the two 16-byte pieces the f32 tile kernel's lane reads -- are contiguous (16 bytes), so gemm_x3_kernel<..., APL = true> stages
// plane tiles by LDS-DMA and reads a fragment with one ds_read_b128 per plane instead of splitting the f32 tile in registers for every column block.
// Plane p of row m at out + p * stride + m * H (u16 elements).  Same arithmetic as ada_rmsnorm_kernel, same planes as x3_split8 -> bitwise the same GEMM.
__global__ __launch_bounds__(256) void ada_rmsnorm_planes_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ wb,
                                                                 u16* __restrict__ out, size_t stride, int n, int H, float eps) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int m = blockIdx.x * 4 + w;
    if (m >= n) return;
    const float* xr = x + (size_t)m * H;
    float ss = 0.f;
    for (int c = lane * 4; c < H; c += 256) {
        const f32x4 v = *(const f32x4*)(xr + c);
        ss += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    ss = wave_sum(ss);
    const float rstd = 1.0f / sqrtf(ss / (float)H + eps);
    for (int c = lane * 4; c < H; c += 256) {
        const f32x4 v = *(const f32x4*)(xr + c), gg = *(const f32x4*)(g + c), ww = *(const f32x4*)(wb + c), bb = *(const f32x4*)(wb + H + c);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = ww[j] * ((v[j] * rstd) * gg[j]) + bb[j];
        v2u hh, mm, ll;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float a0 = o[2 * i], a1 = o[2 * i + 1];
            const uint32_t h = pack_bf16x2(a0, a1);
            const float r0 = a0 - __uint_as_float(h << 16), r1 = a1 - __uint_as_float(h & 0xffff0000u);      // exact
            const uint32_t mi = pack_bf16x2(r0, r1);
            const float s0 = r0 - __uint_as_float(mi << 16), s1 = r1 - __uint_as_float(mi & 0xffff0000u);   // exact
            hh[i] = h; mm[i] = mi; ll[i] = pack_bf16x2(s0, s1);
        }
        const int q = c & 31;
        u16* dst = out + (size_t)m * H + (c & ~31) + (((q & 15) >> 2) << 3) + ((q >> 4) << 2);
        *(v2u*)dst = hh;
        *(v2u*)(dst + stride) = mm;
        *(v2u*)(dst + 2 * stride) = ll;
    }
}

int launch_ada_rmsnorm_planes(const float* x, const float* g, const float* wb, void* out, size_t plane_stride, int n_tok, int H, float eps, hipStream_t st) {
    if (n_tok <= 0) return ITTS_OK;
    if (H % 32 || plane_stride % 8) { itts_set_error("ada_rmsnorm_planes: hidden size %d must be a multiple of 32 and the plane stride of 8", H); return ITTS_ERR_ARG; }
    hipLaunchKernelGGL(ada_rmsnorm_planes_kernel, dim3(ceil_div(n_tok, 4)), dim3(256), 0, st, x, g, wb, (u16*)out, plane_stride, n_tok, H, eps);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

int launch_ada_rmsnorm(const float* x, const float* g, const float* wb, void* out, int n_tok, int H, float eps, int prec, hipStream_t st,
                       const int* row_map) {
    if (n_tok <= 0) return ITTS_OK;
    if (H % 4) { itts_set_error("ada_rmsnorm: hidden size %d must be a multiple of 4", H); return ITTS_ERR_ARG; }
    const dim3 grid(ceil_div(n_tok, 4));
    if (prec == PREC_BF16) hipLaunchKernelGGL(ada_rmsnorm_kernel<true>, grid, dim3(256), 0, st, x, g, wb, out, n_tok, H, eps, row_map);
    else hipLaunchKernelGGL(ada_rmsnorm_kernel<false>, grid, dim3(256), 0, st, x, g, wb, out, n_tok, H, eps, row_map);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

// FinalLayer input and norm: v = wavenet_out * mask + res_projection; LayerNorm(v) (no affine, eps 1e-6) * (1 + scale) + shift,
// (shift | scale) = mod[0:W] | mod[W:2W]  (diffusion_transformer.py:96-100, :248-253)
template <bool BF16>
__global__ __launch_bounds__(256) void final_ln_mod_kernel(const float* __restrict__ wn, const float* __restrict__ rp,
                                                           const float* __restrict__ mod, void* __restrict__ out, SeqTab tab, int W) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int m = blockIdx.x * 4 + w;
    if (m >= tab.n_tok) return;
    const float mask = tab.tok_t[m] < tab.seq_len[tab.tok_seq[m]] ? 1.f : 0.f;
    const float* a = wn + (size_t)m * W;
    const float* b = rp + (size_t)m * W;
    float s = 0.f;
    for (int c = lane * 4; c < W; c += 256) {
        const f32x4 va = *(const f32x4*)(a + c), vb = *(const f32x4*)(b + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) s += va[j] * mask + vb[j];
    }
    const float mean = wave_sum(s) / (float)W;
    float q = 0.f;
    for (int c = lane * 4; c < W; c += 256) {
        const f32x4 va = *(const f32x4*)(a + c), vb = *(const f32x4*)(b + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = (va[j] * mask + vb[j]) - mean; q += d * d; }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)W + 1e-6f);
    for (int c = lane * 4; c < W; c += 256) {
        const f32x4 va = *(const f32x4*)(a + c), vb = *(const f32x4*)(b + c), sh = *(const f32x4*)(mod + c), sc = *(const f32x4*)(mod + W + c);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (((va[j] * mask + vb[j]) - mean) * rstd) * (1.0f + sc[j]) + sh[j];
        store_act4<BF16>(out, (size_t)m * W + c, o);
    }
}

int launch_final_ln_mod(const float* wn_out, const float* rp, const float* mod, void* out, const SeqTab& tab, int W, int prec, hipStream_t st) {
    if (tab.n_tok <= 0) return ITTS_OK;
    if (W % 4) { itts_set_error("final_ln: width %d must be a multiple of 4", W); return ITTS_ERR_ARG; }
    const dim3 grid(ceil_div(tab.n_tok, 4));
    if (prec == PREC_BF16) hipLaunchKernelGGL(final_ln_mod_kernel<true>, grid, dim3(256), 0, st, wn_out, rp, mod, out, tab, W);
    else hipLaunchKernelGGL(final_ln_mod_kernel<false>, grid, dim3(256), 0, st, wn_out, rp, mod, out, tab, W);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

// ================================================================================================================
// f32 -> act dtype with zero padding of the K dimension; output row m reads input row m % src_rows (the CFG branches share x)
// ================================================================================================================
template <bool BF16>
__global__ __launch_bounds__(256) void cast_pad_kernel(const float* __restrict__ in, void* __restrict__ out, int rows, int src_rows,
                                                       int C_in, int C_out, const int* __restrict__ row_map) {
    const int c4n = C_out >> 2;
    const size_t total = (size_t)rows * c4n;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int m = (int)(i / c4n), c = (int)(i - (size_t)m * c4n) * 4;
        const float* r = in + (size_t)((row_map ? row_map[m] : m) % src_rows) * C_in;
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (c + j) < C_in ? r[c + j] : 0.f;
        store_act4<BF16>(out, (size_t)m * C_out + c, v);
    }
}

int launch_cast_pad(const float* in, void* out, int rows, int src_rows, int C_in, int C_out, int prec, hipStream_t st, const int* row_map) {
    if (rows <= 0) return ITTS_OK;
    if (C_out % 4 || C_out < C_in) { itts_set_error("cast_pad: C_out=%d must be a multiple of 4 and >= C_in=%d", C_out, C_in); return ITTS_ERR_ARG; }
    const size_t total = (size_t)rows * (C_out >> 2);
    const unsigned grid = (unsigned)((total + 255) / 256 < 65536 * 8 ? (total + 255) / 256 : 65536 * 8);
    if (prec == PREC_BF16) hipLaunchKernelGGL(cast_pad_kernel<true>, dim3(grid), dim3(256), 0, st, in, out, rows, src_rows, C_in, C_out, row_map);
    else hipLaunchKernelGGL(cast_pad_kernel<false>, dim3(grid), dim3(256), 0, st, in, out, rows, src_rows, C_in, C_out, row_map);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

// ================================================================================================================
// RoPE (interleaved pairs) + split of the fused QKV output.  Block = 4 consecutive frames of one sequence.
//   Q -> [n_tok][H] act;  K -> [seq][head][t_pad][64] act;  V -> V^T [seq][head][64][t_pad] act (4 frames = one 8- / 16-byte
//   store per channel; the flash kernels' PV product wants consecutive KEYS per lane).
// ================================================================================================================
template <bool BF16>
__global__ __launch_bounds__(256) void rope_split_kernel(const float* __restrict__ qkv, const float* __restrict__ rope,
                                                         void* __restrict__ q, void* __restrict__ k, void* __restrict__ v, SeqTab tab,
                                                         int heads, int t_pad) {
    const int s = blockIdx.y, t0 = blockIdx.x * 4;
    const int T = tab.seq_T[s];
    if (t0 >= T) return;
    const int H = heads * 64, tid = threadIdx.x;
    const int m0 = tab.seq_start[s] + t0;
    const int nt = (T - t0) < 4 ? (T - t0) : 4;
    for (int tt = 0; tt < nt; ++tt) {
        const float* row = qkv + (size_t)(m0 + tt) * 3 * H;
        const int t = t0 + tt;
        for (int p = tid; p < H; p += 256) {                      // p < H/2: a pair of q, else a pair of k
            const int which = p >= (H >> 1);
            const int pp = which ? p - (H >> 1) : p;
            const int h = pp >> 5, i = pp & 31, c = h * 64 + 2 * i;
            const float x0 = row[which * H + c], x1 = row[which * H + c + 1];
            const float cs = rope[((size_t)t * 32 + i) * 2], sn = rope[((size_t)t * 32 + i) * 2 + 1];
            const float y0 = x0 * cs - x1 * sn, y1 = x1 * cs + x0 * sn;
            const size_t o = which ? (((size_t)(s * heads + h) * t_pad + t) * 64 + 2 * i) : ((size_t)(m0 + tt) * H + c);
            void* dst = which ? k : q;
            if constexpr (BF16) *(uint32_t*)((u16*)dst + o) = pack_bf16x2(y0, y1);
            else { ((float*)dst)[o] = y0; ((float*)dst)[o + 1] = y1; }
        }
    }
    for (int c = tid; c < H; c += 256) {
        const int h = c >> 6, d = c & 63;
        float val[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) val[tt] = tt < nt ? qkv[(size_t)(m0 + tt) * 3 * H + 2 * H + c] : 0.f;
        if constexpr (BF16) {
            const v2u pk{pack_bf16x2(val[0], val[1]), pack_bf16x2(val[2], val[3])};      // frames past T stay 0
            *(v2u*)((u16*)v + ((size_t)(s * heads + h) * 64 + d) * t_pad + t0) = pk;
        } else {                                                                          // f32: the same V^T image (t0 % 4 == 0, t_pad % 64 == 0)
            *(f32x4*)((float*)v + ((size_t)(s * heads + h) * 64 + d) * t_pad + t0) = f32x4{val[0], val[1], val[2], val[3]};
        }
    }
}

int launch_rope_split(const float* qkv, const float* rope, void* q, void* k, void* v, const SeqTab& tab, int heads, int t_pad, int prec,
                      hipStream_t st) {
    if (tab.n_tok <= 0) return ITTS_OK;
    if (t_pad % 64 || t_pad < tab.t_max) { itts_set_error("rope_split: t_pad=%d must be a multiple of 64 and >= %d", t_pad, tab.t_max); return ITTS_ERR_ARG; }
    const dim3 grid(ceil_div(tab.t_max, 4), tab.n_seq);
    if (prec == PREC_BF16) hipLaunchKernelGGL(rope_split_kernel<true>, grid, dim3(256), 0, st, qkv, rope, q, k, v, tab, heads, t_pad);
    else hipLaunchKernelGGL(rope_split_kernel<false>, grid, dim3(256), 0, st, qkv, rope, q, k, v, tab, heads, t_pad);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

// ================================================================================================================
// Non-causal attention, f32, scalar form: one wave per (query, head); 16 lanes per key, 4 keys per step, online softmax with libm expf.
// Kept as the independent A/B reference of flash_attn_f32_kernel below (ITTS_F32_ATTN=scalar; tests/test_gpu_s2mel.py): 25 TFLOP/s-class,
// 335 ms per Euler step at 8 x 2443 frames (profiles/r03b).  V is the V^T image [seq][head][64][t_pad].
// ================================================================================================================
__global__ __launch_bounds__(64) void attn_f32_kernel(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
                                                      float* __restrict__ O, SeqTab tab, int heads, int t_pad) {
    const int m = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
    const int s = tab.tok_seq[m], len = tab.seq_len[s];
    const int H = heads * 64, sub = lane & 15, grp = lane >> 4;
    const f32x4 q = *(const f32x4*)(Q + (size_t)m * H + h * 64 + sub * 4);
    const float* Kb = K + ((size_t)(s * heads + h) * t_pad) * 64;
    const float* Vb = V + ((size_t)(s * heads + h) * 64) * t_pad;
    float m_run = -INFINITY, l_run = 0.f;
    f32x4 acc{0.f, 0.f, 0.f, 0.f};
    for (int t0 = 0; t0 < len; t0 += 4) {
        const int t = t0 + grp;
        const bool ok = t < len;
        const int tc = ok ? t : len - 1;
        const f32x4 kf = *(const f32x4*)(Kb + (size_t)tc * 64 + sub * 4);
        const f32x4 vf{Vb[(size_t)(sub * 4 + 0) * t_pad + tc], Vb[(size_t)(sub * 4 + 1) * t_pad + tc], Vb[(size_t)(sub * 4 + 2) * t_pad + tc],
                       Vb[(size_t)(sub * 4 + 3) * t_pad + tc]};
        float sc = (q[0] * kf[0] + q[1] * kf[1]) + (q[2] * kf[2] + q[3] * kf[3]);
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) sc += __shfl_xor(sc, o, 64);
        sc *= 0.125f;
        if (ok) {
            const float nm = fmaxf(m_run, sc);
            const float al = expf(m_run - nm), p = expf(sc - nm);
            l_run = l_run * al + p;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = acc[j] * al + p * vf[j];
            m_run = nm;
        }
    }
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) {
        const float om = __shfl_xor(m_run, o, 64), ol = __shfl_xor(l_run, o, 64);
        const float nm = fmaxf(m_run, om);
        const float sa = (m_run == -INFINITY) ? 0.f : expf(m_run - nm), sb = (om == -INFINITY) ? 0.f : expf(om - nm);
        l_run = l_run * sa + ol * sb;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = acc[j] * sa + __shfl_xor(acc[j], o, 64) * sb;
        m_run = nm;
    }
    if (lane < 16) {
        f32x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = l_run > 0.f ? acc[j] / l_run : 0.f;
        *(f32x4*)(O + (size_t)m * H + h * 64 + sub * 4) = r;
    }
}

// ================================================================================================================
// Non-causal flash attention on v_mfma_f32_16x16x32_bf16.  Block = 64 queries of one (sequence, head): 4 waves x 16 queries.
//   S^T = K Q^T   : A = K tile (rows = keys, k = d: 8 consecutive d per lane = row-major K), B = Q^T (lane: its query's 8 d)
//                   -> C layout: lane (g, q) holds keys 16*kt + 4g + r, r < 4, of query q = lane & 15.
//   softmax       : per query = per lane column; the reduction over keys is 16 in-lane values and two xor-shuffles (16, 32).
//   O^T = V^T P^T : A = V^T tile (rows = d, k = keys), B = P^T straight from the S^T registers -- no transpose through LDS:
//                   the ROWS of the S^T MFMA tiles are a permutation of the keys (row 4a + c of sub-tile kt is key
//                   32 (kt >> 1) + 8 a + 4 (kt & 1) + c; a K fragment lane simply reads that key's row), chosen so that lane g's
//                   eight P values of half kx are keys 32 kx + 8 g .. + 7 in order = the standard B fragment, and the V^T fragment
//                   is ONE 16-byte read.  (A first version kept the natural key order and read V^T as two 8-byte runs: every
//                   such ds_read2_b64 took 16 LDS cycles, half of them bank conflicts -- all of the kernel's conflicts, PMC
//                   ablation in profiles/r02s.)
//   K tile [64 keys][64 d] and V^T tile [64 d][64 keys] are staged in LDS: 128-byte rows, the eight 16-byte pieces of a row
//   XOR-permuted -- V^T row r by (r >> 1) & 7, K row r by ((r >> 1) & 1) | (((r >> 3) & 3) << 1) (the sixteen key rows a K fragment
//   touches are {0-3, 8-11, 16-19, 24-27} + const) -- conflict-free for the staging writes and both 16-byte fragment reads under
//   the real lane grouping of ds_read_b128 (4 groups of 16 non-contiguous lanes).  A padded 144-byte stride (the usual
//   recipe) measured 43 % of all LDS cycles as bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, profiles/r02l).
//   The next tile's global loads are in flight while the current one feeds the MFMAs.
// ================================================================================================================
#define FA_LD 64          // bf16 elements per LDS row (128 B, pieces swizzled)
// Three measured choices are built in (A/B and ablation variants live in the frozen copy under tools/microbench/ablate_src/, which
// tools/microbench/flash_ablate.hip builds): the LDS stage is a compile-time constant (the key loop is unrolled by two: immediate offsets instead
// of address adds); the accumulators are rescaled only when some query's running maximum moved (wave-uniform branch); the row sums of P ride on
// the PV MFMAs (a fifth m-tile of ones: the normaliser is the sum of the bf16 probabilities the PV product uses).  Same-box A/B at 64 x 2443 frames
// x 8 heads, two sub-tiles: none 1284 us, stage + rescale 1200, all three 1180 (160 registers).

// QS = 16-query sub-tiles per wave (block = 64 * QS queries).  Every K / V^T fragment read from LDS feeds QS MFMAs: with one
// sub-tile the kernel is LDS-bound (16 KB of fragment reads per 16 MFMAs per wave), with four the MFMA pipe is the limit.
// max over the lanes l, l ^ 16, l ^ 32, l ^ 48 (the four key groups of one query column) on the VALU: v_permlane16_swap /
// v_permlane32_swap (gfx950) exchange 16- / 32-lane halves between two registers.  The shuffle form (__shfl_xor -> ds_bpermute)
// put four LDS-pipe round trips per key tile on the softmax's critical path, queued behind the V^T fragment reads.
__device__ __forceinline__ float fa_colmax(float x) {
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1\n\tv_max_f32 %0, %0, %1\n\tv_mov_b32 %1, %0\n\ts_nop 1\n\t"
                 "v_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_max_f32 %0, %0, %1"
                 : "+v"(a), "+v"(b));
    return a;
}
// max of three as compiler-generated code (one v_max3_f32).  NOT inline asm: in the kernels' non-tail path this is the first reader of S^T
// accumulators written one to three MFMAs earlier, and hipcc's hazard recognizer pads only instructions it generated itself ("XDL write VGPR ->
// VALU read": s_nop 7 in front of its own consumer, nothing in front of an asm statement -- tools/microbench/mfma_asm_hazard.hip).  The asm form
// read accumulators inside that window: a row maximum taken from stale registers still gives a valid softmax (the maximum is only a shift), but
// one whose rounding depends on how the waves of a SIMD happened to interleave.
__device__ __forceinline__ float fa_max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

template <int QS>
__global__ __launch_bounds__(256, (QS == 1 ? 4 : QS == 2 ? 2 : 1)) void flash_attn_bf16_kernel(const u16* __restrict__ Q, const u16* __restrict__ K, const u16* __restrict__ Vt,
                                                              u16* __restrict__ O, SeqTab tab, int heads, int t_pad, float scale_log2e) {
    __shared__ __attribute__((aligned(16))) u16 ks[2][64 * FA_LD];     // two stages: one barrier per key tile
    __shared__ __attribute__((aligned(16))) u16 vs[2][64 * FA_LD];
    const int s = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * (64 * QS);
    const int T = tab.seq_T[s], len = tab.seq_len[s];
    if (q0 >= T) return;
    const int H = heads * 64;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, c16 = lane & 15;
    const size_t row0 = (size_t)tab.seq_start[s];
    v4u qf[QS][2];
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) {
        int qi = q0 + (w * QS + qs) * 16 + c16;
        qi = qi < T ? qi : T - 1;
        const u16* qrow = Q + (row0 + qi) * H + h * 64;
        qf[qs][0] = *(const v4u*)(qrow + g * 8);
        qf[qs][1] = *(const v4u*)(qrow + 32 + g * 8);
    }
    const u16* Kb = K + ((size_t)(s * heads + h) * t_pad) * 64;
    const u16* Vb = Vt + ((size_t)(s * heads + h) * 64) * t_pad;
    const int r0 = tid >> 3, p0 = tid & 7;                         // staging: chunk tid + 256 i -> row r0 + 32 i, 16-byte piece p0
    v4u kreg[2], vreg[2];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = r0 + 32 * i;
            kreg[i] = *(const v4u*)(Kb + (size_t)(k0 + r) * 64 + p0 * 8);
            vreg[i] = *(const v4u*)(Vb + (size_t)r * t_pad + k0 + p0 * 8);
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = r0 + 32 * i;
            const int pk = (p0 ^ (((r >> 1) & 1) | (((r >> 3) & 3) << 1))) * 8;      // r and r + 32 share both swizzles
            const int pv = (p0 ^ ((r >> 1) & 7)) * 8;
            *(v4u*)(ks[buf] + r * FA_LD + pk) = kreg[i];
            *(v4u*)(vs[buf] + r * FA_LD + pv) = vreg[i];
        }
    };
    // fragment read offsets (elements) inside a stage.  K: lane row c16 = 4 a + c of sub-tile kt reads key row 8 a + c (+ 32 (kt >> 1)
    // + 4 (kt & 1): immediates); V^T: row c16 of a 16-row group.
    const int krow = 8 * (c16 >> 2) + (c16 & 3);
    const int ksw = ((c16 >> 1) & 1) | ((c16 >> 2) << 1), vsw = (c16 >> 1) & 7;
    int k_off[2], v_off[2];
#pragma unroll
    for (int kx = 0; kx < 2; ++kx) {
        k_off[kx] = krow * FA_LD + (((kx * 4 + g) ^ ksw) << 3);
        v_off[kx] = c16 * FA_LD + (((kx * 4 + g) ^ vsw) << 3);                                  // keys 32 kx + 8 g .. + 7
    }
    f32x4 o[QS][4], osum[QS];                                      // osum: row 0 of a fifth V^T m-tile of ones = the row sums of P
    float m_run[QS];
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) {
        m_run[qs] = -INFINITY;
        osum[qs] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) o[qs][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const unsigned one2 = c16 == 0 ? 0x3f803f80u : 0u;              // bf16 1.0 pairs in row 0 of the ones tile
    const v4u ones_frag{one2, one2, one2, one2};
    fetch(0);
    stage(0);
    __syncthreads();
    // pin the Q fragments here: hipcc otherwise sinks their loads to the loop's doorstep and then has to guard their first use
    // INSIDE the loop with s_waitcnt vmcnt(0) -- which also drains the next key tile's prefetch every iteration
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) {
        asm volatile("" : "+v"(qf[qs][0]));
        asm volatile("" : "+v"(qf[qs][1]));
    }
    // One key tile; the LDS stage is a compile-time constant (the loop is unrolled by two).
    auto tile = [&](auto bufc, int k0) {
        const int BUF = bufc;
        const bool more = k0 + 64 < len;                           // block-uniform
        if (more) fetch(k0 + 64);                                  // in flight under this tile's MFMAs
        const u16* kt_s = ks[BUF];
        const u16* vt_s = vs[BUF];
        f32x4 st[QS][4];
        // S^T: all four key sub-tiles with the first 32 d, then the second 32 d -- consecutive MFMAs never chain on one accumulator
#pragma unroll
        for (int kx = 0; kx < 2; ++kx)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const v4u a = *(const v4u*)(kt_s + (32 * (kt >> 1) + 4 * (kt & 1)) * FA_LD + k_off[kx]);
#pragma unroll
                for (int qs = 0; qs < QS; ++qs)
                    st[qs][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, qf[qs][kx]),
                                                                         kx == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : st[qs][kt], 0, 0, 0);
            }
        const bool tail = k0 + 64 > len;                           // block-uniform: only the last tile holds masked keys
        v4u pb[QS][2];
#pragma unroll
        for (int qs = 0; qs < QS; ++qs) {
            if (tail) {                                            // uniform branch: full tiles carry no per-element select
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if ((k0 + 32 * (kt >> 1) + 8 * g + 4 * (kt & 1) + r) >= len) st[qs][kt][r] = -INFINITY;
            }
            // softmax in the exp2 domain on the RAW scores: p = exp2(s * c - m * c), c = log2(e) / 8, one fma per element;
            // in-lane max by v_max3, across the query's four lanes by permlane swaps
            const float t0 = fa_max3(st[qs][0][0], st[qs][0][1], st[qs][0][2]), t1 = fa_max3(st[qs][1][0], st[qs][1][1], st[qs][1][2]);
            const float t2 = fa_max3(st[qs][2][0], st[qs][2][1], st[qs][2][2]), t3 = fa_max3(st[qs][3][0], st[qs][3][1], st[qs][3][2]);
            const float u0 = fa_max3(t0, t1, st[qs][0][3]), u1 = fa_max3(t2, t3, st[qs][1][3]);
            const float m_new = fa_colmax(fa_max3(u0, u1, fa_max3(st[qs][2][3], st[qs][3][3], m_run[qs])));   // >= m_run, finite
            if (__builtin_amdgcn_ballot_w64(m_new > m_run[qs]) != 0) {                       // wave-uniform: skipped when no maximum moved
                const float alpha = __builtin_amdgcn_exp2f((m_run[qs] - m_new) * scale_log2e);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[qs][mt][r] *= alpha;
                osum[qs][0] *= alpha;
                m_run[qs] = m_new;
            }
            // plain f32 VALU on purpose (the file is built with -fno-slp-vectorize): beside MFMAs a v_pk_fma/mul/add_f32 costs ~13
            // cycles more than the two scalar ops it replaces (MI355X guide)
            const float off = -m_new * scale_log2e;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) st[qs][kt][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[qs][kt][r], scale_log2e, off));
#pragma unroll
            for (int kx = 0; kx < 2; ++kx)
                pb[qs][kx] = v4u{pack_bf16x2(st[qs][2 * kx][0], st[qs][2 * kx][1]), pack_bf16x2(st[qs][2 * kx][2], st[qs][2 * kx][3]),
                                 pack_bf16x2(st[qs][2 * kx + 1][0], st[qs][2 * kx + 1][1]), pack_bf16x2(st[qs][2 * kx + 1][2], st[qs][2 * kx + 1][3])};
        }
#pragma unroll
        for (int kx = 0; kx < 2; ++kx) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const v4u a = *(const v4u*)(vt_s + mt * 16 * FA_LD + v_off[kx]);
#pragma unroll
                for (int qs = 0; qs < QS; ++qs)
                    o[qs][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, pb[qs][kx]),
                                                                        o[qs][mt], 0, 0, 0);
            }
#pragma unroll
            for (int qs = 0; qs < QS; ++qs)
                osum[qs] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ones_frag), __builtin_bit_cast(bf16x8_t, pb[qs][kx]),
                                                                   osum[qs], 0, 0, 0);
        }
        if (more) stage(BUF ^ 1);                                  // the other stage was last read one tile ago (barrier below that tile)
        __syncthreads();
    };
    for (int k0 = 0; k0 < len; k0 += 128) {
        tile(std::integral_constant<int, 0>{}, k0);
        if (k0 + 64 >= len) break;
        tile(std::integral_constant<int, 1>{}, k0 + 64);
    }
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) {
        const float ls = __shfl(osum[qs][0], c16, 64);             // row 0 of the ones tile lives in lanes 0..15 (g = 0), element 0
        const int qi = q0 + (w * QS + qs) * 16 + c16;
        if (qi < T) {
            const float inv = ls > 0.f ? 1.0f / ls : 0.f;
            u16* orow = O + (row0 + qi) * H + h * 64;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const v2u pk{pack_bf16x2(o[qs][mt][0] * inv, o[qs][mt][1] * inv), pack_bf16x2(o[qs][mt][2] * inv, o[qs][mt][3] * inv)};
                *(v2u*)(orow + mt * 16 + g * 4) = pk;
            }
        }
    }
}

// ================================================================================================================
// Non-causal flash attention in exact f32 on v_mfma_f32_16x16x4_f32 (the s2mel f32 mode: the reference runs the DiT with autocast off,
// infer_v2_5.py:827-828; gpt_fast/model.py:262-307).  Block = 64 queries of one (sequence, head), 4 waves x 16 queries; key tiles of 64.
//   S^T = K Q^T   : A = K fragment (lane (g, r): key row 16 kt + r, d = 16 ks + 4 g + j -- one 16-byte LDS read feeds MFMAs j = 0..3),
//                   B = Q^T from registers (lane (g, q): the same four d of its query) -> C: lane (g, q) holds keys 16 kt + 4 g + r.
//   O^T = V^T P^T : A = V^T fragment (lane (g, r): row d = 16 dt + r, keys 16 kt + 4 g + j: again one 16-byte read per four MFMAs),
//                   B = P^T = the S^T accumulators themselves: MFMA j of key sub-tile kt contracts over the keys {16 kt + 4 g + j}_g,
//                   and lane (g, q) holds exactly that key's probability in st[kt][j] -- no transpose, no LDS round trip.
//   MFMAs are issued j-outer over four independent accumulators (dependent latency 40 cycles, issue 32).  Per key tile and wave:
//   128 MFMAs (4096 cycles) against 32 fragment reads and ~60 VALU softmax instructions: matrix-pipe bound (157 TFLOP/s peak).
//   K tile [64 keys][64 d] and V^T tile [64 d][64 keys] f32 = 16 KiB each, two stages (64 KiB: two blocks per CU), filled by LDS-DMA
//   (global_load_lds_dwordx4: a wave instruction writes four 256-byte rows lane-linearly); the sixteen 16-byte pieces of row r are
//   XOR-permuted by r & 15 on the SOURCE side, so a fragment read (piece 4 ks + g of rows 16 kt + r) is conflict-free under
//   ds_read_b128's real lane groups ({0-3, 12-15, 20-27}, ...: aligned pairs / quads of r within one g, MI355X guide).
//   softmax in the exp2 domain on the raw scores (f32 v_exp_f32, ~1 ulp), running max across the query's four lanes by permlane swaps.
// ================================================================================================================
#define FA32_STAGE 32768
#define FA32_LDS (2 * FA32_STAGE)
// QS = 16-query sub-tiles per wave (block = 64 QS queries): every K / V^T fragment read and every LDS-DMA piece of a key tile then feeds
// QS times the MFMAs (an LDS-DMA piece costs 60-185 cycles of issue beside MFMAs, MI355X guide: 8 pieces per wave against 4096 MFMA
// cycles at QS = 1).
template <int QS>
__global__ __launch_bounds__(256, 2) void flash_attn_f32_kernel(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ Vt,
                                                                float* __restrict__ O, SeqTab tab, int heads, int t_pad, float scale_log2e) {
    extern __shared__ __attribute__((aligned(16))) char fa32_sm[];     // [2][K 16 KiB | V^T 16 KiB]
    const int s = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * (64 * QS);
    const int T = tab.seq_T[s], len = tab.seq_len[s];
    if (q0 >= T) return;
    const int H = heads * 64;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, c16 = lane & 15;
    const size_t row0 = (size_t)tab.seq_start[s];
    int qi[QS];
    bool q_ok[QS];
    f32x4 qf[QS][4];
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) {
        qi[qs] = q0 + (w * QS + qs) * 16 + c16;
        q_ok[qs] = qi[qs] < T;
        qi[qs] = q_ok[qs] ? qi[qs] : T - 1;
        const float* qrow = Q + (row0 + qi[qs]) * H + h * 64 + g * 4;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[qs][ks] = *(const f32x4*)(qrow + ks * 16);
    }
    const float* Kb = K + ((size_t)(s * heads + h) * t_pad) * 64;
    const float* Vb = Vt + ((size_t)(s * heads + h) * 64) * t_pad;
    // DMA sources of this lane: wave w stages 1 KiB chunks 4 w .. 4 w + 3 of each operand = rows 16 w + 4 i + (lane >> 4)
    const float* ksrc[4];
    const float* vsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 16 * w + 4 * i + g;
        const int piece = c16 ^ (r & 15);
        ksrc[i] = Kb + (size_t)r * 64 + piece * 4;
        vsrc[i] = Vb + (size_t)r * t_pad + piece * 4;
    }
    auto issue = [&](int k0, int buf) {
        char* base = fa32_sm + buf * FA32_STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ksrc[i] + (size_t)k0 * 64),
                                             (__attribute__((address_space(3))) void*)(base + (w * 4 + i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vsrc[i] + k0),
                                             (__attribute__((address_space(3))) void*)(base + 16384 + (w * 4 + i) * 1024), 16, 0, 0);
        }
    };
    // fragment read offsets: row (16 x + c16) * 256 B + slot ((4 y + g) ^ c16) * 16 B; x = sub-tile (kt / dt), y = k-step (ks / kt)
    int f_off[4];
#pragma unroll
    for (int y = 0; y < 4; ++y) f_off[y] = c16 * 256 + (((4 * y + g) ^ c16) << 4);
    f32x4 o[QS][4];
    float m_run[QS], l_run[QS];
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) {
        m_run[qs] = -INFINITY;
        l_run[qs] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[qs][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    issue(0, 0);
    int it = 0;
    for (int k0 = 0; k0 < len; k0 += 64, ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                               // tile `it` is in LDS; everybody is done with tile it - 1
        if (k0 + 64 < len) issue(k0 + 64, (it + 1) & 1);               // block-uniform; in flight under this tile's MFMAs
        const char* kt_s = fa32_sm + (it & 1) * FA32_STAGE;
        const char* vt_s = kt_s + 16384;
        f32x4 st[QS][4];
#pragma unroll
        for (int qs = 0; qs < QS; ++qs)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) st[qs][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            f32x4 a[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) a[kt] = *(const f32x4*)(kt_s + kt * 4096 + f_off[ks]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int qs = 0; qs < QS; ++qs)
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
                        st[qs][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kt][j], qf[qs][ks][j], st[qs][kt], 0, 0, 0);
        }
        const bool tail = k0 + 64 > len;                               // block-uniform: only the last tile holds masked keys
#pragma unroll
        for (int qs = 0; qs < QS; ++qs) {
            if (tail) {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (k0 + 16 * kt + 4 * g + r >= len) st[qs][kt][r] = -INFINITY;
            }
            const float t0 = fa_max3(st[qs][0][0], st[qs][0][1], st[qs][0][2]), t1 = fa_max3(st[qs][1][0], st[qs][1][1], st[qs][1][2]);
            const float t2 = fa_max3(st[qs][2][0], st[qs][2][1], st[qs][2][2]), t3 = fa_max3(st[qs][3][0], st[qs][3][1], st[qs][3][2]);
            const float u0 = fa_max3(t0, t1, st[qs][0][3]), u1 = fa_max3(t2, t3, st[qs][1][3]);
            const float m_new = fa_colmax(fa_max3(u0, u1, fa_max3(st[qs][2][3], st[qs][3][3], m_run[qs])));      // >= m_run, finite (key 0 is valid)
            if (__builtin_amdgcn_ballot_w64(m_new > m_run[qs]) != 0) {
                const float alpha = __builtin_amdgcn_exp2f((m_run[qs] - m_new) * scale_log2e);             // exp2(-inf) = 0 on the first tile
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[qs][dt][r] *= alpha;
                l_run[qs] *= alpha;
                m_run[qs] = m_new;
            }
            const float off = -m_new * scale_log2e;
            float psum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(st[qs][kt][r], scale_log2e, off));
                    st[qs][kt][r] = p;
                    psum += p;
                }
            l_run[qs] += psum;
        }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            f32x4 a[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) a[dt] = *(const f32x4*)(vt_s + dt * 4096 + f_off[kt]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int qs = 0; qs < QS; ++qs)
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt)
                        o[qs][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[dt][j], st[qs][kt][j], o[qs][dt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) {
        float ls = l_run[qs];                                          // the four key groups' shares of the row sum
        ls += __shfl_xor(ls, 16, 64);
        ls += __shfl_xor(ls, 32, 64);
        if (q_ok[qs]) {
            const float inv = ls > 0.f ? 1.0f / ls : 0.f;
            float* orow = O + (row0 + qi[qs]) * H + h * 64 + g * 4;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                *(f32x4*)(orow + dt * 16) = f32x4{o[qs][dt][0] * inv, o[qs][dt][1] * inv, o[qs][dt][2] * inv, o[qs][dt][3] * inv};
        }
    }
}

// ================================================================================================================
// Non-causal flash attention of the fp32x3 mode: f32-accurate S = Q K^T and O = P V on the bf16 matrix pipe, every f32 operand carried EXACTLY
// as three bf16 planes (x = h + m + l, 8 + 8 + 8 significand bits; the scheme of gpt_kernels.hip::gemm_x3_kernel) and every f32 product
// replaced by NPROD plane products accumulated in f32: 8 = every term down to 2^-24 |ab| (hh, hm, mh, hl, lh, mm, ml, lm), 6 drops ml and lm.
//   K and V^T arrive as planes (written once by the wqkv GEMM's epilogue, GemmArgs::kv_planes: plane p of K at Kp + p * pstride in the bf16
//   mode's [seq][head][t_pad][64] image, of V^T in its [seq][head][64][t_pad] image); Q (f32 rows) is split by the wave that owns the
//   queries, once, before the key loop; P -- the f32 probabilities in the S^T accumulators -- is split in registers per key tile (11 VALU
//   ops per pair) and goes straight into the B operand of the PV MFMAs, as in flash_attn_bf16_kernel (same permutation of the key rows
//   of the S^T sub-tiles, same LDS images and swizzles per plane).  Softmax statistics, the running maximum, the row sums of the f32 P and the
//   output accumulators are f32 (the f32 flash kernel's arithmetic).
//   Block = 256 queries of one (sequence, head): 8 waves x 2 sub-tiles of 16 queries; key tiles of 64.  Per key tile and wave 2 x 16 x NPROD
//   MFMAs (v_mfma_f32_16x16x32_bf16) against 24 fragment reads (each feeds 2 x NPROD / 3 .. MFMAs) and ~300 VALU instructions.
//   LDS: two stages of [K h | K m | K l | V^T h | V^T m | V^T l] x 8 KiB = 96 KiB (one block = two waves per SIMD), filled by LDS-DMA: wave w
//   stages rows 8 w .. 8 w + 7 of each plane tile with one instruction, the 16-byte pieces XOR-permuted on the source side.
// ================================================================================================================
#define FX3_PLANE 8192
#define FX3_STAGE (6 * FX3_PLANE)
#define FX3_LDS (2 * FX3_STAGE)

__device__ __forceinline__ void fx3_split8(const f32x4 p0, const f32x4 p1, v4u& H, v4u& M, v4u& L) {
    const float x[8] = {p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = x[2 * i], b = x[2 * i + 1];
        const uint32_t h = pack_bf16x2(a, b);
        const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);      // exact
        const uint32_t m = pack_bf16x2(ra, rb);
        const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);    // exact
        H[i] = h; M[i] = m; L[i] = pack_bf16x2(sa, sb);
    }
}

template <int NPROD>
__global__ __launch_bounds__(512) void flash_attn_x3_kernel(const float* __restrict__ Q, const u16* __restrict__ Kp, const u16* __restrict__ Vp,
                                                            size_t pstride, float* __restrict__ O, SeqTab tab, int heads, int t_pad, float scale_log2e) {
    extern __shared__ __attribute__((aligned(16))) char fx3_sm[];
    const int s = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 256;
    const int T = tab.seq_T[s], len = tab.seq_len[s];
    if (q0 >= T) return;
    const int H = heads * 64;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: LDS-DMA destinations (M0) without v_readfirstlane
    const int g = lane >> 4, c16 = lane & 15;
    const size_t row0 = (size_t)tab.seq_start[s];
    // plane pairs (A = K or V^T plane, B = Q or P plane), smallest terms first; NPROD = 6 skips the two 2^-24 cross terms ml, lm
    constexpr int PA[8] = {2, 1, 2, 0, 1, 1, 0, 0};
    constexpr int PB[8] = {1, 2, 0, 2, 1, 0, 1, 0};
    int qi[2];
    bool q_ok[2];
    v4u qp[2][3][2];                                               // [sub-tile][plane][d half]: d = 32 kx + 8 g .. + 7 of query c16
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
        qi[qs] = q0 + (w * 2 + qs) * 16 + c16;
        q_ok[qs] = qi[qs] < T;
        qi[qs] = q_ok[qs] ? qi[qs] : T - 1;
        const float* qrow = Q + (row0 + qi[qs]) * H + h * 64 + g * 8;
#pragma unroll
        for (int kx = 0; kx < 2; ++kx) {
            const f32x4 p0 = *(const f32x4*)(qrow + 32 * kx), p1 = *(const f32x4*)(qrow + 32 * kx + 4);
            fx3_split8(p0, p1, qp[qs][0][kx], qp[qs][1][kx], qp[qs][2][kx]);
        }
    }
    const u16* Kb = Kp + ((size_t)(s * heads + h) * t_pad) * 64;
    const u16* Vb = Vp + ((size_t)(s * heads + h) * 64) * t_pad;
    // LDS-DMA sources of this lane: row 8 w + (lane >> 3) of a plane tile, LDS slot lane & 7 <- global piece slot ^ swizzle(row)
    const int rr = 8 * w + (lane >> 3), slot = lane & 7;
    const u16* ksrc = Kb + (size_t)rr * 64 + ((slot ^ (((rr >> 1) & 1) | (((rr >> 3) & 3) << 1))) << 3);
    const u16* vsrc = Vb + (size_t)rr * t_pad + ((slot ^ ((rr >> 1) & 7)) << 3);
    auto issue = [&](int k0, int buf) {
        char* base = fx3_sm + buf * FX3_STAGE + w * 1024;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ksrc + p * pstride + (size_t)k0 * 64),
                                             (__attribute__((address_space(3))) void*)(base + p * FX3_PLANE), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vsrc + p * pstride + k0),
                                             (__attribute__((address_space(3))) void*)(base + (3 + p) * FX3_PLANE), 16, 0, 0);
        }
    };
    // fragment read offsets (bytes) inside a plane tile: K lane row c16 = 4 a + c of sub-tile kt reads key row 8 a + c (+ 32 (kt >> 1) + 4 (kt & 1));
    // V^T: row c16 of a 16-row group (flash_attn_bf16_kernel's images)
    const int krow = 8 * (c16 >> 2) + (c16 & 3);
    const int ksw = ((c16 >> 1) & 1) | ((c16 >> 2) << 1), vsw = (c16 >> 1) & 7;
    int k_off[2], v_off[2];
#pragma unroll
    for (int kx = 0; kx < 2; ++kx) {
        k_off[kx] = (krow * 64 + (((kx * 4 + g) ^ ksw) << 3)) * 2;
        v_off[kx] = (c16 * 64 + (((kx * 4 + g) ^ vsw) << 3)) * 2;       // keys 32 kx + 8 g .. + 7
    }
    f32x4 o[2][4];
    float m_run[2], l_run[2];
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
        m_run[qs] = -INFINITY;
        l_run[qs] = 0.f;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) o[qs][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    issue(0, 0);
    int it = 0;
    for (int k0 = 0; k0 < len; k0 += 64, ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                               // tile `it` is in LDS; everybody is done with tile it - 1
        if (k0 + 64 < len) issue(k0 + 64, (it + 1) & 1);               // block-uniform; in flight under this tile's MFMAs
        const char* kt_s = fx3_sm + (it & 1) * FX3_STAGE;
        const char* vt_s = kt_s + 3 * FX3_PLANE;
        f32x4 st[2][4];
#pragma unroll
        for (int qs = 0; qs < 2; ++qs)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) st[qs][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        // S^T = K Q^T: two key sub-tiles at a time (their three plane fragments in registers), products q outermost so that consecutive
        // MFMAs never chain on one accumulator (four independent ones per product)
#pragma unroll
        for (int kx = 0; kx < 2; ++kx)
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                v4u ka[2][3];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int kt = 2 * kp + j;
#pragma unroll
                    for (int p = 0; p < 3; ++p) ka[j][p] = *(const v4u*)(kt_s + p * FX3_PLANE + (32 * (kt >> 1) + 4 * (kt & 1)) * 128 + k_off[kx]);
                }
#pragma unroll
                for (int q = 8 - NPROD; q < 8; ++q)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int qs = 0; qs < 2; ++qs)
                            st[qs][2 * kp + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ka[j][PA[q]]),
                                                                                         __builtin_bit_cast(bf16x8_t, qp[qs][PB[q]][kx]), st[qs][2 * kp + j], 0, 0, 0);
            }
        const bool tail = k0 + 64 > len;                               // block-uniform: only the last tile holds masked keys
        v4u pb[2][3][2];                                               // P planes: [sub-tile][plane][key half]
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) {
            if (tail) {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if ((k0 + 32 * (kt >> 1) + 8 * g + 4 * (kt & 1) + r) >= len) st[qs][kt][r] = -INFINITY;
            }
            const float t0 = fa_max3(st[qs][0][0], st[qs][0][1], st[qs][0][2]), t1 = fa_max3(st[qs][1][0], st[qs][1][1], st[qs][1][2]);
            const float t2 = fa_max3(st[qs][2][0], st[qs][2][1], st[qs][2][2]), t3 = fa_max3(st[qs][3][0], st[qs][3][1], st[qs][3][2]);
            const float u0 = fa_max3(t0, t1, st[qs][0][3]), u1 = fa_max3(t2, t3, st[qs][1][3]);
            const float m_new = fa_colmax(fa_max3(u0, u1, fa_max3(st[qs][2][3], st[qs][3][3], m_run[qs])));      // >= m_run, finite (key 0 is valid)
            if (__builtin_amdgcn_ballot_w64(m_new > m_run[qs]) != 0) {
                const float alpha = __builtin_amdgcn_exp2f((m_run[qs] - m_new) * scale_log2e);             // exp2(-inf) = 0 on the first tile
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[qs][mt][r] *= alpha;
                l_run[qs] *= alpha;
                m_run[qs] = m_new;
            }
            const float off = -m_new * scale_log2e;
            float psum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(st[qs][kt][r], scale_log2e, off));
                    st[qs][kt][r] = p;
                    psum += p;
                }
            l_run[qs] += psum;
#pragma unroll
            for (int kx = 0; kx < 2; ++kx) fx3_split8(st[qs][2 * kx], st[qs][2 * kx + 1], pb[qs][0][kx], pb[qs][1][kx], pb[qs][2][kx]);
        }
        // O^T += V^T P^T
#pragma unroll
        for (int kx = 0; kx < 2; ++kx)
#pragma unroll
            for (int mp = 0; mp < 2; ++mp) {
                v4u va[2][3];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int p = 0; p < 3; ++p) va[j][p] = *(const v4u*)(vt_s + p * FX3_PLANE + (2 * mp + j) * 2048 + v_off[kx]);
#pragma unroll
                for (int q = 8 - NPROD; q < 8; ++q)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int qs = 0; qs < 2; ++qs)
                            o[qs][2 * mp + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, va[j][PA[q]]),
                                                                                        __builtin_bit_cast(bf16x8_t, pb[qs][PB[q]][kx]), o[qs][2 * mp + j], 0, 0, 0);
            }
    }
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
        float ls = l_run[qs];                                          // the four key groups' shares of the row sum
        ls += __shfl_xor(ls, 16, 64);
        ls += __shfl_xor(ls, 32, 64);
        if (q_ok[qs]) {
            const float inv = ls > 0.f ? 1.0f / ls : 0.f;
            float* orow = O + (row0 + qi[qs]) * H + h * 64 + g * 4;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                *(f32x4*)(orow + mt * 16) = f32x4{o[qs][mt][0] * inv, o[qs][mt][1] * inv, o[qs][mt][2] * inv, o[qs][mt][3] * inv};
        }
    }
}

// Order-independent 64-bit checksum of a buffer (diagnostics: itts_s2mel_set_trace): sum over the 32-bit words of word * (odd function of the word
// index), accumulated with integer atomics -- associative, so the value does not depend on which wave adds first.
__global__ __launch_bounds__(256) void trace_hash_kernel(const uint32_t* __restrict__ p, size_t n_words, unsigned long long* __restrict__ out) {
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_words; i += (size_t)gridDim.x * 256)
        acc += (unsigned long long)p[i] * (2 * (unsigned long long)i + 0x9E3779B97F4A7C15ull | 1ull);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}
int launch_trace_hash(const void* p, size_t bytes, unsigned long long* out, hipStream_t st) {
    const size_t n = bytes / 4;
    if (!n) return ITTS_OK;
    const unsigned blocks = (unsigned)((n + 256 * 16 - 1) / (256 * 16) < 2048 ? (n + 256 * 16 - 1) / (256 * 16) : 2048);
    hipLaunchKernelGGL(trace_hash_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, st, (const uint32_t*)p, n, out);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

// f32 [n] -> three bf16 planes (plane p at out + p * n): the unit-level entry point's path to the x3 attention operands
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ in, u16* __restrict__ out, size_t n) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    if (i >= n) return;
    const float a = in[i], b = in[i + 1];
    const uint32_t h = pack_bf16x2(a, b);
    const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
    const uint32_t m = pack_bf16x2(ra, rb);
    const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);
    *(uint32_t*)(out + i) = h;
    *(uint32_t*)(out + n + i) = m;
    *(uint32_t*)(out + 2 * n + i) = pack_bf16x2(sa, sb);
}
int launch_split_planes(const float* in, void* out, size_t n, hipStream_t st) {
    if (n == 0) return ITTS_OK;
    if (n & 1) { itts_set_error("split_planes: n must be even"); return ITTS_ERR_ARG; }
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, st, in, (u16*)out, n);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

// q f32 [n_tok][H]; kp / vp: three bf16 planes each (plane stride = n_seq * heads * t_pad * 64 elements); out f32 [n_tok][H]
int launch_s2mel_attention_x3(const void* q, const void* kp, const void* vp, void* out, const SeqTab& tab, int heads, int t_pad, hipStream_t st) {
    if (tab.n_tok <= 0) return ITTS_OK;
    static ItPerDevice<bool> attr_set_pd;
    bool& attr_set = attr_set_pd.cur();
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)flash_attn_x3_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, FX3_LDS));
        HIP_TRY(hipFuncSetAttribute((const void*)flash_attn_x3_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, FX3_LDS));
        attr_set = true;
    }
    const float scale_log2e = 0.125f * 1.4426950408889634f;
    const size_t pstride = (size_t)tab.n_seq * heads * t_pad * 64;
    const dim3 grid(ceil_div(tab.t_max, 256), heads, tab.n_seq);
    if (itts_opt(OPT_X3_PRODUCTS) == 8)
        hipLaunchKernelGGL(flash_attn_x3_kernel<8>, grid, dim3(512), FX3_LDS, st, (const float*)q, (const u16*)kp, (const u16*)vp, pstride, (float*)out, tab, heads, t_pad, scale_log2e);
    else
        hipLaunchKernelGGL(flash_attn_x3_kernel<6>, grid, dim3(512), FX3_LDS, st, (const float*)q, (const u16*)kp, (const u16*)vp, pstride, (float*)out, tab, heads, t_pad, scale_log2e);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

int launch_s2mel_attention(const void* q, const void* k, const void* v, void* out, const SeqTab& tab, int heads, int t_pad, int prec,
                           hipStream_t st) {
    if (tab.n_tok <= 0) return ITTS_OK;
    if (prec == PREC_BF16) {
        const float scale_log2e = 0.125f * 1.4426950408889634f;     // 1 / sqrt(64) * log2(e)
        // query sub-tiles per wave (option fa_qs forces a variant).  One sub-tile is LDS-bound (every K / V^T fragment read feeds one
        // MFMA: SQ_LDS_IDX_ACTIVE at 70-90 % of the kernel's cycles, profiles/r02l), two halve the fragment traffic per MFMA and
        // still fit three waves per SIMD; four need 256 registers (one wave per SIMD) and lose to latency.
        const int force_qs = itts_opt(OPT_FA_QS);
        const int qs = force_qs ? force_qs : 2;
#define FA_LAUNCH(QS_) hipLaunchKernelGGL(flash_attn_bf16_kernel<QS_>, dim3(ceil_div(tab.t_max, 64 * QS_), heads, tab.n_seq), dim3(256), 0, st, \
                                          (const u16*)q, (const u16*)k, (const u16*)v, (u16*)out, tab, heads, t_pad, scale_log2e)
        if (qs >= 4) FA_LAUNCH(4); else if (qs == 2) FA_LAUNCH(2); else FA_LAUNCH(1);
#undef FA_LAUNCH
    } else {
        const bool scalar = itts_opt(OPT_F32_ATTN_SCALAR) != 0;      // the one-wave-per-query A/B reference
        if (scalar) {
            hipLaunchKernelGGL(attn_f32_kernel, dim3(tab.n_tok, heads), dim3(64), 0, st, (const float*)q, (const float*)k, (const float*)v, (float*)out, tab, heads, t_pad);
        } else {
            const int qs = itts_opt(OPT_FA32_QS) == 1 ? 1 : 2;
            static ItPerDevice<bool> attr_set_pd;
    bool& attr_set = attr_set_pd.cur();
            if (!attr_set) {
                HIP_TRY(hipFuncSetAttribute((const void*)flash_attn_f32_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, FA32_LDS));
                HIP_TRY(hipFuncSetAttribute((const void*)flash_attn_f32_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, FA32_LDS));
                attr_set = true;
            }
            const float scale_log2e = 0.125f * 1.4426950408889634f;
            if (qs == 2)
                hipLaunchKernelGGL(flash_attn_f32_kernel<2>, dim3(ceil_div(tab.t_max, 128), heads, tab.n_seq), dim3(256), FA32_LDS, st, (const float*)q,
                                   (const float*)k, (const float*)v, (float*)out, tab, heads, t_pad, scale_log2e);
            else
                hipLaunchKernelGGL(flash_attn_f32_kernel<1>, dim3(ceil_div(tab.t_max, 64), heads, tab.n_seq), dim3(256), FA32_LDS, st, (const float*)q,
                                   (const float*)k, (const float*)v, (float*)out, tab, heads, t_pad, scale_log2e);
        }
    }
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

// ================================================================================================================
// element-wise pieces
// ================================================================================================================
template <bool BF16>
__global__ __launch_bounds__(256) void swiglu_kernel(const float* __restrict__ in, void* __restrict__ out, int n, int I) {
    const int i4n = I >> 2;
    const size_t total = (size_t)n * i4n;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int m = (int)(i / i4n), c = (int)(i - (size_t)m * i4n) * 4;
        const f32x4 a = *(const f32x4*)(in + (size_t)m * 2 * I + c), b = *(const f32x4*)(in + (size_t)m * 2 * I + I + c);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (a[j] / (1.0f + expf(-a[j]))) * b[j];
        store_act4<BF16>(out, (size_t)m * I + c, o);
    }
}

int launch_swiglu(const float* in, void* out, int n_tok, int I, int prec, hipStream_t st) {
    if (n_tok <= 0) return ITTS_OK;
    if (I % 4) { itts_set_error("swiglu: width %d must be a multiple of 4", I); return ITTS_ERR_ARG; }
    const size_t total = (size_t)n_tok * (I >> 2);
    const unsigned grid = (unsigned)((total + 255) / 256 < 65536 * 8 ? (total + 255) / 256 : 65536 * 8);
    if (prec == PREC_BF16) hipLaunchKernelGGL(swiglu_kernel<true>, dim3(grid), dim3(256), 0, st, in, out, n_tok, I);
    else hipLaunchKernelGGL(swiglu_kernel<false>, dim3(grid), dim3(256), 0, st, in, out, n_tok, I);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

// SConv1d reflect padding of (k-1)*d frames, left = total - total/2 (encodec.py:212-228); a sequence not longer than the pad is
// zero-extended on the right first (pad1d, :96-113): source index outside [0, T) after reflection reads 0.
__device__ __forceinline__ int reflect_src(int p, int Tv) {       // Tv = virtual (zero-extended) length
    if (p < 0) p = -p;
    if (p >= Tv) p = 2 * (Tv - 1) - p;
    return p;
}

template <bool BF16>
__global__ __launch_bounds__(256) void im2col_reflect_kernel(const float* __restrict__ x, void* __restrict__ col, SeqTab tab, int W, int k,
                                                             int dil) {
    const int w4n = W >> 2;
    const size_t total = (size_t)tab.n_tok * k * w4n;
    const int pad_total = (k - 1) * dil, right = pad_total / 2, left = pad_total - right;
    const int maxpad = left > right ? left : right;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % w4n) * 4;
        const size_t mj = i / w4n;
        const int j = (int)(mj % k), m = (int)(mj / k);
        const int s = tab.tok_seq[m], t = tab.tok_t[m], T = tab.seq_T[s];
        const int Tv = T <= maxpad ? maxpad + 1 : T;
        const int src = reflect_src(t + j * dil - left, Tv);
        f32x4 v{0.f, 0.f, 0.f, 0.f};
        if (src >= 0 && src < T) v = *(const f32x4*)(x + (size_t)(tab.seq_start[s] + src) * W + c);
        store_act4<BF16>(col, ((size_t)m * k + j) * W + c, v);
    }
}

int launch_im2col_reflect(const float* x, void* col, const SeqTab& tab, int W, int k, int dilation, int prec, hipStream_t st) {
    if (tab.n_tok <= 0) return ITTS_OK;
    if (W % 4 || k < 1 || dilation < 1) { itts_set_error("im2col: bad W=%d k=%d dilation=%d", W, k, dilation); return ITTS_ERR_ARG; }
    const size_t total = (size_t)tab.n_tok * k * (W >> 2);
    const unsigned grid = (unsigned)((total + 255) / 256 < 65536 * 8 ? (total + 255) / 256 : 65536 * 8);
    if (prec == PREC_BF16) hipLaunchKernelGGL(im2col_reflect_kernel<true>, dim3(grid), dim3(256), 0, st, x, col, tab, W, k, dilation);
    else hipLaunchKernelGGL(im2col_reflect_kernel<false>, dim3(grid), dim3(256), 0, st, x, col, tab, W, k, dilation);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

template <bool BF16>
__global__ __launch_bounds__(256) void wn_gate_kernel(const float* __restrict__ in, const float* __restrict__ gvec, void* __restrict__ out, int n, int W) {
    const int w4n = W >> 2;
    const size_t total = (size_t)n * w4n;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int m = (int)(i / w4n), c = (int)(i - (size_t)m * w4n) * 4;
        const f32x4 a = *(const f32x4*)(in + (size_t)m * 2 * W + c), b = *(const f32x4*)(in + (size_t)m * 2 * W + W + c);
        const f32x4 ga = *(const f32x4*)(gvec + c), gb = *(const f32x4*)(gvec + W + c);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = tanhf(a[j] + ga[j]) * (1.0f / (1.0f + expf(-(b[j] + gb[j]))));
        store_act4<BF16>(out, (size_t)m * W + c, o);
    }
}

int launch_wn_gate(const float* in, const float* g, void* out, int n_tok, int W, int prec, hipStream_t st) {
    if (n_tok <= 0) return ITTS_OK;
    const size_t total = (size_t)n_tok * (W >> 2);
    const unsigned grid = (unsigned)((total + 255) / 256 < 65536 * 8 ? (total + 255) / 256 : 65536 * 8);
    if (prec == PREC_BF16) hipLaunchKernelGGL(wn_gate_kernel<true>, dim3(grid), dim3(256), 0, st, in, g, out, n_tok, W);
    else hipLaunchKernelGGL(wn_gate_kernel<false>, dim3(grid), dim3(256), 0, st, in, g, out, n_tok, W);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

__global__ __launch_bounds__(256) void wn_update_kernel(const float* __restrict__ rs, float* __restrict__ x, float* __restrict__ out, SeqTab tab,
                                                        int W, int first, int last) {
    const int w4n = W >> 2;
    const size_t total = (size_t)tab.n_tok * w4n;
    const int ld = last ? W : 2 * W;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int m = (int)(i / w4n), c = (int)(i - (size_t)m * w4n) * 4;
        const float* r = rs + (size_t)m * ld;
        f32x4 sk;
        if (!last) {
            const float mask = tab.tok_t[m] < tab.seq_len[tab.tok_seq[m]] ? 1.f : 0.f;
            const f32x4 res = *(const f32x4*)(r + c);
            f32x4 xv = *(const f32x4*)(x + (size_t)m * W + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) xv[j] = (xv[j] + res[j]) * mask;
            *(f32x4*)(x + (size_t)m * W + c) = xv;
            sk = *(const f32x4*)(r + W + c);
        } else {
            sk = *(const f32x4*)(r + c);
        }
        if (!first) {
            const f32x4 ov = *(const f32x4*)(out + (size_t)m * W + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) sk[j] += ov[j];
        }
        *(f32x4*)(out + (size_t)m * W + c) = sk;
    }
}

int launch_wn_update(const float* rs, float* x, float* out, const SeqTab& tab, int W, int first, int last, hipStream_t st) {
    if (tab.n_tok <= 0) return ITTS_OK;
    const size_t total = (size_t)tab.n_tok * (W >> 2);
    const unsigned grid = (unsigned)((total + 255) / 256 < 65536 * 8 ? (total + 255) / 256 : 65536 * 8);
    hipLaunchKernelGGL(wn_update_kernel, dim3(grid), dim3(256), 0, st, rs, x, out, tab, W, first, last);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

// solver state xs [n_tok / n_branch][C] (the cond branch's rows); d [n_tok][C] estimator output, null branch in the second half
// tail_base (optional): d holds the estimator output only for the TAIL rows of every sequence (capi_s2mel.hip, dead-row elimination) --
// frame t of sequence s is row tail_base[s] + t of d, the null branch's rows follow tail_half rows later; prompt frames are never read.
__global__ __launch_bounds__(256) void euler_update_kernel(float* __restrict__ xs, const float* __restrict__ d, SeqTab tab,
                                                           const int* __restrict__ prompt_len, int C, int n_branch, float dt, float cfg_rate,
                                                           const int* __restrict__ tail_base, int tail_half) {
    const int n_half = tab.n_tok / n_branch;
    const size_t total = (size_t)n_half * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int m = (int)(i / C);
        const int sq = tab.tok_seq[m], t = tab.tok_t[m];
        float v = 0.f;
        if (t >= prompt_len[sq]) {
            const size_t j = tail_base ? (size_t)(tail_base[sq] + t) * C + (i - (size_t)m * C) : i;
            float dphi = d[j];
            if (n_branch == 2) dphi = (1.0f + cfg_rate) * dphi - cfg_rate * d[j + (tail_base ? (size_t)tail_half * C : total)];
            v = xs[i] + dt * dphi;
        }
        xs[i] = v;
    }
}

int launch_euler_update(float* xs, const float* d, const SeqTab& tab, const int* prompt_len, int C, int n_branch, float dt, float cfg_rate,
                        hipStream_t st, const int* tail_base, int tail_half) {
    if (tab.n_tok <= 0) return ITTS_OK;
    const size_t total = (size_t)(tab.n_tok / n_branch) * C;
    const unsigned grid = (unsigned)((total + 255) / 256 < 65536 * 8 ? (total + 255) / 256 : 65536 * 8);
    hipLaunchKernelGGL(euler_update_kernel, dim3(grid), dim3(256), 0, st, xs, d, tab, prompt_len, C, n_branch, dt, cfg_rate, tail_base, tail_half);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}
