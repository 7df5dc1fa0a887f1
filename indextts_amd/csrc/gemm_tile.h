// Device helpers shared by the tile GEMM kernels of gpt_kernels.hip (bf16 / f32 MFMA tiles; built WITH SLP vectorisation: the GPT sampler's
// bit-exact fixtures depend on it, build.py) and of gemm_x3.hip (the fp32x3 tile kernel; built WITHOUT it): tile constants, bf16 / plane
// conversions, and the LDS-transposed vector epilogues (residual, SwiGLU, gate, wqkv + RoPE + Q / K / V^T scatter, WaveNet res / skip).
#pragma once
#include "gpt_kernels.h"
#include <type_traits>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));   // native vector (HIP's uint4 struct defeats SROA in loops)

__device__ __forceinline__ float gelu_new_f(float x) {
    const float c = 0.7978845608028654f;   // sqrt(2/pi)
    return 0.5f * x * (1.0f + tanhf(c * (x + 0.044715f * x * x * x)));
}

#define PF_BM 128
#define PF_BN 128
#define PF_BK 64
#define PF_GM 8
#define PF_LDS 67584     // 2 x (A 16 KiB | W 16 KiB) operand buffers; the epilogue's transposed image [128][132] f32 is the larger
#ifndef PF_SCHED
#define PF_SCHED 1       // explicit LDS-read / MFMA interleave in the tile kernel's main loop (build with -DPF_SCHED=0 for A/B)
#endif

__device__ __forceinline__ uint32_t pf_pack2(float a, float b) { return (uint32_t)f32_to_bf16(a) | ((uint32_t)f32_to_bf16(b) << 16); }

template <int EPI>
__device__ __forceinline__ void pf_epilogue(const GemmArgs& a, int mbase, int nbase, int lane, f32x4 v) {
    const int n = nbase + (lane & 15);
    if (n >= a.N) return;
    const float bias = a.bias ? a.bias[n] : 0.f;
    int which = 0, c = n;
    if constexpr (EPI == EPI_QKV || EPI == EPI_QKV_ROPE) { which = n / a.D; c = n - which * a.D; }
    if constexpr (EPI == EPI_QKV_ROPE) {
        // RoPE pairs are adjacent columns = adjacent lanes: partner value by one xor-shuffle (gpt_fast/model.py:348-360)
        const int hd = c >> 6, d = c & 63, i = d >> 1;
        const bool odd = d & 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = mbase + (lane >> 4) * 4 + r;
            const int mc = m < a.M ? m : a.M - 1;
            const float val = v[r] + bias;
            const float partner = __shfl_xor(val, 1, 64);
            const int s = a.tok_seq[mc], t = a.tok_t[mc];
            float y = val;
            if (which < 2) {
                const float cs = a.rope[((size_t)t * 32 + i) * 2], sn = a.rope[((size_t)t * 32 + i) * 2 + 1];
                y = odd ? val * cs + partner * sn : val * cs - partner * sn;
            }
            if (m < a.M) {
                if (which == 0) ((u16*)a.out_act)[(size_t)m * a.D + c] = f32_to_bf16(y);
                else if (which == 1) ((u16*)a.kcache)[(((size_t)s * a.H + hd) * a.Tmax + t) * 64 + d] = f32_to_bf16(y);
                else ((u16*)a.vcache)[(((size_t)s * a.H + hd) * 64 + d) * a.Tmax + t] = f32_to_bf16(y);
            }
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = mbase + (lane >> 4) * 4 + r;
        if (m >= a.M) continue;
        const float val = v[r] + bias;
        if constexpr (EPI == EPI_STORE_F32) {
            a.out_f32[(size_t)m * a.ldo + n] = val;
            if (a.out_act2) ((u16*)a.out_act2)[(size_t)m * a.ldo + n] = f32_to_bf16(val);
        } else if constexpr (EPI == EPI_RESIDUAL) {
            const float nv = a.out_f32[(size_t)m * a.ldo + n] + val;
            a.out_f32[(size_t)m * a.ldo + n] = nv;
            if (a.out_act2) ((u16*)a.out_act2)[(size_t)m * a.ldo + n] = f32_to_bf16(nv);
        } else if constexpr (EPI == EPI_GELU_ACT) ((u16*)a.out_act)[(size_t)m * a.ldo + n] = f32_to_bf16(gelu_new_f(val));
        else if constexpr (EPI == EPI_WN_RS) {                   // wavenet.py:158-165
            if (a.wn_last || n >= a.D) {
                float* o = a.out2 + (size_t)m * a.D + (a.wn_last ? n : n - a.D);
                *o = a.wn_first ? val : *o + val;
            } else {
                const float mask = a.tok_t[m] < a.seq_len[a.tok_seq[m]] ? 1.f : 0.f;
                float* o = a.out_f32 + (size_t)m * a.D + n;
                *o = (*o + val) * mask;
            }
        } else {                                                   // EPI_QKV
            if (which == 0) {
                a.qbuf[(size_t)m * a.D + c] = val;
            } else {
                const int b = m / a.S, si = m - b * a.S;
                const int pos = *a.pos_ptr + si;
                const size_t o = (((size_t)b * (a.seq_mul > 1 ? a.seq_mul : 1) * a.H + (c >> 6)) * a.Tmax + pos) * 64 + (c & 63);
                ((u16*)(which == 1 ? a.kcache : a.vcache))[o] = f32_to_bf16(val);
            }
        }
    }
}

typedef __bf16 pf_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float pf_f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int v2u_t __attribute__((ext_vector_type(2)));
// v_cvt_pk_bf16_f32 (gfx950): hardware round-to-nearest-even of two f32
__device__ __forceinline__ uint32_t pf_cvt2(float a, float b) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(pf_f32x2_t{a, b}, pf_bf16x2_t)); }
__device__ __forceinline__ v2u_t pf_cvt4(f32x4 v) { return v2u_t{pf_cvt2(v[0], v[1]), pf_cvt2(v[2], v[3])}; }

// f32 x 4 -> three bf16 planes (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m): h + m + l == x exactly), 4 consecutive elements per plane
__device__ __forceinline__ void pf_split4(const f32x4 v, v2u_t& H, v2u_t& M, v2u_t& L) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float a = v[2 * i], b = v[2 * i + 1];
        const uint32_t h = pf_cvt2(a, b);
        const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);      // exact
        const uint32_t m = pf_cvt2(ra, rb);
        const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);    // exact
        H[i] = h; M[i] = m; L[i] = pf_cvt2(sa, sb);
    }
}
// 4 consecutive elements at element index idx of each of the three planes (8-byte aligned) / one element per plane
__device__ __forceinline__ void pf_store_planes4(void* base, size_t stride, size_t idx, const f32x4 v) {
    v2u_t h, m, l;
    pf_split4(v, h, m, l);
    u16* p = (u16*)base + idx;
    *(v2u_t*)p = h; *(v2u_t*)(p + stride) = m; *(v2u_t*)(p + 2 * stride) = l;
}
__device__ __forceinline__ void pf_store_planes1(void* base, size_t stride, size_t idx, float x) {
    const u16 h = f32_to_bf16(x);
    const float r = x - bf16_to_f32(h);
    const u16 m = f32_to_bf16(r);
    u16* p = (u16*)base + idx;
    p[0] = h; p[stride] = m; p[2 * stride] = f32_to_bf16(r - bf16_to_f32(m));
}

// Tile store of the bf16 tile kernel: the 128 x 128 f32 accumulator tile has been transposed through LDS (ct, row-major, the
// 16-float column groups XOR-swizzled by (row >> 2) & 3), so every thread owns 4 CONSECUTIVE columns of a row and the global
// accesses are 16-byte (f32) / 8-byte (bf16) pieces of full lines.  The MFMA accumulator layout itself gives each lane 4 rows
// of one column: 64-byte row segments, half-used lines and read-modify-write at that granularity (measured: the N = 512 f32
// residual GEMMs of the s2mel DiT ran at 1.4 TB/s of output traffic).
// sigmoid / tanh of the fused s2mel epilogues through v_exp_f32 + v_rcp_f32 (about 1 ulp each; the results are rounded to bf16):
// expf + an IEEE division per element made the SwiGLU / gate epilogues as long as a K = 512 main loop.
__device__ __forceinline__ float pf_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float pf_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x)); }
// F32 instantiations of the tile kernel (the s2mel f32 mode: the reference runs this stage with autocast off, infer_v2_5.py:827-828):
// activations, shadows and the K / V^T images are f32, and the gate functions are the libm ones the separate f32 kernels use
// (swiglu_kernel<false>, wn_gate_kernel<false>) -- the main loop is 8x longer per byte than the bf16 one, the epilogue hides under it.
template <bool F32> __device__ __forceinline__ float pf_sigmoid_t(float x) { if constexpr (F32) return 1.0f / (1.0f + expf(-x)); else return pf_sigmoid(x); }
template <bool F32> __device__ __forceinline__ float pf_tanh_t(float x) { if constexpr (F32) return tanhf(x); else return pf_tanh(x); }
template <bool F32> __device__ __forceinline__ void pf_store_act4(void* base, size_t idx, f32x4 v) {      // 4 consecutive act-dtype elements
    if constexpr (F32) *(f32x4*)((float*)base + idx) = v;
    else *(v2u_t*)((u16*)base + idx) = pf_cvt4(v);
}

// Row metadata of a tile region for the epilogues that need the row's (sequence, frame): staged ONCE per region into LDS (meta[row],
// meta[ROWS + row]) by pf_stage_meta -- read per chunk from global they were two dependent L2 round trips in front of every RoPE
// table load / masked store (the wqkv GEMM ran 35 % behind the SwiGLU GEMM of the same K).
//   EPI_QKV_ROPE: (tok_seq, tok_t);  EPI_WN_RS: (tok_t < seq_len[tok_seq] as 0 / 1, unused)
template <int EPI, int ROWS>
__device__ __forceinline__ void pf_stage_meta(const GemmArgs& a, int* meta, int m0, int ltid) {
    if constexpr (EPI == EPI_QKV_ROPE || EPI == EPI_WN_RS) {
        if (ltid < ROWS) {
            int m = m0 + ltid;
            m = m < a.M ? m : a.M - 1;
            const int sq = a.tok_seq[m], t = a.tok_t[m];
            if constexpr (EPI == EPI_QKV_ROPE) { meta[ltid] = sq; meta[ROWS + ltid] = t; }
            else meta[ltid] = (a.wn_last || t < a.seq_len[sq]) ? 1 : 0;
        }
    }
}

// RoPE of two adjacent (even, odd) pairs: y = (v0 c0 - v1 s0, v1 c0 + v0 s0, v2 c1 - v3 s1, v3 c1 + v2 s1), cs = (c0, s0, c1, s1), each component one
// product rounded and one fma -- bit for bit what hipcc's contraction made of the plain expression.  The products pass through an opaque register
// barrier so that the SLP vectoriser (this file is built with it) cannot fuse the four components into v_pk_mul_f32 / v_pk_fma_f32 with op_sel
// operand selection: that packed form -- an IN-PLACE `v_pk_mul_f32 v[n:n+1], v[n:n+1], ... op_sel:[0,1] op_sel_hi:[0,0]` whose low source register
// feeds both halves, followed by the v_pk_fma_f32 that subtracts its low result -- is where the run-to-run differences of the fused wqkv epilogue
// came from: in a solve of 13 layers about one quarter-wave (16 lanes, all of one K / Q row) per two calls stored v2 c1 instead of v2 c1 - v3 s1 in
// exactly the component that in-place product feeds, only while a second block shared the CU (profiles/r05a/capture.log).
__device__ __forceinline__ f32x4 pf_rope4(const f32x4 v, const f32x4 cs) {
    float t0 = v[1] * cs[1], t1 = v[1] * cs[0], t2 = v[3] * cs[3], t3 = v[3] * cs[2];
    asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
    float y0 = __builtin_fmaf(v[0], cs[0], -t0);
    asm volatile("" : "+v"(y0));
    float y1 = __builtin_fmaf(v[0], cs[1], t1);
    asm volatile("" : "+v"(y1));
    float y2 = __builtin_fmaf(v[2], cs[2], -t2);
    asm volatile("" : "+v"(y2));
    float y3 = __builtin_fmaf(v[2], cs[3], t3);
    asm volatile("" : "+v"(y3));
    return f32x4{y0, y1, y2, y3};
}

template <int EPI, int ROWS, int NT, bool F32 = false>            // ROWS x 128 columns of the tile, stored by NT threads (tid < NT)
__device__ __forceinline__ void pf_store_tile(const GemmArgs& a, const float* ct, const int* meta, int m0, int n0, int tid) {
    constexpr bool PAIR = (EPI == EPI_SWIGLU || EPI == EPI_GATE);
    constexpr int CH = PAIR ? 16 : 32;                              // 4-column chunks per tile row
    constexpr int RSTEP = NT / CH;                                  // a thread keeps its column chunk and walks down the rows
    const int half = a.N >> 1;
    const int c4 = tid % CH, row0 = tid / CH;
    // ---- column-derived quantities: once per thread ----
    const int j = c4 >> 2, cc = (c4 & 3) * 4;                       // PAIR: pair j of the region, column inside the 16-wide tile
    const int n = PAIR ? (n0 >> 1) + j * 16 + cc : n0 + c4 * 4;     // output column (inside a half for the pair epilogues)
    if (PAIR ? n >= half : n >= a.N) return;
    const f32x4 zero4{0.f, 0.f, 0.f, 0.f};
    f32x4 b1 = zero4, b2 = zero4;
    if constexpr (EPI == EPI_GATE) {
        b1 = *(const f32x4*)(a.gvec + n);
        b2 = *(const f32x4*)(a.gvec + half + n);
        if (a.bias) {
            const f32x4 x1 = *(const f32x4*)(a.bias + n), x2 = *(const f32x4*)(a.bias + half + n);
#pragma unroll
            for (int q = 0; q < 4; ++q) { b1[q] += x1[q]; b2[q] += x2[q]; }
        }
    } else if constexpr (!PAIR) {
        if (a.bias) b1 = *(const f32x4*)(a.bias + n);
    }
    int which = 0, c = n;
    if constexpr (EPI == EPI_QKV || EPI == EPI_QKV_ROPE) { which = n / a.D; c = n - which * a.D; }
    const int hd = c >> 6, d = c & 63;
    const int ca = PAIR ? (2 * j) * 16 + cc : c4 * 4, cb = (2 * j + 1) * 16 + cc;      // columns inside the LDS image
#pragma unroll 4
    for (int i = 0; i < ROWS / RSTEP; ++i) {
        const int row = row0 + i * RSTEP;
        const int m = m0 + row;
        if (m >= a.M) break;
        const int sw = ((row >> 2) & 3) << 4;
        if constexpr (PAIR) {
            const f32x4 va = *(const f32x4*)(ct + row * 128 + (ca ^ sw));
            const f32x4 vb = *(const f32x4*)(ct + row * 128 + (cb ^ sw));
            f32x4 o;
            if constexpr (EPI == EPI_SWIGLU) {
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = va[q] * pf_sigmoid_t<F32>(va[q]) * vb[q];                       // silu(w1 x) * (w3 x)
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = pf_tanh_t<F32>(va[q] + b1[q]) * pf_sigmoid_t<F32>(vb[q] + b2[q]);      // commons.py:133-141
            }
            pf_store_act4<F32>(a.out_act, (size_t)m * half + n, o);
        } else {
            f32x4 v = *(const f32x4*)(ct + row * 128 + (ca ^ sw));
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] += b1[q];
            if constexpr (EPI == EPI_STORE_F32) {
                *(f32x4*)(a.out_f32 + (size_t)m * a.ldo + n) = v;
                if (a.out_act2) pf_store_act4<F32>(a.out_act2, (size_t)m * a.ldo + n, v);      // act-dtype shadow: the next GEMM's A operand
            } else if constexpr (EPI == EPI_RESIDUAL) {
                f32x4* o = (f32x4*)(a.out_f32 + (size_t)m * a.ldo + n);
                const f32x4 old = *o;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] += old[q];
                *o = v;
                if (a.out_act2) pf_store_act4<F32>(a.out_act2, (size_t)m * a.ldo + n, v);
            } else if constexpr (EPI == EPI_GELU_ACT) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = gelu_new_f(v[q]);
                if constexpr (F32) *(f32x4*)((float*)a.out_act + (size_t)m * a.ldo + n) = v;
                else *(v2u_t*)((u16*)a.out_act + (size_t)m * a.ldo + n) = v2u_t{(uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16),
                                                                               (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16)};
            } else if constexpr (EPI == EPI_WN_RS) {
                if (a.wn_last || n >= a.D) {
                    f32x4* o = (f32x4*)(a.out2 + (size_t)m * a.D + (a.wn_last ? n : n - a.D));
                    if (!a.wn_first) {
                        const f32x4 old = *o;
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] += old[q];
                    }
                    *o = v;
                } else {
                    const float mask = (float)meta[row];
                    f32x4* o = (f32x4*)(a.out_f32 + (size_t)m * a.D + n);
                    const f32x4 old = *o;
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = (old[q] + v[q]) * mask;
                    *o = v;
                    if (a.out_act2) pf_store_act4<F32>(a.out_act2, (size_t)m * a.D + n, v);
                }
            } else if constexpr (EPI == EPI_QKV) {                    // GPT prefill: q f32, K / V appended to the bf16 cache
                if (which == 0) {
                    *(f32x4*)(a.qbuf + (size_t)m * a.D + c) = v;
                } else {
                    const int b = m / a.S, si = m - b * a.S;
                    const int pos = *a.pos_ptr + si;
                    const size_t o = (((size_t)b * (a.seq_mul > 1 ? a.seq_mul : 1) * a.H + hd) * a.Tmax + pos) * 64 + d;
                    if constexpr (F32) *(f32x4*)((float*)(which == 1 ? a.kcache : a.vcache) + o) = v;
                    else *(v2u_t*)((u16*)(which == 1 ? a.kcache : a.vcache) + o) =
                        v2u_t{(uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16), (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16)};
                }
            } else {                                                   // EPI_QKV_ROPE (s2mel): both RoPE pairs of the chunk are in-thread
                const int sq = meta[row], t = meta[ROWS + row];
                if (which < 2) {
                    const f32x4 cs = *(const f32x4*)(a.rope + ((size_t)t * 32 + (d >> 1)) * 2);       // (cos, sin) of pairs d/2, d/2 + 1
                    const f32x4 y = pf_rope4(v, cs);
                    if (which == 0) pf_store_act4<F32>(a.out_act, (size_t)m * a.D + c, y);
                    else if (F32 && a.kv_planes) pf_store_planes4(a.kcache, a.kv_planes, (((size_t)sq * a.H + hd) * a.Tmax + t) * 64 + d, y);
                    else pf_store_act4<F32>(a.kcache, (((size_t)sq * a.H + hd) * a.Tmax + t) * 64 + d, y);
                } else if constexpr (F32) {                            // only when D % 128 != 0 (else pf_store_vt takes the V regions)
                    const size_t o = (((size_t)sq * a.H + hd) * 64 + d) * a.Tmax + t;
                    if (a.kv_planes) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) pf_store_planes1(a.vcache, a.kv_planes, o + (size_t)q * a.Tmax, v[q]);
                    } else {
                        float* vt = (float*)a.vcache + o;
#pragma unroll
                        for (int q = 0; q < 4; ++q) vt[(size_t)q * a.Tmax] = v[q];
                    }
                } else {
                    u16* vt = (u16*)a.vcache + (((size_t)sq * a.H + hd) * 64 + d) * a.Tmax + t;
                    const v2u_t pk = pf_cvt4(v);
                    vt[0] = (u16)(pk.x & 0xffffu);
                    vt[(size_t)a.Tmax] = (u16)(pk.x >> 16);
                    vt[(size_t)2 * a.Tmax] = (u16)(pk.y & 0xffffu);
                    vt[(size_t)3 * a.Tmax] = (u16)(pk.y >> 16);
                }
            }
        }
    }
}

// V^T part of the fused wqkv epilogue (EPI_QKV_ROPE, 128-column tile regions that lie inside the V columns; D % 128 == 0): the
// accumulators were written to LDS TRANSPOSED (ctT [128 columns][ROWS + 4], each lane's 4 consecutive rows = one 16-byte write), so
// a thread owns 4 consecutive frames of one (head, d) row of V^T and a wave's stores walk along t: 8-byte stores when the frame
// run is 4-aligned, 2-byte stores into shared lines otherwise.  (Read from the row-major image the same stores were one 2-byte
// element per line per lane: the wqkv GEMM ran at 425 TFLOP/s against 790 for the SwiGLU GEMM of the same K.)
template <int ROWS, int NT, bool F32 = false>
__device__ __forceinline__ void pf_store_vt(const GemmArgs& a, const float* ctT, int m0, int n0, int tid) {
    constexpr int RQ = ROWS / 4, CSTEP = NT / RQ;                   // a thread keeps its 4-frame run and walks over the columns
    const int rq = tid % RQ, col0 = tid / RQ;
    const int m = m0 + 4 * rq;
    if (m >= a.M) return;
    const int ml = m + 3 < a.M ? m + 3 : a.M - 1;
    const int s0 = a.tok_seq[m], t0 = a.tok_t[m];
    const bool run = m + 3 < a.M && a.tok_seq[ml] == s0;          // rows of one sequence are consecutive frames
    const bool wide = run && (((t0 | a.Tmax) & 3) == 0);
    int sq[4], tq[4];
    if (!run) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int mq = m + q < a.M ? m + q : a.M - 1; sq[q] = a.tok_seq[mq]; tq[q] = a.tok_t[mq]; }
    }
#pragma unroll 4
    for (int i = 0; i < 128 / CSTEP; ++i) {
        const int col = col0 + i * CSTEP;
        const int n = n0 + col;
        if (n >= a.N) break;
        f32x4 v = *(const f32x4*)(ctT + col * (ROWS + 4) + 4 * rq);
        if (a.bias) {
            const float b = a.bias[n];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] += b;
        }
        const int c = n - 2 * a.D, hd = c >> 6, d = c & 63;
        if constexpr (F32) {                                       // f32 V^T image: 16-byte stores along t when the run is 4-aligned
            if (a.kv_planes) {                                     // ... or its three bf16 planes (8-byte stores)
                if (run && wide) {
                    pf_store_planes4(a.vcache, a.kv_planes, (((size_t)s0 * a.H + hd) * 64 + d) * a.Tmax + t0, v);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (m + q < a.M) {
                            const size_t o = run ? (((size_t)s0 * a.H + hd) * 64 + d) * a.Tmax + t0 + q : (((size_t)sq[q] * a.H + hd) * 64 + d) * a.Tmax + tq[q];
                            pf_store_planes1(a.vcache, a.kv_planes, o, v[q]);
                        }
                }
                continue;
            }
            if (run) {
                float* vt = (float*)a.vcache + (((size_t)s0 * a.H + hd) * 64 + d) * a.Tmax + t0;
                if (wide) *(f32x4*)vt = v;
                else { vt[0] = v[0]; vt[1] = v[1]; vt[2] = v[2]; vt[3] = v[3]; }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (m + q < a.M) ((float*)a.vcache)[(((size_t)sq[q] * a.H + hd) * 64 + d) * a.Tmax + tq[q]] = v[q];
            }
            continue;
        }
        const v2u_t pk = pf_cvt4(v);
        if (run) {
            u16* vt = (u16*)a.vcache + (((size_t)s0 * a.H + hd) * 64 + d) * a.Tmax + t0;
            if (wide) {
                *(v2u_t*)vt = pk;
            } else {
                vt[0] = (u16)(pk.x & 0xffffu); vt[1] = (u16)(pk.x >> 16); vt[2] = (u16)(pk.y & 0xffffu); vt[3] = (u16)(pk.y >> 16);
            }
        } else {
            const u16 e[4] = {(u16)(pk.x & 0xffffu), (u16)(pk.x >> 16), (u16)(pk.y & 0xffffu), (u16)(pk.y >> 16)};
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (m + q < a.M) ((u16*)a.vcache)[(((size_t)sq[q] * a.H + hd) * 64 + d) * a.Tmax + tq[q]] = e[q];
        }
    }
}

static inline bool pf_vec_ok(const GemmArgs& a) {
    return (a.N % 4 == 0) && (a.ldo % 4 == 0 || (a.epi != EPI_STORE_F32 && a.epi != EPI_RESIDUAL && a.epi != EPI_GELU_ACT)) && (a.D % 4 == 0);
}
static inline bool pf_f32_ok(const GemmArgs& a) {
    return a.K % 32 == 0 && a.lda % 4 == 0 && a.nsplit == 1 && a.epi != EPI_PARTIAL && pf_vec_ok(a) && (((uintptr_t)a.A) & 15) == 0;
}
