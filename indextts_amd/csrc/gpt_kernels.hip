// GPT-2 speech-token decoder kernels for gfx950 (MI355X).
//
// Reference arithmetic replaced (paths relative to the reference repo root):
//   GPT2Block / GPT2Attention / GPT2MLP     indextts/gpt/transformers_gpt2.py:591-667,129-348,571-585
//   KV concat per step (torch.cat)          indextts/gpt/transformers_gpt2.py:325-328  -> in-place cache append
//   ln_f + lm_head = final_norm o mel_head  indextts/gpt/model_v2.py:54,186
//   logits processors + token selection     indextts/gpt/transformers_generation_utils.py:900-901,1035-1044,3220-3256
//   decode-step embedding + position rule   indextts/gpt/model_v2.py:158-161
//
// Two precisions share every kernel: PREC_F32 (parity mode: f32 weights/KV, v_mfma_f32_16x16x4_f32 = exact f32)
// and PREC_BF16 (weights/KV/GEMM inputs bf16, v_mfma_f32_16x16x32_bf16, f32 accumulate).  The residual stream,
// LayerNorm statistics, softmax and logits are f32 in both.
//
// Decode is HBM-bound (weights once per step + B x KV): weights are pre-packed in MFMA B-fragment order so each
// wave-level load is one contiguous 1 KiB (16 B/lane); a block's 4 waves split K and reduce through LDS so even
// N = 1280 gives >= 240 blocks (with 4 K-slices); K/V rows are read as 16 B per lane (8 or 16 lanes per key).
#include "gemm_tile.h"


// ================================================================================================================
// LayerNorm (+ fused split-K reduce, bias, residual write-back, optional second LayerNorm)
// one wave per row, NE = D/64 elements per lane kept in registers
// ================================================================================================================
// Every operand of a row (x, the 4 split-K partials, the previous GEMM's bias, both affine pairs) is fetched in ONE
// phase of 16-byte loads issued back to back: the kernel is a latency chain (launch -> loads -> 2-4 wave reductions ->
// stores), so each extra dependent load phase costs a full L2/HBM round trip.  Which operands exist is a template
// parameter (FLAGS) -- a runtime branch around a group of loads makes hipcc wait for the previous group first.
//   lane holds elements 4*lane + 256*i + {0..3}, i < NV = D/256.
enum { LN_PARTIAL = 1, LN_BIAS = 2, LN_SECOND = 4 };

// One row by one wave: loads, split-K reduce / bias, (optional) residual write-back to `xw`, one or two normalisations; leaves the result in v.
// Shared by ln_kernel and the LayerNorm-fused decode GEMM (gemm_decode_ln_kernel) so both produce the same bits.
template <int NV, int FLAGS>
__device__ __forceinline__ void ln_row(const float* __restrict__ xr, float* __restrict__ xw, const float* __restrict__ pp, size_t ps,
                                       const float* __restrict__ bias_prev, const float* __restrict__ ga1, const float* __restrict__ be1,
                                       const float* __restrict__ ga2, const float* __restrict__ be2, int D, float eps, int lane, f32x4 (&v)[NV]) {
    // Contraction is spelled out (hipcc's default, fp-contract=fast, lets the backend fuse a * b + c or not depending on the surrounding
    // code: the same source gave 23 FMAs inside ln_kernel and 72 inside the fused GEMM, i.e. last-bit differences between the two).  The
    // forms below are the ones ln_kernel compiled to before they were pinned: squares as products then sequential adds, fma for
    // sum * (1 / D) + eps and for t * gamma + beta.
#pragma clang fp contract(off)
    f32x4 p0[NV], p1[NV], p2[NV], p3[NV], bp[NV], g1[NV], b1[NV], g2[NV], b2[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = 4 * lane + 256 * i;
        v[i] = *(const f32x4*)(xr + e);
        if constexpr (FLAGS & LN_PARTIAL) {
            p0[i] = *(const f32x4*)(pp + e); p1[i] = *(const f32x4*)(pp + ps + e);
            p2[i] = *(const f32x4*)(pp + 2 * ps + e); p3[i] = *(const f32x4*)(pp + 3 * ps + e);
        }
        if constexpr (FLAGS & LN_BIAS) bp[i] = *(const f32x4*)(bias_prev + e);
        g1[i] = *(const f32x4*)(ga1 + e);
        b1[i] = *(const f32x4*)(be1 + e);
        if constexpr (FLAGS & LN_SECOND) { g2[i] = *(const f32x4*)(ga2 + e); b2[i] = *(const f32x4*)(be2 + e); }
    }
    __builtin_amdgcn_sched_barrier(0);          // keep hipcc from sinking the affine loads below the reductions
    if constexpr (FLAGS & LN_PARTIAL) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[i][j] += (p0[i][j] + p1[i][j]) + (p2[i][j] + p3[i][j]);
    }
    if constexpr (FLAGS & LN_BIAS) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[i][j] += bp[i][j];
    }
    if constexpr (FLAGS & (LN_PARTIAL | LN_BIAS)) {
        if (xw) {
#pragma unroll
            for (int i = 0; i < NV; ++i) *(f32x4*)(xw + 4 * lane + 256 * i) = v[i];
        }
    }
    const float invD = 1.0f / (float)D;
    auto normalise = [&](const f32x4 (&g)[NV], const f32x4 (&bb)[NV]) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        const float mean = wave_sum(s) * invD;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; q += d * d; }
        const float rstd = 1.0f / sqrtf(__builtin_fmaf(wave_sum(q), invD, eps));
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[i][j] = __builtin_fmaf((v[i][j] - mean) * rstd, g[i][j], bb[i][j]);
    };
    normalise(g1, b1);
    if constexpr (FLAGS & LN_SECOND) normalise(g2, b2);
}

template <int NV, bool OUT_BF16, int FLAGS>
__global__ __launch_bounds__(256) void ln_kernel(LnArgs a) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = blockIdx.x * (int)(blockDim.x >> 6) + w;
    if (r >= a.rows) return;
    const int D = a.D;
    const size_t in_r = (size_t)r * a.in_row_mul + a.in_row_add;
    float* xr = a.x + in_r * D;
    f32x4 v[NV];
    ln_row<NV, FLAGS>(xr, xr, a.partial + (size_t)r * D, (size_t)a.rows * D, a.bias_prev, a.g1, a.b1, a.g2, a.b2, D, a.eps, lane, v);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const size_t o = (size_t)r * D + 4 * lane + 256 * i;
        if (OUT_BF16 && !a.out_f32) {
            uint2 pk;
            pk.x = (uint32_t)f32_to_bf16(v[i][0]) | ((uint32_t)f32_to_bf16(v[i][1]) << 16);
            pk.y = (uint32_t)f32_to_bf16(v[i][2]) | ((uint32_t)f32_to_bf16(v[i][3]) << 16);
            *(uint2*)((u16*)a.out + o) = pk;
        } else {
            *(f32x4*)((float*)a.out + o) = v[i];
        }
    }
}

template <int NV, bool OUT_BF16>
static void launch_ln_flags(const LnArgs& a, dim3 grid, hipStream_t st) {
    const int flags = (a.partial ? LN_PARTIAL : 0) | (a.bias_prev ? LN_BIAS : 0) | (a.g2 ? LN_SECOND : 0);
    // decode (a few dozen rows): one wave per block so the rows spread over as many CUs as there are rows
    const int wpb = a.rows <= 256 ? 1 : 4;
    grid = dim3(ceil_div(a.rows, wpb));
#define LN_F(F) case F: hipLaunchKernelGGL((ln_kernel<NV, OUT_BF16, F>), grid, dim3(64 * wpb), 0, st, a); break;
    switch (flags) { LN_F(0) LN_F(1) LN_F(2) LN_F(3) LN_F(4) LN_F(5) LN_F(6) LN_F(7) }
#undef LN_F
}

template <bool OUT_BF16>
static int launch_ln_t(const LnArgs& a, hipStream_t st) {
    dim3 grid(ceil_div(a.rows, 4));
#define LN_CASE(NV) case NV: launch_ln_flags<NV, OUT_BF16>(a, grid, st); break;
    switch (a.D / 256) {
        LN_CASE(1) LN_CASE(2) LN_CASE(3) LN_CASE(4) LN_CASE(5) LN_CASE(6) LN_CASE(8)
        default:
            itts_set_error("layernorm: model_dim %d unsupported (need 256 * {1,2,3,4,5,6,8})", a.D);
            return ITTS_ERR_ARG;
    }
#undef LN_CASE
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

// model_dim not a multiple of 256 (small test models): one 4-byte element per lane per step, loads in phases
template <int NE, bool OUT_BF16>
__global__ __launch_bounds__(256) void ln_small_kernel(LnArgs a) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + w;
    if (r >= a.rows) return;
    const int D = a.D;
    const size_t in_r = (size_t)r * a.in_row_mul + a.in_row_add;
    float* xr = a.x + in_r * D;
    float v[NE];
    const bool upd = (a.partial != nullptr) || (a.bias_prev != nullptr);
    // phases of independent, unconditional loads (each phase under one wave-uniform branch) so they are issued
    // back-to-back; a dependent per-element "load, add, store" chain costs one memory round trip per element
#pragma unroll
    for (int i = 0; i < NE; ++i) v[i] = xr[lane + 64 * i];
    if (a.partial) {       // nsplit == 4 (checked on the host)
        const size_t ps = (size_t)a.rows * D;
        const float* pp = a.partial + (size_t)r * D + lane;
        float p0[NE], p1[NE], p2[NE], p3[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            p0[i] = pp[64 * i]; p1[i] = pp[ps + 64 * i]; p2[i] = pp[2 * ps + 64 * i]; p3[i] = pp[3 * ps + 64 * i];
        }
#pragma unroll
        for (int i = 0; i < NE; ++i) v[i] += (p0[i] + p1[i]) + (p2[i] + p3[i]);
    }
    if (a.bias_prev) {
#pragma unroll
        for (int i = 0; i < NE; ++i) v[i] += a.bias_prev[lane + 64 * i];
    }
    if (upd) {
#pragma unroll
        for (int i = 0; i < NE; ++i) xr[lane + 64 * i] = v[i];
    }
    const float invD = 1.0f / (float)D;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NE; ++i) s += v[i];
    float mean = wave_sum(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NE; ++i) { const float d = v[i] - mean; q += d * d; }
    float rstd = 1.0f / sqrtf(wave_sum(q) * invD + a.eps);
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int idx = lane + 64 * i;
        v[i] = (v[i] - mean) * rstd * a.g1[idx] + a.b1[idx];
    }
    if (a.g2) {
        s = 0.f;
#pragma unroll
        for (int i = 0; i < NE; ++i) s += v[i];
        mean = wave_sum(s) * invD;
        q = 0.f;
#pragma unroll
        for (int i = 0; i < NE; ++i) { const float d = v[i] - mean; q += d * d; }
        rstd = 1.0f / sqrtf(wave_sum(q) * invD + a.eps);
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int idx = lane + 64 * i;
            v[i] = (v[i] - mean) * rstd * a.g2[idx] + a.b2[idx];
        }
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int idx = lane + 64 * i;
        if (OUT_BF16 && !a.out_f32) ((u16*)a.out)[(size_t)r * D + idx] = f32_to_bf16(v[i]);
        else ((float*)a.out)[(size_t)r * D + idx] = v[i];
    }
}

template <bool OUT_BF16>
static int launch_ln_small_t(const LnArgs& a, hipStream_t st) {
    dim3 grid(ceil_div(a.rows, 4));
#define LN_CASE(NE) case NE: hipLaunchKernelGGL((ln_small_kernel<NE, OUT_BF16>), grid, dim3(256), 0, st, a); break;
    switch (a.D / 64) {
        LN_CASE(1) LN_CASE(2) LN_CASE(4) LN_CASE(6) LN_CASE(8) LN_CASE(10) LN_CASE(12) LN_CASE(14) LN_CASE(16) LN_CASE(20) LN_CASE(24) LN_CASE(32)
        default:
            itts_set_error("layernorm: model_dim %d unsupported (need 64 * {1,2,4,6,8,10,12,14,16,20,24,32})", a.D);
            return ITTS_ERR_ARG;
    }
#undef LN_CASE
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

// any width (the 160-wide stacked fbank rows in front of w2v-bert's feature projection): one wave per row, strided passes,
// f32 in / f32 out, no fused split-K reduce
__global__ __launch_bounds__(256) void ln_generic_kernel(LnArgs a) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + w;
    if (r >= a.rows) return;
    const int D = a.D;
    const float* xr = a.x + ((size_t)r * a.in_row_mul + a.in_row_add) * D;
    float* o = (float*)a.out + (size_t)r * D;
    const float invD = 1.0f / (float)D;
    float s = 0.f;
    for (int i = lane; i < D; i += 64) s += xr[i];
    const float mean = wave_sum(s) * invD;
    float q = 0.f;
    for (int i = lane; i < D; i += 64) { const float d = xr[i] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * invD + a.eps);
    for (int i = lane; i < D; i += 64) o[i] = (xr[i] - mean) * rstd * a.g1[i] + a.b1[i];
}

int launch_ln(const LnArgs& a, int prec, hipStream_t st) {
    if (a.rows <= 0) return ITTS_OK;
    if (a.D % 256) {      // smaller test models: the scalar-load variant
        if (a.D % 64) {
            if (a.partial || a.bias_prev || a.g2 || (prec == PREC_BF16 && !a.out_f32)) {
                itts_set_error("layernorm: D %% 64 != 0 is supported for a plain f32 -> f32 LayerNorm only");
                return ITTS_ERR_ARG;
            }
            hipLaunchKernelGGL(ln_generic_kernel, dim3(ceil_div(a.rows, 4)), dim3(256), 0, st, a);
            HIP_TRY(hipGetLastError());
            return ITTS_OK;
        }
        return prec == PREC_BF16 ? launch_ln_small_t<true>(a, st) : launch_ln_small_t<false>(a, st);
    }
    if (a.partial && a.nsplit != 4) { itts_set_error("layernorm: fused split-K reduce expects 4 slices"); return ITTS_ERR_ARG; }
    return prec == PREC_BF16 ? launch_ln_t<true>(a, st) : launch_ln_t<false>(a, st);
}

// ================================================================================================================
// GEMM on MFMA.  Packed weights: [N/16][K/KB][64 lanes][16 bytes]
//   bf16 (KB = 32): lane holds W[kb*32 + (lane>>4)*8 + j][nt*16 + (lane&15)], j = 0..7
//   f32  (KB = 16): lane holds W[kb*16 + (lane>>4)*4 + j][nt*16 + (lane&15)], j = 0..3 (MFMA j of the group)
// A is row-major act dtype; the matching 16 bytes of row (lane&15) are loaded straight from global (L2-resident).
// ================================================================================================================

template <bool BF16>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& a, int mbase, int nbase, int lane, int z, f32x4 v) {
    const int n = nbase + (lane & 15);
    if (n >= a.N) return;
    const float bias = (a.bias && a.epi != EPI_PARTIAL) ? a.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = mbase + (lane >> 4) * 4 + r;
        if (m >= a.M) continue;
        const float val = v[r] + bias;
        switch (a.epi) {
            case EPI_STORE_F32: a.out_f32[(size_t)m * a.ldo + n] = val; break;
            case EPI_RESIDUAL: a.out_f32[(size_t)m * a.ldo + n] += val; break;
            case EPI_GELU_ACT: {
                const float g = gelu_new_f(val);
                if (BF16) ((u16*)a.out_act)[(size_t)m * a.ldo + n] = f32_to_bf16(g);
                else ((float*)a.out_act)[(size_t)m * a.ldo + n] = g;
                break;
            }
            case EPI_PARTIAL: a.partial[((size_t)z * a.M + m) * a.N + n] = val; break;
            case EPI_QKV: {
                const int b = m / a.S, si = m - b * a.S;
                const int which = n / a.D, c = n - which * a.D;
                if (which == 0) {
                    a.qbuf[(size_t)m * a.D + c] = val;
                } else {
                    const size_t pb = a.seq_map ? (size_t)a.seq_map[b] : (size_t)b * (a.seq_mul > 1 ? a.seq_mul : 1);
                    const int pos = *a.pos_ptr + si - (a.pos_shift ? a.pos_shift[pb] : 0);
                    const size_t o = ((pb * a.H + (c >> 6)) * a.Tmax + pos) * 64 + (c & 63);
                    void* cache = which == 1 ? a.kcache : a.vcache;
                    if (pos >= a.Tmax) break;                   // a finished row of a long-running session (its slot awaits an admission): nothing to keep
                    if (BF16) ((u16*)cache)[o] = f32_to_bf16(val);
                    else ((float*)cache)[o] = val;
                }
                break;
            }
        }
    }
}

template <bool BF16, int MT, int NT, bool KSPLIT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, KSPLIT ? 1 : 4))) void gemm_kernel(GemmArgs a) {
    constexpr int KB = BF16 ? 32 : 16;
    constexpr int ESZ = BF16 ? 2 : 4;
    extern __shared__ __attribute__((aligned(16))) float red[];   // KSPLIT: [4][MT*NT][64] f32x4
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nkb = a.K / KB;
    const int z = blockIdx.z;
    const int kb_lo = (int)((long long)z * nkb / a.nsplit);
    const int kb_hi = (int)((long long)(z + 1) * nkb / a.nsplit);
    const int ntiles = (a.N + 15) >> 4;
    const int nt0 = KSPLIT ? blockIdx.x * NT : (blockIdx.x * 4 + w) * NT;
    const int kb_start = KSPLIT ? kb_lo + w : kb_lo;
    const int kb_step = KSPLIT ? 4 : 1;
    const int m0 = blockIdx.y * MT * 16;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const char* arow[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int m = m0 + mt * 16 + (lane & 15);
        m = m < a.M ? m : a.M - 1;
        arow[mt] = (const char*)a.A + ((size_t)m * a.lda + (lane >> 4) * (KB / 4)) * ESZ;
    }
    const v4u* wp[NT];
    bool nt_ok[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        nt_ok[nt] = (nt0 + nt) < ntiles;
        const int t = nt_ok[nt] ? nt0 + nt : ntiles - 1;
        wp[nt] = (const v4u*)a.Wp + (size_t)t * nkb * 64 + lane;
    }

    auto mfma_step = [&](const v4u (&af)[MT], const v4u (&bfr)[NT]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if constexpr (BF16) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8_t, af[mt]), __builtin_bit_cast(bf16x8_t, bfr[nt]), acc[mt][nt], 0, 0, 0);
                } else {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[mt].x), __uint_as_float(bfr[nt].x), acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[mt].y), __uint_as_float(bfr[nt].y), acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[mt].z), __uint_as_float(bfr[nt].z), acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[mt].w), __uint_as_float(bfr[nt].w), acc[mt][nt], 0, 0, 0);
                }
            }
    };

    if constexpr (KSPLIT && MT < 4) {
        // Decode with <= 32 rows: activation fragments straight from L2 (the slab is small), every load of the wave's
        // K range issued before the first MFMA (sched_barrier keeps hipcc from re-serialising them behind vmcnt(0)).
        constexpr int PF = 10;
        for (int kb0 = kb_start; kb0 < kb_hi; kb0 += PF * kb_step) {
            v4u bq[PF][NT], aq[PF][MT];
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const int kbi = kb0 + i * kb_step;
                const bool ok = kbi < kb_hi;
                const int kbc = ok ? kbi : kb0;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    v4u v = *(const v4u*)(wp[nt] + (size_t)kbc * 64);
                    if (!ok) v = v4u{0u, 0u, 0u, 0u};
                    bq[i][nt] = v;
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) aq[i][mt] = *(const v4u*)(arow[mt] + (size_t)kbc * KB * ESZ);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < PF; ++i) mfma_step(aq[i], bq[i]);
        }
    } else if constexpr (KSPLIT) {
        // Decode (weight streaming), 64 rows.  Measured with ablations (tools/gemm_ablate.py): the HBM weight stream is NOT
        // the bottleneck at M = 64, the activation operand is -- fragment-shaped loads (16 rows x 64 B per wave
        // instruction) straight from L2 cost 4-17 us per GEMM.  So the block stages its activation slab through LDS
        // in full 1280-byte row segments (coalesced 16 B per lane, 8 full lines per wave instruction), 20 k-blocks
        // (one 1280-byte segment per row) at a time, and the waves read their A fragments with ds_read_b128
        // (row stride padded by 16 B: the 16 rows of a fragment hit 16 distinct 4-bank groups).  All weight fragments of
        // a 40-k-block super-chunk are issued up front so the HBM latency overlaps the staging.
        constexpr int CH = 20;                        // k-blocks per staged chunk: CH * 64 B = 1280 B per row
        constexpr int ROWB = CH * 64 + 16;            // padded LDS row stride in bytes
        constexpr int ROWS = MT * 16;
        constexpr int NLD = ROWS * (CH * 4) / 256;    // 16-byte pieces per thread per chunk (5 * MT)
        char* lds_a = (char*)red + (size_t)4 * MT * NT * 64 * 16;     // behind the reduction scratch
        const char* abase = (const char*)a.A;
        const size_t lda_b = (size_t)a.lda * ESZ;
        auto load_b5 = [&](int cb, v4u (&bq)[5][NT]) {       // this wave's 5 k-blocks of the chunk starting at cb
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int kbi = cb + w + 4 * i;
                const bool ok = kbi < kb_hi;
                const int kbc = ok ? kbi : kb_lo;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    v4u v = *(const v4u*)(wp[nt] + (size_t)kbc * 64);
                    if (!ok) v = v4u{0u, 0u, 0u, 0u};
                    bq[i][nt] = v;
                }
            }
        };
        auto stage = [&](int cb) {                    // activation rows [m0, m0 + ROWS) x k-blocks [cb, cb + CH) -> LDS
            v4u st[NLD];
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int idx = threadIdx.x + 256 * i;
                const int row = idx / (CH * 4), piece = idx - row * (CH * 4);
                int m = m0 + row;
                m = m < a.M ? m : a.M - 1;
                long long kbyte = ((long long)cb * KB) * ESZ + piece * 16;          // byte offset within the row
                const long long kmax = (long long)a.K * ESZ - 16;
                kbyte = kbyte < kmax ? kbyte : kmax;                                 // tail chunk: clamp (unused k)
                st[i] = *(const v4u*)(abase + (size_t)m * lda_b + kbyte);
            }
            __syncthreads();                          // previous chunk's fragments consumed
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int idx = threadIdx.x + 256 * i;
                const int row = idx / (CH * 4), piece = idx - row * (CH * 4);
                *(v4u*)(lds_a + row * ROWB + piece * 16) = st[i];
            }
            __syncthreads();
        };
        auto compute = [&](const v4u (&bq)[5][NT]) {
#pragma unroll
            for (int i5 = 0; i5 < 5; ++i5) {
                const int kl = w + 4 * i5;            // k-block within the chunk
                v4u af[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    af[mt] = *(const v4u*)(lds_a + (mt * 16 + (lane & 15)) * ROWB + kl * 64 + (lane >> 4) * 16);
                mfma_step(af, bq[i5]);
            }
        };
#pragma unroll 1
        for (int sc0 = kb_lo; sc0 < kb_hi; sc0 += 2 * CH) {
            v4u bq0[5][NT], bq1[5][NT];
            load_b5(sc0, bq0);                        // HBM weight stream of the first half: in flight during staging
            stage(sc0);
            __builtin_amdgcn_sched_barrier(0);        // keep the second half's loads below the staging (register budget)
            load_b5(sc0 + CH, bq1);                   // second half: in flight during the first half's MFMAs + staging
            compute(bq0);
            if (sc0 + CH < kb_hi) {                   // block-uniform
                stage(sc0 + CH);
                compute(bq1);
            }
        }
    } else {
#pragma unroll 2
        for (int kb = kb_start; kb < kb_hi; kb += kb_step) {
            v4u af[MT], bfr[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[mt] = *(const v4u*)(arow[mt] + (size_t)kb * KB * ESZ);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bfr[nt] = wp[nt][(size_t)kb * 64];
            mfma_step(af, bfr);
        }
    }

    if constexpr (KSPLIT) {
        f32x4* r4 = (f32x4*)red;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) r4[((size_t)w * (MT * NT) + mt * NT + nt) * 64 + lane] = acc[mt][nt];
        __syncthreads();
        for (int tile = w; tile < MT * NT; tile += 4) {
            f32x4 s = r4[(size_t)tile * 64 + lane];
#pragma unroll
            for (int ww = 1; ww < 4; ++ww) {
                const f32x4 o = r4[((size_t)ww * (MT * NT) + tile) * 64 + lane];
                s[0] += o[0]; s[1] += o[1]; s[2] += o[2]; s[3] += o[3];
            }
            const int mt = tile / NT, nt = tile - mt * NT;
            if ((nt0 + nt) < ntiles) gemm_epilogue<BF16>(a, m0 + mt * 16, (nt0 + nt) * 16, lane, z, s);
        }
    } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                if (nt_ok[nt]) gemm_epilogue<BF16>(a, m0 + mt * 16, (nt0 + nt) * 16, lane, z, acc[mt][nt]);
    }
}

// ================================================================================================================
// Prefill / teacher-forced GEMM, bf16: C[M,N] = A[M,K] * W on v_mfma_f32_16x16x32_bf16.
//   block tile 128 x 128, BK = 64, 4 waves as 2 (M) x 2 (N), each 64 x 64 = 4 x 4 MFMA tiles (64 accumulator registers);
//   both operands go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction), two LDS buffers:
//   tile t+1 is in flight while tile t feeds the MFMAs; one barrier per K tile.
//   A image: 16 chunks of [8 rows][128 B]; a chunk is filled by ONE instruction reading 8 full 128-byte lines, the 16-byte
//   pieces of a row XOR-permuted by (row16 >> 1) on the SOURCE side (the LDS destination of an LDS-DMA is lane-linear) so
//   the 16 rows of a fragment read hit 16 distinct 16-byte bank slots.  W is already stored in fragment order
//   ([N/16][K/32][64 lanes][16 B]): a chunk is one contiguous KiB and its ds_read_b128 is lane-linear.
//   Block -> tile map: XCD-aware (consecutive tiles of one XCD's share) and grouped 8 m-tiles x all n-tiles so the
//   blocks resident on an XCD reuse A rows and W columns out of that XCD's L2.
// ================================================================================================================
// F32 = true: the same kernel on v_mfma_f32_16x16x4_f32 (exact f32: the s2mel f32 mode and the GPT parity mode's prefill).  A K tile is
// 32 f32 = the same 128 bytes per row, the packed f32 weights ([N/16][K/16][64 lanes][16 B]) give the same 2 KiB per n-tile and K tile,
// so the LDS images, the DMA issue and the fragment read offsets are byte-identical; a 16-byte fragment piece now holds 4 k-values =
// 4 MFMAs (lane group kg supplies k = 16 s2 + 4 kg + j to MFMA j), issued j-outer so consecutive MFMAs never chain on one accumulator
// (40-cycle dependent latency vs 32-cycle issue).  Per output element the accumulation order equals gemm_kernel<false>'s (k-blocks
// ascending, MFMAs x, y, z, w) -> bitwise the same results.  Per K tile a wave issues 128 MFMAs x 32 cycles against 16 fragment reads
// and 32 KiB of DMA per block: MFMA-bound (157 TFLOP/s peak).
template <int EPI, bool CONV = false, bool VEC = true, bool F32 = false>      // VEC: the LDS-transposed vector epilogue (N, ldo, D multiples of 4)
__global__ __launch_bounds__(256) void gemm_prefill_kernel(GemmArgs a) {
    constexpr int ES = F32 ? 4 : 2;                                   // operand element size
    constexpr int BK = 128 / ES;                                      // K tile: 128 bytes per row (64 bf16 / 32 f32)
    extern __shared__ __attribute__((aligned(16))) char pf_sm[];      // [2][A 16 KiB | W 16 KiB]
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // scalar: LDS-DMA destinations (M0) need no v_readfirstlane
    const int wr = w >> 1, wc = w & 1;
    const int n_mt = (a.M + PF_BM - 1) / PF_BM, n_nt = (a.N + PF_BN - 1) / PF_BN;
    const int total = n_mt * n_nt, per = (total + 7) >> 3;
    const int t = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (t >= total) return;
    const int g = t / (PF_GM * n_nt), first_m = g * PF_GM;
    const int gm = (n_mt - first_m) < PF_GM ? (n_mt - first_m) : PF_GM;
    const int r = t - g * PF_GM * n_nt;
    const int bn = r / gm, bm = first_m + (r - bn * gm);
    const int m0 = bm * PF_BM, nt0 = bn * (PF_BN / 16);
    const int nkb = F32 ? a.K >> 4 : a.K >> 5, nk = a.K / BK;
    const int ntiles = (a.N + 15) >> 4;

    // staging sources of this lane: 4 A chunks (rows) and 4 W chunks per wave per K tile
    const char* asrc[4];
    const char* bsrc[4];
    int cv_t[4], cv_T[4];
    const char* cv_base[4];
    const char* cv_zero[4];
    const int cv_kpt = CONV ? a.conv_W / BK : 1;               // K tiles per tap
    const int cv_left = CONV ? (a.conv_taps - 1) * a.conv_dil - ((a.conv_taps - 1) * a.conv_dil) / 2 : 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = w * 4 + i;                               // A chunk: tile rows c*8 .. c*8+7
        const int row_t = c * 8 + (lane >> 3), row16 = row_t & 15;
        const int piece = (lane & 7) ^ ((row16 >> 1) & 7);
        int m = m0 + row_t;
        m = m < a.M ? m : a.M - 1;
        asrc[i] = (const char*)a.A + (size_t)m * a.lda * ES + piece * 16;
        if constexpr (CONV) {                                  // implicit im2col: remember the row's frame and sequence extent
            const int sq = a.tok_seq[m];
            cv_t[i] = a.tok_t[m];
            cv_T[i] = a.seq_T[sq];
            cv_base[i] = (const char*)a.A + (size_t)a.seq_start[sq] * a.lda * ES + piece * 16;
            cv_zero[i] = (const char*)a.zero_row + piece * 16;
        }
        const int nblk = w * 2 + (i >> 1), kb = i & 1;          // W chunk (n-block, k-block of the pair)
        int nt = nt0 + nblk;
        nt = nt < ntiles ? nt : ntiles - 1;
        bsrc[i] = (const char*)a.Wp + ((size_t)nt * nkb + kb) * 1024 + lane * 16;
    }
    auto issue = [&](int kt, int buf) {
        char* base = pf_sm + buf * 32768;
        int tap = 0, rem = kt;
        if constexpr (CONV) { tap = kt / cv_kpt; rem = kt - tap * cv_kpt; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const char* ap = asrc[i] + (size_t)kt * 128;
            if constexpr (CONV) {
                const int maxpad = cv_left;                                        // left >= right
                const int Tv = cv_T[i] <= maxpad ? maxpad + 1 : cv_T[i];           // zero-extended length of very short inputs
                int p = cv_t[i] + tap * a.conv_dil - cv_left;
                p = p < 0 ? -p : p;
                p = p >= Tv ? 2 * (Tv - 1) - p : p;
                const bool ok = p >= 0 && p < cv_T[i];
                ap = ok ? cv_base[i] + ((size_t)p * a.lda + (size_t)rem * BK) * ES : cv_zero[i];
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ap,
                                             (__attribute__((address_space(3))) void*)(base + (w * 4 + i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc[i] + (size_t)kt * 2048),
                                             (__attribute__((address_space(3))) void*)(base + 16384 + ((w * 2 + (i >> 1)) * 2 + (i & 1)) * 1024), 16, 0, 0);
        }
    };


    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets (bytes inside a buffer)
    const int row16 = lane & 15, kg = lane >> 4;
    int a_off[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const int pos = (s2 * 4 + kg) ^ ((row16 >> 1) & 7);
        a_off[s2] = (row16 >> 3) * 1024 + ((row16 & 7) * 8 + pos) * 16;
    }
    const int a_wave = wr * 4 * 2048;                           // 4 m-blocks of 2 chunks each
    const int b_wave = 16384 + wc * 4 * 2048 + lane * 16;

    issue(0, 0);
    // one K tile: this tile's operands have landed (every wave's pieces: wait + barrier), the next tile is requested into the other stage (a branch:
    // its own basic block, the DMA burst sits between the barrier and the first fragment read), then the MFMAs
    auto k_tile = [&](int kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
        const char* base = pf_sm + (kt & 1) * 32768;
        // Both k-steps' fragments are read up front into separate registers and the second step's reads are interleaved with
        // the first step's MFMAs (sched_group_barrier: 2 MFMAs per LDS read): left to itself hipcc emitted read-all / wait /
        // 8 MFMAs / 2 reads / wait / ..., i.e. four exposed LDS round trips per K tile.
        v4u af0[4], bf0[4], af1[4], bf1[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) af0[mt] = *(const v4u*)(base + a_wave + mt * 2048 + a_off[0]);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bf0[nt] = *(const v4u*)(base + b_wave + (nt * 2 + 0) * 1024);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) af1[mt] = *(const v4u*)(base + a_wave + mt * 2048 + a_off[1]);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bf1[nt] = *(const v4u*)(base + b_wave + (nt * 2 + 1) * 1024);
        if constexpr (F32) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float((s2 ? af1 : af0)[mt][j]),
                                                                               __uint_as_float((s2 ? bf1 : bf0)[nt][j]), acc[mt][nt], 0, 0, 0);
        } else {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af0[mt]),
                                                                          __builtin_bit_cast(bf16x8_t, bf0[nt]), acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af1[mt]),
                                                                          __builtin_bit_cast(bf16x8_t, bf1[nt]), acc[mt][nt], 0, 0, 0);
        }
        if (PF_SCHED && F32) {                                      // f32: 8 MFMAs (256 cycles) per read of the second k-step's fragments
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 64, 0);
        }
        if (PF_SCHED && !F32) {
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);      // 8 DS reads: the first k-step's fragments
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // 2 MFMAs ...
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // ... then one read of the second k-step
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);     // the second k-step's MFMAs
        }
    };
    for (int kt = 0; kt < nk; ++kt) k_tile(kt);
    // Epilogue.  Vector path (every shape of the engine except an odd-width GPT head: N, ldo, D multiples of 4): transpose the
    // accumulator tile through LDS and store full-line pieces (pf_store_tile); otherwise (VEC = false instantiations of the
    // GPT epilogues) the per-lane element path.
    if constexpr (VEC) {
        __syncthreads();                                           // every wave is done with the operand buffers
        float* ct = (float*)pf_sm;                                 // [128][128] f32 (row-major) or [128][132] (transposed, V^T tiles)
        const int g = lane >> 4, c16 = lane & 15;
        if (EPI == EPI_QKV_ROPE && a.D % 128 == 0 && nt0 * 16 >= 2 * a.D) {       // block-uniform: a V tile
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) *(f32x4*)(ct + (wc * 64 + nt * 16 + c16) * 132 + wr * 64 + mt * 16 + g * 4) = acc[mt][nt];
            __syncthreads();
            pf_store_vt<128, 256, F32>(a, ct, m0, nt0 * 16, threadIdx.x);
            return;
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    ct[(wr * 64 + mt * 16 + g * 4 + r) * 128 + ((wc * 64 + nt * 16 + c16) ^ (g << 4))] = acc[mt][nt][r];
        int* meta = (int*)(pf_sm + 65536);                          // 1 KiB of the slack above the 64 KiB row-major image
        pf_stage_meta<EPI, 128>(a, meta, m0, threadIdx.x);
        __syncthreads();
        pf_store_tile<EPI, 128, 256, F32>(a, ct, meta, m0, nt0 * 16, threadIdx.x);
    } else {
        static_assert(!F32, "the f32 tile kernel has the vector epilogue only");
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int ntile = nt0 + wc * 4 + nt;
                if (ntile < ntiles) pf_epilogue<EPI>(a, m0 + wr * 64 + mt * 16, ntile * 16, lane, acc[mt][nt]);
            }
    }
}


template <int EPI, bool CONV = false>
static int launch_gemm_prefill_e(const GemmArgs& a, hipStream_t st) {
    const int n_mt = ceil_div(a.M, PF_BM), n_nt = ceil_div(a.N, PF_BN);
    const int per = ceil_div(n_mt * n_nt, 8);
    static ItPerDevice<bool> attr_set_pd;
    bool& attr_set = attr_set_pd.cur();
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)gemm_prefill_kernel<EPI, CONV, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PF_LDS));
        if constexpr (EPI <= EPI_QKV) HIP_TRY(hipFuncSetAttribute((const void*)gemm_prefill_kernel<EPI, CONV, false>, hipFuncAttributeMaxDynamicSharedMemorySize, PF_LDS));
        attr_set = true;
    }
    if (pf_vec_ok(a)) {
        hipLaunchKernelGGL((gemm_prefill_kernel<EPI, CONV, true>), dim3(per * 8), dim3(256), PF_LDS, st, a);
    } else {
        if constexpr (EPI <= EPI_QKV) {
            hipLaunchKernelGGL((gemm_prefill_kernel<EPI, CONV, false>), dim3(per * 8), dim3(256), PF_LDS, st, a);
        } else {
            itts_set_error("gemm: the fused s2mel epilogue %d needs N, D multiples of 4", EPI);
            return ITTS_ERR_ARG;
        }
    }
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

// ================================================================================================================
// 256 x 256 tile GEMM, bf16, 8 waves (2 along M x 4 along N; each wave 128 x 64 = 8 x 4 MFMA tiles), BK = 64, for the big-M GEMMs
// of the s2mel DiT / WaveNet (M = 2 x frames of the whole batch).
//   Why a second tile kernel: a 64 x 64 wave tile reads 0.5 KiB of LDS fragments per MFMA, a 128 x 64 one 0.375 KiB, and the
//   128 x 128 kernel drains its LDS-DMA queue (vmcnt(0) + barrier) once per K tile.  Measured (profiles/r02f..r02h): +10..14 % on
//   the GPT prefill shapes (K = 1280 / 5120: 790 -> 900 TFLOP/s), +13 % on the K = 512 residual GEMMs of the DiT, par on its other
//   GEMMs, whose time is set by their epilogues (one block per CU: nothing overlaps a tile's epilogue or first loads).
//   Schedule: the K tile is computed as 4 quadrants of the wave's output (phases); each phase = [fragment reads of the quadrant,
//   issue one half-operand LDS-DMA of the NEXT K tile, counted vmcnt] barrier [16 MFMAs] barrier.  The waves form two groups
//   (waves 0-3 / 4-7 = one wave of each group per SIMD) that run half a phase apart: while one group's MFMAs occupy the matrix
//   pipe the other group reads its fragments, so neither the LDS reads nor the DMA issue are exposed.
//   LDS-DMA ordering: a half-operand is waited for (counted s_waitcnt vmcnt by every issuing wave, before the phase's first
//   barrier) one phase before it is first read; a buffer is restaged one K tile after its last read.
//   LDS images, fragment read offsets, packed-weight format, accumulation order per output element (k ascending) and the
//   epilogues are those of gemm_prefill_kernel -> the two kernels are bitwise interchangeable.
// ================================================================================================================
#define T2_REGION 34816  // epilogue image of one [64 rows][128 cols] region: row-major 32 KiB, or transposed [128][68] f32
#define T2_LDS (4 * T2_REGION)   // >= the 2 x 64 KiB operand buffers

template <int VM> __device__ __forceinline__ void t2_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM) : "memory"); }

// F32 = true: the same schedule on v_mfma_f32_16x16x4_f32 (the s2mel f32 mode): a K tile is 32 f32 = the same 128 bytes per row and 2 KiB per
// n-tile of packed f32 weights, so the LDS images, DMA issue and fragment offsets are byte-identical; a 16-byte fragment piece feeds 4
// MFMAs (k-step s2, element j: k = 16 s2 + 4 kg + j), issued j-outer like gemm_prefill_kernel<.., F32 = true> -> bitwise its results.  A phase
// is then 64 MFMAs x 32 cycles against the same 12 fragment reads and 2-4 DMA pieces: half the staged bytes per MFMA of the 128 x 128 kernel
// (whose staging costs 8-10 % whatever the schedule, profiles/r03n).  Measured (tools/microbench/gemm_f32_ablate.hip -DUSE_T256,
// profiles/r03n/gemm_f32_tile256.log): identical output bits, 1-3 % SLOWER than the 128 x 128 kernel at M = 312 704 (129-133 vs 131-137
// TFLOP/s), far behind it at small M (one block per CU: tail rounds) -> not instantiated in the product; the microbench builds it.
template <int EPI, bool CONV = false, bool F32 = false>
__global__ __launch_bounds__(512) void gemm_tile256_kernel(GemmArgs a) {
    constexpr int ES = F32 ? 4 : 2;                                    // operand element size
    constexpr int BK = 128 / ES;                                       // K tile: 128 bytes per row
    extern __shared__ __attribute__((aligned(16))) char t2_sm[];      // [2][A 32 KiB | W 32 KiB]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wr = w >> 2, wc = w & 3;                                 // wr = wave group = M half of the tile
    const int n_mt = (a.M + 255) / 256, n_nt = (a.N + 255) / 256;
    const int total = n_mt * n_nt, per = (total + 7) >> 3;
    const int t = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (t >= total) return;
    const int gq = t / (PF_GM * n_nt), first_m = gq * PF_GM;
    const int gm = (n_mt - first_m) < PF_GM ? (n_mt - first_m) : PF_GM;
    const int rr = t - gq * PF_GM * n_nt;
    const int bn = rr / gm, bm = first_m + (rr - bn * gm);
    const int m0 = bm * 256, nt0 = bn * 16;
    const int nkb = F32 ? a.K >> 4 : a.K >> 5, nk = a.K / BK;
    const int ntiles = (a.N + 15) >> 4;

    // Staging sources of this lane.  Operand halves follow the quadrant order: A half h = m-tiles {8 wr' + 4 h + 0..3, wr' = 0, 1},
    // W half h = n-tiles {4 wc' + 2 h + 0..1, wc' = 0..3}; a half is 16 chunks of 1 KiB = 2 per wave.
    const char* asrc[2][2];
    const char* bsrc[2][2];
    int a_dst[2][2], b_dst[2][2];
    int cv_t[2][2], cv_T[2][2];
    const char* cv_base[2][2];
    const char* cv_zero[2][2];
    const int cv_kpt = CONV ? a.conv_W / BK : 1;               // K tiles per tap
    const int cv_left = CONV ? (a.conv_taps - 1) * a.conv_dil - ((a.conv_taps - 1) * a.conv_dil) / 2 : 0;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int cidx = w * 2 + i;
            const int chunk = (((cidx >> 3) * 8 + h * 4 + ((cidx >> 1) & 3)) << 1) + (cidx & 1);   // 8 tile rows each
            const int row_t = chunk * 8 + (lane >> 3), row16 = row_t & 15;
            const int piece = (lane & 7) ^ ((row16 >> 1) & 7);
            int m = m0 + row_t;
            m = m < a.M ? m : a.M - 1;
            asrc[h][i] = (const char*)a.A + (size_t)m * a.lda * ES + piece * 16;
            a_dst[h][i] = chunk * 1024;
            if constexpr (CONV) {
                const int sq = a.tok_seq[m];
                cv_t[h][i] = a.tok_t[m];
                cv_T[h][i] = a.seq_T[sq];
                cv_base[h][i] = (const char*)a.A + (size_t)a.seq_start[sq] * a.lda * ES + piece * 16;
                cv_zero[h][i] = (const char*)a.zero_row + piece * 16;
            }
            const int ntl = (w >> 1) * 4 + h * 2 + (w & 1);     // n-tile inside the block tile; k-block of the pair = i
            int nt = nt0 + ntl;
            nt = nt < ntiles ? nt : ntiles - 1;
            bsrc[h][i] = (const char*)a.Wp + ((size_t)nt * nkb + i) * 1024 + lane * 16;
            b_dst[h][i] = 32768 + (ntl * 2 + i) * 1024;
        }
    auto issue_a = [&](int kt, int buf, int h) {
        int tap = 0, rem = kt;
        if constexpr (CONV) { tap = kt / cv_kpt; rem = kt - tap * cv_kpt; }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const char* ap = asrc[h][i] + (size_t)kt * 128;
            if constexpr (CONV) {
                const int maxpad = cv_left;
                const int Tv = cv_T[h][i] <= maxpad ? maxpad + 1 : cv_T[h][i];
                int p = cv_t[h][i] + tap * a.conv_dil - cv_left;
                p = p < 0 ? -p : p;
                p = p >= Tv ? 2 * (Tv - 1) - p : p;
                const bool ok = p >= 0 && p < cv_T[h][i];
                ap = ok ? cv_base[h][i] + ((size_t)p * a.lda + (size_t)rem * BK) * ES : cv_zero[h][i];
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ap,
                                             (__attribute__((address_space(3))) void*)(t2_sm + buf * 65536 + a_dst[h][i]), 16, 0, 0);
        }
    };
    auto issue_b = [&](int kt, int buf, int h) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc[h][i] + (size_t)kt * 2048),
                                             (__attribute__((address_space(3))) void*)(t2_sm + buf * 65536 + b_dst[h][i]), 16, 0, 0);
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int row16 = lane & 15, kg = lane >> 4;
    int a_off[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const int pos = (s2 * 4 + kg) ^ ((row16 >> 1) & 7);
        a_off[s2] = (row16 >> 3) * 1024 + ((row16 & 7) * 8 + pos) * 16;
    }
    const int a_wave = wr * 8 * 2048;
    const int b_wave = 32768 + wc * 4 * 2048 + lane * 16;

    v4u af[4][2], b0[2][2], b1[2][2];
#define T2_READ_A(H_)                                                                                     \
    _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2)     \
        af[mt][s2] = *(const v4u*)(base + a_wave + ((H_) * 4 + mt) * 2048 + a_off[s2]);
#define T2_READ_B(DST_, H_)                                                                               \
    _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2)     \
        DST_[nt][s2] = *(const v4u*)(base + b_wave + (((H_) * 2 + nt) * 2 + s2) * 1024);
#define T2_MMA(MH_, NH_, BF_)                                                                             \
    if constexpr (F32) {                                                                                  \
        _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) _Pragma("unroll") for (int j = 0; j < 4; ++j)    \
            _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)  \
                acc[(MH_) * 4 + mt][(NH_) * 2 + nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(               \
                    __uint_as_float(af[mt][s2][j]), __uint_as_float(BF_[nt][s2][j]), acc[(MH_) * 4 + mt][(NH_) * 2 + nt], 0, 0, 0);  \
    } else {                                                                                              \
    _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)     \
        _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                  \
            acc[(MH_) * 4 + mt][(NH_) * 2 + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                \
                __builtin_bit_cast(bf16x8_t, af[mt][s2]), __builtin_bit_cast(bf16x8_t, BF_[nt][s2]), acc[(MH_) * 4 + mt][(NH_) * 2 + nt], 0, 0, 0);  \
    }
#define T2_PRE_MMA()                                        \
    __builtin_amdgcn_s_barrier();                           \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
    __builtin_amdgcn_sched_barrier(0);                      \
    __builtin_amdgcn_s_setprio(1);
#define T2_POST_MMA()                                       \
    __builtin_amdgcn_s_setprio(0);                          \
    __builtin_amdgcn_sched_barrier(0);                      \
    __builtin_amdgcn_s_barrier();                           \
    asm volatile("" ::: "memory");                          \
    __builtin_amdgcn_sched_barrier(0);

    issue_a(0, 0, 0);
    issue_b(0, 0, 0);
    issue_b(0, 0, 1);
    issue_a(0, 0, 1);
    t2_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (wr == 1) __builtin_amdgcn_s_barrier();                 // group 1 runs half a phase behind group 0
    __builtin_amdgcn_sched_barrier(0);
    for (int kt = 0; kt < nk; ++kt) {
        const char* base = t2_sm + (kt & 1) * 65536;
        const int nb = (kt + 1) & 1;
        const bool more = kt + 1 < nk;                          // block-uniform
        // phase 0: quadrant (m-tiles 0-3, n-tiles 0-1)
        T2_READ_B(b0, 0)
        __builtin_amdgcn_sched_barrier(0);
        T2_READ_A(0)
        if (more) { issue_a(kt + 1, nb, 0); t2_wait_vm<4>(); } else t2_wait_vm<0>();
        T2_PRE_MMA()
        T2_MMA(0, 0, b0)
        T2_POST_MMA()
        // phase 1: (m-tiles 0-3, n-tiles 2-3)
        T2_READ_B(b1, 1)
        if (more) { issue_b(kt + 1, nb, 0); t2_wait_vm<4>(); }
        T2_PRE_MMA()
        T2_MMA(0, 1, b1)
        T2_POST_MMA()
        // phase 2: (m-tiles 4-7, n-tiles 2-3)
        T2_READ_A(1)
        if (more) { issue_b(kt + 1, nb, 1); t2_wait_vm<6>(); }
        T2_PRE_MMA()
        T2_MMA(1, 1, b1)
        T2_POST_MMA()
        // phase 3: (m-tiles 4-7, n-tiles 0-1)
        if (more) { issue_a(kt + 1, nb, 1); t2_wait_vm<4>(); }
        T2_PRE_MMA()
        T2_MMA(1, 0, b0)
        T2_POST_MMA()
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();                 // group 0 catches up
#undef T2_READ_A
#undef T2_READ_B
#undef T2_MMA
#undef T2_PRE_MMA
#undef T2_POST_MMA
    // Epilogue: two rounds (the wave's m-tiles 0-3, then 4-7) through four [64 rows][128 cols] LDS regions, region w >> 1 written
    // and stored by waves 2 (w >> 1), 2 (w >> 1) + 1.
    __syncthreads();
    const int g = lane >> 4, c16 = lane & 15;
    const int region = w >> 1;
    float* ct = (float*)(t2_sm + region * T2_REGION);
    const int rn0 = nt0 * 16 + (region & 1) * 128;              // first column of the region
    const bool v_region = EPI == EPI_QKV_ROPE && a.D % 128 == 0 && rn0 >= 2 * a.D;
    const int ltid = threadIdx.x & 127;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (v_region) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) *(f32x4*)(ct + ((wc & 1) * 64 + nt * 16 + c16) * 68 + mt * 16 + g * 4) = acc[h * 4 + mt][nt];
        } else {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) ct[(mt * 16 + g * 4 + r) * 128 + (((wc & 1) * 64 + nt * 16 + c16) ^ (g << 4))] = acc[h * 4 + mt][nt][r];
        }
        const int rm0 = m0 + wr * 128 + h * 64;
        int* meta = (int*)((char*)ct + 32768);                     // 512 B of the region's slack above the 32 KiB row-major image
        if (!v_region) pf_stage_meta<EPI, 64>(a, meta, rm0, ltid);
        __syncthreads();
        if (rn0 < a.N) {
            if (v_region) pf_store_vt<64, 128, F32>(a, ct, rm0, rn0, ltid);
            else pf_store_tile<EPI, 64, 128, F32>(a, ct, meta, rm0, rn0, ltid);
        }
        if (h == 0) __syncthreads();
    }
}

template <int EPI, bool CONV = false>
static int launch_gemm_tile256_e(const GemmArgs& a, hipStream_t st) {
    const int n_mt = ceil_div(a.M, 256), n_nt = ceil_div(a.N, 256);
    const int per = ceil_div(n_mt * n_nt, 8);
    static ItPerDevice<bool> attr_set_pd;
    bool& attr_set = attr_set_pd.cur();
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)gemm_tile256_kernel<EPI, CONV>, hipFuncAttributeMaxDynamicSharedMemorySize, T2_LDS));
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_tile256_kernel<EPI, CONV>), dim3(per * 8), dim3(512), T2_LDS, st, a);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

// ================================================================================================================
// 256 x 128 tile GEMM, bf16, FOUR waves (2 along M x 2 along N, each 128 x 64 = 8 x 4 MFMA tiles), K step 32, three-stage
// LDS-DMA ring, two blocks per CU.
//   An attempt to keep the 256 x 256 kernel's wave tile (0.375 KiB of LDS fragment reads per MFMA) AND two blocks per CU
//   (72 KiB each), so that one block's epilogue / first loads overlap the other's main loop.  Measured (profiles/r02h): par with
//   the 256 x 256 kernel on the DiT shapes, behind both other kernels on the GPT prefill shapes (64-byte row pieces per DMA).
//   Kept for the A/B harness (ITTS_TILE256=2); not selected by default.
//   Ring: step k reads stage k % 3; the DMA of step k + 2 is issued at the top of step k into the stage read at step k - 1 (every
//   wave is past that step's closing barrier); a counted s_waitcnt vmcnt(6) + barrier at the bottom of step k leaves exactly
//   that DMA in flight and makes step k + 1's data visible.  The DMA queue never drains inside a tile.
//   A image per stage: [16 chunks][16 rows][64 B], 16-byte pieces XOR-permuted by (row >> 2) & 3 on the source side
//   (conflict-free ds_read_b128 fragments); W image: the packed fragments as they are.  Accumulation order per output element
//   and the epilogues are those of the other two tile kernels (bitwise interchangeable).
// ================================================================================================================
#define T3_STAGE 24576
#define T3_LDS (3 * T3_STAGE)   // >= 2 epilogue regions of T2_REGION

template <int EPI, bool CONV = false>
__global__ __launch_bounds__(256, 2) void gemm_tile_4w_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char t3_sm[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wr = w >> 1, wc = w & 1;
    const int n_mt = (a.M + 255) / 256, n_nt = (a.N + 127) / 128;
    const int total = n_mt * n_nt, per = (total + 7) >> 3;
    const int t = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (t >= total) return;
    const int gq = t / (PF_GM * n_nt), first_m = gq * PF_GM;
    const int gm = (n_mt - first_m) < PF_GM ? (n_mt - first_m) : PF_GM;
    const int rr = t - gq * PF_GM * n_nt;
    const int bn = rr / gm, bm = first_m + (rr - bn * gm);
    const int m0 = bm * 256, nt0 = bn * 8;
    const int nkb = a.K >> 5;
    const int ntiles = (a.N + 15) >> 4;

    // staging sources of this lane: 4 A chunks (16 rows x 64 B each) and 2 W chunks per K step
    const char* asrc[4];
    const char* bsrc[2];
    int cv_t[4], cv_T[4];
    const char* cv_base[4];
    const char* cv_zero[4];
    const int cv_kpt = CONV ? a.conv_W >> 5 : 1;               // K steps per tap
    const int cv_left = CONV ? (a.conv_taps - 1) * a.conv_dil - ((a.conv_taps - 1) * a.conv_dil) / 2 : 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = lane >> 2, piece = (lane & 3) ^ ((row >> 2) & 3);
        int m = m0 + (w * 4 + i) * 16 + row;
        m = m < a.M ? m : a.M - 1;
        asrc[i] = (const char*)a.A + ((size_t)m * a.lda + piece * 8) * 2;
        if constexpr (CONV) {
            const int sq = a.tok_seq[m];
            cv_t[i] = a.tok_t[m];
            cv_T[i] = a.seq_T[sq];
            cv_base[i] = (const char*)a.A + ((size_t)a.seq_start[sq] * a.lda + piece * 8) * 2;
            cv_zero[i] = (const char*)a.zero_row + piece * 16;
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int nt = nt0 + w * 2 + i;
        nt = nt < ntiles ? nt : ntiles - 1;
        bsrc[i] = (const char*)a.Wp + (size_t)nt * nkb * 1024 + lane * 16;
    }
    auto issue = [&](int ks, char* stage) {
        int tap = 0, rem = ks;
        if constexpr (CONV) { tap = ks / cv_kpt; rem = ks - tap * cv_kpt; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const char* ap = asrc[i] + (size_t)ks * 64;
            if constexpr (CONV) {
                const int maxpad = cv_left;
                const int Tv = cv_T[i] <= maxpad ? maxpad + 1 : cv_T[i];
                int p = cv_t[i] + tap * a.conv_dil - cv_left;
                p = p < 0 ? -p : p;
                p = p >= Tv ? 2 * (Tv - 1) - p : p;
                const bool ok = p >= 0 && p < cv_T[i];
                ap = ok ? cv_base[i] + ((size_t)p * a.lda + (size_t)rem * 32) * 2 : cv_zero[i];
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ap,
                                             (__attribute__((address_space(3))) void*)(stage + (w * 4 + i) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc[i] + (size_t)ks * 1024),
                                             (__attribute__((address_space(3))) void*)(stage + 16384 + (w * 2 + i) * 1024), 16, 0, 0);
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int row16 = lane & 15, kg = lane >> 4;
    const int a_rd = wr * 8 * 1024 + row16 * 64 + ((kg ^ ((row16 >> 2) & 3)) << 4);
    const int b_rd = 16384 + wc * 4 * 1024 + lane * 16;

    char* s_cur = t3_sm;                      // stage read at this step
    char* s_nxt = t3_sm + T3_STAGE;           // step + 1
    char* s_far = t3_sm + 2 * T3_STAGE;       // step + 2 (read at step - 1)
    issue(0, s_cur);
    if (nkb > 1) { issue(1, s_nxt); t2_wait_vm<6>(); } else t2_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    for (int ks = 0; ks < nkb; ++ks) {
        const bool far = ks + 2 < nkb;                              // block-uniform
        if (far) issue(ks + 2, s_far);
        v4u af[8], bf[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bf[nt] = *(const v4u*)(s_cur + b_rd + nt * 1024);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) af[mt] = *(const v4u*)(s_cur + a_rd + mt * 1024);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[mt]), __builtin_bit_cast(bf16x8_t, bf[nt]),
                                                                      acc[mt][nt], 0, 0, 0);
        if (far) t2_wait_vm<6>(); else t2_wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        char* tmp = s_cur; s_cur = s_nxt; s_nxt = s_far; s_far = tmp;
    }
    // Epilogue: two rounds (m-tiles 0-3, 4-7 of every wave) through two [64 rows][128 cols] LDS regions, region wr written and
    // stored by waves 2 wr, 2 wr + 1.  (The last barrier of the loop has every wave past its fragment reads.)
    const int g = lane >> 4, c16 = lane & 15;
    float* ct = (float*)(t3_sm + wr * T2_REGION);
    const int rn0 = nt0 * 16;
    const bool v_region = EPI == EPI_QKV_ROPE && a.D % 128 == 0 && rn0 >= 2 * a.D;
    const int ltid = threadIdx.x & 127;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (v_region) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) *(f32x4*)(ct + (wc * 64 + nt * 16 + c16) * 68 + mt * 16 + g * 4) = acc[h * 4 + mt][nt];
        } else {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) ct[(mt * 16 + g * 4 + r) * 128 + ((wc * 64 + nt * 16 + c16) ^ (g << 4))] = acc[h * 4 + mt][nt][r];
        }
        const int rm0 = m0 + wr * 128 + h * 64;
        int* meta = (int*)((char*)ct + 32768);
        if (!v_region) pf_stage_meta<EPI, 64>(a, meta, rm0, ltid);
        __syncthreads();
        if (v_region) pf_store_vt<64, 128>(a, ct, rm0, rn0, ltid);
        else pf_store_tile<EPI, 64, 128>(a, ct, meta, rm0, rn0, ltid);
        if (h == 0) __syncthreads();
    }
}

template <int EPI, bool CONV = false>
static int launch_gemm_tile_4w_e(const GemmArgs& a, hipStream_t st) {
    const int n_mt = ceil_div(a.M, 256), n_nt = ceil_div(a.N, 128);
    const int per = ceil_div(n_mt * n_nt, 8);
    static ItPerDevice<bool> attr_set_pd;
    bool& attr_set = attr_set_pd.cur();
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)gemm_tile_4w_kernel<EPI, CONV>, hipFuncAttributeMaxDynamicSharedMemorySize, T3_LDS));
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_tile_4w_kernel<EPI, CONV>), dim3(per * 8), dim3(256), T3_LDS, st, a);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

// The 256 x 256 kernel holds one block per CU (136 KiB of LDS): it needs several rounds of tiles to amortise the tail, and the
// vector epilogue (which every shape of the engine has).  ITTS_TILE256=0 keeps the 128 x 128 kernel (A/B, bitwise equal output).
// Which tile kernel: 0 = 128 x 128 (four waves, two blocks per CU), 1 = 256 x 256 (eight waves, one block per CU), 2 = 256 x 128
// (four waves, two blocks per CU).  ITTS_TILE256 = 0 | 1 | 2 forces one (A/B; the three are bitwise interchangeable).
static int pick_tile_kernel(const GemmArgs& a) {
    const int mode = itts_opt(OPT_TILE256);
    if (mode == 0) return 0;
    if (!pf_vec_ok(a) || a.N % 128) return 0;
    if (mode == 1 || mode == 2) return mode;
    // The 256 x 256 kernel holds one block per CU: it needs whole rounds of tiles.  Measured (profiles/r02f..r02i): ahead of the
    // 128 x 128 kernel by 10-14 % on K >= 1280 at >= 2 rounds (GPT prefill at B = 64), by 3 % over the s2mel solve at >= 4 rounds
    // (-16 % on the residual / plain-store GEMMs, par on the rest), behind it below that (tail of the last round).
    const long long tiles = (long long)ceil_div(a.M, 256) * ceil_div(a.N, 256);
    if (tiles >= 1024 || (tiles >= 480 && a.K >= 1024)) return 1;
    return 0;
}

static int launch_gemm_prefill(const GemmArgs& a, hipStream_t st) {
    if (a.epi == EPI_GATE && a.conv_taps > 0 &&
        (a.conv_W % PF_BK || a.K != a.conv_taps * a.conv_W || a.lda != a.conv_W || !a.tok_seq || !a.tok_t || !a.seq_start || !a.seq_T || !a.zero_row)) {
        itts_set_error("gemm tap mode: need conv_W %% 64 == 0, K == taps * conv_W, lda == conv_W and the sequence tables");
        return ITTS_ERR_ARG;
    }
    const int tk = pick_tile_kernel(a);
#define ITTS_TILE_DISPATCH(FN)                                                                                                          \
    switch (a.epi) {                                                                                                                     \
        case EPI_STORE_F32: return FN<EPI_STORE_F32>(a, st);                                                                             \
        case EPI_RESIDUAL: return FN<EPI_RESIDUAL>(a, st);                                                                               \
        case EPI_GELU_ACT: return FN<EPI_GELU_ACT>(a, st);                                                                               \
        case EPI_QKV: return FN<EPI_QKV>(a, st);                                                                                         \
        case EPI_SWIGLU: return FN<EPI_SWIGLU>(a, st);                                                                                   \
        case EPI_GATE: return a.conv_taps > 0 ? FN<EPI_GATE, true>(a, st) : FN<EPI_GATE>(a, st);                                         \
        case EPI_QKV_ROPE: return FN<EPI_QKV_ROPE>(a, st);                                                                               \
        case EPI_WN_RS: return FN<EPI_WN_RS>(a, st);                                                                                     \
        default: break;                                                                                                                  \
    }
    if (tk == 1) { ITTS_TILE_DISPATCH(launch_gemm_tile256_e) }
    if (tk == 2) { ITTS_TILE_DISPATCH(launch_gemm_tile_4w_e) }
#undef ITTS_TILE_DISPATCH
    switch (a.epi) {
        case EPI_STORE_F32: return launch_gemm_prefill_e<EPI_STORE_F32>(a, st);
        case EPI_RESIDUAL: return launch_gemm_prefill_e<EPI_RESIDUAL>(a, st);
        case EPI_GELU_ACT: return launch_gemm_prefill_e<EPI_GELU_ACT>(a, st);
        case EPI_QKV: return launch_gemm_prefill_e<EPI_QKV>(a, st);
        case EPI_SWIGLU: return launch_gemm_prefill_e<EPI_SWIGLU>(a, st);
        case EPI_GATE:
            return a.conv_taps > 0 ? launch_gemm_prefill_e<EPI_GATE, true>(a, st) : launch_gemm_prefill_e<EPI_GATE>(a, st);
        case EPI_QKV_ROPE: return launch_gemm_prefill_e<EPI_QKV_ROPE>(a, st);
        case EPI_WN_RS: return launch_gemm_prefill_e<EPI_WN_RS>(a, st);
        default: itts_set_error("gemm prefill: unsupported epilogue %d", a.epi); return ITTS_ERR_ARG;
    }
}

// f32 instantiations of the 128 x 128 tile kernel (vector epilogue only: N, ldo, D multiples of 4).  Two blocks fit a CU (66 KiB of
// LDS each, 64 accumulator registers), so one block's epilogue overlaps the other's main loop.
// (A variant that loads the weight fragments straight into registers instead of staging them through LDS -- half the LDS-DMA pieces per
// K tile -- measured 8 % SLOWER on the s2mel shapes, 109 vs 119 TFLOP/s at M = 312 704, profiles/r03j: the LDS-staged kernel stays.)
template <int EPI, bool CONV = false>
static int launch_gemm_prefill_f32_e(const GemmArgs& a, hipStream_t st) {
    const int n_mt = ceil_div(a.M, PF_BM), n_nt = ceil_div(a.N, PF_BN);
    const int per = ceil_div(n_mt * n_nt, 8);
    static ItPerDevice<bool> attr_set_pd;
    bool& attr_set = attr_set_pd.cur();
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)gemm_prefill_kernel<EPI, CONV, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PF_LDS));
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_prefill_kernel<EPI, CONV, true, true>), dim3(per * 8), dim3(256), PF_LDS, st, a);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

static int launch_gemm_prefill_f32(const GemmArgs& a, hipStream_t st) {
    if (a.epi == EPI_GATE && a.conv_taps > 0 &&
        (a.conv_W % 32 || a.K != a.conv_taps * a.conv_W || a.lda != a.conv_W || !a.tok_seq || !a.tok_t || !a.seq_start || !a.seq_T || !a.zero_row)) {
        itts_set_error("gemm tap mode (f32): need conv_W %% 32 == 0, K == taps * conv_W, lda == conv_W and the sequence tables");
        return ITTS_ERR_ARG;
    }
    switch (a.epi) {
        case EPI_STORE_F32: return launch_gemm_prefill_f32_e<EPI_STORE_F32>(a, st);
        case EPI_RESIDUAL: return launch_gemm_prefill_f32_e<EPI_RESIDUAL>(a, st);
        case EPI_GELU_ACT: return launch_gemm_prefill_f32_e<EPI_GELU_ACT>(a, st);
        case EPI_QKV: return launch_gemm_prefill_f32_e<EPI_QKV>(a, st);
        case EPI_SWIGLU: return launch_gemm_prefill_f32_e<EPI_SWIGLU>(a, st);
        case EPI_GATE: return a.conv_taps > 0 ? launch_gemm_prefill_f32_e<EPI_GATE, true>(a, st) : launch_gemm_prefill_f32_e<EPI_GATE>(a, st);
        case EPI_QKV_ROPE: return launch_gemm_prefill_f32_e<EPI_QKV_ROPE>(a, st);
        case EPI_WN_RS: return launch_gemm_prefill_f32_e<EPI_WN_RS>(a, st);
        default: itts_set_error("gemm prefill (f32): unsupported epilogue %d", a.epi); return ITTS_ERR_ARG;
    }
}

// ================================================================================================================
// Decode GEMM, bf16, 16 / 32 / 64 activation rows per block (MT m-tiles; 64-row slices of larger batches).
//   Same decomposition as gemm_kernel<true,4,1,true> -- one 16-column n-tile per block, the block's K slice (<= 1280)
//   split over the 4 waves by k-block (w, w+4, ...), LDS reduce, shared epilogue -- but the activation slab no longer
//   goes global -> registers -> ds_write in two serial phases: the whole [64 rows][<=1280 k] slab (<= 160 KiB) is DMA'd
//   into LDS with global_load_lds_dwordx4 in ONE burst issued together with the wave's weight fragments, so every byte
//   the block needs is in flight before the first wait.  A image as in the prefill kernel: 1 KiB chunks of
//   [8 rows][128 B], 16-byte pieces XOR-permuted on the source side (bank-conflict-free fragment reads).
//   Accumulation order per output is unchanged -> bitwise equal to the register-staged kernel.
// ================================================================================================================
// Epilogue of the decode GEMM, specialised per epilogue kind (a runtime switch per element compiled to a branch tree with
// the bias and cache-position loads INSIDE it, each followed by vmcnt(0): several serial memory round trips per tile).  Here
// the per-lane bias and the cache position are fetched at kernel entry, together with the weight stream, and the tile's
// column-derived quantities (q / K / V selector, head, offset inside the head) are computed once per tile.
template <int EPI>
__device__ __forceinline__ void decode_epilogue(const GemmArgs& a, int mbase, int nbase, int lane, int z, f32x4 v, float bias, int pos) {
    const int n = nbase + (lane & 15);
    if (n >= a.N) return;
    const int mrow = mbase + (lane >> 4) * 4;
    if constexpr (EPI == EPI_PARTIAL) {
        float* dst = a.partial + ((size_t)z * a.M + mrow) * a.N + n;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (mrow + r < a.M) dst[(size_t)r * a.N] = v[r];
    } else if constexpr (EPI == EPI_STORE_F32) {
        float* dst = a.out_f32 + (size_t)mrow * a.ldo + n;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (mrow + r < a.M) dst[(size_t)r * a.ldo] = v[r] + bias;
    } else if constexpr (EPI == EPI_GELU_ACT) {
        u16* dst = (u16*)a.out_act + (size_t)mrow * a.ldo + n;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (mrow + r < a.M) dst[(size_t)r * a.ldo] = f32_to_bf16(gelu_new_f(v[r] + bias));
    } else {                                                       // EPI_QKV (the tile never straddles q | K | V: D % 16 == 0)
        const int which = nbase / a.D;                             // wave-uniform
        const int c = n - which * a.D;
        if (which == 0) {
            float* dst = a.qbuf + (size_t)mrow * a.D + c;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (mrow + r < a.M) dst[(size_t)r * a.D] = v[r] + bias;
        } else {
            u16* cache = (u16*)(which == 1 ? a.kcache : a.vcache);
            const int sm = a.seq_mul > 1 ? a.seq_mul : 1;
            const size_t head_off = (size_t)(c >> 6) * a.Tmax * 64 + (c & 63);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mrow + r;
                if (m < a.M) {
                    int b = m, si = 0;
                    if (a.S != 1) { b = m / a.S; si = m - b * a.S; }          // uniform; decode has S == 1
                    const size_t pb = a.seq_map ? (size_t)a.seq_map[b] : (size_t)b * sm;
                    const int own = pos + si - (a.pos_shift ? a.pos_shift[pb] : 0);          // the cache row's own position (admitted rows)
                    if (own < a.Tmax)                                                          // (past it: a finished row of a long-running session)
                        cache[pb * a.H * a.Tmax * 64 + head_off + (size_t)own * 64] = f32_to_bf16(v[r] + bias);
                }
            }
        }
    }
}

template <int NT, int MT, bool WNT, int EPI>     // WNT: non-temporal policy on the weight stream (each fragment is read by ONE block, once per step)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, MT == 4 ? 1 : 4))) void gemm_decode64_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];     // [kp][8 row groups][1 KiB]; reused for the reduction
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nkb = a.K >> 5;
    const int z = blockIdx.z;
    // the launcher takes this kernel only when K/32 divides evenly into nsplit slices of an even number of k-blocks; the
    // slice length comes from the host (a 64-bit division here costs ~300 scalar instructions before the first load)
    const int nkl = a.kb_slice;                                    // k-blocks in this block's slice (<= 40, even)
    const int kb_lo = z * nkl;
    const int kps = nkl >> 1;                                      // 128-byte k-pairs in the slice (<= 20)
    const int ntiles = (a.N + 15) >> 4;
    const int nt0 = blockIdx.x * NT;                               // NT n-tiles (16 columns each) per block
    constexpr int RG = 2 * MT;                                     // 8-row groups of the slab (MT m-tiles of 16 rows)
    const int m0 = blockIdx.y * (16 * MT);
    // epilogue operands, requested now: wave w finishes tiles w, w + 4, ... whose n-tile is always nt0 + (w mod NT)
    const int j_epi = w & (NT - 1);
    float bias_epi = 0.f;
    int pos_epi = 0;
    if constexpr (EPI != EPI_PARTIAL) {
        if (a.bias) {
            int nb = (nt0 + j_epi) * 16 + (lane & 15);
            nb = nb < a.N ? nb : a.N - 1;
            bias_epi = a.bias[nb];
        }
    }
    if constexpr (EPI == EPI_QKV) pos_epi = *a.pos_ptr;

    // weight fragments of this wave's k-blocks: straight to registers, all issued now
    v4u bq[10][NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int t = (nt0 + j) < ntiles ? nt0 + j : ntiles - 1;
        const v4u* wp = (const v4u*)a.Wp + ((size_t)t * nkb + kb_lo) * 64 + lane;
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const int kl = w + 4 * i;
            const bool ok = kl < nkl;
            const v4u* src = wp + (size_t)(ok ? kl : 0) * 64;
            v4u v = WNT ? __builtin_nontemporal_load(src) : *src;
            if (!ok) v = v4u{0u, 0u, 0u, 0u};
            bq[i][j] = v;
        }
    }
    // activation slab: chunk c = kp * RG + rg; wave w DMAs k-pairs [5w, 5w + 5) of all row groups
    {
        const char* arow[RG];
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) {
            const int row = rg * 8 + (lane >> 3), row16 = row & 15;
            const int piece = (lane & 7) ^ ((row16 >> 1) & 7);
            int m = m0 + row;
            m = m < a.M ? m : a.M - 1;
            arow[rg] = (const char*)a.A + ((size_t)m * a.lda + (size_t)kb_lo * 32 + piece * 8) * 2;
        }
        // a.dma_rot (default on; ITTS_DECODE_ROT=0 is the A/B switch): every block starts its sweep of the slab at a different k-pair, so
        // the CUs of an XCD do not walk the same L2 lines in lock step (same bytes, same LDS image, different issue order): 1.583 ->
        // 1.560 ms per token at 64 rows x 400 tokens, two alternating pairs (profiles/r03n/decode_rot.log)
        const int rot = a.dma_rot ? (int)(blockIdx.x % 5u) : 0;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            int jj = j + rot;
            jj = jj >= 5 ? jj - 5 : jj;
            const int kp = w * 5 + jj;
            if (kp < kps) {                                        // wave-uniform
#pragma unroll
                for (int rg = 0; rg < RG; ++rg)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(arow[rg] + (size_t)kp * 128),
                                                     (__attribute__((address_space(3))) void*)(dsm + (kp * RG + rg) * 1024), 16, 0, 0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[mt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int row16 = lane & 15, kg = lane >> 4;
    const int a_lane = (row16 >> 3) * 1024 + (row16 & 7) * 128;
    const int sw = (row16 >> 1) & 7;
    // Straight-line MFMA phase: the A fragments of k-block i+1 are read from LDS while k-block i feeds the MFMAs.  A k-block
    // past the slice (small K) re-reads the last valid one -- finite data against zeroed weights -- so there is no branch
    // per k-block (a branch keeps the compiler from hoisting the LDS reads: read, wait, 4 MFMAs, read, wait, ...).
    auto lds_frag = [&](int i, v4u (&af)[MT]) {
        int kl = w + 4 * i;
        kl = kl < nkl ? kl : nkl - 1;
        const int kp = kl >> 1, pos = (((kl & 1) << 2) + kg) ^ sw;
        const char* base = dsm + kp * (RG * 1024) + a_lane + pos * 16;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[mt] = *(const v4u*)(base + mt * 2048);
    };
    v4u af_cur[MT], af_nxt[MT];
    lds_frag(0, af_cur);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        if (i + 1 < 10) lds_frag(i + 1, af_nxt);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af_cur[mt]), __builtin_bit_cast(bf16x8_t, bq[i][j]),
                                                                     acc[mt][j], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af_cur[mt] = af_nxt[mt];
    }
    __syncthreads();                                               // every wave is done with the slab: reuse it
    f32x4* r4 = (f32x4*)dsm;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NT; ++j) r4[((size_t)w * (MT * NT) + mt * NT + j) * 64 + lane] = acc[mt][j];
    __syncthreads();
    for (int tile = w; tile < MT * NT; tile += 4) {                // MT * NT output tiles over the 4 waves
        f32x4 s = r4[(size_t)tile * 64 + lane];
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) {
            const f32x4 o = r4[((size_t)ww * (MT * NT) + tile) * 64 + lane];
            s[0] += o[0]; s[1] += o[1]; s[2] += o[2]; s[3] += o[3];
        }
        const int mt = tile / NT;                                   // tile - mt * NT == j_epi
        if (nt0 + j_epi < ntiles) decode_epilogue<EPI>(a, m0 + mt * 16, (nt0 + j_epi) * 16, lane, z, s, bias_epi, pos_epi);
    }
}



template <int NT, int MT, bool WNT, int EPI>
static int launch_gemm_decode64_e(const GemmArgs& a, int ntiles, size_t lds, hipStream_t st) {
    static ItPerDevice<int> attr_state_pd;                         // per device: 0 unknown, 1 ok, -1 the device refuses the LDS size
    int& attr_state = attr_state_pd.cur();
    if (attr_state == 0) {
        const hipError_t e = hipFuncSetAttribute((const void*)gemm_decode64_kernel<NT, MT, WNT, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 40960 * MT);
        attr_state = (e == hipSuccess) ? 1 : -1;
        if (e != hipSuccess) (void)hipGetLastError();
    }
    if (attr_state < 0) return -1;
    GemmArgs a2 = a;
    a2.kb_slice = (a.K / 32) / a.nsplit;                           // exact: the caller checked (K/32) % (2 * nsplit) == 0
    const int rot = itts_opt(OPT_DECODE_ROT);
    a2.dma_rot = rot;
    hipLaunchKernelGGL((gemm_decode64_kernel<NT, MT, WNT, EPI>), dim3(ceil_div(ntiles, NT), ceil_div(a.M, 16 * MT), a.nsplit), dim3(256), lds, st, a2);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

template <int NT, int MT, bool WNT>
static int launch_gemm_decode64_w(const GemmArgs& a, int ntiles, size_t lds, hipStream_t st) {
    switch (a.epi) {
        case EPI_STORE_F32: return launch_gemm_decode64_e<NT, MT, WNT, EPI_STORE_F32>(a, ntiles, lds, st);
        case EPI_GELU_ACT: return launch_gemm_decode64_e<NT, MT, WNT, EPI_GELU_ACT>(a, ntiles, lds, st);
        case EPI_PARTIAL: return launch_gemm_decode64_e<NT, MT, WNT, EPI_PARTIAL>(a, ntiles, lds, st);
        case EPI_QKV: return launch_gemm_decode64_e<NT, MT, WNT, EPI_QKV>(a, ntiles, lds, st);
        default: return -1;                                        // EPI_RESIDUAL is a prefill epilogue: register-path kernels
    }
}

template <int NT, int MT>
static int launch_gemm_decode64_nt(const GemmArgs& a, int ntiles, size_t lds, hipStream_t st) {
    // ITTS_DECODE_WNT=1: non-temporal policy on the weight stream (A/B switch; outputs are identical either way).  Measured
    // (profiles/r02a): no change at 64 rows (1.532 vs 1.536 ms/token), 6 % SLOWER at 8 rows (1.124 vs 1.056) -> default off.
    const bool wnt = itts_opt(OPT_DECODE_WNT) != 0;
    return wnt ? launch_gemm_decode64_w<NT, MT, true>(a, ntiles, lds, st) : launch_gemm_decode64_w<NT, MT, false>(a, ntiles, lds, st);
}

// (32-row blocks above 32 rows -- 80 KiB slabs, two blocks per CU, every weight tile streamed by ceil(M / 32) blocks -- measured 0.5-0.7 % SLOWER
// at 40 / 48 / 64 rows, profiles/r04n: closed.)
// n-tiles per block: the slab is 160 KiB of LDS, i.e. ONE block per CU, so a grid above 256 blocks runs in rounds; take the
// smallest NT (1, 2, 4) that fits the launch into a single round (more columns per block also amortise the slab DMA).
static int launch_gemm_decode64(const GemmArgs& a, hipStream_t st) {
    const int ntiles = (a.N + 15) / 16;
    const int slice_kb = ceil_div(a.K / 32, a.nsplit);
    if (a.M <= 32) {                                               // 16 / 32 rows: 40 / 80 KiB slabs, several blocks per CU
        const int mt = a.M <= 16 ? 1 : 2;
        size_t l2 = (size_t)((slice_kb + 1) / 2) * 2048 * mt;
        if (l2 < (size_t)mt * 4096) l2 = (size_t)mt * 4096;
        return mt == 1 ? launch_gemm_decode64_nt<1, 1>(a, ntiles, l2, st) : launch_gemm_decode64_nt<1, 2>(a, ntiles, l2, st);
    }
    size_t lds = (size_t)((slice_kb + 1) / 2) * 8192;
    const int other = ceil_div(a.M, 64) * a.nsplit;
    const int force_nt = itts_opt(OPT_DECODE_NT);
    int nt = 1;
    if (force_nt) nt = force_nt;
    else if (ntiles * other > 256) nt = (ceil_div(ntiles, 2) * other > 256) ? 4 : 2;
    if (lds < (size_t)nt * 16384) lds = (size_t)nt * 16384;        // reduction scratch: 4 waves x 4*NT tiles x 1 KiB
    switch (nt) {
        case 1: return launch_gemm_decode64_nt<1, 4>(a, ntiles, lds, st);
        case 2: return launch_gemm_decode64_nt<2, 4>(a, ntiles, lds, st);
        default: return launch_gemm_decode64_nt<4, 4>(a, ntiles, lds, st);
    }
}

// ================================================================================================================
// Decode GEMM with the LayerNorm in front of it fused into the operand staging, for at most 16 rows (one utterance, its beams, the
// 8-utterance shard a rank of the 8-GPU run decodes): at that size a token step is a chain of 172 launches of 4-6 us each and the two
// LayerNorm launches of a layer are a quarter of it.  Two kernels: this one for 1-4 rows (4 waves, one row and one quarter of the K split per
// wave), gemm_decode_lnw_kernel below for 5-16 rows.  Measured (profiles/r04a, r04h; ms per token at 560 tokens, fused / separate launches):
// 1 row 0.776 / 0.915, 4 rows 0.874 / 0.957, 5 rows 0.935 / 0.963, 8 rows 0.977 / 0.986, 12 rows 1.110 / 1.016, 16 rows 1.294 / 1.099 -- every
// block repeats the LayerNorm of every row (25 KB of L2 reads per row with the split-K partials), which outgrows the launch it removes above 8 rows.
//   Same decomposition as gemm_decode64_kernel<1, 1> (one 16-column n-tile per block, K <= 1280 split over the 4 waves by k-block, LDS
//   reduce, decode_epilogue), but the activation slab is not DMA'd from a LayerNorm kernel's output: wave w < M computes row w itself with
//   ln_row (ln_kernel's arithmetic: split-K reduce of the previous GEMM's 4 partials + its bias + residual, then the normalisation) and
//   writes the bf16 row into the slab image (row r of k-pair kp: chunk 2 kp, 128-byte row r & 7, 16-byte pieces XOR-permuted by (r >> 1) & 7
//   -- what the DMA of gemm_decode64_kernel produces), while its weight fragments are in flight.  Every block repeats the 1-4 rows of
//   LayerNorm (<= 30 KB of L2 reads per row); block 0 stores the updated residual to ln_x_out, a different buffer than ln_x (the other blocks
//   are still reading ln_x), and the host alternates the two.  Rows M..15 of the MFMA tile hold whatever the LDS held: MFMA rows are
//   independent and their outputs are never stored.  Same bf16 operands, same MFMAs in the same order -> bitwise the unfused path.
// ================================================================================================================
template <int NV, int EPI, int FLAGS, int NWV>       // NWV = 4 waves per block: wave w normalises row w and runs its quarter of the K split
                                                    // (1-4 rows; 5-16 rows: gemm_decode_lnw_kernel below)
__global__ __launch_bounds__(NWV * 64) void gemm_decode_ln_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];     // [kp][2 row groups][1 KiB]; reused for the reduction
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    constexpr int NKL = NV * 8;                                    // 32-wide k-blocks of the row (K = 256 NV)
    const int ntiles = (a.N + 15) >> 4;
    const int nt0 = blockIdx.x;
    float bias_epi = 0.f;
    int pos_epi = 0;
    if (a.bias) {
        int nb = nt0 * 16 + (lane & 15);
        nb = nb < a.N ? nb : a.N - 1;
        bias_epi = a.bias[nb];
    }
    if constexpr (EPI == EPI_QKV) pos_epi = *a.pos_ptr;
    // weight fragments of this wave's k-blocks (w, w + 4, ...): straight to registers, all issued now
    constexpr int NI = (NKL + 3) / 4;
    v4u bq[NI];
    if (NWV == 4 || w < 4) {
        const int t = nt0 < ntiles ? nt0 : ntiles - 1;
        const v4u* wp = (const v4u*)a.Wp + (size_t)t * NKL * 64 + lane;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int kl = w + 4 * i;
            const bool ok = kl < NKL;
            v4u v = *(wp + (size_t)(ok ? kl : 0) * 64);
            if (!ok) v = v4u{0u, 0u, 0u, 0u};
            bq[i] = v;
        }
    }
    for (int r = w; r < a.M; r += NWV) {                           // wave-uniform: this wave's row(s) -- one for <= NWV rows, two for 9-16 rows on 8 waves
        const int D = a.K;
        f32x4 v[NV];
        ln_row<NV, FLAGS>(a.ln_x + (size_t)r * D, (blockIdx.x == 0) ? a.ln_x_out + (size_t)r * D : (float*)nullptr,
                          a.ln_partial + (size_t)r * D, (size_t)a.M * D, a.ln_bias_prev, a.ln_g, a.ln_b, nullptr, nullptr, D, a.ln_eps, lane, v);
        char* row = dsm + (r >> 3) * 1024 + (r & 7) * 128 + ((((lane & 15) >> 1) ^ ((r >> 1) & 7)) << 4) + (lane & 1) * 8;
#pragma unroll
        for (int i = 0; i < NV; ++i) {                             // elements 4 lane + 256 i ..+3 = k-pair (lane >> 4) + 4 i, piece (lane & 15) >> 1
            uint2 pk;
            pk.x = (uint32_t)f32_to_bf16(v[i][0]) | ((uint32_t)f32_to_bf16(v[i][1]) << 16);
            pk.y = (uint32_t)f32_to_bf16(v[i][2]) | ((uint32_t)f32_to_bf16(v[i][3]) << 16);
            *(uint2*)(row + ((lane >> 4) + 4 * i) * 2048) = pk;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if (NWV == 4 || w < 4) {                                       // wave-uniform
        const int row16 = lane & 15, kg = lane >> 4;
        const int a_lane = (row16 >> 3) * 1024 + (row16 & 7) * 128;
        const int sw = (row16 >> 1) & 7;
        auto lds_frag = [&](int i) -> v4u {
            int kl = w + 4 * i;
            kl = kl < NKL ? kl : NKL - 1;
            const int kp = kl >> 1, pos = (((kl & 1) << 2) + kg) ^ sw;
            return *(const v4u*)(dsm + kp * 2048 + a_lane + pos * 16);
        };
        v4u af_cur = lds_frag(0), af_nxt = af_cur;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (i + 1 < NI) af_nxt = lds_frag(i + 1);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af_cur), __builtin_bit_cast(bf16x8_t, bq[i]), acc, 0, 0, 0);
            af_cur = af_nxt;
        }
    }
    __syncthreads();                                               // every wave is done with the slab: reuse it
    f32x4* r4 = (f32x4*)dsm;
    if (NWV == 4 || w < 4) r4[(size_t)w * 64 + lane] = acc;
    __syncthreads();
    if (w == 0) {
        f32x4 sacc = r4[lane];
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) {
            const f32x4 o = r4[(size_t)ww * 64 + lane];
            sacc[0] += o[0]; sacc[1] += o[1]; sacc[2] += o[2]; sacc[3] += o[3];
        }
        if (nt0 < ntiles) decode_epilogue<EPI>(a, 0, nt0 * 16, lane, 0, sacc, bias_epi, pos_epi);
    }
}

// The same fusion for 5-16 rows.  At that size the one-tile-per-block kernel above loses (profiles/r04a: every one of its 240-320 blocks repeats
// the LayerNorm of every row -- 25 KB of L2 reads per row with the split-K partials -- and ln_row's registers leave one 8-wave block per CU, so the
// launch runs in two rounds).  Here a block owns NT n-tiles (2 or 4: 60-160 blocks, one round, the redundant LayerNorm reads cut by NT) and the
// two jobs sit on different waves: waves 0-3 request their weight fragments (NT x 10 x 16 B per lane, all in flight at once) and later run the
// MFMAs with the usual 4-way K split, waves 4-7 normalise the rows (row w - 4, w, w + 4, ...: ln_row, the same bits as ln_kernel) into the slab
// image meanwhile.  Wave j < NT reduces and finishes tile j.  Same operands, same MFMA order per output -> bitwise the unfused path.
template <int NV, int EPI, int FLAGS, int NT>
__global__ __launch_bounds__(512) void gemm_decode_lnw_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];     // [kp][2 row groups][1 KiB]; reused for the reduction
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    constexpr int NKL = NV * 8;                                    // 32-wide k-blocks of the row (K = 256 NV)
    constexpr int NI = (NKL + 3) / 4;
    const int ntiles = (a.N + 15) >> 4;
    const int nt0 = blockIdx.x * NT;
    float bias_epi = 0.f;
    int pos_epi = 0;
    v4u bq[NI][NT];
    if (w < 4) {                                                   // wave-uniform
        if (w < NT && a.bias) {
            int nb = (nt0 + w) * 16 + (lane & 15);
            nb = nb < a.N ? nb : a.N - 1;
            bias_epi = a.bias[nb];
        }
        if constexpr (EPI == EPI_QKV) pos_epi = *a.pos_ptr;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int t = (nt0 + j) < ntiles ? nt0 + j : ntiles - 1;
            const v4u* wp = (const v4u*)a.Wp + (size_t)t * NKL * 64 + lane;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int kl = w + 4 * i;
                const bool ok = kl < NKL;
                v4u v = *(wp + (size_t)(ok ? kl : 0) * 64);
                if (!ok) v = v4u{0u, 0u, 0u, 0u};
                bq[i][j] = v;
            }
        }
    } else {
        for (int r = w - 4; r < a.M; r += 4) {
            const int D = a.K;
            f32x4 v[NV];
            ln_row<NV, FLAGS>(a.ln_x + (size_t)r * D, (blockIdx.x == 0) ? a.ln_x_out + (size_t)r * D : (float*)nullptr,
                              a.ln_partial + (size_t)r * D, (size_t)a.M * D, a.ln_bias_prev, a.ln_g, a.ln_b, nullptr, nullptr, D, a.ln_eps, lane, v);
            char* row = dsm + (r >> 3) * 1024 + (r & 7) * 128 + ((((lane & 15) >> 1) ^ ((r >> 1) & 7)) << 4) + (lane & 1) * 8;
#pragma unroll
            for (int i = 0; i < NV; ++i) {                         // elements 4 lane + 256 i ..+3 = k-pair (lane >> 4) + 4 i, piece (lane & 15) >> 1
                uint2 pk;
                pk.x = (uint32_t)f32_to_bf16(v[i][0]) | ((uint32_t)f32_to_bf16(v[i][1]) << 16);
                pk.y = (uint32_t)f32_to_bf16(v[i][2]) | ((uint32_t)f32_to_bf16(v[i][3]) << 16);
                *(uint2*)(row + ((lane >> 4) + 4 * i) * 2048) = pk;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (w < 4) {                                                   // wave-uniform
        const int row16 = lane & 15, kg = lane >> 4;
        const int a_lane = (row16 >> 3) * 1024 + (row16 & 7) * 128;
        const int sw = (row16 >> 1) & 7;
        auto lds_frag = [&](int i) -> v4u {
            int kl = w + 4 * i;
            kl = kl < NKL ? kl : NKL - 1;
            const int kp = kl >> 1, pos = (((kl & 1) << 2) + kg) ^ sw;
            return *(const v4u*)(dsm + kp * 2048 + a_lane + pos * 16);
        };
        v4u af_cur = lds_frag(0), af_nxt = af_cur;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (i + 1 < NI) af_nxt = lds_frag(i + 1);
#pragma unroll
            for (int j = 0; j < NT; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af_cur), __builtin_bit_cast(bf16x8_t, bq[i][j]), acc[j], 0, 0, 0);
            af_cur = af_nxt;
        }
    }
    __syncthreads();                                               // every wave is done with the slab: reuse it
    f32x4* r4 = (f32x4*)dsm;
    if (w < 4) {
#pragma unroll
        for (int j = 0; j < NT; ++j) r4[((size_t)w * NT + j) * 64 + lane] = acc[j];
    }
    __syncthreads();
    if (w < NT) {
        f32x4 sacc = r4[(size_t)w * 64 + lane];
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) {
            const f32x4 o = r4[((size_t)ww * NT + w) * 64 + lane];
            sacc[0] += o[0]; sacc[1] += o[1]; sacc[2] += o[2]; sacc[3] += o[3];
        }
        if (nt0 + w < ntiles) decode_epilogue<EPI>(a, 0, (nt0 + w) * 16, lane, 0, sacc, bias_epi, pos_epi);
    }
}

template <int NV, int EPI>
static int launch_gemm_decode_ln_e(const GemmArgs& a, hipStream_t st) {
    const int ntiles = (a.N + 15) / 16;
    size_t lds = (size_t)NV * 4 * 2048;                            // NV * 8 k-blocks = NV * 4 k-pairs of 2 KiB (two 8-row groups)
    if (lds < 4096) lds = 4096;                                    // reduction scratch: 4 waves x 1 KiB
    const bool part = a.ln_partial || a.ln_bias_prev;
    if (part && (!a.ln_partial || !a.ln_bias_prev || !a.ln_x_out || a.ln_x_out == a.ln_x)) { itts_set_error("gemm_decode_ln: partials need a bias and a separate ln_x_out"); return ITTS_ERR_ARG; }
#define LN_GEMM_LAUNCH(NWV_)                                                                                                             \
    do {                                                                                                                                 \
        if (part) hipLaunchKernelGGL((gemm_decode_ln_kernel<NV, EPI, LN_PARTIAL | LN_BIAS, NWV_>), dim3(ntiles), dim3(NWV_ * 64), lds, st, a); \
        else hipLaunchKernelGGL((gemm_decode_ln_kernel<NV, EPI, 0, NWV_>), dim3(ntiles), dim3(NWV_ * 64), lds, st, a);                     \
    } while (0)
    // 5-16 rows: the wide kernel; option decode_ln_nt = its n-tiles per block (2, the default, or 4)
    const int wide_nt = itts_opt(OPT_DECODE_LN_NT);
#define LNW_GEMM_LAUNCH(NT_)                                                                                                             \
    do {                                                                                                                                 \
        const dim3 grid(ceil_div(ntiles, NT_));                                                                                          \
        if (part) hipLaunchKernelGGL((gemm_decode_lnw_kernel<NV, EPI, LN_PARTIAL | LN_BIAS, NT_>), grid, dim3(512), lds, st, a);          \
        else hipLaunchKernelGGL((gemm_decode_lnw_kernel<NV, EPI, 0, NT_>), grid, dim3(512), lds, st, a);                                  \
    } while (0)
    if (a.M <= 4) LN_GEMM_LAUNCH(4);
    else {
        if (lds < 16384) lds = 16384;                               // reduction scratch: 4 waves x 4 tiles x 1 KiB
        if (wide_nt == 4) LNW_GEMM_LAUNCH(4); else LNW_GEMM_LAUNCH(2);
    }
#undef LNW_GEMM_LAUNCH
#undef LN_GEMM_LAUNCH
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

// shapes the fused kernel takes: bf16, 1-16 rows, K = model_dim in {256, 512, 1280}, one K slice, QKV / GELU epilogue (plain store: the unit op)
bool gemm_decode_ln_ok(int M, int K, int epi) {
    return M >= 1 && M <= 16 && (K == 256 || K == 512 || K == 1280) && (epi == EPI_QKV || epi == EPI_GELU_ACT || epi == EPI_STORE_F32);
}

int launch_gemm_decode_ln(const GemmArgs& a, hipStream_t st) {
    if (!gemm_decode_ln_ok(a.M, a.K, a.epi) || a.nsplit != 1 || !a.ln_x || !a.ln_g || !a.ln_b) {
        itts_set_error("gemm_decode_ln: unsupported call (M=%d K=%d epi=%d nsplit=%d)", a.M, a.K, a.epi, a.nsplit);
        return ITTS_ERR_ARG;
    }
#define LN_GEMM_CASE(NV_)                                                                                        \
    case NV_: return a.epi == EPI_QKV ? launch_gemm_decode_ln_e<NV_, EPI_QKV>(a, st) : a.epi == EPI_GELU_ACT ? launch_gemm_decode_ln_e<NV_, EPI_GELU_ACT>(a, st) \
                                                                                                               : launch_gemm_decode_ln_e<NV_, EPI_STORE_F32>(a, st);
    switch (a.K / 256) { LN_GEMM_CASE(1) LN_GEMM_CASE(2) LN_GEMM_CASE(5) default: break; }
#undef LN_GEMM_CASE
    return ITTS_ERR_ARG;
}

template <bool BF16, int MT, int NT, bool KSPLIT>
static int launch_gemm_cfg(const GemmArgs& a, hipStream_t st) {
    const int ntiles = (a.N + 15) / 16;
    const int per_block = KSPLIT ? NT : 4 * NT;
    dim3 grid(ceil_div(ntiles, per_block), ceil_div(a.M, MT * 16), a.nsplit);
    const size_t lds = KSPLIT ? (size_t)4 * MT * NT * 64 * 16 + (MT >= 4 ? (size_t)MT * 16 * (20 * 64 + 16) : 0) : 0;
    hipLaunchKernelGGL((gemm_kernel<BF16, MT, NT, KSPLIT>), grid, dim3(256), lds, st, a);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

template <bool BF16>
static int launch_gemm_t(const GemmArgs& a, bool prefill, hipStream_t st) {
    if (prefill) {
        // bf16, K a multiple of the 64-deep K tile, 16-byte aligned rows: the LDS-DMA tile kernel; else the direct-load one
        const bool old_path = itts_opt(OPT_PREFILL_GEMM) == 0;
        if (BF16 && !old_path && a.K % PF_BK == 0 && a.lda % 8 == 0 && a.nsplit == 1 && a.epi != EPI_PARTIAL) return launch_gemm_prefill(a, st);
        if constexpr (!BF16) {
            // f32: the LDS-DMA tile kernel on the f32 MFMA (bitwise the register-path kernel's results; ITTS_F32_TILE=0 forces the
            // latter for the plain epilogues -- the A/B switch of tests/test_gpu_gpt.py / test_gpu_s2mel.py)
            const bool f32_reg = itts_opt(OPT_F32_TILE) == 0;
            const bool fused = a.epi > EPI_QKV || a.out_act2 != nullptr;
            if (pf_f32_ok(a) && (fused || !f32_reg)) return launch_gemm_prefill_f32(a, st);
        }
        if (a.epi > EPI_QKV || a.out_act2) { itts_set_error("gemm: epilogue %d / shadow output needs a tile kernel (bf16: K %% 64 == 0, lda %% 8 == 0; f32: K %% 32 == 0, lda %% 4 == 0, N / ldo / D %% 4 == 0)", a.epi); return ITTS_ERR_ARG; }
        return launch_gemm_cfg<BF16, 8, 2, false>(a, st);
    }
    if constexpr (BF16) {
        // bf16 decode: the LDS-DMA slab kernel whenever the K slice fits its image (ITTS_DECODE_GEMM=0: the register-path kernels)
        const bool old_path = itts_opt(OPT_DECODE_GEMM) == 0;
        // every K slice must hold an EVEN number of 32-wide k-blocks: the slab is DMA'd in 128-byte k-pairs and a slice with
        // an odd count would pair its last k-block with bytes of the neighbouring one (D = 128, 384, 640, ... with 4 slices)
        if (!old_path && ceil_div(a.K / 32, a.nsplit) <= 40 && a.lda % 8 == 0 && (a.K / 32) % (2 * a.nsplit) == 0) {
            const int rc = launch_gemm_decode64(a, st);
            if (rc >= 0) return rc;
        }
    }
    if (a.M <= 16) return launch_gemm_cfg<BF16, 1, 1, true>(a, st);
    if (a.M <= 32) return launch_gemm_cfg<BF16, 2, 1, true>(a, st);
    return launch_gemm_cfg<BF16, 4, 1, true>(a, st);
}

// diagnostics: resident blocks per CU the runtime predicts for the tile GEMM kernels at their launch configuration (residual epilogue)
int gemm_tile_occupancy(int prec, int* blocks) {
    int n = 0;
    hipError_t e;
    if (prec == PREC_F32) {
        (void)hipFuncSetAttribute((const void*)gemm_prefill_kernel<EPI_RESIDUAL, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PF_LDS);
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)gemm_prefill_kernel<EPI_RESIDUAL, false, true, true>, 256, PF_LDS);
    } else if (prec == PREC_F32X3) {
        return gemm_x3_occupancy(blocks);
    } else {
        (void)hipFuncSetAttribute((const void*)gemm_prefill_kernel<EPI_RESIDUAL, false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, PF_LDS);
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)gemm_prefill_kernel<EPI_RESIDUAL, false, true, false>, 256, PF_LDS);
    }
    if (e != hipSuccess) { itts_set_error("occupancy query: %s", hipGetErrorString(e)); return ITTS_ERR_HIP; }
    *blocks = n;
    return ITTS_OK;
}

int launch_gemm(const GemmArgs& a, int prec, bool prefill, hipStream_t st) {
    if (a.M <= 0 || a.N <= 0) return ITTS_OK;
    const int KB = prec == PREC_F32 ? 16 : 32;
    if (a.K % KB || a.nsplit < 1 || (a.K / KB) < a.nsplit) {
        itts_set_error("gemm: K=%d must be a multiple of %d and >= nsplit=%d blocks", a.K, KB, a.nsplit);
        return ITTS_ERR_ARG;
    }
    if (a.nsplit > 1 && a.epi != EPI_PARTIAL) { itts_set_error("gemm: split-K needs EPI_PARTIAL"); return ITTS_ERR_ARG; }
    if (prec == PREC_F32X3) return launch_gemm_x3(a, st);           // f32 operands as three bf16 planes (weights packed for it): no other kernel reads that image
    return prec == PREC_BF16 ? launch_gemm_t<true>(a, prefill, st) : launch_gemm_t<false>(a, prefill, st);
}

// ================================================================================================================
// Attention of one query against the KV cache (decode: nq = 1; prefill / latent pass: one block per query).
//   keys pad[b] .. pos0+qi, softmax in f32, exact skip of left-pad keys (additive finfo.min mask == weight 0).
// LPK lanes share one key row (16 B each): bf16 8 lanes x 8 dims, f32 16 lanes x 4 dims; a wave-load covers 64 / LPK keys.
//
// The arithmetic is defined over 16 CANONICAL KEY STREAMS, independent of the launch geometry: the keys of a query are cut into
// chunks of 64 (chunk c = keys first + 64 c ...), stream r owns chunks r, r + 16, r + 32, ...; inside a stream every lane group
// runs an online softmax over its keys in ascending order, the lane groups of the stream are merged by a fixed butterfly, and the 16
// stream results are merged flat, r = 0 .. 15, by the block's first wave.  A block of NW waves (4, 8 or 16) hosts streams w, w + NW,
// ... on wave w -- every geometry performs the same operations on the same operands in the same order, so the output bits do not
// depend on NW, on the batch size that picked it, or on where a row sits in the batch (tests/test_gpu_gpt.py: row alone == row in
// the 64-row batch).
//   Why: the first version (one online softmax per lane group over ALL keys, 4 waves) is a chain of one dependent memory round trip
// per 32 keys of the block; at 1-16 rows the launch has < 1 block per CU, nothing hides that chain, and the kernel cost 9-25 us per
// layer (the largest launch of a small-batch token step, profiles/r03x/decode_step_timeline_b1.txt).  Here a wave issues the K and
// V loads of 8 wave-loads (a whole bf16 chunk) before it touches the first, and at small batches 16 waves put up to 1024 keys of a
// (row, head) in flight at once: one round trip for any context the model allows.
// ================================================================================================================
template <bool BF16>
__device__ __forceinline__ float exp_sel(float x) { return BF16 ? __expf(x) : expf(x); }

#define ATTN_STREAMS 16
template <bool BF16, int NW, bool RMAP>      // RMAP: the beam search's row map is in use (a template flag: a run-time `rmap ? load : pb` costs a
                                            // branch and a vmcnt(0) in front of every group's loads, which serialises them again)
__global__ __launch_bounds__(NW * 64) void attn_kernel(AttnArgs a) {
    constexpr int LPK = BF16 ? 8 : 16;      // lanes per key
    constexpr int DPL = 64 / LPK;           // dims per lane
    constexpr int KPW = 64 / LPK;           // keys per wave-load
    constexpr int CH = 64;                  // keys per chunk
    constexpr int LPC = CH / KPW;           // wave-loads per chunk (8 / 16)
    constexpr int UB = 8;                   // wave-loads in flight per batch
    __shared__ float sm_m[ATTN_STREAMS], sm_l[ATTN_STREAMS], sm_acc[ATTN_STREAMS][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H;
    const int qi = blockIdx.y;
    const int sm = a.seq_mul > 1 ? a.seq_mul : 1;
    const int pb = a.seq_map ? a.seq_map[b] : b * sm;              // physical cache row / pad entry of this sequence
    int last = *a.pos_ptr + qi - (a.pos_shift ? a.pos_shift[pb] : 0);
    last = last < a.Tmax ? last : a.Tmax - 1;                      // (a finished row of a long-running session stays inside its cache row)
    const int first = a.pad ? a.pad[pb] : 0;
    const int sub = lane % LPK, grp = lane / LPK;
    const size_t qrow = (size_t)b * a.nq + qi;
    const int* rmap = a.row_map;
    if constexpr (RMAP) {
        if (a.row_map_alt && a.step_ptr && (*a.step_ptr & 1)) rmap = a.row_map_alt;
    }

    float q[DPL];
#pragma unroll
    for (int d = 0; d < DPL; ++d) q[d] = a.qbuf[qrow * a.D + h * 64 + sub * DPL + d];

    for (int r = w; r < ATTN_STREAMS; r += NW) {                   // the streams this wave hosts
        float m_run = -INFINITY, l_run = 0.f, acc[DPL];
#pragma unroll
        for (int d = 0; d < DPL; ++d) acc[d] = 0.f;
        for (int c = r; first + c * CH <= last; c += ATTN_STREAMS) {
            const int tb = first + c * CH;
#pragma unroll
            for (int h0 = 0; h0 < LPC; h0 += UB) {
                if (tb + h0 * KPW > last) break;                   // wave-uniform: the rest of the chunk is past the query
                bool okv[UB];
                int tcv[UB], prowv[UB];
                v4u kraw[UB], vraw[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int t = tb + (h0 + u) * KPW + grp;
                    okv[u] = t <= last;
                    tcv[u] = okv[u] ? t : last;
                    if constexpr (RMAP) prowv[u] = rmap[(size_t)b * a.Tmax + tcv[u]];
                    else prowv[u] = pb;
                }
#pragma unroll
                for (int u = 0; u < UB; ++u) {                     // every K / V load of the batch before the first use
                    const size_t off = (((size_t)prowv[u] * a.H + h) * a.Tmax + tcv[u]) * 64 + sub * DPL;
                    if constexpr (BF16) {
                        kraw[u] = *(const v4u*)((const u16*)a.kcache + off);
                        vraw[u] = *(const v4u*)((const u16*)a.vcache + off);
                    } else {
                        kraw[u] = *(const v4u*)((const float*)a.kcache + off);
                        vraw[u] = *(const v4u*)((const float*)a.vcache + off);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);                 // keep hipcc from sinking the later loads below the first use
#pragma unroll
                for (int u = 0; u < UB; ++u) {                     // then the keys in ascending order
                    float kf[DPL], vf[DPL];
                    if constexpr (BF16) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            kf[2 * i] = __uint_as_float(kraw[u][i] << 16);
                            kf[2 * i + 1] = __uint_as_float(kraw[u][i] & 0xffff0000u);
                            vf[2 * i] = __uint_as_float(vraw[u][i] << 16);
                            vf[2 * i + 1] = __uint_as_float(vraw[u][i] & 0xffff0000u);
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) { kf[i] = __uint_as_float(kraw[u][i]); vf[i] = __uint_as_float(vraw[u][i]); }
                    }
                    float s = 0.f;
#pragma unroll
                    for (int d = 0; d < DPL; ++d) s = fmaf(q[d], kf[d], s);
#pragma unroll
                    for (int o = 1; o < LPK; o <<= 1) s += __shfl_xor(s, o, 64);
                    s *= 0.125f;    // / sqrt(64)
                    if (okv[u]) {
                        const float nm = fmaxf(m_run, s);
                        const float sc = exp_sel<BF16>(m_run - nm);      // m_run = -inf -> 0
                        const float p = exp_sel<BF16>(s - nm);
                        l_run = l_run * sc + p;
#pragma unroll
                        for (int d = 0; d < DPL; ++d) acc[d] = acc[d] * sc + p * vf[d];
                        m_run = nm;
                    }
                }
            }
        }
        // merge the KPW lane groups of the stream
#pragma unroll
        for (int o = LPK; o < 64; o <<= 1) {
            const float om = __shfl_xor(m_run, o, 64), ol = __shfl_xor(l_run, o, 64);
            const float nm = fmaxf(m_run, om);
            const float sa = (m_run == -INFINITY) ? 0.f : exp_sel<BF16>(m_run - nm);
            const float sb = (om == -INFINITY) ? 0.f : exp_sel<BF16>(om - nm);
            l_run = l_run * sa + ol * sb;
#pragma unroll
            for (int d = 0; d < DPL; ++d) {
                const float oa = __shfl_xor(acc[d], o, 64);
                acc[d] = acc[d] * sa + oa * sb;
            }
            m_run = nm;
        }
        if (lane < LPK) {
            sm_m[r] = m_run;
            sm_l[r] = l_run;
#pragma unroll
            for (int d = 0; d < DPL; ++d) sm_acc[r][sub * DPL + d] = acc[d];
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {                                        // flat merge of the 16 streams, r ascending
        const int d = threadIdx.x;
        float nm = sm_m[0];
#pragma unroll
        for (int r = 1; r < ATTN_STREAMS; ++r) nm = fmaxf(nm, sm_m[r]);
        float l = 0.f, o = 0.f;
#pragma unroll
        for (int r = 0; r < ATTN_STREAMS; ++r) {
            const float sc = (sm_m[r] == -INFINITY) ? 0.f : exp_sel<BF16>(sm_m[r] - nm);
            l += sm_l[r] * sc;
            o += sm_acc[r][d] * sc;
        }
        const float res = l > 0.f ? o / l : 0.f;
        const size_t oo = qrow * a.D + h * 64 + d;
        if (BF16) ((u16*)a.out)[oo] = f32_to_bf16(res);
        else ((float*)a.out)[oo] = res;
    }
}

// ================================================================================================================
// Causal attention of MANY queries per sequence against the KV cache on the matrix pipe (prefill, the teacher-forced latent pass, any S > 1 pass
// without a beam row map): one block = (sequence, head, 64 consecutive queries), wave w = 16 of them, flash-style online softmax over 64-key tiles.
// The reference's counterpart is flash_attn_varlen_func (accel/attention.py:132-141) / the eager causal attention of transformers_gpt2.py:591-667.
//   S^T = K Q^T (keys as MFMA rows): a lane then owns 4 keys x 1 query, so the per-query maximum / sum is a reduction over its own registers and two
//   shuffles, and P^T leaves the accumulators already in B-operand order for O^T = V^T P^T.  K fragments come straight from the cache rows (16-byte
//   pieces of a key's 64 dims); V is staged per tile into LDS transposed ([d][key]) so that a V^T fragment is two 8-byte reads.  Nothing but the
//   attention output is written: K / V are in the cache from the wqkv epilogue.  Left padding: keys pad[b] .. pos0 + qi, exactly as attn_kernel;
//   a query left of its sequence's first key (a pad position) gets 0, as there.
//   BF16 (cache bf16): v_mfma_f32_16x16x32_bf16.  The f32 queries (scaled by 1/8, exact) and the probabilities are carried as bf16 hi + lo pairs -- two
//   MFMAs each, 16 significant bits -- so the only rounding that differs from attn_kernel's f32 FMAs is the accumulation order.
//   F32 (cache f32, the parity mode): v_mfma_f32_16x16x4_f32 on exact f32 operands; V through LDS row-major.
// ================================================================================================================
#define APF_VROW 68          // bf16 V^T image: [64 d][68] (64 keys + 4 pad: 136-byte rows, 8-byte aligned fragment reads)
#define APF_VROW32 65        // f32 V image: [64 keys][65]

template <bool BF16>
__global__ __launch_bounds__(256) void attn_prefill_mfma_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char vsm[BF16 ? 64 * APF_VROW * 2 : 64 * APF_VROW32 * 4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H;
    const int sm = a.seq_mul > 1 ? a.seq_mul : 1;
    const int pb = a.seq_map ? a.seq_map[b] : b * sm;
    const int first = a.pad ? a.pad[pb] : 0;
    const int pos0 = *a.pos_ptr;
    const int qb0 = blockIdx.y * 64;                                  // first query of the block
    const int q_hi_blk = (qb0 + 63 < a.nq ? qb0 + 63 : a.nq - 1);
    const int kmax_blk = pos0 + q_hi_blk;                             // last key any query of the block sees (a written cache row)
    const int qw0 = qb0 + w * 16;
    const int kmax_w = pos0 + (qw0 + 15 < a.nq ? qw0 + 15 : a.nq - 1);
    const bool wave_on = qw0 < a.nq;
    const int qi = qw0 + c16;
    const bool q_ok = qi < a.nq;
    const int last_q = pos0 + qi;
    const size_t qrow = (size_t)b * a.nq + (q_ok ? qi : a.nq - 1);
    const size_t head_base = ((size_t)pb * a.H + h) * a.Tmax;        // cache row index of key 0 of this (sequence, head)

    float m_run = -INFINITY, l_run = 0.f;
    f32x4 acc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // the query operand: lane (query c16, k-group g)
    v4u qh[2], ql[2];
    f32x4 q4[4];
    if constexpr (BF16) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const float* qp = a.qbuf + qrow * a.D + h * 64 + 32 * ks + 8 * g;
            const f32x4 x0 = *(const f32x4*)qp, x1 = *(const f32x4*)(qp + 4);
            const float x[8] = {x0[0] * 0.125f, x0[1] * 0.125f, x0[2] * 0.125f, x0[3] * 0.125f, x1[0] * 0.125f, x1[1] * 0.125f, x1[2] * 0.125f, x1[3] * 0.125f};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t hi = pf_cvt2(x[2 * i], x[2 * i + 1]);
                qh[ks][i] = hi;
                ql[ks][i] = pf_cvt2(x[2 * i] - __uint_as_float(hi << 16), x[2 * i + 1] - __uint_as_float(hi & 0xffff0000u));
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            q4[c] = *(const f32x4*)(a.qbuf + qrow * a.D + h * 64 + 16 * c + 4 * g);
#pragma unroll
            for (int i = 0; i < 4; ++i) q4[c][i] *= 0.125f;
        }
    }

    for (int t0 = first; t0 <= kmax_blk; t0 += 64) {
        __syncthreads();                                              // the previous tile's V image has been read by every wave
        // ---- stage V of keys t0 .. t0 + 63 (rows past the block's last key are clamped: finite data, weight 0) ----
        if constexpr (BF16) {
            u16* vt = (u16*)vsm;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int key = p * 32 + (threadIdx.x >> 3), dg = threadIdx.x & 7;
                int t = t0 + key;
                t = t < kmax_blk ? t : kmax_blk;
                const v4u raw = *(const v4u*)((const u16*)a.vcache + (head_base + t) * 64 + dg * 8);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    vt[(dg * 8 + 2 * i) * APF_VROW + key] = (u16)(raw[i] & 0xffffu);
                    vt[(dg * 8 + 2 * i + 1) * APF_VROW + key] = (u16)(raw[i] >> 16);
                }
            }
        } else {
            float* v32 = (float*)vsm;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int key = p * 16 + (threadIdx.x >> 4), dq = threadIdx.x & 15;
                int t = t0 + key;
                t = t < kmax_blk ? t : kmax_blk;
                const f32x4 raw = *(const f32x4*)((const float*)a.vcache + (head_base + t) * 64 + dq * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) v32[key * APF_VROW32 + dq * 4 + i] = raw[i];
            }
        }
        const bool tile_on = wave_on && t0 <= kmax_w;                 // wave-uniform
        f32x4 st[4];
        float sc = 1.f;
        if (tile_on) {
            // ---- S^T = K Q^T for the four 16-key tiles ----
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
                int t = t0 + 16 * kt + c16;
                t = t < kmax_blk ? t : kmax_blk;
                if constexpr (BF16) {
                    const u16* kp = (const u16*)a.kcache + (head_base + t) * 64 + 8 * g;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const v4u ka = *(const v4u*)(kp + 32 * ks);
                        st[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ka), __builtin_bit_cast(bf16x8_t, ql[ks]), st[kt], 0, 0, 0);
                        st[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ka), __builtin_bit_cast(bf16x8_t, qh[ks]), st[kt], 0, 0, 0);
                    }
                } else {
                    const float* kp = (const float*)a.kcache + (head_base + t) * 64 + 4 * g;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const f32x4 k4 = *(const f32x4*)(kp + 16 * c);
#pragma unroll
                        for (int i = 0; i < 4; ++i) st[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(k4[i], q4[c][i], st[kt], 0, 0, 0);
                    }
                }
            }
            // ---- mask, online softmax per query column ----
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int t = t0 + 16 * kt + 4 * g + r;
                    const bool ok = q_ok && t >= first && t <= last_q;
                    st[kt][r] = ok ? st[kt][r] : -INFINITY;
                    mx = fmaxf(mx, st[kt][r]);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float nm = fmaxf(m_run, mx);
            const float mu = nm == -INFINITY ? 0.f : nm;               // a column with no key yet: every exponent below is exp(-inf) = 0
            sc = exp_sel<BF16>(m_run - mu);
            float ps = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { st[kt][r] = exp_sel<BF16>(st[kt][r] - mu); ps += st[kt][r]; }
            l_run = l_run * sc + ps;                                   // this lane's keys only; the lane groups are added at the end
            m_run = nm;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[dt][r] *= sc;
        }
        __syncthreads();                                              // the V image is complete
        if (tile_on) {
            // ---- O^T += V^T P^T ----
            if constexpr (BF16) {
                const u16* vt = (const u16*)vsm;
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    v4u ph, pl;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const f32x4& s = st[2 * jp + (i >> 1)];
                        const float x0 = s[2 * (i & 1)], x1 = s[2 * (i & 1) + 1];
                        const uint32_t hi = pf_cvt2(x0, x1);
                        ph[i] = hi;
                        pl[i] = pf_cvt2(x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u));
                    }
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        const u16* vp = vt + (16 * dt + c16) * APF_VROW + 32 * jp + 4 * g;
                        const v2u_t lo2 = *(const v2u_t*)vp, hi2 = *(const v2u_t*)(vp + 16);
                        const v4u va{lo2.x, lo2.y, hi2.x, hi2.y};
                        acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, va), __builtin_bit_cast(bf16x8_t, pl), acc[dt], 0, 0, 0);
                        acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, va), __builtin_bit_cast(bf16x8_t, ph), acc[dt], 0, 0, 0);
                    }
                }
            } else {
                const float* v32 = (const float*)vsm;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt)
                            acc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(v32[(16 * kt + 4 * g + r) * APF_VROW32 + 16 * dt + c16], st[kt][r], acc[dt], 0, 0, 0);
            }
        }
    }
    if (!q_ok) return;
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
    const size_t oo = qrow * a.D + h * 64 + 4 * g;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        const f32x4 o{acc[dt][0] * inv, acc[dt][1] * inv, acc[dt][2] * inv, acc[dt][3] * inv};
        if constexpr (BF16) *(v2u_t*)((u16*)a.out + oo + 16 * dt) = v2u_t{(uint32_t)f32_to_bf16(o[0]) | ((uint32_t)f32_to_bf16(o[1]) << 16),
                                                                         (uint32_t)f32_to_bf16(o[2]) | ((uint32_t)f32_to_bf16(o[3]) << 16)};
        else *(f32x4*)((float*)a.out + oo + 16 * dt) = o;
    }
}

template <bool BF16, int NW>
static void launch_attention_nw(const AttnArgs& a, dim3 grid, hipStream_t st) {
    if (a.row_map) hipLaunchKernelGGL((attn_kernel<BF16, NW, true>), grid, dim3(NW * 64), 0, st, a);
    else hipLaunchKernelGGL((attn_kernel<BF16, NW, false>), grid, dim3(NW * 64), 0, st, a);
}

int launch_attention(const AttnArgs& a, int prec, hipStream_t st) {
    if (a.nseq <= 0 || a.nq <= 0) return ITTS_OK;
    if (a.D != a.H * 64) { itts_set_error("attention: head_dim must be 64 (D=%d H=%d)", a.D, a.H); return ITTS_ERR_ARG; }
    if (a.nq > 65535) { itts_set_error("attention: more than 65535 queries per sequence"); return ITTS_ERR_ARG; }
    // S > 1 passes without a beam row map: the causal MFMA kernel (option prefill_attn: -1 = the bf16 mode only -- the f32 parity mode keeps the
    // canonical-stream kernel, whose arithmetic is bit-identical between a prefill and the decode steps that follow it --, 0 never, 1 both precisions)
    const int pa = itts_opt(OPT_PREFILL_ATTN);
    if (a.nq > 1 && !a.row_map && (pa == 1 || (pa < 0 && prec == PREC_BF16)) && (a.D % 4) == 0) {
        dim3 gridp(a.nseq * a.H, (a.nq + 63) / 64);
        if (prec == PREC_BF16) hipLaunchKernelGGL((attn_prefill_mfma_kernel<true>), gridp, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((attn_prefill_mfma_kernel<false>), gridp, dim3(256), 0, st, a);
        HIP_TRY(hipGetLastError());
        return ITTS_OK;
    }
    dim3 grid(a.nseq * a.H, a.nq);
    // waves per block (option attn_waves forces 4 / 8 / 16; every choice gives the same bits).  Measured (profiles/r04a/decode_bench.log,
    // ms per token at 560 tokens, 16 / 8 / 4 waves): 1 row 0.776 / 0.789 / 0.846, 8 rows 1.056 / 1.065 / 1.117, 16 rows 1.376 / 1.331 / 1.356,
    // 32 rows 1.389 / 1.291 / 1.328, 64 rows 1.782 / 1.612 / 1.639; prefill (one block per query) 66 / 45 / 36 ms at 64 rows -> decode
    // launches of at most one block per CU take 16 waves (one round trip for the whole context), larger decode launches 8, prefill 4.
    const long long blocks = (long long)grid.x * grid.y;
    int nw = itts_opt(OPT_ATTN_WAVES);
    if (nw != 4 && nw != 8 && nw != 16) nw = a.nq > 1 ? 4 : blocks <= 256 ? 16 : 8;
    if (prec == PREC_BF16) { if (nw == 16) launch_attention_nw<true, 16>(a, grid, st); else if (nw == 8) launch_attention_nw<true, 8>(a, grid, st); else launch_attention_nw<true, 4>(a, grid, st); }
    else { if (nw == 16) launch_attention_nw<false, 16>(a, grid, st); else if (nw == 8) launch_attention_nw<false, 8>(a, grid, st); else launch_attention_nw<false, 4>(a, grid, st); }
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

// ================================================================================================================
// Token selection for one step: repetition penalty -> [temperature -> top-k -> top-p -> inverse-CDF draw] | argmax,
// finished rows forced to the pad (= stop) token, seen-set / finished update, and the next step's input embedding.
// One block per row.  Processor order and domains: generation_utils.py:900-901,1035-1044; sampling :3247-3256.
// ================================================================================================================
#define SAMPLE_CAP 128

__device__ __forceinline__ uint32_t f2key(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ double rng_uniform(unsigned long long seed, unsigned long long a, unsigned long long b) {
    unsigned long long x = seed + 0x9E3779B97F4A7C15ull * (a + 1) + 0xBF58476D1CE4E5B9ull * (b + 1);
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}

// Radix-select bucket pick by one wave: hist[256] counts, find the highest byte value b with
// count(bins > b) < kk <= count(bins >= b); returns b and writes the count of strictly higher bins to *above.
__device__ __forceinline__ int pick_bucket_wave(const unsigned* hist, unsigned kk, int lane, unsigned* above) {
    const unsigned h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
    unsigned suf = h0 + h1 + h2 + h3;                    // becomes the inclusive suffix sum over lanes >= lane
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = __shfl_down(suf, o, 64);
        if (lane + o < 64) suf += v;
    }
    const unsigned long long m = __ballot(suf >= kk);    // lanes whose suffix reaches kk; the highest one holds the bucket
    const int sel = 63 - __builtin_clzll(m | 1ull);
    unsigned higher = __shfl_down(suf, 1, 64);
    if (lane == 63) higher = 0;                          // count in lanes strictly above
    int b = 0;
    unsigned ab = 0;
    if (lane == sel) {
        unsigned cum = higher;
        if (cum + h3 >= kk) { b = 4 * lane + 3; ab = cum; }
        else { cum += h3;
            if (cum + h2 >= kk) { b = 4 * lane + 2; ab = cum; }
            else { cum += h2;
                if (cum + h1 >= kk) { b = 4 * lane + 1; ab = cum; }
                else { cum += h1; b = 4 * lane; ab = cum; } } }
    }
    b = __shfl(b, sel, 64);
    *above = __shfl(ab, sel, 64);
    return b;
}

// k-th largest order-preserving key of a score row in LDS (k <= 64, V <= 256 * TOPK_KPT), by a block of 256 threads, without atomics or a
// serial bucket scan.  Measured against the 4-pass radix select (profiles/r03x/sample_topk.log, identical ids): par in sample_kernel (0.934 vs
// 0.934 ms per token at 1 row, 1.51 vs 1.50 at 64 rows: the selection is not what that kernel's 48 us are made of), -2 % per token in the beam
// kernel (1.113 -> 1.088 at 1 x 3 rows, 2.40 -> 2.355 at 64 x 3), whose radix passes each ended in one thread scanning 256 buckets -> default there.
//   level 1: every wave bisects the 32 key bits over the keys its own lanes hold in registers -- count(key >= candidate) is one ballot + popcount
//            per register key, no cross-wave traffic -- to the k-th largest of ITS keys, T_w.  The global top-k lies inside the union of the waves'
//            local top-k sets: the wave emits its keys > T_w (fewer than k) and pads with copies of T_w to exactly k entries.
//   level 2: wave 0 bisects the <= 4 k emitted keys the same way -> the global k-th largest key.  Two barriers in all.
// A lane past the row end holds key 0 (below every real key; candidates are >= 1).
#define TOPK_KPT 33
__device__ __forceinline__ uint32_t topk_kth_key(const float* sl, int V, int k, int tid, uint32_t* cand /* LDS [4][64] */, uint32_t* result /* LDS [1] */) {
    const int lane = tid & 63, w = tid >> 6;
    uint32_t key[TOPK_KPT];
#pragma unroll
    for (int j = 0; j < TOPK_KPT; ++j) {
        const int i = tid + 256 * j;
        key[j] = i < V ? f2key(sl[i]) : 0u;
    }
    uint32_t T = 0;
    // (A lower bound from the per-lane maxima + compaction of the keys above it before the bisection -- about 170 ballot rounds instead of 1056 --
    // gave identical ids and no measurable gain at 1 / 8 / 64 rows, profiles/r04p: removed.)
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t c = T | (1u << bit);
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < TOPK_KPT; ++j) cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(key[j] >= c));
        if (cnt >= k) T = c;                                          // wave-uniform
    }
    {   // emit: keys > T (fewer than k of them), then copies of T up to k entries
        int base = 0;
        const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
        for (int j = 0; j < TOPK_KPT; ++j) {
            const bool p = key[j] > T;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(p);
            if (p) { const int o = base + __builtin_popcountll(m & lt); if (o < 64) cand[w * 64 + o] = key[j]; }
            base += __builtin_popcountll(m);
        }
        if (lane >= base && lane < 64) cand[w * 64 + lane] = lane < k ? T : 0u;      // pad to k with T, the rest of the 64 slots with 0
    }
    __syncthreads();
    if (w == 0) {
        const uint32_t c0 = cand[lane], c1 = cand[64 + lane], c2 = cand[128 + lane], c3 = cand[192 + lane];
        uint32_t G = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t c = G | (1u << bit);
            const int cnt = __builtin_popcountll(__builtin_amdgcn_ballot_w64(c0 >= c)) + __builtin_popcountll(__builtin_amdgcn_ballot_w64(c1 >= c)) +
                            __builtin_popcountll(__builtin_amdgcn_ballot_w64(c2 >= c)) + __builtin_popcountll(__builtin_amdgcn_ballot_w64(c3 >= c));
            if (cnt >= k) G = c;
        }
        if (lane == 0) *result = G;
    }
    __syncthreads();
    return *result;
}

// ---- TypicalLogitsWarper (indextts/utils/typical_sampling.py:9-30; appended after the repetition penalty by
// UnifiedVoice.inference_speech, model_v2.py:794-799) on one LDS score row `sl[V]`; `q[V]` is scratch.
//   s_i = | -log_softmax(x)_i - H |,  tokens ordered by ascending s, kept while the softmax mass of the tokens before
//   them (f64 running sum rounded to f32, as torch.cumsum does on CPU) is < mass; everything with s above the crossing
//   token's s becomes -inf; the `min_keep` smallest-s tokens always stay.
// No sort: the crossing key is found by a bitwise search over the 32-bit key space (keys = bits of s >= 0), each probe a
// fixed-order f64 block reduction of the mass below the candidate key -- deterministic, 32 probes.
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ void typical_filter(float* sl, float* q, int V, float mass, int min_keep, int tid) {
    __shared__ float tf_red[4];
    __shared__ double td_red[4];
    __shared__ unsigned tk_key[4];
    __shared__ int tk_idx[4];
    const int w = tid >> 6;
    const bool lead = (tid & 63) == 0;
    float mx = -INFINITY;
    for (int i = tid; i < V; i += 256) mx = fmaxf(mx, sl[i]);
    mx = wave_max(mx);
    if (lead) tf_red[w] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(tf_red[0], tf_red[1]), fmaxf(tf_red[2], tf_red[3]));
    __syncthreads();
    float se = 0.f;
    for (int i = tid; i < V; i += 256) se += expf(sl[i] - mx);
    se = wave_sum(se);
    if (lead) tf_red[w] = se;
    __syncthreads();
    const float sum = (tf_red[0] + tf_red[1]) + (tf_red[2] + tf_red[3]);
    const float lse = logf(sum);
    __syncthreads();
    float en = 0.f;
    for (int i = tid; i < V; i += 256) {
        const float nl = (sl[i] - mx) - lse;
        const float t = nl * expf(nl);
        if (t == t) en += t;                              // nansum
        q[i] = expf(sl[i] - mx) / sum;
    }
    en = wave_sum(en);
    if (lead) tf_red[w] = en;
    __syncthreads();
    const float ent = -((tf_red[0] + tf_red[1]) + (tf_red[2] + tf_red[3]));
    auto key_of = [&](int i) -> unsigned { return __float_as_uint(fabsf((-((sl[i] - mx) - lse)) - ent)); };
    unsigned T = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned cand = T | (1u << bit);
        double m = 0.0;
        for (int i = tid; i < V; i += 256)
            if (key_of(i) < cand) m += (double)q[i];
        m = wave_sum_f64(m);
        __syncthreads();                                  // previous probe's readers are done with td_red
        if (lead) td_red[w] = m;
        __syncthreads();
        const double tot = (td_red[0] + td_red[1]) + (td_red[2] + td_red[3]);
        if ((float)tot < mass) T = cand;
    }
    // the min_keep (1 or 2) smallest (key, index) stay regardless; the smallest always passes key <= T
    int keep2 = -1;
    if (min_keep > 1) {
        int first = -1;
        for (int round = 0; round < 2; ++round) {
            unsigned bk = 0xFFFFFFFFu; int bi = 0x7fffffff;
            for (int i = tid; i < V; i += 256) {
                if (i == first) continue;
                const unsigned k = key_of(i);
                if (k < bk || (k == bk && i < bi)) { bk = k; bi = i; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned ok = __shfl_xor(bk, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (ok < bk || (ok == bk && oi < bi)) { bk = ok; bi = oi; }
            }
            __syncthreads();
            if (lead) { tk_key[w] = bk; tk_idx[w] = bi; }
            __syncthreads();
            for (int ww = 0; ww < 4; ++ww)
                if (tk_key[ww] < bk || (tk_key[ww] == bk && tk_idx[ww] < bi)) { bk = tk_key[ww]; bi = tk_idx[ww]; }
            if (round == 0) first = bi; else keep2 = bi;
        }
    }
    for (int i = tid; i < V; i += 256)
        if (key_of(i) > T && i != keep2) sl[i] = -INFINITY;
    __syncthreads();
}

// The step / cache-position counters advance once per token.  Every block of the step's last kernel reads them at entry, so
// the LAST block to finish (arrival ticket) can bump them: a separate 1-thread kernel for this costs a full launch slot
// (4.3 us in the round-1 trace).  state = {step, pos, ticket}.
__device__ __forceinline__ void advance_when_last(int* state, int nblocks) {
    if (!state) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const int t = atomicAdd(&state[2], 1);
        if (t == nblocks - 1) {
            state[2] = 0;
            state[0] += 1;
            state[1] += 1;
            __threadfence();
        }
    }
}

// tools/microbench/sample_stamps.hip builds this kernel with -DITTS_SAMPLE_STAMPS: thread 0 of every block stores the constant-rate clock
// (s_memrealtime, 100 MHz) at eight phase boundaries into a.stamps[block][8]; nothing is compiled in the product.
#ifdef ITTS_SAMPLE_STAMPS
#define SAMPLE_STAMP(K_) do { if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 8 + (K_)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SAMPLE_STAMP(K_) do { } while (0)
#endif
__global__ __launch_bounds__(256) void sample_kernel(SampleArgs a) {
    extern __shared__ float sl[];                    // [V] processed scores
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_mask, s_kk, s_count;
    __shared__ float red_v[4];
    __shared__ int red_i[4];
    __shared__ int s_tok;
    __shared__ int cand_i[SAMPLE_CAP];
    __shared__ float cand_v[SAMPLE_CAP], cand_e[SAMPLE_CAP];
    const int b = blockIdx.x, tid = threadIdx.x, V = a.V;
    const int step = *a.step_ptr;
    SAMPLE_STAMP(0);
    const int u = a.row_slot ? a.row_slot[b] : b;                   // the utterance this dense row carries
    const int rstep = step - (a.row_step0 ? a.row_step0[u] : 0);    // the row's own step (rows admitted into a running batch start later)
    const float* lg = a.logits + (size_t)b * V;
    unsigned char* seen = a.seen + (size_t)u * V;
    const bool pen = a.rep_penalty != 1.0f;
    const bool temp = a.do_sample && a.temperature != 1.0f;
    const bool typical = a.typical_mass > 0.f;
    // (Issuing every load of the row before the first use -- the stamps put this phase at 6.5 us of 33 dependent round trips -- changed nothing
    // measurable at token level, profiles/r03x: removed.)
    for (int i = tid; i < V; i += 256) {
        float x = lg[i];
        if (pen && seen[i]) x = x < 0.f ? x * a.rep_penalty : x / a.rep_penalty;
        if (temp && !typical) x = x / a.temperature;
        sl[i] = x;
    }
    __syncthreads();
    SAMPLE_STAMP(1);
    if (typical) {
        typical_filter(sl, sl + V, V, a.typical_mass, a.min_keep, tid);
        if (temp) {
            for (int i = tid; i < V; i += 256) sl[i] = sl[i] / a.temperature;
            __syncthreads();
        }
    }

    if (!a.do_sample) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int i = tid; i < V; i += 256) {
            const float x = sl[i];
            if (x > bv) { bv = x; bi = i; }        // ascending i per thread: first max kept
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if ((tid & 63) == 0) { red_v[tid >> 6] = bv; red_i[tid >> 6] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int ww = 1; ww < 4; ++ww)
                if (red_v[ww] > bv || (red_v[ww] == bv && red_i[ww] < bi)) { bv = red_v[ww]; bi = red_i[ww]; }
            s_tok = bi == 0x7fffffff ? 0 : bi;
        }
    } else {
        // ---- top-k threshold by 4-pass radix select on order-preserving keys ----
        const int k = max(a.top_k, a.min_keep) < V ? max(a.top_k, a.min_keep) : V;
        if (tid == 0) { s_prefix = 0; s_mask = 0; s_kk = (unsigned)k; s_count = 0; }
        uint32_t kth;
        if (V <= 256 * TOPK_KPT && k <= 64 && !a.radix_select) {     // ballot bisection (topk_kth_key); the radix select stays as the A/B path
            __syncthreads();                                           // s_count = 0 visible before the survivors are counted below
            kth = topk_kth_key(sl, V, k, tid, hist, &s_prefix);
        } else {
        for (int pass = 3; pass >= 0; --pass) {
            hist[tid] = 0;
            __syncthreads();
            const unsigned prefix = s_prefix, mask = s_mask;
            const int shift = pass * 8;
            for (int i = tid; i < V; i += 256) {
                const uint32_t key = f2key(sl[i]);
                if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (tid < 64) {
                unsigned above;
                const int bsel = pick_bucket_wave(hist, s_kk, tid, &above);
                if (tid == 0) {
                    s_prefix = prefix | ((unsigned)bsel << shift);
                    s_mask = mask | (255u << shift);
                    s_kk = s_kk - above;
                }
            }
            __syncthreads();
        }
        kth = s_prefix;
        }
        SAMPLE_STAMP(2);
        for (int i = tid; i < V; i += 256) {
            if (f2key(sl[i]) >= kth && sl[i] > -INFINITY) {   // masked (-inf) entries carry no probability
                const unsigned slot = atomicAdd(&s_count, 1u);
                if (slot < SAMPLE_CAP) { cand_i[slot] = i; cand_v[slot] = sl[i]; }
            }
        }
        __syncthreads();
        {   // parallel rank sort of the survivors, ascending by (value, index), and their exponentials
            const int n = (int)(s_count < SAMPLE_CAP ? s_count : SAMPLE_CAP);
            float myv = 0.f; int myi = 0, rank = 0;
            if (tid < n) {
                myv = cand_v[tid]; myi = cand_i[tid];
                for (int j = 0; j < n; ++j) {
                    const float vj = cand_v[j]; const int ij = cand_i[j];
                    rank += (vj < myv || (vj == myv && ij < myi)) ? 1 : 0;
                }
            }
            __syncthreads();
            if (tid < n) { cand_v[rank] = myv; cand_i[rank] = myi; }
            __syncthreads();
        }
        SAMPLE_STAMP(3);
        {   // exponentials in parallel; the float / double sums below keep the sequential ascending order
            const int n = (int)(s_count < SAMPLE_CAP ? s_count : SAMPLE_CAP);
            if (tid < n) cand_e[tid] = expf(cand_v[tid] - cand_v[n - 1]);
            __syncthreads();
            if (tid == 0) {
                int lo = 0;                                   // first kept element after top-p
                if (a.top_p < 1.0f) {
                    float sum = 0.f;
                    for (int i = 0; i < n; ++i) sum += cand_e[i];
                    const float thr = (float)(1.0 - (double)a.top_p);
                    double cum = 0.0;
                    const int keep = a.min_keep < 1 ? 1 : a.min_keep;
                    for (int i = 0; i < n - keep; ++i) {
                        cum += (double)(cand_e[i] / sum);
                        if ((float)cum <= thr) lo = i + 1; else break;
                    }
                }
                float sum = 0.f;                              // renormalised softmax over the kept set
                for (int i = lo; i < n; ++i) sum += cand_e[i];
                red_v[0] = sum;
                red_i[0] = lo;
            }
            __syncthreads();
            const int lo = red_i[0];
            const float sum = red_v[0];
            // kept set in vocabulary order: parallel rank sort on the (unique) token index
            float myp = 0.f; int myi = 0, rank = 0;
            if (tid >= lo && tid < n) {
                myp = cand_e[tid] / sum; myi = cand_i[tid];
                for (int q = lo; q < n; ++q) rank += (cand_i[q] < myi) ? 1 : 0;
            }
            __syncthreads();
            if (tid >= lo && tid < n) { cand_v[lo + rank] = myp; cand_i[lo + rank] = myi; }
            __syncthreads();
            if (tid == 0) {
                double total = 0.0;
                for (int i = lo; i < n; ++i) total += (double)cand_v[i];
                const double ur = a.uniforms ? a.uniforms[(size_t)(rstep < a.max_new ? rstep : a.max_new - 1) * (a.uniforms_stride > 0 ? a.uniforms_stride : a.B) + u]
                                             : rng_uniform(a.seed_ptr ? *a.seed_ptr : a.seed, (unsigned long long)rstep, (unsigned long long)u);
                const double tgt = ur * total;
                double cum = 0.0;
                int pick = cand_i[n - 1];
                for (int i = lo; i < n; ++i) {
                    cum += (double)cand_v[i];
                    if (cum > tgt) { pick = cand_i[i]; break; }
                }
                s_tok = pick;
            }
        }
    }
    __syncthreads();
    SAMPLE_STAMP(4);
    if (tid == 0) {
        int tok = s_tok;
        if (a.finished[u]) tok = a.stop_token;               // :3256 finished rows emit pad (= stop)
        if (a.row_limit && rstep >= a.row_limit[u]) tok = a.stop_token;         // this utterance's own max_mel_tokens
        if (rstep >= a.max_new) tok = a.stop_token;                             // (a session's step counter runs past max_new: the row's own step bounds it)
        else a.tokens[(size_t)u * a.max_new + rstep] = tok;
        if (tok == a.stop_token) a.finished[u] = 1;
        seen[tok] = 1;
        s_tok = tok;
    }
    __syncthreads();
    SAMPLE_STAMP(5);
    if (a.x_next) {
        const int tok = s_tok;
        int p = rstep + a.pos_offset;
        p = p < a.n_mel_pos ? p : a.n_mel_pos - 1;
        for (int d = tid; d < a.D; d += 256)
            a.x_next[(size_t)b * a.D + d] = a.mel_emb[(size_t)tok * a.D + d] + a.mel_pos[(size_t)p * a.D + d];
    }
    SAMPLE_STAMP(6);
    advance_when_last(a.adv_state, (int)gridDim.x);
    SAMPLE_STAMP(7);
}

// The selection kernels keep the whole score row in LDS ([V] f32, twice that with typical sampling: 65.5 KB at the production
// vocabulary of 8194, above the 64 KiB a kernel gets without asking).  Raise the kernel's dynamic-LDS limit once to what the
// launch needs; the CU has 160 KiB.
template <class K>
static int ensure_dyn_lds(K kernel, size_t bytes, size_t* granted, const char* who) {
    constexpr size_t kStatic = 8192;                    // static __shared__ of these kernels, rounded up
    if (bytes + kStatic > 160 * 1024) {
        itts_set_error("%s: vocabulary needs %zu bytes of LDS, above the 160 KiB of a CU", who, bytes);
        return ITTS_ERR_ARG;
    }
    if (bytes > 48 * 1024 && bytes > *granted) {
        HIP_TRY(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        *granted = bytes;
    }
    return ITTS_OK;
}

int launch_sample(const SampleArgs& a, hipStream_t st) {
    if (a.B <= 0) return ITTS_OK;
    if (a.do_sample && (a.top_k <= 0 || a.top_k > 64)) {
        itts_set_error("sampling: top_k must be in 1..64 on the device path (got %d)", a.top_k);
        return ITTS_ERR_ARG;
    }
    if (a.typical_mass != 0.f && !(a.typical_mass > 0.f && a.typical_mass < 1.f)) {
        itts_set_error("`typical_mass` has to be a float > 0 and < 1, but is %g", (double)a.typical_mass);
        return ITTS_ERR_ARG;
    }
    const size_t lds = (size_t)a.V * sizeof(float) * (a.typical_mass > 0.f ? 2 : 1);
    static ItPerDevice<size_t> granted_pd;
    size_t& granted = granted_pd.cur();
    if (int rc = ensure_dyn_lds(sample_kernel, lds, &granted, "sampling")) return rc;
    // top-k threshold: the radix select here (par with the ballot bisection at 1 row, 1 % ahead at 64 rows), the bisection in the beam
    // kernel (-2 %: its radix pass ended in a serial 256-bucket scan); ITTS_SAMPLE_RADIX=0 / 1 forces one of them in both (profiles/r03x)
    const int radix = itts_opt(OPT_SAMPLE_RADIX) < 0 ? 1 : itts_opt(OPT_SAMPLE_RADIX);
    SampleArgs a2 = a;
    a2.radix_select = radix;
    hipLaunchKernelGGL(sample_kernel, dim3(a.B), dim3(256), lds, st, a2);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

__global__ void advance_kernel(int* step_ptr, int* pos_ptr) {
    *step_ptr += 1;
    *pos_ptr += 1;
}

int launch_advance(int* step_ptr, int* pos_ptr, hipStream_t st) {
    hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(1), 0, st, step_ptr, pos_ptr);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

// ================================================================================================================
// Beam search / beam-sample: one block per utterance.
//   per beam: log_softmax -> repetition penalty (on log-probs) -> [temperature -> top-k -> top-p]  (processors act on
//   log-probs here, generation_utils.py:3475-3479) -> + beam score; union over beams; 2*nb candidates by multinomial
//   without replacement (inverse CDF over the flattened beam-major index, one uniform per draw) or top-2*nb; sorted by
//   score; BeamSearchScorer.process (transformers_beam_search.py:215-305) incl. hypothesis heap and the
//   early_stopping=False done rule (:979-996).
// ================================================================================================================
#define BEAM_CAP 64            // survivors kept per beam after top-k (top_k <= 64)

// Phase 1, one block per sequence row (utterance x beam): processed log-probs of the row -> its survivors (token, score +
// beam score) in vocabulary-independent order (ascending by (value, index)), written to surv_*[row][BEAM_CAP].
__global__ __launch_bounds__(256) void beam_rows_kernel(BeamArgs a) {
    extern __shared__ float sl[];                        // [V] processed log-probs (+ [V] scratch with typical sampling)
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_mask, s_kk, s_count;
    __shared__ float red[4];
    __shared__ float s_m, s_lse;
    __shared__ int ci[BEAM_CAP];
    __shared__ float cv[BEAM_CAP];
    __shared__ float ue[BEAM_CAP];
    __shared__ int s_lo;
    const int row = blockIdx.x, tid = threadIdx.x, V = a.V, nb = a.nb;
    const int b = row / nb, j = row - b * nb;
    const int step = *a.step_ptr;
    const int par = step & 1;
    if (a.done[b]) { if (tid == 0) a.surv_n[row] = 0; return; }
    const bool pen = a.rep_penalty != 1.0f;
    const bool temp = a.do_sample && a.temperature != 1.0f;
    const int ksel_raw = a.do_sample ? max(a.top_k, a.min_keep) : 2 * nb;
    const int ksel = ksel_raw < V ? ksel_raw : V;
    {
        const float* lg = a.logits + (size_t)(a.logits_shared ? b : row) * V;
        const unsigned char* seen = a.seen[par] + (size_t)row * V;
        // log_softmax
        float mx = -INFINITY;
        for (int i = tid; i < V; i += 256) { const float x = lg[i]; sl[i] = x; mx = fmaxf(mx, x); }
        mx = wave_max(mx);
        if ((tid & 63) == 0) red[tid >> 6] = mx;
        __syncthreads();
        if (tid == 0) s_m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        __syncthreads();
        const float m = s_m;
        float se = 0.f;
        for (int i = tid; i < V; i += 256) se += expf(sl[i] - m);
        se = wave_sum(se);
        if ((tid & 63) == 0) red[tid >> 6] = se;
        __syncthreads();
        if (tid == 0) { s_lse = logf((red[0] + red[1]) + (red[2] + red[3])); s_prefix = 0; s_mask = 0; s_kk = (unsigned)ksel; s_count = 0; }
        __syncthreads();
        const float lse = s_lse;
        const bool typical = a.typical_mass > 0.f;
        for (int i = tid; i < V; i += 256) {
            float x = (sl[i] - m) - lse;
            if (pen && seen[i]) x = x < 0.f ? x * a.rep_penalty : x / a.rep_penalty;
            if (temp && !typical) x = x / a.temperature;
            sl[i] = x;
        }
        __syncthreads();
        if (typical) {
            typical_filter(sl, sl + V, V, a.typical_mass, a.min_keep, tid);
            if (temp) {
                for (int i = tid; i < V; i += 256) sl[i] = sl[i] / a.temperature;
                __syncthreads();
            }
        }
        // top-ksel threshold: ballot bisection (topk_kth_key), or the radix select (A/B path, and for rows it does not cover)
        uint32_t kth;
        if (V <= 256 * TOPK_KPT && ksel <= 64 && !a.radix_select) {
            kth = topk_kth_key(sl, V, ksel, tid, hist, &s_prefix);
        } else {
        for (int pass = 3; pass >= 0; --pass) {
            hist[tid] = 0;
            __syncthreads();
            const unsigned prefix = s_prefix, mask = s_mask;
            const int shift = pass * 8;
            for (int i = tid; i < V; i += 256) {
                const uint32_t key = f2key(sl[i]);
                if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (tid == 0) {
                unsigned cum = 0, kk = s_kk;
                int bsel = 0;
                for (int bb = 255; bb >= 0; --bb) {
                    if (cum + hist[bb] >= kk) { bsel = bb; break; }
                    cum += hist[bb];
                }
                s_prefix = prefix | ((unsigned)bsel << shift);
                s_mask = mask | (255u << shift);
                s_kk = kk - cum;
            }
            __syncthreads();
        }
        kth = s_prefix;
        }
        for (int i = tid; i < V; i += 256) {
            if (f2key(sl[i]) >= kth && sl[i] > -INFINITY) {
                const unsigned slot = atomicAdd(&s_count, 1u);
                if (slot < BEAM_CAP) { ci[slot] = i; cv[slot] = sl[i]; }
            }
        }
        __syncthreads();
        {   // ascending by (value, index): parallel rank sort (same order as a stable insertion sort on that key)
            const int n = (int)(s_count < BEAM_CAP ? s_count : BEAM_CAP);
            float myv = 0.f; int myi = 0, rank = 0;
            if (tid < n) {
                myv = cv[tid]; myi = ci[tid];
                for (int q = 0; q < n; ++q) {
                    const float vq = cv[q]; const int iq = ci[q];
                    rank += (vq < myv || (vq == myv && iq < myi)) ? 1 : 0;
                }
            }
            __syncthreads();
            if (tid < n) { cv[rank] = myv; ci[rank] = myi; }
            __syncthreads();
            if (tid < n) ue[tid] = expf(cv[tid] - cv[n - 1]);
            __syncthreads();
            if (tid == 0) {
                int lo = 0;
                if (a.do_sample && a.top_p < 1.0f) {
                    float sum = 0.f;
                    for (int i = 0; i < n; ++i) sum += ue[i];
                    const float thr = (float)(1.0 - (double)a.top_p);
                    double cum = 0.0;
                    const int keep = a.min_keep < 1 ? 1 : a.min_keep;
                    for (int i = 0; i < n - keep; ++i) {
                        cum += (double)(ue[i] / sum);
                        if ((float)cum <= thr) lo = i + 1; else break;
                    }
                }
                s_lo = lo;
            }
            __syncthreads();
            const int lo = s_lo;
            const float bs = a.beam_scores[row];
            if (tid >= lo && tid < n) {
                a.surv_idx[(size_t)row * BEAM_CAP + tid - lo] = j * V + ci[tid];
                a.surv_val[(size_t)row * BEAM_CAP + tid - lo] = cv[tid] + bs;
            }
            if (tid == 0) a.surv_n[row] = n - lo;
        }
    }
}

// Phase 2, one block per utterance: union of its beams' survivors, candidate draw / top, scorer.
__global__ __launch_bounds__(256) void beam_step_kernel(BeamArgs a) {
    __shared__ int ui[BEAM_MAX * BEAM_CAP];              // union: flat index beam*V + token
    __shared__ float uv[BEAM_MAX * BEAM_CAP];
    __shared__ float ue[BEAM_MAX * BEAM_CAP];            // exp(uv - max) of the union
    __shared__ float s_mxv;
    const int b = blockIdx.x, tid = threadIdx.x, V = a.V, nb = a.nb;
    const int step = *a.step_ptr;
    if (a.done[b]) {                                     // :255-264 finished utterance: pad tokens, score 0
        if (tid < nb) {
            a.next_scores[b * nb + tid] = 0.f;
            a.next_tokens[b * nb + tid] = a.stop_token;
            a.next_indices[b * nb + tid] = b * nb + tid;
        }
        return;
    }
    int un = 0;
    for (int j = 0; j < nb; ++j) {                       // beams in order, each beam's survivors in its own order
        const int row = b * nb + j, n = a.surv_n[row];
        if (tid < n) { ui[un + tid] = a.surv_idx[(size_t)row * BEAM_CAP + tid]; uv[un + tid] = a.surv_val[(size_t)row * BEAM_CAP + tid]; }
        un += n;
    }
    __syncthreads();
    // ---- the union in vocabulary order of the flattened (beam-major) row: parallel rank sort on the unique flat index ----
    {
        float myv = 0.f; int myi = 0, rank = 0;
        if (tid < un) {
            myv = uv[tid]; myi = ui[tid];
            for (int q = 0; q < un; ++q) rank += (ui[q] < myi) ? 1 : 0;
        }
        __syncthreads();
        if (tid < un) { uv[rank] = myv; ui[rank] = myi; }
        __syncthreads();
        if (a.do_sample) {
            if (tid == 0) {
                float mxv = -INFINITY;
                for (int i = 0; i < un; ++i) mxv = fmaxf(mxv, uv[i]);
                s_mxv = mxv;
            }
            __syncthreads();
            if (tid < un) ue[tid] = expf(uv[tid] - s_mxv);
            __syncthreads();
        }
    }
    if (tid != 0) return;
    // ---- candidate selection over the union (thread 0; <= nb*64 entries) ----
    const int ncand = 2 * nb;
    int ctok[2 * BEAM_MAX], cidx[2 * BEAM_MAX];
    float csc[2 * BEAM_MAX];
    int nc = 0;
    if (a.do_sample) {
        float sum = 0.f;
        for (int i = 0; i < un; ++i) sum += ue[i];
        for (int i = 0; i < un; ++i) ue[i] = ue[i] / sum;          // probabilities; a drawn entry is zeroed
        for (int d = 0; d < ncand && d < un; ++d) {
            double total = 0.0;
            for (int i = 0; i < un; ++i) total += (double)ue[i];
            const double u = a.uniforms ? a.uniforms[((size_t)step * a.B + b) * ncand + d]
                                        : rng_uniform(a.seed_ptr ? *a.seed_ptr : a.seed, (unsigned long long)step * 8 + d, (unsigned long long)b);
            const double tgt = u * total;
            double cum = 0.0;
            int pick = -1, lastfree = -1;
            for (int i = 0; i < un; ++i) {
                if (ui[i] < 0) continue;                              // already drawn
                lastfree = i;
                cum += (double)ue[i];
                if (cum > tgt) { pick = i; break; }
            }
            if (pick < 0) pick = lastfree;
            ctok[nc] = ui[pick] % V; cidx[nc] = ui[pick] / V; csc[nc] = uv[pick]; ++nc;
            ui[pick] = -1 - ui[pick];
            ue[pick] = 0.f;
        }
    } else {
        for (int d = 0; d < ncand && d < un; ++d) {       // top-2nb by score, ties -> lower flat index
            int best = -1;
            for (int i = 0; i < un; ++i)
                if (ui[i] >= 0 && (best < 0 || uv[i] > uv[best])) best = i;
            ctok[nc] = ui[best] % V; cidx[nc] = ui[best] / V; csc[nc] = uv[best]; ++nc;
            ui[best] = -1 - ui[best];                     // taken
        }
    }
    for (int i = 1; i < nc; ++i) {                        // sort candidates by score, descending (stable)
        const float v = csc[i];
        const int t = ctok[i], x = cidx[i];
        int q = i - 1;
        while (q >= 0 && csc[q] < v) { csc[q + 1] = csc[q]; ctok[q + 1] = ctok[q]; cidx[q + 1] = cidx[q]; --q; }
        csc[q + 1] = v; ctok[q + 1] = t; cidx[q + 1] = x;
    }
    // ---- BeamSearchScorer.process ----
    const int gen_len = step + 1;                          // cur_len - decoder_prompt_len
    BeamHyp* hy = a.hyps + (size_t)b * BEAM_MAX;
    int nh = a.n_hyps[b];
    float worst = a.worst[b];
    int filled = 0;
    for (int rank = 0; rank < nc && filled < nb; ++rank) {
        if (ctok[rank] == a.stop_token) {
            if (rank >= nb) continue;
            const float sc = csc[rank] / powf((float)gen_len, a.length_penalty);
            if (nh < nb || sc > worst) {                  // BeamHypotheses.add (:955-976)
                BeamHyp nhyp{sc, step, b * nb + cidx[rank], 0};
                if (nh < nb) {
                    hy[nh++] = nhyp;
                    worst = fminf(sc, worst);
                } else {
                    int wi = 0;                           // replace the worst, new worst = second worst of the old+new set
                    for (int q = 1; q < nh; ++q) if (hy[q].score < hy[wi].score) wi = q;
                    hy[wi] = nhyp;
                    float w2 = hy[0].score;
                    for (int q = 1; q < nh; ++q) w2 = fminf(w2, hy[q].score);
                    worst = w2;
                }
            }
        } else {
            a.next_scores[b * nb + filled] = csc[rank];
            a.next_tokens[b * nb + filled] = ctok[rank];
            a.next_indices[b * nb + filled] = b * nb + cidx[rank];
            ++filled;
        }
    }
    for (; filled < nb; ++filled) {                       // cannot happen with >= nb non-EOS candidates; keep state sane
        a.next_scores[b * nb + filled] = -1e9f;
        a.next_tokens[b * nb + filled] = a.stop_token;
        a.next_indices[b * nb + filled] = b * nb;
    }
    a.n_hyps[b] = nh;
    a.worst[b] = worst;
    if (nh >= nb) {                                       // is_done, early_stopping=False (:979-996)
        const float highest = csc[0] / powf((float)gen_len, a.length_penalty);
        if (worst >= highest) a.done[b] = 1;
    }
}

// one block per sequence row: inherit the chosen parent's history (seen set, KV row map), record the step
__global__ __launch_bounds__(256) void beam_apply_kernel(BeamArgs a) {
    const int i = blockIdx.x, tid = threadIdx.x, V = a.V;
    const int step = *a.step_ptr;
    const int par = step & 1;
    const int src = a.next_indices[i], tok = a.next_tokens[i];
    const unsigned char* so = a.seen[par] + (size_t)src * V;
    unsigned char* sn = a.seen[1 - par] + (size_t)i * V;
    for (int k = tid; k < V; k += 256) sn[k] = so[k];
    const int npos = a.S + step;                           // cache index of this token's K/V in the next forward
    const int* mo = a.row_map[par] + (size_t)src * a.Tmax;
    int* mn = a.row_map[1 - par] + (size_t)i * a.Tmax;
    for (int t = tid; t < npos && t < a.Tmax; t += 256) mn[t] = mo[t];
    __syncthreads();
    if (tid == 0) {
        if (tok >= 0 && tok < V) sn[tok] = 1;
        if (npos < a.Tmax) mn[npos] = i;
        a.hist_tok[(size_t)step * a.B * a.nb + i] = tok;
        a.hist_par[(size_t)step * a.B * a.nb + i] = src;
        a.beam_scores[i] = a.next_scores[i];
    }
    int p = step + a.pos_offset;
    p = p < a.n_mel_pos ? p : a.n_mel_pos - 1;
    const int tk = (tok >= 0 && tok < V) ? tok : 0;
    for (int d = tid; d < a.D; d += 256)
        a.x_next[(size_t)i * a.D + d] = a.mel_emb[(size_t)tk * a.D + d] + a.mel_pos[(size_t)p * a.D + d];
    advance_when_last(a.adv_state, (int)gridDim.x);
}

int launch_beam_step(const BeamArgs& a, hipStream_t st) {
    if (a.nb < 2 || a.nb > BEAM_MAX) { itts_set_error("beam: num_beams must be 2..%d", BEAM_MAX); return ITTS_ERR_ARG; }
    if (a.do_sample && (a.top_k <= 0 || max(a.top_k, a.min_keep) > BEAM_CAP)) {
        itts_set_error("beam-sample: top_k must be in 1..%d on the device path (got %d)", BEAM_CAP, a.top_k);
        return ITTS_ERR_ARG;
    }
    if (a.typical_mass != 0.f && !(a.typical_mass > 0.f && a.typical_mass < 1.f)) {
        itts_set_error("`typical_mass` has to be a float > 0 and < 1, but is %g", (double)a.typical_mass);
        return ITTS_ERR_ARG;
    }
    const size_t lds = (size_t)a.V * sizeof(float) * (a.typical_mass > 0.f ? 2 : 1);
    static ItPerDevice<size_t> granted_pd;
    size_t& granted = granted_pd.cur();
    if (int rc = ensure_dyn_lds(beam_rows_kernel, lds, &granted, "beam search")) return rc;
    const int radix = itts_opt(OPT_SAMPLE_RADIX) < 0 ? 0 : itts_opt(OPT_SAMPLE_RADIX);
    BeamArgs a2 = a;
    a2.radix_select = radix;
    hipLaunchKernelGGL(beam_rows_kernel, dim3(a.B * a.nb), dim3(256), lds, st, a2);
    hipLaunchKernelGGL(beam_step_kernel, dim3(a.B), dim3(256), 0, st, a);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

int launch_beam_apply(const BeamArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(beam_apply_kernel, dim3(a.B * a.nb), dim3(256), 0, st, a);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}
