// The fp32x3 tile GEMM of the flow-matching stage, in its own translation unit so that it is built WITHOUT SLP vectorisation (build.py): in
// gpt_kernels.hip -- which needs the vectoriser for the GPT sampler's bit-exact fixtures -- the operand split's subtractions had to go through a
// one-instruction asm to stay scalar, and every such statement cost an s_nop in front of its consumer (the hazard recognizer pads inline asm
// blindly); here they are plain C and the K loop carries ~45 fewer issue slots per K tile.
#include "gemm_tile.h"

// ================================================================================================================
// f32-accurate GEMM on the bf16 matrix pipe ("f32x3"): every f32 operand is carried EXACTLY as three bf16 planes
//   x = xh + xm + xl,  xh = bf16(x), xm = bf16(x - xh), xl = bf16(x - xh - xm)        (8 + 8 + 8 significand bits, no bit dropped)
// and an f32 product is the sum of the plane products, each of them exact in the f32 accumulator of v_mfma_f32_16x16x32_bf16:
//   NPROD = 8: every term down to 2^-24 |a b| (hh, hm, mh, mm, hl, lh, ml, lm; only ll, 2^-32, is dropped) -- closer to the exact
//              product than one f32 rounding;  NPROD = 6: without ml / lm (<= 2^-24 |a b| each in the worst case; their sum measured at
//              5.7e-9 of the result's RMS on N(0, 1) operands at K = 512 by a CPU emulation of the plane arithmetic: 1/50 of the native f32
//              GEMM's own error against f64).
// Accumulation stays f32.  The native f32 MFMA runs at 1/16 of the bf16 rate, so 8 bf16 MFMAs per K = 32 tile pair replace 8 f32
// MFMAs of twice the issue time: 2x the native-f32 matrix rate (2.67x with 6 products) -- tests/test_gpu_gemm_x3.py holds the
// result to an f64 GEMM and compares its error with the native f32 kernel's on the same operands.
//   Activations stay f32 in HBM and LDS: the A tile is DMA'd exactly like the f32 tile kernel's ([128 rows][32 k] f32, same image and
//   swizzle) and each wave splits its fragments in registers (11 VALU ops per pair of values, software-pipelined under the MFMAs).  The
//   lane's eight k-values are the two 16-byte pieces the f32 kernel reads (k = 4 kg + j and 16 + 4 kg + j): conflict-free, and the
//   weights are packed with the same k permutation.  Weights are split once on the host (itts_pack_gemm_weight, precision 2):
//   [N/16][K/32][3 planes][64 lanes][16 B], read by the waves straight into registers (no LDS stage).  LDS: two 16 KiB A stages; the
//   66 KiB epilogue image is the allocation -> two blocks per CU.
//   Measured (profiles/r03e..r03i): 155-165 TFLOP/s f32-equivalent with 8 products (native f32 MFMA kernel: 128), ~190 with 6 -- not
//   the 2x the instruction rates promise.  Variants that changed nothing: weights through LDS (80 / 66 KiB), the split as a burst or
//   interleaved, v_mfma_f32_32x32x16_bf16 (slower: 132).  The SQ counters show the matrix pipe 42 % busy with the waves issue-stalled,
//   and rocm-smi shows why the ceiling is low: under this kernel the socket sits at its ~1.25 kW power limit and the engine clock falls
//   from 2.39 GHz (the native f32 solve holds it at 1.2 kW) to ~2.03 GHz -- eight bf16 MFMAs cost more energy than the one f32 MFMA
//   they replace, so the power cap, not the issue rate, prices this mode.
#define X3_LDS PF_LDS            // 2 x 16 KiB of A stages; the epilogue's transposed image (+ row metadata) is the larger
#define X3_LDS_ONE (96 * 1024)   // an LDS request that admits ONE block per CU (option x3_pin = 1: diagnostic)

// A pointer known to be wave-uniform, pinned to SGPRs: with the wave index read as a scalar (v_readfirstlane of threadIdx.x >> 6) the weight-fragment
// bases, the A tile's base and every LDS-DMA destination are SGPR values, the per-lane part of a global address is a 32-bit offset, and the K loop
// carries no v_readfirstlane / 64-bit VALU address arithmetic (measured on the standalone model of this kernel, tools/microbench/x3_gemm_lab.hip:
// +3 ... +9 % by shape, profiles/r05b/x3_gemm_lab2.log -- the loop is bound by the waves' instruction issue: ~3 non-MFMA instructions per MFMA).
typedef const __attribute__((address_space(1))) char* pf_gptr_t;
__device__ __forceinline__ pf_gptr_t pf_uni(const char* p) {
    const uint64_t u = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return (pf_gptr_t)(((uint64_t)hi << 32) | lo);
}

// f32 x 8 -> three bf16 planes (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m); 11 VALU ops per pair of values: this translation unit is built without
// SLP vectorisation, so the subtractions stay scalar v_sub_f32 -- packed v_pk_add_f32 beside MFMAs costs ~13 cycles more than the pair it replaces)
__device__ __forceinline__ void x3_split8(const f32x4 p0, const f32x4 p1, v4u& H, v4u& M, v4u& L) {
    const float x[8] = {p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = x[2 * i], b = x[2 * i + 1];
        const uint32_t h = pf_cvt2(a, b);
        const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);              // exact
        const uint32_t m = pf_cvt2(ra, rb);
        const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);            // exact
        H[i] = h; M[i] = m; L[i] = pf_cvt2(sa, sb);
    }
}

template <int EPI, bool CONV = false, int NPROD = 8, bool SCHED = true, bool APL = false>
// APL: the A operand arrives as three bf16 planes (GemmArgs::a_planes, written once by the
// producer in fragment order): the K tile's three plane images are LDS-DMA'd (3 x 8 KiB) and a fragment is one ds_read_b128 per plane -- no split at all.
__global__ __launch_bounds__(256, 2) void gemm_x3_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char pf_sm[];      // [2][A 16 KiB] operand stages; the epilogue image is the larger
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // the wave index as a scalar (pf_uni)
    const int wr = w >> 1, wc = w & 1;
    const int n_mt = (a.M + PF_BM - 1) / PF_BM, n_nt = (a.N + PF_BN - 1) / PF_BN;
    const int total = n_mt * n_nt, per = (total + 7) >> 3;
    const int t = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (t >= total) return;
    const int g0 = t / (PF_GM * n_nt), first_m = g0 * PF_GM;
    const int gm = (n_mt - first_m) < PF_GM ? (n_mt - first_m) : PF_GM;
    const int r = t - g0 * PF_GM * n_nt;
    const int bn = r / gm, bm = first_m + (r - bn * gm);
    const int m0 = bm * PF_BM, nt0 = bn * (PF_BN / 16);
    const int nk = a.K >> 5;                                           // K tiles of 32
    const int ntiles = (a.N + 15) >> 4;

    constexpr int STAGE = APL ? 24576 : 16384;                          // bytes of one A stage
    const char* asrc[APL ? 6 : 4];                                      // tap mode: per-lane sources
    uint32_t aoff[APL ? 6 : 4];                                         // otherwise: (block-uniform base) + 32-bit lane offset
    const char* abase = (const char*)a.A + (size_t)m0 * a.lda * (APL ? 2 : 4);
    int cv_t[4], cv_T[4];
    const char* cv_base[4];
    const char* cv_zero[4];
    const int cv_kpt = CONV ? a.conv_W / 32 : 1;
    const int cv_left = CONV ? (a.conv_taps - 1) * a.conv_dil - ((a.conv_taps - 1) * a.conv_dil) / 2 : 0;
    if constexpr (APL) {
        // plane p, chunk c = 16 tile rows of 64 bytes (32 k of bf16): lane l fetches row 16 c + (l >> 2), 16-byte piece (l & 3) ^ ((l >> 4) & 3) into
        // LDS slot l of the chunk (source-side XOR: a fragment read of one k-group over 16 rows then covers 16 different bank quads)
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int pl = i >> 1, c = w + 4 * (i & 1);
            int m = m0 + c * 16 + (lane >> 2);
            m = m < a.M ? m : a.M - 1;
            const int piece = (lane & 3) ^ ((lane >> 4) & 3);
            asrc[i] = (const char*)a.A + ((size_t)pl * a.a_planes + (size_t)m * a.lda) * 2 + piece * 16;
            aoff[i] = (uint32_t)(m - m0) * (uint32_t)a.lda * 2u + piece * 16;
        }
    } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = w * 4 + i;                               // A chunk: tile rows c*8 .. c*8+7
        const int row_t = c * 8 + (lane >> 3), row16 = row_t & 15;
        const int piece = (lane & 7) ^ ((row16 >> 1) & 7);
        int m = m0 + row_t;
        m = m < a.M ? m : a.M - 1;
        asrc[i] = (const char*)a.A + (size_t)m * a.lda * 4 + piece * 16;
        aoff[i] = (uint32_t)(m - m0) * (uint32_t)a.lda * 4u + piece * 16;
        if constexpr (CONV) {
            const int sq = a.tok_seq[m];
            cv_t[i] = a.tok_t[m];
            cv_T[i] = a.seq_T[sq];
            cv_base[i] = (const char*)a.A + (size_t)a.seq_start[sq] * a.lda * 4 + piece * 16;
            cv_zero[i] = (const char*)a.zero_row + piece * 16;
        }
    }
    }
    // The weight fragments never touch LDS: they are stored in fragment order (one contiguous KiB per (n-tile, K tile, plane)), so the
    // wave loads its own twelve straight into registers with plain coalesced loads, one K tile ahead (two register sets, the loop is
    // unrolled by two).  An LDS-DMA piece costs 60-185 cycles of issue beside MFMAs (MI355X guide).
    const char* wbase[4];                                               // wave-uniform
    const uint32_t lane16 = lane * 16;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        int ntile = nt0 + wc * 4 + nt;
        ntile = ntile < ntiles ? ntile : ntiles - 1;
        wbase[nt] = (const char*)a.Wp + (size_t)ntile * nk * 3072;
    }
    auto issue_a = [&](int kt, int buf) {
        char* base = pf_sm + buf * STAGE;
        if constexpr (APL) {
#pragma unroll
            for (int i = 0; i < 6; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pf_uni(abase + (size_t)(i >> 1) * a.a_planes * 2 + (size_t)kt * 64) + aoff[i]),
                                                 (__attribute__((address_space(3))) void*)(base + (i >> 1) * 8192 + (w + 4 * (i & 1)) * 1024), 16, 0, 0);
            return;
        }
        int tap = 0, rem = kt;
        if constexpr (CONV) { tap = kt / cv_kpt; rem = kt - tap * cv_kpt; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const char* ap = asrc[i] + (size_t)kt * 128;
            if constexpr (CONV) {
                const int maxpad = cv_left;
                const int Tv = cv_T[i] <= maxpad ? maxpad + 1 : cv_T[i];
                int p = cv_t[i] + tap * a.conv_dil - cv_left;
                p = p < 0 ? -p : p;
                p = p >= Tv ? 2 * (Tv - 1) - p : p;
                const bool ok = p >= 0 && p < cv_T[i];
                ap = ok ? cv_base[i] + ((size_t)p * a.lda + (size_t)rem * 32) * 4 : cv_zero[i];
            }
            if constexpr (CONV)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ap,
                                                 (__attribute__((address_space(3))) void*)(base + (w * 4 + i) * 1024), 16, 0, 0);
            else
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pf_uni(abase + (size_t)kt * 128) + aoff[i]),
                                                 (__attribute__((address_space(3))) void*)(base + (w * 4 + i) * 1024), 16, 0, 0);
        }
    };
    auto load_w = [&](int kt, v4u (&bw)[4][3]) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int p = 0; p < 3; ++p) bw[nt][p] = *(const __attribute__((address_space(1))) v4u*)(pf_uni(wbase[nt] + (size_t)kt * 3072) + lane16 + p * 1024);
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int row16 = lane & 15, kg = lane >> 4;
    int a_off[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const int pos = (s2 * 4 + kg) ^ ((row16 >> 1) & 7);
        a_off[s2] = (row16 >> 3) * 1024 + ((row16 & 7) * 8 + pos) * 16;
    }
    const int a_wave = wr * 4 * 2048;
    const int a_off_pl = (row16 * 4 + (kg ^ ((row16 >> 2) & 3))) * 16;     // APL: slot of k-group kg in row row16 of a 1 KiB chunk (16 rows x 64 B)

    // one K tile: `bw` holds its weight fragments (loaded during the previous tile), `bn_` receives the next tile's.  Software pipeline
    // over the four m-tiles: the operand split of m-tile mt + 1 (two LDS reads, 44 VALU ops) is issued in the shadow of m-tile mt's
    // 4 x NPROD MFMAs (SCHED: 2 MFMAs, then 3 VALU ops, ...) instead of as a burst in front of them.
    auto ktile = [&](int kt, v4u (&bw)[4][3], v4u (&bn_)[4][3]) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this tile's A DMA and weight loads have landed
        __syncthreads();                                           // ... for every wave; the other A stage is free again
        if (kt + 1 < nk) {
            issue_a(kt + 1, (kt + 1) & 1);
            load_w(kt + 1, bn_);
        }
        // (A raised wave priority -- s_setprio 1 / 2 / 3 -- on the MFMA / split section from here to the end of the K tile, the request section above
        // at 0, changes nothing: GEMM shapes within +-1.5 %, 64-utterance solve 321.8 vs 321.3 / 319.6 / 322.0 ms, profiles/r06b.  Not kept.)
        const char* base = pf_sm + (kt & 1) * STAGE;
        v4u ap[3], an[3];
        if constexpr (APL) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) ap[pl] = *(const v4u*)(base + pl * 8192 + wr * 4096 + a_off_pl);
        } else {
            const f32x4 p0 = *(const f32x4*)(base + a_wave + a_off[0]);
            const f32x4 p1 = *(const f32x4*)(base + a_wave + a_off[1]);
            x3_split8(p0, p1, ap[0], ap[1], ap[2]);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            if (mt < 3) {
                if constexpr (APL) {
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) an[pl] = *(const v4u*)(base + pl * 8192 + wr * 4096 + (mt + 1) * 1024 + a_off_pl);
                } else {
                const f32x4 p0 = *(const f32x4*)(base + a_wave + (mt + 1) * 2048 + a_off[0]);
                const f32x4 p1 = *(const f32x4*)(base + a_wave + (mt + 1) * 2048 + a_off[1]);
                x3_split8(p0, p1, an[0], an[1], an[2]);
                }
            }
            // plane pairs, smallest terms first; four independent accumulators between two MFMAs on the same one
            constexpr int PA[8] = {2, 1, 2, 0, 1, 1, 0, 0};
            constexpr int PB[8] = {1, 2, 0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int q = 8 - NPROD; q < 8; ++q)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ap[PA[q]]),
                                                                          __builtin_bit_cast(bf16x8_t, bw[nt][PB[q]]), acc[mt][nt], 0, 0, 0);
            if (APL && mt < 3) {
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);          // the next m-tile's three plane fragments, then this one's MFMAs
                __builtin_amdgcn_sched_group_barrier(0x008, 4 * NPROD, 0);
            }
            if (!APL && SCHED && mt < 3) {
                // A REQUEST, not a guarantee: hipcc's group solver does not resolve this pipeline (ISA of this build: the splits come out as bursts of
                // 36 / 52 / 44 / 44 VALU ops between clumps of 24 MFMAs; the two waves a SIMD holds overlap each other's bursts instead).  An exact-fit
                // request -- 22 x (1 MFMA, 2 VALU ops) + the rest, one scheduling region per m-tile -- is resolved and measures the same (DESIGN
                // section 8 item 1b, profiles/r05r): the matrix pipe is not waiting for VALU work.
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);          // the next m-tile's two fragment pieces
#pragma unroll
                for (int i = 0; i < 2 * NPROD; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);      // 2 MFMAs ...
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);      // ... 3 VALU ops of the next split
                }
            }
            if (mt < 3) { ap[0] = an[0]; ap[1] = an[1]; ap[2] = an[2]; }
        }
    };
    v4u bw0[4][3], bw1[4][3];
    issue_a(0, 0);
    load_w(0, bw0);
    for (int kt = 0; kt < nk; kt += 2) {
        ktile(kt, bw0, bw1);
        if (kt + 1 < nk) ktile(kt + 1, bw1, bw0);
    }
    // epilogue: the f32 tile kernel's vector path
    __syncthreads();
    float* ct = (float*)pf_sm;
    const int g = lane >> 4, c16 = lane & 15;
    if (EPI == EPI_QKV_ROPE && a.D % 128 == 0 && nt0 * 16 >= 2 * a.D) {       // block-uniform: a V tile
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) *(f32x4*)(ct + (wc * 64 + nt * 16 + c16) * 132 + wr * 64 + mt * 16 + g * 4) = acc[mt][nt];
        __syncthreads();
        pf_store_vt<128, 256, true>(a, ct, m0, nt0 * 16, threadIdx.x);
        return;
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                ct[(wr * 64 + mt * 16 + g * 4 + rr) * 128 + ((wc * 64 + nt * 16 + c16) ^ (g << 4))] = acc[mt][nt][rr];
    int* meta = (int*)(pf_sm + 65536);
    pf_stage_meta<EPI, 128>(a, meta, m0, threadIdx.x);
    __syncthreads();
    pf_store_tile<EPI, 128, 256, true>(a, ct, meta, m0, nt0 * 16, threadIdx.x);
}


// ================================================================================================================
// The same GEMM with EIGHT waves per 128 x 128 block: 4 (rows) x 2 (columns), wave tile 32 x 64 -- two m-tiles instead of four, so the accumulators
// (32 registers), ONE set of weight fragments (48) and the operand planes fit 128 registers and two resident blocks put FOUR waves on a SIMD instead
// of two.  The product structure's ablations show its stalls as serialised per wave (requests, barrier, split, store: each +12 ... +25 % when removed,
// profiles/r05b) and two waves per SIMD as too few to hide them; the standalone model of this form measured +2 ... +4 % on every s2mel shape
// (tools/microbench/x3_gemm_lab4.hip, profiles/r06j).  Weights go through LDS (no register double buffer fits): one LDS-DMA'd copy of the K tile's 8
// n-tiles x 3 planes per block (24 KiB, 3 pieces per wave) beside the A stage (16 KiB, 2 pieces per wave): 2 x 40 KiB of stages = 80 KiB, exactly two
// blocks per CU; the epilogue image reuses it.  Per output element the MFMA sequence -- K tiles in order, plane pairs in order -- is the 4-wave kernel's:
// the results are bit-identical (tests/test_gpu_gemm_x3.py).  6 plane products; option x3_waves = 4 selects the 4-wave kernel.
#define X3W8_STAGE (16384 + 24576)
#define X3W8_LDS (2 * X3W8_STAGE)
template <int EPI, bool CONV = false>
__global__ __launch_bounds__(512, 2) void gemm_x3w8_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char pf_sm[];      // [2][A 16 KiB | W 24 KiB]; the epilogue image is smaller
    constexpr int NPROD = 6;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = w >> 1, wc = w & 1;
    const int n_mt = (a.M + PF_BM - 1) / PF_BM, n_nt = (a.N + PF_BN - 1) / PF_BN;
    const int total = n_mt * n_nt, per = (total + 7) >> 3;
    const int t = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (t >= total) return;
    const int g0 = t / (PF_GM * n_nt), first_m = g0 * PF_GM;
    const int gm = (n_mt - first_m) < PF_GM ? (n_mt - first_m) : PF_GM;
    const int r = t - g0 * PF_GM * n_nt;
    const int bn = r / gm, bm = first_m + (r - bn * gm);
    const int m0 = bm * PF_BM, nt0 = bn * (PF_BN / 16);
    const int nk = a.K >> 5;
    const int ntiles = (a.N + 15) >> 4;

    // A pieces of this wave: tile rows (2 w + i) * 8 .. + 7 (the 4-wave kernel's image and swizzle)
    const char* asrc[2];
    uint32_t aoff[2];
    const char* abase = (const char*)a.A + (size_t)m0 * a.lda * 4;
    int cv_t[2], cv_T[2];
    const char* cv_base[2];
    const char* cv_zero[2];
    const int cv_kpt = CONV ? a.conv_W / 32 : 1;
    const int cv_left = CONV ? (a.conv_taps - 1) * a.conv_dil - ((a.conv_taps - 1) * a.conv_dil) / 2 : 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = w * 2 + i;
        const int row_t = c * 8 + (lane >> 3), row16 = row_t & 15;
        const int piece = (lane & 7) ^ ((row16 >> 1) & 7);
        int m = m0 + row_t;
        m = m < a.M ? m : a.M - 1;
        asrc[i] = (const char*)a.A + (size_t)m * a.lda * 4 + piece * 16;
        aoff[i] = (uint32_t)(m - m0) * (uint32_t)a.lda * 4u + piece * 16;
        if constexpr (CONV) {
            const int sq = a.tok_seq[m];
            cv_t[i] = a.tok_t[m];
            cv_T[i] = a.seq_T[sq];
            cv_base[i] = (const char*)a.A + (size_t)a.seq_start[sq] * a.lda * 4 + piece * 16;
            cv_zero[i] = (const char*)a.zero_row + piece * 16;
        }
    }
    // W pieces of this wave: q = w, w + 8, w + 16 of the block's 24 (n-tile q / 3, plane q % 3), each one contiguous KiB per K tile
    const char* wdma[3];
    const uint32_t lane16 = lane * 16;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int q = w + 8 * i;
        int ntile = nt0 + q / 3;
        ntile = ntile < ntiles ? ntile : ntiles - 1;
        wdma[i] = (const char*)a.Wp + (size_t)ntile * nk * 3072 + (q % 3) * 1024;
    }
    auto issue = [&](int kt, int buf) {
        char* base = pf_sm + buf * X3W8_STAGE;
        int tap = 0, rem = kt;
        if constexpr (CONV) { tap = kt / cv_kpt; rem = kt - tap * cv_kpt; }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if constexpr (CONV) {
                const int maxpad = cv_left;
                const int Tv = cv_T[i] <= maxpad ? maxpad + 1 : cv_T[i];
                int p = cv_t[i] + tap * a.conv_dil - cv_left;
                p = p < 0 ? -p : p;
                p = p >= Tv ? 2 * (Tv - 1) - p : p;
                const bool ok = p >= 0 && p < cv_T[i];
                const char* ap = ok ? cv_base[i] + ((size_t)p * a.lda + (size_t)rem * 32) * 4 : cv_zero[i];
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ap,
                                                 (__attribute__((address_space(3))) void*)(base + (w * 2 + i) * 1024), 16, 0, 0);
            } else {
                (void)asrc;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pf_uni(abase + (size_t)kt * 128) + aoff[i]),
                                                 (__attribute__((address_space(3))) void*)(base + (w * 2 + i) * 1024), 16, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pf_uni(wdma[i] + (size_t)kt * 3072) + lane16),
                                             (__attribute__((address_space(3))) void*)(base + 16384 + (w + 8 * i) * 1024), 16, 0, 0);
    };

    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int row16 = lane & 15, kg = lane >> 4;
    int a_off[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const int pos = (s2 * 4 + kg) ^ ((row16 >> 1) & 7);
        a_off[s2] = (row16 >> 3) * 1024 + ((row16 & 7) * 8 + pos) * 16;
    }
    const int a_wave = wr * 2 * 2048;

    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this tile's A and W pieces have landed
        __syncthreads();                                           // ... for every wave; the other stage is free again
        if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
        const char* base = pf_sm + (kt & 1) * X3W8_STAGE;
        v4u bw[4][3];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int p = 0; p < 3; ++p) bw[nt][p] = *(const v4u*)(base + 16384 + ((wc * 4 + nt) * 3 + p) * 1024 + lane16);
        v4u ap[3], an[3];
        {
            const f32x4 p0 = *(const f32x4*)(base + a_wave + a_off[0]);
            const f32x4 p1 = *(const f32x4*)(base + a_wave + a_off[1]);
            x3_split8(p0, p1, ap[0], ap[1], ap[2]);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            if (mt < 1) {
                const f32x4 p0 = *(const f32x4*)(base + a_wave + (mt + 1) * 2048 + a_off[0]);
                const f32x4 p1 = *(const f32x4*)(base + a_wave + (mt + 1) * 2048 + a_off[1]);
                x3_split8(p0, p1, an[0], an[1], an[2]);
            }
            constexpr int PA[8] = {2, 1, 2, 0, 1, 1, 0, 0};
            constexpr int PB[8] = {1, 2, 0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int q = 8 - NPROD; q < 8; ++q)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ap[PA[q]]),
                                                                          __builtin_bit_cast(bf16x8_t, bw[nt][PB[q]]), acc[mt][nt], 0, 0, 0);
            if (mt < 1) {
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);          // the next m-tile's two fragment pieces
#pragma unroll
                for (int i = 0; i < 2 * NPROD; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                }
                ap[0] = an[0]; ap[1] = an[1]; ap[2] = an[2];
            }
        }
    }
    // epilogue: the 4-wave kernel's, with this kernel's wave tiles in the image and 512 storing threads
    __syncthreads();
    float* ct = (float*)pf_sm;
    const int g = lane >> 4, c16 = lane & 15;
    if (EPI == EPI_QKV_ROPE && a.D % 128 == 0 && nt0 * 16 >= 2 * a.D) {       // block-uniform: a V tile
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) *(f32x4*)(ct + (wc * 64 + nt * 16 + c16) * 132 + wr * 32 + mt * 16 + g * 4) = acc[mt][nt];
        __syncthreads();
        pf_store_vt<128, 512, true>(a, ct, m0, nt0 * 16, threadIdx.x);
        return;
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                ct[(wr * 32 + mt * 16 + g * 4 + rr) * 128 + ((wc * 64 + nt * 16 + c16) ^ (g << 4))] = acc[mt][nt][rr];
    int* meta = (int*)(pf_sm + 65536);
    pf_stage_meta<EPI, 128>(a, meta, m0, threadIdx.x);
    __syncthreads();
    pf_store_tile<EPI, 128, 512, true>(a, ct, meta, m0, nt0 * 16, threadIdx.x);
}

template <int EPI, bool CONV = false>
static int launch_gemm_x3_e(const GemmArgs& a, hipStream_t st) {
    const int n_mt = ceil_div(a.M, PF_BM), n_nt = ceil_div(a.N, PF_BN);
    const int per = ceil_div(n_mt * n_nt, 8);
    const int nprod = itts_opt(OPT_X3_PRODUCTS) == 6 ? 6 : 8;
    const bool sched = itts_opt(OPT_X3_SCHED) != 0;                 // A/B switch of the MFMA / split interleave
    static ItPerDevice<bool> attr_set_pd;
    bool& attr_set = attr_set_pd.cur();
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)gemm_x3_kernel<EPI, CONV, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS_ONE));
        HIP_TRY(hipFuncSetAttribute((const void*)gemm_x3_kernel<EPI, CONV, 6, true>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS_ONE));
        HIP_TRY(hipFuncSetAttribute((const void*)gemm_x3_kernel<EPI, CONV, 8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS_ONE));
        HIP_TRY(hipFuncSetAttribute((const void*)gemm_x3_kernel<EPI, CONV, 6, false>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS_ONE));
        attr_set = true;
    }
    const dim3 grid(per * 8), blk(256);
    // Residency: two blocks per CU for every variant.  (Round 4 pinned the variants that are not the default one to one block per CU because their
    // solves differed run to run when two blocks shared a CU; the differences came from the fused wqkv epilogue's packed RoPE arithmetic, pf_rope4,
    // not from the main loop.  Option x3_pin = 1 restores the pin as a diagnostic.)
    const size_t lds = ((nprod == 6 && sched) || itts_opt(OPT_X3_PIN) == 0) ? X3_LDS : X3_LDS_ONE;
    if constexpr (!CONV && (EPI == EPI_QKV_ROPE || EPI == EPI_SWIGLU)) {     // the two GEMMs behind an adaptive RMSNorm: A as bf16 planes
        if (a.a_planes) {
            if (nprod != 6 || !sched) { itts_set_error("gemm (f32x3): the plane-operand form exists for the shipped variant only (6 products, interleaved)"); return ITTS_ERR_ARG; }
            static ItPerDevice<bool> apl_set_pd;
            bool& apl_set = apl_set_pd.cur();
            if (!apl_set) {
                HIP_TRY(hipFuncSetAttribute((const void*)gemm_x3_kernel<EPI, CONV, 6, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS_ONE));
                apl_set = true;
            }
            hipLaunchKernelGGL((gemm_x3_kernel<EPI, CONV, 6, true, true>), grid, blk, lds, st, a);
            HIP_TRY(hipGetLastError());
            return ITTS_OK;
        }
    } else if (a.a_planes) { itts_set_error("gemm (f32x3): plane operands are supported for the wqkv / SwiGLU GEMMs only"); return ITTS_ERR_ARG; }
    if (nprod == 6 && sched && itts_opt(OPT_X3_WAVES) == 8 && itts_opt(OPT_X3_PIN) == 0) {     // the default: eight waves per block (gemm_x3w8_kernel)
        static ItPerDevice<bool> w8_set_pd;
        bool& w8_set = w8_set_pd.cur();
        if (!w8_set) {
            HIP_TRY(hipFuncSetAttribute((const void*)gemm_x3w8_kernel<EPI, CONV>, hipFuncAttributeMaxDynamicSharedMemorySize, X3W8_LDS));
            w8_set = true;
        }
        hipLaunchKernelGGL((gemm_x3w8_kernel<EPI, CONV>), grid, dim3(512), X3W8_LDS, st, a);
        HIP_TRY(hipGetLastError());
        return ITTS_OK;
    }
    if (nprod == 6 && sched) hipLaunchKernelGGL((gemm_x3_kernel<EPI, CONV, 6, true>), grid, blk, lds, st, a);
    else if (nprod == 6) hipLaunchKernelGGL((gemm_x3_kernel<EPI, CONV, 6, false>), grid, blk, lds, st, a);
    else if (sched) hipLaunchKernelGGL((gemm_x3_kernel<EPI, CONV, 8, true>), grid, blk, lds, st, a);
    else hipLaunchKernelGGL((gemm_x3_kernel<EPI, CONV, 8, false>), grid, blk, lds, st, a);
    HIP_TRY(hipGetLastError());
    return ITTS_OK;
}

int launch_gemm_x3(const GemmArgs& a, hipStream_t st) {
    if (!pf_f32_ok(a)) {
        itts_set_error("gemm (f32x3): needs K %% 32 == 0, lda %% 4 == 0, N / ldo / D %% 4 == 0, 16-byte aligned rows, no split-K (M=%d N=%d K=%d lda=%d epi=%d)",
                       a.M, a.N, a.K, a.lda, a.epi);
        return ITTS_ERR_ARG;
    }
    if (a.epi == EPI_GATE && a.conv_taps > 0 &&
        (a.conv_W % 32 || a.K != a.conv_taps * a.conv_W || a.lda != a.conv_W || !a.tok_seq || !a.tok_t || !a.seq_start || !a.seq_T || !a.zero_row)) {
        itts_set_error("gemm tap mode (f32x3): need conv_W %% 32 == 0, K == taps * conv_W, lda == conv_W and the sequence tables");
        return ITTS_ERR_ARG;
    }
    switch (a.epi) {
        case EPI_STORE_F32: return launch_gemm_x3_e<EPI_STORE_F32>(a, st);
        case EPI_RESIDUAL: return launch_gemm_x3_e<EPI_RESIDUAL>(a, st);
        case EPI_SWIGLU: return launch_gemm_x3_e<EPI_SWIGLU>(a, st);
        case EPI_GATE: return a.conv_taps > 0 ? launch_gemm_x3_e<EPI_GATE, true>(a, st) : launch_gemm_x3_e<EPI_GATE>(a, st);
        case EPI_QKV_ROPE: return launch_gemm_x3_e<EPI_QKV_ROPE>(a, st);
        case EPI_WN_RS: return launch_gemm_x3_e<EPI_WN_RS>(a, st);
        default: itts_set_error("gemm (f32x3): unsupported epilogue %d", a.epi); return ITTS_ERR_ARG;
    }
}


// diagnostics: resident blocks per CU the runtime predicts for the kernel at its launch configuration (residual epilogue)
int gemm_x3_occupancy(int* blocks) {
    int n = 0;
    (void)hipFuncSetAttribute((const void*)gemm_x3_kernel<EPI_RESIDUAL, false, 6, true>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS);
    const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)gemm_x3_kernel<EPI_RESIDUAL, false, 6, true>, 256, X3_LDS);
    if (e != hipSuccess) { itts_set_error("occupancy query: %s", hipGetErrorString(e)); return ITTS_ERR_HIP; }
    *blocks = n;
    return ITTS_OK;
}
