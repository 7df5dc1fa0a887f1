// GPU box: the f32 -> three-bf16-plane split (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)) with the residuals taken by v_dot2c_f32_bf16
// (gfx950: D += A.lo * B.lo + A.hi * B.hi on bf16 pairs, f32 accumulator) instead of shift / and + v_sub_f32:
//     r0 = x0 + h.lo * (-1) + h.hi * 0        r1 = x1 + h.lo * 0 + h.hi * (-1)
// 7 VALU instructions per pair of values instead of 11.  x - h is exactly representable (h keeps the top 8 significand bits of x), so any correctly
// rounded implementation returns the and / sub residual bit for bit -- IF the dot unit does not truncate the accumulator when it aligns the addends
// and does not flush.  This program (1) compares the planes of both forms bitwise over random bit patterns of every exponent plus the rounding edge
// cases (carry into the next binade, ties, signed zeros, subnormals), with the constants as opaque SGPR values and as compiler-chosen immediates;
// (2) times both forms in a register-resident loop (VALU only) and beside MFMAs issued by the same wave.
// build: hipcc --offload-arch=gfx950 -O2 -fno-slp-vectorize -o bin/split_dot2 split_dot2.hip
// Measured (profiles/r05k): with SGPR constants the planes are bit-identical for every operand between 2^-101 and the top binade (137 of 8.4 M pairs
// differ, all outside); with the constant left to the compiler (it folds 0x0000bf80 into the inline constant -1.0) almost every pair differs -- the
// inline constant is not the packed pair.  Alone the dot2c form is 8 % faster at equal instruction count, but BESIDE MFMAs it is 6 % slower than
// shift / and / sub with 64 more instructions (dot ops contend with the MFMAs), and the x3 GEMMs lost 9-12 %, the 64-utterance solve 4-10 %
// (same box, same bits: profiles/r05k/ab_dot2_vs_shipped.txt).  Not shipped.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t cvt2(float a, float b) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{a, b}, bf2)); }

// MODE 0: shift / and + v_sub_f32 (the shipped form).  1: v_dot2c with the two constants in SGPRs.  2: v_dot2c, constants left to the compiler.
template <int MODE>
__device__ __forceinline__ void split2(float a, float b, uint32_t& H, uint32_t& M, uint32_t& L, uint32_t c0, uint32_t c1) {
    const uint32_t h = cvt2(a, b);
    float ra, rb;
    if (MODE == 0) { ra = a - __uint_as_float(h << 16); rb = b - __uint_as_float(h & 0xffff0000u); }
    else {
        ra = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, h), __builtin_bit_cast(bf2, c0), a, false);
        rb = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, h), __builtin_bit_cast(bf2, c1), b, false);
    }
    const uint32_t m = cvt2(ra, rb);
    float sa, sb;
    if (MODE == 0) { sa = ra - __uint_as_float(m << 16); sb = rb - __uint_as_float(m & 0xffff0000u); }
    else {
        sa = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, m), __builtin_bit_cast(bf2, c0), ra, false);
        sb = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, m), __builtin_bit_cast(bf2, c1), rb, false);
    }
    H = h; M = m; L = cvt2(sa, sb);
}

template <int MODE>
__device__ __forceinline__ void consts(uint32_t& c0, uint32_t& c1) {
    c0 = 0x0000bf80u; c1 = 0xbf800000u;                              // (-1, 0) and (0, -1) as bf16 pairs (lo, hi)
    if (MODE == 1) asm volatile("" : "+s"(c0), "+s"(c1));
}

template <int MODE>
__global__ void planes_kernel(const float* __restrict__ x, uint32_t* __restrict__ out, size_t pairs) {
    uint32_t c0, c1;
    consts<MODE>(c0, c1);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < pairs; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h, m, l;
        split2<MODE>(x[2 * i], x[2 * i + 1], h, m, l, c0, c1);
        out[3 * i] = h; out[3 * i + 1] = m; out[3 * i + 2] = l;
    }
}

// timing: 32 values per lane (one K tile of the x3 GEMM's wave: 4 m-tiles x 8), ITER rounds; WITH_MFMA adds the 96 MFMAs of the K tile
template <int MODE, bool WITH_MFMA>
__global__ __launch_bounds__(256, 2) void time_kernel(const float* __restrict__ x, float* __restrict__ sink, long long* cyc, int iters) {
    uint32_t c0, c1;
    consts<MODE>(c0, c1);
    float v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = x[(threadIdx.x * 32 + i) & 4095];
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    v4u bw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bw[i] = v4u{0x3f803f80u + i, 0x3f003f00u, 0x3e803e80u, 0x3f803f80u};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            v4u P[3];
#pragma unroll
            for (int i = 0; i < 4; ++i) { uint32_t h, m, l; split2<MODE>(v[mt * 8 + 2 * i], v[mt * 8 + 2 * i + 1], h, m, l, c0, c1); P[0][i] = h; P[1][i] = m; P[2][i] = l; }
            if (WITH_MFMA) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int q = 0; q < 6; ++q)
                        acc[mt * 4 + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, P[q % 3]), __builtin_bit_cast(bf16x8_t, bw[nt]),
                                                                                   acc[mt * 4 + nt], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[mt][i] += __uint_as_float(P[0][i] ^ P[1][i] ^ P[2][i]);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) v[mt * 8 + i] = v[mt * 8 + i] * 1.0000001f + 1e-3f;        // new values next round (2 more VALU per pair in every mode)
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

static uint32_t rng_state = 0x12345u;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 17; rng_state ^= rng_state << 5; return rng_state; }

int main() {
    const size_t n = 1u << 24, pairs = n / 2;
    float* hx = (float*)malloc(n * 4);
    uint32_t* hb = (uint32_t*)hx;
    size_t k = 0;
    // edge cases first: every exponent x mantissas that tie, carry or sit at the ends, both signs
    const uint32_t mant[] = {0u, 1u, 0x7fffffu, 0x7fff80u, 0x7fff7fu, 0x008000u, 0x008001u, 0x007fffu, 0x018000u, 0x00ff80u, 0x400000u, 0x3fffffu, 0x7f8000u, 0x7f7fffu,
                             0x000080u, 0x000040u, 0x0000c0u, 0x00807fu, 0x0080ffu, 0x555555u, 0x2aaaaau};
    for (uint32_t e = 0; e < 255; ++e)
        for (uint32_t mi = 0; mi < sizeof(mant) / 4; ++mi)
            for (uint32_t s = 0; s < 2; ++s) hb[k++] = (s << 31) | (e << 23) | mant[mi];
    const size_t edge = k;
    for (; k < n; ++k) {                                               // random patterns, finite only; half of them in the engine's magnitude range
        uint32_t b = rnd();
        if (k & 1) b = (b & 0x807fffffu) | ((100u + (rnd() % 40u)) << 23);
        if (((b >> 23) & 255u) == 255u) b &= 0xbfffffffu;
        hb[k] = b;
    }
    float* dx; uint32_t* d[3];
    CK(hipMalloc(&dx, n * 4));
    CK(hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice));
    uint32_t* ho[3];
    for (int m = 0; m < 3; ++m) { CK(hipMalloc(&d[m], pairs * 12)); ho[m] = (uint32_t*)malloc(pairs * 12); }
    planes_kernel<0><<<2048, 256>>>(dx, d[0], pairs);
    planes_kernel<1><<<2048, 256>>>(dx, d[1], pairs);
    planes_kernel<2><<<2048, 256>>>(dx, d[2], pairs);
    CK(hipDeviceSynchronize());
    for (int m = 0; m < 3; ++m) CK(hipMemcpy(ho[m], d[m], pairs * 12, hipMemcpyDeviceToHost));
    for (int m = 1; m < 3; ++m) {
        size_t bad = 0, bad_sub = 0, bad_norm = 0, shown = 0;
        for (size_t i = 0; i < pairs; ++i) {
            if (ho[m][3 * i] == ho[0][3 * i] && ho[m][3 * i + 1] == ho[0][3 * i + 1] && ho[m][3 * i + 2] == ho[0][3 * i + 2]) continue;
            ++bad;
            const uint32_t e0 = (hb[2 * i] >> 23) & 255u, e1 = (hb[2 * i + 1] >> 23) & 255u;
            const bool tiny = e0 < 26u || e1 < 26u, huge = e0 >= 254u || e1 >= 254u;         // a residual can be subnormal below 2^-101; bf16(x) can overflow at the top
            if (tiny || huge) ++bad_sub; else ++bad_norm;
            if (!tiny && !huge && shown < 8) {
                ++shown;
                printf("  mode %d pair %zu x = %08x %08x : h %08x/%08x m %08x/%08x l %08x/%08x\n", m, i, hb[2 * i], hb[2 * i + 1], ho[m][3 * i], ho[0][3 * i],
                       ho[m][3 * i + 1], ho[0][3 * i + 1], ho[m][3 * i + 2], ho[0][3 * i + 2]);
            }
        }
        printf("planes, dot2 form %s vs shift/and/sub: %zu pairs (%zu edge values first), %zu differ: %zu with an operand below 2^-101 or in the top binade, %zu others\n",
               m == 1 ? "(SGPR constants)" : "(compiler constants)", pairs, edge, bad, bad_sub, bad_norm);
    }
    // timing
    float* sink; long long* dc;
    const int blocks = 512, iters = 2000;
    CK(hipMalloc(&sink, blocks * 256 * 4)); CK(hipMalloc(&dc, blocks * 8));
    long long* hc = (long long*)malloc(blocks * 8);
    for (int pass = 0; pass < 2; ++pass)
        for (int cfg = 0; cfg < 4; ++cfg) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0));
            if (cfg == 0) time_kernel<0, false><<<blocks, 256>>>(dx, sink, dc, iters);
            if (cfg == 1) time_kernel<1, false><<<blocks, 256>>>(dx, sink, dc, iters);
            if (cfg == 2) time_kernel<0, true><<<blocks, 256>>>(dx, sink, dc, iters);
            if (cfg == 3) time_kernel<1, true><<<blocks, 256>>>(dx, sink, dc, iters);
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(hc, dc, blocks * 8, hipMemcpyDeviceToHost));
            double avg = 0; for (int b = 0; b < blocks; ++b) avg += (double)hc[b]; avg /= blocks;
            if (pass) printf("%s%s: %.3f ms, %.0f clock64 ticks per round of 32 values per lane%s\n", cfg & 1 ? "dot2c split" : "shift/and/sub split",
                             cfg & 2 ? " + 96 MFMAs" : "", ms, avg / iters,
                             cfg & 2 ? " (x3 GEMM K tile of one wave; 2 blocks of 4 waves per CU)" : "");
        }
    return 0;
}
