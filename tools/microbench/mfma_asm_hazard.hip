// GPU box: is an inline-asm VALU instruction that reads the result of the MFMA issued just before it protected against the
// "XDL write VGPR -> VALU read" hazard?  hipcc's hazard recognizer pads compiler-generated VALU consumers of an MFMA result with s_nop; it
// does not look inside inline asm.  The flash-attention kernels take the row maximum of the S^T accumulators with `asm("v_max3_f32 ...")`
// (fa_max3: no NaN canonicalisation) and in their non-tail path that asm is the FIRST reader of accumulators written one to three MFMAs
// earlier.  Here: one v_mfma_f32_16x16x32_bf16 (VGPR destination, operands changing every iteration), GAP wait states of s_nop, an asm
// v_max3_f32 of its first three results; the same maximum is taken again by compiler-generated code (which gets the padding) and compared.
// A mismatch = the asm read the register before the MFMA wrote it.
// build: hipcc --offload-arch=gfx950 -O2 -mllvm -amdgpu-mfma-vgpr-form=1 -o mfma_asm_hazard mfma_asm_hazard.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int GAP, int NMFMA>      // GAP < 0: nothing between the MFMA and the asm; n >= 0: s_nop n.  NMFMA independent MFMAs are issued first (the last is the probed one)
__global__ __launch_bounds__(256) void probe(const uint32_t* in, unsigned* bad, int iters) {
    unsigned n_bad = 0;
    v4u a = *(const v4u*)(in + ((blockIdx.x * 256 + threadIdx.x) & 4095) * 4);
    v4u b = *(const v4u*)(in + ((blockIdx.x * 256 + threadIdx.x + 977) & 4095) * 4);
    f32x4 keep = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        a.x = a.x * 1664525u + 1013904223u; a.x = (a.x & 0x3f7f3f7fu) | 0x3c003c00u;      // bf16 pairs in a tame range, new every iteration
        b.y = b.y * 22695477u + 1u;         b.y = (b.y & 0x3f7f3f7fu) | 0x3c003c00u;
        const f32x4 c = {(float)(it & 1023), 0.5f, -1.f, 2.f};
        f32x4 d = c;
#pragma unroll
        for (int k = 0; k < NMFMA - 1; ++k)
            keep = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, b), __builtin_bit_cast(bf16x8_t, a), keep, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), d, 0, 0, 0);
        float early;
        if (GAP < 0) asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(early) : "v"(d[0]), "v"(d[1]), "v"(d[2]));
        else asm volatile("s_nop %4\n\tv_max3_f32 %0, %1, %2, %3" : "=v"(early) : "v"(d[0]), "v"(d[1]), "v"(d[2]), "n"(GAP));
        const float late = fmaxf(fmaxf(d[0], d[1]), d[2]);           // compiler-generated: padded by the hazard recognizer
        if (early != late) ++n_bad;
        keep[0] += d[3] * 1e-30f;
    }
    if (keep[0] == 123.456f) ++n_bad;
    if (n_bad) atomicAdd(bad, n_bad);
}

template <int GAP, int NMFMA>
static int run(const uint32_t* in, unsigned* cnt, int blocks, int threads, int iters, const char* what) {
    CK(hipMemset(cnt, 0, 4));
    probe<GAP, NMFMA><<<blocks, threads>>>(in, cnt, iters);
    CK(hipDeviceSynchronize());
    unsigned h;
    CK(hipMemcpy(&h, cnt, 4, hipMemcpyDeviceToHost));
    printf("%-34s gap %2d  MFMAs in front %d  blocks %4d x %3d threads: %u of %lld lane-samples differ\n", what, GAP, NMFMA - 1, blocks, threads, h,
           (long long)blocks * threads * iters);
    return 0;
}

int main() {
    uint32_t* in;
    unsigned* cnt;
    CK(hipMalloc(&in, 4096 * 16));
    CK(hipMalloc(&cnt, 4));
    uint32_t host[4096 * 4];
    uint32_t s = 12345u;
    for (int i = 0; i < 4096 * 4; ++i) { s = s * 1664525u + 1013904223u; host[i] = (s & 0x3f7f3f7fu) | 0x3c003c00u; }
    CK(hipMemcpy(in, host, sizeof(host), hipMemcpyHostToDevice));
    const int iters = 4096;
    // one wave per SIMD (256 blocks x 256 threads = 1 block per CU) and a crowded chip (2048 blocks)
    for (int crowd = 0; crowd < 2; ++crowd) {
        const int blocks = crowd ? 2048 : 256;
        const char* what = crowd ? "crowded (8 blocks per CU)" : "one block per CU";
        if (run<-1, 1>(in, cnt, blocks, 256, iters, what)) return 1;
        if (run<0, 1>(in, cnt, blocks, 256, iters, what)) return 1;
        if (run<3, 1>(in, cnt, blocks, 256, iters, what)) return 1;
        if (run<7, 1>(in, cnt, blocks, 256, iters, what)) return 1;
        if (run<11, 1>(in, cnt, blocks, 256, iters, what)) return 1;
        if (run<15, 1>(in, cnt, blocks, 256, iters, what)) return 1;
        if (run<-1, 2>(in, cnt, blocks, 256, iters, what)) return 1;
        if (run<-1, 4>(in, cnt, blocks, 256, iters, what)) return 1;
    }
    // a single wave on the whole chip (no co-issue at all)
    if (run<-1, 1>(in, cnt, 1, 64, iters, "single wave")) return 1;
    if (run<3, 1>(in, cnt, 1, 64, iters, "single wave")) return 1;
    if (run<11, 1>(in, cnt, 1, 64, iters, "single wave")) return 1;
    return 0;
}
