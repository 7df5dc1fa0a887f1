// GPU box: the instruction sequence of the fused wqkv epilogue's RoPE as hipcc's SLP vectoriser emitted it (gemm_prefill_kernel<EPI_QKV_ROPE>, bf16):
//   global_load_dwordx4 v[30:33] (cos0 sin0 cos1 sin1) ; v_mov_b32 v34, v21 ; s_waitcnt vmcnt(0)
//   v_pk_mul_f32 v[36:37], v[6:7], v[30:31] op_sel:[1,1] op_sel_hi:[1,0]
//   v_pk_mul_f32 v[34:35], v[34:35], v[32:33] op_sel:[0,1] op_sel_hi:[0,0]          <- IN PLACE, its low source register feeds both halves
//   v_pk_fma_f32 v[38:39], v[6:7], v[30:31], v[36:37] ... neg ; v_pk_fma_f32 v[6:7], ... ; v_pk_fma_f32 v[30:31], v[20:21], v[32:33], v[34:35] ... neg ; ...
// In the engine about one quarter-wave per 2 x 13 launches stored v2 c1 instead of v2 c1 - v3 s1 (the component fed by the in-place product's low half),
// and only while a second block shared the CU (profiles/r05a/capture.log).  Here the sequence runs verbatim (asm, fixed registers) in "probe" blocks
// whose results are compared with compiler-generated scalar arithmetic, beside "hammer" blocks that keep the CU's matrix pipe, LDS-DMA path and LDS
// busy the way the GEMM main loop of the neighbouring block does.  Second kernel: the same sequence with the product written to a separate register pair.
// build: hipcc --offload-arch=gfx950 -O2 -o bin/pk_inplace_hazard pk_inplace_hazard.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#define CLOB "v6", "v7", "v8", "v20", "v21", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41"
#define HEAD "v_mov_b32 v6, %4\n\tv_mov_b32 v7, %5\n\tv_mov_b32 v20, %6\n\tv_mov_b32 v21, %7\n\tv_mov_b32 v30, %8\n\tv_mov_b32 v31, %9\n\t" \
             "ds_read_b32 v8, %10\n\tglobal_load_dwordx4 v[30:33], v[30:31], off\n\tv_mov_b32 v34, v21\n\ts_waitcnt vmcnt(0)\n\t"                \
             "v_pk_mul_f32 v[36:37], v[6:7], v[30:31] op_sel:[1,1] op_sel_hi:[1,0]\n\t"
#define TAIL "v_cvt_pk_bf16_f32 v6, v38, v7\n\tv_cvt_pk_bf16_f32 v7, v30, v21\n\ts_waitcnt lgkmcnt(0)\n\t"                                      \
             "v_mov_b32 %0, v6\n\tv_mov_b32 %1, v7\n\tv_mov_b32 %2, v30\n\tv_mov_b32 %3, v21"
#define INPLACE "v_pk_mul_f32 v[34:35], v[34:35], v[32:33] op_sel:[0,1] op_sel_hi:[0,0]\n\t"                                                      \
                "v_pk_fma_f32 v[38:39], v[6:7], v[30:31], v[36:37] op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"                            \
                "v_pk_fma_f32 v[6:7], v[6:7], v[30:31], v[36:37] op_sel_hi:[0,1,1]\n\t"                                                            \
                "v_pk_fma_f32 v[30:31], v[20:21], v[32:33], v[34:35] op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"                          \
                "v_pk_fma_f32 v[20:21], v[20:21], v[32:33], v[34:35] op_sel_hi:[0,1,1]\n\t"
#define APART "v_pk_mul_f32 v[40:41], v[34:35], v[32:33] op_sel:[0,1] op_sel_hi:[0,0]\n\t"                                                        \
              "v_pk_fma_f32 v[38:39], v[6:7], v[30:31], v[36:37] op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"                              \
              "v_pk_fma_f32 v[6:7], v[6:7], v[30:31], v[36:37] op_sel_hi:[0,1,1]\n\t"                                                              \
              "v_pk_fma_f32 v[30:31], v[20:21], v[32:33], v[40:41] op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"                            \
              "v_pk_fma_f32 v[20:21], v[20:21], v[32:33], v[40:41] op_sel_hi:[0,1,1]\n\t"

__device__ __forceinline__ uint32_t cvt2(float a, float b) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{a, b}, bf2));
}

template <bool INPL>
__global__ __launch_bounds__(256) void probe(const float* rope, const float* vals, unsigned* bad, unsigned* bad_comp, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (blockIdx.x & 1) {
        // hammer: LDS-DMA pieces, fragment reads and MFMAs, as the neighbouring block's GEMM main loop
        f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        const char* src = (const char*)vals + ((size_t)(blockIdx.x * 4 + w) & 1023) * 4096 + lane * 16;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ((it + i) & 3) * 1024),
                                                 (__attribute__((address_space(3))) void*)(sm + 8192 + ((it & 1) * 16 + w * 4 + i) * 1024), 16, 0, 0);
            const v4u a = *(const v4u*)(sm + 8192 + (((it + 1) & 1) * 16 + w * 4) * 1024 + lane * 16);
            const v4u b = *(const v4u*)(sm + 8192 + (((it + 1) & 1) * 16 + w * 4 + 1) * 1024 + lane * 16);
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc[n], 0, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 1.2345e-30f) sink[threadIdx.x] = acc[0][0];
        return;
    }
    unsigned n_bad = 0, comp = 0;
    float* lds_f = (float*)sm;
    lds_f[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    uint32_t s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        const f32x4 v = *(const f32x4*)(vals + ((s >> 8) & 0xfffff) * 4);          // the accumulator values (LDS image in the engine)
        const float* rp = rope + ((s >> 6) & 0x3fff) * 4;                          // (cos0, sin0, cos1, sin1) of this row / column chunk
        uint32_t p01, p23;
        float y2, y3;
        const uint32_t lds_addr = (uint32_t)(threadIdx.x * 4);
        if (INPL)
            asm volatile(HEAD INPLACE TAIL : "=v"(p01), "=v"(p23), "=v"(y2), "=v"(y3)
                         : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"((uint32_t)(uintptr_t)rp), "v"((uint32_t)((uintptr_t)rp >> 32)), "v"(lds_addr)
                         : CLOB, "memory");
        else
            asm volatile(HEAD APART TAIL : "=v"(p01), "=v"(p23), "=v"(y2), "=v"(y3)
                         : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"((uint32_t)(uintptr_t)rp), "v"((uint32_t)((uintptr_t)rp >> 32)), "v"(lds_addr)
                         : CLOB, "memory");
        const f32x4 cs = *(const f32x4*)rp;
        float t0 = v[1] * cs[1], t1 = v[1] * cs[0], t2 = v[3] * cs[3], t3 = v[3] * cs[2];
        asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
        float r0 = __builtin_fmaf(v[0], cs[0], -t0);
        asm volatile("" : "+v"(r0));
        float r1 = __builtin_fmaf(v[0], cs[1], t1);
        asm volatile("" : "+v"(r1));
        float r2 = __builtin_fmaf(v[2], cs[2], -t2);
        asm volatile("" : "+v"(r2));
        float r3 = __builtin_fmaf(v[2], cs[3], t3);
        asm volatile("" : "+v"(r3));
        const bool b01 = p01 != cvt2(r0, r1), b2 = __float_as_uint(y2) != __float_as_uint(r2), b3 = __float_as_uint(y3) != __float_as_uint(r3);
        if (b01 || b2 || b3 || p23 != cvt2(r2, r3)) { ++n_bad; comp |= (b01 ? 1u : 0u) | (b2 ? 4u : 0u) | (b3 ? 8u : 0u); }
    }
    if (n_bad) { atomicAdd(bad, n_bad); atomicOr(bad_comp, comp); }
}

int main() {
    float *rope, *vals, *sink;
    unsigned* cnt;
    const size_t NV = (size_t)(1 << 20) * 4 + 4096;
    CK(hipMalloc(&rope, (1 << 14) * 16 + 64));
    CK(hipMalloc(&vals, NV * 4));
    CK(hipMalloc(&sink, 4096));
    CK(hipMalloc(&cnt, 8));
    {
        float* h = (float*)malloc(NV * 4);
        uint32_t s = 99u;
        for (size_t i = 0; i < NV; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 22)); }
        CK(hipMemcpy(vals, h, NV * 4, hipMemcpyHostToDevice));
        float* r = (float*)malloc((1 << 14) * 16);
        for (int i = 0; i < (1 << 14); ++i) { const float a0 = i * 0.37f, a1 = i * 0.2775f; r[4 * i] = cosf(a0); r[4 * i + 1] = sinf(a0); r[4 * i + 2] = cosf(a1); r[4 * i + 3] = sinf(a1); }
        CK(hipMemcpy(rope, r, (1 << 14) * 16, hipMemcpyHostToDevice));
    }
    const int iters = 20000;
    for (int rep = 0; rep < 3; ++rep)
        for (int inpl = 1; inpl >= 0; --inpl)
            for (int crowd = 0; crowd < 2; ++crowd) {
                const int blocks = crowd ? 1024 : 512;                 // 2 (1 probe + 1 hammer) or 4 blocks per CU
                CK(hipMemset(cnt, 0, 8));
                if (inpl) probe<true><<<blocks, 256, 49152>>>(rope, vals, cnt, cnt + 1, iters, sink);
                else probe<false><<<blocks, 256, 49152>>>(rope, vals, cnt, cnt + 1, iters, sink);
                CK(hipGetLastError());
                CK(hipDeviceSynchronize());
                unsigned h[2];
                CK(hipMemcpy(h, cnt, 8, hipMemcpyDeviceToHost));
                printf("%-26s %d blocks per CU: %u of %lld lane-samples differ (component mask 0x%x: 1 = y0|y1, 4 = y2, 8 = y3)\n",
                       inpl ? "in-place v_pk_mul_f32" : "separate destination", blocks / 256, h[0], (long long)(blocks / 2) * 256 * iters, h[1]);
                fflush(stdout);
            }
    return 0;
}
