// GPU box: may a VALU instruction overwrite a VGPR that the MFMA issued just before it reads as SrcA / SrcB?  hipcc's hazard recognizer knows no
// write-after-read hazard on the A / B operands of an MFMA (only on SrcC), and the fp32x3 GEMM's interleaved operand split produces exactly that
// pattern under register pressure (8-product variant: `v_mfma ... v[200:203], ...; v_mfma ... v[200:203], ...; v_add_f32 v200, ...`, gpt_kernels.hip
// disassembly) -- the variants that are not bit-stable when two blocks share a CU.  If the matrix pipe queues an MFMA behind another wave's and reads
// its operands late, the overwrite would reach it.
// Here, in one asm statement on fixed registers: NM back-to-back v_mfma_f32_16x16x32_bf16 reading A = v[8:11], B = v[12:15], then an overwrite of v8
// (or v12) by the instruction under test, GAP wait states later; the accumulators are compared with the same sequence whose overwrite goes to an
// unrelated register.  Any difference = the MFMA saw the overwritten operand.  Run alone (one wave per SIMD) and with 2 / 4 / 8 waves per SIMD that
// keep the matrix pipe contended.
// build: hipcc --offload-arch=gfx950 -O2 -o bin/mfma_war_hazard mfma_war_hazard.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#define MFMA1 "v_mfma_f32_16x16x32_bf16 v[20:23], v[8:11], v[12:15], v[20:23]\n\t"
#define MFMA2 "v_mfma_f32_16x16x32_bf16 v[24:27], v[8:11], v[12:15], v[24:27]\n\t" MFMA1
#define MFMA4 "v_mfma_f32_16x16x32_bf16 v[32:35], v[8:11], v[12:15], v[32:35]\n\tv_mfma_f32_16x16x32_bf16 v[28:31], v[8:11], v[12:15], v[28:31]\n\t" MFMA2

// SEQ = the MFMAs, OVER = the overwriting instruction (writes v8.. or v12..; its sources are v16..v19 = garbage), SAFE = the same instruction writing v36..
#define PROBE_KERNEL(NAME, SEQ, OVER, SAFE)                                                                                                      \
    __global__ __launch_bounds__(256) void NAME(const uint32_t* in, unsigned* bad, int iters) {                                                  \
        unsigned n_bad = 0;                                                                                                                       \
        uint32_t s = in[(blockIdx.x * 256 + threadIdx.x) & 4095];                                                                                 \
        for (int it = 0; it < iters; ++it) {                                                                                                      \
            uint32_t q[12];                                                                                                                       \
            for (int i = 0; i < 12; ++i) { s = s * 1664525u + 1013904223u; q[i] = (s & 0x3f7f3f7fu) | 0x3c003c00u; }                              \
            float r[4], e[4];                                                                                                                     \
            for (int pass = 0; pass < 2; ++pass) {                                                                                                \
                float o0, o1, o2, o3;                                                                                                             \
                if (pass == 0)                                                                                                                    \
                    asm volatile("v_mov_b32 v8, %4\n\tv_mov_b32 v9, %5\n\tv_mov_b32 v10, %6\n\tv_mov_b32 v11, %7\n\t"                             \
                                 "v_mov_b32 v12, %8\n\tv_mov_b32 v13, %9\n\tv_mov_b32 v14, %10\n\tv_mov_b32 v15, %11\n\t"                         \
                                 "v_mov_b32 v16, %12\n\tv_mov_b32 v17, %13\n\tv_mov_b32 v18, %14\n\tv_mov_b32 v19, %15\n\t"                       \
                                 "v_mov_b32 v20, 0\n\tv_mov_b32 v21, 0\n\tv_mov_b32 v22, 0\n\tv_mov_b32 v23, 0\n\t"                               \
                                 "v_mov_b32 v24, 0\n\tv_mov_b32 v25, 0\n\tv_mov_b32 v26, 0\n\tv_mov_b32 v27, 0\n\t"                               \
                                 "v_mov_b32 v28, 0\n\tv_mov_b32 v29, 0\n\tv_mov_b32 v30, 0\n\tv_mov_b32 v31, 0\n\t"                               \
                                 "v_mov_b32 v32, 0\n\tv_mov_b32 v33, 0\n\tv_mov_b32 v34, 0\n\tv_mov_b32 v35, 0\n\t"                               \
                                 "s_nop 7\n\t" SEQ OVER "\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"                                               \
                                 "v_mov_b32 %0, v20\n\tv_mov_b32 %1, v21\n\tv_mov_b32 %2, v22\n\tv_mov_b32 %3, v23"                               \
                                 : "=v"(o0), "=v"(o1), "=v"(o2), "=v"(o3)                                                                         \
                                 : "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]), "v"(q[5]), "v"(q[6]), "v"(q[7]), "v"(q[8]), "v"(q[9]),  \
                                   "v"(q[10]), "v"(q[11])                                                                                         \
                                 : "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23",  \
                                   "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39"); \
                else                                                                                                                              \
                    asm volatile("v_mov_b32 v8, %4\n\tv_mov_b32 v9, %5\n\tv_mov_b32 v10, %6\n\tv_mov_b32 v11, %7\n\t"                             \
                                 "v_mov_b32 v12, %8\n\tv_mov_b32 v13, %9\n\tv_mov_b32 v14, %10\n\tv_mov_b32 v15, %11\n\t"                         \
                                 "v_mov_b32 v16, %12\n\tv_mov_b32 v17, %13\n\tv_mov_b32 v18, %14\n\tv_mov_b32 v19, %15\n\t"                       \
                                 "v_mov_b32 v20, 0\n\tv_mov_b32 v21, 0\n\tv_mov_b32 v22, 0\n\tv_mov_b32 v23, 0\n\t"                               \
                                 "v_mov_b32 v24, 0\n\tv_mov_b32 v25, 0\n\tv_mov_b32 v26, 0\n\tv_mov_b32 v27, 0\n\t"                               \
                                 "v_mov_b32 v28, 0\n\tv_mov_b32 v29, 0\n\tv_mov_b32 v30, 0\n\tv_mov_b32 v31, 0\n\t"                               \
                                 "v_mov_b32 v32, 0\n\tv_mov_b32 v33, 0\n\tv_mov_b32 v34, 0\n\tv_mov_b32 v35, 0\n\t"                               \
                                 "s_nop 7\n\t" SEQ SAFE "\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"                                               \
                                 "v_mov_b32 %0, v20\n\tv_mov_b32 %1, v21\n\tv_mov_b32 %2, v22\n\tv_mov_b32 %3, v23"                               \
                                 : "=v"(o0), "=v"(o1), "=v"(o2), "=v"(o3)                                                                         \
                                 : "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]), "v"(q[5]), "v"(q[6]), "v"(q[7]), "v"(q[8]), "v"(q[9]),  \
                                   "v"(q[10]), "v"(q[11])                                                                                         \
                                 : "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23",  \
                                   "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39"); \
                if (pass == 0) { r[0] = o0; r[1] = o1; r[2] = o2; r[3] = o3; } else { e[0] = o0; e[1] = o1; e[2] = o2; e[3] = o3; }               \
            }                                                                                                                                     \
            if (r[0] != e[0] || r[1] != e[1] || r[2] != e[2] || r[3] != e[3]) ++n_bad;                                                            \
        }                                                                                                                                         \
        if (n_bad) atomicAdd(bad, n_bad);                                                                                                         \
    }

// the probed MFMA (v[20:23]) is the LAST of SEQ: the overwrite follows it immediately (or GAP wait states later)
PROBE_KERNEL(k_mov_a_1, MFMA1, "v_mov_b32 v8, v16", "v_mov_b32 v36, v16")
PROBE_KERNEL(k_mov_a_2, MFMA2, "v_mov_b32 v8, v16", "v_mov_b32 v36, v16")
PROBE_KERNEL(k_mov_a_4, MFMA4, "v_mov_b32 v8, v16", "v_mov_b32 v36, v16")
PROBE_KERNEL(k_mov_a3_4, MFMA4, "v_mov_b32 v11, v16", "v_mov_b32 v36, v16")
PROBE_KERNEL(k_mov_b_2, MFMA2, "v_mov_b32 v12, v16", "v_mov_b32 v36, v16")
PROBE_KERNEL(k_mov_b_4, MFMA4, "v_mov_b32 v12, v16", "v_mov_b32 v36, v16")
PROBE_KERNEL(k_add_a_2, MFMA2, "v_add_f32 v8, v16, v17", "v_add_f32 v36, v16, v17")
PROBE_KERNEL(k_add_a_4, MFMA4, "v_add_f32 v8, v16, v17", "v_add_f32 v36, v16, v17")
PROBE_KERNEL(k_pk_a_2, MFMA2, "v_pk_add_f32 v[8:9], v[16:17], v[18:19]", "v_pk_add_f32 v[36:37], v[16:17], v[18:19]")
PROBE_KERNEL(k_pk_a_4, MFMA4, "v_pk_add_f32 v[8:9], v[16:17], v[18:19]", "v_pk_add_f32 v[36:37], v[16:17], v[18:19]")
PROBE_KERNEL(k_cvt_a_2, MFMA2, "v_cvt_pk_bf16_f32 v8, v16, v17", "v_cvt_pk_bf16_f32 v36, v16, v17")
PROBE_KERNEL(k_cvt_a_4, MFMA4, "v_cvt_pk_bf16_f32 v8, v16, v17", "v_cvt_pk_bf16_f32 v36, v16, v17")
PROBE_KERNEL(k_mov_a_4_gap1, MFMA4, "s_nop 0\n\tv_mov_b32 v8, v16", "s_nop 0\n\tv_mov_b32 v36, v16")
// control: the overwrite BEFORE the last MFMA with no wait state (a read-after-write the hardware must interlock or the recognizer pads) is not probed here

typedef void (*kern_t)(const uint32_t*, unsigned*, int);
static int run(kern_t k, const char* name, const uint32_t* in, unsigned* cnt, int blocks, int iters, const char* what) {
    CK(hipMemset(cnt, 0, 4));
    k<<<blocks, 256>>>(in, cnt, iters);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    unsigned h;
    CK(hipMemcpy(&h, cnt, 4, hipMemcpyDeviceToHost));
    printf("%-40s %-28s %u of %lld lane-samples differ\n", name, what, h, (long long)blocks * 256 * iters);
    return 0;
}

int main() {
    uint32_t* in;
    unsigned* cnt;
    CK(hipMalloc(&in, 4096 * 4));
    CK(hipMalloc(&cnt, 4));
    uint32_t host[4096];
    uint32_t s = 12345u;
    for (int i = 0; i < 4096; ++i) { s = s * 1664525u + 1013904223u; host[i] = s; }
    CK(hipMemcpy(in, host, sizeof(host), hipMemcpyHostToDevice));
    const int iters = 2048;
    struct { kern_t k; const char* name; } ks[] = {
        {k_mov_a_1, "1 MFMA, v_mov over A[0]"}, {k_mov_a_2, "2 MFMAs, v_mov over A[0]"}, {k_mov_a_4, "4 MFMAs, v_mov over A[0]"},
        {k_mov_a3_4, "4 MFMAs, v_mov over A[3]"}, {k_mov_b_2, "2 MFMAs, v_mov over B[0]"}, {k_mov_b_4, "4 MFMAs, v_mov over B[0]"},
        {k_add_a_2, "2 MFMAs, v_add_f32 over A[0]"}, {k_add_a_4, "4 MFMAs, v_add_f32 over A[0]"},
        {k_pk_a_2, "2 MFMAs, v_pk_add_f32 over A[0:1]"}, {k_pk_a_4, "4 MFMAs, v_pk_add_f32 over A[0:1]"},
        {k_cvt_a_2, "2 MFMAs, v_cvt_pk_bf16_f32 over A[0]"}, {k_cvt_a_4, "4 MFMAs, v_cvt_pk_bf16_f32 over A[0]"},
        {k_mov_a_4_gap1, "4 MFMAs, s_nop 0, v_mov over A[0]"},
    };
    const int crowd[] = {256, 512, 1024, 2048};
    const char* cname[] = {"1 wave per SIMD", "2 waves per SIMD", "4 waves per SIMD", "8 waves per SIMD"};
    for (auto& k : ks)
        for (int c = 0; c < 4; ++c)
            if (run(k.k, k.name, in, cnt, crowd[c], iters, cname[c])) return 1;
    return 0;
}
