// GPU box: is `v_cvt_pk_bf16_f32 d0, a, b` safe when the NEXT instruction overwrites its source register b?  (write-after-read on a source of the
// gfx950 packed conversion.)  hipcc emits exactly that pair in the bf16 RoPE epilogue of the tile GEMM --
//     v_pk_fma_f32 v[6:7], ...            ; y1 lands in v7
//     v_cvt_pk_bf16_f32 v6, v38, v7       ; (y0, y1) -> v6
//     v_cvt_pk_bf16_f32 v7, v30, v21      ; (y2, y3) -> v7     <- overwrites the first conversion's source
// -- and that epilogue's Q / K outputs are not bit-stable run to run.  Here the same three instructions run in inline asm on
// changing data and the packed results are compared with conversions done on copies (no register reuse).  A mismatch = the first conversion saw the
// second one's result in its source.
// build: hipcc --offload-arch=gfx950 -O2 -o cvt_war_hazard cvt_war_hazard.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int MODE>      // 0: cvt, cvt (WAR on src1)   1: a VALU op producing the source first, then cvt, cvt   2: with an s_nop between the conversions (control)
__global__ __launch_bounds__(256) void probe(const float* in, unsigned* bad, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    float y0 = in[t & 4095], y1 = in[(t + 1111) & 4095], y2 = in[(t + 2222) & 4095], y3 = in[(t + 3333) & 4095];
    unsigned n_bad = 0;
    for (int it = 0; it < iters; ++it) {
        y0 = y0 * 1.0009765625f + 0.37f; y1 = y1 * 0.99951171875f - 0.11f; y2 = y2 * 1.001953125f + 0.05f; y3 = y3 * 0.998046875f - 0.21f;
        if (fabsf(y0) > 1e4f) y0 *= 1e-4f;
        if (fabsf(y2) > 1e4f) y2 *= 1e-4f;
        // reference: conversions with no register reuse
        unsigned r0, r1;
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=&v"(r0) : "v"(y0), "v"(y1));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=&v"(r1) : "v"(y2), "v"(y3));
        unsigned d0;
        float b = y1;                                   // the register that is read by the first conversion and overwritten by the second
        if (MODE == 0) {
            asm volatile("v_cvt_pk_bf16_f32 %0, %2, %1\n\tv_cvt_pk_bf16_f32 %1, %3, %4" : "=&v"(d0), "+v"(b) : "v"(y0), "v"(y2), "v"(y3));
        } else if (MODE == 1) {
            // a VALU op writes the source right before the conversions, as the epilogue's v_pk_fma does
            asm volatile("v_fma_f32 %1, %1, 1.0, 0\n\tv_cvt_pk_bf16_f32 %0, %2, %1\n\tv_cvt_pk_bf16_f32 %1, %3, %4" : "=&v"(d0), "+v"(b) : "v"(y0), "v"(y2), "v"(y3));
        } else {
            asm volatile("v_cvt_pk_bf16_f32 %0, %2, %1\n\ts_nop 4\n\tv_cvt_pk_bf16_f32 %1, %3, %4" : "=&v"(d0), "+v"(b) : "v"(y0), "v"(y2), "v"(y3));
        }
        if (d0 != r0 || __builtin_bit_cast(unsigned, b) != r1) ++n_bad;
    }
    if (n_bad) atomicAdd(bad, n_bad);
}

template <int MODE>
static int run(const float* in, unsigned* cnt, int blocks, const char* what) {
    const int iters = 4096;
    CK(hipMemset(cnt, 0, 4));
    probe<MODE><<<blocks, 256>>>(in, cnt, iters);
    CK(hipDeviceSynchronize());
    unsigned h;
    CK(hipMemcpy(&h, cnt, 4, hipMemcpyDeviceToHost));
    printf("mode %d (%s), %4d blocks: %u of %lld lane-samples differ\n", MODE, what, blocks, h, (long long)blocks * 256 * iters);
    return 0;
}

int main() {
    float* in;
    unsigned* cnt;
    CK(hipMalloc(&in, 4096 * 4));
    CK(hipMalloc(&cnt, 4));
    float host[4096];
    unsigned s = 777u;
    for (int i = 0; i < 4096; ++i) { s = s * 1664525u + 1013904223u; host[i] = ((int)(s >> 8) % 20001 - 10000) * 1e-3f; }
    CK(hipMemcpy(in, host, sizeof(host), hipMemcpyHostToDevice));
    for (int blocks : {256, 2048}) {
        if (run<0>(in, cnt, blocks, "cvt; cvt overwriting the first one's source")) return 1;
        if (run<1>(in, cnt, blocks, "v_fma -> cvt -> cvt")) return 1;
        if (run<2>(in, cnt, blocks, "control: s_nop 4 between the conversions")) return 1;
    }
    return 0;
}
