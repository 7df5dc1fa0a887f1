// Shared device/host helpers for the gfx950 kernels (wave = 64 lanes everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define ITTS_OK 0
#define ITTS_ERR_ARG 1
#define ITTS_ERR_HIP 2
#define ITTS_ERR_STATE 3
#define ITTS_ERR_NOMEM 4

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            itts_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return ITTS_ERR_HIP;                                                              \
        }                                                                                     \
    } while (0)

void itts_set_error(const char* fmt, ...);

// Run-time options (options.hip; public face: itts_set_option / itts_get_option in include/indextts_hip.h).  Launchers read the
// current value at launch time; itts_opt_epoch() changes whenever any value does (cached hipGraphs are keyed on it).
enum { OPT_DECODE_FUSE_LN, OPT_DECODE_GEMM, OPT_DECODE_ROT, OPT_DECODE_WNT, OPT_DECODE_NT, OPT_PREFILL_GEMM, OPT_TILE256, OPT_F32_TILE,
       OPT_X3_PRODUCTS, OPT_X3_SCHED, OPT_X3_ATTN, OPT_SAMPLE_RADIX, OPT_GPT_COMPACT, OPT_ATTN_WAVES, OPT_S2MEL_FUSED, OPT_FA_QS,
       OPT_F32_ATTN_SCALAR, OPT_FA32_QS, OPT_AA_ACT, OPT_CONV_BM, OPT_H3_KERNEL, OPT_DECODE_LN_NT, OPT_X3_APLANES, OPT_X3_PIN, OPT_PREFILL_ATTN, OPT_VOC_ACT_PLANES, OPT_X3_WAVES, OPT_COUNT };
int itts_opt(int id);
unsigned itts_opt_epoch();

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Every object that owns device memory (weights, streams, events) records the device that was current when it was created;
// each C-ABI entry point taking such a handle runs under this guard, so a caller whose current device differs (a process
// driving several GPUs) neither allocates on nor launches to the wrong one.
struct ItDevGuard {
    int prev = -1;
    bool switched = false;
    explicit ItDevGuard(int dev) {
        if (dev < 0) return;
        if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipGetLastError(); }
        if (prev != dev && hipSetDevice(dev) == hipSuccess) switched = true;
    }
    ~ItDevGuard() { if (switched && prev >= 0) (void)hipSetDevice(prev); }
    ItDevGuard(const ItDevGuard&) = delete;
    ItDevGuard& operator=(const ItDevGuard&) = delete;
};
static inline int itts_current_device() {
    int d = -1;
    if (hipGetDevice(&d) != hipSuccess) { d = -1; (void)hipGetLastError(); }
    return d;
}
// device that owns a device pointer, -1 if unknown (host pointer / query unsupported)
static inline int itts_ptr_device(const void* p) {
    hipPointerAttribute_t at;
    if (!p || hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return at.type == hipMemoryTypeDevice ? at.device : -1;
}

// hipFuncSetAttribute (dynamic-LDS limits) is per DEVICE: a "done once" flag kept per device ordinal, so a process driving several GPUs
// raises the limit on each of them (a process-wide static left the second GPU's kernels at the 64 KiB default).
template <class T>
struct ItPerDevice {
    T v[32] = {};
    T& cur() {
        const int d = itts_current_device();
        return v[(d >= 0 && d < 32) ? d : 0];
    }
};

#ifdef __HIPCC__
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float bf16_to_f32(u16 h) { return __uint_as_float(((uint32_t)h) << 16); }
// round-to-nearest-even, NaN-preserving (same rule as torch .bfloat16())
__device__ __forceinline__ u16 f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u16)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (u16)(u >> 16);
}
#endif

static inline u16 host_f32_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u16)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (u16)(u >> 16);
}
