// Does touching the NEXT kernel's weights from inside the CURRENT kernel shorten a chain of small weight-streaming kernels?
// ("a parallel hipGraph branch that touches layer i + 1's weights into the 256 MB Infinity Cache while layer i runs
// -- microbench it first".)  The decode step is a chain  LN (5 us, almost no bytes) -> GEMM (3.3 / 9.8 / 13.1 MB of weights, 5-8 us);
// a GEMM's weights do not depend on anything, only its activations do.  This tool replays that chain from a hipGraph:
//   small(i)  : 64 blocks with a dependent load -> reduce -> store (the LayerNorm's shape), optionally + 192 blocks that touch one dword
//               per 128-byte (or 64-byte) line of buffer i
//   reader(i) : 256 blocks x 4 waves, every wave issues all its 16-byte-per-lane loads of buffer i at once (the decode GEMM's weight
//               stream), xor-reduces, stores one word per block; depends on small(i)'s output
// over NBUF rotating buffers (NBUF x size > 256 MB, so a buffer is out of every cache when its turn comes again).
// Modes: 0 no prefetch; 1 prefetch by blocks on the SAME XCD that will read the lines (blockIdx % 8 round-robin: L2 + Infinity Cache);
//        2 prefetch by blocks of the NEXT XCD (Infinity Cache only); 3 = 1 at a 64-byte stride; 4 the reader re-reads ONE buffer (cache-hot
//        bound); 5 the READER of buffer i touches buffer i + 1 (GEMM -> GEMM seam, e.g. fc -> fc2) after issuing its own loads, same XCD.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench/bin/weight_prefetch tools/microbench/weight_prefetch.hip ; run under timeout 120.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef unsigned v4u __attribute__((ext_vector_type(4)));

// touch lines of `buf` that reader blocks of XCD ((xcd + shift) & 7) will read; `nq` prefetch blocks per XCD cooperate, this is number q
__device__ __forceinline__ unsigned touch(const char* buf, size_t chunk, int xcd, int shift, int q, int nq, int stride) {
    const int tx = (xcd + shift) & 7;
    const size_t lines_per_chunk = chunk / stride;
    const size_t total = 32 * lines_per_chunk;                     // 32 reader chunks live on one XCD
    unsigned acc = 0;
    const size_t step = (size_t)nq * blockDim.x;
    for (size_t l0 = (size_t)q * blockDim.x + threadIdx.x; l0 < total; l0 += 4 * step) {      // four independent loads in flight per thread
        unsigned v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t l = l0 + k * step;
            const size_t lc = l < total ? l : l0;
            const size_t c = lc / lines_per_chunk, r = lc - c * lines_per_chunk;
            v[k] = *(const unsigned*)(buf + (c * 8 + tx) * chunk + r * stride);
        }
        acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
    }
    return acc;
}

__global__ __launch_bounds__(256) void small_kernel(const unsigned* dep, unsigned* out, const char* pf_buf, size_t chunk, int mode, int npf) {
    if (blockIdx.x < 64) {                                         // the LayerNorm-shaped part: dependent load, reduction, store
        unsigned v = dep[(blockIdx.x * 256 + threadIdx.x) & 255];
        for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
        __shared__ unsigned s[4];
        if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) out[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
        return;
    }
    const int g = blockIdx.x, p = g - 64;
    const unsigned a = touch(pf_buf, chunk, g & 7, mode == 2 ? 1 : 0, p >> 3, npf >> 3, mode == 3 ? 64 : 128);
    if (a == 0x9e3779b9u) out[64 + p] = a;                         // keeps the loads alive
}

template <int NL>
__global__ __launch_bounds__(256) void reader_kernel(const char* buf, size_t chunk, const unsigned* dep, unsigned* out, const char* next_buf, int mode) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const v4u* src = (const v4u*)(buf + (size_t)blockIdx.x * chunk) + (size_t)w * NL * 64 + lane;
    v4u f[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) f[i] = src[(size_t)i * 64];
    unsigned pf = 0;
    if (mode == 5) pf = touch(next_buf, chunk, blockIdx.x & 7, 0, blockIdx.x >> 3, 32, 128);
    unsigned d = dep[lane & 63];                                   // activations: depend on the previous kernel
    v4u x = f[0];
#pragma unroll
    for (int i = 1; i < NL; ++i) x ^= f[i];
    unsigned r = x[0] ^ x[1] ^ x[2] ^ x[3] ^ d ^ pf;
    for (int o = 32; o; o >>= 1) r ^= __shfl_xor(r, o);
    if (lane == 0) out[blockIdx.x * 4 + w] = r;
}

static float run(int mode, size_t size, int nbuf, std::vector<char*>& bufs, unsigned* s_out, unsigned* r_out, int npairs, bool no_small, bool no_reader) {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    const size_t chunk = size / 256;
    const int nl = (int)(chunk / 4096);                            // 1 KiB wave loads per wave
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < npairs; ++i) {
        const char* b = bufs[mode == 4 ? 0 : i % nbuf];
        const char* bn = bufs[(i + 1) % nbuf];
        const bool pf = mode == 1 || mode == 2 || mode == 3;
        if (!no_small) hipLaunchKernelGGL(small_kernel, dim3(pf ? 256 : 64), dim3(256), 0, st, r_out, s_out, b, chunk, mode, 192);
        if (!no_reader) {
            switch (nl) {
                case 3: hipLaunchKernelGGL(reader_kernel<3>, dim3(256), dim3(256), 0, st, b, chunk, s_out, r_out, bn, mode); break;
                case 9: hipLaunchKernelGGL(reader_kernel<9>, dim3(256), dim3(256), 0, st, b, chunk, s_out, r_out, bn, mode); break;
                case 12: hipLaunchKernelGGL(reader_kernel<12>, dim3(256), dim3(256), 0, st, b, chunk, s_out, r_out, bn, mode); break;
                default: printf("unsupported chunk %zu\n", chunk); exit(1);
            }
        }
    }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, st));
        CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    CK(hipStreamDestroy(st));
    return best * 1000.f / npairs;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int npairs = 192;
    unsigned *s_out, *r_out;
    CK(hipMalloc(&s_out, 4096));
    CK(hipMalloc(&r_out, 8192));
    CK(hipMemset(s_out, 0, 4096));
    CK(hipMemset(r_out, 0, 8192));
    const size_t sizes[3] = {3 * 4096 * 256, 9 * 4096 * 256, 12 * 4096 * 256};      // 3.1 / 9.4 / 12.6 MB: proj, qkv, fc / fc2
    for (int si = 0; si < 3; ++si) {
        const size_t size = sizes[si];
        const int nbuf = (int)((640u << 20) / size) + 1;
        std::vector<char*> bufs(nbuf);
        for (int i = 0; i < nbuf; ++i) { CK(hipMalloc(&bufs[i], size)); CK(hipMemset(bufs[i], i + 1, size)); }
        CK(hipDeviceSynchronize());
        const float t_small = run(0, size, nbuf, bufs, s_out, r_out, npairs, false, true);
        const float t_read = run(0, size, nbuf, bufs, s_out, r_out, npairs, true, false);
        printf("size %.1f MB x %d buffers: small alone %.2f us, reader alone %.2f us (%.2f TB/s)\n", size / 1e6, nbuf, t_small, t_read, size / t_read * 1e-6);
        const char* names[6] = {"no prefetch", "prefetch same XCD (128 B)", "prefetch next XCD (128 B)", "prefetch same XCD (64 B)", "one hot buffer", "reader touches next buffer"};
        for (int mode = 0; mode < 6; ++mode) {
            const float t = run(mode, size, nbuf, bufs, s_out, r_out, npairs, false, false);
            printf("  mode %d %-30s: %.2f us per (small + reader) pair\n", mode, names[mode], t);
        }
        // reader -> reader chain with and without the in-reader touch (GEMM -> GEMM seam)
        const float t_rr = run(0, size, nbuf, bufs, s_out, r_out, npairs, true, false);
        const float t_rr5 = run(5, size, nbuf, bufs, s_out, r_out, npairs, true, false);
        printf("  reader chain: %.2f us per reader; with in-reader touch of the next buffer %.2f us\n", t_rr, t_rr5);
        for (int i = 0; i < nbuf; ++i) CK(hipFree(bufs[i]));
    }
    return 0;
}
