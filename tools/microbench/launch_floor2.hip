// What sets the per-kernel floor inside a captured decode step?  Graph of 172 dependent launches, variants:
//  0 tiny kernel, 1 block          1 tiny, 240 blocks x 256          2 240 blocks + 80 KB dynamic LDS
//  3 240 blocks, 200-byte kernarg  4 two kernels alternating          5 high-VGPR kernel (forced ~200 regs)
//  6 240 blocks + 80 KB LDS + waves_per_eu(1,1)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
struct Big { int* p; int n; int pad[46]; };
__global__ void k_tiny(int* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1; }
__global__ void k_tiny2(int* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 2; }
__global__ void k_lds(int* p, int n) { extern __shared__ int sm[]; int i = blockIdx.x * blockDim.x + threadIdx.x; sm[threadIdx.x] = i; __syncthreads(); if (i < n) p[i] += sm[threadIdx.x ^ 1] & 1; }
__global__ __attribute__((amdgpu_waves_per_eu(1, 1))) void k_lds_w1(int* p, int n) { extern __shared__ int sm[]; int i = blockIdx.x * blockDim.x + threadIdx.x; sm[threadIdx.x] = i; __syncthreads(); if (i < n) p[i] += sm[threadIdx.x ^ 1] & 1; }
__global__ void k_big(Big b) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < b.n) b.p[i] += b.pad[3]; }
__global__ __launch_bounds__(256) void k_regs(int* p, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v[160];
#pragma unroll
    for (int j = 0; j < 160; ++j) v[j] = (float)(i + j);
    if (n < 0) {   // never true at run time; keeps the registers live
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 160; ++j) v[j] = v[j] * v[(j + 1) % 160] + 1.0f;
    }
    float s = 0; 
#pragma unroll
    for (int j = 0; j < 160; ++j) s += v[j];
    if (i < n) p[i] += (int)s & 1;
}
int main() {
    int* d; (void)hipMalloc(&d, 1 << 22); (void)hipMemset(d, 0, 1 << 22);
    hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    (void)hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    (void)hipFuncSetAttribute((const void*)k_lds_w1, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    Big bg{}; bg.p = d; bg.n = 61440;
    for (int variant = 0; variant < 7; ++variant) {
        auto launch = [&](int idx) {
            switch (variant) {
                case 0: hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, d, 64); break;
                case 1: hipLaunchKernelGGL(k_tiny, dim3(240), dim3(256), 0, st, d, 61440); break;
                case 2: hipLaunchKernelGGL(k_lds, dim3(240), dim3(256), 80 * 1024, st, d, 61440); break;
                case 3: hipLaunchKernelGGL(k_big, dim3(240), dim3(256), 0, st, bg); break;
                case 4: if (idx & 1) hipLaunchKernelGGL(k_tiny, dim3(240), dim3(256), 0, st, d, 61440);
                        else hipLaunchKernelGGL(k_tiny2, dim3(240), dim3(256), 0, st, d, 61440); break;
                case 5: hipLaunchKernelGGL(k_regs, dim3(240), dim3(256), 0, st, d, 61440); break;
                case 6: hipLaunchKernelGGL(k_lds_w1, dim3(240), dim3(256), 80 * 1024, st, d, 61440); break;
            }
        };
        hipGraph_t g; hipGraphExec_t ge;
        (void)hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < 172; ++i) launch(i);
        (void)hipStreamEndCapture(st, &g);
        (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        for (int i = 0; i < 5; ++i) (void)hipGraphLaunch(ge, st);
        (void)hipStreamSynchronize(st);
        auto t0 = std::chrono::high_resolution_clock::now();
        for (int i = 0; i < 50; ++i) (void)hipGraphLaunch(ge, st);
        (void)hipStreamSynchronize(st);
        auto t1 = std::chrono::high_resolution_clock::now();
        printf("variant %d graph(172 nodes): %.2f us per kernel\n", variant,
               std::chrono::duration<double, std::micro>(t1 - t0).count() / 50 / 172);
    }
    return 0;
}
