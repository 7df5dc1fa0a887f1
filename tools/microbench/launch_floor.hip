// Kernel-boundary floor on this box: N dependent trivial kernels on one stream, eager vs hipGraph replay.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void tiny(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void tiny_big(int* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1; }
int main() {
    int* d; hipMalloc(&d, 1 << 20); hipMemset(d, 0, 1 << 20);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    const int N = 2000;
    for (int variant = 0; variant < 2; ++variant) {
        auto launch = [&]() { if (variant == 0) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, d);
                              else hipLaunchKernelGGL(tiny_big, dim3(256), dim3(256), 0, st, d, 65536); };
        for (int i = 0; i < 100; ++i) launch();
        hipStreamSynchronize(st);
        auto t0 = std::chrono::high_resolution_clock::now();
        for (int i = 0; i < N; ++i) launch();
        hipStreamSynchronize(st);
        auto t1 = std::chrono::high_resolution_clock::now();
        printf("variant %d eager: %.2f us per kernel\n", variant, std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < 172; ++i) launch();
        hipStreamEndCapture(st, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        for (int i = 0; i < 5; ++i) hipGraphLaunch(ge, st);
        hipStreamSynchronize(st);
        t0 = std::chrono::high_resolution_clock::now();
        for (int i = 0; i < 50; ++i) hipGraphLaunch(ge, st);
        hipStreamSynchronize(st);
        t1 = std::chrono::high_resolution_clock::now();
        printf("variant %d graph(172 nodes): %.2f us per kernel, %.1f us per replay\n", variant,
               std::chrono::duration<double, std::micro>(t1 - t0).count() / 50 / 172, std::chrono::duration<double, std::micro>(t1 - t0).count() / 50);
    }
    return 0;
}
