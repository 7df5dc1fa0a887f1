// Grid-barrier cost on MI355X, to settle whether a persistent per-layer decode kernel can beat the 1.75 us in-graph kernel boundary
// (the "GPT decode" lever).  NOT run yet: written at the end of round 2 when the GPU budget was spent; build with
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/bin/grid_barrier tools/microbench/grid_barrier.hip
// and run under `timeout 60` on the box.  Every spin loop is bounded (SPIN_CAP polls): if the blocks are not co-resident, or a
// barrier variant is wrong, the kernel gives up, raises the `aborted` flag and returns -- it cannot hang the GPU.
//
// Each iteration: every block does a token piece of dependent work (reads the value the "next" block wrote in the previous
// iteration -- a real cross-block dependency, so a broken barrier shows up as a wrong checksum), then a barrier:
//   mode 0  flat        one device-scope counter, every block's thread 0 polls it
//   mode 1  flag        one counter; the last arriver publishes the epoch in a separate flag line that the others poll
//   mode 2  xcd-tree    blocks are grouped by blockIdx % 8 (workgroups are dealt round-robin to the eight XCDs): a counter per group,
//                       the last arriver of each group bumps the top counter, the last of those publishes one epoch flag PER GROUP
//                       (each group polls its own 128-byte line: eight lines shared by 32 blocks each instead of one shared by 256)
// Reported: microseconds per (work + barrier) iteration for grids of 64 / 128 / 256 blocks of 256 threads, next to the same work as
// a chain of dependent kernels replayed from a hipGraph (the structure the decode step has today).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <vector>

#define SPIN_CAP 2000000
#define LINE 32                                   // unsigneds per 128-byte line

struct Sync {
    unsigned* top;        // [LINE]      top counter (monotonic)
    unsigned* grp;        // [8][LINE]   per-group counters (monotonic)
    unsigned* flag;       // [8][LINE]   epoch flags (flag[0] is the single flag of mode 1)
    unsigned* aborted;    // [1]
};

__device__ __forceinline__ unsigned ld_acq(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_rel(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned add_rel(unsigned* p) { return __hip_atomic_fetch_add(p, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT); }

// returns false when the poll budget ran out (somebody else may have aborted as well)
__device__ __forceinline__ bool spin_until(const unsigned* p, unsigned want, unsigned* aborted) {
    for (int i = 0; i < SPIN_CAP; ++i) {
        if ((int)(ld_acq(p) - want) >= 0) return true;
        if ((i & 1023) == 1023 && ld_acq(aborted)) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    st_rel(aborted, 1u);
    return false;
}

template <int MODE>
__device__ __forceinline__ bool grid_barrier(const Sync& s, unsigned epoch /* 1, 2, ... */, int nblk) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __threadfence();
        if (MODE == 0) {
            add_rel(s.top);
            ok = spin_until(s.top, epoch * (unsigned)nblk, s.aborted);
        } else if (MODE == 1) {
            const unsigned old = add_rel(s.top);
            if (old + 1 == epoch * (unsigned)nblk) st_rel(s.flag, epoch);
            else ok = spin_until(s.flag, epoch, s.aborted);
        } else {
            const int g = blockIdx.x & 7;
            const int ng = (nblk >> 3) + ((int)(blockIdx.x & 7) < (nblk & 7) ? 1 : 0);      // blocks in this group
            const int groups = nblk < 8 ? nblk : 8;
            const unsigned old = add_rel(s.grp + g * LINE);
            bool publish = false;
            if (old + 1 == epoch * (unsigned)ng) {
                const unsigned t = add_rel(s.top);
                publish = t + 1 == epoch * (unsigned)groups;
            }
            if (publish) {
                for (int q = 0; q < groups; ++q) st_rel(s.flag + q * LINE, epoch);
            } else {
                ok = spin_until(s.flag + g * LINE, epoch, s.aborted);
            }
        }
        __threadfence();
    }
    __syncthreads();
    return ok;
}

// work: block b reads slot (b + 1) % nblk written in the previous iteration and writes its own slot: a chain across blocks
template <int MODE>
__global__ __launch_bounds__(256) void persistent_kernel(Sync s, unsigned* slots, int iters, unsigned* checksum) {
    const int nblk = gridDim.x, b = blockIdx.x;
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        if (threadIdx.x == 0) {                                   // double-buffered: iteration `it` reads parity it & 1, writes the other one
            const unsigned v = ld_acq(slots + ((it & 1) * 256 + (b + 1) % nblk) * LINE);
            acc += v;
            st_rel(slots + (((it + 1) & 1) * 256 + b) * LINE, v + 1u);
        }
        if (!grid_barrier<MODE>(s, (unsigned)(it + 1), nblk)) break;
    }
    if (threadIdx.x == 0) atomicAdd(checksum, acc);
}

// the same chain with a kernel boundary as the barrier: node i reads `in`, writes `out` (the host alternates the two buffers)
__global__ __launch_bounds__(256) void step_kernel(const unsigned* in, unsigned* out, unsigned* acc_out) {
    const int nblk = gridDim.x, b = blockIdx.x;
    if (threadIdx.x == 0) {
        const unsigned v = in[((b + 1) % nblk) * LINE];
        acc_out[b * LINE] += v;
        out[b * LINE] = v + 1u;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

template <int MODE>
static int run_mode(int nblk, int iters, hipStream_t st, Sync s, unsigned* slots, unsigned* checksum) {
    CK(hipMemsetAsync(s.top, 0, LINE * 4, st));
    CK(hipMemsetAsync(s.grp, 0, 8 * LINE * 4, st));
    CK(hipMemsetAsync(s.flag, 0, 8 * LINE * 4, st));
    CK(hipMemsetAsync(s.aborted, 0, 4, st));
    CK(hipMemsetAsync(slots, 0, 2 * 256 * LINE * 4, st));
    CK(hipMemsetAsync(checksum, 0, 4, st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL(persistent_kernel<MODE>, dim3(nblk), dim3(256), 0, st, s, slots, iters, checksum);
    CK(hipGetLastError());
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned ab = 0, ck = 0;
    CK(hipMemcpy(&ab, s.aborted, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&ck, checksum, 4, hipMemcpyDeviceToHost));
    // with a correct barrier every block reads, at iteration it, the value it (all slots advance together): sum = nblk * iters (iters - 1) / 2
    const unsigned long long want = (unsigned long long)nblk * iters * (iters - 1) / 2;
    printf("mode %d, %3d blocks: %.3f us per work + barrier  (%s, checksum %s)\n", MODE, nblk, ms * 1e3 / iters, ab ? "ABORTED: poll budget ran out" : "ok",
           (unsigned)(want & 0xffffffffu) == ck ? "ok" : "MISMATCH");
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return 0;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    Sync s;
    unsigned *slots, *checksum, *acc;
    CK(hipMalloc(&s.top, LINE * 4));
    CK(hipMalloc(&s.grp, 8 * LINE * 4));
    CK(hipMalloc(&s.flag, 8 * LINE * 4));
    CK(hipMalloc(&s.aborted, 4));
    CK(hipMalloc(&slots, 2 * 256 * LINE * 4));
    CK(hipMalloc(&acc, 256 * LINE * 4));
    CK(hipMalloc(&checksum, 4));
    for (int nblk : {64, 128, 256}) {
        if (run_mode<0>(nblk, iters, st, s, slots, checksum)) return 1;
        if (run_mode<1>(nblk, iters, st, s, slots, checksum)) return 1;
        if (run_mode<2>(nblk, iters, st, s, slots, checksum)) return 1;
        // the same work as dependent kernels in a hipGraph (172 nodes per replay, like one decode step)
        CK(hipMemsetAsync(slots, 0, 2 * 256 * LINE * 4, st));
        CK(hipMemsetAsync(acc, 0, 256 * LINE * 4, st));
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 172; ++i)
            hipLaunchKernelGGL(step_kernel, dim3(nblk), dim3(256), 0, st, slots + (i & 1) * 256 * LINE, slots + ((i + 1) & 1) * 256 * LINE, acc);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        auto t0 = std::chrono::high_resolution_clock::now();
        const int reps = 30;
        for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        auto t1 = std::chrono::high_resolution_clock::now();
        printf("graph,  %3d blocks: %.3f us per work + kernel boundary\n", nblk, std::chrono::duration<double, std::micro>(t1 - t0).count() / reps / 172);
        hipGraphExecDestroy(ge);
        hipGraphDestroy(g);
    }
    return 0;
}
