// GPU box: does vmcnt retire in ISSUE ORDER across the two kinds of vector loads this code base mixes -- loads into VGPRs and
// LDS-DMA loads (global_load_lds_*)?  Every counted wait (s_waitcnt vmcnt(N), N > 0) in a kernel that has both kinds in flight
// assumes it.  Each case issues an OLD load from a cold line (a fresh 32 KiB-strided line of a 2 GiB buffer: HBM + TLB miss) and a
// YOUNG load from a hot line (read just before: TCP / L2 hit), waits vmcnt(1) -- "at most one outstanding", which under in-order
// retirement means the OLD one has landed -- and looks at the OLD load's destination, pre-filled with a sentinel:
//   case 0: old = VGPR load, young = VGPR load   (control: the ISA's in-order guarantee)
//   case 1: old = VGPR load, young = LDS-DMA
//   case 2: old = LDS-DMA,  young = VGPR load
//   case 3: old = LDS-DMA,  young = LDS-DMA
// Output: per case, how many of the (lane, wave, trial) samples still saw the sentinel after the wait.
// build: hipcc --offload-arch=gfx950 -O2 -o vmcnt_order vmcnt_order.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define SENT 0xDEADBEEFu
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int CASE>
__global__ __launch_bounds__(64) void probe(const uint32_t* cold, const uint32_t* hot, unsigned* stale, unsigned* wrong, int trials, size_t stride_u32) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[2][64];
    const int lane = threadIdx.x;
    const uint32_t m0_old = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&lds[0][0];
    const uint32_t m0_young = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&lds[1][0];
    const uint32_t l_old = m0_old + lane * 4;
    unsigned n_stale = 0, n_wrong = 0;
    for (int t = 0; t < trials; ++t) {
        const uint32_t* cp = cold + ((size_t)blockIdx.x * trials + t) * stride_u32 + lane;     // a fresh line per (wave, trial)
        const uint32_t* hp = hot + lane;
        uint32_t warm = *hp;
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(warm)::"memory");
        lds[0][lane] = SENT;
        lds[1][lane] = SENT;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        uint32_t r_old = SENT, r_young = SENT ^ warm ^ warm, seen = 0;
        if (CASE == 0) {
            asm volatile("global_load_dword %0, %3, off\n\tglobal_load_dword %1, %4, off\n\ts_waitcnt vmcnt(1)\n\tv_mov_b32 %2, %0\n\ts_waitcnt vmcnt(0)"
                         : "+v"(r_old), "+v"(r_young), "=&v"(seen) : "v"(cp), "v"(hp) : "memory");
        } else if (CASE == 1) {
            asm volatile("global_load_dword %0, %3, off\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dword %4, off\n\ts_waitcnt vmcnt(1)\n\tv_mov_b32 %2, %0\n\ts_waitcnt vmcnt(0)"
                         : "+v"(r_old), "+v"(r_young), "=&v"(seen) : "v"(cp), "v"(hp), "s"(m0_young) : "memory", "m0");
        } else if (CASE == 2) {
            asm volatile("s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dword %3, off\n\tglobal_load_dword %1, %4, off\n\ts_waitcnt vmcnt(1)\n\tds_read_b32 %2, %6\n\ts_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(0)"
                         : "+v"(r_old), "+v"(r_young), "=&v"(seen) : "v"(cp), "v"(hp), "s"(m0_old), "v"(l_old) : "memory", "m0");
        } else {
            asm volatile("s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dword %3, off\n\ts_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dword %4, off\n\ts_waitcnt vmcnt(1)\n\tds_read_b32 %2, %6\n\ts_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(0)"
                         : "+v"(r_old), "+v"(r_young), "=&v"(seen) : "v"(cp), "v"(hp), "s"(m0_old), "v"(l_old), "s"(m0_young) : "memory", "m0");
        }
        const uint32_t expect = 0x10000000u + (uint32_t)((((size_t)blockIdx.x * trials + t) * stride_u32 + lane) & 0x0fffffffu);
        if (seen == SENT) ++n_stale;
        else if (seen != expect) ++n_wrong;
        // after the full wait the destination must hold the data (sanity of the probe itself)
        uint32_t fin = (CASE <= 1) ? r_old : lds[0][lane];
        if (fin != expect) ++n_wrong;
    }
    if (n_stale) atomicAdd(stale, n_stale);
    if (n_wrong) atomicAdd(wrong, n_wrong);
}

__global__ void fill(uint32_t* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0x10000000u + (uint32_t)(i & 0x0fffffffu);
}

int main() {
    const int blocks = 1024, trials = 64;
    const size_t stride_u32 = 8192;                                  // 32 KiB between the cold lines
    const size_t n = (size_t)blocks * trials * stride_u32;           // 2 GiB
    uint32_t *cold, *hot;
    unsigned* cnt;
    CK(hipMalloc(&cold, n * 4));
    CK(hipMalloc(&hot, 4096));
    CK(hipMalloc(&cnt, 64));
    fill<<<4096, 256>>>(cold, n);
    fill<<<1, 256>>>(hot, 1024);
    CK(hipDeviceSynchronize());
    const char* names[4] = {"old VGPR load, young VGPR load (control)", "old VGPR load, young LDS-DMA", "old LDS-DMA, young VGPR load", "old LDS-DMA, young LDS-DMA"};
    for (int rep = 0; rep < 2; ++rep)
        for (int c = 0; c < 4; ++c) {
            CK(hipMemset(cnt, 0, 64));
            // evict: sweep the cold buffer's first 1 GiB again so that the probed lines are not cache / MALL resident
            fill<<<4096, 256>>>(cold, n);
            CK(hipDeviceSynchronize());
            if (c == 0) probe<0><<<blocks, 64>>>(cold, hot, cnt, cnt + 1, trials, stride_u32);
            else if (c == 1) probe<1><<<blocks, 64>>>(cold, hot, cnt, cnt + 1, trials, stride_u32);
            else if (c == 2) probe<2><<<blocks, 64>>>(cold, hot, cnt, cnt + 1, trials, stride_u32);
            else probe<3><<<blocks, 64>>>(cold, hot, cnt, cnt + 1, trials, stride_u32);
            CK(hipDeviceSynchronize());
            unsigned h[2];
            CK(hipMemcpy(h, cnt, 8, hipMemcpyDeviceToHost));
            printf("case %d (%s): %u of %d samples saw the sentinel after s_waitcnt vmcnt(1); probe self-check failures %u\n", c, names[c], h[0],
                   blocks * trials * 64, h[1]);
        }
    return 0;
}
