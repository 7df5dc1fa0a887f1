// Where sample_kernel's ~40 us per launch go (first result, profiles/r03x/sample_stamps.log: top-k threshold 17 us with either
// selection, row staging 6.5, thread-0 tail 5.5, rank sort 4.2): builds the product kernel with
// -DITTS_SAMPLE_STAMPS (thread 0 of every block stores s_memrealtime, 100 MHz, at eight phase boundaries) and prints the mean phase lengths.
//   phases: 0 entry -> 1 row staged in LDS (penalty, temperature) -> 2 top-k threshold -> 3 survivors collected + rank-sorted -> 4 token chosen
//           (top-p, renormalise, multinomial on thread 0) -> 5 token / seen / finished written -> 6 next-step embedding written -> 7 advanced
//           (threadfence + ticket)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I indextts_amd/csrc -mllvm -amdgpu-mfma-vgpr-form=1 -DITTS_SAMPLE_STAMPS \
//         tools/microbench/sample_stamps.hip -o tools/microbench/bin/sample_stamps && tools/microbench/bin/sample_stamps 1 && ... 64
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../indextts_amd/csrc/common.h"
void itts_set_error(const char* fmt, ...) { (void)fmt; }
#include "../../indextts_amd/csrc/gpt_kernels.hip"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 1, V = 8194, D = 1280, max_new = 64, reps = 40;
    float *logits, *mel_emb, *mel_pos, *x_next;
    unsigned char *seen, *finished;
    long long* tokens;
    int* state;
    unsigned long long* stamps;
    CK(hipMalloc(&logits, (size_t)B * V * 4)); CK(hipMalloc(&mel_emb, (size_t)V * D * 4)); CK(hipMalloc(&mel_pos, (size_t)(max_new + 8) * D * 4));
    CK(hipMalloc(&x_next, (size_t)B * D * 4)); CK(hipMalloc(&seen, (size_t)B * V)); CK(hipMalloc(&finished, B));
    CK(hipMalloc(&tokens, (size_t)B * max_new * 8)); CK(hipMalloc(&state, 64)); CK(hipMalloc(&stamps, (size_t)B * 8 * 8));
    CK(hipMemset(seen, 0, (size_t)B * V)); CK(hipMemset(finished, 0, B)); CK(hipMemset(state, 0, 64)); CK(hipMemset(mel_emb, 0, (size_t)V * D * 4));
    CK(hipMemset(mel_pos, 0, (size_t)(max_new + 8) * D * 4));
    std::vector<float> h((size_t)B * V);
    unsigned x = 777;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (((x >> 8) & 0xffff) / 65536.0f - 0.5f) * 12.0f; }
    CK(hipMemcpy(logits, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    SampleArgs s{};
    s.logits = logits; s.seen = seen; s.finished = finished; s.tokens = tokens; s.step_ptr = state; s.seed = 12345; s.B = B; s.V = V; s.max_new = max_new;
    s.do_sample = 1; s.top_k = 30; s.min_keep = 1; s.top_p = 0.8f; s.temperature = 0.8f; s.rep_penalty = 10.0f; s.typical_mass = 0.f;
    s.stop_token = 8193; s.mel_emb = mel_emb; s.mel_pos = mel_pos; s.x_next = x_next; s.D = D; s.pos_offset = 2; s.n_mel_pos = max_new + 8;
    s.adv_state = state; s.uniforms_stride = B; s.stamps = stamps;
    std::vector<unsigned long long> hs((size_t)B * 8);
    double acc[8] = {0};
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float total_ms = 0;
    for (int r = 0; r < reps + 2; ++r) {
        CK(hipMemset(state, 0, 64));                       // step 0 every time: the token row stays inside `tokens`
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        if (launch_sample(s, 0) != ITTS_OK) { printf("launch failed\n"); return 1; }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
        if (r < 2) continue;                               // warm-up launches
        total_ms += ms;
        for (int b = 0; b < B; ++b)
            for (int k = 1; k < 8; ++k) acc[k] += (double)(hs[(size_t)b * 8 + k] - hs[(size_t)b * 8 + k - 1]) * 0.01;      // 100 MHz ticks -> us
    }
    const char* names[8] = {"", "row -> LDS (penalty, temperature)", "top-k threshold", "survivors + rank sort", "top-p / renormalise / pick (thread 0)",
                            "token, seen, finished", "next-step embedding", "advance (fence + ticket)"};
    printf("sample_kernel B=%d V=%d: %.1f us per launch by HIP events; mean phase lengths over blocks (us):\n", B, V, total_ms * 1000 / reps);
    double sum = 0;
    for (int k = 1; k < 8; ++k) { printf("  %-44s %7.2f\n", names[k], acc[k] / reps / B); sum += acc[k] / reps / B; }
    printf("  %-44s %7.2f\n", "sum (entry -> exit of a block)", sum);
    // the chosen tokens (step 0 of every row): every selection variant must print the same fingerprint
    std::vector<long long> ht((size_t)B * max_new);
    CK(hipMemcpy(ht.data(), tokens, ht.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long fp = 0;
    for (int b = 0; b < B; ++b) fp = fp * 1000003ull + (unsigned long long)ht[(size_t)b * max_new];
    printf("  tokens fingerprint %016llx (row 0 -> %lld)\n", fp, ht[0]);
    return 0;
}
