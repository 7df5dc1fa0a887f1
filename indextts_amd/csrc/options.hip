// Run-time options of the engine: ONE documented table behind itts_set_option / itts_get_option (include/indextts_hip.h).
// The library reads no environment variable.  Every option selects between kernels (or kernel geometries) that are held to
// the same results by tests -- most of them bitwise -- and exists so that those tests, and the measurement tools, can switch
// paths inside one process; the defaults are the measured-best paths and are what the shipped host code runs.
#include <atomic>
#include <stdlib.h>
#include <string.h>

#include "../../include/indextts_hip.h"
#include "common.h"

namespace {
struct OptDef {
    const char* name;
    int def, lo, hi;
    const char* doc;
    const char* allowed = nullptr;        // discrete legal values inside [lo, hi] ("6,8"); nullptr: every integer of the range
};
// order == the OPT_* enum of common.h
const OptDef kDefs[OPT_COUNT] = {
    {"decode_fuse_ln", 1, 0, 2, "GPT decode: LayerNorm (+ split-K reduce, bias, residual) inside the consuming GEMM -- 1: steps of 1-8 rows (measured faster), 2: up to 16 rows (measured slower above 8, kept for the A/B record), 0: separate ln_kernel launches; bitwise equal"},
    {"decode_gemm", 1, 0, 1, "bf16 decode GEMMs on the LDS-DMA slab kernel (0: register-path gemm_kernel; bitwise equal)"},
    {"decode_rot", 1, 0, 1, "per-block rotation of the slab DMA issue order in the decode GEMM (same bytes, same LDS image)"},
    {"decode_wnt", 0, 0, 1, "non-temporal policy on the decode GEMM's weight stream"},
    {"decode_nt", 0, 0, 4, "force the n-tiles per block of the 64-row decode GEMM (0: smallest that fits one round of blocks)", "0,1,2,4"},
    {"prefill_gemm", 1, 0, 1, "bf16 prefill GEMMs on the LDS-DMA tile kernels (0: register-path gemm_kernel; bitwise equal)"},
    {"tile256", -1, -1, 2, "bf16 tile GEMM: -1 pick by shape, 0 always 128x128, 1 always 256x256, 2 always 256x128 (bitwise equal)"},
    {"f32_tile", 1, 0, 1, "f32 GEMMs with plain epilogues on the f32-MFMA tile kernel (0: register-path kernel; bitwise equal)"},
    {"x3_products", 6, 6, 8, "plane products per f32 product of the fp32x3 GEMMs and attention: 6 (hh, hm, mh, hl, lh, mm; drops m*l and l*m, 2^-24 |ab| each: measured error vs f64 <= the native f32-MFMA kernels' on every benchmarked shape) or 8 (every term down to 2^-24 |ab|; two blocks per CU like the default since the packed-RoPE fix -- option x3_pin = 1 restores round 4's one-block pin as a diagnostic)", "6,8"},
    {"x3_sched", 1, 0, 1, "fp32x3 GEMM: interleave the operand split with the MFMAs (0: split as a burst; bitwise equal)"},
    {"x3_attn", 1, 0, 1, "fp32x3 s2mel: attention products on bf16 planes too (K / V^T written as three planes by the wqkv epilogue, flash_attn_x3_kernel); 0: the f32-MFMA flash kernel"},
    {"sample_radix", -1, -1, 1, "top-k threshold: -1 per-kernel default (radix select in sample_kernel, ballot bisection in the beam kernels), 0 bisection, 1 radix select (identical ids)"},
    {"gpt_compact", 1, 0, 1, "row compaction of ragged decode batches (0 disables it for every handle; identical ids)"},
    {"attn_waves", 0, 0, 16, "waves per block of the KV-cache attention kernel: 0 pick by shape, else 4 / 8 / 16 (the 16 canonical key streams are mapped onto them; bitwise equal)", "0,4,8,16"},
    {"s2mel_fused", 1, 0, 2, "s2mel: fused GEMM epilogues, sampled when a handle is created -- 1: fused, 0: separate element-wise kernels (2: round 4's spelling of 'the bf16 wqkv epilogue fused too'; the same as 1 since the RoPE fix of round 5)"},
    {"fa_qs", 0, 0, 4, "bf16 flash attention: query sub-tiles per wave (0: pick by shape)", "0,1,2,4"},
    {"f32_attn_scalar", 0, 0, 1, "f32 s2mel attention on the one-wave-per-query reference kernel (the A/B path of the f32 flash kernel)"},
    {"fa32_qs", 2, 1, 2, "f32 flash attention: query sub-tiles per wave"},
    {"aa_act", 2, 0, 2, "anti-aliased activation kernel variant (2: swizzled LDS tiles)"},
    {"conv_bm", 0, 0, 128, "force the co-tile height of conv_mfma_kernel (0: pick by channel count)", "0,32,64,96,128"},
    {"h3_kernel", 1, 0, 1, "f16x3 vocoder conv: 1 window kernel, 0 two-stage kernel"},
    {"decode_ln_nt", 2, 2, 4, "LayerNorm-fused decode GEMM at 5-16 rows (weights on waves 0-3, LayerNorm on waves 4-7): n-tiles per block, 2 or 4 (bitwise equal)", "2,4"},
    {"x3_aplanes", 0, 0, 1, "fp32x3 s2mel: the adaptive-RMSNorm outputs as three bf16 planes in fragment order, wqkv / w1|w3 GEMMs without an operand split (bitwise equal)"},
    {"x3_pin", 0, 0, 1, "fp32x3 GEMM: 1 pins the variants that are not the default one (8 products, burst split) to one block per CU -- round 4's workaround for their run-to-run differences, kept as a diagnostic; not needed since the RoPE fix of round 5 (DESIGN.md section 9)"},
    {"prefill_attn", -1, -1, 1, "attention of S > 1 passes (prefill, latent pass): -1 causal MFMA kernel in the bf16 mode, canonical-stream kernel in the f32 parity mode; 0 canonical-stream kernel (one block per query) always; 1 MFMA kernel in both precisions"},
    {"voc_act_planes", 1, 0, 1, "vocoder, bf16x3 conv mode: the anti-aliased activation in front of an x3 conv writes the conv's three operand planes itself (aa_act_planes_kernel); 0: f32 activation + split pass (bit-identical planes)"},
    {"x3_waves", 8, 4, 8, "fp32x3 GEMM: waves per 128 x 128 block -- 8 (4 x 2, wave tile 32 x 64, weights through LDS: four waves per SIMD at two blocks per CU; 64-utterance solve 314.7 vs 323.0 ms per Euler step, profiles/r06k) or 4 (2 x 2, wave tile 64 x 64, weights in registers: round 5's kernel); bitwise equal", "4,8"},
};
std::atomic<int> g_val[OPT_COUNT];
std::atomic<unsigned> g_epoch{1};
std::atomic<bool> g_init{false};

void ensure_init() {
    if (g_init.load(std::memory_order_acquire)) return;
    static std::atomic_flag once = ATOMIC_FLAG_INIT;
    if (!once.test_and_set()) {
        for (int i = 0; i < OPT_COUNT; ++i) g_val[i].store(kDefs[i].def, std::memory_order_relaxed);
        g_init.store(true, std::memory_order_release);
    } else {
        while (!g_init.load(std::memory_order_acquire)) {}
    }
}
int find(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < OPT_COUNT; ++i)
        if (strcmp(kDefs[i].name, name) == 0) return i;
    return -1;
}
}  // namespace

int itts_opt(int id) {
    ensure_init();
    return g_val[id].load(std::memory_order_relaxed);
}
unsigned itts_opt_epoch() { return g_epoch.load(std::memory_order_relaxed); }

extern "C" int itts_set_option(const char* name, int value) {
    ensure_init();
    const int i = find(name);
    if (i < 0) { itts_set_error("set_option: unknown option '%s'", name ? name : "(null)"); return ITTS_ERR_ARG; }
    if (value < kDefs[i].lo || value > kDefs[i].hi) {
        itts_set_error("set_option: %s = %d outside [%d, %d]", name, value, kDefs[i].lo, kDefs[i].hi);
        return ITTS_ERR_ARG;
    }
    if (kDefs[i].allowed) {                                         // discrete set: the launchers size LDS / pick instantiations per listed value only
        bool ok = false;
        for (const char* p = kDefs[i].allowed; *p;) {
            char* end;
            const long v = strtol(p, &end, 10);
            ok |= v == value;
            p = *end == ',' ? end + 1 : end;
        }
        if (!ok) { itts_set_error("set_option: %s = %d is not one of {%s}", name, value, kDefs[i].allowed); return ITTS_ERR_ARG; }
    }
    if (g_val[i].exchange(value) != value) g_epoch.fetch_add(1);      // cached decode graphs bake kernel choices in: a new epoch retires them
    return ITTS_OK;
}
extern "C" int itts_get_option(const char* name, int* value) {
    ensure_init();
    const int i = find(name);
    if (i < 0 || !value) { itts_set_error("get_option: unknown option '%s'", name ? name : "(null)"); return ITTS_ERR_ARG; }
    *value = g_val[i].load();
    return ITTS_OK;
}
extern "C" int itts_reset_options(void) {
    ensure_init();
    bool changed = false;
    for (int i = 0; i < OPT_COUNT; ++i) changed |= g_val[i].exchange(kDefs[i].def) != kDefs[i].def;
    if (changed) g_epoch.fetch_add(1);
    return ITTS_OK;
}
extern "C" int itts_option_count(void) { return OPT_COUNT; }
extern "C" const char* itts_option_name(int index) { return (index >= 0 && index < OPT_COUNT) ? kDefs[index].name : nullptr; }
extern "C" const char* itts_option_doc(int index) { return (index >= 0 && index < OPT_COUNT) ? kDefs[index].doc : nullptr; }
extern "C" int itts_option_default(int index) { return (index >= 0 && index < OPT_COUNT) ? kDefs[index].def : 0; }
