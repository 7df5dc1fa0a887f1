// C-ABI entry points for the s2mel flow-matching decoder: DiT estimator + CFG Euler solver (include/indextts_hip.h).
//
// Reference call replaced: `self.s2mel.models['cfm'].inference(cat_condition, x_lens, ref_mel, style, None, 25,
// inference_cfg_rate=0.7)` (indextts/infer_v2_5.py:841-845) = BASECFM.solve_euler around DiT.forward
// (indextts/s2mel/modules/flow_matching.py:57-115, diffusion_transformer.py:186-257).
//
// What stays on the host side (vectors, once per call): the timestep embeddings and everything that depends on t only -- the
// AdaLN (weight | bias) vectors of every norm, the WaveNet conditioning, the final-layer modulation -- and the step-invariant
// part of cond_x_merge_linear (prompt, content and style columns).  They arrive as `mods` / `const_in`.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "../../include/indextts_hip.h"
#include "gpt_kernels.h"
#include "s2mel_kernels.h"

// Element-wise stages live in the GEMM epilogues (no f32 round trip of the wide intermediates) and the paired weights are packed
// tile-interleaved for them -- in both precisions: the f32 mode (what the reference computes: autocast is off around the s2mel
// stage, infer_v2_5.py:827-828) runs the same fused structure on the f32-MFMA instantiations of the tile kernel.
// Option s2mel_fused = 0 (read when the handle is created: the weight packing depends on it) forces the separate element-wise
// kernels -- the A/B switch of tests/test_gpu_s2mel.py.
static bool s2_fused(int precision, bool opt_fused) {
    if (precision == PREC_F32X3) return true;                   // the f32x3 GEMM kernel exists with the fused epilogues only
    return (precision == PREC_BF16 || precision == PREC_F32) && opt_fused;
}

struct S2Layer {
    void *w_qkv = 0, *w_o = 0, *w_13 = 0, *w_2 = 0, *w_skip_a = 0, *w_skip_b = 0;
    float *g_attn = 0, *g_ffn = 0, *b_skip = 0;
};
struct S2Wn {
    void *w_in = 0, *w_rs = 0;
    float *b_in = 0, *b_rs = 0;
};

struct itts_s2mel {
    itts_s2mel_config cfg;
    int I = 0, Kx = 0;                          // SwiGLU width, padded K of the mel-channel GEMMs
    std::map<std::string, std::pair<std::vector<float>, std::vector<int64_t>>> host;   // raw tensors until finalize
    std::vector<S2Layer> layers;
    std::vector<S2Wn> wn;
    float* g_norm = 0;
    void *w_x = 0, *w_sl_a = 0, *w_sl_b = 0, *w_c1 = 0, *w_rp = 0, *w_fl = 0, *w_c2 = 0;
    float *b_sl = 0, *b_c1 = 0, *b_rp = 0, *b_fl = 0, *b_c2 = 0;
    std::vector<void*> owned;
    bool finalized = false;
    bool opt_fused = true;                      // option s2mel_fused as it stood at itts_s2mel_create (non-zero)
    bool opt_fused_qkv = true;                  // ... and whether the wqkv GEMM carries RoPE + the Q / K / V^T scatter in its epilogue (see s2_estimator)
    int device = -1;
    // optional HIP-event timing per kernel class (itts_s2mel_set_profiling): events are recorded on the launch stream around
    // every launch of the last solve / estimator call
    bool profiling = false;
    std::vector<hipEvent_t> ev_pool;
    struct Rec { int cls; double flops; };
    std::vector<Rec> recs;
    hipStream_t prof_stream = nullptr;
    // Dead-row elimination in itts_s2mel_solve (itts_s2mel_set_tail): the Euler step never reads the estimator's output at prompt frames
    // (flow_matching.py:107 zeroes them), and everything after the last attention is row-wise except the WaveNet's +-halo frames of
    // context -- so that part runs on the TAIL of every sequence only (frames >= prompt_len - halo).  Device tables of the tail layout:
    SeqTab tail{};                         // tok_seq / tok_t (relative to the cut) / seq_start / seq_T / seq_len of the tail rows
    const int* tail_src = nullptr;         // [tail.n_tok] full-layout row of each tail row
    const int* tail_base = nullptr;        // [n_seq] tail row of frame t of sequence s = tail_base[s] + t
    bool tail_set = false;
    // Diagnostics (itts_s2mel_set_trace): a checksum of every stage's output buffer of the following estimator / solve calls, in launch order, into a
    // caller-provided device array -- two runs on the same inputs are compared entry by entry to find the first stage that is not bit-stable.
    unsigned long long* trace = nullptr;
    int trace_cap = 0, trace_n = 0;
    int trace_wanted = 0;                  // entries the last call asked for (> trace_cap: the trace stopped at capacity)
    std::vector<const char*> trace_labels;
    // ... and, for the stages whose label starts with capture_prefix, a copy of the stage's output bytes (itts_s2mel_set_capture): the stage
    // checksums say WHICH stage differed between two runs, the captured images say which elements and how.
    char* capture = nullptr;
    size_t capture_bytes = 0, capture_used = 0;
    std::string capture_prefix;
    std::vector<long long> capture_off;    // per trace entry: byte offset of its image in the capture buffer, or -1
    std::vector<size_t> capture_len;
};

// record the checksum of a stage output (no-op unless a trace buffer is set)
static int s2_trace(itts_s2mel* h, hipStream_t st, const char* label, const void* p, size_t bytes) {
    if (!h->trace) return ITTS_OK;
    ++h->trace_wanted;
    if (h->trace_n >= h->trace_cap) return ITTS_OK;
    if ((int)h->trace_labels.size() <= h->trace_n) { h->trace_labels.push_back(label); h->capture_off.push_back(-1); h->capture_len.push_back(0); }
    else { h->trace_labels[h->trace_n] = label; h->capture_off[h->trace_n] = -1; h->capture_len[h->trace_n] = 0; }
    if (h->capture && !h->capture_prefix.empty() && strncmp(label, h->capture_prefix.c_str(), h->capture_prefix.size()) == 0) {
        const size_t at = (h->capture_used + 255) & ~(size_t)255;
        if (at + bytes <= h->capture_bytes) {
            HIP_TRY(hipMemcpyAsync(h->capture + at, p, bytes, hipMemcpyDeviceToDevice, st));
            h->capture_off[h->trace_n] = (long long)at;
            h->capture_len[h->trace_n] = bytes;
            h->capture_used = at + bytes;
        }
    }
    return launch_trace_hash(p, bytes, h->trace + h->trace_n++, st);
}
// a call starts: entry 0 again, the checksum words cleared (the hash kernel ADDS into its word)
static int s2_trace_begin(itts_s2mel* h, hipStream_t st) {
    h->trace_n = 0;
    h->trace_wanted = 0;
    h->capture_used = 0;
    if (h->trace && h->trace_cap > 0) HIP_TRY(hipMemsetAsync(h->trace, 0, (size_t)h->trace_cap * 8, st));
    return ITTS_OK;
}

enum { S2_GEMM = 0, S2_ATTN = 1, S2_OTHER = 2, S2_CLASSES = 3 };

struct S2Prof {
    itts_s2mel* h;
    hipStream_t st;
    bool on;
    size_t idx;
    S2Prof(itts_s2mel* h_, hipStream_t st_, int cls, double flops) : h(h_), st(st_), on(h_->profiling), idx(0) {
        if (!on) return;
        idx = h->recs.size();
        h->recs.push_back({cls, flops});
        while (h->ev_pool.size() < 2 * (idx + 1)) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) { on = false; h->recs.pop_back(); return; }
            h->ev_pool.push_back(e);
        }
        (void)hipEventRecord(h->ev_pool[2 * idx], st);
    }
    ~S2Prof() { if (on) (void)hipEventRecord(h->ev_pool[2 * idx + 1], st); }
};

static int s2_upload(itts_s2mel* h, const void* host, size_t bytes, void** dst) {
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, bytes));
    h->owned.push_back(d);
    HIP_TRY(hipMemcpy(d, host, bytes, hipMemcpyHostToDevice));
    *dst = d;
    return ITTS_OK;
}

static int s2_intermediate(int H) {             // gpt_fast/model.py:62-65
    int n = (int)(2 * (4 * (long long)H) / 3);
    return n % 256 == 0 ? n : n + 256 - (n % 256);
}

extern "C" int itts_s2mel_create(const itts_s2mel_config* cfg, itts_s2mel** out) {
    if (!cfg || !out) { itts_set_error("s2mel_create: null"); return ITTS_ERR_ARG; }
    const itts_s2mel_config& c = *cfg;
    if (c.hidden_dim != c.num_heads * 64 || c.hidden_dim % 64 || c.depth < 1 || c.in_channels < 1 || c.in_channels % 4 || c.wavenet_hidden % 64 ||
        c.wavenet_layers < 1 || c.wavenet_kernel < 1 || (c.wavenet_kernel & 1) == 0 || c.wavenet_dilation_rate < 1 ||
        (c.precision != PREC_F32 && c.precision != PREC_BF16 && c.precision != PREC_F32X3)) {
        itts_set_error("s2mel_create: unsupported config (hidden=%d heads=%d: head_dim must be 64; wavenet=%d x %d k=%d; prec=%d)", c.hidden_dim,
                       c.num_heads, c.wavenet_hidden, c.wavenet_layers, c.wavenet_kernel, c.precision);
        return ITTS_ERR_ARG;
    }
    itts_s2mel* h = new itts_s2mel();
    h->cfg = c;
    h->opt_fused = itts_opt(OPT_S2MEL_FUSED) != 0;
    h->opt_fused_qkv = true;
    h->I = s2_intermediate(c.hidden_dim);
    h->Kx = (c.in_channels + 63) / 64 * 64;
    h->layers.resize(c.depth);
    h->wn.resize(c.wavenet_layers);
    h->device = itts_current_device();
    *out = h;
    return ITTS_OK;
}

extern "C" int itts_s2mel_device(const itts_s2mel* h) { return h ? h->device : -1; }

extern "C" void itts_s2mel_destroy(itts_s2mel* h) {
    if (!h) return;
    ItDevGuard dg(h->device);
    for (void* p : h->owned) (void)hipFree(p);
    for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
    delete h;
}

extern "C" int itts_s2mel_set_profiling(itts_s2mel* h, int enable) {
    if (!h) { itts_set_error("s2mel_set_profiling: null"); return ITTS_ERR_ARG; }
    h->profiling = enable != 0;
    h->recs.clear();
    return ITTS_OK;
}

// Diagnostics: dev_u64 = capacity zeroed 64-bit words on the handle's device (nullptr clears).  Every following estimator / solve call restarts
// at entry 0 and adds one checksum per stage output (itts_s2mel_trace_count entries, itts_s2mel_trace_label names them).
extern "C" int itts_s2mel_set_trace(itts_s2mel* h, void* dev_u64, int capacity) {
    if (!h || capacity < 0) { itts_set_error("s2mel_set_trace: bad args"); return ITTS_ERR_ARG; }
    h->trace = (unsigned long long*)dev_u64;
    h->trace_cap = dev_u64 ? capacity : 0;
    h->trace_n = 0;
    return ITTS_OK;
}
extern "C" int itts_s2mel_trace_count(const itts_s2mel* h) { return h ? h->trace_n : 0; }
// entries the last call WANTED to write: larger than itts_s2mel_trace_count when the trace stopped at its capacity
extern "C" int itts_s2mel_trace_wanted(const itts_s2mel* h) { return h ? h->trace_wanted : 0; }
// Diagnostics: besides its checksum, keep a COPY of every traced stage output whose label starts with `label_prefix` (packed into dev_buf in
// launch order, 256-byte aligned; stages that no longer fit are skipped).  Needs a trace (itts_s2mel_set_trace); nullptr clears.
extern "C" int itts_s2mel_set_capture(itts_s2mel* h, void* dev_buf, size_t bytes, const char* label_prefix) {
    if (!h || (dev_buf && !label_prefix)) { itts_set_error("s2mel_set_capture: bad args"); return ITTS_ERR_ARG; }
    h->capture = (char*)dev_buf;
    h->capture_bytes = dev_buf ? bytes : 0;
    h->capture_used = 0;
    h->capture_prefix = dev_buf ? label_prefix : "";
    return ITTS_OK;
}
// byte offset in the capture buffer of trace entry `index` of the last call (-1: not captured) and its length
extern "C" long long itts_s2mel_capture_offset(const itts_s2mel* h, int index, size_t* bytes) {
    if (!h || index < 0 || index >= (int)h->capture_off.size() || index >= h->trace_n) return -1;
    if (bytes) *bytes = h->capture_len[index];
    return h->capture_off[index];
}
extern "C" const char* itts_s2mel_trace_label(const itts_s2mel* h, int i) {
    return (h && i >= 0 && i < (int)h->trace_labels.size()) ? h->trace_labels[i] : nullptr;
}

// Totals of the last solve / estimator call per class (0 GEMMs on the MFMA tile kernels, 1 attention, 2 everything else):
// GPU milliseconds between the events bracketing each launch, launch count, algorithmic FLOPs.  Synchronises the stream.
extern "C" int itts_s2mel_profile_read(itts_s2mel* h, double* ms, double* launches, double* flops) {
    if (!h || !ms || !launches || !flops) { itts_set_error("s2mel_profile_read: null"); return ITTS_ERR_ARG; }
    ItDevGuard dg(h->device);
    for (int i = 0; i < S2_CLASSES; ++i) ms[i] = launches[i] = flops[i] = 0;
    if (h->recs.empty()) return ITTS_OK;
    HIP_TRY(hipStreamSynchronize(h->prof_stream));
    for (size_t i = 0; i < h->recs.size(); ++i) {
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, h->ev_pool[2 * i], h->ev_pool[2 * i + 1]));
        ms[h->recs[i].cls] += t; launches[h->recs[i].cls] += 1; flops[h->recs[i].cls] += h->recs[i].flops;
    }
    return ITTS_OK;
}

// Tensors by reference state-dict name with weight-norm folded into `.weight` (host side): see itts_s2mel_finalize for the list.
extern "C" int itts_s2mel_load_tensor(itts_s2mel* h, const char* name, const float* data, const int64_t* shape, int ndim) {
    if (!h || !name || !data || !shape || ndim < 1 || ndim > 3) { itts_set_error("s2mel_load_tensor: bad args"); return ITTS_ERR_ARG; }
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
    auto& e = h->host[name];
    e.first.assign(data, data + n);
    e.second.assign(shape, shape + ndim);
    h->finalized = false;
    return ITTS_OK;
}

namespace {
struct Fin {
    itts_s2mel* h;
    std::string missing;
    const std::vector<float>* get(const std::string& name, std::initializer_list<int64_t> shape) {
        auto it = h->host.find(name);
        if (it == h->host.end()) { missing += name + " "; return nullptr; }
        const auto& sh = it->second.second;
        size_t want = 1, have = 1;
        for (auto v : shape) want *= (size_t)v;
        for (auto v : sh) have *= (size_t)v;
        if (want != have) { missing += name + "(shape) "; return nullptr; }
        return &it->second.first;
    }
    // pack a [K][N] row-major host matrix (K padded with zero rows to Kp) and upload
    int pack_kn(const std::vector<float>& kn, int K, int Kp, int N, void** dst) {
        std::vector<float> m((size_t)Kp * N, 0.f);
        memcpy(m.data(), kn.data(), (size_t)K * N * sizeof(float));
        std::vector<char> pk(itts_packed_gemm_bytes(Kp, N, h->cfg.precision));
        int rc = itts_pack_gemm_weight(m.data(), Kp, N, 0, h->cfg.precision, pk.data());
        if (rc) return rc;
        return s2_upload(h, pk.data(), pk.size(), dst);
    }
    // nn.Linear weight w [N][ldw] -> columns [c0, c0 + K) as a [K][N] matrix
    static std::vector<float> cols_kn(const std::vector<float>& w, int N, int ldw, int c0, int K) {
        std::vector<float> kn((size_t)K * N);
        for (int n = 0; n < N; ++n)
            for (int k = 0; k < K; ++k) kn[(size_t)k * N + n] = w[(size_t)n * ldw + c0 + k];
        return kn;
    }
    // [K][N] with N = two halves -> n-tiles interleaved (tile 2j: first half's columns 16j.., tile 2j+1: second half's 16j..): the
    // layout the tile kernel's pair epilogues (EPI_SWIGLU / EPI_GATE) expect; identity with the separate element-wise kernels
    std::vector<float> pair_tiles(const std::vector<float>& kn, int K, int N) const {
        if (!s2_fused(h->cfg.precision, h->opt_fused)) return kn;
        const int half = N / 2;
        std::vector<float> o((size_t)K * N);
        for (int k = 0; k < K; ++k)
            for (int j = 0; j < half / 16; ++j)
                for (int c = 0; c < 16; ++c) {
                    o[(size_t)k * N + (2 * j) * 16 + c] = kn[(size_t)k * N + 16 * j + c];
                    o[(size_t)k * N + (2 * j + 1) * 16 + c] = kn[(size_t)k * N + half + 16 * j + c];
                }
        return o;
    }
    int vec(const std::vector<float>* v, float** dst) { return s2_upload(h, v->data(), v->size() * sizeof(float), (void**)dst); }
};
}  // namespace

extern "C" int itts_s2mel_finalize(itts_s2mel* h) {
    if (!h) { itts_set_error("s2mel_finalize: null"); return ITTS_ERR_ARG; }
    ItDevGuard dg(h->device);
    const itts_s2mel_config& c = h->cfg;
    const int H = c.hidden_dim, I = h->I, W = c.wavenet_hidden, C = c.in_channels, Kx = h->Kx, k = c.wavenet_kernel, L = c.wavenet_layers;
    Fin f{h, ""};
    int rc = ITTS_OK;
    const std::string P = "estimator.";
#define NEED(var, name, ...) const std::vector<float>* var = f.get(name, {__VA_ARGS__})
    for (int i = 0; i < c.depth && rc == ITTS_OK; ++i) {
        const std::string Lp = P + "transformer.layers." + std::to_string(i) + ".";
        S2Layer& Ly = h->layers[i];
        NEED(wqkv, Lp + "attention.wqkv.weight", 3 * H, H);
        NEED(wo, Lp + "attention.wo.weight", H, H);
        NEED(w1, Lp + "feed_forward.w1.weight", I, H);
        NEED(w3, Lp + "feed_forward.w3.weight", I, H);
        NEED(w2, Lp + "feed_forward.w2.weight", H, I);
        NEED(ga, Lp + "attention_norm.norm.weight", H);
        NEED(gf, Lp + "ffn_norm.norm.weight", H);
        const bool skip = i > c.depth / 2;
        const std::vector<float>* ws = skip ? f.get(Lp + "skip_in_linear.weight", {H, 2 * H}) : nullptr;
        const std::vector<float>* bs = skip ? f.get(Lp + "skip_in_linear.bias", {H}) : nullptr;
        if (!wqkv || !wo || !w1 || !w3 || !w2 || !ga || !gf || (skip && (!ws || !bs))) continue;
        rc = f.pack_kn(Fin::cols_kn(*wqkv, 3 * H, H, 0, H), H, H, 3 * H, &Ly.w_qkv);
        if (!rc) rc = f.pack_kn(Fin::cols_kn(*wo, H, H, 0, H), H, H, H, &Ly.w_o);
        if (!rc) {                                              // [w1 ; w3] -> one GEMM of N = 2I
            std::vector<float> w13(*w1);
            w13.insert(w13.end(), w3->begin(), w3->end());
            rc = f.pack_kn(f.pair_tiles(Fin::cols_kn(w13, 2 * I, H, 0, H), H, 2 * I), H, H, 2 * I, &Ly.w_13);
        }
        if (!rc) rc = f.pack_kn(Fin::cols_kn(*w2, H, I, 0, I), I, I, H, &Ly.w_2);
        if (!rc) rc = f.vec(ga, &Ly.g_attn);
        if (!rc) rc = f.vec(gf, &Ly.g_ffn);
        if (!rc && skip) {
            rc = f.pack_kn(Fin::cols_kn(*ws, H, 2 * H, 0, H), H, H, H, &Ly.w_skip_a);
            if (!rc) rc = f.pack_kn(Fin::cols_kn(*ws, H, 2 * H, H, H), H, H, H, &Ly.w_skip_b);
            if (!rc) rc = f.vec(bs, &Ly.b_skip);
        }
    }
    NEED(gn, P + "transformer.norm.norm.weight", H);
    const std::vector<float>* wmerge = nullptr;                    // [H][C + C + content + style]: only its first C columns are used here
    {
        auto it = h->host.find(P + "cond_x_merge_linear.weight");
        if (it == h->host.end() || it->second.second.size() != 2 || it->second.second[0] != H) f.missing += P + "cond_x_merge_linear.weight ";
        else wmerge = &it->second.first;
    }
    NEED(wsl, P + "skip_linear.weight", H, H + C);
    NEED(bsl, P + "skip_linear.bias", H);
    NEED(wc1, P + "conv1.weight", W, H);
    NEED(bc1, P + "conv1.bias", W);
    NEED(wrp, P + "res_projection.weight", W, H);
    NEED(brp, P + "res_projection.bias", W);
    NEED(wfl, P + "final_layer.linear.weight", W, W);
    NEED(bfl, P + "final_layer.linear.bias", W);
    NEED(wc2, P + "conv2.weight", C, W);
    NEED(bc2, P + "conv2.bias", C);
    if (rc == ITTS_OK && gn && wmerge && wsl && bsl && wc1 && bc1 && wrp && brp && wfl && bfl && wc2 && bc2) {
        const int ldm = (int)h->host[P + "cond_x_merge_linear.weight"].second.back();
        if (ldm < C) { itts_set_error("s2mel_finalize: cond_x_merge_linear has %d input columns, fewer than the mel channels", ldm); return ITTS_ERR_ARG; }
        rc = f.vec(gn, &h->g_norm);
        if (!rc) rc = f.pack_kn(Fin::cols_kn(*wmerge, H, ldm, 0, C), C, Kx, H, &h->w_x);          // the x columns only
        if (!rc) rc = f.pack_kn(Fin::cols_kn(*wsl, H, H + C, 0, H), H, H, H, &h->w_sl_a);
        if (!rc) rc = f.pack_kn(Fin::cols_kn(*wsl, H, H + C, H, C), C, Kx, H, &h->w_sl_b);
        if (!rc) rc = f.vec(bsl, &h->b_sl);
        if (!rc) rc = f.pack_kn(Fin::cols_kn(*wc1, W, H, 0, H), H, H, W, &h->w_c1);
        if (!rc) rc = f.vec(bc1, &h->b_c1);
        if (!rc) rc = f.pack_kn(Fin::cols_kn(*wrp, W, H, 0, H), H, H, W, &h->w_rp);
        if (!rc) rc = f.vec(brp, &h->b_rp);
        if (!rc) rc = f.pack_kn(Fin::cols_kn(*wfl, W, W, 0, W), W, W, W, &h->w_fl);
        if (!rc) rc = f.vec(bfl, &h->b_fl);
        if (!rc) rc = f.pack_kn(Fin::cols_kn(*wc2, C, W, 0, W), W, W, C, &h->w_c2);
        if (!rc) rc = f.vec(bc2, &h->b_c2);
    }
    for (int i = 0; i < L && rc == ITTS_OK; ++i) {
        const std::string a = P + "wavenet.in_layers." + std::to_string(i) + ".conv.conv.", b = P + "wavenet.res_skip_layers." + std::to_string(i) + ".conv.conv.";
        const int ro = i < L - 1 ? 2 * W : W;
        NEED(wi, a + "weight", 2 * W, W, k);
        NEED(bi, a + "bias", 2 * W);
        NEED(wr, b + "weight", ro, W, 1);
        NEED(br, b + "bias", ro);
        if (!wi || !bi || !wr || !br) continue;
        // conv weight [2W][W][k] -> GEMM matrix [K = k*W][N = 2W] with K index j*W + c (the im2col column order)
        std::vector<float> kn((size_t)k * W * 2 * W);
        for (int n = 0; n < 2 * W; ++n)
            for (int cc = 0; cc < W; ++cc)
                for (int j = 0; j < k; ++j) kn[((size_t)j * W + cc) * 2 * W + n] = (*wi)[((size_t)n * W + cc) * k + j];
        rc = f.pack_kn(f.pair_tiles(kn, k * W, 2 * W), k * W, k * W, 2 * W, &h->wn[i].w_in);
        if (!rc) rc = f.vec(bi, &h->wn[i].b_in);
        if (!rc) rc = f.pack_kn(Fin::cols_kn(*wr, ro, W, 0, W), W, W, ro, &h->wn[i].w_rs);
        if (!rc) rc = f.vec(br, &h->wn[i].b_rs);
    }
#undef NEED
    if (rc) return rc;
    if (!f.missing.empty()) { itts_set_error("s2mel_finalize: missing or mis-shaped tensors: %s", f.missing.c_str()); return ITTS_ERR_STATE; }
    h->host.clear();
    h->finalized = true;
    return ITTS_OK;
}

extern "C" int itts_s2mel_mods_per_step(const itts_s2mel* h) {
    if (!h) return 0;
    const itts_s2mel_config& c = h->cfg;
    return c.depth * 4 * c.hidden_dim + 2 * c.hidden_dim + c.wavenet_layers * 2 * c.wavenet_hidden + 2 * c.wavenet_hidden;
}

// ---- workspace -------------------------------------------------------------------------------------------------
static size_t s_a256(size_t x) { return (x + 255) & ~(size_t)255; }

struct S2Ws {
    float *X, *X2, *BIG, *WX, *OUT, *RP, *D;
    char *HB, *QA, *AO, *FC, *COL, *XA, *SK, *KC, *VC, *WXA, *ZR, *HBP;
    size_t hbp_stride;                                             // f32x3: plane stride (u16 elements) of HBP, the adaptive-norm output as three bf16 planes
    size_t sk_stride, total;
};

static S2Ws s2_carve(const itts_s2mel* h, char* base, int n_tok, int n_seq, int t_pad) {
    const itts_s2mel_config& c = h->cfg;
    const size_t N = (size_t)n_tok, H = c.hidden_dim, I = h->I, W = c.wavenet_hidden, C = c.in_channels, Kx = h->Kx;
    const size_t esz = c.precision == PREC_BF16 ? 2 : 4;
    const size_t big = std::max(std::max(3 * H, 2 * I), 2 * W);
    const size_t hw = std::max(H, W), fw = std::max(I, W);
    S2Ws w;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += s_a256(bytes); return p; };
    w.X = (float*)take(N * H * 4);
    w.X2 = (float*)take(N * H * 4);
    w.BIG = (float*)take(N * big * 4);
    w.WX = (float*)take(N * W * 4);
    w.OUT = (float*)take(N * W * 4);
    w.RP = (float*)take(N * W * 4);
    w.D = (float*)take(N * C * 4);
    w.HB = take(N * hw * esz);
    w.QA = take(N * H * esz);
    w.AO = take(N * H * esz);
    w.FC = take(N * fw * esz);
    w.COL = take(N * (size_t)c.wavenet_kernel * W * esz);
    w.XA = take(N * Kx * esz);
    w.sk_stride = s_a256(N * H * esz);
    w.SK = take(w.sk_stride * (size_t)(c.depth / 2));
    w.WXA = take(N * W * esz);                                     // act-dtype shadow of WX: the tap-mode GEMM's A operand
    w.hbp_stride = (N * H + 127) / 128 * 128;
    w.HBP = take(c.precision == PREC_F32X3 ? w.hbp_stride * 6 : 0);  // f32x3: the two per-layer adaptive-RMSNorm outputs as bf16 planes
    // K / V^T images: act dtype, or -- fp32x3 -- room for the three bf16 planes of every element (6 bytes; the f32 image of option
    // x3_attn = 0 fits in the same buffer)
    const size_t kv = (size_t)n_seq * c.num_heads * t_pad * 64 * (c.precision == PREC_F32X3 ? 6 : esz);
    w.ZR = take(256);                                              // a zero row (adjacent to K / V: cleared by the same memset)
    w.KC = take(kv);
    w.VC = take(kv);
    w.total = off + 256;
    return w;
}

extern "C" size_t itts_s2mel_workspace_bytes(const itts_s2mel* h, int n_tok, int n_seq, int t_max) {
    if (!h || n_tok <= 0 || n_seq <= 0 || t_max <= 0) return 0;
    return s2_carve(h, nullptr, n_tok, n_seq, (t_max + 63) / 64 * 64).total;
}

// ---- one estimator call ----------------------------------------------------------------------------------------
static int s2_launch_gemm(itts_s2mel* h, const GemmArgs& g, hipStream_t st) {
    S2Prof ps(h, st, S2_GEMM, 2.0 * g.M * (double)g.N * g.K);
    return launch_gemm(g, h->cfg.precision, true, st);
}

// shadow: optional bf16 copy [M][ldo] of the f32 result written by the same epilogue (fused bf16 path only) -- the operand of the
// GEMM that consumes it next, instead of a separate cast pass over the f32 matrix
static int s2_gemm(itts_s2mel* h, const void* A, int lda, const void* Wp, const float* bias, float* out, int ldo, int M, int N, int K,
                   int epi, hipStream_t st, void* shadow = nullptr) {
    GemmArgs g{};
    g.A = A; g.lda = lda; g.Wp = Wp; g.bias = bias; g.M = M; g.N = N; g.K = K; g.nsplit = 1; g.epi = epi; g.out_f32 = out; g.ldo = ldo; g.D = N;
    g.out_act2 = shadow;
    return s2_launch_gemm(h, g, st);
}

// x_src [src_rows][C] f32 (row m of the token matrix reads x_src row m % src_rows); d_out [n_tok][C]
// attn_flops: 4 * hidden * sum_s T_s * len_s of one attention call (for the profile records only)
// tail (optional): run the stages after the transformer on the tail rows only; d_out is then [tail->n_tok][C] in the tail layout
static int s2_estimator(itts_s2mel* h, const S2Ws& w, const SeqTab& tab, int t_pad, const float* x_src, int src_rows, const float* const_in,
                        const float* mods, const float* rope, float* d_out, hipStream_t st, double attn_flops = 0.0,
                        const SeqTab* tail = nullptr, const int* tail_src = nullptr) {
    S2Prof whole(h, st, S2_OTHER, 0.0);          // the call's wall span; the GEMM / attention spans inside are subtracted by the reader
    const itts_s2mel_config& c = h->cfg;
    const int H = c.hidden_dim, I = h->I, W = c.wavenet_hidden, C = c.in_channels, Kx = h->Kx, N = tab.n_tok, prec = c.precision;
    const int nh = c.num_heads;
    const bool fused = s2_fused(prec, h->opt_fused);
    const bool x3_attn = prec == PREC_F32X3 && itts_opt(OPT_X3_ATTN) != 0;      // attention products on bf16 planes (flash_attn_x3_kernel)
    const size_t kv_plane = (size_t)tab.n_seq * nh * t_pad * 64;
    // f32x3: the adaptive-RMSNorm outputs (the A operands of wqkv and w1|w3) leave the norm kernel as three bf16 planes in fragment order, so those
    // two GEMMs issue no operand split at all (option x3_aplanes = 0: f32 rows + the in-register split; bitwise the same results)
    const bool a_planes = prec == PREC_F32X3 && fused && itts_opt(OPT_X3_APLANES) != 0 && itts_opt(OPT_X3_PRODUCTS) == 6 && itts_opt(OPT_X3_SCHED) != 0 && H % 32 == 0;
    int rc;
    float *X = w.X, *X2 = w.X2;
    const size_t esz = prec == PREC_BF16 ? 2 : 4;                  // activation element size
    const size_t kvb = kv_plane * (prec == PREC_F32X3 ? (x3_attn ? 6 : 4) : esz);
#define S2_TRACE(label, ptr, bytes) do { if (h->trace && (rc = s2_trace(h, st, label, ptr, bytes))) return rc; } while (0)
    // x_in = cond_x_merge_linear([x^T | prompt | cond | style]): the x columns here, the rest (+ bias) is const_in
    if ((rc = launch_cast_pad(x_src, w.XA, N, src_rows, C, Kx, prec, st))) return rc;
    HIP_TRY(hipMemcpyAsync(X, const_in, (size_t)N * H * 4, hipMemcpyDeviceToDevice, st));
    S2_TRACE("cast_pad(x) -> XA", w.XA, (size_t)N * Kx * esz);
    if ((rc = s2_gemm(h, w.XA, Kx, h->w_x, nullptr, X, H, N, H, Kx, EPI_RESIDUAL, st))) return rc;
    S2_TRACE("x_in GEMM -> X", X, (size_t)N * H * 4);
    int n_skip = 0;
    for (int i = 0; i < c.depth; ++i) {
        const S2Layer& L = h->layers[i];
        if (i > c.depth / 2) {                                     // U-ViT: x = skip_in_linear([x | skip])
            if (!fused && (rc = launch_cast_pad(X, w.HB, N, N, H, H, prec, st))) return rc;     // fused: the previous w2 GEMM left the bf16 copy in HB
            --n_skip;
            if ((rc = s2_gemm(h, w.HB, H, L.w_skip_a, L.b_skip, X2, H, N, H, H, EPI_STORE_F32, st))) return rc;
            if ((rc = s2_gemm(h, w.SK + w.sk_stride * n_skip, H, L.w_skip_b, nullptr, X2, H, N, H, H, EPI_RESIDUAL, st))) return rc;
            S2_TRACE("skip_in GEMMs -> X", X2, (size_t)N * H * 4);
            float* tmp = X; X = X2; X2 = tmp;
        }
        if (a_planes) {
            if ((rc = launch_ada_rmsnorm_planes(X, L.g_attn, mods + (size_t)i * 4 * H, w.HBP, w.hbp_stride, N, H, c.norm_eps, st))) return rc;
            S2_TRACE("ada_rmsnorm(attn) -> HB planes", w.HBP, w.hbp_stride * 6);
        } else {
            if ((rc = launch_ada_rmsnorm(X, L.g_attn, mods + (size_t)i * 4 * H, w.HB, N, H, c.norm_eps, prec, st))) return rc;
            S2_TRACE("ada_rmsnorm(attn) -> HB", w.HB, (size_t)N * H * esz);
        }
        // (Round 4 ran the bf16 mode's wqkv GEMM with a plain store + rope_split because its fused epilogue was not bit-stable run to run; the cause was
        // the SLP-packed RoPE arithmetic of pf_store_tile -- see pf_rope4 -- and every mode carries RoPE + the Q / K / V^T scatter in the epilogue again.)
        if (fused && h->opt_fused_qkv) {                           // wqkv + RoPE + Q / K / V^T scatter in one epilogue
            GemmArgs g{};
            g.A = w.HB; g.lda = H; g.Wp = L.w_qkv; g.M = N; g.N = 3 * H; g.K = H; g.nsplit = 1; g.epi = EPI_QKV_ROPE;
            if (a_planes) { g.A = w.HBP; g.a_planes = w.hbp_stride; }
            g.out_act = w.QA; g.kcache = w.KC; g.vcache = w.VC; g.D = H; g.H = nh; g.Tmax = t_pad;
            g.tok_seq = tab.tok_seq; g.tok_t = tab.tok_t; g.rope = rope;
            g.kv_planes = x3_attn ? kv_plane : 0;
            if ((rc = s2_launch_gemm(h, g, st))) return rc;
        } else {
            if ((rc = s2_gemm(h, w.HB, H, L.w_qkv, nullptr, w.BIG, 3 * H, N, 3 * H, H, EPI_STORE_F32, st))) return rc;
            if ((rc = launch_rope_split(w.BIG, rope, w.QA, w.KC, w.VC, tab, nh, t_pad, prec, st))) return rc;
        }
        S2_TRACE("wqkv -> Q", w.QA, (size_t)N * H * esz);
        S2_TRACE("wqkv -> K", w.KC, kvb);
        S2_TRACE("wqkv -> V^T", w.VC, kvb);
        { S2Prof ps(h, st, S2_ATTN, attn_flops);
          if (x3_attn) rc = launch_s2mel_attention_x3(w.QA, w.KC, w.VC, w.AO, tab, nh, t_pad, st);
          else rc = launch_s2mel_attention(w.QA, w.KC, w.VC, w.AO, tab, nh, t_pad, prec, st);
          if (rc) return rc; }
        S2_TRACE("attention -> AO", w.AO, (size_t)N * H * esz);
        if ((rc = s2_gemm(h, w.AO, H, L.w_o, nullptr, X, H, N, H, H, EPI_RESIDUAL, st))) return rc;
        S2_TRACE("wo GEMM -> X", X, (size_t)N * H * 4);
        if (a_planes) {
            if ((rc = launch_ada_rmsnorm_planes(X, L.g_ffn, mods + (size_t)i * 4 * H + 2 * H, w.HBP, w.hbp_stride, N, H, c.norm_eps, st))) return rc;
            S2_TRACE("ada_rmsnorm(ffn) -> HB planes", w.HBP, w.hbp_stride * 6);
        } else {
            if ((rc = launch_ada_rmsnorm(X, L.g_ffn, mods + (size_t)i * 4 * H + 2 * H, w.HB, N, H, c.norm_eps, prec, st))) return rc;
            S2_TRACE("ada_rmsnorm(ffn) -> HB", w.HB, (size_t)N * H * esz);
        }
        if (fused) {                                               // [w1 ; w3] GEMM with the SwiGLU combine in the epilogue
            GemmArgs g{};
            g.A = w.HB; g.lda = H; g.Wp = L.w_13; g.M = N; g.N = 2 * I; g.K = H; g.nsplit = 1; g.epi = EPI_SWIGLU; g.out_act = w.FC; g.D = I;
            if (a_planes) { g.A = w.HBP; g.a_planes = w.hbp_stride; }
            if ((rc = s2_launch_gemm(h, g, st))) return rc;
        } else {
            if ((rc = s2_gemm(h, w.HB, H, L.w_13, nullptr, w.BIG, 2 * I, N, 2 * I, H, EPI_STORE_F32, st))) return rc;
            if ((rc = launch_swiglu(w.BIG, w.FC, N, I, prec, st))) return rc;
        }
        // fused: the residual epilogue also writes the bf16 copy the U-ViT wiring needs -- the saved skip (first half of the stack)
        // or the next layer's skip_in_linear operand (second half)
        void* shadow = nullptr;
        if (fused && i < c.depth / 2) shadow = w.SK + w.sk_stride * n_skip;
        else if (fused && i + 1 < c.depth && i + 1 > c.depth / 2) shadow = w.HB;
        S2_TRACE("w13 + SwiGLU -> FC", w.FC, (size_t)N * I * esz);
        if ((rc = s2_gemm(h, w.FC, I, L.w_2, nullptr, X, H, N, H, I, EPI_RESIDUAL, st, shadow))) return rc;
        S2_TRACE("w2 GEMM -> X", X, (size_t)N * H * 4);
        if (i < c.depth / 2) {
            if (!fused && (rc = launch_cast_pad(X, w.SK + w.sk_stride * n_skip, N, N, H, H, prec, st))) return rc;
            ++n_skip;
        }
    }
    const float* m_norm = mods + (size_t)c.depth * 4 * H;
    const float* m_gc = m_norm + 2 * H;
    const float* m_fl = m_gc + (size_t)c.wavenet_layers * 2 * W;
    // From here on every stage is row-wise except the WaveNet's few frames of context: with a tail layout only the rows the Euler step
    // reads (and their halo) are computed -- gathered by the norm / cast kernels through tail_src -- with the tail's own sequence tables.
    const SeqTab& tt = tail ? *tail : tab;
    const void* XAt = w.XA;
    if (tail) {
        if ((rc = launch_cast_pad(x_src, w.QA, tail->n_tok, src_rows, C, Kx, prec, st, tail_src))) return rc;      // QA is free after the last attention
        XAt = w.QA;
    }
    const int NT = tt.n_tok;
    // x_res = skip_linear([transformer.norm(x) | x^T])
    if ((rc = launch_ada_rmsnorm(X, h->g_norm, m_norm, w.HB, NT, H, c.norm_eps, prec, st, tail ? tail_src : nullptr))) return rc;
    S2_TRACE("final ada_rmsnorm -> HB", w.HB, (size_t)NT * H * esz);
    if ((rc = s2_gemm(h, w.HB, H, h->w_sl_a, h->b_sl, X2, H, NT, H, H, EPI_STORE_F32, st))) return rc;
    if ((rc = s2_gemm(h, XAt, Kx, h->w_sl_b, nullptr, X2, H, NT, H, Kx, EPI_RESIDUAL, st, fused ? w.HB : nullptr))) return rc;
    if (!fused && (rc = launch_cast_pad(X2, w.HB, NT, NT, H, H, prec, st))) return rc;
    S2_TRACE("skip_linear -> X2", X2, (size_t)NT * H * 4);
    if ((rc = s2_gemm(h, w.HB, H, h->w_c1, h->b_c1, w.WX, W, NT, W, H, EPI_STORE_F32, st, fused ? w.WXA : nullptr))) return rc;
    S2_TRACE("conv1 -> WX", w.WX, (size_t)NT * W * 4);
    if ((rc = s2_gemm(h, w.HB, H, h->w_rp, h->b_rp, w.RP, W, NT, W, H, EPI_STORE_F32, st))) return rc;
    // WaveNet (wavenet.py:143-166)
    int dil = 1;
    // (fused: WXA, the bf16 shadow of WX that the tap-mode GEMM reads, was written by the conv1 GEMM above)
    for (int i = 0; i < c.wavenet_layers; ++i) {
        const S2Wn& Wn = h->wn[i];
        const int last = i == c.wavenet_layers - 1;
        if (!fused && (rc = launch_im2col_reflect(w.WX, w.COL, tt, W, c.wavenet_kernel, dil, prec, st))) return rc;
        const int ro = last ? W : 2 * W;
        if (fused) {                                               // gate in the in_layer epilogue, residual / skip update in the res_skip one
            GemmArgs g{};
            // dilated reflect-padded conv as a GEMM whose A operand is an implicit im2col of the bf16 shadow (no [n][k*W] buffer)
            g.A = w.WXA; g.lda = W; g.Wp = Wn.w_in; g.bias = Wn.b_in; g.M = NT; g.N = 2 * W; g.K = c.wavenet_kernel * W;
            g.nsplit = 1; g.epi = EPI_GATE; g.out_act = w.FC; g.gvec = m_gc + (size_t)i * 2 * W; g.D = W;
            g.conv_taps = c.wavenet_kernel; g.conv_dil = dil; g.conv_W = W; g.tok_seq = tt.tok_seq; g.tok_t = tt.tok_t;
            g.seq_start = tt.seq_start; g.seq_T = tt.seq_T; g.zero_row = w.ZR;
            if ((rc = s2_launch_gemm(h, g, st))) return rc;
            GemmArgs r{};
            r.A = w.FC; r.lda = W; r.Wp = Wn.w_rs; r.bias = Wn.b_rs; r.M = NT; r.N = ro; r.K = W; r.nsplit = 1; r.epi = EPI_WN_RS;
            r.out_f32 = w.WX; r.out2 = w.OUT; r.D = W; r.wn_first = i == 0; r.wn_last = last; r.out_act2 = last ? nullptr : w.WXA;
            r.tok_seq = tt.tok_seq; r.tok_t = tt.tok_t; r.seq_len = tt.seq_len;
            S2_TRACE("wavenet in_layer + gate -> FC", w.FC, (size_t)NT * W * esz);
            if ((rc = s2_launch_gemm(h, r, st))) return rc;
            S2_TRACE("wavenet res_skip -> WX", w.WX, (size_t)NT * W * 4);
            S2_TRACE("wavenet res_skip -> OUT", w.OUT, (size_t)NT * W * 4);
        } else {
            if ((rc = s2_gemm(h, w.COL, c.wavenet_kernel * W, Wn.w_in, Wn.b_in, w.BIG, 2 * W, NT, 2 * W, c.wavenet_kernel * W, EPI_STORE_F32, st))) return rc;
            if ((rc = launch_wn_gate(w.BIG, m_gc + (size_t)i * 2 * W, w.FC, NT, W, prec, st))) return rc;
            if ((rc = s2_gemm(h, w.FC, W, Wn.w_rs, Wn.b_rs, w.BIG, ro, NT, ro, W, EPI_STORE_F32, st))) return rc;
            if ((rc = launch_wn_update(w.BIG, w.WX, w.OUT, tt, W, i == 0, last, st))) return rc;
        }
        dil *= c.wavenet_dilation_rate;
    }
    // FinalLayer + conv2
    if ((rc = launch_final_ln_mod(w.OUT, w.RP, m_fl, w.HB, tt, W, prec, st))) return rc;
    if ((rc = s2_gemm(h, w.HB, W, h->w_fl, h->b_fl, w.BIG, W, NT, W, W, EPI_STORE_F32, st, fused ? w.FC : nullptr))) return rc;
    if (!fused && (rc = launch_cast_pad(w.BIG, w.FC, NT, NT, W, W, prec, st))) return rc;
    S2_TRACE("final_layer -> FC", w.FC, (size_t)NT * W * esz);
    if ((rc = s2_gemm(h, w.FC, W, h->w_c2, h->b_c2, d_out, C, NT, C, W, EPI_STORE_F32, st))) return rc;
    S2_TRACE("conv2 -> output", d_out, (size_t)NT * C * 4);
    return ITTS_OK;
#undef S2_TRACE
}

static int s2_check(const itts_s2mel* h, const void* a, const void* b, const char* who) {
    const int da = itts_ptr_device(a), db = itts_ptr_device(b);
    if ((da >= 0 && da != h->device) || (db >= 0 && db != h->device)) {
        itts_set_error("%s: tensors are on device %d/%d but the model was created on device %d", who, da, db, h->device);
        return ITTS_ERR_ARG;
    }
    return ITTS_OK;
}

static SeqTab s2_tab(const int32_t* tok_seq, const int32_t* tok_t, const int32_t* seq_start, const int32_t* seq_T, const int32_t* seq_len,
                     int n_seq, int n_tok, int t_max) {
    SeqTab t;
    t.tok_seq = tok_seq; t.tok_t = tok_t; t.seq_start = seq_start; t.seq_T = seq_T; t.seq_len = seq_len;
    t.n_seq = n_seq; t.n_tok = n_tok; t.t_max = t_max;
    return t;
}

extern "C" int itts_s2mel_estimator(itts_s2mel* h, const float* x, const float* const_in, const float* mods, const float* rope,
                                    const int32_t* tok_seq, const int32_t* tok_t, const int32_t* seq_start, const int32_t* seq_T,
                                    const int32_t* seq_len, int n_seq, int n_tok, int t_max, float* d_out, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    if (!h || !x || !const_in || !mods || !rope || !tok_seq || !tok_t || !seq_start || !seq_T || !seq_len || !d_out || !workspace) {
        itts_set_error("s2mel_estimator: null pointer");
        return ITTS_ERR_ARG;
    }
    if (!h->finalized) { itts_set_error("s2mel_estimator: call itts_s2mel_finalize first"); return ITTS_ERR_STATE; }
    if (n_seq <= 0 || n_tok <= 0 || t_max <= 0) { itts_set_error("s2mel_estimator: bad sizes"); return ITTS_ERR_ARG; }
    ItDevGuard dg(h->device);
    if (int rc = s2_check(h, x, workspace, "s2mel_estimator")) return rc;
    const int t_pad = (t_max + 63) / 64 * 64;
    const S2Ws w0 = s2_carve(h, nullptr, n_tok, n_seq, t_pad);
    if (workspace_bytes < w0.total) { itts_set_error("s2mel_estimator: workspace too small (%zu < %zu)", workspace_bytes, w0.total); return ITTS_ERR_ARG; }
    char* base = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    const S2Ws w = s2_carve(h, base, n_tok, n_seq, t_pad);
    hipStream_t st = (hipStream_t)stream;
    const size_t kv = (size_t)((char*)w.VC - (char*)w.KC);
    HIP_TRY(hipMemsetAsync(w.ZR, 0, (size_t)((char*)w.KC - (char*)w.ZR) + 2 * kv, st));   // zero row; keys / values past a sequence's end must be finite
    const SeqTab tab = s2_tab(tok_seq, tok_t, seq_start, seq_T, seq_len, n_seq, n_tok, t_max);
    h->recs.clear();
    h->prof_stream = st;
    if (int rc = s2_trace_begin(h, st)) return rc;
    return s2_estimator(h, w, tab, t_pad, x, n_tok, const_in, mods, rope, d_out, st);
}

extern "C" int itts_s2mel_solve(itts_s2mel* h, float* x_state, const float* const_in, const float* mods, const float* rope,
                                const int32_t* tok_seq, const int32_t* tok_t, const int32_t* seq_start, const int32_t* seq_T,
                                const int32_t* seq_len, const int32_t* prompt_len, int n_seq, int n_tok, int t_max, int n_branch,
                                int n_steps, const float* t_span, float cfg_rate, void* workspace, size_t workspace_bytes, void* stream) {
    if (!h || !x_state || !const_in || !mods || !rope || !tok_seq || !tok_t || !seq_start || !seq_T || !seq_len || !prompt_len || !t_span || !workspace) {
        itts_set_error("s2mel_solve: null pointer");
        return ITTS_ERR_ARG;
    }
    if (!h->finalized) { itts_set_error("s2mel_solve: call itts_s2mel_finalize first"); return ITTS_ERR_STATE; }
    if (n_seq <= 0 || n_tok <= 0 || t_max <= 0 || n_steps < 1 || (n_branch != 1 && n_branch != 2) || n_tok % n_branch || n_seq % n_branch) {
        itts_set_error("s2mel_solve: bad sizes (n_seq=%d n_tok=%d n_branch=%d n_steps=%d)", n_seq, n_tok, n_branch, n_steps);
        return ITTS_ERR_ARG;
    }
    ItDevGuard dg(h->device);
    if (int rc = s2_check(h, x_state, workspace, "s2mel_solve")) return rc;
    const int t_pad = (t_max + 63) / 64 * 64;
    const S2Ws w0 = s2_carve(h, nullptr, n_tok, n_seq, t_pad);
    if (workspace_bytes < w0.total) { itts_set_error("s2mel_solve: workspace too small (%zu < %zu)", workspace_bytes, w0.total); return ITTS_ERR_ARG; }
    char* base = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    const S2Ws w = s2_carve(h, base, n_tok, n_seq, t_pad);
    hipStream_t st = (hipStream_t)stream;
    const size_t kv = (size_t)((char*)w.VC - (char*)w.KC);
    HIP_TRY(hipMemsetAsync(w.ZR, 0, (size_t)((char*)w.KC - (char*)w.ZR) + 2 * kv, st));
    const SeqTab tab = s2_tab(tok_seq, tok_t, seq_start, seq_T, seq_len, n_seq, n_tok, t_max);
    const int mps = itts_s2mel_mods_per_step(h);
    h->recs.clear();
    h->prof_stream = st;
    if (int rc = s2_trace_begin(h, st)) return rc;
    for (int step = 0; step < n_steps; ++step) {                   // flow_matching.py:84-113
        const bool use_tail = h->tail_set && h->tail.n_seq == n_seq && h->tail.n_tok > 0 && h->tail.n_tok <= n_tok && h->tail.n_tok % n_branch == 0;
        int rc = s2_estimator(h, w, tab, t_pad, x_state, n_tok / n_branch, const_in, mods + (size_t)step * mps, rope, w.D, st, 0.0,
                              use_tail ? &h->tail : nullptr, use_tail ? h->tail_src : nullptr);
        if (rc) return rc;
        const float dt = t_span[step + 1] - t_span[step];
        if ((rc = launch_euler_update(x_state, w.D, tab, prompt_len, h->cfg.in_channels, n_branch, dt, cfg_rate, st,
                                      use_tail ? h->tail_base : nullptr, use_tail ? h->tail.n_tok / n_branch : 0))) return rc;
    }
    return ITTS_OK;
}

// Dead-row elimination for the following itts_s2mel_solve calls (see itts_s2mel.tail): device tables of the tail layout -- per tail row
// its sequence and its frame RELATIVE to the sequence's cut, per sequence the first tail row / tail frames / valid tail frames -- plus
// tail_src [n_tail] (full-layout row of a tail row) and tail_base [n_seq] (tail row of frame t = tail_base[s] + t).  The caller keeps
// the arrays alive; all-null clears.  The cut of a sequence must lie at least one WaveNet receptive field before its first target frame.
extern "C" int itts_s2mel_set_tail(itts_s2mel* h, const int32_t* tok_seq, const int32_t* tok_t, const int32_t* seq_start, const int32_t* seq_T,
                                   const int32_t* seq_len, const int32_t* tail_src, const int32_t* tail_base, int n_seq, int n_tail, int t_max) {
    if (!h) { itts_set_error("s2mel_set_tail: null"); return ITTS_ERR_ARG; }
    if (!tok_seq) { h->tail_set = false; return ITTS_OK; }
    if (!tok_t || !seq_start || !seq_T || !seq_len || !tail_src || !tail_base || n_seq <= 0 || n_tail <= 0 || t_max <= 0) {
        itts_set_error("s2mel_set_tail: bad args");
        return ITTS_ERR_ARG;
    }
    h->tail = s2_tab(tok_seq, tok_t, seq_start, seq_T, seq_len, n_seq, n_tail, t_max);
    h->tail_src = tail_src; h->tail_base = tail_base; h->tail_set = true;
    return ITTS_OK;
}

// ---- unit-level entry point (parity tests): RoPE + split + non-causal attention of one layer ------------------------------
// qkv f32 [n_tok][3H] (the fused wqkv output) -> out act dtype [n_tok][H] = softmax(rope(q) rope(k)^T / 8, keys < seq_len) v.
// scratch: q act [n_tok][H] + K + V act [n_seq * heads * t_pad * 64] each, t_pad = t_max rounded up to 64.
// precision ITTS_PREC_F32X3: q / K / V^T are produced in f32, K and V^T are then split into three bf16 planes each (+ 2 x 6 bytes per
// element of scratch) and the products run on flash_attn_x3_kernel with option x3_products plane products; out is f32.
extern "C" size_t itts_s2mel_attention_scratch_bytes(int n_tok, int n_seq, int heads, int t_max, int precision) {
    const size_t esz = precision == PREC_BF16 ? 2 : 4;
    const size_t t_pad = (size_t)(t_max + 63) / 64 * 64;
    const size_t planes = precision == PREC_F32X3 ? 2 * s_a256((size_t)n_seq * heads * t_pad * 64 * 6) : 0;
    return s_a256((size_t)n_tok * heads * 64 * esz) + 2 * s_a256((size_t)n_seq * heads * t_pad * 64 * esz) + planes + 256;
}

extern "C" int itts_s2mel_attention_forward(const float* qkv, const float* rope, const int32_t* tok_seq, const int32_t* tok_t,
                                            const int32_t* seq_start, const int32_t* seq_T, const int32_t* seq_len, int n_seq, int n_tok,
                                            int t_max, int heads, int precision, void* out, void* scratch, size_t scratch_bytes, void* stream) {
    if (!qkv || !rope || !tok_seq || !tok_t || !seq_start || !seq_T || !seq_len || !out || !scratch) { itts_set_error("s2mel_attention: null pointer"); return ITTS_ERR_ARG; }
    if (n_seq <= 0 || n_tok <= 0 || t_max <= 0 || heads <= 0 || (precision != PREC_F32 && precision != PREC_BF16 && precision != PREC_F32X3)) { itts_set_error("s2mel_attention: bad sizes"); return ITTS_ERR_ARG; }
    if (scratch_bytes < itts_s2mel_attention_scratch_bytes(n_tok, n_seq, heads, t_max, precision)) { itts_set_error("s2mel_attention: scratch too small"); return ITTS_ERR_ARG; }
    const size_t esz = precision == PREC_BF16 ? 2 : 4;
    const int t_pad = (t_max + 63) / 64 * 64;
    char* base = (char*)(((uintptr_t)scratch + 255) & ~(uintptr_t)255);
    char* q = base;
    char* k = q + s_a256((size_t)n_tok * heads * 64 * esz);
    const size_t kv = s_a256((size_t)n_seq * heads * t_pad * 64 * esz);
    char* v = k + kv;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(k, 0, 2 * kv, st));
    const SeqTab tab = s2_tab(tok_seq, tok_t, seq_start, seq_T, seq_len, n_seq, n_tok, t_max);
    int rc = launch_rope_split(qkv, rope, q, k, v, tab, heads, t_pad, precision == PREC_F32X3 ? PREC_F32 : precision, st);
    if (rc) return rc;
    if (precision == PREC_F32X3) {
        const size_t n = (size_t)n_seq * heads * t_pad * 64;
        char* kp = v + kv;
        char* vp = kp + s_a256(n * 6);
        if ((rc = launch_split_planes((const float*)k, kp, n, st))) return rc;
        if ((rc = launch_split_planes((const float*)v, vp, n, st))) return rc;
        return launch_s2mel_attention_x3(q, kp, vp, out, tab, heads, t_pad, st);
    }
    return launch_s2mel_attention(q, k, v, out, tab, heads, t_pad, precision, st);
}
