// C-ABI entry points for the GPT speech-token decoder (see include/indextts_hip.h for the reference call sites).
#include <math.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/indextts_hip.h"
#include "gpt_kernels.h"

// ---- host-side weight packing ----------------------------------------------------------------------------------
extern "C" size_t itts_packed_gemm_bytes(int K, int N, int precision) {
    if (precision == PREC_F32X3) return (size_t)((N + 15) / 16) * (K / 32) * 3 * 64 * 16;       // three bf16 planes per 32-deep k-block
    const int KB = precision == PREC_BF16 ? 32 : 16;
    return (size_t)((N + 15) / 16) * (K / KB) * 64 * 16;
}

// w: [K][N] row-major (HF Conv1D) when transposed == 0, [N][K] (nn.Linear) when transposed != 0
extern "C" int itts_pack_gemm_weight(const float* w, int K, int N, int transposed, int precision, void* out) {
    const int KB = precision == PREC_F32 ? 16 : 32;
    if (!w || !out || K <= 0 || N <= 0 || K % KB) {
        itts_set_error("pack_gemm: K=%d must be a positive multiple of %d", K, KB);
        return ITTS_ERR_ARG;
    }
    if (precision != PREC_F32 && precision != PREC_BF16 && precision != PREC_F32X3) { itts_set_error("pack_gemm: precision %d", precision); return ITTS_ERR_ARG; }
    const int ntiles = (N + 15) / 16, nkb = K / KB, per = KB / 4;   // per = k elements per lane
    if (precision == PREC_F32X3) {
        // [N/16][K/32][3 planes][64 lanes][8 bf16]: x = h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m) (exact: 24 bits);
        // lane (kg, n) element j holds k = 32 kb + 4 kg + j (j < 4) or 32 kb + 16 + 4 kg + (j - 4): the two 16-byte pieces of the
        // f32 activation row that gemm_x3_kernel's lane reads
        auto b2f = [](uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; };
        for (int nt = 0; nt < ntiles; ++nt)
            for (int kb = 0; kb < nkb; ++kb)
                for (int lane = 0; lane < 64; ++lane) {
                    const int n = nt * 16 + (lane & 15), kg = lane >> 4;
                    for (int j = 0; j < 8; ++j) {
                        const int k = kb * 32 + (j < 4 ? 4 * kg + j : 16 + 4 * kg + (j - 4));
                        float v = 0.f;
                        if (n < N) v = transposed ? w[(size_t)n * K + k] : w[(size_t)k * N + n];
                        const uint16_t h = host_f32_to_bf16(v);
                        const float r1 = v - b2f(h);
                        const uint16_t m = host_f32_to_bf16(r1);
                        const float r2 = r1 - b2f(m);
                        const uint16_t l = host_f32_to_bf16(r2);
                        const size_t blk = ((size_t)nt * nkb + kb) * 3;
                        ((u16*)out)[((blk + 0) * 64 + lane) * 8 + j] = h;
                        ((u16*)out)[((blk + 1) * 64 + lane) * 8 + j] = m;
                        ((u16*)out)[((blk + 2) * 64 + lane) * 8 + j] = l;
                    }
                }
        return ITTS_OK;
    }
    for (int nt = 0; nt < ntiles; ++nt)
        for (int kb = 0; kb < nkb; ++kb)
            for (int lane = 0; lane < 64; ++lane) {
                const int n = nt * 16 + (lane & 15);
                const size_t base = (((size_t)nt * nkb + kb) * 64 + lane);
                for (int j = 0; j < per; ++j) {
                    const int k = kb * KB + (lane >> 4) * per + j;
                    float v = 0.f;
                    if (n < N) v = transposed ? w[(size_t)n * K + k] : w[(size_t)k * N + n];
                    if (precision == PREC_BF16) ((u16*)out)[base * 8 + j] = host_f32_to_bf16(v);
                    else ((float*)out)[base * 4 + j] = v;
                }
            }
    return ITTS_OK;
}

// ---- model object ----------------------------------------------------------------------------------------------
struct GLayer {
    float *ln1_g = 0, *ln1_b = 0, *ln2_g = 0, *ln2_b = 0;
    void *w_qkv = 0, *w_proj = 0, *w_fc = 0, *w_fc2 = 0;
    float *b_qkv = 0, *b_proj = 0, *b_fc = 0, *b_fc2 = 0;
};

struct itts_gpt {
    itts_gpt_config cfg;
    std::vector<GLayer> layers;
    float *lnf_g = 0, *lnf_b = 0, *fn_g = 0, *fn_b = 0;
    void* w_head = 0;
    float* b_head = 0;
    float *mel_emb = 0, *mel_pos = 0;
    std::vector<void*> owned;
    bool finalized = false;
    hipStream_t stream = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr, ev_t0 = nullptr, ev_t1 = nullptr, ev_t2 = nullptr;
    int* host_flag = nullptr;          // pinned
    unsigned char* host_fin = nullptr; // pinned, capacity fin_cap
    int fin_cap = 0;
    float last_prefill_ms = 0, last_decode_ms = 0;
    int last_steps = 0;
    int device = -1;                   // device current at itts_gpt_create: owns weights, stream, events
    // Instantiated decode-step graphs, reused across generate calls.  A graph bakes in every pointer and scalar of the step's
    // 171 launches, so the key is everything those derive from: workspace base, carve shape (rows, bucketed prompt length,
    // cache stride), output / uniforms pointers, the generation parameters.  Small LRU (a serving process sees a handful of
    // (batch, prompt-bucket) shapes).
    struct GraphEntry {
        const void* base; const void* tokens; const void* uniforms; const void* aux0; const void* aux1; const void* aux2;
        int nseq, nb, Sb, Tmax, S;           // S: only where the step bakes the exact prompt length in (beam kernels), else 0
        unsigned opt_epoch;                  // itts_opt_epoch() at capture: a graph bakes the kernel choices of the options in
        itts_gen_params gp;
        hipGraphExec_t exec;
        unsigned long long stamp;
    };
    std::vector<GraphEntry> graphs;
    unsigned long long graph_clock = 0;
    int graph_hits = 0, graph_captures = 0;
    // chunked generation (itts_gpt_generate_chunk): tokens generated so far and the shape / workspace of the call being resumed
    int chunk_steps = 0, chunk_nseq = 0, chunk_S = 0, chunk_max_new = 0;
    const void* chunk_ws = nullptr;
    bool chunk_admitted = false;           // a row has been admitted into the suspended loop: the decode step reads the per-row position shifts
    // row compaction of a ragged decode batch (itts_gpt_set_compaction): finished utterances leave the running batch in steps of
    // `compact_gran` rows, so the step's cost follows the live rows.  cur_slots[i] = utterance carried by dense row i.
    bool compact = true;
    int compact_gran = 8;
    std::vector<int> cur_slots;
    bool cur_mapped = false;
    int* host_map = nullptr;               // pinned [2][map_cap]: new slot map | gather sources
    int map_cap = 0;
    long long last_row_steps = 0;          // sum over the decode steps of the rows that step ran (itts_gpt_compaction_stats)
    int last_compactions = 0;
    const int32_t* row_limits = nullptr;   // device [row_limits_n] per-utterance token caps for the next generate calls, or null
    int row_limits_n = 0;
    int chunk_return_finished = 0;         // itts_gpt_set_chunk_return: a chunk call returns at a flag check once that many utterances have finished
};
#define GRAPH_CACHE_MAX 24        // a ragged batch replays one graph per live-row bucket (8 at the bench shape) beside the callers' own shapes
// prompt lengths are bucketed to multiples of 32 for the workspace carve and the cache stride, so that prompts of nearby lengths
// share one workspace layout and therefore one decode graph
static inline int s_bucket(int S) { return (S + 31) & ~31; }

static hipGraphExec_t graph_lookup(itts_gpt* h, const itts_gpt::GraphEntry& k) {
    for (auto& e : h->graphs)
        if (e.base == k.base && e.tokens == k.tokens && e.uniforms == k.uniforms && e.aux0 == k.aux0 && e.aux1 == k.aux1 && e.aux2 == k.aux2 &&
            e.nseq == k.nseq && e.nb == k.nb && e.Sb == k.Sb && e.Tmax == k.Tmax && e.S == k.S && e.opt_epoch == k.opt_epoch && memcmp(&e.gp, &k.gp, sizeof(k.gp)) == 0) {
            e.stamp = ++h->graph_clock;
            ++h->graph_hits;
            return e.exec;
        }
    return nullptr;
}
static void graph_insert(itts_gpt* h, itts_gpt::GraphEntry k, hipGraphExec_t exec) {
    k.exec = exec;
    k.stamp = ++h->graph_clock;
    ++h->graph_captures;
    if ((int)h->graphs.size() >= GRAPH_CACHE_MAX) {
        size_t old = 0;
        for (size_t i = 1; i < h->graphs.size(); ++i) if (h->graphs[i].stamp < h->graphs[old].stamp) old = i;
        (void)hipGraphExecDestroy(h->graphs[old].exec);
        h->graphs[old] = k;
    } else {
        h->graphs.push_back(k);
    }
}

static int g_upload(itts_gpt* h, const void* host, size_t bytes, void** dst) {
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, bytes));
    h->owned.push_back(d);
    HIP_TRY(hipMemcpy(d, host, bytes, hipMemcpyHostToDevice));
    *dst = d;
    return ITTS_OK;
}

extern "C" int itts_gpt_create(const itts_gpt_config* cfg, itts_gpt** out) {
    if (!cfg || !out) { itts_set_error("gpt_create: null"); return ITTS_ERR_ARG; }
    const itts_gpt_config& c = *cfg;
    const int KB = c.precision == PREC_BF16 ? 32 : 16;
    if (c.layers < 1 || c.heads < 1 || c.model_dim != c.heads * 64 || c.model_dim % KB || c.vocab < 2 ||
        (c.precision != PREC_F32 && c.precision != PREC_BF16) || c.n_mel_pos < 3) {
        itts_set_error("gpt_create: unsupported config (layers=%d dim=%d heads=%d: head_dim must be 64; vocab=%d prec=%d)",
                       c.layers, c.model_dim, c.heads, c.vocab, c.precision);
        return ITTS_ERR_ARG;
    }
    itts_gpt* h = new itts_gpt();
    h->cfg = c;
    h->device = itts_current_device();
    h->layers.resize(c.layers);
    *out = h;
    return ITTS_OK;
}

extern "C" int itts_gpt_device(const itts_gpt* h) { return h ? h->device : -1; }

extern "C" void itts_gpt_destroy(itts_gpt* h) {
    if (!h) return;
    ItDevGuard dg(h->device);
    for (auto& e : h->graphs) (void)hipGraphExecDestroy(e.exec);
    for (void* p : h->owned) (void)hipFree(p);
    if (h->host_flag) (void)hipHostFree(h->host_flag);
    if (h->host_fin) (void)hipHostFree(h->host_fin);
    if (h->host_map) (void)hipHostFree(h->host_map);
    if (h->ev_in) { (void)hipEventDestroy(h->ev_in); (void)hipEventDestroy(h->ev_out); (void)hipEventDestroy(h->ev_t0);
                    (void)hipEventDestroy(h->ev_t1); (void)hipEventDestroy(h->ev_t2); }
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

static bool g_eat(const char*& s, const char* lit) {
    size_t n = strlen(lit);
    if (strncmp(s, lit, n) == 0) { s += n; return true; }
    return false;
}

static size_t numel(const int64_t* shape, int ndim) {
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
    return n;
}

static int load_vec(itts_gpt* h, const float* data, const int64_t* shape, int ndim, size_t want, float** dst, const char* name) {
    if (numel(shape, ndim) != want) { itts_set_error("%s: expected %zu elements, got %zu", name, want, numel(shape, ndim)); return ITTS_ERR_ARG; }
    return g_upload(h, data, want * sizeof(float), (void**)dst);
}

static int load_mat(itts_gpt* h, const float* data, const int64_t* shape, int ndim, int K, int N, int transposed, void** dst,
                    const char* name) {
    const int64_t d0 = transposed ? N : K, d1 = transposed ? K : N;
    if (ndim != 2 || shape[0] != d0 || shape[1] != d1) {
        itts_set_error("%s: expected shape [%lld,%lld]", name, (long long)d0, (long long)d1);
        return ITTS_ERR_ARG;
    }
    std::vector<char> pk(itts_packed_gemm_bytes(K, N, h->cfg.precision));
    int rc = itts_pack_gemm_weight(data, K, N, transposed, h->cfg.precision, pk.data());
    if (rc) return rc;
    return g_upload(h, pk.data(), pk.size(), dst);
}

// Reference state-dict names: gpt.h.{i}.{ln_1,attn.c_attn,attn.c_proj,ln_2,mlp.c_fc,mlp.c_proj}.{weight,bias},
// gpt.ln_f, final_norm, mel_head, mel_embedding.weight, mel_pos_embedding.emb.weight.  HF Conv1D weights are [in, out].
extern "C" int itts_gpt_load_tensor(itts_gpt* h, const char* name, const float* data, const int64_t* shape, int ndim) {
    if (!h || !name || !data || !shape || ndim < 1 || ndim > 2) { itts_set_error("gpt_load_tensor: bad args"); return ITTS_ERR_ARG; }
    ItDevGuard dg(h->device);
    const itts_gpt_config& c = h->cfg;
    const int D = c.model_dim;
    const char* s = name;
    h->finalized = false;
    if (g_eat(s, "gpt.h.")) {
        int i = 0;
        if (*s < '0' || *s > '9') { itts_set_error("bad name %s", name); return ITTS_ERR_ARG; }
        while (*s >= '0' && *s <= '9') i = i * 10 + (*s++ - '0');
        if (i >= c.layers || !g_eat(s, ".")) { itts_set_error("bad layer index in %s", name); return ITTS_ERR_ARG; }
        GLayer& L = h->layers[i];
        if (!strcmp(s, "ln_1.weight")) return load_vec(h, data, shape, ndim, D, &L.ln1_g, name);
        if (!strcmp(s, "ln_1.bias")) return load_vec(h, data, shape, ndim, D, &L.ln1_b, name);
        if (!strcmp(s, "ln_2.weight")) return load_vec(h, data, shape, ndim, D, &L.ln2_g, name);
        if (!strcmp(s, "ln_2.bias")) return load_vec(h, data, shape, ndim, D, &L.ln2_b, name);
        if (!strcmp(s, "attn.c_attn.weight")) return load_mat(h, data, shape, ndim, D, 3 * D, 0, &L.w_qkv, name);
        if (!strcmp(s, "attn.c_attn.bias")) return load_vec(h, data, shape, ndim, 3 * D, &L.b_qkv, name);
        if (!strcmp(s, "attn.c_proj.weight")) return load_mat(h, data, shape, ndim, D, D, 0, &L.w_proj, name);
        if (!strcmp(s, "attn.c_proj.bias")) return load_vec(h, data, shape, ndim, D, &L.b_proj, name);
        if (!strcmp(s, "mlp.c_fc.weight")) return load_mat(h, data, shape, ndim, D, 4 * D, 0, &L.w_fc, name);
        if (!strcmp(s, "mlp.c_fc.bias")) return load_vec(h, data, shape, ndim, 4 * D, &L.b_fc, name);
        if (!strcmp(s, "mlp.c_proj.weight")) return load_mat(h, data, shape, ndim, 4 * D, D, 0, &L.w_fc2, name);
        if (!strcmp(s, "mlp.c_proj.bias")) return load_vec(h, data, shape, ndim, D, &L.b_fc2, name);
    }
    if (!strcmp(name, "gpt.ln_f.weight")) return load_vec(h, data, shape, ndim, D, &h->lnf_g, name);
    if (!strcmp(name, "gpt.ln_f.bias")) return load_vec(h, data, shape, ndim, D, &h->lnf_b, name);
    if (!strcmp(name, "final_norm.weight")) return load_vec(h, data, shape, ndim, D, &h->fn_g, name);
    if (!strcmp(name, "final_norm.bias")) return load_vec(h, data, shape, ndim, D, &h->fn_b, name);
    if (!strcmp(name, "mel_head.weight")) return load_mat(h, data, shape, ndim, D, c.vocab, 1, &h->w_head, name);
    if (!strcmp(name, "mel_head.bias")) return load_vec(h, data, shape, ndim, c.vocab, &h->b_head, name);
    if (!strcmp(name, "mel_embedding.weight")) return load_vec(h, data, shape, ndim, (size_t)c.vocab * D, &h->mel_emb, name);
    if (!strcmp(name, "mel_pos_embedding.emb.weight")) return load_vec(h, data, shape, ndim, (size_t)c.n_mel_pos * D, &h->mel_pos, name);
    itts_set_error("gpt_load_tensor: unrecognised tensor name '%s'", name);
    return ITTS_ERR_ARG;
}

extern "C" int itts_gpt_finalize(itts_gpt* h) {
    if (!h) { itts_set_error("gpt_finalize: null"); return ITTS_ERR_ARG; }
    ItDevGuard dg(h->device);
    std::string missing;
    for (size_t i = 0; i < h->layers.size(); ++i) {
        const GLayer& L = h->layers[i];
        if (!L.ln1_g || !L.ln1_b || !L.ln2_g || !L.ln2_b || !L.w_qkv || !L.w_proj || !L.w_fc || !L.w_fc2 || !L.b_qkv || !L.b_proj ||
            !L.b_fc || !L.b_fc2)
            missing += "gpt.h." + std::to_string(i) + ".* ";
    }
    if (!h->lnf_g || !h->lnf_b) missing += "gpt.ln_f ";
    if (!h->fn_g || !h->fn_b) missing += "final_norm ";
    if (!h->w_head || !h->b_head) missing += "mel_head ";
    if (!h->mel_emb) missing += "mel_embedding.weight ";
    if (!h->mel_pos) missing += "mel_pos_embedding.emb.weight ";
    if (!missing.empty()) { itts_set_error("gpt_finalize: missing tensors: %s", missing.c_str()); return ITTS_ERR_STATE; }
    if (!h->stream) {
        HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&h->ev_out, hipEventDisableTiming));
        HIP_TRY(hipEventCreate(&h->ev_t0));
        HIP_TRY(hipEventCreate(&h->ev_t1));
        HIP_TRY(hipEventCreate(&h->ev_t2));
        HIP_TRY(hipHostMalloc((void**)&h->host_flag, 64, hipHostMallocDefault));
    }
    h->finalized = true;
    return ITTS_OK;
}

// caller tensors must live on the device that owns the engine (weights, stream): a mismatch would fault or go through
// silent peer access
static int check_same_device(const itts_gpt* h, const void* in, const void* ws, const char* who) {
    const int di = itts_ptr_device(in), dw = itts_ptr_device(ws);
    if ((di >= 0 && di != h->device) || (dw >= 0 && dw != h->device)) {
        itts_set_error("%s: tensors are on device %d/%d but the engine was created on device %d", who, di, dw, h->device);
        return ITTS_ERR_ARG;
    }
    return ITTS_OK;
}

// ---- workspace -------------------------------------------------------------------------------------------------
static size_t a256(size_t x) { return (x + 255) & ~(size_t)255; }

struct GptWs {
    char *kc, *vc;      // [L][nseq][H][Tmax][64] cache dtype
    float* x;           // [rows][D]
    float* x2;          // [16][D] second residual buffer of the LayerNorm-fused decode GEMMs (1-16 rows; run_layers alternates x / x2)
    char* hbuf;         // [rows][D] act
    float* qbuf;        // [rows][D]
    char* attn;         // [rows][D] act
    char* fc;           // [rows][4D] act
    float* partial;     // [4][nseq][D]
    char* hlast;        // [nseq][D] act
    float* logits;      // [nseq][V]
    unsigned char* seen;     // [nseq][V]
    unsigned char* finished; // [nseq]
    int* state;         // [0] step, [1] pos
    int* pad;           // [nseq]
    int* pen_ids;       // [16]
    int* slot_map;      // [nseq] dense row -> utterance (row compaction; unused while the batch is uncompacted)
    int* gather_src;    // [nseq] compaction: the old dense row each new dense row is taken from
    int* row_step0;     // [nseq] the step at which the utterance joined the running batch (itts_gpt_admit_rows; 0 otherwise)
    int* row_shift;     // [nseq] batch position counter - the cache row's own position (0 for the rows of the first call; admitted rows keep their
                        // keys at their own positions 0 .. prompt + generated, whatever the running batch's counter says)
    // beam search (nb > 1)
    unsigned char* seen2;    // second seen buffer
    int* row_map[2];         // [nseq][Tmax]
    float* beam_scores; float* next_scores; int* next_tokens; int* next_indices;   // [nseq]
    BeamHyp* hyps; int* n_hyps; float* worst; unsigned char* done;                 // per utterance
    int* hist_tok; int* hist_par;                                                   // [max_new][nseq]
    int* surv_idx; float* surv_val; int* surv_n;                                    // [nseq][64], [nseq]
    size_t total;
    size_t layer_cache_bytes;
};

static GptWs carve(const itts_gpt_config& c, char* base, int nseq, int S, int Tmax, int nb = 1) {
    GptWs w;
    const size_t esz = c.precision == PREC_BF16 ? 2 : 4;
    const size_t D = c.model_dim, rows = (size_t)nseq * S;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += a256(bytes); return p; };
    w.layer_cache_bytes = a256((size_t)nseq * D * Tmax * esz);
    w.kc = take(w.layer_cache_bytes * c.layers);
    w.vc = take(w.layer_cache_bytes * c.layers);
    w.x = (float*)take(rows * D * 4);
    w.x2 = (float*)take((size_t)16 * D * 4);
    w.hbuf = take(rows * D * esz);
    w.qbuf = (float*)take(rows * D * 4);
    w.attn = take(rows * D * esz);
    w.fc = take(rows * 4 * D * esz);
    w.partial = (float*)take((size_t)4 * nseq * D * 4);
    w.hlast = take((size_t)nseq * D * esz);
    w.logits = (float*)take((size_t)nseq * c.vocab * 4);
    w.seen = (unsigned char*)take((size_t)nseq * c.vocab);
    w.finished = (unsigned char*)take(nseq);
    w.state = (int*)take(64);
    w.pad = (int*)take((size_t)nseq * 4);
    w.pen_ids = (int*)take(64);
    w.slot_map = (int*)take((size_t)nseq * 4);
    w.gather_src = (int*)take((size_t)nseq * 4);
    w.row_step0 = (int*)take((size_t)nseq * 4);
    w.row_shift = (int*)take((size_t)nseq * 4);
    w.seen2 = nullptr; w.row_map[0] = w.row_map[1] = nullptr;
    if (nb > 1) {
        const int B = nseq / nb, max_new = Tmax - S;
        w.seen2 = (unsigned char*)take((size_t)nseq * c.vocab);
        w.row_map[0] = (int*)take((size_t)nseq * Tmax * 4);
        w.row_map[1] = (int*)take((size_t)nseq * Tmax * 4);
        w.beam_scores = (float*)take((size_t)nseq * 4);
        w.next_scores = (float*)take((size_t)nseq * 4);
        w.next_tokens = (int*)take((size_t)nseq * 4);
        w.next_indices = (int*)take((size_t)nseq * 4);
        w.hyps = (BeamHyp*)take((size_t)B * BEAM_MAX * sizeof(BeamHyp));
        w.n_hyps = (int*)take((size_t)B * 4);
        w.worst = (float*)take((size_t)B * 4);
        w.done = (unsigned char*)take((size_t)B);
        w.hist_tok = (int*)take((size_t)max_new * nseq * 4);
        w.hist_par = (int*)take((size_t)max_new * nseq * 4);
        w.surv_idx = (int*)take((size_t)nseq * 64 * 4);
        w.surv_val = (float*)take((size_t)nseq * 64 * 4);
        w.surv_n = (int*)take((size_t)nseq * 4);
    }
    w.total = off + 256;
    return w;
}

extern "C" size_t itts_gpt_workspace_bytes(const itts_gpt* h, int nseq, int S, int Tmax) {
    if (!h || nseq <= 0 || S <= 0 || Tmax < S) return 0;
    return carve(h->cfg, nullptr, nseq, s_bucket(S), Tmax + s_bucket(S) - S).total + 256;
}
extern "C" size_t itts_gpt_beam_workspace_bytes(const itts_gpt* h, int n_utts, int num_beams, int S, int Tmax) {
    if (!h || n_utts <= 0 || num_beams < 2 || num_beams > BEAM_MAX || S <= 0 || Tmax <= S) return 0;
    return carve(h->cfg, nullptr, n_utts * num_beams, s_bucket(S), Tmax + s_bucket(S) - S, num_beams).total + 256;
}

// ---- small state kernels ---------------------------------------------------------------------------------------
__global__ void set_state_kernel(int* state, int step, int pos) { state[0] = step; state[1] = pos; state[2] = 0; }
__global__ void set_seed_kernel(int* state, unsigned long long seed) { *(unsigned long long*)(state + 4) = seed; }   // state[4..5]
__global__ void fill_i64_kernel(long long* p, long long v, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void mark_seen_kernel(unsigned char* seen, const int* ids, int n_ids, int V) {
    const int b = blockIdx.x;
    if ((int)threadIdx.x < n_ids) {
        const int id = ids[threadIdx.x];
        if (id >= 0 && id < V) seen[(size_t)b * V + id] = 1;
    }
}

// row compaction: x_new[i] = x_old[src[i]] through a scratch copy (src[i] >= i, but rows are moved by independent blocks)
__global__ void gather_rows_kernel(const float* __restrict__ x, const int* __restrict__ src, float* __restrict__ tmp, int D) {
    const int i = blockIdx.x;
    const float* s = x + (size_t)src[i] * D;
    for (int c = threadIdx.x; c < D; c += blockDim.x) tmp[(size_t)i * D + c] = s[c];
}
__global__ void copy_rows_kernel(const float* __restrict__ tmp, float* __restrict__ x, int D) {
    const int i = blockIdx.x;
    for (int c = threadIdx.x; c < D; c += blockDim.x) x[(size_t)i * D + c] = tmp[(size_t)i * D + c];
}

// ---- one transformer pass --------------------------------------------------------------------------------------
// rows = nseq*S new positions (S = 1 for a decode step).  prefill: direct residual epilogues + big-tile GEMMs;
// decode: split-K partials reduced inside the next LayerNorm kernel.
// x_cur (optional out): the buffer that holds the residual stream after the pass (w.x, or w.x2 when the LayerNorm-fused decode GEMMs ran)
static int run_layers(itts_gpt* h, const GptWs& w, int nseq, int S, int Tmax, bool prefill, const int* pos_ptr, const int* pad,
                      bool* pending, hipStream_t st, bool beam = false, int seq_mul = 1, const int* seq_map = nullptr, float** x_cur = nullptr,
                      const int* pos_shift = nullptr) {
    const itts_gpt_config& c = h->cfg;
    const int D = c.model_dim, prec = c.precision, rows = nseq * S;
    int rc;
    const float* pend_bias = nullptr;   // bias of a GEMM whose partials are pending reduction
    // Decode steps of 1-8 rows (one utterance, its beams, the 8-utterance shard of an 8-GPU run): the two LayerNorm launches of a layer are fused into
    // the GEMMs that consume them (gemm_decode_ln_kernel at 1-4 rows, gemm_decode_lnw_kernel at 5-8: same arithmetic, bitwise the same results; option
    // decode_fuse_ln = 0 is the A/B switch).  The fused kernel's block 0 writes the updated residual to the OTHER buffer (its sibling blocks still read
    // the current one): cur / alt alternate.  Measured (profiles/r04a, r04h; ms per token at 560 tokens, fused / separate): 1 row 0.776 / 0.915, 4 rows
    // 0.874 / 0.957, 5 rows 0.935 / 0.963, 8 rows 0.977 / 0.986 -- and 12 rows 1.110 / 1.016, 16 rows 1.294 / 1.099: every block repeats the LayerNorm
    // of every row, which outgrows the launch it removes.  decode_fuse_ln = 2 keeps the 9-16-row form reachable (bitwise equal, tested).
    const int fuse_opt = itts_opt(OPT_DECODE_FUSE_LN);
    const bool fuse_ln = fuse_opt != 0 && rows <= (fuse_opt >= 2 ? 16 : 8) && !prefill && S == 1 && prec == PREC_BF16 &&
                         gemm_decode_ln_ok(rows, D, EPI_QKV) && w.x2 != nullptr;
    float *cur = w.x, *alt = w.x2;
    for (int l = 0; l < c.layers; ++l) {
        const GLayer& L = h->layers[l];
        LnArgs ln{};
        ln.x = cur; ln.partial = pend_bias ? w.partial : nullptr; ln.nsplit = 4; ln.bias_prev = pend_bias;
        ln.g1 = L.ln1_g; ln.b1 = L.ln1_b; ln.g2 = nullptr; ln.b2 = nullptr; ln.out = w.hbuf; ln.out_f32 = 0;
        ln.rows = rows; ln.D = D; ln.in_row_mul = 1; ln.in_row_add = 0; ln.eps = c.ln_eps;
        if (!fuse_ln && (rc = launch_ln(ln, prec, st))) return rc;

        GemmArgs g{};
        g.A = w.hbuf; g.lda = D; g.Wp = L.w_qkv; g.bias = L.b_qkv; g.M = rows; g.N = 3 * D; g.K = D; g.nsplit = 1; g.epi = EPI_QKV;
        g.qbuf = w.qbuf; g.kcache = w.kc + w.layer_cache_bytes * l; g.vcache = w.vc + w.layer_cache_bytes * l;
        g.pos_ptr = pos_ptr; g.S = S; g.H = c.heads; g.Tmax = Tmax; g.D = D; g.seq_mul = seq_mul; g.seq_map = seq_map; g.pos_shift = pos_shift;
        if (fuse_ln) {
            g.ln_x = cur; g.ln_partial = ln.partial; g.ln_bias_prev = pend_bias; g.ln_g = L.ln1_g; g.ln_b = L.ln1_b; g.ln_eps = c.ln_eps;
            g.ln_x_out = pend_bias ? alt : nullptr;
            if ((rc = launch_gemm_decode_ln(g, st))) return rc;
            if (pend_bias) { float* t = cur; cur = alt; alt = t; }
        } else if ((rc = launch_gemm(g, prec, prefill, st))) return rc;
        pend_bias = nullptr;

        AttnArgs at{};
        at.qbuf = w.qbuf; at.kcache = g.kcache; at.vcache = g.vcache; at.pad = pad; at.pos_ptr = pos_ptr; at.pos_shift = pos_shift;
        at.row_map = beam ? w.row_map[0] : nullptr; at.row_map_alt = beam ? w.row_map[1] : nullptr; at.step_ptr = beam ? w.state : nullptr;
        at.out = w.attn; at.nseq = nseq; at.H = c.heads; at.nq = S; at.Tmax = Tmax; at.D = D; at.seq_mul = seq_mul; at.seq_map = seq_map;
        if ((rc = launch_attention(at, prec, st))) return rc;

        GemmArgs p{};
        p.A = w.attn; p.lda = D; p.Wp = L.w_proj; p.bias = L.b_proj; p.M = rows; p.N = D; p.K = D; p.D = D;
        if (prefill) { p.nsplit = 1; p.epi = EPI_RESIDUAL; p.out_f32 = w.x; p.ldo = D; }
        else { p.nsplit = 4; p.epi = EPI_PARTIAL; p.partial = w.partial; }
        if ((rc = launch_gemm(p, prec, prefill, st))) return rc;

        LnArgs ln2 = ln;
        ln2.x = cur;
        ln2.partial = prefill ? nullptr : w.partial; ln2.bias_prev = prefill ? nullptr : L.b_proj;
        ln2.g1 = L.ln2_g; ln2.b1 = L.ln2_b;
        if (!fuse_ln && (rc = launch_ln(ln2, prec, st))) return rc;

        GemmArgs f{};
        f.A = w.hbuf; f.lda = D; f.Wp = L.w_fc; f.bias = L.b_fc; f.M = rows; f.N = 4 * D; f.K = D; f.nsplit = 1; f.epi = EPI_GELU_ACT;
        f.out_act = w.fc; f.ldo = 4 * D; f.D = D;
        if (fuse_ln) {                                             // (decode: the proj GEMM left partials, so the residual moves to `alt`)
            f.ln_x = cur; f.ln_partial = w.partial; f.ln_bias_prev = L.b_proj; f.ln_g = L.ln2_g; f.ln_b = L.ln2_b; f.ln_eps = c.ln_eps;
            f.ln_x_out = alt;
            if ((rc = launch_gemm_decode_ln(f, st))) return rc;
            float* t = cur; cur = alt; alt = t;
        } else if ((rc = launch_gemm(f, prec, prefill, st))) return rc;

        GemmArgs f2{};
        f2.A = w.fc; f2.lda = 4 * D; f2.Wp = L.w_fc2; f2.bias = L.b_fc2; f2.M = rows; f2.N = D; f2.K = 4 * D; f2.D = D;
        if (prefill) { f2.nsplit = 1; f2.epi = EPI_RESIDUAL; f2.out_f32 = w.x; f2.ldo = D; }
        else { f2.nsplit = 4; f2.epi = EPI_PARTIAL; f2.partial = w.partial; pend_bias = L.b_fc2; }
        if ((rc = launch_gemm(f2, prec, prefill, st))) return rc;
    }
    *pending = pend_bias != nullptr;   // decode: the last FC2's partials are reduced by the caller's final LayerNorm
    if (x_cur) *x_cur = cur;
    else if (cur != w.x) { itts_set_error("run_layers: the fused-LayerNorm path needs the caller to take x_cur"); return ITTS_ERR_STATE; }
    return ITTS_OK;
}

// final norm(s) + head: rows_out rows, input row r*mul+add
static int run_head(itts_gpt* h, const GptWs& w, int nseq, int mul, int add, bool pending, hipStream_t st, float* x = nullptr) {
    const itts_gpt_config& c = h->cfg;
    const int D = c.model_dim, prec = c.precision;
    int rc;
    LnArgs ln{};
    ln.x = x ? x : w.x;                  // (the residual stream: w.x2 after a decode pass on the LayerNorm-fused GEMMs)
    ln.partial = pending ? w.partial : nullptr; ln.nsplit = 4; ln.bias_prev = pending ? h->layers.back().b_fc2 : nullptr;
    ln.g1 = h->lnf_g; ln.b1 = h->lnf_b; ln.g2 = h->fn_g; ln.b2 = h->fn_b; ln.out = w.hlast; ln.out_f32 = 0;
    ln.rows = nseq; ln.D = D; ln.in_row_mul = mul; ln.in_row_add = add; ln.eps = c.ln_eps;
    if ((rc = launch_ln(ln, prec, st))) return rc;
    GemmArgs g{};
    g.A = w.hlast; g.lda = D; g.Wp = h->w_head; g.bias = h->b_head; g.M = nseq; g.N = c.vocab; g.K = D; g.nsplit = 1;
    g.epi = EPI_STORE_F32; g.out_f32 = w.logits; g.ldo = c.vocab; g.D = D;
    return launch_gemm(g, prec, false, st);
}

// rows: dense rows of this launch; n_utts: utterances of the call (stride of the uniform stream); mapped: the batch is compacted
static SampleArgs make_sample(itts_gpt* h, const GptWs& w, const itts_gen_params& gp, int nseq, long long* tokens,
                              const double* uniforms, int n_utts = 0, bool mapped = false) {
    const itts_gpt_config& c = h->cfg;
    SampleArgs s{};
    s.logits = w.logits; s.seen = w.seen; s.finished = w.finished; s.tokens = tokens; s.step_ptr = w.state;
    s.uniforms = uniforms; s.seed = gp.seed; s.B = nseq; s.V = c.vocab; s.max_new = gp.max_new_tokens;
    s.do_sample = gp.do_sample; s.top_k = gp.top_k; s.min_keep = gp.min_tokens_to_keep < 1 ? 1 : gp.min_tokens_to_keep;
    s.top_p = gp.top_p; s.temperature = gp.temperature; s.rep_penalty = gp.repetition_penalty;
    s.typical_mass = gp.typical_mass;
    s.stop_token = c.stop_mel_token; s.mel_emb = h->mel_emb; s.mel_pos = h->mel_pos; s.x_next = w.x; s.D = c.model_dim;
    s.pos_offset = gp.pos_offset; s.n_mel_pos = c.n_mel_pos;
    s.seed_ptr = (const unsigned long long*)(w.state + 4);
    s.row_slot = mapped ? w.slot_map : nullptr;
    s.uniforms_stride = n_utts > 0 ? n_utts : nseq;
    s.row_limit = (h->row_limits && h->row_limits_n == s.uniforms_stride) ? h->row_limits : nullptr;
    s.row_step0 = w.row_step0;
    return s;
}

// rows: the dense rows this step runs (= n_utts until finished utterances have been compacted away)
// shifted: a row has been admitted into the batch (itts_gpt_admit_rows) -- the QKV epilogues and the attention read the per-row position shifts.
// Until then they are all zero and the step does not load them (the load sits behind the slot-map load in the QKV epilogue: ~1 % of a
// 1-8-row token step, profiles/r06e vs r05p).
static int decode_step(itts_gpt* h, const GptWs& w, const itts_gen_params& gp, int rows, int n_utts, bool mapped, int Tmax, long long* tokens,
                       const double* uniforms, hipStream_t st, bool shifted) {
    bool pending = false;
    float* xc = w.x;
    int rc = run_layers(h, w, rows, 1, Tmax, false, w.state + 1, w.pad, &pending, st, false, 1, mapped ? w.slot_map : nullptr, &xc,
                        shifted ? w.row_shift : nullptr);
    if (rc) return rc;
    if ((rc = run_head(h, w, rows, 1, 0, pending, st, xc))) return rc;
    SampleArgs s = make_sample(h, w, gp, rows, tokens, uniforms, n_utts, mapped);
    s.adv_state = w.state;                      // the sample kernel's last block advances step / pos
    return launch_sample(s, st);
}

// step_limit: stop after that many generated tokens (<= max_new_tokens) -- the chunked form used for streaming; resume: continue
// the decode loop of an earlier chunk call from the device state left in the same workspace (no prefill, no state init).
static int gpt_generate_impl(itts_gpt* h, const float* prefix_embeds, const int32_t* pad_lens, int nseq, int S,
                             const itts_gen_params* gpp, const int32_t* penalty_ids, int n_penalty_ids,
                             const double* uniforms, int64_t* codes_out, int32_t* n_steps_out, void* workspace,
                             size_t workspace_bytes, int use_graph, void* caller_stream, int step_limit, bool resume) {
    if (!h || (!prefix_embeds && !resume) || !gpp || !codes_out || !n_steps_out || !workspace) { itts_set_error("gpt_generate: null pointer"); return ITTS_ERR_ARG; }
    if (!h->finalized) { itts_set_error("gpt_generate: call itts_gpt_finalize first"); return ITTS_ERR_STATE; }
    ItDevGuard dg(h->device);
    if (int rcd = check_same_device(h, resume ? (const void*)codes_out : (const void*)prefix_embeds, workspace, "gpt_generate")) return rcd;
    const itts_gpt_config& c = h->cfg;
    const itts_gen_params gp = *gpp;
    if (nseq <= 0 || S <= 0 || gp.max_new_tokens <= 0) { itts_set_error("gpt_generate: nseq, S, max_new_tokens must be > 0"); return ITTS_ERR_ARG; }
    // The first call's step counter is bounded by max_new_tokens.  A resumed loop's is not: every row is bounded by its OWN step (the sampler emits
    // the stop token from the row's step max_new_tokens / its row limit on and stores nothing; a finished row's K / V stay inside its cache row), so a
    // session runs for as long as the caller keeps admitting rows (itts_gpt_admit_rows).
    const bool chunked = step_limit > 0;                               // itts_gpt_generate_chunk (itts_gpt_generate passes 0)
    if (step_limit < 1 || (step_limit > gp.max_new_tokens && !resume)) step_limit = gp.max_new_tokens;
    if (resume && (h->chunk_steps < 1 || h->chunk_nseq != nseq || h->chunk_S != S || h->chunk_max_new != gp.max_new_tokens ||
                   h->chunk_ws != workspace)) {
        itts_set_error("gpt_generate_chunk: resume without a matching first chunk (same workspace, nseq, S, max_new_tokens)");
        return ITTS_ERR_STATE;
    }
    if (gp.num_beams != 1) { itts_set_error("gpt_generate: num_beams=%d not supported by the device loop yet (use 1)", gp.num_beams); return ITTS_ERR_ARG; }
    if (gp.max_new_tokens + gp.pos_offset > c.n_mel_pos + 1) {
        itts_set_error("gpt_generate: max_new_tokens=%d exceeds the mel position table (%d rows)", gp.max_new_tokens, c.n_mel_pos);
        return ITTS_ERR_ARG;
    }
    if (n_penalty_ids < 0 || n_penalty_ids > 16) { itts_set_error("gpt_generate: at most 16 initial penalty ids"); return ITTS_ERR_ARG; }
    if ((size_t)nseq * c.heads > 2147483647u / 4 || S > 65535) { itts_set_error("gpt_generate: batch too large"); return ITTS_ERR_ARG; }
    const int Sb = s_bucket(S), Tmax = Sb + gp.max_new_tokens;      // cache stride / carve shape (bucketed prompt length)
    const GptWs w0 = carve(c, nullptr, nseq, Sb, Tmax);
    if (workspace_bytes < w0.total) { itts_set_error("gpt_generate: workspace too small (%zu < %zu)", workspace_bytes, w0.total); return ITTS_ERR_ARG; }
    char* base = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    const GptWs w = carve(c, base, nseq, Sb, Tmax);
    hipStream_t st = h->stream, cs = (hipStream_t)caller_stream;
    long long* tokens = (long long*)codes_out;
    int rc;

    if (h->fin_cap < nseq) {
        if (h->host_fin) (void)hipHostFree(h->host_fin);
        HIP_TRY(hipHostMalloc((void**)&h->host_fin, (size_t)nseq, hipHostMallocDefault));
        h->fin_cap = nseq;
    }
    // order after the caller's stream
    HIP_TRY(hipEventRecord(h->ev_in, cs));
    HIP_TRY(hipStreamWaitEvent(st, h->ev_in, 0));

    int steps = resume ? h->chunk_steps : 1;
    if (!resume) {
    // ---- state init ----
    HIP_TRY(hipMemsetAsync(w.seen, 0, (size_t)nseq * c.vocab, st));
    HIP_TRY(hipMemsetAsync(w.finished, 0, nseq, st));
    HIP_TRY(hipMemsetAsync(w.row_step0, 0, (size_t)nseq * 4, st));
    HIP_TRY(hipMemsetAsync(w.row_shift, 0, (size_t)nseq * 4, st));
    if (pad_lens) HIP_TRY(hipMemcpyAsync(w.pad, pad_lens, (size_t)nseq * 4, hipMemcpyDeviceToDevice, st));
    else HIP_TRY(hipMemsetAsync(w.pad, 0, (size_t)nseq * 4, st));
    if (n_penalty_ids > 0) {
        HIP_TRY(hipMemcpyAsync(w.pen_ids, penalty_ids, (size_t)n_penalty_ids * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(mark_seen_kernel, dim3(nseq), dim3(64), 0, st, w.seen, w.pen_ids, n_penalty_ids, c.vocab);
    }
    {
        const size_t n = (size_t)nseq * gp.max_new_tokens;
        hipLaunchKernelGGL(fill_i64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, tokens, (long long)c.stop_mel_token, n);
    }
    hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(1), 0, st, w.state, 0, 0);
    hipLaunchKernelGGL(set_seed_kernel, dim3(1), dim3(1), 0, st, w.state, (unsigned long long)gp.seed);
    HIP_TRY(hipMemcpyAsync(w.x, prefix_embeds, (size_t)nseq * S * c.model_dim * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipGetLastError());

    // ---- prefill: all S positions, logits of the last one, first token ----
    HIP_TRY(hipEventRecord(h->ev_t0, st));
    bool pending = false;
    rc = run_layers(h, w, nseq, S, Tmax, true, w.state + 1, w.pad, &pending, st);
    if (rc) return rc;
    if ((rc = run_head(h, w, nseq, S, S - 1, pending, st))) return rc;
    hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(1), 0, st, w.state, 0, S);
    {
        SampleArgs s = make_sample(h, w, gp, nseq, tokens, uniforms, nseq, false);
        if ((rc = launch_sample(s, st))) return rc;
    }
    hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(1), 0, st, w.state, 1, S);
    } else {
        HIP_TRY(hipEventRecord(h->ev_t0, st));
    }
    HIP_TRY(hipEventRecord(h->ev_t1, st));

    // ---- decode loop ----
    // The running batch starts as the nseq utterances in order.  Every `check_every` steps the finished flags come to the host; when
    // enough utterances have finished, the survivors are compacted to the front (their cache rows stay where they are: the QKV
    // epilogue, the attention kernel and the sampler find an utterance's state through the slot map) and the step is replayed from
    // the graph of the smaller batch -- the step's cost follows the live rows in steps of `compact_gran`.
    if (h->map_cap < nseq) {
        if (h->host_map) (void)hipHostFree(h->host_map);
        HIP_TRY(hipHostMalloc((void**)&h->host_map, (size_t)2 * nseq * sizeof(int), hipHostMallocDefault));
        h->map_cap = nseq;
    }
    if (!resume) {
        h->chunk_admitted = false;
        h->cur_slots.resize(nseq);
        for (int i = 0; i < nseq; ++i) h->cur_slots[i] = i;
        h->cur_mapped = false;
        h->last_row_steps = 0;
        h->last_compactions = 0;
    }
    const bool env_off = itts_opt(OPT_GPT_COMPACT) == 0;
    const bool compact = h->compact && !env_off;
    hipGraphExec_t exec = nullptr;
    bool graph_ok = false;
    auto get_graph = [&](int rows, bool mapped) -> int {
        exec = nullptr; graph_ok = false;
        if (!(use_graph && gp.max_new_tokens > 1)) return ITTS_OK;
        itts_gpt::GraphEntry key{};
        key.base = base; key.tokens = tokens; key.uniforms = uniforms; key.nseq = rows; key.nb = 1; key.Sb = Sb; key.Tmax = Tmax; key.gp = gp; key.gp.seed = 0;          // the seed lives in device memory
        key.opt_epoch = itts_opt_epoch();
        key.S = nseq;                                                  // utterances of the call (uniform stride, limits)
        key.aux0 = mapped ? (const void*)w.slot_map : nullptr;
        key.aux1 = (h->row_limits && h->row_limits_n == nseq) ? (const void*)h->row_limits : nullptr;
        key.aux2 = h->chunk_admitted ? (const void*)w.row_shift : nullptr;
        exec = graph_lookup(h, key);
        graph_ok = exec != nullptr;
        if (graph_ok) return ITTS_OK;
        // capture one decode step (all step-varying state lives in device memory); kept in the handle for later calls
        hipGraph_t graph = nullptr;
        int rcc = ITTS_OK;
        hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        if (e == hipSuccess) {
            rcc = decode_step(h, w, gp, rows, nseq, mapped, Tmax, tokens, uniforms, st, h->chunk_admitted);
            e = hipStreamEndCapture(st, &graph);
            if (rcc == ITTS_OK && e == hipSuccess && graph) {
                e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
                graph_ok = (e == hipSuccess && exec);
            }
            if (graph) (void)hipGraphDestroy(graph);
        }
        if (!graph_ok) {
            (void)hipGetLastError();
            itts_set_error("gpt_generate: hipGraph capture failed (%s); rerun with use_graph=0", hipGetErrorString(e));
            return ITTS_ERR_HIP;
        }
        graph_insert(h, key, exec);
        return ITTS_OK;
    };
    int rows = (int)h->cur_slots.size();
    bool mapped = h->cur_mapped;
    if ((rc = get_graph(rows, mapped))) return rc;
    const int check_every = 8;
    while (steps < step_limit) {
        if (graph_ok) { HIP_TRY(hipGraphLaunch(exec, st)); }
        else if ((rc = decode_step(h, w, gp, rows, nseq, mapped, Tmax, tokens, uniforms, st, h->chunk_admitted))) return rc;
        ++steps;
        h->last_row_steps += rows;
        if (steps % check_every == 0 && steps < step_limit) {
            HIP_TRY(hipMemcpyAsync(h->host_fin, w.finished, nseq, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            int live = 0;
            for (int i = 0; i < nseq; ++i) live += h->host_fin[i] ? 0 : 1;
            if (live == 0) break;
            if (chunked && h->chunk_return_finished > 0 && nseq - live >= h->chunk_return_finished) break;        // slots to refill
            const int g = h->compact_gran < 1 ? 1 : h->compact_gran;
            const int want = ((live + g - 1) / g) * g;
            if (compact && want < rows) {
                // new dense order: the live rows in their current order, then finished rows up to the bucket size (they keep emitting stop)
                int* nm = h->host_map;
                int* src = h->host_map + nseq;
                int k = 0;
                for (int i = 0; i < rows; ++i)
                    if (!h->host_fin[h->cur_slots[i]]) { nm[k] = h->cur_slots[i]; src[k] = i; ++k; }
                for (int i = 0; i < rows && k < want; ++i)
                    if (h->host_fin[h->cur_slots[i]]) { nm[k] = h->cur_slots[i]; src[k] = i; ++k; }
                HIP_TRY(hipMemcpyAsync(w.slot_map, nm, (size_t)want * sizeof(int), hipMemcpyHostToDevice, st));
                HIP_TRY(hipMemcpyAsync(w.gather_src, src, (size_t)want * sizeof(int), hipMemcpyHostToDevice, st));
                hipLaunchKernelGGL(gather_rows_kernel, dim3(want), dim3(256), 0, st, w.x, w.gather_src, w.qbuf, c.model_dim);
                hipLaunchKernelGGL(copy_rows_kernel, dim3(want), dim3(256), 0, st, w.qbuf, w.x, c.model_dim);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipStreamSynchronize(st));                      // the pinned staging buffer is reused by the next compaction
                h->cur_slots.assign(nm, nm + want);
                rows = want;
                mapped = true;
                h->cur_mapped = true;
                ++h->last_compactions;
                if ((rc = get_graph(rows, mapped))) return rc;
            }
        }
    }
    HIP_TRY(hipEventRecord(h->ev_t2, st));
    HIP_TRY(hipEventRecord(h->ev_out, st));
    HIP_TRY(hipStreamWaitEvent(cs, h->ev_out, 0));
    HIP_TRY(hipStreamSynchronize(st));
    (void)hipEventElapsedTime(&h->last_prefill_ms, h->ev_t0, h->ev_t1);
    (void)hipEventElapsedTime(&h->last_decode_ms, h->ev_t1, h->ev_t2);
    h->last_steps = steps;
    h->chunk_steps = steps; h->chunk_nseq = nseq; h->chunk_S = S; h->chunk_max_new = gp.max_new_tokens; h->chunk_ws = workspace;
    *n_steps_out = steps;
    return ITTS_OK;
}

extern "C" int itts_gpt_generate(itts_gpt* h, const float* prefix_embeds, const int32_t* pad_lens, int nseq, int S,
                                 const itts_gen_params* gpp, const int32_t* penalty_ids, int n_penalty_ids,
                                 const double* uniforms, int64_t* codes_out, int32_t* n_steps_out, void* workspace,
                                 size_t workspace_bytes, int use_graph, void* caller_stream) {
    return gpt_generate_impl(h, prefix_embeds, pad_lens, nseq, S, gpp, penalty_ids, n_penalty_ids, uniforms, codes_out, n_steps_out,
                             workspace, workspace_bytes, use_graph, caller_stream, 0, false);
}

extern "C" int itts_gpt_generate_chunk(itts_gpt* h, const float* prefix_embeds, const int32_t* pad_lens, int nseq, int S,
                                       const itts_gen_params* gpp, const int32_t* penalty_ids, int n_penalty_ids,
                                       const double* uniforms, int64_t* codes_out, int32_t step_limit, int32_t* n_steps_out,
                                       void* workspace, size_t workspace_bytes, int use_graph, void* caller_stream) {
    return gpt_generate_impl(h, prefix_embeds, pad_lens, nseq, S, gpp, penalty_ids, n_penalty_ids, uniforms, codes_out, n_steps_out,
                             workspace, workspace_bytes, use_graph, caller_stream, step_limit, prefix_embeds == nullptr);
}

// ---- admission of new utterances into a running decode batch -------------------------------------------------------------------------
// Design reference: the reference's serving path keeps a decode batch running and puts a newly arrived request into a free slot while the other
// rows keep generating (backends/trt/serving/triton_server.py:96-305, backends/trt/pipeline/pipeline.py:459-548 on TRT-LLM's in-flight batching).
// Here: between two itts_gpt_generate_chunk calls the loop is suspended with its state on the device; the batch's position counter stands at
// pos = S + steps - 1.  A new utterance takes the cache row of a FINISHED one and keeps its OWN positions: its prompt (S_new positions, at most the
// session's prompt bucket) is prefilled on an admission workspace of its own (K / V to a scratch cache, copied to positions 0 .. S_new - 1 of the
// slot's cache rows), row_shift[slot] = pos - S_new lets the QKV epilogue and the attention kernel find the row's own position under the shared
// counter, its first token is sampled into column 0 of the slot's code row (refilled with the stop token), and row_step0[slot] = steps - 1 makes
// the sampler index the row's token column, position embedding, uniform / RNG stream and token limit by the row's OWN step.  Nothing of the row
// depends on the step it joined at -- the attention's key streams are laid out from the row's first valid key -- so the admitted row generates,
// bit for bit, the ids it generates alone (tests/test_gpu_admission.py), and a session is not bounded by the mel position table: only each row is.
__global__ void admit_state_kernel(const int* __restrict__ slots, const int* __restrict__ pad_new, unsigned char* seen, unsigned char* finished, int* pad,
                                   int* row_step0, int* row_shift, const int* __restrict__ pen_ids, int n_ids, int V, int step0, int shift,
                                   long long* tokens, int max_new, long long stop_token, int* row_limit, const int* __restrict__ limits_new) {
    const int i = blockIdx.x, u = slots[i];
    unsigned char* sr = seen + (size_t)u * V;
    for (int c = threadIdx.x; c < V; c += blockDim.x) sr[c] = 0;
    long long* tr = tokens + (size_t)u * max_new;
    for (int c = threadIdx.x; c < max_new; c += blockDim.x) tr[c] = stop_token;
    __syncthreads();
    if ((int)threadIdx.x < n_ids) {
        const int id = pen_ids[threadIdx.x];
        if (id >= 0 && id < V) sr[id] = 1;
    }
    if (threadIdx.x == 0) {
        finished[u] = 0; pad[u] = pad_new[i]; row_step0[u] = step0; row_shift[u] = shift;
        if (row_limit && limits_new) row_limit[u] = limits_new[i];
    }
}
// K / V of positions [0, n_pos) of the admission cache's row i -> the running batch's cache row slots[i] (both [L][rows][H][T][64], own T strides)
__global__ void copy_kv_rows_kernel(const char* __restrict__ src, char* __restrict__ dst, const int* __restrict__ slots, int H, int T_src, int T_dst,
                                    int n_pos, int row_bytes, size_t src_layer, size_t dst_layer) {
    const int i = blockIdx.x / H, hd = blockIdx.x - i * H, l = blockIdx.y;
    const char* s = src + (size_t)l * src_layer + ((size_t)i * H + hd) * T_src * row_bytes;
    char* d = dst + (size_t)l * dst_layer + ((size_t)slots[i] * H + hd) * T_dst * row_bytes;
    const size_t n16 = (size_t)n_pos * row_bytes / 16;
    for (size_t c = threadIdx.x; c < n16; c += blockDim.x) ((uint4*)d)[c] = ((const uint4*)s)[c];
}
// dense rows back to utterance order: tmp[d] = x[src[d]] (zero where the utterance is not in the running batch), then the admitted rows
__global__ void regather_rows_kernel(const float* __restrict__ x, const int* __restrict__ src, float* __restrict__ tmp, int D) {
    const int d = blockIdx.x, j = src[d];
    for (int c = threadIdx.x; c < D; c += blockDim.x) tmp[(size_t)d * D + c] = j >= 0 ? x[(size_t)j * D + c] : 0.f;
}
__global__ void place_rows_kernel(const float* __restrict__ xa, const int* __restrict__ slots, float* __restrict__ x, int D) {
    const int i = blockIdx.x;
    for (int c = threadIdx.x; c < D; c += blockDim.x) x[(size_t)slots[i] * D + c] = xa[(size_t)i * D + c];
}

extern "C" size_t itts_gpt_admit_workspace_bytes(const itts_gpt* h, int n_new, int S_new) {
    if (!h || n_new <= 0 || S_new <= 0) return 0;
    const int Sb = s_bucket(S_new);
    return carve(h->cfg, nullptr, n_new, Sb, Sb + 8).total + a256((size_t)n_new * 8) + 512;
}

extern "C" int itts_gpt_admit_rows(itts_gpt* h, const float* prefix_embeds, const int32_t* pad_lens, const int32_t* slots, int n_new, int S_new,
                                   const int32_t* row_limits_new, const itts_gen_params* gpp, const int32_t* penalty_ids, int n_penalty_ids,
                                   const double* uniforms, int64_t* codes_out, void* workspace, size_t workspace_bytes, void* admit_workspace,
                                   size_t admit_bytes, void* caller_stream) {
    if (!h || !prefix_embeds || !pad_lens || !slots || !gpp || !codes_out || !workspace || !admit_workspace) { itts_set_error("gpt_admit_rows: null pointer"); return ITTS_ERR_ARG; }
    if (!h->finalized) { itts_set_error("gpt_admit_rows: call itts_gpt_finalize first"); return ITTS_ERR_STATE; }
    ItDevGuard dg(h->device);
    if (int rcd = check_same_device(h, prefix_embeds, workspace, "gpt_admit_rows")) return rcd;
    const itts_gpt_config& c = h->cfg;
    const itts_gen_params gp = *gpp;
    const int nseq = h->chunk_nseq, S = h->chunk_S, k = h->chunk_steps;
    if (k < 1 || h->chunk_ws != workspace || gp.max_new_tokens != h->chunk_max_new || gp.num_beams != 1) {
        itts_set_error("gpt_admit_rows: no suspended itts_gpt_generate_chunk loop on this workspace with these parameters");
        return ITTS_ERR_STATE;
    }
    const int Sb = s_bucket(S), Tmax = Sb + gp.max_new_tokens;
    if (S_new < 1 || S_new > Sb) {      // a cache row holds Sb prompt positions + max_new_tokens generated ones
        itts_set_error("gpt_admit_rows: S_new = %d outside 1 .. %d (the session's prompt bucket)", S_new, Sb);
        return ITTS_ERR_ARG;
    }
    const bool limited = h->row_limits && h->row_limits_n == nseq;
    if (limited != (row_limits_new != nullptr)) {
        itts_set_error("gpt_admit_rows: row_limits_new must be given exactly when per-utterance limits are installed (itts_gpt_set_row_limits, %d entries)", nseq);
        return ITTS_ERR_ARG;
    }
    if (n_new < 1 || n_new > nseq) { itts_set_error("gpt_admit_rows: n_new = %d outside 1 .. %d", n_new, nseq); return ITTS_ERR_ARG; }
    if (n_penalty_ids < 0 || n_penalty_ids > 16) { itts_set_error("gpt_admit_rows: at most 16 initial penalty ids"); return ITTS_ERR_ARG; }
    const GptWs w0 = carve(c, nullptr, nseq, Sb, Tmax);
    if (workspace_bytes < w0.total) { itts_set_error("gpt_admit_rows: workspace too small"); return ITTS_ERR_ARG; }
    const GptWs w = carve(c, (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255), nseq, Sb, Tmax);
    const int Sba = s_bucket(S_new), Ta = Sba + 8;
    const GptWs wa0 = carve(c, nullptr, n_new, Sba, Ta);
    if (admit_bytes < wa0.total + a256((size_t)n_new * 8) + 256) { itts_set_error("gpt_admit_rows: admission workspace too small (%zu < %zu)", admit_bytes, wa0.total + a256((size_t)n_new * 8) + 256); return ITTS_ERR_ARG; }
    char* abase = (char*)(((uintptr_t)admit_workspace + 255) & ~(uintptr_t)255);
    const GptWs wa = carve(c, abase, n_new, Sba, Ta);
    int* slots_dev = (int*)(abase + wa0.total);
    int* limits_dev = slots_dev + n_new;
    hipStream_t st = h->stream, cs = (hipStream_t)caller_stream;
    HIP_TRY(hipEventRecord(h->ev_in, cs));
    HIP_TRY(hipStreamWaitEvent(st, h->ev_in, 0));
    // the slots must be distinct finished utterances of the running batch
    if (h->fin_cap < nseq) { itts_set_error("gpt_admit_rows: no finished-flag buffer (run a chunk first)"); return ITTS_ERR_STATE; }
    HIP_TRY(hipMemcpyAsync(h->host_fin, w.finished, nseq, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    std::vector<char> taken(nseq, 0);
    for (int i = 0; i < n_new; ++i) {
        const int u = slots[i];
        if (u < 0 || u >= nseq || taken[u] || !h->host_fin[u]) {
            itts_set_error("gpt_admit_rows: slot %d (entry %d) is out of range, repeated or still generating", u, i);
            return ITTS_ERR_ARG;
        }
        taken[u] = 1;
    }
    int rc;
    HIP_TRY(hipMemcpyAsync(slots_dev, slots, (size_t)n_new * 4, hipMemcpyHostToDevice, st));
    if (limited) HIP_TRY(hipMemcpyAsync(limits_dev, row_limits_new, (size_t)n_new * 4, hipMemcpyHostToDevice, st));
    if (n_penalty_ids > 0) HIP_TRY(hipMemcpyAsync(wa.pen_ids, penalty_ids, (size_t)n_penalty_ids * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(wa.pad, pad_lens, (size_t)n_new * 4, hipMemcpyDeviceToDevice, st));
    // (the limits are written here, after every check above has passed: a rejected call leaves the running rows' caps alone)
    hipLaunchKernelGGL(admit_state_kernel, dim3(n_new), dim3(256), 0, st, slots_dev, wa.pad, w.seen, w.finished, w.pad, w.row_step0, w.row_shift,
                       wa.pen_ids, n_penalty_ids, c.vocab, k - 1, S + k - 1 - S_new, (long long*)codes_out, gp.max_new_tokens,
                       (long long)c.stop_mel_token, limited ? (int*)h->row_limits : nullptr, limited ? limits_dev : nullptr);
    // prefill of the new rows on the admission workspace: all S_new positions, logits of the last one
    hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(1), 0, st, wa.state, 0, 0);
    HIP_TRY(hipMemcpyAsync(wa.x, prefix_embeds, (size_t)n_new * S_new * c.model_dim * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipGetLastError());
    bool pending = false;
    if ((rc = run_layers(h, wa, n_new, S_new, Ta, true, wa.state + 1, wa.pad, &pending, st))) return rc;
    if ((rc = run_head(h, wa, n_new, S_new, S_new - 1, pending, st))) return rc;
    // first token of every new row: sampled with the running batch's per-utterance state (seen set, finished flag, code row, uniform / RNG stream),
    // into column 0 of the slot's code row: the row's own step is 0
    hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(1), 0, st, wa.state, k - 1, S_new);
    {
        SampleArgs s = make_sample(h, w, gp, n_new, (long long*)codes_out, uniforms, nseq, false);
        s.logits = wa.logits; s.x_next = wa.x; s.step_ptr = wa.state; s.row_slot = slots_dev; s.adv_state = nullptr;
        if ((rc = launch_sample(s, st))) return rc;
    }
    // K / V of the prompt into the slots' cache rows
    {
        const int rb = 64 * (c.precision == PREC_BF16 ? 2 : 4);
        hipLaunchKernelGGL(copy_kv_rows_kernel, dim3(n_new * c.heads, c.layers), dim3(256), 0, st, wa.kc, w.kc, slots_dev, c.heads, Ta, Tmax, S_new, rb,
                           wa.layer_cache_bytes, w.layer_cache_bytes);
        hipLaunchKernelGGL(copy_kv_rows_kernel, dim3(n_new * c.heads, c.layers), dim3(256), 0, st, wa.vc, w.vc, slots_dev, c.heads, Ta, Tmax, S_new, rb,
                           wa.layer_cache_bytes, w.layer_cache_bytes);
    }
    // the running batch back in utterance order (uncompacted), the admitted rows' next-step inputs in their slots
    {
        if (h->map_cap < nseq) { itts_set_error("gpt_admit_rows: no row-map buffer (run a chunk first)"); return ITTS_ERR_STATE; }
        int* src = h->host_map;
        for (int d = 0; d < nseq; ++d) src[d] = -1;
        for (int j = 0; j < (int)h->cur_slots.size(); ++j) src[h->cur_slots[j]] = j;
        HIP_TRY(hipMemcpyAsync(w.gather_src, src, (size_t)nseq * sizeof(int), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(regather_rows_kernel, dim3(nseq), dim3(256), 0, st, w.x, w.gather_src, w.qbuf, c.model_dim);
        hipLaunchKernelGGL(copy_rows_kernel, dim3(nseq), dim3(256), 0, st, w.qbuf, w.x, c.model_dim);
        hipLaunchKernelGGL(place_rows_kernel, dim3(n_new), dim3(256), 0, st, wa.x, slots_dev, w.x, c.model_dim);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(h->ev_out, st));
    HIP_TRY(hipStreamWaitEvent(cs, h->ev_out, 0));
    HIP_TRY(hipStreamSynchronize(st));
    h->cur_slots.resize(nseq);
    for (int i = 0; i < nseq; ++i) h->cur_slots[i] = i;
    h->cur_mapped = false;
    h->chunk_admitted = true;
    return ITTS_OK;
}

// ---- beam search / beam-sample ------------------------------------------------------------------------------------
// The beams of an utterance share ONE copy of the prompt's K/V (cache row b*nb, written by a prefill over the B unique
// prompts): prompt positions map there, generated positions to the row's own cache row.
__global__ void beam_init_kernel(int* row_map0, float* beam_scores, float* worst, int* n_hyps, unsigned char* done, int nseq, int nb,
                                 int Tmax, int S) {
    const int i = blockIdx.x;
    for (int t = threadIdx.x; t < Tmax; t += blockDim.x) row_map0[(size_t)i * Tmax + t] = t < S ? (i / nb) * nb : i;
    if (threadIdx.x == 0) {
        beam_scores[i] = (i % nb == 0) ? 0.f : -1e9f;             // generation_utils.py:3408-3410
        if (i % nb == 0) { worst[i / nb] = 1e9f; n_hyps[i / nb] = 0; done[i / nb] = 0; }
    }
}

static BeamArgs make_beam(itts_gpt* h, const GptWs& w, const itts_gen_params& gp, int B, int nb, int S, int Tmax, const double* uniforms) {
    const itts_gpt_config& c = h->cfg;
    BeamArgs a{};
    a.logits = w.logits; a.seen[0] = w.seen; a.seen[1] = w.seen2; a.row_map[0] = w.row_map[0]; a.row_map[1] = w.row_map[1];
    a.beam_scores = w.beam_scores; a.next_scores = w.next_scores; a.next_tokens = w.next_tokens; a.next_indices = w.next_indices;
    a.hyps = w.hyps; a.n_hyps = w.n_hyps; a.worst = w.worst; a.done = w.done; a.hist_tok = w.hist_tok; a.hist_par = w.hist_par;
    a.surv_idx = w.surv_idx; a.surv_val = w.surv_val; a.surv_n = w.surv_n;
    a.step_ptr = w.state; a.uniforms = uniforms; a.seed = gp.seed; a.B = B; a.nb = nb; a.V = c.vocab; a.max_new = gp.max_new_tokens;
    a.Tmax = Tmax; a.S = S; a.do_sample = gp.do_sample; a.top_k = gp.top_k;
    a.min_keep = gp.min_tokens_to_keep < 1 ? 1 : gp.min_tokens_to_keep;
    a.top_p = gp.top_p; a.temperature = gp.temperature; a.rep_penalty = gp.repetition_penalty; a.length_penalty = gp.length_penalty;
    a.typical_mass = gp.typical_mass;
    a.stop_token = c.stop_mel_token; a.mel_emb = h->mel_emb; a.mel_pos = h->mel_pos; a.x_next = w.x; a.D = c.model_dim;
    a.pos_offset = gp.pos_offset; a.n_mel_pos = c.n_mel_pos;
    a.seed_ptr = (const unsigned long long*)(w.state + 4);
    return a;
}

static int decode_step_beam(itts_gpt* h, const GptWs& w, const BeamArgs& ba, int nseq, int Tmax, hipStream_t st) {
    bool pending = false;
    float* xc = w.x;
    int rc = run_layers(h, w, nseq, 1, Tmax, false, w.state + 1, w.pad, &pending, st, true, 1, nullptr, &xc);
    if (rc) return rc;
    if ((rc = run_head(h, w, nseq, 1, 0, pending, st, xc))) return rc;
    if ((rc = launch_beam_step(ba, st))) return rc;
    BeamArgs bb = ba;
    bb.adv_state = w.state;                     // the apply kernel's last block advances step / pos
    return launch_beam_apply(bb, st);
}

extern "C" int itts_gpt_generate_beam(itts_gpt* h, const float* prefix_embeds, const int32_t* pad_lens, int n_utts, int num_beams,
                                      int S, const itts_gen_params* gpp, const int32_t* penalty_ids, int n_penalty_ids,
                                      const double* uniforms, int32_t* hist_tok_out, int32_t* hist_par_out, float* beam_scores_out,
                                      float* hyps_out, int32_t* n_hyps_out, uint8_t* done_out, int32_t* n_steps_out,
                                      void* workspace, size_t workspace_bytes, int use_graph, void* caller_stream) {
    if (!h || !prefix_embeds || !gpp || !hist_tok_out || !hist_par_out || !beam_scores_out || !hyps_out || !n_hyps_out || !done_out ||
        !n_steps_out || !workspace) { itts_set_error("gpt_generate_beam: null pointer"); return ITTS_ERR_ARG; }
    if (!h->finalized) { itts_set_error("gpt_generate_beam: call itts_gpt_finalize first"); return ITTS_ERR_STATE; }
    ItDevGuard dg(h->device);
    if (int rcd = check_same_device(h, prefix_embeds, workspace, "gpt_generate_beam")) return rcd;
    const itts_gpt_config& c = h->cfg;
    const itts_gen_params gp = *gpp;
    const int nb = num_beams, B = n_utts, nseq = B * nb;
    if (B <= 0 || nb < 2 || nb > BEAM_MAX || S <= 0 || gp.max_new_tokens <= 0) { itts_set_error("gpt_generate_beam: bad sizes"); return ITTS_ERR_ARG; }
    if (gp.max_new_tokens + gp.pos_offset > c.n_mel_pos + 1) { itts_set_error("gpt_generate_beam: max_new_tokens exceeds the mel position table"); return ITTS_ERR_ARG; }
    if (n_penalty_ids < 0 || n_penalty_ids > 16) { itts_set_error("gpt_generate_beam: at most 16 initial penalty ids"); return ITTS_ERR_ARG; }
    const int Sb = s_bucket(S), Tmax = Sb + gp.max_new_tokens;
    const GptWs w0 = carve(c, nullptr, nseq, Sb, Tmax, nb);
    if (workspace_bytes < w0.total) { itts_set_error("gpt_generate_beam: workspace too small (%zu < %zu)", workspace_bytes, w0.total); return ITTS_ERR_ARG; }
    char* base = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    const GptWs w = carve(c, base, nseq, Sb, Tmax, nb);
    hipStream_t st = h->stream, cs = (hipStream_t)caller_stream;
    int rc;
    if (h->fin_cap < nseq) {
        if (h->host_fin) (void)hipHostFree(h->host_fin);
        HIP_TRY(hipHostMalloc((void**)&h->host_fin, (size_t)nseq, hipHostMallocDefault));
        h->fin_cap = nseq;
    }
    HIP_TRY(hipEventRecord(h->ev_in, cs));
    HIP_TRY(hipStreamWaitEvent(st, h->ev_in, 0));
    HIP_TRY(hipMemsetAsync(w.seen, 0, (size_t)nseq * c.vocab, st));
    HIP_TRY(hipMemsetAsync(w.seen2, 0, (size_t)nseq * c.vocab, st));
    HIP_TRY(hipMemsetAsync(w.hist_tok, 0, (size_t)gp.max_new_tokens * nseq * 4, st));
    HIP_TRY(hipMemsetAsync(w.hist_par, 0, (size_t)gp.max_new_tokens * nseq * 4, st));
    HIP_TRY(hipMemsetAsync(w.hyps, 0, (size_t)B * BEAM_MAX * sizeof(BeamHyp), st));
    if (pad_lens) HIP_TRY(hipMemcpyAsync(w.pad, pad_lens, (size_t)nseq * 4, hipMemcpyDeviceToDevice, st));
    else HIP_TRY(hipMemsetAsync(w.pad, 0, (size_t)nseq * 4, st));
    if (n_penalty_ids > 0) {
        HIP_TRY(hipMemcpyAsync(w.pen_ids, penalty_ids, (size_t)n_penalty_ids * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(mark_seen_kernel, dim3(nseq), dim3(64), 0, st, w.seen, w.pen_ids, n_penalty_ids, c.vocab);
    }
    hipLaunchKernelGGL(beam_init_kernel, dim3(nseq), dim3(256), 0, st, w.row_map[0], w.beam_scores, w.worst, w.n_hyps, w.done, nseq, nb, Tmax, S);
    hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(1), 0, st, w.state, 0, 0);
    hipLaunchKernelGGL(set_seed_kernel, dim3(1), dim3(1), 0, st, w.state, (unsigned long long)gp.seed);
    // the nb rows of an utterance carry the same prompt (repeat_interleave): prefill the B unique prompts only
    const size_t row_bytes = (size_t)S * c.model_dim * 4;
    HIP_TRY(hipMemcpy2DAsync(w.x, row_bytes, prefix_embeds, row_bytes * nb, row_bytes, B, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipGetLastError());

    HIP_TRY(hipEventRecord(h->ev_t0, st));
    bool pending = false;
    rc = run_layers(h, w, B, S, Tmax, true, w.state + 1, w.pad, &pending, st, false, nb);
    if (rc) return rc;
    if ((rc = run_head(h, w, B, S, S - 1, pending, st))) return rc;
    hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(1), 0, st, w.state, 0, S);
    const BeamArgs ba = make_beam(h, w, gp, B, nb, S, Tmax, uniforms);
    BeamArgs ba0 = ba;
    ba0.logits_shared = 1;                       // one logits row per utterance after the shared prefill
    if ((rc = launch_beam_step(ba0, st))) return rc;
    if ((rc = launch_beam_apply(ba, st))) return rc;
    hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(1), 0, st, w.state, 1, S);
    HIP_TRY(hipEventRecord(h->ev_t1, st));

    int steps = 1;
    hipGraphExec_t exec = nullptr;
    bool graph_ok = false;
    if (use_graph && gp.max_new_tokens > 1) {
        itts_gpt::GraphEntry key{};
        key.base = base; key.uniforms = uniforms; key.nseq = nseq; key.nb = nb; key.Sb = Sb; key.Tmax = Tmax; key.S = S; key.gp = gp; key.gp.seed = 0;
        key.opt_epoch = itts_opt_epoch();
        exec = graph_lookup(h, key);
        graph_ok = exec != nullptr;
        if (!graph_ok) {
            hipGraph_t graph = nullptr;
            hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
            if (e == hipSuccess) {
                rc = decode_step_beam(h, w, ba, nseq, Tmax, st);
                e = hipStreamEndCapture(st, &graph);
                if (rc == ITTS_OK && e == hipSuccess && graph) {
                    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
                    graph_ok = (e == hipSuccess && exec);
                }
                if (graph) (void)hipGraphDestroy(graph);
            }
            if (!graph_ok) {
                (void)hipGetLastError();
                itts_set_error("gpt_generate_beam: hipGraph capture failed (%s); rerun with use_graph=0", hipGetErrorString(e));
                return ITTS_ERR_HIP;
            }
            graph_insert(h, key, exec);
        }
    }
    // the reference loop stops when every utterance is done (checked right after the scorer) or at max_length
    auto all_done = [&](bool* out) -> int {
        HIP_TRY(hipMemcpyAsync(h->host_fin, w.done, B, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        bool all = true;
        for (int i = 0; i < B; ++i) all = all && h->host_fin[i];
        *out = all;
        return ITTS_OK;
    };
    bool fin = false;
    if ((rc = all_done(&fin))) return rc;
    while (!fin && steps < gp.max_new_tokens) {
        if (graph_ok) { HIP_TRY(hipGraphLaunch(exec, st)); }
        else if ((rc = decode_step_beam(h, w, ba, nseq, Tmax, st))) return rc;
        ++steps;
        // finished utterances are frozen on the device (their block returns early), so a late check only costs idle
        // steps; the host reports min(steps, the step at which the last utterance finished) like the reference loop
        if (steps % 4 == 0 && (rc = all_done(&fin))) return rc;
    }
    HIP_TRY(hipEventRecord(h->ev_t2, st));
    HIP_TRY(hipMemcpyAsync(hist_tok_out, w.hist_tok, (size_t)gp.max_new_tokens * nseq * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(hist_par_out, w.hist_par, (size_t)gp.max_new_tokens * nseq * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(beam_scores_out, w.beam_scores, (size_t)nseq * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(hyps_out, w.hyps, (size_t)B * BEAM_MAX * sizeof(BeamHyp), hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(n_hyps_out, w.n_hyps, (size_t)B * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(done_out, w.done, (size_t)B, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipEventRecord(h->ev_out, st));
    HIP_TRY(hipStreamWaitEvent(cs, h->ev_out, 0));
    HIP_TRY(hipStreamSynchronize(st));
    (void)hipEventElapsedTime(&h->last_prefill_ms, h->ev_t0, h->ev_t1);
    (void)hipEventElapsedTime(&h->last_decode_ms, h->ev_t1, h->ev_t2);
    h->last_steps = steps;
    *n_steps_out = steps;
    return ITTS_OK;
}

// Row compaction of ragged decode batches: on by default; granularity = rows per bucket (the decode graph is captured once per bucket).
extern "C" int itts_gpt_set_compaction(itts_gpt* h, int enable, int granularity) {
    if (!h) { itts_set_error("gpt_set_compaction: null"); return ITTS_ERR_ARG; }
    h->compact = enable != 0;
    if (granularity > 0) h->compact_gran = granularity;
    return ITTS_OK;
}

// In-flight batching: a chunk call returns early -- at one of the finished-flag checks the loop makes every 8 steps anyway -- once at least
// `finished_rows` utterances of the batch have finished (counting the ones that had finished before the call), so the caller can refill their slots
// (itts_gpt_admit_rows) without polling in short chunks.  0 = off (run to step_limit).
extern "C" int itts_gpt_set_chunk_return(itts_gpt* h, int finished_rows) {
    if (!h || finished_rows < 0) { itts_set_error("gpt_set_chunk_return: bad args"); return ITTS_ERR_ARG; }
    h->chunk_return_finished = finished_rows;
    return ITTS_OK;
}

// Per-utterance caps on the generated tokens for the following itts_gpt_generate / _chunk calls whose batch has exactly n utterances
// (a batch merges requests that carry their own max_mel_tokens): limits is a DEVICE int32 [n] the caller keeps alive; null clears.
extern "C" int itts_gpt_set_row_limits(itts_gpt* h, const int32_t* limits, int n) {
    if (!h || (limits && n <= 0)) { itts_set_error("gpt_set_row_limits: bad args"); return ITTS_ERR_ARG; }
    h->row_limits = limits;
    h->row_limits_n = limits ? n : 0;
    return ITTS_OK;
}

// Of the last generate call: sum over its decode steps of the rows each step ran, and the number of compactions.
extern "C" int itts_gpt_compaction_stats(const itts_gpt* h, int64_t* row_steps, int32_t* compactions) {
    if (!h) return ITTS_ERR_ARG;
    if (row_steps) *row_steps = h->last_row_steps;
    if (compactions) *compactions = h->last_compactions;
    return ITTS_OK;
}

extern "C" int itts_gpt_graph_stats(const itts_gpt* h, int32_t* captures, int32_t* hits) {
    if (!h) return ITTS_ERR_ARG;
    if (captures) *captures = h->graph_captures;
    if (hits) *hits = h->graph_hits;
    return ITTS_OK;
}

extern "C" int itts_gpt_last_timing(const itts_gpt* h, float* prefill_ms, float* decode_ms, int32_t* steps) {
    if (!h) return ITTS_ERR_ARG;
    if (prefill_ms) *prefill_ms = h->last_prefill_ms;
    if (decode_ms) *decode_ms = h->last_decode_ms;
    if (steps) *steps = h->last_steps;
    return ITTS_OK;
}

// Teacher-forced pass over full sequences: out = final_norm(ln_f(stack(x)))  (UnifiedVoice.forward / get_logits,
// indextts/gpt/model_v2.py:528-554,596-646).  x [nseq][S][D] f32 device, out [nseq][S][D] f32 device.
extern "C" int itts_gpt_forward_latent(itts_gpt* h, const float* x, int nseq, int S, float* out, void* workspace,
                                       size_t workspace_bytes, void* caller_stream) {
    if (!h || !x || !out || !workspace) { itts_set_error("gpt_forward_latent: null pointer"); return ITTS_ERR_ARG; }
    if (!h->finalized) { itts_set_error("gpt_forward_latent: call itts_gpt_finalize first"); return ITTS_ERR_STATE; }
    ItDevGuard dg(h->device);
    if (int rcd = check_same_device(h, x, workspace, "gpt_forward_latent")) return rcd;
    const itts_gpt_config& c = h->cfg;
    if (nseq <= 0 || S <= 0 || S > 65535) { itts_set_error("gpt_forward_latent: bad shape"); return ITTS_ERR_ARG; }
    const GptWs w0 = carve(c, nullptr, nseq, S, S);
    if (workspace_bytes < w0.total) { itts_set_error("gpt_forward_latent: workspace too small (%zu < %zu)", workspace_bytes, w0.total); return ITTS_ERR_ARG; }
    char* base = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    const GptWs w = carve(c, base, nseq, S, S);
    hipStream_t st = h->stream, cs = (hipStream_t)caller_stream;
    HIP_TRY(hipEventRecord(h->ev_in, cs));
    HIP_TRY(hipStreamWaitEvent(st, h->ev_in, 0));
    hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(1), 0, st, w.state, 0, 0);
    HIP_TRY(hipMemcpyAsync(w.x, x, (size_t)nseq * S * c.model_dim * 4, hipMemcpyDeviceToDevice, st));
    bool pending = false;
    int rc = run_layers(h, w, nseq, S, S, true, w.state + 1, nullptr, &pending, st);
    if (rc) return rc;
    LnArgs ln{};
    ln.x = w.x; ln.g1 = h->lnf_g; ln.b1 = h->lnf_b; ln.g2 = h->fn_g; ln.b2 = h->fn_b; ln.out = out; ln.out_f32 = 1;
    ln.rows = nseq * S; ln.D = c.model_dim; ln.in_row_mul = 1; ln.in_row_add = 0; ln.eps = c.ln_eps; ln.nsplit = 1;
    if ((rc = launch_ln(ln, c.precision, st))) return rc;
    HIP_TRY(hipEventRecord(h->ev_out, st));
    HIP_TRY(hipStreamWaitEvent(cs, h->ev_out, 0));
    HIP_TRY(hipStreamSynchronize(st));
    return ITTS_OK;
}

extern "C" int itts_gemm_tile_occupancy(int precision, int32_t* blocks_per_cu) {
    if (!blocks_per_cu) { itts_set_error("gemm_tile_occupancy: null"); return ITTS_ERR_ARG; }
    int n = 0;
    const int rc = gemm_tile_occupancy(precision, &n);
    *blocks_per_cu = n;
    return rc;
}

// ---- unit-level entry points (parity tests) --------------------------------------------------------------------
extern "C" int itts_gemm_forward(const void* A, const void* Wp, const float* bias, float* out, int M, int N, int K,
                                 int precision, int prefill_tiles, int gelu, void* stream) {
    if (!A || !Wp || !out) { itts_set_error("gemm_forward: null pointer"); return ITTS_ERR_ARG; }
    GemmArgs g{};
    g.A = A; g.lda = K; g.Wp = Wp; g.bias = bias; g.M = M; g.N = N; g.K = K; g.nsplit = 1; g.D = N;
    g.epi = EPI_STORE_F32; g.out_f32 = out; g.ldo = N;
    (void)gelu;
    return launch_gemm(g, precision, prefill_tiles != 0, (hipStream_t)stream);
}

extern "C" int itts_gemm_ln_forward(const float* x, const float* partial, const float* bias_prev, const float* ln_gamma, const float* ln_beta,
                                    float eps, const void* Wp, const float* bias, float* out, float* x_out, int M, int N, int K, void* stream) {
    if (!x || !ln_gamma || !ln_beta || !Wp || !out) { itts_set_error("gemm_ln_forward: null pointer"); return ITTS_ERR_ARG; }
    if (!gemm_decode_ln_ok(M, K, EPI_STORE_F32)) { itts_set_error("gemm_ln_forward: M = %d (1..16), K = %d (256, 512, 1280) unsupported", M, K); return ITTS_ERR_ARG; }
    GemmArgs g{};
    g.Wp = Wp; g.bias = bias; g.M = M; g.N = N; g.K = K; g.nsplit = 1; g.D = N;
    g.epi = EPI_STORE_F32; g.out_f32 = out; g.ldo = N;
    g.ln_x = x; g.ln_partial = partial; g.ln_bias_prev = bias_prev; g.ln_g = ln_gamma; g.ln_b = ln_beta; g.ln_eps = eps; g.ln_x_out = x_out;
    return launch_gemm_decode_ln(g, (hipStream_t)stream);
}

extern "C" int itts_layernorm_forward(const float* x, const float* gamma, const float* beta, const float* gamma2,
                                      const float* beta2, float* out, int rows, int D, float eps, void* stream) {
    if (!x || !gamma || !beta || !out) { itts_set_error("layernorm_forward: null pointer"); return ITTS_ERR_ARG; }
    LnArgs ln{};
    ln.x = const_cast<float*>(x); ln.g1 = gamma; ln.b1 = beta; ln.g2 = gamma2; ln.b2 = beta2; ln.out = out; ln.out_f32 = 1;
    ln.rows = rows; ln.D = D; ln.in_row_mul = 1; ln.in_row_add = 0; ln.eps = eps; ln.nsplit = 1;
    return launch_ln(ln, PREC_F32, (hipStream_t)stream);
}
