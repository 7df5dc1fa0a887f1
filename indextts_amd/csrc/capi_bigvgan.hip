// C-ABI entry points for the BigVGAN vocoder (see include/indextts_hip.h for the reference call sites replaced).
#include <math.h>
#include <stdarg.h>
#include <string.h>
#include <map>
#include <string>
#include <vector>

#include "../../include/indextts_hip.h"
#include "bigvgan_kernels.h"

// ---- error plumbing (shared by all capi_* files) -------------------------------------------------------------
static thread_local char g_err[1024] = "";
void itts_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* itts_last_error(void) { return g_err; }
extern "C" int itts_abi_version(void) { return ITTS_ABI_VERSION; }
extern "C" int itts_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        itts_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return -1;
    }
    return n;
}

// ---- host-side packing ---------------------------------------------------------------------------------------
extern "C" size_t itts_packed_conv_floats(int Cout, int Cin, int k) {
    return (size_t)((Cout + 31) / 32) * k * (Cin / 8) * 64 * 4;
}

// generic packer over an accessor w(co, ci, j)
template <class F>
static void pack_generic(F w, int Cout, int Cin, int k, float* out) {
    const int n_cosub = (Cout + 31) / 32, cin8 = Cin / 8;
    for (int cs = 0; cs < n_cosub; ++cs)
        for (int j = 0; j < k; ++j)
            for (int c8 = 0; c8 < cin8; ++c8)
                for (int lane = 0; lane < 64; ++lane)
                    for (int s = 0; s < 4; ++s) {
                        const int co = cs * 32 + (lane & 31);
                        const int ci = c8 * 8 + 2 * s + (lane >> 5);
                        const size_t o = ((((size_t)cs * k + j) * cin8 + c8) * 64 + lane) * 4 + s;
                        out[o] = co < Cout ? w(co, ci, j) : 0.f;
                    }
}

extern "C" int itts_pack_conv1d_weight(const float* w, int Cout, int Cin, int k, float* out) {
    if (!w || !out || Cout <= 0 || Cin <= 0 || Cin % 8 || k <= 0) {
        itts_set_error("pack_conv1d: bad args Cout=%d Cin=%d k=%d (C_in must be a multiple of 8)", Cout, Cin, k);
        return ITTS_ERR_ARG;
    }
    pack_generic([&](int co, int ci, int j) { return w[((size_t)co * Cin + ci) * k + j]; }, Cout, Cin, k, out);
    return ITTS_OK;
}

extern "C" int itts_pack_convT_weight(const float* w, int Cin, int Cout, int k, int u, int phase, float* out) {
    if (!w || !out || Cin % 8 || u < 1 || k < u || k % u || ((k - u) & 1) || phase < 0 || phase >= u) {
        itts_set_error("pack_convT: bad args Cin=%d Cout=%d k=%d u=%d phase=%d (need k a multiple of u with k-u even)", Cin, Cout, k, u, phase);
        return ITTS_ERR_ARG;
    }
    // k/u taps per phase: tap jj of phase r uses kernel index r + jj*u and reads x[m - jj]  (k == u: one tap, padding 0 --
    // the 4/4 upsamplers of the v1.5 vocoder; k == 2u: two taps -- BigVGAN-v2)
    pack_generic([&](int co, int ci, int jj) { return w[((size_t)ci * Cout + co) * k + phase + jj * u]; }, Cout, Cin, k / u,
                 out);
    return ITTS_OK;
}

// ---- unit entry points ---------------------------------------------------------------------------------------
extern "C" int itts_aa_act_forward(const float* x, float* y, const float* alpha, const float* beta,
                                   const float* up_filter, const float* down_filter, int B, int C, int T,
                                   const int32_t* lens, int len_mult, int logscale, void* stream) {
    if (!x || !y || !alpha || !beta || !up_filter || !down_filter || B < 0 || C < 0 || T < 0) {
        itts_set_error("aa_act: null pointer or negative dim");
        return ITTS_ERR_ARG;
    }
    if (B > 65535 || C > 65535) { itts_set_error("aa_act: B, C must be <= 65535"); return ITTS_ERR_ARG; }
    return launch_aa_act(x, y, alpha, beta, up_filter, down_filter, B, C, T, lens, len_mult < 1 ? 1 : len_mult,
                         logscale, (hipStream_t)stream);
}

static int conv1d_impl(const float* x, const float* wpk, const float* bias, const float* bias_b, const float* res,
                       float* y, int B, int Cin, int Cout, int T, int k, int dil, const int* lens, int len_mult,
                       int acc_mode, float div, hipStream_t st) {
    ConvArgs a;
    a.x = x; a.y = y; a.wpk = wpk; a.bias = bias; a.bias_b = bias_b; a.res = res; a.lens = lens;
    a.len_mult_in = a.len_mult_out = len_mult;
    a.Cin = Cin; a.Cout = Cout; a.Tin = T; a.Tout = T;
    a.k = k; a.tap_base = -((k - 1) / 2) * dil; a.tap_step = dil; a.ostride = 1; a.ooff = 0; a.m_extra = 0;
    a.acc_mode = acc_mode; a.div = div;
    return launch_conv(a, B, st);
}

extern "C" int itts_conv1d_forward(const float* x, const float* wpk, const float* bias, const float* bias_b,
                                   const float* res, float* y, int B, int Cin, int Cout, int T, int k, int dilation,
                                   const int32_t* lens, int len_mult, int acc_mode, float div, void* stream) {
    if (!x || !wpk || !y || B < 0 || T < 0 || k < 1 || (k & 1) == 0 || dilation < 1) {
        itts_set_error("conv1d: bad args (odd k, dilation >= 1 required)");
        return ITTS_ERR_ARG;
    }
    if (B > 65535) { itts_set_error("conv1d: B must be <= 65535"); return ITTS_ERR_ARG; }
    if (T == 0 || B == 0) return ITTS_OK;
    return conv1d_impl(x, wpk, bias, bias_b, res, y, B, Cin, Cout, T, k, dilation, lens, len_mult < 1 ? 1 : len_mult,
                       acc_mode, div, (hipStream_t)stream);
}

// ---- f16 x 3 split-operand conv as a unit op (tests / microbenchmarks; the model path packs at load time) ----------------------
extern "C" size_t itts_conv1d_h3_packed_bytes(int Cout, int Cin, int k) { return (Cin > 0 && Cin % 32 == 0 && Cout > 0 && k > 0) ? conv_h3_packed_bytes(Cout, Cin, k) : 0; }
extern "C" int itts_pack_conv1d_h3_weight(const float* w, int Cout, int Cin, int k, void* out) { return conv_h3_pack(w, Cout, Cin, k, out); }
extern "C" size_t itts_conv1d_h3_scratch_bytes(int B, int Cin, int T) { return (size_t)B * T * Cin * 4 + 512; }
extern "C" int itts_conv1d_h3_forward(const float* x, const void* wp3, const float* bias, const float* res, float* y, int B, int Cin, int Cout,
                                      int T, int k, int dilation, const int32_t* lens, int len_mult, int acc_mode, float div, void* scratch,
                                      void* stream) {
    if (!x || !wp3 || !y || !scratch || B < 0 || T < 0 || acc_mode < 0 || acc_mode > 2) { itts_set_error("conv1d_h3_forward: bad args"); return ITTS_ERR_ARG; }
    if (B == 0 || T == 0) return ITTS_OK;
    hipStream_t st = (hipStream_t)stream;
    char* base = (char*)(((uintptr_t)scratch + 255) & ~(uintptr_t)255);
    const size_t half = (size_t)B * T * Cin * 2;
    void* zr = base + half * 2;                           // the zero row lives behind the two halves
    HIP_TRY(hipMemsetAsync(zr, 0, 64, st));
    int rc = launch_split_tm(x, base, base + half, B, Cin, T, lens, len_mult < 1 ? 1 : len_mult, nullptr, st);
    if (rc) return rc;
    ConvH3Args g{};
    g.xh = base; g.xl = base + half; g.wp = wp3; g.bias = bias; g.res = res; g.y = y; g.zero_row = zr; g.lens = lens; g.len_mult = len_mult < 1 ? 1 : len_mult;
    g.B = B; g.Cin = Cin; g.Cout = Cout; g.T = T; g.k = k; g.dil = dilation; g.acc_mode = acc_mode; g.div = div;
    return launch_conv_h3(g, st);
}

// ---- bf16 x 3 plane-operand conv as a unit op (tests / microbenchmarks; the model path packs at load time) ----------------------------
extern "C" size_t itts_conv1d_x3_packed_bytes(int Cout, int Cin, int k) { return (Cin > 0 && Cin % 32 == 0 && Cout > 0 && k > 0) ? conv_x3_packed_bytes(Cout, Cin, k) : 0; }
extern "C" int itts_pack_conv1d_x3_weight(const float* w, int Cout, int Cin, int k, void* out) { return conv_x3_pack(w, Cout, Cin, k, out); }
extern "C" size_t itts_conv1d_x3_scratch_bytes(int B, int Cin, int T) { return (size_t)B * T * Cin * 6 + 512; }
extern "C" int itts_conv1d_x3_forward(const float* x, const void* wp3, const float* bias, const float* res, float* y, int B, int Cin, int Cout,
                                      int T, int k, int dilation, const int32_t* lens, int len_mult, int acc_mode, float div, void* scratch,
                                      void* stream) {
    if (!x || !wp3 || !y || !scratch || B < 0 || T < 0 || acc_mode < 0 || acc_mode > 2) { itts_set_error("conv1d_x3_forward: bad args"); return ITTS_ERR_ARG; }
    if (B == 0 || T == 0) return ITTS_OK;
    hipStream_t st = (hipStream_t)stream;
    char* base = (char*)(((uintptr_t)scratch + 255) & ~(uintptr_t)255);
    const size_t planes = (size_t)B * T * Cin * 6;
    void* zr = base + planes;                             // the zero row lives behind the three planes
    HIP_TRY(hipMemsetAsync(zr, 0, 64, st));
    int rc = launch_split_tm3(x, base, B, Cin, T, lens, len_mult < 1 ? 1 : len_mult, st);
    if (rc) return rc;
    ConvX3Args g{};
    g.xp = base; g.wp = wp3; g.bias = bias; g.res = res; g.y = y; g.zero_row = zr; g.lens = lens; g.len_mult = len_mult < 1 ? 1 : len_mult;
    g.B = B; g.Cin = Cin; g.Cout = Cout; g.T = T; g.k = k; g.dil = dilation; g.acc_mode = acc_mode; g.div = div;
    return launch_conv_x3(g, st);
}

static int convT_impl(const float* x, const float* wpk_phases, const float* bias, const float* bias_b, float* y, int B,
                      int Cin, int Cout, int Tin, int k, int u, const int* lens, int len_mult_in, hipStream_t st) {
    const int p = (k - u) / 2, ntaps = k / u;
    const size_t per_phase = itts_packed_conv_floats(Cout, Cin, ntaps);
    for (int r = 0; r < u; ++r) {
        ConvArgs a;
        a.x = x; a.y = y; a.wpk = wpk_phases + per_phase * r; a.bias = bias; a.bias_b = bias_b; a.res = nullptr;
        a.lens = lens; a.len_mult_in = len_mult_in; a.len_mult_out = len_mult_in * u;
        a.Cin = Cin; a.Cout = Cout; a.Tin = Tin; a.Tout = Tin * u;
        a.k = ntaps; a.tap_base = 0; a.tap_step = -1; a.ostride = u; a.ooff = r - p; a.m_extra = ntaps - 1;
        a.acc_mode = 0; a.div = 1.f;
        int rc = launch_conv(a, B, st);
        if (rc) return rc;
    }
    return ITTS_OK;
}

extern "C" int itts_conv_transpose1d_forward(const float* x, const float* wpk_phases, const float* bias,
                                             const float* bias_b, float* y, int B, int Cin, int Cout, int Tin, int k,
                                             int u, const int32_t* lens, int len_mult_in, void* stream) {
    if (!x || !wpk_phases || !y || u < 1 || k < u || k % u || ((k - u) & 1)) {
        itts_set_error("conv_transpose1d: need k a multiple of u with k-u even (k=%d u=%d)", k, u);
        return ITTS_ERR_ARG;
    }
    if (B <= 0 || Tin <= 0) return ITTS_OK;
    return convT_impl(x, wpk_phases, bias, bias_b, y, B, Cin, Cout, Tin, k, u, lens, len_mult_in < 1 ? 1 : len_mult_in,
                      (hipStream_t)stream);
}

// ---- model object --------------------------------------------------------------------------------------------
struct DevBuf {
    float* p = nullptr;
    size_t n = 0;
};
struct ConvL {
    int Cin = 0, Cout = 0, k = 0;
    DevBuf w, b;
    bool has_w = false, has_b = false;
    void* w3 = nullptr;                 // f16 x 3 mode: split weight fragments (conv_h3_pack), resblock convs with enough channels only
};
struct ActL {
    int C = 0;
    DevBuf alpha, beta, fu, fd;
    bool has_a = false, has_b = false, has_fu = false, has_fd = false;
};

struct itts_bigvgan {
    itts_bigvgan_config cfg;
    ConvL conv_pre;
    std::vector<ConvL> ups;             // w = u packed phases
    std::vector<ConvL> convs1, convs2;  // [resblock n][dilation d] flattened n*ND + d
    std::vector<ActL> acts;             // [n][2*ND] flattened
    ActL act_post;
    DevBuf post_w, post_b;
    bool has_post_w = false, has_post_b = false;
    ConvL cond_layer;                   // raw [Cout][cond_dim] (not packed)
    std::vector<ConvL> conds;
    DevBuf default_filter;
    bool finalized = false;
    std::vector<float*> owned;
    int total_up = 1;
    // optional per-launch HIP-event profiling (itts_bigvgan_set_profiling)
    bool profiling = false;
    std::vector<hipEvent_t> ev_pool;
    struct Rec { int cls; double flops; double bytes; };
    std::vector<Rec> recs;      // one per launch of the last forward; events 2i, 2i+1
    hipStream_t prof_stream = nullptr;
    int device = -1;            // device current at itts_bigvgan_create: owns the weights and profiling events
    // opt-in conv mode (itts_bigvgan_set_conv_mode): 0 = exact f32 MFMA (default, the parity mode), 1 = f16 x 3 split operands for
    // the resblock convs with >= h3_min_c channels
    int conv_mode = 0, h3_min_c = 96;
    void* zero_row = nullptr;
};

// profiling classes
enum { PC_CONV = 0, PC_CONVT = 1, PC_ACT = 2, PC_POST = 3, PC_COUNT = 4 };

struct ProfScope {
    itts_bigvgan* h;
    hipStream_t st;
    bool on;
    size_t idx;
    ProfScope(itts_bigvgan* h_, hipStream_t st_, int cls, double flops, double bytes) : h(h_), st(st_), on(h_->profiling), idx(0) {
        if (!on) return;
        idx = h->recs.size();
        h->recs.push_back({cls, flops, bytes});
        while (h->ev_pool.size() < 2 * (idx + 1)) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) { on = false; return; }
            h->ev_pool.push_back(e);
        }
        (void)hipEventRecord(h->ev_pool[2 * idx], st);
    }
    ~ProfScope() {
        if (on) (void)hipEventRecord(h->ev_pool[2 * idx + 1], st);
    }
};

static int upload(itts_bigvgan* h, const float* host, size_t n, DevBuf* dst) {
    if (dst->p) {   // reload: replace
        dst->p = nullptr;
    }
    float* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, n * sizeof(float)));
    h->owned.push_back(d);
    HIP_TRY(hipMemcpy(d, host, n * sizeof(float), hipMemcpyHostToDevice));
    dst->p = d;
    dst->n = n;
    return ITTS_OK;
}

static int stage_channels(const itts_bigvgan_config& c, int i) { return c.upsample_initial_channel >> (i + 1); }

extern "C" int itts_bigvgan_create(const itts_bigvgan_config* cfg, itts_bigvgan** out) {
    if (!cfg || !out) { itts_set_error("bigvgan_create: null"); return ITTS_ERR_ARG; }
    const itts_bigvgan_config& c = *cfg;
    if (c.num_upsamples < 1 || c.num_upsamples > 8 || c.num_kernels < 1 || c.num_kernels > 4 || c.num_dilations < 1 ||
        c.num_dilations > 4 || c.in_channels % 8 || c.in_channels <= 0) {
        itts_set_error("bigvgan_create: unsupported config (ups=%d kernels=%d dil=%d in=%d)", c.num_upsamples,
                       c.num_kernels, c.num_dilations, c.in_channels);
        return ITTS_ERR_ARG;
    }
    for (int i = 0; i < c.num_upsamples; ++i) {
        const int uk = c.upsample_kernel_sizes[i], uu = c.upsample_rates[i];
        if (uu < 1 || uk < uu || uk % uu || ((uk - uu) & 1)) {
            itts_set_error("bigvgan_create: upsampler %d needs k a multiple of the stride with k-stride even (k=%d u=%d)", i, uk, uu);
            return ITTS_ERR_ARG;
        }
        if (stage_channels(c, i) < 1 || (c.upsample_initial_channel >> i) % 8) {
            itts_set_error("bigvgan_create: channel count at stage %d not a multiple of 8", i);
            return ITTS_ERR_ARG;
        }
    }
    if (stage_channels(c, c.num_upsamples - 1) % 8) {
        itts_set_error("bigvgan_create: final channel count %d not a multiple of 8", stage_channels(c, c.num_upsamples - 1));
        return ITTS_ERR_ARG;
    }
    for (int j = 0; j < c.num_kernels; ++j) {
        if ((c.resblock_kernel_sizes[j] & 1) == 0) { itts_set_error("even resblock kernel"); return ITTS_ERR_ARG; }
        for (int d = 0; d < c.num_dilations; ++d)
            if ((c.resblock_kernel_sizes[j] - 1) * c.resblock_dilations[j][d] > 64) {
                itts_set_error("bigvgan_create: receptive span of k=%d d=%d exceeds the 64-sample LDS halo",
                               c.resblock_kernel_sizes[j], c.resblock_dilations[j][d]);
                return ITTS_ERR_ARG;
            }
    }
    itts_bigvgan* h = new itts_bigvgan();
    h->cfg = c;
    h->device = itts_current_device();
    const int nres = c.num_upsamples * c.num_kernels;
    h->ups.resize(c.num_upsamples);
    h->conds.resize(c.num_upsamples);
    h->convs1.resize((size_t)nres * c.num_dilations);
    h->convs2.resize((size_t)nres * c.num_dilations);
    h->acts.resize((size_t)nres * 2 * c.num_dilations);
    h->total_up = 1;
    for (int i = 0; i < c.num_upsamples; ++i) h->total_up *= c.upsample_rates[i];
    *out = h;
    return ITTS_OK;
}

extern "C" int itts_bigvgan_device(const itts_bigvgan* h) { return h ? h->device : -1; }

extern "C" int itts_bigvgan_set_conv_mode(itts_bigvgan* h, int mode, int min_channels) {
    if (!h || mode < 0 || mode > 2) { itts_set_error("bigvgan_set_conv_mode: mode 0 (f32 MFMA), 1 (f16 x 3 split operands) or 2 (bf16 x 3 plane operands)"); return ITTS_ERR_ARG; }
    for (const ConvL& L : h->convs1)
        if (L.has_w) { itts_set_error("bigvgan_set_conv_mode: call before loading the weights"); return ITTS_ERR_STATE; }
    ItDevGuard dg(h->device);
    h->conv_mode = mode;
    if (min_channels > 0) h->h3_min_c = min_channels;
    if (mode >= 1 && !h->zero_row) {
        void* z = nullptr;
        HIP_TRY(hipMalloc(&z, 256));
        HIP_TRY(hipMemset(z, 0, 256));
        h->owned.push_back((float*)z);
        h->zero_row = z;
    }
    return ITTS_OK;
}

// f16 x 3 mode: 1 if a forward since the last call met an activation that is not finite or outside the f16 range (the result of
// that forward is invalid), else 0; synchronises the device and clears the flag.  < 0 on error.
extern "C" int itts_bigvgan_range_check(itts_bigvgan* h) {
    if (!h) return -1;
    if (!h->zero_row) return 0;
    ItDevGuard dg(h->device);
    int v = 0;
    if (hipMemcpy(&v, (char*)h->zero_row + 128, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (v && hipMemset((char*)h->zero_row + 128, 0, 4) != hipSuccess) return -1;
    return v ? 1 : 0;
}

extern "C" void itts_bigvgan_destroy(itts_bigvgan* h) {
    if (!h) return;
    ItDevGuard dg(h->device);
    for (float* p : h->owned) (void)hipFree(p);
    for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
    delete h;
}

static bool parse_idx(const char*& s, int* v) {
    if (*s < '0' || *s > '9') return false;
    int x = 0;
    while (*s >= '0' && *s <= '9') x = x * 10 + (*s++ - '0');
    *v = x;
    return true;
}
static bool eat(const char*& s, const char* lit) {
    size_t n = strlen(lit);
    if (strncmp(s, lit, n) == 0) { s += n; return true; }
    return false;
}

static int load_conv(itts_bigvgan* h, ConvL* L, const char* what, const float* data, const int64_t* shape, int ndim,
                     int Cout, int Cin, int k, bool resblock = false) {
    if (!strcmp(what, "weight")) {
        if (ndim != 3 || shape[0] != Cout || shape[1] != Cin || shape[2] != k) {
            itts_set_error("conv weight shape mismatch: got [%lld,%lld,%lld] want [%d,%d,%d]", (long long)shape[0],
                           (long long)(ndim > 1 ? shape[1] : -1), (long long)(ndim > 2 ? shape[2] : -1), Cout, Cin, k);
            return ITTS_ERR_ARG;
        }
        std::vector<float> pk(itts_packed_conv_floats(Cout, Cin, k));
        int rc = itts_pack_conv1d_weight(data, Cout, Cin, k, pk.data());
        if (rc) return rc;
        L->Cin = Cin; L->Cout = Cout; L->k = k; L->has_w = true;
        L->w3 = nullptr;
        if (resblock && h->conv_mode == 2 && Cin % 32 == 0 && Cin >= h->h3_min_c) {                        // bf16 x 3: three weight planes in fragment order
            std::vector<char> p3(conv_x3_packed_bytes(Cout, Cin, k));
            rc = conv_x3_pack(data, Cout, Cin, k, p3.data());
            if (rc) return rc;
            void* d = nullptr;
            HIP_TRY(hipMalloc(&d, p3.size()));
            h->owned.push_back((float*)d);
            HIP_TRY(hipMemcpy(d, p3.data(), p3.size(), hipMemcpyHostToDevice));
            L->w3 = d;
        }
        if (resblock && h->conv_mode == 1 && Cin % 32 == 0 && Cin >= h->h3_min_c) {
            std::vector<char> p3(conv_h3_packed_bytes(Cout, Cin, k));
            rc = conv_h3_pack(data, Cout, Cin, k, p3.data());
            if (rc) return rc;
            void* d = nullptr;
            HIP_TRY(hipMalloc(&d, p3.size()));
            h->owned.push_back((float*)d);
            HIP_TRY(hipMemcpy(d, p3.data(), p3.size(), hipMemcpyHostToDevice));
            L->w3 = d;
        }
        return upload(h, pk.data(), pk.size(), &L->w);
    }
    if (!strcmp(what, "bias")) {
        if (ndim != 1 || shape[0] != Cout) { itts_set_error("conv bias shape mismatch"); return ITTS_ERR_ARG; }
        L->has_b = true;
        return upload(h, data, Cout, &L->b);
    }
    itts_set_error("unknown conv tensor '%s'", what);
    return ITTS_ERR_ARG;
}

static int load_act(itts_bigvgan* h, ActL* A, const char* what, const float* data, const int64_t* shape, int ndim, int C) {
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
    A->C = C;
    if (!strcmp(what, "act.alpha") || !strcmp(what, "act.beta")) {
        if ((int)n != C) { itts_set_error("activation param has %zu elements, want %d", n, C); return ITTS_ERR_ARG; }
        if (what[4] == 'a') { A->has_a = true; return upload(h, data, n, &A->alpha); }
        A->has_b = true;
        return upload(h, data, n, &A->beta);
    }
    if (!strcmp(what, "upsample.filter") || !strcmp(what, "downsample.lowpass.filter")) {
        if (n != 12) { itts_set_error("anti-alias filter must have 12 taps, got %zu", n); return ITTS_ERR_ARG; }
        if (what[0] == 'u') { A->has_fu = true; return upload(h, data, n, &A->fu); }
        A->has_fd = true;
        return upload(h, data, n, &A->fd);
    }
    itts_set_error("unknown activation tensor '%s'", what);
    return ITTS_ERR_ARG;
}

extern "C" int itts_bigvgan_load_tensor(itts_bigvgan* h, const char* name, const float* data, const int64_t* shape,
                                        int ndim) {
    if (!h || !name || !data || !shape || ndim < 1 || ndim > 3) { itts_set_error("load_tensor: bad args"); return ITTS_ERR_ARG; }
    ItDevGuard dg(h->device);
    const itts_bigvgan_config& c = h->cfg;
    const char* s = name;
    int i, j, d;
    h->finalized = false;
    if (eat(s, "conv_pre.")) return load_conv(h, &h->conv_pre, s, data, shape, ndim, c.upsample_initial_channel, c.in_channels, 7);
    if (eat(s, "conv_post.")) {
        const int ch = stage_channels(c, c.num_upsamples - 1);
        if (!strcmp(s, "weight")) {
            if (ndim != 3 || shape[0] != 1 || shape[1] != ch || shape[2] != 7) { itts_set_error("conv_post.weight shape"); return ITTS_ERR_ARG; }
            h->has_post_w = true;
            return upload(h, data, (size_t)ch * 7, &h->post_w);
        }
        if (!strcmp(s, "bias")) { h->has_post_b = true; return upload(h, data, 1, &h->post_b); }
    }
    if (eat(s, "ups.")) {
        if (!parse_idx(s, &i) || i >= c.num_upsamples || !eat(s, ".0.")) { itts_set_error("bad name %s", name); return ITTS_ERR_ARG; }
        const int Cin = c.upsample_initial_channel >> i, Cout = stage_channels(c, i);
        const int k = c.upsample_kernel_sizes[i], u = c.upsample_rates[i];
        ConvL* L = &h->ups[i];
        if (!strcmp(s, "weight")) {
            if (ndim != 3 || shape[0] != Cin || shape[1] != Cout || shape[2] != k) { itts_set_error("%s: shape mismatch", name); return ITTS_ERR_ARG; }
            const size_t per = itts_packed_conv_floats(Cout, Cin, k / u);
            std::vector<float> pk(per * u);
            for (int r = 0; r < u; ++r) {
                int rc = itts_pack_convT_weight(data, Cin, Cout, k, u, r, pk.data() + per * r);
                if (rc) return rc;
            }
            L->Cin = Cin; L->Cout = Cout; L->k = k; L->has_w = true;
            return upload(h, pk.data(), pk.size(), &L->w);
        }
        if (!strcmp(s, "bias")) {
            if (ndim != 1 || shape[0] != Cout) { itts_set_error("%s: shape mismatch", name); return ITTS_ERR_ARG; }
            L->has_b = true;
            return upload(h, data, Cout, &L->b);
        }
    }
    if (eat(s, "resblocks.")) {
        if (!parse_idx(s, &i) || i >= c.num_upsamples * c.num_kernels || !eat(s, ".")) { itts_set_error("bad name %s", name); return ITTS_ERR_ARG; }
        const int ch = stage_channels(c, i / c.num_kernels);
        const int k = c.resblock_kernel_sizes[i % c.num_kernels];
        if (eat(s, "convs1.")) {
            if (!parse_idx(s, &d) || d >= c.num_dilations || !eat(s, ".")) { itts_set_error("bad name %s", name); return ITTS_ERR_ARG; }
            return load_conv(h, &h->convs1[(size_t)i * c.num_dilations + d], s, data, shape, ndim, ch, ch, k, true);
        }
        if (eat(s, "convs2.")) {
            if (!parse_idx(s, &d) || d >= c.num_dilations || !eat(s, ".")) { itts_set_error("bad name %s", name); return ITTS_ERR_ARG; }
            return load_conv(h, &h->convs2[(size_t)i * c.num_dilations + d], s, data, shape, ndim, ch, ch, k, true);
        }
        if (eat(s, "activations.")) {
            if (!parse_idx(s, &j) || j >= 2 * c.num_dilations || !eat(s, ".")) { itts_set_error("bad name %s", name); return ITTS_ERR_ARG; }
            return load_act(h, &h->acts[(size_t)i * 2 * c.num_dilations + j], s, data, shape, ndim, ch);
        }
    }
    if (eat(s, "activation_post.")) return load_act(h, &h->act_post, s, data, shape, ndim, stage_channels(c, c.num_upsamples - 1));
    if (c.cond_dim > 0 && eat(s, "cond_layer.")) {
        ConvL* L = &h->cond_layer;
        const int Cout = c.upsample_initial_channel;
        if (!strcmp(s, "weight")) { L->has_w = true; L->Cout = Cout; return upload(h, data, (size_t)Cout * c.cond_dim, &L->w); }
        if (!strcmp(s, "bias")) { L->has_b = true; return upload(h, data, Cout, &L->b); }
    }
    if (c.cond_dim > 0 && eat(s, "conds.")) {
        if (!parse_idx(s, &i) || i >= c.num_upsamples || !eat(s, ".")) { itts_set_error("bad name %s", name); return ITTS_ERR_ARG; }
        ConvL* L = &h->conds[i];
        const int Cout = stage_channels(c, i);
        if (!strcmp(s, "weight")) { L->has_w = true; L->Cout = Cout; return upload(h, data, (size_t)Cout * c.cond_dim, &L->w); }
        if (!strcmp(s, "bias")) { L->has_b = true; return upload(h, data, Cout, &L->b); }
    }
    itts_set_error("bigvgan_load_tensor: unrecognised tensor name '%s'", name);
    return ITTS_ERR_ARG;
}

// Kaiser-windowed sinc, 12 taps, cutoff 0.25, half-width 0.3 (filter.py:30-62) for checkpoints without filter buffers.
static double bessel_i0(double x) {
    double sum = 1.0, term = 1.0;
    for (int k = 1; k < 64; ++k) {
        term *= (x / (2.0 * k)) * (x / (2.0 * k));
        sum += term;
        if (term < 1e-18 * sum) break;
    }
    return sum;
}
static void default_filter12(float* f) {
    const int K = 12, half = 6;
    const double cutoff = 0.25, half_width = 0.3, pi = 3.14159265358979323846;
    const double delta_f = 4 * half_width;
    const double A = 2.285 * (half - 1) * pi * delta_f + 7.95;
    double beta = 0.0;
    if (A > 50.0) beta = 0.1102 * (A - 8.7);
    else if (A >= 21.0) beta = 0.5842 * pow(A - 21.0, 0.4) + 0.07886 * (A - 21.0);
    double v[12], sum = 0;
    for (int n = 0; n < K; ++n) {
        const double r = 2.0 * n / (K - 1) - 1.0;
        const double win = bessel_i0(beta * sqrt(1.0 - r * r)) / bessel_i0(beta);
        const double t = (n - half) + 0.5;
        const double xx = 2 * cutoff * t;
        const double sinc = xx == 0 ? 1.0 : sin(pi * xx) / (pi * xx);
        v[n] = 2 * cutoff * win * sinc;
        sum += v[n];
    }
    for (int n = 0; n < K; ++n) f[n] = (float)(v[n] / sum);
}

extern "C" int itts_bigvgan_finalize(itts_bigvgan* h) {
    if (!h) { itts_set_error("finalize: null"); return ITTS_ERR_ARG; }
    ItDevGuard dg(h->device);
    const itts_bigvgan_config& c = h->cfg;
    std::string missing;
    auto need = [&](bool ok, const std::string& n) { if (!ok) missing += n + " "; };
    need(h->conv_pre.has_w, "conv_pre.weight");
    need(h->conv_pre.has_b, "conv_pre.bias");
    need(h->has_post_w, "conv_post.weight");
    if (c.use_bias_at_final) need(h->has_post_b, "conv_post.bias");
    for (int i = 0; i < c.num_upsamples; ++i) {
        need(h->ups[i].has_w, "ups." + std::to_string(i) + ".0.weight");
        need(h->ups[i].has_b, "ups." + std::to_string(i) + ".0.bias");
        if (c.cond_dim > 0 && c.cond_in_each_up_layer) need(h->conds[i].has_w && h->conds[i].has_b, "conds." + std::to_string(i));
    }
    if (c.cond_dim > 0) need(h->cond_layer.has_w && h->cond_layer.has_b, "cond_layer");
    for (size_t n = 0; n < h->convs1.size(); ++n) {
        need(h->convs1[n].has_w && h->convs1[n].has_b, "resblocks.convs1#" + std::to_string(n));
        need(h->convs2[n].has_w && h->convs2[n].has_b, "resblocks.convs2#" + std::to_string(n));
    }
    bool need_default = false;
    for (size_t n = 0; n < h->acts.size(); ++n) {
        need(h->acts[n].has_a && (c.activation == 1 || h->acts[n].has_b), "resblocks.activations#" + std::to_string(n));
        if (!h->acts[n].has_fu || !h->acts[n].has_fd) need_default = true;
    }
    need(h->act_post.has_a && (c.activation == 1 || h->act_post.has_b), "activation_post");
    if (!h->act_post.has_fu || !h->act_post.has_fd) need_default = true;
    if (!missing.empty()) {
        itts_set_error("bigvgan_finalize: missing tensors: %s", missing.c_str());
        return ITTS_ERR_STATE;
    }
    if (need_default && !h->default_filter.p) {
        float f[12];
        default_filter12(f);
        int rc = upload(h, f, 12, &h->default_filter);
        if (rc) return rc;
    }
    auto fix = [&](ActL& a) {
        if (!a.has_fu) a.fu = h->default_filter;
        if (!a.has_fd) a.fd = h->default_filter;
        if (c.activation == 1) a.beta = a.alpha;   // Snake: same parameter for frequency and magnitude
    };
    for (auto& a : h->acts) fix(a);
    fix(h->act_post);
    h->finalized = true;
    return ITTS_OK;
}

static size_t max_stage_floats(const itts_bigvgan* h, int B, int T) {
    const itts_bigvgan_config& c = h->cfg;
    size_t mx = (size_t)c.upsample_initial_channel * T;
    size_t t = T;
    for (int i = 0; i < c.num_upsamples; ++i) {
        t *= c.upsample_rates[i];
        size_t v = (size_t)stage_channels(c, i) * t;
        if (v > mx) mx = v;
    }
    return mx * (size_t)B;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t itts_bigvgan_workspace_bytes(const itts_bigvgan* h, int B, int T) {
    if (!h || B <= 0 || T <= 0) return 0;
    const itts_bigvgan_config& c = h->cfg;
    size_t cond = 0;
    if (c.cond_dim > 0) {
        cond = align256((size_t)B * c.upsample_initial_channel * 4);
        for (int i = 0; i < c.num_upsamples; ++i) cond += align256((size_t)B * stage_channels(c, i) * 4);
    }
    return (size_t)(h->conv_mode == 2 ? 9 : h->conv_mode == 1 ? 8 : 7) * align256(max_stage_floats(h, B, T) * sizeof(float)) + cond + 256;
}

extern "C" int itts_bigvgan_forward(itts_bigvgan* h, const float* x, const int32_t* lens, const float* spk, float* wav,
                                    int B, int T, void* workspace, size_t workspace_bytes, void* stream) {
    if (!h || !x || !wav || !workspace) { itts_set_error("bigvgan_forward: null pointer"); return ITTS_ERR_ARG; }
    if (!h->finalized) { itts_set_error("bigvgan_forward: call itts_bigvgan_finalize first"); return ITTS_ERR_STATE; }
    ItDevGuard dg(h->device);
    {
        const int dx = itts_ptr_device(x), dw = itts_ptr_device(workspace);
        if ((dx >= 0 && dx != h->device) || (dw >= 0 && dw != h->device)) {
            itts_set_error("bigvgan_forward: tensors are on device %d/%d but the model was created on device %d", dx, dw, h->device);
            return ITTS_ERR_ARG;
        }
    }
    if (B <= 0 || T <= 0) return ITTS_OK;
    if (B > 65535) { itts_set_error("bigvgan_forward: B must be <= 65535"); return ITTS_ERR_ARG; }
    const itts_bigvgan_config& c = h->cfg;
    if (c.cond_dim > 0 && !spk) { itts_set_error("bigvgan_forward: this model needs a speaker embedding"); return ITTS_ERR_ARG; }
    if (workspace_bytes < itts_bigvgan_workspace_bytes(h, B, T)) {
        itts_set_error("bigvgan_forward: workspace too small (%zu < %zu)", workspace_bytes, itts_bigvgan_workspace_bytes(h, B, T));
        return ITTS_ERR_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t bufsz = align256(max_stage_floats(h, B, T) * sizeof(float));
    char* base = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    const int nbuf = h->conv_mode == 2 ? 9 : h->conv_mode == 1 ? 8 : 7;     // (the three bf16 planes of mode 2 take buffers 7 and 8)
    float* buf[8];
    for (int i = 0; i < 8; ++i) buf[i] = (float*)(base + bufsz * (i < nbuf ? i : 0));
    char* condp = base + bufsz * nbuf;
    float* SP = buf[7];                                  // f16 x 3 mode: the token-major (hi, lo) halves of the conv input
    float *P = buf[0], *X = buf[1], *XS = buf[2], *T1 = buf[3], *T2 = buf[4], *RA = buf[5], *RB = buf[6];
    const int ND = c.num_dilations;
    int rc;

    // speaker-conditioning biases (v1): bias_b[b][co] = cond_i(spk[b])
    const float* cond0 = nullptr;
    std::vector<const float*> cond_up(c.num_upsamples, nullptr);
    if (c.cond_dim > 0) {
        float* p = (float*)condp;
        rc = launch_cond_bias(spk, h->cond_layer.w.p, h->cond_layer.b.p, p, B, c.upsample_initial_channel, c.cond_dim, st);
        if (rc) return rc;
        cond0 = p;
        condp += align256((size_t)B * c.upsample_initial_channel * 4);
        if (c.cond_in_each_up_layer)
            for (int i = 0; i < c.num_upsamples; ++i) {
                float* q = (float*)condp;
                rc = launch_cond_bias(spk, h->conds[i].w.p, h->conds[i].b.p, q, B, stage_channels(c, i), c.cond_dim, st);
                if (rc) return rc;
                cond_up[i] = q;
                condp += align256((size_t)B * stage_channels(c, i) * 4);
            }
    }

    h->recs.clear();
    h->prof_stream = st;
    auto conv_flops = [&](int Cin, int Cout, int k, int t) { return 2.0 * Cin * Cout * k * (double)t * B; };
    auto tensor_bytes = [&](int C, int t) { return 4.0 * C * (double)t * B; };
    // conv_pre (bigvgan.py:362; v1 models.py:224-226)
    { ProfScope ps(h, st, PC_CONV, conv_flops(c.in_channels, c.upsample_initial_channel, 7, T), tensor_bytes(c.in_channels + c.upsample_initial_channel, T));
    rc = conv1d_impl(x, h->conv_pre.w.p, h->conv_pre.b.p, cond0, nullptr, P, B, c.in_channels, c.upsample_initial_channel, T,
                     7, 1, lens, 1, 0, 1.f, st); }
    if (rc) return rc;

    // a resblock conv: exact f32 MFMA, or (opt-in) the activation split into token-major f16 (hi, lo) + the f16 x 3 kernel
    // x3 mode: the activation in front of an x3 conv writes the conv's operand planes itself (aa_act_planes_kernel: no f32 activation tensor, no
    // split pass; bit-identical planes).  Option voc_act_planes = 0 keeps the two-kernel path for the A/B test.
    const bool act_planes = h->conv_mode == 2 && itts_opt(OPT_VOC_ACT_PLANES) != 0 && itts_opt(OPT_AA_ACT) == 2;
    auto x3_conv = [&](const ConvL& L, int ch_, int kk_, int dil_) { return L.w3 && h->conv_mode == 2 && conv_x3_supported(ch_, ch_, kk_, dil_); };
    auto res_conv = [&](const ConvL& L, const float* xin, const float* res, float* yout, int ch_, int t_, int kk_, int dil_, int mult_, int mode_,
                        float div_, bool planes_ready) -> int {
        if (!L.w3 || (h->conv_mode == 2 && !conv_x3_supported(ch_, ch_, kk_, dil_)))
            return conv1d_impl(xin, L.w.p, L.b.p, nullptr, res, yout, B, ch_, ch_, t_, kk_, dil_, lens, mult_, mode_, div_, st);
        if (h->conv_mode == 2) {                         // three token-major bf16 planes + the six-product window kernel
            int rc3 = planes_ready ? ITTS_OK : launch_split_tm3(xin, SP, B, ch_, t_, lens, mult_, st);
            if (rc3) return rc3;
            ConvX3Args g{};
            g.xp = SP; g.wp = L.w3; g.bias = L.b.p; g.res = res; g.y = yout; g.zero_row = h->zero_row; g.lens = lens; g.len_mult = mult_;
            g.B = B; g.Cin = ch_; g.Cout = ch_; g.T = t_; g.k = kk_; g.dil = dil_; g.acc_mode = mode_; g.div = div_;
            return launch_conv_x3(g, st);
        }
        void* sh = SP;
        void* sl = (char*)SP + (size_t)B * t_ * ch_ * 2;
        int rc2 = launch_split_tm(xin, sh, sl, B, ch_, t_, lens, mult_, (int*)((char*)h->zero_row + 128), st);
        if (rc2) return rc2;
        ConvH3Args g{};
        g.xh = sh; g.xl = sl; g.wp = L.w3; g.bias = L.b.p; g.res = res; g.y = yout; g.zero_row = h->zero_row; g.lens = lens; g.len_mult = mult_;
        g.B = B; g.Cin = ch_; g.Cout = ch_; g.T = t_; g.k = kk_; g.dil = dil_; g.acc_mode = mode_; g.div = div_;
        return launch_conv_h3(g, st);
    };

    int t_cur = T, mult = 1;
    for (int i = 0; i < c.num_upsamples; ++i) {
        const int Cin = c.upsample_initial_channel >> i, ch = stage_channels(c, i);
        const int u = c.upsample_rates[i], k = c.upsample_kernel_sizes[i];
        { ProfScope ps(h, st, PC_CONVT, conv_flops(Cin, ch, k / u, t_cur * u), tensor_bytes(Cin, t_cur) + tensor_bytes(ch, t_cur * u));
          rc = convT_impl(P, h->ups[i].w.p, h->ups[i].b.p, cond_up[i], X, B, Cin, ch, t_cur, k, u, lens, mult, st); }
        if (rc) return rc;
        t_cur *= u;
        mult *= u;
        for (int j = 0; j < c.num_kernels; ++j) {
            const int n = i * c.num_kernels + j;
            const int kk = c.resblock_kernel_sizes[j];
            const float* cur = X;
            for (int d = 0; d < ND; ++d) {
                const ActL& a1 = h->acts[(size_t)n * 2 * ND + 2 * d];
                const ActL& a2 = h->acts[(size_t)n * 2 * ND + 2 * d + 1];
                const ConvL& c1 = h->convs1[(size_t)n * ND + d];
                const ConvL& c2 = h->convs2[(size_t)n * ND + d];
                const bool p1 = act_planes && x3_conv(c1, ch, kk, c.resblock_dilations[j][d]), p2 = act_planes && x3_conv(c2, ch, kk, 1);
                { ProfScope ps(h, st, PC_ACT, 0, tensor_bytes(2 * ch, t_cur));
                  rc = p1 ? launch_aa_act_planes(cur, SP, a1.alpha.p, a1.beta.p, a1.fu.p, a1.fd.p, B, ch, t_cur, lens, mult, c.snake_logscale, st)
                          : launch_aa_act(cur, T1, a1.alpha.p, a1.beta.p, a1.fu.p, a1.fd.p, B, ch, t_cur, lens, mult, c.snake_logscale, st); }
                if (rc) return rc;
                { ProfScope ps(h, st, PC_CONV, conv_flops(ch, ch, kk, t_cur), tensor_bytes(2 * ch, t_cur));
                  rc = res_conv(c1, T1, nullptr, T2, ch, t_cur, kk, c.resblock_dilations[j][d], mult, 0, 1.f, p1); }
                if (rc) return rc;
                { ProfScope ps(h, st, PC_ACT, 0, tensor_bytes(2 * ch, t_cur));
                  rc = p2 ? launch_aa_act_planes(T2, SP, a2.alpha.p, a2.beta.p, a2.fu.p, a2.fd.p, B, ch, t_cur, lens, mult, c.snake_logscale, st)
                          : launch_aa_act(T2, T1, a2.alpha.p, a2.beta.p, a2.fu.p, a2.fd.p, B, ch, t_cur, lens, mult, c.snake_logscale, st); }
                if (rc) return rc;
                if (d < ND - 1) {
                    float* nxt = (cur == RA) ? RB : RA;
                    { ProfScope ps(h, st, PC_CONV, conv_flops(ch, ch, kk, t_cur), tensor_bytes(3 * ch, t_cur));
                      rc = res_conv(c2, T1, cur, nxt, ch, t_cur, kk, 1, mult, 0, 1.f, p2); }
                    if (rc) return rc;
                    cur = nxt;
                } else {
                    // block output r_j = conv2(..) + cur, folded into the MRF sum: xs = r_0; xs += r_1; ...; /num_kernels
                    int mode = (j == 0) ? 0 : 1;
                    if (j == c.num_kernels - 1) mode = (c.num_kernels == 1) ? 0 : 2;
                    { ProfScope ps(h, st, PC_CONV, conv_flops(ch, ch, kk, t_cur), tensor_bytes((mode ? 4 : 3) * ch, t_cur));
                      rc = res_conv(c2, T1, cur, XS, ch, t_cur, kk, 1, mult, mode, (float)c.num_kernels, p2); }
                    if (rc) return rc;
                }
            }
        }
        float* tmp = P; P = XS; XS = tmp;
    }
    const int ch = stage_channels(c, c.num_upsamples - 1);
    { ProfScope ps(h, st, PC_ACT, 0, tensor_bytes(2 * ch, t_cur));
      rc = launch_aa_act(P, T1, h->act_post.alpha.p, h->act_post.beta.p, h->act_post.fu.p, h->act_post.fd.p, B, ch, t_cur, lens, mult,
                         c.snake_logscale, st); }
    if (rc) return rc;
    ProfScope ps(h, st, PC_POST, 2.0 * ch * 7 * (double)t_cur * B, tensor_bytes(ch + 1, t_cur));
    return launch_conv_post(T1, wav, h->post_w.p, c.use_bias_at_final ? h->post_b.p : nullptr, B, ch, t_cur, 7, lens, mult,
                            c.use_tanh_at_final, st);
}

// ---- HIP-event profiling of the forward pass ------------------------------------------------------------------
extern "C" int itts_bigvgan_set_profiling(itts_bigvgan* h, int enable) {
    if (!h) { itts_set_error("set_profiling: null"); return ITTS_ERR_ARG; }
    h->profiling = enable != 0;
    h->recs.clear();
    return ITTS_OK;
}

// Totals of the LAST forward per kernel class (0 Conv1d MFMA, 1 ConvTranspose1d MFMA phases, 2 anti-aliased
// activation, 3 conv_post): GPU milliseconds between the events bracketing each launch (recorded on the stream the
// kernels ran on), launch count, algorithmic FLOPs and algorithmic tensor bytes.  Synchronises that stream.
// Per-launch records of the LAST forward, in launch order: out[4*i + {0,1,2,3}] = {class, ms, flops, bytes}.
// Returns the number of launches (<= max_records written).  Synchronises the launch stream.
extern "C" int itts_bigvgan_profile_records(itts_bigvgan* h, double* out, int max_records) {
    if (!h || !out) { itts_set_error("profile_records: null"); return -1; }
    ItDevGuard dg(h->device);
    if (h->recs.empty()) return 0;
    if (hipStreamSynchronize(h->prof_stream) != hipSuccess) return -1;
    int n = 0;
    for (size_t i = 0; i < h->recs.size() && n < max_records; ++i, ++n) {
        float t = 0.f;
        (void)hipEventElapsedTime(&t, h->ev_pool[2 * i], h->ev_pool[2 * i + 1]);
        out[4 * n + 0] = h->recs[i].cls; out[4 * n + 1] = t; out[4 * n + 2] = h->recs[i].flops; out[4 * n + 3] = h->recs[i].bytes;
    }
    return (int)h->recs.size();
}

extern "C" int itts_bigvgan_profile_read(itts_bigvgan* h, double* ms, double* launches, double* flops, double* bytes) {
    if (!h || !ms || !launches || !flops || !bytes) { itts_set_error("profile_read: null"); return ITTS_ERR_ARG; }
    ItDevGuard dg(h->device);
    for (int i = 0; i < PC_COUNT; ++i) ms[i] = launches[i] = flops[i] = bytes[i] = 0;
    if (h->recs.empty()) return ITTS_OK;
    HIP_TRY(hipStreamSynchronize(h->prof_stream));
    for (size_t i = 0; i < h->recs.size(); ++i) {
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, h->ev_pool[2 * i], h->ev_pool[2 * i + 1]));
        const auto& r = h->recs[i];
        ms[r.cls] += t; launches[r.cls] += 1; flops[r.cls] += r.flops; bytes[r.cls] += r.bytes;
    }
    return ITTS_OK;
}


// ================================================================================================================
// Streaming vocoder (BASELINE configs[4]): one utterance, mel frames pushed in chunks, waveform emitted as soon as it is
// final.  Exact overlap-save: a chunk is synthesised together with `halo` frames of real context on both sides (the
// generator's receptive field) and only the centre is emitted, so the concatenated output equals the one-shot forward --
// the reference's streaming path (backends/trt/pipeline/streaming.py:57-172) re-synthesises overlapping chunks and
// Hann-crossfades them instead.  Output lags the input by `halo` frames until the last push.
// ================================================================================================================
struct itts_bigvgan_stream {
    itts_bigvgan* h;
    int chunk, halo, cap;        // cap = frames the context buffer holds
    int received, emitted;       // absolute frame counters
    int base;                    // absolute index of ctx column 0
    float* ctx;                  // [C][cap] the frames [base, received)
    float* xwin;                 // [C][cap] contiguous window handed to the forward
    float* wwin;                 // [cap * total_up]
};

extern "C" int itts_bigvgan_stream_open(itts_bigvgan* h, int chunk_frames, int halo_frames, itts_bigvgan_stream** out) {
    if (!h || !out) { itts_set_error("bigvgan_stream_open: null pointer"); return ITTS_ERR_ARG; }
    if (!h->finalized) { itts_set_error("bigvgan_stream_open: call itts_bigvgan_finalize first"); return ITTS_ERR_STATE; }
    if (chunk_frames <= 0 || halo_frames < 0) { itts_set_error("bigvgan_stream_open: bad chunk/halo"); return ITTS_ERR_ARG; }
    ItDevGuard dg(h->device);
    auto* s = new itts_bigvgan_stream();
    s->h = h; s->chunk = chunk_frames; s->halo = halo_frames; s->cap = 2 * chunk_frames + 2 * halo_frames;
    s->received = s->emitted = s->base = 0;
    s->ctx = s->xwin = s->wwin = nullptr;
    const size_t C = (size_t)h->cfg.in_channels;
    hipError_t e = hipMalloc((void**)&s->ctx, C * s->cap * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&s->xwin, C * s->cap * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&s->wwin, (size_t)s->cap * h->total_up * sizeof(float));
    if (e != hipSuccess) {
        itts_set_error("bigvgan_stream_open: hipMalloc -> %s", hipGetErrorString(e));
        if (s->ctx) (void)hipFree(s->ctx);
        if (s->xwin) (void)hipFree(s->xwin);
        delete s;
        return ITTS_ERR_HIP;
    }
    *out = s;
    return ITTS_OK;
}

extern "C" void itts_bigvgan_stream_close(itts_bigvgan_stream* s) {
    if (!s) return;
    ItDevGuard dg(s->h->device);
    (void)hipFree(s->ctx); (void)hipFree(s->xwin); (void)hipFree(s->wwin);
    delete s;
}

extern "C" size_t itts_bigvgan_stream_workspace_bytes(const itts_bigvgan_stream* s) {
    return s ? itts_bigvgan_workspace_bytes(s->h, 1, s->cap) : 0;
}

extern "C" int itts_bigvgan_stream_push(itts_bigvgan_stream* s, const float* mel, int ld_mel, int n_frames, int is_last,
                                        const float* spk, float* wav_out, int32_t* n_samples_out, void* workspace,
                                        size_t workspace_bytes, void* stream) {
    if (!s || !wav_out || !n_samples_out || !workspace || (n_frames > 0 && !mel)) { itts_set_error("bigvgan_stream_push: null pointer"); return ITTS_ERR_ARG; }
    if (n_frames < 0 || n_frames > s->chunk || (n_frames > 0 && ld_mel < n_frames)) { itts_set_error("bigvgan_stream_push: n_frames %d outside 0..%d", n_frames, s->chunk); return ITTS_ERR_ARG; }
    ItDevGuard dg(s->h->device);
    hipStream_t st = (hipStream_t)stream;
    const int C = s->h->cfg.in_channels, up = s->h->total_up;
    *n_samples_out = 0;
    if (n_frames > 0) {                                      // append [C][n_frames] at column received - base
        const int col = s->received - s->base;
        HIP_TRY(hipMemcpy2DAsync(s->ctx + col, (size_t)s->cap * 4, mel, (size_t)ld_mel * 4, (size_t)n_frames * 4, C,
                                 hipMemcpyDeviceToDevice, st));
        s->received += n_frames;
    }
    const int emit_to = is_last ? s->received : s->received - s->halo;     // frames whose right context is complete
    if (emit_to <= s->emitted) return ITTS_OK;
    const int w0 = (s->emitted - s->halo) > s->base ? (s->emitted - s->halo) : s->base;   // window [w0, w1)
    const int w1 = (emit_to + s->halo) < s->received ? (emit_to + s->halo) : s->received;
    const int Tw = w1 - w0;
    HIP_TRY(hipMemcpy2DAsync(s->xwin, (size_t)Tw * 4, s->ctx + (w0 - s->base), (size_t)s->cap * 4, (size_t)Tw * 4, C,
                             hipMemcpyDeviceToDevice, st));
    int rc = itts_bigvgan_forward(s->h, s->xwin, nullptr, spk, s->wwin, 1, Tw, workspace, workspace_bytes, stream);
    if (rc) return rc;
    const int n_out = (emit_to - s->emitted) * up;
    HIP_TRY(hipMemcpyAsync(wav_out, s->wwin + (size_t)(s->emitted - w0) * up, (size_t)n_out * 4, hipMemcpyDeviceToDevice, st));
    *n_samples_out = n_out;
    s->emitted = emit_to;
    // keep [emitted - halo, received): slide the context to column 0 (through xwin: source and destination overlap)
    const int nb = (s->emitted - s->halo) > s->base ? (s->emitted - s->halo) : s->base;
    if (nb > s->base) {
        const int keep = s->received - nb;
        if (keep > 0) {
            HIP_TRY(hipMemcpy2DAsync(s->xwin, (size_t)keep * 4, s->ctx + (nb - s->base), (size_t)s->cap * 4, (size_t)keep * 4, C,
                                     hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemcpy2DAsync(s->ctx, (size_t)s->cap * 4, s->xwin, (size_t)keep * 4, (size_t)keep * 4, C,
                                     hipMemcpyDeviceToDevice, st));
        }
        s->base = nb;
    }
    return ITTS_OK;
}
