// Launcher prototypes for the GPT decoder kernels (gpt_kernels.hip), shared with capi_gpt.hip.
#pragma once
#include "common.h"

enum { PREC_F32 = 0, PREC_BF16 = 1,
       PREC_F32X3 = 2 };   // f32 activations; GEMMs on the bf16 matrix pipe with every f32 operand carried exactly as three bf16 planes
                           // (gpt_kernels.hip::gemm_x3_kernel).  Everything that is not a GEMM runs its f32 code.

// ---- LayerNorm with fused split-K reduce / bias / residual update ---------------------------------------------
struct LnArgs {
    float* x;                // [rows_in][D] residual stream (updated in place when partial/bias_prev given)
    const float* partial;    // [nsplit][rows][D] f32 partial GEMM outputs or null
    int nsplit;
    const float* bias_prev;  // [D] bias of the GEMM whose partials are being reduced, or null
    const float* g1; const float* b1;   // first LayerNorm affine
    const float* g2; const float* b2;   // optional second LayerNorm (ln_f -> final_norm), null to skip
    void* out;               // [rows][D] act dtype (bf16 / f32) or f32 when out_f32 != 0
    int out_f32;
    int rows, D;
    int in_row_mul, in_row_add;   // input row of output row r = r*mul + add (gather of last rows); 1,0 = identity
    float eps;
};
int launch_ln(const LnArgs& a, int prec, hipStream_t st);

// ---- GEMM  C[M,N] = A[M,K] * W[K,N]  on MFMA ------------------------------------------------------------------
enum { EPI_STORE_F32 = 0, EPI_GELU_ACT = 1, EPI_PARTIAL = 2, EPI_RESIDUAL = 3, EPI_QKV = 4,
       // bf16 prefill-tile kernel only (the s2mel DiT, gpt_kernels.hip::pf_epilogue_pair / pf_epilogue):
       EPI_SWIGLU = 5,      // W packed with n-tiles interleaved (2j: w1 columns 16j.., 2j+1: w3 columns 16j..): out_act[m][n] = silu(a) * b
       EPI_GATE = 6,        // same interleave of the WaveNet in_layer halves: out_act[m][n] = tanh(a + bias + g) * sigmoid(b + bias' + g')
       EPI_QKV_ROPE = 7,    // fused wqkv: RoPE on q / k, Q -> out_act [m][D], K -> kcache [seq][H][Tmax][64], V^T -> vcache [seq][H][64][Tmax]
       EPI_WN_RS = 8 };     // WaveNet res_skip: n < D: out_f32[m][n] = (out_f32[m][n] + v) * mask(m); n >= D: out2[m][n - D] (=|+=) v
struct GemmArgs {
    const void* A; int lda;          // act dtype [M][lda]
    const void* Wp;                  // packed weights (itts_pack_gemm_weight)
    const float* bias;               // [N] or null
    int M, N, K;
    int nsplit;                      // K slices (EPI_PARTIAL), else 1
    int epi;
    float* out_f32; int ldo;         // EPI_STORE_F32 / EPI_RESIDUAL (in-place add)
    void* out_act;                   // EPI_GELU_ACT: act dtype [M][ldo]
    float* partial;                  // EPI_PARTIAL: [nsplit][M][N]
    // EPI_QKV: n < D -> qbuf[m][n]; D <= n < 2D -> K cache; else V cache, at cache position *pos_ptr + (m % S) [- pos_shift[cache row]]
    float* qbuf; void* kcache; void* vcache;
    const int* pos_ptr; int S, H, Tmax, D;
    const int* pos_shift;            // [cache rows] or null (decode steps only): the batch's position counter runs ahead of cache row r's OWN position by
                                     // pos_shift[r] -- a row admitted into a running batch (itts_gpt_admit_rows) keeps its keys at its own positions
    size_t a_planes;                 // f32x3 tile GEMM: when non-zero, A is THREE bf16 planes (plane p at (u16*)A + p * a_planes, rows of lda elements, each
                                     // 32-column group stored in fragment order: ada_rmsnorm_planes_kernel) instead of f32 rows -- no in-register split
    int kb_slice;                    // filled by the decode-GEMM launcher: 32-wide k-blocks per K slice
    int dma_rot;                     // per-block rotation of the slab DMA issue order (ITTS_DECODE_ROT=0 turns it off: A/B switch)
    // LayerNorm fused into a decode GEMM's operand staging (gemm_decode_ln_kernel: at most 4 rows, K == model_dim, bf16): when ln_x is set the
    // A operand is LayerNorm(ln_x [+ the 4 split-K partials ln_partial + ln_bias_prev]) computed by every block (one wave per row, ln_kernel's
    // arithmetic) straight into the LDS slab; block 0 also stores the updated residual rows to ln_x_out (a DIFFERENT buffer: the other
    // blocks still read ln_x).  A / lda are unused then.
    const float* ln_x; float* ln_x_out; const float* ln_partial; const float* ln_bias_prev; const float* ln_g; const float* ln_b; float ln_eps;
    // s2mel epilogues (packed token rows): sequence / frame of a row, valid frames per sequence, RoPE table [t][32][2],
    // per-step conditioning vector (EPI_GATE), second f32 output (EPI_WN_RS) with its overwrite / last-layer switches
    const int* tok_seq; const int* tok_t; const int* seq_len; const float* rope; const float* gvec;
    float* out2; int wn_first, wn_last;
    // tap mode of the bf16 tile kernel (conv_taps > 0): the A operand is an IMPLICIT im2col of a [n_tok][conv_W] matrix -- K
    // index j * conv_W + c reads row src(m, j) = the frame t + j * conv_dil - left of m's sequence, reflect-padded at the
    // sequence ends (SConv1d, encodec.py:212-228); a source outside the sequence (zero extension of very short inputs) reads
    // `zero_row`.  K = conv_taps * conv_W, lda = conv_W.
    int conv_taps, conv_dil, conv_W;
    const int* seq_start; const int* seq_T; const void* zero_row;
    void* out_act2;                  // EPI_WN_RS / EPI_STORE_F32 / EPI_RESIDUAL (bf16 tile kernels): bf16 shadow copy [M][ldo] of the f32
                                     // output (the next GEMM's A operand; for EPI_WN_RS the next layer's tap-mode operand, stride D)
    int seq_mul;                     // EPI_QKV: cache row of sequence b is b * seq_mul (0/1 = identity); beam prefill writes only row b*nb
    const int* seq_map;              // EPI_QKV (decode): when non-null, the cache row of dense row b is seq_map[b] (row compaction of a
                                     // ragged batch: finished rows leave the running batch, the survivors keep their cache rows)
    size_t kv_planes;                // EPI_QKV_ROPE, f32 / f32x3 kernels: when non-zero, K and V^T are written as THREE bf16 planes (h, m, l with
                                     // h + m + l == the f32 value exactly; plane p at element offset p * kv_planes of kcache / vcache, each plane in
                                     // the bf16 mode's image) -- the operands of flash_attn_x3_kernel (s2mel_kernels.hip)
};
int launch_gemm(const GemmArgs& a, int prec, bool prefill, hipStream_t st);
int launch_gemm_x3(const GemmArgs& a, hipStream_t st);      // gemm_x3.hip: the fp32x3 tile kernel (its own translation unit, built without SLP vectorisation)
int gemm_x3_occupancy(int* blocks);
bool gemm_decode_ln_ok(int M, int K, int epi);                 // shapes of the LayerNorm-fused decode GEMM (bf16, 1-4 rows)
int launch_gemm_decode_ln(const GemmArgs& a, hipStream_t st);  // A = LayerNorm(ln_x [+ ln_partial + ln_bias_prev]) built inside the kernel
int gemm_tile_occupancy(int prec, int* blocks);      // diagnostics: predicted resident blocks per CU of the 128 x 128 tile kernel

// ---- attention over the KV cache ------------------------------------------------------------------------------
struct AttnArgs {
    const float* qbuf;       // [nseq*nq][D]
    const void* kcache; const void* vcache;   // [nseq][H][Tmax][64] cache dtype
    const int* pad;          // [nseq] first valid key (left padding), or null
    const int* row_map;      // [nseq][Tmax] physical row holding position t of sequence b (beam indirection) or null
    const int* row_map_alt;  // second buffer: the map in use is (*step_ptr & 1) ? row_map_alt : row_map
    const int* step_ptr;
    const int* pos_ptr;      // cache index of query 0
    const int* pos_shift;    // [cache rows] or null: query 0 of cache row r sits at *pos_ptr - pos_shift[r] (GemmArgs::pos_shift)
    void* out;               // [nseq*nq][D] act dtype
    int nseq, H, nq, Tmax, D;
    int seq_mul;             // sequence b reads cache row / pad entry b * seq_mul when no row map is given (0/1 = identity)
    const int* seq_map;      // when non-null (compacted decode batch): dense row b reads cache row / pad entry seq_map[b]
};
int launch_attention(const AttnArgs& a, int prec, hipStream_t st);

// ---- token selection ------------------------------------------------------------------------------------------
struct SampleArgs {
    const float* logits;     // [B][V]
    unsigned char* seen;     // [B][V] ids already in input_ids (repetition penalty set)
    unsigned char* finished; // [B]
    long long* tokens;       // [B][max_new]
    const int* step_ptr;     // generated-token index of this step
    const double* uniforms;  // [max_new][B] or null (then an internal counter RNG with `seed` is used)
    unsigned long long seed;
    int B, V, max_new;
    int do_sample, top_k, min_keep;
    float top_p, temperature, rep_penalty;
    float typical_mass;      // 0 = off, else TypicalLogitsWarper(mass) between the repetition penalty and the warpers
    int stop_token;
    // next-step embedding: x_next[b] = mel_emb[tok] + mel_pos[step + pos_offset]
    const float* mel_emb; const float* mel_pos; float* x_next; int D; int pos_offset; int n_mel_pos;
    int* adv_state;          // {step, pos, ticket}: when non-null the last block to finish advances step and pos (decode steps)
    const unsigned long long* seed_ptr;   // when non-null the RNG seed is read from device memory (keeps a captured step seed-free)
    // compacted decode batch: dense row b is utterance row_slot[b] -- its seen-set, finished flag, token row, uniform / RNG stream and
    // token limit are indexed by the utterance, logits and x_next by the dense row.  uniforms_stride = utterances of the call.
    const int* row_slot; int uniforms_stride;
    int radix_select;        // filled by the launcher (ITTS_SAMPLE_RADIX=1): the 4-pass radix-select top-k instead of the ballot bisection (A/B switch)
    unsigned long long* stamps;   // microbench builds only (-DITTS_SAMPLE_STAMPS): [B][8] phase time stamps, else null
    const int* row_limit;    // [utterances] or null: per-utterance cap on generated tokens (a batch merges requests with their own
                             // max_mel_tokens): from token index row_limit[u] on, the row emits the stop token
    const int* row_step0;    // [utterances] or null: the step at which utterance u joined the running batch (itts_gpt_admit_rows; 0 for the rows of
                             // the first call).  The row's own step (step - row_step0[u]) indexes its token column, its mel position embedding,
                             // its uniform / RNG stream and its token limit -- what the row would see decoded alone; from its own step max_new
                             // on a row emits the stop token and stores nothing (a session's step counter may run past max_new).
};
int launch_sample(const SampleArgs& a, hipStream_t st);
int launch_advance(int* step_ptr, int* pos_ptr, hipStream_t st);

// ---- beam search / beam-sample step (GenerationMixin._beam_search + BeamSearchScorer.process) ------------------
#define BEAM_MAX 4
struct BeamHyp { float score; int step; int row; int pad; };
struct BeamArgs {
    const float* logits;          // [B*nb][V]
    unsigned char* seen[2];       // [B*nb][V], parity = step & 1 (read), 1 - parity (written by apply)
    int* row_map[2];              // [B*nb][Tmax]
    float* beam_scores;           // [B*nb]
    float* next_scores; int* next_tokens; int* next_indices;   // [B*nb]
    int* surv_idx; float* surv_val; int* surv_n;                // [B*nb][64], [B*nb]: per-row survivors (phase 1 -> phase 2)
    BeamHyp* hyps;                // [B][BEAM_MAX]
    int* n_hyps;                  // [B]
    float* worst;                 // [B]
    unsigned char* done;          // [B]
    int* hist_tok; int* hist_par; // [max_new][B*nb]
    const int* step_ptr;
    int logits_shared;            // 1: logits hold ONE row per utterance (first step after the shared-prompt prefill)
    const double* uniforms;       // [max_new][B][2*nb] or null
    unsigned long long seed;
    int B, nb, V, max_new, Tmax, S;
    int do_sample, top_k, min_keep;
    float top_p, temperature, rep_penalty, length_penalty;
    float typical_mass;
    int stop_token;
    const float* mel_emb; const float* mel_pos; float* x_next; int D; int pos_offset; int n_mel_pos;
    int* adv_state;               // as SampleArgs::adv_state, honoured by the apply kernel (the step's last launch)
    const unsigned long long* seed_ptr;   // as SampleArgs::seed_ptr
    int radix_select;             // as SampleArgs::radix_select (filled by the launcher)
};
int launch_beam_step(const BeamArgs& a, hipStream_t st);
int launch_beam_apply(const BeamArgs& a, hipStream_t st);
