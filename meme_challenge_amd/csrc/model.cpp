// Whole-model kernel schedule of the UNITER encoder: the library owns the
// sequence of launches for UniterModel.forward (model/model.py:336-367,
// BertLayer.forward model/layer.py:166-170) and for its backward, so the host
// makes a handful of calls per training step instead of ~60 framework
// dispatches per layer.
//
// Streams: the forward and the backward dgrad chain run on `stream`; weight /
// bias gradient GEMMs of layer l run on `side_stream` behind an event, so they
// overlap layer l-1's dgrad chain and fill the CUs that the partial last wave
// of each GEMM leaves idle.  Every layer owns its backward scratch, so the two
// streams never race on a buffer (288 GB of HBM: ~3 GB for UNITER-base B=16).
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>
#include "common.h"
#include "philox.h"

// internal helpers implemented in other translation units
int launch_add_f32(float* out, const float* a, const float* b, size_t n, hipStream_t st);
int gemm_f32_run(int cfg, int tag, int a_kmajor, int b_kmajor, int M, int N, int K, const float* A, int lda,
                 const float* B, int ldb, float* C, int ldc, int epilogue, const float* bias,
                 const float* aux_in, float* aux_out, int ld_aux, int beta, float* colsum_part, void* stream);
int gemm_bf16_run(int cfg, int a_kmajor, int b_kmajor, int M, int N, int K, const float* A, int lda,
                  const float* B, int ldb, float* C, int ldc, int epilogue, const float* bias,
                  const float* aux_in, float* aux_out, int ld_aux, int beta, float* colsum_part, void* stream);
int gemm_bf16res_run(int cfg, int a_kmajor, int b_kmajor, int M, int N, int K, const void* A, int lda, const void* B,
                     int ldb, float* C, int ldc, void* Cb, int ldcb, int epilogue, const float* bias,
                     const float* aux_in, float* aux_out, int ld_aux, int beta, float* colsum_part, void* stream);
int gemm_bf16v2_run(int cfg, int nsplit, int a_kmajor, int b_kmajor, int M, int N, int K, const void* A, int lda,
                    const void* B, int ldb, float* C, int ldc, long c_split_stride, void* Cb, int ldcb, int epilogue,
                    const float* bias, const void* aux_in, int aux_in_bf16, void* aux_out, int aux_out_bf16,
                    int ld_aux, int beta, void* stream);
int gemm_f32_wgrad_group(int n, const int* Mo, const int* No, int K, const float* const* A, const float* const* B,
                         float* const* dW, int overwrite, void* stream);
int gemm_x3_run(int cfg, int nsplit, int a_kmajor, int b_kmajor, int M, int N, int K, const void* A, int lda, int psa,
                const void* B, int ldb, int psb, float* C, int ldc, long c_split_stride, void* Cx, int ldcx, int pscx,
                int epilogue, const float* bias, const float* aux_in, float* aux_out, int ld_aux, void* stream,
                float* colsum_part, void* sk_ws, size_t sk_ws_bytes);
size_t gemm_x3_sk_ws_bytes();
int gemm_x3_wgrad_default_cfg();
int gemm_x3_pick_split(int M, int N, int K);
int gemm_x3_pick_split_on(int M, int N, int K, int avail);
int gemm_x3_wgrad_group(int cfg, int n, const int* Mo, const int* No, int K, const void* const* A, const void* const* B,
                        float* const* dW, void* stream, int overwrite, int max_wgs, uniter_x3_riders_t* riders,
                        void* sk_ws, size_t sk_ws_bytes);
int gemm_x3_wgrad_group_slots(int cfg, int n, const int* Mo, const int* No, int max_wgs, int K, size_t sk_ws_bytes);
int gemm_x3_wgrad_group_balanced_wgs(int cfg, int n, const int* Mo, const int* No);
int gemm_bf16v2_pick_split(int M, int N, int K);
int gemm_b1p_pick_split(int M, int N, int K, int avail);
int gemm_b1p_run(int cfg, int nsplit, int b_kmajor, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                 float* C, int ldc, long c_split_stride, void* Cb, int ldcb, int epilogue, const float* bias,
                 const void* aux_in, int aux_in_bf16, void* aux_out, int aux_out_bf16, int ld_aux, float* colpart, void* stream);
int gemm_b1p_wgrad_group_slots(int n, const int* Mo, const int* No, int max_wgs);
int gemm_bf16v2_wgrad_pieces(int M, int N, int K);
int gemm_bf16v2_wgrad_group(int cfg, int n, const int* Mo, const int* No, int K, const void* const* A,
                            const void* const* B, float* const* dW, void* stream, int overwrite, int max_wgs,
                            uniter_x3_riders_t* riders);
int gemm_bf16v2_wgrad_group_slots(int n, const int* Mo, const int* No, int max_wgs);
int gemm_bf16v2_wgrad_group_balanced_wgs(int n, const int* Mo, const int* No);
int finalize_partials(const float* part, int nparts, size_t stride, float* out, int N, int beta, hipStream_t st);
int finalize_partials_jobs(int njobs, const float* const* part, const int* nparts, const size_t* stride,
                           float* const (*outs)[3], const int* nout, const int* seg, hipStream_t st);
int ln_bwd_partial_rows(int M);
int launch_masked_rowsum(const float* x, const int64_t* masks, float* out, int rows, int D, hipStream_t st);

namespace {

enum { P_WORD = 0, P_POS, P_TYPE, P_ELN_G, P_ELN_B, P_IMG_W, P_IMG_B, P_ILN_G, P_ILN_B, P_PLN_G, P_PLN_B,
       P_POSL_W, P_POSL_B, P_MASK_EMB, P_FLN_G, P_FLN_B, P_LAYER0 };
enum { L_QW = 0, L_QB, L_KW, L_KB, L_VW, L_VB, L_OW, L_OB, L_LN1_G, L_LN1_B, L_W1, L_B1, L_W2, L_B2, L_LN2_G,
       L_LN2_B, L_COUNT };

const char* kEmbNames[P_LAYER0] = {
    "embeddings.word_embeddings.weight", "embeddings.position_embeddings.weight",
    "embeddings.token_type_embeddings.weight", "embeddings.LayerNorm.weight", "embeddings.LayerNorm.bias",
    "img_embeddings.img_linear.weight", "img_embeddings.img_linear.bias",
    "img_embeddings.img_layer_norm.weight", "img_embeddings.img_layer_norm.bias",
    "img_embeddings.pos_layer_norm.weight", "img_embeddings.pos_layer_norm.bias",
    "img_embeddings.pos_linear.weight", "img_embeddings.pos_linear.bias",
    "img_embeddings.mask_embedding.weight", "img_embeddings.LayerNorm.weight", "img_embeddings.LayerNorm.bias"};
const char* kLayerNames[L_COUNT] = {
    "attention.self.query.weight", "attention.self.query.bias", "attention.self.key.weight",
    "attention.self.key.bias", "attention.self.value.weight", "attention.self.value.bias",
    "attention.output.dense.weight", "attention.output.dense.bias", "attention.output.LayerNorm.weight",
    "attention.output.LayerNorm.bias", "intermediate.dense.weight", "intermediate.dense.bias",
    "output.dense.weight", "output.dense.bias", "output.LayerNorm.weight", "output.LayerNorm.bias"};

struct Carver {
  char* base; size_t off;
  explicit Carver(void* b) : base((char*)b), off(0) {}
  float* f(size_t n) { float* p = (float*)(base + off); off += align_up(n * sizeof(float), 256); return p; }
  unsigned short* h(size_t n) { unsigned short* p = (unsigned short*)(base + off); off += align_up(n * 2, 256); return p; }
  void* raw(size_t n) { void* p = base + off; off += align_up(n, 256); return p; }
};

struct LayerBufs {   // saved activations + backward scratch of one layer
  float *qkv, *lse, *ctx, *t1, *z1, *mean1, *rstd1, *y1, *u, *hact, *t2, *z2, *mean2, *rstd2, *y2;
  float *dz2, *g2, *du, *dy1, *dz1, *g1, *dctx, *dqkv, *delta, *dx, *du_csum;
  unsigned short* keepb;      // dropout keep flags of the attention probabilities (forward -> dQ)
  float* qb_part;             // [B, 3H] per-sample column sums of dqkv from the attention backward kernels
  void *ln_ws1, *ln_ws2;      // column partials of the two LayerNorm backward passes (finalized on the side stream)
  // precision 2: bf16 copies that feed the bf16-resident GEMMs (hact and du exist only in bf16 there)
  // precision 3: the same tensors as x3 pieces [rows][3][cols] (three times the elements) for the fp32-accurate products
  unsigned short *ctxb, *y1b, *hactb, *y2b, *g2b, *dub, *g1b, *dqkvb;
};

struct Plan {
  int B, T, R, L, S, T0, M, mode;
  bool has_txt, has_img;
  bool gelu_d;      // forward stored gelu'(u) in LayerBufs::u rather than u
  float *feat_eff, *imgfc, *img_stats, *cat, *emb;
  std::vector<LayerBufs> layers;
  float *dcat, *d_imgfc, *d_posfc, *d_feat, *dsum, *dpk;
  unsigned short* embb;   // precision 2: bf16 copy of the embedding output
  bool res;               // this plan was carved for precision 2
  bool x3;                // this plan was carved for precision 3: the *b buffers hold x3 pieces [rows][3][cols], embb likewise
  int ns_kh;              // precision 3: k-pieces of the attention-output forward product (N = K = hidden)
  // precision 3: k-pieces of the BACKWARD pass's N = hidden products (FFN-up / attention-output / QKV input gradients), chosen for the
  // CUs the backward pass may use: all of them, or what a data-parallel exchange leaves (uniter_model_set_cu_reserve)
  int ns_ki_b, ns_kh_b, ns_k3h_b;
  // precision 2: k-pieces of the GEMMs whose N is the hidden size (their fp32 outputs are that many slabs, summed by
  // the LayerNorm row pass that consumes them): K = intermediate (FFN-down forward, FFN-up dgrad), K = 3 hidden (QKV dgrad)
  int ns_ki, ns_k3h;
  bool packed;      // rows = valid positions only (uniter_batch_t::cu_seqlens)
  int wg_group;           // precision 2: the layer's four weight gradients as one whole-K-tile launch (0 = stream-K, 1 / 4 = LDS stages cfg)
  float* wg_slabs;        // precision 2: k-piece slabs of the split-K weight-gradient GEMMs (side stream, reused by every layer)
  unsigned char* hid_keepb;   // hidden-dropout keep flags of every layer's two sites, drawn ahead (round 5); stride hid_stride per site
  size_t hid_stride;
  bool hid_on;                // the forward pass drew them (this plan's row passes read them, forward and backward)
  void *ln_ws, *col_ws, *emb_ws, *attn_ws;
  void *emb_ws2, *emb_ws3;   // round 6: the text branch of the embedding backward and the region projection's bias column sums run on
                             // the auxiliary stream beside the image branch: partial-sum workspaces of their own
  size_t ln_ws_bytes, col_ws_bytes, emb_ws_bytes, attn_ws_bytes;
  // precision 3: workspaces of the balanced walk of the 128 x 256-tile products (gemm_split3.hip): one for the main stream's
  // launches, one for the weight-gradient stream's (they run at the same time); their flag words are cleared by every forward pass
  void *sk_main, *sk_side;
  size_t sk_bytes;
  size_t total;
};

}  // namespace

struct uniter_model {
  uniter_config_t cfg;
  int n_params;
  std::vector<float*> p, g;
  std::vector<std::string> names;
  std::vector<hipEvent_t> ev_main, ev_side, ev_mid;
  int precision = 0;        // 0 = fp32 MFMA GEMMs, 1 = bf16 MFMA GEMMs (fp32 storage)
  bool mirror_paired = false;               // precision 3: the encoder layers' weights sit in the mirror in the paired-row layout
  const unsigned short* mirror = nullptr;   // bf16 copy of the flat parameter buffer (precision 2)
  const float* mirror_base = nullptr;
  size_t mirror_numel = 0;
  const unsigned short* WB(int l, int k) const { return mirror + (LP(l, k) - mirror_base); }
  std::vector<hipEvent_t> ready;   // [embeddings, layer 0 .. nl-1]: parameters usable once the event has fired (consumed by the next forward)
  // profiling
  int prof_kind = 0;         // UNITER_K_* to time, or -1 = every kind
  std::vector<hipEvent_t> prof_ev;
  std::vector<int> prof_tag;     // kind of each recorded event pair
  size_t prof_used = 0;
  // in-kernel launch stamps of the GEMMs (uniter_prof_enable_stamps)
  unsigned long long* stamp_buf = nullptr;   // device: [cap][2][STAMP_WGS] = per-workgroup start / end, 100 MHz ticks
  size_t stamp_cap = 0, stamp_used = 0;
  std::vector<int> stamp_tag;
  // state carried from forward to backward
  Plan plan;
  uniter_batch_t batch;
  float* hidden_out = nullptr;
  const float* d_hidden = nullptr;
  int all_layers = 0;
  uint64_t seed = 0;
  uint32_t offset = 0;
  hipStream_t st = nullptr, side = nullptr;
  bool bwd_open = false;
  int cu_reserve = 0;             // uniter_model_set_cu_reserve: CUs the persistent launches of this model's calls leave free
  hipStream_t aux = nullptr;      // uniter_model_set_aux_stream: launches that depend on nothing the step computes (dropout keep flags)
  hipEvent_t ev_aux0 = nullptr, ev_aux1 = nullptr, ev_aux2 = nullptr;
  hipEvent_t ev_emb0 = nullptr, ev_emb1 = nullptr, ev_emb2 = nullptr;      // embedding backward: main <-> auxiliary stream
  double* norm_parts = nullptr;   // uniter_model_set_norm_partials: layer l's clip-norm partial sums at norm_parts + l * norm_stride
  size_t norm_stride = 0;
  bool wg_overwrite = false; // the next backward pass overwrites the encoder's weight gradients (uniter_model_set_wgrad_overwrite)
  uint64_t generation = 0;   // bumped by every forward: a backward must belong to the LATEST forward (one plan / workspace per model)

  float* P(int i) const { return p[i]; }
  float* G(int i) const { return g[i]; }
  float* LP(int l, int k) const { return p[P_LAYER0 + l * L_COUNT + k]; }
  float* LG(int l, int k) const { return g[P_LAYER0 + l * L_COUNT + k]; }
  int pooler_w() const { return P_LAYER0 + cfg.num_hidden_layers * L_COUNT; }
};

namespace {

int check_cfg(const uniter_config_t* c) {
  UCHECK_ARG(c, "config is NULL");
  UCHECK_SHAPE(c->hidden_size > 0 && c->num_attention_heads > 0 &&
               c->hidden_size == 64 * c->num_attention_heads,
               "hidden_size (%d) must be 64 * num_attention_heads (%d): head_dim 64 only", c->hidden_size,
               c->num_attention_heads);
  UCHECK_SHAPE(c->hidden_size <= 1024, "hidden_size %d > 1024 unsupported", c->hidden_size);
  UCHECK_SHAPE(c->intermediate_size % 4 == 0 && c->img_dim % 4 == 0, "intermediate_size / img_dim must be multiples of 4");
  UCHECK_SHAPE(c->num_hidden_layers >= 1 && c->vocab_size > 0 && c->max_position_embeddings > 0 &&
               c->type_vocab_size >= 2, "bad config");
  return 0;
}

int x3_backward_cus(const uniter_model* m);
// precision 2, round 6: which products run on the persistent kernels of gemm_bf16_p.hip (bit mask, see gemm_v2 / backward_layer)
int b16_persist_mask() {
  // default 0: built, parity-tested through the C ABI, faster isolated -- and the STEP is slower with every subset of them (same-box A/B,
  // profiles/r06_b16_persist_ab.txt: 4.39 ms with none, 4.43-4.75 with bits 1 / 2 / 4 / 7 / 12 / 15): their 96-144 KB workgroups own a
  // CU, where two 64-KB workgroups of gemm_dma_kernel let the two backward streams and the kernel boundaries of the forward pass overlap
  static const int m = [] { const char* e = getenv("UNITER_B16_PERSIST"); return e ? atoi(e) : 0; }();
  return m;
}

void make_plan(const uniter_model* m, Plan& pl, void* ws, int B, int T, int R, int L, bool has_txt,
               bool has_img, bool has_masks, int mode, int Mrows = -1) {
  const uniter_config_t& c = m->cfg;
  const int H = c.hidden_size, I = c.intermediate_size, nl = c.num_hidden_layers, nh = c.num_attention_heads;
  pl.B = B; pl.T = T; pl.R = R; pl.L = L; pl.mode = mode; pl.has_txt = has_txt; pl.has_img = has_img;
  pl.T0 = has_txt ? T : 0;
  pl.S = pl.T0 + (has_img ? R : 0);
  pl.packed = Mrows >= 0;
  pl.M = pl.packed ? Mrows : B * L;
  const size_t M = (size_t)pl.M;
  Carver cv(ws);
  const size_t BR = (size_t)B * (has_img ? R : 0);
  pl.feat_eff = has_masks ? cv.f(BR * c.img_dim) : nullptr;
  pl.imgfc = cv.f(BR * H);
  pl.img_stats = cv.f(BR * 6);
  pl.cat = cv.f((size_t)B * pl.S * H);
  pl.emb = cv.f(M * H);
  pl.res = m->precision == 2;
  pl.x3 = m->precision == 3;
  pl.hid_keepb = nullptr; pl.hid_stride = 0; pl.hid_on = false;
  const size_t pc = pl.x3 ? 3 : 1;       // bf16 elements per value in the operand copies
  pl.embb = (pl.res || pl.x3) ? cv.h(pc * M * H) : nullptr;
  pl.ns_ki = pl.res ? gemm_bf16v2_pick_split(pl.M, H, I) : pl.x3 ? gemm_x3_pick_split(pl.M, H, I) : 1;
  pl.ns_k3h = pl.res ? gemm_bf16v2_pick_split(pl.M, H, 3 * H) : pl.x3 ? gemm_x3_pick_split(pl.M, H, 3 * H) : 1;
  if (pl.res && (b16_persist_mask() & 2)) {
    // round 6: the N = hidden products on the persistent kernels are planned by their own cost model (one workgroup per CU: two
    // k-pieces fill the chip where gemm_dma_kernel's two workgroups per CU wanted four -- two slabs less for the row pass behind)
    const int a = gemm_b1p_pick_split(pl.M, H, I, 0), b = gemm_b1p_pick_split(pl.M, H, 3 * H, 0);
    if (a > 1) pl.ns_ki = a;
    if (b > 1) pl.ns_k3h = b;
  }
  pl.ns_kh = pl.x3 ? gemm_x3_pick_split(pl.M, H, H) : 1;
  pl.ns_ki_b = pl.ns_ki; pl.ns_k3h_b = pl.ns_k3h; pl.ns_kh_b = pl.ns_kh;
  if (pl.x3) {
    // (the forward pass runs before any collective of its step: it keeps every CU; the backward pass's products are planned for the
    // CUs it will have)
    const int avail_b = x3_backward_cus(m);
    pl.ns_ki = gemm_x3_pick_split_on(pl.M, H, I, 0); pl.ns_kh = gemm_x3_pick_split_on(pl.M, H, H, 0);
    pl.ns_ki_b = gemm_x3_pick_split_on(pl.M, H, I, avail_b);
    pl.ns_kh_b = gemm_x3_pick_split_on(pl.M, H, H, avail_b);
    pl.ns_k3h_b = pl.ns_k3h = gemm_x3_pick_split_on(pl.M, H, 3 * H, avail_b);
  }
  const size_t s_ki = (size_t)pl.ns_ki, s_k3h = (size_t)pl.ns_k3h, s_kh = (size_t)pl.ns_kh;
  const size_t s_ki_b = (size_t)pl.ns_ki_b, s_kh_b = (size_t)pl.ns_kh_b;
  pl.layers.resize(nl);
  const bool save = mode != 0;
  auto alloc_fwd = [&](LayerBufs& b) {
    b.qkv = cv.f(M * 3 * H); b.lse = cv.f((size_t)B * nh * L); b.ctx = cv.f(M * H); b.t1 = cv.f(s_kh * M * H);
    b.z1 = save ? cv.f(M * H) : nullptr; b.mean1 = cv.f(M); b.rstd1 = cv.f(M); b.y1 = cv.f(M * H);
    b.u = save ? cv.f(M * I) : nullptr; b.hact = cv.f(M * I); b.t2 = cv.f(s_ki * M * H);
    b.z2 = save ? cv.f(M * H) : nullptr; b.mean2 = cv.f(M); b.rstd2 = cv.f(M); b.y2 = cv.f(M * H);
    if (pl.res || pl.x3) { b.ctxb = cv.h(pc * M * H); b.y1b = cv.h(pc * M * H); b.hactb = cv.h(pc * M * I); b.y2b = cv.h(pc * M * H); }
  };
  if (save) {
    for (int l = 0; l < nl; ++l) {
      LayerBufs& b = pl.layers[l];
      b = LayerBufs();
      alloc_fwd(b);
      b.dz2 = cv.f(M * H); b.g2 = cv.f(M * H); b.du = cv.f(M * I); b.dy1 = cv.f(s_ki_b * M * H); b.dz1 = cv.f(M * H);
      b.g1 = cv.f(M * H); b.dctx = cv.f(s_kh_b * M * H); b.dqkv = cv.f(M * 3 * H); b.delta = cv.f((size_t)B * nh * L);
      b.dx = cv.f(s_k3h * M * H);
      b.du_csum = cv.f((size_t)((M + 31) / 32) * I);
      b.qb_part = cv.f((size_t)B * 3 * H);
      b.keepb = (unsigned short*)cv.raw(uniter_attn_keep_bits_bytes(B, L, nh));
      b.ln_ws1 = cv.raw(uniter_ln_bwd_ws_bytes(pl.M, H));
      b.ln_ws2 = cv.raw(uniter_ln_bwd_ws_bytes(pl.M, H));
      if (pl.res || pl.x3) { b.g2b = cv.h(pc * M * H); b.dub = cv.h(pc * M * I); b.g1b = cv.h(pc * M * H); b.dqkvb = cv.h(pc * M * 3 * H); }
    }
  } else {
    // inference: every layer reuses one set of buffers; the layer output ping-pongs
    LayerBufs base = LayerBufs();
    alloc_fwd(base);
    float* yB = cv.f(M * H);
    for (int l = 0; l < nl; ++l) {
      pl.layers[l] = base;
      if (l & 1) pl.layers[l].y2 = yB;
    }
  }
  // UNITER_X3_BALANCED (default 0: off; bit 0 = the grouped weight gradients, bit 1 = the forward products with a bias epilogue): the
  // balanced walk is built and parity-tested, and LOSES in the step -- 9.40 -> 9.81 ms (bit 0), 9.40 -> 9.52 (bit 1) at configs[1],
  // +0.8 % only for UNITER-large's weight gradients (384 tiles: two rounds): the CUs a 216-tile launch leaves idle are not idle in the
  // step (the other stream's input-gradient kernels take them), the hand-over costs ~4 k-tiles per workgroup, and a workgroup
  // that waits for a partial sum holds its CU (profiles/r05_balanced_walk_ab.txt)
  static const int sk_env = [] { const char* e = getenv("UNITER_X3_BALANCED"); return e ? atoi(e) : 0; }();
  pl.sk_bytes = gemm_x3_sk_ws_bytes();
  pl.sk_main = (pl.x3 && (sk_env & 2)) ? cv.raw(pl.sk_bytes) : nullptr;
  pl.sk_side = (pl.x3 && save && (sk_env & 1)) ? cv.raw(pl.sk_bytes) : nullptr;
  if (save) {
    pl.hid_stride = align_up(uniter_hidden_keep_bits_bytes(M * H), 256);
    pl.hid_keepb = (unsigned char*)cv.raw(pl.hid_stride * 2 * nl);
    pl.dcat = cv.f((size_t)B * pl.S * H);
    pl.d_imgfc = cv.f(BR * H);
    pl.d_posfc = cv.f(BR * H);
    pl.d_feat = has_masks ? cv.f(BR * c.img_dim) : nullptr;
    pl.dsum = cv.f(M * H);
    pl.dpk = cv.f(M * H);          // packed mode: d_hidden gathered to the valid rows
    pl.ln_ws_bytes = uniter_ln_bwd_ws_bytes(pl.M, H);
    pl.ln_ws = cv.raw(pl.ln_ws_bytes);
    const int maxN = 3 * H > I ? 3 * H : I;
    pl.col_ws_bytes = uniter_colsum_ws_bytes(pl.M, maxN);
    pl.col_ws = cv.raw(pl.col_ws_bytes);
    const int rows = B * (T > R ? T : R);
    pl.emb_ws_bytes = uniter_embed_bwd_ws_bytes(rows, H);
    pl.emb_ws = cv.raw(pl.emb_ws_bytes);
    pl.emb_ws2 = cv.raw(pl.emb_ws_bytes);
    pl.emb_ws3 = cv.raw(pl.emb_ws_bytes);
    pl.attn_ws_bytes = uniter_attn_bwd_ws_bytes(B, L, c.num_attention_heads);
    pl.attn_ws = cv.raw(pl.attn_ws_bytes);
    pl.wg_slabs = pl.res ? cv.f((size_t)4 * 3 * H * (I > 3 * H ? I : 3 * H)) : nullptr;
    // 1 (default): the four weight gradients of a layer as one launch of whole-K tiles (gemm_bf16_dma.hip); 0: four
    // stream-K launches with float atomics; 2: as 1 with the three early products launched next to the attention
    // backward (measured slower); 4: as 1 with three LDS stages (measured slower)
    static const int wg_env = [] { const char* e = getenv("UNITER_WGRAD_GROUP"); return e ? atoi(e) : 1; }();
    pl.wg_group = (pl.res && H % 8 == 0 && I % 8 == 0 && (wg_env == 1 || wg_env == 2 || wg_env == 4)) ? wg_env : 0;
    // the grouped kernel addresses its operands with 31-bit buffer offsets: very long batches keep the stream-K launches
    if ((size_t)(M + 64) * (I > 3 * H ? I : 3 * H) * 2 >= (1ull << 31)) pl.wg_group = 0;
  }
  pl.total = cv.off;
}

struct ProfScope {
  uniter_model* m; hipStream_t st; bool on;
  ProfScope(uniter_model* m_, int kind, hipStream_t st_) : m(m_), st(st_), on(false) {
    if (m->stamp_buf && kind && kind != UNITER_K_ATTN_FWD && kind != UNITER_K_ATTN_BWD && kind != UNITER_K_LN &&
        kind != UNITER_K_LN_BWD && m->stamp_used < m->stamp_cap) {
      m->stamp_tag[m->stamp_used] = kind;                     // the GEMM launched inside this scope takes the slot
      g_uniter_stamp_slot = m->stamp_buf + (size_t)2 * STAMP_WGS * m->stamp_used++;
    }
    if (m->prof_kind && kind && (m->prof_kind == kind || m->prof_kind < 0) && m->prof_used + 2 <= m->prof_ev.size()) {
      on = true;
      m->prof_tag[m->prof_used / 2] = kind;
      hipEventRecord(m->prof_ev[m->prof_used], st);
    }
  }
  ~ProfScope() {
    if (on) { hipEventRecord(m->prof_ev[m->prof_used + 1], st); m->prof_used += 2; }
  }
};

int gemm(uniter_model* m, int kind, hipStream_t st, int akm, int bkm, int M, int N, int K, const float* A,
         int lda, const float* B, int ldb, float* C, int ldc, int epi, const float* bias, const float* aux_in,
         float* aux_out, int ld_aux, int beta, float* colsum_part = nullptr) {
  ProfScope ps(m, kind, st);
  // weight gradients: UNITER_WGRAD_CFG = 21 selects 128 x 128 stream-K tiles (measured slower: DESIGN.md section 9), 24 the 64 x 64 ones
  static const int wg_cfg = [] { const char* e = getenv("UNITER_WGRAD_CFG"); return e ? atoi(e) : 0; }();
  // UNITER_WGRAD_WHOLE: bit mask over a layer's weight gradients by output shape (1: [H, I] output.dense, 2: [I, H]
  // intermediate.dense, 4: [H, H] attention output, 8: [3H, H] query|key|value): whole-K tiles (cfg 25) instead of stream-K.
  // Default 15: measured 13.95 against 14.06 ms per step with every shape on whole tiles -- the k-synchronous tiles of a
  // tile row share their operand panel in L2 (the LayerNorm backward beside them runs at 17 instead of 26 us) -- and
  // without float atomics the fp32 weight gradients are bit-reproducible and a first backward pass may overwrite them.
  static const int wg_whole = [] { const char* e = getenv("UNITER_WGRAD_WHOLE"); return e ? atoi(e) : 15; }();
  int cfg = 0;
  const bool wgrad = kind == UNITER_K_GEMM_WGRAD && akm && bkm && (beta == 1 || beta == -1);
  if (wgrad && (m->precision == 0 || m->precision == 3)) {
    cfg = wg_cfg;
    const int H_ = m->cfg.hidden_size, I_ = m->cfg.intermediate_size;
    const int bit = (M == H_ && N == I_) ? 1 : (M == I_ && N == H_) ? 2 : (M == H_ && N == H_) ? 4 : (M == 3 * H_ && N == H_) ? 8 : 0;
    if (wg_whole & bit) cfg = 25;
  }
  if (beta == -1) {
    // overwrite requested (uniter_model_set_wgrad_overwrite): whole-K tiles store their result; every form that adds
    // partial tiles with float atomics gets a cleared output first
    if (cfg == 25) beta = 0;
    else { UCHECK_HIP(hipMemsetAsync(C, 0, (size_t)M * ldc * sizeof(float), st)); beta = 1; }
  }
  // wave priority of the critical path's GEMMs over the side stream's weight gradients (common.h).  fp32: measured WORSE
  // (14.09 against 13.73 ms per step with level 2: the input-gradient GEMM then starves the weight-gradient launch it
  // shares the matrix pipe with, and the chain waits for that launch at the layer's end), so the default is 0 here; the
  // bf16-resident GEMMs below gain 0.6 % from level 2 (their bound is operand staging, not the pipe)
  static const int main_prio = [] { const char* e = getenv("UNITER_MAIN_PRIO"); return e ? atoi(e) : 0; }();
  g_uniter_launch_prio = kind == UNITER_K_GEMM_WGRAD ? 0 : main_prio;
  if (m->precision == 1 || m->precision == 2)      // embeddings' projections (fp32 inputs) also run on the bf16 pipe in mode 2
    return gemm_bf16_run(0, akm, bkm, M, N, K, A, lda, B, ldb, C, ldc, epi, bias, aux_in, aux_out, ld_aux, beta,
                         colsum_part, st);
  return gemm_f32_run(cfg, kind == UNITER_K_GEMM_FFN_UP_FWD, akm, bkm, M, N, K, A, lda, B, ldb, C, ldc, epi, bias,
                      aux_in, aux_out, ld_aux, beta, colsum_part, st);
}

// precision 2: both operands already bf16 in memory.  Forward and input-gradient products run on the LDS-DMA kernel
// (gemm_bf16_dma.hip); aux operands (gelu' in, gelu' out) are bf16 there, residual gradients fp32.
int gemm_v2(uniter_model* m, int kind, hipStream_t st, int bkm, int M, int N, int K, const void* A, int lda,
            const void* B, int ldb, float* C, int ldc, int nsplit, unsigned short* Cb, int ldcb, int epi,
            const float* bias, const void* aux_in, int aux_in_b16, void* aux_out, int aux_out_b16, int ld_aux,
            float* colpart = nullptr) {
  ProfScope ps(m, kind, st);
  static const int main_prio = [] { const char* e = getenv("UNITER_MAIN_PRIO_BF16"); return e ? atoi(e) : 2; }();
  g_uniter_launch_prio = main_prio;
  // round 6: the persistent loader / compute kernels (csrc/gemm_bf16_p.hip) where they win isolated (profiles/r06_b1p_lab.txt):
  // the query|key|value projection on 128 x 192 tiles (252 for 256 CUs: 15.4 against 16.9 us), the N = hidden products with k-pieces
  // on persistent 128 x 128 tiles (FFN-down forward 20.4 against 22.6 us, FFN-up / QKV input gradients 22.1 / 19.1 against
  // 24.1 / 20.5).  The N = intermediate products tie (128 x 256 tiles stage a quarter less per k-tile -- 0.75 against 0.93 us, the
  // vendor library's slope -- and pay 2 us more fixed cost on a 12-k-tile loop): they keep gemm_dma_kernel.  UNITER_B16_PERSIST is a
  // bit mask: 1 = QKV forward, 2 = N = hidden products, 8 = the wide products as well (4: the grouped weight gradients, below)
  int cfg = 0;
  const int mask = b16_persist_mask();
  const bool epi_ok = epi == UNITER_EPI_NONE || epi == UNITER_EPI_BIAS || epi == UNITER_EPI_ADD || epi == UNITER_EPI_BIAS_GELU_D || epi == UNITER_EPI_MUL;
  if (epi_ok && K % 64 == 0) {
    if ((mask & 1) && kind == UNITER_K_GEMM_QKV_FWD && !bkm && N % 192 == 0) cfg = 8;
    else if ((mask & 2) && nsplit > 1 && N <= 1024) cfg = 6;
    else if ((mask & 8) && nsplit == 1 && N >= 2048) cfg = 7;
  }
  if (colpart) {      // (the product that writes dU also leaves its column partials: the persistent 128 x 256 kernel's epilogue)
    UCHECK_ARG(epi_ok && K % 64 == 0 && nsplit == 1, "gemm_v2: column partials ride on the persistent kernels");
    return gemm_b1p_run(7, 1, bkm, M, N, K, A, lda, B, ldb, C, ldc, (long)M * ldc, Cb, ldcb, epi, bias, aux_in, aux_in_b16, aux_out,
                        aux_out_b16, ld_aux, colpart, st);
  }
  return gemm_bf16v2_run(cfg, nsplit, 0, bkm, M, N, K, A, lda, B, ldb, C, ldc, (long)M * ldc, Cb, ldcb, epi, bias, aux_in,
                         aux_in_b16, aux_out, aux_out_b16, ld_aux, 0, st);
}
// precision 3: the attention's products on the bf16 pipe too (csrc/attention_x3.hip) -- L <= 192 and, with dropout, keep flags drawn
// ahead of the kernel.  UNITER_ATTN_X3=0 keeps the fp32-MFMA kernels of attention_f32.hip (A/B measurements).
bool attn_x3_products(int L, float p_drop, bool keep_flags_ready) {
  static const bool on = [] { const char* e = getenv("UNITER_ATTN_X3"); return !(e && e[0] == '0'); }();
  return on && L <= uniter_attn_x3_max_len() && (p_drop == 0.f || keep_flags_ready);
}
// precision 2: the bf16 attention in attention_x3.hip's decomposition (no scratch hand-over).  UNITER_ATTN_B16X=0 keeps
// attention_bf16.hip's kernels (A/B measurements).
bool attn_b16x(int L, float p_drop, bool keep_flags_ready) {
  static const bool on = [] { const char* e = getenv("UNITER_ATTN_B16X"); return !(e && e[0] == '0'); }();
  return on && L <= uniter_attn_x3_max_len() && (p_drop == 0.f || keep_flags_ready);
}
// precision 3: A = x3 activations [M][3][K]; W = the piece-major x3 mirror of an encoder weight (row stride ldw, piece stride =
// the flat parameter buffer's length); outputs fp32 (nsplit slabs) or x3 [M][3][N]; aux operands fp32 (csrc/gemm_split3.hip)
int gemm_x3(uniter_model* m, int kind, hipStream_t st, int bkm, int M, int N, int K, const void* A, const unsigned short* W,
            int ldw, float* C, int nsplit, unsigned short* Cx, int epi, const float* bias, const float* aux_in, float* aux_out,
            float* colsum_part = nullptr) {
  ProfScope ps(m, kind, st);
  static const int cfg_all = [] { const char* e = getenv("UNITER_X3_CFG"); return e ? atoi(e) : 0; }();
  // (lab switch: the FFN-up forward product alone on another tile geometry -- 3 = 128 x 128, whose workgroups leave register room
  // for a resident optimizer wave, which the 128 x 256 geometry's 504 of 512 registers per SIMD do not)
  static const int cfg_ffn_up = [] { const char* e = getenv("UNITER_X3_CFG_FFN_UP_FWD"); return e ? atoi(e) : 0; }();
  const int cfg = (kind == UNITER_K_GEMM_FFN_UP_FWD && cfg_ffn_up) ? cfg_ffn_up : cfg_all;
  static const int main_prio = [] { const char* e = getenv("UNITER_MAIN_PRIO_X3"); return e ? atoi(e) : 0; }();
  g_uniter_launch_prio = main_prio;
  // (every product of this helper runs on the main stream: the balanced walk's main-stream workspace)
  // (cfg | 64: the weight's pieces in the paired-row layout -- full 128-byte source lines for the loaders of the forward products)
  return gemm_x3_run(cfg | (m->mirror_paired ? 64 : 0), nsplit, 0, bkm, M, N, K, A, 3 * K, K, W, ldw, (int)m->mirror_numel, C, N, (long)M * N, Cx, 3 * N, N, epi,
                     bias, aux_in, aux_out, N, st, colsum_part, m->plan.sk_main, m->plan.sk_main ? m->plan.sk_bytes : 0);
}
// CUs the backward pass's persistent launches may use: the chip's minus what the caller reserved for a gradient exchange
int x3_backward_cus(const uniter_model* m) {
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 8 ? n / 8 * 8 : 256;
  }();
  const int a = (cus - m->cu_reserve) / 8 * 8;
  return a >= 8 ? a : 8;
}
// precision 3: side work of a layer's backward rides on its grouped weight-gradient launch (default geometry only)
bool x3_riders_enabled() {
  static const bool on = [] {
    const char* e = getenv("UNITER_X3_RIDERS");
    const char* c = getenv("UNITER_X3_CFG");
    return !(e && e[0] == '0') && (!c || atoi(c) == 0 || atoi(c) == 3);
  }();
  return on;
}
// precision 3: dU's column sums (the bias gradient of intermediate.dense) as column partials of the product that writes dU + a
// reduction job of the riders, instead of ones-MFMAs in the weight-gradient launch: whenever that launch runs on 128 x 256 tiles
bool x3_colpart_on() {
  static const bool cfg_ok = [] { const char* c = getenv("UNITER_X3_CFG"); return !c || atoi(c) == 0 || atoi(c) >= 3; }();
  return x3_riders_enabled() && cfg_ok && gemm_x3_wgrad_default_cfg() == 4;
}
// precision 2: the same riders on the grouped bf16 weight-gradient launch -- built, parity-tested, and OFF by default
// (UNITER_B16_RIDERS=1 switches them on): measured 4.42 -> 4.51 ms per step, two same-box pairs.  In this mode the two backward
// streams really overlap (two 64-KB workgroups share a CU) and the weight-gradient stream idles half of every layer: the separate
// column-sum / finalize / sum-of-squares launches cost the step nothing there, while the riders lengthen the launch the
// input-gradient chain shares its CUs with (the fp32x3 mode is the opposite case: its streams serialise, every launch is step time)
bool b16_riders_enabled() {
  static const bool on = [] { const char* e = getenv("UNITER_B16_RIDERS"); return e && e[0] == '1'; }();
  return on;
}
// round 6: with the grouped launch on the persistent 128 x 256-tile kernel (UNITER_B16_PERSIST bit 4) a workgroup of it owns its CU, the
// two backward streams time-slice as in the fp32x3 mode, and every side launch is step time: there the riders are ON unless
// UNITER_B16_RIDERS=0
bool b16_p7() { return (b16_persist_mask() & 4) != 0; }
bool b16_riders_on(const Plan& pl) {
  static const bool off = [] { const char* e = getenv("UNITER_B16_RIDERS"); return e && e[0] == '0'; }();
  // (on the 128 x 256 geometry the bias gradient of intermediate.dense must come from column partials -- bit 8 and the gelu' form of the
  // forward epilogue: a separate pass over dU would write it outside the launch, and its squares would reach no clip-norm slot)
  if (b16_p7()) return pl.res && pl.wg_group == 1 && !off && (b16_persist_mask() & 8) && pl.gelu_d;
  return pl.res && pl.wg_group == 1 && b16_riders_enabled();
}
// ... and dU's column sums (the bias gradient of intermediate.dense) start in the epilogue of the product that writes dU when that product
// runs on the persistent 128 x 256 kernel (bit 8), finished by a fourth reduction job of the riders
bool b16_colpart_on(const Plan& pl) { return b16_p7() && b16_riders_on(pl); }
// the side work of layer l's backward as riders of its grouped weight-gradient launch (product 0 must be dW1 = dU^T y1: its A
// operand's column sums are intermediate.dense's bias gradient)
void fill_layer_riders(uniter_x3_riders_t& x, const uniter_model* m, int l, const LayerBufs& lb, int M, int B, int H, bool fused_qb,
                       bool colpart = false) {
  x.colsum_out = colpart ? nullptr : m->LG(l, L_B1);
  const int lnp = ln_bwd_partial_rows(M);
  x.njobs = fused_qb ? 3 : 2;
  x.part[0] = (const float*)lb.ln_ws2; x.part[1] = (const float*)lb.ln_ws1; x.part[2] = lb.qb_part;
  x.nparts[0] = x.nparts[1] = lnp; x.nparts[2] = B;
  x.stride[0] = x.stride[1] = x.stride[2] = 3 * H;
  x.n[0] = x.n[1] = x.n[2] = 3 * H;
  x.seg[0] = x.seg[1] = H; x.seg[2] = 3 * H;
  x.out[0][0] = m->LG(l, L_LN2_G); x.out[0][1] = m->LG(l, L_LN2_B); x.out[0][2] = m->LG(l, L_B2);
  x.out[1][0] = m->LG(l, L_LN1_G); x.out[1][1] = m->LG(l, L_LN1_B); x.out[1][2] = m->LG(l, L_OB);
  x.out[2][0] = m->LG(l, L_QB);
  if (!fused_qb) {      // (no per-sample query|key|value partials: the column-partial job, if any, takes slot 2)
    x.part[2] = nullptr; x.out[2][0] = nullptr;
  }
  if (colpart) {
    const int j = x.njobs++, I = m->cfg.intermediate_size;
    x.part[j] = lb.du_csum; x.nparts[j] = (M + 63) / 64; x.stride[j] = I; x.n[j] = I; x.seg[j] = I;
    x.out[j][0] = m->LG(l, L_B1); x.out[j][1] = x.out[j][2] = nullptr;
  }
}
int split_x3(const float* src, unsigned short* dst, int rows, int cols, hipStream_t st) {
  return uniter_split3(src, rows, cols, cols, dst, (size_t)3 * cols, (size_t)cols, st);
}
// weight gradients (both operands k-major, stream-K, fp32 atomics into the gradient buffer): gemm_bf16.hip
int gemm_r(uniter_model* m, int kind, hipStream_t st, int akm, int bkm, int M, int N, int K, const void* A, int lda,
           const void* B, int ldb, float* C, int ldc, unsigned short* Cb, int ldcb, int epi, const float* bias,
           const float* aux_in, float* aux_out, int ld_aux, int beta, float* colsum_part = nullptr) {
  ProfScope ps(m, kind, st);
  return gemm_bf16res_run(0, akm, bkm, M, N, K, A, lda, B, ldb, C, ldc, Cb, ldcb, epi, bias, aux_in, aux_out, ld_aux,
                          beta, colsum_part, st);
}
// dW[Mo, No] += A^T B with A [K, Mo], B [K, No] bf16: split-K slabs + one reduce pass where that beats stream-K atomics
int wgrad_b16(uniter_model* m, const Plan& pl, hipStream_t st, int Mo, int No, int K, const void* A, const void* B, float* dW) {
  if (m->wg_overwrite) UCHECK_HIP(hipMemsetAsync(dW, 0, (size_t)Mo * No * sizeof(float), st));      // these forms accumulate
  const int pieces = gemm_bf16v2_wgrad_pieces(Mo, No, K);
  if (pieces == 0 || !pl.wg_slabs)
    return gemm_r(m, UNITER_K_GEMM_WGRAD, st, 1, 1, Mo, No, K, A, Mo, B, No, dW, No, nullptr, 0, UNITER_EPI_NONE, nullptr,
                  nullptr, nullptr, 0, 1);
  {
    ProfScope ps(m, UNITER_K_GEMM_WGRAD, st);
    UCHECK_RC(gemm_bf16v2_run(0, pieces, 1, 1, Mo, No, K, A, Mo, B, No, pl.wg_slabs, No, (long)Mo * No, nullptr, 0,
                              UNITER_EPI_NONE, nullptr, nullptr, 0, nullptr, 0, 0, 0, st));
  }
  return uniter_slab_reduce_add(pl.wg_slabs, pieces, (size_t)Mo * No, dW, (size_t)Mo * No, st);
}
int cast_b(const float* src, unsigned short* dst, size_t n, hipStream_t st) { return uniter_cast_bf16(src, dst, n, st); }

int validate_batch(const uniter_model* m, const uniter_batch_t* b) {
  UCHECK_ARG(b, "batch is NULL");
  UCHECK_ARG(b->B > 0 && b->L > 0, "batch: B and L must be positive");
  UCHECK_ARG(b->input_ids || b->img_feat, "batch: need input_ids and/or img_feat");
  UCHECK_ARG(b->attention_mask, "batch: attention_mask is NULL");
  if (b->input_ids) UCHECK_ARG(b->position_ids && b->T > 0, "batch: text needs position_ids and T > 0");
  if (b->img_feat) UCHECK_ARG(b->img_pos_feat && b->R > 0, "batch: image needs img_pos_feat and R > 0");
  const int S = (b->input_ids ? b->T : 0) + (b->img_feat ? b->R : 0);
  if (!(b->input_ids && b->img_feat && b->gather_index))
    UCHECK_SHAPE(b->L == S || (b->input_ids && b->img_feat && b->L <= S),
                 "batch: without gather_index L (%d) must equal T+R (%d)", b->L, S);
  UCHECK_SHAPE(b->T <= m->cfg.max_position_embeddings || !b->input_ids, "batch: T exceeds max_position_embeddings");
  return 0;
}

}  // namespace

// ------------------------------------------------------------------ C ABI ---
extern "C" int uniter_num_params(const uniter_config_t* cfg) {
  if (!cfg) return 0;
  return P_LAYER0 + cfg->num_hidden_layers * L_COUNT + 2;
}

extern "C" const char* uniter_param_name(const uniter_config_t* cfg, int i) {
  static thread_local char buf[128];
  if (!cfg || i < 0 || i >= uniter_num_params(cfg)) return nullptr;
  if (i < P_LAYER0) return kEmbNames[i];
  const int nl = cfg->num_hidden_layers;
  if (i < P_LAYER0 + nl * L_COUNT) {
    const int l = (i - P_LAYER0) / L_COUNT, k = (i - P_LAYER0) % L_COUNT;
    snprintf(buf, sizeof(buf), "encoder.layer.%d.%s", l, kLayerNames[k]);
    return buf;
  }
  return i == P_LAYER0 + nl * L_COUNT ? "pooler.dense.weight" : "pooler.dense.bias";
}

extern "C" int uniter_param_shape(const uniter_config_t* c, int i, int64_t* rows, int64_t* cols) {
  if (!c || !rows || !cols || i < 0 || i >= uniter_num_params(c)) return UNITER_E_ARG;
  const int64_t H = c->hidden_size, I = c->intermediate_size, D = c->img_dim;
  int64_t r = H, cc = 0;   // cols == 0 -> 1-D of length rows
  if (i < P_LAYER0) {
    switch (i) {
      case P_WORD: r = c->vocab_size; cc = H; break;
      case P_POS: r = c->max_position_embeddings; cc = H; break;
      case P_TYPE: r = c->type_vocab_size; cc = H; break;
      case P_IMG_W: r = H; cc = D; break;
      case P_POSL_W: r = H; cc = 7; break;
      case P_MASK_EMB: r = 2; cc = D; break;
      default: break;
    }
  } else if (i < P_LAYER0 + c->num_hidden_layers * L_COUNT) {
    switch ((i - P_LAYER0) % L_COUNT) {
      case L_QW: case L_KW: case L_VW: case L_OW: cc = H; break;
      case L_W1: r = I; cc = H; break;
      case L_B1: r = I; break;
      case L_W2: r = H; cc = I; break;
      default: break;
    }
  } else if (i == P_LAYER0 + c->num_hidden_layers * L_COUNT) {
    cc = H;
  }
  *rows = r; *cols = cc;
  return 0;
}

extern "C" int uniter_model_create(const uniter_config_t* cfg, float* const* params, float* const* grads,
                                   int n_params, uniter_model_t** out) {
  UCHECK_RC(check_cfg(cfg));
  UCHECK_ARG(params && out, "model_create: null pointer");
  UCHECK_ARG(n_params == uniter_num_params(cfg), "model_create: expected %d parameters, got %d",
             uniter_num_params(cfg), n_params);
  uniter_model* m = new uniter_model();
  m->cfg = *cfg;
  m->n_params = n_params;
  m->p.assign(params, params + n_params);
  if (grads) m->g.assign(grads, grads + n_params); else m->g.assign(n_params, nullptr);
  const size_t H = cfg->hidden_size;
  for (int i = 0; i < n_params; ++i) {
    if (!m->p[i] || ((uintptr_t)m->p[i] & 15)) {
      uniter_set_error("model_create: parameter %d (%s) is NULL or not 16-byte aligned", i, uniter_param_name(cfg, i));
      delete m; return UNITER_E_ARG;
    }
  }
  for (int l = 0; l < cfg->num_hidden_layers; ++l) {
    const bool ok = m->LP(l, L_KW) == m->LP(l, L_QW) + H * H && m->LP(l, L_VW) == m->LP(l, L_QW) + 2 * H * H &&
                    m->LP(l, L_KB) == m->LP(l, L_QB) + H && m->LP(l, L_VB) == m->LP(l, L_QB) + 2 * H;
    const bool gok = !grads || (m->LG(l, L_KW) == m->LG(l, L_QW) + H * H && m->LG(l, L_VW) == m->LG(l, L_QW) + 2 * H * H &&
                                m->LG(l, L_KB) == m->LG(l, L_QB) + H && m->LG(l, L_VB) == m->LG(l, L_QB) + 2 * H);
    if (!ok || !gok) {
      uniter_set_error("model_create: layer %d query/key/value weights (and biases, and their grads) must be contiguous", l);
      delete m; return UNITER_E_ARG;
    }
  }
  const int nl = cfg->num_hidden_layers;
  m->ev_main.resize(nl + 1); m->ev_side.resize(nl + 1); m->ev_mid.resize(nl + 1);
  for (int i = 0; i <= nl; ++i) {
    hipError_t e1 = hipEventCreateWithFlags(&m->ev_main[i], hipEventDisableTiming);
    if (e1 == hipSuccess) e1 = hipEventCreateWithFlags(&m->ev_mid[i], hipEventDisableTiming);
    hipError_t e2 = hipEventCreateWithFlags(&m->ev_side[i], hipEventDisableTiming);
    if (e1 != hipSuccess || e2 != hipSuccess) {
      uniter_set_error("model_create: hipEventCreate failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
      delete m; return (int)(e1 != hipSuccess ? e1 : e2);
    }
  }
  *out = m;
  return 0;
}

extern "C" void uniter_model_destroy(uniter_model_t* m) {
  if (!m) return;
  for (auto e : m->ev_main) if (e) hipEventDestroy(e);
  for (auto e : m->ev_side) if (e) hipEventDestroy(e);
  for (auto e : m->ev_mid) if (e) hipEventDestroy(e);
  for (auto e : m->prof_ev) if (e) hipEventDestroy(e);
  if (m->stamp_buf) (void)hipFree(m->stamp_buf);
  if (m->ev_aux0) hipEventDestroy(m->ev_aux0);
  if (m->ev_aux1) hipEventDestroy(m->ev_aux1);
  if (m->ev_aux2) hipEventDestroy(m->ev_aux2);
  if (m->ev_emb0) hipEventDestroy(m->ev_emb0);
  if (m->ev_emb1) hipEventDestroy(m->ev_emb1);
  if (m->ev_emb2) hipEventDestroy(m->ev_emb2);
  delete m;
}

extern "C" size_t uniter_model_ws_bytes(const uniter_model_t* m, int B, int T, int R, int L, int train) {
  if (!m) return 0;
  Plan pl;
  // sizes do not depend on pointers: carve from address 0; assume both modalities + masks (upper bound)
  make_plan(m, pl, nullptr, B, T > 0 ? T : 0, R > 0 ? R : 0, L, T > 0, R > 0, true, train);
  return pl.total + 256;
}

extern "C" int uniter_model_forward(uniter_model_t* m, const uniter_batch_t* b, float* hidden_out,
                                    int all_layers, int train, uint64_t seed, uint32_t offset, void* ws,
                                    size_t ws_bytes, void* stream) {
  UCHECK_ARG(m && hidden_out && ws, "model_forward: null pointer");
  g_uniter_cu_reserve = 0;      // (no collective of this step is in flight during its forward pass: the reserve is the backward pass's)
  UCHECK_ARG(train >= 0 && train <= 2, "model_forward: train must be 0, 1 or 2");
  UCHECK_RC(validate_batch(m, b));
  UCHECK_ARG(((uintptr_t)ws & 255) == 0, "model_forward: workspace must be 256-byte aligned");
  const uniter_config_t& c = m->cfg;
  const int H = c.hidden_size, I = c.intermediate_size, nl = c.num_hidden_layers, nh = c.num_attention_heads;
  const bool has_txt = b->input_ids != nullptr, has_img = b->img_feat != nullptr;
  Plan& pl = m->plan;
  const bool packed = b->cu_seqlens != nullptr;
  if (packed) {
    UCHECK_ARG(b->pack_src && b->pack_dst, "model_forward: packed mode needs pack_src and pack_dst");
    UCHECK_SHAPE(b->Mp > 0 && (int64_t)b->Mp <= (int64_t)b->B * b->L, "model_forward: bad Mp %d", b->Mp);
    UCHECK_SHAPE(b->L <= uniter_attn_varlen_max_len(), "model_forward: packed mode supports L <= %d (got %d)",
                 uniter_attn_varlen_max_len(), b->L);
  }
  make_plan(m, pl, ws, b->B, b->T, b->R, b->L, has_txt, has_img, b->img_masks != nullptr, train,
            packed ? b->Mp : -1);
  if (pl.total > ws_bytes) {
    uniter_set_error("model_forward: workspace too small (%zu < %zu)", ws_bytes, pl.total);
    return UNITER_E_WS;
  }
  hipStream_t st = (hipStream_t)stream;
  const float ph = train == 1 ? c.hidden_dropout_prob : 0.f;
  const float pa = train == 1 ? c.attention_probs_dropout_prob : 0.f;
  const int B = b->B, T = b->T, R = b->R, L = b->L, S = pl.S, M = pl.M;
  const bool save = train != 0;
  static const bool gelu_d_env = [] { const char* e = getenv("UNITER_GELU_D"); return !(e && e[0] == '0'); }();
  const bool gelu_d = gelu_d_env || m->precision == 3;      // UNITER_GELU_D=0: A/B switch (store the pre-activation, libm erf twice)
  pl.gelu_d = gelu_d;
  m->batch = *b; m->hidden_out = hidden_out; m->all_layers = all_layers; m->seed = seed; m->offset = offset;
  m->bwd_open = false;
  ++m->generation;

  const bool res = pl.res, x3 = pl.x3;
  // L <= 192: attention on the bf16 pipe (attention_bf16.hip), which also writes the bf16 copies of its outputs
  const bool attn_b16 = res && L <= uniter_attn_varlen_max_len();

  // the flag words of the balanced walk's two workspaces (16 KB each): zero before the first launch that uses them -- the launches
  // leave them zero, but the workspace is the caller's and may be a new one
  if (pl.sk_main) UCHECK_HIP(hipMemsetAsync(pl.sk_main, 0, 16384, st));
  if (pl.sk_side) UCHECK_HIP(hipMemsetAsync(pl.sk_side, 0, 16384, st));

  // dropout keep flags of the attention probabilities of ALL layers, drawn in one elementwise launch ahead of everything --
  // a function of (seed, offset) alone, so it runs while the optimizer still streams the word table the text branch waits for
  // (the L <= 192 kernels then read them; the backward pass reads the same words)
  static const bool pregen_env = [] { const char* e = getenv("UNITER_KEEP_PREGEN"); return !e || e[0] != '0'; }();
  bool keep_on_aux = false;
  bool keep_pre = pregen_env && save && pa > 0.f && L <= uniter_attn_varlen_max_len() && (attn_b16 || packed || save);
  if (keep_pre) {
    const size_t stride = nl > 1 ? (size_t)((char*)pl.layers[1].keepb - (char*)pl.layers[0].keepb) : 0;
    for (int l = 1; l < nl && keep_pre; ++l)
      keep_pre = (size_t)((char*)pl.layers[l].keepb - (char*)pl.layers[0].keepb) == stride * l;
    if (keep_pre) {
      // round 5: on the auxiliary stream when the caller gave one (uniter_model_set_aux_stream) -- 60 us of Philox rounds that
      // depend on (seed, offset) alone, beside the head of the forward pass (region projection, embeddings: small launches that
      // leave the chip empty) instead of in front of it; layer 0's attention waits for the event
      hipStream_t ks = st;
      if (m->aux && m->aux != st) {
        if (!m->ev_aux0) UCHECK_HIP(hipEventCreateWithFlags(&m->ev_aux0, hipEventDisableTiming));
        if (!m->ev_aux1) UCHECK_HIP(hipEventCreateWithFlags(&m->ev_aux1, hipEventDisableTiming));
        UCHECK_HIP(hipEventRecord(m->ev_aux0, st));            // the flags' last readers (the previous backward pass) ran on `st`
        UCHECK_HIP(hipStreamWaitEvent(m->aux, m->ev_aux0, 0));
        ks = m->aux;
      }
      UCHECK_RC(uniter_attn_keep_bits_gen(pl.layers[0].keepb, stride, nl, B, L, nh, pa, seed, offset, SITE_ATTN_PROBS(0),
                                          SITE_ATTN_PROBS(1) - SITE_ATTN_PROBS(0), ks));
      if (ks != st) { UCHECK_HIP(hipEventRecord(m->ev_aux1, ks)); keep_on_aux = true; }
    }
  }
  // the hidden dropout's keep flags (the two dropout + residual + LayerNorm passes of every layer, forward AND backward) behind them
  // on the same stream: built, bit-identical (tests/test_layernorm_gpu.py) and OFF by default (UNITER_HIDDEN_PREGEN=1 switches it on).
  // Alone a row pass is 1.5 / 1.8 us faster without its ten Philox rounds per group; in the step it is not -- there the passes wait
  // for memory, the Philox rounds run in that shadow, and the generator launch is one more kernel beside the forward's head:
  // 9.87 / 9.87 ms without against 9.96 / 9.86 with (fp32x3), 4.58 / 4.60 against 4.60 / 4.61 (bf16), two same-box pairs each
  static const bool hid_env = [] { const char* e = getenv("UNITER_HIDDEN_PREGEN"); return e && e[0] == '1'; }();
  pl.hid_on = false;
  if (hid_env && save && ph > 0.f && m->aux && m->aux != st && pl.hid_keepb) {
    if (!m->ev_aux0) UCHECK_HIP(hipEventCreateWithFlags(&m->ev_aux0, hipEventDisableTiming));
    if (!m->ev_aux2) UCHECK_HIP(hipEventCreateWithFlags(&m->ev_aux2, hipEventDisableTiming));
    if (!keep_on_aux) {      // (no attention flags were drawn on the stream: order it behind the flags' last readers ourselves)
      UCHECK_HIP(hipEventRecord(m->ev_aux0, st));
      UCHECK_HIP(hipStreamWaitEvent(m->aux, m->ev_aux0, 0));
    }
    UCHECK_RC(uniter_hidden_keep_bits_gen(pl.hid_keepb, pl.hid_stride, 2 * nl, SITE_ATTN_OUT(0), SITE_FFN_OUT(0),
                                          SITE_ATTN_OUT(1) - SITE_ATTN_OUT(0), (size_t)M * H, ph, seed, offset, m->aux));
    UCHECK_HIP(hipEventRecord(m->ev_aux2, m->aux));
    pl.hid_on = true;
  }

  // parameters still being written by an optimizer step on another stream (uniter_model_set_ready_events)
  // ready = [embeddings, layer 0 .. nl-1] or, with one more event, [embeddings WITHOUT the word table, layers.., word table]:
  // the image branch (region projection, 3 LayerNorms) then runs while the optimizer still streams the 89-MB word table
  const bool gated = (int)m->ready.size() == nl + 1 || (int)m->ready.size() == nl + 2;
  const bool word_split = (int)m->ready.size() == nl + 2;
  if (gated) UCHECK_HIP(hipStreamWaitEvent(st, m->ready[0], 0));

  // ---- embeddings (model/model.py:321-334) ----
  if (has_img) {
    const float* feat = b->img_feat;
    if (b->img_masks) {
      UCHECK_RC(uniter_img_mask_add(b->img_feat, b->img_masks, m->P(P_MASK_EMB), pl.feat_eff, B * R, c.img_dim, st));
      feat = pl.feat_eff;
    }
    // region-feature projection (model/model.py:267): 576 x 768 outputs are 108 tiles of 64 x 64 for 1024 workgroup slots
    // with K = 2048 -- as a stream-K accumulation on top of the bias rows it fills the chip (fp32: 58 -> 25 us)
    const long t64 = (long)((B * R + 63) / 64) * ((H + 63) / 64);
    static const bool img_sk = [] { const char* e = getenv("UNITER_IMG_SK"); return !e || e[0] != '0'; }();
    // (native fp32 only: the float atomics make the sum's order vary from run to run -- 1e-7 in fp32, but the bf16 mode amplifies any
    // such difference through its rounding stages, DESIGN.md section 2, and the fp32x3 mode's forward and backward are bit-reproducible
    // run to run, which tests/test_model_gpu.py::test_cu_reserve_for_a_gradient_exchange_changes_no_result relies on -- round 6 tried
    // the stream-K form there and took it out again for that reason)
    if (img_sk && m->precision == 0 && t64 >= 8 && t64 <= 600 && c.img_dim >= 1024 && c.img_dim % 64 == 0 && H % 4 == 0) {
      UCHECK_RC(uniter_bias_rows(m->P(P_IMG_B), pl.imgfc, B * R, H, st));
      UCHECK_RC(gemm(m, 0, st, 0, 0, B * R, H, c.img_dim, feat, c.img_dim, m->P(P_IMG_W), c.img_dim, pl.imgfc, H,
                     UNITER_EPI_NONE, nullptr, nullptr, nullptr, 0, 1));
    } else {
      UCHECK_RC(gemm(m, 0, st, 0, 0, B * R, H, c.img_dim, feat, c.img_dim, m->P(P_IMG_W), c.img_dim, pl.imgfc, H,
                     UNITER_EPI_BIAS, m->P(P_IMG_B), nullptr, nullptr, 0, 0));
    }
    UCHECK_RC(uniter_img_embed_fwd(pl.imgfc, b->img_pos_feat, b->img_type_ids, m->P(P_POSL_W), m->P(P_POSL_B),
                                   m->P(P_TYPE), m->P(P_ILN_G), m->P(P_ILN_B), m->P(P_PLN_G), m->P(P_PLN_B),
                                   m->P(P_FLN_G), m->P(P_FLN_B), pl.cat, save ? pl.img_stats : nullptr, B, R,
                                   pl.T0, S, H, c.type_vocab_size, ph, seed, offset, st));
  }
  if (has_txt) {
    if (word_split) UCHECK_HIP(hipStreamWaitEvent(st, m->ready[nl + 1], 0));
    UCHECK_RC(uniter_txt_embed_fwd(b->input_ids, b->position_ids, b->txt_type_ids, m->P(P_WORD), m->P(P_POS),
                                   m->P(P_TYPE), m->P(P_ELN_G), m->P(P_ELN_B), pl.cat, B, T, S, H,
                                   c.vocab_size, c.max_position_embeddings, c.type_vocab_size, b->pos_bcast,
                                   ph, seed, offset, st));
  }
  const bool joint = has_txt && has_img;
  // (round 6) the joint rows and their operand copy for layer 0's first product in ONE launch: this chain -- text embeddings, gather,
  // split -- is what the first encoder product waits for behind the optimizer's update of the word table
  static const bool gather_ex = [] { const char* e = getenv("UNITER_GATHER_EX"); return !(e && e[0] == '0'); }();
  const bool fuse_copy = gather_ex && !packed && (res || x3) && H <= 1024 && m->mirror != nullptr;
  if (packed) UCHECK_RC(uniter_row_gather(pl.cat, b->pack_src, pl.emb, M, H, B * S, st));
  else if (fuse_copy) UCHECK_RC(uniter_gather_rows_ex(pl.cat, joint ? b->gather_index : nullptr, pl.emb, pl.embb, x3 ? 1 : 2, B, S, L, H, st));
  else UCHECK_RC(uniter_gather_rows(pl.cat, joint ? b->gather_index : nullptr, pl.emb, B, S, L, H, st));
  const size_t PH = (size_t)B * L * H;          // one layer of the padded output
  if ((res || x3) && !fuse_copy) {
    UCHECK_ARG(m->mirror != nullptr, "model_forward: precision 2 / 3 needs uniter_model_set_weight_mirror");
    if (res) UCHECK_RC(cast_b(pl.emb, pl.embb, (size_t)M * H, st));
    else UCHECK_RC(split_x3(pl.emb, pl.embb, M, H, st));
  }
  const unsigned short* xb = pl.embb;
  // ---- encoder (model/model.py:282-292; model/layer.py:166-170) ----
  const float* x = pl.emb;
  for (int l = 0; l < nl; ++l) {
    LayerBufs& lb = pl.layers[l];
    if (gated) UCHECK_HIP(hipStreamWaitEvent(st, m->ready[1 + l], 0));
    if (l == 0 && keep_on_aux) UCHECK_HIP(hipStreamWaitEvent(st, m->ev_aux1, 0));      // the keep flags of every layer are drawn
    float* y2 = packed ? lb.y2 : (all_layers ? hidden_out + l * PH : (l == nl - 1 ? hidden_out : lb.y2));
    if (attn_b16)      // the bf16 attention kernels read Q, K, V as bf16: write only that (in the qkv buffer's place)
      UCHECK_RC(gemm_v2(m, UNITER_K_GEMM_QKV_FWD, st, 0, M, 3 * H, H, xb, H, m->WB(l, L_QW), H, nullptr, 3 * H, 1,
                        (unsigned short*)lb.qkv, 3 * H, UNITER_EPI_BIAS, m->LP(l, L_QB), nullptr, 0, nullptr, 0, 0));
    else if (res)
      UCHECK_RC(gemm_v2(m, UNITER_K_GEMM_QKV_FWD, st, 0, M, 3 * H, H, xb, H, m->WB(l, L_QW), H, lb.qkv, 3 * H, 1,
                        nullptr, 0, UNITER_EPI_BIAS, m->LP(l, L_QB), nullptr, 0, nullptr, 0, 0));
    else if (x3)
      UCHECK_RC(gemm_x3(m, UNITER_K_GEMM_QKV_FWD, st, 0, M, 3 * H, H, xb, m->WB(l, L_QW), H, lb.qkv, 1, nullptr,
                        UNITER_EPI_BIAS, m->LP(l, L_QB), nullptr, nullptr));
    else
      UCHECK_RC(gemm(m, UNITER_K_GEMM_QKV_FWD, st, 0, 0, M, 3 * H, H, x, H, m->LP(l, L_QW), H, lb.qkv, 3 * H,
                     UNITER_EPI_BIAS, m->LP(l, L_QB), nullptr, nullptr, 0, 0));
    {
      ProfScope ps(m, UNITER_K_ATTN_FWD, st);
      if (attn_b16 && attn_b16x(L, pa, keep_pre))      // precision 2: the attention products run on the bf16 pipe as well
        UCHECK_RC(uniter_attn_b16x_fwd(lb.qkv, 1, packed ? nullptr : b->attention_mask, packed ? b->cu_seqlens : nullptr, lb.ctx,
                                       lb.ctxb, lb.lse, pa > 0.f ? lb.keepb : nullptr, B, L, nh, pa, st));
      else if (attn_b16)
        UCHECK_RC(uniter_attn_bf16_fwd_pre(lb.qkv, 1, packed ? nullptr : b->attention_mask, packed ? b->cu_seqlens : nullptr,
                                           lb.ctx, lb.ctxb, lb.lse, save ? lb.keepb : nullptr, keep_pre ? 1 : 0, B, L, nh, pa,
                                           seed, offset, SITE_ATTN_PROBS(l), st));
      else if (x3 && attn_x3_products(L, pa, keep_pre))      // the attention's own products on the bf16 pipe as well
        UCHECK_RC(uniter_attn_x3_fwd(lb.qkv, packed ? nullptr : b->attention_mask, packed ? b->cu_seqlens : nullptr, lb.ctx,
                                     lb.ctxb, lb.lse, pa > 0.f ? lb.keepb : nullptr, B, L, nh, pa, st));
      else if (x3 && L <= uniter_attn_varlen_max_len())      // the context leaves the kernel as x3 pieces too
        UCHECK_RC(uniter_attn_fwd_pre_x3(lb.qkv, packed ? nullptr : b->attention_mask, packed ? b->cu_seqlens : nullptr,
                                         lb.ctx, lb.ctxb, lb.lse, save ? lb.keepb : nullptr, keep_pre ? 1 : 0, B, L, nh, pa,
                                         seed, offset, SITE_ATTN_PROBS(l), st));
      else if (packed || (save && L <= uniter_attn_varlen_max_len()))
        UCHECK_RC(uniter_attn_fwd_pre(lb.qkv, packed ? nullptr : b->attention_mask, packed ? b->cu_seqlens : nullptr,
                                      lb.ctx, nullptr, lb.lse, save ? lb.keepb : nullptr, keep_pre ? 1 : 0, B, L, nh, pa,
                                      seed, offset, SITE_ATTN_PROBS(l), st));
      else
        UCHECK_RC(uniter_attn_fwd(lb.qkv, b->attention_mask, lb.ctx, lb.lse, B, L, nh, pa, seed, offset,
                                  SITE_ATTN_PROBS(l), st));
    }
    if (res) {
      if (!attn_b16) UCHECK_RC(cast_b(lb.ctx, lb.ctxb, (size_t)M * H, st));
      UCHECK_RC(gemm_v2(m, UNITER_K_GEMM_ATTN_OUT_FWD, st, 0, M, H, H, lb.ctxb, H, m->WB(l, L_OW), H, lb.t1, H, 1,
                        nullptr, 0, UNITER_EPI_BIAS, m->LP(l, L_OB), nullptr, 0, nullptr, 0, 0));
    } else if (x3) {
      // the context as x3 pieces: operand of the projection below and of its weight gradient (L <= 192: the attention kernel wrote them)
      if (L > uniter_attn_varlen_max_len()) UCHECK_RC(split_x3(lb.ctx, lb.ctxb, M, H, st));
      UCHECK_RC(gemm_x3(m, UNITER_K_GEMM_ATTN_OUT_FWD, st, 0, M, H, H, lb.ctxb, m->WB(l, L_OW), H, lb.t1, pl.ns_kh, nullptr,
                        UNITER_EPI_BIAS, m->LP(l, L_OB), nullptr, nullptr));
    } else {
      UCHECK_RC(gemm(m, UNITER_K_GEMM_ATTN_OUT_FWD, st, 0, 0, M, H, H, lb.ctx, H, m->LP(l, L_OW), H, lb.t1, H,
                     UNITER_EPI_BIAS, m->LP(l, L_OB), nullptr, nullptr, 0, 0));
    }
    if (l == 0 && pl.hid_on) UCHECK_HIP(hipStreamWaitEvent(st, m->ev_aux2, 0));      // every layer's hidden keep flags are drawn
    {
      ProfScope ps(m, UNITER_K_LN, st);
      g_uniter_drop_bits = pl.hid_on ? pl.hid_keepb + (size_t)(2 * l) * pl.hid_stride : nullptr;
      if (x3)
        UCHECK_RC(uniter_ln_fwd_slabs_x3(lb.t1, pl.ns_kh, (size_t)M * H, x, m->LP(l, L_LN1_G), m->LP(l, L_LN1_B), lb.z1, lb.y1,
                                         lb.y1b, save ? lb.mean1 : nullptr, save ? lb.rstd1 : nullptr, M, H, ph, seed, offset,
                                         SITE_ATTN_OUT(l), st));
      else
      UCHECK_RC(uniter_ln_fwd_b16(lb.t1, x, m->LP(l, L_LN1_G), m->LP(l, L_LN1_B), lb.z1, lb.y1, res ? lb.y1b : nullptr,
                                  save ? lb.mean1 : nullptr, save ? lb.rstd1 : nullptr, M, H, ph, seed, offset,
                                  SITE_ATTN_OUT(l), st));
    }
    if (res) {
      // the activation and gelu'(u) (or u) only exist in bf16: operand of FFN-down / of its weight gradient, factor of dU
      UCHECK_RC(gemm_v2(m, UNITER_K_GEMM_FFN_UP_FWD, st, 0, M, I, H, lb.y1b, H, m->WB(l, L_W1), H, nullptr, I, 1,
                        lb.hactb, I, gelu_d ? UNITER_EPI_BIAS_GELU_D : UNITER_EPI_BIAS_GELU, m->LP(l, L_B1), nullptr, 0,
                        save ? (void*)lb.u : (void*)lb.hact, 1, I));
      UCHECK_RC(gemm_v2(m, UNITER_K_GEMM_FFN_DOWN_FWD, st, 0, M, H, I, lb.hactb, I, m->WB(l, L_W2), I, lb.t2, H, pl.ns_ki,
                        nullptr, 0, UNITER_EPI_BIAS, m->LP(l, L_B2), nullptr, 0, nullptr, 0, 0));
    } else if (x3) {
      // the activation exists only as x3 pieces (operand of FFN-down and of its weight gradient); gelu'(u) stays fp32
      UCHECK_RC(gemm_x3(m, UNITER_K_GEMM_FFN_UP_FWD, st, 0, M, I, H, lb.y1b, m->WB(l, L_W1), H, nullptr, 1, lb.hactb,
                        UNITER_EPI_BIAS_GELU_D, m->LP(l, L_B1), nullptr, save ? lb.u : lb.hact));
      UCHECK_RC(gemm_x3(m, UNITER_K_GEMM_FFN_DOWN_FWD, st, 0, M, H, I, lb.hactb, m->WB(l, L_W2), I, lb.t2, pl.ns_ki, nullptr,
                        UNITER_EPI_BIAS, m->LP(l, L_B2), nullptr, nullptr));
    } else {
      UCHECK_RC(gemm(m, UNITER_K_GEMM_FFN_UP_FWD, st, 0, 0, M, I, H, lb.y1, H, m->LP(l, L_W1), H, lb.hact, I,
                     gelu_d ? UNITER_EPI_BIAS_GELU_D : UNITER_EPI_BIAS_GELU, m->LP(l, L_B1), nullptr, lb.u, I, 0));
      UCHECK_RC(gemm(m, UNITER_K_GEMM_FFN_DOWN_FWD, st, 0, 0, M, H, I, lb.hact, I, m->LP(l, L_W2), I, lb.t2, H,
                     UNITER_EPI_BIAS, m->LP(l, L_B2), nullptr, nullptr, 0, 0));
    }
    {
      ProfScope ps(m, UNITER_K_LN, st);
      g_uniter_drop_bits = pl.hid_on ? pl.hid_keepb + (size_t)(2 * l + 1) * pl.hid_stride : nullptr;
      if (x3)
        UCHECK_RC(uniter_ln_fwd_slabs_x3(lb.t2, pl.ns_ki, (size_t)M * H, lb.y1, m->LP(l, L_LN2_G), m->LP(l, L_LN2_B), lb.z2, y2,
                                         lb.y2b, save ? lb.mean2 : nullptr, save ? lb.rstd2 : nullptr, M, H, ph, seed, offset,
                                         SITE_FFN_OUT(l), st));
      else
      UCHECK_RC(uniter_ln_fwd_slabs(lb.t2, pl.ns_ki, (size_t)M * H, lb.y1, m->LP(l, L_LN2_G), m->LP(l, L_LN2_B), lb.z2, y2,
                                    res ? lb.y2b : nullptr, save ? lb.mean2 : nullptr, save ? lb.rstd2 : nullptr, M, H,
                                    ph, seed, offset, SITE_FFN_OUT(l), st));
    }
    lb.y2 = y2;
    x = y2;
    if (res || x3) xb = lb.y2b;
    if (packed && (all_layers || l == nl - 1)) {
      // padded [B, L, H] view for the caller: valid rows scattered, padded positions zero
      float* dst = all_layers ? hidden_out + l * PH : hidden_out;
      UCHECK_HIP(hipMemsetAsync(dst, 0, PH * sizeof(float), st));
      UCHECK_RC(uniter_row_scatter_add(y2, b->pack_dst, dst, M, H, B * L, st));
    }
  }
  m->ready.clear();
  return 0;
}

extern "C" int uniter_model_backward_begin(uniter_model_t* m, const uniter_batch_t* b, const float* d_hidden,
                                           int all_layers, uint64_t seed, uint32_t offset, void* ws,
                                           size_t ws_bytes, void* stream, void* side_stream) {
  UCHECK_ARG(m && d_hidden && ws, "backward_begin: null pointer");
  if (m->plan.mode == 0 || m->hidden_out == nullptr) {
    uniter_set_error("backward_begin: no training-mode forward precedes this call");
    return UNITER_E_STATE;
  }
  UCHECK_ARG(m->seed == seed && m->offset == offset && m->all_layers == all_layers,
             "backward_begin: seed/offset/all_layers differ from the forward call");
  for (int i = 0; i < m->n_params; ++i)
    UCHECK_ARG(m->g[i] != nullptr, "backward: gradient buffer %d (%s) is NULL", i, uniter_param_name(&m->cfg, i));
  (void)ws_bytes; (void)b;
  m->d_hidden = d_hidden;
  m->st = (hipStream_t)stream;
  m->side = side_stream ? (hipStream_t)side_stream : (hipStream_t)stream;
  m->bwd_open = true;
  return 0;
}

extern "C" int uniter_model_backward_layer(uniter_model_t* m, int l) {
  UCHECK_ARG(m, "backward_layer: null model");
  g_uniter_cu_reserve = m->cu_reserve;
  if (!m->bwd_open) { uniter_set_error("backward_layer: call uniter_model_backward_begin first"); return UNITER_E_STATE; }
  const uniter_config_t& c = m->cfg;
  const int H = c.hidden_size, I = c.intermediate_size, nl = c.num_hidden_layers, nh = c.num_attention_heads;
  UCHECK_ARG(l >= 0 && l < nl, "backward_layer: bad layer %d", l);
  Plan& pl = m->plan;
  const int B = pl.B, L = pl.L, M = pl.M;
  const float ph = pl.mode == 1 ? c.hidden_dropout_prob : 0.f;
  const float pa = pl.mode == 1 ? c.attention_probs_dropout_prob : 0.f;
  hipStream_t st = m->st, sd = m->side;
  LayerBufs& lb = pl.layers[l];
  const float* x = l == 0 ? pl.emb : pl.layers[l - 1].y2;
  const size_t MH = (size_t)M * H;

  // upstream gradient w.r.t. this layer's output (packed mode: the caller's padded gradient gathered
  // to the valid rows; gradients at padded positions are dropped, as nothing was computed there)
  const float* dy;
  const float* dh = nullptr;
  if (l == nl - 1 || m->all_layers) {
    dh = m->d_hidden + (m->all_layers ? (size_t)l * ((size_t)B * L * H) : 0);
    if (pl.packed) {
      UCHECK_RC(uniter_row_gather(dh, m->batch.pack_dst, pl.dpk, M, H, B * L, st));
      dh = pl.dpk;
    }
  }
  if (l == nl - 1) {
    dy = dh;
  } else if (m->all_layers) {
    UCHECK_RC(launch_add_f32(pl.dsum, pl.layers[l + 1].dx, dh, MH, st));
    dy = pl.dsum;
  } else {
    dy = pl.layers[l + 1].dx;
  }
  // the dropped gradient in fp32: the operand of the fp32 GEMMs; the bf16-resident GEMMs read its bf16 copy only, so
  // the fp32 store (8 MB per pass) is skipped there
  const bool x3 = pl.x3;
  float* g2 = ph > 0.f ? ((pl.res || x3) ? nullptr : lb.g2) : lb.dz2;
  float* g1 = ph > 0.f ? ((pl.res || x3) ? nullptr : lb.g1) : lb.dz1;
  // precision 2: the next layer's dx may be split-K slabs (summed here); never for the last layer (dy is the caller's
  // gradient), with all_layers (dy is dsum) or in layer 0 (the embedding backward reads a plain dx)
  const int ns_dy = (l == nl - 1 || m->all_layers) ? 1 : pl.ns_k3h;
  const int ns_dx = (l == 0 || m->all_layers) ? 1 : pl.ns_k3h;
  {
    ProfScope ps(m, UNITER_K_LN_BWD, st);
    g_uniter_drop_bits = pl.hid_on ? pl.hid_keepb + (size_t)(2 * l + 1) * pl.hid_stride : nullptr;      // (the forward pass's flags)
    if (x3)
      UCHECK_RC(uniter_ln_bwd_rows_slabs_x3(dy, ns_dy, MH, lb.z2, lb.mean2, lb.rstd2, m->LP(l, L_LN2_G), lb.dz2, g2, lb.g2b, 1, M, H,
                                            ph, m->seed, m->offset, SITE_FFN_OUT(l), lb.ln_ws2, pl.ln_ws_bytes, st));
    else
    UCHECK_RC(uniter_ln_bwd_rows_slabs(dy, ns_dy, MH, lb.z2, lb.mean2, lb.rstd2, m->LP(l, L_LN2_G), lb.dz2, g2,
                                       pl.res ? lb.g2b : nullptr, 1, M, H, ph, m->seed, m->offset, SITE_FFN_OUT(l),
                                       lb.ln_ws2, pl.ln_ws_bytes, st));
  }
  // FFN down dgrad (+ GELU'), FFN up dgrad (+ residual grad)
  // the GEMM's epilogue also emits per-32-row column sums of du (= partial bias gradients of
  // intermediate.dense): saves a 32 MB re-read of du
  const bool fuse_db1 = H % 64 == 0;
  const bool res = pl.res;
  const bool attn_b16 = res && L <= uniter_attn_varlen_max_len();
  // the L <= 192 attention backward kernels also emit the per-sample column sums of dqkv (the fused
  // query|key|value bias gradient before its sum over the batch): no 24 MB re-read of dqkv
  const bool fused_qb = L <= uniter_attn_varlen_max_len();
  // precision 3 with the x3 attention backward: the attention-output input gradient leaves as k-pieces (the kernel sums them)
  static const bool dctx_split = [] { const char* e = getenv("UNITER_DCTX_SPLIT"); return !(e && e[0] == '0'); }();      // A/B switch
  const int ns_dctx = (x3 && fused_qb && dctx_split && attn_x3_products(L, pa, pa > 0.f)) ? pl.ns_kh_b : 1;
  const int epi_du = pl.gelu_d ? UNITER_EPI_MUL : UNITER_EPI_DGELU;
  if (res) {
    UCHECK_RC(gemm_v2(m, UNITER_K_GEMM_DGRAD, st, 1, M, I, H, lb.g2b, H, m->WB(l, L_W2), I, nullptr, I, 1, lb.dub, I,
                      epi_du, nullptr, lb.u, 1, nullptr, 0, I, b16_colpart_on(pl) ? lb.du_csum : nullptr));
    UCHECK_RC(gemm_v2(m, UNITER_K_GEMM_DGRAD, st, 1, M, H, I, lb.dub, I, m->WB(l, L_W1), H, lb.dy1, H, pl.ns_ki, nullptr, 0,
                      UNITER_EPI_ADD, nullptr, lb.dz2, 0, nullptr, 0, H));
  } else if (x3) {
    // dU = (g2 . W2) * gelu'(u) exists only as x3 pieces: operand of the next product and of intermediate.dense's weight gradient
    // (with riders on the 128 x 256 weight-gradient geometry the bias gradient of intermediate.dense -- dU's column sums -- starts
    // here: one row of partial sums per 64 rows of dU, finished by a reduction job of the riders)
    UCHECK_RC(gemm_x3(m, UNITER_K_GEMM_DGRAD, st, 1, M, I, H, lb.g2b, m->WB(l, L_W2), I, nullptr, 1, lb.dub, UNITER_EPI_MUL,
                      nullptr, lb.u, nullptr, x3_colpart_on() ? lb.du_csum : nullptr));
    UCHECK_RC(gemm_x3(m, UNITER_K_GEMM_DGRAD, st, 1, M, H, I, lb.dub, m->WB(l, L_W1), H, lb.dy1, pl.ns_ki_b, nullptr, UNITER_EPI_ADD,
                      nullptr, lb.dz2, nullptr));
  } else {
    UCHECK_RC(gemm(m, UNITER_K_GEMM_DGRAD, st, 0, 1, M, I, H, g2, H, m->LP(l, L_W2), I, lb.du, I, epi_du,
                   nullptr, lb.u, nullptr, I, 0, fuse_db1 ? lb.du_csum : nullptr));
    UCHECK_RC(gemm(m, UNITER_K_GEMM_DGRAD, st, 0, 1, M, H, I, lb.du, I, m->LP(l, L_W1), H, lb.dy1, H, UNITER_EPI_ADD,
                   nullptr, lb.dz2, nullptr, H, 0));
  }
  {
    ProfScope ps(m, UNITER_K_LN_BWD, st);
    g_uniter_drop_bits = pl.hid_on ? pl.hid_keepb + (size_t)(2 * l) * pl.hid_stride : nullptr;
    if (x3)
      UCHECK_RC(uniter_ln_bwd_rows_slabs_x3(lb.dy1, pl.ns_ki_b, MH, lb.z1, lb.mean1, lb.rstd1, m->LP(l, L_LN1_G), lb.dz1, g1, lb.g1b,
                                            1, M, H, ph, m->seed, m->offset, SITE_ATTN_OUT(l), lb.ln_ws1, pl.ln_ws_bytes, st));
    else
    UCHECK_RC(uniter_ln_bwd_rows_slabs(lb.dy1, res ? pl.ns_ki : 1, MH, lb.z1, lb.mean1, lb.rstd1, m->LP(l, L_LN1_G), lb.dz1,
                                       g1, res ? lb.g1b : nullptr, 1, M, H, ph, m->seed, m->offset, SITE_ATTN_OUT(l),
                                       lb.ln_ws1, pl.ln_ws_bytes, st));
  }
  const bool wg_early = res && pl.wg_group >= 2 && sd != st;
  if (wg_early) {
    // the three weight gradients whose operands exist now (FFN down / up, attention output: 324 whole-K tiles) start on
    // the side stream here, next to the attention backward (192 workgroups with large LDS: a quarter of the CUs idle)
    UCHECK_HIP(hipEventRecord(m->ev_mid[l], st));
    UCHECK_HIP(hipStreamWaitEvent(sd, m->ev_mid[l], 0));
    const int Mo[3] = {I, H, H}, No[3] = {H, I, H};
    const void* const As[3] = {lb.dub, lb.g2b, lb.g1b};
    const void* const Bs[3] = {lb.y1b, lb.hactb, lb.ctxb};
    float* const dWs[3] = {m->LG(l, L_W1), m->LG(l, L_W2), m->LG(l, L_OW)};
    {
      ProfScope ps(m, UNITER_K_GEMM_WGRAD, sd);
      UCHECK_RC(gemm_bf16v2_wgrad_group(1, 3, Mo, No, M, As, Bs, dWs, sd, m->wg_overwrite ? 1 : 0, 0, nullptr));
    }
    UCHECK_RC(uniter_colsum_bf16_add(lb.dub, M, I, I, m->LG(l, L_B1), sd));      // intermediate.dense bias gradient
  }
  if (res) {
    UCHECK_RC(gemm_v2(m, UNITER_K_GEMM_DGRAD, st, 1, M, H, H, lb.g1b, H, m->WB(l, L_OW), H, lb.dctx, H, 1, nullptr, 0,
                      UNITER_EPI_NONE, nullptr, nullptr, 0, nullptr, 0, 0));
  } else if (x3) {
    // (two k-pieces when the x3 attention backward follows: it sums the slabs while it reads them -- 126 tiles do not fill the chip)
    UCHECK_RC(gemm_x3(m, UNITER_K_GEMM_DGRAD, st, 1, M, H, H, lb.g1b, m->WB(l, L_OW), H, lb.dctx, ns_dctx, nullptr, UNITER_EPI_NONE,
                      nullptr, nullptr, nullptr));
  } else {
    UCHECK_RC(gemm(m, UNITER_K_GEMM_DGRAD, st, 0, 1, M, H, H, g1, H, m->LP(l, L_OW), H, lb.dctx, H, UNITER_EPI_NONE,
                   nullptr, nullptr, nullptr, 0, 0));
  }
  {
    ProfScope ps(m, UNITER_K_ATTN_BWD, st);
    if (attn_b16 && attn_b16x(L, pa, pa > 0.f))      // (the forward pass of a saved plan left the keep flags)
      UCHECK_RC(uniter_attn_b16x_bwd(lb.qkv, 1, pl.packed ? nullptr : m->batch.attention_mask,
                                     pl.packed ? m->batch.cu_seqlens : nullptr, lb.ctx, lb.lse, lb.dctx, nullptr, lb.dqkvb,
                                     lb.qb_part, pa > 0.f ? lb.keepb : nullptr, lb.delta, B, L, nh, pa, st));
    else if (attn_b16)
      UCHECK_RC(uniter_attn_bf16_bwd(lb.qkv, 1, pl.packed ? nullptr : m->batch.attention_mask,
                                     pl.packed ? m->batch.cu_seqlens : nullptr, lb.ctx, lb.lse, lb.dctx,
                                     nullptr /* fp32 dqkv has no reader in this mode: bias partials are fused */,
                                     lb.dqkvb, lb.qb_part, lb.keepb, lb.delta, B, L, nh, pa, m->seed, m->offset,
                                     SITE_ATTN_PROBS(l), pl.attn_ws, pl.attn_ws_bytes, st));
    else if (fused_qb && x3 && attn_x3_products(L, pa, pa > 0.f))      // (the forward pass of a saved plan left the keep flags)
      UCHECK_RC(uniter_attn_x3_bwd(lb.qkv, pl.packed ? nullptr : m->batch.attention_mask,
                                   pl.packed ? m->batch.cu_seqlens : nullptr, lb.ctx, lb.lse, lb.dctx, ns_dctx, (size_t)M * H,
                                   nullptr, lb.dqkvb, lb.qb_part, pa > 0.f ? lb.keepb : nullptr, lb.delta, B, L, nh, pa, st));
    else if (fused_qb && x3)      // dqkv leaves the kernel as x3 pieces only: nothing reads an fp32 copy in this mode
      UCHECK_RC(uniter_attn_bwd_ex_x3(lb.qkv, pl.packed ? nullptr : m->batch.attention_mask,
                                      pl.packed ? m->batch.cu_seqlens : nullptr, lb.ctx, lb.lse, lb.dctx, nullptr, lb.dqkvb,
                                      lb.qb_part, lb.keepb, lb.delta, B, L, nh, pa, m->seed, m->offset, SITE_ATTN_PROBS(l),
                                      pl.attn_ws, pl.attn_ws_bytes, st));
    else if (fused_qb)
      UCHECK_RC(uniter_attn_bwd_ex(lb.qkv, pl.packed ? nullptr : m->batch.attention_mask,
                                   pl.packed ? m->batch.cu_seqlens : nullptr, lb.ctx, lb.lse, lb.dctx, lb.dqkv, nullptr,
                                   lb.qb_part, lb.keepb, lb.delta, B, L, nh, pa, m->seed, m->offset, SITE_ATTN_PROBS(l),
                                   pl.attn_ws, pl.attn_ws_bytes, st));
    else
      UCHECK_RC(uniter_attn_bwd(lb.qkv, m->batch.attention_mask, lb.ctx, lb.lse, lb.dctx, lb.dqkv, lb.delta, B, L,
                                nh, pa, m->seed, m->offset, SITE_ATTN_PROBS(l), pl.attn_ws, pl.attn_ws_bytes, st));
  }
  if (res) {
    if (!attn_b16) UCHECK_RC(cast_b(lb.dqkv, lb.dqkvb, (size_t)M * 3 * H, st));
    UCHECK_RC(gemm_v2(m, UNITER_K_GEMM_DGRAD, st, 1, M, H, 3 * H, lb.dqkvb, 3 * H, m->WB(l, L_QW), H, lb.dx, H, ns_dx,
                      nullptr, 0, UNITER_EPI_ADD, nullptr, lb.dz1, 0, nullptr, 0, H));
  } else if (x3) {
    if (!fused_qb) UCHECK_RC(split_x3(lb.dqkv, lb.dqkvb, M, 3 * H, st));
    UCHECK_RC(gemm_x3(m, UNITER_K_GEMM_DGRAD, st, 1, M, H, 3 * H, lb.dqkvb, m->WB(l, L_QW), H, lb.dx, ns_dx, nullptr, UNITER_EPI_ADD,
                      nullptr, lb.dz1, nullptr));
  } else {
    UCHECK_RC(gemm(m, UNITER_K_GEMM_DGRAD, st, 0, 1, M, H, 3 * H, lb.dqkv, 3 * H, m->LP(l, L_QW), H, lb.dx, H,
                   UNITER_EPI_ADD, nullptr, lb.dz1, nullptr, H, 0));
  }

  // weight / bias gradients on the side stream (accumulate into the bound grad buffers)
  if (sd != st) {
    UCHECK_HIP(hipEventRecord(m->ev_main[l], st));
    UCHECK_HIP(hipStreamWaitEvent(sd, m->ev_main[l], 0));
  }
  // precision 3 (round 5): the column reductions below, the bias gradient of intermediate.dense and the layer's share of the clip
  // norm RIDE on the grouped weight-gradient launch (uniter_wgrad_x3_group_riders) -- no finalize / column-sum / sum-of-squares
  // launches on this stream.  UNITER_X3_RIDERS=0 keeps the separate launches (A/B measurements)
  const bool riders_b16 = b16_riders_on(pl);
  const bool riders_on = (x3 && x3_riders_enabled()) || riders_b16;
  // LayerNorm / dense-bias gradients: the column reductions of the two row passes above, the attention backward's
  // per-sample query|key|value bias partials and (fp32 mode) the dU column partials -- ONE launch for all of them
  if (!riders_on) {
    const int lnp = ln_bwd_partial_rows(M);
    const float* parts[4] = {(const float*)lb.ln_ws2, (const float*)lb.ln_ws1, fused_qb ? lb.qb_part : nullptr,
                             (!res && !x3 && fuse_db1) ? lb.du_csum : nullptr};
    const int nparts[4] = {lnp, lnp, B, (M + 31) / 32};
    const size_t strides[4] = {(size_t)3 * H, (size_t)3 * H, (size_t)3 * H, (size_t)I};
    float* const outs[4][3] = {{m->LG(l, L_LN2_G), m->LG(l, L_LN2_B), m->LG(l, L_B2)},
                               {m->LG(l, L_LN1_G), m->LG(l, L_LN1_B), m->LG(l, L_OB)},
                               {m->LG(l, L_QB), nullptr, nullptr},
                               {m->LG(l, L_B1), nullptr, nullptr}};
    const int nout[4] = {3, 3, 1, 1};
    const int seg[4] = {H, H, 3 * H, I};
    UCHECK_RC(finalize_partials_jobs(4, parts, nparts, strides, outs, nout, seg, sd));
  }
  if (res) {
    const unsigned short* xb = l == 0 ? pl.embb : pl.layers[l - 1].y2b;
    if (wg_early) {
      const int Mo[1] = {3 * H}, No[1] = {H};
      const void* const As[1] = {lb.dqkvb};
      const void* const Bs[1] = {xb};
      float* const dWs[1] = {m->LG(l, L_QW)};
      ProfScope ps(m, UNITER_K_GEMM_WGRAD, sd);
      UCHECK_RC(gemm_bf16v2_wgrad_group(1, 1, Mo, No, M, As, Bs, dWs, sd, m->wg_overwrite ? 1 : 0, 0, nullptr));
    } else if (pl.wg_group) {
      // all four weight gradients of the layer as one launch: 432 whole-K tiles for 512 workgroup slots (UNITER-base)
      const int Mo[4] = {I, H, 3 * H, H}, No[4] = {H, I, H, H};
      const void* const As[4] = {lb.dub, lb.g2b, lb.dqkvb, lb.g1b};
      const void* const Bs[4] = {lb.y1b, lb.hactb, xb, lb.ctxb};
      float* const dWs[4] = {m->LG(l, L_W1), m->LG(l, L_W2), m->LG(l, L_QW), m->LG(l, L_OW)};
      const int b16_wgs = 0;      // (the grid cap of UNITER_WGRAD_GROUP_WGS / its default)
      if (riders_b16) {
        uniter_x3_riders_t x;
        memset(&x, 0, sizeof(x));
        const bool p7 = b16_p7(), colpart = b16_colpart_on(pl);
        const int slots = p7 ? gemm_b1p_wgrad_group_slots(4, Mo, No, b16_wgs) : gemm_bf16v2_wgrad_group_slots(4, Mo, No, b16_wgs);
        if (m->norm_parts) {
          UCHECK_ARG((size_t)slots <= m->norm_stride, "backward_layer: %d clip-norm slots per layer, room for %zu (uniter_model_set_norm_partials)",
                     slots, m->norm_stride);
          x.ssq = m->norm_parts + (size_t)l * m->norm_stride;
        }
        fill_layer_riders(x, m, l, lb, M, B, H, fused_qb, colpart);
        ProfScope ps(m, UNITER_K_GEMM_WGRAD, sd);
        UCHECK_RC(gemm_bf16v2_wgrad_group(p7 ? 7 : 1, 4, Mo, No, M, As, Bs, dWs, sd, m->wg_overwrite ? 1 : 0, b16_wgs, &x));
      } else {
        {
          ProfScope ps(m, UNITER_K_GEMM_WGRAD, sd);
          // (round 6: 216 whole-K tiles of 128 x 256 on the persistent kernel -- 47 against 61 us alone, UNITER_B16_PERSIST bit 4)
          const int wcfg = pl.wg_group == 4 ? 4 : ((b16_persist_mask() & 4) ? 7 : 1);
          UCHECK_RC(gemm_bf16v2_wgrad_group(wcfg, 4, Mo, No, M, As, Bs, dWs, sd, m->wg_overwrite ? 1 : 0, b16_wgs, nullptr));
        }
        UCHECK_RC(uniter_colsum_bf16_add(lb.dub, M, I, I, m->LG(l, L_B1), sd));      // intermediate.dense bias gradient
      }
    } else {
      UCHECK_RC(wgrad_b16(m, pl, sd, H, I, M, lb.g2b, lb.hactb, m->LG(l, L_W2)));
      UCHECK_RC(wgrad_b16(m, pl, sd, I, H, M, lb.dub, lb.y1b, m->LG(l, L_W1)));
      UCHECK_RC(uniter_colsum_bf16_add(lb.dub, M, I, I, m->LG(l, L_B1), sd));      // intermediate.dense bias gradient
      UCHECK_RC(wgrad_b16(m, pl, sd, H, H, M, lb.g1b, lb.ctxb, m->LG(l, L_OW)));
      UCHECK_RC(wgrad_b16(m, pl, sd, 3 * H, H, M, lb.dqkvb, xb, m->LG(l, L_QW)));
    }
  } else if (x3) {
    // the layer's four weight gradients as ONE persistent launch of whole-K 128 x 128 tiles on x3 operands (432 tiles for
    // UNITER-base), no atomics; UNITER_WGRAD_X3_WGS caps its grid (0 = one workgroup per CU)
    const unsigned short* xb = l == 0 ? pl.embb : pl.layers[l - 1].y2b;
    const int Mo[4] = {I, H, 3 * H, H}, No[4] = {H, I, H, H};
    const void* const As[4] = {lb.dub, lb.g2b, lb.dqkvb, lb.g1b};
    const void* const Bs[4] = {lb.y1b, lb.hactb, xb, lb.ctxb};
    float* const dWs[4] = {m->LG(l, L_W1), m->LG(l, L_W2), m->LG(l, L_QW), m->LG(l, L_OW)};
    static const int x3_cfg = [] { const char* e = getenv("UNITER_X3_CFG"); return e ? atoi(e) : 0; }();
    static const int x3_wgs_env = [] { const char* e = getenv("UNITER_WGRAD_X3_WGS"); return e ? atoi(e) : -1; }();
    // layer 0's launch runs BEHIND the input-gradient chain, beside the embedding backward's row passes on the main stream: the
    // smallest grid that needs no more rounds (432 tiles: 216 workgroups walk two tiles each, as 256 would) leaves 40 CUs to
    // those passes -- their 190-register waves find no room on a CU a persistent 144-KB workgroup holds (img_embed_bwd 112 us
    // for 576 rows when it has to wait for one).  UNITER_WGRAD_X3_WGS = n forces a grid for every layer (0 = one per CU)
    const int x3_wgs = x3_wgs_env >= 0 ? x3_wgs_env : (l == 0 ? gemm_x3_wgrad_group_balanced_wgs(0, 4, Mo, No) : 0);
    if (riders_on) {
      // product 0 is dW1 = dU^T y1: its A operand's column sums are intermediate.dense's bias gradient
      uniter_x3_riders_t x;
      memset(&x, 0, sizeof(x));
      const int slots = gemm_x3_wgrad_group_slots(0, 4, Mo, No, x3_wgs, M, pl.sk_side ? pl.sk_bytes : 0);
      if (m->norm_parts) {
        UCHECK_ARG((size_t)slots <= m->norm_stride, "backward_layer: %d clip-norm slots per layer, room for %zu (uniter_model_set_norm_partials)",
                   slots, m->norm_stride);
        x.ssq = m->norm_parts + (size_t)l * m->norm_stride;
      }
      fill_layer_riders(x, m, l, lb, M, B, H, fused_qb, x3_colpart_on());
      ProfScope ps(m, UNITER_K_GEMM_WGRAD, sd);
      UCHECK_RC(gemm_x3_wgrad_group(x3_cfg == 3 ? 3 : 0, 4, Mo, No, M, As, Bs, dWs, sd, m->wg_overwrite ? 1 : 0, x3_wgs, &x,
                                    pl.sk_side, pl.sk_side ? pl.sk_bytes : 0));
    } else {
      {
        ProfScope ps(m, UNITER_K_GEMM_WGRAD, sd);
        UCHECK_RC(gemm_x3_wgrad_group(x3_cfg, 4, Mo, No, M, As, Bs, dWs, sd, m->wg_overwrite ? 1 : 0, x3_wgs, nullptr,
                                      pl.sk_side, pl.sk_side ? pl.sk_bytes : 0));
      }
      UCHECK_RC(uniter_colsum_x3_add(lb.dub, M, I, I, m->LG(l, L_B1), sd));      // intermediate.dense bias gradient
    }
  } else {
    const int wb = m->wg_overwrite ? -1 : 1;      // -1: overwrite (or clear, then accumulate) -- see gemm()
    // UNITER_WGRAD_GROUP_F32=1: EVERY layer's four weight gradients as ONE persistent launch of whole-K tiles (1728 tiles for
    // 1024 slots, uniter_wgrad_f32_group).  Measured (three same-box pairs): the launch itself runs at 0.47-0.49 of the fp32 peak
    // in situ (four launches: 0.39), both gradient families together at 0.78 (0.67), the attention backward beside it at 97
    // instead of 151 us -- and the STEP is 0.7-1.1 % slower (13.87-13.90 vs 13.72-13.75 ms): its 1024 resident workgroups
    // hold every CU's LDS for 370 us, and each kernel of the input-gradient chain queues for slots before its first
    // workgroup starts (a wait no per-launch figure shows).  So layers nl-1 .. 1 keep four launches.
    // Layer 0 is the exception (UNITER_WGRAD_GROUP_F32 = 2, the default; 0 = never, 1 = every layer): its weight gradients run
    // BEHIND the input-gradient chain, alone on the chip, where four launches of 576 / 576 / 144 / 432 tiles leave a quarter
    // of the 1024 slots idle and one balanced launch does not: 13.69 -> 13.55 ms per step (three same-box pairs).
    static const int group_mode = [] { const char* e = getenv("UNITER_WGRAD_GROUP_F32"); return e ? atoi(e) : 2; }();
    const bool group_env = group_mode == 1 || (group_mode == 2 && l == 0);
    static const bool whole_all = [] { const char* e = getenv("UNITER_WGRAD_WHOLE"); return !e || atoi(e) == 15; }();
    static const bool cfg_default = [] { const char* e = getenv("UNITER_WGRAD_CFG"); return !e || atoi(e) == 0; }();
    bool grouped = false;
    if (group_env && whole_all && cfg_default && (m->precision == 0) && fuse_db1) {
      const int Mo[4] = {H, I, H, 3 * H}, No[4] = {I, H, H, H};
      const float* const As[4] = {g2, lb.du, g1, lb.dqkv};
      const float* const Bs[4] = {lb.hact, lb.y1, lb.ctx, x};
      float* const dWs[4] = {m->LG(l, L_W2), m->LG(l, L_W1), m->LG(l, L_OW), m->LG(l, L_QW)};
      ProfScope ps(m, UNITER_K_GEMM_WGRAD, sd);
      const int rc = gemm_f32_wgrad_group(4, Mo, No, M, As, Bs, dWs, m->wg_overwrite ? 1 : 0, sd);
      if (rc == 0) grouped = true;
      else if (rc != UNITER_E_SHAPE) return rc;
    }
    if (!grouped) {
    UCHECK_RC(gemm(m, UNITER_K_GEMM_WGRAD, sd, 1, 1, H, I, M, g2, H, lb.hact, I, m->LG(l, L_W2), I, UNITER_EPI_NONE,
                   nullptr, nullptr, nullptr, 0, wb));
    UCHECK_RC(gemm(m, UNITER_K_GEMM_WGRAD, sd, 1, 1, I, H, M, lb.du, I, lb.y1, H, m->LG(l, L_W1), H, UNITER_EPI_NONE,
                   nullptr, nullptr, nullptr, 0, wb));
    if (!fuse_db1) UCHECK_RC(uniter_colsum_f32(lb.du, M, I, I, m->LG(l, L_B1), 1, pl.col_ws, pl.col_ws_bytes, sd));
    UCHECK_RC(gemm(m, UNITER_K_GEMM_WGRAD, sd, 1, 1, H, H, M, g1, H, lb.ctx, H, m->LG(l, L_OW), H, UNITER_EPI_NONE,
                   nullptr, nullptr, nullptr, 0, wb));
    UCHECK_RC(gemm(m, UNITER_K_GEMM_WGRAD, sd, 1, 1, 3 * H, H, M, lb.dqkv, 3 * H, x, H, m->LG(l, L_QW), H,
                   UNITER_EPI_NONE, nullptr, nullptr, nullptr, 0, wb));
    }
  }
  if (!fused_qb) UCHECK_RC(uniter_colsum_f32(lb.dqkv, M, 3 * H, 3 * H, m->LG(l, L_QB), 1, pl.col_ws, pl.col_ws_bytes, sd));
  if (sd != st) UCHECK_HIP(hipEventRecord(m->ev_side[l], sd));
  return 0;
}

extern "C" int uniter_model_backward_embed(uniter_model_t* m) {
  UCHECK_ARG(m, "backward_embed: null model");
  g_uniter_cu_reserve = m->cu_reserve;
  if (!m->bwd_open) { uniter_set_error("backward_embed: call uniter_model_backward_begin first"); return UNITER_E_STATE; }
  const uniter_config_t& c = m->cfg;
  const int H = c.hidden_size;
  Plan& pl = m->plan;
  const uniter_batch_t& b = m->batch;
  const int B = pl.B, T = pl.T, R = pl.R, L = pl.L, S = pl.S;
  const float ph = pl.mode == 1 ? c.hidden_dropout_prob : 0.f;
  hipStream_t st = m->st, sd = m->side;
  const float* demb = pl.layers[0].dx;
  const bool joint = pl.has_txt && pl.has_img;
  if (pl.packed) {
    UCHECK_HIP(hipMemsetAsync(pl.dcat, 0, (size_t)B * S * H * sizeof(float), st));
    UCHECK_RC(uniter_row_scatter_add(demb, b.pack_src, pl.dcat, pl.M, H, B * S, st));
  } else {
    UCHECK_RC(uniter_gather_rows_bwd(demb, joint ? b.gather_index : nullptr, pl.dcat, B, S, L, H, st));
  }
  // Round 6: the tail of the step is a chain of latency-bound launches behind the last input gradient (text-embedding backward 55 us,
  // image-embedding backward 67, the 7-d projection's and the region projection's weight gradients 15 + 52, column sums: 220 us on one
  // stream, profiles/r06_timeline_f32x3.txt), while layer 0's weight-gradient launch holds most of the chip on the side stream.  The
  // text branch and the image branch share nothing but dcat: with an auxiliary stream the text branch (and, behind the image rows'
  // pass, the region projection's bias column sums) runs there, beside the image branch.  Same kernels, same results; the token-type
  // table's gradient is the one buffer both branches add to (row 0 / the text rows' types there, row 1 / the image rows' types here):
  // with explicit type ids both scatter by float atomics, without them each finalizes into its own row -- no ordering between them
  // is needed either way.  UNITER_EMBED_BWD_PAR=0 keeps one stream.
  static const bool par_env = [] { const char* e = getenv("UNITER_EMBED_BWD_PAR"); return !(e && e[0] == '0'); }();
  hipStream_t ax = (par_env && m->aux && m->aux != st && m->aux != sd && pl.has_txt && pl.has_img) ? m->aux : nullptr;
  if (ax) {
    if (!m->ev_emb0) UCHECK_HIP(hipEventCreateWithFlags(&m->ev_emb0, hipEventDisableTiming));
    if (!m->ev_emb1) UCHECK_HIP(hipEventCreateWithFlags(&m->ev_emb1, hipEventDisableTiming));
    if (!m->ev_emb2) UCHECK_HIP(hipEventCreateWithFlags(&m->ev_emb2, hipEventDisableTiming));
    UCHECK_HIP(hipEventRecord(m->ev_emb0, st));               // dcat is complete
    UCHECK_HIP(hipStreamWaitEvent(ax, m->ev_emb0, 0));
  }
  if (pl.has_txt)
    UCHECK_RC(uniter_txt_embed_bwd(pl.dcat, b.input_ids, b.position_ids, b.txt_type_ids, m->P(P_WORD),
                                   m->P(P_POS), m->P(P_TYPE), m->P(P_ELN_G), m->G(P_WORD), m->G(P_POS),
                                   m->G(P_TYPE), m->G(P_ELN_G), m->G(P_ELN_B), B, T, S, H, c.vocab_size,
                                   c.max_position_embeddings, c.type_vocab_size, b.pos_bcast, ph, m->seed,
                                   m->offset, ax ? pl.emb_ws2 : pl.emb_ws, pl.emb_ws_bytes, ax ? ax : st));
  if (pl.has_img) {
    UCHECK_RC(uniter_img_embed_bwd(pl.dcat, pl.imgfc, b.img_pos_feat, b.img_type_ids, m->P(P_POSL_W),
                                   m->P(P_POSL_B), m->P(P_TYPE), m->P(P_ILN_G), m->P(P_ILN_B), m->P(P_PLN_G),
                                   m->P(P_PLN_B), m->P(P_FLN_G), pl.img_stats, pl.d_imgfc, pl.d_posfc,
                                   m->G(P_POSL_W), m->G(P_POSL_B), m->G(P_TYPE), m->G(P_ILN_G), m->G(P_ILN_B),
                                   m->G(P_PLN_G), m->G(P_PLN_B), m->G(P_FLN_G), m->G(P_FLN_B), B, R, pl.T0, S, H,
                                   c.type_vocab_size, ph, m->seed, m->offset, pl.emb_ws, pl.emb_ws_bytes, st));
    const float* feat = b.img_masks ? pl.feat_eff : b.img_feat;
    // NOT pl.col_ws: the side stream's bias-gradient reductions of layer 0 may still be using it
    if (ax) {      // the region projection's bias column sums beside its weight gradient: behind the text branch on the auxiliary stream
      UCHECK_HIP(hipEventRecord(m->ev_emb1, st));             // d_imgfc is written
      UCHECK_HIP(hipStreamWaitEvent(ax, m->ev_emb1, 0));
      UCHECK_RC(uniter_colsum_f32(pl.d_imgfc, B * R, H, H, m->G(P_IMG_B), 1, pl.emb_ws3, pl.emb_ws_bytes, ax));
    }
    UCHECK_RC(gemm(m, UNITER_K_GEMM_WGRAD, st, 1, 1, H, c.img_dim, B * R, pl.d_imgfc, H, feat, c.img_dim,
                   m->G(P_IMG_W), c.img_dim, UNITER_EPI_NONE, nullptr, nullptr, nullptr, 0, 1));
    if (!ax) UCHECK_RC(uniter_colsum_f32(pl.d_imgfc, B * R, H, H, m->G(P_IMG_B), 1, pl.emb_ws, pl.emb_ws_bytes, st));
    if (b.img_masks) {
      // d(img_feat + mask_emb[img_masks]) = d_imgfc @ W_img; row 1 of mask_embedding sums the masked rows
      UCHECK_RC(gemm(m, UNITER_K_GEMM_DGRAD, st, 0, 1, B * R, c.img_dim, H, pl.d_imgfc, H, m->P(P_IMG_W),
                     c.img_dim, pl.d_feat, c.img_dim, UNITER_EPI_NONE, nullptr, nullptr, nullptr, 0, 0));
      UCHECK_RC(launch_masked_rowsum(pl.d_feat, b.img_masks, m->G(P_MASK_EMB) + c.img_dim, B * R, c.img_dim, st));
    }
  }
  if (ax) {           // join the auxiliary stream's branch
    UCHECK_HIP(hipEventRecord(m->ev_emb2, ax));
    UCHECK_HIP(hipStreamWaitEvent(st, m->ev_emb2, 0));
  }
  if (sd != st) {   // join: everything the caller does next on `stream` sees all gradients
    UCHECK_HIP(hipEventRecord(m->ev_side[c.num_hidden_layers], sd));
    UCHECK_HIP(hipStreamWaitEvent(st, m->ev_side[c.num_hidden_layers], 0));
  }
  m->bwd_open = false;
  m->wg_overwrite = false;       // one backward pass: the next one accumulates again
  return 0;
}

extern "C" int uniter_model_set_cu_reserve(uniter_model_t* m, int cus) {
  UCHECK_ARG(m && cus >= 0 && cus <= 128, "set_cu_reserve: 0 .. 128 CUs");
  m->cu_reserve = cus;
  return 0;
}

extern "C" int uniter_model_set_aux_stream(uniter_model_t* m, void* aux_stream) {
  UCHECK_ARG(m, "set_aux_stream: null model");
  m->aux = (hipStream_t)aux_stream;
  return 0;
}

extern "C" int uniter_model_set_norm_partials(uniter_model_t* m, double* parts, size_t stride_doubles) {
  UCHECK_ARG(m && (parts == nullptr || stride_doubles > 0), "set_norm_partials: bad argument");
  m->norm_parts = parts;
  m->norm_stride = parts ? stride_doubles : 0;
  return 0;
}

extern "C" int uniter_model_norm_partials_per_layer(const uniter_model_t* m) {
  if (!m) return 0;
  const int H = m->cfg.hidden_size, I = m->cfg.intermediate_size;
  const int Mo[4] = {I, H, 3 * H, H}, No[4] = {H, I, H, H};
  if (m->precision == 2) {
    // (the grouped launch is the default form of this precision: UNITER_WGRAD_GROUP unset or 1; very long batches fall back)
    // (answers for the plan of the LAST forward: the grouped launch is that plan's choice -- very long batches keep stream-K)
    if (!(m->plan.mode != 0 && b16_riders_on(m->plan))) return 0;
    // (as in the fp32x3 mode: a plan beyond the fused query|key|value bias partials announces no slots -- ADVICE r05)
    if (m->plan.L > uniter_attn_varlen_max_len()) return 0;
    return b16_p7() ? gemm_b1p_wgrad_group_slots(4, Mo, No, 0) : gemm_bf16v2_wgrad_group_slots(4, Mo, No, 0);
  }
  if (m->precision != 3 || !x3_riders_enabled()) return 0;
  //  a plan whose joint length is beyond the fused query|key|value bias partials (L > uniter_attn_varlen_max_len(), or
  // UNITER_ATTN_SPLIT=0) writes that bias gradient by a separate column-sum launch BEHIND the riders' launch: its squares would reach
  // no slot.  Such a plan announces no slots at all -- the caller reduces the layer's bucket itself, as in the other precisions
  if (m->plan.mode != 0 && m->plan.L > uniter_attn_varlen_max_len()) return 0;
  static const int x3_wgs = [] { const char* e = getenv("UNITER_WGRAD_X3_WGS"); return e && atoi(e) > 0 ? atoi(e) : 0; }();
  // (the plan of the LAST forward: its row count is the reduction length, its workspace the balanced walk's)
  return gemm_x3_wgrad_group_slots(0, 4, Mo, No, x3_wgs, m->plan.M, m->plan.sk_side ? m->plan.sk_bytes : 0);
}

extern "C" int uniter_model_set_wgrad_overwrite(uniter_model_t* m, int on) {
  UCHECK_ARG(m, "set_wgrad_overwrite: null model");
  m->wg_overwrite = on != 0;
  return 0;
}

extern "C" int uniter_model_backward(uniter_model_t* m, const uniter_batch_t* b, const float* d_hidden,
                                     int all_layers, uint64_t seed, uint32_t offset, void* ws, size_t ws_bytes,
                                     void* stream, void* side_stream) {
  UCHECK_RC(uniter_model_backward_begin(m, b, d_hidden, all_layers, seed, offset, ws, ws_bytes, stream, side_stream));
  for (int l = m->cfg.num_hidden_layers - 1; l >= 0; --l) UCHECK_RC(uniter_model_backward_layer(m, l));
  return uniter_model_backward_embed(m);
}

extern "C" uint64_t uniter_model_generation(const uniter_model_t* m) { return m ? m->generation : 0; }

extern "C" int uniter_model_set_ready_events(uniter_model_t* m, void* const* events, int n) {
  UCHECK_ARG(m && (n == 0 || (events && (n == m->cfg.num_hidden_layers + 1 || n == m->cfg.num_hidden_layers + 2))),
             "set_ready_events: need num_hidden_layers + 1 events (embeddings, layer 0 ..), one more (the word table, last) or n = 0");
  m->ready.clear();
  for (int i = 0; i < n; ++i) {
    UCHECK_ARG(events[i], "set_ready_events: event %d is NULL", i);
    m->ready.push_back((hipEvent_t)events[i]);
  }
  return 0;
}

extern "C" int uniter_model_set_weight_mirror(uniter_model_t* m, const float* flat_base, const void* mirror_bf16,
                                              size_t numel) {
  UCHECK_ARG(m, "set_weight_mirror: null model");
  if (!mirror_bf16) { m->mirror = nullptr; m->mirror_base = nullptr; m->mirror_numel = 0; return 0; }
  UCHECK_ARG(flat_base && numel > 0, "set_weight_mirror: bad argument");
  const uniter_config_t& c = m->cfg;
  static const int ws[4] = {L_QW, L_OW, L_W1, L_W2};
  for (int l = 0; l < c.num_hidden_layers; ++l)
    for (int k = 0; k < 4; ++k) {
      const float* p = m->LP(l, ws[k]);
      UCHECK_ARG(p >= flat_base && p < flat_base + numel && (((uintptr_t)(p - flat_base) * 2) & 15) == 0,
                 "set_weight_mirror: layer %d weight %d is not inside the flat buffer on a 16-byte bf16 boundary", l, k);
    }
  m->mirror = (const unsigned short*)mirror_bf16; m->mirror_base = flat_base; m->mirror_numel = numel;
  return 0;
}

extern "C" int uniter_model_set_weight_pairing(uniter_model_t* m, int on) {
  UCHECK_ARG(m, "set_weight_pairing: null model");
  UCHECK_SHAPE(!on || (m->cfg.hidden_size % 64 == 0 && m->cfg.intermediate_size % 64 == 0),
               "set_weight_pairing: hidden and intermediate sizes must be multiples of 64 (a 64-element chunk of the optimizer is two whole 32-element units of one row)");
  m->mirror_paired = on != 0;
  return 0;
}

extern "C" int uniter_model_set_precision(uniter_model_t* m, int precision) {
  UCHECK_ARG(m && precision >= 0 && precision <= 3, "set_precision: 0 (fp32), 1 (bf16 MFMA, fp32 operands), 2 (bf16-resident operands) "
             "or 3 (fp32 from three bf16 pieces)");
  UCHECK_ARG(precision < 2 || m->mirror, "set_precision: precision 2 / 3 needs uniter_model_set_weight_mirror first");
  UCHECK_SHAPE(precision != 3 || (m->cfg.intermediate_size % 32 == 0 && m->cfg.hidden_size % 32 == 0),
               "set_precision: the fp32x3 mode needs hidden_size and intermediate_size %% 32 == 0 (got %d, %d): every forward and "
               "input-gradient product runs on 32-deep k-tiles", m->cfg.hidden_size, m->cfg.intermediate_size);
  UCHECK_SHAPE(precision != 3 || 2 * m->mirror_numel * 2 + (size_t)4096 * m->cfg.intermediate_size * 2 < (1ull << 31),
               "set_precision: the fp32x3 mode addresses the weight pieces with 31-bit offsets (%zu parameters are too many)", m->mirror_numel);
  UCHECK_SHAPE(precision != 2 || (m->cfg.intermediate_size % 64 == 0 && m->cfg.hidden_size % 64 == 0),
               "set_precision: the bf16-resident mode needs hidden_size and intermediate_size %% 64 == 0 (got %d, %d): every "
               "forward and input-gradient product runs on 64-deep k-tiles", m->cfg.hidden_size, m->cfg.intermediate_size);
  m->precision = precision;
  return 0;
}

extern "C" int uniter_prof_enable(uniter_model_t* m, int kind) {
  UCHECK_ARG(m && kind >= -1 && kind < UNITER_K_COUNT, "prof_enable: bad argument");
  m->prof_kind = kind;
  m->prof_used = 0;
  const size_t want = kind < 0 ? 65536 : 8192;       // event pairs x 2: ~220 timed launches per step when every kind is on
  if (kind && m->prof_ev.size() < want) {
    const size_t have = m->prof_ev.size();
    m->prof_ev.resize(want);
    m->prof_tag.resize(want / 2);
    for (size_t i = have; i < want; ++i) UCHECK_HIP(hipEventCreate(&m->prof_ev[i]));
  }
  return 0;
}

extern "C" int uniter_prof_enable_stamps(uniter_model_t* m, int on, void* stream) {
  UCHECK_ARG(m, "prof_enable_stamps: null model");
  const size_t cap = 1 << 14;                        // launches; 16 KB each
  const size_t bytes = cap * 2 * STAMP_WGS * sizeof(unsigned long long);
  if (on && !m->stamp_buf) {
    UCHECK_HIP(hipMalloc((void**)&m->stamp_buf, bytes));
    m->stamp_cap = cap;
    m->stamp_tag.resize(cap);
  }
  if (on) {
    UCHECK_HIP(hipMemset(m->stamp_buf, 0, bytes));   // 0 = workgroup did not run (the clock never reads 0)
    m->stamp_used = 0;
  } else if (m->stamp_buf) {
    (void)hipFree(m->stamp_buf);
    m->stamp_buf = nullptr; m->stamp_cap = m->stamp_used = 0;
  }
  (void)stream;
  g_uniter_stamp_slot = nullptr;
  return 0;
}

extern "C" int uniter_prof_collect_stamps(uniter_model_t* m, int* n_launches, double* total_ms, int n_kinds) {
  UCHECK_ARG(m && n_launches && total_ms && n_kinds >= UNITER_K_COUNT, "prof_collect_stamps: need UNITER_K_COUNT slots");
  for (int k = 0; k < n_kinds; ++k) { n_launches[k] = 0; total_ms[k] = 0.0; }
  if (!m->stamp_buf || m->stamp_used == 0) return 0;
  UCHECK_HIP(hipDeviceSynchronize());
  const size_t per = (size_t)2 * STAMP_WGS;
  std::vector<unsigned long long> h(m->stamp_used * per);
  UCHECK_HIP(hipMemcpy(h.data(), m->stamp_buf, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  for (size_t i = 0; i < m->stamp_used; ++i) {
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int w = 0; w < STAMP_WGS; ++w) {
      const unsigned long long a = h[i * per + w], b = h[i * per + STAMP_WGS + w];
      if (a && a < t0) t0 = a;
      if (b > t1) t1 = b;
    }
    if (t0 == ~0ull || t1 < t0) continue;                  // the scope launched no stamped kernel
    n_launches[m->stamp_tag[i]] += 1;
    total_ms[m->stamp_tag[i]] += (double)(t1 - t0) * 1e-5;  // 100 MHz ticks -> ms
  }
  return 0;
}

// every stamped launch in launch order: kind and [first workgroup's start, last workgroup's end] in microseconds since the
// first stamped launch's start (the GPU's 100-MHz clock): where on the step's time axis each GEMM sits, with nothing added to
// the streams -- e.g. how long the chip takes from the last backward GEMM of a step to the first forward GEMM of the next
extern "C" int uniter_prof_stamp_spans(uniter_model_t* m, int* kinds, double* start_us, double* end_us, int cap, int* n) {
  UCHECK_ARG(m && kinds && start_us && end_us && n && cap >= 0, "prof_stamp_spans: null pointer");
  *n = 0;
  if (!m->stamp_buf || m->stamp_used == 0) return 0;
  UCHECK_HIP(hipDeviceSynchronize());
  const size_t per = (size_t)2 * STAMP_WGS;
  std::vector<unsigned long long> h(m->stamp_used * per);
  UCHECK_HIP(hipMemcpy(h.data(), m->stamp_buf, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  unsigned long long base = 0;
  for (size_t i = 0; i < m->stamp_used && *n < cap; ++i) {
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int w = 0; w < STAMP_WGS; ++w) {
      const unsigned long long a = h[i * per + w], b = h[i * per + STAMP_WGS + w];
      if (a && a < t0) t0 = a;
      if (b > t1) t1 = b;
    }
    if (t0 == ~0ull || t1 < t0) continue;
    if (*n == 0) base = t0;
    kinds[*n] = m->stamp_tag[i];
    start_us[*n] = ((double)t0 - (double)base) * 1e-2;
    end_us[*n] = ((double)t1 - (double)base) * 1e-2;
    ++*n;
  }
  return 0;
}

// milliseconds during which at least one stamped launch of a kind in `kind_mask` (bit k = UNITER_K_* k) was running:
// the union of the launches' [first start, last end] intervals -- what two streams that share the chip took TOGETHER
extern "C" int uniter_prof_stamps_union(uniter_model_t* m, unsigned kind_mask, double* union_ms) {
  UCHECK_ARG(m && union_ms, "prof_stamps_union: null pointer");
  *union_ms = 0.0;
  if (!m->stamp_buf || m->stamp_used == 0) return 0;
  UCHECK_HIP(hipDeviceSynchronize());
  const size_t per = (size_t)2 * STAMP_WGS;
  std::vector<unsigned long long> h(m->stamp_used * per);
  UCHECK_HIP(hipMemcpy(h.data(), m->stamp_buf, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  std::vector<std::pair<unsigned long long, unsigned long long>> iv;
  for (size_t i = 0; i < m->stamp_used; ++i) {
    if (!((kind_mask >> m->stamp_tag[i]) & 1u)) continue;
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int w = 0; w < STAMP_WGS; ++w) {
      const unsigned long long a = h[i * per + w], b = h[i * per + STAMP_WGS + w];
      if (a && a < t0) t0 = a;
      if (b > t1) t1 = b;
    }
    if (t0 != ~0ull && t1 >= t0) iv.emplace_back(t0, t1);
  }
  std::sort(iv.begin(), iv.end());
  unsigned long long tot = 0, cs = 0, ce = 0;
  bool open = false;
  for (auto& p : iv) {
    if (!open) { cs = p.first; ce = p.second; open = true; }
    else if (p.first > ce) { tot += ce - cs; cs = p.first; ce = p.second; }
    else if (p.second > ce) ce = p.second;
  }
  if (open) tot += ce - cs;
  *union_ms = (double)tot * 1e-5;
  return 0;
}

extern "C" int uniter_prof_collect_kinds(uniter_model_t* m, int* n_launches, double* total_ms, int n_kinds) {
  UCHECK_ARG(m && n_launches && total_ms && n_kinds >= UNITER_K_COUNT, "prof_collect_kinds: need UNITER_K_COUNT slots");
  for (int k = 0; k < n_kinds; ++k) { n_launches[k] = 0; total_ms[k] = 0.0; }
  for (size_t i = 0; i + 1 < m->prof_used; i += 2) {
    UCHECK_HIP(hipEventSynchronize(m->prof_ev[i + 1]));
    float ms = 0.f;
    UCHECK_HIP(hipEventElapsedTime(&ms, m->prof_ev[i], m->prof_ev[i + 1]));
    const int k = m->prof_tag[i / 2];
    n_launches[k] += 1;
    total_ms[k] += ms;
  }
  m->prof_used = 0;
  return 0;
}

extern "C" int uniter_prof_collect(uniter_model_t* m, int* n_launches, double* total_ms) {
  UCHECK_ARG(m && n_launches && total_ms, "prof_collect: null pointer");
  double tot = 0.0;
  int n = 0;
  for (size_t i = 0; i + 1 < m->prof_used; i += 2) {
    UCHECK_HIP(hipEventSynchronize(m->prof_ev[i + 1]));
    float ms = 0.f;
    UCHECK_HIP(hipEventElapsedTime(&ms, m->prof_ev[i], m->prof_ev[i + 1]));
    tot += ms;
    ++n;
  }
  *n_launches = n; *total_ms = tot;
  m->prof_used = 0;
  return 0;
}
