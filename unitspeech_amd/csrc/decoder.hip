// Host side of libunitspeech_hip.so: weight store, workspace plan, U-Net schedule and the reverse-diffusion loop
// behind the C ABI of include/unitspeech_hip.h.  Mirrors `GradLogPEstimator2d.forward` (unitspeech/unitspeech.py:164-201)
// and `UnitSpeech.reverse_diffusion` (:333-374); the arithmetic lives in conv_igemm.hip / ops.hip / attn.hip.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/unitspeech_hip.h"
#include "kernels.h"
#include "pack_f16.h"

using namespace us;

namespace {

thread_local std::string g_last_error;

struct DevBuf {
  float* p = nullptr;
  size_t n = 0;
};

enum class Kind { RAW, CONV_OIHW, CONVT_IOHW };

constexpr int kWtotalMaxC = 256;

struct Slot {
  std::string key;
  std::vector<int64_t> shape;
  Kind kind = Kind::RAW;
  int bk = 0;            // packing chunk for conv kinds
  DevBuf buf;            // RAW: reference layout; conv kinds: packed layout
  DevBuf dg;             // conv kinds (and to_out): repack for the data-gradient GEMM [tap][Cout/bk_dg][Cin][bk_dg]
  int bk_dg = 0;
  DevBuf wino;           // 3x3 stride-1 convs of the Winograd levels: U = G g G^T, [16][Cin/bk][Cout][bk]
  bool direct_f16 = false;  // buf (direct-form pack of a conv that does not run in the Winograd domain) holds the f16x3 form
  bool wino_f16 = false;    // wino holds the f16x3 form (two interleaved fp16 planes per value) instead of fp32
  bool wino_dg_f16 = false; // ... and so does wino_dg
  bool dg_f16 = false;      // dg holds the f16x3 form (data-gradient GEMM of a direct convolution; K = output channels in 32-chunks)
  int level = -1;        // U-Net level of a ResnetBlock conv
  DevBuf wino_dg;        // ... and of the rotated, channel-swapped filter for the data gradient, [16][Cout/bk_dg][Cin][bk_dg]
  bool want_wino = false;
  // inference-only second Winograd form with 4-wide tiles (wino4.hip): U of F(4x4,3x3) / F(2x4,3x3), [F][Cin/32][Cout][32] two-plane fp16.
  // Packed at load unless the handle is in training mode (us_decoder_set_training: an optimiser step would otherwise re-pack 36 + 16 + 16
  // matrices per convolution); `wino4_valid` says whether it matches the loaded weights -- a stale one is never used (F(2x2) serves)
  DevBuf wino4;
  int wino4_form = 0;
  bool wino4_valid = false;
  bool dg_as_1x1 = false;  // RAW [Cout][Cin][1][1] tensor that also needs a dgrad pack (attention to_out)
  bool is_qkv = false;     // attention to_qkv: a second forward pack with the rows in qkv_src_row() order (ConvArgs::attn_part_ctx)
  DevBuf qkv_rows;         // ... in the same operand form as `buf` (f16x3 planes or fp32)
  DevBuf q_raw;            // ... and W_q = rows 0..127 in the reference layout [128][C] (launch_attn_wtotal)
  float* grad = nullptr; // caller-owned gradient buffer (reference layout) for the current backward call
  bool grad_unscaled = false;   // ... already multiplied by the inverse loss scale by the pass that wrote it (conv_wgrad's unpack)
  bool loaded = false;
};

struct ConvW {
  Slot* w = nullptr;
  Slot* b = nullptr;
  int cin = 0, cout = 0, kh = 1, kw = 1;
};
struct ResnetW {
  int cin, cout, level;
  bool first = false;      // 2-channel input layer (direct kernel, raw weights)
  Slot *mlp_w, *mlp_b;
  ConvW c1, c2, res;
  Slot *g1, *b1, *g2, *b2;
  bool has_res = false;
  int index = 0;           // position in execution order (tproj / stats slots)
};
struct AttnW {
  int dim, level;
  Slot* g;
  ConvW qkv;
  Slot *out_w, *out_b;
};
struct ResampleW {
  ConvW conv;
  int dim, level;
};

int pick_bk(int cin) { return (cin % 32 == 0) ? 32 : 16; }
constexpr const char* kWino4Default = "0,44,44,24";      // US_WINO4 (us_decoder::wino4_level_form)

}  // namespace

namespace us {
void set_last_error(const char* msg) { g_last_error = msg ? msg : ""; }
}

struct us_decoder {
  us_config cfg{};
  int device = 0;
  std::vector<std::unique_ptr<Slot>> slots;
  std::map<std::string, Slot*> by_key;
  std::vector<int> C;   // channels per level
  // topology
  struct Down { ResnetW r1, r2; AttnW a; bool has_ds; ResampleW ds; };
  struct Up { ResnetW r1, r2; AttnW a; ResampleW us; };
  std::vector<Down> downs;
  ResnetW mid1, mid2;
  AttnW mid_attn;
  std::vector<Up> ups;
  ConvW final_conv3;
  Slot *final_g, *final_b, *final_w1, *final_b1;
  Slot *text_uncon, *spk_uncon, *mlp0_w, *mlp0_b, *mlp2_w, *mlp2_b;
  int n_resnets = 0;
  bool f16x3 = true;         // US_F16X3=0: every Winograd GEMM on the fp32 matrix instruction.  Default: those with 32-divisible
  int f16x3_min_level = 0;   // channel counts at levels >= US_F16X3_MIN_LEVEL run as three fp16 MFMA products of split operands
  bool f16x3_dgrad = true;   // US_F16X3_DGRAD=0: data gradients of the direct convolutions stay on fp32 MFMA
  bool f16x3_direct = true;  // US_F16X3_DIRECT=0: direct convolutions (1x1, stride 2, transposed, non-Winograd 3x3) stay on fp32 MFMA
  long long wino_fuse_min_wgs = 400;   // US_WINO_FUSE_MIN_WGS: fused output transform when the launch keeps this many workgroups
  long long wino_fuse_min_wgs_small = 200;       // US_WINO_FUSE_MIN_WGS_SMALL: ... for matrices of at most
  long long wino_fuse_small_kn = 512 * 256;      // US_WINO_FUSE_SMALL_KN elements per frequency
  bool wino_fuse_gn = true;  // US_WINO_FUSE_GN=0: block1's gn_apply as its own pass
  bool presplit = true;      // US_PRESPLIT=0: block1's GroupNorm output stays fp32 for the direct block2 convolution (split in the kernel)
  bool attn_fuse = true;     // US_ATTN_FUSE=0: to_qkv writes q | k | v and attn_ctx_partial_kernel re-reads k, v (the training path's form)
  bool xcd_z = true;         // US_XCD_Z=0: the Winograd-domain GEMMs dealt to the XCDs by tile only, not by whole frequencies (A/B)
  // parameter-gradient chains of a backward pass on side streams (train_host.inc, BwdCtx); US_WGRAD_STREAM=0: everything on the caller's
  // (measured, fine-tune iteration / pre-training step: 1 side stream 10.29 ms / 61.5 ms, 2: 10.43 / 62.5, 3: 11.7 / 62.9)
  int wgrad_side_streams = 1;
  std::vector<hipStream_t> wgrad_streams;
  std::vector<hipEvent_t> wgrad_events;
  // attention with q folded away, out = x + g (W_total x + b_o) with W_total = W_out blockdiag(ctx^T) W_q (attention()): bit l = U-Net level l
  // (C <= kWtotalMaxC only: the projection's K grows from 128 to C); US_ATTN_WTOTAL
  // (measured at B = 1: level 0 only +0.9 %; level 0 and the 128-channel up level +0.6 %; levels 0-1 incl. the 256-channel one +0.0 %)
  int attn_wtotal_levels = 0x1;
  int attn_wtotal_max_c = 128;       // US_ATTN_WTOTAL_MAXC (<= kWtotalMaxC)
  int attn_chunk_rows = 64;          // US_ATTN_CHUNK: rows per chunk of the to_qkv epilogue's online-softmax partials (64 | 128 = its row tile)
  bool fuse_final = true;    // US_FUSE_FINAL=0: the final Block's GroupNorm + Mish as its own launch before the 1x1 projection
  bool wino_narrow = false; // US_WINO_NARROW=1: Winograd also for convolutions with cout <= dim below level 0 (add_resnet)
  int wino_max_level = 99;  // US_WINO_MAX_LEVEL (experiment): levels beyond it run direct
  int wino_min_level = 1;   // ResnetBlock 3x3 convs at U-Net levels >= this run as Winograd F(2x2,3x3); US_WINO_MIN_LEVEL, 99 = off.  Level 0
                            // runs direct since the f16x3 kernels: at 80 x T the 4x-expanded V costs more than 2.25x fewer MFMA FLOPs save
  // F(4x4,3x3) / F(2x4,3x3) Winograd in inference, per U-Net level: 0 = F(2x2) (wino.hip), 44, 24 (wino4.hip); US_WINO4="l0,l1,l2,l3"
  int wino4_level_form[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  bool training = false;    // us_decoder_set_training
  bool wino_splitk = false; // US_WINO_SPLITK=1: K-sliced Winograd-domain GEMMs in training passes (wino_conv)
  float* zeros = nullptr;   // zero page read by out-of-image convolution taps
  // f16x3 operand range (kernels.h): one device word that every split ORs into when it meets a value beyond the fp16 range, and a
  // pinned host word us_range_status copies it to.  `exact`: the handle was created with US_CREATE_EXACT_FP32 (no f16x3 anywhere).
  unsigned* range_flag = nullptr;
  unsigned* range_host = nullptr;
  bool exact = false;
  // us_estimator_backward: {2^k, 2^-k} of the gradient entering the pass, and the table of everything it returns (scaled back in one launch)
  float* grad_scale = nullptr;
  CopyEnt* grad_tab_dev = nullptr;
  static constexpr size_t kLinJobs = 64;
  LinJob *lin_tab_dev = nullptr, *lin_tab_bwd_dev = nullptr;   // job tables of the batched time projections (forward / backward)
  // RAW tensors (biases, GroupNorm affine, MLP weights: ~130 small ones) are copied by ONE table-driven launch per weight sync
  // instead of one hipMemcpyAsync each (fine-tuning re-loads every tensor after every optimiser step)
  std::vector<CopyEnt> pending_copies;
  // ... and the f16x3 packs of the convolution weights (forward and data-gradient forms) by ONE table-driven launch (pack_f16.h)
  std::vector<PackJob> pending_packs;
  PackJob* pack_tab_dev = nullptr;
  static constexpr size_t kStageBytes = 64 << 10;      // one staging buffer holds either table
  CopyEnt* copy_tab_dev = nullptr;
  CopyEnt* copy_tab_host[4] = {nullptr, nullptr, nullptr, nullptr};   // pinned staging ring
  static constexpr int kCaptureTabs = 32;
  size_t copy_tab_capture_bytes[kCaptureTabs] = {};
  CopyEnt* copy_tab_capture[kCaptureTabs] = {};    // write-once staging for uploads recorded into a HIP graph (no allocation is legal
  int copy_tab_capture_used = 0;                    // while a stream captures): one per captured weight sync
  hipEvent_t copy_tab_ev[4] = {nullptr, nullptr, nullptr, nullptr};
  int copy_tab_i = 0;
  size_t copy_tab_cap = 0;
  // saved-activation records of us_estimator_forward_train calls that have not been consumed by a backward yet, by tape id
  // (each lives in its caller's workspace; the oldest is dropped beyond kMaxTapes)
  std::map<uint64_t, std::shared_ptr<void>> tapes;
  uint64_t tape_serial = 0;
  static constexpr size_t kMaxTapes = 16;
  std::string err;

  // ---- sampled kernel timing (bench.py roofline leg) ----
  struct ProfRec { hipEvent_t a, b; double flops; int kind; int f16 = 0; };   // kind 0 = conv_igemm launch, 1 = whole evaluation
  bool prof_enabled = false;
  bool prof_active = false;           // true while the sampled evaluation is being enqueued
  std::vector<ProfRec> prof_pending;
  std::vector<hipEvent_t> prof_pool;
  double prof_conv_ms = 0, prof_conv_flops = 0, prof_eval_ms = 0;
  long long prof_conv_launches = 0, prof_evals = 0;
  double prof_f16_ms = 0, prof_f16_flops = 0;      // the f16x3 launches among them (flops: fp32-equivalent, 2 * M * N * K)
  long long prof_f16_launches = 0;
  hipEvent_t prof_event() {
    if (!prof_pool.empty()) { hipEvent_t e = prof_pool.back(); prof_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
  }

  int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    err = buf;
    g_last_error = buf;
    return code;
  }

  Slot* add(const std::string& key, std::vector<int64_t> shape, Kind kind = Kind::RAW, int bk = 0) {
    slots.emplace_back(new Slot());
    Slot* s = slots.back().get();
    s->key = key;
    s->shape = std::move(shape);
    s->kind = kind;
    s->bk = bk;
    by_key[key] = s;
    return s;
  }

  ConvW add_conv(const std::string& p, int cout, int cin, int k, bool bias, bool transposed = false) {
    ConvW c;
    c.cin = cin; c.cout = cout; c.kh = c.kw = k;
    if (transposed)
      c.w = add(p + ".weight", {cin, cout, k, k}, Kind::CONVT_IOHW, pick_bk(cin));
    else
      c.w = add(p + ".weight", {cout, cin, k, k}, Kind::CONV_OIHW, pick_bk(cin));
    c.w->bk_dg = pick_bk(cout);
    c.b = bias ? add(p + ".bias", {cout}) : nullptr;
    return c;
  }

  ResnetW add_resnet(const std::string& p, int cin, int cout, int level) {
    ResnetW r;
    r.cin = cin; r.cout = cout; r.level = level;
    r.first = (cin % 16 != 0);
    const int temb = cfg.dim + cfg.spk_emb_dim;
    r.mlp_w = add(p + ".mlp.1.weight", {cout, temb});
    r.mlp_b = add(p + ".mlp.1.bias", {cout});
    if (r.first) {
      r.c1.cin = cin; r.c1.cout = cout; r.c1.kh = r.c1.kw = 3;
      r.c1.w = add(p + ".block1.block.0.weight", {cout, cin, 3, 3});
      r.c1.b = add(p + ".block1.block.0.bias", {cout});
    } else {
      r.c1 = add_conv(p + ".block1.block.0", cout, cin, 3, true);
    }
    r.g1 = add(p + ".block1.block.1.weight", {cout});
    r.b1 = add(p + ".block1.block.1.bias", {cout});
    r.c2 = add_conv(p + ".block2.block.0", cout, cout, 3, true);
    r.g2 = add(p + ".block2.block.1.weight", {cout});
    r.b2 = add(p + ".block2.block.1.bias", {cout});
    // Winograd F(2x2,3x3) from US_WINO_MIN_LEVEL down -- except where the output is as narrow as level 0 (cout <= dim: the last up
    // level, 512 -> 128 and 128 -> 128 at 40 x T/2).  There the GEMMs have a single column tile, so nothing amortises the 4x-expanded V
    // (503 MB written and read for the 512-channel input at B' = 3) and the direct form is faster: measured 341 -> ~260 us and
    // 95 -> ~75 us per convolution, transform included (US_WINO_NARROW=1 brings the Winograd form back).
    if (level >= wino_min_level && level <= wino_max_level && (cout > cfg.dim || wino_narrow)) {
      if (!r.first) r.c1.w->want_wino = true;
      r.c2.w->want_wino = true;
    }
    if (!r.first) r.c1.w->level = level;
    r.c2.w->level = level;
    r.has_res = cin != cout;
    if (r.has_res) {
      if (r.first) {
        r.res.cin = cin; r.res.cout = cout;
        r.res.w = add(p + ".res_conv.weight", {cout, cin, 1, 1});
        r.res.b = add(p + ".res_conv.bias", {cout});
      } else {
        r.res = add_conv(p + ".res_conv", cout, cin, 1, true);
      }
    }
    r.index = n_resnets++;
    return r;
  }

  AttnW add_attn(const std::string& p, int dim, int level) {
    AttnW a;
    a.dim = dim; a.level = level;
    a.g = add(p + ".fn.g", {1});
    a.qkv = add_conv(p + ".fn.fn.to_qkv", 3 * kHidden, dim, 1, false);
    a.qkv.w->is_qkv = true;
    a.out_w = add(p + ".fn.fn.to_out.weight", {dim, kHidden, 1, 1});
    a.out_w->dg_as_1x1 = true;
    a.out_w->bk_dg = pick_bk(dim);
    a.out_b = add(p + ".fn.fn.to_out.bias", {dim});
    return a;
  }

  // key order == reference state_dict order (downs, ups, mid, final; unitspeech/unitspeech.py:136-162)
  void build() {
    const int L = cfg.n_mults;
    C.resize(L);
    for (int i = 0; i < L; ++i) C[i] = cfg.dim * cfg.dim_mults[i];
    text_uncon = add("text_uncon", {1, cfg.n_feats, 1});
    spk_uncon = add("spk_uncon", {1, 1, cfg.spk_emb_dim});
    mlp0_w = add("estimator.mlp.0.weight", {4 * cfg.dim, cfg.dim});
    mlp0_b = add("estimator.mlp.0.bias", {4 * cfg.dim});
    mlp2_w = add("estimator.mlp.2.weight", {cfg.dim, 4 * cfg.dim});
    mlp2_b = add("estimator.mlp.2.bias", {cfg.dim});
    downs.resize(L);
    for (int l = 0; l < L; ++l) {
      std::string p = "estimator.downs." + std::to_string(l);
      int cin = l == 0 ? 2 : C[l - 1];
      downs[l].r1 = add_resnet(p + ".0", cin, C[l], l);
      downs[l].r2 = add_resnet(p + ".1", C[l], C[l], l);
      downs[l].a = add_attn(p + ".2", C[l], l);
      downs[l].has_ds = l < L - 1;
      if (downs[l].has_ds) {
        downs[l].ds.conv = add_conv(p + ".3.conv", C[l], C[l], 3, true);
        downs[l].ds.dim = C[l];
        downs[l].ds.level = l;
      }
    }
    ups.resize(L - 1);
    for (int u = 0; u < L - 1; ++u) {
      std::string p = "estimator.ups." + std::to_string(u);
      int level = L - 1 - u;
      int cout = C[level - 1];
      ups[u].r1 = add_resnet(p + ".0", 2 * C[level], cout, level);
      ups[u].r2 = add_resnet(p + ".1", cout, cout, level);
      ups[u].a = add_attn(p + ".2", cout, level);
      ups[u].us.conv = add_conv(p + ".3.conv", cout, cout, 4, true, true);
      ups[u].us.dim = cout;
      ups[u].us.level = level;
    }
    mid1 = add_resnet("estimator.mid_block1", C[L - 1], C[L - 1], L - 1);
    mid_attn = add_attn("estimator.mid_attn", C[L - 1], L - 1);
    mid2 = add_resnet("estimator.mid_block2", C[L - 1], C[L - 1], L - 1);
    final_conv3 = add_conv("estimator.final_block.block.0", cfg.dim, cfg.dim, 3, true);
    if (wino_min_level <= 0) final_conv3.w->want_wino = true;
    final_conv3.w->level = 0;
    final_g = add("estimator.final_block.block.1.weight", {cfg.dim});
    final_b = add("estimator.final_block.block.1.bias", {cfg.dim});
    final_w1 = add("estimator.final_conv.weight", {1, cfg.dim, 1, 1});
    final_b1 = add("estimator.final_conv.bias", {1});
  }
};

namespace {

// enqueue the deferred RAW-tensor copies (us_decoder_load_weight) as one launch; called by every entry point that computes
int flush_copies(us_decoder* h, hipStream_t st);

#define US_HIP(h, expr)                                                                              \
  do {                                                                                               \
    hipError_t _e = (expr);                                                                          \
    if (_e != hipSuccess) return (h)->fail(US_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

size_t numel(const std::vector<int64_t>& s) {
  size_t n = 1;
  for (auto v : s) n *= (size_t)v;
  return n;
}

// ---- workspace arena --------------------------------------------------------------------------------
struct Arena {
  char* base;
  size_t off = 0, cap;
  explicit Arena(void* b, size_t c) : base(static_cast<char*>(b)), cap(c) {}
  template <typename T>
  T* alloc(size_t count) {
    off = (off + 255) & ~size_t(255);
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += count * sizeof(T);
    return p;
  }
};

struct Buffers {
  float *in2, *posemb, *mlp_h, *temb, *tproj;
  std::vector<size_t> tproj_off;   // per resnet, in floats
  int tproj_ld = 0;
  double* stats;                   // [n_gn][Bp][8][2][kStatSlots]
  size_t stats_count = 0;
  std::vector<float*> D, P, Q, S1, S2, QKV, CAT;
  float* U0;
  float *part_ctx, *part_m, *part_s, *ctx, *weff, *colM, *colS, *ctx_split;
  float* wtotal = nullptr;         // [Bp][C][C] W_out blockdiag(ctx^T) W_q of the levels whose attention folds q away (attention())
  float *wino_v = nullptr, *wino_m = nullptr;   // Winograd-domain input / product tensors [16][Bp][tiles][C]
  float* splitk = nullptr;         // split-K slab scratch of the conv kernel
  size_t splitk_floats = 0;
  bool tproj_ready = false;        // true when tproj already holds this evaluation's time projections
};

void plan(us_decoder* h, Arena& A, int Bp, int T, Buffers& b, bool train = false) {
  const int L = h->cfg.n_mults, F = h->cfg.n_feats;
  const size_t B = (size_t)Bp;
  b.in2 = A.alloc<float>(B * F * T * 2);
  b.posemb = A.alloc<float>(B * h->cfg.dim);
  b.mlp_h = A.alloc<float>(B * 4 * h->cfg.dim);
  b.temb = A.alloc<float>(B * (h->cfg.dim + h->cfg.spk_emb_dim));
  // per-resnet time projections, one row block per resnet
  b.tproj_off.assign(h->n_resnets, 0);
  size_t toff = 0;
  auto reg = [&](const ResnetW& r) { b.tproj_off[r.index] = toff; toff += B * r.cout; };
  for (auto& d : h->downs) { reg(d.r1); reg(d.r2); }
  reg(h->mid1); reg(h->mid2);
  for (auto& u : h->ups) { reg(u.r1); reg(u.r2); }
  b.tproj = A.alloc<float>(toff);
  b.stats_count = (size_t)(2 * h->n_resnets + 1) * B * kGroups * kStatStride;
  b.stats = A.alloc<double>(b.stats_count);
  b.D.assign(L, nullptr); b.P.assign(L, nullptr); b.Q.assign(L, nullptr); b.S1.assign(L, nullptr);
  b.S2.assign(L, nullptr); b.QKV.assign(L, nullptr); b.CAT.assign(L, nullptr);
  size_t max_n = 0;
  int max_c = 0;
  for (int l = 0; l < L; ++l) {
    size_t n = (size_t)(F >> l) * (T >> l);
    size_t c = h->C[l];
    if (n > max_n) max_n = n;
    if ((int)c > max_c) max_c = (int)c;
    if (l > 0) b.D[l] = A.alloc<float>(B * n * h->C[l - 1]);
    b.P[l] = A.alloc<float>(B * n * c);
    b.Q[l] = A.alloc<float>(B * n * c);
    b.S1[l] = A.alloc<float>(B * n * c);
    b.S2[l] = A.alloc<float>(B * n * c);
    b.QKV[l] = A.alloc<float>(B * n * 3 * kHidden);
    if (l > 0) b.CAT[l] = A.alloc<float>(B * n * 2 * c);
  }
  b.U0 = A.alloc<float>(B * (size_t)F * T * h->C[0]);
  size_t nch = (max_n + 63) / 64;        // 64-row chunks of the fused to_qkv epilogue (the separate kernel's are 128 rows)
  b.part_ctx = A.alloc<float>(B * nch * kHeads * kDimHead * kDimHead);
  b.part_m = A.alloc<float>(B * nch * kHidden);
  b.part_s = A.alloc<float>(B * nch * kHidden);
  b.ctx = A.alloc<float>(B * kHeads * kDimHead * kDimHead);
  b.colM = A.alloc<float>(B * kHidden);
  b.colS = A.alloc<float>(B * kHidden);
  b.ctx_split = A.alloc<float>(attn_merge_scratch_floats(Bp));
  b.splitk_floats = B * ((size_t)4 << 20);  // 4 Mi floats per item: the conv launcher's bound on its split-K slabs
  b.splitk = A.alloc<float>(b.splitk_floats);
  b.weff = A.alloc<float>(B * (size_t)max_c * kHidden);
  if (!train && h->attn_wtotal_levels) b.wtotal = A.alloc<float>(B * (size_t)kWtotalMaxC * kWtotalMaxC);
  // Winograd scratch, sized from the convolutions that use it: V holds the conv's input channels and M its output channels;
  // the data gradient (training plans) swaps the two
  size_t wv = 0, wm = 0;
  auto need = [&](const ConvW& c, int l) {
    if (!c.w || !c.w->want_wino) return;
    const size_t tiles = (size_t)(((F >> l) + 1) / 2) * (((T >> l) + 1) / 2);
    const size_t kin = train ? (size_t)std::max(c.cin, c.cout) : (size_t)c.cin;
    const size_t kout = train ? (size_t)std::max(c.cin, c.cout) : (size_t)c.cout;
    wv = std::max(wv, 16 * B * tiles * kin);
    wm = std::max(wm, 16 * B * tiles * kout);
    if (!train && c.w->wino4.p) {
      int t4h, t4w;
      wino4_tiles(c.w->wino4_form, F >> l, T >> l, &t4h, &t4w);
      const size_t f4 = (size_t)wino4_freqs(c.w->wino4_form) * B * t4h * t4w;
      wv = std::max(wv, f4 * c.cin);
      wm = std::max(wm, f4 * c.cout);
    }
  };
  auto need_r = [&](const ResnetW& r) { need(r.c1, r.level); need(r.c2, r.level); };
  for (auto& d : h->downs) { need_r(d.r1); need_r(d.r2); }
  need_r(h->mid1); need_r(h->mid2);
  for (auto& u : h->ups) { need_r(u.r1); need_r(u.r2); }
  need(h->final_conv3, 0);
  if (wv) { b.wino_v = A.alloc<float>(wv); b.wino_m = A.alloc<float>(wm); }
}

// ---- one estimator evaluation -------------------------------------------------------------------------
struct EvalCtx {
  us_decoder* h;
  hipStream_t s;
  Buffers* b;
  int Bp, T;
  const float* mask;   // [Bm][T]
  int Bm;
  int gn_slot = 0;
  int splitk_by_batch = 0;   // training contexts: see ConvArgs::splitk_by_batch
  bool infer = false;        // an inference evaluation: the 4-wide Winograd forms (wino4.hip) may serve its 3x3 convolutions
};

double* next_stats(EvalCtx& e) {
  double* p = e.b->stats + (size_t)(e.gn_slot++) * e.Bp * kGroups * kStatStride;
  return p;
}

// frame mask of the OUTPUT tensor (at `level_out` resolution) applied by the epilogue
void set_omask(EvalCtx& e, ConvArgs& a, int level_out) {
  a.omask = e.mask;
  a.omask_ld = e.T;
  a.omask_step = 1 << level_out;
  a.omask_bmod = e.Bm;
}

ConvArgs base_args(EvalCtx& e, const ConvW& w, const float* in, int in_ld, int Hin, int Win, float* out, int out_ld, int Hout,
                   int Wout) {
  ConvArgs a;
  memset(&a, 0, sizeof a);
  a.in = in; a.in_ld = in_ld;
  a.wt = w.w->buf.p;
  a.bias = w.b ? w.b->buf.p : nullptr;
  a.out = out; a.out_ld = out_ld;
  a.B = e.Bp; a.Hin = Hin; a.Win = Win; a.Cin = w.cin; a.Hout = Hout; a.Wout = Wout; a.Cout = w.cout;
  a.Hs = Hout; a.Ws = Wout; a.ostep = 1; a.istride = 1;
  a.bk = w.w->bk;
  a.f16 = w.w->direct_f16 ? 2 : 0;      // f16x3 with the fp32 activations split inside the kernel
  a.omask_bmod = 1;
  a.zeros = e.h->zeros;
  a.splitk_ws = e.b->splitk;
  a.splitk_by_batch = e.splitk_by_batch;
  a.splitk_ws_floats = (long long)e.b->splitk_floats;
  return a;
}

// every implicit-GEMM launch goes through here so the sampled evaluation can bracket it with HIP events
hipError_t run_conv(EvalCtx& e, const ConvArgs& a) {
  us_decoder* h = e.h;
  if (!h->prof_active) return launch_conv_igemm(a, e.s);
  us_decoder::ProfRec r;
  r.a = h->prof_event();
  r.b = h->prof_event();
  r.kind = 0;
  r.flops = 2.0 * a.B * (double)a.Hs * a.Ws * a.Cout * (double)a.Cin * a.ntaps * (a.nphase > 1 ? a.nphase : 1);
  r.f16 = a.f16 ? 1 : 0;
  (void)hipEventRecord(r.a, e.s);
  hipError_t err = launch_conv_igemm(a, e.s);
  (void)hipEventRecord(r.b, e.s);
  h->prof_pending.push_back(r);
  return err;
}

// Winograd F(2x2,3x3) convolution (wino.hip): V = B^T d B, 16 GEMMs M_f = V_f U_f on the implicit-GEMM kernel, out = A^T M A
// (+ bias, GroupNorm sums; + add, * frame mask for data gradients).  U: 16 matrices [K/bk][N][bk], K input and N output channels.
struct WinoEpi {
  const float* bias = nullptr;
  double* stats = nullptr;
  const float* add = nullptr; int add_ld = 0;
  bool mask_out = false;
};
// form4 / U4: the 4-wide-tile form of this convolution (Slot::wino4), or 0 / null
hipError_t wino_conv(EvalCtx& e, const float* in, int in_ld, const float* U, int K, int N, int bk, int level, float* out, int out_ld,
                     const WinoEpi& ep, const WinoGnArgs* gn = nullptr, bool f16 = false, int form4 = 0, const float* U4 = nullptr) {
  const int H = e.h->cfg.n_feats >> level, W = e.T >> level;
  Buffers& b = *e.b;
  if (form4 && U4 && e.infer && !ep.add && !ep.mask_out && !(gn && gn->h_out) && (!gn || in_ld == K) &&
      (long long)H * W * in_ld * 4 < (1LL << 31) && (long long)H * W * K < (1LL << 30)) {       // (launch_wino4_input's 32-bit offsets; else F(2x2))
    // F(4x4,3x3) / F(2x4,3x3): input transform (+ block1's GroupNorm), one GEMM over all items per frequency, output transform
    int t4h, t4w;
    wino4_tiles(form4, H, W, &t4h, &t4w);
    const int nf = wino4_freqs(form4);
    hipError_t err4 = launch_wino4_input(form4, in, in_ld, b.wino_v, e.Bp, H, W, K, gn, e.s);
    if (err4 != hipSuccess) return err4;
    ConvArgs a;
    memset(&a, 0, sizeof a);
    a.f16 = 1;
    a.in = b.wino_v; a.in_ld = K;
    a.wt = U4; a.wt_bstride = (long long)N * K;
    a.Hin = t4h; a.Win = t4w; a.Cin = K; a.Cout = N;
    a.Hs = t4h; a.Ws = t4w; a.istride = 1;
    a.bk = 32;
    a.omask_bmod = 1;
    a.zeros = e.h->zeros;
    a.ntaps = 1;
    a.set_tap(0, 0, 0, 0);
    a.out = b.wino_m; a.out_ld = N;
    a.B = nf; a.Hin = a.Hs = a.Hout = e.Bp * t4h; a.Wout = t4w; a.ostep = 1;
    a.xcd_z = e.h->xcd_z && K >= 256 && N >= 256;
    err4 = run_conv(e, a);
    if (err4 != hipSuccess) return err4;
    return launch_wino4_output(form4, b.wino_m, ep.bias, out, out_ld, ep.stats, e.Bp, H, W, N, e.s);
  }
  const int th = (H + 1) / 2, tw = (W + 1) / 2;
  // gn: `in` is block1's raw conv output (ld == K); its GroupNorm + Mish + time embedding are evaluated inside the transform
  // f16: U is in the f16x3 form and V is written the same way (two interleaved fp16 planes per value, same bytes and strides)
  hipError_t err = gn ? (in_ld == K ? launch_gn_wino_input(in, b.wino_v, e.Bp, H, W, K, *gn, e.s, f16) : hipErrorInvalidValue)
                      : launch_wino_input(in, in_ld, b.wino_v, e.Bp, H, W, K, e.s, f16);
  if (err != hipSuccess) return err;
  ConvArgs a;
  memset(&a, 0, sizeof a);
  a.f16 = f16 ? 1 : 0;
  a.in = b.wino_v; a.in_ld = K;
  a.wt = U; a.wt_bstride = (long long)N * K;
  a.Hin = th; a.Win = tw; a.Cin = K; a.Cout = N;
  a.Hs = th; a.Ws = tw; a.istride = 1;
  a.bk = bk;
  a.omask_bmod = 1;
  a.zeros = e.h->zeros;
  a.ntaps = 1;
  a.set_tap(0, 0, 0, 0);
  // Output transform inside the GEMM kernel (one workgroup walks the 16 frequencies of its tile) when that still leaves
  // enough workgroups for the chip; otherwise 16x more, shorter workgroups and a separate transform pass.  Both forms add in
  // the same order, so the choice (which depends on the batch) does not change a single bit of the result.
  const long long fused_wgs = (long long)((th * tw + 63) / 64) * ((N + 127) / 128) * e.Bp;
  // A fused workgroup streams its column tile's weights for all 16 frequencies (16 * K * 128 values) for only 64 tiles of rows, four
  // times the weight traffic of the separate form's 256-row tiles: with small matrices (K * N <= 512 * 256) it already pays at about
  // one workgroup per CU, with the 1024-wide ones only when the launch is several waves deep.
  const long long min_wgs = (long long)K * N <= e.h->wino_fuse_small_kn ? e.h->wino_fuse_min_wgs_small : e.h->wino_fuse_min_wgs;
  if (fused_wgs >= min_wgs) {
    a.wino_out = 1;
    a.B = e.Bp; a.Hout = H; a.Wout = W; a.ostep = 2;
    a.out = out; a.out_ld = out_ld;
    a.bias = ep.bias;
    a.stats = ep.stats;
    a.add = ep.add; a.add_ld = ep.add_ld;
    if (ep.mask_out) set_omask(e, a, level);
    us_decoder* h = e.h;
    if (!h->prof_active) return launch_conv_igemm(a, e.s);
    us_decoder::ProfRec r;
    r.a = h->prof_event(); r.b = h->prof_event(); r.kind = 0;
    r.flops = 2.0 * 16 * e.Bp * (double)th * tw * N * (double)K;
    r.f16 = a.f16 ? 1 : 0;
    (void)hipEventRecord(r.a, e.s);
    err = launch_conv_igemm(a, e.s);
    (void)hipEventRecord(r.b, e.s);
    h->prof_pending.push_back(r);
    return err;
  }
  // separate form: the items of one frequency are contiguous in V and in M, and a 1x1 tap never leaves its row, so a frequency's
  // GEMM runs over all Bp * th * tw rows at once (no partial tile per item: a level-3 item has only 320 rows)
  a.out = b.wino_m; a.out_ld = N;
  a.B = 16; a.Hin = a.Hs = a.Hout = e.Bp * th; a.Wout = tw; a.ostep = 1;
  // whole frequencies per XCD pay where a frequency's workgroups share operand tiles (nt > 1 column tiles re-reading A, K long enough
  // for the shared reads to matter): with 128 input or output channels the same placement measured 2-20 % SLOWER
  a.xcd_z = e.h->xcd_z && K >= 256 && N >= 256;
  // training passes (one crop: a frequency's GEMM is a single row tile): K may be sliced, the output transform sums the slabs (ConvArgs::splitk_raw)
  int ks = 1;
  // (measured NEGATIVE at one crop: fine-tune iteration 8.57 -> 9.05 ms -- the chains these launches sit on are paced by launch latency, and
  // 4-6x the workgroups take the chip from the weight-gradient stream beside them -- so it is off unless US_WINO_SPLITK=1)
  if (e.splitk_by_batch && e.h->wino_splitk && f16) {
    a.splitk_raw = 1;
    a.splitk_ws = b.splitk; a.splitk_ws_floats = (long long)b.splitk_floats;
    a.ksplit_out = &ks;
  }
  err = run_conv(e, a);
  if (err != hipSuccess) return err;
  WinoOutExtra x{};
  x.add = ep.add; x.add_ld = ep.add_ld;
  if (ep.mask_out) { x.mask = e.mask; x.mask_ld = e.T; x.mask_step = 1 << level; x.mask_bmod = e.Bm; }
  if (ks > 1) return launch_wino_output(b.splitk, ep.bias, out, out_ld, ep.stats, e.Bp, H, W, N, e.s, &x, ks, 16LL * e.Bp * th * tw * N);
  return launch_wino_output(b.wino_m, ep.bias, out, out_ld, ep.stats, e.Bp, H, W, N, e.s, &x);
}

hipError_t conv3x3_wino(EvalCtx& e, const ConvW& w, const float* in, int in_ld, int level, float* out, int out_ld, double* stats,
                        const WinoGnArgs* gn = nullptr) {
  WinoEpi ep;
  ep.bias = w.b ? w.b->buf.p : nullptr;
  ep.stats = stats;
  const bool use4 = w.w->wino4.p && w.w->wino4_valid && w.w->wino_f16;
  return wino_conv(e, in, in_ld, w.w->wino.p, w.cin, w.cout, w.w->bk, level, out, out_ld, ep, gn, w.w->wino_f16, use4 ? w.w->wino4_form : 0,
                   use4 ? w.w->wino4.p : nullptr);
}

// in_split: `in` is the two-plane fp16 form written by gn_apply(out_split) (direct f16x3 convolutions only: direct_presplit())
hipError_t conv3x3(EvalCtx& e, const ConvW& w, const float* in, int in_ld, int level, float* out, int out_ld, double* stats,
                   bool in_split = false) {
  if (w.w->wino.p && e.b->wino_v) return in_split ? hipErrorInvalidValue : conv3x3_wino(e, w, in, in_ld, level, out, out_ld, stats);
  const int H = e.h->cfg.n_feats >> level, W = e.T >> level;
  ConvArgs a = base_args(e, w, in, in_ld, H, W, out, out_ld, H, W);
  if (in_split) {
    if (a.f16 != 2) return hipErrorInvalidValue;
    a.f16 = 1;
    a.direct_presplit = 1;
  }
  a.ntaps = 9;
  for (int ky = 0; ky < 3; ++ky)
    for (int kx = 0; kx < 3; ++kx) a.set_tap(ky * 3 + kx, ky - 1, kx - 1, ky * 3 + kx);
  a.stats = stats;
  return run_conv(e, a);
}

// in_split: `in` holds the two-plane fp16 form (its producer stored it with out_split; f16x3 convolutions only); out_split: store it so
hipError_t conv1x1(EvalCtx& e, const ConvW& w, const float* in, int in_ld, int level, bool mask_out, float* out, int out_ld,
                   const float* add, int add_ld, const float* alpha, const float* wt_override, long long wt_bstride,
                   const float* bias_override, bool in_split = false, bool out_split = false, float* out2 = nullptr, int out2_ld = 0) {
  const int H = e.h->cfg.n_feats >> level, W = e.T >> level;
  ConvArgs a = base_args(e, w, in, in_ld, H, W, out, out_ld, H, W);
  a.out2 = out2; a.out2_ld = out2_ld;
  if (in_split) {
    if (a.f16 != 2) return hipErrorInvalidValue;
    a.f16 = 1;
    a.direct_presplit = 1;
  }
  a.out_split = out_split ? 1 : 0;
  a.ntaps = 1;
  a.set_tap(0, 0, 0, 0);
  if (mask_out) set_omask(e, a, level);
  a.add = add; a.add_ld = add_ld; a.alpha = alpha;
  if (wt_override) { a.wt = wt_override; a.wt_bstride = wt_bstride; a.bias = bias_override; }
  return run_conv(e, a);
}

hipError_t conv_down(EvalCtx& e, const ConvW& w, const float* in, int in_ld, int level, float* out, int out_ld, bool in_split = false) {
  const int H = e.h->cfg.n_feats >> level, W = e.T >> level;
  ConvArgs a = base_args(e, w, in, in_ld, H, W, out, out_ld, H / 2, W / 2);
  if (in_split) {
    if (a.f16 != 2) return hipErrorInvalidValue;
    a.f16 = 1;
    a.direct_presplit = 1;
  }
  a.ntaps = 9;
  a.istride = 2;
  for (int ky = 0; ky < 3; ++ky)
    for (int kx = 0; kx < 3; ++kx) a.set_tap(ky * 3 + kx, ky - 1, kx - 1, ky * 3 + kx);
  set_omask(e, a, level + 1);     // consumers of the downsampled tensor mask it (next ResnetBlock, :54,:74)
  return run_conv(e, a);
}

// ConvTranspose2d(k=4, s=2, p=1): output row oy = 2*iy - 1 + ky.  Output phase py = oy & 1 receives exactly two
// kernel rows: py=0 -> (ky=1, iy=m), (ky=3, iy=m-1); py=1 -> (ky=2, iy=m), (ky=0, iy=m+1)   (same along x).
hipError_t conv_up(EvalCtx& e, const ConvW& w, const float* in, int in_ld, int level, float* out, int out_ld, bool in_split = false,
                   bool out_split = false) {
  const int H = e.h->cfg.n_feats >> level, W = e.T >> level;
  static const int KY[2][2] = {{1, 3}, {2, 0}};
  static const int DY[2][2] = {{0, -1}, {0, 1}};
  // all four output phases in one launch (ConvArgs::nphase)
  ConvArgs a = base_args(e, w, in, in_ld, H, W, out, out_ld, 2 * H, 2 * W);
  if (in_split) {
    if (a.f16 != 2) return hipErrorInvalidValue;
    a.f16 = 1;
    a.direct_presplit = 1;
  }
  a.out_split = out_split ? 1 : 0;
  a.Hs = H; a.Ws = W; a.ostep = 2;
  a.ntaps = 4;
  a.nphase = 4;
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px)
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) a.set_phase_tap(py * 2 + px, i * 2 + j, DY[py][i], DY[px][j], KY[py][i] * 4 + KY[px][j]);
  set_omask(e, a, level - 1);   // the upsampled tensor is only consumed masked (next level's ResnetBlock / final Block)
  return run_conv(e, a);
}

hipError_t gn_apply(EvalCtx& e, const float* y, int level, int C, const double* stats, const Slot* g, const Slot* bta,
                    const float* temb, const float* res, int res_ld, bool res_masked, bool post_mask, float* out, int out_ld,
                    bool out_split = false, float* split_copy = nullptr) {
  GnApplyArgs a;
  memset(&a, 0, sizeof a);
  a.y = y; a.y_ld = C;
  a.stats = stats;
  a.gamma = g->buf.p; a.beta = bta->buf.p;
  a.mask = e.mask; a.mask_ld = e.T; a.mask_step = 1 << level; a.mask_bmod = e.Bm;
  a.temb = temb; a.temb_ld = C;
  a.res = res; a.res_ld = res_ld; a.res_masked = res_masked ? 1 : 0;
  a.post_mask = post_mask ? 1 : 0;
  a.out_split = out_split ? 1 : 0;
  a.out = out; a.out_ld = out_ld;
  a.out2 = split_copy; a.out2_ld = C;
  a.B = e.Bp; a.H = e.h->cfg.n_feats >> level; a.W = e.T >> level; a.C = C;
  return launch_gn_apply(a, e.s);
}

#define CK(expr)                          \
  do {                                    \
    hipError_t _e = (expr);               \
    if (_e != hipSuccess) return _e;      \
  } while (0)

// block1's output has ONE consumer, block2's convolution: where that runs as a direct f16x3 convolution (level 0), gn_apply writes the
// two-plane fp16 form and the convolution skips its in-kernel split (-15 % on the level-0 3x3 in tools/conv_bench).  `tmp`: a buffer
// of the activation's size other than the GroupNorm's input (the planes of a channel quad overlap its neighbour's fp32 input).
inline bool direct_presplit(EvalCtx& e, const ConvW& w, int C, int tmp_ld) {
  return e.h->presplit && !(w.w->wino.p && e.b->wino_v) && w.w->direct_f16 && C % 8 == 0 && tmp_ld == C;
}

// ResnetBlock (unitspeech/unitspeech.py:58-75).  `in` must already be masked.  mask_out: the result is only consumed
// through `x * mask` (next ResnetBlock / concat), so the mask is applied to what is stored; false when the consumer
// is the attention, which reads the raw tensor (padded frames included, :91).
// in_split: `in` is in the two-plane fp16 form (only legal when block1's convolution and res_conv are direct f16x3 convolutions: resnet_takes_split())
// split_out: the block also stores its result in the two-plane form there (ld = cout), for the next block's `in_copy`;
// in_copy: the input once more in the two-plane form (ld = cin): block1's direct f16x3 convolution reads it without the in-kernel split
// while the identity residual keeps reading the fp32 `in` (next_takes_split_copy()).
hipError_t resnet(EvalCtx& e, const ResnetW& r, const float* in, int in_ld, float* out, int out_ld, bool mask_out, bool in_split = false,
                  float* split_out = nullptr, const float* in_copy = nullptr) {
  Buffers& b = *e.b;
  const int l = r.level;
  float* S1 = b.S1[l];
  float* S2 = b.S2[l];
  const float* tproj = b.tproj + b.tproj_off[r.index];
  double* st1 = next_stats(e);
  double* st2 = next_stats(e);
  if (in_split && !r.has_res) return hipErrorInvalidValue;      // the identity residual would need the fp32 tensor
  if (in_copy && (in_split || r.has_res)) return hipErrorInvalidValue;
  if (split_out && (out_ld != r.cout || split_out == out)) return hipErrorInvalidValue;
  // with a split copy to write, res_conv runs first into a spare buffer and block2's GroupNorm pass -- the last writer then -- adds it
  // (the same two fp32 operands, (conv + bias) + h, as when the convolution's epilogue adds h) and stores both forms
  float* R = nullptr;
  if (split_out && r.has_res) {
    if (r.cout > 3 * kHidden) return hipErrorInvalidValue;
    R = b.QKV[l];                                                // the level's attention scratch: free while a ResnetBlock runs
    CK(conv1x1(e, r.res, in, in_ld, l, false, R, r.cout, nullptr, 0, nullptr, nullptr, 0, nullptr, in_split));
  }
  if (in_copy) CK(conv3x3(e, r.c1, in_copy, r.c1.cin, l, S1, r.cout, st1, true));
  else CK(conv3x3(e, r.c1, in, in_ld, l, S1, r.cout, st1, in_split));
  if (r.c2.w->wino.p && b.wino_v && e.h->wino_fuse_gn && gn_wino_input_supported(r.cout)) {
    // block1's GroupNorm + Mish + time embedding (pre-masked for block2's `x * mask`, :54) evaluated inside the Winograd input
    // transform of block2's conv: h1 is never written
    WinoGnArgs g{};
    g.stats = st1; g.gamma = r.g1->buf.p; g.beta = r.b1->buf.p; g.temb = tproj;
    g.mask = e.mask; g.mask_ld = e.T; g.mask_step = 1 << l; g.mask_bmod = e.Bm;
    CK(conv3x3_wino(e, r.c2, S1, r.cout, l, S2, r.cout, st2, &g));
  } else {
    // block1 output + time embedding, pre-masked for block2's `x * mask` (:54)
    if (direct_presplit(e, r.c2, r.cout, out_ld) && out != in) {
      CK(gn_apply(e, S1, l, r.cout, st1, r.g1, r.b1, tproj, nullptr, 0, false, true, out, r.cout, true));      // `out` is free until the end
      CK(conv3x3(e, r.c2, out, r.cout, l, S2, r.cout, st2, true));
    } else {
      CK(gn_apply(e, S1, l, r.cout, st1, r.g1, r.b1, tproj, nullptr, 0, false, true, S1, r.cout));
      CK(conv3x3(e, r.c2, S1, r.cout, l, S2, r.cout, st2));
    }
  }
  if (R) {
    CK(gn_apply(e, S2, l, r.cout, st2, r.g2, r.b2, nullptr, R, r.cout, false, mask_out, out, out_ld, false, split_out));
  } else if (r.has_res) {
    CK(gn_apply(e, S2, l, r.cout, st2, r.g2, r.b2, nullptr, nullptr, 0, false, false, out, out_ld));
    CK(conv1x1(e, r.res, in, in_ld, l, mask_out, out, out_ld, out, out_ld, nullptr, nullptr, 0, nullptr, in_split));
  } else {
    CK(gn_apply(e, S2, l, r.cout, st2, r.g2, r.b2, nullptr, in, in_ld, false, mask_out, out, out_ld, false, split_out));
  }
  return hipSuccess;
}

// r2 of a level takes its input (r1's output) a second time in the two-plane form when its block1 convolution is a direct f16x3 one and
// it has no res_conv (the identity residual reads the fp32 copy); the copy lives in r2's own output buffer, which is free until its end
inline bool next_takes_split_copy(EvalCtx& e, const ResnetW& r2) {
  static const bool on = [] { const char* p = getenv("US_SPLIT_COPY"); return !p || atoi(p) != 0; }();
  return on && e.h->presplit && e.h->f16x3 && e.h->f16x3_direct && !r2.has_res && !r2.first && direct_presplit(e, r2.c1, r2.c1.cin, r2.c1.cin) &&
         r2.c1.cin == r2.cout;
}

// a ResnetBlock whose two readers of its input (block1's 3x3, res_conv) are both direct f16x3 convolutions can take that input pre-split
inline bool resnet_takes_split(EvalCtx& e, const ResnetW& r) {
  return e.h->presplit && !r.first && r.has_res && !(r.c1.w->wino.p && e.b->wino_v) && r.c1.w->direct_f16 && r.res.w->direct_f16;
}

// Residual(Rezero(LinearAttention)) (unitspeech/unitspeech.py:78-106)
// out_split: store the result in the two-plane fp16 form (every consumer is a direct f16x3 convolution)
hipError_t attention(EvalCtx& e, const AttnW& at, const float* in, int in_ld, float* out, int out_ld, bool out_split = false) {
  Buffers& b = *e.b;
  const int l = at.level;
  const int n = (e.h->cfg.n_feats >> l) * (e.T >> l);
  const int nch = attn_nchunks(n);
  float* qkv = b.QKV[l];
  int q_ld = 3 * kHidden;
  bool q_split = false;
  int nparts = nch;
  if (e.h->attn_fuse && at.qkv.w->qkv_rows.p && at.qkv.w->q_raw.p && b.wtotal && ((e.h->attn_wtotal_levels >> l) & 1) && at.dim <= e.h->attn_wtotal_max_c &&
      at.dim % 32 == 0) {
    // q folded away: to_qkv computes k | v only, for the n-reduction in its epilogue, and stores nothing; then
    // out = x + g (W_total x + b_o), W_total[b] = W_out blockdiag(ctx[b]^T) W_q: a C x C projection of x per item instead of q = W_q x written
    // ([n][128]), read back and projected with K = 128 (a third less to_qkv work and 2 * n * 128 floats less traffic; worth it where C <= 256)
    const int H = e.h->cfg.n_feats >> l, W = e.T >> l;
    const int nch64 = (n + e.h->attn_chunk_rows - 1) / e.h->attn_chunk_rows;
    ConvArgs a = base_args(e, at.qkv, in, in_ld, H, W, qkv, kHidden, H, W);
    a.attn_rows = e.h->attn_chunk_rows;
    a.Cout = 2 * kHidden;
    a.wt_rows = 3 * kHidden;
    // rows kHidden.. of every K-chunk of the qkv_src_row-ordered pack (per head k_h | v_h)
    a.wt = at.qkv.w->qkv_rows.p + (size_t)kHidden * at.qkv.w->bk;
    a.ntaps = 1;
    a.set_tap(0, 0, 0, 0);
    a.attn_part_ctx = b.part_ctx; a.attn_part_m = b.part_m; a.attn_part_s = b.part_s; a.attn_nchunks = nch64;
    a.attn_q_cols = 0;
    CK(run_conv(e, a));
    const bool f16 = e.h->f16x3 && e.h->f16x3_direct;
    CK(launch_attn_merge(b.part_ctx, b.part_m, b.part_s, e.Bp, nch64, b.ctx, b.colM, b.colS, b.ctx_split, at.out_w->buf.p, nullptr, at.dim,
                         pick_bk(kHidden), f16, e.s));
    const int bkc = pick_bk(at.dim);
    CK(launch_attn_wtotal(b.ctx, at.out_w->buf.p, at.qkv.w->q_raw.p, b.wtotal, e.Bp, at.dim, bkc, f16, e.s));
    ConvW tot;
    tot.cin = at.dim; tot.cout = at.dim;
    Slot tmp;             // only .bk / .buf / .direct_f16 are read by base_args
    tmp.bk = bkc; tmp.buf.p = b.wtotal; tmp.direct_f16 = f16;
    tot.w = &tmp; tot.b = nullptr;
    return conv1x1(e, tot, in, in_ld, l, true, out, out_ld, in, in_ld, at.g->buf.p, b.wtotal, (long long)at.dim * at.dim, at.out_b->buf.p,
                   false, out_split);
  }
  if (e.h->attn_fuse && at.qkv.w->qkv_rows.p) {
    // to_qkv with the n-reduction in its epilogue: only q is written ([n][128]); chunks of 64 rows (ConvArgs::attn_part_ctx)
    const int H = e.h->cfg.n_feats >> l, W = e.T >> l;
    const int nch64 = (n + e.h->attn_chunk_rows - 1) / e.h->attn_chunk_rows;
    q_ld = kHidden;
    ConvArgs a = base_args(e, at.qkv, in, in_ld, H, W, qkv, q_ld, H, W);
    a.attn_rows = e.h->attn_chunk_rows;
    a.wt = at.qkv.w->qkv_rows.p;
    a.ntaps = 1;
    a.set_tap(0, 0, 0, 0);
    a.attn_part_ctx = b.part_ctx; a.attn_part_m = b.part_m; a.attn_part_s = b.part_s; a.attn_nchunks = nch64;
    a.attn_q_cols = kHidden;
    // q has one reader, the folded to_out convolution below: where that runs as f16x3 it takes q pre-split
    static int q_split_env = -1;                     // US_Q_SPLIT=0: q stays fp32 (A/B)
    if (q_split_env < 0) { const char* ev = getenv("US_Q_SPLIT"); q_split_env = ev ? atoi(ev) : 0; }      // measured: -0.4 % (K = 128: not split-bound)
    q_split = q_split_env && e.h->presplit && e.h->f16x3 && e.h->f16x3_direct && a.f16 == 2;
    a.out_split = q_split ? 1 : 0;
    CK(run_conv(e, a));
    nparts = nch64;
  } else {
    CK(conv1x1(e, at.qkv, in, in_ld, l, false, qkv, 3 * kHidden, nullptr, 0, nullptr, nullptr, 0, nullptr));
    CK(launch_attn_ctx_partial(qkv, e.Bp, n, b.part_ctx, b.part_m, b.part_s, nch, e.s));
  }
  const int bk = pick_bk(kHidden);
  const bool f16 = e.h->f16x3 && e.h->f16x3_direct;
  // chunk partials -> ctx (+ the softmax statistics) -> W_eff = W_out blockdiag(ctx^T), two launches
  CK(launch_attn_merge(b.part_ctx, b.part_m, b.part_s, e.Bp, nparts, b.ctx, b.colM, b.colS, b.ctx_split, at.out_w->buf.p, b.weff, at.dim, bk,
                       f16, e.s));
  ConvW eff;
  eff.cin = kHidden; eff.cout = at.dim;
  Slot tmp;             // only .bk / .buf / .direct_f16 are read by base_args
  tmp.bk = bk; tmp.buf.p = b.weff; tmp.direct_f16 = f16;
  eff.w = &tmp; eff.b = nullptr;
  // every consumer of an attention output masks it (Downsample / Upsample input, skip concat, mid blocks)
  return conv1x1(e, eff, qkv, q_ld, l, true, out, out_ld, in, in_ld, at.g->buf.p, b.weff, (long long)at.dim * kHidden,
                 at.out_b->buf.p, q_split, out_split);
}

int upload_bytes(us_decoder* h, const void* src, size_t nbytes, void* dev, hipStream_t st);

// the ResnetBlocks' time projections as LinJobs in execution order (kernels.h); out / gy / gW / gb are filled by the caller
inline int proj_jobs(us_decoder* h, std::vector<LinJob>& jobs, std::vector<const ResnetW*>& blocks) {
  jobs.clear();
  blocks.clear();
  int o0 = 0;
  auto add = [&](const ResnetW& r) {
    LinJob j;
    memset(&j, 0, sizeof j);
    j.W = r.mlp_w->buf.p; j.bias = r.mlp_b->buf.p; j.o0 = o0; j.cout = r.cout; j.out_ld = r.cout;
    jobs.push_back(j);
    blocks.push_back(&r);
    o0 += r.cout;
  };
  for (auto& d : h->downs) { add(d.r1); add(d.r2); }
  add(h->mid1); add(h->mid2);
  for (auto& u : h->ups) { add(u.r1); add(u.r2); }
  return o0;
}

hipError_t time_embedding(EvalCtx& e, const float* t, const float* spk) {
  us_decoder* h = e.h;
  Buffers& b = *e.b;
  const int dim = h->cfg.dim, S = h->cfg.spk_emb_dim, td = dim + S;
  CK(launch_pos_emb(t, b.posemb, e.Bp, dim, h->cfg.pe_scale, e.s));
  CK(launch_linear(b.posemb, dim, h->mlp0_w->buf.p, h->mlp0_b->buf.p, b.mlp_h, 4 * dim, e.Bp, dim, 4 * dim, false, e.s));
  CK(launch_linear(b.mlp_h, 4 * dim, h->mlp2_w->buf.p, h->mlp2_b->buf.p, b.temb, td, e.Bp, 4 * dim, dim, true, e.s));
  CK(launch_copy_rows(spk, S, e.Bp, b.temb + dim, td, e.Bp, S, e.s));
  // the 22 projections mlp(temb) of the ResnetBlocks (unitspeech.py:61,72) share their input: one launch from a job table
  std::vector<LinJob> jobs;
  std::vector<const ResnetW*> blocks;
  const int total = proj_jobs(h, jobs, blocks);
  if (h->lin_tab_dev && jobs.size() <= us_decoder::kLinJobs) {
    for (size_t i = 0; i < jobs.size(); ++i) jobs[i].out = b.tproj + b.tproj_off[blocks[i]->index];
    if (upload_bytes(h, jobs.data(), jobs.size() * sizeof(LinJob), h->lin_tab_dev, e.s) != US_OK) return hipErrorUnknown;
    return launch_linear_multi(h->lin_tab_dev, (int)jobs.size(), total, b.temb, td, e.Bp, td, true, e.s);
  }
  auto proj = [&](const ResnetW& r) {
    return launch_linear(b.temb, td, r.mlp_w->buf.p, r.mlp_b->buf.p, b.tproj + b.tproj_off[r.index], r.cout, e.Bp, td, r.cout, true,
                         e.s);
  };
  for (auto& d : h->downs) { CK(proj(d.r1)); CK(proj(d.r2)); }
  CK(proj(h->mid1)); CK(proj(h->mid2));
  for (auto& u : h->ups) { CK(proj(u.r1)); CK(proj(u.r2)); }
  return hipSuccess;
}

// x: [Bx][F][T]; mu: [Bmu][F][T]; items b' < n_text_uncond use text_uncon instead of mu; spk: [Bp][S]; t: [Bp]
hipError_t estimator_eval_impl(EvalCtx& e, const float* x, int Bx, const float* mu, int Bmu, int n_text_uncond, const float* t,
                               const float* spk, float* out);

// `sample` marks the evaluation whose launches are bracketed with events when profiling is enabled
hipError_t estimator_eval(EvalCtx& e, const float* x, int Bx, const float* mu, int Bmu, int n_text_uncond, const float* t,
                          const float* spk, float* out, bool sample = false) {
  us_decoder* h = e.h;
  if (!(h->prof_enabled && sample)) return estimator_eval_impl(e, x, Bx, mu, Bmu, n_text_uncond, t, spk, out);
  us_decoder::ProfRec r;
  r.a = h->prof_event();
  r.b = h->prof_event();
  r.kind = 1;
  r.flops = 0;
  (void)hipEventRecord(r.a, e.s);
  h->prof_active = true;
  hipError_t err = estimator_eval_impl(e, x, Bx, mu, Bmu, n_text_uncond, t, spk, out);
  h->prof_active = false;
  (void)hipEventRecord(r.b, e.s);
  h->prof_pending.push_back(r);
  return err;
}

hipError_t estimator_eval_impl(EvalCtx& e, const float* x, int Bx, const float* mu, int Bmu, int n_text_uncond, const float* t,
                               const float* spk, float* out) {
  us_decoder* h = e.h;
  Buffers& b = *e.b;
  const int L = h->cfg.n_mults, F = h->cfg.n_feats, T = e.T;
  e.gn_slot = 0;
  CK(hipMemsetAsync(b.stats, 0, b.stats_count * sizeof(double), e.s));
  if (!b.tproj_ready) CK(time_embedding(e, t, spk));
  CK(launch_stack_inputs(x, Bx, mu, Bmu, n_text_uncond, h->text_uncon->buf.p, e.mask, e.Bm, b.in2, e.Bp, F, T, e.s));

  // Tensors that only direct f16x3 convolutions read are stored in the two-plane fp16 form by their producer's epilogue (no split work
  // in the consumers, which then run the 128-row tile):
  //   hid_split[l]  level l's attention output.  Level 0: its one reader is the Downsample (the level-0 skip is never popped,
  //                 unitspeech.py:180,190-192).  Deeper: the Downsample plus the up path's first ResnetBlock (the concat's second half),
  //                 which qualifies where that block runs direct (resnet_takes_split: the narrow last up level); the concat's first
  //                 half, the Upsample output of the level below, is then stored the same way.
  //   up_split[l]   the up path's attention output at level l: its one reader is the Upsample.
  //   fin_split     the last Upsample's output: its one reader is the final Block's convolution.
  std::vector<char> hid_split(L, 0), up_split(L, 0);
  bool fin_split = false;
  if (h->presplit && h->f16x3 && h->f16x3_direct && h->attn_fuse) {
    for (int l = 0; l < L; ++l) {
      auto& d = h->downs[l];
      if (!d.has_ds || !d.ds.conv.w->direct_f16 || h->C[l] % 8 != 0) continue;
      if (l == 0) hid_split[0] = 1;
      else if (l <= L - 2) hid_split[l] = resnet_takes_split(e, h->ups[L - 1 - l].r1) && h->ups[L - 1 - l - 1].us.conv.w->direct_f16 ? 1 : 0;
    }
    for (int u = 0; u < L - 1; ++u) up_split[h->ups[u].r1.level] = h->ups[u].us.conv.w->direct_f16 ? 1 : 0;
    fin_split = L > 1 && !(h->final_conv3.w->wino.p && b.wino_v) && h->final_conv3.w->direct_f16 && h->ups[L - 2].us.conv.w->direct_f16;
  }
  const float* cur = nullptr;
  int cur_ld = 0;
  for (int l = 0; l < L; ++l) {
    auto& d = h->downs[l];
    const int c = h->C[l];
    const bool r2_copy = next_takes_split_copy(e, d.r2);
    if (l == 0) {
      // first ResnetBlock: 2-channel convs on the direct kernel
      const ResnetW& r = d.r1;
      double* st1 = next_stats(e);
      double* st2 = next_stats(e);
      // (the block's 1x1 res_conv output is not stored: its GroupNorm pass below re-evaluates it from the two input channels)
      CK(launch_first_conv(b.in2, r.c1.w->buf.p, r.c1.b->buf.p, r.res.w->buf.p, r.res.b->buf.p, b.S1[0], nullptr, st1, e.Bp, F, T, c, e.s));
      if (r.c2.w->wino.p && b.wino_v && h->wino_fuse_gn && gn_wino_input_supported(c)) {
        WinoGnArgs g{};     // as in resnet(): block1's GroupNorm + Mish + time embedding inside block2's input transform
        g.stats = st1; g.gamma = r.g1->buf.p; g.beta = r.b1->buf.p; g.temb = b.tproj + b.tproj_off[r.index];
        g.mask = e.mask; g.mask_ld = e.T; g.mask_step = 1; g.mask_bmod = e.Bm;
        CK(conv3x3_wino(e, r.c2, b.S1[0], c, 0, b.S2[0], c, st2, &g));
      } else {
        if (direct_presplit(e, r.c2, c, c)) {      // as in resnet(): P[0], this block's output buffer, holds the two-plane form meanwhile
          CK(gn_apply(e, b.S1[0], 0, c, st1, r.g1, r.b1, b.tproj + b.tproj_off[r.index], nullptr, 0, false, true, b.P[0], c, true));
          CK(conv3x3(e, r.c2, b.P[0], c, 0, b.S2[0], c, st2, true));
        } else {
          CK(gn_apply(e, b.S1[0], 0, c, st1, r.g1, r.b1, b.tproj + b.tproj_off[r.index], nullptr, 0, false, true, b.S1[0], c));
          CK(conv3x3(e, r.c2, b.S1[0], c, 0, b.S2[0], c, st2));
        }
      }
      {
        GnApplyArgs ga;
        memset(&ga, 0, sizeof ga);
        ga.y = b.S2[0]; ga.y_ld = c; ga.stats = st2; ga.gamma = r.g2->buf.p; ga.beta = r.b2->buf.p;
        ga.mask = e.mask; ga.mask_ld = e.T; ga.mask_step = 1; ga.mask_bmod = e.Bm;
        ga.res2_in = b.in2; ga.res2_w = r.res.w->buf.p; ga.res2_b = r.res.b->buf.p;
        ga.post_mask = 1;
        ga.out = b.P[0]; ga.out_ld = c;
        if (r2_copy) { ga.out2 = b.Q[0]; ga.out2_ld = c; }
        ga.B = e.Bp; ga.H = F; ga.W = T; ga.C = c;
        CK(launch_gn_apply(ga, e.s));
      }
    } else {
      CK(resnet(e, d.r1, cur, cur_ld, b.P[l], c, true, false, r2_copy ? b.Q[l] : nullptr));
    }
    CK(resnet(e, d.r2, b.P[l], c, b.Q[l], c, false, false, nullptr, r2_copy ? b.Q[l] : nullptr));
    float* hid = l == 0 ? b.P[0] : b.CAT[l] + c;
    int hid_ld = l == 0 ? c : 2 * c;
    CK(attention(e, d.a, b.Q[l], c, hid, hid_ld, hid_split[l]));
    if (d.has_ds) {
      CK(conv_down(e, d.ds.conv, hid, hid_ld, l, b.D[l + 1], c, hid_split[l]));
      cur = b.D[l + 1];
      cur_ld = c;
    } else {
      cur = hid;
      cur_ld = hid_ld;
    }
  }
  const int lm = L - 1, cm = h->C[lm];
  CK(resnet(e, h->mid1, cur, cur_ld, b.P[lm], cm, false));
  CK(attention(e, h->mid_attn, b.P[lm], cm, b.Q[lm], cm));
  float* xcat = L > 1 ? b.CAT[lm] : b.P[lm];
  CK(resnet(e, h->mid2, b.Q[lm], cm, xcat, L > 1 ? 2 * cm : cm, true));
  const float* fin = xcat;
  int fin_ld = cm;
  for (int u = 0; u < L - 1; ++u) {
    auto& up = h->ups[u];
    const int l = up.r1.level, co = up.r1.cout;
    const bool r2_copy = next_takes_split_copy(e, up.r2);
    CK(resnet(e, up.r1, b.CAT[l], 2 * h->C[l], b.P[l], co, true, l <= L - 2 && hid_split[l], r2_copy ? b.Q[l] : nullptr));
    CK(resnet(e, up.r2, b.P[l], co, b.Q[l], co, false, false, nullptr, r2_copy ? b.Q[l] : nullptr));
    CK(attention(e, up.a, b.Q[l], co, b.P[l], co, up_split[l]));
    float* dst = (l - 1 >= 1) ? b.CAT[l - 1] : b.U0;
    int dst_ld = (l - 1 >= 1) ? 2 * h->C[l - 1] : h->C[0];
    CK(conv_up(e, up.us.conv, b.P[l], co, l, dst, dst_ld, up_split[l], (l - 1 >= 1) ? hid_split[l - 1] != 0 : fin_split));
    fin = dst;
    fin_ld = dst_ld;
  }
  // final Block + 1x1 projection (unitspeech/unitspeech.py:198-201)
  double* stf = next_stats(e);
  const int c0 = h->C[0];
  CK(conv3x3(e, h->final_conv3, fin, fin_ld, 0, b.S1[0], c0, stf, fin_split));
  if (h->fuse_final) {      // GroupNorm + Mish + mask inside the 1x1 projection: the normalised tensor is never stored
    CK(launch_final_conv(b.S1[0], c0, h->final_w1->buf.p, h->final_b1->buf.p, e.mask, T, e.Bm, out, e.Bp, F, T, c0, e.s, stf,
                         h->final_g->buf.p, h->final_b->buf.p));
    return hipSuccess;
  }
  CK(gn_apply(e, b.S1[0], 0, c0, stf, h->final_g, h->final_b, nullptr, nullptr, 0, false, false, b.S1[0], c0));
  CK(launch_final_conv(b.S1[0], c0, h->final_w1->buf.p, h->final_b1->buf.p, e.mask, T, e.Bm, out, e.Bp, F, T, c0, e.s));
  return hipSuccess;
}

// ---- time / speaker projections for a block of diffusion steps ------------------------------------------------
// `mlp(time_emb)` of every ResnetBlock depends only on (t_i, spk_emb) (unitspeech.py:165-168, 61, 72), so the sampler
// computes it for kTimeBlock steps at once instead of launching 24 tiny kernels per evaluation.
constexpr int kTimeBlock = 64;

inline size_t sum_res_cout(us_decoder* h) {
  size_t c = 0;
  for (auto& d : h->downs) c += d.r1.cout + d.r2.cout;
  c += h->mid1.cout + h->mid2.cout;
  for (auto& u : h->ups) c += u.r1.cout + u.r2.cout;
  return c;
}
inline size_t time_block_floats(us_decoder* h, int Bp) {
  const size_t rows = (size_t)kTimeBlock * Bp;
  const int dim = h->cfg.dim, td = dim + h->cfg.spk_emb_dim;
  return rows * (sum_res_cout(h) + td + 4 * dim + dim + 1) + 64 * 8;
}
struct TimeBlock {
  float *t_all, *posemb, *mlp_h, *temb, *tproj;   // rows = steps * Bp
};
inline void plan_time_block(us_decoder* h, Arena& A, int Bp, TimeBlock& tb) {
  const size_t rows = (size_t)kTimeBlock * Bp;
  const int dim = h->cfg.dim, td = dim + h->cfg.spk_emb_dim;
  tb.t_all = A.alloc<float>(rows);
  tb.posemb = A.alloc<float>(rows * dim);
  tb.mlp_h = A.alloc<float>(rows * 4 * dim);
  tb.temb = A.alloc<float>(rows * td);
  tb.tproj = A.alloc<float>(rows * sum_res_cout(h));
}
// steps [i0, i0+ns): t values from the host coefficient table; layout of tb.tproj: per resnet r a [ns*Bp][cout_r] block
// starting at ns * tproj_off[r] (tproj_off from plan()): step i's rows sit at + (i-i0)*Bp*cout_r.
inline hipError_t compute_time_block(EvalCtx& e, TimeBlock& tb, const float* coef_host, int i0, int ns, const float* spk_cfg) {
  us_decoder* h = e.h;
  const int dim = h->cfg.dim, S = h->cfg.spk_emb_dim, td = dim + S;
  const int rows = ns * e.Bp;
  for (int i = 0; i < ns; ++i) CK(launch_fill(tb.t_all + (size_t)i * e.Bp, coef_host[(size_t)(i0 + i) * 8 + 6], e.Bp, e.s));
  CK(launch_pos_emb(tb.t_all, tb.posemb, rows, dim, h->cfg.pe_scale, e.s));
  CK(launch_linear(tb.posemb, dim, h->mlp0_w->buf.p, h->mlp0_b->buf.p, tb.mlp_h, 4 * dim, rows, dim, 4 * dim, false, e.s));
  CK(launch_linear(tb.mlp_h, 4 * dim, h->mlp2_w->buf.p, h->mlp2_b->buf.p, tb.temb, td, rows, 4 * dim, dim, true, e.s));
  CK(launch_copy_rows(spk_cfg, S, e.Bp, tb.temb + dim, td, rows, S, e.s));
  auto proj = [&](const ResnetW& r) {
    return launch_linear(tb.temb, td, r.mlp_w->buf.p, r.mlp_b->buf.p, tb.tproj + (size_t)ns * e.b->tproj_off[r.index], r.cout, rows, td,
                         r.cout, true, e.s);
  };
  for (auto& d : h->downs) { CK(proj(d.r1)); CK(proj(d.r2)); }
  CK(proj(h->mid1)); CK(proj(h->mid2));
  for (auto& u : h->ups) { CK(proj(u.r1)); CK(proj(u.r2)); }
  return hipSuccess;
}

#include "train_host.inc"

// ents[0..n) -> dev through the pinned staging ring (or a write-once table of its own while the stream is being captured)
int upload_bytes(us_decoder* h, const void* src, size_t nbytes, void* dev, hipStream_t st);
int upload_table(us_decoder* h, const CopyEnt* ents, size_t n, CopyEnt* dev, hipStream_t st) {
  if (n > h->copy_tab_cap) return h->fail(US_EINVAL, "internal: copy table overflow");
  return upload_bytes(h, ents, n * sizeof(CopyEnt), dev, st);
}
int upload_bytes(us_decoder* h, const void* ents, size_t nbytes, void* dev, hipStream_t st) {
  if (nbytes > us_decoder::kStageBytes) return h->fail(US_EINVAL, "internal: staging table overflow");
  const size_t n = nbytes;
  const int slot = h->copy_tab_i++ & 3;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(st, &cap);
  CopyEnt* host = h->copy_tab_host[slot];
  if (cap != hipStreamCaptureStatusNone) {
    // a captured upload reads its host source at every replay: give it storage of its own that is never rewritten
    // ... and a table with these very bytes recorded by an earlier capture (the job tables of a FineTuneGraph hold the same pointers every
    // time one is built on this handle) serves this one too, so that per-speaker loops that re-capture do not run out of entries
    host = nullptr;
    for (int i = 0; i < h->copy_tab_capture_used && !host; ++i)
      if (h->copy_tab_capture_bytes[i] == n && memcmp(h->copy_tab_capture[i], ents, n) == 0) host = h->copy_tab_capture[i];
    if (!host) {
      if (h->copy_tab_capture_used >= us_decoder::kCaptureTabs)
        return h->fail(US_EINVAL, "more than %d distinct weight syncs / backward passes recorded into HIP graphs with this handle", us_decoder::kCaptureTabs);
      h->copy_tab_capture_bytes[h->copy_tab_capture_used] = n;
      host = h->copy_tab_capture[h->copy_tab_capture_used++];
    }
  } else if (h->copy_tab_ev[slot]) {
    (void)hipEventSynchronize(h->copy_tab_ev[slot]);      // the upload that last used this staging slot has completed
  }
  memcpy(host, ents, n);
  US_HIP(h, hipMemcpyAsync(dev, host, n, hipMemcpyHostToDevice, st));
  if (cap == hipStreamCaptureStatusNone) {
    if (!h->copy_tab_ev[slot]) US_HIP(h, hipEventCreateWithFlags(&h->copy_tab_ev[slot], hipEventDisableTiming));
    US_HIP(h, hipEventRecord(h->copy_tab_ev[slot], st));
  }
  return US_OK;
}

int flush_packs(us_decoder* h, hipStream_t st) {
  const size_t n = h->pending_packs.size();
  if (n == 0) return US_OK;
  int total = 0;
  for (auto& j : h->pending_packs) { j.blk0 = total; j.nblk = pack_job_blocks(j); total += j.nblk; }
  int rc = upload_bytes(h, h->pending_packs.data(), n * sizeof(PackJob), h->pack_tab_dev, st);
  if (rc) return rc;
  RangeScope range_scope(h->range_flag);
  US_HIP(h, launch_pack_table(h->pack_tab_dev, (int)n, total, st));
  h->pending_packs.clear();
  return US_OK;
}

int flush_copies(us_decoder* h, hipStream_t st) {
  int rcp = flush_packs(h, st);
  if (rcp) return rcp;
  const size_t n = h->pending_copies.size();
  if (n == 0) return US_OK;
  int rc = upload_table(h, h->pending_copies.data(), n, h->copy_tab_dev, st);
  if (rc) return rc;
  US_HIP(h, launch_copy_table(h->copy_tab_dev, (int)n, st));
  h->pending_copies.clear();
  return US_OK;
}

int check_ready(us_decoder* h) {
  for (auto& s : h->slots)
    if (!s->loaded) return h->fail(US_EWEIGHTS, "weight '%s' has not been loaded", s->key.c_str());
  if (!h->pending_copies.empty() || !h->pending_packs.empty())
    return h->fail(US_EWEIGHTS, "us_decoder_load_weight calls have not been followed by us_decoder_flush_weights");
  return US_OK;
}

int check_shape(us_decoder* h, int Bp, int T) {
  const int q = 1 << (h->cfg.n_mults - 1);
  if (Bp <= 0 || T <= 0 || T % q != 0 || T % 4 != 0)
    return h->fail(US_EINVAL, "unsupported shape Bp=%d T=%d (T must be a positive multiple of %d)", Bp, T, q > 4 ? q : 4);
  if (h->cfg.n_feats % q != 0) return h->fail(US_EINVAL, "n_feats=%d is not divisible by %d", h->cfg.n_feats, q);
  return US_OK;
}

}  // namespace

// =======================================================================================================
// C ABI
// =======================================================================================================
extern "C" {

int us_decoder_create(us_handle* out, const us_config* cfg) { return us_decoder_create_ex(out, cfg, 0u); }

int us_decoder_create_ex(us_handle* out, const us_config* cfg, unsigned flags) {
  if (!out || !cfg) { g_last_error = "null argument"; return US_EINVAL; }
  if (flags & ~(unsigned)US_CREATE_EXACT_FP32) { g_last_error = "unknown creation flag"; return US_EINVAL; }
  if (cfg->n_mults < 1 || cfg->n_mults > 6 || cfg->dim < 16 || cfg->dim % 16 != 0 || cfg->n_feats <= 0 || cfg->spk_emb_dim <= 0 ||
      cfg->spk_emb_dim % 4 != 0) {
    g_last_error = "unsupported configuration (dim must be a multiple of 16, 1 <= n_mults <= 6)";
    return US_EINVAL;
  }
  for (int i = 0; i < cfg->n_mults; ++i) {
    int c = cfg->dim * cfg->dim_mults[i];
    int cg = c / kGroups;
    if (cfg->dim_mults[i] < 1 || c % 16 != 0 || (cg & (cg - 1)) != 0) {
      g_last_error = "unsupported dim_mults (channels/8 must be a power of two)";
      return US_EINVAL;
    }
  }
  std::unique_ptr<us_decoder> h(new us_decoder());
  h->cfg = *cfg;
  if (hipGetDevice(&h->device) != hipSuccess) { g_last_error = "no HIP device"; return US_EHIP; }
  hipError_t e = conv_igemm_init();
  if (e != hipSuccess) { g_last_error = std::string("conv_igemm_init: ") + hipGetErrorString(e); return US_EHIP; }
  if (const char* wl = getenv("US_WINO_MIN_LEVEL")) h->wino_min_level = atoi(wl);
  if (const char* wl = getenv("US_WINO_MAX_LEVEL")) h->wino_max_level = atoi(wl);
  if (const char* ws = getenv("US_WINO_SPLITK")) h->wino_splitk = atoi(ws) != 0;
  {
    // per-level 4-wide Winograd form of the inference path (wino4.hip), "l0,l1,l2,l3": 0 = F(2x2,3x3), 44 = F(4x4,3x3), 24 = F(2x4,3x3)
    const char* w4 = getenv("US_WINO4");
    std::string spec = w4 ? w4 : kWino4Default;
    size_t pos = 0;
    for (int l = 0; l < 8 && pos <= spec.size(); ++l) {
      const size_t comma = spec.find(',', pos);
      const std::string tok = spec.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
      const int f = tok.empty() ? 0 : atoi(tok.c_str());
      h->wino4_level_form[l] = wino4_form_ok(f) ? f : 0;
      if (comma == std::string::npos) break;
      pos = comma + 1;
    }
    if (flags & US_CREATE_EXACT_FP32)
      for (int l = 0; l < 8; ++l) h->wino4_level_form[l] = 0;
  }
  if (const char* wl = getenv("US_WINO_NARROW")) h->wino_narrow = atoi(wl) != 0;
  if (const char* wf = getenv("US_WINO_FUSE_MIN_WGS")) h->wino_fuse_min_wgs = atoll(wf);
  if (const char* wf = getenv("US_WINO_FUSE_MIN_WGS_SMALL")) h->wino_fuse_min_wgs_small = atoll(wf);
  if (const char* wf = getenv("US_WINO_FUSE_SMALL_KN")) h->wino_fuse_small_kn = atoll(wf);
  if (const char* wf = getenv("US_F16X3")) h->f16x3 = atoi(wf) != 0;
  if (const char* wf = getenv("US_F16X3_MIN_LEVEL")) h->f16x3_min_level = atoi(wf);
  if (const char* wf = getenv("US_F16X3_DIRECT")) h->f16x3_direct = atoi(wf) != 0;
  if (const char* wf = getenv("US_F16X3_DGRAD")) h->f16x3_dgrad = atoi(wf) != 0;
  if (const char* wf = getenv("US_WINO_FUSE_GN")) h->wino_fuse_gn = atoi(wf) != 0;
  if (const char* wf = getenv("US_XCD_Z")) h->xcd_z = atoi(wf) != 0;
  if (const char* wf = getenv("US_PRESPLIT")) h->presplit = atoi(wf) != 0;
  if (const char* wf = getenv("US_ATTN_FUSE")) h->attn_fuse = atoi(wf) != 0;
  if (const char* ff = getenv("US_FUSE_FINAL")) h->fuse_final = atoi(ff) != 0;
  if (const char* aw = getenv("US_ATTN_WTOTAL")) h->attn_wtotal_levels = atoi(aw);
  if (const char* ac = getenv("US_ATTN_CHUNK")) h->attn_chunk_rows = atoi(ac) == 128 ? 128 : 64;
  if (const char* aw = getenv("US_ATTN_WTOTAL_MAXC")) h->attn_wtotal_max_c = atoi(aw) > kWtotalMaxC ? kWtotalMaxC : atoi(aw);
  if (const char* ws = getenv("US_WGRAD_STREAM")) h->wgrad_side_streams = atoi(ws) < 0 ? 0 : (atoi(ws) > Tape::kSideMax ? Tape::kSideMax : atoi(ws));
  if (flags & US_CREATE_EXACT_FP32) h->f16x3 = false;
  h->exact = !h->f16x3;
  h->build();
  if (hipMalloc(reinterpret_cast<void**>(&h->range_flag), 256) != hipSuccess || hipMemset(h->range_flag, 0, 256) != hipSuccess ||
      hipHostMalloc(reinterpret_cast<void**>(&h->range_host), 256) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&h->grad_scale), 256) != hipSuccess) {
    g_last_error = "allocation of the range word failed";
    return US_EHIP;
  }
  *h->range_host = 0u;
  {
    int max_cin = 2 * h->C.back();
    for (int c : h->C) if (2 * c > max_cin) max_cin = 2 * c;
    size_t zb = ((size_t)max_cin + 64) * sizeof(float);
    if (hipMalloc(reinterpret_cast<void**>(&h->zeros), zb) != hipSuccess || hipMemset(h->zeros, 0, zb) != hipSuccess) {
      g_last_error = "hipMalloc failed for the zero page";
      return US_EHIP;
    }
  }
  for (auto& s : h->slots) {
    // a conv that lives in the Winograd domain keeps only its two Winograd packs (forward, data gradient)
    bool ok = true;
    s->buf.n = numel(s->shape);      // element count of the tensor (also sizes gradient / scratch buffers), allocated or not
    if (!s->want_wino) {
      ok = hipMalloc(reinterpret_cast<void**>(&s->buf.p), s->buf.n * sizeof(float)) == hipSuccess;
      if (ok && (s->kind != Kind::RAW || s->dg_as_1x1)) {
        s->dg.n = s->buf.n;
        ok = hipMalloc(reinterpret_cast<void**>(&s->dg.p), s->dg.n * sizeof(float)) == hipSuccess;
      }
    }
    if (ok && s->is_qkv) {
      s->qkv_rows.n = s->buf.n;
      ok = hipMalloc(reinterpret_cast<void**>(&s->qkv_rows.p), s->qkv_rows.n * sizeof(float)) == hipSuccess;
      s->q_raw.n = (size_t)kHidden * s->shape[1];
      ok = ok && hipMalloc(reinterpret_cast<void**>(&s->q_raw.p), s->q_raw.n * sizeof(float)) == hipSuccess;
    }
    if (ok && s->want_wino) {
      s->wino.n = s->wino_dg.n = (size_t)16 * s->shape[0] * s->shape[1];
      ok = hipMalloc(reinterpret_cast<void**>(&s->wino.p), s->wino.n * sizeof(float)) == hipSuccess &&
           hipMalloc(reinterpret_cast<void**>(&s->wino_dg.p), s->wino_dg.n * sizeof(float)) == hipSuccess;
    }
    if (ok && s->dg.p && h->f16x3 && h->f16x3_direct && h->f16x3_dgrad) {
      const long long kk = s->kind == Kind::CONVT_IOHW ? s->shape[1] : s->shape[0];      // the data-gradient GEMM sums over the output channels
      s->dg_f16 = kk % 32 == 0;
    }
    if (ok && !s->want_wino && s->kind != Kind::RAW && h->f16x3 && h->f16x3_direct) {
      const long long cin = s->kind == Kind::CONV_OIHW ? s->shape[1] : s->shape[0];
      s->direct_f16 = cin % 32 == 0;
    }
    if (ok && s->want_wino && h->f16x3 && s->level >= h->f16x3_min_level) {
      // the forward GEMM sums over Cin, the data-gradient GEMM over Cout: each needs its K in whole 32-channel chunks
      s->wino_f16 = s->shape[1] % 32 == 0;
      s->wino_dg_f16 = s->shape[0] % 32 == 0;
    }
    // (not for K = Cin < US_WINO4_MIN_K (256): a frequency's GEMM is then four chunks long and the 36 of them run at 100 TFLOP/s; the level-1
    // 128 -> 256 convolution took 91 + 25 + 50 us (GEMM, input, output transform) against 92 + 35 in the fused-output F(2x2) form)
    static const int wino4_min_k = [] { const char* p = getenv("US_WINO4_MIN_K"); return p ? atoi(p) : 256; }();
    if (ok && s->want_wino && s->wino_f16 && s->level >= 0 && s->level < 8 && wino4_form_ok(h->wino4_level_form[s->level]) && s->shape[0] % 8 == 0 &&
        s->shape[1] >= wino4_min_k) {
      s->wino4_form = h->wino4_level_form[s->level];
      s->wino4.n = (size_t)wino4_freqs(s->wino4_form) * s->shape[0] * s->shape[1];
      ok = hipMalloc(reinterpret_cast<void**>(&s->wino4.p), s->wino4.n * sizeof(float)) == hipSuccess;
    }
    if (!ok) {
      g_last_error = "hipMalloc failed for weight store";
      for (auto& t : h->slots) { if (t->buf.p) (void)hipFree(t->buf.p); if (t->dg.p) (void)hipFree(t->dg.p); if (t->wino.p) (void)hipFree(t->wino.p); if (t->wino_dg.p) (void)hipFree(t->wino_dg.p); if (t->wino4.p) (void)hipFree(t->wino4.p); if (t->qkv_rows.p) (void)hipFree(t->qkv_rows.p); if (t->q_raw.p) (void)hipFree(t->q_raw.p); }
      return US_EHIP;
    }
  }
  h->copy_tab_cap = h->slots.size() + 8;      // (+ the input gradients of a backward's table)
  if (h->copy_tab_cap * sizeof(CopyEnt) > us_decoder::kStageBytes || 3 * h->slots.size() * sizeof(PackJob) > us_decoder::kStageBytes) {
    g_last_error = "architecture too large for the weight-table staging buffers";
    return US_EINVAL;
  }
  bool tab_ok = hipMalloc(reinterpret_cast<void**>(&h->copy_tab_dev), us_decoder::kStageBytes) == hipSuccess &&
                hipMalloc(reinterpret_cast<void**>(&h->grad_tab_dev), us_decoder::kStageBytes) == hipSuccess &&
                hipMalloc(reinterpret_cast<void**>(&h->lin_tab_dev), us_decoder::kLinJobs * sizeof(LinJob)) == hipSuccess &&
                hipMalloc(reinterpret_cast<void**>(&h->lin_tab_bwd_dev), us_decoder::kLinJobs * sizeof(LinJob)) == hipSuccess &&
                hipMalloc(reinterpret_cast<void**>(&h->pack_tab_dev), us_decoder::kStageBytes) == hipSuccess;
  for (int i = 0; i < 4 && tab_ok; ++i)
    tab_ok = hipHostMalloc(reinterpret_cast<void**>(&h->copy_tab_host[i]), us_decoder::kStageBytes) == hipSuccess;
  for (int i = 0; i < us_decoder::kCaptureTabs && tab_ok; ++i)
    tab_ok = hipHostMalloc(reinterpret_cast<void**>(&h->copy_tab_capture[i]), us_decoder::kStageBytes) == hipSuccess;
  if (!tab_ok) { g_last_error = "allocation of the weight-copy table failed"; return US_EHIP; }
  *out = h.release();
  return US_OK;
}

int us_decoder_flush_weights(us_handle h, us_stream stream) {
  if (!h) { g_last_error = "null argument"; return US_EINVAL; }
  return flush_copies(h, static_cast<hipStream_t>(stream));
}

int us_decoder_destroy(us_handle h) {
  if (!h) return US_OK;
  for (auto& s : h->slots) { if (s->buf.p) (void)hipFree(s->buf.p); if (s->dg.p) (void)hipFree(s->dg.p); if (s->wino.p) (void)hipFree(s->wino.p); if (s->wino_dg.p) (void)hipFree(s->wino_dg.p); if (s->wino4.p) (void)hipFree(s->wino4.p); if (s->qkv_rows.p) (void)hipFree(s->qkv_rows.p); if (s->q_raw.p) (void)hipFree(s->q_raw.p); }
  if (h->zeros) (void)hipFree(h->zeros);
  if (h->copy_tab_dev) (void)hipFree(h->copy_tab_dev);
  if (h->grad_tab_dev) (void)hipFree(h->grad_tab_dev);
  if (h->lin_tab_dev) (void)hipFree(h->lin_tab_dev);
  if (h->lin_tab_bwd_dev) (void)hipFree(h->lin_tab_bwd_dev);
  if (h->pack_tab_dev) (void)hipFree(h->pack_tab_dev);
  if (h->range_flag) (void)hipFree(h->range_flag);
  if (h->range_host) (void)hipHostFree(h->range_host);
  if (h->grad_scale) (void)hipFree(h->grad_scale);
  for (int i = 0; i < 4; ++i) { if (h->copy_tab_host[i]) (void)hipHostFree(h->copy_tab_host[i]); if (h->copy_tab_ev[i]) (void)hipEventDestroy(h->copy_tab_ev[i]); }
  for (int i = 0; i < us_decoder::kCaptureTabs; ++i) if (h->copy_tab_capture[i]) (void)hipHostFree(h->copy_tab_capture[i]);
  for (auto& r : h->prof_pending) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  for (auto& e : h->prof_pool) (void)hipEventDestroy(e);
  for (auto& e : h->wgrad_events) (void)hipEventDestroy(e);
  for (auto& st : h->wgrad_streams) (void)hipStreamDestroy(st);
  delete h;
  return US_OK;
}

int us_decoder_num_weights(us_handle h) { return h ? (int)h->slots.size() : 0; }
int us_decoder_num_loaded(us_handle h) {
  if (!h) return 0;
  int n = 0;
  for (auto& s : h->slots) n += s->loaded ? 1 : 0;
  return n;
}
const char* us_decoder_weight_key(us_handle h, int i) {
  if (!h || i < 0 || i >= (int)h->slots.size()) return nullptr;
  return h->slots[i]->key.c_str();
}

int us_decoder_load_weight(us_handle h, const char* key, const float* data, const int64_t* shape, int ndim, us_stream stream) {
  if (!h || !key || !data || !shape) { g_last_error = "null argument"; return US_EINVAL; }
  auto it = h->by_key.find(key);
  if (it == h->by_key.end()) return h->fail(US_ENOKEY, "unknown state_dict key '%s'", key);
  Slot* s = it->second;
  if (ndim != (int)s->shape.size()) return h->fail(US_ESHAPE, "'%s': expected %d dims, got %d", key, (int)s->shape.size(), ndim);
  for (int i = 0; i < ndim; ++i)
    if (shape[i] != s->shape[i]) return h->fail(US_ESHAPE, "'%s': dim %d is %lld, expected %lld", key, i, (long long)shape[i], (long long)s->shape[i]);
  hipStream_t st = static_cast<hipStream_t>(stream);
  RangeScope range_scope(h->range_flag);      // the f16x3 packs report a weight beyond the fp16 range
  // f16x3 packs are deferred like the RAW copies: one table-driven launch at us_decoder_flush_weights (`data` must stay valid until then)
  auto defer = [&](PackJob j) {
    for (auto& p : h->pending_packs)
      if (p.dst == j.dst) { p = j; return; }      // re-loaded before the flush: the later source wins
    h->pending_packs.push_back(j);
  };
  auto wino_f16 = [&](float* dst, bool dgrad) {
    defer(PackJob{data, reinterpret_cast<_Float16*>(dst), 0, (int)s->shape[0], (int)s->shape[1], 3, 3, dgrad ? 1 : 0, 0, 0, 0});
  };
  auto conv_f16 = [&](float* dst, int Cout, int Cin, int KH, int KW, bool oihw, bool qkv_rows = false) {
    defer(PackJob{data, reinterpret_cast<_Float16*>(dst), 1, Cout, Cin, KH, KW, oihw ? 1 : 0, qkv_rows ? 1 : 0, 0, 0});
  };
  switch (s->kind) {
    case Kind::RAW:
      // deferred: one table-driven launch for all RAW tensors at the next us_decoder_flush_weights / computing entry point.  `data`
      // must stay valid until then (the Python host flushes at the end of every weight sync).
      {
        bool replaced = false;
        for (auto& pc : h->pending_copies)
          if (pc.dst == s->buf.p) { pc.src = data; replaced = true; break; }      // re-loaded before the flush: the later source wins
        if (!replaced) h->pending_copies.push_back(CopyEnt{data, s->buf.p, (long long)s->buf.n});
      }
      // data-gradient packs: GEMM-K = output channels, GEMM-N = input channels.  f16x3: the forward pack routine with the channel
      // roles swapped (an OIHW tensor read as "IOHW" with I = its O)
      if (s->dg_as_1x1 && s->dg_f16)
        conv_f16(s->dg.p, (int)s->shape[1], (int)s->shape[0], 1, 1, false);
      else if (s->dg_as_1x1)
        US_HIP(h, launch_pack_dgrad_weight(data, s->dg.p, (int)s->shape[0], (int)s->shape[1], 1, 1, true, s->bk_dg, st));
      break;
    case Kind::CONV_OIHW:
      // a conv that runs in the Winograd domain never reads its direct-form pack (conv3x3 takes the Winograd branch whenever
      // wino.p is set; every workspace plan of this handle then has the V/M scratch)
      if (s->wino4.p) {
        // the inference-only 4-wide form: packed now (from `data`, in stream order) unless the handle is in training mode
        s->wino4_valid = !h->training;
        if (s->wino4_valid) US_HIP(h, launch_wino4_pack_weight(s->wino4_form, data, s->wino4.p, (int)s->shape[0], (int)s->shape[1], st));
      }
      if (s->wino.p && s->wino_f16) wino_f16(s->wino.p, false);
      else if (s->wino.p) US_HIP(h, launch_wino_pack_weight(data, s->wino.p, (int)s->shape[0], (int)s->shape[1], s->bk, st));
      else if (s->direct_f16) conv_f16(s->buf.p, (int)s->shape[0], (int)s->shape[1], (int)s->shape[2], (int)s->shape[3], true);
      else US_HIP(h, launch_pack_conv_weight(data, s->buf.p, (int)s->shape[0], (int)s->shape[1], (int)s->shape[2], (int)s->shape[3], true, s->bk, st));
      if (s->q_raw.p) {       // rows 0..127 of [384][C][1][1]: the tensor's first 128 * C floats, through the deferred copy table
        bool replaced = false;
        for (auto& pc : h->pending_copies)
          if (pc.dst == s->q_raw.p) { pc.src = data; replaced = true; break; }
        if (!replaced) h->pending_copies.push_back(CopyEnt{data, s->q_raw.p, (long long)s->q_raw.n});
      }
      if (s->qkv_rows.p && s->direct_f16) conv_f16(s->qkv_rows.p, (int)s->shape[0], (int)s->shape[1], 1, 1, true, true);
      else if (s->qkv_rows.p) US_HIP(h, launch_pack_conv_weight(data, s->qkv_rows.p, (int)s->shape[0], (int)s->shape[1], 1, 1, true, s->bk, st, true));
      if (s->wino_dg.p && s->wino_dg_f16) wino_f16(s->wino_dg.p, true);
      else if (s->wino_dg.p) US_HIP(h, launch_wino_pack_weight(data, s->wino_dg.p, (int)s->shape[0], (int)s->shape[1], s->bk_dg, st, true));
      else if (s->dg_f16) conv_f16(s->dg.p, (int)s->shape[1], (int)s->shape[0], (int)s->shape[2], (int)s->shape[3], false);
      else US_HIP(h, launch_pack_dgrad_weight(data, s->dg.p, (int)s->shape[0], (int)s->shape[1], (int)s->shape[2], (int)s->shape[3], true, s->bk_dg, st));
      break;
    case Kind::CONVT_IOHW:
      if (s->direct_f16) conv_f16(s->buf.p, (int)s->shape[1], (int)s->shape[0], (int)s->shape[2], (int)s->shape[3], false);
      else US_HIP(h, launch_pack_conv_weight(data, s->buf.p, (int)s->shape[1], (int)s->shape[0], (int)s->shape[2], (int)s->shape[3], false, s->bk, st));
      if (s->dg_f16) conv_f16(s->dg.p, (int)s->shape[0], (int)s->shape[1], (int)s->shape[2], (int)s->shape[3], true);
      else US_HIP(h, launch_pack_dgrad_weight(data, s->dg.p, (int)s->shape[1], (int)s->shape[0], (int)s->shape[2], (int)s->shape[3], false, s->bk_dg, st));
      break;
  }
  s->loaded = true;
  return US_OK;
}

int us_decoder_set_training(us_handle h, int training) {
  if (!h) { g_last_error = "null argument"; return US_EINVAL; }
  h->training = training != 0;
  return US_OK;
}

int us_decoder_stale_inference_forms(us_handle h) {
  if (!h) return 0;
  int n = 0;
  for (auto& s : h->slots) n += (s->wino4.p && s->loaded && !s->wino4_valid) ? 1 : 0;
  return n;
}

size_t us_workspace_bytes(us_handle h, int Bp, int T) {
  if (!h || Bp <= 0 || T <= 0) return 0;
  Arena A(nullptr, 0);
  Buffers b;
  plan(h, A, Bp, T, b);
  return A.off + 256;
}

size_t us_sampler_workspace_bytes(us_handle h, int mb, int T, int n_cfg) {
  if (!h || mb <= 0 || T <= 0 || n_cfg < 1) return 0;
  const size_t FT = (size_t)h->cfg.n_feats * T;
  const size_t Bp = (size_t)mb * n_cfg;
  // xt, score planes, spk rows, t, normalised spk_uncon (+ alignment slack) + one estimator workspace
  size_t own = (mb * FT + Bp * FT + Bp * h->cfg.spk_emb_dim + Bp + h->cfg.spk_emb_dim) * sizeof(float) + 16 * 256;
  own += time_block_floats(h, (int)Bp) * sizeof(float);
  return own + us_workspace_bytes(h, (int)Bp, T);
}

int us_estimator_forward(us_handle h, const float* x, const float* mask, const float* mu, const float* t, const float* spk,
                         float* out, int Bp, int T, void* workspace, size_t workspace_bytes, us_stream stream) {
  if (!h || !x || !mask || !mu || !t || !spk || !out || !workspace) { g_last_error = "null argument"; return US_EINVAL; }
  int rc = check_ready(h);
  if (rc) return rc;
  rc = check_shape(h, Bp, T);
  if (rc) return rc;
  if (workspace_bytes < us_workspace_bytes(h, Bp, T))
    return h->fail(US_EWORKSPACE, "workspace too small: %zu < %zu", workspace_bytes, us_workspace_bytes(h, Bp, T));
  Arena A(workspace, workspace_bytes);
  Buffers b;
  plan(h, A, Bp, T, b);
  EvalCtx e{h, static_cast<hipStream_t>(stream), &b, Bp, T, mask, Bp};
  e.infer = true;
  RangeScope range_scope(h->range_flag);
  US_HIP(h, estimator_eval(e, x, Bp, mu, Bp, 0, t, spk, out, true));
  return US_OK;
}

int us_range_status(us_handle h, unsigned* status, int reset, us_stream stream) {
  if (!h || !status) { g_last_error = "null argument"; return US_EINVAL; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  US_HIP(h, hipMemcpyAsync(h->range_host, h->range_flag, sizeof(unsigned), hipMemcpyDeviceToHost, st));
  if (reset) US_HIP(h, hipMemsetAsync(h->range_flag, 0, sizeof(unsigned), st));
  US_HIP(h, hipStreamSynchronize(st));
  *status = *h->range_host;
  return US_OK;
}

int us_range_status_async(us_handle h, unsigned* status_host, int reset, us_stream stream) {
  if (!h || !status_host) { g_last_error = "null argument"; return US_EINVAL; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  US_HIP(h, hipMemcpyAsync(status_host, h->range_flag, sizeof(unsigned), hipMemcpyDeviceToHost, st));
  if (reset) US_HIP(h, hipMemsetAsync(h->range_flag, 0, sizeof(unsigned), st));
  return US_OK;
}

int us_step_coefficients(int N, float beta_min, float beta_max, float* coef) {
  if (N < 1 || !coef) { g_last_error = "bad argument"; return US_EINVAL; }
  // fp32 arithmetic in the reference's order (`reverse_diffusion` :338-347, `register_beta` :235-271), including
  // the fp64 promotion of alphas_cumprod_prev-derived tables (:238-241) before the final fp32 cast (:271).
  const double hstep = 1.0 / N;
  std::vector<float> tt(N), acp_ext(N + 1);
  for (int i = 0; i < N; ++i) {
    float t = (float)(1.0 - (i + 0.5) * hstep);
    tt[i] = t;
    float cum = beta_min * t + (float)(0.5 * ((double)beta_max - (double)beta_min)) * (t * t);
    acp_ext[i] = expf(-cum);
  }
  acp_ext[N] = 1.f;
  std::vector<float> betas(N), acp(N);
  for (int j = 0; j < N; ++j) betas[N - 1 - j] = 1.f - acp_ext[j] / acp_ext[j + 1];
  float prod = 1.f;
  for (int j = 0; j < N; ++j) {
    float alpha = 1.f - betas[j];
    prod = j == 0 ? alpha : prod * alpha;
    acp[j] = prod;
  }
  for (int i = 0; i < N; ++i) {
    const int idx = N - 1 - i;
    const double acp_prev_d = idx == 0 ? 1.0 : (double)acp[idx - 1];
    const float acp_prev = (float)acp_prev_d;
    const float pv = (float)((double)betas[idx] * (1.0 - acp_prev_d) / (double)(1.f - acp[idx]));
    const float s1m = sqrtf(1.f - acp[idx]);
    const float sigma = sqrtf(pv);
    float* c = coef + (size_t)i * 8;
    c[0] = 1.f / sqrtf(acp[idx]);
    c[1] = sqrtf(1.f / acp[idx] - 1.f) * s1m;
    c[2] = sqrtf(acp_prev);
    c[3] = sqrtf(1.f - acp_prev - sigma * sigma);
    c[4] = s1m;
    c[5] = idx == 0 ? 0.f : sigma;
    c[6] = tt[i];
    c[7] = 0.f;
  }
  return US_OK;
}

int us_reverse_diffusion(us_handle h, const float* z, const float* mask, const float* cond, const float* spk, const float* noise,
                         uint64_t seed, int64_t utt_offset, int B, int T, int N, float w_text, float w_spk, const float* coef_host,
                         int micro_batch, const float* mel_range_host, float* out, void* workspace, size_t workspace_bytes,
                         us_stream stream) {
  if (!h || !z || !mask || !cond || !spk || !out || !workspace) { g_last_error = "null argument"; return US_EINVAL; }
  if (N < 1) return h->fail(US_EINVAL, "n_timesteps must be >= 1");
  int rc = check_ready(h);
  if (rc) return rc;
  const bool use_t = w_text > 0.f, use_s = w_spk > 0.f;
  const int n_cfg = 1 + (use_t ? 1 : 0) + (use_s ? 1 : 0);
  const int mode = use_t && use_s ? 3 : use_t ? 2 : use_s ? 1 : 0;
  int mbs = micro_batch > 0 ? micro_batch : 8;
  if (mbs > B) mbs = B;
  rc = check_shape(h, mbs * n_cfg, T);
  if (rc) return rc;
  const size_t need = us_sampler_workspace_bytes(h, mbs, T, n_cfg);
  if (workspace_bytes < need) return h->fail(US_EWORKSPACE, "workspace too small: %zu < %zu", workspace_bytes, need);

  std::vector<float> coef_own;
  if (!coef_host) {
    coef_own.resize((size_t)N * 8);
    us_step_coefficients(N, h->cfg.beta_min, h->cfg.beta_max, coef_own.data());
    coef_host = coef_own.data();
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int F = h->cfg.n_feats, S = h->cfg.spk_emb_dim;
  const size_t FT = (size_t)F * T;
  RangeScope range_scope(h->range_flag);

  Arena A(workspace, workspace_bytes);
  float* xt = A.alloc<float>((size_t)mbs * FT);
  float* score = A.alloc<float>((size_t)mbs * n_cfg * FT);
  float* spk_cfg = A.alloc<float>((size_t)mbs * n_cfg * S);
  float* tbuf = A.alloc<float>((size_t)mbs * n_cfg);
  float* spk_un = A.alloc<float>((size_t)S);
  TimeBlock tblock;
  plan_time_block(h, A, mbs * n_cfg, tblock);
  Buffers bufs;
  const size_t est_off = (A.off + 255) & ~size_t(255);
  if (use_s) US_HIP(h, launch_l2_normalize(h->spk_uncon->buf.p, spk_un, S, s));     // :358

  for (int b0 = 0; b0 < B; b0 += mbs) {
    const int mb = (B - b0) < mbs ? (B - b0) : mbs;
    const int Bp = mb * n_cfg;
    Arena AE(static_cast<char*>(workspace) + est_off, workspace_bytes - est_off);
    plan(h, AE, Bp, T, bufs);
    const float* z_b = z + (size_t)b0 * FT;
    const float* cond_b = cond + (size_t)b0 * FT;
    const float* mask_b = mask + (size_t)b0 * T;
    const float* spk_b = spk + (size_t)b0 * S;
    US_HIP(h, launch_mul_mask(z_b, mask_b, xt, mb, F, T, s));                         // xt = z * mask (:349)
    // speaker rows per CFG branch (:301-317): mode 3 [spk, uncon, spk], mode 2 [spk, spk], mode 1 [uncon, spk]
    for (int br = 0; br < n_cfg; ++br) {
      const bool uncon = (mode == 3 && br == 1) || (mode == 1 && br == 0);
      US_HIP(h, launch_copy_rows(uncon ? spk_un : spk_b, S, uncon ? 1 : mb, spk_cfg + (size_t)br * mb * S, S, mb, S, s));
    }
    const int n_text_uncond = use_t ? mb : 0;
    EvalCtx e{h, s, &bufs, Bp, T, mask_b, mb};
    e.infer = true;
    const std::vector<size_t> base_off = bufs.tproj_off;     // per-resnet offsets of ONE evaluation (plan())
    std::vector<int> couts(h->n_resnets, 0);
    for (auto& d : h->downs) { couts[d.r1.index] = d.r1.cout; couts[d.r2.index] = d.r2.cout; }
    couts[h->mid1.index] = h->mid1.cout; couts[h->mid2.index] = h->mid2.cout;
    for (auto& u : h->ups) { couts[u.r1.index] = u.r1.cout; couts[u.r2.index] = u.r2.cout; }
    float* eval_tproj = bufs.tproj;
    int blk0 = 0, blk_n = 0;
    for (int i = 0; i < N; ++i) {
      const float* c = coef_host + (size_t)i * 8;
      if (i >= blk0 + blk_n) {            // (re)fill the time-projection block
        blk0 = i;
        blk_n = (N - i) < kTimeBlock ? (N - i) : kTimeBlock;
        bufs.tproj = eval_tproj;
        bufs.tproj_off = base_off;
        US_HIP(h, compute_time_block(e, tblock, coef_host, blk0, blk_n, spk_cfg));
      }
      bufs.tproj = tblock.tproj;
      for (int r = 0; r < h->n_resnets; ++r)
        bufs.tproj_off[r] = (size_t)blk_n * base_off[r] + (size_t)(i - blk0) * Bp * couts[r];
      bufs.tproj_ready = true;
      US_HIP(h, estimator_eval(e, xt, mb, cond_b, mb, n_text_uncond, tbuf, spk_cfg, score, i == N / 2));
      SamplerArgs sa;
      memset(&sa, 0, sizeof sa);
      sa.xt = xt; sa.score = score; sa.mask = mask_b; sa.out = xt;
      sa.noise = noise ? noise + ((size_t)i * B + b0) * FT : nullptr;
      sa.B = mb; sa.F = F; sa.T = T; sa.mode = mode;
      sa.w_text = w_text; sa.w_spk = w_spk;
      sa.c0 = c[0]; sa.c1 = c[1]; sa.c2 = c[2]; sa.c3 = c[3]; sa.c4 = c[4]; sa.c5 = c[5];
      sa.seed = seed; sa.utt0 = utt_offset + b0; sa.step = i;
      US_HIP(h, launch_sampler_update(sa, s));
    }
    // :373 `xt * mask`, with the caller's mel de-normalisation (inference.py:140) folded into the same pass when asked for
    US_HIP(h, launch_finish_mel(xt, mask_b, out + (size_t)b0 * FT, mb, F, T, mel_range_host, s));
  }
  return US_OK;
}

int us_fill_normal(float* out, size_t n, uint64_t seed, uint64_t key, us_stream stream) {
  if (!out && n) { g_last_error = "null argument"; return US_EINVAL; }
  hipError_t e = launch_fill_normal(out, n, seed, key, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) { g_last_error = hipGetErrorString(e); return US_EHIP; }
  return US_OK;
}

double us_estimator_flops(us_handle h, int T) {
  if (!h) return 0.0;
  const int L = h->cfg.n_mults, F = h->cfg.n_feats;
  auto npx = [&](int l) { return (double)(F >> l) * (T >> l); };
  double fl = 0.0;
  auto conv = [&](double n, int cin, int cout, int taps) { fl += 2.0 * n * cin * cout * taps; };
  auto res = [&](const ResnetW& r) {
    double n = npx(r.level);
    conv(n, r.cin, r.cout, 9);
    conv(n, r.cout, r.cout, 9);
    if (r.has_res) conv(n, r.cin, r.cout, 1);
    fl += 2.0 * (h->cfg.dim + h->cfg.spk_emb_dim) * r.cout;
  };
  auto att = [&](const AttnW& a) {
    double n = npx(a.level);
    conv(n, a.dim, 3 * kHidden, 1);
    conv(n, kHidden, a.dim, 1);
    fl += 2.0 * 2.0 * n * kHeads * kDimHead * kDimHead;   // the two einsums
  };
  for (auto& d : h->downs) {
    res(d.r1); res(d.r2); att(d.a);
    if (d.has_ds) conv(npx(d.ds.level + 1), d.ds.dim, d.ds.dim, 9);
  }
  res(h->mid1); att(h->mid_attn); res(h->mid2);
  for (auto& u : h->ups) {
    res(u.r1); res(u.r2); att(u.a);
    conv(npx(u.us.level), u.us.dim, u.us.dim, 16);      // transposed 4x4: 16 MACs per INPUT pixel per channel pair
  }
  conv(npx(0), h->cfg.dim, h->cfg.dim, 9);
  conv(npx(0), h->cfg.dim, 1, 1);
  fl += 2.0 * (double)h->cfg.dim * 4 * h->cfg.dim * 2;       // time MLP
  (void)L;
  return fl;
}

int us_profile_enable(us_handle h, int enable) {
  if (!h) return US_EINVAL;
  h->prof_enabled = enable != 0;
  return US_OK;
}

int us_profile_read(us_handle h, double* conv_ms, double* conv_flops, int64_t* conv_launches, double* eval_ms, int64_t* evals,
                    int reset) {
  if (!h) return US_EINVAL;
  for (auto& r : h->prof_pending) {
    float ms = 0.f;
    if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess)
      return h->fail(US_EHIP, "profile event read failed (device not synchronised?)");
    if (r.kind == 0) {
      h->prof_conv_ms += ms; h->prof_conv_flops += r.flops; h->prof_conv_launches++;
      if (r.f16) { h->prof_f16_ms += ms; h->prof_f16_flops += r.flops; h->prof_f16_launches++; }
    }
    else { h->prof_eval_ms += ms; h->prof_evals++; }
    h->prof_pool.push_back(r.a);
    h->prof_pool.push_back(r.b);
  }
  h->prof_pending.clear();
  if (conv_ms) *conv_ms = h->prof_conv_ms;
  if (conv_flops) *conv_flops = h->prof_conv_flops;
  if (conv_launches) *conv_launches = h->prof_conv_launches;
  if (eval_ms) *eval_ms = h->prof_eval_ms;
  if (evals) *evals = h->prof_evals;
  if (reset) {
    h->prof_conv_ms = h->prof_conv_flops = h->prof_eval_ms = 0; h->prof_conv_launches = h->prof_evals = 0;
    h->prof_f16_ms = h->prof_f16_flops = 0; h->prof_f16_launches = 0;
  }
  return US_OK;
}

int us_profile_read_f16(us_handle h, double* f16_ms, double* f16_flops, int64_t* f16_launches) {
  if (!h) return US_EINVAL;
  if (f16_ms) *f16_ms = h->prof_f16_ms;
  if (f16_flops) *f16_flops = h->prof_f16_flops;
  if (f16_launches) *f16_launches = h->prof_f16_launches;
  return US_OK;
}

#include "train_abi.inc"

// ---- one building block on its own (parity tests against the reference's per-module outputs) ------------------------------
int us_debug_block(us_handle h, int kind, const char* prefix, int level, const float* x, const float* mask, const float* temb, float* out,
                   int B, int T, void* workspace, size_t workspace_bytes, us_stream stream) {
  if (!h || !prefix || !x || !mask || !out || !workspace) { g_last_error = "null argument"; return US_EINVAL; }
  int rc = check_ready(h);
  if (rc) return rc;
  rc = check_shape(h, B, T);
  if (rc) return rc;
  if (level < 0 || level >= h->cfg.n_mults) return h->fail(US_EINVAL, "level %d out of range", level);
  if (workspace_bytes < us_workspace_bytes(h, B, T)) return h->fail(US_EWORKSPACE, "workspace too small");
  const std::string p(prefix);
  hipStream_t s = static_cast<hipStream_t>(stream);
  Arena A(workspace, workspace_bytes);
  Buffers b;
  plan(h, A, B, T, b);
  EvalCtx e{h, s, &b, B, T, mask, B};
  e.infer = true;
  RangeScope range_scope(h->range_flag);
  US_HIP(h, hipMemsetAsync(b.stats, 0, b.stats_count * sizeof(double), s));
  std::vector<const ResnetW*> rs;
  std::vector<const AttnW*> as;
  for (auto& d : h->downs) { rs.push_back(&d.r1); rs.push_back(&d.r2); as.push_back(&d.a); }
  rs.push_back(&h->mid1); rs.push_back(&h->mid2); as.push_back(&h->mid_attn);
  for (auto& u : h->ups) { rs.push_back(&u.r1); rs.push_back(&u.r2); as.push_back(&u.a); }
  if (kind == US_DEBUG_TEMB) {
    // x = t [B]; out [B][2 * dim] = SinusoidalPosEmb(t) | mlp(SinusoidalPosEmb(t))  (unitspeech.py:109-121,133-134,165-166), by the launches
    // time_embedding() makes
    const int dim = h->cfg.dim;
    US_HIP(h, launch_pos_emb(x, b.posemb, B, dim, h->cfg.pe_scale, s));
    US_HIP(h, launch_linear(b.posemb, dim, h->mlp0_w->buf.p, h->mlp0_b->buf.p, b.mlp_h, 4 * dim, B, dim, 4 * dim, false, s));
    US_HIP(h, launch_linear(b.mlp_h, 4 * dim, h->mlp2_w->buf.p, h->mlp2_b->buf.p, b.temb, dim + h->cfg.spk_emb_dim, B, 4 * dim, dim, true, s));
    US_HIP(h, launch_copy_rows(b.posemb, dim, B, out, 2 * dim, B, dim, s));
    US_HIP(h, launch_copy_rows(b.temb, dim + h->cfg.spk_emb_dim, B, out + dim, 2 * dim, B, dim, s));
    return US_OK;
  }
  if (kind == US_DEBUG_BLOCK || kind == US_DEBUG_RESNET) {
    for (const ResnetW* r0 : rs) {
      if (r0->mlp_w->key != p + ".mlp.1.weight") continue;
      if (r0->first) return h->fail(US_EINVAL, "the 2-channel first block has no stand-alone entry");
      ResnetW r = *r0;
      r.level = level;                   // geometry (H, W, mask stride) of the requested level; the weights are the module's own
      if (kind == US_DEBUG_BLOCK) {      // Block = block1 of that ResnetBlock: Mish(GroupNorm(conv(x))) * mask, x pre-masked (:46-55)
        double* st = next_stats(e);
        US_HIP(h, conv3x3(e, r.c1, x, r.cin, level, b.S1[level], r.cout, st));
        US_HIP(h, gn_apply(e, b.S1[level], level, r.cout, st, r.g1, r.b1, nullptr, nullptr, 0, false, false, out, r.cout));
        return US_OK;
      }
      if (!temb) return h->fail(US_EINVAL, "ResnetBlock needs temb");
      const int td = h->cfg.dim + h->cfg.spk_emb_dim;
      US_HIP(h, launch_linear(temb, td, r.mlp_w->buf.p, r.mlp_b->buf.p, b.tproj + b.tproj_off[r.index], r.cout, B, td, r.cout, true, s));
      US_HIP(h, resnet(e, r, x, r.cin, out, r.cout, false));
      return US_OK;
    }
    return h->fail(US_ENOKEY, "no ResnetBlock '%s'", prefix);
  }
  if (kind == US_DEBUG_ATTENTION) {
    for (const AttnW* a0 : as) {
      if (a0->g->key != p + ".fn.g") continue;
      AttnW a = *a0;
      a.level = level;
      US_HIP(h, attention(e, a, x, a.dim, out, a.dim));
      return US_OK;
    }
    return h->fail(US_ENOKEY, "no attention '%s'", prefix);
  }
  if (kind == US_DEBUG_DOWN) {
    for (auto& d : h->downs)
      if (d.has_ds && d.ds.conv.w->key == p + ".conv.weight") {
        if (level + 1 >= h->cfg.n_mults) return h->fail(US_EINVAL, "no level below %d", level);
        US_HIP(h, conv_down(e, d.ds.conv, x, d.ds.dim, level, out, d.ds.dim));
        return US_OK;
      }
    return h->fail(US_ENOKEY, "no Downsample '%s'", prefix);
  }
  if (kind == US_DEBUG_UP) {
    for (auto& u : h->ups)
      if (u.us.conv.w->key == p + ".conv.weight") {
        if (level < 1) return h->fail(US_EINVAL, "no level above 0");
        US_HIP(h, conv_up(e, u.us.conv, x, u.us.dim, level, out, u.us.dim));
        return US_OK;
      }
    return h->fail(US_ENOKEY, "no Upsample '%s'", prefix);
  }
  return h->fail(US_EINVAL, "unknown block kind %d", kind);
}

const char* us_last_error(us_handle h) { return h ? h->err.c_str() : g_last_error.c_str(); }

}  // extern "C"
