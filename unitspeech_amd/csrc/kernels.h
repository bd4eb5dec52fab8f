// Internal launcher interface between the decoder host code (decoder.hip) and the gfx950 kernels.
// Activation layout everywhere inside the library: pixel-major "NHWC" fp32, i.e. [B'][H][W][ld] with the
// channel index fastest and `ld` (>= C) the pixel stride in floats, so that a producer can write straight
// into one half of a skip-connection concat buffer (unitspeech/unitspeech.py:192).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace us {

// Correctly rounded single operations that the compiler may NOT contract into an fma.  hipcc compiles with -ffp-contract=fast and
// HIP's __fmul_rn / __fadd_rn are plain operators to it (a `__fadd_rn(__fmul_rn(a, b), c)` becomes one v_fma_f32), so expressions
// that must round like the reference's separate tensor ops (sampler update, forward diffusion, mel de-normalisation) use these.
#if defined(__HIPCC__)
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float sub_rn(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}
#endif

// ---- f16x3 operand range ---------------------------------------------------------------------------------------------
// An f16x3 GEMM operand is two fp16 planes, x ~= hi + lo * 2^-11 (hi = fp16(x), lo = fp16((x - hi) * 2^11)): exact to 22 bits for
// |x| < 65520, the first magnitude fp16 rounds to infinity.  The reference computes in plain fp32 (unitspeech/unitspeech.py:46-96), so a
// larger value must not be replaced quietly: every place that forms the two planes lets it become an infinity (the GEMM's outputs are
// then non-finite, never a clamped look-alike), lets a NaN stay a NaN, and ORs a bit into the handle's range word, which
// us_range_status reports and the Python mirror answers by re-running the call on the exact-fp32 MFMA path.
constexpr float kF16Over = 65520.f;
enum : unsigned {
  kRangeAct = 1u,      // an activation (or Winograd-domain value, or gradient) beyond the fp16 range met an f16x3 split
  kRangeWeight = 2u,   // a weight (or folded attention weight) did
};
#if defined(__HIPCC__)
typedef _Float16 us_half;
__device__ __forceinline__ void split_f16x3(float x, us_half& hi, us_half& lo, bool& over) {
  const us_half h = (us_half)x;                 // +-inf beyond the range, NaN stays NaN
  hi = h;
  lo = (us_half)((x - (float)h) * 2048.f);      // |x - h| <= ulp(h) / 2 <= 16: the scaled remainder always fits
  over |= fabsf(x) >= kF16Over;                 // false for a NaN: the reference would carry it too
}
__device__ __forceinline__ void range_report(unsigned* flag, bool over, unsigned bit) {
  if (flag && over) atomicOr(flag, bit);        // rare by construction: no contention to speak of
}
#endif
// The range word of the handle whose entry point is running on this host thread (decoder.hip sets it around every call that can
// launch a split; null outside): launchers read it when they fill their kernel arguments.
unsigned* current_range_flag();
void set_range_flag(unsigned* p);
struct RangeScope {        // RAII: the handle's range word is current while an entry point enqueues work
  unsigned* prev;
  explicit RangeScope(unsigned* p) : prev(current_range_flag()) { set_range_flag(p); }
  ~RangeScope() { set_range_flag(prev); }
  RangeScope(const RangeScope&) = delete;
  RangeScope& operator=(const RangeScope&) = delete;
};

constexpr int kHeads = 4;       // unitspeech/unitspeech.py:79
constexpr int kDimHead = 32;    // unitspeech/unitspeech.py:79
constexpr int kHidden = kHeads * kDimHead;
constexpr int kGroups = 8;      // unitspeech/unitspeech.py:47
// GroupNorm partial sums: per (item, group) a (sum, sum of squares) pair, each kept as kStatSlots fp64 partials that producers
// hit by workgroup index and readers add up.  A single accumulator per pair made every workgroup of a convolution queue on the
// same 48 addresses: measured on the level-0 3x3 (3,840 workgroups) 100 us of a 400 us launch, 47 of 253 on the fused Winograd form.
constexpr int kStatSlots = 16;
constexpr int kStatStride = 2 * kStatSlots;        // doubles per (item, group)
constexpr int kMaxTaps = 16;

// ---- implicit-GEMM convolution on fp32 MFMA -------------------------------------------------------
struct ConvArgs {
  const float* in;       // [B][Hin][Win][in_ld], channels [0,Cin); already multiplied by the frame mask where the
                         // reference masks this operand
  const float* wt;       // packed [tap][Cin/BK][Cout][BK]
  const float* bias;     // [Cout] or null
  const float* omask;    // optional frame mask applied to the stored value: column ox reads omask[ox*omask_step]
  const float* add;      // optional addend, pixel-indexed like out
  const float* alpha;    // optional device scalar: out = (add + alpha*(acc + bias)) * omask
  const float* zeros;    // zero page (>= max Cin floats) that out-of-image taps read
  float* out;            // [B][Hout][Wout][out_ld]
  double* stats;         // optional GroupNorm partial sums [B][8][2][kStatSlots] (sum, sumsq) of acc + bias
  long long wt_bstride;  // per-item weight stride in floats (0 = shared weights)
  int xcd_z;             // blockIdx.z enumerates independent GEMMs (Winograd frequencies): deal whole z-slices to an XCD (conv_igemm.hip)
  int wt_bdiv;           // item b reads the weights at wt + (b / max(wt_bdiv,1)) * wt_bstride (Winograd: one matrix per frequency)
  int in_ld, out_ld, add_ld;
  int B, Hin, Win, Cin, Hout, Wout, Cout;
  int Hs, Ws;            // output sub-grid handled by this launch
  int oy0, ox0, ostep;   // output pixel = (oy0 + my*ostep, ox0 + mx*ostep)
  int istride;           // input pixel  = (my*istride + dy[tap], mx*istride + dx[tap])
  int ntaps;
  int omask_ld, omask_step, omask_bmod;
  int bk;                // 16 or 32: channel chunk the weights were packed for
  int tm;                // rows per workgroup: 128, 64 or 0 = choose from the grid size
  int ksplit;            // set by the launcher: K slices per tile (1 = single pass)
  int nt;                // set by the launcher: column tiles (grid x = row tiles * ksplit * nt, column tile fastest)
  int f16;               // f16x3 GEMM (bk = 32): `wt` holds two interleaved fp16 planes per value; 1: so does `in` (Winograd domain),
                         // 2: `in` is a plain fp32 tensor that the kernel splits on the fly (direct convolutions)
  int wino_out;          // 1: `in` = V [16][B][Hs][Ws][Cin], `wt` = 16 matrices wt_bstride apart; Winograd output transform in the kernel
  int splitk_by_batch;   // 1: the split-K slice count may depend on the batch (training); 0: per-item geometry only, so that an
                         // utterance's result never depends on what it is batched with (sampling)
  float* splitk_ws;      // optional scratch for split-K partial slabs (splitk_ws_floats floats); null disables split-K
  long long splitk_ws_floats;
  unsigned* range_flag;  // set by the launcher (current_range_flag()): where an in-kernel f16x3 split reports an operand beyond the fp16 range
  int direct_presplit;   // f16 = 1 on a DIRECT convolution (its producer stored the input with out_split), as opposed to a Winograd-domain
                         // GEMM: the tile follows the direct rule (a function of the item's geometry only, never of the batch)
  int out_split;         // 1: the stored tensor is the two-plane fp16 form (per 8 channels 8 hi | 8 lo, same bytes as fp32; Cout, out_ld and
                         // the channel offset of `out` multiples of 8) that an f16x3 convolution takes as a pre-split A operand (f16 = 1): for
                         // tensors whose every consumer is such a convolution (no split-K; a value beyond the fp16 range is reported)
  // to_qkv of the linear attention with the n-reduction in its epilogue (inference): `wt` holds the rows in qkv_src_row() order, so
  // column tile 0 is q (stored to `out`, ld 128) and every wave of tiles 1 / 2 holds k_h | v_h of one head: it takes the column
  // maxima, exp(k - m), the column sums and ctx_h = exp(k - m)^T v over the workgroup's 64 rows (accumulator registers as MFMA
  // operands, no LDS transpose) and writes them as chunk (item, row tile) of the online-softmax partials; k and v never reach memory
  float* attn_part_ctx;  // [B][attn_nchunks][4][32][32]; non-null selects the mode (single pass, 64-row tiles)
  float* attn_part_m;    // [B][attn_nchunks][128]
  float* attn_part_s;    // [B][attn_nchunks][128]
  int attn_nchunks;      // ceil(Hs * Ws / attn_rows)
  int attn_rows;         // rows per chunk = rows per workgroup: 64 (0 means 64) or 128
  int attn_q_cols;       // kHidden: column tile 0 is q and is stored; 0: the launch computes k | v only (Cout = 2 * kHidden, nothing stored:
                         // the caller folds W_q into the output projection, launch_attn_wtotal)
  int wt_rows;           // rows per K-chunk of the packed weight when the launch uses only Cout of them (0: Cout)
  int splitk_raw;        // Winograd-domain GEMMs of a training pass (B = frequencies): the launcher may slice K and leave the raw slabs
                         // [ks][B][Ms][Cout] in splitk_ws for the output transform to sum in slice order (no finish launch); the slice count
  int* ksplit_out;       // ... is returned here (host pointer; 1 = `out` was written as usual)
  float* out2;           // optional second output, pixel-indexed like out (ld out2_ld): acc + bias BEFORE alpha / add / mask.  Training keeps
  int out2_ld;           // the Rezero branch's fn(x) this way: the gain's gradient is sum(grad_out * fn(x)) (unitspeech/unitspeech.py:36-43)
#if defined(US_STAMP) || defined(US_LIFE)
  unsigned long long* stamp_out;   // diagnostic build only: per (workgroup, wave) {cycles at the vmcnt wait, at the barrier, in the body, steps}
#endif
  int debug;             // timing ablations for tools/conv_bench (0 in production): 1 = no DMA after the prologue,
                         // 2 = no fragment reads after the first, 4 = no barriers
  unsigned long long dy_bits, dx_bits, wtap_bits;   // 4 bits per tap: dy+8, dx+8, weight tap index
  void set_tap(int i, int dy, int dx, int wtap) {
    dy_bits |= (unsigned long long)(dy + 8) << (4 * i);
    dx_bits |= (unsigned long long)(dx + 8) << (4 * i);
    wtap_bits |= (unsigned long long)wtap << (4 * i);
  }
  // nphase = 4: the four output phases (oy0, ox0) = (ph >> 1, ph & 1) of a stride-2 transposed convolution in ONE launch
  // (grid z = B * 4), each with its own tap table; oy0 / ox0 / dy_bits / dx_bits / wtap_bits above are then unused.  One
  // 4x larger grid instead of four sub-wave ones back to back.
  int nphase;
  unsigned long long ph_dy[4], ph_dx[4], ph_wtap[4];
  void set_phase_tap(int ph, int i, int dy, int dx, int wtap) {
    ph_dy[ph] |= (unsigned long long)(dy + 8) << (4 * i);
    ph_dx[ph] |= (unsigned long long)(dx + 8) << (4 * i);
    ph_wtap[ph] |= (unsigned long long)wtap << (4 * i);
  }
};
#if defined(__HIPCC__)
__device__ __forceinline__ void stat_add(double* stats, long long b, int group, int which, unsigned slot, double v) {
  atomicAdd(stats + ((b * kGroups + group) * 2 + which) * kStatSlots + (slot & (kStatSlots - 1)), v);
}
__device__ __forceinline__ double stat_read(const double* stats, long long b, int group, int which) {
  const double* p = stats + ((b * kGroups + group) * 2 + which) * kStatSlots;
  double t = 0.0;
#pragma unroll
  for (int i = 0; i < kStatSlots; ++i) t += p[i];
  return t;
}
#endif
// packed (tile-ordered) row of the attention's to_qkv weight -> the reference's output channel (q | k | v, heads x 32 each,
// unitspeech/unitspeech.py:89-90): rows 0..127 q; then per head h the 64 rows k_h (32) | v_h (32)
inline int qkv_src_row(int cp) {
  if (cp < kHidden) return cp;
  const int t = cp - kHidden, h = t >> 6, w = t & 63;
  return (w < kDimHead ? kHidden : 2 * kHidden) + h * kDimHead + (w & (kDimHead - 1));
}
hipError_t launch_conv_igemm(const ConvArgs& a, hipStream_t s);
hipError_t conv_igemm_init();   // one-time function attributes (dynamic LDS size)

// Repack reference-layout weights into [tap][Cin/bk][Cout][bk].
//   oihw = true : src is Conv2d  [Cout][Cin][KH][KW]
//   oihw = false: src is ConvTranspose2d [Cin][Cout][KH][KW]
//   qkv_rows: packed row r holds source output channel qkv_src_row(r) (Cout = 384: the attention's to_qkv, see ConvArgs::attn_part_ctx)
hipError_t launch_pack_conv_weight(const float* src, float* dst, int Cout, int Cin, int KH, int KW, bool oihw, int bk,
                                   hipStream_t s, bool qkv_rows = false);
// the same pack as two interleaved fp16 planes per value (f16x3 GEMM, bk = 32; same size)
hipError_t launch_pack_conv_weight_f16(const float* src, float* dst, int Cout, int Cin, int KH, int KW, bool oihw, hipStream_t s,
                                       bool qkv_rows = false);

// ---- first layer (2 input channels) ---------------------------------------------------------------
// in2: [Bp][F][T][2] = (mu, x) already masked; writes conv3x3 (pad 1) -> y[Bp][F*T][C] and the ResnetBlock's
// 1x1 res_conv -> r[Bp][F*T][C].  w3: reference OIHW [C][2][3][3]; w1: [C][2].
// stats (optional, zeroed by the caller): GroupNorm(8) partial sums [Bp][8][2] of y
hipError_t launch_first_conv(const float* in2, const float* w3, const float* b3, const float* w1, const float* b1, float* y,
                             float* r, double* stats, int Bp, int F, int T, int C, hipStream_t s);
// Builds in2 from planar inputs.  x: [Bx][F][T], mu: [Bmu][F][T], mu_feat: [F] (text_uncon, broadcast over T);
// item b' reads x[b' % Bx], mask[b' % Bm] and mu_feat when b' < n_text_uncond, else mu[b' % Bmu].
hipError_t launch_stack_inputs(const float* x, int Bx, const float* mu, int Bmu, int n_text_uncond, const float* mu_feat,
                               const float* mask, int Bm, float* in2, int Bp, int F, int T, hipStream_t s);

// ---- GroupNorm(8) + Mish ----------------------------------------------------------------------------
// stats[B][8][2] += (sum, sumsq) over [n][C] per item; must be zeroed beforehand.
hipError_t launch_gn_stats(const float* y, int ld, int B, int n, int C, double* stats, hipStream_t s);
// out = mish(gn(y)) * mask (+ temb[b][c]) (+ res[p][c] [* mask]), then * mask again when post_mask
// (Block :46-55, ResnetBlock :69-75; post_mask pre-applies the `x * mask` of the tensor's consumers)
struct GnApplyArgs {
  const float* y; int y_ld;
  const double* stats;
  const float* gamma; const float* beta;
  const float* mask; int mask_ld, mask_step, mask_bmod;   // frame mask, column w reads mask[w*mask_step]
  const float* temb; int temb_ld;                          // optional [B][temb_ld] per-channel addend
  const float* res; int res_ld; int res_masked;            // optional residual (times mask when res_masked)
  const float* res2_in; const float* res2_w; const float* res2_b;   // or (res == null): residual = Conv2d(2, C, 1) of the stacked input
                                                           // [B][H*W][2], weight [C][2], bias [C] (the first ResnetBlock's res_conv)
  int post_mask;
  unsigned* range_flag;                                    // set by the launcher; out_split reports here
  int out_split;                                           // out receives the two-plane fp16 form (per 8 channels: 8 hi | 8 lo, same bytes
                                                           // as fp32) that a direct f16x3 convolution takes as a pre-split A operand;
                                                           // out must not alias y (a quad's lo plane lands on its neighbour's input)
  float* out; int out_ld;
  float* out2; int out2_ld;                                // optional second copy of the result in the two-plane form (out stays fp32): a tensor
                                                           // read both as a residual (fp32) and by a direct f16x3 convolution (pre-split)
  int B, H, W, C;
};
hipError_t launch_gn_apply(const GnApplyArgs& a, hipStream_t s);
// final_conv (1x1, C->1) of masked input, times mask: out[b][p] = (b0 + sum_c w[c]*h[p][c]*m)*m        (:199-201)
hipError_t launch_final_conv(const float* h, int ld, const float* w, const float* b0, const float* mask, int mask_ld,
                             int mask_bmod, float* out, int B, int H, int W, int C, hipStream_t s, const double* stats = nullptr,
                             const float* gamma = nullptr, const float* beta = nullptr);   // stats: GroupNorm + Mish + mask of `h` on the fly

// ---- linear attention --------------------------------------------------------------------------------
// qkv: [B][n][384] (q | k | v, each heads*32).  Stage 1: per 128-row chunk, column max / sum-exp of k and the
// partial context exp(k-m)^T v per head.  Stage 2: combine chunks -> ctx[B][4][32][32] (softmax-normalised).
// Stage 3: fold ctx into to_out: weff[B] packed [1][128/bk][C][bk], weff[c][h*32+d] = sum_e Wout[c][h*32+e]*ctx[h][d][e].
hipError_t launch_attn_ctx_partial(const float* qkv, int B, int n, float* part_ctx, float* part_m, float* part_s, int nchunks,
                                   hipStream_t s);
// colM / colS ([B][128], required): the softmax column max and sum-exp (also kept for the backward pass)
// split_ws: scratch of kAttnMaxSplit * B * 4096 floats for the per-range slabs (null = single range)
constexpr int kAttnMaxSplit = 16;
hipError_t launch_attn_ctx_finalize(const float* part_ctx, const float* part_m, const float* part_s, int B, int nchunks,
                                    float* ctx, float* colM, float* colS, float* split_ws, hipStream_t s);
// f16 = true (bk = 32): weff as two interleaved fp16 planes per value, for the f16x3 form of the folded to_out convolution
// W_total[b] = W_out blockdiag(ctx[b]^T) W_q  ([C][C]; the attention's q never exists: out = x + g (W_total x + b_o)), stored in the conv kernel's
// packed layout for K = N = C (f16: two-plane form, bk 32; else fp32 [C/bk][C][bk]).  wq: rows 0..127 of to_qkv's weight, [128][C].
hipError_t launch_attn_wtotal(const float* ctx, const float* wout, const float* wq, float* wtotal, int B, int C, int bk, bool f16, hipStream_t s);
hipError_t launch_attn_weff(const float* ctx, const float* wout /*[C][128]*/, float* weff, int B, int C, int bk, hipStream_t s,
                            bool f16 = false);
// The same merge in two launches (ranges of chunks with their own maxima, then the ranges), with the fold of ctx into to_out's weights
// (launch_attn_weff) done by the second one when weff != null.  scratch: attn_merge_scratch_floats(B) floats.
inline size_t attn_merge_scratch_floats(int B) { return (size_t)kAttnMaxSplit * B * kHeads * (kDimHead * kDimHead + 2 * kDimHead); }
hipError_t launch_attn_merge(const float* part_ctx, const float* part_m, const float* part_s, int B, int nchunks, float* ctx, float* colM,
                             float* colS, float* scratch, const float* wout, float* weff, int C, int bk, bool f16, hipStream_t s);
inline int attn_nchunks(int n) { return (n + 127) / 128; }

// ---- small dense layers ----------------------------------------------------------------------------
// out[r][o] = bias[o] + sum_i W[o][i] * f(in[r][i]),  f = mish when mish_in
hipError_t launch_linear(const float* in, int in_ld, const float* W, const float* bias, float* out, int out_ld, int rows,
                         int in_dim, int out_dim, bool mish_in, hipStream_t s);
// SinusoidalPosEmb (:109-121): emb[r][0:half]=sin, [half:dim]=cos of scale*t[r]*exp(-k*ln(1e4)/(half-1))
hipError_t launch_pos_emb(const float* t, float* emb, int rows, int dim, float scale, hipStream_t s);
// dst[r][0:n] = src[src_index ? src_index[r] : r % src_rows][0:n]   (row gather/copy into a strided destination)
hipError_t launch_copy_rows(const float* src, int src_ld, int src_rows, float* dst, int dst_ld, int rows, int n, hipStream_t s);
hipError_t launch_fill(float* dst, float value, int n, hipStream_t s);
// table-driven copy of many small fp32 tensors in one launch: tab[i] = {src, dst, n} in device memory
struct CopyEnt { const float* src; float* dst; long long n; };
hipError_t launch_copy_table(const CopyEnt* tab_dev, int n_entries, hipStream_t s);
// dst[i] = src[i] * scale_dev[0] for every entry of the same kind of table (in place when src == dst)
hipError_t launch_scale_table(const CopyEnt* tab_dev, int n_entries, const float* scale_dev, hipStream_t s);
// out[0] = 2^k with max|x| * 2^k in [2^(target_log2 - 1), 2^target_log2), out[1] = 2^-k (1, 1 for an all-zero or non-finite x); glue.hip
hipError_t launch_pow2_scale(const float* x, long long n, int target_log2, float* out2, hipStream_t s);
hipError_t launch_scale(const float* x, const float* scalar_dev, float* out, long long n, hipStream_t s);
// dst = src / ||src||_2 over n elements (spk_uncon normalisation, :358)
hipError_t launch_l2_normalize(const float* src, float* dst, int n, hipStream_t s);

// ---- sampler ---------------------------------------------------------------------------------------
// CFG combine (:320-330) + ancestral update (:273-296, :366-370) on [B][F*T] planes.
// score: [n_cfg*B][F*T] in branch-major order.  mode: 3 = text+spk, 2 = text only, 1 = spk only, 0 = none.
struct SamplerArgs {
  const float* xt; const float* score; const float* noise; const float* mask;   // mask [B][T]
  float* out;
  int B, F, T, mode;
  float w_text, w_spk;
  float c0, c1, c2, c3, c4, c5;
  // built-in generator (used when noise == null and c5 != 0)
  unsigned long long seed; long long utt0; int step;
};
hipError_t launch_sampler_update(const SamplerArgs& a, hipStream_t s);
hipError_t launch_mul_mask(const float* x, const float* mask, float* out, int B, int F, int T, hipStream_t s);
// out = x * mask, then (mel_range_host != null) the de-normalisation (v + 1) / 2 * (max - min) + min of inference.py:140;
// mel_range_host = {mel_min, mel_max} on the host
hipError_t launch_finish_mel(const float* x, const float* mask, float* out, int B, int F, int T, const float* mel_range_host, hipStream_t s);
// data gradient of the 2-channel first ResnetBlock back to the planar inputs (glue.hip); gmu / gx: [B][F][T] or null
hipError_t launch_first_conv_dgrad(const float* gy, const float* gr, const float* w3, const float* w1, const float* mask, float* gmu,
                                   float* gx, int B, int F, int T, int C, hipStream_t s);
// thread-local "last error" text of the library (decoder.hip), for entry points that have no handle
void set_last_error(const char* msg);
hipError_t launch_fill_normal(float* out, size_t n, unsigned long long seed, unsigned long long key, hipStream_t s);

// ---- backward pass (train.hip) ---------------------------------------------------------------------------
// gw[tap][Cout][Cin] (+ item*gw_bstride) += sum over output pixels of gy[opix][co] * x[ipix(tap)][ci]; same sub-grid /
// tap geometry fields as ConvArgs.  gw must be zeroed by the caller (fp32 atomics).
struct WgradArgs {
  const float* gy; int gy_ld;
  const float* x; int x_ld;
  float* gw; long long gw_bstride;
  int B, Hin, Win, Cin, Hout, Wout, Cout;
  int Hs, Ws, oy0, ox0, ostep, istride, ntaps;
  int chunk;             // output pixels per workgroup (multiple of 8)
  int vchunk;            // f16x3 kernel, shared gw (gw_bstride == 0): pixels per workgroup of the range CONCATENATED over the B items
                         // (0 = per-item chunks as above); set by launch_wgrad
  int exact;             // 1: the exact-fp32 MFMA form (a handle created with US_CREATE_EXACT_FP32), whatever US_WGRAD_F16 says
  const float* gy_amax;  // device float: max |gy| over (at least) the pixels and channels this launch reads, taken by launch_wgrad_amax
                         // into a zeroed word; the f16x3 form derives its exact power-of-two scale of the gradient operand from it
  int overwrite;         // 1: gw holds garbage; legal only when every element has exactly one writer (launch_wgrad_single_writer)
  int div_addr;          // f16x3 kernel: 1 = recompute every DMA row's coordinates by division in every stage (the form until round 4; US_WGRAD_DIV=1)
  unsigned long long dy_bits, dx_bits, wtap_bits;
  void set_tap(int i, int dy, int dx, int wtap) {
    dy_bits |= (unsigned long long)(dy + 8) << (4 * i);
    dx_bits |= (unsigned long long)(dx + 8) << (4 * i);
    wtap_bits |= (unsigned long long)wtap << (4 * i);
  }
};
hipError_t launch_wgrad(const WgradArgs& a, hipStream_t s);
// *out = max(*out, max |g[r][c]|) over rows x C (C, ld multiples of 4); *out must be zero (or an earlier maximum to extend)
hipError_t launch_wgrad_amax(const float* g, int ld, long long rows, int C, float* out, hipStream_t s);
// true when launch_wgrad(a) gives every gw element exactly one writing workgroup (LDS kernel, one pixel chunk per item, one item
// or per-item gw): the caller may then skip zeroing gw and set a.overwrite
bool launch_wgrad_single_writer(const WgradArgs& a);
// scale (optional device scalar): dst = unpacked src * scale[0]
hipError_t launch_unpack_wgrad(const float* src, float* dst, int Cout, int Cin, int taps, bool oihw, hipStream_t s, const float* scale = nullptr);
hipError_t launch_pack_dgrad_weight(const float* src, float* dst, int Cout, int Cin, int KH, int KW, bool oihw, int bk, hipStream_t s);
hipError_t launch_colsum(const float* g, int ld, long long rows, int C, const float* scale, float* out, hipStream_t s);
hipError_t launch_rowsum_per_item(const float* g, int ld, int B, long long n, int C, float* out, hipStream_t s);

// GroupNorm(8)+Mish backward (two passes inside one launcher).  g = gradient w.r.t. the masked activation.
struct GnBwdArgs {
  const float* y; int y_ld;          // saved conv output (pre-norm)
  const double* stats;               // forward (sum, sumsq) [B][8][2]
  const float* gamma; const float* beta;
  const float* g; int g_ld;
  const float* mask; int mask_ld, mask_step, mask_bmod;
  float* gy; int gy_ld;              // gradient w.r.t. y
  float* ggamma; float* gbeta;       // [C], accumulated (zeroed by the caller)
  float* gbias;                      // optional [C]: column sums of gy (bias gradient of the producing conv)
  float* gy_amax;                    // optional zeroed word: receives max |gy| (float bits; launch_wgrad_amax's result without its pass)
  double* gsum;                      // scratch [B][8][2], zeroed by the caller
  int B, H, W, C;
};
hipError_t launch_gn_bwd(const GnBwdArgs& a, hipStream_t s);

hipError_t launch_attn_bwd_gctx(const float* qkv, const float* gO, int B, int n, float* gctx, hipStream_t s);
hipError_t launch_attn_bwd_kv(const float* qkv, const float* ctx, const float* gctx, const float* colM, const float* colS, int B,
                              int n, float* gqkv, hipStream_t s);
hipError_t launch_attn_bwd_wout(const float* M1, const float* ctx, const float* wout, const float* g, int B, int C, float* gwout,
                                float* gg, hipStream_t s);
hipError_t launch_attn_bwd_bias(const float* colsumG, const float* bout, const float* g, int C, float* gbout, float* gg, hipStream_t s);
hipError_t launch_attn_weff_dgrad(const float* ctx, const float* wout, float* dst, int B, int C, int bk, hipStream_t s);
// gb0_partials: final_bwd_blocks(B, H, W, C) * B fp64 partial sums of the bias gradient, item-major (launch_reduce_finalize adds them up)
int final_bwd_blocks(int B, int H, int W, int C);
hipError_t launch_final_bwd(const float* go, const float* h, int ld, const float* w, const float* mask, int mask_ld, int mask_bmod,
                            float* gh, float* gw, double* gb0_partials, int B, int H, int W, int C, hipStream_t s);
// Fixed-order fp64 reductions to one scalar (train.hip): partials[i] = the i-th block's share of sum_{r, c} a[r][c] * b[r][c], i <
// dot_partial_blocks(rows, C) <= kRedBlocks; launch_reduce_finalize: dst[j][0] += scale[j][0] * (sum of job j's partials in index order)
constexpr int kRedBlocks = 256;
int dot_partial_blocks(long long rows, int C);
hipError_t launch_dot_partial(const float* a, int a_ld, const float* b, int b_ld, long long rows, int C, double* partials, hipStream_t s);
struct RedJobs {
  static constexpr int kMax = 12;
  const double* p[kMax]; int n[kMax]; float* dst[kMax]; const float* scale[kMax];     // scale: optional device scalar
  int count;
};
hipError_t launch_reduce_finalize(const RedJobs& jobs, hipStream_t s);
hipError_t launch_first_conv_wgrad(const float* in2, const float* gy, const float* gr, int Bp, int F, int T, int C, float* gw3,
                                   float* gw1, hipStream_t s);
hipError_t launch_add2(const float* a, int a_ld, const float* b, int b_ld, float* out, int out_ld, long long rows, int C, hipStream_t s);
// gW/gb accumulate f(x)-weighted sums; gx (optional, pre-zeroed) accumulates gy*W WITHOUT the mish' factor of a mish-input
// layer: call launch_mul_mish_grad(gx, x) once after all layers sharing that input have been added
hipError_t launch_linear_bwd(const float* gy, int gy_ld, const float* W, const float* x, int x_ld, int rows, int in_dim, int out_dim,
                             bool mish_in, float* gW, float* gb, float* gx, int gx_ld, hipStream_t s);
hipError_t launch_mul_mish_grad(float* g, int g_ld, const float* x, int x_ld, int rows, int n, hipStream_t s);
// The same layers as ONE launch per direction for a set of linear layers that share their input (the 22 time projections of the ResnetBlocks,
// unitspeech.py:61,72): job j owns the output columns [o0, o0 + cout) of the concatenation.  Forward: out_j[r][o] = bias_j[o] + W_j[o] . f(in[r]);
// backward: gW_j / gb_j accumulated from gy_j (ld cout), gx[r][k] += sum_j sum_o gy_j[r][o] W_j[o][k].
struct LinJob {
  const float* W; const float* bias; float* out;       // forward ([cout][in_dim], [cout], [rows][out_ld])
  const float* gy; float* gW; float* gb;               // backward ([rows][cout], [cout][in_dim], [cout])
  int out_ld, o0, cout, pad;
};
hipError_t launch_linear_multi(const LinJob* jobs_dev, int njobs, int total_out, const float* in, int in_ld, int rows, int in_dim, bool mish_in,
                               hipStream_t s);
hipError_t launch_linear_bwd_multi(const LinJob* jobs_dev, int njobs, int total_out, const float* x, int x_ld, int rows, int in_dim,
                                   bool mish_in, float* gx, int gx_ld, hipStream_t s);

}  // namespace us

// ---- Winograd F(2x2, 3x3) for the stride-1 3x3 convolutions of the low-resolution levels (wino.hip) -----------------
// conv3x3 = 16 independent [tiles x Cin] x [Cin x Cout] GEMMs (run by the implicit-GEMM kernel as 1x1 convolutions over
// 16*B "items") between an input transform V = B^T d B and an output transform Y = A^T M A: 2.25x fewer MFMA FLOPs for
// 4x-expanded intermediate tensors, which pays where activations are small next to the weights (levels >= 1).
namespace us {
// U[f][Cout][Cin] = (G g G^T)[f], packed per frequency like a 1x1 conv: dst[f][Cin/bk][Cout][bk]; src Conv2d OIHW 3x3
// dgrad = true: the transform of the 180-degree-rotated, channel-swapped filter (data gradient), dst[16][Cout/bk][Cin][bk]
hipError_t launch_wino_pack_weight(const float* src, float* dst, int Cout, int Cin, int bk, hipStream_t s, bool dgrad = false);
// x: [B][H][W][x_ld] (C channels) -> V: [16][B][th][tw][C], th = ceil(H/2), tw = ceil(W/2); zero padding outside the image
// split = true: V is written as two interleaved fp16 planes per value (same bytes; C % 8 == 0) for the f16x3 GEMM
// (conv_igemm_kernel<.., F16 = true>)
hipError_t launch_wino_input(const float* x, int x_ld, float* V, int B, int H, int W, int C, hipStream_t s, bool split = false);
// V of d = (mish(GroupNorm(y)) * mask + temb) * mask, y a raw conv output [B][H][W][C] (ld = C): block1's gn_apply of a
// ResnetBlock folded into the input transform of its second convolution (C a multiple of 32)
struct WinoGnArgs {
  const double* stats;                 // [B][8][2] sums of y
  const float *gamma, *beta, *temb;    // temb: [B][C] or null
  const float* mask; int mask_ld, mask_step, mask_bmod;
  unsigned* range_flag;                // set by the launcher
  float* h_out;                        // optional [B][H][W][C]: the activation d itself is stored too (training keeps h1 for the weight gradient)
};
bool gn_wino_input_supported(int C);
hipError_t launch_gn_wino_input(const float* y, float* V, int B, int H, int W, int C, const WinoGnArgs& g, hipStream_t s, bool split = false);
// M: [16][B][th][tw][C] -> out[B][H][W][out_ld] = A^T M A + bias; optional GroupNorm partial sums [B][8][2] of the result
// optional epilogue of the separate output transform: out = (Y + bias + add) * mask  (data gradients: residual sum + frame mask)
struct WinoOutExtra {
  const float* add; int add_ld;                          // pixel-indexed like out
  const float* mask; int mask_ld, mask_step, mask_bmod;  // column ox reads mask[ox * mask_step]
};
// nslab > 1: M is nslab split-K slabs of the product, slab_stride floats apart, summed here in slab order (ConvArgs::splitk_raw)
hipError_t launch_wino_output(const float* M, const float* bias, float* out, int out_ld, double* stats, int B, int H, int W, int C,
                              hipStream_t s, const WinoOutExtra* extra = nullptr, int nslab = 1, long long slab_stride = 0);
// f16x3 form of launch_wino_pack_weight (bk = 32): dst holds two interleaved fp16 planes per value, same size and row structure
hipError_t launch_wino_pack_weight_f16(const float* src, float* dst, int Cout, int Cin, hipStream_t s, bool dgrad = false);

// ---- Winograd F(MH x MW, 3x3) with 4-wide tiles (wino4.hip; inference, f16x3 operands only) -----------------------------------------
// form = 10 * MH + MW: 44 = F(4x4,3x3), 36 frequencies, 2.25 Winograd-domain values per pixel; 24 = F(2x4,3x3), 24 frequencies, 3 per pixel
// (F(2x2): 16 frequencies, 4 per pixel).  Tile grid th = ceil(H / MH) x tw = ceil(W / MW).  V: [F][B][th][tw][C] in the two-plane fp16 form;
// U: [F][Cin/32][Cout][32] likewise (src: Conv2d OIHW 3x3); M: [F][B][th][tw][C] fp32.
bool wino4_form_ok(int form);
int wino4_freqs(int form);
void wino4_tiles(int form, int H, int W, int* th, int* tw);
// gn (optional): x is block1's raw conv output (ld == C) and d = (mish(GroupNorm(x)) * mask + temb) * mask is evaluated on the fly (h_out unsupported)
hipError_t launch_wino4_input(int form, const float* x, int x_ld, float* V, int B, int H, int W, int C, const WinoGnArgs* gn, hipStream_t s);
hipError_t launch_wino4_output(int form, const float* M, const float* bias, float* out, int out_ld, double* stats, int B, int H, int W, int C,
                               hipStream_t s);
hipError_t launch_wino4_pack_weight(int form, const float* src, float* dst, int Cout, int Cin, hipStream_t s);
}  // namespace us
