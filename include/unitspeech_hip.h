/*
 * unitspeech_hip.h -- C ABI of the MI355X-native UnitSpeech diffusion decoder (libunitspeech_hip.so).
 *
 * The reference has no plugin/FFI layer: its hot path sits behind the Python class API of
 * `unitspeech/unitspeech.py`.  Each entry point below names the reference method it
 * replaces.  Conventions: every function returns 0 on success or a negative US_E* code and never throws;
 * all tensor pointers are DEVICE pointers to contiguous fp32 unless marked "host"; the caller owns every
 * buffer; work is enqueued on the given hipStream_t and the library does not synchronise; a handle is bound
 * to the device that was current at creation and is not thread-safe.
 */
#ifndef UNITSPEECH_HIP_H
#define UNITSPEECH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct us_decoder* us_handle;
typedef void* us_stream;              /* hipStream_t */

enum {
  US_OK = 0,
  US_EINVAL = -1,      /* bad argument / unsupported shape */
  US_ENOKEY = -2,      /* unknown state_dict key */
  US_ESHAPE = -3,      /* tensor shape does not match the configured architecture */
  US_EWEIGHTS = -4,    /* forward called before every weight was loaded */
  US_EWORKSPACE = -5,  /* workspace too small */
  US_EHIP = -6         /* a HIP runtime call failed (see us_last_error) */
};

/* Constructor arguments of `UnitSpeech.__init__` (unitspeech/unitspeech.py:221) /
 * `GradLogPEstimator2d.__init__` (:125).  heads=4, dim_head=32, groups=8 are fixed by the reference
 * (:79, :47).  dim must be a multiple of 16; n_mults <= 6. */
typedef struct us_config {
  int32_t n_feats;      /* 80 */
  int32_t dim;          /* 128 */
  int32_t n_mults;      /* 4 */
  int32_t dim_mults[6]; /* 1,2,4,8 */
  int32_t spk_emb_dim;  /* 256 */
  float beta_min;       /* 0.05 */
  float beta_max;       /* 20.0 */
  float pe_scale;       /* 1000 */
} us_config;

/* UnitSpeech(...) constructor.  Allocates the packed device weight store (not the weights' values). */
int us_decoder_create(us_handle* out, const us_config* cfg);
/* The same with creation flags.  US_CREATE_EXACT_FP32: every GEMM of this handle runs on the exact-fp32 matrix instruction
 * (v_mfma_f32_32x32x2_f32, a k-ordered fmaf chain) instead of the f16x3 form (three fp16 MFMA products of two-plane split operands,
 * fp32-accurate for operands inside the fp16 range): the path for tensors beyond +-65504, see us_range_status. */
enum { US_CREATE_EXACT_FP32 = 1 };
int us_decoder_create_ex(us_handle* out, const us_config* cfg, unsigned flags);
int us_decoder_destroy(us_handle h);

/* f16x3 operand range.  The reference computes in plain fp32 (unitspeech/unitspeech.py:46-96); the default handle forms its GEMM
 * products from fp16 planes, which represent |x| < 65520.  A larger operand is never clamped: it becomes an infinity (the affected
 * outputs are non-finite, as loud as an overflow can be; a NaN input stays a NaN, as in the reference) and the split that met it
 * ORs a bit into a per-handle device word: US_RANGE_ACT for an activation, Winograd-domain value or gradient, US_RANGE_WEIGHT for a
 * weight at load time (or a folded attention weight).  us_range_status copies the word to *status (0 = every result since the last
 * reset is fp32-accurate), optionally clears it, and WAITS for `stream`; the _async form only enqueues the copy into the caller's
 * (pinned) host word.  A caller that sees a non-zero status repeats the call on a US_CREATE_EXACT_FP32 handle -- the Python mirror
 * does so by itself (unitspeech_amd/unitspeech.py: _Engine.range_status, UnitSpeech._run_checked). */
enum { US_RANGE_ACT = 1, US_RANGE_WEIGHT = 2 };
int us_range_status(us_handle h, unsigned* status, int reset, us_stream stream);
int us_range_status_async(us_handle h, unsigned* status_host, int reset, us_stream stream);

/* `load_state_dict` for one tensor: `key` is the reference state_dict key (e.g.
 * "estimator.downs.0.0.block1.block.0.weight"), `data` a device pointer in the reference's own layout
 * (Conv2d OIHW, ConvTranspose2d IOHW, Linear [out,in]).  The library repacks into its MFMA-friendly layout
 * on `stream`.  Re-loading a key (fine-tuning) is allowed. */
int us_decoder_load_weight(us_handle h, const char* key, const float* data, const int64_t* shape, int ndim,
                           us_stream stream);
/* Completes a batch of us_decoder_load_weight calls: tensors the library keeps in the reference's own layout (biases,
 * GroupNorm affine, MLP weights, the unconditional embeddings) are copied, and the split-precision packs of the convolution
 * weights are written, by ONE table-driven launch each, enqueued here: every `data` pointer handed to us_decoder_load_weight
 * must stay valid until this call has been made on the same stream.  Every computing entry point refuses to run (US_EWEIGHTS)
 * while loads are pending. */
int us_decoder_flush_weights(us_handle h, us_stream stream);
/* Training mode of the weight store.  Inference runs the stride-1 3x3 convolutions of the low-resolution levels as Winograd F(4x4,3x3) /
 * F(2x4,3x3) where enabled (US_WINO4; csrc/wino4.hip), from a third pack of those weights (36 / 24 matrices per convolution) that the
 * training path never reads.  training != 0: us_decoder_load_weight skips that pack -- an optimiser step re-loads every tensor -- and
 * marks it stale; inference calls then run the F(2x2,3x3) form for such a tensor (correct, slower, rounding of that form) until it is
 * loaded again with training == 0.  us_decoder_stale_inference_forms: how many loaded tensors are in that state (a host mirror
 * re-loads them before its first inference call after training: unitspeech_amd/unitspeech.py, _Engine.sync_weights). */
int us_decoder_set_training(us_handle h, int training);
int us_decoder_stale_inference_forms(us_handle h);
/* Number of state_dict tensors the configured architecture has / that have been loaded so far. */
int us_decoder_num_weights(us_handle h);
int us_decoder_num_loaded(us_handle h);
/* Name of the i-th expected key (state_dict order); NULL when out of range. */
const char* us_decoder_weight_key(us_handle h, int i);

/* Scratch bytes needed by one us_estimator_forward call on Bp items of T frames. */
size_t us_workspace_bytes(us_handle h, int Bp, int T);
/* Scratch bytes needed by us_reverse_diffusion for a micro-batch of `mb` utterances (n_cfg branches each). */
size_t us_sampler_workspace_bytes(us_handle h, int mb, int T, int n_cfg);

/* `GradLogPEstimator2d.forward(x, mask, mu, t, spk_emb)` (unitspeech/unitspeech.py:164-201).
 * x, mu, out: [Bp, n_feats, T]; mask: [Bp, 1, T] (0/1); t: [Bp]; spk: [Bp, 1, spk_emb_dim].  T % 2^(n_mults-1) == 0. */
int us_estimator_forward(us_handle h, const float* x, const float* mask, const float* mu, const float* t,
                         const float* spk, float* out, int Bp, int T, void* workspace, size_t workspace_bytes,
                         us_stream stream);

/* `UnitSpeech.forward` == `reverse_diffusion(z, mask, cond, spk_emb, n_timesteps, text_gradient_scale,
 * spk_gradient_scale)` (unitspeech/unitspeech.py:333-391), for any B with per-item B=1 semantics.
 * z, cond, out: [B, n_feats, T]; mask: [B,1,T]; spk: [B,1,spk_emb_dim].
 * noise: [N, B, n_feats, T] explicit gaussian draws replacing `torch.randn` at :367, or NULL to use the
 *        built-in counter-based generator keyed by (seed, utterance index + utt_offset, step).
 * coef_host: optional HOST table [N][8] of per-step scalars (see us_step_coefficients); NULL = computed
 *        by the library.
 * micro_batch: utterances processed together (0 = library default); workspace must hold
 *        us_sampler_workspace_bytes(h, min(micro_batch, B), T, n_cfg).
 * mel_range_host: NULL, or HOST {mel_min, mel_max}: the caller's next step, the mel de-normalisation
 *        `(y + 1) / 2 * (mel_max - mel_min) + mel_min` (inference.py:140) that feeds the vocoder, is applied in the
 *        sampler's last pass (same fp32 operation order), so `out` is the vocoder's input [B, n_feats, T]. */
int us_reverse_diffusion(us_handle h, const float* z, const float* mask, const float* cond, const float* spk,
                         const float* noise, uint64_t seed, int64_t utt_offset, int B, int T, int n_timesteps,
                         float text_gradient_scale, float spk_gradient_scale, const float* coef_host,
                         int micro_batch, const float* mel_range_host, float* out, void* workspace, size_t workspace_bytes,
                         us_stream stream);

/* Host helper: the per-step scalars the sampler update consumes, [N][8] fp32:
 * {sqrt_recip_acp, sqrt_recipm1_acp*sqrt_1m_acp, sqrt(acp_prev), sqrt(1-acp_prev-sigma^2), sqrt_1m_acp,
 *  [idx!=0]*sigma, t_i, 0} for i = 0..N-1 (`register_beta` :235-271, `p_mean_variance` :273-296). */
int us_step_coefficients(int n_timesteps, float beta_min, float beta_max, float* coef_host);

/* Fill out[n] with N(0,1) draws of the built-in generator (Philox4x32-10 + Box-Muller), stream (seed, key). */
int us_fill_normal(float* out, size_t n, uint64_t seed, uint64_t key, us_stream stream);

/* FLOPs (2*MAC of conv + attention einsums + MLPs) of one estimator evaluation per item. */
double us_estimator_flops(us_handle h, int T);

/* ---- training (fine-tune) path: `loss_t` forward + `loss.backward()` through the score network -------------------
 * us_estimator_forward_train == us_estimator_forward, but every tensor the backward needs is kept inside `workspace`
 * (sized by us_train_workspace_bytes), which must stay untouched until us_estimator_backward has been enqueued or the
 * tape has been released.  *tape_id names this forward's record; several may be live at once (each in its own
 * workspace; beyond 16 the oldest is dropped).
 * us_estimator_backward(tape_id, grad_out [B, n_feats, T]) writes d loss / d parameter for every `estimator.*` state_dict
 * key into the caller's buffers (reference layout and shape of that key; overwritten, not accumulated).  keys[i]/grads[i]
 * pair a key with its device buffer; all estimator keys must be present.  B, T must be the forward's.  Fails with
 * US_EINVAL when tape_id is not live (consumed, released or evicted) -- a backward never runs on another forward's tape.
 * grad_x, grad_mu [B, n_feats, T], grad_spk [B, spk_emb_dim]: optional (NULL = not wanted) gradients w.r.t. the inputs
 * x, mu and spk_emb (the reference's trainers reach the text / unit encoder through them: train_STEP1.py:381,
 * train_STEP2.py:299).  The tape is consumed by the call (unless US_BACKWARD_KEEP_TAPE, below).  Outside a stream capture the call
 * runs the weight-gradient launches on a second stream of the handle, fenced against `stream` by events on both sides: when it
 * returns, everything it enqueued is ordered before whatever the caller enqueues on `stream` next (US_WGRAD_STREAM=0: one stream). */
size_t us_train_workspace_bytes(us_handle h, int B, int T);
int us_estimator_forward_train(us_handle h, const float* x, const float* mask, const float* mu, const float* t,
                               const float* spk, float* out, int B, int T, void* workspace, size_t workspace_bytes,
                               uint64_t* tape_id, us_stream stream);
/* flags: US_BACKWARD_GRADS_ZEROED (bit 0) = the gradient buffers are already zero (e.g. views of one zero-filled blob): skips 228
 * fill launches.  US_BACKWARD_KEEP_TAPE (bit 1) = the tape stays live after a successful call: for a caller that re-runs the SAME forward
 * launches into the SAME workspace itself (a captured HIP graph of the forward, replayed per iteration) and then calls the backward
 * again -- the record names buffers, not values.  Release it with us_tape_release.
 * Range of grad_out: any.  The backward GEMMs split their fp32 operands into two fp16 planes, which carry full
 * precision from about 6e-5 upwards, while the gradient of a mean-reduced loss over B*F*T elements is ~1/(B*F*T) and a caller's loss
 * weight or accumulation factor comes on top: the entry point itself multiplies grad_out by the power of two that brings its largest
 * magnitude to [2^-7, 2^-6) (chosen on the device from the data), runs the pass, and multiplies everything it returns by the inverse
 * -- exact, because the pass is linear in grad_out.  grad_out is not modified. */
enum { US_BACKWARD_GRADS_ZEROED = 1, US_BACKWARD_KEEP_TAPE = 2 };
int us_estimator_backward(us_handle h, uint64_t tape_id, const float* grad_out, int B, int T, const char* const* keys,
                          float* const* grads, int n_grads, int flags, float* grad_x, float* grad_mu, float* grad_spk,
                          us_stream stream);
/* Drop a tape whose backward will never run (its workspace may then be reused). */
int us_tape_release(us_handle h, uint64_t tape_id);
/* 1 when us_estimator_backward OVERWRITES every element of this key's gradient buffer (the convolution weights: 99.9 % of the gradient
 * bytes), so the buffer needs no zero-fill under US_BACKWARD_GRADS_ZEROED; 0 when the backward accumulates into it; -1: unknown key. */
int us_grad_is_overwritten(us_handle h, const char* key);

/* ---- elementwise steps of the training objective (no handle) -------------------------------------------------------
 * `forward_diffusion(x0, mask, t)` (unitspeech/unitspeech.py:376-384) with the gaussian draw z passed in:
 *   xt = (x0 * exp(-c/2) + z * sqrt(1 - exp(-c))) * mask, z_masked = z * mask, c = beta_min*t + (beta_max-beta_min)/2*t^2.
 *   z == NULL: xt = x0 * exp(-c/2) * mask only (the backward of xt w.r.t. x0, applied to a gradient).
 * `loss_t`'s objective (:403-404): loss[0] = sum((score * sqrt(1 - exp(-c)) + z_masked)^2) / (sum(mask) * F); dscore
 *   (optional) = d loss / d score.  scratch: us_diffusion_loss_scratch_bytes(B, F, T) bytes. */
int us_forward_diffusion(const float* x0, const float* mask, const float* t, const float* z, float* xt, float* z_masked,
                         int B, int F, int T, float beta_min, float beta_max, us_stream stream);
size_t us_diffusion_loss_scratch_bytes(int B, int F, int T);
int us_diffusion_loss(const float* score, const float* z_masked, const float* t, const float* mask, float* loss,
                      float* dscore, int B, int F, int T, float beta_min, float beta_max, void* scratch,
                      size_t scratch_bytes, us_stream stream);
/* out[i] = x[i] * scalar_dev[0] (chain rule with a device-resident upstream gradient); out = x * mask[b][t] on [B,F,T]. */
int us_scale(const float* x, const float* scalar_dev, float* out, size_t n, us_stream stream);
int us_mul_mask(const float* x, const float* mask, float* out, int B, int F, int T, us_stream stream);
/* scale_and_inverse[0] = 2^k with max|x| * 2^k in [2^(target_log2 - 1), 2^target_log2), [1] = 2^-k (device floats; 1, 1 for an all-zero
 * or non-finite x): an exact, data-driven scaling factor (what us_estimator_backward applies to its grad_out internally); feed [0] and
 * [1] to us_scale. */
int us_pow2_scale(const float* x, size_t n, int target_log2, float* scale_and_inverse, us_stream stream);
/* `fine_tune`'s segment crop (:458-486).  cond_x [B,F,Lu], y [B,F,Ly], attn [B,Lu,Ly]; start/count: DEVICE int64 [B]
 * (crop offset and number of valid frames min(y_length, segment_size) per item).  Writes y_cut, cond_y [B,F,segment_size]
 * (cond_y = attn_cut^T cond_x, masked) and seg_mask [B,segment_size]. */
int us_finetune_segment(const float* cond_x, const float* y, const float* attn, const int64_t* start, const int64_t* count,
                        float* y_cut, float* cond_y, float* seg_mask, int B, int F, int Lu, int Ly, int segment_size,
                        us_stream stream);

/* ---- conditioning producer of `execute_text_to_speech` (:424-438; the text encoder and duration predictor stay the
 * caller's modules) -----------------------------------------------------------------------------------------------------
 * us_tts_durations: w_ceil[B,L] = ceil(exp(logw) * x_mask) * length_scale, y_lengths[B] (int64) = max(sum_l w_ceil, 1).
 * us_tts_align: `generate_path` (unitspeech/util.py:27-40) + `attn^T cond_x` + `sequence_mask`: cond_y [B,F,Tp] (frame t
 *   takes the column of the symbol whose duration interval contains t; zeros at t >= y_lengths[b]), y_mask [B,Tp]
 *   (optional), attn [B,L,Tp] 0/1 (optional).  cond_x [B,F,L]. */
int us_tts_durations(const float* logw, const float* x_mask, float* w_ceil, int64_t* y_lengths, int B, int L,
                     float length_scale, us_stream stream);
int us_tts_align(const float* cond_x, const float* w_ceil, const float* x_mask, const int64_t* y_lengths, float* cond_y,
                 float* attn, float* y_mask, int B, int F, int L, int Tp, us_stream stream);

/* ---- the two learned modules of the conditioning producer, inference -------------------------------
 * `Encoder` (unitspeech/encoder.py:253-308; text encoder and unit encoder are two instances) and `DurationPredictor`
 * (unitspeech/duration_predictor.py:24-63, reverse=True).  Same conventions as the decoder handle: the caller owns the
 * activation scratch (us_frontend_workspace_bytes), nothing is allocated or freed by a forward call, and a call made while
 * another device than the handle's is current is refused (US_EINVAL).  Dropout is the identity (eval mode); `n_contentvec > 0`
 * (encoder.py:281: a Linear instead of the Embedding) and `heads_share=False` are not built -- no configuration of the
 * reference uses them (conf/hydra_config.py:85-116). */
typedef struct us_frontend* us_frontend_handle;
typedef struct us_encoder_config {
  int32_t n_vocab;          /* len(symbols) + 1 (text) / n_units (unit encoder) */
  int32_t n_feats;          /* 80 */
  int32_t n_channels;       /* 192 */
  int32_t filter_channels;  /* 768 */
  int32_t n_heads;          /* 2 */
  int32_t n_layers;         /* 6 */
  int32_t kernel_size;      /* 3 (FFN convolutions; the prenet's 3 layers of kernel 5 are fixed, encoder.py:283) */
  int32_t window_size;      /* 4; 0 = no relative-position terms (window_size=None) */
} us_encoder_config;
typedef struct us_duration_config {
  int32_t in_channels;      /* 192 */
  int32_t filter_channels;  /* 256 */
  int32_t kernel_size;      /* 3 */
  int32_t spk_emb_dim;      /* 256; 0 = no speaker conditioning (g = None) */
} us_duration_config;
int us_encoder_create(us_frontend_handle* out, const us_encoder_config* cfg);
int us_duration_predictor_create(us_frontend_handle* out, const us_duration_config* cfg);
int us_frontend_destroy(us_frontend_handle h);
/* `load_state_dict` for one tensor, reference key and layout (Conv1d [out,in,k], Embedding [vocab,channels], emb_rel_* [1,2W+1,D]). */
int us_frontend_load_weight(us_frontend_handle h, const char* key, const float* data, const int64_t* shape, int ndim, us_stream stream);
int us_frontend_num_weights(us_frontend_handle h);
const char* us_frontend_weight_key(us_frontend_handle h, int i);
const char* us_frontend_last_error(us_frontend_handle h);
/* `Encoder.forward(x, x_lengths)` (:294-308): ids [B,L] int64, lengths [B] int64 (device) ->
 * mu_x [B,n_feats,L], x [B,n_channels,L], x_mask [B,1,L]. */
size_t us_frontend_workspace_bytes(us_frontend_handle h, int B, int L);
int us_encoder_forward(us_frontend_handle h, const int64_t* ids, const int64_t* lengths, float* mu_x, float* x, float* x_mask, int B, int L,
                       void* workspace, size_t workspace_bytes, us_stream stream);
/* `DurationPredictor.forward(x, x_mask, w=None, g=g, reverse=True)` (:47-63): x [B,in_channels,L], x_mask [B,1,L],
 * g [B,1,spk_emb_dim] (NULL iff spk_emb_dim == 0) -> logw [B,1,L]. */
int us_duration_predictor_forward(us_frontend_handle h, const float* x, const float* x_mask, const float* g, float* logw, int B, int L,
                                  void* workspace, size_t workspace_bytes, us_stream stream);

/* ---- one building block of the score network on its own (parity tests against per-module reference outputs) ---------
 * prefix: the module's state_dict prefix ("estimator.downs.1.1", "estimator.downs.1.2", "estimator.downs.1.3",
 * "estimator.ups.1.3"); level: resolution level whose geometry (n_feats >> level) x (T >> level) the block runs at.
 * x / out: pixel-major [B][H][W][C] (the library's internal activation layout), x already multiplied by the frame mask
 * where the reference masks the operand; mask: full-resolution [B][T]; temb: [B][dim + spk_emb_dim] (ResnetBlock only).
 * Output masking follows the library's producer-side convention (attention / Downsample / Upsample outputs are stored
 * masked): pass an all-ones mask to obtain the reference module's raw output.
 * US_DEBUG_TEMB (prefix ignored): x = t [B], out [B][2 * dim] = SinusoidalPosEmb(t) | mlp(SinusoidalPosEmb(t))
 * (unitspeech/unitspeech.py:109-121, 133-134, 165-166). */
enum { US_DEBUG_BLOCK = 0, US_DEBUG_RESNET = 1, US_DEBUG_ATTENTION = 2, US_DEBUG_DOWN = 3, US_DEBUG_UP = 4, US_DEBUG_TEMB = 5 };
int us_debug_block(us_handle h, int kind, const char* prefix, int level, const float* x, const float* mask,
                   const float* temb, float* out, int B, int T, void* workspace, size_t workspace_bytes, us_stream stream);

/* Sampled kernel timing for the roofline report.  When enabled, the middle evaluation of every
 * us_reverse_diffusion micro-batch (and every us_estimator_forward) brackets each implicit-GEMM convolution launch,
 * and the evaluation as a whole, with HIP events on the caller's stream.  us_profile_read waits for the recorded
 * events and returns the accumulated totals: time and algorithmic FLOPs (2*MAC) of the conv launches, their
 * count, and the time / count of the sampled evaluations. */
int us_profile_enable(us_handle h, int enable);
int us_profile_read(us_handle h, double* conv_ms, double* conv_flops, int64_t* conv_launches, double* eval_ms,
                    int64_t* evals, int reset);
/* The share of those conv launches that ran as f16x3 GEMMs (three fp16 MFMA products of two-plane split operands at fp32
 * accuracy): their time, their fp32-equivalent FLOPs (2*M*N*K; the fp16 matrix cores execute three times that) and their
 * count, as accumulated by the last us_profile_read (call it first, with reset = 0). */
int us_profile_read_f16(us_handle h, double* f16_ms, double* f16_flops, int64_t* f16_launches);

/* Gradient clipping + Adam over all parameter tensors in three launches.  Replaces torch.nn.utils.clip_grad_norm_(params, max_norm)
 * followed by torch.optim.Adam(lr, betas, eps, weight_decay=0).step()  (reference finetune.py:163-165, train_STEP1.py).
 *   p, g, m, v  device arrays [n_tensors] of device pointers: parameter, gradient, exp_avg, exp_avg_sq (fp32, same numel)
 *   numel       device [n_tensors]
 *   blk_tensor, blk_off  device [n_blocks]: block i works on elements [blk_off[i], blk_off[i] + 4096) of tensor blk_tensor[i]
 *   lr, beta1, beta2, eps  python-double hyper-parameters; 1 - beta, lr / (1 - beta1^step), sqrt(1 - beta2^step) are evaluated
 *               in double and rounded to fp32 once, as torch.optim.Adam's scalar arguments are
 *   step        1-based Adam step
 *   max_norm    > 0: total L2 norm over all gradients, g *= min(1, max_norm / (norm + 1e-6)) in place first; <= 0: no clipping
 *   partial     device scratch, n_blocks + 2 floats; on return partial[n_blocks] = total norm, partial[n_blocks+1] = coefficient
 * Enqueues on `stream`, never synchronises. */
int us_clip_adam_step(void* const* p, void* const* g, void* const* m, void* const* v, const int64_t* numel,
                      const int32_t* blk_tensor, const int64_t* blk_off, int n_tensors, int n_blocks, double lr, double beta1,
                      double beta2, double eps, int step, float max_norm, float* partial, us_stream stream);

/* Last error message of this handle (or of the library when h == NULL). */
const char* us_last_error(us_handle h);

#ifdef __cplusplus
}
#endif
#endif /* UNITSPEECH_HIP_H */
