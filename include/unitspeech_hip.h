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
int us_decoder_destroy(us_handle h);

/* `load_state_dict` for one tensor: `key` is the reference state_dict key (e.g.
 * "estimator.downs.0.0.block1.block.0.weight"), `data` a device pointer in the reference's own layout
 * (Conv2d OIHW, ConvTranspose2d IOHW, Linear [out,in]).  The library repacks into its MFMA-friendly layout
 * on `stream`.  Re-loading a key (fine-tuning) is allowed. */
int us_decoder_load_weight(us_handle h, const char* key, const float* data, const int64_t* shape, int ndim,
                           us_stream stream);
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
 *        us_sampler_workspace_bytes(h, min(micro_batch, B), T, n_cfg). */
int us_reverse_diffusion(us_handle h, const float* z, const float* mask, const float* cond, const float* spk,
                         const float* noise, uint64_t seed, int64_t utt_offset, int B, int T, int n_timesteps,
                         float text_gradient_scale, float spk_gradient_scale, const float* coef_host,
                         int micro_batch, float* out, void* workspace, size_t workspace_bytes, us_stream stream);

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
 * (sized by us_train_workspace_bytes), which must stay untouched until us_estimator_backward has been enqueued.
 * us_estimator_backward(grad_out [B, n_feats, T]) writes d loss / d parameter for every `estimator.*` state_dict key
 * into the caller's buffers (reference layout and shape of that key; overwritten, not accumulated).  keys[i]/grads[i]
 * pair a key with its device buffer; all estimator keys must be present.  Input gradients are not produced (the
 * reference never needs them: finetune.py:131-165 optimises decoder parameters only). */
size_t us_train_workspace_bytes(us_handle h, int B, int T);
int us_estimator_forward_train(us_handle h, const float* x, const float* mask, const float* mu, const float* t,
                               const float* spk, float* out, int B, int T, void* workspace, size_t workspace_bytes,
                               us_stream stream);
/* flags: bit 0 = the gradient buffers are already zero (e.g. views of one zero-filled blob): skips 228 fill launches. */
int us_estimator_backward(us_handle h, const float* grad_out, const char* const* keys, float* const* grads, int n_grads,
                          int flags, us_stream stream);

/* Sampled kernel timing for the roofline report.  When enabled, the middle evaluation of every
 * us_reverse_diffusion micro-batch (and every us_estimator_forward) brackets each implicit-GEMM convolution launch,
 * and the evaluation as a whole, with HIP events on the caller's stream.  us_profile_read waits for the recorded
 * events and returns the accumulated totals: time and algorithmic FLOPs (2*MAC) of the conv launches, their
 * count, and the time / count of the sampled evaluations. */
int us_profile_enable(us_handle h, int enable);
int us_profile_read(us_handle h, double* conv_ms, double* conv_flops, int64_t* conv_launches, double* eval_ms,
                    int64_t* evals, int reset);

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
