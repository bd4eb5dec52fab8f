// Elementwise / gather kernels on either side of the score network, and their C-ABI entry points (no handle needed):
//   training side  `forward_diffusion` (unitspeech/unitspeech.py:376-384), the score-matching loss of `loss_t` (:393-405) with
//                  its gradient, the segment crop of `fine_tune` (:452-493), the data gradient of the 2-channel first layer
//                  (d loss / d x, d loss / d mu for callers that train an encoder through the decoder, train_STEP1.py:381,
//                  train_STEP2.py:299);
//   inference side the conditioning producer's alignment step of `execute_text_to_speech` (:424-438: durations -> lengths ->
//                  `generate_path` -> attn^T cond_x) and the mel de-normalisation that follows the sampler (inference.py:140).
// All HBM-bound, a few bytes per element; planar [B][F][T] boundary tensors are read / written with coalesced rows.
#include <hip/hip_runtime.h>

#include <cstdio>

#include "../../include/unitspeech_hip.h"
#include "kernels.h"

namespace us {

namespace {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// cumulative noise of `get_noise(t, beta_min, beta_max, cumulative=True)` (:204-209) in the reference's fp32 order:
// beta_min * t + (0.5 * (beta_max - beta_min)) * t^2, the bracket being a Python double rounded to fp32 once
__device__ __forceinline__ float cum_noise(float t, float beta_min, float half_delta) {
  return add_rn(mul_rn(beta_min, t), mul_rn(half_delta, mul_rn(t, t)));
}

// xt = (x0 * exp(-c/2) + z * sqrt(1 - exp(-c))) * mask,  zm = z * mask        (z == null: the x0 term only, i.e. the
// backward of xt w.r.t. x0 applied to a gradient)
__global__ __launch_bounds__(256) void forward_diffusion_kernel(const float* __restrict__ x0, const float* __restrict__ mask,
                                                                const float* __restrict__ t, const float* __restrict__ z,
                                                                float* __restrict__ xt, float* __restrict__ zm, int B, int F, int T,
                                                                float beta_min, float half_delta) {
  const long long FT = (long long)F * T, total = (long long)B * FT;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int b = (int)(i / FT);
    const int tt = (int)(i % T);
    const float c = cum_noise(t[b], beta_min, half_delta);
    const float a = expf(mul_rn(-0.5f, c));
    const float m = mask[(long long)b * T + tt];
    float v = mul_rn(x0[i], a);
    if (z) {
      const float sd = sqrtf(sub_rn(1.f, expf(-c)));
      const float zz = z[i];
      v = add_rn(v, mul_rn(zz, sd));
      zm[i] = mul_rn(zz, m);
    }
    xt[i] = mul_rn(v, m);
  }
}

// loss = sum((score * sigma_b + z)^2) / (sum(mask) * F)   (:403-404);  dscore = d loss / d score.
// Pass 1: every block re-sums the (tiny) mask in a fixed order, writes dscore and one fp64 partial; pass 2 adds the partials
// in block order: the loss is reproducible run to run.
__global__ __launch_bounds__(256) void diffusion_loss_partial_kernel(const float* __restrict__ score, const float* __restrict__ zm,
                                                                     const float* __restrict__ t, const float* __restrict__ mask,
                                                                     float* __restrict__ dscore, double* __restrict__ partial, int B, int F,
                                                                     int T, float beta_min, float half_delta) {
  __shared__ double red[4];
  __shared__ float s_den;
  {
    float ms = 0.f;
    for (int i = threadIdx.x; i < B * T; i += 256) ms += mask[i];
    double w = wsum((double)ms);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) s_den = (float)(red[0] + red[1] + red[2] + red[3]) * (float)F;
    __syncthreads();
  }
  const float den = s_den;
  const long long FT = (long long)F * T, total = (long long)B * FT;
  double acc = 0.0;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int b = (int)(i / FT);
    const float c = cum_noise(t[b], beta_min, half_delta);
    const float sd = sqrtf(sub_rn(1.f, expf(-c)));
    const float r = add_rn(mul_rn(score[i], sd), zm[i]);
    acc += (double)mul_rn(r, r);
    if (dscore) dscore[i] = 2.f * r * sd / den;
  }
  acc = wsum(acc);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    if (blockIdx.x == 0) partial[gridDim.x] = (double)den;
  }
}
__global__ __launch_bounds__(64) void diffusion_loss_final_kernel(const double* __restrict__ partial, int n, float* __restrict__ loss) {
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) acc += partial[i];
  acc = wsum(acc);
  if (threadIdx.x == 0) loss[0] = (float)acc / (float)partial[n];
}

// out[0] = 2^k with k such that max|x| * 2^k lies in [2^(target-1), 2^target), out[1] = 2^-k; 1 and 1 for an all-zero or non-finite x.
// One workgroup: the operand is a gradient tensor of a few hundred thousand elements, read once more by the caller's scaling pass.
__global__ __launch_bounds__(1024) void pow2_scale_kernel(const float* __restrict__ x, long long n, int target_log2, float* __restrict__ out) {
  __shared__ float red[16];
  float mx = 0.f;
  for (long long i = threadIdx.x; i < n; i += 1024) mx = fmaxf(mx, fabsf(x[i]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w) mx = fmaxf(mx, red[w]);
    int k = 0;
    const unsigned bits = __float_as_uint(mx);
    const int e = (int)((bits >> 23) & 255u);
    if (mx > 0.f && e != 255 && e != 0) k = (target_log2 - 1) - (e - 127);      // floor(log2 mx) = e - 127 for a normal float
    k = k < -60 ? -60 : (k > 60 ? 60 : k);
    out[0] = __uint_as_float((unsigned)(127 + k) << 23);
    out[1] = __uint_as_float((unsigned)(127 - k) << 23);
  }
}

__global__ void scale_by_scalar_kernel(const float* __restrict__ x, const float* __restrict__ s, float* __restrict__ out, long long n) {
  const float k = s[0];
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) out[i] = x[i] * k;
}

// Data gradient of the first ResnetBlock's two 2-channel convolutions (3x3 block1 conv on gy, 1x1 res_conv on gr) back to the
// planar inputs: gin[c][p] = mask * (sum_{tap,co} gy[p - d(tap)][co] * w3[co][c][tap] + sum_co gr[p][co] * w1[co][c]),
// c = 0: mu, c = 1: x (`torch.stack([mu, x], 1)`, :170; block1 and res_conv both see `x * mask`, :54,:74).
// One wave per output pixel, lanes over the C output channels (coalesced rows), weights in LDS as [tap][c][co].
__global__ __launch_bounds__(256) void first_conv_dgrad_kernel(const float* __restrict__ gy, const float* __restrict__ gr,
                                                               const float* __restrict__ w3, const float* __restrict__ w1,
                                                               const float* __restrict__ mask, float* __restrict__ gmu,
                                                               float* __restrict__ gx, int B, int F, int T, int C) {
  extern __shared__ float sw[];         // [9][2][C] + [2][C]
  for (int i = threadIdx.x; i < 18 * C; i += 256) {
    const int co = i % C, c = (i / C) % 2, tap = i / (2 * C);
    sw[i] = w3[(co * 2 + c) * 9 + tap];
  }
  for (int i = threadIdx.x; i < 2 * C; i += 256) sw[18 * C + i] = w1[(i % C) * 2 + (i / C)];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const long long npix = (long long)B * F * T;
  for (long long p = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); p < npix; p += (long long)gridDim.x * 4) {
    const int b = (int)(p / ((long long)F * T));
    const int rem = (int)(p - (long long)b * F * T);
    const int f = rem / T, tt = rem - f * T;
    float a0 = 0.f, a1 = 0.f;
    // forward: y[q] += w3[.][c][ky][kx] * in[q + (ky-1, kx-1)]  =>  gin[p] += gy[p - (ky-1, kx-1)] * w3[.][c][ky][kx]
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int qf = f - (ky - 1);
      if ((unsigned)qf >= (unsigned)F) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int qt = tt - (kx - 1);
        if ((unsigned)qt >= (unsigned)T) continue;
        const float* row = gy + (((long long)b * F + qf) * T + qt) * C;
        const float* w = sw + (ky * 3 + kx) * 2 * C;
        for (int co = lane; co < C; co += 64) {
          const float g = row[co];
          a0 = fmaf(g, w[co], a0);
          a1 = fmaf(g, w[C + co], a1);
        }
      }
    }
    {
      const float* row = gr + p * C;
      const float* w = sw + 18 * C;
      for (int co = lane; co < C; co += 64) {
        const float g = row[co];
        a0 = fmaf(g, w[co], a0);
        a1 = fmaf(g, w[C + co], a1);
      }
    }
    a0 = wsum(a0);
    a1 = wsum(a1);
    if (lane == 0) {
      const float m = mask[(long long)b * T + tt];
      if (gmu) gmu[p] = a0 * m;
      if (gx) gx[p] = a1 * m;
    }
  }
}

// ---- conditioning producer: durations -> lengths, path, aligned conditioning ------------------------------------------
// w_ceil = ceil(exp(logw) * x_mask) * length_scale (:424-425), y_lengths = clamp_min(sum(w_ceil), 1).long() (:427)
__global__ __launch_bounds__(64) void tts_durations_kernel(const float* __restrict__ logw, const float* __restrict__ x_mask,
                                                           float* __restrict__ w_ceil, long long* __restrict__ y_lengths, int L,
                                                           float length_scale) {
  const int b = blockIdx.x;
  float acc = 0.f;
  for (int l = threadIdx.x; l < L; l += 64) {
    const float w = mul_rn(ceilf(mul_rn(expf(logw[(long long)b * L + l]), x_mask[(long long)b * L + l])), length_scale);
    w_ceil[(long long)b * L + l] = w;
    acc += w;                       // whole frame counts: exact in fp32 in any order (below 2^24)
  }
  acc = wsum(acc);
  if (threadIdx.x == 0) y_lengths[b] = (long long)fmaxf(acc, 1.f);
}

// `generate_path` (unitspeech/util.py:27-40) + `attn^T cond_x` (:437-438) + `sequence_mask` (:431): frame t of item b belongs to
// the symbol l with cum[l-1] <= t < cum[l] (cum = running sum of w_ceil); frames at or beyond y_lengths[b], and symbols masked out
// by x_mask, produce zeros.  One block per (item, 256-frame stretch); the running sum is rebuilt per block in LDS (L <= a few
// hundred), then every thread binary-searches its frame.  attn (optional) is [B][L][Tp], written densely (zeros included).
__global__ __launch_bounds__(256) void tts_align_kernel(const float* __restrict__ cond_x, const float* __restrict__ w_ceil,
                                                        const float* __restrict__ x_mask, const long long* __restrict__ y_lengths,
                                                        float* __restrict__ cond_y, float* __restrict__ attn, float* __restrict__ y_mask,
                                                        int F, int L, int Tp) {
  extern __shared__ float cum[];      // [L]
  const int b = blockIdx.y;
  if (threadIdx.x == 0) {
    float run = 0.f;
    for (int l = 0; l < L; ++l) { run += w_ceil[(long long)b * L + l]; cum[l] = run; }     // torch.cumsum order
  }
  __syncthreads();
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= Tp) return;
  const bool live = (long long)t < y_lengths[b];
  // first l with cum[l] > t
  int lo = 0, hi = L;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (cum[mid] > (float)t) hi = mid; else lo = mid + 1;
  }
  const int sym = (live && lo < L && x_mask[(long long)b * L + lo] != 0.f) ? lo : -1;
  if (y_mask) y_mask[(long long)b * Tp + t] = live ? 1.f : 0.f;
  for (int f = 0; f < F; ++f)
    cond_y[((long long)b * F + f) * Tp + t] = sym >= 0 ? cond_x[((long long)b * F + f) * L + sym] : 0.f;
  if (attn)
    for (int l = 0; l < L; ++l) attn[((long long)b * L + l) * Tp + t] = (l == sym) ? 1.f : 0.f;
}

// `fine_tune` crop (:458-486): y_cut[b][:, j] = y[b][:, start_b + j], cond_y[b][:, j] = (sum_l attn[b][l][start_b + j] *
// cond_x[b][:, l]) * mask, mask[b][j] = j < n_b, for j < segment; zeros beyond n_b.  attn is the caller's dense [B][Lu][Ly] matrix.
__global__ __launch_bounds__(256) void finetune_segment_kernel(const float* __restrict__ cond_x, const float* __restrict__ y,
                                                               const float* __restrict__ attn, const long long* __restrict__ start,
                                                               const long long* __restrict__ count, float* __restrict__ y_cut,
                                                               float* __restrict__ cond_y, float* __restrict__ seg_mask, int B, int F,
                                                               int Lu, int Ly, int seg) {
  const long long total = (long long)B * F * seg;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int j = (int)(i % seg);
    const int f = (int)((i / seg) % F);
    const int b = (int)(i / ((long long)seg * F));
    const long long src = start[b] + j;
    const bool live = j < count[b] && src >= 0 && src < Ly;      // (a window reaching past y reads zeros, never out of bounds)
    float yc = 0.f, cy = 0.f;
    if (live) {
      yc = y[((long long)b * F + f) * Ly + src];
      // four chains in flight (one was 200 dependent load pairs: 80 us for a 176-frame crop); with a hard alignment -- one 1 per frame,
      // what generate_path and MAS produce -- every order of the sum gives the same bits
      const float* ap = attn + (long long)b * Lu * Ly + src;
      const float* cp = cond_x + ((long long)b * F + f) * Lu;
      float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
      int l = 0;
#pragma unroll 2
      for (; l + 4 <= Lu; l += 4) {
        c0 = fmaf(ap[(long long)l * Ly], cp[l], c0);
        c1 = fmaf(ap[(long long)(l + 1) * Ly], cp[l + 1], c1);
        c2 = fmaf(ap[(long long)(l + 2) * Ly], cp[l + 2], c2);
        c3 = fmaf(ap[(long long)(l + 3) * Ly], cp[l + 3], c3);
      }
      for (; l < Lu; ++l) c0 = fmaf(ap[(long long)l * Ly], cp[l], c0);
      cy = (c0 + c1) + (c2 + c3);
    }
    y_cut[i] = yc;
    cond_y[i] = cy;
    if (f == 0) seg_mask[(long long)b * seg + j] = live ? 1.f : 0.f;
  }
}

// mel = (x * mask + 1) / 2 * (mel_max - mel_min) + mel_min in the reference's fp32 operation order (inference.py:140 applied to
// the sampler's final `xt * mask`, :373); mel_range == null: the mask multiply only
__global__ void finish_mel_kernel(const float* __restrict__ x, const float* __restrict__ mask, float* __restrict__ out, int B, int F, int T,
                                  int denorm, float mel_min, float mel_span) {
  const long long total = (long long)B * F * T;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(i % T);
    const int b = (int)(i / ((long long)F * T));
    float v = mul_rn(x[i], mask[(long long)b * T + t]);
    if (denorm) v = add_rn(mul_rn(mul_rn(add_rn(v, 1.f), 0.5f), mel_span), mel_min);
    out[i] = v;
  }
}

inline int nblocks(long long total, int cap = 2048) {
  long long b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

hipError_t launch_finish_mel(const float* x, const float* mask, float* out, int B, int F, int T, const float* mel_range_host, hipStream_t s) {
  const int denorm = mel_range_host ? 1 : 0;
  const float mn = denorm ? mel_range_host[0] : 0.f;
  const float span = denorm ? (mel_range_host[1] - mel_range_host[0]) : 0.f;     // fp32 subtraction, as `mel_max - mel_min` on fp32 tensors
  hipLaunchKernelGGL(finish_mel_kernel, dim3(nblocks((long long)B * F * T)), dim3(256), 0, s, x, mask, out, B, F, T, denorm, mn, span);
  return hipGetLastError();
}

hipError_t launch_first_conv_dgrad(const float* gy, const float* gr, const float* w3, const float* w1, const float* mask, float* gmu,
                                   float* gx, int B, int F, int T, int C, hipStream_t s) {
  const long long npix = (long long)B * F * T;
  long long blocks = (npix + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(first_conv_dgrad_kernel, dim3((int)blocks), dim3(256), (size_t)20 * C * sizeof(float), s, gy, gr, w3, w1, mask, gmu, gx,
                     B, F, T, C);
  return hipGetLastError();
}

hipError_t launch_pow2_scale(const float* x, long long n, int target_log2, float* out2, hipStream_t s) {
  hipLaunchKernelGGL(pow2_scale_kernel, dim3(1), dim3(1024), 0, s, x, n, target_log2, out2);
  return hipGetLastError();
}

hipError_t launch_scale(const float* x, const float* scalar_dev, float* out, long long n, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(scale_by_scalar_kernel, dim3(nblocks(n)), dim3(256), 0, s, x, scalar_dev, out, n);
  return hipGetLastError();
}

}  // namespace us

using namespace us;

namespace {
int fail(const char* what, hipError_t e) {
  char buf[256];
  snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
  us::set_last_error(buf);
  return US_EHIP;
}
int bad(const char* msg) {
  us::set_last_error(msg);
  return US_EINVAL;
}
inline float half_delta(float beta_min, float beta_max) { return (float)(0.5 * ((double)beta_max - (double)beta_min)); }
}  // namespace

extern "C" {

int us_forward_diffusion(const float* x0, const float* mask, const float* t, const float* z, float* xt, float* z_masked, int B, int F, int T,
                         float beta_min, float beta_max, us_stream stream) {
  if (!x0 || !mask || !t || !xt || (z && !z_masked) || B <= 0 || F <= 0 || T <= 0) return bad("us_forward_diffusion: bad argument");
  hipLaunchKernelGGL(forward_diffusion_kernel, dim3(nblocks((long long)B * F * T)), dim3(256), 0, static_cast<hipStream_t>(stream), x0,
                     mask, t, z, xt, z_masked, B, F, T, beta_min, half_delta(beta_min, beta_max));
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? US_OK : fail("us_forward_diffusion", e);
}

size_t us_diffusion_loss_scratch_bytes(int B, int F, int T) {
  return (size_t)(nblocks((long long)B * F * T, 512) + 1) * sizeof(double);
}

int us_diffusion_loss(const float* score, const float* z_masked, const float* t, const float* mask, float* loss, float* dscore, int B, int F,
                      int T, float beta_min, float beta_max, void* scratch, size_t scratch_bytes, us_stream stream) {
  if (!score || !z_masked || !t || !mask || !loss || !scratch || B <= 0 || F <= 0 || T <= 0) return bad("us_diffusion_loss: bad argument");
  if (scratch_bytes < us_diffusion_loss_scratch_bytes(B, F, T)) return bad("us_diffusion_loss: scratch too small");
  const int nb = nblocks((long long)B * F * T, 512);
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(diffusion_loss_partial_kernel, dim3(nb), dim3(256), 0, s, score, z_masked, t, mask, dscore, static_cast<double*>(scratch),
                     B, F, T, beta_min, half_delta(beta_min, beta_max));
  hipLaunchKernelGGL(diffusion_loss_final_kernel, dim3(1), dim3(64), 0, s, static_cast<const double*>(scratch), nb, loss);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? US_OK : fail("us_diffusion_loss", e);
}

int us_scale(const float* x, const float* scalar_dev, float* out, size_t n, us_stream stream) {
  if (!x || !scalar_dev || !out) return bad("us_scale: null argument");
  if (n == 0) return US_OK;
  hipLaunchKernelGGL(scale_by_scalar_kernel, dim3(nblocks((long long)n)), dim3(256), 0, static_cast<hipStream_t>(stream), x, scalar_dev, out,
                     (long long)n);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? US_OK : fail("us_scale", e);
}

int us_pow2_scale(const float* x, size_t n, int target_log2, float* scale_and_inverse, us_stream stream) {
  if (!x || !scale_and_inverse || n == 0 || target_log2 < -100 || target_log2 > 100) return bad("us_pow2_scale: bad argument");
  hipLaunchKernelGGL(pow2_scale_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), x, (long long)n, target_log2, scale_and_inverse);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? US_OK : fail("us_pow2_scale", e);
}

int us_mul_mask(const float* x, const float* mask, float* out, int B, int F, int T, us_stream stream) {
  if (!x || !mask || !out || B <= 0 || F <= 0 || T <= 0) return bad("us_mul_mask: bad argument");
  hipError_t e = launch_mul_mask(x, mask, out, B, F, T, static_cast<hipStream_t>(stream));
  return e == hipSuccess ? US_OK : fail("us_mul_mask", e);
}

int us_finetune_segment(const float* cond_x, const float* y, const float* attn, const int64_t* start, const int64_t* count, float* y_cut,
                        float* cond_y, float* seg_mask, int B, int F, int Lu, int Ly, int segment_size, us_stream stream) {
  if (!cond_x || !y || !attn || !start || !count || !y_cut || !cond_y || !seg_mask || B <= 0 || F <= 0 || Lu <= 0 || Ly <= 0 || segment_size <= 0)
    return bad("us_finetune_segment: bad argument");
  hipLaunchKernelGGL(finetune_segment_kernel, dim3(nblocks((long long)B * F * segment_size)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     cond_x, y, attn, reinterpret_cast<const long long*>(start), reinterpret_cast<const long long*>(count), y_cut, cond_y,
                     seg_mask, B, F, Lu, Ly, segment_size);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? US_OK : fail("us_finetune_segment", e);
}

int us_tts_durations(const float* logw, const float* x_mask, float* w_ceil, int64_t* y_lengths, int B, int L, float length_scale,
                     us_stream stream) {
  if (!logw || !x_mask || !w_ceil || !y_lengths || B <= 0 || L <= 0) return bad("us_tts_durations: bad argument");
  hipLaunchKernelGGL(tts_durations_kernel, dim3(B), dim3(64), 0, static_cast<hipStream_t>(stream), logw, x_mask, w_ceil,
                     reinterpret_cast<long long*>(y_lengths), L, length_scale);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? US_OK : fail("us_tts_durations", e);
}

int us_tts_align(const float* cond_x, const float* w_ceil, const float* x_mask, const int64_t* y_lengths, float* cond_y, float* attn,
                 float* y_mask, int B, int F, int L, int Tp, us_stream stream) {
  if (!cond_x || !w_ceil || !x_mask || !y_lengths || !cond_y || B <= 0 || F <= 0 || L <= 0 || Tp <= 0) return bad("us_tts_align: bad argument");
  if ((size_t)L * sizeof(float) > 60000) return bad("us_tts_align: more than 15000 symbols per utterance");
  hipLaunchKernelGGL(tts_align_kernel, dim3((Tp + 255) / 256, B), dim3(256), (size_t)L * sizeof(float), static_cast<hipStream_t>(stream),
                     cond_x, w_ceil, x_mask, reinterpret_cast<const long long*>(y_lengths), cond_y, attn, y_mask, F, L, Tp);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? US_OK : fail("us_tts_align", e);
}

}  // extern "C"
