// Conditioning producer of `execute_text_to_speech`: the text / unit `Encoder`
// (/root/reference/unitspeech/encoder.py:253-308) and the `DurationPredictor` (unitspeech/duration_predictor.py:24-63), inference
// (eval mode: every Dropout is the identity).
//
// The work is tiny next to the diffusion loop (a 6-layer, 192-channel transformer over a few hundred symbols: ~2 GFLOP per
// utterance, once, against 60 TFLOP for the 50 decoder evaluations), so the kernels are plain fp32 FMA code with a fixed
// summation order -- no MFMA, no split reductions -- laid out for coalesced access: activations are kept channel-last
// [B][L][C] (a symbol's channels contiguous), convolution weights are re-packed once to [tap][Cin][Cout].  What matters
// here is that the arithmetic follows the reference statement by statement (mask placement, LayerNorm formula and eps,
// score scaling before the relative term is added, -1e4 fill, two separate value sums), which the goldens of
// tools/make_goldens_frontend.py check.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/unitspeech_hip.h"
#include "kernels.h"

namespace us {
namespace {

// ---- kernels -----------------------------------------------------------------------------------------------------------

// x[b][l][:] = emb[ids[b][l]][:] * scale   (encoder.py:295; an id outside the table gives NaNs rather than a silent clamp)
__global__ void fe_embed_kernel(const long long* ids, const float* emb, float* out, int n_vocab, int C, float scale) {
  const long long row = blockIdx.x;
  const long long id = ids[row];
  const bool ok = id >= 0 && id < n_vocab;
  for (int c = threadIdx.x; c < C; c += blockDim.x)
    out[row * C + c] = ok ? mul_rn(emb[id * C + c], scale) : __builtin_nanf("");
}

// mask[b][l] = l < lengths[b]   (unitspeech/util.py sequence_mask)
__global__ void fe_length_mask_kernel(const long long* lengths, float* mask, int L) {
  const int b = blockIdx.y, l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l < L) mask[(long long)b * L + l] = l < lengths[b] ? 1.f : 0.f;
}

struct Conv1dArgs {
  const float* in;      // [B][L][Cin]
  const float* w;       // [K][Cin][Cout]
  const float* bias;    // [Cout]
  const float* mask;    // [B][L]
  const float* add;     // [B][L][Cout] or null: added before the output mask (prenet: x_org + proj(x))
  float* out;           // [B][L][Cout]
  int L, Cin, Cout, K;
  int mask_in, relu, mask_out;
};
constexpr int kConvTT = 4;       // symbols per workgroup
constexpr int kConvCo = 64;      // output channels per workgroup (one per lane)
constexpr int kConvKg = 8;       // the (tap, ci) range is cut into 8 contiguous parts, one per wave

// torch.nn.Conv1d(padding = K/2) on channel-last rows.  A workgroup owns kConvTT symbols x 64 output channels; its
// (kConvTT + K - 1) x Cin input rows sit in LDS.  Lane = output channel (the weight row [tap][ci][:] is read coalesced and reused
// for the kConvTT symbols), wave = eighth of the (tap, ci) range; the eight partial sums meet in LDS and are added in a fixed
// order ((bias + p0) + p1 + ... + p7), each an fp32 FMA chain in (tap, ci) order: deterministic, batch-independent.
__global__ void __launch_bounds__(64 * kConvKg) fe_conv1d_kernel(Conv1dArgs a) {
  extern __shared__ float xs[];                     // [(kConvTT + K - 1)][Cin], then [kConvKg][kConvTT][64] partial sums
  const int b = blockIdx.y, t0 = blockIdx.x * kConvTT, pad = a.K / 2;
  const int rows = kConvTT + a.K - 1;
  float* part = xs + rows * a.Cin;
  for (int i = threadIdx.x; i < rows * a.Cin; i += blockDim.x) {
    const int r = i / a.Cin, ci = i - r * a.Cin, t = t0 + r - pad;
    float v = 0.f;
    if (t >= 0 && t < a.L) {
      v = a.in[((long long)b * a.L + t) * a.Cin + ci];
      if (a.mask_in) v *= a.mask[(long long)b * a.L + t];
    }
    xs[i] = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, kg = threadIdx.x >> 6;
  const int co = blockIdx.z * kConvCo + lane;
  const int n_red = a.K * a.Cin;                                   // flattened (tap, ci)
  const int lo = (int)(((long long)n_red * kg) / kConvKg), hi = (int)(((long long)n_red * (kg + 1)) / kConvKg);
  float acc[kConvTT];
#pragma unroll
  for (int tt = 0; tt < kConvTT; ++tt) acc[tt] = 0.f;
  if (co < a.Cout) {
    if (kg == 0) {
      const float bv = a.bias[co];
#pragma unroll
      for (int tt = 0; tt < kConvTT; ++tt) acc[tt] = bv;
    }
    // input row of (symbol tt, tap k) is tt + k, so element (tt, k, ci) sits at xs[tt * Cin + r] with r = k * Cin + ci
    const float* wp = a.w + co;
#pragma unroll 8
    for (int r = lo; r < hi; ++r) {
      const float wv = wp[(long long)r * a.Cout];
#pragma unroll
      for (int tt = 0; tt < kConvTT; ++tt) acc[tt] = __builtin_fmaf(xs[tt * a.Cin + r], wv, acc[tt]);
    }
  }
#pragma unroll
  for (int tt = 0; tt < kConvTT; ++tt) part[(kg * kConvTT + tt) * kConvCo + lane] = acc[tt];
  __syncthreads();
  if (kg == 0 && co < a.Cout) {
#pragma unroll
    for (int tt = 0; tt < kConvTT; ++tt) {
      const int t = t0 + tt;
      if (t >= a.L) break;
      float v = acc[tt];
#pragma unroll
      for (int g = 1; g < kConvKg; ++g) v = add_rn(v, part[(g * kConvTT + tt) * kConvCo + lane]);
      const long long o = ((long long)b * a.L + t) * a.Cout + co;
      if (a.relu) v = v > 0.f ? v : 0.f;
      if (a.add) v = add_rn(a.add[o], v);
      if (a.mask_out) v *= a.mask[(long long)b * a.L + t];
      a.out[o] = v;
    }
  }
}

struct LnArgs {
  const float* in;      // [rows][C]
  const float* add;     // [rows][C] or null: in + add is normalised (transformer residual)
  const float* gamma; const float* beta;
  const float* mask;    // [rows] or null: output multiplied by it
  float* out;
  int C; float eps; int relu;
};
constexpr int kLnMaxPerLane = 16;      // C <= 1024

// LayerNorm over the channels of one symbol (encoder.py:21-30: mean, mean of squared deviations, (x - mean) * rsqrt(var + eps)
// * gamma + beta; duration_predictor.py:16-21 is the same arithmetic through F.layer_norm).  One wave per symbol.
__global__ void __launch_bounds__(64) fe_layernorm_kernel(LnArgs a) {
  const long long row = blockIdx.x;
  const int lane = threadIdx.x;
  float v[kLnMaxPerLane];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < kLnMaxPerLane; ++i) {
    const int c = lane + 64 * i;
    v[i] = 0.f;
    if (c < a.C) {
      v[i] = a.in[row * a.C + c];
      if (a.add) v[i] = add_rn(v[i], a.add[row * a.C + c]);
      s += v[i];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / (float)a.C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < kLnMaxPerLane; ++i) {
    const int c = lane + 64 * i;
    if (c < a.C) { const float d = sub_rn(v[i], mean); q = __builtin_fmaf(d, d, q); }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = 1.f / sqrtf(add_rn(q / (float)a.C, a.eps));
  const float m = a.mask ? a.mask[row] : 1.f;
#pragma unroll
  for (int i = 0; i < kLnMaxPerLane; ++i) {
    const int c = lane + 64 * i;
    if (c < a.C) {
      float y = add_rn(mul_rn(mul_rn(sub_rn(v[i], mean), rstd), a.gamma[c]), a.beta[c]);
      if (a.relu) y = y > 0.f ? y : 0.f;
      if (a.mask) y *= m;
      a.out[row * a.C + c] = y;
    }
  }
}

struct AttnArgs {
  const float* q; const float* k; const float* v;     // [B][L][C], head h owns channels [h * D, (h + 1) * D)
  const float* rel_k; const float* rel_v;             // [2W+1][D] (heads_share) or null
  const float* mask;                                  // [B][L]
  float* out;                                         // [B][L][C]
  int L, C, D, W;
  float inv_unused, sqrt_d;
};

// MultiHeadAttention.attention (encoder.py:115-144) for one (query i, head, item):
//   score[j] = q_i.k_j / sqrt(D) + [|j-i| <= W] q_i.rel_k[j-i+W] / sqrt(D);  -1e4 where mask_i * mask_j == 0;  softmax over j;
//   out = sum_j p[j] v_j  +  sum_{|d| <= W, 0 <= i+d < L} p[i+d] rel_v[d+W]
// (the reference reaches the same terms by zero-padding the 2W+1 embeddings to 2L-1 and skewing, :154-182).
__global__ void __launch_bounds__(128) fe_rel_attention_kernel(AttnArgs a) {
  extern __shared__ float sm[];          // p[L], q[D], red[128]
  float* p = sm;
  float* qs = sm + a.L;
  float* red = qs + a.D;
  const int i = blockIdx.x, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  const long long base = (long long)b * a.L * a.C + (long long)h * a.D;
  for (int d = tid; d < a.D; d += blockDim.x) qs[d] = a.q[base + (long long)i * a.C + d];
  __syncthreads();
  const float mi = a.mask[(long long)b * a.L + i];
  float mx = -INFINITY;
  for (int j = tid; j < a.L; j += blockDim.x) {
    const float* kj = a.k + base + (long long)j * a.C;
    float s = 0.f;
    for (int d = 0; d < a.D; ++d) s = __builtin_fmaf(qs[d], kj[d], s);
    s = s / a.sqrt_d;
    const int off = j - i;
    if (a.rel_k && off >= -a.W && off <= a.W) {
      const float* rk = a.rel_k + (long long)(off + a.W) * a.D;
      float r = 0.f;
      for (int d = 0; d < a.D; ++d) r = __builtin_fmaf(qs[d], rk[d], r);
      s = add_rn(s, r / a.sqrt_d);
    }
    if (mi * a.mask[(long long)b * a.L + j] == 0.f) s = -1e4f;
    p[j] = s;
    mx = fmaxf(mx, s);
  }
  red[tid] = mx;
  __syncthreads();
  for (int o = 64; o > 0; o >>= 1) {
    if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]);
    __syncthreads();
  }
  mx = red[0];
  __syncthreads();
  float sum = 0.f;
  for (int j = tid; j < a.L; j += blockDim.x) {
    const float e = expf(p[j] - mx);
    p[j] = e;
    sum += e;
  }
  red[tid] = sum;
  __syncthreads();
  for (int o = 64; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  sum = red[0];
  for (int j = tid; j < a.L; j += blockDim.x) p[j] = p[j] / sum;
  __syncthreads();
  for (int d = tid; d < a.D; d += blockDim.x) {
    float o1 = 0.f;
    for (int j = 0; j < a.L; ++j) o1 = __builtin_fmaf(p[j], a.v[base + (long long)j * a.C + d], o1);
    if (a.rel_v) {
      float o2 = 0.f;
      for (int off = -a.W; off <= a.W; ++off) {
        const int j = i + off;
        if (j >= 0 && j < a.L) o2 = __builtin_fmaf(p[j], a.rel_v[(long long)(off + a.W) * a.D + d], o2);
      }
      o1 = add_rn(o1, o2);
    }
    a.out[base + (long long)i * a.C + d] = o1;
  }
}

// [B][L][C] -> [B][C][L] (the reference's channel-first tensors), optionally times mask[b][l]; and the reverse
__global__ void fe_to_channel_first_kernel(const float* in, const float* mask, float* out, int L, int C) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, l0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    const int l = l0 + r, c = c0 + threadIdx.x;
    tile[r][threadIdx.x] = (l < L && c < C) ? in[((long long)b * L + l) * C + c] : 0.f;
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    const int c = c0 + r, l = l0 + threadIdx.x;
    if (l < L && c < C) {
      float v = tile[threadIdx.x][r];
      if (mask) v *= mask[(long long)b * L + l];
      out[((long long)b * C + c) * L + l] = v;
    }
  }
}

// out[b][l][0:C] = x[b][c][l] (channel-first in), out[b][l][C:C+S] = g[b][s]   (duration_predictor.py:49-50: cat with the repeated g)
__global__ void fe_gather_concat_kernel(const float* x_cf, const float* g, float* out, int L, int C, int S) {
  const int b = blockIdx.y, l = blockIdx.x;
  float* o = out + ((long long)b * L + l) * (C + S);
  for (int c = threadIdx.x; c < C; c += blockDim.x) o[c] = x_cf[((long long)b * C + c) * L + l];
  for (int s = threadIdx.x; s < S; s += blockDim.x) o[C + s] = g[(long long)b * S + s];
}

// w[co][ci][k] (torch Conv1d) -> [k][ci][co]
__global__ void fe_pack_conv_kernel(const float* w, float* out, int Cout, int Cin, int K) {
  const long long n = (long long)Cout * Cin * K;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int co = (int)(i % Cout);
    const long long r = i / Cout;
    const int ci = (int)(r % Cin), k = (int)(r / Cin);
    out[i] = w[((long long)co * Cin + ci) * K + k];
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------------

struct Weight {
  std::vector<int64_t> shape;
  float* dev = nullptr;        // reference layout
  float* packed = nullptr;     // conv weights: [K][Cin][Cout]
  bool loaded = false;
  size_t numel() const { size_t n = 1; for (auto s : shape) n *= (size_t)s; return n; }
};

}  // namespace
}  // namespace us

struct us_frontend {
  int kind = 0;                        // 0: Encoder, 1: DurationPredictor
  us_encoder_config ec{};
  us_duration_config dc{};
  int device = 0;
  std::vector<std::string> keys;       // state_dict order
  std::map<std::string, us::Weight> w;
  std::string err;
};

namespace us {
namespace {

int fe_fail(us_frontend* h, int code, const std::string& msg) {
  if (h) h->err = msg;
  set_last_error(msg.c_str());
  return code;
}
int fe_hip(us_frontend* h, const char* what, hipError_t e) {
  return fe_fail(h, US_EHIP, std::string(what) + ": " + hipGetErrorString(e));
}

void add_key(us_frontend* h, const std::string& k, std::vector<int64_t> shape) {
  h->keys.push_back(k);
  h->w[k].shape = std::move(shape);
}
void add_conv(us_frontend* h, const std::string& p, int cout, int cin, int k) {
  add_key(h, p + ".weight", {cout, cin, k});
  add_key(h, p + ".bias", {cout});
}
void add_norm(us_frontend* h, const std::string& p, int c) {
  add_key(h, p + ".gamma", {c});
  add_key(h, p + ".beta", {c});
}

constexpr int kPrenetLayers = 3, kPrenetKernel = 5;      // encoder.py:283-284

// state_dict of Encoder (module registration order of encoder.py:270-291)
void encoder_keys(us_frontend* h) {
  const auto& c = h->ec;
  const int C = c.n_channels, D = C / c.n_heads;
  add_key(h, "emb.weight", {c.n_vocab, C});
  for (int i = 0; i < kPrenetLayers; ++i) add_conv(h, "prenet.conv_layers." + std::to_string(i), C, C, kPrenetKernel);
  for (int i = 0; i < kPrenetLayers; ++i) add_norm(h, "prenet.norm_layers." + std::to_string(i), C);
  add_conv(h, "prenet.proj", C, C, 1);
  for (int i = 0; i < c.n_layers; ++i) {
    const std::string p = "encoder.attn_layers." + std::to_string(i);
    if (c.window_size > 0) {
      add_key(h, p + ".emb_rel_k", {1, 2 * c.window_size + 1, D});
      add_key(h, p + ".emb_rel_v", {1, 2 * c.window_size + 1, D});
    }
    for (const char* n : {".conv_q", ".conv_k", ".conv_v", ".conv_o"}) add_conv(h, p + n, C, C, 1);
  }
  for (int i = 0; i < c.n_layers; ++i) add_norm(h, "encoder.norm_layers_1." + std::to_string(i), C);
  for (int i = 0; i < c.n_layers; ++i) {
    const std::string p = "encoder.ffn_layers." + std::to_string(i);
    add_conv(h, p + ".conv_1", c.filter_channels, C, c.kernel_size);
    add_conv(h, p + ".conv_2", C, c.filter_channels, c.kernel_size);
  }
  for (int i = 0; i < c.n_layers; ++i) add_norm(h, "encoder.norm_layers_2." + std::to_string(i), C);
  add_conv(h, "proj_m", c.n_feats, C, 1);
}

void duration_keys(us_frontend* h) {
  const auto& c = h->dc;
  add_conv(h, "conv_1", c.filter_channels, c.in_channels + c.spk_emb_dim, c.kernel_size);
  add_norm(h, "norm_1", c.filter_channels);
  add_conv(h, "conv_2", c.filter_channels, c.filter_channels, c.kernel_size);
  add_norm(h, "norm_2", c.filter_channels);
  add_conv(h, "proj", 1, c.filter_channels, 1);
}

// activation scratch of one forward call, in floats (the caller owns it: us_frontend_workspace_bytes)
size_t fe_scratch_floats(const us_frontend* h, long long rows) {
  if (h->kind == 0) return (size_t)rows * (6 * (size_t)h->ec.n_channels + (size_t)h->ec.filter_channels);      // x, x_org/y, q, k, v, a (C each) + h1 (F)
  return (size_t)rows * ((size_t)(h->dc.in_channels + h->dc.spk_emb_dim) + 2 * (size_t)h->dc.filter_channels);
}
// the handle is bound to the device that was current at creation: weights live there, launches go to a stream of that device
int fe_device(us_frontend* h, const char* what) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev != h->device)
    return fe_fail(h, US_EINVAL, std::string(what) + ": the current device (" + std::to_string(dev) + ") is not the handle's (" +
                                     std::to_string(h->device) + ")");
  return US_OK;
}

int fe_ready(us_frontend* h, const char* what) {
  for (const auto& k : h->keys)
    if (!h->w[k].loaded) return fe_fail(h, US_EWEIGHTS, std::string(what) + ": weight '" + k + "' has not been loaded");
  return US_OK;
}

int conv1d(us_frontend* h, hipStream_t s, const std::string& prefix, const float* in, float* out, const float* mask, const float* add,
           int B, int L, bool mask_in, bool relu, bool mask_out) {
  const Weight& w = h->w[prefix + ".weight"];
  Conv1dArgs a{};
  a.in = in; a.w = w.packed; a.bias = h->w[prefix + ".bias"].dev; a.mask = mask; a.add = add; a.out = out;
  a.L = L; a.Cout = (int)w.shape[0]; a.Cin = (int)w.shape[1]; a.K = (int)w.shape[2];
  a.mask_in = mask_in; a.relu = relu; a.mask_out = mask_out;
  const size_t lds = ((size_t)(kConvTT + a.K - 1) * a.Cin + (size_t)kConvKg * kConvTT * kConvCo) * sizeof(float);
  if (lds > 64 * 1024) return fe_fail(h, US_EINVAL, "front-end conv1d: (4 + K - 1) * Cin rows do not fit 64 KB of LDS");
  hipLaunchKernelGGL(fe_conv1d_kernel, dim3((L + kConvTT - 1) / kConvTT, B, (a.Cout + kConvCo - 1) / kConvCo), dim3(64 * kConvKg), lds, s, a);
  return US_OK;
}

int layernorm(us_frontend* h, hipStream_t s, const std::string& prefix, const float* in, const float* add, float* out, const float* mask,
              long long rows, int C, float eps, bool relu) {
  if (C > 64 * kLnMaxPerLane) return fe_fail(h, US_EINVAL, "front-end LayerNorm: more than 1024 channels");
  LnArgs a{};
  a.in = in; a.add = add; a.gamma = h->w[prefix + ".gamma"].dev; a.beta = h->w[prefix + ".beta"].dev; a.mask = mask; a.out = out;
  a.C = C; a.eps = eps; a.relu = relu;
  hipLaunchKernelGGL(fe_layernorm_kernel, dim3((unsigned)rows), dim3(64), 0, s, a);
  return US_OK;
}

void to_channel_first(hipStream_t s, const float* in, const float* mask, float* out, int B, int L, int C) {
  hipLaunchKernelGGL(fe_to_channel_first_kernel, dim3((L + 31) / 32, (C + 31) / 32, B), dim3(32, 8), 0, s, in, mask, out, L, C);
}

}  // namespace
}  // namespace us

extern "C" {

using namespace us;

int us_encoder_create(us_frontend_handle* out, const us_encoder_config* cfg) {
  if (!out || !cfg) return fe_fail(nullptr, US_EINVAL, "us_encoder_create: null argument");
  const auto& c = *cfg;
  if (c.n_vocab <= 0 || c.n_feats <= 0 || c.n_channels <= 0 || c.filter_channels <= 0 || c.n_heads <= 0 || c.n_layers < 0 ||
      c.kernel_size <= 0 || c.kernel_size % 2 == 0 || c.window_size < 0 || c.n_channels % c.n_heads != 0)
    return fe_fail(nullptr, US_EINVAL, "us_encoder_create: bad configuration (odd kernel_size, n_channels divisible by n_heads)");
  if (c.n_channels > 1024 || c.filter_channels > 1024)
    return fe_fail(nullptr, US_EINVAL, "us_encoder_create: more than 1024 channels");
  auto* h = new us_frontend();
  h->kind = 0; h->ec = c;
  (void)hipGetDevice(&h->device);
  encoder_keys(h);
  *out = h;
  return US_OK;
}

int us_duration_predictor_create(us_frontend_handle* out, const us_duration_config* cfg) {
  if (!out || !cfg) return fe_fail(nullptr, US_EINVAL, "us_duration_predictor_create: null argument");
  const auto& c = *cfg;
  if (c.in_channels <= 0 || c.filter_channels <= 0 || c.kernel_size <= 0 || c.kernel_size % 2 == 0 || c.spk_emb_dim < 0 ||
      c.filter_channels > 1024 || c.in_channels + c.spk_emb_dim > 1024)
    return fe_fail(nullptr, US_EINVAL, "us_duration_predictor_create: bad configuration");
  auto* h = new us_frontend();
  h->kind = 1; h->dc = c;
  (void)hipGetDevice(&h->device);
  duration_keys(h);
  *out = h;
  return US_OK;
}

int us_frontend_destroy(us_frontend_handle h) {
  if (!h) return US_OK;
  for (auto& kv : h->w) {
    if (kv.second.dev) (void)hipFree(kv.second.dev);
    if (kv.second.packed) (void)hipFree(kv.second.packed);
  }
  delete h;
  return US_OK;
}

int us_frontend_num_weights(us_frontend_handle h) { return h ? (int)h->keys.size() : 0; }
const char* us_frontend_weight_key(us_frontend_handle h, int i) {
  return (h && i >= 0 && i < (int)h->keys.size()) ? h->keys[i].c_str() : nullptr;
}
const char* us_frontend_last_error(us_frontend_handle h) { return h ? h->err.c_str() : us_last_error(nullptr); }

int us_frontend_load_weight(us_frontend_handle h, const char* key, const float* data, const int64_t* shape, int ndim, us_stream stream) {
  if (!h || !key || !data || !shape) return fe_fail(h, US_EINVAL, "us_frontend_load_weight: null argument");
  auto it = h->w.find(key);
  if (it == h->w.end()) return fe_fail(h, US_ENOKEY, std::string("us_frontend_load_weight: unknown key '") + key + "'");
  Weight& w = it->second;
  bool same = ndim == (int)w.shape.size();
  for (int i = 0; same && i < ndim; ++i) same = shape[i] == w.shape[i];
  if (!same) return fe_fail(h, US_ESHAPE, std::string("us_frontend_load_weight: shape of '") + key + "' does not match the configuration");
  int rcd = fe_device(h, "us_frontend_load_weight");
  if (rcd != US_OK) return rcd;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t n = w.numel();
  hipError_t e;
  if (!w.dev && (e = hipMalloc(&w.dev, n * sizeof(float))) != hipSuccess) return fe_hip(h, "hipMalloc(weight)", e);
  if ((e = hipMemcpyAsync(w.dev, data, n * sizeof(float), hipMemcpyDeviceToDevice, s)) != hipSuccess) return fe_hip(h, "hipMemcpyAsync(weight)", e);
  if (ndim == 3) {       // Conv1d weight
    if (!w.packed && (e = hipMalloc(&w.packed, n * sizeof(float))) != hipSuccess) return fe_hip(h, "hipMalloc(packed weight)", e);
    hipLaunchKernelGGL(fe_pack_conv_kernel, dim3((unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256)), dim3(256), 0, s, w.dev, w.packed,
                       (int)shape[0], (int)shape[1], (int)shape[2]);
    if ((e = hipGetLastError()) != hipSuccess) return fe_hip(h, "fe_pack_conv_kernel", e);
  }
  w.loaded = true;
  return US_OK;
}

size_t us_frontend_workspace_bytes(us_frontend_handle h, int B, int L) {
  if (!h || B <= 0 || L <= 0) return 0;
  return fe_scratch_floats(h, (long long)B * L) * sizeof(float) + 256;
}

int us_encoder_forward(us_frontend_handle h, const int64_t* ids, const int64_t* lengths, float* mu_x, float* x_out, float* x_mask, int B,
                       int L, void* workspace, size_t workspace_bytes, us_stream stream) {
  if (!h || h->kind != 0) return fe_fail(h, US_EINVAL, "us_encoder_forward: not an encoder handle");
  if (!ids || !lengths || !mu_x || !x_out || !x_mask || B <= 0 || L <= 0) return fe_fail(h, US_EINVAL, "us_encoder_forward: bad argument");
  if (B > 65535) return fe_fail(h, US_EINVAL, "us_encoder_forward: more than 65535 items");
  int rc = fe_ready(h, "us_encoder_forward");
  if (rc != US_OK) return rc;
  const auto& c = h->ec;
  const int C = c.n_channels, F = c.filter_channels, D = C / c.n_heads;
  const size_t attn_lds = ((size_t)L + D + 128) * sizeof(float);
  if (attn_lds > 64 * 1024) return fe_fail(h, US_EINVAL, "us_encoder_forward: more than ~16000 symbols per utterance");
  const long long rows = (long long)B * L;
  if ((rc = fe_device(h, "us_encoder_forward")) != US_OK) return rc;
  if (!workspace || workspace_bytes < us_frontend_workspace_bytes(h, B, L))
    return fe_fail(h, US_EWORKSPACE, "us_encoder_forward: workspace too small (us_frontend_workspace_bytes)");
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* x = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~uintptr_t(255));
  float* y = x + rows * C;
  float* q = y + rows * C;
  float* k = q + rows * C;
  float* v = k + rows * C;
  float* at = v + rows * C;
  float* h1 = at + rows * C;
  const float* mask = x_mask;      // [B][1][L] == [B][L]
  hipLaunchKernelGGL(fe_length_mask_kernel, dim3((L + 255) / 256, B), dim3(256), 0, s, reinterpret_cast<const long long*>(lengths), x_mask, L);
  hipLaunchKernelGGL(fe_embed_kernel, dim3((unsigned)rows), dim3(C >= 256 ? 256 : 64), 0, s, reinterpret_cast<const long long*>(ids),
                     h->w["emb.weight"].dev, x, c.n_vocab, C, sqrtf((float)C));
  // prenet (ConvReluNorm.forward, encoder.py:58-65): x_org stays in `x`; q / k ping-pong
  const float* cur = x;
  float* pp[2] = {q, k};
  for (int i = 0; i < kPrenetLayers; ++i) {
    float* t = pp[i & 1];
    if ((rc = conv1d(h, s, "prenet.conv_layers." + std::to_string(i), cur, t, mask, nullptr, B, L, true, false, false)) != US_OK) return rc;
    if ((rc = layernorm(h, s, "prenet.norm_layers." + std::to_string(i), t, nullptr, t, nullptr, rows, C, 1e-4f, true)) != US_OK) return rc;
    cur = t;
  }
  if ((rc = conv1d(h, s, "prenet.proj", cur, y, mask, x, B, L, false, false, true)) != US_OK) return rc;      // (x_org + proj(x)) * x_mask
  { float* t = x; x = y; y = t; }
  // transformer blocks (EncoderModule.forward, :239-250); `x` is masked on entry to every block (LayerNorm 2 writes it masked)
  for (int i = 0; i < c.n_layers; ++i) {
    const std::string ap = "encoder.attn_layers." + std::to_string(i);
    if ((rc = conv1d(h, s, ap + ".conv_q", x, q, mask, nullptr, B, L, false, false, false)) != US_OK) return rc;
    if ((rc = conv1d(h, s, ap + ".conv_k", x, k, mask, nullptr, B, L, false, false, false)) != US_OK) return rc;
    if ((rc = conv1d(h, s, ap + ".conv_v", x, v, mask, nullptr, B, L, false, false, false)) != US_OK) return rc;
    AttnArgs a{};
    a.q = q; a.k = k; a.v = v; a.mask = mask; a.out = at;
    a.rel_k = c.window_size > 0 ? h->w[ap + ".emb_rel_k"].dev : nullptr;
    a.rel_v = c.window_size > 0 ? h->w[ap + ".emb_rel_v"].dev : nullptr;
    a.L = L; a.C = C; a.D = D; a.W = c.window_size; a.sqrt_d = sqrtf((float)D);
    hipLaunchKernelGGL(fe_rel_attention_kernel, dim3(L, c.n_heads, B), dim3(128), attn_lds, s, a);
    if ((rc = conv1d(h, s, ap + ".conv_o", at, y, mask, nullptr, B, L, false, false, false)) != US_OK) return rc;
    if ((rc = layernorm(h, s, "encoder.norm_layers_1." + std::to_string(i), x, y, x, nullptr, rows, C, 1e-4f, false)) != US_OK) return rc;
    const std::string fp = "encoder.ffn_layers." + std::to_string(i);
    if ((rc = conv1d(h, s, fp + ".conv_1", x, h1, mask, nullptr, B, L, true, true, false)) != US_OK) return rc;
    if ((rc = conv1d(h, s, fp + ".conv_2", h1, y, mask, nullptr, B, L, true, false, true)) != US_OK) return rc;
    if ((rc = layernorm(h, s, "encoder.norm_layers_2." + std::to_string(i), x, y, x, mask, rows, C, 1e-4f, false)) != US_OK) return rc;
  }
  if ((rc = conv1d(h, s, "proj_m", x, q, mask, nullptr, B, L, false, false, true)) != US_OK) return rc;
  to_channel_first(s, q, nullptr, mu_x, B, L, c.n_feats);
  to_channel_first(s, x, nullptr, x_out, B, L, C);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? US_OK : fe_hip(h, "us_encoder_forward", e);
}

int us_duration_predictor_forward(us_frontend_handle h, const float* x, const float* x_mask, const float* g, float* logw, int B, int L,
                                  void* workspace, size_t workspace_bytes, us_stream stream) {
  if (!h || h->kind != 1) return fe_fail(h, US_EINVAL, "us_duration_predictor_forward: not a duration-predictor handle");
  const auto& c = h->dc;
  if (!x || !x_mask || !logw || B <= 0 || L <= 0 || B > 65535) return fe_fail(h, US_EINVAL, "us_duration_predictor_forward: bad argument");
  if ((c.spk_emb_dim > 0) != (g != nullptr))
    return fe_fail(h, US_EINVAL, "us_duration_predictor_forward: g must be given exactly when the module was built with spk_emb_dim > 0");
  int rc = fe_ready(h, "us_duration_predictor_forward");
  if (rc != US_OK) return rc;
  const int Cin = c.in_channels + c.spk_emb_dim, F = c.filter_channels;
  const long long rows = (long long)B * L;
  if ((rc = fe_device(h, "us_duration_predictor_forward")) != US_OK) return rc;
  if (!workspace || workspace_bytes < us_frontend_workspace_bytes(h, B, L))
    return fe_fail(h, US_EWORKSPACE, "us_duration_predictor_forward: workspace too small (us_frontend_workspace_bytes)");
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* xin = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~uintptr_t(255));
  float* a1 = xin + rows * Cin;
  float* a2 = a1 + rows * F;
  hipLaunchKernelGGL(fe_gather_concat_kernel, dim3(L, B), dim3(256), 0, s, x, g, xin, L, c.in_channels, c.spk_emb_dim);
  // conv(x * mask) -> relu -> LayerNorm, twice (duration_predictor.py:51-58); proj(x * mask) * mask (:59)
  if ((rc = conv1d(h, s, "conv_1", xin, a1, x_mask, nullptr, B, L, true, true, false)) != US_OK) return rc;
  if ((rc = layernorm(h, s, "norm_1", a1, nullptr, a1, nullptr, rows, F, 1e-5f, false)) != US_OK) return rc;
  if ((rc = conv1d(h, s, "conv_2", a1, a2, x_mask, nullptr, B, L, true, true, false)) != US_OK) return rc;
  if ((rc = layernorm(h, s, "norm_2", a2, nullptr, a2, nullptr, rows, F, 1e-5f, false)) != US_OK) return rc;
  if ((rc = conv1d(h, s, "proj", a2, logw, x_mask, nullptr, B, L, true, false, true)) != US_OK) return rc;      // [B][L][1] == [B][1][L]
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? US_OK : fe_hip(h, "us_duration_predictor_forward", e);
}

}  // extern "C"
