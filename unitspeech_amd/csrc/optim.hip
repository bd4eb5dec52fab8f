// Gradient clipping + Adam for the fine-tune / pre-training loop (finetune.py:163-165: loss.backward();
// torch.nn.utils.clip_grad_norm_(decoder.parameters(), max_norm=1); optimizer.step() with torch.optim.Adam(lr=2e-5)), as
// three launches over ALL parameter tensors through a device-side pointer table instead of torch's ~10 multi-tensor passes:
//   1. per-block sum of squares of the gradients   2. fixed-order reduction -> total norm, clip coefficient
//   3. g *= coef (as clip_grad_norm_ does, in place), Adam moments and parameter update in torch's operation order.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "../../include/unitspeech_hip.h"

namespace {

constexpr int kChunk = 4096;      // elements per block (16 per thread)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void sumsq_kernel(const float* const* __restrict__ g, const int64_t* __restrict__ numel,
                                                    const int32_t* __restrict__ blk_tensor, const int64_t* __restrict__ blk_off,
                                                    float* __restrict__ partial) {
  __shared__ float red[4];
  const int t = blk_tensor[blockIdx.x];
  const int64_t off = blk_off[blockIdx.x];
  const float* gp = g[t] + off;
  int64_t n = numel[t] - off;
  if (n > kChunk) n = kChunk;
  float acc = 0.f;
  if ((reinterpret_cast<uintptr_t>(gp) & 15) == 0) {
    const int n4 = (int)(n >> 2);
    for (int i = threadIdx.x; i < n4; i += 256) {
      f32x4 v = reinterpret_cast<const f32x4*>(gp)[i];
      acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    for (int i = (n4 << 2) + threadIdx.x; i < n; i += 256) acc += gp[i] * gp[i];
  } else {
    for (int i = threadIdx.x; i < n; i += 256) acc += gp[i] * gp[i];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// partial[n_blocks] = total L2 norm, partial[n_blocks + 1] = clip coefficient min(1, max_norm / (norm + 1e-6))
__global__ __launch_bounds__(1024) void norm_finish_kernel(float* __restrict__ partial, int n_blocks, float max_norm) {
  __shared__ double red[1024];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n_blocks; i += 1024) acc += (double)partial[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(red[0]);
    float coef = max_norm / (norm + 1e-6f);
    if (coef > 1.f) coef = 1.f;
    partial[n_blocks] = norm;
    partial[n_blocks + 1] = coef;
  }
}

struct AdamConst { float lr, beta1, beta2, omb1, omb2, eps, step_size, bc2_sqrt; };   // omb = 1 - beta, rounded from double like torch's scalars

__device__ __forceinline__ void adam_one(float& p, float& g, float& m, float& v, float coef, const AdamConst& c) {
  g = g * coef;                                             // clip_grad_norm_: grad.mul_(clip_coef)
  m = m + c.omb1 * (g - m);                                 // exp_avg.lerp_(grad, 1 - beta1)
  v = v * c.beta2;                                          // exp_avg_sq.mul_(beta2)
  v = v + (c.omb2 * g) * g;                                 //           .addcmul_(grad, grad, value = 1 - beta2)
  const float denom = sqrtf(v) / c.bc2_sqrt + c.eps;        // (exp_avg_sq.sqrt() / sqrt(bias_correction2)).add_(eps)
  p = p + (-c.step_size) * (m / denom);                     // param.addcdiv_(exp_avg, denom, value = -lr / bias_correction1)
}

__global__ __launch_bounds__(256) void adam_kernel(float* const* __restrict__ p, float* const* __restrict__ g, float* const* __restrict__ m,
                                                   float* const* __restrict__ v, const int64_t* __restrict__ numel,
                                                   const int32_t* __restrict__ blk_tensor, const int64_t* __restrict__ blk_off,
                                                   const float* __restrict__ coef_ptr, AdamConst c) {
  const int t = blk_tensor[blockIdx.x];
  const int64_t off = blk_off[blockIdx.x];
  float* pp = p[t] + off;
  float* gp = g[t] + off;
  float* mp = m[t] + off;
  float* vp = v[t] + off;
  int64_t n = numel[t] - off;
  if (n > kChunk) n = kChunk;
  const float coef = coef_ptr ? *coef_ptr : 1.f;
  const bool al = ((reinterpret_cast<uintptr_t>(pp) | reinterpret_cast<uintptr_t>(gp) | reinterpret_cast<uintptr_t>(mp) |
                    reinterpret_cast<uintptr_t>(vp)) & 15) == 0;
  int done = 0;
  if (al) {
    const int n4 = (int)(n >> 2);
    for (int i = threadIdx.x; i < n4; i += 256) {
      f32x4 P = reinterpret_cast<f32x4*>(pp)[i], G = reinterpret_cast<f32x4*>(gp)[i];
      f32x4 M = reinterpret_cast<f32x4*>(mp)[i], V = reinterpret_cast<f32x4*>(vp)[i];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float pk = P[k], gk = G[k], mk = M[k], vk = V[k];
        adam_one(pk, gk, mk, vk, coef, c);
        P[k] = pk; G[k] = gk; M[k] = mk; V[k] = vk;
      }
      reinterpret_cast<f32x4*>(pp)[i] = P;
      reinterpret_cast<f32x4*>(gp)[i] = G;
      reinterpret_cast<f32x4*>(mp)[i] = M;
      reinterpret_cast<f32x4*>(vp)[i] = V;
    }
    done = n4 << 2;
  }
  for (int i = done + threadIdx.x; i < n; i += 256) adam_one(pp[i], gp[i], mp[i], vp[i], coef, c);
}

}  // namespace

extern "C" int us_clip_adam_step(void* const* p, void* const* g, void* const* m, void* const* v, const int64_t* numel,
                                 const int32_t* blk_tensor, const int64_t* blk_off, int n_tensors, int n_blocks, double lr_d, double beta1_d,
                                 double beta2_d, double eps_d, int step, float max_norm, float* partial, us_stream stream) {
  const float lr = (float)lr_d, beta1 = (float)beta1_d, beta2 = (float)beta2_d, eps = (float)eps_d;
  if (!p || !g || !m || !v || !numel || !blk_tensor || !blk_off || !partial || n_tensors <= 0 || n_blocks <= 0 || step < 1) return US_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const float* coef = nullptr;
  if (max_norm > 0.f) {
    hipLaunchKernelGGL(sumsq_kernel, dim3(n_blocks), dim3(256), 0, s, reinterpret_cast<const float* const*>(g), numel, blk_tensor, blk_off,
                       partial);
    hipLaunchKernelGGL(norm_finish_kernel, dim3(1), dim3(1024), 0, s, partial, n_blocks, max_norm);
    coef = partial + n_blocks + 1;
  }
  AdamConst c;
  c.lr = lr; c.beta1 = beta1; c.beta2 = beta2; c.eps = eps;
  c.omb1 = (float)(1.0 - beta1_d); c.omb2 = (float)(1.0 - beta2_d);
  // torch.optim.adam._single_tensor_adam: python-double scalars, applied to fp32 tensors
  const double bc1 = 1.0 - pow(beta1_d, (double)step), bc2 = 1.0 - pow(beta2_d, (double)step);
  c.step_size = (float)(lr_d / bc1);
  c.bc2_sqrt = (float)sqrt(bc2);
  hipLaunchKernelGGL(adam_kernel, dim3(n_blocks), dim3(256), 0, s, reinterpret_cast<float* const*>(p), reinterpret_cast<float* const*>(g),
                     reinterpret_cast<float* const*>(m), reinterpret_cast<float* const*>(v), numel, blk_tensor, blk_off, coef, c);
  return hipGetLastError() == hipSuccess ? US_OK : US_EHIP;
}
