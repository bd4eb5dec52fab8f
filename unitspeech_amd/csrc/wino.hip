// Winograd F(2x2, 3x3) transforms (Lavin & Gray) for the 3x3 / stride 1 / pad 1 convolutions of `Block`
// (unitspeech/unitspeech.py:48).  With d a 4x4 input tile (tiles step by 2 pixels), g a 3x3 filter:
//   Y(2x2) = A^T [ (G g G^T) .* (B^T d B) ] A
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
// The 16 element-wise products summed over Cin are 16 GEMMs, executed by conv_igemm_kernel.  All transforms are exact
// up to fp32 rounding of sums of at most 4 terms with coefficients in {0, +-1, +-1/2}.
#include "kernels.h"

namespace us {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// one thread = one (tile, channel quad); grid (blocks, B)
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, int x_ld, float* __restrict__ V, int B, int H, int W,
                                                         int C) {
  const int C4 = C >> 2;
  const int th = (H + 1) >> 1, tw = (W + 1) >> 1;
  const int b = blockIdx.y;
  const long long per_item = (long long)th * tw * C4;
  const long long plane = (long long)B * th * tw * C;          // floats per frequency
  const float* xb = x + (long long)b * H * W * x_ld;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < per_item; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4;
    long long t = i / C4;
    const int tx = (int)(t % tw);
    const int ty = (int)(t / tw);
    f32x4 d[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int iy = 2 * ty - 1 + r;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ix = 2 * tx - 1 + q;
        const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        d[r][q] = ok ? *reinterpret_cast<const f32x4*>(xb + ((long long)iy * W + ix) * x_ld + c) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    // rows: t = B^T d
    f32x4 tt[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      tt[0][q] = d[0][q] - d[2][q];
      tt[1][q] = d[1][q] + d[2][q];
      tt[2][q] = d[2][q] - d[1][q];
      tt[3][q] = d[1][q] - d[3][q];
    }
    float* vb = V + (((long long)b * th + ty) * tw + tx) * C + c;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      // columns: v = t B
      f32x4 v0 = tt[r][0] - tt[r][2], v1 = tt[r][1] + tt[r][2], v2 = tt[r][2] - tt[r][1], v3 = tt[r][1] - tt[r][3];
      *reinterpret_cast<f32x4*>(vb + (long long)(r * 4 + 0) * plane) = v0;
      *reinterpret_cast<f32x4*>(vb + (long long)(r * 4 + 1) * plane) = v1;
      *reinterpret_cast<f32x4*>(vb + (long long)(r * 4 + 2) * plane) = v2;
      *reinterpret_cast<f32x4*>(vb + (long long)(r * 4 + 3) * plane) = v3;
    }
  }
}

hipError_t launch_wino_input(const float* x, int x_ld, float* V, int B, int H, int W, int C, hipStream_t s) {
  if (C % 4 != 0 || x_ld % 4 != 0) return hipErrorInvalidValue;
  const long long per_item = (long long)((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
  int blocks = (int)((per_item + 255) / 256);
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(wino_input_kernel, dim3(blocks, B), dim3(256), 0, s, x, x_ld, V, B, H, W, C);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ M, const float* __restrict__ bias, float* __restrict__ out,
                                                          int out_ld, double* __restrict__ stats, int B, int H, int W, int C, WinoOutExtra x) {
  __shared__ double s_g[kGroups][2];
  const int C4 = C >> 2;
  const int th = (H + 1) >> 1, tw = (W + 1) >> 1;
  const int b = blockIdx.y;
  const long long per_item = (long long)th * tw * C4;
  const long long plane = (long long)B * th * tw * C;
  const int cg = C / kGroups;
  if (threadIdx.x < kGroups * 2) s_g[threadIdx.x >> 1][threadIdx.x & 1] = 0.0;
  __syncthreads();
  // a thread keeps one channel quad (grid stride is a multiple of C/4 for the power-of-two widths of the U-Net)
  // fp64 per-thread partials (see conv_igemm_kernel<.., true>: both forms must agree to fp64 rounding)
  double t1[4] = {0, 0, 0, 0}, t2[4] = {0, 0, 0, 0};
  int tc = -1;
  auto flush = [&]() {
    if (tc >= 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        atomicAdd(&s_g[(tc + k) / cg][0], t1[k]);
        atomicAdd(&s_g[(tc + k) / cg][1], t2[k]);
        t1[k] = t2[k] = 0;
      }
    }
  };
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < per_item; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4;
    long long t = i / C4;
    const int tx = (int)(t % tw);
    const int ty = (int)(t / tw);
    const float* mb = M + (((long long)b * th + ty) * tw + tx) * C + c;
    f32x4 m[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) m[r][q] = *reinterpret_cast<const f32x4*>(mb + (long long)(r * 4 + q) * plane);
    // Y = A^T M A, accumulated frequency by frequency in the order f = 0..15 with coefficients At[r][f/4] * At[q][f%4]
    // (At = [1 1 1 0; 0 1 -1 -1]) -- the same sequence of additions as the fused form inside conv_igemm_kernel<.., true>,
    // so a result does not depend on which form a launch geometry selects
    f32x4 y[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int fi = 0; fi < 4; ++fi)
#pragma unroll
          for (int fj = 0; fj < 4; ++fj) {
            const int ci = r == 0 ? (fi < 3 ? 1 : 0) : (fi == 0 ? 0 : (fi == 1 ? 1 : -1));
            const int cj = q == 0 ? (fj < 3 ? 1 : 0) : (fj == 0 ? 0 : (fj == 1 ? 1 : -1));
            if (ci * cj == 1) acc += m[fi][fj];
            if (ci * cj == -1) acc -= m[fi][fj];
          }
        y[r][q] = acc;
      }
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) bv = *reinterpret_cast<const f32x4*>(bias + c);
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int q = 0; q < 2; ++q) y[r][q] += bv;
    if (stats && tc != c) { flush(); tc = c; }
    float* ob = out + (long long)b * H * W * out_ld + c;
    const float* ab = x.add ? x.add + (long long)b * H * W * x.add_ld + c : nullptr;
    const float* mk = x.mask ? x.mask + (long long)(b % x.mask_bmod) * x.mask_ld : nullptr;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int oy = 2 * ty + r, ox = 2 * tx + q;
        if (oy < H && ox < W) {
          f32x4 v = y[r][q];                   // statistics (forward only) are of the plain conv output
          if (ab) v += *reinterpret_cast<const f32x4*>(ab + ((long long)oy * W + ox) * x.add_ld);
          if (mk) v *= mk[ox * x.mask_step];
          *reinterpret_cast<f32x4*>(ob + ((long long)oy * W + ox) * out_ld) = v;
          if (stats) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { t1[k] += (double)y[r][q][k]; t2[k] += (double)(y[r][q][k] * y[r][q][k]); }
          }
        }
      }
  }
  if (stats) {
    flush();
    __syncthreads();
    if (threadIdx.x < kGroups * 2)
      atomicAdd(&stats[((long long)b * kGroups + (threadIdx.x >> 1)) * 2 + (threadIdx.x & 1)], s_g[threadIdx.x >> 1][threadIdx.x & 1]);
  }
}

hipError_t launch_wino_output(const float* M, const float* bias, float* out, int out_ld, double* stats, int B, int H, int W, int C,
                              hipStream_t s, const WinoOutExtra* extra) {
  if (C % 4 != 0 || out_ld % 4 != 0 || C % kGroups != 0 || (extra && extra->add && extra->add_ld % 4 != 0)) return hipErrorInvalidValue;
  WinoOutExtra x{};
  if (extra) x = *extra;
  if (x.mask_bmod < 1) x.mask_bmod = 1;
  const long long per_item = (long long)((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
  int blocks = (int)((per_item + 255) / 256);
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(wino_output_kernel, dim3(blocks, B), dim3(256), 0, s, M, bias, out, out_ld, stats, B, H, W, C, x);
  return hipGetLastError();
}

// U = G g G^T per (co, ci); forward: dst[f][ci/bk][co][bk].  dgrad: the data gradient of a 3x3 / stride 1 / pad 1 conv is the
// conv of the output gradient with the filter rotated by 180 degrees and the channel roles swapped, so
// g'[ky][kx] = g[2-ky][2-kx], GEMM-K = co, GEMM-N = ci: dst[f][co/bk][ci][bk]
__global__ void wino_pack_weight_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cout, int Cin, int bk, int dgrad) {
  const long long total = (long long)Cout * Cin;
  const int nchunk = (dgrad ? Cout : Cin) / bk;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Cin), co = (int)(i / Cin);
    const float* gs = src + i * 9;
    float g[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) g[k] = dgrad ? gs[8 - k] : gs[k];
    float gg[4][3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      gg[0][q] = g[q];
      gg[1][q] = 0.5f * (g[q] + g[3 + q] + g[6 + q]);
      gg[2][q] = 0.5f * (g[q] - g[3 + q] + g[6 + q]);
      gg[3][q] = g[6 + q];
    }
    const long long base = dgrad ? ((long long)(co / bk) * Cin + ci) * bk + co % bk : ((long long)(ci / bk) * Cout + co) * bk + ci % bk;
    const long long fstride = (long long)nchunk * (dgrad ? Cin : Cout) * bk;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float u0 = gg[r][0], u1 = 0.5f * (gg[r][0] + gg[r][1] + gg[r][2]), u2 = 0.5f * (gg[r][0] - gg[r][1] + gg[r][2]), u3 = gg[r][2];
      dst[base + (r * 4 + 0) * fstride] = u0;
      dst[base + (r * 4 + 1) * fstride] = u1;
      dst[base + (r * 4 + 2) * fstride] = u2;
      dst[base + (r * 4 + 3) * fstride] = u3;
    }
  }
}

hipError_t launch_wino_pack_weight(const float* src, float* dst, int Cout, int Cin, int bk, hipStream_t s, bool dgrad) {
  if ((dgrad ? Cout : Cin) % bk != 0) return hipErrorInvalidValue;
  long long total = (long long)Cout * Cin;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(wino_pack_weight_kernel, dim3(blocks), dim3(256), 0, s, src, dst, Cout, Cin, bk, dgrad ? 1 : 0);
  return hipGetLastError();
}

}  // namespace us
