// Winograd F(2x2, 3x3) transforms (Lavin & Gray) for the 3x3 / stride 1 / pad 1 convolutions of `Block`
// (unitspeech/unitspeech.py:48).  With d a 4x4 input tile (tiles step by 2 pixels), g a 3x3 filter:
//   Y(2x2) = A^T [ (G g G^T) .* (B^T d B) ] A
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
// The 16 element-wise products summed over Cin are 16 GEMMs, executed by conv_igemm_kernel.  All transforms are exact
// up to fp32 rounding of sums of at most 4 terms with coefficients in {0, +-1, +-1/2}.
#include "kernels.h"
#include "pack_f16.h"

namespace us {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- f16x3 storage of V (conv_igemm_kernel<.., F16 = true>) -----------------------------------------------------------------
// SPLIT = true: every value is kept as two fp16 planes, v ~= hi + lo * 2^-11 with hi = fp16(v), lo = fp16((v - hi) * 2^11), interleaved
// per group of 8 channels as [8 x hi][8 x lo] so that a pixel row keeps its fp32 byte length (C * 4).  `idx` is the fp32 element
// index of a channel quad (c % 4 == 0): its hi quad sits at half index 2 * (idx - idx % 8) + idx % 8, its lo quad 8 halves further.
// A Winograd-domain value beyond the fp16 range (|v| >= 65520; V = B^T d B reaches 4x the activation) is reported through `over`
// (kernels.h: split_f16x3), never clamped; the GEMM's fp32 accumulation carries the same error as an fp32 GEMM.
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split_f16(const f32x4& v, half4& hi, half4& lo, bool& over) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    us_half h, l;
    split_f16x3(v[k], h, l, over);
    hi[k] = h;
    lo[k] = l;
  }
}
template <bool SPLIT>
__device__ __forceinline__ void store_v(float* __restrict__ V, long long idx, const f32x4& v, bool& over) {
  if (!SPLIT) {
    *reinterpret_cast<f32x4*>(V + idx) = v;
  } else {
    _Float16* vh = reinterpret_cast<_Float16*>(V);
    half4 hi, lo;
    split_f16(v, hi, lo, over);
    const long long o = 2 * (idx & ~7LL) + (idx & 7);
    *reinterpret_cast<half4*>(vh + o) = hi;
    *reinterpret_cast<half4*>(vh + o + 8) = lo;
  }
}

// one thread = one (tile, channel quad); grid (blocks, B)
template <bool SPLIT>
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, int x_ld, float* __restrict__ V, int B, int H, int W,
                                                         int C, unsigned* range_flag) {
  bool over = false;
  const int C4 = C >> 2;
  const int th = (H + 1) >> 1, tw = (W + 1) >> 1;
  const int b = blockIdx.y;
  const long long per_item = (long long)th * tw * C4;
  const long long plane = (long long)B * th * tw * C;          // floats per frequency
  const float* xb = x + (long long)b * H * W * x_ld;
  // (32-bit index arithmetic, per_item < 2^31 host-checked: the four 64-bit divisions per tile cost more than the transform)
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)per_item; i += gridDim.x * 256u) {
    const unsigned t = i / (unsigned)C4;
    const int c = (int)(i - t * (unsigned)C4) * 4;
    const int ty = (int)(t / (unsigned)tw);
    const int tx = (int)(t - (unsigned)ty * (unsigned)tw);
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
    const long long vi = (((long long)b * th + ty) * tw + tx) * C + c;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      // columns: v = t B
      f32x4 v0 = tt[r][0] - tt[r][2], v1 = tt[r][1] + tt[r][2], v2 = tt[r][2] - tt[r][1], v3 = tt[r][1] - tt[r][3];
      store_v<SPLIT>(V, vi + (long long)(r * 4 + 0) * plane, v0, over);
      store_v<SPLIT>(V, vi + (long long)(r * 4 + 1) * plane, v1, over);
      store_v<SPLIT>(V, vi + (long long)(r * 4 + 2) * plane, v2, over);
      store_v<SPLIT>(V, vi + (long long)(r * 4 + 3) * plane, v3, over);
    }
  }
  if (SPLIT) range_report(range_flag, over, kRangeAct);
}

hipError_t launch_wino_input(const float* x, int x_ld, float* V, int B, int H, int W, int C, hipStream_t s, bool split) {
  if (C % 4 != 0 || x_ld % 4 != 0) return hipErrorInvalidValue;
  const long long per_item = (long long)((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
  if (per_item >= (1LL << 31)) return hipErrorInvalidValue;
  int blocks = (int)((per_item + 255) / 256);
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  if (split) hipLaunchKernelGGL(wino_input_kernel<true>, dim3(blocks, B), dim3(256), 0, s, x, x_ld, V, B, H, W, C, current_range_flag());
  else hipLaunchKernelGGL(wino_input_kernel<false>, dim3(blocks, B), dim3(256), 0, s, x, x_ld, V, B, H, W, C, (unsigned*)nullptr);
  return hipGetLastError();
}

// ---- block1's GroupNorm + Mish + time embedding evaluated straight into the Winograd domain ---------------------------------
// V = B^T d B with d = (mish(GroupNorm(y)) * mask + temb) * mask (gn_apply_kernel's block-1 form, ops.hip) computed on the fly:
// the activation h1 = d of a ResnetBlock (unitspeech.py:69-71) is never written.  A workgroup owns a block of 4 x 8 tiles and a
// slab of 32 channels: it evaluates d once per pixel of the 10 x 18 patch (1.4x the interior, against 4x if every tile did its
// own 16 pixels) into LDS, then every thread transforms one (tile, channel quad).
constexpr int kGwTY = 4, kGwTX = 8, kGwCS = 32;
constexpr int kGwPR = 2 * kGwTY + 2, kGwPC = 2 * kGwTX + 2;     // patch rows / cols
constexpr int kGwLd = kGwCS + 4;                                 // padded pixel stride in LDS (floats)

__device__ __forceinline__ float gw_mish(float x) {              // == mish_f (ops.hip)
  if (x > 20.f) return x;
  const float w = __expf(x);
  const float u = w * (w + 2.f);
  return x * (u * __builtin_amdgcn_rcpf(u + 2.f));
}

template <bool SPLIT>
__global__ __launch_bounds__(256) void gn_wino_input_kernel(const float* __restrict__ y, float* __restrict__ V, int B, int H, int W, int C,
                                                            WinoGnArgs g) {
  __shared__ __attribute__((aligned(16))) float patch[kGwPR * kGwPC * kGwLd];
  __shared__ float s_sc[kGwCS], s_sh[kGwCS], s_te[kGwCS];
  const int th = (H + 1) >> 1, tw = (W + 1) >> 1;
  const int nslab = C / kGwCS;
  const int b = blockIdx.z;
  const int slab = blockIdx.y % nslab, tby = blockIdx.y / nslab, tbx = blockIdx.x;
  const int c0 = slab * kGwCS;
  const int ty0 = tby * kGwTY, tx0 = tbx * kGwTX;
  const int cg = C / kGroups;
  if (threadIdx.x < kGwCS) {
    const int c = c0 + threadIdx.x;
    const double cnt = (double)H * W * cg;
    const double mean = stat_read(g.stats, b, c / cg, 0) / cnt;
    double var = stat_read(g.stats, b, c / cg, 1) / cnt - mean * mean;
    if (var < 0) var = 0;
    const float meanf = (float)mean, rstd = (float)(1.0 / sqrt(var + 1e-5));
    const float sc = rstd * g.gamma[c];
    s_sc[threadIdx.x] = sc;
    s_sh[threadIdx.x] = g.beta[c] - meanf * sc;
    s_te[threadIdx.x] = g.temb ? g.temb[(long long)b * C + c] : 0.f;
  }
  __syncthreads();
  const float* yb = y + (long long)b * H * W * C + c0;
  const float* mb = g.mask + (long long)(b % g.mask_bmod) * g.mask_ld;
  for (int idx = threadIdx.x; idx < kGwPR * kGwPC * (kGwCS / 4); idx += 256) {
    const int q = idx % (kGwCS / 4), px = idx / (kGwCS / 4);
    const int pr = px / kGwPC, pc = px - pr * kGwPC;
    const int iy = 2 * ty0 - 1 + pr, ix = 2 * tx0 - 1 + pc;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
      const f32x4 raw = *reinterpret_cast<const f32x4*>(yb + ((long long)iy * W + ix) * C + 4 * q);
      const float m = mb[ix * g.mask_step];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = (gw_mish(raw[k] * s_sc[4 * q + k] + s_sh[4 * q + k]) * m + s_te[4 * q + k]) * m;
    }
    *reinterpret_cast<f32x4*>(patch + px * kGwLd + 4 * q) = v;
    // the patch's interior (its tiles' own 2 x 2 pixels) belongs to this workgroup alone: the activation is stored from here
    if (g.h_out && pr >= 1 && pr <= kGwPR - 2 && pc >= 1 && pc <= kGwPC - 2 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
      *reinterpret_cast<f32x4*>(g.h_out + (((long long)b * H + iy) * W + ix) * C + c0 + 4 * q) = v;
  }
  __syncthreads();
  const int q = threadIdx.x % (kGwCS / 4), t = threadIdx.x / (kGwCS / 4);
  const int tr = t / kGwTX, tc = t - tr * kGwTX;
  const int ty = ty0 + tr, tx = tx0 + tc;
  if (ty >= th || tx >= tw) return;
  bool over = false;
  f32x4 d[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) d[r][qq] = *reinterpret_cast<const f32x4*>(patch + ((2 * tr + r) * kGwPC + 2 * tc + qq) * kGwLd + 4 * q);
  f32x4 tt[4][4];
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) {
    tt[0][qq] = d[0][qq] - d[2][qq];
    tt[1][qq] = d[1][qq] + d[2][qq];
    tt[2][qq] = d[2][qq] - d[1][qq];
    tt[3][qq] = d[1][qq] - d[3][qq];
  }
  const long long plane = (long long)B * th * tw * C;
  const long long vi = (((long long)b * th + ty) * tw + tx) * C + c0 + 4 * q;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    f32x4 v0 = tt[r][0] - tt[r][2], v1 = tt[r][1] + tt[r][2], v2 = tt[r][2] - tt[r][1], v3 = tt[r][1] - tt[r][3];
    store_v<SPLIT>(V, vi + (long long)(r * 4 + 0) * plane, v0, over);
    store_v<SPLIT>(V, vi + (long long)(r * 4 + 1) * plane, v1, over);
    store_v<SPLIT>(V, vi + (long long)(r * 4 + 2) * plane, v2, over);
    store_v<SPLIT>(V, vi + (long long)(r * 4 + 3) * plane, v3, over);
  }
  if (SPLIT) range_report(g.range_flag, over, kRangeAct);
}

bool gn_wino_input_supported(int C) { return C % kGwCS == 0 && C % kGroups == 0; }

hipError_t launch_gn_wino_input(const float* y, float* V, int B, int H, int W, int C, const WinoGnArgs& g, hipStream_t s, bool split) {
  if (!gn_wino_input_supported(C) || !g.stats || !g.gamma || !g.beta || !g.mask) return hipErrorInvalidValue;
  const int th = (H + 1) / 2, tw = (W + 1) / 2;
  const int gy = ((th + kGwTY - 1) / kGwTY) * (C / kGwCS);
  if (gy > 65535 || B > 65535) return hipErrorInvalidValue;
  WinoGnArgs x = g;
  x.range_flag = current_range_flag();
  if (x.mask_bmod < 1) x.mask_bmod = 1;
  if (split) hipLaunchKernelGGL(gn_wino_input_kernel<true>, dim3((tw + kGwTX - 1) / kGwTX, gy, B), dim3(256), 0, s, y, V, B, H, W, C, x);
  else hipLaunchKernelGGL(gn_wino_input_kernel<false>, dim3((tw + kGwTX - 1) / kGwTX, gy, B), dim3(256), 0, s, y, V, B, H, W, C, x);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ M, const float* __restrict__ bias, float* __restrict__ out,
                                                          int out_ld, double* __restrict__ stats, int B, int H, int W, int C, WinoOutExtra x,
                                                          int nslab, long long slab_stride) {
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
  // (32-bit index arithmetic: per_item < 2^31, host-checked; the four 64-bit divisions per tile were a third of this kernel's instructions)
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)per_item; i += gridDim.x * 256u) {
    const unsigned t = i / (unsigned)C4;
    const int c = (int)(i - t * (unsigned)C4) * 4;
    const int ty = (int)(t / (unsigned)tw);
    const int tx = (int)(t - (unsigned)ty * (unsigned)tw);
    const float* mb = M + (((long long)b * th + ty) * tw + tx) * C + c;
    f32x4 m[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = *reinterpret_cast<const f32x4*>(mb + (long long)(r * 4 + q) * plane);
        for (int k = 1; k < nslab; ++k) v += *reinterpret_cast<const f32x4*>(mb + k * slab_stride + (long long)(r * 4 + q) * plane);      // split-K slabs, in slice order
        m[r][q] = v;
      }
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
    // Final merge.  Every lane used to add its 8 fp64 partials to the 16 LDS words by atomics: 2,048 serialised updates per block
    // (rocprofv3 had this kernel at 46 % LDS bank conflicts and 3.5 TB/s).  When a thread keeps its channel quad for life (grid stride a
    // multiple of C/4: every power-of-two width) and the quads of a GroupNorm group are neighbouring lanes, the group's partials are
    // first summed across those lanes by shuffles (fp64 sums: the order does not show at fp32 resolution) and one lane adds them.
    const int seg = C4 / kGroups;          // lanes per group: 4 (C = 128) ... 32 (C = 1,024)
    const bool wave_merge = (((long long)gridDim.x * 256) % C4) == 0 && seg >= 1 && seg <= 64 && (seg & (seg - 1)) == 0 && (64 % seg) == 0 &&
                            (C4 % 64 == 0 || 64 % C4 == 0);
    if (wave_merge) {
      double s1 = (t1[0] + t1[1]) + (t1[2] + t1[3]), s2 = (t2[0] + t2[1]) + (t2[2] + t2[3]);
      for (int off = 1; off < seg; off <<= 1) {
        s1 += __shfl_xor(s1, off);
        s2 += __shfl_xor(s2, off);
      }
      const int lane = threadIdx.x & 63;
      if ((lane & (seg - 1)) == 0 && tc >= 0) {
        atomicAdd(&s_g[tc / cg][0], s1);
        atomicAdd(&s_g[tc / cg][1], s2);
      }
    } else {
      flush();
    }
    __syncthreads();
    if (threadIdx.x < kGroups * 2)
      stat_add(stats, b, threadIdx.x >> 1, threadIdx.x & 1, blockIdx.x, s_g[threadIdx.x >> 1][threadIdx.x & 1]);
  }
}

hipError_t launch_wino_output(const float* M, const float* bias, float* out, int out_ld, double* stats, int B, int H, int W, int C,
                              hipStream_t s, const WinoOutExtra* extra, int nslab, long long slab_stride) {
  if (C % 4 != 0 || out_ld % 4 != 0 || C % kGroups != 0 || (extra && extra->add && extra->add_ld % 4 != 0)) return hipErrorInvalidValue;
  WinoOutExtra x{};
  if (extra) x = *extra;
  if (x.mask_bmod < 1) x.mask_bmod = 1;
  const long long per_item = (long long)((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
  if (per_item >= (1LL << 31)) return hipErrorInvalidValue;
  int blocks = (int)((per_item + 255) / 256);
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(wino_output_kernel, dim3(blocks, B), dim3(256), 0, s, M, bias, out, out_ld, stats, B, H, W, C, x, nslab < 1 ? 1 : nslab, slab_stride);
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

// f16x3 form of the same pack (bk = 32): U as two interleaved fp16 planes, dst (halves) [f][K/32][N][4 groups x (8 hi | 8 lo)],
// K = Cin, N = Cout (forward) or K = Cout, N = Cin (dgrad).  One thread owns 8 consecutive K indices of one N index, i.e. for
// every frequency one whole 32-byte (8 hi | 8 lo) piece: 16-byte stores, neighbouring threads fill neighbouring pieces of a row
// (fine-tuning re-packs every weight after every optimiser step: the element-wise 2-byte scatter of the first version took 2.1 of
// an iteration's 14 ms).
__global__ __launch_bounds__(256) void wino_pack_weight_f16_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, int Cout, int Cin,
                                                                   int dgrad, unsigned* range_flag) {
  bool over = false;
  wino_pack_f16_body(src, dst, Cout, Cin, dgrad, blockIdx.x, gridDim.x, over);      // pack_f16.h
  range_report(range_flag, over, kRangeWeight);
}

// every pending f16x3 pack of a weight sync in ONE launch: a block finds its job by bisection over the jobs' first-block indices
__global__ __launch_bounds__(256) void pack_table_kernel(const PackJob* __restrict__ jobs, int n_jobs, unsigned* range_flag) {
  int lo = 0, hi = n_jobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((int)blockIdx.x >= jobs[mid].blk0) lo = mid; else hi = mid - 1;
  }
  const PackJob j = jobs[lo];
  bool over = false;
  if (j.kind == 0) wino_pack_f16_body(j.src, j.dst, j.Cout, j.Cin, j.a, (long long)blockIdx.x - j.blk0, j.nblk, over);
  else conv_pack_f16_body(j.src, j.dst, j.Cout, j.Cin, j.KH, j.KW, j.a, j.b, (long long)blockIdx.x - j.blk0, j.nblk, over);
  range_report(range_flag, over, kRangeWeight);
}

hipError_t launch_pack_table(const PackJob* jobs_dev, int n_jobs, int total_blocks, hipStream_t s) {
  if (n_jobs <= 0 || total_blocks <= 0) return hipSuccess;
  hipLaunchKernelGGL(pack_table_kernel, dim3(total_blocks), dim3(256), 0, s, jobs_dev, n_jobs, current_range_flag());
  return hipGetLastError();
}

hipError_t launch_wino_pack_weight_f16(const float* src, float* dst, int Cout, int Cin, hipStream_t s, bool dgrad) {
  if ((dgrad ? Cout : Cin) % 32 != 0) return hipErrorInvalidValue;
  long long total = (long long)Cout * Cin / 8;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(wino_pack_weight_f16_kernel, dim3(blocks), dim3(256), 0, s, src, reinterpret_cast<_Float16*>(dst), Cout, Cin, dgrad ? 1 : 0,
                     current_range_flag());
  return hipGetLastError();
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
