// HBM-bound kernels of the decoder: GroupNorm(8)+Mish, the 2-channel first layer, final 1x1 projection,
// time/speaker embedding MLPs, CFG combine + ancestral sampler update and the counter-based gaussian source.
// All of them are streaming passes with 16-byte coalesced accesses along the channel (pixel-major layout) or
// the time axis (planar [B][F][T] boundary tensors); reductions use wave64 shuffles.
#include "kernels.h"

namespace us {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// f16x3 range word of the handle whose entry point runs on this host thread (kernels.h)
static thread_local unsigned* t_range_flag = nullptr;
unsigned* current_range_flag() { return t_range_flag; }
void set_range_flag(unsigned* p) { t_range_flag = p; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// Mish (unitspeech/unitspeech.py:13-15): x*tanh(softplus(x)), softplus threshold 20.
// tanh(log(1+w)) = ((1+w)^2-1)/((1+w)^2+1) = w(w+2)/(w(w+2)+2) with w = e^x: one exp, one divide, no cancellation.
// v_exp_f32 / v_rcp_f32 forms (about 1 ulp each): the library expf and IEEE division made the GroupNorm+Mish pass
// VALU-bound at 3 TB/s instead of HBM-bound.  (Round 4: `__frcp_rn` turned out to BE an IEEE division on this compiler -- v_div_scale / v_rcp /
// four fmas / v_div_fmas / v_div_fixup per value -- so the reciprocal is spelled __builtin_amdgcn_rcpf now.)
__device__ __forceinline__ float mish_f(float x) {
  if (x > 20.f) return x;              // softplus(x) = x beyond the threshold and tanh(x > 20) == 1 in fp32
  float w = __expf(x);
  float u = w * (w + 2.f);
  return x * (u * __builtin_amdgcn_rcpf(u + 2.f));
}

// ---------------------------------------------------------------------------------------------------
// input stacking: in2[b'][f][t][0:2] = (mu, x) * mask        (unitspeech/unitspeech.py:170 + Block's x*mask :54)
// ---------------------------------------------------------------------------------------------------
__global__ void stack_inputs_kernel(const float* __restrict__ x, int Bx, const float* __restrict__ mu, int Bmu, int n_text_uncond,
                                    const float* __restrict__ mu_feat, const float* __restrict__ mask, int Bm, float* __restrict__ in2,
                                    int Bp, int F, int T) {
  const long long total = (long long)Bp * F * T;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int t = (int)(i % T);
    long long r = i / T;
    int f = (int)(r % F);
    int b = (int)(r / F);
    float m = mask[(long long)(b % Bm) * T + t];
    float xv = x[((long long)(b % Bx) * F + f) * T + t];
    float mv = b < n_text_uncond ? mu_feat[f] : mu[((long long)(b % Bmu) * F + f) * T + t];
    float2 o;
    o.x = mv * m;
    o.y = xv * m;
    reinterpret_cast<float2*>(in2)[i] = o;
  }
}

hipError_t launch_stack_inputs(const float* x, int Bx, const float* mu, int Bmu, int n_text_uncond, const float* mu_feat,
                               const float* mask, int Bm, float* in2, int Bp, int F, int T, hipStream_t s) {
  long long total = (long long)Bp * F * T;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(stack_inputs_kernel, dim3(blocks), dim3(256), 0, s, x, Bx, mu, Bmu, n_text_uncond, mu_feat, mask, Bm, in2, Bp, F, T);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// first layer: Conv2d(2, C, 3, pad 1) and res_conv Conv2d(2, C, 1) of downs.0.0 (unitspeech.py:48,66)
// One workgroup = FC_TW consecutive frames of one mel row; the (3 x (FC_TW + 2) x 2) masked input patch sits in LDS.
// Thread = (channel quad, run of consecutive frames): its 4 x 18 weights live in registers, the 3 x 3 x 2 window slides along the run (three
// 8-byte LDS reads per frame), and every frame ends in ONE 16-byte store -- a wave writes whole 512-byte pixel rows.  Round 3's form
// (thread = one channel: 18 broadcast LDS reads and a 4-byte store per output element) was bound by LDS instruction issue: 87 us for the
// 126 MB it writes at 80 x 1024, 1.5 TB/s.  Same fmaf chain per output element (bias, then ci, ky, kx), so the same bits.
// ---------------------------------------------------------------------------------------------------
constexpr int FC_TW = 128;
__global__ __launch_bounds__(256) void first_conv_kernel(const float* __restrict__ in2, const float* __restrict__ w3,
                                                         const float* __restrict__ b3, const float* __restrict__ w1,
                                                         const float* __restrict__ b1, float* __restrict__ y, float* __restrict__ r,
                                                         double* __restrict__ stats, int F, int T, int C) {
  __shared__ __attribute__((aligned(8))) float patch[3][FC_TW + 2][2];
  __shared__ double gred[kGroups][2];    // fp64: the merge order of the per-thread fp32 partials must not show in the result
  if (threadIdx.x < kGroups * 2) gred[threadIdx.x >> 1][threadIdx.x & 1] = 0.0;
  const int b = blockIdx.z, f = blockIdx.y, t0 = blockIdx.x * FC_TW;
  const int tid = threadIdx.x;
  for (int i = tid; i < 3 * (FC_TW + 2) * 2; i += 256) {
    int c = i & 1, xx = (i >> 1) % (FC_TW + 2), yy = (i >> 1) / (FC_TW + 2);
    int ff = f + yy - 1, tt = t0 + xx - 1;
    float v = 0.f;
    if (ff >= 0 && ff < F && tt >= 0 && tt < T) v = in2[(((long long)b * F + ff) * T + tt) * 2 + c];
    patch[yy][xx][c] = v;
  }
  __syncthreads();
  const int C4 = C >> 2;
  const int qpb = C4 < 256 ? C4 : 256;              // quads handled side by side
  const int nrun = 256 / qpb;                       // runs of frames side by side (threads beyond qpb * nrun idle: C / 4 not a divisor of 256)
  const int flen = (FC_TW + nrun - 1) / nrun;
  const int run = tid / qpb, ql = tid - run * qpb;
  const int cg = C / kGroups;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  if (run < nrun) {
    for (int q = ql; q < C4; q += qpb) {
      const int co = 4 * q;
      float w[4][18];
      f32x4 bb, r0, r1, rb;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int i = 0; i < 18; ++i) w[k][i] = w3[(co + k) * 18 + i];   // OIHW: [co][ci][ky][kx]
        bb[k] = b3[co + k]; r0[k] = w1[(co + k) * 2]; r1[k] = w1[(co + k) * 2 + 1]; rb[k] = b1[co + k];
      }
      f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
      const int xlo = run * flen, xhi = (xlo + flen < FC_TW ? xlo + flen : FC_TW);
      // the window: win[ky][kx] = (channel 0, channel 1) of patch[ky][xx + kx]
      f32x2 win[3][3];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        win[ky][1] = *reinterpret_cast<const f32x2*>(&patch[ky][xlo][0]);
        win[ky][2] = *reinterpret_cast<const f32x2*>(&patch[ky][xlo + 1][0]);
      }
      for (int xx = xlo; xx < xhi; ++xx) {
        if (t0 + xx >= T) break;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          win[ky][0] = win[ky][1];
          win[ky][1] = win[ky][2];
          win[ky][2] = *reinterpret_cast<const f32x2*>(&patch[ky][xx + 2][0]);
        }
        f32x4 acc = bb;
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
              for (int k = 0; k < 4; ++k) acc[k] = fmaf(w[k][ci * 9 + ky * 3 + kx], win[ky][kx][ci], acc[k]);
        const long long p = ((long long)b * F + f) * T + t0 + xx;
        *reinterpret_cast<f32x4*>(y + p * C + co) = acc;
        s1 += acc;
        s2 += acc * acc;
        if (r) {
          f32x4 rv;
#pragma unroll
          for (int k = 0; k < 4; ++k) rv[k] = fmaf(r1[k], win[1][1][1], fmaf(r0[k], win[1][1][0], rb[k]));
          *reinterpret_cast<f32x4*>(r + p * C + co) = rv;
        }
      }
      if (stats) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          atomicAdd(&gred[(co + k) / cg][0], (double)s1[k]);
          atomicAdd(&gred[(co + k) / cg][1], (double)s2[k]);
        }
      }
    }
  }
  if (stats) {
    __syncthreads();
    if (threadIdx.x < kGroups * 2)
      stat_add(stats, b, threadIdx.x >> 1, threadIdx.x & 1, blockIdx.x + blockIdx.y * gridDim.x, gred[threadIdx.x >> 1][threadIdx.x & 1]);
  }
}

hipError_t launch_first_conv(const float* in2, const float* w3, const float* b3, const float* w1, const float* b1, float* y,
                             float* r, double* stats, int Bp, int F, int T, int C, hipStream_t s) {
  if (C % 4 != 0) return hipErrorInvalidValue;
  dim3 grid((T + FC_TW - 1) / FC_TW, F, Bp);
  hipLaunchKernelGGL(first_conv_kernel, grid, dim3(256), 0, s, in2, w3, b3, w1, b1, y, r, stats, F, T, C);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// GroupNorm statistics (standalone; the conv epilogue normally produces them)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ y, int ld, int n, int C, double* __restrict__ stats) {
  __shared__ double red[kGroups][2];
  const int b = blockIdx.y, tid = threadIdx.x;
  if (tid < kGroups * 2) red[tid >> 1][tid & 1] = 0.0;
  __syncthreads();
  const int cg = C / kGroups;
  const long long total = (long long)n * C;
  const float* yb = y + (long long)b * n * ld;
  // thread walks elements i = chunk*256+tid (channel-fastest => coalesced); group changes per element
  double s1[kGroups], s2[kGroups];
#pragma unroll
  for (int g = 0; g < kGroups; ++g) s1[g] = s2[g] = 0.0;
  for (long long i = blockIdx.x * 256LL + tid; i < total; i += (long long)gridDim.x * 256) {
    int c = (int)(i % C);
    long long p = i / C;
    float v = yb[p * ld + c];
    int g = c / cg;
#pragma unroll
    for (int k = 0; k < kGroups; ++k)
      if (k == g) { s1[k] += v; s2[k] += (double)v * v; }
  }
#pragma unroll
  for (int g = 0; g < kGroups; ++g) {
    double a = wave_sum_d(s1[g]), q = wave_sum_d(s2[g]);
    if ((tid & 63) == 0) { atomicAdd(&red[g][0], a); atomicAdd(&red[g][1], q); }
  }
  __syncthreads();
  if (tid < kGroups * 2) stat_add(stats, b, tid >> 1, tid & 1, blockIdx.x, red[tid >> 1][tid & 1]);
}

hipError_t launch_gn_stats(const float* y, int ld, int B, int n, int C, double* stats, hipStream_t s) {
  long long total = (long long)n * C;
  int blocks = (int)((total + 256 * 16 - 1) / (256 * 16));
  if (blocks < 1) blocks = 1;
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(gn_stats_kernel, dim3(blocks, B), dim3(256), 0, s, y, ld, n, C, stats);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// GroupNorm apply + Mish + mask (+ time-embedding addend) (+ masked residual)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_apply_kernel(GnApplyArgs a) {
  // A thread owns one channel quad for its whole life (C/4 divides 256 or is a multiple of it for every width of the
  // U-Net), so the per-channel scale/shift are folded once: z = y*sc + sh with sc = rstd*gamma, sh = beta - mean*sc.
  const int b = blockIdx.y;
  const int C4 = a.C >> 2;
  const int cg = a.C / kGroups;
  const long long n = (long long)a.H * a.W;
  const double cnt = (double)n * cg;
  __shared__ float s_mean[kGroups], s_rstd[kGroups];
  if (threadIdx.x < kGroups) {
    double mean = stat_read(a.stats, b, threadIdx.x, 0) / cnt;
    double var = stat_read(a.stats, b, threadIdx.x, 1) / cnt - mean * mean;
    if (var < 0) var = 0;
    s_mean[threadIdx.x] = (float)mean;
    s_rstd[threadIdx.x] = (float)(1.0 / sqrt(var + 1e-5));
  }
  __syncthreads();
  const float* yb = a.y + (long long)b * n * a.y_ld;
  float* ob = a.out + (long long)b * n * a.out_ld;
  float* o2b = a.out2 ? a.out2 + (long long)b * n * a.out2_ld : nullptr;
  const float* rb = a.res ? a.res + (long long)b * n * a.res_ld : nullptr;
  const float* mb = a.mask + (long long)(b % a.mask_bmod) * a.mask_ld;
  const float* tb = a.temb ? a.temb + (long long)b * a.temb_ld : nullptr;
  const long long total = n * C4;
  const long long stride = (long long)gridDim.x * 256;
  long long i = blockIdx.x * 256LL + threadIdx.x;
  const bool fixed_quad = (stride % C4) == 0;      // then (i % C4) never changes for this thread
  bool over = false;
  int c = (int)(i % C4) * 4;
  f32x4 sc, sh, te = {0.f, 0.f, 0.f, 0.f};
  // res2: the residual is the 2-channel 1x1 res_conv of the first ResnetBlock, evaluated here from the stacked input (2 floats per pixel)
  // exactly as first_conv_kernel would have stored it (same fmaf chain): 126 MB less written and read per item at 80 x 1024
  f32x4 r2w0 = {0.f, 0.f, 0.f, 0.f}, r2w1 = r2w0, r2b = r2w0;
  const float* in2b = a.res2_in ? a.res2_in + (long long)b * n * 2 : nullptr;
  auto load_quad = [&](int cc) {
    if (in2b) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { r2w0[k] = a.res2_w[(cc + k) * 2]; r2w1[k] = a.res2_w[(cc + k) * 2 + 1]; }
      r2b = *reinterpret_cast<const f32x4*>(a.res2_b + cc);
    }
    f32x4 ga = *reinterpret_cast<const f32x4*>(a.gamma + cc);
    f32x4 be = *reinterpret_cast<const f32x4*>(a.beta + cc);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int gk = (cc + k) / cg;
      sc[k] = s_rstd[gk] * ga[k];
      sh[k] = be[k] - s_mean[gk] * sc[k];
    }
    if (tb) te = *reinterpret_cast<const f32x4*>(tb + cc);
  };
  if (i < total) load_quad(c);
  long long p = i / C4;
  int w = (int)(p % a.W);
  const long long rpi = stride / C4;               // rows advanced per iteration (fixed_quad only)
  const int wstep = (int)(rpi % a.W);
  for (; i < total; i += stride) {
    if (!fixed_quad) {
      p = i / C4;
      w = (int)(p % a.W);
      c = (int)(i - p * C4) * 4;
      load_quad(c);
    }
    const float m = mb[w * a.mask_step];
    f32x4 v = *reinterpret_cast<const f32x4*>(yb + p * a.y_ld + c);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = mish_f(v[k] * sc[k] + sh[k]) * m;
    o += te;
    if (rb) {
      f32x4 rr = *reinterpret_cast<const f32x4*>(rb + p * a.res_ld + c);
      o += a.res_masked ? rr * m : rr;
    } else if (in2b) {
      const float2 x2 = *reinterpret_cast<const float2*>(in2b + p * 2);
      f32x4 rr;
#pragma unroll
      for (int k = 0; k < 4; ++k) rr[k] = fmaf(r2w1[k], x2.y, fmaf(r2w0[k], x2.x, r2b[k]));
      o += a.res_masked ? rr * m : rr;
    }
    if (a.post_mask) o *= m;
    if (a.out_split) {
      // hi = fp16(x), lo = fp16((x - hi) * 2^11); a value beyond the fp16 range is reported, not clamped (kernels.h)
      typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
      half4_t hi, lo;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        us_half h, l;
        split_f16x3(o[k], h, l, over);
        hi[k] = h;
        lo[k] = l;
      }
      _Float16* oh = reinterpret_cast<_Float16*>(ob + p * a.out_ld) + 2 * (c & ~7) + (c & 7);
      *reinterpret_cast<half4_t*>(oh) = hi;
      *reinterpret_cast<half4_t*>(oh + 8) = lo;
    } else {
      *reinterpret_cast<f32x4*>(ob + p * a.out_ld + c) = o;
    }
    if (o2b) {
      typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
      half4_t hi, lo;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        us_half h, l;
        split_f16x3(o[k], h, l, over);
        hi[k] = h;
        lo[k] = l;
      }
      _Float16* oh = reinterpret_cast<_Float16*>(o2b + p * a.out2_ld) + 2 * (c & ~7) + (c & 7);
      *reinterpret_cast<half4_t*>(oh) = hi;
      *reinterpret_cast<half4_t*>(oh + 8) = lo;
    }
    if (fixed_quad) {
      p += rpi;
      w += wstep;
      if (w >= a.W) w -= a.W;
    }
  }
  if (a.out_split || a.out2) range_report(a.range_flag, over, kRangeAct);
}

hipError_t launch_gn_apply(const GnApplyArgs& a_in, hipStream_t s) {
  GnApplyArgs a = a_in;
  a.range_flag = current_range_flag();
  if (a.C % 4 != 0 || a.C % kGroups != 0 || a.y_ld % 4 != 0 || a.out_ld % 4 != 0 || (a.res && a.res_ld % 4 != 0))
    return hipErrorInvalidValue;
  if (a.out_split && (a.C % 8 != 0 || a.out_ld % 8 != 0 || a.out == a.y)) return hipErrorInvalidValue;
  if (a.out2 && (a.C % 8 != 0 || a.out2_ld % 8 != 0 || a.out2 == a.y || a.out2 == a.out || a.out2 == a.res)) return hipErrorInvalidValue;
  long long total = (long long)a.H * a.W * (a.C / 4);
  int blocks = (int)((total + 256 * 4 - 1) / (256 * 4));
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gn_apply_kernel, dim3(blocks, a.B), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// final_conv: 1x1, C -> 1 on the masked block output, times mask            (unitspeech.py:199-201)
// half a wave (32 lanes x float4 = 128 channels per pass) per pixel.  GN = true: `h` is the raw output of the final Block's
// convolution and GroupNorm + Mish + mask (unitspeech.py:198, Block :27-33) are applied on the fly with gn_apply_kernel's arithmetic,
// so the normalised tensor is never stored (126 MB less written and read per item at 80 x 1024).
// ---------------------------------------------------------------------------------------------------
template <bool GN>
__global__ __launch_bounds__(256) void final_conv_kernel(const float* __restrict__ h, int ld, const float* __restrict__ w,
                                                         const float* __restrict__ b0, const float* __restrict__ mask, int mask_ld,
                                                         int mask_bmod, float* __restrict__ out, int W, long long n, int C,
                                                         const double* __restrict__ stats, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta) {
  const int b = blockIdx.y;
  const int l32 = threadIdx.x & 31;
  const long long half0 = (blockIdx.x * 256LL + threadIdx.x) >> 5;
  const long long nhalf = ((long long)gridDim.x * 256) >> 5;
  const float* hb = h + (long long)b * n * ld;
  const float* mb = mask + (long long)(b % mask_bmod) * mask_ld;
  const float bias = b0[0];
  __shared__ float s_mean[kGroups], s_rstd[kGroups];
  const int cg = C / kGroups;
  if constexpr (GN) {
    if (threadIdx.x < kGroups) {
      const double cnt = (double)n * cg;
      double mean = stat_read(stats, b, threadIdx.x, 0) / cnt;
      double var = stat_read(stats, b, threadIdx.x, 1) / cnt - mean * mean;
      if (var < 0) var = 0;
      s_mean[threadIdx.x] = (float)mean;
      s_rstd[threadIdx.x] = (float)(1.0 / sqrt(var + 1e-5));
    }
    __syncthreads();
  }
  f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = sc;
  auto fold = [&](int c) {
    f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c);
    f32x4 be = *reinterpret_cast<const f32x4*>(beta + c);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int gk = (c + k) / cg;
      sc[k] = s_rstd[gk] * ga[k];
      sh[k] = be[k] - s_mean[gk] * sc[k];
    }
  };
  const bool one_pass = C <= 128;
  if constexpr (GN) {
    if (one_pass && l32 * 4 < C) fold(l32 * 4);
  }
  for (long long p = half0; p < n; p += nhalf) {
    float acc = 0.f;
    const float m = mb[(int)(p % W)];
    for (int c = l32 * 4; c < C; c += 128) {
      f32x4 v = *reinterpret_cast<const f32x4*>(hb + p * ld + c);
      f32x4 ww = *reinterpret_cast<const f32x4*>(w + c);
      if constexpr (GN) {
        if (!one_pass) fold(c);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = mish_f(v[k] * sc[k] + sh[k]) * m;
      }
      acc += v[0] * ww[0] + v[1] * ww[1] + v[2] * ww[2] + v[3] * ww[3];
    }
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (l32 == 0) out[(long long)b * n + p] = (acc * m + bias) * m;
  }
}

hipError_t launch_final_conv(const float* h, int ld, const float* w, const float* b0, const float* mask, int mask_ld,
                             int mask_bmod, float* out, int B, int H, int W, int C, hipStream_t s, const double* stats,
                             const float* gamma, const float* beta) {
  if (C % 4 != 0) return hipErrorInvalidValue;
  if (stats && (C % kGroups != 0 || !gamma || !beta)) return hipErrorInvalidValue;
  long long n = (long long)H * W;
  int blocks = (int)((n + 7) / 8);
  if (blocks > 2048) blocks = 2048;
  if (stats)
    hipLaunchKernelGGL(final_conv_kernel<true>, dim3(blocks, B), dim3(256), 0, s, h, ld, w, b0, mask, mask_ld, mask_bmod, out, W, n, C,
                       stats, gamma, beta);
  else
    hipLaunchKernelGGL(final_conv_kernel<false>, dim3(blocks, B), dim3(256), 0, s, h, ld, w, b0, mask, mask_ld, mask_bmod, out, W, n, C,
                       nullptr, nullptr, nullptr);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// small dense layers: one wave per output element
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ in, int in_ld, const float* __restrict__ W,
                                                     const float* __restrict__ bias, float* __restrict__ out, int out_ld, int in_dim,
                                                     int out_dim, int mish_in) {
  const int r = blockIdx.y;
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (o >= out_dim) return;
  const float* x = in + (long long)r * in_ld;
  const float* w = W + (long long)o * in_dim;
  float acc = 0.f;
  for (int i = lane; i < in_dim; i += 64) {
    float v = x[i];
    if (mish_in) v = mish_f(v);
    acc = fmaf(w[i], v, acc);
  }
  acc = wave_sum(acc);
  if (lane == 0) out[(long long)r * out_ld + o] = acc + (bias ? bias[o] : 0.f);
}

// several layers on one input: one wave per output element of the concatenated outputs (LinJob, kernels.h); per element the arithmetic of
// linear_kernel, so the result does not depend on which of the two computed it
__global__ __launch_bounds__(256) void linear_multi_kernel(const LinJob* __restrict__ jobs, int njobs, int total_out, const float* __restrict__ in,
                                                           int in_ld, int in_dim, int mish_in) {
  const int r = blockIdx.y;
  const int og = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (og >= total_out) return;
  int j = 0;
  while (j + 1 < njobs && og >= jobs[j + 1].o0) ++j;
  const LinJob jb = jobs[j];
  const int o = og - jb.o0;
  const float* x = in + (long long)r * in_ld;
  const float* w = jb.W + (long long)o * in_dim;
  float acc = 0.f;
  for (int i = lane; i < in_dim; i += 64) {
    float v = x[i];
    if (mish_in) v = mish_f(v);
    acc = fmaf(w[i], v, acc);
  }
  acc = wave_sum(acc);
  if (lane == 0) jb.out[(long long)r * jb.out_ld + o] = acc + (jb.bias ? jb.bias[o] : 0.f);
}

hipError_t launch_linear_multi(const LinJob* jobs_dev, int njobs, int total_out, const float* in, int in_ld, int rows, int in_dim, bool mish_in,
                               hipStream_t s) {
  if (rows <= 0 || njobs <= 0 || total_out <= 0) return hipSuccess;
  hipLaunchKernelGGL(linear_multi_kernel, dim3((total_out + 3) / 4, rows), dim3(256), 0, s, jobs_dev, njobs, total_out, in, in_ld, in_dim,
                     mish_in ? 1 : 0);
  return hipGetLastError();
}

hipError_t launch_linear(const float* in, int in_ld, const float* W, const float* bias, float* out, int out_ld, int rows,
                         int in_dim, int out_dim, bool mish_in, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(linear_kernel, dim3((out_dim + 3) / 4, rows), dim3(256), 0, s, in, in_ld, W, bias, out, out_ld, in_dim, out_dim,
                     mish_in ? 1 : 0);
  return hipGetLastError();
}

__global__ void pos_emb_kernel(const float* __restrict__ t, float* __restrict__ emb, int rows, int dim, float scale) {
  const int half = dim / 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * half) return;
  const int r = i / half, k = i % half;
  // emb = exp(arange(half) * -(log(10000)/(half-1))); arg = scale * t * emb      (unitspeech.py:116-119)
  const float e = (float)(9.210340371976184 / (double)(half - 1));   // math.log(10000)/(half-1) rounded to fp32
  // the exponential through fp64 and rounded once: correctly rounded, which is what torch's CPU exp returns for all 64 arguments of the
  // full-size model; the device expf is an ulp off for some k, and an ulp of `freq` is an ulp of an argument near 1000: 3e-5 in sin / cos
  // (tests/test_hip_parity_r2.py::test_time_embedding_vs_reference_golden)
  const float freq = (float)exp((double)((float)k * -e));
  const float arg = scale * t[r] * freq;
  emb[(long long)r * dim + k] = sinf(arg);
  emb[(long long)r * dim + half + k] = cosf(arg);
}

hipError_t launch_pos_emb(const float* t, float* emb, int rows, int dim, float scale, hipStream_t s) {
  int total = rows * (dim / 2);
  hipLaunchKernelGGL(pos_emb_kernel, dim3((total + 255) / 256), dim3(256), 0, s, t, emb, rows, dim, scale);
  return hipGetLastError();
}

__global__ void copy_rows_kernel(const float* __restrict__ src, int src_ld, int src_rows, float* __restrict__ dst, int dst_ld, int rows,
                                 int n) {
  const long long total = (long long)rows * n;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int r = (int)(i / n), c = (int)(i % n);
    dst[(long long)r * dst_ld + c] = src[(long long)(r % src_rows) * src_ld + c];
  }
}

hipError_t launch_copy_rows(const float* src, int src_ld, int src_rows, float* dst, int dst_ld, int rows, int n, hipStream_t s) {
  long long total = (long long)rows * n;
  if (total <= 0) return hipSuccess;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(copy_rows_kernel, dim3(blocks), dim3(256), 0, s, src, src_ld, src_rows, dst, dst_ld, rows, n);
  return hipGetLastError();
}

// one block column per table entry (blockIdx.y), grid-stride over its elements: ~130 RAW weight tensors in one launch
__global__ __launch_bounds__(256) void copy_table_kernel(const CopyEnt* __restrict__ tab) {
  const CopyEnt e = tab[blockIdx.y];
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < e.n; i += (long long)gridDim.x * 256) e.dst[i] = e.src[i];
}

hipError_t launch_copy_table(const CopyEnt* tab_dev, int n_entries, hipStream_t s) {
  if (n_entries <= 0) return hipSuccess;
  hipLaunchKernelGGL(copy_table_kernel, dim3(64, n_entries), dim3(256), 0, s, tab_dev);
  return hipGetLastError();
}

__global__ void fill_kernel(float* __restrict__ dst, float value, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = value;
}

__global__ __launch_bounds__(256) void scale_table_kernel(const CopyEnt* __restrict__ tab, const float* __restrict__ scale) {
  const CopyEnt e = tab[blockIdx.y];
  const float k = scale[0];
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < e.n; i += (long long)gridDim.x * 256) e.dst[i] = e.src[i] * k;
}

hipError_t launch_scale_table(const CopyEnt* tab_dev, int n_entries, const float* scale_dev, hipStream_t s) {
  if (n_entries <= 0) return hipSuccess;
  hipLaunchKernelGGL(scale_table_kernel, dim3(64, n_entries), dim3(256), 0, s, tab_dev, scale_dev);
  return hipGetLastError();
}

// 16-byte stores, four per thread: the weight-gradient scratch of a 1024-channel layer is 38 MB per zero-fill
__global__ __launch_bounds__(256) void fill4_kernel(float* __restrict__ dst, float value, int n) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const f32x4 v = {value, value, value, value};
  const int q = n >> 2;
  f32x4* d4 = reinterpret_cast<f32x4*>(dst);
  const int base = blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = base + u * 256;
    if (i < q) d4[i] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[(q << 2) + threadIdx.x] = value;
}

hipError_t launch_fill(float* dst, float value, int n, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  if (n >= 4096 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
    hipLaunchKernelGGL(fill4_kernel, dim3(((n >> 2) + 1023) / 1024), dim3(256), 0, s, dst, value, n);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(fill_kernel, dim3((n + 255) / 256), dim3(256), 0, s, dst, value, n);
  return hipGetLastError();
}

__global__ __launch_bounds__(64) void l2_normalize_kernel(const float* __restrict__ src, float* __restrict__ dst, int n) {
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) acc += src[i] * src[i];
  acc = wave_sum(acc);
  const float nrm = sqrtf(acc);
  for (int i = threadIdx.x; i < n; i += 64) dst[i] = src[i] / nrm;
}

hipError_t launch_l2_normalize(const float* src, float* dst, int n, hipStream_t s) {
  hipLaunchKernelGGL(l2_normalize_kernel, dim3(1), dim3(64), 0, s, src, dst, n);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// counter-based gaussian source: Philox4x32-10 -> 4 uniforms -> 2 Box-Muller pairs
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
    uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

// normals for the 4 consecutive elements [4*q, 4*q+4) of stream (seed, key)
__device__ __forceinline__ f32x4 normal4(unsigned long long seed, unsigned long long key, unsigned long long q) {
  uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32), (uint32_t)key, (uint32_t)(key >> 32)};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  const float k = 2.3283064365386963e-10f;   // 2^-32
  float u0 = ((float)c[0] + 0.5f) * k, u1 = ((float)c[1] + 0.5f) * k;
  float u2 = ((float)c[2] + 0.5f) * k, u3 = ((float)c[3] + 0.5f) * k;
  u0 = fminf(fmaxf(u0, 1e-12f), 1.f);
  u2 = fminf(fmaxf(u2, 1e-12f), 1.f);
  float r0 = sqrtf(-2.f * logf(u0)), r1 = sqrtf(-2.f * logf(u2));
  float s0, c0, s1, c1;
  sincosf(6.283185307179586f * u1, &s0, &c0);
  sincosf(6.283185307179586f * u3, &s1, &c1);
  return f32x4{r0 * c0, r0 * s0, r1 * c1, r1 * s1};
}

__global__ void fill_normal_kernel(float* __restrict__ out, size_t n, unsigned long long seed, unsigned long long key) {
  const size_t nq = (n + 3) / 4;
  for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < nq; q += (size_t)gridDim.x * blockDim.x) {
    f32x4 v = normal4(seed, key, q);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (q * 4 + k < n) out[q * 4 + k] = v[k];
  }
}

hipError_t launch_fill_normal(float* out, size_t n, unsigned long long seed, unsigned long long key, hipStream_t s) {
  if (n == 0) return hipSuccess;
  size_t nq = (n + 3) / 4;
  int blocks = (int)((nq + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(fill_normal_kernel, dim3(blocks), dim3(256), 0, s, out, n, seed, key);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// CFG combine + ancestral update.  Operation order follows the reference's tensor expressions so that an
// identical score gives a bit-identical update (no FMA contraction: explicit mul_rn/add_rn).
//   score = s + w_t*(s - s_tu) + w_s*(s - s_su)                                   (unitspeech.py:322-324)
//   x0    = c0*xt + c1*score                                                      (:273-278)
//   mean  = c2*x0 - (c3*score)*c4                                                 (:283-287)
//   xt'   = (mean + c5*noise) * mask                                              (:370)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sampler_update_kernel(SamplerArgs a) {
  const long long FT = (long long)a.F * a.T;
  const long long total = (long long)a.B * FT;
  const long long BFT = total;
  for (long long q = blockIdx.x * 256LL + threadIdx.x; q * 4 < total; q += (long long)gridDim.x * 256) {
    const long long i0 = q * 4;
    const int b = (int)(i0 / FT);
    const long long rem = i0 - (long long)b * FT;
    const int t = (int)(rem % a.T);       // T % 4 == 0 (T is a multiple of 8): the 4 elements share item and row
    f32x4 x = *reinterpret_cast<const f32x4*>(a.xt + i0);
    f32x4 m = *reinterpret_cast<const f32x4*>(a.mask + (long long)b * a.T + t);
    f32x4 sc;
    if (a.mode == 3) {
      f32x4 s_tu = *reinterpret_cast<const f32x4*>(a.score + i0);
      f32x4 s_su = *reinterpret_cast<const f32x4*>(a.score + BFT + i0);
      f32x4 s = *reinterpret_cast<const f32x4*>(a.score + 2 * BFT + i0);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        sc[k] = add_rn(add_rn(s[k], mul_rn(a.w_text, sub_rn(s[k], s_tu[k]))),
                          mul_rn(a.w_spk, sub_rn(s[k], s_su[k])));
    } else if (a.mode == 2 || a.mode == 1) {
      f32x4 s_u = *reinterpret_cast<const f32x4*>(a.score + i0);
      f32x4 s = *reinterpret_cast<const f32x4*>(a.score + BFT + i0);
      const float w = a.mode == 2 ? a.w_text : a.w_spk;
#pragma unroll
      for (int k = 0; k < 4; ++k) sc[k] = add_rn(s[k], mul_rn(w, sub_rn(s[k], s_u[k])));
    } else {
      sc = *reinterpret_cast<const f32x4*>(a.score + i0);
    }
    f32x4 nz = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.c5 != 0.f) {
      if (a.noise) {
        nz = *reinterpret_cast<const f32x4*>(a.noise + i0);
      } else {
        // stream key = (utterance, step); counter = element quad inside the utterance => independent of sharding
        unsigned long long key = ((unsigned long long)(a.utt0 + b) << 20) ^ (unsigned long long)(unsigned)a.step;
        nz = normal4(a.seed, key, (unsigned long long)(rem >> 2));
      }
    }
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float x0 = add_rn(mul_rn(a.c0, x[k]), mul_rn(a.c1, sc[k]));
      float mean = sub_rn(mul_rn(a.c2, x0), mul_rn(mul_rn(a.c3, sc[k]), a.c4));
      o[k] = mul_rn(add_rn(mean, mul_rn(a.c5, nz[k])), m[k]);
    }
    *reinterpret_cast<f32x4*>(a.out + i0) = o;
  }
}

hipError_t launch_sampler_update(const SamplerArgs& a, hipStream_t s) {
  if (a.T % 4 != 0) return hipErrorInvalidValue;
  long long total = (long long)a.B * a.F * a.T;
  int blocks = (int)((total / 4 + 255) / 256);
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(sampler_update_kernel, dim3(blocks), dim3(256), 0, s, a);
  return hipGetLastError();
}

__global__ void mul_mask_kernel(const float* __restrict__ x, const float* __restrict__ mask, float* __restrict__ out, int B, int F, int T) {
  const long long total = (long long)B * F * T;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int t = (int)(i % T);
    int b = (int)(i / ((long long)F * T));
    out[i] = x[i] * mask[(long long)b * T + t];
  }
}

hipError_t launch_mul_mask(const float* x, const float* mask, float* out, int B, int F, int T, hipStream_t s) {
  long long total = (long long)B * F * T;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(mul_mask_kernel, dim3(blocks), dim3(256), 0, s, x, mask, out, B, F, T);
  return hipGetLastError();
}

}  // namespace us
