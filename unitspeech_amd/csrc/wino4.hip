// Winograd F(MH x MW, 3x3) with 4-wide output tiles, MH x MW in {4x4, 2x4}, for the stride-1 3x3 convolutions of `Block`
// (unitspeech/unitspeech.py:48) at the low-resolution levels of the score network (inference).
//
// F(2x2,3x3) (wino.hip) spends 16 multiplications per 4 output pixels, F(4x4,3x3) 36 per 16, F(2x4,3x3) 24 per 8: 2.25 / 3 instead of 4
// Winograd-domain values per pixel, i.e. 1.78x / 1.33x fewer MFMA FLOPs in the per-frequency GEMMs AND as much less V / M traffic in
// the transform passes.  The price is rounding: the transforms of a 6-point Cook-Toom algorithm have coefficients beyond {0, +-1, +-1/2}.
// On the usual points {0, +-1, +-2} the emulated fp32 pipeline is 5x F(2x2)'s error; on {0, +-5/8, +-3/2} (tools/gen_wino4_coef.py,
// tools/scratch/wino_points.py) it is 2.8x (F(4x4)) and 1.7x (F(2x4)), with every entry of B^T and A^T still exact in fp32.  G is not dyadic:
// U = G g G^T is formed in fp64 at pack time and rounded once to the 22-bit two-plane operand form.
//
// Data flow (separate form only): V[f] = (B^T d B)[f] for f = (i, j), written as the two interleaved fp16 planes of the f16x3 GEMM
// (wino.hip: store_v), optionally with block1's GroupNorm + Mish + time embedding evaluated on the fly; M[f] = V[f] U[f] by
// conv_igemm_kernel as NH * NW independent 1x1 convolutions; Y = A^T M A + bias with the GroupNorm partial sums.
// Tile grid: th = ceil(H / MH) x tw = ceil(W / MW); taps outside the image read zeros, outputs outside it are not stored.
#include "kernels.h"
#include "pack_f16.h"
#include "wino4_coef.h"

namespace us {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half4w __attribute__((ext_vector_type(4)));

namespace {

template <int M> struct Coef;
template <> struct Coef<4> {
  static constexpr int N = 6;
  static constexpr float bt(int i, int j) { return wino4::kBT4[i][j]; }
  static constexpr float at(int i, int j) { return wino4::kAT4[i][j]; }
  static constexpr double g(int i, int j) { return wino4::kG4[i][j]; }
};
template <> struct Coef<2> {
  static constexpr int N = 4;
  static constexpr float bt(int i, int j) { return wino4::kBT2[i][j]; }
  static constexpr float at(int i, int j) { return wino4::kAT2[i][j]; }
  static constexpr double g(int i, int j) { return wino4::kG2[i][j]; }
};

// a * x summed into acc with the coefficient known at compile time after unrolling: zeros vanish, +-1 become adds
__device__ __forceinline__ void axpy(f32x4& acc, float c, const f32x4& x, bool& first) {
  if (c == 0.f) return;
  if (first) { acc = c == 1.f ? x : (c == -1.f ? -x : x * c); first = false; return; }
  if (c == 1.f) acc += x;
  else if (c == -1.f) acc -= x;
  else acc += x * c;
}

// V = B^T d B carries up to (0.879 + 2.64 + 1)^2 = 20x the activation's magnitude for F(4x4) (F(2x2): 4x), which would move the fp16 range of
// the two-plane operand from 65,504 / 4 to 65,504 / 20 of activation.  The transforms are linear, so the INPUT is scaled by 2^-5 (exact; folded
// into the mask and time-embedding factors of the GroupNorm form, one multiply per loaded value otherwise) and the output transform multiplies Y
// by 2^5 inside the bias add (an fma instead of an add): V overflows at |activation| ~ 100,000 now, beyond the direct convolutions' own 65,504.
// Powers of two: every product and sum is the unscaled one times 2^-5 bit for bit (down to values of 2e-3, where hi becomes an fp16 subnormal and
// the lo plane carries the difference at its 1.5e-11 floor).
constexpr float kVScale = 0x1p-5f, kVUnscale = 0x1p+5f;

// 1-D input transform t = B^T d along one axis.  The points are symmetric (+-a, +-b), so rows 1 / 2 and 3 / 4 of B^T are an even part plus /
// minus an odd part: 18 multiply-adds for the six outputs instead of 22.  Coefficients come from the generated table (wino4_coef.h).
template <int M> struct Bt;
template <> struct Bt<4> {
  __device__ static __forceinline__ void apply(const f32x4 (&d)[6], f32x4 (&o)[6]) {
    using K = Coef<4>;
    static_assert(K::bt(2, 1) == -K::bt(1, 1) && K::bt(2, 3) == -K::bt(1, 3) && K::bt(2, 2) == K::bt(1, 2) && K::bt(1, 4) == 1.f, "symmetric points");
    static_assert(K::bt(4, 1) == -K::bt(3, 1) && K::bt(4, 3) == -K::bt(3, 3) && K::bt(4, 2) == K::bt(3, 2) && K::bt(3, 4) == 1.f, "symmetric points");
    static_assert(K::bt(0, 4) == 1.f && K::bt(5, 5) == 1.f && K::bt(0, 1) == 0.f && K::bt(0, 3) == 0.f && K::bt(5, 2) == 0.f && K::bt(5, 4) == 0.f, "layout");
    o[0] = d[0] * K::bt(0, 0) + (d[2] * K::bt(0, 2) + d[4]);
    o[5] = d[1] * K::bt(5, 1) + (d[3] * K::bt(5, 3) + d[5]);
    const f32x4 e1 = d[2] * K::bt(1, 2) + d[4], o1 = d[1] * K::bt(1, 1) + d[3] * K::bt(1, 3);
    o[1] = e1 + o1;
    o[2] = e1 - o1;
    const f32x4 e2 = d[2] * K::bt(3, 2) + d[4], o2 = d[1] * K::bt(3, 1) + d[3] * K::bt(3, 3);
    o[3] = e2 + o2;
    o[4] = e2 - o2;
  }
};
template <> struct Bt<2> {
  __device__ static __forceinline__ void apply(const f32x4 (&d)[4], f32x4 (&o)[4]) {
    using K = Coef<2>;
    static_assert(K::bt(0, 0) == -1.f && K::bt(0, 2) == 1.f && K::bt(1, 1) == 1.f && K::bt(1, 2) == 1.f && K::bt(2, 1) == -1.f && K::bt(2, 2) == 1.f &&
                      K::bt(3, 1) == -1.f && K::bt(3, 3) == 1.f, "F(2, 3) on {0, +-1, inf}");
    o[0] = d[2] - d[0];
    o[1] = d[2] + d[1];
    o[2] = d[2] - d[1];
    o[3] = d[3] - d[1];
  }
};

// ---- input transform: one thread = one (tile, channel quad); grid (blocks, B) -----------------------------------------------------
// GN: x is block1's raw convolution output y (ld = C) and d = (mish(GroupNorm(y)) * mask + temb) * mask is evaluated per loaded pixel
// (gn_apply_kernel's block-1 form, unitspeech.py:69-71; h1 is never written).  Every tile evaluates its own (MH + 2)(MW + 2) pixels, 2.25x
// (1.5 x 1.5 per axis) the interior; neighbouring tiles' re-reads of the halo are served by L1 / L2.
// The first version of this kernel ran at 2 TB/s: its VALU work, not its bytes, set the time (rocprofv3: 67 us for 204 MB): 36 loads and
// 72 stores with 64-bit address arithmetic each, a branch per pixel, a full IEEE division inside every Mish.  Now: pixels come through a
// buffer descriptor (an offset beyond it reads zeros: no branches, the padding for free), V is addressed as uniform plane base + one
// 32-bit offset, the Mish reciprocal is v_rcp_f32, and the range check is a running maximum instead of a compare per value.
template <int MH, int MW, bool GN>
__global__ __launch_bounds__(256) void wino4_input_kernel(const float* __restrict__ x, int x_ld, float* __restrict__ V, int B, int H, int W,
                                                          int C, WinoGnArgs g, unsigned* range_flag) {
  constexpr int NH = Coef<MH>::N, NW = Coef<MW>::N;
  const int C4 = C >> 2;
  const int th = (H + MH - 1) / MH, tw = (W + MW - 1) / MW;
  const int b = blockIdx.y;
  const unsigned per_item = (unsigned)th * tw * C4;
  const long long plane_h = 2LL * B * th * tw * C;             // halves per frequency
  const unsigned item_bytes = (unsigned)H * W * x_ld * 4u;     // < 2^31 (host-checked)
  __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (long long)b * H * W * x_ld), 0, (int)item_bytes, 0x00020000);
  const float* mb = GN ? g.mask + (long long)(b % g.mask_bmod) * g.mask_ld : nullptr;
  const int cg = C / kGroups;
  _Float16* vb = reinterpret_cast<_Float16*>(V) + 2LL * (long long)b * th * tw * C;
  int tc = -1;
  f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = sc, te = sc;
  float amax = 0.f;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < per_item; i += gridDim.x * 256u) {
    const unsigned t = i / C4;
    const int c = (int)(i - t * C4) * 4;
    const int ty = (int)(t / tw), tx = (int)(t - (unsigned)ty * tw);
    if (GN && tc != c) {        // (a thread keeps its channel quad whenever the grid stride is a multiple of C / 4: every U-Net width)
      tc = c;
      const double cnt = (double)H * W * cg;
      const double mean = stat_read(g.stats, b, c / cg, 0) / cnt;
      double var = stat_read(g.stats, b, c / cg, 1) / cnt - mean * mean;
      if (var < 0) var = 0;
      const float meanf = (float)mean, rstd = (float)(1.0 / sqrt(var + 1e-5));
      const f32x4 ga = *reinterpret_cast<const f32x4*>(g.gamma + c), be = *reinterpret_cast<const f32x4*>(g.beta + c);
      sc = ga * rstd;
      sh = be - sc * meanf;
      te = g.temb ? *reinterpret_cast<const f32x4*>(g.temb + (long long)b * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
      te *= kVScale;            // (mish * m + te) * m * 2^-5 == (mish * (m * 2^-5) + te * 2^-5) * m
    }
    // byte offsets per row / column; an invalid one has bit 31 set, so the sum is beyond the descriptor's range and the load returns zeros
    unsigned coff[NW];
    float cm[NW];
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      const int ix = MW * tx - 1 + q;
      const bool ok = (unsigned)ix < (unsigned)W;
      coff[q] = ok ? ((unsigned)ix * (unsigned)x_ld + (unsigned)c) * 4u : 0x80000000u;      // (OR-ed below: two invalid flags must not carry out)
      if (GN) cm[q] = ok ? mb[ix * g.mask_step] : 0.f;      // (the frame mask, 0 / 1)
    }
    // along W first: tr[r][j] = sum_q BT_w[j][q] d[r][q]
    f32x4 tr[NH][NW];
#pragma unroll
    for (int r = 0; r < NH; ++r) {
      const int iy = MH * ty - 1 + r;
      const bool rok = (unsigned)iy < (unsigned)H;
      const unsigned roff = rok ? (unsigned)iy * (unsigned)W * (unsigned)x_ld * 4u : 0x80000000u;
      f32x4 d[NW];
#pragma unroll
      for (int q = 0; q < NW; ++q) {
        f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)((roff + (coff[q] & 0x7fffffffu)) | (coff[q] & 0x80000000u)), 0, 0));
        if (GN) {
          const float m = rok ? cm[q] : 0.f;       // outside the image: (mish(shift) * 0 + temb) * 0 = 0, the zero padding
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float z = v[k] * sc[k] + sh[k];
            const float w = __expf(z);
            const float u = w * (w + 2.f);
            const float mi = z > 20.f ? z : z * (u * __builtin_amdgcn_rcpf(u + 2.f));
            v[k] = (mi * (m * kVScale) + te[k]) * m;
          }
        } else {
          v *= kVScale;
        }
        d[q] = v;
      }
      Bt<MW>::apply(d, tr[r]);
    }
    const unsigned vo = ((unsigned)(ty * tw + tx) * (unsigned)C + (unsigned)c);      // fp32-element index inside (frequency, item)
    const unsigned ho = 2u * (vo & ~7u) + (vo & 7u);                                  // its hi quad in halves; lo 8 halves further
    // then along H: v[i][j] = sum_r BT_h[i][r] tr[r][j]
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      f32x4 col[NH], vv[NH];
#pragma unroll
      for (int r = 0; r < NH; ++r) col[r] = tr[r][j];
      Bt<MH>::apply(col, vv);
#pragma unroll
      for (int ii = 0; ii < NH; ++ii) {
        half4w hi, lo;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float xv = vv[ii][k];
          amax = fmaxf(amax, fabsf(xv));             // (a NaN never enters the maximum: it stays a NaN in the planes, as in fp32)
          const us_half h = (us_half)xv;
          hi[k] = h;
          lo[k] = (us_half)((xv - (float)h) * 2048.f);
        }
        _Float16* pf = vb + (long long)(ii * NW + j) * plane_h;      // uniform: a scalar base per frequency
        *reinterpret_cast<half4w*>(pf + ho) = hi;
        *reinterpret_cast<half4w*>(pf + ho + 8) = lo;
      }
    }
  }
  range_report(range_flag, amax >= kF16Over, kRangeAct);
}

// ---- output transform: one thread = one (tile, channel quad); Y = A_h^T M A_w + bias, GroupNorm partial sums of the result ---------
template <int MH, int MW>
__global__ __launch_bounds__(256) void wino4_output_kernel(const float* __restrict__ M, const float* __restrict__ bias, float* __restrict__ out,
                                                           int out_ld, double* __restrict__ stats, int B, int H, int W, int C) {
  using CH = Coef<MH>;
  using CW = Coef<MW>;
  constexpr int NH = CH::N, NW = CW::N;
  __shared__ double s_g[kGroups][2];
  const int C4 = C >> 2;
  const int th = (H + MH - 1) / MH, tw = (W + MW - 1) / MW;
  const int b = blockIdx.y;
  const long long per_item = (long long)th * tw * C4;
  const long long plane = (long long)B * th * tw * C;
  const int cg = C / kGroups;
  if (threadIdx.x < kGroups * 2) s_g[threadIdx.x >> 1][threadIdx.x & 1] = 0.0;
  __syncthreads();
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
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)per_item; i += gridDim.x * 256u) {      // (32-bit: per_item < 2^31, host-checked)
    const unsigned t = i / (unsigned)C4;
    const int c = (int)(i - t * (unsigned)C4) * 4;
    const int ty = (int)(t / (unsigned)tw);
    const int tx = (int)(t - (unsigned)ty * (unsigned)tw);
    const float* mb = M + (((long long)b * th + ty) * tw + tx) * C + c;
    f32x4 y[MH][MW];
#pragma unroll
    for (int r = 0; r < MH; ++r)
#pragma unroll
      for (int q = 0; q < MW; ++q) y[r][q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int fi = 0; fi < NH; ++fi) {
      f32x4 m[NW];
#pragma unroll
      for (int fj = 0; fj < NW; ++fj) m[fj] = *reinterpret_cast<const f32x4*>(mb + (long long)(fi * NW + fj) * plane);
      // u[q] = sum_fj AT_w[q][fj] m[fj], then y[r][q] += AT_h[r][fi] u[q]: one fixed order of additions for every output
#pragma unroll
      for (int q = 0; q < MW; ++q) {
        f32x4 u = {0.f, 0.f, 0.f, 0.f};
        bool first = true;
#pragma unroll
        for (int fj = 0; fj < NW; ++fj) axpy(u, CW::at(q, fj), m[fj], first);
#pragma unroll
        for (int r = 0; r < MH; ++r) {
          bool never = false;
          axpy(y[r][q], CH::at(r, fi), u, never);
        }
      }
    }
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) bv = *reinterpret_cast<const f32x4*>(bias + c);
    if (stats && tc != c) { flush(); tc = c; }
    float* ob = out + (long long)b * H * W * out_ld + c;
#pragma unroll
    for (int r = 0; r < MH; ++r)
#pragma unroll
      for (int q = 0; q < MW; ++q) {
        const int oy = MH * ty + r, ox = MW * tx + q;
        if (oy < H && ox < W) {
          const f32x4 v = y[r][q] * kVUnscale + bv;       // (V was formed from the input times 2^-5: kVScale)
          *reinterpret_cast<f32x4*>(ob + ((long long)oy * W + ox) * out_ld) = v;
          if (stats) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { t1[k] += (double)v[k]; t2[k] += (double)(v[k] * v[k]); }
          }
        }
      }
  }
  if (stats) {
    // as wino_output_kernel: a GroupNorm group's channel quads are neighbouring lanes; their fp64 partials are merged by shuffles
    const int seg = C4 / kGroups;
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

// ---- weight pack: U[f] = (G_h g G_w^T)[f] in fp64, rounded once to the two-plane form; dst (halves) [f][Cin/32][Cout][4 x (8 hi | 8 lo)] ----
// One thread owns 8 consecutive input channels of one output channel: for every frequency one 32-byte piece (two 16-byte stores).
template <int MH, int MW>
__global__ __launch_bounds__(256) void wino4_pack_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, int Cout, int Cin,
                                                         unsigned* range_flag) {
  using CH = Coef<MH>;
  using CW = Coef<MW>;
  constexpr int NH = CH::N, NW = CW::N;
  bool over = false;
  const long long total = (long long)(Cin / 8) * Cout;
  const long long fstride = (long long)Cin * Cout * 2;          // halves per frequency
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int g4 = (int)(i & 3);
    const long long rn = i >> 2;                                // (k / 32) * Cout + n
    const int n = (int)(rn % Cout), kc = (int)(rn / Cout);
    const int k0 = kc * 32 + g4 * 8;
    float gk[8][9];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const float* gs = src + ((long long)n * Cin + k0 + kk) * 9;
#pragma unroll
      for (int t = 0; t < 9; ++t) gk[kk][t] = gs[t];
    }
    _Float16* d = dst + (rn * 32 + g4 * 8) * 2;
#pragma unroll
    for (int fi = 0; fi < NH; ++fi)
#pragma unroll
      for (int fj = 0; fj < NW; ++fj) {
        pk_half8 hi, lo;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          double u = 0.0;
#pragma unroll
          for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int bb = 0; bb < 3; ++bb) {
              const double cf = CH::g(fi, a) * CW::g(fj, bb);
              if (cf != 0.0) u += cf * (double)gk[kk][a * 3 + bb];
            }
          us_half h, l;
          split_f16x3((float)u, h, l, over);
          hi[kk] = h;
          lo[kk] = l;
        }
        *reinterpret_cast<pk_half8*>(d + (long long)(fi * NW + fj) * fstride) = hi;
        *reinterpret_cast<pk_half8*>(d + (long long)(fi * NW + fj) * fstride + 8) = lo;
      }
  }
  range_report(range_flag, over, kRangeWeight);
}

int grid_for(long long per_item) {
  long long blocks = (per_item + 255) / 256;
  return (int)(blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks));
}

}  // namespace

bool wino4_form_ok(int form) { return form == 44 || form == 24; }
int wino4_freqs(int form) { return form == 44 ? 36 : (form == 24 ? 24 : 0); }
void wino4_tiles(int form, int H, int W, int* th, int* tw) {
  const int mh = form / 10, mw = form % 10;
  *th = (H + mh - 1) / mh;
  *tw = (W + mw - 1) / mw;
}

hipError_t launch_wino4_input(int form, const float* x, int x_ld, float* V, int B, int H, int W, int C, const WinoGnArgs* gn, hipStream_t s) {
  if (!wino4_form_ok(form) || C % 8 != 0 || x_ld % 4 != 0 || B > 65535) return hipErrorInvalidValue;
  // 32-bit byte offsets inside one item (buffer descriptor) and 32-bit element offsets inside one (frequency, item) plane of V
  if ((long long)H * W * x_ld * 4 >= (1LL << 31) || (long long)H * W * C >= (1LL << 30)) return hipErrorInvalidValue;
  WinoGnArgs g{};
  if (gn) {
    g = *gn;
    if (!g.stats || !g.gamma || !g.beta || !g.mask || g.h_out || x_ld != C || C % kGroups != 0 || (C / kGroups) % 4 != 0) return hipErrorInvalidValue;
    if (g.mask_bmod < 1) g.mask_bmod = 1;
  }
  int th, tw;
  wino4_tiles(form, H, W, &th, &tw);
  const dim3 grid(grid_for((long long)th * tw * (C / 4)), B);
  unsigned* rf = current_range_flag();
  if (form == 44) {
    if (gn) hipLaunchKernelGGL((wino4_input_kernel<4, 4, true>), grid, dim3(256), 0, s, x, x_ld, V, B, H, W, C, g, rf);
    else hipLaunchKernelGGL((wino4_input_kernel<4, 4, false>), grid, dim3(256), 0, s, x, x_ld, V, B, H, W, C, g, rf);
  } else {
    if (gn) hipLaunchKernelGGL((wino4_input_kernel<2, 4, true>), grid, dim3(256), 0, s, x, x_ld, V, B, H, W, C, g, rf);
    else hipLaunchKernelGGL((wino4_input_kernel<2, 4, false>), grid, dim3(256), 0, s, x, x_ld, V, B, H, W, C, g, rf);
  }
  return hipGetLastError();
}

hipError_t launch_wino4_output(int form, const float* M, const float* bias, float* out, int out_ld, double* stats, int B, int H, int W, int C,
                               hipStream_t s) {
  if (!wino4_form_ok(form) || C % 4 != 0 || out_ld % 4 != 0 || C % kGroups != 0 || B > 65535) return hipErrorInvalidValue;
  int th, tw;
  wino4_tiles(form, H, W, &th, &tw);
  if ((long long)th * tw * (C / 4) >= (1LL << 31)) return hipErrorInvalidValue;      // (the kernel's 32-bit tile index)
  const dim3 grid(grid_for((long long)th * tw * (C / 4)), B);
  if (form == 44) hipLaunchKernelGGL((wino4_output_kernel<4, 4>), grid, dim3(256), 0, s, M, bias, out, out_ld, stats, B, H, W, C);
  else hipLaunchKernelGGL((wino4_output_kernel<2, 4>), grid, dim3(256), 0, s, M, bias, out, out_ld, stats, B, H, W, C);
  return hipGetLastError();
}

hipError_t launch_wino4_pack_weight(int form, const float* src, float* dst, int Cout, int Cin, hipStream_t s) {
  if (!wino4_form_ok(form) || Cin % 32 != 0) return hipErrorInvalidValue;
  const long long total = (long long)Cout * Cin / 8;
  long long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  unsigned* rf = current_range_flag();
  if (form == 44) hipLaunchKernelGGL((wino4_pack_kernel<4, 4>), dim3((int)blocks), dim3(256), 0, s, src, reinterpret_cast<_Float16*>(dst), Cout, Cin, rf);
  else hipLaunchKernelGGL((wino4_pack_kernel<2, 4>), dim3((int)blocks), dim3(256), 0, s, src, reinterpret_cast<_Float16*>(dst), Cout, Cin, rf);
  return hipGetLastError();
}

}  // namespace us
