// Implicit-GEMM convolution for gfx950 on the exact-fp32 matrix instruction v_mfma_f32_32x32x2_f32.
//
// Serves every GEMM-shaped op of the U-Net score network (unitspeech/unitspeech.py:124-201):
//   3x3 Conv2d of Block (:48), 1x1 res_conv (:66) / to_qkv (:83) / to_out (:84, with per-item folded weights),
//   Downsample 3x3 stride 2 (:30) and Upsample ConvTranspose2d 4x4 stride 2 (:21, as 4 output-phase launches
//   of a 2x2 tap set).
// GEMM view: M = output pixels of one item (flattened sub-grid), N = Cout, K = taps * Cin.
//
// Tile: 128 (pixels) x 128 (channels) per 256-thread workgroup, 4 waves as 2x2, each wave 64x64 = 2x2 MFMA
// 32x32 blocks (64 accumulator VGPRs).  K advances in chunks of BK input channels of one tap; A (activation
// rows, gathered per tap with zero padding and the frame mask applied on the way in) and B (weights, pre-packed
// so a tile is one contiguous 128*BK slab) are staged global -> registers -> LDS, double-buffered, one barrier
// per chunk.  LDS rows are padded to BK+4 floats: the ds_read_b128 fragment reads (lane = row, 4 consecutive k)
// are then conflict-free (row*(BK+4)/4 mod 16 is a permutation of 0..15 over any 16 consecutive rows).
// k-index convention inside an 8-channel sub-step: lane half hh supplies channels 4*hh+j to MFMA j (j=0..3) on
// BOTH operands, so the pairs (j, 4+j) are summed by instruction j; the K-sum is complete, only its order differs.
#include "kernels.h"

namespace us {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TM = 128, TN = 128;

template <int BK>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(ConvArgs a) {
  constexpr int LD = BK + 4;
  constexpr int QPR = BK / 4;     // float4 per tile row
  constexpr int RPP = 256 / QPR;  // rows per pass
  constexpr int NP = TM / RPP;    // passes
  constexpr int TILE = TM * LD;   // floats per operand tile
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l32 = lane & 31, hh = lane >> 5;
  const int b = blockIdx.z;
  const int m0 = blockIdx.x * TM, n0 = blockIdx.y * TN;
  const int Ms = a.Hs * a.Ws;
  const int q = tid % QPR, r0 = tid / QPR;
  const int nchunk = a.Cin / BK;
  const int S = a.ntaps * nchunk;

  int my[NP], mx[NP];
  bool mv[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    int m = m0 + r0 + j * RPP;
    mv[j] = m < Ms;
    int mc = mv[j] ? m : 0;
    my[j] = mc / a.Ws;
    mx[j] = mc - my[j] * a.Ws;
  }
  // B rows (output channels) this thread stages; clamped for memory safety (columns >= Cout are never stored)
  int brow[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    int n = n0 + r0 + j * RPP;
    brow[j] = n < a.Cout ? n : a.Cout - 1;
  }
  const float* wt_b = a.wt + (long long)b * a.wt_bstride;
  const float* in_b = a.in + (long long)b * a.Hin * a.Win * a.in_ld;
  const float* mask_b = a.mask ? a.mask + (long long)(b % a.mask_bmod) * a.mask_ld : nullptr;

  const float* aptr[NP];
  float amul[NP];
  bool aok[NP];
  const float* wtap = nullptr;

  auto setup_tap = [&](int tap) {
    const int dy = a.dy[tap], dx = a.dx[tap];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      int iy = my[j] * a.istride + dy, ix = mx[j] * a.istride + dx;
      bool ok = mv[j] && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
      aok[j] = ok;
      int iyc = ok ? iy : 0, ixc = ok ? ix : 0;
      aptr[j] = in_b + ((long long)iyc * a.Win + ixc) * a.in_ld + q * 4;
      amul[j] = mask_b ? mask_b[ixc * a.mask_step] : 1.f;
    }
    wtap = wt_b + (long long)a.wtap[tap] * nchunk * a.Cout * BK;
  };

  f32x4 ra[NP], rb[NP];
  auto prefetch = [&](int ch) {
#pragma unroll
    for (int j = 0; j < NP; ++j) ra[j] = *reinterpret_cast<const f32x4*>(aptr[j] + ch * BK);
    const float* wb = wtap + (long long)ch * a.Cout * BK + q * 4;
#pragma unroll
    for (int j = 0; j < NP; ++j) rb[j] = *reinterpret_cast<const f32x4*>(wb + (long long)brow[j] * BK);
  };
  auto stage = [&](int buf) {
    float* As = smem + buf * 2 * TILE;
    float* Bs = As + TILE;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      f32x4 v = ra[j] * amul[j];
      if (!aok[j]) v = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(As + (r0 + j * RPP) * LD + q * 4) = v;
      *reinterpret_cast<f32x4*>(Bs + (r0 + j * RPP) * LD + q * 4) = rb[j];
    }
  };

  // Two-level accumulation: the MFMA chain (an exact fp32 fma chain) runs over kFlushK = 128 K-elements, then is
  // folded into `total`.  A single chain over K = 9*Cin (up to 18,432) would carry ~0.2*sqrt(K) ulp of rounding
  // error (19 ulp at K = 9,216); chunks of ~sqrt(K) bring it to ~3 ulp, on par with a blocked CPU sgemm.
  constexpr int kFlushSteps = 128 / BK;
  f32x16 acc[2][2], total[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; total[i][j][r] = 0.f; }
  int since_flush = 0;

  int tap_n = 0, ch_n = 0;
  setup_tap(0);
  prefetch(0);
  stage(0);
  __syncthreads();
  if (++ch_n == nchunk) { ch_n = 0; ++tap_n; }

  for (int step = 0; step < S; ++step) {
    const bool has_next = step + 1 < S;
    if (has_next) {
      if (ch_n == 0) setup_tap(tap_n);
      prefetch(ch_n);
    }
    const float* As = smem + (step & 1) * 2 * TILE + (wm * 64 + l32) * LD + hh * 4;
    const float* Bs = smem + (step & 1) * 2 * TILE + TILE + (wn * 64 + l32) * LD + hh * 4;
#pragma unroll
    for (int s = 0; s < BK / 8; ++s) {
      f32x4 af[2], bf[2];
      af[0] = *reinterpret_cast<const f32x4*>(As + s * 8);
      af[1] = *reinterpret_cast<const f32x4*>(As + 32 * LD + s * 8);
      bf[0] = *reinterpret_cast<const f32x4*>(Bs + s * 8);
      bf[1] = *reinterpret_cast<const f32x4*>(Bs + 32 * LD + s * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][j], bf[0][j], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][j], bf[1][j], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1][j], bf[0][j], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1][j], bf[1][j], acc[1][1], 0, 0, 0);
      }
    }
    if (has_next) {
      stage((step + 1) & 1);
      if (++ch_n == nchunk) { ch_n = 0; ++tap_n; }
    }
    if (++since_flush == kFlushSteps) {
      since_flush = 0;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          total[i][j] += acc[i][j];
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] += total[i][j];

  // ---- epilogue: C/D layout of the 32x32 block: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) ----
  const bool dense = (a.ostep == 1 && a.oy0 == 0 && a.ox0 == 0 && a.Hs == a.Hout && a.Ws == a.Wout);
  const float alpha = a.alpha ? *a.alpha : 1.f;
  float* out_b = a.out + (long long)b * a.Hout * a.Wout * a.out_ld;
  const float* add_b = a.add ? a.add + (long long)b * a.Hout * a.Wout * a.add_ld : nullptr;
  float gsum[2] = {0.f, 0.f}, gsq[2] = {0.f, 0.f};
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int n = n0 + wn * 64 + nb * 32 + l32;
    const bool nv = n < a.Cout;
    const float bias = (a.bias && nv) ? a.bias[n] : 0.f;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (m < Ms && nv) {
          long long pix;
          if (dense) {
            pix = m;
          } else {
            int yy = m / a.Ws, xx = m - yy * a.Ws;
            pix = (long long)(a.oy0 + yy * a.ostep) * a.Wout + a.ox0 + xx * a.ostep;
          }
          float v = acc[mb][nb][r] + bias;
          gsum[nb] += v;
          gsq[nb] += v * v;
          v *= alpha;
          if (add_b) v += add_b[pix * a.add_ld + n];
          out_b[pix * a.out_ld + n] = v;
        }
      }
    }
  }
  if (a.stats) {
    // GroupNorm(8) partial sums of the stored (pre-alpha/add) values; Cout/8 is a power of two (host-checked)
    const int cg = a.Cout / kGroups;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      float s1 = gsum[nb], s2 = gsq[nb];
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      const int seg = cg < 32 ? cg : 32;
      for (int off = 1; off < seg; off <<= 1) {
        s1 += __shfl_xor(s1, off);
        s2 += __shfl_xor(s2, off);
      }
      const int n = n0 + wn * 64 + nb * 32 + l32;
      if (hh == 0 && (l32 % seg) == 0 && n < a.Cout) {
        double* st = a.stats + ((long long)b * kGroups + n / cg) * 2;
        atomicAdd(st, (double)s1);
        atomicAdd(st + 1, (double)s2);
      }
    }
  }
}

static size_t lds_bytes(int bk) { return (size_t)2 * 2 * TM * (bk + 4) * sizeof(float); }

hipError_t conv_igemm_init() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<32>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(32));
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<16>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(16));
}

hipError_t launch_conv_igemm(const ConvArgs& a, hipStream_t s) {
  if (a.B <= 0 || a.Hs <= 0 || a.Ws <= 0) return hipSuccess;
  if ((a.bk != 16 && a.bk != 32) || a.Cin % a.bk != 0 || a.ntaps < 1 || a.ntaps > kMaxTaps) return hipErrorInvalidValue;
  if (a.in_ld % 4 != 0) return hipErrorInvalidValue;   // 16-byte vector loads of the activation rows
  if (a.stats) {
    int cg = a.Cout / kGroups;
    if (a.Cout % kGroups != 0 || (cg & (cg - 1)) != 0) return hipErrorInvalidValue;
  }
  dim3 grid((a.Hs * a.Ws + TM - 1) / TM, (a.Cout + TN - 1) / TN, a.B);
  if (a.bk == 32)
    hipLaunchKernelGGL(conv_igemm_kernel<32>, grid, dim3(256), lds_bytes(32), s, a);
  else
    hipLaunchKernelGGL(conv_igemm_kernel<16>, grid, dim3(256), lds_bytes(16), s, a);
  return hipGetLastError();
}

// ---- weight repack --------------------------------------------------------------------------------
__global__ void pack_conv_weight_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cout, int Cin, int KH, int KW,
                                        int oihw, int bk) {
  const long long total = (long long)KH * KW * Cout * Cin;
  const int nchunk = Cin / bk;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int k = (int)(i % bk);
    long long t = i / bk;
    int co = (int)(t % Cout); t /= Cout;
    int ch = (int)(t % nchunk);
    int tap = (int)(t / nchunk);
    int ci = ch * bk + k;
    int ky = tap / KW, kx = tap % KW;
    long long si = oihw ? (((long long)co * Cin + ci) * KH + ky) * KW + kx
                        : (((long long)ci * Cout + co) * KH + ky) * KW + kx;
    dst[i] = src[si];
  }
}

hipError_t launch_pack_conv_weight(const float* src, float* dst, int Cout, int Cin, int KH, int KW, bool oihw, int bk,
                                   hipStream_t s) {
  if (Cin % bk != 0) return hipErrorInvalidValue;
  long long total = (long long)KH * KW * Cout * Cin;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(blocks), dim3(256), 0, s, src, dst, Cout, Cin, KH, KW, oihw ? 1 : 0, bk);
  return hipGetLastError();
}

}  // namespace us
