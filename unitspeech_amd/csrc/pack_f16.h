// Bodies of the two f16x3 weight packs (Winograd-domain U = G g G^T, direct-form taps) as device functions over an explicit
// (block, number of blocks) pair: the stand-alone kernels of wino.hip / conv_igemm.hip call them with their own grid, and
// pack_table_kernel (wino.hip) runs ANY NUMBER of packs in one launch from a job table -- fine-tuning re-packs every convolution
// weight, forward and data-gradient form, after every optimiser step: 120 launches of 5-20 us were 1.3 ms of a 12 ms iteration.
#pragma once
#include "kernels.h"

namespace us {

typedef _Float16 pk_half8 __attribute__((ext_vector_type(8)));

// U as two interleaved fp16 planes, dst (halves) [f][K/32][N][4 groups x (8 hi | 8 lo)], K = Cin, N = Cout (forward) or K = Cout,
// N = Cin (dgrad: the transform of the 180-degree-rotated, channel-swapped filter).  One thread owns 8 consecutive K indices of one N
// index, i.e. for every frequency one whole 32-byte (8 hi | 8 lo) piece: 16-byte stores.
__device__ __forceinline__ void wino_pack_f16_body(const float* __restrict__ src, _Float16* __restrict__ dst, int Cout, int Cin, int dgrad,
                                                   long long blk, long long nblk, bool& over) {
  const int K = dgrad ? Cout : Cin, N = dgrad ? Cin : Cout;
  const long long total = (long long)(K / 8) * N;          // thread = (k group of 8, n), k group fastest within a 32-channel row
  const long long fstride = (long long)K * N * 2;          // halves per frequency
  for (long long i = blk * 256 + threadIdx.x; i < total; i += nblk * 256) {
    const int g4 = (int)(i & 3);                            // group of 8 inside the 32-channel chunk
    const long long rn = i >> 2;                            // (k / 32) * N + n
    const int n = (int)(rn % N), kc = (int)(rn / N);
    const int k0 = kc * 32 + g4 * 8;
    pk_half8 hi[16], lo[16];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int k = k0 + kk;
      const int co = dgrad ? k : n, ci = dgrad ? n : k;
      const float* gs = src + ((long long)co * Cin + ci) * 9;
      float g[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) g[t] = dgrad ? gs[8 - t] : gs[t];
      float gg[4][3];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        gg[0][q] = g[q];
        gg[1][q] = 0.5f * (g[q] + g[3 + q] + g[6 + q]);
        gg[2][q] = 0.5f * (g[q] - g[3 + q] + g[6 + q]);
        gg[3][q] = g[6 + q];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float u[4] = {gg[r][0], 0.5f * (gg[r][0] + gg[r][1] + gg[r][2]), 0.5f * (gg[r][0] - gg[r][1] + gg[r][2]), gg[r][2]};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          us_half h, l;
          split_f16x3(u[q], h, l, over);
          hi[r * 4 + q][kk] = h;
          lo[r * 4 + q][kk] = l;
        }
      }
    }
    _Float16* d = dst + (rn * 32 + g4 * 8) * 2;
#pragma unroll
    for (int f = 0; f < 16; ++f) {
      *reinterpret_cast<pk_half8*>(d + f * fstride) = hi[f];
      *reinterpret_cast<pk_half8*>(d + f * fstride + 8) = lo[f];
    }
  }
}

__device__ __forceinline__ int qkv_src_row_dev(int cp) {       // == qkv_src_row (kernels.h)
  if (cp < kHidden) return cp;
  const int t = cp - kHidden, h = t >> 6, w = t & 63;
  return (w < kDimHead ? kHidden : 2 * kHidden) + h * kDimHead + (w & (kDimHead - 1));
}

// direct form (bk = 32): dst (halves) [tap][Cin/32][Cout][4 groups x (8 hi | 8 lo)], same bytes and row structure as the fp32 pack.
// One thread = one 32-byte piece (8 input channels of one (tap, output channel)): two 16-byte stores.
__device__ __forceinline__ void conv_pack_f16_body(const float* __restrict__ src, _Float16* __restrict__ dst, int Cout, int Cin, int KH, int KW,
                                                   int oihw, int qkv_rows, long long blk, long long nblk, bool& over) {
  const long long total = (long long)KH * KW * Cout * (Cin / 8);
  const int nchunk = Cin / 32;
  for (long long i = blk * 256 + threadIdx.x; i < total; i += nblk * 256) {
    const int g4 = (int)(i & 3);
    long long t = i >> 2;
    int co = (int)(t % Cout); t /= Cout;
    if (qkv_rows) co = qkv_src_row_dev(co);
    const int ch = (int)(t % nchunk);
    const int tap = (int)(t / nchunk);
    const int ky = tap / KW, kx = tap % KW;
    pk_half8 hi, lo;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int ci = ch * 32 + g4 * 8 + kk;
      const long long si = oihw ? (((long long)co * Cin + ci) * KH + ky) * KW + kx : (((long long)ci * Cout + co) * KH + ky) * KW + kx;
      us_half h, l;
      split_f16x3(src[si], h, l, over);
      hi[kk] = h;
      lo[kk] = l;
    }
    _Float16* d = dst + i * 16;
    *reinterpret_cast<pk_half8*>(d) = hi;
    *reinterpret_cast<pk_half8*>(d + 8) = lo;
  }
}

// one pack of the table: kind 0 = wino_pack_f16_body (a = dgrad), kind 1 = conv_pack_f16_body (a = oihw, b = qkv_rows);
// blocks [blk0, blk0 + nblk) of the launch belong to it
struct PackJob {
  const float* src;
  _Float16* dst;
  int kind, Cout, Cin, KH, KW, a, b;
  int blk0, nblk;
};
hipError_t launch_pack_table(const PackJob* jobs_dev, int n_jobs, int total_blocks, hipStream_t s);
inline int pack_job_blocks(const PackJob& j) {      // as the stand-alone launchers size their grids
  const long long total = j.kind == 0 ? (long long)j.Cout * j.Cin / 8 : (long long)j.KH * j.KW * j.Cout * (j.Cin / 8);
  long long b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace us
