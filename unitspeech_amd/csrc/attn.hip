// Linear attention of the U-Net (unitspeech/unitspeech.py:78-96), heads=4, dim_head=32, on pixel-major qkv.
//
//   k = softmax_n(k)            over ALL n = H*W positions, unmasked                       (:91)
//   ctx[h][d][e] = sum_n k[h][d][n] v[h][e][n]                                            (:92)
//   out[h][e][n] = sum_d ctx[h][d][e] q[h][d][n] ; to_out(out)                           (:93-95)
//
// ctx is a 32x32 matrix per head, so to_out(ctx^T q) is folded into a per-item 1x1 convolution of q with
// W_eff = W_out * blockdiag(ctx^T) which the MFMA implicit-GEMM kernel then applies (with the Rezero gain and the
// residual in its epilogue).  The n-reduction is split into 128-row chunks; each chunk keeps its own column max
// (online-softmax form) and the chunks are merged with exp(m_chunk - m) weights.
#include "kernels.h"

namespace us {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kChunk = 128;
constexpr int kQkvLd = 3 * kHidden;

// grid (nchunks, B), 256 threads: wave = head.  lane (c = lane&31, hh = lane>>5) owns column c of the head and the
// rows r0 + 2j + hh, which is exactly the A/B operand layout of v_mfma_f32_32x32x2_f32 (A[i=d][k=hh], B[k=hh][j=e]).
__global__ __launch_bounds__(256) void attn_ctx_partial_kernel(const float* __restrict__ qkv, int n, float* __restrict__ part_ctx,
                                                               float* __restrict__ part_m, float* __restrict__ part_s, int nchunks) {
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int lane = threadIdx.x & 63, h = threadIdx.x >> 6;
  const int c = lane & 31, hh = lane >> 5;
  const float* base = qkv + (long long)b * n * kQkvLd;
  const int r0 = chunk * kChunk;
  const float* kp = base + kHidden + h * kDimHead + c;
  const float* vp = base + 2 * kHidden + h * kDimHead + c;

  float kv[kChunk / 2];
  float m = -INFINITY;
#pragma unroll
  for (int j = 0; j < kChunk / 2; ++j) {
    int row = r0 + 2 * j + hh;
    int rc = row < n ? row : n - 1;
    float v = kp[(long long)rc * kQkvLd];
    kv[j] = row < n ? v : -INFINITY;
    m = fmaxf(m, kv[j]);
  }
  m = fmaxf(m, __shfl_xor(m, 32));

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < kChunk / 2; ++j) {
    int row = r0 + 2 * j + hh;
    int rc = row < n ? row : n - 1;
    float p = expf(kv[j] - m);          // exp(-inf) = 0 for rows beyond n
    float v = vp[(long long)rc * kQkvLd];
    v = row < n ? v : 0.f;
    s += p;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(p, v, acc, 0, 0, 0);
  }
  s += __shfl_xor(s, 32);

  const long long blk = (long long)b * nchunks + chunk;
  float* pc = part_ctx + (blk * kHeads + h) * (kDimHead * kDimHead);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int d = (r & 3) + 8 * (r >> 2) + 4 * hh;
    pc[d * kDimHead + c] = acc[r];
  }
  if (hh == 0) {
    part_m[blk * kHidden + h * kDimHead + c] = m;
    part_s[blk * kHidden + h * kDimHead + c] = s;
  }
}

hipError_t launch_attn_ctx_partial(const float* qkv, int B, int n, float* part_ctx, float* part_m, float* part_s, int nchunks,
                                   hipStream_t s) {
  hipLaunchKernelGGL(attn_ctx_partial_kernel, dim3(nchunks, B), dim3(256), 0, s, qkv, n, part_ctx, part_m, part_s, nchunks);
  return hipGetLastError();
}

// Merge of the chunk partials, two launches:
//   stats : grid (B*heads), 1024 threads: column max M[d] and S[d] = sum_c s_c[d] exp(m_c[d] - M[d])
//   accum : grid (B*heads, nsplit), 1024 threads = (d, e): ctx[d][e] += sum_{c in split} exp(m_c[d]-M[d]) part_c[d][e] / S[d]
// (ctx zeroed by the launcher; splitting the chunk range keeps the level-0 merge, 640 chunks per item, off the critical path)
__global__ __launch_bounds__(1024) void attn_ctx_stats_kernel(const float* __restrict__ part_m, const float* __restrict__ part_s,
                                                              int nchunks, float* __restrict__ colM, float* __restrict__ colS) {
  const int b = blockIdx.x / kHeads, h = blockIdx.x % kHeads;
  const int d = threadIdx.x >> 5, e = threadIdx.x & 31;
  const float* pm = part_m + (long long)b * nchunks * kHidden + h * kDimHead + d;
  const float* ps = part_s + (long long)b * nchunks * kHidden + h * kDimHead + d;
  float M = -INFINITY;
  for (int ch = e; ch < nchunks; ch += 32) M = fmaxf(M, pm[(long long)ch * kHidden]);
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) M = fmaxf(M, __shfl_xor(M, off));
  float S = 0.f;
  for (int ch = e; ch < nchunks; ch += 32) S += ps[(long long)ch * kHidden] * expf(pm[(long long)ch * kHidden] - M);
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) S += __shfl_xor(S, off);
  if (e == 0) {
    colM[(long long)b * kHidden + h * kDimHead + d] = M;
    colS[(long long)b * kHidden + h * kDimHead + d] = S;
  }
}

__global__ __launch_bounds__(1024) void attn_ctx_accum_kernel(const float* __restrict__ part_ctx, const float* __restrict__ part_m,
                                                              const float* __restrict__ colM, const float* __restrict__ colS, int nchunks,
                                                              int per_split, float* __restrict__ ctx, float* __restrict__ split_ws) {
  const int b = blockIdx.x / kHeads, h = blockIdx.x % kHeads;
  const int d = threadIdx.x >> 5, e = threadIdx.x & 31;
  const float M = colM[(long long)b * kHidden + h * kDimHead + d], S = colS[(long long)b * kHidden + h * kDimHead + d];
  const int lo = blockIdx.y * per_split;
  int hi = lo + per_split;
  if (hi > nchunks) hi = nchunks;
  const float* pm = part_m + (long long)b * nchunks * kHidden + h * kDimHead + d;
  const float* pc = part_ctx + ((long long)b * nchunks * kHeads + h) * (kDimHead * kDimHead) + d * kDimHead + e;
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
  int ch = lo;
  for (; ch + 4 <= hi; ch += 4) {
    float m0 = pm[(long long)ch * kHidden], m1 = pm[(long long)(ch + 1) * kHidden], m2 = pm[(long long)(ch + 2) * kHidden],
          m3 = pm[(long long)(ch + 3) * kHidden];
    float c0 = pc[(long long)ch * kHeads * 1024], c1 = pc[(long long)(ch + 1) * kHeads * 1024], c2 = pc[(long long)(ch + 2) * kHeads * 1024],
          c3 = pc[(long long)(ch + 3) * kHeads * 1024];
    acc0 = fmaf(expf(m0 - M), c0, acc0);
    acc1 = fmaf(expf(m1 - M), c1, acc1);
    acc2 = fmaf(expf(m2 - M), c2, acc2);
    acc3 = fmaf(expf(m3 - M), c3, acc3);
  }
  for (; ch < hi; ++ch) acc0 = fmaf(expf(pm[(long long)ch * kHidden] - M), pc[(long long)ch * kHeads * 1024], acc0);
  const float v = ((acc0 + acc1) + (acc2 + acc3)) / S;
  // one slab per chunk range; attn_ctx_reduce_kernel adds them in a fixed order (float atomics would make ctx, and with it
  // every later activation, vary from run to run)
  float* dst = (gridDim.y == 1 ? ctx : split_ws + (long long)blockIdx.y * gridDim.x * (kDimHead * kDimHead)) +
               ((long long)b * kHeads + h) * (kDimHead * kDimHead) + d * kDimHead + e;
  *dst = v;
}

__global__ __launch_bounds__(1024) void attn_ctx_reduce_kernel(const float* __restrict__ split_ws, int nsplit, long long slab,
                                                               float* __restrict__ ctx) {
  const long long i = blockIdx.x * 1024LL + threadIdx.x;
  float acc = split_ws[i];
  for (int k = 1; k < nsplit; ++k) acc += split_ws[i + k * slab];
  ctx[i] = acc;
}

hipError_t launch_attn_ctx_finalize(const float* part_ctx, const float* part_m, const float* part_s, int B, int nchunks,
                                    float* ctx, float* colM, float* colS, float* split_ws, hipStream_t s) {
  hipLaunchKernelGGL(attn_ctx_stats_kernel, dim3(B * kHeads), dim3(1024), 0, s, part_m, part_s, nchunks, colM, colS);
  int nsplit = (nchunks + 31) / 32;
  if (nsplit > kAttnMaxSplit) nsplit = kAttnMaxSplit;
  if (nsplit < 1 || !split_ws) nsplit = 1;
  const int per_split = (nchunks + nsplit - 1) / nsplit;
  nsplit = (nchunks + per_split - 1) / per_split;
  hipLaunchKernelGGL(attn_ctx_accum_kernel, dim3(B * kHeads, nsplit), dim3(1024), 0, s, part_ctx, part_m, colM, colS, nchunks, per_split, ctx,
                     split_ws);
  if (nsplit > 1)
    hipLaunchKernelGGL(attn_ctx_reduce_kernel, dim3(B * kHeads), dim3(1024), 0, s, split_ws, nsplit,
                       (long long)B * kHeads * kDimHead * kDimHead, ctx);
  return hipGetLastError();
}

// weff[b] in the conv kernel's packed layout [1 tap][128/bk][C][bk]:
//   weff[co][h*32+d] = sum_e wout[co][h*32+e] * ctx[b][h][d][e]
__global__ __launch_bounds__(256) void attn_weff_kernel(const float* __restrict__ ctx, const float* __restrict__ wout,
                                                        float* __restrict__ weff, int C, int bk, int f16, unsigned* range_flag) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= C * kHidden) return;
  const int co = i / kHidden, hd = i % kHidden;
  const int h = hd / kDimHead, d = hd % kDimHead;
  const float* cx = ctx + ((long long)b * kHeads + h) * (kDimHead * kDimHead) + d * kDimHead;
  const float* w = wout + (long long)co * kHidden + h * kDimHead;
  float acc = 0.f;
#pragma unroll
  for (int e = 0; e < kDimHead; ++e) acc = fmaf(w[e], cx[e], acc);
  const long long o = (long long)b * C * kHidden + ((long long)(hd / bk) * C + co) * bk;
  if (!f16) {
    weff[o + hd % bk] = acc;
  } else {          // two interleaved fp16 planes (conv_igemm_kernel F16 operand format, bk = 32)
    _Float16* w16 = reinterpret_cast<_Float16*>(weff + o);
    const int k = hd % 32;
    us_half h, l;
    bool over = false;
    split_f16x3(acc, h, l, over);
    w16[(k / 8) * 16 + k % 8] = h;
    w16[(k / 8) * 16 + 8 + k % 8] = l;
    range_report(range_flag, over, kRangeWeight);
  }
}

hipError_t launch_attn_weff(const float* ctx, const float* wout, float* weff, int B, int C, int bk, hipStream_t s, bool f16) {
  if (f16 && bk != 32) return hipErrorInvalidValue;
  hipLaunchKernelGGL(attn_weff_kernel, dim3((C * kHidden + 255) / 256, B), dim3(256), 0, s, ctx, wout, weff, C, bk, f16 ? 1 : 0,
                     current_range_flag());
  return hipGetLastError();
}

}  // namespace us
