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
typedef float f32x4 __attribute__((ext_vector_type(4)));

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

// ---- merge of the chunk partials in two launches (replaces stats + accum + reduce [+ weff]) -------------------------------------
// Hierarchical online softmax: launch 1, grid (nrange, B*heads): block (r, b, h) folds the chunks of range r with the range's own
// column maxima M_r[d]: S_r[d] = sum_c s_c[d] exp(m_c[d] - M_r[d]), ctx_r[d][e] = sum_c exp(m_c[d] - M_r[d]) part_c[d][e] (fixed
// order: reproducible).  Launch 2, grid (B*heads): M = max_r M_r, S = sum_r S_r exp(M_r - M), ctx = sum_r ctx_r exp(M_r - M) / S ->
// ctx, colM, colS, and (inference) the head's slice of the folded to_out weights W_eff straight after, while ctx is still in LDS.
__global__ __launch_bounds__(1024) void attn_merge_kernel(const float* __restrict__ part_ctx, const float* __restrict__ part_m,
                                                          const float* __restrict__ part_s, int nchunks, int per_range,
                                                          float* __restrict__ r_ctx, float* __restrict__ r_m, float* __restrict__ r_s) {
  __shared__ float s_m[64][kDimHead + 1];         // column maxima of up to 64 chunks at a time
  __shared__ float s_M[kDimHead], s_S[kDimHead];
  const int bh = blockIdx.y, b = bh / kHeads, h = bh % kHeads;
  const int d = threadIdx.x >> 5, e = threadIdx.x & 31;
  const int lo = blockIdx.x * per_range;
  const int hi = min(lo + per_range, nchunks);
  const float* pm = part_m + (long long)b * nchunks * kHidden + h * kDimHead;
  const float* ps = part_s + (long long)b * nchunks * kHidden + h * kDimHead;
  const float* pc = part_ctx + ((long long)b * nchunks * kHeads + h) * (kDimHead * kDimHead) + d * kDimHead + e;
  // range maximum per column d: thread (d, e) scans chunks lo + e, lo + e + 32, ... (each row of part_m is read as 32 contiguous floats by
  // the 32 threads that share e... ) -- laid out so that lanes read contiguous addresses: lane = d
  float M = -INFINITY;
  for (int ch = lo + d; ch < hi; ch += 32) M = fmaxf(M, pm[(long long)ch * kHidden + e]);      // thread (d = chunk offset, e = column)
  // reduce over the 32 `d` rows of the block for every column e through LDS
  s_m[d][e] = M;
  __syncthreads();
  if (d == 0) {
    float mm = s_m[0][e];
#pragma unroll
    for (int k = 1; k < 32; ++k) mm = fmaxf(mm, s_m[k][e]);
    s_M[e] = mm;
  }
  __syncthreads();
  // S_r[col]: same chunk-strided scan, column = e
  {
    const float Mc = s_M[e];
    float S = 0.f;
    for (int ch = lo + d; ch < hi; ch += 32) S += ps[(long long)ch * kHidden + e] * expf(pm[(long long)ch * kHidden + e] - Mc);
    __syncthreads();
    s_m[d][e] = S;
    __syncthreads();
    if (d == 0) {
      float ss = 0.f;
#pragma unroll
      for (int k = 0; k < 32; ++k) ss += s_m[k][e];
      s_S[e] = ss;
    }
  }
  __syncthreads();
  // ctx_r[d][e]: thread (d, e) walks the chunks; the weights exp(m_c[d] - M_r[d]) of 64 chunks at a time come from LDS
  const float Md = s_M[d];
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
  for (int base = lo; base < hi; base += 64) {
    const int nb = min(64, hi - base);
    __syncthreads();
    for (int i = threadIdx.x; i < nb * kDimHead; i += 1024) {
      const int c = i >> 5, col = i & 31;
      s_m[c][col] = expf(pm[(long long)(base + c) * kHidden + col] - s_M[col]);
    }
    __syncthreads();
    // eight independent loads in flight per thread (each chunk's value is 4 KB from the next: the loop is latency, not bandwidth)
    int c = 0;
    for (; c + 8 <= nb; c += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = pc[(long long)(base + c + u) * kHeads * 1024];
      acc0 = fmaf(s_m[c][d], v[0], acc0);     acc1 = fmaf(s_m[c + 1][d], v[1], acc1);
      acc2 = fmaf(s_m[c + 2][d], v[2], acc2); acc3 = fmaf(s_m[c + 3][d], v[3], acc3);
      acc0 = fmaf(s_m[c + 4][d], v[4], acc0); acc1 = fmaf(s_m[c + 5][d], v[5], acc1);
      acc2 = fmaf(s_m[c + 6][d], v[6], acc2); acc3 = fmaf(s_m[c + 7][d], v[7], acc3);
    }
    for (; c < nb; ++c) acc0 = fmaf(s_m[c][d], pc[(long long)(base + c) * kHeads * 1024], acc0);
  }
  (void)Md;
  const long long o = ((long long)blockIdx.x * gridDim.y + bh);
  r_ctx[o * 1024 + d * kDimHead + e] = (acc0 + acc1) + (acc2 + acc3);
  if (d == 0) { r_m[o * kDimHead + e] = s_M[e]; r_s[o * kDimHead + e] = s_S[e]; }
}

__global__ __launch_bounds__(1024) void attn_final_kernel(const float* __restrict__ r_ctx, const float* __restrict__ r_m,
                                                          const float* __restrict__ r_s, int nrange, float* __restrict__ ctx,
                                                          float* __restrict__ colM, float* __restrict__ colS, const float* __restrict__ wout,
                                                          float* __restrict__ weff, int C, int bk, int f16, unsigned* range_flag) {
  __shared__ float s_ctx[kDimHead][kDimHead + 1];
  __shared__ float s_w[16][kDimHead + 1];          // exp(M_r - M) per range and column (nrange <= 16)
  __shared__ float s_M[kDimHead], s_S[kDimHead];
  // grid (B*heads, output-channel tiles of 128): every tile block re-derives the head's ctx (a few KB of range partials), tile 0 stores it
  const int bh = blockIdx.x, b = bh / kHeads, h = bh % kHeads;
  const int d = threadIdx.x >> 5, e = threadIdx.x & 31;
  const int nbh = gridDim.x;
  const bool writer = blockIdx.y == 0;
  if (d == 0) {
    float M = -INFINITY;
    for (int r = 0; r < nrange; ++r) M = fmaxf(M, r_m[((long long)r * nbh + bh) * kDimHead + e]);
    float S = 0.f;
    for (int r = 0; r < nrange; ++r) {
      const float w = expf(r_m[((long long)r * nbh + bh) * kDimHead + e] - M);
      s_w[r][e] = w;
      S += r_s[((long long)r * nbh + bh) * kDimHead + e] * w;
    }
    s_M[e] = M; s_S[e] = S;
    if (writer) {
      colM[(long long)b * kHidden + h * kDimHead + e] = M;
      colS[(long long)b * kHidden + h * kDimHead + e] = S;
    }
  }
  __syncthreads();
  float acc = 0.f;
  for (int r = 0; r < nrange; ++r) acc = fmaf(s_w[r][d], r_ctx[((long long)r * nbh + bh) * 1024 + d * kDimHead + e], acc);
  const float v = acc / s_S[d];
  if (writer) ctx[((long long)b * kHeads + h) * 1024 + d * kDimHead + e] = v;
  if (!weff) return;
  s_ctx[d][e] = v;
  __syncthreads();
  // weff[co][h*32 + dd] = sum_e wout[co][h*32 + e] * ctx[dd][e]: thread t takes (co = tile * 128 + t / 32 + 32 k, dd = t % 32)
  const int dd = threadIdx.x & 31;
  bool over = false;
  const int co_end = min(C, ((int)blockIdx.y + 1) * 128);
  // this head's 32 columns of the tile's 128 rows of W_out through LDS (one coalesced pass instead of 32 dependent loads per output)
  __shared__ float s_wo[128][kDimHead + 1];
  for (int i = threadIdx.x; i < 128 * kDimHead; i += 1024) {
    const int r = i >> 5, k = i & 31, co = blockIdx.y * 128 + r;
    s_wo[r][k] = co < C ? wout[(long long)co * kHidden + h * kDimHead + k] : 0.f;
  }
  __syncthreads();
  for (int co = blockIdx.y * 128 + (threadIdx.x >> 5); co < co_end; co += 32) {
    const float* w = s_wo[co - blockIdx.y * 128];
    float a2 = 0.f;
#pragma unroll
    for (int k = 0; k < kDimHead; ++k) a2 = fmaf(w[k], s_ctx[dd][k], a2);
    const int hd = h * kDimHead + dd;
    const long long o = (long long)b * C * kHidden + ((long long)(hd / bk) * C + co) * bk;
    if (!f16) {
      weff[o + hd % bk] = a2;
    } else {
      _Float16* w16 = reinterpret_cast<_Float16*>(weff + o);
      const int k = hd % 32;
      us_half hh_, ll_;
      split_f16x3(a2, hh_, ll_, over);
      w16[(k / 8) * 16 + k % 8] = hh_;
      w16[(k / 8) * 16 + 8 + k % 8] = ll_;
    }
  }
  range_report(range_flag, over, kRangeWeight);
}

// scratch: (kAttnMaxSplit * B * heads) * (1024 + 64) floats
hipError_t launch_attn_merge(const float* part_ctx, const float* part_m, const float* part_s, int B, int nchunks, float* ctx, float* colM,
                             float* colS, float* scratch, const float* wout, float* weff, int C, int bk, bool f16, hipStream_t s) {
  if (f16 && bk != 32) return hipErrorInvalidValue;
  int nrange = (nchunks + 63) / 64;
  if (nrange > kAttnMaxSplit) nrange = kAttnMaxSplit;
  if (nrange < 1) nrange = 1;
  const int per_range = (nchunks + nrange - 1) / nrange;
  nrange = (nchunks + per_range - 1) / per_range;
  const long long nbh = (long long)B * kHeads;
  float* r_ctx = scratch;
  float* r_m = r_ctx + (long long)kAttnMaxSplit * nbh * 1024;
  float* r_s = r_m + (long long)kAttnMaxSplit * nbh * kDimHead;
  hipLaunchKernelGGL(attn_merge_kernel, dim3(nrange, (unsigned)nbh), dim3(1024), 0, s, part_ctx, part_m, part_s, nchunks, per_range, r_ctx, r_m, r_s);
  hipLaunchKernelGGL(attn_final_kernel, dim3((unsigned)nbh, weff ? (unsigned)((C + 127) / 128) : 1u), dim3(1024), 0, s, r_ctx, r_m, r_s, nrange, ctx, colM, colS, wout, weff, C, bk,
                     f16 ? 1 : 0, current_range_flag());
  return hipGetLastError();
}

// W_total[b][n][k] = sum_j W_out[n][j] T[j][k],  T[h*32 + e][k] = sum_d ctx[b][h][d][e] W_q[h*32 + d][k]
// (out = W_out (ctx^T q) with q = W_q x, unitspeech.py:95-100, re-associated: the [n][128] tensor q is never formed).
// Block (k tile of 32, item): T's tile in LDS, then every thread owns pieces of 8 consecutive k of one output channel n: 32 bytes of the
// conv kernel's packed weight ([C/32][C][4 groups x (8 hi | 8 lo)] in the two-plane form, [C/bk][C][bk] floats otherwise).
__global__ __launch_bounds__(256) void attn_wtotal_kernel(const float* __restrict__ ctx, const float* __restrict__ wout,
                                                          const float* __restrict__ wq, float* __restrict__ wtotal, int C, int bk, int f16,
                                                          unsigned* range_flag) {
  // grid (k tile of 32, item, n tile of 32): every block re-derives its k tile of T (131 K multiply-adds) and stages its 32 rows of W_out.
  // All operands go through LDS first (coalesced loads), rows padded to 36 floats so that 4 consecutive k are one 16-byte read.
  __shared__ __attribute__((aligned(16))) float s_T[kHidden][36];
  __shared__ float s_C[kHidden][33];                                      // ctx[b] as [h*32 + d][e]
  __shared__ __attribute__((aligned(16))) float s_QW[kHidden * 36];       // W_q's k tile [h*32 + d][kk]; after T: 32 rows of W_out [nl][j] (ld 129)
  const int b = blockIdx.y, k0 = blockIdx.x * 32, n0 = blockIdx.z * 32;
  const float* cx = ctx + (long long)b * kHeads * kDimHead * kDimHead;
  for (int i = threadIdx.x; i < kHidden * 32; i += 256) {
    s_C[i >> 5][i & 31] = cx[i];
    s_QW[(i >> 5) * 36 + (i & 31)] = wq[(long long)(i >> 5) * C + k0 + (i & 31)];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kHidden * 8; i += 256) {        // (j, group of 4 k)
    const int j = i >> 3, k4 = (i & 7) * 4;
    const int h = j / kDimHead, e = j % kDimHead;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < kDimHead; ++d) {
      const float c = s_C[h * kDimHead + d][e];
      const f32x4 q = *reinterpret_cast<const f32x4*>(&s_QW[(h * kDimHead + d) * 36 + k4]);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = fmaf(c, q[t], acc[t]);
    }
    *reinterpret_cast<f32x4*>(&s_T[j][k4]) = acc;
  }
  __syncthreads();
  float* s_W = s_QW;                                               // 32 * 129 <= 128 * 36 floats
  for (int i = threadIdx.x; i < 32 * kHidden; i += 256) s_W[(i / kHidden) * (kHidden + 1) + i % kHidden] = wout[(long long)(n0 + i / kHidden) * kHidden + i % kHidden];
  __syncthreads();
  bool over = false;
  const int nl = threadIdx.x >> 3, g8 = threadIdx.x & 7;          // one output channel, 4 consecutive k
  const int n = n0 + nl;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int j = 0; j < kHidden; ++j) {
    const float wj = s_W[nl * (kHidden + 1) + j];
    const f32x4 tv = *reinterpret_cast<const f32x4*>(&s_T[j][g8 * 4]);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = fmaf(wj, tv[t], acc[t]);
  }
  float* base = wtotal + (long long)b * C * C;
  if (f16) {
    typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
    half4_t hi, lo;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      us_half h_, l_;
      split_f16x3(acc[t], h_, l_, over);
      hi[t] = h_;
      lo[t] = l_;
    }
    // piece (k tile, n, group of 8 k) = 8 hi | 8 lo halves; this thread owns half of each plane
    _Float16* d = reinterpret_cast<_Float16*>(base) + (((long long)blockIdx.x * C + n) * 4 + (g8 >> 1)) * 16 + (g8 & 1) * 4;
    *reinterpret_cast<half4_t*>(d) = hi;
    *reinterpret_cast<half4_t*>(d + 8) = lo;
    range_report(range_flag, over, kRangeWeight);
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int k = k0 + g8 * 4 + t;
      base[((long long)(k / bk) * C + n) * bk + k % bk] = acc[t];
    }
  }
}

hipError_t launch_attn_wtotal(const float* ctx, const float* wout, const float* wq, float* wtotal, int B, int C, int bk, bool f16, hipStream_t s) {
  if (C % 32 != 0 || (f16 && bk != 32) || bk <= 0 || C % bk != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(attn_wtotal_kernel, dim3(C / 32, B, C / 32), dim3(256), 0, s, ctx, wout, wq, wtotal, C, bk, f16 ? 1 : 0, current_range_flag());
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
