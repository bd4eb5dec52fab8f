// Backward-pass kernels of the score network (fine-tune / training step: `loss.backward()` through
// `GradLogPEstimator2d`, reference finetune.py:163, unitspeech/unitspeech.py:393-405).
//
// Data gradients of every convolution reuse the forward implicit-GEMM kernel (conv_igemm.hip) on weights repacked
// with the channel roles swapped; this file holds what has no forward counterpart: the weight-gradient GEMM
// (reduction over pixels, both operands pixel-major so MFMA fragments are plain coalesced row loads), GroupNorm+Mish
// backward, the linear-attention backward pieces, column sums (bias gradients) and the small dense layers.
#include "kernels.h"

namespace us {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// d/dz [ z * tanh(softplus(z)) ]  (softplus threshold 20 as in the forward)
// v_exp_f32 / v_rcp_f32 forms (about 1 ulp each), as mish_f of the forward (ops.hip): with the library expf and two IEEE divisions the
// GroupNorm backward ran at a fifth of the HBM rate (15 ms of the 61 ms pre-training step)
__device__ __forceinline__ float mish_grad(float z) {
  if (z > 20.f) return 1.f;
  float w = __expf(z);
  float u = w * (w + 2.f);
  float th = u * __builtin_amdgcn_rcpf(u + 2.f);        // tanh(softplus(z)); v_rcp_f32 (__frcp_rn is an IEEE division sequence here)
  float sg = w * __builtin_amdgcn_rcpf(1.f + w);        // sigmoid(z) = d softplus / dz
  return th + z * (1.f - th * th) * sg;
}

// ---------------------------------------------------------------------------------------------------
// weight gradient: gw[tap][co][ci] += sum_p gy[outpix(p)][co] * x[inpix(p, tap)][ci]
// grid (pixel chunks over all items, co-tiles * ci-tiles (64x64), taps); 4 waves split the chunk's pixels,
// are summed through LDS and leave with one fp32 atomic per element.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs a) {
  __shared__ float red[64 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l32 = lane & 31, hh = lane >> 5;
  const int Ms = a.Hs * a.Ws;
  const int chunks_per_item = (Ms + a.chunk - 1) / a.chunk;
  const int b = blockIdx.x / chunks_per_item;
  const int m_lo = (blockIdx.x % chunks_per_item) * a.chunk;
  const int nci = (a.Cin + 63) / 64;
  const int co0 = (blockIdx.y / nci) * 64, ci0 = (blockIdx.y % nci) * 64;
  const int tap = blockIdx.z;
  const int dy = (int)((a.dy_bits >> (4 * tap)) & 15) - 8;
  const int dx = (int)((a.dx_bits >> (4 * tap)) & 15) - 8;
  const int wt_i = (int)((a.wtap_bits >> (4 * tap)) & 15);
  const float* gy_b = a.gy + (long long)b * a.Hout * a.Wout * a.gy_ld;
  const float* x_b = a.x + (long long)b * a.Hin * a.Win * a.x_ld;
  const bool cov0 = co0 + l32 < a.Cout, cov1 = co0 + 32 + l32 < a.Cout;
  const bool civ0 = ci0 + l32 < a.Cin, civ1 = ci0 + 32 + l32 < a.Cin;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int per_wave = a.chunk / 4;
  const int m_beg = m_lo + wave * per_wave;
  int m_end = m_beg + per_wave;
  if (m_end > Ms) m_end = Ms;
  for (int m = m_beg + hh; m < m_end + hh; m += 2) {   // both halves run the same trip count; lanes past the end feed zeros
    const bool pv = m < m_end;
    const int mc = pv ? m : m_beg;
    const int yy = mc / a.Ws, xx = mc - yy * a.Ws;
    const long long opix = (long long)(a.oy0 + yy * a.ostep) * a.Wout + a.ox0 + xx * a.ostep;
    const int iy = yy * a.istride + dy, ix = xx * a.istride + dx;
    const bool iv = pv && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
    const long long ipix = iv ? (long long)iy * a.Win + ix : 0;
    const float* gp = gy_b + opix * a.gy_ld + co0 + l32;
    const float* xp = x_b + ipix * a.x_ld + ci0 + l32;
    float a0 = (pv && cov0) ? gp[0] : 0.f, a1 = (pv && cov1) ? gp[32] : 0.f;
    float b0 = (iv && civ0) ? xp[0] : 0.f, b1 = (iv && civ1) ? xp[32] : 0.f;
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
  }
  // cross-wave sum: element (row = co, col = ci) lives at red[row*64 + col]
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh, col = j * 32 + l32;
            if (w == 0) red[row * 64 + col] = acc[i][j][r];
            else red[row * 64 + col] += acc[i][j][r];
          }
    }
    __syncthreads();
  }
  float* gw = a.gw + (long long)b * a.gw_bstride + (long long)wt_i * a.Cout * a.Cin;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    int row = i >> 6, col = i & 63;
    if (co0 + row < a.Cout && ci0 + col < a.Cin) atomicAdd(&gw[(long long)(co0 + row) * a.Cin + ci0 + col], red[i]);
  }
}

// LDS-staged form for channel counts that are multiples of 128 (every conv of the full-size U-Net except the 2-channel first
// layer): workgroup tile 128 (co) x 128 (ci), 4 waves of 64 x 64, pixels in stages of 32 rows staged by LDS-DMA
// (`buffer_load_dwordx4 ... lds`, one wave instruction = 2 pixel rows x 128 channels, out-of-image taps and rows past the
// chunk read zeros through the descriptor's range check), double-buffered.  32 FLOP per LDS byte instead of one global load
// per MFMA operand.
// Grid order of the LDS-staged weight-gradient kernels.  The taps of one (pixel chunk, channel tile) read the same gy rows and the same
// x rows shifted by a pixel or a row, so they should run back to back on ONE XCD (workgroups are dealt round-robin over the 8 XCDs in
// linear order, each XCD with its own L2): with the tap as the slowest grid index every tap re-streamed both tensors from memory
// (nine times 460 MB per level-0 convolution at 32 crops).  Linear id L -> XCD L % 8, position j = L / 8 on it; consecutive j take the
// taps of one (chunk, tile), then the next (chunk, tile) of that XCD.
struct WgIdx { int bx, by, tap; };
__device__ __forceinline__ WgIdx wgrad_index() {
  WgIdx r{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
  const unsigned nxy = gridDim.x * gridDim.y, nt = gridDim.z;
  if ((nxy & 7u) == 0) {
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned xcd = lin & 7u, j = lin >> 3;
    const unsigned rest = (j / nt) * 8u + xcd;      // < nxy because (nxy / 8) * nt positions per XCD
    r.tap = (int)(j % nt);
    r.bx = (int)(rest % gridDim.x);
    r.by = (int)(rest / gridDim.x);
  }
  return r;
}

constexpr int kWgKP = 32;
__device__ __forceinline__ void wg_blds16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff_bytes, float* lds_dst_wave_uniform) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst_wave_uniform, 16, (int)voff_bytes, 0, 0, 0);
}

__global__ __launch_bounds__(256, 2) void wgrad_lds_kernel(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float wsm[];      // 2 stages x (A [32][128] + B [32][128])
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l32 = lane & 31, hh = lane >> 5;
  const int Ms = a.Hs * a.Ws;
  const int chunks_per_item = (Ms + a.chunk - 1) / a.chunk;
  const WgIdx wi = wgrad_index();
  const int b = wi.bx / chunks_per_item;
  const int m_lo = (wi.bx % chunks_per_item) * a.chunk;
  int m_hi = m_lo + a.chunk;
  if (m_hi > Ms) m_hi = Ms;
  const int nci = a.Cin / 128;
  const int co0 = (wi.by / nci) * 128, ci0 = (wi.by % nci) * 128;
  const int tap = wi.tap;
  const int dy = (int)((a.dy_bits >> (4 * tap)) & 15) - 8;
  const int dx = (int)((a.dx_bits >> (4 * tap)) & 15) - 8;
  const int wt_i = (int)((a.wtap_bits >> (4 * tap)) & 15);
  const unsigned gy_bytes = (unsigned)a.Hout * (unsigned)a.Wout * (unsigned)a.gy_ld * 4u;
  const unsigned x_bytes = (unsigned)a.Hin * (unsigned)a.Win * (unsigned)a.x_ld * 4u;
  const __amdgpu_buffer_rsrc_t rsrc_g =
      __builtin_amdgcn_make_buffer_rsrc((void*)(a.gy + (long long)b * a.Hout * a.Wout * a.gy_ld), 0, (int)gy_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_x =
      __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (long long)b * a.Hin * a.Win * a.x_ld), 0, (int)x_bytes, 0x00020000);

  // this lane's DMA role: instruction j of this wave covers stage rows wave*8 + 2j + (lane>>5), channels (lane&31)*4..+3
  const int lrow = lane >> 5, lch = (lane & 31) * 4;
  auto dma = [&](int stage_m0, int buf) {
    float* As = wsm + buf * (2 * kWgKP * 128);
    float* Bs = As + kWgKP * 128;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = wave * 8 + 2 * j;
      const int m = stage_m0 + r + lrow;
      unsigned goff = gy_bytes, xoff = x_bytes;      // out of range: zeros
      if (m < m_hi) {
        const int yy = m / a.Ws, xx = m - yy * a.Ws;
        goff = ((unsigned)((a.oy0 + yy * a.ostep) * a.Wout + a.ox0 + xx * a.ostep) * (unsigned)a.gy_ld + (unsigned)(co0 + lch)) * 4u;
        const int iy = yy * a.istride + dy, ix = xx * a.istride + dx;
        if ((unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win)
          xoff = ((unsigned)(iy * a.Win + ix) * (unsigned)a.x_ld + (unsigned)(ci0 + lch)) * 4u;
      }
      wg_blds16(rsrc_g, goff, As + r * 128);
      wg_blds16(rsrc_x, xoff, Bs + r * 128);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nstage = (m_hi - m_lo + kWgKP - 1) / kWgKP;
  dma(m_lo, 0);
  __syncthreads();
  for (int st = 0; st < nstage; ++st) {
    if (st + 1 < nstage) dma(m_lo + (st + 1) * kWgKP, (st + 1) & 1);
    const float* As = wsm + (st & 1) * (2 * kWgKP * 128);
    const float* Bs = As + kWgKP * 128;
    // lane half hh takes stage rows 16*hh + k: any pairing of the 32 rows into 16 two-row MFMA steps gives the same sum
    const float* ap = As + (16 * hh) * 128 + wm * 64 + l32;
    const float* bp = Bs + (16 * hh) * 128 + wn * 64 + l32;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float a0 = ap[k * 128], a1 = ap[k * 128 + 32];
      const float b0 = bp[k * 128], b1 = bp[k * 128 + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }
  float* gw = a.gw + (long long)b * a.gw_bstride + (long long)wt_i * a.Cout * a.Cin;
  const bool single = chunks_per_item == 1 && (a.B == 1 || a.gw_bstride != 0);    // the only writer of these elements
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        const int ci = ci0 + wn * 64 + j * 32 + l32;
        float* dst = gw + (long long)co * a.Cin + ci;
        if (single) *dst = a.overwrite ? acc[i][j][r] : *dst + acc[i][j][r];
        else atomicAdd(dst, acc[i][j][r]);
      }
}

// f16x3 form of the LDS-staged kernel (same grid, tile and arguments): the reduction runs over PIXELS, so the fp16 MFMA's eight
// consecutive k values of a lane are eight pixels of one channel -- a transpose of the pixel-major activations.  A stage of 16
// pixels x 128 channels per operand is DMA'd to LDS as it lies in memory (fp32, [pixel][channel]); the workgroup then converts it
// ONCE into the fragment image (per channel 64 bytes: hi plane of the 16 pixels, lo plane; 16-byte chunks XOR-swizzled by
// (channel >> 2) & 3 so that both the converting writes and the fragment reads spread over the banks) and every wave reads its
// fragments from there with ds_read_b128: each value is split once per workgroup, not once per wave that uses it.
// The gradient operand is scaled by an exact, per-launch power of two (below).  hi = fp16(v), lo = fp16((v - hi) * 2^11);
// C = C_hh + 2^-11 C_x as in conv_igemm.hip.
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
constexpr int kWgKP16 = 16;

template <bool CARRIED>
__global__ __launch_bounds__(256, 3) void wgrad_f16_kernel(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float wsm[];      // raw: 2 stages x (A [16][128] + B [16][128]) fp32; then the fragment image
  constexpr int RAW = 2 * kWgKP16 * 128;                           // floats per raw stage (both operands)
  float* cimg = wsm + 2 * RAW;                                     // [2 operands][128 channels][16 floats = 64 bytes]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l32 = lane & 31, hh = lane >> 5;
  const int Ms = a.Hs * a.Ws;
  const int chunks_per_item = (Ms + a.chunk - 1) / a.chunk;
  const WgIdx wi = wgrad_index();
  // Pixel range of this workgroup.  vchunk: positions of the B items laid end to end, so that a workgroup of a deep level (220 pixels
  // per crop, 2,304 tile-taps) sums over many crops in registers before its 64 KB of atomic adds -- at 32 crops those were 0.9 ms of
  // every 1024-channel launch (one workgroup per crop, tile and tap: 1.2 GB of atomics against the chip's ~1.3 TB/s).
  const bool vmode = a.vchunk > 0;
  const int b = vmode ? 0 : wi.bx / chunks_per_item;
  const int m_lo = vmode ? wi.bx * a.vchunk : (wi.bx % chunks_per_item) * a.chunk;
  const int m_end = vmode ? a.B * Ms : Ms;
  int m_hi = m_lo + (vmode ? a.vchunk : a.chunk);
  if (m_hi > m_end) m_hi = m_end;
  const int nci = a.Cin / 128;
  const int co0 = (wi.by / nci) * 128, ci0 = (wi.by % nci) * 128;
  const int tap = wi.tap;
  const int dy = (int)((a.dy_bits >> (4 * tap)) & 15) - 8;
  const int dx = (int)((a.dx_bits >> (4 * tap)) & 15) - 8;
  const int wt_i = (int)((a.wtap_bits >> (4 * tap)) & 15);
  const unsigned gy_item = (unsigned)a.Hout * (unsigned)a.Wout * (unsigned)a.gy_ld * 4u, x_item = (unsigned)a.Hin * (unsigned)a.Win * (unsigned)a.x_ld * 4u;
  const unsigned gy_bytes = vmode ? gy_item * (unsigned)a.B : gy_item;      // vmode: one descriptor over all items (< 2^31 bytes, host-checked)
  const unsigned x_bytes = vmode ? x_item * (unsigned)a.B : x_item;
  const float* gy_b = a.gy + (long long)b * a.Hout * a.Wout * a.gy_ld;
  const __amdgpu_buffer_rsrc_t rsrc_g = __builtin_amdgcn_make_buffer_rsrc((void*)gy_b, 0, (int)gy_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_x =
      __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (long long)b * a.Hin * a.Win * a.x_ld), 0, (int)x_bytes, 0x00020000);

  // ---- scale of the gradient operand.  Gradients are small (a mean-reduced loss puts dL/dy around 1e-5, below fp16's normal range), so
  // gy is multiplied by the power of two that brings the largest magnitude of the WHOLE tensor (a.gy_amax, taken exactly by
  // wgrad_amax_kernel just before this launch) to [2^14, 2^15): exact, undone on the accumulators at the end, and no scaled value can
  // leave the fp16 range.  (Until round 3 every workgroup estimated the maximum of its own slice from a sample of its rows with 2^5 of
  // headroom and clamped what the sample missed: a peaked gradient, e.g. behind the attention softmax, was saturated silently.)
  const float mx = a.gy_amax ? *a.gy_amax : 0.f;
  // 2^(14 - floor(log2 mx)), kept inside the normal range; an all-zero (or non-finite: it then propagates) tensor is left unscaled
  int e2 = (int)((__float_as_uint(mx) >> 23) & 255u) - 127;
  int se = 14 - e2;
  if (!(mx > 0.f) || e2 > 127) se = 0;
  se = se < -100 ? -100 : (se > 100 ? 100 : se);
  const float g_scale = __uint_as_float((unsigned)(127 + se) << 23), g_unscale = __uint_as_float((unsigned)(127 - se) << 23);

  // DMA role as in wgrad_lds_kernel: instruction j of this wave covers stage rows wave*4 + 2j + (lane>>5), channels (lane&31)*4..+3
  const int lrow = lane >> 5, lch = (lane & 31) * 4;
  // The coordinates of a lane's two rows advance by 16 pixels per stage.  Recomputing them from the pixel index (two divisions and six
  // 32-bit multiplies per row: 36 quarter-rate integer instructions per stage, ~580 issue cycles beside 384 cycles of MFMA) was the largest
  // VALU item of the loop; they are carried instead: position in the row, row, and the two byte offsets, moved by wave-uniform increments
  // (one wrap per step at most: Ws >= 16; narrower grids keep the division form).
  constexpr bool carried = CARRIED;      // (the launcher: Ws >= 16)
  int s_pos[2];          // (row << 16) | position in the row
  unsigned s_g[2], s_x[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int vm = m_lo + wave * 4 + 2 * j + lrow;
    const int bi = vm / Ms, m = vm - bi * Ms;        // (bi = 0 outside vmode)
    const int yy = m / a.Ws, xx = m - yy * a.Ws;
    s_pos[j] = (yy << 16) | xx;
    s_g[j] = (unsigned)bi * gy_item + ((unsigned)((a.oy0 + yy * a.ostep) * a.Wout + a.ox0 + xx * a.ostep) * (unsigned)a.gy_ld + (unsigned)(co0 + lch)) * 4u;
    // (wraps below zero where the tap leaves the image: used only where it does not)
    s_x[j] = (unsigned)bi * x_item + ((unsigned)((yy * a.istride + dy) * a.Win + xx * a.istride + dx) * (unsigned)a.x_ld + (unsigned)(ci0 + lch)) * 4u;
  }
  const unsigned g_dx = (unsigned)(kWgKP16 * a.ostep * a.gy_ld) * 4u, x_dx = (unsigned)(kWgKP16 * a.istride * a.x_ld) * 4u;
  const unsigned g_row = (unsigned)((a.ostep * a.Wout - a.Ws * a.ostep) * a.gy_ld) * 4u, x_row = (unsigned)((a.istride * a.Win - a.Ws * a.istride) * a.x_ld) * 4u;
  const unsigned g_item = gy_item - (unsigned)(a.Hs * a.ostep * a.Wout * a.gy_ld) * 4u, x_itemd = x_item - (unsigned)(a.Hs * a.istride * a.Win * a.x_ld) * 4u;
  auto dma = [&](int stage_m0, int buf) {
    float* As = wsm + buf * RAW;
    float* Bs = As + kWgKP16 * 128;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = wave * 4 + 2 * j;
      unsigned goff = gy_bytes, xoff = x_bytes;      // out of range: zeros
      if (carried) {
        const int iy = __mul24(s_pos[j] >> 16, a.istride) + dy, ix = __mul24(s_pos[j] & 0xffff, a.istride) + dx;      // (full-rate 24-bit multiplies)
        const bool pv = stage_m0 + r + lrow < m_hi;
        goff = pv ? s_g[j] : gy_bytes;
        xoff = (pv && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win) ? s_x[j] : x_bytes;
        s_pos[j] += kWgKP16; s_g[j] += g_dx; s_x[j] += x_dx;
        if ((s_pos[j] & 0xffff) >= a.Ws) {
          s_pos[j] += 0x10000 - a.Ws; s_g[j] += g_row; s_x[j] += x_row;
          if ((s_pos[j] >> 16) >= a.Hs) { s_pos[j] -= a.Hs << 16; s_g[j] += g_item; s_x[j] += x_itemd; }
        }
      } else {
        const int vm = stage_m0 + r + lrow;
        if (vm < m_hi) {
          const int bi = vm / Ms, m = vm - bi * Ms;    // (bi = 0 outside vmode)
          const int yy = m / a.Ws, xx = m - yy * a.Ws;
          goff = (unsigned)bi * gy_item + ((unsigned)((a.oy0 + yy * a.ostep) * a.Wout + a.ox0 + xx * a.ostep) * (unsigned)a.gy_ld + (unsigned)(co0 + lch)) * 4u;
          const int iy = yy * a.istride + dy, ix = xx * a.istride + dx;
          if ((unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win)
            xoff = (unsigned)bi * x_item + ((unsigned)(iy * a.Win + ix) * (unsigned)a.x_ld + (unsigned)(ci0 + lch)) * 4u;
        }
      }
      wg_blds16(rsrc_g, goff, As + r * 128);
      wg_blds16(rsrc_x, xoff, Bs + r * 128);
    }
  };
  // conversion role: unit u = (operand, channel, pixel octet); thread t takes (operand 0, channel t & 127, octet t >> 7) and the same
  // of operand 1.  16-byte chunk j of a channel's 64 bytes (hi octet 0, hi octet 1, lo octet 0, lo octet 1) sits at j ^ ((c >> 2) & 3).
  const int cv_c = tid & 127, cv_o = tid >> 7;
  const int cv_sw = (cv_c >> 2) & 3;
  auto convert = [&](int buf) {
#pragma unroll
    for (int op = 0; op < 2; ++op) {
      const float* src = wsm + buf * RAW + op * (kWgKP16 * 128) + (cv_o * 8) * 128 + cv_c;
      const float sc = op == 0 ? g_scale : 1.f;
      half8_t hi, lo;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        // no clamp: the scaled gradient cannot leave the fp16 range (exact maximum above); an ACTIVATION beyond it becomes an infinity
        // here as in the forward convolution that consumed the same tensor, which is where the range event was reported (kernels.h)
        const float v = src[k * 128] * sc;
        const _Float16 h = (_Float16)v;
        hi[k] = h;
        lo[k] = (_Float16)((v - (float)h) * 2048.f);
      }
      float* row = cimg + (op * 128 + cv_c) * 16;
      *reinterpret_cast<half8_t*>(row + ((cv_o ^ cv_sw) << 2)) = hi;
      *reinterpret_cast<half8_t*>(row + (((2 + cv_o) ^ cv_sw) << 2)) = lo;
    }
  };

  f32x16 acc[2][2], accx[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; accx[i][j][r] = 0.f; }

  const int nstage = (m_hi - m_lo + kWgKP16 - 1) / kWgKP16;
  // Two raw buffers at three workgroups per CU (48 KB each) measured better than three buffers (two stages in flight) at two
  // workgroups per CU: pre-training step 79.2 vs 88.3 ms (fp32 MFMA form: 85.4).
  dma(m_lo, 0);
  for (int st = 0; st < nstage; ++st) {
    // this wave's pieces of stage st have landed; after the barrier everybody's have, and everybody has read the image of st-1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#ifndef US_WGRAD_ABL
#define US_WGRAD_ABL 0      // timing ablations (wrong results): 1 conversion only at stage 0, 2 no fragment reads / MFMAs, 4 no loads in the loop
#endif
    if (st + 1 < nstage && !(US_WGRAD_ABL & 4)) dma(m_lo + (st + 1) * kWgKP16, (st + 1) & 1);      // its buffer was converted at stage st-1
    if (!(US_WGRAD_ABL & 1) || st == 0) convert(st & 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (US_WGRAD_ABL & 2) continue;
    half8_t ah[2], al[2], bh[2], bl[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ca = wm * 64 + i * 32 + l32, cb = wn * 64 + i * 32 + l32;
      const float* ra = cimg + ca * 16;
      const float* rb = cimg + (128 + cb) * 16;
      const int sa = (ca >> 2) & 3, sb = (cb >> 2) & 3;
      ah[i] = *reinterpret_cast<const half8_t*>(ra + ((hh ^ sa) << 2));
      al[i] = *reinterpret_cast<const half8_t*>(ra + (((2 + hh) ^ sa) << 2));
      bh[i] = *reinterpret_cast<const half8_t*>(rb + ((hh ^ sb) << 2));
      bl[i] = *reinterpret_cast<const half8_t*>(rb + (((2 + hh) ^ sb) << 2));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], accx[i][j], 0, 0, 0);
        accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], accx[i][j], 0, 0, 0);
      }
  }
  float* gw = a.gw + (long long)b * a.gw_bstride + (long long)wt_i * a.Cout * a.Cin;
  const bool single = vmode ? gridDim.x == 1 : (chunks_per_item == 1 && (a.B == 1 || a.gw_bstride != 0));    // the only writer of these elements
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        const int ci = ci0 + wn * 64 + j * 32 + l32;
        float* dst = gw + (long long)co * a.Cin + ci;
        const float v = __builtin_fmaf(accx[i][j][r], 0x1p-11f, acc[i][j][r]) * g_unscale;
        if (single) *dst = a.overwrite ? v : *dst + v;
        else atomicAdd(dst, v);
      }
}

// max |gy| over an [rows][C] window of a pixel-major tensor (ld floats per row), as float bits into *out (zeroed by the caller):
// non-negative floats order like their bit patterns, so one atomicMax per wave does it; an infinity wins, a NaN is skipped (fmaxf).
__global__ __launch_bounds__(256) void wgrad_amax_kernel(const float* __restrict__ g, int ld, long long rows, int C, unsigned* __restrict__ out) {
  const int C4 = C >> 2;
  const long long total = rows * C4;
  float mx = 0.f;
  // (at most 512 blocks, launch_wgrad_amax: four 16-byte loads in flight per thread, or a level-0 tensor of a pre-training batch is read at
  // a fraction of the HBM rate)
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += 4 * stride) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long iu = i + u * stride;
      const long long il = iu < total ? iu : i;
      const long long r = ld == C ? 0 : il / C4;           // (a dense tensor is one long row)
      const long long e = ld == C ? il * 4 : r * ld + (il - r * C4) * 4;
      v[u] = *reinterpret_cast<const f32x4*>(g + e);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[u][0]), fabsf(v[u][1])), fmaxf(fabsf(v[u][2]), fabsf(v[u][3]))));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  // one atomic per block on the ONE result word (they serialise in the L2: 4,096 of them per launch were tens of microseconds)
  __shared__ float s_mx[4];
  if ((threadIdx.x & 63) == 0) s_mx[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    mx = fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]));
    if (mx > 0.f) atomicMax(out, __float_as_uint(mx));
  }
}

hipError_t launch_wgrad_amax(const float* g, int ld, long long rows, int C, float* out, hipStream_t s) {
  if (C % 4 != 0 || ld % 4 != 0) return hipErrorInvalidValue;
  const long long total = rows * (C / 4);
  long long blocks = (total + 256 * 8 - 1) / (256 * 8);
  if (blocks < 1) blocks = 1;
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(wgrad_amax_kernel, dim3((unsigned)blocks), dim3(256), 0, s, g, ld, rows, C, reinterpret_cast<unsigned*>(out));
  return hipGetLastError();
}

static bool wgrad_use_lds(const WgradArgs& a) {
  static int use_lds = -1;
  if (use_lds < 0) { const char* e = getenv("US_WGRAD_LDS"); use_lds = e ? atoi(e) : 1; }
  return use_lds && a.Cout % 128 == 0 && a.Cin % 128 == 0 && a.gy_ld % 4 == 0 && a.x_ld % 4 == 0 &&
         (long long)a.Hout * a.Wout * a.gy_ld * 4 < (1LL << 31) && (long long)a.Hin * a.Win * a.x_ld * 4 < (1LL << 31);
}

bool launch_wgrad_single_writer(const WgradArgs& a) {
  const int Ms = a.Hs * a.Ws;
  return wgrad_use_lds(a) && a.chunk >= Ms && (a.B == 1 || a.gw_bstride != 0);
}

hipError_t launch_wgrad(const WgradArgs& a, hipStream_t s) {
  if (a.B <= 0 || a.Hs <= 0 || a.Ws <= 0) return hipSuccess;
  if (a.chunk % 8 != 0 || a.chunk <= 0) return hipErrorInvalidValue;
  const int Ms = a.Hs * a.Ws;
  if (a.overwrite && !launch_wgrad_single_writer(a)) return hipErrorInvalidValue;
  if (wgrad_use_lds(a)) {
    static bool attr_set = false;
    const int lds = 2 * 2 * kWgKP * 128 * (int)sizeof(float);
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) return e;
      attr_set = true;
    }
    dim3 g2(a.B * ((Ms + a.chunk - 1) / a.chunk), (a.Cout / 128) * (a.Cin / 128), a.ntaps);
    static int use_f16 = -1;       // US_WGRAD_F16=0: the exact-fp32 MFMA form
    if (use_f16 < 0) { const char* e = getenv("US_WGRAD_F16"); use_f16 = e ? atoi(e) : 1; }
    if (use_f16 && !a.exact) {
      if (!a.gy_amax) return hipErrorInvalidValue;      // the f16x3 form needs the exact maximum of gy (launch_wgrad_amax)
      WgradArgs a2 = a;
      static int div_addr = -1;      // US_WGRAD_DIV=1: the per-stage division form of the DMA addresses (A/B)
      if (div_addr < 0) { const char* e = getenv("US_WGRAD_DIV"); div_addr = e ? atoi(e) : 0; }
      a2.div_addr = div_addr;
      static int target_wgs = -1;
      if (target_wgs < 0) { const char* e = getenv("US_WGRAD_WGS"); target_wgs = e ? atoi(e) : 1536; }
      const long long V = (long long)a.B * Ms;
      if (a.B > 1 && a.gw_bstride == 0 && !a.overwrite && (long long)a.B * a.Hout * a.Wout * a.gy_ld * 4 < (1LL << 31) &&
          (long long)a.B * a.Hin * a.Win * a.x_ld * 4 < (1LL << 31) && V < (1LL << 30)) {
        // one pixel range over all items, cut so that ~target workgroups result (never finer than the per-item rule would cut one item)
        const long long others = (long long)g2.y * g2.z;
        long long nch = (target_wgs + others - 1) / others;
        if (nch < 1) nch = 1;
        long long vc = (V + nch - 1) / nch;
        if (vc < a.chunk) vc = a.chunk;
        vc = (vc + 15) / 16 * 16;
        a2.vchunk = (int)vc;
        g2.x = (unsigned)((V + vc - 1) / vc);
      }
      const int lds16 = (2 * 2 * kWgKP16 * 128 + 2 * 128 * 16) * (int)sizeof(float);
      static bool attr16_set = false;
      if (!attr16_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_f16_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds16);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_f16_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds16);
        if (e != hipSuccess) return e;
        attr16_set = true;
      }
      if (a2.Ws >= kWgKP16 && a2.Ws < 0x8000 && a2.Hs < 0x8000 && !a2.div_addr) hipLaunchKernelGGL(wgrad_f16_kernel<true>, g2, dim3(256), lds16, s, a2);
      else hipLaunchKernelGGL(wgrad_f16_kernel<false>, g2, dim3(256), lds16, s, a2);
      return hipGetLastError();
    }
    hipLaunchKernelGGL(wgrad_lds_kernel, g2, dim3(256), lds, s, a);
    return hipGetLastError();
  }
  dim3 grid(a.B * ((Ms + a.chunk - 1) / a.chunk), ((a.Cout + 63) / 64) * ((a.Cin + 63) / 64), a.ntaps);
  hipLaunchKernelGGL(wgrad_kernel, grid, dim3(256), 0, s, a);
  return hipGetLastError();
}

// packed [KH*KW][Cout][Cin] -> reference layout (Conv2d [Cout][Cin][KH][KW] or ConvTranspose2d [Cin][Cout][KH][KW])
__global__ void unpack_wgrad_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cout, int Cin, int taps, int oihw,
                                    const float* __restrict__ scale) {
  const float k = scale ? scale[0] : 1.f;          // the inverse of the backward pass's loss scale (a power of two: exact)
  const long long total = (long long)taps * Cout * Cin;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int t = (int)(i % taps);
    long long r = i / taps;
    int c2 = (int)(r % (oihw ? Cin : Cout));
    int c1 = (int)(r / (oihw ? Cin : Cout));
    int co = oihw ? c1 : c2, ci = oihw ? c2 : c1;
    dst[i] = src[((long long)t * Cout + co) * Cin + ci] * k;
  }
}

// Conv2d layout without the four 64-bit divisions per element of the generic kernel: block row = output channel, thread = input channel,
// the taps in a register loop (reads coalesced over ci per tap, 4 * taps contiguous bytes written per thread)
template <int TAPS>
__global__ __launch_bounds__(256) void unpack_wgrad_oihw_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cout, int Cin,
                                                                const float* __restrict__ scale, int vec_ok) {
  const float k = scale ? scale[0] : 1.f;
  const int co = blockIdx.y;
  const int ci0 = blockIdx.x * 256;
  const int ci = ci0 + threadIdx.x;
  const int nci = Cin - ci0 < 256 ? Cin - ci0 : 256;          // channels of this block
  if (TAPS > 1 && vec_ok && (nci & 3) == 0 && (Cin & 3) == 0) {      // (vec_ok: dst is 16-byte aligned)
    // the block's nci * TAPS results are one contiguous stretch of dst: turned around through LDS and written as 16-byte vectors (a thread's
    // TAPS floats at a 36-byte stride were 4-byte stores into 18 cache lines per wave instruction: 1 TB/s on 476 MB per iteration)
    __shared__ float st[256 * TAPS];
    if (ci < Cin) {
#pragma unroll
      for (int t = 0; t < TAPS; ++t) st[threadIdx.x * TAPS + t] = src[((long long)t * Cout + co) * Cin + ci] * k;
    }
    __syncthreads();
    f32x4* d4 = reinterpret_cast<f32x4*>(dst + ((long long)co * Cin + ci0) * TAPS);
    const f32x4* s4 = reinterpret_cast<const f32x4*>(st);
    for (int i = threadIdx.x; i < nci * TAPS / 4; i += 256) d4[i] = s4[i];
    return;
  }
  if (ci >= Cin) return;
  float v[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t) v[t] = src[((long long)t * Cout + co) * Cin + ci];
  float* d = dst + ((long long)co * Cin + ci) * TAPS;
#pragma unroll
  for (int t = 0; t < TAPS; ++t) d[t] = v[t] * k;
}

hipError_t launch_unpack_wgrad(const float* src, float* dst, int Cout, int Cin, int taps, bool oihw, hipStream_t s, const float* scale) {
  static const bool generic = [] { const char* p = getenv("US_UNPACK_GENERIC"); return p && atoi(p) != 0; }();
  if (!generic && oihw && (taps == 9 || taps == 1) && Cout <= 65535) {
    const dim3 grid((Cin + 255) / 256, Cout);
    const int vec_ok = (reinterpret_cast<uintptr_t>(dst) & 15) == 0 ? 1 : 0;
    if (taps == 9) hipLaunchKernelGGL(unpack_wgrad_oihw_kernel<9>, grid, dim3(256), 0, s, src, dst, Cout, Cin, scale, vec_ok);
    else hipLaunchKernelGGL(unpack_wgrad_oihw_kernel<1>, grid, dim3(256), 0, s, src, dst, Cout, Cin, scale, vec_ok);
    return hipGetLastError();
  }
  long long total = (long long)taps * Cout * Cin;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(blocks), dim3(256), 0, s, src, dst, Cout, Cin, taps, oihw ? 1 : 0, scale);
  return hipGetLastError();
}

// Repack for the data gradient: the dgrad of a conv is a conv with channel roles swapped.
//   dst[tap][Cout/bk][Cin][bk] (GEMM-N = Cin, GEMM-K = Cout); src Conv2d OIHW or ConvTranspose2d IOHW
__global__ void pack_dgrad_weight_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cout, int Cin, int KH, int KW,
                                         int oihw, int bk) {
  const long long total = (long long)KH * KW * Cout * Cin;
  const int nchunk = Cout / bk;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int k = (int)(i % bk);
    long long t = i / bk;
    int ci = (int)(t % Cin); t /= Cin;
    int ch = (int)(t % nchunk);
    int tap = (int)(t / nchunk);
    int co = ch * bk + k;
    int ky = tap / KW, kx = tap % KW;
    long long si = oihw ? (((long long)co * Cin + ci) * KH + ky) * KW + kx
                        : (((long long)ci * Cout + co) * KH + ky) * KW + kx;
    dst[i] = src[si];
  }
}

hipError_t launch_pack_dgrad_weight(const float* src, float* dst, int Cout, int Cin, int KH, int KW, bool oihw, int bk,
                                    hipStream_t s) {
  if (Cout % bk != 0) return hipErrorInvalidValue;
  long long total = (long long)KH * KW * Cout * Cin;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_dgrad_weight_kernel, dim3(blocks), dim3(256), 0, s, src, dst, Cout, Cin, KH, KW, oihw ? 1 : 0, bk);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// column sums: out[c] += scale * sum over B*n rows of g[row][c]       (bias gradients)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ g, int ld, long long rows, int C, const float* __restrict__ scale,
                                                     float* __restrict__ out) {
  const int c = blockIdx.y * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  __shared__ float red[4][64];
  float acc = 0.f;
  if (c < C) {
    // four loads in flight per thread (one dependent load per iteration made these reductions 16-20 us at one crop)
    const long long st = (long long)gridDim.x * 4;
    long long r = blockIdx.x * 4LL + rl;
    float a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (; r + 3 * st < rows; r += 4 * st) {
      acc += g[r * ld + c];
      a1 += g[(r + st) * ld + c];
      a2 += g[(r + 2 * st) * ld + c];
      a3 += g[(r + 3 * st) * ld + c];
    }
    for (; r < rows; r += st) acc += g[r * ld + c];
    acc = (acc + a1) + (a2 + a3);
  }
  red[rl][threadIdx.x & 63] = acc;
  __syncthreads();
  if (rl == 0 && c < C) {
    float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    atomicAdd(&out[c], v * (scale ? *scale : 1.f));
  }
}

hipError_t launch_colsum(const float* g, int ld, long long rows, int C, const float* scale, float* out, hipStream_t s) {
  int bx = (int)((rows + 255) / 256);
  if (bx < 1) bx = 1;
  if (bx > 256) bx = 256;
  hipLaunchKernelGGL(colsum_kernel, dim3(bx, (C + 63) / 64), dim3(256), 0, s, g, ld, rows, C, scale, out);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// GroupNorm(8)+Mish backward.  Forward: xn=(y-mu)*rstd; z=xn*gamma+beta; a=mish(z); consumer sees a*m.
//   dz = g*m*mish'(z);  ggamma[c] += dz*xn;  gbeta[c] += dz;  per (item, group): S1 = sum dz*gamma, S2 = sum dz*gamma*xn
//   gy = rstd*(dz*gamma - S1/N - xn*S2/N);  gbias_conv[c] += gy
// ---------------------------------------------------------------------------------------------------
struct GnStat { float mean, rstd; };
__device__ __forceinline__ GnStat gn_stat(const double* stats, long long b, int group, double cnt) {
  double mean = stat_read(stats, b, group, 0) / cnt;
  double var = stat_read(stats, b, group, 1) / cnt - mean * mean;
  if (var < 0) var = 0;
  GnStat r;
  r.mean = (float)mean;
  r.rstd = (float)(1.0 / sqrt(var + 1e-5));
  return r;
}

// A thread owns one channel quad for its whole life (C/4 divides 256 or is a multiple of it for every width of the U-Net; any
// other width takes the re-derive-per-element branch), reads 16 bytes per operand and keeps its per-channel and per-group sums
// in registers: one round of LDS atomics per thread at the end instead of two per element.
constexpr int kGnFlush = 16;
template <int PASS>
__global__ __launch_bounds__(256) void gn_bwd_kernel(GnBwdArgs a) {
  const int b = blockIdx.y;
  const int C = a.C, cg = C / kGroups, C4 = C >> 2;
  const long long n = (long long)a.H * a.W;
  const double cnt = (double)n * cg;
  __shared__ float s_mean[kGroups], s_rstd[kGroups], s_s1[kGroups], s_s2[kGroups];
  __shared__ float s_ch[2][1024];          // per-channel partial sums (C <= 1024)
  __shared__ double s_grp[kGroups][2];
  if (threadIdx.x < kGroups) {
    GnStat st = gn_stat(a.stats, b, threadIdx.x, cnt);
    s_mean[threadIdx.x] = st.mean;
    s_rstd[threadIdx.x] = st.rstd;
    if (PASS == 2) {
      const double* gs = a.gsum + ((long long)b * kGroups + threadIdx.x) * 2;
      s_s1[threadIdx.x] = (float)(gs[0] / cnt);
      s_s2[threadIdx.x] = (float)(gs[1] / cnt);
    }
    s_grp[threadIdx.x][0] = 0.0;
    s_grp[threadIdx.x][1] = 0.0;
  }
  for (int i = threadIdx.x; i < C; i += 256) { s_ch[0][i] = 0.f; s_ch[1][i] = 0.f; }
  __syncthreads();
  const float* yb = a.y + (long long)b * n * a.y_ld;
  const float* gb = a.g + (long long)b * n * a.g_ld;
  float* ob = PASS == 2 ? a.gy + (long long)b * n * a.gy_ld : nullptr;
  const float* mb = a.mask + (long long)(b % a.mask_bmod) * a.mask_ld;
  const long long total = n * C4;
  const long long stride = (long long)gridDim.x * 256;
  long long i = blockIdx.x * 256LL + threadIdx.x;
  const bool fixed_quad = (stride % C4) == 0;
  int c = (int)(i % C4) * 4;
  f32x4 ga, be, mean4, rstd4, s14, s24;
  f32x4 ch0 = {0.f, 0.f, 0.f, 0.f}, ch1 = {0.f, 0.f, 0.f, 0.f};    // PASS 1: (sum dz*xn, sum dz); PASS 2: (sum gy, -)
  // group sums of this thread: fp32 over at most kGnFlush elements, then into the fp64 LDS accumulators (the fp64 adds per element were
  // as expensive as the rest of the pass)
  float g1[4] = {0.f, 0.f, 0.f, 0.f}, g2[4] = {0.f, 0.f, 0.f, 0.f};
  int since_flush = 0;
  float gy_max = 0.f;              // PASS 2: max |gy| of this thread (the f16x3 weight-gradient kernel scales gy by it: WgradArgs::gy_amax)
  int cur = -1;
  auto flush = [&]() {
    if (cur < 0) return;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      atomicAdd(&s_ch[0][cur + k], ch0[k]);
      if (PASS == 1) {
        atomicAdd(&s_ch[1][cur + k], ch1[k]);
        atomicAdd(&s_grp[(cur + k) / cg][0], (double)g1[k]);
        atomicAdd(&s_grp[(cur + k) / cg][1], (double)g2[k]);
        g1[k] = g2[k] = 0.f;
      }
      ch0[k] = ch1[k] = 0.f;
    }
  };
  auto load_quad = [&](int cc) {
    flush();
    cur = cc;
    ga = *reinterpret_cast<const f32x4*>(a.gamma + cc);
    be = *reinterpret_cast<const f32x4*>(a.beta + cc);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int gk = (cc + k) / cg;
      mean4[k] = s_mean[gk];
      rstd4[k] = s_rstd[gk];
      if (PASS == 2) { s14[k] = s_s1[gk]; s24[k] = s_s2[gk]; }
    }
  };
  if (i < total) load_quad(c);
  long long p = i / C4;
  int w = (int)(p % a.W);
  const long long rpi = stride / C4;
  const int wstep = (int)(rpi % a.W);
  auto element = [&](const f32x4& yv, const f32x4& gv, float m, long long pp) {
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float xn = (yv[k] - mean4[k]) * rstd4[k];
      const float z = xn * ga[k] + be[k];
      const float dz = gv[k] * m * mish_grad(z);
      if (PASS == 1) {
        ch0[k] += dz * xn;
        ch1[k] += dz;
        g1[k] += dz * ga[k];
        g2[k] += dz * ga[k] * xn;
      } else {
        o[k] = rstd4[k] * (dz * ga[k] - s14[k] - xn * s24[k]);
        ch0[k] += o[k];
        gy_max = fmaxf(gy_max, fabsf(o[k]));
      }
    }
    if (PASS == 2) *reinterpret_cast<f32x4*>(ob + pp * a.gy_ld + c) = o;
    if (PASS == 1 && ++since_flush == kGnFlush) {      // (one thread rarely gets this far: 4-10 elements per quad at training sizes)
      const int keep = cur;
      flush();
      cur = keep;
      since_flush = 0;
    }
  };
  if (fixed_quad) {
    // The launch is capped at ~512 blocks (the atomics at a block's end, launch_gn_bwd), i.e. two per CU: with one pair of 16-byte loads
    // in flight per thread the pass ran at 1-2 TB/s at a pre-training batch.  Four elements' loads are issued before the first is used.
    constexpr int U = 4;
    while (i < total) {
      f32x4 yv[U], gv[U];
      float mm[U];
      long long pp[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        pp[u] = p;
        const bool ok = i + u * stride < total;
        const long long pl = ok ? p : 0;
        mm[u] = mb[(ok ? w : 0) * a.mask_step];
        yv[u] = *reinterpret_cast<const f32x4*>(yb + pl * a.y_ld + c);
        gv[u] = *reinterpret_cast<const f32x4*>(gb + pl * a.g_ld + c);
        p += rpi;
        w += wstep;
        if (w >= a.W) w -= a.W;
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (i + u * stride < total) element(yv[u], gv[u], mm[u], pp[u]);
      i += U * stride;
    }
  } else {
    for (; i < total; i += stride) {
      p = i / C4;
      w = (int)(p % a.W);
      c = (int)(i - p * C4) * 4;
      if (c != cur) load_quad(c);
      const float m = mb[w * a.mask_step];
      const f32x4 yv = *reinterpret_cast<const f32x4*>(yb + p * a.y_ld + c);
      const f32x4 gv = *reinterpret_cast<const f32x4*>(gb + p * a.g_ld + c);
      element(yv, gv, m, p);
    }
  }
  flush();
  __syncthreads();
  if (PASS == 1) {
    for (int k = threadIdx.x; k < C; k += 256) {
      atomicAdd(&a.ggamma[k], s_ch[0][k]);
      atomicAdd(&a.gbeta[k], s_ch[1][k]);
    }
    if (threadIdx.x < kGroups * 2)
      atomicAdd(&a.gsum[((long long)b * kGroups + (threadIdx.x >> 1)) * 2 + (threadIdx.x & 1)], s_grp[threadIdx.x >> 1][threadIdx.x & 1]);
  } else {
    if (a.gbias)
      for (int k = threadIdx.x; k < C; k += 256) atomicAdd(&a.gbias[k], s_ch[0][k]);
    if (a.gy_amax) {               // as wgrad_amax_kernel: non-negative floats order like their bit patterns; one atomic per block
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) gy_max = fmaxf(gy_max, __shfl_xor(gy_max, off));
      __syncthreads();             // (s_mean is dead: every thread is past its loop)
      if ((threadIdx.x & 63) == 0) s_mean[threadIdx.x >> 6] = gy_max;
      __syncthreads();
      if (threadIdx.x == 0) {
        gy_max = fmaxf(fmaxf(s_mean[0], s_mean[1]), fmaxf(s_mean[2], s_mean[3]));
        if (gy_max > 0.f) atomicMax(reinterpret_cast<unsigned*>(a.gy_amax), __float_as_uint(gy_max));
      }
    }
  }
}

hipError_t launch_gn_bwd(const GnBwdArgs& a, hipStream_t s) {
  if (a.C > 1024 || a.C % kGroups != 0 || a.C % 4 != 0 || a.y_ld % 4 != 0 || a.g_ld % 4 != 0 || a.gy_ld % 4 != 0) return hipErrorInvalidValue;
  long long total = (long long)a.H * a.W * (a.C / 4);
  int blocks = (int)((total + 256 * 4 - 1) / (256 * 4));
  if (blocks < 1) blocks = 1;
  if (blocks > 1024) blocks = 1024;
  // every block ends with one atomic per channel on the SAME C addresses (gamma / beta / conv-bias gradients), which the L2 serialises:
  // at 32 crops, 14,080 blocks of a level-0 launch queued 28,000 atomics per address.  Cap the blocks of a launch (all items) instead.
  // (measured: pre-training step at 32 crops 62.1 ms uncapped, 55.1-55.8 for caps of 256 ... 1,024 blocks; one crop: 10.28 -> 10.15 ms
  // per fine-tune iteration at 256)
  static int target = -1;
  if (target < 0) { const char* e = getenv("US_GN_BWD_BLOCKS"); target = e ? atoi(e) : 512; }
  if (target > 0) {
    if (blocks > target / 2) blocks = target / 2;
    if ((long long)blocks * a.B > target) blocks = target / a.B;
    if (blocks < 8) blocks = 8;
  }
  hipLaunchKernelGGL(gn_bwd_kernel<1>, dim3(blocks, a.B), dim3(256), 0, s, a);
  hipLaunchKernelGGL(gn_bwd_kernel<2>, dim3(blocks, a.B), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// attention backward pieces (see decoder.hip attention_backward for the algebra)
// ---------------------------------------------------------------------------------------------------
// gctx[b][h][d][e] += sum_n q[n][h*32+d] * gO[n][h*32+e]          (q: cols [0,128) of qkv, ld 384; gO: ld 128)
__global__ __launch_bounds__(256) void attn_bwd_gctx_kernel(const float* __restrict__ qkv, const float* __restrict__ gO, int n,
                                                            float* __restrict__ gctx) {
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int lane = threadIdx.x & 63, h = threadIdx.x >> 6;
  const int c = lane & 31, hh = lane >> 5;
  const float* qp = qkv + (long long)b * n * (3 * kHidden) + h * kDimHead + c;
  const float* gp = gO + (long long)b * n * kHidden + h * kDimHead + c;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int r0 = chunk * 128;
  // (32 row pairs' loads in flight: with 8 the 64 iterations of a block were 8 round trips to memory, 23 us per launch at one crop)
#pragma unroll 32
  for (int j = 0; j < 64; ++j) {
    int row = r0 + 2 * j + hh;
    bool v = row < n;
    int rc = v ? row : n - 1;
    float qa = qp[(long long)rc * (3 * kHidden)], gb = gp[(long long)rc * kHidden];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v ? qa : 0.f, v ? gb : 0.f, acc, 0, 0, 0);
  }
  float* out = gctx + ((long long)b * kHeads + h) * (kDimHead * kDimHead);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int d = (r & 3) + 8 * (r >> 2) + 4 * hh;
    atomicAdd(&out[d * kDimHead + c], acc[r]);
  }
}

hipError_t launch_attn_bwd_gctx(const float* qkv, const float* gO, int B, int n, float* gctx, hipStream_t s) {
  hipLaunchKernelGGL(attn_bwd_gctx_kernel, dim3((n + 127) / 128, B), dim3(256), 0, s, qkv, gO, n, gctx);
  return hipGetLastError();
}

// Per pixel and head: P = exp(k - M)/S;  t[d] = sum_e v[e]*gctx[d][e];  gk[d] = P[d]*(t[d] - cc[d]);  gv[e] = sum_d P[d]*gctx[d][e]
// with cc[d] = sum_e ctx[d][e]*gctx[d][e].  Writes gk, gv into columns [128,384) of gqkv (ld 384).
__global__ __launch_bounds__(256) void attn_bwd_kv_kernel(const float* __restrict__ qkv, const float* __restrict__ ctx,
                                                          const float* __restrict__ gctx, const float* __restrict__ colM,
                                                          const float* __restrict__ colS, int n, float* __restrict__ gqkv) {
  __shared__ float sg[kHeads][kDimHead][kDimHead + 1];    // gctx[h][d][e]
  __shared__ float sgt[kHeads][kDimHead][kDimHead + 1];   // gctx[h][e][d]^T
  __shared__ float scc[kHidden];
  const int b = blockIdx.y;
  const float* gc = gctx + (long long)b * kHeads * kDimHead * kDimHead;
  const float* cx = ctx + (long long)b * kHeads * kDimHead * kDimHead;
  for (int i = threadIdx.x; i < kHeads * kDimHead * kDimHead; i += 256) {
    int h = i >> 10, d = (i >> 5) & 31, e = i & 31;
    float v = gc[i];
    sg[h][d][e] = v;
    sgt[h][e][d] = v;
  }
  if (threadIdx.x < kHidden) {
    const int h = threadIdx.x >> 5, d = threadIdx.x & 31;
    float acc = 0.f;
    for (int e = 0; e < kDimHead; ++e) acc += cx[(h * kDimHead + d) * kDimHead + e] * gc[(h * kDimHead + d) * kDimHead + e];
    scc[threadIdx.x] = acc;
  }
  __syncthreads();
  const int grp = threadIdx.x >> 5, l = threadIdx.x & 31;     // 8 groups of 32 lanes: 2 pixels x 4 heads per pass
  const int h = grp & 3;
  const float M = colM[(long long)b * kHidden + h * kDimHead + l], S = colS[(long long)b * kHidden + h * kDimHead + l];
  const float cc = scc[h * kDimHead + l];
  for (long long p = blockIdx.x * 2LL + (grp >> 2); p < n; p += (long long)gridDim.x * 2) {
    const float* row = qkv + ((long long)b * n + p) * (3 * kHidden);
    const float kk = row[kHidden + h * kDimHead + l];
    const float vv = row[2 * kHidden + h * kDimHead + l];
    const float P = expf(kk - M) / S;
    float t = 0.f, gv = 0.f;
#pragma unroll 8
    for (int j = 0; j < kDimHead; ++j) {
      float vj = __shfl(vv, j, 32), Pj = __shfl(P, j, 32);
      t += vj * sg[h][l][j];          // d = l, e = j
      gv += Pj * sgt[h][l][j];        // e = l, d = j : gctx[j][l]
    }
    float* orow = gqkv + ((long long)b * n + p) * (3 * kHidden);
    orow[kHidden + h * kDimHead + l] = P * (t - cc);
    orow[2 * kHidden + h * kDimHead + l] = gv;
  }
}

// The same on the matrix cores.  Per head the two products are 32 x 32 GEMMs over a block of 32 pixels,
//   T^T[d][p] = sum_e gctx[d][e] v[p][e],   GV^T[e][p] = sum_d gctx[d][e] P[p][d],
// with the pixel on the MFMA's column index: a lane owns one pixel (it loads the pixel's 32 k and 32 v values of the head as eight 16-byte
// vectors each and feeds v / P as the B operand straight from those registers, element 2j + hh at step j), gctx is the A operand (16 + 16
// registers per lane, loaded once per workgroup: wave = head), and the results come back with the pixel on the lane and 16 of the head's
// channels in the registers -- four 16-byte stores per output and pixel.  v_mfma_f32_32x32x2_f32: exact fp32 products and sums, as the
// shuffle form's fmas (another order).  The shuffle form spent 32 x (2 shuffles + 2 LDS reads + 2 fmas) per pixel and head: 1.1 ms for the
// level-0 attention of a pre-training batch.
__global__ __launch_bounds__(256) void attn_bwd_kv_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ ctx,
                                                               const float* __restrict__ gctx, const float* __restrict__ colM,
                                                               const float* __restrict__ colS, int n, float* __restrict__ gqkv) {
  __shared__ __attribute__((aligned(16))) float s_m[kHidden], s_is[kHidden], s_cc[kHidden];
  const int b = blockIdx.y;
  const float* gc = gctx + (long long)b * kHeads * kDimHead * kDimHead;
  const float* cx = ctx + (long long)b * kHeads * kDimHead * kDimHead;
  if (threadIdx.x < kHidden) {
    const int hd = threadIdx.x;              // h * 32 + d
    float acc = 0.f;
    for (int e = 0; e < kDimHead; ++e) acc += cx[hd * kDimHead + e] * gc[hd * kDimHead + e];
    s_cc[hd] = acc;
    s_m[hd] = colM[(long long)b * kHidden + hd];
    s_is[hd] = 1.f / colS[(long long)b * kHidden + hd];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, h = threadIdx.x >> 6;
  const int l32 = lane & 31, hh = lane >> 5;
  const float* gh = gc + h * kDimHead * kDimHead;
  float a1[16], a2[16];                      // A operands: gctx[d = l32][e = 2j + hh];  gctx[d = 2j + hh][e = l32]
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    a1[j] = gh[l32 * kDimHead + 2 * j + hh];
    a2[j] = gh[(2 * j + hh) * kDimHead + l32];
  }
  const int nblk = (n + 31) / 32;
  for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int p = blk * 32 + l32;
    const bool pv = p < n;
    const float* row = qkv + ((long long)b * n + (pv ? p : 0)) * (3 * kHidden) + h * kDimHead;
    f32x4 kq[8], vq[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      kq[i] = *reinterpret_cast<const f32x4*>(row + kHidden + 4 * i);
      vq[i] = *reinterpret_cast<const f32x4*>(row + 2 * kHidden + 4 * i);
    }
    // P[d] = exp(k[d] - M[d]) / S[d] for the pixel's 32 channels of this head (the statistics: uniform LDS reads)
    float P[32];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f32x4 m4 = *reinterpret_cast<const f32x4*>(s_m + h * kDimHead + 4 * i);
      const f32x4 is4 = *reinterpret_cast<const f32x4*>(s_is + h * kDimHead + 4 * i);
#pragma unroll
      for (int k = 0; k < 4; ++k) P[4 * i + k] = pv ? expf(kq[i][k] - m4[k]) * is4[k] : 0.f;
    }
    f32x16 T, GV;
#pragma unroll
    for (int r = 0; r < 16; ++r) { T[r] = 0.f; GV[r] = 0.f; }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float v0 = vq[j >> 1][(2 * j) & 3], v1 = vq[j >> 1][(2 * j + 1) & 3];
      const float bv = pv ? (hh ? v1 : v0) : 0.f;
      const float bp = hh ? P[2 * j + 1] : P[2 * j];
      T = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], bv, T, 0, 0, 0);
      GV = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[j], bp, GV, 0, 0, 0);
    }
    if (pv) {
      float* orow = gqkv + ((long long)b * n + p) * (3 * kHidden) + h * kDimHead;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // registers 4g .. 4g + 3 hold rows 8g + 4hh + (0..3): channel d (of T^T) resp. e (of GV^T)
        const int d0 = 8 * g + 4 * hh;
        const f32x4 cc4 = *reinterpret_cast<const f32x4*>(s_cc + h * kDimHead + d0);
        f32x4 gk4, gv4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float pd = hh ? P[8 * g + 4 + k] : P[8 * g + k];
          gk4[k] = pd * (T[4 * g + k] - cc4[k]);
          gv4[k] = GV[4 * g + k];
        }
        *reinterpret_cast<f32x4*>(orow + kHidden + d0) = gk4;
        *reinterpret_cast<f32x4*>(orow + 2 * kHidden + d0) = gv4;
      }
    }
  }
}

hipError_t launch_attn_bwd_kv(const float* qkv, const float* ctx, const float* gctx, const float* colM, const float* colS, int B,
                              int n, float* gqkv, hipStream_t s) {
  static const bool mfma = [] { const char* e = getenv("US_ATTN_BWD_KV_MFMA"); return !e || atoi(e) != 0; }();
  if (mfma) {
    int bm = (n + 31) / 32;                  // 32-pixel blocks of an item; a workgroup's prologue (gctx, cc) wants several of them at a batch
    const int cap = 1024 / B < 8 ? 8 : 1024 / B;
    if (bm > cap) bm = cap;
    hipLaunchKernelGGL(attn_bwd_kv_mfma_kernel, dim3(bm, B), dim3(256), 0, s, qkv, ctx, gctx, colM, colS, n, gqkv);
    return hipGetLastError();
  }
  int bx = (n + 1) / 2;
  if (bx > 1024) bx = 1024;
  hipLaunchKernelGGL(attn_bwd_kv_kernel, dim3(bx, B), dim3(256), 0, s, qkv, ctx, gctx, colM, colS, n, gqkv);
  return hipGetLastError();
}

// From M1[b] = G^T q  ([C][128], fp32, per item) and ctx:
//   gWout[c][h*32+e] += g * sum_b sum_d M1[b][c][h*32+d] * ctx[b][h][d][e]
//   gg += sum_b <weff_nat[b], M1[b]>   with weff_nat[b][c][hd] = sum_e Wout[c][he]*ctx[b][h][d][e]  (recomputed here)
__global__ __launch_bounds__(256) void attn_bwd_wout_kernel(const float* __restrict__ M1, const float* __restrict__ ctx,
                                                            const float* __restrict__ wout, const float* __restrict__ g, int B, int C,
                                                            float* __restrict__ gwout, float* __restrict__ gg) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  float dot = 0.f;
  if (i < C * kHidden) {
    const int c = i / kHidden, he = i % kHidden, h = he / kDimHead, e = he % kDimHead;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) {
      const float* m1 = M1 + ((long long)b * C + c) * kHidden + h * kDimHead;
      const float* cx = ctx + ((long long)b * kHeads + h) * kDimHead * kDimHead;
      float weff = 0.f;                       // weff_nat[b][c][h*32 + e'] with e' := e playing the role of d
      if (gg) {
        for (int d = 0; d < kDimHead; ++d) {
          acc += m1[d] * cx[d * kDimHead + e];
          weff += wout[(long long)c * kHidden + h * kDimHead + d] * cx[e * kDimHead + d];
        }
      } else {
        for (int d = 0; d < kDimHead; ++d) acc += m1[d] * cx[d * kDimHead + e];
      }
      dot += weff * m1[e];
    }
    atomicAdd(&gwout[i], acc * g[0]);
  }
  if (!gg) return;         // the gain's gradient comes from launch_dot_partial (fixed-order fp64 sum of grad_out * fn(x)), not from M1
  dot = wsum(dot);
  __shared__ float s_dot[4];                  // one atomic per block on the single gain-gradient word
  if ((threadIdx.x & 63) == 0) s_dot[threadIdx.x >> 6] = dot;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(gg, (s_dot[0] + s_dot[1]) + (s_dot[2] + s_dot[3]));
}

hipError_t launch_attn_bwd_wout(const float* M1, const float* ctx, const float* wout, const float* g, int B, int C, float* gwout,
                                float* gg, hipStream_t s) {
  hipLaunchKernelGGL(attn_bwd_wout_kernel, dim3((C * kHidden + 255) / 256), dim3(256), 0, s, M1, ctx, wout, g, B, C, gwout, gg);
  return hipGetLastError();
}

// gg += sum_c bout[c]*colsumG[c];  gbout[c] += g*colsumG[c]
__global__ void attn_bwd_bias_kernel(const float* __restrict__ colsumG, const float* __restrict__ bout, const float* __restrict__ g, int C,
                                     float* __restrict__ gbout, float* __restrict__ gg) {
  float dot = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    dot += bout[c] * colsumG[c];
    atomicAdd(&gbout[c], g[0] * colsumG[c]);
  }
  if (!gg) return;
  dot = wsum(dot);
  if ((threadIdx.x & 63) == 0) atomicAdd(gg, dot);
}

// ---------------------------------------------------------------------------------------------------
// Deterministic scalar reductions (the Rezero gains' gradients, the final projection's bias gradient).  Rezero(fn)(x) = fn(x) * g
// (unitspeech/unitspeech.py:36-43), so autograd's g.grad = sum(grad_out * fn(x)): a sum of ~1e6 products that cancel to a few 1e-5.
// Formed from M1 = G^T q and W_eff (two more roundings in front of the cancellation) and added up by float atomics, the eight gains
// were 5e-4 ... 1e-3 away from the oracle and moved from run to run; the reference's own fp32 result is within 4e-6 of its fp64 one
// (tests/golden/grads_full_8x176_fp64.npz).  Here: every block sums its fixed share of the products in fp64 (a float product is exact in
// a double), writes ONE partial, and reduce_finalize_kernel adds the partials in index order.  No atomics, no dependence on arrival order.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wsum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ void block_partial_store(double v, double* dst) {      // 256 threads; every thread calls
  __shared__ double s_p[4];
  v = wsum_d(v);
  if ((threadIdx.x & 63) == 0) s_p[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) *dst = (s_p[0] + s_p[1]) + (s_p[2] + s_p[3]);
}

__global__ __launch_bounds__(256) void dot_partial_kernel(const float* __restrict__ a, int a_ld, const float* __restrict__ b, int b_ld,
                                                          long long rows, int C, double* __restrict__ partials) {
  const int C4 = C >> 2;
  const long long total = rows * C4;
  double acc = 0.0;
  if (total < (1LL << 31)) {
    // (32-bit indices, and two elements' loads in flight: the same elements, in the same order, into the same accumulator)
    const unsigned stride = gridDim.x * 256u, tot = (unsigned)total, c4 = (unsigned)C4;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < tot; i += 2u * stride) {
      const unsigned i1 = i + stride < tot ? i + stride : i;
      const unsigned r0 = i / c4, r1 = i1 / c4;
      const unsigned c0 = (i - r0 * c4) * 4u, c1 = (i1 - r1 * c4) * 4u;
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(a + (long long)r0 * a_ld + c0);
      const f32x4 y0 = *reinterpret_cast<const f32x4*>(b + (long long)r0 * b_ld + c0);
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(a + (long long)r1 * a_ld + c1);
      const f32x4 y1 = *reinterpret_cast<const f32x4*>(b + (long long)r1 * b_ld + c1);
      acc += ((double)x0[0] * (double)y0[0] + (double)x0[1] * (double)y0[1]) + ((double)x0[2] * (double)y0[2] + (double)x0[3] * (double)y0[3]);
      if (i + stride < tot)
        acc += ((double)x1[0] * (double)y1[0] + (double)x1[1] * (double)y1[1]) + ((double)x1[2] * (double)y1[2] + (double)x1[3] * (double)y1[3]);
    }
    block_partial_store(acc, partials + blockIdx.x);
    return;
  }
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / C4;
    const int c = (int)(i - r * C4) * 4;
    const f32x4 x = *reinterpret_cast<const f32x4*>(a + r * a_ld + c);
    const f32x4 y = *reinterpret_cast<const f32x4*>(b + r * b_ld + c);
    acc += ((double)x[0] * (double)y[0] + (double)x[1] * (double)y[1]) + ((double)x[2] * (double)y[2] + (double)x[3] * (double)y[3]);
  }
  block_partial_store(acc, partials + blockIdx.x);
}

int dot_partial_blocks(long long rows, int C) {
  long long b = (rows * (C >> 2) + 256 * 8 - 1) / (256 * 8);
  return (int)(b < 1 ? 1 : (b > kRedBlocks ? kRedBlocks : b));
}

hipError_t launch_dot_partial(const float* a, int a_ld, const float* b, int b_ld, long long rows, int C, double* partials, hipStream_t s) {
  if (C % 4 != 0 || a_ld % 4 != 0 || b_ld % 4 != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(dot_partial_kernel, dim3(dot_partial_blocks(rows, C)), dim3(256), 0, s, a, a_ld, b, b_ld, rows, C, partials);
  return hipGetLastError();
}

// dst[j][0] += scale[j] * sum_i p[j][i], i = 0 .. n[j] - 1 in index order (one wave per job: lane l takes i = l, l + 64, ..., then a fixed tree)
__global__ __launch_bounds__(64) void reduce_finalize_kernel(RedJobs jobs) {
  const int j = blockIdx.x;
  const double* p = jobs.p[j];
  double acc = 0.0;
  for (int i = threadIdx.x; i < jobs.n[j]; i += 64) acc += p[i];
  acc = wsum_d(acc);
  if (threadIdx.x == 0) {
    const double sc = jobs.scale[j] ? (double)jobs.scale[j][0] : 1.0;
    jobs.dst[j][0] += (float)(acc * sc);
  }
}

hipError_t launch_reduce_finalize(const RedJobs& jobs, hipStream_t s) {
  if (jobs.count <= 0) return hipSuccess;
  if (jobs.count > RedJobs::kMax) return hipErrorInvalidValue;
  hipLaunchKernelGGL(reduce_finalize_kernel, dim3(jobs.count), dim3(64), 0, s, jobs);
  return hipGetLastError();
}

hipError_t launch_attn_bwd_bias(const float* colsumG, const float* bout, const float* g, int C, float* gbout, float* gg, hipStream_t s) {
  hipLaunchKernelGGL(attn_bwd_bias_kernel, dim3(1), dim3(256), 0, s, colsumG, bout, g, C, gbout, gg);
  return hipGetLastError();
}

// weff for the dgrad GEMM: K = C (output channels of to_out), N = 128:  dst[b][C/bk][128][bk], value weff_nat[b][c][hd]
__global__ __launch_bounds__(256) void attn_weff_dgrad_kernel(const float* __restrict__ ctx, const float* __restrict__ wout,
                                                              float* __restrict__ dst, int C, int bk) {
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
  dst[(long long)b * C * kHidden + ((long long)(co / bk) * kHidden + hd) * bk + co % bk] = acc;
}

hipError_t launch_attn_weff_dgrad(const float* ctx, const float* wout, float* dst, int B, int C, int bk, hipStream_t s) {
  hipLaunchKernelGGL(attn_weff_dgrad_kernel, dim3((C * kHidden + 255) / 256, B), dim3(256), 0, s, ctx, wout, dst, C, bk);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// final projection backward: out[p] = (b0 + sum_c w[c]*h[p][c]*m)*m
//   gh[p][c] = go[p]*m*w[c];  gw[c] += sum_p go*m*h[p][c];  gb0 += sum_p go*m
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void final_bwd_kernel(const float* __restrict__ go, const float* __restrict__ h, int ld,
                                                        const float* __restrict__ w, const float* __restrict__ mask, int mask_ld,
                                                        int mask_bmod, int W, long long n, int C, float* __restrict__ gh,
                                                        float* __restrict__ gw, double* __restrict__ gb0_partials) {
  const int b = blockIdx.y;
  __shared__ float s_w[1024];
  for (int i = threadIdx.x; i < C; i += 256) s_w[i] = 0.f;
  __syncthreads();
  const float* mb = mask + (long long)(b % mask_bmod) * mask_ld;
  const long long total = n * C;
  double gbacc = 0.0;
  // a thread keeps its channel when the grid stride is a multiple of C (every full-size launch): its products are summed in a register
  // and reach the LDS accumulator once
  const bool fixed_c = ((long long)gridDim.x * 256) % C == 0;
  const int C4 = C >> 2;
  if ((C & 3) == 0 && (ld & 3) == 0 && ((long long)gridDim.x * 256) % C4 == 0 && n * C4 < (1LL << 31)) {
    // Every full-size launch: a thread owns one channel quad for its whole life and walks the pixels with a fixed stride -- 16-byte
    // loads and stores, no division in the loop (the scalar form below spends two 64-bit divisions on every float: 25 us at one crop
    // for 7 MB, 0.5 ms at a pre-training batch, as the first kernel of every backward pass).
    const unsigned stride = gridDim.x * 256u, total4 = (unsigned)(n * C4);
    unsigned i = blockIdx.x * 256u + threadIdx.x;
    const int c = (int)(i % (unsigned)C4) * 4;
    const unsigned rpi = stride / (unsigned)C4;                 // pixels advanced per iteration
    unsigned p = i / (unsigned)C4;
    int wcol = (int)(p % (unsigned)W);
    const int wstep = (int)(rpi % (unsigned)W);
    const f32x4 w4 = *reinterpret_cast<const f32x4*>(w + c);
    f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
    const float* gob = go + (long long)b * n;
    const float* hb = h + (long long)b * n * ld + c;
    float* ghb = gh + (long long)b * n * ld + c;
    for (; i < total4; i += stride) {
      const float g = gob[p] * mb[wcol];
      const f32x4 hv = *reinterpret_cast<const f32x4*>(hb + (long long)p * ld);
      acc4 += hv * g;
      *reinterpret_cast<f32x4*>(ghb + (long long)p * ld) = w4 * g;
      if (c == 0) gbacc += (double)g;
      p += rpi;
      wcol += wstep;
      if (wcol >= W) wcol -= W;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) atomicAdd(&s_w[c + k], acc4[k]);
  } else {
  float wacc = 0.f;
  int c_mine = -1;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long long p = i / C;
    const float g = go[(long long)b * n + p] * mb[(int)(p % W)];
    const long long idx = ((long long)b * n + p) * ld + c;
    if (fixed_c) { wacc += g * h[idx]; c_mine = c; }
    else atomicAdd(&s_w[c], g * h[idx]);
    gh[idx] = g * w[c];
    if (c == 0) gbacc += (double)g;
  }
  if (fixed_c && c_mine >= 0) atomicAdd(&s_w[c_mine], wacc);
  }
  // the bias gradient sum_p go * m: one fp64 partial per block, added in index order by reduce_finalize_kernel (no atomics: reproducible)
  block_partial_store(gbacc, gb0_partials + (long long)b * gridDim.x + blockIdx.x);
  for (int i = threadIdx.x; i < C; i += 256) atomicAdd(&gw[i], s_w[i]);
}

int final_bwd_blocks(int B, int H, int W, int C) {
  long long n = (long long)H * W;
  int blocks = (int)((n * C + 256 * 16 - 1) / (256 * 16));      // (four quads per thread in the vector form)
  if (blocks < 1) blocks = 1;
  if (blocks > 512) blocks = 512;
  // every block ends with C atomics on the same C addresses of gw: cap the launch at 512 blocks over all items, as the GroupNorm backward
  // (launch_gn_bwd)
  if ((long long)blocks * B > 512) blocks = 512 / B < 4 ? 4 : 512 / B;
  return blocks;
}

hipError_t launch_final_bwd(const float* go, const float* h, int ld, const float* w, const float* mask, int mask_ld, int mask_bmod,
                            float* gh, float* gw, double* gb0_partials, int B, int H, int W, int C, hipStream_t s) {
  if (C > 1024) return hipErrorInvalidValue;
  long long n = (long long)H * W;
  hipLaunchKernelGGL(final_bwd_kernel, dim3(final_bwd_blocks(B, H, W, C), B), dim3(256), 0, s, go, h, ld, w, mask, mask_ld, mask_bmod, W, n, C,
                     gh, gw, gb0_partials);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// first layer (2 input channels) weight gradients: gw3[co][ci][ky][kx] += sum_p gy[p][co]*in2[p+d][ci];
// gw1[co][ci] += sum_p gr[p][co]*in2[p][ci]     (reference layouts, written with atomics)
// ---------------------------------------------------------------------------------------------------
constexpr int kFcwMaxT = 2048;      // frames per item the LDS patch of first_conv_wgrad_kernel holds (3 rows x (T + 2) x 2 channels)
__global__ __launch_bounds__(256) void first_conv_wgrad_kernel(const float* __restrict__ in2, const float* __restrict__ gy,
                                                               const float* __restrict__ gr, int F, int T, int C, int Bp, int ipb,
                                                               float* __restrict__ gw3, float* __restrict__ gw1) {
  // one block = one mel row of `ipb` items.  The three input rows it touches sit in LDS, zero-padded at both ends and for rows outside
  // the image, so the frame loop has no branches and the compiler can keep several frames' gradient loads in flight (the loop used to
  // wait for one pair of loads per frame: 1.2 ms at 32 crops, 0.11 ms at one, for 0.2 GB of traffic).  Thread = (output channel,
  // frame parity); the two parities meet in LDS before the atomics.
  extern __shared__ float fsm[];                    // patch [3][T + 2][2], then the parity-1 partial sums [C][20]
  float* patch = fsm;
  float* part = fsm + 3 * (T + 2) * 2;
  const int f = blockIdx.x;
  const int half = threadIdx.x >> 7;                // frame parity (C <= 128 per pass)
  for (int c0 = 0; c0 < C; c0 += 128) {
    const int co = c0 + (threadIdx.x & 127);
    float a3[18], a1[2];
#pragma unroll
    for (int i = 0; i < 18; ++i) a3[i] = 0.f;
    a1[0] = a1[1] = 0.f;
    // several items per block (ipb): every block ends in 20 atomics per channel on the SAME 2,560 addresses, which at 32 crops was most
    // of the kernel's time
    for (int b = blockIdx.y * ipb; b < (blockIdx.y + 1) * ipb && b < Bp; ++b) {
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * (T + 2) * 2; i += 256) {
      const int c = i & 1, xx = (i >> 1) % (T + 2), yy = (i >> 1) / (T + 2);
      const int ff = f + yy - 1, tt = xx - 1;
      patch[i] = (ff >= 0 && ff < F && tt >= 0 && tt < T) ? in2[(((long long)b * F + ff) * T + tt) * 2 + c] : 0.f;
    }
    __syncthreads();
    if (co < C) {
      const float* gyp = gy + (((long long)b * F + f) * T) * C + co;
      const float* grp = gr + (((long long)b * F + f) * T) * C + co;
#pragma unroll 4
      for (int t = half; t < T; t += 2) {
        const float g = gyp[(long long)t * C], r = grp[(long long)t * C];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const float* ip = patch + ((ky * (T + 2)) + t + kx) * 2;      // frame t + kx - 1 of row f + ky - 1
            a3[ky * 3 + kx] = fmaf(g, ip[0], a3[ky * 3 + kx]);
            a3[9 + ky * 3 + kx] = fmaf(g, ip[1], a3[9 + ky * 3 + kx]);
          }
        const float* ip = patch + ((T + 2) + t + 1) * 2;
        a1[0] = fmaf(r, ip[0], a1[0]);
        a1[1] = fmaf(r, ip[1], a1[1]);
      }
    }
    }   // items of this block
    if (half == 1 && co < C) {
#pragma unroll
      for (int i = 0; i < 18; ++i) part[(co - c0) * 20 + i] = a3[i];
      part[(co - c0) * 20 + 18] = a1[0];
      part[(co - c0) * 20 + 19] = a1[1];
    }
    __syncthreads();
    if (half == 0 && co < C) {
#pragma unroll
      for (int i = 0; i < 18; ++i) atomicAdd(&gw3[co * 18 + i], a3[i] + part[(co - c0) * 20 + i]);
      atomicAdd(&gw1[co * 2], a1[0] + part[(co - c0) * 20 + 18]);
      atomicAdd(&gw1[co * 2 + 1], a1[1] + part[(co - c0) * 20 + 19]);
    }
    __syncthreads();
  }
}

hipError_t launch_first_conv_wgrad(const float* in2, const float* gy, const float* gr, int Bp, int F, int T, int C, float* gw3,
                                   float* gw1, hipStream_t s) {
  if (T > kFcwMaxT) return hipErrorInvalidValue;
  const size_t lds = ((size_t)3 * (T + 2) * 2 + 128 * 20) * sizeof(float);
  int ipb = (int)(((long long)Bp * F + 639) / 640);       // items per block: about 640 blocks (2.5 per CU)
  if (ipb < 1) ipb = 1;
  if (ipb > 8) ipb = 8;
  hipLaunchKernelGGL(first_conv_wgrad_kernel, dim3(F, (Bp + ipb - 1) / ipb), dim3(256), lds, s, in2, gy, gr, F, T, C, Bp, ipb, gw3, gw1);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// small dense layers: y[r][o] = bias[o] + sum_i W[o][i]*f(x[r][i])
//   gW[o][i] += sum_r gy[r][o]*f(x[r][i]);  gb[o] += sum_r gy[r][o];  gx[r][i] (+)= f'(x[r][i]) * sum_o gy[r][o]*W[o][i]
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float mish_fwd(float x) {
  if (x > 20.f) return x;
  float w = expf(x);
  float u = w * (w + 2.f);
  return x * (u / (u + 2.f));
}

__global__ void linear_bwd_w_kernel(const float* __restrict__ gy, int gy_ld, const float* __restrict__ x, int x_ld, int rows, int in_dim,
                                    int out_dim, int mish_in, float* __restrict__ gW, float* __restrict__ gb) {
  const long long i = blockIdx.x * 256LL + threadIdx.x;
  if (i >= (long long)out_dim * in_dim) return;
  const int o = (int)(i / in_dim), k = (int)(i % in_dim);
  float acc = 0.f, accb = 0.f;
  for (int r = 0; r < rows; ++r) {
    float xv = x[(long long)r * x_ld + k];
    if (mish_in) xv = mish_fwd(xv);
    const float g = gy[(long long)r * gy_ld + o];
    acc += g * xv;
    accb += g;
  }
  atomicAdd(&gW[i], acc);
  if (k == 0 && gb) atomicAdd(&gb[o], accb);
}

// gx[r][k] += sum_{o in slice} gy[r][o] * W[o][k]   (raw: the mish'(x) factor of a mish-input layer is applied once by
// mul_mish_grad_kernel after every layer sharing that input has been accumulated)
__global__ __launch_bounds__(256) void linear_bwd_x_kernel(const float* __restrict__ gy, int gy_ld, const float* __restrict__ W, int in_dim,
                                                           int out_dim, int per_slice, float* __restrict__ gx, int gx_ld) {
  const int r = blockIdx.y;
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= in_dim) return;
  const int o0 = blockIdx.z * per_slice;
  int o1 = o0 + per_slice;
  if (o1 > out_dim) o1 = out_dim;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int o = o0;
  for (; o + 4 <= o1; o += 4) {
    a0 = fmaf(gy[(long long)r * gy_ld + o], W[(long long)o * in_dim + k], a0);
    a1 = fmaf(gy[(long long)r * gy_ld + o + 1], W[(long long)(o + 1) * in_dim + k], a1);
    a2 = fmaf(gy[(long long)r * gy_ld + o + 2], W[(long long)(o + 2) * in_dim + k], a2);
    a3 = fmaf(gy[(long long)r * gy_ld + o + 3], W[(long long)(o + 3) * in_dim + k], a3);
  }
  for (; o < o1; ++o) a0 = fmaf(gy[(long long)r * gy_ld + o], W[(long long)o * in_dim + k], a0);
  atomicAdd(gx + (long long)r * gx_ld + k, (a0 + a1) + (a2 + a3));
}

__global__ void mul_mish_grad_kernel(float* __restrict__ g, int g_ld, const float* __restrict__ x, int x_ld, int rows, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * n) return;
  const int r = i / n, k = i % n;
  g[(long long)r * g_ld + k] *= mish_grad(x[(long long)r * x_ld + k]);
}

// LinJob sets (kernels.h): the weight / bias gradients of all jobs in one launch, the input gradient in another
__global__ void linear_bwd_w_multi_kernel(const LinJob* __restrict__ jobs, int njobs, int total_out, const float* __restrict__ x, int x_ld, int rows,
                                          int in_dim, int mish_in) {
  const long long i = blockIdx.x * 256LL + threadIdx.x;
  if (i >= (long long)total_out * in_dim) return;
  const int og = (int)(i / in_dim), k = (int)(i % in_dim);
  int j = 0;
  while (j + 1 < njobs && og >= jobs[j + 1].o0) ++j;
  const LinJob jb = jobs[j];
  const int o = og - jb.o0;
  float acc = 0.f, accb = 0.f;
  for (int r = 0; r < rows; ++r) {
    float xv = x[(long long)r * x_ld + k];
    if (mish_in) xv = mish_fwd(xv);
    const float g = jb.gy[(long long)r * jb.cout + o];
    acc += g * xv;
    accb += g;
  }
  atomicAdd(&jb.gW[(long long)o * in_dim + k], acc);
  if (k == 0 && jb.gb) atomicAdd(&jb.gb[o], accb);
}

__global__ __launch_bounds__(256) void linear_bwd_x_multi_kernel(const LinJob* __restrict__ jobs, int njobs, int total_out, int in_dim,
                                                                 int per_slice, float* __restrict__ gx, int gx_ld) {
  const int r = blockIdx.y;
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= in_dim) return;
  const int o0 = blockIdx.z * per_slice;
  int o1 = o0 + per_slice;
  if (o1 > total_out) o1 = total_out;
  int j = 0;
  while (j + 1 < njobs && o0 >= jobs[j + 1].o0) ++j;
  float a0 = 0.f, a1 = 0.f;
  for (int og = o0; og < o1;) {
    while (j + 1 < njobs && og >= jobs[j + 1].o0) ++j;
    const LinJob jb = jobs[j];
    int oe = jb.o0 + jb.cout;
    if (oe > o1) oe = o1;
    const float* gy = jb.gy + (long long)r * jb.cout - jb.o0;         // indexed by the global column
    const float* W = jb.W - (long long)jb.o0 * in_dim;
    int o = og;
    for (; o + 2 <= oe; o += 2) {
      a0 = fmaf(gy[o], W[(long long)o * in_dim + k], a0);
      a1 = fmaf(gy[o + 1], W[(long long)(o + 1) * in_dim + k], a1);
    }
    if (o < oe) a0 = fmaf(gy[o], W[(long long)o * in_dim + k], a0);
    og = oe;
  }
  atomicAdd(gx + (long long)r * gx_ld + k, a0 + a1);
}

hipError_t launch_linear_bwd_multi(const LinJob* jobs_dev, int njobs, int total_out, const float* x, int x_ld, int rows, int in_dim,
                                   bool mish_in, float* gx, int gx_ld, hipStream_t s) {
  if (rows <= 0 || njobs <= 0 || total_out <= 0) return hipSuccess;
  const long long tot = (long long)total_out * in_dim;
  hipLaunchKernelGGL(linear_bwd_w_multi_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, jobs_dev, njobs, total_out, x, x_ld, rows,
                     in_dim, mish_in ? 1 : 0);
  if (gx) {
    int nslice = (total_out + 63) / 64;
    if (nslice > 160) nslice = 160;
    const int per_slice = (total_out + nslice - 1) / nslice;
    hipLaunchKernelGGL(linear_bwd_x_multi_kernel, dim3((in_dim + 255) / 256, rows, nslice), dim3(256), 0, s, jobs_dev, njobs, total_out, in_dim,
                       per_slice, gx, gx_ld);
  }
  return hipGetLastError();
}

hipError_t launch_mul_mish_grad(float* g, int g_ld, const float* x, int x_ld, int rows, int n, hipStream_t s) {
  hipLaunchKernelGGL(mul_mish_grad_kernel, dim3((rows * n + 255) / 256), dim3(256), 0, s, g, g_ld, x, x_ld, rows, n);
  return hipGetLastError();
}

// gx must be zeroed by the caller before the first layer accumulates into it; see linear_bwd_x_kernel for the mish' factor
hipError_t launch_linear_bwd(const float* gy, int gy_ld, const float* W, const float* x, int x_ld, int rows, int in_dim, int out_dim,
                             bool mish_in, float* gW, float* gb, float* gx, int gx_ld, hipStream_t s) {
  long long tot = (long long)out_dim * in_dim;
  hipLaunchKernelGGL(linear_bwd_w_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, gy, gy_ld, x, x_ld, rows, in_dim, out_dim,
                     mish_in ? 1 : 0, gW, gb);
  if (gx) {
    int nslice = (out_dim + 63) / 64;
    if (nslice > 32) nslice = 32;
    const int per_slice = (out_dim + nslice - 1) / nslice;
    hipLaunchKernelGGL(linear_bwd_x_kernel, dim3((in_dim + 255) / 256, rows, nslice), dim3(256), 0, s, gy, gy_ld, W, in_dim, out_dim, per_slice,
                       gx, gx_ld);
  }
  return hipGetLastError();
}

// out[b][c] = sum over the item's n pixels of g[p][c]       (time-embedding gradient of a ResnetBlock)
__global__ __launch_bounds__(256) void rowsum_per_item_kernel(const float* __restrict__ g, int ld, long long n, int C, float* __restrict__ out) {
  const int b = blockIdx.z;
  const int c = blockIdx.y * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  __shared__ float red[4][64];
  float acc = 0.f;
  if (c < C) {
    const float* gb = g + (long long)b * n * ld + c;
    const long long st = (long long)gridDim.x * 4;
    long long r = blockIdx.x * 4LL + rl;
    float a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (; r + 3 * st < n; r += 4 * st) {
      acc += gb[r * ld];
      a1 += gb[(r + st) * ld];
      a2 += gb[(r + 2 * st) * ld];
      a3 += gb[(r + 3 * st) * ld];
    }
    for (; r < n; r += st) acc += gb[r * ld];
    acc = (acc + a1) + (a2 + a3);
  }
  red[rl][threadIdx.x & 63] = acc;
  __syncthreads();
  if (rl == 0 && c < C) atomicAdd(&out[(long long)b * C + c], red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

hipError_t launch_rowsum_per_item(const float* g, int ld, int B, long long n, int C, float* out, hipStream_t s) {
  int bx = (int)((n + 255) / 256);
  if (bx < 1) bx = 1;
  if (bx > 128) bx = 128;
  hipLaunchKernelGGL(rowsum_per_item_kernel, dim3(bx, (C + 63) / 64, B), dim3(256), 0, s, g, ld, n, C, out);
  return hipGetLastError();
}

// out[r][0:C] = a[r][0:C] + b[r][0:C] with independent row strides (skip-connection gradient merge)
__global__ void add2_kernel(const float* __restrict__ a, int a_ld, const float* __restrict__ b, int b_ld, float* __restrict__ out, int out_ld,
                            long long rows, int C) {
  const long long total = rows * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / C;
    const int c = (int)(i % C);
    out[r * out_ld + c] = a[r * a_ld + c] + b[r * b_ld + c];
  }
}

hipError_t launch_add2(const float* a, int a_ld, const float* b, int b_ld, float* out, int out_ld, long long rows, int C, hipStream_t s) {
  long long total = rows * C;
  int blocks = (int)((total + 255) / 256);
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(add2_kernel, dim3(blocks), dim3(256), 0, s, a, a_ld, b, b_ld, out, out_ld, rows, C);
  return hipGetLastError();
}

}  // namespace us
