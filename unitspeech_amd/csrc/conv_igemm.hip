// Implicit-GEMM convolution for gfx950 on the exact-fp32 matrix instruction v_mfma_f32_32x32x2_f32.
//
// Serves every GEMM-shaped op of the U-Net score network (unitspeech/unitspeech.py:124-201):
//   3x3 Conv2d of Block (:48) -- directly, or as the 16 per-frequency GEMMs of its Winograd F(2x2,3x3) form (wino.hip), with
//   the output transform optionally inside this kernel (WINO instantiation) --, 1x1 res_conv (:66) / to_qkv (:83) / to_out
//   (:84, with per-item folded weights), Downsample 3x3 stride 2 (:30) and Upsample ConvTranspose2d 4x4 stride 2 (:21, the 4
//   output phases of a 2x2 tap set in one launch), and in training the data gradients of all of them.
// GEMM view: M = output pixels of one item (flattened sub-grid), N = Cout, K = taps * Cin.
//
// Workgroup = 256 threads = 4 waves as 2 (M) x 2 (N); tile TM x 128 with TM = 64 (wave tile 32x64; what every launch uses:
// three co-resident workgroups per CU hide the per-chunk barrier best) or TM = 128 (wave tile 64x64 = 2x2 MFMA 32x32 blocks).
// K advances in chunks of BK input channels of one tap.  Both operands go global -> LDS directly
// (`buffer_load_dwordx4 ... offen lds`: SGPR buffer descriptor + 32-bit per-lane byte offset; no staging registers, no
// ds_write): the A rows are gathered per tap through per-lane offsets, out-of-image taps get an offset beyond the descriptor's
// range and read zeros; the B rows come from weights pre-packed as [tap][Cin/BK][Cout][BK] so a tile row is one
// contiguous BK*4-byte line.  The LDS image is lane-linear as the DMA requires; bank conflicts of the
// ds_read_b128 fragment reads are removed by an XOR swizzle applied on the SOURCE side (lane `pos` of a row
// fetches 16-byte chunk pos ^ swz(row)) and again on the read side, swz(row) = (row / (64/BK)) % (BK/4): the 16
// rows of a ds_read_b128 lane group then hit 16 distinct 16-byte slots of the 256-byte bank window.
// Two LDS buffers; the DMA for chunk s+1 is issued before the MFMAs of chunk s; one barrier per chunk.
//
// Inputs are expected to be already multiplied by the frame mask by their producer (every consumer of such a
// tensor in the reference applies `x * mask`, :54,:74); the epilogue can apply the mask to what it stores.
// k-index convention inside an 8-channel sub-step: lane half hh supplies channels 4*hh+j to MFMA j (j=0..3) on
// BOTH operands, so the pairs (j, 4+j) are summed by instruction j; the K-sum is complete, only its order differs.
#include <type_traits>
#include "kernels.h"
#include "pack_f16.h"

namespace us {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int TN = 128;      // columns per workgroup of the NB = 2 instantiations (host-side sizing)

// One wave-wide LDS-DMA piece: 64 lanes x 16 bytes, lane l lands at lds_dst + 16*l.  `buffer_load_dwordx4 ... offen lds`
// (SGPR descriptor + 32-bit per-lane byte offset + scalar offset) instead of `global_load_lds_dwordx4` (64-bit per-lane
// address): measured on MI355X in the kernel's own skeleton (tools/conv_bench), 6 pieces per 32 MFMAs cost 15 % of the
// MFMA rate as global_load_lds and nothing as buffer loads.  A lane whose offset is >= num_records reads zeros
// (raw-buffer range check), which implements the zero padding of out-of-image taps.
__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff_bytes, unsigned soff_bytes, float* lds_dst_wave_uniform) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst_wave_uniform, 16, (int)voff_bytes,
                                           (int)soff_bytes, 0, 0);
}

// WINO = true: Winograd-domain GEMMs with the output transform in registers.  `in` is V [16][B][th][tw][Cin], `wt` the 16
// per-frequency matrices (wt_bstride apart); the workgroup runs its [TM tiles x TN channels] GEMM for the frequencies
// f = 0..15 in turn and folds each M_f into the four output accumulators Y[r][q] += At[r][f/4] * At[q][f%4] * M_f
// (coefficients 0/+-1, same order as wino_output_kernel, so both forms give the same bits); the epilogue then runs once per
// output position (oy0, ox0) = (r, q) of the 2x2 tile with ostep = 2.  Saves the [16][B][tiles][Cout] round trip through
// HBM and the output-transform pass, for 16x fewer (16x longer) workgroups: used where those still fill the chip.
//
// F16 = true ("f16x3"): the same GEMM at fp32 accuracy on the fp16 matrix cores (v_mfma_f32_32x32x16_f16, 16x the fp32 MFMA
// rate per instruction).  Both operands arrive as TWO fp16 planes per value, x ~= hi + lo * 2^-11 with hi = fp16(x) and
// lo = fp16((x - hi) * 2^11) (22 significant bits; written by the Winograd input transforms / at weight load), stored
// interleaved so that a row of 32 channels is still 128 bytes: [8 x hi][8 x lo] per group of 8 channels.  The DMA, the LDS image
// and its swizzle are therefore byte-for-byte those of the fp32 kernel; only the fragment reads (one 16-byte chunk = 8
// channels of one plane = one MFMA operand) and the MFMAs differ.  Three products per 16-deep step,
//   C_hh += a_hi b_hi;   C_x += a_hi b_lo + a_lo b_hi;   C = C_hh + 2^-11 C_x        (a_lo b_lo ~ 2^-22 is dropped),
// i.e. 16/3 = 5.3x the fp32 MFMA rate; fp16 products are exact in the fp32 accumulator, so the error is the fp32 accumulation's
// (measured: mean error of a K = 1,152 GEMM 3.40e-7 of mean|C| against 3.38e-7 for an fp32 sgemm).  The scaled lo plane is a
// normal fp16 whenever hi is, so no operand scaling is needed for |x| in [6e-5, 65504]; smaller values lose nothing that matters
// (absolute error floor 1.5e-11); a larger one is never clamped: it turns into an infinity and sets the handle's range word
// (kernels.h: split_f16x3 / range_report), which the caller answers by re-running on the exact-fp32 path.
//
// NWM = waves along M (2: 256 threads, 4: 512 threads = two waves per SIMD); NSTG = LDS operand buffers.  NSTG = 3 is the f16x3
// GEMMs' pipeline: their 32-channel step is 24 MFMAs of 32 cycles per wave instead of 32 of 64, too short to cover the latency
// of a DMA issued one step ahead (measured with two buffers: 29 % of the fp16 MFMA rate, waves parked on vmcnt(0) at the
// barrier), so chunk s+2 is issued at step s, a counted `s_waitcnt vmcnt(pieces of one chunk)` retires only chunk s, and the
// barrier is a raw s_barrier (a __syncthreads() would drain the chunk still in flight).
//
// ASPLIT (with F16): the A operand is an ordinary fp32 activation tensor (every direct convolution: 3x3, 1x1, stride 2, transposed);
// it goes through the DMA and the LDS image unchanged and each wave splits the 8 channels of a fragment into the two fp16 planes in
// registers (v_cvt_pk_f16_f32, back-conversion, subtract, scale, v_cvt_pk_f16_f32: about 4 VALU operations per value, issued
// beside the MFMAs).  Only the weights are pre-split.
// NB = 32-column accumulator blocks per wave: 2 (tile TM x 128, what every launch used through round 3) or 1 (tile TM x 64: half the work per
// workgroup, for Winograd-domain GEMMs whose 128-column grid is a fraction of a round of the chip -- a B = 1 level-3 launch is 384 workgroups on
// 256 CUs, two rounds for 1.5 rounds of work; launch_conv_igemm decides)
template <int BK, int WM, bool WINO, bool F16 = false, int NWM = 2, int NSTG = 2, bool ASPLIT = false, int NB = 2>
__global__ __launch_bounds__(NWM * 128, 2) void conv_igemm_kernel(ConvArgs a) {
  constexpr int TN = 64 * NB;        // columns per workgroup: 2 waves (N) x NB blocks of 32
  constexpr int NW = 2 * NWM;        // waves per workgroup, NWM (M) x 2 (N)
  constexpr int TM = NWM * WM;       // rows (pixels) per workgroup
  constexpr int MB = WM / 32;        // 32-row MFMA blocks per wave along M
  constexpr int CPR = BK / 4;        // 16-byte chunks per tile row
  constexpr int RPI = 64 / CPR;      // rows covered by one wave-wide DMA instruction
  constexpr int IA = TM / RPI / NW;  // DMA instructions per wave for the A tile
  constexpr int IB = TN / RPI / NW;  // ... for the B tile
  static_assert(NSTG == 2 || NSTG == 3, "two or three operand buffers");
  constexpr int SWZ_DIV = 64 / BK;   // rows per 256-byte bank window
  constexpr int BUF = (TM + TN) * BK;  // floats per LDS buffer
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
#ifdef US_LIFE      // diagnostic build (tools/conv_bench life): 100 MHz real-time stamps at a workgroup's entry, loop start, loop end and exit
  unsigned long long life_t[5];
#define US_LIFE_AT(i) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(life_t[i])::"memory")
  US_LIFE_AT(0);
#endif
  const int l32 = lane & 31, hh = lane >> 5;
  // Workgroups are dealt round-robin over the 8 XCDs in linear order (x fastest, then z), each XCD with its own 4 MB L2.
  // Winograd-domain GEMMs (xcd_z: blockIdx.z = frequency, every frequency its own A and B): XCD k takes the k-th contiguous
  // eighth of the WHOLE (z, x) range, i.e. whole frequencies, so that the 32 workgroups of a frequency -- which between them read
  // each A tile nt times and each B tile mt times -- share ONE L2 (dealt by x alone they sat on all eight: 342 MB of fabric
  // traffic per launch against 130 MB of operands, rocprofv3 FETCH_SIZE).
  int bt = (int)blockIdx.x, bz = (int)blockIdx.z;
  const bool xcd_z = a.xcd_z && ((gridDim.x * gridDim.z) & 7u) == 0 && !(a.debug & 256);
  if (xcd_z) {
    const unsigned lin = blockIdx.x + gridDim.x * blockIdx.z;
    const unsigned lp = (lin & 7u) * ((gridDim.x * gridDim.z) >> 3) + (lin >> 3);
    bz = (int)(lp / gridDim.x);
    bt = (int)(lp - (unsigned)bz * gridDim.x);
  }
  const int phase = a.nphase > 1 ? bz % a.nphase : 0;
  const int b = a.nphase > 1 ? bz / a.nphase : bz;
  const unsigned long long dy_bits = a.nphase > 1 ? a.ph_dy[phase] : a.dy_bits, dx_bits = a.nphase > 1 ? a.ph_dx[phase] : a.dx_bits,
                           wtap_bits = a.nphase > 1 ? a.ph_wtap[phase] : a.wtap_bits;
  const int a_oy0 = a.nphase > 1 ? (phase >> 1) : a.oy0, a_ox0 = a.nphase > 1 ? (phase & 1) : a.ox0;
  // split-K: blockIdx.x = m_tile * ksplit + ks; slice ks sums the chunks [s_lo, s_lo + S) of the ntaps * nchunk total
  // Otherwise (linear id % 8 == blockIdx.x % 8 when gridDim.x is a multiple of 8): give XCD k the k-th contiguous eighth of the tile range instead of every eighth
  // tile: the 3x3 taps of neighbouring tiles (the image rows above and below) are then fetched into ONE L2 instead of all eight.
  // The grid's x dimension enumerates (row tile, column tile) pairs with the column tile fastest, so the column tiles of one row
  // tile -- which re-read the same A rows -- run back to back on the same XCD (the re-reads used to come from the Infinity Cache:
  // 765 MB of fabric traffic per level-2 Winograd GEMM against 270 MB of operands, rocprofv3 FETCH_SIZE).
  if (!xcd_z && (gridDim.x & 7u) == 0 && !(a.debug & 256)) bt = (bt & 7) * (int)(gridDim.x >> 3) + (bt >> 3);
  const int bx = bt / a.nt;
  const int ks = a.ksplit > 1 ? bx % a.ksplit : 0;
  const int m0 = (a.ksplit > 1 ? bx / a.ksplit : bx) * TM, n0 = (bt - bx * a.nt) * TN;
  const int Ms = a.Hs * a.Ws;
  const int nchunk = a.Cin / BK;
  const int S_all = a.ntaps * nchunk;
  const int s_lo = a.ksplit > 1 ? (int)(((long long)S_all * ks) / a.ksplit) : 0;
  const int S = a.ksplit > 1 ? (int)(((long long)S_all * (ks + 1)) / a.ksplit) - s_lo : S_all;

  // ---- DMA source bookkeeping: this lane feeds LDS position (row = rbase + lane/CPR, chunk slot = lane%CPR) ----
  const int lrow = lane / CPR, lpos = lane % CPR;
  int my[IA], mx[IA], achunk[IA];
  bool mv[IA];
#pragma unroll
  for (int j = 0; j < IA; ++j) {
    const int r = (wave * IA + j) * RPI + lrow;      // row inside the A tile
    const int m = m0 + r;
    mv[j] = m < Ms;
    const int mc = mv[j] ? m : 0;
    my[j] = mc / a.Ws;
    mx[j] = mc - my[j] * a.Ws;
    achunk[j] = (lpos ^ ((r / SWZ_DIV) % CPR)) * 4;   // source chunk (floats) for this LDS slot
  }
  unsigned boff[IB];    // byte offset of this lane's B source inside a [Cout][BK] slab
#pragma unroll
  for (int j = 0; j < IB; ++j) {
    const int r = (wave * IB + j) * RPI + lrow;
    int n = n0 + r;
    n = n < a.Cout ? n : a.Cout - 1;                  // clamp: columns >= Cout are never stored
    boff[j] = (unsigned)(n * BK + (lpos ^ ((r / SWZ_DIV) % CPR)) * 4) * 4u;
  }
  // buffer descriptors: A = this item's activation tensor (range-checked), B = this item's packed weights
  const unsigned a_bytes = (unsigned)a.Hin * (unsigned)a.Win * (unsigned)a.in_ld * 4u;     // < 2^31 (host-checked)
  const long long a_item = (long long)a.Hin * a.Win * a.in_ld;
  __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (long long)b * a_item), 0, (int)a_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.wt + (WINO ? 0LL : (long long)(a.wt_bdiv > 1 ? b / a.wt_bdiv : b) * a.wt_bstride)), 0, 0x7fffffff, 0x00020000);

  unsigned aoff[IA];    // per-lane byte offset of the current tap's source row chunk (>= a_bytes: reads zeros)
  unsigned wtap_bytes = 0;
  const int wrows = a.wt_rows ? a.wt_rows : a.Cout;      // rows per K-chunk of the packed weight (ConvArgs::wt_rows)
  auto setup_tap = [&](int tap) {
    const int dy = (int)((dy_bits >> (4 * tap)) & 15) - 8;
    const int dx = (int)((dx_bits >> (4 * tap)) & 15) - 8;
    const int wt_i = (int)((wtap_bits >> (4 * tap)) & 15);
#pragma unroll
    for (int j = 0; j < IA; ++j) {
      const int iy = my[j] * a.istride + dy, ix = mx[j] * a.istride + dx;
      const bool ok = mv[j] && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
      const unsigned off = ((unsigned)(iy * a.Win + ix) * (unsigned)a.in_ld + (unsigned)achunk[j]) * 4u;
      aoff[j] = ok ? off : a_bytes;
#ifdef US_CONV_ABLATE
      if (a.debug & 32) aoff[j] = (unsigned)(j * 4096 + lane * 16) & 0xffff;      // every load hits one hot 64 KB window
#endif
    }
    wtap_bytes = (unsigned)wt_i * (unsigned)nchunk * (unsigned)wrows * (unsigned)(BK * 4);
#ifdef US_CONV_ABLATE
    if (a.debug & 32) wtap_bytes = 0;
#endif
  };
  auto dma = [&](int ch, int buf) {
    float* As = smem + buf * BUF;
    float* Bs = As + TM * BK;
    unsigned ach = (unsigned)ch * (unsigned)(BK * 4);
    unsigned wb = wtap_bytes + (unsigned)ch * (unsigned)wrows * (unsigned)(BK * 4);
#ifdef US_CONV_ABLATE
    if (a.debug & 32) { ach = 0; wb = 0; }
#endif
#pragma unroll
    for (int j = 0; j < IA; ++j) blds16(rsrc_a, aoff[j], ach, As + (wave * IA + j) * RPI * BK);
#pragma unroll
    for (int j = 0; j < IB; ++j) blds16(rsrc_b, boff[j], wb, Bs + (wave * IB + j) * RPI * BK);
  };

  // Two-level accumulation: the MFMA chain (an exact fp32 fma chain) runs over 128 K-elements, then is folded into
  // `total`.  A single chain over K = 9*Cin (up to 18,432) would carry ~0.2*sqrt(K) ulp of rounding error (19 ulp
  // at K = 9,216); chunks of ~sqrt(K) bring it to ~3 ulp, on par with a blocked CPU sgemm.
#ifndef US_FLUSH_K
#define US_FLUSH_K 128
#endif
  constexpr int kFlushSteps = US_FLUSH_K / BK;
  f32x16 acc[MB][NB], total[MB][NB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; total[i][j][r] = 0.f; }
  int since_flush = 0;

  // fragment read offsets (floats): row R, 16-byte chunk (2s+hh) ^ swz(R); swz only depends on the lane
  const int sw = (l32 / SWZ_DIV) % CPR;
  const int a_row = (wm * WM + l32) * BK;
  const int b_row = TM * BK + (wn * (32 * NB) + l32) * BK;

#ifndef US_PRIO_MODE
#define US_PRIO_MODE 0
#endif
#if US_PRIO_MODE == 1
  // static per-workgroup priority: co-resident workgroups that would otherwise march in lockstep (same code, same
  // barrier cadence, one shared matrix pipe per SIMD) take turns instead
  {
    const unsigned hsh = (blockIdx.x * 2654435761u + blockIdx.y * 40503u + blockIdx.z * 7u) >> 13;
    if ((hsh & 3) == 1) __builtin_amdgcn_s_setprio(1);
    else if ((hsh & 3) == 2) __builtin_amdgcn_s_setprio(2);
    else if ((hsh & 3) == 3) __builtin_amdgcn_s_setprio(3);
  }
#endif
  constexpr int NY = WINO ? 4 : 1;
  f32x16 Y[NY][MB][NB];       // WINO: the 2x2 output accumulators (dead otherwise)
  if (WINO) {
#pragma unroll
    for (int y = 0; y < NY; ++y)
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) Y[y][i][j][r] = 0.f;
  }
  // WINO: the 16 frequencies form ONE software-pipelined sequence of 16 * nchunk steps (the DMA of a frequency's first chunk
  // is in flight while the previous frequency's last chunk is multiplied), so a K = Cin of 128 does not drain the pipeline
  // 16 times per workgroup.
  auto setup_freq = [&](int f) {
    rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + ((long long)f * a.B + b) * a_item), 0, (int)a_bytes, 0x00020000);
    rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)(a.wt + (long long)f * a.wt_bstride), 0, 0x7fffffff, 0x00020000);
  };
  // a frequency whose K = Cin fits one accumulation chain never touches `total`; F16: `acc` is C_hh and `total` is C_x
  const bool two_level = !F16 && (!WINO || nchunk > kFlushSteps);
  auto fold = [&](int f) {
    // At = [1 1 1 0; 0 1 -1 -1]: coefficient of M_f (f = 4*fi + fj) in output (r, q) is At[r][fi] * At[q][fj]; 7 of the 16
    // frequencies feed one output, 6 feed two, 4 feed four: zero coefficients are skipped (wave-uniform branches)
    const int fi = f >> 2, fj = f & 3;
    const float r0 = fi < 3 ? 1.f : 0.f, r1 = fi == 0 ? 0.f : (fi == 1 ? 1.f : -1.f);
    const float q0 = fj < 3 ? 1.f : 0.f, q1 = fj == 0 ? 0.f : (fj == 1 ? 1.f : -1.f);
    const float cf[4] = {r0 * q0, r0 * q1, r1 * q0, r1 * q1};
    if (F16) {
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) { acc[i][j][r] = __builtin_fmaf(total[i][j][r], 0x1p-11f, acc[i][j][r]); total[i][j][r] = 0.f; }
    }
    if (two_level) {
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          acc[i][j] += total[i][j];
#pragma unroll
          for (int r = 0; r < 16; ++r) total[i][j][r] = 0.f;
        }
    }
#pragma unroll
    for (int y = 0; y < NY; ++y) {
      if (cf[y] != 0.f) {
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
          for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) Y[y][i][j][r] = __builtin_fmaf(cf[y], acc[i][j][r], Y[y][i][j][r]);
      }
    }
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    since_flush = 0;
  };
  const int S_run = WINO ? 16 * nchunk : S;      // steps of this workgroup's pipeline
  int tap_n = s_lo / nchunk, ch_n = s_lo - tap_n * nchunk;   // WINO: tap_n counts frequencies
  if (WINO) setup_freq(0);
  setup_tap(WINO ? 0 : tap_n);
  dma(ch_n, 0);
  if (++ch_n == nchunk) { ch_n = 0; ++tap_n; }
  if (NSTG == 3) {
    if (S_run > 1) {
      if (ch_n == 0) {
        if (WINO) setup_freq(tap_n);
        else setup_tap(tap_n);
      }
      dma(ch_n, 1);
      if (++ch_n == nchunk) { ch_n = 0; ++tap_n; }
    }
  } else {
    __syncthreads();
  }

  // Fragment reads are double-buffered in registers (set 0 / set 1) and the MFMAs of a step's LAST sub-step are
  // deferred until after the barrier and after the next step's first fragment reads have been issued, so the LDS
  // latency behind the barrier is covered by 8*MB MFMAs instead of idling the matrix pipe.
  constexpr int NS = F16 ? 2 : BK / 8;    // F16: two 16-deep steps per 32-channel chunk
  constexpr int NP = F16 ? 2 : 1;         // fp16 planes per operand
  f32x4 fa0[MB * NP], fb0[NB * NP], fa1[MB * NP], fb1[NB * NP];
  float amax = 0.f;                       // ASPLIT: largest magnitude this wave split (range report after the loop)
  auto load_frags = [&](f32x4* fa, f32x4* fb, const float* base, int s_) {
    if (F16) {
      // chunk (2 * kgroup + plane) of the row; lane half hh supplies channels 8 * (2 s + hh) .. + 7 of the chunk's 32
#pragma unroll
      for (int p = 0; p < 2; ++p) {
#ifdef US_ABL_LDS      // timing experiment: half the fragment reads (the lo planes alias the hi planes; results are wrong)
        if (p == 1 && !ASPLIT) {
          for (int i = 0; i < MB; ++i) fa[i * 2 + 1] = fa[i * 2];
          for (int nb = 0; nb < NB; ++nb) fb[nb * 2 + 1] = fb[nb * 2];
          continue;
        }
#endif
        const int co = ((((2 * s_ + hh) << 1) + p) ^ sw) * 4;
        if (!ASPLIT) {
#pragma unroll
          for (int i = 0; i < MB; ++i) fa[i * 2 + p] = *reinterpret_cast<const f32x4*>(base + a_row + i * 32 * BK + co);
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) fb[nb * 2 + p] = *reinterpret_cast<const f32x4*>(base + b_row + nb * 32 * BK + co);
      }
      if (ASPLIT) {
        // fp32 row: the same 8 channels are two 16-byte chunks of 4 floats; split them here
#pragma unroll
        for (int i = 0; i < MB; ++i) {
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(base + a_row + i * 32 * BK + ((((2 * s_ + hh) << 1)) ^ sw) * 4);
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(base + a_row + i * 32 * BK + ((((2 * s_ + hh) << 1) + 1) ^ sw) * 4);
          half8 hi, lo;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            // no clamp: a value beyond the fp16 range becomes an infinity (non-finite outputs, never a look-alike), a NaN stays one, and
            // the running maximum (one v_max3_f32 per pair, abs modifiers free; a NaN never enters it) reports it once after the loop
            const float x0 = v0[k], x1 = v1[k];
            amax = fmaxf(amax, fmaxf(fabsf(x0), fabsf(x1)));
            const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
            hi[k] = h0; hi[4 + k] = h1;
            lo[k] = (_Float16)((x0 - (float)h0) * 2048.f);
            lo[4 + k] = (_Float16)((x1 - (float)h1) * 2048.f);
          }
          fa[i * 2] = __builtin_bit_cast(f32x4, hi);
          fa[i * 2 + 1] = __builtin_bit_cast(f32x4, lo);
        }
      }
#ifndef US_NO_SCHEDBAR
      __builtin_amdgcn_sched_barrier(0);
#endif
      return;
    }
    const int co = ((2 * s_ + hh) ^ sw) * 4;
#pragma unroll
    for (int i = 0; i < MB; ++i) fa[i] = *reinterpret_cast<const f32x4*>(base + a_row + i * 32 * BK + co);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) fb[nb] = *reinterpret_cast<const f32x4*>(base + b_row + nb * 32 * BK + co);
    // keep the reads AHEAD of the MFMAs that follow in program order (hipcc otherwise sinks them behind the
    // MFMA block and waits for them at once, exposing the LDS latency)
#ifndef US_NO_SCHEDBAR
    __builtin_amdgcn_sched_barrier(0);
#endif
  };
  // NSTG == 3: the DMA of chunk s+2 is issued from inside the first MFMA block of step s (it has two steps to land), so its
  // issue slots (M0 write + buffer_load per piece) run under executing MFMAs instead of ahead of them with the matrix pipe idle
  bool dma_pending = false;
  int dma_buf = 0;
  // F16: the pieces of a chunk (IA + IB LDS-DMA instructions per wave) are spread over the step's MFMAs instead of issued back to
  // back: eight waves issuing 48 of them at once fill the CU's one address queue, and the waves (and the MFMAs queued behind their
  // loads) wait there (tools/conv_bench: the burst costs 15 % of the Winograd-domain GEMMs, twice as many instructions 40 %)
#ifndef US_DMA_SPREAD
#define US_DMA_SPREAD 1
#endif
  constexpr int NPIECE = IA + IB;
  constexpr int SLOTS = 6 * MB * NB;                  // F16 MFMAs of a wave per chunk
  constexpr int PSTRIDE = SLOTS / NPIECE > 0 ? SLOTS / NPIECE : 1;
  unsigned pend_ach = 0, pend_wb = 0;
  auto dma_piece = [&](int k) {
    if (!dma_pending) return;
#ifndef US_NO_SCHEDBAR
    __builtin_amdgcn_sched_barrier(0);
#endif
    if (k == 0) {
      if (ch_n == 0) {
        if (WINO) setup_freq(tap_n);
        else setup_tap(tap_n);
      }
      pend_ach = (unsigned)ch_n * (unsigned)(BK * 4);
      pend_wb = wtap_bytes + (unsigned)ch_n * (unsigned)wrows * (unsigned)(BK * 4);
    }
    float* As = smem + dma_buf * BUF;
    float* Bs = As + TM * BK;
    if (k < IA) blds16(rsrc_a, aoff[k < IA ? k : 0], pend_ach, As + (wave * IA + k) * RPI * BK);
    else blds16(rsrc_b, boff[k < IA ? 0 : k - IA], pend_wb, Bs + (wave * IB + (k - IA)) * RPI * BK);
    if (k == NPIECE - 1) {
      if (++ch_n == nchunk) { ch_n = 0; ++tap_n; }
      dma_pending = false;
    }
#ifndef US_NO_SCHEDBAR
    __builtin_amdgcn_sched_barrier(0);
#endif
  };
  auto dma_slot = [&](int sidx) {
    if (US_DMA_SPREAD && F16 && sidx % PSTRIDE == 0 && sidx / PSTRIDE < NPIECE) dma_piece(sidx / PSTRIDE);
  };
  auto dma_late = [&]() {
    if (NSTG == 3 && dma_pending) {
#ifndef US_NO_SCHEDBAR
      __builtin_amdgcn_sched_barrier(0);
#endif
      if (ch_n == 0) {
        if (WINO) setup_freq(tap_n);
        else setup_tap(tap_n);
      }
      dma(ch_n, dma_buf);
      if (++ch_n == nchunk) { ch_n = 0; ++tap_n; }
      dma_pending = false;
#ifndef US_NO_SCHEDBAR
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
  };
  auto mma = [&](const f32x4* fa, const f32x4* fb, int phase = 0) {      // phase: first / second 16-deep step of the chunk (F16)
    if (F16) {
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const half8 ah = __builtin_bit_cast(half8, fa[i * 2]), al = __builtin_bit_cast(half8, fa[i * 2 + 1]);
          const half8 bh = __builtin_bit_cast(half8, fb[j * 2]), bl = __builtin_bit_cast(half8, fb[j * 2 + 1]);
          const int s0 = phase * (SLOTS / 2) + (i * NB + j) * 3;
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i][j], 0, 0, 0);
          dma_slot(s0);
          total[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, total[i][j], 0, 0, 0);
          dma_slot(s0 + 1);
#ifndef US_EXP_TWO_PRODUCT      // experiment build: drop a_lo * b_hi, i.e. the A operand at fp16 precision
          total[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, total[i][j], 0, 0, 0);
#endif
          dma_slot(s0 + 2);
          if (!US_DMA_SPREAD && i == 0 && j == 0) dma_late();
        }
      return;
    }
#if US_PRIO_MODE == 2
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < MB; ++i) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[i][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][j], fb[nb][j], acc[i][nb], 0, 0, 0);
      }
#if US_PRIO_MODE == 2
    __builtin_amdgcn_s_setprio(0);
#endif
  };
  auto step_done = [&]() {
    if (two_level && ++since_flush == kFlushSteps) {
      since_flush = 0;
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          total[i][j] += acc[i][j];
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
    }
  };

#ifdef US_LIFE
  US_LIFE_AT(1);
#endif
  if (NSTG == 3) {
    // static priority for the second-dispatched half of an 8-wave workgroup (the arbitration loser of every segment); experiment bit
    if (NW == 8 && (a.debug & 512) && wave >= 4) __builtin_amdgcn_s_setprio(1);
    // chunk s lives in buffer s % 3.  At step s: wait for this wave's pieces of chunk s (chunk s+1 may stay in flight), barrier
    // (everybody's pieces have landed, and everybody is done reading chunk s-1, whose buffer chunk s+2 is about to overwrite),
    // issue chunk s+2, multiply chunk s.
    int cur = 0;
#ifdef US_STAMP      // diagnostic build (tools/conv_bench): where does a step of the three-buffer loop spend its cycles?
    unsigned long long st_wait = 0, st_bar = 0, st_body = 0, st_t0, st_t1, st_t2;
#define US_STAMP_AT(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory")
    US_STAMP_AT(st_t0);
#endif
    for (int step = 0; step < S_run; ++step) {
      if (step + 1 < S_run) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IA + IB) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef US_STAMP
      US_STAMP_AT(st_t1);
      st_wait += st_t1 - st_t0;
#endif
#ifdef US_CONV_ABLATE
      if (!(a.debug & 1024))       // timing ablation: no barrier in the three-buffer loop (races: results are wrong)
#endif
      __builtin_amdgcn_s_barrier();
#ifdef US_STAMP
      US_STAMP_AT(st_t2);
      st_bar += st_t2 - st_t1;
#endif
#ifdef US_ABL_DMA       // timing experiment: no loads in the loop (stale tiles; results are wrong)
      if (false) {
#else
      if (step + 2 < S_run) {
#endif
        dma_pending = true;
        dma_buf = cur == 0 ? 2 : cur - 1;
        // fp32 MFMA form, and the first step of the deferred loop (whose first MFMA block does not run): all pieces at once
        if (!F16 || (a.debug & 64) || (US_DMA_SPREAD && step == 0 && !WINO)) dma_late();
      }
      const float* base = smem + cur * BUF;
      // as in the two-buffer loop, the MFMAs of a chunk's last 16-deep step run after the next barrier, under the first
      // fragment reads of the next chunk (8 waves x 16 ds_read_b128 queue up behind every barrier)
      if (WINO && F16) {
        // fused output transform: the four Y accumulators leave no room for a second fragment set (256 VGPRs at two waves per
        // SIMD), so the two 16-deep steps of a chunk run one after the other; the co-resident workgroups cover the read latency
        load_frags(fa0, fb0, base, 0);
        mma(fa0, fb0, 0);
        load_frags(fa0, fb0, base, 1);
        mma(fa0, fb0, 1);
        if ((step + 1) % nchunk == 0) fold(step / nchunk);
        cur = cur == 2 ? 0 : cur + 1;
        continue;
      }
      load_frags(fa0, fb0, base, 0);
      if (step > 0) {
        mma(fa1, fb1, 0);
        step_done();
        if (WINO && step % nchunk == 0) fold(step / nchunk - 1);     // ... which completed a frequency
      }
      load_frags(fa1, fb1, base, 1);
      mma(fa0, fb0, 1);
      if (NS == 4) {
        load_frags(fa0, fb0, base, 2);
        mma(fa1, fb1);
        load_frags(fa1, fb1, base, 3);
        mma(fa0, fb0);
      }
      cur = cur == 2 ? 0 : cur + 1;
#ifdef US_STAMP
      US_STAMP_AT(st_t0);
      st_body += st_t0 - st_t2;
#endif
    }
#ifdef US_STAMP
    if (a.stamp_out && lane == 0) {
      unsigned long long* o = a.stamp_out + ((blockIdx.x + gridDim.x * blockIdx.z) * NW + wave) * 4;
      o[0] = st_wait; o[1] = st_bar; o[2] = st_body; o[3] = (unsigned long long)S_run;
    }
#endif
    if (!(WINO && F16)) {
      mma(fa1, fb1);
      step_done();
      if (WINO) fold(15);
    }
  } else {
  for (int step = 0; step < S_run; ++step) {
    const bool has_next = step + 1 < S_run;
#ifndef US_DMA_SPREAD2
#define US_DMA_SPREAD2 0       // two-buffer loop: spreading measured within noise either way (three workgroups per CU already interleave)
#endif
    if (has_next && US_DMA_SPREAD2 && F16 && step > 0) {
      dma_pending = true;          // the pieces go out between this step's MFMAs (dma_slot)
      dma_buf = (step + 1) & 1;
    } else if (has_next) {
      if (ch_n == 0) {
        if (WINO) setup_freq(tap_n);
        else setup_tap(tap_n);
      }
#ifdef US_CONV_ABLATE
      if (!(a.debug & 1))
#endif
      dma(ch_n, (step + 1) & 1);
      if (++ch_n == nchunk) { ch_n = 0; ++tap_n; }
    }
    const float* base = smem + (step & 1) * BUF;
#ifdef US_CONV_ABLATE      // timing ablations of tools/conv_bench only (results are wrong by construction)
    if ((a.debug & 2) && step > 0) {
      mma(fa1, fb1); step_done(); mma(fa0, fb0);
      if (NS == 4) { mma(fa1, fb1); mma(fa0, fb0); }
      if (!(a.debug & 4)) __syncthreads();
      continue;
    }
#endif
#ifdef US_NO_DEFER
    load_frags(fa0, fb0, base, 0);
    load_frags(fa1, fb1, base, 1);
    mma(fa0, fb0);
    if (NS == 4) {
      load_frags(fa0, fb0, base, 2);
      mma(fa1, fb1);
      load_frags(fa1, fb1, base, 3);
      mma(fa0, fb0);
    }
    mma(fa1, fb1);
    step_done();
    if (WINO && (step + 1) % nchunk == 0) fold(step / nchunk);
#else
    load_frags(fa0, fb0, base, 0);
    if (step > 0) {           // last sub-step of the previous chunk (fragments were read before the barrier)
      mma(fa1, fb1, 0);
      step_done();
      if (WINO && step % nchunk == 0) fold(step / nchunk - 1);     // ... which completed a frequency
    }
    load_frags(fa1, fb1, base, 1);
    mma(fa0, fb0, 1);
    if (NS == 4) {
      load_frags(fa0, fb0, base, 2);
      mma(fa1, fb1);
      load_frags(fa1, fb1, base, 3);
      mma(fa0, fb0);
    }
#endif
#ifdef US_CONV_ABLATE
    if (!(a.debug & 4))
#endif
    __syncthreads();   // drains this wave's DMA and fragment reads (vmcnt(0), lgkmcnt(0)) and orders every wave's
                       // reads of this buffer before its next overwrite
  }
#ifndef US_NO_DEFER
  mma(fa1, fb1);
  if (WINO) fold(15);
#endif
  }   // NSTG == 2
#ifdef US_LIFE
  US_LIFE_AT(2);
#endif
  if (ASPLIT) range_report(a.range_flag, amax >= kF16Over, kRangeAct);
  if (!WINO) {
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        if (F16) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = __builtin_fmaf(total[i][j][r], 0x1p-11f, acc[i][j][r]);
        } else {
          acc[i][j] += total[i][j];
        }
      }
  }

  // ---- epilogue: C/D layout of the 32x32 block: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) ----
  const bool dense = (a.ostep == 1 && a_oy0 == 0 && a_ox0 == 0 && a.Hs == a.Hout && a.Ws == a.Wout);
  const float alpha = a.alpha ? *a.alpha : 1.f;
  float* out_b = a.out + (long long)b * a.Hout * a.Wout * a.out_ld;
  const float* add_b = a.add ? a.add + (long long)b * a.Hout * a.Wout * a.add_ld : nullptr;
  const float* om_b = a.omask ? a.omask + (long long)(b % a.omask_bmod) * a.omask_ld : nullptr;
  // GroupNorm partial sums: fp32 per lane, except in the Winograd form where they are kept in fp64 so that the fused and the
  // separate output transform (chosen by launch geometry, i.e. by the batch) agree to fp64 rounding
  typedef typename std::conditional<WINO, double, float>::type stat_t;
  stat_t gsum[NB] = {}, gsq[NB] = {};
#ifdef US_CONV_ABLATE
  if (a.debug & 8) {       // timing ablation: no epilogue at all (keep the accumulators alive)
    float keep = 0.f;
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) keep += acc[i][j][r];
    if (keep == 1.2345e-30f) a.out[0] = keep;
    return;
  }
#endif
  if (a.ksplit > 1) {
    // raw partial sums into this slice's slab [ks][B][Ms][Cout]; splitk_finish_kernel sums the slices in a fixed order and
    // applies bias / Rezero / residual / mask / GroupNorm sums (deterministic, unlike float atomics)
    float* slab = a.splitk_ws + (((long long)ks * a.B + b) * Ms) * a.Cout;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * WM + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (m < Ms) {
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            const int n = n0 + wn * (32 * NB) + nb * 32 + l32;
            if (n < a.Cout) slab[(long long)m * a.Cout + n] = acc[mb][nb][r];
          }
        }
      }
    return;
  }
  // Each 32x32 accumulator block (column on the lane, rows in registers) is turned around through a private LDS patch
  // so that a lane owns 4 consecutive channels of one pixel: 16-byte stores / residual loads, 8 pixel rows x 128 bytes per
  // wave instruction, 4x fewer memory instructions than the native layout (the epilogue was half the time of the K=128
  // 1x1 convolutions and 11 % of the level-0 3x3s).
  __syncthreads();                                   // every wave is done with the operand buffers
  if constexpr (!WINO && (MB == 1 || MB == 2) && NB == 2) {
    if (a.attn_part_ctx && n0 >= a.attn_q_cols) {
      // to_qkv column tiles 1 / 2: this wave's two 32-column blocks are k_h and v_h of head h for 32 of the tile's 64 rows
      // (qkv_src_row).  Linear attention (unitspeech/unitspeech.py:91-92): k = softmax over ALL n positions, ctx = k v^T; here the
      // online-softmax partial of the 64 rows: column maxima m, P = exp(k - m), s = sum_rows P, ctx = P^T v.  A 32x32 accumulator
      // block has its column on the lane and its rows in the registers, which is exactly the operand layout of
      // v_mfma_f32_32x32x2_f32 for a product that sums over ROWS: register r of P and of V go in as they stand.
      const int h = ((n0 - a.attn_q_cols) / TN) * 2 + wn;
      float* xm = smem;                  // [wn][wm][32]  column maxima of each wave
      float* xs = smem + 128;            // [wn][32]      column sums of wave wm = 1
      float* xc = smem + 256;            // [wn][32][32]  ctx of wave wm = 1
      // (MB = 2: the 128-row tile, ConvArgs::attn_rows = 128 -- each wave holds 64 rows of k_h | v_h as two 32-row blocks; one chunk of
      // partials per 128 rows: half the hand-offs, half the partials to merge)
      const int mrow0 = m0 + wm * WM + 4 * hh;
      float kmax = -INFINITY;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool ok = mrow0 + mb * 32 + (r & 3) + 8 * (r >> 2) < Ms;
          kmax = fmaxf(kmax, ok ? acc[mb][0][r] : -INFINITY);
        }
      kmax = fmaxf(kmax, __shfl_xor(kmax, 32));
      if (hh == 0) xm[(wn * 2 + wm) * 32 + l32] = kmax;
      __syncthreads();
      const float mcol = fmaxf(xm[(wn * 2) * 32 + l32], xm[(wn * 2 + 1) * 32 + l32]);     // finite: row m0 is always valid
      f32x16 ctx;
#pragma unroll
      for (int r = 0; r < 16; ++r) ctx[r] = 0.f;
      float ssum = 0.f;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool ok = mrow0 + mb * 32 + (r & 3) + 8 * (r >> 2) < Ms;
          const float p = ok ? expf(acc[mb][0][r] - mcol) : 0.f;
          const float v = ok ? acc[mb][1][r] : 0.f;
          ssum += p;
          ctx = __builtin_amdgcn_mfma_f32_32x32x2f32(p, v, ctx, 0, 0, 0);
        }
      ssum += __shfl_xor(ssum, 32);
      if (wm == 1) {
        if (hh == 0) xs[wn * 32 + l32] = ssum;
#pragma unroll
        for (int r = 0; r < 16; ++r) xc[wn * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * hh) * 32 + l32] = ctx[r];
      }
      __syncthreads();
      if (wm == 0) {
        const long long blk = (long long)b * a.attn_nchunks + bx;
        float* pc = a.attn_part_ctx + (blk * kHeads + h) * (kDimHead * kDimHead);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int d = (r & 3) + 8 * (r >> 2) + 4 * hh;
          pc[d * kDimHead + l32] = ctx[r] + xc[wn * 1024 + d * 32 + l32];
        }
        if (hh == 0) {
          a.attn_part_m[blk * kHidden + h * kDimHead + l32] = mcol;
          a.attn_part_s[blk * kHidden + h * kDimHead + l32] = ssum + xs[wn * 32 + l32];
        }
      }
      return;
    }
  }
  float* tr = smem + wave * (32 * 36);               // 32 rows x (32 + 4 pad) floats per wave
  const bool need_xy = !dense || om_b != nullptr;
  const int trow = lane >> 3, tc4 = (lane & 7) * 4;  // read-back role: row trow + 8k, channels tc4..tc4+3
  // WINO with an odd image height / width: the last tile row / column has output positions outside the image
  const bool edge = WINO && (((a.Hout | a.Wout) & 1) != 0);
  // bias of this wave's two 32-column blocks, loaded ONCE: a global load inside the block loop would make every block wait
  // (vmcnt retires in order) for the previous block's stores to reach memory
  bool out_over = false;          // out_split: a stored value beyond the fp16 range
  float bias_col[NB];
  f32x4 bias_quad[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int ncol = n0 + wn * (32 * NB) + nb * 32;
    bias_col[nb] = (a.bias && ncol + l32 < a.Cout) ? a.bias[ncol + l32] : 0.f;
    bias_quad[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.bias && ncol + tc4 < a.Cout) bias_quad[nb] = *reinterpret_cast<const f32x4*>(a.bias + ncol + tc4);
  }
#pragma unroll
  for (int yi = 0; yi < NY; ++yi) {
  const f32x16 (&A)[MB][NB] = WINO ? Y[yi] : acc;
  const int oy0 = WINO ? (yi >> 1) : a_oy0, ox0 = WINO ? (yi & 1) : a_ox0;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m_base = m0 + wm * WM + mb * 32;
    int yb = 0, xb = 0;
    if (need_xy) { yb = m_base / a.Ws; xb = m_base - yb * a.Ws; }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int ncol = n0 + wn * (32 * NB) + nb * 32;
      {   // GroupNorm sums in the native layout (this lane's column, its 16 rows)
        const int n = ncol + l32;
        const float bv = bias_col[nb];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2) + 4 * hh;
          const int m = m_base + dr;
          const float v = A[mb][nb][r] + bv;
          bool ok = m < Ms && n < a.Cout;
          if (edge) {
            const int yy = m / a.Ws, xx = m - yy * a.Ws;
            ok = ok && (oy0 + 2 * yy < a.Hout) && (ox0 + 2 * xx < a.Wout);
          }
          if (ok) { gsum[nb] += (stat_t)v; gsq[nb] += (stat_t)(v * v); }
          tr[dr * 36 + l32] = A[mb][nb][r];
        }
      }
      const int n = ncol + tc4;
      const f32x4 b4 = bias_quad[nb];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int dr = trow + 8 * k;
        const int m = m_base + dr;
        f32x4 v = *reinterpret_cast<const f32x4*>(tr + dr * 36 + tc4);
        if (m < Ms && n < a.Cout) {
          int yy = yb, xx = xb + dr;
          if (need_xy) while (xx >= a.Ws) { xx -= a.Ws; ++yy; }
          const int ox = ox0 + xx * a.ostep;
          const int oy = oy0 + yy * a.ostep;
          if (edge && (oy >= a.Hout || ox >= a.Wout)) continue;
          const long long pix = dense ? (long long)m : (long long)oy * a.Wout + ox;
          v = v + b4;
          if (a.out2) *reinterpret_cast<f32x4*>(a.out2 + ((long long)b * a.Hout * a.Wout + pix) * a.out2_ld + n) = v;
          v = v * alpha;
          if (add_b) v += *reinterpret_cast<const f32x4*>(add_b + pix * a.add_ld + n);
          if (om_b) v *= om_b[ox * a.omask_step];
          if (a.out_split) {
            typedef _Float16 half4o __attribute__((ext_vector_type(4)));
            half4o hi, lo;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              us_half h_, l_;
              split_f16x3(v[q], h_, l_, out_over);
              hi[q] = h_;
              lo[q] = l_;
            }
            _Float16* oh = reinterpret_cast<_Float16*>(out_b + pix * a.out_ld) + 2 * (n & ~7) + (n & 7);
            *reinterpret_cast<half4o*>(oh) = hi;
            *reinterpret_cast<half4o*>(oh + 8) = lo;
          } else {
            *reinterpret_cast<f32x4*>(out_b + pix * a.out_ld + n) = v;
          }
        }
      }
    }
  }
  }   // output positions
  if (a.out_split) range_report(a.range_flag, out_over, kRangeAct);
  if (a.stats) {
    // GroupNorm(8) partial sums of the conv output (pre-alpha/add/mask); Cout/8 is a power of two (host-checked)
    const int cg = a.Cout / kGroups;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      stat_t s1 = gsum[nb], s2 = gsq[nb];
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      const int seg = cg < 32 ? cg : 32;
      for (int off = 1; off < seg; off <<= 1) {
        s1 += __shfl_xor(s1, off);
        s2 += __shfl_xor(s2, off);
      }
      const int n = n0 + wn * (32 * NB) + nb * 32 + l32;
      if (hh == 0 && (l32 % seg) == 0 && n < a.Cout) {
        stat_add(a.stats, b, n / cg, 0, bx, (double)s1);
        stat_add(a.stats, b, n / cg, 1, bx, (double)s2);
      }
    }
  }
#ifdef US_LIFE
  US_LIFE_AT(4);                                         // every store issued
#ifdef US_LIFE_DRAIN
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the stores have left the wave)
#endif
  US_LIFE_AT(3);
  if (a.stamp_out && lane == 0) {
    unsigned long long* o = a.stamp_out + ((blockIdx.x + gridDim.x * blockIdx.z) * NW + wave) * 4;
    o[0] = life_t[0]; o[1] = life_t[1]; o[2] = life_t[2]; o[3] = life_t[3];
  }
#endif
}

// Second half of a split-K convolution: out = epilogue(sum_ks slab[ks]) with the same epilogue as the single-pass kernel.
__global__ __launch_bounds__(256) void splitk_finish_kernel(ConvArgs a) {
  __shared__ double s_g[kGroups][2];
  const int b = blockIdx.y;
  const int Ms = a.Hs * a.Ws;
  const int C4 = a.Cout >> 2;
  const long long total = (long long)Ms * C4;
  if (threadIdx.x < kGroups * 2) s_g[threadIdx.x >> 1][threadIdx.x & 1] = 0.0;
  __syncthreads();
  // GroupNorm sums: every thread keeps fp32 partials of its own elements (fixed order), merged in fp64 (order-insensitive
  // at fp32 resolution) -- float atomics here would make the statistics, hence the output, vary from run to run
  float t1[4] = {0.f, 0.f, 0.f, 0.f}, t2[4] = {0.f, 0.f, 0.f, 0.f};
  int tn = -1;
  const bool dense = (a.ostep == 1 && a.oy0 == 0 && a.ox0 == 0 && a.Hs == a.Hout && a.Ws == a.Wout);
  const float alpha = a.alpha ? *a.alpha : 1.f;
  float* out_b = a.out + (long long)b * a.Hout * a.Wout * a.out_ld;
  const float* add_b = a.add ? a.add + (long long)b * a.Hout * a.Wout * a.add_ld : nullptr;
  const float* om_b = a.omask ? a.omask + (long long)(b % a.omask_bmod) * a.omask_ld : nullptr;
  const long long slice = (long long)a.B * Ms * a.Cout;
  const int cg = a.Cout / kGroups;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int n = (int)(i % C4) * 4;
    const int m = (int)(i / C4);
    const float* p = a.splitk_ws + ((long long)b * Ms + m) * a.Cout + n;
    f32x4 v = *reinterpret_cast<const f32x4*>(p);
    // (eight slices' loads in flight; the sum keeps its slice order)
    int k = 1;
    for (; k + 8 <= a.ksplit; k += 8) {
      f32x4 u[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) u[j] = *reinterpret_cast<const f32x4*>(p + (k + j) * slice);
#pragma unroll
      for (int j = 0; j < 8; ++j) v += u[j];
    }
    for (; k < a.ksplit; ++k) v += *reinterpret_cast<const f32x4*>(p + k * slice);
    if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + n);
    if (a.stats) {
      if (tn >= 0 && tn != n) {          // (only when the grid stride is not a multiple of Cout/4: flush and restart)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          atomicAdd(&s_g[(tn + k) / cg][0], (double)t1[k]);
          atomicAdd(&s_g[(tn + k) / cg][1], (double)t2[k]);
          t1[k] = t2[k] = 0.f;
        }
      }
      tn = n;
#pragma unroll
      for (int k = 0; k < 4; ++k) { t1[k] += v[k]; t2[k] += v[k] * v[k]; }
    }
    const int yy = m / a.Ws, xx = m - yy * a.Ws;
    const int ox = a.ox0 + xx * a.ostep;
    const long long pix = dense ? (long long)m : (long long)(a.oy0 + yy * a.ostep) * a.Wout + ox;
    if (a.out2) *reinterpret_cast<f32x4*>(a.out2 + ((long long)b * a.Hout * a.Wout + pix) * a.out2_ld + n) = v;
    v *= alpha;
    if (add_b) v += *reinterpret_cast<const f32x4*>(add_b + pix * a.add_ld + n);
    if (om_b) v *= om_b[ox * a.omask_step];
    *reinterpret_cast<f32x4*>(out_b + pix * a.out_ld + n) = v;
  }
  if (a.stats) {
    if (tn >= 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        atomicAdd(&s_g[(tn + k) / cg][0], (double)t1[k]);
        atomicAdd(&s_g[(tn + k) / cg][1], (double)t2[k]);
      }
    }
    __syncthreads();
    if (threadIdx.x < kGroups * 2)
      stat_add(a.stats, b, threadIdx.x >> 1, threadIdx.x & 1, blockIdx.x, s_g[threadIdx.x >> 1][threadIdx.x & 1]);
  }
}

static size_t lds_bytes(int bk, int tm, int nstg = 2, int tn = TN) { return (size_t)nstg * (tm + tn) * bk * sizeof(float); }

template <int BK, int WM, bool WINO, bool F16 = false, int NWM = 2, int NSTG = 2, bool ASPLIT = false>
static hipError_t set_attr() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<BK, WM, WINO, F16, NWM, NSTG, ASPLIT>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(BK, NWM * WM, NSTG));
}

hipError_t conv_igemm_init() {
  hipError_t e;
  if ((e = set_attr<32, 64, false>()) != hipSuccess) return e;
  if ((e = set_attr<32, 32, false>()) != hipSuccess) return e;
  if ((e = set_attr<16, 64, false>()) != hipSuccess) return e;
  if ((e = set_attr<16, 32, false>()) != hipSuccess) return e;
  if ((e = set_attr<32, 32, true>()) != hipSuccess) return e;
  if ((e = set_attr<32, 64, false, true>()) != hipSuccess) return e;
  if ((e = set_attr<32, 64, false, true, 4, 3>()) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<32, 64, false, true, 4, 3, false, 1>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(32, 256, 3, 64))) != hipSuccess) return e;
  if ((e = set_attr<32, 64, false, true, 4, 3, true>()) != hipSuccess) return e;
  if ((e = set_attr<32, 64, false, true, 2, 2, true>()) != hipSuccess) return e;
  if ((e = set_attr<32, 32, false, true, 2, 2, true>()) != hipSuccess) return e;
  if ((e = set_attr<32, 32, false, true>()) != hipSuccess) return e;
  if ((e = set_attr<32, 32, true, true>()) != hipSuccess) return e;
  if ((e = set_attr<32, 32, true, true, 2, 3>()) != hipSuccess) return e;
  return set_attr<16, 32, true>();
}

static int g_tm64_threshold = -1;
static int g_splitk = -1;
static int g_f16_tm = -1;      // US_F16_TM: rows per workgroup of the f16x3 GEMMs (128 or 64)

hipError_t launch_conv_igemm(const ConvArgs& a_in, hipStream_t s) {
  ConvArgs a = a_in;
  a.range_flag = current_range_flag();
  if (a.B <= 0 || a.Hs <= 0 || a.Ws <= 0) return hipSuccess;
  if (!a.in || !a.wt || !a.out) return hipErrorInvalidValue;
  if ((a.bk != 16 && a.bk != 32) || a.Cin % a.bk != 0 || a.ntaps < 1 || a.ntaps > kMaxTaps) return hipErrorInvalidValue;
  if (a.in_ld % 4 != 0) return hipErrorInvalidValue;   // 16-byte DMA pieces of the activation rows
  if (a.Cout % 4 != 0 || a.out_ld % 4 != 0 || (a.add && a.add_ld % 4 != 0)) return hipErrorInvalidValue;   // 16-byte epilogue
  // 32-bit buffer offsets: one item's activation tensor and the packed weights must stay below 2 GiB
  if ((long long)a.Hin * a.Win * a.in_ld * 4 >= (1LL << 31) || (long long)kMaxTaps * a.Cout * a.Cin * 4 >= (1LL << 31))
    return hipErrorInvalidValue;
  if (a.stats) {
    int cg = a.Cout / kGroups;
    if (a.Cout % kGroups != 0 || (cg & (cg - 1)) != 0) return hipErrorInvalidValue;
  }
  if (a.out2 && (a.out2_ld % 4 != 0 || a.wino_out || a.out_split || a.attn_part_ctx)) return hipErrorInvalidValue;
  if (g_f16_tm < 0) {
    const char* e = getenv("US_F16_TM");
    g_f16_tm = e ? atoi(e) : 0;
  }
  if (g_tm64_threshold < 0) {
    const char* e = getenv("US_TM64_THRESHOLD");
    // measured on MI355X (tools/conv_bench, bench.py): three co-resident 64-row workgroups per CU (136 VGPRs, 48 KB LDS)
    // hide the per-chunk barrier better than two 128-row ones at every U-Net shape and batch measured (B'=3 and 24)
    g_tm64_threshold = e ? atoi(e) : (1 << 30);
  }
  const int Ms = a.Hs * a.Ws;
  int tn = TN;                        // 64: the half-width tile of the Winograd-domain GEMMs (conv_igemm_kernel<.., NB = 1>), chosen below
  int nt = (a.Cout + TN - 1) / TN;
  int tm = a.tm;
  if (a.f16) {
    // f16x3: the weights are pre-split; f16 = 1: so is the A operand (Winograd-domain GEMMs, planes written by the input
    // transforms); f16 = 2: A is a plain fp32 activation tensor, split in the kernel (any tap geometry)
    if (a.bk != 32) return hipErrorInvalidValue;
    if (a.f16 == 1 && !a.direct_presplit && a.ntaps == 1 && a.istride == 1 && !a.bias && !a.add && !a.splitk_raw) a.splitk_ws = nullptr;    // Winograd-domain GEMMs
    if (tm == 0 && g_f16_tm > 0) tm = g_f16_tm;
    if (tm == 0) {
      if (a.f16 == 1 && (a.ntaps > 1 || a.direct_presplit)) {
        // a direct convolution whose producer wrote the two-plane form: no split work in the kernel, so the 128-row tile (wave tile
        // 64 x 64, two workgroups per CU) wins over three 64-row ones: +1.4 % end to end with four such convolutions, 256 rows +1.0 %
        // (US_TM_PRESPLIT overrides).  A function of the item's geometry only, like the rule below.
        static int tm_pre = -1;
        if (tm_pre < 0) { const char* e = getenv("US_TM_PRESPLIT"); tm_pre = e ? atoi(e) : 0; }
        tm = tm_pre > 0 ? tm_pre : ((long long)a.Hs * a.Ws >= 512 ? 128 : 64);
        static int tm_pre1 = -1;                     // 1x1 with a pre-split input (the folded to_out, K = 128): US_TM_PRESPLIT_1X1
        if (tm_pre1 < 0) { const char* e = getenv("US_TM_PRESPLIT_1X1"); tm_pre1 = e ? atoi(e) : 64; }
        if (a.ntaps == 1 && a.nphase <= 1) tm = tm_pre1;
      } else if (a.f16 == 1) {
        // Winograd-domain GEMMs (all items of a frequency in one M range).  Round 3's rule (256 rows whenever that leaves 384 workgroups)
        // ignored wave quantisation: a B = 1 level-3 launch of 384 such workgroups is 1.5 per CU, i.e. two rounds for 1.5 rounds of work.
        // Model: the busiest CU runs ceil(workgroups / 256) of them, each costing its rows; the 128-row form (two co-resident workgroups
        // per CU) wins when that product is smaller (US_TM_MODEL=0: the old rule).  The tile never changes a result: every output element
        // sums its K chunks in the same order in both forms.
        static int tm_model = -1;
        if (tm_model < 0) { const char* e = getenv("US_TM_MODEL"); tm_model = e ? atoi(e) : 15; }
        const long long per256 = (long long)((a.Hs * a.Ws + 255) / 256) * nt * a.B;
        const long long per128 = (long long)((a.Hs * a.Ws + 127) / 128) * nt * a.B;
        const long long cost256 = ((per256 + 255) / 256) * 256, cost128 = ((per128 + 255) / 256) * 128;
        // (tm_model = percent by which the 128-row form must undercut the 256-row one: the two-buffer four-wave kernel runs its rows
        // slower than the three-buffer eight-wave one)
        if (tm_model) tm = cost128 * (100 + tm_model) < cost256 * 100 ? 128 : 256;
        else tm = per256 >= 384 ? 256 : 128;
        // ... and 256 rows x 64 columns: half the work per workgroup at the eight-wave kernel's rate per row (each wave keeps one 32-column
        // block: 12 instead of 24 MFMAs per chunk against the same A fragments).  B = 1 level-3 launch, 768 of them: three rounds of half
        // work instead of two of full -- measured 94.5 -> 89.5 us, a third of what the model promises: the half-width kernel runs its work ~15 %
        // slower.  US_TN64=0 switches it off; the margin (percent) covers the thinner MFMA : LDS-read ratio.
        // short K (a level-1 / level-2 frequency of the 4-wide forms is 4 ... 16 chunks): the eight-wave kernel is one workgroup per CU and cannot
        // hide a workgroup's prologue and epilogue behind another's loop; the 128-row kernel has two per CU.  K <= 512: B = 1 +0.6 %, B = 8
        // +1.3 % (3,792 -> 3,841 frames/s); 64-row tiles no better.  US_WINO_TM_SHORTK = rows where K <= US_WINO_SHORTK (0: the model above).
        static int tm_short = -1, k_short = -1;
        if (tm_short < 0) { const char* e = getenv("US_WINO_TM_SHORTK"); tm_short = e ? atoi(e) : 128; }
        if (k_short < 0) { const char* e = getenv("US_WINO_SHORTK"); k_short = e ? atoi(e) : 512; }
        if (tm_short > 0 && a.Cin <= k_short) tm = tm_short;
        static int tn64 = -1;
        if (tn64 < 0) { const char* e = getenv("US_TN64"); tn64 = e ? atoi(e) : 25; }
        if (tn64 > 0 && a.Cout % 64 == 0 && a.Cin >= 1024 && a.Hs * a.Ws > 128) {      // (K = 256 ... 512: twice the prologues and epilogues per unit of work cost more than the rounds saved: measured 34 -> 38 us, 102 -> 120 us)
          const long long per64 = (long long)((a.Hs * a.Ws + 255) / 256) * ((a.Cout + 63) / 64) * a.B;
          const long long cost64 = ((per64 + 255) / 256) * 128;
          // against the 256 x 128 form (the 128-row one measured no better than it wherever the model preferred it)
          if (cost64 * (100 + tn64) < cost256 * 100) { tm = 256; tn = 64; nt = (a.Cout + 63) / 64; }
        }
      } else {
        // direct convolutions (A split in the kernel): three co-resident 64-row workgroups per CU beat the larger tiles at every
        // U-Net shape (tools/conv_bench: 272 vs 251 vs 210 TFLOP/s on the level-0 3x3).  Never a function of the batch, so that
        // the split-K slicing below, hence the summation order, does not depend on what an utterance is batched with.
        tm = 64;
        static int tm_1x1 = -1, tm_taps = -1;        // experiment knobs: US_TM_1X1 / US_TM_TAPS = 64 | 128 | 256
        if (tm_1x1 < 0) { const char* e = getenv("US_TM_1X1"); tm_1x1 = e ? atoi(e) : 0; }
        if (tm_taps < 0) { const char* e = getenv("US_TM_TAPS"); tm_taps = e ? atoi(e) : 0; }
        if (a.ntaps == 1 && a.nphase <= 1 && tm_1x1 > 0 && (long long)a.Hs * a.Ws >= 4 * tm_1x1) tm = tm_1x1;
        if ((a.ntaps > 1 || a.nphase > 1) && tm_taps > 0 && (long long)a.Hs * a.Ws >= 4 * tm_taps) tm = tm_taps;
      }
    }
  }
  if (a.wino_out) {
    // fused Winograd output transform: ntaps = 1, ostep = 2, Hs x Ws = tile grid, Hout x Wout = image, no split-K
    if (a.ntaps != 1 || a.ostep != 2 || a.istride != 1 || a.alpha) return hipErrorInvalidValue;
    if (2 * a.Hs < a.Hout || 2 * a.Ws < a.Wout) return hipErrorInvalidValue;
    tm = 64;
    a.splitk_ws = nullptr;
  }
  if (a.out_split) {
    if (a.Cout % 8 != 0 || a.out_ld % 8 != 0 || (reinterpret_cast<uintptr_t>(a.out) & 31) != 0) return hipErrorInvalidValue;
    a.splitk_ws = nullptr;        // (the slab finish writes fp32)
  }
  if (a.attn_part_ctx) {
    // to_qkv with the attention reduction in the epilogue: three 128-column tiles, 64-row tiles (one chunk of partials each), one pass
    if (!((a.Cout == 3 * kHidden && a.attn_q_cols == kHidden) || (a.Cout == 2 * kHidden && a.attn_q_cols == 0)) || a.ntaps != 1 || a.istride != 1 || a.ostep != 1 || a.wino_out || a.nphase > 1 || a.bias || a.add || a.alpha ||
        a.stats || a.f16 == 1 || !a.attn_part_m || !a.attn_part_s || (a.attn_rows != 0 && a.attn_rows != 64 && a.attn_rows != 128) ||
        a.attn_nchunks != (a.Hs * a.Ws + (a.attn_rows ? a.attn_rows : 64) - 1) / (a.attn_rows ? a.attn_rows : 64))
      return hipErrorInvalidValue;
    tm = a.attn_rows ? a.attn_rows : 64;
    a.splitk_ws = nullptr;
  }
  if (a.nphase > 1) {
    if (a.nphase != 4 || a.ostep != 2 || a.wino_out) return hipErrorInvalidValue;
    a.splitk_ws = nullptr;            // the merged grid is 4x larger already; slabs are per (item, single phase)
  }
  if (tm == 0) {
    const long long wgs128 = (long long)((Ms + 127) / 128) * nt * a.B;
    tm = wgs128 < g_tm64_threshold ? 64 : 128;
  }
  // split-K for launches that cannot fill the chip (small images / short utterances / fine-tune crops): slice the
  // taps*Cin/BK chunks over `ksplit` workgroups per tile, partial slabs in `splitk_ws`, summed by splitk_finish_kernel
  const int mt = (Ms + tm - 1) / tm;
  const long long tiles = (long long)mt * nt * a.B;
  const int S_all = a.ntaps * (a.Cin / a.bk);
  a.ksplit = 1;
  if (g_splitk < 0) {
    const char* e = getenv("US_SPLITK");
    g_splitk = e ? atoi(e) : 1;
  }
  // The slice count depends on the per-item tile count only (never on the batch), so an utterance's result does not depend
  // on what it is batched or sharded with (tests: ...shard_independence).
  const long long tiles_item = a.splitk_by_batch ? tiles : (long long)mt * nt;
  if (g_splitk && a.splitk_ws && tiles_item < 128 && S_all >= 8 && a.Cout % 4 == 0 && a.out_ld % 4 == 0 && (!a.add || a.add_ld % 4 == 0)) {
    long long k = (256 + tiles_item - 1) / tiles_item;      // aim at >= 256 workgroups per item
    if (k > S_all / 4) k = S_all / 4;                        // at least 4 chunks per slice
    if (k > 512 / tiles_item) k = 512 / tiles_item;          // bounds the slab: k * Ms * Cout <= 512 tiles = 4 Mi floats per item
    if (k > 32) k = 32;
    if (k >= 2 && k * (long long)a.B * Ms * a.Cout <= a.splitk_ws_floats) a.ksplit = (int)k;
  }
  if (a.splitk_raw) {
    // Winograd-domain GEMMs of one fine-tune crop: 16 frequencies x (1 row tile x 8 column tiles) = 128 workgroups walking 32 chunks each
    // (33 us, 26 such launches per iteration).  Slice K so that the launch is ~3 workgroups per CU; the output transform sums the slabs.
    a.ksplit = 1;
    if (g_splitk && a.splitk_ws && !a.wino_out && !a.out_split && !a.stats && !a.add && !a.bias && tiles < 384 && S_all >= 16) {
      long long k = (768 + tiles - 1) / tiles;
      if (k > S_all / 4) k = S_all / 4;
      if (k > 8) k = 8;
      while (k >= 2 && k * (long long)a.B * Ms * a.Cout > a.splitk_ws_floats) --k;
      if (k >= 2) a.ksplit = (int)k;
    }
    if (a.ksplit_out) *a.ksplit_out = a.ksplit;
  }
  a.nt = nt;
  dim3 grid(mt * a.ksplit * nt, 1, a.B * (a.nphase > 1 ? a.nphase : 1));
  const size_t lds = lds_bytes(a.bk, tm);
  if (a.f16 == 2) {
    if (tm == 256) hipLaunchKernelGGL((conv_igemm_kernel<32, 64, false, true, 4, 3, true>), grid, dim3(512), lds_bytes(32, 256, 3), s, a);
    else if (tm == 128) hipLaunchKernelGGL((conv_igemm_kernel<32, 64, false, true, 2, 2, true>), grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL((conv_igemm_kernel<32, 32, false, true, 2, 2, true>), grid, dim3(256), lds, s, a);
  } else if (a.f16) {
    if (a.wino_out && !(a.debug & 128)) hipLaunchKernelGGL((conv_igemm_kernel<32, 32, true, true, 2, 3>), grid, dim3(256), lds_bytes(32, 64, 3), s, a);
    else if (a.wino_out) hipLaunchKernelGGL((conv_igemm_kernel<32, 32, true, true>), grid, dim3(256), lds, s, a);
    else if (tm == 256 && tn == 64) hipLaunchKernelGGL((conv_igemm_kernel<32, 64, false, true, 4, 3, false, 1>), grid, dim3(512), lds_bytes(32, 256, 3, 64), s, a);
    else if (tm == 256) hipLaunchKernelGGL((conv_igemm_kernel<32, 64, false, true, 4, 3>), grid, dim3(512), lds_bytes(32, 256, 3), s, a);
    else if (tm == 128) hipLaunchKernelGGL((conv_igemm_kernel<32, 64, false, true>), grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL((conv_igemm_kernel<32, 32, false, true>), grid, dim3(256), lds, s, a);
  } else if (a.wino_out) {
    if (a.bk == 32)
      hipLaunchKernelGGL((conv_igemm_kernel<32, 32, true>), grid, dim3(256), lds, s, a);
    else
      hipLaunchKernelGGL((conv_igemm_kernel<16, 32, true>), grid, dim3(256), lds, s, a);
  } else if (a.bk == 32 && tm == 128)
    hipLaunchKernelGGL((conv_igemm_kernel<32, 64, false>), grid, dim3(256), lds, s, a);
  else if (a.bk == 32)
    hipLaunchKernelGGL((conv_igemm_kernel<32, 32, false>), grid, dim3(256), lds, s, a);
  else if (tm == 128)
    hipLaunchKernelGGL((conv_igemm_kernel<16, 64, false>), grid, dim3(256), lds, s, a);
  else
    hipLaunchKernelGGL((conv_igemm_kernel<16, 32, false>), grid, dim3(256), lds, s, a);
  if (a.ksplit > 1 && !a.splitk_raw) {
    long long total = (long long)Ms * (a.Cout / 4);
    int blocks = (int)((total + 1023) / 1024);
    if (blocks < 1) blocks = 1;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(splitk_finish_kernel, dim3(blocks, a.B), dim3(256), 0, s, a);
  }
  return hipGetLastError();
}

// ---- weight repack --------------------------------------------------------------------------------
__global__ void pack_conv_weight_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cout, int Cin, int KH, int KW,
                                        int oihw, int bk, int qkv_rows) {
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
    if (qkv_rows) co = qkv_src_row_dev(co);
    long long si = oihw ? (((long long)co * Cin + ci) * KH + ky) * KW + kx
                        : (((long long)ci * Cout + co) * KH + ky) * KW + kx;
    dst[i] = src[si];
  }
}

// f16x3 form (bk = 32): body in pack_f16.h (shared with the table-driven launch of wino.hip)
__global__ __launch_bounds__(256) void pack_conv_weight_f16_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, int Cout, int Cin,
                                                                   int KH, int KW, int oihw, unsigned* range_flag, int qkv_rows) {
  bool over = false;
  conv_pack_f16_body(src, dst, Cout, Cin, KH, KW, oihw, qkv_rows, blockIdx.x, gridDim.x, over);
  range_report(range_flag, over, kRangeWeight);
}

hipError_t launch_pack_conv_weight_f16(const float* src, float* dst, int Cout, int Cin, int KH, int KW, bool oihw, hipStream_t s,
                                       bool qkv_rows) {
  if (Cin % 32 != 0 || (qkv_rows && Cout != 3 * kHidden)) return hipErrorInvalidValue;
  long long total = (long long)KH * KW * Cout * (Cin / 8);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(pack_conv_weight_f16_kernel, dim3(blocks), dim3(256), 0, s, src, reinterpret_cast<_Float16*>(dst), Cout, Cin, KH, KW,
                     oihw ? 1 : 0, current_range_flag(), qkv_rows ? 1 : 0);
  return hipGetLastError();
}

hipError_t launch_pack_conv_weight(const float* src, float* dst, int Cout, int Cin, int KH, int KW, bool oihw, int bk,
                                   hipStream_t s, bool qkv_rows) {
  if (Cin % bk != 0 || (qkv_rows && Cout != 3 * kHidden)) return hipErrorInvalidValue;
  long long total = (long long)KH * KW * Cout * Cin;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(blocks), dim3(256), 0, s, src, dst, Cout, Cin, KH, KW, oihw ? 1 : 0, bk, qkv_rows ? 1 : 0);
  return hipGetLastError();
}

}  // namespace us
