// Micro-benchmark of the implicit-GEMM convolution kernel on synthetic shapes (build: see tools/conv_bench.sh).
// usage: conv_bench [debug_mask]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <array>
#include <algorithm>
#include "../unitspeech_amd/csrc/kernels.h"
using namespace us;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// Calibration: a pure v_mfma_f32_32x32x2_f32 stream (no memory traffic), `waves` waves per SIMD on every CU; reports the
// TFLOP/s the device sustains and the shader clock (s_memtime ticks per s_memrealtime 100 MHz tick).
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void mfma_peak_kernel(float* out, int iters, unsigned long long* clk) {
  f32x16 a0, a1, a2, a3;
  for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; a2[r] = 0.f; a3[r] = 0.f; }
  float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f + 0.5f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

// in-place fp32 -> f16x3 operand format (two interleaved fp16 planes per group of 8 values), for timing the F16 instantiations
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
__global__ void to_f16x2_kernel(float* x, size_t n8) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    float v[8];
    for (int k = 0; k < 8; ++k) v[k] = x[i * 8 + k];
    _Float16* h = reinterpret_cast<_Float16*>(x + i * 8);
    for (int k = 0; k < 8; ++k) { _Float16 hi = (_Float16)v[k]; h[k] = hi; h[8 + k] = (_Float16)((v[k] - (float)hi) * 2048.f); }
  }
}

// pure v_mfma_f32_32x32x16_f16 stream on random-ish operands: the ceiling of the f16x3 GEMMs (x 1/3 in fp32-equivalent FLOPs)
typedef _Float16 half8v __attribute__((ext_vector_type(8)));
__global__ void mfma_f16_peak_kernel(float* out, int iters, unsigned long long* clk) {
  f32x16 a0, a1, a2, a3;
  for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; a2[r] = 0.f; a3[r] = 0.f; }
  half8v x, y;
  for (int k = 0; k < 8; ++k) { x[k] = (_Float16)(0.37f * ((threadIdx.x * 7 + k * 13) % 29) - 5.f); y[k] = (_Float16)(0.11f * ((threadIdx.x * 3 + k * 5 + blockIdx.x) % 31) - 1.7f); }
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, x, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, y, a3, 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

typedef float f32x4c __attribute__((ext_vector_type(4)));
__global__ void mfma_f16_16x16_peak_kernel(float* out, int iters, unsigned long long* clk) {
  f32x4c a[8];
  for (int i = 0; i < 8; ++i) a[i] = f32x4c{0.f, 0.f, 0.f, 0.f};
  half8v x, y;
  for (int k = 0; k < 8; ++k) { x[k] = (_Float16)(0.37f * ((threadIdx.x * 7 + k * 13) % 29) - 5.f); y[k] = (_Float16)(0.11f * ((threadIdx.x * 3 + k * 5 + blockIdx.x) % 31) - 1.7f); }
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16((j & 1) ? x : y, (j & 2) ? x : y, a[j], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int j = 0; j < 8; ++j) s += a[j][0] + a[j][1] + a[j][2] + a[j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

// Structural calibration: the conv kernel's inner-loop skeleton (TM=64 variant: 2 accumulators, 32 MFMAs per step) with
// its ingredients switched on one at a time.  FLAGS: 1 = barrier per step, 2 = 12 ds_read_b128 per step, 4 = 6 LDS-DMA
// pieces per step (double-buffered like the real kernel).
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int FLAGS, int PIECES = 6, int BUF = 0>
__global__ __launch_bounds__(256, 2) void skeleton_kernel(const float* __restrict__ g, float* out, int steps) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x16 a0, a1;
  for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
  f32x4v fa = {1.f, 2.f, 3.f, 4.f}, fb0 = {1.f, 1.f, 2.f, 2.f}, fb1 = {3.f, 1.f, 2.f, 1.f};
  const float* src = g + ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  for (int s = 0; s < steps; ++s) {
    float* buf = lds + ((s + 1) & 1) * 6144;
    if (FLAGS & 4) {
      if (BUF) {
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, 0x7fffffff, 0x00020000);
        const int voff = (blockIdx.x * 256 + threadIdx.x) * 16;
#pragma unroll
        for (int j = 0; j < PIECES; ++j)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(buf + (wave * 6 + j) * 256), 16, voff,
                                                   ((s * PIECES + j) & 63) * 262144, 0, 0);
      } else {
#pragma unroll
        for (int j = 0; j < PIECES; ++j)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)((s * PIECES + j) & 63) * 65536),
                                           (__attribute__((address_space(3))) void*)(buf + (wave * 6 + j) * 256), 16, 0, 0);
      }
    }
    const float* rd = lds + (s & 1) * 6144 + lane * 4;
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      if (FLAGS & 2) {
        fa = *reinterpret_cast<const f32x4v*>(rd + sub * 256);
        fb0 = *reinterpret_cast<const f32x4v*>(rd + 1024 + sub * 256);
        fb1 = *reinterpret_cast<const f32x4v*>(rd + 2048 + sub * 256);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j], fb0[j], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j], fb1[j], a1, 0, 0, 0);
      }
    }
    if (FLAGS & 1) __syncthreads();
  }
  float sum = 0.f;
  for (int r = 0; r < 16; ++r) sum += a0[r] + a1[r];
  out[blockIdx.x * 256 + threadIdx.x] = sum;
}

template <int FLAGS, int PIECES = 6, int BUF = 0>
static void run_skeleton(const float* g, float* out) {
  const int steps = 288, blocks = 256 * 3 * 2;     // 3 resident workgroups per CU, 2 rounds
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&skeleton_kernel<FLAGS, PIECES, BUF>), hipFuncAttributeMaxDynamicSharedMemorySize, 49152);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((skeleton_kernel<FLAGS, PIECES, BUF>), dim3(blocks), dim3(256), 49152, 0, g, out, steps);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((skeleton_kernel<FLAGS, PIECES, BUF>), dim3(blocks), dim3(256), 49152, 0, g, out, steps);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  double fl = (double)blocks * 4 * steps * 32 * 4096.0;
  printf("skeleton flags=%d (1=barrier 2=ds_read 4=lds-dma) pieces/step=%d %s: %7.1f us  %6.1f TFLOP/s\n", FLAGS, PIECES,
         BUF ? "buffer_load..lds" : "global_load_lds", ms * 1e3, fl / ms / 1e9);
}

static void calibrate() {
  float* out; unsigned long long* clk;
  CK(hipMalloc(&out, 256 * 8 * 256 * 4)); CK(hipMalloc(&clk, 16));
  {
    float* g; CK(hipMalloc(&g, (size_t)64 * 65536 * 4 + 1536 * 256 * 16)); CK(hipMemset(g, 0, (size_t)64 * 65536 * 4 + 1536 * 256 * 16));
    run_skeleton<0>(g, out); run_skeleton<1>(g, out); run_skeleton<2>(g, out); run_skeleton<3>(g, out);
    run_skeleton<7>(g, out); run_skeleton<7, 4>(g, out); run_skeleton<7, 3>(g, out); run_skeleton<7, 2>(g, out);
    run_skeleton<7, 6, 1>(g, out); run_skeleton<7, 4, 1>(g, out); run_skeleton<7, 3, 1>(g, out); run_skeleton<7, 2, 1>(g, out);
    CK(hipFree(g));
  }
  for (int wps : {1, 2}) {
    int blocks = 256 * wps, iters = 40000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(mfma_f16_peak_kernel, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(mfma_f16_peak_kernel, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    double fl = (double)blocks * 4 * iters * 4 * 32768.0;
    printf("calibration: pure MFMA f16 32x32x16 stream, %d wave(s)/SIMD: %.1f TFLOP/s (= %.1f fp32-equivalent as f16x3), shader clock %.3f GHz\n",
           wps, fl / ms / 1e9, fl / ms / 1e9 / 3, (double)h[0] / (double)h[1] * 0.1);
  }
  for (int wps : {1, 2}) {
    int blocks = 256 * wps, iters = 20000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(mfma_f16_16x16_peak_kernel, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(mfma_f16_16x16_peak_kernel, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    double fl = (double)blocks * 4 * iters * 8 * 16384.0;
    printf("calibration: pure MFMA f16 16x16x32 stream, %d wave(s)/SIMD: %.1f TFLOP/s (= %.1f fp32-equivalent as f16x3), shader clock %.3f GHz\n",
           wps, fl / ms / 1e9, fl / ms / 1e9 / 3, (double)h[0] / (double)h[1] * 0.1);
  }
  for (int wps : {1, 2, 4}) {
    int blocks = 256 * wps;   // 256-thread blocks: 4 waves = 1 per SIMD
    int iters = 20000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    double fl = (double)blocks * 4 * iters * 4 * 4096.0;
    printf("calibration: pure MFMA f32 stream, %d wave(s)/SIMD: %.1f TFLOP/s, shader clock %.3f GHz\n", wps, fl / ms / 1e9,
           (double)h[0] / (double)h[1] * 0.1);
  }
}

struct Shape { const char* name; int B, H, W, Cin, Cout, taps; int wino = 0; };   // wino: B = 16 frequencies x 3 items, one weight matrix per frequency

int main(int argc, char** argv) {
  int debug_arg = argc > 1 ? atoi(argv[1]) : 0;
  const int debug = debug_arg;
  CK(conv_igemm_init());
  if (debug == 0) calibrate();
  Shape shapes[] = {{"L0 3x3 128->128", 3, 80, 1024, 128, 128, 9}, {"L1 3x3 256->256", 3, 40, 512, 256, 256, 9},
                    {"L2 3x3 512->512", 3, 20, 256, 512, 512, 9}, {"L3 3x3 1024->1024", 3, 10, 128, 1024, 1024, 9},
                    {"L3 3x3 2048->512", 3, 10, 128, 2048, 512, 9}, {"L0 1x1 128->384", 3, 80, 1024, 128, 384, 1},
                    {"L0 1x1 128->128", 3, 80, 1024, 128, 128, 1},
                    {"K1152 small-spatial", 48, 20, 256, 128, 128, 9}, {"K1152 W=128 rows", 3, 640, 128, 128, 128, 9},
                    {"K2304 L0-spatial", 3, 80, 1024, 256, 128, 9}, {"1x1 K=1024 ->128", 3, 80, 1024, 1024, 128, 1},
                    {"L3 1x1 1024->384", 3, 10, 128, 1024, 384, 1}, {"L3 1x1 128->1024", 3, 10, 128, 128, 1024, 1},
                    {"L3 1x1 512->1024", 3, 10, 128, 512, 1024, 1}, {"L2 1x1 512->384", 3, 20, 256, 512, 384, 1},
                    {"W1 gemm 256->256", 48, 20, 256, 256, 256, 1, 1}, {"W1 gemm 512->256", 48, 20, 256, 512, 256, 1, 1},
                    {"W2 gemm 512->512", 48, 10, 128, 512, 512, 1, 1}, {"W3 gemm 1024->1024", 48, 5, 64, 1024, 1024, 1, 1},
                    {"W3 gemm 2048->512", 48, 5, 64, 2048, 512, 1, 1}, {"W3 gemm 512->512", 48, 5, 64, 512, 512, 1, 1},
                    {"W0 gemm 128->128", 48, 40, 512, 128, 128, 1, 1},
                    // the same GEMMs with the 3 items of a frequency folded into M (what the decoder launches)
                    {"G1 gemm 256->256", 16, 60, 256, 256, 256, 1, 1}, {"G1 gemm 512->128", 16, 60, 256, 512, 128, 1, 1},
                    {"G2 gemm 512->512", 16, 30, 128, 512, 512, 1, 1}, {"G2 gemm 1024->256", 16, 30, 128, 1024, 256, 1, 1},
                    {"G3 gemm 1024->1024", 16, 15, 64, 1024, 1024, 1, 1}, {"G3 gemm 2048->512", 16, 15, 64, 2048, 512, 1, 1},
                    {"G3 gemm 512->512", 16, 15, 64, 512, 512, 1, 1},
                    {"H3 gemm 1024->1024", 24, 15, 32, 1024, 1024, 1, 1}, {"H2 gemm 512->512", 36, 15, 64, 512, 512, 1, 1}, {"G0 gemm 128->128", 16, 120, 512, 128, 128, 1, 1},
                    // wino = 2: output transform fused (B = items, H x W = tile grid, output 2H x 2W)
                    {"F0 fused 128->128", 3, 40, 512, 128, 128, 1, 2}, {"F1 fused 256->256", 3, 20, 256, 256, 256, 1, 2},
                    {"F1 fused 512->256", 3, 20, 256, 512, 256, 1, 2}, {"F2 fused 512->512 B24", 24, 10, 128, 512, 512, 1, 2},
                    {"F3 fused 1024->1024 B24", 24, 5, 64, 1024, 1024, 1, 2}};
  float* zeros; CK(hipMalloc(&zeros, 16384)); CK(hipMemset(zeros, 0, 16384));
  float* skws = nullptr; const long long skfl = 12LL << 20;
  if (getenv("CB_SPLITK")) CK(hipMalloc(&skws, skfl * 4));
  for (auto& sh : shapes) {
    if (getenv("CB_ONLY") && !strstr(sh.name, getenv("CB_ONLY"))) continue;
    size_t n_in = (size_t)sh.B * sh.H * sh.W * sh.Cin * (sh.wino == 2 ? 16 : 1), n_out = (size_t)sh.B * sh.H * sh.W * sh.Cout * (sh.wino == 2 ? 4 : 1);
    size_t n_w = (size_t)sh.taps * sh.Cin * sh.Cout * (sh.wino == 1 ? sh.B / std::max(sh.B / 16, 1) : (sh.wino ? 16 : 1));
    float *in, *out, *w, *bias;
    CK(hipMalloc(&in, n_in * 4)); CK(hipMalloc(&out, n_out * 4)); CK(hipMalloc(&w, n_w * 4)); CK(hipMalloc(&bias, sh.Cout * 4));
    CK(launch_fill_normal(in, n_in, 1, 1, 0)); CK(launch_fill_normal(w, n_w, 1, 2, 0)); CK(launch_fill_normal(bias, sh.Cout, 1, 3, 0));
    const bool f16 = getenv("CB_F16") != nullptr;
    const bool presplit = getenv("CB_PRESPLIT") != nullptr;      // direct convolutions with a pre-split activation tensor (f16 = 1)
    if (f16) {
      if (sh.wino || presplit) hipLaunchKernelGGL(to_f16x2_kernel, dim3(4096), dim3(256), 0, 0, in, n_in / 8);
      hipLaunchKernelGGL(to_f16x2_kernel, dim3(4096), dim3(256), 0, 0, w, n_w / 8);
    }
    const int n_ab = getenv("CB_AB") ? 2 : 1;           // CB_AB=<bit>: every case also with that debug bit set, back to back
    const int ab_bit = getenv("CB_AB") ? atoi(getenv("CB_AB")) : 0;
    for (int ab = 0; ab < n_ab; ++ab)
    for (int tm : {0, 256, 128, 64}) {
      if (tm == 0 && !(getenv("CB_TM") && atoi(getenv("CB_TM")) == 0)) continue;
      if (getenv("CB_TM") && atoi(getenv("CB_TM")) != tm) continue;
      if (tm == 256 && !(f16 && sh.wino != 2)) continue;
      const int debug = debug_arg | ((ab != 0) != (getenv("CB_AB_FIRST") != nullptr) ? ab_bit : 0);
      ConvArgs a; memset(&a, 0, sizeof a);
      a.in = in; a.in_ld = sh.Cin; a.wt = w; a.bias = bias; a.out = out; a.out_ld = sh.Cout; a.zeros = zeros;
      a.B = sh.B; a.Hin = a.Hout = a.Hs = sh.H; a.Win = a.Wout = a.Ws = sh.W; a.Cin = sh.Cin; a.Cout = sh.Cout;
      a.ostep = 1; a.istride = 1; a.ntaps = sh.taps; a.bk = 32; a.tm = tm; a.omask_bmod = 1; a.debug = debug;
      a.splitk_ws = skws; a.splitk_ws_floats = skfl;
      a.f16 = f16 ? ((sh.wino || presplit) ? 1 : 2) : 0;
      static double* stats = nullptr;
      if (!stats) { CK(hipMalloc(&stats, 1 << 20)); CK(hipMemset(stats, 0, 1 << 20)); }
      if (getenv("CB_STATS") && sh.wino != 1) a.stats = stats;
      if (sh.wino == 1 && !getenv("CB_NO_XCDZ")) a.xcd_z = 1;
      if (sh.wino == 1) { a.wt_bstride = (long long)sh.Cin * sh.Cout; a.wt_bdiv = sh.B / 16; a.splitk_ws = nullptr; }
      if (sh.wino == 2) { a.wt_bstride = (long long)sh.Cin * sh.Cout; a.wino_out = 1; a.ostep = 2; a.Hout = 2 * sh.H; a.Wout = 2 * sh.W; a.splitk_ws = nullptr; }
      if (sh.taps == 9) { for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) a.set_tap(ky * 3 + kx, ky - 1, kx - 1, ky * 3 + kx); }
      else a.set_tap(0, 0, 0, 0);
#if defined(US_STAMP) || defined(US_LIFE)
      static unsigned long long* stamps = nullptr;
      const size_t n_st = (size_t)1 << 22;
      if (!stamps) CK(hipMalloc(&stamps, n_st * 8));
      CK(hipMemset(stamps, 0, n_st * 8));
      a.stamp_out = stamps;
#endif
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      for (int i = 0; i < 3; ++i) CK(launch_conv_igemm(a, 0));
#ifdef US_LIFE
      {
        // one more launch on a clean buffer: per workgroup (wave 0) entry / loop start / loop end / exit on the 100 MHz clock
        CK(hipDeviceSynchronize());
        CK(hipMemset(stamps, 0, n_st * 8));
        CK(launch_conv_igemm(a, 0));
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> hst(n_st);
        CK(hipMemcpy(hst.data(), stamps, n_st * 8, hipMemcpyDeviceToHost));
        std::vector<std::array<double, 4>> wg;
        unsigned long long t0 = ~0ull;
        for (size_t i = 0; i + 3 < n_st; i += 4) if (hst[i + 3]) t0 = std::min(t0, hst[i]);
        for (size_t i = 0; i + 3 < n_st; i += 4) if (hst[i + 3]) wg.push_back({(hst[i] - t0) * 0.01, (hst[i + 1] - t0) * 0.01, (hst[i + 2] - t0) * 0.01, (hst[i + 3] - t0) * 0.01});
        if (!wg.empty()) {
          std::sort(wg.begin(), wg.end());
          double pro = 0, loop = 0, epi = 0, last = 0;
          for (auto& w : wg) { pro += w[1] - w[0]; loop += w[2] - w[1]; epi += w[3] - w[2]; last = std::max(last, w[3]); }
          const double n = (double)wg.size();
          printf("   life over %zu waves (us): prologue %.2f, loop %.2f, epilogue+stores %.2f; last exit at %.1f\n", wg.size(), pro / n, loop / n, epi / n, last);
          printf("   entry times (us), deciles:");
          for (int d = 0; d <= 10; ++d) printf(" %.1f", wg[std::min(wg.size() - 1, wg.size() * d / 10)][0]);
          printf("\n   exit times (us), deciles:");
          std::vector<double> ex; for (auto& w : wg) ex.push_back(w[3]);
          std::sort(ex.begin(), ex.end());
          for (int d = 0; d <= 10; ++d) printf(" %.1f", ex[std::min(ex.size() - 1, ex.size() * d / 10)]);
          printf("\n");
        }
      }
#endif
#ifdef US_STAMP
      {
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> hst(n_st);
        CK(hipMemcpy(hst.data(), stamps, n_st * 8, hipMemcpyDeviceToHost));
        double w = 0, b = 0, bd = 0, st = 0; long long cnt = 0;
        for (size_t i = 0; i + 3 < n_st; i += 4) if (hst[i + 3]) { w += hst[i]; b += hst[i + 1]; bd += hst[i + 2]; st += hst[i + 3]; ++cnt; }
        if (cnt) printf("   stamps over %lld waves: per step %.0f cycles = vmcnt wait %.0f + barrier %.0f + body %.0f\n", cnt, (w + b + bd) / st, w / st, b / st, bd / st);
      }
#endif
      const int reps = 10;
      float ms = 0.f;
      if (getenv("CB_COLD")) {        // flush L2 / Infinity Cache with a 1 GiB fill before every timed launch
        static float* trash = nullptr;
        if (!trash) CK(hipMalloc(&trash, (size_t)1 << 30));
        for (int i = 0; i < reps; ++i) {
          CK(hipMemsetAsync(trash, i, (size_t)1 << 30, 0));
          CK(hipEventRecord(e0, 0));
          CK(launch_conv_igemm(a, 0));
          CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
          float t; CK(hipEventElapsedTime(&t, e0, e1)); ms += t;
        }
        ms /= reps;
      } else {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) CK(launch_conv_igemm(a, 0));
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
      }
      double fl = 2.0 * sh.B * sh.H * sh.W * (double)sh.Cin * sh.Cout * sh.taps * (sh.wino == 2 ? 16 : 1);
      printf("%-20s tm=%3d %s debug=%d  %8.1f us  %6.1f TFLOP/s\n", sh.name, tm, f16 ? "f16x3" : "fp32 ", debug, ms * 1e3, fl / ms / 1e9);
    }
    CK(hipFree(in)); CK(hipFree(out)); CK(hipFree(w)); CK(hipFree(bias));
  }
  return 0;
}
