// Research prototype (not part of the product): fp32-accurate GEMM on the bf16 matrix cores by operand splitting.
//   a = a1 + a2 + a3 (three bf16 planes, 8 significant bits each), same for b; C += a1b1 + (a1b2 + a2b1) + (a2b2 + a1b3 + a3b1)
//   on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: 6 bf16 MFMAs per fp32-equivalent 32x32x16 block = 16/6 = 2.67x the
//   fp32 MFMA rate at (emulated on the CPU) fp32-level error.  This program measures, on the GPU, the rate of a simple
//   128x128-tile kernel for 1 / 3 / 6 products and the error of each against an fp64 host reference.
// build + run:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/bf16x6_gemm.cpp -o /tmp/bf16x6 && /tmp/bf16x6
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x)                                                                                       \
  do {                                                                                              \
    hipError_t e_ = (x);                                                                            \
    if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } \
  } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void split3_kernel(const float* __restrict__ x, __bf16* __restrict__ p1, __bf16* __restrict__ p2, __bf16* __restrict__ p3,
                              long long n) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float v = x[i];
    const __bf16 a1 = (__bf16)v;
    const float r1 = v - (float)a1;
    const __bf16 a2 = (__bf16)r1;
    const float r2 = r1 - (float)a2;
    p1[i] = a1;
    p2[i] = a2;
    p3[i] = (__bf16)r2;
  }
}

constexpr int TM = 128, TN = 128, BK = 32, LDK = BK + 8;   // padded LDS rows (80 B): conflict-free 16-byte fragment reads

// A: [3][M][K] bf16 planes, B: [3][N][K] bf16 planes, C: [M][N] fp32 = A * B^T.  M, N multiples of 128, K of 32.
template <int NPROD>
__global__ __launch_bounds__(256) void gemm_split_kernel(const __bf16* __restrict__ A, const __bf16* __restrict__ B, float* __restrict__ C,
                                                         int M, int N, int K) {
  constexpr int NPL = NPROD == 1 ? 1 : (NPROD == 3 ? 2 : 3);   // planes needed
  __shared__ __attribute__((aligned(16))) __bf16 As[NPL][TM][LDK];
  __shared__ __attribute__((aligned(16))) __bf16 Bs[NPL][TN][LDK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, l32 = lane & 31, hh = lane >> 5;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
  const long long planeA = (long long)M * K, planeB = (long long)N * K;
  // staging role: 16 bytes (8 bf16) of row (tid / 4) [+ 64], k segment tid % 4
  const int srow = tid >> 2, sseg = (tid & 3) * 8;
  bf16x8 ra[NPL][2], rb[NPL][2];
  auto gload = [&](int k0) {
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        ra[p][h] = *reinterpret_cast<const bf16x8*>(A + p * planeA + (long long)(m0 + srow + 64 * h) * K + k0 + sseg);
        rb[p][h] = *reinterpret_cast<const bf16x8*>(B + p * planeB + (long long)(n0 + srow + 64 * h) * K + k0 + sseg);
      }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        *reinterpret_cast<bf16x8*>(&As[p][srow + 64 * h][sseg]) = ra[p][h];
        *reinterpret_cast<bf16x8*>(&Bs[p][srow + 64 * h][sseg]) = rb[p][h];
      }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  gload(0);
  for (int k0 = 0; k0 < K; k0 += BK) {
    lstore();
    __syncthreads();
    if (k0 + BK < K) gload(k0 + BK);          // next chunk's global loads fly under this chunk's MFMAs
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      bf16x8 a[2][NPL], b[2][NPL];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
          a[i][p] = *reinterpret_cast<const bf16x8*>(&As[p][wm * 64 + i * 32 + l32][s * 16 + 8 * hh]);
          b[i][p] = *reinterpret_cast<const bf16x8*>(&Bs[p][wn * 64 + i * 32 + l32][s * 16 + 8 * hh]);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x16 c = acc[i][j];
          if (NPROD == 6) {            // smallest terms first
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);
          }
          if (NPROD >= 3) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
          }
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);
          acc[i][j] = c;
        }
    }
    __syncthreads();
  }
  // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        const int col = n0 + wn * 64 + j * 32 + l32;
        C[(long long)row * N + col] = acc[i][j][r];
      }
}

template <int NPROD>
static void run(const __bf16* dA, const __bf16* dB, float* dC, int M, int N, int K, const std::vector<float>& hA, const std::vector<float>& hB) {
  dim3 grid(N / TN, M / TM);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(gemm_split_kernel<NPROD>, grid, dim3(256), 0, 0, dA, dB, dC, M, N, K);
  CK(hipGetLastError());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 10;
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm_split_kernel<NPROD>, grid, dim3(256), 0, 0, dA, dB, dC, M, N, K);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  std::vector<float> hC((size_t)M * N);
  CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
  // fp64 reference on a sample of entries
  std::mt19937 g(7);
  double err = 0, mag = 0, worst = 0;
  const int samples = 2000;
  for (int sidx = 0; sidx < samples; ++sidx) {
    const int r = g() % M, c = g() % N;
    double ref = 0;
    for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)r * K + k] * (double)hB[(size_t)c * K + k];
    const double d = fabs((double)hC[(size_t)r * N + c] - ref);
    err += d; mag += fabs(ref); if (d > worst) worst = d;
  }
  printf("products=%d  %4dx%4dx%4d  %8.1f us  %7.1f TFLOP/s fp32-equivalent   mean|err|/mean|C| = %.3e  (max %.3e)\n", NPROD, M, N, K,
         ms * 1e3, 2.0 * M * N * K / ms / 1e9, err / mag, worst / (mag / samples));
}

int main() {
  const int M = 4096, N = 4096;
  for (int K : {1024, 4096}) {
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    std::mt19937 g(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (auto& v : hA) v = nd(g);
    for (auto& v : hB) v = nd(g);
    float *dA, *dB, *dC;
    __bf16 *pA, *pB;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMalloc(&pA, hA.size() * 2 * 3)); CK(hipMalloc(&pB, hB.size() * 2 * 3));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(split3_kernel, dim3(2048), dim3(256), 0, 0, dA, pA, pA + hA.size(), pA + 2 * hA.size(), (long long)hA.size());
    hipLaunchKernelGGL(split3_kernel, dim3(2048), dim3(256), 0, 0, dB, pB, pB + hB.size(), pB + 2 * hB.size(), (long long)hB.size());
    CK(hipDeviceSynchronize());
    run<1>(pA, pB, dC, M, N, K, hA, hB);
    run<3>(pA, pB, dC, M, N, K, hA, hB);
    run<6>(pA, pB, dC, M, N, K, hA, hB);
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC)); CK(hipFree(pA)); CK(hipFree(pB));
  }
  return 0;
}
