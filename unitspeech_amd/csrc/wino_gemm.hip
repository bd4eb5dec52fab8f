// Experimental (US_WINO_BF16X6=1): the Winograd-domain GEMMs of the two low-resolution U-Net levels at fp32 accuracy on the bf16
// matrix cores.  Both operands are kept as three bf16 planes (x = x1 + x2 + x3, 8 significant bits each; V planes written by the
// input transforms, U planes at weight load); the kernel forms the six products of order >= 2^-16,
//   C += a3 b1 + a1 b3 + a2 b2 + a2 b1 + a1 b2 + a1 b1      (smallest first)
// on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: 16/6 = 2.67x the fp32 MFMA rate, error at the fp32 level (4e-7 of mean|C| at
// K = 1,024 on normal data against 3e-7 for an fp32 sgemm; tools/bf16x6_gemm.cpp).
//
// Per frequency f:  M_f [rows][N] = V_f [rows][K] . U_f [N][K]^T,  rows = B * tiles (the items of a frequency are contiguous).
// Workgroup tile 128 x 128, 4 waves of 64 x 64, K in chunks of 32: operands go global -> registers -> LDS (padded 80-byte rows:
// conflict-free 16-byte fragment reads), the next chunk's global loads are issued before the current chunk's MFMAs.
#include <cstdlib>
#include "kernels.h"

namespace us {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
constexpr int GT = 128, GBK = 32, GLDK = GBK + 8;
}

__global__ __launch_bounds__(256) void wino_gemm_bf16x6_kernel(const __bf16* __restrict__ V, const __bf16* __restrict__ U, float* __restrict__ M,
                                                               int rows, int N, int K) {
  __shared__ __attribute__((aligned(16))) __bf16 As[3][GT][GLDK];
  __shared__ __attribute__((aligned(16))) __bf16 Bs[3][GT][GLDK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, l32 = lane & 31, hh = lane >> 5;
  const int f = blockIdx.z;
  const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
  const long long planeV = 16LL * rows * K, planeU = 16LL * N * K;
  const __bf16* Vf = V + (long long)f * rows * K;
  const __bf16* Uf = U + (long long)f * N * K;
  const int srow = tid >> 2, sseg = (tid & 3) * 8;        // staging: 16 bytes of row srow (+64), k segment sseg
  int ar[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int r = m0 + srow + 64 * h;
    ar[h] = r < rows ? r : rows - 1;                      // rows past the end re-read the last one; their results are never stored
  }
  bf16x8 ra[3][2], rb[3][2];
  auto gload = [&](int k0) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        ra[p][h] = *reinterpret_cast<const bf16x8*>(Vf + p * planeV + (long long)ar[h] * K + k0 + sseg);
        rb[p][h] = *reinterpret_cast<const bf16x8*>(Uf + p * planeU + (long long)(n0 + srow + 64 * h) * K + k0 + sseg);
      }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int p = 0; p < 3; ++p)
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
  for (int k0 = 0; k0 < K; k0 += GBK) {
    lstore();
    __syncthreads();
    if (k0 + GBK < K) gload(k0 + GBK);
#pragma unroll
    for (int s = 0; s < GBK / 16; ++s) {
      bf16x8 a[2][3], b[2][3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          a[i][p] = *reinterpret_cast<const bf16x8*>(&As[p][wm * 64 + i * 32 + l32][s * 16 + 8 * hh]);
          b[i][p] = *reinterpret_cast<const bf16x8*>(&Bs[p][wn * 64 + i * 32 + l32][s * 16 + 8 * hh]);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x16 c = acc[i][j];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);
          acc[i][j] = c;
        }
    }
    __syncthreads();
  }
  float* Mf = M + (long long)f * rows * N;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        const int col = n0 + wn * 64 + j * 32 + l32;
        if (row < rows) Mf[(long long)row * N + col] = acc[i][j][r];
      }
}

// Big-tile form: 256 x 256 per workgroup, 8 waves of 64 (rows) x 128 (columns), dynamic LDS (2 x 3 x 256 x 80 B = 120 KB, one
// workgroup per CU).  The 128 x 128 kernel above reads (128 + 128) operand rows per 16,384 outputs, i.e. 48 KB per 1,536 MFMA cycles
// and workgroup -- with two workgroups per CU that is the 64 B/clk a CU can take from L2, which is what holds it at 40 % of the
// six-product rate; this one reads half as much per output.
__global__ __launch_bounds__(512) void wino_gemm_bf16x6_big_kernel(const __bf16* __restrict__ V, const __bf16* __restrict__ U, float* __restrict__ M,
                                                                   int rows, int N, int K) {
  constexpr int BT = 256;
  extern __shared__ __attribute__((aligned(16))) __bf16 big_smem[];
  __bf16 (*As)[BT][GLDK] = reinterpret_cast<__bf16 (*)[BT][GLDK]>(big_smem);
  __bf16 (*Bs)[BT][GLDK] = reinterpret_cast<__bf16 (*)[BT][GLDK]>(big_smem + 3 * BT * GLDK);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, l32 = lane & 31, hh = lane >> 5;       // 4 x 2 waves
  const int f = blockIdx.z;
  const int m0 = blockIdx.y * BT, n0 = blockIdx.x * BT;
  const long long planeV = 16LL * rows * K, planeU = 16LL * N * K;
  const __bf16* Vf = V + (long long)f * rows * K;
  const __bf16* Uf = U + (long long)f * N * K;
  const int srow = tid >> 2, sseg = (tid & 3) * 8;        // 128 rows x 4 segments per pass, two passes per plane and operand
  int ar[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int r = m0 + srow + 128 * h;
    ar[h] = r < rows ? r : rows - 1;
  }
  bf16x8 ra[3][2], rb[3][2];
  auto gload = [&](int k0) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        ra[p][h] = *reinterpret_cast<const bf16x8*>(Vf + p * planeV + (long long)ar[h] * K + k0 + sseg);
        rb[p][h] = *reinterpret_cast<const bf16x8*>(Uf + p * planeU + (long long)(n0 + srow + 128 * h) * K + k0 + sseg);
      }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        *reinterpret_cast<bf16x8*>(&As[p][srow + 128 * h][sseg]) = ra[p][h];
        *reinterpret_cast<bf16x8*>(&Bs[p][srow + 128 * h][sseg]) = rb[p][h];
      }
  };
  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  gload(0);
  for (int k0 = 0; k0 < K; k0 += GBK) {
    lstore();
    __syncthreads();
    if (k0 + GBK < K) gload(k0 + GBK);
#pragma unroll
    for (int s = 0; s < GBK / 16; ++s) {
      bf16x8 a[2][3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) a[i][p] = *reinterpret_cast<const bf16x8*>(&As[p][wm * 64 + i * 32 + l32][s * 16 + 8 * hh]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bf16x8 b[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) b[p] = *reinterpret_cast<const bf16x8*>(&Bs[p][wn * 128 + j * 32 + l32][s * 16 + 8 * hh]);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          f32x16 c = acc[i][j];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[0], c, 0, 0, 0);
          acc[i][j] = c;
        }
      }
    }
    __syncthreads();
  }
  float* Mf = M + (long long)f * rows * N;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        const int col = n0 + wn * 128 + j * 32 + l32;
        if (row < rows) Mf[(long long)row * N + col] = acc[i][j][r];
      }
}

bool wino_gemm_bf16x6_supported(int N, int K) { return N % GT == 0 && K % GBK == 0; }

// V: [3][16][rows][K] bf16, U: [3][16][N][K] bf16, M: [16][rows][N] fp32
hipError_t launch_wino_gemm_bf16x6(const void* V, const void* U, float* M, int rows, int N, int K, hipStream_t s) {
  if (!wino_gemm_bf16x6_supported(N, K) || rows <= 0) return hipErrorInvalidValue;
  static int big = -1;
  if (big < 0) { const char* e = getenv("US_BF16X6_BIG"); big = e ? atoi(e) : 1; }      // 0: 128 x 128 tiles, 1: 256 x 256
  if (big && N % 256 == 0) {
    static bool attr = false;
    const int lds = 2 * 3 * 256 * GLDK * (int)sizeof(__bf16);
    if (!attr) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_gemm_bf16x6_big_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) return e;
      attr = true;
    }
    dim3 g2(N / 256, (rows + 255) / 256, 16);
    hipLaunchKernelGGL(wino_gemm_bf16x6_big_kernel, g2, dim3(512), lds, s, reinterpret_cast<const __bf16*>(V), reinterpret_cast<const __bf16*>(U), M,
                       rows, N, K);
    return hipGetLastError();
  }
  dim3 grid(N / GT, (rows + GT - 1) / GT, 16);
  hipLaunchKernelGGL(wino_gemm_bf16x6_kernel, grid, dim3(256), 0, s, reinterpret_cast<const __bf16*>(V), reinterpret_cast<const __bf16*>(U), M, rows,
                     N, K);
  return hipGetLastError();
}

// U = G g G^T per (co, ci) as three bf16 planes: dst[p][f][co][ci] (K = ci contiguous); src Conv2d OIHW 3x3
__global__ void wino_pack_weight_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, int Cout, int Cin) {
  const long long total = (long long)Cout * Cin;
  const long long plane = 16 * total;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const float* g = src + i * 9;
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
        const long long o = (long long)(r * 4 + q) * total + i;      // [f][co][ci]
        const __bf16 a1 = (__bf16)u[q];
        const float r1 = u[q] - (float)a1;
        const __bf16 a2 = (__bf16)r1;
        dst[o] = a1;
        dst[plane + o] = a2;
        dst[2 * plane + o] = (__bf16)(r1 - (float)a2);
      }
    }
  }
}

hipError_t launch_wino_pack_weight_bf16(const float* src, void* dst, int Cout, int Cin, hipStream_t s) {
  long long total = (long long)Cout * Cin;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(wino_pack_weight_bf16_kernel, dim3(blocks), dim3(256), 0, s, src, reinterpret_cast<__bf16*>(dst), Cout, Cin);
  return hipGetLastError();
}

}  // namespace us
