// Dev probe (not part of the product): fp32 GEMM through the bf16 matrix pipe by exact operand splitting.
//   a = a1 + a2 + a3 (three bf16, round-to-nearest splits: exact for every finite fp32), same for b; the products
//   a_i * b_j are exact in fp32, so  a*b = sum of 9 bf16 products; dropping a2b3, a3b2, a3b3 (<= 2^-26 |ab|) leaves 6.
// Part A: accuracy of one 32x32 tile against an fp64 reference for K = 96 .. 3136: fp32 MFMA, 3 / 6 / 9 bf16 products.
// Part B: chip-wide rate of the 6-product loop (operands in registers, A split in the loop) against the fp32 MFMA loop.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pk_bf16_rn(float lo, float hi) {   // two floats -> packed bf16 (RN), lo in bits 0..15
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  bf2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}
// split 8 floats into three bf16x8 planes (RN at every level; residuals are exact fp32 subtractions)
__device__ __forceinline__ void split8(const float* a, bf16x8& p1, bf16x8& p2, bf16x8& p3) {
  u32x4 q1, q2, q3;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float x = a[2 * i], y = a[2 * i + 1];
    const unsigned w1 = pk_bf16_rn(x, y);
    const float rx = x - __uint_as_float(w1 << 16), ry = y - __uint_as_float(w1 & 0xffff0000u);
    const unsigned w2 = pk_bf16_rn(rx, ry);
    const float sx = rx - __uint_as_float(w2 << 16), sy = ry - __uint_as_float(w2 & 0xffff0000u);
    const unsigned w3 = pk_bf16_rn(sx, sy);
    q1[i] = w1; q2[i] = w2; q3[i] = w3;
  }
  p1 = __builtin_bit_cast(bf16x8, q1); p2 = __builtin_bit_cast(bf16x8, q2); p3 = __builtin_bit_cast(bf16x8, q3);
}

// one wave: C[32][32] = A[32][K] * B[32][K]^T  (both row-major with K contiguous); mode 0 fp32 MFMA, 3 / 6 / 9 bf16 products
__global__ void tile_kernel(const float* A, const float* B, float* C, int K, int mode) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  if (mode == 0) {
    for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k + h], B[r * K + k + h], acc, 0, 0, 0);
  } else {
    for (int k = 0; k < K; k += 16) {
      float a[8], b[8];
      for (int j = 0; j < 8; ++j) { a[j] = A[r * K + k + 8 * h + j]; b[j] = B[r * K + k + 8 * h + j]; }
      bf16x8 a1, a2, a3, b1, b2, b3;
      split8(a, a1, a2, a3); split8(b, b1, b2, b3);
      // smallest terms first
      if (mode >= 9) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b3, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b3, acc, 0, 0, 0);
      }
      if (mode >= 6) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
    }
  }
  for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
}

// rate: every wave keeps NT accumulators, B planes in registers (pre-split), A re-split from fp32 registers every step
template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(const float* src, float* dst, int iters) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[3];
  for (int t = 0; t < 3; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  float a[8], b[3][8];
  for (int j = 0; j < 8; ++j) { a[j] = src[(blockIdx.x * 256 + threadIdx.x) * 8 % 65536 + j]; for (int t = 0; t < 3; ++t) b[t][j] = src[(threadIdx.x * 8 + j + 977 * t) % 65536]; }
  if (MODE == 0) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[t][j], acc[t], 0, 0, 0);
      a[it & 7] += 1e-9f;
    }
  } else {
    bf16x8 b1[3], b2[3], b3[3];
    for (int t = 0; t < 3; ++t) split8(b[t], b1[t], b2[t], b3[t]);
    for (int it = 0; it < iters; ++it) {
      bf16x8 a1, a2, a3;
      split8(a, a1, a2, a3);        // the A operand arrives as fp32 and is split in the loop (VALU beside the MFMAs)
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1[t], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3[t], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2[t], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1[t], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2[t], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1[t], acc[t], 0, 0, 0);
      }
      a[it & 7] += 1e-9f;
    }
  }
  float s = 0.f;
  for (int t = 0; t < 3; ++t) for (int i = 0; i < 16; ++i) s += acc[t][i];
  dst[blockIdx.x * 256 + threadIdx.x] = s + (float)lane;
}

int main() {
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  printf("accuracy of one 32x32 tile, error normalised by sum_k |a_k b_k| (fp32 rounding unit 2^-24 = 6.0e-8)\n");
  for (int dist = 0; dist < 2; ++dist)
    for (int K : {96, 384, 1536, 3136}) {
      std::vector<float> A(32 * K), B(32 * K), C(1024);
      for (auto& v : A) v = dist ? nd(rng) * std::exp(2.f * nd(rng)) : nd(rng);      // dist 1: wide dynamic range
      for (auto& v : B) v = dist ? nd(rng) * std::exp(2.f * nd(rng)) : nd(rng);
      float *dA, *dB, *dC;
      hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096);
      hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
      printf("%s K=%4d:", dist ? "lognormal-scaled" : "normal          ", K);
      for (int mode : {0, 3, 6, 9}) {
        hipLaunchKernelGGL(tile_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, mode);
        hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
        double worst = 0, rms = 0;
        for (int i = 0; i < 32; ++i)
          for (int j = 0; j < 32; ++j) {
            double ref = 0, mag = 0;
            for (int k = 0; k < K; ++k) { double p = (double)A[i * K + k] * (double)B[j * K + k]; ref += p; mag += std::fabs(p); }
            double e = std::fabs((double)C[i * 32 + j] - ref) / mag;
            worst = std::max(worst, e); rms += e * e;
          }
        printf("  %s max %.2e rms %.2e", mode == 0 ? "fp32-mfma" : mode == 3 ? "bf16x3" : mode == 6 ? "bf16x6" : "bf16x9", worst, std::sqrt(rms / 1024));
      }
      printf("\n");
      hipFree(dA); hipFree(dB); hipFree(dC);
    }
  // rate
  std::vector<float> src(65536);
  for (auto& v : src) v = nd(rng);
  float *dS, *dD;
  hipMalloc(&dS, src.size() * 4); hipMalloc(&dD, 4096 * 256 * 4);
  hipMemcpy(dS, src.data(), src.size() * 4, hipMemcpyHostToDevice);
  const int iters = 20000, blocks = 1024;      // 4 blocks of 4 waves per CU
  for (int mode = 0; mode < 2; ++mode) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(blocks), dim3(256), 0, 0, dS, dD, iters);
      else hipLaunchKernelGGL(rate_kernel<6>, dim3(blocks), dim3(256), 0, 0, dS, dD, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      // fp32-equivalent flops: per iteration and wave 3 tiles of 32x32 with K = 16
      const double flops = (double)blocks * 4 * iters * 3 * 32 * 32 * 16 * 2;
      printf("%s: %.2f ms, %.1f TFLOP/s fp32-equivalent\n", mode == 0 ? "fp32 MFMA loop (8 x 32x32x2 per tile-step)" : "bf16 x6 loop (6 x 32x32x16 per tile-step, A split in the loop)", ms, flops / ms / 1e9);
    }
  }
  return 0;
}
