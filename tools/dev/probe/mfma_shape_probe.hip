// Dev probe (not part of the product): sustained rate of the six-product pattern of gemm_split_kernel as bare MFMA loops on random
// bf16 operands held in registers (two operand sets alternating per k-step so the inputs toggle like a real k-loop):
//   shape 0: v_mfma_f32_32x32x16_bf16, wave tile 32 x 96 (3 accumulators of 16), 18 MFMAs per 16-deep step
//   shape 1: v_mfma_f32_16x16x32_bf16, wave tile 32 x 96 (12 accumulators of 4), 72 MFMAs per 32-deep step
// Usage: mfma_shape_probe [waves_per_simd=2] [zero=0]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k32(const u32x4* __restrict__ src, float* __restrict__ out, int iters) {
  const int tid = threadIdx.x + blockIdx.x * 256;
  bf16x8 a[2][3], b[2][3][3];
  for (int s = 0; s < 2; ++s)
    for (int p = 0; p < 3; ++p) {
      a[s][p] = __builtin_bit_cast(bf16x8, src[(tid * 32 + s * 3 + p) & 0xfffff]);
      for (int t = 0; t < 3; ++t) b[s][t][p] = __builtin_bit_cast(bf16x8, src[(tid * 32 + 6 + s * 9 + t * 3 + p) & 0xfffff]);
    }
  f32x16 acc[3];
  for (int t = 0; t < 3; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][2], b[s][t][0], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][0], b[s][t][2], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][1], b[s][t][1], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][1], b[s][t][0], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][0], b[s][t][1], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][0], b[s][t][0], acc[t], 0, 0, 0);
      }
  }
  float v = 0.f;
  for (int t = 0; t < 3; ++t) for (int r = 0; r < 16; ++r) v += acc[t][r];
  out[tid] = v;
}

__global__ __launch_bounds__(256) void k16(const u32x4* __restrict__ src, float* __restrict__ out, int iters) {
  const int tid = threadIdx.x + blockIdx.x * 256;
  bf16x8 a[2][3], b[6][3];
  for (int m = 0; m < 2; ++m)
    for (int p = 0; p < 3; ++p) a[m][p] = __builtin_bit_cast(bf16x8, src[(tid * 32 + m * 3 + p) & 0xfffff]);
  for (int t = 0; t < 6; ++t)
    for (int p = 0; p < 3; ++p) b[t][p] = __builtin_bit_cast(bf16x8, src[(tid * 32 + 6 + t * 3 + p) & 0xfffff]);
  f32x4 acc[2][6];
  for (int m = 0; m < 2; ++m) for (int t = 0; t < 6; ++t) for (int r = 0; r < 4; ++r) acc[m][t][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m][2], b[t][0], acc[m][t], 0, 0, 0);
        acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m][0], b[t][2], acc[m][t], 0, 0, 0);
        acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m][1], b[t][1], acc[m][t], 0, 0, 0);
        acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m][1], b[t][0], acc[m][t], 0, 0, 0);
        acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m][0], b[t][1], acc[m][t], 0, 0, 0);
        acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m][0], b[t][0], acc[m][t], 0, 0, 0);
      }
    // rotate the operand registers so consecutive steps see different inputs (cheap: swaps of whole registers)
    bf16x8 t0 = a[0][0]; a[0][0] = a[1][1]; a[1][1] = a[0][2]; a[0][2] = a[1][0]; a[1][0] = a[0][1]; a[0][1] = a[1][2]; a[1][2] = t0;
  }
  float v = 0.f;
  for (int m = 0; m < 2; ++m) for (int t = 0; t < 6; ++t) for (int r = 0; r < 4; ++r) v += acc[m][t][r];
  out[tid] = v;
}

int main(int argc, char** argv) {
  const int wps = argc > 1 ? atoi(argv[1]) : 2, zero = argc > 2 ? atoi(argv[2]) : 0;
  const int blocks = 256 * wps;
  std::vector<unsigned> h(4 << 20);
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  for (auto& w : h) {
    float x = nd(rng), y = nd(rng);
    unsigned ux, uy; memcpy(&ux, &x, 4); memcpy(&uy, &y, 4);
    w = zero ? 0u : ((ux >> 16) | (uy & 0xffff0000u));
  }
  u32x4* src; float* out;
  hipMalloc(&src, h.size() * 4); hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int shape = 0; shape < 2; ++shape) {
    const int iters = shape == 0 ? 2000 : 1000;     // both: 2000 x 36 x 32768 flop x ... same executed flop per wave
    for (int rep = 0; rep < 2; ++rep) {
      const int launches = 40;
      hipEventRecord(e0);
      for (int i = 0; i < launches; ++i) {
        if (shape == 0) hipLaunchKernelGGL(k32, dim3(blocks), dim3(256), 0, 0, src, out, iters);
        else hipLaunchKernelGGL(k16, dim3(blocks), dim3(256), 0, 0, src, out, iters);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double mf = shape == 0 ? 36.0 * 2 * 32 * 32 * 16 : 72.0 * 2 * 16 * 16 * 32;
      const double flop = (double)blocks * 4 * iters * mf * launches;
      if (rep) printf("shape %s  waves/SIMD %d  %s: %.3f ms per launch, %.1f TFLOP/s executed (%.1f fp32-equivalent)\n", shape == 0 ? "32x32x16" : "16x16x32", wps,
                      zero ? "zeros" : "random", ms / launches, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / 6);
    }
  }
  return 0;
}
