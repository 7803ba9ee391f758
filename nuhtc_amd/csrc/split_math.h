// Device helpers shared by the kernels that run fp32 products on the bf16 matrix pipe (gemm.hip, mlp.hip): the exact three-way
// bf16 split of an fp32 value, the order of the six products, and the erf-GELU of the FFN.
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// erf-GELU (torch.nn.GELU default, mmcv FFN act_cfg).  erf by the clamped odd rational x·P(x²)/Q(x²) (degrees 6 / 4 in x²:
// 11 fma + 1 rcp), max abs error 4.5e-7 over all floats -- half the instructions of the libm erff, which was 10-15 % of
// the FFN fc1 launches at K = 96..192.
__device__ __forceinline__ float gelu_erf(float v) {
  float x = v * 0.70710678118654752440f;
  x = fminf(fmaxf(x, -4.0f), 4.0f);
  const float x2 = x * x;
  float pn = -2.72614225801306e-10f;
  pn = fmaf(pn, x2, 2.77068142495902e-08f);
  pn = fmaf(pn, x2, -2.10102402082508e-06f);
  pn = fmaf(pn, x2, -5.69250639462346e-05f);
  pn = fmaf(pn, x2, -7.34990630326855e-04f);
  pn = fmaf(pn, x2, -2.95459980854025e-03f);
  pn = fmaf(pn, x2, -1.60960333262415e-02f);
  float qd = -1.45660718464996e-05f;
  qd = fmaf(qd, x2, -2.13374055278905e-04f);
  qd = fmaf(qd, x2, -1.68282697438203e-03f);
  qd = fmaf(qd, x2, -7.37332916720468e-03f);
  qd = fmaf(qd, x2, -1.42647390514189e-02f);
  const float e = x * pn * __builtin_amdgcn_rcpf(qd);
  return 0.5f * v * (1.0f + e);
}

__device__ __forceinline__ unsigned pk_bf16_rn(float lo, float hi) {   // two floats -> packed bf16 (round to nearest even)
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  bf2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}

// x, y -> dword of plane 1, 2, 3 (bf16(x) in the low half): round to nearest at every level, the residuals are exact fp32 subtractions
struct Split3 { unsigned p1, p2, p3; };
__device__ __forceinline__ Split3 split3_pair(float x, float y) {
  Split3 o;
  o.p1 = pk_bf16_rn(x, y);
  const float rx = x - __uint_as_float(o.p1 << 16), ry = y - __uint_as_float(o.p1 & 0xffff0000u);
  o.p2 = pk_bf16_rn(rx, ry);
  const float sx = rx - __uint_as_float(o.p2 << 16), sy = ry - __uint_as_float(o.p2 & 0xffff0000u);
  o.p3 = pk_bf16_rn(sx, sy);
  return o;
}
#define NUHTC_SPLIT3_INTO(P_, d_, x_, y_) { const Split3 s3_ = split3_pair(x_, y_); (P_)[0][d_] = s3_.p1; (P_)[1][d_] = s3_.p2; (P_)[2][d_] = s3_.p3; }

// the six products of one 16-deep step, smallest terms first: a3b1, a1b3, a2b2, a2b1, a1b2, a1b1 (planes 0..2 = a1..a3)
__device__ __forceinline__ f32x16 mfma_split6(const u32x4 (&a)[3], const u32x4 (&b)[3], f32x16 acc) {
#define NUHTC_M6(ia_, ib_) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[ia_]), __builtin_bit_cast(bf16x8, b[ib_]), acc, 0, 0, 0);
  NUHTC_M6(2, 0) NUHTC_M6(0, 2) NUHTC_M6(1, 1) NUHTC_M6(1, 0) NUHTC_M6(0, 1) NUHTC_M6(0, 0)
#undef NUHTC_M6
  return acc;
}

// the same six products with the operand roles of the transposed kernels (a = weights as the MFMA's A operand, b = activations as
// its B operand) in the order gemm_split_kernel issues them (activation plane, weight plane): (3,1) (1,3) (2,2) (2,1) (1,2) (1,1)
__device__ __forceinline__ f32x16 mfma_split6_wa(const u32x4 (&w)[3], const u32x4 (&act)[3], f32x16 acc) {
#define NUHTC_M6(iw_, ia_) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w[iw_]), __builtin_bit_cast(bf16x8, act[ia_]), acc, 0, 0, 0);
  NUHTC_M6(0, 2) NUHTC_M6(2, 0) NUHTC_M6(1, 1) NUHTC_M6(0, 1) NUHTC_M6(1, 0) NUHTC_M6(0, 0)
#undef NUHTC_M6
  return acc;
}
