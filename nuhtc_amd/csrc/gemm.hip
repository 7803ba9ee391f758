// fp32 GEMM on the CDNA4 matrix cores: C = epilogue(A · Wᵀ), v_mfma_f32_32x32x2_f32 (exact f32, k-ordered fma
// chain), 128 x (32·NT) x 32 block tile, 4 waves (each 32 rows x 32·NT cols), double-buffered LDS with a
// k-major XOR-swizzled image (conflict-free transposing writes and fragment reads), register prefetch of the
// next k-tile.  Serves every dense layer of the path: Swin QKV/proj/FFN/merge linears (mmdet swin.py:88,115,
// mmcv FFN, transformer.py:384), FPN / RPN / semantic / mask-head convolutions as NHWC implicit GEMM
// (fpn.py:152-179, rpn_head.py:62-68, fused_semantic_head.py:97-111, htc_mask_head.py:22-39), the bbox-head
// FCs (convfc_bbox_head.py:158-196) and the attention-pool similarity / aggregation products
// (nuhtc/models/roi_extractors_cus.py:228-235).
#include <cstdlib>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BM 128

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

template <int NT, int BK>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
  constexpr int BN = 32 * NT;
  constexpr int KC = BK / 4;          // float4 chunks per tile row
  constexpr int RPP = 256 / KC;       // rows staged per pass of the 256 threads
  constexpr int NA = BM / RPP;        // passes for the A tile
  constexpr int NB = (BN + RPP - 1) / RPP;   // passes for the W tile (last one may be partial)
  __shared__ float lds[2 * BK * (BM + BN)];
  float* As = lds;                 // [2][BK][BM]
  float* Bs = lds + 2 * BK * BM;   // [2][BK][BN]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int i32 = lane & 31, half = lane >> 5;
  const int nTilesN = p.N / BN;
  const int tile_m = blockIdx.x / nTilesN, tile_n = blockIdx.x % nTilesN;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int z = blockIdx.z;

  int Meff = p.M;
  if (p.m_dev) {
    int md = *p.m_dev * p.m_mul;
    Meff = md < Meff ? md : Meff;
  }
  if (m0 >= Meff) return;

  const float* __restrict__ A = p.A + (long long)z * p.sA;
  const float* __restrict__ Wt = p.W + (long long)z * p.sW;
  float* __restrict__ C = p.C + (long long)z * p.sC;

  // ---- per-thread staging assignment: float4 slots idx = tid + 256*j -> (row = idx/8, kc = idx%8)
  const int kc = tid % KC;
  const int rbase = tid / KC;   // 0..RPP-1
  const float* a_ptr[NA];
  int a_y[NA], a_x[NA];
  bool a_ok[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    int m = m0 + rbase + RPP * j;
    a_ok[j] = m < Meff;
    if (p.amode == A_PLAIN) {
      a_ptr[j] = A + (long long)(a_ok[j] ? m : 0) * p.lda + kc * 4;
      a_y[j] = a_x[j] = 0;
    } else {
      int mm = a_ok[j] ? m : 0;
      int hw = p.cH * p.cW;
      int b = mm / hw, r = mm - b * hw;
      int y = r / p.cW, x = r - y * p.cW;
      a_y[j] = y;
      a_x[j] = x;
      a_ptr[j] = A + ((long long)(b * p.cH + y) * p.cW + x) * p.cC + kc * 4;
    }
  }
  const float* w_ptr[NB];
  bool w_ok[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    w_ok[j] = rbase + RPP * j < BN;
    w_ptr[j] = Wt + (long long)(n0 + (w_ok[j] ? rbase + RPP * j : 0)) * p.K + kc * 4;
  }

  float4 ra[NA], rb[NB];
  auto load_tile = [&](int kt) {
    if (p.amode == A_PLAIN) {
#pragma unroll
      for (int j = 0; j < NA; ++j)
        ra[j] = a_ok[j] ? *reinterpret_cast<const float4*>(a_ptr[j] + kt * BK) : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      int kk = kt * BK;
      int tap = kk / p.cC, c0 = kk - tap * p.cC;
      int ky = tap / 3 - 1, kx = tap - (tap / 3) * 3 - 1;
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        int yy = a_y[j] + ky, xx = a_x[j] + kx;
        bool ok = a_ok[j] && yy >= 0 && yy < p.cH && xx >= 0 && xx < p.cW;
        ra[j] = ok ? *reinterpret_cast<const float4*>(a_ptr[j] + (long long)(ky * p.cW + kx) * p.cC + c0)
                   : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j)
      if (w_ok[j]) rb[j] = *reinterpret_cast<const float4*>(w_ptr[j] + kt * BK);
  };
  auto store_tile = [&](int buf) {
    float* as = As + buf * BK * BM;
    float* bs = Bs + buf * BK * BN;
    const int sw = kc << 2;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      int r = (rbase + RPP * j) ^ sw;
      as[(kc * 4 + 0) * BM + r] = ra[j].x;
      as[(kc * 4 + 1) * BM + r] = ra[j].y;
      as[(kc * 4 + 2) * BM + r] = ra[j].z;
      as[(kc * 4 + 3) * BM + r] = ra[j].w;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if (!w_ok[j]) continue;
      int r = (rbase + RPP * j) ^ sw;
      bs[(kc * 4 + 0) * BN + r] = rb[j].x;
      bs[(kc * 4 + 1) * BN + r] = rb[j].y;
      bs[(kc * 4 + 2) * BN + r] = rb[j].z;
      bs[(kc * 4 + 3) * BN + r] = rb[j].w;
    }
  };

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int nk = p.K / BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
    const float* as = As + buf * BK * BM + wave * 32;
    const float* bs = Bs + buf * BK * BN;
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
      const int k = 2 * ks + half;
      const int sw = (k >> 2) << 2;
      const float a = as[k * BM + (i32 ^ sw)];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float b = bs[k * BN + ((32 * t + i32) ^ sw)];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
      }
    }
    if (kt + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds column n = n0 + 32t + i32 and rows (r&3) + 8(r>>2) + 4·half of the wave's 32
  const float* ri = p.cos_ri ? p.cos_ri + (long long)z * p.sRi : nullptr;
  const float* rj = p.cos_rj ? p.cos_rj + (long long)z * p.sRj : nullptr;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (m >= Meff) continue;
    long long drow = m;
    if (p.store == ST_ROWMAP) {
      int d = p.row_map[m];
      if (d < 0) continue;
      drow = d;
    }
    long long uprow = 0;
    if (p.up) {
      int hw = p.upH * p.upW;
      int b = m / hw, rr = m - b * hw;
      int y = rr / p.upW, x = rr - y * p.upW;
      uprow = ((long long)b * (p.upH >> 1) + (y >> 1)) * (p.upW >> 1) + (x >> 1);
    }
    long long dbase = 0;
    if (p.store == ST_DECONV2) {
      // rows m = (d, y, x) on a cH x cW grid; columns n = (kh*2+kw)*ldc + oc -> out[(d, 2y+kh, 2x+kw), oc]
      int hw = p.cH * p.cW;
      int b = m / hw, rr = m - b * hw;
      int y = rr / p.cW, x = rr - y * p.cW;
      dbase = ((long long)b * (2 * p.cH) + 2 * y) * (2 * p.cW) + 2 * x;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = n0 + 32 * t + i32;
      float v = acc[t][r] * p.alpha;
      if (p.bias) v += p.bias[n];
      if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
      else if (p.act == ACT_GELU) v = gelu_erf(v);
      else if (p.act == ACT_COS) v = fmaxf(v * ri[m] * rj[n] - p.cos_tau, 0.f) + p.cos_tau;
      if (p.up) v += p.up[uprow * p.N + n];
      if (p.res) v += p.res[drow * p.ldr + n];
      if (p.store == ST_DECONV2) {
        int tap = n / p.ldc, oc = n - tap * p.ldc;
        C[(dbase + (tap >> 1) * (2 * p.cW) + (tap & 1)) * p.ldc + oc] = v;
      } else {
        C[drow * p.ldc + n] = v;
      }
    }
  }
}

static int g_bk = 0;
int launch_gemm(const GemmParams& p, hipStream_t s) {
  if (p.M <= 0) return 0;
  if (g_bk == 0) { const char* e = getenv("NUHTC_GEMM_BK"); g_bk = e ? atoi(e) : 16; if (g_bk != 32) g_bk = 16; }
  if (p.K % 32 != 0 || p.N % 32 != 0) return NUHTC_E_INVALID;
  if (p.amode == A_CONV3 && (p.cC % 32 != 0 || p.K != 9 * p.cC)) return NUHTC_E_INVALID;
  int nt = (p.N % 96 == 0) ? 3 : (p.N % 128 == 0) ? 4 : (p.N % 64 == 0) ? 2 : 1;
  int bn = 32 * nt;
  dim3 grid(cdiv(p.M, BM) * (p.N / bn), 1, p.batch > 0 ? p.batch : 1);
  GemmParams q = p;
  if (q.alpha == 0.f) q.alpha = 1.f;
  const double nb = p.batch > 0 ? p.batch : 1;
  const char* tag = "gemm";
  if (prof_enabled()) {   // per-shape tags, e.g. "gemm_kernel<3>|N288|K96" (strings live for the process lifetime)
    static std::map<long long, std::string> names;
    long long key = ((long long)nt << 40) | ((long long)p.N << 20) | p.K | ((long long)(p.amode == A_CONV3) << 44);
    auto it = names.find(key);
    if (it == names.end())
      it = names.emplace(key, "gemm_kernel<" + std::to_string(nt) + ">|N" + std::to_string(p.N) + "|K" + std::to_string(p.K) + (p.amode == A_CONV3 ? "|conv3" : "")).first;
    tag = it->second.c_str();
  }
  // algorithmic work of the launch (device-side row counts are not known here: the capacity M is an upper bound)
  ProfScope ps(tag, 2.0 * p.M * p.N * p.K * nb, 4.0 * nb * ((double)p.M * p.K + (double)p.N * p.K + (double)p.M * p.N), s);
  if (g_bk == 16) {
    switch (nt) {
      case 1: hipLaunchKernelGGL((gemm_kernel<1, 16>), grid, dim3(256), 0, s, q); break;
      case 2: hipLaunchKernelGGL((gemm_kernel<2, 16>), grid, dim3(256), 0, s, q); break;
      case 3: hipLaunchKernelGGL((gemm_kernel<3, 16>), grid, dim3(256), 0, s, q); break;
      default: hipLaunchKernelGGL((gemm_kernel<4, 16>), grid, dim3(256), 0, s, q); break;
    }
  } else {
    switch (nt) {
      case 1: hipLaunchKernelGGL((gemm_kernel<1, 32>), grid, dim3(256), 0, s, q); break;
      case 2: hipLaunchKernelGGL((gemm_kernel<2, 32>), grid, dim3(256), 0, s, q); break;
      case 3: hipLaunchKernelGGL((gemm_kernel<3, 32>), grid, dim3(256), 0, s, q); break;
      default: hipLaunchKernelGGL((gemm_kernel<4, 32>), grid, dim3(256), 0, s, q); break;
    }
  }
  return hipGetLastError() == hipSuccess ? 0 : NUHTC_E_HIP;
}
