// Swin-T front-end and per-block non-GEMM kernels (gfx950, wave64).
//   preproc       : uint8 x2 bilinear resize (cv2 INTER_LINEAR fixed point) + channel swap + normalise
//                   (mmdet/datasets/pipelines/transforms.py:207-236,686-700)
//   patch_embed   : conv4x4 s4 (3->96) + LayerNorm(96)          (mmdet/models/utils/transformer.py:236-257)
//   layernorm     : LN with optional row gather (LN1 + pad + cyclic roll + window partition in one pass;
//                   padding rows are zeros *after* the norm)    (mmdet/models/backbones/swin.py:182-226,360)
//   merge_ln      : PatchMerging 2x2 gather + LN(4C)            (transformer.py:363-385)
//   window_attn   : softmax(q·s·kᵀ + relpos_bias + shift_mask)·v per (window, head), N=49, d=32
//                   (swin.py:79-117); one wave per (window, head), K/V staged in LDS, one query row per lane,
//                   scores/softmax entirely in registers.
#include <cmath>
#include <cstdlib>

#include "common.h"
#include "split_math.h"   // f32x16 / v4f typedefs, the exact bf16 split and its six products

// ----------------------------------------------------------------------------- preproc
// cv2.resize(INTER_LINEAR) on uint8 exactly as OpenCV's 8-bit fixed-point path computes it (the per-axis tables are
// built on the host by cv_linear_tables() the way cv::resize builds xofs/ialpha/yofs/ibeta):
//   horizontal  S = p[sx]*a0 + p[sx1]*a1                  (11 fractional bits, border columns: one tap at 2048)
//   vertical    ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16)   two separately truncated products, rows clamped, weights kept
//   dst         (v + 2) >> 2
// then mmcv.imnormalize (channel swap, (x-mean)*(1/std)).   mmdet/datasets/pipelines/transforms.py:207-236,686-700
struct NormConst { float mean[3]; float istd[3]; };

void cv_linear_tables(int ssize, int dsize, bool horizontal, std::vector<int>& tab) {
  tab.resize((size_t)dsize * 4);
  const double scale = 1.0 / ((double)dsize / (double)ssize);
  for (int d = 0; d < dsize; ++d) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s0 = (int)floorf(f);
    f -= (float)s0;
    if (horizontal) {   // cv::resize resets the weight at the left / right border ...
      if (s0 < 0) { f = 0.f; s0 = 0; }
      if (s0 >= ssize - 1) { f = 0.f; s0 = ssize - 1; }
    }
    const int w0 = (int)lrintf((1.f - f) * 2048.f), w1 = (int)lrintf(f * 2048.f);   // saturate_cast<short>: round half to even
    int i0 = s0, i1 = s0 + 1;
    i0 = i0 < 0 ? 0 : (i0 > ssize - 1 ? ssize - 1 : i0);   // ... the vertical pass only clamps the row indices
    i1 = i1 < 0 ? 0 : (i1 > ssize - 1 ? ssize - 1 : i1);
    tab[(size_t)d * 4 + 0] = i0; tab[(size_t)d * 4 + 1] = i1; tab[(size_t)d * 4 + 2] = w0; tab[(size_t)d * 4 + 3] = w1;
  }
}

__global__ void preproc_kernel(const uint8_t* __restrict__ tiles, float* __restrict__ img, int B, int th, int tw, int Hn, int Wn, int Hv, int Wv,
                               const int4* __restrict__ xtab, const int4* __restrict__ ytab, int swap, NormConst nc) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = (long long)B * Hn * Wn;
  if (idx >= total) return;
  int x = idx % Wn;
  int y = (idx / Wn) % Hn;
  int b = idx / ((long long)Wn * Hn);
  float* o = img + idx * 3;
  if (x >= Wv || y >= Hv) {   // Pad(size_divisor=32, pad_val=0) acts after Normalize: zeros right of / below the resized image
    o[0] = 0.f; o[1] = 0.f; o[2] = 0.f;
    return;
  }
  const int4 tx = xtab[x], ty = ytab[y];
  const uint8_t* t = tiles + (long long)b * th * tw * 3;
  const uint8_t* p00 = t + ((long long)ty.x * tw + tx.x) * 3;
  const uint8_t* p01 = t + ((long long)ty.x * tw + tx.y) * 3;
  const uint8_t* p10 = t + ((long long)ty.y * tw + tx.x) * 3;
  const uint8_t* p11 = t + ((long long)ty.y * tw + tx.y) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    int sc = swap ? 2 - c : c;
    int s0 = p00[sc] * tx.z + p01[sc] * tx.w;
    int s1 = p10[sc] * tx.z + p11[sc] * tx.w;
    int u = (((ty.z * (s0 >> 4)) >> 16) + ((ty.w * (s1 >> 4)) >> 16) + 2) >> 2;
    o[c] = ((float)u - nc.mean[c]) * nc.istd[c];
  }
}

int launch_preproc(const uint8_t* tiles, float* img, int B, int th, int tw, int Hn, int Wn, int Hv, int Wv, const int* xtab, const int* ytab, int swap,
                   const float* mean_istd, hipStream_t s) {
  ProfScope ps("preproc", 0, (double)B * (3.0 * th * tw + 12.0 * Hn * Wn), s);
  NormConst nc;
  for (int i = 0; i < 3; ++i) { nc.mean[i] = mean_istd[i]; nc.istd[i] = mean_istd[3 + i]; }
  long long total = (long long)B * Hn * Wn;
  hipLaunchKernelGGL(preproc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, tiles, img, B, th, tw, Hn, Wn, Hv, Wv,
                     (const int4*)xtab, (const int4*)ytab, swap, nc);
  return hipGetLastError() == hipSuccess ? 0 : NUHTC_E_HIP;
}

// ----------------------------------------------------------------------------- patch embed
// w: [48][96] k-major with k = (kh*4+kw)*3 + c (NHWC pixel order); 32 tokens per block, 8 lanes x 12 channels per token
__global__ __launch_bounds__(256) void patch_embed_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                          const float* __restrict__ bias, const float* __restrict__ g,
                                                          const float* __restrict__ beta, float* __restrict__ tok, int nTok,
                                                          int Hn, int Wn) {
  __shared__ float wl[48 * 96];
  __shared__ float pl[32 * 49];
  const int tid = threadIdx.x;
  for (int e = tid; e < 48 * 96; e += 256) wl[e] = w[e];
  const int Wt = Wn >> 2, Ht = Hn >> 2;
  const int t0 = blockIdx.x * 32;
  for (int e = tid; e < 32 * 48; e += 256) {
    int tl = e / 48, k = e - tl * 48;
    int t = t0 + tl;
    float v = 0.f;
    if (t < nTok) {
      int tx = t % Wt, ty = (t / Wt) % Ht, b = t / (Wt * Ht);
      int kh = k / 12, r = k - kh * 12;   // r = kw*3 + c
      v = img[(((long long)b * Hn + 4 * ty + kh) * Wn + 4 * tx) * 3 + r];
    }
    pl[tl * 49 + k] = v;
  }
  __syncthreads();
  const int tl = tid >> 3, cg = tid & 7;
  float acc[12];
#pragma unroll
  for (int j = 0; j < 12; ++j) acc[j] = 0.f;
  for (int k = 0; k < 48; ++k) {
    float x = pl[tl * 49 + k];
#pragma unroll
    for (int j = 0; j < 12; ++j) acc[j] = fmaf(x, wl[k * 96 + cg + 8 * j], acc[j]);
  }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < 12; ++j) { acc[j] += bias[cg + 8 * j]; sum += acc[j]; }
  sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4);
  const float mean = sum * (1.0f / 96.0f);
  float var = 0.f;
#pragma unroll
  for (int j = 0; j < 12; ++j) { float d = acc[j] - mean; var = fmaf(d, d, var); }
  var += __shfl_xor(var, 1); var += __shfl_xor(var, 2); var += __shfl_xor(var, 4);
  const float rstd = 1.0f / sqrtf(var * (1.0f / 96.0f) + 1e-5f);
  const int t = t0 + tl;
  if (t < nTok) {
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      int c = cg + 8 * j;
      tok[(long long)t * 96 + c] = (acc[j] - mean) * rstd * g[c] + beta[c];
    }
  }
}

// The same with the pre-processing inside (round 5): the 48 inputs of a token are computed from the uint8 tile on the way into LDS -- cv2's 8-bit
// linear resize from the per-axis tables, Normalize, zero Pad: the arithmetic of preproc_kernel, element for element, so the tokens are the
// same bits -- and the normalised image (12 bytes per network pixel written and read back) never exists.  Every image pixel belongs to exactly
// one 4 x 4 patch: nothing is computed twice.
__global__ __launch_bounds__(256) void patch_embed_tiles_kernel(const uint8_t* __restrict__ tiles, int th, int tw, int Hv, int Wv,
                                                                const int4* __restrict__ xtab, const int4* __restrict__ ytab, int swap, NormConst nc,
                                                                const float* __restrict__ w, const float* __restrict__ bias, const float* __restrict__ g,
                                                                const float* __restrict__ beta, float* __restrict__ tok, int nTok, int Hn, int Wn) {
  __shared__ float wl[48 * 96];
  __shared__ float pl[32 * 49];
  const int tid = threadIdx.x;
  for (int e = tid; e < 48 * 96; e += 256) wl[e] = w[e];
  const int Wt = Wn >> 2, Ht = Hn >> 2;
  const int t0 = blockIdx.x * 32;
  for (int e = tid; e < 32 * 48; e += 256) {
    int tl = e / 48, k = e - tl * 48;
    int t = t0 + tl;
    float v = 0.f;
    if (t < nTok) {
      int tx = t % Wt, ty = (t / Wt) % Ht, b = t / (Wt * Ht);
      int kh = k / 12, r = k - kh * 12;   // r = kw*3 + c
      const int kw = r / 3, c = r - kw * 3;
      const int x = 4 * tx + kw, y = 4 * ty + kh;
      if (x < Wv && y < Hv) {             // (right of / below the resized image: the zeros of Pad)
        const int4 ax = xtab[x], ay = ytab[y];
        const uint8_t* tb = tiles + (long long)b * th * tw * 3;
        const int sc = swap ? 2 - c : c;
        const int s0 = tb[((long long)ay.x * tw + ax.x) * 3 + sc] * ax.z + tb[((long long)ay.x * tw + ax.y) * 3 + sc] * ax.w;
        const int s1 = tb[((long long)ay.y * tw + ax.x) * 3 + sc] * ax.z + tb[((long long)ay.y * tw + ax.y) * 3 + sc] * ax.w;
        const int u = (((ay.z * (s0 >> 4)) >> 16) + ((ay.w * (s1 >> 4)) >> 16) + 2) >> 2;
        v = ((float)u - nc.mean[c]) * nc.istd[c];
      }
    }
    pl[tl * 49 + k] = v;
  }
  __syncthreads();
  const int tl = tid >> 3, cg = tid & 7;
  float acc[12];
#pragma unroll
  for (int j = 0; j < 12; ++j) acc[j] = 0.f;
  for (int k = 0; k < 48; ++k) {
    float x = pl[tl * 49 + k];
#pragma unroll
    for (int j = 0; j < 12; ++j) acc[j] = fmaf(x, wl[k * 96 + cg + 8 * j], acc[j]);
  }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < 12; ++j) { acc[j] += bias[cg + 8 * j]; sum += acc[j]; }
  sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4);
  const float mean = sum * (1.0f / 96.0f);
  float var = 0.f;
#pragma unroll
  for (int j = 0; j < 12; ++j) { float d = acc[j] - mean; var = fmaf(d, d, var); }
  var += __shfl_xor(var, 1); var += __shfl_xor(var, 2); var += __shfl_xor(var, 4);
  const float rstd = 1.0f / sqrtf(var * (1.0f / 96.0f) + 1e-5f);
  const int t = t0 + tl;
  if (t < nTok) {
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      int c = cg + 8 * j;
      tok[(long long)t * 96 + c] = (acc[j] - mean) * rstd * g[c] + beta[c];
    }
  }
}

int launch_patch_embed_tiles(const uint8_t* tiles, int B, int th, int tw, int Hn, int Wn, int Hv, int Wv, const int* xtab, const int* ytab, int swap,
                             const float* mean_istd, const float* w, const float* b, const float* g, const float* beta, float* tok, hipStream_t s) {
  ProfScope ps("patch_embed", 2.0 * 48 * 96 * B * (Hn / 4) * (Wn / 4), (double)B * 3.0 * th * tw + 4.0 * 96 * B * (Hn / 4) * (Wn / 4), s);
  NormConst nc;
  for (int i = 0; i < 3; ++i) { nc.mean[i] = mean_istd[i]; nc.istd[i] = mean_istd[3 + i]; }
  int nTok = B * (Hn / 4) * (Wn / 4);
  hipLaunchKernelGGL(patch_embed_tiles_kernel, dim3(cdiv(nTok, 32)), dim3(256), 0, s, tiles, th, tw, Hv, Wv, (const int4*)xtab, (const int4*)ytab, swap, nc,
                     w, b, g, beta, tok, nTok, Hn, Wn);
  return hipGetLastError() == hipSuccess ? 0 : NUHTC_E_HIP;
}

int launch_patch_embed(const float* img, const float* w, const float* b, const float* g, const float* beta, float* tok, int B,
                       int Hn, int Wn, hipStream_t s) {
  ProfScope ps("patch_embed", 2.0 * 48 * 96 * B * (Hn / 4) * (Wn / 4), 4.0 * B * Hn * Wn * 3 + 4.0 * 96 * B * (Hn / 4) * (Wn / 4), s);
  int nTok = B * (Hn / 4) * (Wn / 4);
  hipLaunchKernelGGL(patch_embed_kernel, dim3(cdiv(nTok, 32)), dim3(256), 0, s, img, w, b, g, beta, tok, nTok, Hn, Wn);
  return hipGetLastError() == hipSuccess ? 0 : NUHTC_E_HIP;
}

// ----------------------------------------------------------------------------- LayerNorm (+ row gather)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <int NV>   // NV = ceil(C/256) 16-byte chunks per lane (C % 4 == 0): one row per wave
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const int* __restrict__ src_map,
                                                        const int* __restrict__ dst_map, const float* __restrict__ g,
                                                        const float* __restrict__ b, float* __restrict__ y,
                                                        float* __restrict__ pad_dst, const float* __restrict__ pad_val,
                                                        int rows, int C) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int C4 = C >> 2;
  long long src = src_map ? src_map[row] : row;
  if (src < 0) {
    if (pad_dst) {   // padding row of a window: its QKV row is the bias
      v4f* pr = reinterpret_cast<v4f*>(pad_dst + row * 3 * C);
      const v4f* pv = reinterpret_cast<const v4f*>(pad_val);
      for (int c = lane; c < 3 * C4; c += 64) pr[c] = pv[c];
    } else {
      v4f* yr = reinterpret_cast<v4f*>(y + row * C);
#pragma unroll
      for (int j = 0; j < NV; ++j) { int c = lane + 64 * j; if (c < C4) yr[c] = (v4f){0.f, 0.f, 0.f, 0.f}; }
    }
    return;
  }
  v4f* yr = reinterpret_cast<v4f*>(y + (dst_map ? (long long)dst_map[row] : row) * C);
  const v4f* xr = reinterpret_cast<const v4f*>(x + src * C);
  v4f v[NV];
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    int c = lane + 64 * j;
    v[j] = c < C4 ? xr[c] : (v4f){0.f, 0.f, 0.f, 0.f};
    sum += (v[j].x + v[j].y) + (v[j].z + v[j].w);
  }
  const float mean = wave_sum(sum) / (float)C;
  float var = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    int c = lane + 64 * j;
    if (c < C4) {
      v[j] -= mean;
      var = fmaf(v[j].x, v[j].x, fmaf(v[j].y, v[j].y, fmaf(v[j].z, v[j].z, fmaf(v[j].w, v[j].w, var))));
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(var) / (float)C + 1e-5f);
  const v4f* g4 = reinterpret_cast<const v4f*>(g);
  const v4f* b4 = reinterpret_cast<const v4f*>(b);
#pragma unroll
  for (int j = 0; j < NV; ++j) { int c = lane + 64 * j; if (c < C4) yr[c] = v[j] * rstd * g4[c] + b4[c]; }
}

// C = 96 (stage 0, the most rows): a 32-lane half-wave normalises LN96_R rows, 24 lanes x one 16-byte load per row.  The map
// entries of the LN96_R rows are loaded together and then the rows themselves, so four 384-byte rows per half-wave are in flight
// instead of one (with one row per half-wave and the row load waiting for its map entry the kernel ran at 3.5 TB/s against
// 6.3 TB/s for the unmapped form).  Per-row arithmetic and summation order are unchanged.
constexpr int LN96_R = 4;
__global__ __launch_bounds__(256) void layernorm96_kernel(const float* __restrict__ x, const int* __restrict__ src_map,
                                                          const int* __restrict__ dst_map, const float* __restrict__ g,
                                                          const float* __restrict__ b, float* __restrict__ y,
                                                          float* __restrict__ pad_dst, const float* __restrict__ pad_val,
                                                          int rows) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const int l = threadIdx.x & 31;
  const long long row0 = ((long long)blockIdx.x * 8 + (threadIdx.x >> 5)) * LN96_R;
  if (row0 >= rows) return;
  const bool act = l < 24;
  long long src[LN96_R], dst[LN96_R];
#pragma unroll
  for (int k = 0; k < LN96_R; ++k) {
    const long long row = row0 + k < rows ? row0 + k : rows - 1;
    src[k] = src_map ? src_map[row] : row;
    dst[k] = dst_map ? (long long)dst_map[row] : row;
  }
  v4f v[LN96_R];
#pragma unroll
  for (int k = 0; k < LN96_R; ++k)
    v[k] = (act && src[k] >= 0) ? reinterpret_cast<const v4f*>(x + src[k] * 96)[l] : (v4f){0.f, 0.f, 0.f, 0.f};
  const v4f gg = act ? reinterpret_cast<const v4f*>(g)[l] : (v4f){0.f, 0.f, 0.f, 0.f};
  const v4f bb = act ? reinterpret_cast<const v4f*>(b)[l] : (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < LN96_R; ++k) {
    const long long row = row0 + k;
    if (row >= rows) break;
    if (src[k] < 0) {
      if (pad_dst) {   // padding row of a window: its QKV row (288 floats) is the bias
        v4f* pr = reinterpret_cast<v4f*>(pad_dst + row * 288);
        const v4f* pv = reinterpret_cast<const v4f*>(pad_val);
        for (int c = l; c < 72; c += 32) pr[c] = pv[c];
      } else if (act) {
        reinterpret_cast<v4f*>(y + row * 96)[l] = (v4f){0.f, 0.f, 0.f, 0.f};
      }
      continue;
    }
    float sum = (v[k].x + v[k].y) + (v[k].z + v[k].w);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / 96.0f;
    v4f d = act ? v[k] - mean : (v4f){0.f, 0.f, 0.f, 0.f};
    float var = fmaf(d.x, d.x, fmaf(d.y, d.y, fmaf(d.z, d.z, d.w * d.w)));
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) var += __shfl_xor(var, o);
    const float rstd = 1.0f / sqrtf(var / 96.0f + 1e-5f);
    if (act) reinterpret_cast<v4f*>(y + dst[k] * 96)[l] = d * rstd * gg + bb;
  }
}

static int layernorm_any(const float* x, const int* src_map, const int* dst_map, const float* g, const float* b, float* y,
                         float* pad_dst, const float* pad_val, int rows, int C, hipStream_t s) {
  if (rows <= 0) return 0;
  if (C == 96) {
    hipLaunchKernelGGL(layernorm96_kernel, dim3(cdiv(rows, 8 * LN96_R)), dim3(256), 0, s, x, src_map, dst_map, g, b, y, pad_dst, pad_val, rows);
    return hipGetLastError() == hipSuccess ? 0 : NUHTC_E_HIP;
  }
  dim3 grid(cdiv(rows, 4)), blk(256);
  if (C % 4 != 0) return NUHTC_E_INVALID;
  int nv = cdiv(C, 256);
  if (nv <= 1) hipLaunchKernelGGL(layernorm_kernel<1>, grid, blk, 0, s, x, src_map, dst_map, g, b, y, pad_dst, pad_val, rows, C);
  else if (nv <= 2) hipLaunchKernelGGL(layernorm_kernel<2>, grid, blk, 0, s, x, src_map, dst_map, g, b, y, pad_dst, pad_val, rows, C);
  else if (nv <= 3) hipLaunchKernelGGL(layernorm_kernel<3>, grid, blk, 0, s, x, src_map, dst_map, g, b, y, pad_dst, pad_val, rows, C);
  else return NUHTC_E_INVALID;
  return hipGetLastError() == hipSuccess ? 0 : NUHTC_E_HIP;
}

int launch_layernorm(const float* x, const int* src_map, const float* g, const float* b, float* y, int rows, int C, hipStream_t s) {
  { static const int& skip_ = dev_knob_ref("SKIP", 0); if (skip_ & 2) return 0; }   // dev: ablation of the step (tools/dev/r04_ablate.py)
  ProfScope ps("layernorm", 0, 8.0 * rows * C, s);
  return layernorm_any(x, src_map, nullptr, g, b, y, nullptr, nullptr, rows, C, s);
}

int launch_layernorm_windows(const float* x, const int* src_map, const int* dst_map, const float* g, const float* b, float* y,
                             float* pad_dst, const float* pad_val, int rows, int C, hipStream_t s) {
  { static const int& skip_ = dev_knob_ref("SKIP", 0); if (skip_ & 2) return 0; }   // dev: ablation of the step (tools/dev/r04_ablate.py)
  ProfScope ps("layernorm", 0, 8.0 * rows * C, s);
  if (!src_map || !dst_map || !pad_dst || !pad_val) return NUHTC_E_INVALID;
  return layernorm_any(x, src_map, dst_map, g, b, y, pad_dst, pad_val, rows, C, s);
}

// ---- LayerNorm statistics only (round 5): the norms of Swin stages 2-4 ride in the A path of the linear that consumes them
// (gemm.hip, A_LN).  In the engine the statistics are left by the epilogue of the GEMM that produced the tensor; this kernel is the
// stand-alone form (nuhtc_op_ln_gemm, dev fallback LN_IN_A=1): it reads each row once and leaves 8 bytes -- the mean and the sum of
// squared deviations, computed with the loads, the two passes and the summation order of layernorm_kernel.  One wave per row.
template <int NV>
__global__ __launch_bounds__(256) void ln_stats_kernel(const float* __restrict__ x, float* __restrict__ stats, int rows, int C) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int C4 = C >> 2;
  if (row >= rows) return;
  const v4f* xr = reinterpret_cast<const v4f*>(x + row * C);
  v4f v[NV];
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    int c = lane + 64 * j;
    v[j] = c < C4 ? xr[c] : (v4f){0.f, 0.f, 0.f, 0.f};
    sum += (v[j].x + v[j].y) + (v[j].z + v[j].w);
  }
  const float mean = wave_sum(sum) / (float)C;
  float var = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    int c = lane + 64 * j;
    if (c < C4) {
      v[j] -= mean;
      var = fmaf(v[j].x, v[j].x, fmaf(v[j].y, v[j].y, fmaf(v[j].z, v[j].z, fmaf(v[j].w, v[j].w, var))));
    }
  }
  const float m2 = wave_sum(var);
  if (lane == 0) *reinterpret_cast<float2*>(stats + 2 * row) = make_float2(mean, m2);
}

int launch_ln_stats(const float* x, float* stats, int rows, int C, hipStream_t s) {
  { static const int& skip_ = dev_knob_ref("SKIP", 0); if (skip_ & 2) return 0; }   // dev: ablation of the step (tools/dev/r04_ablate.py)
  if (rows <= 0) return 0;
  if (C % 4 != 0) return NUHTC_E_INVALID;
  ProfScope ps("layernorm", 0, 4.0 * rows * C + 8.0 * rows, s);
  dim3 grid(cdiv(rows, 4)), blk(256);
  const int nv = cdiv(C, 256);
  if (nv <= 1) hipLaunchKernelGGL(ln_stats_kernel<1>, grid, blk, 0, s, x, stats, rows, C);
  else if (nv <= 2) hipLaunchKernelGGL(ln_stats_kernel<2>, grid, blk, 0, s, x, stats, rows, C);
  else if (nv <= 3) hipLaunchKernelGGL(ln_stats_kernel<3>, grid, blk, 0, s, x, stats, rows, C);
  else return NUHTC_E_INVALID;
  return hipGetLastError() == hipSuccess ? 0 : NUHTC_E_HIP;
}

// ----------------------------------------------------------------------------- PatchMerging gather + LN(4C)
template <int NV>   // NV = 4C/64
__global__ __launch_bounds__(256) void merge_ln_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                       const float* __restrict__ b, float* __restrict__ y, int rows, int H, int W,
                                                       int C) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int H2 = H >> 1, W2 = W >> 1;
  int x2 = row % W2, y2 = (row / W2) % H2;
  long long bb = row / ((long long)W2 * H2);
  const int C4 = 4 * C;
  float v[NV];
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    int k = lane + 64 * j;            // k = q*C + c, q = kh*2 + kw
    int q = k / C, c = k - q * C;
    v[j] = x[((bb * H + 2 * y2 + (q >> 1)) * W + 2 * x2 + (q & 1)) * C + c];
    sum += v[j];
  }
  const float mean = wave_sum(sum) / (float)C4;
  float var = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) { float d = v[j] - mean; var = fmaf(d, d, var); }
  const float rstd = 1.0f / sqrtf(wave_sum(var) / (float)C4 + 1e-5f);
  float* yr = y + row * C4;
#pragma unroll
  for (int j = 0; j < NV; ++j) { int k = lane + 64 * j; yr[k] = (v[j] - mean) * rstd * g[k] + b[k]; }
}

int launch_merge_ln(const float* x, const float* g, const float* b, float* y, int B, int H, int W, int C, hipStream_t s) {
  { static const int& skip_ = dev_knob_ref("SKIP", 0); if (skip_ & 2) return 0; }   // dev: ablation of the step (tools/dev/r04_ablate.py)
  ProfScope ps("merge_ln", 0, 8.0 * B * H * W * C, s);
  int rows = B * (H / 2) * (W / 2);
  dim3 grid(cdiv(rows, 4)), blk(256);
  int nv = 4 * C / 64;
  if (4 * C % 64) return NUHTC_E_INVALID;
  if (nv == 6) hipLaunchKernelGGL(merge_ln_kernel<6>, grid, blk, 0, s, x, g, b, y, rows, H, W, C);
  else if (nv == 12) hipLaunchKernelGGL(merge_ln_kernel<12>, grid, blk, 0, s, x, g, b, y, rows, H, W, C);
  else if (nv == 24) hipLaunchKernelGGL(merge_ln_kernel<24>, grid, blk, 0, s, x, g, b, y, rows, H, W, C);
  else return NUHTC_E_INVALID;
  return hipGetLastError() == hipSuccess ? 0 : NUHTC_E_HIP;
}

// ----------------------------------------------------------------------------- window attention
// ---- one wave per (window, head), no LDS, no barriers ---------------------------------------------------
// Sᵀ = K·(s·Q)ᵀ and Oᵀ = Vᵀ·Pᵀ on v_mfma_f32_32x32x2_f32 with the 49 tokens padded to 2 x 32.  Computing the transposed
// products puts the query on the lane and the keys in the accumulator registers, so (a) the softmax over keys is a
// within-lane reduction plus one lane^32 exchange, and (b) the probability tile is already in B-operand position for the
// second product (it sums over the accumulator's row index): no data movement between the two GEMMs.  Q/K fragments are
// 64 contiguous bytes per lane read straight from the qkv rows; V is read one dword per MFMA step (a full 128-byte row per
// half-wave).  biasT is the relative-position bias transposed to [head][key][query] so lanes read it contiguously.

__global__ __launch_bounds__(256, 3) void window_attn_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ biasP,
                                                               const float* __restrict__ maskP, const int* __restrict__ mask_any,
                                                               const int* __restrict__ out_map, float* __restrict__ out, int nPairs,
                                                               int nWperImg, int C, int nH) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pair = blockIdx.x * 4 + wave;
  if (pair >= nPairs) return;
  const int win = pair / nH, head = pair - win * nH;
  const int l32 = lane & 31, half = lane >> 5;
  const long long ld = 3LL * C;
  const float* base = qkv + (long long)win * WS2 * ld + head * HEAD_DIM;
  // additive score terms, packed for the lanes ([ti][q][lane][4], engine.hip pack_attn_terms): the q-th 16-byte load of a wave is 1 KB of contiguous memory.
  // Only the windows on the shifted image's last row / column of windows carry a non-zero shift mask.
  // dev probes of round 6 (wrong results): 16 = K / Q / V taken as if they arrived already split by the QKV launch (their 44 of the 68 operand
  // splits per pair replaced by bit moves; the 24 splits of the probabilities stay); 32 = half as many K / Q / V bytes again are loaded (three
  // bf16 planes are 6 bytes per element against 4): 16 + 32 together = what a producer-split QKV image could give this kernel at most
#if defined(NUHTC_ATTN_PROBE) && (NUHTC_ATTN_PROBE & 16)
#define ATTN_SPLIT_IN(P_, d_, x_, y_) { (P_)[0][d_] = __float_as_uint(x_); (P_)[1][d_] = __float_as_uint(y_); (P_)[2][d_] = __float_as_uint(x_) ^ __float_as_uint(y_); }
#else
#define ATTN_SPLIT_IN(P_, d_, x_, y_) NUHTC_SPLIT3_INTO(P_, d_, x_, y_)
#endif
  const float* bP = biasP + (long long)head * 4096 + (half * 32 + l32) * 4;                      // + ti * 2048
  const int wimg = win % nWperImg;
  const float* mP = (maskP && mask_any[wimg]) ? maskP + (long long)wimg * 4096 + (half * 32 + l32) * 4 : nullptr;
  const float scale = 0.17677669529663687f;   // 32^-0.5

  // K fragments of both key tiles: lane (j, half) holds K[tj*32 + j][half*16 .. +16)
  v4f kf[2][4];
#pragma unroll
  for (int tj = 0; tj < 2; ++tj) {
    const int j = min(tj * 32 + l32, WS2 - 1);
    const v4f* kp = reinterpret_cast<const v4f*>(base + j * ld + C + half * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) kf[tj][q] = kp[q];
  }
  // V operand of the second product, shared by both query tiles: lane (d = l32, half), one dword per MFMA step
  // (registers r >= NR1 of the second key tile hold keys 49..63 in both half-waves: masked out, never loaded or multiplied)
  constexpr int NR1 = 9;
  float vv[2][16];
#pragma unroll
  for (int tj = 0; tj < 2; ++tj)
#pragma unroll
    for (int r = 0; r < (tj ? NR1 : 16); ++r) {
      const int j = min(tj * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, WS2 - 1);
      vv[tj][r] = base[j * ld + 2 * C + l32];
    }
#pragma unroll 1
  for (int ti = 0; ti < 2; ++ti) {
    const int i = min(ti * 32 + l32, WS2 - 1);
    // everything this query tile reads is requested before the first MFMA: the Q fragments, and the bias (+ shift mask)
    // terms, which do not depend on the scores and arrive behind the 32 MFMAs of the first product
    v4f qf[4];
    {
      const v4f* qp = reinterpret_cast<const v4f*>(base + i * ld + half * 16);
#pragma unroll
      for (int q = 0; q < 4; ++q) qf[q] = qp[q];
    }
    float bb[2][16];
    {
      const v4f* bp = reinterpret_cast<const v4f*>(bP + ti * 2048);
      v4f t4[7];
#pragma unroll
      for (int q = 0; q < 7; ++q) t4[q] = bp[q * 64];          // registers 0..15 of key tile 0, 0..11 of key tile 1 (NR1 = 9 are used)
      if (mP) {
        const v4f* mp = reinterpret_cast<const v4f*>(mP + ti * 2048);
#pragma unroll
        for (int q = 0; q < 7; ++q) t4[q] += mp[q * 64];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) { bb[0][4 * q] = t4[q].x; bb[0][4 * q + 1] = t4[q].y; bb[0][4 * q + 2] = t4[q].z; bb[0][4 * q + 3] = t4[q].w; }
#pragma unroll
      for (int q = 0; q < 3; ++q) { bb[1][4 * q] = t4[4 + q].x; bb[1][4 * q + 1] = t4[4 + q].y; bb[1][4 * q + 2] = t4[4 + q].z; bb[1][4 * q + 3] = t4[4 + q].w; }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) qf[q] *= scale;
    f32x16 st[2];
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[tj][r] = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        st[tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[tj][q].x, qf[q].x, st[tj], 0, 0, 0);
        st[tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[tj][q].y, qf[q].y, st[tj], 0, 0, 0);
        st[tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[tj][q].z, qf[q].z, st[tj], 0, 0, 0);
        st[tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[tj][q].w, qf[q].w, st[tj], 0, 0, 0);
      }
    }
    // + relative-position bias (+ shift mask), keys >= 49 masked out; softmax over the keys of query i
    float mx = -3.0e38f;
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int r = 0; r < (tj ? NR1 : 16); ++r) {
        const int j = tj * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = st[tj][r] + bb[tj][r];
        v = j < WS2 ? v : -3.0e38f;
        st[tj][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int r = 0; r < (tj ? NR1 : 16); ++r) {
        const float e = __expf(st[tj][r] - mx);
        st[tj][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 32);
    const float rsum = 1.0f / sum;
    // Oᵀ[d][i] = sum_j V[j][d] * P[i][j]: A = V rows (lane = d), B = the probability registers themselves
    f32x16 ot;
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[r] = 0.f;
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int r = 0; r < (tj ? NR1 : 16); ++r) ot = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[tj][r], st[tj][r] * rsum, ot, 0, 0, 0);
    long long orow = (long long)win * WS2 + ti * 32 + l32;
    if (ti * 32 + l32 < WS2 && out_map) orow = out_map[orow];
    if (ti * 32 + l32 < WS2 && orow >= 0) {
      float* op = out + orow * C + head * HEAD_DIM + 4 * half;
#pragma unroll
      for (int g = 0; g < 4; ++g)   // registers 4g..4g+3 are d = 8g + 4*half + 0..3
        *reinterpret_cast<v4f*>(op + 8 * g) = (v4f){ot[4 * g], ot[4 * g + 1], ot[4 * g + 2], ot[4 * g + 3]};
    }
  }
}

// ---- the same attention on the bf16 matrix pipe with exactly split operands (the default pipe, NUHTC_PIPE_BF16_SPLIT) -------------
// Same arrangement (one wave per (window, head), transposed products, query on the lane, probability registers = B operand of the
// second product); every fp32 operand is split into three bf16 planes and a 32 x 32 x 16 product step is six v_mfma_f32_32x32x16_bf16
// (split_math.h: what gemm_split_kernel does): 84 MFMAs of 32 cycles per (window, head) instead of 114 of 64.
//   S^T = K (sQ)^T: head_dim 32 = two k-steps; lane (j, half) element e of k-step s is channel 16 half + 8 s + e -- the 64 contiguous
//   bytes of a K / Q row the fp32 kernel loads, so both operands keep their loads.
//   O^T = V^T P^T: the keys are the k axis; accumulator register r of lane (i, half) holds key (r & 3) + 8 (r >> 2) + 4 half of its key
//   tile, so registers 8 s' .. 8 s' + 7 of a probability tile ARE the 8 k elements of k-step s' when the V operand is loaded in the same
//   key order (it always was).  Keys 0..47 are three such k-steps; key 48, alone in the fourth, is added by 16 fp32 FMAs per lane
//   (V[48] broadcast, P[i][48] from the lane that holds it) instead of six MFMAs and two splits over one key.
__global__ __launch_bounds__(256, 3) void window_attn_split_kernel(const float* __restrict__ qkv, const float* __restrict__ biasP,
                                                                const float* __restrict__ maskP, const int* __restrict__ mask_any,
                                                                const int* __restrict__ out_map, float* __restrict__ out, int nPairs,
                                                                int nWperImg, int C, int nH, const unsigned long long* __restrict__ padbits,
                                                                int bias_row) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pair = blockIdx.x * 4 + wave;
  if (pair >= nPairs) return;
  const int win = pair / nH, head = pair - win * nH;
  const int l32 = lane & 31, half = lane >> 5;
  const unsigned ld = 3u * C;
  const int wimg = win % nWperImg;
  // Padding rows of the window (tokens F.pad added, swin.py:341-343) all hold the QKV bias: they are never read -- bit j of `pad` set ->
  // row `bias_row` (one row, in L1 after its first use) stands in for row j.  Everything here is wave-uniform (scalar registers).
  const unsigned long long pad = padbits ? padbits[wimg] : 0ull;
  const unsigned row0 = (unsigned)win * WS2;
  const float* base = qkv + head * HEAD_DIM;             // + ATTN_ROW(j): element offset of row j of this window (one v_mad_u64_u32: any image size)
#define ATTN_ROW(j_) ((unsigned long long)(((pad >> (j_)) & 1ull) ? (unsigned)bias_row : row0 + (unsigned)(j_)) * (unsigned long long)ld)
  const float* bP = biasP + (long long)head * 4096 + (half * 32 + l32) * 4;                      // + ti * 2048
  const float* mP = (maskP && mask_any[wimg]) ? maskP + (long long)wimg * 4096 + (half * 32 + l32) * 4 : nullptr;
  const float scale = 0.17677669529663687f;   // 32^-0.5

  // Every row the (window, head) pair needs -- K and Q of both tiles, V -- is requested up front, before the first split: the kernel is
  // bound by the latency of these lane-per-row loads, and with Q of the second tile and V in flight beside K the wave waits once
  // instead of three times (0.735 -> 0.67 ms per step; a persistent form that also prefetches the NEXT pair needs 256 registers,
  // spills and runs at two waves per SIMD: 0.86 ms, not kept)
  // The operands want a row per lane (lane (j, half) = the 64 bytes 16 half .. of row j), which as a LOAD is 64 separate requests per
  // instruction.  The rows are therefore fetched the way memory likes it -- instruction q of a 32-row tile covers rows 8q .. 8q+7, eight
  // consecutive lanes one 128-byte row: 16 requests of 64 bytes -- and each tile is turned into the operand layout through a private
  // 4.5 KB slice of LDS (4 ds_write_b128 + 4 ds_read_b128, 144-byte row pitch: both conflict-free; DS operations of a wave execute in
  // order, no barrier).  The texture path's request rate, not HBM, was what the launches waited for (tools/dev/r04_attn_probe.sh).
  __shared__ __attribute__((aligned(16))) float tbuf[4][32 * 36];
  float* tb = tbuf[wave];
  const int lr = lane >> 3, lc = lane & 7;
  v4f kfa[2][4], qfa[2][4];
#pragma unroll
  for (int tj = 0; tj < 2; ++tj) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = min(tj * 32 + 8 * q + lr, WS2 - 1);
#if defined(NUHTC_ATTN_PROBE) && (NUHTC_ATTN_PROBE & 4)     // dev probe (wrong results): no K / Q loads
      kfa[tj][q] = (v4f){0.01f * lane, 0.02f, 0.03f * q, 0.04f}; qfa[tj][q] = (v4f){0.02f, 0.01f * lane, 0.01f, 0.03f * q};
#else
      const unsigned long long ro = ATTN_ROW(j);
      kfa[tj][q] = *reinterpret_cast<const v4f*>(base + ro + C + lc * 4);
      qfa[tj][q] = *reinterpret_cast<const v4f*>(base + ro + lc * 4);
#endif
    }
  }
#if defined(NUHTC_ATTN_PROBE) && (NUHTC_ATTN_PROBE & 32)    // dev probe: + 50 % K / Q / V bytes (the V third of the same rows, every other row group; folded into K so that it stays)
  {
    v4f ex[2][4];
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j = min(tj * 32 + 8 * q + lr, WS2 - 1);
        ex[tj][q] = *reinterpret_cast<const v4f*>(base + ATTN_ROW(j) + 2 * C + lc * 4);
      }
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int q = 0; q < 4; ++q) kfa[tj][q] += ex[tj][q] * 1e-30f;
  }
#endif
  // raw_[q] (rows 8q + lr, chunk lc) -> out_[p] (row l32, chunk 4 half + p)
#define ATTN_TO_OPERAND(raw_, out_)                                                                            \
  {                                                                                                            \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) *reinterpret_cast<v4f*>(tb + (8 * q + lr) * 36 + lc * 4) = raw_[q];   \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                                     \
    _Pragma("unroll") for (int p_ = 0; p_ < 4; ++p_) out_[p_] = *reinterpret_cast<const v4f*>(tb + l32 * 36 + (4 * half + p_) * 4);   \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                                     \
  }
  float vv[3][8];
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int r = 8 * (g & 1) + e;
#if defined(NUHTC_ATTN_PROBE) && (NUHTC_ATTN_PROBE & 2)     // dev probe: no V loads
      vv[g][e] = 0.001f * (lane + r);
#else
      vv[g][e] = base[ATTN_ROW((g >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) + 2 * C + l32];
#endif
    }
  __builtin_amdgcn_sched_barrier(0);
  // K planes of both key tiles: kp[tj][s][plane], lane (j, half) holds channels 16 half + 8 s + 0..7 of key tj*32 + j
  u32x4 kp[2][2][3];
#pragma unroll
  for (int tj = 0; tj < 2; ++tj) {
    v4f kf[4];
    ATTN_TO_OPERAND(kfa[tj], kf)
#pragma unroll
    for (int sk = 0; sk < 2; ++sk) {
      ATTN_SPLIT_IN(kp[tj][sk], 0, kf[2 * sk].x, kf[2 * sk].y)
      ATTN_SPLIT_IN(kp[tj][sk], 1, kf[2 * sk].z, kf[2 * sk].w)
      ATTN_SPLIT_IN(kp[tj][sk], 2, kf[2 * sk + 1].x, kf[2 * sk + 1].y)
      ATTN_SPLIT_IN(kp[tj][sk], 3, kf[2 * sk + 1].z, kf[2 * sk + 1].w)
    }
  }
  u32x4 vp[3][3];
#pragma unroll
  for (int ti = 0; ti < 2; ++ti) {
    v4f qf[4];
    ATTN_TO_OPERAND(qfa[ti], qf)
    // the additive score terms (relative-position bias + shift mask) are loaded straight into the score accumulators: the products are
    // accumulated on top of them (the Q split below covers the latency of these loads; 25 registers less than adding them afterwards)
    f32x16 st[2];
    {
      const v4f* bp = reinterpret_cast<const v4f*>(bP + ti * 2048);
      v4f t4[7];
#pragma unroll
#if defined(NUHTC_ATTN_PROBE) && (NUHTC_ATTN_PROBE & 1)     // dev probe: no bias loads
      for (int q = 0; q < 7; ++q) t4[q] = (v4f){0.f, 0.01f, 0.f, 0.02f};
#else
      for (int q = 0; q < 7; ++q) t4[q] = bp[q * 64];          // registers 0..15 of key tile 0, 0..11 of key tile 1 (9 are used)
#endif
#pragma unroll
      for (int q = 0; q < 4; ++q) { st[0][4 * q] = t4[q].x; st[0][4 * q + 1] = t4[q].y; st[0][4 * q + 2] = t4[q].z; st[0][4 * q + 3] = t4[q].w; }
#pragma unroll
      for (int q = 0; q < 3; ++q) { st[1][4 * q] = t4[4 + q].x; st[1][4 * q + 1] = t4[4 + q].y; st[1][4 * q + 2] = t4[4 + q].z; st[1][4 * q + 3] = t4[4 + q].w; }
#pragma unroll
      for (int r = 12; r < 16; ++r) st[1][r] = 0.f;
    }
    u32x4 qpl[2][3];
#pragma unroll
    for (int sk = 0; sk < 2; ++sk) {
      const v4f a = qf[2 * sk] * scale, b = qf[2 * sk + 1] * scale;
      ATTN_SPLIT_IN(qpl[sk], 0, a.x, a.y)
      ATTN_SPLIT_IN(qpl[sk], 1, a.z, a.w)
      ATTN_SPLIT_IN(qpl[sk], 2, b.x, b.y)
      ATTN_SPLIT_IN(qpl[sk], 3, b.z, b.w)
    }
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int sk = 0; sk < 2; ++sk) st[tj] = mfma_split6(kp[tj][sk], qpl[sk], st[tj]);
    if (mP) {      // the shift mask of the few windows that have one (last row / column of windows of a shifted block): after the products
      const v4f* mp = reinterpret_cast<const v4f*>(mP + ti * 2048);
      v4f m4[7];
#pragma unroll
      for (int q = 0; q < 7; ++q) m4[q] = mp[q * 64];
#pragma unroll
      for (int q = 0; q < 4; ++q) { st[0][4 * q] += m4[q].x; st[0][4 * q + 1] += m4[q].y; st[0][4 * q + 2] += m4[q].z; st[0][4 * q + 3] += m4[q].w; }
#pragma unroll
      for (int q = 0; q < 3; ++q) { st[1][4 * q] += m4[4 + q].x; st[1][4 * q + 1] += m4[4 + q].y; st[1][4 * q + 2] += m4[4 + q].z; st[1][4 * q + 3] += m4[4 + q].w; }
    }
    // keys >= 49 masked out; softmax over the keys of query i
    constexpr int NR1 = 9;
    float mx = -3.0e38f;
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int r = 0; r < (tj ? NR1 : 16); ++r) {
        const int j = tj * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = st[tj][r];
        v = j < WS2 ? v : -3.0e38f;
        st[tj][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int r = 0; r < (tj ? NR1 : 16); ++r) {
        const float e = __expf(st[tj][r] - mx);
        st[tj][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 32);
    const float rsum = 1.0f / sum;
    if (ti == 0) {
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        ATTN_SPLIT_IN(vp[g], 0, vv[g][0], vv[g][1])
        ATTN_SPLIT_IN(vp[g], 1, vv[g][2], vv[g][3])
        ATTN_SPLIT_IN(vp[g], 2, vv[g][4], vv[g][5])
        ATTN_SPLIT_IN(vp[g], 3, vv[g][6], vv[g][7])
      }
    }
    // O^T[d][i] = sum_j V[j][d] P[i][j]: A = V planes (lane = d), B = the planes of the probability registers themselves
    f32x16 ot;
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[r] = 0.f;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      u32x4 pp[3];
      const int tj = g >> 1, r0 = 8 * (g & 1);
      NUHTC_SPLIT3_INTO(pp, 0, st[tj][r0] * rsum, st[tj][r0 + 1] * rsum)
      NUHTC_SPLIT3_INTO(pp, 1, st[tj][r0 + 2] * rsum, st[tj][r0 + 3] * rsum)
      NUHTC_SPLIT3_INTO(pp, 2, st[tj][r0 + 4] * rsum, st[tj][r0 + 5] * rsum)
      NUHTC_SPLIT3_INTO(pp, 3, st[tj][r0 + 6] * rsum, st[tj][r0 + 7] * rsum)
      ot = mfma_split6(vp[g], pp, ot);
    }
    {   // key 48: register 8 of key tile 1 in the lower half-wave; V[48][d] for the 16 rows d = 8 g + 4 half + 0..3 of this lane's registers
      v4f v48[4];
#pragma unroll
#if defined(NUHTC_ATTN_PROBE) && (NUHTC_ATTN_PROBE & 2)
      for (int g = 0; g < 4; ++g) v48[g] = (v4f){0.1f, 0.2f, 0.3f, 0.4f};
#else
      for (int g = 0; g < 4; ++g) v48[g] = *reinterpret_cast<const v4f*>(base + ATTN_ROW(48) + 2 * C + 8 * g + 4 * half);
#endif
      const float p48 = __shfl(st[1][8], l32) * rsum;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        ot[4 * g] = fmaf(v48[g].x, p48, ot[4 * g]); ot[4 * g + 1] = fmaf(v48[g].y, p48, ot[4 * g + 1]);
        ot[4 * g + 2] = fmaf(v48[g].z, p48, ot[4 * g + 2]); ot[4 * g + 3] = fmaf(v48[g].w, p48, ot[4 * g + 3]);
      }
    }
    long long orow = (long long)win * WS2 + ti * 32 + l32;
    if (ti * 32 + l32 < WS2 && out_map) orow = out_map[orow];
#if defined(NUHTC_ATTN_PROBE) && (NUHTC_ATTN_PROBE & 8)     // dev probe: no output stores
    if (ti * 32 + l32 < WS2 && orow >= 0 && ot[0] == 12345.f) {
#else
    if (ti * 32 + l32 < WS2 && orow >= 0) {
#endif
      float* op = out + orow * C + head * HEAD_DIM + 4 * half;
#pragma unroll
      for (int g = 0; g < 4; ++g)   // registers 4g..4g+3 are d = 8g + 4*half + 0..3
        *reinterpret_cast<v4f*>(op + 8 * g) = (v4f){ot[4 * g], ot[4 * g + 1], ot[4 * g + 2], ot[4 * g + 3]};
    }
  }
}
#undef ATTN_TO_OPERAND
#undef ATTN_ROW
#undef ATTN_SPLIT_IN

int launch_window_attn(const float* qkv, const float* biasP, const float* maskP, const int* mask_any, const int* out_map, float* out,
                       int nWinTotal, int nWperImg, int C, int nH, int split_pipe, hipStream_t s, const unsigned long long* padbits, int bias_row) {
  { static const int& skip_ = dev_knob_ref("SKIP", 0); if (skip_ & 1) return 0; }   // dev: ablation of the step (tools/dev/r04_ablate.py)
  ProfScope ps(C == 96 ? "window_attn|c96" : C == 192 ? "window_attn|c192" : C == 384 ? "window_attn|c384" : C == 768 ? "window_attn|c768" : "window_attn",   // (tags group by the part before '|')
               4.0 * 49 * 49 * 32 * nWinTotal * nH, 16.0 * 49 * C * nWinTotal, s);
  int nPairs = nWinTotal * nH;
  if (nPairs <= 0) return 0;
  if (!biasP || (maskP && !mask_any) || (padbits && !split_pipe)) return NUHTC_E_INVALID;
  if (split_pipe)
    hipLaunchKernelGGL(window_attn_split_kernel, dim3(cdiv(nPairs, 4)), dim3(256), 0, s, qkv, biasP, maskP, mask_any, out_map, out, nPairs, nWperImg, C, nH, padbits, bias_row);
  else
    hipLaunchKernelGGL(window_attn_mfma_kernel, dim3(cdiv(nPairs, 4)), dim3(256), 0, s, qkv, biasP, maskP, mask_any, out_map, out, nPairs, nWperImg, C, nH);
  return hipGetLastError() == hipSuccess ? 0 : NUHTC_E_HIP;
}

const char* nuhtc_tu_probe_swin() {
#if defined(NUHTC_ATTN_PROBE) && NUHTC_ATTN_PROBE
  return "NUHTC_ATTN_PROBE";
#else
  return nullptr;
#endif
}
