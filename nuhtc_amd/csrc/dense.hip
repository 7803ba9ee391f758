// Small dense-head helpers (gfx950): semantic-head fusion, 64->1 pointwise conv, row norms, batched transpose.
#include "common.h"

// FusedSemanticHead fusion (mmdet/models/roi_heads/mask_heads/fused_semantic_head.py:97-104):
//   x = relu(L0(f0)) + sum_i relu(Li(bilinear_align_corners(f_i -> HxW)))
// The 1x1 convs Li are applied at the native resolution first (g_i = Li(f_i) incl. bias); a pointwise affine map
// commutes with bilinear interpolation (weights sum to 1), so relu(Li(up(f_i))) == relu(up(g_i)) up to rounding.
__device__ __forceinline__ float4 bilin_ac(const float* __restrict__ g, int b, int h, int w, int y, int x, int H, int W, int c4) {
  const float sy = h > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;   // at::area_pixel_compute_scale, align_corners=True
  const float sx = w > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
  const float fy = sy * y, fx = sx * x;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
  const float ly = fy - y0, lx = fx - x0, hy = 1.f - ly, hx = 1.f - lx;
  const float4* p = reinterpret_cast<const float4*>(g) + (long long)b * h * w * 16 + c4;
  const float4 v00 = p[(y0 * w + x0) * 16], v01 = p[(y0 * w + x1) * 16], v10 = p[(y1 * w + x0) * 16], v11 = p[(y1 * w + x1) * 16];
  float4 r;
  r.x = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
  r.y = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
  r.z = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
  r.w = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
  return r;
}

__global__ void sem_fuse_kernel(const float* __restrict__ g0, const float* __restrict__ g1, const float* __restrict__ g2,
                                const float* __restrict__ g3, float* __restrict__ out, int B, int H, int W) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // (b, y, x, c4)
  long long total = (long long)B * H * W * 16;
  if (idx >= total) return;
  int c4 = idx & 15;
  long long pix = idx >> 4;
  int x = pix % W, y = (pix / W) % H, b = pix / ((long long)W * H);
  float4 a = reinterpret_cast<const float4*>(g0)[idx];
  float4 r = make_float4(fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(a.z, 0.f), fmaxf(a.w, 0.f));
  const float* gs[3] = {g1, g2, g3};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    int h = H >> (i + 1), w = W >> (i + 1);
    float4 u = bilin_ac(gs[i], b, h, w, y, x, H, W, c4);
    r.x += fmaxf(u.x, 0.f); r.y += fmaxf(u.y, 0.f); r.z += fmaxf(u.z, 0.f); r.w += fmaxf(u.w, 0.f);
  }
  reinterpret_cast<float4*>(out)[idx] = r;
}

int launch_sem_fuse(const float* g0, const float* g1, const float* g2, const float* g3, float* out, int B, int H, int W,
                    hipStream_t s) {
  ProfScope ps("sem_fuse", 0, 4.0 * 64 * B * H * W * 2.33, s);
  long long total = (long long)B * H * W * 16;
  hipLaunchKernelGGL(sem_fuse_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, g0, g1, g2, g3, out, B, H, W);
  return hipGetLastError() == hipSuccess ? 0 : NUHTC_E_HIP;
}

// y[row] = act(dot(x[row, 0:64], w) + b);  one 16-lane group per row (float4 per lane), 4 rows per wave.  Grid-stride over the
// rows: with a device-side row count the launch is sized by capacity, and a workgroup per 16 rows of an (often 90 % empty)
// capacity spent more time on early exits than on the rows.
__global__ void conv1x1_n1_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                  float* __restrict__ y, long long rows, const int* __restrict__ rows_dev, int rows_mul, int sigmoid) {
  if (rows_dev) { long long rd = (long long)(*rows_dev) * rows_mul; rows = rd < rows ? rd : rows; }
  const int l = threadIdx.x & 15;
  const float4 ww = reinterpret_cast<const float4*>(w)[l];
  const float bias = b[0];
  const long long stride = ((long long)gridDim.x * blockDim.x) >> 4;
  // (all 16 lanes of a row group run the same trip count, so the shuffles below always see their partners)
  for (long long row = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4; row < rows; row += stride) {
    const float4 v = reinterpret_cast<const float4*>(x)[row * 16 + l];
    float acc = v.x * ww.x + v.y * ww.y + v.z * ww.z + v.w * ww.w;
    acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 4); acc += __shfl_xor(acc, 8);
    if (l == 0) {
      const float u = acc + bias;
      y[row] = sigmoid ? 1.0f / (1.0f + expf(-u)) : u;
    }
  }
}

int launch_conv1x1_n1(const float* x, const float* w, const float* b, float* y, int rows, int C, hipStream_t s) {
  if (C != 64) return NUHTC_E_INVALID;
  long long threads = (long long)rows * 16;
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(conv1x1_n1_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, x, w, b, y, (long long)rows,
                     (const int*)nullptr, 1, 0);
  return hipGetLastError() == hipSuccess ? 0 : NUHTC_E_HIP;
}

int launch_conv1x1_n1_dev(const float* x, const float* w, const float* b, float* y, int rows_cap, const int* rows_dev, int rows_mul,
                          int sigmoid, hipStream_t s) {
  long long threads = (long long)rows_cap * 16;
  if (rows_cap <= 0) return 0;
  const long long nb = (threads + 255) / 256;
  hipLaunchKernelGGL(conv1x1_n1_kernel, dim3((unsigned)(nb < 8192 ? nb : 8192)), dim3(256), 0, s, x, w, b, y, (long long)rows_cap,
                     rows_dev, rows_mul, sigmoid);
  return hipGetLastError() == hipSuccess ? 0 : NUHTC_E_HIP;
}

// inv[row] = 1 / max(||x_row||_2, 1e-8)   (torch cosine_similarity eps), C = 64, 16 lanes per row
__global__ void rownorm_inv_kernel(const float* __restrict__ x, float* __restrict__ inv, long long rows) {
  long long row = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int l = threadIdx.x & 15;
  float acc = 0.f;
  if (row < rows) {
    float4 v = reinterpret_cast<const float4*>(x)[row * 16 + l];
    acc = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 4); acc += __shfl_xor(acc, 8);
  if (row < rows && l == 0) inv[row] = 1.0f / fmaxf(sqrtf(acc), 1e-8f);
}

int launch_rownorm_inv(const float* x, float* inv, int rows, int C, hipStream_t s) {
  if (C != 64) return NUHTC_E_INVALID;
  long long threads = (long long)rows * 16;
  hipLaunchKernelGGL(rownorm_inv_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, x, inv, (long long)rows);
  return hipGetLastError() == hipSuccess ? 0 : NUHTC_E_HIP;
}

// y[b][c][r] = x[b][r][c]   (rows x cols -> cols x rows), 32x32 LDS tiles
__global__ void transpose_kernel(const float* __restrict__ x, float* __restrict__ y, int rows, int cols) {
  __shared__ float t[32][33];
  const float* xb = x + (long long)blockIdx.z * rows * cols;
  float* yb = y + (long long)blockIdx.z * rows * cols;
  int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: ty 0..7
  for (int j = ty; j < 32; j += 8) {
    int r = r0 + j, c = c0 + tx;
    t[j][tx] = (r < rows && c < cols) ? xb[(long long)r * cols + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    int c = c0 + j, r = r0 + tx;
    if (r < rows && c < cols) yb[(long long)c * rows + r] = t[tx][j];
  }
}

int launch_transpose(const float* x, float* y, int batch, int rows, int cols, hipStream_t s) {
  hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32), batch), dim3(256), 0, s, x, y, rows, cols);
  return hipGetLastError() == hipSuccess ? 0 : NUHTC_E_HIP;
}
