// RoI path of the HTC-lite cascade on the device (gfx950, wave64; compiled with -ffp-contract=off so that box
// arithmetic follows the reference's float32 mul/add sequence).
//   attn_pool_kernel   : AttentionRoIExtractor levels 2,3 — similarity-weighted global mean for every possible
//                        RoI centre of a level, computed once per image and shared by all cascade stages and
//                        the mask branch (nuhtc/models/roi_extractors_cus.py:220-238)
//   roi_feat_kernel    : fused RoI feature: RoIAlign(lvl0) + RoIAlign(lvl1) + G2 + G3 (+ semantic RoIAlign14,
//                        2x2 average-pooled for the 7x7 box head) — one wave per RoI, lane = channel, NHWC maps
//                        (roi_extractors_cus.py:194-259, nuhtc/models/htc_roi_head_cus.py:187-199,2322-2335;
//                         mmcv RoIAlign avg/aligned semantics as in oracle/ops_c.c)
//   bbox_tail_kernel   : NormedLinear cls + fc_reg + cascade refinement (normed_predictor.py:33-38,
//                        convfc_bbox_head.py:190-196, bbox_head.py:459-496)
//   det_candidates     : score ensemble, Seesaw activation, decode, threshold, (roi,class) expansion
//                        (htc_roi_head_cus.py:2283-2303, seesaw_loss.py:157-175, nuhtc/models/bbox_head.py:12-102)
//   paste_kernel       : _do_paste_mask / get_seg_masks (fcn_mask_head.py:229-307,344-412) -> bit-packed masks
//   tile_post_kernel   : tools/infer_wsi.py:510-531,60-84 margin/min-area filter + greedy mask-NMS (popcount IoU)
#include "roi.h"

__device__ __forceinline__ float wsum64(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ------------------------------------------------------------------------------------------- attention pooling table
// G[b, q, c] = mean_p( F[b,p,c] * (relu(cos(F[b,q], F[b,p]) - tau) + tau) ),  F: [B, HW, 64] (NHWC level map)
__global__ __launch_bounds__(256) void attn_pool_kernel(const float* __restrict__ F, float* __restrict__ G, int HW, float tau) {
  __shared__ float part[4][64];
  const int q = blockIdx.x, b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* Fb = F + (long long)b * HW * 64;
  const float qv = Fb[(long long)q * 64 + lane];
  const float qn = fmaxf(sqrtf(wsum64(qv * qv)), 1e-8f);
  const float qh = qv / qn;
  float acc = 0.f;
  for (int p = wave; p < HW; p += 4) {
    const float v = Fb[(long long)p * 64 + lane];
    const float pn = fmaxf(sqrtf(wsum64(v * v)), 1e-8f);
    const float cs = wsum64(qh * (v / pn));
    const float sim = fmaxf(cs - tau, 0.f) + tau;
    acc += v * sim;
  }
  part[wave][lane] = acc;
  __syncthreads();
  if (wave == 0) {
    float s = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
    G[((long long)b * HW + q) * 64 + lane] = s / (float)HW;
  }
}

// The same table as the reference computes it when its feature maps are on a CUDA device: AttentionRoIExtractor casts this branch to fp16
// there (`roi_dtype = torch.float16 if feats[0].is_cuda`, roi_extractors_cus.py:203,231), so every tensor operation of :231-237 rounds its
// result to fp16 (reductions accumulate in fp32 and round once): feat.to(fp16); cosine_similarity of torch 1.13 = sum(x1*x2) /
// sqrt(clamp_min(sum(x1*x1) * sum(x2*x2), eps^2)) with eps^2 = 1e-16 -> 0 in fp16; relu(cos - thres) + thres with the Python scalars applied in
// fp32; feat * sim; mean over the map.  The fp16 values are then added into the fp32 RoI features (:247).  nuhtc_config.att_pool_fp16 = 1
// selects it (default 0: the fp32 arithmetic of the reference's CPU path); accumulation order is this kernel's own.
__device__ __forceinline__ float rh(float x) { return (float)(_Float16)x; }      // round to nearest even fp16 (overflow -> inf), back to fp32
__global__ __launch_bounds__(256) void attn_pool_fp16_kernel(const float* __restrict__ F, float* __restrict__ G, int HW, float tau) {
  __shared__ float part[4][64];
  const int q = blockIdx.x, b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* Fb = F + (long long)b * HW * 64;
  const float qv = rh(Fb[(long long)q * 64 + lane]);
  const float w1 = rh(wsum64(rh(qv * qv)));
  float acc = 0.f;
  for (int p = wave; p < HW; p += 4) {
    const float v = rh(Fb[(long long)p * 64 + lane]);
    const float w12 = rh(wsum64(rh(qv * v)));
    const float w2 = rh(wsum64(rh(v * v)));
    const float n12 = rh(sqrtf(fmaxf(rh(w1 * w2), 0.f)));
    const float cs = rh(w12 / n12);
    const float sim = rh(fmaxf(rh(cs - tau), 0.f) + tau);     // (fmaxf drops a NaN cosine -- 0 / 0 -- the way torch's relu does not: see below)
    acc += rh(v * (cs != cs ? cs : sim));                       // NaN propagates as in the reference
  }
  part[wave][lane] = acc;
  __syncthreads();
  if (wave == 0) {
    const float s = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
    G[((long long)b * HW + q) * 64 + lane] = rh(s * (1.0f / (float)HW));
  }
}

int launch_attn_pool_fp16(const float* F, float* G, int B, int HW, float tau, hipStream_t s) {
  ProfScope ps("attn_pool_fp16", 4.0 * 64 * (double)HW * HW * B, 0, s);
  hipLaunchKernelGGL(attn_pool_fp16_kernel, dim3(HW, B), dim3(256), 0, s, F, G, HW, tau);
  return hipGetLastError() == hipSuccess ? 0 : NUHTC_E_HIP;
}

int launch_attn_pool(const float* F, float* G, int B, int HW, float tau, hipStream_t s) {
  ProfScope ps("attn_pool", 4.0 * 64 * (double)HW * HW * B, 0, s);
  hipLaunchKernelGGL(attn_pool_kernel, dim3(HW, B), dim3(256), 0, s, F, G, HW, tau);
  return hipGetLastError() == hipSuccess ? 0 : NUHTC_E_HIP;
}

// ------------------------------------------------------------------------------------------- RoI list assembly
// rois[b] = cat(cc_boxes[b], rpn_dets[b]) (htc_roi_head_cus.py:339), flattened over the batch (bbox2roi)
__global__ __launch_bounds__(256) void build_rois_kernel(const float* __restrict__ cc_boxes, const int* __restrict__ cc_counts, int cc_cap,
                                                         const float* __restrict__ rpn_dets, const int* __restrict__ rpn_counts, int rpn_cap,
                                                         const float* __restrict__ fixed, int n_fixed, float* __restrict__ rois,
                                                         int* __restrict__ roi_off, int* __restrict__ roi_cnt, int* __restrict__ total, int B) {
  __shared__ int off[257];
  const int tid = threadIdx.x;
  if (tid == 0) {
    int acc = 0;
    for (int b = 0; b < B; ++b) {
      int n = fixed ? n_fixed : (cc_boxes ? cc_counts[b] : 0) + rpn_counts[b];
      off[b] = acc;
      roi_off[b] = acc;
      roi_cnt[b] = n;
      acc += n;
    }
    off[B] = acc;
    *total = acc;
  }
  __syncthreads();
  for (int b = 0; b < B; ++b) {
    const int o = off[b], n = off[b + 1] - off[b];
    const int ncc = fixed ? 0 : (cc_boxes ? cc_counts[b] : 0);
    for (int j = tid; j < n; j += 256) {
      const float* src = fixed ? fixed + ((long long)b * n_fixed + j) * 4
                               : (j < ncc ? cc_boxes + ((long long)b * cc_cap + j) * 4 : rpn_dets + ((long long)b * rpn_cap + (j - ncc)) * 5);
      float* d = rois + (long long)(o + j) * 5;
      d[0] = (float)b; d[1] = src[0]; d[2] = src[1]; d[3] = src[2]; d[4] = src[3];
    }
  }
}

int launch_build_rois(const float* cc_boxes, const int* cc_counts, int cc_cap, const float* rpn_dets, const int* rpn_counts, int rpn_cap,
                      const float* fixed, int n_fixed, float* rois, int* roi_off, int* roi_cnt, int* total, int B, hipStream_t s) {
  if (B > 256) return NUHTC_E_INVALID;
  hipLaunchKernelGGL(build_rois_kernel, dim3(1), dim3(256), 0, s, cc_boxes, cc_counts, cc_cap, rpn_dets, rpn_counts, rpn_cap, fixed, n_fixed,
                     rois, roi_off, roi_cnt, total, B);
  return hipGetLastError() == hipSuccess ? 0 : NUHTC_E_HIP;
}

// ------------------------------------------------------------------------------------------- fused RoI features
struct Tap { int o00, o01, o10, o11; float w1, w2, w3, w4; };   // offsets in units of 64-channel pixels; weight 0 when invalid

// mmcv bilinear_interpolate pre-computation for one sample point (y, x) on an H x W map
__device__ __forceinline__ Tap make_tap(float y, float x, int H, int W) {
  Tap t;
  if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) {
    t.o00 = t.o01 = t.o10 = t.o11 = 0;
    t.w1 = t.w2 = t.w3 = t.w4 = 0.f;
    return t;
  }
  if (y <= 0.f) y = 0.f;
  if (x <= 0.f) x = 0.f;
  int yl = (int)y, xl = (int)x, yh, xh;
  if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
  if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
  const float ly = y - (float)yl, lx = x - (float)xl, hy = 1.0f - ly, hx = 1.0f - lx;
  t.o00 = yl * W + xl; t.o01 = yl * W + xh; t.o10 = yh * W + xl; t.o11 = yh * W + xh;
  t.w1 = hy * hx; t.w2 = hy * lx; t.w3 = ly * hx; t.w4 = ly * lx;
  return t;
}

// RoIAlign (avg, aligned) of one bin, all 64 channels of the wave; fb = map of image b (NHWC, 64 ch), lane = channel
__device__ __forceinline__ float roi_bin(const float* __restrict__ fb, int H, int W, float x1, float y1, float bw, float bh, int gw, int gh,
                                         int pw, int ph, int lane) {
  float acc = 0.f;
  for (int iy = 0; iy < gh; ++iy) {
    const float y = y1 + (float)ph * bh + ((float)iy + 0.5f) * bh / (float)gh;
    for (int ix = 0; ix < gw; ++ix) {
      const float x = x1 + (float)pw * bw + ((float)ix + 0.5f) * bw / (float)gw;
      const Tap t = make_tap(y, x, H, W);
      const float v1 = fb[(long long)t.o00 * 64 + lane], v2 = fb[(long long)t.o01 * 64 + lane];
      const float v3 = fb[(long long)t.o10 * 64 + lane], v4 = fb[(long long)t.o11 * 64 + lane];
      acc += t.w1 * v1 + t.w2 * v2 + t.w3 * v3 + t.w4 * v4;
    }
  }
  const int cnt = gh * gw;
  return acc / (float)(cnt > 1 ? cnt : 1);
}

struct RoiGeom { float x1, y1, bw, bh; int gw, gh; };
__device__ __forceinline__ RoiGeom roi_geom(const float* roi, float scale, int P, int sr) {
  RoiGeom g;
  g.x1 = roi[1] * scale - 0.5f;
  g.y1 = roi[2] * scale - 0.5f;
  const float x2 = roi[3] * scale - 0.5f, y2 = roi[4] * scale - 0.5f;
  const float rw = x2 - g.x1, rh = y2 - g.y1;
  g.bw = rw / (float)P;
  g.bh = rh / (float)P;
  g.gw = sr > 0 ? sr : (int)ceilf(rw / (float)P);
  g.gh = sr > 0 ? sr : (int)ceilf(rh / (float)P);
  return g;
}

// ---- LDS-staged variant for the 7x7 box-head features ------------------------------------------------------------
// Nuclei-sized RoIs touch at most an 8x8 pixel footprint on the stride-4 maps (5x5 on stride 8).  One wave per RoI
// copies the three footprints (FPN level 0, semantic embedding, FPN level 1) into LDS with direct global->LDS loads
// (one 256-byte pixel per instruction, no VGPR staging, all in flight at once), builds the per-axis bilinear tables
// of the regular sampling grid once (14 sample columns + 14 sample rows per map, computed by 28 lanes in parallel), and
// then evaluates the 49 bins from LDS with exactly the arithmetic of roi_bin() (same products, same summation order), so
// both paths are bit-identical.  RoIs with a larger footprint fall back to the global-memory path.
#define TP0 12
#define TP1 7
#define TS0 8      // small-tile variant of the LDS path
#define TS1 5
struct AxisEnt { int lo, hi; float l, h; };

// per-axis part of mmcv's bilinear_interpolate for sample coordinate c on an axis of `size` pixels
__device__ __forceinline__ AxisEnt axis_entry(float c, int size, bool& valid) {
  AxisEnt e;
  valid = !(c < -1.0f || c > (float)size);
  if (!valid) { e.lo = e.hi = 0; e.l = e.h = 0.f; return e; }
  if (c <= 0.f) c = 0.f;
  int lo = (int)c, hi;
  if (lo >= size - 1) { hi = lo = size - 1; c = (float)lo; } else hi = lo + 1;
  e.lo = lo; e.hi = hi;
  e.l = c - (float)lo;
  e.h = 1.0f - e.l;
  return e;
}

__device__ __forceinline__ int half_min(int v) {   // min over the 32-lane half this lane belongs to
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ int half_max(int v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}

struct LevelPlan { AxisEnt ent; int fx0, fy0, fw, fh; bool ok, empty; };

// lanes 0..n-1 build the x entries, lanes 32..32+n-1 the y entries of a P x P bin grid with g samples per bin and axis
// (band variant: only the y entries [yfirst, yfirst + ycount) take part, e.g. the sample rows of one output bin row; the
// footprint limits are then tpw x tph)
__device__ __forceinline__ LevelPlan plan_band(const RoiGeom& g, int P, int gsx, int gsy, int H, int W, int tpw, int tph, int yfirst, int ycount,
                                               int lane) {
  LevelPlan lp;
  const bool is_y = lane >= 32;
  const int idx = lane & 31;
  const int gsamp = is_y ? gsy : gsx;
  const int n = P * gsamp;
  const bool active = is_y ? (idx >= yfirst && idx < yfirst + ycount && idx < n) : idx < n;
  const int pb = active ? idx / gsamp : 0, is = active ? idx - pb * gsamp : 0;
  const float start = is_y ? g.y1 : g.x1, bs = is_y ? g.bh : g.bw;
  const float c = start + (float)pb * bs + ((float)is + 0.5f) * bs / (float)gsamp;   // same expression as roi_bin()
  bool valid;
  lp.ent = axis_entry(c, is_y ? H : W, valid);
  valid = valid && active;
  const int lo = half_min(valid ? lp.ent.lo : (1 << 30));
  const int hi = half_max(valid ? lp.ent.hi : -1);
  const int x_lo = __shfl(lo, 0), x_hi = __shfl(hi, 0), y_lo = __shfl(lo, 32), y_hi = __shfl(hi, 32);
  lp.empty = x_hi < 0 || y_hi < 0;
  lp.fx0 = lp.empty ? 0 : x_lo; lp.fy0 = lp.empty ? 0 : y_lo;
  lp.fw = lp.empty ? 0 : x_hi - x_lo + 1; lp.fh = lp.empty ? 0 : y_hi - y_lo + 1;
  lp.ok = lp.fw <= tpw && lp.fh <= tph;
  const int f0 = is_y ? lp.fy0 : lp.fx0;
  if (valid) { lp.ent.lo -= f0; lp.ent.hi -= f0; }   // invalid entries keep offset 0 with zero weights
  else { lp.ent.lo = lp.ent.hi = 0; lp.ent.l = lp.ent.h = 0.f; }
  return lp;
}
__device__ __forceinline__ LevelPlan plan_level(const RoiGeom& g, int P, int gsamp, int H, int W, int tp, int lane) {
  return plan_band(g, P, gsamp, gsamp, H, W, tp, tp, 0, 32, lane);
}

// the block's 4 waves copy the footprint with 16-byte global->LDS loads: one instruction moves 4 pixels (4 x 256 bytes; the
// LDS image is pixel-major, so the 64 lanes' 16-byte pieces are contiguous there, while every lane forms its own source
// address -- pixels of different footprint rows are not adjacent in the map)
__device__ __forceinline__ void stage_tile(const float* __restrict__ map, int H, int W, int b, const LevelPlan& lp, float* tile, int lane, int wave) {
  const int npx = lp.fh * lp.fw;
  const int sub = lane >> 4, c4 = (lane & 15) * 4;
  for (int p4 = wave * 4; p4 < npx; p4 += 16) {
    const int pp = p4 + sub;
    if (pp < npx) {
      const int yy = pp / lp.fw, xx = pp - yy * lp.fw;
      const float* src = map + (((long long)b * H + lp.fy0 + yy) * W + lp.fx0 + xx) * 64 + c4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(tile + p4 * 64), 16, 0, 0);
    }
  }
}

// one bin from the LDS image: identical arithmetic to roi_bin() (weights hy*hx.., sample sum iy-outer ix-inner, / count)
template <int G>
__device__ __forceinline__ float bin_lds(const float* tile, int fw, const AxisEnt* tx, const AxisEnt* ty, int pw, int ph, int lane) {
  float acc = 0.f;
#pragma unroll
  for (int iy = 0; iy < G; ++iy) {
    const AxisEnt ey = ty[ph * G + iy];
#pragma unroll
    for (int ix = 0; ix < G; ++ix) {
      const AxisEnt ex = tx[pw * G + ix];
      const float w1 = ey.h * ex.h, w2 = ey.h * ex.l, w3 = ey.l * ex.h, w4 = ey.l * ex.l;
      const float v1 = tile[(ey.lo * fw + ex.lo) * 64 + lane], v2 = tile[(ey.lo * fw + ex.hi) * 64 + lane];
      const float v3 = tile[(ey.hi * fw + ex.lo) * 64 + lane], v4 = tile[(ey.hi * fw + ex.hi) * 64 + lane];
      acc += w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
    }
  }
  return acc / (float)(G * G);
}

// Two channels per lane (packed fp32: v_pk_mul_f32 / v_pk_add_f32, ds_read_b64): the same operations in the same order as
// bin_lds() on each channel, at half the instruction count -- the RoI kernels are bound by VALU issue (address and weight
// arithmetic per tap), not by LDS or memory.  `cp2` = 2 * (lane & 31) is the lane's first channel; the two half-waves
// work on different bins, so the axis tables are read per lane.
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
template <int GX, int GY = GX>
__device__ __forceinline__ v2f bin_lds2(const float* tile, const AxisEnt* tx, const AxisEnt* ty, int pw, int ph) {
  v2f acc = {0.f, 0.f};
#pragma unroll
  for (int iy = 0; iy < GY; ++iy) {
    const AxisEnt ey = ty[ph * GY + iy];
#pragma unroll
    for (int ix = 0; ix < GX; ++ix) {
      const AxisEnt ex = tx[pw * GX + ix];
      const float w1 = ey.h * ex.h, w2 = ey.h * ex.l, w3 = ey.l * ex.h, w4 = ey.l * ex.l;
      // (the tables of this path hold element offsets: x entries * 64, y entries * fw * 64; `tile` already includes cp2)
      const v2f v1 = *reinterpret_cast<const v2f*>(tile + (ey.lo + ex.lo));
      const v2f v2 = *reinterpret_cast<const v2f*>(tile + (ey.lo + ex.hi));
      const v2f v3 = *reinterpret_cast<const v2f*>(tile + (ey.hi + ex.lo));
      const v2f v4 = *reinterpret_cast<const v2f*>(tile + (ey.hi + ex.hi));
      acc += w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
    }
  }
  return acc / (float)(GX * GY);
}

// RoIs too large for the square LDS tiles but with at most 2x2 semantic samples per 14x14 bin and sides up to SM_MAXSIDE px
// at network scale: every nucleus-sized box of a 40x slide (40-100 px after the x2 resize).  Their footprint (up to 30x30
// pixels of 256 bytes per map) fits no LDS tile, and gathering the taps of every sample from L2 moves each pixel 4-16 times.
// RoIAlign is separable, though: a bin is  sum_y sum_x wy[y] * wx[x] * F[y][x]  with per-axis weights that are the sums of
// the bilinear tap weights of the bin's samples (divided by the samples per axis).  One wave therefore STREAMS a RoI's
// footprint row by row: a row (<= 32 pixels x 64 channels) is copied into a two-slot LDS ring by direct global->LDS loads
// one row ahead, contracted along x into the 7 bin columns (6 merged taps per bin, static LDS offsets from a per-bin base)
// and scattered into the 49 accumulators with the row's 7 y-weights.  Every footprint pixel is read once, there is no
// block barrier (one wave = one block), and the arithmetic per RoI drops from ~4700 tap evaluations to ~65 rows x 90 FMAs.
// The sum runs in a different order than mmcv's sample loop (rounding differences of a few 1e-7 relative).
#define SM_MAXSIDE 112      // RoI side (network px) up to which the stream kernel applies: footprint <= 30 px on stride 4
#define SM_ROWPX 36         // LDS ring slot: 32 footprint pixels + room for the padded taps of the last bin
#define SM_J 5              // merged taps per bin and axis: samples of a bin lie within bw * (S-1)/S <= 3 px, so they touch <= 5 pixels
#define SM_FH 32            // rows of the dense y-weight table
#define SMF_W 4             // roi_feat7_stream_few_kernel: waves per RoI
#define SMF_MAX 1024        // mid-size RoIs per launch up to which that form runs instead of roi_feat7_stream_kernel

struct StreamTabs {
  float wx[7][SM_J];        // merged x weights of bin pw, tap j = pixel xlo[pw] + j
  int xlo[7];               // first footprint pixel (relative to fx0) of bin pw
  float wy[7][SM_FH];       // merged y weights of bin ph at footprint row yrel
  // round 5: the samples of both axes (7 bins x <= 4 each), computed ONCE per map by the lane that owns them; the three tables above are
  // built from these entries instead of recomputing a sample (coordinate, division, clamps: ~25 instructions) for every table entry it touches
  AxisEnt smp[2][28];
  unsigned char smv[2][28];   // (bytes: with the tables the kernel's LDS must stay within 20 KB, eight workgroups per CU)
};

// wx / wy from the stored samples: the sums run over a bin's samples in the order of the original loops, with the same terms
__device__ __forceinline__ void sm_weights(StreamTabs* tb, int fx0, int fy0, int Sx, int Sy, int lane) {
  if (lane < 7 * SM_J) {
    const int pw = lane / SM_J, j = lane - pw * SM_J;
    const int px = fx0 + tb->xlo[pw] + j;
    float wsum = 0.f;
    for (int is = 0; is < Sx; ++is) {
      const AxisEnt q = tb->smp[0][pw * Sx + is];
      if (tb->smv[0][pw * Sx + is]) { if (q.lo == px) wsum += q.h; if (q.hi == px) wsum += q.l; }
    }
    tb->wx[pw][j] = wsum / (float)Sx;
  }
  for (int t = lane; t < 7 * SM_FH; t += 64) {
    const int ph = t / SM_FH, yr = t - ph * SM_FH;
    float wsum = 0.f;
    for (int is = 0; is < Sy; ++is) {
      const AxisEnt q = tb->smp[1][ph * Sy + is];
      if (tb->smv[1][ph * Sy + is]) { if (q.lo == fy0 + yr) wsum += q.h; if (q.hi == fy0 + yr) wsum += q.l; }
    }
    tb->wy[ph][yr] = wsum / (float)Sy;
  }
}

// the samples of lane (axis, idx) into the table, and the first footprint pixel of every bin column (min over the bin's valid x samples:
// Sx is 2 or 4 and a bin's samples sit in neighbouring lanes, so one or two exchanges)
__device__ __forceinline__ void sm_store_samples(StreamTabs* tb, const AxisEnt& e, bool valid, bool is_y, int idx, int S, int Sx, int fx0) {
  if (idx < 28) { tb->smp[is_y ? 1 : 0][idx] = e; tb->smv[is_y ? 1 : 0][idx] = (valid && idx < 7 * S) ? 1 : 0; }
  int m = (valid && !is_y && idx < 7 * Sx) ? e.lo - fx0 : (1 << 30);
  m = min(m, __shfl_xor(m, 1));
  if (Sx > 2) m = min(m, __shfl_xor(m, 2));
  if (!is_y && idx < 7 * Sx && idx % Sx == 0) tb->xlo[idx / Sx] = m == (1 << 30) ? 0 : m;
}

// sample s (0 .. 7*S-1) of an axis: mmcv's coordinate and bilinear entry
__device__ __forceinline__ AxisEnt sm_sample(float start, float bs, int S, int s, int size, bool& valid) {
  const int pb = s / S, is = s - pb * S;
  const float c = start + (float)pb * bs + ((float)is + 0.5f) * bs / (float)S;   // same expression as roi_bin()
  return axis_entry(c, size, valid);
}

// one map of one RoI: adds RoIAlign(7x7, Sx x Sy samples per bin) of `map` into acc[49] (lane = channel)
__device__ __forceinline__ void sm_accumulate(const float* __restrict__ map, int H, int W, int b, float x1, float y1, float bw, float bh,
                                              int Sx, int Sy, StreamTabs* tb, float* ring, float (&acc)[49], int lane) {
  // ---- footprint bounds over the valid samples (lanes 0..27: x samples, 32..59: y samples)
  const bool is_y = lane >= 32;
  const int idx = lane & 31;
  const int S = is_y ? Sy : Sx;
  bool valid = false;
  AxisEnt e{0, 0, 0.f, 0.f};
  if (idx < 7 * S) e = sm_sample(is_y ? y1 : x1, is_y ? bh : bw, S, idx, is_y ? H : W, valid);
  const int lo = half_min(valid ? e.lo : (1 << 30)), hi = half_max(valid ? e.hi : -1);
  const int fx0 = __shfl(lo, 0), fx1 = __shfl(hi, 0), fy0 = __shfl(lo, 32), fy1 = __shfl(hi, 32);
  if (fx1 < 0 || fy1 < 0) return;                       // every sample of an axis lies outside the map: the term is 0
  const int fw = fx1 - fx0 + 1, fh = min(fy1 - fy0 + 1, SM_FH);     // (roi_classify_kernel admits only RoIs with fw <= 32, fh <= SM_FH)
  // ---- stream the rows: slot (yr & 1) of the ring holds row yr.  Row yr + 1 is loaded into registers (4 pixels = 1 KB per
  // wave instruction, 16 lanes x 16 bytes per pixel) before row yr is contracted and written to the other slot afterwards
  // (a wave-wide global->LDS instruction costs the SIMD ~100 issue cycles here, a dwordx4 load + ds_write_b128 pair ~25)
  const float* base = map + (((long long)b * H + fy0) * W + fx0) * 64;
  const int sub = lane >> 4, c4 = (lane & 15) * 4;
  const int n4 = (fw + 3) >> 2;
  v4f stg[8];
  auto load_row = [&](int yr) {
    const float* src = base + (long long)yr * W * 64;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      int px = 4 * q + sub;
      px = px < fw ? px : fw - 1;                       // groups past the footprint repeat its last pixel (never past the map)
      if (q < n4) stg[q] = *reinterpret_cast<const v4f*>(src + px * 64 + c4);
    }
  };
  auto store_row = [&](int yr) {
    float* dst = ring + (yr & 1) * (SM_ROWPX * 64);
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (q < n4) *reinterpret_cast<v4f*>(dst + (4 * q + sub) * 64 + c4) = stg[q];
  };
  load_row(0);                                         // round 5: the first row is requested BEFORE the tables are built and lands while they are
  // ---- merged per-axis weights from the stored samples.  x: lane = (bin, tap) of the 7 x SM_J table; y: 4 entries of the 7 x SM_FH table per lane
  sm_store_samples(tb, e, valid, is_y, idx, S, Sx, fx0);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  sm_weights(tb, fx0, fy0, Sx, Sy, lane);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  float wxr[7][SM_J];
  int xbase[7];
#pragma unroll
  for (int pw = 0; pw < 7; ++pw) {
    xbase[pw] = tb->xlo[pw] * 64 + lane;
#pragma unroll
    for (int j = 0; j < SM_J; ++j) wxr[pw][j] = tb->wx[pw][j];
  }
  store_row(0);
  for (int yr = 0; yr < fh; ++yr) {
    if (yr + 1 < fh) load_row(yr + 1);                 // stays in flight while row yr is contracted
    __builtin_amdgcn_wave_barrier();
    const float* row = ring + (yr & 1) * (SM_ROWPX * 64);
    float T[7];
#pragma unroll
    for (int pw = 0; pw < 7; ++pw) {
      float t = 0.f;
#pragma unroll
      for (int j = 0; j < SM_J; ++j) t = fmaf(wxr[pw][j], row[xbase[pw] + j * 64], t);
      T[pw] = t;
    }
#pragma unroll
    for (int ph = 0; ph < 7; ++ph) {
      const float wyv = tb->wy[ph][yr];
#pragma unroll
      for (int pw = 0; pw < 7; ++pw) acc[ph * 7 + pw] = fmaf(wyv, T[pw], acc[ph * 7 + pw]);
    }
    if (yr + 1 < fh) store_row(yr + 1);                // (waits for the loads; the other slot's readers are this wave itself)
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ __launch_bounds__(64) void roi_feat7_stream_kernel(RoiFeatParams p) {
  __shared__ __attribute__((aligned(16))) float ring[2 * SM_ROWPX * 64];
  __shared__ StreamTabs tabs;
  const int lane = threadIdx.x;
  const int nm = p.fb_count[1];
  if ((int)blockIdx.x >= nm || (p.stream_few && nm <= SMF_MAX)) return;     // (short lists: roi_feat7_stream_few_kernel)
  // ring pixels beyond a footprint's width are read with zero weights: clear the ring once so they never hold NaN bits
  // (afterwards they hold stale map values, which are finite)
  for (int t = lane; t < 2 * SM_ROWPX * 64; t += 64) ring[t] = 0.f;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  // jobs are fetched from a counter (fb_count[3], zeroed with the other counts): the grid is only as large as what is resident at
  // once, and RoIs differ fourfold in footprint rows
  for (;;) {
    int job = 0;
    if (lane == 0) job = atomicAdd(&p.fb_count[3], 1);
    job = __shfl(job, 0);
    if (job >= nm) break;
    const int r = p.mid_list[job];
    const float* roi = p.rois + (long long)r * 5;
    const int b = (int)roi[0];
    float acc[49];
#pragma unroll
    for (int k = 0; k < 49; ++k) acc[k] = 0.f;
    const RoiGeom g0 = roi_geom(roi, 0.25f, 7, 2), g1 = roi_geom(roi, 0.125f, 7, 2), gs = roi_geom(roi, 0.25f, 14, 0);
    const bool sem2 = gs.gw == 2 || gs.gh == 2;          // (classify guarantees gw, gh in {1, 2})
    // FPN level 0 (+ the semantic term when it is sampled at the same points: one sample per 14x14 bin), level 1, and -- when
    // the semantic 14x14 grid takes 2 samples per bin on an axis -- the semantic map itself: average-pooled 2x2 that is 2*g
    // samples per 7x7 bin and axis on the 14-grid's geometry (a 7-grid bin is two 14-grid bins)
    for (int m = 0; m < (sem2 ? 3 : 2); ++m) {
      const float* map = m == 0 ? (sem2 ? p.x0 : p.x0sem) : m == 1 ? p.x1 : p.sem;
      const RoiGeom& g = m == 0 ? g0 : m == 1 ? g1 : gs;
      const float bmul = m == 2 ? 2.f : 1.f;
      sm_accumulate(map, m == 1 ? p.H1 : p.H0, m == 1 ? p.W1 : p.W0, b, g.x1, g.y1, bmul * g.bw, bmul * g.bh, m == 2 ? 2 * gs.gw : 2,
                    m == 2 ? 2 * gs.gh : 2, &tabs, ring, acc, lane);
    }
    // attention-pooled levels 2, 3: one vector per RoI, added to every bin
    float gsum = 0.f;
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      const int Hl = l ? p.H3 : p.H2, Wl = l ? p.W3 : p.W2;
      const float st = l ? 32.f : 16.f;
      float cx = floorf((roi[1] + roi[3]) / (2.0f * st)), cy = floorf((roi[2] + roi[4]) / (2.0f * st));
      cx = fminf(fmaxf(cx, 0.f), (float)(Wl - 1));
      cy = fminf(fmaxf(cy, 0.f), (float)(Hl - 1));
      const float* G = l ? p.G3 : p.G2;
      gsum += G[(((long long)b * Hl + (int)cy) * Wl + (int)cx) * 64 + lane];
    }
#pragma unroll
    for (int k = 0; k < 49; ++k) acc[k] += gsum;
    float* out = p.out + (long long)r * 49 * 64;
#pragma unroll
    for (int k = 0; k < 49; ++k) out[k * 64 + lane] = acc[k];
  }
}

// ---- the same kernel for a SHORT list of mid-size RoIs (the usual load has a few dozen of them per batch).  One wave per RoI makes
// a RoI a chain of one memory latency per footprint row (~0.2 ms for ~100 rows over its maps): while the whole list fits the chip
// at once, a RoI gets a workgroup of SMF_W waves that all LOAD rows -- SMF_W rows per step into one half of a ring of 2 x SMF_W row
// slots, one barrier per step -- and wave 0 alone contracts them, in row order with the arithmetic of sm_accumulate: the features
// are bit-identical to the one-wave kernel's, so a RoI's result does not depend on how many others the batch holds.
struct StreamFewShared {
  StreamTabs tabs;
  int fx0, fy0, fw, fh;
};

__device__ __forceinline__ void smf_accumulate(const float* __restrict__ map, int H, int W, int b, float x1, float y1, float bw, float bh,
                                               int Sx, int Sy, StreamFewShared* sh, float* ring, float (&acc)[49], int lane, int wave) {
  StreamTabs* tb = &sh->tabs;
  if (wave == 0) {
    // ---- footprint bounds and merged weights: the statements of sm_accumulate
    const bool is_y = lane >= 32;
    const int idx = lane & 31;
    const int S = is_y ? Sy : Sx;
    bool valid = false;
    AxisEnt e{0, 0, 0.f, 0.f};
    if (idx < 7 * S) e = sm_sample(is_y ? y1 : x1, is_y ? bh : bw, S, idx, is_y ? H : W, valid);
    const int lo = half_min(valid ? e.lo : (1 << 30)), hi = half_max(valid ? e.hi : -1);
    const int fx0 = __shfl(lo, 0), fx1 = __shfl(hi, 0), fy0 = __shfl(lo, 32), fy1 = __shfl(hi, 32);
    const bool none = fx1 < 0 || fy1 < 0;
    const int fw = none ? 0 : fx1 - fx0 + 1, fh = none ? 0 : min(fy1 - fy0 + 1, SM_FH);
    if (lane == 0) { sh->fx0 = fx0; sh->fy0 = fy0; sh->fw = fw; sh->fh = fh; }
    if (!none) {
      sm_store_samples(tb, e, valid, is_y, idx, S, Sx, fx0);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      __builtin_amdgcn_wave_barrier();
      sm_weights(tb, fx0, fy0, Sx, Sy, lane);
    }
  }
  __syncthreads();
  const int fx0 = sh->fx0, fy0 = sh->fy0, fw = sh->fw, fh = sh->fh;
  if (fh == 0) { __syncthreads(); return; }             // every sample of an axis lies outside the map (workgroup-uniform)
  float wxr[7][SM_J];
  int xbase[7];
  if (wave == 0) {
#pragma unroll
    for (int pw = 0; pw < 7; ++pw) {
      xbase[pw] = tb->xlo[pw] * 64 + lane;
#pragma unroll
      for (int j = 0; j < SM_J; ++j) wxr[pw][j] = tb->wx[pw][j];
    }
  }
  const float* base = map + (((long long)b * H + fy0) * W + fx0) * 64;
  const int sub = lane >> 4, c4 = (lane & 15) * 4;
  const int n4 = (fw + 3) >> 2;
  v4f stg[8];
  auto load_row = [&](int yr) {
    const float* src = base + (long long)yr * W * 64;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      int px = 4 * q + sub;
      px = px < fw ? px : fw - 1;
      if (q < n4) stg[q] = *reinterpret_cast<const v4f*>(src + px * 64 + c4);
    }
  };
  auto store_row = [&](int yr) {                        // ring slot of row yr: half (yr / SMF_W) & 1, slot yr % SMF_W
    float* dst = ring + (((yr / SMF_W) & 1) * SMF_W + (yr % SMF_W)) * (SM_ROWPX * 64);
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (q < n4) *reinterpret_cast<v4f*>(dst + (4 * q + sub) * 64 + c4) = stg[q];
  };
  if (wave < fh) { load_row(wave); store_row(wave); }
  __syncthreads();
  for (int y0 = 0; y0 < fh; y0 += SMF_W) {
    const int ynext = y0 + SMF_W + wave;
    if (ynext < fh) load_row(ynext);                    // stays in flight while this step's rows are contracted
    if (wave == 0) {
      for (int yr = y0; yr < min(y0 + SMF_W, fh); ++yr) {
        const float* row = ring + (((yr / SMF_W) & 1) * SMF_W + (yr % SMF_W)) * (SM_ROWPX * 64);
        float T[7];
#pragma unroll
        for (int pw = 0; pw < 7; ++pw) {
          float t = 0.f;
#pragma unroll
          for (int j = 0; j < SM_J; ++j) t = fmaf(wxr[pw][j], row[xbase[pw] + j * 64], t);
          T[pw] = t;
        }
#pragma unroll
        for (int ph = 0; ph < 7; ++ph) {
          const float wyv = tb->wy[ph][yr];
#pragma unroll
          for (int pw = 0; pw < 7; ++pw) acc[ph * 7 + pw] = fmaf(wyv, T[pw], acc[ph * 7 + pw]);
        }
      }
    }
    if (ynext < fh) store_row(ynext);                   // the other half: its readers finished before the previous barrier
    __syncthreads();
  }
}

__global__ __launch_bounds__(64 * SMF_W) void roi_feat7_stream_few_kernel(RoiFeatParams p) {
  __shared__ __attribute__((aligned(16))) float ring[2 * SMF_W * SM_ROWPX * 64];
  __shared__ StreamFewShared sh;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nm = p.fb_count[1];
  if (nm > SMF_MAX || (int)blockIdx.x >= nm) return;
  for (int t = threadIdx.x; t < 2 * SMF_W * SM_ROWPX * 64; t += 64 * SMF_W) ring[t] = 0.f;     // (see roi_feat7_stream_kernel)
  __syncthreads();
  for (int job = blockIdx.x; job < nm; job += gridDim.x) {
    const int r = p.mid_list[job];
    const float* roi = p.rois + (long long)r * 5;
    const int b = (int)roi[0];
    float acc[49];
#pragma unroll
    for (int k = 0; k < 49; ++k) acc[k] = 0.f;
    const RoiGeom g0 = roi_geom(roi, 0.25f, 7, 2), g1 = roi_geom(roi, 0.125f, 7, 2), gs = roi_geom(roi, 0.25f, 14, 0);
    const bool sem2 = gs.gw == 2 || gs.gh == 2;
    for (int m = 0; m < (sem2 ? 3 : 2); ++m) {
      const float* map = m == 0 ? (sem2 ? p.x0 : p.x0sem) : m == 1 ? p.x1 : p.sem;
      const RoiGeom& g = m == 0 ? g0 : m == 1 ? g1 : gs;
      const float bmul = m == 2 ? 2.f : 1.f;
      smf_accumulate(map, m == 1 ? p.H1 : p.H0, m == 1 ? p.W1 : p.W0, b, g.x1, g.y1, bmul * g.bw, bmul * g.bh, m == 2 ? 2 * gs.gw : 2,
                     m == 2 ? 2 * gs.gh : 2, &sh, ring, acc, lane, wave);
    }
    if (wave == 0) {
      float gsum = 0.f;
#pragma unroll
      for (int l = 0; l < 2; ++l) {
        const int Hl = l ? p.H3 : p.H2, Wl = l ? p.W3 : p.W2;
        const float st = l ? 32.f : 16.f;
        float cx = floorf((roi[1] + roi[3]) / (2.0f * st)), cy = floorf((roi[2] + roi[4]) / (2.0f * st));
        cx = fminf(fmaxf(cx, 0.f), (float)(Wl - 1));
        cy = fminf(fmaxf(cy, 0.f), (float)(Hl - 1));
        const float* G = l ? p.G3 : p.G2;
        gsum += G[(((long long)b * Hl + (int)cy) * Wl + (int)cx) * 64 + lane];
      }
#pragma unroll
      for (int k = 0; k < 49; ++k) acc[k] += gsum;
      float* out = p.out + (long long)r * 49 * 64;
#pragma unroll
      for (int k = 0; k < 49; ++k) out[k * 64 + lane] = acc[k];
    }
  }
}

// RoIs beyond the stream kernel's limits (sides over SM_MAXSIDE px, or more than 2 x 2 semantic samples per 14 x 14 bin: merged
// clumps, component proposals up to the whole tile).  Same separable form -- a bin is sum_y sum_x wy[y] wx[x] F[y][x] with merged
// per-axis weights -- for footprints of any size.  One workgroup of 7 x BG_RS waves per RoI: wave (pw, rs) owns bin COLUMN pw and
// every BG_RS-th footprint row: per row it requests all taps of its column at once (lane = channel: one coalesced 256-byte pixel
// per load, merged weights from LDS), contracts them to one value and adds it into its 7 bin-row sums with the row's y weights.
// A wave thus carries 7 accumulators and up to BG_J loads in flight, the workgroup has 14 rows x columns going at once (a whole-tile
// proposal at the usual load is a latency chain: four waves with all seven columns each took 0.34 ms for one), and every footprint
// pixel is read once per map instead of once per sample tap (the former one-block-per-bin gathers: 21.6 ms per step at 100-200 px).
#define BG_J 40             // merged taps per bin column: a 7-grid bin of a 1024-px box spans 36.6 stride-4 pixels + 2 (BG_J % 8 == 0)
#define BG_FH 264           // footprint rows
#define BG_S 24             // samples per 7-grid bin and axis (2 x the adaptive 14-grid count: boxes up to 12 x 14 x 4 = 672 px); beyond: roi_feat7_giant_kernel
#define BG_RS 2             // row splits
#define BG_RPS 2            // rows per step and wave
#define BG_NT (7 * BG_RS * 64)
struct BigTabs {
  __attribute__((aligned(16))) float wx[7][BG_J];   // rows 16-byte aligned; entries from the padded tap count on are zero
  int xlo[7], span[7];
  float wy[7][BG_FH];
  int fx0, fy0, fw, fh;
  AxisEnt smp[2][7 * BG_S];        // the samples of both axes, computed once per map
  unsigned char smv[2][7 * BG_S];
  int blo[2], bhi[2];
};

// one map of one RoI: adds this wave's column of RoIAlign(7x7, Sx x Sy samples per bin) into acc[ph] (lane = channel)
__device__ __forceinline__ void bg_accumulate(const float* __restrict__ map, int H, int W, int b, float x1, float y1, float bw, float bh, int Sx, int Sy,
                                              BigTabs* tb, float (&acc)[7]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pw = wave % 7, rs = wave / 7;
  AxisEnt (&smp)[2][7 * BG_S] = tb->smp;
  unsigned char (&smv)[2][7 * BG_S] = tb->smv;
  // ---- the samples of both axes, computed once (wave 0: x, wave 1: y, strided) and kept in LDS; footprint bounds
  if (wave < 2) {
    const bool is_y = wave == 1;
    const int S = is_y ? Sy : Sx;
    int lo = 1 << 30, hi = -1;
    for (int sidx = lane; sidx < 7 * S; sidx += 64) {
      bool v;
      const AxisEnt e = sm_sample(is_y ? y1 : x1, is_y ? bh : bw, S, sidx, is_y ? H : W, v);
      smp[is_y][sidx] = e;
      smv[is_y][sidx] = v;
      if (v) { lo = min(lo, e.lo); hi = max(hi, e.hi); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }
    if (lane == 0) { tb->blo[wave] = lo; tb->bhi[wave] = hi; }
  }
  __syncthreads();
  if (tid == 0) {
    const int xl = tb->blo[0], xh = tb->bhi[0], yl = tb->blo[1], yh = tb->bhi[1];
    const bool empty = xh < 0 || yh < 0;
    tb->fx0 = empty ? 0 : xl; tb->fy0 = empty ? 0 : yl;
    tb->fw = empty ? 0 : xh - xl + 1; tb->fh = empty ? 0 : min(yh - yl + 1, BG_FH);
  }
  if (tid >= 64 && tid < 71) {                          // first pixel and span of each bin column
    const int q = tid - 64;
    int m = 1 << 30, mh = -1;
    for (int is = 0; is < Sx; ++is)
      if (smv[0][q * Sx + is]) { m = min(m, smp[0][q * Sx + is].lo); mh = max(mh, smp[0][q * Sx + is].hi); }
    tb->xlo[q] = m == (1 << 30) ? 0 : m;               // (absolute for now)
    tb->span[q] = mh < 0 ? 0 : mh - m + 1;
  }
  __syncthreads();
  const int fx0 = tb->fx0, fy0 = tb->fy0, fw = tb->fw, fh = tb->fh;
  if (fw == 0 || fh == 0) return;                       // every sample of an axis lies outside the map (workgroup-uniform)   // every sample of an axis lies outside the map (block-uniform)
  int J = 1;
#pragma unroll
  for (int q = 0; q < 7; ++q) J = max(J, tb->span[q]);
  J = min(J, BG_J);
  // taps are read in groups of 8: a bin column's first pixel is moved left where its padded range would leave the footprint (the
  // weights below are built from these starts, so they move with it); rows narrower than the padded range take the clamped path
  const int Jr = (J + 7) & ~7;
  const bool padded_ok = fw >= Jr;
  if (padded_ok && tid < 7) tb->xlo[tid] = min(tb->xlo[tid], fx0 + fw - Jr);
  __syncthreads();
  // ---- merged per-axis weights from the sample table (sums in sample order: deterministic)
  for (int t = tid; t < 7 * BG_J; t += BG_NT) {
    const int q = t / BG_J, j = t - q * BG_J;
    float wsum = 0.f;
    if (j < (padded_ok ? Jr : J)) {                   // (a start moved left puts the bin's pixels at taps up to Jr - 1)
      const int px = tb->xlo[q] + j;
      for (int is = 0; is < Sx; ++is) {
        const AxisEnt e = smp[0][q * Sx + is];
        if (smv[0][q * Sx + is]) { if (e.lo == px) wsum += e.h; if (e.hi == px) wsum += e.l; }
      }
    }
    tb->wx[q][j] = wsum / (float)Sx;
  }
  for (int t = tid; t < 7 * fh; t += BG_NT) {
    const int ph = t / fh, yr = t - ph * fh;
    float wsum = 0.f;
    for (int is = 0; is < Sy; ++is) {
      const AxisEnt e = smp[1][ph * Sy + is];
      if (smv[1][ph * Sy + is]) { if (e.lo == fy0 + yr) wsum += e.h; if (e.hi == fy0 + yr) wsum += e.l; }
    }
    tb->wy[ph][yr] = wsum / (float)Sy;
  }
  __syncthreads();
  // ---- this wave's column, its rows
  const int x0 = tb->xlo[pw] - fx0;
  const float* base = map + (((long long)b * H + fy0) * W + fx0 + (padded_ok ? x0 : 0)) * 64 + lane;
  const v4f* w4 = reinterpret_cast<const v4f*>(tb->wx[pw]);
  if (padded_ok) {
    // BG_RPS rows per step: their taps (BG_RPS x 16 per chunk) are all requested before the first multiply, so a step exposes one
    // memory latency for BG_RPS rows (a box is a latency chain: with two rows per step a whole-tile box took 0.2 ms)
    for (int yr = rs; yr < fh; yr += BG_RPS * BG_RS) {
      const float* rowp[BG_RPS];
      bool have[BG_RPS];                                  // wave-uniform
#pragma unroll
      for (int k = 0; k < BG_RPS; ++k) {
        const int y = yr + k * BG_RS;
        have[k] = y < fh;
        rowp[k] = base + (long long)(have[k] ? y : yr) * W * 64;
      }
      float t0[BG_RPS], t1[BG_RPS];
#pragma unroll
      for (int k = 0; k < BG_RPS; ++k) { t0[k] = 0.f; t1[k] = 0.f; }
      for (int j = 0; j < Jr; j += 16) {
        float v[BG_RPS][16];
        const bool two = j + 8 < Jr;
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int k = 0; k < BG_RPS; ++k) v[k][u] = rowp[k][(j + u) * 64];
        if (two) {
#pragma unroll
          for (int u = 8; u < 16; ++u)
#pragma unroll
            for (int k = 0; k < BG_RPS; ++k) v[k][u] = rowp[k][(j + u) * 64];
        }
        const v4f wa = w4[j >> 2], wb = w4[(j >> 2) + 1];
#pragma unroll
        for (int k = 0; k < BG_RPS; ++k) {
          t0[k] = fmaf(wa.x, v[k][0], t0[k]); t1[k] = fmaf(wa.y, v[k][1], t1[k]); t0[k] = fmaf(wa.z, v[k][2], t0[k]); t1[k] = fmaf(wa.w, v[k][3], t1[k]);
          t0[k] = fmaf(wb.x, v[k][4], t0[k]); t1[k] = fmaf(wb.y, v[k][5], t1[k]); t0[k] = fmaf(wb.z, v[k][6], t0[k]); t1[k] = fmaf(wb.w, v[k][7], t1[k]);
        }
        if (two) {
          const v4f wc = w4[(j >> 2) + 2], wd = w4[(j >> 2) + 3];
#pragma unroll
          for (int k = 0; k < BG_RPS; ++k) {
            t0[k] = fmaf(wc.x, v[k][8], t0[k]); t1[k] = fmaf(wc.y, v[k][9], t1[k]); t0[k] = fmaf(wc.z, v[k][10], t0[k]); t1[k] = fmaf(wc.w, v[k][11], t1[k]);
            t0[k] = fmaf(wd.x, v[k][12], t0[k]); t1[k] = fmaf(wd.y, v[k][13], t1[k]); t0[k] = fmaf(wd.z, v[k][14], t0[k]); t1[k] = fmaf(wd.w, v[k][15], t1[k]);
          }
        }
      }
      // rows enter the bin sums in ascending order (the order of the two-rows-per-step form: results are unchanged)
#pragma unroll
      for (int k = 0; k < BG_RPS; ++k) {
        if (have[k]) {
          const float t = t0[k] + t1[k];
          const int y = yr + k * BG_RS;
#pragma unroll
          for (int ph = 0; ph < 7; ++ph) acc[ph] = fmaf(tb->wy[ph][y], t, acc[ph]);
        }
      }
    }
  } else {
    for (int yr = rs; yr < fh; yr += BG_RS) {
      const float* row = base + (long long)yr * W * 64;
      float t0 = 0.f;
      for (int j = 0; j < J; ++j) t0 = fmaf(tb->wx[pw][j], row[min(x0 + j, fw - 1) * 64], t0);
#pragma unroll
      for (int ph = 0; ph < 7; ++ph) acc[ph] = fmaf(tb->wy[ph][yr], t0, acc[ph]);
    }
  }
  __syncthreads();                                       // the tables are rebuilt for the next map
}

__global__ __launch_bounds__(BG_NT) void roi_feat7_big_kernel(RoiFeatParams p) {
  __shared__ BigTabs tabs;
  __shared__ float part[BG_RS - 1][7][7][64];             // [row split - 1][pw][ph][channel]
  __shared__ float totl[7][7][64];                        // [pw][ph][channel]: sum over the maps (row split 0's waves, one slot per thread)
  __shared__ int s_job;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pw = wave % 7, rs = wave / 7;
  const int nb = p.fb_count[0];
  // A box is a latency chain (tables, then footprint rows in steps of one memory latency).  While the big boxes are few -- the usual
  // load: a few dozen component proposals per batch -- each of a box's maps goes to a workgroup of its own and
  // roi_feat7_big_combine_kernel adds the maps' partial results in map order (a kernel boundary: an in-kernel hand-over needs an
  // agent-scope release per workgroup, which writes back the whole L2 under the LDS-tile kernel's output stream -- measured slower
  // than no split).  The sums are formed in the same order in both modes (rows of a map, row splits, maps, attention term), so a
  // box's features do not depend on how many other big boxes the batch holds.
  const bool split = p.big_part && nb <= p.big_split_max;
  const int njobs = split ? 3 * nb : nb;
  for (;;) {                                              // jobs from a counter (fb_count[4]): boxes differ widely in rows
    if (threadIdx.x == 0) s_job = atomicAdd(&p.fb_count[4], 1);
    __syncthreads();
    const int job = s_job;
    __syncthreads();
    if (job >= njobs) break;
    const int bi = split ? job / 3 : job, only = split ? job - 3 * bi : -1;
    const int r = p.fb_list[bi];
    const float* roi = p.rois + (long long)r * 5;
    const int b = (int)roi[0];
    bool sem_sep;                                         // the semantic 14 x 14 grid takes its own samples
    { const RoiGeom gs = roi_geom(roi, 0.25f, 14, 0); sem_sep = gs.gw != 1 || gs.gh != 1; }
    const int nmaps = sem_sep ? 3 : 2;
    if (only >= nmaps) continue;                          // (workgroup-uniform)
    if (rs == 0) {
#pragma unroll
      for (int ph = 0; ph < 7; ++ph) totl[pw][ph][lane] = 0.f;
    }
    for (int m = 0; m < nmaps; ++m) {
      if (only >= 0 && m != only) continue;
      float acc[7];
#pragma unroll
      for (int k = 0; k < 7; ++k) acc[k] = 0.f;
      const float* map = m == 0 ? (sem_sep ? p.x0 : p.x0sem) : m == 1 ? p.x1 : p.sem;
      const RoiGeom g = m == 0 ? roi_geom(roi, 0.25f, 7, 2) : m == 1 ? roi_geom(roi, 0.125f, 7, 2) : roi_geom(roi, 0.25f, 14, 0);
      const float bmul = m == 2 ? 2.f : 1.f;             // a 7-grid bin is two 14-grid bins: 2 g samples per bin and axis
      bg_accumulate(map, m == 1 ? p.H1 : p.H0, m == 1 ? p.W1 : p.W0, b, g.x1, g.y1, bmul * g.bw, bmul * g.bh, m == 2 ? 2 * g.gw : 2,
                    m == 2 ? 2 * g.gh : 2, &tabs, acc);
      if (rs > 0) {
#pragma unroll
        for (int ph = 0; ph < 7; ++ph) part[rs - 1][pw][ph][lane] = acc[ph];
      }
      __syncthreads();
      if (rs == 0) {
#pragma unroll
        for (int ph = 0; ph < 7; ++ph) {
          float v = acc[ph];
#pragma unroll
          for (int q = 0; q < BG_RS - 1; ++q) v += part[q][pw][ph][lane];
          if (split) p.big_part[(((long long)bi * 3 + m) * 49 + ph * 7 + pw) * 64 + lane] = v;
          else totl[pw][ph][lane] += v;
        }
      }
      // (the next map's bg_accumulate passes several barriers before a wave writes `part` again)
    }
    if (split) continue;                                  // roi_feat7_big_combine_kernel adds the maps (workgroup-uniform)
    if (rs == 0) {
      float gsum = 0.f;
#pragma unroll
      for (int l = 0; l < 2; ++l) {
        const int Hl = l ? p.H3 : p.H2, Wl = l ? p.W3 : p.W2;
        const float st = l ? 32.f : 16.f;
        float cx = floorf((roi[1] + roi[3]) / (2.0f * st)), cy = floorf((roi[2] + roi[4]) / (2.0f * st));
        cx = fminf(fmaxf(cx, 0.f), (float)(Wl - 1));
        cy = fminf(fmaxf(cy, 0.f), (float)(Hl - 1));
        const float* G = l ? p.G3 : p.G2;
        gsum += G[(((long long)b * Hl + (int)cy) * Wl + (int)cx) * 64 + lane];
      }
      float* out = p.out + (long long)r * 49 * 64;
#pragma unroll
      for (int ph = 0; ph < 7; ++ph) out[(ph * 7 + pw) * 64 + lane] = totl[pw][ph][lane] + gsum;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(448) void roi_feat7_big_combine_kernel(RoiFeatParams p) {
  const int nb = p.fb_count[0];
  if (!(p.big_part && nb <= p.big_split_max)) return;
  const int lane = threadIdx.x & 63, pw = threadIdx.x >> 6;
  for (int bi = blockIdx.x; bi < nb; bi += gridDim.x) {
    const int r = p.fb_list[bi];
    const float* roi = p.rois + (long long)r * 5;
    const int b = (int)roi[0];
    bool sem_sep;
    { const RoiGeom gs = roi_geom(roi, 0.25f, 14, 0); sem_sep = gs.gw != 1 || gs.gh != 1; }
    const int nmaps = sem_sep ? 3 : 2;
    float gsum = 0.f;
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      const int Hl = l ? p.H3 : p.H2, Wl = l ? p.W3 : p.W2;
      const float st = l ? 32.f : 16.f;
      float cx = floorf((roi[1] + roi[3]) / (2.0f * st)), cy = floorf((roi[2] + roi[4]) / (2.0f * st));
      cx = fminf(fmaxf(cx, 0.f), (float)(Wl - 1));
      cy = fminf(fmaxf(cy, 0.f), (float)(Hl - 1));
      const float* G = l ? p.G3 : p.G2;
      gsum += G[(((long long)b * Hl + (int)cy) * Wl + (int)cx) * 64 + lane];
    }
    float* out = p.out + (long long)r * 49 * 64;
#pragma unroll
    for (int ph = 0; ph < 7; ++ph) {
      float t = 0.f;
      for (int m = 0; m < nmaps; ++m) t += p.big_part[(((long long)bi * 3 + m) * 49 + ph * 7 + pw) * 64 + lane];
      out[(ph * 7 + pw) * 64 + lane] = t + gsum;
    }
  }
}

// Boxes beyond the tables of roi_feat7_big_kernel (more than BG_S samples per bin and axis or BG_FH footprint rows: boxes over
// ~670 px, which only network inputs above 512 px can hold): mmcv's own sample loop, one workgroup per RoI, a bin per wave in
// turn.  Slow and general; classified last (roi_classify_kernel, class 4: the tail of fb_list).
__global__ __launch_bounds__(256) void roi_feat7_giant_kernel(RoiFeatParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ng = p.fb_count[2];
  for (int job = blockIdx.x; job < ng; job += gridDim.x) {
    const int r = p.fb_list[p.list_cap - 1 - job];
    const float* roi = p.rois + (long long)r * 5;
    const int b = (int)roi[0];
    const RoiGeom g0 = roi_geom(roi, 0.25f, 7, 2), g1 = roi_geom(roi, 0.125f, 7, 2), gs = roi_geom(roi, 0.25f, 14, 0);
    const float* f0 = p.x0 + (long long)b * p.H0 * p.W0 * 64;
    const float* f1 = p.x1 + (long long)b * p.H1 * p.W1 * 64;
    const float* fs = p.sem + (long long)b * p.H0 * p.W0 * 64;
    float gsum = 0.f;
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      const int Hl = l ? p.H3 : p.H2, Wl = l ? p.W3 : p.W2;
      const float st = l ? 32.f : 16.f;
      float cx = floorf((roi[1] + roi[3]) / (2.0f * st)), cy = floorf((roi[2] + roi[4]) / (2.0f * st));
      cx = fminf(fmaxf(cx, 0.f), (float)(Wl - 1));
      cy = fminf(fmaxf(cy, 0.f), (float)(Hl - 1));
      const float* G = l ? p.G3 : p.G2;
      gsum += G[(((long long)b * Hl + (int)cy) * Wl + (int)cx) * 64 + lane];
    }
    for (int bin = wave; bin < 49; bin += 4) {
      const int ph = bin / 7, pw = bin - ph * 7;
      float v = roi_bin(f0, p.H0, p.W0, g0.x1, g0.y1, g0.bw, g0.bh, g0.gw, g0.gh, pw, ph, lane);
      v += roi_bin(f1, p.H1, p.W1, g1.x1, g1.y1, g1.bw, g1.bh, g1.gw, g1.gh, pw, ph, lane);
      float sm = 0.f;                                   // adaptive_avg_pool2d 14 -> 7: the four 14-grid bins of this 7-grid bin
      for (int q = 0; q < 4; ++q) sm += roi_bin(fs, p.H0, p.W0, gs.x1, gs.y1, gs.bw, gs.bh, gs.gw, gs.gh, 2 * pw + (q & 1), 2 * ph + (q >> 1), lane);
      p.out[(long long)r * 49 * 64 + bin * 64 + lane] = v + sm * 0.25f + gsum;
    }
  }
}

// pre-pass: which RoIs fit the LDS tiles (one wave per RoI, the same plan code as the main kernel)
__global__ __launch_bounds__(1024) void roi_classify_kernel(RoiFeatParams p) {
  // 16 RoIs per workgroup, and ONE atomicAdd per workgroup and list: the RoIs of a list count themselves in LDS first.  With an atomicAdd per
  // RoI a real slide's load -- every one of 17 k boxes mid-size -- queued 17 k atomics on one counter: 193 us per stage against 8 us at the
  // synthetic load, a quarter of the RoI-feature time (round 4, rocprofv3 trace of the 40-100 px fixed load)
  __shared__ int cnt[3], base[3];          // lists: 0 = big boxes (class 2), 1 = mid-size (class 1), 2 = giant (class 4)
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 16 + (threadIdx.x >> 6);
  const bool valid = r < *p.r_dev;
  if (threadIdx.x < 3) cnt[threadIdx.x] = 0;
  __syncthreads();
  int li = -1, slot = -1;
  if (valid) {
    const float* roi = p.rois + (long long)r * 5;
    const RoiGeom g0 = roi_geom(roi, 0.25f, 7, 2), g1 = roi_geom(roi, 0.125f, 7, 2), gs = roi_geom(roi, 0.25f, 14, 0);
    const bool sem_g1 = gs.gw == 1 && gs.gh == 1;
    const LevelPlan l0 = plan_level(g0, 7, 2, p.H0, p.W0, TP0, lane);
    const LevelPlan l1 = plan_level(g1, 7, 2, p.H1, p.W1, TP1, lane);
    // 0: LDS tiles; 1: stream kernel (at most 2x2 semantic samples per 14x14 bin, sides up to SM_MAXSIDE px: footprints of at
    // most 30 x 30 pixels on stride 4, bins spanning at most SM_J pixels); 2: one block per bin (big proposals)
    const float rwn = roi[3] - roi[1], rhn = roi[4] - roi[2];
    int cls = (sem_g1 && l0.ok && l1.ok) ? 0 : (gs.gw <= 2 && gs.gh <= 2 && rwn <= (float)SM_MAXSIDE && rhn <= (float)SM_MAXSIDE) ? 1 : 2;
    // class 0 is split by footprint: 0 = fits the small tiles (8x8 / 5x5 pixels: boxes up to ~24 px, the usual nucleus), 3 = needs
    // the 12x12 / 7x7 tiles; the small variant takes a third of the LDS, so twice as many RoIs are in flight per CU
    if (cls == 0 && !(l0.fw <= TS0 && l0.fh <= TS0 && l1.fw <= TS1 && l1.fh <= TS1)) cls = 3;
    // class 2 boxes beyond the tables of the big-box kernel (samples per bin, footprint rows, taps per bin column): class 4
    if (cls == 2 && (2 * gs.gw > BG_S || 2 * gs.gh > BG_S || rhn * 0.25f + 4.f > (float)BG_FH || rwn * 0.25f / 7.f + 3.f > (float)BG_J)) cls = 4;
    if (lane == 0) {
      p.fb_flag[r] = (unsigned char)cls;
      li = cls == 2 ? 0 : cls == 1 ? 1 : cls == 4 ? 2 : -1;
      if (li >= 0) slot = atomicAdd(&cnt[li], 1);
    }
  }
  __syncthreads();
  if (threadIdx.x < 3 && cnt[threadIdx.x] > 0) base[threadIdx.x] = atomicAdd(&p.fb_count[threadIdx.x], cnt[threadIdx.x]);
  __syncthreads();
  if (li == 0) p.fb_list[base[0] + slot] = r;
  else if (li == 2) p.fb_list[p.list_cap - 1 - (base[2] + slot)] = r;      // giant boxes from the end of the same list
  else if (li == 1) p.mid_list[base[1] + slot] = r;
}

template <int T0, int T1, int FLAG>
__global__ __launch_bounds__(256) void roi_feat7_lds_kernel(RoiFeatParams p) {
  __shared__ float tile0[T0 * T0 * 64];
  __shared__ float tile1[T1 * T1 * 64];
  __shared__ AxisEnt tab[2][2][16];
  // one RoI per block: the 4 waves share the staged footprints and split the 49 bins, so each SIMD holds 4 waves of
  // 4 different RoIs (LDS allows 4 blocks per CU) and the LDS / global latencies of one hide behind the others
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x;
  if (r >= *p.r_dev || p.fb_flag[r] != FLAG) return;
  const float* roi = p.rois + (long long)r * 5;
  const int b = (int)roi[0];
  const int cp2 = 2 * (lane & 31), hw = lane >> 5;     // bin loop: 2 channels per lane, one bin per half-wave
  v2f gsum[2];
#pragma unroll
  for (int l = 0; l < 2; ++l) {
    const int Hl = l ? p.H3 : p.H2, Wl = l ? p.W3 : p.W2;
    const float st = l ? 32.f : 16.f;
    float cx = floorf((roi[1] + roi[3]) / (2.0f * st)), cy = floorf((roi[2] + roi[4]) / (2.0f * st));
    cx = fminf(fmaxf(cx, 0.f), (float)(Wl - 1));
    cy = fminf(fmaxf(cy, 0.f), (float)(Hl - 1));
    const float* G = l ? p.G3 : p.G2;
    gsum[l] = *reinterpret_cast<const v2f*>(G + (((long long)b * Hl + (int)cy) * Wl + (int)cx) * 64 + cp2);
  }
  float* out = p.out + (long long)r * 49 * 64;
  const RoiGeom g0 = roi_geom(roi, 0.25f, 7, 2), g1 = roi_geom(roi, 0.125f, 7, 2), gs = roi_geom(roi, 0.25f, 14, 0);
  const bool sem_g1 = gs.gw == 1 && gs.gh == 1;
  const LevelPlan l0 = plan_level(g0, 7, 2, p.H0, p.W0, T0, lane);   // every wave computes the same plan
  const LevelPlan l1 = plan_level(g1, 7, 2, p.H1, p.W1, T1, lane);
  // With one sample per bin the 14x14 semantic grid (fused_semantic_head -> adaptive_avg_pool2d to 7x7,
  // htc_roi_head_cus.py) samples exactly the 2x2-per-bin points of the 7x7 grid on the same stride-4 geometry, and both
  // results are averaged over the same 4 samples: by linearity one interpolation of the pre-added map x0 + sem serves both.
  if (!(sem_g1 && l0.ok && l1.ok)) return;   // (cannot happen: roi_classify_kernel ran the same test)
  if (wave == 0) {
    const int ax = lane >> 5, idx = lane & 31;
    if (idx < 14) {      // pixel indices -> element offsets inside the staged tile (x: * 64 channels, y: * row pitch)
      AxisEnt e0 = l0.ent, e1 = l1.ent;
      const int m0 = ax ? l0.fw * 64 : 64, m1 = ax ? l1.fw * 64 : 64;
      e0.lo *= m0; e0.hi *= m0; e1.lo *= m1; e1.hi *= m1;
      tab[0][ax][idx] = e0; tab[1][ax][idx] = e1;
    }
  }
  stage_tile(p.x0sem, p.H0, p.W0, b, l0, tile0, lane, wave);
  stage_tile(p.x1, p.H1, p.W1, b, l1, tile1, lane, wave);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  const v2f zero2 = {0.f, 0.f};
  for (int pair = wave; pair < 25; pair += 4) {         // bins 2*pair and 2*pair + 1 (bin 49 does not exist)
    const int bin = 2 * pair + hw;
    const bool live = bin < 49;
    const int bc = live ? bin : 48;
    const int ph = bc / 7, pw = bc - ph * 7;
    v2f v = zero2;
    v += l0.empty ? zero2 : bin_lds2<2>(tile0 + cp2, tab[0][0], tab[0][1], pw, ph);
    v += l1.empty ? zero2 : bin_lds2<2>(tile1 + cp2, tab[1][0], tab[1][1], pw, ph);
    v += gsum[0];
    v += gsum[1];
    if (live) *reinterpret_cast<v2f*>(out + bin * 64 + cp2) = v;
  }
}

// 14x14 mask features: 4 waves per RoI, bins interleaved across the waves
__global__ __launch_bounds__(256) void roi_feat14_kernel(RoiFeatParams p) {
  __shared__ float tile0[TP0 * TP0 * 64];
  __shared__ float tile1[TP1 * TP1 * 64];
  __shared__ AxisEnt tab[2][2][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x;
  if (r >= *p.r_dev) return;
  const float* roi = p.rois + (long long)r * 5;
  const int b = (int)roi[0];
  const RoiGeom g0 = roi_geom(roi, 0.25f, 14, 0), g1 = roi_geom(roi, 0.125f, 14, 0);
  float* out = p.out + (long long)r * 196 * 64;
  // Nuclei-sized detections take one sample per bin (adaptive grid ceil(size / 14) = 1) and touch at most 8x8 / 5x5
  // pixels of the stride-4 / stride-8 maps: same LDS-staged, packed-fp32 evaluation as the 7x7 kernel.  The semantic term
  // is sampled on the same grid as FPN level 0 (same scale, bins and sample count): x0 + sem gives both in one pass.
  const bool one = g0.gw == 1 && g0.gh == 1 && g1.gw == 1 && g1.gh == 1;
  const LevelPlan l0 = plan_level(g0, 14, 1, p.H0, p.W0, TP0, lane);
  const LevelPlan l1 = plan_level(g1, 14, 1, p.H1, p.W1, TP1, lane);
  const int part = blockIdx.y;     // big detections are split over gridDim.y blocks (a few of them set the launch's duration)
  if (one && l0.ok && l1.ok) {   // block-uniform
    if (part != 0) return;
    const int cp2 = 2 * (lane & 31), hw = lane >> 5;
    v2f gsum[2];
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      const int Hl = l ? p.H3 : p.H2, Wl = l ? p.W3 : p.W2;
      const float st = l ? 32.f : 16.f;
      float cx = floorf((roi[1] + roi[3]) / (2.0f * st)), cy = floorf((roi[2] + roi[4]) / (2.0f * st));
      cx = fminf(fmaxf(cx, 0.f), (float)(Wl - 1));
      cy = fminf(fmaxf(cy, 0.f), (float)(Hl - 1));
      const float* G = l ? p.G3 : p.G2;
      gsum[l] = *reinterpret_cast<const v2f*>(G + (((long long)b * Hl + (int)cy) * Wl + (int)cx) * 64 + cp2);
    }
    if (wave == 0) {
      const int ax = lane >> 5, idx = lane & 31;
      if (idx < 14) {
        AxisEnt e0 = l0.ent, e1 = l1.ent;
        const int m0 = ax ? l0.fw * 64 : 64, m1 = ax ? l1.fw * 64 : 64;
        e0.lo *= m0; e0.hi *= m0; e1.lo *= m1; e1.hi *= m1;
        tab[0][ax][idx] = e0; tab[1][ax][idx] = e1;
      }
    }
    stage_tile(p.x0sem, p.H0, p.W0, b, l0, tile0, lane, wave);
    stage_tile(p.x1, p.H1, p.W1, b, l1, tile1, lane, wave);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const v2f zero2 = {0.f, 0.f};
    for (int pair = wave; pair < 98; pair += 4) {
      const int bin = 2 * pair + hw;
      const int ph = bin / 14, pw = bin - ph * 14;
      v2f v = zero2;
      v += l0.empty ? zero2 : bin_lds2<1>(tile0 + cp2, tab[0][0], tab[0][1], pw, ph);
      v += l1.empty ? zero2 : bin_lds2<1>(tile1 + cp2, tab[1][0], tab[1][1], pw, ph);
      v += gsum[0];
      v += gsum[1];
      *reinterpret_cast<v2f*>(out + bin * 64 + cp2) = v;
    }
    return;
  }
  // larger detections: gathers straight from the maps
  const float* f0 = p.x0sem + (long long)b * p.H0 * p.W0 * 64;
  const float* f1 = p.x1 + (long long)b * p.H1 * p.W1 * 64;
  float gsum[2];
#pragma unroll
  for (int l = 0; l < 2; ++l) {
    const int Hl = l ? p.H3 : p.H2, Wl = l ? p.W3 : p.W2;
    const float st = l ? 32.f : 16.f;
    float cx = floorf((roi[1] + roi[3]) / (2.0f * st)), cy = floorf((roi[2] + roi[4]) / (2.0f * st));
    cx = fminf(fmaxf(cx, 0.f), (float)(Wl - 1));
    cy = fminf(fmaxf(cy, 0.f), (float)(Hl - 1));
    const float* G = l ? p.G3 : p.G2;
    gsum[l] = G[(((long long)b * Hl + (int)cy) * Wl + (int)cx) * 64 + lane];
  }
  for (int bin = part * 4 + wave; bin < 196; bin += 4 * gridDim.y) {
    const int ph = bin / 14, pw = bin - ph * 14;
    float v = 0.f;
    v += roi_bin(f0, p.H0, p.W0, g0.x1, g0.y1, g0.bw, g0.bh, g0.gw, g0.gh, pw, ph, lane);
    v += roi_bin(f1, p.H1, p.W1, g1.x1, g1.y1, g1.bw, g1.bh, g1.gw, g1.gh, pw, ph, lane);
    v += gsum[0];
    v += gsum[1];
    out[bin * 64 + lane] = v;
  }
}

int launch_roi_feat(const RoiFeatParams& p, int P, int r_cap, hipStream_t s, hipStream_t side, hipEvent_t ev_fork, hipEvent_t ev_join, hipStream_t side2,
                    hipEvent_t ev_join2) {
  { static const int& skip_ = dev_knob_ref("SKIP", 0); if (skip_ & 32) return 0; }   // dev: ablation of the step (tools/dev/r04_ablate.py)
  ProfScope ps(P == 7 ? "roi_feat7" : "roi_feat14", 0, 0, s);
  if (r_cap <= 0) return 0;
  if (P == 7) {
    if (hipMemsetAsync(p.fb_count, 0, 8 * sizeof(int), s) != hipSuccess) return NUHTC_E_HIP;      // three list lengths, two job counters
    hipLaunchKernelGGL(roi_classify_kernel, dim3(cdiv(r_cap, 16)), dim3(1024), 0, s, p);
    // three size classes side by side: the LDS-tile kernels on the caller's stream, the mid-size stream kernel on `side`, the
    // big-box kernels on `side2` (at the usual load the latter two hold a few dozen boxes each and are latency chains of ~0.2 ms:
    // one after the other they outlasted the LDS kernels, which take as long for thousands of nucleus-sized boxes)
    const bool fork = side && ev_fork && ev_join;
    const bool fork2 = fork && side2 && ev_join2;
    if (fork && (hipEventRecord(ev_fork, s) != hipSuccess || hipStreamWaitEvent(side, ev_fork, 0) != hipSuccess)) return NUHTC_E_HIP;
    if (fork2 && hipStreamWaitEvent(side2, ev_fork, 0) != hipSuccess) return NUHTC_E_HIP;
    // Grids are sized to what is resident at once (the kernels walk their lists with a grid stride): the lists are short at the
    // usual load, and every workgroup of a larger grid still has to be placed on a chip the LDS-tile kernel keeps full before it can
    // find its list empty and leave.  (Even so the side kernels' workgroups are placed mostly as the LDS-tile kernel drains: a
    // higher stream priority for them costs the batches in flight 12 %, launching the short LDS-tile variant first changes nothing.)
    if (p.stream_few) hipLaunchKernelGGL(roi_feat7_stream_few_kernel, dim3(r_cap < 512 ? r_cap : 512), dim3(64 * SMF_W), 0, fork ? side : s, p);
    hipLaunchKernelGGL(roi_feat7_stream_kernel, dim3(r_cap < 2048 ? r_cap : 2048), dim3(64), 0, fork ? side : s, p);
    if (fork && hipEventRecord(ev_join, side) != hipSuccess) return NUHTC_E_HIP;
    hipStream_t sb = fork2 ? side2 : fork ? side : s;
    hipLaunchKernelGGL(roi_feat7_giant_kernel, dim3(r_cap < 256 ? r_cap : 256), dim3(256), 0, sb, p);      // (usually empty: first, so the chain the join waits for ends with the combine)
    hipLaunchKernelGGL(roi_feat7_big_kernel, dim3(std::min(512, 3 * r_cap)), dim3(BG_NT), 0, sb, p);
    if (p.big_part) hipLaunchKernelGGL(roi_feat7_big_combine_kernel, dim3(std::min(128, r_cap)), dim3(448), 0, sb, p);
    if (fork2 && hipEventRecord(ev_join2, side2) != hipSuccess) return NUHTC_E_HIP;
    if (fork && !fork2 && hipEventRecord(ev_join, side) != hipSuccess) return NUHTC_E_HIP;
    hipLaunchKernelGGL((roi_feat7_lds_kernel<TS0, TS1, 0>), dim3(r_cap), dim3(256), 0, s, p);
    hipLaunchKernelGGL((roi_feat7_lds_kernel<TP0, TP1, 3>), dim3(r_cap), dim3(256), 0, s, p);
    if (fork && hipStreamWaitEvent(s, ev_join, 0) != hipSuccess) return NUHTC_E_HIP;
    if (fork2 && hipStreamWaitEvent(s, ev_join2, 0) != hipSuccess) return NUHTC_E_HIP;
  } else if (P == 14) hipLaunchKernelGGL(roi_feat14_kernel, dim3(r_cap, 7), dim3(256), 0, s, p);
  else return NUHTC_E_INVALID;
  return hipGetLastError() == hipSuccess ? 0 : NUHTC_E_HIP;
}

// stand-alone RoIAlign on an NHWC map (kernel-level parity test of the mmcv semantics)
__global__ __launch_bounds__(256) void roi_align_kernel(const float* __restrict__ feat, int H, int W, const float* __restrict__ rois, int R, int P,
                                                        float scale, int sr, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const float* roi = rois + (long long)r * 5;
  const RoiGeom g = roi_geom(roi, scale, P, sr);
  const float* fb = feat + (long long)((int)roi[0]) * H * W * 64;
  for (int ph = 0; ph < P; ++ph)
    for (int pw = 0; pw < P; ++pw)
      out[((long long)r * P * P + ph * P + pw) * 64 + lane] = roi_bin(fb, H, W, g.x1, g.y1, g.bw, g.bh, g.gw, g.gh, pw, ph, lane);
}

int launch_roi_align(const float* feat, int N, int H, int W, int C, const float* rois, int R, const int* r_dev, int P, float scale, int sr,
                     float* out, int accumulate, hipStream_t s) {
  if (C != 64 || r_dev || accumulate) return NUHTC_E_INVALID;
  if (R <= 0) return 0;
  hipLaunchKernelGGL(roi_align_kernel, dim3(cdiv(R, 4)), dim3(256), 0, s, feat, H, W, rois, R, P, scale, sr, out);
  return hipGetLastError() == hipSuccess ? 0 : NUHTC_E_HIP;
}

// ------------------------------------------------------------------------------------------- bbox head tail
__device__ __forceinline__ void delta2bbox_dev(const float* roi4, const float* d, const float* stds, float img_w, float img_h, float* o) {
  const float MR = 4.135166556742356f;
  const float dx = d[0] * stds[0], dy = d[1] * stds[1];
  float dw = d[2] * stds[2], dh = d[3] * stds[3];
  dw = fminf(fmaxf(dw, -MR), MR);
  dh = fminf(fmaxf(dh, -MR), MR);
  const float pxc = (roi4[0] + roi4[2]) * 0.5f, pyc = (roi4[1] + roi4[3]) * 0.5f;
  const float pw = roi4[2] - roi4[0], ph = roi4[3] - roi4[1];
  const float gx = pxc + pw * dx, gy = pyc + ph * dy;
  const float gw = pw * expf(dw), gh = ph * expf(dh);
  o[0] = fminf(fmaxf(gx - gw * 0.5f, 0.f), img_w);
  o[1] = fminf(fmaxf(gy - gh * 0.5f, 0.f), img_h);
  o[2] = fminf(fmaxf(gx + gw * 0.5f, 0.f), img_w);
  o[3] = fminf(fmaxf(gy + gh * 0.5f, 0.f), img_h);
}

// one wave per RoI: h [R][256] -> cls [R][16] (nc+2 used), reg [R][4]; optional in-place cascade refinement of the roi
__global__ __launch_bounds__(256) void bbox_tail_kernel(BboxTailParams p) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= *p.r_dev) return;
  const float4 hv = reinterpret_cast<const float4*>(p.h + (long long)r * 256)[lane];
  const float nrm = sqrtf(wsum64(hv.x * hv.x + hv.y * hv.y + hv.z * hv.z + hv.w * hv.w));
  const float den = nrm + 1e-6f;
  const float4 xn = make_float4(hv.x / den * 20.f, hv.y / den * 20.f, hv.z / den * 20.f, hv.w / den * 20.f);
  const int nout = p.nc + 2;
  float keep = 0.f;
  for (int n = 0; n < nout + 4; ++n) {
    const float4 w = reinterpret_cast<const float4*>(p.w + (long long)n * 256)[lane];
    const float4 x = n < nout ? xn : hv;
    float d = wsum64(x.x * w.x + x.y * w.y + x.z * w.z + x.w * w.w) + p.b[n];
    if (lane == n) keep = d;
  }
  if (lane < nout) p.cls[(long long)r * 16 + lane] = keep;
  if (lane >= nout && lane < nout + 4) p.reg[(long long)r * 4 + (lane - nout)] = keep;
  if (p.refine) {
    float d[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) d[j] = __shfl(keep, nout + j);
    if (lane == 0) {
      float* roi = p.rois + (long long)r * 5;
      float o[4];
      delta2bbox_dev(roi + 1, d, p.stds, p.img_w, p.img_h, o);
      roi[1] = o[0]; roi[2] = o[1]; roi[3] = o[2]; roi[4] = o[3];
    }
  }
}

int launch_bbox_tail(const BboxTailParams& p, int r_cap, hipStream_t s) {
  ProfScope ps("bbox_tail", 0, 0, s);
  if (p.nc + 6 > 64) return NUHTC_E_INVALID;
  hipLaunchKernelGGL(bbox_tail_kernel, dim3(cdiv(r_cap, 4)), dim3(256), 0, s, p);
  return hipGetLastError() == hipSuccess ? 0 : NUHTC_E_HIP;
}

// ------------------------------------------------------------------------------------------- detection candidates
__device__ int block_exscan(int v, int* lds, int* total) {   // blockDim.x == 1024
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
  if (lane == 63) lds[wave] = incl;
  __syncthreads();
  if (wave == 0) {
    int w = lane < 16 ? lds[lane] : 0;
    int wi = w;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { int t = __shfl_up(wi, o); if (lane >= o) wi += t; }
    if (lane < 16) lds[lane] = wi - w;
    if (lane == 15) lds[16] = wi;
  }
  __syncthreads();
  int res = lds[wave] + incl - v;
  *total = lds[16];
  __syncthreads();
  return res;
}

__global__ __launch_bounds__(1024) void det_candidates_kernel(DetCandParams p) {
  __shared__ int sc[17];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int off = p.roi_off[b], n = p.roi_cnt[b];
  const int nc = p.nc;
  float* cb = p.cand_boxes + (long long)b * p.cap * 4;
  float* cs = p.cand_scores + (long long)b * p.cap;
  int* ci = p.cand_ids + (long long)b * p.cap;
  int written = 0;
  for (int base = 0; base < n; base += 1024) {
    const int j = base + tid;
    float sc_c[16];
    float box[4] = {0, 0, 0, 0};
    int cnt = 0;
    if (j < n) {
      const long long r = off + j;
      float l[16];
      float mx = -3.0e38f;
      for (int c = 0; c < nc + 2; ++c) {
        // sum(ms_scores) / 3.0  (htc_roi_head_cus.py:2283-2286): ((c0 + c1) + c2) / 3
        l[c] = ((p.cls0[r * 16 + c] + p.cls1[r * 16 + c]) + p.cls2[r * 16 + c]) / 3.0f;
      }
      for (int c = 0; c < nc; ++c) mx = fmaxf(mx, l[c]);
      float sum = 0.f;
      float e[16];
      for (int c = 0; c < nc; ++c) { e[c] = expf(l[c] - mx); sum += e[c]; }
      const float mo = fmaxf(l[nc], l[nc + 1]);
      const float e0 = expf(l[nc] - mo), e1 = expf(l[nc + 1] - mo);
      const float pos = e0 / (e0 + e1);
      delta2bbox_dev(p.rois + r * 5 + 1, p.reg2 + r * 4, p.stds, p.img_w, p.img_h, box);
#pragma unroll
      for (int q = 0; q < 4; ++q) box[q] = box[q] / p.scale;
      for (int c = 0; c < nc; ++c) {
        sc_c[c] = (e[c] / sum) * pos;
        if (sc_c[c] > p.score_thr) ++cnt;
      }
    }
    int tot;
    int o = block_exscan(cnt, sc, &tot) + written;
    if (j < n) {
      for (int c = 0; c < nc; ++c)
        if (sc_c[c] > p.score_thr) {
          if (o < p.cap) {
            cb[o * 4 + 0] = box[0]; cb[o * 4 + 1] = box[1]; cb[o * 4 + 2] = box[2]; cb[o * 4 + 3] = box[3];
            cs[o] = sc_c[c];
            ci[o] = c;
          }
          ++o;
        }
    }
    written += tot;
  }
  if (tid == 0) p.cand_count[b] = written < p.cap ? written : p.cap;
}

int launch_det_candidates(const DetCandParams& p, int B, hipStream_t s) {
  ProfScope ps("det_candidates", 0, 0, s);
  if (p.nc + 2 > 16) return NUHTC_E_INVALID;
  hipLaunchKernelGGL(det_candidates_kernel, dim3(B), dim3(1024), 0, s, p);
  return hipGetLastError() == hipSuccess ? 0 : NUHTC_E_HIP;
}

// labels of the kept detections + mask-branch RoIs (det boxes back in network pixels), compacted over the batch
__global__ __launch_bounds__(256) void det_finish_kernel(DetFinishParams p) {
  __shared__ int off[257];
  const int tid = threadIdx.x;
  if (tid == 0) {
    int acc = 0;
    for (int b = 0; b < p.B; ++b) {
      int n = p.det_counts[b] < p.limit ? p.det_counts[b] : p.limit;
      p.det_counts[b] = n;
      off[b] = acc; p.det_off[b] = acc; acc += n;
    }
    off[p.B] = acc;
    *p.det_total = acc;
  }
  __syncthreads();
  for (int b = 0; b < p.B; ++b) {
    const int n = off[b + 1] - off[b];
    for (int j = tid; j < n; j += 256) {
      const int src = p.keep_src[(long long)b * p.max_keep + j];
      p.labels[(long long)b * p.max_keep + j] = p.cand_ids[src];
      const float* d = p.dets + ((long long)b * p.max_keep + j) * 5;
      float* m = p.mask_rois + (long long)(off[b] + j) * 5;
      m[0] = (float)b; m[1] = d[0] * p.scale; m[2] = d[1] * p.scale; m[3] = d[2] * p.scale; m[4] = d[3] * p.scale;
    }
  }
}

int launch_det_finish(const DetFinishParams& p, hipStream_t s) {
  if (p.B > 256) return NUHTC_E_INVALID;
  hipLaunchKernelGGL(det_finish_kernel, dim3(1), dim3(256), 0, s, p);
  return hipGetLastError() == hipSuccess ? 0 : NUHTC_E_HIP;
}

// ------------------------------------------------------------------------------------------- mask paste
// one block per (detection slot, tile); 28x28 probabilities -> bit-packed H x W mask (bit x&31 of word x>>5)
__global__ __launch_bounds__(256) void paste_kernel(PasteParams p) {
  __shared__ float pr[28 * 28];
  __shared__ int s_area;
  const int j = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int n = p.det_counts[b];
  const int wpr = p.W >> 5;
  unsigned* out = p.masks + ((long long)b * p.max_keep + j) * p.H * wpr;
  if (j >= n) return;
  const long long d = p.det_off[b] + j;
  for (int e = tid; e < 784; e += 256) pr[e] = p.prob[d * 784 + e];
  if (tid == 0) s_area = 0;
  __syncthreads();
  const float* box = p.mask_rois + d * 5 + 1;   // network pixels; get_seg_masks divides by scale_factor again
  const float x0 = box[0] / p.scale, y0 = box[1] / p.scale, x1 = box[2] / p.scale, y1 = box[3] / p.scale;
  // CPU semantics (skip_empty=True, one instance per chunk): only the integer hull of the box is sampled
  const int hx0 = max((int)floorf(x0) - 1, 0), hy0 = max((int)floorf(y0) - 1, 0);
  const int hx1 = min((int)ceilf(x1) + 1, p.vW), hy1 = min((int)ceilf(y1) + 1, p.vH);
  int area = 0;
  for (int wi = tid; wi < p.H * wpr; wi += 256) {
    const int y = wi / wpr, wx = wi - y * wpr;
    unsigned bits = 0;
    if (y >= hy0 && y < hy1 && (wx << 5) < hx1 && ((wx << 5) + 32) > hx0) {
      float gy = ((float)y + 0.5f - y0) / (y1 - y0) * 2.0f - 1.0f;
      if (isinf(gy)) gy = 0.f;
      const float iy = ((gy + 1.0f) * 28.0f - 1.0f) / 2.0f;
      const float fy = floorf(iy);
      const int iy0 = (int)fy, iy1 = iy0 + 1;
      const float wy1 = iy - fy, wy0 = (fy + 1.0f) - iy;   // torch: nw = (ix_se - ix)*(iy_se - iy) with ix_se = ix_nw + 1
      for (int k = 0; k < 32; ++k) {
        const int x = (wx << 5) + k;
        if (x < hx0 || x >= hx1) continue;
        float gx = ((float)x + 0.5f - x0) / (x1 - x0) * 2.0f - 1.0f;
        if (isinf(gx)) gx = 0.f;
        const float ix = ((gx + 1.0f) * 28.0f - 1.0f) / 2.0f;
        const float fx = floorf(ix);
        const int ix0 = (int)fx, ix1 = ix0 + 1;
        const float wx1 = ix - fx, wx0 = (fx + 1.0f) - ix;
        float v = 0.f;
        const bool okx0 = ix0 >= 0 && ix0 < 28, okx1 = ix1 >= 0 && ix1 < 28;
        const bool oky0 = iy0 >= 0 && iy0 < 28, oky1 = iy1 >= 0 && iy1 < 28;
        if (oky0 && okx0) v += pr[iy0 * 28 + ix0] * (wx0 * wy0);
        if (oky0 && okx1) v += pr[iy0 * 28 + ix1] * (wx1 * wy0);
        if (oky1 && okx0) v += pr[iy1 * 28 + ix0] * (wx0 * wy1);
        if (oky1 && okx1) v += pr[iy1 * 28 + ix1] * (wx1 * wy1);
        if (v >= p.thr) bits |= 1u << k;   // NaN compares false, like the reference
      }
    }
    out[wi] = bits;
    area += __popc(bits);
  }
  area = (int)wsum64((float)area);   // exact: counts <= 2^24
  if ((tid & 63) == 0) atomicAdd(&s_area, area);
  __syncthreads();
  if (tid == 0 && p.areas) p.areas[(long long)b * p.max_keep + j] = s_area;
}

int launch_paste(const PasteParams& p, int B, hipStream_t s) {
  ProfScope ps("paste", 0, 0, s);
  hipLaunchKernelGGL(paste_kernel, dim3(p.max_keep, B), dim3(256), 0, s, p);
  return hipGetLastError() == hipSuccess ? 0 : NUHTC_E_HIP;
}

// ------------------------------------------------------------------------------------------- per-tile filter + mask-NMS
// tools/infer_wsi.py:486-531: detections are visited in class-major order (np.concatenate of the per-class lists),
// filtered by margin / min_area, ordered by np.argsort(score)[::-1] and greedily suppressed at mask IoU > thr.
constexpr int TP_NT = 1024;        // threads of tile_post_kernel (16 waves: the pair tests of a tile run side by side)
constexpr int TP_BITS = 512;       // candidates up to which the suppression bit matrix fits LDS
__global__ __launch_bounds__(TP_NT) void tile_post_kernel(TilePostParams p) {
  __shared__ unsigned long long okey[2048];   // sort keys
  __shared__ float4 sbox[2048];               // boxes of the candidates in visiting order (LDS: the pair loop is latency-bound)
  __shared__ int sarea[2048];
  __shared__ short sidx[2048];
  __shared__ unsigned char sup[2048];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n = p.det_counts[b];
  const int K = p.max_keep;
  const float* dets = p.dets + (long long)b * K * 5;
  const int* labels = p.labels + (long long)b * K;
  const int* areas = p.areas + (long long)b * K;
  unsigned char* keep = p.keep + (long long)b * K;
  const int wpr = p.W >> 5, words = p.H * wpr;
  int npad = 2; while (npad < n) npad <<= 1;
  // class-major position of detection j: (#dets with smaller label) + (#dets with equal label before j)
  for (int j = tid; j < npad; j += TP_NT) {
    unsigned long long key = ~0ull;
    if (j < n) {
      int pos = 0;
      for (int i = 0; i < n; ++i) pos += (labels[i] < labels[j]) || (labels[i] == labels[j] && i < j);
      const float* d = dets + j * 5;
      const bool ok = d[0] >= (float)p.margin && d[1] >= (float)p.margin && d[2] <= (float)(p.vW - p.margin) &&
                      d[3] <= (float)(p.vH - p.margin) && areas[j] >= p.min_area;
      // order: score desc, ties by class-major position desc (reverse of a stable ascending argsort)
      if (ok) key = ((unsigned long long)(~__float_as_uint(d[4])) << 32) | ((unsigned)(0xFFFF - pos) << 16) | (unsigned)j;
    }
    okey[j] = key;
    sup[j] = 0;
    if (j < n) keep[j] = 0;
  }
  __syncthreads();
  for (int k = 2; k <= npad; k <<= 1)
    for (int jj = k >> 1; jj > 0; jj >>= 1) {
      for (int t = tid; t < (npad >> 1); t += TP_NT) {
        int lo = ((t / jj) * (jj << 1)) + (t % jj), hi = lo + jj;
        bool asc = ((lo & k) == 0);
        unsigned long long a = okey[lo], c = okey[hi];
        if ((a > c) == asc) { okey[lo] = c; okey[hi] = a; }
      }
      __syncthreads();
    }
  for (int j = tid; j < n; j += TP_NT) {
    const bool valid = okey[j] != ~0ull;
    const int i = valid ? (int)(okey[j] & 0xFFFF) : 0;
    const float* d = dets + i * 5;
    sbox[j] = make_float4(d[0], d[1], d[2], d[3]);
    sarea[j] = areas[i];
    sidx[j] = valid ? (short)i : (short)-1;
  }
  __syncthreads();
  int m = 0;   // candidates that passed the filter sort first
  while (m < n && sidx[m] >= 0) ++m;
  const unsigned* masks = p.masks + (long long)b * K * words;
  const int lane = tid & 63, wave = tid >> 6;
  constexpr int NWV = TP_NT / 64;
  if (m <= TP_BITS) {
    // All pair tests first, side by side: wave w takes the candidates a = w, w + 16, ...; its lanes test the hulls of the later
    // candidates c (masks live inside their box hulls: no overlap of hulls -> IoU 0) and the few overlapping ones get a wave-wide
    // popcount of the AND over the rows of a's hull.  supb[a] = the later candidates a would suppress if kept.  The greedy pass
    // itself is then a walk over that bit matrix by one wave (lane w holds word w of the removed set): same keep set as visiting
    // the pairs in order, since a test's outcome does not depend on which candidates are still alive.
    __shared__ unsigned long long supb[TP_BITS][TP_BITS / 64];
    for (int i = tid; i < m * (TP_BITS / 64); i += TP_NT) supb[i / (TP_BITS / 64)][i % (TP_BITS / 64)] = 0ull;
    __syncthreads();
    for (int a = wave; a < m; a += NWV) {
      const float4 di = sbox[a];
      const unsigned* mi = masks + (long long)sidx[a] * words;
      const int y0 = max((int)floorf(di.y) - 1, 0), y1 = min((int)ceilf(di.w) + 1, p.H);   // rows of mask a's hull
      for (int c0 = a + 1; c0 < m; c0 += 64) {
        const int c = c0 + lane;
        bool ov = false;
        if (c < m) {
          const float4 dj = sbox[c];
          ov = fminf(di.z, dj.z) + 2.f > fmaxf(di.x, dj.x) - 2.f && fminf(di.w, dj.w) + 2.f > fmaxf(di.y, dj.y) - 2.f;
        }
        unsigned long long todo = __ballot(ov);
        while (todo) {
          const int l = __ffsll((long long)todo) - 1;
          todo &= todo - 1;
          const int cc = c0 + l;
          const unsigned* mj = masks + (long long)sidx[cc] * words;
          int cnt = 0;
          for (int wv = y0 * wpr + lane; wv < y1 * wpr; wv += 64) cnt += __popc(mi[wv] & mj[wv]);
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
          if (lane == 0) {
            const int uni = sarea[a] + sarea[cc] - cnt;
            if (uni > 0 && (double)cnt / (double)uni > p.thr) supb[a][cc >> 6] |= 1ull << (cc & 63);    // (this wave owns row a)
          }
        }
      }
    }
    __syncthreads();
    if (wave == 0) {
      unsigned long long removed = 0ull;                 // lane w < 8: word w of the removed set
      for (int a = 0; a < m; ++a) {
        const unsigned long long wa = __shfl(removed, a >> 6);
        if ((wa >> (a & 63)) & 1ull) continue;           // uniform
        if (lane == 0) keep[sidx[a]] = 1;
        if (lane < TP_BITS / 64) removed |= supb[a][lane];
      }
    }
    return;
  }
  // more candidates than the bit matrix holds: for a kept candidate a, the pairs (a, c > a) are independent of each other, so every wave
  // takes its own c and there is one barrier per kept candidate
  for (int a = 0; a < m; ++a) {
    if (sup[a]) continue;   // uniform: sup[] only changes between barriers
    const int i = sidx[a];
    if (tid == 0) keep[i] = 1;
    const float4 di = sbox[a];
    const unsigned* mi = masks + (long long)i * words;
    const int y0 = max((int)floorf(di.y) - 1, 0), y1 = min((int)ceilf(di.w) + 1, p.H);   // rows of mask i's hull
    for (int c = a + 1 + wave; c < m; c += NWV) {
      if (sup[c]) continue;
      const float4 dj = sbox[c];
      const bool ov = fminf(di.z, dj.z) + 2.f > fmaxf(di.x, dj.x) - 2.f && fminf(di.w, dj.w) + 2.f > fmaxf(di.y, dj.y) - 2.f;
      if (!ov) continue;
      const unsigned* mj = masks + (long long)sidx[c] * words;
      int cnt = 0;
      for (int wv = y0 * wpr + lane; wv < y1 * wpr; wv += 64) cnt += __popc(mi[wv] & mj[wv]);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
      if (lane == 0) {
        const int uni = sarea[a] + sarea[c] - cnt;
        if (uni > 0 && (double)cnt / (double)uni > p.thr) sup[c] = 1;
      }
    }
    __syncthreads();
  }
}

int launch_tile_post(const TilePostParams& p, int B, hipStream_t s) {
  ProfScope ps("tile_post", 0, 0, s);
  if (p.max_keep > 2048) return NUHTC_E_INVALID;
  hipLaunchKernelGGL(tile_post_kernel, dim3(B), dim3(TP_NT), 0, s, p);
  return hipGetLastError() == hipSuccess ? 0 : NUHTC_E_HIP;
}
