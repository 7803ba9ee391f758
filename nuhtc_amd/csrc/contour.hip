// Outer-contour extraction of the kept instance masks on the GPU, so that only vertex lists cross PCIe on the WSI path.
// Replaces `mask2inst` of tools/infer_wsi.py:51-54, `cv2.findContours(mask, RETR_TREE, CHAIN_APPROX_SIMPLE)[0][0]`, for the
// masks nuhtc_infer leaves in device memory, vertex for vertex:
//   * which border: OpenCV lists the top-level outer border it found LAST first (every border is linked at the head of its
//     parent's child list), i.e. the outer border of the component -- not enclosed in a hole of another one -- whose first
//     pixel in raster order comes last;
//   * the walk: Suzuki-Abe border following with Freeman codes 0 = east counted counter-clockwise on the screen: from the
//     start pixel search clockwise from west for the first set neighbour (i1), then repeatedly counter-clockwise from the
//     code after the one pointing back; stop when stepping from i1 back into the start;
//   * CHAIN_APPROX_SIMPLE: a pixel is a vertex when the code leaving it differs from the code that entered it (for the
//     start: code(i1) ^ 4).
// Checked against oracle/contour.py (the published algorithm with the full border labelling and hierarchy) on engine masks
// and on fragmented / nested / border-touching shapes (tests/test_hip_api.py).
//
// One wave per detection slot: the wave copies the bit-packed tile mask to LDS, lane 0 walks the border of the component
// holding the first set pixel, marking the pixels it visits and writing vertices straight to the output; the wave then looks
// for a run start (set pixel, clear west neighbour) the walk did not visit.  None (one blob, no hole: the usual nucleus) ->
// done.  Otherwise the wave floods the background that is 4-connected to the image frame (bit-parallel, run filling by carry
// propagation), and every further run start whose west neighbour lies in that region and was not visited starts another
// top-level border, walked the same way; the last one walked is the answer.
#include "common.h"

struct ContourParams {
  const uint32_t* masks;   // [B*max_per_img][H][wpr] bit-packed rows
  const uint8_t* keep;     // [B*max_per_img] or null (= every detection below counts[b])
  const int32_t* counts;   // [B]
  int max_per_img, H, W, wpr;
  int cap;                 // output vertices per instance
  int16_t* xy;             // [B*max_per_img][cap][2]
  int32_t* n;              // [B*max_per_img] vertex count; 0 = not traced (not kept / empty); -1 = more than cap vertices (trace on the host)
};

#define CT_WAVES 2

// Freeman codes, 0 = east, counter-clockwise on the screen: dx = {1,1,0,-1,-1,-1,0,1}, dy = {0,-1,-1,-1,0,1,1,1}; 2 bits hold d + 1
#define CDX(k) ((int)((0x901Au >> (2 * (k))) & 3u) - 1)
#define CDY(k) ((int)((0xA901u >> (2 * (k))) & 3u) - 1)

// bits of `m` reachable from the seed bits `x` (x subset of m) along runs of consecutive ones of m, inside one word
__device__ __forceinline__ uint32_t run_fill(uint32_t x, uint32_t m) {
  const uint32_t up = (((m + x) ^ m) & m) | x;
  const uint32_t xr = __brev(x), mr = __brev(m);
  const uint32_t dn = __brev((((mr + xr) ^ mr) & mr) | xr);
  return up | dn;
}

__global__ __launch_bounds__(64 * CT_WAVES) void contour_kernel(ContourParams p, int total) {
  extern __shared__ uint32_t smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int det = blockIdx.x * CT_WAVES + wave;
  if (det >= total) return;
  const int b = det / p.max_per_img, r = det - b * p.max_per_img;
  if (r >= p.counts[b] || (p.keep && !p.keep[det])) {
    if (lane == 0) p.n[det] = 0;
    return;
  }
  const int H = p.H, W = p.W, wpr = p.wpr, words = H * wpr;
  uint32_t* M = smem + wave * 3 * words;     // the mask
  uint32_t* V = M + words;                   // pixels visited by a border walk
  uint32_t* O = V + words;                   // background connected to the image frame (slow path only)
  const uint32_t* gm = p.masks + (long long)det * words;
  // ---- first set pixel in raster order and the bounding rows / word columns; only that rectangle is staged in LDS
  int first = 0x7fffffff, wy0 = 0x7fffffff, wy1 = -1, wx0 = 0x7fffffff, wx1 = -1;
  for (int i = lane; i < words; i += 64) {
    if (gm[i]) {
      const int y = i / wpr, x = i - y * wpr;
      first = min(first, i); wy0 = min(wy0, y); wy1 = max(wy1, y); wx0 = min(wx0, x); wx1 = max(wx1, x);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    first = min(first, __shfl_xor(first, o));
    wy0 = min(wy0, __shfl_xor(wy0, o)); wy1 = max(wy1, __shfl_xor(wy1, o));
    wx0 = min(wx0, __shfl_xor(wx0, o)); wx1 = max(wx1, __shfl_xor(wx1, o));
  }
  if (first == 0x7fffffff) {          // empty mask
    if (lane == 0) p.n[det] = 0;
    return;
  }
  const int bw = wx1 - wx0 + 1, items = (wy1 - wy0 + 1) * bw;
  for (int t = lane; t < items; t += 64) {
    const int i = (wy0 + t / bw) * wpr + wx0 + t % bw;
    M[i] = gm[i];
    V[i] = 0;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  int16_t* out = p.xy + (long long)det * p.cap * 2;
  const int cap = p.cap;

  auto fg = [&](int y, int x) -> bool {       // everything outside the staged rectangle is background
    return y >= wy0 && y <= wy1 && x >= wx0 * 32 && x < (wx1 + 1) * 32 && ((M[y * wpr + (x >> 5)] >> (x & 31)) & 1u);
  };
  // lane 0: walk the outer border that starts at pixel `start` (raster index), mark it in V, emit vertices; returns the count
  auto walk = [&](int start) -> int {
    const int y0 = start / W, x0 = start - y0 * W;
    int nout = 0;
    auto emit = [&](int x, int y) {
      if (nout < cap) { out[2 * nout] = (int16_t)x; out[2 * nout + 1] = (int16_t)y; }
      ++nout;
    };
    V[y0 * wpr + (x0 >> 5)] |= 1u << (x0 & 31);
    int s = 4;
    do { s = (s - 1) & 7; } while (!fg(y0 + CDY(s), x0 + CDX(s)) && s != 4);
    if (s == 4) { emit(x0, y0); return nout; }      // isolated pixel
    const int y1 = y0 + CDY(s), x1 = x0 + CDX(s);
    int cy = y0, cx = x0, prev_s = s ^ 4;
    for (int it = 0; it < 4 * H * W + 8; ++it) {
      int ny, nx;
      do { ++s; ny = cy + CDY(s & 7); nx = cx + CDX(s & 7); } while (!fg(ny, nx));
      s &= 7;
      if (s != prev_s) { emit(cx, cy); prev_s = s; }
      if (ny == y0 && nx == x0 && cy == y1 && cx == x1) break;
      cy = ny; cx = nx;
      V[cy * wpr + (cx >> 5)] |= 1u << (cx & 31);
      s = (s + 4) & 7;
    }
    return nout;
  };
  // wave: first run start (set pixel whose west neighbour is clear) not visited yet, optionally only those whose west
  // neighbour lies in O (pixels left of the image count as outside); raster index or 0x7fffffff
  auto next_start = [&](bool need_outer) -> int {
    int best = 0x7fffffff;
    for (int t = lane; t < items; t += 64) {
      const int y = wy0 + t / bw, xw = wx0 + t % bw, i = y * wpr + xw;
      const uint32_t w = M[i];
      if (!w) continue;
      const uint32_t west = (w << 1) | (xw > wx0 ? M[i - 1] >> 31 : 0u);
      uint32_t c = w & ~west & ~V[i];
      if (need_outer) c &= (O[i] << 1) | (xw > wx0 ? O[i - 1] >> 31 : 1u);
      if (c) best = min(best, y * W + xw * 32 + __ffs(c) - 1);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = min(best, __shfl_xor(best, o));
    return best;
  };

  int nout = 0;
  if (lane == 0) nout = walk((first / wpr) * W + (first % wpr) * 32 + __ffs(M[first]) - 1);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  int cand = next_start(false);
  if (cand != 0x7fffffff) {
    // ---- more than one component, or a hole: flood the frame-connected background inside the staged rectangle; whatever
    // lies outside the rectangle (or outside the image: cv::findContours pads with zeros) is background connected to the frame
    for (int t = lane; t < items; t += 64) O[(wy0 + t / bw) * wpr + wx0 + t % bw] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int iter = 0; iter < H * W; ++iter) {
      bool changed = false;
      for (int t = lane; t < items; t += 64) {
        const int y = wy0 + t / bw, xw = wx0 + t % bw, i = y * wpr + xw;
        const uint32_t bg = ~M[i], cur = O[i];
        uint32_t nb = (cur << 1) | (cur >> 1);
        nb |= xw > wx0 ? O[i - 1] >> 31 : 1u;
        nb |= xw < wx1 ? O[i + 1] << 31 : 0x80000000u;
        nb |= y > wy0 ? O[i - wpr] : ~0u;
        nb |= y < wy1 ? O[i + wpr] : ~0u;
        const uint32_t nw = run_fill((cur | nb) & bg, bg);
        if (nw != cur) { O[i] = nw; changed = true; }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      __builtin_amdgcn_wave_barrier();
      if (!__any(changed)) break;
    }
    // ---- every unvisited run start on the frame-connected background starts another top-level outer border
    for (int guard = 0; guard < H * W; ++guard) {
      cand = next_start(true);
      if (cand == 0x7fffffff) break;
      if (lane == 0) nout = walk(cand);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (lane == 0) p.n[det] = nout <= cap ? nout : -1;
}

int launch_contours(const uint32_t* masks, const uint8_t* keep, const int32_t* counts, int B, int max_per_img, int H, int W,
                    int cap, int16_t* xy, int32_t* n, hipStream_t s) {
  if (B <= 0) return 0;
  if (W % 32 != 0 || cap < 1 || H > 32767 || W > 32767) return NUHTC_E_INVALID;
  ContourParams p{masks, keep, counts, max_per_img, H, W, W / 32, cap, xy, n};
  const int total = B * max_per_img;
  const size_t lds = (size_t)CT_WAVES * 3 * H * (W / 32) * sizeof(uint32_t);
  if (lds > 160 * 1024) return NUHTC_E_INVALID;     // tiles beyond ~660x660: trace on the host (nuhtc_amd/contours.py)
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)contour_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return NUHTC_E_HIP;
    attr_set = true;
  }
  ProfScope ps("contours", 0, 0, s);
  hipLaunchKernelGGL(contour_kernel, dim3(cdiv(total, CT_WAVES)), dim3(64 * CT_WAVES), lds, s, p, total);
  return hipGetLastError() == hipSuccess ? 0 : NUHTC_E_HIP;
}

// ----------------------------------------------------------------------------- export of the kept detections
// After nuhtc_infer (+ nuhtc_mask_contours): compacts the detections that survived the per-tile filter + mask-NMS, in
// (tile, slot) order, into dense device buffers -- index, box + score, label, contour length, contour vertices, bit-packed
// mask -- so the host fetches them with a handful of fixed-size asynchronous copies (tools/infer_wsi.py:486-539 reads the
// same fields out of `result`).  One block: flags -> exclusive scan -> rows copied by whole waves.

__global__ __launch_bounds__(1024) void export_scan_kernel(ExportParams p, int32_t* pos) {
  __shared__ int part[1024];
  const int tid = threadIdx.x, total = p.B * p.K;
  const int per = (total + 1023) / 1024;
  const int lo = tid * per, hi = min(lo + per, total);
  int s = 0;
  for (int i = lo; i < hi; ++i) {
    const int b = i / p.K, r = i - b * p.K;
    s += (r < p.counts[b] && p.keep[i]) ? 1 : 0;
  }
  part[tid] = s;
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int t = 0; t < 1024; ++t) { const int v = part[t]; part[t] = acc; acc += v; }
    *p.n_out = acc;
  }
  __syncthreads();
  int acc = part[tid];
  for (int i = lo; i < hi; ++i) {
    const int b = i / p.K, r = i - b * p.K;
    const bool k = r < p.counts[b] && p.keep[i];
    pos[i] = k ? acc : -1;
    acc += k ? 1 : 0;
  }
}

__global__ __launch_bounds__(256) void export_copy_kernel(ExportParams p, const int32_t* pos) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= p.B * p.K) return;
  const int d = pos[i];
  if (d < 0 || d >= p.cap) return;
  if (lane == 0) { p.idx[d] = i; p.labels_out[d] = p.labels[i]; p.cn_out[d] = p.contour_n ? p.contour_n[i] : 0; }
  if (lane < 5) p.boxes_out[d * 5 + lane] = p.boxes[(long long)i * 5 + lane];
  if (p.words % 4 == 0) {         // every mask starts on a 16-byte boundary
    const uint4* ms = reinterpret_cast<const uint4*>(p.masks + (long long)i * p.words);
    uint4* md = reinterpret_cast<uint4*>(p.words_out + (long long)d * p.words);
    for (int t = lane; t < p.words / 4; t += 64) md[t] = ms[t];
  } else {                        // odd word counts (e.g. a 65 x 96 tile): dword copies
    const uint32_t* ms = p.masks + (long long)i * p.words;
    uint32_t* md = p.words_out + (long long)d * p.words;
    for (int t = lane; t < p.words; t += 64) md[t] = ms[t];
  }
  if (p.contour_xy) {
    const uint32_t* xs = reinterpret_cast<const uint32_t*>(p.contour_xy + (long long)i * p.ccap * 2);   // one (x, y) pair per dword
    uint32_t* xd = reinterpret_cast<uint32_t*>(p.xy_out + (long long)d * p.ccap * 2);
    const int nv = p.contour_n ? max(p.contour_n[i], 0) : p.ccap;
    for (int t = lane; t < min(nv, p.ccap); t += 64) xd[t] = xs[t];
  }
}

int launch_export_kept(const ExportParams& p, int32_t* pos_scratch, hipStream_t s) {
  if (p.B <= 0) return 0;
  ProfScope ps("export", 0, 0, s);
  hipLaunchKernelGGL(export_scan_kernel, dim3(1), dim3(1024), 0, s, p, pos_scratch);
  hipLaunchKernelGGL(export_copy_kernel, dim3(cdiv(p.B * p.K, 4)), dim3(256), 0, s, p, pos_scratch);
  return hipGetLastError() == hipSuccess ? 0 : NUHTC_E_HIP;
}


// ---- tight crops of the exported masks ------------------------------------------------------------------------------------
// The slide loop keeps, per kept detection, the mask cropped to its bounding rectangle (tools/infer_wsi.py:533-566 builds the
// polygon from it, the cross-tile merge compares the crops).  Cropping 8 KB bit images one by one on the host was the slowest
// part of the loop; here the exported masks (words_out of nuhtc_export_kept) are cropped on the device into one word pool:
// bounds + popcount per detection (one wave each), an exclusive scan of the crop sizes, then the rows shifted so that crop
// column 0 is bit 0 of word 0 (the layout nuhtc_merge_overlap takes).
__global__ __launch_bounds__(256) void crop_bounds_kernel(const uint32_t* __restrict__ words, const int32_t* __restrict__ n_dev, int cap, int H, int wpr,
                                                          int32_t* __restrict__ box, int32_t* __restrict__ area, int32_t* __restrict__ size) {
  __shared__ unsigned colw[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int d = blockIdx.x * 4 + wave;
  if (d >= cap) return;
  const int n = min(*n_dev, cap);
  if (d >= n) { if (lane == 0) { size[d] = 0; area[d] = 0; box[d * 4] = box[d * 4 + 1] = box[d * 4 + 2] = box[d * 4 + 3] = 0; } return; }
  colw[wave][lane] = 0u;
  const uint32_t* m = words + (long long)d * H * wpr;
  int ymin = 1 << 30, ymax = -1, pc = 0;
  for (int t = lane; t < H * wpr; t += 64) {
    const unsigned w = m[t];
    if (w) {
      const int y = t / wpr;
      ymin = min(ymin, y); ymax = max(ymax, y);
      pc += __popc(w);
      atomicOr(&colw[wave][t - y * wpr], w);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ymin = min(ymin, __shfl_xor(ymin, o)); ymax = max(ymax, __shfl_xor(ymax, o)); pc += __shfl_xor(pc, o);
  }
  // first / last set column: lane j looks at column word j (DS operations of a wave complete in order: the atomics above are done)
  const unsigned cw = lane < wpr ? colw[wave][lane] : 0u;
  int xmin = cw ? lane * 32 + __ffs((int)cw) - 1 : 1 << 30;
  int xmax = cw ? lane * 32 + 31 - __clz((int)cw) : -1;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { xmin = min(xmin, __shfl_xor(xmin, o)); xmax = max(xmax, __shfl_xor(xmax, o)); }
  if (lane == 0) {
    const bool any = ymax >= 0;
    box[d * 4 + 0] = any ? xmin : 0; box[d * 4 + 1] = any ? ymin : 0; box[d * 4 + 2] = any ? xmax + 1 : 0; box[d * 4 + 3] = any ? ymax + 1 : 0;
    area[d] = pc;
    size[d] = any ? (ymax + 1 - ymin) * ((xmax + 1 - xmin + 31) >> 5) : 0;
  }
}

__global__ __launch_bounds__(1024) void crop_scan_kernel(const int32_t* __restrict__ size, int cap, int32_t* __restrict__ off) {
  __shared__ int part[1024];
  const int tid = threadIdx.x;
  const int per = (cap + 1023) / 1024;
  const int lo = tid * per, hi = min(lo + per, cap);
  int s = 0;
  for (int i = lo; i < hi; ++i) s += size[i];
  part[tid] = s;
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int t = 0; t < 1024; ++t) { const int v = part[t]; part[t] = acc; acc += v; }
    off[cap] = acc;                       // total words the crops need
  }
  __syncthreads();
  int acc = part[tid];
  for (int i = lo; i < hi; ++i) { off[i] = acc; acc += size[i]; }
}

__global__ __launch_bounds__(256) void crop_write_kernel(const uint32_t* __restrict__ words, const int32_t* __restrict__ n_dev, int cap, int H, int wpr,
                                                         const int32_t* __restrict__ box, const int32_t* __restrict__ off, uint32_t* __restrict__ pool, int pool_cap) {
  const int lane = threadIdx.x & 63;
  const int d = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (d >= min(*n_dev, cap)) return;
  const int x0 = box[d * 4], y0 = box[d * 4 + 1], x1 = box[d * 4 + 2], y1 = box[d * 4 + 3];
  const int cw = (x1 - x0 + 31) >> 5, h = y1 - y0;
  const int o = off[d];
  if (h <= 0 || o + h * cw > pool_cap) return;          // crops past the pool are the caller's to cut from the full masks
  const uint32_t* m = words + (long long)d * H * wpr;
  const int w0 = x0 >> 5, sh = x0 & 31;
  for (int t = lane; t < h * cw; t += 64) {
    const int r = t / cw, j = t - r * cw;
    const uint32_t* row = m + (long long)(y0 + r) * wpr;
    const unsigned lo = row[w0 + j] >> sh;
    const unsigned hi = (sh && w0 + j + 1 < wpr) ? row[w0 + j + 1] << (32 - sh) : 0u;
    pool[o + t] = lo | hi;
  }
}

int launch_export_crops(const uint32_t* words, const int32_t* n_dev, int cap, int H, int wpr, int32_t* box, int32_t* area, int32_t* off, int32_t* size_scratch,
                        uint32_t* pool, int pool_cap, hipStream_t s) {
  if (cap <= 0) return 0;
  if (wpr > 64) return NUHTC_E_INVALID;
  ProfScope ps("export", 0, 0, s);
  hipLaunchKernelGGL(crop_bounds_kernel, dim3(cdiv(cap, 4)), dim3(256), 0, s, words, n_dev, cap, H, wpr, box, area, size_scratch);
  hipLaunchKernelGGL(crop_scan_kernel, dim3(1), dim3(1024), 0, s, size_scratch, cap, off);
  hipLaunchKernelGGL(crop_write_kernel, dim3(cdiv(cap, 4)), dim3(256), 0, s, words, n_dev, cap, H, wpr, box, off, pool, pool_cap);
  return hipGetLastError() == hipSuccess ? 0 : NUHTC_E_HIP;
}
