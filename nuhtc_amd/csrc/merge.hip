// Cross-tile duplicate removal on the GPU: the greedy overlap suppression of
// tools/nuclei_merge.py:62-174 `merge_overlap`, strategy 'probability' -- detections visited in descending score order,
// every still-alive one removes all later ones it overlaps with IoU > threshold -- over all detections of a slide (1e5-1e6).
//
// Overlap measure (NUHTC_OVERLAP_POLYGON, the reference's): IoU of the shapely polygons of the GeoJSON rings, i.e. of the
// outer contour cv2.findContours lists first for each instance mask (tools/infer_wsi.py:51-54), run through
// `buffer(0)` + "largest part" when the ring touches itself (nuclei_merge.py:37-59).  The ring runs through the centres of
// the border pixels, so its polygon is NOT the pixel set: it misses the half-pixel rim, cuts 8-connected corners
// diagonally, and one-pixel-wide parts have no area.  It is computed here exactly and without tracing anything:
//   * the polygon of a traced ring = union of the unit cells between 4 neighbouring pixel centres that have >= 3 corners in
//     the pixel set F (4 corners: the whole cell; 3 corners: the triangle on them), F = the selected component with its holes
//     filled.  Every such region is a union of the four quarter triangles a cell's two diagonals cut (N, E, S, W), so
//     area = 1/4 * number of quarters, and area(P ∩ Q) = 1/4 * popcount of the AND of the quarter planes: integers;
//   * merge_prepare_kernel (one wave per detection, bit-parallel floods in LDS) turns each mask crop into that pixel set:
//     component selection as OpenCV orders its list (the top-level 8-connected component found last), hole filling
//     (flood of the background from the frame), then the parts of the region that hang together by more than a point or a
//     line (flood over the cells through shared sides) of which the largest is kept;
//   * merge_pairs_kernel derives the quarter planes of both crops on the fly from two consecutive bit rows.
// oracle/merge_poly.py restates the same measure with slab-wise trapezoids on the traced rings; keep sets are compared bit
// for bit (tests/test_merge.py).  NUHTC_OVERLAP_MASK keeps the pixel-set IoU (oracle/merge.py).
//
// The sequential greedy pass is equivalent to a fixed point on the overlap graph: i is kept iff no overlapping neighbour of
// higher priority (higher score; ties: lower index) is kept.  So:
//   1. detections are hashed into a uniform grid (CELL px) by the cells their mask box touches (count / scan / fill);
//   2. one wave per detection tests every higher-priority candidate of its cells: box overlap, then the popcounts over the
//      intersection rectangle (each lane one candidate); candidates with IoU > thr are appended to the detection's
//      suppressor list -- 24 inline entries, further ones in chunks of a spill pool that the host grows and retries if it
//      runs out (a pair is handled in the one cell that holds the top-left of the box overlap);
//   3. rounds of: undecided i becomes dead if a suppressor is alive, alive if all suppressors are dead -- until nothing
//      changes (chains are short: a handful of rounds).
// Integer work throughout.
#include <algorithm>
#include <cstring>
#include <vector>

#include "common.h"

#define MG_CELL 64
#define MG_MAXSUP 24     // higher-priority overlapping neighbours kept inline per detection; more go to the spill pool

struct MergeArgs {
  const int32_t* boxes;    // [n][4] x0,y0,x1,y1 (exclusive) of the mask crop, slide pixels
  const float* scores;     // [n]
  const int32_t* areas;    // [n] set pixels (mask mode) / quarter cells of the polygon (polygon mode, computed by merge_prepare_kernel)
  const uint32_t* bits;    // bit-packed crops, rows of (w+31)/32 words, bit (x&31) of word x>>5 (polygon mode: the polygons' pixel sets)
  const int64_t* bit_off;  // [n] word offset of each crop
  long long n;
  double thr;
  int ox, oy, ncx, ncy;    // grid origin (pixels) and size (cells)
  int* cell_count;         // [ncx*ncy + 1]
  int* cell_start;         // [ncx*ncy + 1]
  int* cell_items;         // [sum of counts]
  int* sup;                // [n][MG_MAXSUP]
  int* nsup;               // [n]
  uint8_t* state;          // [n] 0 undecided, 1 alive, 2 dead
  int* flags;              // [0] changed, [1] spill pool exhausted, [2] spill pool top
  int* spill;              // chunks {next chunk (-1 = none), count, items...}
  int* spill_head;         // [n] first chunk of a detection, -1 = none
  int spill_cap;
  int polygon;             // overlap measure: 0 pixel sets, 1 polygons of the traced rings
  // merge_prepare_kernel
  const uint32_t* raw_bits;   // the mask crops as given
  uint32_t* poly_bits;        // out: pixel sets of the polygons, same layout
  int32_t* poly_area4;        // out: quarter cells
  int max_words;              // LDS buffer size (words) per bit image
};

__device__ __forceinline__ bool mg_prior(const MergeArgs& a, int j, int i) {   // j is visited before i
  const float sj = a.scores[j], si = a.scores[i];
  return sj > si || (sj == si && j < i);
}

__global__ void merge_count_kernel(MergeArgs a, int fill) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const int32_t* b = a.boxes + i * 4;
  if (b[2] <= b[0] || b[3] <= b[1]) return;
  const int cx0 = (b[0] - a.ox) / MG_CELL, cx1 = (b[2] - 1 - a.ox) / MG_CELL;
  const int cy0 = (b[1] - a.oy) / MG_CELL, cy1 = (b[3] - 1 - a.oy) / MG_CELL;
  for (int cy = cy0; cy <= cy1; ++cy)
    for (int cx = cx0; cx <= cx1; ++cx) {
      const int c = cy * a.ncx + cx;
      if (!fill) atomicAdd(&a.cell_count[c], 1);
      else a.cell_items[a.cell_start[c] + atomicAdd(&a.cell_count[c], 1)] = (int)i;
    }
}

// exclusive scan of cell_count -> cell_start (single block, any length); cell_count is zeroed for the fill pass
__global__ __launch_bounds__(1024) void merge_scan_kernel(MergeArgs a) {
  __shared__ int part[1024];
  const int ncell = a.ncx * a.ncy, tid = threadIdx.x;
  const int per = (ncell + 1023) / 1024;
  const int lo = tid * per, hi = min(lo + per, ncell);
  int s = 0;
  for (int c = lo; c < hi; ++c) s += a.cell_count[c];
  part[tid] = s;
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int t = 0; t < 1024; ++t) { const int v = part[t]; part[t] = acc; acc += v; }
    a.cell_start[ncell] = acc;
  }
  __syncthreads();
  int acc = part[tid];
  for (int c = lo; c < hi; ++c) {
    a.cell_start[c] = acc;
    acc += a.cell_count[c];
    a.cell_count[c] = 0;
  }
}

// 32 bits of row `row` of a crop starting at column xrel (relative to the crop), zero outside the crop
__device__ __forceinline__ uint32_t mg_word(const uint32_t* row, int wpr, int w, int xrel) {
  if (xrel >= w || xrel <= -32) return 0u;
  uint32_t out;
  if (xrel >= 0) {
    const int wi = xrel >> 5, sh = xrel & 31;
    out = row[wi] >> sh;
    if (sh && wi + 1 < wpr) out |= row[wi + 1] << (32 - sh);
  } else {
    out = row[0] << (-xrel);
  }
  return out;
}

// quarter cells shared by two polygons in up to 31 cells of one cell row: a0/a1 = 32 pixels of rows y / y+1 of the first
// pixel set, b0/b1 of the second, cm = mask of the cells to count.  Cell corners TL,TR / BL,BR = bits k,k+1 of the two rows;
// quarter N (top) is inside iff TL & TR & (BL | BR), E iff TR & BR & (TL | BL), S iff BL & BR & (TL | TR), W iff TL & BL & (TR | BR)
__device__ __forceinline__ int mg_quarters(uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1, uint32_t cm) {
  const uint32_t aTL = a0, aTR = a0 >> 1, aBL = a1, aBR = a1 >> 1;
  const uint32_t bTL = b0, bTR = b0 >> 1, bBL = b1, bBR = b1 >> 1;
  const uint32_t n = aTL & aTR & (aBL | aBR) & bTL & bTR & (bBL | bBR);
  const uint32_t e = aTR & aBR & (aTL | aBL) & bTR & bBR & (bTL | bBL);
  const uint32_t s = aBL & aBR & (aTL | aTR) & bBL & bBR & (bTL | bTR);
  const uint32_t w = aTL & aBL & (aTR | aBR) & bTL & bBL & (bTR | bBR);
  return __popc(n & cm) + __popc(e & cm) + __popc(s & cm) + __popc(w & cm);
}

// ----------------------------------------------------------------------------- mask crop -> pixel set of its ring's polygon
// bits of `m` reachable from the seed bits `x` (x subset of m) along runs of consecutive ones of m, inside one word
__device__ __forceinline__ uint32_t mg_run_fill(uint32_t x, uint32_t m) {
  const uint32_t up = (((m + x) ^ m) & m) | x;
  const uint32_t xr = __brev(x), mr = __brev(m);
  return up | __brev((((mr + xr) ^ mr) & mr) | xr);
}

#define MG_SYNC()                                         \
  do {                                                    \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); \
    __builtin_amdgcn_wave_barrier();                      \
  } while (0)

struct MgImg { int h, w, wpr, words, lane; uint32_t last_mask; };

// cur <- every pixel of `in` that is 8-connected to the seed pixels already in cur (wave-wide, LDS, in place)
__device__ void mg_flood8(const MgImg& g, const uint32_t* in, uint32_t* cur) {
  for (int iter = 0; iter < g.h * g.w + 2; ++iter) {
    bool changed = false;
    for (int t = g.lane; t < g.words; t += 64) {
      const uint32_t m = in[t];
      if (!m) continue;
      const int y = t / g.wpr, xw = t - y * g.wpr;
      uint32_t acc = 0;
      for (int dy = -1; dy <= 1; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= g.h) continue;
        const uint32_t v = cur[yy * g.wpr + xw];
        acc |= v | (v << 1) | (v >> 1);
        if (xw > 0) acc |= cur[yy * g.wpr + xw - 1] >> 31;
        if (xw < g.wpr - 1) acc |= cur[yy * g.wpr + xw + 1] << 31;
      }
      const uint32_t nw = mg_run_fill(acc & m, m);
      if (nw != cur[t]) { cur[t] = nw; changed = true; }
    }
    MG_SYNC();
    if (!__any(changed)) break;
  }
}

// out <- background (pixels not in `in`) that is 4-connected to the frame around the crop (everything outside the crop is
// background connected to the frame: the crop is the bounding box of the mask, and cv::findContours pads the image with zeros)
__device__ void mg_flood_frame(const MgImg& g, const uint32_t* in, uint32_t* out) {
  for (int t = g.lane; t < g.words; t += 64) out[t] = (t % g.wpr == g.wpr - 1) ? ~g.last_mask : 0u;   // columns past the crop
  MG_SYNC();
  for (int iter = 0; iter < g.h * g.w + 2; ++iter) {
    bool changed = false;
    for (int t = g.lane; t < g.words; t += 64) {
      const int y = t / g.wpr, xw = t - y * g.wpr;
      const uint32_t bg = ~in[t], cur = out[t];
      uint32_t nb = (cur << 1) | (cur >> 1);
      nb |= xw > 0 ? out[t - 1] >> 31 : 1u;
      nb |= xw < g.wpr - 1 ? out[t + 1] << 31 : 0x80000000u;
      nb |= y > 0 ? out[t - g.wpr] : ~0u;
      nb |= y < g.h - 1 ? out[t + g.wpr] : ~0u;
      const uint32_t nw = mg_run_fill((cur | nb) & bg, bg);
      if (nw != cur) { out[t] = nw; changed = true; }
    }
    MG_SYNC();
    if (!__any(changed)) break;
  }
}

__device__ __forceinline__ int mg_wave_min(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ int mg_wave_sum(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// raster index (y * 32*wpr + x) of the first set pixel of an LDS bit image, 0x7fffffff if empty
__device__ int mg_first(const MgImg& g, const uint32_t* b) {
  int best = 0x7fffffff;
  for (int t = g.lane; t < g.words; t += 64)
    if (b[t]) { best = min(best, t * 32 + __ffs(b[t]) - 1); break; }
  return mg_wave_min(best);
}

__global__ __launch_bounds__(64) void merge_prepare_kernel(MergeArgs a) {
  extern __shared__ uint32_t smem[];
  const long long i = blockIdx.x;
  const int lane = threadIdx.x;
  const int32_t* bx = a.boxes + i * 4;
  const int w = bx[2] - bx[0], h = bx[3] - bx[1];
  if (w <= 0 || h <= 0) {
    if (lane == 0) a.poly_area4[i] = 0;
    return;
  }
  MgImg g;
  g.h = h; g.w = w; g.wpr = (w + 31) >> 5; g.words = h * g.wpr; g.lane = lane;
  g.last_mask = (w & 31) ? (1u << (w & 31)) - 1u : ~0u;
  const uint32_t* src = a.raw_bits + a.bit_off[i];
  uint32_t* dst = a.poly_bits + a.bit_off[i];
  uint32_t* M = smem;                       // the mask crop                      | later: region cells
  uint32_t* A = M + a.max_words;            // current component                  | later: links to the right neighbour cell
  uint32_t* F = A + a.max_words;            // frame background, then pixel set F
  uint32_t* R = F + a.max_words;            // remaining components               | later: links to the cell below
  uint32_t* S = R + a.max_words;            // selected component                 | later: current part
  uint32_t* Q = S + a.max_words;            // frame background of the whole mask | later: remaining cells
  uint32_t* Kb = Q + a.max_words;           // best part
  for (int t = lane; t < g.words; t += 64) {
    uint32_t v = src[t];
    if (t % g.wpr == g.wpr - 1) v &= g.last_mask;
    M[t] = v;
    A[t] = 0;
  }
  MG_SYNC();
  int first = mg_first(g, M);
  if (first == 0x7fffffff) {
    for (int t = lane; t < g.words; t += 64) dst[t] = 0;
    if (lane == 0) a.poly_area4[i] = 0;
    return;
  }
  // ---- 1. the component cv2.findContours lists first: the top-level 8-connected component found last in the raster scan
  if (lane == 0) A[first >> 5] = 1u << (first & 31);
  MG_SYNC();
  mg_flood8(g, M, A);
  bool same = true;
  for (int t = lane; t < g.words; t += 64) same &= A[t] == M[t];
  const uint32_t* C = A;
  if (__any(!same)) {
    mg_flood_frame(g, M, Q);
    for (int t = lane; t < g.words; t += 64) { R[t] = M[t] & ~A[t]; S[t] = A[t]; }   // the first component is top-level
    MG_SYNC();
    for (int guard = 0; guard < g.h * g.w; ++guard) {
      first = mg_first(g, R);
      if (first == 0x7fffffff) break;
      for (int t = lane; t < g.words; t += 64) A[t] = 0;
      MG_SYNC();
      if (lane == 0) A[first >> 5] = 1u << (first & 31);
      MG_SYNC();
      mg_flood8(g, M, A);
      // top-level iff the background pixel west of its first pixel is connected to the frame
      const int t0 = first >> 5, b0 = first & 31;
      bool top;
      if (b0 > 0) top = (Q[t0] >> (b0 - 1)) & 1u;
      else top = (t0 % g.wpr == 0) ? true : (Q[t0 - 1] >> 31) & 1u;
      for (int t = lane; t < g.words; t += 64) {
        if (top) S[t] = A[t];
        R[t] &= ~A[t];
      }
      MG_SYNC();
    }
    C = S;
  }
  // ---- 2. fill its holes: F = everything the frame-connected background of ~C does not reach
  mg_flood_frame(g, C, F);
  for (int t = lane; t < g.words; t += 64) {
    uint32_t v = ~F[t];
    if (t % g.wpr == g.wpr - 1) v &= g.last_mask;
    F[t] = v;
  }
  MG_SYNC();
  // ---- 3. region cells (>= 3 corners in F) and their links through shared sides (both end points of the side in F)
  uint32_t* RC = M; uint32_t* LR = A; uint32_t* LD = R; uint32_t* K = S; uint32_t* REM = Q;
  for (int t = lane; t < g.words; t += 64) {
    const int y = t / g.wpr, xw = t - y * g.wpr;
    uint32_t rc = 0;
    if (y < g.h - 1) {
      const uint32_t f0 = F[t], f1 = F[t + g.wpr];
      const uint32_t TR = (f0 >> 1) | (xw < g.wpr - 1 ? F[t + 1] << 31 : 0u), BR = (f1 >> 1) | (xw < g.wpr - 1 ? F[t + g.wpr + 1] << 31 : 0u);
      rc = (f0 & TR & (f1 | BR)) | (f1 & BR & (f0 | TR));
    }
    RC[t] = rc;
  }
  MG_SYNC();
  int ncell = 0;
  for (int t = lane; t < g.words; t += 64) {
    const int y = t / g.wpr, xw = t - y * g.wpr;
    const uint32_t rc = RC[t];
    uint32_t lr = 0, ld = 0;
    if (rc) {
      const uint32_t f0 = F[t], f1 = F[t + g.wpr];
      const uint32_t TR = (f0 >> 1) | (xw < g.wpr - 1 ? F[t + 1] << 31 : 0u), BR = (f1 >> 1) | (xw < g.wpr - 1 ? F[t + g.wpr + 1] << 31 : 0u);
      const uint32_t rnext = (rc >> 1) | (xw < g.wpr - 1 ? RC[t + 1] << 31 : 0u);
      lr = rc & rnext & TR & BR;                                   // cell x <-> cell x+1
      if (y < g.h - 2) ld = rc & RC[t + g.wpr] & f1 & BR;          // cell (x,y) <-> cell (x,y+1)
    }
    LR[t] = lr; LD[t] = ld; REM[t] = rc; K[t] = 0; Kb[t] = 0;
    ncell += __popc(rc);
  }
  ncell = mg_wave_sum(ncell);
  MG_SYNC();
  // ---- 4. parts of the region that hang together through sides; keep the largest (area in quarter cells: 4 / 2 per cell)
  int best_area = -1;
  bool single = false;
  for (int guard = 0; guard < g.h * g.w && ncell > 0; ++guard) {
    first = mg_first(g, REM);
    if (first == 0x7fffffff) break;
    for (int t = lane; t < g.words; t += 64) K[t] = 0;
    MG_SYNC();
    if (lane == 0) K[first >> 5] = 1u << (first & 31);
    MG_SYNC();
    for (int iter = 0; iter < g.h * g.w + 2; ++iter) {
      bool changed = false;
      for (int t = lane; t < g.words; t += 64) {
        const uint32_t rc = RC[t];
        if (!rc) continue;
        const int y = t / g.wpr, xw = t - y * g.wpr;
        const uint32_t k = K[t], lr = LR[t];
        uint32_t in = ((k & lr) << 1) | ((k >> 1) & lr);
        if (xw > 0) in |= (K[t - 1] & LR[t - 1]) >> 31;
        if (xw < g.wpr - 1) in |= (K[t + 1] << 31) & lr;
        if (y > 0) in |= K[t - g.wpr] & LD[t - g.wpr];
        if (y < g.h - 2) in |= K[t + g.wpr] & LD[t];
        const uint32_t nw = k | (in & rc);
        if (nw != k) { K[t] = nw; changed = true; }
      }
      MG_SYNC();
      if (!__any(changed)) break;
    }
    int area = 0, cnt = 0;
    for (int t = lane; t < g.words; t += 64) {
      const uint32_t k = K[t];
      if (!k) continue;
      const int y = t / g.wpr, xw = t - y * g.wpr;
      const uint32_t f0 = F[t], f1 = F[t + g.wpr];
      const uint32_t TR = (f0 >> 1) | (xw < g.wpr - 1 ? F[t + 1] << 31 : 0u), BR = (f1 >> 1) | (xw < g.wpr - 1 ? F[t + g.wpr + 1] << 31 : 0u);
      area += 2 * __popc(k) + 2 * __popc(k & f0 & TR & f1 & BR);
      cnt += __popc(k);
    }
    area = mg_wave_sum(area);
    cnt = mg_wave_sum(cnt);
    if (guard == 0 && cnt == ncell) { single = true; best_area = area; break; }     // one part: the usual case
    const bool better = area > best_area;
    for (int t = lane; t < g.words; t += 64) {
      if (better) Kb[t] = K[t];
      REM[t] &= ~K[t];
    }
    if (better) best_area = area;
    MG_SYNC();
  }
  // ---- 5. pixel set of the kept part = corners of its cells that are in F
  const uint32_t* KK = single ? K : Kb;
  for (int t = lane; t < g.words; t += 64) {
    uint32_t v = F[t];
    if (!single) {
      const int y = t / g.wpr, xw = t - y * g.wpr;
      const uint32_t k = y < g.h - 1 ? KK[t] : 0u, ku = y > 0 ? KK[t - g.wpr] : 0u;
      uint32_t c = k | (k << 1) | ku | (ku << 1);
      if (xw > 0) c |= ((y < g.h - 1 ? KK[t - 1] : 0u) | (y > 0 ? KK[t - g.wpr - 1] : 0u)) >> 31;
      v &= c;
    }
    dst[t] = v;
  }
  if (lane == 0) a.poly_area4[i] = best_area < 0 ? 0 : best_area;
}

__global__ __launch_bounds__(256) void merge_pairs_kernel(MergeArgs a) {
  const int lane = threadIdx.x & 63;
  const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= a.n) return;
  const int32_t* bi = a.boxes + i * 4;
  const int ix0 = bi[0], iy0 = bi[1], ix1 = bi[2], iy1 = bi[3];
  if (ix1 <= ix0 || iy1 <= iy0) {
    if (lane == 0) { a.nsup[i] = 0; a.state[i] = 2; }      // empty mask: never kept
    return;
  }
  const int iw = ix1 - ix0, iwpr = (iw + 31) >> 5;
  const uint32_t* ibits = a.bits + a.bit_off[i];
  const int iarea = a.areas[i];
  const int cx0 = (ix0 - a.ox) / MG_CELL, cx1 = (ix1 - 1 - a.ox) / MG_CELL;
  const int cy0 = (iy0 - a.oy) / MG_CELL, cy1 = (iy1 - 1 - a.oy) / MG_CELL;
  int count = 0, head = -1;      // wave-uniform: suppressors found so far, first spill chunk
  for (int cy = cy0; cy <= cy1; ++cy)
    for (int cx = cx0; cx <= cx1; ++cx) {
      const int c = cy * a.ncx + cx;
      const int beg = a.cell_start[c], end = a.cell_start[c + 1];
      for (int base = beg; base < end; base += 64) {
        const int k = base + lane;
        bool hit = false;
        int j = -1;
        if (k < end) {
          j = a.cell_items[k];
          if (j != (int)i && mg_prior(a, j, (int)i)) {
            const int32_t* bj = a.boxes + (long long)j * 4;
            const int X0 = max(ix0, bj[0]), Y0 = max(iy0, bj[1]), X1 = min(ix1, bj[2]), Y1 = min(iy1, bj[3]);
            // the pair is handled in the cell that holds the top-left corner of the box intersection
            if (X1 > X0 && Y1 > Y0 && (X0 - a.ox) / MG_CELL == cx && (Y0 - a.oy) / MG_CELL == cy) {
              const int jw = bj[2] - bj[0], jwpr = (jw + 31) >> 5;
              const uint32_t* jbits = a.bits + a.bit_off[j];
              int inter = 0;
              if (!a.polygon) {
                for (int y = Y0; y < Y1; ++y) {
                  const uint32_t* ri = ibits + (long long)(y - iy0) * iwpr;
                  const uint32_t* rj = jbits + (long long)(y - bj[1]) * jwpr;
                  for (int x = X0; x < X1; x += 32) {
                    uint32_t m = mg_word(ri, iwpr, iw, x - ix0) & mg_word(rj, jwpr, jw, x - bj[0]);
                    if (X1 - x < 32) m &= (1u << (X1 - x)) - 1u;
                    inter += __popc(m);
                  }
                }
              } else {
                // cells (x, y) = unit squares between pixel centres (x,y)..(x+1,y+1); 31 cells per 32-pixel word
                for (int y = Y0; y < Y1 - 1; ++y) {
                  const uint32_t* ri = ibits + (long long)(y - iy0) * iwpr;
                  const uint32_t* rj = jbits + (long long)(y - bj[1]) * jwpr;
                  for (int x = X0; x < X1 - 1; x += 31) {
                    const uint32_t cm = X1 - 1 - x < 31 ? (1u << (X1 - 1 - x)) - 1u : 0x7fffffffu;
                    inter += mg_quarters(mg_word(ri, iwpr, iw, x - ix0), mg_word(ri + iwpr, iwpr, iw, x - ix0),
                                         mg_word(rj, jwpr, jw, x - bj[0]), mg_word(rj + jwpr, jwpr, jw, x - bj[0]), cm);
                  }
                }
              }
              const int uni = iarea + a.areas[j] - inter;
              hit = uni > 0 && (double)inter / (double)uni > a.thr;
            }
          }
        }
        const unsigned long long bal = __ballot(hit);
        const int nh = __popcll(bal);
        if (nh) {
          const int pos = count + __popcll(bal & ((1ull << lane) - 1ull));
          const int first_ov = max(count, MG_MAXSUP);            // first list position of this batch that does not fit inline
          const int n_ov = count + nh - first_ov;                // wave-uniform
          int base = -1;
          if (n_ov > 0) {
            if (lane == 0) {
              base = atomicAdd(&a.flags[2], n_ov + 2);
              if (base + n_ov + 2 > a.spill_cap) { a.flags[1] = 1; base = -1; }
              else { a.spill[base] = head; a.spill[base + 1] = n_ov; }
            }
            base = __shfl(base, 0);
            if (base >= 0) head = base;
          }
          if (hit) {
            if (pos < MG_MAXSUP) a.sup[i * MG_MAXSUP + pos] = j;
            else if (base >= 0) a.spill[base + 2 + pos - first_ov] = j;
          }
        }
        count += nh;
      }
    }
  if (lane == 0) {
    a.nsup[i] = count;
    a.spill_head[i] = head;
    a.state[i] = count == 0 ? 1 : 0;       // nothing of higher priority overlaps: kept
  }
}

__global__ void merge_round_kernel(MergeArgs a) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n || a.state[i] != 0) return;
  const int ns = min(a.nsup[i], MG_MAXSUP);
  bool all_dead = true;
  for (int k = 0; k < ns; ++k) {
    const uint8_t st = a.state[a.sup[i * MG_MAXSUP + k]];
    if (st == 1) { a.state[i] = 2; a.flags[0] = 1; return; }
    if (st == 0) all_dead = false;
  }
  for (int c = a.spill_head[i]; c >= 0; c = a.spill[c]) {
    const int cn = a.spill[c + 1];
    for (int k = 0; k < cn; ++k) {
      const uint8_t st = a.state[a.spill[c + 2 + k]];
      if (st == 1) { a.state[i] = 2; a.flags[0] = 1; return; }
      if (st == 0) all_dead = false;
    }
  }
  if (all_dead) { a.state[i] = 1; a.flags[0] = 1; }
}

__global__ void merge_finish_kernel(MergeArgs a, uint8_t* keep) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < a.n) keep[i] = a.state[i] == 1;
}

__global__ void merge_maxwords_kernel(MergeArgs a, int* out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const int32_t* b = a.boxes + i * 4;
  const int w = b[2] - b[0], h = b[3] - b[1];
  if (w > 0 && h > 0) atomicMax(out, h * ((w + 31) >> 5));
}

extern "C" int nuhtc_merge_overlap(int device, const int32_t* boxes, const float* scores, const int32_t* areas, const uint32_t* bits,
                                   const int64_t* bit_off, int64_t n, int64_t n_words, int overlap, double thr, int x_min, int y_min,
                                   int x_max, int y_max, uint8_t* keep_dev, void* stream) {
  if (n < 0 || (n > 0 && (!boxes || !scores || !bits || !bit_off || !keep_dev)) || x_max < x_min || y_max < y_min ||
      n > 2000000000LL || (overlap != NUHTC_OVERLAP_MASK && overlap != NUHTC_OVERLAP_POLYGON) || (overlap == NUHTC_OVERLAP_MASK && n > 0 && !areas) ||
      (overlap == NUHTC_OVERLAP_POLYGON && n_words < 0))
    return NUHTC_E_INVALID;
  if (n == 0) return 0;
  if (hipSetDevice(device) != hipSuccess) return NUHTC_E_HIP;
  hipStream_t s = (hipStream_t)stream;
  MergeArgs a;
  memset(&a, 0, sizeof(a));
  a.boxes = boxes; a.scores = scores; a.areas = areas; a.bits = bits; a.bit_off = bit_off; a.n = n; a.thr = thr;
  a.polygon = overlap == NUHTC_OVERLAP_POLYGON;
  a.ox = x_min; a.oy = y_min;
  a.ncx = (x_max - x_min) / MG_CELL + 1; a.ncy = (y_max - y_min) / MG_CELL + 1;
  const long long ncell = (long long)a.ncx * a.ncy;
  if (ncell > (1LL << 28)) return NUHTC_E_INVALID;
  std::vector<void*> tmp;
  auto alloc = [&](void** p, size_t bytes) { if (hipMalloc(p, bytes ? bytes : 16) != hipSuccess) return false; tmp.push_back(*p); return true; };
  auto release = [&]() { for (void* p : tmp) hipFree(p); };
  int rc = 0;
  int h_flags[3] = {0, 0, 0};
  long long total = 0;
  const unsigned nb = (unsigned)((n + 255) / 256);
  if (!alloc((void**)&a.cell_count, (ncell + 1) * sizeof(int)) || !alloc((void**)&a.cell_start, (ncell + 1) * sizeof(int)) ||
      !alloc((void**)&a.sup, (size_t)n * MG_MAXSUP * sizeof(int)) || !alloc((void**)&a.nsup, (size_t)n * sizeof(int)) ||
      !alloc((void**)&a.spill_head, (size_t)n * sizeof(int)) ||
      !alloc((void**)&a.state, (size_t)n) || !alloc((void**)&a.flags, 4 * sizeof(int))) { rc = NUHTC_E_HIP; goto done; }
  if (hipMemsetAsync(a.cell_count, 0, (ncell + 1) * sizeof(int), s) != hipSuccess || hipMemsetAsync(a.flags, 0, 4 * sizeof(int), s) != hipSuccess) { rc = NUHTC_E_HIP; goto done; }
  if (a.polygon) {
    // mask crops -> pixel sets of the ring polygons + their areas in quarter cells (scratch copies; the inputs stay untouched)
    int h_maxw = 0;
    int* d_maxw = a.flags + 3;
    hipLaunchKernelGGL(merge_maxwords_kernel, dim3(nb), dim3(256), 0, s, a, d_maxw);
    if (hipMemcpyAsync(&h_maxw, d_maxw, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { rc = NUHTC_E_HIP; goto done; }
    a.max_words = h_maxw > 0 ? h_maxw : 1;
    const size_t lds = (size_t)7 * a.max_words * sizeof(uint32_t);
    if (lds > 160 * 1024) { rc = NUHTC_E_INVALID; goto done; }       // a crop beyond ~430x430 pixels is no nucleus
    if (hipFuncSetAttribute((const void*)merge_prepare_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) { rc = NUHTC_E_HIP; goto done; }
    if (!alloc((void**)&a.poly_bits, (size_t)(n_words > 0 ? n_words : 1) * sizeof(uint32_t)) || !alloc((void**)&a.poly_area4, (size_t)n * sizeof(int32_t))) { rc = NUHTC_E_HIP; goto done; }
    a.raw_bits = bits;
    hipLaunchKernelGGL(merge_prepare_kernel, dim3((unsigned)n), dim3(64), lds, s, a);
    a.bits = a.poly_bits;
    a.areas = a.poly_area4;
  }
  hipLaunchKernelGGL(merge_count_kernel, dim3(nb), dim3(256), 0, s, a, 0);
  hipLaunchKernelGGL(merge_scan_kernel, dim3(1), dim3(1024), 0, s, a);
  {
    int h_total = 0;
    if (hipMemcpyAsync(&h_total, a.cell_start + ncell, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { rc = NUHTC_E_HIP; goto done; }
    total = h_total;
  }
  if (!alloc((void**)&a.cell_items, (size_t)(total > 0 ? total : 1) * sizeof(int))) { rc = NUHTC_E_HIP; goto done; }
  hipLaunchKernelGGL(merge_count_kernel, dim3(nb), dim3(256), 0, s, a, 1);
  // suppressor lists: 24 inline + chunks in a spill pool; a dense clump that exhausts the pool makes the pass run again with
  // a larger one
  a.spill_cap = (int)std::min<long long>(std::max<long long>(1 << 16, n), 1LL << 30);
  for (int attempt = 0;; ++attempt) {
    if (!alloc((void**)&a.spill, (size_t)a.spill_cap * sizeof(int))) { rc = NUHTC_E_HIP; goto done; }
    if (hipMemsetAsync(a.flags, 0, 3 * sizeof(int), s) != hipSuccess) { rc = NUHTC_E_HIP; goto done; }
    hipLaunchKernelGGL(merge_pairs_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, a);
    if (hipMemcpyAsync(h_flags, a.flags, 3 * sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { rc = NUHTC_E_HIP; goto done; }
    if (!h_flags[1]) break;
    if (attempt >= 6 || a.spill_cap >= (1 << 30)) { rc = NUHTC_E_CAPACITY; goto done; }
    a.spill_cap = (int)std::min<long long>(std::max<long long>((long long)h_flags[2] + 1024, 4LL * a.spill_cap), 1LL << 30);
  }
  for (int round = 0; round < 4096; ++round) {
    if (hipMemsetAsync(a.flags, 0, sizeof(int), s) != hipSuccess) { rc = NUHTC_E_HIP; goto done; }
    for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(merge_round_kernel, dim3(nb), dim3(256), 0, s, a);
    if (hipMemcpyAsync(h_flags, a.flags, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { rc = NUHTC_E_HIP; goto done; }
    if (!h_flags[0]) break;
  }
  hipLaunchKernelGGL(merge_finish_kernel, dim3(nb), dim3(256), 0, s, a, keep_dev);
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) rc = NUHTC_E_HIP;
done:
  release();
  return rc;
}
