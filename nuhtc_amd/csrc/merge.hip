// Cross-tile duplicate removal on the GPU: the greedy overlap suppression of
// tools/nuclei_merge.py:62-174 `merge_overlap`, strategy 'probability' -- detections visited in descending score order,
// every still-alive one removes all later ones it overlaps with IoU > threshold -- over all detections of a slide (1e5-1e6).
// The reference intersects shapely polygons with an STRtree in a Python loop; here IoU is taken on the instance masks the
// polygons are traced from (bit-packed crops in slide coordinates, the engine's own output format), as in the host-side
// definition this build has used from the start.
//
// The sequential greedy pass is equivalent to a fixed point on the overlap graph: i is kept iff no overlapping neighbour of
// higher priority (higher score; ties: lower index) is kept.  So:
//   1. detections are hashed into a uniform grid (CELL px) by the cells their mask box touches (count / scan / fill);
//   2. one wave per detection tests every higher-priority candidate of its cells: box overlap, then popcount of the AND of
//      the two bit crops over the intersection rectangle (each lane one candidate); candidates with IoU > thr are appended
//      to the detection's suppressor list (a pair is handled in the one cell that holds the top-left of the box overlap);
//   3. rounds of: undecided i becomes dead if a suppressor is alive, alive if all suppressors are dead -- until nothing
//      changes (chains are short: a handful of rounds).
// Integer work throughout: the keep set is bit-identical to the sequential oracle (oracle/merge.py).
#include <cstring>
#include <vector>

#include "common.h"

#define MG_CELL 64
#define MG_MAXSUP 24     // higher-priority overlapping neighbours per detection (overflow -> NUHTC_E_CAPACITY)

struct MergeArgs {
  const int32_t* boxes;    // [n][4] x0,y0,x1,y1 (exclusive) of the mask crop, slide pixels
  const float* scores;     // [n]
  const int32_t* areas;    // [n] set pixels
  const uint32_t* bits;    // bit-packed crops, rows of (w+31)/32 words, bit (x&31) of word x>>5
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
  int* flags;              // [0] changed, [1] overflow
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
  int count = 0;      // wave-uniform
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
              for (int y = Y0; y < Y1; ++y) {
                const uint32_t* ri = ibits + (long long)(y - iy0) * iwpr;
                const uint32_t* rj = jbits + (long long)(y - bj[1]) * jwpr;
                for (int x = X0; x < X1; x += 32) {
                  uint32_t m = mg_word(ri, iwpr, iw, x - ix0) & mg_word(rj, jwpr, jw, x - bj[0]);
                  if (X1 - x < 32) m &= (1u << (X1 - x)) - 1u;
                  inter += __popc(m);
                }
              }
              const int uni = iarea + a.areas[j] - inter;
              hit = uni > 0 && (double)inter / (double)uni > a.thr;
            }
          }
        }
        const unsigned long long bal = __ballot(hit);
        if (hit) {
          const int pos = count + __popcll(bal & ((1ull << lane) - 1ull));
          if (pos < MG_MAXSUP) a.sup[i * MG_MAXSUP + pos] = j;
        }
        count += __popcll(bal);
      }
    }
  if (lane == 0) {
    if (count > MG_MAXSUP) { a.flags[1] = 1; count = MG_MAXSUP; }
    a.nsup[i] = count;
    a.state[i] = count == 0 ? 1 : 0;       // nothing of higher priority overlaps: kept
  }
}

__global__ void merge_round_kernel(MergeArgs a) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n || a.state[i] != 0) return;
  const int ns = a.nsup[i];
  bool all_dead = true;
  for (int k = 0; k < ns; ++k) {
    const uint8_t st = a.state[a.sup[i * MG_MAXSUP + k]];
    if (st == 1) { a.state[i] = 2; a.flags[0] = 1; return; }
    if (st == 0) all_dead = false;
  }
  if (all_dead) { a.state[i] = 1; a.flags[0] = 1; }
}

__global__ void merge_finish_kernel(MergeArgs a, uint8_t* keep) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < a.n) keep[i] = a.state[i] == 1;
}

extern "C" int nuhtc_merge_overlap(int device, const int32_t* boxes, const float* scores, const int32_t* areas, const uint32_t* bits,
                                   const int64_t* bit_off, int64_t n, double thr, int x_min, int y_min, int x_max, int y_max,
                                   uint8_t* keep_dev, void* stream) {
  if (n < 0 || (n > 0 && (!boxes || !scores || !areas || !bits || !bit_off || !keep_dev)) || x_max < x_min || y_max < y_min ||
      n > 2000000000LL)
    return NUHTC_E_INVALID;
  if (n == 0) return 0;
  if (hipSetDevice(device) != hipSuccess) return NUHTC_E_HIP;
  hipStream_t s = (hipStream_t)stream;
  MergeArgs a;
  memset(&a, 0, sizeof(a));
  a.boxes = boxes; a.scores = scores; a.areas = areas; a.bits = bits; a.bit_off = bit_off; a.n = n; a.thr = thr;
  a.ox = x_min; a.oy = y_min;
  a.ncx = (x_max - x_min) / MG_CELL + 1; a.ncy = (y_max - y_min) / MG_CELL + 1;
  const long long ncell = (long long)a.ncx * a.ncy;
  if (ncell > (1LL << 28)) return NUHTC_E_INVALID;
  std::vector<void*> tmp;
  auto alloc = [&](void** p, size_t bytes) { if (hipMalloc(p, bytes ? bytes : 16) != hipSuccess) return false; tmp.push_back(*p); return true; };
  auto release = [&]() { for (void* p : tmp) hipFree(p); };
  int rc = 0;
  int h_flags[2] = {0, 0};
  long long total = 0;
  const unsigned nb = (unsigned)((n + 255) / 256);
  if (!alloc((void**)&a.cell_count, (ncell + 1) * sizeof(int)) || !alloc((void**)&a.cell_start, (ncell + 1) * sizeof(int)) ||
      !alloc((void**)&a.sup, (size_t)n * MG_MAXSUP * sizeof(int)) || !alloc((void**)&a.nsup, (size_t)n * sizeof(int)) ||
      !alloc((void**)&a.state, (size_t)n) || !alloc((void**)&a.flags, 2 * sizeof(int))) { rc = NUHTC_E_HIP; goto done; }
  if (hipMemsetAsync(a.cell_count, 0, (ncell + 1) * sizeof(int), s) != hipSuccess || hipMemsetAsync(a.flags, 0, 2 * sizeof(int), s) != hipSuccess) { rc = NUHTC_E_HIP; goto done; }
  hipLaunchKernelGGL(merge_count_kernel, dim3(nb), dim3(256), 0, s, a, 0);
  hipLaunchKernelGGL(merge_scan_kernel, dim3(1), dim3(1024), 0, s, a);
  {
    int h_total = 0;
    if (hipMemcpyAsync(&h_total, a.cell_start + ncell, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { rc = NUHTC_E_HIP; goto done; }
    total = h_total;
  }
  if (!alloc((void**)&a.cell_items, (size_t)(total > 0 ? total : 1) * sizeof(int))) { rc = NUHTC_E_HIP; goto done; }
  hipLaunchKernelGGL(merge_count_kernel, dim3(nb), dim3(256), 0, s, a, 1);
  hipLaunchKernelGGL(merge_pairs_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, a);
  for (int round = 0; round < 4096; ++round) {
    if (hipMemsetAsync(a.flags, 0, sizeof(int), s) != hipSuccess) { rc = NUHTC_E_HIP; goto done; }
    for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(merge_round_kernel, dim3(nb), dim3(256), 0, s, a);
    if (hipMemcpyAsync(h_flags, a.flags, 2 * sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { rc = NUHTC_E_HIP; goto done; }
    if (h_flags[1]) { rc = NUHTC_E_CAPACITY; goto done; }
    if (!h_flags[0]) break;
  }
  hipLaunchKernelGGL(merge_finish_kernel, dim3(nb), dim3(256), 0, s, a, keep_dev);
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) rc = NUHTC_E_HIP;
done:
  release();
  return rc;
}
