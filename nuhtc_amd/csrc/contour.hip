// Outer-contour extraction of the kept instance masks on the GPU, so that only vertex lists cross PCIe on the WSI path.
// Replaces `mask2inst` of tools/infer_wsi.py:51-54 (cv2.findContours(mask, RETR_TREE, CHAIN_APPROX_SIMPLE)[0][0]) for the
// masks nuhtc_infer leaves in device memory.  Same algorithm as the host mirror nuhtc_amd/contours.py
// `trace_outer_contour` (Moore border following of the component holding the first foreground pixel in raster order,
// 8-connectivity, Jacob's stopping criterion, then removal of the vertices where the step direction does not change), which
// is what the parity test compares against vertex by vertex.
//
// One wave per detection slot.  The wave copies the instance's bit-packed tile mask to LDS (coalesced), finds the first
// set bit with a wave reduction, lane 0 walks the border (a serial chain of LDS bit tests: ~100-300 steps for a nucleus)
// into an LDS point list, and the whole wave compresses that list (direction-change test + ballot prefix) into the output.
#include "common.h"

#define RAW_CAP 2048   // border pixels kept in LDS per instance before compression

struct ContourParams {
  const uint32_t* masks;   // [B*max_per_img][H][wpr] bit-packed rows
  const uint8_t* keep;     // [B*max_per_img] or null (= every detection below counts[b])
  const int32_t* counts;   // [B]
  int max_per_img, H, W, wpr;
  int cap;                 // output vertices per instance
  int16_t* xy;             // [B*max_per_img][cap][2]
  int32_t* n;              // [B*max_per_img] vertex count; 0 = not traced (not kept / empty); -1 = overflow (trace on the host)
};

template <bool LDS_MASK>
__global__ __launch_bounds__(256) void contour_kernel(ContourParams p, int total) {
  extern __shared__ uint32_t smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int det = blockIdx.x * 4 + wave;
  if (det >= total) return;
  const int b = det / p.max_per_img, r = det - b * p.max_per_img;
  if (r >= p.counts[b] || (p.keep && !p.keep[det])) {
    if (lane == 0) p.n[det] = 0;
    return;
  }
  const int words = p.H * p.wpr;
  const int per_wave = (LDS_MASK ? words : 0) + RAW_CAP;
  uint32_t* lm = smem + wave * per_wave;                 // LDS image of the mask (when it fits)
  uint32_t* raw = lm + (LDS_MASK ? words : 0);           // border points, x | y << 16
  const uint32_t* gm = p.masks + (long long)det * words;
  // ---- stage the mask and find the first foreground pixel in raster order
  int first = 0x7fffffff;
  for (int i = lane; i < words; i += 64) {
    const uint32_t w = gm[i];
    if (LDS_MASK) lm[i] = w;
    if (w && i < first) first = i;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) first = min(first, __shfl_xor(first, o));
  if (first == 0x7fffffff) {          // empty mask
    if (lane == 0) p.n[det] = 0;
    return;
  }
  const uint32_t* m = LDS_MASK ? lm : gm;
  // (all 64 lanes of the wave run the same control flow up to here; LDS writes of this wave are visible to it after the
  // implicit wave-level ordering of ds operations -- no block barrier: waves of a block are independent)
  __builtin_amdgcn_s_waitcnt(0xC07F);
  int nraw = 0;
  bool overflow = false;
  if (lane == 0) {
    const int H = p.H, W = p.W, wpr = p.wpr;
    const int y0 = first / wpr, x0 = (first - y0 * wpr) * 32 + __ffs(m[first]) - 1;
    auto fg = [&](int y, int x) -> bool {
      return y >= 0 && y < H && x >= 0 && x < W && ((m[y * wpr + (x >> 5)] >> (x & 31)) & 1u);
    };
    // 8-neighbourhood clockwise from east (x right, y down), 2 bits per direction holding d + 1:
    // dx = {1,1,0,-1,-1,-1,0,1}, dy = {0,1,1,1,0,-1,-1,-1}
#define DXK(k) ((int)((0x901Au >> (2 * (k))) & 3u) - 1)
#define DYK(k) ((int)((0x01A9u >> (2 * (k))) & 3u) - 1)
    raw[0] = (uint32_t)x0 | ((uint32_t)y0 << 16);
    nraw = 1;
    bool any = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) any |= fg(y0 + DYK(k), x0 + DXK(k));
    if (any) {
      int cy = y0, cx = x0, d = 4, start_d = -1;
      const int limit = 4 * H * W + 8;
      for (int it = 0; it < limit; ++it) {
        int k = 0, ny = 0, nx = 0;
        bool found = false;
        for (int t = 1; t <= 8; ++t) {
          k = (d + t) & 7;
          ny = cy + DYK(k);
          nx = cx + DXK(k);
          if (fg(ny, nx)) { found = true; break; }
        }
        if (!found) break;
        if (cy == y0 && cx == x0) {
          if (start_d < 0) start_d = k;
          else if (k == start_d) break;          // back at the start, leaving in the same direction
        }
        cy = ny;
        cx = nx;
        if (nraw >= RAW_CAP) { overflow = true; break; }
        raw[nraw++] = (uint32_t)cx | ((uint32_t)cy << 16);
        d = (k + 4) & 7;
      }
      if (!overflow && nraw > 1 && raw[nraw - 1] == raw[0]) --nraw;
    }
#undef DXK
#undef DYK
  }
  nraw = __shfl(nraw, 0);
  overflow = __shfl((int)overflow, 0) != 0;
  if (overflow) {
    if (lane == 0) p.n[det] = -1;
    return;
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);
  int16_t* out = p.xy + (long long)det * p.cap * 2;
  if (nraw <= 2) {                      // nothing to compress (contours.py returns the points as they are)
    if (lane < nraw && lane < p.cap) {
      out[2 * lane] = (int16_t)(raw[lane] & 0xffff);
      out[2 * lane + 1] = (int16_t)(raw[lane] >> 16);
    }
    if (lane == 0) p.n[det] = nraw <= p.cap ? nraw : -1;
    return;
  }
  // ---- CHAIN_APPROX_SIMPLE: keep the points where the step to the next point differs from the step from the previous one
  int nout = 0;
  for (int base = 0; base < nraw; base += 64) {
    const int i = base + lane;
    bool kp = false;
    uint32_t cur = 0;
    if (i < nraw) {
      cur = raw[i];
      const uint32_t nx = raw[i + 1 < nraw ? i + 1 : 0], pv = raw[i > 0 ? i - 1 : nraw - 1];
      const int cx = cur & 0xffff, cy = cur >> 16;
      const int sx = (int)(nx & 0xffff) - cx, sy = (int)(nx >> 16) - cy;
      const int qx = cx - (int)(pv & 0xffff), qy = cy - (int)(pv >> 16);
      kp = sx != qx || sy != qy;
    }
    const unsigned long long bal = __ballot(kp);
    const int pos = nout + __popcll(bal & ((1ull << lane) - 1ull));
    if (kp && pos < p.cap) {
      out[2 * pos] = (int16_t)(cur & 0xffff);
      out[2 * pos + 1] = (int16_t)(cur >> 16);
    }
    nout += __popcll(bal);
  }
  if (nout == 0) {                      // (cannot happen for a closed border of > 2 points; mirrors `p[:1]`)
    if (lane == 0) { out[0] = (int16_t)(raw[0] & 0xffff); out[1] = (int16_t)(raw[0] >> 16); }
    nout = 1;
  }
  if (lane == 0) p.n[det] = nout <= p.cap ? nout : -1;
}

int launch_contours(const uint32_t* masks, const uint8_t* keep, const int32_t* counts, int B, int max_per_img, int H, int W,
                    int cap, int16_t* xy, int32_t* n, hipStream_t s) {
  if (B <= 0) return 0;
  if (W % 32 != 0 || cap < 1 || H > 32767 || W > 32767) return NUHTC_E_INVALID;
  ContourParams p{masks, keep, counts, max_per_img, H, W, W / 32, cap, xy, n};
  const int total = B * max_per_img;
  const int words = H * (W / 32);
  ProfScope ps("contours", 0, 0, s);
  dim3 grid(cdiv(total, 4)), blk(256);
  const size_t lds_full = 4 * (size_t)(words + RAW_CAP) * sizeof(uint32_t);
  if (lds_full <= 64 * 1024) hipLaunchKernelGGL(contour_kernel<true>, grid, blk, lds_full, s, p, total);
  else hipLaunchKernelGGL(contour_kernel<false>, grid, blk, 4 * RAW_CAP * sizeof(uint32_t), s, p, total);
  return hipGetLastError() == hipSuccess ? 0 : NUHTC_E_HIP;
}
