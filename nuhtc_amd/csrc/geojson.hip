// Host-side text writers for the QuPath GeoJSON documents of the WSI path.  The reference builds one Python dict per nucleus and lets
// json.dump walk the list (tools/infer_wsi.py:533-585 build, :659-664 dump): ~40 us per feature, 6.5 s for the 157 000 nuclei of a
// 10 000-tile slide on the rank that writes -- more than the GPUs need for the inference.  Here every rank serialises ITS records from the
// arrays it already holds (closed rings as one int32 vertex block + offsets, scores and centres as doubles): the text json.dump would
// produce, byte for byte, on a few host threads at memory speed; the bytes ride in the slide's one all-gather and rank 0 concatenates
// them.  No device code in this file.
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "common.h"

namespace {

inline int int_len(int32_t v) {
  uint32_t u = v < 0 ? 0u - (uint32_t)v : (uint32_t)v;
  int k = v < 0 ? 2 : 1;
  while (u >= 10) { u /= 10; ++k; }
  return k;
}

inline char* put_int(char* o, int32_t v) {     // decimal text of v, as json.dumps writes a Python int
  char tmp[12];
  uint32_t u = v < 0 ? 0u - (uint32_t)v : (uint32_t)v;
  int k = 0;
  do { tmp[k++] = (char)('0' + u % 10); u /= 10; } while (u);
  if (v < 0) *o++ = '-';
  while (k) *o++ = tmp[--k];
  return o;
}

// float.__repr__ as json.dumps writes it: the shortest digits that round-trip; exponent form below 1e-4 and from 1e16 on, positional
// otherwise with ".0" after an integral value; NaN / Infinity / -Infinity spelled json's way.  `o` holds at least 40 bytes.
inline int py_repr(double v, char* o) {
  if (std::isnan(v)) { memcpy(o, "NaN", 3); return 3; }
  if (std::isinf(v)) { if (v < 0) { memcpy(o, "-Infinity", 9); return 9; } memcpy(o, "Infinity", 8); return 8; }
  const double a = std::fabs(v);
  if (a != 0.0 && (a < 1e-4 || a >= 1e16)) return (int)(std::to_chars(o, o + 40, v, std::chars_format::scientific).ptr - o);
  char* e = std::to_chars(o, o + 40, v, std::chars_format::fixed).ptr;
  bool dot = false;
  for (const char* p = o; p < e; ++p) dot |= *p == '.';
  if (!dot) { *e++ = '.'; *e++ = '0'; }
  return (int)(e - o);
}

template <typename F>
void parallel_for(int64_t n, int threads, F fn) {       // fn(lo, hi) over contiguous blocks of [0, n)
  if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
  threads = threads < 1 ? 1 : threads > 16 ? 16 : threads;
  if (n < 4096 || threads == 1) { fn((int64_t)0, n); return; }
  std::vector<std::thread> pool;
  const int64_t per = (n + threads - 1) / threads;
  for (int t = 0; t < threads; ++t) {
    const int64_t lo = t * per, hi = lo + per < n ? lo + per : n;
    if (lo >= hi) break;
    pool.emplace_back([=] { fn(lo, hi); });
  }
  for (auto& th : pool) th.join();
}

struct Pieces {
  size_t lh = 0, lm[64], lt[64];
  const char* head;
  const char* const* mid;
  const char* const* tail;
  bool init(const char* h, const char* const* m, const char* const* t, int n_labels) {
    if (!h || !m || !t || n_labels <= 0 || n_labels > 64) return false;
    head = h; mid = m; tail = t;
    lh = strlen(h);
    for (int l = 0; l < n_labels; ++l) {
      if (!m[l] || !t[l]) return false;
      lm[l] = strlen(m[l]); lt[l] = strlen(t[l]);
    }
    return true;
  }
};

}  // namespace

// see include/nuhtc_hip.h
extern "C" int64_t nuhtc_write_ring_features(const int32_t* verts, const int64_t* ring_off, const int32_t* label, const double* score, int64_t n,
                                             const char* head, const char* const* mid, const char* const* tail, int32_t n_labels,
                                             char* out, int64_t cap, int64_t* feat_start, int32_t threads) {
  if (n < 0) return NUHTC_E_INVALID;
  if (n == 0) return 0;
  Pieces pc;
  if (!verts || !ring_off || !label || !score || !feat_start || !pc.init(head, mid, tail, n_labels)) return NUHTC_E_INVALID;
  for (int64_t i = 0; i < n; ++i)
    if (label[i] < 0 || label[i] >= n_labels || ring_off[i + 1] < ring_off[i]) return NUHTC_E_INVALID;
  // pass 1: the exact length of every record (its separator included) -> feat_start by a prefix sum
  parallel_for(n, threads, [&](int64_t lo, int64_t hi) {
    char tmp[48];
    for (int64_t i = lo; i < hi; ++i) {
      const int l = label[i];
      int64_t len = (int64_t)pc.lh + pc.lm[l] + pc.lt[l] + py_repr(score[i], tmp) + 2;
      const int64_t nv = ring_off[i + 1] - ring_off[i];
      for (int64_t v = ring_off[i]; v < ring_off[i + 1]; ++v) len += int_len(verts[2 * v]) + int_len(verts[2 * v + 1]);
      len += 4 * nv + (nv > 1 ? 2 * (nv - 1) : 0);        // "[", ", ", "]" per vertex; ", " between vertices
      feat_start[i + 1] = len;
    }
  });
  feat_start[0] = 0;
  for (int64_t i = 0; i < n; ++i) feat_start[i + 1] += feat_start[i];
  const int64_t total = feat_start[n] - 2;                  // no separator after the last record
  if (!out || cap < total) return total;                    // sizing call: nothing written
  parallel_for(n, threads, [&](int64_t lo, int64_t hi) {
    for (int64_t i = lo; i < hi; ++i) {
      const int l = label[i];
      char* o = out + feat_start[i];
      memcpy(o, pc.head, pc.lh); o += pc.lh;
      for (int64_t v = ring_off[i]; v < ring_off[i + 1]; ++v) {
        if (v > ring_off[i]) { *o++ = ','; *o++ = ' '; }
        *o++ = '[';
        o = put_int(o, verts[2 * v]);
        *o++ = ','; *o++ = ' ';
        o = put_int(o, verts[2 * v + 1]);
        *o++ = ']';
      }
      memcpy(o, pc.mid[l], pc.lm[l]); o += pc.lm[l];
      o += py_repr(score[i], o);
      memcpy(o, pc.tail[l], pc.lt[l]); o += pc.lt[l];
      if (i + 1 < n) { *o++ = ','; *o++ = ' '; }
    }
  });
  return total;
}

extern "C" int64_t nuhtc_write_point_features(const double* xy, const int32_t* label, const double* score, int64_t n,
                                              const char* head, const char* const* mid, const char* const* tail, int32_t n_labels,
                                              char* out, int64_t cap) {
  if (n < 0) return NUHTC_E_INVALID;
  if (n == 0) return 0;
  Pieces pc;
  if (!xy || !label || !score || !pc.init(head, mid, tail, n_labels)) return NUHTC_E_INVALID;
  size_t mx = 0;
  for (int l = 0; l < n_labels; ++l) mx = pc.lm[l] + pc.lt[l] > mx ? pc.lm[l] + pc.lt[l] : mx;
  const int64_t bound = n * (int64_t)(pc.lh + mx + 3 * 32 + 4);     // a double's text is at most 24 bytes
  if (!out || cap < bound) return bound;
  char* o = out;
  for (int64_t i = 0; i < n; ++i) {
    const int l = label[i];
    if (l < 0 || l >= n_labels) return NUHTC_E_INVALID;
    memcpy(o, pc.head, pc.lh); o += pc.lh;
    o += py_repr(xy[2 * i], o);
    *o++ = ','; *o++ = ' ';
    o += py_repr(xy[2 * i + 1], o);
    memcpy(o, pc.mid[l], pc.lm[l]); o += pc.lm[l];
    o += py_repr(score[i], o);
    memcpy(o, pc.tail[l], pc.lt[l]); o += pc.lt[l];
    if (i + 1 < n) { *o++ = ','; *o++ = ' '; }
  }
  return o - out;
}

extern "C" int64_t nuhtc_join_features(const char* text, const int64_t* feat_start, const int64_t* pick, int64_t n_pick, char* out, int64_t cap,
                                       int32_t threads) {
  if (n_pick < 0) return NUHTC_E_INVALID;
  if (n_pick == 0) return 0;
  if (!text || !feat_start || !pick) return NUHTC_E_INVALID;
  std::vector<int64_t> dst((size_t)n_pick + 1);
  dst[0] = 0;
  for (int64_t k = 0; k < n_pick; ++k) {
    const int64_t len = feat_start[pick[k] + 1] - feat_start[pick[k]];      // the record and a separator
    if (len < 2) return NUHTC_E_INVALID;
    dst[k + 1] = dst[k] + len;
  }
  const int64_t total = dst[n_pick] - 2;
  if (!out || cap < total) return total;
  parallel_for(n_pick, threads, [&](int64_t lo, int64_t hi) {
    for (int64_t k = lo; k < hi; ++k) {
      const int64_t len = dst[k + 1] - dst[k] - 2;
      memcpy(out + dst[k], text + feat_start[pick[k]], (size_t)len);
      if (k + 1 < n_pick) { out[dst[k] + len] = ','; out[dst[k] + len + 1] = ' '; }
    }
  });
  return total;
}

// ---- rings of a written GeoJSON back into mask crops (tools/nuclei_merge.py on the GPU).  A ring infer_wsi.py writes is the traced outer border of
// one 8-connected pixel component (cv2.findContours(...)[0][0], tools/infer_wsi.py:51-58): vertices on pixel centres, edges along the 8 chain
// directions.  The pixels inside or on it are that component with its holes filled -- exactly the set nuhtc_merge_overlap derives from a
// detection's mask crop before it measures polygons (csrc/merge.hip) -- so filling the ring gives the merge the input the masks would have given.
// Fill: the border is drawn (integer steps along every edge), the outside is flooded from a one-pixel frame through 4-neighbours (an 8-connected
// border cannot be crossed that way), everything not outside is the component.
extern "C" int64_t nuhtc_fill_rings(const int32_t* verts, const int64_t* ring_off, int64_t n, int32_t* boxes, int32_t* areas, int64_t* word_off,
                                    uint32_t* bits, int64_t cap_words, int32_t threads) {
  if (n < 0) return NUHTC_E_INVALID;
  if (n == 0) return 0;
  if (!verts || !ring_off || !boxes || !areas || !word_off) return NUHTC_E_INVALID;
  int64_t total = 0;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t a = ring_off[i], b = ring_off[i + 1];
    if (b <= a) return NUHTC_E_INVALID;
    int32_t x0 = verts[2 * a], x1 = x0, y0 = verts[2 * a + 1], y1 = y0;
    for (int64_t v = a; v < b; ++v) {
      const int32_t x = verts[2 * v], y = verts[2 * v + 1];
      x0 = x < x0 ? x : x0; x1 = x > x1 ? x : x1; y0 = y < y0 ? y : y0; y1 = y > y1 ? y : y1;
      const int64_t nx = v + 1 < b ? v + 1 : a;
      const int64_t dx = (int64_t)verts[2 * nx] - x, dy = (int64_t)verts[2 * nx + 1] - y;
      if (dx != 0 && dy != 0 && (dx < 0 ? -dx : dx) != (dy < 0 ? -dy : dy)) return NUHTC_E_INVALID;      // not a chain-direction edge: not a traced ring
    }
    if ((int64_t)x1 - x0 > 65535 || (int64_t)y1 - y0 > 65535) return NUHTC_E_INVALID;
    boxes[4 * i] = x0; boxes[4 * i + 1] = y0; boxes[4 * i + 2] = x1 + 1; boxes[4 * i + 3] = y1 + 1;
    word_off[i] = total;
    total += (int64_t)(y1 - y0 + 1) * ((x1 - x0 + 1 + 31) / 32);
  }
  if (!bits || cap_words < total) return total;
  parallel_for(n, threads, [&](int64_t lo, int64_t hi) {
    std::vector<uint8_t> g;
    std::vector<int32_t> stack;
    for (int64_t i = lo; i < hi; ++i) {
      const int32_t x0 = boxes[4 * i], y0 = boxes[4 * i + 1];
      const int w = boxes[4 * i + 2] - x0, h = boxes[4 * i + 3] - y0, W = w + 2, H = h + 2;
      g.assign((size_t)W * H, 0);
      const int64_t a = ring_off[i], b = ring_off[i + 1];
      for (int64_t v = a; v < b; ++v) {                        // the border, edge by edge (both end points included)
        const int64_t nx = v + 1 < b ? v + 1 : a;
        int x = verts[2 * v] - x0 + 1, y = verts[2 * v + 1] - y0 + 1;
        const int ex = verts[2 * nx] - x0 + 1, ey = verts[2 * nx + 1] - y0 + 1;
        const int sx = (ex > x) - (ex < x), sy = (ey > y) - (ey < y);
        g[(size_t)y * W + x] = 1;
        while (x != ex || y != ey) { x += sx; y += sy; g[(size_t)y * W + x] = 1; }
      }
      stack.clear();
      stack.push_back(0);
      g[0] = 2;
      while (!stack.empty()) {                                  // the outside, from the frame, through 4-neighbours
        const int p = stack.back();
        stack.pop_back();
        const int px = p % W, py = p / W;
        if (px > 0 && !g[p - 1]) { g[p - 1] = 2; stack.push_back(p - 1); }
        if (px + 1 < W && !g[p + 1]) { g[p + 1] = 2; stack.push_back(p + 1); }
        if (py > 0 && !g[p - W]) { g[p - W] = 2; stack.push_back(p - W); }
        if (py + 1 < H && !g[p + W]) { g[p + W] = 2; stack.push_back(p + W); }
      }
      const int wpr = (w + 31) / 32;
      uint32_t* dst = bits + word_off[i];
      int32_t area = 0;
      for (int y = 0; y < h; ++y)
        for (int k = 0; k < wpr; ++k) {
          uint32_t word = 0;
          for (int t = 0; t < 32 && 32 * k + t < w; ++t)
            if (g[(size_t)(y + 1) * W + 32 * k + t + 1] != 2) { word |= 1u << t; ++area; }
          dst[(size_t)y * wpr + k] = word;
        }
      areas[i] = area;
    }
  });
  return total;
}
