// Proposal generation on the device (gfx950, wave64).  Compiled with -ffp-contract=off: box arithmetic follows the
// reference's separate float32 mul/add steps so that threshold decisions match the CPU path.
//   rpn_level_kernel : sigmoid, per-level top-k by (score desc, index asc) via radix select + LDS bitonic sort,
//                      anchor generation, delta2bbox, clamp, min-size filter
//                      (mmdet/models/dense_heads/rpn_head.py:150-229, core/anchor/anchor_generator.py:151-281,
//                       core/bbox/coder/delta_xywh_bbox_coder.py:230-260)
//   nms_*            : mmcv batched_nms / nms (rpn_head.py:232, nuhtc/models/bbox_head.py:93): per-id coordinate
//                      offset in f32, stable sort by score, 64x64 bit-mask IoU matrix, chunked greedy reduce
//   cc_*             : "watershed" proposals (nuhtc/models/htc_roi_head_cus.py:283-342) = upsample x4 + 5x5
//                      Gaussian + >0 + open(5x5,2) + hole fill + 4-connected components in raster order + boxes
//                      (the watershed call is an identity on these inputs)
#include "common.h"
#include "proposals.h"

typedef unsigned long long u64;

__device__ __forceinline__ unsigned f2key(float f) { return __float_as_uint(f); }   // scores are >= 0: uint order == float order

// in-LDS bitonic sort of n (power of two) u64 keys, ascending, by the whole block
__device__ void bitonic_sort_lds(u64* keys, int n) {
  for (int k = 2; k <= n; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < (n >> 1); t += blockDim.x) {
        int lo = ((t / j) * (j << 1)) + (t % j);   // t -> pair (lo, lo+j)
        int hi = lo + j;
        bool asc = ((lo & k) == 0);
        u64 a = keys[lo], b = keys[hi];
        if ((a > b) == asc) { keys[lo] = b; keys[hi] = a; }
      }
      __syncthreads();
    }
  }
}

__device__ __forceinline__ int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

// block-wide exclusive scan of one int per thread (blockDim.x == 1024), returns exclusive prefix, *total gets the sum
__device__ int block_exscan_1024(int v, int* lds16, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
  if (lane == 63) lds16[wave] = incl;
  __syncthreads();
  if (wave == 0) {
    int w = lane < 16 ? lds16[lane] : 0;
    int wi = w;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { int t = __shfl_up(wi, o); if (lane >= o) wi += t; }
    if (lane < 16) lds16[lane] = wi - w;
    if (lane == 15) lds16[16] = wi;
  }
  __syncthreads();
  int res = lds16[wave] + incl - v;
  *total = lds16[16];
  __syncthreads();
  return res;
}

// ------------------------------------------------------------------------------------------- RPN per-level selection
__global__ __launch_bounds__(1024) void rpn_level_kernel(RpnLevels lv, RpnSelParams p) {
  __shared__ u64 keys[4096];
  __shared__ int hist[256];
  __shared__ int sc16[17];
  __shared__ unsigned s_prefix;
  __shared__ int s_need, s_cnt, s_eq;
  const int b = blockIdx.x, L = blockIdx.y;
  const int tid = threadIdx.x;
  const int h = lv.h[L], w = lv.w[L], stride = lv.stride[L];
  const int n = h * w * 3;
  const float* out = lv.out[L] + (long long)b * h * w * 32;
  const int k = p.nms_pre;
  auto score_of = [&](int idx) -> float {
    int pix = idx / 3, a = idx - pix * 3;
    float x = out[(long long)pix * 32 + a];
    return 1.0f / (1.0f + expf(-x));
  };
  int nsel;
  // Keys of this (image, level) in registers: RPN_KPT per thread cover 49 152 anchors (level 0 of a 512 x 512 input); the loads of
  // the strided logits are independent and issued together, and the four counting passes never go back to memory (one
  // outstanding load per thread and pass was what this kernel spent its time on).  Larger levels use the scratch row.
  constexpr int RPN_KPT = 48;
  const bool in_regs = n <= RPN_KPT * 1024;
  unsigned myk[RPN_KPT];
  if (n > k) {
    unsigned* kk = p.keys + (long long)(b * 4 + L) * p.key_stride;
    if (in_regs) {
#pragma unroll
      for (int j = 0; j < RPN_KPT; ++j) {
        const int idx = tid + j * 1024;
        myk[j] = idx < n ? f2key(score_of(idx)) : 0u;
      }
    } else {
      for (int idx = tid; idx < n; idx += 1024) kk[idx] = f2key(score_of(idx));
      __syncthreads();
    }
    // radix select: find key T of rank k (descending) over 32-bit score keys
    unsigned prefix = 0, maskbits = 0;
    int need = k;   // still to take from the candidates matching `prefix` under `maskbits`
    // RPN scores crowd into a few bins (most anchors score near 0): every wave first adds one count for each of its two most
    // crowded bins, the rest is spread over the bins (the low-byte passes)
    auto count = [&](unsigned key, bool in, int shift) {
      const unsigned bin = (key >> shift) & 255;
      bool mine = in;
#pragma unroll
      for (int round = 0; round < 2; ++round) {
        const u64 todo = __ballot(mine);
        if (!todo) break;
        const int leader = __ffsll((long long)todo) - 1;
        const unsigned b0 = __shfl(bin, leader);
        const u64 grp = __ballot(mine && bin == b0);
        if ((tid & 63) == leader) atomicAdd(&hist[b0], __popcll(grp));
        mine = mine && bin != b0;
      }
      if (mine) atomicAdd(&hist[bin], 1);
    };
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      for (int i = tid; i < 256; i += 1024) hist[i] = 0;
      __syncthreads();
      if (in_regs) {
#pragma unroll
        for (int j = 0; j < RPN_KPT; ++j) {
          const int idx = tid + j * 1024;
          if (j * 1024 < n) count(myk[j], idx < n && (myk[j] & maskbits) == prefix, shift);
        }
      } else {
        for (int base = 0; base < n; base += 1024) {
          const int idx = base + tid;
          const unsigned key = idx < n ? kk[idx] : 0u;
          count(key, idx < n && (key & maskbits) == prefix, shift);
        }
      }
      __syncthreads();
      if (tid == 0) {
        int acc = 0, bin = 255;
        for (; bin >= 0; --bin) {
          if (acc + hist[bin] >= need) break;
          acc += hist[bin];
        }
        if (bin < 0) bin = 0;
        s_prefix = prefix | ((unsigned)bin << shift);
        s_need = need - acc;
        s_eq = hist[bin];               // after the last pass: how many keys equal the rank-k key
      }
      __syncthreads();
      prefix = s_prefix;
      need = s_need;
      maskbits |= 0xFFu << shift;
      __syncthreads();
    }
    const unsigned T = prefix;   // keys > T are all taken; `need` of the keys == T, lowest index first
    if (tid == 0) s_cnt = 0;
    for (int i = tid; i < 4096; i += 1024) keys[i] = ~0ull;
    __syncthreads();
    if ((k - need) + s_eq <= 4096) {
      // usual case: every key >= T fits the sort image; (key desc, index asc) order puts the `need` lowest-index ties
      // first, so the list is simply cut at k after the sort.  One counter bump per wave.
      auto emit = [&](int idx, unsigned key) {
        const bool take = idx < n && key >= T;
        const u64 m = __ballot(take);
        int pos0 = 0;
        if ((tid & 63) == 0 && m) pos0 = atomicAdd(&s_cnt, __popcll(m));
        pos0 = __shfl(pos0, 0);
        if (take) keys[pos0 + __popcll(m & ((1ull << (tid & 63)) - 1ull))] = ((u64)(~key) << 32) | (unsigned)idx;
      };
      if (in_regs) {
#pragma unroll
        for (int j = 0; j < RPN_KPT; ++j)
          if (j * 1024 < n) emit(tid + j * 1024, myk[j]);
      } else {
        for (int base = 0; base < n; base += 1024) emit(base + tid, base + tid < n ? kk[base + tid] : 0u);
      }
    } else {
      // a plateau of equal scores wider than the image: take exactly `need` of them by index rank (ordered block scans)
      int eq_before = 0;
      auto emit = [&](int idx, unsigned key) {
        int is_eq = (idx < n && key == T) ? 1 : 0;
        int tot;
        int rank = block_exscan_1024(is_eq, sc16, &tot) + eq_before;
        bool take = idx < n && (key > T || (is_eq && rank < need));
        if (take) {
          int pos = atomicAdd(&s_cnt, 1);
          keys[pos] = ((u64)(~key) << 32) | (unsigned)idx;
        }
        eq_before += tot;
      };
      if (in_regs) {
#pragma unroll
        for (int j = 0; j < RPN_KPT; ++j)
          if (j * 1024 < n) emit(tid + j * 1024, myk[j]);
      } else {
        for (int base = 0; base < n; base += 1024) emit(base + tid, base + tid < n ? kk[base + tid] : 0u);
      }
    }
    __syncthreads();
    nsel = k;       // s_cnt >= k entries were written; the sort leaves the k selected ones in front
    bitonic_sort_lds(keys, 4096);
  } else {
    // n <= nms_pre: the reference does not sort (rpn_head.py:167) -> index order
    for (int i = tid; i < 4096; i += 1024) keys[i] = i < n ? (((u64)(~f2key(score_of(i)))) << 32 | (unsigned)i) : ~0ull;
    __syncthreads();
    nsel = n;
  }
  // decode + min-size filter, ordered compaction into the level's slot
  const float ratios[3] = {0.5f, 1.0f, 2.0f};
  float* cb = p.cand_boxes + ((long long)(b * 4 + L) * p.slot) * 4;
  float* cs = p.cand_scores + (long long)(b * 4 + L) * p.slot;
  int written = 0;
  for (int base = 0; base < nsel; base += 1024) {
    int pos = base + tid;
    bool valid = false;
    float x1 = 0, y1 = 0, x2 = 0, y2 = 0, score = 0;
    if (pos < nsel) {
      u64 key = keys[pos];
      int idx = (int)(key & 0xFFFFFFFFu);
      score = __uint_as_float(~(unsigned)(key >> 32));
      int pix = idx / 3, a = idx - pix * 3;
      int py = pix / w, px = pix - py * w;
      // AnchorGenerator: scales [4], ratios (.5,1,2), centre offset 0 (anchor_generator.py:151-194)
      float hr = sqrtf(ratios[a]);
      float wr = 1.0f / hr;
      float wsz = (float)stride * wr * 4.0f, hsz = (float)stride * hr * 4.0f;
      float sx = (float)px * (float)stride, sy = (float)py * (float)stride;
      float ax1 = -0.5f * wsz + sx, ay1 = -0.5f * hsz + sy, ax2 = 0.5f * wsz + sx, ay2 = 0.5f * hsz + sy;
      const float* d = out + (long long)pix * 32 + 3 + 4 * a;
      float dx = d[0], dy = d[1], dw = d[2], dh = d[3];
      const float MR = 4.135166556742356f;   // |log(16/1000)|
      dw = fminf(fmaxf(dw, -MR), MR);
      dh = fminf(fmaxf(dh, -MR), MR);
      float pxc = (ax1 + ax2) * 0.5f, pyc = (ay1 + ay2) * 0.5f;
      float pw = ax2 - ax1, ph = ay2 - ay1;
      float gx = pxc + pw * dx, gy = pyc + ph * dy;
      float gw = pw * expf(dw), gh = ph * expf(dh);
      x1 = gx - gw * 0.5f; y1 = gy - gh * 0.5f; x2 = gx + gw * 0.5f; y2 = gy + gh * 0.5f;
      x1 = fminf(fmaxf(x1, 0.f), (float)p.img_w); x2 = fminf(fmaxf(x2, 0.f), (float)p.img_w);
      y1 = fminf(fmaxf(y1, 0.f), (float)p.img_h); y2 = fminf(fmaxf(y2, 0.f), (float)p.img_h);
      valid = (x2 - x1 > p.min_size) && (y2 - y1 > p.min_size);
    }
    int tot;
    int off = block_exscan_1024(valid ? 1 : 0, sc16, &tot) + written;
    if (valid) {
      cb[off * 4 + 0] = x1; cb[off * 4 + 1] = y1; cb[off * 4 + 2] = x2; cb[off * 4 + 3] = y2;
      cs[off] = score;
    }
    written += tot;
  }
  if (tid == 0) p.cand_count[b * 4 + L] = written;
}

int launch_rpn_select(const RpnLevels& lv, const RpnSelParams& p, int B, hipStream_t s) {
  { static const int& skip_ = dev_knob_ref("SKIP", 0); if (skip_ & 128) return 0; }   // dev: ablation of the step (tools/dev/r04_ablate.py)
  ProfScope ps("rpn_select", 0, 0, s);
  if (p.nms_pre > 4096 || p.slot < p.nms_pre || !p.keys) return NUHTC_E_INVALID;
  for (int l = 0; l < 4; ++l)
    if (lv.h[l] * lv.w[l] * 3 > p.key_stride) return NUHTC_E_INVALID;
  hipLaunchKernelGGL(rpn_level_kernel, dim3(B, 4), dim3(1024), 0, s, lv, p);
  return hipGetLastError() == hipSuccess ? 0 : NUHTC_E_HIP;
}

// ------------------------------------------------------------------------------------------- NMS
// Gather the per-group slots of image b into one ordered candidate list (group-major), apply the batched_nms
// coordinate offset (id * (max_coord + 1), float32) and sort by (score desc, position asc).
__global__ __launch_bounds__(1024) void nms_prepare_kernel(NmsParams p) {
  extern __shared__ u64 keys[];      // npad_max entries
  __shared__ float red[16];
  __shared__ float s_max;
  const int b = blockIdx.x, tid = threadIdx.x;
  // group offsets
  int goff[NMS_MAX_GROUPS + 1];
  goff[0] = 0;
  for (int g = 0; g < p.n_groups; ++g) goff[g + 1] = goff[g] + p.group_count[b * p.n_groups + g];
  const int n = goff[p.n_groups];
  const int npad = next_pow2(n < 2 ? 2 : n);
  // max coordinate over all candidate boxes (boxes.max() of mmcv batched_nms)
  float mx = -3.0e38f;
  for (int i = tid; i < n; i += 1024) {
    int g = 0;
    while (i >= goff[g + 1]) ++g;
    const float* bx = p.boxes + ((long long)(b * p.n_groups + g) * p.slot + (i - goff[g])) * 4;
    mx = fmaxf(mx, fmaxf(fmaxf(bx[0], bx[1]), fmaxf(bx[2], bx[3])));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  if (tid == 0) { float m = red[0]; for (int i = 1; i < 16; ++i) m = fmaxf(m, red[i]); s_max = m; }
  __syncthreads();
  const float maxp1 = s_max + 1.0f;
  for (int i = tid; i < npad; i += 1024) {
    u64 key = ~0ull;
    if (i < n) {
      int g = 0;
      while (i >= goff[g + 1]) ++g;
      float sc = p.scores[(long long)(b * p.n_groups + g) * p.slot + (i - goff[g])];
      key = ((u64)(~f2key(sc)) << 32) | (unsigned)i;
    }
    keys[i] = key;
  }
  __syncthreads();
  bitonic_sort_lds(keys, npad);
  for (int r = tid; r < n; r += 1024) {
    int i = (int)(keys[r] & 0xFFFFFFFFu);
    int g = 0;
    while (i >= goff[g + 1]) ++g;
    long long src = (long long)(b * p.n_groups + g) * p.slot + (i - goff[g]);
    const float* bx = p.boxes + src * 4;
    int id = p.ids ? p.ids[src] : g;
    float off = (float)id * maxp1;
    float* sb = p.sorted_boxes + ((long long)b * p.cap + r) * 4;
    sb[0] = bx[0] + off; sb[1] = bx[1] + off; sb[2] = bx[2] + off; sb[3] = bx[3] + off;
    p.sorted_src[(long long)b * p.cap + r] = (int)src;
  }
  if (tid == 0) p.n_total[b] = n;
}

// mmcv nms_cuda: thread t of block (cb, rb) compares row rb*64+t with the 64 boxes of column block cb
__global__ __launch_bounds__(64) void nms_mask_kernel(NmsParams p) {
  const int b = blockIdx.z, rb = blockIdx.y, cb = blockIdx.x;
  const int n = p.n_total[b];
  if (cb < rb || rb * 64 >= n || cb * 64 >= n) return;
  __shared__ float4 cbx[64];
  const float4* sb = reinterpret_cast<const float4*>(p.sorted_boxes) + (long long)b * p.cap;
  const int t = threadIdx.x;
  const int ncol = min(n - cb * 64, 64);
  if (t < ncol) cbx[t] = sb[cb * 64 + t];
  __syncthreads();
  const int row = rb * 64 + t;
  if (row >= n) return;
  const float4 a = sb[row];
  const float sa = (a.z - a.x) * (a.w - a.y);
  u64 bits = 0;
  const int start = (rb == cb) ? t + 1 : 0;
  for (int j = start; j < ncol; ++j) {
    const float4 c = cbx[j];
    float left = fmaxf(a.x, c.x), right = fminf(a.z, c.z);
    float top = fmaxf(a.y, c.y), bottom = fminf(a.w, c.w);
    float wdt = fmaxf(right - left, 0.f), hgt = fmaxf(bottom - top, 0.f);
    float inter = wdt * hgt;
    float sb2 = (c.z - c.x) * (c.w - c.y);
    float ovr = inter / (sa + sb2 - inter);
    if (ovr > p.iou_thr) bits |= 1ull << j;
  }
  p.mask[((long long)b * p.cap + row) * (p.cap / 64) + cb] = bits;
}

// greedy reduce, 64 sorted rows at a time; stops after max_keep kept rows
__global__ __launch_bounds__(256) void nms_reduce_kernel(NmsParams p) {
  __shared__ u64 removed[NMS_MAX_CAP / 64];
  __shared__ u64 s_keepbits;
  __shared__ int s_kept;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int n = p.n_total[b];
  const int nw = p.cap / 64;
  const int nchunks = (n + 63) / 64;
  for (int i = tid; i < nw; i += 256) removed[i] = 0;
  if (tid == 0) s_kept = 0;
  __syncthreads();
  const u64* mask = p.mask + (long long)b * p.cap * nw;
  for (int c = 0; c < nchunks; ++c) {
    const int kept_before = s_kept;
    if (kept_before >= p.max_keep) break;
    if (tid < 64) {
      const int row = c * 64 + lane;
      u64 d = row < n ? mask[(long long)row * nw + c] : 0ull;   // upper-triangular diagonal word
      u64 cur = removed[c];
      u64 keep = 0;
      int room = p.max_keep - kept_before;
      const int rows_here = min(64, n - c * 64);
      for (int i = 0; i < rows_here; ++i) {
        u64 di = __shfl(d, i);
        if (!((cur >> i) & 1ull) && room > 0) { keep |= 1ull << i; cur |= di; --room; }
      }
      if (lane == 0) s_keepbits = keep;
      // emit kept rows in order
      if ((keep >> lane) & 1ull) {
        int k = kept_before + __popcll(keep & ((1ull << lane) - 1ull));
        int src = p.sorted_src[(long long)b * p.cap + row];
        const float* bx = p.boxes + (long long)src * 4;
        float* o = p.out_dets + ((long long)b * p.max_keep + k) * 5;
        o[0] = bx[0]; o[1] = bx[1]; o[2] = bx[2]; o[3] = bx[3]; o[4] = p.scores[src];
        p.out_src[(long long)b * p.max_keep + k] = src;
      }
      if (lane == 0) s_kept = kept_before + __popcll(keep);
    }
    __syncthreads();
    const u64 keep = s_keepbits;
    if (keep) {
      for (int wv = c + 1 + tid; wv < nw; wv += 256) {
        u64 acc = 0;
        u64 kb = keep;
        while (kb) {
          int i = __ffsll((long long)kb) - 1;
          kb &= kb - 1;
          acc |= mask[(long long)(c * 64 + i) * nw + wv];
        }
        removed[wv] |= acc;
      }
    }
    __syncthreads();
  }
  if (tid == 0) p.out_counts[b] = s_kept < p.max_keep ? s_kept : p.max_keep;
}

// rows of `mask` that nms_mask_kernel never writes must read as zero: words cb < rb are unused by the reduce (it only
// reads words >= its chunk), words beyond n likewise; so no clearing pass is needed.
int launch_nms(const NmsParams& p, int B, hipStream_t s) {
  { static const int& skip_ = dev_knob_ref("SKIP", 0); if (skip_ & 128) return 0; }   // dev: ablation of the step (tools/dev/r04_ablate.py)
  ProfScope ps("nms", 0, 0, s);
  if (p.cap % 64 || p.cap > NMS_MAX_CAP || p.n_groups > NMS_MAX_GROUPS) return NUHTC_E_INVALID;
  size_t lds = (size_t)p.cap_pow2 * sizeof(u64);
  hipLaunchKernelGGL(nms_prepare_kernel, dim3(B), dim3(1024), lds, s, p);
  hipLaunchKernelGGL(nms_mask_kernel, dim3(p.cap / 64, p.cap / 64, B), dim3(64), 0, s, p);
  hipLaunchKernelGGL(nms_reduce_kernel, dim3(B), dim3(256), 0, s, p);
  return hipGetLastError() == hipSuccess ? 0 : NUHTC_E_HIP;
}

// ---- level-wise variant ------------------------------------------------------------------------------------------------
// mmcv batched_nms offsets the boxes of each group (RPN level) so that groups never overlap: the greedy pass over the
// global score order is exactly one independent greedy pass per group, followed by a merge of the survivors by score.
// Running the groups separately cuts the pair tests from n^2/2 to sum(n_g^2)/2 (3.5x for 3000+3000+3000+768), runs the
// serial reduce chains of the groups side by side, and keeps the result identical (same offset boxes, same IoU test, same
// (score desc, position asc) order, same first max_keep survivors).
__global__ __launch_bounds__(1024) void nms_prepare_levels_kernel(NmsParams p) {
  extern __shared__ u64 keys[];      // pow2 >= slot entries: one group per block
  __shared__ float red[16];
  __shared__ float s_max;
  const int b = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
  int goff[NMS_MAX_GROUPS + 1], astart[NMS_MAX_GROUPS + 1];
  goff[0] = astart[0] = 0;
  for (int h = 0; h < p.n_groups; ++h) {
    const int c = p.group_count[b * p.n_groups + h];
    goff[h + 1] = goff[h] + c;
    astart[h + 1] = astart[h] + ((c + 63) & ~63);
  }
  const int n = goff[p.n_groups];
  // max coordinate over the candidate boxes of ALL groups (boxes.max() of mmcv batched_nms)
  float mx = -3.0e38f;
  for (int i = tid; i < n; i += 1024) {
    int h = 0;
    while (i >= goff[h + 1]) ++h;
    const float* bx = p.boxes + ((long long)(b * p.n_groups + h) * p.slot + (i - goff[h])) * 4;
    mx = fmaxf(mx, fmaxf(fmaxf(bx[0], bx[1]), fmaxf(bx[2], bx[3])));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  if (tid == 0) { float m = red[0]; for (int i = 1; i < 16; ++i) m = fmaxf(m, red[i]); s_max = m; }
  __syncthreads();
  const float off = (float)g * (s_max + 1.0f);
  // this group's candidates by (score desc, position asc)
  const int cnt = goff[g + 1] - goff[g];
  const int npad = next_pow2(cnt < 2 ? 2 : cnt);
  const long long base = (long long)(b * p.n_groups + g) * p.slot;
  for (int i = tid; i < npad; i += 1024) keys[i] = i < cnt ? (((u64)(~f2key(p.scores[base + i]))) << 32 | (unsigned)i) : ~0ull;
  __syncthreads();
  bitonic_sort_lds(keys, npad);
  for (int r = tid; r < cnt; r += 1024) {
    const int i = (int)(keys[r] & 0xFFFFFFFFu);
    const long long src = base + i;
    const float* bx = p.boxes + src * 4;
    const long long a = (long long)b * p.cap + astart[g] + r;
    float* sb = p.sorted_boxes + a * 4;
    sb[0] = bx[0] + off; sb[1] = bx[1] + off; sb[2] = bx[2] + off; sb[3] = bx[3] + off;
    p.sorted_src[a] = (int)src;
    p.sorted_pos[a] = goff[g] + i;      // group-major position: the tie-break of the reference's single global sort
  }
  if (tid == 0) {
    p.seg_start[b * p.n_groups + g] = astart[g];
    p.seg_n[b * p.n_groups + g] = cnt;
    if (g == 0) p.n_total[b] = astart[p.n_groups];
  }
}

// One wave per 64 x 64 tile of a group's upper triangle.  blockIdx.x enumerates only those tiles (group after group, row after
// row): the full (cap/64)^2 grid was 374 k one-wave workgroups per launch of which 85 % returned at once, and that many
// dispatches starved whatever ran beside it on the other stream.  Column areas are staged with the boxes; a column that no
// row of the tile touches is skipped before the division (inter == 0 gives 0 > thr false whatever the union is).
__global__ __launch_bounds__(256) void nms_mask_levels_kernel(NmsParams p) {
  // four waves per workgroup, each with a tile of its own: a quarter of the dispatches of the one-wave form, whose 72 k workgroups per
  // launch kept the dispatcher busy enough to slow the semantic branch's kernels on the other stream several-fold while it ran
  const int b = blockIdx.z, wv4 = threadIdx.x >> 6;
  int t = blockIdx.x * 4 + wv4, rs = 0, rn = 0, nc = 0, g = 0;
  for (; g < p.n_groups; ++g) {
    rs = p.seg_start[b * p.n_groups + g];
    rn = p.seg_n[b * p.n_groups + g];
    nc = (rn + 63) >> 6;
    const int np = nc * (nc + 1) / 2;
    if (t < np) break;
    t -= np;
  }
  if (g == p.n_groups) return;
  int r = 0;
  while (t >= nc - r) { t -= nc - r; ++r; }
  const int rb = (rs >> 6) + r, cb = rb + t;               // segments start on multiples of 64
  const int end = rs + rn;                                // one past the group's last row
  __shared__ float4 cbx4[4][64];
  __shared__ float car4[4][64];
  float4* cbx = cbx4[wv4];
  float* car = car4[wv4];
  const float4* sb = reinterpret_cast<const float4*>(p.sorted_boxes) + (long long)b * p.cap;
  const int lane = threadIdx.x & 63;
  const int ncol = min(end - cb * 64, 64);
  if (ncol <= 0) return;
  if (lane < ncol) {
    const float4 c = sb[cb * 64 + lane];
    cbx[lane] = c;
    car[lane] = (c.z - c.x) * (c.w - c.y);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");      // the wave reads back only what it wrote itself
  __builtin_amdgcn_wave_barrier();
  const int row = rb * 64 + lane;
  const bool live = row < end;
  const float4 a = sb[live ? row : end - 1];
  const float sa = (a.z - a.x) * (a.w - a.y);
  u64 bits = 0;
  const int start = (rb == cb) ? lane + 1 : 0;
  for (int j = (rb == cb) ? 1 : 0; j < ncol; ++j) {
    const float4 c = cbx[j];
    float left = fmaxf(a.x, c.x), right = fminf(a.z, c.z);
    float top = fmaxf(a.y, c.y), bottom = fminf(a.w, c.w);
    float wdt = fmaxf(right - left, 0.f), hgt = fmaxf(bottom - top, 0.f);
    float inter = wdt * hgt;
    const bool mine = live && j >= start;
    if (__ballot(mine && inter > 0.f) == 0) continue;
    float ovr = inter / (sa + car[j] - inter);
    if (mine && ovr > p.iou_thr) bits |= 1ull << j;
  }
  if (live) p.mask[((long long)b * p.cap + row) * (p.cap / 64) + cb] = bits;
}

// greedy reduce of one group (block = (image, group)): survivors of each 64-row chunk -> keepbits; at most max_keep per group
__global__ __launch_bounds__(256) void nms_reduce_levels_kernel(NmsParams p) {
  __shared__ u64 removed[NMS_MAX_CAP / 64 + 4];
  __shared__ unsigned char klist[64];
  __shared__ int s_nk;
  __shared__ int s_kept;
  const int b = blockIdx.x, g = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  const int start = p.seg_start[b * p.n_groups + g], n = p.seg_n[b * p.n_groups + g];
  const int nw = p.cap / 64;
  const int c0 = start / 64, nchunks = (n + 63) / 64;
  for (int i = tid; i < nchunks; i += 256) removed[i] = 0;
  if (tid == 0) s_kept = 0;
  __syncthreads();
  const u64* mask = p.mask + (long long)b * p.cap * nw;
  u64* kb_out = p.keepbits + (long long)b * nw;
  for (int c = 0; c < nchunks; ++c) {
    const int kept_before = s_kept;
    if (kept_before >= p.max_keep) {                      // the rest of the group cannot reach the output
      for (int i = c + tid; i < nchunks; i += 256) kb_out[c0 + i] = 0;
      break;
    }
    if (tid < 64) {
      const int row = start + c * 64 + lane;
      u64 d = c * 64 + lane < n ? mask[(long long)row * nw + c0 + c] : 0ull;
      u64 cur = removed[c];
      u64 keep = 0;
      int room = p.max_keep - kept_before;
      const int rows_here = min(64, n - c * 64);
      for (int i = 0; i < rows_here; ++i) {
        u64 di = __shfl(d, i);
        if (!((cur >> i) & 1ull) && room > 0) { keep |= 1ull << i; cur |= di; --room; }
      }
      if ((keep >> lane) & 1ull) klist[__popcll(keep & ((1ull << lane) - 1ull))] = (unsigned char)lane;
      if (lane == 0) { s_nk = __popcll(keep); kb_out[c0 + c] = keep; s_kept = kept_before + __popcll(keep); }
    }
    __syncthreads();
    // OR the mask rows of the kept boxes into the removed set of the later chunks: one (kept row, word) pair per thread
    // step, so all loads of a chunk are independent and in flight together
    const int nk = s_nk, nwr = nchunks - (c + 1);
    for (int idx = tid; idx < nk * nwr; idx += 256) {
      const int i = klist[idx / nwr], wv = c + 1 + idx % nwr;
      const u64 v = mask[(long long)(start + c * 64 + i) * nw + c0 + wv];
      if (v) atomicOr(&removed[wv], v);
    }
    __syncthreads();
  }
}

// survivors of all groups -> global (score desc, position asc) order -> first max_keep rows of the output
__global__ __launch_bounds__(1024) void nms_select_kernel(NmsParams p) {
  __shared__ u64 keys[8192];
  __shared__ int s_n;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) s_n = 0;
  __syncthreads();
  const int nw = p.cap / 64;
  const int nchunks = p.n_total[b] / 64;
  const u64* kb = p.keepbits + (long long)b * nw;
  for (int idx = tid; idx < nchunks * 64; idx += 1024) {
    if ((kb[idx >> 6] >> (idx & 63)) & 1ull) {
      const long long a = (long long)b * p.cap + idx;
      const float sc = p.scores[p.sorted_src[a]];
      const int k = atomicAdd(&s_n, 1);
      if (k < 8192) keys[k] = ((u64)(~f2key(sc)) << 32) | ((u64)(unsigned)p.sorted_pos[a] << 16) | (unsigned)idx;
    }
  }
  __syncthreads();
  const int n = min(s_n, 8192);
  const int npad = next_pow2(n < 2 ? 2 : n);
  for (int i = n + tid; i < npad; i += 1024) keys[i] = ~0ull;
  __syncthreads();
  bitonic_sort_lds(keys, npad);
  const int nout = min(n, p.max_keep);
  for (int k = tid; k < nout; k += 1024) {
    const int idx = (int)(keys[k] & 0xFFFFu);
    const int src = p.sorted_src[(long long)b * p.cap + idx];
    const float* bx = p.boxes + (long long)src * 4;
    float* o = p.out_dets + ((long long)b * p.max_keep + k) * 5;
    o[0] = bx[0]; o[1] = bx[1]; o[2] = bx[2]; o[3] = bx[3]; o[4] = p.scores[src];
    p.out_src[(long long)b * p.max_keep + k] = src;
  }
  if (tid == 0) p.out_counts[b] = nout;
}

int launch_nms_levels(const NmsParams& p, int B, hipStream_t s) {
  { static const int& skip_ = dev_knob_ref("SKIP", 0); if (skip_ & 128) return 0; }   // dev: ablation of the step (tools/dev/r04_ablate.py)
  ProfScope ps("nms", 0, 0, s);
  if (p.cap % 64 || p.cap > NMS_MAX_CAP + 64 * NMS_MAX_GROUPS || p.cap >= 65536 || p.n_groups > NMS_MAX_GROUPS || p.ids ||
      p.n_groups * p.max_keep > 8192 || !p.seg_start || !p.seg_n || !p.sorted_pos || !p.keepbits)
    return NUHTC_E_INVALID;
  int slot_pow2 = 2;
  while (slot_pow2 < p.slot) slot_pow2 <<= 1;
  size_t lds = (size_t)slot_pow2 * sizeof(u64);
  hipLaunchKernelGGL(nms_prepare_levels_kernel, dim3(B, p.n_groups), dim3(1024), lds, s, p);
  const int slot_chunks = (p.slot + 63) / 64;      // a group holds at most `slot` candidates
  hipLaunchKernelGGL(nms_mask_levels_kernel, dim3(cdiv(p.n_groups * (slot_chunks * (slot_chunks + 1) / 2), 4), 1, B), dim3(256), 0, s, p);
  hipLaunchKernelGGL(nms_reduce_levels_kernel, dim3(B, p.n_groups), dim3(256), 0, s, p);
  hipLaunchKernelGGL(nms_select_kernel, dim3(B), dim3(1024), 0, s, p);
  return hipGetLastError() == hipSuccess ? 0 : NUHTC_E_HIP;
}

int nms_set_attributes() {
  // the sort image can exceed the default 64 KiB dynamic-LDS limit (gfx950 has 160 KiB per workgroup)
  hipError_t e = hipFuncSetAttribute((const void*)nms_prepare_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NMS_MAX_CAP * 8);
  if (e == hipSuccess)
    e = hipFuncSetAttribute((const void*)nms_prepare_levels_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NMS_MAX_CAP * 8);
  return e == hipSuccess ? 0 : NUHTC_E_HIP;
}

// ------------------------------------------------------------------------------------------- connected-component proposals
struct Gauss5 { float k[25]; };

// semantic logits (h x w) -> bilinear x4 (align_corners=True) -> 5x5 Gaussian (reflect pad) -> > 0
// One block = a 64 x 32 output tile: the 68 x 36 up-sampled halo is evaluated once into LDS (ten independent evaluations per
// thread, so the four taps of all of them are in flight together), then every thread blurs a column of 8 outputs (same taps, same
// summation order as a plain row-major 5x5 loop: the mask is a sign test, the order is part of the result).
constexpr int CCM_TW = 64, CCM_TH = 32;
__global__ __launch_bounds__(256) void cc_mask_kernel(const float* __restrict__ pred, unsigned char* __restrict__ m, int h, int w, int H,
                                                      int W, Gauss5 gk) {
  __shared__ float up[CCM_TH + 4][CCM_TW + 4 + 1];
  const int b = blockIdx.z;
  const int X0 = blockIdx.x * CCM_TW, Y0 = blockIdx.y * CCM_TH;
  const float* pb = pred + (long long)b * h * w;
  const float sy = h > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
  const float sx = w > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
  constexpr int NE = (CCM_TH + 4) * (CCM_TW + 4);
#pragma unroll 2
  for (int e = threadIdx.x; e < NE; e += 256) {
    int ly = e / (CCM_TW + 4), lx = e - ly * (CCM_TW + 4);
    int Y = Y0 + ly - 2, X = X0 + lx - 2;
    // reflect (no edge repeat): -1 -> 1, H -> H-2
    if (Y < 0) Y = -Y; if (Y >= H) Y = 2 * H - 2 - Y;
    if (X < 0) X = -X; if (X >= W) X = 2 * W - 2 - X;
    Y = Y < 0 ? 0 : Y; X = X < 0 ? 0 : X;                    // halo cells beyond the reflected range (partial tiles) are never used
    float fy = sy * Y, fx = sx * X;
    int y0 = (int)fy, x0 = (int)fx;
    y0 = y0 < h - 1 ? y0 : h - 1; x0 = x0 < w - 1 ? x0 : w - 1;
    int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    float ly1 = fy - y0, lx1 = fx - x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
    up[ly][lx] = ly0 * (lx0 * pb[y0 * w + x0] + lx1 * pb[y0 * w + x1]) + ly1 * (lx0 * pb[y1 * w + x0] + lx1 * pb[y1 * w + x1]);
  }
  __syncthreads();
  const int tx = threadIdx.x & 63, ty = (threadIdx.x >> 6) * 8;
  const int X = X0 + tx;
  if (X >= W) return;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int Y = Y0 + ty + r;
    if (Y >= H) break;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j) acc += gk.k[i * 5 + j] * up[ty + r + i][tx + j];
    m[((long long)b * H + Y) * W + X] = acc > 0.f ? 1 : 0;
  }
}

// 9-tap binary min (erode) / max (dilate) along x or y; outside the image counts as 0
template <int DILATE, int VERT>
__global__ void morph9_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ out, int H, int W, long long total) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  int x = idx % W, y = (idx / W) % H;
  const unsigned char* base = in + (idx - (long long)y * W - x);
  int r = DILATE ? 0 : 1;
#pragma unroll
  for (int d = -4; d <= 4; ++d) {
    int yy = VERT ? y + d : y, xx = VERT ? x : x + d;
    int v = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? base[(long long)yy * W + xx] : 0;
    r = DILATE ? (r | v) : (r & v);
  }
  out[idx] = (unsigned char)r;
}

__device__ __forceinline__ int uf_find(int* lab, int a) {
  int p = lab[a];
  while (p != a) { a = p; p = lab[a]; }
  return a;
}
__device__ __forceinline__ void uf_union(int* lab, int a, int b) {
  while (true) {
    a = uf_find(lab, a);
    b = uf_find(lab, b);
    if (a == b) return;
    if (a < b) { int t = a; a = b; b = t; }   // a > b: hang a under b (roots are component minima)
    int old = atomicMin(&lab[a], b);
    if (old == a) return;
    a = old;
  }
}

// labels of the pixels whose mask value == target (4-connectivity); others -1.  lab is per image (H*W ints).
// Initial label = first pixel of the horizontal run inside the wave's 64-pixel chunk (one ballot per wave), so that
// only chunk-boundary pixels need a union with their left neighbour, and a union with the pixel above is needed only
// where a new overlap of two runs begins (left or upper-left neighbour not in the set).
__global__ void ccl_init_kernel(const unsigned char* __restrict__ m, int* __restrict__ lab, int target, int HW, int W, long long total) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool t = idx < total && m[idx] == target;
  const unsigned long long bits = __ballot(t);
  if (idx >= total) return;
  const int L = threadIdx.x & 63;
  const int p = (int)(idx % HW);
  const int x = p % W;
  int out = -1;
  if (t) {
    const unsigned long long zeros_left = ~bits & ((1ull << L) - 1ull);
    int start = zeros_left ? 64 - __clzll((long long)zeros_left) : 0;   // lane after the nearest 0 to the left
    start = max(start, L - x);                                            // runs do not cross the row start
    out = p - (L - start);
  }
  lab[idx] = out;
}
__global__ void ccl_merge_kernel(const unsigned char* __restrict__ m, int* __restrict__ lab, int target, int H, int W, long long total) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  if (m[idx] != target) return;
  const int HW = H * W;
  int p = idx % HW;
  int* l = lab + (idx - p);
  const unsigned char* mm = m + (idx - p);
  int x = p % W, y = p / W;
  const bool left = x > 0 && mm[p - 1] == target;
  if (left && (threadIdx.x & 63) == 0) uf_union(l, p, p - 1);          // run continues from the previous chunk
  if (y > 0 && mm[p - W] == target && !(left && mm[p - W - 1] == target)) uf_union(l, p, p - W);
}
__global__ void ccl_flatten_kernel(int* __restrict__ lab, int HW, long long total) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  if (lab[idx] < 0) return;
  int p = idx % HW;
  int* l = lab + (idx - p);
  l[p] = uf_find(l, p);
}
// background components touching the image border are "outside"; flag their roots
__global__ void cc_border_kernel(const int* __restrict__ lab, unsigned char* __restrict__ touch, int H, int W, int B) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  int per = 2 * (H + W);
  if (idx >= B * per) return;
  int b = idx / per, r = idx - b * per;
  int x, y;
  if (r < W) { y = 0; x = r; }
  else if (r < 2 * W) { y = H - 1; x = r - W; }
  else if (r < 2 * W + H) { x = 0; y = r - 2 * W; }
  else { x = W - 1; y = r - 2 * W - H; }
  long long base = (long long)b * H * W;
  int l = lab[base + y * W + x];
  if (l >= 0) touch[base + l] = 1;
}
// binary_fill_holes: background pixels whose component does not touch the border become foreground
__global__ void cc_fill_kernel(const unsigned char* __restrict__ m, const int* __restrict__ lab, const unsigned char* __restrict__ touch,
                               unsigned char* __restrict__ out, int HW, long long total) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  int v = m[idx];
  if (!v) {
    long long base = idx - (idx % HW);
    v = touch[base + lab[idx]] ? 0 : 1;
  }
  out[idx] = (unsigned char)v;
}
// per-root statistics, addressed by the root's pixel index: area, min x, min y, max x, max y.
// A wave covers 64 consecutive pixels; when they lie in one row (W % 64 == 0) the lanes of each distinct root are reduced
// with ballots (count = popcount, min/max x = first/last lane of the group) and one lane issues the 5 atomics, so a row
// segment crossing k components costs 5k atomics instead of 5 per pixel.
__global__ void cc_stats_kernel(const int* __restrict__ lab, int* __restrict__ stats, int H, int W, long long total) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int l = idx < total ? lab[idx] : -1;
  const int HW = H * W;
  const int p = idx < total ? (int)(idx % HW) : 0;
  const long long base = idx - p;
  const int x = p % W, y = p / W;
  const int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(l >= 0);
  if (todo == 0) return;
  if ((W & 63) == 0) {
    const long long base0 = __shfl(base, __ffsll((long long)todo) - 1);   // all lanes of a wave share image and row here
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int l0 = __shfl(l, leader);
      const unsigned long long grp = __ballot(l == l0) & todo;
      if (lane == leader) {
        const int first = leader, last = 63 - __clzll((long long)grp);
        const int x0 = x, x1 = x + (last - first);
        int* s = stats + (base0 + l0) * 5;
        atomicAdd(&s[0], __popcll(grp));
        atomicMin(&s[1], x0); atomicMin(&s[2], y); atomicMax(&s[3], x1); atomicMax(&s[4], y);
      }
      todo &= ~grp;
    }
    return;
  }
  if (l < 0) return;
  int* s = stats + (base + l) * 5;
  atomicAdd(&s[0], 1);
  atomicMin(&s[1], x); atomicMin(&s[2], y); atomicMax(&s[3], x); atomicMax(&s[4], y);
}
// The same statistics with the atomics of a 64-column x CCS_ROWS-row strip collected in LDS first: every row segment's groups go into
// a small hash table keyed by the root (LDS atomics), and the strip issues one set of 5 global atomics per component it touches --
// global atomics execute at the memory side at ~50 ns per wave-instruction and CU (MI355X_MICROARCH.md), and a 25-px nucleus crossed
// by 25 row segments paid 125 of them; per strip it pays 5-10.  Min / max / sum commute: same statistics.
#define CCS_ROWS 32
#define CCS_TAB 256
__global__ __launch_bounds__(256) void cc_stats_strip_kernel(const int* __restrict__ lab, int* __restrict__ stats, int H, int W) {
  __shared__ int tkey[CCS_TAB];            // root + 1, 0 = empty
  __shared__ int tval[CCS_TAB][5];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < CCS_TAB; i += 256) { tkey[i] = 0; tval[i][0] = 0; tval[i][1] = 1 << 30; tval[i][2] = 1 << 30; tval[i][3] = -1; tval[i][4] = -1; }
  __syncthreads();
  const int x = blockIdx.x * 64 + lane;
  const long long img = (long long)blockIdx.z * H * W;
  for (int r = wave; r < CCS_ROWS; r += 4) {
    const int y = blockIdx.y * CCS_ROWS + r;
    if (y >= H) break;
    const int l = lab[img + (long long)y * W + x];
    unsigned long long todo = __ballot(l >= 0);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int l0 = __shfl(l, leader);
      const unsigned long long grp = __ballot(l == l0) & todo;
      if (lane == leader) {
        const int x1 = x + (63 - __clzll((long long)grp) - leader);
        int h = (l0 * 0x9E3779B1u) >> 24;             // 8-bit hash of the root
        int tries = 0;
        for (; tries < CCS_TAB; ++tries) {
          const int old = atomicCAS(&tkey[h], 0, l0 + 1);
          if (old == 0 || old == l0 + 1) break;
          h = (h + 1) & (CCS_TAB - 1);
        }
        if (tries < CCS_TAB) {
          atomicAdd(&tval[h][0], __popcll(grp));
          atomicMin(&tval[h][1], x); atomicMin(&tval[h][2], y); atomicMax(&tval[h][3], x1); atomicMax(&tval[h][4], y);
        } else {                                         // table full (more than 256 components in one strip): straight to memory
          int* sg = stats + (img + l0) * 5;
          atomicAdd(&sg[0], __popcll(grp));
          atomicMin(&sg[1], x); atomicMin(&sg[2], y); atomicMax(&sg[3], x1); atomicMax(&sg[4], y);
        }
      }
      todo &= ~grp;
    }
  }
  __syncthreads();
  for (int i = tid; i < CCS_TAB; i += 256) {
    const int k = tkey[i];
    if (k) {
      int* sg = stats + (img + (k - 1)) * 5;
      atomicAdd(&sg[0], tval[i][0]);
      atomicMin(&sg[1], tval[i][1]); atomicMin(&sg[2], tval[i][2]); atomicMax(&sg[3], tval[i][3]); atomicMax(&sg[4], tval[i][4]);
    }
  }
}
__global__ void cc_stats_init_kernel(int* __restrict__ stats, long long total) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  int* s = stats + idx * 5;
  s[0] = 0; s[1] = 1 << 30; s[2] = 1 << 30; s[3] = -1; s[4] = -1;
}
// roots that pass the area filter are appended (unordered) to a per-image list by all pixels in parallel ...
__global__ void cc_collect_kernel(const int* __restrict__ lab, const int* __restrict__ stats, int* __restrict__ list, int* __restrict__ nlist,
                                  int HW, int min_area, int max_area, long long total) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int p = (int)(idx % HW);
  if (lab[idx] != p) return;
  const int a = stats[idx * 5];
  if (!(a > min_area && a < max_area)) return;
  const int b = (int)(idx / HW);
  const int k = atomicAdd(&nlist[b], 1);
  if (k < CC_LIST_CAP) list[(long long)b * CC_LIST_CAP + k] = p;
}
// ... and one block per image sorts the list: ascending root index == raster order of first pixel == scipy.ndimage.label
// numbering; boxes [xmin, ymin, xmax+1, ymax+1]
__global__ __launch_bounds__(1024) void cc_emit_kernel(const int* __restrict__ stats, const int* __restrict__ list, const int* __restrict__ nlist,
                                                       float* __restrict__ boxes, int* __restrict__ counts, int* __restrict__ overflow, int HW, int cap) {
  __shared__ u64 keys[CC_LIST_CAP];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n_all = nlist[b];
  const int n = n_all < CC_LIST_CAP ? n_all : CC_LIST_CAP;
  const int npad = next_pow2(n < 2 ? 2 : n);
  for (int i = tid; i < npad; i += 1024) keys[i] = i < n ? (u64)(unsigned)list[(long long)b * CC_LIST_CAP + i] : ~0ull;
  __syncthreads();
  bitonic_sort_lds(keys, npad);
  const int* st = stats + (long long)b * HW * 5;
  for (int i = tid; i < n && i < cap; i += 1024) {
    const int* s = st + (long long)(int)keys[i] * 5;
    float* o = boxes + ((long long)b * cap + i) * 4;
    o[0] = (float)s[1]; o[1] = (float)s[2]; o[2] = (float)(s[3] + 1); o[3] = (float)(s[4] + 1);
  }
  if (tid == 0) {
    counts[b] = n < cap ? n : cap;
    if (n_all > cap) atomicAdd(&overflow[0], 1);
  }
}

int launch_cc_proposals(const CcParams& p, int B, hipStream_t s) {
  { static const int& skip_ = dev_knob_ref("SKIP", 0); if (skip_ & 128) return 0; }   // dev: ablation of the step (tools/dev/r04_ablate.py)
  ProfScope ps("cc_proposals", 0, 0, s);
  const int H = p.img_h, W = p.img_w, HW = H * W;
  const long long total = (long long)B * HW;
  const unsigned nb = (unsigned)((total + 255) / 256);
  Gauss5 gk;
  {   // torchvision gaussian_blur(kernel_size=5): sigma = 0.15*5+0.35 = 1.1, float32 like torch
    float pdf[5], sum = 0.f;
    for (int i = 0; i < 5; ++i) { float x = (float)(i - 2) / 1.1f; pdf[i] = expf(-0.5f * (x * x)); sum += pdf[i]; }
    float k1[5];
    for (int i = 0; i < 5; ++i) k1[i] = pdf[i] / sum;
    for (int i = 0; i < 5; ++i)
      for (int j = 0; j < 5; ++j) gk.k[i * 5 + j] = k1[i] * k1[j];
  }
  unsigned char *A = p.mask_a, *Bm = p.mask_b;
  hipLaunchKernelGGL(cc_mask_kernel, dim3(cdiv(W, CCM_TW), cdiv(H, CCM_TH), B), dim3(256), 0, s, p.sem_pred, A, p.h, p.w, H, W, gk);
  // open(5x5, 2) == erode 9x9 then dilate 9x9 (zero outside), separable
  hipLaunchKernelGGL((morph9_kernel<0, 0>), dim3(nb), dim3(256), 0, s, A, Bm, H, W, total);
  hipLaunchKernelGGL((morph9_kernel<0, 1>), dim3(nb), dim3(256), 0, s, Bm, A, H, W, total);
  hipLaunchKernelGGL((morph9_kernel<1, 0>), dim3(nb), dim3(256), 0, s, A, Bm, H, W, total);
  hipLaunchKernelGGL((morph9_kernel<1, 1>), dim3(nb), dim3(256), 0, s, Bm, A, H, W, total);
  // hole fill: label background, keep only border-connected background
  hipLaunchKernelGGL(ccl_init_kernel, dim3(nb), dim3(256), 0, s, A, p.labels, 0, HW, W, total);
  hipLaunchKernelGGL(ccl_merge_kernel, dim3(nb), dim3(256), 0, s, A, p.labels, 0, H, W, total);
  hipLaunchKernelGGL(ccl_flatten_kernel, dim3(nb), dim3(256), 0, s, p.labels, HW, total);
  if (hipMemsetAsync(p.touch, 0, (size_t)total, s) != hipSuccess) return NUHTC_E_HIP;
  hipLaunchKernelGGL(cc_border_kernel, dim3(cdiv(B * 2 * (H + W), 256)), dim3(256), 0, s, p.labels, p.touch, H, W, B);
  hipLaunchKernelGGL(cc_fill_kernel, dim3(nb), dim3(256), 0, s, A, p.labels, p.touch, Bm, HW, total);
  // label the filled foreground, collect per-component stats, emit boxes in raster order of first pixel
  hipLaunchKernelGGL(ccl_init_kernel, dim3(nb), dim3(256), 0, s, Bm, p.labels, 1, HW, W, total);
  hipLaunchKernelGGL(ccl_merge_kernel, dim3(nb), dim3(256), 0, s, Bm, p.labels, 1, H, W, total);
  hipLaunchKernelGGL(ccl_flatten_kernel, dim3(nb), dim3(256), 0, s, p.labels, HW, total);
  hipLaunchKernelGGL(cc_stats_init_kernel, dim3(nb), dim3(256), 0, s, p.stats, total);
  if ((W & 63) == 0) hipLaunchKernelGGL(cc_stats_strip_kernel, dim3(W / 64, cdiv(H, CCS_ROWS), B), dim3(256), 0, s, p.labels, p.stats, H, W);
  else hipLaunchKernelGGL(cc_stats_kernel, dim3(nb), dim3(256), 0, s, p.labels, p.stats, H, W, total);
  if (hipMemsetAsync(p.nlist, 0, sizeof(int) * B, s) != hipSuccess) return NUHTC_E_HIP;
  hipLaunchKernelGGL(cc_collect_kernel, dim3(nb), dim3(256), 0, s, p.labels, p.stats, p.list, p.nlist, HW, p.min_area, HW / 4, total);
  hipLaunchKernelGGL(cc_emit_kernel, dim3(B), dim3(1024), 0, s, p.stats, p.list, p.nlist, p.boxes, p.counts, p.overflow, HW, p.cap);
  return hipGetLastError() == hipSuccess ? 0 : NUHTC_E_HIP;
}
