// 3x3 convolution, 64 -> 64 channels, NHWC fp32, on the bf16 matrix pipe with exactly split operands (the arithmetic of
// gemm_split_kernel: six v_mfma_f32_32x32x16_bf16 per 16-deep step, fp32 accumulation, same k order tap-major / channel-minor).
// Serves the FPN output convs (mmdet fpn.py:173-179), the RPN conv (rpn_head.py:62-68), the four semantic-head convs
// (fused_semantic_head.py:97-111) and the mask-head convs (htc_mask_head.py:22-39).
//
// The implicit-GEMM loader of gemm_split_kernel re-reads every input pixel nine times (once per tap) from L2 and re-splits it
// nine times in the MFMA waves' instruction stream.  Here a workgroup owns an 8 x 16 tile of output pixels: the (8+2) x (16+2)
// input halo is read ONCE, split ONCE into its three bf16 planes and kept in LDS; all nine taps then read their A fragments
// from that image at shifted pixel addresses -- the main loop has no vector-ALU work and no activation traffic at all, only LDS
// fragment reads and MFMAs, with the 64 x 64 weights of one tap streamed through a double-buffered LDS image beside them.
// The product is computed transposed (Outᵀ[32 channels][32 pixels] = W · Aᵀ: weights are the MFMA's A operand, pixels its B operand)
// so that a lane ends up with 4 x 4 consecutive channels of ONE pixel: bias, activation and 16-byte NHWC stores need no transpose.
//
// LDS image of the halo: pixel pitch 400 B (64 channels x 3 planes x 2 B + 16: an odd number of 16-byte units) and row pitch
// 7424 B (a multiple of 256 B), which makes the fragment reads of a wave -- 16 consecutive pixels of two adjacent rows --
// conflict-free for ds_read_b128 (lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31}); weight image: 400 B per output channel.
#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "common.h"
#include "split_math.h"

#ifndef NUHTC_CONV_PROBE_EPI
#define NUHTC_CONV_PROBE_EPI 0   // dev probe of the fused epilogue (wrong results): 1 no stores, 2 no residual loads, 4 no exchange barrier, 8 no second product
#endif
#define CV_TH 8
#define CV_TW 16
#define CV_PIX 400                         // bytes per halo pixel in LDS
#define CV_ROW 7424                        // bytes per halo row: (CV_TW + 2) * CV_PIX = 7200, padded to a multiple of 256
#define CV_A_BYTES ((CV_TH + 2) * CV_ROW)  // 74240
#define CV_WCOL 400                        // bytes per output channel of one tap's weights (8 k-groups x 48 B + 16)
#define CV_W_BYTES (64 * CV_WCOL)          // 25600
#define CV_LDS (CV_A_BYTES + 2 * CV_W_BYTES)
// fused pointwise layer (Conv3Fuse): per-wave exchange slots for the partial sums of the second product, the 64 -> 1 partials,
// and the three small vectors (conv bias, pointwise bias, 64 -> 1 weight) the epilogue reads
#define CV_X_OFF CV_LDS
#define CV_X_SLOT 4096
#define CV_XN_OFF (CV_X_OFF + 8 * CV_X_SLOT)
#define CV_K_OFF (CV_XN_OFF + 8 * 256)
#define CV_LDS_FUSED (CV_K_OFF + 1024)     // 161 280 of the 163 840 bytes of a CU

struct Conv3Params {
  const float* in;        // [nimg][H][W][64]
  float* out;             // [nimg][H][W][64]
  const char* wsplit;     // [64 out][72 k-groups][3 planes][8 bf16], k = tap * 64 + channel (gemm_make_split of the packed conv weight)
  const float* bias;      // [64] or null
  const int* nimg_dev;    // optional device-side image count (images >= *nimg_dev are skipped)
  int nimg, H, W, act;
  int tiles_x, tiles_y;
  unsigned long long* stamps;   // dev instrumentation (-DNUHTC_CONV_STAMPS), null otherwise
  // fused pointwise layer (template N2 > 0), see Conv3Fuse
  const char* w2f;        // [2 channel halves][N2 / 32][2 k-steps][3 planes][64 lanes][16 B]
  const float* bias2;
  float* out2;
  const float* res2;
  float* out3;
  const float* wn1;
  const float* bn1;
  float* outn1;
  int act2, store_out;
  // maps of several sizes in one launch (Conv3Fuse.n_more): segment k covers tiles [s_t0[k], s_t0[k + 1]) of the launch's tile list;
  // segment 0 repeats in / out2 / H / W / tiles_x above.  nseg = 1 for every other launch (then s_t0 is not read).
  int nseg;
  const float* s_in[4];
  float* s_out2[4];
  int s_H[4], s_W[4], s_tx[4], s_timg[4], s_t0[5];
};

// N2 = 0: the plain convolution.  N2 = 32 / 64: a pointwise layer 64 -> N2 on the (biased, activated) output tile is computed in
// the epilogue.  A wave holds 32 of the 64 channels of its 32 pixels: with the pointwise weight's k axis permuted on the host
// (k' = 16 u + 8 half + e  <->  channel 16 u + 8 (e / 4) + 4 half + e % 4, the accumulator layout of the 32x32 MFMA) accumulator
// registers 8u .. 8u+7 ARE the B operand of k-step u, so each wave multiplies its own channel half (weight fragments held in
// registers for the workgroup's lifetime) and the two waves of a pixel group add their partial sums through LDS.
template <int N2>
__global__ __launch_bounds__(512, 1) void conv3_split_kernel(Conv3Params p) {
  extern __shared__ __attribute__((aligned(256))) char lds[];
  char* Apl = lds;
  char* Wb = lds + CV_A_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i32 = lane & 31, half = lane >> 5;
  // Persistent workgroups (one per CU): workgroup b serves XCD b & 7 (workgroups are dealt round-robin over the 8 XCDs, which
  // have an L2 each) and walks that XCD's contiguous share of the tile list with stride gridDim / 8: neighbouring tiles, which
  // share halo rows, meet in one L2.  With a device-side image count only the real images are shared out.
  const int nimg = p.nimg_dev ? min(p.nimg, *p.nimg_dev) : p.nimg;
  const int ntile = p.nseg > 1 ? p.s_t0[p.nseg] : nimg * p.tiles_y * p.tiles_x;
  // the map a tile belongs to (wave-uniform): its geometry and pointers; one map unless the launch carries several (s_t0 past nseg = INT_MAX)
  struct Seg { const float* in; float* out2; int H, W, tx, timg, t0; };
  auto seg_of = [&](int t) {
    Seg g;
    if (p.nseg > 1) {
      const int k = (t >= p.s_t0[1]) + (t >= p.s_t0[2]) + (t >= p.s_t0[3]);
      g.in = p.s_in[k]; g.out2 = p.s_out2[k]; g.H = p.s_H[k]; g.W = p.s_W[k]; g.tx = p.s_tx[k]; g.timg = p.s_timg[k]; g.t0 = p.s_t0[k];
    } else {
      g.in = p.in; g.out2 = p.out2; g.H = p.H; g.W = p.W; g.tx = p.tiles_x; g.timg = p.tiles_y * p.tiles_x; g.t0 = 0;
    }
    return g;
  };
  const int per_xcd = (ntile + 7) >> 3;
  const int xcd = blockIdx.x & 7, xstride = gridDim.x >> 3;
  const int t_end = min((xcd + 1) * per_xcd, ntile);
  int tile = xcd * per_xcd + (blockIdx.x >> 3);
  if (tile >= t_end) return;

  // ---- weight staging: 3 x 16-byte pieces per thread and tap (column q / 24, piece q % 24 of the tap's 384 bytes of that column)
  const char* wsrc[3];
  int wdst[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int q = tid + 512 * j, n = q / 24, c = q - n * 24;
    wsrc[j] = p.wsplit + (long long)n * (72 * 48) + c * 16;
    wdst[j] = n * CV_WCOL + c * 16;
  }
  u32x4 wst[3];
#define CV_RAW_BARRIER() { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0) */ __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
#define CV_W_LOAD(tap_) { _Pragma("unroll") for (int j = 0; j < 3; ++j) wst[j] = *reinterpret_cast<const u32x4*>(wsrc[j] + (tap_) * 384); }
#define CV_W_STORE(tap_) { _Pragma("unroll") for (int j = 0; j < 3; ++j) *reinterpret_cast<u32x4*>(Wb + ((tap_) & 1) * CV_W_BYTES + wdst[j]) = wst[j]; }

  // ---- halo staging: item = (pixel, 16-byte piece of its 256 bytes); 180 * 16 = 2880 items over 512 threads.  The loads of the
  // NEXT tile are issued before the main loop of the current one and stay in registers across it.
  constexpr int NPX = (CV_TH + 2) * (CV_TW + 2), NIT = NPX * 16, NJ = (NIT + 511) / 512;
  v4f hv[NJ];
  int hdst[NJ];          // LDS offset of the item's 8 bytes of plane 0 (-1: no item)
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int it = tid + 512 * j;
    const int px = it >> 4, c4 = it & 15;
    const int hy = px / (CV_TW + 2), hx = px - hy * (CV_TW + 2);
    hdst[j] = it < NIT ? hy * CV_ROW + hx * CV_PIX + (c4 >> 1) * 48 + (c4 & 1) * 8 : -1;
  }
#define CV_HALO_LOAD(tile_)                                                                                       \
  { const Seg g_ = seg_of(tile_);                                                                                 \
    const int tl_ = (tile_) - g_.t0, img_ = tl_ / g_.timg, tr_ = tl_ - img_ * g_.timg;                            \
    const int ty_ = tr_ / g_.tx, tx_ = tr_ - ty_ * g_.tx;                                                         \
    const float* map_ = g_.in + (long long)img_ * g_.H * g_.W * 64;                                               \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                              \
      const int it = tid + 512 * j;                                                                               \
      const int px = it >> 4, c4 = it & 15;                                                                       \
      const int hy = px / (CV_TW + 2), hx = px - hy * (CV_TW + 2);                                                \
      const int y = ty_ * CV_TH - 1 + hy, x = tx_ * CV_TW - 1 + hx;                                               \
      const bool ok = it < NIT && y >= 0 && y < g_.H && x >= 0 && x < g_.W;                                       \
      hv[j] = ok ? *reinterpret_cast<const v4f*>(map_ + ((long long)y * g_.W + x) * 64 + c4 * 4) : v4f{0.f, 0.f, 0.f, 0.f}; \
    } }

  CV_HALO_LOAD(tile)
  CV_W_LOAD(0)
  // bias of this lane's 16 output channels: loaded once per workgroup (a load in the epilogue would queue behind the next tile's
  // halo loads and expose their HBM latency: vector-memory operations complete in order)
  v4f bias4[N2 ? 1 : 4];
  constexpr int OB = N2 / 32;                      // 32-row blocks of the pointwise layer's output
  u32x4 w2f[OB ? OB : 1][2][3];
  const int chalf = wave >> 2;
  if constexpr (N2 == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      bias4[q] = p.bias ? *reinterpret_cast<const v4f*>(p.bias + 32 * (wave >> 2) + 4 * half + 8 * q) : v4f{0.f, 0.f, 0.f, 0.f};
  } else {
    float* kst = reinterpret_cast<float*>(lds + CV_K_OFF);      // [0,64) conv bias, [64,128) pointwise bias, [128,192) 64 -> 1 weight, [192] its bias
    if (tid < 64) kst[tid] = p.bias ? p.bias[tid] : 0.f;
    else if (tid < 64 + N2) kst[tid] = p.bias2 ? p.bias2[tid - 64] : 0.f;
    else if (tid >= 128 && tid < 192) kst[tid] = p.wn1 ? p.wn1[tid - 128] : 0.f;
    else if (tid == 192) kst[tid] = p.bn1 ? *p.bn1 : 0.f;
#pragma unroll
    for (int ob = 0; ob < OB; ++ob)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          w2f[ob][u][pl] = *reinterpret_cast<const u32x4*>(p.w2f + ((((long long)(chalf * OB + ob) * 2 + u) * 3 + pl) * 64 + lane) * 16);
  }

  // ---- this wave: pixels (row 2 (wave & 3) + (i32 >> 4), column i32 & 15) of the tile x output channels 32 (wave >> 2) ..
  const int prow = 2 * (wave & 3) + (i32 >> 4), pcol = i32 & 15;
  const char* a_lane = Apl + prow * CV_ROW + pcol * CV_PIX + half * 48;            // + (ky * CV_ROW + kx * CV_PIX) + s * 96 + plane * 16
  const char* w_lane = Wb + (32 * (wave >> 2) + i32) * CV_WCOL + half * 48;        // + buf * CV_W_BYTES + s * 96 + plane * 16

#ifdef NUHTC_CONV_STAMPS
  unsigned long long sSplit = 0, sMain = 0, sEpi = 0, sBar = 0, stt = __builtin_amdgcn_s_memtime(), sk0 = stt, nTiles = 0;
#define CSTAMP(v_) { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); v_ += n_ - stt; stt = n_; __builtin_amdgcn_sched_barrier(0); }
#else
#define CSTAMP(v_)
#endif
  // Deferred stores (plain convolution): the persistent workgroups of a launch run in step, so storing a tile in its epilogue made every
  // CU write at once -- 8 MB bursts at the HBM write rate, 4-6 k of a tile's 27 k cycles with the matrix pipe idle (stamps: r04).  A tile's
  // four 16-byte stores per lane are instead issued one per tap under the NEXT tile's main loop (each right after that tap's weight
  // load, so the first vector-memory operation queued behind a store is the load of the tap after next: two taps for the store to be
  // acknowledged before anything waits on it); the last tile's are flushed after the loop.
  v4f pend[N2 ? 1 : 4];
  float* pend_o = nullptr;
  for (; tile < t_end; tile += xstride) {
    const Seg sg = seg_of(tile);
    const int tl = tile - sg.t0, img = tl / sg.timg, tr = tl - img * sg.timg;
    const int ty = tr / sg.tx, tx = tr - ty * sg.tx;
    const int y0 = ty * CV_TH, x0 = tx * CV_TW;
    // ---- split the halo once, three planes into LDS; weights of tap 0
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (hdst[j] >= 0) {
        char* dst = Apl + hdst[j];
        const Split3 a = split3_pair(hv[j].x, hv[j].y), b = split3_pair(hv[j].z, hv[j].w);
        *reinterpret_cast<uint2*>(dst) = make_uint2(a.p1, b.p1);
        *reinterpret_cast<uint2*>(dst + 16) = make_uint2(a.p2, b.p2);
        *reinterpret_cast<uint2*>(dst + 32) = make_uint2(a.p3, b.p3);
      }
    }
    CV_W_STORE(0)
    CSTAMP(sSplit)
    CV_RAW_BARRIER()       // LDS writes only: nothing waits for the global stores of the previous tile's epilogue
    CSTAMP(sBar)
    CV_W_LOAD(1)

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // ---- main loop over 9 taps x 4 k-steps, software-pipelined TWO steps deep: the six fragments of step i+2 are read between
    // the first MFMAs of step i (three register sets), so the one lgkmcnt wait per step finds reads that were issued a whole
    // step (192 MFMA cycles) earlier.  Per tap: the weights of tap+1 go into the other weight image at its first step (registers
    // loaded a tap earlier); the barrier sits before its third step, whose reads are the first ones of tap+1.
    u32x4 fw[3][3], fa[3][3];
#define CV_READ_ONE(k_, buf_, tap_, s_)                                                                            \
    { const int ky_ = (tap_) / 3, kx_ = (tap_) - ky_ * 3;                                                          \
      if ((k_) < 3) fw[buf_][k_] = *reinterpret_cast<const u32x4*>(w_lane + ((tap_) & 1) * CV_W_BYTES + (s_) * 96 + (k_) * 16);    \
      else fa[buf_][(k_) - 3] = *reinterpret_cast<const u32x4*>(a_lane + ky_ * CV_ROW + kx_ * CV_PIX + (s_) * 96 + ((k_) - 3) * 16); }
#define CV_MFMA_ONE(k_, buf_)                                                                                      \
    { constexpr int iw_[6] = {0, 2, 1, 0, 1, 0}, ia_[6] = {2, 0, 1, 1, 0, 0};   /* (activation, weight) plane order of gemm_split_kernel */ \
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[buf_][iw_[k_]]), __builtin_bit_cast(bf16x8, fa[buf_][ia_[k_]]), acc, 0, 0, 0); }
#pragma unroll
    for (int k = 0; k < 6; ++k) CV_READ_ONE(k, 0, 0, 0)
#pragma unroll
    for (int k = 0; k < 6; ++k) CV_READ_ONE(k, 1, 0, 1)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
      for (int sidx = 0; sidx < 4; ++sidx) {
        const int i = tap * 4 + sidx;
        if (sidx == 0 && tap < 8) {
          CV_W_STORE(tap + 1)
          if (tap < 7) CV_W_LOAD(tap + 2)
        }
        if constexpr (N2 == 0) {
          if (sidx == 0 && tap < 4 && pend_o) *reinterpret_cast<v4f*>(pend_o + 8 * tap) = pend[tap];     // the previous tile's output
        }
        // next tile's halo: requested behind the last weight load of this tile (vector-memory operations complete in issue order: a
        // weight load queued behind these would make its tap wait for them), three taps before the epilogue
        if (sidx == 1 && tap == 6 && tile + xstride < t_end) CV_HALO_LOAD(tile + xstride)
        if (sidx == 2 && tap < 8) CV_RAW_BARRIER()   // only this wave's LDS operations are waited for: the next tile's halo loads stay in flight
        const int ntap = sidx >= 2 ? tap + 1 : tap, ns = (sidx + 2) & 3;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          CV_MFMA_ONE(k, i % 3)
          __builtin_amdgcn_sched_barrier(0);
#ifdef NUHTC_CONV_PROBE_READS   // dev probe (wrong results): the weight fragments of every other step are not read -- a quarter of the main loop's LDS reads
          if (i < 34 && k < 3) {
            if (!((ns & 1) && 2 * k < 3)) CV_READ_ONE(2 * k, (i + 2) % 3, ntap, ns)
            if (!((ns & 1) && 2 * k + 1 < 3)) CV_READ_ONE(2 * k + 1, (i + 2) % 3, ntap, ns)
          }
#else
          if (i < 34 && k < 3) { CV_READ_ONE(2 * k, (i + 2) % 3, ntap, ns) CV_READ_ONE(2 * k + 1, (i + 2) % 3, ntap, ns) }
#endif
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
#undef CV_READ_ONE
#undef CV_MFMA_ONE

    CSTAMP(sMain)
    // ---- epilogue: register r of lane (pixel, half) is output channel 32 (wave >> 2) + (r & 3) + 8 (r >> 2) + 4 half
    const int y = y0 + prow, x = x0 + pcol;
    const bool inside = y < sg.H && x < sg.W;
    const long long pix = ((long long)img * sg.H + y) * sg.W + x;
    if constexpr (N2 == 0) {
      pend_o = inside ? p.out + pix * 64 + 32 * (wave >> 2) + 4 * half : nullptr;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v4f v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
        v += bias4[q];
        if (p.act == ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        pend[q] = v;
      }
    } else {
      const float* kst = reinterpret_cast<const float*>(lds + CV_K_OFF);
      const int cb = 32 * chalf + 4 * half;
      u32x4 vp[2][3];                    // the output tile as the B operand of the second product: k-step u = registers 8u .. 8u+7
      float n1 = 0.f;
      // the residual of the second output is requested before this tile's first store: a load's data is waited for together with
      // every vector-memory operation issued before it, and stores under load take microseconds to be acknowledged
      v4f res[N2 == 64 ? 4 : 1];
      if constexpr (N2 == 64) {
        if (p.out3 && inside && !(NUHTC_CONV_PROBE_EPI & 2)) {
#pragma unroll
          for (int q = 0; q < 4; ++q) res[q] = *reinterpret_cast<const v4f*>(p.res2 + pix * 64 + 32 * chalf + 4 * half + 8 * q);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v4f v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
        v += *reinterpret_cast<const v4f*>(kst + cb + 8 * q);
        if (p.act == ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (p.store_out && inside && !(NUHTC_CONV_PROBE_EPI & 1)) *reinterpret_cast<v4f*>(p.out + pix * 64 + cb + 8 * q) = v;
        if (p.outn1) {
          const v4f wv = *reinterpret_cast<const v4f*>(kst + 128 + cb + 8 * q);
          n1 = fmaf(v.x, wv.x, n1); n1 = fmaf(v.y, wv.y, n1); n1 = fmaf(v.z, wv.z, n1); n1 = fmaf(v.w, wv.w, n1);
        }
        const int u = q >> 1, d0 = 2 * (q & 1);
        NUHTC_SPLIT3_INTO(vp[u], d0, v.x, v.y)
        NUHTC_SPLIT3_INTO(vp[u], d0 + 1, v.z, v.w)
      }
      f32x16 acc2[OB];
#pragma unroll
      for (int ob = 0; ob < OB; ++ob) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[ob][r] = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u) if (!(NUHTC_CONV_PROBE_EPI & 8)) acc2[ob] = mfma_split6_wa(w2f[ob][u], vp[u], acc2[ob]);
      }
      // the other channel half's partial sums: N2 = 64: this wave keeps output block `chalf` and hands over the other one;
      // N2 = 32: it keeps registers 8 chalf .. 8 chalf + 7 of the one block and hands over the other eight
      char* xs = lds + CV_X_OFF + wave * CV_X_SLOT + lane * 16;
      const char* xr = lds + CV_X_OFF + (wave ^ 4) * CV_X_SLOT + lane * 16;
      if constexpr (N2 == 64) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const v4f a0 = {acc2[0][4 * q], acc2[0][4 * q + 1], acc2[0][4 * q + 2], acc2[0][4 * q + 3]};
          const v4f a1 = {acc2[1][4 * q], acc2[1][4 * q + 1], acc2[1][4 * q + 2], acc2[1][4 * q + 3]};
          *reinterpret_cast<v4f*>(xs + q * 1024) = chalf ? a0 : a1;
        }
      } else {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const v4f a0 = {acc2[0][4 * q], acc2[0][4 * q + 1], acc2[0][4 * q + 2], acc2[0][4 * q + 3]};
          const v4f a1 = {acc2[0][8 + 4 * q], acc2[0][8 + 4 * q + 1], acc2[0][8 + 4 * q + 2], acc2[0][8 + 4 * q + 3]};
          *reinterpret_cast<v4f*>(xs + q * 1024) = chalf ? a0 : a1;
        }
      }
      float* xn = reinterpret_cast<float*>(lds + CV_XN_OFF);
      if (p.outn1) {
        n1 += __shfl_xor(n1, 32);
        xn[wave * 64 + lane] = n1;
      }
      if (!(NUHTC_CONV_PROBE_EPI & 4)) CV_RAW_BARRIER()
      if (p.outn1 && chalf == 0 && half == 0 && inside) p.outn1[pix] = (n1 + xn[(wave ^ 4) * 64 + lane]) + kst[192];
      if constexpr (N2 == 64) {
        const int c2 = 32 * chalf + 4 * half;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const v4f mine = chalf ? v4f{acc2[1][4 * q], acc2[1][4 * q + 1], acc2[1][4 * q + 2], acc2[1][4 * q + 3]}
                                 : v4f{acc2[0][4 * q], acc2[0][4 * q + 1], acc2[0][4 * q + 2], acc2[0][4 * q + 3]};
          const v4f other = *reinterpret_cast<const v4f*>(xr + q * 1024);
          // channel half 0's partial first, whichever wave does the addition
          v4f v = (chalf ? other + mine : mine + other) + *reinterpret_cast<const v4f*>(kst + 64 + c2 + 8 * q);
          if (p.act2 == ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          if (inside && !(NUHTC_CONV_PROBE_EPI & 1)) {
            *reinterpret_cast<v4f*>(sg.out2 + pix * 64 + c2 + 8 * q) = v;
            if (p.out3) *reinterpret_cast<v4f*>(p.out3 + pix * 64 + c2 + 8 * q) = res[q] + v;
          }
        }
      } else {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const v4f mine = chalf ? v4f{acc2[0][8 + 4 * q], acc2[0][8 + 4 * q + 1], acc2[0][8 + 4 * q + 2], acc2[0][8 + 4 * q + 3]}
                                 : v4f{acc2[0][4 * q], acc2[0][4 * q + 1], acc2[0][4 * q + 2], acc2[0][4 * q + 3]};
          const v4f other = *reinterpret_cast<const v4f*>(xr + q * 1024);
          const int c2 = 16 * chalf + 8 * q + 4 * half;          // registers 8 chalf + 4 q .. + 3  ->  channels 8 (2 chalf + q) + 4 half ..
          v4f v = (chalf ? other + mine : mine + other) + *reinterpret_cast<const v4f*>(kst + 64 + c2);
          if (p.act2 == ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          if (inside && !(NUHTC_CONV_PROBE_EPI & 1)) *reinterpret_cast<v4f*>(sg.out2 + pix * 32 + c2) = v;
        }
      }
    }
    CV_W_LOAD(0)
    CSTAMP(sEpi)
    CV_RAW_BARRIER()       // every wave is done reading this tile's halo and weight images before the next tile overwrites them
#ifdef NUHTC_CONV_STAMPS
    CSTAMP(sBar)
    ++nTiles;
#endif
  }
  if constexpr (N2 == 0) {
    if (pend_o) {
#pragma unroll
      for (int q = 0; q < 4; ++q) *reinterpret_cast<v4f*>(pend_o + 8 * q) = pend[q];
    }
  }
#ifdef NUHTC_CONV_STAMPS
  if (lane == 0 && p.stamps) {
    unsigned long long* o = p.stamps + ((long long)blockIdx.x * 8 + wave) * 8;
    o[0] = sSplit; o[1] = sMain; o[2] = sEpi; o[3] = sBar; o[4] = nTiles; o[5] = __builtin_amdgcn_s_memtime() - sk0; o[6] = sk0; o[7] = 0;
  }
#endif
#undef CSTAMP
#undef CV_RAW_BARRIER
#undef CV_W_LOAD
#undef CV_W_STORE
#undef CV_HALO_LOAD
}

bool conv3_split_supported(const GemmParams& p) {      // (with p.fuse the pointwise layer rides in the epilogue; checked at launch)
  return p.amode == A_CONV3 && p.Wsplit && p.cC == 64 && p.N == 64 && p.K == 576 && p.lda == 576 && p.ldc == 64 && !p.res && !p.up && p.store == ST_PLAIN &&
         (p.act == ACT_NONE || p.act == ACT_RELU) && p.alpha == 1.f && p.batch <= 1 && p.M % (p.cH * p.cW) == 0 && (!p.m_dev || p.m_mul == p.cH * p.cW);
}

// [N2][64] fp32 -> the fragment image the fused epilogue holds in registers (see conv3_split_kernel<N2>)
static inline unsigned short cv_bf16_rn(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static inline float cv_bf16_f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
int conv3_pack_fuse(const float* w2, int N2, void** out_dev) {
  if (N2 != 32 && N2 != 64) return NUHTC_E_INVALID;
  const int OB = N2 / 32;
  std::vector<unsigned short> img((size_t)2 * OB * 2 * 3 * 64 * 8);
  for (int ch = 0; ch < 2; ++ch)
    for (int ob = 0; ob < OB; ++ob)
      for (int u = 0; u < 2; ++u)
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 8; ++e) {
            const int i32 = lane & 31, h = lane >> 5;
            const int k = 32 * ch + 16 * u + 8 * (e >> 2) + 4 * h + (e & 3);
            const float w = w2[(size_t)(32 * ob + i32) * 64 + k];
            const unsigned short b1 = cv_bf16_rn(w);
            const float r1 = w - cv_bf16_f(b1);
            const unsigned short b2 = cv_bf16_rn(r1);
            const unsigned short b3 = cv_bf16_rn(r1 - cv_bf16_f(b2));
            const size_t base = ((((size_t)(ch * OB + ob) * 2 + u) * 3) * 64 + lane) * 8 + e;
            img[base] = b1; img[base + 64 * 8] = b2; img[base + 2 * 64 * 8] = b3;
          }
  void* d = nullptr;
  if (hipMalloc(&d, img.size() * 2) != hipSuccess) return NUHTC_E_HIP;
  if (hipMemcpy(d, img.data(), img.size() * 2, hipMemcpyHostToDevice) != hipSuccess) { hipFree(d); return NUHTC_E_HIP; }
  *out_dev = d;
  return 0;
}

template <int N2>
static int conv3_raise_lds() {       // more than the default 64 KB of dynamic LDS: raised once per device and instantiation
  static std::map<int, bool> done;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return NUHTC_E_HIP;
  if (!done[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_split_kernel<N2>), hipFuncAttributeMaxDynamicSharedMemorySize, N2 ? CV_LDS_FUSED : CV_LDS) != hipSuccess)
      return NUHTC_E_HIP;
    done[dev] = true;
  }
  return 0;
}

int launch_conv3_split(const GemmParams& g, hipStream_t s) {
  Conv3Params p;
  memset(&p, 0, sizeof(p));
  p.in = g.A; p.out = g.C; p.wsplit = reinterpret_cast<const char*>(g.Wsplit); p.bias = g.bias; p.nimg_dev = g.m_dev;
  p.nimg = g.M / (g.cH * g.cW); p.H = g.cH; p.W = g.cW; p.act = g.act;
  p.tiles_x = cdiv(g.cW, CV_TW); p.tiles_y = cdiv(g.cH, CV_TH);
  int ntile = p.nimg * p.tiles_x * p.tiles_y;
  if (ntile <= 0) return 0;
  int n2 = 0;
  p.nseg = 1;
  if (const Conv3Fuse* f = g.fuse) {
    if ((f->N2 != 32 && f->N2 != 64) || !f->w2f || !f->out2 || (f->out3 && (!f->res2 || f->N2 != 64)) || (f->outn1 && !f->wn1) ||
        (f->act2 != ACT_NONE && f->act2 != ACT_RELU) || (f->store_out && !g.C))
      return NUHTC_E_INVALID;
    n2 = f->N2;
    p.w2f = reinterpret_cast<const char*>(f->w2f); p.bias2 = f->bias2; p.out2 = f->out2; p.res2 = f->res2; p.out3 = f->out3;
    p.wn1 = f->wn1; p.bn1 = f->bn1; p.outn1 = f->outn1; p.act2 = f->act2; p.store_out = f->store_out;
    if (f->n_more) {      // maps of other sizes through the same layers: one tile list, segment after segment
      if (f->n_more < 0 || f->n_more > 3 || f->store_out || f->out3 || f->outn1 || g.m_dev) return NUHTC_E_INVALID;
      p.nseg = 1 + f->n_more;
      p.s_in[0] = p.in; p.s_out2[0] = p.out2; p.s_H[0] = p.H; p.s_W[0] = p.W; p.s_tx[0] = p.tiles_x; p.s_timg[0] = p.tiles_x * p.tiles_y; p.s_t0[0] = 0;
      for (int k = 1; k <= 4; ++k) p.s_t0[k] = 0x7fffffff;
      p.s_t0[1] = ntile;
      for (int k = 1; k < p.nseg; ++k) {
        if (!f->more_in[k - 1] || !f->more_out2[k - 1] || f->more_H[k - 1] <= 0 || f->more_W[k - 1] <= 0) return NUHTC_E_INVALID;
        p.s_in[k] = f->more_in[k - 1]; p.s_out2[k] = f->more_out2[k - 1]; p.s_H[k] = f->more_H[k - 1]; p.s_W[k] = f->more_W[k - 1];
        p.s_tx[k] = cdiv(p.s_W[k], CV_TW); p.s_timg[k] = p.s_tx[k] * cdiv(p.s_H[k], CV_TH);
        p.s_t0[k + 1] = p.s_t0[k] + p.nimg * p.s_timg[k];
      }
      ntile = p.s_t0[p.nseg];
    }
  }
  {
    const int rc = n2 == 64 ? conv3_raise_lds<64>() : n2 == 32 ? conv3_raise_lds<32>() : conv3_raise_lds<0>();
    if (rc) return rc;
  }
  int ncu = 256;
  {
    static std::map<int, int> cus;
    static std::mutex mu2;
    std::lock_guard<std::mutex> lock(mu2);
    int dev = 0;
    hipGetDevice(&dev);
    auto it = cus.find(dev);
    if (it == cus.end()) {
      hipDeviceProp_t prop;
      it = cus.emplace(dev, hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256).first;
    }
    ncu = it->second;
  }
  const int grid = std::min(((ntile + 7) >> 3) * 8, (ncu + 7) / 8 * 8);        // one workgroup per CU (LDS), a multiple of the 8 XCDs
#ifdef NUHTC_CONV_STAMPS
  static unsigned long long* stamp_buf = nullptr;
  if (!stamp_buf && hipMalloc(&stamp_buf, 8ull * 8 * 8 * 512) != hipSuccess) return NUHTC_E_HIP;
  p.stamps = stamp_buf;
#endif
  if (n2 == 64) hipLaunchKernelGGL(conv3_split_kernel<64>, dim3(grid), dim3(512), CV_LDS_FUSED, s, p);
  else if (n2 == 32) hipLaunchKernelGGL(conv3_split_kernel<32>, dim3(grid), dim3(512), CV_LDS_FUSED, s, p);
  else hipLaunchKernelGGL(conv3_split_kernel<0>, dim3(grid), dim3(512), CV_LDS, s, p);
#ifdef NUHTC_CONV_STAMPS
  {
    static int cnt = 0, dump_at = -1;   // launch NUHTC_STAMP_AT of the process is dumped to /tmp/conv_stamps.txt
    if (dump_at < 0) { const char* e = getenv("NUHTC_STAMP_AT"); dump_at = e ? atoi(e) : 200; }
    if (++cnt == dump_at) {
      hipDeviceSynchronize();
      std::vector<unsigned long long> h((size_t)grid * 8 * 8);
      hipMemcpy(h.data(), stamp_buf, h.size() * 8, hipMemcpyDeviceToHost);
      FILE* f = fopen("/tmp/conv_stamps.txt", "w");
      fprintf(f, "# H %d W %d nimg %d ntile %d grid %d N2 %d store_out %d out3 %d\n", p.H, p.W, p.nimg, ntile, grid, n2, p.store_out, p.out3 ? 1 : 0);
      for (int b = 0; b < grid; ++b) for (int w = 0; w < 8; ++w) { auto* o = &h[((size_t)b * 8 + w) * 8]; fprintf(f, "%d %d %llu %llu %llu %llu %llu %llu %llu\n", b, w, o[0], o[1], o[2], o[3], o[4], o[5], o[6]); }
      fclose(f);
    }
  }
#endif
  return hipGetLastError() == hipSuccess ? 0 : NUHTC_E_HIP;
}

const char* nuhtc_tu_probe_conv() {
#if NUHTC_CONV_PROBE_EPI
  return "NUHTC_CONV_PROBE_EPI";
#elif defined(NUHTC_CONV_PROBE_READS)
  return "NUHTC_CONV_PROBE_READS";
#else
  return nullptr;
#endif
}
