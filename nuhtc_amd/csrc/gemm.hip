// fp32 GEMM on the CDNA4 matrix cores: C = epilogue(A · Wᵀ), v_mfma_f32_32x32x2_f32 (exact f32 fma chain).
// Block = 4 waves arranged WM x WN, each wave owns MT x NT sub-tiles of 32x32 (block tile BM x BN = 32·MT·WM x 32·NT·WN),
// k-tile BK (16 or 32), double-buffered LDS image in row-major [row][BK+4] (128-bit conflict-free staging writes and
// fragment reads), register prefetch of the next k-tile issued ahead of the MFMAs.  Serves every dense layer of the
// path: Swin QKV/proj/FFN/merge linears (mmdet swin.py:88,115, mmcv FFN, transformer.py:384), FPN / RPN / semantic /
// mask-head convolutions as NHWC implicit GEMM (fpn.py:152-179, rpn_head.py:62-68, fused_semantic_head.py:97-111,
// htc_mask_head.py:22-39), the bbox-head FCs (convfc_bbox_head.py:158-196) and the attention-pool similarity /
// aggregation products (nuhtc/models/roi_extractors_cus.py:228-235).
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>
#include <cstdio>

#include "common.h"

#include "split_math.h"   // vector typedefs, gelu_erf, the bf16 split helpers

// ---- epilogue shared by the fp32 and the split-bf16 main loops
template <int MT, int NT, int WM, int WN>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x16 (&acc)[MT][NT], float* lds, float* __restrict__ C, int z, int m0, int n0,
                                              int Meff, const float* ln_rstd = nullptr /* A_LN: LDS, 1 / sqrt(var + eps) per row of the block tile */) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int i32 = lane & 31, half = lane >> 5;
  // ---- epilogue.  The accumulators hold a 32x32 sub-tile with the column on the lane and 16 rows in the registers, which
  // would mean 16 single-dword stores (and residual loads) per lane per sub-tile: at K = 96..384 that store / load issue
  // costs as much as the whole k-loop.  Each wave therefore transposes its sub-tile through a private 4 KB slice of the
  // (now idle) staging LDS -- 16 conflict-free ds_write_b32, 4 ds_read_b128 -- so that a lane owns 4 consecutive columns
  // of 4 rows: every global access of the epilogue (bias, residual, FPN top-down term, the store) is one 16-byte
  // instruction covering 8 full 128-byte row segments per wave.
  const float* ri = p.cos_ri ? p.cos_ri + (long long)z * p.sRi : nullptr;
  const float* rj = p.cos_rj ? p.cos_rj + (long long)z * p.sRj : nullptr;
  __syncthreads();                              // every wave is done reading the k-loop's LDS tiles
  float* tb = lds + wave * (32 * 32);           // this wave's transpose tile [row][col]
  const int rr = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    // bookkeeping of the 4 rows this lane stores (rows 8j + rr of the sub-tile): validity, scatter map, FPN parent /
    // deconv base row; all map reads are issued together from clamped addresses
    int mrow[4], drow[4], aux[4];
    unsigned okmask = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + (wm * MT + mi) * 32 + 8 * j + rr;
      if (m < Meff) okmask |= 1u << j;
      mrow[j] = m < Meff ? m : Meff - 1;
      drow[j] = mrow[j];
      aux[j] = 0;
    }
    if (p.store == ST_ROWMAP) {
      int d[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) d[j] = p.row_map[mrow[j]];
#pragma unroll
      for (int j = 0; j < 4; ++j) { if (d[j] < 0) okmask &= ~(1u << j); drow[j] = d[j] >= 0 ? d[j] : 0; }
    }
    if (p.up) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int hw = p.upH * p.upW;
        const int b = mrow[j] / hw, q = mrow[j] - b * hw;
        const int y = q / p.upW, x = q - y * p.upW;
        aux[j] = (b * (p.upH >> 1) + (y >> 1)) * (p.upW >> 1) + (x >> 1);
      }
    } else if (p.store == ST_DECONV2) {
      // rows m = (d, y, x) on a cH x cW grid; columns n = (kh*2+kw)*ldc + oc -> out[(d, 2y+kh, 2x+kw), oc]
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int hw = p.cH * p.cW;
        const int b = mrow[j] / hw, q = mrow[j] - b * hw;
        const int y = q / p.cW, x = q - y * p.cW;
        aux[j] = (b * (2 * p.cH) + 2 * y) * (2 * p.cW) + 2 * x;
      }
    }
    float riv[4];
    if (p.act == ACT_COS) {
#pragma unroll
      for (int j = 0; j < 4; ++j) riv[j] = ri[mrow[j]];
    }
    float lnr[4] = {1.f, 1.f, 1.f, 1.f};      // A_LN: the row's 1 / sqrt(var + eps) scales its sums before the bias
    if (ln_rstd) {
#pragma unroll
      for (int j = 0; j < 4; ++j) lnr[j] = ln_rstd[(wm * MT + mi) * 32 + 8 * j + rr];
    }
    // stats_out: this workgroup's 96 columns of each row it stores, as {mean, sum of squared deviations} -- sums of (v - pivot) and
    // (v - pivot)^2 over the row's 96 values with the row's first value as pivot (a sample of the row: nothing cancels), gathered
    // while the tiles pass through the registers anyway
    float s_piv[4] = {0.f, 0.f, 0.f, 0.f}, s_1[4] = {0.f, 0.f, 0.f, 0.f}, s_2[4] = {0.f, 0.f, 0.f, 0.f};
    // Vector-memory operations retire in issue order, so a load issued after a column tile's stores would wait for those
    // stores to be acknowledged (thousands of cycles under load) before its data counts as landed.  Bias / column norms of
    // every column tile are therefore loaded before the first store, and the row-dependent terms (residual, FPN parent)
    // of tile t+1 are requested (into the registers tile t's terms just left) before tile t is stored: the wait for them
    // never has a store ahead of it.
    // (launch_gemm rejects bias together with the cosine epilogue and residual together with the FPN term, so one column
    // vector per tile and one row-term array serve all epilogues)
    const float* colp = p.act == ACT_COS ? rj : p.bias;
    const float* rowp = p.res ? p.res : p.up;
    const int rowld = p.res ? p.ldr : p.N;
    unsigned rowoff[4];   // element offsets (launch_gemm checks they fit 32 bits)
#pragma unroll
    for (int j = 0; j < 4; ++j) rowoff[j] = (unsigned)(p.res ? drow[j] : aux[j]) * (unsigned)rowld + (unsigned)(n0 + wn * NT * 32 + c4);
    v4f colv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      colv[t] = v4f{0.f, 0.f, 0.f, 0.f};
      if (colp) colv[t] = *reinterpret_cast<const v4f*>(colp + n0 + (wn * NT + t) * 32 + c4);
    }
    v4f rowv[4];
#define EPI_LOADS(t_)                                                                                                   \
  if (rowp) {                                                                                                           \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                       \
        rowv[j] = *reinterpret_cast<const v4f*>(rowp + rowoff[j] + (t_) * 32);                                          \
  }
    EPI_LOADS(0)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) tb[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + i32] = acc[mi][t][r];
      const int n = n0 + (wn * NT + t) * 32 + c4;            // this lane's 4 columns
      // (DS operations of one wave execute in order: the reads below see the writes above, and the next sub-tile's writes
      // cannot overtake these reads)
      v4f v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = *reinterpret_cast<const v4f*>(tb + (8 * j + rr) * 32 + c4) * p.alpha;
        if (ln_rstd) v[j] *= lnr[j];
        if (p.bias) v[j] += colv[t];
        if (p.act == ACT_RELU) {
          v[j].x = fmaxf(v[j].x, 0.f); v[j].y = fmaxf(v[j].y, 0.f); v[j].z = fmaxf(v[j].z, 0.f); v[j].w = fmaxf(v[j].w, 0.f);
        } else if (p.act == ACT_GELU) {
          v[j].x = gelu_erf(v[j].x); v[j].y = gelu_erf(v[j].y); v[j].z = gelu_erf(v[j].z); v[j].w = gelu_erf(v[j].w);
        } else if (p.act == ACT_COS) {
          v[j].x = fmaxf(v[j].x * riv[j] * colv[t].x - p.cos_tau, 0.f) + p.cos_tau;
          v[j].y = fmaxf(v[j].y * riv[j] * colv[t].y - p.cos_tau, 0.f) + p.cos_tau;
          v[j].z = fmaxf(v[j].z * riv[j] * colv[t].z - p.cos_tau, 0.f) + p.cos_tau;
          v[j].w = fmaxf(v[j].w * riv[j] * colv[t].w - p.cos_tau, 0.f) + p.cos_tau;
        }
        if (rowp) v[j] += rowv[j];
      }
      if (p.stats_out) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (t == 0) s_piv[j] = __shfl(v[j].x, lane & ~7);
          const v4f d = v[j] - s_piv[j];
          s_1[j] += (d.x + d.y) + (d.z + d.w);
          s_2[j] = fmaf(d.x, d.x, fmaf(d.y, d.y, fmaf(d.z, d.z, fmaf(d.w, d.w, s_2[j]))));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < NT) EPI_LOADS(t + 1)          // requested before this tile's stores are issued
      __builtin_amdgcn_sched_barrier(0);
#ifdef NUHTC_GEMM_NOSTORE   // dev probe: the launch without its output stores (results are lost): what hiding the store phase could buy at most
#ifndef NUHTC_GEMM_NOSTORE_K   // (-DNUHTC_GEMM_NOSTORE_K=3136: only the launches of that depth lose their stores)
#define NUHTC_GEMM_NOSTORE_K 0
#endif
      if (p.alpha == 12345.f || (NUHTC_GEMM_NOSTORE_K && p.K != NUHTC_GEMM_NOSTORE_K))
#endif
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (!((okmask >> j) & 1u)) continue;
        if (p.store == ST_DECONV2) {
          const int tap = n / p.ldc, oc = n - tap * p.ldc;     // ldc % 4 == 0: the 4 columns share one tap
          *reinterpret_cast<v4f*>(C + ((long long)aux[j] + (tap >> 1) * (2 * p.cW) + (tap & 1)) * p.ldc + oc) = v[j];
        } else {
          *reinterpret_cast<v4f*>(C + (long long)drow[j] * p.ldc + n) = v[j];
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the live ranges of one column tile from overlapping the next
    }
    if (p.stats_out) {
      constexpr float inv_n = 1.0f / (32.0f * NT * WN);
      const int ntn = p.N / (32 * NT * WN), tile_n = n0 / (32 * NT * WN);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) { s_1[j] += __shfl_xor(s_1[j], o); s_2[j] += __shfl_xor(s_2[j], o); }
        if ((lane & 7) == 0 && ((okmask >> j) & 1u)) {
          const float dm = s_1[j] * inv_n;
          *reinterpret_cast<float2*>(p.stats_out + ((long long)drow[j] * ntn + tile_n) * 2) = make_float2(s_piv[j] + dm, fmaxf(s_2[j] - s_1[j] * dm, 0.f));
        }
      }
    }
#undef EPI_LOADS
  }
}

template <int MT, int NT, int WM, int WN, int BK, int AMODE>
__global__ __launch_bounds__(256, (MT * NT <= 3 ? 4 : 3)) void gemm_kernel(GemmParams p) {   // 4 (3) blocks per CU: <= 128 (168) registers
  static_assert(WM * WN == 4, "4 waves per block");
  constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
  constexpr int KC = BK / 4;          // float4 chunks per tile row
  constexpr int RPP = 256 / KC;       // rows staged per pass of the 256 threads
  constexpr int NA = (BM + RPP - 1) / RPP, NB = (BN + RPP - 1) / RPP;   // staging passes (the last may be partial)
  // LDS image: row-major [row][BK + 4] for both operands (the +4 pad makes the 128-bit fragment reads and the 128-bit
  // staging writes bank-conflict free).  MFMA k-step s of a tile multiplies k = s (lanes 0-31) and k = s + BK/2 (lanes
  // 32-63), so each lane's fragments of a whole tile are BK/2 contiguous floats = BK/8 ds_read_b128.
  constexpr int LDK = BK + 4;
  constexpr int KH = BK / 2;
  __shared__ __attribute__((aligned(16))) float lds[2 * LDK * (BM + BN)];
  float* As = lds;                  // [2][BM][LDK]
  float* Bs = lds + 2 * LDK * BM;   // [2][BN][LDK]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int i32 = lane & 31, half = lane >> 5;
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (ids with equal id % 8 share an L2), so all
  // n-tiles of one m-tile get ids of the same residue, adjacent in time
  const int nTilesN = p.N / BN;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tile_m = (slot / nTilesN) * 8 + xcd, tile_n = slot % nTilesN;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int z = blockIdx.z;

  int Meff = p.M;
  if (p.m_dev) {
    int md = *p.m_dev * p.m_mul;
    Meff = md < Meff ? md : Meff;
  }
  if (m0 >= Meff) return;

#ifdef NUHTC_GEMM_STAMPS   // dev instrumentation (tools/dev/stamps.py): per-wave phase times and in-loop waits
  unsigned long long st0 = __builtin_amdgcn_s_memtime(), st1 = 0, st2 = 0, st3 = 0, st5 = 0, st6 = 0;
#define STAMP(x_) x_
#else
#define STAMP(x_)
#endif
  const float* __restrict__ A = p.A + (long long)z * p.sA;
  const float* __restrict__ Wt = p.W + (long long)z * p.sW;
  float* __restrict__ C = p.C + (long long)z * p.sC;

  // ---- per-thread staging assignment: float4 slot (row = tid/KC + RPP*j, kc = tid%KC)
  // Rows past Meff are clamped to the last valid row (their results are never stored) and out-of-image conv taps are
  // redirected to a page of zeros: every staging load is unconditional, so the loads of tile kt+1 stay in flight across
  // the MFMAs of tile kt instead of being fenced by exec-mask branches.
  const int kc = tid % KC;
  const int rbase = tid / KC;
  const float* a_ptr[NA];
  unsigned a_ok[NA];   // A_CONV3: bit t set = tap t (ky = t/3 - 1, kx = t%3 - 1) of this row's pixel lies inside the image
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    int r = rbase + RPP * j;
    r = r < BM ? r : BM - 1;
    int m = m0 + r;
    m = m < Meff ? m : Meff - 1;
    if (AMODE == A_PLAIN) {
      a_ptr[j] = A + (long long)m * p.lda + kc * 4;
      a_ok[j] = 0;
    } else {
      const int hw = p.cH * p.cW;
      const int rr = m - (m / hw) * hw;
      const int y = rr / p.cW, x = rr - y * p.cW;
      unsigned bits = 0;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        if (yy >= 0 && yy < p.cH && xx >= 0 && xx < p.cW) bits |= 1u << t;
      }
      a_ok[j] = bits;
      a_ptr[j] = A + (long long)m * p.cC + kc * 4;   // NHWC: row m is pixel m
    }
  }
  const float* w_ptr[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    int r = rbase + RPP * j;
    r = r < BN ? r : BN - 1;
    w_ptr[j] = Wt + (long long)(n0 + r) * p.K + kc * 4;
  }
  // a partial last staging pass (e.g. BN = 96 rows with 64 rows per pass) belongs to whole waves (a wave stages RPP/4 = 16
  // consecutive rows per pass): the surplus waves skip it behind a scalar branch, so no lane is ever masked and the
  // per-CU vector-memory path, which bounds the K = 96..192 launches, carries no duplicate rows
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  int a_srow[NA], w_srow[NB];
#pragma unroll
  for (int j = 0; j < NA; ++j) { int r = rbase + RPP * j; a_srow[j] = r < BM ? r : BM - 1; }
#pragma unroll
  for (int j = 0; j < NB; ++j) { int r = rbase + RPP * j; w_srow[j] = r < BN ? r : BN - 1; }

  v4f ra[NA], rb[NB];
  int cv_tap = 0, cv_c0 = 0;   // A_CONV3: tap and channel offset of the next k-tile to load (k-tiles are visited in order)
  long long cv_off = 0;
  // One staging load / LDS store / fragment read per call, so the main loop can place them one by one between MFMAs
  // (macros rather than lambdas: arrays captured by reference were being demoted to scratch memory).
#define CONV_BEGIN()                                                                                            \
  if (AMODE != A_PLAIN) {                                                                                       \
    const int ky = cv_tap / 3 - 1, kx = cv_tap - (cv_tap / 3) * 3 - 1;                                          \
    cv_off = (long long)(ky * p.cW + kx) * p.cC + cv_c0;                                                        \
  }
#define CONV_END()                                                                                              \
  if (AMODE != A_PLAIN) {                                                                                       \
    cv_c0 += BK;                                                                                                \
    if (cv_c0 == p.cC) { cv_c0 = 0; cv_tap = cv_tap < 8 ? cv_tap + 1 : 8; }   /* (the clamped tail reload is unused) */ \
  }
#define LOAD_ONE(f_, kt_)                                                                                       \
  {                                                                                                             \
    if ((f_) < NA) {                                                                                            \
      const int j = (f_) < NA ? (f_) : 0;                                                                       \
      if (AMODE == A_PLAIN) {                                                                                   \
        ra[j] = *reinterpret_cast<const v4f*>(a_ptr[j] + (kt_) * BK);                                           \
      } else {                                                                                                  \
        const bool ok = (a_ok[j] >> cv_tap) & 1u;                                                               \
        const float* src = ok ? a_ptr[j] + cv_off : p.zeros;   /* out-of-image taps read a page of zeros */      \
        ra[j] = *reinterpret_cast<const v4f*>(src);                                                             \
      }                                                                                                         \
    } else if ((f_) < NA + NB) {                                                                                \
      const int j = (f_) >= NA && (f_) < NA + NB ? (f_) - NA : 0;                                               \
      if (wave_u * (RPP / 4) + RPP * j < BN) rb[j] = *reinterpret_cast<const v4f*>(w_ptr[j] + (kt_) * BK);      \
    }                                                                                                           \
  }
#define STORE_ONE(f_, buf_)                                                                                     \
  {                                                                                                             \
    if ((f_) < NA) {                                                                                            \
      const int j = (f_) < NA ? (f_) : 0;                                                                       \
      *reinterpret_cast<v4f*>(As + (buf_) * LDK * BM + a_srow[j] * LDK + kc * 4) = ra[j];                        \
    } else if ((f_) < NA + NB) {                                                                                \
      const int j = (f_) >= NA && (f_) < NA + NB ? (f_) - NA : 0;                                               \
      if (wave_u * (RPP / 4) + RPP * j < BN)                                                                    \
        *reinterpret_cast<v4f*>(Bs + (buf_) * LDK * BN + w_srow[j] * LDK + kc * 4) = rb[j];                      \
    }                                                                                                           \
  }
#define LOAD_TILE(kt_)                                                                                          \
  {                                                                                                             \
    CONV_BEGIN()                                                                                                \
    _Pragma("unroll") for (int f = 0; f < NA + NB; ++f) LOAD_ONE(f, kt_)                                        \
    CONV_END()                                                                                                  \
  }
#define STORE_TILE(buf_) { _Pragma("unroll") for (int f = 0; f < NA + NB; ++f) STORE_ONE(f, buf_) }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][t][r] = 0.f;

  // ---- main loop, software-pipelined at half-tile granularity (BK = 16: two groups of 4 k-steps = 4·MT·NT MFMAs each).
  // Every non-MFMA instruction of the k-tile is a "filler" placed alone in the gap after one MFMA, because a wave-wide
  // LDS write (13 cycles), global load (~12) or LDS read (4) occupies the SIMD's issue port: bunched together at the
  // barrier they open a gap no co-resident wave reliably covers, one per MFMA they fit inside the MFMA's own 64 cycles.
  //   group A: MFMAs on fragments FA (tile t, half 0) | fillers: read FB <- LDS(t, half 1), then write regs(t+1) -> LDS
  //   lgkmcnt(0) + one barrier
  //   group B: MFMAs on FB                            | fillers: global loads(t+2) -> regs, then read FA <- LDS(t+1, half 0)
  // so no MFMA waits for LDS (its operands were requested half a tile earlier) and the staging loads have a whole tile
  // (>= 1500 cycles) to land before the LDS writes need them.
  static_assert(BK == 16, "pipelined main loop is written for BK = 16");
  v4f fa_a[MT], fa_b[NT], fb_a[MT], fb_b[NT];
  const float* arow0 = As + (wm * MT * 32 + i32) * LDK + half * KH;
  const float* brow0 = Bs + (wn * NT * 32 + i32) * LDK + half * KH;
  constexpr int NMF = 4 * MT * NT;                  // MFMAs per group
  constexpr int NRD = MT + NT, NST = NA + NB;       // fragment reads per half tile; staging loads (= LDS writes) per tile
  constexpr int FPG = (NRD + NST + NMF - 1) / NMF;  // fillers per MFMA gap (1 except for the smallest wave tile)
#define READ_ONE(f_, buf_, q_, FA_, FB_)                                                                       \
  {                                                                                                            \
    if ((f_) < MT) {                                                                                           \
      const int mi = (f_) < MT ? (f_) : 0;                                                                     \
      FA_[mi] = *reinterpret_cast<const v4f*>(arow0 + (buf_) * LDK * BM + 32 * mi * LDK + 4 * (q_));           \
    } else if ((f_) < MT + NT) {                                                                               \
      const int t = (f_) >= MT && (f_) < MT + NT ? (f_) - MT : 0;                                              \
      FB_[t] = *reinterpret_cast<const v4f*>(brow0 + (buf_) * LDK * BN + 32 * t * LDK + 4 * (q_));             \
    }                                                                                                          \
  }
#define READ_FRAGS(buf_, q_, FA_, FB_) { _Pragma("unroll") for (int f = 0; f < NRD; ++f) READ_ONE(f, buf_, q_, FA_, FB_) }
#define MFMA_AT(i_, FA_, FB_)                                                                                  \
  {                                                                                                            \
    const int e = (i_) / (MT * NT), mi = ((i_) % (MT * NT)) / NT, t = (i_) % NT;                               \
    acc[mi][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA_[mi][e], FB_[t][e], acc[mi][t], 0, 0, 0);             \
  }
#define MFMA_GROUP(FA_, FB_) { _Pragma("unroll") for (int i = 0; i < NMF; ++i) MFMA_AT(i, FA_, FB_) }

  const int nk = p.K / BK;
  LOAD_TILE(0)
  STORE_TILE(0)
  __syncthreads();
  LOAD_TILE(nk > 1 ? 1 : 0)
  READ_FRAGS(0, 0, fa_a, fa_b)
  STAMP(__builtin_amdgcn_s_waitcnt(0xC07F); st1 = __builtin_amdgcn_s_memtime();)
  int kt = 0;
#define KT_BODY(LD_)                                                                                           \
  {                                                                                                            \
    const int buf = kt & 1;                                                                                    \
    _Pragma("unroll") for (int i = 0; i < NMF; ++i) {                                                          \
      MFMA_AT(i, fa_a, fa_b)                                                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
      _Pragma("unroll") for (int u = 0; u < FPG; ++u) {                                                        \
        const int f = i * FPG + u;                                                                             \
        STAMP(if (f == NRD) { unsigned long long w0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0x0070); st6 += __builtin_amdgcn_s_memtime() - w0; }) \
        if (f < NRD) READ_ONE(f, buf, 1, fb_a, fb_b)                                                           \
        else STORE_ONE(f - NRD, buf ^ 1) /* waits (vmcnt) for that staging load, issued a whole k-tile ago */  \
      }                                                                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
    }                                                                                                          \
    __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0): this wave's LDS writes have landed */                   \
    STAMP(st5 -= __builtin_amdgcn_s_memtime();)                                                                \
    __builtin_amdgcn_s_barrier();       /* raw barrier: no vmcnt(0) (nothing is in flight here anyway) */      \
    STAMP(st5 += __builtin_amdgcn_s_memtime();)                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    if (LD_) CONV_BEGIN()                                                                                      \
    _Pragma("unroll") for (int i = 0; i < NMF; ++i) {                                                          \
      MFMA_AT(i, fb_a, fb_b)                                                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
      _Pragma("unroll") for (int u = 0; u < FPG; ++u) {                                                        \
        const int f = i * FPG + u;                                                                             \
        if (f < NST) { if (LD_) LOAD_ONE(f, kt + 2) }                                                          \
        else READ_ONE(f - NST, buf ^ 1, 0, fa_a, fa_b)                                                         \
      }                                                                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
    }                                                                                                          \
    if (LD_) CONV_END()                                                                                        \
  }
  for (; kt + 2 < nk; ++kt) KT_BODY(1)   // tiles with a successor two ahead: no conditional inside the body
  KT_BODY(0)                             // tile nk-2 (K >= 32): nothing left to load
  ++kt;
#undef KT_BODY
  READ_FRAGS(kt & 1, 1, fb_a, fb_b)
  MFMA_GROUP(fa_a, fa_b)
  MFMA_GROUP(fb_a, fb_b)
#undef CONV_BEGIN
#undef CONV_END
#undef LOAD_ONE
#undef STORE_ONE
#undef LOAD_TILE
#undef STORE_TILE
#undef READ_ONE
#undef READ_FRAGS
#undef MFMA_AT
#undef MFMA_GROUP

  STAMP(st2 = __builtin_amdgcn_s_memtime();)
  gemm_epilogue<MT, NT, WM, WN>(p, acc, lds, C, z, m0, n0, Meff);
#ifdef NUHTC_GEMM_STAMPS
  st3 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0)
  if (lane == 0 && p.stamps) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned long long* o = p.stamps + ((long long)blockIdx.x * 4 + wave) * 8;
    o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3; o[4] = __builtin_amdgcn_s_memtime(); o[5] = hw; o[6] = st5; o[7] = st6;
  }
#endif
#undef STAMP
}

// ======================================================================================================================
// fp32 GEMM on the bf16 matrix pipe by exact operand splitting.
// gfx950's fp32 MFMA runs at 1/16 of the bf16 rate.  An fp32 number is exactly the sum of three bf16 numbers (24 significand
// bits = 8 + 8 + 8; round-to-nearest splits, the residuals are exact fp32 subtractions), a bf16 x bf16 product is exact in
// fp32, and v_mfma_f32_32x32x16_bf16 accumulates in fp32.  So  a*b = sum_ij a_i*b_j  over nine exact products; the three
// smallest are dropped (round-to-nearest splits: |a2| <= 2^-8 |a|, |a3| <= 2^-16 |a|, so a2*b3 and a3*b2 are <= 2^-24 |a*b| each
// and a3*b3 <= 2^-32: at worst 2^-23 |a*b| per product, one fp32 rounding unit): SIX bf16 MFMAs of depth 16 (192
// cycles) replace the eight fp32 MFMAs of depth 2 (512 cycles) of the same 32x32x16 product.  Measured against fp64
// (tools/dev/probe/bf16split_probe.hip, K = 96 .. 3136, normal and wide-range operands): max error 1.2-1.6e-7 of sum|a*b|
// against 1.4-2.1e-7 for the fp32 MFMA chain, identical with all nine products -- the result is fp32 arithmetic (exact
// products, fp32 accumulation), only the summation order differs, and the 16-deep dot products round less often.
//   W is split once at nuhtc_finalize (gemm_make_split) into Wsplit[n][k/8][plane 0..2][8 bf16]: the 96 bytes a column
//   needs per 16-deep k-tile are contiguous in HBM and in the LDS image (112-byte column pitch: conflict-free b128 reads);
//   A stays fp32 in HBM and LDS (same staging as the fp32 kernel, implicit 3x3-conv loader included) and is split in
//   registers after the fragment read: 44 VALU instructions per k-tile and wave beside 6*NT MFMAs.
// Block = 4 waves x 32 rows, NT column tiles of 32, BK = 16, double-buffered LDS, one barrier per k-tile; epilogue shared.
template <int MT, int NT, int AMODE>
__global__ __launch_bounds__(256, (MT * NT <= 4 ? 3 : 2)) void gemm_split_kernel(GemmParams p) {
  constexpr int WM = 4, WN = 1, BK = 16;
  constexpr int BM = 128 * MT, BN = 32 * NT;            // a wave owns MT row tiles of 32 (rows (wave * MT + mi) * 32 ..) x NT column tiles
  constexpr int KC = BK / 4, RPP = 256 / KC, NA = BM / RPP;       // A staging: float4 slots, 64 rows per pass, 2 * MT passes
  constexpr int LDK = BK + 4;                                      // A image [row][20 floats]
  constexpr int BP = 28;                                           // B image [col][112 bytes = 28 floats]: 2 k-groups x 3 planes x 16 B + pad
  constexpr int NCH = BN * 6, NB = (NCH + 255) / 256;              // B staging: 16-byte chunks per k-tile, passes of 256 threads
  constexpr int LDSF = 2 * (LDK * BM + BP * BN) > 4 * 32 * 32 ? 2 * (LDK * BM + BP * BN) : 4 * 32 * 32;
  __shared__ __attribute__((aligned(16))) float lds[LDSF + (AMODE == A_LN ? 2 * BM : 0)];  // A_LN: + the block tile's rows' 1 / sqrt(var + eps) and means
  float* As = lds;                       // [2][BM][LDK]
  float* Bs = lds + 2 * LDK * BM;        // [2][BN][BP]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int i32 = lane & 31, half = lane >> 5;
  const int nTilesN = p.N / BN;
  // A_LN: the FIRST workgroups of the launch (a multiple of 8 of them, so that the tile workgroups keep their XCDs) fill the window-padding
  // rows of C with pad_val (the QKV bias), 16 rows per wave, beside the first round of tiles; at the end of the grid they were the launch's tail
  int bid = blockIdx.x;
  if (AMODE == A_LN && p.n_pad > 0) {
    const int pad_blocks = ((p.n_pad + 63) / 64 + 7) / 8 * 8;
    if (bid < pad_blocks) {
      const int r0 = bid * 64 + wave * 16;
      const v4f* src = reinterpret_cast<const v4f*>(p.pad_val);
      for (int r = r0; r < r0 + 16 && r < p.n_pad; ++r) {
        v4f* dst = reinterpret_cast<v4f*>(p.C + (long long)p.pad_rows[r] * p.ldc);
        for (int c = lane; c < p.N / 4; c += 64) dst[c] = src[c];
      }
      return;
    }
    bid -= pad_blocks;
  }
  const int xcd = bid & 7, slot = bid >> 3;
  // Workgroups go round the XCDs (each with its own L2); an XCD's consecutive slots walk the column tiles of one row tile (its A rows are
  // fetched once into that L2 and every slice of W once per row tile) -- or, row_fastest, the row tiles of one column tile: when W is larger
  // than an L2 and the XCD's share of A is not (the stage-4 QKV / fc1 linears: 10-14 MB of split weights, 1.5 MB of rows per XCD), W then
  // crosses the fabric once per XCD instead of once per row tile.  Same time (round 4: `r04_ab_tile_order.txt`), a third less traffic there.
  const int rowTilesPerXcd = ((p.M + BM - 1) / BM + 7) / 8;
  const int tile_m = p.row_fastest ? (slot % rowTilesPerXcd) * 8 + xcd : (slot / nTilesN) * 8 + xcd;
  const int tile_n = p.row_fastest ? slot / rowTilesPerXcd : slot % nTilesN;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int z = blockIdx.z;
  int Meff = p.M;
  if (p.m_dev) {
    int md = *p.m_dev * p.m_mul;
    Meff = md < Meff ? md : Meff;
  }
  if (m0 >= Meff) return;
#ifdef NUHTC_GEMM_STAMPS
  unsigned long long st0 = __builtin_amdgcn_s_memtime(), st1 = 0, st2 = 0, st3 = 0, st5 = 0, st6 = 0;
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#define STAMP(x_) x_
#else
#define STAMP(x_)
#endif
  const float* __restrict__ A = p.A + (long long)z * p.sA;
  float* __restrict__ C = p.C + (long long)z * p.sC;
  const char* __restrict__ Wsp = reinterpret_cast<const char*>(p.Wsplit);

  // ---- A staging assignment (as in gemm_kernel: clamped rows, out-of-image conv taps read a page of zeros).  Four consecutive lanes
  // fetch one 64-byte row segment (one request of the texture path).  ds_write_b128 is served in groups of 8 consecutive lanes against
  // 32 banks, i.e. two rows of four 16-byte slots: with the row pitch of 20 dwords (5 slots: what makes the fragment reads conflict-free)
  // rows r and r + 1 overlap in one slot (5 = 4 + 1), rows r and r + 4 do not (20 = 4 mod 8) -- so a lane quad q stages row
  // 8 (q / 8) + 4 (q & 1) + ((q >> 1) & 3) of the pass: every staging store is conflict-free (round 3 paired rows r, r + 1: every
  // store took twice its LDS cycles; PMC: 30 % of the kernel's LDS-active cycles were conflict cycles)
  static_assert(KC == 4 && RPP == 64, "staging lane map below assumes BK = 16");
  const int kc = tid & 3, rbase = ((tid >> 5) << 3) + (((tid >> 2) & 1) << 2) + ((tid >> 3) & 3);
  const float* a_ptr[NA];
  unsigned a_ok[NA];
  float a_mean[NA];          // A_LN: the mean of the row this thread stages (subtracted on the way into LDS)
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    int m = m0 + rbase + RPP * j;
    m = m < Meff ? m : Meff - 1;
    a_mean[j] = 0.f;
    if (AMODE == A_PLAIN) {
      a_ptr[j] = A + (long long)m * p.lda + kc * 4;
      a_ok[j] = 0;
    } else if (AMODE == A_LN) {
      const long long src = p.a_rows ? p.a_rows[m] : m;
      a_ptr[j] = A + src * p.lda + kc * 4;
      a_ok[j] = 0;
    } else {
      const int hw = p.cH * p.cW;
      const int rr = m - (m / hw) * hw;
      const int y = rr / p.cW, x = rr - y * p.cW;
      unsigned bits = 0;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        if (yy >= 0 && yy < p.cH && xx >= 0 && xx < p.cW) bits |= 1u << t;
      }
      a_ok[j] = bits;
      a_ptr[j] = A + (long long)m * p.cC + kc * 4;
    }
  }
  // ---- B staging assignment: chunk q = tid + 256 j.  A column's 96 bytes are three 32-byte units (two 16-byte pieces, one lane pair
  // each); a store group of 8 lanes = 4 units must hit 8 different slots of 8: slot(n, c) = 7 n + c = c - n (mod 8), so the units of
  // the EVEN columns in natural order (column major, unit minor), then those of the odd columns, advance the slot pair by exactly 2
  // per unit -- any four consecutive units tile the 8 slots (round 3 walked all columns in order: the 2 pieces of the next column
  // landed on the banks of piece 0).  Lane quads still fetch 64 contiguous bytes or two 32-byte runs.
  static_assert((3 * BN / 2) % 4 == 0, "a store group must not straddle the two column classes");
  const char* w_ptr[NB];
  int w_lds[NB];
  const long long wpitch = (long long)(p.K / 8) * 48;             // bytes per column of Wsplit
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    int q = tid + 256 * j;
    q = q < NCH ? q : NCH - 1;
    const int t = q >> 1, par = t / (3 * BN / 2), t2 = t - par * (3 * BN / 2), m = t2 / 3, u = t2 - 3 * m;
    const int n = 2 * m + par, c = 2 * u + (q & 1);
    w_ptr[j] = Wsp + (long long)(n0 + n) * wpitch + c * 16;
    w_lds[j] = n * BP + c * 4;                                      // float index inside a Bs buffer
  }
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  v4f ra[NA];
  u32x4 rb[NB];
  int cv_tap = 0, cv_c0 = 0;
  long long cv_off = 0;
  const int seg_k = (AMODE == A_LN && p.seg_k) ? p.seg_k : 0x7fffffff, seg_skip = AMODE == A_LN ? p.seg_rows * p.lda - p.seg_k : 0;
#define S_AOFF(kt_) ((kt_) * BK + (AMODE == A_LN && (kt_) * BK >= seg_k ? seg_skip : 0))   /* the A row's element offset of k-tile kt_ (A_LN: second segment) */
#define S_LOAD_TILE(kt_)                                                                                       \
  {                                                                                                            \
    if (AMODE == A_CONV3) {                                                                                    \
      const int ky = cv_tap / 3 - 1, kx = cv_tap - (cv_tap / 3) * 3 - 1;                                       \
      cv_off = (long long)(ky * p.cW + kx) * p.cC + cv_c0;                                                     \
    }                                                                                                          \
    _Pragma("unroll") for (int j = 0; j < NA; ++j) {                                                           \
      if (AMODE != A_CONV3) ra[j] = *reinterpret_cast<const v4f*>(a_ptr[j] + S_AOFF(kt_));                     \
      else {                                                                                                   \
        const bool ok = (a_ok[j] >> cv_tap) & 1u;                                                              \
        ra[j] = *reinterpret_cast<const v4f*>(ok ? a_ptr[j] + cv_off : p.zeros);                               \
      }                                                                                                        \
    }                                                                                                          \
    _Pragma("unroll") for (int j = 0; j < NB; ++j)                                                             \
      if (64 * wave_u + 256 * j < NCH) rb[j] = *reinterpret_cast<const u32x4*>(w_ptr[j] + (long long)(kt_) * 96); \
    if (AMODE == A_CONV3) {                                                                                    \
      cv_c0 += BK;                                                                                             \
      if (cv_c0 == p.cC) { cv_c0 = 0; cv_tap = cv_tap < 8 ? cv_tap + 1 : 8; }                                  \
    }                                                                                                          \
  }
#define S_STORE_TILE(buf_)                                                                                     \
  {                                                                                                            \
    _Pragma("unroll") for (int j = 0; j < NA; ++j)                                                             \
      *reinterpret_cast<v4f*>(As + (buf_) * LDK * BM + (rbase + RPP * j) * LDK + kc * 4) = AMODE == A_LN ? ra[j] - a_mean[j] : ra[j]; \
    _Pragma("unroll") for (int j = 0; j < NB; ++j)                                                             \
      if (64 * wave_u + 256 * j < NCH) *reinterpret_cast<u32x4*>(Bs + (buf_) * BP * BN + w_lds[j]) = rb[j];    \
  }
  f32x16 acc[MT][NT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][t][r] = 0.f;
  const float* arow0 = As + (wave * MT * 32 + i32) * LDK + half * 8;    // this lane's 8 consecutive k of its row (row tile mi: + mi * 32 * LDK)
  const float* brow0 = Bs + i32 * BP + half * 12;                  // its k-group of column i32: 3 planes x 16 B

  // ---- main loop, software-pipelined around the one barrier per k-tile.  The 6*NT MFMAs of tile kt (column tile after
  // column tile) are split in two groups; everything else is a "filler" placed in the gap after an MFMA:
  //   group A (first half)  | LDS writes of tile kt+1 (registers loaded a whole tile ago), then at once the HBM loads of tile kt+2
  //                         | into the registers just freed: a k-tile lasts ~1700 cycles with three workgroups per CU (a third
  //                         | of the fp32 kernel's), so the loads need the whole tile to land (PMC: with the loads in group B,
  //                         | 39 % of the wave cycles were spent parked at waitcnt / the barrier)
  //   lgkmcnt(0) + barrier
  //   group B (second half) | A fragment read of tile kt+1, the split of that A fragment into the NEXT set of bf16 planes, and
  //                         | -- as soon as a column tile's last MFMA has issued -- the read of its B fragments for tile kt+1
  //                         | into the same registers (one set of B registers, refilled in a rolling way)
  // so no MFMA waits for LDS or for the split: its operands were produced half a tile earlier.
  constexpr int NMF = 6 * NT * MT, PB = NMF / 2;                   // MFMAs per tile (column tile outer, row tile, product); position of the barrier
  constexpr int TM = 6 * MT;                                       // MFMAs per column tile
  constexpr int NST = NA + NB;                                     // staging stores (= loads) per tile
  constexpr int GB = NMF - PB;                                     // gaps of group B
  constexpr int NUA = 2 * NST;                                     // filler units of group A: LDS writes of tile kt+1, then HBM loads of tile kt+2
  constexpr int FPA = (NUA + PB - 1) / PB;
  constexpr int NSP = 4 * MT;                                      // split pairs per tile
  constexpr int FPB = GB >= NSP ? 1 : 2, SOFF = FPB * GB - NSP;    // group B: the split pairs sit in its last gaps (A fragments are read right after the barrier)
  v4f a_lo[MT], a_hi[MT];
  u32x4 bq[NT][3];
  u32x4 pc[MT][3], pn[MT][3];                                      // bf16 planes of the A fragments: current tile, next tile
#define S_READ_B(t_, buf_)                                                                                     \
  { _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                                           \
      bq[t_][pl] = *reinterpret_cast<const u32x4*>(brow0 + (buf_) * BP * BN + (t_) * 32 * BP + pl * 4); }
#define S_READ_A(h_, buf_)   /* unit h_ = 2 * mi + half of the 8 floats */                                     \
  { const int mi = (h_) >> 1;                                                                                  \
    if (((h_) & 1) == 0) a_lo[mi] = *reinterpret_cast<const v4f*>(arow0 + (buf_) * LDK * BM + mi * 32 * LDK);  \
    else a_hi[mi] = *reinterpret_cast<const v4f*>(arow0 + (buf_) * LDK * BM + mi * 32 * LDK + 4); }
  // split floats 2i, 2i+1 of the fragment into dword i of the three planes (round to nearest at every level, exact residuals)
#define S_SPLIT_PAIR(u_, PP_)   /* unit u_ = 4 * mi + pair */                                                  \
  {                                                                                                            \
    const int mi = (u_) >> 2, i_ = (u_) & 3;                                                                   \
    u32x4 (&P_)[3] = PP_[mi];                                                                                  \
    const float x = (i_) == 0 ? a_lo[mi].x : (i_) == 1 ? a_lo[mi].z : (i_) == 2 ? a_hi[mi].x : a_hi[mi].z;     \
    const float y = (i_) == 0 ? a_lo[mi].y : (i_) == 1 ? a_lo[mi].w : (i_) == 2 ? a_hi[mi].y : a_hi[mi].w;     \
    S_SPLIT_BODY(P_, i_, x, y)                                                                                 \
  }
#ifdef NUHTC_GEMM_PROBE_NOSPLIT   // dev probe (wrong results): the A operand's three-way split left out of the k-loop -- the most a producer-side split could give the consumer
#define S_SPLIT_BODY(P_, i_, x, y)                                                                             \
  { P_[0][(i_)] = __float_as_uint(x); P_[1][(i_)] = __float_as_uint(y); P_[2][(i_)] = __float_as_uint(x) ^ __float_as_uint(y); }
#else
#define S_SPLIT_BODY(P_, i_, x, y)                                                                             \
  {                                                                                                            \
    const unsigned w1 = pk_bf16_rn(x, y);                                                                      \
    const float rx = x - __uint_as_float(w1 << 16), ry = y - __uint_as_float(w1 & 0xffff0000u);                \
    const unsigned w2 = pk_bf16_rn(rx, ry);                                                                    \
    const float sx = rx - __uint_as_float(w2 << 16), sy = ry - __uint_as_float(w2 & 0xffff0000u);              \
    P_[0][(i_)] = w1; P_[1][(i_)] = w2; P_[2][(i_)] = pk_bf16_rn(sx, sy);                                      \
  }
#endif
#define S_LOAD_ONE(f_, kt_)                                                                                   \
  {                                                                                                            \
    if ((f_) < NA) {                                                                                           \
      const int j = (f_) < NA ? (f_) : 0;                                                                      \
      if (AMODE != A_CONV3) ra[j] = *reinterpret_cast<const v4f*>(a_ptr[j] + S_AOFF(kt_));                     \
      else {                                                                                                   \
        const bool ok = (a_ok[j] >> cv_tap) & 1u;                                                              \
        ra[j] = *reinterpret_cast<const v4f*>(ok ? a_ptr[j] + cv_off : p.zeros);                               \
      }                                                                                                        \
    } else if ((f_) < NST) {                                                                                   \
      const int j = (f_) >= NA && (f_) < NST ? (f_) - NA : 0;                                                  \
      if (64 * wave_u + 256 * j < NCH) rb[j] = *reinterpret_cast<const u32x4*>(w_ptr[j] + (long long)(kt_) * 96); \
    }                                                                                                          \
  }
#define S_STORE_ONE(f_, buf_)                                                                                  \
  {                                                                                                            \
    if ((f_) < NA) {                                                                                           \
      const int j = (f_) < NA ? (f_) : 0;                                                                      \
      *reinterpret_cast<v4f*>(As + (buf_) * LDK * BM + (rbase + RPP * j) * LDK + kc * 4) = AMODE == A_LN ? ra[j] - a_mean[j] : ra[j]; \
    } else if ((f_) < NST) {                                                                                   \
      const int j = (f_) >= NA && (f_) < NST ? (f_) - NA : 0;                                                  \
      if (64 * wave_u + 256 * j < NCH) *reinterpret_cast<u32x4*>(Bs + (buf_) * BP * BN + w_lds[j]) = rb[j];    \
    }                                                                                                          \
  }
#define S_CONV_BEGIN()                                                                                         \
  if (AMODE == A_CONV3) {                                                                                      \
    const int ky = cv_tap / 3 - 1, kx = cv_tap - (cv_tap / 3) * 3 - 1;                                         \
    cv_off = (long long)(ky * p.cW + kx) * p.cC + cv_c0;                                                       \
  }
#define S_CONV_END()                                                                                           \
  if (AMODE == A_CONV3) {                                                                                      \
    cv_c0 += BK;                                                                                               \
    if (cv_c0 == p.cC) { cv_c0 = 0; cv_tap = cv_tap < 8 ? cv_tap + 1 : 8; }                                    \
  }
#define S_MFMA(i_)                                                                                             \
  {                                                                                                            \
    const int t = (i_) / TM, mi = ((i_) / 6) % MT, e = (i_) % 6;   /* smallest terms first: a3b1, a1b3, a2b2, a2b1, a1b2, a1b1 */ \
    const int ia = e == 0 ? 2 : (e == 2 || e == 3) ? 1 : 0, ib = e == 1 ? 2 : (e == 2 || e == 4) ? 1 : 0;      \
    acc[mi][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pc[mi][ia]), __builtin_bit_cast(bf16x8, bq[t][ib]), \
                                                         acc[mi][t], 0, 0, 0);                                 \
  }
  const int nk = p.K / BK;
  S_LOAD_TILE(0)
  if (AMODE == A_LN) {
    // The rows' statistics from their partials (equal shares of the K elements each; launch_ln_stats: one, a producer GEMM's epilogue:
    // one per 96 columns), merged in order while the first tile's loads are in flight: mean = mean of the partial means, sum of squared
    // deviations = sum of the partial ones + share * sum (partial mean - mean)^2.  Thread r < BM merges row r of the block tile and
    // leaves mean and 1 / sqrt(var + eps) in LDS: the staging threads pick up their rows' means, the epilogue the scale factors.
    if (tid < BM) {
      int m = m0 + tid;
      m = m < Meff ? m : Meff - 1;
      const long long src = p.a_rows ? p.a_rows[m] : m;
      // two segments: the partials of the row are two runs of ln_nparts / 2, `seg_rows` source rows (of ln_nparts / 4 partials each) apart
      const float2* pp = reinterpret_cast<const float2*>(p.ln_part) + src * (p.seg_k ? p.ln_nparts >> 2 : p.ln_nparts);
      const int half_parts = p.seg_k ? p.ln_nparts >> 1 : 16, second = p.seg_k ? p.seg_rows * (p.ln_nparts >> 2) - half_parts : 0;
      float2 q[16];
      float msum = 0.f, m2 = 0.f;
#pragma unroll
      for (int t = 0; t < 16; ++t) if (t < p.ln_nparts) q[t] = pp[t + (t >= half_parts ? second : 0)];
#pragma unroll
      for (int t = 0; t < 16; ++t) if (t < p.ln_nparts) { msum += q[t].x; m2 += q[t].y; }
      const float mean = msum / (float)p.ln_nparts;
      float dev = 0.f;
#pragma unroll
      for (int t = 0; t < 16; ++t) if (t < p.ln_nparts) { const float d = q[t].x - mean; dev = fmaf(d, d, dev); }
      m2 = fmaf((float)(p.K / p.ln_nparts), dev, m2);
      lds[LDSF + tid] = 1.0f / sqrtf(m2 / (float)p.K + 1e-5f);
      lds[LDSF + BM + tid] = mean;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NA; ++j) a_mean[j] = lds[LDSF + BM + rbase + RPP * j];
  }
  S_STORE_TILE(0)
  __syncthreads();
  if (nk > 1) S_LOAD_TILE(1)
#pragma unroll
  for (int h = 0; h < 2 * MT; ++h) S_READ_A(h, 0)
#pragma unroll
  for (int t = 0; t < NT; ++t) S_READ_B(t, 0)
#pragma unroll
  for (int u = 0; u < NSP; ++u) S_SPLIT_PAIR(u, pc)
  int kt = 0;
  STAMP(__builtin_amdgcn_s_waitcnt(0xC07F); st1 = __builtin_amdgcn_s_memtime();)
  // HN: tile kt+1 exists (its LDS writes, fragment reads and split); HN2: tile kt+2 exists (its HBM loads)
#define S_BODY(HN, HN2)                                                                                        \
  {                                                                                                            \
    const int buf = kt & 1;                                                                                    \
    if (HN2) S_CONV_BEGIN()                                                                                    \
    _Pragma("unroll") for (int i = 0; i < PB; ++i) {                                                           \
      S_MFMA(i)                                                                                                \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
      _Pragma("unroll") for (int u = 0; u < FPA; ++u) {                                                        \
        const int f = i * FPA + u;                                                                             \
        STAMP(if (f == 0 && HN) { unsigned long long w0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0x0070); st6 += __builtin_amdgcn_s_memtime() - w0; }) \
        if (f < NST) { if (HN) S_STORE_ONE(f, buf ^ 1) }                                                       \
        else if (f < NUA) { if (HN2) S_LOAD_ONE(f - NST, kt + 2) }                                             \
      }                                                                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
    }                                                                                                          \
    if (HN2) S_CONV_END()                                                                                      \
    __builtin_amdgcn_s_waitcnt(0xC07F);   /* lgkmcnt(0): this wave's LDS writes have landed */                 \
    STAMP(st5 -= __builtin_amdgcn_s_memtime();)                                                                \
    __builtin_amdgcn_s_barrier();                                                                              \
    STAMP(st5 += __builtin_amdgcn_s_memtime();)                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    if (HN) { _Pragma("unroll") for (int h = 0; h < 2 * MT; ++h) S_READ_A(h, buf ^ 1) }                        \
    if (HN) { _Pragma("unroll") for (int t = 0; t < NT; ++t) if (TM * t + TM <= PB) S_READ_B(t, buf ^ 1) }    \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    _Pragma("unroll") for (int i = PB; i < NMF; ++i) {                                                         \
      S_MFMA(i)                                                                                                \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
      _Pragma("unroll") for (int u = 0; u < FPB; ++u) {                                                        \
        const int f = (i - PB) * FPB + u - SOFF;                                                               \
        if (f >= 0 && f < NSP) { if (HN) S_SPLIT_PAIR(f, pn) }                                                 \
      }                                                                                                        \
      if (HN && i % TM == TM - 1 && i / TM * TM + TM > PB) S_READ_B(i / TM, buf ^ 1)   /* this column tile is done: refill */ \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
    }                                                                                                          \
    if (HN) { _Pragma("unroll") for (int mi = 0; mi < MT; ++mi) { pc[mi][0] = pn[mi][0]; pc[mi][1] = pn[mi][1]; pc[mi][2] = pn[mi][2]; } } \
  }
  for (; kt + 2 < nk; ++kt) S_BODY(1, 1)
  if (kt + 1 < nk) { S_BODY(1, 0) ++kt; }
  S_BODY(0, 0)
#undef S_BODY
#undef S_MFMA
#undef S_CONV_BEGIN
#undef S_CONV_END
#undef S_LOAD_ONE
#undef S_STORE_ONE
#undef S_SPLIT_PAIR
#undef S_SPLIT_BODY
#undef S_READ_A
#undef S_READ_B
  __syncthreads();
#undef S_LOAD_TILE
#undef S_AOFF
#undef S_STORE_TILE
  STAMP(st2 = __builtin_amdgcn_s_memtime();)
  gemm_epilogue<MT, NT, WM, WN>(p, acc, lds, C, z, m0, n0, Meff, AMODE == A_LN ? lds + LDSF : nullptr);
#ifdef NUHTC_GEMM_STAMPS
  st3 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0)
  if (lane == 0 && p.stamps) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned long long* o = p.stamps + ((long long)blockIdx.x * 4 + wave) * 8;
    o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3; o[4] = __builtin_amdgcn_s_memtime(); o[5] = hw; o[6] = st5;
    o[7] = __builtin_amdgcn_s_memrealtime() - rt0;   /* 100 MHz ticks of this wave's life (replaces the load-wait stamp) */
  }
#endif
#undef STAMP
}

// block tile 128 x (32·NT): one 32-row strip per wave, NT accumulators (128x128 with 64x64 per wave, 256x64 and BK = 32 were
// measured slower on every shape of the path and are not instantiated)
template <int MT, int NT, int WM, int WN>
static void launch_cfg(const GemmParams& q, int mtiles, hipStream_t s) {
  constexpr int BN = 32 * NT * WN;
  dim3 grid(cdiv(mtiles, 8) * 8 * (q.N / BN), 1, q.batch > 0 ? q.batch : 1);
  if (q.amode == A_CONV3) hipLaunchKernelGGL((gemm_kernel<MT, NT, WM, WN, 16, A_CONV3>), grid, dim3(256), 0, s, q);
  else hipLaunchKernelGGL((gemm_kernel<MT, NT, WM, WN, 16, A_PLAIN>), grid, dim3(256), 0, s, q);
}

// per-device page of zeros for the out-of-image taps of the implicit-GEMM convolutions (allocated on first use)
static const float* zero_page() {
  static std::map<int, float*> pages;
  static std::mutex mu;              // engines of different host threads may convolve for the first time together
  std::lock_guard<std::mutex> lock(mu);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  auto it = pages.find(dev);
  if (it != pages.end()) return it->second;
  float* z = nullptr;
  if (hipMalloc(&z, 256) != hipSuccess || hipMemset(z, 0, 256) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return nullptr;
  pages[dev] = z;
  return z;
}

// ---- split weights: Wsplit[n][k/8][3][8 bf16] on the same device, owned by the engine that uploaded the weight (engine.hip egemm)
static inline unsigned short bf16_rn_bits(float f) {     // round to nearest even; a NaN stays a NaN (the integer carry would turn some into 0 / Inf)
  unsigned u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x0040u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
static inline float bf16_bits_to_float(unsigned short h) {
  unsigned u = (unsigned)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// host split of a constant weight matrix -> device buffer Wsplit[n][k/8][plane][8 bf16] (caller frees with hipFree)
int gemm_make_split(const float* w_host, int N, int K, void** out) {
  if (!w_host || !out || N <= 0 || K <= 0 || K % 8) return NUHTC_E_INVALID;
  std::vector<unsigned short> sp((size_t)N * K * 3);
  for (int n = 0; n < N; ++n)
    for (int k = 0; k < K; ++k) {
      const float b = w_host[(size_t)n * K + k];
      const unsigned short b1 = bf16_rn_bits(b);
      const float r1 = b - bf16_bits_to_float(b1);           // exact
      const unsigned short b2 = bf16_rn_bits(r1);
      const float r2 = r1 - bf16_bits_to_float(b2);          // exact
      const unsigned short b3 = bf16_rn_bits(r2);
      unsigned short* dst = sp.data() + (((size_t)n * (K / 8) + k / 8) * 3) * 8 + (k & 7);
      dst[0] = b1; dst[8] = b2; dst[16] = b3;
    }
  void* d = nullptr;
  if (hipMalloc(&d, sp.size() * 2) != hipSuccess) return NUHTC_E_HIP;
  if (hipMemcpy(d, sp.data(), sp.size() * 2, hipMemcpyHostToDevice) != hipSuccess) { hipFree(d); return NUHTC_E_HIP; }
  *out = d;
  return 0;
}

template <int MT, int NT>
static void launch_split(const GemmParams& q, hipStream_t s) {
  const int mtiles = cdiv(q.M, 128 * MT);
  dim3 grid(cdiv(mtiles, 8) * 8 * (q.N / (32 * NT)), 1, q.batch > 0 ? q.batch : 1);
  if (q.amode == A_LN && q.n_pad > 0) grid.x += cdiv(cdiv(q.n_pad, 64), 8) * 8;          // workgroups that fill the padding rows (gemm_split_kernel)
  // dev: dynamic LDS the kernel never touches (NUHTC_GEMM_LDS_PAD bytes): caps the workgroups of this kernel per CU below what its
  // registers allow, which leaves register and LDS room on every CU for OTHER kernels' workgroups (the memory-bound kernels of the
  // batches in flight) instead of a third GEMM workgroup
  static const int& lds_pad = dev_knob_ref("GEMM_LDS_PAD", 0);
  const unsigned pad = (unsigned)lds_pad;
  if (q.amode == A_CONV3) hipLaunchKernelGGL((gemm_split_kernel<MT, NT, A_CONV3>), grid, dim3(256), pad, s, q);
  else if (q.amode == A_LN) {
    if constexpr (NT == 3 || (NT == 2 && MT == 1)) hipLaunchKernelGGL((gemm_split_kernel<MT, NT, A_LN>), grid, dim3(256), pad, s, q);   // launch_gemm admits A_LN for 96-column tiles and N = 64
  } else hipLaunchKernelGGL((gemm_split_kernel<MT, NT, A_PLAIN>), grid, dim3(256), pad, s, q);
}

bool conv3_fuse_available() {
  static const int& conv_halo = dev_knob_ref("CONV_HALO", 1);
  static const int& conv_fuse = dev_knob_ref("CONV_FUSE", 1);
  return conv_halo && conv_fuse;
}

int launch_gemm(const GemmParams& p, hipStream_t s) {
  if (p.M <= 0) return 0;
  if (p.K % 32 != 0 || p.N % 32 != 0) return NUHTC_E_INVALID;
  if (p.amode == A_CONV3 && (p.cC % 32 != 0 || p.K != 9 * p.cC)) return NUHTC_E_INVALID;
  if ((p.res && p.up) || (p.act == ACT_COS && p.bias)) return NUHTC_E_INVALID;
  // LayerNorm in the A path: the split kernel's 96-column form only (the Swin linears that follow a norm), statistics required
  if ((p.amode == A_LN) != (p.ln_part != nullptr) ||
      (p.amode == A_LN && (!p.Wsplit || (p.N % 96 != 0 && p.N != 64) || p.batch > 1 || p.ln_nparts < 1 || p.ln_nparts > 16 || p.K % p.ln_nparts != 0 ||
                           (p.seg_k && (p.K != 2 * p.seg_k || p.seg_k % 16 != 0 || p.ln_nparts % 4 != 0 || !p.a_rows || p.seg_rows < 1)) || (p.n_pad > 0 && (!p.pad_rows || !p.pad_val)))))
    return NUHTC_E_INVALID;
  // statistics for the next linear's LayerNorm: the split kernel's 96-column form, rows stored whole (plain or row-mapped)
  if (p.stats_out && (!p.Wsplit || p.N % 96 != 0 || p.batch > 1 || (p.store != ST_PLAIN && p.store != ST_ROWMAP) || p.amode == A_CONV3)) return NUHTC_E_INVALID;
  if ((p.res && (long long)p.M * p.ldr >= (1ll << 31)) || (p.up && (long long)p.M * p.N >= (1ll << 31))) return NUHTC_E_INVALID;
  GemmParams q = p;
  if (q.alpha == 0.f) q.alpha = 1.f;
  // a product whose caller passes the weight's exact bf16 split (GemmParams.Wsplit) runs on the bf16 pipe; the engine passes it for
  // products of depth >= 96 only (engine.hip egemm)
  if (q.Wsplit && q.batch > 1) q.Wsplit = nullptr;
  // column-tile width: the widest the shape allows, narrowed for small problems until the launch has enough workgroups for the
  // 256 CUs (a 128x96 tile grid of a few hundred blocks leaves most SIMDs with one wave or none).  Launches whose row count lives
  // on the device (RoI / detection lists) are sized by capacity; about half of it is populated at the bench load.
  int nt = (p.N % 96 == 0) ? 3 : (p.N % 128 == 0) ? 4 : (p.N % 64 == 0) ? 2 : 1;
  {
    static const int& fill = dev_knob_ref("GEMM_FILL", 500);
    static const int& sfill = dev_knob_ref("SPLIT_FILL", 256);
    static const int& force_nt = dev_knob_ref("SPLIT_NT", 0);   // dev: forces the column-tile width where N allows it
    long long mt = cdiv(p.M, 128) * (long long)(p.batch > 0 ? p.batch : 1);
    if (p.m_dev) mt = (mt + 1) / 2;
    // the split kernel pays the operand split once per row tile and column-tile pass, so narrow column tiles cost it more than
    // they cost the fp32 kernel: 96-column tiles are kept at any grid size (measured on every N % 96 == 0 shape of the path,
    // tools/dev/nt_sweep.sh: stage-4 linears 93 / 63 / 153 us against 104 / 70 / 177 with 32 columns), the others narrow only
    // below 256 workgroups (NUHTC_SPLIT_FILL, dev)
    const int limit = q.Wsplit ? sfill : fill;
    if (!(q.Wsplit && nt == 3) && q.amode != A_LN)      // (A_LN exists for 96- and 64-column tiles only)
      while (nt > 1 && mt * (p.N / (32 * nt)) < limit) { if (nt > 2 && p.N % 64 == 0) nt = 2; else nt = 1; }
    if (force_nt > 0 && p.N % (32 * force_nt) == 0) nt = force_nt;
    if (q.amode == A_LN && nt != 3 && nt != 2) return NUHTC_E_INVALID;      // no such instantiation: launch_split would launch nothing
    if (q.stats_out && nt != 3) return NUHTC_E_INVALID;                     // the statistics epilogue writes one partial per 32 * nt columns; every consumer reads 96-column partials
  }
  if (q.amode == A_CONV3 && !q.zeros) {
    q.zeros = zero_page();
    if (!q.zeros) return NUHTC_E_HIP;
  }
  static const int& conv_halo = dev_knob_ref("CONV_HALO", 1);
  const bool halo = conv_halo && conv3_split_supported(q);      // 3x3 convolutions: halo tile in LDS, split once (conv.hip)
  if (p.fuse && !halo) return NUHTC_E_INVALID;                  // a fused pointwise layer exists on that path only (the caller checks conv3_fuse_available)
  const double nb = p.batch > 0 ? p.batch : 1;
  const char* tag = "gemm";
  if (prof_enabled()) {   // per-shape tags, e.g. "gemm_kernel<3>|N288|K96" (strings live for the process lifetime)
    static std::map<long long, std::string> names;
    long long key = ((long long)nt << 40) | ((long long)p.N << 20) | p.K | ((long long)(p.amode == A_CONV3) << 44) | ((long long)halo << 45) | ((long long)(p.fuse ? p.fuse->N2 / 32 : 0) << 46) | ((long long)(p.fuse ? p.fuse->n_more : 0) << 52) |
                    ((long long)(p.amode == A_LN) << 50) | ((long long)(p.stats_out != nullptr) << 51);
    auto it = names.find(key);
    if (it == names.end())
      it = names.emplace(key, "gemm_kernel<" + std::to_string(nt) + ">|N" + std::to_string(p.N) + "|K" + std::to_string(p.K) + (p.amode == A_CONV3 ? (halo ? "|conv3halo" : "|conv3") : "") + (p.fuse ? "+pw" + std::to_string(p.fuse->N2) + (p.fuse->n_more ? "x" + std::to_string(1 + p.fuse->n_more) + "maps" : "") : "") +
                                    (p.amode == A_LN ? "|ln" : "") + (p.stats_out ? "|stats" : "")).first;   // "|ln": LayerNorm in the A path; "|stats": LayerNorm partials in the epilogue
    tag = it->second.c_str();
  }
  // algorithmic work of the launch (a device-side row count is applied when the records are read)
  // bytes: A, W, C once each, plus the row term the epilogue adds (residual: M x N; FPN parent at half resolution: M x N / 4)
  const double row_term = p.res ? (double)p.M * p.N : p.up ? 0.25 * p.M * p.N : 0.0;
  // a fused pointwise layer adds its own product and outputs, and takes away the convolution's output when that is not stored
  const double fuse_flop = p.fuse ? 2.0 * p.M * 64 * (p.fuse->N2 + (p.fuse->outn1 ? 1 : 0)) : 0.0;
  const double fuse_bytes = p.fuse ? 4.0 * p.M * (p.fuse->N2 * (p.fuse->out3 ? 3.0 : 1.0) + (p.fuse->outn1 ? 1 : 0) - (p.fuse->store_out ? 0 : p.N)) : 0.0;
  // further maps through the same fused layers (Conv3Fuse.n_more): their rows' convolution and pointwise work, map in and second output out
  double more_rows = 0.0;
  if (p.fuse && p.fuse->n_more > 0 && p.amode == A_CONV3)
    for (int k = 0; k < p.fuse->n_more && k < 3; ++k) more_rows += (double)(p.M / (p.cH * p.cW)) * p.fuse->more_H[k] * p.fuse->more_W[k];
  const double more_flop = p.fuse ? 2.0 * more_rows * 64 * (p.K + p.fuse->N2) : 0.0, more_bytes = p.fuse ? 4.0 * more_rows * (64 + p.fuse->N2) : 0.0;
  ProfScope ps(tag, 2.0 * p.M * p.N * p.K * nb + fuse_flop + more_flop, 4.0 * nb * ((double)p.M * p.K + (double)p.N * p.K + (double)p.M * p.N + row_term) + fuse_bytes + more_bytes, s);
  ps.device_rows(p.m_dev, p.m_mul, p.M);
#ifdef NUHTC_GEMM_STAMPS
  static unsigned long long* stamp_buf = nullptr;
  if (!stamp_buf && hipMalloc(&stamp_buf, 8ull * 8 * 4 * 65536) != hipSuccess) return NUHTC_E_HIP;
  q.stamps = stamp_buf;
#endif
  if (q.Wsplit && !halo && q.amode != A_CONV3 && !q.m_dev) {
    // tile order inside an XCD (gemm_split_kernel): row tiles first when the split weight (6 B per element) overflows an XCD's 4 MB of L2 and the
    // XCD's eighth of the A rows does not
    static const int& order_knob = dev_knob_ref("SPLIT_ROW_FASTEST", -1);      // dev: 0 / 1 force
    const double w_bytes = 6.0 * q.N * q.K, a_xcd_bytes = 4.0 * ((double)q.M / 8.0) * q.K;
    q.row_fastest = order_knob >= 0 ? order_knob : (w_bytes > 6.0e6 && a_xcd_bytes < 3.0e6);
  }
  if (halo) {
    const int rc = launch_conv3_split(q, s);
    if (rc) return rc;
  } else if (q.Wsplit) {
    // 256-row block tiles (two row tiles per wave: half the weight bytes per flop from L2) where the launch still fills the
    // chip with them; NUHTC_SPLIT_MT=1 / 2 forces one form (dev)
    static const int& force_mt = dev_knob_ref("SPLIT_MT", 0);
    const long long blocks2 = (long long)cdiv(p.M, 256) * (p.N / (32 * nt));
    // (round 3, same-box A/B of every 96-column shape left on this kernel -- the M = 262144 stage-1 linears moved to mlp.hip: 128-row
    // tiles are as fast or faster everywhere, N1152|K384 0.568 vs 0.648 ms, N1536|K384 0.759 vs 0.787, N768|K3072 0.281 vs 0.357; the
    // 256-row form is kept for launches of at least 2048 such tiles, i.e. M >= 131072 at the path's widths)
    // (nuhtc_config.schedule = NUHTC_SCHED_THROUGHPUT: with other batches in flight the under-filled tail of a launch is
    // filled by their kernels and the 256-row tile's lower LDS traffic per MFMA wins: +1.5 % on four batches in flight, -3 % alone)
    const bool mt2 = force_mt ? force_mt == 2 : (!p.m_dev && nt == 3 && (p.throughput || (blocks2 >= 512 && p.M >= 131072)));
    if (mt2 && nt == 3) launch_split<2, 3>(q, s);
    else if (nt == 1) launch_split<1, 1>(q, s);
    else if (nt == 2) launch_split<1, 2>(q, s);
    else if (nt == 3) launch_split<1, 3>(q, s);
    else if (nt == 6) launch_split<1, 6>(q, s);
    else launch_split<1, 4>(q, s);
  } else if (nt == 1) launch_cfg<1, 1, 4, 1>(q, cdiv(p.M, 128), s);
  else if (nt == 2) launch_cfg<1, 2, 4, 1>(q, cdiv(p.M, 128), s);
  else if (nt == 3) launch_cfg<1, 3, 4, 1>(q, cdiv(p.M, 128), s);
  else launch_cfg<1, 4, 4, 1>(q, cdiv(p.M, 128), s);
#ifdef NUHTC_GEMM_STAMPS
  {
    static int cnt = 0, dump_at = -1;   // the 5th launch of the process (or launch NUHTC_STAMP_AT) is dumped to /tmp/stamps.txt
    if (dump_at < 0) { const char* e = getenv("NUHTC_STAMP_AT"); dump_at = e ? atoi(e) : 5; }
    if (++cnt == dump_at) {
      hipDeviceSynchronize();
      int nb_ = cdiv(cdiv(p.M, 128), 8) * 8 * (p.N / (32 * nt));
      if (nb_ > 65536) nb_ = 65536;
      std::vector<unsigned long long> h((size_t)nb_ * 32);
      hipMemcpy(h.data(), stamp_buf, h.size() * 8, hipMemcpyDeviceToHost);
      FILE* f = fopen("/tmp/stamps.txt", "w");
      for (int b = 0; b < nb_; ++b) for (int w = 0; w < 4; ++w) { auto* o = &h[((size_t)b * 4 + w) * 8]; fprintf(f, "%d %d %llu %llu %llu %llu %llu %llu %llu %llu\n", b, w, o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7]); }
      fclose(f);
    }
  }
#endif
  return hipGetLastError() == hipSuccess ? 0 : NUHTC_E_HIP;
}

const char* nuhtc_tu_probe_gemm() {
#ifdef NUHTC_GEMM_NOSTORE
  return "NUHTC_GEMM_NOSTORE";
#elif defined(NUHTC_GEMM_PROBE_NOSPLIT)
  return "NUHTC_GEMM_PROBE_NOSPLIT";
#else
  return nullptr;
#endif
}
