// Fused FFN half of a Swin block:  x <- x + W2 · gelu(W1 · LN2(x) + b1) + b2      (mmdet swin.py:365-367, mmcv FFN; LN = norm2)
// in ONE kernel: the 4C-wide hidden tensor never leaves the CU, LN2 is computed on the token tile the product reads, and the
// operand split of the activations is done once per token tile instead of once per column-tile pass.  HBM traffic per token:
// x read (+ a second, cache-served read for the residual) and x written, against LN2 (r + w) + fc1 (r + 4 w) + fc2 (4 r + r + w)
// of the three separate launches.
//
// Arithmetic: the same exact three-way bf16 split as gemm_split_kernel (six v_mfma_f32_32x32x16_bf16 per 16-deep step, fp32
// accumulation).  Both products are computed TRANSPOSED, with the token on the lane:
//     Hᵀ[32 hidden][32 tokens] = W1[chunk] · Xnᵀ           A operand = W1 rows (LDS), B operand = the token tile's planes (registers)
//     Outᵀ[C][32 tokens]      += W2[:, chunk] · gelu(Hᵀ)    A operand = W2 rows (LDS), B operand = planes of the accumulator of Hᵀ
// An accumulator register r of lane (token, half) holds row (r & 3) + 8 (r >> 2) + 4 half of a 32-row tile; the MFMA's k index of
// lane-half `half`, element e is free to mean any hidden unit as long as both operands agree, so with the weights' k axis
// pre-permuted on the host (k' = 16 u + 8 half + e  <->  k = 16 u + 8 (e / 4) + 4 half + e % 4) registers 8u .. 8u+7 of the Hᵀ
// accumulator ARE the B operand of k-step u of the second product: GELU and the split happen in place, no shuffle, no LDS.  The
// same permutation on W1's k axis makes the x tile's load layout (16-byte pieces at channels 32 t + 8 q + 4 half) both the B operand
// of the first product and the layout of the output accumulators, so the residual add and the store need no transpose either.
//
// Block = NW waves x 32 tokens; the weights stream through LDS in chunks of 32 hidden units ([W1 rows of the chunk | W2 columns
// of the chunk], contiguous in HBM: nuhtc_finalize packs `Wstream`), double buffered, one barrier per chunk.
#include <cstring>
#include <mutex>
#include <set>

#include "common.h"
#include "split_math.h"

struct MlpParams {
  const float* x_in;      // [T][C]
  float* x_out;           // [T][C] (may alias x_in: a wave reads only the rows it writes)
  const float* ln_g;      // [C]
  const float* ln_b;      // [C]
  const char* wstream;    // per chunk: 32 rows x (C/8) k-groups x 3 planes x 8 bf16 of W1p, then C rows x 4 k-groups x 3 planes x 8 bf16 of W2p
  const float* b1;        // [4C]
  const float* b2;        // [C]
  int T;
  unsigned long long* stamps;   // dev instrumentation (-DNUHTC_MLP_STAMPS), null otherwise
  // attention projection in front of the FFN (template PROJ):  x <- x + Wp att + bp  first (mmdet swin.py:360-363), then the FFN half on it
  const float* att;       // [T][C] attention output in TOKEN order (the attention kernel scatters through the window -> token map)
  const char* pstream;    // C rows x (C/8) k-groups x 3 planes x 8 bf16 of the k-permuted projection weight (lnqkv_pack_rows)
  const float* bp;        // [C]
  // LayerNorm partial of every row stored, {mean, sum of squared deviations} over its C = 96 values (what a GEMM's stats_out leaves per 96
  // columns, gemm.hip): the PatchMerging norm behind stage 1 rides in the reduction linear's A path.  Null: none
  float* stats_out;       // [T][2]
};

template <int C>
struct MlpGeom {
  static constexpr int HID = 4 * C, NCHUNK = HID / 32, KS1 = C / 16, CT = C / 32;
  static constexpr int R1 = (C / 8) * 48;           // bytes of a W1 row (all k-groups, 3 planes)
  static constexpr int P1 = R1 + 16;                // LDS pitch: an odd number of 16-byte units = conflict-free ds_read_b128
  static constexpr int R2 = 4 * 48, P2 = R2 + 16;   // a W2 row of one chunk: 4 k-groups
  static constexpr int CHUNK_BYTES = 32 * R1 + C * R2;
  static constexpr int PIECES = CHUNK_BYTES / 16;
  static constexpr int W1BUF = 32 * P1, W2BUF = C * P2;   // bytes of one W1 / W2 chunk image
  static constexpr int W2OFF = 2 * W1BUF;           // W1 images: 2 slots (chunk & 1); W2 images: 3 slots (chunk % 3, see the stagger)
  static constexpr int VEC_OFF = W2OFF + 3 * W2BUF; // then: ln_g[C], ln_b[C], b2[C], b1[HID] as floats
  static constexpr int LDS_BYTES = VEC_OFF + (4 * C + HID) * 4;      // ln_g, ln_b, b2, b1, then the projection bias
  static constexpr int PPIECES = C * (R1 / 16);     // the projection weight image: C rows at pitch P1, in the (not yet used) W2 slots
  static_assert(C * P1 <= 3 * W2BUF, "projection image fits the three W2 slots");
  static_assert((P1 / 16) % 2 == 1 && (P2 / 16) % 2 == 1, "odd pitch");
};

template <int C, int NW, bool STAG, bool PROJ>
__global__ __launch_bounds__(64 * NW, 1) void swin_mlp_kernel(MlpParams p) {
  using G = MlpGeom<C>;
  constexpr int NT = 64 * NW;
  constexpr int NP = (G::PIECES + NT - 1) / NT;     // 16-byte staging pieces per thread and chunk
  extern __shared__ __attribute__((aligned(16))) char lds[];
#ifdef NUHTC_MLP_STAMPS   // dev: per-wave phase times (tools/dev/mlp_stamps.py)
  unsigned long long sk0 = __builtin_amdgcn_s_memtime(), sk1 = 0, sA = 0, sV = 0, sB = 0, sS = 0, sW = 0, sk2 = 0, stt = 0;
#define MSTAMP(x_) x_
#else
#define MSTAMP(x_)
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i32 = lane & 31, half = lane >> 5;
  float* vec = reinterpret_cast<float*>(lds + G::VEC_OFF);
  const float* ldg = vec;
  const float* ldb = vec + C;
  const float* lb2 = vec + 2 * C;
  const float* lb1 = vec + 3 * C;

  // ---- staging assignment: piece q = tid + NT * j of a chunk -> LDS offset inside the chunk's W1 image (q in the W1 part) or W2 image
  int st_lds[NP];
  bool st_w2[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    int q = tid + NT * j;
    q = q < G::PIECES ? q : G::PIECES - 1;
    st_w2[j] = q >= 32 * (G::R1 / 16);
    if (!st_w2[j]) {
      const int row = q / (G::R1 / 16), col = q - row * (G::R1 / 16);
      st_lds[j] = row * G::P1 + col * 16;
    } else {
      const int q2 = q - 32 * (G::R1 / 16);
      const int row = q2 / (G::R2 / 16), col = q2 - row * (G::R2 / 16);
      st_lds[j] = G::W2OFF + row * G::P2 + col * 16;
    }
  }
  const char* wsrc = p.wstream + (long long)tid * 16;
  u32x4 stg[NP];
#define MLP_LOAD_CHUNK(c_)                                                                                        \
  { _Pragma("unroll") for (int j = 0; j < NP; ++j)                                                                \
      if (NT * j + NT <= G::PIECES || tid + NT * j < G::PIECES)                                                   \
        stg[j] = *reinterpret_cast<const u32x4*>(wsrc + (long long)(c_) * G::CHUNK_BYTES + (long long)NT * j * 16); }
#define MLP_STORE_CHUNK(c_)   /* chunk c_: W1 image slot c_ & 1, W2 image slot c_ % 3 */                          \
  { const int o1 = ((c_) & 1) * G::W1BUF, o2 = ((c_) % 3) * G::W2BUF;                                             \
    _Pragma("unroll") for (int j = 0; j < NP; ++j)                                                                \
      if (NT * j + NT <= G::PIECES || tid + NT * j < G::PIECES)                                                   \
        *reinterpret_cast<u32x4*>(lds + st_lds[j] + (st_w2[j] ? o2 : o1)) = stg[j]; }

  // ---- this lane's token row (clamped: rows past T are computed on the last row and never stored)
  const long long tok = (long long)blockIdx.x * (32 * NW) + wave * 32 + i32;
  const bool tok_ok = tok < p.T;
  const float* xrow = p.x_in + (tok_ok ? tok : (long long)p.T - 1) * C + 4 * half;

  v4f xv[G::CT][4];
  if constexpr (PROJ) {
    // ---- attention projection + residual first:  x' = x + Wp att + bp.  The projection weight (C x C, k axis permuted like W1) is staged
    // into the three W2 slots, which chunk 0 does not need before the barrier below; the product is the transposed one of the FFN
    // (out channel rows from LDS, the token tile's attention planes as the B operand), so its accumulators have the layout of the x
    // loads and x' stays in the registers the LN reads.  x' is also written to x_out at once: the FFN's residual re-reads it there
    // (the same lane, microseconds later, served by L2 like the re-read of x_in without the projection).
    constexpr int NPP = (G::PPIECES + NT - 1) / NT;
    u32x4 pst[NPP];
#pragma unroll
    for (int j = 0; j < NPP; ++j)
      if (NT * j + NT <= G::PPIECES || tid + NT * j < G::PPIECES)
        pst[j] = *reinterpret_cast<const u32x4*>(p.pstream + ((long long)tid + (long long)NT * j) * 16);
    const float* arow = p.att + (tok_ok ? tok : (long long)p.T - 1) * C + 4 * half;
    v4f av[G::CT][4];
#pragma unroll
    for (int t = 0; t < G::CT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) av[t][q] = *reinterpret_cast<const v4f*>(arow + 32 * t + 8 * q);
#pragma unroll
    for (int t = 0; t < G::CT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) xv[t][q] = *reinterpret_cast<const v4f*>(xrow + 32 * t + 8 * q);
    for (int i = tid; i < C; i += NT) { vec[i] = p.ln_g[i]; vec[C + i] = p.ln_b[i]; vec[2 * C + i] = p.b2[i]; vec[3 * C + G::HID + i] = p.bp[i]; }
    for (int i = tid; i < G::HID; i += NT) vec[3 * C + i] = p.b1[i];
#pragma unroll
    for (int j = 0; j < NPP; ++j) {
      const int q = tid + NT * j;
      if (NT * j + NT <= G::PPIECES || q < G::PPIECES) {
        const int row = q / (G::R1 / 16), col = q - row * (G::R1 / 16);
        *reinterpret_cast<u32x4*>(lds + G::W2OFF + row * G::P1 + col * 16) = pst[j];
      }
    }
    __syncthreads();
    MLP_LOAD_CHUNK(0)
    u32x4 ap[G::KS1][3];
#pragma unroll
    for (int t = 0; t < G::CT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int sk = 2 * t + (q >> 1), d0 = 2 * (q & 1);
        NUHTC_SPLIT3_INTO(ap[sk], d0, av[t][q].x, av[t][q].y)
        NUHTC_SPLIT3_INTO(ap[sk], d0 + 1, av[t][q].z, av[t][q].w)
      }
    const float* lbp = vec + 3 * C + G::HID;
    const char* wp_lane = lds + G::W2OFF + i32 * G::P1 + half * 48;
#pragma unroll
    for (int t = 0; t < G::CT; ++t) {
      f32x16 pa;
#pragma unroll
      for (int r = 0; r < 16; ++r) pa[r] = 0.f;
#pragma unroll
      for (int sk = 0; sk < G::KS1; ++sk) {
        u32x4 wf[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) wf[pl] = *reinterpret_cast<const u32x4*>(wp_lane + t * 32 * G::P1 + sk * 96 + pl * 16);
        pa = mfma_split6(wf, ap[sk], pa);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const v4f bb = *reinterpret_cast<const v4f*>(lbp + 32 * t + 8 * q + 4 * half);
        // (the GEMM epilogue this replaces: (sum + bias) + residual)
        xv[t][q] = (v4f{pa[4 * q], pa[4 * q + 1], pa[4 * q + 2], pa[4 * q + 3]} + bb) + xv[t][q];
      }
    }
    if (tok_ok) {
      float* orow = p.x_out + tok * C + 4 * half;
#pragma unroll
      for (int t = 0; t < G::CT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<v4f*>(orow + 32 * t + 8 * q) = xv[t][q];
    }
    __syncthreads();               // every wave is done with the projection image before chunk 0's W2 part overwrites it
  } else {
    MLP_LOAD_CHUNK(0)
    for (int i = tid; i < C; i += NT) { vec[i] = p.ln_g[i]; vec[C + i] = p.ln_b[i]; vec[2 * C + i] = p.b2[i]; }
    for (int i = tid; i < G::HID; i += NT) vec[3 * C + i] = p.b1[i];
#pragma unroll
    for (int t = 0; t < G::CT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) xv[t][q] = *reinterpret_cast<const v4f*>(xrow + 32 * t + 8 * q);
  }
  // with PROJ the residual of the epilogue is x', which lives in x_out (x_in == x_out in the engine's in-place use)
  const float* rrow = PROJ ? p.x_out + (tok_ok ? tok : (long long)p.T - 1) * C + 4 * half : xrow;

  // ---- LN2 of the token tile + operand split: xp[s][plane] = B operand of k-step s (8 channels: 32t + 8q + 4 half + i, q = 2(s&1) + e/4)
  u32x4 xp[G::KS1][3];
  {
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < G::CT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) sum += (xv[t][q].x + xv[t][q].y) + (xv[t][q].z + xv[t][q].w);
    sum += __shfl_xor(sum, 32);
    const float mean = sum / (float)C;
    float var = 0.f;
#pragma unroll
    for (int t = 0; t < G::CT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        xv[t][q] -= mean;
        var = fmaf(xv[t][q].x, xv[t][q].x, fmaf(xv[t][q].y, xv[t][q].y, fmaf(xv[t][q].z, xv[t][q].z, fmaf(xv[t][q].w, xv[t][q].w, var))));
      }
    var += __shfl_xor(var, 32);
    const float rstd = 1.0f / sqrtf(var / (float)C + 1e-5f);
    MLP_STORE_CHUNK(0)
    __syncthreads();               // chunk 0 and the vectors are in LDS
    MLP_LOAD_CHUNK(1)
#pragma unroll
    for (int t = 0; t < G::CT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const v4f gg = *reinterpret_cast<const v4f*>(ldg + 32 * t + 8 * q + 4 * half);
        const v4f bb = *reinterpret_cast<const v4f*>(ldb + 32 * t + 8 * q + 4 * half);
        const v4f v = xv[t][q] * rstd * gg + bb;
        const int s = 2 * t + (q >> 1), d0 = 2 * (q & 1);
        NUHTC_SPLIT3_INTO(xp[s], d0, v.x, v.y)
        NUHTC_SPLIT3_INTO(xp[s], d0 + 1, v.z, v.w)
      }
  }

  f32x16 acc[G::CT];
#pragma unroll
  for (int t = 0; t < G::CT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const char* w1_lane = lds + i32 * G::P1 + half * 48;                       // + slot * W1BUF + s * 96 + plane * 16
  const char* w2_lane = lds + G::W2OFF + i32 * G::P2 + half * 48;            // + slot * W2BUF + t * 32 * P2 + u * 96 + plane * 16
  f32x16 h;
  u32x4 hp[2][3];
  // Hᵀ chunk = W1[chunk] · Xnᵀ
#define MLP_GEMM1(c_)                                                                                             \
  { _Pragma("unroll") for (int r = 0; r < 16; ++r) h[r] = 0.f;                                                    \
    const char* wb = w1_lane + ((c_) & 1) * G::W1BUF;                                                             \
    _Pragma("unroll") for (int s = 0; s < G::KS1; ++s) {                                                          \
      u32x4 wf[3];                                                                                                \
      _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) wf[pl] = *reinterpret_cast<const u32x4*>(wb + s * 96 + pl * 16); \
      h = mfma_split6(wf, xp[s], h);                                                                              \
    } }
  // + b1, GELU, split in place: registers 8u .. 8u+7 are the B operand of k-step u of the second product
#ifdef NUHTC_MLP_PROBE_NOGELU      // dev probe (wrong results): the activation left out -- what the erf costs the fused FFN
#define gelu_erf(x_) (x_)
#endif
#define MLP_ACT(c_)                                                                                               \
  { _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                               \
      const v4f bb = *reinterpret_cast<const v4f*>(lb1 + (c_) * 32 + 8 * q + 4 * half);                           \
      const float v0 = gelu_erf(h[4 * q] + bb.x), v1 = gelu_erf(h[4 * q + 1] + bb.y);                             \
      const float v2 = gelu_erf(h[4 * q + 2] + bb.z), v3 = gelu_erf(h[4 * q + 3] + bb.w);                         \
      const int u = q >> 1, d0 = 2 * (q & 1);                                                                     \
      NUHTC_SPLIT3_INTO(hp[u], d0, v0, v1)                                                                        \
      NUHTC_SPLIT3_INTO(hp[u], d0 + 1, v2, v3)                                                                    \
    } }
  // Outᵀ += W2[:, chunk] · gelu(Hᵀ)
#define MLP_GEMM2(c_)                                                                                             \
  { const char* wb = w2_lane + ((c_) % 3) * G::W2BUF;                                                             \
    _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                                 \
      _Pragma("unroll") for (int t = 0; t < G::CT; ++t) {                                                         \
        u32x4 wf[3];                                                                                              \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) wf[pl] = *reinterpret_cast<const u32x4*>(wb + t * 32 * G::P2 + u * 96 + pl * 16); \
        acc[t] = mfma_split6(wf, hp[u], acc[t]);                                                                  \
      } }

  // Stagger: the two waves of a SIMD run the same chunk loop between the same barriers, so without it both are in their matrix
  // phase together and in their GELU / split (VALU) phase together.  The second half of the waves (4..7: the SIMD partners of
  // 0..3) defers the second product of a chunk by one iteration -- B(j-1), A(j), V(j) against A(j), V(j), B(j) -- so that one
  // partner's VALU phase runs beside the other's MFMAs.  Their W2 image of chunk j-1 must survive iteration j: three W2 slots.
  const bool stag = STAG && __builtin_amdgcn_readfirstlane(wave) >= NW / 2;
  MSTAMP(sk1 = __builtin_amdgcn_s_memtime();)
#pragma unroll 1
  for (int ch = 0; ch < G::NCHUNK; ++ch) {
    MSTAMP(stt = __builtin_amdgcn_s_memtime();)
    if (stag && ch > 0) MLP_GEMM2(ch - 1)
    MSTAMP(__builtin_amdgcn_sched_barrier(0); { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); sB += n_ - stt; stt = n_; } __builtin_amdgcn_sched_barrier(0);)
    MLP_GEMM1(ch)
    MSTAMP(__builtin_amdgcn_sched_barrier(0); { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); sA += n_ - stt; stt = n_; } __builtin_amdgcn_sched_barrier(0);)
    MLP_ACT(ch)
    MSTAMP(__builtin_amdgcn_sched_barrier(0); { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); sV += n_ - stt; stt = n_; } __builtin_amdgcn_sched_barrier(0);)
    if (!stag) MLP_GEMM2(ch)
    MSTAMP(__builtin_amdgcn_sched_barrier(0); { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); sB += n_ - stt; stt = n_; } __builtin_amdgcn_sched_barrier(0);)
    // next chunk into its slots (their last readers passed the barrier at the end of the previous iteration)
    if (ch + 1 < G::NCHUNK) {
      MLP_STORE_CHUNK(ch + 1)
      if (ch + 2 < G::NCHUNK) MLP_LOAD_CHUNK(ch + 2)
    }
    MSTAMP(__builtin_amdgcn_sched_barrier(0); { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); sS += n_ - stt; stt = n_; } __builtin_amdgcn_sched_barrier(0);)
    __syncthreads();
    MSTAMP(__builtin_amdgcn_sched_barrier(0); { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); sW += n_ - stt; stt = n_; } __builtin_amdgcn_sched_barrier(0);)
  }
  if (stag) MLP_GEMM2(G::NCHUNK - 1)
  MSTAMP(sk2 = __builtin_amdgcn_s_memtime();)
#undef MLP_GEMM1
#undef MLP_ACT
#ifdef NUHTC_MLP_PROBE_NOGELU
#undef gelu_erf
#endif
#undef MLP_GEMM2
#undef MLP_LOAD_CHUNK
#undef MLP_STORE_CHUNK

  // ---- epilogue: + b2 + x (re-read: served by L2 / Infinity Cache, the tile was read a few microseconds ago), store
  // (PROJ: the residual is x', which this lane stored after its x lines had been loaded.  Store and re-read go through the same CU's vector
  // L1, which is coherent for the accesses of one CU -- the AMDGPU memory model needs no invalidate at workgroup scope in the default
  // non-tgsplit mode -- so the plain load below sees x'.  An agent-scope acquire or per-element agent-scope loads here were measured: +0.08-0.10 ms
  // per step for nothing.)
  if (tok_ok) {
    float* orow = p.x_out + tok * C + 4 * half;
    float s_piv = 0.f, s_1 = 0.f, s_2 = 0.f;       // stats_out: sums of (v - pivot) and (v - pivot)^2, pivot = the row's first value
#pragma unroll
    for (int t = 0; t < G::CT; ++t) {
      v4f res[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) res[q] = *reinterpret_cast<const v4f*>(rrow + 32 * t + 8 * q);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const v4f bb = *reinterpret_cast<const v4f*>(lb2 + 32 * t + 8 * q + 4 * half);
        v4f o = {acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
        o = o + bb + res[q];
        *reinterpret_cast<v4f*>(orow + 32 * t + 8 * q) = o;
        if (p.stats_out) {
          if (t == 0 && q == 0) s_piv = __shfl(o.x, i32);      // the half-wave 0 lane of this token holds channel 0
          const v4f d = o - s_piv;
          s_1 += (d.x + d.y) + (d.z + d.w);
          s_2 = fmaf(d.x, d.x, fmaf(d.y, d.y, fmaf(d.z, d.z, fmaf(d.w, d.w, s_2))));
        }
      }
    }
    if (p.stats_out) {      // the token's other 48 channels are in lane ^ 32 (same token: active as well)
      s_1 += __shfl_xor(s_1, 32);
      s_2 += __shfl_xor(s_2, 32);
      const float dm = s_1 * (1.0f / C);
      if (half == 0) *reinterpret_cast<float2*>(p.stats_out + tok * 2) = make_float2(s_piv + dm, fmaxf(s_2 - s_1 * dm, 0.f));
    }
  }
#ifdef NUHTC_MLP_STAMPS
  __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0): the stores have been issued and acknowledged
  if (lane == 0 && p.stamps) {
    unsigned long long* o = p.stamps + ((long long)blockIdx.x * NW + wave) * 8;
    o[0] = sk1 - sk0; o[1] = sA; o[2] = sV; o[3] = sB; o[4] = sS; o[5] = sW; o[6] = __builtin_amdgcn_s_memtime() - sk2; o[7] = sk0;
  }
#endif
#undef MSTAMP
}

// ======================================================================================================================
// Fused LN1 + QKV linear of a Swin block (C = 96):  qkv[vrow[r]] = Wqkv · LN1(x[ctok[r]]) + b   for the T real tokens r in window
// order (mmdet swin.py:341-356,88: norm1, pad, roll, window partition, qkv).  Replaces layernorm_windows + the K = 96 GEMM (three
// column-tile passes that each re-read and re-split the normalised rows): one read of x, LN and the operand split once per token
// tile, the 288 output features in three chunks of 96 (q, k, v) whose weights stream through a double-buffered LDS image.
// Same transposed arrangement as swin_mlp_kernel: token on the lane, W rows as the MFMA's A operand, so a lane owns 4 x 4
// consecutive features of its token per 32-feature tile and stores them as 16-byte pieces.  The padding rows of the window image
// (their qkv is the bias) are filled by qkv_pad_rows_kernel.
struct LnQkvParams {
  const float* x;         // [tokens][C]
  float* qkv;             // window image [rows][3C]
  const int* src_tok;     // [T] compact window-order row -> token
  const int* dst_row;     // [T] compact row -> row of the window image
  const float* ln_g;
  const float* ln_b;
  const char* wstream;    // per chunk of 96 features: 96 rows x (C/8) k-groups x 3 planes x 8 bf16 of the k-permuted weight
  const float* bias;      // [3C]
  int T;
};

template <int C>
struct QkvGeom {
  static constexpr int KS1 = C / 16, CT = C / 32;
  static constexpr int R1 = (C / 8) * 48, P1 = R1 + 16;
  static constexpr int CHUNK_ROWS = C;                 // 96 features = 3 tiles of 32
  static constexpr int CHUNK_BYTES = CHUNK_ROWS * R1, PIECES = CHUNK_BYTES / 16;
  static constexpr int WBUF = CHUNK_ROWS * P1;
  static constexpr int VEC_OFF = 2 * WBUF;             // then ln_g[C], ln_b[C], bias[3C]
  static constexpr int LDS_BYTES = VEC_OFF + 5 * C * 4;
};

template <int C, int NW>
__global__ __launch_bounds__(64 * NW, 1) void swin_lnqkv_kernel(LnQkvParams p) {
  using G = QkvGeom<C>;
  constexpr int NT = 64 * NW;
  constexpr int NP = (G::PIECES + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i32 = lane & 31, half = lane >> 5;
  float* vec = reinterpret_cast<float*>(lds + G::VEC_OFF);
  const float* ldg = vec;
  const float* ldb = vec + C;
  const float* lbias = vec + 2 * C;

  int st_lds[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    int q = tid + NT * j;
    q = q < G::PIECES ? q : G::PIECES - 1;
    const int row = q / (G::R1 / 16), col = q - row * (G::R1 / 16);
    st_lds[j] = row * G::P1 + col * 16;
  }
  const char* wsrc = p.wstream + (long long)tid * 16;
  u32x4 stg[NP];
#define QKV_LOAD_CHUNK(c_)                                                                                        \
  { _Pragma("unroll") for (int j = 0; j < NP; ++j)                                                                \
      if (NT * j + NT <= G::PIECES || tid + NT * j < G::PIECES)                                                   \
        stg[j] = *reinterpret_cast<const u32x4*>(wsrc + (long long)(c_) * G::CHUNK_BYTES + (long long)NT * j * 16); }
#define QKV_STORE_CHUNK(c_)                                                                                       \
  { _Pragma("unroll") for (int j = 0; j < NP; ++j)                                                                \
      if (NT * j + NT <= G::PIECES || tid + NT * j < G::PIECES)                                                   \
        *reinterpret_cast<u32x4*>(lds + ((c_) & 1) * G::WBUF + st_lds[j]) = stg[j]; }
#define QKV_RAW_BARRIER() { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0) */ __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }

  QKV_LOAD_CHUNK(0)
  for (int i = tid; i < C; i += NT) { vec[i] = p.ln_g[i]; vec[C + i] = p.ln_b[i]; }
  for (int i = tid; i < 3 * C; i += NT) vec[2 * C + i] = p.bias[i];

  const long long r = (long long)blockIdx.x * (32 * NW) + wave * 32 + i32;      // compact window-order row of this lane
  const bool r_ok = r < p.T;
  const long long rc = r_ok ? r : (long long)p.T - 1;
  const float* xrow = p.x + (long long)p.src_tok[rc] * C + 4 * half;
  float* orow = p.qkv + (long long)p.dst_row[rc] * (3 * C) + 4 * half;

  u32x4 xp[G::KS1][3];
  {
    v4f xv[G::CT][4];
#pragma unroll
    for (int t = 0; t < G::CT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) xv[t][q] = *reinterpret_cast<const v4f*>(xrow + 32 * t + 8 * q);
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < G::CT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) sum += (xv[t][q].x + xv[t][q].y) + (xv[t][q].z + xv[t][q].w);
    sum += __shfl_xor(sum, 32);
    const float mean = sum / (float)C;
    float var = 0.f;
#pragma unroll
    for (int t = 0; t < G::CT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        xv[t][q] -= mean;
        var = fmaf(xv[t][q].x, xv[t][q].x, fmaf(xv[t][q].y, xv[t][q].y, fmaf(xv[t][q].z, xv[t][q].z, fmaf(xv[t][q].w, xv[t][q].w, var))));
      }
    var += __shfl_xor(var, 32);
    const float rstd = 1.0f / sqrtf(var / (float)C + 1e-5f);
    QKV_STORE_CHUNK(0)
    __syncthreads();
    QKV_LOAD_CHUNK(1)
#pragma unroll
    for (int t = 0; t < G::CT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const v4f gg = *reinterpret_cast<const v4f*>(ldg + 32 * t + 8 * q + 4 * half);
        const v4f bb = *reinterpret_cast<const v4f*>(ldb + 32 * t + 8 * q + 4 * half);
        const v4f v = xv[t][q] * rstd * gg + bb;
        const int s = 2 * t + (q >> 1), d0 = 2 * (q & 1);
        NUHTC_SPLIT3_INTO(xp[s], d0, v.x, v.y)
        NUHTC_SPLIT3_INTO(xp[s], d0 + 1, v.z, v.w)
      }
  }

  // ---- 3 chunks x 3 feature tiles x KS1 k-steps; weight fragments read two steps ahead (three register sets)
  const char* w_lane = lds + i32 * G::P1 + half * 48;          // + slot * WBUF + tile * 32 * P1 + s * 96 + plane * 16
  constexpr int NSTEP = 3 * G::CT * G::KS1;                    // steps of 6 MFMAs over the whole kernel (54)
  constexpr int SPC = G::CT * G::KS1;                          // steps per chunk (18)
  u32x4 fw[3][3];
#define QKV_READ(k_, set_, step_)                                                                                 \
  { const int ch_ = (step_) / SPC, tl_ = ((step_) % SPC) / G::KS1, s_ = (step_) % G::KS1;                         \
    fw[set_][k_] = *reinterpret_cast<const u32x4*>(w_lane + (ch_ & 1) * G::WBUF + tl_ * 32 * G::P1 + s_ * 96 + (k_) * 16); }
#pragma unroll
  for (int k = 0; k < 3; ++k) QKV_READ(k, 0, 0)
#pragma unroll
  for (int k = 0; k < 3; ++k) QKV_READ(k, 1, 1)
  f32x16 h;
#pragma unroll
  for (int step = 0; step < NSTEP; ++step) {
    const int ch = step / SPC, tl = (step % SPC) / G::KS1, s = step % G::KS1;
    if (s == 0) {
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) h[rr] = 0.f;
    }
    // weights of chunk ch+1: into the other image at the first step of chunk ch (registers loaded a chunk earlier); the barrier
    // sits before the step whose look-ahead reads are the first ones of chunk ch+1 (two steps before the chunk's end)
    if (step % SPC == 0 && ch < 2) {
      QKV_STORE_CHUNK(ch + 1)
      if (ch < 1) QKV_LOAD_CHUNK(ch + 2)
    }
    if (step % SPC == SPC - 2 && ch < 2) QKV_RAW_BARRIER()
    {
      constexpr int iw_[6] = {0, 2, 1, 0, 1, 0}, ia_[6] = {2, 0, 1, 1, 0, 0};   // (activation, weight) plane order of gemm_split_kernel
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        h = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[step % 3][iw_[k]]), __builtin_bit_cast(bf16x8, xp[s][ia_[k]]), h, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (k < 3 && step + 2 < NSTEP) QKV_READ(k, (step + 2) % 3, step + 2)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (s == G::KS1 - 1 && r_ok) {      // feature tile done: + bias, store (features ch * C + 32 tl + 8 q + 4 half + 0..3)
      const int f0 = ch * C + 32 * tl;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const v4f bb = *reinterpret_cast<const v4f*>(lbias + f0 + 8 * q + 4 * half);
        const v4f o = v4f{h[4 * q], h[4 * q + 1], h[4 * q + 2], h[4 * q + 3]} + bb;
        *reinterpret_cast<v4f*>(orow + f0 + 8 * q) = o;
      }
    }
  }
#undef QKV_READ
#undef QKV_LOAD_CHUNK
#undef QKV_STORE_CHUNK
#undef QKV_RAW_BARRIER
}

// rows of the window image that hold padding tokens: qkv = bias (LN of a zero-padded token is 0, swin.py:341-343)
__global__ __launch_bounds__(256) void qkv_pad_rows_kernel(float* __restrict__ qkv, const int* __restrict__ rows, int n, const float* __restrict__ bias, int C3) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  v4f* dst = reinterpret_cast<v4f*>(qkv + (long long)rows[i] * C3);
  const v4f* src = reinterpret_cast<const v4f*>(bias);
  for (int c = lane; c < C3 / 4; c += 64) dst[c] = src[c];
}

// ---- host side: the weight stream of one block (permuted k axes, split planes, chunk-major)
static inline unsigned short mlp_bf16_rn(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x0040u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
static inline float mlp_bf16_f(unsigned short h) {
  unsigned u = (unsigned)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static inline void mlp_split3(float b, unsigned short* dst) {   // dst[0], dst[8], dst[16] = planes 1..3 of one element of a k-group
  const unsigned short b1 = mlp_bf16_rn(b);
  const float r1 = b - mlp_bf16_f(b1);
  const unsigned short b2 = mlp_bf16_rn(r1);
  const float r2 = r1 - mlp_bf16_f(b2);
  dst[0] = b1; dst[8] = b2; dst[16] = mlp_bf16_rn(r2);
}
// position k' of a permuted 16-group -> original position: k' = 8 hh + e  <->  8 (e / 4) + 4 hh + e % 4
static inline int mlp_perm16(int kp) { const int hh = (kp >> 3) & 1, e = kp & 7; return 8 * (e >> 2) + 4 * hh + (e & 3); }

size_t mlp_stream_bytes(int C) { return (size_t)(4 * C / 32) * ((size_t)32 * (C / 8) * 48 + (size_t)C * 4 * 48); }

// w1 [4C][C], w2 [C][4C] (row-major, fp32) -> the stream swin_mlp_kernel reads
void mlp_pack_stream(const float* w1, const float* w2, int C, std::vector<unsigned short>& out) {
  const int HID = 4 * C, nch = HID / 32;
  out.assign(mlp_stream_bytes(C) / 2, 0);
  unsigned short* o = out.data();
  for (int j = 0; j < nch; ++j) {
    for (int r = 0; r < 32; ++r)
      for (int kg = 0; kg < C / 8; ++kg) {
        for (int e = 0; e < 8; ++e) {
          const int kp = 8 * kg + e, k = (kp & ~15) + mlp_perm16(kp & 15);
          mlp_split3(w1[(size_t)(32 * j + r) * C + k], o + e);
        }
        o += 24;
      }
    for (int c = 0; c < C; ++c)
      for (int kg = 0; kg < 4; ++kg) {
        for (int e = 0; e < 8; ++e) {
          const int kp = 8 * kg + e, k = 32 * j + (kp & ~15) + mlp_perm16(kp & 15);
          mlp_split3(w2[(size_t)c * HID + k], o + e);
        }
        o += 24;
      }
  }
}

bool mlp_supported(int C) { return C == 96; }

size_t proj_stream_bytes(int C) { return (size_t)C * (C / 8) * 48; }
// w [C][C] -> rows in order, k axis permuted like the MLP's W1 (the image swin_mlp_kernel<PROJ> stages)
void proj_pack_stream(const float* w, int C, std::vector<unsigned short>& out) {
  out.assign(proj_stream_bytes(C) / 2, 0);
  unsigned short* o = out.data();
  for (int r = 0; r < C; ++r)
    for (int kg = 0; kg < C / 8; ++kg) {
      for (int e = 0; e < 8; ++e) {
        const int kp = 8 * kg + e, k = (kp & ~15) + mlp_perm16(kp & 15);
        mlp_split3(w[(size_t)r * C + k], o + e);
      }
      o += 24;
    }
}

int launch_swin_mlp(const float* x_in, float* x_out, const float* ln_g, const float* ln_b, const void* wstream, const float* b1, const float* b2,
                    int T, int C, hipStream_t s, const float* att, const void* pstream, const float* bp, float* stats_out) {
  { static const int& skip_ = dev_knob_ref("SKIP", 0); if (skip_ & 16) return 0; }   // dev: ablation of the step (tools/dev/r04_ablate.py)
  if (T <= 0) return 0;
  if (!mlp_supported(C) || !wstream) return NUHTC_E_INVALID;
  const bool proj = att != nullptr;
  if (proj && (!pstream || !bp)) return NUHTC_E_INVALID;
  MlpParams p{x_in, x_out, ln_g, ln_b, reinterpret_cast<const char*>(wstream), b1, b2, T, nullptr, att, reinterpret_cast<const char*>(pstream), bp, stats_out};
#ifdef NUHTC_MLP_STAMPS
  static unsigned long long* stamp_buf = nullptr;
  if (!stamp_buf && hipMalloc(&stamp_buf, 8ull * 8 * 8 * 4096) != hipSuccess) return NUHTC_E_HIP;
  p.stamps = stamp_buf;
#endif
  // algorithmic work: both products; bytes: x read and written once (+ the weight stream once)
  // (with the projection: + 2 T C^2 flop, + the attention rows read once)
  ProfScope ps("swin_mlp", (proj ? 18.0 : 16.0) * T * C * C, (proj ? 12.0 : 8.0) * T * C + (double)mlp_stream_bytes(C) + (proj ? (double)proj_stream_bytes(C) : 0.0), s);
  constexpr int NW = 8;
  static const int& stagger = dev_knob_ref("MLP_STAGGER", 1);
  auto kern = proj ? (stagger ? &swin_mlp_kernel<96, NW, true, true> : &swin_mlp_kernel<96, NW, false, true>)
                   : (stagger ? &swin_mlp_kernel<96, NW, true, false> : &swin_mlp_kernel<96, NW, false, false>);
  {
    static std::set<std::pair<int, const void*>> done;      // the kernel needs more than the default 64 KB of dynamic LDS: raised once per device
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return NUHTC_E_HIP;
    const auto key = std::make_pair(dev, reinterpret_cast<const void*>(kern));
    if (!done.count(key)) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, MlpGeom<96>::LDS_BYTES) != hipSuccess)
        return NUHTC_E_HIP;
      done.insert(key);
    }
  }
  hipLaunchKernelGGL(kern, dim3(cdiv(T, 32 * NW)), dim3(64 * NW), MlpGeom<96>::LDS_BYTES, s, p);
#ifdef NUHTC_MLP_STAMPS
  {
    static int cnt = 0, dump_at = -1;   // launch NUHTC_STAMP_AT (default 40) of the process is dumped to /tmp/mlp_stamps.txt
    if (dump_at < 0) { const char* e = getenv("NUHTC_STAMP_AT"); dump_at = e ? atoi(e) : 40; }
    if (++cnt == dump_at) {
      hipDeviceSynchronize();
      int nb = cdiv(T, 32 * NW);
      if (nb > 4096) nb = 4096;
      std::vector<unsigned long long> h((size_t)nb * NW * 8);
      hipMemcpy(h.data(), stamp_buf, h.size() * 8, hipMemcpyDeviceToHost);
      FILE* f = fopen("/tmp/mlp_stamps.txt", "w");
      for (int b = 0; b < nb; ++b) for (int w = 0; w < NW; ++w) { auto* o = &h[((size_t)b * NW + w) * 8]; fprintf(f, "%d %d %llu %llu %llu %llu %llu %llu %llu %llu\n", b, w, o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7]); }
      fclose(f);
    }
  }
#endif
  return hipGetLastError() == hipSuccess ? 0 : NUHTC_E_HIP;
}

// ---- LN1 + QKV (C = 96)
bool lnqkv_supported(int C) { return C == 96; }
size_t lnqkv_stream_bytes(int C) { return (size_t)3 * C * (C / 8) * 48; }
// w [3C][C] -> chunk-major stream (rows in order: the three chunks are q, k, v), k axis permuted like the MLP's W1
void lnqkv_pack_stream(const float* w, int C, std::vector<unsigned short>& out) {
  out.assign(lnqkv_stream_bytes(C) / 2, 0);
  unsigned short* o = out.data();
  for (int r = 0; r < 3 * C; ++r)
    for (int kg = 0; kg < C / 8; ++kg) {
      for (int e = 0; e < 8; ++e) {
        const int kp = 8 * kg + e, k = (kp & ~15) + mlp_perm16(kp & 15);
        mlp_split3(w[(size_t)r * C + k], o + e);
      }
      o += 24;
    }
}

int launch_swin_lnqkv(const float* x, float* qkv, const int* src_tok, const int* dst_row, const int* pad_rows, int n_pad, const float* ln_g, const float* ln_b,
                      const void* wstream, const float* bias, int T, int C, hipStream_t s) {
  { static const int& skip_ = dev_knob_ref("SKIP", 0); if (skip_ & 16) return 0; }   // dev: ablation of the step (tools/dev/r04_ablate.py)
  if (T <= 0) return 0;
  if (!lnqkv_supported(C) || !wstream) return NUHTC_E_INVALID;
  LnQkvParams p{x, qkv, src_tok, dst_row, ln_g, ln_b, reinterpret_cast<const char*>(wstream), bias, T};
  ProfScope ps("swin_lnqkv", 6.0 * T * C * C, 16.0 * T * C + (double)lnqkv_stream_bytes(C), s);
  constexpr int NW = 8;
  auto kern = &swin_lnqkv_kernel<96, NW>;
  {
    static std::set<int> done;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return NUHTC_E_HIP;
    if (!done.count(dev)) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, QkvGeom<96>::LDS_BYTES) != hipSuccess) return NUHTC_E_HIP;
      done.insert(dev);
    }
  }
  if (n_pad > 0) hipLaunchKernelGGL(qkv_pad_rows_kernel, dim3(cdiv(n_pad, 4)), dim3(256), 0, s, qkv, pad_rows, n_pad, bias, 3 * C);
  hipLaunchKernelGGL(kern, dim3(cdiv(T, 32 * NW)), dim3(64 * NW), QkvGeom<96>::LDS_BYTES, s, p);
  return hipGetLastError() == hipSuccess ? 0 : NUHTC_E_HIP;
}

const char* nuhtc_tu_probe_mlp() {
#ifdef NUHTC_MLP_PROBE_NOGELU
  return "NUHTC_MLP_PROBE_NOGELU";
#else
  return nullptr;
#endif
}
