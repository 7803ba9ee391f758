// Internal definitions shared by the HIP translation units of libnuhtc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "../../include/nuhtc_hip.h"

#define WS 7            // Swin window size
#define WS2 49
#define HEAD_DIM 32
#define FPN_C 64
#define FC_C 256

struct HostTensor {
  std::vector<float> data;
  std::vector<int64_t> shape;
};

struct BufInfo {
  void* ptr;
  std::vector<int64_t> shape;
  int dtype;  // 0 f32, 1 i32, 2 u8, 3 u32
};

struct EngineError {
  int code;
  std::string msg;
};

#define HIP_CHECK(e, expr)                                                                              \
  do {                                                                                                  \
    hipError_t _err = (expr);                                                                           \
    if (_err != hipSuccess) {                                                                           \
      (e)->err = std::string(#expr) + " failed: " + hipGetErrorString(_err) + " at " + __FILE__ + ":" + \
                 std::to_string(__LINE__);                                                              \
      return NUHTC_E_HIP;                                                                               \
    }                                                                                                   \
  } while (0)

#define FAIL(e, code, message) \
  do {                         \
    (e)->err = (message);      \
    return (code);             \
  } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// per-kernel HIP-event timing (prof.hip); `tag` must be a string literal
struct ProfScope {
  ProfScope(const char* tag, double flops, double bytes, hipStream_t s);
  ~ProfScope();
  void device_rows(const int* m_dev, int m_mul, int m_cap);   // flops / bytes were given for m_cap rows, the real count is *m_dev * m_mul
  hipStream_t s_;
  int idx_;
};
bool prof_enabled();
int dev_knob(const char* name, int dflt);   // development switch NUHTC_<name> (environment, or nuhtc_dev_knob at run time)
int& dev_knob_ref(const char* name, int dflt);   // the same, as a reference launch code looks up once and reads on every launch

// ----------------------------------------------------------------------------- GEMM (gemm.hip)
// C[row_map(m), n] = epilogue( sum_k A(m,k) * W[n,k] )      fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32
enum AMode { A_PLAIN = 0, A_CONV3 = 1, A_LN = 2 /* plain rows, LayerNorm'ed on their way into the product (ln_stats) */ };
enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_COS = 3 /* relu(v*ri[m]*rj[n]-tau)+tau */ };
enum StoreMode { ST_PLAIN = 0, ST_ROWMAP = 1, ST_DECONV2 = 2 };

// A pointwise (1x1) layer computed in the epilogue of the 3x3 convolution that feeds it (conv.hip): the convolution's output tile
// never leaves the registers on its way into the second product.
struct Conv3Fuse {
  int N2;                 // output channels of the pointwise layer: 32 or 64 (its input is the convolution's 64 channels)
  const void* w2f;        // device image of its [N2][64] weight from conv3_pack_fuse()
  const float* bias2;     // [N2] or null
  float* out2;            // [M][N2]
  int act2;               // ACT_NONE / ACT_RELU
  const float* res2;      // optional (N2 == 64): out3 = res2 + out2
  float* out3;
  const float* wn1;       // optional 64 -> 1 layer on the convolution's output: outn1[m] = wn1 . conv[m] + *bn1
  const float* bn1;
  float* outn1;
  int store_out;          // the convolution's own output is stored to C (1) or dropped (0)
  // Further maps of other sizes that take the SAME convolution and pointwise layer in the same launch (the RPN head over the FPN levels,
  // rpn_head.py:62-68: one set of weights for all levels): map k = more_in[k] -> more_out2[k], the launch's image count of more_H[k] x
  // more_W[k] pixels each.  Only with store_out = 0 and neither out3 nor outn1; the tiles of all maps share one persistent grid.
  int n_more;
  const float* more_in[3];
  float* more_out2[3];
  int more_H[3], more_W[3];
};
int conv3_pack_fuse(const float* w2_host, int N2, void** out_dev);      // caller hipFree()s the image

struct GemmParams {
  const float* A;
  const float* W;      // [N][K] row-major
  const void* Wsplit;  // optional: W split into three bf16 planes (gemm_make_split, kept in the engine's wsplit table); then the bf16 matrix pipe is used
  const float* bias;   // [N] or null
  float* C;
  int M, N, K;
  int lda, ldc;        // row strides in floats
  const int* m_dev;    // optional device-side row count (rows >= min(M, *m_dev * m_mul) are skipped)
  int m_mul;
  int amode;
  // A_LN (split pipe only): row m of the product is LayerNorm(A[src]), src = a_rows ? a_rows[m] : m, with the affine part folded into W and
  // bias by the caller (W' = W diag(gamma), bias' = bias + W beta): the loader subtracts the row's mean from every element it stages and the
  // epilogue multiplies the row's sums by its 1 / sqrt(var + eps) before the bias -- the normalised rows never exist in memory.  The
  // statistics come as `ln_nparts` partials per SOURCE row, ln_part[(src * ln_nparts + t) * 2 + {0, 1}] = {mean, sum of squared deviations}
  // of 1 / ln_nparts of the row's K elements each (launch_ln_stats: one partial; a producer GEMM's epilogue: one per 96 columns, see
  // stats_out), merged in order with the pairwise-update formula by the workgroup that stages the row.
  const float* ln_part;
  int ln_nparts;
  const int* a_rows;
  // A_LN over TWO contiguous segments per row (the PatchMerging gather, transformer.py:363-385: the 2 x 2 tokens of a merged row are two runs of 2 C
  // floats, `seg_rows` source rows apart): seg_k = 0 -> one segment; else K = 2 seg_k, columns k >= seg_k come from A[(src + seg_rows) * lda + k - seg_k]
  // and the row's partials from the two runs of ln_nparts / 2 partials at source rows src and src + seg_rows (each source row has ln_nparts / 4)
  int seg_k, seg_rows;
  // A_LN: `n_pad` rows `pad_rows` of C (row pitch ldc) are filled with pad_val[0..N) by extra workgroups of the launch (the window-padding
  // rows of the QKV image, whose value is the ORIGINAL bias: LN of a zero-padded token is 0)
  const int* pad_rows;
  int n_pad;
  const float* pad_val;
  // producer side (split pipe, N % 96 == 0): besides storing C the epilogue leaves, per destination row and 96-column tile,
  // stats_out[(row * (N / 96) + tile) * 2 + {0, 1}] = {mean, sum of squared deviations} of the 96 values it stored -- the partials the
  // LayerNorm of the NEXT linear's A path needs (round 5: no statistics pass over the tensor at all)
  float* stats_out;
  int cH, cW, cC;      // A_CONV3: NHWC image geometry (rows = b*cH*cW + y*cW + x), K = 9*cC
  const float* zeros;  // A_CONV3: >= 16 bytes of zeros that out-of-image taps read (launch_gemm supplies one when null)
  int act;
  float alpha;         // multiplies the accumulated sum before bias (1.0 default)
  const float* res;    // optional residual, added after activation: res[rmap(m)*ldr + n]
  int ldr;
  const float* up;     // optional FPN top-down term: up[(b, y/2, x/2), n], coarse map is (cH/2... ) see upH/upW
  int upH, upW;        // geometry of the *fine* map whose rows m index (b, y, x); coarse = ceil(/2)
  const int* row_map;  // ST_ROWMAP: destination row per m (-1 = skip); also used for `res`
  int store;
  const float* cos_ri; // ACT_COS: per-row and per-col reciprocal norms
  const float* cos_rj;
  float cos_tau;
  // batching (blockIdx.z)
  int batch;
  long long sA, sW, sC, sRi, sRj;
  unsigned long long* stamps;   // dev instrumentation (-DNUHTC_GEMM_STAMPS), null otherwise
  int row_fastest;              // split kernel: an XCD walks the ROW tiles of one column tile first (chosen by launch_gemm when the weight slice, not the A rows, is what overflows L2)
  int throughput;               // host-side: prefer the block tile that does most work per LDS byte (256-row tiles) over the one that fills the chip soonest
  const Conv3Fuse* fuse;        // host-side: pointwise layer fused into an A_CONV3 product (only on the conv.hip path; else NUHTC_E_INVALID)
};
int launch_gemm(const GemmParams& p, hipStream_t s);
// exact three-way bf16 split of a constant weight matrix (host copy given) -> device buffer Wsplit[n][k/8][plane][8 bf16]; a launch whose
// GemmParams.Wsplit points at it runs on the bf16 matrix pipe (gemm.hip).  The caller owns the buffer (hipFree).
int gemm_make_split(const float* w_host, int N, int K, void** out_dev);

// 3x3 convolution 64 -> 64 with the input halo resident in LDS as bf16 planes (conv.hip); launch_gemm routes A_CONV3 products with a
// split weight there
bool conv3_split_supported(const GemmParams& p);
bool conv3_fuse_available();      // false when the dev knob CONV_HALO routes 3x3 convolutions to the implicit-GEMM path
int launch_conv3_split(const GemmParams& p, hipStream_t s);

// ----------------------------------------------------------------------------- fused FFN half of a Swin block (mlp.hip)
// x_out = x_in + W2 gelu(W1 LN(x_in) + b1) + b2 in one kernel on the split-bf16 pipe; `wstream` from mlp_pack_stream (device copy)
bool mlp_supported(int C);
size_t mlp_stream_bytes(int C);
void mlp_pack_stream(const float* w1, const float* w2, int C, std::vector<unsigned short>& out);
// With `att` (attention output [T][C] in token order), `pstream` (proj_pack_stream of the projection weight, device copy) and `bp`:
// the attention projection and its residual run in front, x' = x_in + Wp att + bp, and the FFN half on x' (x_in == x_out required).
size_t proj_stream_bytes(int C);
void proj_pack_stream(const float* w, int C, std::vector<unsigned short>& out);
int launch_swin_mlp(const float* x_in, float* x_out, const float* ln_g, const float* ln_b, const void* wstream, const float* b1, const float* b2,
                    int T, int C, hipStream_t s, const float* att = nullptr, const void* pstream = nullptr, const float* bp = nullptr,
                    float* stats_out = nullptr /* [T][2]: LayerNorm partial {mean, sum of squared deviations} of every row stored */);

// fused LN1 + QKV linear (mlp.hip, C = 96): qkv window image rows of the T real tokens + the bias rows of the padding tokens
bool lnqkv_supported(int C);
void lnqkv_pack_stream(const float* w, int C, std::vector<unsigned short>& out);
int launch_swin_lnqkv(const float* x, float* qkv, const int* src_tok, const int* dst_row, const int* pad_rows, int n_pad, const float* ln_g, const float* ln_b,
                      const void* wstream, const float* bias, int T, int C, hipStream_t s);

// ----------------------------------------------------------------------------- contours (contour.hip)
// outer contour (cv2 RETR first contour, CHAIN_APPROX_SIMPLE) of every kept instance mask; n: 0 = none, -1 = overflow
int launch_contours(const uint32_t* masks, const uint8_t* keep, const int32_t* counts, int B, int max_per_img, int H, int W,
                    int cap, int16_t* xy, int32_t* n, hipStream_t s);

// compaction of the kept detections of a batch into dense export buffers (contour.hip, nuhtc_export_kept)
struct ExportParams {
  const float* boxes; const int32_t* labels; const int32_t* counts; const uint8_t* keep; const uint32_t* masks;
  const int32_t* contour_n; const int16_t* contour_xy;
  int B, K, words, ccap, cap;
  int32_t* n_out; int64_t* idx; float* boxes_out; int32_t* labels_out; int32_t* cn_out; int16_t* xy_out; uint32_t* words_out;
};
int launch_export_kept(const ExportParams& p, int32_t* pos_scratch, hipStream_t s);
int launch_export_crops(const uint32_t* words, const int32_t* n_dev, int cap, int H, int wpr, int32_t* box, int32_t* area, int32_t* off, int32_t* size_scratch,
                        uint32_t* pool, int pool_cap, hipStream_t s);

// ----------------------------------------------------------------------------- Swin kernels (swin.hip)
// xtab / ytab: dev int4 per output column / row {src index 0, src index 1, weight 0, weight 1} from cv_linear_tables()
void cv_linear_tables(int ssize, int dsize, bool horizontal, std::vector<int>& tab);
int launch_preproc(const uint8_t* tiles, float* img, int B, int th, int tw, int Hn, int Wn, int Hv, int Wv, const int* xtab, const int* ytab, int swap,
                   const float* mean_istd, hipStream_t s);
int launch_patch_embed_tiles(const uint8_t* tiles, int B, int th, int tw, int Hn, int Wn, int Hv, int Wv, const int* xtab, const int* ytab, int swap,
                             const float* mean_istd, const float* w, const float* b, const float* g, const float* beta, float* tok, hipStream_t s);
int launch_patch_embed(const float* img, const float* w, const float* b, const float* g, const float* beta, float* tok,
                       int B, int Hn, int Wn, hipStream_t s);
// LayerNorm of `rows` rows of C channels: dst row m reads src row src_map[m] (or m when src_map==null); src_map[m]<0 -> zeros
int launch_layernorm(const float* x, const int* src_map, const float* g, const float* b, float* y, int rows, int C, hipStream_t s);
// per-row LayerNorm statistics for a product in A_LN mode, as ONE partial per row (GemmParams.ln_part with ln_nparts = 1):
// stats[2r] = mean, stats[2r + 1] = sum of squared deviations of row r of x (rows [0, rows)), computed with the loads, the two passes and
// the summation order of layernorm_kernel.  Stand-alone op and dev fallback: in the engine the producer GEMM's epilogue leaves the partials.
int launch_ln_stats(const float* x, float* stats, int rows, int C, hipStream_t s);
// LN1 of a Swin block over the `rows` window rows: row r with src_map[r] >= 0 is normalised into y[dst_map[r]] (the compact,
// padding-free window order); a padding row writes pad_val[0..3C) (the QKV bias) into pad_dst[r] (the window QKV image).
int launch_layernorm_windows(const float* x, const int* src_map, const int* dst_map, const float* g, const float* b, float* y,
                             float* pad_dst, const float* pad_val, int rows, int C, hipStream_t s);
// PatchMerging gather + LN(4C): out[(b,y2,x2), (kh*2+kw)*C + c] (weights pre-permuted to this order)
int launch_merge_ln(const float* x, const float* g, const float* b, float* y, int B, int H, int W, int C, hipStream_t s);
// window attention: qkv [nWin*49, 3C] -> out rows of C; biasP [nH][4096] / maskP [nW][4096]: the relative-position bias and the
// shift mask packed per lane (engine.hip pack_attn_terms), mask_any [nW] flags the windows with a non-zero mask (maskP null: no shift);
// window row r is written to out row out_map[r] (skipped when negative), or to row r when out_map is null
int launch_window_attn(const float* qkv, const float* biasP, const float* maskP, const int* mask_any, const int* out_map, float* out,
                       int nWinTotal, int nWperImg, int C, int nH, int split_pipe /* 1: bf16 pipe, exactly split operands */, hipStream_t s,
                       const unsigned long long* padbits = nullptr /* split pipe: [nWperImg], bit j = row j of the window is padding: row bias_row is read instead */,
                       int bias_row = 0);

// ----------------------------------------------------------------------------- dense heads (dense.hip)
int launch_sem_fuse(const float* g0, const float* g1, const float* g2, const float* g3, float* out, int B, int H, int W,
                    hipStream_t s);
int launch_conv1x1_n1(const float* x, const float* w, const float* b, float* y, int rows, int C, hipStream_t s);
int launch_rownorm_inv(const float* x, float* inv, int rows, int C, hipStream_t s);
int launch_transpose(const float* x, float* y, int batch, int rows, int cols, hipStream_t s);

// ----------------------------------------------------------------------------- RoI path (roi.hip)
int launch_roi_align(const float* feat, int N, int H, int W, int C, const float* rois, int R, const int* r_dev, int P,
                     float scale, int sr, float* out, int accumulate, hipStream_t s);

// Result-altering dev probes (macros that leave loads, stores or arithmetic out to time what is left: WRONG RESULTS).  Each
// translation unit that has such switches reports the ones it was compiled with (nullptr = none); nuhtc_create refuses to make an
// engine from a library that carries any unless the process says NUHTC_DEV=1 (a stray NUHTC_EXTRA_CFLAGS in the environment of a
// production build must not corrupt outputs silently).
const char* nuhtc_tu_probe_conv();
const char* nuhtc_tu_probe_gemm();
const char* nuhtc_tu_probe_mlp();
const char* nuhtc_tu_probe_swin();
