// Engine state of libnuhtc_hip.so (host side).
#pragma once
#include <unordered_map>

#include "common.h"

struct StageGeom {
  int H, W, C, nH;     // token grid, channels, heads
  int Hp, Wp, nW;      // padded grid (multiple of 7) and windows per image
  int* map[2];         // dev: window-row -> token index (-1 = padding), [max_batch*nW*49], un-shifted / shifted
  int* cidx[2];        // dev: window-row -> index among the non-padding window rows (-1 = padding); tile b's rows are [b*H*W, (b+1)*H*W)
  int* ctok[2];        // dev: that compact index -> token index, [max_batch*H*W]
  int* vrow[2];        // dev: that compact index -> window row
  int* prow[2];        // dev: the padding rows of the window image (map < 0), per-tile lists back to back; npad per tile
  int npad;
  // the attention kernel of the split pipe never reads a padding row: bit j of padbits[shift][window of the image] says row j of the window
  // is one, and the kernel reads row `bias_row` (one row behind the window image of the largest batch, holding the block's QKV bias) instead
  unsigned long long* padbits[2];
  int* brow;           // dev: {bias_row}, the one-entry "padding row list" of the launches that write it
  int bias_row;
  float* mask;         // dev: shift mask, packed per lane [nW][4096] (engine.hip pack_attn_terms)
  int* mask_any;       // dev: [nW] 1 where the window's mask has a non-zero entry
};

struct BlockW {
  float *n1g, *n1b, *relbT /* relative-position bias packed per lane [nH][4096] */, *qkv_w, *qkv_b, *proj_w, *proj_b, *n2g, *n2b, *f1_w, *f1_b, *f2_w, *f2_b;
  // LayerNorm in the A path of the linear that follows it (stages 2-4 on the split pipe, gemm.hip A_LN): the norm's affine part folded into
  // the linear -- W' = W diag(gamma), b' = b + W beta (fp64 on the host, rounded once) -- null where the norm runs as a kernel of its own
  float *qkv_wln, *qkv_bln, *f1_wln, *f1_bln;
  void* qkv_stream;    // fused LN1 + QKV (mlp.hip): k-permuted split planes of qkv_w, null where LN + GEMM run separately
  void* mlp_stream;    // fused FFN half (mlp.hip): chunk-major split planes of f1_w / f2_w, null where the three separate launches run
  void* proj_stream;   // attention projection in front of the fused FFN half (mlp.hip): k-permuted split planes of proj_w, null where proj is a GEMM launch
};

struct nuhtc_engine {
  nuhtc_config cfg;
  int device = 0;
  std::string err;
  std::map<std::string, HostTensor> raw;
  std::map<std::string, std::vector<int64_t>> schema;   // names / shapes nuhtc_load_weight accepts
  std::map<std::string, BufInfo> bufs;
  std::vector<void*> allocs;
  // fp32 weight pointer -> its exact bf16 split (gemm_make_split), filled by upload_gemm_weight at finalize and read by egemm for every
  // launch: the engine's own table (a handle is not thread-safe by contract, so no lock); the buffers are in `allocs`
  // the entry records the [N][K] geometry the split was made for: egemm hands it out only to a launch of exactly that geometry
  struct WSplit { const void* planes; int N, K; };
  std::unordered_map<const float*, WSplit> wsplit;
  size_t bytes_allocated = 0;
  bool finalized = false;
  bool debug_tokens = false;
  int lastB = 0;
  int Hn = 0, Wn = 0;     // network input = pad_shape: img_shape rounded up to a multiple of 32
  int Hv = 0, Wv = 0;     // img_shape = scale_factor * valid image size (boxes are clipped to it)
  int vh = 0, vw = 0;     // ori_shape = the image inside the tile buffer
  int *rs_xtab = nullptr, *rs_ytab = nullptr;   // cv2 linear-resize tables (swin.hip preproc)

  StageGeom st[4];
  // backbone weights
  float *pe_w = nullptr, *pe_b = nullptr, *pe_g = nullptr, *pe_beta = nullptr;
  std::vector<BlockW> blocks[4];
  float *on_g[4], *on_b[4], *mg_g[3], *mg_b[3], *mg_w[3];
  float *mg_wln[3] = {}, *mg_bln[3] = {};   // PatchMerging norm folded into its reduction linear (gemm.hip A_LN, two segments): W diag(gamma), W beta; split pipe only
  int* mg_src[3] = {};                     // dev: merged row -> its top-left token (b, 2 y2, 2 x2) of the stage's token tensor, [max_batch * H/2 * W/2]
  // neck / dense heads
  float *lat_w[4], *lat_b[4], *fpn_w[4], *fpn_b[4];
  float *rpn_w, *rpn_b, *rpn_hw, *rpn_hb;
  void *rpn_hf, *sem_lf[4], *sem_ef;   // pointwise layers as conv3_pack_fuse images (split pipe): computed in the epilogue of the 3x3 convolution before them
  float *sem_lw[4], *sem_lb[4], *sem_cw[4], *sem_cb[4], *sem_ew, *sem_eb, *sem_gw, *sem_gb;
  // roi heads
  float *fc1_w[3], *fc1_b[3], *fc2_w[3], *fc2_b[3], *head_w[3], *head_b[3];   // head_w: [16][256] rows 0..nc+1 normed cls, then 4 reg
  float *mk_w[4], *mk_b[4], *mk_up_w, *mk_up_b, *mk_lw, *mk_lb;

  // workspace
  float *img, *tokA, *tokB, *xw, *qkv, *att, *hid;
  // Output norms of the stages (swin.py:756-762) in the A path of the FPN lateral that consumes them (round 5): every stage keeps its own token
  // buffer and the partials of its final tensor until the neck has run; c[st] then exists only on request (nuhtc_get_buffer computes it)
  float* tok[4] = {};           // tok[0] = tokA, tok[1] = tokB, two more
  float* ln_out[4] = {};        // partials of the stage's final tensor, left by its last block's FFN
  float *lat_wln[4] = {}, *lat_bln[4] = {};   // lateral 1x1 with the output norm folded in: W diag(gamma), b + W beta
  // pre-processing inside the patch embedding (round 5): the tiles of the running call; `img` is then computed only on request (nuhtc_get_buffer)
  const uint8_t* in_tiles = nullptr;
  int in_swap = 0;
  bool img_stale = false;
  bool out_ln_folded = false;   // the last run took that path (c[st] is stale until requested)
  int last_batch = 0;
  float* ln_part2 = nullptr;   // the partials the patch-merging GEMM leaves for the next stage's first block (it READS ln_part in the same launch)
  float* ln_part = nullptr;    // LayerNorm partials of the current token tensor, [token][C / 96][2] = {mean, sum of squared deviations} per 96 channels:
                               // written by the epilogue of the GEMM that produced the tensor (proj, fc2, patch merging), read by the next A_LN linear
  float *c[4], *lat[4], *x[4], *rpn[4], *semg[4];
  float *tmpA, *tmpB, *tmpR, *sem_feat, *sem_pred, *x0sem;   // tmpR: RPN conv output (side stream)
  // proposals / roi path
  int roi_cap = 0;          // rois per tile: max_cc_proposals + rpn_max_per_img
  int cand_cap = 0;         // rpn candidates per tile (<= 4 * nms_pre), det candidates per tile
  int* overflow = nullptr;  // dev int[4]
  int* overflow_host = nullptr;   // pinned int[4]: nuhtc_check copies the flags on the caller's stream (a synchronous hipMemcpy goes through the null stream and waited ~6 ms per call in a loop that keeps four batches in flight)
  int32_t* export_pos = nullptr;   // nuhtc_export_kept scratch [max_batch * max_per_img]
  int32_t* crop_size = nullptr;    // nuhtc_export_crops scratch [max_batch * max_per_img]
  hipStream_t own = nullptr;        // a stream for the caller to run this engine on (nuhtc_stream): created right before the two side streams
  hipStream_t side = nullptr;       // proposal selection / NMS run here, concurrently with the semantic head on the caller's stream
  hipEvent_t ev_rpn = nullptr, ev_side = nullptr, ev_fpn = nullptr;
  hipStream_t side2 = nullptr;      // the big-box RoI kernel runs here, beside the stream kernel (side) and the LDS-tile kernels (caller's stream)
  hipEvent_t ev_side2 = nullptr;
  struct RoiWs* rw = nullptr;
};

// uploads a GEMM weight [N][K] and, unless cfg.matrix_pipe == NUHTC_PIPE_FP32, its exact bf16 split (gemm.hip) into e->wsplit
int upload_gemm_weight(nuhtc_engine* e, float** dst, const std::vector<float>& v, int N, int K);
// launch_gemm with the split of p.W taken from the engine's table (bf16 matrix pipe) when the engine has one
int egemm(nuhtc_engine* e, GemmParams p, hipStream_t s);
int finalize_roi(nuhtc_engine* e);
int alloc_roi_workspace(nuhtc_engine* e);
// rois_fixed != null -> fixed-load mode
int run_roi_path(nuhtc_engine* e, int B, const float* rois_fixed, int n_rois, int n_dets, hipStream_t s, const nuhtc_dets* out);
int run_backbone(nuhtc_engine* e, int B, hipStream_t s);
int run_neck_heads(nuhtc_engine* e, int B, hipStream_t s);
int launch_conv1x1_n1_dev(const float* x, const float* w, const float* b, float* y, int rows_cap, const int* rows_dev, int rows_mul,
                          int sigmoid, hipStream_t s);
