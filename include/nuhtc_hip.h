/* nuhtc_hip.h — C ABI of libnuhtc_hip.so, the MI355X (gfx950) engine for NuHTC's htc_lite_swin
 * tile-inference path.
 *
 * What it replaces in the reference (paths relative to the boyden/NuHTC tree, `mmdet/` =
 * thirdparty/mmdetection/mmdet/): the reference has no native code of its own; its native seam is
 * mmcv's pybind extension (`mmcv._ext`: roi_align_forward, nms; call sites
 * mmdet/models/roi_heads/roi_extractors/base_roi_extractor.py:53-58, mmdet/models/dense_heads/rpn_head.py:232,
 * nuhtc/models/bbox_head.py:93) plus torch ATen.  This library replaces the whole device side of
 * `inference_detector(model, imgs)` (mmdet/apis/inference.py:90-153 ->
 * nuhtc/models/htc_cus.py:110-121 `simple_test`), i.e. everything between "uint8 tiles" and
 * "(bbox_results, segm_results)", and additionally the per-tile filter + mask-NMS of
 * tools/infer_wsi.py:510-531,60-84.
 *
 * Conventions
 *   - every function returns 0 on success or a negative NUHTC_E_* code; nothing throws across the ABI;
 *     nuhtc_last_error() gives the message of the last failure on that engine (or of a failed create).
 *   - plain pointers and sizes only.  `dev` pointers are HIP device pointers owned by the caller
 *     (e.g. torch tensors); `host` pointers are host memory.  The engine owns weights + workspace.
 *   - all work is enqueued on the `stream` passed in (a hipStream_t cast to void*, NULL = default
 *     stream); functions that return counts to the host synchronise that stream, others do not.
 *   - one engine per (device, stream user); an engine is not thread-safe; engines are independent.
 */
#ifndef NUHTC_HIP_H
#define NUHTC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NUHTC_ABI_VERSION 10

enum {
  NUHTC_OK = 0,
  NUHTC_E_INVALID = -1,   /* bad argument / unsupported configuration */
  NUHTC_E_HIP = -2,       /* a HIP runtime call failed */
  NUHTC_E_STATE = -3,     /* call order violated (e.g. infer before finalize, missing weight) */
  NUHTC_E_CAPACITY = -4,  /* a per-tile capacity (proposals) overflowed; results would be truncated */
  NUHTC_E_NOTFOUND = -5   /* unknown weight / buffer name */
};

/* channel handling of the two reference entry points:
 *   0: tools/infer.py      (file -> BGR -> to_rgb): tile channels are used as given (RGB vs RGB means)
 *   1: tools/infer_wsi.py  (RGB ndarray run through the BGR pipeline): channels are reversed first     */
enum { NUHTC_CH_AS_IS = 0, NUHTC_CH_SWAP = 1 };

/* How the fp32 matrix products of the path (Swin linears, convolutions, FCs and -- since ABI v6 -- the two products of window attention)
 * are executed.  Both are fp32 arithmetic: fp32
 * operands and results, exact products, fp32 accumulation.
 *   NUHTC_PIPE_BF16_SPLIT (default): every fp32 operand is split exactly into three bf16 numbers (8 + 8 + 8 significand bits)
 *       and the product runs as six v_mfma_f32_32x32x16_bf16 per 16-deep step.  The six products are exact; the three cross terms
 *       a2*b3, a3*b2 (each <= 2^-24 |a*b|: round-to-nearest splits give |a2| <= 2^-8 |a|, |a3| <= 2^-16 |a|) and a3*b3 (<= 2^-32)
 *       are dropped, i.e. at worst 2^-23 |a*b| per product, one fp32 rounding unit -- not bit-identical to an fp32 fma chain.
 *       Measured error against fp64 is at or below that of the fp32 MFMA chain (csrc/gemm.hip, DESIGN.md 4,
 *       tests/test_hip_dense.py), and parity with the reference is instance-exact with every disagreement explained by a
 *       threshold (tests/parity_util.py), not bitwise.
 *   NUHTC_PIPE_FP32: v_mfma_f32_32x32x2_f32, bitwise an fp32 fma chain (1/16 of the bf16 MFMA rate on gfx950). */
enum { NUHTC_PIPE_BF16_SPLIT = 0, NUHTC_PIPE_FP32 = 1 };

/* What an engine's launch schedule is tuned for (same arithmetic, same results bit for bit):
 *   NUHTC_SCHED_LATENCY (default): one batch at a time as fast as possible -- 128-row block tiles for the Swin linears (the launch
 *       fills the chip soonest), the RPN branch and the big / mid-size RoI classes on the engine's two side streams beside the
 *       caller's stream.
 *   NUHTC_SCHED_THROUGHPUT: for engines that run beside others (nuhtc_amd.pipeline.EnginePipeline sets it) -- 256-row block
 *       tiles (fewer LDS bytes and operand splits per MFMA; another batch's kernels fill the under-filled tail of a launch) and
 *       every kernel of the batch on the caller's stream (the other batches are the concurrency; forks and joins inside a batch
 *       only add cross-stream waits and kernels that compete with their own batch).  Measured with four batches in flight:
 *       +1.5 % for the tiles, +4 % for the single stream; a batch alone is 3-5 % slower this way. */
enum { NUHTC_SCHED_LATENCY = 0, NUHTC_SCHED_THROUGHPUT = 1 };

typedef struct nuhtc_engine nuhtc_engine;

/* Mirrors the `model` / `test_cfg` / `test_pipeline` keys of
 * configs/nuhtc/htc_lite_swin_pytorch_fpn_{PanNuke,CoNSeP,...}_seasaw_CAS.py that the path reads. */
typedef struct nuhtc_config {
  int32_t abi_version;       /* = NUHTC_ABI_VERSION */
  int32_t num_classes;       /* 5 (PanNuke) / 4 (CoNSeP) ... ; 1..14              config:5   */
  int32_t tile_h, tile_w;    /* size of the uint8 input buffers and of the output masks in pixels (256); tile_w % 32 == 0 */
  int32_t valid_h, valid_w;  /* the image inside the buffer (top-left corner); 0 = the whole tile.  Like the reference's test
                              * pipeline the image is resized (img_shape = scale * valid), normalised and zero-padded to the next
                              * multiple of 32 (pad_shape, Pad(size_divisor=32), transforms.py:570-); boxes are clipped to img_shape,
                              * the component proposals are computed at img_shape, masks are pasted into valid_h x valid_w
                              * (ori_shape) and are zero outside it.  scale * valid must be integers. */
  int32_t max_batch;         /* workspace is sized for this many tiles per nuhtc_infer call  */
  float   scale_factor;      /* 2.0 : MultiScaleFlipAug(scale_factor) = 80/mag (tools/infer_wsi.py:416-419); 1..8 */
  float   mean[3], std[3];   /* img_norm_cfg                                     config:8   */
  /* test_cfg.rpn                                                                config:256-261 */
  int32_t rpn_nms_pre;       /* 3000 */
  int32_t rpn_max_per_img;   /* 1000 */
  float   rpn_nms_iou;       /* 0.7  */
  float   rpn_min_bbox_size; /* 10   */
  /* test_cfg.rcnn                                                               config:262-266 */
  float   score_thr;         /* 0.35 */
  float   nms_iou;           /* 0.5  */
  int32_t max_per_img;       /* 500  */
  float   mask_thr_binary;   /* 0.5  */
  /* roi_head                                                                    config:72-160 */
  float   att_thres;         /* 0.965926 : AttentionRoIExtractor thres */
  int32_t watershed_proposal;/* 1 : prepend connected-component proposals (htc_roi_head_cus.py:2217-2221) */
  int32_t max_cc_proposals;  /* capacity for those per tile (the reference has no cap; overflow -> NUHTC_E_CAPACITY) */
  float   stage_stds[3][4];  /* bbox_coder.target_stds of the 3 cascade stages  config:97,115,133 */
  /* tools/infer_wsi.py post-processing (args.margin, args.min_area, mask_nms thr)  infer_wsi.py:510-526 */
  int32_t margin;            /* 2    */
  int32_t min_area;          /* 10   */
  float   mask_nms_thr;      /* 0.05 */
  int32_t matrix_pipe;       /* NUHTC_PIPE_BF16_SPLIT (default) or NUHTC_PIPE_FP32 */
  int32_t schedule;          /* NUHTC_SCHED_LATENCY (default) or NUHTC_SCHED_THROUGHPUT (v5) */
  int32_t att_pool_fp16;     /* 0 (default): the attention-pool branch of AttentionRoIExtractor in fp32, as the reference computes it on a CPU
                              * device (the north star's "fp32 tolerance").  1 (v10): as the reference computes it when its
                              * feature maps are on a CUDA device -- it casts that branch to fp16 there (nuhtc/models/roi_extractors_cus.py:203,231):
                              * every tensor operation of :231-237 rounds to fp16 (reductions accumulate in fp32), the fp16 result is added
                              * into the fp32 RoI features.  Only the level-2 / level-3 tables change; see INTEGRATION.md section 6. */
} nuhtc_config;

/* Fills `cfg` with the PanNuke defaults listed above. */
void nuhtc_default_config(nuhtc_config* cfg);

/* Creates an engine on HIP device `device`.  Replaces `build_detector(cfg.model)` +
 * `.to(device)` of nuhtc/apis/inference.py:11-57. */
int nuhtc_create(const nuhtc_config* cfg, int device, nuhtc_engine** out);
void nuhtc_destroy(nuhtc_engine* e);
const char* nuhtc_last_error(const nuhtc_engine* e);

/* Uploads one tensor of the mmdet state_dict (SURVEY Appendix B naming, fp32, host memory, C order).
 * Replaces `load_checkpoint(model, ckpt)` (nuhtc/apis/inference.py:44).  Names outside the path's schema are rejected with
 * NUHTC_E_NOTFOUND (buffers such as relative_position_index, loss_cls.cum_samples, roi_head.kernel, EMA / optimizer entries
 * are the caller's to drop), a wrong shape or ndim > 4 with NUHTC_E_INVALID; nuhtc_last_error() names the tensor. */
int nuhtc_load_weight(nuhtc_engine* e, const char* name, const float* host_data, const int64_t* shape, int ndim);

/* Checks that every tensor of the path was loaded and pre-packs weights (NHWC / k-major layouts,
 * NormedLinear row normalisation, relative-position bias (nH,49,49), shift masks, window maps). */
int nuhtc_finalize(nuhtc_engine* e);

/* Per-tile results in device memory, capacity `max_per_img` rows per tile (caller-allocated).
 * Row r of tile b lives at index b*max_per_img + r.  Rows are in the reference's NMS order
 * (score descending); `bbox2result` class grouping is done by the host mirror. */
typedef struct nuhtc_dets {
  float*    boxes;      /* dev [B*max_per_img*5]  x1,y1,x2,y2,score in original-tile pixels */
  int32_t*  labels;     /* dev [B*max_per_img] */
  int32_t*  counts;     /* dev [B]  detections per tile */
  uint32_t* masks;      /* dev [B*max_per_img * tile_h * (tile_w/32)] bit-packed rows, bit (x&31) of word x>>5; may be NULL */
  int32_t*  areas;      /* dev [B*max_per_img] mask pixel counts; may be NULL */
  uint8_t*  keep;       /* dev [B*max_per_img] 1 = survives the infer_wsi.py margin/min_area filter + mask-NMS; may be NULL */
} nuhtc_dets;

/* The hot path: B tiles (B <= max_batch) of tile_h x tile_w x 3 uint8 HWC in device memory ->
 * detections.  Replaces `inference_detector(model, [ndarray]*B)` (mmdet/apis/inference.py:90) and,
 * when out->keep != NULL, tools/infer_wsi.py:510-531.  Enqueues on `stream`; does not synchronise. */
int nuhtc_infer(nuhtc_engine* e, const uint8_t* tiles_dev, int B, int channel_mode, void* stream, const nuhtc_dets* out);

/* Fixed-load variant for benchmarking with synthetic weights: the proposal stage is
 * computed but replaced by `rois_dev` (dev [B*n_rois*4] x1,y1,x2,y2 in network pixels), and exactly
 * `n_dets` highest-scoring (roi,class) pairs per tile are carried into the mask branch. */
int nuhtc_infer_fixed_load(nuhtc_engine* e, const uint8_t* tiles_dev, int B, int channel_mode, const float* rois_dev,
                           int n_rois, int n_dets, void* stream, const nuhtc_dets* out);

/* Outer contours of the instance masks of a finished nuhtc_infer, traced on the device so that only vertex lists leave it.
 * Replaces `mask2inst` (tools/infer_wsi.py:51-54: cv2.findContours(mask, RETR_TREE, CHAIN_APPROX_SIMPLE)[0][0]) applied
 * to every detection that survived the per-tile filter + mask-NMS (:533-539).  For detection slot s = b*max_per_img + r
 * with r < dets->counts[b] and (dets->keep == NULL or dets->keep[s]):  n_dev[s] = number of vertices written to
 * xy_dev[s*cap*2 ..] as (x, y) int16 pairs in tile pixels, open ring (the caller repeats the first point and adds the
 * tile origin); n_dev[s] = -1 if the contour has more than `cap` vertices or more than 2048 border pixels (trace that
 * one on the host); n_dev[s] = 0 for slots that are not traced.  dets->masks and dets->counts must be non-NULL.
 * Enqueues on `stream`; does not synchronise. */
int nuhtc_mask_contours(nuhtc_engine* e, const nuhtc_dets* dets, int B, int cap, int16_t* xy_dev, int32_t* n_dev, void* stream);

/* Compacts the kept detections of a finished nuhtc_infer (slots r < counts[b] with keep set), in (tile, slot) order, into
 * dense device buffers of capacity `cap` rows, so that a slide loop fetches a batch's results with a few fixed-size
 * asynchronous copies instead of one copy per detection (the fields tools/infer_wsi.py:486-539 reads out of `result`):
 * n_dev [2]: [0] = number of kept detections (may exceed cap: then only the first cap rows were written and the caller falls
 * back to reading the detection buffers directly), [1] = the capacity flag of that inference (non-zero: nuhtc_check would return
 * NUHTC_E_CAPACITY; v5); idx_dev [cap] = b * max_per_img + r; boxes_dev [cap][5]; labels_dev
 * [cap]; cn_dev [cap] / xy_dev [cap][contour_cap][2] = contour length / vertices from nuhtc_mask_contours (contour_n /
 * contour_xy may be NULL: cn = 0, no vertices); words_dev [cap][tile_h * tile_w / 32] = the bit-packed masks.
 * Enqueues on `stream`; does not synchronise. */
int nuhtc_export_kept(nuhtc_engine* e, const nuhtc_dets* dets, int B, const int32_t* contour_n, const int16_t* contour_xy, int contour_cap,
                      int cap, int32_t* n_dev, int64_t* idx_dev, float* boxes_dev, int32_t* labels_dev, int32_t* cn_dev, int16_t* xy_dev,
                      uint32_t* words_dev, void* stream);

/* Crops the masks nuhtc_export_kept compacted (words_dev [n][tile_h * tile_w / 32], n = min(*n_dev, cap)) to their bounding
 * rectangles, on the device: crop_box_dev [cap][4] = x0, y0, x1, y1 in tile pixels (x1 / y1 exclusive; zeros for an empty mask),
 * crop_area_dev [cap] = set pixels, crop_off_dev [cap + 1] = word offset of each crop in crop_words_dev (entry cap = total words the
 * crops need: when it exceeds pool_cap the crops past the pool were not written and the caller cuts those from words_dev),
 * crop_words_dev = rows of (w + 31) / 32 words with crop column x in bit x & 31 of word x >> 5 -- the layout nuhtc_merge_overlap
 * takes.  Replaces the per-detection numpy slicing of the slide loop (tools/infer_wsi.py:533-566 works on such crops).
 * Enqueues on `stream`; does not synchronise. */
int nuhtc_export_crops(nuhtc_engine* e, const uint32_t* words_dev, const int32_t* n_dev, int cap, int32_t* crop_box_dev, int32_t* crop_area_dev,
                       int32_t* crop_off_dev, uint32_t* crop_words_dev, int pool_cap, void* stream);

/* Overlap measure of nuhtc_merge_overlap. */
enum {
  NUHTC_OVERLAP_MASK = 0,     /* IoU of the instance masks (pixel sets) */
  NUHTC_OVERLAP_POLYGON = 1   /* the reference's: IoU of the shapely polygons of the rings tools/infer_wsi.py writes (first contour of
                                 cv2.findContours through the border-pixel centres; buffer(0) + largest part when the ring touches
                                 itself, tools/nuclei_merge.py:37-59), computed exactly from the mask crops (csrc/merge.hip) */
};

/* Cross-tile duplicate removal over all detections of a slide: tools/nuclei_merge.py:62-174 `merge_overlap`, strategy
 * 'probability' (visit in descending score, ties by lower index; an alive detection removes every later one whose overlap
 * with it exceeds `thr`, compared as a double division like the reference's `inter.area / (a.area + b.area - inter.area)`).
 * All pointers are device memory of `device`:  boxes [n][4] int32 = x0,y0,x1,y1 (x1,y1 exclusive) of each mask crop in
 * slide pixels (the bounding box of the mask's set pixels); scores [n]; areas [n] = set pixels (NUHTC_OVERLAP_MASK only, may
 * be NULL otherwise); bits = the crops, bit-packed row by row, (x1-x0+31)/32 uint32 words per row, pixel x of a row in bit
 * (x&31) of word x>>5; bit_off [n] = word offset of each crop in `bits`; n_words = total words of `bits`.
 * x_min..y_max bound all boxes (the slide extent).  keep_dev [n] receives 1 for kept detections.  Allocates its own scratch
 * (about 120 bytes per detection, plus a copy of `bits` in polygon mode), runs on `stream` and synchronises it before
 * returning.  A detection may have any number of higher-scored overlapping neighbours (24 are kept inline, the rest in a
 * spill pool that is grown and the pass repeated when it runs out); NUHTC_E_CAPACITY only if 2^30 spill entries do not
 * suffice.  NUHTC_E_INVALID in polygon mode if a crop exceeds about 430x430 pixels. */
int nuhtc_merge_overlap(int device, const int32_t* boxes, const float* scores, const int32_t* areas, const uint32_t* bits,
                        const int64_t* bit_off, int64_t n, int64_t n_words, int overlap, double thr, int x_min, int y_min,
                        int x_max, int y_max, uint8_t* keep_dev, void* stream);

/* Synchronises `stream` and reports whether the last nuhtc_infer overflowed max_cc_proposals on some tile (returns
 * NUHTC_E_CAPACITY) — call before trusting the results.  That is the only capacity of the path that is not the reference's
 * own cap: RPN candidates, RoIs and detection candidates are sized for their worst case (nms_pre per level,
 * max_cc_proposals + rpn_max_per_img, RoIs x classes), detections are cut at max_per_img as the reference cuts them. */
int nuhtc_check(nuhtc_engine* e, void* stream);

/* Parity-test access to intermediate tensors of the last nuhtc_infer call (device pointers into the
 * engine workspace; valid until the next call).  Names: "img", "c0".."c3", "x0".."x3", "rpn0".."rpn3",
 * "sem_pred", "sem_feat", "rpn_props", "rpn_counts", "cc_mask", "cc_props", "cc_counts", "rois", "roi_counts",
 * "cls0".."cls2", "reg0".."reg2", "bbox_feats", "mask_prob", "tokens<stage><block>" ...
 * shape receives up to 6 dims; *dtype: 0=f32, 1=i32, 2=u8, 3=u32.
 * Two of them are not written by the step any more (either matrix pipe) and are computed by this call, synchronising the device: "img" (the
 * pre-processing runs inside the patch embedding) and "c0".."c3" (the stages' output norms run inside the FPN laterals; computed from the
 * stages' token buffers).  LIFETIME of "img": the engine keeps only the POINTER of the last nuhtc_infer's `tiles` and reads it again here --
 * the caller must keep that device buffer alive and unchanged until it has fetched "img" (or never ask for it); a step that fails drops
 * the pointer ("img" then returns whatever the buffer held). */
int nuhtc_get_buffer(nuhtc_engine* e, const char* name, void** dev_ptr, int64_t* shape, int* ndim, int* dtype);

/* Stand-alone ops for kernel-level parity tests (all pointers device memory, fp32). */
/* C[M,N] = act(A[M,K] * W[N,K]^T + bias[N]);  act: 0 none, 1 relu, 2 gelu(erf).  K%32==0, N%32==0. */
int nuhtc_op_gemm(nuhtc_engine* e, const float* A, const float* W, const float* bias, float* C, int M, int N, int K,
                  int act, void* stream);
/* The same product on the bf16 matrix pipe with exactly split operands (NUHTC_PIPE_BF16_SPLIT); W_host = host copy of W_dev (the
 * split of a constant weight is made on the host, as at nuhtc_finalize).  Synchronises `stream`. */
int nuhtc_op_gemm_split(nuhtc_engine* e, const float* A, const float* W_dev, const float* W_host, const float* bias, float* C, int M,
                        int N, int K, int act, void* stream);
/* ABI v8 (round 5).  A linear behind a LayerNorm with the norm in the product's A path (csrc/gemm.hip A_LN; what the QKV and fc1 linears of
 * Swin stages 2-4 run: mmdet swin.py:358,365): C[M,N] = act(LN(X[rows[m]])[M,K] * W[N,K]^T + bias[N]) with LN = LayerNorm(K, eps 1e-5,
 * ln_g, ln_b).  W, bias, ln_g, ln_b are HOST arrays (the norm's affine part is folded into the linear on the host exactly as nuhtc_finalize
 * folds it: W' = W diag(ln_g) rounded once to fp32 and split exactly, bias' = bias + W ln_b in fp64); X [T][K] and `rows` (device int32 [M]
 * indices into X, or NULL for the identity with M <= T) are device memory.  The row statistics come from ln_stats_kernel (csrc/swin.hip).
 * N % 96 == 0, K % 32 == 0.  Synchronises `stream`. */
int nuhtc_op_ln_gemm(nuhtc_engine* e, const float* X_dev, int T, const int* rows_dev, const float* W_host, const float* bias_host, const float* ln_g_host,
                     const float* ln_b_host, float* C_dev, int M, int N, int K, int act, void* stream);
/* The same linear fed the way the engine feeds it: a producer product Y[row_map(m)] = A[M,Kp] * Wp[K,Kp]^T + bp (+ res[row_map(m)]) whose
 * epilogue leaves, per row and 96 columns, {mean, sum of squared deviations} of what it stored (GemmParams.stats_out), and the A_LN linear
 * C = act(LN(Y) * W^T + bias) that merges those partials -- no pass over Y computes statistics.  Y_dev [M][K] and C_dev [M][N] are outputs;
 * row_map_dev (device int32 [M], a permutation of 0..M-1) and res_dev ([M][K]) may be NULL.  K % 96 == 0, N % 96 == 0.  Synchronises `stream`. */
int nuhtc_op_gemm_ln_gemm(nuhtc_engine* e, const float* A_dev, const float* Wp_host, const float* bp_host, const float* res_dev, const int* row_map_dev,
                          const float* W_host, const float* bias_host, const float* ln_g_host, const float* ln_b_host, float* Y_dev, float* C_dev, int M,
                          int Kp, int K, int N, int act, void* stream);
/* ABI v9 (round 5).  PatchMerging (mmdet/models/utils/transformer.py:363-385: nn.Unfold(2, stride 2), LayerNorm(4C), Linear(4C -> 2C, no bias)) as
 * the engine runs it on the split pipe: ONE product whose A rows are gathered from the token tensor as two runs of 2C floats (the 2 x 2 tokens of
 * a merged row), the norm in the A path (csrc/gemm.hip A_LN with seg_k) and the row statistics merged from per-token partials over 96 channels.
 * X_dev [B*H*W][C] tokens; W_host [2C][4C], ln_g_host / ln_b_host [4C] in the REFERENCE's column order k = c*4 + kh*2 + kw (re-ordered here exactly as
 * nuhtc_finalize re-orders them); Y_dev [B*(H/2)*(W/2)][2C].  H, W even, C % 96 == 0.  Synchronises `stream`. */
int nuhtc_op_merge_ln_gemm(nuhtc_engine* e, const float* X_dev, int B, int H, int W, int C, const float* W_host, const float* ln_g_host,
                           const float* ln_b_host, float* Y_dev, void* stream);
/* The fused FFN half of a Swin block (csrc/mlp.hip; mmdet swin.py:365-367): out[T,C] = x + W2 gelu(W1 LN(x) + b1) + b2 with
 * LN = LayerNorm(C, eps 1e-5, ln_g, ln_b), W1 [4C][C], W2 [C][4C] given as HOST arrays (packed like nuhtc_finalize packs them),
 * everything else device memory.  C must be a width the fused kernel serves (96).  Synchronises `stream`. */
int nuhtc_op_swin_mlp(nuhtc_engine* e, const float* x_dev, const float* ln_g_dev, const float* ln_b_dev, const float* w1_host,
                      const float* b1_dev, const float* w2_host, const float* b2_dev, float* out_dev, int T, int C, void* stream);
/* ABI v6 (round 4).  The same kernel with the attention projection in front (mmdet swin.py:360-367, the second half of a Swin block from the
 * attention output on):  x' = x + Wp att + bp;  out = x' + W2 gelu(W1 LN(x') + b1) + b2.  att [T,C] in token order, Wp [C][C] given as a HOST
 * array, everything else as in nuhtc_op_swin_mlp.  Synchronises `stream`. */
int nuhtc_op_swin_proj_mlp(nuhtc_engine* e, const float* x_dev, const float* att_dev, const float* wp_host, const float* bp_dev,
                           const float* ln_g_dev, const float* ln_b_dev, const float* w1_host, const float* b1_dev, const float* w2_host,
                           const float* b2_dev, float* out_dev, int T, int C, void* stream);
/* mmcv RoIAlign(avg, aligned=True) on an NHWC map: feat [N,H,W,C=64], rois [R,5] -> out [R,P,P,C]. */
int nuhtc_op_roi_align(nuhtc_engine* e, const float* feat_nhwc, int N, int H, int W, const float* rois, int R, int P,
                       float spatial_scale, int sampling_ratio, float* out, void* stream);
/* mmcv nms on n boxes (n <= 16384): keep_idx[0..*count) in descending-score order (ties: lower index). */
int nuhtc_op_nms(nuhtc_engine* e, const float* boxes, const float* scores, int n, float iou_thr, int32_t* keep_idx,
                 int32_t* count_dev, void* stream);

/* A HIP stream owned by the engine (valid after nuhtc_finalize) that a caller MAY run this engine on,
 * and should when it keeps several engines busy at once or raises GPU_MAX_HW_QUEUES above the runtime's default of 4: the stream
 * is created next to the engine's two internal side streams, which puts the three on different pipes of the command processor
 * (batches in flight).  Any other stream remains valid for every entry point.
 * LIFETIME: the engine's streams are POOLED for the life of the process, not destroyed: nuhtc_destroy hands the (own, side, side2)
 * triple back to a per-device pool and the next engine created on that device reuses the same handles (PyTorch's allocator touches a
 * block's allocation stream when it frees the block, long after the engine is gone).  A caller must therefore NOT use, wait on or
 * record into the handle after nuhtc_destroy: it may already belong to an unrelated engine. */
void* nuhtc_stream(nuhtc_engine* e);

/* Host-thread placement (v7; v8: original-mask bookkeeping, restore, explicit-root test entry).  Restricts the CALLING thread (threads
 * it creates later inherit the mask) to the CPUs of the NUMA node the device is attached to (/sys/bus/pci/devices/<bdf>/local_cpulist),
 * intersected with the mask the thread had BEFORE its first placement (kept per thread: a thread placed for a GPU of one socket can be
 * placed again for a GPU of the other).  The thread that submits an engine's work should run there: the command processor reads every
 * dispatch packet from host memory last written by the submitter, and from the other socket of a two-socket host that costs 1.4-2.9 us
 * per packet -- 0.3-0.4 ms per step of the back-to-back dense launches.  Returns 0 (bound, or already inside the
 * node), NUHTC_E_NOTFOUND when the host exposes no NUMA node for the device (nothing changed), NUHTC_E_STATE when the caller's own mask
 * has no CPU of that node (nothing changed: the caller chose otherwise), NUHTC_E_INVALID for a string that is not a PCI address (hex
 * digits, ':' and '.'), NUHTC_E_HIP for a bad device.  nuhtc_restore_host_thread gives the calling thread the mask it had before its first
 * placement (0 also when it was never placed).  No counterpart in the reference: its launcher (tools/test.py:100-103,179-183 -> mmcv
 * init_dist) leaves the placement of a rank to the operating system.  NEVER called implicitly by the library, and since v8 not by the
 * Python host either unless asked (`init_detector(..., bind_host=True)`, NUHTC_HOST_AFFINITY=1, or the entry points that own their
 * process: bench.py, tools/infer_wsi.py, tools/bench_wsi.py).  _pci takes the PCI address ("0000:75:00.0") instead of a device index;
 * _at is the test entry point: the same code against a sysfs tree under `sysfs_root` (the NUHTC_SYSFS_ROOT variable of v7 is gone). */
int nuhtc_bind_host_thread(int device);
int nuhtc_bind_host_thread_pci(const char* pci_bdf);
int nuhtc_bind_host_thread_at(const char* sysfs_root, const char* pci_bdf);
int nuhtc_restore_host_thread(void);

/* Text of the QuPath documents of a slide (v10).  The reference builds one dict per nucleus (tools/infer_wsi.py:550-585) and json.dump()s
 * the lists (:659-664); these write the same bytes from arrays.  Host memory only, no device work, callable from any thread.
 * head / mid / tail are NUL-terminated pieces of the feature template per class (the host cuts them out of json.dumps of one template
 * feature, so key order, separators and the classification block are json's own); numbers are written the way json.dumps writes a Python
 * int / float (float.__repr__: shortest round-trip digits, exponent form below 1e-4 and from 1e16 on).  Records are separated by ", ";
 * the enclosing brackets are the caller's.
 *   nuhtc_write_ring_features : record i = head | "[x, y], [x, y], ..." of ring i (verts[ring_off[i] .. ring_off[i+1]), int32 pairs) |
 *       mid[label[i]] | repr(score[i]) | tail[label[i]].  Fills feat_start[0..n]: record i is out[feat_start[i] .. feat_start[i+1] - 2).
 *       `threads` host threads share the records (0: the hardware's, at most 16).
 *   nuhtc_write_point_features: record i = head | repr(xy[2i]) ", " repr(xy[2i+1]) | mid[label[i]] | repr(score[i]) | tail[label[i]].
 *   nuhtc_join_features       : the records pick[0..n_pick) of a text written by nuhtc_write_ring_features (same feat_start convention),
 *       joined by ", " (the merged document: the records the cross-tile merge kept).
 * Each returns the number of bytes written.  When `out` is NULL or `cap` is below what the text needs, nothing is written and the
 * needed size is returned (exact for the ring and join writers, an upper bound for the point writer): call once to size.
 * Negative: NUHTC_E_INVALID (null argument, label outside [0, n_labels), n_labels > 64, decreasing offsets). */
int64_t nuhtc_write_ring_features(const int32_t* verts, const int64_t* ring_off, const int32_t* label, const double* score, int64_t n,
                                  const char* head, const char* const* mid, const char* const* tail, int32_t n_labels,
                                  char* out, int64_t cap, int64_t* feat_start, int32_t threads);
int64_t nuhtc_write_point_features(const double* xy, const int32_t* label, const double* score, int64_t n,
                                   const char* head, const char* const* mid, const char* const* tail, int32_t n_labels,
                                   char* out, int64_t cap);
int64_t nuhtc_join_features(const char* text, const int64_t* feat_start, const int64_t* pick, int64_t n_pick, char* out, int64_t cap,
                            int32_t threads);

/* Rings of a written GeoJSON back into mask crops (v10; tools/nuclei_merge.py on the GPU).  A ring tools/infer_wsi.py writes is the traced outer
 * border of one 8-connected pixel component (`cv2.findContours(...)[0][0]`, :51-58): vertices on pixel centres, edges along the 8 chain directions.
 * The pixels inside or on it are that component with its holes filled -- what nuhtc_merge_overlap derives from a detection's mask crop before it
 * measures polygons -- so the filled rings are a valid input of nuhtc_merge_overlap.  verts: int32 pairs, ring i = verts[ring_off[i] .. ring_off[i+1])
 * WITHOUT the repeated closing vertex.  Fills boxes[n][4] (x0, y0, x1, y1 exclusive), areas[n] (set pixels), word_off[n] and the bit-packed crops
 * (rows of (w + 31) / 32 words, pixel x in bit x & 31 of word x >> 5).  Returns the number of words; with `bits` NULL or `cap_words` too small only
 * boxes and word_off are filled and the needed size is returned.  NUHTC_E_INVALID: an empty ring, an edge that is not horizontal, vertical or diagonal
 * (not a traced ring: use the polygon path of the host), a ring wider or taller than 65535 pixels.  Host memory, any thread. */
int64_t nuhtc_fill_rings(const int32_t* verts, const int64_t* ring_off, int64_t n, int32_t* boxes, int32_t* areas, int64_t* word_off,
                         uint32_t* bits, int64_t cap_words, int32_t threads);

/* Per-kernel timing with HIP events recorded on the launch stream (process-wide switch; off by default).
 * nuhtc_profile_read synchronises the device and writes one text line per kernel tag,
 * "tag launches total_ms algorithmic_flops algorithmic_bytes", then resets the records. */
int nuhtc_profile_enable(int on);
int nuhtc_profile_read(char* buf, size_t cap);
/* Shader clock under load (measurement): enqueues on `stream` a one-wave kernel that spins for `ticks_100mhz` periods of the 100 MHz
 * reference clock and writes out_dev[0] = shader cycles elapsed, out_dev[1] = reference ticks elapsed.  Launched on a stream of
 * its own beside the kernels being timed, out[0] / out[1] x 100 MHz is the clock the chip held under them.  Does not synchronise. */
int nuhtc_clock_probe(int device, uint64_t ticks_100mhz, uint64_t* out_dev, void* stream);
/* Development builds (-DNUHTC_DEV) only: set an integer switch of the launch heuristics (the NUHTC_<NAME> environment variables) at
 * run time.  The default build compiles every switch to its default and returns NUHTC_E_STATE here. */
int nuhtc_dev_knob(const char* name, int value);

#ifdef __cplusplus
}
#endif
#endif /* NUHTC_HIP_H */
