// Parameter blocks of the RoI-path kernels (roi.hip).
#pragma once
#include "common.h"

struct RoiFeatParams {
  const float* rois;    // [R][5] b,x1,y1,x2,y2 (network pixels)
  const int* r_dev;     // device-side R
  const float *x0, *x1; // FPN levels 0,1 NHWC (stride 4, 8)
  const float *G2, *G3; // attention-pool tables [B][H*W][64] of levels 2,3
  const float* sem;     // semantic embedding NHWC at stride 4
  const float* x0sem;   // x0 + sem (P = 7 LDS path: both are sampled at the same points, so one interpolation serves both)
  int H0, W0, H1, W1, H2, W2, H3, W3;
  float* out;           // [R][P*P][64]
  int* fb_count;        // [8] P=7: [0] big RoIs (row-streaming workgroup each), [1] mid-size RoIs (stream kernel), [2] giant RoIs (the rest fit the LDS tiles); [3], [4] job counters of the stream / big-box kernels
  int* fb_list;         // [list_cap] big RoIs from the front, giant RoIs from the back
  int list_cap;
  int* mid_list;        // [R] mid-size RoIs
  float* big_part;      // optional [big_split_max][3 maps][49][64]: per-map partial features of the big boxes while they are few (one workgroup per map)
  int big_split_max;
  int stream_few;       // short mid-size lists go to roi_feat7_stream_few_kernel (several loading waves per RoI, same sums)
  unsigned char* fb_flag; // [R] 0 / 3 = LDS tiles (small / large), 1 = stream kernel, 2 = big-box kernel, 4 = giant (sample loop)
};

struct BboxTailParams {
  const float* h;       // [R][256] (after the two shared FCs)
  const float* w;       // [nc+6][256]: rows 0..nc+1 = row-normalised fc_cls, then 4 fc_reg rows
  const float* b;       // [nc+6]
  int nc;
  const int* r_dev;
  float* cls;           // [R][16]
  float* reg;           // [R][4]
  int refine;           // 1: regress_by_class in place on rois
  float* rois;          // [R][5]
  float stds[4];
  float img_w, img_h;
};

struct DetCandParams {
  const float* rois;                  // [R][5] after two refinements
  const float *cls0, *cls1, *cls2;    // [R][16]
  const float* reg2;                  // [R][4]
  const int *roi_off, *roi_cnt;       // [B]
  int nc;
  float stds[4];
  float img_w, img_h, scale, score_thr;
  float* cand_boxes;                  // [B][cap][4]
  float* cand_scores;                 // [B][cap]
  int* cand_ids;                      // [B][cap]
  int* cand_count;                    // [B]
  int cap;
};

struct DetFinishParams {
  int B, max_keep;
  const float* dets;       // [B][max_keep][5]
  const int* keep_src;     // [B][max_keep] flat candidate index
  const int* cand_ids;     // flat [B*cap]
  int* det_counts;         // [B] (clamped to `limit` in place)
  int limit;
  int* labels;             // [B][max_keep]
  float* mask_rois;        // [D][5]
  int* det_off;            // [B]
  int* det_total;          // scalar
  float scale;
};

struct PasteParams {
  const float* prob;       // [D][28*28]
  const float* mask_rois;  // [D][5]
  const int *det_off, *det_counts;
  int max_keep, H, W;      // mask buffer rows / columns (W % 32 == 0)
  int vH, vW;              // ori_shape: the canvas get_seg_masks pastes into (<= H, W); bits outside stay 0
  float scale, thr;
  unsigned* masks;         // [B][max_keep][H][W/32]
  int* areas;              // [B][max_keep]
};

struct TilePostParams {
  const float* dets;
  const int* labels;
  const int* areas;
  const int* det_counts;
  const unsigned* masks;
  unsigned char* keep;
  int max_keep, H, W, margin, min_area;
  int vH, vW;              // image size the margin filter refers to (<= H, W)
  double thr;
};

int launch_attn_pool(const float* F, float* G, int B, int HW, float tau, hipStream_t s);
int launch_attn_pool_fp16(const float* F, float* G, int B, int HW, float tau, hipStream_t s);   // the reference-on-CUDA arithmetic (nuhtc_config.att_pool_fp16)
int launch_build_rois(const float* cc_boxes, const int* cc_counts, int cc_cap, const float* rpn_dets, const int* rpn_counts, int rpn_cap,
                      const float* fixed, int n_fixed, float* rois, int* roi_off, int* roi_cnt, int* total, int B, hipStream_t s);
// `side` / `ev_fork` / `ev_join` (optional): the mid-size-RoI gather kernel of the P = 7 path runs on `side` beside the LDS-path
// kernel and is joined back into `s` before returning
int launch_roi_feat(const RoiFeatParams& p, int P, int r_cap, hipStream_t s, hipStream_t side = nullptr, hipEvent_t ev_fork = nullptr,
                    hipEvent_t ev_join = nullptr, hipStream_t side2 = nullptr, hipEvent_t ev_join2 = nullptr);
int launch_bbox_tail(const BboxTailParams& p, int r_cap, hipStream_t s);
int launch_det_candidates(const DetCandParams& p, int B, hipStream_t s);
int launch_det_finish(const DetFinishParams& p, hipStream_t s);
int launch_paste(const PasteParams& p, int B, hipStream_t s);
int launch_tile_post(const TilePostParams& p, int B, hipStream_t s);
