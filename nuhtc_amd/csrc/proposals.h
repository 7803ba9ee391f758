// Parameter blocks of the proposal / NMS kernels (proposals.hip).
#pragma once
#include "common.h"

#define NMS_MAX_CAP 16384     // candidates per image the sort + mask matrix can take (128 KiB of LDS keys)
#define NMS_MAX_GROUPS 16
#define CC_LIST_CAP 4096      // components per tile that can pass the area filter before the ordered emit (opened masks: <= ~2600)

struct RpnLevels {
  const float* out[4];   // [B, h*w, 32]: cols 0-2 objectness logits, 3.. deltas (anchor-major)
  int h[4], w[4], stride[4];
};

struct RpnSelParams {
  int nms_pre;
  int slot;             // per (image, level) capacity of the candidate slots (>= nms_pre)
  float* cand_boxes;    // [B][4][slot][4]
  float* cand_scores;   // [B][4][slot]
  int* cand_count;      // [B][4]
  int img_h, img_w;
  float min_size;
  unsigned* keys;       // [B][4][key_stride] scratch: the sortable score keys of one (image, level), written once and re-read by the select passes
  int key_stride;       // >= anchors of the largest level
};

// Generic batched NMS over `n_groups` slots per image.
struct NmsParams {
  const float* boxes;     // [B][n_groups][slot][4]
  const float* scores;    // [B][n_groups][slot]
  const int* ids;         // optional per-candidate id (class); null -> id = group index (RPN level)
  const int* group_count; // [B][n_groups]
  int n_groups, slot;
  int cap;                // >= n_groups*slot rounded up to 64; leading dimension of the sorted arrays / mask rows
  int cap_pow2;           // power of two >= max candidates (LDS sort image)
  float iou_thr;
  int max_keep;
  float* sorted_boxes;    // [B][cap][4] (offset boxes in sorted order)
  int* sorted_src;        // [B][cap] flat source index into boxes/scores
  int* n_total;           // [B]
  unsigned long long* mask;  // [B][cap][cap/64]
  float* out_dets;        // [B][max_keep][5]
  int* out_src;           // [B][max_keep]
  int* out_counts;        // [B]
  // level-wise variant (launch_nms_levels): groups never suppress each other, so each runs its own NMS
  int* seg_start;         // [B][n_groups] first row of the group's 64-aligned segment in the sorted arrays
  int* seg_n;             // [B][n_groups] candidates of the group
  int* sorted_pos;        // [B][cap] group-major position of the sorted row (tie-break of the reference's global sort)
  unsigned long long* keepbits;   // [B][cap/64] survivors per 64-row chunk
};

struct CcParams {
  const float* sem_pred;  // [B][h][w]
  int h, w, img_h, img_w;
  int min_area, cap;
  unsigned char *mask_a, *mask_b, *touch;   // [B][img_h*img_w]
  int* labels;            // [B][img_h*img_w]
  int* stats;             // [B][img_h*img_w][5]
  int* list;              // [B][CC_LIST_CAP] roots passing the area filter (unordered)
  int* nlist;             // [B]
  float* boxes;           // [B][cap][4]
  int* counts;            // [B]
  int* overflow;          // int[4]
};

int launch_rpn_select(const RpnLevels& lv, const RpnSelParams& p, int B, hipStream_t s);
int launch_nms(const NmsParams& p, int B, hipStream_t s);
// same result as launch_nms for ids == null (id = group): per-group NMS in parallel, then the survivors merged by score;
// needs cap >= sum(round_up(group counts, 64)) and n_groups * max_keep <= 8192
int launch_nms_levels(const NmsParams& p, int B, hipStream_t s);
int nms_set_attributes();
int launch_cc_proposals(const CcParams& p, int B, hipStream_t s);
