#include "engine.h"
int finalize_roi(nuhtc_engine* e) { return 0; }
int alloc_roi_workspace(nuhtc_engine* e) { return 0; }
int run_roi_path(nuhtc_engine* e, int B, const float* rois_fixed, int n_rois, int n_dets, hipStream_t s, const nuhtc_dets* out) { return 0; }
int nuhtc_op_roi_align(nuhtc_engine* e, const float* feat_nhwc, int N, int H, int W, const float* rois, int R, int P,
                       float spatial_scale, int sampling_ratio, float* out, void* stream) { return NUHTC_E_INVALID; }
int nuhtc_op_nms(nuhtc_engine* e, const float* boxes, const float* scores, int n, float iou_thr, int32_t* keep_idx,
                 int32_t* count_dev, void* stream) { return NUHTC_E_INVALID; }
