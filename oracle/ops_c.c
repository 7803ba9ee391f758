/* ORACLE (test infrastructure, not product code).
 *
 * Plain-C restatement of the two native ops the reference path takes from mmcv-full==1.7.2
 * (un-vendored; not under /root/reference), float32 like mmcv's `float` instantiation:
 *   roi_align_forward  - mmcv ops.RoIAlign forward, pool_mode='avg', aligned=True
 *       call sites: mmdet/models/roi_heads/roi_extractors/base_roi_extractor.py:53-58,
 *                   nuhtc/models/roi_extractors_cus.py:198,218
 *   nms_f32            - mmcv ops.nms, offset 0, strict '>' suppression
 *       call sites: mmdet/models/dense_heads/rpn_head.py:232, nuhtc/models/bbox_head.py:93
 * PARITY UNPINNED by the reference (no golden numbers for these ops exist in its tests);
 * pinned by hand-derived known answers in tests/test_oracle_ops.py and cross-checked against
 * the independent numpy restatement oracle/ops_np.py.
 *
 * build: gcc -O2 -fPIC -shared -ffp-contract=off oracle/ops_c.c -o oracle/libnuhtc_oracle.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static float bilinear(const float* f, int H, int W, float y, float x) {
  if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return 0.0f;
  if (y <= 0) y = 0;
  if (x <= 0) x = 0;
  int yl = (int)y, xl = (int)x, yh, xh;
  if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
  if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
  float ly = y - yl, lx = x - xl, hy = 1.0f - ly, hx = 1.0f - lx;
  float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
  return w1 * f[yl * W + xl] + w2 * f[yl * W + xh] + w3 * f[yh * W + xl] + w4 * f[yh * W + xh];
}

/* feat (N,C,H,W), rois (R,5) [b,x1,y1,x2,y2], out (R,C,P,P) */
void roi_align_forward(const float* feat, int N, int C, int H, int W, const float* rois, int R, int P,
                       float scale, int sampling_ratio, float* out) {
  (void)N;
#pragma omp parallel for schedule(dynamic, 8)
  for (int r = 0; r < R; ++r) {
    const float* roi = rois + 5 * r;
    int b = (int)roi[0];
    float x1 = roi[1] * scale - 0.5f, y1 = roi[2] * scale - 0.5f;
    float x2 = roi[3] * scale - 0.5f, y2 = roi[4] * scale - 0.5f;
    float rw = x2 - x1, rh = y2 - y1;
    float bh = rh / (float)P, bw = rw / (float)P;
    int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)P);
    int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)P);
    float count = (float)(gh * gw > 1 ? gh * gw : 1);
    for (int c = 0; c < C; ++c) {
      const float* f = feat + ((size_t)b * C + c) * H * W;
      for (int ph = 0; ph < P; ++ph)
        for (int pw = 0; pw < P; ++pw) {
          float acc = 0.0f;
          for (int iy = 0; iy < gh; ++iy) {
            float y = y1 + ph * bh + (iy + 0.5f) * bh / (float)gh;
            for (int ix = 0; ix < gw; ++ix) {
              float x = x1 + pw * bw + (ix + 0.5f) * bw / (float)gw;
              acc += bilinear(f, H, W, y, x);
            }
          }
          out[(((size_t)r * C + c) * P + ph) * P + pw] = acc / count;
        }
    }
  }
}

typedef struct { float s; int i; } si_t;
static int cmp_desc(const void* a, const void* b) {
  const si_t *x = a, *y = b;
  if (x->s > y->s) return -1;
  if (x->s < y->s) return 1;
  return x->i - y->i; /* ties: lower index first (deterministic rule shared with the HIP engine) */
}

/* boxes (n,4), scores (n) -> keep[] (indices, descending score); returns count */
int nms_f32(const float* boxes, const float* scores, int n, float thr, int64_t* keep) {
  si_t* ord = malloc(sizeof(si_t) * (n > 0 ? n : 1));
  unsigned char* sup = calloc(n > 0 ? n : 1, 1);
  for (int i = 0; i < n; ++i) { ord[i].s = scores[i]; ord[i].i = i; }
  qsort(ord, n, sizeof(si_t), cmp_desc);
  int k = 0;
  for (int a = 0; a < n; ++a) {
    if (sup[a]) continue;
    int i = ord[a].i;
    keep[k++] = i;
    float ix1 = boxes[4 * i], iy1 = boxes[4 * i + 1], ix2 = boxes[4 * i + 2], iy2 = boxes[4 * i + 3];
    float iarea = (ix2 - ix1) * (iy2 - iy1);
    for (int c = a + 1; c < n; ++c) {
      if (sup[c]) continue;
      int j = ord[c].i;
      float xx1 = fmaxf(ix1, boxes[4 * j]), yy1 = fmaxf(iy1, boxes[4 * j + 1]);
      float xx2 = fminf(ix2, boxes[4 * j + 2]), yy2 = fminf(iy2, boxes[4 * j + 3]);
      float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
      float inter = w * h;
      float area = (boxes[4 * j + 2] - boxes[4 * j]) * (boxes[4 * j + 3] - boxes[4 * j + 1]);
      float ovr = inter / (iarea + area - inter);
      if (ovr > thr) sup[c] = 1;
    }
  }
  free(ord); free(sup);
  return k;
}
