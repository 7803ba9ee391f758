// libnuhtc_hip.so — C ABI (include/nuhtc_hip.h), weight packing, workspace and the launch sequence of the
// htc_lite_swin tile-inference path (reference call stack: SURVEY §3.3; nuhtc/models/htc_cus.py:110-121,
// nuhtc/models/htc_roi_head_cus.py:2184-2372).  Host code only enqueues kernels; there is no CPU fallback.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "common.h"
#include "engine.h"

// =============================================================================== small helpers
static const int DEPTHS[4] = {2, 2, 6, 2};
static const int NHEADS[4] = {3, 6, 12, 24};

static thread_local std::string g_create_error;

void nuhtc_default_config(nuhtc_config* c) {
  memset(c, 0, sizeof(*c));
  c->abi_version = NUHTC_ABI_VERSION;
  c->num_classes = 5;
  c->tile_h = c->tile_w = 256;
  c->max_batch = 16;
  c->scale_factor = 2.0f;
  const float mean[3] = {123.675f, 116.28f, 103.53f}, std[3] = {58.395f, 57.12f, 57.375f};
  for (int i = 0; i < 3; ++i) { c->mean[i] = mean[i]; c->std[i] = std[i]; }
  c->rpn_nms_pre = 3000; c->rpn_max_per_img = 1000; c->rpn_nms_iou = 0.7f; c->rpn_min_bbox_size = 10.f;
  c->score_thr = 0.35f; c->nms_iou = 0.5f; c->max_per_img = 500; c->mask_thr_binary = 0.5f;
  c->att_thres = 0.965926f;
  c->watershed_proposal = 1;
  c->max_cc_proposals = 512;
  const float st[3][4] = {{0.1f, 0.1f, 0.2f, 0.2f}, {0.05f, 0.05f, 0.1f, 0.1f}, {0.033f, 0.033f, 0.067f, 0.067f}};
  memcpy(c->stage_stds, st, sizeof(st));
  c->margin = 2; c->min_area = 10; c->mask_nms_thr = 0.05f;
  c->matrix_pipe = NUHTC_PIPE_BF16_SPLIT;
  c->schedule = NUHTC_SCHED_LATENCY;
  c->att_pool_fp16 = 0;
}

const char* nuhtc_last_error(const nuhtc_engine* e) { return e ? e->err.c_str() : g_create_error.c_str(); }

int nuhtc_create(const nuhtc_config* cfg, int device, nuhtc_engine** out) {
  if (!cfg || !out) { g_create_error = "null argument"; return NUHTC_E_INVALID; }
  if (cfg->abi_version != NUHTC_ABI_VERSION) { g_create_error = "abi_version mismatch"; return NUHTC_E_INVALID; }
  if (cfg->tile_h <= 0 || cfg->tile_w <= 0 || cfg->tile_w % 32) { g_create_error = "tile_h / tile_w must be positive and tile_w a multiple of 32 (bit-packed mask rows)"; return NUHTC_E_INVALID; }
  if (cfg->valid_h < 0 || cfg->valid_w < 0 || cfg->valid_h > cfg->tile_h || cfg->valid_w > cfg->tile_w) { g_create_error = "valid_h / valid_w must lie in [0, tile] (0 = the whole tile)"; return NUHTC_E_INVALID; }
  {
    // resized image = mmcv.rescale_size: int(size*scale + 0.5); the per-axis factors new/old (mmdet Resize: w_scale, h_scale) must
    // both equal scale_factor, i.e. scale*size is an integer.  Pad(size_divisor=32) then rounds the network input up.
    const int vh = cfg->valid_h ? cfg->valid_h : cfg->tile_h, vw = cfg->valid_w ? cfg->valid_w : cfg->tile_w;
    const double sh = (double)vh * cfg->scale_factor, sw = (double)vw * cfg->scale_factor;
    if (!(cfg->scale_factor >= 1.0f && cfg->scale_factor <= 8.0f) || sh != floor(sh) || sw != floor(sw)) {
      g_create_error = "scale_factor (80/mag) must be in [1,8] and scale_factor * image size must be integers";
      return NUHTC_E_INVALID;
    }
    if (sh < 32 || sw < 32) { g_create_error = "the resized image must be at least 32 x 32"; return NUHTC_E_INVALID; }
  }
  // class logits live in rows of 16 floats: num_classes + 2 (objectness pair) values per RoI, see bbox_tail_kernel
  if (cfg->num_classes < 1 || cfg->num_classes > 14) { g_create_error = "num_classes out of range (1..14)"; return NUHTC_E_INVALID; }
  if (cfg->max_batch < 1 || cfg->max_batch > 256) { g_create_error = "max_batch out of range"; return NUHTC_E_INVALID; }
  if (cfg->rpn_nms_pre < 1 || cfg->rpn_nms_pre > 4096 || cfg->rpn_max_per_img < 1 || cfg->rpn_max_per_img > 4096) { g_create_error = "rpn_nms_pre / rpn_max_per_img out of range (<=4096)"; return NUHTC_E_INVALID; }
  if (cfg->max_per_img < 1 || cfg->max_per_img > 2048) { g_create_error = "max_per_img out of range"; return NUHTC_E_INVALID; }
  if (cfg->max_cc_proposals < 0 || cfg->max_cc_proposals > 4096) { g_create_error = "max_cc_proposals out of range"; return NUHTC_E_INVALID; }
  if (cfg->schedule != NUHTC_SCHED_LATENCY && cfg->schedule != NUHTC_SCHED_THROUGHPUT) { g_create_error = "schedule must be NUHTC_SCHED_LATENCY or NUHTC_SCHED_THROUGHPUT"; return NUHTC_E_INVALID; }
  if (cfg->att_pool_fp16 != 0 && cfg->att_pool_fp16 != 1) { g_create_error = "att_pool_fp16 must be 0 or 1"; return NUHTC_E_INVALID; }
  if (cfg->matrix_pipe != NUHTC_PIPE_BF16_SPLIT && cfg->matrix_pipe != NUHTC_PIPE_FP32) { g_create_error = "matrix_pipe must be NUHTC_PIPE_BF16_SPLIT or NUHTC_PIPE_FP32"; return NUHTC_E_INVALID; }
  {
    const char* probes[4] = {nuhtc_tu_probe_conv(), nuhtc_tu_probe_gemm(), nuhtc_tu_probe_mlp(), nuhtc_tu_probe_swin()};
    const char* dev = getenv("NUHTC_DEV");
    for (const char* pr : probes)
      if (pr && !(dev && dev[0] == '1')) {
        g_create_error = std::string("this library was compiled with the result-altering dev probe ") + pr + " (wrong results by design); rebuild without it, or set NUHTC_DEV=1 for a timing experiment";
        return NUHTC_E_STATE;
      }
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { g_create_error = "no such HIP device"; return NUHTC_E_HIP; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) { g_create_error = "hipGetDeviceProperties failed"; return NUHTC_E_HIP; }
  if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos) { g_create_error = std::string("this library is built for gfx950 only, device is ") + prop.gcnArchName; return NUHTC_E_INVALID; }
  nuhtc_engine* e = new nuhtc_engine();
  e->cfg = *cfg;
  e->device = device;
  *out = e;
  return 0;
}

// The three streams of an engine (the one handed out by nuhtc_stream + two side streams) live for the whole process: a closed
// engine returns them to a per-device pool and the next engine takes them over.  The stream handed out may be in use by the
// caller's allocator after the engine is gone (PyTorch's caching allocator records events on the stream a block was allocated
// on when the block is freed: a destroyed stream there is a segmentation fault), and a pooled triple keeps the placement it was
// created with (see nuhtc_finalize).
struct StreamTriple { hipStream_t own, side, side2; };
static std::mutex g_stream_mu;
static std::map<int, std::vector<StreamTriple>> g_stream_pool;

void nuhtc_destroy(nuhtc_engine* e) {
  if (!e) return;
  hipSetDevice(e->device);
  hipDeviceSynchronize();
  if (e->own && e->side && e->side2) {
    std::lock_guard<std::mutex> lock(g_stream_mu);
    g_stream_pool[e->device].push_back(StreamTriple{e->own, e->side, e->side2});
  } else {
    if (e->side) hipStreamDestroy(e->side);
    if (e->side2) hipStreamDestroy(e->side2);
  }
  if (e->ev_side2) hipEventDestroy(e->ev_side2);
  if (e->ev_rpn) hipEventDestroy(e->ev_rpn);
  if (e->ev_side) hipEventDestroy(e->ev_side);
  if (e->ev_fpn) hipEventDestroy(e->ev_fpn);
  if (e->overflow_host) hipHostFree(e->overflow_host);
  for (void* p : e->allocs) hipFree(p);
  delete e;
}

void* nuhtc_stream(nuhtc_engine* e) { return e ? (void*)e->own : nullptr; }

// Names and shapes of the state_dict entries the path reads (the same table as nuhtc_amd/weights.py:schema).
static std::map<std::string, std::vector<int64_t>> weight_schema(int nc) {
  std::map<std::string, std::vector<int64_t>> s;
  auto wb = [&](const std::string& p, std::vector<int64_t> w) { s[p + ".weight"] = w; s[p + ".bias"] = {w[0]}; };
  wb("backbone.patch_embed.projection", {96, 3, 4, 4});
  s["backbone.patch_embed.norm.weight"] = {96}; s["backbone.patch_embed.norm.bias"] = {96};
  for (int st = 0; st < 4; ++st) {
    const int64_t C = 96 << st;
    for (int b = 0; b < DEPTHS[st]; ++b) {
      const std::string p = "backbone.stages." + std::to_string(st) + ".blocks." + std::to_string(b) + ".";
      s[p + "norm1.weight"] = {C}; s[p + "norm1.bias"] = {C}; s[p + "norm2.weight"] = {C}; s[p + "norm2.bias"] = {C};
      s[p + "attn.w_msa.relative_position_bias_table"] = {169, NHEADS[st]};
      wb(p + "attn.w_msa.qkv", {3 * C, C}); wb(p + "attn.w_msa.proj", {C, C});
      wb(p + "ffn.layers.0.0", {4 * C, C}); wb(p + "ffn.layers.1", {C, 4 * C});
    }
    if (st < 3) {
      const std::string p = "backbone.stages." + std::to_string(st) + ".downsample.";
      s[p + "norm.weight"] = {4 * C}; s[p + "norm.bias"] = {4 * C}; s[p + "reduction.weight"] = {2 * C, 4 * C};
    }
    s["backbone.norm" + std::to_string(st) + ".weight"] = {C}; s["backbone.norm" + std::to_string(st) + ".bias"] = {C};
    wb("neck.lateral_convs." + std::to_string(st) + ".conv", {64, C, 1, 1});
    wb("neck.fpn_convs." + std::to_string(st) + ".conv", {64, 64, 3, 3});
    wb("roi_head.semantic_head.lateral_convs." + std::to_string(st) + ".conv", {64, 64, 1, 1});
    wb("roi_head.semantic_head.convs." + std::to_string(st) + ".conv", {64, 64, 3, 3});
    wb("roi_head.mask_head.0.convs." + std::to_string(st) + ".conv", {64, 64, 3, 3});
  }
  wb("rpn_head.rpn_conv", {64, 64, 3, 3}); wb("rpn_head.rpn_cls", {3, 64, 1, 1}); wb("rpn_head.rpn_reg", {12, 64, 1, 1});
  for (int k = 0; k < 3; ++k) {
    const std::string p = "roi_head.bbox_head." + std::to_string(k) + ".";
    wb(p + "shared_fcs.0", {256, 3136}); wb(p + "shared_fcs.1", {256, 256}); wb(p + "fc_cls", {nc + 2, 256}); wb(p + "fc_reg", {4, 256});
  }
  wb("roi_head.mask_head.0.upsample", {64, 64, 2, 2}); wb("roi_head.mask_head.0.conv_logits", {1, 64, 1, 1});
  wb("roi_head.mask_head.0.conv_res.conv", {64, 64, 1, 1});   // unused at test time (res_feat is None), accepted
  wb("roi_head.semantic_head.conv_embedding.conv", {64, 64, 1, 1}); wb("roi_head.semantic_head.conv_logits", {1, 64, 1, 1});
  return s;
}

int nuhtc_load_weight(nuhtc_engine* e, const char* name, const float* host, const int64_t* shape, int ndim) {
  if (!e) return NUHTC_E_INVALID;
  if (!name || !host || !shape) FAIL(e, NUHTC_E_INVALID, "nuhtc_load_weight: null argument");
  if (ndim < 1 || ndim > 4) FAIL(e, NUHTC_E_INVALID, std::string("nuhtc_load_weight: ") + name + ": ndim must be 1..4");
  if (e->finalized) FAIL(e, NUHTC_E_STATE, "load_weight after finalize");
  if (e->schema.empty()) e->schema = weight_schema(e->cfg.num_classes);
  auto it = e->schema.find(name);
  if (it == e->schema.end()) FAIL(e, NUHTC_E_NOTFOUND, std::string("nuhtc_load_weight: unknown weight name: ") + name);
  HostTensor t;
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
  if (t.shape != it->second) FAIL(e, NUHTC_E_INVALID, std::string("nuhtc_load_weight: bad shape for ") + name);
  t.data.assign(host, host + n);
  e->raw[name] = std::move(t);
  return 0;
}

// =============================================================================== device allocation
static int dev_alloc(nuhtc_engine* e, void** p, size_t bytes) {
  bytes = (bytes + 255) & ~(size_t)255;
  if (bytes == 0) bytes = 256;
  HIP_CHECK(e, hipMalloc(p, bytes));
  e->allocs.push_back(*p);
  e->bytes_allocated += bytes;
  return 0;
}

static int upload(nuhtc_engine* e, float** dst, const std::vector<float>& v) {
  int rc = dev_alloc(e, (void**)dst, v.size() * sizeof(float));
  if (rc) return rc;
  HIP_CHECK(e, hipMemcpy(*dst, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
  return 0;
}

int upload_gemm_weight(nuhtc_engine* e, float** dst, const std::vector<float>& v, int N, int K) {
  int rc = upload(e, dst, v);
  if (rc) return rc;
  if (e->cfg.matrix_pipe == NUHTC_PIPE_FP32) return 0;
  if ((size_t)N * K != v.size()) FAIL(e, NUHTC_E_INVALID, "upload_gemm_weight: shape mismatch");
  void* sp = nullptr;
  rc = gemm_make_split(v.data(), N, K, &sp);
  if (rc) FAIL(e, rc, "gemm_make_split failed");
  e->allocs.push_back(sp);
  e->wsplit[*dst] = {sp, N, K};
  return 0;
}

int egemm(nuhtc_engine* e, GemmParams p, hipStream_t s) {
  // products of depth < 96 stay on the fp32 MFMA kernel (4 k-tiles: prologue and epilogue dominate and the fp32 kernel keeps 4
  // workgroups per CU; measured 0.28 vs 0.32-0.40 ms per step for the 64x64 pointwise layers), batched products too
  if (!p.Wsplit && p.batch <= 1 && p.K >= 96) {
    auto it = e->wsplit.find(p.W);
    if (it != e->wsplit.end()) {
      // a split of another geometry under this pointer would be read out of bounds by the kernel, silently: refuse it (the first
      // rows of a weight with the same K are a valid product: the split is row-major in n)
      if (it->second.K != p.K || p.N > it->second.N) FAIL(e, NUHTC_E_STATE, "egemm: the weight's bf16 split was made for another [N][K]");
      p.Wsplit = it->second.planes;
    }
  }
  {   // dev: ablation of the step (tools/dev/r04_ablate.py): 4 = 3x3 convolutions, 8 = 96-column split GEMMs, 64 = every other product
    static const int& skip_ = dev_knob_ref("SKIP", 0);
    if (skip_ && (p.amode == A_CONV3 ? (skip_ & 4) : (p.Wsplit && p.N % 96 == 0) ? (skip_ & 8) : (skip_ & 64))) return 0;
  }
  return launch_gemm(p, s);
}

static int upload_fuse(nuhtc_engine* e, void** dst, const std::vector<float>& w, int N2) {
  *dst = nullptr;
  if (e->cfg.matrix_pipe != NUHTC_PIPE_BF16_SPLIT) return 0;
  if (w.size() != (size_t)N2 * 64) FAIL(e, NUHTC_E_INVALID, "upload_fuse: shape mismatch");
  const int rc = conv3_pack_fuse(w.data(), N2, dst);
  if (rc) FAIL(e, rc, "conv3_pack_fuse failed");
  e->allocs.push_back(*dst);
  return 0;
}

static int upload_i(nuhtc_engine* e, int** dst, const std::vector<int>& v) {
  int rc = dev_alloc(e, (void**)dst, v.size() * sizeof(int));
  if (rc) return rc;
  HIP_CHECK(e, hipMemcpy(*dst, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice));
  return 0;
}

template <typename T>
static int ws(nuhtc_engine* e, T** p, const char* name, std::vector<int64_t> shape, int dtype) {
  size_t n = 1;
  for (auto d : shape) n *= (size_t)d;
  int rc = dev_alloc(e, (void**)p, n * sizeof(T));
  if (rc) return rc;
  if (name) e->bufs[name] = BufInfo{(void*)*p, shape, dtype};
  return 0;
}

static const HostTensor* raw(nuhtc_engine* e, const std::string& name, std::initializer_list<int64_t> shape) {
  auto it = e->raw.find(name);
  if (it == e->raw.end()) { e->err = "missing weight: " + name; return nullptr; }
  std::vector<int64_t> s(shape);
  if (it->second.shape != s) { e->err = "bad shape for weight: " + name; return nullptr; }
  return &it->second;
}

#define RAW(var, name, ...)                                 \
  const HostTensor* var = raw(e, (name), {__VA_ARGS__});    \
  if (!var) return NUHTC_E_STATE;

// [O][I][3][3] -> [O][(ky*3+kx)*I + i]
static std::vector<float> pack_conv3(const HostTensor& w, int O, int I) {
  std::vector<float> p((size_t)O * 9 * I);
  for (int o = 0; o < O; ++o)
    for (int i = 0; i < I; ++i)
      for (int t = 0; t < 9; ++t) p[((size_t)o * 9 + t) * I + i] = w.data[((size_t)o * I + i) * 9 + t];
  return p;
}

// =============================================================================== finalize
// window_attn_mfma_kernel reads the additive score terms (relative-position bias, shift mask) per lane: the lane of query i = 32 ti + l32
// in half-wave `half` needs, for key tile tj, the 16 accumulator registers r <-> key 32 tj + (r & 3) + 8 (r >> 2) + 4 half.  Packed
// as [ti][q][lane = 32 half + l32][4] with q = (16 tj + r) / 4 (4096 floats per 49 x 49 table): the q-th 16-byte load of a wave covers 1 KB of
// contiguous memory (round 4: with a lane's 32 floats contiguous, [ti][lane][32], every load instruction touched 64 different cache lines; the
// loads of a table that is the same for every window cost 0.08 of the launches' 0.66 ms per step, tools/dev/r04_attn_probe.sh).
static void pack_attn_terms(const float* qk /* [49][49] query-major */, float* out /* 4096 */) {
  for (int ti = 0; ti < 2; ++ti)
    for (int half = 0; half < 2; ++half)
      for (int l = 0; l < 32; ++l)
        for (int tj = 0; tj < 2; ++tj)
          for (int r = 0; r < 16; ++r) {
            const int i = ti * 32 + l, j = tj * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int f = tj * 16 + r, lane = half * 32 + l;
            out[ti * 2048 + ((f >> 2) * 64 + lane) * 4 + (f & 3)] = (i < WS2 && j < WS2) ? qk[i * WS2 + j] : 0.f;
          }
}

static int build_stage_maps(nuhtc_engine* e, int s) {
  StageGeom& g = e->st[s];
  const int B = e->cfg.max_batch;
  // window-row -> token maps (un-shifted and shifted), (mmdet/models/backbones/swin.py:182-226,254-283)
  for (int sh = 0; sh < 2; ++sh) {
    std::vector<int> m((size_t)B * g.nW * WS2);
    for (int b = 0; b < B; ++b)
      for (int wy = 0; wy < g.Hp / WS; ++wy)
        for (int wx = 0; wx < g.Wp / WS; ++wx)
          for (int py = 0; py < WS; ++py)
            for (int px = 0; px < WS; ++px) {
              int ys = wy * WS + py, xs = wx * WS + px;                        // coordinates in the rolled frame
              int y = sh ? (ys + 3) % g.Hp : ys, x = sh ? (xs + 3) % g.Wp : xs;  // roll(-3): rolled[ys] = padded[(ys+3) % Hp]
              size_t row = ((size_t)b * g.nW + (size_t)wy * (g.Wp / WS) + wx) * WS2 + py * WS + px;
              m[row] = (y < g.H && x < g.W) ? (b * g.H * g.W + y * g.W + x) : -1;
            }
    int rc = upload_i(e, &g.map[sh], m);
    if (rc) return rc;
    // the non-padding window rows in window order: the QKV / proj GEMMs run on these only (padding rows of the window
    // image hold the QKV bias, see launch_layernorm_windows)
    std::vector<int> ci(m.size()), ct, vr;
    ct.reserve((size_t)B * g.H * g.W);
    vr.reserve((size_t)B * g.H * g.W);
    for (size_t r = 0; r < m.size(); ++r) {
      ci[r] = m[r] >= 0 ? (int)ct.size() : -1;
      if (m[r] >= 0) { ct.push_back(m[r]); vr.push_back((int)r); }
    }
    rc = upload_i(e, &g.cidx[sh], ci);
    if (rc) return rc;
    rc = upload_i(e, &g.ctok[sh], ct);
    if (rc) return rc;
    rc = upload_i(e, &g.vrow[sh], vr);
    if (rc) return rc;
    std::vector<int> pr;                  // padding rows: tile b's are [b * npad, (b + 1) * npad)
    for (size_t r = 0; r < m.size(); ++r)
      if (m[r] < 0) pr.push_back((int)r);
    g.npad = (int)(pr.size() / (size_t)B);
    if (pr.empty()) pr.push_back(0);
    rc = upload_i(e, &g.prow[sh], pr);
    if (rc) return rc;
    std::vector<int> pb((size_t)g.nW * 2, 0);          // 49 bits per window of the image (tile 0's windows; every tile has the same)
    for (int w = 0; w < g.nW; ++w)
      for (int j = 0; j < WS2; ++j)
        if (m[(size_t)w * WS2 + j] < 0) pb[(size_t)2 * w + (j >> 5)] |= 1 << (j & 31);
    rc = upload_i(e, reinterpret_cast<int**>(&g.padbits[sh]), pb);
    if (rc) return rc;
  }
  g.bias_row = B * g.nW * WS2;
  {
    int rc = upload_i(e, &g.brow, std::vector<int>(1, g.bias_row));
    if (rc) return rc;
  }
  // shift mask on the padded grid (swin.py:197-218)
  std::vector<int> ids((size_t)g.Hp * g.Wp);
  auto region = [](int v, int n) { return v < n - WS ? 0 : (v < n - 3 ? 1 : 2); };
  for (int y = 0; y < g.Hp; ++y)
    for (int x = 0; x < g.Wp; ++x) ids[(size_t)y * g.Wp + x] = region(y, g.Hp) * 3 + region(x, g.Wp);
  std::vector<float> mask((size_t)g.nW * WS2 * WS2);
  for (int wy = 0; wy < g.Hp / WS; ++wy)
    for (int wx = 0; wx < g.Wp / WS; ++wx) {
      int w = wy * (g.Wp / WS) + wx;
      for (int p = 0; p < WS2; ++p)
        for (int q = 0; q < WS2; ++q) {
          int ip = ids[(size_t)(wy * WS + p / WS) * g.Wp + wx * WS + p % WS];
          int iq = ids[(size_t)(wy * WS + q / WS) * g.Wp + wx * WS + q % WS];
          mask[((size_t)w * WS2 + p) * WS2 + q] = ip == iq ? 0.f : -100.f;
        }
    }
  std::vector<float> mp((size_t)g.nW * 4096);
  std::vector<int> any(g.nW, 0);
  for (int w = 0; w < g.nW; ++w) {
    pack_attn_terms(mask.data() + (size_t)w * WS2 * WS2, mp.data() + (size_t)w * 4096);
    for (int i = 0; i < WS2 * WS2; ++i)
      if (mask[(size_t)w * WS2 * WS2 + i] != 0.f) { any[w] = 1; break; }
  }
  int rc = upload(e, &g.mask, mp);
  if (rc) return rc;
  return upload_i(e, &g.mask_any, any);
}

static int fold_ln(const float* W_host, const float* bias_host, const float* g, const float* b, int N, int K, std::vector<float>& w2, std::vector<float>& b2);

int nuhtc_finalize(nuhtc_engine* e) {
  if (!e) return NUHTC_E_INVALID;
  if (e->finalized) FAIL(e, NUHTC_E_STATE, "finalize called twice");
  HIP_CHECK(e, hipSetDevice(e->device));
  const nuhtc_config& c = e->cfg;
  const int B = c.max_batch;
  e->vh = c.valid_h ? c.valid_h : c.tile_h; e->vw = c.valid_w ? c.valid_w : c.tile_w;
  e->Hv = (int)(e->vh * (double)c.scale_factor + 0.5); e->Wv = (int)(e->vw * (double)c.scale_factor + 0.5);   // img_shape
  const int Hn = (e->Hv + 31) / 32 * 32, Wn = (e->Wv + 31) / 32 * 32;                                            // pad_shape
  e->Hn = Hn; e->Wn = Wn;
  int rc;
  std::vector<float> on_g_host[4], on_b_host[4];      // the stages' output norms, for the fold into the FPN laterals
  {
    std::vector<int> tx, ty;
    cv_linear_tables(e->vw, e->Wv, true, tx);
    cv_linear_tables(e->vh, e->Hv, false, ty);
    if ((rc = upload_i(e, &e->rs_xtab, tx)) || (rc = upload_i(e, &e->rs_ytab, ty))) return rc;
  }
  // ---- geometry
  for (int s = 0; s < 4; ++s) {
    StageGeom& g = e->st[s];
    g.H = Hn >> (2 + s); g.W = Wn >> (2 + s); g.C = 96 << s; g.nH = NHEADS[s];
    g.Hp = cdiv(g.H, WS) * WS; g.Wp = cdiv(g.W, WS) * WS;
    g.nW = (g.Hp / WS) * (g.Wp / WS);
    if ((rc = build_stage_maps(e, s))) return rc;
  }
  // ---- patch embed: [96][3][4][4] -> [48][96], k = (kh*4+kw)*3 + c
  {
    RAW(w, "backbone.patch_embed.projection.weight", 96, 3, 4, 4);
    RAW(b, "backbone.patch_embed.projection.bias", 96);
    RAW(g, "backbone.patch_embed.norm.weight", 96);
    RAW(be, "backbone.patch_embed.norm.bias", 96);
    std::vector<float> p(48 * 96);
    for (int o = 0; o < 96; ++o)
      for (int ch = 0; ch < 3; ++ch)
        for (int kh = 0; kh < 4; ++kh)
          for (int kw = 0; kw < 4; ++kw) p[((kh * 4 + kw) * 3 + ch) * 96 + o] = w->data[((o * 3 + ch) * 4 + kh) * 4 + kw];
    if ((rc = upload(e, &e->pe_w, p)) || (rc = upload(e, &e->pe_b, b->data)) || (rc = upload(e, &e->pe_g, g->data)) ||
        (rc = upload(e, &e->pe_beta, be->data)))
      return rc;
  }
  // ---- Swin blocks
  std::vector<int> rel(WS2 * WS2);
  for (int a = 0; a < WS2; ++a)
    for (int b2 = 0; b2 < WS2; ++b2) rel[a * WS2 + b2] = (a / WS - b2 / WS + WS - 1) * (2 * WS - 1) + (a % WS - b2 % WS + WS - 1);
  for (int s = 0; s < 4; ++s) {
    const int C = 96 << s, nH = NHEADS[s];
    for (int b = 0; b < DEPTHS[s]; ++b) {
      BlockW bw{};
      std::string p = "backbone.stages." + std::to_string(s) + ".blocks." + std::to_string(b) + ".";
      RAW(n1w, p + "norm1.weight", C); RAW(n1b, p + "norm1.bias", C);
      RAW(tab, p + "attn.w_msa.relative_position_bias_table", 169, nH);
      RAW(qw, p + "attn.w_msa.qkv.weight", 3 * C, C); RAW(qb, p + "attn.w_msa.qkv.bias", 3 * C);
      RAW(pw, p + "attn.w_msa.proj.weight", C, C); RAW(pb, p + "attn.w_msa.proj.bias", C);
      RAW(n2w, p + "norm2.weight", C); RAW(n2b, p + "norm2.bias", C);
      RAW(f1w, p + "ffn.layers.0.0.weight", 4 * C, C); RAW(f1b, p + "ffn.layers.0.0.bias", 4 * C);
      RAW(f2w, p + "ffn.layers.1.weight", C, 4 * C); RAW(f2b, p + "ffn.layers.1.bias", C);
      std::vector<float> rb((size_t)nH * WS2 * WS2);
      for (int h = 0; h < nH; ++h)
        for (int i = 0; i < WS2 * WS2; ++i) rb[(size_t)h * WS2 * WS2 + i] = tab->data[(size_t)rel[i] * nH + h];
      std::vector<float> rbT((size_t)nH * 4096);        // per-lane packed terms of the attention kernel (pack_attn_terms)
      for (int h = 0; h < nH; ++h) pack_attn_terms(rb.data() + (size_t)h * WS2 * WS2, rbT.data() + (size_t)h * 4096);
      if ((rc = upload(e, &bw.relbT, rbT))) return rc;
      if ((rc = upload(e, &bw.n1g, n1w->data)) || (rc = upload(e, &bw.n1b, n1b->data)) ||           (rc = upload_gemm_weight(e, &bw.qkv_w, qw->data, 3 * C, C)) || (rc = upload(e, &bw.qkv_b, qb->data)) || (rc = upload_gemm_weight(e, &bw.proj_w, pw->data, C, C)) ||
          (rc = upload(e, &bw.proj_b, pb->data)) || (rc = upload(e, &bw.n2g, n2w->data)) || (rc = upload(e, &bw.n2b, n2b->data)) ||
          (rc = upload_gemm_weight(e, &bw.f1_w, f1w->data, 4 * C, C)) || (rc = upload(e, &bw.f1_b, f1b->data)) || (rc = upload_gemm_weight(e, &bw.f2_w, f2w->data, C, 4 * C)) ||
          (rc = upload(e, &bw.f2_b, f2b->data)))
        return rc;
      if (e->cfg.matrix_pipe == NUHTC_PIPE_BF16_SPLIT && !lnqkv_supported(C) && !mlp_supported(C)) {
        // the two norms of the block ride in the A path of the linear behind them (gemm.hip A_LN): y = ((x - mean) rstd gamma + beta) W^T + b
        //   = rstd ((x - mean) (W diag gamma)^T) + (b + W beta); W' is rounded once to fp32 (its split is exact from there), b' summed in fp64
        auto fold = [&](const std::vector<float>& W, const std::vector<float>& bias, const std::vector<float>& gam, const std::vector<float>& bet, int N,
                        float** wdev, float** bdev) -> int {
          std::vector<float> w2((size_t)N * C), b2(N);
          for (int n = 0; n < N; ++n) {
            double acc = bias[n];
            for (int k = 0; k < C; ++k) {
              w2[(size_t)n * C + k] = W[(size_t)n * C + k] * gam[k];
              acc += (double)W[(size_t)n * C + k] * (double)bet[k];
            }
            b2[n] = (float)acc;
          }
          int r = upload_gemm_weight(e, wdev, w2, N, C);
          return r ? r : upload(e, bdev, b2);
        };
        if ((rc = fold(qw->data, qb->data, n1w->data, n1b->data, 3 * C, &bw.qkv_wln, &bw.qkv_bln)) ||
            (rc = fold(f1w->data, f1b->data, n2w->data, n2b->data, 4 * C, &bw.f1_wln, &bw.f1_bln)))
          return rc;
      }
      if (e->cfg.matrix_pipe == NUHTC_PIPE_BF16_SPLIT && lnqkv_supported(C)) {
        std::vector<unsigned short> st;
        lnqkv_pack_stream(qw->data.data(), C, st);
        if ((rc = dev_alloc(e, &bw.qkv_stream, st.size() * 2))) return rc;
        HIP_CHECK(e, hipMemcpy(bw.qkv_stream, st.data(), st.size() * 2, hipMemcpyHostToDevice));
      }
      if (e->cfg.matrix_pipe == NUHTC_PIPE_BF16_SPLIT && mlp_supported(C)) {
        std::vector<unsigned short> st;
        mlp_pack_stream(f1w->data.data(), f2w->data.data(), C, st);
        if ((rc = dev_alloc(e, &bw.mlp_stream, st.size() * 2))) return rc;
        HIP_CHECK(e, hipMemcpy(bw.mlp_stream, st.data(), st.size() * 2, hipMemcpyHostToDevice));
        proj_pack_stream(pw->data.data(), C, st);
        if ((rc = dev_alloc(e, &bw.proj_stream, st.size() * 2))) return rc;
        HIP_CHECK(e, hipMemcpy(bw.proj_stream, st.data(), st.size() * 2, hipMemcpyHostToDevice));
      }
      e->blocks[s].push_back(bw);
    }
    {
      std::string p = "backbone.norm" + std::to_string(s) + ".";
      RAW(w, p + "weight", C); RAW(b, p + "bias", C);
      if ((rc = upload(e, &e->on_g[s], w->data)) || (rc = upload(e, &e->on_b[s], b->data))) return rc;
      on_g_host[s] = w->data; on_b_host[s] = b->data;
    }
    if (s < 3) {
      // PatchMerging: nn.Unfold order k = c*4 + q (q = kh*2+kw)  ->  gather order k' = q*C + c   (transformer.py:363-385)
      std::string p = "backbone.stages." + std::to_string(s) + ".downsample.";
      RAW(nw, p + "norm.weight", 4 * C); RAW(nb, p + "norm.bias", 4 * C); RAW(rw, p + "reduction.weight", 2 * C, 4 * C);
      std::vector<float> g2(4 * C), b2(4 * C), w2((size_t)2 * C * 4 * C);
      for (int q = 0; q < 4; ++q)
        for (int ch = 0; ch < C; ++ch) {
          g2[q * C + ch] = nw->data[ch * 4 + q];
          b2[q * C + ch] = nb->data[ch * 4 + q];
          for (int n = 0; n < 2 * C; ++n) w2[(size_t)n * 4 * C + q * C + ch] = rw->data[(size_t)n * 4 * C + ch * 4 + q];
        }
      if ((rc = upload(e, &e->mg_g[s], g2)) || (rc = upload(e, &e->mg_b[s], b2)) || (rc = upload_gemm_weight(e, &e->mg_w[s], w2, 2 * C, 4 * C))) return rc;
      if (e->cfg.matrix_pipe == NUHTC_PIPE_BF16_SPLIT && e->st[s].H % 2 == 0 && e->st[s].W % 2 == 0) {
        // the merging norm in the A path of the reduction linear (gemm.hip A_LN, two segments per row): W' = W diag(gamma), b' = W beta
        std::vector<float> wl, bl;
        fold_ln(w2.data(), nullptr, g2.data(), b2.data(), 2 * C, 4 * C, wl, bl);
        const StageGeom& g = e->st[s];
        std::vector<int> src((size_t)B * (g.H / 2) * (g.W / 2));
        for (int b = 0; b < B; ++b)
          for (int y2 = 0; y2 < g.H / 2; ++y2)
            for (int x2 = 0; x2 < g.W / 2; ++x2) src[((size_t)b * (g.H / 2) + y2) * (g.W / 2) + x2] = (b * g.H + 2 * y2) * g.W + 2 * x2;
        if ((rc = upload_gemm_weight(e, &e->mg_wln[s], wl, 2 * C, 4 * C)) || (rc = upload(e, &e->mg_bln[s], bl)) || (rc = upload_i(e, &e->mg_src[s], src))) return rc;
      }
    }
  }
  // ---- FPN
  for (int i = 0; i < 4; ++i) {
    const int C = 96 << i;
    RAW(lw, "neck.lateral_convs." + std::to_string(i) + ".conv.weight", 64, C, 1, 1);
    RAW(lb, "neck.lateral_convs." + std::to_string(i) + ".conv.bias", 64);
    RAW(fw, "neck.fpn_convs." + std::to_string(i) + ".conv.weight", 64, 64, 3, 3);
    RAW(fb, "neck.fpn_convs." + std::to_string(i) + ".conv.bias", 64);
    if ((rc = upload_gemm_weight(e, &e->lat_w[i], lw->data, 64, C)) || (rc = upload(e, &e->lat_b[i], lb->data)) ||
        (rc = upload_gemm_weight(e, &e->fpn_w[i], pack_conv3(*fw, 64, 64), 64, 576)) || (rc = upload(e, &e->fpn_b[i], fb->data)))
      return rc;
    if (e->cfg.matrix_pipe == NUHTC_PIPE_BF16_SPLIT) {      // the stage's output norm folded into its lateral (gemm.hip A_LN, N = 64)
      std::vector<float> wl, bl;
      fold_ln(lw->data.data(), lb->data.data(), on_g_host[i].data(), on_b_host[i].data(), 64, C, wl, bl);
      if ((rc = upload_gemm_weight(e, &e->lat_wln[i], wl, 64, C)) || (rc = upload(e, &e->lat_bln[i], bl))) return rc;
    }
  }
  // ---- RPN: 3x3 conv, then cls(3)+reg(12) fused into one N=32 pointwise layer (cols 0-2 cls, 3-14 reg, rest 0)
  {
    RAW(cw, "rpn_head.rpn_conv.weight", 64, 64, 3, 3); RAW(cb, "rpn_head.rpn_conv.bias", 64);
    RAW(kw, "rpn_head.rpn_cls.weight", 3, 64, 1, 1); RAW(kb, "rpn_head.rpn_cls.bias", 3);
    RAW(rw, "rpn_head.rpn_reg.weight", 12, 64, 1, 1); RAW(rb, "rpn_head.rpn_reg.bias", 12);
    std::vector<float> w(32 * 64, 0.f), b(32, 0.f);
    for (int n = 0; n < 3; ++n) { b[n] = kb->data[n]; for (int k = 0; k < 64; ++k) w[n * 64 + k] = kw->data[n * 64 + k]; }
    for (int n = 0; n < 12; ++n) { b[3 + n] = rb->data[n]; for (int k = 0; k < 64; ++k) w[(3 + n) * 64 + k] = rw->data[n * 64 + k]; }
    if ((rc = upload_gemm_weight(e, &e->rpn_w, pack_conv3(*cw, 64, 64), 64, 576)) || (rc = upload(e, &e->rpn_b, cb->data)) ||
        (rc = upload_gemm_weight(e, &e->rpn_hw, w, 32, 64)) || (rc = upload(e, &e->rpn_hb, b)) || (rc = upload_fuse(e, &e->rpn_hf, w, 32)))
      return rc;
  }
  // ---- semantic head
  {
    const std::string p = "roi_head.semantic_head.";
    for (int i = 0; i < 4; ++i) {
      RAW(lw, p + "lateral_convs." + std::to_string(i) + ".conv.weight", 64, 64, 1, 1);
      RAW(lb, p + "lateral_convs." + std::to_string(i) + ".conv.bias", 64);
      RAW(cw, p + "convs." + std::to_string(i) + ".conv.weight", 64, 64, 3, 3);
      RAW(cb, p + "convs." + std::to_string(i) + ".conv.bias", 64);
      if ((rc = upload_gemm_weight(e, &e->sem_lw[i], lw->data, 64, 64)) || (rc = upload(e, &e->sem_lb[i], lb->data)) || (rc = upload_fuse(e, &e->sem_lf[i], lw->data, 64)) ||
          (rc = upload_gemm_weight(e, &e->sem_cw[i], pack_conv3(*cw, 64, 64), 64, 576)) || (rc = upload(e, &e->sem_cb[i], cb->data)))
        return rc;
    }
    RAW(ew, p + "conv_embedding.conv.weight", 64, 64, 1, 1); RAW(eb, p + "conv_embedding.conv.bias", 64);
    RAW(gw, p + "conv_logits.weight", 1, 64, 1, 1); RAW(gb, p + "conv_logits.bias", 1);
    if ((rc = upload_gemm_weight(e, &e->sem_ew, ew->data, 64, 64)) || (rc = upload(e, &e->sem_eb, eb->data)) || (rc = upload_fuse(e, &e->sem_ef, ew->data, 64)) || (rc = upload(e, &e->sem_gw, gw->data)) ||
        (rc = upload(e, &e->sem_gb, gb->data)))
      return rc;
  }
  if ((rc = finalize_roi(e))) return rc;

  // ---- workspace (sized for max_batch)
  const StageGeom& g0 = e->st[0];
  if ((rc = ws(e, &e->img, "img", {B, Hn, Wn, 3}, 0))) return rc;
  size_t max_tok = 0, max_win = 0, max_qkv = 0, max_hid = 0;
  const int max_c = e->st[3].C;      // + one row of 3 max_c behind the window image: the bias row of StageGeom::bias_row
  for (int s = 0; s < 4; ++s) {
    const StageGeom& g = e->st[s];
    max_tok = std::max(max_tok, (size_t)g.H * g.W * g.C);
    max_win = std::max(max_win, (size_t)g.nW * WS2 * g.C);
    max_qkv = std::max(max_qkv, (size_t)g.nW * WS2 * 3 * g.C);
    max_hid = std::max(max_hid, (size_t)g.H * g.W * 4 * g.C);
  }
  if ((rc = ws(e, &e->tokA, "tokens", {B, (int64_t)max_tok}, 0)) || (rc = ws(e, &e->tokB, nullptr, {B, (int64_t)max_tok}, 0)) ||
      (rc = ws(e, &e->xw, nullptr, {B, (int64_t)max_win}, 0)) || (rc = ws(e, &e->qkv, nullptr, {(int64_t)B * (int64_t)max_qkv + 3 * (int64_t)max_c}, 0)) ||
      (rc = ws(e, &e->att, nullptr, {B, (int64_t)max_win}, 0)) || (rc = ws(e, &e->hid, nullptr, {B, (int64_t)std::max(max_hid, max_qkv)}, 0)))
    return rc;
  e->tok[0] = e->tokA; e->tok[1] = e->tokB;
  for (int s = 2; s < 4; ++s)
    if ((rc = ws(e, &e->tok[s], nullptr, {B, (int64_t)e->st[s].H * e->st[s].W * e->st[s].C}, 0))) return rc;
  for (int s = 0; s < 4; ++s)
    if ((rc = ws(e, &e->ln_out[s], nullptr, {B, (int64_t)e->st[1].H * e->st[1].W, 8}, 0))) return rc;
  if ((rc = ws(e, &e->ln_part, nullptr, {B, (int64_t)e->st[1].H * e->st[1].W, 8}, 0)) ||     // (stage 1, one partial per token: the same again)
      (rc = ws(e, &e->ln_part2, nullptr, {B, (int64_t)e->st[1].H * e->st[1].W, 8}, 0)))
    return rc;     // rows x (C / 96) x 2 is the same in stages 2-4
  for (int s = 0; s < 4; ++s) {
    const StageGeom& g = e->st[s];
    std::string n = std::to_string(s);
    if ((rc = ws(e, &e->c[s], ("c" + n).c_str(), {B, g.H, g.W, g.C}, 0)) || (rc = ws(e, &e->lat[s], ("lat" + n).c_str(), {B, g.H, g.W, 64}, 0)) ||
        (rc = ws(e, &e->x[s], ("x" + n).c_str(), {B, g.H, g.W, 64}, 0)) || (rc = ws(e, &e->rpn[s], ("rpn" + n).c_str(), {B, g.H, g.W, 32}, 0)) ||
        (rc = ws(e, &e->semg[s], nullptr, {B, g.H, g.W, 64}, 0)))
      return rc;
  }
  if ((rc = ws(e, &e->tmpA, nullptr, {B, g0.H, g0.W, 64}, 0)) || (rc = ws(e, &e->tmpB, nullptr, {B, g0.H, g0.W, 64}, 0)) ||
      (rc = ws(e, &e->tmpR, nullptr, {B, g0.H, g0.W, 64}, 0)) ||
      (rc = ws(e, &e->sem_feat, "sem_feat", {B, g0.H, g0.W, 64}, 0)) || (rc = ws(e, &e->x0sem, "x0sem", {B, g0.H, g0.W, 64}, 0)) || (rc = ws(e, &e->sem_pred, "sem_pred", {B, g0.H, g0.W}, 0)))
    return rc;
  if ((rc = alloc_roi_workspace(e))) return rc;
  {
    // The side stream carries the RPN branch (and the mid-size RoI class) beside the main stream's semantic branch.  Its NMS
    // launches are large grids of one-wave workgroups that slow a co-running main-stream kernel tenfold while they last (a 20 us
    // kernel of the component-proposal chain takes 200 us beside nms_mask_levels_kernel).  Running the branch at the lowest
    // stream priority (NUHTC_SIDE_PRIO=1, dev) frees the main stream but stretches the RPN chain by the same amount, and the
    // join then waits for it: measured 11.48-11.52 against 11.41-11.46 ms per step, so the default stays equal priority.
    // The stream handed out by nuhtc_stream() and the two side streams are created back to back: the runtime deals its hardware
    // queues to streams in creation order and the queues go round the command processor's four pipes, so the three end up on
    // three different pipes whatever GPU_MAX_HW_QUEUES is.  (Two queues of one pipe that wait on each other's events stall each
    // other: with 8-24 queues an engine whose side stream shared the pipe of the caller's stream ran 30 % slower.)
    bool pooled = false;
    {
      std::lock_guard<std::mutex> lock(g_stream_mu);
      auto& pool = g_stream_pool[e->device];
      if (!pool.empty()) { e->own = pool.back().own; e->side = pool.back().side; e->side2 = pool.back().side2; pool.pop_back(); pooled = true; }
    }
    if (!pooled) {
      HIP_CHECK(e, hipStreamCreateWithFlags(&e->own, hipStreamNonBlocking));
      int least = 0, greatest = 0;
      HIP_CHECK(e, hipDeviceGetStreamPriorityRange(&least, &greatest));
      HIP_CHECK(e, hipStreamCreateWithPriority(&e->side, hipStreamNonBlocking, dev_knob("SIDE_PRIO", 0) ? least : 0));
      HIP_CHECK(e, hipStreamCreateWithFlags(&e->side2, hipStreamNonBlocking));
    }
  }
  HIP_CHECK(e, hipEventCreateWithFlags(&e->ev_rpn, hipEventDisableTiming));
  HIP_CHECK(e, hipEventCreateWithFlags(&e->ev_side, hipEventDisableTiming));
  HIP_CHECK(e, hipEventCreateWithFlags(&e->ev_fpn, hipEventDisableTiming));
  HIP_CHECK(e, hipEventCreateWithFlags(&e->ev_side2, hipEventDisableTiming));
  HIP_CHECK(e, hipDeviceSynchronize());
  e->raw.clear();
  e->finalized = true;
  return 0;
}

// =============================================================================== dense part of the path
static GemmParams gp(const float* A, const float* W, const float* bias, float* C, int M, int N, int K) {
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = A; p.W = W; p.bias = bias; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = K; p.ldc = N; p.alpha = 1.f; p.m_mul = 1;
  return p;
}

#define RUN(expr)                                                                          \
  do {                                                                                     \
    int _rc = (expr);                                                                      \
    if (_rc) { e->err = std::string(#expr) + " failed (" + std::to_string(_rc) + ")"; return _rc; } \
  } while (0)

static int conv3x3(nuhtc_engine* e, const float* in, const float* w, const float* b, float* out, int nimg, int H, int W, int act,
                   const int* m_dev, int m_mul, hipStream_t s, const Conv3Fuse* fuse = nullptr) {
  GemmParams p = gp(in, w, b, out, nimg * H * W, 64, 576);
  p.amode = A_CONV3; p.cH = H; p.cW = W; p.cC = 64; p.act = act; p.m_dev = m_dev; p.m_mul = m_mul; p.fuse = fuse;
  return egemm(e, p, s);
}
static Conv3Fuse pointwise(int N2, const void* w2f, const float* bias2, float* out2, int act2, int store_out) {
  Conv3Fuse f;
  memset(&f, 0, sizeof(f));
  f.N2 = N2; f.w2f = w2f; f.bias2 = bias2; f.out2 = out2; f.act2 = act2; f.store_out = store_out;
  return f;
}

int run_backbone(nuhtc_engine* e, int B, hipStream_t s) {
  const int Hn = e->Hn, Wn = e->Wn;
  // the Swin linears take the block-tile form of the engine's schedule (nuhtc_config.schedule, gemm.hip)
  auto linear = [&](GemmParams p) { p.throughput = e->cfg.schedule == NUHTC_SCHED_THROUGHPUT; return egemm(e, p, s); };
  {
    // dev: 1 (the tree) = resize + Normalize + Pad inside the patch embedding (one launch, no `img` tensor); 0 = preproc_kernel, then patch_embed_kernel
    static const int& preproc_fused = dev_knob_ref("PREPROC_FUSED", 1);
    float mi[6];
    for (int i = 0; i < 3; ++i) { mi[i] = e->cfg.mean[i]; mi[3 + i] = (float)(1.0 / (double)e->cfg.std[i]); }
    e->img_stale = preproc_fused != 0;
    if (preproc_fused) {
      RUN(launch_patch_embed_tiles(e->in_tiles, B, e->cfg.tile_h, e->cfg.tile_w, Hn, Wn, e->Hv, e->Wv, e->rs_xtab, e->rs_ytab, e->in_swap, mi, e->pe_w, e->pe_b, e->pe_g, e->pe_beta,
                                   e->tokA, s));
    } else {
      RUN(launch_preproc(e->in_tiles, e->img, B, e->cfg.tile_h, e->cfg.tile_w, Hn, Wn, e->Hv, e->Wv, e->rs_xtab, e->rs_ytab, e->in_swap, mi, s));
      RUN(launch_patch_embed(e->img, e->pe_w, e->pe_b, e->pe_g, e->pe_beta, e->tokA, B, Hn, Wn, s));
    }
  }
  float* x = e->tok[0];
  // dev: 0 = the norms of stages 2-4 as kernels of their own (round 4); 1 = in the A path of the linear behind them, statistics by a kernel
  // of their own; 2 (the tree) = statistics left by the epilogue of the GEMM that produced the tensor
  static const int& ln_in_a = dev_knob_ref("LN_IN_A", 2);
  // dev: 1 (the tree) = the PatchMerging norms as well (two-segment rows, statistics from the last block's FFN); 0 = merge_ln_kernel + plain GEMM
  static const int& merge_ln_in_a = dev_knob_ref("MERGE_LN_IN_A", 1);
  const float* first_part = e->ln_part;      // where the first block of the stage finds its LN1 partials: ln_part2 behind a merging linear in A_LN form
  // dev: 1 (the tree) = the stages' output norms in the A path of the FPN laterals (run_neck_heads); 0 = layernorm kernels writing c[st]
  static const int& out_ln_in_a = dev_knob_ref("OUT_LN_IN_A", 1);
  e->out_ln_folded = e->lat_wln[0] && ln_in_a >= 2 && out_ln_in_a;
  e->last_batch = B;
  for (int st = 0; st < 4; ++st) {
    const StageGeom& g = e->st[st];
    const int T = B * g.H * g.W, Mw = B * g.nW * WS2, C = g.C;
    const bool merge_a = st < 3 && e->mg_wln[st] && ln_in_a >= 2 && merge_ln_in_a && !e->blocks[st].empty();
    const bool final_stats = merge_a || e->out_ln_folded;      // the last block's FFN leaves the partials of the stage's final tensor in ln_out[st]
    float* xalt = st < 3 ? e->tok[st + 1] : nullptr;
    for (size_t b = 0; b < e->blocks[st].size(); ++b) {
      const BlockW& w = e->blocks[st][b];
      const int sh = (int)(b & 1);
      // x += proj(attn(LN1(x)))      (mmdet swin.py:356-363)
      // Only the T real tokens go through the two linears: LN1 writes them in window order without the padding rows
      // (xw, T rows), the QKV GEMM scatters its rows into the window image, whose padding rows are the QKV bias
      // (LN of a zero-padded token is 0 after swin.py:341-343's F.pad, so its qkv is the bias), attention writes the
      // non-padding rows of its output compactly again and proj scatters them back to token order.
      static const int& fused_qkv = dev_knob_ref("FUSED_QKV", 1);
      // On the split pipe the attention kernel never reads a padding row (StageGeom::padbits): the launch that writes the window image writes ONE
      // bias row instead of the padding rows (stage 4: 72 % of the image's rows, stages 2-3: 20 %).  dev knob 0 = round 4's image
      static const int& attn_padbits = dev_knob_ref("ATTN_PADBITS", 1);
      const bool split_attn = e->cfg.matrix_pipe != NUHTC_PIPE_FP32;
      bool one_bias_row = false;
      if (w.qkv_stream && fused_qkv) {       // one kernel: LN1, window gather, QKV linear (mlp.hip) + the bias rows of the padding tokens
        one_bias_row = split_attn && attn_padbits;
        RUN(launch_swin_lnqkv(x, e->qkv, g.ctok[sh], g.vrow[sh], one_bias_row ? g.brow : g.prow[sh], one_bias_row ? 1 : B * g.npad, w.n1g, w.n1b, w.qkv_stream, w.qkv_b, T, C, s));
      } else if (w.qkv_wln && ln_in_a) {     // the norm rides in the linear's A path; the launch's extra workgroups write the bias rows of the padding tokens
        const bool epi = ln_in_a >= 2;       // the statistics were left by the epilogue of the GEMM that produced x (fc2, or the patch merging)
        if (!epi) RUN(launch_ln_stats(x, e->ln_part, T, C, s));
        GemmParams p = gp(x, w.qkv_wln, w.qkv_bln, e->qkv, T, 3 * C, C);
        p.amode = A_LN; p.ln_part = epi && b == 0 ? first_part : e->ln_part; p.ln_nparts = epi ? C / 96 : 1; p.a_rows = g.ctok[sh];
        one_bias_row = split_attn && attn_padbits;
        p.pad_rows = one_bias_row ? g.brow : g.prow[sh]; p.n_pad = one_bias_row ? 1 : B * g.npad; p.pad_val = w.qkv_b;
        p.store = ST_ROWMAP; p.row_map = g.vrow[sh];
        RUN(linear(p));
      } else {
      RUN(launch_layernorm_windows(x, g.map[sh], g.cidx[sh], w.n1g, w.n1b, e->xw, e->qkv, w.qkv_b, Mw, C, s));
      {
        GemmParams p = gp(e->xw, w.qkv_w, w.qkv_b, e->qkv, T, 3 * C, C);
        p.store = ST_ROWMAP; p.row_map = g.vrow[sh];
        RUN(linear(p));
      }
      }
      // x += W2·gelu(W1·LN2(x))      (swin.py:365-367, mmcv FFN)
      static const int& fused_mlp = dev_knob_ref("FUSED_MLP", 1);
      static const int& fused_proj = dev_knob_ref("FUSED_PROJ", 1);
      const bool mlp1 = w.mlp_stream && fused_mlp, proj1 = mlp1 && w.proj_stream && fused_proj;
      // Where the projection rides in front of the fused FFN kernel (stage 1, round 4) the attention kernel writes its rows in TOKEN
      // order (window row -> token map) instead of the compact window order the projection GEMM scatters from
      RUN(launch_window_attn(e->qkv, w.relbT, sh ? g.mask : nullptr, sh ? g.mask_any : nullptr, proj1 ? g.map[sh] : g.cidx[sh], e->att, B * g.nW, g.nW, C, g.nH,
                             split_attn, s, one_bias_row ? g.padbits[sh] : nullptr, g.bias_row));
      if (!proj1) {
        GemmParams p = gp(e->att, w.proj_w, w.proj_b, x, T, C, C);
        p.store = ST_ROWMAP; p.row_map = g.ctok[sh]; p.res = x; p.ldr = C;
        if (w.f1_wln && ln_in_a >= 2) p.stats_out = e->ln_part;      // LN2 rides in fc1: its statistics leave with the rows
        RUN(linear(p));
      }
      if (mlp1) {      // one kernel: [attention projection + residual,] LN2, both linears, GELU and the residual (mlp.hip)
        RUN(launch_swin_mlp(x, x, w.n2g, w.n2b, w.mlp_stream, w.f1_b, w.f2_b, T, C, s, proj1 ? e->att : nullptr, w.proj_stream, w.proj_b,
                            final_stats && b + 1 == e->blocks[st].size() ? e->ln_out[st] : nullptr));      // the merging / output norm's partials leave with the last block's rows
      } else {
      if (w.f1_wln && ln_in_a) {
        const bool epi = ln_in_a >= 2;
        if (!epi) RUN(launch_ln_stats(x, e->ln_part, T, C, s));
        GemmParams p = gp(x, w.f1_wln, w.f1_bln, e->hid, T, 4 * C, C);
        p.amode = A_LN; p.ln_part = e->ln_part; p.ln_nparts = epi ? C / 96 : 1;
        p.act = ACT_GELU;
        RUN(linear(p));
      } else {
      RUN(launch_layernorm(x, nullptr, w.n2g, w.n2b, e->xw, T, C, s));
      {
        GemmParams p = gp(e->xw, w.f1_w, w.f1_b, e->hid, T, 4 * C, C);
        p.act = ACT_GELU;
        RUN(linear(p));
      }
      }
      {
        GemmParams p = gp(e->hid, w.f2_w, w.f2_b, x, T, C, 4 * C);
        p.res = x; p.ldr = C;
        if (ln_in_a >= 2 && b + 1 < e->blocks[st].size() && e->blocks[st][b + 1].qkv_wln) p.stats_out = e->ln_part;   // LN1 of the next block rides in its QKV linear
        if (final_stats && b + 1 == e->blocks[st].size()) p.stats_out = e->ln_out[st];                                   // ... the merging norm in the reduction linear, the output norm in the FPN lateral
        RUN(linear(p));
      }
      }
      if (e->debug_tokens) {
        auto it = e->bufs.find("tok_s" + std::to_string(st) + "b" + std::to_string(b));
        if (it != e->bufs.end()) hipMemcpyAsync(it->second.ptr, x, (size_t)T * C * sizeof(float), hipMemcpyDeviceToDevice, s);
      }
    }
    if (!e->out_ln_folded) RUN(launch_layernorm(x, nullptr, e->on_g[st], e->on_b[st], e->c[st], T, C, s));   // swin.py:756-762 (tokens == NHWC); else: in the lateral's A path
    if (st < 3) {
      const bool next_ln = ln_in_a >= 2 && !e->blocks[st + 1].empty() && e->blocks[st + 1][0].qkv_wln;   // LN1 of the next stage's first block rides in its QKV linear
      if (merge_a) {     // transformer.py:363-385 in one launch: row m = LayerNorm of the 2 x 2 tokens at mg_src[m] (two runs of 2 C floats, W tokens apart)
        GemmParams p = gp(x, e->mg_wln[st], e->mg_bln[st], xalt, T / 4, 2 * C, 4 * C);
        p.lda = C; p.amode = A_LN; p.ln_part = e->ln_out[st]; p.ln_nparts = 4 * (C / 96); p.a_rows = e->mg_src[st]; p.seg_k = 2 * C; p.seg_rows = g.W;
        if (next_ln) p.stats_out = e->ln_part2;      // not ln_part: other workgroups of this launch are still reading it
        RUN(linear(p));
        first_part = e->ln_part2;
      } else {
        RUN(launch_merge_ln(x, e->mg_g[st], e->mg_b[st], e->xw, B, g.H, g.W, C, s));
        GemmParams p = gp(e->xw, e->mg_w[st], nullptr, xalt, T / 4, 2 * C, 4 * C);
        if (next_ln) p.stats_out = e->ln_part;
        RUN(linear(p));
        first_part = e->ln_part;
      }
      x = xalt;
    }
  }
  return 0;
}

int run_neck_heads(nuhtc_engine* e, int B, hipStream_t s) {
  // FPN (mmdet/models/necks/fpn.py:152-179): laterals coarse->fine with the nearest-upsampled coarser lateral added in the epilogue
  for (int i = 3; i >= 0; --i) {
    const StageGeom& g = e->st[i];
    GemmParams p = gp(e->c[i], e->lat_w[i], e->lat_b[i], e->lat[i], B * g.H * g.W, 64, g.C);
    if (e->out_ln_folded) {      // c[i] = LayerNorm(tok[i]) is never written: the lateral takes the stage's raw tokens and its partials (gemm.hip A_LN)
      p.A = e->tok[i]; p.W = e->lat_wln[i]; p.bias = e->lat_bln[i];
      p.amode = A_LN; p.ln_part = e->ln_out[i]; p.ln_nparts = g.C / 96;
    }
    if (i < 3) { p.up = e->lat[i + 1]; p.upH = g.H; p.upW = g.W; }
    RUN(egemm(e, p, s));
  }
  // Pointwise layers that follow a 3x3 convolution are computed in that convolution's epilogue on the split pipe (conv.hip,
  // Conv3Fuse): the semantic head's lateral 1x1 rides on the FPN output conv of its level, the RPN's cls + reg layer on the RPN
  // conv (whose output is then never stored), conv_logits + conv_embedding (+ x0 + sem) on the semantic head's last conv.
  const bool fuse = e->rpn_hf && conv3_fuse_available();
  for (int i = 0; i < 4; ++i) {
    const StageGeom& g = e->st[i];
    if (fuse) {
      const Conv3Fuse f = pointwise(64, e->sem_lf[i], e->sem_lb[i], e->semg[i], ACT_NONE, 1);
      RUN(conv3x3(e, e->lat[i], e->fpn_w[i], e->fpn_b[i], e->x[i], B, g.H, g.W, ACT_NONE, nullptr, 1, s, &f));
    } else {
      RUN(conv3x3(e, e->lat[i], e->fpn_w[i], e->fpn_b[i], e->x[i], B, g.H, g.W, ACT_NONE, nullptr, 1, s));
    }
  }
  // RPN head (mmdet/models/dense_heads/rpn_head.py:62-68).  The RPN branch (conv + 1x1 heads here, proposal selection and
  // NMS in run_roi_path) and the semantic branch below both depend only on the FPN maps: the RPN branch runs on the side
  // stream from here on, so the tails of either branch's launches are filled by the other's blocks; joined before build_rois.
  // (throughput schedule: the branch stays on the caller's stream, the fork below is then a no-op between a stream and itself)
  hipStream_t s2 = e->cfg.schedule == NUHTC_SCHED_THROUGHPUT ? s : e->side;
  if (s2 != s && (hipEventRecord(e->ev_fpn, s) != hipSuccess || hipStreamWaitEvent(s2, e->ev_fpn, 0) != hipSuccess))
    FAIL(e, NUHTC_E_HIP, "side-stream fork failed");
  // the RPN head shares its weights across the levels (rpn_head.py:62-68 runs forward_single per level with the same modules): the four
  // maps go through ONE launch of the fused conv + cls/reg kernel -- the tiles of levels 1-3 (a third of level 0's) fill the tail of
  // level 0's persistent grid instead of three launches of 32-512 tiles on 256 CUs (dev knob RPN_ONE_LAUNCH=0: one launch per level)
  static const int& rpn_one = dev_knob_ref("RPN_ONE_LAUNCH", 1);
  if (fuse && rpn_one) {
    Conv3Fuse f = pointwise(32, e->rpn_hf, e->rpn_hb, e->rpn[0], ACT_NONE, 0);
    f.n_more = 3;
    for (int i = 1; i < 4; ++i) { f.more_in[i - 1] = e->x[i]; f.more_out2[i - 1] = e->rpn[i]; f.more_H[i - 1] = e->st[i].H; f.more_W[i - 1] = e->st[i].W; }
    RUN(conv3x3(e, e->x[0], e->rpn_w, e->rpn_b, e->tmpR, B, e->st[0].H, e->st[0].W, ACT_RELU, nullptr, 1, s2, &f));
  }
  for (int i = 0; i < 4 && !(fuse && rpn_one); ++i) {
    const StageGeom& g = e->st[i];
    if (fuse) {
      const Conv3Fuse f = pointwise(32, e->rpn_hf, e->rpn_hb, e->rpn[i], ACT_NONE, 0);
      RUN(conv3x3(e, e->x[i], e->rpn_w, e->rpn_b, e->tmpR, B, g.H, g.W, ACT_RELU, nullptr, 1, s2, &f));
    } else {
      RUN(conv3x3(e, e->x[i], e->rpn_w, e->rpn_b, e->tmpR, B, g.H, g.W, ACT_RELU, nullptr, 1, s2));
      RUN(egemm(e, gp(e->tmpR, e->rpn_hw, e->rpn_hb, e->rpn[i], B * g.H * g.W, 32, 64), s2));
    }
  }
  if (s2 != s && hipEventRecord(e->ev_rpn, s2) != hipSuccess) FAIL(e, NUHTC_E_HIP, "hipEventRecord failed");   // RPN maps ready (side stream)
  // FusedSemanticHead (fused_semantic_head.py:97-111)
  if (!fuse)
    for (int i = 0; i < 4; ++i) {
      const StageGeom& g = e->st[i];
      RUN(egemm(e, gp(e->x[i], e->sem_lw[i], e->sem_lb[i], e->semg[i], B * g.H * g.W, 64, 64), s));
    }
  const StageGeom& g0 = e->st[0];
  RUN(launch_sem_fuse(e->semg[0], e->semg[1], e->semg[2], e->semg[3], e->tmpA, B, g0.H, g0.W, s));
  float* a = e->tmpA;
  float* b = e->tmpB;
  for (int j = 0; j < 4; ++j) {
    if (fuse && j == 3) {
      // conv_logits (64 -> 1), conv_embedding (+ ReLU) and x0 + sem for the 7x7 RoI features, all from the tile in registers
      Conv3Fuse f = pointwise(64, e->sem_ef, e->sem_eb, e->sem_feat, ACT_RELU, 0);
      f.res2 = e->x[0]; f.out3 = e->x0sem;
      f.wn1 = e->sem_gw; f.bn1 = e->sem_gb; f.outn1 = e->sem_pred;
      RUN(conv3x3(e, a, e->sem_cw[j], e->sem_cb[j], b, B, g0.H, g0.W, ACT_RELU, nullptr, 1, s, &f));
      return 0;
    }
    RUN(conv3x3(e, a, e->sem_cw[j], e->sem_cb[j], b, B, g0.H, g0.W, ACT_RELU, nullptr, 1, s));
    std::swap(a, b);
  }
  RUN(launch_conv1x1_n1(a, e->sem_gw, e->sem_gb, e->sem_pred, B * g0.H * g0.W, 64, s));
  {
    GemmParams p = gp(a, e->sem_ew, e->sem_eb, e->sem_feat, B * g0.H * g0.W, 64, 64);
    p.act = ACT_RELU;
    RUN(egemm(e, p, s));
    // x0 + sem for the 7x7 RoI features (roi.hip: one interpolation serves the FPN level-0 and the semantic term)
    p.C = e->x0sem; p.res = e->x[0]; p.ldr = 64;
    RUN(egemm(e, p, s));
  }
  return 0;
}

// =============================================================================== public entry points
static int check_infer_args(nuhtc_engine* e, const uint8_t* tiles, int B) {
  if (!e) return NUHTC_E_INVALID;
  if (!e->finalized) FAIL(e, NUHTC_E_STATE, "nuhtc_infer before nuhtc_finalize");
  if (!tiles || B < 1 || B > e->cfg.max_batch) FAIL(e, NUHTC_E_INVALID, "bad tiles pointer or batch size (1..max_batch)");
  return 0;
}

// one step; a step that fails forgets the caller's tile pointer (nuhtc_get_buffer("img") reads it again: include/nuhtc_hip.h)
static int run_step(nuhtc_engine* e, int B, const float* rois, int n_rois, int n_dets, hipStream_t s, const nuhtc_dets* out) {
  int rc = run_backbone(e, B, s);
  if (!rc) rc = run_neck_heads(e, B, s);
  if (!rc) rc = run_roi_path(e, B, rois, n_rois, n_dets, s, out);
  if (rc) e->in_tiles = nullptr;
  return rc;
}

int nuhtc_infer(nuhtc_engine* e, const uint8_t* tiles, int B, int channel_mode, void* stream, const nuhtc_dets* out) {
  int rc = check_infer_args(e, tiles, B);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  HIP_CHECK(e, hipSetDevice(e->device));
  e->lastB = B;
  e->in_tiles = tiles; e->in_swap = channel_mode == NUHTC_CH_SWAP;      // the backbone's first launch reads the tiles (swin.hip patch_embed_tiles_kernel)
  return run_step(e, B, nullptr, 0, 0, s, out);
}

int nuhtc_infer_fixed_load(nuhtc_engine* e, const uint8_t* tiles, int B, int channel_mode, const float* rois, int n_rois, int n_dets,
                           void* stream, const nuhtc_dets* out) {
  int rc = check_infer_args(e, tiles, B);
  if (rc) return rc;
  if (!rois || n_rois < 1 || n_rois > e->roi_cap || n_dets < 1 || n_dets > e->cfg.max_per_img) FAIL(e, NUHTC_E_INVALID, "bad fixed-load arguments");
  hipStream_t s = (hipStream_t)stream;
  HIP_CHECK(e, hipSetDevice(e->device));
  e->lastB = B;
  e->in_tiles = tiles; e->in_swap = channel_mode == NUHTC_CH_SWAP;      // the backbone's first launch reads the tiles (swin.hip patch_embed_tiles_kernel)
  return run_step(e, B, rois, n_rois, n_dets, s, out);
}

int nuhtc_mask_contours(nuhtc_engine* e, const nuhtc_dets* dets, int B, int cap, int16_t* xy, int32_t* n, void* stream) {
  if (!e) return NUHTC_E_INVALID;
  if (!dets || !dets->masks || !dets->counts || !xy || !n || B < 1 || B > e->cfg.max_batch || cap < 1)
    FAIL(e, NUHTC_E_INVALID, "bad nuhtc_mask_contours arguments");
  HIP_CHECK(e, hipSetDevice(e->device));
  int rc = launch_contours(dets->masks, dets->keep, dets->counts, B, e->cfg.max_per_img, e->cfg.tile_h, e->cfg.tile_w, cap, xy, n,
                           (hipStream_t)stream);
  if (rc) FAIL(e, rc, "contour launch failed (tile width must be a multiple of 32)");
  return 0;
}


int nuhtc_export_kept(nuhtc_engine* e, const nuhtc_dets* dets, int B, const int32_t* contour_n, const int16_t* contour_xy, int contour_cap, int cap,
                      int32_t* n_dev, int64_t* idx_dev, float* boxes_dev, int32_t* labels_dev, int32_t* cn_dev, int16_t* xy_dev, uint32_t* words_dev,
                      void* stream) {
  if (!e) return NUHTC_E_INVALID;
  if (!dets || !dets->boxes || !dets->labels || !dets->counts || !dets->keep || !dets->masks || B < 1 || B > e->cfg.max_batch || cap < 1 ||
      !n_dev || !idx_dev || !boxes_dev || !labels_dev || !cn_dev || !words_dev || (contour_xy && (!xy_dev || contour_cap < 1)))
    FAIL(e, NUHTC_E_INVALID, "bad nuhtc_export_kept arguments");
  HIP_CHECK(e, hipSetDevice(e->device));
  if (!e->export_pos) {
    int rc = dev_alloc(e, (void**)&e->export_pos, (size_t)e->cfg.max_batch * e->cfg.max_per_img * sizeof(int32_t));
    if (rc) return rc;
  }
  ExportParams p{dets->boxes, dets->labels, dets->counts, dets->keep, dets->masks, contour_n, contour_xy, B, e->cfg.max_per_img,
                 e->cfg.tile_h * (e->cfg.tile_w / 32), contour_cap, cap, n_dev, idx_dev, boxes_dev, labels_dev, cn_dev, xy_dev, words_dev};
  int rc = launch_export_kept(p, e->export_pos, (hipStream_t)stream);
  if (rc) FAIL(e, rc, "export launch failed");
  // n_dev[1]: the capacity flag nuhtc_check reports, so that a caller that fetches results through this export needs no second
  // round trip (and may have the engine's next batch enqueued already, which resets the flag)
  if (e->overflow) HIP_CHECK(e, hipMemcpyAsync(n_dev + 1, e->overflow, sizeof(int32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  else HIP_CHECK(e, hipMemsetAsync(n_dev + 1, 0, sizeof(int32_t), (hipStream_t)stream));
  return 0;
}

int nuhtc_export_crops(nuhtc_engine* e, const uint32_t* words_dev, const int32_t* n_dev, int cap, int32_t* crop_box_dev, int32_t* crop_area_dev,
                       int32_t* crop_off_dev, uint32_t* crop_words_dev, int pool_cap, void* stream) {
  if (!e) return NUHTC_E_INVALID;
  if (!words_dev || !n_dev || cap < 1 || !crop_box_dev || !crop_area_dev || !crop_off_dev || !crop_words_dev || pool_cap < 1)
    FAIL(e, NUHTC_E_INVALID, "bad nuhtc_export_crops arguments");
  HIP_CHECK(e, hipSetDevice(e->device));
  const int max_cap = e->cfg.max_batch * e->cfg.max_per_img;
  if (cap > max_cap) FAIL(e, NUHTC_E_INVALID, "nuhtc_export_crops: cap exceeds max_batch * max_per_img");
  if (!e->crop_size) {
    int rc = dev_alloc(e, (void**)&e->crop_size, (size_t)max_cap * sizeof(int32_t));
    if (rc) return rc;
  }
  int rc = launch_export_crops(words_dev, n_dev, cap, e->cfg.tile_h, e->cfg.tile_w / 32, crop_box_dev, crop_area_dev, crop_off_dev, e->crop_size, crop_words_dev,
                               pool_cap, (hipStream_t)stream);
  if (rc) FAIL(e, rc, "crop export launch failed");
  return 0;
}

int nuhtc_check(nuhtc_engine* e, void* stream) {
  if (!e) return NUHTC_E_INVALID;
  HIP_CHECK(e, hipSetDevice(e->device));
  if (e->overflow && !e->overflow_host) HIP_CHECK(e, hipHostMalloc((void**)&e->overflow_host, 4 * sizeof(int), hipHostMallocDefault));
  if (e->overflow) HIP_CHECK(e, hipMemcpyAsync(e->overflow_host, e->overflow, 4 * sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_CHECK(e, hipStreamSynchronize((hipStream_t)stream));
  if (e->overflow && e->overflow_host[0]) FAIL(e, NUHTC_E_CAPACITY, "connected-component proposals exceeded max_cc_proposals on at least one tile");
  return 0;
}

int nuhtc_get_buffer(nuhtc_engine* e, const char* name, void** ptr, int64_t* shape, int* ndim, int* dtype) {
  if (!e || !name || !ptr) return NUHTC_E_INVALID;
  if (strcmp(name, "__enable_token_dump") == 0) {
    // allocate per-block token snapshots (parity tests only)
    if (!e->debug_tokens) {
      for (int s = 0; s < 4; ++s)
        for (int b = 0; b < DEPTHS[s]; ++b) {
          float* p;
          const StageGeom& g = e->st[s];
          int rc = ws(e, &p, ("tok_s" + std::to_string(s) + "b" + std::to_string(b)).c_str(), {e->cfg.max_batch, g.H * g.W, g.C}, 0);
          if (rc) return rc;
        }
      e->debug_tokens = true;
    }
    *ptr = nullptr;
    if (ndim) *ndim = 0;
    return 0;
  }
  auto it = e->bufs.find(name);
  if (it == e->bufs.end()) FAIL(e, NUHTC_E_NOTFOUND, std::string("unknown buffer: ") + name);
  if (e->out_ln_folded && name[0] == 'c' && name[1] >= '0' && name[1] <= '3' && name[2] == 0) {
    // the stage's output norm ran inside the FPN lateral: the tensor is computed now, from the stage's tokens, by the kernel that writes it on the
    // other path (parity tests read c0..c3)
    const int st = name[1] - '0';
    const StageGeom& g = e->st[st];
    HIP_CHECK(e, hipSetDevice(e->device));
    HIP_CHECK(e, hipDeviceSynchronize());
    const int rc = launch_layernorm(e->tok[st], nullptr, e->on_g[st], e->on_b[st], e->c[st], e->last_batch * g.H * g.W, g.C, nullptr);
    if (rc) FAIL(e, rc, "layernorm for a requested c buffer failed");
    HIP_CHECK(e, hipDeviceSynchronize());
  }
  if (e->img_stale && strcmp(name, "img") == 0 && e->in_tiles) {
    // the pre-processing ran inside the patch embedding: the normalised image is computed now, by the kernel that writes it on the other path, from
    // the tiles of the last call (which must still be what they were: parity tests read `img` right after the call)
    float mi[6];
    for (int i = 0; i < 3; ++i) { mi[i] = e->cfg.mean[i]; mi[3 + i] = (float)(1.0 / (double)e->cfg.std[i]); }
    HIP_CHECK(e, hipSetDevice(e->device));
    HIP_CHECK(e, hipDeviceSynchronize());
    const int rc = launch_preproc(e->in_tiles, e->img, e->last_batch, e->cfg.tile_h, e->cfg.tile_w, e->Hn, e->Wn, e->Hv, e->Wv, e->rs_xtab, e->rs_ytab, e->in_swap, mi, nullptr);
    if (rc) FAIL(e, rc, "preproc for the requested img buffer failed");
    HIP_CHECK(e, hipDeviceSynchronize());
  }
  *ptr = it->second.ptr;
  if (ndim) *ndim = (int)it->second.shape.size();
  if (shape)
    for (size_t i = 0; i < it->second.shape.size() && i < 6; ++i) shape[i] = it->second.shape[i];
  if (dtype) *dtype = it->second.dtype;
  return 0;
}

int nuhtc_op_gemm(nuhtc_engine* e, const float* A, const float* W, const float* bias, float* C, int M, int N, int K, int act, void* stream) {
  if (!e || !A || !W || !C) return NUHTC_E_INVALID;
  HIP_CHECK(e, hipSetDevice(e->device));
  GemmParams p = gp(A, W, bias, C, M, N, K);
  p.act = act;
  int rc = launch_gemm(p, (hipStream_t)stream);
  if (rc) FAIL(e, rc, "gemm launch failed (K%32, N%32 required)");
  return 0;
}

int nuhtc_op_gemm_split(nuhtc_engine* e, const float* A, const float* W_dev, const float* W_host, const float* bias, float* C, int M, int N,
                        int K, int act, void* stream) {
  if (!e || !A || !W_dev || !W_host || !C) return NUHTC_E_INVALID;
  HIP_CHECK(e, hipSetDevice(e->device));
  void* sp = nullptr;              // a private split of this call's weight: the registry of the engines' weights is not touched
  int rc = gemm_make_split(W_host, N, K, &sp);
  if (rc) FAIL(e, rc, "gemm_make_split failed (K % 8)");
  GemmParams p = gp(A, W_dev, bias, C, M, N, K);
  p.act = act;
  p.Wsplit = sp;
  rc = launch_gemm(p, (hipStream_t)stream);
  hipError_t he = hipStreamSynchronize((hipStream_t)stream);
  hipFree(sp);
  if (rc) FAIL(e, rc, "gemm launch failed (K%32, N%32 required)");
  if (he != hipSuccess) FAIL(e, NUHTC_E_HIP, "gemm kernel failed");
  return 0;
}

static int fold_ln(const float* W_host, const float* bias_host, const float* g, const float* b, int N, int K, std::vector<float>& w2, std::vector<float>& b2) {
  w2.resize((size_t)N * K); b2.resize(N);
  for (int n = 0; n < N; ++n) {
    double acc = bias_host ? bias_host[n] : 0.0;
    for (int k = 0; k < K; ++k) {
      w2[(size_t)n * K + k] = W_host[(size_t)n * K + k] * g[k];
      acc += (double)W_host[(size_t)n * K + k] * (double)b[k];
    }
    b2[n] = (float)acc;
  }
  return 0;
}

int nuhtc_op_ln_gemm(nuhtc_engine* e, const float* X_dev, int T, const int* rows_dev, const float* W_host, const float* bias_host, const float* ln_g_host,
                     const float* ln_b_host, float* C_dev, int M, int N, int K, int act, void* stream) {
  if (!e || !X_dev || !W_host || !ln_g_host || !ln_b_host || !C_dev || M < 1 || N < 1 || K < 1 || T < 1) return NUHTC_E_INVALID;
  HIP_CHECK(e, hipSetDevice(e->device));
  std::vector<float> w2, b2;
  fold_ln(W_host, bias_host, ln_g_host, ln_b_host, N, K, w2, b2);
  void* sp = nullptr;
  int rc = gemm_make_split(w2.data(), N, K, &sp);
  if (rc) FAIL(e, rc, "gemm_make_split failed (K % 8)");
  float *wd = nullptr, *bd = nullptr, *st = nullptr;
  hipError_t he = hipMalloc(&wd, w2.size() * 4);
  if (he == hipSuccess) he = hipMalloc(&bd, b2.size() * 4);
  if (he == hipSuccess) he = hipMalloc(&st, (size_t)T * 8);
  if (he == hipSuccess) he = hipMemcpy(wd, w2.data(), w2.size() * 4, hipMemcpyHostToDevice);
  if (he == hipSuccess) he = hipMemcpy(bd, b2.data(), b2.size() * 4, hipMemcpyHostToDevice);
  if (he == hipSuccess) {
    rc = launch_ln_stats(X_dev, st, T, K, (hipStream_t)stream);
    if (!rc) {
      GemmParams p = gp(X_dev, wd, bd, C_dev, M, N, K);
      p.act = act; p.Wsplit = sp; p.amode = A_LN; p.ln_part = st; p.ln_nparts = 1; p.a_rows = rows_dev;
      rc = launch_gemm(p, (hipStream_t)stream);
    }
    he = hipStreamSynchronize((hipStream_t)stream);
  }
  hipFree(sp); hipFree(wd); hipFree(bd); hipFree(st);
  if (rc) FAIL(e, rc, "ln_gemm launch failed (N % 96, K % 32 required)");
  if (he != hipSuccess) FAIL(e, NUHTC_E_HIP, "ln_gemm failed");
  return 0;
}

int nuhtc_op_gemm_ln_gemm(nuhtc_engine* e, const float* A_dev, const float* Wp_host, const float* bp_host, const float* res_dev, const int* row_map_dev,
                          const float* W_host, const float* bias_host, const float* ln_g_host, const float* ln_b_host, float* Y_dev, float* C_dev, int M,
                          int Kp, int K, int N, int act, void* stream) {
  if (!e || !A_dev || !Wp_host || !W_host || !ln_g_host || !ln_b_host || !Y_dev || !C_dev || M < 1 || N < 1 || K < 1 || Kp < 1 || K % 96) return NUHTC_E_INVALID;
  HIP_CHECK(e, hipSetDevice(e->device));
  std::vector<float> w2, b2;
  fold_ln(W_host, bias_host, ln_g_host, ln_b_host, N, K, w2, b2);
  void *sp = nullptr, *spp = nullptr;
  int rc = gemm_make_split(w2.data(), N, K, &sp);
  if (!rc) rc = gemm_make_split(Wp_host, K, Kp, &spp);
  if (rc) { hipFree(sp); FAIL(e, rc, "gemm_make_split failed (K % 8)"); }
  float *wd = nullptr, *bd = nullptr, *st = nullptr, *wpd = nullptr, *bpd = nullptr;
  hipError_t he = hipMalloc(&wd, w2.size() * 4);
  if (he == hipSuccess) he = hipMalloc(&bd, b2.size() * 4);
  if (he == hipSuccess) he = hipMalloc(&st, (size_t)M * (K / 96) * 8);
  if (he == hipSuccess) he = hipMalloc(&wpd, (size_t)K * Kp * 4);
  if (he == hipSuccess && bp_host) he = hipMalloc(&bpd, (size_t)K * 4);
  if (he == hipSuccess) he = hipMemcpy(wd, w2.data(), w2.size() * 4, hipMemcpyHostToDevice);
  if (he == hipSuccess) he = hipMemcpy(bd, b2.data(), b2.size() * 4, hipMemcpyHostToDevice);
  if (he == hipSuccess) he = hipMemcpy(wpd, Wp_host, (size_t)K * Kp * 4, hipMemcpyHostToDevice);
  if (he == hipSuccess && bp_host) he = hipMemcpy(bpd, bp_host, (size_t)K * 4, hipMemcpyHostToDevice);
  if (he == hipSuccess) {
    GemmParams q = gp(A_dev, wpd, bpd, Y_dev, M, K, Kp);              // the producer: Y[row_map(m)] = A Wp^T + bp (+ res), statistics on the way out
    q.Wsplit = spp; q.stats_out = st;
    if (res_dev) { q.res = res_dev; q.ldr = K; }
    if (row_map_dev) { q.store = ST_ROWMAP; q.row_map = row_map_dev; }
    rc = launch_gemm(q, (hipStream_t)stream);
    if (!rc) {
      GemmParams p = gp(Y_dev, wd, bd, C_dev, M, N, K);
      p.act = act; p.Wsplit = sp; p.amode = A_LN; p.ln_part = st; p.ln_nparts = K / 96;
      rc = launch_gemm(p, (hipStream_t)stream);
    }
    he = hipStreamSynchronize((hipStream_t)stream);
  }
  hipFree(sp); hipFree(spp); hipFree(wd); hipFree(bd); hipFree(st); hipFree(wpd); hipFree(bpd);
  if (rc) FAIL(e, rc, "gemm_ln_gemm launch failed");
  if (he != hipSuccess) FAIL(e, NUHTC_E_HIP, "gemm_ln_gemm failed");
  return 0;
}

int nuhtc_op_merge_ln_gemm(nuhtc_engine* e, const float* X_dev, int B, int H, int W, int C, const float* W_host, const float* ln_g_host, const float* ln_b_host,
                           float* Y_dev, void* stream) {
  if (!e || !X_dev || !W_host || !ln_g_host || !ln_b_host || !Y_dev || B < 1 || H < 2 || W < 2 || (H & 1) || (W & 1) || C < 96 || C % 96) return NUHTC_E_INVALID;
  HIP_CHECK(e, hipSetDevice(e->device));
  const int T = B * H * W, M = T / 4;
  std::vector<float> g2(4 * C), b2(4 * C), w2((size_t)2 * C * 4 * C), wl, bl;      // gather order k' = q*C + c, q = kh*2 + kw (nuhtc_finalize)
  for (int q = 0; q < 4; ++q)
    for (int ch = 0; ch < C; ++ch) {
      g2[q * C + ch] = ln_g_host[ch * 4 + q];
      b2[q * C + ch] = ln_b_host[ch * 4 + q];
      for (int n = 0; n < 2 * C; ++n) w2[(size_t)n * 4 * C + q * C + ch] = W_host[(size_t)n * 4 * C + ch * 4 + q];
    }
  fold_ln(w2.data(), nullptr, g2.data(), b2.data(), 2 * C, 4 * C, wl, bl);
  std::vector<int> src(M);
  for (int b = 0; b < B; ++b)
    for (int y2 = 0; y2 < H / 2; ++y2)
      for (int x2 = 0; x2 < W / 2; ++x2) src[((size_t)b * (H / 2) + y2) * (W / 2) + x2] = (b * H + 2 * y2) * W + 2 * x2;
  void* sp = nullptr;
  int rc = gemm_make_split(wl.data(), 2 * C, 4 * C, &sp);
  if (rc) FAIL(e, rc, "gemm_make_split failed");
  float *wd = nullptr, *bd = nullptr, *st = nullptr;
  int* sd = nullptr;
  hipError_t he = hipMalloc(&wd, wl.size() * 4);
  if (he == hipSuccess) he = hipMalloc(&bd, bl.size() * 4);
  if (he == hipSuccess) he = hipMalloc(&st, (size_t)T * (C / 96) * 8);
  if (he == hipSuccess) he = hipMalloc(&sd, (size_t)M * 4);
  if (he == hipSuccess) he = hipMemcpy(wd, wl.data(), wl.size() * 4, hipMemcpyHostToDevice);
  if (he == hipSuccess) he = hipMemcpy(bd, bl.data(), bl.size() * 4, hipMemcpyHostToDevice);
  if (he == hipSuccess) he = hipMemcpy(sd, src.data(), (size_t)M * 4, hipMemcpyHostToDevice);
  if (he == hipSuccess) {
    rc = launch_ln_stats(X_dev, st, T * (C / 96), 96, (hipStream_t)stream);       // a partial per token and 96 channels: what the producers' epilogues leave
    if (!rc) {
      GemmParams p = gp(X_dev, wd, bd, Y_dev, M, 2 * C, 4 * C);
      p.Wsplit = sp; p.lda = C; p.amode = A_LN; p.ln_part = st; p.ln_nparts = 4 * (C / 96); p.a_rows = sd; p.seg_k = 2 * C; p.seg_rows = W;
      rc = launch_gemm(p, (hipStream_t)stream);
    }
    he = hipStreamSynchronize((hipStream_t)stream);
  }
  hipFree(sp); hipFree(wd); hipFree(bd); hipFree(st); hipFree(sd);
  if (rc) FAIL(e, rc, "merge_ln_gemm launch failed");
  if (he != hipSuccess) FAIL(e, NUHTC_E_HIP, "merge_ln_gemm failed");
  return 0;
}

int nuhtc_op_swin_mlp(nuhtc_engine* e, const float* x_dev, const float* ln_g_dev, const float* ln_b_dev, const float* w1_host, const float* b1_dev,
                      const float* w2_host, const float* b2_dev, float* out_dev, int T, int C, void* stream) {
  if (!e || !x_dev || !ln_g_dev || !ln_b_dev || !w1_host || !b1_dev || !w2_host || !b2_dev || !out_dev || T < 1) return NUHTC_E_INVALID;
  if (!mlp_supported(C)) FAIL(e, NUHTC_E_INVALID, "nuhtc_op_swin_mlp: unsupported channel count");
  HIP_CHECK(e, hipSetDevice(e->device));
  std::vector<unsigned short> st;
  mlp_pack_stream(w1_host, w2_host, C, st);
  void* d = nullptr;
  HIP_CHECK(e, hipMalloc(&d, st.size() * 2));
  if (hipMemcpy(d, st.data(), st.size() * 2, hipMemcpyHostToDevice) != hipSuccess) { hipFree(d); FAIL(e, NUHTC_E_HIP, "weight stream upload failed"); }
  int rc = launch_swin_mlp(x_dev, out_dev, ln_g_dev, ln_b_dev, d, b1_dev, b2_dev, T, C, (hipStream_t)stream);
  hipError_t he = hipStreamSynchronize((hipStream_t)stream);
  hipFree(d);
  if (rc) FAIL(e, rc, "swin_mlp launch failed");
  if (he != hipSuccess) FAIL(e, NUHTC_E_HIP, "swin_mlp kernel failed");
  return 0;
}

int nuhtc_op_swin_proj_mlp(nuhtc_engine* e, const float* x_dev, const float* att_dev, const float* wp_host, const float* bp_dev, const float* ln_g_dev,
                           const float* ln_b_dev, const float* w1_host, const float* b1_dev, const float* w2_host, const float* b2_dev, float* out_dev,
                           int T, int C, void* stream) {
  if (!e || !x_dev || !att_dev || !wp_host || !bp_dev || !ln_g_dev || !ln_b_dev || !w1_host || !b1_dev || !w2_host || !b2_dev || !out_dev || T < 1) return NUHTC_E_INVALID;
  if (!mlp_supported(C)) FAIL(e, NUHTC_E_INVALID, "nuhtc_op_swin_proj_mlp: unsupported channel count");
  HIP_CHECK(e, hipSetDevice(e->device));
  std::vector<unsigned short> st, sp;
  mlp_pack_stream(w1_host, w2_host, C, st);
  proj_pack_stream(wp_host, C, sp);
  void *d = nullptr, *dp = nullptr;
  HIP_CHECK(e, hipMalloc(&d, st.size() * 2));
  if (hipMalloc(&dp, sp.size() * 2) != hipSuccess) { hipFree(d); FAIL(e, NUHTC_E_HIP, "hipMalloc failed"); }
  bool ok = hipMemcpy(d, st.data(), st.size() * 2, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(dp, sp.data(), sp.size() * 2, hipMemcpyHostToDevice) == hipSuccess;
  // the kernel works in place (its FFN residual re-reads x' where the projection stored it): out <- x first
  if (ok && out_dev != x_dev) ok = hipMemcpyAsync(out_dev, x_dev, (size_t)T * C * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess;
  int rc = ok ? launch_swin_mlp(out_dev, out_dev, ln_g_dev, ln_b_dev, d, b1_dev, b2_dev, T, C, (hipStream_t)stream, att_dev, dp, bp_dev) : NUHTC_E_HIP;
  hipError_t he = hipStreamSynchronize((hipStream_t)stream);
  hipFree(d);
  hipFree(dp);
  if (rc) FAIL(e, rc, "swin_proj_mlp launch failed");
  if (he != hipSuccess) FAIL(e, NUHTC_E_HIP, "swin_proj_mlp kernel failed");
  return 0;
}

#ifdef NUHTC_DEV
// dev (tools/dev/r04_state_buffers.py): gives one of the Swin activation buffers new memory (the old allocation stays until nuhtc_destroy, so the
// new one lands elsewhere).  which: 0 tokA, 1 tokB, 2 xw, 3 qkv, 4 att, 5 hid; returns the new device address through *addr
extern "C" int nuhtc_dev_realloc(nuhtc_engine* e, int which, unsigned long long* addr) {
  if (!e || which < 0 || which > 5) return NUHTC_E_INVALID;
  HIP_CHECK(e, hipSetDevice(e->device));
  HIP_CHECK(e, hipDeviceSynchronize());
  float** slots[6] = {&e->tokA, &e->tokB, &e->xw, &e->qkv, &e->att, &e->hid};
  void* base = nullptr;
  size_t bytes = 0;
  HIP_CHECK(e, hipMemGetAddressRange((hipDeviceptr_t*)&base, &bytes, (hipDeviceptr_t)*slots[which]));
  void* p = nullptr;
  HIP_CHECK(e, hipMalloc(&p, bytes));
  HIP_CHECK(e, hipMemset(p, 0, bytes));
  e->allocs.push_back(p);
  *slots[which] = (float*)p;
  if (which == 0) e->bufs["tokens"].ptr = p;
  e->tok[0] = e->tokA; e->tok[1] = e->tokB;
  if (addr) *addr = (unsigned long long)p;
  return 0;
}
#endif
