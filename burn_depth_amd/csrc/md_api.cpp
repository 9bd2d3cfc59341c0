// extern "C" surface of libmi_depth.so (include/mi_depth.h).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "md_engine.h"

using namespace md;

namespace {
struct DevBuf {  // scoped device allocation for the stand-alone test operators
  void* p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  int alloc(size_t bytes) {
    if (hipMalloc(&p, bytes + 4096) != hipSuccess) {
      set_error("hipMalloc(%zu) failed", bytes);
      return MD_ERR_OOM;
    }
    return hipMemset(p, 0, bytes + 4096) == hipSuccess ? MD_OK : MD_ERR_HIP;
  }
};
hipStream_t pick_stream(md_device_t dev, void* stream) { return stream ? (hipStream_t)stream : dev->stream; }
int ke_of(int prec) { return prec == MD_PREC_F32 ? 32 : (prec == MD_PREC_FP8 ? 128 : 64); }
// MD_PREC_F16X2: MFMA terms of a product with these weights -- 2 when every value is an exact IEEE half, else 3
int split_terms_of(const float* w_dev, long n, hipStream_t st, int* terms) {
  unsigned bad = 0;
  MD_TRY(count_inexact_f16(w_dev, n, st, &bad));
  *terms = bad == 0 ? 2 : 3;
  return MD_OK;
}
size_t esz_of(int prec) { return prec == MD_PREC_F32 ? 4 : (prec == MD_PREC_FP8 ? 1 : 2); }
}  // namespace

extern "C" {

const char* md_last_error(void) { return get_error(); }
const char* md_version(void) { return "mi_depth 0.1 (gfx950)"; }

int md_device_open(int hip_ordinal, md_device_t* out) {
  if (!out) MD_FAIL(MD_ERR_INVALID_ARG, "out is null");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) MD_FAIL(MD_ERR_HIP, "no HIP device available");
  if (hip_ordinal < 0 || hip_ordinal >= n) MD_FAIL(MD_ERR_INVALID_ARG, "device ordinal %d out of range [0,%d)", hip_ordinal, n);
  MD_HIP(hipSetDevice(hip_ordinal));
  md_device_s* d = new md_device_s();
  d->ordinal = hip_ordinal;
  // a BLOCKING stream: implicitly ordered with the legacy default stream, so a caller that hands over
  // buffers produced on the null stream (and passes stream = NULL) needs no extra synchronisation
  if (hipStreamCreateWithFlags(&d->stream, hipStreamDefault) != hipSuccess) {
    delete d;
    MD_FAIL(MD_ERR_HIP, "hipStreamCreate failed");
  }
  *out = d;
  return MD_OK;
}

int md_device_close(md_device_t dev) {
  if (!dev) return MD_OK;
  (void)hipSetDevice(dev->ordinal);
  if (dev->stream) (void)hipStreamDestroy(dev->stream);
  delete dev;
  return MD_OK;
}

int md_device_synchronize(md_device_t dev) {
  if (!dev) MD_FAIL(MD_ERR_INVALID_ARG, "device is null");
  MD_HIP(hipSetDevice(dev->ordinal));
  MD_HIP(hipDeviceSynchronize());
  return MD_OK;
}

void md_depth_pro_cfg_default(md_depth_pro_cfg* cfg) {
  if (!cfg) return;
  cfg->patch_encoder_preset = "dinov2l16_384";
  cfg->image_encoder_preset = "dinov2l16_384";
  cfg->fov_encoder_preset = "dinov2l16_384";
  cfg->decoder_features = 256;
  cfg->use_fov_head = 1;
  cfg->interpolation = MD_INTERP_CUSTOM;
  cfg->precision = MD_PREC_BF16;
  cfg->max_batch = 1;
  cfg->ln_eps = 1e-6f;
}

int md_depth_pro_create(md_device_t dev, const md_depth_pro_cfg* cfg, uint64_t seed, int init_scheme, md_model_t* out) {
  if (!dev || !out) MD_FAIL(MD_ERR_INVALID_ARG, "device/out is null");
  ModelCfg mc;
  MD_TRY(parse_cfg(cfg, &mc));
  md_model_t m = nullptr;
  MD_TRY(model_create(dev, mc, &m));
  int s = model_init_seeded(m, seed, init_scheme);
  if (s != MD_OK) {
    model_destroy(m);
    return s;
  }
  *out = m;
  return MD_OK;
}

int md_depth_pro_load_with_config(md_device_t dev, const md_depth_pro_cfg* cfg, const char* path, md_model_t* out) {
  if (!dev || !out) MD_FAIL(MD_ERR_INVALID_ARG, "device/out is null");
  ModelCfg mc;
  MD_TRY(parse_cfg(cfg, &mc));
  // fail on I/O problems before touching the GPU (RecorderError path, mod.rs:193-208)
  {
    Container probe;
    MD_TRY(read_container(path, &probe));
  }
  md_model_t m = nullptr;
  MD_TRY(model_create(dev, mc, &m));
  int s = model_load_container(m, path);
  if (s != MD_OK) {
    model_destroy(m);
    return s;
  }
  *out = m;
  return MD_OK;
}

int md_depth_pro_load(md_device_t dev, const char* path, md_model_t* out) {
  md_depth_pro_cfg c;
  md_depth_pro_cfg_default(&c);
  return md_depth_pro_load_with_config(dev, &c, path, out);
}

int md_checkpoint_info(const char* path, int index, const char** name, const char** dtype, int* rank, int64_t shape[8],
                       int* is_burn_record) {
  static thread_local Container c;
  static thread_local std::string loaded;
  static thread_local std::vector<std::string> names;
  if (!path) MD_FAIL(MD_ERR_INVALID_ARG, "checkpoint path is null");
  if (loaded != path || index < 0) {  // (re)read on a new path and on every count query
    c = Container();
    loaded.clear();
    MD_TRY(read_container(path, &c));
    loaded = path;
    names.clear();
    for (auto& kv : c.tensors) names.push_back(kv.first);
    c.bytes.clear();  // only the directory is kept
    c.bytes.shrink_to_fit();
  }
  if (is_burn_record) *is_burn_record = c.burn_record ? 1 : 0;
  if (index >= 0 && index < (int)names.size()) {
    const ContainerTensor& t = c.tensors[names[index]];
    if (t.shape.size() > 8) MD_FAIL(MD_ERR_FORMAT, "tensor `%s` has rank %zu", names[index].c_str(), t.shape.size());
    if (name) *name = names[index].c_str();
    if (dtype) *dtype = t.dtype.c_str();
    if (rank) *rank = (int)t.shape.size();
    if (shape)
      for (size_t i = 0; i < t.shape.size(); ++i) shape[i] = t.shape[i];
  }
  return (int)names.size();
}

int md_checkpoint_read_tensor(const char* path, const char* name, float* out_host, size_t count) {
  if (!path || !name || !out_host) MD_FAIL(MD_ERR_INVALID_ARG, "null argument");
  Container c;
  MD_TRY(read_container(path, &c));
  auto it = c.tensors.find(name);
  if (it == c.tensors.end()) MD_FAIL(MD_ERR_INVALID_ARG, "checkpoint `%s` has no tensor `%s`", path, name);
  size_t n = 1;
  for (auto d : it->second.shape) n *= (size_t)d;
  if (n != count) MD_FAIL(MD_ERR_SHAPE, "tensor `%s` has %zu elements, got %zu", name, n, count);
  return container_tensor_to_f32(c, it->second, out_host, count);
}

int md_model_param_count(md_model_t m) { return m ? (int)m->params.size() : 0; }

int md_model_param_info(md_model_t m, int index, const char** name, size_t* count) {
  if (!m || index < 0 || index >= (int)m->params.size()) MD_FAIL(MD_ERR_INVALID_ARG, "parameter index %d out of range", index);
  if (name) *name = m->params[index].name.c_str();
  if (count) *count = m->params[index].count();
  return MD_OK;
}

int md_model_set_tensor(md_model_t m, const char* name, const float* host_data, size_t count) {
  if (!m || !name || !host_data) MD_FAIL(MD_ERR_INVALID_ARG, "null argument");
  if (m->parent) MD_FAIL(MD_ERR_INVALID_ARG, "a fork shares its root's weights: set tensors on the root model");
  auto it = m->pindex.find(name);
  if (it == m->pindex.end()) MD_FAIL(MD_ERR_INVALID_ARG, "unknown parameter `%s`", name);
  if (m->params[it->second].count() != count)
    MD_FAIL(MD_ERR_SHAPE, "parameter `%s` has %zu elements, got %zu", name, m->params[it->second].count(), count);
  MD_HIP(hipSetDevice(m->dev->ordinal));
  MD_HIP(hipMemcpy(m->w32[it->second], host_data, count * 4, hipMemcpyHostToDevice));
  m->committed = false;
  return MD_OK;
}

int md_model_get_tensor(md_model_t m, const char* name, float* host_data, size_t count) {
  if (!m || !name || !host_data) MD_FAIL(MD_ERR_INVALID_ARG, "null argument");
  auto it = m->pindex.find(name);
  if (it == m->pindex.end()) MD_FAIL(MD_ERR_INVALID_ARG, "unknown parameter `%s`", name);
  if (m->params[it->second].count() != count)
    MD_FAIL(MD_ERR_SHAPE, "parameter `%s` has %zu elements, got %zu", name, m->params[it->second].count(), count);
  MD_HIP(hipSetDevice(m->dev->ordinal));
  MD_HIP(hipMemcpy(host_data, m->w32[it->second], count * 4, hipMemcpyDeviceToHost));
  return MD_OK;
}

int md_model_commit_weights(md_model_t m) {
  if (!m) MD_FAIL(MD_ERR_INVALID_ARG, "model is null");
  return model_commit(m);
}

int md_model_round_weights_f16(md_model_t m) { return model_round_weights_f16(m); }

int md_model_weight_arena(md_model_t m, void** device_ptr, size_t* bytes) {
  if (!m || !device_ptr || !bytes) MD_FAIL(MD_ERR_INVALID_ARG, "null argument");
  if (m->parent) MD_FAIL(MD_ERR_INVALID_ARG, "a fork shares its root's weights: take the arena of the root model");
  *device_ptr = m->w32_base;
  *bytes = m->w32_bytes;
  m->committed = false;  // the caller is about to overwrite it (broadcast); commit afterwards
  return MD_OK;
}

int md_model_destroy(md_model_t m) { return model_destroy(m); }

int md_model_fork(md_model_t m, md_model_t* out) { return model_fork(m, out); }

int md_depth_pro_infer(md_model_t m, const float* nchw, int B, int H, int W, int in_kind, float* depth,
                       float* focallength_px, float* fovx_deg, float* fovy_rad, int out_kind, void* stream) {
  if (!nchw) MD_FAIL(MD_ERR_INVALID_ARG, "input pointer is null");
  if (m && m->kind != 0) MD_FAIL(MD_ERR_INVALID_ARG, "not a Depth Pro model");
  return model_infer(m, nchw, B, H, W, in_kind, depth, focallength_px, fovx_deg, fovy_rad, out_kind, (hipStream_t)stream,
                     nullptr, 0);
}

int md_depth_pro_decoder_from_features(md_model_t m, const md_nchw_view* features, int levels, int B, int in_kind,
                                       float* out_features, float* out_lowres, float* const* out_fusions, int out_kind,
                                       void* stream) {
  return model_decoder_from_features(m, features, levels, B, in_kind, out_features, out_lowres, out_fusions, out_kind, (hipStream_t)stream);
}

int md_depth_pro_head_debug(md_model_t m, const md_nchw_view* feature, int B, int in_kind, const md_head_debug* out, int out_kind,
                            void* stream) {
  return model_head_debug(m, feature, B, in_kind, out, out_kind, (hipStream_t)stream);
}

int md_depth_pro_infer_windows(md_model_t m, const float* nchw, int B, int H, int W, int in_kind, float* depth,
                               float* focallength_px, float* fovx_deg, float* fovy_rad, int out_kind, int parts,
                               float* window_ms, float* tail_ms, void* stream) {
  if (!nchw) MD_FAIL(MD_ERR_INVALID_ARG, "input pointer is null");
  if (m && m->kind != 0) MD_FAIL(MD_ERR_INVALID_ARG, "not a Depth Pro model");
  if (parts < 1 || parts > 64) MD_FAIL(MD_ERR_INVALID_ARG, "parts = %d (1 .. 64)", parts);
  ShardPlan sp;
  sp.parts = parts;
  sp.part = -1;
  sp.window_ms = window_ms;
  sp.tail_ms = tail_ms;
  return model_infer_sharded(m, nchw, B, H, W, in_kind, depth, focallength_px, fovx_deg, fovy_rad, out_kind, (hipStream_t)stream, sp);
}

int md_infer_from_rgb(md_model_t m, const uint8_t* rgb, size_t rgb_len, int w, int h, int in_kind, float* depth,
                      float* focallength_px, float* fovy_rad, int out_kind, void* stream) {
  if (!rgb) MD_FAIL(MD_ERR_INVALID_ARG, "rgb pointer is null");
  if (w <= 0 || h <= 0) MD_FAIL(MD_ERR_SHAPE, "invalid image size %dx%d", w, h);
  // inference.rs:85-95: length check comes first and is an Err, not a panic
  const size_t expected = (size_t)w * (size_t)h * 3;
  if (rgb_len != expected) MD_FAIL(MD_ERR_SHAPE, "expected %zu RGB bytes for %dx%d, got %zu", expected, w, h, rgb_len);
  if (m && m->kind != 0) MD_FAIL(MD_ERR_INVALID_ARG, "not a Depth Pro model");
  return model_infer(m, nullptr, 1, h, w, in_kind, depth, focallength_px, nullptr, fovy_rad, out_kind, (hipStream_t)stream,
                     rgb, rgb_len);
}

int md_model_enable_graph(md_model_t m, int enable) {
  if (!m) MD_FAIL(MD_ERR_INVALID_ARG, "model is null");
  m->graph_enabled = enable != 0;
  return MD_OK;
}

int md_model_query(md_model_t m, const char* key, int64_t* out) {
  if (!m || !key || !out) MD_FAIL(MD_ERR_INVALID_ARG, "null argument");
  const std::string k = key;
  if (k == "img_size") *out = m->S;
  else if (k == "patch_window") *out = m->win;
  else if (k == "interpolation") *out = m->cfg.interpolation;
  else if (k == "precision") *out = m->prec;
  else if (k == "max_batch") *out = m->cfg.max_batch;
  else if (k == "num_params") {
    int64_t n = 0;
    for (auto& p : m->params) n += (int64_t)p.count();
    *out = n;
  } else if (k == "workspace_bytes") *out = (int64_t)m->ws.cap;
  else if (k == "weight_bytes") *out = m->parent ? 0 : (int64_t)(m->w32_bytes + m->wpk_bytes);  // a fork owns none
  else if (k == "tiles_per_image") *out = m->steps0 * m->steps0 + m->steps1 * m->steps1 + 1;
  else if (k == "seq_stride") *out = m->SS;
  else if (k == "is_fork") *out = m->parent ? 1 : 0;
  else if (k == "forks") *out = m->forks.load();
  else if (k == "weight_terms") *out = model_root(m)->wterms;
  else if (k == "allocs") *out = m->alloc_count;
  else if (k == "da3_shape_builds") *out = da3_shape_builds(m);
  else if (k == "batch_invariant") *out = m->batch_invariant ? 1 : 0;
  else if (k == "ln_fold") *out = m->ln_fold_opt;
  else if (k == "ln_fold_active") *out = m->ln_fold_on() ? 1 : 0;
  else if (model_decoder_query(m, k, out)) {}
  else MD_FAIL(MD_ERR_INVALID_ARG, "unknown query key `%s`", key);
  return MD_OK;
}

int md_model_set_option(md_model_t m, const char* key, int64_t value) {
  if (!m || !key) MD_FAIL(MD_ERR_INVALID_ARG, "null argument");
  const std::string k = key;
  if (k == "batch_invariant") {
    if (value != 0 && value != 1) MD_FAIL(MD_ERR_INVALID_ARG, "batch_invariant takes 0 or 1");
    if (m->batch_invariant != (value != 0)) {
      m->batch_invariant = value != 0;
      MD_HIP(hipSetDevice(m->dev->ordinal));
      MD_HIP(hipDeviceSynchronize());
      for (auto& kv : m->graphs)  // captured graphs hold the kernel forms of the old setting
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
      m->graphs.clear();
    }
    return MD_OK;
  }
  if (k == "ln_fold") {
    if (value < 0 || value > 4) MD_FAIL(MD_ERR_INVALID_ARG, "ln_fold takes 0 (off), 1 (automatic), 2 (on whenever the model can), 3 / 4 (bench diagnostics: unfolded schedule through the fold-form consumer kernels on neutral statistics / the fold with the ln_finish launch)");
    if (value >= 2 && !m->ln_fold_can) MD_FAIL(MD_ERR_UNSUPPORTED, "ln_fold: this model cannot fold its LayerNorms (16-bit Depth Pro models of width %% 256 == 0 can)");
    if (m->ln_fold_opt != (int)value) {
      m->ln_fold_opt = (int)value;
      MD_HIP(hipSetDevice(m->dev->ordinal));
      MD_HIP(hipDeviceSynchronize());
      for (auto& kv : m->graphs)  // captured graphs hold the launches of the old setting
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
      m->graphs.clear();
    }
    return MD_OK;
  }
  MD_FAIL(MD_ERR_INVALID_ARG, "unknown option `%s`", key);
}

int md_model_enable_taps(md_model_t m, int enable) {
  if (!m) MD_FAIL(MD_ERR_INVALID_ARG, "model is null");
  m->taps_enabled = enable != 0;
  return MD_OK;
}

int md_model_read_tap(md_model_t m, const char* name, float* host_data, size_t capacity, int64_t dims[4]) {
  if (!m || !name) MD_FAIL(MD_ERR_INVALID_ARG, "null argument");
  auto it = m->taps.find(name);
  if (it == m->taps.end() || !it->second.dev) MD_FAIL(MD_ERR_INVALID_ARG, "tap `%s` was not captured", name);
  if (dims) memcpy(dims, it->second.dims, sizeof(int64_t) * 4);
  if (!host_data) return MD_OK;
  if (capacity < it->second.count) MD_FAIL(MD_ERR_SHAPE, "tap `%s` needs %zu floats, capacity %zu", name, it->second.count, capacity);
  MD_HIP(hipSetDevice(m->dev->ordinal));
  MD_HIP(hipDeviceSynchronize());
  MD_HIP(hipMemcpy(host_data, it->second.dev, it->second.count * 4, hipMemcpyDeviceToHost));
  return MD_OK;
}

int md_model_enable_timing(md_model_t m, int enable) {
  if (!m) MD_FAIL(MD_ERR_INVALID_ARG, "model is null");
  m->timing_enabled = enable != 0;
  return MD_OK;
}

int md_model_set_timing_filter(md_model_t m, const char* family) {
  if (!m) MD_FAIL(MD_ERR_INVALID_ARG, "model is null");
  m->timing_filter = family ? family : "";
  return MD_OK;
}

int md_model_read_timing(md_model_t m, const char** names, float* ms, int* calls, int cap, int* n) {
  if (!m || !n) MD_FAIL(MD_ERR_INVALID_ARG, "null argument");
  MD_HIP(hipSetDevice(m->dev->ordinal));
  MD_HIP(hipDeviceSynchronize());
  m->timing_names_out.clear();
  std::vector<float> tot;
  std::vector<int> cnt;
  for (auto& t : m->timing) {
    float e = 0.f;
    if (hipEventElapsedTime(&e, t.a, t.b) != hipSuccess) continue;
    size_t i = 0;
    for (; i < m->timing_names_out.size(); ++i)
      if (m->timing_names_out[i] == t.name) break;
    if (i == m->timing_names_out.size()) {
      m->timing_names_out.push_back(t.name);
      tot.push_back(0.f);
      cnt.push_back(0);
    }
    tot[i] += e;
    cnt[i] += 1;
  }
  for (auto& t : m->timing) {  // entries accumulate across infers until they are read
    (void)hipEventDestroy(t.a);
    (void)hipEventDestroy(t.b);
  }
  m->timing.clear();
  *n = (int)m->timing_names_out.size();
  for (int i = 0; i < *n && i < cap; ++i) {
    if (names) names[i] = m->timing_names_out[i].c_str();
    if (ms) ms[i] = tot[i];
    if (calls) calls[i] = cnt[i];
  }
  return MD_OK;
}

int md_model_read_launch_order(md_model_t m, const char** names, int cap, int* n) {
  if (!m || !n) MD_FAIL(MD_ERR_INVALID_ARG, "null argument");
  *n = (int)m->timing.size();
  for (int i = 0; i < *n && i < cap; ++i)
    if (names) names[i] = m->timing[i].name.c_str();
  return MD_OK;
}

// ------------------------------------------------------------------------------------------------
// stand-alone operators
// ------------------------------------------------------------------------------------------------
int md_op_rgb_to_input(md_device_t dev, const uint8_t* rgb_dev, size_t rgb_len, int w, int h, float* out_dev, void* stream) {
  if (!dev || !rgb_dev || !out_dev) MD_FAIL(MD_ERR_INVALID_ARG, "null argument");
  if (w <= 0 || h <= 0 || rgb_len != (size_t)w * h * 3)
    MD_FAIL(MD_ERR_SHAPE, "expected %zu RGB bytes for %dx%d, got %zu", (size_t)w * h * 3, w, h, rgb_len);
  MD_HIP(hipSetDevice(dev->ordinal));
  return launch_rgb_to_input(rgb_dev, w, h, out_dev, pick_stream(dev, stream));
}

int md_op_resize_bilinear(md_device_t dev, const float* in_dev, int B, int C, int H, int W, float* out_dev, int OH, int OW,
                          int method, void* stream) {
  if (!dev || !in_dev || !out_dev) MD_FAIL(MD_ERR_INVALID_ARG, "null argument");
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) MD_FAIL(MD_ERR_SHAPE, "invalid input shape");
  if (method != MD_INTERP_CUSTOM && method != MD_INTERP_BURN) MD_FAIL(MD_ERR_INVALID_ARG, "unknown interpolation method %d", method);
  MD_HIP(hipSetDevice(dev->ordinal));
  return launch_resize_bilinear(in_dev, B * C, H, W, out_dev, OH, OW, method, 0, pick_stream(dev, stream));
}

int md_op_pyramid_patchify(md_device_t dev, const float* x_dev, int B, int S, int window, int patch, int method, int precision,
                           int force_generic, void* out_dev, int* rows_out, int* cols_out, void* stream) {
  if (!dev) MD_FAIL(MD_ERR_INVALID_ARG, "null argument");
  if (B <= 0 || S <= 0 || window <= 0 || patch <= 0 || S != 4 * window || window % patch != 0) MD_FAIL(MD_ERR_SHAPE, "pyramid: S must be 4 * window and window a multiple of the patch size");
  if (precision != MD_PREC_BF16 && precision != MD_PREC_F32 && precision != MD_PREC_F16 && precision != MD_PREC_F16X2) MD_FAIL(MD_ERR_INVALID_ARG, "unknown precision %d", precision);
  PyramidGeom g;
  g.B = B; g.S = S; g.win = window; g.ps = patch; g.method = method;
  split_geometry(S, window, 0.25f, &g.stride0, &g.steps0);      // encoder.rs:329
  split_geometry(S / 2, window, 0.5f, &g.stride1, &g.steps1);   // encoder.rs:330
  const int grid = window / patch;
  if (rows_out) *rows_out = (g.steps0 * g.steps0 + g.steps1 * g.steps1 + 1) * B * grid * grid;
  if (cols_out) *cols_out = 3 * patch * patch;
  if (!x_dev || !out_dev) return MD_OK;  // geometry query
  MD_HIP(hipSetDevice(dev->ordinal));
  return launch_pyramid_patchify(x_dev, g, out_dev, precision, pick_stream(dev, stream), force_generic != 0);
}

int md_op_resize_nhwc(md_device_t dev, const void* in_dev, int B, int H, int W, int C, void* out_dev, int OH, int OW, int method,
                      int precision, void* stream) {
  if (!dev || !in_dev || !out_dev) MD_FAIL(MD_ERR_INVALID_ARG, "null argument");
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) MD_FAIL(MD_ERR_SHAPE, "invalid shape");
  if (method != MD_INTERP_CUSTOM && method != MD_INTERP_BURN) MD_FAIL(MD_ERR_INVALID_ARG, "unknown interpolation method %d", method);
  if (precision != MD_PREC_BF16 && precision != MD_PREC_F32 && precision != MD_PREC_F16) MD_FAIL(MD_ERR_INVALID_ARG, "unknown precision %d", precision);
  MD_HIP(hipSetDevice(dev->ordinal));
  return launch_resize_nhwc(in_dev, B, H, W, C, C, out_dev, OH, OW, C, method, nullptr, precision, pick_stream(dev, stream));
}

int md_op_resize_output_size(int H, int W, float scale_h, float scale_w, int* oh, int* ow) {
  if (!oh || !ow) MD_FAIL(MD_ERR_INVALID_ARG, "null argument");
  // interpolate.rs:24-27
  const long a = (long)std::floor((float)H * scale_h), b = (long)std::floor((float)W * scale_w);
  *oh = (int)(a > 1 ? a : 1);
  *ow = (int)(b > 1 ? b : 1);
  return MD_OK;
}

int md_op_split(md_device_t dev, const float* in_dev, int B, int C, int S, int window, float overlap, float* out_dev,
                int* steps_out, void* stream) {
  if (!dev || !in_dev) MD_FAIL(MD_ERR_INVALID_ARG, "null argument");
  if (B <= 0 || C <= 0 || S <= 0 || window <= 0) MD_FAIL(MD_ERR_SHAPE, "invalid shape");
  int stride, steps;
  split_geometry(S, window, overlap, &stride, &steps);
  if (steps_out) *steps_out = steps;
  if (!out_dev) return MD_OK;  // geometry query
  if ((steps - 1) * stride + window > S) MD_FAIL(MD_ERR_SHAPE, "window %d with stride %d does not tile %d", window, stride, S);
  MD_HIP(hipSetDevice(dev->ordinal));
  return launch_split(in_dev, B, C, S, window, stride, steps, out_dev, pick_stream(dev, stream));
}

int md_op_merge(md_device_t dev, const float* in_dev, int tiles, int C, int h, int w, int batch, int padding, float* out_dev,
                int* out_h, int* out_w, void* stream) {
  if (!dev) MD_FAIL(MD_ERR_INVALID_ARG, "null argument");
  if (tiles <= 0 || batch <= 0 || C <= 0 || h <= 0 || w <= 0 || padding < 0) MD_FAIL(MD_ERR_SHAPE, "invalid shape");
  const int steps = (int)std::lround(std::sqrt((double)(tiles / batch)));  // encoder.rs:240
  if (steps <= 0 || steps * steps * batch != tiles) MD_FAIL(MD_ERR_SHAPE, "%d tiles is not steps^2 * batch(%d)", tiles, batch);
  if (steps > 1 && (h - 2 * padding <= 0 || w - 2 * padding <= 0)) MD_FAIL(MD_ERR_SHAPE, "padding %d too large for %dx%d tiles", padding, h, w);
  const int OH = merged_extent(h, steps, padding), OW = merged_extent(w, steps, padding);
  if (out_h) *out_h = OH;
  if (out_w) *out_w = OW;
  if (!out_dev || !in_dev) return MD_OK;
  MD_HIP(hipSetDevice(dev->ordinal));
  return launch_merge(in_dev, batch, C, h, w, steps, padding, out_dev, OH, OW, pick_stream(dev, stream));
}

int md_op_layernorm(md_device_t dev, const float* x_dev, const float* gamma_dev, const float* beta_dev, int rows, int D,
                    float eps, float* out_dev, void* stream) {
  if (!dev || !x_dev || !out_dev) MD_FAIL(MD_ERR_INVALID_ARG, "null argument");
  if (rows <= 0) MD_FAIL(MD_ERR_SHAPE, "invalid rows");
  MD_HIP(hipSetDevice(dev->ordinal));
  SeqGroups g;
  memset(&g, 0, sizeof(g));
  g.ngroups = 1;
  g.nseq[0] = rows;
  g.a[0] = gamma_dev;
  g.b[0] = beta_dev;
  return launch_layernorm(x_dev, out_dev, rows, D, eps, 1, g, MD_PREC_F32, 1, pick_stream(dev, stream));
}

int md_op_linear(md_device_t dev, const float* x_dev, const float* w_dev, const float* bias_dev, int M, int N, int K, int act,
                 int precision, float* out_dev, void* stream) {
  return md_op_linear_tile(dev, x_dev, w_dev, bias_dev, M, N, K, act, precision, TILE_AUTO, out_dev, stream);
}

int md_op_linear_tile(md_device_t dev, const float* x_dev, const float* w_dev, const float* bias_dev, int M, int N, int K,
                      int act, int precision, int tile, float* out_dev, void* stream) {
  if (!dev || !x_dev || !w_dev || !out_dev) MD_FAIL(MD_ERR_INVALID_ARG, "null argument");
  if (M <= 0 || N <= 0 || K <= 0) MD_FAIL(MD_ERR_SHAPE, "invalid shape");
  const KsplitScope ksplit(1);  // the stand-alone operator exercises the k-split form of the 64 x 64 kernel (what the DA3 engine runs)
  const bool storage_out = (precision & MD_OP_STORAGE_OUT) && (precision & 0xff) != MD_PREC_F32;
  precision &= 0xff;
  if (precision != MD_PREC_BF16 && precision != MD_PREC_F32 && precision != MD_PREC_FP8 && precision != MD_PREC_F16 && precision != MD_PREC_F16X2) MD_FAIL(MD_ERR_INVALID_ARG, "unknown precision %d", precision);
  if (K % ke_of(precision) != 0) MD_FAIL(MD_ERR_UNSUPPORTED, "K=%d must be a multiple of %d", K, ke_of(precision));
  MD_HIP(hipSetDevice(dev->ordinal));
  hipStream_t st = pick_stream(dev, stream);
  const bool split = precision == MD_PREC_F16X2;
  DevBuf xa, wa, ws;
  MD_TRY(xa.alloc((size_t)M * K * esz_of(precision) * (split ? 2 : 1)));
  MD_TRY(wa.alloc((size_t)N * K * esz_of(precision) * (split ? 3 : 1)));
  GemmParams p;
  int terms = 1;
  if (split) {
    // split-half operands: x rows [hi | lo], w rows [W | W] when every weight is an exact half, else [Wh | Wh | Wl]
    MD_TRY(split_terms_of(w_dev, (long)N * K, st, &terms));
    MD_TRY(launch_f32_to_rows(x_dev, (long)M * K, xa.p, precision, st, K));
    PackEntry e;
    e.kind = PACK_NK; e.d0 = N; e.d1 = K; e.k = 1; e.kp = K; e.terms = terms; e.dst = wa.p;
    MD_TRY(pack_weight(w_dev, e, precision, st));
  } else if (precision == MD_PREC_FP8) {
    // stand-alone fp8 check: activations on the static scale 8/448 (the engine's LayerNorm-output scale), weights
    // per output row
    const float xs = 8.0f / 448.0f;
    MD_TRY(ws.alloc((size_t)N * 4));
    MD_TRY(launch_f32_to_fp8(x_dev, (long)M * K, 1.0f / xs, xa.p, st));
    MD_TRY(launch_pack_fp8_rows(w_dev, N, K, K, wa.p, (float*)ws.p, st));
    p.wscale[0] = (const float*)ws.p;
    p.ascale = xs;
  } else {
    MD_TRY(launch_f32_to_rows(x_dev, (long)M * K, xa.p, precision, st));
    MD_TRY(launch_f32_to_rows(w_dev, (long)N * K, wa.p, precision, st));
  }
  p.N = N; p.K = K * terms; p.ngroups = 1; p.g_rows[0] = M; p.W[0] = wa.p; p.A = xa.p; p.lda = split ? 2 * K : K;
  p.a_wrap = terms == 3 ? 2 * K / ke_of(precision) : 0;
  p.epi = EPI_STORE; p.act = act; p.out_f32 = 1; p.bias[0] = bias_dev; p.out = out_dev; p.ldo = N;
  DevBuf ob;
  if (storage_out) {
    MD_TRY(ob.alloc((size_t)M * N * 2 * (split ? 2 : 1)));
    p.out_f32 = 0; p.out = ob.p;
    if (split) { p.ldo = 2 * N; p.o_plane = N; }
  }
  MD_TRY(launch_gemm(p, A_DENSE, precision, tile, st));
  if (storage_out) MD_TRY(launch_rows_to_f32(ob.p, (long)M * N, out_dev, (precision == MD_PREC_F16 || split) ? precision : MD_PREC_BF16, st, N));
  MD_HIP(hipStreamSynchronize(st));
  return MD_OK;
}

int md_op_attention(md_device_t dev, const float* qkv_dev, int T, int N, int heads, int precision, float* out_dev, void* stream) {
  if (!dev || !qkv_dev || !out_dev) MD_FAIL(MD_ERR_INVALID_ARG, "null argument");
  if (T <= 0 || N <= 0 || heads <= 0) MD_FAIL(MD_ERR_SHAPE, "invalid shape");
  MD_HIP(hipSetDevice(dev->ordinal));
  hipStream_t st = pick_stream(dev, stream);
  const int D = heads * 64, SS = (N + 3) / 4 * 4, kpad = (N + 63) / 64 * 64;
  const size_t es = esz_of(precision) * (precision == MD_PREC_F16X2 ? 2 : 1);  // split-half: two planes per element
  DevBuf qk, vT, ao, sc;
  MD_TRY(qk.alloc(((size_t)T * SS + 64) * 2 * D * es));
  MD_TRY(vT.alloc((size_t)T * heads * 64 * kpad * es));
  MD_TRY(ao.alloc(((size_t)T * SS + 64) * D * es));
  MD_TRY(launch_qkv_split(qkv_dev, T, N, heads, SS, kpad, qk.p, vT.p, attn_qscale(precision), precision, st));
  if (precision != MD_PREC_F32) {
    DevBuf redo;  // zeroed flags: 577-token bf16 launches take the assembly kernel, like the model's
    MD_TRY(redo.alloc((size_t)attention_redo_ints(T * heads) * 4));
    if (precision == MD_PREC_BF16) MD_TRY(attention_asm_prepare());
    MD_TRY(launch_attention(qk.p, vT.p, ao.p, T, SS, N, heads, D, kpad, precision, st, 0.f, (long)T * heads * 64 * kpad, (int*)redo.p));
    MD_HIP(hipStreamSynchronize(st));  // `redo` is released at the end of this scope
  } else {
    MD_TRY(sc.alloc((size_t)T * heads * SS * kpad * 4));
    GemmParams p;
    p.N = SS; p.K = 64; p.ngroups = 1; p.g_rows[0] = N;
    p.batch = T * heads; p.batch_inner = heads;
    p.A = qk.p; p.lda = 2 * D; p.a_bs[0] = (long)SS * 2 * D; p.a_bs[1] = 64;
    p.W[0] = (const float*)qk.p + D; p.ldw = 2 * D; p.w_bs[0] = (long)SS * 2 * D; p.w_bs[1] = 64;
    p.epi = EPI_STORE; p.out_f32 = 1; p.out = sc.p; p.ldo = kpad;
    p.o_bs[0] = (long)heads * SS * kpad; p.o_bs[1] = (long)SS * kpad;
    MD_TRY(launch_gemm(p, A_DENSE, precision, TILE_128x128, st));
    MD_TRY(launch_softmax_rows((float*)sc.p, (long)T * heads * SS, N, kpad, 0.125f, st));
    GemmParams q;
    q.N = 64; q.K = kpad; q.ngroups = 1; q.g_rows[0] = N;
    q.batch = T * heads; q.batch_inner = heads;
    q.A = sc.p; q.lda = kpad; q.a_bs[0] = (long)heads * SS * kpad; q.a_bs[1] = (long)SS * kpad;
    q.W[0] = vT.p; q.ldw = kpad; q.w_bs[0] = (long)heads * 64 * kpad; q.w_bs[1] = 64L * kpad;
    q.epi = EPI_STORE; q.out = ao.p; q.ldo = D; q.o_bs[0] = (long)SS * D; q.o_bs[1] = 64;
    MD_TRY(launch_gemm(q, A_DENSE, precision, TILE_128x128, st));
  }
  MD_TRY(launch_unpad_rows(ao.p, T, N, SS, D, out_dev, precision, st));
  MD_HIP(hipStreamSynchronize(st));
  return MD_OK;
}

int md_op_conv3x3(md_device_t dev, const float* x_dev, const float* w_dev, const float* bias_dev, int B, int Cin, int H, int W,
                  int Cout, int pre_relu, int precision, float* out_dev, void* stream) {
  if (!dev || !x_dev || !w_dev || !out_dev) MD_FAIL(MD_ERR_INVALID_ARG, "null argument");
  if (B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) MD_FAIL(MD_ERR_SHAPE, "invalid shape");
  const bool storage_out = (precision & MD_OP_STORAGE_OUT) && (precision & 0xff) != MD_PREC_F32;
  precision &= 0xff;
  if (Cin % ke_of(precision) != 0 || Cout % 4 != 0)
    MD_FAIL(MD_ERR_UNSUPPORTED, "conv3x3: Cin=%d must be a multiple of %d and Cout=%d of 4", Cin, ke_of(precision), Cout);
  MD_HIP(hipSetDevice(dev->ordinal));
  hipStream_t st = pick_stream(dev, stream);
  const size_t es = esz_of(precision);
  const bool split = precision == MD_PREC_F16X2;
  DevBuf xa, wa, oa, zp;
  MD_TRY(xa.alloc((size_t)B * H * W * Cin * es * (split ? 2 : 1)));
  MD_TRY(wa.alloc((size_t)Cout * 9 * Cin * es * (split ? 3 : 1)));
  MD_TRY(oa.alloc((size_t)B * H * W * Cout * 4));
  MD_TRY(zp.alloc(4096));
  MD_TRY(launch_nchw_to_nhwc(x_dev, B, Cin, H, W, xa.p, precision, pre_relu, st));
  PackEntry e;
  e.kind = PACK_CONV3; e.d0 = Cout; e.d1 = Cin; e.k = 3; e.kp = Cin; e.dst = wa.p;
  if (split) MD_TRY(split_terms_of(w_dev, (long)Cout * Cin * 9, st, &e.terms));
  MD_TRY(pack_weight(w_dev, e, precision, st));
  GemmParams p;
  p.N = Cout; p.K = 9 * Cin * e.terms; p.ngroups = 1; p.g_rows[0] = B * H * W; p.W[0] = wa.p;
  p.A = xa.p; p.cH = H; p.cW = W; p.cC = split ? 2 * Cin : Cin; p.cCk = split ? e.terms * Cin : 0; p.zero_page = zp.p;
  p.a_wrap = e.terms == 3 ? 2 * Cin / ke_of(precision) : 0;
  p.epi = EPI_STORE; p.out_f32 = storage_out ? 0 : 1; p.bias[0] = bias_dev; p.out = oa.p; p.ldo = Cout;
  if (split && storage_out) { p.ldo = 2 * Cout; p.o_plane = Cout; }
  MD_TRY(launch_gemm(p, A_CONV3, precision, TILE_AUTO, st));
  MD_TRY(launch_nhwc_to_nchw(oa.p, B, Cout, H, W, Cout, 0, out_dev, storage_out ? precision : MD_PREC_F32, st));
  MD_HIP(hipStreamSynchronize(st));
  return MD_OK;
}

int md_op_deconv2x2(md_device_t dev, const float* x_dev, const float* w_dev, const float* bias_dev, int B, int Cin, int H,
                    int W, int Cout, int precision, float* out_dev, void* stream) {
  if (!dev || !x_dev || !w_dev || !out_dev) MD_FAIL(MD_ERR_INVALID_ARG, "null argument");
  if (B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) MD_FAIL(MD_ERR_SHAPE, "invalid shape");
  const bool storage_out = (precision & MD_OP_STORAGE_OUT) && (precision & 0xff) != MD_PREC_F32;
  precision &= 0xff;
  if (Cin % ke_of(precision) != 0 || Cout % 4 != 0)
    MD_FAIL(MD_ERR_UNSUPPORTED, "deconv: Cin=%d must be a multiple of %d and Cout=%d of 4", Cin, ke_of(precision), Cout);
  MD_HIP(hipSetDevice(dev->ordinal));
  hipStream_t st = pick_stream(dev, stream);
  const size_t es = esz_of(precision);
  const bool split = precision == MD_PREC_F16X2;
  DevBuf xa, wa, oa;
  MD_TRY(xa.alloc((size_t)B * H * W * Cin * es * (split ? 2 : 1)));
  MD_TRY(wa.alloc((size_t)4 * Cout * Cin * es * (split ? 3 : 1)));
  MD_TRY(oa.alloc((size_t)B * 4 * H * W * Cout * 4));
  MD_TRY(launch_nchw_to_nhwc(x_dev, B, Cin, H, W, xa.p, precision, 0, st));
  PackEntry e;
  e.kind = PACK_DECONV; e.d0 = Cin; e.d1 = Cout; e.k = 2; e.kp = Cin; e.dst = wa.p;
  if (split) MD_TRY(split_terms_of(w_dev, (long)Cin * Cout * 4, st, &e.terms));
  MD_TRY(pack_weight(w_dev, e, precision, st));
  GemmParams p;
  p.N = 4 * Cout; p.K = Cin * e.terms; p.ngroups = 1; p.g_rows[0] = B * H * W; p.W[0] = wa.p; p.A = xa.p; p.lda = split ? 2 * Cin : Cin;
  p.a_wrap = e.terms == 3 ? 2 * Cin / ke_of(precision) : 0;
  p.epi = EPI_PIXSHUF; p.out_f32 = storage_out ? 0 : 1; p.bias[0] = bias_dev; p.out = oa.p; p.ldo = Cout;
  if (split && storage_out) { p.ldo = 2 * Cout; p.o_plane = Cout; }
  p.psH = H; p.psW = W; p.psC = Cout; p.ps_coff = 0;
  MD_TRY(launch_gemm(p, A_DENSE, precision, TILE_AUTO, st));
  MD_TRY(launch_nhwc_to_nchw(oa.p, B, Cout, 2 * H, 2 * W, Cout, 0, out_dev, storage_out ? precision : MD_PREC_F32, st));
  MD_HIP(hipStreamSynchronize(st));
  return MD_OK;
}

int md_op_conv2d_direct(md_device_t dev, const float* x_dev, const float* w_dev, const float* bias_dev, int B, int Cin, int H,
                        int W, int Cout, int k, int stride, int pad, int relu, float* out_dev, void* stream) {
  if (!dev || !x_dev || !w_dev || !out_dev) MD_FAIL(MD_ERR_INVALID_ARG, "null argument");
  if (B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || k <= 0 || stride <= 0 || pad < 0) MD_FAIL(MD_ERR_SHAPE, "invalid shape");
  const int OH = (H + 2 * pad - k) / stride + 1, OW = (W + 2 * pad - k) / stride + 1;
  if (H + 2 * pad < k || W + 2 * pad < k) MD_FAIL(MD_ERR_SHAPE, "input %dx%d smaller than kernel %d", H, W, k);
  MD_HIP(hipSetDevice(dev->ordinal));
  hipStream_t st = pick_stream(dev, stream);
  DevBuf xa, wa, oa;
  MD_TRY(xa.alloc((size_t)B * H * W * Cin * 4));
  MD_TRY(wa.alloc((size_t)Cout * k * k * Cin * 4));
  MD_TRY(oa.alloc((size_t)B * OH * OW * Cout * 4));
  MD_TRY(launch_nchw_to_nhwc(x_dev, B, Cin, H, W, xa.p, MD_PREC_F32, 0, st));
  PackEntry e;
  e.kind = PACK_DIRECT; e.d0 = Cout; e.d1 = Cin; e.k = k; e.kp = Cin; e.f32 = 1; e.dst = wa.p;
  MD_TRY(pack_weight(w_dev, e, MD_PREC_F32, st));
  MD_TRY(launch_conv_direct(xa.p, MD_PREC_F32, nullptr, B, H, W, Cin, (const float*)wa.p, bias_dev, Cout, k, stride, pad, relu,
                            (float*)oa.p, st));
  MD_TRY(launch_nhwc_to_nchw(oa.p, B, Cout, OH, OW, Cout, 0, out_dev, MD_PREC_F32, st));
  MD_HIP(hipStreamSynchronize(st));
  return MD_OK;
}

int md_op_fov_to_focal(float fovx_deg, int H, int W, float* focal_px, float* fovy_rad) {
  if (H <= 0 || W <= 0) MD_FAIL(MD_ERR_SHAPE, "invalid image size");
  fov_scalar_host(fovx_deg, H, W, focal_px, fovy_rad);
  return MD_OK;
}

namespace {
__attribute__((unused)) int fill_random(void* dst, size_t elems, int precision, uint64_t seed, float scale, hipStream_t st) {
  std::vector<float> h(std::min<size_t>(elems, (size_t)1 << 22));
  uniform_stream("bench", seed, h.size(), -scale, scale, h.data());
  DevBuf tmp;
  MD_TRY(tmp.alloc(h.size() * 4));
  MD_HIP(hipMemcpy(tmp.p, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  const size_t es = esz_of(precision);
  for (size_t off = 0; off < elems; off += h.size()) {
    const size_t n = std::min(h.size(), elems - off);
    if (precision == MD_PREC_FP8)
      MD_TRY(launch_f32_to_fp8((const float*)tmp.p, (long)(n & ~(size_t)3), 448.0f / scale * 0.25f, (char*)dst + off, st));
    else
      MD_TRY(launch_f32_to_rows((const float*)tmp.p, (long)n, (char*)dst + off * es, precision, st));
  }
  MD_HIP(hipStreamSynchronize(st));
  return MD_OK;
}
}  // namespace

int md_debug_gemm_direct_store(int on) { return md::gemm_direct_store(on); }
int md_debug_gemm_persistent(int on) { return md::gemm_persistent(on); }
int md_debug_gemm_stagger(int which, int ticks) {
  if (which < 0 || which > 3 || ticks < 0) return MD_ERR_INVALID_ARG;
  md::gemm_stagger(which, ticks);
  return MD_OK;
}

int md_gemm_ksplit_launches(void) { return (int)(md::gemm_ksplit_launches() & 0x7fffffff); }

int md_gemm_pick_tile(int M, int N, int K, int precision) {
  if (M <= 0 || N <= 0 || K <= 0) MD_FAIL(MD_ERR_INVALID_ARG, "md_gemm_pick_tile: M=%d N=%d K=%d", M, N, K);
  md::GemmParams p;
  p.N = N; p.K = K; p.ngroups = 1; p.g_rows[0] = M;
  return md::gemm_pick_tile(p, precision);
}

int md_bench_gemm(md_device_t dev, int mode, int M, int N, int K, int aux0, int aux1, int precision, int tile, int iters,
                  float* avg_ms) {
  const int dbg = tile >> 8;  // timing-only ablation flags ride in the upper bits of `tile`
  tile &= 0xff;
  if (!dev || !avg_ms || M <= 0 || N <= 0 || K <= 0 || iters <= 0) MD_FAIL(MD_ERR_INVALID_ARG, "invalid argument");
  MD_HIP(hipSetDevice(dev->ordinal));
  hipStream_t st = dev->stream;
  const size_t es = esz_of(precision);
  DevBuf a, w, o, zp;
  const size_t kw = mode == 1 ? (size_t)9 * K : (size_t)K;
  MD_TRY(a.alloc((size_t)M * K * es));
  MD_TRY(w.alloc((size_t)N * kw * es));
  MD_TRY(o.alloc((size_t)M * N * (es < 2 ? 2 : es)));
  MD_TRY(zp.alloc(4096));
  MD_TRY(fill_random(a.p, (size_t)M * K, precision, 1, 1.0f, st));
  MD_TRY(fill_random(w.p, (size_t)N * kw, precision, 2, 0.05f, st));
  GemmParams p;
  p.N = N; p.ngroups = 1; p.g_rows[0] = M; p.W[0] = w.p; p.A = a.p;
  p.epi = EPI_STORE; p.out = o.p; p.ldo = N; p.debug_flags = (dbg & 3) | ((dbg & 64) ? 4 : 0) | ((dbg & 128) ? 8 : 0);
  DevBuf bias;
  DevBuf xres, lsc;
  if (dbg & 16) {  // proj / fc2-style epilogue: x(f32) += scale * (acc + bias)
    MD_TRY(xres.alloc((size_t)M * N * 4));
    MD_TRY(lsc.alloc((size_t)N * 8));
    MD_TRY(fill_random(lsc.p, (size_t)N * 2, MD_PREC_F32, 6, 0.1f, st));
    p.epi = EPI_RESID_LS; p.out = xres.p; p.ldo = N; p.scale[0] = (const float*)lsc.p; p.bias[0] = (const float*)lsc.p + N;
  }
  if (dbg & 4) {  // fc1-style epilogue: bias + GELU
    MD_TRY(bias.alloc((size_t)N * 4));
    MD_TRY(fill_random(bias.p, (size_t)N, MD_PREC_F32, 5, 0.1f, st));
    p.bias[0] = (const float*)bias.p; p.act = ACT_GELU;
  }
  if ((dbg & 8) && mode != 1) {  // deconv k2s2 epilogue: pixel shuffle of N = 4 * psC columns; aux0 x aux1 = input pixel grid (batch folded into H)
    if ((long)aux0 * aux1 != M || N % 4 != 0) MD_FAIL(MD_ERR_SHAPE, "pixel-shuffle bench: aux0*aux1 must equal M, N = 4*C");
    MD_TRY(bias.alloc((size_t)N * 4));
    MD_TRY(fill_random(bias.p, (size_t)N, MD_PREC_F32, 5, 0.1f, st));
    p.epi = EPI_PIXSHUF; p.bias[0] = (const float*)bias.p; p.psH = aux0; p.psW = aux1; p.psC = N / 4; p.ps_f = 2; p.ldo = N / 4; p.ps_coff = 0;
  }
  int amode = A_DENSE;
  if (mode == 1) {
    if ((long)aux0 * aux1 != M) MD_FAIL(MD_ERR_SHAPE, "conv bench: H*W must equal M");
    amode = A_CONV3; p.K = 9 * K; p.cH = aux0; p.cW = aux1; p.cC = K; p.zero_page = zp.p;
  } else {
    p.K = K; p.lda = K;
  }
  for (int i = 0; i < 2; ++i) MD_TRY(launch_gemm(p, amode, precision, tile, st));
  if (dbg & 32) {  // per-block phase stamps of ONE launch (100 MHz s_memrealtime), printed to stderr
    const long blocks = (long)((M + 255) / 256) * ((N + 255) / 256);
    DevBuf sb;
    MD_TRY(sb.alloc((size_t)blocks * 128));
    p.stamps = (unsigned long long*)sb.p;
    MD_TRY(launch_gemm(p, amode, precision, tile, st));
    MD_HIP(hipStreamSynchronize(st));
    std::vector<unsigned long long> h((size_t)blocks * 16);
    MD_HIP(hipMemcpy(h.data(), sb.p, h.size() * 8, hipMemcpyDeviceToHost));
    p.stamps = nullptr;
    double d[7] = {0, 0, 0, 0, 0, 0, 0}, cyc = 0;
    unsigned long long tmin = ~0ull, tmax = 0;
    for (long b = 0; b < blocks; ++b) {
      const unsigned long long* q = &h[(size_t)b * 16];
      for (int k = 0; k < 7; ++k) d[k] += (double)(q[k + 1] - q[k]);
      cyc += (double)(q[9] - q[8]);
      tmin = std::min(tmin, q[0]);
      tmax = std::max(tmax, q[7]);
    }
    fprintf(stderr, "[stamps] blocks=%ld  setup %.2f  issue %.2f  first-wait %.2f  mainloop %.2f  epi-barrier %.2f  epi-body %.2f  store-drain %.2f us "
                    "(wave 0, mean per block); kernel span %.1f us; mainloop %.0f shader cycles = %.3f GHz\n",
            blocks, d[0] / blocks / 100.0, d[1] / blocks / 100.0, d[2] / blocks / 100.0, d[3] / blocks / 100.0, d[4] / blocks / 100.0,
            d[5] / blocks / 100.0, d[6] / blocks / 100.0, (double)(tmax - tmin) / 100.0, cyc / blocks, cyc / (d[3] * 10.0));
  }
  hipEvent_t e0, e1;
  MD_HIP(hipEventCreate(&e0));
  MD_HIP(hipEventCreate(&e1));
  MD_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) MD_TRY(launch_gemm(p, amode, precision, tile, st));
  MD_HIP(hipEventRecord(e1, st));
  MD_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  MD_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *avg_ms = ms / iters;
  return MD_OK;
}

int md_bench_attention_ex(md_device_t dev, int T, int n_tokens, int heads, int precision, float qk_scale, int iters, float* avg_ms) {
  if (!dev || !avg_ms || T <= 0 || n_tokens <= 0 || heads <= 0 || iters <= 0) MD_FAIL(MD_ERR_INVALID_ARG, "invalid argument");
  if (precision != MD_PREC_BF16 && precision != MD_PREC_F16) MD_FAIL(MD_ERR_INVALID_ARG, "attention bench: bf16 or f16");
  MD_HIP(hipSetDevice(dev->ordinal));
  hipStream_t st = dev->stream;
  const int D = heads * 64, SS = (n_tokens + 3) / 4 * 4, kpad = (n_tokens + 63) / 64 * 64;
  DevBuf qk, vT, ao;
  MD_TRY(qk.alloc(((size_t)T * SS + 64) * 2 * D * 2));
  MD_TRY(vT.alloc((size_t)T * heads * 64 * kpad * 2));
  MD_TRY(ao.alloc(((size_t)T * SS + 64) * D * 2));
  MD_TRY(fill_random(qk.p, (size_t)T * SS * 2 * D, precision, 3, qk_scale, st));
  MD_TRY(fill_random(vT.p, (size_t)T * heads * 64 * kpad, precision, 4, 1.0f, st));
  DevBuf redo;
  MD_TRY(redo.alloc((size_t)attention_redo_ints(T * heads) * 4));
  if (precision == MD_PREC_BF16) MD_TRY(attention_asm_prepare());
  for (int i = 0; i < 2; ++i) MD_TRY(launch_attention(qk.p, vT.p, ao.p, T, SS, n_tokens, heads, D, kpad, precision, st, 0.f, 0, (int*)redo.p));
  hipEvent_t e0, e1;
  MD_HIP(hipEventCreate(&e0));
  MD_HIP(hipEventCreate(&e1));
  MD_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) MD_TRY(launch_attention(qk.p, vT.p, ao.p, T, SS, n_tokens, heads, D, kpad, precision, st, 0.f, 0, (int*)redo.p));
  MD_HIP(hipEventRecord(e1, st));
  MD_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  MD_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *avg_ms = ms / iters;
  return MD_OK;
}

int md_bench_attention_qkv(md_device_t dev, const float* qkv_dev, int T, int n_tokens, int heads, int precision, int iters, float* avg_ms,
                           long* redo_units_per_launch) {
  if (!dev || !qkv_dev || !avg_ms || T <= 0 || n_tokens <= 0 || heads <= 0 || iters <= 0) MD_FAIL(MD_ERR_INVALID_ARG, "invalid argument");
  if (precision != MD_PREC_BF16 && precision != MD_PREC_F16) MD_FAIL(MD_ERR_INVALID_ARG, "attention bench: bf16 or f16");
  MD_HIP(hipSetDevice(dev->ordinal));
  hipStream_t st = dev->stream;
  const int D = heads * 64, SS = (n_tokens + 3) / 4 * 4, kpad = (n_tokens + 63) / 64 * 64;
  DevBuf qk, vT, ao, redo;
  MD_TRY(qk.alloc(((size_t)T * SS + 64) * 2 * D * 2));
  MD_TRY(vT.alloc((size_t)T * heads * 64 * kpad * 2));
  MD_TRY(ao.alloc(((size_t)T * SS + 64) * D * 2));
  MD_TRY(redo.alloc((size_t)attention_redo_ints(T * heads) * 4));
  MD_TRY(launch_qkv_split(qkv_dev, T, n_tokens, heads, SS, kpad, qk.p, vT.p, attn_qscale(precision), precision, st));
  if (precision == MD_PREC_BF16) MD_TRY(attention_asm_prepare());
  for (int i = 0; i < 2; ++i) MD_TRY(launch_attention(qk.p, vT.p, ao.p, T, SS, n_tokens, heads, D, kpad, precision, st, 0.f, 0, (int*)redo.p));
  MD_HIP(hipStreamSynchronize(st));
  (void)attention_asm_redo_units(1);
  hipEvent_t e0, e1;
  MD_HIP(hipEventCreate(&e0));
  MD_HIP(hipEventCreate(&e1));
  MD_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) MD_TRY(launch_attention(qk.p, vT.p, ao.p, T, SS, n_tokens, heads, D, kpad, precision, st, 0.f, 0, (int*)redo.p));
  MD_HIP(hipEventRecord(e1, st));
  MD_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  MD_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *avg_ms = ms / iters;
  if (redo_units_per_launch) {
    const long u = attention_asm_redo_units(1);
    *redo_units_per_launch = u < 0 ? -1 : u / iters;
  }
  return MD_OK;
}

int md_debug_attention_asm(int on) { return attention_allow_asm(on); }
long md_debug_attention_asm_launches(void) { return attention_asm_launches(); }
long md_debug_attention_redo_units(md_device_t dev, int reset) {
  if (!dev || hipSetDevice(dev->ordinal) != hipSuccess) return -1;
  return attention_asm_redo_units(reset);
}

int md_bench_attention(md_device_t dev, int T, int n_tokens, int heads, int iters, float* avg_ms) {
  return md_bench_attention_ex(dev, T, n_tokens, heads, MD_PREC_BF16, 0.7f, iters, avg_ms);
}

namespace {
int parse_da3_cfg(const md_da3_cfg* c, Da3Cfg* out) {
  if (!c || !c->variant) MD_FAIL(MD_ERR_INVALID_ARG, "config is null");
  Da3Cfg d;
  d.variant = c->variant;
  ViTDims v;
  v.in_chans = 3; v.mlp_ratio = 4; v.ps = 14;
  if (d.variant == "metric_large") {  // depth_anything3/mod.rs:153-156, dpt.rs:41-58
    v.preset = "da3_vitl14"; v.D = 1024; v.depth = 24; v.heads = 16; v.img = 518;
    d.image_size = 518; d.features = 256;
    const int oc[4] = {256, 512, 1024, 1024}, hk[4] = {4, 11, 17, 23};
    for (int i = 0; i < 4; ++i) { d.out_channels[i] = oc[i]; d.hook_ids[i] = hk[i]; }
  } else if (d.variant == "tiny") {
    v.preset = "da3_tiny14"; v.D = 256; v.depth = 4; v.heads = 4; v.img = 70;
    d.image_size = 70; d.features = 64;
    const int oc[4] = {64, 128, 256, 256}, hk[4] = {0, 1, 2, 3};
    for (int i = 0; i < 4; ++i) { d.out_channels[i] = oc[i]; d.hook_ids[i] = hk[i]; }
  } else if (d.variant == "small") {  // mod.rs:158-171,190-196; dpt.rs:60-79
    v.preset = "da3_vits14"; v.D = 384; v.depth = 12; v.heads = 6; v.img = 518;
    d.image_size = 518; d.features = 64; d.output_dim = 2; d.dual_head = true; d.ext_block_start = 4;
    d.camera_encoder = true;
    const int oc[4] = {48, 96, 192, 384}, hk[4] = {5, 7, 9, 11};
    for (int i = 0; i < 4; ++i) { d.out_channels[i] = oc[i]; d.hook_ids[i] = hk[i]; }
  } else if (d.variant == "tiny_dual") {  // test-only: same topology, 6 blocks of width 128
    v.preset = "da3_tinydual14"; v.D = 128; v.depth = 6; v.heads = 2; v.img = 70;
    d.image_size = 70; d.features = 64; d.output_dim = 2; d.dual_head = true; d.ext_block_start = 2;
    d.camera_encoder = true; d.cam_trunk_depth = 2;
    const int oc[4] = {48, 96, 64, 128}, hk[4] = {2, 3, 4, 5};
    for (int i = 0; i < 4; ++i) { d.out_channels[i] = oc[i]; d.hook_ids[i] = hk[i]; }
  } else {
    MD_FAIL(MD_ERR_INVALID_ARG, "unknown Depth-Anything-v3 variant `%s`", c->variant);
  }
  for (int i = 0; i < 4; ++i) { v.hook_ids[i] = d.hook_ids[i]; v.feat_dims[i] = d.out_channels[i]; }
  d.vit = v;
  if (c->image_size > 0) {
    if (c->image_size % v.ps != 0) MD_FAIL(MD_ERR_SHAPE, "image size %d must be divisible by patch size %d", c->image_size, v.ps);
    d.image_size = c->image_size;
  }
  if (c->image_width > 0) {
    if (c->image_width % v.ps != 0) MD_FAIL(MD_ERR_SHAPE, "image width %d must be divisible by patch size %d", c->image_width, v.ps);
    d.image_width = c->image_width;
  }
  d.precision = c->precision;
  d.max_batch = c->max_batch > 0 ? c->max_batch : 1;
  d.ln_eps = c->ln_eps > 0.f ? c->ln_eps : 1e-6f;
  if (d.precision != MD_PREC_BF16 && d.precision != MD_PREC_F32 && d.precision != MD_PREC_FP8 && d.precision != MD_PREC_F16 && d.precision != MD_PREC_F16X2)
    MD_FAIL(MD_ERR_INVALID_ARG, "unknown precision %d", d.precision);
  *out = d;
  return MD_OK;
}
}  // namespace

void md_da3_cfg_default(md_da3_cfg* cfg) {
  if (!cfg) return;
  cfg->variant = "metric_large";
  cfg->image_size = 0;
  cfg->image_width = 0;
  cfg->precision = MD_PREC_BF16;
  cfg->max_batch = 1;
  cfg->ln_eps = 1e-6f;
}

int md_da3_create(md_device_t dev, const md_da3_cfg* cfg, uint64_t seed, int init_scheme, md_model_t* out) {
  if (!dev || !out) MD_FAIL(MD_ERR_INVALID_ARG, "device/out is null");
  Da3Cfg dc;
  MD_TRY(parse_da3_cfg(cfg, &dc));
  md_model_t m = nullptr;
  MD_TRY(da3_create(dev, dc, &m));
  int s = da3_init_seeded(m, seed, init_scheme);
  if (s != MD_OK) {
    model_destroy(m);
    return s;
  }
  *out = m;
  return MD_OK;
}

int md_da3_load(md_device_t dev, const md_da3_cfg* cfg, const char* path, md_model_t* out) {
  if (!dev || !out) MD_FAIL(MD_ERR_INVALID_ARG, "device/out is null");
  Da3Cfg dc;
  MD_TRY(parse_da3_cfg(cfg, &dc));
  {
    Container probe;
    MD_TRY(read_container(path, &probe));
  }
  md_model_t m = nullptr;
  MD_TRY(da3_create(dev, dc, &m));
  int s = da3_load_container(m, path);
  if (s != MD_OK) {
    model_destroy(m);
    return s;
  }
  *out = m;
  return MD_OK;
}

int md_da3_infer(md_model_t m, const float* nchw, int B, int H, int W, int in_kind, float* depth, int out_kind, void* stream) {
  return da3_infer(m, nchw, B, H, W, in_kind, depth, out_kind, (hipStream_t)stream);
}

int md_da3_infer_ex(md_model_t m, const float* nchw, int B, int H, int W, int in_kind, const md_da3_outputs* out, int out_kind,
                    void* stream) {
  if (!out) MD_FAIL(MD_ERR_INVALID_ARG, "outputs struct is null");
  Da3Outputs o;
  o.depth = out->depth; o.depth_confidence = out->depth_confidence; o.aux = out->aux; o.aux_confidence = out->aux_confidence;
  o.pose_encoding = out->pose_encoding; o.extrinsics = out->extrinsics; o.intrinsics = out->intrinsics;
  return da3_infer_ex(m, nchw, B, H, W, in_kind, o, out_kind, (hipStream_t)stream);
}

int md_da3_infer_with_camera(md_model_t m, const float* nchw, int B, int H, int W, int in_kind, const float* extrinsics,
                             const float* intrinsics, int views, const md_da3_outputs* out, int out_kind, void* stream) {
  if (!out) MD_FAIL(MD_ERR_INVALID_ARG, "outputs struct is null");
  if (!extrinsics || !intrinsics) MD_FAIL(MD_ERR_INVALID_ARG, "extrinsics/intrinsics is null (md_da3_infer_ex is the call without camera inputs)");
  if (views < 1) MD_FAIL(MD_ERR_SHAPE, "camera inputs need at least one view, got %d", views);
  Da3Outputs o;
  o.depth = out->depth; o.depth_confidence = out->depth_confidence; o.aux = out->aux; o.aux_confidence = out->aux_confidence;
  o.pose_encoding = out->pose_encoding; o.extrinsics = out->extrinsics; o.intrinsics = out->intrinsics;
  o.cam_extrinsics = extrinsics; o.cam_intrinsics = intrinsics; o.cam_views = views;
  return da3_infer_ex(m, nchw, B, H, W, in_kind, o, out_kind, (hipStream_t)stream);
}

int md_da3_infer_raw(md_model_t m, const float* nchw, int B, int H, int W, int in_kind, float* logits, int out_kind, void* stream) {
  if (!logits) MD_FAIL(MD_ERR_INVALID_ARG, "logits is null");
  Da3Outputs o;
  o.raw_logits = logits;
  return da3_infer_ex(m, nchw, B, H, W, in_kind, o, out_kind, (hipStream_t)stream);
}

int md_da3_infer_from_tokens(md_model_t m, const float* const* tokens, int tokens_per_image, int B, int H, int W, int in_kind,
                             const md_da3_outputs* out, int out_kind, void* stream) {
  if (!out) MD_FAIL(MD_ERR_INVALID_ARG, "outputs struct is null");
  if (!tokens) MD_FAIL(MD_ERR_INVALID_ARG, "tokens is null");
  Da3Outputs o;
  o.depth = out->depth; o.depth_confidence = out->depth_confidence; o.aux = out->aux; o.aux_confidence = out->aux_confidence;
  o.pose_encoding = out->pose_encoding; o.extrinsics = out->extrinsics; o.intrinsics = out->intrinsics;
  for (int i = 0; i < 4; ++i) o.tokens[i] = tokens[i];
  if (!o.tokens[0]) MD_FAIL(MD_ERR_LEVELS, "Backbone returned fewer hooks (0) than requested (4)");
  o.tokens_per_image = tokens_per_image;
  return da3_infer_ex(m, nullptr, B, H, W, in_kind, o, out_kind, (hipStream_t)stream);
}

int md_da3_param_inventory(const md_da3_cfg* cfg, int init_scheme, int index, const char** name, size_t* count, float* lo,
                           float* hi) {
  Da3Cfg dc;
  MD_TRY(parse_da3_cfg(cfg, &dc));
  static thread_local std::vector<ParamSpec> specs;
  specs = da3_param_specs(dc, init_scheme);
  if (index >= 0 && index < (int)specs.size()) {
    if (name) *name = specs[index].name.c_str();
    if (count) *count = specs[index].count();
    if (lo) *lo = specs[index].lo;
    if (hi) *hi = specs[index].hi;
  }
  return (int)specs.size();
}

int md_param_inventory(const md_depth_pro_cfg* cfg, int init_scheme, int index, const char** name, size_t* count, float* lo,
                       float* hi) {
  ModelCfg mc;
  MD_TRY(parse_cfg(cfg, &mc));
  static thread_local std::vector<ParamSpec> specs;
  specs = depth_pro_param_specs(mc, init_scheme);
  if (index >= 0 && index < (int)specs.size()) {
    if (name) *name = specs[index].name.c_str();
    if (count) *count = specs[index].count();
    if (lo) *lo = specs[index].lo;
    if (hi) *hi = specs[index].hi;
  }
  return (int)specs.size();
}

int md_uniform_stream(const char* name, uint64_t seed, size_t count, float lo, float hi, float* out_host) {
  if (!name || !out_host) MD_FAIL(MD_ERR_INVALID_ARG, "null argument");
  uniform_stream(name, seed, count, lo, hi, out_host);
  return MD_OK;
}

int md_split_geometry(int image_size, int window, float overlap, int* stride, int* steps) {
  if (!stride || !steps || image_size <= 0 || window <= 0) MD_FAIL(MD_ERR_INVALID_ARG, "invalid argument");
  split_geometry(image_size, window, overlap, stride, steps);
  return MD_OK;
}

int md_feature_padding(int window, int stride, int feature_size) { return feature_padding(window, stride, feature_size); }

}  // extern "C"
