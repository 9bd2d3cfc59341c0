// Host-side parameter inventory / seeded init / container reader. Compiled with
// -ffp-contract=off so that the generator matches burn_depth_amd/weights.py bit for bit.
#include "md_weights.h"
#include "md_engine.h"

#include <cctype>
#include <cmath>
#include <cstring>
#include <fstream>

namespace md {

// ------------------------------------------------------------------------------------------------
// presets (layers/vit.rs:23-43)
// ------------------------------------------------------------------------------------------------
bool vit_dims_from_preset(const char* preset, ViTDims* out) {
  if (!preset) return false;
  ViTDims v;
  v.preset = preset;
  if (v.preset == "dinov2l16_384" || v.preset == "dinov2l16_128") {
    v.D = 1024; v.depth = 24; v.heads = 16; v.mlp_ratio = 4; v.ps = 16;
    v.img = v.preset == "dinov2l16_384" ? 384 : 128;
    const int h[4] = {5, 11, 17, 23}, d[4] = {256, 512, 1024, 1024};
    for (int i = 0; i < 4; ++i) { v.hook_ids[i] = h[i]; v.feat_dims[i] = d[i]; }
  } else if (v.preset == "tiny16_128") {
    v.D = 256; v.depth = 4; v.heads = 4; v.mlp_ratio = 4; v.ps = 16; v.img = 128;
    const int h[4] = {1, 2, 3, 3}, d[4] = {64, 128, 256, 256};
    for (int i = 0; i < 4; ++i) { v.hook_ids[i] = h[i]; v.feat_dims[i] = d[i]; }
  } else {
    return false;
  }
  *out = v;
  return true;
}

int parse_cfg(const md_depth_pro_cfg* c, ModelCfg* out) {
  if (!c) MD_FAIL(MD_ERR_INVALID_ARG, "config is null");
  ModelCfg m;
  if (!vit_dims_from_preset(c->patch_encoder_preset, &m.pv))
    MD_FAIL(MD_ERR_INVALID_ARG, "unsupported ViT preset `%s`", c->patch_encoder_preset ? c->patch_encoder_preset : "(null)");
  if (!vit_dims_from_preset(c->image_encoder_preset, &m.iv))
    MD_FAIL(MD_ERR_INVALID_ARG, "unsupported ViT preset `%s`", c->image_encoder_preset ? c->image_encoder_preset : "(null)");
  m.has_fov_vit = c->fov_encoder_preset != nullptr && c->fov_encoder_preset[0] != 0;
  if (m.has_fov_vit && !vit_dims_from_preset(c->fov_encoder_preset, &m.fv))
    MD_FAIL(MD_ERR_INVALID_ARG, "unsupported ViT preset `%s`", c->fov_encoder_preset);
  m.use_fov_head = c->use_fov_head != 0;
  m.F = c->decoder_features;
  m.interpolation = c->interpolation;
  m.precision = c->precision;
  m.max_batch = c->max_batch > 0 ? c->max_batch : 1;
  m.ln_eps = c->ln_eps > 0.f ? c->ln_eps : 1e-6f;
  if (m.F <= 0 || m.F % 64 != 0) MD_FAIL(MD_ERR_UNSUPPORTED, "decoder_features=%d must be a positive multiple of 64", m.F);
  if (m.interpolation != MD_INTERP_CUSTOM && m.interpolation != MD_INTERP_BURN)
    MD_FAIL(MD_ERR_INVALID_ARG, "unknown interpolation method %d", m.interpolation);
  if (m.precision != MD_PREC_BF16 && m.precision != MD_PREC_F32 && m.precision != MD_PREC_F16 && m.precision != MD_PREC_F16X2)
    MD_FAIL(MD_ERR_INVALID_ARG, "unknown precision %d", m.precision);
  *out = m;
  return MD_OK;
}

// ------------------------------------------------------------------------------------------------
// inventory -- keep in lock-step with weights.py::depth_pro_param_specs
// ------------------------------------------------------------------------------------------------
namespace {
struct SpecBuilder {
  std::vector<ParamSpec>& out;
  bool par;
  void add(const std::string& name, std::vector<int64_t> shape, double lo, double hi) {
    out.push_back(ParamSpec{name, std::move(shape), (float)lo, (float)hi});
  }
  void sym(const std::string& name, std::vector<int64_t> shape, double b) { add(name, std::move(shape), -b, b); }
  void lin(const std::string& name, int fan_out, int fan_in) {
    const double b = par ? std::sqrt(3.0 / fan_in) : std::sqrt(1.0 / fan_in);
    sym(name + ".weight", {fan_out, fan_in}, b);
    sym(name + ".bias", {fan_out}, par ? 0.1 : std::sqrt(1.0 / fan_in));
  }
  void conv(const std::string& name, int cout, int cin, int k, bool bias, bool relu_after = false, double gain = 0.0) {
    const int fan = cin * k * k;
    double b;
    if (par) {
      b = std::sqrt((relu_after ? 6.0 : 3.0) / fan);
      if (gain != 0.0) b *= gain;
    } else {
      b = std::sqrt(1.0 / fan);
    }
    sym(name + ".weight", {cout, cin, k, k}, b);
    if (bias) sym(name + ".bias", {cout}, par ? 0.1 : b);
  }
  void deconv(const std::string& name, int cin, int cout, bool bias) {
    const double b = par ? std::sqrt(3.0 / cin) : std::sqrt(1.0 / (cout * 4));
    sym(name + ".weight", {cin, cout, 2, 2}, b);
    if (bias) sym(name + ".bias", {cout}, par ? 0.1 : b);
  }
  void vit(const std::string& prefix, const ViTDims& v) {
    const int D = v.D, P = v.ps, C = v.in_chans, hidden = D * v.mlp_ratio;
    const int fan = C * P * P;
    const double b = par ? std::sqrt(3.0 / fan) : std::sqrt(1.0 / fan);
    sym(prefix + ".patch_embed.proj.weight", {D, C, P, P}, b);
    sym(prefix + ".patch_embed.proj.bias", {D}, par ? 0.1 : b);
    sym(prefix + ".cls_token", {1, 1, D}, par ? 0.5 : 1e-6);
    sym(prefix + ".pos_embed", {1, v.ntok(), D}, par ? 0.3 : 0.02 * std::sqrt(3.0));
    for (int i = 0; i < v.depth; ++i) {
      const std::string blk = prefix + ".blocks." + std::to_string(i);
      for (const char* n : {"norm1", "norm2"}) {
        if (par) add(blk + "." + n + ".gamma", {D}, 0.5, 1.5); else add(blk + "." + n + ".gamma", {D}, 1.0, 1.0);
        if (par) sym(blk + "." + n + ".beta", {D}, 0.1); else add(blk + "." + n + ".beta", {D}, 0.0, 0.0);
      }
      lin(blk + ".attn.qkv", 3 * D, D);
      lin(blk + ".attn.proj", D, D);
      if (par) add(blk + ".ls1.gamma", {D}, 0.05, 0.3); else add(blk + ".ls1.gamma", {D}, 1.0, 1.0);
      lin(blk + ".mlp.fc1", hidden, D);
      lin(blk + ".mlp.fc2", D, hidden);
      if (par) add(blk + ".ls2.gamma", {D}, 0.05, 0.3); else add(blk + ".ls2.gamma", {D}, 1.0, 1.0);
    }
    if (par) add(prefix + ".norm.gamma", {D}, 0.5, 1.5); else add(prefix + ".norm.gamma", {D}, 1.0, 1.0);
    if (par) sym(prefix + ".norm.beta", {D}, 0.1); else add(prefix + ".norm.beta", {D}, 0.0, 0.0);
  }
  void pub(const std::string& name, int dim_in, int dim_out, int layers, int dim_int) {
    const int inter = dim_int > 0 ? dim_int : dim_out;
    conv(name + ".projection", inter, dim_in, 1, false);
    for (int l = 0; l < layers; ++l) deconv(name + ".upsample." + std::to_string(l), l == 0 ? inter : dim_out, dim_out, false);
  }
};
}  // namespace

std::vector<ParamSpec> depth_pro_param_specs(const ModelCfg& cfg, int scheme) {
  std::vector<ParamSpec> specs;
  SpecBuilder sb{specs, scheme == MD_INIT_PARITY};
  const bool par = sb.par;
  const int* dims = cfg.pv.feat_dims;
  const int F = cfg.F, E = cfg.pv.D;
  sb.vit("encoder.patch_encoder", cfg.pv);
  sb.vit("encoder.image_encoder", cfg.iv);
  sb.pub("encoder.upsample_latent0", E, F, 3, dims[0]);
  sb.pub("encoder.upsample_latent1", E, dims[0], 2, 0);
  sb.pub("encoder.upsample0", E, dims[1], 1, 0);
  sb.pub("encoder.upsample1", E, dims[2], 1, 0);
  sb.pub("encoder.upsample2", E, dims[3], 1, 0);
  sb.deconv("encoder.upsample_lowres", cfg.iv.D, dims[3], true);
  sb.conv("encoder.fuse_lowres", dims[3], dims[3] * 2, 1, true);
  const int ddims[5] = {F, dims[0], dims[1], dims[2], dims[3]};
  for (int l = 1; l < 5; ++l) sb.conv("decoder.convs." + std::to_string(l) + ".conv", F, ddims[l], 3, false);
  for (int l = 0; l < 5; ++l) {
    const std::string f = "decoder.fusions." + std::to_string(l);
    for (const char* r : {"resnet1", "resnet2"}) {
      sb.conv(f + "." + r + ".conv1", F, F, 3, true, true);
      sb.conv(f + "." + r + ".conv2", F, F, 3, true, true, 0.5);
    }
    if (l != 0) sb.deconv(f + ".deconv", F, F, false);
    sb.conv(f + ".out_conv", F, F, 1, true);
  }
  sb.conv("head.conv0", F / 2, F, 3, true);
  sb.deconv("head.deconv", F / 2, F / 2, true);
  sb.conv("head.conv1", 32, F / 2, 3, true, true);
  if (par) {
    sb.add("head.conv_out.weight", {1, 32, 1, 1}, 0.0, 0.08);
    sb.add("head.conv_out.bias", {1}, 0.05, 0.05);
  } else {
    sb.sym("head.conv_out.weight", {1, 32, 1, 1}, std::sqrt(1.0 / 32));
    sb.add("head.conv_out.bias", {1}, 0.0, 0.0);  // depth_pro/mod.rs:92-95
  }
  if (cfg.use_fov_head) {
    std::string last;
    int last_cin;
    if (cfg.has_fov_vit) {
      sb.vit("fov.encoder", cfg.fv);
      const int fan = cfg.fv.D;
      const double b = par ? std::sqrt(3.0 / fan) : std::sqrt(1.0 / fan);
      sb.sym("fov.encoder_proj.weight", {F / 2, fan}, b);
      sb.sym("fov.encoder_proj.bias", {F / 2}, par ? 0.1 : b);
      sb.conv("fov.downsample_blocks.0.conv", F / 2, F, 3, true, true);
      sb.conv("fov.head_blocks.0.conv", F / 4, F / 2, 3, true, true);
      sb.conv("fov.head_blocks.1.conv", F / 8, F / 4, 3, true, true);
      last = "fov.head_blocks.2.conv";
    } else {
      sb.conv("fov.head_blocks.0.conv", F / 2, F, 3, true, true);
      sb.conv("fov.head_blocks.1.conv", F / 4, F / 2, 3, true, true);
      sb.conv("fov.head_blocks.2.conv", F / 8, F / 4, 3, true, true);
      last = "fov.head_blocks.3.conv";
    }
    last_cin = F / 8;
    const int fan = last_cin * 36;
    const double b = par ? std::sqrt(3.0 / fan) : std::sqrt(1.0 / fan);
    sb.sym(last + ".weight", {1, last_cin, 6, 6}, b);
    if (par) sb.add(last + ".bias", {1}, 55.0, 55.0); else sb.sym(last + ".bias", {1}, b);
  }
  return specs;
}

// keep in lock-step with weights.py::da3_param_specs
std::vector<ParamSpec> da3_param_specs(const Da3Cfg& cfg, int scheme) {
  std::vector<ParamSpec> specs;
  SpecBuilder sb{specs, scheme == MD_INIT_PARITY};
  const bool par = sb.par;
  const std::string bp = "backbone.pretrained";
  sb.vit(bp, cfg.vit);
  const int* oc = cfg.out_channels;
  const int Fh = cfg.features, D = cfg.vit.D, din = cfg.dual_head ? 2 * D : D;
  const std::string hp = cfg.dual_head ? "head_dual" : "head_mono";
  auto deconv = [&](const std::string& name, int cin, int cout, int k) {
    const double b = par ? std::sqrt(3.0 / cin) : std::sqrt(1.0 / (cout * k * k));
    sb.sym(name + ".weight", {cin, cout, k, k}, b);
    sb.sym(name + ".bias", {cout}, par ? 0.1 : b);
  };
  auto norm = [&](const std::string& name, int dim) {
    if (par) sb.add(name + ".gamma", {dim}, 0.5, 1.5); else sb.add(name + ".gamma", {dim}, 1.0, 1.0);
    if (par) sb.sym(name + ".beta", {dim}, 0.1); else sb.add(name + ".beta", {dim}, 0.0, 0.0);
  };
  auto lin = [&](const std::string& name, int fan_out, int fan_in, double gain, bool fov_bias) {
    const double b = (par ? std::sqrt(3.0 / fan_in) : std::sqrt(1.0 / fan_in)) * gain;
    sb.sym(name + ".weight", {fan_out, fan_in}, b);
    if (par && fov_bias)
      sb.add(name + ".bias", {fan_out}, 0.6, 1.2);
    else
      sb.sym(name + ".bias", {fan_out}, par ? 0.1 : std::sqrt(1.0 / fan_in));
  };
  if (cfg.dual_head) {  // burn_dino extras (mod.rs:190-196)
    for (int i = cfg.ext_block_start; i < cfg.vit.depth; ++i) {
      norm(bp + ".blocks." + std::to_string(i) + ".attn.q_norm", 64);
      norm(bp + ".blocks." + std::to_string(i) + ".attn.k_norm", 64);
    }
    sb.sym(bp + ".camera_token", {1, 2, D}, par ? 0.5 : 1e-6);
    norm(hp + ".norm", din);
  }
  for (int i = 0; i < 4; ++i) sb.conv(hp + ".projects." + std::to_string(i), oc[i], din, 1, true);
  deconv(hp + ".resize_layers.0.conv_t", oc[0], oc[0], 4);
  deconv(hp + ".resize_layers.1.conv_t", oc[1], oc[1], 2);
  sb.conv(hp + ".resize_layers.3.conv", oc[3], oc[3], 3, true);
  for (int i = 0; i < 4; ++i) sb.conv(hp + ".scratch.layer" + std::to_string(i + 1) + "_rn", Fh, oc[i], 3, false);
  auto refinenets = [&](const std::string& suffix) {
    for (int i = 1; i <= 4; ++i) {
      const std::string r = hp + ".scratch.refinenet" + std::to_string(i) + suffix;
      if (i != 4) {
        sb.conv(r + ".residual1.conv1", Fh, Fh, 3, true, true);
        sb.conv(r + ".residual1.conv2", Fh, Fh, 3, true, true, 0.5);
      }
      sb.conv(r + ".residual2.conv1", Fh, Fh, 3, true, true);
      sb.conv(r + ".residual2.conv2", Fh, Fh, 3, true, true, 0.5);
      sb.conv(r + ".out_conv", Fh, Fh, 1, true);
    }
  };
  refinenets("");
  sb.conv(hp + ".scratch.output_conv1", Fh / 2, Fh, 3, true);
  sb.conv(hp + ".scratch.output_conv2.conv1", 32, Fh / 2, 3, true, true);
  sb.conv(hp + ".scratch.output_conv2.conv2", cfg.output_dim, 32, 1, true, false, 0.5);
  if (cfg.dual_head) {
    refinenets("_aux");
    for (int lvl = 0; lvl < cfg.aux_levels; ++lvl) {  // AuxPreHead (dpt.rs:1085-1113)
      int cin = Fh;
      for (int j = 0; j < cfg.aux_out1_conv_num; ++j) {
        const int cout = j % 2 == 0 ? Fh / 2 : Fh;
        sb.conv(hp + ".scratch.output_conv1_aux." + std::to_string(lvl) + ".layers." + std::to_string(j), cout, cin, 3, true);
        cin = cout;
      }
    }
    for (int lvl = 0; lvl < cfg.aux_levels; ++lvl) {  // AuxOutputHead (dpt.rs:1146-1192)
      const std::string o = hp + ".scratch.output_conv2_aux." + std::to_string(lvl);
      sb.conv(o + ".reduce", 32, Fh / 2, 3, true, true);
      if (lvl == 0) norm(o + ".norm.layer_norm", 32);
      sb.conv(o + ".project", cfg.aux_output_dim, 32, 1, true, false, 0.5);
    }
    lin("camera_decoder.backbone_1", din, din, par ? std::sqrt(2.0) : 1.0, false);  // camera.rs:113-141
    lin("camera_decoder.backbone_2", din, din, par ? std::sqrt(2.0) : 1.0, false);
    lin("camera_decoder.fc_t", 3, din, 1.0, false);
    lin("camera_decoder.fc_qvec", 4, din, 1.0, false);
    lin("camera_decoder.fc_fov", 2, din, 0.25, true);
  }
  if (cfg.camera_encoder) {  // CameraEncoder (camera.rs:50-87, 206-234): dim_in = target_dim = 9, dim_out = embed_dim
    const std::string ce = "camera_encoder";
    auto ls = [&](const std::string& name) {
      if (par) sb.add(name, {D}, 0.05, 0.3); else sb.add(name, {D}, 1.0, 1.0);
    };
    lin(ce + ".pose_branch.fc1", D / 2, 9, 1.0, false);
    lin(ce + ".pose_branch.fc2", D, D / 2, 1.0, false);
    norm(ce + ".token_norm", D);
    for (int i = 0; i < cfg.cam_trunk_depth; ++i) {
      const std::string blk = ce + ".trunk." + std::to_string(i);
      norm(blk + ".norm1", D);
      norm(blk + ".norm2", D);
      lin(blk + ".attn.qkv", 3 * D, D, 1.0, false);
      lin(blk + ".attn.proj", D, D, 1.0, false);
      ls(blk + ".ls1.gamma");
      lin(blk + ".mlp.fc1", 4 * D, D, 1.0, false);
      lin(blk + ".mlp.fc2", D, 4 * D, 1.0, false);
      ls(blk + ".ls2.gamma");
    }
    norm(ce + ".trunk_norm", D);
  }
  return specs;
}

// ------------------------------------------------------------------------------------------------
// counter-based generator
// ------------------------------------------------------------------------------------------------
uint64_t fnv1a64(const std::string& s) {
  uint64_t h = 0xCBF29CE484222325ull;
  for (unsigned char ch : s) {
    h ^= ch;
    h *= 0x100000001B3ull;
  }
  return h;
}

void uniform_stream(const std::string& name, uint64_t seed, size_t count, float lo, float hi, float* out) {
  if (lo == hi) {
    for (size_t i = 0; i < count; ++i) out[i] = lo;
    return;
  }
  const uint64_t golden = 0x9E3779B97F4A7C15ull;
  const uint64_t key = fnv1a64(name) ^ (seed * golden);
  const double w = (double)hi - (double)lo;
  const double dlo = (double)lo;
  for (size_t i = 0; i < count; ++i) {
    uint64_t z = key + (uint64_t)(i + 1) * golden;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    const double u = ((double)(z >> 40) + 0.5) * (1.0 / 16777216.0);
    volatile double prod = w * u;  // one rounding, never fused with the add
    out[i] = (float)(dlo + prod);
  }
}

// ------------------------------------------------------------------------------------------------
// safetensors subset reader
// ------------------------------------------------------------------------------------------------
namespace {
struct Json {
  const char* p;
  const char* e;
  bool ok = true;
  void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
  bool eat(char c) { ws(); if (p < e && *p == c) { ++p; return true; } return false; }
  std::string str() {
    ws();
    std::string s;
    if (p >= e || *p != '"') { ok = false; return s; }
    ++p;
    while (p < e && *p != '"') {
      if (*p == '\\' && p + 1 < e) {
        ++p;
        switch (*p) {
          case 'n': s += '\n'; break;
          case 't': s += '\t'; break;
          case 'u': p += 4; s += '?'; break;
          default: s += *p; break;
        }
        ++p;
      } else {
        s += *p++;
      }
    }
    if (p >= e) { ok = false; return s; }
    ++p;
    return s;
  }
  double num() {
    ws();
    char* end = nullptr;
    double v = strtod(p, &end);
    if (end == p) ok = false;
    p = end;
    return v;
  }
  void skip() {  // skip any value
    ws();
    if (p >= e) { ok = false; return; }
    if (*p == '"') { str(); return; }
    if (*p == '{') { ++p; if (eat('}')) return; do { str(); if (!eat(':')) { ok = false; return; } skip(); } while (ok && eat(',')); if (!eat('}')) ok = false; return; }
    if (*p == '[') { ++p; if (eat(']')) return; do { skip(); } while (ok && eat(',')); if (!eat(']')) ok = false; return; }
    while (p < e && *p != ',' && *p != '}' && *p != ']') ++p;
  }
};
}  // namespace

// ------------------------------------------------------------------------------------------------
// Burn record reader: `NamedMpkFileRecorder<HalfPrecisionSettings>` files (`.mpk`), the argument of the reference's
// `DepthPro::load(&device, path)` (depth_pro/mod.rs:193-208) and `DepthAnything3::load_file` (example/correctness.rs:977-982).
// Format as published by Burn 0.19 (burn-core record/file.rs, record/tensor.rs; the crate is not vendored in the reference
// tree and no `.mpk` exists there -- `.gitignore:8,17,32` --, so this reader is UNVALIDATED ON A REAL BURN RECORD): one
// MessagePack map { "metadata": {...}, "item": <record> } where <record> nests maps by field name (rmp_serde::to_vec_named),
// `Vec<Module>` as arrays, `Option::None` as nil, and every parameter as
//   { "id": str, "param": { "bytes": bin, "shape": [..], "dtype": "F16" | "F32" | "BF16" } }.
// The walker is tolerant: any map holding `bytes` + `shape` is a tensor; `param` / `item` wrappers and `id` entries do not
// contribute to the dotted path; everything that is not a map or an array is skipped. The dotted paths are the Burn field
// paths of the engine's parameter inventory.
// ------------------------------------------------------------------------------------------------
namespace {
struct MsgPack {
  const uint8_t* base;
  const uint8_t* p;
  const uint8_t* e;
  bool ok = true;
  int depth = 0;
  bool need(size_t n) { if ((size_t)(e - p) < n) { ok = false; return false; } return true; }
  uint64_t be(int n) {
    if (!need((size_t)n)) return 0;
    uint64_t v = 0;
    for (int i = 0; i < n; ++i) v = (v << 8) | *p++;
    return v;
  }
  // header of the next value: kind 'm' map, 'a' array, 's' str, 'b' bin, 'i' integer (value in `len`), 'x' anything else
  // (already consumed); len = entries / elements / bytes
  char head(uint64_t* len) {
    *len = 0;
    if (!need(1)) return 'x';
    const uint8_t t = *p++;
    if (t <= 0x7f) { *len = t; return 'i'; }
    if (t >= 0x80 && t <= 0x8f) { *len = t & 0x0f; return 'm'; }
    if (t >= 0x90 && t <= 0x9f) { *len = t & 0x0f; return 'a'; }
    if (t >= 0xa0 && t <= 0xbf) { *len = t & 0x1f; return 's'; }
    if (t >= 0xe0) { *len = (uint64_t)(int64_t)(int8_t)t; return 'i'; }
    switch (t) {
      case 0xc0: case 0xc2: case 0xc3: return 'x';                    // nil, false, true
      case 0xc4: *len = be(1); return 'b';
      case 0xc5: *len = be(2); return 'b';
      case 0xc6: *len = be(4); return 'b';
      case 0xc7: { const uint64_t n = be(1); if (need(n + 1)) p += n + 1; return 'x'; }  // ext 8 / 16 / 32: type byte + payload
      case 0xc8: { const uint64_t n = be(2); if (need(n + 1)) p += n + 1; return 'x'; }
      case 0xc9: { const uint64_t n = be(4); if (need(n + 1)) p += n + 1; return 'x'; }
      case 0xca: if (need(4)) p += 4; return 'x';                     // float 32 / 64
      case 0xcb: if (need(8)) p += 8; return 'x';
      case 0xcc: *len = be(1); return 'i';
      case 0xcd: *len = be(2); return 'i';
      case 0xce: *len = be(4); return 'i';
      case 0xcf: *len = be(8); return 'i';
      case 0xd0: *len = (uint64_t)(int64_t)(int8_t)be(1); return 'i';
      case 0xd1: *len = (uint64_t)(int64_t)(int16_t)be(2); return 'i';
      case 0xd2: *len = (uint64_t)(int64_t)(int32_t)be(4); return 'i';
      case 0xd3: *len = be(8); return 'i';
      case 0xd4: if (need(2)) p += 2; return 'x';                     // fixext 1 / 2 / 4 / 8 / 16
      case 0xd5: if (need(3)) p += 3; return 'x';
      case 0xd6: if (need(5)) p += 5; return 'x';
      case 0xd7: if (need(9)) p += 9; return 'x';
      case 0xd8: if (need(17)) p += 17; return 'x';
      case 0xd9: *len = be(1); return 's';
      case 0xda: *len = be(2); return 's';
      case 0xdb: *len = be(4); return 's';
      case 0xdc: *len = be(2); return 'a';
      case 0xdd: *len = be(4); return 'a';
      case 0xde: *len = be(2); return 'm';
      case 0xdf: *len = be(4); return 'm';
      default: ok = false; return 'x';                                // 0xc1: never used
    }
  }
  void skip() {  // any value
    if (++depth > 64) { ok = false; --depth; return; }
    uint64_t n;
    const char k = head(&n);
    if (k == 's' || k == 'b') { if (need(n)) p += n; }
    else if (k == 'a') { for (uint64_t i = 0; ok && i < n; ++i) skip(); }
    else if (k == 'm') { for (uint64_t i = 0; ok && i < 2 * n; ++i) skip(); }
    --depth;
  }
  std::string key() {  // a map key: a string (integer keys are rendered as decimal text)
    uint64_t n;
    const char k = head(&n);
    if (k == 's' && need(n)) { std::string s((const char*)p, (size_t)n); p += n; return s; }
    if (k == 'i') return std::to_string((long long)n);
    if (k == 'b' && need(n)) { std::string s((const char*)p, (size_t)n); p += n; return s; }
    ok = false;
    return std::string();
  }
};

// tensor leaf: `p` stands behind the header of a map with `n` entries that holds `bytes` and `shape`
int mpk_tensor(MsgPack& mp, uint64_t n, const std::string& name, Container* out) {
  ContainerTensor t;
  t.dtype = "F32";
  bool have_bytes = false, have_shape = false;
  for (uint64_t i = 0; mp.ok && i < n; ++i) {
    const std::string k = mp.key();
    if (!mp.ok) break;
    if (k == "bytes") {
      uint64_t len;
      const char kind = mp.head(&len);
      if (kind != 'b' || !mp.need(len)) MD_FAIL(MD_ERR_FORMAT, "Burn record: tensor `%s` does not hold its bytes as a MessagePack bin", name.c_str());
      t.begin = (size_t)(mp.p - mp.base);
      t.end = t.begin + (size_t)len;
      mp.p += len;
      have_bytes = true;
    } else if (k == "shape") {
      uint64_t len;
      if (mp.head(&len) != 'a') MD_FAIL(MD_ERR_FORMAT, "Burn record: malformed shape of `%s`", name.c_str());
      for (uint64_t j = 0; mp.ok && j < len; ++j) {
        uint64_t d;
        if (mp.head(&d) != 'i') MD_FAIL(MD_ERR_FORMAT, "Burn record: malformed shape of `%s`", name.c_str());
        t.shape.push_back((int64_t)d);
      }
      have_shape = true;
    } else if (k == "dtype") {
      const uint8_t* save = mp.p;
      uint64_t len;
      const char kind = mp.head(&len);
      if (kind == 's' && mp.need(len)) {
        t.dtype.assign((const char*)mp.p, (size_t)len);
        mp.p += len;
      } else if (kind == 'm' && len == 1) {  // an externally tagged variant: {"F16": nil}
        t.dtype = mp.key();
        mp.skip();
      } else {
        mp.p = save;
        mp.skip();
      }
    } else {
      mp.skip();
    }
  }
  if (!mp.ok || !have_bytes || !have_shape) MD_FAIL(MD_ERR_FORMAT, "Burn record: malformed tensor `%s`", name.c_str());
  for (auto& ch : t.dtype) ch = (char)toupper((unsigned char)ch);
  if (t.dtype != "F16" && t.dtype != "F32" && t.dtype != "BF16") MD_FAIL(MD_ERR_FORMAT, "Burn record: tensor `%s` has unsupported dtype `%s`", name.c_str(), t.dtype.c_str());
  if (out->tensors.count(name)) MD_FAIL(MD_ERR_FORMAT, "Burn record: two tensors at `%s`", name.c_str());
  out->tensors[name] = t;
  return MD_OK;
}

int mpk_walk(MsgPack& mp, const std::string& path, Container* out) {
  if (++mp.depth > 64) MD_FAIL(MD_ERR_FORMAT, "Burn record: nesting deeper than 64 levels");
  const uint8_t* start = mp.p;
  uint64_t n;
  const char kind = mp.head(&n);
  int rc = MD_OK;
  if (kind == 'm') {
    // first pass over the keys of this level: is it a tensor leaf?
    const uint8_t* body = mp.p;
    bool has_bytes = false, has_shape = false;
    for (uint64_t i = 0; mp.ok && i < n; ++i) {
      const std::string k = mp.key();
      has_bytes |= k == "bytes";
      has_shape |= k == "shape";
      mp.skip();
    }
    if (!mp.ok) MD_FAIL(MD_ERR_FORMAT, "Burn record: truncated or malformed MessagePack near `%s`", path.c_str());
    mp.p = body;
    if (has_bytes && has_shape) {
      rc = mpk_tensor(mp, n, path, out);
    } else {
      for (uint64_t i = 0; rc == MD_OK && mp.ok && i < n; ++i) {
        const std::string k = mp.key();
        if (!mp.ok) break;
        if (k == "id") { mp.skip(); continue; }
        const bool wrapper = k == "param" || k == "item";
        rc = mpk_walk(mp, wrapper ? path : (path.empty() ? k : path + "." + k), out);
      }
    }
  } else if (kind == 'a') {
    for (uint64_t i = 0; rc == MD_OK && mp.ok && i < n; ++i) rc = mpk_walk(mp, path.empty() ? std::to_string(i) : path + "." + std::to_string(i), out);
  } else if (kind == 's' || kind == 'b') {
    if (mp.need(n)) mp.p += n;
  }
  (void)start;
  --mp.depth;
  if (rc == MD_OK && !mp.ok) MD_FAIL(MD_ERR_FORMAT, "Burn record: truncated or malformed MessagePack near `%s`", path.c_str());
  return rc;
}

int read_burn_record(const char* path, Container* out) {
  MsgPack mp{out->bytes.data(), out->bytes.data(), out->bytes.data() + out->bytes.size()};
  uint64_t n;
  if (mp.head(&n) != 'm') MD_FAIL(MD_ERR_FORMAT, "checkpoint `%s` is neither a safetensors container nor a Burn record", path);
  bool found = false;
  for (uint64_t i = 0; mp.ok && i < n; ++i) {
    const std::string k = mp.key();
    if (!mp.ok) break;
    if (k == "item") {
      MD_TRY(mpk_walk(mp, std::string(), out));
      found = true;
    } else if (k == "metadata") {
      const uint8_t* save = mp.p;
      uint64_t mlen;
      if (mp.head(&mlen) == 'm') {  // {float, int, format, version, settings}: kept as strings where they are strings
        for (uint64_t j = 0; mp.ok && j < mlen; ++j) {
          const std::string mk = mp.key();
          const uint8_t* vs = mp.p;
          uint64_t vl;
          if (mp.head(&vl) == 's' && mp.need(vl)) { out->metadata[mk].assign((const char*)mp.p, (size_t)vl); mp.p += vl; }
          else { mp.p = vs; mp.skip(); }
        }
      } else {
        mp.p = save;
        mp.skip();
      }
    } else {
      mp.skip();
    }
  }
  if (!mp.ok) MD_FAIL(MD_ERR_FORMAT, "Burn record `%s`: truncated or malformed MessagePack", path);
  if (!found) MD_FAIL(MD_ERR_FORMAT, "Burn record `%s`: no `item` entry", path);
  if (out->tensors.empty()) MD_FAIL(MD_ERR_FORMAT, "Burn record `%s`: no tensors found", path);
  out->data_off = 0;
  out->burn_record = true;
  out->metadata["container"] = "burn_mpk";
  return MD_OK;
}
}  // namespace

int read_container(const char* path, Container* out) {
  if (!path) MD_FAIL(MD_ERR_INVALID_ARG, "checkpoint path is null");
  std::ifstream f(path, std::ios::binary | std::ios::ate);
  if (!f) MD_FAIL(MD_ERR_IO, "cannot open checkpoint `%s`", path);
  const std::streamsize sz = f.tellg();
  if (sz < 8) MD_FAIL(MD_ERR_FORMAT, "checkpoint `%s` is too short (%ld bytes)", path, (long)sz);
  f.seekg(0);
  out->bytes.resize((size_t)sz);
  if (!f.read((char*)out->bytes.data(), sz)) MD_FAIL(MD_ERR_IO, "short read on `%s`", path);
  uint64_t hlen = 0;
  memcpy(&hlen, out->bytes.data(), 8);
  // safetensors: 8-byte little-endian header length, then a JSON object. Anything else that opens with a MessagePack map
  // (fixmap 0x80-0x8f, map16 0xde, map32 0xdf) is read as a Burn record (`DepthPro::load`'s argument, mod.rs:193-208).
  const bool st_like = hlen <= (uint64_t)sz - 8 && hlen >= 2 && out->bytes[8] == '{';
  const uint8_t b0 = out->bytes[0];
  if (!st_like && ((b0 >= 0x80 && b0 <= 0x8f) || b0 == 0xde || b0 == 0xdf)) return read_burn_record(path, out);
  if (hlen > (uint64_t)sz - 8) MD_FAIL(MD_ERR_FORMAT, "container header length %llu exceeds file size", (unsigned long long)hlen);
  out->data_off = 8 + (size_t)hlen;
  const size_t data_len = (size_t)sz - out->data_off;
  Json j{(const char*)out->bytes.data() + 8, (const char*)out->bytes.data() + 8 + hlen};
  if (!j.eat('{')) MD_FAIL(MD_ERR_FORMAT, "container header is not a JSON object");
  if (!j.eat('}')) {
    do {
      std::string key = j.str();
      if (!j.ok || !j.eat(':')) MD_FAIL(MD_ERR_FORMAT, "malformed container header near `%s`", key.c_str());
      if (key == "__metadata__") {
        if (!j.eat('{')) MD_FAIL(MD_ERR_FORMAT, "malformed __metadata__");
        if (!j.eat('}')) {
          do {
            std::string k = j.str();
            if (!j.eat(':')) MD_FAIL(MD_ERR_FORMAT, "malformed __metadata__");
            out->metadata[k] = j.str();
          } while (j.ok && j.eat(','));
          if (!j.eat('}')) MD_FAIL(MD_ERR_FORMAT, "malformed __metadata__");
        }
      } else {
        ContainerTensor t;
        if (!j.eat('{')) MD_FAIL(MD_ERR_FORMAT, "malformed entry `%s`", key.c_str());
        do {
          std::string k = j.str();
          if (!j.eat(':')) MD_FAIL(MD_ERR_FORMAT, "malformed entry `%s`", key.c_str());
          if (k == "dtype") {
            t.dtype = j.str();
          } else if (k == "shape") {
            if (!j.eat('[')) MD_FAIL(MD_ERR_FORMAT, "malformed shape of `%s`", key.c_str());
            if (!j.eat(']')) {
              do { t.shape.push_back((int64_t)j.num()); } while (j.ok && j.eat(','));
              if (!j.eat(']')) MD_FAIL(MD_ERR_FORMAT, "malformed shape of `%s`", key.c_str());
            }
          } else if (k == "data_offsets") {
            if (!j.eat('[')) MD_FAIL(MD_ERR_FORMAT, "malformed offsets of `%s`", key.c_str());
            t.begin = (size_t)j.num();
            if (!j.eat(',')) MD_FAIL(MD_ERR_FORMAT, "malformed offsets of `%s`", key.c_str());
            t.end = (size_t)j.num();
            if (!j.eat(']')) MD_FAIL(MD_ERR_FORMAT, "malformed offsets of `%s`", key.c_str());
          } else {
            j.skip();
          }
        } while (j.ok && j.eat(','));
        if (!j.ok || !j.eat('}')) MD_FAIL(MD_ERR_FORMAT, "malformed entry `%s`", key.c_str());
        if (t.begin > t.end || t.end > data_len)
          MD_FAIL(MD_ERR_FORMAT, "tensor `%s` data range [%zu,%zu) outside the file", key.c_str(), t.begin, t.end);
        out->tensors[key] = t;
      }
    } while (j.ok && j.eat(','));
    if (!j.ok || !j.eat('}')) MD_FAIL(MD_ERR_FORMAT, "malformed container header");
  }
  return MD_OK;
}

static float f16_to_f32(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1f, man = h & 0x3ffu;
  uint32_t u;
  if (exp == 0) {
    if (man == 0) {
      u = sign;
    } else {
      int e = -1;
      do { ++e; man <<= 1; } while (!(man & 0x400u));
      u = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
    }
  } else if (exp == 31) {
    u = sign | 0x7f800000u | (man << 13);
  } else {
    u = sign | ((exp + 127 - 15) << 23) | (man << 13);
  }
  float f;
  memcpy(&f, &u, 4);
  return f;
}

int container_tensor_to_f32(const Container& c, const ContainerTensor& t, float* dst, size_t count) {
  const uint8_t* src = c.bytes.data() + c.data_off + t.begin;
  const size_t nbytes = t.end - t.begin;
  if (t.dtype == "F32") {
    if (nbytes != count * 4) MD_FAIL(MD_ERR_FORMAT, "F32 tensor has %zu bytes, expected %zu", nbytes, count * 4);
    memcpy(dst, src, nbytes);
  } else if (t.dtype == "F16") {
    if (nbytes != count * 2) MD_FAIL(MD_ERR_FORMAT, "F16 tensor has %zu bytes, expected %zu", nbytes, count * 2);
    for (size_t i = 0; i < count; ++i) { uint16_t h; memcpy(&h, src + 2 * i, 2); dst[i] = f16_to_f32(h); }
  } else if (t.dtype == "BF16") {
    if (nbytes != count * 2) MD_FAIL(MD_ERR_FORMAT, "BF16 tensor has %zu bytes, expected %zu", nbytes, count * 2);
    for (size_t i = 0; i < count; ++i) { uint16_t h; memcpy(&h, src + 2 * i, 2); uint32_t u = (uint32_t)h << 16; memcpy(&dst[i], &u, 4); }
  } else {
    MD_FAIL(MD_ERR_FORMAT, "unsupported container dtype `%s`", t.dtype.c_str());
  }
  return MD_OK;
}

}  // namespace md
