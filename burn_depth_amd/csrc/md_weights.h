// Parameter inventory, seeded synthetic init and safetensors container reader (host side).
// Bit-identical twin of burn_depth_amd/weights.py.
#pragma once

#include <map>
#include <string>
#include <vector>

#include "md_common.h"

namespace md {

struct ViTDims {
  std::string preset;
  int in_chans = 3, D = 0, depth = 0, heads = 0, mlp_ratio = 4, img = 0, ps = 16;
  int hook_ids[4] = {0, 0, 0, 0};
  int feat_dims[4] = {0, 0, 0, 0};
  int grid() const { return img / ps; }
  int ntok() const { return grid() * grid() + 1; }
  bool same_arch(const ViTDims& o) const {
    return in_chans == o.in_chans && D == o.D && depth == o.depth && heads == o.heads && mlp_ratio == o.mlp_ratio &&
           img == o.img && ps == o.ps;
  }
};

// layers/vit.rs:23-43 (+ the test-only tiny preset). Returns false for an unknown preset.
bool vit_dims_from_preset(const char* preset, ViTDims* out);

struct ModelCfg {
  ViTDims pv, iv, fv;
  bool has_fov_vit = true;
  bool use_fov_head = true;
  int F = 256;  // decoder_features
  int interpolation = MD_INTERP_CUSTOM;
  int precision = MD_PREC_BF16;
  int max_batch = 1;
  float ln_eps = 1e-6f;
  int img_size() const { return pv.img * 4; }  // encoder.rs:139-140
};

int parse_cfg(const md_depth_pro_cfg* c, ModelCfg* out);

struct ParamSpec {
  std::string name;
  std::vector<int64_t> shape;
  float lo, hi;
  size_t count() const {
    size_t n = 1;
    for (auto d : shape) n *= (size_t)d;
    return n;
  }
};

std::vector<ParamSpec> depth_pro_param_specs(const ModelCfg& cfg, int scheme);

uint64_t fnv1a64(const std::string& s);
// element i of the stream: lo + (hi-lo) * (top24(splitmix64(key + (i+1)*golden)) + 0.5) / 2^24
void uniform_stream(const std::string& name, uint64_t seed, size_t count, float lo, float hi, float* out);

// ---- checkpoint files: the engine's safetensors container, or a Burn record (`.mpk`) ----
struct ContainerTensor {
  std::string dtype;  // F32 | F16 | BF16
  std::vector<int64_t> shape;
  size_t begin = 0, end = 0;  // byte offsets inside the data section
};
struct Container {
  std::vector<uint8_t> bytes;  // whole file
  size_t data_off = 0;
  std::map<std::string, ContainerTensor> tensors;
  std::map<std::string, std::string> metadata;
  // true: the file is a Burn `NamedMpkFileRecorder` record (what `DepthPro::load` reads, depth_pro/mod.rs:193-208) -- tensor
  // names are Burn field paths and `nn::Linear` weights are stored [d_input, d_output] (the records are written behind
  // `PyTorchToBurnAdapter`, tool/import_da3.rs:199, which transposes PyTorch's [out, in]); the loader transposes them back
  bool burn_record = false;
};
// Reads `path`, dispatching on its first bytes: a safetensors header (8-byte length + JSON object) or a MessagePack map.
int read_container(const char* path, Container* out);
// Convert tensor `t` to fp32 into dst (count elements).
int container_tensor_to_f32(const Container& c, const ContainerTensor& t, float* dst, size_t count);

}  // namespace md
