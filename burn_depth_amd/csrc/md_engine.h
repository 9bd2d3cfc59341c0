// Depth Pro engine: device-resident weights, static workspace plan, forward schedule.
#pragma once

#include <atomic>
#include <cstdint>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "kernels/gemm.h"
#include "kernels/ops.h"
#include "md_common.h"
#include "md_weights.h"

struct md_device_s {
  int ordinal = 0;
  hipStream_t stream = nullptr;
};

namespace md {

struct Arena {
  char* base = nullptr;
  size_t cap = 0, off = 0;
  void* take(size_t bytes) {
    size_t o = align_up(off, 256);
    if (o + bytes > cap) return nullptr;
    off = o + bytes;
    return base + o;
  }
};

// split geometry (encoder.rs:196-206) and feature padding (encoder.rs:28-38)
void split_geometry(int image_size, int patch_size, float overlap, int* stride, int* steps);
int feature_padding(int patch_size, int stride, int feature_patch_size);
// merged-map pixel -> (tile j, tile i, ty, tx) (encoder.rs:234-282 inverted)
void merge_source(int Y, int X, int h, int w, int steps, int pad, int* j, int* i, int* ty, int* tx);
int merged_extent(int h, int steps, int pad);

// PACK_HEAD_W / PACK_HEAD_B: the depth head's `deconv k2s2 (+bias) -> conv 3x3` pair (mod.rs:105-108, nothing between
// them) composed at commit into ONE 3x3 convolution on the deconv's input grid with 4 x Cout output columns (one group
// per output parity) and its nine position-class bias vectors (compose_head_kernel in md_engine.hip).
// PACK_C1C3_W / PACK_C1C3_B: a biased 1x1 convolution followed by a 3x3 convolution (the decoder's last `out_conv` and the
// head's `conv0`, decoder.rs:137 -> mod.rs:105: nothing between them) composed into one 3x3 convolution and its nine
// position-class bias vectors (compose_c1c3_kernel; the border classes are applied by launch_border_bias_fix).
enum PackKind : int { PACK_NK = 0, PACK_CONV3 = 1, PACK_DECONV = 2, PACK_DIRECT = 3, PACK_HEAD_W = 4, PACK_HEAD_B = 5, PACK_C1C3_W = 6, PACK_C1C3_B = 7 };

struct PackEntry {
  int param = -1;   // index into params (PACK_HEAD_*: the deconv weight [Cin,Cmid,2,2]; PACK_C1C3_*: the 1x1 weight [Cmid,Cin])
  int param2 = -1;  // PACK_DECONV only: 1x1 conv weight [Cout,Cout] composed behind the deconv at commit; PACK_HEAD_*: conv weight [Cout,Cmid,3,3]
  int param3 = -1, param4 = -1;  // PACK_HEAD_B / C1C3_B: bias of the first layer [Cmid], bias of the second [Cout]; PACK_DECONV pair (k == 4): param3 = Cmid
  int kind = PACK_NK;
  int d0 = 0, d1 = 0, k = 1;  // NK: N, K | CONV3: Cout, Cin | DECONV: Cin, Cout | DIRECT: Cout, Cin, k | HEAD_W/B, C1C3_W/B: Cout, Cin (k = Cmid)
  int kp = 0;       // padded contraction length per tap (elements)
  int terms = 1;    // MD_PREC_F16X2: copies of the contraction per row -- 2 = [W | W] (f16-exact weight), 3 = [Wh | Wh | Wl]; set at commit
  int f32 = 0;      // packed as f32 regardless of precision (direct conv)
  void* dst = nullptr;
  size_t bytes = 0;
};

struct VitBlockW {
  const float *n1g, *n1b, *n2g, *n2b, *qkv_b, *proj_b, *ls1, *fc1_b, *fc2_b, *ls2;
  const void *qkv_w, *proj_w, *fc1_w, *fc2_w;
  // LayerNorm fold (gemm.h GemmParams::ln_*; null when the model cannot fold): c / d of norm1 -> qkv [3D] and of norm2 -> fc1 [4D]
  const float *qkv_c = nullptr, *qkv_d = nullptr, *fc1_c = nullptr, *fc1_d = nullptr;
};
struct VitW {
  const void* pe_w;
  const float *pe_b, *cls, *pos, *norm_g, *norm_b;
  std::vector<VitBlockW> blk;
};

struct Tap {
  float* dev = nullptr;
  int64_t dims[4] = {0, 0, 0, 0};
  size_t count = 0;
};

struct TimingEntry {
  std::string name;
  hipEvent_t a, b;
};

}  // namespace md

struct md_model_s {
  md_device_t dev = nullptr;
  md::ModelCfg cfg;
  int prec = MD_PREC_BF16;
  int esz = 2;  // bytes per operand element
  int ke = 64;  // contraction elements per 128-byte LDS row
  int xm = 1;   // planes per activation element: 2 in MD_PREC_F16X2 (rows are [hi | lo])
  int wterms = 1;  // MD_PREC_F16X2: MFMA terms of a product with a plain (not composed) weight -- 2 when every such weight is
                   // f16-exact (an f16 checkpoint, mod.rs:206), else 3; decided by model_commit. 1 in the one-plane modes
  size_t vt_plane = 0;  // MD_PREC_F16X2: elements from the hi to the lo plane of V^T

  // ---- parameters ----
  std::vector<md::ParamSpec> params;
  std::unordered_map<std::string, int> pindex;
  std::vector<float*> w32;  // fp32 master copy on device, one per param
  char* w32_base = nullptr;
  size_t w32_bytes = 0;
  std::vector<md::PackEntry> packs;
  std::unordered_map<std::string, int> pack_index;
  char* wpk_base = nullptr;
  size_t wpk_bytes = 0;
  bool committed = false;
  unsigned commit_gen = 0;  // bumped by every model_commit (part of the graph-replay key)
  int ngroups = 2;
  md::VitW vit[3];
  float head_b_host = 0.f;

  // ---- workspace ----
  md::Arena ws;
  struct Buffers;
  Buffers* buf = nullptr;
  void* zero_page = nullptr;
  long alloc_count = 0;  // device / pinned-host allocations made by infer calls (staging growth, index tables): md_model_query("allocs")
  std::map<int, int*> index_tables;  // per batch size B: device int32 blob
  struct IndexSet {
    int *hi = nullptr, *mid = nullptr, *x2 = nullptr, *img = nullptr, *fov = nullptr;
  };
  std::map<int, IndexSet> index_sets;

  // ---- debug ----
  bool taps_enabled = false;
  std::map<std::string, md::Tap> taps;
  bool timing_enabled = false;
  std::string timing_filter;  // non-empty: only launches of this family are timed
  std::vector<md::TimingEntry> timing;
  std::vector<std::string> timing_names_out;

  // ---- hipGraph replay of the launch schedule (md_model_enable_graph) ----
  bool graph_enabled = false;
  // md_model_set_option("batch_invariant"): no launch-size-dependent kernel form (k-split GEMM, small-launch attention): an image's result
  // has the same bits alone and inside a batch, like the reference's `infer` (a pure batch map, encoder.rs:216-225)
  bool batch_invariant = false;
  // md_model_set_option("ln_fold"): the LayerNorms between the ViT's GEMMs folded into those GEMMs (run_vit; gemm.h GemmParams::ln_*).
  // 1 = automatic (on for 16-bit Depth Pro models of width % 256 == 0 whose sequences have >= 256 tokens: the launches that run the
  // 256 x 256 tiles anyway), 0 = off, 2 = on whenever the model can (small presets too: the four GEMMs of a block then keep to the
  // 256 x 256 kernel). A MODEL-level choice, never a per-launch one: a window of a call computes the same bits as the whole call.
  int ln_fold_opt = 1;
  bool ln_fold_can = false;    // 16-bit Depth Pro, D % 256 == 0: the workspace and the fold vectors exist
  float* lnfold_base = nullptr;  // root: the c / d vectors of every block (VitBlockW points into it)
  bool ln_fold_on() const { return ln_fold_can && (ln_fold_opt == 2 || ln_fold_opt == 4 || (ln_fold_opt == 1 && NT >= 256)); }  // (3: a bench diagnostic, run_vit)
  struct GraphEntry {
    int seen = 0;
    hipGraphExec_t exec = nullptr;
  };
  std::map<std::vector<uintptr_t>, GraphEntry> graphs;  // key: stream, shapes and every in/out pointer
  unsigned graphs_gen = 0;  // the root's commit_gen the entries of `graphs` were captured under (a fork drops them when it moves)

  // ---- md_model_fork: a fork shares the parameter / packed-weight arenas of its root model (never frees them) and
  //      owns its workspace, index tables, taps, timing, graphs and default stream ----
  md_model_s* parent = nullptr;   // root model of a fork (forks of forks attach to the root)
  std::atomic<int> forks{0};      // live forks of this root (model_fork / model_destroy may run on different threads)
  hipStream_t own_stream = nullptr;  // a fork's default stream (stream == NULL in infer)

  // ---- model kind: 0 = Depth Pro, 1 = Depth-Anything-v3 (state in md_da3.hip) ----
  int kind = 0;
  struct Da3State;
  Da3State* da3 = nullptr;
  // geometry shared by create/infer
  int S = 0, win = 0, g = 0, P = 0, NT = 0, SS = 0, kpad = 0;
  int steps0 = 0, stride0 = 0, steps1 = 0, stride1 = 0, pad_hi = 0, pad_mid = 0;
  int mh_hi = 0, mh_mid = 0;  // merged extents
};

namespace md {

int model_create(md_device_t dev, const ModelCfg& cfg, md_model_t* out);
int model_init_seeded(md_model_t m, uint64_t seed, int scheme);
int model_load_container(md_model_t m, const char* path);
int model_commit(md_model_t m);
int model_round_weights_f16(md_model_t m);
int model_destroy(md_model_t m);
int model_fork(md_model_t src, md_model_t* out);
inline md_model_s* model_root(md_model_s* m) { return m->parent ? m->parent : m; }
int model_infer(md_model_t m, const float* nchw, int B, int H, int W, int in_kind, float* depth, float* focal,
                float* fovx, float* fovy, int out_kind, hipStream_t stream, const uint8_t* rgb, size_t rgb_len);
// DepthPro::decoder_from_features / head_debug (depth_pro/mod.rs:262-307): the decoder / the depth head alone on caller tensors
int model_decoder_from_features(md_model_t m, const md_nchw_view* features, int levels, int B, int in_kind, float* out_features,
                                float* out_lowres, float* const* out_fusions, int out_kind, hipStream_t stream);
int model_head_debug(md_model_t m, const md_nchw_view* feature, int B, int in_kind, const md_head_debug* out, int out_kind,
                     hipStream_t stream);
// "decoder_levels" / "decoder_features" / "decoder_level{l}_channels" / "decoder_level{l}_size" of md_model_query; false = not such a key
bool model_decoder_query(md_model_t m, const std::string& key, int64_t* out);
// Tile-parallel mode (SURVEY 8(e) "optional second mode": the 35 + 2 ViT sequences of one image never interact before
// `merge`, layers/encoder.rs:329-348, 379-390): the ViT stage of ONE call is split into `parts` windows of the sequence
// range; a rank runs its window, every other part's final tokens and hook rows travel to the root, the root runs the rest.
struct ShardSegment {
  void* ptr;
  size_t bytes;
};
struct ShardPlan {
  int parts = 1;
  int part = 0;  // this rank's window; -1 = every window one after the other on this device (diagnostic / single-GPU test)
  int root = 0;  // the part whose rank runs encoder tail, decoder, head and FOV and owns the outputs
  // called once behind this rank's window, on the engine's stream: seg[p] = {final tokens, hook 0, hook 1} rows of part p
  int (*exchange)(void* ctx, int parts, const ShardSegment (*seg)[3], hipStream_t st) = nullptr;
  void* ctx = nullptr;
  float* window_ms = nullptr;  // part == -1: GPU milliseconds of each window [parts] ...
  float* tail_ms = nullptr;    // ... and of everything behind the ViT stage
};
int model_infer_sharded(md_model_t m, const float* nchw, int B, int H, int W, int in_kind, float* depth, float* focal,
                        float* fovx, float* fovy, int out_kind, hipStream_t stream, const ShardPlan& sp);
// the model's grow-only input staging buffer, holding `elems` floats: filled from `nchw` (host or device) when it is given
int model_stage_input(md_model_t m, const float* nchw, size_t elems, int in_kind, hipStream_t stream, float** dev);
int pack_weight(const float* src, const PackEntry& e, int prec, hipStream_t s);
// number of values of w[0..n) that are not exactly representable as an IEEE half (synchronises the stream)
int count_inexact_f16(const float* w, long n, hipStream_t s, unsigned* out);
void fov_scalar_host(float fovx_deg, int H, int W, float* focal_px, float* fovy_rad);

// ---- Depth-Anything-v3 (md_da3.hip) ----
struct Da3Cfg {
  std::string variant = "metric_large";
  ViTDims vit;
  int image_size = 518, image_width = 0, features = 256, output_dim = 1;  // image_size = rows; image_width 0 = square
  int out_channels[4] = {256, 512, 1024, 1024};
  int hook_ids[4] = {4, 11, 17, 23};
  int precision = MD_PREC_BF16, max_batch = 1;
  float ln_eps = 1e-6f;
  // `small`: dual head + camera decoder + burn_dino backbone extras from block `ext_block_start` on
  bool dual_head = false;
  int ext_block_start = -1, aux_levels = 4, aux_out1_conv_num = 5, aux_output_dim = 7;
  float rope_frequency = 100.f, qk_norm_eps = 1e-5f;
  // `CameraEncoderConfig` (camera.rs:12-37; mod.rs:164-168): only runs under `infer_with_camera`
  bool camera_encoder = false;
  int cam_heads = 16, cam_trunk_depth = 4;
  float cam_ln_eps = 1e-5f;
};
std::vector<ParamSpec> da3_param_specs(const Da3Cfg& cfg, int scheme);
int da3_create(md_device_t dev, const Da3Cfg& cfg, md_model_t* out);
int da3_init_seeded(md_model_t m, uint64_t seed, int scheme);
int da3_load_container(md_model_t m, const char* path);
int da3_infer(md_model_t m, const float* nchw, int B, int H, int W, int in_kind, float* depth, int out_kind,
              hipStream_t stream);
// DepthAnything3Inference (mod.rs:231-239); null = not wanted. aux is [B, aux_output_dim-1, 8ph, 8pw].
struct Da3Outputs {
  float *depth = nullptr, *depth_confidence = nullptr, *aux = nullptr, *aux_confidence = nullptr;
  float *pose_encoding = nullptr, *extrinsics = nullptr, *intrinsics = nullptr;
  // `infer_with_camera` (mod.rs:301-309): known world-to-camera extrinsics [B, views, 3, 4] and intrinsics [B, views, 3, 3], in the
  // memory kind of the input image. Both set + a model with a camera encoder => the encoded token conditions the backbone.
  const float *cam_extrinsics = nullptr, *cam_intrinsics = nullptr;
  int cam_views = 0;
  // `infer_from_tokens` (mod.rs:389-469): the head alone on caller-supplied hook tokens. tokens[i] = [B, tokens_per_image, din] fp32
  // in the memory kind `in_kind`; tokens_per_image = P (patch rows only) or P + 1 (a leading cls row is skipped, `patch_token_start`)
  const float* tokens[4] = {nullptr, nullptr, nullptr, nullptr};
  int tokens_per_image = 0;
  // `infer_raw` (mod.rs:364-380): [B, output_dim, H, W] -- the dual head's main logits before the activations, the mono head's
  // `forward_raw` result. When set, `depth` may be NULL and nothing else is produced.
  float* raw_logits = nullptr;
};
int da3_infer_ex(md_model_t m, const float* nchw, int B, int H, int W, int in_kind, const Da3Outputs& out, int out_kind,
                 hipStream_t stream);
void da3_destroy_state(md_model_t m);
long da3_shape_builds(md_model_t m);  // input sizes whose tables were built so far (0 for Depth Pro models)
int da3_on_commit(md_model_t m);  // re-derives the (interpolated) position table from the weights
int model_load_params_from_container(md_model_t m, const char* path);

}  // namespace md
