/*
 * mi_depth.h -- C ABI of libmi_depth.so, the MI355X-native (gfx950) drop-in for the hot path
 * of mosure/burn_depth: DepthPro::load / DepthPro::infer (and the helpers either side of it).
 *
 * Every entry point cites the reference interface it replaces (path:line under the reference
 * repository).  Plain pointers and sizes only; no torch / HIP types in the signatures (a
 * hipStream_t is passed as void*).  All functions return 0 (MD_OK) or a negative md_status and
 * record a message retrievable with md_last_error() -- the reference's panics
 * (expect!/assert!/panic!) become checked preconditions with distinct codes, never aborts.
 *
 * Threading: one in-flight infer per md_model_t (the workspace arena is per model), matching
 * the reference's `infer(&self)` + per-caller Mutex usage (crates/bevy_burn_depth/src/lib.rs:18,29).
 */
#ifndef MI_DEPTH_H
#define MI_DEPTH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct md_device_s* md_device_t;
typedef struct md_model_s* md_model_t;

typedef enum md_status {
  MD_OK = 0,
  MD_ERR_INVALID_ARG = -1, /* null pointer, unknown preset/key (vit.rs:49-50 panic)            */
  MD_ERR_SHAPE = -2,       /* bad B/H/W, RGB length mismatch (inference.rs:90-95 Err)           */
  MD_ERR_IO = -3,          /* file cannot be read (RecorderError, mod.rs:193-208)               */
  MD_ERR_FORMAT = -4,      /* container malformed / tensor missing or wrong shape               */
  MD_ERR_HIP = -5,         /* HIP runtime failure                                               */
  MD_ERR_UNSUPPORTED = -6, /* configuration outside what the kernels support                    */
  MD_ERR_NO_FOV = -7,      /* "FOV head required for focal length" (mod.rs:329 expect)          */
  MD_ERR_OOM = -8,         /* workspace arena exhausted                                         */
  MD_ERR_LEVELS = -9       /* decoder level-count mismatch (decoder.rs:200-205 panic)           */
} md_status;

typedef enum md_mem_kind { MD_MEM_HOST = 0, MD_MEM_DEVICE = 1 } md_mem_kind;
/* Arithmetic type of the MFMA operands; accumulation, LayerNorm, softmax and the final
 * focal/clamp/reciprocal are fp32 in both modes. */
/* MD_PREC_FP8 (Depth-Anything-v3 only, BASELINE config 5): bf16 everywhere except the four ViT linear layers
 * (qkv, proj, fc1, fc2), which run on OCP e4m3 MFMA operands -- weights quantised per output channel at commit,
 * activations with static per-tensor scales -- with fp32 accumulation. */
/* MD_PREC_F16: IEEE half MFMA operands (v_mfma_f32_*_f16, the bf16 rate) -- the reference stores its weights as f16
 * (`HalfPrecisionSettings`, depth_pro/mod.rs:206), so checkpoint weights are exact operands and activations carry 3
 * more mantissa bits than bf16; stores saturate at +-65504. The accurate fast mode. */
/* MD_PREC_F16X2 (Depth Pro): the accurate FAST mode. Every activation that feeds an MFMA is kept as two IEEE-half planes,
 * value = hi + lo (hi = f16(x), lo = f16(x - hi): 22 significant bits); the reference's checkpoints are f16
 * (`HalfPrecisionSettings`, depth_pro/mod.rs:206), so a weight is an exact f16 operand and every product is two
 * v_mfma_f32_*_f16 with fp32 accumulation (W.x_hi + W.x_lo). Weights that are NOT f16-exact (fp32 checkpoints, the layer
 * products composed at commit) are kept as hi + lo too and cost a third MFMA (W_lo.x_hi); md_model_query("weight_terms")
 * says which form the committed weights took. q.k^T runs on three terms, softmax and every sum stay fp32. */
typedef enum md_precision { MD_PREC_BF16 = 0, MD_PREC_F32 = 1, MD_PREC_FP8 = 2, MD_PREC_F16 = 3, MD_PREC_F16X2 = 4 } md_precision;
/* depth_pro/interpolate.rs:11-22 */
/* Stand-alone operator checks only (md_op_linear*, md_op_conv3x3, md_op_deconv2x2): OR into `precision` to route the
 * result through the engine's storage type (bf16 in the BF16 / FP8 modes) before it is widened to the fp32 output --
 * the store epilogues the engine itself uses -- instead of the fp32 store. Ignored for MD_PREC_F32. */
#define MD_OP_STORAGE_OUT 0x100
typedef enum md_interp { MD_INTERP_CUSTOM = 0, MD_INTERP_BURN = 1 } md_interp;
/* synthetic initialisation (no trained weights exist in the reference tree) */
typedef enum md_init_scheme { MD_INIT_REFERENCE = 0, MD_INIT_PARITY = 1 } md_init_scheme;

/* DepthProConfig, depth_pro/mod.rs:35-66 (same fields, same defaults via md_depth_pro_cfg_default). */
typedef struct md_depth_pro_cfg {
  const char* patch_encoder_preset; /* "dinov2l16_384" | "dinov2l16_128" | "tiny16_128" */
  const char* image_encoder_preset;
  const char* fov_encoder_preset;   /* NULL = FOV head without its own ViT (fov.rs:118-155) */
  int decoder_features;             /* 256 */
  int use_fov_head;                 /* 1 */
  int interpolation;                /* md_interp, default CUSTOM */
  int precision;                    /* md_precision, engine-side addition */
  int max_batch;                    /* images per infer call the workspace is sized for */
  float ln_eps;                     /* burn_dino's LayerNorm eps is not visible; default 1e-6 */
} md_depth_pro_cfg;

/* Last error message of the calling thread ("" if none). Maps RecorderError / String errors. */
const char* md_last_error(void);
/* Library version string. */
const char* md_version(void);

/* `<B as Backend>::Device::default()` (README.md:21, src/lib.rs:15-22): one device = one GPU. */
int md_device_open(int hip_ordinal, md_device_t* out);
int md_device_close(md_device_t dev);
int md_device_synchronize(md_device_t dev); /* `B::sync(&device)` (bench/inference.rs:46) */

/* `DepthProConfig::default()` (depth_pro/mod.rs:54-66). */
void md_depth_pro_cfg_default(md_depth_pro_cfg* cfg);

/* `DepthPro::new(&device, cfg)` (depth_pro/mod.rs:145-191): seeded synthetic initialisation. */
int md_depth_pro_create(md_device_t dev, const md_depth_pro_cfg* cfg, uint64_t seed, int init_scheme,
                        md_model_t* out);
/* `DepthPro::load(&device, path)` (depth_pro/mod.rs:193-198): default config. `path` is either
 *  - the reference's own checkpoint: a Burn `NamedMpkFileRecorder<HalfPrecisionSettings>` record (`.mpk`, mod.rs:206) --
 *    MessagePack, tensors keyed by Burn field path, `nn::Linear` weights [d_input, d_output] (transposed on load); the reader
 *    follows Burn 0.19's published record layout and is UNVALIDATED ON A REAL BURN RECORD (none exists in the reference tree);
 *  - or the engine's safetensors container keyed by the same field paths (tools/import_weights.py).
 * The format is recognised from the first bytes of the file, not from its name. */
int md_depth_pro_load(md_device_t dev, const char* path, md_model_t* out);
/* `DepthPro::load_with_config` (depth_pro/mod.rs:200-208). */
int md_depth_pro_load_with_config(md_device_t dev, const md_depth_pro_cfg* cfg, const char* path,
                                  md_model_t* out);
/* Host-only view of a checkpoint file (either format above; no device needed): the number of tensors it holds, and for
 * 0 <= index < that number the tensor's name (Burn field path), dtype ("F16" | "F32" | "BF16"), rank and shape as stored in
 * the file (a Burn record's Linear weights read [d_input, d_output] here). `name` / `dtype` point into thread-local storage
 * that lives until the next call on this thread. Returns the tensor count, or a negative MD_ERR_* code. */
int md_checkpoint_info(const char* path, int index, const char** name, const char** dtype, int* rank, int64_t shape[8],
                       int* is_burn_record);
/* The values of one tensor of a checkpoint file widened to fp32, in the file's own element order (`count` must match). */
int md_checkpoint_read_tensor(const char* path, const char* name, float* out_host, size_t count);
/* `Module::load_record` / `into_record` (src/lib.rs:163-177): read or replace one named
 * parameter with host fp32 data. `count` = number of elements and must match. */
int md_model_set_tensor(md_model_t m, const char* name, const float* host_data, size_t count);
int md_model_get_tensor(md_model_t m, const char* name, float* host_data, size_t count);
/* Number of parameters / name+element count of the i-th one (fixed inventory order). */
int md_model_param_count(md_model_t m);
int md_model_param_info(md_model_t m, int index, const char** name, size_t* count);
/* Must be called after md_model_set_tensor calls and before the next infer: re-packs the
 * MFMA operand copies (bf16, [N][K] / tap-major layouts) from the fp32 master weights. */
int md_model_commit_weights(md_model_t m);
/* `DepthPro::load` reads an f16 record (`NamedMpkFileRecorder<HalfPrecisionSettings>`, depth_pro/mod.rs:193-208): every
 * parameter of a loaded reference model is an IEEE half widened to f32. Rounds the fp32 master copy of every parameter the
 * same way, in place, and commits -- turns a seeded (md_depth_pro_create) or fp32-loaded model into what the reference's
 * f16 checkpoint of the same weights would give. In MD_PREC_F16X2 the committed weights are then exact MFMA operands
 * (two terms per product instead of three: md_model_query "weight_terms"). */
int md_model_round_weights_f16(md_model_t m);
/* The packed device-resident weight arena (for an RCCL broadcast from rank 0). */
int md_model_weight_arena(md_model_t m, void** device_ptr, size_t* bytes);
int md_model_destroy(md_model_t m);
/* `Clone` of a loaded model / sharing `&DepthPro` between threads (`DepthPro` is `Module + Clone + Debug`,
 * depth_pro/mod.rs:119-126; `infer(&self)`, mod.rs:312; the viewer shares one model behind an Arc,
 * crates/bevy_burn_depth/src/lib.rs:18,29): a second inference context on the SAME weights. The fork aliases the root
 * model's parameter and packed-operand arenas (no copy of the 5.6 GB) and owns a workspace arena, index tables, taps,
 * timing, graphs and a default stream of its own, so one infer per context may be in flight concurrently (the
 * threading rule at the top of this file then holds per context). set_tensor / commit / weight_arena are rejected on
 * a fork; the root must be destroyed after its forks (MD_ERR_INVALID_ARG otherwise). Depth Pro models only:
 * Depth-Anything-v3 models keep per-shape tables beside their workspace, like the reference's `CachedDepthAnything3`
 * (depth_anything3/mod.rs:44,67-70, not Sync) -> MD_ERR_UNSUPPORTED. */
int md_model_fork(md_model_t m, md_model_t* out);

/* `DepthPro::infer(&self, x)` (depth_pro/mod.rs:312-364). Input NCHW fp32, ImageNet-normalised,
 * any H x W (resized to img_size^2 and back like the reference). Outputs (DepthProInference,
 * mod.rs:128-133): depth[B*H*W], focallength_px[B], fovx_deg[B], fovy_rad[B]; any output pointer
 * may be NULL to skip it. in_kind/out_kind say whether the pointers are host or device memory.
 * `stream` is a hipStream_t (NULL = the model's own stream); the call is asynchronous for
 * device outputs and synchronises for host outputs. Host pointers may be pageable memory: they travel through pinned bounce
 * buffers owned by the model (grow-only, like the device staging for H x W != img_size: no allocation per call). */
int md_depth_pro_infer(md_model_t m, const float* nchw, int B, int H, int W, int in_kind, float* depth,
                       float* focallength_px, float* fovx_deg, float* fovy_rad, int out_kind, void* stream);

/* One NCHW fp32 tensor handed across the boundary with its shape: `data` is [B, channels, height, width] (B is the call's). */
typedef struct md_nchw_view {
  const float* data;
  int channels, height, width;
} md_nchw_view;

/* `DepthPro::decoder_from_features(&self, features: &[Tensor<B,4>]) -> (Tensor<B,4>, Tensor<B,4>, Vec<Tensor<B,4>>)`
 * (depth_pro/mod.rs:262-267 = `MultiresConvDecoder::forward_with_debug`, layers/decoder.rs:195-222): the decoder alone on
 * CALLER-SUPPLIED encoder features -- the entry the reference's harness uses to replay the decoder on PyTorch's encoder
 * features (example/correctness.rs:538-560). `features[l]`, l = 0 .. levels-1, finest first: the default configuration takes
 * [B,256,768,768], [B,256,384,384], [B,512,192,192], [B,1024,96,96], [B,1024,48,48] (md_model_query "decoder_levels",
 * "decoder_level{l}_channels", "decoder_level{l}_size"). Outputs, NCHW fp32, any pointer may be NULL to skip it:
 *   out_features [B, F, s0, s0]  the fused feature map (what the depth head takes),
 *   out_lowres   [B, F, s4, s4]  `convs[last](features[last])`, the FOV network's input,
 *   out_fusions[l]               the output of fusion block l (index 0 = finest, as the reference returns them after its
 *                                `reverse()`): [B, F, 2 s_l, 2 s_l] for l >= 1, [B, F, s0, s0] for l = 0 (== out_features).
 * A wrong number of levels -> MD_ERR_LEVELS (the reference panics, decoder.rs:200-205); a level whose shape is not the
 * model's -> MD_ERR_SHAPE (Burn panics on the mismatched convolution / addition). Runs in the model's precision mode on the kernels `md_depth_pro_infer` runs; it is a debug entry (taps
 * machinery: eager, allocates its fp32 staging per call), not a hot path. */
int md_depth_pro_decoder_from_features(md_model_t m, const md_nchw_view* features, int levels, int B, int in_kind,
                                       float* out_features, float* out_lowres, float* const* out_fusions, int out_kind,
                                       void* stream);

/* `HeadDebug` (depth_pro/mod.rs:135-142): the six tensors `DepthPro::head_debug(&self, feature)` returns (mod.rs:289-307).
 * NCHW fp32; NULL fields are skipped. With s = the decoder feature's size (768 by default) and F = decoder_features:
 * conv0 [B,F/2,s,s], deconv [B,F/2,2s,2s], conv1 and relu [B,32,2s,2s], pre_out and canonical [B,1,2s,2s]. */
typedef struct md_head_debug {
  float* conv0;
  float* deconv;
  float* conv1;
  float* relu;
  float* pre_out;
  float* canonical;
} md_head_debug;

/* `DepthPro::head_debug(&self, feature: Tensor<B,4>) -> HeadDebug` (depth_pro/mod.rs:289-307): the depth head layer by layer
 * on a CALLER-SUPPLIED decoder feature [B, F, s, s] (example/correctness.rs:382-390 feeds it the decoder's output). Every
 * layer runs UN-FUSED here (conv0 3x3, deconv k2s2, conv1 3x3, relu, conv_out 1x1, relu), each tensor materialised in the
 * model's precision mode; `md_depth_pro_infer` runs the same arithmetic with conv_out . relu behind conv1's accumulators and
 * the deconv composed into conv1, so in MD_PREC_F32 the two agree to fp32 rounding and in the 16-bit
 * modes to the rounding of the materialised intermediates. A shape that is not the model's -> MD_ERR_SHAPE. Debug entry. */
int md_depth_pro_head_debug(md_model_t m, const md_nchw_view* feature, int B, int in_kind, const md_head_debug* out,
                            int out_kind, void* stream);

/* The same call with the ViT stage run as `parts` consecutive windows of its 37 B sequences (35 B patch tiles + B image +
 * B fov, layers/encoder.rs:329-348, 409; fov.rs:203) on THIS device: the launches the `parts` ranks of
 * md_comm_depth_pro_infer_tiles issue, one rank after the other, without the exchange. Results are bit-identical to
 * md_depth_pro_infer (the tiles never interact before `merge`). Single-GPU test and projection tool of the tile-parallel
 * mode: window_ms[parts] / tail_ms (either may be NULL) receive the GPU milliseconds of each window and of everything
 * behind the ViT stage, so max(window_ms) + tail_ms is the device time of one frame on `parts` GPUs before the exchange.
 * 1 <= parts <= 64; synchronises the stream when timings are requested. */
int md_depth_pro_infer_windows(md_model_t m, const float* nchw, int B, int H, int W, int in_kind, float* depth,
                               float* focallength_px, float* fovx_deg, float* fovy_rad, int out_kind, int parts,
                               float* window_ms, float* tail_ms, void* stream);

/* `infer_from_rgb` + `rgb_to_input_tensor` (src/inference.rs:79-137): packed RGB bytes,
 * row-major, `rgb_len` must equal w*h*3 (else MD_ERR_SHAPE, as the reference's Err). B = 1. */
int md_infer_from_rgb(md_model_t m, const uint8_t* rgb, size_t rgb_len, int w, int h, int in_kind,
                      float* depth, float* focallength_px, float* fovy_rad, int out_kind, void* stream);

/* ---- Depth-Anything-v3 ---------------------------------------------------------------------------------
 * "metric_large" = `DepthAnything3Config::metric_large()` (depth_anything3/mod.rs:153-156): ViT-L/14, 518x518,
 * hooks [4,11,17,23], mono head `DepthAnything3HeadConfig::metric_large` (dpt.rs:41-58).
 * "small" = `DepthAnything3Config::small()` (mod.rs:158-171): ViT-S/14 with QK-norm / 2-D RoPE / camera token /
 * concatenated hooks from block 4 (mod.rs:190-196), hooks [5,7,9,11], dual head (dpt.rs:60-79) and the camera
 * decoder (camera.rs:113-199). "tiny" / "tiny_dual" are test-only reductions of the two (70x70).
 * The model handle is an md_model_t: set/get_tensor, commit, query, timing and destroy work on it unchanged. */
typedef struct md_da3_cfg {
  const char* variant; /* "metric_large" | "small" | "tiny" | "tiny_dual" */
  int image_size;      /* square input side, a multiple of 14; 0 = the variant's native size (518 / 70). Other
                        * sizes interpolate the position embedding bicubically (DINOv2 interpolate_pos_encoding,
                        * offset 0.1) once, when the weights are committed. */
  int precision;       /* md_precision */
  int max_batch;
  float ln_eps;        /* backbone LayerNorm eps (burn_dino's is not visible; default 1e-6) */
  int image_width;     /* 0 = square (image_size x image_size); else the input is image_size rows x image_width columns,
                        * both multiples of 14 -- `DepthAnything3::infer` only asserts divisibility (mod.rs:509-520) */
} md_da3_cfg;
void md_da3_cfg_default(md_da3_cfg* cfg);
/* `DepthAnything3::new(&device, cfg)` (depth_anything3/mod.rs:253-286): seeded synthetic weights. */
int md_da3_create(md_device_t dev, const md_da3_cfg* cfg, uint64_t seed, int init_scheme, md_model_t* out);
/* `DepthAnything3::new(cfg).load_file(path, ..)` (example/correctness.rs:977-982): a Burn `.mpk` record or the engine's
 * safetensors container, as md_depth_pro_load. */
int md_da3_load(md_device_t dev, const md_da3_cfg* cfg, const char* path, md_model_t* out);
/* `DepthAnything3::infer(&self, x)` (depth_anything3/mod.rs:288-291): NCHW fp32 in, depth [B*H*W] out.
 * H and W may be ANY multiples of the patch size (mod.rs:509-520 asserts only that; else MD_ERR_SHAPE). The model keeps
 * per-size tables like the reference's `PosEmbedCache` (dpt.rs:784-833, keyed by shape) and burn_dino's interpolated
 * position embedding: the first call at a new size builds them (host work) and, if the size needs more workspace than any
 * size before it, grows the arena; later calls at that size find everything cached (md_model_query "da3_shape_builds" /
 * "allocs" stop moving). image_size x image_width of the config is just the size the model is prepared for at creation.
 * Up to 16 sizes stay cached (least recently used first out).
 * Numerics across batch sizes: in the 16-bit operand modes (bf16 / f16 / f16x2) a launch that leaves most CUs idle behind a long
 * contraction splits it over wave groups, and a small attention launch splits its keys over two groups (DESIGN.md sections 5.1, 9),
 * so image i of a batch and the same image alone agree to rounding, not to the bit; the fp32 mode and every Depth Pro call are bit-identical across batch sizes. */
int md_da3_infer(md_model_t m, const float* nchw, int B, int H, int W, int in_kind, float* depth, int out_kind,
                 void* stream);
/* `DepthAnything3Inference` (depth_anything3/mod.rs:231-239) for the dual-head `small` variant. Every pointer
 * except `depth` may be NULL (that output is then not computed). Shapes, fp32, batch-major:
 *   depth, depth_confidence [B,H,W]; aux [B,6,8*H/14,8*W/14] (ray values); aux_confidence [B,8*H/14,8*W/14];
 *   pose_encoding [B,1,9] = (t3 | quat xyzw | fov_h fov_w); extrinsics [B,1,3,4] (world-to-camera);
 *   intrinsics [B,1,3,3] (camera.rs:281-358). The mono variant accepts `depth` only (else MD_ERR_UNSUPPORTED). */
typedef struct md_da3_outputs {
  float* depth;
  float* depth_confidence;
  float* aux;
  float* aux_confidence;
  float* pose_encoding;
  float* extrinsics;
  float* intrinsics;
} md_da3_outputs;
int md_da3_infer_ex(md_model_t m, const float* nchw, int B, int H, int W, int in_kind, const md_da3_outputs* out,
                    int out_kind, void* stream);
/* `DepthAnything3::infer_with_camera` (depth_anything3/mod.rs:301-309 -> 522-531): known cameras condition the backbone.
 * extrinsics [B, views, 3, 4] (world-to-camera) and intrinsics [B, views, 3, 3], fp32, in the same memory kind as `nchw`;
 * 1 <= views <= 16. The camera encoder (camera.rs:50-110: pose encoding of each view -> PoseBranch -> token_norm -> a trunk
 * of transformer blocks over the view tokens -> trunk_norm -> mean over views) yields one token per image, which takes the
 * place of the learned camera token in the backbone. A variant without a camera encoder (`metric_large`) ignores the
 * camera inputs, as the reference's match does (mod.rs:522-527), and the call equals md_da3_infer_ex. */
int md_da3_infer_with_camera(md_model_t m, const float* nchw, int B, int H, int W, int in_kind, const float* extrinsics,
                             const float* intrinsics, int views, const md_da3_outputs* out, int out_kind, void* stream);
/* `DepthAnything3::infer_raw` (depth_anything3/mod.rs:364-380): logits [B, C, H, W] fp32. Dual head (`small`): C = 2, the main
 * branch's `depth_logits` before the activations (depth = exp(ch 0), confidence = exp(ch 1) + 1; dpt.rs:271,337-354,443-469).
 * Mono head (`metric_large`): C = 1, `forward_raw`'s result (its exp activation applied, dpt.rs:700). */
int md_da3_infer_raw(md_model_t m, const float* nchw, int B, int H, int W, int in_kind, float* logits, int out_kind, void* stream);
/* `DepthAnything3::infer_from_tokens(patches, height, width)` (depth_anything3/mod.rs:389-469; the head-only comparison of
 * example/da3_small_correctness.rs:278-322): the DPT head alone on caller-supplied hook tokens. tokens[0..3]: the four hooks'
 * `DinoIntermediate::patches`, each [B, tokens_per_image, din] fp32 in the memory kind `in_kind`; din = embed_dim (mono head) or
 * 2 * embed_dim (dual head: cat(local, final-normed)); tokens_per_image = (H/14)*(W/14) patch rows, or one more with a leading row
 * that is skipped (`patch_token_start`, mod.rs:419-424); anything else is MD_ERR_SHAPE. The head applies its own token LayerNorm.
 * No camera prediction (mod.rs:468 passes None): pose_encoding / extrinsics / intrinsics must be NULL (MD_ERR_UNSUPPORTED).
 * The trace the reference returns beside the inference (aux_stage_necks, aux_head_input) is read through the taps. */
int md_da3_infer_from_tokens(md_model_t m, const float* const* tokens, int tokens_per_image, int B, int H, int W, int in_kind,
                             const md_da3_outputs* out, int out_kind, void* stream);
int md_da3_param_inventory(const md_da3_cfg* cfg, int init_scheme, int index, const char** name, size_t* count,
                           float* lo, float* hi);

/* hipGraph replay: with it enabled, an infer call whose (stream, B, H, W, input pointer, output pointers) were seen
 * before is replayed from an instantiated graph of the launch schedule (first call eager, second call captured).
 * Only all-device-memory calls at the configured image size are eligible; timing / tap modes run eagerly. */
int md_model_enable_graph(md_model_t m, int enable);

/* `img_size()` (mod.rs:296), `interpolation_method()` (mod.rs:308) and friends.
 * keys: "img_size", "patch_window", "interpolation", "precision", "max_batch", "num_params",
 *       "workspace_bytes", "weight_bytes", "tiles_per_image", "seq_stride", "is_fork", "forks",
 *       "weight_terms" (MFMA terms per product with a plain weight: 1; MD_PREC_F16X2: 2 = f16-exact weights, 3 otherwise),
 *       "allocs" (device / pinned-host allocations the infer calls of this model have made so far: staging buffers for
 *       host pointers and non-native input sizes grow on demand and are then reused, so the count stops moving once the
 *       largest shapes have been seen), "da3_shape_builds" (Depth-Anything-v3: input sizes whose tables were built),
 *       "decoder_levels", "decoder_features", "decoder_level{0..4}_channels", "decoder_level{0..4}_size" (Depth Pro: the
 *       shapes md_depth_pro_decoder_from_features / md_depth_pro_head_debug take). */
int md_model_query(md_model_t m, const char* key, int64_t* out);

/* Model options. "batch_invariant" (0 | 1, default 0): the reference's `infer` is a pure batch map (the batch is only ever concatenated,
 * layers/encoder.rs:216-225; depth_anything3/mod.rs:495-564) -- an image gives the same tensor alone and inside a batch. Depth Pro models
 * always do (bit for bit, tested). Depth-Anything-v3 models in the 16-bit modes pick two kernel forms by LAUNCH SIZE (the k-split
 * GEMM of small long-K launches and the two-key-group attention of few-workgroup launches, DESIGN.md sections 5.1 / 5.2): deterministic,
 * but another summation order -- the last bits of an image's result can then differ between B = 1 and B = 8. With the option set
 * neither form is used (config 2: ~8 % slower at B = 1) and the batch map is exact. Also a md_model_query key.
 * "ln_fold" (0 | 1 | 2 | 3, default 1; Depth Pro): the LayerNorms between the GEMMs of a ViT block (burn_dino block order, called from
 * /root/reference/src/model/depth_pro/layers/encoder.rs:346-348) run inside those GEMMs instead of as launches (DESIGN.md section
 * 5.1.1): LN(x) W^T + b = rstd (round(gamma x) W^T - mu c) + d. Same values in exact arithmetic; the operand rounding falls on
 * gamma x instead of LN(x). 1 = automatic: on for the 16-bit modes when the ViT is 1024 wide and its sequences have >= 256 tokens
 * (the default configuration); 0 = off (stand-alone LayerNorm launches); 2 = on whenever the model can (MD_ERR_UNSUPPORTED when it
 * cannot: fp32 / fp8 modes, other widths); 3 = a bench diagnostic (the unfolded schedule through the fold-form kernels on neutral
 * statistics). A MODEL-level choice: batch sizes, sequence windows and forks compute the same bits. Set it before md_model_fork (a fork
 * copies its root's setting when it is made) and to the same value on every rank of md_comm_depth_pro_infer_tiles (latency mode at >= 4
 * ranks: 0 is faster there, the windows are too small for the 256 x 256 tiles the fold keeps to). Query keys: "ln_fold", "ln_fold_active". */
int md_model_set_option(md_model_t m, const char* key, int64_t value);

/* Debug taps (EncoderDebug encoder.rs:106-123, HeadDebug mod.rs:135-142, fusion outputs
 * mod.rs:285-287). After an infer, copy the named intermediate (converted to NCHW fp32, the
 * reference's layout) to host memory. `dims` receives up to 4 dims. Names follow
 * example/correctness.rs:98-122: encoder_feature_{0..4}, encoder_merge_latent{0,1},
 * encoder_merge_x{0,1,2}, decoder_fusion_{0..4}, decoder_feature, decoder_lowres_feature,
 * head_conv0, head_deconv, canonical_inverse_depth, fov_deg, split_x{0,1,2}.
 * Pass host_data = NULL to query dims only. Taps must be enabled before the infer.
 * Depth-Anything-v3 models (`DepthTrace` / `infer_with_trace`, depth_anything3/mod.rs:241-246,329-362, and the head's
 * stages, dpt.rs:587-731): backbone_tokens_{0..3} [B, P, D | 2D] (the hook patch tokens the head receives),
 * stage_{0..3} (prepare_stage outputs), layer{1..4}_rn, refinenet{4..1} (+ "_aux" for the dual head's second pyramid),
 * output_conv1, head_input (resized + UV table), aux_neck, aux_head_input; camera_token [B, D] (the camera encoder's
 * result, after md_da3_infer_with_camera). */
int md_model_enable_taps(md_model_t m, int enable);
int md_model_read_tap(md_model_t m, const char* name, float* host_data, size_t capacity, int64_t dims[4]);

/* ---- stand-alone operators on device pointers (the reference's public helpers; used by the
 * parity tests to check each kernel against the oracle) ------------------------------------ */
/* `rgb_to_input_tensor` (src/inference.rs:79-121) on device: u8 HWC -> fp32 NCHW. */
int md_op_rgb_to_input(md_device_t dev, const uint8_t* rgb_dev, size_t rgb_len, int w, int h, float* out_dev,
                       void* stream);
/* `resize_bilinear_align_corners_false(x, [oh,ow], method)` (interpolate.rs:123-134), fp32 NCHW. */
int md_op_resize_bilinear(md_device_t dev, const float* in_dev, int B, int C, int H, int W, float* out_dev,
                          int OH, int OW, int method, void* stream);
/* The front of `DepthProEncoder::forward` as the engine runs it (encoder.rs:326-344): pyramid x1 = resize(x, 0.5),
 * x2 = resize(x, 0.25); split(x0, 0.25) | split(x1, 0.5) | x2 concatenated on dim 0 ([35B,3,win,win]); and the ViT's
 * patch extraction, written as the A matrix of the patch-embed GEMM: out[(tile * P + py * g + px)][c * ps^2 + ky * ps + kx]
 * in `precision`'s storage type. x [B,3,S,S] fp32 with S = 4 * window. rows_out / cols_out receive the matrix shape (pass
 * x_dev = out_dev = NULL to query it). force_generic != 0 selects the grid-stride kernel that serves
 * InterpolationMethod::Burn also for Custom (the one-read LDS-staged kernel is the default for Custom, patch 16). */
int md_op_pyramid_patchify(md_device_t dev, const float* x_dev, int B, int S, int window, int patch, int method, int precision,
                           int force_generic, void* out_dev, int* rows_out, int* cols_out, void* stream);
/* `resize_bilinear(tensor, [oh, ow], _)` of the Depth-Anything-v3 head (depth_anything3/interpolate.rs:7-47) on the
 * engine's NHWC feature-map layout: in [B,H,W,C] -> out [B,OH,OW,C], elements of `precision`'s storage type (bf16 / f16
 * / f32), C a multiple of 8. method MD_INTERP_BURN = align_corners=True (what that head uses), MD_INTERP_CUSTOM = False. */
int md_op_resize_nhwc(md_device_t dev, const void* in_dev, int B, int H, int W, int C, void* out_dev, int OH, int OW,
                      int method, int precision, void* stream);
/* `resize_bilinear_scale` (interpolate.rs:136-145): writes the output dims to oh/ow. */
int md_op_resize_output_size(int H, int W, float scale_h, float scale_w, int* oh, int* ow);
/* `DepthProEncoder::split` (encoder.rs:190-232): fp32 NCHW [B,C,S,S] -> [steps^2*B,C,win,win]. */
int md_op_split(md_device_t dev, const float* in_dev, int B, int C, int S, int window, float overlap,
                float* out_dev, int* steps_out, void* stream);
/* `DepthProEncoder::merge` (encoder.rs:234-282): fp32 NCHW tiles -> stitched map. */
int md_op_merge(md_device_t dev, const float* in_dev, int tiles, int C, int h, int w, int batch, int padding,
                float* out_dev, int* out_h, int* out_w, void* stream);
/* LayerNorm over the last dim (burn nn::LayerNorm as used by burn_dino): x[rows,D] fp32 -> fp32. */
int md_op_layernorm(md_device_t dev, const float* x_dev, const float* gamma_dev, const float* beta_dev, int rows,
                    int D, float eps, float* out_dev, void* stream);
/* Linear: out[M,N] = act(x[M,K] @ w[N,K]^T + bias), fp32 in/out; operands rounded per
 * `precision`. act: 0 none, 1 relu, 2 gelu(erf). (burn nn::Linear) */
int md_op_linear(md_device_t dev, const float* x_dev, const float* w_dev, const float* bias_dev, int M, int N,
                 int K, int act, int precision, float* out_dev, void* stream);
/* Same with an explicit GEMM tile configuration (0 = 256x256, 1 = 128x128, 2 = 256x32, 99 = auto);
 * lets the parity tests and the bench exercise every tile shape. */
int md_op_linear_tile(md_device_t dev, const float* x_dev, const float* w_dev, const float* bias_dev, int M, int N,
                      int K, int act, int precision, int tile, float* out_dev, void* stream);
/* Multi-head attention core on a fused qkv tensor [T, N, 3*heads*64] (timm layout), fp32 in/out:
 * softmax(q k^T / 8) v -> [T, N, heads*64]. (burn_dino attention, quiet_softmax=false) */
int md_op_attention(md_device_t dev, const float* qkv_dev, int T, int N, int heads, int precision, float* out_dev,
                    void* stream);
/* Conv2d 3x3 stride 1 pad 1 (burn nn::Conv2d): fp32 NCHW in/out, w [Cout,Cin,3,3]. */
int md_op_conv3x3(md_device_t dev, const float* x_dev, const float* w_dev, const float* bias_dev, int B, int Cin,
                  int H, int W, int Cout, int pre_relu, int precision, float* out_dev, void* stream);
/* ConvTranspose2d k=2 s=2 (burn nn::ConvTranspose2d): fp32 NCHW, w [Cin,Cout,2,2]. */
int md_op_deconv2x2(md_device_t dev, const float* x_dev, const float* w_dev, const float* bias_dev, int B, int Cin,
                    int H, int W, int Cout, int precision, float* out_dev, void* stream);
/* Generic small Conv2d (any k/stride/pad), fp32 exact; the FOV head path (fov.rs:16-49). */
int md_op_conv2d_direct(md_device_t dev, const float* x_dev, const float* w_dev, const float* bias_dev, int B,
                        int Cin, int H, int W, int Cout, int k, int stride, int pad, int relu, float* out_dev,
                        void* stream);
/* `fovy_from_fovx_rad` (mod.rs:370-414) + focal length (mod.rs:330-336) on host scalars. */
int md_op_fov_to_focal(float fovx_deg, int H, int W, float* focal_px, float* fovy_rad);

/* Kernel micro-benchmark: times `iters` launches of the GEMM kernel (random bf16/f32 operands resident
 * in HBM, plain store epilogue, out element = operand type) with HIP events on the launch stream and
 * returns the average milliseconds per launch. mode: 0 dense GEMM [M,K]x[N,K]^T; 1 conv3x3 over an
 * NHWC [1,H,W,K] image with M = H*W (pass H in `aux0`, W in `aux1`), N = Cout.
 * Timing-only ablation flags ride in `tile >> 8` (results are then meaningless): 1 no in-loop global loads, 2 every
 * k-tile re-reads k-tile 0, 4 bias + GELU epilogue (fc1), 8 pixel-shuffle epilogue of a k2s2 deconvolution (mode 0:
 * `aux0` x `aux1` = input pixel grid, N = 4*Cout), 16 fp32 read-modify-write epilogue (proj / fc2), 32 per-workgroup
 * phase stamps of one launch printed to stderr (s_memrealtime at seven points + shader clock around the main loop),
 * 64 no global stores, 128 no staging writes. tools/kernel_bench.py names the combinations. */
int md_bench_gemm(md_device_t dev, int mode, int M, int N, int K, int aux0, int aux1, int precision, int tile, int iters,
                  float* avg_ms);
/* Host-only diagnostic (no GPU work): the tile the engine's launch cost model picks for a dense [M,K] x [N,K]^T GEMM in
 * `precision` -- 0 = 256x256, 1 = 128x128, 2 = 256x32 (N <= 32), 3 = 128x64, 4 = 64x64. */
int md_gemm_pick_tile(int M, int N, int K, int precision);
/* Host-only diagnostic: launches of the 64 x 64 GEMM kernel that split their contraction over wave groups inside the workgroup
 * (gemm_kernel's KSPLIT) since the library was loaded (modulo 2^31) -- lets a test check that the form it
 * means to exercise actually ran. */
int md_gemm_ksplit_launches(void);
/* PROCESS-WIDE A/B switch (default 1; returns the previous value): may the lean 2-byte store epilogues of the 256 x 256 GEMM kernel
 * (fc1, the q | k tiles of qkv, convolutions without residual inputs) store straight from the accumulator layout -- the W tile's
 * LDS image in a permuted row order, one 16-byte store per lane and (m-block, column half) -- instead of staging the tile through
 * LDS? Same values, same bits; for benches and the bit-identity test. Graphs captured before a change keep
 * their form. */
int md_debug_gemm_direct_store(int on);
/* PROCESS-WIDE A/B switch (a mask, default 15; returns the previous value): which launches of the 256 x 256 GEMM kernel with >= 1024 tiles (the QKV projection: 768)
 * may run as a persistent tile loop -- one workgroup per CU, the next tile's first k-tile requested before the current tile's epilogue
 * (DESIGN.md section 5.1.2): 1 = the fc1 form (dense A, bias (+ LayerNorm fold) + GELU, direct stores: gemm256p_kernel), 2 = the fused
 * QKV projection (one-plane types), 4 = the read-modify-write GEMMs proj / fc2 (gemm256r_kernel), 8 = the implicit 3 x 3 GEMMs of the decoder's residual units (bias; with or without residual
 * inputs / a relu'd second output). Same arithmetic, same bits. */
int md_debug_gemm_persistent(int mask);
/* Timing switch of the tile loops: half of every XCD's workgroups start `ticks` (10 ns each) after the other half, so that one half's
 * epilogues (proj / fc2: the fp32 residual stream's read + write, HBM-bound) meet the other half's main loops instead of each other.
 * which: 0 the read-modify-write loop at <= 16 k-tiles per tile (proj; default 2000), 1 the same at more (fc2; 0), 2 the fc1 loop (0),
 * 3 the QKV loop (0). Same bits. MD_ERR_INVALID_ARG for another `which` or negative ticks. */
int md_debug_gemm_stagger(int which, int ticks);
/* Same for the fused bf16 attention kernel: T sequences of n_tokens, `heads` heads of 64. */
int md_bench_attention(md_device_t dev, int T, int n_tokens, int heads, int iters, float* avg_ms);
/* The same with the operand type (MD_PREC_BF16 | MD_PREC_F16) and the range of the random q / k values, uniform in
 * +-qk_scale (q is taken as already carrying the softmax scale): 0.7 gives logits of a few units (the bf16 kernel's fast
 * body), 4.0 and above logits beyond its +-32 check (the running-maximum body). */
int md_bench_attention_ex(md_device_t dev, int T, int n_tokens, int heads, int precision, float qk_scale, int iters,
                          float* avg_ms);
/* The same on CALLER-supplied operands: qkv_dev = [T, n_tokens, 3 * heads * 64] fp32 on the device, rows q | k | v as the fused QKV
 * projection of /root/reference/src/model/depth_pro/layers/vit.rs:45-68 (burn_dino's attention) produces them; q is scaled by
 * 1/sqrt(64) inside. *redo_units_per_launch (may be NULL) = the (sequence, head) units per launch whose row sums left the assembly
 * kernel's fast range and were recomputed by the running-maximum body (-1: the assembly kernel is not in use). For stress operands
 * with outlier logits (plain softmax, vit.rs:60: outliers are legal inputs). */
int md_bench_attention_qkv(md_device_t dev, const float* qkv_dev, int T, int n_tokens, int heads, int precision, int iters,
                           float* avg_ms, long* redo_units_per_launch);
/* PROCESS-WIDE (round 5: per host thread): may bf16 attention launches of exactly 577 tokens (Depth Pro: 576 patches + the class
 * token) take the assembly-owned gfx950 kernel (kernels/attn577_gfx950.s)? Default 1; returns the previous value. 0 runs the HIP
 * kernel that every other shape runs -- an A/B switch for benches and parity tests, not a numerics option: both forms compute the
 * same sums (the assembly kernel adds the rounded probabilities on the matrix pipe). Graphs captured before a change keep their form. */
int md_debug_attention_asm(int on);
/* Launches of the assembly-owned attention kernel since the library was loaded: what a bench line may say about the form it timed. */
long md_debug_attention_asm_launches(void);
/* (sequence, head) units the assembly kernel flagged for the running-maximum body on `dev` since the last reset (reset != 0 clears
 * the counter). Synchronises the device. -1: the code object is not loaded there. 0 over a whole run = the fast body served every unit. */
long md_debug_attention_redo_units(md_device_t dev, int reset);

/* ---- multi-GPU: RCCL over xGMI behind the C ABI ------------------------------------------------------------------
 * BASELINE north_star: "independent images shard naturally across the 8 GPUs of one node with RCCL broadcast of weights and
 * gather of depth maps over xGMI", reached by the (Rust) host through this FFI layer. One process (or thread) per GPU;
 * DepthPro::infer itself never communicates -- B is a pure batch dimension (encoder.rs:216-225,249-255). The reference has
 * no collectives: these calls sit next to `DepthPro::load` / `infer` in a multi-GPU host, see INTEGRATION.md
 * section 4. All buffers are device pointers of the communicator's device; transfers are asynchronous on `stream` (NULL =
 * the device's stream, the one md_depth_pro_infer uses for stream == NULL), so they order with the inference around them. */
typedef struct md_comm_s* md_comm_t;
#define MD_COMM_ID_BYTES 128
/* A fresh rendezvous id (ncclUniqueId). The root creates it; the host hands the 128 bytes to every rank out of band
 * (environment, file, its own RPC) -- the one thing the library cannot do for a multi-process launch. */
int md_comm_unique_id(uint8_t id[MD_COMM_ID_BYTES]);
/* Collective over all ranks: joins the communicator of `world_size` ranks as `rank` on this device. */
int md_comm_init_rank(md_device_t dev, const uint8_t id[MD_COMM_ID_BYTES], int world_size, int rank, md_comm_t* out);
int md_comm_rank(md_comm_t c, int* rank, int* world_size);
/* The number of ranks RCCL itself reports for the communicator (`ncclCommCount`). */
int md_comm_count(md_comm_t c, int* ranks_seen);
int md_comm_destroy(md_comm_t c);
/* Collective: `DepthPro::load` happens on `root` only; its fp32 parameter arena is broadcast into every rank's model (same
 * config) in 1-GiB buckets and every rank commits (packs its own MFMA operand copies). Synchronises the device's stream. */
int md_comm_broadcast_weights(md_comm_t c, md_model_t m, int root);
/* Collective: the root holds `world_size` shards of `elems_per_rank` floats back to back (rank-major; e.g. [world*B,3,H,W]);
 * every rank -- the root too -- ends up with its shard in `shard_dev`. One group of ncclSend / ncclRecv (xGMI is point to
 * point: the root's links carry the shards in parallel). `all_dev` is ignored on the other ranks. */
int md_comm_scatter_images(md_comm_t c, const float* all_dev, float* shard_dev, size_t elems_per_rank, int root, void* stream);
/* Collective: the inverse for the results (depth [B,H,W] per rank -> [world*B,H,W] on the root). */
int md_comm_gather_depth(md_comm_t c, const float* shard_dev, float* all_dev, size_t elems_per_rank, int root, void* stream);

/* Tile-parallel `DepthPro::infer` for ONE call (second mode: the only way more GPUs shorten the latency of a
 * single image). Every rank of `comm` calls it with its replica of the same committed weights (md_comm_broadcast_weights):
 *   1. the root's input [B,3,H,W] (host or device pointer; NULL on the other ranks) is broadcast to every rank's staging buffer;
 *   2. rank r runs pyramid + patchify (0.02 ms) and the three ViT encoders on sequences [37B*r/n, 37B*(r+1)/n) of the 37 B
 *      (the sliding-window tiles of layers/encoder.rs:329-348 and the image / fov sequences never interact before `merge`);
 *   3. the final tokens of every window and the two hook outputs of its high-resolution tiles (encoder.rs:375-390) go to the
 *      root as ONE group of ncclSend / ncclRecv (B = 1, bf16: 100 MB in total, 1/n of it per link);
 *   4. the root runs merge, encoder tail, decoder, head and FOV network and fills the outputs; the other ranks return after
 *      their send (their output pointers are ignored and may be NULL).
 * The result on the root is bit-identical to md_depth_pro_infer on one GPU. Eager only (no graph replay). */
int md_comm_depth_pro_infer_tiles(md_comm_t comm, md_model_t model, const float* nchw, int B, int H, int W, int in_kind,
                                  float* depth, float* focallength_px, float* fovx_deg, float* fovy_rad, int out_kind,
                                  int root, void* stream);
/* The tile-parallel call with a LOOPBACK transport: the `parts` ranks are `parts` inference contexts on ONE device (a model and its
 * md_model_fork contexts, or separately created models with the same weights), each runs ITS window of the ViT stage in its own
 * workspace, and where md_comm_depth_pro_infer_tiles would ncclSend / ncclRecv a part's final tokens and hook rows, the root copies
 * them out of that part's workspace (sender and receiver sizes are compared: MD_ERR_INVALID_ARG on a mismatch). Everything of the
 * N > 1 code path except RCCL itself runs -- window clipping per rank, the segment table, the root-only tail -- on one GPU; the
 * result is bit-identical to md_depth_pro_infer. Test entry (single-GPU boxes cannot form a two-rank communicator). */
int md_depth_pro_infer_tiles_loopback(const md_model_t* models, int parts, int root, const float* nchw, int B, int H, int W,
                                      int in_kind, float* depth, float* focallength_px, float* fovx_deg, float* fovy_rad,
                                      int out_kind, void* stream);

/* ---- host-only utilities (no GPU needed) ---------------------------------------------------- */
/* The parameter inventory of `DepthPro::new` for a config: returns the number of parameters; for
 * 0 <= index < count also the name (static storage, valid until the next call from this thread),
 * element count and the uniform range [lo, hi) the seeded initialiser draws from. */
int md_param_inventory(const md_depth_pro_cfg* cfg, int init_scheme, int index, const char** name, size_t* count,
                       float* lo, float* hi);
/* The seeded generator itself: `count` values of stream (name, seed) in [lo, hi). */
int md_uniform_stream(const char* name, uint64_t seed, size_t count, float lo, float hi, float* out_host);
/* Split geometry (encoder.rs:196-206) and feature padding (encoder.rs:28-38). */
int md_split_geometry(int image_size, int window, float overlap, int* stride, int* steps);
int md_feature_padding(int window, int stride, int feature_size);

/* Per-kernel-family timing (HIP events recorded on the stream each kernel is launched on; enable
 * first). Entries accumulate over infer calls until read; reading sums them by family name into
 * names/ms/calls (capacity `cap`, count in *n) and clears them. */
int md_model_enable_timing(md_model_t m, int enable);
/* Restrict the events to ONE kernel family (e.g. "fc1_gemm"); NULL or "" = every family. Two event records per launch
 * cost about 0.7 % of a Depth Pro step when every one of its ~226 launches is timed; a throughput measurement times the
 * family it reports against its roofline inside the timed region and the rest in a separate pass. */
int md_model_set_timing_filter(md_model_t m, const char* family);
int md_model_read_timing(md_model_t m, const char** names, float* ms, int* calls, int cap, int* n);
/* Family name of every kernel launch recorded since timing was enabled / last read, in launch order
 * (one entry per kernel launch; does not clear). Lets a rocprofv3 trace be mapped to families. */
int md_model_read_launch_order(md_model_t m, const char** names, int cap, int* n);

#ifdef __cplusplus
}
#endif
#endif /* MI_DEPTH_H */
