// Launchers for the HBM-bound and small kernels of the Depth Pro path (device pointers only).
#pragma once

#include "../md_common.h"

namespace md {

// a1  rgb_to_input_tensor (src/inference.rs:79-121): u8 HWC -> f32 NCHW, one image.
int launch_rgb_to_input(const uint8_t* rgb, int w, int h, float* out, hipStream_t s);

// a2  bilinear resize, fp32 NCHW (interpolate.rs:54-121). method: MD_INTERP_*.
// post: 0 none, 1 = 1/clamp(v,1e-4,1e4) (DepthPro::infer tail, mod.rs:356).
int launch_resize_bilinear(const float* in, int planes, int H, int W, float* out, int OH, int OW, int method,
                           int post, hipStream_t s);

// a2+a3 fused: image pyramid (x1 = resize 0.5, x2 = resize 0.25; encoder.rs:326-327), sliding
// window split (encoder.rs:190-232) and 16x16 patch extraction, written as the A matrix of the
// patch-embed GEMM: out[(tile*P + py*g + px)][c*ps*ps + ky*ps + kx], T = bf16 or float.
// Tile order = reference cat order: level0 (j*steps0+i)*B+b, then level1, then level2 (x2).
struct PyramidGeom {
  int B, S, win, ps;         // S = 4*win input size, ps = ViT patch size
  int steps0, stride0;       // level 0 (overlap .25)
  int steps1, stride1;       // level 1 (overlap .5)
  int method;                // MD_INTERP_*
};
// force_generic: the grid-stride kernel that serves InterpolationMethod::Burn / odd geometries also for Custom (parity tests).
int launch_pyramid_patchify(const float* x, const PyramidGeom& g, void* patches, int prec, hipStream_t s, bool force_generic = false);

// Generic ViT patch extraction (any patch size, e.g. 14 for DA3): fp32 NCHW [B,3,H,W] ->
// A[(b*ph + py)*pw + px][c*ps*ps + ky*ps + kx], row length Kp >= 3*ps*ps (tail zero-filled).
// cls_x != nullptr: the launch also writes the cls rows (cls + pos0) and zero padding rows of the fp32 residual stream cls_x
// [B * S, D] (launch_cls_init's work for one sequence group)
int launch_patchify(const float* x, int B, int H, int W, int ps, int Kp, void* out, int prec, hipStream_t s, float* cls_x = nullptr,
                    int S = 0, int n_tokens = 0, int D = 0, const float* cls = nullptr, const float* pos0 = nullptr);

// Bilinear resize of an NHWC tensor (element type by prec), align_corners per `method`
// (MD_INTERP_BURN = align_corners=True, depth_anything3/interpolate.rs:7-47), optional per-pixel
// fp32 addend table [OH*OW, C] (already scaled; the DA3 UV position embedding, dpt.rs:799-828).
int launch_resize_nhwc(const void* in, int B, int H, int W, int C, long ld_in, void* out, int OH, int OW, long ld_out,
                       int method, const float* addend, int prec, hipStream_t s);

// Border correction of a 3x3 convolution (pad 1) that was composed with a preceding biased 1x1 convolution: the 1x1's bias
// reaches the output only through taps inside the map, so the composed bias is one of nine position-class vectors
// bias9[3*ry + rx][C] (ry, rx: 0 first, 1 interior, 2 last row / column). The convolution itself adds the interior vector;
// this adds bias9[class] - bias9[4] to the 2H + 2W - 4 border pixels of every image of the NHWC map (needs H, W >= 2).
int launch_border_bias_fix(void* map, int B, int H, int W, int C, long ld, const float* bias9, int prec, hipStream_t s);

// a3 stand-alone split on fp32 NCHW (debug tap / md_op_split).
int launch_split(const float* x, int B, int C, int S, int win, int stride, int steps, float* out, hipStream_t s);
// a6 stand-alone merge on fp32 NCHW (encoder.rs:234-282).
int launch_merge(const float* tiles, int B, int C, int h, int w, int steps, int pad, float* out, int OH, int OW,
                 hipStream_t s);

// cls token + pos_embed[0] rows and zero padding rows of the fp32 residual stream.
// x[seq*S + 0] = cls_g + pos_g[0]; x[seq*S + n_tokens ..] = 0, group g by sequence ranges.
struct SeqGroups {
  int ngroups;
  int seq0[4];  // first sequence of each group
  int nseq[4];
  const float* a[4];  // per-group pointers (meaning depends on the kernel)
  const float* b[4];
};
int launch_cls_init(float* x, int nseq_total, int S, int n_tokens, int D, const SeqGroups& g, hipStream_t s);

// K3 LayerNorm over D (biased variance), fp32 rows -> T rows (bf16/f32) or fp32 (out_f32).
// gamma/beta per group (a = gamma, b = beta); NULL gamma = non-affine. Rows grouped by sequence.
// prec == MD_PREC_FP8 (and !out_f32): rows of OCP e4m3 bytes, value * fp8_inv_scale, saturating.
// tok0 != nullptr: row 0 of every sequence is first replaced by tok0[seq * tok0_stride ..] (written back through x_rw == x)
int launch_layernorm(const float* x, void* out, long rows, int D, float eps, int S, const SeqGroups& g, int prec,
                     int out_f32, hipStream_t s, float fp8_inv_scale = 1.f, const float* tok0 = nullptr, int tok0_stride = 0,
                     float* x_rw = nullptr);
// The vectors of a LayerNorm folded into the linear layer behind it (gemm.h GemmParams::ln_*): c[n] = sum_k gamma[k] Wr[n][k],
// d[n] = bias[n] + sum_k beta[k] Wr[n][k] with Wr = the weight as its MFMA operand holds it (rounded to bf16 / f16; split-half: hi + lo),
// so that mu * c cancels the mean's share of the accumulators exactly as the operands produced it. W = fp32 master [N][K]; fp64 sums.
int launch_ln_fold_vectors(const float* W, const float* gamma, const float* beta, const float* bias, int N, int K, int prec, float* c,
                           float* d, hipStream_t s);
// LayerNorm fold: the per-row partials the producer GEMM wrote ([rows][4][2]: mean and centred sum of squares of each 256-column tile)
// combined (Chan: every term non-negative) into ab[row] = (rstd, -mu * rstd) for the consumer GEMM's epilogue. D = 1024.
// ab[row] = (a, b) for `rows` rows (the diagnostic form of the fold: neutral statistics)
int launch_fill_pairs(float* ab, long rows, float a, float b, hipStream_t s);
int launch_ln_finish(const float* parts, float* ab, long rows, float inv_n, float eps, hipStream_t s);
// fp32 rows -> T rows (hooks: un-normalised tokens), same row layout.
int launch_convert_rows(const float* x, void* out, long count, int prec, hipStream_t s, int width = 0);

// layout converters (debug taps, stand-alone ops)
// ld: logical elements between consecutive output pixels (0 = C); padding columns are left untouched
int launch_nchw_to_nhwc(const float* in, int B, int C, int H, int W, void* out, int prec, int relu, hipStream_t s, long ld = 0);
// head_debug's un-fused tail: pre_out = <relu_map pixel, w> + bias, canonical = relu(pre_out); fp32 [pixels] each (either may be NULL)
int launch_head_tail_debug(const void* relu_map, long ld, long pixels, int C, const float* w, const float* bias, float* pre_out,
                           float* canonical, int prec, hipStream_t s);
int launch_nhwc_to_nchw(const void* in, int B, int C, int H, int W, long ld, int coff, float* out, int prec,
                        hipStream_t s);
// width: logical row width -- needed by MD_PREC_F16X2, whose rows are [hi: width | lo: width] (ignored otherwise)
int launch_rows_to_f32(const void* in, long count, float* out, int prec, hipStream_t s, int width = 0);
int launch_f32_to_rows(const float* in, long count, void* out, int prec, hipStream_t s, int width = 0);

// Small direct convolution, NHWC, fp32 accumulate (FOV head, fov.rs:16-49).
// in: element type by in_prec (MD_PREC_BF16 -> bf16, MD_PREC_F32 -> float); w packed
// [Cout][kh][kw][Cin] f32; add: optional f32 NHWC tensor added to the INPUT (fov.rs:185).
// out_ld: elements between consecutive output pixels (0 = Cout): lets several small heads write the columns of one row.
int launch_conv_direct(const void* in, int in_prec, const float* add, int B, int H, int W, int Cin, const float* w,
                       const float* bias, int Cout, int k, int stride, int pad, int relu, float* out, hipStream_t s, int out_ld = 0);

// ---- fp8 (OCP e4m3) MFMA operands (MD_PREC_FP8) ----
// out[i] = e4m3(clamp(in[i] * inv_scale, +-448)); n a multiple of 4.
int launch_f32_to_fp8(const float* in, long n, float inv_scale, void* out, hipStream_t s);
// W [N][K] f32 -> e4m3 [N][Kp] (K zero-padded to Kp) with one scale per output row: scale[n] = amax_n / 448 (1 for an
// all-zero row); the GEMM epilogue multiplies the accumulator by activation_scale * scale[n].
int launch_pack_fp8_rows(const float* w, int N, int K, int Kp, void* out, float* scale, hipStream_t s);

// ---- Depth-Anything-v3 `small` backbone extras (burn_dino, restated -- see oracle/da3_ref.py) ----
// Per-head affine LayerNorm(64) of q and k followed by the 2-D rotary embedding, in place on qk [rows, 2D] T.
// rope_cos/rope_sin: [max_pos + 1][16] tables (angle = pos * base^(-f/16)). global_pos: every patch at (1,1).
int launch_qk_norm_rope(void* qk, long rows, int S, int n_tokens, int D, int heads, int pw, const float* q_gamma,
                        const float* q_beta, const float* k_gamma, const float* k_beta, float eps, const float* rope_cos,
                        const float* rope_sin, int global_pos, float q_scale, int prec, hipStream_t s);
// x[b*S + 0, :] = src[b*src_stride + 0:D] for every sequence (the camera token replaces the cls slot): src_stride 0 = the one
// learned token for all, D = one encoded token per image (`infer_with_camera`)
int launch_set_token0(float* x, int nseq, int S, int D, const float* src, hipStream_t s, int src_stride = 0);

// ---- Depth-Anything-v3 camera encoder (kernels/camera.hip; camera.rs:50-110) ----
#define MD_CAM_MAX_VIEWS 16
struct CamEncW {
  static constexpr int kMaxDepth = 8;
  const float *fc1_w, *fc1_b, *fc2_w, *fc2_b;  // PoseBranch: [D/2, 9], [D, D/2]
  const float *tn_g, *tn_b, *on_g, *on_b;      // token_norm, trunk_norm
  struct Blk {
    const float *n1g, *n1b, *n2g, *n2b, *qkv_w, *qkv_b, *proj_w, *proj_b, *ls1, *fc1_w, *fc1_b, *fc2_w, *fc2_b, *ls2;
  } blk[kMaxDepth];
  int depth;
};
// camera DECODER (camera.rs:143-199, 281-358): cam [B, din] fp32 -> pose [B, 9] (+ extrinsics [B, 12], intrinsics [B, 9], either may be
// null); h1 / h2 = [B, din] scratch. Three launches: two multi-workgroup din x din linears (+ReLU), one heads + tail kernel.
struct CamDecW {
  const float *w1, *b1, *w2, *b2, *wt, *bt, *wq, *bq, *wf, *bf;
};
int launch_camera_decoder(const float* cam, int B, int din, const CamDecW& w, int H, int W, float* h1, float* h2, float* pose, float* extr,
                          float* intr, hipStream_t s);
size_t camera_encoder_scratch_floats(int B, int V, int D);
// extr [B, V, 3, 4] world-to-camera, intr [B, V, 3, 3] (device) -> out [B, D] (device); one workgroup per image
int launch_camera_encoder(const float* extr, const float* intr, int B, int V, int D, int heads, int H, int W, float eps_tok, float eps_blk,
                          const CamEncW& w, float* scratch, float* out, hipStream_t s);
// hook = LayerNorm_head( cat( x_local, LayerNorm_final(x) ) ) -> T rows [rows, 2D]; optional raw token-0 concat
// [nseq, 2D] f32 (the camera feature).
int launch_hook_cat_ln(const float* x_local, const float* x, long rows, int S, int n_tokens, int D, const float* norm_g,
                       const float* norm_b, float eps_final, const float* head_g, const float* head_b, float eps_head,
                       void* out, float* cam_out, int prec, hipStream_t s);
// pose encoding [B,9] = (t3, quat xyzw, fov_h, fov_w) -> world-to-camera extrinsics [B,3,4], intrinsics [B,3,3]
// (camera.rs:281-416)
int launch_pose_to_camera(const float* pose, int B, int H, int W, float* extrinsics, float* intrinsics, hipStream_t s);

// fov degrees -> focal length / fovy / ratio (mod.rs:330-346, 370-414). All [B] f32.
int launch_fov_post(const float* fov_deg, int B, int H, int W, float* focal_px, float* fovy_rad, float* ratio,
                    hipStream_t s);
// inv = canonical * ratio[b]; post 1: depth = 1/clamp(inv); post 0: keep inv (then resize).
int launch_depth_post(const float* canonical, const float* ratio, int B, long hw, float* out, int post, hipStream_t s);

// softmax over rows of fp32 scores (fp32 attention path): row length ld, valid n, scale.
int launch_softmax_rows(float* s, long rows, int n_valid, int ld, float scale, hipStream_t st);

// test helpers for the stand-alone attention op: fused fp32 qkv [T, N, 3*heads*64] (timm layout) ->
// engine layout (qk rows padded to SS per sequence, V transposed), and padded rows -> [T, N, D] fp32.
int launch_qkv_split(const float* qkv, int T, int N, int heads, int SS, int kpad, void* qk, void* vT, float q_scale, int prec,
                     hipStream_t s);
int launch_unpad_rows(const void* in, int T, int N, int SS, int D, float* out, int prec, hipStream_t s);

// The softmax scale the fused attention expects to find folded into q: head_dim^-0.5 * log2(e) (head_dim = 64). The QKV
// epilogue (GemmParams::qscale), the q/k-norm + RoPE kernel and the stand-alone op's splitter apply it before q's one
// rounding to the operand type; the kernel then computes p = 2^(q'.k - m). The fp32 three-kernel path keeps q unscaled.
constexpr float kAttnQScale = 0.125f * 1.4426950408889634f;
inline float attn_qscale(int prec) { return prec == MD_PREC_F32 ? 1.0f : kAttnQScale; }

// fused multi-head attention on bf16 / f16 operands (K5; prec = MD_PREC_BF16 | MD_PREC_F16): qk [rows, 2D] (q | k),
// q PRE-SCALED by kAttnQScale, vT [seq][heads][64][kpad], out [rows, D].
// out_fp8_inv > 0 (bf16 only): the output rows are OCP e4m3 bytes (value * out_fp8_inv, saturating).
// prec = MD_PREC_F16X2: split-half operands -- qk rows [q_hi | q_lo | k_hi | k_lo] (4D wide), the lo plane of V^T `v_plane`
// elements behind its hi plane, out rows [hi: D | lo: D]; scores on three MFMA terms, P.V on two or three (attention.hip).
// redo: attention_redo_ints(nseq * heads) zero-initialised ints owned by the caller's context (one buffer per stream that may run attention
// at a time: the flags, then the compacted list of raised ones), or null. With it, bf16 launches of exactly 577 tokens take the assembly-owned kernel (attn577_gfx950.s) once attention_asm_prepare()
// has loaded its code object on the device; the flags it raises are consumed and cleared inside the same call.
int launch_attention(const void* qk, const void* vT, void* out, int nseq, int S, int n_tokens, int heads, int D,
                     int kpad, int prec, hipStream_t s, float out_fp8_inv = 0.f, long v_plane = 0, int* redo = nullptr);
// Loads the embedded code object on the CURRENT device (idempotent; not capturable -- call it when a context is created).
int attention_asm_prepare();
// Process-wide: may launch_attention take the assembly kernel? Default 1; returns the previous value (A/B runs, the bench).
int attention_allow_asm(int on);
// launches of the assembly kernel since the library was loaded (what a bench line says about the form it measured)
long attention_asm_launches();
// ints of a `redo` buffer for nunits = nseq * heads units
int attention_redo_ints(int nunits);
// units the assembly kernel flagged on the current device since the last reset (synchronises the device; -1: not loaded)
long attention_asm_redo_units(int reset);
// Per host thread: may launch_attention pick its small-launch form (64 queries x two key groups per workgroup for launches of few
// workgroups over long sequences -- another summation order than the plain form's)? Default 1. Returns the previous value. A model
// in batch-invariant mode (md_model_set_option) runs with 0: one image gives the same bits alone and inside a batch.
int attention_allow_small(int on);
struct AttnSmallScope {
  explicit AttnSmallScope(int on) : prev_(attention_allow_small(on)) {}
  ~AttnSmallScope() { attention_allow_small(prev_); }
  AttnSmallScope(const AttnSmallScope&) = delete;
  AttnSmallScope& operator=(const AttnSmallScope&) = delete;

 private:
  int prev_;
};

}  // namespace md
