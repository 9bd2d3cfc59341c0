// Depth-Anything-v3 on the same kernel set: `metric_large` (ViT-L/14 + mono DPT head) and `small` (ViT-S/14 with
// the burn_dino extras -- QK-norm, 2-D RoPE, local/global alternation, camera token, concatenated hooks -- + dual
// DPT head with the aux/ray branch + camera decoder; reference: mod.rs:158-216, dpt.rs:153-513, camera.rs:113-199,
// 281-416; the backbone extras are restated from the public model, see oracle/da3_ref.py).
//
// reference: src/model/depth_anything3/mod.rs:288-291,495-624 (infer), dpt.rs:515-731 (mono head),
// dpt.rs:784-932 (UV position embedding), dpt.rs:1194-1301 (fusion blocks), interpolate.rs:7-47.
// The backbone is burn_dino's plain DINOv2 ViT (un-vendored; restated, parity unpinned): hooks are the
// final-LayerNorm'ed outputs of the hook blocks with the cls token dropped.
//
// Layout: tokens [b*SS + token, D] (fp32 residual, T operands), head feature maps NHWC. The mono head is
// 1x1 conv (GEMM over gathered token rows, + UV position table as an epilogue addend) -> k4s4 / k2s2
// ConvTranspose (GEMM + pixel shuffle) or 3x3 stride-2 conv (implicit GEMM with stride) -> 3x3 convs,
// residual units and align-corners-true bilinear resizes in NHWC -> fused tail
// exp(w_out . relu(conv1) + b) written straight to the depth output.
#include <cmath>
#include <cstring>

#include "md_engine.h"
#include "md_engine_util.h"

using namespace md;

struct md_model_s::Da3State {
  Da3Cfg cfg;
  int ih = 0, iw = 0;  // configured input size (rows x columns; any multiples of the patch size)
  int ph = 0, pw = 0, P = 0, NT = 0, SS = 0, kpad = 0, Kp = 0, h3h = 0, h3w = 0;
  VitW vit;
  // workspace
  void *patches = nullptr, *xn = nullptr, *qk = nullptr, *vT = nullptr, *ao = nullptr, *hbuf = nullptr;
  float *xres = nullptr, *lnf = nullptr, *scores = nullptr, *xin = nullptr;
  void* hookn[4] = {0, 0, 0, 0};
  void *sp[4] = {0, 0, 0, 0}, *sr[4] = {0, 0, 0, 0}, *rn[4] = {0, 0, 0, 0}, *rnr[4] = {0, 0, 0, 0};
  void *t = nullptr, *x = nullptr, *xr = nullptr, *y = nullptr, *up = nullptr, *o = nullptr, *c1 = nullptr, *c1r = nullptr;
  float* depth_stage = nullptr;
  size_t depth_stage_elems = 0;
  // ---- per-shape tables: the reference's `PosEmbedCache` (dpt.rs:784-833: UV tables built once per (C, h, w, W, H) key and
  //      kept) and burn_dino's position-embedding interpolation, keyed by the input size. `infer` takes any H x W that are
  //      multiples of the patch size (mod.rs:509-520); the first call at a new size builds its tables (host work + uploads),
  //      later calls at that size find them here. At most kMaxShapes sizes are kept (least recently used goes first).
  struct ShapeTables {
    void* pos_stage[4] = {0, 0, 0, 0};  // T [P, cp(oc)] = 0.1 * UV embedding
    float* pos_final = nullptr;         // f32 [H*W, F/2]
    float* pos_aux = nullptr;           // f32 [8ph*8pw, F/2] = 2 * 0.1 * UV table (added twice, dpt.rs:428-435)
    float* pos_used = nullptr;          // [NT, D] position embedding interpolated to this grid (null: the native grid)
    float *rope_cos = nullptr, *rope_sin = nullptr;  // [max(ph, pw) + 2][16]
    std::map<int, int*> tok_index;      // per B: [B*P] -> row b*SS + 1 + p
    unsigned long last_use = 0;
  };
  static constexpr int kMaxShapes = 16;
  std::map<std::pair<int, int>, ShapeTables> shapes;
  unsigned long use_clock = 0;
  long table_builds = 0;              // shapes built so far (md_model_query "da3_shape_builds")
  // the CURRENT shape's tables (aliases into `shapes`)
  void* pos_stage[4] = {0, 0, 0, 0};
  float* pos_final = nullptr;
  std::map<int, int*>* tok_index = nullptr;
  float* pos_used = nullptr;
  int native_grid = 0;
  // ---- `small` (dual head) ----
  std::string hp = "head_mono";       // head parameter prefix
  int din = 0;                        // head input width: D (mono) or 2D (concatenated hooks)
  float* xlocal = nullptr;            // [rows, D] fp32: residual stream after the last LOCAL block
  float* rope_cos = nullptr;          // current shape's tables (aliases)
  float* rope_sin = nullptr;
  float *cam_raw = nullptr, *cam_h1 = nullptr, *cam_h2 = nullptr, *pose = nullptr, *extr = nullptr, *intr = nullptr;
  float* pos_aux = nullptr;
  float *conf_stage = nullptr, *aux_stage = nullptr;  // device staging when the caller wants host outputs
  // camera encoder (`infer_with_camera`): grow-only scratch = staged inputs [B*V*21] | encoded tokens [B*D] | kernel scratch
  float* cam_enc_ws = nullptr;
  size_t cam_enc_cap = 0;
  // `infer_from_tokens`: caller-supplied hook tokens staged as fp32 rows [max_batch * SS + 64, din] (grow-only, zero-filled once)
  float* tok_stage = nullptr;
  size_t tok_stage_cap = 0;
  void *up2 = nullptr, *o2 = nullptr;
  std::vector<float> main_bias, aux_bias;             // output_conv2.conv2.bias, output_conv2_aux.<last>.project.bias
  // ---- MD_PREC_FP8: the four ViT linear layers on e4m3 operands (weights per output channel, static activation scales) ----
  bool fp8 = false;
  char* w8_base = nullptr;                            // one allocation: per block qkv | proj | fc1 | fc2 (e4m3) + their scales
  struct Fp8Block {
    void* w[4] = {0, 0, 0, 0};
    float* s[4] = {0, 0, 0, 0};
  };
  std::vector<Fp8Block> w8;
  static constexpr float kActScale = 8.0f / 448.0f;   // LayerNorm output, attention output
  static constexpr float kHidScale = 16.0f / 448.0f;  // GELU output
};

namespace md {

// dpt.rs:835-932 -- the reference's table, including its transposed pixel index (dpt.rs:879)
static void sincos(int dim, float position, float* out) {
  const int half = dim / 2;
  for (int i = 0; i < half; ++i) {
    const float exponent = half > 0 ? (float)i / (float)half : 0.f;
    const float omega = powf(100.0f, -exponent);
    out[i] = sinf(position * omega);
  }
  const int remaining = dim - half;
  for (int i = 0; i < remaining; ++i) {
    const float exponent = remaining > 0 ? (float)i / (float)remaining : 0.f;
    const float omega = powf(100.0f, -exponent);
    out[half + i] = cosf(position * omega);
  }
}

// returns NHWC [h*w][C] scaled by `ratio`
static std::vector<float> build_pos_table_nhwc(int C, int h, int w, int image_w, int image_h, float ratio) {
  const float aspect = (float)image_w / (float)image_h;
  const float diag = sqrtf(aspect * aspect + 1.0f);
  const float span_x = aspect / diag, span_y = 1.0f / diag;
  const float left_x = -span_x * ((float)w - 1.0f) / (float)w, right_x = span_x * ((float)w - 1.0f) / (float)w;
  const float top_y = -span_y * ((float)h - 1.0f) / (float)h, bottom_y = span_y * ((float)h - 1.0f) / (float)h;
  auto lin = [](float a, float b, int n, int i) { return n <= 1 ? a : a + ((b - a) / ((float)n - 1.0f)) * (float)i; };
  const int xc = C / 2, yc = C - xc;
  std::vector<float> ex((size_t)w * xc), ey((size_t)h * yc);
  for (int i = 0; i < w; ++i) sincos(xc, lin(left_x, right_x, w, i), ex.data() + (size_t)i * xc);
  for (int i = 0; i < h; ++i) sincos(yc, lin(top_y, bottom_y, h, i), ey.data() + (size_t)i * yc);
  std::vector<float> t((size_t)h * w * C);
  // reference: chw[c*h*w + (x_idx*height + y_idx)], then viewed as [C, h, w]
  for (int xi = 0; xi < w; ++xi)
    for (int yi = 0; yi < h; ++yi) {
      const size_t pix = (size_t)xi * h + yi;  // flat pixel index inside the [h, w] view
      float* dst = t.data() + pix * C;
      for (int c = 0; c < xc; ++c) dst[c] = ex[(size_t)xi * xc + c] * ratio;
      for (int c = 0; c < yc; ++c) dst[xc + c] = ey[(size_t)yi * yc + c] * ratio;
    }
  return t;
}

// DINOv2 `interpolate_pos_encoding`: bicubic (A = -0.75, align_corners = False), scale factor
// (grid + 0.1) / native_grid per axis, source index = (dst + 0.5) / scale - 0.5, border-clamped taps.
// burn_dino's version is not visible (parity unpinned); this restates the public DINOv2 code path
// (torch.nn.functional.interpolate with scale_factor, which the oracle calls directly).
static void cubic_coeffs(float t, float w[4]) {
  const float A = -0.75f;
  const float x0 = t + 1.0f, x1 = t, x2 = 1.0f - t, x3 = 2.0f - t;
  w[0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
  w[1] = ((A + 2.0f) * x1 - (A + 3.0f)) * x1 * x1 + 1.0f;
  w[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
  w[3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
}

static std::vector<float> interpolate_pos_embed(const std::vector<float>& pos, int M, int D, int ph, int pw) {
  std::vector<float> out((size_t)(1 + ph * pw) * D);
  memcpy(out.data(), pos.data(), (size_t)D * 4);  // class token position is kept
  const float sy = 1.0f / (((float)ph + 0.1f) / (float)M), sx = 1.0f / (((float)pw + 0.1f) / (float)M);
  for (int oy = 0; oy < ph; ++oy) {
    const float fy = sy * ((float)oy + 0.5f) - 0.5f;
    const int iy = (int)floorf(fy);
    float wy[4];
    cubic_coeffs(fy - (float)iy, wy);
    for (int ox = 0; ox < pw; ++ox) {
      const float fx = sx * ((float)ox + 0.5f) - 0.5f;
      const int ix = (int)floorf(fx);
      float wx[4];
      cubic_coeffs(fx - (float)ix, wx);
      float* dst = out.data() + (size_t)(1 + oy * pw + ox) * D;
      for (int c = 0; c < D; ++c) dst[c] = 0.f;
      for (int a = 0; a < 4; ++a) {
        const int yy = std::min(std::max(iy - 1 + a, 0), M - 1);
        for (int b = 0; b < 4; ++b) {
          const int xx = std::min(std::max(ix - 1 + b, 0), M - 1);
          const float wgt = wy[a] * wx[b];
          const float* src = pos.data() + (size_t)(1 + yy * M + xx) * D;
          for (int c = 0; c < D; ++c) dst[c] += wgt * src[c];
        }
      }
    }
  }
  return out;
}

static void da3_drop_shapes(md_model_s* m);
static int da3_set_shape(md_model_s* m, int H, int W, bool force);

int da3_on_commit(md_model_t m) {
  md_model_s::Da3State* d = m->da3;
  const int D = d->cfg.vit.D, M = d->native_grid;
  if (d->fp8) {  // e4m3 copies of the ViT linear weights, one scale per output channel
    const char* names[4] = {".attn.qkv.weight", ".attn.proj.weight", ".mlp.fc1.weight", ".mlp.fc2.weight"};
    const int nn[4] = {3 * D, D, 4 * D, D}, kk[4] = {D, D, D, 4 * D};
    for (int i = 0; i < d->cfg.vit.depth; ++i)
      for (int j = 0; j < 4; ++j) {
        const float* src = P32(m, "backbone.pretrained.blocks." + std::to_string(i) + names[j]);
        if (!src) MD_FAIL(MD_ERR_FORMAT, "missing ViT weight for the fp8 pack");
        MD_TRY(launch_pack_fp8_rows(src, nn[j], kk[j], kk[j], d->w8[i].w[j], d->w8[i].s[j], m->dev->stream));
      }
    MD_HIP(hipStreamSynchronize(m->dev->stream));
  }
  {
    const Da3Cfg& c = d->cfg;
    d->main_bias.assign(c.output_dim, 0.f);
    MD_HIP(hipMemcpy(d->main_bias.data(), P32(m, d->hp + ".scratch.output_conv2.conv2.bias"), c.output_dim * 4, hipMemcpyDeviceToHost));
    if (c.dual_head) {
      d->aux_bias.assign(c.aux_output_dim, 0.f);
      MD_HIP(hipMemcpy(d->aux_bias.data(), P32(m, d->hp + ".scratch.output_conv2_aux." + std::to_string(c.aux_levels - 1) + ".project.bias"),
                       c.aux_output_dim * 4, hipMemcpyDeviceToHost));
    }
  }
  (void)D; (void)M;
  // the interpolated position embeddings were made from the previous weights: drop every cached shape and rebuild the
  // current one from the committed parameters
  da3_drop_shapes(m);
  return da3_set_shape(m, d->ih, d->iw, /*force=*/true);
}

static int da3_plan(md_model_s* m, bool dry, size_t* total_out) {
  md_model_s::Da3State* d = m->da3;
  const Da3Cfg& c = d->cfg;
  const int B = c.max_batch, D = c.vit.D, F = c.features;
  const int esz = m->esz * m->xm;  // bytes per LOGICAL element of a T tensor (MD_PREC_F16X2: two half planes)
  const size_t SS2 = (size_t)d->ih * d->iw;  // pixels of one input image
  const int* oc = c.out_channels;
  size_t total = 0;
  auto take = [&](size_t bytes) -> void* {
    bytes = align_up(bytes + 256, 256);
    total += bytes;
    return dry ? nullptr : m->ws.take(bytes);
  };
#define DA3_TAKE(field, type, bytes)                                          \
  do {                                                                        \
    void* _p = take(bytes);                                                   \
    if (!dry) {                                                               \
      if (!_p) MD_FAIL(MD_ERR_OOM, "workspace arena exhausted at " #field);   \
      d->field = (type)_p;                                                    \
    }                                                                         \
  } while (0)
  const size_t rows = (size_t)B * d->SS + 64;
  const int ph = d->ph, pw = d->pw;
  const size_t P = d->P;
  auto cp = [&](int ch) { return (size_t)round_up(ch, m->ke); };
  DA3_TAKE(xin, float*, (size_t)B * 3 * SS2 * 4);
  DA3_TAKE(patches, void*, (size_t)B * P * d->Kp * esz);
  DA3_TAKE(xres, float*, rows * D * 4);
  DA3_TAKE(lnf, float*, rows * D * 4);
  DA3_TAKE(xn, void*, rows * D * esz);
  DA3_TAKE(qk, void*, rows * 2 * D * esz);
  DA3_TAKE(vT, void*, (size_t)B * c.vit.heads * 64 * d->kpad * esz);
  m->vt_plane = m->xm == 2 ? (size_t)B * c.vit.heads * 64 * d->kpad : 0;
  DA3_TAKE(ao, void*, rows * D * esz);
  DA3_TAKE(hbuf, void*, rows * 4 * D * esz);
  if (m->prec == MD_PREC_F32) DA3_TAKE(scores, float*, (size_t)B * c.vit.heads * d->SS * d->kpad * 4);
  for (int s = 0; s < 4; ++s) DA3_TAKE(hookn[s], void*, rows * d->din * esz);
  if (c.dual_head) {
    DA3_TAKE(xlocal, float*, rows * D * 4);
    DA3_TAKE(cam_raw, float*, (size_t)B * d->din * 4);
    DA3_TAKE(cam_h1, float*, (size_t)B * d->din * 4);
    DA3_TAKE(cam_h2, float*, (size_t)B * d->din * 4);
    DA3_TAKE(pose, float*, (size_t)B * 9 * 4);
    DA3_TAKE(extr, float*, (size_t)B * 12 * 4);
    DA3_TAKE(intr, float*, (size_t)B * 9 * 4);
    DA3_TAKE(conf_stage, float*, (size_t)B * SS2 * 4);
    DA3_TAKE(aux_stage, float*, (size_t)B * c.aux_output_dim * 64 * ph * pw * 4);
  }
  const size_t px[4] = {(size_t)16 * ph * pw, (size_t)4 * ph * pw, (size_t)ph * pw, (size_t)d->h3h * d->h3w};
  for (int s = 0; s < 4; ++s) {
    DA3_TAKE(sp[s], void*, (size_t)B * P * cp(oc[s]) * esz);
    if (s != 2) DA3_TAKE(sr[s], void*, (size_t)B * px[s] * cp(oc[s]) * esz);
    DA3_TAKE(rn[s], void*, (size_t)B * px[s] * cp(F) * esz);
    DA3_TAKE(rnr[s], void*, (size_t)B * px[s] * cp(F) * esz);
  }
  const size_t big = (size_t)B * 64 * ph * pw * cp(F) * esz;  // 8ph x 8pw
  // the dual head runs its two fusion pyramids in the same launches: every pyramid map holds 2B images (main | aux)
  const size_t pg = c.dual_head ? 2 : 1;
  DA3_TAKE(t, void*, pg * big / 4);
  DA3_TAKE(x, void*, pg * big / 4);
  DA3_TAKE(xr, void*, pg * big / 4);
  DA3_TAKE(y, void*, pg * big / 4);
  DA3_TAKE(up, void*, pg * big);
  DA3_TAKE(o, void*, pg * big);
  if (c.dual_head) {  // ping-pong maps of the aux neck (it runs beside the main tail, which reads `o`)
    DA3_TAKE(up2, void*, big);
    DA3_TAKE(o2, void*, big);
  }
  DA3_TAKE(c1, void*, (size_t)B * 64 * ph * pw * cp(F / 2) * esz);
  DA3_TAKE(c1r, void*, (size_t)B * SS2 * cp(F / 2) * esz);
#undef DA3_TAKE
  if (total_out) *total_out = total + 4096;
  return MD_OK;
}

static void da3_set_geometry(md_model_s* m, int H, int W) {
  md_model_s::Da3State* d = m->da3;
  const ViTDims& v = d->cfg.vit;
  d->ih = H;
  d->iw = W;
  d->ph = H / v.ps;
  d->pw = W / v.ps;
  d->P = d->ph * d->pw;
  d->NT = d->P + 1;
  d->SS = round_up(d->NT, 4);
  d->kpad = round_up(d->NT, 64);
  d->Kp = round_up(3 * v.ps * v.ps, m->ke);
  d->h3h = (d->ph + 2 - 3) / 2 + 1;
  d->h3w = (d->pw + 2 - 3) / 2 + 1;
  m->SS = d->SS;
}

static void da3_free_tables(md_model_s::Da3State::ShapeTables& t) {
  for (int s = 0; s < 4; ++s)
    if (t.pos_stage[s]) (void)hipFree(t.pos_stage[s]);
  if (t.pos_final) (void)hipFree(t.pos_final);
  if (t.pos_aux) (void)hipFree(t.pos_aux);
  if (t.pos_used) (void)hipFree(t.pos_used);
  if (t.rope_cos) (void)hipFree(t.rope_cos);
  if (t.rope_sin) (void)hipFree(t.rope_sin);
  for (auto& kv : t.tok_index) (void)hipFree(kv.second);
  t = md_model_s::Da3State::ShapeTables();
}

static void da3_drop_graphs(md_model_s* m) {  // captured graphs hold workspace / table pointers of the shape they were captured at
  for (auto& kv : m->graphs)
    if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
  m->graphs.clear();
}

static void da3_drop_shapes(md_model_s* m) {
  md_model_s::Da3State* d = m->da3;
  (void)hipDeviceSynchronize();
  for (auto& kv : d->shapes) da3_free_tables(kv.second);
  d->shapes.clear();
  d->tok_index = nullptr;
  da3_drop_graphs(m);
}

// UV position tables, interpolated position embedding and RoPE tables of one input size (PosEmbedCache::add /
// build_positional_embedding, dpt.rs:784-932; burn_dino's pos-embed interpolation, restated: interpolate_pos_embed above)
static int da3_build_tables(md_model_s* m, md_model_s::Da3State::ShapeTables& t) {
  md_model_s::Da3State* d = m->da3;
  const Da3Cfg& c = d->cfg;
  const int IH = d->ih, IW = d->iw, F = c.features, D = c.vit.D, M = d->native_grid;
  const int* oc = c.out_channels;
  hipStream_t st = m->dev->stream;
  float* tmp = nullptr;
  size_t tmp_elems = 0;
  for (int s = 0; s < 4; ++s) tmp_elems = std::max(tmp_elems, (size_t)d->P * round_up(oc[s], m->ke));
  MD_HIP(hipMalloc((void**)&tmp, tmp_elems * 4));
  for (int s = 0; s < 4; ++s) {
    std::vector<float> tab = build_pos_table_nhwc(oc[s], d->ph, d->pw, IW, IH, 0.1f);
    const int ld = round_up(oc[s], m->ke);
    std::vector<float> padded((size_t)d->P * ld, 0.f);
    for (int p = 0; p < d->P; ++p) memcpy(&padded[(size_t)p * ld], &tab[(size_t)p * oc[s]], (size_t)oc[s] * 4);
    MD_HIP(hipMalloc(&t.pos_stage[s], padded.size() * m->esz * m->xm + 256));
    MD_HIP(hipMemcpy(tmp, padded.data(), padded.size() * 4, hipMemcpyHostToDevice));
    MD_TRY(launch_f32_to_rows(tmp, (long)padded.size(), t.pos_stage[s], m->prec, st, ld));
    MD_HIP(hipStreamSynchronize(st));
  }
  MD_HIP(hipFree(tmp));
  {
    std::vector<float> tab = build_pos_table_nhwc(F / 2, IH, IW, IW, IH, 0.1f);
    MD_HIP(hipMalloc((void**)&t.pos_final, tab.size() * 4));
    MD_HIP(hipMemcpy(t.pos_final, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
  }
  if (c.dual_head) {
    // aux head input = neck + 0.1*UV + 0.1*UV (added twice, dpt.rs:428-435)
    std::vector<float> ta = build_pos_table_nhwc(F / 2, 8 * d->ph, 8 * d->pw, IW, IH, 0.1f);
    for (auto& x : ta) x = x + x;
    MD_HIP(hipMalloc((void**)&t.pos_aux, ta.size() * 4));
    MD_HIP(hipMemcpy(t.pos_aux, ta.data(), ta.size() * 4, hipMemcpyHostToDevice));
    // 2-D RoPE tables: angle(pos, f) = pos * base^(-2f/32), f < 16 (fp32 like the oracle)
    const int npos = std::max(d->ph, d->pw) + 2;
    std::vector<float> rc((size_t)npos * 16), rs((size_t)npos * 16);
    for (int pz = 0; pz < npos; ++pz)
      for (int f = 0; f < 16; ++f) {
        const float inv = 1.0f / powf(c.rope_frequency, (float)(2 * f) / 32.0f);
        const float ang = (float)pz * inv;
        rc[(size_t)pz * 16 + f] = cosf(ang);
        rs[(size_t)pz * 16 + f] = sinf(ang);
      }
    MD_HIP(hipMalloc((void**)&t.rope_cos, rc.size() * 4));
    MD_HIP(hipMalloc((void**)&t.rope_sin, rs.size() * 4));
    MD_HIP(hipMemcpy(t.rope_cos, rc.data(), rc.size() * 4, hipMemcpyHostToDevice));
    MD_HIP(hipMemcpy(t.rope_sin, rs.data(), rs.size() * 4, hipMemcpyHostToDevice));
  }
  if (d->ph != M || d->pw != M) {  // position embedding interpolated from the parameter's native grid
    const float* pos_param = P32(m, "backbone.pretrained.pos_embed");
    std::vector<float> pos((size_t)(1 + M * M) * D);
    MD_HIP(hipMemcpy(pos.data(), pos_param, pos.size() * 4, hipMemcpyDeviceToHost));
    std::vector<float> ip = interpolate_pos_embed(pos, M, D, d->ph, d->pw);
    MD_HIP(hipMalloc((void**)&t.pos_used, ip.size() * 4));
    MD_HIP(hipMemcpy(t.pos_used, ip.data(), ip.size() * 4, hipMemcpyHostToDevice));
  }
  d->table_builds += 1;
  return MD_OK;
}

// Makes H x W the model's current input size: geometry, workspace plan (the arena grows when the new size needs more --
// never per call at a size seen before), and the size's tables from the cache (built on first use). A call at the current
// size returns at once.
static int da3_set_shape(md_model_s* m, int H, int W, bool force) {
  md_model_s::Da3State* d = m->da3;
  const ViTDims& v = d->cfg.vit;
  if (H <= 0 || W <= 0 || H % v.ps != 0 || W % v.ps != 0)  // mod.rs:509-520
    MD_FAIL(MD_ERR_SHAPE, "Input %dx%d must be divisible by patch size %d", H, W, v.ps);
  const auto key = std::make_pair(H, W);
  if (!force && H == d->ih && W == d->iw && d->tok_index) {
    d->shapes[key].last_use = ++d->use_clock;
    return MD_OK;
  }
  if ((long)(H / v.ps) * (W / v.ps) + 1 > 60000) MD_FAIL(MD_ERR_UNSUPPORTED, "input %dx%d: more than 60000 tokens", H, W);
  MD_HIP(hipSetDevice(m->dev->ordinal));
  MD_HIP(hipDeviceSynchronize());  // nothing may still run on the plan that is about to be replaced
  const int oh = d->ih, ow = d->iw;
  da3_set_geometry(m, H, W);
  size_t need = 0;
  da3_plan(m, true, &need);
  if (need > m->ws.cap) {  // grow-only arena
    if (m->ws.base) (void)hipFree(m->ws.base);
    m->ws.base = nullptr;
    m->ws.cap = 0;
    if (hipMalloc((void**)&m->ws.base, need) != hipSuccess) {
      if (oh > 0) da3_set_geometry(m, oh, ow);
      d->tok_index = nullptr;  // forces a rebuild of the plan on the next call
      MD_FAIL(MD_ERR_OOM, "hipMalloc of %zu bytes for the %dx%d workspace failed (max_batch=%d)", need, H, W, d->cfg.max_batch);
    }
    m->ws.cap = need;
    m->alloc_count += 1;
    da3_drop_graphs(m);
  }
  // padding rows / channels / keys must be finite zeros for every kernel: the buffers move with the plan, so the arena is
  // cleared whenever the plan changes (a few hundred MB at HBM speed, once per change of size)
  MD_HIP(hipMemset(m->ws.base, 0, m->ws.cap));
  m->ws.off = 0;
  MD_TRY(da3_plan(m, false, nullptr));
  auto it = d->shapes.find(key);
  if (it == d->shapes.end()) {
    if ((int)d->shapes.size() >= md_model_s::Da3State::kMaxShapes) {  // evict the least recently used size
      auto lru = d->shapes.begin();
      for (auto j = d->shapes.begin(); j != d->shapes.end(); ++j)
        if (j->second.last_use < lru->second.last_use) lru = j;
      da3_free_tables(lru->second);
      d->shapes.erase(lru);
      da3_drop_graphs(m);
    }
    md_model_s::Da3State::ShapeTables t;
    const int st = da3_build_tables(m, t);
    if (st != MD_OK) {
      da3_free_tables(t);
      d->tok_index = nullptr;
      return st;
    }
    it = d->shapes.emplace(key, t).first;
  }
  md_model_s::Da3State::ShapeTables& t = it->second;
  t.last_use = ++d->use_clock;
  for (int s = 0; s < 4; ++s) d->pos_stage[s] = t.pos_stage[s];
  d->pos_final = t.pos_final;
  d->pos_aux = t.pos_aux;
  d->rope_cos = t.rope_cos;
  d->rope_sin = t.rope_sin;
  d->pos_used = t.pos_used;
  d->tok_index = &t.tok_index;
  d->vit.pos = t.pos_used ? t.pos_used : P32(m, "backbone.pretrained.pos_embed");
  MD_HIP(hipDeviceSynchronize());
  return MD_OK;
}

int da3_create(md_device_t dev, const Da3Cfg& cfg, md_model_t* out) {
  if (!dev || !out) MD_FAIL(MD_ERR_INVALID_ARG, "device/model pointer is null");
  const ViTDims& v = cfg.vit;
  if (v.D != v.heads * 64 || v.D % 64 != 0 || v.D > 1024) MD_FAIL(MD_ERR_UNSUPPORTED, "ViT width %d / heads %d unsupported", v.D, v.heads);
  const int img_h = cfg.image_size, img_w = cfg.image_width > 0 ? cfg.image_width : cfg.image_size;
  if (img_h % v.ps != 0 || img_w % v.ps != 0 || img_h <= 0 || img_w <= 0)  // mod.rs:509-520
    MD_FAIL(MD_ERR_SHAPE, "Input %dx%d must be divisible by patch size %d", img_h, img_w, v.ps);
  if (cfg.features % 64 != 0 || cfg.output_dim != (cfg.dual_head ? 2 : 1))
    MD_FAIL(MD_ERR_UNSUPPORTED, "head features %d / output_dim %d unsupported", cfg.features, cfg.output_dim);
  for (int s = 0; s < 4; ++s)
    if (cfg.out_channels[s] % 4 != 0) MD_FAIL(MD_ERR_UNSUPPORTED, "head out_channels must be multiples of 4");
  if (cfg.dual_head && (cfg.ext_block_start < 0 || cfg.ext_block_start >= v.depth))
    MD_FAIL(MD_ERR_INVALID_ARG, "dual head needs the extended backbone (ext_block_start %d)", cfg.ext_block_start);
  MD_HIP(hipSetDevice(dev->ordinal));
  md_model_s* m = new md_model_s();
  m->dev = dev;
  m->kind = 1;
  // MD_PREC_FP8 is bf16 everywhere except the four ViT linear layers (e4m3 operands)
  const bool fp8 = cfg.precision == MD_PREC_FP8;
  if (fp8 && v.D % 128 != 0) MD_FAIL(MD_ERR_UNSUPPORTED, "fp8 operands need a ViT width that is a multiple of 128 (got %d)", v.D);
  m->prec = fp8 ? MD_PREC_BF16 : cfg.precision;
  m->esz = m->prec == MD_PREC_F32 ? 4 : 2;
  m->ke = 128 / m->esz;
  m->xm = m->prec == MD_PREC_F16X2 ? 2 : 1;  // split-half operands: activation rows are [hi | lo]
  m->wterms = m->xm == 2 ? 3 : 1;            // decided by model_commit: 2 when every plain weight is an exact half (an f16 record)
  m->cfg.max_batch = cfg.max_batch;
  m->cfg.precision = m->prec;
  m->da3 = new md_model_s::Da3State();
  md_model_s::Da3State* d = m->da3;
  d->cfg = cfg;
  d->cfg.precision = m->prec;
  d->fp8 = fp8;
  d->hp = cfg.dual_head ? "head_dual" : "head_mono";
  d->din = cfg.dual_head ? 2 * v.D : v.D;
  d->native_grid = v.img / v.ps;  // the pos_embed parameter's grid (37 for ViT-L/14 @ 518)
  da3_set_geometry(m, img_h, img_w);
  m->S = cfg.image_size;
  auto fail = [&](int code) {
    model_destroy(m);
    return code;
  };
  m->params = da3_param_specs(cfg, MD_INIT_REFERENCE);
  size_t off = 0;
  std::vector<size_t> offs;
  for (size_t i = 0; i < m->params.size(); ++i) {
    m->pindex[m->params[i].name] = (int)i;
    offs.push_back(off);
    off += align_up(m->params[i].count() * 4, 256);
  }
  m->w32_bytes = off;
  if (hipMalloc((void**)&m->w32_base, m->w32_bytes) != hipSuccess) {
    set_error("hipMalloc of %zu bytes for the fp32 weights failed", m->w32_bytes);
    return fail(MD_ERR_OOM);
  }
  (void)hipMemset(m->w32_base, 0, m->w32_bytes);
  for (size_t i = 0; i < m->params.size(); ++i) m->w32.push_back((float*)(m->w32_base + offs[i]));

  const int D = v.D, F = cfg.features;
  const int* oc = cfg.out_channels;
  const std::string bp = "backbone.pretrained";
  add_pack(m, bp + ".patch_embed.proj.weight", PACK_NK, D, 3 * v.ps * v.ps, 1);
  for (int i = 0; i < v.depth; ++i) {
    const std::string b = bp + ".blocks." + std::to_string(i);
    add_pack(m, b + ".attn.qkv.weight", PACK_NK, 3 * D, D, 1);
    add_pack(m, b + ".attn.proj.weight", PACK_NK, D, D, 1);
    add_pack(m, b + ".mlp.fc1.weight", PACK_NK, 4 * D, D, 1);
    add_pack(m, b + ".mlp.fc2.weight", PACK_NK, D, 4 * D, 1);
  }
  const std::string hp = d->hp;
  for (int s = 0; s < 4; ++s) add_pack(m, hp + ".projects." + std::to_string(s) + ".weight", PACK_NK, oc[s], d->din, 1);
  add_pack(m, hp + ".resize_layers.0.conv_t.weight", PACK_DECONV, oc[0], oc[0], 4);
  add_pack(m, hp + ".resize_layers.1.conv_t.weight", PACK_DECONV, oc[1], oc[1], 2);
  add_pack(m, hp + ".resize_layers.3.conv.weight", PACK_CONV3, oc[3], oc[3], 3);
  for (int s = 0; s < 4; ++s) add_pack(m, hp + ".scratch.layer" + std::to_string(s + 1) + "_rn.weight", PACK_CONV3, F, oc[s], 3);
  for (const char* suffix : {"", "_aux"}) {
    if (suffix[0] && !cfg.dual_head) continue;
    for (int i = 1; i <= 4; ++i) {
      const std::string r = hp + ".scratch.refinenet" + std::to_string(i) + suffix;
      for (const char* u : {"residual1", "residual2"}) {
        add_pack(m, r + "." + u + ".conv1.weight", PACK_CONV3, F, F, 3);
        add_pack(m, r + "." + u + ".conv2.weight", PACK_CONV3, F, F, 3);
      }
      add_pack(m, r + ".out_conv.weight", PACK_NK, F, F, 1);
    }
  }
  add_pack(m, hp + ".scratch.output_conv1.weight", PACK_CONV3, F / 2, F, 3);
  add_pack(m, hp + ".scratch.output_conv2.conv1.weight", PACK_CONV3, 32, F / 2, 3);
  if (cfg.dual_head) {  // only the last aux level reaches the outputs (build_aux_logits, dpt.rs:405-440)
    const std::string lv = std::to_string(cfg.aux_levels - 1);
    int cin = F;
    for (int j = 0; j < cfg.aux_out1_conv_num; ++j) {
      const int cout = j % 2 == 0 ? F / 2 : F;
      add_pack(m, hp + ".scratch.output_conv1_aux." + lv + ".layers." + std::to_string(j) + ".weight", PACK_CONV3, cout, cin, 3);
      cin = cout;
    }
    add_pack(m, hp + ".scratch.output_conv2_aux." + lv + ".reduce.weight", PACK_CONV3, 32, F / 2, 3);
  }
  size_t poff = 0;
  for (auto& e : m->packs) {
    e.dst = (void*)poff;
    poff += align_up(e.bytes + 256, 256);
  }
  m->wpk_bytes = poff;
  if (hipMalloc((void**)&m->wpk_base, m->wpk_bytes) != hipSuccess) {
    set_error("hipMalloc of %zu bytes for the packed weights failed", m->wpk_bytes);
    return fail(MD_ERR_OOM);
  }
  (void)hipMemset(m->wpk_base, 0, m->wpk_bytes);
  for (auto& e : m->packs) e.dst = m->wpk_base + (size_t)e.dst;

  if (fp8) {
    const size_t per_block = (size_t)12 * D * D + (size_t)(3 * D + D + 4 * D + D) * 4 + 8 * 256;
    if (hipMalloc((void**)&d->w8_base, per_block * v.depth) != hipSuccess) {
      set_error("hipMalloc of %zu bytes for the fp8 weights failed", per_block * v.depth);
      return fail(MD_ERR_OOM);
    }
    (void)hipMemset(d->w8_base, 0, per_block * v.depth);
    char* q = d->w8_base;
    const int nn[4] = {3 * D, D, 4 * D, D}, kk[4] = {D, D, D, 4 * D};
    for (int i = 0; i < v.depth; ++i) {
      md_model_s::Da3State::Fp8Block b8;
      for (int j = 0; j < 4; ++j) {
        b8.w[j] = q;
        q += align_up((size_t)nn[j] * kk[j], 256);
        b8.s[j] = (float*)q;
        q += align_up((size_t)nn[j] * 4, 256);
      }
      d->w8.push_back(b8);
    }
  }
  VitW& w = d->vit;
  w.pe_w = PK(m, bp + ".patch_embed.proj.weight");
  w.pe_b = P32(m, bp + ".patch_embed.proj.bias");
  w.cls = P32(m, bp + ".cls_token");
  w.pos = P32(m, bp + ".pos_embed");
  w.norm_g = P32(m, bp + ".norm.gamma");
  w.norm_b = P32(m, bp + ".norm.beta");
  for (int i = 0; i < v.depth; ++i) {
    const std::string b = bp + ".blocks." + std::to_string(i);
    VitBlockW k;
    k.n1g = P32(m, b + ".norm1.gamma"); k.n1b = P32(m, b + ".norm1.beta");
    k.n2g = P32(m, b + ".norm2.gamma"); k.n2b = P32(m, b + ".norm2.beta");
    k.qkv_w = PK(m, b + ".attn.qkv.weight"); k.qkv_b = P32(m, b + ".attn.qkv.bias");
    k.proj_w = PK(m, b + ".attn.proj.weight"); k.proj_b = P32(m, b + ".attn.proj.bias");
    k.ls1 = P32(m, b + ".ls1.gamma"); k.ls2 = P32(m, b + ".ls2.gamma");
    k.fc1_w = PK(m, b + ".mlp.fc1.weight"); k.fc1_b = P32(m, b + ".mlp.fc1.bias");
    k.fc2_w = PK(m, b + ".mlp.fc2.weight"); k.fc2_b = P32(m, b + ".mlp.fc2.bias");
    w.blk.push_back(k);
  }

  if (hipMalloc(&m->zero_page, 4096) != hipSuccess) return fail(MD_ERR_OOM);
  (void)hipMemset(m->zero_page, 0, 4096);
  // workspace for the configured size + its tables (PosEmbedCache, dpt.rs:784-833: built once per shape); other sizes get
  // theirs on their first infer call (da3_set_shape)
  {
    d->ih = 0;
    d->iw = 0;
    const int st = da3_set_shape(m, img_h, img_w, true);
    if (st != MD_OK) return fail(st);
  }
  m->alloc_count = 0;  // count what the infer calls allocate, not the construction
  (void)hipDeviceSynchronize();
  *out = m;
  return MD_OK;
}

long da3_shape_builds(md_model_t m) { return (m && m->da3) ? m->da3->table_builds : 0; }

void da3_destroy_state(md_model_t m) {
  if (m && m->da3 && m->da3->w8_base) (void)hipFree(m->da3->w8_base);
  if (!m || !m->da3) return;
  for (auto& kv : m->da3->shapes) da3_free_tables(kv.second);
  m->da3->shapes.clear();
  if (m->da3->depth_stage) (void)hipFree(m->da3->depth_stage);
  if (m->da3->cam_enc_ws) (void)hipFree(m->da3->cam_enc_ws);
  if (m->da3->tok_stage) (void)hipFree(m->da3->tok_stage);
  delete m->da3;
  m->da3 = nullptr;
}

int da3_init_seeded(md_model_t m, uint64_t seed, int scheme) {
  if (scheme != MD_INIT_REFERENCE && scheme != MD_INIT_PARITY) MD_FAIL(MD_ERR_INVALID_ARG, "unknown init scheme %d", scheme);
  MD_HIP(hipSetDevice(m->dev->ordinal));
  std::vector<ParamSpec> specs = da3_param_specs(m->da3->cfg, scheme);
  if (specs.size() != m->params.size()) MD_FAIL(MD_ERR_FORMAT, "internal: inventory mismatch");
  std::vector<float> tmp;
  for (size_t i = 0; i < specs.size(); ++i) {
    const size_t n = specs[i].count();
    tmp.resize(n);
    uniform_stream(specs[i].name, seed, n, specs[i].lo, specs[i].hi, tmp.data());
    MD_HIP(hipMemcpy(m->w32[i], tmp.data(), n * 4, hipMemcpyHostToDevice));
  }
  return model_commit(m);
}

int da3_load_container(md_model_t m, const char* path) {
  MD_TRY(model_load_params_from_container(m, path));
  return model_commit(m);
}

static int da3_tok_index(md_model_s* m, int B, int** out) {
  md_model_s::Da3State* d = m->da3;
  auto it = d->tok_index->find(B);
  if (it != d->tok_index->end()) {
    *out = it->second;
    return MD_OK;
  }
  std::vector<int> h((size_t)B * d->P);
  for (int b = 0; b < B; ++b)
    for (int p = 0; p < d->P; ++p) h[(size_t)b * d->P + p] = b * d->SS + 1 + p;
  int* dev = nullptr;
  MD_HIP(hipMalloc((void**)&dev, h.size() * 4));
  MD_HIP(hipMemcpy(dev, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  (*d->tok_index)[B] = dev;
  m->alloc_count += 1;
  *out = dev;
  return MD_OK;
}

static int da3_infer_eager(md_model_t m, const float* nchw, int B, int H, int W, int in_kind, const Da3Outputs& outp, int out_kind,
                           hipStream_t stream);

int da3_infer_ex(md_model_t m, const float* nchw, int B, int H, int W, int in_kind, const Da3Outputs& outp, int out_kind,
                 hipStream_t stream) {
  if (!m || m->kind != 1 || !m->da3) MD_FAIL(MD_ERR_INVALID_ARG, "not a Depth-Anything-v3 model");
  auto body = [&]() { return da3_infer_eager(m, nchw, B, H, W, in_kind, outp, out_kind, stream); };
  if (!m->graph_enabled) return body();
  MD_HIP(hipSetDevice(m->dev->ordinal));
  hipStream_t st = stream ? stream : m->dev->stream;
  const bool eligible = nchw && !outp.tokens[0] && outp.depth && !outp.raw_logits && in_kind == MD_MEM_DEVICE && out_kind == MD_MEM_DEVICE && m->committed && B > 0 &&
                        B <= m->da3->cfg.max_batch && H == m->da3->ih && W == m->da3->iw;
  const std::vector<uintptr_t> key = {(uintptr_t)st, (uintptr_t)B, (uintptr_t)H, (uintptr_t)W, (uintptr_t)nchw, (uintptr_t)outp.depth,
                                      (uintptr_t)outp.depth_confidence, (uintptr_t)outp.aux, (uintptr_t)outp.aux_confidence,
                                      (uintptr_t)outp.pose_encoding, (uintptr_t)outp.extrinsics, (uintptr_t)outp.intrinsics,
                                      (uintptr_t)outp.cam_extrinsics, (uintptr_t)outp.cam_intrinsics, (uintptr_t)outp.cam_views};
  return run_with_graph(m, st, key, eligible, body);
}

static int da3_infer_eager(md_model_t m, const float* nchw, int B, int H, int W, int in_kind, const Da3Outputs& outp, int out_kind,
                           hipStream_t stream) {
  if (!m || m->kind != 1 || !m->da3) MD_FAIL(MD_ERR_INVALID_ARG, "not a Depth-Anything-v3 model");
  if (!m->committed) MD_FAIL(MD_ERR_INVALID_ARG, "weights were modified; call md_model_commit_weights first");
  const KsplitScope ksplit(m->batch_invariant ? 0 : 1);  // small long-K launches may split their contraction inside the workgroup (kernels/gemm.h)
  const AttnSmallScope attn_small(m->batch_invariant ? 0 : 1);  // ... and few-workgroup attention launches their keys (kernels/ops.h)
  const bool from_tokens = outp.tokens[0] != nullptr;
  if ((!nchw && !from_tokens) || (!outp.depth && !outp.raw_logits)) MD_FAIL(MD_ERR_INVALID_ARG, "null pointer");
  if (outp.raw_logits && (outp.depth || outp.depth_confidence || outp.aux || outp.aux_confidence || outp.pose_encoding || outp.extrinsics || outp.intrinsics))
    MD_FAIL(MD_ERR_INVALID_ARG, "infer_raw returns the main logits only");
  md_model_s::Da3State* d = m->da3;
  const Da3Cfg& c = d->cfg;
  const ViTDims& v = c.vit;
  if (!c.dual_head && (outp.depth_confidence || outp.aux || outp.aux_confidence || outp.pose_encoding || outp.extrinsics || outp.intrinsics))
    MD_FAIL(MD_ERR_UNSUPPORTED, "the mono head produces depth only (finalize_inference, mod.rs:587-624)");
  if (B <= 0 || H <= 0 || W <= 0) MD_FAIL(MD_ERR_SHAPE, "invalid input shape [%d,3,%d,%d]", B, H, W);
  if (H % v.ps != 0 || W % v.ps != 0)  // depth_anything3/mod.rs:509-520 (assert -> checked precondition)
    MD_FAIL(MD_ERR_SHAPE, "Input %dx%d must be divisible by patch size %d", H, W, v.ps);
  if (B > c.max_batch) MD_FAIL(MD_ERR_SHAPE, "batch %d exceeds max_batch %d", B, c.max_batch);
  MD_HIP(hipSetDevice(m->dev->ordinal));
  MD_TRY(da3_set_shape(m, H, W, false));  // any multiple of the patch size (mod.rs:509-520); a no-op at the current size
  hipStream_t st = stream ? stream : m->dev->stream;
  Run r{m, st, B};
  const int D = v.D, heads = v.heads, SS = d->SS, NT = d->NT, P = d->P, ph = d->ph, pw = d->pw, F = c.features;
  const int IH = d->ih, IW = d->iw;
  const int din = d->din;
  const int* oc = c.out_channels;
  const int Fp = cpad(m, F), F2 = F / 2, F2p = cpad(m, F2);
  const std::string hp = d->hp, bp = "backbone.pretrained";
  const float* x_dev = nchw;
  if (in_kind == MD_MEM_HOST && !from_tokens) {
    MD_HIP(hipMemcpyAsync(d->xin, nchw, (size_t)B * 3 * H * W * 4, hipMemcpyHostToDevice, st));
    x_dev = d->xin;
  }
  auto Wk = [&](const std::string& n) { return PK(m, n); };
  auto Bi = [&](const std::string& n) { return P32(m, n); };
  if (from_tokens) {
    // ---- `infer_from_tokens` (mod.rs:405-469): no backbone, no camera prediction; the head's token LayerNorm (mono: non-affine,
    //      dpt.rs:761-766; dual: the affine `norm`, dpt.rs:308) on the caller's hook tokens, patch rows only ----
    if (outp.pose_encoding || outp.extrinsics || outp.intrinsics)
      MD_FAIL(MD_ERR_UNSUPPORTED, "infer_from_tokens has no camera prediction (finalize_inference(head_output, None), mod.rs:468)");
    const int T = outp.tokens_per_image;
    if (T != P && T != P + 1)  // mod.rs:419-424: tokens == expected -> patch_start 0, else patch_token_start = 1
      MD_FAIL(MD_ERR_SHAPE, "%d tokens per image for a %dx%d input: expected %d patch rows (or %d with a leading cls row)", T, H, W, P, P + 1);
    for (int hk = 0; hk < 4; ++hk)
      if (!outp.tokens[hk]) MD_FAIL(MD_ERR_LEVELS, "Backbone returned fewer hooks (%d) than requested (4)", hk);
    const int start = T == P ? 0 : 1;
    const size_t need = ((size_t)c.max_batch * SS + 64) * din;
    if (need > d->tok_stage_cap) {
      MD_HIP(hipStreamSynchronize(st));
      if (d->tok_stage) MD_HIP(hipFree(d->tok_stage));
      d->tok_stage = nullptr; d->tok_stage_cap = 0;
      MD_HIP(hipMalloc((void**)&d->tok_stage, need * 4));
      MD_HIP(hipMemset(d->tok_stage, 0, need * 4));
      d->tok_stage_cap = need;
      m->alloc_count += 1;
    }
    SeqGroups tg;
    memset(&tg, 0, sizeof(tg));
    tg.ngroups = 1;
    tg.nseq[0] = B;
    tg.a[0] = c.dual_head ? Bi(hp + ".norm.gamma") : nullptr;
    tg.b[0] = c.dual_head ? Bi(hp + ".norm.beta") : nullptr;
    const hipMemcpyKind kind = in_kind == MD_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    for (int hk = 0; hk < 4; ++hk) {
      for (int b = 0; b < B; ++b)  // patch rows of image b -> rows b*SS + 1 .. of the staging tensor (the layout the head gathers from)
        MD_HIP(hipMemcpyAsync(d->tok_stage + ((size_t)b * SS + 1) * din, outp.tokens[hk] + ((size_t)b * T + start) * din, (size_t)P * din * 4,
                              kind, st));
      r.begin("layernorm");
      MD_TRY(launch_layernorm(d->tok_stage, d->hookn[hk], (long)B * SS, din, 1e-5f, SS, tg, m->prec, 0, st));
      r.end();
      if (m->taps_enabled) MD_TRY(r.tap_token_rows(("backbone_tokens_" + std::to_string(hk)).c_str(), d->tok_stage, SS, 1, P, din, din, 0));
    }
  } else {
  // ---- camera encoder (`infer_with_camera`, mod.rs:522-527: a model without one ignores the camera inputs) ----
  float* cam_tok = nullptr;
  if (c.camera_encoder && c.dual_head && outp.cam_extrinsics && outp.cam_intrinsics) {
    const int V = outp.cam_views;
    if (V < 1 || V > MD_CAM_MAX_VIEWS) MD_FAIL(MD_ERR_SHAPE, "camera inputs with %d views (1..%d supported)", V, MD_CAM_MAX_VIEWS);
    const size_t n_in = (size_t)B * V * 21, n_tok = align_up((size_t)B * D, 64);
    const size_t need = align_up(n_in, 64) + n_tok + camera_encoder_scratch_floats(B, V, D);
    if (need > d->cam_enc_cap) {
      MD_HIP(hipStreamSynchronize(st));
      da3_drop_graphs(m);  // graphs captured with camera inputs hold the old buffer's addresses (cam_tok, the staged inputs)
      if (d->cam_enc_ws) MD_HIP(hipFree(d->cam_enc_ws));
      d->cam_enc_ws = nullptr; d->cam_enc_cap = 0;
      MD_HIP(hipMalloc((void**)&d->cam_enc_ws, need * 4));
      d->cam_enc_cap = need;
      m->alloc_count += 1;
    }
    const float *e_dev = outp.cam_extrinsics, *k_dev = outp.cam_intrinsics;
    if (in_kind == MD_MEM_HOST) {
      MD_HIP(hipMemcpyAsync(d->cam_enc_ws, outp.cam_extrinsics, (size_t)B * V * 12 * 4, hipMemcpyHostToDevice, st));
      MD_HIP(hipMemcpyAsync(d->cam_enc_ws + (size_t)B * V * 12, outp.cam_intrinsics, (size_t)B * V * 9 * 4, hipMemcpyHostToDevice, st));
      e_dev = d->cam_enc_ws; k_dev = d->cam_enc_ws + (size_t)B * V * 12;
    }
    cam_tok = d->cam_enc_ws + align_up(n_in, 64);
    CamEncW w;
    memset(&w, 0, sizeof(w));
    const std::string ce = "camera_encoder.";
    w.fc1_w = Bi(ce + "pose_branch.fc1.weight"); w.fc1_b = Bi(ce + "pose_branch.fc1.bias");
    w.fc2_w = Bi(ce + "pose_branch.fc2.weight"); w.fc2_b = Bi(ce + "pose_branch.fc2.bias");
    w.tn_g = Bi(ce + "token_norm.gamma"); w.tn_b = Bi(ce + "token_norm.beta");
    w.on_g = Bi(ce + "trunk_norm.gamma"); w.on_b = Bi(ce + "trunk_norm.beta");
    w.depth = c.cam_trunk_depth;
    if (w.depth > CamEncW::kMaxDepth) MD_FAIL(MD_ERR_UNSUPPORTED, "camera encoder trunk of %d blocks", w.depth);
    for (int i = 0; i < w.depth; ++i) {
      const std::string bk = ce + "trunk." + std::to_string(i) + ".";
      CamEncW::Blk& k = w.blk[i];
      k.n1g = Bi(bk + "norm1.gamma"); k.n1b = Bi(bk + "norm1.beta"); k.n2g = Bi(bk + "norm2.gamma"); k.n2b = Bi(bk + "norm2.beta");
      k.qkv_w = Bi(bk + "attn.qkv.weight"); k.qkv_b = Bi(bk + "attn.qkv.bias");
      k.proj_w = Bi(bk + "attn.proj.weight"); k.proj_b = Bi(bk + "attn.proj.bias"); k.ls1 = Bi(bk + "ls1.gamma");
      k.fc1_w = Bi(bk + "mlp.fc1.weight"); k.fc1_b = Bi(bk + "mlp.fc1.bias");
      k.fc2_w = Bi(bk + "mlp.fc2.weight"); k.fc2_b = Bi(bk + "mlp.fc2.bias"); k.ls2 = Bi(bk + "ls2.gamma");
      if (!k.qkv_w || !k.ls2) MD_FAIL(MD_ERR_FORMAT, "camera encoder block %d is not in the inventory", i);
    }
    if (!w.fc1_w || !w.on_b) MD_FAIL(MD_ERR_FORMAT, "camera encoder is not in the inventory");
    r.begin("camera_encoder");
    MD_TRY(launch_camera_encoder(e_dev, k_dev, B, V, D, c.cam_heads, H, W, c.cam_ln_eps, c.ln_eps, w, cam_tok + n_tok, cam_tok, st));
    r.end();
    MD_TRY(r.tap_f32("camera_token", cam_tok, B, D, 0, 0));  // CameraEncoder::forward's result (camera.rs:89-110)
  }
  // ---- backbone ----
  r.begin("patchify");  // + the cls rows (cls + pos[0]) and the zero padding rows of the residual stream, in the same launch
  MD_TRY(launch_patchify(x_dev, B, H, W, v.ps, d->Kp, d->patches, m->prec, st, d->xres, SS, NT, D, d->vit.cls, d->vit.pos));
  r.end();
  SeqGroups sg;
  memset(&sg, 0, sizeof(sg));
  sg.ngroups = 1;
  sg.nseq[0] = B;
  {
    GemmParams p;
    p.N = D; p.ngroups = 1; p.g_rows[0] = B * P; p.W[0] = d->vit.pe_w; p.bias[0] = d->vit.pe_b; p.pos[0] = d->vit.pos;
    p.A = d->patches;
    split_dense_a(m, p, d->Kp, d->Kp, 0);
    p.epi = EPI_PATCH_EMBED; p.out = d->xres; p.ldo = D; p.seq_stride = SS; p.seq_patches = P; p.embed = D;
    r.begin("patch_embed");
    MD_TRY(launch_gemm(p, A_DENSE, m->prec, TILE_AUTO, st));
    r.end();
  }
  const long rows = (long)B * SS;
  auto dense = [&](GemmParams& p) { p.ngroups = 1; p.g_rows[0] = (int)rows; };
  int hook_slot = 0;
  // The residual stream lives in `xcur`. A GLOBAL block of the extended backbone needs the state behind the last LOCAL block
  // for its hook (cat(x_local, LayerNorm(x))): its output projection therefore writes x + ls * (...) into the other buffer
  // (GemmParams::resid_src) and the two swap -- the copy of the whole stream that used to precede every global block is gone.
  float *xcur = d->xres, *xalt = d->xlocal;
  const int ext0 = c.dual_head ? c.ext_block_start : v.depth + 1;
  for (int i = 0; i < v.depth; ++i) {
    const VitBlockW& k = d->vit.blk[i];
    const bool ext = i >= ext0, is_global = ext && (i % 2 == 1);
    // entering block ext0 the camera token takes the cls slot -- the encoder's (mod.rs:522-531) or the learned reference-view one:
    // this block's first LayerNorm replaces row 0 of every sequence on its way in (and writes it back to the residual stream)
    const float* tok0 = i == ext0 ? (cam_tok ? cam_tok : Bi(bp + ".camera_token")) : nullptr;
    const int tok0_stride = (i == ext0 && cam_tok) ? D : 0;
    // MD_PREC_FP8: the operands of the four linear layers are e4m3 (LayerNorm / attention / GELU outputs are
    // written as e4m3 on static scales; weights were quantised per output channel at commit)
    const bool f8 = d->fp8;
    const int lin_prec = f8 ? MD_PREC_FP8 : m->prec;
    const float a_inv = 1.0f / md_model_s::Da3State::kActScale, h_inv = 1.0f / md_model_s::Da3State::kHidScale;
    sg.a[0] = k.n1g; sg.b[0] = k.n1b;
    r.begin("layernorm");
    MD_TRY(launch_layernorm(xcur, d->xn, rows, D, c.ln_eps, SS, sg, lin_prec, 0, st, a_inv, tok0, tok0_stride, tok0 ? xcur : nullptr));
    r.end();
    {
      GemmParams p;
      p.N = 3 * D; dense(p); p.W[0] = k.qkv_w; p.bias[0] = k.qkv_b; p.A = d->xn;
      if (f8) { p.K = D; p.lda = D; } else split_dense_a(m, p, D, D, 0);
      p.v_plane = (long)m->vt_plane;
      p.epi = EPI_QKV; p.out = d->qk; p.vT = d->vT; p.seq_stride = SS; p.embed = D; p.heads = heads; p.kpad = d->kpad; p.qscale = attn_qscale(m->prec);
      if (f8) { p.W[0] = d->w8[i].w[0]; p.wscale[0] = d->w8[i].s[0]; p.ascale = md_model_s::Da3State::kActScale; }
      if (ext && !f8) {  // per-head q/k LayerNorm + 2-D RoPE in this GEMM's epilogue (global blocks: every patch at position (1, 1))
        const std::string a = bp + ".blocks." + std::to_string(i) + ".attn.";
        p.qkn_g[0] = Bi(a + "q_norm.gamma"); p.qkn_b[0] = Bi(a + "q_norm.beta");
        p.qkn_g[1] = Bi(a + "k_norm.gamma"); p.qkn_b[1] = Bi(a + "k_norm.beta");
        p.qkn_eps = c.qk_norm_eps; p.rope_cos = d->rope_cos; p.rope_sin = d->rope_sin;
        p.rope_pw = pw; p.rope_global = is_global ? 1 : 0; p.rope_ntok = NT;
      }
      r.begin("qkv_gemm");
      MD_TRY(launch_gemm(p, A_DENSE, lin_prec, TILE_AUTO, st));
      r.end();
    }
    if (ext && f8) {  // e4m3 operands: the separate kernel (the fused epilogue is not built for the block-scaled GEMM)
      const std::string a = bp + ".blocks." + std::to_string(i) + ".attn.";
      r.begin("qk_norm_rope");
      MD_TRY(launch_qk_norm_rope(d->qk, rows, SS, NT, D, heads, pw, Bi(a + "q_norm.gamma"), Bi(a + "q_norm.beta"),
                                 Bi(a + "k_norm.gamma"), Bi(a + "k_norm.beta"), c.qk_norm_eps, d->rope_cos, d->rope_sin,
                                 is_global ? 1 : 0, attn_qscale(m->prec), m->prec, st));
      r.end();
    }
    if (m->prec != MD_PREC_F32) {
      r.begin("attention");
      MD_TRY(launch_attention(d->qk, d->vT, d->ao, B, SS, NT, heads, D, d->kpad, m->prec, st, f8 ? a_inv : 0.f, (long)m->vt_plane));
      r.end();
    } else {
      GemmParams p;
      p.N = SS; p.K = 64; p.ngroups = 1; p.g_rows[0] = NT; p.batch = B * heads; p.batch_inner = heads;
      p.A = d->qk; p.lda = 2 * D; p.a_bs[0] = (long)SS * 2 * D; p.a_bs[1] = 64;
      p.W[0] = (const float*)d->qk + D; p.ldw = 2 * D; p.w_bs[0] = (long)SS * 2 * D; p.w_bs[1] = 64;
      p.epi = EPI_STORE; p.out_f32 = 1; p.out = d->scores; p.ldo = d->kpad;
      p.o_bs[0] = (long)heads * SS * d->kpad; p.o_bs[1] = (long)SS * d->kpad;
      r.begin("attn_scores_f32");
      MD_TRY(launch_gemm(p, A_DENSE, m->prec, TILE_128x128, st));
      r.end();
      r.begin("attn_softmax_f32");
      MD_TRY(launch_softmax_rows(d->scores, (long)B * heads * SS, NT, d->kpad, 0.125f, st));
      r.end();
      GemmParams q;
      q.N = 64; q.K = d->kpad; q.ngroups = 1; q.g_rows[0] = NT; q.batch = B * heads; q.batch_inner = heads;
      q.A = d->scores; q.lda = d->kpad; q.a_bs[0] = (long)heads * SS * d->kpad; q.a_bs[1] = (long)SS * d->kpad;
      q.W[0] = d->vT; q.ldw = d->kpad; q.w_bs[0] = (long)heads * 64 * d->kpad; q.w_bs[1] = 64L * d->kpad;
      q.epi = EPI_STORE; q.out = d->ao; q.ldo = D; q.o_bs[0] = (long)SS * D; q.o_bs[1] = 64;
      r.begin("attn_pv_f32");
      MD_TRY(launch_gemm(q, A_DENSE, m->prec, TILE_128x128, st));
      r.end();
    }
    {
      GemmParams p;
      p.N = D; dense(p); p.W[0] = k.proj_w; p.bias[0] = k.proj_b; p.scale[0] = k.ls1;
      p.A = d->ao;
      if (f8) { p.K = D; p.lda = D; } else split_dense_a(m, p, D, D, 0);
      p.epi = EPI_RESID_LS; p.out = xcur; p.ldo = D;
      if (is_global) {  // x_i = x_{i-1} + ...: read the last local state, write the other buffer, keep x_{i-1} for the hook
        p.resid_src = xcur; p.out = xalt;
        std::swap(xcur, xalt);
      }
      if (f8) { p.W[0] = d->w8[i].w[1]; p.wscale[0] = d->w8[i].s[1]; p.ascale = md_model_s::Da3State::kActScale; }
      r.begin("proj_gemm");
      MD_TRY(launch_gemm(p, A_DENSE, lin_prec, TILE_AUTO, st));
      r.end();
    }
    sg.a[0] = k.n2g; sg.b[0] = k.n2b;
    r.begin("layernorm");
    MD_TRY(launch_layernorm(xcur, d->xn, rows, D, c.ln_eps, SS, sg, lin_prec, 0, st, a_inv));
    r.end();
    {
      GemmParams p;
      p.N = 4 * D; dense(p); p.W[0] = k.fc1_w; p.bias[0] = k.fc1_b; p.A = d->xn;
      if (f8) { p.K = D; p.lda = D; } else split_dense_a(m, p, D, D, 0);
      p.epi = EPI_STORE; p.act = ACT_GELU; p.out = d->hbuf;
      split_out(m, p, 4 * D, true);
      if (f8) {
        p.W[0] = d->w8[i].w[2]; p.wscale[0] = d->w8[i].s[2]; p.ascale = md_model_s::Da3State::kActScale;
        p.out_fp8 = 1; p.out_inv_scale = h_inv;
      }
      r.begin("fc1_gemm");
      MD_TRY(launch_gemm(p, A_DENSE, lin_prec, f8 ? TILE_256x256 : TILE_AUTO, st));
      r.end();
    }
    {
      GemmParams p;
      p.N = D; dense(p); p.W[0] = k.fc2_w; p.bias[0] = k.fc2_b; p.scale[0] = k.ls2;
      p.A = d->hbuf;
      if (f8) { p.K = 4 * D; p.lda = 4 * D; } else split_dense_a(m, p, 4 * D, 4 * D, 0);
      p.epi = EPI_RESID_LS; p.out = xcur; p.ldo = D;
      if (f8) { p.W[0] = d->w8[i].w[3]; p.wscale[0] = d->w8[i].s[3]; p.ascale = md_model_s::Da3State::kHidScale; }
      r.begin("fc2_gemm");
      MD_TRY(launch_gemm(p, A_DENSE, lin_prec, TILE_AUTO, st));
      r.end();
    }
    if (c.dual_head) {
      // hooks = LayerNorm_head(cat(x after the last local block, LayerNorm_final(x))); the camera feature is
      // token 0 of the raw concat at the last hook
      bool hooked = false;
      for (int hk = 0; hk < 4; ++hk) hooked |= c.hook_ids[hk] == i;
      const float* xl = is_global ? xalt : xcur;  // a local block is its own "last local" state; behind a global block the other buffer holds it
      for (int hk = 0; hk < 4; ++hk)
        if (c.hook_ids[hk] == i) {
          r.begin("hook_cat_ln");
          MD_TRY(launch_hook_cat_ln(xl, xcur, rows, SS, NT, D, d->vit.norm_g, d->vit.norm_b, c.ln_eps, Bi(hp + ".norm.gamma"),
                                    Bi(hp + ".norm.beta"), 1e-5f, d->hookn[hk], hk == 3 ? d->cam_raw : nullptr, m->prec, st));
          r.end();
          if (m->taps_enabled) {  // DepthTrace::backbone_tokens (mod.rs:241-246,344-347): cat(x_local, LayerNorm_final(x)) patch rows
            const std::string tn = "backbone_tokens_" + std::to_string(hk);
            sg.a[0] = d->vit.norm_g; sg.b[0] = d->vit.norm_b;
            MD_TRY(launch_layernorm(xcur, d->lnf, rows, D, c.ln_eps, SS, sg, m->prec, 1, st));
            MD_TRY(r.tap_token_rows(tn.c_str(), xl, SS, 1, P, D, 2 * D, 0));
            MD_TRY(r.tap_token_rows(tn.c_str(), d->lnf, SS, 1, P, D, 2 * D, D));
          }
          ++hook_slot;
        }
      (void)hooked;
    } else {
      // hooks (mod.rs:202-215): final LayerNorm of the block output, then the head's non-affine token
      // norm (apply_token_norm, dpt.rs:761-766: biased variance, eps 1e-5). A block may feed several hooks.
      for (int hk = 0; hk < 4; ++hk)
        if (c.hook_ids[hk] == i) {
          sg.a[0] = d->vit.norm_g; sg.b[0] = d->vit.norm_b;
          r.begin("layernorm");
          MD_TRY(launch_layernorm(xcur, d->lnf, rows, D, c.ln_eps, SS, sg, m->prec, 1, st));
          r.end();
          if (m->taps_enabled) {  // DepthTrace::backbone_tokens (mod.rs:241-246,344-347)
            const std::string tn = "backbone_tokens_" + std::to_string(hk);
            MD_TRY(r.tap_token_rows(tn.c_str(), d->lnf, SS, 1, P, D, D, 0));
          }
          sg.a[0] = nullptr; sg.b[0] = nullptr;
          r.begin("layernorm");
          MD_TRY(launch_layernorm(d->lnf, d->hookn[hk], rows, D, 1e-5f, SS, sg, m->prec, 0, st));
          r.end();
          ++hook_slot;
        }
    }
  }
  if (hook_slot < 4) MD_FAIL(MD_ERR_LEVELS, "Backbone returned fewer hooks (%d) than requested (4)", hook_slot);  // mod.rs:532-537
  }  // !from_tokens

  // ---- DPT head: prepare_stage (dpt.rs:282-317 / 649-689) ----
  int* tok_idx = nullptr;
  MD_TRY(da3_tok_index(m, B, &tok_idx));
  const int sh[4] = {4 * ph, 2 * ph, ph, d->h3h}, sw[4] = {4 * pw, 2 * pw, pw, d->h3w};  // stage sizes (rows, columns)
  for (int s = 0; s < 4; ++s) {
    const int ocp = cpad(m, oc[s]);
    const std::string ps = hp + ".projects." + std::to_string(s);
    {  // 1x1 projection over gathered patch tokens + 0.1 * UV position table
      GemmParams p;
      p.N = oc[s]; p.ngroups = 1; p.g_rows[0] = B * P; p.W[0] = Wk(ps + ".weight"); p.bias[0] = Bi(ps + ".bias");
      p.A = d->hookn[s]; p.a_index = tok_idx;
      split_dense_a(m, p, din, din, 0);
      p.epi = EPI_STORE; p.out = d->sp[s];
      split_out(m, p, ocp, true);
      p.res1 = d->pos_stage[s]; p.ldr = p.ldo; p.r_plane = p.o_plane; p.res_mod = P;
      r.begin("head_proj");
      MD_TRY(launch_gemm(p, A_INDEXED, m->prec, TILE_AUTO, st));
      r.end();
    }
    const void* feat = d->sp[s];
    if (s == 0 || s == 1) {  // ConvTranspose k4s4 / k2s2 (+bias)
      const int f = s == 0 ? 4 : 2;
      const std::string rl = hp + ".resize_layers." + std::to_string(s) + ".conv_t";
      GemmParams p;
      p.N = f * f * oc[s]; p.ngroups = 1; p.g_rows[0] = B * P; p.W[0] = Wk(rl + ".weight"); p.bias[0] = Bi(rl + ".bias");
      p.A = d->sp[s];
      split_dense_a(m, p, ocp, ocp, 0);
      p.epi = EPI_PIXSHUF; p.out = d->sr[s];
      split_out(m, p, ocp, true);
      p.psH = ph; p.psW = pw; p.psC = oc[s]; p.ps_f = f;
      r.begin("head_deconv");
      MD_TRY(launch_gemm(p, A_DENSE, m->prec, TILE_AUTO, st));
      r.end();
      feat = d->sr[s];
    } else if (s == 3) {  // Conv2d 3x3 stride 2 pad 1 (+bias)
      GemmParams p;
      p.N = oc[3]; p.ngroups = 1; p.g_rows[0] = B * d->h3h * d->h3w;
      p.W[0] = Wk(hp + ".resize_layers.3.conv.weight"); p.bias[0] = Bi(hp + ".resize_layers.3.conv.bias");
      p.A = d->sp[3]; p.cH = ph; p.cW = pw; p.cOH = d->h3h; p.cOW = d->h3w; p.cstride = 2; p.zero_page = m->zero_page;
      split_conv_a(m, p, ocp, 0);
      p.epi = EPI_STORE; p.out = d->sr[3];
      split_out(m, p, ocp, true);
      r.begin("head_conv_s2");
      MD_TRY(launch_gemm(p, A_CONV3, m->prec, TILE_AUTO, st));
      r.end();
      feat = d->sr[3];
    }
    // layerN_rn: 3x3, no bias -> features (+ relu copy for the residual units)
    MD_TRY(conv3(r, "head_conv3x3", feat, sh[s], sw[s], ocp, Wk(hp + ".scratch.layer" + std::to_string(s + 1) + "_rn.weight"), nullptr,
                 F, d->rn[s], Fp, ACT_NONE, nullptr, nullptr, d->rnr[s]));
    if (m->taps_enabled) {  // prepare_stage output and its layerN_rn map (dpt.rs:649-703)
      MD_TRY(r.tap_nhwc(("stage_" + std::to_string(s)).c_str(), feat, oc[s], sh[s], sw[s], ocp));
      MD_TRY(r.tap_nhwc(("layer" + std::to_string(s + 1) + "_rn").c_str(), d->rn[s], F, sh[s], sw[s], Fp));
    }
  }
  // ---- the head's tails. Dual head: the main fusion pyramid (depth, confidence) and the aux fusion pyramid (rays, confidence) have
  //      the same shapes and share their inputs (the layerN_rn maps): with the aux outputs wanted they run in the SAME launches as
  //      two weight groups -- group 0 = main on images [0, B), group 1 = aux on images [B, 2B) of every pyramid map. A 64-feature
  //      3x3 convolution costs ~11 us at 37^2 and ~14 us at 296^2 (launch floor + nine dependent k-tiles, not throughput), so the
  //      second group is nearly free where a second chain of launches was not (round 4: 44 -> 22 pyramid launches; the side-stream
  //      form overlapped only a third of the aux pyramid, profiles/r04_cfg2_branches.txt). ----
  struct Bufs { void *t, *x, *xr, *y, *up, *o; };
  Bufs bf{d->t, d->x, d->xr, d->y, d->up, d->o};
  const bool want_aux = c.dual_head && (outp.aux || outp.aux_confidence);
  const bool want_cam = c.dual_head && !from_tokens && (outp.pose_encoding || outp.extrinsics || outp.intrinsics);
  // Everything runs on the caller's stream. Rounds 3-4 ran the aux branch and the camera decoder on side streams (parallel branches of
  // the captured graph): every fork / join cost more than the overlap gave back (config 2, all outputs: 1.97 ms with two side streams,
  // 1.80 on one stream, 1.70 with the aux branch alone on a side stream; with the pyramids grouped 1.74 with a side stream for the aux
  // tail against 1.63 without -- profiles/r04_cfg2_branches.txt).
  Run& ra = r;  // aux neck + tail
  Run& rc = r;  // camera decoder
  const int G = want_aux ? 2 : 1;
  const char* const sfx[2] = {"", "_aux"};
  const size_t px_bytes = (size_t)Fp * m->esz * m->xm;  // one pixel of an F-channel map
  // 3x3 convolution F -> F over [G*B, hh, ww] with one weight set per group; `in_shared`: both groups read images [0, B) of `in`;
  // res1_shared: the same for the first residual input
  auto conv3g = [&](Run& rr, const void* in, bool in_shared, int hh, int ww, const std::string& rfb, const std::string& unit, void* out, int act,
                    const void* res1, bool res1_shared, const void* res2, void* out2) -> int {
    GemmParams p;
    const int M = B * hh * ww;
    p.N = F; p.ngroups = G;
    for (int g = 0; g < G; ++g) {
      p.g_rows[g] = M; p.g_row0[g] = g * M; p.g_arow0[g] = in_shared ? 0 : g * M;
      p.W[g] = Wk(rfb + sfx[g] + unit + ".weight"); p.bias[g] = Bi(rfb + sfx[g] + unit + ".bias");
    }
    p.A = in; p.cH = hh; p.cW = ww; p.zero_page = m->zero_page;
    split_conv_a(m, p, Fp, 0);
    p.epi = EPI_STORE; p.act = act; p.out = out; p.out2 = out2;
    split_out(m, p, Fp, true);
    p.res1 = res1; p.res2 = res2; p.ldr = p.ldo; p.r_plane = (res1 || res2) ? p.o_plane : 0;
    if (res1 && res1_shared && G > 1) p.res_mod = M;
    rr.begin("head_conv3x3");
    int s2 = launch_gemm(p, A_CONV3, m->prec, TILE_AUTO, rr.st);
    rr.end();
    return s2;
  };
  // ResidualConvUnit (dpt.rs:1248-1252): out = x + conv2(relu(conv1(relu(x)))) [+ extra]
  auto rcu = [&](Run& rr, const std::string& rfb, const std::string& unit, int hh, int ww, const void* x, const void* xr, bool x_shared,
                 const void* extra, void* out, void* out_relu) -> int {
    MD_TRY(conv3g(rr, xr, x_shared, hh, ww, rfb, unit + ".conv1", bf.t, ACT_RELU, nullptr, false, nullptr, nullptr));
    return conv3g(rr, bf.t, false, hh, ww, rfb, unit + ".conv2", out, ACT_NONE, x, x_shared, extra, out_relu);
  };
  // the four FeatureFusionBlocks (dpt.rs:1206-1222) from the coarsest stage up; result in bf.o at 8ph x 8pw (G*B images)
  const int target[4] = {8 * ph, 4 * ph, 2 * ph, ph}, targw[4] = {8 * pw, 4 * pw, 2 * pw, pw};  // output size of refinenet1..4
  auto pyramid = [&](Run& rr) -> int {
    const void* top = nullptr;
    for (int lvl = 3; lvl >= 0; --lvl) {
      const std::string rfb = hp + ".scratch.refinenet" + std::to_string(lvl + 1);
      const void *yx, *yxr;
      bool y_shared;
      if (lvl == 3) {
        yx = d->rn[3]; yxr = d->rnr[3]; y_shared = true;
      } else {
        MD_TRY(rcu(rr, rfb, ".residual1", sh[lvl], sw[lvl], d->rn[lvl], d->rnr[lvl], true, top, bf.x, bf.xr));
        yx = bf.x; yxr = bf.xr; y_shared = false;
      }
      MD_TRY(rcu(rr, rfb, ".residual2", sh[lvl], sw[lvl], yx, yxr, y_shared, nullptr, bf.y, nullptr));
      rr.begin("head_resize");
      MD_TRY(launch_resize_nhwc(bf.y, G * B, sh[lvl], sw[lvl], F, Fp, bf.up, target[lvl], targw[lvl], Fp, MD_INTERP_BURN, nullptr, m->prec, rr.st));
      rr.end();
      {  // out_conv 1x1 (+bias), one weight set per group
        GemmParams p;
        const int M2 = B * target[lvl] * targw[lvl];
        p.N = F; p.ngroups = G;
        for (int g = 0; g < G; ++g) {
          p.g_rows[g] = M2; p.g_row0[g] = g * M2; p.g_arow0[g] = g * M2;
          p.W[g] = Wk(rfb + sfx[g] + ".out_conv.weight"); p.bias[g] = Bi(rfb + sfx[g] + ".out_conv.bias");
        }
        p.A = bf.up;
        split_dense_a(m, p, Fp, Fp, 0);
        p.epi = EPI_STORE; p.out = bf.o;
        split_out(m, p, Fp, true);
        rr.begin("head_out_conv");
        MD_TRY(launch_gemm(p, A_DENSE, m->prec, TILE_AUTO, rr.st));
        rr.end();
      }
      top = bf.o;
      if (m->taps_enabled)  // FeatureFusionBlock outputs (dpt.rs:705-720), main and aux pyramids
        for (int g = 0; g < G; ++g)
          MD_TRY(rr.tap_nhwc(("refinenet" + std::to_string(lvl + 1) + sfx[g]).c_str(),
                             (const char*)bf.o + (size_t)g * B * target[lvl] * targw[lvl] * px_bytes, F, target[lvl], targw[lvl], Fp));
    }
    return MD_OK;
  };
  // fused tail: out_c[img][pixel] = act_c(w_c . relu(conv3x3(in) + b1) + b_c) for `nch` channels in ONE launch over a
  // [B, hh, ww, 64-padded] map (ConvStack / `reduce` + `project`, dpt.rs:481-513,1287-1290)
  struct TailCh { const float* w; float b; int act; float* out; long bstride; };
  auto tail = [&](Run& rr, const char* name, const void* in, int hh, int ww, const void* w1, const float* b1, const TailCh* chs, int nch) -> int {
    if (nch <= 0) return MD_OK;
    GemmParams p;
    p.N = 32; p.ngroups = 1; p.g_rows[0] = B * hh * ww; p.W[0] = w1;
    p.A = in; p.cH = hh; p.cW = ww; p.zero_page = m->zero_page;
    split_conv_a(m, p, F2p, 0);
    p.epi = EPI_HEAD; p.bias[0] = b1; p.head_nch = nch; p.head_plane = hh * ww;
    for (int i = 0; i < nch; ++i) {
      p.head_wc[i] = chs[i].w; p.head_bs[i] = chs[i].b; p.head_acts[i] = chs[i].act; p.head_out[i] = chs[i].out; p.head_bstride[i] = chs[i].bstride;
    }
    rr.begin(name);
    int s2 = launch_gemm(p, A_CONV3, m->prec, TILE_256x32, rr.st);
    rr.end();
    return s2;
  };
  auto host_out = [&](hipStream_t hs, float* dst, const float* src, size_t n) -> int {
    if (out_kind == MD_MEM_HOST && dst) MD_HIP(hipMemcpyAsync(dst, src, n * 4, hipMemcpyDeviceToHost, hs));
    return MD_OK;
  };
  const size_t out_elems = (size_t)B * IH * IW;
  const size_t stage_elems = out_elems * (outp.raw_logits ? c.output_dim : 1);
  float* depth_dev = outp.raw_logits ? outp.raw_logits : outp.depth;
  if (out_kind == MD_MEM_HOST) {
    if (d->depth_stage_elems < stage_elems) {
      if (d->depth_stage) (void)hipFree(d->depth_stage);
      d->depth_stage = nullptr; d->depth_stage_elems = 0;
      MD_HIP(hipMalloc((void**)&d->depth_stage, stage_elems * 4));
      m->alloc_count += 1;
      d->depth_stage_elems = stage_elems;
    }
    depth_dev = d->depth_stage;
  }
  // ---- aux branch (build_aux_logits, dpt.rs:356-441): aux fusion pyramid on the same layerN_rn maps -> last level's 5-conv
  //      neck -> + 2 x 0.1 x UV -> reduce 3x3 -> ReLU -> project 1x1 (7 ch: 6 ray values + confidence) ----
  MD_TRY(pyramid(r));  // both pyramids (two weight groups) when the aux outputs are wanted
  // output_conv1 (main, dpt.rs:337-344) and the first convolution of the aux neck (dpt.rs:1085-1113) are both 3x3 F -> F/2 on the
  // 8ph x 8pw results of their pyramids: one launch, two weight groups, into the (now free) `up` map -- main | aux
  const void* c1_map = d->c1;
  const void* aux_cur = nullptr;
  const std::string lv = std::to_string(c.aux_levels - 1);
  if (want_aux) {
    GemmParams p;
    const int M = B * 8 * ph * 8 * pw;
    const std::string n0 = hp + ".scratch.output_conv1_aux." + lv + ".layers.0";
    p.N = F2; p.ngroups = 2;
    for (int g = 0; g < 2; ++g) { p.g_rows[g] = M; p.g_row0[g] = g * M; p.g_arow0[g] = g * M; }
    p.W[0] = Wk(hp + ".scratch.output_conv1.weight"); p.bias[0] = Bi(hp + ".scratch.output_conv1.bias");
    p.W[1] = Wk(n0 + ".weight"); p.bias[1] = Bi(n0 + ".bias");
    p.A = bf.o; p.cH = 8 * ph; p.cW = 8 * pw; p.zero_page = m->zero_page;
    split_conv_a(m, p, Fp, 0);
    p.epi = EPI_STORE; p.out = bf.up;
    split_out(m, p, F2p, true);
    r.begin("head_conv3x3");
    MD_TRY(launch_gemm(p, A_CONV3, m->prec, TILE_AUTO, st));
    r.end();
    c1_map = bf.up;
    aux_cur = (const char*)bf.up + (size_t)M * F2p * m->esz * m->xm;
  } else {
    MD_TRY(conv3(r, "head_conv3x3", d->o, 8 * ph, 8 * pw, Fp, Wk(hp + ".scratch.output_conv1.weight"), Bi(hp + ".scratch.output_conv1.bias"), F2,
                 d->c1, F2p, ACT_NONE, nullptr, nullptr, nullptr));
  }
  if (want_aux) {
    const int ah = 8 * ph, aw = 8 * pw;
    const void* cur = aux_cur;  // the neck's first convolution ran beside output_conv1
    void* pp[2] = {d->up2, d->o2};
    int cin = F2;
    for (int j = 1; j < c.aux_out1_conv_num; ++j) {
      const int cout = j % 2 == 0 ? F / 2 : F;
      const std::string n = hp + ".scratch.output_conv1_aux." + lv + ".layers." + std::to_string(j);
      MD_TRY(conv3(ra, "aux_conv3x3", cur, ah, aw, cpad(m, cin), Wk(n + ".weight"), Bi(n + ".bias"), cout, pp[j & 1], cpad(m, cout), ACT_NONE,
                   nullptr, nullptr, nullptr));
      cur = pp[j & 1];
      cin = cout;
    }
    void* hin = cur == d->up2 ? d->o2 : d->up2;
    ra.begin("head_resize");
    MD_TRY(launch_resize_nhwc(cur, B, ah, aw, F2, F2p, hin, ah, aw, F2p, MD_INTERP_BURN, d->pos_aux, m->prec, ra.st));
    ra.end();
    if (m->taps_enabled) {  // DepthTrace::aux_stage_necks (last level) / aux_head_input (mod.rs:241-246)
      MD_TRY(ra.tap_nhwc("aux_neck", cur, F2, ah, aw, F2p));
      MD_TRY(ra.tap_nhwc("aux_head_input", hin, F2, ah, aw, F2p));
    }
    const std::string oh = hp + ".scratch.output_conv2_aux." + lv;
    const size_t plane = (size_t)ah * aw;
    const int K7 = c.aux_output_dim;
    const float* pw_ = Bi(oh + ".project.weight");
    TailCh chs[8];
    int nch = 0;
    for (int ch = 0; ch < K7; ++ch) {  // aux lands as [B, 6, h, w], the confidence as [B, h, w]; host outputs through [B, 7, h, w] staging
      const bool conf = ch == K7 - 1;
      float* user = conf ? outp.aux_confidence : outp.aux;
      if (!user) continue;
      TailCh t;
      t.w = pw_ + 32 * ch; t.b = d->aux_bias[ch]; t.act = conf ? 3 : 2;
      if (out_kind == MD_MEM_HOST) { t.out = d->aux_stage + (size_t)ch * plane; t.bstride = (long)K7 * plane; }
      else if (conf) { t.out = user; t.bstride = (long)plane; }
      else { t.out = user + (size_t)ch * plane; t.bstride = (long)(K7 - 1) * plane; }
      chs[nch++] = t;
    }
    MD_TRY(tail(ra, "aux_tail_fused", hin, ah, aw, Wk(oh + ".reduce.weight"), Bi(oh + ".reduce.bias"), chs, nch));
    if (out_kind == MD_MEM_HOST)
      for (int ch = 0; ch < K7; ++ch) {
        const bool conf = ch == K7 - 1;
        float* user = conf ? outp.aux_confidence : outp.aux;
        if (!user) continue;
        for (int b = 0; b < B; ++b)
          MD_TRY(host_out(ra.st, conf ? user + (size_t)b * plane : user + ((size_t)b * (K7 - 1) + ch) * plane,
                          d->aux_stage + ((size_t)b * K7 + ch) * plane, plane));
      }
  }
  // ---- camera decoder (camera.rs:143-199) on the raw camera feature of the last hook, fp32 ----
  if (want_cam) {
    // pose = (t3 | quat4 | relu(fov2)) rows [B, 9]; device outputs are written straight into the caller's buffers
    const bool dev_out = out_kind == MD_MEM_DEVICE;
    float* pose = (dev_out && outp.pose_encoding) ? outp.pose_encoding : d->pose;
    float* extr = (dev_out && outp.extrinsics) ? outp.extrinsics : d->extr;
    float* intr = (dev_out && outp.intrinsics) ? outp.intrinsics : d->intr;
    CamDecW w;
    auto CW = [&](const char* n) { return Bi(std::string("camera_decoder.") + n); };
    w.w1 = CW("backbone_1.weight"); w.b1 = CW("backbone_1.bias"); w.w2 = CW("backbone_2.weight"); w.b2 = CW("backbone_2.bias");
    w.wt = CW("fc_t.weight"); w.bt = CW("fc_t.bias"); w.wq = CW("fc_qvec.weight"); w.bq = CW("fc_qvec.bias");
    w.wf = CW("fc_fov.weight"); w.bf = CW("fc_fov.bias");
    rc.begin("camera_decoder");
    MD_TRY(launch_camera_decoder(d->cam_raw, B, din, w, H, W, d->cam_h1, d->cam_h2, pose, outp.extrinsics ? extr : nullptr,
                                 outp.intrinsics ? intr : nullptr, rc.st));
    rc.end();
    if (!dev_out) {
      if (outp.pose_encoding) MD_HIP(hipMemcpyAsync(outp.pose_encoding, pose, (size_t)B * 9 * 4, hipMemcpyDeviceToHost, rc.st));
      if (outp.extrinsics) MD_HIP(hipMemcpyAsync(outp.extrinsics, extr, (size_t)B * 12 * 4, hipMemcpyDeviceToHost, rc.st));
      if (outp.intrinsics) MD_HIP(hipMemcpyAsync(outp.intrinsics, intr, (size_t)B * 9 * 4, hipMemcpyDeviceToHost, rc.st));
    }
  }
  // ---- main branch: output_conv1 -> resize to the image size (+ UV table) -> output_conv2 + activation ----
  r.begin("head_resize");
  MD_TRY(launch_resize_nhwc(c1_map, B, 8 * ph, 8 * pw, F2, F2p, d->c1r, IH, IW, F2p, MD_INTERP_BURN, d->pos_final, m->prec, st));
  r.end();
  if (m->taps_enabled) {
    MD_TRY(r.tap_nhwc("output_conv1", c1_map, F2, 8 * ph, 8 * pw, F2p));
    MD_TRY(r.tap_nhwc("head_input", d->c1r, F2, IH, IW, F2p));  // resized + UV table: the input of output_conv2
  }
  {
    const void* w1 = Wk(hp + ".scratch.output_conv2.conv1.weight");
    const float* b1 = Bi(hp + ".scratch.output_conv2.conv1.bias");
    const float* w2 = Bi(hp + ".scratch.output_conv2.conv2.weight");
    TailCh chs[8];
    int nch = 0;
    float* cd = nullptr;
    if (outp.raw_logits) {
      // infer_raw (mod.rs:364-380): the dual head hands out `depth_logits` = output_conv2's result as it is (dpt.rs:337-354, 271);
      // the mono head's `forward_raw` has its activation applied (dpt.rs:700)
      if (c.output_dim > 8) MD_FAIL(MD_ERR_UNSUPPORTED, "infer_raw with %d channels", c.output_dim);
      for (int ch = 0; ch < c.output_dim; ++ch)
        chs[nch++] = TailCh{w2 + 32 * ch, d->main_bias[ch], c.dual_head ? 2 : 1, depth_dev + (size_t)ch * IH * IW, (long)c.output_dim * IH * IW};
      MD_TRY(tail(r, "head_tail_fused", d->c1r, IH, IW, w1, b1, chs, nch));
      MD_TRY(host_out(st, outp.raw_logits, depth_dev, stage_elems));
      if (out_kind == MD_MEM_HOST) MD_HIP(hipStreamSynchronize(st));
      return MD_OK;
    }
    chs[nch++] = TailCh{w2, d->main_bias[0], 1, depth_dev, (long)IH * IW};  // depth = exp(ch 0)
    if (c.dual_head && outp.depth_confidence) {  // confidence = exp(last channel) + 1 (select_conf_channel, ExpP1)
      cd = out_kind == MD_MEM_HOST ? d->conf_stage : outp.depth_confidence;
      chs[nch++] = TailCh{w2 + 32 * (c.output_dim - 1), d->main_bias[c.output_dim - 1], 3, cd, (long)IH * IW};
    }
    MD_TRY(tail(r, "head_tail_fused", d->c1r, IH, IW, w1, b1, chs, nch));
    MD_TRY(host_out(st, outp.depth, depth_dev, out_elems));
    if (cd) MD_TRY(host_out(st, outp.depth_confidence, cd, out_elems));
  }
  if (out_kind == MD_MEM_HOST) MD_HIP(hipStreamSynchronize(st));
  return MD_OK;
}

int da3_infer(md_model_t m, const float* nchw, int B, int H, int W, int in_kind, float* depth, int out_kind,
              hipStream_t stream) {
  Da3Outputs o;
  o.depth = depth;
  return da3_infer_ex(m, nchw, B, H, W, in_kind, o, out_kind, stream);
}

}  // namespace md
