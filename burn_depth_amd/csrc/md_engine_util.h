// Helpers shared by the model engines (Depth Pro: md_engine.hip, Depth-Anything-v3: md_da3.hip).
#pragma once

#include <algorithm>
#include <string>

#include "md_engine.h"

namespace md {

inline int round_up(int v, int a) { return (v + a - 1) / a * a; }

// elements of a packed weight; `terms` copies of the contraction per row (MD_PREC_F16X2: 2 or 3, PackEntry::terms)
inline size_t pack_elems(const PackEntry& e, int terms = 1) {
  const size_t kpt = (size_t)e.kp * terms;
  switch (e.kind) {
    case PACK_NK: return (size_t)e.d0 * kpt;
    case PACK_CONV3: return (size_t)e.d0 * 9 * kpt;
    case PACK_DECONV: return (size_t)e.k * e.k * e.d1 * kpt;
    case PACK_HEAD_W: return (size_t)4 * e.d0 * 9 * kpt;
    case PACK_HEAD_B: return (size_t)9 * e.d0;
    case PACK_C1C3_W: return (size_t)e.d0 * 9 * kpt;
    case PACK_C1C3_B: return (size_t)9 * e.d0;
    default: return (size_t)e.d0 * e.d1 * e.k * e.k;
  }
}
// allocation size of a pack: split-half models reserve three terms (the form is only known once the weights are committed)
inline size_t pack_bytes(const md_model_s* m, const PackEntry& e) {
  if (e.f32) return pack_elems(e) * 4;
  return pack_elems(e, m->xm == 2 ? 3 : 1) * (m->prec == MD_PREC_F32 ? 4 : 2);
}

inline const float* P32(md_model_s* m, const std::string& name) {
  auto it = m->pindex.find(name);
  return it == m->pindex.end() ? nullptr : m->w32[it->second];
}
inline const void* PK(md_model_s* m, const std::string& name) {
  auto it = m->pack_index.find(name);
  return it == m->pack_index.end() ? nullptr : m->packs[it->second].dst;
}

inline void add_pack(md_model_s* m, const std::string& name, int kind, int d0, int d1, int k, bool f32 = false) {
  auto it = m->pindex.find(name);
  if (it == m->pindex.end()) return;
  PackEntry e;
  e.param = it->second;
  e.kind = kind;
  e.d0 = d0;
  e.d1 = d1;
  e.k = k;
  e.f32 = f32 ? 1 : 0;
  const int contraction = kind == PACK_NK ? d1 : kind == PACK_CONV3 ? d1 : kind == PACK_DECONV ? d0 : d1;
  e.kp = f32 ? contraction : round_up(contraction, m->ke);
  e.bytes = pack_bytes(m, e);
  m->pack_index[name] = (int)m->packs.size();
  m->packs.push_back(e);
}

// a second packed form of a parameter under a name of its own (e.g. a convolution weight both as a direct-convolution and as an
// implicit-GEMM operand)
inline void add_pack_as(md_model_s* m, const std::string& pack_name, const std::string& param, int kind, int d0, int d1, int k) {
  auto it = m->pindex.find(param);
  if (it == m->pindex.end()) return;
  PackEntry e;
  e.param = it->second;
  e.kind = kind;
  e.d0 = d0;
  e.d1 = d1;
  e.k = k;
  e.f32 = 0;
  const int contraction = kind == PACK_DECONV ? d0 : d1;
  e.kp = round_up(contraction, m->ke);
  e.bytes = pack_bytes(m, e);
  m->pack_index[pack_name] = (int)m->packs.size();
  m->packs.push_back(e);
}

// deconv k2s2 (no bias) followed by a 1x1 conv: packed as ONE deconv whose weight is their product
// W'[ci][co][q] = sum_m Wd[ci][m][q] * Wo[co][m]  (decoder.rs:124-141 applies out_conv right after deconv)
inline void add_pack_composed(md_model_s* m, const std::string& name, const std::string& deconv, const std::string& conv1x1,
                              int cin, int cout) {
  auto a = m->pindex.find(deconv), b = m->pindex.find(conv1x1);
  if (a == m->pindex.end() || b == m->pindex.end()) return;
  PackEntry e;
  e.param = a->second;
  e.param2 = b->second;
  e.kind = PACK_DECONV;
  e.d0 = cin;
  e.d1 = cout;
  e.k = 2;
  e.kp = round_up(cin, m->ke);
  e.bytes = pack_bytes(m, e);
  m->pack_index[name] = (int)m->packs.size();
  m->packs.push_back(e);
}

// depth head: deconv k2s2 (+bias) -> conv 3x3 (+bias) composed into one 3x3 conv with 4 * cout columns on the deconv's
// input grid (`name`.weight, PACK_CONV3 layout) and nine position-class bias vectors (`name`.bias, f32 [9][cout])
inline void add_pack_head_fused(md_model_s* m, const std::string& name, const std::string& deconv, const std::string& conv,
                                int cin, int cmid, int cout) {
  auto wd = m->pindex.find(deconv + ".weight"), bd = m->pindex.find(deconv + ".bias");
  auto wc = m->pindex.find(conv + ".weight"), bc = m->pindex.find(conv + ".bias");
  if (wd == m->pindex.end() || bd == m->pindex.end() || wc == m->pindex.end() || bc == m->pindex.end()) return;
  PackEntry e;
  e.param = wd->second; e.param2 = wc->second; e.param3 = bd->second; e.param4 = bc->second;
  e.kind = PACK_HEAD_W;
  e.d0 = cout; e.d1 = cin; e.k = cmid;
  e.kp = round_up(cin, m->ke);
  e.bytes = pack_bytes(m, e);
  m->pack_index[name + ".weight"] = (int)m->packs.size();
  m->packs.push_back(e);
  e.kind = PACK_HEAD_B;
  e.f32 = 1;
  e.bytes = pack_bytes(m, e);
  m->pack_index[name + ".bias"] = (int)m->packs.size();
  m->packs.push_back(e);
}

// conv 1x1 (+bias) -> conv 3x3 pad 1 (+bias) composed into one 3x3 convolution (`name`.weight, PACK_CONV3 layout) and nine
// position-class bias vectors (`name`.bias, f32 [9][cout]); c1 = the 1x1 conv [cmid, cin], c3 = the 3x3 conv [cout, cmid]
inline void add_pack_c1c3(md_model_s* m, const std::string& name, const std::string& c1, const std::string& c3, int cin, int cmid,
                          int cout) {
  auto w1 = m->pindex.find(c1 + ".weight"), b1 = m->pindex.find(c1 + ".bias");
  auto w3 = m->pindex.find(c3 + ".weight"), b3 = m->pindex.find(c3 + ".bias");
  if (w1 == m->pindex.end() || b1 == m->pindex.end() || w3 == m->pindex.end() || b3 == m->pindex.end()) return;
  PackEntry e;
  e.param = w1->second; e.param2 = w3->second; e.param3 = b1->second; e.param4 = b3->second;
  e.kind = PACK_C1C3_W;
  e.d0 = cout; e.d1 = cin; e.k = cmid;
  e.kp = round_up(cin, m->ke);
  e.bytes = pack_bytes(m, e);
  m->pack_index[name + ".weight"] = (int)m->packs.size();
  m->packs.push_back(e);
  e.kind = PACK_C1C3_B;
  e.f32 = 1;
  e.bytes = pack_bytes(m, e);
  m->pack_index[name + ".bias"] = (int)m->packs.size();
  m->packs.push_back(e);
}

// two bias-free k2s2 deconvolutions in a row (encoder.rs:146-152, nothing between them) = ONE k4s4 deconvolution on the
// weight product W''[ci][co][2 dy1 + dy2][2 dx1 + dx2] = sum_m Wa[ci][m][dy1][dx1] * Wb[m][co][dy2][dx2]
inline void add_pack_deconv_pair(md_model_s* m, const std::string& name, const std::string& a, const std::string& b, int cin,
                                 int cmid, int cout) {
  auto wa = m->pindex.find(a), wb = m->pindex.find(b);
  if (wa == m->pindex.end() || wb == m->pindex.end()) return;
  PackEntry e;
  e.param = wa->second;
  e.param2 = wb->second;
  e.param3 = cmid;  // PACK_DECONV with k == 4 and param2: the middle channel count rides here
  e.kind = PACK_DECONV;
  e.d0 = cin;
  e.d1 = cout;
  e.k = 4;
  e.kp = round_up(cin, m->ke);
  e.bytes = pack_bytes(m, e);
  m->pack_index[name] = (int)m->packs.size();
  m->packs.push_back(e);
}

// Runs `body` (the launch schedule of one infer) eagerly the first time a (stream, shapes, pointers) key is
// seen -- that call allocates index tables and sets function attributes --, captures it into a hipGraph the second
// time and replays the instantiated graph from then on. Timing / tap modes and host-side buffers always run eagerly.
template <typename F>
inline int run_with_graph(md_model_s* m, hipStream_t st, const std::vector<uintptr_t>& key, bool eligible, F&& body) {
  if (!m->graph_enabled || !eligible || m->timing_enabled || m->taps_enabled) return body();
  {
    md_model_s::GraphEntry& e = m->graphs[key];  // not held across body(): a body that regrows a buffer drops every graph (and this entry)
    if (e.exec) {
      MD_HIP(hipGraphLaunch(e.exec, st));
      return MD_OK;
    }
    if (e.seen++ == 0) return body();
  }
  MD_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
  const int s = body();
  hipGraph_t g = nullptr;
  const hipError_t ce = hipStreamEndCapture(st, &g);
  if (s != MD_OK || ce != hipSuccess || !g) {
    if (g) (void)hipGraphDestroy(g);
    m->graphs.erase(key);
    if (s != MD_OK) return s;
    MD_FAIL(MD_ERR_HIP, "stream capture failed: %s", hipGetErrorString(ce));
  }
  hipGraphExec_t ex = nullptr;
  const hipError_t ie = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (ie != hipSuccess || !ex) {
    m->graphs.erase(key);
    MD_FAIL(MD_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(ie));
  }
  m->graphs[key].exec = ex;
  MD_HIP(hipGraphLaunch(ex, st));
  return MD_OK;
}

struct Run {
  md_model_s* m;
  hipStream_t st;
  int B;
  int pending = -1;
  void begin(const char* name) {
    if (!m->timing_enabled) return;
    if (!m->timing_filter.empty() && m->timing_filter != name) return;
    TimingEntry t;
    t.name = name;
    (void)hipEventCreate(&t.a);
    (void)hipEventCreate(&t.b);
    (void)hipEventRecord(t.a, st);
    m->timing.push_back(t);
    pending = (int)m->timing.size() - 1;
  }
  void end() {
    if (!m->timing_enabled || pending < 0) return;
    (void)hipEventRecord(m->timing[pending].b, st);
    pending = -1;
  }
  // NHWC T tensor -> NCHW fp32 tap
  int tap_nhwc(const char* name, const void* p, int C, int H, int W, long ld, int coff = 0) {
    if (!m->taps_enabled) return MD_OK;
    Tap& t = m->taps[name];
    const size_t n = (size_t)B * C * H * W;
    if (t.count != n) {
      if (t.dev) (void)hipFree(t.dev);
      MD_HIP(hipMalloc((void**)&t.dev, n * 4));
      t.count = n;
    }
    t.dims[0] = B; t.dims[1] = C; t.dims[2] = H; t.dims[3] = W;
    return launch_nhwc_to_nchw(p, B, C, H, W, ld, coff, t.dev, m->prec, st);
  }
  // token rows of an fp32 [B*S, width] tensor -> tap [B, nrows, dst_width] columns [coff, coff + width): rows row0 .. row0+nrows of
  // every sequence (the patch tokens of a hook, depth_anything3/mod.rs:344-347). Call once per column block.
  int tap_token_rows(const char* name, const float* src, int S, int row0, int nrows, int width, int dst_width, int coff) {
    if (!m->taps_enabled) return MD_OK;
    Tap& t = m->taps[name];
    const size_t n = (size_t)B * nrows * dst_width;
    if (t.count != n) {
      if (t.dev) (void)hipFree(t.dev);
      MD_HIP(hipMalloc((void**)&t.dev, n * 4));
      t.count = n;
    }
    t.dims[0] = B; t.dims[1] = nrows; t.dims[2] = dst_width; t.dims[3] = 0;
    for (int b = 0; b < B; ++b)
      MD_HIP(hipMemcpy2DAsync(t.dev + ((size_t)b * nrows) * dst_width + coff, (size_t)dst_width * 4, src + ((size_t)b * S + row0) * width,
                              (size_t)width * 4, (size_t)width * 4, (size_t)nrows, hipMemcpyDeviceToDevice, st));
    return MD_OK;
  }
  int tap_f32(const char* name, const float* p, int64_t d0, int64_t d1, int64_t d2, int64_t d3) {
    if (!m->taps_enabled) return MD_OK;
    Tap& t = m->taps[name];
    const size_t n = (size_t)d0 * std::max<int64_t>(d1, 1) * std::max<int64_t>(d2, 1) * std::max<int64_t>(d3, 1);
    if (t.count != n) {
      if (t.dev) (void)hipFree(t.dev);
      MD_HIP(hipMalloc((void**)&t.dev, n * 4));
      t.count = n;
    }
    t.dims[0] = d0; t.dims[1] = d1; t.dims[2] = d2; t.dims[3] = d3;
    MD_HIP(hipMemcpyAsync(t.dev, p, n * 4, hipMemcpyDeviceToDevice, st));
    return MD_OK;
  }
};

inline int cpad(const md_model_s* m, int ch) { return (ch + m->ke - 1) / m->ke * m->ke; }

// ---- split-half operands (MD_PREC_F16X2; every function below is the identity on the one-plane modes) ----
// Callers pass LOGICAL padded channel counts (row widths, K); an activation row is physically [hi | lo] and a weight row
// `terms` copies of its contraction (2: [W | W], 3: [Wh | Wh | Wl]). terms == 0 = the model's plain-weight form
// (md_model_s::wterms); the products composed at commit are never f16-exact and pass 3.
// (a fork reads its ROOT's term count: a re-commit on the root may change it while the fork lives, and the packed rows the
// fork's launches read are the root's)
inline int split_terms(const md_model_s* m, int terms) { return m->xm == 1 ? 1 : (terms > 0 ? terms : (m->parent ? m->parent : m)->wterms); }
// dense / indexed A operand of `kp` logical channels per row
inline void split_dense_a(const md_model_s* m, GemmParams& p, int kp, long lda, int terms) {
  const int t = split_terms(m, terms);
  p.K = kp * t;
  p.lda = lda * m->xm;
  p.a_wrap = t == 3 ? 2 * kp / m->ke : 0;
}
// 3x3-convolution A operand: NHWC pixels of `cin_p` logical channels
inline void split_conv_a(const md_model_s* m, GemmParams& p, int cin_p, int terms) {
  const int t = split_terms(m, terms);
  p.cC = cin_p * m->xm;
  p.cCk = m->xm == 1 ? 0 : cin_p * t;
  p.K = 9 * cin_p * t;
  p.a_wrap = t == 3 ? 2 * cin_p / m->ke : 0;
}
// T-typed output (and residual inputs) with `ldo` logical channels per row
inline void split_out(const md_model_s* m, GemmParams& p, long ldo, bool t_out) {
  if (!t_out) { p.ldo = ldo; return; }
  p.ldo = ldo * m->xm;
  p.o_plane = m->xm == 2 ? ldo : 0;
}

// 1x1 conv / linear over NHWC rows.  A may be gathered through `idx`.
inline int gemm_rows(Run& r, const char* name, const void* A, long lda, const int* idx, long M, const void* W, int N, int K,
              const float* bias, void* out, long ldo, int out_f32 = 0, int act = ACT_NONE, int terms = 0) {
  GemmParams p;
  p.N = N; p.ngroups = 1; p.g_rows[0] = (int)M; p.W[0] = W;
  p.A = A; p.a_index = idx;
  split_dense_a(r.m, p, K, lda, terms);
  p.epi = EPI_STORE; p.act = act; p.out_f32 = out_f32; p.bias[0] = bias; p.out = out;
  split_out(r.m, p, ldo, !out_f32);
  r.begin(name);
  int s = launch_gemm(p, idx ? A_INDEXED : A_DENSE, r.m->prec, TILE_AUTO, r.st);
  r.end();
  return s;
}

// ConvTranspose2d k=2 s=2 as GEMM + pixel shuffle (encoder.rs:61-69, decoder.rs:100-105, mod.rs:81-84)
inline int deconv2(Run& r, const char* name, const void* A, long lda, const int* idx, int h, int w, const void* W, int Cin_p,
            int Cout, const float* bias, void* out, long ldo, int coff, void* out2 = nullptr, int f = 2, int terms = 0) {
  GemmParams p;
  p.N = f * f * Cout; p.ngroups = 1; p.g_rows[0] = r.B * h * w; p.W[0] = W; p.ps_f = f;
  p.A = A; p.a_index = idx;
  split_dense_a(r.m, p, Cin_p, lda, terms);
  p.epi = EPI_PIXSHUF; p.bias[0] = bias; p.out = out; p.out2 = out2;
  split_out(r.m, p, ldo, true);
  p.psH = h; p.psW = w; p.psC = Cout; p.ps_coff = coff;
  r.begin(name);
  int s = launch_gemm(p, idx ? A_INDEXED : A_DENSE, r.m->prec, TILE_AUTO, r.st);
  r.end();
  return s;
}

// Conv2d 3x3 s1 p1 over NHWC as implicit GEMM (decoder.rs:55-72,167-175; mod.rs:78-87)
inline int conv3(Run& r, const char* name, const void* in, int H, int W, int Cin_p, const void* Wp, const float* bias,
          int Cout, void* out, long ldo, int act, const void* res1, const void* res2, void* out2, int terms = 0) {
  GemmParams p;
  p.N = Cout; p.ngroups = 1; p.g_rows[0] = r.B * H * W; p.W[0] = Wp;
  p.A = in; p.cH = H; p.cW = W; p.zero_page = r.m->zero_page;
  split_conv_a(r.m, p, Cin_p, terms);
  p.epi = EPI_STORE; p.act = act; p.bias[0] = bias; p.out = out; p.out2 = out2;
  split_out(r.m, p, ldo, true);
  p.res1 = res1; p.res2 = res2; p.ldr = p.ldo; p.r_plane = (res1 || res2) ? p.o_plane : 0;
  r.begin(name);
  int s = launch_gemm(p, A_CONV3, r.m->prec, TILE_AUTO, r.st);
  r.end();
  return s;
}


}  // namespace md
