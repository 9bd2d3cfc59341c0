#include <atomic>
// Host-side validation, tile selection and precision dispatch of the MFMA GEMM family.
// The kernel template lives in gemm_impl.h and is instantiated per precision in
// gemm_bf16.hip / gemm_f32.hip (separate translation units so they compile in parallel).
#include "gemm.h"

namespace md {

int launch_gemm_bf16(GemmParams& p, int amode, int tile, hipStream_t stream);
int launch_gemm_f32(GemmParams& p, int amode, int tile, hipStream_t stream);
int launch_gemm_f16(GemmParams& p, int amode, int tile, hipStream_t stream);
int launch_gemm_fp8(GemmParams& p, int amode, int tile, hipStream_t stream);
int launch_gemm_f16x2(GemmParams& p, int amode, int tile, hipStream_t stream);

// Tile choice = a small cost model of one launch, fitted on MI355X (tools/kernel_bench.py, profiles/r03_tile_model.txt):
// the tiles of a launch are spread over the 256 CUs, a CU works through its n = ceil(tiles / 256) tiles in batches of `cap`
// co-resident workgroups, and a batch of w workgroups advances one 128-byte k-tile in max(lat, w * thr) microseconds -- `lat`
// is the latency-bound time of a lone workgroup (two-stage ring: one memory round trip per k-tile), `thr` the CU's
// throughput-bound time per workgroup and k-tile. t0 = dispatch + first-tile latency + epilogue. The fit reproduces the
// measured times of 24 shapes x 4 tiles to about 10 % (Depth Pro proj / fc2 at B = 1, the Depth-Anything-v3 small / base /
// large linears at 518^2 and 1036^2, the 64- and 256-feature DPT-head convolutions at 37^2 .. 296^2). What it encodes:
// large launches belong on the 256^2 kernel (1.38 us per 256^2 x 64 step against 2.1 - 3.5 for the same area on the smaller
// tiles), launches of less than a round or two belong on the tile that fills the CUs (the 1370-token linears of
// Depth-Anything-v3 small: 99 tiles of 128^2 = 13.6 us, 396 tiles of 64^2 = 7.7 us).
struct TileModel {
  int id, bm, bn, cap;
  double lat, thr, t0;
};
static const TileModel kTileModel[] = {{TILE_256x256, 256, 256, 1, 1.38, 1.38, 5.6},
                                       {TILE_128x128, 128, 128, 2, 0.75, 0.52, 9.0},
                                       {TILE_128x64, 128, 64, 3, 0.42, 0.32, 7.0},
                                       {TILE_64x64, 64, 64, 4, 0.24, 0.22, 5.0}};

static int pick_tile(const GemmParams& p, int prec) {
  if (p.epi == EPI_HEAD || p.N <= 32) return TILE_256x32;
  const int ke = prec == MD_PREC_F32 ? 32 : (prec == MD_PREC_FP8 ? 128 : 64);
  const double kt = (double)(p.K / ke);
  const double slow = prec == MD_PREC_F32 ? 8.0 : 1.0;  // fp32 MFMA: 1/16 of the FLOPs per clock on half the k per tile
  long rows = 0;
  for (int g = 0; g < p.ngroups; ++g) rows += p.g_rows[g];
  int best = TILE_128x128;
  double best_cost = 1e30;
  for (const TileModel& t : kTileModel) {
    if (t.id == TILE_256x256 && (p.N < 256 || rows < 256)) continue;
    long wgs = 0;
    for (int g = 0; g < p.ngroups; ++g) wgs += cdiv(p.g_rows[g], t.bm);
    wgs *= cdiv(p.N, t.bn) * (long)(p.batch > 1 ? p.batch : 1);
    const long n = (wgs + 255) / 256, nb = n / t.cap, r = n % t.cap;
    const double thr = t.thr * slow;
    const double per_k = nb * std::max(t.lat, t.cap * thr) + (r ? std::max(t.lat, r * thr) : 0.0);
    const double cost = t.t0 + kt * per_k;
    if (cost < best_cost) {
      best_cost = cost;
      best = t.id;
    }
  }
  return best;
}

int gemm_pick_tile(const GemmParams& p, int prec) { return pick_tile(p, prec); }

static std::atomic<int> g_direct_store{1};
int gemm_direct_store(int on) { return g_direct_store.exchange(on ? 1 : 0); }
static std::atomic<int> g_persist{15};
int gemm_persistent(int mask) { return g_persist.exchange(mask & 15); }
// (proj at B = 8, 2528 tiles of 16 k-tiles: 13.43 -> 12.72 ms per step with 20 us, 12.92 with 10, 13.03 with 30; fc2, 64 k-tiles: no effect at 15 / 30 us --
// profiles/r06_stagger_ab.txt)
static std::atomic<int> g_stagger[4] = {{2000}, {0}, {0}, {0}};
void gemm_stagger(int which, int ticks) {
  if (which >= 0 && which < 4 && ticks >= 0) g_stagger[which].store(ticks);
}
int gemm_stagger_ticks(int which) { return g_stagger[which & 3].load(std::memory_order_relaxed); }
static thread_local int g_ksplit_ok = 0;
int gemm_allow_ksplit(int on) {  // returns the previous value (KsplitScope restores it)
  const int prev = g_ksplit_ok;
  g_ksplit_ok = on ? 1 : 0;
  return prev;
}
static std::atomic<long long> g_ksplit_launches{0};
void gemm_count_ksplit_launch() { g_ksplit_launches.fetch_add(1, std::memory_order_relaxed); }
long long gemm_ksplit_launches() { return g_ksplit_launches.load(std::memory_order_relaxed); }

int launch_gemm(GemmParams p, int amode, int prec, int tile, hipStream_t stream) {
  const int ke = prec == MD_PREC_F32 ? 32 : (prec == MD_PREC_FP8 ? 128 : 64);
  p.ksplit_ok = g_ksplit_ok;
  p.direct_store = g_direct_store.load(std::memory_order_relaxed);
  p.persist = g_persist.load(std::memory_order_relaxed);
  if (p.ngroups < 1 || p.ngroups > kMaxGroups) MD_FAIL(MD_ERR_INVALID_ARG, "gemm: ngroups %d", p.ngroups);
  if (p.N <= 0 || p.N % 4 != 0) MD_FAIL(MD_ERR_UNSUPPORTED, "gemm: N=%d must be a positive multiple of 4", p.N);
  if (p.K <= 0 || p.K % ke != 0) MD_FAIL(MD_ERR_UNSUPPORTED, "gemm: K=%d must be a multiple of %d", p.K, ke);
  if (amode == A_CONV3) {
    if (p.cstride < 1) MD_FAIL(MD_ERR_INVALID_ARG, "conv3x3: stride %d", p.cstride);
    const int cck = p.cCk > 0 ? p.cCk : p.cC;  // contraction channels per tap (split-half operands: 2 or 3 planes' worth)
    if (p.cC % ke != 0 || cck % ke != 0 || p.K != 9 * cck)
      MD_FAIL(MD_ERR_UNSUPPORTED, "conv3x3: Cin=%d (contraction %d per tap) must be a multiple of %d (K=%d)", p.cC, cck, ke, p.K);
    if (!p.zero_page) MD_FAIL(MD_ERR_INVALID_ARG, "conv3x3: zero page missing");
  }
  if (amode == A_INDEXED && !p.a_index) MD_FAIL(MD_ERR_INVALID_ARG, "gemm: index table missing");
  if (p.batch > 1 && (p.epi != EPI_STORE || p.res1 || p.res2 || p.batch > 65535 || p.batch_inner < 1))
    MD_FAIL(MD_ERR_UNSUPPORTED, "gemm: batching supports the plain store epilogue only (batch=%d)", p.batch);
  if (p.epi == EPI_HEAD && p.N != 32) MD_FAIL(MD_ERR_UNSUPPORTED, "head epilogue needs N == 32");
  if (p.epi == EPI_HEAD && p.head_nch == 0 && !p.head_w) MD_FAIL(MD_ERR_INVALID_ARG, "head epilogue: output weights missing");
  if (p.epi == EPI_HEAD_UP2) {
    if (p.N != 128 || amode != A_CONV3 || p.cstride != 1 || p.cOW > 0 || p.cOH > 0 || p.ngroups != 1 || !p.bias[0] || !p.head_w)
      MD_FAIL(MD_ERR_UNSUPPORTED, "composed head epilogue: a stride-1 3x3 convolution with N == 4 x 32 and the 9 x 32 bias table");
    tile = TILE_128x128;
  }
  for (int g = 0; g < p.ngroups; ++g) {
    if (p.g_rows[g] <= 0) MD_FAIL(MD_ERR_INVALID_ARG, "gemm: group %d has %d rows", g, p.g_rows[g]);
    if (!p.W[g]) MD_FAIL(MD_ERR_INVALID_ARG, "gemm: group %d has no weights", g);
  }
  p.fd_res_mod = make_fastdiv(p.res_mod);
  p.fd_seq_patches = make_fastdiv(p.seq_patches);
  p.fd_seq_stride = make_fastdiv(p.seq_stride);
  p.fd_rope_pw = make_fastdiv(p.rope_pw);
  if (p.qkn_g[0]) {
    if (p.epi != EPI_QKV || !p.qkn_g[1] || !p.qkn_b[0] || !p.qkn_b[1] || !p.rope_cos || !p.rope_sin || p.embed % 64 != 0 || p.seq_stride <= 0 ||
        prec == MD_PREC_FP8)
      MD_FAIL(MD_ERR_INVALID_ARG, "gemm: the fused q/k-norm + RoPE epilogue needs EPI_QKV, both norms, the angle tables and 64-wide heads");
    if (tile != TILE_AUTO && tile != TILE_64x64 && tile != TILE_128x64)
      MD_FAIL(MD_ERR_UNSUPPORTED, "gemm: the fused q/k-norm + RoPE epilogue lives in the 64-column tiles");
  }
  p.fd_psC = make_fastdiv(p.psC);
  if (p.epi == EPI_PIXSHUF) {
    long px = 0;
    for (int g = 0; g < p.ngroups; ++g) px = std::max<long>(px, (long)p.g_row0[g] + p.g_rows[g]);
    const double out_elems = (double)px * p.ps_f * p.ps_f * (double)p.ldo;
    bool no_scale = true;
    for (int g = 0; g < p.ngroups; ++g) no_scale = no_scale && !p.wscale[g];
    p.ps_fast = prec != MD_PREC_F32 && !p.out_f32 && !p.out2 && !p.out_fp8 && no_scale && p.psC % 8 == 0 && p.N % 8 == 0 &&
                p.ldo % 8 == 0 && p.ps_coff % 8 == 0 && out_elems < 4.0e9;
  }
  if (p.epi == EPI_HEAD && p.head_nch > 0) {
    if (p.head_nch > 8 || p.head_plane <= 0) MD_FAIL(MD_ERR_INVALID_ARG, "head epilogue: %d channels / plane %d", p.head_nch, p.head_plane);
    for (int c = 0; c < p.head_nch; ++c)
      if (!p.head_out[c] || !p.head_wc[c]) MD_FAIL(MD_ERR_INVALID_ARG, "head epilogue: output %d missing", c);
    p.fd_head_plane = make_fastdiv(p.head_plane);
  }
  p.fd_psW = make_fastdiv(p.psW);
  p.fd_psH = make_fastdiv(p.psH);
  p.fd_ow = make_fastdiv(p.cOW > 0 ? p.cOW : p.cW);
  p.fd_oh = make_fastdiv(p.cOH > 0 ? p.cOH : p.cH);
  p.fd_cblocks = make_fastdiv((p.cCk > 0 ? p.cCk : p.cC) / ke);
  {
    long rows = 0;
    for (int g = 0; g < p.ngroups; ++g) rows = std::max<long>(rows, (long)p.g_arow0[g] + p.g_rows[g]);
    if (rows >= (1L << 31)) MD_FAIL(MD_ERR_UNSUPPORTED, "gemm: more than 2^31 rows");
  }
  if (tile == TILE_AUTO) tile = pick_tile(p, prec);
  if (p.qkn_g[0] && tile != TILE_64x64 && tile != TILE_128x64) {  // one head per tile: the 64-column tiles only
    long rows = 0;
    for (int g = 0; g < p.ngroups; ++g) rows += p.g_rows[g];
    tile = rows >= 4096 ? TILE_128x64 : TILE_64x64;
  }
  {
    const int bn = (tile == TILE_128x128) ? 128 : (tile == TILE_256x32 ? 32 : ((tile == TILE_128x64 || tile == TILE_64x64) ? 64 : 256));
    const int tn = cdiv(p.N, bn);
    // raster group width, from a sweep on the Depth Pro step (B = 8): up to 12 n-tiles (qkv) plain n-fastest order is
    // best (qkv 24.3 -> 23.3 ms per step against groups of 4); the 16 n-tiles of fc1 run the same in groups of 4 or 8 and
    // slower ungrouped (34.5 / 34.7 / 34.9 ms)
    // Round 5 re-measured it with the fabric-side counter (tools/probes/raster_traffic.sh -> profiles/r05_raster_traffic.txt): qkv in groups
    // of 4 reads HALF the fabric bytes of the plain order (800 against 1572 MB per launch at B = 4) and runs 4 % faster STAND-ALONE (984
    // against 944 TFLOP/s) -- and 3.8 % SLOWER inside the model on one box, twice (24.10 / 24.15 against 24.98 / 25.13 ms per step; fc1 the
    // other way, 36.3 against 35.8: profiles/r05_raster_inmodel.txt). The bytes these launches re-read from the fabric are not what
    // their time is made of; the rule stays.
    p.raster_gn = tn <= 12 ? 0 : 4;
#ifdef MD_DIAG_KNOBS
    // DIAG builds only: MD_GEMM_RASTER_GN = n-tiles per raster group (0 = plain n-fastest) for the L2-traffic A/B of
    // tools/probes/raster_traffic.sh -- the shipped library reads no environment variable
    if (const char* e = getenv("MD_GEMM_RASTER_GN")) p.raster_gn = atoi(e);
#endif
  }
  if (p.epi == EPI_HEAD) tile = TILE_256x32;
  if (prec == MD_PREC_F32) return launch_gemm_f32(p, amode, tile, stream);
  if (prec == MD_PREC_FP8) {
    if (p.out_fp8 && !(p.epi == EPI_STORE && p.act == ACT_GELU)) MD_FAIL(MD_ERR_UNSUPPORTED, "fp8 output is built for the GELU store only");
    return launch_gemm_fp8(p, amode, tile, stream);
  }
  if (p.out_fp8) MD_FAIL(MD_ERR_UNSUPPORTED, "fp8 output needs fp8 operands");
  if (prec == MD_PREC_F16) return launch_gemm_f16(p, amode, tile, stream);
  if (prec == MD_PREC_F16X2) {
    if (p.epi == EPI_QKV && (p.embed % 4 != 0 || p.v_plane <= 0)) MD_FAIL(MD_ERR_INVALID_ARG, "split-half qkv: V^T plane offset missing");
    if (!p.out_f32 && (p.epi == EPI_STORE || p.epi == EPI_PIXSHUF) && p.o_plane <= 0) MD_FAIL(MD_ERR_INVALID_ARG, "split-half output: plane offset missing");
    if ((p.res1 || p.res2) && p.r_plane <= 0) MD_FAIL(MD_ERR_INVALID_ARG, "split-half residual: plane offset missing");
    return launch_gemm_f16x2(p, amode, tile, stream);
  }
  if (p.a_wrap || p.cCk || p.o_plane || p.r_plane || p.v_plane) MD_FAIL(MD_ERR_INVALID_ARG, "gemm: split-half fields set for a one-plane precision");
  return launch_gemm_bf16(p, amode, tile, stream);
}

}  // namespace md
