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

static int pick_tile(const GemmParams& p) {
  if (p.epi == EPI_HEAD || p.N <= 32) return TILE_256x32;
  long rows = 0;
  int t256 = 0, t128 = 0;
  for (int g = 0; g < p.ngroups; ++g) {
    rows += p.g_rows[g];
    t256 += cdiv(p.g_rows[g], 256);
    t128 += cdiv(p.g_rows[g], 128);
  }
  if (p.N < 256 || rows < 256) return TILE_128x128;
  // Wave quantisation decides between the two: a launch takes whole rounds (256 CUs x one 256^2 tile, or x two 128^2
  // tiles), and a 128^2 round costs 0.78 of a 256^2 round -- measured on the Depth-Anything-v3 1036^2 shapes (M = 5477,
  // tools/kernel_bench.py da3_*): qkv 2 rounds of 256^2 54 us against 3 rounds of 128^2 64 us, fc2 1 round 81 us against
  // 1 round 74 us; the same ratio holds for the e4m3 operands. Large M always lands on 256^2 (half the rounds).
  const long b256 = (long)t256 * cdiv(p.N, 256), b128 = (long)t128 * cdiv(p.N, 128);
  const double c256 = (double)cdiv(b256, 256);
  const double c128 = (double)cdiv(b128, 512) * 0.78;
  return c256 <= c128 ? TILE_256x256 : TILE_128x128;
}

int launch_gemm(GemmParams p, int amode, int prec, int tile, hipStream_t stream) {
  const int ke = prec == MD_PREC_F32 ? 32 : (prec == MD_PREC_FP8 ? 128 : 64);
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
  if (tile == TILE_AUTO) tile = pick_tile(p);
  {
    const int bn = (tile == TILE_128x128) ? 128 : (tile == TILE_256x32 ? 32 : 256);
    const int tn = cdiv(p.N, bn);
    // raster group width, from a sweep on the Depth Pro step (B = 8): up to 12 n-tiles (qkv) plain n-fastest order is
    // best (qkv 24.3 -> 23.3 ms per step against groups of 4); the 16 n-tiles of fc1 run the same in groups of 4 or 8 and
    // slower ungrouped (34.5 / 34.7 / 34.9 ms)
    p.raster_gn = tn <= 12 ? 0 : 4;
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
