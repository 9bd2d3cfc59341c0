// Parameters of the MFMA GEMM family: C = A[M,K] * W[N,K]^T with grouped weights, three
// A-operand addressing modes (dense rows / indexed rows / 3x3-conv taps over NHWC) and fused
// epilogues. One kernel template serves every dense contraction of the Depth Pro path
// (SURVEY 8a rows a4, a7, a8, a9, a10, a11).
#pragma once

#include "../md_common.h"

namespace md {

constexpr int kMaxGroups = 4;

enum AMode : int { A_DENSE = 0, A_INDEXED = 1, A_CONV3 = 2 };

enum EpiMode : int {
  EPI_STORE = 0,        // out = act(acc + bias + res1 + res2); optional out2 = relu(out)
  EPI_RESID_LS = 1,     // x(f32) += scale[n] * (acc + bias[n])              (ViT residual + LayerScale)
  EPI_PATCH_EMBED = 2,  // x(f32)[seq*S + 1 + p] = acc + bias + pos[1+p]     (patch embed + pos embed)
  EPI_QKV = 3,          // q,k -> row-major T; v -> transposed V^T[seq][head][d][key]
  EPI_PIXSHUF = 4,      // ConvTranspose2d k=s=2 as GEMM: pixel-shuffle scatter (+bias, + relu copy)
  EPI_HEAD = 5,         // depth head tail: relu(conv1) . w_out + b_out, relu  -> f32 [M]
  EPI_HEAD_UP2 = 6      // the same tail behind the composed `deconv k2s2 -> conv 3x3` (N = 4 x 32: one 32-column group per
                        // output parity, bias = nine position-class vectors [9][32]) -> f32 [B][2H][2W]; conv3x3 A mode only
};

enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2 };

// Exact n / d for 0 <= n < 2^31 by a multiply-high and a shift (round-up method: mul = floor(2^(31+L)/d) + 1,
// L = ceil(log2 d)); the epilogues decompose row indices per output vector, where a hardware-less integer
// division (~40 VALU instructions on the 16-lane SIMDs) would dominate short-K GEMMs.
struct FastDiv {
  unsigned mul = 0, sh = 0, d = 1;
};
inline FastDiv make_fastdiv(int d) {
  FastDiv f;
  f.d = d > 0 ? (unsigned)d : 1u;
  if (f.d == 1) return f;
  unsigned L = 0;
  while ((1ull << L) < f.d) ++L;
  f.mul = (unsigned)((1ull << (31 + L)) / f.d + 1);
  f.sh = L - 1;
  return f;
}

struct GemmParams {
  // ---- problem ----
  int N = 0, K = 0;  // K in elements of T
  int ngroups = 1;
  int g_row0[kMaxGroups] = {0, 0, 0, 0};   // first logical row of each group (output row space)
  int g_rows[kMaxGroups] = {0, 0, 0, 0};   // rows in each group
  int g_arow0[kMaxGroups] = {0, 0, 0, 0};  // first PHYSICAL A row of each group (aliasing allowed)
  int g_tile0[kMaxGroups + 1] = {0, 0, 0, 0, 0};  // prefix sum of m-tiles (filled by the launcher)
  const void* W[kMaxGroups] = {nullptr, nullptr, nullptr, nullptr};  // [N][ldw] per group
  long ldw = 0;                 // elements between consecutive W rows (0 = K)
  // ---- batching (blockIdx.y): batch index by -> (bo, bi) = (by / batch_inner, by % batch_inner);
  //      A, W (all groups) and out advance by bo*stride[0] + bi*stride[1] elements ----
  int batch = 1, batch_inner = 1;
  long a_bs[2] = {0, 0}, w_bs[2] = {0, 0}, o_bs[2] = {0, 0};
  // ---- A operand ----
  const void* A = nullptr;
  long lda = 0;                 // elements between consecutive rows (dense / indexed)
  const int* a_index = nullptr; // A_INDEXED: logical row m reads physical row a_index[m]
  int cH = 0, cW = 0, cC = 0;   // A_CONV3: NHWC input [B, cH, cW, cC]; K = 9*cC (tap-major), pad 1
  int cOH = 0, cOW = 0, cstride = 1;  // output grid (0 = same as input) and stride; M = B*cOH*cOW
  const void* zero_page = nullptr;  // >= 256 zero bytes (A_CONV3 halo)
  // ---- split-half operands (T = f16s_t, MD_PREC_F16X2): an A row is [hi: Kp | lo: Kp] and a W row [W | W] (f16-exact
  //      weights, K = 2 Kp) or [Wh | Wh | Wl] (K = 3 Kp; the third block multiplies A's hi plane again). Both are plain
  //      K-contiguous rows to the main loop; only the A column of the third block wraps:
  int a_wrap = 0;   // > 0: A k-tile (dense / indexed) or channel block inside a tap (conv) kb reads kb - a_wrap when kb >= a_wrap
  int cCk = 0;      // A_CONV3: contraction channels per tap (0 = cC); K = 9 * cCk while the pixel stride stays cC
  long o_plane = 0; // element offset from the hi to the lo plane of out / out2 (the logical padded row width)
  long r_plane = 0; // the same for res1 / res2
  long v_plane = 0; // EPI_QKV: offset from the hi to the lo plane of V^T (q | k rows are [q_hi | q_lo | k_hi | k_lo], each `embed` wide)
  // ---- epilogue ----
  int epi = EPI_STORE;
  int act = ACT_NONE;
  int out_f32 = 0;              // primary output element type: 0 = T, 1 = float
  const float* bias[kMaxGroups] = {nullptr, nullptr, nullptr, nullptr};
  const float* scale[kMaxGroups] = {nullptr, nullptr, nullptr, nullptr};
  const float* pos[kMaxGroups] = {nullptr, nullptr, nullptr, nullptr};
  // fp8 operands: acc is multiplied by ascale * wscale[n] (activation scale x per-output-channel weight scale)
  // before the bias; out_fp8: the GELU store writes e4m3 (value * out_inv_scale), the next GEMM's A operand
  const float* wscale[kMaxGroups] = {nullptr, nullptr, nullptr, nullptr};
  float ascale = 1.f;
  int out_fp8 = 0;
  float out_inv_scale = 1.f;
  void* out = nullptr;
  long ldo = 0;
  const float* resid_src = nullptr;  // EPI_RESID_LS: out = resid_src + scale * (acc + bias) instead of the in-place update (same ld as out)
  void* out2 = nullptr;         // optional relu(out) copy, element type T, same ld
  const void* res1 = nullptr;   // optional residual inputs, element type T
  const void* res2 = nullptr;
  long ldr = 0;
  // ---- LayerNorm folded into the GEMMs on either side of it (round 6; the 256 x 256 kernel's EK 5 / 6 / 7, dense A, 16-bit operands) ----
  // LN(x) W^T + b  =  rstd_m * (round(gamma . x) W^T  -  mu_m * c)  +  d,   c[n] = sum_k gamma[k] W[n][k],  d[n] = b[n] + sum_k beta[k] W[n][k]:
  // the PRODUCER of x (EPI_RESID_LS: proj / fc2) also writes round_T(gamma_next . x_new) as the next GEMM's A operand and, per row and
  // 256-column tile, (sum x, sum x^2) of x_new in fp32; the CONSUMER (qkv / fc1) finishes its accumulators with the row's mu / rstd. The
  // stand-alone LayerNorm launch between them (read 4 B + write 2 B per element at HBM rate, 49 per ViT pass) is gone. The weights stay
  // untouched (f16 checkpoint weights remain exact split-half operands); c, d are built at commit from the operand-rounded weights.
  void* ln_out = nullptr;          // producer: [rows][ln_ldo] T (split-half: lo plane ln_plane elements behind the hi plane)
  long ln_ldo = 0, ln_plane = 0;
  const float* ln_gamma[kMaxGroups] = {nullptr, nullptr, nullptr, nullptr};  // producer: gamma of the NEXT LayerNorm [N]
  float* ln_stats_out = nullptr;   // producer: [rows][N / 256][2] fp32 partial (sum, sum of squares)
  const float* ln_stats = nullptr; // consumer: ln_raw = 1: the producer's array (4 partials per row, combined in the epilogue); 0: [rows][2] (rstd, -mu rstd) from ln_finish
  int ln_raw = 0;
  int ln_parts = 0;
  float ln_inv_n = 0.f, ln_eps = 0.f;
  const float* ln_c[kMaxGroups] = {nullptr, nullptr, nullptr, nullptr};      // consumer: c [N] (bias[] then carries d)
  // 256 x 256 kernel, lean 2-byte store kinds (EK 2 / 4 / 6 / 7, no residual inputs / second output): stores straight from the accumulator
  // layout, no LDS staging. The W tile's LDS image is filled in a PERMUTED row order (the LDS-DMA source rows; LDS addressing, bank
  // pattern and register use unchanged) such that a lane's accumulators of the n-blocks 2h, 2h + 1 are 8 CONSECUTIVE output columns: one
  // 16-byte store per (m-block, h), 16 rows x 64 bytes per wave-instruction, the two h of a row completing its 128-byte line back to
  // back. Same values, same bits as the staged form. Set by launch_gemm for eligible launches (gemm_direct_store()).
  int direct_store = 0;
  int persist = 0;   // set by launch_gemm (gemm_persistent()): which forms may run as persistent tile loops (1 fc1, 2 QKV, 8 3 x 3 convolutions -- lean: gemm256p_kernel, with residual inputs / second output: gemm256r_kernel; 4 read-modify-write: gemm256r_kernel)
  int ptiles = 0;    // persistent form: tiles of the launch (set by launch_256)
  int stagger = 0;   // read-modify-write tile loop: the odd workgroups of every XCD start this many 10-ns ticks late (gemm_stagger(); set by launch_256 per k-tile count)
  int ksplit_ok = 0;            // set by launch_gemm from gemm_allow_ksplit(): the 64 x 64 kernel may split K over wave groups (KSPLIT)
  int res_mod = 0;              // > 0: res1's row = m % res_mod (a per-image table shared by the batch, or an input two weight groups share); res2 is never wrapped
  // EPI_PATCH_EMBED / EPI_QKV
  int seq_stride = 0;           // rows per sequence in the token buffers (tokens padded to x4)
  int seq_patches = 0;          // patches per sequence (EPI_PATCH_EMBED input rows per sequence)
  int embed = 0;                // D
  int heads = 0;
  int kpad = 0;                 // padded key count of V^T rows
  void* vT = nullptr;
  float qscale = 1.f;           // EPI_QKV: the q columns (n < embed) are stored as (acc + bias) * qscale -- the softmax scale
                                // head_dim^-0.5 * log2(e) folded into q BEFORE its one rounding to the operand type, so that the
                                // attention kernel exponentiates the MFMA result directly (v_exp_f32 = 2^x)
  // EPI_QKV with DA3's per-head q / k LayerNorm(64) + 2-D rotary embedding fused in (tiles with BN == 64 = one head of q or k per
  // tile; set qkn_g[0] to ask for it -- the launcher then keeps to the 64-column tiles): q' = rope(LN_q(acc + bias)) * qscale,
  // k' = rope(LN_k(acc + bias)). Token t = row % seq_stride sits at (0, 0) for t == 0 or t >= rope_ntok, else at patch
  // (1 + (t-1) / rope_pw, 1 + (t-1) % rope_pw), or at (1, 1) when rope_global; the first half of the head rotates with the row
  // position, the second with the column position, pairs (j, j + 16), angles from rope_cos / rope_sin [pos][16].
  const float* qkn_g[2] = {nullptr, nullptr};  // q, k gamma [64]
  const float* qkn_b[2] = {nullptr, nullptr};  // q, k beta [64]
  float qkn_eps = 1e-5f;
  const float *rope_cos = nullptr, *rope_sin = nullptr;
  int rope_pw = 1, rope_global = 0, rope_ntok = 0;
  FastDiv fd_rope_pw;
  // EPI_PIXSHUF: input pixel grid [B, psH, psW]; N = f*f*psC (f = ps_f, 2 or 4); out NHWC [B, f*psH, f*psW, ldo] at +ps_coff
  int psH = 0, psW = 0, psC = 0, ps_coff = 0, ps_f = 2;
  int ps_fast = 0;  // set by launch_gemm: bf16 out, no second output, 8-column groups inside one tap, 32-bit element offsets
  // divisors of the epilogue index math (filled by the launcher from the fields above)
  FastDiv fd_res_mod, fd_seq_patches, fd_seq_stride, fd_psC, fd_psW, fd_psH, fd_ow, fd_oh, fd_cblocks;
  int raster_gn = 0;  // n-tiles per raster group (0 = all: plain n-fastest order); set by the launcher
  // block -> tile map, prepared by the launcher (prep_tile_map): the kernel prologue did three runtime integer
  // divisions (float-reciprocal sequences, ~150 dependent cycles each) per workgroup for it
  int map_gn = 1, map_gsz = 1, map_full = 0, map_full_gsz = 0, map_rn = 0;
  FastDiv fd_map_gsz, fd_map_gn, fd_map_rn;
  // timing-only ablations and stamps: honoured by the DIAGNOSTIC instantiation only (md_bench_gemm; results are WRONG
  // when a flag is set): bit0 no in-loop global->LDS loads, bit1 every k-tile re-reads k-tile 0, bit2 no global stores,
  // bit3 no staging writes
  int debug_flags = 0;
  unsigned long long* stamps = nullptr;  // timing-only: [blocks][16] stamps: 8 x s_memrealtime, then 2 x shader clock around the main loop (md_bench_gemm)
  // EPI_HEAD / EPI_HEAD_UP2
  const float* head_w = nullptr;  // [32]
  float head_b = 0.f;
  int head_act = 0;             // 0 relu (Depth Pro, mod.rs:111), 1 exp (DA3, dpt.rs:700), 2 linear, 3 exp + 1 (DA3 confidence, dpt.rs:497)
  // EPI_HEAD with several output channels in ONE launch (the DA3 heads: depth + confidence; 6 ray channels + confidence --
  // dpt.rs:481-513 runs the same 3x3 `reduce` convolution for all of them): channel c = act_c(relu(conv + b1) . head_wc[c] + b_c),
  // written to head_out[c][img * head_bstride[c] + pixel] with img = m / head_plane. head_nch == 0: the one-channel form above.
  int head_nch = 0;
  int head_plane = 0;           // pixels per image (rows m per image)
  FastDiv fd_head_plane;
  float* head_out[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  const float* head_wc[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // [32] each
  long head_bstride[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float head_bs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int head_acts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

// The contraction split inside the workgroup (gemm_kernel's KSPLIT) changes the summation order of a launch that is small enough to
// take it, so a result may differ in its last bits between batch sizes. Depth Pro promises (and tests) bit-identical images across batch
// sizes: the split is off unless the calling thread turned it on -- the Depth-Anything-v3 engine does, around each of its calls.
// process-wide A/B switch of GemmParams::direct_store (default 1); returns the previous value
int gemm_direct_store(int on);
// process-wide A/B switch of GemmParams::persist (a mask: 1 the fc1 form, 2 the QKV projection, 4 the read-modify-write GEMMs, 8 the lean 3 x 3 convolutions; default 15); returns the previous value
int gemm_persistent(int mask);
// start offset (10-ns ticks of the constant 100 MHz counter) between the two halves of a tile loop's workgroups. which: 0 the read-modify-write loop at
// <= 16 k-tiles per tile (proj), 1 the same at more (fc2), 2 the fc1 loop, 3 the QKV loop
void gemm_stagger(int which, int ticks);
int gemm_stagger_ticks(int which);
int gemm_allow_ksplit(int on);  // per host thread; returns the previous value
void gemm_count_ksplit_launch();   // diagnostics: md_gemm_ksplit_launches()
long long gemm_ksplit_launches();
struct KsplitScope {
  explicit KsplitScope(int on) : prev_(gemm_allow_ksplit(on)) {}
  ~KsplitScope() { gemm_allow_ksplit(prev_); }  // nested scopes restore the outer value
  KsplitScope(const KsplitScope&) = delete;
  KsplitScope& operator=(const KsplitScope&) = delete;

 private:
  int prev_;
};

enum GemmTile : int { TILE_256x256 = 0, TILE_128x128 = 1, TILE_256x32 = 2, TILE_128x64 = 3, TILE_64x64 = 4, TILE_AUTO = 99 };

// Launches the kernel. `prec`: MD_PREC_BF16 (T = bf16), MD_PREC_F16 (T = f16), MD_PREC_F16X2 (T = f16s: split-half planes),
// MD_PREC_F32 (T = float) or MD_PREC_FP8 (T = e4m3, dense only).
int launch_gemm(GemmParams p, int amode, int prec, int tile, hipStream_t stream);
// The tile TILE_AUTO resolves to for these parameters (host-only).
int gemm_pick_tile(const GemmParams& p, int prec);

}  // namespace md
