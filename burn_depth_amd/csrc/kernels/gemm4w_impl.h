// The long-K member of the MFMA GEMM family: 256x256 tile, FOUR waves (2 x 2), one per SIMD, wave tile 128 x 128.
//
// Why a second main loop: the 8-wave kernel (gemm256_kernel) runs two waves per SIMD that take turns
// -- LDS fragment reads + LDS-DMA issue | barrier | 32 MFMAs | barrier -- and pays four s_barrier round trips per k-tile and
// 12 fragment reads per 32 MFMAs. With ONE wave per SIMD owning a 128 x 128 accumulator (256 registers of the 512-entry
// unified file):
//   * 16 fragment reads feed 64 MFMAs (2/3 of the LDS read bytes per FLOP);
//   * one barrier per k-tile instead of four;
//   * nothing alternates: the wave's own LDS reads, LDS writes and global loads issue in the vector-issue slots its MFMAs
//     leave free (an MFMA 16x16x32 holds the issue port 8 of its 16 cycles, MI355X_MICROARCH.md cycle constants), so the
//     matrix pipe is fed back to back as long as the compiler interleaves them -- which is why the operands are staged through
//     REGISTERS here (global_load -> ds_write_b128): an LDS-DMA wave-instruction costs ~60 issue cycles of the one wave that
//     also has to issue the MFMAs, a register load + ds_write ~25.
// K loop, two 32-deep k-steps per 64-deep k-tile t (stage c = t & 1 of a two-stage LDS ring, 64 KB per stage):
//   k-step 0: 64 MFMAs on fragments F0(t) | ds_read F1(t) from stage c | ds_write tile t+1 (registers) -> stage c^1, each write
//             followed by the global load of tile t+2 into the registers it freed
//   barrier(t)  (every wave has written tile t+1 and finished reading stage c)
//   k-step 1: 64 MFMAs on F1(t) | ds_read F0(t+1) from stage c^1
// Same LDS image (128-byte rows, 16-byte chunks XOR-swizzled by (row >> 1) & 7), same MFMA operand roles (weight tile = MFMA
// A operand: a lane ends with 4 consecutive output columns of one row) and the same generic epilogue (epilogue4) as the
// 8-wave kernels, so every epilogue kind, grouped weights and indexed A rows work unchanged.
#pragma once

#include "gemm_impl.h"

namespace md {

// ABL: timing-only ablations for md_bench_gemm (results are WRONG): bit 0 no in-loop global loads, bit 1 no LDS writes, bit 2 no
// barrier, bit 3 no fragment reads. The engine launches ABL = 0 only.
template <typename T, int AMODE, int ABL = 0>
__global__ __launch_bounds__(256, 1) void gemm4w_kernel(const GemmParams p) {
  constexpr int BM = 256, BN = 256, NW = 4;
  constexpr int WTM = 128, WTN = 128;
  constexpr int STAGE_BYTES = (BM + BN) * 128;  // A tile | W tile
  typedef typename OutT<T>::type TO;
  constexpr int ESZ = (int)sizeof(T);
  constexpr int KE = 128 / ESZ;
  constexpr int RG = 8;  // row groups (8 rows = 1 KB) of each operand tile per wave: 32 / 4
  static_assert(AMODE == A_DENSE || AMODE == A_INDEXED, "the 4-wave kernel takes dense or indexed A rows");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int nwg = gridDim.x;
  int id;
  {
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  int tile_n, tile_mg;
  if (id < p.map_full_gsz) {
    const int ng = fdiv(id, p.fd_map_gsz), r = id - ng * p.map_gsz;
    tile_mg = fdiv(r, p.fd_map_gn);
    tile_n = ng * p.map_gn + (r - tile_mg * p.map_gn);
  } else {
    const int r = id - p.map_full_gsz;
    tile_mg = fdiv(r, p.fd_map_rn);
    tile_n = p.map_full * p.map_gn + (r - tile_mg * p.map_rn);
  }
  int g = 0;
#pragma unroll
  for (int i = 1; i < kMaxGroups; ++i)
    if (i < p.ngroups && tile_mg >= p.g_tile0[i]) g = i;
  const int g_row0 = MD_SEL_G(p.g_row0, g);
  const int g_arow0 = MD_SEL_G(p.g_arow0, g);
  const int m_base = g_row0 + (tile_mg - MD_SEL_G(p.g_tile0, g)) * BM;
  const int m_end = g_row0 + MD_SEL_G(p.g_rows, g);
  const int n0 = tile_n * BN;
  const char* Wg = (const char*)MD_SEL_G(p.W, g);
  const char* Ab = (const char*)p.A;
  long out_boff = 0;
  if (p.batch > 1) {
    const int by = blockIdx.y;
    const int bo = by / p.batch_inner, bi = by - bo * p.batch_inner;
    Ab += (bo * p.a_bs[0] + bi * p.a_bs[1]) * ESZ;
    Wg += (bo * p.w_bs[0] + bi * p.w_bs[1]) * ESZ;
    out_boff = bo * p.o_bs[0] + bi * p.o_bs[1];
  }

  // ---- per-lane global sources: row group (i * 4 + wave), row inside the group lane >> 3, swizzled 16-byte chunk. Buffer
  //      loads: one descriptor per operand and ONE 32-bit byte offset per lane and row group (16 registers instead of 32 for
  //      64-bit pointers -- the 256 accumulators leave 256 registers for fragments, staging and addresses); the k-tile's
  //      offset is a scalar. launch_4w checks that both operands span less than 4 GB. ----
  const int lrow = lane >> 3, pc = lane & 7;
  const long ldw = p.ldw > 0 ? p.ldw : (long)p.K;
  const auto a_srd = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, (int)0x7fffffff, 0x00020000);
  const auto w_srd = __builtin_amdgcn_make_buffer_rsrc((void*)Wg, 0, (int)0x7fffffff, 0x00020000);
  unsigned voA[RG], voW[RG];
#pragma unroll
  for (int i = 0; i < RG; ++i) {
    const int r = (i * NW + wave) * 8 + lrow;
    const int lc = pc ^ ((r >> 1) & 7);
    int m = m_base + r;
    m = m < m_end ? m : m_end - 1;
    const long am = (long)g_arow0 + (m - g_row0);
    const long arow = AMODE == A_DENSE ? am : (long)p.a_index[am];
    voA[i] = (unsigned)(arow * (long)(p.lda * ESZ) + lc * 16);
    int n = n0 + r;
    n = n < p.N ? n : p.N - 1;
    voW[i] = (unsigned)((long)n * (ldw * ESZ) + lc * 16);
  }
  const int KT = p.K / KE;

  i32x4_t ra[RG], rw[RG];  // one k-tile of this wave's share of both operands in flight
  auto gload1 = [&](int kt, int i) __attribute__((always_inline)) {  // row group i of both operands of k-tile kt
    const int ka = ((p.a_wrap > 0 && kt >= p.a_wrap) ? kt - p.a_wrap : kt) * 128;
    ra[i] = __builtin_bit_cast(i32x4_t, __builtin_amdgcn_raw_buffer_load_b128(a_srd, (int)voA[i], ka, 0));
    rw[i] = __builtin_bit_cast(i32x4_t, __builtin_amdgcn_raw_buffer_load_b128(w_srd, (int)voW[i], kt * 128, 0));
  };
  const int wofs = wave * 1024 + lane * 16;  // this lane's slot inside row group (i * 4 + wave): + i * 4096
  auto lds_write1 = [&](int stage, int i) __attribute__((always_inline)) {
    char* sb = smem + stage * STAGE_BYTES + wofs + i * 4096;
    *(i32x4_t*)sb = ra[i];
    *(i32x4_t*)(sb + BM * 128) = rw[i];
  };

  f32x4acc_t acc[8][8];  // [n16 tile][m16 tile]
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = (f32x4acc_t){0.f, 0.f, 0.f, 0.f};
  const int q16 = lane >> 4, r16 = lane & 15;
  const int lane_off = r16 * 128 + ((((r16 >> 1) & 7) ^ q16) << 4);

  i32x4_t af[2][8], wf[2][8];  // fragments of the two k-steps (double buffer)
  auto frag_a = [&](int stage, int ks, int b) __attribute__((always_inline)) {
    return *(const i32x4_t*)(smem + stage * STAGE_BYTES + wm * WTM * 128 + b * 2048 + (lane_off ^ (ks << 6)));
  };
  auto frag_w = [&](int stage, int ks, int a) __attribute__((always_inline)) {
    return *(const i32x4_t*)(smem + stage * STAGE_BYTES + BM * 128 + wn * WTN * 128 + a * 2048 + (lane_off ^ (ks << 6)));
  };
  // LLVM SchedGroupMask: MFMA 0x8, VMEM_READ 0x20, DS_READ 0x100, DS_WRITE 0x200
#define MD_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

  // ---- prologue: tile 0 -> stage 0, tile 1 -> registers, F0(0) ----
#pragma unroll
  for (int i = 0; i < RG; ++i) gload1(0, i);
#pragma unroll
  for (int i = 0; i < RG; ++i) lds_write1(0, i);
  if (KT > 1) {
#pragma unroll
    for (int i = 0; i < RG; ++i) gload1(1, i);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    wf[0][j] = frag_w(0, 0, j);
    af[0][j] = frag_a(0, 0, j);
  }

  // One k-tile. WR: tile t+1 exists (write it to the other stage); LD: tile t+2 exists (load it into the freed registers).
  // The MFMAs are inline asm with the accumulator pinned to the AGPR half of the register file ("+a"): left to hipcc, the
  // 256 accumulators of a one-wave-per-SIMD kernel end up split between AGPRs and VGPRs and every MFMA drags
  // v_accvgpr_read / _mov / _write copies behind it (242 of them per k-tile in the first build of this kernel). asm volatile
  // statements keep their program order; a sched_barrier on either side of every memory instruction pins it into the MFMA
  // gap the source puts it in: per MFMA (16 cycles of the matrix pipe, 8 of them free on the issue port) at most ONE
  // fragment read, LDS write or global load.
  constexpr bool kBf16 = std::is_same<T, bf16_t>::value;
  // (a macro, not a lambda: an accumulator passed by reference keeps a copy of the whole array in scratch)
#define mfma(w_, a_, c_)                                                                              \
  do {                                                                                                \
    if constexpr (kBf16)                                                                              \
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c_) : "v"(w_), "v"(a_));          \
    else                                                                                              \
      asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c_) : "v"(w_), "v"(a_));           \
  } while (0)
#define MD_PIN(stmt)                        \
  do {                                      \
    __builtin_amdgcn_sched_barrier(0);      \
    stmt;                                   \
    __builtin_amdgcn_sched_barrier(0);      \
  } while (0)
  auto ktile = [&](int t, auto wr_c, auto ld_c) __attribute__((always_inline)) {
    constexpr bool WR = decltype(wr_c)::value, LD = decltype(ld_c)::value;
    const int c = t & 1;
    const int ka = ((p.a_wrap > 0 && t + 2 >= p.a_wrap) ? t + 2 - p.a_wrap : t + 2) * 128, kw = (t + 2) * 128;
    char* wr_base = smem + (c ^ 1) * STAGE_BYTES + wofs;
    // ---- k-step 0: MFMAs on F0(t); F1(t) <- stage c; registers -> stage c^1; tile t+2 -> registers ----
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      mfma(wf[0][a], af[0][0], acc[a][0]);
      if constexpr (!(ABL & 8)) MD_PIN(wf[1][a] = frag_w(c, 1, a));
      mfma(wf[0][a], af[0][1], acc[a][1]);
      if constexpr (!(ABL & 8)) MD_PIN(af[1][a] = frag_a(c, 1, a));
      mfma(wf[0][a], af[0][2], acc[a][2]);
      if constexpr (WR && !(ABL & 2)) MD_PIN(*(i32x4_t*)(wr_base + a * 4096) = ra[a]);
      mfma(wf[0][a], af[0][3], acc[a][3]);
      if constexpr (WR && !(ABL & 2)) MD_PIN(*(i32x4_t*)(wr_base + BM * 128 + a * 4096) = rw[a]);
      mfma(wf[0][a], af[0][4], acc[a][4]);
      if constexpr (LD && !(ABL & 1)) MD_PIN(ra[a] = __builtin_bit_cast(i32x4_t, __builtin_amdgcn_raw_buffer_load_b128(a_srd, (int)voA[a], ka, 0)));
      mfma(wf[0][a], af[0][5], acc[a][5]);
      if constexpr (LD && !(ABL & 1)) MD_PIN(rw[a] = __builtin_bit_cast(i32x4_t, __builtin_amdgcn_raw_buffer_load_b128(w_srd, (int)voW[a], kw, 0)));
      mfma(wf[0][a], af[0][6], acc[a][6]);
      mfma(wf[0][a], af[0][7], acc[a][7]);
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // F1(t) has arrived, tile t+1 is written
    if constexpr (!(ABL & 4)) __builtin_amdgcn_s_barrier();  // every wave: the same -> stage c is free, stage c^1 is complete
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // ---- k-step 1: MFMAs on F1(t); F0(t+1) <- stage c^1 ----
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      mfma(wf[1][a], af[1][0], acc[a][0]);
      if constexpr (WR && !(ABL & 8)) MD_PIN(wf[0][a] = frag_w(c ^ 1, 0, a));
      mfma(wf[1][a], af[1][1], acc[a][1]);
      if constexpr (WR && !(ABL & 8)) MD_PIN(af[0][a] = frag_a(c ^ 1, 0, a));
#pragma unroll
      for (int b = 2; b < 8; ++b) mfma(wf[1][a], af[1][b], acc[a][b]);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  {
    int t = 0;
    for (; t + 2 < KT; ++t) ktile(t, std::true_type(), std::true_type());
    if (t + 1 < KT) {
      ktile(t, std::true_type(), std::false_type());
      ++t;
    }
    ktile(t, std::false_type(), std::false_type());
  }
#undef MD_SGB
#undef MD_PIN
#undef mfma
  // the compiler does not see inside the MFMA statements: their results must have left the matrix pipe before it reads an accumulator
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

  // ---- epilogue: generic per-vector form straight from the accumulator layout ----
  if (p.epi == EPI_HEAD || p.epi == EPI_HEAD_UP2) return;  // not routed here (launch_gemm)
  // (full unrolling is forced: a rolled loop would index the accumulator array at run time and keep all of it in scratch)
#pragma clang loop unroll(full)
  for (int a = 0; a < 8; ++a)
#pragma clang loop unroll(full)
    for (int b = 0; b < 8; ++b) {
      const int m = m_base + wm * WTM + b * 16 + r16;
      const int n = n0 + wn * WTN + a * 16 + 4 * q16;
      if (m < m_end && n < p.N) {
        const f32x4_t v = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
        epilogue4<TO>(p, g, m, n, v, out_boff);
      }
    }
}

template <typename T, int AMODE>
static int launch_4w(GemmParams& p, hipStream_t stream) {
  constexpr int BM = 256, BN = 256;
  int tiles_m = 0;
  for (int g = 0; g < p.ngroups; ++g) {
    p.g_tile0[g] = tiles_m;
    tiles_m += cdiv(p.g_rows[g], BM);
  }
  p.g_tile0[p.ngroups] = tiles_m;
  for (int g = p.ngroups + 1; g <= kMaxGroups; ++g) p.g_tile0[g] = tiles_m;
  const long blocks = (long)tiles_m * cdiv(p.N, BN);
  if (blocks <= 0) return MD_OK;
  if (blocks > 0x7fffffffL) MD_FAIL(MD_ERR_UNSUPPORTED, "gemm: too many tiles (%ld)", blocks);
  prep_tile_map(p, tiles_m, cdiv(p.N, BN));
  constexpr int smem = 2 * (BM + BN) * 128;  // 128 KB
  const dim3 grid((unsigned)blocks, (unsigned)(p.batch > 1 ? p.batch : 1));
  auto go = [&](auto kern, bool* attr_set) -> int {
    if (!*attr_set) {
      MD_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
      *attr_set = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, stream, p);
    MD_HIP(hipGetLastError());
    return MD_OK;
  };
  static bool set[16] = {};
  if constexpr (std::is_same<T, bf16_t>::value && AMODE == A_DENSE) {  // timing-only ablations (md_bench_gemm)
    switch (p.debug_flags & 15) {
      case 1: return go(gemm4w_kernel<T, AMODE, 1>, &set[1]);
      case 2: return go(gemm4w_kernel<T, AMODE, 3>, &set[3]);   // no loads, no writes
      case 3: return go(gemm4w_kernel<T, AMODE, 7>, &set[7]);   // + no barrier
      case 4: return go(gemm4w_kernel<T, AMODE, 15>, &set[15]); // + no fragment reads: MFMAs only
      case 5: return go(gemm4w_kernel<T, AMODE, 4>, &set[4]);   // no barrier only
      default: break;
    }
  }
  return go(gemm4w_kernel<T, AMODE, 0>, &set[0]);
}

}  // namespace md
