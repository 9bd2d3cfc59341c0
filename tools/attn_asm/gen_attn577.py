"""Generator of the assembly-owned Depth Pro attention kernel (gfx950): bf16, 577 tokens, head_dim 64.

  One PERSISTENT workgroup per CU walks the (sequence, head) units; four waves, one per SIMD with the whole 512-register file.
  Wave w owns queries 1 + 144 w .. 144 w + 144 as nine 16-query blocks, and a sixteenth of every key tile for query 0 (the class token:
  its partial O, l of the four waves add at the end -- the fast body keeps no running maximum).  S^T = K.Q^T and O^T = V^T.P^T on
  v_mfma_f32_16x16x32_bf16, the row sums on the matrix pipe (an all-ones A operand), O / l in AGPRs, S / P / fragments in VGPRs.
  K and V^T tiles of 64 keys through a 4-stage LDS ring filled by LDS-DMA three tiles ahead (across units), one counted vmcnt +
  s_barrier per tile; software pipeline over (32-key half, query block) steps: S(n) | exp, pack (n-1) | P.V + sum (n-2).
  A wave's 19-KB staging area in LDS holds, slot by slot, the previous unit's output rows (leaving as whole lines, two stores per
  tile) and then the next unit's Q rows (arriving by LDS-DMA, read into the fragment registers in the last tile); the output work of a
  unit rides beside the MFMAs of its last (one-key) tile.

The contract of burn_depth_amd/csrc/kernels/attention.hip holds (q pre-scaled to log2 units, V^T rows padded to kpad keys with
finite values, p = 2^s with no maximum for bf16: a row sum outside [2^-64, 2^100) raises redo[unit] and the HIP kernel's safe body
runs that unit again).
  python tools/attn_asm/gen_attn577.py > burn_depth_amd/csrc/kernels/attn577_gfx950.s     (make -C burn_depth_amd/csrc asm)
  python tools/attn_asm/gen_attn577.py --force <ablation> ...                             timing-only variants (tools/attn_asm/run_co.py)
Checked on the CPU by tools/attn_asm/emu_test.py / tests/test_attn_asm.py (isa.py: wait-state checker + functional emulator)."""
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa import I, R, v, a, s, M0, VCC, EXEC, Pool, check_hazards  # noqa: E402

NB = 9            # 16-query blocks per wave
NTF = 9           # full 64-key tiles (577 = 9 * 64 + 1)
STAGE = 16384     # K tile 8 KB | V^T tile 8 KB
NSTAGE = 4
RING = STAGE * NSTAGE
OSTAGE = RING + 2048     # behind the ring: 2 KB for the class token's partial sums, then per wave its 144 rows of 128 B (16-byte chunks XOR-swizzled by the row)
OWAVE = 144 * 128 + 1024     # + one 1-KiB piece: the sequence's first rows (the class token's query)
LDS_BYTES = OSTAGE + 4 * OWAVE   # the ring | the class token's partial sums | the output staging


class K:
    def __init__(self):
        self.p = []
        self.vp = Pool("v", 0, 256)
        self.sp = Pool("s", 0, 100)

    def e(self, op, dst=(), src=(), **m):
        ins = I(op, dst, src, **m)
        self.p.append(ins)
        return ins

    def label(self, name):
        self.e("label", (), (name,))

    def nop(self, n):
        self.e("s_nop", (), (n,))


def build(abl=()):
    """abl: timing-only ablations (results are wrong): 'loop2' doubles the tile loop, 'noexp' drops the exponentials and packs,
    'nodma' the LDS-DMA requests inside the loop, 'nolds' the fragment reads inside the loop, 'nostore' the output stores,
    'nopv' / 'nos' the P.V (+ sum) / score MFMAs; schedule experiments 'ilv', 'pvfirst', 'plan2', 'plan3' (profiles/r05_attention_asm_stamps.txt:
    the shipped order won)"""
    abl = set(abl)
    k = K()
    e = k.e
    uniq = [0]

    def lab(stem):
        uniq[0] += 1
        return f"L_{stem}_{uniq[0]}"

    def stamp2(slot):
        if "stamps2" in abl:
            stamp(slot, True)

    def stamp(slot, fine=False):
        """timing experiments ('stamps'): wave 0 of workgroup 0 stores s_memtime into dbg[unit_count * 8 + slot] (qwords behind the flags)"""
        if "stamps" not in abl or (("stamps2" in abl) != fine and slot not in (0, 1, 6)):
            return
        l_skip = lab("nostamp")
        e("s_or_b32", s(96), (s(2), s(65)))          # workgroup 0, wave 0
        e("s_cbranch_scc1", (), (l_skip,))
        e("s_cmp_gt_u32", (), (s(97), 3))
        e("s_cbranch_scc1", (), (l_skip,))
        e("s_or_b32", s(96), (s(92), s(93)))        # a null debug pointer: no stamps
        e("s_cbranch_scc0", (), (l_skip,))
        e("s_memtime", s(94, 2), ())
        e("s_waitcnt", lgkmcnt=0)
        e("v_mov_b32", v(6), (s(94),))             # vt2 | vt3: prologue temporaries (an even pair), dead at every stamp point
        e("v_mov_b32", v(7), (s(95),))
        e("s_lshl_b32", s(96), (s(97), 6))
        e("v_mov_b32", v(5), (s(96),))
        e("global_store_dwordx2", (), (v(5), v(6, 2), s(92, 2)), offset=slot * 8)
        k.label(l_skip)

    # ---------------- registers ----------------
    s_karg = s(0, 2)
    s_wg = s(2)
    s_qk, s_vt, s_out, s_redo = s(4, 2), s(6, 2), s(8, 2), s(10, 2)
    s_S, s_n, s_heads, s_D, s_kpad, s_hlog, s_nunits, s_grid = (s(12 + i) for i in range(8))
    s_kd, s_vd, s_od = s(24, 4), s(28, 4), s(32, 4)           # LDS-DMA descriptors (K, V^T); output rows of the current unit
    s_kso, s_vso, s_dst, s_cnt = s(36), s(37), s(38), s(39)   # DMA cursors (K / V^T source offsets of the next tile, LDS stage), loop counter
    t0, t1, t2, t3 = s(40), s(41), s(42), s(43)
    s_mmain, s_mcls = s(44, 2), s(46, 2)
    s_bad = s(48, 2)
    s_kp0, s_kp1, s_vp0, s_vp1 = s(50), s(51), s(52), s(53)   # per-wave piece offsets inside a tile (source side)
    s_mk, s_mv = s(54), s(55)                                # per-wave piece offsets inside a stage (LDS side)
    s_t4, s_t5 = s(56), s(57)
    s_bo = s(58)                                              # block source / destination offset cursor
    s_tmp64 = s(60, 2)
    s_lo, s_hi = s(62), s(63)
    s_dmat = s(64)                                            # index (inside its unit) of the next tile the DMA requests
    s_w, s_rowB, s_kpadB = s(65), s(66), s(67)
    s_unit, s_unit_n, s_last = s(68), s(69), s(70)
    s_qd_n, s_kd_n, s_vd_n, s_od_n = s(72, 4), s(76, 4), s(80, 4), s(84, 4)   # the NEXT unit's descriptors
    s_rd, s_rd_n = s(88, 2), s(90, 2)                         # redo flag address of the current / next unit
    s_od_p = s(20, 4)                                         # the PREVIOUS unit's output rows (its staged blocks leave during this unit's tiles)
    s_qso, s_qm0 = s(59), s(71)                               # the next unit's Q rows: source offset / LDS address of the next 16-row block
    s_tbo = s(3)                                              # their row offset cursor

    vp = k.vp
    v_tid = vp.take()          # v0 = work-item id
    v_lane, v_g, vt0, vt1, vt2, vt3, v_sp = (vp.take() for _ in range(7))   # v0..v7: prologue / epilogue temporaries; inside the tile
    v_trk = [v(0, 4), v(4, 4)]                                               # loop the two quads carry the previous unit's rows to their stores
    v_q = [vp.take(8, 4) for _ in range(NB)]       # Q fragments: [k-step 0 | k-step 1]
    v_qc = vp.take(8, 4)
    v_kf = [vp.take(16, 4) for _ in range(2)]      # K fragments of a 32-key half: (bsel * 2 + kstep) * 4
    v_vf = [vp.take(16, 4) for _ in range(2)]      # V^T fragments of a 32-key half: db * 4
    v_kc = vp.take(8, 4)                           # class-token K fragments (this wave's 16 keys of the tile)
    v_vc = vp.take(16, 4)                          # class-token V^T fragments: db * 4 + {lo, lo, zero, zero}
    v_sb = [vp.take(8, 4) for _ in range(2)]       # scores of a step: kb0 | kb1
    v_pb = [vp.take(4, 4) for _ in range(2)]
    v_sc = vp.take(4, 4)
    v_pc = vp.take(4, 4)
    v_ones = vp.take(4, 4)
    v_koff = [vp.take() for _ in range(2)]         # LDS read cursors (stage included)
    v_voff = [vp.take() for _ in range(2)]
    v_kcoff = [vp.take() for _ in range(2)]
    v_vcoff = vp.take()
    v_dk = [vp.take() for _ in range(2)]           # LDS-DMA source offsets (even / odd piece)
    v_dv = [vp.take() for _ in range(2)]
    v_qrdc, v_r, v_oco = vp.take(), vp.take(), vp.take()
    v_qrd = [vp.take() for _ in range(2)]          # Q fragments in the wave's staging area: row n, chunk (4 s + g) ^ f(n)
    v_zero = vp.take()
    v_lmin, v_lsum = vp.take(), vp.take()          # the unit's row sums: smallest, and their total (an inf or a NaN shows in the total)
    v_scr = vp.take()                              # class-token scratch address of this lane
    v_ost, v_ord, v_oo2 = vp.take(), vp.take(), vp.take()   # output staging: write address, read cursor, row-wise store offset
    v_e = [v_kc[i] for i in range(8)]              # epilogue temporaries: the class token's K fragments are dead there (v_e[6:7]: even pair)
    a_o = [[a((b * 4 + db) * 4, 4) for db in range(4)] for b in range(NB)]
    a_oc = [a(NB * 16 + db * 4, 4) for db in range(4)]
    a_l = [a(NB * 16 + 16 + b * 4, 4) for b in range(NB)]
    a_lc = a(NB * 16 + 16 + NB * 4, 4)
    n_acc = NB * 16 + 16 + NB * 4 + 4

    # ---------------- prologue ----------------
    e("s_load_dwordx8", s(4, 8), (s_karg, 0x0))
    e("s_load_dwordx8", s(12, 8), (s_karg, 0x20))
    e("v_and_b32", v_lane, (63, v_tid))
    e("v_lshrrev_b32", vt0, (6, v_tid))
    e("v_and_b32", v_r, (15, v_lane))
    e("v_lshrrev_b32", v_g, (4, v_lane))
    e("v_mov_b32", v_zero, (0,))
    e("v_readfirstlane_b32", s_w, (vt0,))
    e("s_waitcnt", lgkmcnt=0)
    k.nop(3)
    if "stamps" in abl:
        e("s_load_dwordx2", s(92, 2), (s_karg, 0x40))
        e("s_mov_b32", s(97), (0,))
        e("s_waitcnt", lgkmcnt=0)
        stamp(0)
    e("s_lshl_b32", s_rowB, (s_D, 2))
    e("s_lshl_b32", s_kpadB, (s_kpad, 1))

    def unit_descriptors():
        """the descriptors of unit s_unit_n = (sequence, head) into the `next` set"""
        seq, head = s_t4, s_t5
        e("s_lshr_b32", seq, (s_unit_n, s_hlog))
        e("s_sub_u32", t0, (s_heads, 1))
        e("s_and_b32", head, (s_unit_n, t0))
        # q | k rows: qk + (seq * S) * rowB + head * 128
        e("s_mul_i32", t0, (seq, s_S))
        e("s_mul_hi_u32", t1, (t0, s_rowB))
        e("s_mul_i32", t0, (t0, s_rowB))
        e("s_lshl_b32", t2, (head, 7))
        e("s_add_u32", t0, (t0, t2))
        e("s_addc_u32", t1, (t1, 0))
        e("s_add_u32", s_qd_n[0], (s_qk[0], t0))
        e("s_addc_u32", s_qd_n[1], (s_qk[1], t1))
        e("s_lshl_b32", t2, (s_D, 1))
        e("s_add_u32", s_kd_n[0], (s_qd_n[0], t2))
        e("s_addc_u32", s_kd_n[1], (s_qd_n[1], 0))
        e("s_and_b32", s_qd_n[1], (s_qd_n[1], 0xFFFF))
        e("s_and_b32", s_kd_n[1], (s_kd_n[1], 0xFFFF))
        e("s_mul_i32", s_qd_n[2], (s_n, s_rowB))
        e("s_sub_u32", t2, (s_n, 1))
        e("s_mul_i32", s_kd_n[2], (t2, s_rowB))
        e("s_add_u32", s_kd_n[2], (s_kd_n[2], 128))
        e("s_mov_b32", s_qd_n[3], (0x00020000,))
        e("s_mov_b32", s_kd_n[3], (0x00020000,))
        # V^T: vT + unit * 64 * kpadB
        e("s_lshl_b32", t2, (s_kpadB, 6))
        e("s_mul_hi_u32", t1, (s_unit_n, t2))
        e("s_mul_i32", t3, (s_unit_n, t2))
        e("s_add_u32", s_vd_n[0], (s_vt[0], t3))
        e("s_addc_u32", s_vd_n[1], (s_vt[1], t1))
        e("s_and_b32", s_vd_n[1], (s_vd_n[1], 0xFFFF))
        e("s_mov_b32", s_vd_n[2], (t2,))
        e("s_mov_b32", s_vd_n[3], (0x00020000,))
        # redo flag
        e("s_lshl_b32", t3, (s_unit_n, 2))
        e("s_add_u32", s_rd_n[0], (s_redo[0], t3))
        e("s_addc_u32", s_rd_n[1], (s_redo[1], 0))
        # output rows: out + (seq * S) * (2 D) + head * 128
        e("s_mul_i32", t0, (seq, s_S))
        e("s_lshl_b32", t2, (s_D, 1))
        e("s_mul_hi_u32", t1, (t0, t2))
        e("s_mul_i32", t0, (t0, t2))
        e("s_lshl_b32", t3, (head, 7))
        e("s_add_u32", t0, (t0, t3))
        e("s_addc_u32", t1, (t1, 0))
        e("s_add_u32", s_od_n[0], (s_out[0], t0))
        e("s_addc_u32", s_od_n[1], (s_out[1], t1))
        e("s_and_b32", s_od_n[1], (s_od_n[1], 0xFFFF))
        e("s_mul_i32", s_od_n[2], (s_n, t2))
        e("s_mov_b32", s_od_n[3], (0x00020000,))

    e("s_mov_b32", s_unit_n, (s_wg,))
    unit_descriptors()
    for j in range(0, 4, 2):
        e("s_mov_b64", s_kd[j:j + 2], (s_kd_n[j:j + 2],))
        e("s_mov_b64", s_vd[j:j + 2], (s_vd_n[j:j + 2],))
    # per-wave piece offsets
    e("s_lshl_b32", t0, (s_w, 4))             # 16 w = first K / V^T row of this wave's two pieces
    e("s_mul_i32", s_kp0, (t0, s_rowB))
    e("s_lshl_b32", t1, (s_rowB, 3))
    e("s_add_u32", s_kp1, (s_kp0, t1))
    e("s_mul_i32", s_vp0, (t0, s_kpadB))
    e("s_lshl_b32", t1, (s_kpadB, 3))
    e("s_add_u32", s_vp1, (s_vp0, t1))
    e("s_lshl_b32", s_mk, (s_w, 11))          # 2 w * 1024
    e("s_add_u32", s_mv, (s_mk, 8192))
    # LDS-DMA source offsets of a lane: row8 = lane >> 3, pc = lane & 7
    e("v_lshrrev_b32", vt0, (3, v_lane))      # row8
    e("v_and_b32", vt1, (7, v_lane))          # pc
    e("v_lshrrev_b32", vt2, (1, vt0))         # row8 >> 1
    e("v_and_b32", vt3, (1, vt2))
    e("v_lshlrev_b32", vt3, (1, vt3))         # fK (even piece) = R1 << 1
    e("v_xor_b32", vt3, (vt3, vt1))
    e("v_lshlrev_b32", vt3, (4, vt3))
    e("v_mul_lo_u32", v_dk[0], (vt0, s_rowB))
    e("v_add_u32", v_dk[0], (v_dk[0], vt3))
    e("v_xor_b32", v_dk[1], (64, v_dk[0]))
    e("v_xor_b32", vt3, (vt2, vt1))           # pc ^ (row8 >> 1): fV of an even piece
    e("v_lshlrev_b32", vt3, (4, vt3))
    e("v_mul_lo_u32", v_dv[0], (vt0, s_kpadB))
    e("v_add_u32", v_dv[0], (v_dv[0], vt3))
    e("v_xor_b32", v_dv[1], (64, v_dv[0]))

    in_loop = [False]

    def dma_piece(which):
        """one 1-KiB piece of the tile at the DMA cursors: which = 0 / 1 (K even / odd), 2 / 3 (V^T even / odd)"""
        if in_loop[0] and "nodma" in abl:
            return
        if which < 2:
            e("s_add_u32", M0, (s_dst, s_mk))
            if which == 1:
                e("s_add_u32", M0, (M0, 1024))
            e("s_add_u32", s_t4, (s_kso, s_kp0 if which == 0 else s_kp1))
            e("buffer_load_dwordx4", (), (v_dk[which], s_kd, s_t4), lds=True)
        else:
            e("s_add_u32", M0, (s_dst, s_mv))
            if which == 3:
                e("s_add_u32", M0, (M0, 1024))
            e("s_add_u32", s_t5, (s_vso, s_vp0 if which == 2 else s_vp1))
            e("buffer_load_dwordx4", (), (v_dv[which - 2], s_vd, s_t5), lds=True)

    def dma_advance():
        """behind a tile's four pieces: the next tile of the unit, or -- behind its last tile -- the first tile of the NEXT unit"""
        l_sw, l_done = lab("dma_switch"), lab("dma_next")
        e("s_add_u32", s_dmat, (s_dmat, 1))
        e("s_cmp_eq_u32", (), (s_dmat, NTF + 1))
        e("s_cbranch_scc1", (), (l_sw,))
        e("s_lshl_b32", t0, (s_rowB, 6))
        e("s_add_u32", s_kso, (s_kso, t0))
        e("s_add_u32", s_vso, (s_vso, 128))
        e("s_branch", (), (l_done,))
        k.label(l_sw)
        e("s_mov_b32", s_dmat, (0,))
        e("s_mov_b32", s_kso, (0,))
        e("s_mov_b32", s_vso, (0,))
        for j in range(0, 4, 2):
            e("s_mov_b64", s_kd[j:j + 2], (s_kd_n[j:j + 2],))
            e("s_mov_b64", s_vd[j:j + 2], (s_vd_n[j:j + 2],))
        k.label(l_done)
        e("s_add_u32", s_dst, (s_dst, STAGE))
        e("s_and_b32", s_dst, (s_dst, RING - 1))

    def q_reset():
        """Q cursors at the NEXT unit's first block: rows 1 + 144 w ..., the wave's staging area"""
        e("s_mul_i32", s_qso, (s_w, 144))
        e("s_add_u32", s_qso, (s_qso, 1))
        e("s_mul_i32", s_qso, (s_qso, s_rowB))
        e("s_mul_i32", s_qm0, (s_w, OWAVE))
        e("s_add_u32", s_qm0, (s_qm0, OSTAGE))

    def q_piece(pc_):
        """one 8-row piece of the next unit's Q rows into the 2-KB slot of the staging area whose output block has just left (LDS-DMA: whole
        lines; 20 per-lane loads of the MFMA fragments touched 64 separate 16-byte segments each and cost ~120 cycles apiece)"""
        e("s_add_u32", M0, (s_qm0, 1024 * pc_))
        if pc_:
            e("s_lshl_b32", s_t4, (s_rowB, 3))
            e("s_add_u32", s_t4, (s_t4, s_qso))
        else:
            k.nop(0)
        e("buffer_load_dwordx4", (), (v_dk[pc_], s_qd_n, s_t4 if pc_ else s_qso), lds=True)
        if pc_:
            e("s_lshl_b32", s_t4, (s_rowB, 4))
            e("s_add_u32", s_qso, (s_qso, s_t4))
            e("s_add_u32", s_qm0, (s_qm0, 2048))

    def q_cls_piece():
        e("s_mul_i32", s_t4, (s_w, OWAVE))
        e("s_add_u32", M0, (s_t4, OSTAGE + 144 * 128))
        k.nop(0)
        e("buffer_load_dwordx4", (), (v_dk[0], s_qd_n, 0), lds=True)

    def q_reads():
        """the staged Q rows into the MFMA fragments (B operand: lane (n, g) <- Q[row n][32 s + 8 g ..])"""
        for b in range(NB):
            for st in range(2):
                e("ds_read_b128", v_q[b][4 * st:4 * st + 4], (v_qrd[st],), offset=b * 2048)
        for st in range(2):
            e("ds_read_b128", v_qc[4 * st:4 * st + 4], (v_qrdc,), offset=64 * st)

    e("s_mov_b32", s_kso, (0,))
    e("s_mov_b32", s_vso, (0,))
    e("s_mov_b32", s_dst, (0,))
    e("s_mov_b32", s_dmat, (0,))
    for _ in range(3):
        for wpc in range(4):
            dma_piece(wpc)
        dma_advance()
    q_reset()
    for _ in range(NB):
        q_piece(0)
        q_piece(1)
    q_cls_piece()
    e("v_lshlrev_b32", v_oco, (3, v_g))
    # Q fragment read addresses: staging + n * 128 + (((4 s + g) ^ f(n)) << 4), f(n) = ((n >> 1) & 1) << 1 | ((n >> 3) & 1) << 2 (the K image's swizzle)
    e("s_mul_i32", t0, (s_w, OWAVE))
    e("s_add_u32", t0, (t0, OSTAGE))
    e("v_and_b32", vt1, (2, v_r))
    e("v_lshrrev_b32", vt2, (3, v_r))
    e("v_lshlrev_b32", vt2, (2, vt2))
    e("v_or_b32", vt1, (vt1, vt2))
    e("v_xor_b32", vt1, (vt1, v_g))
    e("v_lshlrev_b32", vt1, (4, vt1))
    e("v_lshlrev_b32", vt0, (7, v_r))
    e("v_add_u32", v_qrd[0], (vt0, vt1))
    e("v_add_u32", v_qrd[0], (t0, v_qrd[0]))
    e("v_xor_b32", v_qrd[1], (64, v_qrd[0]))
    e("v_lshlrev_b32", v_qrdc, (4, v_g))
    e("s_add_u32", t0, (t0, 144 * 128))
    e("v_add_u32", v_qrdc, (t0, v_qrdc))
    # constants
    for j in range(4):
        e("v_mov_b32", v_ones[j], (0x3F803F80,))
    for db in range(4):
        e("v_mov_b32", v_vc[db * 4 + 2], (0,))
        e("v_mov_b32", v_vc[db * 4 + 3], (0,))
    e("v_mov_b32", v_pc[2], (0,))
    e("v_mov_b32", v_pc[3], (0,))
    e("v_cmp_eq_u32", s_mmain, (v_g, 0))
    e("s_mov_b64", s_mcls, (0,))
    e("s_cmp_eq_u32", (), (s_w, 0))
    e("s_cbranch_scc0", (), ("L_not_w0",))
    e("s_mov_b64", s_mcls, (s_mmain,))
    k.label("L_not_w0")
    e("s_mov_b32", s_lo, (0x1f800000,))     # 2^-64
    e("s_mov_b32", s_hi, (0x71800000,))     # 2^100
    # class-token scratch: [wave][17][4 lane groups] floats behind the ring; a lane with n == 0 owns slot g
    e("s_mul_i32", t0, (s_w, 17 * 16))
    e("v_lshlrev_b32", v_scr, (2, v_g))
    e("v_add_u32", v_scr, (t0, v_scr))
    e("v_add_u32", v_scr, (RING, v_scr))
    # output staging (128-byte rows, 16-byte chunks XOR-swizzled by row & 7): a lane (n, g) writes its 4 values of d-block db at row n,
    # logical chunk 2 db + (g >> 1), byte (g & 1) * 8 -- address = v_ost ^ (32 db) + block * 2048; a lane l reads physical chunk l & 7 of
    # row l >> 3 (+ 8) = logical chunk (l & 7) ^ ((l >> 3) & 7), which is where it stores it in the output row
    e("s_mul_i32", t0, (s_w, OWAVE))
    e("s_add_u32", t0, (t0, OSTAGE))
    e("v_lshrrev_b32", vt0, (1, v_g))
    e("v_and_b32", vt1, (7, v_r))
    e("v_xor_b32", vt0, (vt0, vt1))               # X = (g >> 1) ^ (n & 7)
    e("v_lshlrev_b32", vt0, (4, vt0))
    e("v_and_b32", vt1, (1, v_g))
    e("v_lshlrev_b32", vt1, (3, vt1))
    e("v_add_u32", vt0, (vt0, vt1))
    e("v_lshlrev_b32", v_ost, (7, v_r))
    e("v_add_u32", v_ost, (v_ost, vt0))
    e("v_add_u32", v_ost, (t0, v_ost))
    e("v_lshrrev_b32", vt0, (3, v_lane))          # row of the read-back
    e("v_and_b32", vt1, (7, v_lane))              # physical chunk
    e("v_lshlrev_b32", v_ord, (7, vt0))
    e("v_lshlrev_b32", vt2, (4, vt1))
    e("v_add_u32", v_ord, (v_ord, vt2))
    e("v_add_u32", v_ord, (t0, v_ord))
    e("v_xor_b32", vt1, (vt1, vt0))               # logical chunk (rows below 8: row & 7 = row)
    e("v_lshlrev_b32", vt1, (4, vt1))
    e("s_mul_i32", t0, (s_w, 144))
    e("s_add_u32", t0, (t0, 1))
    e("v_add_u32", vt0, (t0, vt0))                # query row 1 + 144 w + (l >> 3)
    e("s_lshl_b32", t2, (s_D, 1))
    e("v_mul_lo_u32", v_oo2, (vt0, t2))
    e("v_add_u32", v_oo2, (v_oo2, vt1))
    for j in range(4):
        e("s_mov_b32", s_od_p[j], (0,))           # no previous unit yet: a descriptor of zero records drops the stores
        e("s_mov_b32", s_od[j], (0,))
    e("s_mov_b32", s_tbo, (0,))
    # LDS read cursors
    #  K:  (8 (r >> 2) + (r & 3)) * 128 + (((4 s + g) ^ f) << 4), f = ((r >> 1) & 1) << 1 | ((r >> 2) & 1) << 2
    e("v_lshrrev_b32", vt0, (2, v_r))
    e("v_lshlrev_b32", vt0, (3, vt0))
    e("v_and_b32", vt1, (3, v_r))
    e("v_or_b32", vt0, (vt0, vt1))
    e("v_lshlrev_b32", vt0, (7, vt0))         # row * 128
    e("v_and_b32", vt1, (6, v_r))              # bits 1, 2 of r in place = f
    e("v_xor_b32", vt2, (vt1, v_g))           # (0 + g) ^ f
    e("v_lshlrev_b32", vt2, (4, vt2))
    e("v_add_u32", v_koff[0], (vt0, vt2))
    e("v_xor_b32", v_koff[1], (64, v_koff[0]))
    #  V^T: r * 128 + (((4 half + g) ^ ((r >> 1) & 7)) << 4)
    e("v_lshlrev_b32", vt0, (7, v_r))
    e("v_lshrrev_b32", vt1, (1, v_r))
    e("v_xor_b32", vt2, (vt1, v_g))
    e("v_lshlrev_b32", vt2, (4, vt2))
    e("v_add_u32", v_voff[0], (vt0, vt2))
    e("v_xor_b32", v_voff[1], (64, v_voff[0]))
    #  class-token K: (16 w + r) * 128 + (((4 s + g) ^ fc) << 4), fc = ((r >> 1) & 1) << 1 | ((r >> 3) & 1) << 2
    e("s_lshl_b32", t0, (s_w, 4))
    e("v_add_u32", vt0, (t0, v_r))
    e("v_lshlrev_b32", vt0, (7, vt0))
    e("v_and_b32", vt1, (2, v_r))
    e("v_lshrrev_b32", vt2, (3, v_r))
    e("v_lshlrev_b32", vt2, (2, vt2))
    e("v_or_b32", vt1, (vt1, vt2))
    e("v_xor_b32", vt1, (vt1, v_g))
    e("v_lshlrev_b32", vt1, (4, vt1))
    e("v_add_u32", v_kcoff[0], (vt0, vt1))
    e("v_xor_b32", v_kcoff[1], (64, v_kcoff[0]))
    #  class-token V^T: r * 128 + (((2 w + (g >> 1)) ^ (r >> 1)) << 4) + (g & 1) * 8
    e("v_lshlrev_b32", vt0, (7, v_r))
    e("v_lshrrev_b32", vt1, (1, v_g))
    e("s_lshl_b32", t0, (s_w, 1))
    e("v_add_u32", vt1, (t0, vt1))
    e("v_lshrrev_b32", vt2, (1, v_r))
    e("v_xor_b32", vt1, (vt1, vt2))
    e("v_lshlrev_b32", vt1, (4, vt1))
    e("v_and_b32", vt2, (1, v_g))
    e("v_lshlrev_b32", vt2, (3, vt2))
    e("v_add_u32", v_vcoff, (vt0, vt1))
    e("v_add_u32", v_vcoff, (v_vcoff, vt2))

    # ---------------- pipeline pieces ----------------
    def bump(regs):
        for r_ in regs:
            e("v_add_u32", r_, (STAGE, r_))
        for r_ in regs:
            e("v_and_b32", r_, (RING - 1, r_))

    def k_read(hb, j, half_imm):
        if in_loop[0] and "nolds" in abl:
            return
        bsel, st = j >> 1, j & 1
        e("ds_read_b128", v_kf[hb][4 * j:4 * j + 4], (v_koff[st],), offset=bsel * 512 + half_imm * 4096)

    def v_read(hb, db, half):
        if in_loop[0] and "nolds" in abl:
            return
        e("ds_read_b128", v_vf[hb][4 * db:4 * db + 4], (v_voff[half],), offset=8192 + db * 2048)

    def cls_reads_k():
        for st in range(2):
            e("ds_read_b128", v_kc[4 * st:4 * st + 4], (v_kcoff[st],))

    def cls_reads_v():
        for db in range(4):
            e("ds_read_b64", v_vc[4 * db:4 * db + 2], (v_vcoff,), offset=8192 + db * 2048)

    def mf(dst, A, B, C):
        return I("v_mfma_f32_16x16x32_bf16", (dst,), (A, B, C))

    def s_mfmas(n):
        b, half = n % NB, (n // NB) & 1
        sb, kf = v_sb[n & 1], v_kf[half]
        return [mf(sb[0:4], kf[0:4], v_q[b][0:4], 0), mf(sb[4:8], kf[8:12], v_q[b][0:4], 0),
                mf(sb[0:4], kf[4:8], v_q[b][4:8], sb[0:4]), mf(sb[4:8], kf[12:16], v_q[b][4:8], sb[4:8])]

    def e_valu(n):
        sb, pb = v_sb[n & 1], v_pb[n & 1]
        ex = [I("v_exp_f32", (sb[j],), (sb[j],)) for j in range(8)]
        cv = [I("v_cvt_pk_bf16_f32", (pb[j],), (sb[2 * j], sb[2 * j + 1])) for j in range(4)]
        return ex, cv

    def pv_mfmas(n, fresh=False):
        """fresh: the unit's first products into these accumulators (C = 0 instead of the previous unit's sums)"""
        b, half = n % NB, (n // NB) & 1
        pb, vf = v_pb[n & 1], v_vf[half]
        return [mf(a_o[b][db], vf[4 * db:4 * db + 4], pb, 0 if fresh else a_o[b][db]) for db in range(4)] + \
               [mf(a_l[b], v_ones, pb, 0 if fresh else a_l[b])]

    def emit_slot(n, do_s, do_e, do_pv, extras_head=(), extras=(), fresh=False):
        """MFMAs of S(n) then P.V(n-2); the exponentials and packs of step n-1 between them; `extras` one behind each of the first MFMAs"""
        for x in extras_head:
            x()
        sm_ = s_mfmas(n) if do_s and "nos" not in abl else []
        pm_ = pv_mfmas(n - 2, fresh) if do_pv and "nopv" not in abl else []
        ms = sm_ + pm_
        if "ilv" in abl and len(sm_) == 4 and len(pm_) == 5:      # experiment: S and P.V MFMAs alternate (more distance inside the S chains)
            ms = [sm_[0], pm_[0], sm_[1], pm_[1], sm_[2], pm_[2], sm_[3], pm_[3], pm_[4]]
        if "pvfirst" in abl and len(sm_) == 4 and len(pm_) == 5:  # experiment: P.V in front of S
            ms = pm_ + sm_
        if "noexp" in abl:
            do_e = False
        ex, cv = e_valu(n - 1) if do_e else ([], [])
        order = ex[:6] + [ex[6], cv[0], ex[7], cv[1], cv[2], cv[3]] if do_e else []
        gaps = max(len(ms), 1)
        per = [[] for _ in range(gaps)]
        if do_e:
            # one exponential behind each of the first six MFMAs, then (exp, pack) pairs, the last two packs behind the last MFMA
            plan = [[0], [1], [2], [3], [4], [5], [6, 7], [8, 9], [10, 11]]
            if "plan2" in abl:      # experiment: two exponentials behind each of the first four MFMAs, the packs behind the rest
                order = ex + cv
                plan = [[0, 1], [2, 3], [4, 5], [6, 7], [8], [9], [10], [11], []]
            if "plan3" in abl:      # experiment: packs as early as their exponentials allow
                order = [ex[0], ex[1], ex[2], cv[0], ex[3], ex[4], cv[1], ex[5], ex[6], cv[2], ex[7], cv[3]]
                plan = [[0], [1], [2, 3], [4], [5, 6], [7], [8, 9], [10], [11]]
            if len(ms) == 9:
                for gi, idxs in enumerate(plan):
                    per[gi] = [order[j] for j in idxs]
            else:
                q = 0
                for gi in range(gaps):
                    take = (len(order) - q + (gaps - gi) - 1) // (gaps - gi)
                    per[gi] = order[q:q + take]
                    q += take
        ext = list(extras)
        if not ms:
            for x in ext:
                x()
            for ins in per[0]:
                k.p.append(ins)
            return
        for gi, m_ in enumerate(ms):
            k.p.append(m_)
            if ext:
                ext.pop(0)()
            for ins in per[gi]:
                k.p.append(ins)
        for x in ext:
            x()

    def cls_s():
        k.p.append(mf(v_sc, v_kc[0:4], v_qc[0:4], 0))
        k.p.append(mf(v_sc, v_kc[4:8], v_qc[4:8], v_sc))

    def cls_e():
        for j in range(4):
            e("v_exp_f32", v_sc[j], (v_sc[j],))
        k.nop(0)
        e("v_cvt_pk_bf16_f32", v_pc[0], (v_sc[0], v_sc[1]))
        e("v_cvt_pk_bf16_f32", v_pc[1], (v_sc[2], v_sc[3]))

    def cls_pv(fresh=False):
        for db in range(4):
            k.p.append(mf(a_oc[db], v_vc[4 * db:4 * db + 4], v_pc, 0 if fresh else a_oc[db]))
        k.p.append(mf(a_lc, v_ones, v_pc, 0 if fresh else a_lc))

    def tile(first):
        for n in range(18):
            head, ext = [], []
            if n == 0:
                head.append(lambda: e("s_waitcnt", lgkmcnt=0))
            if first and n == 1:
                head.append(lambda: k.nop(7))   # E(0) eight states behind S(0): slot 0 of the first tile carries four MFMAs only
            if n == 1:
                ext.append(cls_reads_k)
            if n == 2:
                ext.append(cls_reads_v)
            if 2 <= n <= 5:
                ext.append(lambda j=n - 2: k_read(1, j, 1))
            if 3 <= n <= 6:
                ext.append(lambda db=n - 3: v_read(1, db, 1))
            if n == 9:
                # every fragment of this tile is in registers (read in slots 1-6, waited for here): behind the barrier its stage takes
                # tile t + 3, and tile t + 1 -- requested two tiles ago -- is visible to every wave
                head.append(lambda: e("s_waitcnt", lgkmcnt=0))
                if first:
                    head.append(lambda: stamp2(2))
                # tile t + 1's pieces must have landed. Younger than them, per tile: two trickled stores, two Q pieces, the next tile's four
                # pieces -- twelve in the steady state; at the tiles around a unit's start fewer (tile 0: stores and Q pieces of the previous
                # unit's tile 8, the hand-over's four pieces, the class token's store = 9; tile 1: 10): nine is right everywhere
                head.append(lambda: e("s_waitcnt", vmcnt=9))
                head.append(lambda: e("s_barrier"))
                if first:
                    head.append(lambda: stamp2(3))
            if 9 <= n <= 12:
                ext.append(lambda wpc=n - 9: dma_piece(wpc))
            if n == 10:
                head.append(cls_s)
                head.append(lambda: bump(v_koff))
            if n == 11:
                head.append(lambda: bump(v_voff))
            if 11 <= n <= 14:
                ext.append(lambda j=n - 11: k_read(0, j, 0))
            if 12 <= n <= 15:
                ext.append(lambda db=n - 12: v_read(0, db, 0))
            if n == 13:
                head.append(cls_e)
            if n == 15:
                head.append(lambda: cls_pv(first))
            if n in (7, 8):     # one staged block (16 rows) of the PREVIOUS unit per tile: read back here, stored behind this tile's requests
                ext.append(lambda h=n - 7: e("ds_read_b128", v_trk[h], (v_ord,), offset=1024 * h))
            if n in (13, 14) and "nostore" not in abl:
                def trickle_store(h=n - 13):
                    if h:
                        e("s_lshl_b32", s_t5, (s_D, 4))        # 8 rows further
                        e("s_add_u32", s_t5, (s_t5, s_tbo))
                    e("buffer_store_dwordx4", (), (v_trk[h], v_oo2, s_od_p, s_t5 if h else s_tbo))
                ext.append(trickle_store)
            if n in (15, 16):   # ... and the slot of the staging area it came from takes 16 rows of the NEXT unit's Q
                ext.append(lambda h=n - 15: q_piece(h))
            if first and n == 17:
                ext.append(q_cls_piece)
            if n == 15:
                def trickle_advance():
                    e("v_add_u32", v_ord, (2048, v_ord))
                    e("s_lshl_b32", t0, (s_D, 5))              # 16 rows
                    e("s_add_u32", s_tbo, (s_tbo, t0))
                head.append(trickle_advance)
            if n == 16:
                head.append(lambda: bump(v_kcoff + [v_vcoff]))
                head.append(dma_advance)
            emit_slot(n, True, not (first and n < 1), not (first and n < 2), head, ext, fresh=first and n - 2 < NB)

    # ---------------- the first unit: wait for its first tiles and Q, read the first fragments ----------------
    e("s_waitcnt", vmcnt=0)
    e("s_barrier")
    for j in range(4):
        k_read(0, j, 0)
    for db in range(4):
        v_read(0, db, 0)
    q_reads()
    k.label("L_unit")
    for j in range(0, 4, 2):
        e("s_mov_b64", s_od_p[j:j + 2], (s_od[j:j + 2],))     # the unit just finished: its rows wait in the staging area
        e("s_mov_b64", s_od[j:j + 2], (s_od_n[j:j + 2],))
    e("s_mov_b32", s_tbo, (0,))
    e("s_mov_b64", s_rd, (s_rd_n,))
    e("s_mov_b32", s_unit, (s_unit_n,))
    # the next unit of this workgroup (the last one names itself: its requests re-read tiles nobody uses)
    e("s_add_u32", s_unit_n, (s_unit, s_grid))
    e("s_mov_b32", s_last, (0,))
    e("s_cmp_lt_u32", (), (s_unit_n, s_nunits))
    e("s_cbranch_scc1", (), ("L_has_next",))
    e("s_mov_b32", s_unit_n, (s_unit,))
    e("s_mov_b32", s_last, (1,))
    k.label("L_has_next")
    unit_descriptors()
    e("s_mov_b64", s_bad, (0,))
    q_reset()
    stamp(1)
    tile(True)
    stamp(2)
    stamp2(4)
    e("s_mov_b32", s_cnt, ((NTF - 1) * (2 if "loop2" in abl else 1),))
    k.label("L_tile")
    in_loop[0] = True
    tile(False)
    in_loop[0] = False
    e("s_sub_u32", s_cnt, (s_cnt, 1))
    e("s_cmp_lg_u32", (), (s_cnt, 0))
    e("s_cbranch_scc1", (), ("L_tile",))
    stamp(3)
    # drain: exp / pack of step 17, P.V of steps 16 and 17
    emit_slot(18, False, True, True)
    k.nop(1)
    emit_slot(19, False, False, True)
    # ---------------- last tile: one key (tile-relative key 0 = row 0 of kb0: lanes with g == 0, register 0) ----------------
    # Its MFMAs (9 x 7) leave the vector issue mostly idle, so the unit's output work rides beside them: block b is final once P.V(b) of
    # this tile has been issued, and is normalised, packed and written to the staging area three slots later. The staging slot of block b
    # holds the NEXT unit's Q rows until slot b reads them into the fragment registers (S(b) was their last reader).
    e("s_waitcnt", lgkmcnt=0)
    cls_reads_k()
    cls_reads_v()
    k.nop(7)
    for hb in range(2):
        for j in range(1, 4):
            e("v_mov_b32", v_pb[hb][j], (0,))
    e("v_mov_b32", v_pc[1], (0,))
    e("s_waitcnt", lgkmcnt=0)
    cls_s()                     # the class token's scores first: its K fragments' registers are the output work's temporaries

    def tail_s(b):
        sb = v_sb[b & 1]
        return [mf(sb[0:4], v_kf[0][0:4], v_q[b][0:4], 0), mf(sb[0:4], v_kf[0][4:8], v_q[b][4:8], sb[0:4])]

    def tail_e(b, pad):
        sb, pb = v_sb[b & 1], v_pb[b & 1]
        return ([I("s_nop", (), (7,))] if pad else []) + \
               [I("v_exp_f32", (sb[0],), (sb[0],)), I("s_nop", (), (1,)), I("v_cndmask_b32", (sb[0],), (0, sb[0], s_mmain)),
                I("v_cvt_pk_bf16_f32", (pb[0],), (sb[0], 0)), I("s_nop", (), (1,))]

    def tail_pv(b):
        pb = v_pb[b & 1]
        return [mf(a_o[b][db], v_vf[0][4 * db:4 * db + 4], pb, a_o[b][db]) for db in range(4)] + [mf(a_l[b], v_ones, pb, a_l[b])]

    def out_block(b):
        """normalise, pack and stage block b (its rows leave as whole lines during the next unit's tiles)"""
        # the range check wants every row sum inside [2^-64, 2^100): the smallest of the wave's sums and their total (which an inf or a NaN
        # cannot leave finite) are compared once per unit -- a compare + scalar OR per block stalled on the VALU -> SALU hand-over each time
        o = [I("v_accvgpr_read_b32", (v_e[4],), (a_l[b][0],))]
        if b == 0:
            o += [I("v_mov_b32", (v_lmin,), (v_e[4],)), I("v_mov_b32", (v_lsum,), (v_e[4],))]
        else:
            o += [I("v_min_f32", (v_lmin,), (v_lmin, v_e[4])), I("v_add_f32", (v_lsum,), (v_lsum, v_e[4]))]
        o.append(I("v_rcp_f32", (v_e[4],), (v_e[4],)))
        pk = R("v", v_e[6].i, 2)
        inv2 = R("v", v_e[4].i, 2)       # 1 / l twice: the second operand of the packed multiplies
        for db in range(4):
            o += [I("v_accvgpr_read_b32", (v_e[i],), (a_o[b][db][i],)) for i in range(4)]
            if db == 0:
                o.append(I("v_mov_b32", (v_e[5],), (v_e[4],)))
            else:
                o.append(I("v_xor_b32", (v_sp,), (32 * db, v_ost)))
            o += [I("v_pk_mul_f32", (R("v", v_e[0].i, 2),), (R("v", v_e[0].i, 2), inv2)),
                  I("v_pk_mul_f32", (R("v", v_e[2].i, 2),), (R("v", v_e[2].i, 2), inv2))]
            o += [I("v_cvt_pk_bf16_f32", (pk[0],), (v_e[0], v_e[1])), I("v_cvt_pk_bf16_f32", (pk[1],), (v_e[2], v_e[3])),
                  I("ds_write_b64", (), (v_sp if db else v_ost, pk), offset=b * 2048)]
        return o

    for n in range(NB + 3):
        sm = tail_s(n) if n < NB else []
        pm = tail_pv(n - 2) if 0 <= n - 2 < NB else []
        va = tail_e(n - 1, n <= 2) if 0 <= n - 1 < NB else []
        if n >= NB + 1:     # the last slots carry few or no MFMAs: the sums of the blocks they finish were written a handful of instructions ago
            va.append(I("s_nop", (), (3,)))
        va += out_block(n - 3) if 0 <= n - 3 < NB else []
        for m_ in sm:
            k.p.append(m_)
        if n < NB:      # S(n) was the last reader of this block's Q fragments: the next unit's come in from the staging slot
            # (its two pieces were requested at tile n; every later tile issued eight vector-memory operations -- four K / V^T pieces, two
            # stores, two Q pieces: that many may still be in flight. The last block's wait is a full drain, a whole tail behind its request;
            # the hand-over's barrier then sees the next unit's tile 0, requested at tile 7, landed on every wave)
            e("s_waitcnt", vmcnt=min(63, 8 * (NB - 1 - n)))
            for st in range(2):
                e("ds_read_b128", v_q[n][4 * st:4 * st + 4], (v_qrd[st],), offset=n * 2048)
        per = (len(va) + len(pm) - 1) // len(pm) if pm else len(va)
        q_ = 0
        for m_ in pm:
            k.p.append(m_)
            for ins in va[q_:q_ + per]:
                k.p.append(ins)
            q_ += per
        for ins in va[q_:]:
            k.p.append(ins)
    k.nop(7)
    e("v_exp_f32", v_sc[0], (v_sc[0],))
    k.nop(1)
    e("v_cndmask_b32", v_sc[0], (0, v_sc[0], s_mcls))
    k.nop(1)
    e("v_cvt_pk_bf16_f32", v_pc[0], (v_sc[0], 0))
    k.nop(1)
    cls_pv()
    for st in range(2):
        e("ds_read_b128", v_qc[4 * st:4 * st + 4], (v_qrdc,), offset=64 * st)
    stamp(4)
    # ---------------- hand-over to the next unit: its first tile (requested at tile 7) is visible behind this barrier ----------------
    e("s_barrier")
    for wpc in range(4):
        dma_piece(wpc)          # the next unit's tile 2 into the stage of this unit's tile 8
    dma_advance()
    bump(v_koff + v_voff)
    bump(v_kcoff + [v_vcoff])
    for j in range(4):
        k_read(0, j, 0)
    for db in range(4):
        v_read(0, db, 0)
    stamp(5)
    e("v_mov_b32", v_pc[1], (0,)) if False else None
    # ---------------- epilogue ----------------
    k.nop(7)
    # class-token partials: lanes with n == 0 write O (16 values) and l to scratch[w][j][g]
    e("v_cmp_eq_u32", VCC, (v_r, 0))
    e("s_mov_b64", s_tmp64, (EXEC,))
    for db in range(4):
        for i in range(4):
            e("v_accvgpr_read_b32", v_e[i], (a_oc[db][i],))
        k.nop(1)
        e("s_mov_b64", EXEC, (VCC,))
        for i in range(4):
            e("ds_write_b32", (), (v_scr, v_e[i]), offset=(db * 4 + i) * 16)
        e("s_mov_b64", EXEC, (s_tmp64,))
    e("v_accvgpr_read_b32", v_e[0], (a_lc[0],))
    k.nop(1)
    e("s_mov_b64", EXEC, (VCC,))
    e("ds_write_b32", (), (v_scr, v_e[0]), offset=16 * 16)
    e("s_mov_b64", EXEC, (s_tmp64,))
    e("s_waitcnt", lgkmcnt=0)
    e("s_barrier")
    # every wave finishes ONE 16-channel block of the class token's row (d-block w): the four waves' partial sums of l and of its own block
    # from the scratch, one wait, normalise, one 8-byte store from the lanes with n == 0 (lane group g holds d = 16 w + 4 g + i).
    # (Wave 0 doing the whole row held the workgroup's next barrier for ~1k cycles.)
    e("s_mov_b64", EXEC, (VCC,))
    tmp = [v_sb[0][i] for i in range(8)] + [v_sb[1][i] for i in range(8)] + [v_pb[0][i] for i in range(4)]
    e("s_mul_i32", t0, (s_w, 17 * 16))
    e("v_subrev_u32", v_e[5], (t0, v_scr))           # scratch of wave 0 at this lane's slot
    e("s_lshl_b32", t0, (s_w, 6))
    e("v_add_u32", v_e[6], (t0, v_e[5]))             # ... at value 4 w
    for ww in range(4):
        e("ds_read_b32", tmp[16 + ww], (v_e[5],), offset=ww * 17 * 16 + 16 * 16)
        for i in range(4):
            e("ds_read_b32", tmp[ww * 4 + i], (v_e[6],), offset=ww * 17 * 16 + i * 16)
    e("s_waitcnt", lgkmcnt=0)
    e("v_add_f32", tmp[16], (tmp[16], tmp[17]))
    e("v_add_f32", tmp[18], (tmp[18], tmp[19]))
    e("v_add_f32", v_e[4], (tmp[16], tmp[18]))
    e("v_cmp_nle_f32", VCC, (s_lo, v_e[4]))      # not (2^-64 <= l): too small, or NaN
    e("s_or_b64", s_bad, (s_bad, VCC))
    e("v_cmp_ngt_f32", VCC, (s_hi, v_e[4]))      # not (2^100 > l): too large, inf or NaN
    e("s_or_b64", s_bad, (s_bad, VCC))
    e("v_rcp_f32", v_e[4], (v_e[4],))
    for i in range(4):
        e("v_add_f32", tmp[i], (tmp[i], tmp[4 + i]))
        e("v_add_f32", tmp[8 + i], (tmp[8 + i], tmp[12 + i]))
    for i in range(4):
        e("v_add_f32", tmp[i], (tmp[i], tmp[8 + i]))
    for i in range(4):
        e("v_mul_f32", tmp[i], (tmp[i], v_e[4]))
    e("v_cvt_pk_bf16_f32", v_e[6], (tmp[0], tmp[1]))
    e("v_cvt_pk_bf16_f32", v_e[7], (tmp[2], tmp[3]))
    e("s_lshl_b32", t0, (s_w, 5))
    e("buffer_store_dwordx2", (), (R("v", v_e[6].i, 2), v_oco, s_od, t0))
    k.nop(1)
    e("s_mov_b64", EXEC, (s_tmp64,))
    stamp2(5)
    e("v_add_u32", v_ord, (-NB * 2048, v_ord))   # the next unit's tiles read the staging area from its first block again
    e("v_cmp_nle_f32", VCC, (s_lo, v_lmin))      # not (2^-64 <= the smallest row sum): too small, or NaN
    e("s_or_b64", s_bad, (s_bad, VCC))
    e("v_cmp_ngt_f32", VCC, (s_hi, v_lsum))      # not (2^100 > the total): a sum too large, inf or NaN
    e("s_or_b64", s_bad, (s_bad, VCC))
    # a row sum out of range: this unit runs again in the HIP kernel's safe body
    e("s_or_b32", t0, (s_bad[0], s_bad[1]))
    e("s_cbranch_scc0", (), ("L_flag_done",))
    e("v_mov_b32", v_e[0], (1,))
    e("s_mov_b64", s_tmp64, (EXEC,))
    e("s_mov_b64", EXEC, (1,))
    k.nop(1)
    e("global_store_dword", (), (v_zero, v_e[0], s_rd))
    e("s_mov_b64", EXEC, (s_tmp64,))
    k.label("L_flag_done")
    stamp(6)
    if "stamps" in abl:
        e("s_add_u32", s(97), (s(97), 1))
    e("s_cmp_eq_u32", (), (s_last, 0))
    e("s_cbranch_scc1", (), ("L_unit",))
    # the workgroup's last unit: nobody is left to carry its rows out
    e("s_mov_b32", s_tbo, (0,))
    e("s_lshl_b32", t1, (s_D, 4))             # 8 rows
    for b in range(NB):
        for h in range(2):
            e("ds_read_b128", v_trk[h], (v_ord,), offset=b * 2048 + 1024 * h)
        e("s_waitcnt", lgkmcnt=0)
        if "nostore" not in abl:
            e("buffer_store_dwordx4", (), (v_trk[0], v_oo2, s_od, s_tbo))
            e("s_add_u32", s_tbo, (s_tbo, t1))
            e("buffer_store_dwordx4", (), (v_trk[1], v_oo2, s_od, s_tbo))
            e("s_add_u32", s_tbo, (s_tbo, t1))
            k.nop(1)
    e("s_waitcnt", vmcnt=0)
    e("s_endpgm")
    # at L_unit the younger vector-memory operations are this wave's output stores (the class token's four, wave 0 only, are the oldest
    # of them): with no more than the main stores in flight, the Q loads and every LDS-DMA piece in front of them have landed
    k.n_vgpr = k.vp.next
    k.n_acc = n_acc
    k.kernarg = 72 if "stamps" in abl else 64
    return k


HEADER = """\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"
\t.amdhsa_code_object_version 6
\t.text
\t.protected\tmd_attn577_bf16
\t.globl\tmd_attn577_bf16
\t.p2align\t8
\t.type\tmd_attn577_bf16,@function
md_attn577_bf16:
"""

FOOTER = """\t.section\t.rodata,"a",@progbits
\t.p2align\t6, 0x0
\t.amdhsa_kernel md_attn577_bf16
\t\t.amdhsa_group_segment_fixed_size {lds}
\t\t.amdhsa_private_segment_fixed_size 0
\t\t.amdhsa_kernarg_size {karg}
\t\t.amdhsa_user_sgpr_count 2
\t\t.amdhsa_user_sgpr_kernarg_segment_ptr 1
\t\t.amdhsa_system_sgpr_workgroup_id_x 1
\t\t.amdhsa_system_sgpr_workgroup_id_y 0
\t\t.amdhsa_system_sgpr_workgroup_id_z 0
\t\t.amdhsa_system_vgpr_workitem_id 0
\t\t.amdhsa_next_free_vgpr 512
\t\t.amdhsa_next_free_sgpr {nsgpr}
\t\t.amdhsa_accum_offset 256
\t\t.amdhsa_reserve_vcc 1
\t\t.amdhsa_float_round_mode_32 0
\t\t.amdhsa_float_round_mode_16_64 0
\t\t.amdhsa_float_denorm_mode_32 3
\t\t.amdhsa_float_denorm_mode_16_64 3
\t\t.amdhsa_dx10_clamp 1
\t\t.amdhsa_ieee_mode 1
\t.end_amdhsa_kernel
\t.text
\t.amdgpu_metadata
---
amdhsa.kernels:
  - .agpr_count:     256
    .args:
      - {{.address_space: global, .offset: 0, .size: 8, .value_kind: global_buffer}}
      - {{.address_space: global, .offset: 8, .size: 8, .value_kind: global_buffer}}
      - {{.address_space: global, .offset: 16, .size: 8, .value_kind: global_buffer}}
      - {{.address_space: global, .offset: 24, .size: 8, .value_kind: global_buffer}}
      - {{.offset: 32, .size: 4, .value_kind: by_value}}
      - {{.offset: 36, .size: 4, .value_kind: by_value}}
      - {{.offset: 40, .size: 4, .value_kind: by_value}}
      - {{.offset: 44, .size: 4, .value_kind: by_value}}
      - {{.offset: 48, .size: 4, .value_kind: by_value}}
      - {{.offset: 52, .size: 4, .value_kind: by_value}}
      - {{.offset: 56, .size: 4, .value_kind: by_value}}
      - {{.offset: 60, .size: 4, .value_kind: by_value}}
{dbgarg}    .group_segment_fixed_size: {lds}
    .kernarg_segment_align: 8
    .kernarg_segment_size: {karg}
    .max_flat_workgroup_size: 256
    .name:           md_attn577_bf16
    .private_segment_fixed_size: 0
    .sgpr_count:     {nsgpr}
    .sgpr_spill_count: 0
    .symbol:         md_attn577_bf16.kd
    .uniform_work_group_size: 1
    .uses_dynamic_stack: false
    .vgpr_count:     512
    .vgpr_spill_count: 0
    .wavefront_size: 64
amdhsa.target:   amdgcn-amd-amdhsa--gfx950
amdhsa.version:
  - 1
  - 2
...
\t.end_amdgpu_metadata
"""


def render(k):
    out = [HEADER]
    for ins in k.p:
        t = ins.text()
        out.append(t + "\n" if ins.op == "label" else "\t" + t + "\n")
    karg = getattr(k, "kernarg", 64)
    dbgarg = "      - {.address_space: global, .offset: 64, .size: 8, .value_kind: global_buffer}\n" if karg > 64 else ""
    out.append(FOOTER.format(lds=LDS_BYTES, karg=karg, dbgarg=dbgarg, nsgpr=100 if karg > 64 else 96))
    return "".join(out)


if __name__ == "__main__":
    kk = build([a_ for a_ in sys.argv[1:] if not a_.startswith("-")])
    bad = check_hazards(kk.p)
    if bad and "--force" not in sys.argv:
        for b_ in bad[:40]:
            print("HAZARD", b_, file=sys.stderr)
        sys.exit(1)
    print(f"; generated by tools/attn_asm/gen_attn577.py: {len(kk.p)} instructions, {kk.n_vgpr} VGPRs, {kk.n_acc} AGPRs", file=sys.stderr)
    sys.stdout.write(render(kk))
