"""A small gfx950 instruction IR with (1) an assembly printer, (2) a conservative hazard / wait-state checker and (3) a functional
emulator (numpy, 64 lanes) for the instruction subset the generated attention kernel uses. Test infrastructure for
tools/attn_asm/gen_attn577.py: the emulator checks data flow and layouts (MFMA operand layouts, LDS swizzles, LDS-DMA addressing)
on the CPU; timing, wait states and counters are not modelled -- the checker enforces the wait-state rules measured from hipcc's
own output on gfx950 (an MFMA 16x16x32 result needs 8 states before any reader but the accumulate chain, a VALU or transcendental
result 1 before an MFMA / VALU reader -- the checker asks for 2)."""
import numpy as np


class R:
    """register operand: kind in v / a / s, first index, count"""
    __slots__ = ("k", "i", "n")

    def __init__(self, k, i, n=1):
        self.k, self.i, self.n = k, i, n

    def __getitem__(self, j):
        if isinstance(j, slice):
            st = j.start or 0
            sp = self.n if j.stop is None else j.stop
            return R(self.k, self.i + st, sp - st)
        assert 0 <= j < self.n
        return R(self.k, self.i + j, 1)

    def regs(self):
        return [(self.k, self.i + j) for j in range(self.n)]

    def __str__(self):
        if self.k in ("m0", "vcc", "exec", "scc"):
            return self.k
        return f"{self.k}{self.i}" if self.n == 1 else f"{self.k}[{self.i}:{self.i + self.n - 1}]"


M0 = R("m0", 0)
VCC = R("vcc", 0)
EXEC = R("exec", 0)
OFF = "off"


def v(i, n=1):
    return R("v", i, n)


def a(i, n=1):
    return R("a", i, n)


def s(i, n=1):
    return R("s", i, n)


class Pool:
    def __init__(self, kind, lo, hi):
        self.kind, self.next, self.hi = kind, lo, hi

    def take(self, n=1, align=1):
        self.next = (self.next + align - 1) // align * align
        r = R(self.kind, self.next, n)
        self.next += n
        assert self.next <= self.hi, f"out of {self.kind} registers"
        return r


class I:
    """one instruction: op, destination operands, source operands, modifiers"""

    def __init__(self, op, dst=(), src=(), **mods):
        self.op = op
        self.dst = list(dst) if isinstance(dst, (list, tuple)) else [dst]
        self.src = list(src) if isinstance(src, (list, tuple)) else [src]
        self.mods = mods
        self.comment = mods.pop("comment", None)

    # ---- classification (hazard checker) ----
    def kind(self):
        o = self.op
        if o.startswith("v_mfma"):
            return "mfma"
        if o in ("v_exp_f32", "v_rcp_f32", "v_log_f32", "v_rsq_f32"):
            return "trans"
        if o.startswith("v_"):
            return "valu"
        if o.startswith("ds_"):
            return "ds"
        if o.startswith("buffer_") or o.startswith("global_"):
            return "vmem"
        if o.startswith("s_"):
            return "salu"
        return "other"

    def text(self):
        o, d, sr, m = self.op, self.dst, self.src, self.mods
        f = lambda x: (x if isinstance(x, str) else (str(x) if isinstance(x, R) else fmt_imm(x)))
        if o == "label":
            return f"{sr[0]}:"
        if o == "s_waitcnt":
            parts = []
            if "vmcnt" in m:
                parts.append(f"vmcnt({m['vmcnt']})")
            if "lgkmcnt" in m:
                parts.append(f"lgkmcnt({m['lgkmcnt']})")
            return "s_waitcnt " + " ".join(parts)
        if o in ("s_barrier", "s_endpgm"):
            return o
        if o == "s_nop":
            return f"s_nop {sr[0]}"
        if o.startswith("s_cbranch") or o == "s_branch":
            return f"{o} {sr[0]}"
        if o.startswith("s_load"):
            return f"{o} {f(d[0])}, {f(sr[0])}, {hex(sr[1])}"
        if o.startswith("ds_read"):
            t = f"{o} {f(d[0])}, {f(sr[0])}"
            if m.get("offset"):
                t += f" offset:{m['offset']}"
            return t
        if o.startswith("ds_write"):
            t = f"{o} {f(sr[0])}, {f(sr[1])}"
            if m.get("offset"):
                t += f" offset:{m['offset']}"
            return t
        if o.startswith("buffer_load") and m.get("lds"):
            t = f"{o} {f(sr[0])}, {f(sr[1])}, {f(sr[2])} offen"
            if m.get("offset"):
                t += f" offset:{m['offset']}"
            return t + " lds"
        if o.startswith("buffer_load"):
            t = f"{o} {f(d[0])}, {f(sr[0])}, {f(sr[1])}, {f(sr[2])} offen"
            if m.get("offset"):
                t += f" offset:{m['offset']}"
            return t
        if o.startswith("buffer_store"):
            t = f"{o} {f(sr[0])}, {f(sr[1])}, {f(sr[2])}, {f(sr[3])} offen"
            if m.get("offset"):
                t += f" offset:{m['offset']}"
            return t
        if o == "s_memtime":
            return f"s_memtime {f(d[0])}"
        if o.startswith("global_store"):
            t = f"{o} {f(sr[0])}, {f(sr[1])}, {f(sr[2])}"
            if m.get("offset"):
                t += f" offset:{m['offset']}"
            return t
        ops = [f(x) for x in d] + [f(x) for x in sr]
        return f"{o} " + ", ".join(ops)


def fmt_imm(x):
    if isinstance(x, float):
        if x == 0.0:
            return "0"
        if x in (0.5, 1.0, 2.0, 4.0, -0.5, -1.0, -2.0, -4.0):
            return repr(x)
        return hex(int(np.float32(x).view(np.uint32)))
    if isinstance(x, int):
        return str(x) if -16 <= x <= 64 else hex(x & 0xFFFFFFFF)
    return str(x)


# ------------------------------------------------------------------------------------------------
# hazard checker (straight-line; the generator calls it on the unrolled stream including one loop wrap)
# ------------------------------------------------------------------------------------------------
MFMA_STATES = 9      # 8 measured (hipcc: s_nop 7 behind v_mfma_f32_16x16x32_bf16) + 1 of margin
VALU_STATES = 2      # 1 measured


def states_of(ins):
    if ins.op == "s_nop":
        return ins.src[0] + 1
    if ins.op == "label":
        return 0
    return 1


def check_hazards(prog, name=""):
    """every reader / over-writer of an MFMA result other than the accumulate chain, and every MFMA / VALU reader of a fresh VALU
    result, must be the required number of wait states behind the producer. Returns the list of violations."""
    last = {}  # reg -> (position in states, kind, instruction index, is C-chain capable)
    pos = 0
    bad = []
    for idx, ins in enumerate(prog):
        k = ins.kind()
        if ins.op == "label":
            continue
        reads = []
        for x in ins.src:
            if isinstance(x, R) and x.k in ("v", "a"):
                reads += x.regs()
        writes = []
        for x in ins.dst:
            if isinstance(x, R) and x.k in ("v", "a"):
                writes += x.regs()
        cchain = set()
        if k == "mfma":  # the C operand taken whole as D of the previous MFMA on it: no states needed
            c = ins.src[2]
            if isinstance(c, R) and c.regs() == ins.dst[0].regs():
                cchain = set(c.regs())
        for r in reads + writes:
            if r not in last:
                continue
            p, pk, pidx = last[r]
            gap = pos - p - 1  # states between producer and this instruction
            if pk == "mfma":
                if r in cchain:
                    continue
                if gap < MFMA_STATES:
                    bad.append((name, idx, ins.text(), f"{gap} states behind MFMA #{pidx} on {r}"))
            elif pk in ("valu", "trans") and r in reads:
                # measured (hipcc, gfx950): one state between a transcendental and any vector reader, and between a VALU write and an
                # MFMA reading it; plain VALU -> VALU is interlocked. (v_readfirstlane behind a VALU write: kept at the same margin)
                if pk == "trans":
                    need = VALU_STATES if k in ("mfma", "valu", "trans") else 0
                else:
                    need = VALU_STATES if (k == "mfma" or ins.op == "v_readfirstlane_b32") else 0
                if gap < need:
                    bad.append((name, idx, ins.text(), f"{gap} states behind {pk} #{pidx} on {r}"))
        for r in writes:
            last[r] = (pos, k, idx)
        pos += states_of(ins)
    return bad


# ------------------------------------------------------------------------------------------------
# emulator
# ------------------------------------------------------------------------------------------------
def bf16_to_f32(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16(f):
    u = np.asarray(f, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32) & 0xFFFF
    nan = np.isnan(np.asarray(f, np.float32))
    return np.where(nan, 0x7FC0, r).astype(np.uint32)


class Mem:
    """flat fake global memory: named numpy byte buffers at fixed base addresses"""

    def __init__(self):
        self.bufs = []  # (base, array)
        self.next = 0x10000000

    def add(self, arr):
        b = np.ascontiguousarray(arr).view(np.uint8).reshape(-1).copy()
        base = self.next
        self.bufs.append((base, b))
        self.next = (base + b.size + 0xFFFF) // 0x10000 * 0x10000 + 0x10000
        return base

    def find(self, addr):
        for base, b in self.bufs:
            if base <= addr < base + b.size:
                return base, b
        raise RuntimeError(f"global access outside every buffer: {hex(int(addr))}")

    def read(self, addr, n):
        base, b = self.find(addr)
        o = int(addr - base)
        if o + n > b.size:
            raise RuntimeError("global read past a buffer")
        return b[o:o + n]

    def write(self, addr, data):
        base, b = self.find(addr)
        o = int(addr - base)
        if o + len(data) > b.size:
            raise RuntimeError("global write past a buffer")
        b[o:o + len(data)] = data

    def get(self, base):
        for bb, b in self.bufs:
            if bb == base:
                return b
        raise KeyError


class Wave:
    def __init__(self):
        self.v = np.zeros((256, 64), np.uint32)
        self.a = np.zeros((256, 64), np.uint32)
        self.s = np.zeros(128, np.uint32)
        self.m0 = 0
        self.vcc = 0
        self.exec = (1 << 64) - 1
        self.scc = 0
        self.pc = 0
        self.done = False
        self.at_barrier = False


class Emu:
    def __init__(self, prog, mem, lds_bytes=163840):
        self.prog = prog
        self.labels = {ins.src[0]: i for i, ins in enumerate(prog) if ins.op == "label"}
        self.mem = mem
        self.lds = np.zeros(lds_bytes, np.uint8)
        self.count = 0

    # operand access
    def rd(self, w, x, lanes=True):
        if isinstance(x, R):
            if x.k == "v":
                return w.v[x.i]
            if x.k == "a":
                return w.a[x.i]
            if x.k == "s":
                return np.full(64, w.s[x.i], np.uint32) if lanes else np.uint32(w.s[x.i])
            if x.k == "m0":
                return np.full(64, w.m0, np.uint32) if lanes else np.uint32(w.m0)
            raise NotImplementedError(x.k)
        if isinstance(x, float):
            u = np.float32(x).view(np.uint32)
            return np.full(64, u, np.uint32) if lanes else u
        u = np.uint32(int(x) & 0xFFFFFFFF)
        return np.full(64, u, np.uint32) if lanes else u

    def rds(self, w, x):
        return int(self.rd(w, x, lanes=False))

    def rd64(self, w, x):
        return int(w.s[x.i]) | (int(w.s[x.i + 1]) << 32)

    def wr(self, w, x, val):
        val = np.asarray(val).astype(np.uint32)
        if x.k == "v":
            m = self.emask(w)
            w.v[x.i] = np.where(m, val, w.v[x.i])
        elif x.k == "a":
            m = self.emask(w)
            w.a[x.i] = np.where(m, val, w.a[x.i])
        elif x.k == "s":
            w.s[x.i] = np.uint32(val if np.ndim(val) == 0 else val[0])
        elif x.k == "m0":
            w.m0 = int(val if np.ndim(val) == 0 else val[0])
        else:
            raise NotImplementedError(x.k)

    def emask(self, w):
        return np.array([(w.exec >> i) & 1 for i in range(64)], bool)

    def desc(self, w, x):
        base = int(w.s[x.i]) | ((int(w.s[x.i + 1]) & 0xFFFF) << 32)
        return base, int(w.s[x.i + 2])

    def run(self, waves, max_steps=10_000_000):
        """round-robin between barriers: a wave runs until s_barrier / s_endpgm; the barrier opens when every live wave waits"""
        while True:
            live = [w for w in waves if not w.done]
            if not live:
                return
            progressed = False
            for w in live:
                if w.at_barrier:
                    continue
                self.run_wave(w, max_steps)
                progressed = True
            live = [w for w in waves if not w.done]
            if live and all(w.at_barrier for w in live):
                for w in live:
                    w.at_barrier = False
                progressed = True
            if not progressed:
                raise RuntimeError("deadlock")

    def run_wave(self, w, max_steps):
        P = self.prog
        while True:
            self.count += 1
            if self.count > max_steps:
                raise RuntimeError("step limit")
            ins = P[w.pc]
            w.pc += 1
            o = ins.op
            if o == "label" or o == "s_nop" or o == "s_waitcnt":
                continue
            if o == "s_barrier":
                w.at_barrier = True
                return
            if o == "s_endpgm":
                w.done = True
                return
            self.step(w, ins)

    def step(self, w, ins):
        o, d, sr, m = ins.op, ins.dst, ins.src, ins.mods
        u32 = np.uint32
        if o.startswith("s_load_dword"):
            n = {"s_load_dword": 1, "s_load_dwordx2": 2, "s_load_dwordx4": 4, "s_load_dwordx8": 8, "s_load_dwordx16": 16}[o]
            addr = self.rd64(w, sr[0]) + sr[1]
            data = self.mem.read(addr, 4 * n).view(np.uint32)
            for j in range(n):
                w.s[d[0].i + j] = data[j]
            return
        if o in ("s_mov_b32",):
            self.wr(w, d[0], self.rds(w, sr[0]))
            return
        if o == "s_mov_b64":
            if d[0].k == "exec":
                if isinstance(sr[0], int):
                    w.exec = sr[0] if sr[0] >= 0 else (1 << 64) - 1
                elif sr[0].k == "vcc":
                    w.exec = w.vcc
                else:
                    w.exec = self.rd64(w, sr[0])
            else:
                if isinstance(sr[0], R):
                    val = self.rd64(w, sr[0]) if sr[0].k == "s" else (w.vcc if sr[0].k == "vcc" else w.exec)
                else:
                    val = int(sr[0])
                w.s[d[0].i] = u32(val & 0xFFFFFFFF)
                w.s[d[0].i + 1] = u32(val >> 32)
            return
        if o in ("s_add_u32", "s_add_i32", "s_addc_u32", "s_sub_u32", "s_sub_i32", "s_mul_i32", "s_mul_hi_u32", "s_lshl_b32", "s_lshr_b32",
                 "s_and_b32", "s_or_b32", "s_xor_b32", "s_min_u32"):
            x, y = self.rds(w, sr[0]), self.rds(w, sr[1])
            if o in ("s_add_u32", "s_add_i32"):
                r = x + y
                w.scc = 1 if r > 0xFFFFFFFF else 0
            elif o == "s_addc_u32":
                r = x + y + w.scc
                w.scc = 1 if r > 0xFFFFFFFF else 0
            elif o in ("s_sub_u32", "s_sub_i32"):
                r = x - y
                w.scc = 1 if x < y else 0
            elif o == "s_min_u32":
                r = min(x, y)
                w.scc = 1 if x < y else 0
            elif o == "s_mul_i32":
                r = x * y
            elif o == "s_mul_hi_u32":
                r = (x * y) >> 32
            elif o == "s_lshl_b32":
                r = x << (y & 31)
            elif o == "s_lshr_b32":
                r = x >> (y & 31)
            elif o == "s_and_b32":
                r = x & y
                w.scc = 1 if (r & 0xFFFFFFFF) else 0
            elif o == "s_or_b32":
                r = x | y
                w.scc = 1 if (r & 0xFFFFFFFF) else 0
            else:
                r = x ^ y
                w.scc = 1 if (r & 0xFFFFFFFF) else 0
            self.wr(w, d[0], r & 0xFFFFFFFF)
            return
        if o == "s_or_b64":
            def g64(x):
                if isinstance(x, R) and x.k == "vcc":
                    return w.vcc
                if isinstance(x, R) and x.k == "exec":
                    return w.exec
                return self.rd64(w, x)
            r = g64(sr[0]) | g64(sr[1])
            w.s[d[0].i] = u32(r & 0xFFFFFFFF)
            w.s[d[0].i + 1] = u32(r >> 32)
            w.scc = 1 if r else 0
            return
        if o.startswith("s_cmp_"):
            x, y = self.rds(w, sr[0]), self.rds(w, sr[1])
            sx = x - (1 << 32) if x & 0x80000000 else x
            sy = y - (1 << 32) if y & 0x80000000 else y
            w.scc = int({"s_cmp_lt_u32": x < y, "s_cmp_eq_u32": x == y, "s_cmp_lg_u32": x != y, "s_cmp_ge_u32": x >= y,
                         "s_cmp_lt_i32": sx < sy, "s_cmp_gt_i32": sx > sy, "s_cmp_gt_u32": x > y}[o])
            return
        if o == "s_cbranch_scc1":
            if w.scc:
                w.pc = self.labels[sr[0]]
            return
        if o == "s_cbranch_scc0":
            if not w.scc:
                w.pc = self.labels[sr[0]]
            return
        if o == "s_cbranch_vccz":
            if w.vcc == 0:
                w.pc = self.labels[sr[0]]
            return
        if o == "s_cbranch_vccnz":
            if w.vcc != 0:
                w.pc = self.labels[sr[0]]
            return
        if o == "s_branch":
            w.pc = self.labels[sr[0]]
            return
        # ---- VALU ----
        if o == "v_readfirstlane_b32":
            w.s[d[0].i] = self.rd(w, sr[0])[0]
            return
        if o in ("v_mov_b32", "v_accvgpr_write_b32", "v_accvgpr_read_b32"):
            self.wr(w, d[0], self.rd(w, sr[0]))
            return
        if o in ("v_lshlrev_b32", "v_lshrrev_b32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_mul_lo_u32",
                 "v_mul_u32_u24"):
            x = self.rd(w, sr[0]).astype(np.uint64)
            y = self.rd(w, sr[1]).astype(np.uint64)
            if o == "v_lshlrev_b32":
                r = y << (x & 31)
            elif o == "v_lshrrev_b32":
                r = y >> (x & 31)
            elif o == "v_and_b32":
                r = x & y
            elif o == "v_or_b32":
                r = x | y
            elif o == "v_xor_b32":
                r = x ^ y
            elif o == "v_add_u32":
                r = x + y
            elif o == "v_sub_u32":
                r = x - y
            elif o == "v_subrev_u32":
                r = y - x
            elif o == "v_mul_u32_u24":
                r = (x & 0xFFFFFF) * (y & 0xFFFFFF)
            else:
                r = x * y
            self.wr(w, d[0], (r & 0xFFFFFFFF).astype(np.uint32))
            return
        if o == "v_mad_u32_u24":
            x = self.rd(w, sr[0]).astype(np.uint64) & 0xFFFFFF
            y = self.rd(w, sr[1]).astype(np.uint64) & 0xFFFFFF
            z = self.rd(w, sr[2]).astype(np.uint64)
            self.wr(w, d[0], ((x * y + z) & 0xFFFFFFFF).astype(np.uint32))
            return
        if o in ("v_exp_f32", "v_rcp_f32"):
            x = self.rd(w, sr[0]).view(np.float32)
            with np.errstate(all="ignore"):
                r = np.exp2(x.astype(np.float64)).astype(np.float32) if o == "v_exp_f32" else (np.float32(1.0) / x)
            self.wr(w, d[0], r.view(np.uint32))
            return
        if o in ("v_mul_f32", "v_add_f32", "v_max_f32", "v_min_f32"):
            x = self.rd(w, sr[0]).view(np.float32)
            y = self.rd(w, sr[1]).view(np.float32)
            with np.errstate(all="ignore"):
                r = {"v_mul_f32": x * y, "v_add_f32": x + y, "v_max_f32": np.maximum(x, y), "v_min_f32": np.minimum(x, y)}[o]
            self.wr(w, d[0], r.astype(np.float32).view(np.uint32))
            return
        if o == "v_pk_mul_f32":
            for j in range(2):
                x = self.rd(w, sr[0][j]).view(np.float32)
                y = self.rd(w, sr[1][j]).view(np.float32)
                with np.errstate(all="ignore"):
                    self.wr(w, d[0][j], (x * y).astype(np.float32).view(np.uint32))
            return
        if o == "v_cvt_pk_bf16_f32":
            x = self.rd(w, sr[0]).view(np.float32)
            y = self.rd(w, sr[1]).view(np.float32)
            self.wr(w, d[0], f32_to_bf16(x) | (f32_to_bf16(y) << 16))
            return
        if o == "v_cndmask_b32":  # dst = mask ? src1 : src0 ; mask = sgpr pair or vcc
            x = self.rd(w, sr[0])
            y = self.rd(w, sr[1])
            mk = w.vcc if (isinstance(sr[2], R) and sr[2].k == "vcc") else self.rd64(w, sr[2])
            sel = np.array([(mk >> i) & 1 for i in range(64)], bool)
            self.wr(w, d[0], np.where(sel, y, x))
            return
        if o.startswith("v_cmp_"):
            x = self.rd(w, sr[0])
            y = self.rd(w, sr[1])
            fx, fy = x.view(np.float32), y.view(np.float32)
            with np.errstate(all="ignore"):
                res = {"v_cmp_eq_u32": x == y, "v_cmp_ne_u32": x != y, "v_cmp_lt_u32": x < y, "v_cmp_ge_u32": x >= y,
                       "v_cmp_lt_f32": fx < fy, "v_cmp_gt_f32": fx > fy, "v_cmp_nlt_f32": ~(fx < fy), "v_cmp_ngt_f32": ~(fx > fy),
                       "v_cmp_nge_f32": ~(fx >= fy), "v_cmp_nle_f32": ~(fx <= fy)}[o]
            res = res & self.emask(w)
            bits = 0
            for i in range(64):
                if res[i]:
                    bits |= 1 << i
            if d[0].k == "vcc":
                w.vcc = bits
            else:
                w.s[d[0].i] = u32(bits & 0xFFFFFFFF)
                w.s[d[0].i + 1] = u32(bits >> 32)
            return
        if o == "v_mfma_f32_16x16x32_bf16":
            A = np.zeros((16, 32), np.float32)
            B = np.zeros((32, 16), np.float32)
            ar = np.stack([self.rd(w, sr[0][j]) for j in range(4)])  # [4 regs][64 lanes]
            br = np.stack([self.rd(w, sr[1][j]) for j in range(4)])
            for lane in range(64):
                rr, g = lane & 15, lane >> 4
                for j in range(8):
                    wa = ar[j >> 1, lane]
                    wb = br[j >> 1, lane]
                    A[rr, 8 * g + j] = bf16_to_f32(np.array([(wa >> (16 * (j & 1))) & 0xFFFF], np.uint32))[0]
                    B[8 * g + j, rr] = bf16_to_f32(np.array([(wb >> (16 * (j & 1))) & 0xFFFF], np.uint32))[0]
            with np.errstate(all="ignore"):
                Dm = A.astype(np.float64) @ B.astype(np.float64)
            if isinstance(sr[2], R):
                cr = np.stack([self.rd(w, sr[2][j]) for j in range(4)]).view(np.float32)
            else:
                cr = np.zeros((4, 64), np.float32)
            out = np.zeros((4, 64), np.float32)
            with np.errstate(all="ignore"):
                for lane in range(64):
                    n, g = lane & 15, lane >> 4
                    for i in range(4):
                        out[i, lane] = np.float32(Dm[4 * g + i, n] + cr[i, lane])
            for i in range(4):
                self.wr(w, d[0][i], out[i].view(np.uint32))
            return
        # ---- LDS ----
        if o in ("ds_read_b128", "ds_read_b64", "ds_read_b32"):
            n = {"ds_read_b128": 4, "ds_read_b64": 2, "ds_read_b32": 1}[o]
            addr = self.rd(w, sr[0]).astype(np.int64) + m.get("offset", 0)
            em = self.emask(w)
            for lane in range(64):
                if not em[lane]:
                    continue
                ad = int(addr[lane])
                assert ad % (4 * n if n < 4 else 16) == 0, f"misaligned {o} at {ad}"
                words = self.lds[ad:ad + 4 * n].view(np.uint32)
                for j in range(n):
                    w.v[d[0].i + j, lane] = words[j]
            return
        if o in ("ds_write_b32", "ds_write_b64", "ds_write_b128"):
            n = {"ds_write_b32": 1, "ds_write_b64": 2, "ds_write_b128": 4}[o]
            addr = self.rd(w, sr[0]).astype(np.int64) + m.get("offset", 0)
            em = self.emask(w)
            for lane in range(64):
                if not em[lane]:
                    continue
                ad = int(addr[lane])
                for j in range(n):
                    self.lds[ad + 4 * j:ad + 4 * j + 4] = np.array([self.rd(w, sr[1][j])[lane]], np.uint32).view(np.uint8)
            return
        # ---- buffer / global ----
        if o == "buffer_load_dwordx4":
            if m.get("lds"):
                voff, dsc, soff = self.rd(w, sr[0]), sr[1], self.rds(w, sr[2])
            else:
                voff, dsc, soff = self.rd(w, sr[0]), sr[1], self.rds(w, sr[2])
            base, nrec = self.desc(w, dsc)
            imm = m.get("offset", 0)
            em = self.emask(w)
            for lane in range(64):
                if not em[lane]:
                    continue
                off = int(voff[lane]) + imm
                inr = off + 16 <= nrec  # raw buffer: the range check covers the vector offset + the instruction offset
                if inr:
                    data = self.mem.read(base + off + soff, 16).view(np.uint32)
                else:
                    data = np.zeros(4, np.uint32)
                if m.get("lds"):
                    la = (w.m0 & 0x3FFFF) + lane * 16
                    self.lds[la:la + 16] = data.view(np.uint8)
                else:
                    bank = w.a if d[0].k == "a" else w.v
                    for j in range(4):
                        bank[d[0].i + j, lane] = data[j]
            return
        if o in ("buffer_store_dwordx2", "buffer_store_dword", "buffer_store_dwordx4"):
            n = {"buffer_store_dword": 1, "buffer_store_dwordx2": 2, "buffer_store_dwordx4": 4}[o]
            data, voff, dsc, soff = sr[0], self.rd(w, sr[1]), sr[2], self.rds(w, sr[3])
            base, nrec = self.desc(w, dsc)
            imm = m.get("offset", 0)
            em = self.emask(w)
            for lane in range(64):
                if not em[lane]:
                    continue
                off = int(voff[lane]) + imm
                if off + 4 * n > nrec:
                    continue
                words = np.array([self.rd(w, data[j])[lane] for j in range(n)], np.uint32)
                self.mem.write(base + off + soff, words.view(np.uint8))
            return
        if o == "global_store_dword":
            voff, data, sb = self.rd(w, sr[0]), sr[1], self.rd64(w, sr[2])
            em = self.emask(w)
            for lane in range(64):
                if em[lane]:
                    self.mem.write(sb + int(voff[lane]) + m.get("offset", 0), np.array([self.rd(w, data)[lane]], np.uint32).view(np.uint8))
            return
        raise NotImplementedError(o)
