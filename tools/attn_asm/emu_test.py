"""CPU check of the generated attention kernel's data flow: runs the four waves of one workgroup in the numpy emulator (isa.Emu) on
random bf16 q | k rows and V^T, and compares all 577 output rows of that (sequence, head) with a float64 softmax(q k^T) v.
  python tools/attn_asm/emu_test.py        (about a minute)"""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_attn577 as G  # noqa: E402
from isa import Emu, Mem, Wave, bf16_to_f32, f32_to_bf16  # noqa: E402


def main(seed=0, heads=2, head=1, scale=0.35, big_row=None):
    rng = np.random.default_rng(seed)
    N, D, kpad = 577, heads * 64, 640
    q = (rng.standard_normal((N, D)) * scale).astype(np.float32)
    kk = (rng.standard_normal((N, D)) * scale * 4).astype(np.float32)
    vv = rng.standard_normal((N, D)).astype(np.float32)
    if big_row is not None:
        q[big_row] *= 40.0
    qb, kb, vb = f32_to_bf16(q), f32_to_bf16(kk), f32_to_bf16(vv)
    rows = np.zeros((N + 64, 2 * D), np.uint16)
    rows[:N, :D] = qb
    rows[:N, D:] = kb
    rows[N:] = 0x7F80  # +inf in the slack rows: a masked key must not reach the sums
    vT = np.zeros((1, heads, 64, kpad), np.uint16)
    for h in range(heads):
        vT[0, h, :, :N] = vb[:, h * 64:(h + 1) * 64].T
    out = np.zeros((N, D), np.uint16)
    redo = np.zeros(heads, np.uint32)
    mem = Mem()
    a_qk, a_vt, a_out, a_redo = mem.add(rows), mem.add(vT), mem.add(out), mem.add(redo)
    karg = np.zeros(14, np.uint32)
    for i, ad in enumerate((a_qk, a_vt, a_out, a_redo)):
        karg[2 * i] = ad & 0xFFFFFFFF
        karg[2 * i + 1] = ad >> 32
    karg[8:13] = (N, N, heads, D, kpad)
    a_k = mem.add(karg)
    kern = G.build()
    emu = Emu(kern.p, mem, lds_bytes=G.LDS_BYTES)
    waves = []
    for w in range(4):
        wv = Wave()
        wv.s[0], wv.s[1] = a_k & 0xFFFFFFFF, a_k >> 32
        wv.s[2], wv.s[3] = head, 0
        wv.v[0] = np.arange(64, dtype=np.uint32) + 64 * w
        waves.append(wv)
    emu.run(waves)
    got = bf16_to_f32(mem.get(a_out).view(np.uint16).reshape(N, D)[:, head * 64:(head + 1) * 64])
    qf = bf16_to_f32(qb)[:, head * 64:(head + 1) * 64].astype(np.float64)
    kf = bf16_to_f32(kb)[:, head * 64:(head + 1) * 64].astype(np.float64)
    vf = bf16_to_f32(vb)[:, head * 64:(head + 1) * 64].astype(np.float64)
    s_ = qf @ kf.T
    p = np.exp2(s_)
    pb = bf16_to_f32(f32_to_bf16(p.astype(np.float32))).astype(np.float64)
    ref = (pb @ vf) / pb.sum(1, keepdims=True)
    err = np.abs(got - ref)
    tol = 1e-2 * np.abs(ref) + 2e-2
    flag = mem.get(a_redo).view(np.uint32)
    other = bf16_to_f32(mem.get(a_out).view(np.uint16).reshape(N, D)[:, (1 - head) * 64:(2 - head) * 64]) if heads == 2 else None
    print(f"instructions executed {emu.count}, max err {err.max():.4g} (row {np.unravel_index(err.argmax(), err.shape)}), redo flags {flag.tolist()}, "
          f"|score| max {np.abs(s_).max():.1f}")
    ok = bool((err <= tol).all()) and (other is None or not other.any())
    worst = np.argsort(-(err - tol).max(1))[:5]
    if not ok:
        print("rows off:", int(((err > tol).any(1)).sum()), "worst rows", worst.tolist())
        print("row 0 err", err[0].max(), "row 1", err[1].max(), "row 576", err[576].max())
    return ok, flag


if __name__ == "__main__":
    ok, flag = main()
    print("PASS" if ok and flag[1] == 0 else "FAIL")
    sys.exit(0 if ok else 1)
