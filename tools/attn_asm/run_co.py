"""Runs generated variants of the attention kernel straight from their code objects (hipModuleLoad through ctypes; torch only for
device memory and events): timing ablations without rebuilding the library.
  python tools/attn_asm/run_co.py [T] -- variant ...      each variant = a '+'-joined list of gen_attn577.py ablations, '' = the kernel"""
import ctypes as C
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LLVM = "/opt/rocm/lib/llvm/bin"


def build(variant, out):
    args = [a for a in variant.split("+") if a]
    src = subprocess.run([sys.executable, os.path.join(HERE, "gen_attn577.py"), "--force"] + args, check=True, capture_output=True, text=True).stdout
    with open(out + ".s", "w") as f:
        f.write(src)
    subprocess.run([f"{LLVM}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", out + ".s", "-o", out + ".o"], check=True)
    subprocess.run([f"{LLVM}/ld.lld", "-shared", out + ".o", "-o", out + ".co"], check=True)
    return out + ".co"


def main():
    argv = sys.argv[1:]
    T = int(argv[0]) if argv and argv[0].isdigit() else 296
    variants = argv[argv.index("--") + 1:] if "--" in argv else [""]
    hip = C.CDLL("libamdhip64.so")
    heads, N, S, kpad = 16, 577, 580, 640
    D = heads * 64
    torch.manual_seed(0)
    qk = (torch.rand(T * S + 64, 2 * D, device="cuda") * 1.4 - 0.7).to(torch.bfloat16)
    vT = (torch.rand(T, heads, 64, kpad, device="cuda") * 2 - 1).to(torch.bfloat16)
    out = torch.zeros(T * S + 64, D, device="cuda", dtype=torch.bfloat16)
    redo = torch.zeros(T * heads, device="cuda", dtype=torch.int32)

    class Args(C.Structure):
        _fields_ = [("qk", C.c_void_p), ("vT", C.c_void_p), ("out", C.c_void_p), ("redo", C.c_void_p),
                    ("S", C.c_int), ("n", C.c_int), ("heads", C.c_int), ("D", C.c_int), ("kpad", C.c_int), ("hlog", C.c_int),
                    ("nunits", C.c_int), ("grid", C.c_int), ("dbg", C.c_void_p)]

    grid = min(T * heads, 256)
    args = Args(qk.data_ptr(), vT.data_ptr(), out.data_ptr(), redo.data_ptr(), S, N, heads, D, kpad, 4, T * heads, grid, 0)
    dbg = torch.zeros(64, device="cuda", dtype=torch.int64)
    args.dbg = dbg.data_ptr()
    size = C.c_size_t(C.sizeof(args))
    extra = (C.c_void_p * 5)(1, C.cast(C.pointer(args), C.c_void_p), 2, C.cast(C.pointer(size), C.c_void_p), 3)
    fl = 4.0 * T * heads * N * N * 64
    os.makedirs("/tmp/attn_co", exist_ok=True)
    for i, var in enumerate(variants):
        co = build(var, f"/tmp/attn_co/v{i}")
        mod, fn = C.c_void_p(), C.c_void_p()
        assert hip.hipModuleLoad(C.byref(mod), co.encode()) == 0
        assert hip.hipModuleGetFunction(C.byref(fn), mod, b"md_attn577_bf16") == 0

        def launch():
            r = hip.hipModuleLaunchKernel(fn, grid, 1, 1, 256, 1, 1, 0, None, None, extra)
            assert r == 0, r

        for _ in range(3):
            launch()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                launch()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        print(f"{var or 'kernel':28s} {best * 1e3:8.1f} us  {fl / best / 1e9:6.0f} TF  per unit and CU {best * 1e3 / (T * heads / 256):6.2f} us", flush=True)
        if "stamps" in var:
            st = dbg.cpu().numpy().reshape(8, 8)
            names = ["entry", "unit start", "tile 0", "loop", "drain+tail", "hand-over", "epilogue", ""]
            if "stamps2" in var:
                names = ["entry", "unit start", "tile 0 to B(0)", "B(0) wait+barrier", "rest of tile 0", "... to the class token done", "main out", ""]
            for u in range(4):
                row = st[u]
                prev = st[u - 1][6] if u else row[0]
                parts = []
                for j in range(1, 7):
                    parts.append(f"{names[j]} {int(row[j] - prev)}")
                    prev = row[j]
                print(f"   unit {u}: " + ", ".join(parts) + f" | total {int(row[6] - (st[u - 1][6] if u else row[0]))} ticks")
        hip.hipModuleUnload(mod)


if __name__ == "__main__":
    main()
