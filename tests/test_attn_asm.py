"""The assembly-owned Depth Pro attention kernel (burn_depth_amd/csrc/kernels/attn577_gfx950.s) on the CPU: the committed file is what
tools/attn_asm/gen_attn577.py generates, its instruction stream passes the wait-state checker, and its data flow -- MFMA operand
layouts, LDS swizzles, LDS-DMA addressing, the class token's partial sums, the hand-over between units of a persistent workgroup,
the range flag -- reproduces softmax(q k^T) v in the numpy emulator (tools/attn_asm/isa.py). The GPU side is tests/test_gpu_parity.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "attn_asm"))

import gen_attn577 as G  # noqa: E402
import emu_test  # noqa: E402
from isa import I, R, check_hazards, v  # noqa: E402


def test_committed_assembly_is_the_generators_output():
    with open(os.path.join(ROOT, "burn_depth_amd", "csrc", "kernels", "attn577_gfx950.s")) as f:
        committed = f.read()
    assert committed == G.render(G.build()), "run `make -C burn_depth_amd/csrc asm`"


def test_instruction_stream_passes_the_wait_state_checker():
    k = G.build()
    assert check_hazards(k.p) == []
    assert k.n_vgpr <= 256 and k.n_acc <= 256
    # the checker itself: an MFMA result read one instruction later, a fresh VALU result as an MFMA operand
    mf = I("v_mfma_f32_16x16x32_bf16", (v(0, 4),), (v(4, 4), v(8, 4), 0))
    assert check_hazards([mf, I("v_exp_f32", (v(0),), (v(0),))])
    assert check_hazards([I("v_mov_b32", (v(4),), (0,)), mf])
    assert not check_hazards([mf, I("v_mfma_f32_16x16x32_bf16", (v(0, 4),), (v(4, 4), v(8, 4), v(0, 4)))])  # the accumulate chain


def test_emulated_workgroup_matches_softmax_and_raises_only_the_spiked_units_flag():
    # one persistent workgroup over two (sequence, head) units; unit 1 holds a logit of ~110 log2 units (row sum > 2^100)
    ok, flag = emu_test.main(heads=2, nseq=1, grid=1, spike_unit=1)
    assert ok
    assert flag.tolist() == [0, 1]
