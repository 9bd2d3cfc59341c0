"""CPU-only checks of the C-ABI library: it loads, exports every symbol include/mi_depth.h declares,
and its host-side logic (inventory, seeded generator, geometry, scalar tail) agrees with the Python
host code and the oracle.  No compute kernel is launched here."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest
import torch

from burn_depth_amd import _lib
from burn_depth_amd import weights as Wt
from burn_depth_amd.config import DepthProConfig
from burn_depth_amd.depth_pro import _c_cfg
from oracle import depth_pro_ref as R


@pytest.fixture(scope="module")
def lib():
    return _lib.load()


def test_library_exports_every_declared_symbol(lib, repo_root):
    header = open(os.path.join(repo_root, "include", "mi_depth.h")).read()
    declared = set(re.findall(r"^(?:int|long|void|const char\*)\s+(md_[a-z0-9_]+)\s*\(", header, re.M))
    assert declared, "no symbols parsed from the header"
    for name in sorted(declared):
        assert hasattr(lib, name), f"libmi_depth.so does not export {name}"
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))


def test_cfg_default_matches_reference(lib):
    c = _lib.MdDepthProCfg()
    lib.md_depth_pro_cfg_default(C.byref(c))
    # depth_pro/mod.rs:54-66
    assert c.patch_encoder_preset == b"dinov2l16_384" and c.image_encoder_preset == b"dinov2l16_384"
    assert c.fov_encoder_preset == b"dinov2l16_384"
    assert c.decoder_features == 256 and c.use_fov_head == 1 and c.interpolation == 0


@pytest.mark.parametrize("cfg", [DepthProConfig(), DepthProConfig.small_test(), DepthProConfig.tiny_test()])
@pytest.mark.parametrize("scheme", [Wt.INIT_REFERENCE, Wt.INIT_PARITY])
def test_inventory_matches_python(lib, cfg, scheme):
    specs = Wt.depth_pro_param_specs(cfg, scheme)
    c, keep = _c_cfg(cfg)
    n = lib.md_param_inventory(C.byref(c), scheme, -1, None, None, None, None)
    assert n == len(specs)
    for i, s in enumerate(specs):
        name, cnt, lo, hi = C.c_char_p(), C.c_size_t(), C.c_float(), C.c_float()
        lib.md_param_inventory(C.byref(c), scheme, i, C.byref(name), C.byref(cnt), C.byref(lo), C.byref(hi))
        assert name.value.decode() == s.name
        assert cnt.value == int(np.prod(s.shape))
        assert lo.value == np.float32(s.lo) and hi.value == np.float32(s.hi), s.name


@pytest.mark.parametrize("scheme", [Wt.INIT_REFERENCE, Wt.INIT_PARITY])
@pytest.mark.parametrize("variant", ["metric_large", "tiny", "small", "tiny_dual"])
def test_da3_inventory_matches_python(lib, variant, scheme):
    from burn_depth_amd.config import DepthAnything3Config
    cfg = {"metric_large": DepthAnything3Config.metric_large, "tiny": DepthAnything3Config.tiny_test,
           "small": DepthAnything3Config.small, "tiny_dual": DepthAnything3Config.tiny_dual_test}[variant]()
    specs = Wt.da3_param_specs(cfg, scheme)
    c = _lib.MdDa3Cfg(cfg.variant.encode(), 0, 0, 1, 1e-6)
    n = lib.md_da3_param_inventory(C.byref(c), scheme, -1, None, None, None, None)
    assert n == len(specs)
    for i, s in enumerate(specs):
        name, cnt, lo, hi = C.c_char_p(), C.c_size_t(), C.c_float(), C.c_float()
        lib.md_da3_param_inventory(C.byref(c), scheme, i, C.byref(name), C.byref(cnt), C.byref(lo), C.byref(hi))
        assert name.value.decode() == s.name
        assert cnt.value == int(np.prod(s.shape))
        assert lo.value == np.float32(s.lo) and hi.value == np.float32(s.hi), s.name


def test_da3_unknown_variant_is_an_error(lib):
    c = _lib.MdDa3Cfg(b"giant", 0, 0, 1, 1e-6)
    assert lib.md_da3_param_inventory(C.byref(c), 0, -1, None, None, None, None) == _lib.MD_ERR_INVALID_ARG


def test_unknown_preset_is_an_error_not_a_panic(lib):
    # layers/vit.rs:49-50 panics on an unknown preset; the ABI returns MD_ERR_INVALID_ARG
    cfg = DepthProConfig()
    cfg.patch_encoder_preset = "vit_h_14"
    c, keep = _c_cfg(cfg)
    assert lib.md_param_inventory(C.byref(c), 0, -1, None, None, None, None) == _lib.MD_ERR_INVALID_ARG
    assert b"unsupported ViT preset" in lib.md_last_error()


def test_seeded_generator_is_bit_identical(lib):
    for name, seed, n, lo, hi in [("encoder.patch_encoder.blocks.3.attn.qkv.weight", 0, 4099, -0.03125, 0.03125),
                                  ("head.conv_out.weight", 7, 32, 0.0, 0.08),
                                  ("fov.encoder.pos_embed", 123456789, 1000, -0.3, 0.3),
                                  ("x", 1, 17, 0.5, 1.5), ("const", 0, 5, 1.0, 1.0)]:
        got = np.empty(n, dtype=np.float32)
        assert lib.md_uniform_stream(name.encode(), seed, n, lo, hi, got.ctypes.data_as(C.c_void_p)) == 0
        want = Wt.uniform_stream(name, seed, n, lo, hi)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), name
        if lo != hi:
            assert (want >= np.float32(lo)).all() and (want <= np.float32(hi)).all()


def test_split_geometry_and_padding(lib):
    for size, win, ov in [(1536, 384, 0.25), (768, 384, 0.5), (512, 128, 0.25), (256, 128, 0.5), (512, 128, 0.0), (128, 128, 0.5)]:
        st, sp = C.c_int(), C.c_int()
        assert lib.md_split_geometry(size, win, C.c_float(ov), C.byref(st), C.byref(sp)) == 0
        assert (st.value, sp.value) == R.split_geometry(size, win, ov)
        for fs in (8, 24):
            assert lib.md_feature_padding(win, st.value, fs) == R.feature_padding(win, st.value, fs)


def test_resize_output_size(lib):
    for h, w, sh, sw in [(2, 2, 1.5, 0.5), (1536, 1536, 0.5, 0.25), (1, 3, 0.25, 0.25), (360, 540, 0.5, 0.5)]:
        oh, ow = C.c_int(), C.c_int()
        assert lib.md_op_resize_output_size(h, w, C.c_float(sh), C.c_float(sw), C.byref(oh), C.byref(ow)) == 0
        assert (oh.value, ow.value) == (R.compute_output_size(h, sh), R.compute_output_size(w, sw))


def test_fov_scalar_tail_matches_oracle(lib):
    # depth_pro/mod.rs:330-336 and 370-414
    for deg, h, w in [(56.4, 512, 512), (30.0, 360, 540), (100.0, 540, 360), (75.5, 1536, 1536)]:
        f, y = C.c_float(), C.c_float()
        assert lib.md_op_fov_to_focal(C.c_float(deg), h, w, C.byref(f), C.byref(y)) == 0
        fx = torch.tensor([deg], dtype=torch.float32) * torch.tensor(math.pi / 180.0, dtype=torch.float32)
        focal = (w * 0.5) / torch.tan(fx * 0.5)
        fovy = R.fovy_from_fovx_rad(fx, h, w)
        assert abs(f.value - focal.item()) <= 2e-6 * abs(focal.item())
        assert abs(y.value - fovy.item()) <= 1e-6


def test_load_missing_file_is_io_error(lib):
    # RecorderError path of DepthPro::load (depth_pro/mod.rs:193-208): reported before touching the GPU
    h = C.c_void_p()
    fake_dev = C.c_void_p(1)
    code = lib.md_depth_pro_load(fake_dev, b"/nonexistent/depth_pro.safetensors", C.byref(h))
    assert code == _lib.MD_ERR_IO
    assert b"cannot open" in lib.md_last_error()


def test_container_roundtrip(tmp_path):
    cfg = DepthProConfig.tiny_test()
    W = Wt.generate_depth_pro_weights(cfg, 3)
    for dt, tol in [("F32", 0.0), ("F16", 1e-3), ("BF16", 8e-3)]:
        p = str(tmp_path / f"w_{dt}.safetensors")
        Wt.save_container(p, W, Wt.config_metadata(cfg), dt)
        W2, meta = Wt.load_container(p)
        assert meta["model"] == "depth_pro" and set(W2) == set(W)
        k = "encoder.patch_encoder.blocks.0.attn.qkv.weight"
        assert W2[k].shape == W[k].shape
        assert np.abs(W2[k] - W[k]).max() <= tol * np.abs(W[k]).max() + (0 if tol else 0)


def test_tile_cost_model_choices(lib):
    """Host logic of the GEMM launch (gemm.hip::pick_tile): large launches go to the 256^2 kernel,
    launches of less than a round or two to the tile that fills the 256 CUs -- the measured best tiles of
    profiles/r03_tile_model.txt."""
    T256, T128, T256x32, T128x64, T64 = _lib.TILE_256x256, _lib.TILE_128x128, _lib.TILE_256x32, _lib.TILE_128x64, _lib.TILE_64x64
    bf16 = 0
    rows8 = 37 * 8 * 580   # Depth Pro at B = 8: 37 sequences of 580 rows per image
    rows1 = 37 * 580
    for (M, N, K) in [(rows8, 3072, 1024), (rows8, 4096, 1024), (rows8, 1024, 4096), (rows8, 1024, 1024),
                      (rows1, 3072, 1024), (rows1, 1024, 4096), (rows1, 1024, 1024), (8192, 8192, 8192),
                      (5477, 3072, 1024), (5477, 4096, 1024)]:
        assert lib.md_gemm_pick_tile(M, N, K, bf16) == T256, (M, N, K)
    # Depth-Anything-v3 small at 518^2 (1370 tokens, D = 384): 99 tiles of 128^2 would leave 157 CUs idle
    for (M, N, K) in [(1370, 1152, 384), (1370, 384, 384), (1370, 1536, 384), (1370, 384, 1536)]:
        assert lib.md_gemm_pick_tile(M, N, K, bf16) == T64, (M, N, K)
    # Depth-Anything-v3 large at 1036^2: proj / fc2 on 128x64 (measured 21.8 / 62.0 us against 25.2 / 68.2 on 128^2)
    for (M, N, K) in [(5477, 1024, 1024), (5477, 1024, 4096)]:
        assert lib.md_gemm_pick_tile(M, N, K, bf16) == T128x64, (M, N, K)
    assert lib.md_gemm_pick_tile(296 * 296, 64, 576, bf16) == T128x64   # 64-feature head convolution at 296^2 (as a dense shape)
    assert lib.md_gemm_pick_tile(1000, 32, 128, bf16) == T256x32
    assert lib.md_gemm_pick_tile(0, 32, 128, bf16) < 0                   # checked precondition, not a crash
    for prec in (0, 1, 2, 3, 4):                                          # every precision resolves to a tile that exists
        assert lib.md_gemm_pick_tile(700, 512, 256, prec) in (T256, T128, T128x64, T64)
