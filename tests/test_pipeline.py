"""Host pre/post-processing around the hot path (burn_depth_amd/pipeline.py; example/inference.rs, src/model/mod.rs)."""
import numpy as np
import pytest

from burn_depth_amd import pipeline as P


def test_prepare_da3_geometry_matches_the_reference_arithmetic():
    # src/model/mod.rs:176-199 on the repository's test image size (540x360): scale 518/360, 777x518, crop x=129
    rng = np.random.default_rng(0)
    rgb = rng.integers(0, 256, (360, 540, 3), dtype=np.uint8)
    out = P.prepare_depth_anything3_image(rgb, 518)
    assert (out.width, out.height) == (518, 518) and out.rgb.shape == (518, 518, 3) and out.crop is None
    full = P.resize_catmull_rom(rgb, 777, 518)
    assert np.array_equal(out.rgb, full[0:518, 129:129 + 518])
    same = P.prepare_depth_anything3_image(rgb[:200, :200], 200)
    assert np.array_equal(same.rgb, rgb[:200, :200])                     # already target-sized: untouched
    with pytest.raises(ValueError):
        P.prepare_depth_anything3_image(rgb, 0)


def test_catmull_rom_resampler_properties():
    const = np.full((37, 53, 3), 117, np.uint8)
    assert (P.resize_catmull_rom(const, 80, 29) == 117).all()           # weights are normalised
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (16, 20, 3), dtype=np.uint8)
    assert np.array_equal(P.resize_catmull_rom(img, 20, 16), img)        # same size: copy
    ramp = np.tile(np.arange(0, 200, 4, dtype=np.uint8)[None, :, None], (8, 1, 3))
    up = P.resize_catmull_rom(ramp, 100, 8).astype(int)
    assert (np.diff(up[0, 4:-4, 0]) >= 0).all() and abs(int(up[0, 50, 0]) - 100) <= 3  # a linear ramp stays linear
    k = P._catmull_rom(np.array([0.0, 0.5, 1.0, 1.5, 2.0, 2.5], np.float32))
    assert np.allclose(k, [1.0, 0.5625, 0.0, -0.0625, 0.0, 0.0], atol=1e-6)


def test_resize_depth_field_follows_the_reference_sampler():
    v = np.array([[0.0, 1.0], [2.0, 3.0]], np.float32)
    out = P.resize_depth_field(v, 4, 4)
    # centres at -0.25, 0.25, 0.75, 1.25 on both axes; the fraction is taken against the CLAMPED x0, so -0.25
    # extrapolates (x0 = 0, fx = -0.25) and 1.25 collapses onto the last sample (x0 = x1 = 1); worked by hand:
    want = np.array([[-0.75, -0.25, 0.25, 0.5], [0.25, 0.75, 1.25, 1.5], [1.25, 1.75, 2.25, 2.5], [1.75, 2.25, 2.75, 3.0]],
                    np.float32)
    assert np.allclose(out, want, atol=1e-6)
    assert np.array_equal(P.resize_depth_field(v, 2, 2), v)
    assert P.resize_depth_field(v, 1, 1).shape == (1, 1) and P.resize_depth_field(v, 1, 1)[0, 0] == 0.0  # scale 0 -> src (0,0)


def test_depth_normalisation_and_png_round_trip(tmp_path):
    d = np.array([[[1.0, 2.0, np.inf], [3.0, np.nan, 5.0]]], np.float32)
    px = P.depth_to_u8(d)
    assert px.tolist() == [[0, 64, 0], [128, 0, 255]]                   # (v-1)/4*255 rounded; non-finite -> 0
    assert (P.depth_to_u8(np.full((1, 2, 2), np.nan, np.float32)) == 0).all()
    assert (P.depth_to_u8(np.full((1, 2, 2), 7.0, np.float32)) == 0).all()  # zero range -> epsilon, all pixels 0
    with pytest.raises(ValueError, match="batch size of 1"):
        P.depth_to_u8(np.zeros((2, 2, 2), np.float32))
    with pytest.raises(ValueError, match="exceeds depth tensor bounds"):
        P.crop_depth_field(np.zeros((4, 4), np.float32), P.ImageCropRegion(2, 2, 3, 1))
    big = np.linspace(0.5, 9.5, 12 * 10, dtype=np.float32).reshape(1, 12, 10)
    path = str(tmp_path / "sub" / "depth.png")
    out = P.save_depth_map(big, path, crop=P.ImageCropRegion(1, 2, 8, 8), target_dims=(16, 12))
    assert out.shape == (12, 16) and out.min() == 0 and out.max() == 255
    assert np.array_equal(P.read_gray_png(path), out)


def test_any_depth_model_kind_strings():
    assert P.DepthModelKind.DEPTH_PRO.value == "depth-pro" and P.DepthModelKind.DEPTH_ANYTHING3.value == "depth-anything-3"
