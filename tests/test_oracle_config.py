"""The oracle resolves presets through ITS OWN tables (oracle/ref_config.py, restated from the reference); the product has
its own (burn_depth_amd/config.py, csrc/md_weights.cpp). A wrong hook id / grid size / eps on one side must fail a test
instead of being wrong identically on both sides of every parity test."""
import dataclasses

import pytest

from burn_depth_amd import config as P
from oracle import ref_config as O


@pytest.mark.parametrize("preset", sorted(O.VIT_PRESETS))
def test_vit_presets_agree(preset):
    # reference: layers/vit.rs:23-43 (+ vitl, :54-56)
    o, p = O.vit_for(preset), P.vit_config_from_preset(preset)
    for f in ("in_chans", "embed_dim", "depth", "num_heads", "mlp_ratio", "img_size", "patch_size", "ln_eps"):
        assert getattr(o, f) == getattr(p, f), (preset, f)
    assert tuple(o.encoder_feature_layer_ids) == tuple(p.encoder_feature_layer_ids)
    assert tuple(o.encoder_feature_dims) == tuple(p.encoder_feature_dims)
    assert o.grid_size() == p.grid_size() and o.num_tokens == p.num_tokens and o.head_dim == p.head_dim == 64


def test_product_has_no_preset_the_oracle_lacks():
    assert set(P._PRESETS) == set(O.VIT_PRESETS)
    with pytest.raises(ValueError):  # vit.rs:49-50 panics on an unknown preset
        O.vit_for("dinov2b14_224")
    with pytest.raises(ValueError):
        P.vit_config_from_preset("dinov2b14_224")


def test_depth_pro_default_and_ci_configs_agree():
    # reference: depth_pro/mod.rs:54-66 and src/lib.rs:102-112
    for ref, cfg in ((O.DEPTH_PRO_DEFAULT, P.DepthProConfig()), (O.DEPTH_PRO_CI, P.DepthProConfig.small_test())):
        for k, v in ref.items():
            assert getattr(cfg, k) == v, k
        assert cfg.img_size() == O.img_size_for(cfg.patch_encoder_preset)
    assert P.InterpolationMethod.CUSTOM == O.INTERP_CUSTOM and P.InterpolationMethod.BURN == O.INTERP_BURN
    assert O.img_size_for("dinov2l16_384") == 1536 and O.img_size_for("dinov2l16_128") == 512


def test_oracle_follows_ln_eps_of_the_config():
    cfg = P.DepthProConfig()
    cfg.ln_eps = 1e-5
    assert O.vit_for(cfg.patch_encoder_preset, cfg.ln_eps).ln_eps == 1e-5 == cfg.patch_vit().ln_eps


@pytest.mark.parametrize("variant", ["metric_large", "small"])
def test_depth_anything3_variants_agree(variant):
    # reference: depth_anything3/mod.rs:139-171,179-199, dpt.rs:41-79
    cfg = P.DepthAnything3Config.metric_large() if variant == "metric_large" else P.DepthAnything3Config.small()
    assert cfg.image_size == O.DA3_VARIANTS[variant]["image_size"]
    O.check_da3(cfg)  # raises on any disagreement
    bad = dataclasses.replace(cfg, hook_block_ids=(4, 11, 17, 22))
    with pytest.raises(AssertionError):
        O.check_da3(bad)


def test_engine_inventory_uses_the_same_presets():
    """The C++ side has its own preset table (csrc/md_weights.cpp vit_dims_from_preset): its parameter inventory (host-only
    call, no GPU) must have the element counts the ORACLE's table implies."""
    import ctypes as C
    from burn_depth_amd import _lib
    from burn_depth_amd.depth_pro import _c_cfg
    lib = _lib.load()
    for cfg in (P.DepthProConfig(), P.DepthProConfig.small_test(), P.DepthProConfig.tiny_test()):
        v = O.vit_for(cfg.patch_encoder_preset)
        c, keep = _c_cfg(cfg)
        n = lib.md_param_inventory(C.byref(c), 0, -1, None, None, None, None)
        counts = {}
        for i in range(n):
            name, cnt = C.c_char_p(), C.c_size_t()
            lib.md_param_inventory(C.byref(c), 0, i, C.byref(name), C.byref(cnt), None, None)
            counts[name.value.decode()] = cnt.value
        assert counts["encoder.patch_encoder.pos_embed"] == v.num_tokens * v.embed_dim
        assert counts["encoder.patch_encoder.patch_embed.proj.weight"] == v.embed_dim * v.in_chans * v.patch_size ** 2
        assert counts[f"encoder.patch_encoder.blocks.{v.depth - 1}.mlp.fc1.weight"] == v.embed_dim * v.mlp_ratio * v.embed_dim
        assert f"encoder.patch_encoder.blocks.{v.depth}.mlp.fc1.weight" not in counts
        assert counts["encoder.upsample_latent0.projection.weight"] == v.encoder_feature_dims[0] * v.embed_dim
