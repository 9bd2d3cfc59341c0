"""The oracle's OWN configuration tables -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restated from the reference so that a wrong preset in the product's `burn_depth_amd.config` (hook ids, grid size, LayerNorm
eps, ...) cannot be wrong identically on both sides of a parity test: the oracle resolves every preset NAME through the
tables below, and `tests/test_oracle_config.py` asserts that they equal the product's.

* ViT presets            -- /root/reference/src/model/depth_pro/layers/vit.rs:19-43 (fields) + :54-56 (`vitl`: DINOv2 ViT-L)
* DepthProConfig default -- /root/reference/src/model/depth_pro/mod.rs:54-66
* CI test config         -- /root/reference/src/lib.rs:102-112
* InterpolationMethod    -- /root/reference/src/model/depth_pro/interpolate.rs:11-22
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

INTERP_CUSTOM, INTERP_BURN = 0, 1   # interpolate.rs:11-22 (Custom = the default: PyTorch align_corners=False)

# `DinoVisionTransformerConfig::vitl` (burn_dino 0.6.0, called at vit.rs:54-56) = the public DINOv2 ViT-L definition
VITL = dict(embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4)


@dataclass(frozen=True)
class RefViT:
    name: str
    in_chans: int
    embed_dim: int
    depth: int
    num_heads: int
    mlp_ratio: int
    img_size: int
    patch_size: int
    encoder_feature_layer_ids: Tuple[int, ...]
    encoder_feature_dims: Tuple[int, ...]
    ln_eps: float = 1e-6  # burn_dino's LayerNorm eps is not visible in the reference tree: upstream DINOv2's 1e-6

    def grid_size(self) -> int:  # vit.rs:14-18
        return self.img_size // self.patch_size

    @property
    def num_tokens(self) -> int:
        return self.grid_size() ** 2 + 1

    @property
    def head_dim(self) -> int:
        return self.embed_dim // self.num_heads


VIT_PRESETS = {
    # vit.rs:25-32
    "dinov2l16_384": RefViT("dinov2l16_384", 3, VITL["embed_dim"], VITL["depth"], VITL["num_heads"], VITL["mlp_ratio"], 384, 16,
                            (5, 11, 17, 23), (256, 512, 1024, 1024)),
    # vit.rs:33-40
    "dinov2l16_128": RefViT("dinov2l16_128", 3, VITL["embed_dim"], VITL["depth"], VITL["num_heads"], VITL["mlp_ratio"], 128, 16,
                            (5, 11, 17, 23), (256, 512, 1024, 1024)),
    # NOT in the reference: the build's seconds-on-a-CPU preset (same topology, 4 blocks of width 256, head_dim 64)
    "tiny16_128": RefViT("tiny16_128", 3, 256, 4, 4, 4, 128, 16, (1, 2, 3, 3), (64, 128, 256, 256)),
}

# mod.rs:54-66
DEPTH_PRO_DEFAULT = dict(patch_encoder_preset="dinov2l16_384", image_encoder_preset="dinov2l16_384", decoder_features=256,
                         checkpoint_uri=None, fov_encoder_preset="dinov2l16_384", use_fov_head=True, interpolation=INTERP_CUSTOM)
# src/lib.rs:102-112
DEPTH_PRO_CI = dict(DEPTH_PRO_DEFAULT, patch_encoder_preset="dinov2l16_128", image_encoder_preset="dinov2l16_128",
                    fov_encoder_preset="dinov2l16_128", decoder_features=64)


def vit_for(preset: Optional[str], ln_eps: Optional[float] = None) -> Optional[RefViT]:
    """vit.rs:23-43,49-50: an unknown preset is a panic (ValueError here); None stays None (fov.rs:118: no FOV ViT)."""
    if preset is None:
        return None
    try:
        v = VIT_PRESETS[preset]
    except KeyError:
        raise ValueError(f"unsupported ViT preset `{preset}`") from None
    if ln_eps is not None and ln_eps != v.ln_eps:
        v = RefViT(**{**v.__dict__, "ln_eps": float(ln_eps)})
    return v


def img_size_for(patch_preset: str) -> int:
    """encoder.rs:139-140: the network input is 4 x the patch window."""
    return vit_for(patch_preset).img_size * 4


# ---------------------------------------------------------------------------------------------
# Depth-Anything-v3 (src/model/depth_anything3/mod.rs:124-171,179-199; dpt.rs:15-79)
# ---------------------------------------------------------------------------------------------
# `DinoVisionTransformerConfig::vits` (mod.rs:184-186: chosen when head.dim_in < 1024) = the public DINOv2 ViT-S definition
VITS = dict(embed_dim=384, depth=12, num_heads=6, mlp_ratio=4)

DA3_VARIANTS = {
    # DepthAnything3Config::default / metric_large (mod.rs:139-156) + DepthAnything3HeadConfig::metric_large (dpt.rs:41-58)
    "metric_large": dict(image_size=518, patch_size=14, hook_block_ids=(4, 11, 17, 23), dim_in=1024, features=256,
                         out_channels=(256, 512, 1024, 1024), output_dim=1, pos_embed=True, dual_head=False, aux_levels=4,
                         aux_out1_conv_num=5, aux_output_dim=7, vit=VITL, ext_block_start=-1, camera_encoder=None),
    # DepthAnything3Config::small (mod.rs:158-171) + DepthAnything3HeadConfig::small (dpt.rs:60-79); the backbone extras start
    # at block 4 (alt / qk_norm / rope block_start, mod.rs:190-194)
    "small": dict(image_size=518, patch_size=14, hook_block_ids=(5, 7, 9, 11), dim_in=768, features=64,
                  out_channels=(48, 96, 192, 384), output_dim=2, pos_embed=True, dual_head=True, aux_levels=4,
                  aux_out1_conv_num=5, aux_output_dim=7, vit=VITS, ext_block_start=4,
                  # CameraEncoderConfig { dim_out: 384, ..default } (mod.rs:164-167; camera.rs:25-37): trunk_depth 4, 16 heads, mlp 4
                  camera_encoder=dict(heads=16, trunk_depth=4)),
}


def check_da3(cfg) -> None:
    """The oracle refuses a product-side DepthAnything3Config whose named variant disagrees with the table above (the
    test-only reductions `tiny` / `tiny_dual` are not in the reference and pass through)."""
    ref = DA3_VARIANTS.get(getattr(cfg, "variant", None))
    if ref is None:
        return
    v = cfg.vit()
    got = dict(patch_size=cfg.patch_size, hook_block_ids=tuple(cfg.hook_block_ids), dim_in=cfg.dim_in, features=cfg.features,
               out_channels=tuple(cfg.out_channels), output_dim=cfg.output_dim, pos_embed=cfg.pos_embed, dual_head=cfg.dual_head,
               aux_levels=cfg.aux_levels, aux_out1_conv_num=cfg.aux_out1_conv_num, aux_output_dim=cfg.aux_output_dim,
               ext_block_start=cfg.ext_block_start,
               camera_encoder=dict(heads=cfg.cam_heads, trunk_depth=cfg.cam_trunk_depth) if cfg.camera_encoder else None,
               vit=dict(embed_dim=v.embed_dim, depth=v.depth, num_heads=v.num_heads, mlp_ratio=v.mlp_ratio))
    for k, want in ref.items():
        if k == "image_size":
            continue  # any multiple of the patch size is a legal input (mod.rs:509-520)
        if got[k] != want:
            raise AssertionError(f"DepthAnything3Config `{cfg.variant}`: {k} = {got[k]!r}, the reference has {want!r}")
