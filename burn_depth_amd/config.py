"""Configuration objects for the Depth Pro hot path.

Mirrors the reference's configuration surface:

* ``ViTConfig`` / presets      -- /root/reference/src/model/depth_pro/layers/vit.rs:4-43
* ``DepthProConfig``           -- /root/reference/src/model/depth_pro/mod.rs:35-66
* ``InterpolationMethod``      -- /root/reference/src/model/depth_pro/interpolate.rs:11-22

The reference hard-wires ViT-L (burn_dino ``DinoVisionTransformerConfig::vitl``) for both
of its presets; this engine carries the transformer dimensions explicitly so a small
test-only preset (``tiny16_128``) can run the *same* code path in seconds on a CPU oracle.
"""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass, field
from typing import List, Optional

DINOV2_L16_384 = "dinov2l16_384"
DINOV2_L16_128 = "dinov2l16_128"
TINY16_128 = "tiny16_128"  # test-only preset, not in the reference


class InterpolationMethod:
    """reference: depth_pro/interpolate.rs:11-22. ``CUSTOM`` = PyTorch align_corners=False
    (the reference default); ``BURN`` = Burn ``module::interpolate`` = align_corners=True."""

    CUSTOM = 0
    BURN = 1


class Precision:
    """Arithmetic type of the MFMA operands (accumulation is always fp32)."""

    BF16 = 0
    F32 = 1
    FP8 = 2  # Depth-Anything-v3 only: e4m3 operands for the four ViT linear layers, bf16 elsewhere (BASELINE config 5)
    F16 = 3  # IEEE half operands (the reference's checkpoint type, mod.rs:206): bf16's rate, 3 more mantissa bits
    F16X2 = 4  # Depth Pro: activations as two half planes (hi + lo, 22 bits), f16 weights exact: the accurate FAST mode


@dataclass(frozen=True)
class ViTConfig:
    name: str
    in_chans: int
    embed_dim: int
    depth: int
    num_heads: int
    mlp_ratio: int
    img_size: int
    patch_size: int
    encoder_feature_layer_ids: tuple
    encoder_feature_dims: tuple
    ln_eps: float = 1e-6  # burn_dino's value is not visible: config field

    def grid_size(self) -> int:
        return self.img_size // self.patch_size

    @property
    def num_tokens(self) -> int:
        return self.grid_size() ** 2 + 1

    @property
    def head_dim(self) -> int:
        return self.embed_dim // self.num_heads


_PRESETS = {
    # vit.rs:25-32
    DINOV2_L16_384: ViTConfig(DINOV2_L16_384, 3, 1024, 24, 16, 4, 384, 16,
                              (5, 11, 17, 23), (256, 512, 1024, 1024)),
    # vit.rs:33-40
    DINOV2_L16_128: ViTConfig(DINOV2_L16_128, 3, 1024, 24, 16, 4, 128, 16,
                              (5, 11, 17, 23), (256, 512, 1024, 1024)),
    # test-only: same topology, 4 blocks of width 256 (head_dim stays 64)
    TINY16_128: ViTConfig(TINY16_128, 3, 256, 4, 4, 4, 128, 16,
                          (1, 2, 3, 3), (64, 128, 256, 256)),
}


def vit_config_from_preset(preset: str) -> ViTConfig:
    """reference: vit.rs:23-43 (panics on unknown preset -> ValueError here)."""
    try:
        return _PRESETS[preset]
    except KeyError:
        raise ValueError(f"unsupported ViT preset `{preset}`") from None


@dataclass
class DepthProConfig:
    """reference: depth_pro/mod.rs:35-66 (same field names and defaults)."""

    patch_encoder_preset: str = DINOV2_L16_384
    image_encoder_preset: str = DINOV2_L16_384
    decoder_features: int = 256
    checkpoint_uri: Optional[str] = None
    fov_encoder_preset: Optional[str] = DINOV2_L16_384
    use_fov_head: bool = True
    interpolation: int = InterpolationMethod.CUSTOM
    # engine-side additions (not in the reference)
    precision: int = Precision.BF16
    max_batch: int = 1
    ln_eps: float = 1e-6

    @staticmethod
    def small_test() -> "DepthProConfig":
        """reference: src/lib.rs:102-112 (128-window preset, decoder 64)."""
        return DepthProConfig(DINOV2_L16_128, DINOV2_L16_128, 64, None, DINOV2_L16_128)

    @staticmethod
    def tiny_test() -> "DepthProConfig":
        return DepthProConfig(TINY16_128, TINY16_128, 64, None, TINY16_128)

    def patch_vit(self) -> ViTConfig:
        return dataclasses.replace(vit_config_from_preset(self.patch_encoder_preset), ln_eps=self.ln_eps)

    def image_vit(self) -> ViTConfig:
        return dataclasses.replace(vit_config_from_preset(self.image_encoder_preset), ln_eps=self.ln_eps)

    def fov_vit(self) -> Optional[ViTConfig]:
        if self.fov_encoder_preset is None:
            return None
        return dataclasses.replace(vit_config_from_preset(self.fov_encoder_preset), ln_eps=self.ln_eps)

    def img_size(self) -> int:
        """reference: encoder.rs:139-140 (img_size = 4 * patch window)."""
        return self.patch_vit().img_size * 4


# ---------------------------------------------------------------------------------------------
# Depth-Anything-v3 (reference: src/model/depth_anything3/mod.rs:124-172, dpt.rs:15-80)
# ---------------------------------------------------------------------------------------------
DA3_VITL14 = ViTConfig("da3_vitl14", 3, 1024, 24, 16, 4, 518, 14, (4, 11, 17, 23), (256, 512, 1024, 1024))
DA3_TINY14 = ViTConfig("da3_tiny14", 3, 256, 4, 4, 4, 70, 14, (0, 1, 2, 3), (64, 128, 256, 256))  # test-only
DA3_VITS14 = ViTConfig("da3_vits14", 3, 384, 12, 6, 4, 518, 14, (5, 7, 9, 11), (48, 96, 192, 384))
DA3_TINYDUAL14 = ViTConfig("da3_tinydual14", 3, 128, 6, 2, 4, 70, 14, (2, 3, 4, 5), (48, 96, 64, 128))  # test-only


@dataclass
class DepthAnything3Config:
    """`DepthAnything3Config::metric_large()` (depth_anything3/mod.rs:153-156) + head
    (`DepthAnything3HeadConfig::metric_large`, dpt.rs:41-58). `metric_large` = plain ViT-L/14 + mono
    head; `small` = ViT-S/14 with the burn_dino extras + dual head + camera decoder."""

    variant: str = "metric_large"
    image_size: int = 518   # input rows (and columns when image_width == 0)
    patch_size: int = 14
    hook_block_ids: tuple = (4, 11, 17, 23)
    dim_in: int = 1024
    features: int = 256
    out_channels: tuple = (256, 512, 1024, 1024)
    output_dim: int = 1
    pos_embed: bool = True
    precision: int = Precision.BF16
    max_batch: int = 1
    ln_eps: float = 1e-6
    # `small` (mod.rs:158-171, 190-196; dpt.rs:60-79): dual head + camera decoder on a ViT whose blocks
    # >= `ext_block_start` use QK-norm, 2-D RoPE and alternate local/global attention, with the cls slot
    # replaced by a learned camera token at that block and hooks = cat(last local x, norm(x)) (dim 2*D)
    dual_head: bool = False
    ext_block_start: int = -1
    rope_frequency: float = 100.0
    qk_norm_eps: float = 1e-5
    aux_out1_conv_num: int = 5
    aux_output_dim: int = 7
    aux_levels: int = 4
    image_width: int = 0    # 0 = square; else input columns (multiple of 14): `infer` only asserts divisibility (mod.rs:509-520)
    # `CameraEncoderConfig` (camera.rs:12-37; mod.rs:164-168 sets dim_out = embed_dim): only runs under `infer_with_camera`
    camera_encoder: bool = False
    cam_heads: int = 16
    cam_trunk_depth: int = 4
    cam_ln_eps: float = 1e-5   # token_norm / trunk_norm: Burn's `LayerNormConfig::new` default (camera.rs:84-85)

    @staticmethod
    def metric_large() -> "DepthAnything3Config":
        return DepthAnything3Config()

    @staticmethod
    def tiny_test() -> "DepthAnything3Config":
        return DepthAnything3Config("tiny", 70, 14, (0, 1, 2, 3), 256, 64, (64, 128, 256, 256))

    @staticmethod
    def small() -> "DepthAnything3Config":
        """`DepthAnything3Config::small()` (mod.rs:158-171) + `DepthAnything3HeadConfig::small()` (dpt.rs:60-79)."""
        return DepthAnything3Config("small", 518, 14, (5, 7, 9, 11), 768, 64, (48, 96, 192, 384), 2,
                                    dual_head=True, ext_block_start=4, camera_encoder=True)

    @staticmethod
    def tiny_dual_test() -> "DepthAnything3Config":
        return DepthAnything3Config("tiny_dual", 70, 14, (2, 3, 4, 5), 256, 64, (48, 96, 64, 128), 2,
                                    dual_head=True, ext_block_start=2, camera_encoder=True, cam_trunk_depth=2)

    def vit(self) -> ViTConfig:
        base = {"metric_large": DA3_VITL14, "small": DA3_VITS14, "tiny_dual": DA3_TINYDUAL14}.get(self.variant, DA3_TINY14)
        return dataclasses.replace(base, ln_eps=self.ln_eps, encoder_feature_layer_ids=tuple(self.hook_block_ids))

    def img_size(self) -> int:
        return self.image_size
