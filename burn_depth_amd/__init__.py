"""burn_depth_amd -- MI355X-native (gfx950) drop-in for the Depth Pro hot path of mosure/burn_depth.

Everything that computes lives in ``libmi_depth.so`` (hand-written HIP kernels behind the C ABI of
``include/mi_depth.h``); this package is the thin host-side mirror of the reference interface.
Importing ``config`` / ``weights`` needs no GPU and no library; ``depth_pro`` / ``ops`` load the
library and fail loudly when it has not been built.
"""
from .config import DepthProConfig, InterpolationMethod, Precision, ViTConfig, vit_config_from_preset  # noqa: F401

__all__ = ["DepthProConfig", "InterpolationMethod", "Precision", "ViTConfig", "vit_config_from_preset"]
