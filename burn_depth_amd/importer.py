"""Upstream checkpoint -> engine weight container.

The reference ships two converter binaries, `tool/import_depth_pro.rs` and `tool/import_da3.rs`, that turn the
*upstream* PyTorch files (apple/ml-depth-pro `depth_pro.pt`, ByteDance DA3 `model.safetensors`) into Burn records.
This module does the same job for the engine's container (safetensors keyed by the Burn field paths,
`burn_depth_amd.weights.save_container`): upstream key -> Burn field path by the rename rules those tools
encode (`import_depth_pro.rs:344-437`, `import_da3.rs:67-195`), then a strict check against the parameter
inventory of the target configuration (every tensor present, exact shape, nothing unknown).

The rename rules are kept as data (`DEPTH_PRO_RULES`, `da3_rules()`): ordered `(regex, replacement)` pairs, every
matching rule is applied in order, as the reference's `KeyRemapper` does.
"""
from __future__ import annotations

import re
from typing import Dict, Iterable, List, Mapping, Sequence, Tuple

import numpy as np

from . import weights as Wt
from .config import DepthAnything3Config, DepthProConfig

Rule = Tuple[str, str]


def _norm_rules(prefix_regex: str) -> List[Rule]:
    """LayerNorm affine parameters: PyTorch weight/bias -> Burn gamma/beta."""
    return [(rf"^({prefix_regex})\.weight$", r"\1.gamma"), (rf"^({prefix_regex})\.bias$", r"\1.beta")]


def _sequential_rules(prefix: str, mapping: Mapping[int, str]) -> List[Rule]:
    """`nn.Sequential` child index -> named field, for weight and bias."""
    esc = re.escape(prefix)
    return [(rf"^{esc}\.{i}\.(weight|bias)$", rf"{prefix}.{name}.\1") for i, name in mapping.items()]


def _depth_pro_rules() -> List[Rule]:
    r: List[Rule] = []
    # ViT LayerNorms of the three encoders (import_depth_pro.rs:346-361); the FOV encoder sits at index 0 of a
    # Sequential whose index 1 is the token projection (:411-412)
    r += _norm_rules(r"encoder\.(?:patch_encoder|image_encoder)(?:\.blocks\.\d+)?\.norm\d?")
    r += _norm_rules(r"fov\.encoder(?:\.0)?(?:\.blocks\.\d+)?\.norm\d?")
    # ProjectUpsample blocks are Sequentials: 0 = 1x1 projection, 1.. = the k2s2 deconvs (:362-385)
    for k in range(3):
        r += _sequential_rules(f"encoder.upsample{k}", {0: "projection", 1: "upsample.0"})
    for k in range(2):
        r += _sequential_rules(f"encoder.upsample_latent{k}", {0: "projection", 1: "upsample.0", 2: "upsample.1", 3: "upsample.2"})
    r += [(r"^fov\.downsample\.(\d+)\.(weight|bias)$", r"fov.downsample_blocks.\1.conv.\2"),
          (r"^decoder\.convs\.(\d+)\.(weight|bias)$", r"decoder.convs.\1.conv.\2"),
          # ResidualBlock = Sequential(ReLU, conv, ReLU, conv) -> conv1 / conv2 (:398-405)
          (r"^(decoder\.fusions\.\d+\.resnet[12])\.residual\.1\.(weight|bias)$", r"\1.conv1.\2"),
          (r"^(decoder\.fusions\.\d+\.resnet[12])\.residual\.3\.(weight|bias)$", r"\1.conv2.\2"),
          (r"^fov\.encoder\.0\.", "fov.encoder."),
          (r"^fov\.encoder\.1\.(weight|bias)$", r"fov.encoder_proj.\1")]
    # depth head Sequential(conv, deconv, conv, ReLU, conv, ReLU) (:413-416); FOV head Sequential with ReLUs
    r += _sequential_rules("head", {0: "conv0", 1: "deconv", 2: "conv1", 4: "conv_out"})
    r += [(rf"^fov\.head\.{i}\.(weight|bias)$", rf"fov.head_blocks.{j}.conv.\1") for j, i in enumerate((0, 2, 4))]
    return r


DEPTH_PRO_RULES: List[Rule] = _depth_pro_rules()
# upstream buffers with no counterpart in the Burn module (import_depth_pro.rs:439-445 lists them as allowed
# to be absent on the Burn side; when present upstream they are dropped)
DEPTH_PRO_IGNORED = (r"\.mask_token$", r"\.num_batches_tracked$", r"\.register_tokens$")


def da3_rules(head_prefix: str = "head_mono") -> List[Rule]:
    """import_da3.rs:67-195: `head_mono` for metric_large, `head_dual` (+ camera decoder, aux heads) for small."""
    hp = re.escape(head_prefix)
    r: List[Rule] = [(r"^model\.", ""), (r"^head\.", f"{head_prefix}.")]
    # camera decoder Sequentials (import_da3.rs:70-88)
    r += [(r"^cam_dec\.backbone\.0\.(weight|bias)$", r"camera_decoder.backbone_1.\1"),
          (r"^cam_dec\.backbone\.2\.(weight|bias)$", r"camera_decoder.backbone_2.\1"),
          (r"^cam_dec\.fc_fov\.0\.(weight|bias)$", r"camera_decoder.fc_fov.\1"),
          (r"^cam_dec\.", "camera_decoder."), (r"^cam_enc\.", "camera_encoder.")]
    r += _norm_rules(r"backbone\.pretrained\..*\.norm\d+")
    r += _norm_rules(r"backbone\.pretrained\.norm")
    r += _norm_rules(r"backbone\.pretrained\..*\.attn\.[qk]_norm")
    r += _norm_rules(rf"{hp}\..*norm\d*")
    r += [(rf"^({hp}\.resize_layers\.[01])\.(weight|bias)$", r"\1.conv_t.\2"),
          (rf"^({hp}\.resize_layers\.3)\.(weight|bias)$", r"\1.conv.\2"),
          (rf"^({hp}\.scratch\.output_conv2)\.0\.(weight|bias)$", r"\1.conv1.\2"),
          (rf"^({hp}\.scratch\.output_conv2)\.2\.(weight|bias)$", r"\1.conv2.\2"),
          (rf"^({hp}\.scratch\.refinenet\d+(?:_aux)?)\.resConfUnit([12])\.", r"\1.residual\2."),
          # aux heads (import_da3.rs:146-178): pre-head Sequential of convs; output head Sequential
          # (0 reduce conv, 2 LayerNorm2d when present, 5 project conv)
          (rf"^({hp}\.scratch\.output_conv1_aux\.\d+)\.(\d+)\.(weight|bias)$", r"\1.layers.\2.\3"),
          (rf"^({hp}\.scratch\.output_conv2_aux\.\d+)\.0\.(weight|bias)$", r"\1.reduce.\2"),
          (rf"^({hp}\.scratch\.output_conv2_aux\.\d+)\.2\.weight$", r"\1.norm.layer_norm.gamma"),
          (rf"^({hp}\.scratch\.output_conv2_aux\.\d+)\.2\.bias$", r"\1.norm.layer_norm.beta"),
          (rf"^({hp}\.scratch\.output_conv2_aux\.\d+)\.5\.(weight|bias)$", r"\1.project.\2")]
    # camera encoder (import_da3.rs:184-195): trunk norm1/norm2, token_norm, trunk_norm
    r += _norm_rules(r"camera_encoder\..*norm\d*")
    return r


# a metric_large checkpoint has no camera encoder in the Burn module (mod.rs:139-156: `camera_encoder: None`); the importer
# drops upstream `cam_enc.*` tensors there, like the reference's `allow_partial(true)` store (import_da3.rs:199-202)
DA3_IGNORED = (r"\.mask_token$",)
DA3_IGNORED_NO_ENCODER = DA3_IGNORED + (r"^camera_encoder\.",)


class ImportError_(ValueError):
    """Raised with the full list of problems (missing / unexpected / mis-shaped tensors)."""


def remap_key(key: str, rules: Sequence[Rule]) -> str:
    for pat, rep in rules:
        key = re.sub(pat, rep, key)
    return key


def _to_numpy(t) -> np.ndarray:
    if isinstance(t, np.ndarray):
        return t
    import torch
    if isinstance(t, torch.Tensor):
        t = t.detach().cpu()
        if t.dtype in (torch.bfloat16, torch.float16):
            t = t.float()
        return t.numpy()
    raise TypeError(f"unsupported tensor type {type(t)}")


def convert_state_dict(state: Mapping[str, object], specs: Iterable, rules: Sequence[Rule],
                       ignored: Sequence[str] = (), allow_missing: Sequence[str] = ()) -> Dict[str, np.ndarray]:
    """Renames `state` and checks it against `specs` (objects with `.name` and `.shape`). Returns fp32 arrays
    keyed by Burn field path. ConvTranspose2d weights stay `[Cin, Cout, kh, kw]` (PyTorch layout == Burn
    layout, `depth_pro/mod.rs:416-431`); Linear weights stay `[out, in]` as the engine packs them."""
    want = {s.name: tuple(s.shape) for s in specs}
    out: Dict[str, np.ndarray] = {}
    problems: List[str] = []
    for k, v in state.items():
        if any(re.search(p, k) for p in ignored):
            continue
        nk = remap_key(k, rules)
        if any(re.search(p, nk) for p in ignored):
            continue
        if nk not in want:
            problems.append(f"unexpected tensor `{k}` (-> `{nk}`)")
            continue
        a = _to_numpy(v)
        if tuple(a.shape) != want[nk]:
            # scalars saved with a leading singleton / conv 1x1 saved as a matrix are the two benign cases
            if int(np.prod(a.shape)) == int(np.prod(want[nk])) and _squeezed(a.shape) == _squeezed(want[nk]):
                a = a.reshape(want[nk])
            else:
                problems.append(f"`{nk}` has shape {tuple(a.shape)}, expected {want[nk]}")
                continue
        if nk in out:
            problems.append(f"two upstream tensors map to `{nk}`")
            continue
        out[nk] = np.ascontiguousarray(a, dtype=np.float32)
    for name in want:
        if name not in out and not any(re.search(p, name) for p in allow_missing):
            problems.append(f"missing tensor `{name}`")
    if problems:
        raise ImportError_("checkpoint does not match the configuration:\n  " + "\n  ".join(problems[:40])
                           + (f"\n  ... and {len(problems) - 40} more" if len(problems) > 40 else ""))
    return out


def _squeezed(shape) -> Tuple[int, ...]:
    return tuple(int(d) for d in shape if int(d) != 1)


def load_upstream(path: str) -> Dict[str, object]:
    """Reads an upstream checkpoint: `.safetensors`, or a PyTorch zip-pickle (`.pt` / `.pth`) holding a state
    dict (optionally under a `state_dict` / `model` key). Pickles are loaded with `weights_only=True`."""
    if path.endswith(".safetensors"):
        tensors, _ = Wt.load_container(path)
        return tensors
    if path.endswith(".mpk"):  # a Burn record: names are already the Burn field paths (rules are no-ops on them)
        from .mpk import read_mpk
        return read_mpk(path)
    import torch
    obj = torch.load(path, map_location="cpu", weights_only=True)
    for k in ("state_dict", "model"):
        if isinstance(obj, dict) and k in obj and isinstance(obj[k], dict):
            obj = obj[k]
    if not isinstance(obj, dict):
        raise ImportError_(f"{path}: expected a state dict, found {type(obj).__name__}")
    return obj


def import_depth_pro(src: str, dst: str, cfg: DepthProConfig | None = None, dtype: str = "F16") -> Dict[str, np.ndarray]:
    """`depth_pro.pt` -> container readable by `DepthPro.load` / `md_depth_pro_load`. `dtype` F16 mirrors the
    reference's half-precision records (`mod.rs:206`); F32 keeps the upstream values bit-exactly."""
    cfg = cfg or DepthProConfig()
    specs = Wt.depth_pro_param_specs(cfg, Wt.INIT_REFERENCE)
    tensors = convert_state_dict(load_upstream(src), specs, DEPTH_PRO_RULES, DEPTH_PRO_IGNORED)
    Wt.save_container(dst, tensors, metadata=Wt.config_metadata(cfg), dtype=dtype)
    return tensors


def import_da3(src: str, dst: str, cfg: DepthAnything3Config | None = None, dtype: str = "F16") -> Dict[str, np.ndarray]:
    """DA3 `model.safetensors` (metric_large: mono head; small: dual head + camera decoder) -> container readable
    by `DepthAnything3.load_file`."""
    cfg = cfg or DepthAnything3Config()
    specs = Wt.da3_param_specs(cfg, Wt.INIT_REFERENCE)
    tensors = convert_state_dict(load_upstream(src), specs, da3_rules("head_dual" if cfg.dual_head else "head_mono"),
                                 DA3_IGNORED if cfg.camera_encoder else DA3_IGNORED_NO_ENCODER)
    Wt.save_container(dst, tensors, metadata={"model": "depth_anything3", "variant": cfg.variant,
                                              "image_size": str(cfg.image_size)}, dtype=dtype)
    return tensors
