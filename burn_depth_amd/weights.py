"""Depth Pro parameter inventory, seeded synthetic initialisation and weight container.

Parameter names are the reference's Burn field paths after the importer's key remap
(/root/reference/tool/import_depth_pro.rs:344-437):

* ViT (timm names with ``norm*.weight/bias -> gamma/beta``):
  ``<vit>.patch_embed.proj.{weight,bias}``, ``<vit>.cls_token``, ``<vit>.pos_embed``,
  ``<vit>.blocks.N.{norm1,norm2}.{gamma,beta}``, ``<vit>.blocks.N.attn.{qkv,proj}.{weight,bias}``,
  ``<vit>.blocks.N.{ls1,ls2}.gamma``, ``<vit>.blocks.N.mlp.{fc1,fc2}.{weight,bias}``,
  ``<vit>.norm.{gamma,beta}`` with ``<vit>`` in {encoder.patch_encoder, encoder.image_encoder,
  fov.encoder}.
* Linear weights are stored ``[out, in]`` (PyTorch layout); Conv ``[Cout, Cin, kh, kw]``;
  ConvTranspose ``[Cin, Cout, kh, kw]`` (depth_pro/mod.rs:416-431).

There are no trained weights on disk (reference .gitignore drops ``*.pt``/``*.mpk``), so every
test/bench uses *synthetic* weights from a counter-based generator that is implemented twice,
bit-identically: here (numpy) and in ``csrc/md_weights.cpp`` (``md_depth_pro_create``).

Container format: safetensors (8-byte LE header length, JSON header, raw little-endian data),
dtype F32 / F16 / BF16, keyed by the names above, plus ``__metadata__`` carrying the config.
"""
from __future__ import annotations

import json
import math
import struct
from typing import Dict, Iterable, List, NamedTuple, Optional, Tuple

import numpy as np

from .config import DepthAnything3Config, DepthProConfig, ViTConfig

# --------------------------------------------------------------------------------------------
# Parameter inventory
# --------------------------------------------------------------------------------------------

INIT_REFERENCE = 0  # Burn default initialisers: U(+-1/sqrt(fan_in)), LN gamma 1 / beta 0
INIT_PARITY = 1     # variance-preserving ranges so that depth/fov land away from the clamps


class ParamSpec(NamedTuple):
    name: str
    shape: Tuple[int, ...]
    lo: float  # uniform range [lo, hi); lo == hi => constant
    hi: float


def _sym(bound: float) -> Tuple[float, float]:
    return (-bound, bound)


def _vit_specs(prefix: str, v: ViTConfig, scheme: int) -> List[ParamSpec]:
    D, P, C = v.embed_dim, v.patch_size, v.in_chans
    hidden = D * v.mlp_ratio
    out: List[ParamSpec] = []
    par = scheme == INIT_PARITY

    def lin(name, fan_out, fan_in, bias=True):
        b = math.sqrt(3.0 / fan_in) if par else math.sqrt(1.0 / fan_in)
        out.append(ParamSpec(f"{name}.weight", (fan_out, fan_in), *_sym(b)))
        if bias:
            bb = 0.1 if par else math.sqrt(1.0 / fan_in)
            out.append(ParamSpec(f"{name}.bias", (fan_out,), *_sym(bb)))

    fan = C * P * P
    b = math.sqrt(3.0 / fan) if par else math.sqrt(1.0 / fan)
    out.append(ParamSpec(f"{prefix}.patch_embed.proj.weight", (D, C, P, P), *_sym(b)))
    out.append(ParamSpec(f"{prefix}.patch_embed.proj.bias", (D,), *_sym(0.1 if par else b)))
    out.append(ParamSpec(f"{prefix}.cls_token", (1, 1, D), *_sym(0.5 if par else 1e-6)))
    # N(0, 0.02) in DINOv2; a uniform of the same std keeps the generator transcendental-free
    s = 0.02 * math.sqrt(3.0)
    out.append(ParamSpec(f"{prefix}.pos_embed", (1, v.num_tokens, D), *_sym(0.3 if par else s)))
    for i in range(v.depth):
        blk = f"{prefix}.blocks.{i}"
        for n in ("norm1", "norm2"):
            out.append(ParamSpec(f"{blk}.{n}.gamma", (D,), *((0.5, 1.5) if par else (1.0, 1.0))))
            out.append(ParamSpec(f"{blk}.{n}.beta", (D,), *(_sym(0.1) if par else (0.0, 0.0))))
        lin(f"{blk}.attn.qkv", 3 * D, D)
        lin(f"{blk}.attn.proj", D, D)
        out.append(ParamSpec(f"{blk}.ls1.gamma", (D,), *((0.05, 0.3) if par else (1.0, 1.0))))
        lin(f"{blk}.mlp.fc1", hidden, D)
        lin(f"{blk}.mlp.fc2", D, hidden)
        out.append(ParamSpec(f"{blk}.ls2.gamma", (D,), *((0.05, 0.3) if par else (1.0, 1.0))))
    out.append(ParamSpec(f"{prefix}.norm.gamma", (D,), *((0.5, 1.5) if par else (1.0, 1.0))))
    out.append(ParamSpec(f"{prefix}.norm.beta", (D,), *(_sym(0.1) if par else (0.0, 0.0))))
    return out


def depth_pro_param_specs(cfg: DepthProConfig, scheme: int = INIT_REFERENCE) -> List[ParamSpec]:
    """Every parameter of ``DepthPro::new`` (depth_pro/mod.rs:145-191), in a fixed order.

    Shapes follow encoder.rs:127-184, decoder.rs:152-193, mod.rs:77-103, fov.rs:63-166."""
    pv, iv, fv = cfg.patch_vit(), cfg.image_vit(), cfg.fov_vit()
    dims = list(pv.encoder_feature_dims)
    F = cfg.decoder_features
    E = pv.embed_dim
    par = scheme == INIT_PARITY
    specs: List[ParamSpec] = []

    def conv(name, cout, cin, k, bias, relu_after=False, gain=None):
        fan = cin * k * k
        if par:
            b = math.sqrt((6.0 if relu_after else 3.0) / fan)
            if gain is not None:
                b *= gain
        else:
            b = math.sqrt(1.0 / fan)
        specs.append(ParamSpec(f"{name}.weight", (cout, cin, k, k), *_sym(b)))
        if bias:
            specs.append(ParamSpec(f"{name}.bias", (cout,), *_sym(0.1 if par else b)))

    def deconv(name, cin, cout, bias):
        # Burn's ConvTranspose2d fan_in uses channels[1]*k*k (out channels); only the range matters
        fan = cin
        b = math.sqrt(3.0 / fan) if par else math.sqrt(1.0 / (cout * 4))
        specs.append(ParamSpec(f"{name}.weight", (cin, cout, 2, 2), *_sym(b)))
        if bias:
            specs.append(ParamSpec(f"{name}.bias", (cout,), *_sym(0.1 if par else b)))

    specs += _vit_specs("encoder.patch_encoder", pv, scheme)
    specs += _vit_specs("encoder.image_encoder", iv, scheme)

    def pub(name, dim_in, dim_out, layers, dim_int=None):  # ProjectUpsampleBlock, encoder.rs:47-75
        inter = dim_int if dim_int is not None else dim_out
        conv(f"{name}.projection", inter, dim_in, 1, False)
        for l in range(layers):
            deconv(f"{name}.upsample.{l}", inter if l == 0 else dim_out, dim_out, False)

    pub("encoder.upsample_latent0", E, F, 3, dims[0])
    pub("encoder.upsample_latent1", E, dims[0], 2)
    pub("encoder.upsample0", E, dims[1], 1)
    pub("encoder.upsample1", E, dims[2], 1)
    pub("encoder.upsample2", E, dims[3], 1)
    deconv("encoder.upsample_lowres", iv.embed_dim, dims[3], True)
    conv("encoder.fuse_lowres", dims[3], dims[3] * 2, 1, True)

    ddims = [F] + dims  # mod.rs:162-163
    if ddims[0] != F:  # decoder.rs:155-165 (never taken: ddims[0] == F by construction)
        conv("decoder.convs.0.conv", F, ddims[0], 1, False)
    for l in range(1, len(ddims)):
        conv(f"decoder.convs.{l}.conv", F, ddims[l], 3, False)
    for l in range(len(ddims)):
        for r in ("resnet1", "resnet2"):
            conv(f"decoder.fusions.{l}.{r}.conv1", F, F, 3, True, relu_after=True)
            conv(f"decoder.fusions.{l}.{r}.conv2", F, F, 3, True, relu_after=True, gain=0.5)
        if l != 0:
            deconv(f"decoder.fusions.{l}.deconv", F, F, False)
        conv(f"decoder.fusions.{l}.out_conv", F, F, 1, True)

    conv("head.conv0", F // 2, F, 3, True)
    deconv("head.deconv", F // 2, F // 2, True)
    conv("head.conv1", 32, F // 2, 3, True, relu_after=True)
    if par:
        # positive weights + positive bias keep canonical inverse depth off the 1e-4 clamp
        specs.append(ParamSpec("head.conv_out.weight", (1, 32, 1, 1), 0.0, 0.08))
        specs.append(ParamSpec("head.conv_out.bias", (1,), 0.05, 0.05))
    else:
        b = math.sqrt(1.0 / 32)
        specs.append(ParamSpec("head.conv_out.weight", (1, 32, 1, 1), *_sym(b)))
        specs.append(ParamSpec("head.conv_out.bias", (1,), 0.0, 0.0))  # mod.rs:92-95

    if cfg.use_fov_head:
        if fv is not None:
            specs += _vit_specs("fov.encoder", fv, scheme)
            fan = fv.embed_dim
            b = math.sqrt(3.0 / fan) if par else math.sqrt(1.0 / fan)
            specs.append(ParamSpec("fov.encoder_proj.weight", (F // 2, fan), *_sym(b)))
            specs.append(ParamSpec("fov.encoder_proj.bias", (F // 2,), *_sym(0.1 if par else b)))
            conv("fov.downsample_blocks.0.conv", F // 2, F, 3, True, relu_after=True)
            conv("fov.head_blocks.0.conv", F // 4, F // 2, 3, True, relu_after=True)
            conv("fov.head_blocks.1.conv", F // 8, F // 4, 3, True, relu_after=True)
            last = ("fov.head_blocks.2.conv", 1, F // 8, 6)
        else:
            conv("fov.head_blocks.0.conv", F // 2, F, 3, True, relu_after=True)
            conv("fov.head_blocks.1.conv", F // 4, F // 2, 3, True, relu_after=True)
            conv("fov.head_blocks.2.conv", F // 8, F // 4, 3, True, relu_after=True)
            last = ("fov.head_blocks.3.conv", 1, F // 8, 6)
        name, cout, cin, k = last
        fan = cin * k * k
        b = math.sqrt(3.0 / fan) if par else math.sqrt(1.0 / fan)
        specs.append(ParamSpec(f"{name}.weight", (cout, cin, k, k), *_sym(b)))
        # parity scheme: field of view around 55 degrees so tan(fov/2) is well conditioned
        specs.append(ParamSpec(f"{name}.bias", (cout,), *((55.0, 55.0) if par else _sym(b))))
    return specs


def da3_param_specs(cfg: DepthAnything3Config, scheme: int = INIT_REFERENCE) -> List[ParamSpec]:
    """Parameters of `DepthAnything3::new` (depth_anything3/mod.rs:253-286, dpt.rs:153-226,515-568,1002-1083,
    camera.rs:113-141), named as the importer maps them (tool/import_da3.rs:67-195): `backbone.pretrained.*`,
    `head_mono.*` | `head_dual.*`, `camera_decoder.*`, `camera_encoder.*` (camera.rs:50-87: PoseBranch, token_norm, a trunk of
    burn_dino `Block`s, trunk_norm; it only runs under `infer_with_camera`, mod.rs:301-309,522-527). The camera-encoder entries come
    last so that adding them left every earlier index where it was."""
    v = cfg.vit()
    par = scheme == INIT_PARITY
    bp = "backbone.pretrained"
    specs: List[ParamSpec] = list(_vit_specs(bp, v, scheme))
    oc, Fh = cfg.out_channels, cfg.features
    hp = "head_dual" if cfg.dual_head else "head_mono"

    def conv(name, cout, cin, k, bias, relu_after=False, gain=None):
        fan = cin * k * k
        if par:
            b = math.sqrt((6.0 if relu_after else 3.0) / fan)
            if gain is not None:
                b *= gain
        else:
            b = math.sqrt(1.0 / fan)
        specs.append(ParamSpec(f"{name}.weight", (cout, cin, k, k), *_sym(b)))
        if bias:
            specs.append(ParamSpec(f"{name}.bias", (cout,), *_sym(0.1 if par else b)))

    def deconv(name, cin, cout, k):
        b = math.sqrt(3.0 / cin) if par else math.sqrt(1.0 / (cout * k * k))
        specs.append(ParamSpec(f"{name}.weight", (cin, cout, k, k), *_sym(b)))
        specs.append(ParamSpec(f"{name}.bias", (cout,), *_sym(0.1 if par else b)))

    def lin(name, fan_out, fan_in, gain=1.0, bias_range=None):
        b = (math.sqrt(3.0 / fan_in) if par else math.sqrt(1.0 / fan_in)) * gain
        specs.append(ParamSpec(f"{name}.weight", (fan_out, fan_in), *_sym(b)))
        if par and bias_range is not None:
            specs.append(ParamSpec(f"{name}.bias", (fan_out,), *bias_range))
        else:
            specs.append(ParamSpec(f"{name}.bias", (fan_out,), *_sym(0.1 if par else math.sqrt(1.0 / fan_in))))

    def norm(name, dim):
        specs.append(ParamSpec(f"{name}.gamma", (dim,), *((0.5, 1.5) if par else (1.0, 1.0))))
        specs.append(ParamSpec(f"{name}.beta", (dim,), *(_sym(0.1) if par else (0.0, 0.0))))

    if cfg.dual_head:
        # burn_dino extras (mod.rs:190-196): per-head q/k LayerNorm from `ext_block_start` on, camera tokens
        for i in range(cfg.ext_block_start, v.depth):
            norm(f"{bp}.blocks.{i}.attn.q_norm", v.head_dim)
            norm(f"{bp}.blocks.{i}.attn.k_norm", v.head_dim)
        specs.append(ParamSpec(f"{bp}.camera_token", (1, 2, v.embed_dim), *_sym(0.5 if par else 1e-6)))
        norm(f"{hp}.norm", cfg.dim_in)
    for i in range(4):
        conv(f"{hp}.projects.{i}", oc[i], cfg.dim_in, 1, True)
    deconv(f"{hp}.resize_layers.0.conv_t", oc[0], oc[0], 4)
    deconv(f"{hp}.resize_layers.1.conv_t", oc[1], oc[1], 2)
    conv(f"{hp}.resize_layers.3.conv", oc[3], oc[3], 3, True)
    for i in range(4):
        conv(f"{hp}.scratch.layer{i + 1}_rn", Fh, oc[i], 3, False)

    def refinenets(suffix):
        for i in (1, 2, 3, 4):
            r = f"{hp}.scratch.refinenet{i}{suffix}"
            units = ("residual1", "residual2") if i != 4 else ("residual2",)
            for u in units:
                conv(f"{r}.{u}.conv1", Fh, Fh, 3, True, relu_after=True)
                conv(f"{r}.{u}.conv2", Fh, Fh, 3, True, relu_after=True, gain=0.5)
            conv(f"{r}.out_conv", Fh, Fh, 1, True)

    refinenets("")
    conv(f"{hp}.scratch.output_conv1", Fh // 2, Fh, 3, True)
    conv(f"{hp}.scratch.output_conv2.conv1", 32, Fh // 2, 3, True, relu_after=True)
    conv(f"{hp}.scratch.output_conv2.conv2", cfg.output_dim, 32, 1, True, gain=0.5)
    if cfg.dual_head:
        refinenets("_aux")
        for lvl in range(cfg.aux_levels):  # AuxPreHead (dpt.rs:1085-1113): C -> C/2 -> C -> ... 3x3 convs, no activation
            cin = Fh
            for j in range(cfg.aux_out1_conv_num):
                cout = Fh // 2 if j % 2 == 0 else Fh
                conv(f"{hp}.scratch.output_conv1_aux.{lvl}.layers.{j}", cout, cin, 3, True)
                cin = cout
        for lvl in range(cfg.aux_levels):  # AuxOutputHead (dpt.rs:1146-1192); LayerNorm2d only on level 0 (dpt.rs:76)
            o = f"{hp}.scratch.output_conv2_aux.{lvl}"
            conv(f"{o}.reduce", 32, Fh // 2, 3, True, relu_after=True)
            if lvl == 0:
                norm(f"{o}.norm.layer_norm", 32)
            conv(f"{o}.project", cfg.aux_output_dim, 32, 1, True, gain=0.5)
        d = cfg.dim_in  # CameraDecoder (camera.rs:113-141)
        lin("camera_decoder.backbone_1", d, d, gain=math.sqrt(2.0) if par else 1.0)
        lin("camera_decoder.backbone_2", d, d, gain=math.sqrt(2.0) if par else 1.0)
        lin("camera_decoder.fc_t", 3, d)
        lin("camera_decoder.fc_qvec", 4, d)
        lin("camera_decoder.fc_fov", 2, d, gain=0.25, bias_range=(0.6, 1.2))  # parity init keeps relu(fov) > 0
    if cfg.camera_encoder:  # CameraEncoder (camera.rs:50-87, 206-234): dim_in = target_dim = 9, dim_out = embed_dim
        D, ce = v.embed_dim, "camera_encoder"
        lin(f"{ce}.pose_branch.fc1", D // 2, 9)
        lin(f"{ce}.pose_branch.fc2", D, D // 2)
        norm(f"{ce}.token_norm", D)
        for i in range(cfg.cam_trunk_depth):
            blk = f"{ce}.trunk.{i}"
            norm(f"{blk}.norm1", D)
            norm(f"{blk}.norm2", D)
            lin(f"{blk}.attn.qkv", 3 * D, D)
            lin(f"{blk}.attn.proj", D, D)
            specs.append(ParamSpec(f"{blk}.ls1.gamma", (D,), *((0.05, 0.3) if par else (1.0, 1.0))))
            lin(f"{blk}.mlp.fc1", 4 * D, D)
            lin(f"{blk}.mlp.fc2", D, 4 * D)
            specs.append(ParamSpec(f"{blk}.ls2.gamma", (D,), *((0.05, 0.3) if par else (1.0, 1.0))))
        norm(f"{ce}.trunk_norm", D)
    return specs


def generate_da3_weights(cfg: DepthAnything3Config, seed: int = 0, scheme: int = INIT_REFERENCE) -> Dict[str, np.ndarray]:
    return _generate(da3_param_specs(cfg, scheme), seed)


# --------------------------------------------------------------------------------------------
# Counter-based generator (bit-identical twin in csrc/md_weights.cpp)
# --------------------------------------------------------------------------------------------

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for ch in name.encode("utf-8"):
        h ^= ch
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def stream_key(name: str, seed: int) -> int:
    return (fnv1a64(name) ^ ((seed * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)) & 0xFFFFFFFFFFFFFFFF


def uniform_stream(name: str, seed: int, count: int, lo: float, hi: float) -> np.ndarray:
    """``count`` float32 values in [lo, hi): element i = lo + (hi-lo) * (u24(i)+0.5)/2^24 with
    u24 the top 24 bits of splitmix64(key + (i+1)*golden). Arithmetic in float64, one cast."""
    lo32, hi32 = np.float32(lo), np.float32(hi)
    if lo32 == hi32:
        return np.full(count, lo32, dtype=np.float32)
    key = np.uint64(stream_key(name, seed))
    with np.errstate(over="ignore"):
        z = key + (np.arange(1, count + 1, dtype=np.uint64) * _GOLDEN)
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    u = ((z >> np.uint64(40)).astype(np.float64) + 0.5) * (1.0 / 16777216.0)
    w = np.float64(hi32) - np.float64(lo32)
    prod = w * u
    return (np.float64(lo32) + prod).astype(np.float32)


def generate_depth_pro_weights(cfg: DepthProConfig, seed: int = 0,
                               scheme: int = INIT_REFERENCE) -> Dict[str, np.ndarray]:
    """Synthetic weights for ``DepthPro::new`` (random init; depth_pro/mod.rs:145-191)."""
    return _generate(depth_pro_param_specs(cfg, scheme), seed)


def _generate(specs, seed: int) -> Dict[str, np.ndarray]:
    """Every stream is independent of the others (its key is the tensor's name): the 0.95 G values of the default Depth Pro
    inventory took 34 s on one thread -- most of a full-size parity test's host time -- so the tensors are filled by a small
    thread pool (numpy's ufuncs release the GIL). Same values, same order of the returned dict."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    specs = list(specs)

    def one(spec):
        n = int(np.prod(spec.shape))
        return uniform_stream(spec.name, seed, n, spec.lo, spec.hi).reshape(spec.shape)
    workers = max(1, min(16, (os.cpu_count() or 1)))
    if workers == 1 or len(specs) < 4:
        return {s.name: one(s) for s in specs}
    with ThreadPoolExecutor(max_workers=workers) as ex:
        vals = list(ex.map(one, specs))
    return {s.name: v for s, v in zip(specs, vals)}


# --------------------------------------------------------------------------------------------
# Container (safetensors subset)
# --------------------------------------------------------------------------------------------

_DTYPES = {"F32": np.float32, "F16": np.float16}


def _f32_to_bf16_bits(a: np.ndarray) -> np.ndarray:
    u = a.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return r.astype(np.uint16)


def _bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << 16).view(np.float32)


def save_container(path: str, tensors: Dict[str, np.ndarray], metadata: Optional[Dict[str, str]] = None,
                   dtype: str = "F32") -> None:
    """Write a safetensors file. ``dtype`` in {F32, F16, BF16} (the reference's .mpk stores f16,
    depth_pro/mod.rs:206)."""
    header: Dict[str, object] = {}
    if metadata:
        header["__metadata__"] = {str(k): str(v) for k, v in metadata.items()}
    blobs: List[bytes] = []
    off = 0
    for name in sorted(tensors):
        arr = np.ascontiguousarray(tensors[name], dtype=np.float32)
        if dtype == "F32":
            raw = arr.tobytes()
        elif dtype == "F16":
            raw = arr.astype(np.float16).tobytes()
        elif dtype == "BF16":
            raw = _f32_to_bf16_bits(arr).tobytes()
        else:
            raise ValueError(f"unsupported container dtype {dtype}")
        header[name] = {"dtype": dtype, "shape": list(arr.shape), "data_offsets": [off, off + len(raw)]}
        blobs.append(raw)
        off += len(raw)
    hjson = json.dumps(header, separators=(",", ":")).encode("utf-8")
    pad = (8 - len(hjson) % 8) % 8
    hjson += b" " * pad
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(hjson)))
        f.write(hjson)
        for raw in blobs:
            f.write(raw)


def load_container(path: str) -> Tuple[Dict[str, np.ndarray], Dict[str, str]]:
    with open(path, "rb") as f:
        raw = f.read()
    if len(raw) < 8:
        raise ValueError("container too short")
    (hlen,) = struct.unpack("<Q", raw[:8])
    if 8 + hlen > len(raw):
        raise ValueError("container header length exceeds file size")
    header = json.loads(raw[8:8 + hlen].decode("utf-8"))
    meta = header.pop("__metadata__", {})
    data = memoryview(raw)[8 + hlen:]
    out: Dict[str, np.ndarray] = {}
    for name, info in header.items():
        b, e = info["data_offsets"]
        dt = info["dtype"]
        buf = np.frombuffer(data[b:e], dtype=np.uint8)
        if dt == "BF16":
            arr = _bf16_bits_to_f32(buf.view(np.uint16))
        elif dt in _DTYPES:
            arr = buf.view(_DTYPES[dt]).astype(np.float32)
        else:
            raise ValueError(f"unsupported dtype {dt} for {name}")
        out[name] = arr.reshape(info["shape"]).copy()
    return out, meta


def config_metadata(cfg: DepthProConfig) -> Dict[str, str]:
    return {
        "model": "depth_pro",
        "patch_encoder_preset": cfg.patch_encoder_preset,
        "image_encoder_preset": cfg.image_encoder_preset,
        "fov_encoder_preset": cfg.fov_encoder_preset or "",
        "decoder_features": str(cfg.decoder_features),
        "use_fov_head": "1" if cfg.use_fov_head else "0",
    }
