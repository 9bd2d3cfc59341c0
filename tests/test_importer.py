"""Upstream-checkpoint importer: the rename rules of the reference's import tools
(`tool/import_depth_pro.rs:344-437`, `tool/import_da3.rs:67-195`) applied to synthetic upstream-style state
dicts. The upstream key spellings below are written out by hand from the upstream module layouts (timm ViT
names, `nn.Sequential` indices), independently of the importer's rule table."""
import numpy as np
import pytest
import torch

from burn_depth_amd import importer, weights as Wt
from burn_depth_amd.config import DepthAnything3Config, DepthProConfig


def burn_to_upstream_depth_pro(name: str) -> str:
    """Inverse map, spelled out case by case (test-side twin of the rule table)."""
    n = name
    if n.startswith("fov.encoder_proj."):
        return n.replace("fov.encoder_proj.", "fov.encoder.1.")
    if n.startswith("fov.encoder."):
        n = "fov.encoder.0." + n[len("fov.encoder."):]
    if n.endswith(".gamma") and ".norm" in n:
        n = n[:-len("gamma")] + "weight"
    if n.endswith(".beta") and ".norm" in n:
        n = n[:-len("beta")] + "bias"
    for k in ("upsample0", "upsample1", "upsample2", "upsample_latent0", "upsample_latent1"):
        p = f"encoder.{k}."
        if n.startswith(p):
            rest = n[len(p):]
            if rest.startswith("projection."):
                return p + "0." + rest[len("projection."):]
            idx, leaf = rest[len("upsample."):].split(".", 1)
            return p + f"{int(idx) + 1}." + leaf
    if n.startswith("fov.downsample_blocks."):
        i, _, leaf = n[len("fov.downsample_blocks."):].split(".", 2)
        return f"fov.downsample.{i}.{leaf}"
    if n.startswith("fov.head_blocks."):
        i, _, leaf = n[len("fov.head_blocks."):].split(".", 2)
        return f"fov.head.{2 * int(i)}.{leaf}"
    if n.startswith("decoder.convs."):
        i, _, leaf = n[len("decoder.convs."):].split(".", 2)
        return f"decoder.convs.{i}.{leaf}"
    if ".resnet" in n and (".conv1." in n or ".conv2." in n):
        return n.replace(".conv1.", ".residual.1.").replace(".conv2.", ".residual.3.")
    head = {"head.conv0.": "head.0.", "head.deconv.": "head.1.", "head.conv1.": "head.2.", "head.conv_out.": "head.4."}
    for a, b in head.items():
        if n.startswith(a):
            return b + n[len(a):]
    return n


def test_rule_table_examples():
    r = importer.DEPTH_PRO_RULES
    ex = {
        "encoder.patch_encoder.blocks.3.norm1.weight": "encoder.patch_encoder.blocks.3.norm1.gamma",
        "encoder.image_encoder.norm.bias": "encoder.image_encoder.norm.beta",
        "fov.encoder.0.blocks.11.norm2.weight": "fov.encoder.blocks.11.norm2.gamma",
        "fov.encoder.0.pos_embed": "fov.encoder.pos_embed",
        "fov.encoder.1.weight": "fov.encoder_proj.weight",
        "encoder.upsample_latent0.3.weight": "encoder.upsample_latent0.upsample.2.weight",
        "encoder.upsample1.0.weight": "encoder.upsample1.projection.weight",
        "encoder.upsample_lowres.bias": "encoder.upsample_lowres.bias",
        "decoder.fusions.2.resnet1.residual.3.bias": "decoder.fusions.2.resnet1.conv2.bias",
        "decoder.convs.4.weight": "decoder.convs.4.conv.weight",
        "head.4.bias": "head.conv_out.bias",
        "fov.head.4.weight": "fov.head_blocks.2.conv.weight",
        "fov.downsample.0.bias": "fov.downsample_blocks.0.conv.bias",
        "encoder.patch_encoder.blocks.0.attn.qkv.weight": "encoder.patch_encoder.blocks.0.attn.qkv.weight",
        "encoder.patch_encoder.blocks.0.ls1.gamma": "encoder.patch_encoder.blocks.0.ls1.gamma",
    }
    for k, v in ex.items():
        assert importer.remap_key(k, r) == v, k
    d = importer.da3_rules("head_mono")
    ex3 = {
        "model.backbone.pretrained.blocks.7.norm2.bias": "backbone.pretrained.blocks.7.norm2.beta",
        "model.backbone.pretrained.norm.weight": "backbone.pretrained.norm.gamma",
        "model.head.resize_layers.0.weight": "head_mono.resize_layers.0.conv_t.weight",
        "model.head.resize_layers.3.bias": "head_mono.resize_layers.3.conv.bias",
        "model.head.scratch.output_conv2.2.weight": "head_mono.scratch.output_conv2.conv2.weight",
        "model.head.scratch.refinenet3.resConfUnit1.conv2.weight": "head_mono.scratch.refinenet3.residual1.conv2.weight",
        "model.head.projects.2.weight": "head_mono.projects.2.weight",
        "model.head.scratch.layer1_rn.weight": "head_mono.scratch.layer1_rn.weight",
    }
    for k, v in ex3.items():
        assert importer.remap_key(k, d) == v, k


def test_depth_pro_round_trip(tmp_path):
    cfg = DepthProConfig.tiny_test()
    W = Wt.generate_depth_pro_weights(cfg, 3, Wt.INIT_REFERENCE)
    upstream = {burn_to_upstream_depth_pro(k): torch.from_numpy(v.copy()) for k, v in W.items()}
    upstream["encoder.patch_encoder.mask_token"] = torch.zeros(1, 1, 256)  # dropped, as in the reference tool
    assert len(upstream) == len(W) + 1 and set(upstream) != set(W)
    src, dst = str(tmp_path / "depth_pro.pt"), str(tmp_path / "dp.safetensors")
    torch.save(upstream, src)
    importer.import_depth_pro(src, dst, cfg, dtype="F32")
    got, meta = Wt.load_container(dst)
    assert meta["model"] == "depth_pro"
    assert set(got) == set(W)
    for k in W:
        assert got[k].shape == W[k].shape and np.array_equal(got[k], W[k]), k
    # f16 records like the reference's HalfPrecisionSettings
    importer.import_depth_pro(src, dst, cfg, dtype="F16")
    got16, _ = Wt.load_container(dst)
    k = "encoder.patch_encoder.blocks.0.attn.qkv.weight"
    assert np.array_equal(got16[k], W[k].astype(np.float16).astype(np.float32))


def test_depth_pro_rejects_bad_checkpoints(tmp_path):
    cfg = DepthProConfig.tiny_test()
    W = Wt.generate_depth_pro_weights(cfg, 3, Wt.INIT_REFERENCE)
    up = {burn_to_upstream_depth_pro(k): torch.from_numpy(v.copy()) for k, v in W.items()}
    specs = Wt.depth_pro_param_specs(cfg, Wt.INIT_REFERENCE)
    missing = dict(up)
    del missing["head.2.weight"]
    with pytest.raises(importer.ImportError_, match="missing tensor `head.conv1.weight`"):
        importer.convert_state_dict(missing, specs, importer.DEPTH_PRO_RULES, importer.DEPTH_PRO_IGNORED)
    extra = dict(up)
    extra["decoder.fusions.0.deconv.weight"] = torch.zeros(4)
    with pytest.raises(importer.ImportError_, match="unexpected tensor"):
        importer.convert_state_dict(extra, specs, importer.DEPTH_PRO_RULES, importer.DEPTH_PRO_IGNORED)
    bad = dict(up)
    bad["head.0.weight"] = torch.zeros(3, 3)
    with pytest.raises(importer.ImportError_, match="has shape"):
        importer.convert_state_dict(bad, specs, importer.DEPTH_PRO_RULES, importer.DEPTH_PRO_IGNORED)


def test_da3_round_trip(tmp_path):
    cfg = DepthAnything3Config.tiny_test()
    W = Wt.generate_da3_weights(cfg, 5, Wt.INIT_REFERENCE)

    def up(n: str) -> str:
        if n.endswith(".gamma") and ".norm" in n:
            n = n[:-len("gamma")] + "weight"
        elif n.endswith(".beta") and ".norm" in n:
            n = n[:-len("beta")] + "bias"
        if n.startswith("head_mono."):
            n = "head." + n[len("head_mono."):]
            n = n.replace(".conv_t.", ".").replace("resize_layers.3.conv.", "resize_layers.3.")
            n = n.replace("output_conv2.conv1.", "output_conv2.0.").replace("output_conv2.conv2.", "output_conv2.2.")
            n = n.replace(".residual1.", ".resConfUnit1.").replace(".residual2.", ".resConfUnit2.")
        return "model." + n

    upstream = {up(k): v for k, v in W.items()}
    upstream["model.cam_enc.trunk.0.norm1.weight"] = np.zeros(4, np.float32)   # no camera encoder in this variant: dropped
    src, dst = str(tmp_path / "model.safetensors"), str(tmp_path / "da3.safetensors")
    Wt.save_container(src, upstream, dtype="F32")
    importer.import_da3(src, dst, cfg, dtype="F32")
    got, meta = Wt.load_container(dst)
    assert meta["model"] == "depth_anything3" and set(got) == set(W)
    for k in W:
        assert np.array_equal(got[k], W[k]), k


def test_da3_small_round_trip(tmp_path):
    """Dual head + camera decoder + backbone extras: upstream spellings written out by hand (Sequential indices of
    the aux heads and of the camera decoder, `cam_dec.` / `cam_enc.` prefixes, q_norm/k_norm weight/bias)."""
    cfg = DepthAnything3Config.tiny_dual_test()
    W = Wt.generate_da3_weights(cfg, 7, Wt.INIT_REFERENCE)

    def up(n: str) -> str:
        if (n.endswith(".gamma") or n.endswith(".beta")) and ("norm" in n):
            n = n[:n.rfind(".")] + (".weight" if n.endswith(".gamma") else ".bias")
        if n.startswith("head_dual."):
            n = "head." + n[len("head_dual."):]
            n = n.replace(".conv_t.", ".").replace("resize_layers.3.conv.", "resize_layers.3.")
            n = n.replace("output_conv2.conv1.", "output_conv2.0.").replace("output_conv2.conv2.", "output_conv2.2.")
            n = n.replace(".residual1.", ".resConfUnit1.").replace(".residual2.", ".resConfUnit2.")
            n = n.replace(".layers.", ".").replace(".reduce.", ".0.").replace(".project.", ".5.")
            n = n.replace(".norm.layer_norm.", ".2.")
        if n.startswith("camera_decoder."):
            n = "cam_dec." + n[len("camera_decoder."):]
            n = n.replace("backbone_1.", "backbone.0.").replace("backbone_2.", "backbone.2.").replace("fc_fov.", "fc_fov.0.")
        if n.startswith("camera_encoder."):
            n = "cam_enc." + n[len("camera_encoder."):]
        return "model." + n

    upstream = {up(k): v for k, v in W.items()}
    assert "model.cam_dec.backbone.2.weight" in upstream and "model.head.scratch.output_conv2_aux.0.2.weight" in upstream
    assert "model.backbone.pretrained.blocks.3.attn.q_norm.weight" in upstream and "model.head.norm.bias" in upstream
    for k in ("trunk.1.norm2.bias", "token_norm.weight", "trunk_norm.bias", "pose_branch.fc1.weight", "trunk.0.ls1.gamma", "trunk.0.attn.qkv.bias"):
        assert "model.cam_enc." + k in upstream, k     # the camera encoder (import_da3.rs:88,184-195)
    upstream["model.backbone.pretrained.mask_token"] = np.zeros(4, np.float32)   # dropped
    src, dst = str(tmp_path / "model.safetensors"), str(tmp_path / "da3s.safetensors")
    Wt.save_container(src, upstream, dtype="F32")
    importer.import_da3(src, dst, cfg, dtype="F32")
    got, _ = Wt.load_container(dst)
    assert set(got) == set(W)
    for k in W:
        assert np.array_equal(got[k], W[k]), k


def test_burn_mpk_record_round_trip(tmp_path):
    """`.mpk` (Burn NamedMpk, f16) -> container: the reader is validated against files written by burn_depth_amd.mpk
    itself (no `.mpk` exists in the reference tree): nested maps, Vec<Module> as arrays, {id, param{bytes,shape,dtype}}."""
    import msgpack
    from burn_depth_amd import mpk
    cfg = DepthProConfig.tiny_test()
    W = Wt.generate_depth_pro_weights(cfg, 1, Wt.INIT_REFERENCE)
    src, dst = str(tmp_path / "depth_pro.mpk"), str(tmp_path / "dp.safetensors")
    mpk.write_mpk(src, W, dtype="F16")
    doc = msgpack.unpackb(open(src, "rb").read(), raw=False)
    assert set(doc) == {"metadata", "item"} and isinstance(doc["item"]["encoder"]["patch_encoder"]["blocks"], list)
    leaf = doc["item"]["head"]["conv0"]["weight"]
    assert set(leaf) == {"id", "param"} and leaf["param"]["dtype"] == "F16" and leaf["param"]["shape"] == [32, 64, 3, 3]
    got = mpk.read_mpk(src)
    assert set(got) == set(W)
    k = "encoder.patch_encoder.blocks.1.mlp.fc1.weight"
    assert np.array_equal(got[k], W[k].astype(np.float16).astype(np.float32))
    importer.import_depth_pro(src, dst, cfg, dtype="F16")          # .mpk straight into the engine container
    c, _ = Wt.load_container(dst)
    assert np.array_equal(c[k], got[k])
    with pytest.raises(ValueError, match="not a Burn record"):
        open(str(tmp_path / "bad.mpk"), "wb").write(msgpack.packb({"x": 1}))
        mpk.read_mpk(str(tmp_path / "bad.mpk"))
