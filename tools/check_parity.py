"""Counterpart of the reference's `example/correctness.rs`: run the engine on an image and compare with a
PyTorch-side dump (`tool/correctness_depth_pro.py` output, safetensors) using the harness's names and thresholds.

  python tools/check_parity.py --weights depth_pro.safetensors --image test_rgb.npy --reference test.safetensors

`--image`: uint8 RGB array [H,W,3] as .npy (JPEG decoding is out of scope). The image goes through
`infer_from_rgb` exactly like `example/inference.rs` (normalise -> resize to 1536^2 -> infer -> resize back).
`--precision f32` (default) is the parity mode, `f16x2` the accurate fast mode (activations as hi + lo half planes; both are
held to the reference's thresholds); bf16 / f16 report the throughput modes' error against the same dump."""
import argparse
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(argv=None):
    """Returns (exit code, burn_depth_amd.parity.Report or None)."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--weights", required=True)
    ap.add_argument("--image", required=True)
    ap.add_argument("--reference", required=True)
    ap.add_argument("--precision", choices=["f32", "f16x2", "f16", "bf16"], default="f32")
    ap.add_argument("--no-replay", action="store_true", help="skip the decoder / head replay on the dump's features (the reference's BURN_SKIP_DECODER_REPLAY)")
    ap.add_argument("--preset", choices=["full", "small", "tiny"], default="full", help="reduced presets are for the test-suite")
    a = ap.parse_args(argv)
    import torch
    from burn_depth_amd import parity, weights as Wt
    from burn_depth_amd.config import DepthProConfig, Precision
    from burn_depth_amd.depth_pro import DepthPro, Device
    from burn_depth_amd.inference import rgb_to_input_tensor

    ref = parity.load_reference_dump(Wt.load_container(a.reference)[0])
    rgb = np.load(a.image)
    if rgb.dtype != np.uint8 or rgb.ndim != 3 or rgb.shape[2] != 3:
        print(f"--image must be uint8 [H,W,3], got {rgb.dtype} {rgb.shape}", file=sys.stderr)
        return 2, None
    h, w = rgb.shape[:2]
    dev = Device(0)
    cfg = {"full": DepthProConfig, "small": DepthProConfig.small_test, "tiny": DepthProConfig.tiny_test}[a.preset]()
    cfg.precision = {"f32": Precision.F32, "f16x2": Precision.F16X2, "f16": Precision.F16, "bf16": Precision.BF16}[a.precision]
    model = DepthPro.load_with_config(dev, cfg, a.weights)
    model.enable_taps(True)
    out = model.infer(rgb_to_input_tensor(rgb.tobytes(), w, h, dev))
    torch.cuda.synchronize()
    taps = {}
    for name in ([f"encoder_feature_{i}" for i in range(5)] + [f"decoder_fusion_{i}" for i in range(5)] +
                 ["decoder_feature", "decoder_lowres_feature", "head_conv0", "head_deconv", "canonical_inverse_depth"]):
        try:
            taps[name] = model.read_tap(name)
        except Exception:  # noqa: BLE001 -- a tap the engine does not produce is reported as skipped
            pass
    rep = parity.compare(ref, out.depth[0].cpu().numpy(), float(out.fovx_deg[0]), math.degrees(float(out.fovy_rad[0])), taps)
    print("\n".join(rep.lines))
    # decoder replay (correctness.rs:530-560): the DUMP's encoder features through `decoder_from_features`, and the head layer by
    # layer on the dump's decoder feature (`head_debug`, :382-390) -- separates a decoder / head difference from an encoder one
    if a.no_replay:
        pass
    elif len(ref.encoder_features) != model.query("decoder_levels") or any(f.ndim != 4 for f in ref.encoder_features):
        print("Torch reference missing encoder features; skipping decoder replay.")
    else:
        feat, low, fus = model.decoder_from_features([torch.from_numpy(np.ascontiguousarray(f, np.float32)) for f in ref.encoder_features])
        head = None
        dfeat = ref.optional.get("decoder_feature")
        if dfeat is not None and dfeat.ndim == 4:
            hd = model.head_debug(torch.from_numpy(np.ascontiguousarray(dfeat, np.float32)))
            head = {"head_conv0": hd.conv0, "head_deconv": hd.deconv, "head_conv1": hd.conv1, "head_relu": hd.relu,
                    "head_pre_out": hd.pre_out, "canonical_inverse_depth": hd.canonical}
            head = {k: v.cpu().numpy() for k, v in head.items()}
        lines = parity.replay_report(ref, feat.cpu().numpy(), low.cpu().numpy(), [t.cpu().numpy() for t in fus], head)
        print("\n".join(lines))
        rep = rep._replace(lines=rep.lines + lines)
    model.destroy()
    return (0 if rep.ok else 1), rep


def main(argv=None) -> int:
    return run(argv)[0]


if __name__ == "__main__":
    sys.exit(main())
