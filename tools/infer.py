"""Counterpart of the reference's `example/inference.rs`: image -> depth PNG.

  python tools/infer.py --model depth-pro        --checkpoint depth_pro.safetensors --image photo.npy [--output depth.png]
  python tools/infer.py --model depth-anything-3 --checkpoint da3_small.safetensors  --image photo.npy

`--image`: uint8 RGB [H,W,3] as .npy (JPEG decoding is out of scope). Depth-Anything-v3 inputs are resized on the
shortest side (Catmull-Rom) and centre-cropped to the model resolution (src/model/mod.rs:162-210); the depth map is
restored to the original size, min-max normalised and written as an 8-bit PNG (example/inference.rs:103-199)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(argv=None) -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", choices=["depth-pro", "depth-anything-3"], default="depth-pro")
    ap.add_argument("--checkpoint", required=True)
    ap.add_argument("--image", required=True)
    ap.add_argument("--output", default="")
    ap.add_argument("--precision", choices=["bf16", "f32"], default="bf16")
    a = ap.parse_args(argv)
    from burn_depth_amd import pipeline as P
    from burn_depth_amd.config import Precision
    from burn_depth_amd.depth_pro import Device
    if not os.path.exists(a.checkpoint):   # example/inference.rs:52-62
        print(f"Checkpoint `{a.checkpoint}` not found. Run tools/import_weights.py first.", file=sys.stderr)
        return 1
    rgb = np.load(a.image)
    if rgb.dtype != np.uint8 or rgb.ndim != 3 or rgb.shape[2] != 3:
        print(f"--image must be uint8 [H,W,3], got {rgb.dtype} {rgb.shape}", file=sys.stderr)
        return 2
    kind = P.DepthModelKind(a.model)
    try:
        model = P.AnyDepthModel.load(kind, Device(0), a.checkpoint, Precision.BF16 if a.precision == "bf16" else Precision.F32)
    except RuntimeError as e:
        print(str(e), file=sys.stderr)
        return 1
    oh, ow = rgb.shape[:2]
    prep = model.prepare_input_image(rgb)
    out = model.infer_from_rgb(prep)
    restore = (ow, oh) if (prep.width != ow or prep.height != oh or prep.crop is not None) else None
    path = a.output or os.path.join(os.path.dirname(os.path.abspath(a.image)), "depth.png")
    P.save_depth_map(out.depth.cpu().numpy(), path, prep.crop, restore)
    f = getattr(out, "focallength_px", None)
    print(f"Focal length (px): {f.cpu().tolist() if f is not None else 'not provided by this model'}")
    fy = getattr(out, "fovy_rad", None)
    print(f"Vertical FOV (rad): {fy.cpu().tolist() if fy is not None else 'not provided by this model'}")
    print(f"Model `{kind.value}` wrote normalized depth map to {path}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
