"""CLI: convert an upstream checkpoint into the engine's weight container.

  python tools/import_weights.py depth_pro  checkpoints/depth_pro.pt      depth_pro.safetensors  [--f32]
  python tools/import_weights.py da3_large  DA3-metric-large/model.safetensors  da3.safetensors  [--f32]
  python tools/import_weights.py da3_small  DA3-small/model.safetensors         da3s.safetensors [--f32]

Counterpart of the reference's `tool/import_depth_pro.rs` / `tool/import_da3.rs` (which write Burn `.mpk`
records); the output loads through `md_depth_pro_load` / `md_da3_load`."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from burn_depth_amd import importer  # noqa: E402


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("model", choices=["depth_pro", "da3_large", "da3_small"])
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--f32", action="store_true", help="store fp32 instead of the reference's f16 records")
    a = ap.parse_args()
    dtype = "F32" if a.f32 else "F16"
    try:
        if a.model == "depth_pro":
            t = importer.import_depth_pro(a.src, a.dst, dtype=dtype)
        else:
            from burn_depth_amd.config import DepthAnything3Config
            cfg = DepthAnything3Config.small() if a.model == "da3_small" else DepthAnything3Config.metric_large()
            t = importer.import_da3(a.src, a.dst, cfg, dtype=dtype)
    except (importer.ImportError_, OSError) as e:
        print(f"import failed: {e}", file=sys.stderr)
        return 1
    n = sum(int(v.size) for v in t.values())
    print(f"{a.dst}: {len(t)} tensors, {n / 1e6:.1f} M parameters, {dtype}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
