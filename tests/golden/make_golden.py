"""Generates the committed golden vectors under tests/golden/ from the CPU oracle.

The reference (Rust/Burn) can be neither built nor imported in this image, so these
vectors are outputs of `oracle/` -- itself pinned by the reference's known-answer tests
(tests/test_oracle_kats.py) -- on seeded inputs and seeded weights.  They serve two purposes:
  * `-m "not gpu"`: the oracle must keep reproducing them (guards the checker against drift);
  * `-m gpu`: the HIP path is compared with them directly, no oracle in the loop.

Inputs are not stored: they are `torch.manual_seed(seed); torch.rand(B,3,H,W)` normalised like
`rgb_to_input_tensor` (inference.rs:96-117), and the weights come from the counter-based generator
(`burn_depth_amd.weights`, seed 0, INIT_PARITY), both bit-reproducible.

Usage:  python tests/golden/make_golden.py        (rewrites the .npz files next to this script)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from burn_depth_amd import weights as Wt  # noqa: E402
from burn_depth_amd.config import DepthAnything3Config, DepthProConfig  # noqa: E402
from oracle import da3_ref as D3  # noqa: E402
from oracle import depth_pro_ref as R  # noqa: E402

SUB = 4  # spatial subsampling of the stored depth maps (full-map checksums are stored beside them)


def seeded_input(seed, B, H, W):
    torch.manual_seed(seed)
    img = torch.rand(B, 3, H, W)
    return (img - torch.tensor(R.MEAN).view(1, 3, 1, 1)) / torch.tensor(R.STD).view(1, 3, 1, 1)


def depth_pro_case(B=2, seed=0):
    cfg = DepthProConfig.tiny_test()
    S = cfg.img_size()
    W = R.weights_to_torch(Wt.generate_depth_pro_weights(cfg, 0, Wt.INIT_PARITY))
    x = seeded_input(seed, B, S, S)
    with torch.no_grad():
        ref = R.infer(x, W, cfg, q=R.identity, debug=True)
    d = ref["depth"].numpy()
    return dict(
        batch=np.int32(B), seed=np.int32(seed), image_size=np.int32(S), sub=np.int32(SUB),
        depth_sub=d[:, ::SUB, ::SUB].astype(np.float32),
        depth_sum=np.float64(d.astype(np.float64).sum()), depth_sumsq=np.float64((d.astype(np.float64) ** 2).sum()),
        fovx_deg=ref["fovx_deg"].numpy().astype(np.float32), fovy_rad=ref["fovy_rad"].numpy().astype(np.float32),
        focallength_px=ref["focallength_px"].numpy().astype(np.float32),
        canonical_sub=ref["debug"]["canonical"].numpy()[:, :, ::SUB, ::SUB].astype(np.float32),
        enc4_sub=ref["debug"]["encoder"]["features"][4].numpy()[:, ::8].astype(np.float32),
    )


def da3_case(B=1, seed=0):
    cfg = DepthAnything3Config.tiny_test()
    S = cfg.image_size
    W = R.weights_to_torch(Wt.generate_da3_weights(cfg, 0, Wt.INIT_PARITY))
    x = seeded_input(seed, B, S, S)
    with torch.no_grad():
        ref = D3.infer(x, W, cfg)
    d = ref["depth"].numpy()
    return dict(batch=np.int32(B), seed=np.int32(seed), image_size=np.int32(S), depth=d.astype(np.float32))


def da3_dual_case(B=1, seed=0):
    cfg = DepthAnything3Config.tiny_dual_test()
    S = cfg.image_size
    W = R.weights_to_torch(Wt.generate_da3_weights(cfg, 0, Wt.INIT_PARITY))
    x = seeded_input(seed, B, S, S)
    with torch.no_grad():
        ref = D3.infer(x, W, cfg, debug=True)
    out = dict(batch=np.int32(B), seed=np.int32(seed), image_size=np.int32(S))
    for k in ("depth", "depth_confidence", "aux", "aux_confidence", "pose_encoding", "extrinsics", "intrinsics"):
        out[k] = ref[k].numpy().astype(np.float32)
    # `infer_raw` (mod.rs:364-380): the main logits before the activations; `infer_from_tokens` (mod.rs:389-469): the four hooks'
    # patch tokens the head consumes (every 5th token row keeps the fixture small: the test feeds the oracle's full tokens instead)
    out["main_logits"] = ref["debug"]["main_logits"].numpy().astype(np.float32)
    return out


def camera_inputs(B, V, H, W):
    """Fixed world-to-camera extrinsics [B, V, 3, 4] (rotations that take each of the four branches of the reference's
    matrix_to_quaternion, camera.rs:418-514) and pinhole intrinsics [B, V, 3, 3] with fx, fy either side of W/2, H/2."""
    import math

    def rot(axis, ang):
        a = np.asarray(axis, float)
        a /= np.linalg.norm(a)
        K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        return np.eye(3) + math.sin(ang) * K + (1 - math.cos(ang)) * K @ K
    poses = [((1, 2, 3), 0.4), ((1, .1, .1), 3.0), ((.1, 1, .1), 3.0), ((.1, .1, 1), 3.0)]
    E, I = np.zeros((B, V, 3, 4), np.float32), np.zeros((B, V, 3, 3), np.float32)
    for b in range(B):
        for v in range(V):
            i = b * V + v
            ax, ang = poses[i % 4]
            E[b, v, :, :3] = rot(ax, ang + 0.01 * i)
            E[b, v, :, 3] = [0.1 * (i + 1), -0.2, 0.05 * i - 0.3]
            I[b, v] = [[W * (0.4 + 0.15 * (i % 4)), 0, W / 2], [0, H * (0.9 - 0.15 * (i % 4)), H / 2], [0, 0, 1]]
    return torch.from_numpy(E), torch.from_numpy(I)


def da3_camera_case(B=2, V=2, seed=0):
    """`infer_with_camera` (depth_anything3/mod.rs:301-309) on the reduced dual-head variant: the camera encoder's token and the
    outputs it conditions."""
    cfg = DepthAnything3Config.tiny_dual_test()
    S = cfg.image_size
    W = R.weights_to_torch(Wt.generate_da3_weights(cfg, 0, Wt.INIT_PARITY))
    x = seeded_input(seed, B, S, S)
    E, I = camera_inputs(B, V, S, S)
    with torch.no_grad():
        ref = D3.infer(x, W, cfg, debug=True, extrinsics=E, intrinsics=I)
    out = dict(batch=np.int32(B), views=np.int32(V), seed=np.int32(seed), image_size=np.int32(S),
               in_extrinsics=E.numpy(), in_intrinsics=I.numpy(), in_pose_encoding=ref["debug"]["pose_encoding_in"].numpy().astype(np.float32),
               camera_token=ref["debug"]["camera_token"].numpy().astype(np.float32))
    for k in ("depth", "depth_confidence", "aux_confidence", "pose_encoding", "extrinsics"):
        out[k] = ref[k].numpy().astype(np.float32)
    return out


def main():
    np.savez_compressed(os.path.join(HERE, "da3_tiny_dual_camera_f32.npz"), **da3_camera_case())
    np.savez_compressed(os.path.join(HERE, "depth_pro_tiny_f32.npz"), **depth_pro_case())
    np.savez_compressed(os.path.join(HERE, "da3_tiny_f32.npz"), **da3_case())
    np.savez_compressed(os.path.join(HERE, "da3_tiny_dual_f32.npz"), **da3_dual_case())
    for f in ("depth_pro_tiny_f32.npz", "da3_tiny_f32.npz", "da3_tiny_dual_f32.npz", "da3_tiny_dual_camera_f32.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
