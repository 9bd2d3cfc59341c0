"""Reference-dump comparison with the schema and thresholds of the reference's own harness
(`example/correctness.rs`): a PyTorch-side dump (`tool/correctness_depth_pro.py`) is a safetensors file with
`metric_depth [H,W,1]`, `fovy [1]`, `fovx [1]` (degrees), `encoder_feature_{i}`, `decoder_fusion_{i}` and optional
intermediates (`correctness.rs:161-252`); the harness prints mean-abs / max-abs / max-rel per tensor
(`:511-528`) and passes when depth max-abs <= 5e-3, mean-abs <= 1e-3, max-rel <= 5e-3 and both fov diffs <= 1e-3
(`:887-897`). `tools/check_parity.py` feeds the engine's outputs and debug taps (same names) through this."""
from __future__ import annotations

from typing import Dict, List, NamedTuple, Optional

import numpy as np

DEPTH_MAX_ABS_THRESHOLD = 5e-3   # example/correctness.rs:887
DEPTH_MEAN_ABS_THRESHOLD = 1e-3  # :888
DEPTH_MAX_REL_THRESHOLD = 5e-3   # :889
FOVX_THRESHOLD = 1e-3            # :890
FOVY_THRESHOLD = 1e-3            # :891

OPTIONAL_FEATURES = (
    "encoder_merge_latent0", "encoder_merge_latent1", "encoder_latent0_tokens", "encoder_latent1_tokens",
    "encoder_latent0_merge_input", "encoder_latent1_merge_input", "encoder_merge_x0", "encoder_merge_x1",
    "encoder_merge_x2", "canonical_inverse_depth", "decoder_feature", "decoder_lowres_feature", "head_conv0",
    "head_deconv", "head_conv1", "head_relu", "head_pre_out")


class Stats(NamedTuple):
    mean_abs: float
    max_abs: float
    max_rel: float


def compute_stats(ours: np.ndarray, ref: np.ndarray) -> Stats:
    """correctness.rs:486-509: max_rel divides by max(|ref|, 1e-6)."""
    a, b = np.asarray(ours, np.float32).reshape(-1), np.asarray(ref, np.float32).reshape(-1)
    if a.size != b.size:
        raise ValueError(f"element count mismatch {a.size} vs {b.size}")
    d = np.abs(a - b)
    return Stats(float(d.mean()), float(d.max()), float((d / np.maximum(np.abs(b), 1e-6)).max()))


class ReferenceDump(NamedTuple):
    depth: np.ndarray               # [H, W]
    fovx_deg: float
    fovy_deg: float
    encoder_features: List[np.ndarray]
    decoder_fusions: List[np.ndarray]
    optional: Dict[str, np.ndarray]


def load_reference_dump(tensors: Dict[str, np.ndarray]) -> ReferenceDump:
    """`load_torch_reference` (correctness.rs:161-252), with its error messages."""
    if "metric_depth" not in tensors:
        raise ValueError("missing `metric_depth` tensor in reference file")
    d = tensors["metric_depth"]
    if d.ndim != 3 or d.shape[2] != 1:
        raise ValueError(f"expected torch depth shape [H, W, 1], got {list(d.shape)}")
    for k in ("fovy", "fovx"):
        if k not in tensors:
            raise ValueError(f"missing `{k}` tensor in reference file")
        if tuple(tensors[k].shape) != (1,):
            raise ValueError(f"expected {k} shape [1], got {list(tensors[k].shape)}")

    def series(prefix):
        out, i = [], 0
        while f"{prefix}_{i}" in tensors:
            out.append(tensors[f"{prefix}_{i}"])
            i += 1
        return out

    return ReferenceDump(d[:, :, 0], float(tensors["fovx"][0]), float(tensors["fovy"][0]), series("encoder_feature"),
                         series("decoder_fusion"), {k: tensors[k] for k in OPTIONAL_FEATURES if k in tensors})


class Report(NamedTuple):
    lines: List[str]
    depth: Stats
    fovx_diff: float
    fovy_diff: float
    ok: bool


def compare(ref: ReferenceDump, depth: np.ndarray, fovx_deg: float, fovy_deg: float,
            taps: Optional[Dict[str, np.ndarray]] = None) -> Report:
    """The harness's verdict (correctness.rs:860-905). `taps`: engine debug tensors by the dump's names."""
    taps = taps or {}
    lines: List[str] = []
    if tuple(depth.shape) != tuple(ref.depth.shape):
        raise ValueError(f"depth shape mismatch: ours {tuple(depth.shape)} reference {tuple(ref.depth.shape)}")
    ds = compute_stats(depth, ref.depth)
    fx, fy = abs(fovx_deg - ref.fovx_deg), abs(fovy_deg - ref.fovy_deg)
    lines.append(f"depth: mean abs={ds.mean_abs:.6f}, max abs={ds.max_abs:.6f}, max rel={ds.max_rel:.6f}")
    lines.append(f"fovx: ours={fovx_deg:.6f} reference={ref.fovx_deg:.6f} diff={fx:.6f}")
    lines.append(f"fovy: ours={fovy_deg:.6f} reference={ref.fovy_deg:.6f} diff={fy:.6f}")

    def feature(name, theirs):
        ours = taps.get(name)
        if ours is None:
            lines.append(f"{name}: no engine tap; skipping comparison")
        elif tuple(ours.shape) != tuple(theirs.shape):
            lines.append(f"{name}: shape mismatch torch={list(theirs.shape)} ours={list(ours.shape)}")
        else:
            s = compute_stats(ours, theirs)
            lines.append(f"{name}: mean abs={s.mean_abs:.6f}, max abs={s.max_abs:.6f}, max rel={s.max_rel:.6f}")

    for i, t in enumerate(ref.encoder_features):
        feature(f"encoder_feature_{i}", t)
    for i, t in enumerate(ref.decoder_fusions):
        feature(f"decoder_fusion_{i}", t)
    for k, t in ref.optional.items():
        feature(k, t)
    ok = (ds.max_abs <= DEPTH_MAX_ABS_THRESHOLD and ds.mean_abs <= DEPTH_MEAN_ABS_THRESHOLD and
          ds.max_rel <= DEPTH_MAX_REL_THRESHOLD and fx <= FOVX_THRESHOLD and fy <= FOVY_THRESHOLD)
    lines.append("Output matches the reference dump within tolerance." if ok else "Output differs from the reference dump.")
    return Report(lines, ds, fx, fy, ok)


def replay_report(ref: ReferenceDump, feature: np.ndarray, lowres: np.ndarray, fusions: List[np.ndarray],
                  head: Optional[Dict[str, np.ndarray]] = None) -> List[str]:
    """`compare_decoder_with_reference` (correctness.rs:530-660): the engine's `decoder_from_features` on the DUMP's
    encoder features against the dump's decoder tensors, in the harness's "[Replay]" lines; `head`: `head_debug` on the
    dump's `decoder_feature` (head_conv0, head_deconv, head_conv1, head_relu, head_pre_out, canonical_inverse_depth),
    the stage-by-stage head comparison of correctness.rs:382-390,700-760."""
    lines: List[str] = []

    def one(label, ours, theirs):
        if theirs is None:
            lines.append(f"[Replay] Torch reference missing {label}; skipping.")
        elif tuple(ours.shape) != tuple(theirs.shape):
            lines.append(f"[Replay] {label} shape mismatch: torch {list(theirs.shape)}, ours {list(ours.shape)}")
        else:
            st = compute_stats(ours, theirs)
            lines.append(f"[Replay] {label}: mean abs={st.mean_abs:.6f}, max abs={st.max_abs:.6f}, max rel={st.max_rel:.6f}")
            if st.max_abs > 1e-3:  # correctness.rs:576-596: where the largest difference sits
                d = np.abs(np.asarray(ours, np.float32) - np.asarray(theirs, np.float32))
                at = np.unravel_index(int(d.argmax()), d.shape)
                lines.append(f"[Replay] {label} max diff at {list(at)}: ours={float(np.asarray(ours)[at]):.6f}, "
                             f"torch={float(np.asarray(theirs)[at]):.6f}, diff={float(d[at]):.6f}")

    one("Decoder feature", feature, ref.optional.get("decoder_feature"))
    one("Decoder lowres feature", lowres, ref.optional.get("decoder_lowres_feature"))
    if len(fusions) == len(ref.decoder_fusions):
        for i, (a, b) in enumerate(zip(fusions, ref.decoder_fusions)):
            one(f"Decoder fusion {i}", a, b)
    else:
        lines.append(f"[Replay] fusion count mismatch: torch {len(ref.decoder_fusions)}, ours {len(fusions)}")
    for k, v in (head or {}).items():
        one(f"Head {k}", v, ref.optional.get(k))
    return lines
