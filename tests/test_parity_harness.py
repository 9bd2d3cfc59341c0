"""The reference-dump checker (burn_depth_amd/parity.py) follows example/correctness.rs: schema (:161-252),
statistics (:486-509) and thresholds (:887-897)."""
import numpy as np
import pytest

from burn_depth_amd import parity


def _dump(h=6, w=8):
    rng = np.random.default_rng(0)
    t = {"metric_depth": rng.uniform(0.5, 4.0, (h, w, 1)).astype(np.float32), "fovx": np.array([53.0], np.float32),
         "fovy": np.array([40.0], np.float32)}
    for i in range(3):
        t[f"encoder_feature_{i}"] = rng.normal(size=(1, 4, 3, 3)).astype(np.float32)
    t["decoder_fusion_0"] = rng.normal(size=(1, 4, 6, 6)).astype(np.float32)
    t["canonical_inverse_depth"] = rng.uniform(0.1, 1.0, (1, 1, h, w)).astype(np.float32)
    return t


def test_compute_stats_matches_the_reference_definition():
    s = parity.compute_stats(np.array([1.0, 2.0, 0.0]), np.array([1.5, 2.0, 0.0]))
    assert s.mean_abs == pytest.approx(0.5 / 3) and s.max_abs == pytest.approx(0.5) and s.max_rel == pytest.approx(0.5 / 1.5)
    assert parity.compute_stats(np.array([1e-3]), np.array([0.0])).max_rel == pytest.approx(1e-3 / 1e-6)  # |ref| floor 1e-6
    with pytest.raises(ValueError):
        parity.compute_stats(np.zeros(3), np.zeros(4))


def test_schema_loader_and_error_messages():
    t = _dump()
    ref = parity.load_reference_dump(t)
    assert ref.depth.shape == (6, 8) and len(ref.encoder_features) == 3 and len(ref.decoder_fusions) == 1
    assert set(ref.optional) == {"canonical_inverse_depth"}
    bad = dict(t)
    del bad["fovy"]
    with pytest.raises(ValueError, match="missing `fovy` tensor"):
        parity.load_reference_dump(bad)
    bad = dict(t)
    bad["metric_depth"] = t["metric_depth"][:, :, 0]
    with pytest.raises(ValueError, match=r"expected torch depth shape \[H, W, 1\]"):
        parity.load_reference_dump(bad)


def test_verdict_uses_the_reference_thresholds():
    t = _dump()
    ref = parity.load_reference_dump(t)
    d = ref.depth.copy()
    taps = {"encoder_feature_0": t["encoder_feature_0"] + 1e-3, "decoder_fusion_0": np.zeros((1, 4, 5, 5), np.float32)}
    ok = parity.compare(ref, d * (1 + 2e-4), 53.0005, 40.0005, taps)
    assert ok.ok and ok.depth.max_rel < 5e-3
    txt = "\n".join(ok.lines)
    assert "encoder_feature_0: mean abs=0.001000" in txt and "decoder_fusion_0: shape mismatch" in txt
    assert "encoder_feature_1: no engine tap" in txt
    assert not parity.compare(ref, d * 1.006, 53.0, 40.0).ok          # max rel 6e-3 > 5e-3
    assert not parity.compare(ref, d, 53.002, 40.0).ok                # fovx diff 2e-3 > 1e-3
    with pytest.raises(ValueError, match="depth shape mismatch"):
        parity.compare(ref, d[:-1], 53.0, 40.0)


def test_replay_report_follows_the_harness_lines():
    """`compare_decoder_with_reference` (example/correctness.rs:530-660): "[Replay] <label>: mean abs=…" per tensor, the position of
    the largest difference once it exceeds 1e-3, and the harness's wording for missing / mis-shaped tensors."""
    t = _dump()
    rng = np.random.default_rng(1)
    t["decoder_feature"] = rng.standard_normal((1, 4, 6, 8)).astype(np.float32)
    t["head_conv1"] = rng.standard_normal((1, 2, 12, 16)).astype(np.float32)
    ref = parity.load_reference_dump(t)
    fus = [f.copy() for f in ref.decoder_fusions]
    assert len(fus) >= 1
    fus[0] = fus[0].copy()
    fus[0].reshape(-1)[3] += 0.5  # a difference above the harness's 1e-3 print threshold
    head = {"head_conv1": t["head_conv1"] + 1e-5, "head_relu": np.zeros((1, 2, 12, 16), np.float32)}
    lines = parity.replay_report(ref, t["decoder_feature"], np.zeros((1, 4, 3, 4), np.float32), fus, head)
    assert lines[0].startswith("[Replay] Decoder feature: mean abs=0.000000, max abs=0.000000")
    assert lines[1] == "[Replay] Torch reference missing Decoder lowres feature; skipping."
    assert lines[2].startswith("[Replay] Decoder fusion 0: ") and "max abs=0.5" in lines[2]
    assert lines[3].startswith("[Replay] Decoder fusion 0 max diff at ") and "diff=0.500000" in lines[3]
    assert any(l.startswith("[Replay] Head head_conv1: ") for l in lines)
    assert "[Replay] Torch reference missing Head head_relu; skipping." in lines
    short = parity.replay_report(ref, t["decoder_feature"][:, :2], t["decoder_feature"], fus[:-1] if len(fus) > 1 else [])
    assert "shape mismatch" in short[0]
    if len(fus) > 1:
        assert short[-1].startswith("[Replay] fusion count mismatch")


def test_full_size_oracle_frames_come_from_child_processes(tmp_path):
    """tools/gpu_diag.py computes the full-size oracle frames of the GPU suite concurrently in child processes
    (tools/oracle_frames.py): a registered frame is produced by a child, is the frame the in-process oracle gives, and lands
    under the cache key the GPU-side runner asks for. (CPU-sized stand-in: Depth-Anything-v3 `small` at 70 x 70.)"""
    import os
    import sys
    import torch
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    sys.path.insert(0, tools)
    try:
        import gpu_diag as diag
        import oracle_frames
    finally:
        sys.path.remove(tools)
    from burn_depth_amd import weights as Wt
    from burn_depth_amd.config import DepthAnything3Config
    cfg = DepthAnything3Config.small()
    cfg.image_size = 70
    key = ("da3", diag.cfg_key(cfg), 1, Wt.INIT_PARITY, False, "seeded")
    diag._ORACLE_CACHE.pop(key, None)
    diag.prefetch_da3(cfg, 1)
    assert key in diag._PENDING_JOBS
    got = diag.cached(key, lambda: pytest.fail("the registered frame was recomputed in-process"))
    assert key not in diag._PENDING_JOBS and key in diag._ORACLE_CACHE
    want = oracle_frames.da3_frame("small", 70, 1, Wt.INIT_PARITY)["out"]
    assert set(got) == set(want) and torch.equal(got["depth"], want["depth"])
    # a failing child surfaces in the test that asks for its frame
    diag.register_frame(("da3", "broken"), {"fp32": "da3:no_such_variant:70:1:1"})
    with pytest.raises(RuntimeError, match="failed in its child process"):
        diag.cached(("da3", "broken"), lambda: None)
    assert diag.host_cpus() >= 1 and diag.host_mem_gb() > 0
