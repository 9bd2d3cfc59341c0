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
