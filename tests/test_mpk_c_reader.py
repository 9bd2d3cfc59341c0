"""The C ABI's Burn-record (`.mpk`) reader on the CPU (no device call): `md_checkpoint_info` / `md_checkpoint_read_tensor` walk the
same C++ MessagePack reader `md_depth_pro_load` / `md_da3_load` use (`DepthPro::load`'s argument, depth_pro/mod.rs:193-208). The
files come from tests/mpk_fixture.py, a byte-level builder that shares no code with the reader. The GPU half (load a record, infer,
compare with the seeded model; the plain-C caller) is in tests/test_gpu_parity.py."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import mpk_fixture  # noqa: E402

from burn_depth_amd import _lib, weights as Wt  # noqa: E402
from burn_depth_amd.config import DepthAnything3Config, DepthProConfig  # noqa: E402


def directory(path):
    lib = _lib.load()
    burn = C.c_int(-1)
    n = lib.md_checkpoint_info(path.encode(), -1, None, None, None, None, C.byref(burn))
    if n < 0:
        _lib.check(n)
    out = {}
    for i in range(n):
        name, dt, rank, shape = C.c_char_p(), C.c_char_p(), C.c_int(), (C.c_int64 * 8)()
        assert lib.md_checkpoint_info(path.encode(), i, C.byref(name), C.byref(dt), C.byref(rank), C.byref(shape), None) == n
        out[name.value.decode()] = (dt.value.decode(), tuple(shape[j] for j in range(rank.value)))
    return out, bool(burn.value)


def values(path, name, count):
    a = np.empty(count, np.float32)
    _lib.check(_lib.load().md_checkpoint_read_tensor(path.encode(), name.encode(), a.ctypes.data_as(C.c_void_p), count))
    return a


@pytest.mark.parametrize("dtype", ["F16", "F32", "BF16"])
def test_burn_record_directory_and_values(tmp_path, dtype):
    cfg = DepthProConfig.tiny_test()
    W = Wt.generate_depth_pro_weights(cfg, 3, Wt.INIT_REFERENCE)
    path = str(tmp_path / "depth_pro.mpk")
    mpk_fixture.write_record(path, W, dtype=dtype)
    d, burn = directory(path)
    assert burn
    assert set(d) == set(W)  # Vec<Module> arrays, {id, param} wrappers, None / int / float / bool leaves skipped
    for k, (dt, shape) in d.items():
        assert dt == dtype
        want = tuple(W[k].shape)
        assert shape == (want[::-1] if len(want) == 2 else want), k  # nn::Linear weights are [d_input, d_output] in a Burn record
    rnd = {"F16": lambda a: a.astype(np.float16).astype(np.float32), "F32": lambda a: a,
           "BF16": lambda a: (((a.view(np.uint32) + 0x7FFF + ((a.view(np.uint32) >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)}[dtype]
    for k in ("encoder.patch_encoder.blocks.1.mlp.fc1.weight", "head.conv0.weight", "encoder.patch_encoder.pos_embed", "fov.encoder_proj.weight",
              "encoder.upsample_lowres.bias"):
        a = np.ascontiguousarray(W[k].T if W[k].ndim == 2 else W[k], np.float32)
        assert np.array_equal(values(path, k, a.size), rnd(a).reshape(-1)), k


def test_depth_anything3_record_directory(tmp_path):
    cfg = DepthAnything3Config.tiny_dual_test()
    W = Wt.generate_da3_weights(cfg, 0, Wt.INIT_REFERENCE)
    path = str(tmp_path / "da3.mpk")
    mpk_fixture.write_record(path, W)
    d, burn = directory(path)
    assert burn and set(d) == set(W)
    k = "camera_decoder.fc_qvec.weight"
    assert d[k][1] == tuple(W[k].shape)[::-1]


def test_safetensors_container_still_dispatches(tmp_path):
    cfg = DepthProConfig.tiny_test()
    W = Wt.generate_depth_pro_weights(cfg, 3, Wt.INIT_REFERENCE)
    path = str(tmp_path / "c.safetensors")
    Wt.save_container(path, W, metadata=Wt.config_metadata(cfg), dtype="F16")
    d, burn = directory(path)
    assert not burn and set(d) == set(W)
    k = "encoder.patch_encoder.blocks.0.attn.qkv.weight"
    assert d[k] == ("F16", tuple(W[k].shape))  # the engine's own container keeps [out, in]
    assert np.array_equal(values(path, k, W[k].size), W[k].astype(np.float16).astype(np.float32).reshape(-1))


def test_malformed_records_are_format_errors(tmp_path):
    cfg = DepthProConfig.tiny_test()
    W = Wt.generate_depth_pro_weights(cfg, 3, Wt.INIT_REFERENCE)
    good = mpk_fixture.build_record(W)
    lib = _lib.load()

    def code(data):
        p = str(tmp_path / "x.mpk")
        open(p, "wb").write(data)
        return lib.md_checkpoint_info(p.encode(), -1, None, None, None, None, None)
    assert code(good) == len(W)
    assert code(good[:len(good) // 2]) == _lib.MD_ERR_FORMAT                      # truncated inside the record
    assert code(good[:-3]) == _lib.MD_ERR_FORMAT                                   # truncated inside the last leaf
    assert code(b"\x82" + mpk_fixture._str("metadata") + b"\x80" + mpk_fixture._str("nope") + b"\xc0") == _lib.MD_ERR_FORMAT  # no `item`
    assert code(b"\x81" + mpk_fixture._str("item") + b"\x80" + b"\x00" * 8) == _lib.MD_ERR_FORMAT  # no tensors
    assert code(bytes(range(48, 80))) == _lib.MD_ERR_FORMAT                        # neither format
    assert lib.md_checkpoint_info(str(tmp_path / "missing.mpk").encode(), -1, None, None, None, None, None) == _lib.MD_ERR_IO
    # a tensor whose bytes are not a MessagePack bin (an old array-of-numbers record) is refused, not misread
    bad = (b"\x81" + mpk_fixture._str("item") + b"\x81" + mpk_fixture._str("w") + b"\x82" + mpk_fixture._str("bytes") + b"\x92\x01\x02" +
           mpk_fixture._str("shape") + b"\x91\x02")
    assert code(bad + b"\x00" * 8) == _lib.MD_ERR_FORMAT


def test_python_twin_reads_the_fixture(tmp_path):
    """burn_depth_amd/mpk.py (the importer's reader) against the same byte-level fixture: names, [out, in] layout, values."""
    from burn_depth_amd import mpk
    cfg = DepthProConfig.tiny_test()
    W = Wt.generate_depth_pro_weights(cfg, 3, Wt.INIT_REFERENCE)
    path = str(tmp_path / "depth_pro.mpk")
    mpk_fixture.write_record(path, W)
    got = mpk.read_mpk(path)
    assert set(got) == set(W)
    for k in ("encoder.patch_encoder.blocks.1.mlp.fc1.weight", "head.conv0.weight", "fov.encoder_proj.weight"):
        assert got[k].shape == W[k].shape and np.array_equal(got[k], W[k].astype(np.float16).astype(np.float32)), k


def test_checkpoint_readers_survive_mutated_files_under_sanitizers(tmp_path):
    """The two checkpoint parsers read untrusted files: build them for the host with AddressSanitizer + UBSan (GPU sanitizers do not
    exist on the pool; these readers are plain C++) and feed them truncated and byte-flipped mutants of a valid record of each format
    (tests/native/fuzz_checkpoint_readers.cpp). Every mutant must be parsed or refused -- no out-of-bounds access, no overflow."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no host compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "burn_depth_amd", "csrc")
    exe = str(tmp_path / "fuzz_readers")
    build = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                            "-fno-omit-frame-pointer", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + csrc,
                            os.path.join(root, "tests", "native", "fuzz_checkpoint_readers.cpp"), os.path.join(csrc, "md_weights.cpp"),
                            os.path.join(csrc, "md_common.cpp"), "-o", exe], capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-3000:]
    cfg = DepthAnything3Config.tiny_dual_test()
    W = Wt.generate_da3_weights(cfg, 0)
    keep = [k for k in W if k.startswith("camera_") or ".blocks.0." in k or "patch_embed" in k or k.endswith("cls_token")]
    small = {k: W[k] for k in keep}
    assert len(small) > 40
    mpk, st = str(tmp_path / "r.mpk"), str(tmp_path / "r.safetensors")
    mpk_fixture.write_record(mpk, small)
    Wt.save_container(st, small, dtype="F16")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    for path, seed in ((mpk, 11), (st, 12)):
        run = subprocess.run([exe, path, "400", str(seed)], capture_output=True, text=True, timeout=900, env=env)
        assert run.returncode == 0, (run.stdout[-500:], run.stderr[-3000:])
        assert "AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr, run.stderr[-3000:]
        parsed, rejected = [int(run.stdout.split(w)[1].split()[0].strip("(),")) for w in ("parsed", "rejected")]
        assert parsed > 20 and rejected > 20, run.stdout   # both outcomes are exercised
