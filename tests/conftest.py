import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "fullsize: compares with a full-size CPU-oracle frame (30-90 s of host time); runs last, its frame "
                                       "is computed on a background thread under the earlier tests' GPU work (tools/gpu_diag.py prefetch)")


def pytest_sessionstart(session):
    """torch's CPU thread pool sized to the cores this process may really use: the GPU boxes show 256 CPUs under a 16-core cgroup
    quota, and the oracle's GEMMs run at 0.67 of their rate on the default 128 threads (profiles/r05_host_probe.txt)."""
    try:
        import torch
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from oracle_frames import host_cpus
        n = host_cpus()
        if n < torch.get_num_threads():
            torch.set_num_threads(n)
    except Exception:  # noqa: BLE001 -- a sizing hint, never a reason to fail collection
        pass


def pytest_collection_modifyitems(config, items):
    """`fullsize` tests go to the end of the run (their relative order kept): their oracle frames are queued at session start
    and computed while the other tests keep the GPU busy."""
    tail = [i for i in items if i.get_closest_marker("fullsize")]
    if tail:
        items[:] = [i for i in items if not i.get_closest_marker("fullsize")] + tail


@pytest.fixture(scope="session")
def repo_root():
    return ROOT


def pytest_runtest_logreport(report):
    """One unbuffered stderr line per finished test. The GPU suite runs for several minutes and a harness that watches a
    piped stdout sees nothing of pytest's progress dots until the pipe's buffer fills or the run ends; a watchdog that reads
    silence as a hang then kills a healthy run. The longest single test (the full-size oracle comparison) is ~100 s."""
    if report.when == "call" or (report.when == "setup" and report.outcome != "passed"):
        try:
            sys.__stderr__.write(f"[test] {report.outcome:7s} {report.duration:7.1f}s {report.nodeid}\n")
            sys.__stderr__.flush()
        except Exception:
            pass
