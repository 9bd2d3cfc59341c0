"""bench.py's multi-rank entry point, without a GPU: `python bench.py --gpus 2 --dry-run` has to start two ranks by
itself (no WORLD_SIZE in the environment, as the driver calls it), scatter the global batch from rank 0, gather the
per-rank results back and print ONE JSON line that says n_gpus == 2. The dry run swaps the engine for a CPU stand-in and
RCCL for gloo; launcher, process tree, data path and JSON contract are the real ones."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=280, cwd=ROOT)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p, lines


@pytest.mark.timeout(300)
def test_gpus_2_spawns_two_ranks_and_reports_them():
    p, lines = _run(["--gpus", "2", "--dry-run", "--preset", "tiny", "--steps", "3", "--warmup", "1", "--batch", "2"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    assert line["config"]["global_batch"] == 4 and line["config"]["parallelism"] == "dp2"
    assert line["config"]["scatter_inputs_from_rank0"] and line["config"]["gather_depth_to_rank0"]
    assert line["finite_output"] is True          # every gathered map equals the map of the image rank 0 scattered
    assert line["scaling"] == "weak" and line["higher_is_better"] is True and line["value"] > 0


@pytest.mark.timeout(120)
def test_single_rank_dry_run_and_contract_keys():
    p, lines = _run(["--dry-run", "--preset", "tiny", "--steps", "2", "--warmup", "0", "--batch", "1"])
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in line
    assert line["n_gpus"] == 1 and line["vs_baseline"] is None


def test_the_parent_of_a_multi_rank_run_never_imports_torch():
    """`python bench.py --gpus N` must hand over to its child before anything can initialise HIP: the spawn decision sits
    above every torch / engine import, and the module itself imports neither at load time."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("def work_model")]
    assert "import torch" not in head and "burn_depth_amd" not in head.replace("burn_depth_amd.parallel", "")
    main = src[src.index("def main("):]
    assert main.index("spawn_ranks(args.gpus, argv)") < main.index("import torch")
