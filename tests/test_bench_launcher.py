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


@pytest.mark.timeout(300)
def test_native_comm_bookkeeping_walks_clean_on_eight_ranks():
    """`bench.py --gpus 8 --dry-run --native-comm`: the double-buffer / event bookkeeping of the native RCCL path
    (burn_depth_amd.parallel.NativePipeline -- the object the GPU run drives with torch.cuda streams / events and md_comm_*)
    on CPU stand-ins over 8 gloo ranks: every gathered map is the map of the scattered image, and the happens-before checker
    saw no buffer access that a wait does not order."""
    p, lines = _run(["--gpus", "8", "--dry-run", "--native-comm", "--preset", "tiny", "--steps", "5", "--warmup", "2", "--batch", "1"])
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["finite_output"] is True
    assert line["native_comm"]["races"] == [] and line["native_comm"]["steps_walked"] == 7


def test_happens_before_checker_catches_a_missing_wait():
    """The checker itself: the same pipeline with ONE wait removed (the scatter into an input buffer no longer waits for the
    infer that read it / the infer no longer waits for its shard) must report races; the shipped pipeline reports none."""
    import torch
    from burn_depth_amd.parallel import HappensBefore, NativePipeline

    def walk(sabotage, do_scatter=True):
        hb = HappensBefore()
        compute = hb.stream("compute")
        xs, depths = [torch.zeros(1) for _ in range(2)], [torch.zeros(1) for _ in range(2)]

        class Comm:
            def scatter_images(self, a, shard, root, stream):
                i = [j for j, t in enumerate(xs) if t is shard][0]
                hb.access(stream, "scatter", writes=(f"xs[{i}]",))

            def gather_depth(self, shard, a, root, stream):
                i = [j for j, t in enumerate(depths) if t is shard][0]
                hb.access(stream, "gather", reads=(f"depths[{i}]",))

        def infer(slot, cur):
            hb.access(cur, "infer", reads=(f"xs[{slot}]",), writes=(f"depths[{slot}]",))

        class NoWaitEvent(HappensBefore.Event):
            def record(self, stream):  # an event whose record is lost: every wait on it orders nothing
                self.clock = {}
        n = {"i": 0}

        def make_event():
            n["i"] += 1
            return NoWaitEvent() if sabotage and n["i"] == sabotage else hb.event()
        pipe = NativePipeline(Comm(), infer, 2, 0, 0, do_scatter, True, None, xs, depths, [None, None], make_stream=lambda: hb.stream("comm"),
                              make_event=make_event, current_stream=lambda: compute)
        for _ in range(6):
            pipe.step()
        return hb.races
    assert walk(0) == []
    assert any("reads xs[0]" in r for r in walk(1))          # ev_sc[0] lost: infer reads a shard that may not have arrived
    assert any("overwrites xs[0]" in r for r in walk(3))     # ev_inf[0] lost: the next scatter overwrites an input still being read
    # ev_ga[0] lost: infer overwrites a depth map still being gathered. With the scatter on, the wait on the NEXT shard (issued
    # behind that gather on the in-order side stream) already orders it; the wait carries the ordering when nothing is scattered
    assert walk(5) == [] and walk(0, do_scatter=False) == []
    assert any("overwrites depths[0]" in r for r in walk(5, do_scatter=False))
