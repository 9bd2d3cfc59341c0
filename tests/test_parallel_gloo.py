"""world_size-2 rehearsal of the N>1 path on CPU (gloo): image sharding, scatter of inputs, gather of
depth maps to rank 0 -- the same `burn_depth_amd.parallel` functions the RCCL path uses."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from burn_depth_amd.parallel import gather_depth, scatter_images, shard_range


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_total, H, W = 4, 6, 5
        images = torch.arange(n_total * 3 * H * W, dtype=torch.float32).reshape(n_total, 3, H, W) if rank == 0 else None
        like = torch.empty(1, 3, H, W)
        mine = scatter_images(images, n_total, like, src=0)
        b, e = shard_range(n_total, rank, world)
        assert mine.shape[0] == e - b
        full = torch.arange(n_total * 3 * H * W, dtype=torch.float32).reshape(n_total, 3, H, W)
        assert torch.equal(mine, full[b:e])
        # stand-in for the per-rank engine: a "depth" that depends only on the rank's own images
        depth = mine.sum(1) + 1.0
        gathered = [torch.empty_like(depth) for _ in range(world)] if rank == 0 else None
        gather_depth(depth, gathered, dst=0)
        if rank == 0:
            got = torch.cat(gathered, 0)
            assert torch.equal(got, full.sum(1) + 1.0)
        # max-over-ranks timing reduction used by bench.py
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert t.item() == float(world)
        q.put((rank, "ok"))
    except Exception as ex:  # noqa: BLE001
        q.put((rank, f"FAIL {type(ex).__name__}: {ex}"))
    finally:
        dist.destroy_process_group()


def test_shard_range_is_a_balanced_partition():
    for n in (1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(120)
def test_scatter_gather_world_size_2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=100) for _ in range(world)]
    for p in procs:
        p.join(30)
    assert sorted(results) == [(0, "ok"), (1, "ok")], results
