"""One rank of the two-GPU check of the native RCCL entry points (include/mi_depth.h md_comm_*; SURVEY 8(e): images shard across
the GPUs of one node with an RCCL broadcast of the weights and a gather of the depth maps; second mode: the ViT sequences of ONE call
split over the ranks). Started as a FRESH child process per GPU by tests/test_gpu_parity.py::test_two_gpus_* (never an exec of a
process that has touched the GPU), or by hand on a box with two GPUs:

    UID=$(python -c "from burn_depth_amd.parallel import NativeComm; print(NativeComm.unique_id().hex())")
    python tools/two_gpu_check.py 0 2 $UID & python tools/two_gpu_check.py 1 2 $UID

Rank r drives GPU r. The ranks start from DIFFERENT seeded weights; after md_comm_broadcast_weights every rank must hold rank 0's.
Exit code 0 and a line `rank r OK ranks_seen=2` = passed."""
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank, world, uid = int(sys.argv[1]), int(sys.argv[2]), bytes.fromhex(sys.argv[3])
    import torch
    from burn_depth_amd import weights as Wt
    from burn_depth_amd.config import DepthProConfig
    from burn_depth_amd.depth_pro import DepthPro, Device
    from burn_depth_amd.parallel import NativeComm
    torch.cuda.set_device(rank)
    dev = Device(rank)
    comm = NativeComm(dev, uid, world, rank)
    try:
        assert comm.ranks_seen() == world, comm.ranks_seen()
        cfg = DepthProConfig.tiny_test()
        cfg.max_batch = world
        m = DepthPro.new(dev, cfg, seed=rank, init_scheme=Wt.INIT_PARITY)  # rank r starts from its OWN weights
        comm.broadcast_weights(m, root=0)
        st = torch.cuda.current_stream().cuda_stream
        g = torch.Generator().manual_seed(5)
        x_all = torch.randn(world, 3, 512, 512, generator=g).to(f"cuda:{rank}")  # the same on every rank; only the root's is sent
        shard = torch.zeros(1, 3, 512, 512, device=f"cuda:{rank}")
        comm.scatter_images(x_all if rank == 0 else None, shard, root=0, stream=st)
        torch.cuda.synchronize()
        assert torch.equal(shard, x_all[rank:rank + 1]), "scatter delivered another image"
        d = m.infer(shard).depth
        gathered = torch.zeros(world, *d.shape[1:], device=f"cuda:{rank}") if rank == 0 else None
        comm.gather_depth(d, gathered, root=0, stream=st)
        torch.cuda.synchronize()
        want = m.infer(x_all)  # every rank holds rank 0's weights now: the one-GPU result of the whole batch
        if rank == 0:
            assert torch.equal(gathered, want.depth), "scatter -> infer -> gather differs from the one-GPU batch"
        else:
            assert torch.equal(d, want.depth[rank:rank + 1]), "the broadcast weights differ from rank 0's"
        # tile-parallel mode: ONE call's 37 * B sequences split over the ranks, tokens + hook rows exchanged by ncclSend / ncclRecv
        for root in range(world):
            t = comm.infer_tiles(m, x_all if rank == root else None, (world, 512, 512), root=root)
            torch.cuda.synchronize()
            if rank == root:
                assert torch.equal(t.depth, want.depth) and torch.equal(t.fovx_deg, want.fovx_deg), f"infer_tiles root {root}"
        m.destroy()
        print(f"rank {rank} OK ranks_seen={comm.ranks_seen()}", flush=True)
    finally:
        comm.destroy()


if __name__ == "__main__":
    main()
