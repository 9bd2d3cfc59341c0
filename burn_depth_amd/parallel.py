"""Multi-GPU plumbing for the data-parallel hot path: one process per GPU,
torch.distributed over RCCL ("nccl" backend on ROCm) / xGMI.

Images of a batch never interact (B is a pure batch dimension through split, ViT, merge and
decoder; encoder.rs:216-225,249-255), so the path shards by image with NO collective inside the
model.  The only exchanges are
  * a one-time broadcast of the fp32 weight arena from rank 0 (then every rank re-packs its own
    MFMA operand copies), and
  * a gather of the finished depth maps to rank 0.
Both also run on the gloo backend (CPU tensors), which is how the N>1 path is tested without GPUs.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split of `n_items` images over `world` ranks: [begin, end)."""
    base, rem = divmod(n_items, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


class _ArenaView:
    """Zero-copy torch view of raw device memory through __cuda_array_interface__."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def broadcast_weights(model, src: int = 0) -> None:
    """RCCL broadcast of the fp32 master weights (in place), then re-pack on every rank."""
    ptr, nbytes = model.weight_arena()
    view = torch.as_tensor(_ArenaView(ptr, nbytes), device=torch.device("cuda", model.device.ordinal))
    chunk = 1 << 30  # 1 GiB buckets: large enough to be link-bound, small enough for the int32 counts
    for off in range(0, nbytes, chunk):
        dist.broadcast(view[off:off + chunk], src=src)
    torch.cuda.synchronize()
    model.commit_weights()


def gather_depth(depth: torch.Tensor, gathered: Optional[List[torch.Tensor]], dst: int = 0) -> None:
    """Depth maps of every rank -> rank `dst` (works for nccl and gloo)."""
    dist.gather(depth, gathered if dist.get_rank() == dst else None, dst=dst)


def scatter_images(images: Optional[torch.Tensor], n_total: int, like: torch.Tensor, src: int = 0) -> torch.Tensor:
    """Rank `src` holds [n_total,3,H,W]; every rank receives its shard (equal shards required)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    assert n_total % world == 0, "scatter_images needs equal shards"
    per = n_total // world
    out = torch.empty((per,) + tuple(like.shape[1:]), dtype=like.dtype, device=like.device)
    chunks = list(images.split(per, 0)) if rank == src else None
    dist.scatter(out, chunks, src=src)
    return out


class NativeComm:
    """The library's own RCCL communicator (include/mi_depth.h md_comm_*): what a reference-side (Rust) host reaches through
    the FFI layer -- weight broadcast, image scatter and depth gather as grouped ncclSend / ncclRecv on a HIP stream, without
    torch.distributed on the data path. torch.distributed (or any other channel) is only needed to hand the 128-byte
    rendezvous id from rank 0 to the other ranks."""

    def __init__(self, device, unique_id: bytes, world_size: int, rank: int):
        from . import _lib
        self._L = _lib
        self._lib = _lib.load()
        assert len(unique_id) == _lib.MD_COMM_ID_BYTES
        buf = (C.c_uint8 * _lib.MD_COMM_ID_BYTES).from_buffer_copy(unique_id)
        h = C.c_void_p()
        _lib.check(self._lib.md_comm_init_rank(device.handle, C.cast(buf, C.c_void_p), int(world_size), int(rank), C.byref(h)))
        self._h, self.world, self.rank, self.device = h, int(world_size), int(rank), device

    @staticmethod
    def unique_id() -> bytes:
        from . import _lib
        buf = (C.c_uint8 * _lib.MD_COMM_ID_BYTES)()
        _lib.check(_lib.load().md_comm_unique_id(C.cast(buf, C.c_void_p)))
        return bytes(buf)

    @staticmethod
    def from_torch_distributed(device) -> "NativeComm":
        """Rank 0 draws the id and the (already initialised) torch.distributed group carries it to the others."""
        box = [NativeComm.unique_id() if dist.get_rank() == 0 else None]
        if dist.get_world_size() > 1:
            dist.broadcast_object_list(box, src=0)
        return NativeComm(device, box[0], dist.get_world_size(), dist.get_rank())

    def broadcast_weights(self, model, root: int = 0) -> None:
        self._L.check(self._lib.md_comm_broadcast_weights(self._h, model._h, int(root)))

    def scatter_images(self, all_images: Optional[torch.Tensor], shard: torch.Tensor, root: int = 0, stream: int = 0) -> None:
        """root: all_images = [world * B, 3, H, W] on this GPU; every rank receives [B, 3, H, W] into `shard`."""
        p = C.c_void_p(all_images.data_ptr()) if all_images is not None else None
        self._L.check(self._lib.md_comm_scatter_images(self._h, p, C.c_void_p(shard.data_ptr()), shard.numel(), int(root), C.c_void_p(stream)))

    def gather_depth(self, shard: torch.Tensor, all_depth: Optional[torch.Tensor], root: int = 0, stream: int = 0) -> None:
        p = C.c_void_p(all_depth.data_ptr()) if all_depth is not None else None
        self._L.check(self._lib.md_comm_gather_depth(self._h, C.c_void_p(shard.data_ptr()), p, shard.numel(), int(root), C.c_void_p(stream)))

    def infer_tiles(self, model, x: Optional[torch.Tensor], shape, root: int = 0, stream: int = 0):
        """Tile-parallel `DepthPro::infer` of ONE call (md_comm_depth_pro_infer_tiles): every rank calls it with its replica of
        the same weights; `x` [B,3,H,W] (host or device) on the root, None elsewhere; `shape` = (B, H, W) on every rank. The
        root returns a DepthProInference (bit-identical to `model.infer(x)` on one GPU), the other ranks None."""
        from .depth_pro import DepthProInference
        B, H, W = (int(v) for v in shape)
        stream = stream or torch.cuda.current_stream(torch.device("cuda", model.device.ordinal)).cuda_stream  # like model.infer
        if self.rank == root:
            x = x.contiguous().to(torch.float32)
            assert tuple(x.shape) == (B, 3, H, W)
            dev = torch.device("cuda", model.device.ordinal)
            depth = torch.empty((B, H, W), dtype=torch.float32, device=dev)
            focal, fovx, fovy = (torch.empty((B,), dtype=torch.float32, device=dev) for _ in range(3))
            in_kind = self._L.MD_MEM_DEVICE if x.is_cuda else self._L.MD_MEM_HOST
            self._L.check(self._lib.md_comm_depth_pro_infer_tiles(self._h, model._h, C.c_void_p(x.data_ptr()), B, H, W, in_kind,
                                                                  C.c_void_p(depth.data_ptr()), C.c_void_p(focal.data_ptr()),
                                                                  C.c_void_p(fovx.data_ptr()), C.c_void_p(fovy.data_ptr()),
                                                                  self._L.MD_MEM_DEVICE, int(root), C.c_void_p(stream)))
            return DepthProInference(depth, focal, fovx, fovy)
        self._L.check(self._lib.md_comm_depth_pro_infer_tiles(self._h, model._h, None, B, H, W, self._L.MD_MEM_DEVICE, None, None, None, None,
                                                              self._L.MD_MEM_DEVICE, int(root), C.c_void_p(stream)))
        return None

    def destroy(self) -> None:
        if self._h:
            self._lib.md_comm_destroy(self._h)
            self._h = None
