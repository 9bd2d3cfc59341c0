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

    def ranks_seen(self) -> int:
        """`ncclCommCount` of the communicator: the ranks RCCL itself sees."""
        n = C.c_int()
        self._L.check(self._lib.md_comm_count(self._h, C.byref(n)))
        return int(n.value)

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


class NativePipeline:
    """The per-step data path of `bench.py --native-comm` (BASELINE config 4: "including scatter + gather"): the root scatters the
    global batch, every rank infers its shard, the root gathers the depth maps -- scatter of step t+1 and gather of step t-1 on a
    side HIP stream under step t's inference, ordered against the compute stream by events, over `nbuf` (2) input / depth buffers.

    Everything it touches is injected, so the SAME bookkeeping runs on the GPU (torch.cuda streams / events, `NativeComm`,
    `DepthPro.infer_into`) and in `bench.py --dry-run --native-comm` on CPU stand-ins that record every buffer access with the
    stream clocks of `HappensBefore` below: a missing wait is a reported race there, not a once-in-a-while corrupted frame on 8 GPUs.

      comm.scatter_images(all | None, shard, root, stream) / comm.gather_depth(shard, all | None, root, stream)
      infer(slot, stream)               runs the model on xs[slot] -> depths[slot] on `stream`
      make_stream() / make_event() / current_stream()
      a stream has wait_event(ev) and synchronize(); an event has record(stream)
    """

    def __init__(self, comm, infer, nbuf, rank, root, do_scatter, do_gather, global_in, xs, depths, gathered_flat, make_stream, make_event,
                 current_stream, stream_handle=lambda s: s):
        self.comm, self.infer, self.nbuf, self.rank, self.root = comm, infer, nbuf, rank, root
        self.do_scatter, self.do_gather = do_scatter, do_gather
        self.global_in, self.xs, self.depths, self.gathered_flat = global_in, xs, depths, gathered_flat
        self.current_stream, self.handle = current_stream, stream_handle
        self.cstream = make_stream()
        self.ev_sc = [make_event() for _ in range(nbuf)]   # shard `slot` has arrived
        self.ev_inf = [make_event() for _ in range(nbuf)]  # infer of buffer `slot` has finished (input free, depth ready)
        self.ev_ga = [make_event() for _ in range(nbuf)]   # depth buffer `slot` has been gathered
        self.used_inf, self.used_ga = [False] * nbuf, [False] * nbuf
        self.pending_scatter = None
        self.k = 0

    def _scatter(self, slot):
        if self.used_inf[slot]:
            self.cstream.wait_event(self.ev_inf[slot])  # the previous infer on this input buffer has finished
        self.comm.scatter_images(self.global_in if self.rank == self.root else None, self.xs[slot], root=self.root, stream=self.handle(self.cstream))
        self.ev_sc[slot].record(self.cstream)
        self.pending_scatter = slot

    def step(self):
        k, nbuf = self.k, self.nbuf
        slot = k % nbuf
        cur = self.current_stream()
        if self.do_scatter:
            if self.pending_scatter is None:
                self._scatter(slot)
            cur.wait_event(self.ev_sc[slot])
            if nbuf > 1:
                self._scatter((k + 1) % nbuf)  # the next step's shard travels while this step computes
            else:
                self.pending_scatter = None
        if self.do_gather and self.used_ga[slot]:
            cur.wait_event(self.ev_ga[slot])  # depth buffer `slot` was handed to a gather nbuf steps ago
        self.infer(slot, cur)
        self.ev_inf[slot].record(cur)
        self.used_inf[slot] = True
        if self.do_gather:
            self.cstream.wait_event(self.ev_inf[slot])
            self.comm.gather_depth(self.depths[slot], self.gathered_flat[slot] if self.rank == self.root else None, root=self.root,
                                   stream=self.handle(self.cstream))
            self.ev_ga[slot].record(self.cstream)
            self.used_ga[slot] = True
        self.k = k + 1

    def drain(self, device_synchronize):
        self.cstream.synchronize()
        device_synchronize()
        self.pending_scatter = None  # a prefetched shard is dropped: the next step scatters its own

    def last_slot(self):
        return (self.k - 1) % self.nbuf


class HappensBefore:
    """Vector-clock race detector for the CPU stand-ins of `NativePipeline`: every stream carries a clock {stream: count},
    recording an event copies it, waiting on an event joins it; an access to a buffer must happen-after the buffer's last
    write (reads and writes) and after every read since (writes). `races` lists what did not."""

    class Stream:
        def __init__(self, hb, name):
            self.hb, self.name, self.clock = hb, name, {name: 0}

        def wait_event(self, ev):
            if ev.clock is None:
                self.hb.races.append(f"{self.name} waits on an event that was never recorded")
                return
            for k, v in ev.clock.items():
                self.clock[k] = max(self.clock.get(k, 0), v)

        def synchronize(self):  # the host has seen everything this stream did: later work on ANY stream happens after it
            self.hb.host_join(self)

        def tick(self):
            self.clock[self.name] += 1
            return (self.name, self.clock[self.name])

    class Event:
        def __init__(self):
            self.clock = None

        def record(self, stream):
            self.clock = dict(stream.clock)

    def __init__(self):
        self.races, self.streams, self.host = [], [], {}
        self.last_write, self.reads = {}, {}

    def stream(self, name):
        s = HappensBefore.Stream(self, name)
        for k, v in self.host.items():
            s.clock[k] = max(s.clock.get(k, 0), v)
        self.streams.append(s)
        return s

    def event(self):
        return HappensBefore.Event()

    def host_join(self, stream):
        for k, v in stream.clock.items():
            self.host[k] = max(self.host.get(k, 0), v)
        for s in self.streams:  # host-ordered: whatever is issued afterwards sees it
            for k, v in self.host.items():
                s.clock[k] = max(s.clock.get(k, 0), v)

    def _after(self, stream, op):
        return op is None or stream.clock.get(op[0], 0) >= op[1]

    def access(self, stream, what, reads=(), writes=()):
        for b in reads:
            if not self._after(stream, self.last_write.get(b)):
                self.races.append(f"{what} on {stream.name} reads {b} without waiting for its writer {self.last_write[b]}")
        for b in writes:
            if not self._after(stream, self.last_write.get(b)):
                self.races.append(f"{what} on {stream.name} overwrites {b} without waiting for its writer {self.last_write[b]}")
            for r in self.reads.get(b, []):
                if not self._after(stream, r):
                    self.races.append(f"{what} on {stream.name} overwrites {b} while reader {r} may still run")
        op = stream.tick()
        for b in reads:
            self.reads.setdefault(b, []).append(op)
        for b in writes:
            self.last_write[b] = op
            self.reads[b] = []
        return op
