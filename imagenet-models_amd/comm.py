"""RCCL gradient exchange through the C ABI (include/gaext.h "Gradient exchange"; replaces NativeDDP, GA/train.py:514).

One `NativeComm` per process (= per GPU).  The 128-byte communicator id is made by rank 0 inside the library
(ga_comm_unique_id) and handed to the other ranks over whatever process group torch.distributed already has (its store is
only a rendezvous channel here: no tensor ever goes through torch.distributed on this path).  With WORLD_SIZE == 1 no process
group is needed at all -- which is how the one-GPU test box exercises real RCCL calls.

`TrainStep(..., comm=NativeComm())` issues every gradient bucket as ga_allreduce_bucket on a dedicated side stream behind
an event of the backward stream, so the exchange of bucket k overlaps the backward segments of buckets k+1.. exactly as the
torch.distributed path does."""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib as L


class NativeComm:
    def __init__(self, wire='fp32', group=None):
        if wire not in ('fp32', 'bf16'):
            raise ValueError("wire: 'fp32' or 'bf16'")
        self.wire = L.GA_F32 if wire == 'fp32' else L.GA_BF16
        self.lib = L.load()
        if dist.is_available() and dist.is_initialized():
            self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        else:
            self.rank, self.world = 0, 1
        ident = C.create_string_buffer(128)
        if self.rank == 0:
            L.check(self.lib.ga_comm_unique_id(ident), 'ga_comm_unique_id')
        if self.world > 1:
            box = [bytes(ident.raw)]
            dist.broadcast_object_list(box, src=0, group=group)      # rendezvous only: 128 bytes
            ident = C.create_string_buffer(box[0], 128)
        self.handle = C.c_void_p()
        L.check(self.lib.ga_comm_init(C.byref(self.handle), self.rank, self.world, ident), 'ga_comm_init')
        self.stream = torch.cuda.Stream()
        self._ws = None
        self._done = torch.cuda.Event()

    def _workspace(self, n):
        need = int(self.lib.ga_allreduce_workspace(n, self.wire))
        if need == 0:
            return None, 0
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device='cuda')
        return self._ws.data_ptr(), self._ws.numel()

    def allreduce(self, flat, scale=1.0, stream=None):
        """in-place sum over ranks (* scale) of a contiguous fp32 CUDA tensor, enqueued on `stream` (default: the comm stream)"""
        assert flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous()
        s = (stream or self.stream).cuda_stream
        ws, nb = self._workspace(flat.numel())
        L.check(self.lib.ga_allreduce_bucket(self.handle, flat.data_ptr(), flat.numel(), self.wire, float(scale), ws, nb, s),
                'ga_allreduce_bucket')

    def reduce_scatter(self, full, shard, scale=1.0, stream=None):
        assert full.numel() == shard.numel() * self.world
        L.check(self.lib.ga_reduce_scatter_bucket(self.handle, full.data_ptr(), shard.data_ptr(), shard.numel(), float(scale),
                                                  (stream or self.stream).cuda_stream), 'ga_reduce_scatter_bucket')

    def allgather(self, shard, full, stream=None):
        assert full.numel() == shard.numel() * self.world
        L.check(self.lib.ga_allgather_bucket(self.handle, shard.data_ptr(), full.data_ptr(), shard.numel(),
                                             (stream or self.stream).cuda_stream), 'ga_allgather_bucket')

    def broadcast(self, flat, root=0, stream=None):
        assert flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous()
        L.check(self.lib.ga_comm_broadcast(self.handle, flat.data_ptr(), flat.numel(), root, (stream or self.stream).cuda_stream),
                'ga_comm_broadcast')

    def after(self, main):
        """the comm stream waits for everything issued on `main` so far"""
        ev = torch.cuda.Event()
        ev.record(main)
        self.stream.wait_event(ev)

    def join(self, main):
        """`main` waits for everything issued on the comm stream so far"""
        self._done.record(self.stream)
        main.wait_event(self._done)

    def close(self):
        if self.handle:
            self.lib.ga_comm_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
