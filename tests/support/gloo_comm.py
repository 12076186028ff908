"""TEST DOUBLE: a gloo transport for host tensors.

The framework's only transport is RCCL (GPU memory).  The multi-process CPU tests still want
to drive the stock-torch golden backend and its Python sync strategies with world_size > 1
on a machine without a GPU; this class gives them the ``Communicator`` interface over
``torch.distributed`` / gloo, like the reference's own process group
(/root/reference/part2/2a/main.py:197).  Nothing in the package imports it, and what passes
with it says nothing about the native engine's reducer (tests/test_multirank_gpu.py does).
"""
from __future__ import annotations

import datetime
import os

from distributed_machine_learning_amd.parallel.comm import Communicator, _REDUCE_OPS


class HostCommunicator(Communicator):
    """gloo over TCP for host tensors."""

    def __init__(self, master_ip: str, rank: int, world_size: int, timeout_s: float = 300.0):
        import torch.distributed as dist
        self._dist = dist
        self._owns_group = not dist.is_initialized()
        if self._owns_group:
            if os.environ.get("TORCHELASTIC_RUN_ID") and "MASTER_ADDR" in os.environ:
                init = "env://"
            else:
                init = "tcp://" + master_ip
            dist.init_process_group("gloo", init_method=init, world_size=world_size, rank=rank,
                                    timeout=datetime.timedelta(seconds=timeout_s))
        self.rank = dist.get_rank()
        self.world_size = dist.get_world_size()
        self.backend = dist.get_backend()

    def _finish(self, tensor, op):
        if op == "mean":
            tensor.div_(self.world_size)

    def all_reduce(self, tensor, op="sum"):
        assert op in _REDUCE_OPS
        self._dist.all_reduce(tensor, op=self._dist.ReduceOp.SUM)
        self._finish(tensor, op)

    def all_reduce_async(self, tensor, op="sum"):
        assert op in _REDUCE_OPS
        work = self._dist.all_reduce(tensor, op=self._dist.ReduceOp.SUM, async_op=True)
        comm = self

        class _Handle:
            def wait(self_inner):
                work.wait()
                comm._finish(tensor, op)
        return _Handle()

    def gather(self, tensor, slabs, root=0):
        self._dist.gather(tensor, list(slabs) if self.rank == root else None, dst=root)

    def scatter(self, out, chunks, root=0):
        self._dist.scatter(out, list(chunks) if self.rank == root else None, src=root)

    def broadcast(self, tensor, root=0):
        self._dist.broadcast(tensor, src=root)

    def barrier(self):
        self._dist.barrier()

    def destroy(self):
        if self._owns_group and self._dist.is_initialized():
            self._dist.destroy_process_group()
