"""Process-group lifecycle and the collective surface the sync strategies use.

The reference calls ``dist.init_process_group("gloo", init_method="tcp://"+master_ip,
world_size=nodes, rank=rank)`` and later ``destroy_process_group``
(/root/reference/part2/2a/main.py:194-207, part2/2b/main.py:163-176,
part3/main.py:148-161) and uses exactly four collectives: gather, scatter,
all_reduce(SUM) and DDP's broadcast (part2/2a/main.py:101,114, part2/2b/main.py:106,
part3/main.py:137).

Here the process group is ``RcclCommunicator``: a native RCCL communicator
(csrc/comm/rccl_comm.cpp) bootstrapped by publishing the ``ncclUniqueId`` through the
native TCP key-value store bound to ``--master-ip host:port`` (csrc/comm/tcp_store.cpp).
Tensors live in GPU memory; collectives are enqueued on a HIP stream and never block the
host; every host-side wait has a deadline.  There is no host-memory transport: a run on
the CPU device is single-process.

``transport="loopback"`` swaps RCCL for the native test transport (csrc/comm/
loopback_comm.cpp) that lets several ranks share ONE GPU - RCCL itself refuses that - so the
multi-rank behaviour of the engine can be tested on a single-GPU machine.
"""
from __future__ import annotations

import datetime
import os
from typing import Optional, Sequence

import torch

_REDUCE_OPS = ("sum", "mean")


def split_host_port(master_ip: str):
    host, _, port = master_ip.rpartition(":")
    if not host or not port.isdigit():
        raise ValueError(f"--master-ip must be host:port, got {master_ip!r}")
    return host, int(port)


class Communicator:
    """Interface; ``rank``/``world_size``/``backend`` are plain attributes."""

    rank: int = 0
    world_size: int = 1
    backend: str = "none"

    def all_reduce(self, tensor: torch.Tensor, op: str = "sum") -> None:
        raise NotImplementedError

    def all_reduce_async(self, tensor: torch.Tensor, op: str = "sum"):
        """Start an all-reduce; returns a handle with ``wait()``."""
        raise NotImplementedError

    def gather(self, tensor: torch.Tensor, slabs: Optional[Sequence[torch.Tensor]],
               root: int = 0) -> None:
        """Rank ``root`` receives rank r's ``tensor`` into ``slabs[r]``."""
        raise NotImplementedError

    def scatter(self, out: torch.Tensor, chunks: Optional[Sequence[torch.Tensor]],
                root: int = 0) -> None:
        """Rank r receives ``chunks[r]`` of rank ``root`` into ``out``."""
        raise NotImplementedError

    def broadcast(self, tensor: torch.Tensor, root: int = 0) -> None:
        raise NotImplementedError

    def barrier(self) -> None:
        raise NotImplementedError

    def group_start(self) -> None:      # lets a transport fuse a run of calls into one launch
        pass

    def guard_sync(self, timeout_s: float = 300.0) -> None:
        """Wait for enqueued work with a deadline (turns a lost peer into an error)."""
        pass

    def group_end(self) -> None:
        pass

    def destroy(self) -> None:
        pass


class _Done:
    def wait(self) -> None:
        return None


def exchange_bootstrap_bytes(master_ip: str, rank: int, world_size: int, timeout_s: float,
                             make, key: str = "dml_rccl_unique_id"):
    """Rank 0 calls ``make()`` and publishes the bytes; everyone else fetches them.
    Returns ``(bytes, store)``; the caller keeps ``store`` (the native one, or None) alive.

    Stand-alone launches (one process per rank started by hand, by ``launch.py`` or by
    ``bench.py`` itself, like the reference) use the native TCP store listening on
    ``master_ip``.  Under ``torchrun`` that port already belongs to the launcher's
    own rendezvous store, which hands workers a client connection; the id is
    published there instead of opening a second listener on a guessed port.
    """
    if os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "").lower() == "true":
        from torch.distributed import TCPStore
        store = TCPStore(os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]),
                         world_size, is_master=False,
                         timeout=datetime.timedelta(seconds=timeout_s))
        key += "_" + os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")
        if rank == 0:
            blob = bytes(make())
            store.set(key, blob)
        else:
            blob = bytes(store.get(key))
        return blob, None
    from ..ops import native
    host, port = split_host_port(master_ip)
    store = native.lib().TcpStore(host, port, rank == 0, world_size, float(timeout_s))
    if rank == 0:
        blob = bytes(make())
        store.set(key, blob)
    else:
        blob = bytes(store.get(key))
    # nobody may tear the server down before every rank has read the id
    store.barrier("uid_read")
    return blob, store


class RcclCommunicator(Communicator):
    """Native communicator for device tensors (RCCL; ``transport="loopback"`` for tests).

    Every call enqueues on ``stream`` (default: torch's current stream) and returns
    immediately; ordering against compute is by stream order or by the events the
    bucket reducer records.  ``mean`` is implemented with ``ncclAvg``.
    """

    def __init__(self, master_ip: str, rank: int, world_size: int, device_index: int,
                 timeout_s: float = 300.0, transport: Optional[str] = None):
        from ..ops import native
        self._C = native.lib()
        self.rank = rank
        self.world_size = world_size
        self.device_index = device_index
        self.timeout_s = float(timeout_s)
        self.transport = transport or os.environ.get("DML_TRANSPORT", "rccl")
        self.backend = self.transport
        self._store = None
        torch.cuda.set_device(device_index)
        if self.transport == "rccl":
            uid = self._exchange_unique_id(master_ip, rank, world_size, timeout_s)
            self._comm = self._C.RcclComm(uid, rank, world_size, device_index)
        elif self.transport == "loopback":
            _, port = split_host_port(master_ip)
            # (the launcher's per-launch id keeps a crashed run's left-over segment on the same
            # port out of reach; ranks started by hand share the port-only name)
            name = str(port) + "_" + os.environ.get("DML_LAUNCH_ID", "")
            self._comm = self._C.LoopbackComm(name.rstrip("_"), rank, world_size, device_index,
                                              float(timeout_s))
        else:
            raise ValueError(f"unknown transport {self.transport!r} (rccl | loopback)")
        if self._comm.peer_count != world_size:
            raise RuntimeError(f"{self.transport} communicator reports {self._comm.peer_count} "
                               f"ranks, expected {world_size}")

    # -- bootstrap ---------------------------------------------------------------
    def _exchange_unique_id(self, master_ip, rank, world_size, timeout_s) -> bytes:
        uid, self._store = exchange_bootstrap_bytes(master_ip, rank, world_size, timeout_s,
                                                    self._C.rccl_unique_id)
        return uid

    # -- helpers -----------------------------------------------------------------
    @staticmethod
    def _stream(stream=None) -> int:
        s = stream if stream is not None else torch.cuda.current_stream()
        return int(s.cuda_stream)

    @staticmethod
    def _dtype_code(t: torch.Tensor) -> int:
        if t.dtype == torch.float32:
            return 0
        if t.dtype == torch.bfloat16:
            return 1
        if t.dtype == torch.int64:
            return 2
        if t.dtype == torch.uint8:
            return 3
        raise TypeError(f"unsupported dtype for a collective: {t.dtype}")

    @staticmethod
    def _check(t: torch.Tensor):
        if not (t.is_cuda and t.is_contiguous()):
            raise ValueError("collectives need contiguous device tensors")

    # -- collectives -------------------------------------------------------------
    def all_reduce(self, tensor, op="sum", stream=None):
        assert op in _REDUCE_OPS
        self._check(tensor)
        self._comm.all_reduce(tensor.data_ptr(), tensor.numel(), self._dtype_code(tensor),
                              op == "mean", self._stream(stream))

    def all_reduce_async(self, tensor, op="sum"):
        self.all_reduce(tensor, op)      # stream-ordered: nothing for the host to wait on
        return _Done()

    def gather(self, tensor, slabs, root=0, stream=None):
        self._check(tensor)
        ptrs = [s.data_ptr() for s in slabs] if self.rank == root else []
        self._comm.gather(tensor.data_ptr(), ptrs, tensor.numel(), self._dtype_code(tensor),
                          root, self._stream(stream))

    def scatter(self, out, chunks, root=0, stream=None):
        self._check(out)
        ptrs = [c.data_ptr() for c in chunks] if self.rank == root else []
        self._comm.scatter(out.data_ptr(), ptrs, out.numel(), self._dtype_code(out),
                           root, self._stream(stream))

    def broadcast(self, tensor, root=0, stream=None):
        self._check(tensor)
        self._comm.broadcast(tensor.data_ptr(), tensor.numel(), self._dtype_code(tensor),
                             root, self._stream(stream))

    def barrier(self, timeout_s: Optional[float] = None):
        """All ranks have reached this point and the current stream has drained.  Bounded:
        a lost peer raises (and aborts the communicator) instead of hanging every survivor."""
        token = torch.zeros(1, device=f"cuda:{self.device_index}")
        self.all_reduce(token)
        self.guard_sync(self.timeout_s if timeout_s is None else timeout_s)

    def guard_sync(self, timeout_s: float = 300.0) -> None:
        self._comm.wait_stream(self._stream(), float(timeout_s))

    def all_gather_scalars(self, value, dtype=torch.int64) -> torch.Tensor:
        """One scalar per rank, visible to every rank (a W-vector summed with one non-zero
        entry per rank); returns the host tensor.  For self-checks, not for the step."""
        v = torch.zeros(self.world_size, dtype=dtype, device=f"cuda:{self.device_index}")
        v[self.rank] = value
        self.all_reduce(v)
        self.guard_sync(self.timeout_s)
        return v.cpu()

    def volume(self) -> dict:
        """Bytes this rank has put on / taken off its links since the last ``reset_volume`` and
        the number of collectives of each kind, as issued (what the reference's report samples
        per node with an external tool, group25.pdf p4, p6, p8)."""
        return dict(self._comm.volume())

    def reset_volume(self) -> None:
        self._comm.reset_volume()

    def group_start(self):
        self._comm.group_start()

    def group_end(self):
        self._comm.group_end()

    @property
    def native(self):
        return self._comm

    @property
    def peer_count(self) -> int:
        return int(self._comm.peer_count)

    def destroy(self):
        if self._comm is not None:
            torch.cuda.synchronize()
            self._comm.destroy()
            self._comm = None
        self._store = None


def init_communicator(master_ip: str, rank: int, world_size: int, device: torch.device,
                      timeout_s: float = 300.0) -> Communicator:
    """Join the process group.  Gradient synchronisation runs on RCCL between GPUs; there is
    no host-memory transport, so a CPU run cannot take part in one."""
    if device.type != "cuda":
        raise RuntimeError(
            "gradient synchronisation (--sync gather_scatter | all_reduce | ddp) needs one "
            "MI355X per rank: the only transport is RCCL.  On the CPU device run --sync none.")
    index = device.index if device.index is not None else torch.cuda.current_device()
    return RcclCommunicator(master_ip, rank, world_size, index, timeout_s)
