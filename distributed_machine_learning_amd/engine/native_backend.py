"""Native execution backend: the C++ ``VggEngine`` plus the native sync strategies.

One training step is ONE call into C++, ``engine.train_step``: forward, loss, backward (with
the bucket reducer's hooks between layers in the DDP mode), the manual gradient
synchronisation of the gather/scatter and all-reduce modes (``ManualSync``), then the fused
SGD and the weight repacks - all enqueued on torch's current stream; the reference's
equivalent sequence is /root/reference/part2/2b/main.py:52-62.

Flat buffers are torch tensors so parameters can be viewed, broadcast, checkpointed and
compared against the stock-torch model; their internal layout (conv weights
``[cout,3,3,cin]``) is converted to the OIHW ``state_dict`` contract at the boundary.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch

from ..config import SYNC_ALL_REDUCE, SYNC_DDP, SYNC_GATHER_SCATTER, SYNC_NONE, TrainConfig
from ..models.vgg import VGG, VGG_WIDTHS
from ..ops import native
from ..parallel.comm import Communicator
from ..parallel.flat import cap_tail_bucket


class NativeBackend:
    """``cfg.dtype`` selects the engine: ``bf16`` (default) = bf16 activations and MFMA operands
    with fp32 master weights, accumulators and statistics (``VggEngine``); ``fp32`` = the
    reference's precision, every tensor fp32, fp32-input MFMA (``VggEngineF32``)."""

    name = "native"

    def __init__(self, cfg: TrainConfig, device: torch.device, comm: Optional[Communicator],
                 init_state: Optional[Dict[str, torch.Tensor]] = None):
        if cfg.dtype not in ("bf16", "fp32", "auto"):
            raise ValueError(f"native backend: dtype must be bf16 or fp32, got {cfg.dtype!r}")
        self.fp32 = cfg.dtype == "fp32"
        self.act_dtype = torch.float32 if self.fp32 else torch.bfloat16
        self.loader_layout = "nhwc4_fp32" if self.fp32 else "nhwc4_bf16"
        if self.fp32 and cfg.use_graph:
            raise ValueError("--graph replays the bf16 engine's step; the fp32 engine has no "
                             "graph mode")
        self.cfg = cfg
        self.device = device
        self.comm = comm
        self.world = comm.world_size if comm is not None else 1
        C = native.lib()
        widths = [-1 if w == "M" else int(w) for w in VGG_WIDTHS[cfg.arch]]
        engine_cls = C.VggEngineF32 if self.fp32 else C.VggEngine
        # kernel tuning switches (they size the engine's slabs, so they come first)
        if os.environ.get("DML_WGRAD_VARIANT"):
            C.set_wgrad_variant(int(os.environ["DML_WGRAD_VARIANT"]))
        if os.environ.get("DML_WGRAD9"):        # "<co tile>,<ring stages>"
            bco, stages = os.environ["DML_WGRAD9"].split(",")
            C.set_wgrad9_config(int(bco), int(stages))
        if os.environ.get("DML_WGRAD_TARGET"):
            C.set_wgrad_block_target(int(os.environ["DML_WGRAD_TARGET"]))
        if os.environ.get("DML_WGRAD_KSTEP"):
            C.set_wgrad_kstep(int(os.environ["DML_WGRAD_KSTEP"]))
        if os.environ.get("DML_WGRAD_WAVES"):
            C.set_wgrad_waves(int(os.environ["DML_WGRAD_WAVES"]))
        self.engine = engine_cls(widths, cfg.batch_size, cfg.batch_norm, cfg.num_classes)
        e = self.engine
        f32 = dict(dtype=torch.float32, device=device)
        self._params = torch.zeros(e.param_elems, **f32)
        self.grads = torch.zeros(e.param_elems, **f32)
        self._momentum = torch.zeros(e.param_elems, **f32)
        self.bn_buffers = torch.zeros(max(e.bn_elems, 1), **f32)
        self.arena = torch.zeros(e.arena_elems, dtype=self.act_dtype, device=device)
        self.workspace = torch.zeros(e.workspace_elems, **f32)
        e.bind(self._params.data_ptr(), self.grads.data_ptr(), self._momentum.data_ptr(),
               self.bn_buffers.data_ptr() if cfg.batch_norm else 0, self.arena.data_ptr(),
               self.workspace.data_ptr())
        self.slots = list(e.param_slots)
        self.buffer_slots = list(e.buffer_slots)
        self._loss_off = e.loss_offset
        self._loss_view = self.workspace[e.loss_offset:e.loss_offset + 1]
        self._training = True
        # initial weights: the stock module's initialisers under the caller's seed, so every
        # backend (and every rank) starts from the same numbers the reference would
        if init_state is None:
            init_state = VGG(cfg.arch, cfg.batch_norm, cfg.num_classes).state_dict()
        self.load_state_dict(init_state)
        self.reducer = None
        self.manual = None
        self._grad_scale = 1.0
        self._comm_on = True
        if self.world > 1:
            self._verify_replicas()
        self._build_sync()
        # after the reducer exists: the side stream must share a hardware queue with neither
        # the compute stream nor the communication stream
        # layers per hand-off (each costs ~15 us of host time and a cross-queue hop): one layer
        # per hand-off measured best with and without a reducer (0.712 vs 0.732 ms with one)
        group = int(cfg.wgrad_group) or 1
        if self.fp32:
            # the fp32 step is bound by the convolutions' arithmetic: no optimizer fusion, and the
            # weight gradients go to a side stream only to fill the tails of the other kernels
            self._fuse_opt = False
            e.set_wgrad_overlap(bool(cfg.wgrad_overlap) and not cfg.serialize_streams,
                                self._stream())
        else:
            e.set_wgrad_overlap(bool(cfg.wgrad_overlap) and not cfg.serialize_streams, group,
                                self._stream())
            # the optimizer update can run inside backward when the gradients backward leaves
            # behind are final: no sync, or the bucket reducer (the two manual modes
            # synchronise after backward has ended)
            self._fuse_opt = bool(cfg.fused_optimizer) and e.wgrad_overlap and self.manual is None
            # deep layers' update on the side stream, next to the first layers of the next forward
            late = int(os.environ.get("DML_UPDATE_OVERLAP", cfg.update_overlap))
            if late < 0:        # auto: from the first layer of the last spatial sizes (4x4 and below)
                late = next((i for i, row in enumerate(e.layer_table) if row[2] <= 4), 0)
            if e.wgrad_overlap and not self._fuse_opt and not cfg.use_graph:
                e.set_update_overlap(min(late, e.num_conv))
        self._graph_ready = False
        self._x_static = None
        self._t_static = None

    # ------------------------------------------------------------------ sync
    def _grad_views(self) -> List[torch.Tensor]:
        return [self.grads[s.offset:s.offset + s.numel] for s in self.slots]

    def _verify_replicas(self) -> None:
        """Every rank must hold the same model before gradients are exchanged: a mismatch
        would pair up unrelated gradient elements (or hang in a collective of different
        length).  DistributedDataParallel's constructor makes the same check
        (/root/reference/part3/main.py:137); here it covers all three sync modes."""
        e, cfg = self.engine, self.cfg
        mine = [len(self.slots), int(e.param_elems), int(e.bn_elems), int(cfg.batch_norm),
                sum((i + 1) * int(s.numel) for i, s in enumerate(self.slots)) % (1 << 40)]
        t = torch.tensor(mine, dtype=torch.int64, device=self.device)
        ref = t.clone()
        self.comm.broadcast(ref, 0)
        bad = torch.zeros(1, dtype=torch.int64, device=self.device)
        bad[0] = int(not torch.equal(ref, t))
        self.comm.all_reduce(bad)
        self.comm.guard_sync(self.cfg.comm_timeout_s)
        if int(bad.item()):
            raise RuntimeError(
                f"[rank {self.comm.rank}] model layout differs between ranks: mine "
                f"(slots, parameter elements, buffer elements, batch_norm, shape checksum) = "
                f"{mine}, rank 0 has {ref.tolist()}")

    def _tensor_slices(self):
        return [(int(s.offset), int(s.numel)) for s in self.slots]

    def _build_sync(self) -> None:
        cfg, comm = self.cfg, self.comm
        reduce = cfg.resolved_grad_reduce()
        C = native.lib()
        self.sync_name = cfg.sync if comm is not None else SYNC_NONE
        if cfg.sync == SYNC_NONE or comm is None:
            return
        # collectives sum; the 1/world of "mean" is folded into the fused SGD kernel's gradient
        # scale, where it is free
        self._grad_scale = 1.0 / self.world if reduce == "mean" else 1.0
        if cfg.sync in (SYNC_GATHER_SCATTER, SYNC_ALL_REDUCE):
            tensors = self._tensor_slices()
            slabs, stride = 0, 0
            if cfg.sync == SYNC_GATHER_SCATTER and comm.rank == 0:
                # root-side landing area, allocated once (the reference allocates fresh zero
                # tensors every iteration, part2/2a/main.py:96-98,112)
                stride = int(C.ManualSync.packed_elems(tensors))
                self._slab_store = torch.zeros(self.world, stride, dtype=torch.float32,
                                               device=self.device)
                slabs = self._slab_store.data_ptr()
            mode = 0 if cfg.sync == SYNC_GATHER_SCATTER else 1
            self.manual = C.ManualSync(comm.native, mode, self.grads.data_ptr(), tensors, 0,
                                       slabs, stride)
            self.engine.set_manual_sync(self.manual)
            return
        if cfg.sync == SYNC_DDP:
            sizes = [s.numel * 4 for s in self.slots]
            buckets = C.plan_buckets(sizes, int(cfg.bucket_cap_mb * (1 << 20)),
                                     int(cfg.first_bucket_cap_mb * (1 << 20)))
            buckets = cap_tail_bucket(buckets, sizes, int(cfg.tail_bucket_cap_mb * (1 << 20)))
            e = self.engine
            stage = cfg.batch_norm and cfg.broadcast_buffers and e.bn_elems > 0
            if stage:
                # room for the BatchNorm buffers behind the last gradient (same storage offsets
                # for everything the engine addresses)
                grown = torch.zeros(e.param_elems + e.bn_elems, dtype=torch.float32,
                                    device=self.device)
                self.grads = grown
                e.bind(self._params.data_ptr(), self.grads.data_ptr(), self._momentum.data_ptr(),
                       self.bn_buffers.data_ptr(), self.arena.data_ptr(), self.workspace.data_ptr())
            slices = []
            for b, (lo, hi) in enumerate(buckets):
                start = self.slots[lo].offset
                end = self.slots[hi - 1].offset + self.slots[hi - 1].numel
                if stage and b == 0:        # the first bucket ends with the last parameter
                    end = e.param_elems + e.bn_elems
                elif hi == len(self.slots):
                    # up to the padded end of the flat buffer (zeros): whole groups of four
                    end = e.param_elems
                slices.append((start, end - start))
            self.buckets = buckets
            self._bucket_slices = slices
            self.reducer = C.BucketReducer(comm.native, self.grads.data_ptr(), slices, False,
                                           cfg.serialize_streams, self._stream())
            # The bucket that completes with the end of backward can run on the compute stream
            # itself (two cross-queue hand-offs fewer).  With more than one rank it goes through
            # the communication stream like the others unless asked: one stream per communicator
            # is the pattern every RCCL user exercises.
            inline = cfg.inline_last_bucket
            self.reducer.inline_last = (self.world == 1) if inline is None else bool(inline)
            if cfg.grad_comm_dtype == "bf16":
                self._wire = torch.zeros(self.grads.numel(), dtype=torch.bfloat16,
                                         device=self.device)
                self.reducer.set_bf16_payload(self._wire.data_ptr())
            elif cfg.grad_comm_dtype != "fp32":
                raise ValueError(f"grad_comm_dtype must be fp32 or bf16, got {cfg.grad_comm_dtype}")
            self.engine.set_reducer(self.reducer, buckets)
            if stage:
                # DDP's per-forward buffer broadcast: rank 0's running statistics ride in the
                # staging floats the first bucket's slice ends with (see BucketReducer)
                self.reducer.set_buffer_sync(self.bn_buffers.data_ptr(),
                                             self.grads[e.param_elems:].data_ptr(), e.bn_elems)
            if cfg.grad_comm_dtype == "bf16" and not self.fp32:
                # the fused update reads the reduced bf16 values where the collective left them
                # (no widening pass; ``grads`` keeps this rank's own gradients)
                self.reducer.set_keep_on_wire(True)
            # what DistributedDataParallel's constructor does: everyone adopts rank 0's state
            if self.world > 1:
                comm.broadcast(self._params, 0)
                if cfg.batch_norm:
                    comm.broadcast(self.bn_buffers, 0)
                self.engine.refresh_weights(self._stream())
            return
        raise ValueError(cfg.sync)

    # ------------------------------------------- cross-stream / cross-GPU hand-off forms
    def handoff_forms(self) -> Dict[str, str]:
        """How the step's hand-offs are expressed right now: ``engine`` = main stream -> weight-
        gradient stream inside this GPU, ``buckets`` = gradient buckets -> communication stream
        (whose collective lets peers read this GPU's memory).  "value": flag word + wait-value
        packet, "event": event pair."""
        out = {"engine": str(self.engine.hand_off_mode)}
        if self.reducer is not None:
            out["buckets"] = str(self.reducer.departure_mode)
        return out

    def set_handoff_forms(self, by_value: bool) -> None:
        """Switch every hand-off to flag words (True) or event pairs (False).  The schedule and the
        arithmetic are the same either way; only call it between steps, device idle."""
        torch.cuda.synchronize(self.device)
        if by_value:
            try:
                self.engine.set_value_hand_offs(True)
            except Exception:       # noqa: BLE001 - a device without wait-value support keeps events
                pass
            if self.reducer is not None and self.reducer.can_depart_by_value:
                self.reducer.set_departure_by_value(True)
        else:
            self.engine.set_value_hand_offs(False)
            if self.reducer is not None:
                self.reducer.set_departure_by_value(False)

    def snapshot(self):
        """Every piece of training state a step changes (see ``restore``)."""
        return (self.params.clone(), self.momentum.clone(), self.bn_buffers.clone(),
                int(self.engine.bn_steps))

    @torch.no_grad()
    def restore(self, snap) -> None:
        torch.cuda.current_stream().synchronize()
        self.params.copy_(snap[0])
        self.momentum.copy_(snap[1])
        self.bn_buffers.copy_(snap[2])
        self.engine.bn_steps = snap[3]
        if getattr(self, "reducer", None) is not None:
            self.reducer.drop_pending_buffers()
        self.engine.refresh_weights(self._stream())
        torch.cuda.current_stream().synchronize()

    def sync_plan(self):
        """What the step synchronises, for logs and benchmark records."""
        plan = {"mode": self.sync_name, "grad_scale": self._grad_scale,
                "engine_hand_off": str(self.engine.hand_off_mode)}
        if self.reducer is not None:
            plan["buckets_mb"] = [round(c * 4 / (1 << 20), 3) for _, c in self._bucket_slices]
            plan["inline_last_bucket"] = bool(self.reducer.inline_last)
            plan["bucket_departure"] = self.reducer.departure_mode
            plan["wire_dtype"] = self.cfg.grad_comm_dtype
        if self.manual is not None:
            plan["collectives_per_step"] = len(self.slots) * (2 if self.sync_name ==
                                                               SYNC_GATHER_SCATTER else 1)
        return plan

    # ------------------------------------------- communication / compute accounting
    def set_comm_enabled(self, on: bool) -> None:
        """Detach (or re-attach) every gradient collective while keeping the rest of the step
        identical - the compute-only baseline of the overlap report.  Ranks drift apart while
        detached; only use it for timing."""
        self._comm_on = on
        if self.reducer is not None:
            self.engine.set_reducer(self.reducer if on else None, self.buckets if on else [])
        if self.manual is not None:
            self.engine.set_manual_sync(self.manual if on else None)

    def comm_only_step(self) -> None:
        """Issue exactly the gradient collectives of one step, nothing else."""
        if self.reducer is not None:
            for start, count in self._bucket_slices:
                self.comm.all_reduce(self.grads[start:start + count], "sum")
        elif self.manual is not None:
            self.manual.run(self._stream())

    # ------------------------------------------------------------------ steps
    def _join_update(self) -> None:
        """Order the current stream behind a deep-layer update still running on the side
        stream (``VggEngine.set_update_overlap``) before the flat buffers are touched."""
        if hasattr(self.engine, "join_update"):
            self.engine.join_update(self._stream())

    @property
    def params(self) -> torch.Tensor:
        self._join_update()
        return self._params

    @property
    def momentum(self) -> torch.Tensor:
        self._join_update()
        return self._momentum

    @staticmethod
    def _stream() -> int:
        return int(torch.cuda.current_stream().cuda_stream)

    def train_mode(self, on: bool = True) -> None:
        self._training = on

    def _check_batch(self, data: torch.Tensor, target: torch.Tensor) -> int:
        if data.dtype != self.act_dtype or data.dim() != 4 or tuple(data.shape[1:]) != (32, 32, 4):
            raise ValueError(f"native backend expects {self.act_dtype} [B,32,32,4] input, got "
                             f"{data.dtype} {tuple(data.shape)}")
        n = int(data.shape[0])
        if n > self.cfg.batch_size or target.shape[0] != n or target.dtype != torch.int64:
            raise ValueError("batch larger than the engine was built for, or bad target")
        if not (data.is_contiguous() and target.is_contiguous()):
            raise ValueError("inputs must be contiguous")
        return n

    def train_step(self, data: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        cfg, e = self.cfg, self.engine
        n = self._check_batch(data, target)
        s = self._stream()
        if not self.fp32:
            e.set_fused_optimizer(self._fuse_opt, cfg.lr, cfg.momentum, cfg.weight_decay,
                                  self._grad_scale)
        if cfg.use_graph and self.world == 1:
            return self._train_step_graph(data, target, n, s)
        e.train_step(data.data_ptr(), target.data_ptr(), n, cfg.lr, cfg.momentum, cfg.weight_decay,
                     self._grad_scale, s)
        return self._loss_view

    def _train_step_graph(self, data, target, n, s):
        cfg, e = self.cfg, self.engine
        if n != cfg.batch_size:
            raise ValueError("graph mode needs full batches")
        if self._x_static is None:
            # first call runs eagerly (one-time kernel attribute set-up happens outside any
            # capture); inputs are staged in fixed buffers the captured graph will read
            self._x_static = torch.empty_like(data)
            self._t_static = torch.empty_like(target)
            self._graph_stream = torch.cuda.Stream(device=self.device)
            self._x_static.copy_(data)
            self._t_static.copy_(target)
            e.forward_backward(self._x_static.data_ptr(), self._t_static.data_ptr(), n, s)
            e.optimizer_step(cfg.lr, cfg.momentum, cfg.weight_decay, self._grad_scale, s)
            return self.workspace[self._loss_off:self._loss_off + 1]
        # capture needs a non-default stream; order it after the caller's stream and back
        gs = self._graph_stream
        gs.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(gs):
            self._x_static.copy_(data)
            self._t_static.copy_(target)
            if not self._graph_ready:
                e.capture_train_step(self._x_static.data_ptr(), self._t_static.data_ptr(), n,
                                     cfg.lr, cfg.momentum, cfg.weight_decay, 1.0,
                                     int(gs.cuda_stream))
                self._graph_ready = True
            e.replay_train_step(int(gs.cuda_stream))
        torch.cuda.current_stream().wait_stream(gs)
        return self.workspace[self._loss_off:self._loss_off + 1]

    @torch.no_grad()
    def eval_batch(self, data: torch.Tensor, target: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        n = self._check_batch(data, target)
        lo = self._loss_off
        self.workspace[lo + 1:lo + 3].zero_()
        self.engine.eval_step(data.data_ptr(), target.data_ptr(), n, self._stream())
        loss = self.workspace[lo + 1].clone()
        correct = self.workspace[lo + 2:lo + 3].view(torch.int32)[0].clone()
        return loss, correct

    def logits(self, n: int) -> torch.Tensor:
        off = self.engine.logits_offset
        c = self.cfg.num_classes
        return self.workspace[off:off + n * c].view(n, c)

    # ------------------------------------------------------------ state dict
    def _param_view(self, slot) -> torch.Tensor:
        return self.params[slot.offset:slot.offset + slot.numel].view(*slot.shape)

    def state_dict(self) -> Dict[str, torch.Tensor]:
        """Reference layout: conv weights OIHW fp32, BatchNorm buffers included."""
        out: Dict[str, torch.Tensor] = {}
        bufs = {s.name: s for s in self.buffer_slots}
        for slot in self.slots:
            v = self._param_view(slot)
            if len(slot.shape) == 4:
                v = v.permute(0, 3, 1, 2)
            out[slot.name] = v.detach().clone().contiguous()
            if self.cfg.batch_norm and slot.name.endswith(".bias") and \
                    slot.name.replace(".bias", ".running_mean") in bufs:
                base = slot.name[:-len(".bias")]
                for leaf in ("running_mean", "running_var"):
                    b = bufs[f"{base}.{leaf}"]
                    out[f"{base}.{leaf}"] = self.bn_buffers[b.offset:b.offset + b.numel].clone()
                out[f"{base}.num_batches_tracked"] = torch.tensor(
                    int(self.engine.bn_steps), dtype=torch.int64, device=self.device)
        return out

    @torch.no_grad()
    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        from ..models.checkpoint import strip_ddp_prefix, validate_state_dict
        sd = strip_ddp_prefix(sd)
        validate_state_dict(sd, self.cfg.arch, self.cfg.batch_norm)
        for slot in self.slots:
            src = sd[slot.name].to(device=self.device, dtype=torch.float32)
            if len(slot.shape) == 4:
                src = src.permute(0, 2, 3, 1)
            self._param_view(slot).copy_(src)
        steps = 0
        for b in self.buffer_slots:
            self.bn_buffers[b.offset:b.offset + b.numel].copy_(
                sd[b.name].to(device=self.device, dtype=torch.float32))
            steps = int(sd[b.name.rsplit(".", 1)[0] + ".num_batches_tracked"])
        self.engine.bn_steps = steps
        if getattr(self, "reducer", None) is not None:
            self.reducer.drop_pending_buffers()
        self.momentum.zero_()
        self.engine.refresh_weights(self._stream())

    @torch.no_grad()
    def prime(self, batches, steps: int) -> None:
        """Run ``steps`` training steps and put every piece of training state back afterwards
        (master weights, momentum, running statistics and their step count): code objects are
        loaded, RCCL channels and caches are warm and the GPU has reached its steady clocks
        before anything is measured, and the trajectory is the one of an unprimed run.  With
        more than one rank every rank must call it with the same ``steps``."""
        if steps <= 0:
            return
        keep = self.snapshot()
        for i in range(steps):
            self.train_step(*batches[i % len(batches)])
        self.restore(keep)

    def parameters_flat(self) -> torch.Tensor:
        return self.params

    def close(self) -> None:
        torch.cuda.synchronize()
        self.engine.set_reducer(None, [])
        self.engine.set_manual_sync(None)
        self.reducer = None
        self.manual = None
