"""Tensor-level wrappers over the native kernels (``_C``).

Each function checks dtype, shape, device and contiguity on the host, allocates the
output with torch, and enqueues the HIP kernel on torch's current stream.  Activations
are NHWC bf16 (``[B, H, W, C]``), conv weights ``[Cout, 3, 3, Cin]``, parameters and
statistics fp32.  The op-to-reference map is in ``csrc/include/dml_kernels.h``.

There is no eager fallback here: if the extension is missing the import error says how
to build it, so a GPU run can never silently use stock torch ops.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch

from ..config import CIFAR_MEAN, CIFAR_STD

_LIB = None

EPI_NONE, EPI_BIAS, EPI_BIAS_RELU, EPI_MASK = 0, 1, 2, 3


def lib():
    global _LIB
    if _LIB is None:
        try:
            from .. import _C  # type: ignore
        except ImportError as exc:  # pragma: no cover - build problem
            raise ImportError(
                "native extension distributed_machine_learning_amd/_C*.so is missing or does not "
                "load; build it with `python build.py` (hipcc, gfx950)") from exc
        if os.environ.get("DML_SKIP_BUILD_CHECK") != "1":
            from ..buildinfo import tree_hash
            want, have = tree_hash(), getattr(_C, "build_id", "")
            if want != have:
                raise ImportError(
                    f"native extension is stale: built from sources {have or 'unknown'}, the "
                    f"tree next to it hashes to {want}; run `python build.py`")
        _LIB = _C
    return _LIB


def available() -> bool:
    try:
        lib()
        return True
    except ImportError:
        return False


def _stream() -> int:
    return int(torch.cuda.current_stream().cuda_stream)


def _req(t: torch.Tensor, dtype, name: str, ndim: Optional[int] = None) -> torch.Tensor:
    if not t.is_cuda:
        raise ValueError(f"{name}: expected a GPU tensor")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    if ndim is not None and t.dim() != ndim:
        raise ValueError(f"{name}: expected {ndim} dims, got {tuple(t.shape)}")
    return t


def _ptr(t: Optional[torch.Tensor]) -> int:
    return 0 if t is None else t.data_ptr()


# --------------------------------------------------------------------------- conv
def conv3x3_fwd(x: torch.Tensor, wf: torch.Tensor, bias: Optional[torch.Tensor],
                relu: bool = True, workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``y = [relu](conv3x3(x, wf) + bias)``; x ``[B,H,W,Cin]``, wf ``[Cout,3,3,Cin]`` bf16."""
    _req(x, torch.bfloat16, "x", 4)
    _req(wf, torch.bfloat16, "wf", 4)
    b, h, w, cin = x.shape
    cout = wf.shape[0]
    if h != w or tuple(wf.shape[1:]) != (3, 3, cin):
        raise ValueError(f"conv3x3_fwd: x {tuple(x.shape)} vs wf {tuple(wf.shape)}")
    if bias is not None:
        _req(bias, torch.float32, "bias", 1)
        if bias.numel() != cout:
            raise ValueError("conv3x3_fwd: bias size")
    y = torch.empty(b, h, w, cout, dtype=torch.bfloat16, device=x.device)
    ws = _workspace(b, h, cin, cout, x.device, workspace)
    epi = EPI_NONE if bias is None else (EPI_BIAS_RELU if relu else EPI_BIAS)
    lib().conv3x3_igemm(x.data_ptr(), wf.data_ptr(), y.data_ptr(), _ptr(bias), 0, _ptr(ws),
                        0 if ws is None else ws.numel(), b, h, cin, cout, epi, _stream())
    return y


def _workspace(b, hw, cin, cout, device, given):
    need = lib().conv3x3_igemm_workspace_elems(b, hw, cin, cout)
    if need == 0:
        return None
    if given is not None:
        _req(given, torch.float32, "workspace")
        if given.numel() < need:
            raise ValueError("workspace too small")
        return given
    return torch.empty(need, dtype=torch.float32, device=device)


def repack_dgrad(wf: torch.Tensor) -> torch.Tensor:
    """``wd[ci][8-tap][co] = wf[co][tap][ci]`` - the weights in backward-data layout."""
    _req(wf, torch.bfloat16, "wf", 4)
    cout, _, _, cin = wf.shape
    wd = torch.empty(cin, 3, 3, cout, dtype=torch.bfloat16, device=wf.device)
    lib().repack_dgrad_weights(wf.data_ptr(), wd.data_ptr(), cout, cin, _stream())
    return wd


def conv3x3_dgrad(dy: torch.Tensor, wd: torch.Tensor, relu_mask: Optional[torch.Tensor] = None,
                  workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Input gradient; ``wd`` from :func:`repack_dgrad`.  With ``relu_mask`` (the producing
    layer's post-ReLU output) the ReLU backward is applied in the epilogue."""
    _req(dy, torch.bfloat16, "dy", 4)
    _req(wd, torch.bfloat16, "wd", 4)
    b, h, w, cout = dy.shape
    cin = wd.shape[0]
    if h != w or tuple(wd.shape[1:]) != (3, 3, cout):
        raise ValueError(f"conv3x3_dgrad: dy {tuple(dy.shape)} vs wd {tuple(wd.shape)}")
    dx = torch.empty(b, h, w, cin, dtype=torch.bfloat16, device=dy.device)
    if relu_mask is not None:
        _req(relu_mask, torch.bfloat16, "relu_mask", 4)
        if relu_mask.shape != dx.shape:
            raise ValueError("conv3x3_dgrad: mask shape")
    ws = _workspace(b, h, cout, cin, dy.device, workspace)
    lib().conv3x3_igemm(dy.data_ptr(), wd.data_ptr(), dx.data_ptr(), 0, _ptr(relu_mask), _ptr(ws),
                        0 if ws is None else ws.numel(), b, h, cout, cin,
                        EPI_MASK if relu_mask is not None else EPI_NONE, _stream())
    return dx


def conv3x3_wgrad(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    """``dw[co][tap][ci] = sum_p dy[p][co] x[p (+) tap][ci]`` as fp32 ``[Cout,3,3,Cin]``.

    A stand-alone call has the GPU to itself, where the tap-resident kernel is the faster one
    (and, summing per-split slabs, deterministic); shapes it does not cover take the per-tap
    kernel with atomic accumulation."""
    _req(x, torch.bfloat16, "x", 4)
    _req(dy, torch.bfloat16, "dy", 4)
    b, h, w, cin = x.shape
    cout = dy.shape[3]
    if tuple(dy.shape[:3]) != (b, h, w) or h != w:
        raise ValueError("conv3x3_wgrad: shape mismatch")
    if h in (2, 4, 8, 16, 32) and cin % 64 == 0 and cout % 64 == 0:
        return conv3x3_wgrad9(x, dy)
    dw = torch.zeros(cout, 3, 3, cin, dtype=torch.float32, device=x.device)
    lib().conv3x3_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), b, h, cin, cout, _stream())
    return dw


def conv3x3_wgrad_per_tap(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    """The one-tap-per-block kernel with atomic accumulation (what ``conv3x3_wgrad`` falls back
    to; the engines use it in its slab mode)."""
    _req(x, torch.bfloat16, "x", 4)
    _req(dy, torch.bfloat16, "dy", 4)
    b, h, w, cin = x.shape
    cout = dy.shape[3]
    if tuple(dy.shape[:3]) != (b, h, w) or h != w:
        raise ValueError("conv3x3_wgrad: shape mismatch")
    dw = torch.zeros(cout, 3, 3, cin, dtype=torch.float32, device=x.device)
    lib().conv3x3_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), b, h, cin, cout, _stream())
    return dw


def conv3x3_wgrad9(x: torch.Tensor, dy: torch.Tensor, bco: int = 0, splits: int = 0,
                   stages: int = 3, target: int = 0) -> torch.Tensor:
    """The tap-resident weight-gradient kernel on its own (``conv3x3_wgrad`` picks between it and
    the one-tap-per-block kernel).  ``splits`` = 0 takes the kernel's plan for ``target`` blocks;
    the split slabs are summed here with a tensor op (the engine uses ``sum_slabs_batch``)."""
    _req(x, torch.bfloat16, "x", 4)
    _req(dy, torch.bfloat16, "dy", 4)
    b, h, w, cin = x.shape
    cout = dy.shape[3]
    if tuple(dy.shape[:3]) != (b, h, w) or h != w:
        raise ValueError("conv3x3_wgrad9: shape mismatch")
    n = lib().conv3x3_wgrad9_splits(b, h, cin, cout, bco, splits, target)
    dw = torch.empty(cout, 3, 3, cin, dtype=torch.float32, device=x.device)
    slab = torch.empty(n, cout, 3, 3, cin, dtype=torch.float32, device=x.device) if n > 1 else None
    lib().conv3x3_wgrad9(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), b, h, cin, cout, _stream(),
                         _ptr(slab), bco, n, stages)
    return slab.sum(0) if n > 1 else dw


def pack_conv1_weights(w: torch.Tensor) -> torch.Tensor:
    """fp32 ``[64,3,3,3]`` (KRSC) -> bf16 ``[64,64]`` with k = tap*4 + c."""
    _req(w, torch.float32, "w", 4)
    if tuple(w.shape) != (64, 3, 3, 3):
        raise ValueError("pack_conv1_weights: expects [64,3,3,3]")
    out = torch.empty(64, 64, dtype=torch.bfloat16, device=w.device)
    lib().repack_conv1_weights(w.data_ptr(), out.data_ptr(), _stream())
    return out


def conv1_fwd(x4: torch.Tensor, w1: torch.Tensor, bias: torch.Tensor, relu: bool = True) -> torch.Tensor:
    _req(x4, torch.bfloat16, "x4", 4)
    _req(w1, torch.bfloat16, "w1", 2)
    _req(bias, torch.float32, "bias", 1)
    b = x4.shape[0]
    if tuple(x4.shape[1:]) != (32, 32, 4) or tuple(w1.shape) != (64, 64) or bias.numel() != 64:
        raise ValueError("conv1_fwd: expects x [B,32,32,4], w1 [64,64], bias [64]")
    y = torch.empty(b, 32, 32, 64, dtype=torch.bfloat16, device=x4.device)
    lib().conv1_fprop(x4.data_ptr(), w1.data_ptr(), bias.data_ptr(), y.data_ptr(), b, relu, _stream())
    return y


def conv1_wgrad(x4: torch.Tensor, dy: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    _req(x4, torch.bfloat16, "x4", 4)
    _req(dy, torch.bfloat16, "dy", 4)
    b = x4.shape[0]
    if tuple(x4.shape[1:]) != (32, 32, 4) or tuple(dy.shape) != (b, 32, 32, 64):
        raise ValueError("conv1_wgrad: shape mismatch")
    dw = torch.zeros(64, 3, 3, 3, dtype=torch.float32, device=x4.device)
    db = torch.zeros(64, dtype=torch.float32, device=x4.device)
    ws = torch.empty(lib().conv1_wgrad_workspace_elems(b), dtype=torch.float32, device=x4.device)
    lib().conv1_wgrad(x4.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), b,
                      ws.data_ptr(), _stream())
    return dw, db


def conv1_wgrad_relu_pool(x4, act, dy_pooled):
    """Weight / bias gradient of a first layer followed by ReLU + 2x2 max-pool, from the stored
    ReLU output and the gradient of the pooled output.  Returns ``(dw, db)``."""
    _req(x4, torch.bfloat16, "x4", 4)
    _req(act, torch.bfloat16, "act", 4)
    _req(dy_pooled, torch.bfloat16, "dy_pooled", 4)
    b = x4.shape[0]
    if tuple(x4.shape[1:]) != (32, 32, 4) or tuple(act.shape) != (b, 32, 32, 64) or \
            tuple(dy_pooled.shape) != (b, 16, 16, 64):
        raise ValueError("conv1_wgrad_relu_pool: shape mismatch")
    dw = torch.zeros(64, 3, 3, 3, dtype=torch.float32, device=x4.device)
    db = torch.zeros(64, dtype=torch.float32, device=x4.device)
    ws = torch.empty(lib().conv1_wgrad_workspace_elems(b), dtype=torch.float32, device=x4.device)
    lib().conv1_wgrad_relu_pool(x4.data_ptr(), act.data_ptr(), dy_pooled.data_ptr(), dw.data_ptr(),
                                db.data_ptr(), b, ws.data_ptr(), _stream())
    return dw, db


def conv1_wgrad_bn_pool(x4, pre, dy_pooled, gamma, beta, save_mean, save_rstd):
    """Weight gradient of a first layer followed by BatchNorm + ReLU + 2x2 max-pool, from the
    gradient of the pooled output; the full-resolution gradient is never materialised.
    Returns ``(dw, dgamma, dbeta)``."""
    _req(x4, torch.bfloat16, "x4", 4)
    _req(pre, torch.bfloat16, "pre", 4)
    _req(dy_pooled, torch.bfloat16, "dy_pooled", 4)
    b = x4.shape[0]
    if tuple(x4.shape[1:]) != (32, 32, 4) or tuple(pre.shape) != (b, 32, 32, 64) or \
            tuple(dy_pooled.shape) != (b, 16, 16, 64):
        raise ValueError("conv1_wgrad_bn_pool: shape mismatch")
    dev = x4.device
    dgamma = torch.empty(64, dtype=torch.float32, device=dev)
    dbeta = torch.empty(64, dtype=torch.float32, device=dev)
    # reductions only (dx == nullptr): dgamma / dbeta
    lib().bn_relu_pool_bwd(pre.data_ptr(), dy_pooled.data_ptr(), 0, gamma.data_ptr(),
                           beta.data_ptr(), save_mean.data_ptr(), save_rstd.data_ptr(),
                           dgamma.data_ptr(), dbeta.data_ptr(), _reduce_ws(dev).data_ptr(),
                           b, 32, 64, _stream())
    dw = torch.zeros(64, 3, 3, 3, dtype=torch.float32, device=dev)
    ws = torch.empty(lib().conv1_wgrad_workspace_elems(b), dtype=torch.float32, device=dev)
    lib().conv1_wgrad_bn_pool(x4.data_ptr(), pre.data_ptr(), dy_pooled.data_ptr(), gamma.data_ptr(),
                              beta.data_ptr(), save_mean.data_ptr(), save_rstd.data_ptr(),
                              dgamma.data_ptr(), dbeta.data_ptr(), dw.data_ptr(), b, ws.data_ptr(),
                              _stream())
    return dw, dgamma, dbeta


def column_sum(x: torch.Tensor) -> torch.Tensor:
    """Sum over all leading dims of a bf16 tensor whose last dim is the channel."""
    _req(x, torch.bfloat16, "x")
    c = x.shape[-1]
    out = torch.zeros(c, dtype=torch.float32, device=x.device)
    lib().column_sum(x.data_ptr(), out.data_ptr(), x.numel() // c, c, _reduce_ws(x.device).data_ptr(),
                     _stream())
    return out


_REDUCE_WS = {}


def _reduce_ws(device) -> torch.Tensor:
    """Partial-sum slab of the two-kernel column reductions.  One per device; launches that
    share it must be ordered on one stream."""
    key = (device.type, device.index)
    if key not in _REDUCE_WS:
        _REDUCE_WS[key] = torch.zeros(lib().column_reduce_workspace_elems(), dtype=torch.float32,
                                      device=device)
    return _REDUCE_WS[key]


def relu_mask(x: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """``mask > 0 ? x : 0`` (ReLU backward from the ReLU's output), bf16 or fp32."""
    if x.dtype not in (torch.bfloat16, torch.float32):
        raise TypeError("relu_mask: bf16 or fp32")
    _req(x, x.dtype, "x")
    _req(mask, x.dtype, "mask")
    if mask.shape != x.shape or x.numel() % 8:
        raise ValueError("relu_mask: shapes must match, numel % 8 == 0")
    y = torch.empty_like(x)
    fn = lib().relu_mask if x.dtype == torch.bfloat16 else lib().relu_mask_f32
    fn(x.data_ptr(), mask.data_ptr(), y.data_ptr(), x.numel(), _stream())
    return y


# --------------------------------------------------------------------------- pool
def maxpool2x2_fwd(x: torch.Tensor) -> torch.Tensor:
    _req(x, torch.bfloat16, "x", 4)
    b, h, w, c = x.shape
    if h != w:
        raise ValueError("maxpool2x2: square inputs only")
    y = torch.empty(b, h // 2, w // 2, c, dtype=torch.bfloat16, device=x.device)
    lib().maxpool2x2_fwd(x.data_ptr(), y.data_ptr(), b, h, c, _stream())
    return y


def maxpool2x2_bwd(x: torch.Tensor, dy: torch.Tensor, relu_mask: bool = False) -> torch.Tensor:
    _req(x, torch.bfloat16, "x", 4)
    _req(dy, torch.bfloat16, "dy", 4)
    b, h, w, c = x.shape
    if tuple(dy.shape) != (b, h // 2, w // 2, c):
        raise ValueError("maxpool2x2_bwd: shape mismatch")
    dx = torch.empty_like(x)
    lib().maxpool2x2_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), b, h, c, relu_mask, _stream())
    return dx


# --------------------------------------------------------------------- batch norm
def bn_relu_fwd_train(x, gamma, beta, running_mean=None, running_var=None, momentum=0.1,
                      eps=1e-5):
    """Returns ``(y, save_mean, save_rstd)``; running stats are updated in place."""
    _req(x, torch.bfloat16, "x")
    c = x.shape[-1]
    rows = x.numel() // c
    for t, n in ((gamma, "gamma"), (beta, "beta")):
        _req(t, torch.float32, n, 1)
    y = torch.empty_like(x)
    mean = torch.empty(c, dtype=torch.float32, device=x.device)
    rstd = torch.empty(c, dtype=torch.float32, device=x.device)
    ws = _reduce_ws(x.device)
    lib().bn_relu_fwd_train(x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                            _ptr(running_mean), _ptr(running_var), mean.data_ptr(),
                            rstd.data_ptr(), ws.data_ptr(), rows, c, momentum, eps, _stream())
    return y, mean, rstd


def bn_relu_fwd_eval(x, gamma, beta, running_mean, running_var, eps=1e-5):
    _req(x, torch.bfloat16, "x")
    c = x.shape[-1]
    y = torch.empty_like(x)
    lib().bn_relu_fwd_eval(x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                           running_mean.data_ptr(), running_var.data_ptr(), x.numel() // c, c, eps,
                           _stream())
    return y


def bn_relu_bwd(x, dy, gamma, beta, save_mean, save_rstd):
    """Returns ``(dx, dgamma, dbeta)``."""
    _req(x, torch.bfloat16, "x")
    _req(dy, torch.bfloat16, "dy")
    c = x.shape[-1]
    dx = torch.empty_like(x)
    dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
    ws = _reduce_ws(x.device)
    lib().bn_relu_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                      save_mean.data_ptr(), save_rstd.data_ptr(), dgamma.data_ptr(),
                      dbeta.data_ptr(), ws.data_ptr(), x.numel() // c, c, _stream())
    return dx, dgamma, dbeta


def bn_relu_pool_fwd_train(x, gamma, beta, running_mean=None, running_var=None, momentum=0.1,
                           eps=1e-5):
    """BatchNorm + ReLU + 2x2 max-pool; returns ``(pooled, save_mean, save_rstd)``."""
    _req(x, torch.bfloat16, "x", 4)
    b, h, w, c = x.shape
    if h != w or h % 2:
        raise ValueError("bn_relu_pool: square even inputs only")
    y = torch.empty(b, h // 2, w // 2, c, dtype=torch.bfloat16, device=x.device)
    mean = torch.empty(c, dtype=torch.float32, device=x.device)
    rstd = torch.empty(c, dtype=torch.float32, device=x.device)
    lib().bn_relu_pool_fwd_train(x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                 _ptr(running_mean), _ptr(running_var), mean.data_ptr(),
                                 rstd.data_ptr(), _reduce_ws(x.device).data_ptr(), b, h, c,
                                 momentum, eps, _stream())
    return y, mean, rstd


def bn_relu_pool_fwd_eval(x, gamma, beta, running_mean, running_var, eps=1e-5):
    _req(x, torch.bfloat16, "x", 4)
    b, h, w, c = x.shape
    y = torch.empty(b, h // 2, w // 2, c, dtype=torch.bfloat16, device=x.device)
    lib().bn_relu_pool_fwd_eval(x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                running_mean.data_ptr(), running_var.data_ptr(), b, h, c, eps,
                                _stream())
    return y


def bn_relu_pool_bwd(x, dy, gamma, beta, save_mean, save_rstd):
    """``x`` the conv output, ``dy`` the gradient of the pooled tensor; returns
    ``(dx, dgamma, dbeta)``."""
    _req(x, torch.bfloat16, "x", 4)
    _req(dy, torch.bfloat16, "dy", 4)
    b, h, w, c = x.shape
    if tuple(dy.shape) != (b, h // 2, w // 2, c):
        raise ValueError("bn_relu_pool_bwd: shape mismatch")
    dx = torch.empty_like(x)
    dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
    lib().bn_relu_pool_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), gamma.data_ptr(),
                           beta.data_ptr(), save_mean.data_ptr(), save_rstd.data_ptr(),
                           dgamma.data_ptr(), dbeta.data_ptr(), _reduce_ws(x.device).data_ptr(),
                           b, h, c, _stream())
    return dx, dgamma, dbeta


# ------------------------------------------------------------- classifier and loss
def linear_fwd(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    _req(x, torch.bfloat16, "x", 2)
    _req(w, torch.float32, "w", 2)
    _req(bias, torch.float32, "bias", 1)
    b, k = x.shape
    n = w.shape[0]
    if w.shape[1] != k or bias.numel() != n:
        raise ValueError("linear_fwd: shape mismatch")
    out = torch.empty(b, n, dtype=torch.float32, device=x.device)
    lib().linear_fwd(x.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(), b, k, n, _stream())
    return out


def linear_bwd(x, w, dlogits):
    """Returns ``(dx bf16, dw fp32, db fp32)``."""
    _req(x, torch.bfloat16, "x", 2)
    _req(w, torch.float32, "w", 2)
    _req(dlogits, torch.float32, "dlogits", 2)
    b, k = x.shape
    n = w.shape[0]
    dx = torch.empty(b, k, dtype=torch.bfloat16, device=x.device)
    dw = torch.zeros(n, k, dtype=torch.float32, device=x.device)
    db = torch.zeros(n, dtype=torch.float32, device=x.device)
    lib().linear_bwd(x.data_ptr(), w.data_ptr(), dlogits.data_ptr(), dw.data_ptr(), db.data_ptr(),
                     dx.data_ptr(), b, k, n, _stream())
    return dx, dw, db


def softmax_xent(logits: torch.Tensor, target: torch.Tensor, want_grad: bool = True,
                 grad_scale: float = 1.0):
    """Mean cross-entropy.  Returns ``(loss[1], dlogits or None, correct[1] int32)``."""
    _req(logits, torch.float32, "logits", 2)
    _req(target, torch.int64, "target", 1)
    b, c = logits.shape
    loss = torch.zeros(1, dtype=torch.float32, device=logits.device)
    correct = torch.zeros(1, dtype=torch.int32, device=logits.device)
    dl = torch.empty_like(logits) if want_grad else None
    lib().softmax_xent(logits.data_ptr(), target.data_ptr(), loss.data_ptr(), _ptr(dl),
                       correct.data_ptr(), b, c, grad_scale, _stream())
    return loss, dl, correct


def fc_train(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, target: torch.Tensor):
    """Fused classifier training step.  Returns ``(logits, loss[1], correct[1], dx, dw, db)``."""
    _req(x, torch.bfloat16, "x", 2)
    _req(w, torch.float32, "w", 2)
    _req(bias, torch.float32, "bias", 1)
    _req(target, torch.int64, "target", 1)
    b, k = x.shape
    n = w.shape[0]
    dev = x.device
    logits = torch.empty(b, n, dtype=torch.float32, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    correct = torch.empty(1, dtype=torch.int32, device=dev)
    dx = torch.empty(b, k, dtype=torch.bfloat16, device=dev)
    dw = torch.empty(n, k, dtype=torch.float32, device=dev)
    db = torch.empty(n, dtype=torch.float32, device=dev)
    ws = torch.empty(lib().fc_train_workspace_elems(b, k, n), dtype=torch.float32, device=dev)
    lib().fc_train(x.data_ptr(), w.data_ptr(), bias.data_ptr(), target.data_ptr(), logits.data_ptr(),
                   loss.data_ptr(), correct.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(),
                   ws.data_ptr(), b, k, n, _stream())
    return logits, loss, correct, dx, dw, db


# ---------------------------------------------------------------------- optimiser
def sgd_momentum_(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, lr: float, momentum: float,
                  weight_decay: float, grad_scale: float = 1.0,
                  shadow: Optional[torch.Tensor] = None, first_step: bool = False) -> None:
    """In-place fused SGD over flat fp32 buffers (numel a multiple of 4)."""
    for t, n in ((p, "p"), (g, "g"), (m, "m")):
        _req(t, torch.float32, n, 1)
    if not (p.numel() == g.numel() == m.numel()):
        raise ValueError("sgd_momentum_: size mismatch")
    if shadow is not None:
        _req(shadow, torch.bfloat16, "shadow", 1)
        if shadow.numel() < p.numel():
            raise ValueError("sgd_momentum_: shadow too small")
    lib().sgd_momentum(p.data_ptr(), g.data_ptr(), m.data_ptr(), _ptr(shadow), p.numel(), lr,
                       momentum, weight_decay, grad_scale, first_step, _stream())


# --------------------------------------------------------------- data and helpers
def augment_normalize(images_u8: torch.Tensor, aug: Optional[torch.Tensor],
                      dtype: torch.dtype = torch.bfloat16) -> torch.Tensor:
    """uint8 ``[B,32,32,3]`` -> bf16 or fp32 ``[B,32,32,4]`` (crop/flip per ``aug`` row,
    normalised)."""
    _req(images_u8, torch.uint8, "images", 4)
    b = images_u8.shape[0]
    if aug is not None:
        _req(aug, torch.int32, "aug", 2)
    out = torch.empty(b, 32, 32, 4, dtype=dtype, device=images_u8.device)
    fn = lib().augment_normalize if dtype == torch.bfloat16 else lib().augment_normalize_f32
    if dtype not in (torch.bfloat16, torch.float32):
        raise TypeError("augment_normalize: bf16 or fp32 output")
    fn(images_u8.data_ptr(), _ptr(aug), out.data_ptr(), b, list(CIFAR_MEAN), list(CIFAR_STD),
       _stream())
    return out


def nchw_to_nhwc4(x: torch.Tensor) -> torch.Tensor:
    _req(x, torch.float32, "x", 4)
    b = x.shape[0]
    if tuple(x.shape[1:]) != (3, 32, 32):
        raise ValueError("nchw_to_nhwc4: expects [B,3,32,32]")
    out = torch.empty(b, 32, 32, 4, dtype=torch.bfloat16, device=x.device)
    lib().nchw_to_nhwc4(x.data_ptr(), out.data_ptr(), b, _stream())
    return out


def to_nhwc_bf16(x: torch.Tensor) -> torch.Tensor:
    _req(x, torch.float32, "x", 4)
    b, c, h, w = x.shape
    out = torch.empty(b, h, w, c, dtype=torch.bfloat16, device=x.device)
    lib().nchw_f32_to_nhwc_bf16(x.data_ptr(), out.data_ptr(), b, c, h, _stream())
    return out


def to_nchw_f32(x: torch.Tensor) -> torch.Tensor:
    _req(x, torch.bfloat16, "x", 4)
    b, h, w, c = x.shape
    out = torch.empty(b, c, h, w, dtype=torch.float32, device=x.device)
    lib().nhwc_bf16_to_nchw_f32(x.data_ptr(), out.data_ptr(), b, c, h, _stream())
    return out


def sum_slabs_(slabs: torch.Tensor) -> None:
    """``slabs[0] += slabs[1] + ... + slabs[W-1]`` for an fp32 ``[W, n]`` tensor."""
    _req(slabs, torch.float32, "slabs", 2)
    lib().sum_slabs(slabs.data_ptr(), slabs.shape[0], slabs.shape[1], _stream())


def copy_batch_(pairs) -> None:
    """``dst.copy_(src)`` for every ``(dst, src)`` pair of contiguous device tensors with equal
    byte sizes, in one kernel launch (per 24 pairs)."""
    items = []
    for dst, src in pairs:
        if not (dst.is_cuda and src.is_cuda and dst.is_contiguous() and src.is_contiguous()):
            raise ValueError("copy_batch_: contiguous device tensors only")
        nbytes = dst.numel() * dst.element_size()
        if nbytes != src.numel() * src.element_size():
            raise ValueError("copy_batch_: byte sizes differ")
        if nbytes:
            items.append((dst.data_ptr(), src.data_ptr(), nbytes))
    if items:
        lib().copy_batch(items, _stream())


def fill_uniform_(t: torch.Tensor, seed: int, scale: float = 1.0) -> torch.Tensor:
    _req(t, torch.bfloat16, "t")
    lib().fill_uniform_bf16(t.data_ptr(), t.numel(), seed, scale, _stream())
    return t


# ------------------------------------------------------------------ fp32 activations
# The reference's precision: NHWC fp32 tensors, conv weights ``[Cout,3,3,Cin]`` fp32.
F32 = torch.float32


def conv3x3_fwd_f32(x, wf, bias=None, relu=True, tile=0, pm=-1, split=-1):
    _req(x, F32, "x", 4)
    _req(wf, F32, "wf", 4)
    b, h, w, cin = x.shape
    cout = wf.shape[0]
    if h != w or tuple(wf.shape[1:]) != (3, 3, cin):
        raise ValueError(f"conv3x3_fwd_f32: x {tuple(x.shape)} vs wf {tuple(wf.shape)}")
    if bias is not None:
        _req(bias, F32, "bias", 1)
    y = torch.empty(b, h, w, cout, dtype=F32, device=x.device)
    epi = EPI_NONE if bias is None else (EPI_BIAS_RELU if relu else EPI_BIAS)
    lib().conv3x3_igemm_f32(x.data_ptr(), wf.data_ptr(), y.data_ptr(), _ptr(bias), 0, b, h, cin, cout,
                            epi, _stream(), tile, pm, split)
    return y


def repack_dgrad_f32(wf):
    _req(wf, F32, "wf", 4)
    cout, _, _, cin = wf.shape
    wd = torch.empty(cin, 3, 3, cout, dtype=F32, device=wf.device)
    lib().repack_dgrad_weights_f32(wf.data_ptr(), wd.data_ptr(), cout, cin, _stream())
    return wd


def conv3x3_dgrad_f32(dy, wd, relu_mask=None, tile=0, pm=-1, split=-1):
    _req(dy, F32, "dy", 4)
    _req(wd, F32, "wd", 4)
    b, h, w, cout = dy.shape
    cin = wd.shape[0]
    if h != w or tuple(wd.shape[1:]) != (3, 3, cout):
        raise ValueError("conv3x3_dgrad_f32: shape mismatch")
    dx = torch.empty(b, h, w, cin, dtype=F32, device=dy.device)
    if relu_mask is not None:
        _req(relu_mask, F32, "relu_mask", 4)
        if relu_mask.shape != dx.shape:
            raise ValueError("conv3x3_dgrad_f32: mask shape")
    lib().conv3x3_igemm_f32(dy.data_ptr(), wd.data_ptr(), dx.data_ptr(), 0, _ptr(relu_mask), b, h, cout,
                            cin, EPI_MASK if relu_mask is not None else EPI_NONE, _stream(), tile, pm,
                            split)
    return dx


def conv3x3_wgrad_f32(x, dy, splits=0, pm=-1):
    _req(x, F32, "x", 4)
    _req(dy, F32, "dy", 4)
    b, h, w, cin = x.shape
    cout = dy.shape[3]
    if tuple(dy.shape[:3]) != (b, h, w) or h != w:
        raise ValueError("conv3x3_wgrad_f32: shape mismatch")
    dw = torch.empty(cout, 3, 3, cin, dtype=F32, device=x.device)
    n = lib().conv3x3_wgrad_f32_splits(b, h, cin, cout, splits, pm)
    slab = torch.empty(n, dw.numel(), dtype=F32, device=x.device)
    lib().conv3x3_wgrad_f32(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), b, h, cin, cout,
                            slab.data_ptr(), _stream(), splits, pm)
    if n > 1:       # the engine sums the splits with sum_slabs_batch; here a tensor op does
        dw = slab.sum(dim=0).view_as(dw)
    return dw


def conv1_fwd_f32(x4, w, bias, relu=True):
    _req(x4, F32, "x4", 4)
    _req(w, F32, "w", 4)
    _req(bias, F32, "bias", 1)
    b = x4.shape[0]
    if tuple(x4.shape[1:]) != (32, 32, 4) or tuple(w.shape) != (64, 3, 3, 3) or bias.numel() != 64:
        raise ValueError("conv1_fwd_f32: expects x [B,32,32,4], w [64,3,3,3], bias [64]")
    y = torch.empty(b, 32, 32, 64, dtype=F32, device=x4.device)
    lib().conv1_fprop_f32(x4.data_ptr(), w.data_ptr(), bias.data_ptr(), y.data_ptr(), b, relu, _stream())
    return y


def conv1_wgrad_f32(x4, dy):
    _req(x4, F32, "x4", 4)
    _req(dy, F32, "dy", 4)
    b = x4.shape[0]
    if tuple(x4.shape[1:]) != (32, 32, 4) or tuple(dy.shape) != (b, 32, 32, 64):
        raise ValueError("conv1_wgrad_f32: shape mismatch")
    dw = torch.empty(64, 3, 3, 3, dtype=F32, device=x4.device)
    db = torch.empty(64, dtype=F32, device=x4.device)
    ws = torch.empty(lib().conv1_wgrad_f32_workspace_elems(b), dtype=F32, device=x4.device)
    lib().conv1_wgrad_f32(x4.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), b,
                          ws.data_ptr(), _stream())
    return dw, db


def maxpool2x2_fwd_f32(x):
    _req(x, F32, "x", 4)
    b, h, w, c = x.shape
    y = torch.empty(b, h // 2, w // 2, c, dtype=F32, device=x.device)
    lib().maxpool2x2_fwd_f32(x.data_ptr(), y.data_ptr(), b, h, c, _stream())
    return y


def maxpool2x2_bwd_f32(x, dy, relu_mask=False):
    _req(x, F32, "x", 4)
    _req(dy, F32, "dy", 4)
    b, h, w, c = x.shape
    dx = torch.empty_like(x)
    lib().maxpool2x2_bwd_f32(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), b, h, c, relu_mask, _stream())
    return dx


def bn_relu_fwd_train_f32(x, gamma, beta, running_mean=None, running_var=None, momentum=0.1,
                          eps=1e-5):
    _req(x, F32, "x")
    c = x.shape[-1]
    y = torch.empty_like(x)
    mean = torch.empty(c, dtype=F32, device=x.device)
    rstd = torch.empty(c, dtype=F32, device=x.device)
    lib().bn_relu_fwd_train_f32(x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                _ptr(running_mean), _ptr(running_var), mean.data_ptr(),
                                rstd.data_ptr(), _reduce_ws(x.device).data_ptr(), x.numel() // c, c,
                                momentum, eps, _stream())
    return y, mean, rstd


def bn_relu_fwd_eval_f32(x, gamma, beta, running_mean, running_var, eps=1e-5):
    _req(x, F32, "x")
    c = x.shape[-1]
    y = torch.empty_like(x)
    lib().bn_relu_fwd_eval_f32(x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                               running_mean.data_ptr(), running_var.data_ptr(), x.numel() // c, c,
                               eps, _stream())
    return y


def bn_relu_bwd_f32(x, dy, gamma, beta, save_mean, save_rstd):
    _req(x, F32, "x")
    _req(dy, F32, "dy")
    c = x.shape[-1]
    dx = torch.empty_like(x)
    dgamma = torch.empty(c, dtype=F32, device=x.device)
    dbeta = torch.empty(c, dtype=F32, device=x.device)
    lib().bn_relu_bwd_f32(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), gamma.data_ptr(),
                          beta.data_ptr(), save_mean.data_ptr(), save_rstd.data_ptr(),
                          dgamma.data_ptr(), dbeta.data_ptr(), _reduce_ws(x.device).data_ptr(),
                          x.numel() // c, c, _stream())
    return dx, dgamma, dbeta


def bn_relu_pool_fwd_train_f32(x, gamma, beta, running_mean=None, running_var=None, momentum=0.1,
                               eps=1e-5):
    """BatchNorm + ReLU + 2x2 max-pool of an fp32 NHWC conv output in one pass: returns the pooled
    activation and the batch statistics backward needs."""
    _req(x, F32, "x", 4)
    b, h, w, c = x.shape
    if h != w or h % 2:
        raise ValueError("bn_relu_pool: square even inputs only")
    y = torch.empty(b, h // 2, w // 2, c, dtype=F32, device=x.device)
    mean = torch.empty(c, dtype=F32, device=x.device)
    rstd = torch.empty(c, dtype=F32, device=x.device)
    lib().bn_relu_pool_fwd_train_f32(x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                     _ptr(running_mean), _ptr(running_var), mean.data_ptr(),
                                     rstd.data_ptr(), _reduce_ws(x.device).data_ptr(), b, h, c,
                                     momentum, eps, _stream())
    return y, mean, rstd


def bn_relu_pool_fwd_eval_f32(x, gamma, beta, running_mean, running_var, eps=1e-5):
    _req(x, F32, "x", 4)
    b, h, w, c = x.shape
    if h != w or h % 2:
        raise ValueError("bn_relu_pool: square even inputs only")
    y = torch.empty(b, h // 2, w // 2, c, dtype=F32, device=x.device)
    lib().bn_relu_pool_fwd_eval_f32(x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                    running_mean.data_ptr(), running_var.data_ptr(), b, h, c, eps,
                                    _stream())
    return y


def bn_relu_pool_bwd_f32(x, dy, gamma, beta, save_mean, save_rstd):
    """Backward of bn_relu_pool_fwd_train_f32: x the conv output, dy the pooled gradient; returns
    (dx at full resolution, dgamma, dbeta)."""
    _req(x, F32, "x", 4)
    _req(dy, F32, "dy", 4)
    b, h, w, c = x.shape
    if tuple(dy.shape) != (b, h // 2, w // 2, c) or h != w or h % 2:
        raise ValueError("bn_relu_pool_bwd_f32: shape mismatch")
    dx = torch.empty_like(x)
    dgamma = torch.empty(c, dtype=F32, device=x.device)
    dbeta = torch.empty(c, dtype=F32, device=x.device)
    lib().bn_relu_pool_bwd_f32(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), gamma.data_ptr(),
                               beta.data_ptr(), save_mean.data_ptr(), save_rstd.data_ptr(),
                               dgamma.data_ptr(), dbeta.data_ptr(), _reduce_ws(x.device).data_ptr(),
                               b, h, c, _stream())
    return dx, dgamma, dbeta


def column_sum_f32(x):
    _req(x, F32, "x")
    c = x.shape[-1]
    out = torch.empty(c, dtype=F32, device=x.device)
    lib().column_sum_f32(x.data_ptr(), out.data_ptr(), x.numel() // c, c,
                         _reduce_ws(x.device).data_ptr(), _stream())
    return out


def fc_train_f32(x, w, bias, target):
    """Fused classifier step on fp32 features: ``(logits, loss[1], correct[1], dx, dw, db)``."""
    _req(x, F32, "x", 2)
    _req(w, F32, "w", 2)
    b, k = x.shape
    n = w.shape[0]
    dev = x.device
    logits = torch.empty(b, n, dtype=F32, device=dev)
    loss = torch.empty(1, dtype=F32, device=dev)
    correct = torch.empty(1, dtype=torch.int32, device=dev)
    dx = torch.empty(b, k, dtype=F32, device=dev)
    dw = torch.empty(n, k, dtype=F32, device=dev)
    db = torch.empty(n, dtype=F32, device=dev)
    ws = torch.empty(lib().fc_train_workspace_elems(b, k, n), dtype=F32, device=dev)
    lib().fc_train_f32(x.data_ptr(), w.data_ptr(), bias.data_ptr(), target.data_ptr(),
                       logits.data_ptr(), loss.data_ptr(), correct.data_ptr(), dx.data_ptr(),
                       dw.data_ptr(), db.data_ptr(), ws.data_ptr(), b, k, n, _stream())
    return logits, loss, correct, dx, dw, db


def linear_fwd_f32(x, w, bias):
    _req(x, F32, "x", 2)
    b, k = x.shape
    n = w.shape[0]
    out = torch.empty(b, n, dtype=F32, device=x.device)
    lib().linear_fwd_f32(x.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(), b, k, n, _stream())
    return out


def nchw_to_nhwc4_f32(x):
    _req(x, F32, "x", 4)
    b = x.shape[0]
    out = torch.empty(b, 32, 32, 4, dtype=F32, device=x.device)
    lib().nchw_to_nhwc4_f32(x.data_ptr(), out.data_ptr(), b, _stream())
    return out
