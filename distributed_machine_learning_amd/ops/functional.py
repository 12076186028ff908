"""Autograd bindings of the native kernels: build-your-own-model API.

The training engine (``csrc/engine``) runs VGG as a fixed kernel schedule; this module is
the flexible counterpart: each op is a ``torch.autograd.Function`` whose forward and
backward both run the HIP kernels, so any chain of these layers trains under torch's
autograd with stock optimizers (or :class:`..optim.FusedSGD`).  Layout conventions:
activations bf16 NHWC ``[B, H, W, C]`` (the network input ``[B, 32, 32, 4]`` with a zero
fourth channel), parameters fp32 in the reference's shapes (conv weights OIHW), so a
module built from these ops has exactly the reference's ``state_dict``
(/root/reference/part1/model.py:11-46).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import native


def _krsc(weight_oihw: torch.Tensor) -> torch.Tensor:
    return weight_oihw.detach().permute(0, 2, 3, 1).contiguous()


class _Conv3x3(torch.autograd.Function):
    """conv3x3(stride 1, pad 1) + bias (+ ReLU)  <- nn.Conv2d(.., 3, 1, 1) [+ nn.ReLU]"""

    @staticmethod
    def forward(ctx, x, weight, bias, relu: bool):
        first = weight.shape[1] == 3
        w_krsc = _krsc(weight)
        if first:
            wf = native.pack_conv1_weights(w_krsc)
            y = native.conv1_fwd(x, wf, bias.detach(), relu)
        else:
            wf = w_krsc.to(torch.bfloat16)
            y = native.conv3x3_fwd(x, wf, bias.detach(), relu)
        ctx.first, ctx.relu = first, relu
        ctx.save_for_backward(x, wf, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wf, y = ctx.saved_tensors
        dy = dy.contiguous()
        if ctx.relu:
            dy = native.relu_mask(dy, y)
        if ctx.first:
            dw, db = native.conv1_wgrad(x, dy)
            return None, dw.permute(0, 3, 1, 2), db, None
        dw = native.conv3x3_wgrad(x, dy)
        db = native.column_sum(dy)
        dx = None
        if ctx.needs_input_grad[0]:
            # (the backward-data layout is derived here, once per step and only when an input
            # gradient is wanted; the engines keep it resident and refresh it after the update)
            dx = native.conv3x3_dgrad(dy, native.repack_dgrad(wf))
        return dx, dw.permute(0, 3, 1, 2), db, None


class _BatchNormReLU(torch.autograd.Function):
    """BatchNorm2d (training statistics, running-stat update) + ReLU in one pass."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training: bool, momentum: float,
                eps: float):
        if not training:
            return native.bn_relu_fwd_eval(x, gamma.detach(), beta.detach(), running_mean,
                                           running_var, eps)
        y, mean, rstd = native.bn_relu_fwd_train(x, gamma.detach(), beta.detach(), running_mean,
                                                 running_var, momentum, eps)
        ctx.save_for_backward(x, gamma.detach(), beta.detach(), mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        dx, dgamma, dbeta = native.bn_relu_bwd(x, dy.contiguous(), gamma, beta, mean, rstd)
        return dx, dgamma, dbeta, None, None, None, None, None


class _MaxPool2x2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return native.maxpool2x2_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return native.maxpool2x2_bwd(x, dy.contiguous())


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight.detach())
        return native.linear_fwd(x, weight.detach(), bias.detach())

    @staticmethod
    def backward(ctx, dlogits):
        x, w = ctx.saved_tensors
        dx, dw, db = native.linear_bwd(x, w, dlogits.contiguous())
        return dx, dw, db


class _SoftmaxCrossEntropy(torch.autograd.Function):
    """Mean cross-entropy; the gradient is produced by the same kernel as the loss."""

    @staticmethod
    def forward(ctx, logits, target):
        loss, dlogits, _ = native.softmax_xent(logits, target, want_grad=True)
        ctx.save_for_backward(dlogits)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dlogits,) = ctx.saved_tensors
        return dlogits * g, None


def conv3x3(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, relu: bool = False):
    """``weight`` OIHW fp32 (Cin 3 takes the 4-channel padded network input)."""
    return _Conv3x3.apply(x, weight, bias, relu)


def batch_norm_relu(x, gamma, beta, running_mean: Optional[torch.Tensor],
                    running_var: Optional[torch.Tensor], training: bool = True,
                    momentum: float = 0.1, eps: float = 1e-5):
    return _BatchNormReLU.apply(x, gamma, beta, running_mean, running_var, training, momentum, eps)


def max_pool2x2(x):
    return _MaxPool2x2.apply(x)


def linear(x, weight, bias):
    """bf16 ``[B, K]`` features x fp32 ``[N, K]`` weights -> fp32 logits."""
    return _Linear.apply(x, weight, bias)


def softmax_cross_entropy(logits, target):
    return _SoftmaxCrossEntropy.apply(logits, target)


def top1_correct(logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """Number of rows whose arg-max equals the target (int32 ``[1]``)."""
    _, _, correct = native.softmax_xent(logits, target, want_grad=False)
    return correct
