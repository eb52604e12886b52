"""``mse`` — the loss of the reference's training loops and its gradient as ONE HIP launch
(include/nerf_hip.h: nerf_hip_mse_loss).

``((pixels - batch["pixels"].unsqueeze(1)) ** 2).mean()`` (train_conditional_nerf.py:132, examples/example.ipynb
cell 8) is four torch kernels forward and four more backward; at 512 rays per GPU (BASELINE config 5 on 8 GPUs)
each is a launch-bound 4-5 us of a 0.37 ms training step.  The fused form computes the loss and d loss / d pred
in one workgroup — the gradient with autograd's own rounding, (1 / count) * (2 x) — and the backward only scales it
by the incoming gradient.  An empty batch gives 0 (not NaN): the tail shard of a data-parallel epoch.
"""
import ctypes

import torch

from . import _lib


def _check(pred, target):
    if (pred.dim() != 3 or not 1 <= pred.shape[-1] <= 12 or target.shape != (pred.shape[0], pred.shape[-1])
            or not pred.is_cuda or pred.dtype != torch.float32 or target.dtype != torch.float32
            or target.device != pred.device):
        raise ValueError("nerf_amd.loss.mse: pred [N, stages, C] and target [N, C] (C = the network's color_outputs, "
                         "1 .. 12), float32, on one ROCm device")


def _launch(pred, target):
    """(loss [], d loss / d pred) of contiguous ``pred`` [N, stages, C] against ``target`` [N, C]: one launch."""
    loss = torch.empty(1, dtype=torch.float32, device=pred.device)
    grad = torch.empty_like(pred)
    args = _lib.MseArgs()
    args.pred, args.target = _lib.ptr(pred), _lib.ptr(target)
    args.n_rays, args.stages, args.channels = pred.shape[0], pred.shape[1], pred.shape[2]
    args.loss, args.grad = _lib.ptr(loss), _lib.ptr(grad)
    with torch.cuda.device(pred.device):
        stream = torch.cuda.current_stream(pred.device).cuda_stream
        _lib.check(_lib.lib().nerf_hip_mse_loss(ctypes.byref(args), ctypes.c_void_p(stream)), "nerf_hip_mse_loss")
    return loss.reshape(()), grad


class _MseFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        loss, grad = _launch(pred.contiguous(), target.contiguous())
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        (grad,) = ctx.saved_tensors
        return grad * grad_loss, None


def mse(pred, target):
    """mean((pred - target[:, None, :]) ** 2) for pred [N, stages, C] (or [N, C]) and target [N, C], float32 on a
    ROCm device; differentiable with respect to ``pred``."""
    if pred.dim() == 2:
        return mse(pred.unsqueeze(1), target)
    _check(pred, target)
    return _MseFunction.apply(pred, target)


def mse_and_grad(pred, target):
    """``(loss, d loss / d pred)`` without autograd: what a training loop needs to call ``pred.backward(grad)``
    itself — no unit root gradient to fill, no scaling of the saved gradient by it (two launches less than
    ``mse(pred, target).backward()``; ``nerf_amd.trainer.Trainer`` steps this way)."""
    shape = pred.shape
    if pred.dim() == 2:
        pred = pred.unsqueeze(1)
    _check(pred, target)
    with torch.no_grad():
        loss, grad = _launch(pred.detach().contiguous(), target.contiguous())
    return loss, grad.reshape(shape)
