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


class _MseFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        n, stages = pred.shape[0], pred.shape[1]
        pred_c, target_c = pred.contiguous(), target.contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=pred.device)
        grad = torch.empty_like(pred_c)
        args = _lib.MseArgs()
        args.pred, args.target = _lib.ptr(pred_c), _lib.ptr(target_c)
        args.n_rays, args.stages = n, stages
        args.loss, args.grad = _lib.ptr(loss), _lib.ptr(grad)
        with torch.cuda.device(pred.device):
            stream = torch.cuda.current_stream(pred.device).cuda_stream
            _lib.check(_lib.lib().nerf_hip_mse_loss(ctypes.byref(args), ctypes.c_void_p(stream)), "nerf_hip_mse_loss")
        ctx.save_for_backward(grad)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad_loss):
        (grad,) = ctx.saved_tensors
        return grad * grad_loss, None


def mse(pred, target):
    """mean((pred - target[:, None, :]) ** 2) for pred [N, stages, 3] (or [N, 3]) and target [N, 3], float32 on a
    ROCm device; differentiable with respect to ``pred``."""
    if pred.dim() == 2:
        return mse(pred.unsqueeze(1), target)
    if (pred.dim() != 3 or pred.shape[-1] != 3 or target.shape != (pred.shape[0], 3) or not pred.is_cuda
            or pred.dtype != torch.float32 or target.dtype != torch.float32 or target.device != pred.device):
        raise ValueError("nerf_amd.loss.mse: pred [N, stages, 3] and target [N, 3], float32, on one ROCm device")
    return _MseFunction.apply(pred, target)
