"""torch.autograd bridge for the HIP backward (include/nerf_hip.h: nerf_hip_render_backward).

Forward = the fused render kernel in training mode (it also saves activations, LayerNorm
statistics and compositing state to a workspace); backward = three HIP launches that produce
the flat parameter-gradient vector (304,438 fp32, state_dict order).  Rays, fenceposts and
random draws receive no gradient (the reference never asks for one).
"""
import ctypes

import torch

from . import _lib


class RenderRaysFunction(torch.autograd.Function):
    """rgb [N,3], seg [N,50] = f(parameters); inputs after `rng_mode` are the 22 parameters."""

    @staticmethod
    def forward(ctx, model, rays_o, rays_d, num_samples, u, noise, density_noise_std, rng_mode,
                t_values, want_weights, *params):
        lib = _lib.lib()
        n_rays, device = rays_o.shape[0], rays_o.device
        ws_bytes = lib.nerf_hip_train_workspace_bytes(n_rays, num_samples)
        workspace = torch.empty(ws_bytes // 4, dtype=torch.float32, device=device)
        rng_state = model._next_philox_state() if rng_mode else None
        rng_counter = model._philox_device_counter(device) if rng_mode else None     # (graph replay: new draws)
        rgb, seg, _, _, weights = model._launch(n_rays, num_samples, device, rays_o=rays_o,
                                                rays_d=rays_d, u=u, noise=noise,
                                                density_noise_std=density_noise_std, rng_mode=rng_mode,
                                                rng_state=rng_state, rng_counter=rng_counter, t_values=t_values,
                                                want_weights=want_weights, train_workspace=workspace)
        if weights is None:
            weights = rgb.new_empty(0)
        if getattr(model, "keep_workspace", False):      # debugging / stage-parity tests only
            model.last_workspace = workspace
        ctx.set_materialize_grads(False)                 # an unused output arrives as None, not zeros
        ctx.model = model
        ctx.call = (rays_o, rays_d, num_samples, u, noise, density_noise_std, rng_mode, rng_state,
                    t_values)
        ctx.workspace = workspace
        ctx.precision = _lib.PRECISIONS[model.train_precision]   # the data gradient uses the same arithmetic
        ctx.packed = model._last_packed             # the image this forward used (its own buffer)
        ctx.shapes = [p.shape for p in params]
        ctx.save_for_backward(rgb, seg)
        ctx.mark_non_differentiable(weights)
        return rgb, seg, weights

    @staticmethod
    def backward(ctx, d_rgb, d_seg, _d_weights):
        lib = _lib.lib()
        model = ctx.model
        rgb, seg = ctx.saved_tensors
        rays_o, rays_d, num_samples, u, noise, std, rng_mode, rng_state, t_values = ctx.call
        n_rays, device = rays_o.shape[0], rays_o.device
        if d_rgb is None:
            d_rgb = torch.zeros_like(rgb)
        d_rgb = d_rgb.contiguous()
        # d_seg is None when the segmentation output did not reach the loss (no device sync to find
        # out): the kernels then skip the 50-class branch of the compositing backward
        d_seg = d_seg.contiguous() if d_seg is not None and model.segmentation_outputs > 0 else None

        args = _lib.BackwardArgs()
        model._fill_args(args.fwd, n_rays, num_samples, device, rays_o=rays_o, rays_d=rays_d, u=u,
                         t_values=t_values, noise=noise, density_noise_std=std, rng_mode=rng_mode, rng_state=rng_state,
                         packed=ctx.packed, rgb=rgb, seg=seg if model.segmentation_outputs > 0 else None,
                         train_workspace=ctx.workspace,
                         precision=ctx.precision)
        grad = torch.empty(lib.nerf_hip_grad_elements(model.hidden_size, model.enc_inputs, model.num_outputs),
                           dtype=torch.float32, device=device)
        scratch = model._scratch(lib.nerf_hip_backward_scratch_bytes(n_rays, num_samples), device)
        args.d_rgb, args.d_seg = _lib.ptr(d_rgb), _lib.ptr(d_seg)
        args.grad, args.scratch = _lib.ptr(grad), _lib.ptr(scratch)
        with torch.cuda.device(device):
            stream = torch.cuda.current_stream(device).cuda_stream
            _lib.check(lib.nerf_hip_render_backward(ctypes.byref(args), ctypes.c_void_p(stream)),
                       "nerf_hip_render_backward")
        ctx.workspace = None
        grads, off = [], 0
        for shape in ctx.shapes:                  # views of the flat vector, state_dict order
            n = 1
            for d in shape:
                n *= d
            grads.append(grad[off:off + n].view(shape))
            off += n
        model.last_flat_grad = grad
        return (None,) * 10 + tuple(grads)


class FieldFunction(torch.autograd.Function):
    """mean [N,S-1,3], raw [N,S-1,num_outputs] = f(parameters): the per-sample outputs of ``NeRF.forward``
    (nerf/model.py:553-594) with a backward — the reference's ``forward`` is an ordinary ``nn.Module.forward``, a loss
    on its density / colour / segmentation back-propagates into the parameters.  Forward = the training forward (the
    network outputs are already in its workspace; one small launch copies them out), backward =
    nerf_hip_render_backward in its ``d_raw`` form: the compositing backward is skipped, dL/d(raw) enters the
    data-gradient and weight-gradient kernels directly.  ``mean`` does not depend on the parameters."""

    @staticmethod
    def forward(ctx, model, rays_o, rays_d, samples, *params):
        lib = _lib.lib()
        n_rays, num_samples, device = samples.shape[0], samples.shape[-1], rays_o.device
        ws_bytes = lib.nerf_hip_train_workspace_bytes(n_rays, num_samples)
        workspace = torch.empty(ws_bytes // 4, dtype=torch.float32, device=device)
        # field only: nothing is composited (no rgb / seg / weights tensors, no compositing launch) — the backward
        # takes dL/d(raw) and needs none of it
        _, _, mean, raw, _ = model._launch(n_rays, num_samples, device, rays_o=rays_o, rays_d=rays_d,
                                           t_values=samples, per_sample=True, train_workspace=workspace,
                                           composite=False)
        if getattr(model, "keep_workspace", False):      # debugging / stage-parity tests only
            model.last_workspace = workspace
        ctx.model = model
        ctx.call = (rays_o, rays_d, samples)
        ctx.workspace = workspace
        ctx.precision = _lib.PRECISIONS[model.train_precision]
        ctx.packed = model._last_packed
        ctx.shapes = [p.shape for p in params]
        ctx.mark_non_differentiable(mean)
        return mean, raw

    @staticmethod
    def backward(ctx, _d_mean, d_raw):
        lib = _lib.lib()
        model = ctx.model
        rays_o, rays_d, samples = ctx.call
        n_rays, num_samples, device = samples.shape[0], samples.shape[-1], rays_o.device
        if d_raw is None:
            return (None,) * (4 + len(ctx.shapes))
        d_raw = d_raw.contiguous()
        args = _lib.BackwardArgs()
        model._fill_args(args.fwd, n_rays, num_samples, device, rays_o=rays_o, rays_d=rays_d, t_values=samples,
                         packed=ctx.packed, train_workspace=ctx.workspace, precision=ctx.precision)
        grad = torch.empty(lib.nerf_hip_grad_elements(model.hidden_size, model.enc_inputs, model.num_outputs),
                           dtype=torch.float32, device=device)
        scratch = model._scratch(lib.nerf_hip_backward_scratch_bytes(n_rays, num_samples), device)
        args.d_raw = _lib.ptr(d_raw)
        args.grad, args.scratch = _lib.ptr(grad), _lib.ptr(scratch)
        with torch.cuda.device(device):
            stream = torch.cuda.current_stream(device).cuda_stream
            _lib.check(lib.nerf_hip_render_backward(ctypes.byref(args), ctypes.c_void_p(stream)),
                       "nerf_hip_render_backward")
        ctx.workspace = None
        grads, off = [], 0
        for shape in ctx.shapes:
            n = 1
            for d in shape:
                n *= d
            grads.append(grad[off:off + n].view(shape))
            off += n
        model.last_flat_grad = grad
        return (None,) * 4 + tuple(grads)
