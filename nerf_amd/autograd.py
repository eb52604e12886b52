"""Differentiable wrapper of the fused renderer (training path).

``render_rays_function`` is what ``NeRF.render_rays`` calls: without gradients it is a plain
kernel launch; with gradients it goes through ``torch.autograd.Function`` whose backward runs
the HIP backward kernels.  Rays, fenceposts and draws are not differentiated (the reference never
needs it).
"""
import torch


def _needs_grad(model):
    return torch.is_grad_enabled() and any(p.requires_grad for p in model.parameters())


def render_rays_function(model, rays_o, rays_d, num_samples, u, noise, density_noise_std, rng_mode,
                         t_values=None, want_weights=False):
    """-> (rgb [N,3], seg [N,50], weights [N,S-1] or None)."""
    if not _needs_grad(model):
        rgb, seg, _, _, weights = model._launch(
            rays_o.shape[0], num_samples, rays_o.device, rays_o=rays_o, rays_d=rays_d, u=u, noise=noise,
            density_noise_std=density_noise_std, rng_mode=rng_mode, t_values=t_values,
            want_weights=want_weights)
        return rgb, seg, weights
    from .backward import RenderRaysFunction
    rgb, seg, weights = RenderRaysFunction.apply(model, rays_o, rays_d, num_samples, u, noise,
                                                 density_noise_std, rng_mode, t_values, want_weights,
                                                 *model._param_list())
    return rgb, seg, (weights if want_weights else None)
