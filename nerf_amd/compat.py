"""Call-surface shims for the two OLDER API generations that the reference's own script and
notebook still use and that ``nerf/model.py`` as shipped no longer accepts:

  generation A (examples/example.ipynb):  NeRF(normalize_position=6.0);
      render_rays(o, d, near, far, S, randomly_sample, density_noise_std) -> [N,3]
      render_image(cam_o, cam_r, H, W, f, near, far, S)                  -> [B,H,W,3]
  generation B (train_conditional_nerf.py:86-150):  NeRF(normalize_position=, density_inputs=);
      render_rays(o, d, near, far, S, states_x=, ...) -> [N,stages,3]; NeRF.direction_to_rotation_matrix

PARITY UNPINNED.  The source of those generations is not in the repository, so only the
SIGNATURES are reproduced; the semantics below are stated choices, not recovered behaviour:
  * the network is the generation-C network (the only one whose source exists);
  * ``near``/``far`` select LINEAR fenceposts t in [near, far] (stratified when
    ``randomly_sample``), as in Mildenhall et al. 2020, instead of generation C's log-spaced table;
  * ``normalize_position`` sets the half-width of the scene box (rays_min/max = -/+ value);
    ``density_inputs`` and ``states_x`` are accepted and ignored (generation C ignores states too);
  * ``direction_to_rotation_matrix(d)``: camera looking along ``d`` with z up, through the
    reference's ``get_rotation_matrix`` convention.
"""
import torch

from .model import NeRF


class LegacyNeRF(NeRF):
    def __init__(self, normalize_position=20.0, density_inputs=3, stage_axis=True, **kwargs):
        box = float(normalize_position)
        super().__init__(min_x=-box, max_x=box, min_y=-box, max_y=box, min_z=-box, max_z=box, **kwargs)
        self.normalize_position = normalize_position
        self.density_inputs = density_inputs
        self.stage_axis = stage_axis              # True: generation B [N,stages,3]; False: A [N,3]

    @staticmethod
    def direction_to_rotation_matrix(direction):
        eye = direction / direction.norm(dim=-1, keepdim=True)
        z = torch.zeros_like(eye)
        z[..., 2] = 1.0
        up = z - (z * eye).sum(-1, keepdim=True) * eye
        up = up / up.norm(dim=-1, keepdim=True).clamp(min=1e-8)
        return torch.stack([torch.linalg.cross(eye, up, dim=-1), up, -eye], dim=-1)

    def linear_fenceposts(self, n_rays, near, far, num_samples, randomly_sample, device):
        t = torch.linspace(float(near), float(far), num_samples, dtype=torch.float32, device=device)
        t = t.expand(n_rays, num_samples)
        if randomly_sample:
            mid = 0.5 * (t[..., 1:] + t[..., :-1])
            lower = torch.cat([t[..., :1], mid], dim=-1)
            upper = torch.cat([mid, t[..., -1:]], dim=-1)
            t = lower + (upper - lower) * torch.rand(n_rays, num_samples, dtype=torch.float32, device=device)
        return t.contiguous()

    def render_rays(self, rays_o, rays_d, near, far, num_samples, states_x=None, states_d=None,
                    randomly_sample=False, density_noise_std=0.0):
        from .autograd import render_rays_function
        lead = rays_o.shape[:-1]
        flat_o = rays_o.detach().reshape(-1, 3).contiguous()
        flat_d = rays_d.detach().reshape(-1, 3).contiguous()
        n_rays, dev = flat_o.shape[0], flat_o.device
        t = self.linear_fenceposts(n_rays, near, far, num_samples, randomly_sample, dev)
        noise = torch.randn(n_rays, num_samples - 1, dtype=torch.float32, device=dev)
        noise = noise if density_noise_std != 0.0 else None
        rgb, _, _ = render_rays_function(self, flat_o, flat_d, num_samples, None, noise,
                                         float(density_noise_std), 0, t_values=t)
        rgb = rgb.reshape(*lead, self.color_outputs)
        return rgb.unsqueeze(-2) if self.stage_axis else rgb

    def render_image(self, camera_o, camera_r, image_h, image_w, focal_length, near, far, num_samples,
                     states_x=None, states_d=None, max_chunk_size=262144, randomly_sample=False,
                     density_noise_std=0.0):
        batch = camera_o.shape[0]
        rays = self.generate_rays(image_h, image_w, focal_length, dtype=camera_o.dtype,
                                  device=camera_o.device)
        rays = torch.broadcast_to(rays.unsqueeze(0), [batch, image_h, image_w, 3])
        cam_o = torch.broadcast_to(camera_o[:, None, None, :], [batch, image_h, image_w, 3])
        cam_r = torch.broadcast_to(camera_r[:, None, None, :, :], [batch, image_h, image_w, 3, 3])
        rays_o, rays_d = self.rays_to_world_coordinates(rays, cam_o, cam_r)
        rays_o, rays_d = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)
        out = []
        for o_i, d_i in zip(torch.split(rays_o, max_chunk_size), torch.split(rays_d, max_chunk_size)):
            px = self.render_rays(o_i, d_i, near, far, num_samples, randomly_sample=randomly_sample,
                                  density_noise_std=density_noise_std)
            out.append(px[:, -1] if self.stage_axis else px)
        return torch.cat(out).reshape(batch, image_h, image_w, self.color_outputs)
