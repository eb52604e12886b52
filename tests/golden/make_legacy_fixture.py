"""Fixture G9: the reference's trained LEGACY checkpoint as data.

    python tests/golden/make_legacy_fixture.py        (build container only)

``/root/reference/examples/nerf.pth`` is a DATA file of the reference (44 weight tensors of a trained
Lego scene; its network's source is not in the repository).  The GPU box has no
/root/reference, so the tensors are stored as ``tests/golden/g9_legacy_checkpoint.npz`` (fp16 would
lose the parity bar: kept fp32, compressed, 2.3 MB), together with what ``oracle/legacy_oracle.py``
— the CPU statement of the recovered structure, PARITY UNPINNED — computes from them on seeded rays
(so that a change of the oracle itself is noticed) and a 40x40 oracle render of the scene."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import legacy_oracle as L, nerf_oracle as O      # noqa: E402


def main():
    sd = torch.load("/root/reference/examples/nerf.pth", map_location="cpu", weights_only=True)
    assert sorted(sd) == sorted(L.state_dict_keys())
    cfg = L.default_config()
    out = {"param." + k: v.numpy() for k, v in sd.items()}
    g = torch.Generator().manual_seed(9)
    n, S = 48, 40
    cam = torch.tensor([0.0, -3.5, 2.0])
    cam = cam / cam.norm() * 4.03
    cam_o, cam_r = cam[None], O.look_at_pose(cam.tolist())
    rays_o, rays_d = O.image_rays(cam_o, cam_r, 40, 40, 55.5)
    pick = torch.randperm(1600, generator=g)[:n]
    o, d = rays_o[pick], rays_d[pick]
    u = torch.rand(n, S, generator=g)
    noise = torch.randn(n, S, 1, generator=g)
    with torch.no_grad():
        rgb, st = L.render_rays(sd, cfg, o, d, 2.0, 6.0, S, return_stages=True)
        rgb_s = L.render_rays(sd, cfg, o, d, 2.0, 6.0, S, u=u, noise=noise, density_noise_std=0.5)
        img = torch.cat([L.render_rays(sd, cfg, a, b, 2.0, 6.0, 64)
                         for a, b in zip(rays_o.split(400), rays_d.split(400))]).reshape(40, 40, 3)
    out.update(camera_o=cam_o.numpy(), camera_r=cam_r.numpy(), rays_o=o.numpy(), rays_d=d.numpy(),
               u=u.numpy(), noise=noise.numpy(), rgb=rgb.numpy(), rgb_stochastic=rgb_s.numpy(),
               density=st["density"].numpy(), color=st["color"].numpy(), weights=st["weights"].numpy(),
               image40=img.numpy())
    path = os.path.join(HERE, "g9_legacy_checkpoint.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes; image mean", float(img.mean()))


if __name__ == "__main__":
    main()
