"""Generate the golden fixtures under tests/golden/ by running the REFERENCE itself.

Run in the build container only (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports ``nerf.model`` / ``nerf.dataset`` from /root/reference (read-only, not copied),
evaluates generation C of the renderer on seeded inputs and stores inputs + outputs as
small ``.npz`` files.  Fixture ids follow SURVEY.md section 8c (G1..G8; G10: ImageRayDataset; G11: narrow networks).  Everything is
fp32; default init under ``torch.manual_seed(0)``; weights are stored once in
``params_seed0.npz`` and the "x3" variants multiply the six Linear weight matrices by 3.
Each render fixture also stores the last-interval density of every ray so tests can mask
the step discontinuity of the 1e10-wide last interval.
"""
import os
import sys
import warnings

import numpy as np
import torch

REFERENCE = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REFERENCE)
warnings.filterwarnings("ignore")

from nerf.model import NeRF            # noqa: E402  (the reference)
from nerf.dataset import PixelRayDataset, ImageRayDataset  # noqa: E402

LINEAR_SLOTS = (0, 3, 6, 9, 12, 15)


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.0f} KiB")


def make_model(scale=1.0, focal_length=112.0):
    torch.manual_seed(0)
    model = NeRF(focal_length=focal_length)
    if scale != 1.0:
        with torch.no_grad():
            for slot in LINEAR_SLOTS:
                model.prediction_heads[slot].weight.mul_(scale)
    return model


def look_at(camera_o):
    cam = torch.tensor([camera_o], dtype=torch.float32)
    eye = -cam / torch.linalg.norm(cam, dim=-1, keepdim=True)
    z = torch.tensor([[0.0, 0.0, 1.0]])
    up = z - (z * eye).sum(-1, keepdim=True) * eye
    up = up / torch.linalg.norm(up, dim=-1, keepdim=True)
    return cam, NeRF.get_rotation_matrix(eye, up)


def frame_rays(cam_o, cam_r, h, w, focal):
    rays = NeRF.generate_rays(h, w, focal)
    o = torch.broadcast_to(cam_o[:, None, None, :], [1, h, w, 3])
    r = torch.broadcast_to(cam_r[:, None, None, :, :], [1, h, w, 3, 3])
    rays_o, rays_d = NeRF.rays_to_world_coordinates(rays.unsqueeze(0), o, r)
    return rays_o.reshape(h, w, 3), rays_d.reshape(h, w, 3)


def stages(model, rays_o, rays_d, num_samples):
    """Re-run the reference's own methods stage by stage (deterministic path)."""
    with torch.no_grad():
        t = model.sample_along_rays(rays_o, rays_d, num_samples, randomly_sample=False)
        means, covs, h = model.integrated_pe(rays_o, rays_d, t)
        _, density, color, seg = model.forward(rays_o, rays_d, t)
        weights = NeRF.alpha_compositing_coefficients(means, density)
        rgb, seg_out = model.render_rays(rays_o, rays_d, num_samples)
    return dict(t=t, means=means, covs=covs, h=h, density=density, color=color, seg=seg,
                weights=weights, rgb=rgb[:, 0], seg_out=seg_out[:, 0],
                last_density=density[:, -1, 0])


def main():
    cam_o, cam_r = look_at([0.0, -3.0, 2.6])

    # ---- parameters (seed 0, default init) -------------------------------
    model = make_model()
    save("params_seed0", **{k: v for k, v in model.state_dict().items()})

    # ---- G1 / G2: stage vectors, 64 rays of the 100x100 frame, S = 64 ----
    fo, fd = frame_rays(cam_o, cam_r, 100, 100, 112.0)
    idx = torch.arange(64) * 157 % 10000          # scattered pixels
    rays_o, rays_d = fo.reshape(-1, 3)[idx].contiguous(), fd.reshape(-1, 3)[idx].contiguous()
    for name, scale in (("g1_stages", 1.0), ("g2_stages_x3", 3.0)):
        st = stages(make_model(scale), rays_o, rays_d, 64)
        st["h"] = st["h"][:8]
        st["seg"] = st["seg"][:16]
        save(name, rays_o=rays_o, rays_d=rays_d, pixel_index=idx, weight_scale=scale, **st)

    # ---- G3: render_image 1x100x100, S = 64 (BASELINE configs 1/2) -------
    for name, scale in (("g3_image100", 1.0), ("g3_image100_x3", 3.0)):
        m = make_model(scale)
        with torch.no_grad():
            img, seg = m.render_image(cam_o, cam_r, 100, 100, 112.0, 64)
            t = m.sample_along_rays(fo.reshape(-1, 3), fd.reshape(-1, 3), 64, randomly_sample=False)
            _, density, _, _ = m.forward(fo.reshape(-1, 3), fd.reshape(-1, 3), t)
        save(name, camera_o=cam_o, camera_r=cam_r, image=img[0],
             seg_argmax=seg[0].argmax(-1).to(torch.uint8), seg_rows=seg[0, ::25],
             last_density=density[:, -1, 0].reshape(100, 100), weight_scale=scale)

    # ---- G4: 16x16 crop of the 800x800 / f = 896 frame at S = 128, 192 ---
    fo8, fd8 = frame_rays(cam_o, cam_r, 800, 800, 896.0)
    r0, c0 = 392, 392
    crop_o = fo8[r0:r0 + 16, c0:c0 + 16].reshape(-1, 3).contiguous()
    crop_d = fd8[r0:r0 + 16, c0:c0 + 16].reshape(-1, 3).contiguous()
    # plus a crop in the frame corner (largest |d|)
    corner_o = fo8[:16, :16].reshape(-1, 3).contiguous()
    corner_d = fd8[:16, :16].reshape(-1, 3).contiguous()
    for scale, tag in ((1.0, ""), (3.0, "_x3")):
        m = make_model(scale, focal_length=896.0)
        out = {}
        for s in (128, 192):
            for nm, (o, d) in (("center", (crop_o, crop_d)), ("corner", (corner_o, corner_d))):
                st = stages(m, o, d, s)
                out[f"rgb_{nm}_{s}"] = st["rgb"]
                out[f"seg_{nm}_{s}"] = st["seg_out"]
                out[f"last_density_{nm}_{s}"] = st["last_density"]
        save("g4_crop800" + tag, camera_o=cam_o, camera_r=cam_r, row0=r0, col0=c0,
             center_o=crop_o, center_d=crop_d, corner_o=corner_o, corner_d=corner_d,
             weight_scale=scale, **out)

    # ---- G5: stochastic path with captured draws -------------------------
    n, s = 256, 64
    idx5 = torch.arange(n) * 37 % 10000
    o5, d5 = fo.reshape(-1, 3)[idx5].contiguous(), fd.reshape(-1, 3)[idx5].contiguous()
    for scale, tag in ((1.0, ""), (3.0, "_x3")):
        m = make_model(scale)
        torch.manual_seed(1234)
        u = torch.rand(n, s)                      # same order as model.py:432 then :652
        noise = torch.randn(n, s - 1, 1)
        torch.manual_seed(1234)
        with torch.no_grad():
            rgb, seg = m.render_rays(o5, d5, s, randomly_sample=True, density_noise_std=1.0)
        save("g5_stochastic" + tag, rays_o=o5, rays_d=d5, u=u, noise=noise, noise_std=1.0,
             rgb=rgb[:, 0], seg_out=seg[:, 0], weight_scale=scale)

    # ---- G6: one training step (MSE, Adam lr 1e-4), fixed u / noise ------
    for scale, tag in ((1.0, ""), (3.0, "_x3")):
        m = make_model(scale)
        opt = torch.optim.Adam(m.parameters(), lr=1e-4)
        torch.manual_seed(99)
        target = torch.rand(n, 3)
        torch.manual_seed(4321)
        u = torch.rand(n, s)
        noise = torch.randn(n, s - 1, 1)
        torch.manual_seed(4321)
        pixels, _ = m.render_rays(o5, d5, s, randomly_sample=True, density_noise_std=1.0)
        loss = ((pixels - target.unsqueeze(1)) ** 2).mean()
        opt.zero_grad()
        loss.backward()
        grads = {"grad." + k: p.grad.clone() for k, p in m.named_parameters()}
        opt.step()
        after = {"after." + k: p.detach().clone() for k, p in m.named_parameters()
                 if p.numel() <= 256 * 54}
        save("g6_train_step" + tag, rays_o=o5, rays_d=d5, u=u, noise=noise, noise_std=1.0,
             target=target, loss=loss.detach(), rgb=pixels[:, 0].detach(),
             weight_scale=scale, **grads, **after)

    # ---- G7: statics -----------------------------------------------------
    yaw = torch.tensor([0.3, -1.2, 2.5])
    elev = torch.tensor([0.1, 0.7, -0.4])
    eye = NeRF.spherical_to_cartesian(yaw, elev)
    up = torch.tensor([[0.0, 0.0, 1.0]]).repeat(3, 1)
    rays57 = NeRF.generate_rays(5, 7, 112.0)
    rot = NeRF.get_rotation_matrix(eye, up)
    w_o, w_d = NeRF.rays_to_world_coordinates(rays57[None], eye[:, None, None, :] * 2.0,
                                              rot[:, None, None, :, :])
    save("g7_statics", yaw=yaw, elevation=elev, cartesian=eye, up=up, rotation=rot,
         rays_5x7=rays57, rays_800=NeRF.generate_rays(800, 800, 896.0)[::100, ::100],
         world_o=torch.broadcast_to(w_o, w_d.shape), world_d=w_d,
         t64=model.sample_along_rays(torch.zeros(1, 3), torch.zeros(1, 3), 64,
                                     randomly_sample=False)[0],
         t128=model.sample_along_rays(torch.zeros(1, 3), torch.zeros(1, 3), 128,
                                      randomly_sample=False)[0],
         t192=model.sample_along_rays(torch.zeros(1, 3), torch.zeros(1, 3), 192,
                                      randomly_sample=False)[0])

    # ---- G8: PixelRayDataset index decode --------------------------------
    torch.manual_seed(7)
    images = torch.rand(3, 6, 5, 3)
    segm = torch.randint(0, 50, (3, 6, 5))
    poses = torch.eye(4).repeat(3, 1, 1)
    for b in range(3):
        poses[b, :3, :3] = rot[b]
        poses[b, :3, 3] = eye[b] * 2.0
    ds = PixelRayDataset(images, segm, poses, 112.0)
    picks = [0, 1, 4, 5, 29, 30, 61, 89]
    items = [ds[i] for i in picks]
    save("g8_pixel_dataset", images=images, segmentation=segm, poses=poses, picks=picks,
         length=len(ds),
         **{k: torch.stack([it[k] for it in items]) for k in
            ("image_wi", "image_hi", "image_bi", "pixels", "label", "rays", "pose_o", "pose_d",
             "rays_o", "rays_d")})

    # ---- G10: ImageRayDataset (nerf/dataset.py:6-172): block-stratified batches, seeded CPU draws ---------
    torch.manual_seed(11)
    images = torch.rand(4, 12, 10, 3)
    states = torch.randn(4, 5)
    poses = torch.eye(4).repeat(4, 1, 1)
    for b in range(4):
        poses[b, :3, :3] = rot[b % 3]
        poses[b, :3, 3] = eye[b % 3] * (1.5 + b)
    ds = ImageRayDataset(images, poses, states, 112.0, num_vertical_blocks=3, num_horizontal_blocks=2,
                         num_samples_per_block=2)
    picks = [0, 7, 19, 20, 39]
    torch.manual_seed(12)                      # the draws of the five items, in this order
    items = [ds[i] for i in picks]
    save("g10_image_dataset", images=images, states=states, poses=poses, picks=picks, length=len(ds),
         **{k: torch.stack([it[k] for it in items]) for k in
            ("image_bi", "image_hi", "image_wi", "pixels", "states_out", "rays", "pose_o", "pose_d", "rays_o", "rays_d")
            if k != "states_out"},
         states_out=torch.stack([it["states"] for it in items]))

    # ---- G11: NARROW networks (constructor keywords of nerf/model.py:471-475): render + one training gradient ----
    # hidden_size 128 / encoding 32 / 50 classes, hidden 64 / encoding 16 / 7 classes, hidden 40 / encoding 10 / 3 classes
    # — what the kernels instantiated at 8 and 4 register tiles per sample are held to (weights x 2: sharper fields)
    # ---- G12: other color_outputs (constructor keyword of nerf/model.py:471; :541-542, :591-592, :660) -----------------
    # 1 channel with the default 50 classes; 4 channels (crossing into the kernels' second lane group of colors) on a
    # 128-wide network with 9 classes: render, per-sample field, training loss and all 22 gradients, like G11
    for tag, kw in (("h128", dict(hidden_size=128)), ("h64", dict(hidden_size=64, encoding_size=16, segmentation_outputs=7)),
                    ("h40", dict(hidden_size=40, encoding_size=10, segmentation_outputs=3)),
                    ("c1", dict(color_outputs=1)), ("c4", dict(color_outputs=4, hidden_size=128, segmentation_outputs=9))):
        # The network of seed 21 as it comes: no search for a seed without borderline ReLU gates (round 5 did that — with
        # hidden 64 this one has a gate 8.7e-8 from zero, which the kernels take on the other side).  The tests compare
        # gate-aware instead (tests/gate_aware.py); the distance of the closest gate is recorded for information.
        seed = 21
        torch.manual_seed(seed)
        m = NeRF(**kw)
        with torch.no_grad():
            for slot in LINEAR_SLOTS:
                m.prediction_heads[slot].weight.mul_(2.0)
        closest = []
        hooks = [m.prediction_heads[i].register_forward_hook(lambda mod, inp, out: closest.append(float(out.abs().min())))
                 for i in (1, 4, 7, 10, 13)]
        torch.manual_seed(22)
        u = torch.rand(64, 32)
        noise = torch.randn(64, 31, 1)
        target = torch.rand(64, kw.get("color_outputs", 3))
        # the reference's render_rays draws rand [N,S] then randn [N,S-1,1] (model.py:432, :652): replay them
        torch.manual_seed(22)
        pix, _ = m.render_rays(rays_o, rays_d, 32, randomly_sample=True, density_noise_std=0.5)
        for hk in hooks:
            hk.remove()
        st = stages(m, rays_o, rays_d, 48)
        loss = ((pix - target.unsqueeze(1)) ** 2).mean()
        m.zero_grad()
        loss.backward()
        save(("g12_colors_" if tag.startswith("c") else "g11_narrow_") + tag, rays_o=rays_o, rays_d=rays_d, u=u, noise=noise, noise_std=0.5, target=target,
             loss=loss.detach(), rgb=st["rgb"], seg_out=st["seg_out"], density=st["density"], color=st["color"],
             last_density=st["last_density"], init_seed=seed, closest_gate=min(closest),
             **{"param." + k: v for k, v in m.state_dict().items()},
             **{"grad." + k: p.grad for k, p in m.named_parameters()})

if __name__ == "__main__":
    main()
