"""INTEGRATION.md is executed, not only read: the reference-side ctypes stub printed there (section B) is
extracted from the document, bound as ``render_rays`` onto a PLAIN torch module that only holds the
reference's state-dict tensors (the reference's own ``NeRF`` would be that module; it cannot travel to
the GPU box), and held to the reference-generated fixtures G1 / G3 / G5 — so the documented binding
cannot rot.  Also here: the B > 1 stochastic ``render_image`` draws in the reference's order."""
import os
import re

import pytest
import torch
import torch.nn as nn

from conftest import ROOT, golden_params, load_golden, stable_rays
from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu


def integration_stub_source():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stubs = [b for b in blocks if "class _Args(ctypes.Structure)" in b]
    assert len(stubs) == 1, "INTEGRATION.md must carry exactly one reference-side stub"
    return stubs[0]


class ReferenceShapedModule(nn.Module):
    """What the stub needs of the reference's class: ``prediction_heads`` (nerf/model.py:525-542),
    the ``rays_min`` / ``rays_max`` buffers (:509-523) and ``focal_length``.  No renderer code."""

    def __init__(self, focal_length=112.0):
        super().__init__()
        self.focal_length = focal_length
        self.color_outputs, self.segmentation_outputs = 3, 50
        self.register_buffer("rays_min", torch.zeros(1, 1, 3))
        self.register_buffer("rays_max", torch.zeros(1, 1, 3))
        layers = [nn.Linear(96, 256), nn.LayerNorm(256), nn.ReLU()]
        for _ in range(4):
            layers += [nn.Linear(256, 256), nn.LayerNorm(256), nn.ReLU()]
        layers.append(nn.Linear(256, 54))
        self.prediction_heads = nn.Sequential(*layers)


def bound_module(dev, scale=1.0, focal_length=112.0):
    from nerf_amd import _lib
    src = integration_stub_source()
    assert 'ctypes.CDLL("libnerf_hip.so")' in src
    src = src.replace('ctypes.CDLL("libnerf_hip.so")', f"ctypes.CDLL({_lib.LIB_PATH!r})")   # no LD_LIBRARY_PATH here
    space = {"torch": torch}
    exec(compile(src, "INTEGRATION.md:stub", "exec"), space)
    module = ReferenceShapedModule(focal_length)
    module.load_state_dict(golden_params(scale))               # the reference's own keys and tensors
    module = module.to(dev)
    ReferenceShapedModule.render_rays = space["render_rays"]
    return module, space


def test_documented_stub_renders_the_reference_fixtures():
    dev = torch.device("cuda:0")
    for name, scale in (("g1_stages", 1.0), ("g2_stages_x3", 3.0)):
        g = load_golden(name)
        module, space = bound_module(dev, scale)
        with torch.no_grad():
            rgb, seg = module.render_rays(g["rays_o"].to(dev), g["rays_d"].to(dev), 64)
        assert rgb.shape == (64, 1, 3) and seg.shape == (64, 1, 50)
        ok = stable_rays(g["last_density"])
        assert (rgb[:, 0].cpu() - g["rgb"])[ok].abs().max() <= 1e-5
        assert (seg[:, 0].cpu() - g["seg_out"])[ok].abs().max() <= 1e-4
    # the struct the document prints is the header's, field for field
    from nerf_amd import _lib
    assert [(n, t) for n, t in space["_Args"]._fields_] == [(n, t) for n, t in _lib.RenderArgs._fields_]
    # G3: the 100 x 100 frame, ray by ray through the stub (the stub documents render_rays only)
    g = load_golden("g3_image100")
    module, _ = bound_module(dev, 1.0)
    rays_o, rays_d = O.image_rays(g["camera_o"], g["camera_r"], 100, 100, 112.0)
    with torch.no_grad():
        rgb, _ = module.render_rays(rays_o.to(dev), rays_d.to(dev), 64)
    ok = stable_rays(g["last_density"]).reshape(-1)
    assert (rgb[:, 0].cpu() - g["image"].reshape(-1, 3))[ok].abs().max() <= 1e-5


def test_documented_stub_draws_like_the_reference():
    """Stochastic path of the stub: rand [N,S] then randn [N,S-1,1] from torch's generator (nerf/model.py:432,
    :652).  With the device generator re-seeded the same draws feed the product's explicit-draw path (itself
    pinned to the reference by fixture G5): identical pixels."""
    from nerf_amd import NeRF
    dev = torch.device("cuda:0")
    g = load_golden("g5_stochastic")
    module, _ = bound_module(dev, 1.0)
    o, d = g["rays_o"].to(dev), g["rays_d"].to(dev)
    torch.manual_seed(123)
    with torch.no_grad():
        rgb, seg = module.render_rays(o, d, 64, randomly_sample=True, density_noise_std=float(g["noise_std"]))
    torch.manual_seed(123)
    u = torch.rand(256, 64, device=dev)
    noise = torch.randn(256, 63, 1, device=dev)
    model = NeRF()
    model.load_state_dict(golden_params(1.0))
    model = model.to(dev)
    with torch.no_grad():
        want, want_seg = model.render_rays(o, d, 64, randomly_sample=True, density_noise_std=float(g["noise_std"]),
                                           u=u, noise=noise)
    assert torch.equal(rgb, want) and torch.equal(seg, want_seg)


def test_stochastic_render_image_draws_in_the_reference_chunk_order():
    """The reference flattens B*H*W rays and splits THAT list by max_chunk_size, so a chunk may straddle two
    frames, and draws rand then randn per chunk (nerf/model.py:750-761).  B = 2 frames of 5 x 5 with chunks
    of 8 rays: chunk 3 covers rays 24..31 = the last ray of frame 0 and seven of frame 1."""
    from nerf_amd import NeRF
    dev = torch.device("cuda:0")
    model = NeRF(focal_length=5.6)
    model.load_state_dict(golden_params(3.0))
    model = model.to(dev)
    cam_o = torch.tensor([[0.0, -3.0, 2.6], [2.5, 1.5, 2.0]])
    cam_r = torch.cat([O.look_at_pose(c.tolist()) for c in cam_o])
    S, H, W, chunk, std = 24, 5, 5, 8, 0.5
    torch.manual_seed(77)
    with torch.no_grad():
        img, seg = model.render_image(cam_o.to(dev), cam_r.to(dev), H, W, 5.6, S, max_chunk_size=chunk,
                                      randomly_sample=True, density_noise_std=std)
    assert img.shape == (2, H, W, 3) and seg.shape == (2, H, W, 50)
    # the reference's loop, restated with the product's explicit-draw render_rays on the same generator
    rays = [O.image_rays(cam_o[b:b + 1], cam_r[b:b + 1], H, W, 5.6) for b in range(2)]
    rays_o = torch.cat([r[0] for r in rays]).to(dev)
    rays_d = torch.cat([r[1] for r in rays]).to(dev)
    torch.manual_seed(77)
    parts, us, noises = [], [], []
    with torch.no_grad():
        for o_i, d_i in zip(torch.split(rays_o, chunk), torch.split(rays_d, chunk)):
            n = o_i.shape[0]
            u = torch.rand(n, S, device=dev)
            noise = torch.randn(n, S - 1, 1, device=dev)
            us.append(u.cpu()), noises.append(noise.cpu())
            parts.append(model.render_rays(o_i, d_i, S, randomly_sample=True, density_noise_std=std,
                                           u=u, noise=noise)[0][:, 0])
    want = torch.cat(parts).reshape(2, H, W, 3)
    assert (img - want).abs().max() <= 2e-6          # in-kernel ray generation vs ray arrays: same rays to an ulp
    # and against the CPU port of the reference on those very draws
    params = golden_params(3.0)
    cfg = dict(O.default_config(), focal_length=5.6)
    with torch.no_grad():
        ref, _, st = O.render_rays(params, cfg, rays_o.cpu(), rays_d.cpu(), S, u=torch.cat(us),
                                   noise=torch.cat(noises), density_noise_std=std, return_stages=True)
    ok = st["density"][:, -1, 0].abs() > 1e-5
    assert (img.reshape(-1, 3).cpu() - ref)[ok].abs().max() <= 1e-5


def test_deterministic_render_image_leaves_the_generator_where_the_reference_does():
    """On the deterministic path the reference still draws ``randn([n, S-1, 1])`` once per chunk of its loop and
    multiplies it by zero (nerf/model.py:652-654, :757-761).  The one-launch ``render_image`` draws nothing but
    advances torch's device generator by exactly that much (rng="torch"), so the caller's NEXT draw is the one it
    would get after the reference's call.  B = 2 frames of 5 x 7 in chunks of 16 rays: four full chunks + a tail
    of 6, the third chunk straddling the frames."""
    from nerf_amd import NeRF
    dev = torch.device("cuda:0")
    model = NeRF(focal_length=5.6)
    model.load_state_dict(golden_params(1.0))
    model = model.to(dev)
    cam_o = torch.tensor([[0.0, -3.0, 2.6], [2.5, 1.5, 2.0]])
    cam_r = torch.cat([O.look_at_pose(c.tolist()) for c in cam_o])
    S, H, W, chunk = 24, 5, 7, 16
    gen = torch.cuda.default_generators[0]
    torch.manual_seed(123)
    with torch.no_grad():
        model.render_image(cam_o.to(dev), cam_r.to(dev), H, W, 5.6, S, max_chunk_size=chunk)
    offset_here, next_here = gen.get_offset(), torch.rand(33, device=dev)
    # the reference's loop consumes, per chunk, one randn of the chunk's [n, S-1, 1] (and nothing else)
    torch.manual_seed(123)
    for n in [chunk] * (2 * H * W // chunk) + [2 * H * W % chunk]:
        torch.randn(n, S - 1, 1, device=dev)
    assert gen.get_offset() == offset_here and offset_here > 0
    assert torch.equal(torch.rand(33, device=dev), next_here)
    # in-kernel draws (rng="philox") leave torch's generator alone by contract
    model.rng = "philox"
    torch.manual_seed(123)
    with torch.no_grad():
        model.render_image(cam_o.to(dev), cam_r.to(dev), H, W, 5.6, S, max_chunk_size=chunk)
    assert gen.get_offset() == 0
