"""The LEGACY 8 x 256 network of examples/nerf.pth: PARITY UNPINNED — no
reference code exists for it, so the bar is the HIP kernel against oracle/legacy_oracle.py (the CPU
statement of the structure recovered from the checkpoint) at the north-star tolerance, 1e-4 absolute
on rendered RGB, on the reference's own trained weights (fixture G9) and on random ones."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import legacy_oracle as L

CFG = L.default_config()


def checkpoint():
    g = load_golden("g9_legacy_checkpoint")
    return {k[len("param."):]: v for k, v in g.items() if k.startswith("param.")}, g


def test_oracle_reproduces_its_fixture_and_the_checkpoint_layout():
    params, g = checkpoint()
    assert list(L.state_dict_keys()) == [k for k in L.state_dict_keys() if k in params] and len(params) == 44
    assert params["block_0.0.weight"].shape == (256, 60) and params["block_1.0.weight"].shape == (256, 316)
    assert params["block_2.0.weight"].shape == (256, 292) and params["color.weight"].shape == (3, 256)
    assert L.flops_per_sample() == 1261568
    with torch.no_grad():
        rgb = L.render_rays(params, CFG, g["rays_o"], g["rays_d"], 2.0, 6.0, 40)
    assert (rgb - g["rgb"]).abs().max() <= 2e-6
    # the trained scene is a scene: the oracle's 40x40 render has a dark background and a bright object
    img = g["image40"]
    assert float(img[:4].mean()) < 0.02 and float(img[12:30, 10:30].mean()) > 0.2
    enc = L.positional_encoding(torch.tensor([[0.5, -0.25, 0.125]]), 10, CFG["multiplier"])
    assert enc.shape == (1, 60) and abs(float(enc[0, 0]) - np.sin(0.5 * np.pi)) < 1e-6     # [x: sin f0.., cos f0..]
    assert abs(float(enc[0, 10]) - np.cos(0.5 * np.pi)) < 1e-6 and abs(float(enc[0, 20]) - np.sin(-0.25 * np.pi)) < 1e-6


def test_module_tree_takes_the_checkpoint():
    from nerf_amd.legacy import LegacyNeRF8x256
    params, _ = checkpoint()
    model = LegacyNeRF8x256()
    assert sorted(model.state_dict().keys()) == sorted(params.keys())
    model.load_state_dict(params)
    names = {id(p): k for k, p in model.named_parameters()}
    assert [names[id(p)] for p in model._param_list()] == L.state_dict_keys()
    with pytest.raises(RuntimeError):                           # no CPU path
        model.render_rays(torch.zeros(4, 3), torch.ones(4, 3), 2.0, 6.0, 8)


def _model(dev, params, precision="fp32"):
    from nerf_amd.legacy import LegacyNeRF8x256
    model = LegacyNeRF8x256()
    model.load_state_dict(params)
    model.precision = precision
    return model.to(dev)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_trained_checkpoint_vs_oracle(precision):
    """Both arithmetics of the kernel at the same tolerances (f16x3: f16-pair operands, fp32 results)."""
    dev = torch.device("cuda:0")
    params, g = checkpoint()
    model = _model(dev, params, precision)
    o, d = g["rays_o"].to(dev), g["rays_d"].to(dev)
    rgb, raw, weights = model.render_rays(o, d, 2.0, 6.0, 40, per_sample=True)
    assert (raw[..., :1].cpu() - g["density"]).abs().max() <= 2e-3 * max(1.0, float(g["density"].abs().max()))
    assert (raw[..., 1:].cpu() - g["color"]).abs().max() <= 2e-3
    assert (weights.cpu() - g["weights"][..., 0]).abs().max() <= 1e-4
    assert (rgb.cpu() - g["rgb"]).abs().max() <= 1e-4
    rgb_s = model.render_rays(o, d, 2.0, 6.0, 40, randomly_sample=True, density_noise_std=0.5,
                              u=g["u"].to(dev), noise=g["noise"][..., 0].to(dev))
    assert (rgb_s.cpu() - g["rgb_stochastic"]).abs().max() <= 1e-4
    # the frame through in-kernel ray generation, whole and in row blocks
    cam_o, cam_r = g["camera_o"].to(dev), g["camera_r"].to(dev)
    img = model.render_image(cam_o, cam_r, 40, 40, 55.5, 2.0, 6.0, 64)
    assert img.shape == (1, 40, 40, 3)
    assert (img[0].cpu() - g["image40"]).abs().max() <= 1e-4
    parts = [model.render_image(cam_o, cam_r, 40, 40, 55.5, 2.0, 6.0, 64, row_begin=r, row_end=min(r + 13, 40))
             for r in range(0, 40, 13)]
    assert torch.equal(torch.cat(parts, dim=1), img)
    again = model.render_image(cam_o, cam_r, 40, 40, 55.5, 2.0, 6.0, 64)
    assert torch.equal(again, img)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
@pytest.mark.parametrize("n_rays,num_samples", [(1, 2), (5, 17), (37, 33), (130, 64), (3, 100)])
def test_random_weights_ragged_shapes_vs_oracle(n_rays, num_samples, precision):
    dev = torch.device("cuda:0")
    params = L.init_params(seed=n_rays)
    for k in params:                                            # sharper field: densities of both signs, O(1) logits
        if k.endswith(".weight") and params[k].dim() == 2:
            params[k] = params[k] * 2.0
    model = _model(dev, params, precision)
    g = torch.Generator().manual_seed(100 + n_rays)
    o = torch.randn(n_rays, 3, generator=g)
    d = torch.randn(n_rays, 3, generator=g)
    with torch.no_grad():
        ref, st = L.render_rays(params, CFG, o, d, 0.5, 5.0, num_samples, return_stages=True)
    rgb, raw, weights = model.render_rays(o.to(dev), d.to(dev), 0.5, 5.0, num_samples, per_sample=True)
    assert rgb.shape == (n_rays, 3) and raw.shape == (n_rays, num_samples, 4)
    # the 1e10-wide last interval makes its weight a step in the sign of that density
    ok = st["density"][:, -1, 0].abs() > 1e-4
    assert (rgb.cpu() - ref)[ok].abs().max() <= 1e-4 if ok.any() else True
    assert (raw[..., :1].cpu() - st["density"]).abs().max() <= 1e-3
