"""CPU tests: pin the oracle (oracle/nerf_oracle.py) to fixtures generated from the reference
itself (tests/golden/make_golden.py).  Tolerances: <= 2e-6 absolute on rendered RGB
(fp32-vs-fp64 noise floor of the reference is 1.1e-6)."""
import pytest
import torch

from conftest import golden_params, load_golden, stable_rays
from oracle import nerf_oracle as O

CFG = O.default_config()
RGB_TOL = 2e-6


def test_init_params_match_reference_seed0():
    ref = golden_params()
    mine = O.init_params(seed=0)
    assert set(ref) == set(mine)
    for k in ref:
        assert ref[k].shape == mine[k].shape, k
        assert torch.equal(ref[k], mine[k]), k
    assert sum(v.numel() for k, v in mine.items() if k.startswith("prediction")) == 304438


@pytest.mark.parametrize("name,scale", [("g1_stages", 1.0), ("g2_stages_x3", 3.0)])
def test_stage_vectors(name, scale):
    g = load_golden(name)
    p = golden_params(scale)
    with torch.no_grad():
        rgb, seg_out, st = O.render_rays(p, CFG, g["rays_o"], g["rays_d"], 64, return_stages=True)
    assert torch.equal(st["t"], g["t"])
    assert (st["means"] - g["means"]).abs().max() <= 1e-6
    assert (st["covs"] - g["covs"]).abs().max() <= 1e-7
    assert (st["h"][:8] - g["h"]).abs().max() <= 2e-6
    assert (st["density"] - g["density"]).abs().max() <= 1e-5
    assert (st["color"] - g["color"]).abs().max() <= 1e-5
    assert (st["seg"][:16] - g["seg"]).abs().max() <= 1e-5
    ok = stable_rays(g["last_density"])
    assert ok.sum() >= 48
    assert (st["weights"] - g["weights"])[ok].abs().max() <= RGB_TOL
    assert (rgb - g["rgb"])[ok].abs().max() <= RGB_TOL
    assert (seg_out - g["seg_out"])[ok].abs().max() <= 2e-5


@pytest.mark.parametrize("name,scale", [("g3_image100", 1.0), ("g3_image100_x3", 3.0)])
def test_render_image_100(name, scale):
    g = load_golden(name)
    p = golden_params(scale)
    with torch.no_grad():
        img, seg = O.render_image(p, CFG, g["camera_o"], g["camera_r"], 100, 100, 112.0, 64)
    ok = stable_rays(g["last_density"])
    assert ok.float().mean() > 0.9
    assert (img[0] - g["image"])[ok].abs().max() <= RGB_TOL
    assert (seg[0, ::25] - g["seg_rows"])[ok[::25]].abs().max() <= 2e-5
    assert O.psnr(img[0][ok], g["image"][ok]) > 100.0


@pytest.mark.parametrize("name,scale", [("g4_crop800", 1.0), ("g4_crop800_x3", 3.0)])
def test_crop_of_800_frame(name, scale):
    g = load_golden(name)
    p = golden_params(scale)
    cfg = dict(CFG, focal_length=896.0)
    # the crop rays are exactly those render_image builds for the 800x800 frame
    o, d = O.image_rays(g["camera_o"], g["camera_r"], 800, 800, 896.0)
    o, d = o.reshape(800, 800, 3), d.reshape(800, 800, 3)
    r0, c0 = int(g["row0"]), int(g["col0"])
    assert torch.equal(d[r0:r0 + 16, c0:c0 + 16].reshape(-1, 3), g["center_d"])
    assert torch.equal(d[:16, :16].reshape(-1, 3), g["corner_d"])
    for s in (128, 192):
        for nm in ("center", "corner"):
            with torch.no_grad():
                rgb, seg = O.render_rays(p, cfg, g[f"{nm}_o"], g[f"{nm}_d"], s)
            ok = stable_rays(g[f"last_density_{nm}_{s}"])
            assert (rgb - g[f"rgb_{nm}_{s}"])[ok].abs().max() <= RGB_TOL
            assert (seg - g[f"seg_{nm}_{s}"])[ok].abs().max() <= 2e-5


@pytest.mark.parametrize("name,scale", [("g5_stochastic", 1.0), ("g5_stochastic_x3", 3.0)])
def test_stochastic_with_captured_draws(name, scale):
    g = load_golden(name)
    p = golden_params(scale)
    with torch.no_grad():
        rgb, seg = O.render_rays(p, CFG, g["rays_o"], g["rays_d"], 64, u=g["u"],
                                 noise=g["noise"], density_noise_std=float(g["noise_std"]))
    # noise of std 1 keeps the last-interval density far from the step for all but a few rays
    bad = (rgb - g["rgb"]).abs().amax(-1) > RGB_TOL
    assert bad.sum() <= 2
    assert (seg - g["seg_out"])[~bad].abs().max() <= 2e-5


@pytest.mark.parametrize("name,scale", [("g6_train_step", 1.0), ("g6_train_step_x3", 3.0)])
def test_training_gradients(name, scale):
    g = load_golden(name)
    p = {k: v.clone().requires_grad_(k.startswith("prediction")) for k, v in golden_params(scale).items()}
    loss = O.training_loss(p, CFG, g["rays_o"], g["rays_d"], 64, g["target"], g["u"], g["noise"],
                           float(g["noise_std"]))
    assert abs(float(loss.detach()) - float(g["loss"])) <= 1e-6
    loss.backward()
    n_checked = 0
    for k, v in p.items():
        if not k.startswith("prediction"):
            continue
        ref = g["grad." + k]
        scale_ = ref.abs().max().clamp(min=1e-8)
        assert (v.grad - ref).abs().max() <= 2e-4 * scale_ + 1e-9, k
        n_checked += 1
    assert n_checked == 22
    # RGB-only loss: segmentation rows of the last Linear get exactly zero gradient
    assert torch.count_nonzero(p["prediction_heads.15.weight"].grad[4:]) == 0
    # one Adam step (lr 1e-4) on the small tensors
    names = [k for k in p if k.startswith("prediction")]
    opt = torch.optim.Adam([p[k] for k in names], lr=1e-4)
    opt.step()
    for k in names:
        if "after." + k in g:
            assert (p[k].detach() - g["after." + k]).abs().max() <= 2.1e-4, k


def test_statics():
    g = load_golden("g7_statics")
    assert torch.equal(O.pinhole_rays(5, 7, 112.0), g["rays_5x7"])
    assert torch.equal(O.pinhole_rays(800, 800, 896.0)[::100, ::100], g["rays_800"])
    assert torch.equal(O.spherical_to_cartesian(g["yaw"], g["elevation"]), g["cartesian"])
    rot = O.rotation_from_eye_up(g["cartesian"], g["up"])
    assert torch.equal(rot, g["rotation"])
    _, wd = O.rays_to_world(g["rays_5x7"][None], g["cartesian"][:, None, None, :] * 2.0,
                            rot[:, None, None, :, :])
    assert torch.equal(wd, g["world_d"])
    p = golden_params()
    for s in (64, 128, 192):
        assert torch.equal(O.sample_t(p, 1, s)[0], g[f"t{s}"])
    assert abs(float(O.box_diagonal(p)) - 69.28203) < 1e-4


NARROW = {"h128": dict(hidden_size=128), "h64": dict(hidden_size=64, encoding_size=16, segmentation_outputs=7),
          "h40": dict(hidden_size=40, encoding_size=10, segmentation_outputs=3),
          # fixture G12: the reference's own networks with another color_outputs (nerf/model.py:471, :541-542, :591-592)
          "c1": dict(color_outputs=1), "c4": dict(color_outputs=4, hidden_size=128, segmentation_outputs=9)}


def fixture_of(tag):
    return ("g12_colors_" if tag.startswith("c") else "g11_narrow_") + tag


@pytest.mark.parametrize("tag", sorted(NARROW))
def test_narrow_networks_match_the_reference(tag):
    """Fixtures G11 / G12: `NeRF(hidden_size=.., encoding_size=.., segmentation_outputs=.., color_outputs=..)` of the
    reference itself (nerf/model.py:471-475) — render, per-sample field, training loss and all 22 gradients — pins the
    oracle at the network shapes the narrow kernel instantiations and the run-time color count are tested against."""
    g = load_golden(fixture_of(tag))
    cfg = dict(O.default_config(), **NARROW[tag])
    params = {k[6:]: v for k, v in g.items() if k.startswith("param.")}
    with torch.no_grad():
        rgb, seg_out, st = O.render_rays(params, cfg, g["rays_o"], g["rays_d"], 48, return_stages=True)
    assert (st["density"] - g["density"]).abs().max() <= 1e-5 * max(1.0, float(g["density"].abs().max()))
    assert (st["color"] - g["color"]).abs().max() <= 1e-5 * max(1.0, float(g["color"].abs().max()))
    ok = stable_rays(g["last_density"])
    assert ok.sum() >= 48
    assert (rgb - g["rgb"])[ok].abs().max() <= RGB_TOL
    assert (seg_out - g["seg_out"])[ok].abs().max() <= 2e-5
    p = {k: v.clone().requires_grad_(k.startswith("prediction")) for k, v in params.items()}
    loss = O.training_loss(p, cfg, g["rays_o"], g["rays_d"], 32, g["target"], g["u"], g["noise"], float(g["noise_std"]))
    assert abs(float(loss.detach()) - float(g["loss"])) <= 1e-6
    loss.backward()
    for k, v in p.items():
        if v.grad is not None:
            ref = g["grad." + k]
            assert (v.grad - ref).abs().max() <= 2e-5 * ref.abs().max().clamp(min=1e-12), k
