"""BASELINE config 3 (coarse + fine hierarchical sampling).  PARITY UNPINNED: the reference has no
code for it, so the spec is the oracle's restatement of Mildenhall et al. 2020 section 5.2; the HIP
resampler and the two-stage render are held to that."""
import pytest
import torch

from conftest import golden_params, stable_rays
from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu
CFG = O.default_config()


@pytest.fixture(scope="module")
def setup():
    from nerf_amd import NeRF
    dev = torch.device("cuda:0")
    params = golden_params(3.0)
    model = NeRF()
    model.load_state_dict(params)
    return dev, params, model.to(dev)


@pytest.mark.parametrize("s_c,s_f", [(64, 128), (9, 5), (33, 64), (2, 7), (1000, 1000)])
def test_resampler_vs_oracle(setup, s_c, s_f):
    dev, params, model = setup
    torch.manual_seed(s_c)
    n = 130
    t_c = torch.sort(torch.rand(n, s_c) * 60 + 0.1, dim=-1).values
    w = torch.rand(n, s_c - 1) ** 4
    w[3] = 0.0                                       # a ray that hit nothing: uniform PDF
    w[5, : (s_c - 1) // 2] = 0.0
    for u in (None, torch.sort(torch.rand(n, s_f), dim=-1).values.clamp(max=1 - 1e-6)):
        got = model.resample_fenceposts(t_c.to(dev), w.to(dev), s_f, u=None if u is None else u.to(dev))
        ref = O.resample_fenceposts(t_c, w, s_f, u=u)
        assert got.shape == (n, s_c + s_f)
        assert (got[:, 1:] >= got[:, :-1]).all()     # sortedness
        # BIT-EXACT: the kernel restates the oracle operation by operation (torch.cumsum's double accumulator,
        # IEEE division, no FMA contraction, the same tie rule in the search), so even fenceposts inside
        # intervals of near-zero mass — t = t0 + (u - c0) / (c1 - c0) * dt with c1 - c0 ~ 1e-7 — agree exactly.
        assert torch.equal(got.cpu(), ref)
        # the coarse fenceposts survive verbatim in the union
        merged = torch.sort(torch.cat([got.cpu(), t_c], dim=-1), dim=-1).values
        assert (merged[:, 1:] == merged[:, :-1]).sum(-1).min() >= s_c


def test_two_stage_render_vs_oracle(setup):
    dev, params, model = setup
    torch.manual_seed(9)
    cam_o = torch.tensor([[0.0, -3.0, 2.6]])
    rays_o, rays_d = O.image_rays(cam_o, O.look_at_pose([0.0, -3.0, 2.6]), 20, 20, 22.4)
    o, d = rays_o.to(dev), rays_d.to(dev)
    with torch.no_grad():
        img, seg = model.render_rays_hierarchical(o, d, 64, 128)
        # the fenceposts the fine stage used (deterministic, so recomputable from the same pieces)
        t_c = model.sample_along_rays(o, d, 64, randomly_sample=False).contiguous()
        _, _, _, _, w_c = model._launch(400, 64, dev, rays_o=o, rays_d=d, t_values=t_c, want_weights=True)
        t_u = model.resample_fenceposts(t_c, w_c, 128).cpu()
        ref_c, seg_c, w_ref = O.render_rays_t(params, CFG, rays_o, rays_d, O.sample_t(params, 400, 64))
        ref_f, seg_f, _ = O.render_rays_t(params, CFG, rays_o, rays_d, t_u)     # oracle on the SAME posts
        ref_img, ref_seg, _ = O.render_rays_hierarchical(params, CFG, rays_o, rays_d, 64, 128)
        _, _, _, dens_c, _, _ = O.field(params, CFG, rays_o, rays_d, O.sample_t(params, 400, 64))
        _, _, _, dens_f, _, _ = O.field(params, CFG, rays_o, rays_d, t_u)
    assert img.shape == (400, 2, 3) and seg.shape == (400, 2, 50)
    # mask the last-interval step in either stage
    ok = stable_rays(dens_c[:, -1, 0]) & stable_rays(dens_f[:, -1, 0])
    assert ok.sum() > 300
    assert (w_c.cpu() - w_ref)[ok].abs().max() <= 1e-5
    assert (img[:, 0].cpu() - ref_c)[ok].abs().max() <= 1e-5
    assert (img[:, 1].cpu() - ref_f)[ok].abs().max() <= 1e-5           # fused fine render, same posts
    assert (seg[:, 1].cpu() - seg_f)[ok].abs().max() <= 1e-4
    # end to end against the oracle's own pipeline: the coarse weights differ by rounding (1e-7), and fine posts
    # inside near-empty intervals are ill-conditioned in them, which moves a few rays' quadrature slightly
    e2e_all = (img.cpu() - ref_img).abs().amax((-1, -2))
    e2e = e2e_all[ok]
    assert (e2e <= 1e-4).float().mean() >= 0.98 and e2e.max() <= 2e-2
    # ... and that is ALL the loose tail is allowed to be.  (i) A fine post t = t0 + (u - c0) / (c1 - c0) * dt moves
    # by (CDF rounding) x dt / (mass of its interval): rays whose 128 fine posts all land in coarse intervals of PDF
    # mass >= 1e-3 are well conditioned (351 of these 400 rays; with weights perturbed by 3e-7, three times the
    # measured kernel-vs-oracle difference, the ORACLE moves those rays by <= 6e-5 and the others by up to 4e-3) and
    # are held to 1e-4 without exception.
    pdf = (w_ref + 1e-5) / (w_ref + 1e-5).sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros(400, 1), torch.cumsum(pdf, -1)], -1)
    u_mid = ((torch.arange(128.0) + 0.5) / 128).expand(400, 128)
    hit = (torch.searchsorted(cdf.contiguous(), u_mid.contiguous(), right=True) - 1).clamp(0, 62)
    well = torch.gather(pdf, -1, hit).min(-1).values >= 1e-3
    assert (well & ok).sum() >= 300
    assert e2e_all[well & ok].max() <= 1e-4
    # (ii) for EVERY ray the deviation is what the oracle itself shows between its posts and the kernel's posts
    # (ref_f is the oracle on the kernel's union): nothing is left over for the renderer beyond the per-stage 1e-5.
    explained = (ref_f - ref_img[:, 1]).abs().amax(-1)
    assert ((img[:, 1].cpu() - ref_img[:, 1]).abs().amax(-1) <= explained + 1e-5)[ok].all()
    # the fine stage is a genuine refinement: close to the coarse render, not identical to it
    delta = (img[:, 1] - img[:, 0]).abs().max()
    assert 1e-6 < delta < 0.2
    full, _ = model.render_image_hierarchical(cam_o.to(dev), O.look_at_pose([0.0, -3.0, 2.6]).to(dev),
                                              20, 20, 22.4, 64, 128)
    # rays built by torch on the GPU here, on the CPU above: equal to rounding, not bitwise
    diff_all = (full.reshape(-1, 3) - img[:, 1]).abs().amax(-1).cpu()
    diff = diff_all[ok]
    assert (diff <= 1e-4).float().mean() >= 0.98 and diff.max() <= 2e-2
    assert diff_all[well & ok].max() <= 1e-4                    # the well-conditioned rays, as above


def test_two_stage_training_step(setup):
    dev, params, model = setup
    torch.manual_seed(2)
    o, d, tgt = torch.randn(64, 3).to(dev), torch.randn(64, 3).to(dev), torch.rand(64, 3).to(dev)
    model.zero_grad(set_to_none=True)
    img, _ = model.render_rays_hierarchical(o, d, 32, 48, randomly_sample=True, density_noise_std=0.5)
    assert img.requires_grad and img.shape == (64, 2, 3)
    ((img - tgt.unsqueeze(1)) ** 2).mean().backward()        # coarse + fine loss, as in the paper
    g = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    assert torch.isfinite(g).all() and float(g.abs().max()) > 0
