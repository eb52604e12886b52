"""Stage-by-stage GPU parity (through the C ABI): the tensors BETWEEN the kernel's fused steps, each
compared directly with what the reference computed (fixtures G1 / G5) or with the oracle on the same
inputs — fenceposts, Gaussian means AND covariances, the 96 encoded inputs, and per hidden layer the
LayerNorm-normalised activations and 1/std that the training forward saves for the backward.  A
defect in one step shows at that step here, not only through its effect on the pixels."""
import pytest
import torch

from conftest import golden_params, load_golden
from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu
CFG = O.default_config()


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def make_model(dev, scale=1.0):
    from nerf_amd import NeRF
    model = NeRF()
    model.load_state_dict(golden_params(scale))
    return model.to(dev)


@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_gaussians_vs_reference_fixture(dev, precision):
    """means, covs of the kernel's front end vs the reference's cast_rays output (fixture G1);
    the 96 features from them vs the reference's integrated_pos_enc output."""
    g = load_golden("g1_stages")
    model = make_model(dev)
    model.precision = precision
    mean, cov, h = model.integrated_pe(g["rays_o"].to(dev), g["rays_d"].to(dev), g["t"].to(dev))
    assert (mean.cpu() - g["means"]).abs().max() <= 2e-5          # |mean| up to 70
    rel = ((cov.cpu() - g["covs"]).abs() / g["covs"].abs().clamp(min=1e-12)).max()
    assert rel <= 2e-6, float(rel)                                # cancellation-prone: op order matters
    assert (h[:8].cpu() - g["h"]).abs().max() <= 1e-5


def test_fenceposts_vs_reference_fixture(dev):
    """Stratified fenceposts from captured draws vs the reference's sample_along_rays (fixture G5
    holds u; t restated by the oracle, pinned to the reference by tests/test_oracle_golden.py)."""
    g = load_golden("g5_stochastic")
    model = make_model(dev)
    n, S = g["u"].shape
    t = model.fenceposts_used(g["rays_o"].to(dev), g["rays_d"].to(dev), S, randomly_sample=True,
                              u=g["u"].to(dev))
    ref = O.sample_t(golden_params(), n, S, g["u"])
    assert (t.cpu() - ref).abs().max() <= 4e-6                    # 1 ulp at t = 69
    t0 = model.fenceposts_used(g["rays_o"].to(dev), g["rays_d"].to(dev), S)
    assert torch.equal(t0.cpu(), O.sample_t(golden_params(), n, S))


@pytest.mark.parametrize("scale", [1.0, 3.0])
@pytest.mark.parametrize("train_precision", ["fp32", "f16x3"])
def test_saved_stage_tensors_vs_oracle(dev, train_precision, scale):
    """What the training forward saves, read back from the workspace: h against the reference's
    fixture, x_hat / 1/std of every layer against the oracle's LayerNorm internals."""
    import workspace_mirror as W
    g = load_golden("g1_stages" if scale == 1.0 else "g2_stages_x3")
    params = golden_params(scale)
    model = make_model(dev, scale)
    model.train_precision = train_precision
    model.keep_workspace = True
    n, S = g["t"].shape
    rgb, _ = model.render_rays(g["rays_o"].to(dev), g["rays_d"].to(dev), S)
    assert rgb.requires_grad
    torch.cuda.synchronize()
    ws = model.last_workspace
    h = W.saved_h(ws, n, S).cpu()
    assert (h[:8] - g["h"]).abs().max() <= 1e-5
    t = O.sample_t(params, n, S)
    _, _, h_ref, _, _, _ = O.field(params, CFG, g["rays_o"], g["rays_d"], t)
    assert (h - h_ref).abs().max() <= 1e-5
    _, x_hats, rstds = O.mlp_stages(params, h_ref)
    tol = 3e-5 if train_precision == "fp32" else 6e-5
    for layer in range(5):
        x_hat = W.saved_xhat(ws, layer, n, S).cpu()
        err = (x_hat - x_hats[layer]).abs().max()
        assert err <= tol * (layer + 1), (layer, float(err))
        rstd = W.saved_rstd(ws, layer, n, S).cpu()
        rel = ((rstd - rstds[layer]).abs() / rstds[layer]).max()
        assert rel <= 2e-5 * (layer + 1), (layer, float(rel))


def test_philox_draws_are_independent_across_launches(dev):
    """In-kernel draws of consecutive launches (and of launches with the rank folded into the
    offset) share nothing: no equality, no shifted equality (the counter of launch k + 1 used to be
    launch k's advanced by one block: u[k+1][s] == u[k][s+4])."""
    from nerf_amd import NeRF
    model = NeRF().to(dev)
    model.rng = "philox"
    n, S = 64, 64
    o = torch.zeros(n, 3, device=dev)
    d = torch.tensor([[1.0, 0.0, 0.0]], device=dev).repeat(n, 1)
    table = O.sample_t(golden_params(), 1, S)[0]
    mid = 0.5 * (table[1:] + table[:-1])
    lower, upper = torch.cat([table[:1], mid]), torch.cat([mid, table[-1:]])

    def draws(state=None):
        t = model.fenceposts_used(o, d, S, randomly_sample=True, rng_state=state).cpu()
        return ((t - lower) / (upper - lower).clamp(min=1e-12))[:, 1:-1]      # u of the interior posts

    a, b = draws(), draws()
    assert 0.0 <= float(a.min()) and float(a.max()) <= 1.0 + 1e-4
    assert abs(float(a.mean()) - 0.5) < 0.02 and abs(float(b.mean()) - 0.5) < 0.02
    for shift in range(0, 9):
        same = (a[:, shift:] - b[:, :b.shape[1] - shift]).abs() < 1e-4
        assert float(same.float().mean()) < 0.01, shift
        same = (b[:, shift:] - a[:, :a.shape[1] - shift]).abs() < 1e-4
        assert float(same.float().mean()) < 0.01, shift
    # rows (rays) of one launch differ, and an explicit state reproduces exactly
    assert float(((a[1:] - a[:-1]).abs() < 1e-4).float().mean()) < 0.01
    seed = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
    r0, r0b, r1 = draws((seed, 5)), draws((seed, 5)), draws((seed, (1 << 40) | 5))
    assert torch.equal(r0, r0b)
    assert float(((r0 - r1).abs() < 1e-4).float().mean()) < 0.01


def test_backward_of_an_empty_batch_is_zero(dev):
    """An empty data-parallel shard: forward returns empty outputs, backward zero gradients."""
    from nerf_amd import NeRF
    model = NeRF().to(dev)
    pixels, seg = model.render_rays(torch.zeros(0, 3, device=dev), torch.zeros(0, 3, device=dev), 16,
                                    randomly_sample=True, density_noise_std=1.0)
    assert pixels.shape == (0, 1, 3) and seg.shape == (0, 1, 50)
    (pixels ** 2).sum().backward()
    for p in model.parameters():
        assert p.grad is not None and torch.count_nonzero(p.grad) == 0
