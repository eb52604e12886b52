"""Run-to-run determinism of the forward launches (no atomics anywhere in the path, so results must
be bitwise reproducible): both arithmetics of the inference kernel and the training forward, at a
batch that puts two workgroups on every CU (where a data race between a stage's LDS-DMA and its
consumers would show) and at a small one."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_rays,num_samples", [(256, 100), (4096, 64)])
@pytest.mark.parametrize("mode", ["infer-fp32", "infer-f16x3", "train-fp32", "train-f16x3"])
def test_forward_is_bitwise_reproducible(mode, n_rays, num_samples):
    from nerf_amd import NeRF
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = NeRF().to(dev)
    with torch.no_grad():
        for i in (0, 3, 6, 9, 12, 15):
            model.prediction_heads[i].weight.mul_(2.0)
    g = torch.Generator().manual_seed(n_rays)
    o, d = torch.randn(n_rays, 3, generator=g).to(dev), torch.randn(n_rays, 3, generator=g).to(dev)
    u = torch.rand(n_rays, num_samples, generator=g).to(dev)
    noise = torch.randn(n_rays, num_samples - 1, 1, generator=g).to(dev)
    kind, model.precision = mode.split("-")
    model.train_precision = model.precision
    first = None
    for _ in range(12):
        with torch.set_grad_enabled(kind == "train"):
            rgb, seg = model.render_rays(o, d, num_samples, randomly_sample=True, density_noise_std=0.5,
                                         u=u, noise=noise)
        cur = torch.cat([rgb.detach().flatten(), seg.detach().flatten()])
        if first is None:
            first = cur.clone()
        else:
            assert torch.equal(cur, first), float((cur - first).abs().max())


@pytest.mark.parametrize("train_precision", ["fp32", "f16x3"])
def test_training_step_gradients_are_bitwise_reproducible(train_precision):
    """Forward + backward of a 4096 x 64 batch (BASELINE config 5 shape), five times: loss and the
    flat 304,438-element gradient must not move by a bit."""
    from nerf_amd import NeRF
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = NeRF().to(dev)
    model.train_precision = train_precision
    g = torch.Generator().manual_seed(7)
    n, S = 4096, 64
    o, d = torch.randn(n, 3, generator=g).to(dev), torch.randn(n, 3, generator=g).to(dev)
    u = torch.rand(n, S, generator=g).to(dev)
    noise = torch.randn(n, S - 1, 1, generator=g).to(dev)
    target = torch.rand(n, 3, generator=g).to(dev)
    first = None
    for _ in range(5):
        model.zero_grad(set_to_none=True)
        pixels, seg = model.render_rays(o, d, S, randomly_sample=True, density_noise_std=1.0, u=u, noise=noise)
        loss = ((pixels - target.unsqueeze(1)) ** 2).mean() + 1e-3 * (seg ** 2).mean()
        loss.backward()
        cur = torch.cat([loss.detach().reshape(1)] + [p.grad.flatten() for p in model.parameters()])
        if first is None:
            first = cur.clone()
        else:
            assert torch.equal(cur, first), float((cur - first).abs().max())
