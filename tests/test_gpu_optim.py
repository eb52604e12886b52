"""nerf_amd.optim.Adam (one HIP launch over all parameter tensors) against torch.optim.Adam — the optimiser the
reference's loops construct (train_conditional_nerf.py:106-107, :135; examples/example.ipynb cells 7, 8): same
trajectory over many steps on both networks' parameter sets, under HIP-graph replay (the kernel itself counts the
steps), and the options it does not implement are refused.  Also here: nerf_amd.loss.mse, the loops' loss and its
gradient in one launch."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def param_sets():
    from nerf_amd import NeRF
    from nerf_amd.legacy import LegacyNeRF8x256
    torch.manual_seed(0)
    return {"mipnerf": NeRF(), "legacy8x256": LegacyNeRF8x256()}


@pytest.mark.parametrize("which", ["mipnerf", "legacy8x256"])
def test_same_trajectory_as_torch_adam(which):
    from nerf_amd.optim import Adam
    dev = torch.device("cuda:0")
    model = param_sets()[which].to(dev)
    mine = [p.detach().clone().requires_grad_(True) for p in model.parameters()]
    theirs = [p.detach().clone().requires_grad_(True) for p in model.parameters()]
    opt_a, opt_b = Adam(mine, lr=1e-3), torch.optim.Adam(theirs, lr=1e-3)
    gen = torch.Generator(device=dev).manual_seed(1)
    for step in range(50):
        for a, b in zip(mine, theirs):
            g = torch.randn(a.shape, device=dev, generator=gen) * (10.0 ** ((step % 7) - 3))     # 1e-3 .. 1e3
            if step % 5 == 0:
                g = g * (torch.rand(a.shape, device=dev, generator=gen) > 0.5)                     # exact zeros too
            a.grad, b.grad = g.clone(), g.clone()
        opt_a.step()
        opt_b.step()
    for a, b in zip(mine, theirs):
        assert torch.isfinite(a).all()
        # 50 steps of lr 1e-3: parameters moved by up to 5e-2; the two implementations round differently
        # (lerp vs. beta m + (1 - beta) g, pow vs. repeated products): a few ulp of the update per step
        assert (a - b).abs().max() <= 2e-6, float((a - b).abs().max())
    state = opt_a.state_dict()["state"]
    assert float(state["flat"]["step"]) == 50.0 and state["flat"]["exp_avg"].numel() == sum(p.numel() for p in mine)


def test_replays_inside_a_hip_graph():
    from nerf_amd.optim import Adam
    dev = torch.device("cuda:0")
    p = [torch.zeros(300, device=dev, requires_grad=True), torch.zeros(7, 5, device=dev, requires_grad=True)]
    q = [t.detach().clone().requires_grad_(True) for t in p]
    for t in p + q:
        t.grad = torch.ones_like(t)
    opt, ref = Adam(p, lr=1e-2), torch.optim.Adam(q, lr=1e-2)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        opt.step()                                  # state is created outside the capture
    torch.cuda.current_stream(dev).wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        opt.step()
    for _ in range(9):
        graph.replay()
    torch.cuda.synchronize()
    # 10 executed steps: 1 eager + 9 replays (a capture records, it does not run)
    assert float(opt.state_dict()["state"]["flat"]["step"]) == 10.0
    for _ in range(10):
        ref.step()
    for a, b in zip(p, q):
        assert (a - b).abs().max() <= 1e-6


def test_unsupported_options_are_refused():
    from nerf_amd.optim import Adam
    w = torch.zeros(3, device="cuda:0", requires_grad=True)
    with pytest.raises(NotImplementedError):
        Adam([w], weight_decay=0.1)
    with pytest.raises(NotImplementedError):
        Adam([w], amsgrad=True)
    opt = Adam([w])
    with pytest.raises(RuntimeError):
        opt.step()                                  # no gradient


@pytest.mark.parametrize("shape", [(512, 1, 3), (4096, 1, 3), (64, 2, 3), (1000, 3), (1, 1, 3)])
def test_fused_mse_is_the_loops_loss_and_autograds_gradient(shape):
    """nerf_amd.loss.mse against the reference's expression (train_conditional_nerf.py:132) through autograd: the
    loss to summation-order rounding, the gradient BIT FOR BIT ((1 / count) * (2 x), one rounding each)."""
    from nerf_amd.loss import mse
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(3)
    pred = torch.rand(shape, device=dev, generator=gen).requires_grad_(True)
    target = torch.rand(shape[0], 3, device=dev, generator=gen)
    ref_pred = pred.detach().clone().requires_grad_(True)
    want = ((ref_pred - (target.unsqueeze(1) if len(shape) == 3 else target)) ** 2).mean()
    want.backward()
    got = mse(pred, target)
    got.backward()
    assert got.shape == () and abs(float(got.detach()) - float(want.detach())) <= 2e-6 * float(want.detach())
    assert torch.equal(pred.grad, ref_pred.grad)
    # an upstream factor (a weighted sum of losses) scales the gradient like autograd does
    pred.grad = None
    (3.0 * mse(pred, target)).backward()
    assert torch.allclose(pred.grad, 3.0 * ref_pred.grad, rtol=1e-6, atol=0.0)
    # reproducible: one workgroup, a fixed summation order
    assert float(mse(pred, target).detach()) == float(got.detach())


def test_fused_mse_of_an_empty_shard_is_zero():
    from nerf_amd.loss import mse
    dev = torch.device("cuda:0")
    pred = torch.zeros(0, 1, 3, device=dev, requires_grad=True)
    loss = mse(pred, torch.zeros(0, 3, device=dev))
    loss.backward()
    assert float(loss.detach()) == 0.0 and pred.grad.shape == (0, 1, 3)
    with pytest.raises(ValueError):
        mse(torch.zeros(4, 2, device=dev), torch.zeros(4, 3, device=dev))


def test_mse_and_grad_is_the_same_launch_without_autograd():
    from nerf_amd.loss import mse, mse_and_grad
    dev = torch.device("cuda:0")
    pred = torch.rand(300, 1, 3, device=dev).requires_grad_(True)
    target = torch.rand(300, 3, device=dev)
    loss, grad = mse_and_grad(pred, target)
    want = mse(pred, target)
    want.backward()
    assert not loss.requires_grad and float(loss) == float(want.detach()) and torch.equal(grad, pred.grad)
    loss2, grad2 = mse_and_grad(pred[:, 0], target)            # [N, 3] predictions
    assert float(loss2) == float(loss) and torch.equal(grad2, grad[:, 0])


def test_adam_state_round_trips_through_state_dict():
    from nerf_amd.optim import Adam
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    p = [torch.randn(40, 7, device=dev, requires_grad=True), torch.randn(9, device=dev, requires_grad=True)]
    q = [t.detach().clone().requires_grad_(True) for t in p]
    a, b = Adam(p, lr=1e-2), Adam(q, lr=1e-2)
    grads = [[torch.randn_like(t) for t in p] for _ in range(6)]
    for k in range(3):
        for t, u, g in zip(p, q, grads[k]):
            t.grad, u.grad = g.clone(), g.clone()
        a.step(), b.step()
    fresh = Adam(q, lr=1e-2)
    fresh.load_state_dict(b.state_dict())                       # moments and step count move over
    assert float(fresh.state_dict()["state"]["flat"]["step"]) == 3.0
    for k in range(3, 6):
        for t, u, g in zip(p, q, grads[k]):
            t.grad, u.grad = g.clone(), g.clone()
        a.step(), fresh.step()
    for t, u in zip(p, q):
        assert torch.equal(t, u)
