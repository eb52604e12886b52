"""Checker (GPU box): error of layer 0's weight / bias gradient (and layer 1's weight gradient) against the fp64 oracle
gradient in both training arithmetics, with the factors of the split-precision scale BOUND pushed to where it is loosest
(tests/test_gpu_backward.py: test_layer0_weight_gradient_under_a_loose_scale_bound asserts the same configurations).
usage: python tests/tools/bound_margin.py"""
import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_gpu_backward as T
dev = torch.device("cuda:0")
O, CFG = T.O, T.CFG
for (w0, g0, w1) in [(1.0, 1.0, 1.0), (1e-3, 1.0, 1.0), (1.0, 30.0, 1.0), (3e-3, 10.0, 4.0)]:
    torch.manual_seed(77)
    params = T.golden_params(2.0)
    params["prediction_heads.0.weight"] = params["prediction_heads.0.weight"] * w0
    params["prediction_heads.1.weight"] = params["prediction_heads.1.weight"] * g0
    params["prediction_heads.3.weight"] = params["prediction_heads.3.weight"] * w1
    n, S = 64, 33
    o, d = torch.randn(n, 3), torch.randn(n, 3); u = torch.rand(n, S); target = torch.rand(n, 3)
    loss_of = lambda rgb, seg, cast: ((rgb - cast(target)) ** 2).sum() + 1e-3 * (seg ** 2).sum()
    exact = T.fp64_gradients(params, lambda p: loss_of(*O.render_rays(p, CFG, o.double(), d.double(), S, u=u.double()), lambda t: t.double()))
    out = {}
    for prec in ("fp32", "f16x3"):
        T._TRAIN_PRECISION = prec
        model = T.make_model(dev, params)
        rgb, seg = model.render_rays(o.to(dev), d.to(dev), S, randomly_sample=True, u=u.to(dev))
        loss_of(rgb[:, 0], seg[:, 0], lambda t: t.to(dev)).backward()
        g = dict(model.named_parameters())
        out[prec] = {k: T.rel_err(g[k].grad.cpu().double(), exact[k]) for k in ("prediction_heads.0.weight", "prediction_heads.0.bias", "prediction_heads.3.weight")}
    print((w0, g0, w1), {p: {k.split("heads.")[1]: f"{v:.2e}" for k, v in out[p].items()} for p in out})
