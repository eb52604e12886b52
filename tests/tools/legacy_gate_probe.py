"""What the gate-aware gradient comparison of tests/test_legacy_backward.py measures, case by case (GPU box):
HIP gradients against the oracle's autograd (a) as it is, (b) with the ReLU gates taken from the kernel's saved
workspace, in fp32 and in fp64, and how many gates differ between the kernel and the oracle."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import legacy_oracle as L            # noqa: E402
import test_legacy_backward as T                   # noqa: E402
import workspace_mirror as W                       # noqa: E402

CFG = L.default_config()
CASES = [("random", 5, 9), ("random", 64, 33), ("random", 130, 64), ("checkpoint", 48, 40), ("random", 3, 100),
         ("checkpoint", 48, 64), ("checkpoint", 31, 17), ("init", 64, 48)]


def main():
    dev = torch.device("cuda:0")
    for precision in ("fp32", "f16x3"):
        T._TRAIN_PRECISION = precision
        for weights, n_rays, S in CASES:
            if weights == "checkpoint":
                params, g = T.checkpoint()
                o, d, near, far = g["rays_o"][:n_rays], g["rays_d"][:n_rays], 2.0, 6.0
            else:
                params = T.random_params(n_rays) if weights == "random" else L.init_params(seed=n_rays)
                gen = torch.Generator().manual_seed(100 + n_rays)
                o, d = torch.randn(n_rays, 3, generator=gen), torch.randn(n_rays, 3, generator=gen)
                near, far = 0.5, 5.0
            gen = torch.Generator().manual_seed(7 + n_rays)
            u = torch.rand(n_rays, S, generator=gen)
            noise = torch.randn(n_rays, S, 1, generator=gen)
            w_rgb = torch.randn(n_rays, 3, generator=gen)
            model = T.make_model(dev, params)
            model.keep_workspace = True
            rgb = model.render_rays(o.to(dev), d.to(dev), near, far, S, randomly_sample=True, density_noise_std=0.5,
                                    u=u.to(dev), noise=noise[..., 0].to(dev))
            (rgb * w_rgb.to(dev)).sum().backward()
            gates = [gt.cpu() for gt in W.legacy_saved_gates(model.last_workspace, n_rays, S)]

            def loss_of(p, cast, gates=None, record=None):
                out = L.render_rays(p, CFG, cast(o), cast(d), near, far, S, u=cast(u), noise=cast(noise),
                                    density_noise_std=0.5, gates=gates, record=record)
                return (out * cast(w_rgb)).sum()

            own = []
            _, plain = T.oracle_gradients(params, lambda p: loss_of(p, lambda t: t, record=own), torch.float32)
            _, exact = T.oracle_gradients(params, lambda p: loss_of(p, lambda t: t.double()), torch.float64)
            _, gated = T.oracle_gradients(params, lambda p: loss_of(p, lambda t: t, gates), torch.float32)
            _, gated64 = T.oracle_gradients(params, lambda p: loss_of(p, lambda t: t.double(), gates), torch.float64)
            flips = sum(int((a != b).sum()) for a, b in zip(own, gates))
            total = sum(a.numel() for a in gates)
            got = {k: p.grad.cpu() for k, p in model.named_parameters()}
            e = lambda ref: max(T.rel_err(got[k], ref[k]) for k in got)          # noqa: E731
            floor = max(T.rel_err(plain[k], exact[k]) for k in got)
            gfloor = max(T.rel_err(gated[k], gated64[k]) for k in got)
            worst = max(got, key=lambda k: T.rel_err(got[k], gated64[k]))
            print(f"[{precision} {weights} {n_rays}x{S}] vs plain fp32 {e(plain):.2e} (old floor {floor:.2e}) | "
                  f"vs gated fp32 {e(gated):.2e}  vs gated fp64 {e(gated64):.2e}  (gated fp32-vs-fp64 {gfloor:.2e}) | "
                  f"gates differing {flips}/{total} = {flips / total:.2e}; worst tensor {worst}", flush=True)


if __name__ == "__main__":
    main()
