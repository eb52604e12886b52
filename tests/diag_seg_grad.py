"""Diagnostic (not collected): where the segmentation part of a training gradient loses accuracy — the case of
tests/test_gpu_output_width.py::test_gradients_vs_oracle_autograd[(50, 128, 32)]: RGB-only and seg-only losses apart,
the forward's seg output against the fp64 oracle, and the last bias gradient (= sum of dL/d out) of both.
    python tests/diag_seg_grad.py [hidden] [precision]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_output_width as T                          # noqa: E402
from oracle import nerf_oracle as O                         # noqa: E402


def main():
    hidden = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
    shape = (50, hidden, 32)
    dev = torch.device("cuda:0")
    cfg, params, model = T.setup(shape, seed=10 + 50)
    model.train_precision = prec
    n, S = 70, 33
    g = torch.Generator().manual_seed(5)
    o, d = torch.randn(n, 3, generator=g), torch.randn(n, 3, generator=g)
    u = torch.rand(n, S, generator=g)
    noise = torch.randn(n, S - 1, 1, generator=g)
    w_rgb = torch.randn(n, 3, generator=g)
    w_seg = torch.randn(n, 50, generator=g) * 0.05

    def oracle(dtype, use_rgb, use_seg):
        p = {k: v.to(dtype).clone().requires_grad_(k.startswith("prediction")) for k, v in params.items()}
        c = lambda t: t.to(dtype)
        rgb, seg, st = O.render_rays(p, cfg, c(o), c(d), S, u=c(u), noise=c(noise), density_noise_std=0.5, return_stages=True)
        loss = (rgb * c(w_rgb)).sum() * use_rgb + (seg * c(w_seg)).sum() * use_seg
        loss.backward()
        return seg.detach().double(), st["weights"].detach().double(), {k: v.grad.double() for k, v in p.items() if v.grad is not None}

    def kernel(use_rgb, use_seg):
        model.zero_grad(set_to_none=True)
        rgb, seg = model.render_rays(o.to(dev), d.to(dev), S, randomly_sample=True, density_noise_std=0.5, u=u.to(dev),
                                     noise=noise.to(dev))
        loss = (rgb[:, 0] * w_rgb.to(dev)).sum() * use_rgb + (seg[:, 0] * w_seg.to(dev)).sum() * use_seg
        loss.backward()
        return seg[:, 0].detach().cpu().double(), {k: p.grad.detach().cpu().double() for k, p in model.named_parameters()}

    rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp(min=1e-300))
    for name, ur, us in (("rgb only", 1.0, 0.0), ("seg only", 0.0, 1.0), ("both", 1.0, 1.0)):
        s64, w64, g64 = oracle(torch.float64, ur, us)
        s32, _, g32 = oracle(torch.float32, ur, us)
        sk, gk = kernel(ur, us)
        print(f"[{name}] worst tensor: kernel vs fp64 {max(rel(gk[k], g64[k]) for k in g64):.2e}, oracle fp32 vs fp64 "
              f"{max(rel(g32[k], g64[k]) for k in g64):.2e}; last bias: kernel {rel(gk['prediction_heads.15.bias'], g64['prediction_heads.15.bias']):.2e} "
              f"oracle fp32 {rel(g32['prediction_heads.15.bias'], g64['prediction_heads.15.bias']):.2e}")
    # the last bias gradient element by element (= column sums of dL/d out), and against the sum of the dL/d(out) rows the
    # compositing backward left in the workspace (what the weight-gradient kernel summed)
    import workspace_mirror as W
    model.keep_workspace = True
    sk, gk = kernel(1.0, 1.0)
    _, _, g64 = oracle(torch.float64, 1.0, 1.0)
    b_k, b_64 = gk["prediction_heads.15.bias"], g64["prediction_heads.15.bias"]
    lay = W.train_layout(n, S, W.train_width(hidden))
    rows = model.last_workspace[lay["dy5"]:lay["dy5"] + lay["mp"] * 64].view(lay["mp"], 64).cpu().double()
    col = rows.sum(0)[:54]
    scale = float(b_64.abs().max())
    print("last bias, (kernel - fp64) / max, per row:", " ".join(f"{float(x):+.1e}" for x in (b_k - b_64) / scale))
    print("last bias, (sum of workspace dOut rows - fp64) / max:", " ".join(f"{float(x):+.1e}" for x in (col - b_64) / scale))
    print("last bias fp64 / max:", " ".join(f"{float(x):+.2f}" for x in b_64 / scale))
    ds_k, ds_32 = (sk - s64), (s32 - s64)
    heavy = s64.exp() > 1e-3
    print(f"forward seg (log-probabilities): kernel - fp64 max {float(ds_k.abs().max()):.2e} (heavy classes {float(ds_k[heavy].abs().max()):.2e}), "
          f"oracle fp32 - fp64 max {float(ds_32.abs().max()):.2e} (heavy {float(ds_32[heavy].abs().max()):.2e})")
    tot = w64.sum(dim=(1, 2))
    worst = ds_k.abs().max(dim=1).values
    order = worst.argsort(descending=True)[:6]
    for r in order.tolist():
        print(f"   ray {r}: total weight {float(tot[r]):.3e}, largest weight {float(w64[r].max()):.3e}, seg error kernel {float(worst[r]):.2e} "
              f"oracle fp32 {float(ds_32[r].abs().max()):.2e}")


if __name__ == "__main__":
    main()
