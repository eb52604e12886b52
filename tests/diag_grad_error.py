"""Diagnostic (not collected by pytest): per-tensor ABSOLUTE error of the training gradient of both arithmetics against
the fp64 oracle on one 256-ray batch of the parity study's scene — the first Adam step turns a gradient element into
lr * g / (|g| + 1e-8), so it is the absolute error at the 1e-8 scale, not the error relative to a tensor's largest
element, that decides how far the second step's loss lands from the oracle's (tests/psnr_parity.py).
    python tests/diag_grad_error.py [scene dir]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import psnr_parity as P                                     # noqa: E402
from oracle import nerf_oracle as O                         # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r6_psnr")
    if not os.path.exists(os.path.join(out, "scene.npz")):
        P.make_scene(out)
    from nerf_amd import NeRF
    dev = torch.device("cuda:0")
    images, poses, focal, params0 = P.load_scene(out)
    cfg = dict(O.default_config(), focal_length=focal)
    rays_o, rays_d, pixels = P.host_batches(images, poses, focal)
    gen = torch.Generator().manual_seed(5)
    idx, u, noise = P.captured_step(gen, rays_o.shape[0])

    def oracle(dtype):
        p = {k: v.to(dtype).clone().requires_grad_(k.startswith("prediction")) for k, v in params0.items()}
        loss = O.training_loss(p, cfg, rays_o[idx].to(dtype), rays_d[idx].to(dtype), P.SAMPLES, pixels[idx].to(dtype),
                               u.to(dtype), noise.to(dtype), P.NOISE_STD)
        loss.backward()
        return float(loss), {k: v.grad.double() for k, v in p.items() if v.grad is not None}

    l64, g64 = oracle(torch.float64)
    l32, g32 = oracle(torch.float32)
    runs = {"oracle fp32": g32}
    for prec in ("fp32", "f16x3"):
        model = NeRF(focal_length=focal)
        model.load_state_dict(params0)
        model = model.to(dev)
        model.train_precision = prec
        pix, _ = model.render_rays(rays_o[idx].to(dev), rays_d[idx].to(dev), P.SAMPLES, randomly_sample=True,
                                   density_noise_std=P.NOISE_STD, u=u.to(dev), noise=noise.to(dev))
        loss = ((pix - pixels[idx].to(dev).unsqueeze(1)) ** 2).mean()
        loss.backward()
        runs["hip " + prec] = {k: p.grad.detach().cpu().double() for k, p in model.named_parameters()}
        print(f"hip {prec}: loss {float(loss):.9f} (oracle fp64 {l64:.9f}, fp32 {l32:.9f})")
    print(f"{'tensor':32s} {'max|g|':>9s} " + " ".join(f"{n + ' max|e|':>22s} {'rms e':>9s} {'adam flips':>10s}" for n in runs))
    for k in g64:
        ref = g64[k]
        row = f"{k:32s} {float(ref.abs().max()):9.2e} "
        for n, g in runs.items():
            e = g[k] - ref
            upd_ref = ref / (ref.abs() + 1e-8)
            upd = g[k] / (g[k].abs() + 1e-8)
            moved = float(((upd - upd_ref).abs() > 0.1).double().mean())      # share of elements whose first Adam update moves by > 10 %
            row += f"{float(e.abs().max()):22.2e} {float(e.pow(2).mean().sqrt()):9.2e} {moved:10.2e} "
        print(row)


if __name__ == "__main__":
    main()
