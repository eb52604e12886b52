"""One training step (forward + backward + Adam, 4096 rays x 64) at hidden_size 256 / 128 / 64, both arithmetics:
fp32 trains a narrow network at 8 register tiles per sample (its own cost), f16x3 zero-padded in the full-width kernels.
python scripts/bench_train_narrow.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_amd import NeRF
from nerf_amd.optim import Adam
from nerf_amd.loss import mse_and_grad
dev = torch.device("cuda:0")
n, S = 4096, 64
for hidden, enc in ((256, 32), (128, 32), (64, 16)):
    for prec in ("fp32", "f16x3"):
        torch.manual_seed(0)
        model = NeRF(hidden_size=hidden, encoding_size=enc).to(dev)
        model.train_precision = prec
        opt = Adam(model.parameters(), lr=1e-4)
        o, d, tgt = torch.randn(n, 3, device=dev), torch.randn(n, 3, device=dev), torch.rand(n, 3, device=dev)
        def step():
            rgb, _ = model.render_rays(o, d, S, randomly_sample=True, density_noise_std=1.0)
            loss, grad = mse_and_grad(rgb, tgt)
            opt.zero_grad(); rgb.backward(grad); opt.step()
            return loss
        for _ in range(3): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): l = step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        flop = 2 * (3 * enc * hidden + 4 * hidden * hidden + 54 * hidden)
        print(f"hidden {hidden:3d} enc {enc:2d} {prec:5s}: {dt * 1e3:6.3f} ms/step  {n * S / dt:.3e} ray-samples/s  "
              f"{3 * flop * n * (S - 1) / dt / 1e12:6.1f} TFLOP/s (fwd+dgrad+wgrad, own FLOPs)  loss {float(l):.4f}", flush=True)
