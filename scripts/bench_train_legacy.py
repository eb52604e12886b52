"""Timing of one training step (forward + backward + Adam) of the LEGACY 8 x 256 network on the GPU
(the notebook's loop, examples/example.ipynb cell 8): python scripts/bench_train_legacy.py [rays] [samples] [fp32|f16x3]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_amd.legacy import LegacyNeRF8x256, FLOP_PER_SAMPLE
from nerf_amd.optim import Adam
from nerf_amd.loss import mse_and_grad
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
torch.manual_seed(0)
model = LegacyNeRF8x256().to(dev)
model.train_precision = sys.argv[3] if len(sys.argv) > 3 else "fp32"
opt = Adam(model.parameters(), lr=1e-4)
o = torch.randn(n, 3, device=dev) * 0.5; d = torch.randn(n, 3, device=dev); tgt = torch.rand(n, 3, device=dev)
def step():
    rgb = model.render_rays(o, d, 2.0, 6.0, S, randomly_sample=True, density_noise_std=1.0)
    loss, grad = mse_and_grad(rgb, tgt)
    opt.zero_grad(); rgb.backward(grad); opt.step()
    return loss
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 10
for _ in range(K): l = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(f"legacy train step [{model.train_precision}] {n} rays x {S}: {dt*1e3:.2f} ms/step, {n*S/dt:.3e} ray-samples/s, loss {float(l.detach()):.4f}")
print(f"  algorithmic {3*FLOP_PER_SAMPLE*n*S/dt/1e12:.1f} TFLOP/s (fwd+dgrad+wgrad)")
