"""Timing of one training step (fwd + bwd + Adam) on the GPU: BASELINE config 5 batch shape."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_amd import NeRF, _lib
from nerf_amd.optim import Adam
from nerf_amd.loss import mse_and_grad
dev = torch.device('cuda:0')
n, S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 64
torch.manual_seed(0)
hidden = int(sys.argv[3]) if len(sys.argv) > 3 else 256          # hidden_size / encoding_size of the constructor
enc = int(sys.argv[4]) if len(sys.argv) > 4 else 32
model = NeRF(hidden_size=hidden, encoding_size=enc).to(dev)
model.train_precision = sys.argv[2] if len(sys.argv) > 2 else "fp32"
opt = Adam(model.parameters(), lr=1e-4)                               # as nerf_amd/trainer.py
o = torch.randn(n, 3, device=dev); d = torch.randn(n, 3, device=dev); tgt = torch.rand(n, 3, device=dev)
def step():
    rgb, _ = model.render_rays(o, d, S, randomly_sample=True, density_noise_std=1.0)
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
print(f"train step [{model.train_precision} forward] {n} rays x {S}: {dt*1e3:.2f} ms/step, {n*S/dt:.3e} ray-samples/s, loss {float(l.detach()):.4f}")
print(f"  algorithmic {3*2*(3*enc*hidden+4*hidden*hidden+54*hidden)*n*(S-1)/dt/1e12:.1f} TFLOP/s (fwd+dgrad+wgrad), hidden {hidden} enc {enc}")
