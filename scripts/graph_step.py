"""Prototype / timing: one training step (forward + loss + backward [+ fused Adam]) captured in a HIP graph.
usage: python scripts/graph_step.py [rays] [fp32|f16x3]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_amd import NeRF
from nerf_amd.optim import Adam
from nerf_amd.loss import mse_and_grad
dev = torch.device("cuda:0")
n, S = int(sys.argv[1]) if len(sys.argv) > 1 else 512, 64
prec = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
torch.manual_seed(0)
model = NeRF().to(dev)
model.train_precision = prec
opt = Adam(model.parameters(), lr=1e-4)
o = torch.randn(n, 3, device=dev); d = torch.randn(n, 3, device=dev); tgt = torch.rand(n, 3, device=dev)
u = torch.rand(n, S, device=dev); noise = torch.randn(n, S - 1, 1, device=dev)

def step():
    rgb, _ = model.render_rays(o, d, S, randomly_sample=True, density_noise_std=1.0, u=u, noise=noise)
    loss, grad = mse_and_grad(rgb, tgt)
    rgb.backward(grad)
    opt.step()
    return loss

# eager timing first (default stream), then the side-stream warm-up capture needs, then the capture
for _ in range(3):
    opt.zero_grad(set_to_none=True); step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    opt.zero_grad(set_to_none=True); l = step()
torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 20
del l
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    static_loss = step()
torch.cuda.synchronize()
before = [p.detach().clone() for p in model.parameters()]
g.replay(); torch.cuda.synchronize()
changed = any(not torch.equal(a, p.detach()) for a, p in zip(before, model.parameters()))
t0 = time.perf_counter()
for _ in range(50):
    g.replay()
torch.cuda.synchronize(); graphed = (time.perf_counter() - t0) / 50
print(f"{n} rays x {S} [{prec}]: eager {eager*1e3:.3f} ms/step, graph replay {graphed*1e3:.3f} ms/step, "
      f"parameters move on replay: {changed}, loss {float(static_loss.detach()):.4f}")
