"""Which part of a training step survives HIP-graph capture (prototype; run on the GPU box)."""
import faulthandler, os, sys, torch
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_amd import NeRF
dev = torch.device("cuda:0")
which = sys.argv[1]
n, S = 512, 64
torch.manual_seed(0)
model = NeRF().to(dev)
model.train_precision = "f16x3"
o = torch.randn(n, 3, device=dev); d = torch.randn(n, 3, device=dev); tgt = torch.rand(n, 3, device=dev)
u = torch.rand(n, S, device=dev); noise = torch.randn(n, S - 1, 1, device=dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True, capturable=True)

def body():
    if which == "infer":
        with torch.no_grad():
            return model.render_rays(o, d, S, randomly_sample=True, density_noise_std=1.0, u=u, noise=noise)[0].sum()
    rgb, _ = model.render_rays(o, d, S, randomly_sample=True, density_noise_std=1.0, u=u, noise=noise)
    loss = ((rgb - tgt.unsqueeze(1)) ** 2).mean()
    if which == "fwd":
        return loss
    loss.backward()
    if which == "opt":
        opt.step()
    return loss

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        body()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print(which, "warm-up done", flush=True)
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    out = body()
torch.cuda.synchronize()
print(which, "captured", flush=True)
g.replay(); torch.cuda.synchronize()
print(which, "replayed", float(out), flush=True)
