import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from conftest import golden_params
from oracle import nerf_oracle as O
from nerf_amd import NeRF
dev = torch.device('cuda:0')
model = NeRF(); model.load_state_dict(golden_params(3.0)); model = model.to(dev)
s_c, s_f, n = 64, 128, 130
torch.manual_seed(s_c)
t_c = torch.sort(torch.rand(n, s_c) * 60 + 0.1, dim=-1).values
w = torch.rand(n, s_c - 1) ** 4
w[3] = 0.0; w[5, : (s_c - 1) // 2] = 0.0
got = model.resample_fenceposts(t_c.to(dev), w.to(dev), s_f).cpu()
ref = O.resample_fenceposts(t_c, w, s_f)
d = (got-ref).abs()
print("max t err", d.max(), "sorted", bool((got[:,1:]>=got[:,:-1]).all()))
r, c = divmod(int(d.argmax()), got.shape[1])
print("ray", r, "col", c, got[r, c-2:c+3], ref[r, c-2:c+3])
bad = (d > 2e-4).sum(-1)
print("bad per ray", bad[:10], bad.sum())
