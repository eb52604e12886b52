import sys, os, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import golden_params
from oracle import nerf_oracle as O
from nerf_amd import NeRF
dev = torch.device("cuda:0")
model = NeRF(); model.load_state_dict(golden_params(3.0)); model = model.to(dev)
for s_c, s_f in [(64, 128), (9, 5), (33, 64), (2, 7), (1000, 1000)]:
    torch.manual_seed(s_c)
    n = 130
    t_c = torch.sort(torch.rand(n, s_c) * 60 + 0.1, dim=-1).values
    w = torch.rand(n, s_c - 1) ** 4
    w[3] = 0.0; w[5, : (s_c - 1) // 2] = 0.0
    for u in (None, torch.sort(torch.rand(n, s_f), dim=-1).values.clamp(max=1 - 1e-6)):
        got = model.resample_fenceposts(t_c.to(dev), w.to(dev), s_f, u=None if u is None else u.to(dev)).cpu()
        ref = O.resample_fenceposts(t_c, w, s_f, u=u)
        err = (got - ref).abs()
        print(s_c, s_f, u is None, "equal frac", float((got == ref).float().mean()), "max err", float(err.max()))
