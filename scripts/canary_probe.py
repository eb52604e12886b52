"""Diagnostic for builds with -DNERF_EXP_CANARY: runs the training forward directly (no autograd) with
a debug buffer in place of out_t and prints which loop-invariant canary registers changed, where."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_amd import NeRF, _lib
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = NeRF().to(dev)
m.train_precision = prec
print("flags", _lib.build_flags(), flush=True)
n, S = 4096, 64
g = torch.Generator().manual_seed(n)
o = torch.randn(n, 3, generator=g).to(dev); d = torch.randn(n, 3, generator=g).to(dev)
u = torch.rand(n, S, generator=g).to(dev); noise = torch.randn(n, S - 1, generator=g).to(dev)
ws = torch.empty(_lib.lib().nerf_hip_train_workspace_bytes(n, S) // 4, dtype=torch.float32, device=dev)
for rep in range(5):
    dbg = torch.zeros(8 + 8 * 4096, dtype=torch.int32, device=dev)
    m._launch(n, S, dev, rays_o=o, rays_d=d, u=u, noise=noise, density_noise_std=0.5, train_workspace=ws,
              out_t=dbg.view(torch.float32))
    torch.cuda.synchronize()
    h = dbg.cpu()
    k = int(h[0])
    rec = h[8:8 + 8 * min(k, 4096)].view(-1, 8)
    print(f"rep {rep}: {k} canary mismatches", flush=True)
    if k:
        lanes = collections.Counter((int(r[2]) // 16) for r in rec)
        regs = collections.Counter(int(r[3]) for r in rec)
        items = collections.Counter(int(r[4]) for r in rec)
        print("   by lane group:", dict(lanes), " by canary index:", dict(regs), " by item:", dict(sorted(items.items())))
        for r in rec[:16].tolist():
            print("   block %d wave %d lane %d canary %d item %d: got 0x%08x want 0x%08x" % (
                r[0], r[1], r[2], r[3], r[4], r[5] & 0xFFFFFFFF, r[6] & 0xFFFFFFFF))
