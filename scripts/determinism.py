"""Run-to-run determinism of the forward kernels (GPU box): repeats each mode and counts results
that differ bitwise from the first."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_amd import NeRF
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = NeRF().to(dev)
with torch.no_grad():
    for i in (0, 3, 6, 9, 12, 15):
        m.prediction_heads[i].weight.mul_(2.0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for n, S in ((256, 100), (130, 64), (4096, 64)):
    g = torch.Generator().manual_seed(n)
    o = torch.randn(n, 3, generator=g).to(dev); d = torch.randn(n, 3, generator=g).to(dev)
    u = torch.rand(n, S, generator=g).to(dev); noise = torch.randn(n, S - 1, 1, generator=g).to(dev)
    for mode in ("infer-fp32", "infer-f16x3", "train-fp32"):
        kind, prec = mode.split("-")
        m.precision = prec
        first, bad, worst = None, 0, 0.0
        for r in range(reps):
            if kind == "infer":
                with torch.no_grad():
                    rgb, seg = m.render_rays(o, d, S, randomly_sample=True, density_noise_std=0.5, u=u, noise=noise)
            else:
                rgb, seg = m.render_rays(o, d, S, randomly_sample=True, density_noise_std=0.5, u=u, noise=noise)
            cur = torch.cat([rgb.detach().flatten(), seg.detach().flatten()])
            if first is None: first = cur.clone()
            elif not torch.equal(cur, first):
                bad += 1; worst = max(worst, float((cur - first).abs().max()))
        print(f"{n:5d} x {S:3d} {mode:12s}: {bad}/{reps - 1} runs differ, max |diff| {worst:.3g}", flush=True)
# the bench frame
import bench
cam_o, cam_r = bench.look_at(bench.CAMERA)
cam_o, cam_r = cam_o.to(dev), cam_r.to(dev)
m2 = NeRF(focal_length=bench.FOCAL).to(dev)
for prec in ("fp32", "f16x3"):
    m2.precision = prec
    first, bad = None, 0
    n = max(4, reps // 6)
    for r in range(n):
        with torch.no_grad():
            img, seg = m2.render_image(cam_o, cam_r, 800, 800, bench.FOCAL, 128)
        cur = torch.cat([img.flatten(), seg.flatten()])
        if first is None: first = cur.clone()
        elif not torch.equal(cur, first): bad += 1
    print(f"800x800x128 frame {prec}: {bad}/{n - 1} runs differ", flush=True)
