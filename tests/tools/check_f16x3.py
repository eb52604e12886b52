"""f16x3 vs fp32 MLP precision: error against the oracle and frame time (GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nerf_amd import NeRF, _lib
from oracle import nerf_oracle as O

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = NeRF().to(dev)
with torch.no_grad():
    for i in (0, 3, 6, 9, 12, 15):
        model.prediction_heads[i].weight.mul_(3.0)       # the "x3" regime of the fixtures
params = {k: v.detach().cpu() for k, v in model.state_dict().items()}
cfg = O.default_config()
g = torch.Generator().manual_seed(1)
n, S = 512, 64
o = torch.randn(n, 3, generator=g) * 2
d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
ref_rgb, ref_seg = O.render_rays(params, cfg, o, d, S)
for prec in ("fp32", "f16x3"):
    model.precision = prec
    with torch.no_grad():
        rgb, seg = model.render_rays(o.to(dev), d.to(dev), S)
    rgb, seg = rgb.cpu().reshape(ref_rgb.shape), seg.cpu().reshape(ref_seg.shape)
    e = (rgb - ref_rgb).abs().flatten()
    print(prec, "rgb max err", float(e.max()), "q99", float(e.quantile(0.99)), "median", float(e.median()),
          "seg max err", float((seg - ref_seg).abs().max()), "finite", bool(torch.isfinite(rgb).all()))
# frame time
import bench
cam_o, cam_r = bench.look_at(bench.CAMERA)
cam_o, cam_r = cam_o.to(dev), cam_r.to(dev)
m2 = NeRF(focal_length=bench.FOCAL).to(dev)
outs = {}
for prec in ("fp32", "f16x3"):
    m2.precision = prec
    with torch.no_grad():
        for _ in range(2):
            out = m2.render_image(cam_o, cam_r, 800, 800, bench.FOCAL, 128)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            out = m2.render_image(cam_o, cam_r, 800, 800, bench.FOCAL, 128)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    outs[prec] = out
    print(prec, f"{dt*1e3:.1f} ms/frame  {800*800*128/dt:.3e} samples/s")
print("frame rgb max diff f16x3 vs fp32", float((outs["fp32"][0] - outs["f16x3"][0]).abs().max()),
      "seg", float((outs["fp32"][1] - outs["f16x3"][1]).abs().max()))
