"""Diagnostic (GPU box): board power and shader clock (rocm-smi) while one workload loops.
usage: python scripts/power_sample.py infer-fp32|infer-f16x3|train-fp32|train-f16x3 [seconds]"""
import os, subprocess, sys, threading, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_amd import NeRF
what = sys.argv[1]
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = NeRF(focal_length=896.0).to(dev)
kind, prec = what.split("-")
model.precision = model.train_precision = prec
stop = False
samples = []

def sampler():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True).stdout
        samples.append((time.time(), out))
        time.sleep(0.5)

if kind == "infer":
    cam = torch.tensor([[0.0, -3.0, 2.6]])
    eye = -cam / torch.linalg.norm(cam, dim=-1, keepdim=True)
    z = torch.tensor([[0.0, 0.0, 1.0]])
    up = z - (z * eye).sum(-1, keepdim=True) * eye
    cam_o, cam_r = cam.to(dev), NeRF.get_rotation_matrix(eye, up / torch.linalg.norm(up, dim=-1, keepdim=True)).to(dev)
    def work():
        with torch.no_grad():
            model.render_image(cam_o, cam_r, 800, 800, 896.0, 128)
else:
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)
    n, S = 4096, 64
    o, d, tgt = torch.randn(n, 3, device=dev), torch.randn(n, 3, device=dev), torch.rand(n, 3, device=dev)
    def work():
        for _ in range(20):
            opt.zero_grad(set_to_none=True)
            rgb, _ = model.render_rays(o, d, S, randomly_sample=True, density_noise_std=1.0)
            ((rgb - tgt.unsqueeze(1)) ** 2).mean().backward()
            opt.step()
work(); torch.cuda.synchronize()
th = threading.Thread(target=sampler); th.start()
t0 = time.time(); n_it = 0
while time.time() - t0 < seconds:
    work(); n_it += 1
torch.cuda.synchronize()
stop = True; th.join()
import json, re
vals = []
for _, out in samples[2:]:
    try:
        j = json.loads(out)
        card = next(iter(j.values()))
        p = [v for k, v in card.items() if "ower" in k and "W" in k]
        c = [v for k, v in card.items() if "sclk" in k.lower()]
        vals.append((p[:1], c[:1]))
    except Exception as e:
        vals.append((out[:200], str(e)))
print(what, "iterations", n_it, "samples:", vals[:8])
