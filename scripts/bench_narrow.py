"""The headline 800x800x128 frame at hidden_size 256 / 128 / 64 (inference, both arithmetics): kernel time, TFLOP/s on each network's
own FLOP count and fraction of the fp32 MFMA peak.  python scripts/bench_narrow.py [hidden_size encoding_size fp32|f16x3]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nerf_amd import NeRF, _lib
dev = torch.device("cuda:0")
cam_o, cam_r = bench.look_at(bench.CAMERA)
cam_o, cam_r = cam_o.to(dev), cam_r.to(dev)
configs = [(h, e, p) for p in ("fp32", "f16x3") for h, e in ((256, 32), (128, 32), (96, 32), (64, 32), (64, 16), (40, 10))]
if len(sys.argv) > 3:                      # one configuration (under the profiler): hidden_size encoding_size precision
    configs = [(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3])]
for hidden, enc, prec in configs:
    torch.manual_seed(0)
    m = NeRF(focal_length=bench.FOCAL, hidden_size=hidden, encoding_size=enc).to(dev)
    m.precision = prec
    flop = 2 * (3 * enc * hidden + 4 * hidden * hidden + 54 * hidden)
    with torch.no_grad():
        for _ in range(2):
            m.render_image(cam_o, cam_r, 800, 800, bench.FOCAL, 128)
        torch.cuda.synchronize()
        _lib.timing(True); _lib.timing_read(reset=True)
        for _ in range(5):
            m.render_image(cam_o, cam_r, 800, 800, bench.FOCAL, 128)
        torch.cuda.synchronize()
        ms, n = _lib.timing_read(reset=True); _lib.timing(False)
    tf = 640000 * 127 * flop / (ms * 1e-3) / 1e12
    print(f"{prec:5s} hidden {hidden:3d} enc {enc:2d}: kernel {ms:8.2f} ms  {640000 * 128 / (ms * 1e-3):.3e} ray-samples/s  "
          f"{tf:6.1f} TFLOP/s on {flop} FLOP/sample = {tf / (157.3 if prec == 'fp32' else 2516.6):.3f} of the {'fp32' if prec == 'fp32' else 'f16 (x3 executed)'} MFMA peak", flush=True)
