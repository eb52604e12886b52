"""Timings of the other BASELINE.json configurations on one GPU (for DESIGN.md; not the bench line).
usage: python scripts/bench_configs.py [fp32|f16x3]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_amd import NeRF as _NeRF
import bench
PRECISION = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
def NeRF(**kw):
    m = _NeRF(**kw)
    m.precision = PRECISION
    return m
print("precision", PRECISION)
dev = torch.device('cuda:0')
def timeit(fn, n=5, w=2):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
torch.manual_seed(0)
cam_o, cam_r = bench.look_at(bench.CAMERA); cam_o, cam_r = cam_o.to(dev), cam_r.to(dev)
with torch.no_grad():
    m = NeRF(focal_length=112.0).to(dev)
    t = timeit(lambda: m.render_image(cam_o, cam_r, 100, 100, 112.0, 64), n=20, w=3)
    print(f"C2 100x100x64 : {t*1e3:8.2f} ms  {100*100*64/t:.3e} ray-samples/s")
    m = NeRF(focal_length=448.0).to(dev)
    t = timeit(lambda: m.render_image_hierarchical(cam_o, cam_r, 400, 400, 448.0, 64, 128))
    ev = 400*400*(63+191)
    print(f"C3 400x400 64+128 hierarchical: {t*1e3:8.2f} ms  {400*400*192/t:.3e} ray-samples/s  {ev*601088/t/1e12:.1f} TFLOP/s")
    m = NeRF(focal_length=896.0).to(dev)
    t = timeit(lambda: m.render_image(cam_o, cam_r, 800, 800, 896.0, 192))
    print(f"C4 800x800x192: {t*1e3:8.2f} ms  {800*800*192/t:.3e} ray-samples/s  {800*800*191*601088/t/1e12:.1f} TFLOP/s")
    t = timeit(lambda: m.render_image(cam_o, cam_r, 800, 800, 896.0, 192, row_begin=0, row_end=100))
    print(f"C4 shard 100 rows (1/8 frame): {t*1e3:8.2f} ms  -> 8-GPU frame rate if perfectly parallel {800*800*192/t:.3e} ray-samples/s")
