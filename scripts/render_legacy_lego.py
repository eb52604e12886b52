"""Render views of the reference's trained Lego checkpoint (examples/nerf.pth, stored as fixture
tests/golden/g9_legacy_checkpoint.npz) with the fused legacy-network kernel and time it.
python scripts/render_legacy_lego.py [size] [samples]   -> gpurun_out/legacy_lego_<size>.png"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerf_amd import _lib
from nerf_amd.legacy import LegacyNeRF8x256, FLOP_PER_SAMPLE


def look_at(camera_o):
    """Pose looking at the origin, z up, built like the reference's get_rotation_matrix (as bench.py)."""
    from nerf_amd import NeRF
    cam = torch.tensor([list(camera_o)], dtype=torch.float32)
    eye = -cam / torch.linalg.norm(cam, dim=-1, keepdim=True)
    z = torch.tensor([[0.0, 0.0, 1.0]])
    up = z - (z * eye).sum(-1, keepdim=True) * eye
    up = up / torch.linalg.norm(up, dim=-1, keepdim=True)
    return NeRF.get_rotation_matrix(eye, up)


size = int(sys.argv[1]) if len(sys.argv) > 1 else 400
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda", 0)
with np.load(os.path.join(ROOT, "tests", "golden", "g9_legacy_checkpoint.npz")) as z:
    params = {k[6:]: torch.from_numpy(np.array(z[k])) for k in z.files if k.startswith("param.")}
model = LegacyNeRF8x256()
model.load_state_dict(params)
model = model.to(dev)
views = []
for ang in (0.0, 2.1, 4.2):
    cam = torch.tensor([3.5 * np.sin(ang), -3.5 * np.cos(ang), 2.0], dtype=torch.float32)
    cam = cam / cam.norm() * 4.03
    views.append((cam[None], look_at(cam.tolist())))
cam_o = torch.cat([v[0] for v in views]).to(dev)
cam_r = torch.cat([v[1] for v in views]).to(dev)
focal = 138.88887889922103 * size / 100.0
with torch.no_grad():
    img = model.render_image(cam_o, cam_r, size, size, focal, 2.0, 6.0, S)          # warm-up + result
    torch.cuda.synchronize()
    _lib.timing(True); _lib.timing_read(reset=True)
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        model.render_image(cam_o[:1], cam_r[:1], size, size, focal, 2.0, 6.0, S)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    kernel_ms, launches = _lib.timing_read(reset=True); _lib.timing(False)
flop = size * size * S * FLOP_PER_SAMPLE
print(f"legacy 8x256 network, {size}x{size}x{S}: {dt * 1e3:.2f} ms/frame wall, kernel {kernel_ms:.2f} ms "
      f"({launches} launches) = {size * size * S / (kernel_ms * 1e-3):.3e} ray-samples/s, "
      f"{flop / (kernel_ms * 1e-3) / 1e12:.1f} TFLOP/s = {flop / (kernel_ms * 1e-3) / 1e12 / 157.3:.3f} of the fp32 MFMA peak")
from PIL import Image
strip = np.concatenate([(img[i].clamp(0, 1).cpu().numpy() * 255).astype(np.uint8) for i in range(img.shape[0])], axis=1)
out = os.path.join(ROOT, "gpurun_out", f"legacy_lego_{size}.png")
os.makedirs(os.path.dirname(out), exist_ok=True)
Image.fromarray(strip).save(out)
print("saved", out, "mean", float(img.mean()))
