"""The notebook's training run (examples/example.ipynb cells 6-8: 8 x 256 network, tiny_nerf Lego 100 x 100, 64
samples per ray, 1,024-ray batches, Adam lr 1e-4, density_noise_std 1.0, near / far 2 / 6, hold out one view,
PSNR every 1,000 iterations) on the GPU, start to finish.  tiny_nerf_data.npz is not in the reference's
repository and there is no network here, so the scene is a STAND-IN: 106 views of 100 x 100 pixels (tiny_nerf's
count, size and focal length) rendered from the reference's own trained Lego weights (examples/nerf.pth =
fixture tests/golden/g9_legacy_checkpoint.npz) on the upper hemisphere at the checkpoint's camera radius.
A fresh LegacyNeRF8x256 is then trained on them with nerf_amd.trainer.Trainer exactly as the notebook does.
BASELINE.md's curve for the real data set: ~8 dB at 0, ~23 dB at 1,000, ~32 dB at 40,000 iterations.

usage: python scripts/train_lego_standin.py [iterations] [fp32|f16x3] [graph|eager] [legacy8x256|mipnerf|mipnerf128]
(mipnerf: the generation-C network of nerf/model.py on the same views — its own log-spaced samples, no near / far;
 mipnerf128: the same with hidden_size=128, i.e. through the kernels instantiated at 8 register tiles per sample)
writes gpurun_out/lego_standin_<arith>_<graph|eager>_<iterations>.json (iterations, PSNR, seconds) and the same
name .png (held-out view: truth | render)."""
import json, math, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerf_amd.legacy import LegacyNeRF8x256
from nerf_amd.trainer import Trainer

iterations = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
arith = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
graph = (sys.argv[3] if len(sys.argv) > 3 else "graph") == "graph"
network = sys.argv[4] if len(sys.argv) > 4 else "legacy8x256"
dev = torch.device("cuda", 0)
H = W = 100
FOCAL = 138.88887889922103            # tiny_nerf's focal length in pixels (the notebook's data file)
VIEWS, RADIUS = 106, 4.03


def look_at(cam):
    """[R | t] of a camera at `cam` looking at the origin, z up (as bench.py / render_legacy_lego.py)."""
    from nerf_amd import NeRF
    cam = torch.tensor([list(cam)], dtype=torch.float32)
    eye = -cam / torch.linalg.norm(cam, dim=-1, keepdim=True)
    z = torch.tensor([[0.0, 0.0, 1.0]])
    up = z - (z * eye).sum(-1, keepdim=True) * eye
    up = up / torch.linalg.norm(up, dim=-1, keepdim=True)
    pose = torch.eye(4)
    pose[:3, :3] = NeRF.get_rotation_matrix(eye, up)[0]
    pose[:3, 3] = cam[0]
    return pose


# ---- the stand-in scene: views of the reference's trained weights -----------------------------------
with np.load(os.path.join(ROOT, "tests", "golden", "g9_legacy_checkpoint.npz")) as z:
    params = {k[6:]: torch.from_numpy(np.array(z[k])) for k in z.files if k.startswith("param.")}
teacher = LegacyNeRF8x256()
teacher.load_state_dict(params)
teacher = teacher.to(dev)
poses = []
for v in range(VIEWS):                                    # golden-angle spiral over elevations 10 .. 60 degrees
    az = v * math.pi * (3.0 - math.sqrt(5.0))
    el = math.radians(10.0 + 50.0 * ((v * 0.6180339887) % 1.0))
    poses.append(look_at((RADIUS * math.cos(el) * math.sin(az), -RADIUS * math.cos(el) * math.cos(az),
                          RADIUS * math.sin(el))))
poses = torch.stack(poses).to(dev)
with torch.no_grad():
    images = torch.cat([teacher.render_image(poses[v:v + 1, :3, 3].contiguous(), poses[v:v + 1, :3, :3].contiguous(),
                                             H, W, FOCAL, 2.0, 6.0, 128) for v in range(VIEWS)]).clamp(0.0, 1.0)
torch.cuda.synchronize()
print(f"stand-in scene: {VIEWS} views {H}x{W} rendered from examples/nerf.pth, mean {float(images.mean()):.3f}")

# ---- the notebook's run ------------------------------------------------------------------------------
torch.manual_seed(0)
if network.startswith("mipnerf"):
    from nerf_amd import NeRF
    student = NeRF(focal_length=FOCAL, hidden_size=int(network[7:] or 256)).to(dev)
else:
    student = LegacyNeRF8x256().to(dev)                   # fresh PyTorch-default initialisation
student.train_precision = arith
trainer = Trainer(images, poses, FOCAL, batch_size=1024, learning_rate=1e-4, num_samples_per_ray=64,
                  density_noise_std=1.0, log_interval=1000, model=student, rng="torch", graph=graph,
                  near=2.0, far=6.0)
curve = []
torch.cuda.synchronize()
train_seconds = 0.0
done = 0
rays_per_epoch = (VIEWS - 1) * H * W
epochs = iterations * 1024 // rays_per_epoch + 2
t_last = time.perf_counter()
for _ in range(epochs):
    for batch in trainer.dataset.batches(trainer.batch_size, shuffle=True, generator=trainer.sampler):
        trainer.iteration += 1
        trainer.train_step(batch)
        done += 1
        if trainer.iteration % 1000 == 0 or done == iterations:
            torch.cuda.synchronize()
            train_seconds += time.perf_counter() - t_last
            value = trainer.evaluate()
            curve.append({"iteration": trainer.iteration, "psnr_db": value, "train_seconds": train_seconds})
            print(f"iteration {trainer.iteration:6d}: held-out PSNR {value:6.2f} dB after {train_seconds:7.2f} s of training")
            t_last = time.perf_counter()
        if done == iterations:
            break
    if done == iterations:
        break
out = os.path.join(ROOT, "gpurun_out")
os.makedirs(out, exist_ok=True)
result = {"what": "examples/example.ipynb cell 8 on a stand-in Lego scene rendered from examples/nerf.pth",
          "network": network, "train_precision": arith, "graph_replay": graph, "batch_rays": 1024,
          "samples_per_ray": 64, "learning_rate": 1e-4, "iterations": done, "train_seconds": train_seconds,
          "ms_per_iteration": 1e3 * train_seconds / max(done, 1), "curve": curve}
tag = f"{arith}_{'graph' if graph else 'eager'}_{done}" + ("" if network == "legacy8x256" else "_" + network)
with open(os.path.join(out, f"lego_standin_{tag}.json"), "w") as f:
    json.dump(result, f, indent=1)
from PIL import Image
strip = np.concatenate([(trainer.truth[-1][0].clip(0, 1) * 255).astype(np.uint8),
                        (trainer.rendered[-1][0].clip(0, 1) * 255).astype(np.uint8)], axis=1)
Image.fromarray(strip).save(os.path.join(out, f"lego_standin_{tag}.png"))
print(json.dumps({k: v for k, v in result.items() if k != "curve"}))
