"""Training entry point with the reference script's flags (train_conditional_nerf.py:20-49) on
the generation-C renderer.  Single GPU: ``python train_nerf_amd.py --data scene.npz``; data
parallel: ``python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1
train_nerf_amd.py ...`` (one process per GPU, RCCL all-reduce of the 1.2 MB gradient per step).
Flag names AND defaults are the reference's.  ``--data`` takes the reference's pickle (a dict with
``images`` [V,H,W,3], ``poses`` [V,6] = position | viewing direction, ``states``; focal length in
pixels = W * camera-focal-length / camera-ccd-width, train_conditional_nerf.py:70-87), an ``.npz`` in
the tiny_nerf layout of the notebook (images, poses [V,4,4], focal), or the word ``synthetic``: views
of a synthetic teacher field rendered by the renderer itself (the reference's data files are not in
its repository)."""
import argparse
import os

import torch
import torch.distributed as dist

from nerf_amd import trainer as T

if __name__ == "__main__":
    ap = argparse.ArgumentParser("Train a NeRF model with the MI355X renderer")
    # the reference's thirteen flags with its defaults (train_conditional_nerf.py:22-47)
    ap.add_argument("--logging-dir", type=str, default="experiment")
    ap.add_argument("--data", type=str, default="examples/data_for_nerf.pkl")
    ap.add_argument("--epochs", type=int, default=100)
    ap.add_argument("--camera-focal-length", type=float, default=50.0)
    ap.add_argument("--camera-ccd-width", type=float, default=36.0)
    ap.add_argument("--batch-size", type=int, default=1024)
    ap.add_argument("--normalize-position", type=float, default=20.0)      # legacy8x256: positions / this
    ap.add_argument("--learning-rate", type=float, default=0.0001)
    ap.add_argument("--near-plane", type=float, default=0.0)               # legacy8x256: sample range
    ap.add_argument("--far-plane", type=float, default=20.0)               # (generation C: implicit, unused)
    ap.add_argument("--num-samples-per-ray", type=int, default=64)
    ap.add_argument("--density-noise-std", type=float, default=1.0)
    ap.add_argument("--log-interval", type=int, default=1000)
    ap.add_argument("--max-iterations", type=int, default=None)
    # mipnerf: the network of nerf/model.py as shipped (generation C).  legacy8x256: the 8 x 256 sin/cos
    # network of examples/nerf.pth and of the notebook (NeRF(normalize_position=6.0), near 2, far 6)
    ap.add_argument("--network", choices=("mipnerf", "legacy8x256"), default="mipnerf")
    ap.add_argument("--rng", choices=("torch", "philox"), default="philox")
    # arithmetic of the MLP in the held-out renders ...
    ap.add_argument("--precision", choices=("fp32", "f16x3"), default="fp32")
    # ... and of the training steps: forward, data and weight gradient
    ap.add_argument("--train-precision", choices=("fp32", "f16x3"), default="fp32")
    # replay each training step as one HIP graph (launch-bound small batches; implies --rng torch)
    ap.add_argument("--graph", action="store_true")
    args = ap.parse_args()
    if args.graph:
        args.rng = "torch"

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=device)

    if args.data == "synthetic":
        images, poses, focal = T.synthetic_scene(device=device)
    elif args.data.endswith(".npz"):
        images, poses, focal = T.load_scene(args.data, device)
    else:
        images, poses, focal = T.load_pickled_scene(args.data, device, args.camera_focal_length,
                                                    args.camera_ccd_width)
    model = None
    if args.network == "legacy8x256":
        from nerf_amd.legacy import LegacyNeRF8x256
        torch.manual_seed(0)
        model = LegacyNeRF8x256(normalize_position=args.normalize_position).to(device)
        args.rng = "torch"
    run = T.Trainer(images, poses, focal, logging_dir=args.logging_dir, batch_size=args.batch_size,
                    learning_rate=args.learning_rate, num_samples_per_ray=args.num_samples_per_ray,
                    density_noise_std=args.density_noise_std, log_interval=args.log_interval,
                    rng=args.rng, graph=args.graph, model=model, near=args.near_plane, far=args.far_plane)
    run.model.precision = args.precision
    run.model.train_precision = args.train_precision
    run.write_params(vars(args))
    run.fit(epochs=args.epochs, max_iterations=args.max_iterations)
    if run.rank == 0 and run.psnrs:
        print(f"iteration {run.iternums[-1]}: held-out PSNR {float(run.psnrs[-1]):.2f} dB")
    if world > 1:
        dist.destroy_process_group()
