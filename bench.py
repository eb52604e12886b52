"""Headline benchmark: ray-samples/sec of the fused MI355X renderer on a synthetic
800x800 frame at 128 samples per ray (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = NeRF.render_image of one full frame per GPU (rays generated in-kernel from the pose,
deterministic fenceposts, RGB + 50-class segmentation composited): the whole hot path, inputs
(pose, packed parameters) resident in HBM.  Frames are independent, so ranks shard a batch of
N poses one frame each with no data-path collective ("weak" scaling; ``--scaling strong`` splits
ONE frame into row blocks instead).  Rank 0 prints one JSON line.

Two arithmetics of the same kernel (NerfHipRenderArgs.precision), both held to
the same parity tests: "f16x3" (default here: every fp32 product as three f16 MFMAs with fp32
accumulation, v_mfma_f32_16x16x32_f16) and "fp32" (exact-fp32 v_mfma_f32_16x16x4_f32).  The
headline `value` is the selected precision's; at N=1 the other one is measured in the same run and
reported beside it (`other_precision`).

roofline: the render kernel is MFMA-bound; achieved = evaluated samples per launch x 601,088
ALGORITHMIC FLOP / average kernel duration measured with HIP events on the launch stream; peak =
the dense MFMA peak of the instruction's input type (MI355X_MICROARCH.md): 157.3 TFLOP/s fp32,
2516.6 TFLOP/s f16.  The f16x3 path executes 3 MFMA FLOP per algorithmic FLOP (`executed_frac`).
cpu_baseline: the oracle (a torch-CPU port of the reference, oracle/nerf_oracle.py) timed on
this box's host cores on a bounded block of rows of the same frame.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

IMAGE = 800
SAMPLES = 128
FOCAL = 896.0
FLOP_PER_SAMPLE = 601088          # 2*(96*256 + 4*256*256 + 256*54), SURVEY.md section 8d
PEAK_TFLOPS_FP32_MFMA = 157.3     # MI355X_MICROARCH.md, chip-level parameters
PEAK_TFLOPS_F16_MFMA = 2516.6     # dense f16/bf16: 1024 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz
PRECISIONS = {
    "fp32": {"dtype": "f32", "peak": PEAK_TFLOPS_FP32_MFMA, "mfma_per_product": 1,
             "peak_note": "exact-fp32 MFMA (v_mfma_f32_16x16x4_f32) peak; algorithmic FLOPs only"},
    "f16x3": {"dtype": "f16x3 (f32 accumulate)", "peak": PEAK_TFLOPS_F16_MFMA, "mfma_per_product": 3,
              "peak_note": "dense f16 MFMA (v_mfma_f32_16x16x32_f16) peak; `achieved`/`frac` count "
                           "ALGORITHMIC FLOPs, the kernel executes 3 MFMA FLOP per algorithmic FLOP "
                           "(hi.hi + hi.lo + lo.hi): executed_frac = 3 x frac"},
}
CAMERA = (0.0, -3.0, 2.6)


def look_at(camera_o):
    """Pose looking at the origin, z up, built like the reference's get_rotation_matrix."""
    from nerf_amd import NeRF
    cam = torch.tensor([camera_o], dtype=torch.float32)
    eye = -cam / torch.linalg.norm(cam, dim=-1, keepdim=True)
    z = torch.tensor([[0.0, 0.0, 1.0]])
    up = z - (z * eye).sum(-1, keepdim=True) * eye
    up = up / torch.linalg.norm(up, dim=-1, keepdim=True)
    return cam, NeRF.get_rotation_matrix(eye, up)


def cpu_baseline(rows=32):
    """Oracle render of `rows` image rows of the bench frame on the host cores."""
    from oracle import nerf_oracle as O
    threads = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(threads)
    params = O.init_params(seed=0)
    cfg = dict(O.default_config(), focal_length=FOCAL)
    cam_o, cam_r = look_at(CAMERA)
    rays_o, rays_d = O.image_rays(cam_o, cam_r, IMAGE, IMAGE, FOCAL)
    r0 = (IMAGE - rows) // 2
    sl = slice(r0 * IMAGE, (r0 + rows) * IMAGE)
    o, d = rays_o[sl], rays_d[sl]
    with torch.no_grad():
        O.render_rays(params, cfg, o[:1024], d[:1024], SAMPLES)          # warm-up
        t0 = time.perf_counter()
        for a, b in zip(torch.split(o, 1024), torch.split(d, 1024)):
            O.render_rays(params, cfg, a, b, SAMPLES)
        dt = time.perf_counter() - t0
    return {"value": rows * IMAGE * SAMPLES / dt, "unit": "ray-samples/s", "cores": threads,
            "kind": "port",
            "sample": f"{rows} rows x {IMAGE} px of the same 800x800x128 frame, chunks of 1024 rays, "
                      f"{dt:.1f} s, torch {torch.__version__} CPU ops"}


def train_step_timing(dev, rays=4096, samples=64, steps=10, warmup=3):
    """Secondary figure (BASELINE config 5 batch): one optimiser step = training forward + HIP
    backward + Adam on `rays` x `samples`; random rays/targets, stratified draws, noise std 1."""
    from nerf_amd import NeRF
    torch.manual_seed(0)
    model = NeRF().to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    o, d = torch.randn(rays, 3, device=dev), torch.randn(rays, 3, device=dev)
    target = torch.rand(rays, 3, device=dev)

    def step():
        pixels, _ = model.render_rays(o, d, samples, randomly_sample=True, density_noise_std=1.0)
        loss = ((pixels - target.unsqueeze(1)) ** 2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()

    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    tflops = 3 * FLOP_PER_SAMPLE * rays * (samples - 1) / dt / 1e12
    return {"workload": f"{rays} rays x {samples} samples, forward + backward + Adam",
            "ms_per_step": dt * 1e3, "ray_samples_per_s": rays * samples / dt,
            "tflops_fwd_dgrad_wgrad": tflops,
            "arithmetic": "training forward and data gradient on fp32 MFMA, weight gradient on bf16 "
                          "triples (six bf16 MFMAs per product, fp32 accumulate)"}


def profiled_traffic(precision):
    """HBM bytes per launch of the render kernel from the committed rocprofv3 PMC passes of this
    same command (profiles/*_pmc_summary.json, the newest one of this precision; FETCH_SIZE/WRITE_SIZE
    are KiB, FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md).  PMC counters
    cannot be read from inside the timed process, so this is the last profiled value, or None."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json"))):
        if ("f16x3" in os.path.basename(path)) != (precision == "f16x3"):
            continue
        try:
            with open(path) as f:
                d = json.load(f)
            best = ((2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0, os.path.relpath(path, ROOT))
        except (OSError, KeyError, ValueError):
            continue
    return best


def roofline(precision, rays, kernel_ms, launches, with_traffic):
    evaluated = rays * (SAMPLES - 1)
    info = PRECISIONS[precision]
    achieved = evaluated * FLOP_PER_SAMPLE / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else None
    traffic = profiled_traffic(precision) if with_traffic else None
    return {
        "bound": "mfma",
        "achieved": achieved,
        "peak": info["peak"],
        "unit": "TFLOP/s",
        "frac": achieved / info["peak"] if achieved else None,
        "traffic": traffic[0] if traffic else None,
        "traffic_unit": "bytes/launch",
        "traffic_source": traffic[1] if traffic else None,
        "algorithmic_bytes": rays * (12 + 200) + 48,
        "kernel": "nerf_render_fwd_kernel",
        "kernel_ms": kernel_ms,
        "launches_timed": launches,
        "flop_per_launch": evaluated * FLOP_PER_SAMPLE,
        "executed_mfma_flop_per_launch": evaluated * FLOP_PER_SAMPLE * info["mfma_per_product"],
        "executed_frac": achieved * info["mfma_per_product"] / info["peak"] if achieved else None,
        "peak_note": info["peak_note"],
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", choices=("fp32", "f16x3"), default=os.environ.get("NERF_BENCH_PRECISION", "f16x3"))
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    backend = None
    if distributed:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        try:                                      # RCCL over xGMI; only barriers + one scalar reduce use it
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            backend = "nccl"
        except Exception as exc:                  # keep the measurement alive if RCCL cannot come up
            print(f"[bench] nccl init failed ({exc}); falling back to gloo for the barriers", file=sys.stderr)
            dist.init_process_group("gloo")
            backend = "gloo"
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local_rank)

    from nerf_amd import NeRF, _lib
    torch.manual_seed(0)
    model = NeRF(focal_length=FOCAL).to(dev)         # default init, seed 0, on every rank
    model.precision = args.precision

    # a batch of `world` poses on a circle of the same radius; rank r renders frame r (weak) or
    # its row block of frame 0 (strong)
    import math
    if args.scaling == "weak":
        ang = 2.0 * math.pi * rank / max(world, 1)
        r_xy = math.hypot(CAMERA[0], CAMERA[1])
        pose = (r_xy * math.sin(ang), -r_xy * math.cos(ang), CAMERA[2])
        rows = (0, IMAGE)
    else:
        pose = CAMERA
        per = IMAGE // world
        rows = (rank * per, IMAGE if rank == world - 1 else (rank + 1) * per)
    cam_o, cam_r = look_at(pose)
    cam_o, cam_r = cam_o.to(dev), cam_r.to(dev)

    def step():
        with torch.no_grad():
            return model.render_image(cam_o, cam_r, IMAGE, IMAGE, FOCAL, SAMPLES,
                                      row_begin=rows[0], row_end=rows[1])

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed(steps, warmup):
        for _ in range(warmup):
            step()
        fence()
        _lib.timing(True)
        _lib.timing_read(reset=True)
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step()
        fence()
        dt = time.perf_counter() - t0
        kernel_ms, launches = _lib.timing_read(reset=True)
        _lib.timing(False)
        assert torch.isfinite(out[0]).all()
        return dt, kernel_ms, launches

    elapsed, kernel_ms, launches = timed(args.steps, args.warmup)

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend != "gloo" else "cpu")
    if distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    rays_per_rank = (rows[1] - rows[0]) * IMAGE
    total_rays = rays_per_rank * world if args.scaling == "weak" else IMAGE * IMAGE
    value = total_rays * SAMPLES * args.steps / elapsed

    if rank == 0:
        line = {
            "metric": "ray-samples/sec at 800x800x128",
            "value": value,
            "unit": "ray-samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": PRECISIONS[args.precision]["dtype"],
            "data": "synthetic",
            "config": {
                "workload": "NeRF.render_image 800x800, 128 samples/ray (127 evaluated), RGB + 50-class "
                            "segmentation, rays generated in-kernel, default-init weights seed 0",
                "precision": args.precision,
                "rays_per_gpu": rays_per_rank,
                "frames": world if args.scaling == "weak" else 1,
                "sharding": "one frame per GPU" if args.scaling == "weak" else "row blocks of one frame",
                "collectives": "none",
                "rendezvous_backend": backend,
            },
            "roofline": roofline(args.precision, rays_per_rank, kernel_ms, launches,
                                 world == 1 and args.scaling == "weak"),
        }
        if world == 1 and not args.no_cpu_baseline:
            # the other arithmetic of the same kernel, same frame, same run
            other = "fp32" if args.precision == "f16x3" else "f16x3"
            model.precision = other
            o_steps = max(3, args.steps // 4)
            o_elapsed, o_kernel_ms, o_launches = timed(o_steps, 1)
            model.precision = args.precision
            line["other_precision"] = {
                "precision": other, "dtype": PRECISIONS[other]["dtype"],
                "value": rays_per_rank * SAMPLES * o_steps / o_elapsed, "unit": "ray-samples/s",
                "steps": o_steps, "ms_per_step": o_elapsed / o_steps * 1e3,
                "roofline": roofline(other, rays_per_rank, o_kernel_ms, o_launches, True),
            }
        if world == 1 and not args.no_cpu_baseline:
            line["train_step"] = train_step_timing(dev)
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
