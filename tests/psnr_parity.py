"""PSNR parity of the training loop at convergence scale (BASELINE target "PSNR within 0.01 dB of reference").

The loop being replaced is train_conditional_nerf.py:115-153: Adam(lr) on the MSE of ``render_rays`` pixels, held-out
view rendered by ``render_image`` every ``log_interval`` steps, PSNR = -10 ln(mse) / ln 10 (:152-153).  tiny_nerf's
Lego file is not available offline, so the scene is the synthetic stand-in of nerf_amd.trainer.synthetic_scene (views
of a teacher field rendered by the renderer itself).  This file is TEST INFRASTRUCTURE (it drives the oracle, which
only tests/, smoke() and bench.py's cpu_baseline leg may do — hence tests/, not scripts/); pytest does not collect it,
tests/test_gpu_psnr_parity.py imports its pieces for a 300-step version.

From ONE seed (initial parameters, per-step ray indices, stratified draws ``u`` and density noise — all CAPTURED on
the host and fed to every run) it trains

  cpu_mt / cpu_4t / cpu_2t / cpu_1t  the oracle (the reference's ATen ops + autograd, oracle/nerf_oracle.py) with
                 torch Adam on the host, on 8, 4, 2 and 1 threads: the SAME algorithm under four summation orders —
                 six pairs whose PSNR differences are the noise floor any 0.01 dB statement has to be read against
                 (a training trajectory amplifies rounding differences exponentially until they saturate);
  hip_fp32       nerf_amd.trainer.Trainer, reference arithmetic, eager steps;
  hip_f16x3      the same in split-precision arithmetic;
  hip_f16x3_graph  the same, every step one HIP-graph replay (the captured draws enter as static inputs);

and, with the production RNG path (part "rng"), three seeds of Trainer(rng="philox") — in-kernel draws, graph
replayed — against three seeds of the oracle drawing from torch's generator: trajectories that share nothing but the
recipe, compared as distributions.  Held-out PSNR of every run every ``--every`` steps.

    python tests/psnr_parity.py scene   --out gpurun_out/r6_psnr            # GPU: render the scene once
    python tests/psnr_parity.py run     --out gpurun_out/r6_psnr --part captured|rng [--steps 2000]
    python tests/psnr_parity.py merge   --out gpurun_out/r6_psnr --json profiles/r06_psnr_parity.json

``run`` starts the oracle trajectories as CHILD processes before it makes any GPU call (a GPU-initialised process
must not start another program on this pool), then trains the HIP runs itself.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

VIEWS, SIZE, SAMPLES, BATCH, LR, NOISE_STD = 6, 32, 32, 256, 5e-4, 1.0


def captured_step(gen, n_examples, batch=BATCH, samples=SAMPLES):
    """Ray indices, stratified draws and density noise of one step, from the study's host generator."""
    idx = torch.randint(0, n_examples, (batch,), generator=gen)
    u = torch.rand(batch, samples, generator=gen)
    noise = torch.randn(batch, samples - 1, 1, generator=gen)
    return idx, u, noise


def load_scene(out):
    with np.load(os.path.join(out, "scene.npz")) as z:
        return (torch.from_numpy(z["images"]), torch.from_numpy(z["poses"]), float(z["focal"]),
                {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param.")})


def host_batches(images, poses, focal):
    """The training examples as the oracle sees them: PixelRayDataset's decode (nerf/dataset.py:283-291) on the host —
    rays of every pixel of every training view (all but the last), flat index = (image * H + row) * W + col."""
    from oracle import nerf_oracle as O
    train = images[:-1]
    rays_o, rays_d = [], []
    for v in range(train.shape[0]):
        o, d = O.image_rays(poses[v:v + 1, :3, 3], poses[v:v + 1, :3, :3], train.shape[1], train.shape[2], focal)
        rays_o.append(o)
        rays_d.append(d)
    return torch.cat(rays_o), torch.cat(rays_d), train.reshape(-1, 3)


def oracle_run(out, tag, threads, steps, every, seed, own_draws):
    """One oracle trajectory on `threads` host threads.  own_draws: draw u / noise / indices from torch's generator
    seeded with `seed` (the reference's own behaviour, nerf/model.py:432, :652) instead of the captured stream."""
    from oracle import nerf_oracle as O
    torch.set_num_threads(threads)
    images, poses, focal, params0 = load_scene(out)
    cfg = dict(O.default_config(), focal_length=focal)
    rays_o, rays_d, pixels = host_batches(images, poses, focal)
    ref = {k: v.clone().requires_grad_(k.startswith("prediction")) for k, v in params0.items()}
    opt = torch.optim.Adam([ref[k] for k in ref if k.startswith("prediction")], lr=LR)
    gen = torch.Generator().manual_seed(seed)
    cam_o, cam_r = poses[-1:, :3, 3].contiguous(), poses[-1:, :3, :3].contiguous()
    losses, psnrs = [], {}
    t0 = time.perf_counter()
    for step in range(1, steps + 1):
        idx, u, noise = captured_step(gen, rays_o.shape[0])
        loss = O.training_loss(ref, cfg, rays_o[idx], rays_d[idx], SAMPLES, pixels[idx], u, noise, NOISE_STD)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
        if step % every == 0 or step == steps:
            with torch.no_grad():
                render, _ = O.render_image({k: v.detach() for k, v in ref.items()}, cfg, cam_o, cam_r, SIZE, SIZE,
                                           focal, SAMPLES)
            psnrs[step] = float(O.psnr(render, images[-1:]))
            print(f"[{tag}] step {step}: loss {losses[-1]:.6f}, held-out PSNR {psnrs[step]:.4f} dB, "
                  f"{time.perf_counter() - t0:.0f} s", flush=True)
    result = {"tag": tag, "kind": "oracle (CPU port of the reference's ops, torch Adam)", "threads": threads,
              "seed": seed, "lr": LR, "draws": "own torch generator" if own_draws else "captured stream", "steps": steps,
              "seconds": time.perf_counter() - t0, "loss": losses, "psnr": psnrs}
    with open(os.path.join(out, f"traj_{tag}.json"), "w") as f:
        json.dump(result, f)
    return result


def hip_run(out, tag, steps, every, seed, train_precision, graph, rng):
    """One trajectory of the HIP Trainer path.  rng "captured": the study's host stream enters every step as
    batch["u"] / batch["noise"] (static inputs under graph replay); "philox": the production path, in-kernel draws,
    example order from the Trainer's own sampler."""
    from nerf_amd import NeRF
    from nerf_amd import trainer as T
    dev = torch.device("cuda:0")
    images, poses, focal, params0 = load_scene(out)
    images, poses = images.to(dev), poses.to(dev)
    model = NeRF(focal_length=focal)
    model.load_state_dict(params0)
    model = model.to(dev)
    model.train_precision = train_precision
    run = T.Trainer(images, poses, focal, batch_size=BATCH, learning_rate=LR, num_samples_per_ray=SAMPLES,
                    density_noise_std=NOISE_STD, log_interval=10 ** 9, model=model, seed=seed,
                    rng="philox" if rng == "philox" else "torch", graph=graph)
    gen = torch.Generator().manual_seed(seed)
    pick = torch.Generator().manual_seed(seed + 1)
    losses, psnrs = [], {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for step in range(1, steps + 1):
        if rng == "captured":
            idx, u, noise = captured_step(gen, len(run.dataset))
            b = run.dataset.gather(idx.to(dev))
            b["u"], b["noise"] = u.to(dev), noise.to(dev)
        else:
            b = run.dataset.gather(torch.randint(0, len(run.dataset), (BATCH,), generator=pick).to(dev))
        run.iteration += 1
        losses.append(torch.as_tensor(run.train_step(b)).detach().reshape(()).clone())   # (a replay returns its static tensor)
        if step % every == 0 or step == steps:
            psnrs[step] = run.evaluate()
            print(f"[{tag}] step {step}: loss {float(losses[-1]):.6f}, held-out PSNR {psnrs[step]:.4f} dB, "
                  f"{time.perf_counter() - t0:.0f} s", flush=True)
    torch.cuda.synchronize()
    seconds = time.perf_counter() - t0
    result = {"tag": tag, "kind": "HIP Trainer (nerf_amd.trainer.Trainer)", "train_precision": train_precision,
              "graph_replay": bool(graph), "replayed": run._graph is not None, "seed": seed, "lr": LR,
              "draws": "captured stream" if rng == "captured" else "in-kernel Philox", "steps": steps,
              "seconds": seconds, "loss": [float(x) for x in torch.stack(losses).cpu()],
              "psnr": psnrs}
    with open(os.path.join(out, f"traj_{tag}.json"), "w") as f:
        json.dump(result, f)
    return result


def make_scene(out):
    from nerf_amd import trainer as T
    from oracle import nerf_oracle as O
    os.makedirs(out, exist_ok=True)
    dev = torch.device("cuda:0")
    images, poses, focal = T.synthetic_scene(num_views=VIEWS, size=SIZE, num_samples=SAMPLES, device=dev, seed=3)
    params = O.init_params(seed=0, cfg=dict(O.default_config(), focal_length=focal))
    np.savez(os.path.join(out, "scene.npz"), images=images.cpu().numpy(), poses=poses.cpu().numpy(),
             focal=np.float64(focal), **{"param." + k: v.numpy() for k, v in params.items()})
    print(f"scene: {VIEWS} views of {SIZE}x{SIZE}, focal {focal:.2f}, mean {float(images.mean()):.3f}")


def spawn_oracle(out, tag, threads, steps, every, seed, own):
    cmd = [sys.executable, os.path.abspath(__file__), "oracle", "--out", out, "--tag", tag, "--threads", str(threads),
           "--steps", str(steps), "--every", str(every), "--seed", str(seed), "--lr", str(LR)] + (["--own-draws"] if own else [])
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), CUDA_VISIBLE_DEVICES="",
               HIP_VISIBLE_DEVICES="")
    log = open(os.path.join(out, f"traj_{tag}.log"), "w")
    return subprocess.Popen(cmd, env=env, stdout=log, stderr=subprocess.STDOUT)


def merge(out, path):
    """Assemble every trajectory under `out` into one record: PSNR per checkpoint, |dPSNR| of every HIP run against
    cpu_mt, the oracle's own spread (cpu_1t vs cpu_mt) beside it; for the RNG part the mean / min / max over seeds."""
    runs = {}
    for f in sorted(os.listdir(out)):
        if f.startswith("traj_") and f.endswith(".json"):
            with open(os.path.join(out, f)) as fh:
                r = json.load(fh)
            runs[r["tag"]] = r
    record = {"what": "held-out PSNR of the training loop of train_conditional_nerf.py:115-153 on a synthetic stand-in "
                      f"scene ({VIEWS} views {SIZE}x{SIZE}, last held out), {BATCH}-ray batches, {SAMPLES} samples/ray, "
                      f"Adam lr {next(iter(runs.values())).get('lr', LR) if runs else LR}, density noise {NOISE_STD}; oracle = CPU port of the reference (the reference itself "
                      "does not travel to the GPU box)",
              "runs": {t: {k: v for k, v in r.items() if k != "loss"} for t, r in runs.items()}}
    base = runs.get("cpu_mt")
    if base is not None:
        table = {}
        for step in sorted(base["psnr"], key=int):
            row = {"cpu_mt": base["psnr"][step]}
            for t, r in runs.items():
                if r["draws"] == "captured stream" and t != "cpu_mt" and step in r["psnr"]:
                    row[t] = r["psnr"][step]
                    row["d_" + t] = abs(r["psnr"][step] - base["psnr"][step])
            table[step] = row
        oracles = sorted(t for t, r in runs.items() if r["draws"] == "captured stream" and r["kind"].startswith("oracle"))
        steps_sorted = sorted(base["psnr"], key=int)
        late = steps_sorted[len(steps_sorted) // 2:]                   # the second half of the run: "converged"
        floor = {}
        for step in steps_sorted:
            d = [abs(runs[a]["psnr"][step] - runs[b]["psnr"][step]) for i, a in enumerate(oracles) for b in oracles[i + 1:]]
            floor[step] = {"pairs": len(d), "median": float(np.median(d)) if d else None, "max": max(d) if d else None}
        late_mean = {t: float(np.mean([r["psnr"][s] for s in late])) for t, r in runs.items()
                     if r["draws"] == "captured stream"}
        record["captured_draws"] = {
            "oracle_pair_floor": {"what": "|PSNR_a - PSNR_b| over every pair of oracle runs (same algorithm, same draws, "
                                          "thread counts " + ", ".join(str(runs[t]["threads"]) for t in oracles) + ")",
                                  "per_checkpoint": floor},
            "late_mean_psnr": {"what": f"mean held-out PSNR over steps {late[0]} .. {late[-1]}", "values": late_mean,
                               "oracle_spread": max(late_mean[t] for t in oracles) - min(late_mean[t] for t in oracles),
                               "hip_minus_oracle_mean": {t: late_mean[t] - float(np.mean([late_mean[o] for o in oracles]))
                                                         for t in late_mean if t not in oracles}},
            "checkpoints": table,
            "noise_floor": "d_cpu_1t / d_cpu_2t / d_cpu_4t = |PSNR(oracle, n threads) - PSNR(oracle, 8 threads)|: the same "
                           "algorithm, the same draws, other summation orders",
            "max_abs_dpsnr": {t: max(row.get("d_" + t, 0.0) for row in table.values())
                              for t in runs if t != "cpu_mt" and runs[t]["draws"] == "captured stream"},
            "max_rel_loss_deviation": {
                t: float(max(abs(a - b) / max(b, 1e-12) for a, b in zip(runs[t]["loss"], base["loss"])))
                for t in runs if t != "cpu_mt" and runs[t]["draws"] == "captured stream"},
            "first_100_steps_max_rel_loss_deviation": {
                t: float(max(abs(a - b) / max(b, 1e-12) for a, b in zip(runs[t]["loss"][:100], base["loss"][:100])))
                for t in runs if t != "cpu_mt" and runs[t]["draws"] == "captured stream"}}
    own = {"oracle_torch_rng": [r for t, r in runs.items() if r["draws"] == "own torch generator"],
           "hip_philox": [r for t, r in runs.items() if r["draws"] == "in-kernel Philox"]}
    if own["oracle_torch_rng"] and own["hip_philox"]:
        steps = sorted(own["oracle_torch_rng"][0]["psnr"], key=int)
        dist = {}
        for step in steps:
            row = {}
            for name, rs in own.items():
                vals = [r["psnr"][step] for r in rs if step in r["psnr"]]
                row[name] = {"mean": float(np.mean(vals)), "min": min(vals), "max": max(vals), "seeds": len(vals)}
            row["d_mean"] = abs(row["hip_philox"]["mean"] - row["oracle_torch_rng"]["mean"])
            dist[step] = row
        record["independent_draws"] = {"checkpoints": dist,
                                       "note": "trajectories share the recipe only; compare the means against the "
                                               "seed-to-seed spread of either side"}
    with open(path, "w") as f:
        json.dump(record, f, indent=1)
    print(json.dumps({k: v for k, v in record.items() if k != "runs"}, indent=1)[:6000])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cmd", choices=("scene", "run", "oracle", "merge"))
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r6_psnr"))
    ap.add_argument("--part", choices=("captured", "rng"), default="captured")
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--every", type=int, default=100)
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--tag", default="cpu_mt")
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--own-draws", action="store_true")
    ap.add_argument("--seeds", type=int, default=3, help="part rng: seeds per side")
    ap.add_argument("--first-seed", type=int, default=0)
    ap.add_argument("--lr", type=float, default=LR, help="Adam learning rate (the reference's script: 1e-4)")
    ap.add_argument("--oracle-threads", default="1,2,4,8", help="part captured: thread counts of the oracle runs (the last one is the base, cpu_mt)")
    ap.add_argument("--json", default=os.path.join(ROOT, "profiles", "r06_psnr_parity.json"))
    args = ap.parse_args()
    globals()["LR"] = args.lr
    if args.cmd == "scene":
        return make_scene(args.out)
    if args.cmd == "oracle":
        return oracle_run(args.out, args.tag, args.threads, args.steps, args.every, args.seed, args.own_draws)
    if args.cmd == "merge":
        return merge(args.out, args.json)
    # run: the oracle trajectories first, as children, before this process touches the GPU
    if args.part == "captured":
        # four summation orders of the SAME algorithm on the SAME draws (1 + 2 + 4 + 8 = 15 of the box's 16 CPUs)
        threads = [int(t) for t in args.oracle_threads.split(",")]
        kids = [spawn_oracle(args.out, f"cpu_{t}t" if t != threads[-1] else "cpu_mt", t, args.steps, args.every, args.seed, False)
                for t in threads]
        for tag, prec, graph in (("hip_fp32", "fp32", False), ("hip_f16x3", "f16x3", False),
                                 ("hip_f16x3_graph", "f16x3", True), ("hip_fp32_graph", "fp32", True)):
            hip_run(args.out, tag, args.steps, args.every, args.seed, prec, graph, "captured")
    else:
        seeds = list(range(args.first_seed, args.first_seed + args.seeds))
        kids = [spawn_oracle(args.out, f"cpu_rng{s}", 5, args.steps, args.every, 100 + s, True) for s in seeds[:3]]
        for s in seeds:
            hip_run(args.out, f"hip_philox{s}", args.steps, args.every, 200 + s, "f16x3", True, "philox")
        for i in range(3, len(seeds), 3):                     # three oracle runs at a time (5 threads each)
            while any(k.poll() is None for k in kids):
                time.sleep(10)
                print("waiting for the oracle runs", flush=True)
            kids += [spawn_oracle(args.out, f"cpu_rng{s}", 5, args.steps, args.every, 100 + s, True) for s in seeds[i:i + 3]]
    while any(k.poll() is None for k in kids):            # a line a minute: the box takes silence for a hang
        time.sleep(30)
        tails = []
        for f in sorted(os.listdir(args.out)):
            if f.endswith(".log"):
                with open(os.path.join(args.out, f)) as fh:
                    lines = fh.read().strip().splitlines()
                tails.append(lines[-1] if lines else f)
        print(" | ".join(tails)[-400:], flush=True)
    bad = [k.returncode for k in kids if k.returncode != 0]
    if bad:
        raise SystemExit(f"oracle child failed: {bad}")


if __name__ == "__main__":
    main()
