"""GPU test of the N>1 path with 2 ranks sharing the one MI355X of the test box (gloo for the
rendezvous and the collective — RCCL refuses two ranks on one device; the 8-GPU RCCL run is the
driver's scaling bench).  Checks with the real HIP kernels: row-sharded render == whole frame,
all-reduced data-parallel gradients == single-process gradients of the concatenated batch for an
UNEVEN split (weighted by local / global rays, reduced in place on the backward's flat vector), and
the in-kernel draws of the two ranks differ."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import golden_params
from test_parallel_cpu import free_port

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nerf_amd import NeRF, parallel
        dev = torch.device("cuda:0")
        model = NeRF()
        model.load_state_dict(golden_params(3.0))
        model = model.to(dev)
        parallel.broadcast_parameters(model)
        # inference: row blocks, gathered
        cam_o = torch.tensor([[0.0, -3.0, 2.6]], device=dev)
        from oracle import nerf_oracle as O
        cam_r = O.look_at_pose([0.0, -3.0, 2.6]).to(dev)
        with torch.no_grad():
            img, seg, rows = parallel.render_image_sharded(model, cam_o, cam_r, 37, 29, 32.0, 48,
                                                           gather=True)
        # training: an uneven split of the rays (61 + 35), one flat all-reduce, in place
        torch.manual_seed(7)
        o, d, tgt = torch.randn(96, 3), torch.randn(96, 3), torch.rand(96, 3)
        u, noise = torch.rand(96, 40), torch.randn(96, 39, 1)
        lo, hi = (0, 61) if rank == 0 else (61, 96)
        pix, _ = model.render_rays(o[lo:hi].to(dev), d[lo:hi].to(dev), 40, randomly_sample=True,
                                   density_noise_std=1.0, u=u[lo:hi].to(dev), noise=noise[lo:hi].to(dev))
        ((pix[:, 0] - tgt[lo:hi].to(dev)) ** 2).mean().backward()
        reduce = parallel.FlatGradientAllReduce(model.parameters())
        flat = reduce(model.last_flat_grad, (hi - lo) / 96)
        assert reduce.in_place_calls == 1 and flat.data_ptr() == model.last_flat_grad.data_ptr()
        grads = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
        assert torch.equal(grads, flat)                  # p.grad are views of the reduced vector
        # in-kernel (Philox) draws: the rank is part of the key, so the two ranks draw differently
        model.rng = "philox"
        rays_o = torch.zeros(32, 3, device=dev)
        rays_d = torch.tensor([[1.0, 0.0, 0.0]], device=dev).repeat(32, 1)
        t = model.fenceposts_used(rays_o, rays_d, 40, randomly_sample=True)
        torch.save(dict(img=img.cpu(), seg=seg.cpu(), rows=rows, flat=flat.cpu(), t=t.cpu()),
                   os.path.join(out_dir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu(tmp_path):
    mp.spawn(_worker, args=(2, free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "r0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "r1.pt"))
    assert r0["rows"] == (0, 19) and r1["rows"] == (19, 37)
    assert torch.equal(r0["img"], r1["img"]) and torch.equal(r0["flat"], r1["flat"])
    # same seed, same rays, same launch count — different rank: different stratified draws
    assert float(((r0["t"] - r1["t"]).abs() < 1e-6)[:, 1:-1].float().mean()) < 0.02

    from nerf_amd import NeRF
    from oracle import nerf_oracle as O
    dev = torch.device("cuda:0")
    model = NeRF()
    model.load_state_dict(golden_params(3.0))
    model = model.to(dev)
    cam_o = torch.tensor([[0.0, -3.0, 2.6]], device=dev)
    cam_r = O.look_at_pose([0.0, -3.0, 2.6]).to(dev)
    with torch.no_grad():
        img, seg = model.render_image(cam_o, cam_r, 37, 29, 32.0, 48)
    assert torch.equal(img.cpu(), r0["img"]) and torch.equal(seg.cpu(), r0["seg"])

    torch.manual_seed(7)
    o, d, tgt = torch.randn(96, 3), torch.randn(96, 3), torch.rand(96, 3)
    u, noise = torch.rand(96, 40), torch.randn(96, 39, 1)
    pix, _ = model.render_rays(o.to(dev), d.to(dev), 40, randomly_sample=True, density_noise_std=1.0,
                               u=u.to(dev), noise=noise.to(dev))
    ((pix[:, 0] - tgt.to(dev)) ** 2).mean().backward()
    full = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).cpu()
    assert full.numel() == 304438
    scale = full.abs().max()
    assert (r0["flat"] - full).abs().max() <= 2e-6 * scale


def _graph_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nerf_amd import trainer as T
        dev = torch.device("cuda:0")
        images, poses, focal = T.synthetic_scene(num_views=5, size=16, num_samples=24, device=dev)
        run = T.Trainer(images, poses, focal, batch_size=200, learning_rate=5e-4, num_samples_per_ray=24,
                        density_noise_std=0.5, log_interval=10 ** 9, seed=3, graph=True)
        run.model.train_precision = "f16x3"
        first = float(run.fit(epochs=1, max_iterations=1))
        last = float(run.fit(epochs=40))                  # 1,024 rays per epoch: 5 batches of 100 + a tail of 12 per rank
        params = torch.cat([p.detach().reshape(-1) for p in run.model.parameters()]).cpu()
        torch.save(dict(first=first, last=last, params=params, graph_rays=run._graph_rays),
                   os.path.join(out_dir, f"g{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_two_ranks_graph_replayed_training(tmp_path):
    """Data-parallel Trainer(graph=True): backward inside the graph, all-reduce + Adam outside; the two
    replicas must stay bitwise identical (same reduced gradients) and the fit must progress."""
    mp.spawn(_graph_worker, args=(2, free_port(), str(tmp_path)), nprocs=2, join=True)
    g0 = torch.load(os.path.join(tmp_path, "g0.pt"))
    g1 = torch.load(os.path.join(tmp_path, "g1.pt"))
    assert g0["graph_rays"] == 100 and g1["graph_rays"] == 100
    assert torch.equal(g0["params"], g1["params"])
    assert g0["last"] == g0["last"] and g0["last"] < 0.5 * g0["first"]


def _legacy_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nerf_amd import parallel, trainer as T
        from nerf_amd.legacy import LegacyNeRF8x256
        from oracle import legacy_oracle as L
        dev = torch.device("cuda:0")
        model = LegacyNeRF8x256()
        model.load_state_dict(L.init_params(seed=4))
        model = model.to(dev)
        model.train_precision = "f16x3"
        parallel.broadcast_parameters(model)
        # an uneven split of one batch (61 + 35 rays), one in-place all-reduce of the flat 638,468-float gradient
        torch.manual_seed(7)
        o, d, tgt = torch.randn(96, 3) * 0.5, torch.randn(96, 3), torch.rand(96, 3)
        u, noise = torch.rand(96, 24), torch.randn(96, 24)
        lo, hi = (0, 61) if rank == 0 else (61, 96)
        pix = model.render_rays(o[lo:hi].to(dev), d[lo:hi].to(dev), 2.0, 6.0, 24, randomly_sample=True,
                                density_noise_std=1.0, u=u[lo:hi].to(dev), noise=noise[lo:hi].to(dev))
        ((pix - tgt[lo:hi].to(dev)) ** 2).mean().backward()
        reduce = parallel.FlatGradientAllReduce(model.parameters())
        flat = reduce(model.last_flat_grad, (hi - lo) / 96)
        assert reduce.in_place_calls == 1 and flat.data_ptr() == model.last_flat_grad.data_ptr()
        # and the trainer: a few data-parallel steps of the notebook's recipe, replicas must stay identical
        images, poses, focal = T.synthetic_scene(num_views=4, size=12, num_samples=24, device=dev)
        run = T.Trainer(images, poses, focal, batch_size=128, learning_rate=5e-4, num_samples_per_ray=24,
                        density_noise_std=0.5, log_interval=10 ** 9, seed=3, model=model, near=2.0, far=6.0)
        last = float(run.fit(epochs=2))
        params = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()
        torch.save(dict(flat=flat.cpu(), params=params, last=last), os.path.join(out_dir, f"l{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_two_ranks_train_the_legacy_network(tmp_path):
    """The data-parallel contract holds for the legacy 8 x 256 network too: the backward's flat gradient (in
    parameters() order) is all-reduced in place, the weighted sum over uneven shards equals the single-process
    gradient of the whole batch, and `Trainer` keeps two replicas bitwise identical."""
    mp.spawn(_legacy_worker, args=(2, free_port(), str(tmp_path)), nprocs=2, join=True)
    l0 = torch.load(os.path.join(tmp_path, "l0.pt"))
    l1 = torch.load(os.path.join(tmp_path, "l1.pt"))
    assert torch.equal(l0["flat"], l1["flat"]) and torch.equal(l0["params"], l1["params"])
    assert l0["last"] == l0["last"]
    from nerf_amd.legacy import LegacyNeRF8x256
    from oracle import legacy_oracle as L
    dev = torch.device("cuda:0")
    model = LegacyNeRF8x256()
    model.load_state_dict(L.init_params(seed=4))
    model = model.to(dev)
    model.train_precision = "f16x3"
    torch.manual_seed(7)
    o, d, tgt = torch.randn(96, 3) * 0.5, torch.randn(96, 3), torch.rand(96, 3)
    u, noise = torch.rand(96, 24), torch.randn(96, 24)
    pix = model.render_rays(o.to(dev), d.to(dev), 2.0, 6.0, 24, randomly_sample=True, density_noise_std=1.0,
                            u=u.to(dev), noise=noise.to(dev))
    ((pix - tgt.to(dev)) ** 2).mean().backward()
    full = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).cpu()
    assert full.numel() == 638468
    assert (l0["flat"] - full).abs().max() <= 1e-5 * full.abs().max()       # (f16-pair products: ~2^-22 each)


def test_the_drivers_multi_gpu_bench_command_runs_as_a_rehearsal(tmp_path):
    """The driver launches ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W``.  On this one-GPU box the same command runs with
    ``--allow-gloo`` as a rehearsal (two ranks share the device, barriers over gloo; without the flag it must
    refuse): every line of the N > 1 path executes — row blocks, barrier + synchronize fences, MAX over ranks,
    the weak-scaling leg — and the one JSON line says what it was."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
            "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps",
            "2", "--warmup", "1"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    refused = subprocess.run(base, capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=600)
    assert refused.returncode != 0 and "refusing to measure" in refused.stderr
    base[base.index("--master-port") + 1] = str(free_port())
    run = subprocess.run(base + ["--allow-gloo"], capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [x for x in run.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1                                    # rank 0 prints ONE line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "strong"
    assert line["config"]["rendezvous_backend"] == "gloo" and "rehearsal" in line["config"]
    assert line["config"]["rays_per_gpu"] == 400 * 800 and line["config"]["collectives"] == "none"
    # two ranks time-share one GPU: the frame takes about as long as on one rank, never half
    assert 0.5 * 2.2e8 <= line["value"] <= 1.2 * 2.3e8
    assert line["weak_scaling"]["frames"] == 2 and line["weak_scaling"]["value"] > 0.5 * 2.2e8
    # BASELINE config 5 in the N > 1 line: the data-parallel training step with its collective
    dp = line["train_step_dp"]
    assert dp["rendezvous_backend"] == "gloo" and dp["rays_per_rank"] == 2048 and dp["global_batch"] == 4096
    assert dp["gradient_bytes"] == 304438 * 4 and dp["replicas_identical"] is True and dp["parameters_finite"] is True
    assert dp["eager"]["ms_per_step"] > 0 and dp["eager"]["allreduce_ms"] > 0
    assert dp["graph"]["ms_per_step"] > 0 and dp["graph"]["collective_and_optimiser_in_graph"] is False   # gloo: outside
    weak = line["train_step_dp_weak"]
    assert weak["scaling"] == "weak" and weak["rays_per_rank"] == 4096 and weak["global_batch"] == 8192
    assert weak["replicas_identical"] is True and weak["parameters_finite"] is True


def test_plain_bench_gpus_n_starts_its_own_ranks(tmp_path):
    """The OTHER launch form: ``python bench.py --gpus 4 --allow-gloo`` with no torch.distributed.run around it and no
    WORLD_SIZE in the environment.  bench.py must start the ranks itself (a child launcher, before any GPU call of
    its own), relay rank 0's one JSON line and the exit code.  Four ranks share this box's GPU (the pool allows at
    most 6 processes on a card, so the 8-rank form is rehearsed without the GPU: tests/test_bench_contract.py);
    the partitions are the N = 4 ones: 200 rows = 160,000 rays of the frame, 1024 rays of the config-5 batch."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--allow-gloo", "--steps", "2",
                          "--warmup", "1"], capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    assert "starting the ranks" in run.stderr
    lines = [x for x in run.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 4 and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "strong"
    assert line["config"]["rays_per_gpu"] == 160000 and line["config"]["collectives"] == "none"
    assert line["config"]["rendezvous_backend"] == "gloo" and "rehearsal" in line["config"]
    assert 0.5 * 2.2e8 <= line["value"] <= 1.2 * 2.3e8          # four ranks time-share ONE GPU: about one GPU's rate
    assert line["weak_scaling"]["frames"] == 4
    dp = line["train_step_dp"]
    assert dp["rays_per_rank"] == 1024 and dp["global_batch"] == 4096 and dp["gradient_bytes"] == 304438 * 4
    assert dp["replicas_identical"] is True and dp["parameters_finite"] is True and dp["scaling"] == "strong"
    weak = line["train_step_dp_weak"]              # 4096 rays PER RANK: what an 8-GPU record needs to isolate RCCL's cost
    assert weak["scaling"] == "weak" and weak["rays_per_rank"] == 4096 and weak["global_batch"] == 4 * 4096
    assert weak["rendezvous_backend"] == "gloo" and weak["replicas_identical"] is True and weak["parameters_finite"] is True
    assert weak["eager"]["ms_per_step"] > 0 and weak["graph"]["ms_per_step"] > 0


def _rccl_single_rank_worker(rank, world, port, out_dir):
    """ONE rank on RCCL (backend "nccl"): the only form of the RCCL path a one-GPU box can run.  The collective is
    then a (trivial) stream-ordered RCCL launch — what matters here is that it, the scale in front of it and the
    one-launch Adam behind it are CAPTURED in the step's HIP graph and replayed, and that in-kernel Philox draws
    differ from replay to replay (device-resident launch counter)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        import bench
        from nerf_amd import trainer as T

        def fence():
            dist.barrier()
            torch.cuda.synchronize(dev)

        dp = bench.train_step_dp(dev, 0, 1, "nccl", steps=4, warmup=1, fence=fence, rays=512)
        images, poses, focal = T.synthetic_scene(num_views=4, size=12, num_samples=24, device=dev)
        run = T.Trainer(images, poses, focal, batch_size=128, learning_rate=5e-4, num_samples_per_ray=24,
                        density_noise_std=0.5, log_interval=10 ** 9, seed=3, graph=True, rng="philox")
        assert run.distributed and run.collective_in_graph
        before = [p.detach().clone() for p in run.model.parameters()]
        losses = []
        gen = torch.Generator().manual_seed(5)
        for _ in range(14):                                        # 5 eager warm-up steps, capture, 8 replays
            idx = torch.randint(0, len(run.dataset), (128,), generator=gen)
            run.iteration += 1
            losses.append(float(run.train_step(run.dataset.gather(idx.to(dev)))))
        counter = int(run.model._philox_counter.item())
        moved = any(not torch.equal(a, b) for a, b in zip(before, run.model.parameters()))
        torch.save(dict(dp=dp, losses=losses, counter=counter, moved=moved, captured=run._graph is not None),
                   os.path.join(out_dir, "rccl1.pt"))
    finally:
        dist.destroy_process_group()


def test_single_rank_rccl_group_captures_collective_and_optimiser_in_the_step_graph(tmp_path):
    mp.spawn(_rccl_single_rank_worker, args=(1, free_port(), str(tmp_path)), nprocs=1, join=True)
    out = torch.load(os.path.join(tmp_path, "rccl1.pt"), weights_only=False)
    dp = out["dp"]
    assert dp["rendezvous_backend"] == "nccl" and dp["replicas_identical"] and dp["parameters_finite"]
    assert dp["graph"]["collective_and_optimiser_in_graph"] is True, dp["graph"]
    assert dp["graph"]["ms_per_step"] > 0 and dp["eager"]["allreduce_ms"] > 0
    assert out["captured"] and out["moved"] and all(l == l and l < 10 for l in out["losses"])
    assert out["counter"] == 14                                    # one device-side advance per training forward,
                                                                   # replayed ones included: every replay drew anew
