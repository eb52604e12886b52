"""Next rows N1/N2: the dataset mirror against fixture G8 on CPU; the
on-device batched sampler and the training loop on the GPU."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import load_golden


def make_dataset(g, device="cpu"):
    from nerf_amd.dataset import PixelRayDataset
    return PixelRayDataset(g["images"].to(device), g["segmentation"].to(device), g["poses"].to(device),
                           112.0)


def test_getitem_matches_reference_fixture():
    g = load_golden("g8_pixel_dataset")
    ds = make_dataset(g)
    assert len(ds) == int(g["length"]) == 3 * 6 * 5
    for n, idx in enumerate(g["picks"].tolist()):
        item = ds[idx]
        assert set(item) == {"image_wi", "image_hi", "image_bi", "states_x", "states_d", "pixels",
                             "label", "rays", "pose_o", "pose_d", "rays_o", "rays_d"}
        for key in ("image_wi", "image_hi", "image_bi", "pixels", "label", "rays", "pose_o", "pose_d",
                    "rays_o", "rays_d"):
            assert torch.equal(item[key], g[key][n]), (idx, key)
        assert item["image_wi"].dtype == torch.int64 and item["states_x"].shape == (0,)


def test_gather_refuses_cpu():
    ds = make_dataset(load_golden("g8_pixel_dataset"))
    with pytest.raises(RuntimeError):
        ds.gather(torch.arange(4))


@pytest.mark.gpu
def test_gather_equals_collated_getitem():
    g = load_golden("g8_pixel_dataset")
    dev = torch.device("cuda:0")
    ds = make_dataset(g, dev)
    idx = torch.cat([g["picks"], torch.arange(90), torch.tensor([90, 91, 179])])      # ids wrap (% B)
    batch = ds.gather(idx.to(dev))
    cpu = make_dataset(g)
    for n, i in enumerate(idx.tolist()):
        item = cpu[i]
        for key in ("image_wi", "image_hi", "image_bi", "pixels", "label", "rays", "pose_o", "pose_d",
                    "rays_o", "rays_d"):
            assert torch.equal(batch[key][n].cpu(), item[key]), (i, key)
    assert batch["states_x"].shape == (idx.shape[0], 0)
    # an epoch is a permutation: every example exactly once, ranks take disjoint shares
    seen = torch.cat([b["image_bi"][:, 0] * 30 + b["image_hi"][:, 0] * 5 + b["image_wi"][:, 0]
                      for b in ds.batches(32, generator=torch.Generator(device=dev).manual_seed(1))])
    assert torch.equal(seen.sort().values.cpu(), torch.arange(90))
    parts = [torch.cat([b["image_bi"][:, 0] * 30 + b["image_hi"][:, 0] * 5 + b["image_wi"][:, 0]
                        for b in ds.batches(32, generator=torch.Generator(device=dev).manual_seed(1),
                                            rank=r, world_size=2)]) for r in range(2)]
    assert torch.equal(torch.cat(parts).sort().values.cpu(), torch.arange(90))


@pytest.mark.gpu
def test_training_improves_held_out_psnr_and_writes_reference_files(tmp_path):
    from nerf_amd import trainer as T
    dev = torch.device("cuda:0")
    images, poses, focal = T.synthetic_scene(num_views=9, size=24, num_samples=32, device=dev)
    assert images.shape == (9, 24, 24, 3) and float(images.std()) > 0.02
    run = T.Trainer(images, poses, focal, logging_dir=str(tmp_path), batch_size=512, learning_rate=5e-4,
                    num_samples_per_ray=32, density_noise_std=0.0, log_interval=300, seed=1)
    run.write_params(dict(batch_size=512))
    first_loss = float(run.fit(epochs=1, max_iterations=1))
    last_loss = float(run.fit(epochs=1000, max_iterations=901))
    assert run.iternums == [0, 300, 600, 900]
    print("held-out PSNR:", [round(float(p), 2) for p in run.psnrs], "loss", first_loss, "->", last_loss)
    assert last_loss < 0.1 * first_loss                       # the fit itself
    # held-out view: 8 training views of a random field generalise only so far; it must improve
    assert max(float(p) for p in run.psnrs[1:]) > float(run.psnrs[0]) + 1.0
    for name in ("params.json", "model.pth", "psnrs.npy", "iternums.npy", "rendered_images.npy",
                 "ground_truth_images.npy"):
        assert os.path.exists(os.path.join(tmp_path, name)), name
    assert np.load(os.path.join(tmp_path, "rendered_images.npy")).shape == (4, 1, 24, 24, 3)
    assert json.load(open(os.path.join(tmp_path, "params.json")))["batch_size"] == 512
    state = torch.load(os.path.join(tmp_path, "model.pth"))
    assert list(state.keys())[:3] == ["rays_min", "rays_max", "prediction_heads.0.weight"]


@pytest.mark.gpu
@pytest.mark.parametrize("train_precision", ["fp32", "f16x3"])
def test_graph_replayed_training_fits_like_the_eager_loop(train_precision):
    """Trainer(graph=True): after five eager steps every full batch is one HIP-graph replay (forward,
    loss, backward, fused Adam).  It must train — same fit as the eager loop within noise — draw new
    stratified samples on every replay, and fall back to eager for the epoch's short tail batch."""
    from nerf_amd import trainer as T
    dev = torch.device("cuda:0")
    images, poses, focal = T.synthetic_scene(num_views=9, size=24, num_samples=32, device=dev)
    losses = {}
    for graph in (False, True):
        run = T.Trainer(images, poses, focal, batch_size=512, learning_rate=5e-4, num_samples_per_ray=32,
                        density_noise_std=0.0, log_interval=10 ** 9, seed=1, graph=graph)
        run.model.train_precision = train_precision
        first = float(run.fit(epochs=1, max_iterations=1))
        last = float(run.fit(epochs=1000, max_iterations=601))     # 4,608 rays per epoch: 9 x 512, no tail
        losses[graph] = (first, last)
        if graph:
            assert run._graph is not None and run._graph_rays == 512
            before = run._static_loss.clone()
            run._graph.replay()                                    # same batch again: new draws, new step
            torch.cuda.synchronize()
            assert not torch.equal(before, run._static_loss)
    assert losses[True][1] < 0.1 * losses[True][0]
    assert abs(losses[True][1] - losses[False][1]) < 0.5 * losses[False][1] + 1e-4

    # an epoch with a short tail (batch 500 of 4,608 rays -> tail of 108): the tail runs eagerly
    run = T.Trainer(images, poses, focal, batch_size=500, learning_rate=5e-4, num_samples_per_ray=32,
                    density_noise_std=0.0, log_interval=10 ** 9, seed=1, graph=True)
    run.model.train_precision = train_precision
    last = float(run.fit(epochs=3))
    assert run._graph_rays == 500 and last == last and last < 1.0


@pytest.mark.gpu
def test_philox_draws_stay_distinct_when_eager_steps_interleave_with_replays():
    """Trainer(graph=True, rng="philox") with an epoch tail: replays and eager tail steps alternate.  The Philox key
    of a launch is seed ^ (host offset + device counter); the host offset is constant (rank bits) and the device
    counter is the ONE launch sequence, advanced by exactly one behind every drawing launch of either kind — so the
    effective offsets of all steps are pairwise distinct (a host-side count beside it would let an eager step after
    the capture land on the key of the next replay).  Checked on the offsets AND on the fenceposts drawn."""
    from nerf_amd import trainer as T
    dev = torch.device("cuda:0")
    images, poses, focal = T.synthetic_scene(num_views=5, size=12, num_samples=16, device=dev)     # 576 training rays
    run = T.Trainer(images, poses, focal, batch_size=100, learning_rate=5e-4, num_samples_per_ray=16,
                    density_noise_std=0.5, log_interval=10 ** 9, seed=2, graph=True, rng="philox")
    model, offsets, kinds = run.model, [], []
    gen = torch.Generator().manual_seed(9)
    sizes = [100] * 7 + [76, 100, 100, 76, 100, 76, 76, 100]            # full batches replay (from the 6th), tails are eager
    for n in sizes:
        idx = torch.randint(0, len(run.dataset), (n,), generator=gen)
        seed, host = model._next_philox_state()
        before = int(model._philox_device_counter(dev).item())
        replays_before = run._graph is not None
        run.iteration += 1
        run.train_step(run.dataset.gather(idx.to(dev)))
        torch.cuda.synchronize()
        assert int(model._philox_counter.item()) == before + 1          # ONE advance per step, eager or replayed
        offsets.append(host + before)
        kinds.append("replay" if (replays_before and n == 100) else "eager")
    assert len(set(offsets)) == len(offsets) and offsets == sorted(offsets)
    assert kinds.count("replay") >= 4 and "eager" in kinds[8:]         # eager steps did run between replays
    # the draws themselves: fenceposts of the same rays under each step's effective offset differ pairwise
    o, d = torch.randn(8, 3, device=dev), torch.randn(8, 3, device=dev)
    drawn = [model.fenceposts_used(o, d, 16, randomly_sample=True, rng_state=(seed, off)) for off in offsets]
    for i in range(len(drawn)):
        for j in range(i + 1, len(drawn)):
            assert not torch.equal(drawn[i], drawn[j])


def test_load_scene_reads_the_tiny_nerf_layout(tmp_path):
    """tiny_nerf_data.npz (examples/, not shipped: .MISSING_LARGE_BLOBS) holds images [V,H,W,3],
    poses [V,4,4] and a scalar focal; the loader must take a file of that layout (any float dtype)."""
    from nerf_amd import trainer as T
    rng = np.random.default_rng(0)
    images = rng.random((5, 10, 12, 3)).astype(np.float64)
    poses = np.tile(np.eye(4, dtype=np.float32), (5, 1, 1))
    poses[:, :3, 3] = rng.normal(size=(5, 3))
    path = os.path.join(tmp_path, "tiny_nerf_like.npz")
    np.savez(path, images=images, poses=poses, focal=np.array(138.88887889922103))
    got_images, got_poses, focal = T.load_scene(path, "cpu")
    assert got_images.dtype == torch.float32 and got_images.shape == (5, 10, 12, 3)
    assert got_poses.dtype == torch.float32 and got_poses.shape == (5, 4, 4)
    assert isinstance(focal, float) and abs(focal - 138.88887889922103) < 1e-9
    assert torch.equal(got_images, torch.from_numpy(images.astype(np.float32)))
    # the trainer's split: last view held out (train_conditional_nerf.py:89-95)
    assert got_images[:-1].shape[0] == 4
