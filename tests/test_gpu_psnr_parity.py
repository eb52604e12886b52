"""The 300-step version of tests/psnr_parity.py (the 2,000-step study whose record is profiles/r06_psnr_parity.json):
the HIP Trainer and the oracle train the same network from the same parameters on the same captured ray indices,
stratified draws and density noise (train_conditional_nerf.py:115-153), and the held-out PSNR (:152-153) is compared
at steps 100, 200 and 300 — beside tests/test_gpu_training_trajectory.py, which holds the first 40 steps loss by
loss.  A training trajectory amplifies rounding differences exponentially: in the study the oracle on 1, 2, 4 and 8
threads — the SAME algorithm in four summation orders — is 1e-7 apart in loss at step 10 and 5e-4 at step 100, and its
six pairs differ in held-out PSNR by up to 0.003 dB at step 100, 0.03 at 200, 0.04 at 300 and 0.4 at 2,000.  "Within
0.01 dB" cannot be asked of step 300 of ANY two implementations; the bound here is 0.1 dB (2.5 x the oracle's own
largest pair at that length; measured: fp32 0.015, f16x3 0.054), and a graph-replayed run must reproduce its eager twin
bit for bit."""
import json
import os

import pytest
import torch

import psnr_parity as P

pytestmark = pytest.mark.gpu


def test_three_hundred_steps_held_out_psnr(tmp_path):
    out = str(tmp_path)
    P.make_scene(out)
    # the oracle's rays are the Trainer's rays: the host-side decode against the on-device gather, bit for bit
    from nerf_amd import trainer as T
    dev = torch.device("cuda:0")
    images, poses, focal, _ = P.load_scene(out)
    data = T.PixelRayDataset(images[:-1].to(dev), torch.zeros(P.VIEWS - 1, P.SIZE, P.SIZE, dtype=torch.int64, device=dev),
                             poses[:-1].to(dev), focal)
    rays_o, rays_d, pixels = P.host_batches(images, poses, focal)
    idx = torch.randint(0, len(data), (512,), generator=torch.Generator().manual_seed(1))
    b = data.gather(idx.to(dev))
    assert torch.equal(b["rays_o"].cpu(), rays_o[idx]) and torch.equal(b["rays_d"].cpu(), rays_d[idx])
    assert torch.equal(b["pixels"].cpu(), pixels[idx])

    steps, every = 300, 100
    ref = P.oracle_run(out, "cpu_mt", min(16, os.cpu_count() or 1), steps, every, 5, False)
    assert ref["loss"][-1] < 0.5 * ref["loss"][0]                       # it trains
    done = {}
    for tag, precision, graph in (("hip_fp32", "fp32", False), ("hip_f16x3", "f16x3", False),
                                  ("hip_f16x3_graph", "f16x3", True)):
        run = P.hip_run(out, tag, steps, every, 5, precision, graph, "captured")
        assert run["replayed"] == graph
        first = max(abs(a - b) / b for a, b in zip(run["loss"][:40], ref["loss"][:40]))
        worst = max(abs(run["psnr"][s] - ref["psnr"][s]) for s in ref["psnr"])
        print(f"[{tag}] first-40-steps loss deviation {first:.2e}; held-out PSNR "
              + ", ".join(f"{s}: {run['psnr'][s]:.4f} / {ref['psnr'][s]:.4f}" for s in sorted(ref["psnr"]))
              + f" dB (HIP / oracle); max |d| {worst:.4f} dB")
        assert first <= 2e-3
        assert worst <= 0.1, (tag, run["psnr"], ref["psnr"])
        done[tag] = run
    assert done["hip_f16x3_graph"]["psnr"] == done["hip_f16x3"]["psnr"]         # replays are the eager launches, bit for bit
    assert done["hip_f16x3_graph"]["loss"] == done["hip_f16x3"]["loss"]
    P.merge(out, os.path.join(out, "record.json"))
    with open(os.path.join(out, "record.json")) as f:
        record = json.load(f)
    assert set(record["captured_draws"]["max_abs_dpsnr"]) == {"hip_fp32", "hip_f16x3", "hip_f16x3_graph"}
