"""Diagnostic (not collected by pytest): per-tensor ABSOLUTE error of the training gradient of both arithmetics against
the fp64 oracle on one 256-ray batch of the parity study's scene — the first Adam step turns a gradient element into
lr * g / (|g| + 1e-8), so it is the absolute error at the 1e-8 scale, not the error relative to a tensor's largest
element, that decides how far the second step's loss lands from the oracle's (tests/psnr_parity.py).
    python tests/diag_grad_error.py [scene dir]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import psnr_parity as P                                     # noqa: E402
from oracle import nerf_oracle as O                         # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r6_psnr")
    hidden = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    if not os.path.exists(os.path.join(out, "scene.npz")):
        P.make_scene(out)
    from nerf_amd import NeRF
    dev = torch.device("cuda:0")
    images, poses, focal, params0 = P.load_scene(out)
    cfg = dict(O.default_config(), focal_length=focal, hidden_size=hidden)
    if hidden != 256 or scale != 1.0:
        params0 = O.init_params(seed=0, cfg=cfg)
        for slot in O.LINEAR_IDS:
            params0[f"prediction_heads.{slot}.weight"] = params0[f"prediction_heads.{slot}.weight"] * scale
    width = 128 if hidden <= 128 else 256
    rays_o, rays_d, pixels = P.host_batches(images, poses, focal)
    gen = torch.Generator().manual_seed(5)
    idx, u, noise = P.captured_step(gen, rays_o.shape[0])

    def oracle(dtype):
        p = {k: v.to(dtype).clone().requires_grad_(k.startswith("prediction")) for k, v in params0.items()}
        loss = O.training_loss(p, cfg, rays_o[idx].to(dtype), rays_d[idx].to(dtype), P.SAMPLES, pixels[idx].to(dtype),
                               u.to(dtype), noise.to(dtype), P.NOISE_STD)
        loss.backward()
        return float(loss), {k: v.grad.double() for k, v in p.items() if v.grad is not None}

    def oracle_dy(dtype):
        """dL/dy of every hidden layer's pre-LayerNorm output y_L [n, S-1, 256], L = 0..4 (fp64 autograd)."""
        import torch.nn.functional as F
        p = {k: v.to(dtype) for k, v in params0.items()}
        t = O.sample_t(p, 256, P.SAMPLES, u.to(dtype))
        base_radius = 1 / ((3 ** 0.5) * cfg["focal_length"])
        means, covs = O.frustum_gaussians(rays_o[idx].to(dtype), rays_d[idx].to(dtype), t, base_radius)
        h = O.ipe_features(means, covs, -4, cfg["encoding_size"] // 2 - 4)
        ys = []
        x = F.linear(h, p["prediction_heads.0.weight"], p["prediction_heads.0.bias"])
        for norm_slot, lin_slot in zip(O.NORM_IDS, O.LINEAR_IDS[1:]):
            x.requires_grad_(True)
            x.retain_grad()
            ys.append(x)
            x = F.relu(F.layer_norm(x, (x.shape[-1],), p[f"prediction_heads.{norm_slot}.weight"],
                                    p[f"prediction_heads.{norm_slot}.bias"], 1e-5))
            x = F.linear(x, p[f"prediction_heads.{lin_slot}.weight"], p[f"prediction_heads.{lin_slot}.bias"])
        density, color, _ = x.split([1, 3, 50], dim=-1)
        density = density + noise.to(dtype) * P.NOISE_STD
        w = O.composite_weights(means, density)
        rgb = (w * torch.sigmoid(color)).sum(dim=-2)
        ((rgb.unsqueeze(1) - pixels[idx].to(dtype).unsqueeze(1)) ** 2).mean().backward()
        return [y.grad for y in ys]

    dy64 = oracle_dy(torch.float64)
    dy32 = oracle_dy(torch.float32)
    l64, g64 = oracle(torch.float64)
    l32, g32 = oracle(torch.float32)
    runs = {"oracle fp32": g32}
    for prec in ("fp32", "f16x3"):
        model = NeRF(focal_length=focal, hidden_size=hidden)
        model.load_state_dict(params0)
        model = model.to(dev)
        model.train_precision = prec
        model.keep_workspace = True
        pix, _ = model.render_rays(rays_o[idx].to(dev), rays_d[idx].to(dev), P.SAMPLES, randomly_sample=True,
                                   density_noise_std=P.NOISE_STD, u=u.to(dev), noise=noise.to(dev))
        loss = ((pix - pixels[idx].to(dev).unsqueeze(1)) ** 2).mean()
        loss.backward()
        runs["hip " + prec] = {k: p.grad.detach().cpu().double() for k, p in model.named_parameters()}
        import workspace_mirror as W
        lay = W.train_layout(256, P.SAMPLES, width)
        # saved x_hat / gates against the fp64 oracle's
        with torch.no_grad():
            p64 = {k: v.double() for k, v in params0.items()}
            t64 = O.sample_t(p64, 256, P.SAMPLES, u.double())
            m64, c64 = O.frustum_gaussians(rays_o[idx].double(), rays_d[idx].double(), t64, 1 / ((3 ** 0.5) * cfg["focal_length"]))
            h64 = O.ipe_features(m64, c64, -4, cfg["encoding_size"] // 2 - 4)
            _, xh64, _ = O.mlp_stages(p64, h64)
        for L, slot in enumerate((1, 4, 7, 10, 13)):
            xk = W.saved_xhat(model.last_workspace, L, 256, P.SAMPLES, width)[..., :hidden].cpu().double()
            ga, be = params0[f"prediction_heads.{slot}.weight"].double(), params0[f"prediction_heads.{slot}.bias"].double()
            zk, z64 = xk * ga + be, xh64[L] * ga + be
            flips = (zk > 0) != (z64 > 0)
            print(f"   x_hat[{L}] hip {prec}: rms error {float((xk - xh64[L]).pow(2).mean().sqrt()):.2e}  max {float((xk - xh64[L]).abs().max()):.2e}; "
                  f"{int(flips.sum())} gates differ from the fp64 oracle's, largest |z| among them "
                  f"{float(z64[flips].abs().max()) if flips.any() else 0.0:.2e}")
        # network outputs of the training forward (padded tile rows) and dL/d(out) against the fp64 oracle
        with torch.no_grad():
            raw64 = O.mlp(p64, h64)
        got_out = W._rows(model.last_workspace, lay, lay["out"], 64, 256, P.SAMPLES)     # stored as tiles: see below
        tiles = model.last_workspace[lay["out"]:lay["out"] + lay["mp"] * 64].view(-1, 4, 4, 16, 4)     # [tile][T][g][s][r]
        rows = tiles.permute(0, 3, 1, 2, 4).reshape(lay["mp"], 64).view(lay["mp"] // (lay["chunks"] * 16), lay["chunks"] * 16, 64)
        out_k = rows[:256, :P.SAMPLES - 1, :raw64.shape[-1]].cpu().double()
        print(f"   out hip {prec}: max|out| {float(raw64.abs().max()):.2e}  max|e| {float((out_k - raw64).abs().max()):.2e}  rms e "
              f"{float((out_k - raw64).pow(2).mean().sqrt()):.2e}")
        for L in range(5):
            got = W._rows(model.last_workspace, lay, lay["dy"][L], width, 256, P.SAMPLES)[..., :hidden].cpu().double()
            e, e32 = got - dy64[L], dy32[L].double() - dy64[L]
            per_sample = e.mean(-1)                       # an additive per-sample offset shows here
            print(f"   dY[{L}] hip {prec}: max|dy| {float(dy64[L].abs().max()):.2e}  max|e| {float(e.abs().max()):.2e}  rms e "
                  f"{float(e.pow(2).mean().sqrt()):.2e} (oracle fp32: {float(e32.pow(2).mean().sqrt()):.2e})  rms of the "
                  f"per-sample MEAN error {float(per_sample.pow(2).mean().sqrt()):.2e} (oracle fp32: "
                  f"{float(e32.mean(-1).pow(2).mean().sqrt()):.2e})")
        print(f"hip {prec}: loss {float(loss):.9f} (oracle fp64 {l64:.9f}, fp32 {l32:.9f})")
    print(f"{'tensor':32s} {'max|g|':>9s} " + " ".join(f"{n + ' max|e|':>22s} {'rms e':>9s} {'adam flips':>10s}" for n in runs))
    for k in g64:
        ref = g64[k]
        row = f"{k:32s} {float(ref.abs().max()):9.2e} "
        for n, g in runs.items():
            e = g[k] - ref
            upd_ref = ref / (ref.abs() + 1e-8)
            upd = g[k] / (g[k].abs() + 1e-8)
            moved = float(((upd - upd_ref).abs() > 0.1).double().mean())      # share of elements whose first Adam update moves by > 10 %
            row += f"{float(e.abs().max()):22.2e} {float(e.pow(2).mean().sqrt()):9.2e} {moved:10.2e} "
        print(row)


if __name__ == "__main__":
    main()
