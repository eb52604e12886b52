"""Dump corrupted rows of the saved encoded inputs h (see scripts/determinism_stages.py) against the
oracle: which run is right, what the wrong values look like (hex), which lanes/features."""
import os, sys, struct
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nerf_amd import NeRF, _lib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import workspace_mirror as W
from oracle import nerf_oracle as O

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = NeRF().to(dev)
m.train_precision = prec
m.keep_workspace = True
params = {k: v.detach().cpu() for k, v in m.state_dict().items()}
cfg = O.default_config()
n, S = 4096, 64
g = torch.Generator().manual_seed(n)
o = torch.randn(n, 3, generator=g); d = torch.randn(n, 3, generator=g)
u = torch.rand(n, S, generator=g); noise = torch.randn(n, S - 1, 1, generator=g)
t = O.sample_t(params, n, S, u)
_, _, h_ref, _, _, _ = O.field(params, cfg, o, d, t)
lay = W.train_layout(n, S)
order = W.layer0_feature_order()
shown = 0
for r in range(6):
    m.render_rays(o.to(dev), d.to(dev), S, randomly_sample=True, density_noise_std=0.5, u=u.to(dev), noise=noise.to(dev))
    torch.cuda.synchronize()
    h = W.saved_h(m.last_workspace, n, S).cpu()          # reference feature order
    err = (h - h_ref).abs()
    err = torch.where(torch.isnan(err), torch.full_like(err, float("inf")), err)
    bad = err > 1e-4
    print(f"run {r}: {int(bad.any(-1).sum())} samples with a wrong feature, {int(bad.sum())} wrong values", flush=True)
    idx = torch.nonzero(bad)
    feats = {}
    for ray, smp, f in idx.tolist():
        feats[f] = feats.get(f, 0) + 1
    print("   wrong values per reference feature index:", dict(sorted(feats.items())))
    for ray, smp, f in idx[:12].tolist():
        kc = int(torch.nonzero(order == f)[0, 0])          # kernel column
        bits = struct.unpack("<I", struct.pack("<f", float(h[ray, smp, f])))[0]
        part, q = f // 48, f % 48
        print(f"   ray {ray} sample {smp} (chunk {smp // 16} j {smp % 16}) feature {f} = part {part} scale {q // 3} coord {q % 3}"
              f" kernel col {kc} (t {kc // 16} g {(kc % 16) // 4} r {kc % 4}): got {float(h[ray, smp, f])!r} (0x{bits:08x}) want {float(h_ref[ray, smp, f])!r}"
              f"   t0 {float(t[ray, smp]):.4f} o {o[ray].tolist()} d {d[ray].tolist()}")

# persistence analysis of the last run: per wave (block, wave-in-block), which of its 8 items and which
# lanes j show a wrong feature 44 (kernel col 44 = lane group 3, register X[2].x)
bad44 = bad[..., 44]
per_wave = {}
chunks = lay["chunks"]
for ray, smp in torch.nonzero(bad44).tolist():
    tile = ray * chunks + smp // 16
    grp, wave = tile // 4, tile % 4
    block, k = grp % 512, grp // 512
    per_wave.setdefault((block, wave), {}).setdefault(k, set()).add(smp % 16)
print(f"waves with a wrong feature 44: {len(per_wave)} of 2048")
for (block, wave), items in list(sorted(per_wave.items()))[:24]:
    print(f"   block {block:3d} wave {wave}: " + "  ".join(f"item {k}: j={sorted(js)}" for k, js in sorted(items.items())))

# full rows of a few corrupted samples: every feature of lane group 3 (kernel cols 16 t + 12 + r), got vs want
shown = 0
for ray, smp in torch.nonzero(bad.any(-1)).tolist():
    if shown >= 6:
        break
    shown += 1
    print(f"ray {ray} sample {smp}: lane group 3 features (kernel col: got / want)")
    for t_ in range(6):
        for r_ in range(4):
            kc = 16 * t_ + 12 + r_
            f = int(order[kc])
            gv, wv = float(h[ray, smp, f]), float(h_ref[ray, smp, f])
            q = 4 * t_ + r_
            mark = "  <-- WRONG" if abs(gv - wv) > 1e-6 * max(1.0, abs(wv)) and not (abs(wv) < 1e-30 and abs(gv) < 1e-30) else ""
            rel = abs(gv - wv) / max(abs(wv), 1e-37)
            print(f"     col {kc:2d} ({'sin' if q < 12 else 'cos'} p={q % 12:2d} scale {2 ** (8 + (q % 12) // 3)} coord {(q % 12) % 3}): {gv: .6e} / {wv: .6e}  rel {rel:.1e}{mark}")
