"""Where does a run-to-run difference of the training forward first appear?  (GPU box)

Repeats the training forward of one batch, keeps every run's workspace (NeRF.keep_workspace) and
compares the saved stage tensors with the first run's, stage by stage in dataflow order: encoded
inputs h, then per hidden layer the normalised activations x_hat and 1/std, then the padded network
outputs.  Prints, per stage, how many padded samples differ and the first few (ray, chunk, sample)
rows — the first stage with a difference names the step of the kernel that went wrong.

python scripts/determinism_stages.py [fp32|f16x3] [reps]      (NERF_HIP_LIB selects another build)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_amd import NeRF, _lib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import workspace_mirror as W

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = NeRF().to(dev)
m.train_precision = prec
m.keep_workspace = True
with torch.no_grad():
    for i in (0, 3, 6, 9, 12, 15):
        m.prediction_heads[i].weight.mul_(2.0)
print(f"library {_lib.LIB_PATH}  flags {_lib.build_flags()}  train forward {prec}", flush=True)


def stages(ws, n, S):
    lay = W.train_layout(n, S)
    mp = lay["mp"]
    out = [("h", ws[lay["h"]:lay["h"] + mp * 96].view(mp, 96))]
    for L in range(5):
        out.append((f"x_hat[{L}]", ws[lay["xhat"][L]:lay["xhat"][L] + mp * 256].view(mp, 256)))
        out.append((f"rstd[{L}]", ws[lay["rstd"][L]:lay["rstd"][L] + mp].view(mp, 1)))
    out.append(("out", ws[lay["out"]:lay["out"] + mp * 64].view(mp // 16, 1024)))
    return lay, out


total_bad = 0
for n, S in ((256, 100), (4096, 64)):
    g = torch.Generator().manual_seed(n)
    o = torch.randn(n, 3, generator=g).to(dev)
    d = torch.randn(n, 3, generator=g).to(dev)
    u = torch.rand(n, S, generator=g).to(dev)
    noise = torch.randn(n, S - 1, 1, generator=g).to(dev)
    first = first_out = None
    for r in range(reps):
        rgb, seg = m.render_rays(o, d, S, randomly_sample=True, density_noise_std=0.5, u=u, noise=noise)
        torch.cuda.synchronize()
        ws = m.last_workspace.clone()
        cur_out = torch.cat([rgb.detach().flatten(), seg.detach().flatten()])
        if first is None:
            first, first_out = ws, cur_out
            continue
        lay, cur = stages(ws, n, S)
        _, ref = stages(first, n, S)
        line = []
        for (name, a), (_, b) in zip(cur, ref):
            bad = (a != b).any(dim=1)
            k = int(bad.sum())
            if k:
                rows = torch.nonzero(bad)[:4, 0].tolist()
                where = []
                for row in rows:
                    if name == "out":
                        where.append(f"tile {row} = ray {row // lay['chunks']} chunk {row % lay['chunks']}")
                    else:
                        tile = row // 16
                        where.append(f"ray {tile // lay['chunks']} chunk {tile % lay['chunks']} j {row % 16}"
                                     f" cols {torch.nonzero(a[row] != b[row])[:6, 0].tolist()}")
                diff = (a.double() - b.double()).abs()
                diff = torch.where(torch.isnan(diff), torch.full_like(diff, float("inf")), diff)
                r0 = rows[0]
                c0 = int(torch.nonzero(a[r0] != b[r0])[0, 0])
                line.append(f"{name}: {k} rows, max |diff| {float(diff.max()):.3e}, first: this {float(a[r0, c0])!r} "
                            f"first-run {float(b[r0, c0])!r} ({'; '.join(where)})")
        nbad = int((cur_out != first_out).sum())
        total_bad += nbad + len(line)
        print(f"{n} x {S} run {r}: outputs differing {nbad}" + ("" if not line else "\n    " + "\n    ".join(line)),
              flush=True)
print("TOTAL differences:", total_bad)
