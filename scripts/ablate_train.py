"""Per-kernel times of the training step for -D variants of the library (GPU box, under rocprofv3; the
variants compute wrong gradients).  usage: [ABLATE_RAYS=4096] [ABLATE_PREC=fp32|f16x3]
python scripts/ablate_train.py NAME=DEF1,DEF2 ..."""
import csv, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerf_amd import build as B
variants = [("base", [])]
RAYS = os.environ.get("ABLATE_RAYS", "4096")
PREC = os.environ.get("ABLATE_PREC", "f16x3")
for arg in sys.argv[1:]:
    name, _, defs = arg.partition("=")
    variants.append((name, [d for d in defs.split(",") if d]))
for name, defs in variants:
    out = os.path.join(ROOT, "nerf_amd", "csrc", f"libnerf_hip_{name}.so")
    B.build(out=out, defines=defs)
    d = f"/tmp/abl_{name}"
    env = dict(os.environ, NERF_HIP_LIB=out, TMPDIR="/tmp")
    subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--",
                    "python3", os.path.join(ROOT, "scripts", "bench_train.py"), RAYS, PREC], env=env, cwd="/tmp",
                   capture_output=True, text=True)
    f = glob.glob(d + "/*/*kernel_stats.csv") + glob.glob(d + "/*kernel_stats.csv")
    row = {}
    for r in csv.DictReader(open(f[0])):
        for key in ("bwd_data", "render_fwd", "wgrad", "reduce"):
            if key in r["Name"]:
                row[key] = float(r["AverageNs"]) / 1e6
    print(f"{name:24s} " + "  ".join(f"{k} {v:.3f} ms" for k, v in row.items()), flush=True)
