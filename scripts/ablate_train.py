"""Training-step experiments: build variants of libnerf_hip.so with -D switches and time
scripts/bench_train.py with each (run on the GPU box; the variants compute wrong gradients).
usage: python scripts/ablate_train.py [--prec fp32|f16x3] NAME=DEF1,DEF2 ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerf_amd import build as B
argv = sys.argv[1:]
prec = "f16x3"
if argv and argv[0] == "--prec":
    prec, argv = argv[1], argv[2:]
variants = [("base", [])]
for arg in argv:
    name, _, defs = arg.partition("=")
    variants.append((name, [d for d in defs.split(",") if d]))
for name, defs in variants:
    out = os.path.join(ROOT, "nerf_amd", "csrc", f"libnerf_hip_{name}.so")
    B.build(out=out, defines=defs)
    env = dict(os.environ, NERF_HIP_LIB=out)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "bench_train.py"), "4096", prec], env=env,
                       capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("train step")]
    print(f"{name:24s}", line[-1] if line else "FAILED " + r.stderr[-400:], flush=True)
