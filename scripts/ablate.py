"""Kernel experiments: build variants of libnerf_hip.so with -D switches and time the headline
frame with each (run on the GPU box).  usage: python scripts/ablate.py NAME=DEF1,DEF2 ..."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerf_amd import build as B
variants = [("base", [])]
for arg in sys.argv[1:]:
    name, _, defs = arg.partition("=")
    variants.append((name, [d for d in defs.split(",") if d]))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for name, defs in variants:
    out = os.path.join(ROOT, "nerf_amd", "csrc", f"libnerf_hip_{name}.so")
    B.build(out=out, defines=defs)
    env = dict(os.environ, NERF_HIP_LIB=out)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(name, "FAILED", r.stderr[-500:]); continue
    j = json.loads(line[-1])
    print(f"{name:28s} {j['ms_per_step']:8.2f} ms  {j['value']:.4e} samples/s  frac {j['roofline']['frac']:.4f}", flush=True)
