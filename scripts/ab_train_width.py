"""Interleaved A/B of library builds on one training step width: python scripts/ab_train_width.py HIDDEN ENC PRECISION a.so b.so [rounds]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
hidden, enc, prec = sys.argv[1:4]
libs = [a for a in sys.argv[4:] if a.endswith(".so")]
rounds = int(sys.argv[-1]) if not sys.argv[-1].endswith(".so") else 2
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, NERF_HIP_LIB=os.path.join(ROOT, "nerf_amd", "csrc", l))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "bench_train_width.py"), hidden, enc, prec], env=env,
                             capture_output=True, text=True).stdout
        res[l].append(json.loads([x for x in out.splitlines() if x.startswith("{")][-1]))
for key in ("ms_per_step", "forward", "data_gradient", "weight_gradient"):
    for l, v in res.items():
        vals = [x[key] if key == "ms_per_step" else x["kernels_ms"][key] for x in v]
        print(f"hidden {hidden} {prec:5s} {key:16s} {l:24s} ms: " + " ".join(f"{x:.4f}" for x in vals) + f"   min {min(vals):.4f}", flush=True)
