"""Interleaved A/B of library builds on the four training steps (both networks x both arithmetics), one subprocess per
measurement on one GPU:  python scripts/ab_train.py a.so b.so [rounds]      (paths relative to nerf_amd/csrc/)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [a for a in sys.argv[1:] if a.endswith(".so")]
rounds = int(sys.argv[-1]) if not sys.argv[-1].endswith(".so") else 3
CODE = """
import json, sys, torch
sys.path.insert(0, %r)
import bench
dev = torch.device("cuda:0")
out = {}
for p in ("fp32", "f16x3"):
    out["legacy_" + p] = bench.legacy_train_step_timing(dev, train_precision=p, steps=8)["ms_per_step"]
    t = bench.train_step_timing(dev, train_precision=p)
    out["main_" + p] = t["ms_per_step"]
    for k in ("forward", "data_gradient", "weight_gradient"):
        out["main_" + p + "_" + k] = t["kernels_ms"][k]
print(json.dumps(out))
""" % ROOT
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, NERF_HIP_LIB=os.path.join(ROOT, "nerf_amd", "csrc", l))
        out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
        lines = [x for x in out.stdout.splitlines() if x.startswith("{")]
        if not lines:
            print(out.stderr[-2000:])
            sys.exit(1)
        res[l].append(json.loads(lines[-1]))
keys = ["legacy_fp32", "legacy_f16x3", "main_fp32", "main_f16x3"]
keys += [f"main_{p}_{k}" for p in ("fp32", "f16x3") for k in ("forward", "data_gradient", "weight_gradient")]
for key in keys:
    for l, v in res.items():
        print(f"{key:28s} {l:24s} ms: " + " ".join(f"{x[key]:.3f}" for x in v) + f"   min {min(x[key] for x in v):.3f}")
