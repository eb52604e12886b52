"""Interleaved A/B of library builds on what scripts/ab_libs.py (the headline frame) does not cover: the LEGACY
network's 800x800x128 frame in both arithmetics, its split-precision training step, and the main network's
training step in both arithmetics; one subprocess per measurement on one GPU:
python scripts/ab_legacy.py a.so b.so [rounds]      (paths relative to nerf_amd/csrc/)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [a for a in sys.argv[1:] if a.endswith(".so")]
rounds = int(sys.argv[-1]) if not sys.argv[-1].endswith(".so") else 3
CODE = """
import json, sys, torch
sys.path.insert(0, %r)
import bench
dev = torch.device("cuda:0")
r = bench.legacy_workload_timing(dev, steps=3, warmup=1)
t = bench.legacy_train_step_timing(dev, train_precision="f16x3")
m = {p: bench.train_step_timing(dev, train_precision=p)["ms_per_step"] for p in ("fp32", "f16x3")}
print(json.dumps({"fp32": r["kernel_ms"], "f16x3": r["other_precision"]["kernel_ms"], "train_f16x3": t["ms_per_step"],
                  "train_f16x3_fwd_kernel": t["kernels_ms"]["forward"], "train_f16x3_dgrad_kernel": t["kernels_ms"]["data_gradient"],
                  "main_train_fp32": m["fp32"], "main_train_f16x3": m["f16x3"]}))
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
for l, v in res.items():
    for key in ("fp32", "f16x3", "train_f16x3", "train_f16x3_fwd_kernel", "train_f16x3_dgrad_kernel", "main_train_fp32",
                "main_train_f16x3"):
        print(f"{l:28s} {key:24s} ms: " + " ".join(f"{x[key]:.3f}" for x in v) + f"   min {min(x[key] for x in v):.3f}")
