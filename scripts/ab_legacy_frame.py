"""Interleaved A/B of library builds on the legacy 8x256 network: the 800x800x128 frame (fp32 kernel ms) and the fp32 / f16x3
training steps:  python scripts/ab_legacy_frame.py a.so b.so [rounds]      (paths relative to nerf_amd/csrc/)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [a for a in sys.argv[1:] if a.endswith(".so")]
rounds = int(sys.argv[-1]) if not sys.argv[-1].endswith(".so") else 2
CODE = """
import json, sys, torch
sys.path.insert(0, %r)
import bench
dev = torch.device("cuda:0")
out = {}
w = bench.legacy_workload_timing(dev, steps=3)
out["frame_fp32_ms"] = w["ms_per_step"]
out["frame_f16x3_ms"] = w["other_precision"]["ms_per_step"]
for p in ("fp32", "f16x3"):
    t = bench.legacy_train_step_timing(dev, train_precision=p)
    out["step_" + p] = t["ms_per_step"]
    out["step_" + p + "_forward"] = t["kernels_ms"]["forward"]
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
for key in res[libs[0]][0]:
    for l, v in res.items():
        print(f"{key:22s} {l:24s} ms: " + " ".join(f"{x[key]:.3f}" for x in v) + f"   min {min(x[key] for x in v):.3f}")
