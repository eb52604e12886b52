"""Interleaved A/B of library builds in one process group on one GPU: python scripts/ab_libs.py a.so b.so [rounds]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [a for a in sys.argv[1:] if a.endswith(".so")]
rounds = int(sys.argv[-1]) if not sys.argv[-1].endswith(".so") else 3
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, NERF_HIP_LIB=os.path.join(ROOT, "nerf_amd", "csrc", l))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2",
                              "--no-cpu-baseline"], env=env, capture_output=True, text=True).stdout
        j = json.loads([x for x in out.splitlines() if x.startswith("{")][-1])
        res[l].append(j["roofline"]["kernel_ms"])
for l, v in res.items():
    print(f"{l:32s} kernel ms: " + " ".join(f"{x:.2f}" for x in v) + f"   min {min(v):.2f}")
