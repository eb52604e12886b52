"""Interleaved A/B of library builds on the narrow-network frame: python scripts/ab_narrow.py HIDDEN ENC a.so b.so ... [rounds]
(each run is scripts/bench_narrow.py HIDDEN ENC fp32 with NERF_HIP_LIB pointing at one build under nerf_amd/csrc/)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
hidden, enc = sys.argv[1], sys.argv[2]
libs = [a for a in sys.argv[3:] if a.endswith(".so")]
rounds = int(sys.argv[-1]) if not sys.argv[-1].endswith(".so") else 2
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, NERF_HIP_LIB=os.path.join(ROOT, "nerf_amd", "csrc", l))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "bench_narrow.py"), hidden, enc, "fp32"], env=env,
                             capture_output=True, text=True).stdout
        res[l].append(float(re.search(r"kernel\s+([0-9.]+) ms", out).group(1)))
for l, v in res.items():
    print(f"hidden {hidden} enc {enc}  {l:36s} kernel ms: " + " ".join(f"{x:.2f}" for x in v) + f"   min {min(v):.2f}", flush=True)
