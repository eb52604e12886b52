"""Static ISA statistics of one kernel of nerf_render.hip (compile here, no GPU):
python scripts/isa_stats.py [mangled-substring] [-DNAME ...]"""
import collections, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sub = "ILb0ELb1ELb0E"
defs = []
for a in sys.argv[1:]:
    if a.startswith("-D"): defs.append(a)
    else: sub = a
out = os.path.join(ROOT, "gpurun_out", "scratch", "isa.s")
os.makedirs(os.path.dirname(out), exist_ok=True)
sys.path.insert(0, ROOT)
from nerf_amd.build import flags_for
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", *flags_for("nerf_render.hip"), "-S", "--cuda-device-only",
                    "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "nerf_amd", "csrc"),
                    "-Wno-unused-value", "-Rpass-analysis=kernel-resource-usage", *defs,
                    os.path.join(ROOT, "nerf_amd", "csrc", "nerf_render.hip"), "-o", out],
                   capture_output=True, text=True)
lines = r.stderr.splitlines()
for i, l in enumerate(lines):
    if "Function Name" in l and sub in l:
        for m in lines[i:i + 12]:
            if any(k in m for k in ("VGPRs:", "ScratchSize", "SGPRs:", "Occupancy")):
                print(m.split("remark:")[-1].split("[-R")[0].strip())
txt = open(out).read()
import re
name = [m.group(1) for m in re.finditer(r"^(_ZN\S*nerf_render_fwd_kernel\S*):", txt, re.M) if sub in m.group(1)][0]
start = txt.index(name + ":"); end = txt.index(".Lfunc_end", start)
c = collections.Counter(); mf = 0; inloop = 0
total_mf = txt[start:end].count("\tv_mfma")
for l in txt[start:end].splitlines():
    l = l.strip()
    if not l or l.startswith((".", ";")) or l.endswith(":"): continue
    op = l.split()[0]; c[op] += 1
    if op.startswith("v_mfma"): mf += 1
    if "scratch_" in l and 0 < mf < total_mf: inloop += 1
valu = sum(n for k, n in c.items() if k.startswith("v_") and not k.startswith("v_mfma"))
print("mfma", total_mf, "valu", valu, "in-loop scratch ops", inloop, "s_nop", c["s_nop"], "ds_read_b128", c["ds_read_b128"],
      "waitcnt", c["s_waitcnt"], "v_mov", c["v_mov_b32_e32"], "pk_fma", c["v_pk_fma_f32"], "code bytes ~", 6 * sum(c.values()))
