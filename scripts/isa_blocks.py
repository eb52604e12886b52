"""Per-basic-block instruction mix of one kernel in the product build's assembly (nerf_amd/csrc/libnerf_hip.so.obj/*.s;
compile first: python -c "from nerf_amd import build; build.build()"): VALU instructions other than MFMAs, MFMAs, s_nop
and the most frequent opcodes of every block with more than `--min` VALU instructions or any MFMA — the static side of
the instruction diet of NOTES.md section R6d (in the fp32 kernels a VALU instruction is kernel time).
    python scripts/isa_blocks.py nerf_render.hip nerf_render_fwd_kernelILb0ELb0ELb0ELi16E [--min 60]"""
import collections
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
least = int(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else 60
if len(args) > 1 and args[-1].isdigit() and "--min" in sys.argv:
    args = args[:-1]
path = os.path.join(ROOT, "nerf_amd", "csrc", "libnerf_hip.so.obj", args[0] + ".s")
text = open(path).read()
names = [m.group(1) for m in re.finditer(r"^(_Z\S*):", text, re.M) if args[1] in m.group(1)]
assert names, f"no kernel matching {args[1]} in {path}"
start = text.index(names[0] + ":")
body = text[start:text.index(".Lfunc_end", start)].splitlines()
blocks, cur = [], ("entry", [])
for line in body:
    m = re.match(r"^(\.LBB\S+):", line)
    if m:
        blocks.append(cur)
        cur = (m.group(1), [])
    elif line.startswith("\t") and not line.startswith("\t.") and not line.startswith("\t;"):
        cur[1].append(line.strip())
blocks.append(cur)
print(names[0])
total_valu = total_mfma = 0
for name, ins in blocks:
    valu = [i.split()[0] for i in ins if i.startswith("v_") and not i.startswith("v_mfma")]
    mfma = sum(1 for i in ins if i.startswith("v_mfma"))
    total_valu += len(valu)
    total_mfma += mfma
    if len(valu) > least or mfma:
        top = ", ".join(f"{k}:{v}" for k, v in collections.Counter(valu).most_common(10))
        print(f"{name:12s} valu {len(valu):5d}  mfma {mfma:5d}  s_nop {sum(1 for i in ins if i.startswith('s_nop')):4d}  "
              f"branches {sum(1 for i in ins if 'branch' in i):2d} | {top}")
print(f"static total: {total_valu} VALU, {total_mfma} MFMA in {len(blocks)} blocks")
