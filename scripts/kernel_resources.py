"""Registers / scratch / occupancy of the kernels of one source file (compile only, no GPU):
python scripts/kernel_resources.py nerf_backward.hip [name-substring] [--csrc DIR] [-DNAME ...]
Also counts scratch accesses that sit between a kernel's first and last MFMA (spills inside the loops)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerf_amd.build import flags_for  # noqa: E402


def resources(source, csrc=None, defines=()):
    csrc = csrc or os.path.join(ROOT, "nerf_amd", "csrc")
    out = os.path.join("/tmp", f"kres_{os.getpid()}.s")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", *flags_for(source), "-S", "--cuda-device-only",
                        "-I", os.path.join(ROOT, "include"), "-I", csrc, "-Wno-unused-value",
                        "-Rpass-analysis=kernel-resource-usage", *defines, os.path.join(csrc, source), "-o", out],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-3000:])
    txt = open(out).read()
    os.remove(out)
    rows, lines = [], r.stderr.splitlines()
    for i, l in enumerate(lines):
        if "Function Name:" not in l:
            continue
        name = l.split("Function Name:")[1].split("[-R")[0].strip()
        info = {"name": name}
        for m in lines[i + 1:i + 14]:
            for key, tag in (("VGPRs:", "vgpr"), ("AGPRs:", "agpr"), ("ScratchSize [bytes/lane]:", "scratch"),
                             ("SGPRs:", "sgpr"), ("Occupancy [waves/SIMD]:", "occupancy"), ("LDS Size [bytes/block]:", "lds")):
                if key in m and "Spill" not in m.split(key)[0][-8:]:
                    info.setdefault(tag, int(m.split(key)[1].split("[-R")[0].strip()))
        start = txt.find("\n" + name + ":")
        if start >= 0:
            body = txt[start:txt.index(".Lfunc_end", start)]
            ops = [x.strip() for x in body.splitlines()]
            mf = [i for i, x in enumerate(ops) if x.startswith("v_mfma")]
            info["mfma"] = len(mf)
            info["scratch_instructions"] = sum(1 for x in ops if x.startswith("scratch_"))
            info["scratch_in_loops"] = sum(1 for i, x in enumerate(ops) if x.startswith("scratch_") and mf and mf[0] < i < mf[-1])
            # inside an MFMA loop proper: an MFMA within 40 instructions on BOTH sides (phases between two loops,
            # LayerNorm / encoding / compositing, are hundreds of instructions long)
            import bisect
            near = 0
            for i, x in enumerate(ops):
                if x.startswith("scratch_") and mf:
                    k = bisect.bisect_left(mf, i)
                    if 0 < k < len(mf) and i - mf[k - 1] <= 40 and mf[k] - i <= 40:
                        near += 1
            info["scratch_between_mfmas"] = near
        rows.append(info)
    return rows


if __name__ == "__main__":
    args = [a for a in sys.argv[1:]]
    csrc = None
    if "--csrc" in args:
        i = args.index("--csrc")
        csrc = args[i + 1]
        del args[i:i + 2]
    defs = [a for a in args if a.startswith("-D")]
    rest = [a for a in args if not a.startswith("-D")]
    for row in resources(rest[0], csrc, defs):
        if len(rest) > 1 and rest[1] not in row["name"]:
            continue
        short = re.sub(r"^_ZN?\d*_GLOBAL__N_1", "", row["name"])[:70]
        print(f"{short:70s} vgpr {row.get('vgpr')} agpr {row.get('agpr')} sgpr {row.get('sgpr')} scratch {row.get('scratch')} "
              f"(instructions: {row.get('scratch_instructions')}, first..last MFMA: {row.get('scratch_in_loops')}, between MFMAs: {row.get('scratch_between_mfmas')}) waves/SIMD {row.get('occupancy')} mfma {row.get('mfma')}")
