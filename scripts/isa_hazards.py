"""CLI of the static ISA hazard scan (rules: nerf_amd/isa_scan.py) over nerf_amd/csrc/*.hip, compile only:
python scripts/isa_hazards.py [-DNAME ...]      exit code 1 when a rule is violated."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nerf_amd", "csrc")
sys.path.insert(0, ROOT)
from nerf_amd.build import flags_for           # the product's code-generation flags, per file
from nerf_amd.isa_scan import scan, WINDOW     # noqa: F401  (re-exported for the tests)


def compile_to_asm(src, out, defines=()):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", *flags_for(src), "-S", "--cuda-device-only",
           "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-Wno-unused-value",
           *[f"-D{d}" for d in defines], src, "-o", out]
    subprocess.run(cmd, check=True, capture_output=True)


def main(argv):
    defines = [a[2:] for a in argv if a.startswith("-D")]
    total = 0
    with tempfile.TemporaryDirectory() as tmp:
        for name in sorted(f for f in os.listdir(CSRC) if f.endswith(".hip")):
            out = os.path.join(tmp, name + ".s")
            compile_to_asm(os.path.join(CSRC, name), out, defines)
            hits = scan(out)
            print(f"{name}: {len(hits)} violation(s)")
            for kern, ln, s, rule, detail in hits:
                print(f"  [{rule}] {kern} line {ln}: {s}\n      {detail}")
            total += len(hits)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
