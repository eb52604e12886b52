"""Build the kernels of a committed revision against nothing but themselves, as an A/B partner for the working tree:
    python scripts/experiments/build_rev.py [REV=HEAD] [NAME=head]   ->  nerf_amd/csrc/libnerf_hip_NAME.so
(an experiment build: the default loader refuses it; scripts/ab_libs.py / ab_narrow.py / ab_train.py select it through
NERF_HIP_LIB).  The ABI of REV must be the working tree's."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import nerf_amd.build as b  # noqa: E402

rev = sys.argv[1] if len(sys.argv) > 1 else "HEAD"
name = sys.argv[2] if len(sys.argv) > 2 else "head"
tmp = "/tmp/ab_csrc_" + name
shutil.rmtree(tmp, ignore_errors=True)
os.makedirs(tmp)
files = subprocess.run(["git", "-C", ROOT, "ls-tree", "--name-only", rev, "nerf_amd/csrc/"], check=True, capture_output=True,
                       text=True).stdout.split()
for f in files:
    data = subprocess.run(["git", "-C", ROOT, "show", f"{rev}:{f}"], check=True, capture_output=True).stdout
    with open(os.path.join(tmp, os.path.basename(f)), "wb") as out:
        out.write(data)
b.CSRC = tmp
print(b.build(out=os.path.join(ROOT, "nerf_amd", "csrc", f"libnerf_hip_{name}.so"), defines=(f"NERF_HIP_EXPERIMENT={name}",), force=True))
