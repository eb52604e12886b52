"""Build the A/B partner of profiles/r06_a_legacy_fwd_ab.log: the round-6 tree with the LEGACY split-precision training
forward of round 5 (commit 5e3b406: density head on the fp32 tiles, per-lane 64-bit save pointers) — as
nerf_amd/csrc/libnerf_hip_legacy_r5.so, an experiment build the default loader refuses (select it with NERF_HIP_LIB;
scripts/ab_legacy.py does).  The ABI moved between the rounds, so the old library itself cannot be loaded: the old
nerf_legacy.hip is compiled against this tree's headers, with the one line of nerf_fused.h it relies on put back
(per-lane save pointers in the Linear -> ReLU -> LayerNorm branch).
    python scripts/experiments/build_legacy_fwd_r5.py && python scripts/ab_legacy.py libnerf_hip_legacy_r5.so libnerf_hip.so 2"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import nerf_amd.build as b  # noqa: E402

tmp = "/tmp/ab_csrc_legacy_r5"
shutil.rmtree(tmp, ignore_errors=True)
shutil.copytree(os.path.join(ROOT, "nerf_amd", "csrc"), tmp, ignore=shutil.ignore_patterns("*.so*", "*.obj"))
with open(os.path.join(tmp, "nerf_legacy.hip"), "w") as f:
    f.write(subprocess.run(["git", "-C", ROOT, "show", "5e3b406:nerf_amd/csrc/nerf_legacy.hip"], check=True,
                           capture_output=True, text=True).stdout)
path = os.path.join(tmp, "nerf_fused.h")
text = open(path).read()
for new, old in (("n.save_row + row_lane_offset() + T * kTileT", "n.save_row + T * kTileT"),
                 ("save_rstd[stat_lane_offset()] = rstd * save_scale;", "*save_rstd = rstd * save_scale;"),
                 ("save_shift[stat_lane_offset()] = n.shift;", "*save_shift = n.shift;")):
    assert new in text, new
    text = text.replace(new, old)
open(path, "w").write(text)
b.CSRC = tmp
print(b.build(out=os.path.join(ROOT, "nerf_amd", "csrc", "libnerf_hip_legacy_r5.so"),
              defines=("NERF_HIP_EXPERIMENT=legacy_fwd_r5",), force=True))
