"""Round 5: the legacy weight gradient's split-K cap (kLegacyMaxSplits) swept over values that are NOT powers of two.
14 jobs (9 of 256 x 256, 5 light) x splits workgroups on 256 CUs, one workgroup per CU: the heavy jobs' workgroup count
decides how full the last round is.  Builds one library per value from a copy of csrc/ and prints the kernel times
(rocprofv3 --kernel-trace of scripts/bench_train_legacy.py) and the step time.
usage (GPU box): python scripts/experiments/sweep_legacy_splits.py [--build-only] [values ...]"""
import csv
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
VALUES = [int(a) for a in sys.argv[1:] if a.isdigit()] or [64, 56, 60, 72, 80]


def lib_of(v):
    return os.path.join(ROOT, "nerf_amd", "csrc", f"libnerf_hip_lsplit_{v}.so")


def build(v):
    from nerf_amd import build as B
    work = tempfile.mkdtemp(prefix="lsplit_")
    csrc = os.path.join(work, "csrc")
    shutil.copytree(os.path.join(ROOT, "nerf_amd", "csrc"), csrc, ignore=shutil.ignore_patterns("*.so*"))
    p = os.path.join(csrc, "nerf_legacy_backward.hip")
    s = open(p).read()
    assert s.count("constexpr int kLegacyMaxSplits = 64;") == 1
    open(p, "w").write(s.replace("constexpr int kLegacyMaxSplits = 64;", f"constexpr int kLegacyMaxSplits = {v};"))
    B.CSRC = csrc
    return B.build(out=lib_of(v), defines=[f"NERF_HIP_EXPERIMENT=lsplit_{v}"], force=True)


def main():
    if "--build-only" in sys.argv:
        for v in VALUES:
            print(build(v))
        return
    for prec in ("f16x3", "fp32"):
        for v in VALUES * 2:
            if not os.path.exists(lib_of(v)):
                build(v)
            out = os.path.join(ROOT, "gpurun_out", "lsplit", f"{prec}_{v}")
            shutil.rmtree(out, ignore_errors=True)
            r = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", out, "-o", "run", "--",
                                "python3", os.path.join(ROOT, "scripts", "bench_train_legacy.py"), "4096", "64", prec],
                               env=dict(os.environ, NERF_HIP_LIB=lib_of(v), TMPDIR="/tmp"), capture_output=True, text=True, cwd="/tmp")
            step = re.search(r"([\d.]+) ms/step", r.stdout)
            row = {}
            for x in csv.DictReader(open(os.path.join(out, "run_kernel_stats.csv"))):
                for key in ("wgrad", "grad_reduce", "bwd_data", "legacy_fwd"):
                    if key in x["Name"]:
                        row[key] = float(x["AverageNs"]) / 1e3
            print(f"{prec:5s} splits {v:3d}: wgrad {row.get('wgrad', 0):7.1f} us  reduce {row.get('grad_reduce', 0):5.1f} us  "
                  f"dgrad {row.get('bwd_data', 0):7.1f}  fwd {row.get('legacy_fwd', 0):7.1f}  step (under the profiler) {step.group(1) if step else '?'} ms", flush=True)


if __name__ == "__main__":
    main()
