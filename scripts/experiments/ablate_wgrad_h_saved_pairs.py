"""Round 5, item 3 of the round-4 verdict — "the data gradient saves dY as the scaled f16 pair, the weight gradient
loads pairs instead of converting" — as a TIMING-ONLY upper bound for the weight-gradient side (WRONG results).
If dY arrived in HBM as (hi | lo) f16 words (same 4 bytes per element, same DMA, same LDS reads), the A operand of a
k-step would need, per value pair, two byte-permutes (gather the two hi halves, gather the two lo halves) instead of
today's scale x2 + pkrtz + residual x2 + pkrtz; the bias gradient (column sums of dY, today a by-product of the
conversion) would have to come from somewhere else.  Variants of a copy of csrc/:
  base      the product
  permA     A operands = two v_perm_b32 per pair of raw words, bias-gradient adds kept (they would need a third
            conversion back to fp32 in reality: this is the optimistic form)
  permA_nb  the same without the bias-gradient adds (the bias gradient moved into the data gradient)
Prints the kernels' average durations from `rocprofv3 --kernel-trace` of scripts/bench_train.py.
usage (GPU box): python scripts/experiments/ablate_wgrad_h_saved_pairs.py"""
import csv
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

OLD = '''                    if (level == 0) {
                        float x0 = raw[op][2 * pp], x1 = raw[op][2 * pp + 1];
                        if (op > 0) {'''
NEW = '''                    if (ABL_PERM_A && op == 0) {
                        if (level == 0 && !ABL_NO_BIAS) {
                            bsum[na] += raw[op][2 * pp];
                            asm("" : "+v"(bsum[na]));
                            bsum[na] += raw[op][2 * pp + 1];
                        }
                        const unsigned w0 = __builtin_bit_cast(unsigned, raw[op][2 * pp]);
                        const unsigned w1 = __builtin_bit_cast(unsigned, raw[op][2 * pp + 1]);
                        if (level == 1) nh[op][pp] = __builtin_bit_cast(h2, __builtin_amdgcn_perm(w1, w0, 0x05040100u));
                        if (level == 3) {
                            nl[op][pp] = __builtin_bit_cast(h2, __builtin_amdgcn_perm(w1, w0, 0x07060302u));
                            asm volatile("" : "+v"(nl[op][pp]));
                        }
                        return;
                    }
                    if (level == 0) {
                        float x0 = raw[op][2 * pp], x1 = raw[op][2 * pp + 1];
                        if (op > 0) {'''

VARIANTS = [("base", 0, 0), ("permA", 1, 0), ("permA_nb", 1, 1)]


def build(name, perm, nobias):
    from nerf_amd import build as B
    work = tempfile.mkdtemp(prefix="ablp_")
    csrc = os.path.join(work, "csrc")
    shutil.copytree(os.path.join(ROOT, "nerf_amd", "csrc"), csrc, ignore=shutil.ignore_patterns("*.so*"))
    p = os.path.join(csrc, "nerf_backward_common.h")
    s = open(p).read()
    assert s.count(OLD) == 1
    open(p, "w").write(s.replace(OLD, NEW))
    B.CSRC = csrc
    out = os.path.join(ROOT, "nerf_amd", "csrc", f"libnerf_hip_ablp_{name}.so")      # (travels with the snapshot)
    B.build(out=out, defines=[f"NERF_HIP_EXPERIMENT=ablp_{name}", f"ABL_PERM_A={perm}", f"ABL_NO_BIAS={nobias}"], force=True)
    return out


def main():
    if "--build-only" in sys.argv:
        for v in VARIANTS:
            print(build(*v))
        return
    for name, perm, nobias in VARIANTS * 2:
        lib = os.path.join(ROOT, "nerf_amd", "csrc", f"libnerf_hip_ablp_{name}.so")
        if not os.path.exists(lib):
            build(name, perm, nobias)
        out = os.path.join(ROOT, "gpurun_out", "ablp", name)
        shutil.rmtree(out, ignore_errors=True)
        subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", out, "-o", "run", "--",
                        "python3", os.path.join(ROOT, "scripts", "bench_train.py"), "4096", "f16x3"],
                       env=dict(os.environ, NERF_HIP_LIB=lib, TMPDIR="/tmp"), capture_output=True, cwd="/tmp")
        for row in csv.DictReader(open(os.path.join(out, "run_kernel_stats.csv"))):
            if "wgrad" in row["Name"] or "bwd_data" in row["Name"]:
                print(f"{name:9s} {row['Name'].split('::')[-1][:32]:32s} avg {float(row['AverageNs']) / 1e3:8.1f} us  "
                      f"min {float(row['MinNs']) / 1e3:8.1f} us", flush=True)


if __name__ == "__main__":
    main()
