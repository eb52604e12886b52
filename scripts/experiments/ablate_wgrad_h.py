"""Timing-only ablations of the split-precision weight gradient (nerf_wgrad_h_kernel; WRONG results): what would it
gain if its operand conversions were free?  Builds variants of a copy of csrc/ in which the A operands (dY: scale,
split into f16 pairs), the B operands (x_hat: affine, ReLU, split) or both are taken as raw bit patterns from LDS —
same LDS reads, same MFMAs, same DMA stream, no conversion VALU — and prints the kernel's average duration from
`rocprofv3 --kernel-trace` of scripts/bench_train.py for each.
usage (GPU box): python scripts/experiments/ablate_wgrad_h.py"""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

OLD = '''                    Operand r;
                    split8(f32x4{v[0], v[1], v[2], v[3]}, f32x4{v[4], v[5], v[6], v[7]}, r.h, r.l);
                    if (op == 0) an = r;
                    else bnext[bq] = r;
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);'''
NEW = '''                    Operand r;
                    if ((op == 0 && ABL_A) || (op > 0 && ABL_B)) {
                        r.h = __builtin_bit_cast(h8, f32x4{raw[op][0], raw[op][1], raw[op][2], raw[op][3]});
                        r.l = __builtin_bit_cast(h8, f32x4{raw[op][4], raw[op][5], raw[op][6], raw[op][7]});
                    } else {
                        split8(f32x4{v[0], v[1], v[2], v[3]}, f32x4{v[4], v[5], v[6], v[7]}, r.h, r.l);
                    }
                    if (op == 0) an = r;
                    else bnext[bq] = r;
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);'''


def build(name, a, b):
    from nerf_amd import build as B
    work = tempfile.mkdtemp(prefix="ablw_")
    csrc = os.path.join(work, "csrc")
    shutil.copytree(os.path.join(ROOT, "nerf_amd", "csrc"), csrc, ignore=shutil.ignore_patterns("*.so*"))
    p = os.path.join(csrc, "nerf_backward_common.h")
    s = open(p).read()
    assert OLD in s
    open(p, "w").write(s.replace(OLD, NEW))
    B.CSRC = csrc
    out = os.path.join(ROOT, "nerf_amd", "csrc", f"libnerf_hip_ablw_{name}.so")      # (travels with the snapshot)
    B.build(out=out, defines=[f"NERF_HIP_EXPERIMENT=ablw_{name}", f"ABL_A={a}", f"ABL_B={b}"], force=True)
    return out


def main():
    variants = [("base", 0, 0), ("noA", 1, 0), ("noB", 0, 1), ("noAB", 1, 1)]
    if "--build-only" in sys.argv:
        for name, a, b in variants:
            print(build(name, a, b))
        return
    for name, a, b in variants * 2:
        lib = os.path.join(ROOT, "nerf_amd", "csrc", f"libnerf_hip_ablw_{name}.so")
        if not os.path.exists(lib):
            build(name, a, b)
        out = os.path.join(ROOT, "gpurun_out", "ablw", name)
        shutil.rmtree(out, ignore_errors=True)
        subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", out, "-o", "run", "--",
                        "python3", os.path.join(ROOT, "scripts", "bench_train.py"), "4096", "f16x3"],
                       env=dict(os.environ, NERF_HIP_LIB=lib, TMPDIR="/tmp"), capture_output=True, cwd="/tmp")
        import csv
        for row in csv.DictReader(open(os.path.join(out, "run_kernel_stats.csv"))):
            if "wgrad" in row["Name"] or "bwd_data" in row["Name"]:
                print(f"{name:5s} {row['Name'].split('::')[-1][:32]:32s} avg {float(row['AverageNs']) / 1e3:8.1f} us  "
                      f"min {float(row['MinNs']) / 1e3:8.1f} us", flush=True)


if __name__ == "__main__":
    main()
