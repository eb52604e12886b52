"""Average rocprofv3 --pmc counters per dispatch of the render kernel.
usage: python scripts/pmc_summary.py DIR [DIR ...] [--kernel SUBSTR]  (DIRs = rocprofv3 -d outputs)"""
import csv, glob, json, os, sys
args = sys.argv[1:]
kernel = "nerf_render_fwd_kernel"
if "--kernel" in args:
    i = args.index("--kernel"); kernel = args[i + 1]; del args[i:i + 2]
out = {}
for d in args:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        with open(path) as f:
            for row in csv.DictReader(f):
                if kernel not in row["Kernel_Name"]:
                    continue
                acc.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
                acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
                for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size", "LDS_Block_Size"):
                    out[k] = row[k]
        for name, per in acc.items():
            out[name] = sum(per.values()) / len(per)
print(json.dumps(out, indent=1))
