"""A training step with its per-kernel times (bench.train_step_timing) at a chosen width:
python scripts/bench_train_width.py [hidden=64 enc=16 precision=fp32]"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
hidden = int(sys.argv[1]) if len(sys.argv) > 1 else 64
enc = int(sys.argv[2]) if len(sys.argv) > 2 else 16
prec = sys.argv[3] if len(sys.argv) > 3 else "fp32"
t = bench.train_step_timing(torch.device("cuda:0"), hidden=hidden, enc=enc, train_precision=prec)
print(json.dumps({"hidden": hidden, "enc": enc, "precision": prec, "ms_per_step": round(t["ms_per_step"], 4),
                  "min": round(t["ms_per_step_spread"]["min"], 4), "kernels_ms": {k: round(v, 4) for k, v in t["kernels_ms"].items()}}))
