"""The legacy 8 x 256 network's 800 x 800 x 128 frame (bench.legacy_workload_timing), for the profiler:
python scripts/bench_legacy_frame.py [steps=2]"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
t = bench.legacy_workload_timing(torch.device("cuda:0"), steps=int(sys.argv[1]) if len(sys.argv) > 1 else 2)
print(json.dumps({"fp32_ms": t["ms_per_step"], "frac": t["roofline"]["frac"], "f16x3_ms": t["other_precision"]["ms_per_step"]}))
