"""The hidden-64 / 16-scale training step with its per-kernel times (bench.train_step_timing): python scripts/bench_train_hidden64.py"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
t = bench.train_step_timing(torch.device("cuda:0"), hidden=64, enc=16)
print(json.dumps({"ms_per_step": t["ms_per_step"], "spread": t["ms_per_step_spread"], "kernels_ms": t["kernels_ms"]}))
