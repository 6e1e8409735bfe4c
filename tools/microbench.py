#!/usr/bin/env python3
"""Quick GPU timings of single kernels (development aid; bench.py is the contract)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tensorrt-llm_amd"))
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("what", choices=["gemv", "gemm8", "attn"])
ap.add_argument("--m", type=int, default=1)
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--k", type=int, default=4096)
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
if a.what == "gemv":
    sets = max(2, min(72, int(600e6 // (a.n * a.k // 2))))
    print(json.dumps(bench.bench_gemv(dev, 4 * sets, m=a.m, n=a.n, k=a.k, sets=sets)))
elif a.what == "gemm8":
    print(json.dumps(bench.bench_int8_gemm(dev, a.steps, m=a.m, n=a.n, k=a.k)))
else:
    plg, inputs, out, ws, algo, keep = bench.build_attention_case(dev)
    wall, dev_s = bench.time_steps(lambda: plg.enqueue(inputs, [out], ws), a.steps, 3, lambda: None)
    print(json.dumps({"us": dev_s / a.steps * 1e6, "GBps": algo / (dev_s / a.steps) / 1e9}))
