"""Lab: the decode-attention lines of bench.py's secondary block alone (cfg2 ragged, cfg4 slice, cfg5 fp8, 70B GQA bf16 / int8)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tensorrt-llm_amd"))
import torch
import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
cases = [dict(Hq=32, Hkv=32, kv="bf16", ragged=True), dict(Hq=8, Hkv=1, kv="int8", sets=12), dict(Hq=32, Hkv=8, kv="fp8", sets=2),
         dict(Hq=64, Hkv=8, kv="bf16"), dict(Hq=64, Hkv=8, kv="int8", sets=2), dict(Hq=64, Hkv=8, kv="fp8", sets=2),
         dict(Hq=32, Hkv=8, kv="int8", sets=2)]
for rep in range(2):
    for c in cases:
        r = bench.bench_decode_attention(dev, 96 if c.get("sets") == 12 else 48, **c)
        print(f"q{r['q_heads']:>2}/kv{r['kv_heads']:<2} {r['kv_cache']:>4} ctx {r['ctx']}: {r['us_per_call']:7.1f} us  {r['GBps']:7.0f} GB/s  {r['frac_of_8TBps'] * 100:5.1f} %", flush=True)
