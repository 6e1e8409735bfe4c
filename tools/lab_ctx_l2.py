"""Lab: is the MHA prefill at S = 2048 short of the GQA / long-sequence rate because an XCD's four concurrent heads
(4 x 1 MiB of K / V) fill its 4 MiB L2?  Per-tile time = ms * 512 slots / (workgroups * average tiles)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
for kw in (dict(B=4, H=32, Hkv=32, S=2048), dict(B=8, H=32, Hkv=32, S=1024), dict(B=2, H=32, Hkv=32, S=4096),
           dict(B=4, H=32, Hkv=8, S=2048), dict(B=4, H=32, Hkv=4, S=2048), dict(B=4, H=32, Hkv=32, S=2048, D=64)):
    r = bench.bench_context_attention(dev, 10, **kw)
    wgs = kw["B"] * kw["H"] * (kw["S"] // 128)
    tiles = (kw["S"] // 64 + 2) / 2
    print(kw, round(r["ms_per_call"] * 1e3, 1), "us", round(r["TFLOPS_causal"]), "TF  per-tile us:",
          round(r["ms_per_call"] * 1e3 * 512 / (wgs * tiles), 2), flush=True)
