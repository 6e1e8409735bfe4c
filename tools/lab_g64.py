"""Lab: group size 64 against 128 -- decode GEMV (m = 1, 4, 8, 16) and prefill GEMM (m = 4096), W4A16 with zeros."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
for g in (128, 64):
    for m in (1, 4, 8, 16):
        r = bench.bench_gemv(dev, 200, m=m, n=4096, k=4096, g=g, sets=48)
        print("gemv g", g, "m", m, round(r["us_per_call"], 2), "us", flush=True)
    for (m, n, k) in ((4096, 11008, 4096), (4096, 4096, 11008), (512, 11008, 4096)):
        r = bench.bench_prefill_w4a16(dev, 20, m=m, n=n, k=k, g=g)
        print("prefill g", g, m, n, k, round(r["ms_per_call"] * 1e3, 1), "us", round(r["TFLOPS"]), "TF", flush=True)
