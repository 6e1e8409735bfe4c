"""Lab: a dense m x n x k W4A16 GEMM pushed through the grouped MoE launch with ONE expert (32-row integer-domain ring
tiles), to see what that kernel would give the dense 16 <= m <= 24 shapes."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for (m, n, k) in ((16, 28672, 8192), (24, 28672, 8192), (16, 4096, 4096), (16, 3584, 8192), (16, 22016, 4096)):
    r = bench.bench_moe_gemm(dev, 20, tokens=m, top_k=1, experts=1, n=n, k=k)
    print(m, n, k, round(r["us_per_call"], 1), "us", round(r["GBps"]), "GB/s", flush=True)
    if m == 16:
        r2 = bench.bench_gemv(dev, 40, m=m, n=n, k=k, sets=max(2, min(24, int(300e6 // (n * k // 2)))))
        print("   dense path:", round(r2["us_per_call"], 1), "us", flush=True)
