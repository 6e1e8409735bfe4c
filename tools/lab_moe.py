"""Lab: the grouped MoE GEMM lines of the bench."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for tok in (64, 512, 4096):
    r = bench.bench_moe_gemm(dev, 10, tokens=tok)
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()}, flush=True)
