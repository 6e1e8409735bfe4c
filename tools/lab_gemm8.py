"""Lab: int8 GEMM timing at cfg3 and a few other shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
tag = ("PHASED" if os.environ.get("MI355Q_GEMM8_PHASED") else "W128") + " abl=" + os.environ.get("MI355Q_GEMM8_ABL", "0")
shapes = ((4096, 12288, 5120), (8192, 8192, 8192)) if os.environ.get("SHORT") else ((4096, 13824, 5120), (4096, 12288, 5120), (8192, 8192, 8192), (4096, 4096, 4096), (2048, 13824, 5120))
for (m, n, k) in shapes:
    r = bench.bench_int8_gemm(dev, 20, m=m, n=n, k=k)
    print(tag, (m, n, k), {kk: (round(v, 3) if isinstance(v, float) else v) for kk, v in r.items() if kk in ("ms_per_call", "TOPS", "frac_of_int8_peak")}, flush=True)
