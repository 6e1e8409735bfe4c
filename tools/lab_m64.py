import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for (mm, nn, kk, ss) in ((64, 3584, 8192, 24), (32, 3584, 8192, 24), (64, 8192, 3584, 24), (64, 4096, 4096, 24), (16, 3584, 8192, 24)):
    r = bench.bench_gemv(dev, 96, m=mm, n=nn, k=kk, sets=ss)
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k in ("m", "n", "k", "us_per_call", "GBps")}, flush=True)
