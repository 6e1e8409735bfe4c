"""Lab: decode GEMV at the few-n-block TP slices (in-launch split-K of the light kernel against the persistent kernel)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
tag = "LIGHT=%s KS=%s" % (os.environ.get("MI355Q_GEMV_LIGHT"), os.environ.get("MI355Q_GEMV_LIGHT_KS"))
for (mm, nn, kk, ss) in ((1, 1280, 8192, 64), (8, 1280, 8192, 64), (16, 1280, 8192, 64), (1, 2560, 8192, 32), (16, 2560, 8192, 32), (1, 1024, 4096, 128), (1, 2048, 4096, 64), (1, 512, 16384, 64)):
    r = bench.bench_gemv(dev, 288, m=mm, n=nn, k=kk, sets=ss)
    print(tag, (mm, nn, kk), round(r["us_per_call"], 2), "us", flush=True)
