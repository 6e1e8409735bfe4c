"""Lab: run only the new secondary bench lines."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for kw in (dict(Hq=32, Hkv=32, kv="bf16", ragged=True), dict(Hq=8, Hkv=1, kv="int8", sets=12), dict(Hq=32, Hkv=8, kv="fp8", sets=2),
           dict(Hq=64, Hkv=8, kv="bf16"), dict(Hq=64, Hkv=8, kv="int8", sets=2)):
    r = bench.bench_decode_attention(dev, 48, **kw)
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()}, flush=True)
for (mm, nn, kk, ss) in ((1, 1280, 8192, 64), (1, 3584, 8192, 24), (1, 8192, 1024, 72), (1, 8192, 3584, 24), (8, 3584, 8192, 24), (64, 3584, 8192, 24)):
    r = bench.bench_gemv(dev, 144, m=mm, n=nn, k=kk, sets=ss)
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()}, flush=True)
