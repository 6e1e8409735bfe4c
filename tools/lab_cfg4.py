import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for kw in (dict(Hq=8, Hkv=1, kv="int8", sets=12), dict(Hq=8, Hkv=1, kv="bf16", sets=6), dict(Hq=8, Hkv=1, kv="int8", sets=12, B=8), dict(Hq=32, Hkv=8, kv="fp8", sets=2)):
    r = bench.bench_decode_attention(dev, 96, **kw)
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k in ("batch", "q_heads", "kv_heads", "kv_cache", "us_per_call", "GBps")}, flush=True)
