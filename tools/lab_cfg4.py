"""Lab: the cfg4 decode slice (8 q heads on 1 int8 kv head, B=64, ctx 2048) alone."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for sets in (1, 12):
    r = bench.bench_decode_attention(dev, 480, Hq=8, Hkv=1, kv="int8", sets=sets)
    print(os.environ.get("MI355Q_LIB", "default")[-16:], os.environ.get("MI355Q_ATTN_SPLITS"), "sets", sets, round(r["us_per_call"], 2), "us", flush=True)
