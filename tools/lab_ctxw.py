import sys, torch, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "."); sys.path.insert(0, "tensorrt-llm_amd")
import oracle
from test_gpu_context_window import *
dev = torch.device("cuda", 0)
dtype = kv_dtype = torch.float16
window, sink, tpb, lens = 64, 0, 64, [300, 40]
Hq, Hkv, D = 4, 2, 128
rows = make_inputs(lens, Hq, Hkv, D, dtype, seed=window)
cap_blocks = (window + tpb - 1) // tpb + 2
cache = oracle.PagedKVCache(2, Hkv, D, cap_blocks * tpb, tpb, kv_dtype)
cmap = oracle.CyclicMap(window, sink, tpb, cache.max_blocks)
pool = cache.pool.to(dev)
plg = plugin(Hq, Hkv, D, dtype, kv_dtype, tpb)
outs = enqueue(plg, rows, lens, cache, pool, dev, window=window, sink=sink)
refs = oracle.context_attention(rows, lens, Hq, Hkv, D, cache=cache, window=window, cmap=cmap)
for b in range(2):
    o, r = outs[b].float(), refs[b].float()
    print("seq", b, "nan rows out", torch.isnan(o).any(1).nonzero().flatten().tolist()[:40], "ref nan", torch.isnan(r).any())
    d = (o - r).abs().nan_to_num(0).max(1).values
    print(" bad rows", (d > 5e-3).nonzero().flatten().tolist()[:60], d.max().item())
