"""Lab: the MoE layer line of the bench over token counts, release library (what the shipped tile choice gives)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for tok in (int(a) for a in (sys.argv[1:] or ["64", "512", "768", "1024", "2048", "3072", "4096", "8192"])):
    ts = sorted(round(bench.bench_moe_layer(dev, 10, tokens=tok)["us_per_call"]) for _ in range(4))
    print(tok, ts, flush=True)
