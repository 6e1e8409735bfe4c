"""Lab: one context-attention shape, few steps (PMC target)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
print(bench.bench_context_attention(dev, 3, B=4), flush=True)
