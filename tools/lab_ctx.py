"""Lab: time the context-attention kernels at a few shapes (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for kw in (dict(B=4), dict(B=1, S=4096), dict(B=2, H=64, Hkv=8, S=4096, kv="fp8"), dict(B=8, S=512), dict(B=4, D=64, S=2048)):
    print(bench.bench_context_attention(dev, 10, **kw), flush=True)
