"""Lab: time the norm + quant feeder."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for kw in (dict(), dict(hidden=8192, sets=16), dict(tokens=64, hidden=4096, sets=64), dict(tokens=16384, hidden=4096, sets=8)):
    print(bench.bench_rmsnorm_quant(dev, 96, **kw), flush=True)
