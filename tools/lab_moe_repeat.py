"""Lab: run-to-run spread of the MoE layer bench lines (fresh tensors per call of the bench function)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for rnd in range(4):
    a = bench.bench_moe_layer(dev, 10, tokens=64)["us_per_call"]
    b = bench.bench_moe_layer(dev, 10, tokens=4096)["us_per_call"]
    print(rnd, "layer 64:", round(a, 1), "us   layer 4096:", round(b, 1), "us", flush=True)
