"""Lab: time the MoE layer (mi355q_run_moe) at Mixtral shapes; run under rocprofv3 --kernel-trace for the per-kernel split."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for tok in (int(a) for a in (sys.argv[1:] or ["64", "512", "4096"])):
    print(bench.bench_moe_layer(dev, 10, tokens=tok), flush=True)
    print(bench.bench_moe_gemm(dev, 10, tokens=tok), flush=True)
