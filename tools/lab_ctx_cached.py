"""Lab: the CACHED form of the context kernel at past = 0 against the workspace form, same workload (run under
rocprofv3 --kernel-trace --stats to split the prepare and attention kernels)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
for kw in (dict(B=4, H=32, Hkv=32, S=2048), dict(B=4, H=32, Hkv=32, S=2048, cached=True),
           dict(B=2, H=64, Hkv=8, S=4096), dict(B=2, H=64, Hkv=8, S=4096, cached=True)):
    r = bench.bench_context_attention(dev, 10, **kw)
    print(kw, round(r["ms_per_call"] * 1e3, 1), "us", round(r["TFLOPS_causal"]), "TF", flush=True)
