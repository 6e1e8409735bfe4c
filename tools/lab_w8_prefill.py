"""Lab: W8A16 / W4A16 per-channel prefill GEMM through the WeightOnlyQuantMatmul plugin (bf16), TFLOP/s."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tensorrt-llm_amd"))
import torch, bench
from mi355q import functional as F
dev = torch.device("cuda", 0)
for (m, n, k) in ((4096, 11008, 4096), (4096, 4096, 11008), (4096, 4096, 4096), (2048, 28672, 8192)):
    for wt in (1, 2):  # WeightOnlyQuantMatmul weightTypeId: 1 = int8, 2 = int4
        torch.manual_seed(0)
        w = torch.randint(-128, 128, (k, n if wt == 1 else n // 2), dtype=torch.int8, device=dev)
        sc = torch.rand(n, dtype=torch.bfloat16, device=dev)
        x = torch.randn(m, k, dtype=torch.bfloat16, device=dev)
        f = lambda: F.weight_only_quant_matmul(x, w, sc, wt, "bfloat16")
        f()
        _, dev_s = bench.time_steps(f, 20, 2, lambda: None)
        print(m, n, k, "int8" if wt == 1 else "int4", round(dev_s / 20 * 1e6, 1), "us", round(2.0 * m * n * k / (dev_s / 20) / 1e12), "TF", flush=True)
