"""Lab: W4A8-AWQ (FP8_ALPHA) through the group-wise plugin at decode / prefill sizes, against plain W4A16."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tensorrt-llm_amd"))
import torch, bench
from mi355q import functional as F
dev = torch.device("cuda", 0)
torch.manual_seed(0)
for dts, dt in (("float16", torch.float16), ("bfloat16", torch.bfloat16)):
    for (m, n, k) in ((1, 4096, 4096), (16, 11008, 4096), (64, 11008, 4096), (512, 11008, 4096), (4096, 4096, 4096), (4096, 11008, 4096),
                      (4096, 4096, 11008)):
        g = 128
        w = torch.randint(-128, 128, (k, n // 2), dtype=torch.int8, device=dev).view(dt).reshape(k, n // 4)
        sc = torch.rand(k // g, n, dtype=dt, device=dev)
        z = torch.randn(k // g, n, dtype=dt, device=dev)
        pq = torch.rand(1, k, dtype=dt, device=dev)
        al = torch.rand(1, dtype=torch.float32, device=dev)
        x = torch.randn(m, k, dtype=dt, device=dev)
        for algo, name in ((2 + 4, "w4a16 zero+prequant"), (8 + 2 + 4, "w4a8  zero+prequant")):
            f = lambda: F.weight_only_groupwise_quant_matmul(x, pq, w, sc, z, None, al, algo, g, dts)
            f()
            _, s = bench.time_steps(f, 20, 2, lambda: None)
            print(f"{dts} {name} m={m} n={n} k={k}: {s / 20 * 1e6:8.1f} us {2.0 * m * n * k / (s / 20) / 1e12:8.1f} TF", flush=True)
