"""Lab: int8 GEMM variants A/B, interleaved rounds in ONE process on one device (LAB build: switches read per call)."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from mi355q import functional as F
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
VARIANTS = {"w128": {"MI355Q_GEMM8_W128": "1"}, "phased": {}, "w128-noDMA-noLDS": {"MI355Q_GEMM8_W128": "1", "MI355Q_GEMM8_ABL": "3"}, "w128-noDMA": {"MI355Q_GEMM8_W128": "1", "MI355Q_GEMM8_ABL": "1"}}
KEYS = ("MI355Q_GEMM8_W128", "MI355Q_GEMM8_ABL")
shapes = [tuple(int(x) for x in s.split("x")) for s in os.environ.get("SHAPES", "4096x12288x5120,8192x8192x8192,4096x13824x5120").split(",")]
for (m, n, k) in shapes:
    torch.manual_seed(0)
    a = torch.randint(-128, 128, (m, k), dtype=torch.int8, device=dev)
    b = torch.randint(-128, 128, (n, k), dtype=torch.int8, device=dev)
    sa = (torch.randint(1, 10, (m, 1), device=dev).float() * 1e-2)
    sb = (torch.randint(1, 10, (1, n), device=dev).float() * 1e-2)
    plg = F.Plugin("SmoothQuantGemm", [("has_per_channel_scaling", 1, 5), ("has_per_token_scaling", 1, 5), ("type_id", 1, 5)])
    out = torch.empty(m, n, dtype=torch.float16, device=dev)
    res = {v: [] for v in VARIANTS}
    for rnd in range(int(os.environ.get("ROUNDS", 6))):
        for v, env in VARIANTS.items():
            for kk in KEYS:
                os.environ.pop(kk, None)
            os.environ.update(env)
            _, dev_s = bench.time_steps(lambda: plg.enqueue([a, b, sa, sb], [out]), 20, 2, lambda: None)
            res[v].append(2.0 * m * n * k / (dev_s / 20) / 1e12)
    for v, xs in res.items():
        print((m, n, k), f"{v:18s} median {statistics.median(xs):7.1f}  max {max(xs):7.1f}  min {min(xs):7.1f} TOPS", flush=True)
