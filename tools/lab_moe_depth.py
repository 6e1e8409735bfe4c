"""Lab: decode-sized MoE expert GEMMs, 4-slot ring against the ping-pong (MI355Q_PF_DEPTH2=1), interleaved (LAB library)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for tok in (64, 16, 256):
    for rnd in range(3):
        for tag in ("ring4", "depth2"):
            if tag == "depth2":
                os.environ["MI355Q_PF_DEPTH2"] = "1"
            else:
                os.environ.pop("MI355Q_PF_DEPTH2", None)
            g = bench.bench_moe_gemm(dev, 20, tokens=tok)
            print(tok, tag, "gemm us", round(g["us_per_call"], 1), round(g["GBps"]), "GB/s", flush=True)
