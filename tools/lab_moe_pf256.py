"""Lab: MoE expert GEMMs / layer at prefill size, 128-row tiles (MI355Q_PF_NO256=1) against 256 x 256 tiles, interleaved
in one process (LAB library)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for tok in (4096, 2048):
    for rnd in range(3):
        for tag in ("t128", "t256"):
            if tag == "t128":
                os.environ["MI355Q_PF_NO256"] = "1"
            else:
                os.environ.pop("MI355Q_PF_NO256", None)
            g = bench.bench_moe_gemm(dev, 10, tokens=tok)
            l = bench.bench_moe_layer(dev, 10, tokens=tok)
            print(tok, tag, "gemm us", round(g["us_per_call"], 1), round(g["TFLOPS"]), "TF | layer us", round(l["us_per_call"], 1),
                  round(l["TFLOPS"]), "TF", flush=True)
