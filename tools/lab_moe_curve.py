"""Lab: MoE expert GEMM (w1|w3) and whole layer over the token count (Mixtral shapes, int4 experts, top-2)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for tok in (16, 64, 128, 256, 512, 1024, 2048, 4096):
    g = bench.bench_moe_gemm(dev, 10, tokens=tok)
    l = bench.bench_moe_layer(dev, 10, tokens=tok)
    print(tok, "gemm us", round(g["us_per_call"], 1), round(g["TFLOPS"]), "TF", round(g["GBps"]), "GB/s | layer us",
          round(l["us_per_call"], 1), round(l["TFLOPS"]), "TF", flush=True)
