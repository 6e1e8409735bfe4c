"""Lab: rasterisation of the 256 x 256 prefill tiles -- groups of G m-tiles x all n-tiles (MI355Q_PF_GROUP_M=G; 0 = m
fastest over the whole launch), dense W4A16 prefill shapes, the grouped MoE GEMM and the MoE layer; LAB library,
interleaved."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MI355Q_LIB", os.path.join(ROOT, "tensorrt-llm_amd", "lib", "libmi355q_lab.so"))
import torch, bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
cases = [("w4a16 4096x11008x4096", lambda: bench.bench_prefill_w4a16(dev, 10)["ms_per_call"] * 1e3),
         ("w4a16 4096x4096x11008", lambda: bench.bench_prefill_w4a16(dev, 10, n=4096, k=11008)["ms_per_call"] * 1e3),
         ("w4a16 8192x8192x8192", lambda: bench.bench_prefill_w4a16(dev, 5, m=8192, n=8192, k=8192)["ms_per_call"] * 1e3),
         ("moe gemm 4096 tok", lambda: bench.bench_moe_gemm(dev, 10, tokens=4096)["us_per_call"]),
         ("moe layer 4096 tok", lambda: bench.bench_moe_layer(dev, 10, tokens=4096)["us_per_call"])]
for name, fn in cases:
    res = {}
    for rep in range(3):
        for g in (0, 2, 4, 8):
            os.environ["MI355Q_PF_GROUP_M"] = str(g)
            res.setdefault(g, []).append(round(fn()))
    print(name, res, flush=True)
