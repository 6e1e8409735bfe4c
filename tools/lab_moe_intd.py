"""Lab: the grouped MoE GEMM at decode (64 tokens) with the integer-domain tiles against the exact ones
(MI355Q_MOE_EXACT=1), LAB library, interleaved."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MI355Q_LIB", os.path.join(ROOT, "tensorrt-llm_amd", "lib", "libmi355q_lab.so"))
import torch, bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for tok in (64, 256, 512, 1024, 4096):
    res = {"intd": [], "exact": []}
    for rep in range(4):
        for name in ("intd", "exact"):
            if name == "exact":
                os.environ["MI355Q_MOE_EXACT"] = "1"
            else:
                os.environ.pop("MI355Q_MOE_EXACT", None)
            r = bench.bench_moe_gemm(dev, 10, tokens=tok)
            res[name].append(r["us_per_call"])
    print(tok, {k: [round(x, 1) for x in v] for k, v in res.items()}, flush=True)
