"""Lab: W8A16 per-channel prefill GEMM, deferred scale against the exact dequantisation (MI355Q_W8_EXACT=1), LAB library."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MI355Q_LIB", os.path.join(ROOT, "tensorrt-llm_amd", "lib", "libmi355q_lab.so"))
import torch, bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for (m, n, k) in ((4096, 11008, 4096), (4096, 4096, 11008), (4096, 4096, 4096), (2048, 28672, 8192), (512, 11008, 4096)):
    res = {"defer": [], "exact": []}
    for rep in range(3):
        for name in ("defer", "exact"):
            if name == "exact":
                os.environ["MI355Q_W8_EXACT"] = "1"
            else:
                os.environ.pop("MI355Q_W8_EXACT", None)
            r = bench.bench_prefill_w8a16(dev, 10, m=m, n=n, k=k)
            res[name].append(round(r["TFLOPS"]))
    print(m, n, k, res, flush=True)
