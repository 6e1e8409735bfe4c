"""Lab: W4A16 prefill GEMM, 128 x 256 tiles on 4 waves (MI355Q_PF_NO256=1) against 256 x 256 tiles on 8 waves,
interleaved in one process (LAB library: the flag is read per call)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
for (m, n, k) in ((4096, 11008, 4096), (4096, 4096, 11008), (4096, 4096, 4096), (2048, 28672, 8192), (8192, 8192, 8192)):
    res = {"t128": [], "t256": []}
    for rnd in range(4):
        for tag in ("t128", "t256"):
            if tag == "t128":
                os.environ["MI355Q_PF_NO256"] = "1"
            else:
                os.environ.pop("MI355Q_PF_NO256", None)
            r = bench.bench_prefill_w4a16(dev, 20, m=m, n=n, k=k)
            res[tag].append(r["TFLOPS"])
    print(m, n, k, {t: [round(x) for x in v] for t, v in res.items()}, flush=True)
