"""Lab: context attention, shipped 4-wave 16x16x32 form against the 32x32x16 form (MI355Q_CTX_W32=1) and the 8-wave
ping-pong form (MI355Q_CTX_PP=1), LAB library, same workloads."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
forms = sys.argv[1].split(",") if len(sys.argv) > 1 else ["w16", "w32", "pp"]
for kw in (dict(B=4, H=32, Hkv=32, S=2048), dict(B=2, H=64, Hkv=8, S=4096), dict(B=1, H=32, Hkv=32, S=8192)):
    for form in forms * 2:
        os.environ.pop("MI355Q_CTX_W32", None)
        os.environ.pop("MI355Q_CTX_PP", None)
        if form == "w32":
            os.environ["MI355Q_CTX_W32"] = "1"
        if form == "pp":
            os.environ["MI355Q_CTX_PP"] = "1"
        r = bench.bench_context_attention(dev, 20, **kw)
        print(kw, form, round(r["ms_per_call"] * 1e3, 1), "us", round(r["TFLOPS_causal"]), "TF", flush=True)
