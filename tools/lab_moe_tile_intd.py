"""Lab: the grouped w1 GEMM with 128-row integer-domain tiles (MI355Q_PF_NO256=1) against the 256 x 256 exact tile,
and 64-row against 128-row tiles, LAB library, interleaved."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MI355Q_LIB", os.path.join(ROOT, "tensorrt-llm_amd", "lib", "libmi355q_lab.so"))
import torch, bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for tok in (int(a) for a in (sys.argv[1:] or ["512", "1024", "2048", "4096"])):
    res = {}
    for rep in range(3):
        for name, env in (("default", {}), ("no256", {"MI355Q_PF_NO256": "1"}), ("mt4", {"MI355Q_MOE_MT": "4"}), ("mt8", {"MI355Q_MOE_MT": "8", "MI355Q_PF_NO256": "1"})):
            for k in ("MI355Q_PF_NO256", "MI355Q_MOE_MT"):
                os.environ.pop(k, None)
            os.environ.update(env)
            r = bench.bench_moe_gemm(dev, 10, tokens=tok)
            res.setdefault(name, []).append(round(r["us_per_call"], 1))
    print(tok, res, flush=True)
