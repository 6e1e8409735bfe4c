"""Lab: the grouped tile walk for the 128-row (4-wave) prefill tiles: MI355Q_PF_GROUP_M8 (dense launches) and
MI355Q_PF_GROUP_M_MOE (grouped MoE launches below the 256-row tile); LAB library, interleaved."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MI355Q_LIB", os.path.join(ROOT, "tensorrt-llm_amd", "lib", "libmi355q_lab.so"))
import torch, bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dense = [("w8a16 4096x11008x4096", lambda: bench.bench_prefill_w8a16(dev, 10)["ms_per_call"] * 1e3),
         ("w8a16 4096x4096x11008", lambda: bench.bench_prefill_w8a16(dev, 10, n=4096, k=11008)["ms_per_call"] * 1e3),
         ("w4a16 1000x11008x4096", lambda: bench.bench_prefill_w4a16(dev, 10, m=1000)["ms_per_call"] * 1e3),
         ("w4a16 512x4096x11008", lambda: bench.bench_prefill_w4a16(dev, 10, m=512, n=4096, k=11008)["ms_per_call"] * 1e3)]
moe = [("moe layer 512 tok", lambda: bench.bench_moe_layer(dev, 10, tokens=512)["us_per_call"]),
       ("moe layer 1024 tok", lambda: bench.bench_moe_layer(dev, 10, tokens=1024)["us_per_call"]),
       ("moe layer 2048 tok", lambda: bench.bench_moe_layer(dev, 10, tokens=2048)["us_per_call"])]
for var, cases in (("MI355Q_PF_GROUP_M8", dense), ("MI355Q_PF_GROUP_M_MOE", moe)):
    for name, fn in cases:
        res = {}
        for rep in range(3):
            for g in (0, 4, 8, 16):
                os.environ[var] = str(g)
                res.setdefault(g, []).append(round(fn()))
        print(name, res, flush=True)
