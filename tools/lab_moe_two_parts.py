"""Lab: the grouped MoE launch with 256-row tiles in two parts (whole tiles + 64-row tail tiles) against one part
(MI355Q_MOE_ONE_PART=1: every expert's ragged last tile as a full 256-row tile), LAB library, interleaved."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MI355Q_LIB", os.path.join(ROOT, "tensorrt-llm_amd", "lib", "libmi355q_lab.so"))
import torch, bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for tok in (3072, 4096, 8192):
    res = {"two gemm": [], "one gemm": [], "two layer": [], "one layer": []}
    for rep in range(3):
        for name in ("two", "one"):
            os.environ.pop("MI355Q_MOE_ONE_PART", None)
            if name == "one":
                os.environ["MI355Q_MOE_ONE_PART"] = "1"
            res[name + " gemm"].append(bench.bench_moe_gemm(dev, 10, tokens=tok)["us_per_call"])
            res[name + " layer"].append(bench.bench_moe_layer(dev, 10, tokens=tok)["us_per_call"])
    print(tok, {k: [round(x) for x in v] for k, v in res.items()}, flush=True)
